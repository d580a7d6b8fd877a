// rdg_preprocess_bwd.hip -- per-Gaussian backward: conic -> Sigma2D -> (Sigma3D, mean),
// projection -> mean3D, SH backward (clamp mask), Sigma3D -> (scale, raw quaternion), and the camera-pose
// gradient dL/dviewmatrix (a reduction over all Gaussians: DPP wave sums -> per-workgroup partial rows ->
// fixed-order finalize; deterministic, no atomics).
//
// One thread per Gaussian.  HBM-bound: reads the 64-B accumulator row + the forward inputs, writes 59 floats.
// Pose-gradient gates (SURVEY.md §7 open question 4): enable_cov_grad switches the contribution through the
// EWA block (t and W), enable_sh_grad the contribution through campos; Gaussian gradients are unaffected.
#include "rdg_common.h"

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__device__ static const float BSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float BSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f};

// posebuf layout: [0..15] dL/dV (glm flat), [16..18] dL/dcampos
#define RDG_POSE_N 19

template <bool MULTI>
__global__ void __launch_bounds__(256)
rdg_preprocess_bwd_kernel(RdgDev d, const float* __restrict__ view_b, const float* __restrict__ proj,
                          const float* __restrict__ means3D_b, const float* __restrict__ shs,
                          const float* __restrict__ colors, const float* __restrict__ opac,
                          const float* __restrict__ scales, const float* __restrict__ rots_b,
                          const float* __restrict__ cov3Dp, const int32_t* __restrict__ radii_b,
                          const uint8_t* __restrict__ clampedm_b, const RdgRec* __restrict__ rec_b,
                          const float* __restrict__ grow_b,
                          float* __restrict__ posebuf_b, float* __restrict__ dmeans3D_b, float* __restrict__ dmeans2D_b,
                          float* __restrict__ dshs, float* __restrict__ dcolors, float* __restrict__ dopac_b,
                          float* __restrict__ dscales_b, float* __restrict__ drots_b, float* __restrict__ dcov3D,
                          int nviews_arg, int vstride_arg, RdgShAdam sh_adam) {
    const int nviews = MULTI ? nviews_arg : 1, vstride = MULTI ? vstride_arg : 0;   // MULTI = false: the single-camera kernel
    // nviews > 1 (sharded frame-DP owner stage): the same Gaussians under the cameras of a whole step; per-camera
    // inputs / outputs are stacked with a row stride of vstride, SH rows, scales and opacities are shared, and dL/dshs
    // [P,M,3] is the SUM over the cameras (camera 0 stores, the others add: the rows stay cache-resident in between).
    // MULTI runs 128-thread workgroups: besides the SH rows each wave keeps a second LDS tile that ACCUMULATES dL/dshs
    // over the cameras (25 KB per wave; 3 workgroups = 6 waves per CU), so the rows are staged once and the gradient is
    // written once however many cameras the step has.
    constexpr int BT = MULTI ? 128 : 256, NW = BT / 64;
    const int i = blockIdx.x * BT + threadIdx.x;
    // SH coefficients in, SH gradients out: staged per wave through LDS (rdg_rows_to_lds); each lane then works on
    // its own row IN PLACE -- every coefficient is read before its slot is overwritten with the gradient.
    __shared__ float sSH[NW][64 * 49];
    __shared__ float sGR[MULTI ? NW : 1][MULTI ? 64 * 49 : 1];
    // single camera: the SH gradient of a Gaussian stays FACTORED (16 basis values + 3 colour gradients, rdg_common.h) and
    // the staged SH values stay where they are -- the optimizer step at the end reads its parameters from LDS
    __shared__ __attribute__((aligned(16))) float sFac[MULTI ? 1 : NW][MULTI ? 4 : 64 * RDG_FAC];
    float* const myFac = MULTI ? nullptr : sFac[threadIdx.x >> 6] + (threadIdx.x & 63) * RDG_FAC;
    const int sh_row = d.M * 3, sh_stride = sh_row | 1;
    float* const mySH = sSH[threadIdx.x >> 6] + (threadIdx.x & 63) * sh_stride;
    float* const myGR = MULTI ? sGR[threadIdx.x >> 6] + (threadIdx.x & 63) * sh_stride : mySH;
    const long long wave_first = (long long)blockIdx.x * BT + (threadIdx.x >> 6) * 64;
    __shared__ float sPose[NW][RDG_POSE_N];
    if (MULTI && shs && wave_first < d.P) {
        rdg_rows_to_lds(shs, wave_first, d.P, sh_row, sh_stride, sSH[threadIdx.x >> 6], threadIdx.x & 63);
        for (int k = 0; k < sh_row; ++k) myGR[k] = 0.0f;
        rdg_wave_lds_sync();
    }
    for (int vw = 0; vw < nviews; ++vw) {
    const size_t vo = (size_t)vw * vstride;
    const float* __restrict__ view = view_b + 16 * vw;
    const float* __restrict__ means3D = means3D_b + vo * 3;
    const float* __restrict__ rots = rots_b ? rots_b + vo * 4 : nullptr;
    const int32_t* __restrict__ radii = radii_b + vo;
    const uint8_t* __restrict__ clampedm = clampedm_b + vo;
    const RdgRec* __restrict__ rec = rec_b + vo;
    const float* __restrict__ grow = grow_b + vo * RDG_GROW;
    float* __restrict__ posebuf = posebuf_b + (vo / BT) * RDG_POSE_N;
    float* __restrict__ dmeans3D = dmeans3D_b + vo * 3;
    float* __restrict__ dmeans2D = dmeans2D_b + vo * 3;
    float* __restrict__ dopac = dopac_b + vo;
    float* __restrict__ dscales = dscales_b ? dscales_b + vo * 3 : nullptr;
    float* __restrict__ drots = drots_b ? drots_b + vo * 4 : nullptr;
    if (!MULTI && shs && wave_first < d.P) {
        rdg_rows_to_lds(shs, wave_first, d.P, sh_row, sh_stride, sSH[threadIdx.x >> 6], threadIdx.x & 63);
        rdg_wave_lds_sync();
    }
    float V[16], Pm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = view[k]; Pm[k] = proj[k]; }
    float pose[RDG_POSE_N];
#pragma unroll
    for (int k = 0; k < RDG_POSE_N; ++k) pose[k] = 0.0f;

    const bool live = i < d.P && radii[i] > 0;
    if (i < d.P) {
        float dmx = 0.f, dmy = 0.f, dmz = 0.f;
        float gnx = 0.f, gny = 0.f, gop = 0.f;
        float dsc0 = 0.f, dsc1 = 0.f, dsc2 = 0.f;
        float dq0 = 0.f, dq1 = 0.f, dq2 = 0.f, dq3 = 0.f;
        float dS[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float grgb[3] = {0.f, 0.f, 0.f};
        if (live) {
            const float4* gr = reinterpret_cast<const float4*>(grow + (size_t)i * RDG_GROW);
            const float4 ga = gr[0], gb = gr[1], gc = gr[2];
            // the five geometric sums arrive divided by the opacity (a per-splat constant the compositing backward does
            // not multiply every pixel-splat pair with), as moments about (w, dy), w = dx + beta dy the conic's skew
            // coordinate (beta = conic_b / conic_a as the compositing stage staged it, rdg_stage_conic; see rdg_bwd_walk for
            // why): S_w, S_y, -1/2 S_ww, -S_wy, -1/2 S_yy.  With dx = w - beta dy:
            //   -1/2 S_xx = (-1/2 S_ww) - beta (-S_wy) + beta^2 (-1/2 S_yy),   -S_xy = (-S_wy) - 2 beta (-1/2 S_yy)
            const float o_ = opac[i];
            const float m1u = o_ * ga.x, m1y = o_ * ga.y;
            // the conic the compositing kernels used, bit for bit (the record), and their beta from it
            const float4 rq0 = rec[i].q0;
            const float rka = rq0.z, rkb = rq0.w;
            const float rA2 = -1.4426950408889634f * rka, rB = -1.4426950408889634f * rkb;
            const float beta_s = (rA2 < 0.0f) ? rB / rA2 : 0.0f;
            const float gcc = o_ * gb.x;
            const float gcb = fmaf(-2.0f * beta_s, gcc, o_ * ga.w);
            const float gca = fmaf(beta_s, fmaf(beta_s, gcc, -(o_ * ga.w)), o_ * ga.z);
            gop = gb.y;
            grgb[0] = gb.z; grgb[1] = gb.w; grgb[2] = gc.x;
            const float gdepth = gc.y;

            const float x = means3D[3 * i + 0], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
            const float vx = V[0] * x + V[4] * y + V[8] * z + V[12];
            const float vy = V[1] * x + V[5] * y + V[9] * z + V[13];
            const float vz = V[2] * x + V[6] * y + V[10] * z + V[14];
            // dL/dv: "all" feeds means3D; "pose" feeds the viewmatrix (may exclude the EWA path)
            float dvx = 0.f, dvy = 0.f, dvz = 0.f;      // common part (projection + depth)
            float evx = 0.f, evy = 0.f, evz = 0.f;      // EWA part (t in the Jacobian)

            // (2) depth
            dvz += gdepth;

            // recompute Sigma3D
            float S00, S01, S02, S11, S12, S22;
            float R00 = 0, R01 = 0, R02 = 0, R10 = 0, R11 = 0, R12 = 0, R20 = 0, R21 = 0, R22 = 0;
            float s0 = 0, s1 = 0, s2 = 0, qr = 0, qx = 0, qy = 0, qz = 0;
            if (cov3Dp) {
                S00 = cov3Dp[6 * i + 0]; S01 = cov3Dp[6 * i + 1]; S02 = cov3Dp[6 * i + 2];
                S11 = cov3Dp[6 * i + 3]; S12 = cov3Dp[6 * i + 4]; S22 = cov3Dp[6 * i + 5];
            } else {
                s0 = d.smod * scales[3 * i + 0]; s1 = d.smod * scales[3 * i + 1]; s2 = d.smod * scales[3 * i + 2];
                const float4 q = reinterpret_cast<const float4*>(rots)[i];
                qr = q.x; qx = q.y; qy = q.z; qz = q.w;
                R00 = 1.0f - 2.0f * (qy * qy + qz * qz); R01 = 2.0f * (qx * qy - qr * qz); R02 = 2.0f * (qx * qz + qr * qy);
                R10 = 2.0f * (qx * qy + qr * qz); R11 = 1.0f - 2.0f * (qx * qx + qz * qz); R12 = 2.0f * (qy * qz - qr * qx);
                R20 = 2.0f * (qx * qz - qr * qy); R21 = 2.0f * (qy * qz + qr * qx); R22 = 1.0f - 2.0f * (qx * qx + qy * qy);
                const float L00 = R00 * s0, L01 = R01 * s1, L02 = R02 * s2;
                const float L10 = R10 * s0, L11 = R11 * s1, L12 = R12 * s2;
                const float L20 = R20 * s0, L21 = R21 * s1, L22 = R22 * s2;
                S00 = L00 * L00 + L01 * L01 + L02 * L02; S01 = L00 * L10 + L01 * L11 + L02 * L12;
                S02 = L00 * L20 + L01 * L21 + L02 * L22; S11 = L10 * L10 + L11 * L11 + L12 * L12;
                S12 = L10 * L20 + L11 * L21 + L12 * L22; S22 = L20 * L20 + L21 * L21 + L22 * L22;
            }
            // recompute EWA
            const float limx = RDG_FOV_CLAMP * d.tanx, limy = RDG_FOV_CLAMP * d.tany;
            const float txtz = vx / vz, tytz = vy / vz;
            const bool clx = txtz < -limx || txtz > limx, cly = tytz < -limy || tytz > limy;
            const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
            const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
            const float iz = 1.0f / vz, iz2 = iz * iz, iz3 = iz2 * iz;
            const float J00 = d.fx * iz, J02 = -d.fx * tx * iz2, J11 = d.fy * iz, J12 = -d.fy * ty * iz2;
            const float W00 = V[0], W01 = V[4], W02 = V[8], W10 = V[1], W11 = V[5], W12 = V[9], W20 = V[2],
                        W21 = V[6], W22 = V[10];
            const float T00 = J00 * W00 + J02 * W20, T01 = J00 * W01 + J02 * W21, T02 = J00 * W02 + J02 * W22;
            const float T10 = J11 * W10 + J12 * W20, T11 = J11 * W11 + J12 * W21, T12 = J11 * W12 + J12 * W22;
            const float u00 = T00 * S00 + T01 * S01 + T02 * S02, u01 = T00 * S01 + T01 * S11 + T02 * S12,
                        u02 = T00 * S02 + T01 * S12 + T02 * S22;
            const float u10 = T10 * S00 + T11 * S01 + T12 * S02, u11 = T10 * S01 + T11 * S11 + T12 * S12,
                        u12 = T10 * S02 + T11 * S12 + T12 * S22;
            const float a = u00 * T00 + u01 * T01 + u02 * T02 + RDG_DILATION;
            const float b = u00 * T10 + u01 * T11 + u02 * T12;
            const float c = u10 * T10 + u11 * T11 + u12 * T12 + RDG_DILATION;
            const float det = a * c - b * b;
            // (1) ndc path.  The compositing backward hands over the FIRST MOMENTS of G dL/dG over the pixel offsets;
            // dL/dmean2D = -(conic . moments), and the pixel -> ndc factors 0.5 W, 0.5 H are applied here, once per
            // Gaussian, instead of per pixel-splat pair there (conic = Sigma2D^-1 = (c, -b, a) / det)
            if (det != 0.0f) {
                // conic . (m1x, m1y) with m1x = M_u - beta m1y:  ka M_u + (kb - ka beta) m1y,  kb M_u + (kc - kb beta) m1y --
                // the first bracket is the rounding residue of beta (an FMA gives it exactly), the second det(conic) / ka:
                // no term of the size of |conic| |m| is left to cancel
                gnx = -0.5f * (float)d.W * fmaf(rka, m1u, __fmaf_rn(-rka, beta_s, rkb) * m1y);
                // (kc - kb beta = det(conic) / ka = 1 / cov2D_yy: the record carries it, formed without the cancellation)
                gny = -0.5f * (float)d.H * fmaf(rkb, m1u, rec[i].q2.w * m1y);
            }
            {
                const float hx = Pm[0] * vx + Pm[4] * vy + Pm[8] * vz + Pm[12];
                const float hy = Pm[1] * vx + Pm[5] * vy + Pm[9] * vz + Pm[13];
                const float hw = Pm[3] * vx + Pm[7] * vy + Pm[11] * vz + Pm[15];
                const float pw = 1.0f / (hw + 1e-7f);
                const float dhx = gnx * pw, dhy = gny * pw;
                const float dhw = -(gnx * hx + gny * hy) * pw * pw;
                dvx += Pm[0] * dhx + Pm[1] * dhy + Pm[3] * dhw;
                dvy += Pm[4] * dhx + Pm[5] * dhy + Pm[7] * dhw;
                dvz += Pm[8] * dhx + Pm[9] * dhy + Pm[11] * dhw;
            }
            // (3)-(4) conic -> cov2D -> (Sigma3D, T)
            // With scale and rotation given (the only form RoDyGS uses) everything below goes through the PROJECTED AXES
            // u_k = T r_k of the Gaussian (the columns of M = T R S are s_k u_k, so cov2D = sum_k s_k^2 u_k u_k^T + 0.3 I):
            //   det        = sum_{j<l} s_j^2 s_l^2 (u_j x u_l)^2 + 0.3 sum_k s_k^2 |u_k|^2 + 0.09      (Cauchy-Binet: positive terms;
            //                a c - b^2 loses log10(a c / det) digits on a needle-shaped footprint: 3 of 7 at det / (a c) = 1.6e-3)
            //   dL/dcov2D  = G = di adj(Gk) + ddet adj(cov2D),  Gk = dL/dconic,  di = 1 / det,  ddet = -ddi di^2,
            //   ddi        = sum_k s_k^2 B_k + 0.3 (gca + gcc),   B_k = u_k^T adj(Gk) u_k
            //   u_k^T adj(cov2D) u_k = sum_{j != k} s_j^2 (u_j x u_k)^2 + 0.3 |u_k|^2 = A_k:  the axis' own rank-one term drops out
            //                EXACTLY (u_k x u_k = 0) -- formed from the matrix (c u0^2 - 2 b u0 u1 + a u1^2) it is what the other
            //                terms are the small remainder of, for the long axis of a needle
            //   dL/ds_k    = 2 s_k (di B_k + ddet A_k),   G u_k = di adj(Gk) u_k + ddet (sum_{j != k} s_j^2 p_j (p_j . u_k) + 0.3 u_k),
            //                p_j = (u_j1, -u_j0);   dL/dR[:, k] = 2 s_k^2 T^T (G u_k);   dL/dT = 2 sum_k s_k^2 (G u_k) r_k^T.
            // Strict sweep, anisotropic profile (pancakes and needles 10-300x, tests/sweep_cases.py), case 400000 / 349: the scale
            // gradient of a 300 x 6 pixel needle was 4.5e-4 off with float64-exact input rows (the float32 oracle 6e-5) while
            // dL/dcov2D was formed as three numbers and contracted with the axes; finding 3 of DESIGN.md section 2 was the first half
            // of this (the contraction), this is the second (det and the adjugate).
            float da = 0.f, db = 0.f, dc = 0.f;
            float dT00 = 0.f, dT01 = 0.f, dT02 = 0.f, dT10 = 0.f, dT11 = 0.f, dT12 = 0.f;
            float dR00 = 0.f, dR01 = 0.f, dR02 = 0.f, dR10 = 0.f, dR11 = 0.f, dR12 = 0.f, dR20 = 0.f, dR21 = 0.f, dR22 = 0.f;
            if (cov3Dp) {
                // a precomputed covariance has no axes: the matrix form, in the association reverse-mode differentiation of
                // conic = (c, -b, a) / det produces (through 1 / det, then det = a c - b^2; NOT the closed forms
                // da = (-c^2 gca + b c gcb - b^2 gcc) / det^2 ..., which add three terms of the size of c^2 g that cancel)
                if (det != 0.0f) {
                    const float di = 1.0f / det;
                    const float ddi = (gca * c - gcb * b) + gcc * a;          // dL/d(1/det)
                    const float ddet = -(ddi * di) * di;
                    da = gcc * di + ddet * c;
                    dc = gca * di + ddet * a;
                    db = -gcb * di - 2.0f * b * ddet;
                }
                // (4) cov2D -> Sigma3D (6-vector grads count both off-diagonal entries) and T
                dS[0] = T00 * T00 * da + T00 * T10 * db + T10 * T10 * dc;
                dS[3] = T01 * T01 * da + T01 * T11 * db + T11 * T11 * dc;
                dS[5] = T02 * T02 * da + T02 * T12 * db + T12 * T12 * dc;
                dS[1] = 2.0f * T00 * T01 * da + (T00 * T11 + T01 * T10) * db + 2.0f * T10 * T11 * dc;
                dS[2] = 2.0f * T00 * T02 * da + (T00 * T12 + T02 * T10) * db + 2.0f * T10 * T12 * dc;
                dS[4] = 2.0f * T01 * T02 * da + (T01 * T12 + T02 * T11) * db + 2.0f * T11 * T12 * dc;
                dT00 = 2.0f * u00 * da + u10 * db; dT01 = 2.0f * u01 * da + u11 * db; dT02 = 2.0f * u02 * da + u12 * db;
                dT10 = 2.0f * u10 * dc + u00 * db; dT11 = 2.0f * u11 * dc + u01 * db; dT12 = 2.0f * u12 * dc + u02 * db;
            } else if (det != 0.0f) {
                // projected axes
                const float a0x = T00 * R00 + T01 * R10 + T02 * R20, a0y = T10 * R00 + T11 * R10 + T12 * R20;
                const float a1x = T00 * R01 + T01 * R11 + T02 * R21, a1y = T10 * R01 + T11 * R11 + T12 * R21;
                const float a2x = T00 * R02 + T01 * R12 + T02 * R22, a2y = T10 * R02 + T11 * R12 + T12 * R22;
                const float q0 = s0 * s0, q1 = s1 * s1, q2 = s2 * s2;
                const float n0 = a0x * a0x + a0y * a0y, n1 = a1x * a1x + a1y * a1y, n2 = a2x * a2x + a2y * a2y;
                const float x01 = a0x * a1y - a0y * a1x, x02 = a0x * a2y - a0y * a2x, x12 = a1x * a2y - a1y * a2x;
                const float detp = ((q0 * q1) * (x01 * x01) + (q0 * q2) * (x02 * x02) + (q1 * q2) * (x12 * x12)) +
                                   RDG_DILATION * (q0 * n0 + q1 * n1 + q2 * n2) + RDG_DILATION * RDG_DILATION;
                const float di = 1.0f / detp;
                const float hb = 0.5f * gcb;
                // adj(Gk) u_k and B_k
                const float e0x = gcc * a0x - hb * a0y, e0y = gca * a0y - hb * a0x;
                const float e1x = gcc * a1x - hb * a1y, e1y = gca * a1y - hb * a1x;
                const float e2x = gcc * a2x - hb * a2y, e2y = gca * a2y - hb * a2x;
                const float B0 = a0x * e0x + a0y * e0y, B1 = a1x * e1x + a1y * e1y, B2 = a2x * e2x + a2y * e2y;
                const float ddi = (q0 * B0 + q1 * B1 + q2 * B2) + RDG_DILATION * (gca + gcc);
                const float ddet = -(ddi * di) * di;
                // adj(cov2D) u_k without the axis' own term:  sum_{j != k} s_j^2 p_j (p_j . u_k) + 0.3 u_k,  p_j . u_k = -(u_j x u_k)
                // (p_0 . u_1 = -x01, p_1 . u_0 = x01, p_0 . u_2 = -x02, p_2 . u_0 = x02, p_1 . u_2 = -x12, p_2 . u_1 = x12)
                const float c0x = (q1 * x01) * a1y + (q2 * x02) * a2y + RDG_DILATION * a0x;
                const float c0y = -(q1 * x01) * a1x - (q2 * x02) * a2x + RDG_DILATION * a0y;
                const float c1x = -(q0 * x01) * a0y + (q2 * x12) * a2y + RDG_DILATION * a1x;
                const float c1y = (q0 * x01) * a0x - (q2 * x12) * a2x + RDG_DILATION * a1y;
                const float c2x = -(q0 * x02) * a0y - (q1 * x12) * a1y + RDG_DILATION * a2x;
                const float c2y = (q0 * x02) * a0x + (q1 * x12) * a1x + RDG_DILATION * a2y;
                const float A0 = (q1 * (x01 * x01) + q2 * (x02 * x02)) + RDG_DILATION * n0;
                const float A1 = (q0 * (x01 * x01) + q2 * (x12 * x12)) + RDG_DILATION * n1;
                const float A2 = (q0 * (x02 * x02) + q1 * (x12 * x12)) + RDG_DILATION * n2;
                // G u_k
                const float g0x = di * e0x + ddet * c0x, g0y = di * e0y + ddet * c0y;
                const float g1x = di * e1x + ddet * c1x, g1y = di * e1y + ddet * c1y;
                const float g2x = di * e2x + ddet * c2x, g2y = di * e2y + ddet * c2y;
                dsc0 = d.smod * (2.0f * s0) * (di * B0 + ddet * A0);
                dsc1 = d.smod * (2.0f * s1) * (di * B1 + ddet * A1);
                dsc2 = d.smod * (2.0f * s2) * (di * B2 + ddet * A2);
                const float f0 = 2.0f * q0, f1 = 2.0f * q1, f2 = 2.0f * q2;
                dR00 = f0 * (T00 * g0x + T10 * g0y); dR10 = f0 * (T01 * g0x + T11 * g0y); dR20 = f0 * (T02 * g0x + T12 * g0y);
                dR01 = f1 * (T00 * g1x + T10 * g1y); dR11 = f1 * (T01 * g1x + T11 * g1y); dR21 = f1 * (T02 * g1x + T12 * g1y);
                dR02 = f2 * (T00 * g2x + T10 * g2y); dR12 = f2 * (T01 * g2x + T11 * g2y); dR22 = f2 * (T02 * g2x + T12 * g2y);
                // dL/dT = 2 sum_k s_k^2 (G u_k) r_k^T
                dT00 = (f0 * g0x) * R00 + (f1 * g1x) * R01 + (f2 * g2x) * R02;
                dT01 = (f0 * g0x) * R10 + (f1 * g1x) * R11 + (f2 * g2x) * R12;
                dT02 = (f0 * g0x) * R20 + (f1 * g1x) * R21 + (f2 * g2x) * R22;
                dT10 = (f0 * g0y) * R00 + (f1 * g1y) * R01 + (f2 * g2y) * R02;
                dT11 = (f0 * g0y) * R10 + (f1 * g1y) * R11 + (f2 * g2y) * R12;
                dT12 = (f0 * g0y) * R20 + (f1 * g1y) * R21 + (f2 * g2y) * R22;
            }
            // (5) T = J W
            const float dJ00 = W00 * dT00 + W01 * dT01 + W02 * dT02;
            const float dJ02 = W20 * dT00 + W21 * dT01 + W22 * dT02;
            const float dJ11 = W10 * dT10 + W11 * dT11 + W12 * dT12;
            const float dJ12 = W20 * dT10 + W21 * dT11 + W22 * dT12;
            if (d.cov_grad) {
                // dL/dW[i][j] -> V[j*4+i]
                pose[0] += J00 * dT00;  pose[4] += J00 * dT01;  pose[8] += J00 * dT02;
                pose[1] += J11 * dT10;  pose[5] += J11 * dT11;  pose[9] += J11 * dT12;
                pose[2] += J02 * dT00 + J12 * dT10;
                pose[6] += J02 * dT01 + J12 * dT11;
                pose[10] += J02 * dT02 + J12 * dT12;
            }
            // (6) J -> t (clamped t.x / t.y are constants, as in the public 3DGS backward)
            evx = clx ? 0.0f : -d.fx * iz2 * dJ02;
            evy = cly ? 0.0f : -d.fy * iz2 * dJ12;
            evz = -d.fx * iz2 * dJ00 - d.fy * iz2 * dJ11 + 2.0f * d.fx * tx * iz3 * dJ02 + 2.0f * d.fy * ty * iz3 * dJ12;

            // (9) colour
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;  // dL/d(p - campos)
            if (colors) {
                // handled below (dcolors)
            } else {
                const uint8_t cl = clampedm[i];
                const float gr0 = (cl & 1) ? 0.f : grgb[0], gr1 = (cl & 2) ? 0.f : grgb[1],
                            gr2 = (cl & 4) ? 0.f : grgb[2];
                const float gcol[3] = {gr0, gr1, gr2};
                const float camx = -(V[0] * V[12] + V[1] * V[13] + V[2] * V[14]);
                const float camy = -(V[4] * V[12] + V[5] * V[13] + V[6] * V[14]);
                const float camz = -(V[8] * V[12] + V[9] * V[13] + V[10] * V[14]);
                const float ox = x - camx, oy = y - camy, oz = z - camz;
                const float ln = sqrtf(ox * ox + oy * oy + oz * oz);
                const float il = 1.0f / ln;
                const float ux = ox * il, uy = oy * il, uz = oz * il;
                const float* sh = mySH;
                float* dsh = myGR;          // several cameras: accumulated over them in the second tile
                // single camera: only the basis values are kept (bk), the gradient is their product with gcol
                float bk[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) bk[k] = 0.0f;
                bk[0] = SH_C0;
#define RDG_DSH(idx, val) do { if (MULTI) dsh[idx] += (val); } while (0)
                float dRx = 0.f, dRy = 0.f, dRz = 0.f;  // dL/d(unit dir)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) RDG_DSH(ch, SH_C0 * gcol[ch]);
                if (d.deg > 0) {
                    const float b1 = -SH_C1 * uy, b2 = SH_C1 * uz, b3 = -SH_C1 * ux;
                    bk[1] = b1; bk[2] = b2; bk[3] = b3;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float g = gcol[ch];
                        const float h1 = sh[3 + ch], h2 = sh[6 + ch], h3 = sh[9 + ch];
                        RDG_DSH(3 + ch, b1 * g); RDG_DSH(6 + ch, b2 * g); RDG_DSH(9 + ch, b3 * g);
                        dRx += -SH_C1 * h3 * g;
                        dRy += -SH_C1 * h1 * g;
                        dRz += SH_C1 * h2 * g;
                    }
                    if (d.deg > 1) {
                        const float xx = ux * ux, yy = uy * uy, zz = uz * uz, xy = ux * uy, yz = uy * uz, xz = ux * uz;
                        const float b4 = BSH_C2[0] * xy, b5 = BSH_C2[1] * yz, b6 = BSH_C2[2] * (2.0f * zz - xx - yy),
                                    b7 = BSH_C2[3] * xz, b8 = BSH_C2[4] * (xx - yy);
                        bk[4] = b4; bk[5] = b5; bk[6] = b6; bk[7] = b7; bk[8] = b8;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            const float g = gcol[ch];
                            const float h4 = sh[12 + ch], h5 = sh[15 + ch], h6 = sh[18 + ch], h7 = sh[21 + ch],
                                        h8 = sh[24 + ch];
                            RDG_DSH(12 + ch, b4 * g); RDG_DSH(15 + ch, b5 * g); RDG_DSH(18 + ch, b6 * g);
                            RDG_DSH(21 + ch, b7 * g); RDG_DSH(24 + ch, b8 * g);
                            dRx += (BSH_C2[0] * uy * h4 + BSH_C2[2] * -2.0f * ux * h6 + BSH_C2[3] * uz * h7 +
                                    BSH_C2[4] * 2.0f * ux * h8) * g;
                            dRy += (BSH_C2[0] * ux * h4 + BSH_C2[1] * uz * h5 + BSH_C2[2] * -2.0f * uy * h6 +
                                    BSH_C2[4] * -2.0f * uy * h8) * g;
                            dRz += (BSH_C2[1] * uy * h5 + BSH_C2[2] * 4.0f * uz * h6 + BSH_C2[3] * ux * h7) * g;
                        }
                        if (d.deg > 2) {
                            const float b9 = BSH_C3[0] * uy * (3.0f * xx - yy), b10 = BSH_C3[1] * xy * uz,
                                        b11 = BSH_C3[2] * uy * (4.0f * zz - xx - yy),
                                        b12 = BSH_C3[3] * uz * (2.0f * zz - 3.0f * xx - 3.0f * yy),
                                        b13 = BSH_C3[4] * ux * (4.0f * zz - xx - yy), b14 = BSH_C3[5] * uz * (xx - yy),
                                        b15 = BSH_C3[6] * ux * (xx - 3.0f * yy);
                            bk[9] = b9; bk[10] = b10; bk[11] = b11; bk[12] = b12; bk[13] = b13; bk[14] = b14; bk[15] = b15;
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) {
                                const float g = gcol[ch];
                                const float h9 = sh[27 + ch], h10 = sh[30 + ch], h11 = sh[33 + ch], h12 = sh[36 + ch],
                                            h13 = sh[39 + ch], h14 = sh[42 + ch], h15 = sh[45 + ch];
                                RDG_DSH(27 + ch, b9 * g); RDG_DSH(30 + ch, b10 * g); RDG_DSH(33 + ch, b11 * g);
                                RDG_DSH(36 + ch, b12 * g); RDG_DSH(39 + ch, b13 * g); RDG_DSH(42 + ch, b14 * g);
                                RDG_DSH(45 + ch, b15 * g);
                                dRx += (BSH_C3[0] * h9 * 3.0f * 2.0f * xy + BSH_C3[1] * h10 * yz +
                                        BSH_C3[2] * h11 * -2.0f * xy + BSH_C3[3] * h12 * -3.0f * 2.0f * xz +
                                        BSH_C3[4] * h13 * (-3.0f * xx + 4.0f * zz - yy) +
                                        BSH_C3[5] * h14 * 2.0f * xz + BSH_C3[6] * h15 * 3.0f * (xx - yy)) * g;
                                dRy += (BSH_C3[0] * h9 * 3.0f * (xx - yy) + BSH_C3[1] * h10 * xz +
                                        BSH_C3[2] * h11 * (-3.0f * yy + 4.0f * zz - xx) +
                                        BSH_C3[3] * h12 * -3.0f * 2.0f * yz + BSH_C3[4] * h13 * -2.0f * xy +
                                        BSH_C3[5] * h14 * -2.0f * yz + BSH_C3[6] * h15 * -3.0f * 2.0f * xy) * g;
                                dRz += (BSH_C3[1] * h10 * xy + BSH_C3[2] * h11 * 4.0f * 2.0f * yz +
                                        BSH_C3[3] * h12 * 3.0f * (2.0f * zz - xx - yy) +
                                        BSH_C3[4] * h13 * 4.0f * 2.0f * xz + BSH_C3[5] * h14 * (xx - yy)) * g;
                            }
                        }
                    }
                }
                // (the basis values of the inactive degrees stay 0: their coefficients get a zero gradient)
                if (!MULTI) {
                    float4* f4 = reinterpret_cast<float4*>(myFac);
                    f4[0] = make_float4(bk[0], bk[1], bk[2], bk[3]);     f4[1] = make_float4(bk[4], bk[5], bk[6], bk[7]);
                    f4[2] = make_float4(bk[8], bk[9], bk[10], bk[11]);   f4[3] = make_float4(bk[12], bk[13], bk[14], bk[15]);
                    f4[4] = make_float4(gcol[0], gcol[1], gcol[2], 0.0f);
                }
#undef RDG_DSH
                // unit-vector normalisation backward
                const float dotg = ux * dRx + uy * dRy + uz * dRz;
                ddx = (dRx - ux * dotg) * il; ddy = (dRy - uy * dotg) * il; ddz = (dRz - uz * dotg) * il;
                if (d.sh_grad) { pose[16] -= ddx; pose[17] -= ddy; pose[18] -= ddz; }
            }

            // (7) v = V p
            const float avx = dvx + evx, avy = dvy + evy, avz = dvz + evz;  // for means3D
            dmx = V[0] * avx + V[1] * avy + V[2] * avz + ddx;
            dmy = V[4] * avx + V[5] * avy + V[6] * avz + ddy;
            dmz = V[8] * avx + V[9] * avy + V[10] * avz + ddz;
            const float pvx = d.cov_grad ? avx : dvx, pvy = d.cov_grad ? avy : dvy, pvz = d.cov_grad ? avz : dvz;
            pose[0] += x * pvx;  pose[1] += x * pvy;  pose[2] += x * pvz;
            pose[4] += y * pvx;  pose[5] += y * pvy;  pose[6] += y * pvz;
            pose[8] += z * pvx;  pose[9] += z * pvy;  pose[10] += z * pvz;
            pose[12] += pvx;     pose[13] += pvy;     pose[14] += pvz;

            // (8) rotation matrix -> raw quaternion (dL/dR and dL/dscale were formed in (3)-(4))
            if (!cov3Dp) {
                dq0 = 2.0f * (-qz * dR01 + qy * dR02 + qz * dR10 - qx * dR12 - qy * dR20 + qx * dR21);
                dq1 = 2.0f * (qy * dR01 + qz * dR02 + qy * dR10 - 2.0f * qx * dR11 - qr * dR12 + qz * dR20 + qr * dR21 -
                              2.0f * qx * dR22);
                dq2 = 2.0f * (-2.0f * qy * dR00 + qx * dR01 + qr * dR02 + qx * dR10 + qz * dR12 - qr * dR20 + qz * dR21 -
                              2.0f * qy * dR22);
                dq3 = 2.0f * (-2.0f * qz * dR00 - qr * dR01 + qx * dR02 + qr * dR10 - 2.0f * qz * dR11 + qy * dR12 +
                              qx * dR20 + qy * dR21);
            }
        }
        if (shs && !MULTI && (!live || colors)) {     // no SH gradient for this Gaussian: all factors zero
            float4* f4 = reinterpret_cast<float4*>(myFac);
#pragma unroll
            for (int q = 0; q < 5; ++q) f4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (!MULTI && live && (unsigned)(i - d.dn_row0) < (unsigned)d.dn_rows) {
            // densification statistics of the reference's train loop (RdgRasterSettings.densify_*): dL/dmean2D and the
            // radius are in registers here -- 12 B of read-modify-write per visible Gaussian instead of a pass of its own
            const int j = i - d.dn_row0;
            if (d.dn_maxr) d.dn_maxr[j] = fmaxf(d.dn_maxr[j], (float)radii[i]);
            if (d.dn_accum) d.dn_accum[j] += sqrtf(gnx * gnx + gny * gny);
            if (d.dn_denom) d.dn_denom[j] += 1.0f;
        }
        dmeans3D[3 * i + 0] = dmx; dmeans3D[3 * i + 1] = dmy; dmeans3D[3 * i + 2] = dmz;
        dmeans2D[3 * i + 0] = gnx; dmeans2D[3 * i + 1] = gny; dmeans2D[3 * i + 2] = 0.f;
        dopac[i] = gop;
        if (colors) {
            dcolors[3 * i + 0] = grgb[0]; dcolors[3 * i + 1] = grgb[1]; dcolors[3 * i + 2] = grgb[2];
        }
        if (cov3Dp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) dcov3D[6 * i + k] = dS[k];
        } else {
            dscales[3 * i + 0] = dsc0; dscales[3 * i + 1] = dsc1; dscales[3 * i + 2] = dsc2;
            reinterpret_cast<float4*>(drots)[i] = make_float4(dq0, dq1, dq2, dq3);
        }
    }
    if (shs && wave_first < d.P && vw == nviews - 1) {
        rdg_wave_lds_sync();
        if (MULTI)
            rdg_lds_to_rows(dshs, wave_first, d.P, sh_row, sh_stride, sGR[threadIdx.x >> 6], threadIdx.x & 63);
        else if (sh_adam.m)            // optimizer in backward: parameters from the staged tile, gradient from the factors
            rdg_lds_adam_rows_fac(const_cast<float*>(shs), rdg_sh_adam_resolve(sh_adam), wave_first, d.P, sh_row, sh_stride,
                                  sSH[threadIdx.x >> 6], sFac[threadIdx.x >> 6], threadIdx.x & 63);
        else
            rdg_lds_to_rows_fac(dshs, wave_first, d.P, sh_row, sFac[threadIdx.x >> 6], threadIdx.x & 63);
    }
    // pose-gradient reduction: DPP wave sums -> LDS -> ONE partial row per workgroup (no atomics: 16 k waves
    // hammering the same 19 addresses ran 20x slower than the rest of the kernel, and this form is deterministic)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < RDG_POSE_N; ++k) {
        const float sm = rdg_wave_sum_to63(pose[k]);
        if (lane == 63) sPose[wv][k] = sm;
    }
    __syncthreads();
    if (threadIdx.x < RDG_POSE_N)
        posebuf[(size_t)blockIdx.x * RDG_POSE_N + threadIdx.x] =
            MULTI ? sPose[0][threadIdx.x] + sPose[NW - 1][threadIdx.x]
                  : (sPose[0][threadIdx.x] + sPose[1][threadIdx.x]) + (sPose[2][threadIdx.x] + sPose[NW - 1][threadIdx.x]);
    if (nviews > 1) __syncthreads();       // sPose is reused by the next camera
    }
}

// Fixed-order reduction of the per-workgroup partial rows in two small launches (a single workgroup walking all
// ~4 k rows was a 19 us dependent-load chain): RDG_POSE_L2 workgroups fold every RDG_POSE_L2-th row into one
// second-level row each, then one wave adds those and applies the campos = -R^T t chain, writing dL/dviewmatrix
// (glm flat).  Deterministic: the summation order is fixed by the row indices alone.
#define RDG_POSE_L2 32
__global__ void __launch_bounds__(256)
rdg_pose_partial_kernel(const float* __restrict__ posebuf, int nblk, float* __restrict__ part, int view_rows = 0) {
    // blockIdx.y = camera (multi-view launches): its partial rows start view_rows rows further on
    posebuf += (size_t)blockIdx.y * view_rows * RDG_POSE_N;
    part += (size_t)blockIdx.y * RDG_POSE_L2 * RDG_POSE_N;
    __shared__ float sred[8][32];
    const int k = threadIdx.x & 31, grp = threadIdx.x >> 5;   // 8 row groups x 32 component slots
    float acc = 0.0f;
    if (k < RDG_POSE_N) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const int step = RDG_POSE_L2 * 8;
        int r = blockIdx.x * 8 + grp;
        for (; r + 3 * step < nblk; r += 4 * step) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += posebuf[(size_t)(r + step * u) * RDG_POSE_N + k];
        }
        for (; r < nblk; r += step) a[0] += posebuf[(size_t)r * RDG_POSE_N + k];
        acc = (a[0] + a[1]) + (a[2] + a[3]);
    }
    sred[grp][k] = acc;
    __syncthreads();
    if (threadIdx.x < RDG_POSE_N) {
        float t = 0.0f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += sred[g][threadIdx.x];
        part[blockIdx.x * RDG_POSE_N + threadIdx.x] = t;
    }
}

__global__ void __launch_bounds__(64)
rdg_pose_finalize_kernel(const float* __restrict__ view, const float* __restrict__ part, int nrows,
                         float* __restrict__ dview) {
    view += 16 * blockIdx.x; dview += 16 * blockIdx.x;        // blockIdx.x = camera
    part += (size_t)blockIdx.x * nrows * RDG_POSE_N;
    __shared__ float stot[32];
    if (threadIdx.x < RDG_POSE_N) {
        float t = 0.0f;
        for (int r = 0; r < nrows; ++r) t += part[r * RDG_POSE_N + threadIdx.x];
        stot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float g[16], gc[3], V[16];
    for (int c = 0; c < RDG_POSE_N; ++c) {
        if (c < 16) g[c] = stot[c]; else gc[c - 16] = stot[c];
    }
    for (int c = 0; c < 16; ++c) V[c] = view[c];
    // cam_j = -(V[4j+0]*V[12] + V[4j+1]*V[13] + V[4j+2]*V[14])
    for (int j = 0; j < 3; ++j) {
        for (int r = 0; r < 3; ++r) {
            g[4 * j + r] += -V[12 + r] * gc[j];
            g[12 + r] += -V[4 * j + r] * gc[j];
        }
    }
    for (int c = 0; c < 16; ++c) dview[c] = g[c];
}

// RdgRasterSettings.aux_stream of the call in flight on this thread (set by the C-ABI wrapper right before the launch, consumed
// by it: one shot), and the event the fork goes through (one per thread, created on first use -- rdg_pose_fork_prepare() lets a
// caller do that outside a stream capture).
static thread_local hipStream_t g_pose_aux = nullptr;
static thread_local hipEvent_t g_pose_fork_ev = nullptr;
void rdg_set_pose_aux(hipStream_t s) { g_pose_aux = s; }
int rdg_pose_fork_event_ready() {
    if (!g_pose_fork_ev && hipEventCreateWithFlags(&g_pose_fork_ev, hipEventDisableTiming) != hipSuccess) {
        g_pose_fork_ev = nullptr;
        return rdg_set_error("pose fork: hipEventCreate failed");
    }
    return 0;
}

int rdg_launch_preprocess_bwd(const RdgDev& d, const float* means3D, const float* shs, const float* colors,
                              const float* opac, const float* scales, const float* rots, const float* cov3D,
                              const float* view, const float* proj, const int32_t* radii, const void* geom_ws,
                              const float* grow, float* posebuf, float* dmeans3D, float* dmeans2D, float* dshs,
                              float* dcolors, float* dopac, float* dscales, float* drots, float* dcov3D,
                              float* dview, hipStream_t s, const RdgShAdam* sh_adam) {
    const RdgGeomLayout G = rdg_geom_layout(d.P);
    const int nblk = (d.P + 255) / 256;
    if (d.P > 0) {
        hipLaunchKernelGGL(rdg_preprocess_bwd_kernel<false>, dim3(nblk), dim3(256), 0, s, d, view, proj, means3D, shs, colors,
                           opac, scales, rots, cov3D, radii, (const uint8_t*)((const char*)geom_ws + G.clamped),
                           (const RdgRec*)((const char*)geom_ws + G.rec), grow, posebuf, dmeans3D, dmeans2D, dshs, dcolors, dopac, dscales, drots, dcov3D, 1, 0,
                           sh_adam ? *sh_adam : RdgShAdam{});
    }
    // second-level rows live behind the per-workgroup rows (rdg_grad_bytes reserves them)
    float* part = (float*)((char*)posebuf + rdg_align_up((size_t)(nblk > 0 ? nblk : 1) * RDG_POSE_N * 4, 256));
    const int rows = d.P > 0 ? nblk : 0;
    hipStream_t ps = s;
    hipStream_t aux = g_pose_aux;
    g_pose_aux = nullptr;
    if (aux && aux != s) {
        // fork: the pose chain behind an event on `s`, on the caller's second stream (a graph branch under capture)
        if (rdg_pose_fork_event_ready()) return -1;
        if (hipEventRecord(g_pose_fork_ev, s) != hipSuccess || hipStreamWaitEvent(aux, g_pose_fork_ev, 0) != hipSuccess)
            return rdg_set_error("pose fork: event record / wait failed");
        ps = aux;
    }
    hipLaunchKernelGGL(rdg_pose_partial_kernel, dim3(RDG_POSE_L2), dim3(256), 0, ps, posebuf, rows, part);
    hipLaunchKernelGGL(rdg_pose_finalize_kernel, dim3(1), dim3(64), 0, ps, view, part, RDG_POSE_L2, dview);
    return rdg_check_hip(hipGetLastError(), "preprocess_bwd launch");
}

int rdg_launch_preprocess_bwd_views(const RdgDev& d, int32_t nviews, int32_t stride, const float* means3D,
                                    const float* shs, const float* opac, const float* scales, const float* rots,
                                    const float* views, const float* proj, const int32_t* radii, const void* geom_ws,
                                    const float* grow, float* posebuf, float* dmeans3D, float* dmeans2D, float* dshs,
                                    float* dopac, float* dscales, float* drots, hipStream_t s) {
    const RdgGeomLayout G = rdg_geom_layout(nviews * stride);
    const int nblk = (d.P + 127) / 128;
    if (d.P > 0)
        hipLaunchKernelGGL(rdg_preprocess_bwd_kernel<true>, dim3(nblk), dim3(128), 0, s, d, views, proj, means3D, shs,
                           (const float*)nullptr, opac, scales, rots, (const float*)nullptr, radii,
                           (const uint8_t*)((const char*)geom_ws + G.clamped),
                           (const RdgRec*)((const char*)geom_ws + G.rec), grow, posebuf, dmeans3D, dmeans2D, dshs,
                           (float*)nullptr, dopac, dscales, drots, (float*)nullptr, nviews, stride, RdgShAdam{});
    return rdg_check_hip(hipGetLastError(), "preprocess_bwd views launch");
}

// pose-gradient reduction of all slices in two launches (camera v: partial rows [v*view_rows, v*view_rows + nblk))
int rdg_launch_pose_reduce_views(int nviews, int view_rows, int nblk, const float* views, float* posebuf, float* part,
                                 float* dviews, hipStream_t s) {
    hipLaunchKernelGGL(rdg_pose_partial_kernel, dim3(RDG_POSE_L2, nviews), dim3(256), 0, s, posebuf, nblk, part, view_rows);
    hipLaunchKernelGGL(rdg_pose_finalize_kernel, dim3(nviews), dim3(64), 0, s, views, part, RDG_POSE_L2, dviews);
    return rdg_check_hip(hipGetLastError(), "pose reduce views launch");
}
