// rdg_preprocess_fwd.hip -- per-Gaussian forward stage.
//
// BUILT WITH -ffp-contract=off: the arithmetic below is evaluated left to right with separately rounded
// multiplies and adds, exactly as oracle/rasterizer_oracle.py::preprocess writes it, so view-space depth bits,
// radii and tile rectangles (=> tile keys, sort order) are bit-exact against the oracle.  The kernel is
// HBM-bound (236 B read per Gaussian at SH-3), so giving up FMA contraction costs nothing.
//
// One thread per Gaussian, 256-thread blocks.  Besides the 64-B splat record the block leaves its
// tiles_touched sum in block_sums[] so the scan needs no extra pass over P.
#include "rdg_common.h"

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__device__ static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                          -1.0925484305920792f, 0.5462742152960396f};
__device__ static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                          0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                          -0.5900435899266435f};

__device__ __forceinline__ void rdg_rect(float px, float py, int radius, int gx, int gy, int& x0, int& y0,
                                         int& x1, int& y1) {
    float r = (float)radius;
    x0 = min(gx, max(0, (int)((px - r) / (float)RDG_TILE)));
    y0 = min(gy, max(0, (int)((py - r) / (float)RDG_TILE)));
    x1 = min(gx, max(0, (int)((((px + r) + (float)RDG_TILE) - 1.0f) / (float)RDG_TILE)));
    y1 = min(gy, max(0, (int)((((py + r) + (float)RDG_TILE) - 1.0f) / (float)RDG_TILE)));
}

// ln(x) for a positive normal float from +, -, *, / and bit operations ONLY, in a fixed order (this file is built without
// FMA contraction): the oracle restates it operation for operation (oracle/rasterizer_oracle.py::_ln_f32), so the tight
// rectangles -- hence the culled key stream -- are bit-exact against it, which no library logarithm would be.
// x = m 2^e with m in [0.707, 1.414]; ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.172: five terms, 1e-9 absolute.
__device__ __forceinline__ float rdg_ln_exact_ops(float x) {
    const uint32_t u = __float_as_uint(x);
    int e = (int)(u >> 23) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f);
    const float s2 = s * s;
    float p = 0.111111111f * s2 + 0.142857143f;
    p = p * s2 + 0.2f;
    p = p * s2 + 0.333333333f;
    p = p * s2 + 1.0f;
    return (2.0f * s) * p + (float)e * 0.693147181f;
}

// The tile rectangle [x0, x1) x [y0, y1) of a splat, from its RECORD alone (so that a rank which received the record over the
// wire forms the same rectangle bit for bit: rdg_geom_from_records) -- the one place of the library that forms it; every later
// stage reads it from the per-Gaussian `rectd` array.
//   cull = 0: the reference's rule -- the square of half-width `radius` = ceil(3 sqrt(lambda_max)) around the pixel centre.
//             This is what the exported (tile | depth) key stream is held to bit for bit (rdg_bin_forward, the oracle).
//   cull = 1: that rectangle INTERSECTED with the tiles that hold a pixel centre inside the axis-aligned box of the splat's
//             "alpha >= 1/255" ellipse.  A pixel blends the splat only where opacity * exp(-Q / 2) >= 1/255 (and Q >= 0), Q the
//             quadratic form the compositing kernels evaluate, Q = a (dx + beta dy)^2 + dy^2 / cov2D_yy: i.e. Q <= r2 =
//             2 ln(255 opacity).  Over all dx the form is >= dy^2 / cov_yy, over all dy it is >= dx^2 / cov_xx with cov_xx =
//             c_eff cov_yy / a (c_eff = b^2 / a + 1 / cov_yy, the third conic entry as the staged form has it), so the ellipse
//             lies in |dx| <= sqrt(r2 cov_xx), |dy| <= sqrt(r2 cov_yy).  r2 carries the margin of rdg_quadrant_bits (0.02 in the
//             log + 1e-3 relative: orders of magnitude above the rounding of these few operations), so no tile with a blending
//             pixel is ever dropped: image, final_T and every gradient are those of the reference rectangle; only the
//             instances that could not blend anywhere in their tile never enter the lists.  The reference's 3-sigma square is
//             a circle's box around an ellipse that is often thin, and often ends well inside 3 sigma (r2 < 9 for every
//             opacity below 0.35): on the bench frame 27 % of the reference's (tile, Gaussian) instances (32 % at 100 k points)
//             go, of the 32 % (41 %) an exact per-tile test could remove.  A splat that cannot reach 1/255 anywhere
//             (255 opacity < 1) keeps its radius (the reference's visibility) and gets no tile.
__device__ __forceinline__ void rdg_splat_rect(const float4 q0, const float4 q1, const float inv_cyy, int gx, int gy, int cull,
                                               int& x0, int& y0, int& x1, int& y1) {
    rdg_rect(q0.x, q0.y, __float_as_int(q1.w), gx, gy, x0, y0, x1, y1);
    if (!cull) return;
    const float a = q0.z, b = q0.w, o = q1.y;
    const float t255 = 255.0f * o;
    if (!(t255 >= 0.99f)) { x1 = x0; y1 = y0; return; }      // never reaches 1/255 (a NaN opacity is dropped too)
    const float r2 = (2.0f * (rdg_ln_exact_ops(t255) + 0.02f)) * 1.001f;
    const float cyy = 1.0f / inv_cyy;
    const float c_eff = (b * b) / a + inv_cyy;
    const float cxx = (c_eff * cyy) / a;
    const float hx = sqrtf(r2 * cxx), hy = sqrtf(r2 * cyy);
    // pixel columns / rows with a centre inside the box: ceil(p - h) .. floor(p + h), clamped to the tile grid's pixels.
    // Written so that a NaN extent keeps the reference rectangle (fmaxf / fminf return the other operand).
    const float lox = fmaxf(ceilf(q0.x - hx), 0.0f), hix = fminf(floorf(q0.x + hx), (float)(gx * RDG_TILE - 1));
    const float loy = fmaxf(ceilf(q0.y - hy), 0.0f), hiy = fminf(floorf(q0.y + hy), (float)(gy * RDG_TILE - 1));
    if (hix < lox || hiy < loy) { x1 = x0; y1 = y0; return; }  // no pixel centre inside the box
    x0 = max(x0, (int)lox >> 4); x1 = min(x1, ((int)hix >> 4) + 1);
    y0 = max(y0, (int)loy >> 4); y1 = min(y1, ((int)hiy >> 4) + 1);
    if (x1 < x0) x1 = x0;
    if (y1 < y0) y1 = y0;
}
__device__ __forceinline__ uint4 rdg_pack_rectd(int x0, int y0, int x1, int y1, float depth, uint32_t tiles) {
    if (tiles == 0u) return make_uint4(0u, 0u, 0u, 0u);
    return make_uint4((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16), __float_as_uint(depth), tiles);
}

template <bool MULTI>
__global__ void __launch_bounds__(RDG_PRE_BLOCK)
rdg_preprocess_fwd_kernel(RdgDev d, const float* __restrict__ view_b, const float* __restrict__ proj,
                          const float* __restrict__ means3D_b, const float* __restrict__ shs,
                          const float* __restrict__ colors, const float* __restrict__ opac,
                          const float* __restrict__ scales, const float* __restrict__ rots_b,
                          const float* __restrict__ cov3Dp, RdgRec* __restrict__ rec_b,
                          uint32_t* __restrict__ tiles_touched_b, uint8_t* __restrict__ clampedm_b,
                          uint32_t* __restrict__ block_sums_b, int32_t* __restrict__ radii_b, int nviews_arg,
                          int vstride_arg, int blk0_arg, uint4* __restrict__ rectd_b) {
    const int nviews = MULTI ? nviews_arg : 1, vstride = MULTI ? vstride_arg : 0;   // MULTI = false: the single-camera kernel
    // row-range launches of the multi-camera kernel (chunked owner stage, pipelined with the exchange): workgroup
    // blockIdx.x works on rows [(blk0 + blockIdx.x) * RDG_PRE_BLOCK, ...) and d.P is the END row of the range
    const int bid = (MULTI ? blk0_arg : 0) + (int)blockIdx.x;
    // nviews > 1 (sharded frame-DP owner stage): the same Gaussians under the cameras of a whole step.  Time-dependent
    // inputs (means3D, rotations) and all outputs are stacked per camera with a row stride of vstride (multiple of
    // 256); SH rows, scales and opacities are shared -- the SH rows are staged into LDS once for all cameras.
    const int i = bid * RDG_PRE_BLOCK + threadIdx.x;
    // SH rows of the wave's 64 Gaussians: staged through LDS with wave-contiguous loads (rdg_rows_to_lds)
    __shared__ float sSH[RDG_PRE_BLOCK / 64][64 * 49];
    const int sh_row = d.M * 3, sh_stride = sh_row | 1;
    const long long wave_first = (long long)bid * RDG_PRE_BLOCK + (threadIdx.x >> 6) * 64;
    if (shs && wave_first < d.P) {
        rdg_rows_to_lds(shs, wave_first, d.P, sh_row, sh_stride, sSH[threadIdx.x >> 6], threadIdx.x & 63);
        rdg_wave_lds_sync();
    }
    // camera: uniform addresses -> scalar loads into SGPRs (the matrices live on the device because the
    // viewmatrix is the output of autograd-tracked pose math; no host round trip)
    __shared__ uint32_t wsum[RDG_PRE_BLOCK / RDG_WAVE];
    for (int vw = 0; vw < nviews; ++vw) {
    const size_t vo = (size_t)vw * vstride;
    const float* __restrict__ view = view_b + 16 * vw;
    const float* __restrict__ means3D = means3D_b + vo * 3;
    const float* __restrict__ rots = rots_b ? rots_b + vo * 4 : nullptr;
    RdgRec* __restrict__ rec = rec_b + vo;
    uint32_t* __restrict__ tiles_touched = tiles_touched_b + vo;
    uint8_t* __restrict__ clampedm = clampedm_b + vo;
    int32_t* __restrict__ radii = radii_b + vo;
    uint32_t* __restrict__ block_sums = block_sums_b + vo / RDG_PRE_BLOCK;
    uint4* __restrict__ rectd = rectd_b + vo;
    float V[16], Pm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { V[k] = view[k]; Pm[k] = proj[k]; }
    uint32_t my_tiles = 0;
    if (i < d.P) {
        int radius_out = 0;
        uint8_t cl = 0;
        uint4 rd = make_uint4(0u, 0u, 0u, 0u);
        RdgRec R;
        R.q0 = make_float4(0.f, 0.f, 0.f, 0.f);
        R.q1 = R.q0; R.q2 = R.q0; R.q3 = R.q0;
        const float x = means3D[3 * i + 0], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
        const float vx = ((V[0] * x + V[4] * y) + V[8] * z) + V[12];
        const float vy = ((V[1] * x + V[5] * y) + V[9] * z) + V[13];
        const float vz = ((V[2] * x + V[6] * y) + V[10] * z) + V[14];
        bool valid = vz > RDG_NEAR_CULL;
        if (valid) {
            const float hx = ((Pm[0] * vx + Pm[4] * vy) + Pm[8] * vz) + Pm[12];
            const float hy = ((Pm[1] * vx + Pm[5] * vy) + Pm[9] * vz) + Pm[13];
            const float hw = ((Pm[3] * vx + Pm[7] * vy) + Pm[11] * vz) + Pm[15];
            const float pw = 1.0f / (hw + 1e-7f);
            const float ndc_x = hx * pw, ndc_y = hy * pw;

            float S00, S01, S02, S11, S12, S22;
            float nx = 0.f, ny = 0.f, nz = 0.f;
            if (cov3Dp) {
                S00 = cov3Dp[6 * i + 0]; S01 = cov3Dp[6 * i + 1]; S02 = cov3Dp[6 * i + 2];
                S11 = cov3Dp[6 * i + 3]; S12 = cov3Dp[6 * i + 4]; S22 = cov3Dp[6 * i + 5];
            } else {
                const float s0 = d.smod * scales[3 * i + 0], s1 = d.smod * scales[3 * i + 1],
                            s2 = d.smod * scales[3 * i + 2];
                const float4 q = reinterpret_cast<const float4*>(rots)[i];
                const float r = q.x, qx = q.y, qy = q.z, qz = q.w;
                const float R00 = 1.0f - 2.0f * (qy * qy + qz * qz), R01 = 2.0f * (qx * qy - r * qz),
                            R02 = 2.0f * (qx * qz + r * qy);
                const float R10 = 2.0f * (qx * qy + r * qz), R11 = 1.0f - 2.0f * (qx * qx + qz * qz),
                            R12 = 2.0f * (qy * qz - r * qx);
                const float R20 = 2.0f * (qx * qz - r * qy), R21 = 2.0f * (qy * qz + r * qx),
                            R22 = 1.0f - 2.0f * (qx * qx + qy * qy);
                const float L00 = R00 * s0, L01 = R01 * s1, L02 = R02 * s2;
                const float L10 = R10 * s0, L11 = R11 * s1, L12 = R12 * s2;
                const float L20 = R20 * s0, L21 = R21 * s1, L22 = R22 * s2;
                S00 = (L00 * L00 + L01 * L01) + L02 * L02;
                S01 = (L00 * L10 + L01 * L11) + L02 * L12;
                S02 = (L00 * L20 + L01 * L21) + L02 * L22;
                S11 = (L10 * L10 + L11 * L11) + L12 * L12;
                S12 = (L10 * L20 + L11 * L21) + L12 * L22;
                S22 = (L20 * L20 + L21 * L21) + L22 * L22;
                if (d.render_normal) {
                    // shortest axis (first minimum) of R*diag(s), in view space, facing the camera
                    const float sc0 = scales[3 * i + 0], sc1 = scales[3 * i + 1], sc2 = scales[3 * i + 2];
                    int k = 0; float sm = sc0;
                    if (sc1 < sm) { sm = sc1; k = 1; }
                    if (sc2 < sm) { sm = sc2; k = 2; }
                    const float n0 = k == 0 ? R00 : (k == 1 ? R01 : R02);
                    const float n1 = k == 0 ? R10 : (k == 1 ? R11 : R12);
                    const float n2 = k == 0 ? R20 : (k == 1 ? R21 : R22);
                    float nvx = (V[0] * n0 + V[4] * n1) + V[8] * n2;
                    float nvy = (V[1] * n0 + V[5] * n1) + V[9] * n2;
                    float nvz = (V[2] * n0 + V[6] * n1) + V[10] * n2;
                    const float dotv = (nvx * vx + nvy * vy) + nvz * vz;
                    const float sg = dotv > 0.f ? -1.0f : 1.0f;
                    nx = nvx * sg; ny = nvy * sg; nz = nvz * sg;
                }
            }
            // EWA 2-D covariance
            const float limx = RDG_FOV_CLAMP * d.tanx, limy = RDG_FOV_CLAMP * d.tany;
            const float txtz = vx / vz, tytz = vy / vz;
            const float tx = fminf(limx, fmaxf(-limx, txtz)) * vz;
            const float ty = fminf(limy, fmaxf(-limy, tytz)) * vz;
            const float J00 = d.fx / vz, J02 = -(d.fx * tx) / (vz * vz);
            const float J11 = d.fy / vz, J12 = -(d.fy * ty) / (vz * vz);
            const float T00 = J00 * V[0] + J02 * V[2], T01 = J00 * V[4] + J02 * V[6], T02 = J00 * V[8] + J02 * V[10];
            const float T10 = J11 * V[1] + J12 * V[2], T11 = J11 * V[5] + J12 * V[6], T12 = J11 * V[9] + J12 * V[10];
            const float u00 = (T00 * S00 + T01 * S01) + T02 * S02;
            const float u01 = (T00 * S01 + T01 * S11) + T02 * S12;
            const float u02 = (T00 * S02 + T01 * S12) + T02 * S22;
            const float u10 = (T10 * S00 + T11 * S01) + T12 * S02;
            const float u11 = (T10 * S01 + T11 * S11) + T12 * S12;
            const float u12 = (T10 * S02 + T11 * S12) + T12 * S22;
            const float ca = ((u00 * T00 + u01 * T01) + u02 * T02) + RDG_DILATION;
            const float cb = (u00 * T10 + u01 * T11) + u02 * T12;
            const float cc = ((u10 * T10 + u11 * T11) + u12 * T12) + RDG_DILATION;
            const float det = ca * cc - cb * cb;
            valid = det != 0.0f;
            if (valid) {
                const float det_inv = 1.0f / det;
                const float mid = 0.5f * (ca + cc);
                const float disc = sqrtf(fmaxf(RDG_LAMBDA_FLOOR, mid * mid - det));
                const float lam = fmaxf(mid + disc, mid - disc);
                const int radius = (int)ceilf(3.0f * sqrtf(lam));
                const float px = ((ndc_x + 1.0f) * (float)d.W - 1.0f) * 0.5f;
                const float py = ((ndc_y + 1.0f) * (float)d.H - 1.0f) * 0.5f;
                // record fields the rectangle is formed from (rdg_splat_rect reads nothing else)
                const float4 rq0 = make_float4(px, py, cc * det_inv, -cb * det_inv);
                const float4 rq1 = make_float4(ca * det_inv, opac[i], vz, __int_as_float(radius));
                const float inv_cyy = 1.0f / cc;
                int x0, y0, x1, y1;
                rdg_rect(px, py, radius, d.gx, d.gy, x0, y0, x1, y1);
                const int area = (x1 - x0) * (y1 - y0);
                if (area > 0) {
                    // visible by the reference's rule (radius > 0: renderer.py:111); the binning stage may see fewer tiles
                    if (d.cull) rdg_splat_rect(rq0, rq1, inv_cyy, d.gx, d.gy, 1, x0, y0, x1, y1);
                    my_tiles = (uint32_t)((x1 - x0) * (y1 - y0));
                    rd = rdg_pack_rectd(x0, y0, x1, y1, vz, my_tiles);
                    radius_out = radius;
                    float cr, cg, cbl;
                    if (colors) {
                        cr = colors[3 * i + 0]; cg = colors[3 * i + 1]; cbl = colors[3 * i + 2];
                    } else {
                        const float camx = -((V[0] * V[12] + V[1] * V[13]) + V[2] * V[14]);
                        const float camy = -((V[4] * V[12] + V[5] * V[13]) + V[6] * V[14]);
                        const float camz = -((V[8] * V[12] + V[9] * V[13]) + V[10] * V[14]);
                        float dx = x - camx, dy = y - camy, dz = z - camz;
                        const float ln = sqrtf((dx * dx + dy * dy) + dz * dz);
                        dx = dx / ln; dy = dy / ln; dz = dz / ln;
                        const float* sh = sSH[threadIdx.x >> 6] + (threadIdx.x & 63) * sh_stride;
                        float res[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) res[c] = SH_C0 * sh[c];
                        if (d.deg > 0) {
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                res[c] = res[c] - SH_C1 * dy * sh[3 + c] + SH_C1 * dz * sh[6 + c] -
                                         SH_C1 * dx * sh[9 + c];
                            if (d.deg > 1) {
                                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                                const float xy = dx * dy, yz = dy * dz, xz = dx * dz;
#pragma unroll
                                for (int c = 0; c < 3; ++c)
                                    res[c] = res[c] + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] +
                                             SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] +
                                             SH_C2[3] * xz * sh[21 + c] + SH_C2[4] * (xx - yy) * sh[24 + c];
                                if (d.deg > 2) {
#pragma unroll
                                    for (int c = 0; c < 3; ++c)
                                        res[c] = res[c] + SH_C3[0] * dy * (3.0f * xx - yy) * sh[27 + c] +
                                                 SH_C3[1] * xy * dz * sh[30 + c] +
                                                 SH_C3[2] * dy * (4.0f * zz - xx - yy) * sh[33 + c] +
                                                 SH_C3[3] * dz * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                                                 SH_C3[4] * dx * (4.0f * zz - xx - yy) * sh[39 + c] +
                                                 SH_C3[5] * dz * (xx - yy) * sh[42 + c] +
                                                 SH_C3[6] * dx * (xx - 3.0f * yy) * sh[45 + c];
                                }
                            }
                        }
                        res[0] += 0.5f; res[1] += 0.5f; res[2] += 0.5f;
                        if (res[0] < 0.f) cl |= 1;
                        if (res[1] < 0.f) cl |= 2;
                        if (res[2] < 0.f) cl |= 4;
                        cr = fmaxf(res[0], 0.f); cg = fmaxf(res[1], 0.f); cbl = fmaxf(res[2], 0.f);
                    }
                    R.q0 = rq0;
                    // q1.w carries the pixel radius: a record is then self-contained (another rank can bin it, rdg_geom_from_records)
                    R.q1 = rq1;
                    // q2.w = 1 / cov2D_yy = conic_c - conic_b^2 / conic_a without that difference's cancellation: the second
                    // coefficient of the completed square the compositing kernels evaluate (rdg_stage_conic)
                    R.q2 = make_float4(cr, cg, cbl, inv_cyy);
                    R.q3 = make_float4(nx, ny, nz, 0.f);
                }
            }
        }
        rec[i] = R;
        tiles_touched[i] = my_tiles;
        rectd[i] = rd;
        clampedm[i] = cl;
        radii[i] = radius_out;
    }
    // block sum of tiles_touched -> block_sums[bid]
    uint32_t inc = rdg_wave_scan_incl(my_tiles);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[bid] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (nviews > 1) __syncthreads();       // wsum is reused by the next camera
    }
}

// Exclusive scan of block_sums[0..nblk) in place by ONE 1024-thread block; block_sums[nblk] and *num_rendered
// receive the total (D).  nblk <= 16384 at P = 4 M, i.e. runs of <= 16 entries per thread.
__global__ void __launch_bounds__(1024) rdg_scan_block_sums_kernel(uint32_t* __restrict__ block_sums, int nblk,
                                                                   int32_t* __restrict__ num_rendered,
                                                                   uint32_t* __restrict__ zero_buf, int zero_words,
                                                                   int32_t* __restrict__ nren_max) {
    // optional: clear the binning stage's per-tile counters here (this block is otherwise idle most of its life; saves
    // a memset launch and a stream boundary per frame)
    for (int i = threadIdx.x; i < zero_words; i += 1024) zero_buf[i] = 0u;
    // every thread owns a run of consecutive block sums: ONE block scan of the 1024 run totals and two barriers in all
    // (a sweep of 1024 entries at a time took three barriers per sweep, four sweeps at 1 M Gaussians)
    __shared__ uint32_t wtot[16];
    __shared__ unsigned long long wtot64[16];  // the same totals without wrap-around: D >= 2^31 must not pass for a small D
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per = (nblk + 1023) / 1024;
    const int t0 = min(nblk, (int)threadIdx.x * per), t1 = min(nblk, t0 + per);
    uint32_t mine = 0;
    unsigned long long mine64 = 0ull;
    for (int c0 = t0; c0 < t1; c0 += 8) {
        uint32_t vv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) vv[q] = c0 + q < t1 ? block_sums[c0 + q] : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) { mine += vv[q]; mine64 += vv[q]; }
    }
    const uint32_t inc = rdg_wave_scan_incl(mine);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine64 += __shfl_xor(mine64, o);
    if (lane == 63) { wtot[w] = inc; wtot64[w] = mine64; }
    __syncthreads();
    uint32_t run = inc - mine;
    for (uint32_t k = 0; k < w; ++k) run += wtot[k];
    for (int c0 = t0; c0 < t1; c0 += 8) {
        uint32_t vv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) vv[q] = c0 + q < t1 ? block_sums[c0 + q] : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (c0 + q < t1) { block_sums[c0 + q] = run; run += vv[q]; }
    }
    uint32_t carry_s = 0;
    unsigned long long total64 = 0ull;
    if (threadIdx.x == 0)
        for (int k = 0; k < 16; ++k) { carry_s += wtot[k]; total64 += wtot64[k]; }
    if (threadIdx.x == 0) {
        // every later kernel of the frame compares *num_rendered with the capacity of the instance buffers and leaves
        // when it is larger: 2^31 - 1 instances or more (garbage scales: every Gaussian on every tile) saturate, so the
        // frame is skipped and the host raises instead of indexing with a wrapped 32-bit count
        const bool fits = total64 < 0x7fffffffull;
        block_sums[nblk] = fits ? carry_s : 0x7fffffffu;
        *num_rendered = fits ? (int32_t)carry_s : 0x7fffffff;
        // sticky record (RdgRasterSettings.num_rendered_max): one thread per frame, the frames of a stream in order
        if (nren_max) { const int32_t n = fits ? (int32_t)carry_s : 0x7fffffff; if (n > *nren_max) *nren_max = n; }
    }
}

// Gaussian-sharded frame-DP: the records of this camera arrived from the ranks that own the Gaussians; rebuild what
// the binning stage reads besides them (tile counts, radii, per-block sums) from the record itself.
__global__ void __launch_bounds__(RDG_PRE_BLOCK)
rdg_geom_from_records_kernel(int P, int gx, int gy, int cull, const RdgRec* __restrict__ rec,
                             uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ block_sums,
                             int32_t* __restrict__ radii, uint4* __restrict__ rectd) {
    const int i = blockIdx.x * RDG_PRE_BLOCK + threadIdx.x;
    uint32_t my_tiles = 0;
    if (i < P) {
        const float4 q0 = rec[i].q0, q1 = rec[i].q1;
        const int radius = __float_as_int(q1.w);
        uint4 rd = make_uint4(0u, 0u, 0u, 0u);
        bool seen = false;
        // The per-tile sort compares (depth bits << 32 | id) composites as IEEE doubles (v_min_f64 / v_max_f64,
        // rdg_binning.hip rdg_cx): valid only for depth bits of a positive finite float.  This stage's own records
        // satisfy it by the near cull (vz > RDG_NEAR_CULL, which a NaN fails); records that arrived from another rank
        // are checked here -- anything else is dropped (tiles_touched = 0), never sorted.
        const uint32_t dbits = __float_as_uint(q1.z);
        if (radius > 0 && dbits > 0u && dbits < 0x7f800000u) {
            int x0, y0, x1, y1;
            rdg_rect(q0.x, q0.y, radius, gx, gy, x0, y0, x1, y1);
            seen = (x1 - x0) * (y1 - y0) > 0;
            if (seen && cull) rdg_splat_rect(q0, q1, rec[i].q2.w, gx, gy, 1, x0, y0, x1, y1);
            my_tiles = (uint32_t)((x1 - x0) * (y1 - y0));
            rd = rdg_pack_rectd(x0, y0, x1, y1, q1.z, my_tiles);
        }
        tiles_touched[i] = my_tiles;
        rectd[i] = rd;
        radii[i] = seen ? radius : 0;
    }
    __shared__ uint32_t wsum[RDG_PRE_BLOCK / RDG_WAVE];
    uint32_t inc = rdg_wave_scan_incl(my_tiles);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

int rdg_launch_geom_from_records(const RdgDev& d, void* geom_ws, int32_t* radii, int32_t* num_rendered,
                                 hipStream_t s) {
    const RdgGeomLayout L = rdg_geom_layout(d.P);
    char* g = (char*)geom_ws;
    const int nblk = (d.P + RDG_PRE_BLOCK - 1) / RDG_PRE_BLOCK;
    uint32_t* block_sums = (uint32_t*)(g + L.block_sums);
    if (d.P > 0)
        hipLaunchKernelGGL(rdg_geom_from_records_kernel, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d.P, d.gx, d.gy, d.cull,
                           (const RdgRec*)(g + L.rec), (uint32_t*)(g + L.tiles_touched), block_sums, radii,
                           (uint4*)(g + L.rectd));
    hipLaunchKernelGGL(rdg_scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, d.P > 0 ? nblk : 0,
                       num_rendered, (uint32_t*)nullptr, 0, d.nren_max);
    return rdg_check_hip(hipGetLastError(), "geom_from_records launch");
}

int rdg_launch_preprocess_fwd(const RdgDev& d, const float* means3D, const float* shs, const float* colors,
                              const float* opac, const float* scales, const float* rots, const float* cov3D,
                              const float* view, const float* proj, void* geom_ws, int32_t* radii,
                              int32_t* num_rendered, hipStream_t s, uint32_t* zero_buf, size_t zero_words) {
    const RdgGeomLayout L = rdg_geom_layout(d.P);
    char* g = (char*)geom_ws;
    const int nblk = (d.P + RDG_PRE_BLOCK - 1) / RDG_PRE_BLOCK;
    uint32_t* block_sums = (uint32_t*)(g + L.block_sums);
    if (d.P > 0) {
        hipLaunchKernelGGL(rdg_preprocess_fwd_kernel<false>, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d, view, proj, means3D,
                           shs, colors, opac, scales, rots, cov3D, (RdgRec*)(g + L.rec),
                           (uint32_t*)(g + L.tiles_touched), (uint8_t*)(g + L.clamped), block_sums, radii, 1, 0, 0,
                           (uint4*)(g + L.rectd));
    }
    hipLaunchKernelGGL(rdg_scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, block_sums, d.P > 0 ? nblk : 0,
                       num_rendered, zero_buf, (int)zero_words, d.nren_max);
    return rdg_check_hip(hipGetLastError(), "preprocess_fwd launch");
}

// All cameras of a step in one launch (sharded frame-DP owner stage): camera v owns rows [v*stride, v*stride + d.P) of
// a workspace laid out for nviews*stride rows, so that the records of all cameras form ONE contiguous send buffer.
// row0 / d.P: the launch covers rows [row0, row0 + d.P) of the slice (row0 a multiple of RDG_PRE_BLOCK); all pointers
// are those of row 0.
int rdg_launch_preprocess_fwd_views(const RdgDev& d_in, int32_t nviews, int32_t stride, int32_t row0,
                                    const float* means3D, const float* shs, const float* opac, const float* scales,
                                    const float* rots, const float* views, const float* proj, void* geom_ws,
                                    int32_t* radii, hipStream_t s) {
    const RdgGeomLayout L = rdg_geom_layout(nviews * stride);
    char* g = (char*)geom_ws;
    const int nblk = (d_in.P + RDG_PRE_BLOCK - 1) / RDG_PRE_BLOCK;
    RdgDev d = d_in;
    d.P = row0 + d_in.P;
    if (d_in.P > 0)
        hipLaunchKernelGGL(rdg_preprocess_fwd_kernel<true>, dim3(nblk), dim3(RDG_PRE_BLOCK), 0, s, d, views, proj, means3D,
                           shs, (const float*)nullptr, opac, scales, rots, (const float*)nullptr,
                           (RdgRec*)(g + L.rec), (uint32_t*)(g + L.tiles_touched), (uint8_t*)(g + L.clamped),
                           (uint32_t*)(g + L.block_sums), radii, nviews, stride, row0 / RDG_PRE_BLOCK,
                           (uint4*)(g + L.rectd));
    return rdg_check_hip(hipGetLastError(), "preprocess_fwd views launch");
}
