// rdg_render.hip -- per-tile alpha compositing, forward and backward.
//
// Geometry: one 256-thread workgroup per 16x16 tile = 4 wave64, each wave owning one 8x8 pixel quadrant
// (lane -> (lane&7, lane>>3)), so early termination and "does any pixel of my quadrant see this splat" are
// decided per wave, not per 32-wide warp.  Splat records of a tile are staged 256 at a time through LDS with one
// coalesced 64-B gather per thread; inside the loop every lane reads the same LDS address (broadcast).
//
// Wave-level occupancy masks (the wave64 replacement for per-thread rejection): while staging, the thread that
// holds splat j tests the splat's "alpha >= 1/255" ellipse against the rectangle of pixel centres of each of the four
// quadrants (exact up to a safety margin: rdg_quadrant_bits); a ballot per quadrant turns the 256 verdicts into four
// 64-bit masks per wave in LDS.  A wave then walks only the set bits of its own masks with scalar bit ops
// (s_ff1 / s_bitset0) -- on the bench scene 60 % of the (wave, splat) pairs are never touched.  The result is
// identical to visiting every splat; order inside the list is preserved.
//
// What the kernels are bound by (profiles/r02_valu_issue_cost.txt, scripts/render_ablation.sh):
// plain f32 mul / add / fma with VGPR sources issue at twice the rate of everything else (compares, selects, DPP,
// anything with a scalar source, packed f32), the walk is limited by the CU's one scalar unit, and both kernels lose
// about 10 % per lost wave of occupancy.  So: few "slow" instructions per visit, few scalar instructions per visit,
// 8 workgroups per CU.
//
// Forward, per visit: exponent as a completed square (5 instructions), alpha = opacity * exp2 (the 0.99 cap is a scalar
// branch on a per-splat bit), blend test as ONE unsigned compare (rdg_fwd_walk), alpha_eff form of the update with the
// rare "pixel stops here" case repaired in a branch, n_contrib as one move under the lane mask.
//
// Backward, per visit: one select (on G); the colour behind the splat enters only through its product with the pixel's
// dL/dpixel, so the back-to-front recurrence runs on that one scalar, started from the background term; SIX values per
// pixel leave the lane (weight t0 = G dL/dalpha, t0 dx, t0 dx^2, three colour terms) and are reduced over the 8 lanes
// of a pixel row with a TRANSPOSED quad-masked DPP reduction (12 v_add_f32_dpp); the y-moments are formed from the row
// totals (the 8 lanes share dy).  One lane per quad parks the totals with plain LDS stores in a ring PRIVATE to the
// wave (no LDS atomics -- ds_add_f32 into a table shared by the 4 waves was a third of the kernel); every 4 splats the
// wave flushes its ring with 64-B-row global float atomics: 16 consecutive lanes cover the 16 floats of a Gaussian's
// accumulator row, 4 Gaussians per instruction, which is the access shape the global float-atomic unit runs at full
// rate for (MI355X_MICROARCH.md "Global float atomics"), the eight pixel-row partials being summed by the flush.
// Atomics are therefore per (wave, splat) that a pixel actually blended.
//
// RDG_ABL_* / RDG_PARK_ALL / RDG_RING: switches of the ablation and tuning builds (scripts/render_ablation.sh); a
// product build defines none of them.
#include "rdg_common.h"

#define RDG_BATCH 256

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give every XCD one contiguous
// band of tiles -- neighbouring tiles share splats, and a band's splat records then stay in that XCD's L2.
__device__ __forceinline__ int rdg_tile_of_block(int bid, int n_tiles) {
    const int per = (n_tiles + 7) >> 3;
    return (bid & 7) * per + (bid >> 3);
}

// 4-bit mask: bit q set = the splat may reach quadrant q (q = qy*2+qx) of the tile whose top-left pixel is
// (X0, Y0).  A pixel blends the splat only where alpha >= 1/255 and power <= 0, i.e. where the quadratic form
// Q(d) = a dx^2 + 2 b dx dy + c dy^2 <= r2 = 2 ln(255 o).  The test is EXACT for the quadrant's rectangle of pixel
// centres (not the axis-aligned box of the ellipse, which lets diagonal splats through): the minimum of the convex Q
// over a box is 0 if the centre is inside, otherwise it sits on one of the four edges, where it is a clamped 1-D
// parabola.  r2 carries a margin (0.02 in the log + 1e-3 relative) that dwarfs the rounding of this evaluation, so no
// blending pixel is ever skipped; the result is identical to visiting every splat.  A conic that is not positive definite (NaN
// input) drops the splat.
__device__ __forceinline__ float rdg_edge_min(float a, float b2, float c, float inv_c, float ue, float v0, float v1) {
    // min over v in [v0, v1] of a ue^2 + 2 b ue v + c v^2   (b2 = 2 b)
    const float t = b2 * ue;
    const float vs = __builtin_amdgcn_fmed3f(-0.5f * t * inv_c, v0, v1);
    return fmaf(fmaf(c, vs, t), vs, a * ue * ue);
}
__device__ __forceinline__ uint32_t rdg_quadrant_bits(const float4 q0, const float4 q1, const float inv_cyy, float X0, float Y0) {
    // c as the compositing kernels' completed square has it (rdg_stage_conic: a (dx + beta dy)^2 + dy^2 / cov2D_yy, i.e.
    // c = b^2 / a + 1 / cov2D_yy), not the record's float32 conic_c: on a needle the two differ along the long axis by
    // eps a c / det (5e-3 at 300 : 1), which the margin below was never meant to cover
    const float a = q0.z, b = q0.w, o = q1.y;
    const float c = (b * b) / a + inv_cyy;
    const float t255 = 255.0f * o;
    if (!(t255 >= 0.99f)) return 0u;  // alpha can never reach 1/255 (margin below); written so that a NaN opacity is dropped too
    const float det = a * c - b * b;
    if (!(det > 0.0f) || !(a > 0.0f) || !(c > 0.0f)) return 0u;   // not a positive-definite conic (NaN input): never blended
    const float r2 = 2.0f * (__logf(t255) + 0.02f) * 1.001f;
    const float inv_a = 1.0f / a, inv_c = 1.0f / c, b2 = 2.0f * b;
    uint32_t bits = 0u;
#pragma unroll
    for (int qy = 0; qy < 2; ++qy) {
        const float v1 = q0.y - (Y0 + 8.0f * qy), v0 = v1 - 7.0f;       // v = py - y over the quadrant's pixel rows
#pragma unroll
        for (int qx = 0; qx < 2; ++qx) {
            const float u1 = q0.x - (X0 + 8.0f * qx), u0 = u1 - 7.0f;
            const bool inside = u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f;
            const float m = fminf(fminf(rdg_edge_min(a, b2, c, inv_c, u0, v0, v1), rdg_edge_min(a, b2, c, inv_c, u1, v0, v1)),
                                  fminf(rdg_edge_min(c, b2, a, inv_a, v0, u0, u1), rdg_edge_min(c, b2, a, inv_a, v1, u0, u1)));
            // written so that a NaN keeps the quadrant
            bits |= (uint32_t)(inside || !(m > r2)) << (qy * 2 + qx);
        }
    }
    return bits;
}

// Staged form of a splat's conic.  With (A2, B, C2) = -log2(e) * (a, b, c),
//   log2(G) = 0.5 (A2 dx^2 + C2 dy^2) + B dx dy = hA (dx + beta dy)^2 + gam dy^2,
//   hA = 0.5 A2,  beta = B / A2,  gam = 0.5 (C2 - beta B)           (the square completed once per staged splat)
// which is five instructions per pixel-splat pair instead of seven, both terms of one sign (a > 0 for every visible
// splat: the conic is the inverse of a positive-definite matrix), and G = v_exp_f32(log2 G) with no extra multiply.
// Forward and backward evaluate exactly this expression on the same staged values, so they take identical blend / skip
// decisions.
#define RDG_NEG_LOG2E (-1.4426950408889634f)
// gam: c - b^2 / a = det(conic) / a is the small remainder of its two terms on a needle-shaped footprint (det / (a c) =
// 3e-3: 2e-5 of relative error from float32 conic entries -- a SYSTEMATIC stretch of the footprint along its long axis, which
// the signed pixel sums of the backward do not average out: strict sweep 410000 / 223, dL/dmean of a 300-pixel needle 1.05e-4
// off where the float32 oracle's per-pixel noise leaves 1.1e-5).  It equals 1 / cov2D_yy exactly, and the per-Gaussian forward
// has that number without any cancellation: the record carries it (RdgRec.q2.w, `inv_cyy`).
struct RdgConicS { float hA, beta, gam; };
__device__ __forceinline__ RdgConicS rdg_stage_conic(float a, float b, float inv_cyy) {
    const float A2 = RDG_NEG_LOG2E * a, B = RDG_NEG_LOG2E * b;
    RdgConicS s;
    s.hA = 0.5f * A2;
    s.beta = (A2 < 0.0f) ? B / A2 : 0.0f;
    s.gam = (0.5f * RDG_NEG_LOG2E) * inv_cyy;
    return s;
}
__device__ __forceinline__ float rdg_log2_gauss(float hA, float beta, float gam, float dx, float dy) {
    const float w = fmaf(beta, dy, dx);
    return fmaf(hA * w, w, (gam * dy) * dy);
}
// the same, handing out the skew coordinate w = dx + beta dy (the backward takes its x-moments along it)
__device__ __forceinline__ float rdg_log2_gauss_w(float hA, float beta, float gam, float dx, float dy, float& w) {
    w = fmaf(beta, dy, dx);
    return fmaf(hA * w, w, (gam * dy) * dy);
}

// wave votes straight from the ballot (hip's __any/__all go through an int compare per lane)
__device__ __forceinline__ bool rdg_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool rdg_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

__device__ __forceinline__ unsigned long long rdg_uniform_u64(unsigned long long v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// The forward's walk over one 64-slot word of staged splats.  The compositing kernels turned out to be bound by the
// SCALAR unit (one per CU, shared by the four SIMDs: 8 extra scalar instructions per visit cost 31 % in an ablation
// build, 8 extra vector ones 6 %), so the walk is written to spend as few scalar instructions per visit as possible:
//   * s_bitset0 / s_bitset1 on the walk mask and the visit record instead of shift + xor / or,
//   * "power <= 0 and alpha >= threshold" as ONE unsigned compare: the sign bit of power is copied onto the bits of
//     alpha (both terms of the completed square are <= 0, so a blending pair has it set; alpha >= 0), and the lane's
//     threshold carries the same bit -- 0x80000000 | bits(1/255), or 0xffffffff once the pixel has stopped / lies
//     outside the image.  No mask combination on the scalar side,
//   * the alpha cap is only ever tested in words that hold a splat whose opacity exceeds it (CAPPED).
#define RDG_THR_LIVE (0x80000000u | 0x3b808081u)   // sign | bits(1.0f / 255.0f)
#define RDG_THR_DONE 0xffffffffu
template <bool NORMAL, bool CAPPED>
__device__ __forceinline__ void rdg_fwd_walk(unsigned long long mask, const unsigned long long cap, const int sbase,
                                             const uint32_t wbase, const char* sQ0, const char* sQ1, const char* sQ2,
                                             const char* sQ3, const float pixx, const float pixy,
                                             unsigned long long& seen, uint32_t& thrU, float& T,
                                             uint32_t& last_contributor, float& C0, float& C1, float& C2, float& Dp,
                                             float& N0, float& N1, float& N2) {
    while (mask) {
        const int jb = __builtin_ctzll(mask);
        asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(jb));
        // byte offset of the slot, pinned in one VGPR (the compiler otherwise re-materialises it from the scalar for
        // every array: a VALU instruction with a scalar source issues at half the rate of a plain one)
        int aj = (jb << 4) + sbase;
        asm volatile("" : "+v"(aj));
#ifdef RDG_ABL_FXSALU   // ablation build: 8 extra scalar instructions per visit
        asm volatile("s_add_u32 s90, s90, 1\n\ts_add_u32 s91, s91, 1\n\ts_add_u32 s90, s90, 1\n\ts_add_u32 s91, s91, 1\n\t"
                     "s_add_u32 s90, s90, 1\n\ts_add_u32 s91, s91, 1\n\ts_add_u32 s90, s90, 1\n\ts_add_u32 s91, s91, 1" ::: "s90", "s91", "scc");
#endif
#ifdef RDG_ABL_FXVALU   // ablation build: 8 extra (fast-class) vector instructions per visit
        { float xa = T, xb = C0;
          asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\t"
                       "v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %1, %0" : "+v"(xa), "+v"(xb)); }
#endif
        const float4 q0 = *(const float4*)(sQ0 + aj);
        const float2 q1 = *(const float2*)(sQ1 + aj);
        const float dx = q0.x - pixx, dy = q0.y - pixy;
        const float power = rdg_log2_gauss(q0.z, q0.w, q1.x, dx, dy);
        float alpha = q1.y * __builtin_amdgcn_exp2f(power);
        if (CAPPED && ((cap >> jb) & 1ull)) {
            asm volatile("; opacity above the cap" ::: "memory");   // keeps this a scalar branch (not min + select)
            alpha = fminf(RDG_ALPHA_CAP, alpha);
        }
        const uint32_t key = (__float_as_uint(power) & 0x80000000u) | __float_as_uint(alpha);
        // lanes that may blend this splat
        const unsigned long long cand = __builtin_amdgcn_ballot_w64(key >= thrU);
        if (!cand) continue;
        asm("s_bitset1_b64 %0, %1" : "+s"(seen) : "s"(jb));
        // A lane that does not blend the splat runs the same instructions with alpha_eff = 0: T (1 - 0) = T, zero
        // weight.  T >= RDG_T_STOP is an invariant of every lane (the update that would break it is the one that is
        // not applied), so "T (1 - alpha_eff) < RDG_T_STOP" alone says "this pixel stops here"; the stopping lanes
        // are repaired in the (rare) branch below.
        float aeff;
        asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(aeff) : "v"(alpha), "s"(cand));
        float test_T = T * (1.0f - aeff);
        float wgt = aeff * T;
        const float4 q2 = *(const float4*)(sQ2 + aj);
        const uint32_t idx = wbase + (uint32_t)jb;
        unsigned long long upd = cand;   // lanes that blend this splat
        const unsigned long long stopm = __builtin_amdgcn_ballot_w64(test_T < RDG_T_STOP);
        if (stopm) {
            const bool stop = test_T < RDG_T_STOP;
            upd &= ~stopm;
            wgt = stop ? 0.0f : wgt;
            test_T = stop ? T : test_T;
            thrU = stop ? RDG_THR_DONE : thrU;
            if (rdg_all(thrU == RDG_THR_DONE)) { mask = 0ull; }
        }
        T = test_T;
        // last_contributor = upd ? idx : last_contributor as ONE vector instruction (a move under the lane mask; the
        // select would need the scalar index copied into a VGPR first).  The walk runs with all lanes enabled.
        asm volatile("s_mov_b64 exec, %1\n\tv_mov_b32 %0, %2\n\ts_mov_b64 exec, -1"
                     : "+v"(last_contributor) : "s"(upd), "s"(idx));
        C0 += wgt * q2.x; C1 += wgt * q2.y; C2 += wgt * q2.z;
        Dp += wgt * q2.w;
        if (NORMAL) {
            const float4 q3 = *(const float4*)(sQ3 + aj);
            N0 += wgt * q3.x; N1 += wgt * q3.y; N2 += wgt * q3.z;
        }
    }
}


// The compositing loop of one workgroup over list positions [k_begin, k_end) of its tile (k_begin a multiple of 64).
// The whole-tile kernel calls it with (0, list length); the kernels of the split path (lists too long for one
// workgroup, below) with one segment and the transmittance the pixels arrive with.
template <bool NORMAL>
__device__ __forceinline__ void
rdg_fwd_composite(const int k_begin, const int k_end, const uint2 range, const float X0, const float Y0, const float pixx,
                  const float pixy, const uint32_t* __restrict__ point_list, const RdgRec* __restrict__ rec,
                  unsigned long long* __restrict__ hit, float4* sQ0, float4* sQ1, float4* sQ2, float4* sQ3,
                  unsigned long long (*sMask)[4], unsigned long long* sCap, uint32_t& thrU, float& T, float& C0, float& C1,
                  float& C2, float& Dp, float& N0, float& N1, float& N2, uint32_t& last_contributor) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int rounds = (k_end - k_begin + RDG_BATCH - 1) / RDG_BATCH;
    for (int r = 0; r < rounds; ++r) {
        if (__syncthreads_count(thrU == RDG_THR_DONE) == 256) break;
        const int k = k_begin + r * RDG_BATCH + tid;
        uint32_t qbits = 0;
        bool over_cap = false;
        if (k < k_end) {
            const uint32_t id = point_list[range.x + k];
            const RdgRec* p = rec + id;
            const float4 q0 = p->q0, q1 = p->q1, q2 = p->q2;
            const RdgConicS cs = rdg_stage_conic(q0.z, q0.w, q2.w);
            sQ0[tid] = make_float4(q0.x, q0.y, cs.hA, cs.beta);
            *(float2*)&sQ1[tid] = make_float2(cs.gam, q1.y);
            sQ2[tid] = make_float4(q2.x, q2.y, q2.z, q1.z);
            if (NORMAL) sQ3[tid] = p->q3;
            qbits = rdg_quadrant_bits(q0, q1, q2.w, X0, Y0);
            over_cap = q1.y > RDG_ALPHA_CAP;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long m = __ballot((qbits >> q) & 1u);
            if (lane == 0) sMask[q][wv] = m;
        }
        {
            // alpha = min(0.99, opacity * G) with G <= 1 wherever the splat blends: the min can only be active for a
            // splat whose opacity itself exceeds the cap -- a per-splat bit, tested by the scalar unit in the walk
            const unsigned long long m = __ballot(over_cap);
            if (lane == 0) sCap[wv] = m;
        }
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            if (rdg_all(thrU == RDG_THR_DONE)) break;   // once per 64 staged splats; inside the walk only after a pixel stops
            const unsigned long long mask = rdg_uniform_u64(sMask[wv][s]);
            const unsigned long long cap = rdg_uniform_u64(sCap[s]);
            unsigned long long seen = 0ull;    // splats of this 64-slot word that some pixel of the quadrant may blend
            const uint32_t wbase = (uint32_t)(k_begin + r * RDG_BATCH + s * 64 + 1);
            if (cap)
                rdg_fwd_walk<NORMAL, true>(mask, cap, s * 1024, wbase, (const char*)sQ0, (const char*)sQ1, (const char*)sQ2,
                                           (const char*)sQ3, pixx, pixy, seen, thrU, T, last_contributor, C0, C1, C2, Dp,
                                           N0, N1, N2);
            else
                rdg_fwd_walk<NORMAL, false>(mask, cap, s * 1024, wbase, (const char*)sQ0, (const char*)sQ1, (const char*)sQ2,
                                            (const char*)sQ3, pixx, pixy, seen, thrU, T, last_contributor, C0, C1, C2, Dp,
                                            N0, N1, N2);
            if (lane == 0 && seen) hit[(size_t)((k_begin >> 6) + r * 4 + s) * 4 + wv] = seen;
        }
    }
}

// split_min: tiles whose list is longer than this are composited by the split path (rdg_render_seg_* below), launched
// only when the previous frame of this shape had such a list; INT_MAX = this kernel composites every tile itself.
template <bool NORMAL>
__global__ void __launch_bounds__(256)
rdg_render_fwd_kernel(int W, int H, int gx, int n_tiles, const float* __restrict__ bg,
                      const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const RdgRec* __restrict__ rec, long long capacity, const int32_t* __restrict__ num_rendered,
                      float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                      float* __restrict__ out_depth, float* __restrict__ out_normal, float* __restrict__ out_alpha,
                      unsigned long long* __restrict__ hitbits, int split_min, uint4* __restrict__ zero_buf,
                      long long zero_n16) {
    // optional (RdgRasterSettings.zero_grad_ws): the gradient rows of the backward that follows are cleared from here.
    // This kernel is instruction-bound and stores nothing until its last lines: the fill rides along for free (a
    // launch of its own at the head of the backward: 12 us at P = 1 M).  Every workgroup, before any early exit.
    if (zero_buf) {
        const long long per = (zero_n16 + gridDim.x - 1) / gridDim.x;
        const long long z0 = per * blockIdx.x, z1 = z0 + per < zero_n16 ? z0 + per : zero_n16;
        for (long long i = z0 + threadIdx.x; i < z1; i += 256) zero_buf[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // on capacity overflow the binning stage has emptied every tile range: this kernel then renders the background
    const int tile = rdg_tile_of_block(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    // sQ1 = (conic c, opacity, -, -): only its first half is read; same 16-B stride as the others so that one address
    // register serves all four arrays.  sQ2 = (r, g, b, depth).
    __shared__ float4 sQ0[RDG_BATCH], sQ1[RDG_BATCH], sQ2[RDG_BATCH], sQ3[RDG_BATCH];
    __shared__ unsigned long long sMask[4][4];  // [consumer quadrant][staging wave]
    __shared__ unsigned long long sCap[4];      // per staging wave: splats whose opacity exceeds the alpha cap
#ifdef RDG_ABL_FPAD   // ablation build: LDS padding that lowers the occupancy
    __shared__ float sPad[RDG_ABL_FPAD];
    if (W < 0) sPad[threadIdx.x] = 1.0f;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int pxi = tx * RDG_TILE + (wv & 1) * 8 + (lane & 7);
    const int pyi = ty * RDG_TILE + (wv >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < W && pyi < H;
    float pixx = (float)pxi, pixy = (float)pyi;
    // opaque to the compiler: it otherwise re-converts the integer coordinate inside the visit loop (one VALU
    // instruction per visit to save one register)
    asm volatile("" : "+v"(pixx), "+v"(pixy));
    const float X0 = (float)(tx * RDG_TILE), Y0 = (float)(ty * RDG_TILE);
    const uint2 range = ranges[tile];
    const int todo_total = (int)(range.y - range.x);
    if (todo_total > split_min) return;          // composited by the split path
    // visit record for the backward: word w of this tile covers list slots [64 w, 64 w + 63]; the words of a tile
    // start at (range.x / 64 + tile), which cannot overlap the next tile's; zeroed by the launcher
    unsigned long long* const hit = hitbits + ((size_t)(range.x >> 6) + (size_t)tile) * 4;

    uint32_t thrU = inside ? RDG_THR_LIVE : RDG_THR_DONE;   // see rdg_fwd_walk
    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    uint32_t last_contributor = 0;
    rdg_fwd_composite<NORMAL>(0, todo_total, range, X0, Y0, pixx, pixy, point_list, rec, hit, sQ0, sQ1, sQ2, sQ3, sMask,
                              sCap, thrU, T, C0, C1, C2, Dp, N0, N1, N2, last_contributor);
    if (inside) {
        const size_t hw = (size_t)H * W;
        const size_t pid = (size_t)pyi * W + pxi;
        final_T[pid] = T;
        n_contrib[pid] = last_contributor;
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        out_depth[pid] = Dp;
        out_alpha[pid] = 1.0f - T;
        out_normal[pid] = N0; out_normal[hw + pid] = N1; out_normal[2 * hw + pid] = N2;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Split path: tile lists too long for one workgroup.  Front-to-back compositing is associative over list segments --
// a segment is a map (T, C) -> (T t_s, C + T c_s) -- so a list of n > RDG_SPLIT_MIN instances is cut into segments of
// RDG_SPLIT_SEG and composited by as many workgroups (a 200 k-instance tile walked by ONE workgroup took 15 ms):
//   pass 1  every segment's own transmittance product per pixel, t_s (alpha tests only, no early stop);
//   pass 2  every segment again, now knowing the transmittance it starts from, T_in = prod_{s' < s} t_s' -- the exact
//           walk of the whole-tile kernel from there (same blend / skip / stop decisions: the unstopped product is
//           monotone, so "the pixel stopped in an earlier segment" is exactly T_in < 1e-4), partial sums already
//           weighted by the global transmittance, the segment's last contributor and the transmittance it ends with;
//   pass 3  one workgroup per split tile adds the partial sums in list order and writes the pixels.
// Against the one-workgroup walk only the association of the transmittance product differs (last bits).
// The work lists are built on the device by the binning stage (rdg_split_build); the three launches happen only when
// the previous frame of this shape had a list above the threshold (RdgRasterSettings.list_hints bit 0, a host hint that
// decides speed, never the result: without it the whole-tile kernel walks every list itself).
// ---------------------------------------------------------------------------------------------------------
// per (segment, pixel) record, planar: RDG_SEG_F floats x 256 pixels
#define RDG_SEG_TS 0       // pass 1: transmittance product of the segment alone
#define RDG_SEG_TOUT 1     // pass 2: transmittance after the segment (of the pixel's true, possibly stopped, walk)
#define RDG_SEG_C 2        // 2..4 colour, 5 depth, 6..8 normal: partial sums weighted by the global transmittance
#define RDG_SEG_LAST 9     // last contributor inside this segment (list position + 1, uint bits), 0 = none
#define RDG_SEG_LIVE 10    // 1.0 if the pixel entered this segment alive (T_in >= 1e-4 and inside the image)

__device__ __forceinline__ float* rdg_seg_row(float* seg_pix, uint32_t seg, int field) {
    return seg_pix + ((size_t)seg * RDG_SEG_F + field) * RDG_TILE_PIX;
}

__global__ void __launch_bounds__(256)
rdg_render_seg_T_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                        const RdgRec* __restrict__ rec, const uint32_t* __restrict__ sp_header,
                        const uint4* __restrict__ sp_work, float* __restrict__ seg_pix) {
    __shared__ float4 sQ0[RDG_BATCH];
    __shared__ float2 sQ1[RDG_BATCH];
    const int tid = threadIdx.x;
    const uint32_t n_work = sp_header[0];
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const uint4 w = sp_work[wi];                     // (tile, segment, segments of the tile, first segment slot)
        const int tile = (int)w.x;
        const int tx = tile % gx, ty = tile / gx;
        const int wv = tid >> 6, lane = tid & 63;
        const int pxi = tx * RDG_TILE + (wv & 1) * 8 + (lane & 7);
        const int pyi = ty * RDG_TILE + (wv >> 1) * 8 + (lane >> 3);
        const float pixx = (float)pxi, pixy = (float)pyi;
        const uint2 range = ranges[tile];
        const int k_begin = (int)w.y * RDG_SPLIT_SEG, k_end = min(k_begin + RDG_SPLIT_SEG, (int)(range.y - range.x));
        float T = 1.0f;
        for (int k0 = k_begin; k0 < k_end; k0 += RDG_BATCH) {
            __syncthreads();
            const int k = k0 + tid;
            if (k < k_end) {
                const RdgRec* p = rec + point_list[range.x + k];
                const float4 q0 = p->q0, q1 = p->q1;
                const RdgConicS cs = rdg_stage_conic(q0.z, q0.w, p->q2.w);
                sQ0[tid] = make_float4(q0.x, q0.y, cs.hA, cs.beta);
                sQ1[tid] = make_float2(cs.gam, q1.y);
            }
            __syncthreads();
            const int n = min(RDG_BATCH, k_end - k0);
            for (int j = 0; j < n; ++j) {
                // the staged values, the expression and the test of rdg_fwd_walk: identical blend decisions
                const float4 q0 = sQ0[j];
                const float2 q1 = sQ1[j];
                const float dx = q0.x - pixx, dy = q0.y - pixy;
                const float power = rdg_log2_gauss(q0.z, q0.w, q1.x, dx, dy);
                float alpha = q1.y * __builtin_amdgcn_exp2f(power);
                if (q1.y > RDG_ALPHA_CAP) alpha = fminf(RDG_ALPHA_CAP, alpha);
                const uint32_t key = (__float_as_uint(power) & 0x80000000u) | __float_as_uint(alpha);
                if (key >= RDG_THR_LIVE) T = T * (1.0f - alpha);
            }
        }
        rdg_seg_row(seg_pix, w.w + w.y, RDG_SEG_TS)[tid] = T;
    }
}

template <bool NORMAL>
__global__ void __launch_bounds__(256)
rdg_render_seg_kernel(int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const RdgRec* __restrict__ rec, const uint32_t* __restrict__ sp_header,
                      const uint4* __restrict__ sp_work, float* __restrict__ seg_pix,
                      unsigned long long* __restrict__ hitbits) {
    __shared__ float4 sQ0[RDG_BATCH], sQ1[RDG_BATCH], sQ2[RDG_BATCH], sQ3[RDG_BATCH];
    __shared__ unsigned long long sMask[4][4];
    __shared__ unsigned long long sCap[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const uint32_t n_work = sp_header[0];
    for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
        const uint4 w = sp_work[wi];
        const int tile = (int)w.x;
        const int tx = tile % gx, ty = tile / gx;
        const int pxi = tx * RDG_TILE + (wv & 1) * 8 + (lane & 7);
        const int pyi = ty * RDG_TILE + (wv >> 1) * 8 + (lane >> 3);
        const bool inside = pxi < W && pyi < H;
        float pixx = (float)pxi, pixy = (float)pyi;
        asm volatile("" : "+v"(pixx), "+v"(pixy));
        const float X0 = (float)(tx * RDG_TILE), Y0 = (float)(ty * RDG_TILE);
        const uint2 range = ranges[tile];
        const int k_begin = (int)w.y * RDG_SPLIT_SEG, k_end = min(k_begin + RDG_SPLIT_SEG, (int)(range.y - range.x));
        unsigned long long* const hit = hitbits + ((size_t)(range.x >> 6) + (size_t)tile) * 4;
        // the transmittance this segment starts from: the product of the earlier segments' own products, in list order
        float T = 1.0f;
        for (uint32_t s = 0; s < w.y; ++s) T *= rdg_seg_row(seg_pix, w.w + s, RDG_SEG_TS)[tid];
        const bool live = inside && !(T < RDG_T_STOP);
        uint32_t thrU = live ? RDG_THR_LIVE : RDG_THR_DONE;
        float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
        uint32_t last_contributor = 0;
        __syncthreads();          // the staging arrays of the previous work item are free
        rdg_fwd_composite<NORMAL>(k_begin, k_end, range, X0, Y0, pixx, pixy, point_list, rec, hit, sQ0, sQ1, sQ2, sQ3,
                                  sMask, sCap, thrU, T, C0, C1, C2, Dp, N0, N1, N2, last_contributor);
        const uint32_t seg = w.w + w.y;
        rdg_seg_row(seg_pix, seg, RDG_SEG_TOUT)[tid] = T;
        rdg_seg_row(seg_pix, seg, RDG_SEG_C + 0)[tid] = C0;
        rdg_seg_row(seg_pix, seg, RDG_SEG_C + 1)[tid] = C1;
        rdg_seg_row(seg_pix, seg, RDG_SEG_C + 2)[tid] = C2;
        rdg_seg_row(seg_pix, seg, RDG_SEG_C + 3)[tid] = Dp;
        if (NORMAL) {
            rdg_seg_row(seg_pix, seg, RDG_SEG_C + 4)[tid] = N0;
            rdg_seg_row(seg_pix, seg, RDG_SEG_C + 5)[tid] = N1;
            rdg_seg_row(seg_pix, seg, RDG_SEG_C + 6)[tid] = N2;
        }
        rdg_seg_row(seg_pix, seg, RDG_SEG_LAST)[tid] = __uint_as_float(last_contributor);
        rdg_seg_row(seg_pix, seg, RDG_SEG_LIVE)[tid] = live ? 1.0f : 0.0f;
    }
}

template <bool NORMAL>
__global__ void __launch_bounds__(256)
rdg_render_seg_combine_kernel(int W, int H, int gx, const float* __restrict__ bg, const uint32_t* __restrict__ sp_header,
                              const uint4* __restrict__ sp_tiles, const float* __restrict__ seg_pix,
                              float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                              float* __restrict__ out_color, float* __restrict__ out_depth,
                              float* __restrict__ out_normal, float* __restrict__ out_alpha) {
    const int tid = threadIdx.x;
    const uint32_t n_tiles_split = sp_header[1];
    for (uint32_t ti = blockIdx.x; ti < n_tiles_split; ti += gridDim.x) {
        const uint4 t = sp_tiles[ti];                    // (tile, segments, first segment slot, -)
        const int tile = (int)t.x;
        const int tx = tile % gx, ty = tile / gx;
        const int wv = tid >> 6, lane = tid & 63;
        const int pxi = tx * RDG_TILE + (wv & 1) * 8 + (lane & 7);
        const int pyi = ty * RDG_TILE + (wv >> 1) * 8 + (lane >> 3);
        if (!(pxi < W && pyi < H)) continue;
        float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
        uint32_t last = 0;
        for (uint32_t s = 0; s < t.y; ++s) {
            const float* base = seg_pix + (size_t)(t.z + s) * RDG_SEG_F * RDG_TILE_PIX + tid;
            if (base[RDG_SEG_LIVE * RDG_TILE_PIX] == 0.0f) continue;     // the pixel had stopped before this segment
            T = base[RDG_SEG_TOUT * RDG_TILE_PIX];
            C0 += base[(RDG_SEG_C + 0) * RDG_TILE_PIX]; C1 += base[(RDG_SEG_C + 1) * RDG_TILE_PIX];
            C2 += base[(RDG_SEG_C + 2) * RDG_TILE_PIX]; Dp += base[(RDG_SEG_C + 3) * RDG_TILE_PIX];
            if (NORMAL) {
                N0 += base[(RDG_SEG_C + 4) * RDG_TILE_PIX]; N1 += base[(RDG_SEG_C + 5) * RDG_TILE_PIX];
                N2 += base[(RDG_SEG_C + 6) * RDG_TILE_PIX];
            }
            last = max(last, __float_as_uint(base[RDG_SEG_LAST * RDG_TILE_PIX]));
        }
        const size_t hw = (size_t)H * W;
        const size_t pid = (size_t)pyi * W + pxi;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = C0 + T * bg[0];
        out_color[hw + pid] = C1 + T * bg[1];
        out_color[2 * hw + pid] = C2 + T * bg[2];
        out_depth[pid] = Dp;
        out_alpha[pid] = 1.0f - T;
        out_normal[pid] = N0; out_normal[hw + pid] = N1; out_normal[2 * hw + pid] = N2;
    }
}

// Work lists of the split path from the tile ranges: one workgroup, every tile looked at once.  Segment slots of a tile
// are consecutive (first slot = running total), so a pixel's per-segment records can be walked in list order.
__global__ void __launch_bounds__(1024)
rdg_split_build_kernel(int n_tiles, const uint2* __restrict__ ranges, long long capacity,
                       const int32_t* __restrict__ num_rendered, uint32_t* __restrict__ sp_header,
                       uint4* __restrict__ sp_work, uint4* __restrict__ sp_tiles, uint32_t max_seg, uint32_t max_tiles) {
    __shared__ uint32_t sSeg, sTiles;
    if (threadIdx.x == 0) { sSeg = 0u; sTiles = 0u; }
    __syncthreads();
    if ((long long)num_rendered[0] <= capacity)
        for (int i = threadIdx.x; i < n_tiles; i += 1024) {
            const uint2 r = ranges[i];
            const uint32_t n = r.y - r.x;
            if (n > (uint32_t)RDG_SPLIT_MIN) {
                const uint32_t nseg = (n + RDG_SPLIT_SEG - 1) / RDG_SPLIT_SEG;
                const uint32_t base = atomicAdd(&sSeg, nseg), ti = atomicAdd(&sTiles, 1u);
                if (base + nseg <= max_seg && ti < max_tiles) {      // always true (rdg_split_layout)
                    sp_tiles[ti] = make_uint4((uint32_t)i, nseg, base, 0u);
                    for (uint32_t c = 0; c < nseg; ++c) sp_work[base + c] = make_uint4((uint32_t)i, c, nseg, base);
                }
            }
        }
    __syncthreads();
    if (threadIdx.x == 0) { sp_header[0] = min(sSeg, max_seg); sp_header[1] = min(sTiles, max_tiles); }
}

int rdg_launch_split_build(const RdgDev& d, void* bin_ws, int64_t capacity, const void* image_ws,
                           const int32_t* num_rendered, hipStream_t s) {
    const int n_tiles = d.gx * d.gy;
    const RdgBinLayout B = rdg_bin_layout(capacity, n_tiles);
    const RdgSplitLayout SL = rdg_split_layout(capacity);
    const RdgImageLayout I = rdg_image_layout(d.H, d.W);
    char* sp = (char*)bin_ws + B.split;
    hipLaunchKernelGGL(rdg_split_build_kernel, dim3(1), dim3(1024), 0, s, n_tiles,
                       (const uint2*)((const char*)image_ws + I.ranges), (long long)capacity, num_rendered,
                       (uint32_t*)(sp + SL.header), (uint4*)(sp + SL.work), (uint4*)(sp + SL.tiles), SL.max_seg,
                       SL.max_tiles);
    return rdg_check_hip(hipGetLastError(), "split_build launch");
}

int rdg_launch_render_fwd(const RdgDev& d, const float* bg, const void* geom_ws, void* bin_ws,
                          int64_t capacity, void* image_ws, const int32_t* num_rendered, float* out_color,
                          float* out_depth, float* out_normal, float* out_alpha, hipStream_t s) {
    const RdgGeomLayout G = rdg_geom_layout(d.P);
    const int n_tiles = d.gx * d.gy;
    const RdgBinLayout B = rdg_bin_layout(capacity, n_tiles);
    const RdgImageLayout I = rdg_image_layout(d.H, d.W);
    const int npass = (rdg_key_bits(n_tiles) + RDG_SORT_BITS - 1) / RDG_SORT_BITS;
    const char* b = (const char*)bin_ws;
    const uint32_t* plist = (const uint32_t*)(b + ((npass & 1) ? B.vals_b : B.vals_a));
    char* im = (char*)image_ws;
    const int nblk = ((n_tiles + 7) / 8) * 8;
    const RdgRec* rec = (const RdgRec*)((const char*)geom_ws + G.rec);
    const uint2* ranges = (const uint2*)(im + I.ranges);
    float* final_T = (float*)(im + I.final_T);
    uint32_t* n_contrib = (uint32_t*)(im + I.n_contrib);
    // zeroed by the binning stage (rdg_launch_bin), which always runs before this launch
    unsigned long long* hitbits = (unsigned long long*)((char*)bin_ws + B.hit);
    const int split_min = (d.list_hints & 1) ? RDG_SPLIT_MIN : 0x7fffffff;
#define RDG_FWD_LAUNCH(NORMAL)                                                                                      \
    hipLaunchKernelGGL(rdg_render_fwd_kernel<NORMAL>, dim3(nblk), dim3(256), 0, s, d.W, d.H, d.gx, n_tiles, bg,       \
                       ranges, plist, rec, (long long)capacity, num_rendered, final_T, n_contrib,                    \
                       out_color, out_depth, out_normal, out_alpha, hitbits, split_min, (uint4*)d.zero_grad_ws,     \
                       zero_n16)
    const long long zero_n16 = (long long)(rdg_align_up((size_t)(d.P > 0 ? d.P : 1) * RDG_GROW * 4, 256) / 16);
    if (d.render_normal) RDG_FWD_LAUNCH(true); else RDG_FWD_LAUNCH(false);
#undef RDG_FWD_LAUNCH
    if ((d.list_hints & 1)) {
        // long lists: work lists from the ranges, then per-segment transmittance, per-segment composite, ordered combine
        int rc = rdg_launch_split_build(d, bin_ws, capacity, image_ws, num_rendered, s);
        if (rc) return rc;
        const RdgSplitLayout SL = rdg_split_layout(capacity);
        char* sp = (char*)bin_ws + B.split;
        const uint32_t* sp_header = (const uint32_t*)(sp + SL.header);
        const uint4* sp_work = (const uint4*)(sp + SL.work);
        const uint4* sp_tiles = (const uint4*)(sp + SL.tiles);
        float* seg_pix = (float*)(sp + SL.seg_pix);
        const unsigned gseg = SL.max_seg < 2048u ? SL.max_seg : 2048u, gtile = SL.max_tiles < 1024u ? SL.max_tiles : 1024u;
        hipLaunchKernelGGL(rdg_render_seg_T_kernel, dim3(gseg), dim3(256), 0, s, d.W, d.H, d.gx, ranges, plist, rec,
                           sp_header, sp_work, seg_pix);
        if (d.render_normal) {
            hipLaunchKernelGGL(rdg_render_seg_kernel<true>, dim3(gseg), dim3(256), 0, s, d.W, d.H, d.gx, ranges, plist, rec,
                               sp_header, sp_work, seg_pix, hitbits);
            hipLaunchKernelGGL(rdg_render_seg_combine_kernel<true>, dim3(gtile), dim3(256), 0, s, d.W, d.H, d.gx, bg,
                               sp_header, sp_tiles, (const float*)seg_pix, final_T, n_contrib, out_color, out_depth,
                               out_normal, out_alpha);
        } else {
            hipLaunchKernelGGL(rdg_render_seg_kernel<false>, dim3(gseg), dim3(256), 0, s, d.W, d.H, d.gx, ranges, plist, rec,
                               sp_header, sp_work, seg_pix, hitbits);
            hipLaunchKernelGGL(rdg_render_seg_combine_kernel<false>, dim3(gtile), dim3(256), 0, s, d.W, d.H, d.gx, bg,
                               sp_header, sp_tiles, (const float*)seg_pix, final_T, n_contrib, out_color, out_depth,
                               out_normal, out_alpha);
        }
    }
    return rdg_check_hip(hipGetLastError(), "render_fwd launch");
}

// 4 entries = one flush instruction per component row, and 6 KB of ring per workgroup: 8 workgroups per CU instead of 6
// (the kernel is latency-bound: 0.457 -> 0.422 ms; 16 entries: 0.553 ms)
#ifndef RDG_RING
#define RDG_RING 4
#endif
// A ring entry holds, for each of the wave's 16 quads (quad = lane >> 2; two quads = one 8-pixel row of the quadrant),
// six floats: see the reduction in the kernel.  Stride 6 keeps the parking stores 8-byte aligned and the 16 writer
// lanes on 16 different banks.
#define RDG_RING_Q 6
// Flush a wave's ring: 16 consecutive lanes = the 64-B accumulator row of one Gaussian, 4 entries per instruction,
// which is the access shape the global float-atomic unit runs at full rate for.  Lane c sums the eight pixel-row
// partials of component c (foff = where that component sits inside a quad pair).
// `scale` = this lane's constant factor for component (lane & 15), see the derivative block of the kernel.
// DET (deterministic mode, rdg_composite_backward_det): the ring entry names the LIST POSITION of the instance instead of
// the Gaussian, and the wave's totals are STORED to its own 64-B quarter of that position's row in a D-long scratch
// buffer (grow = scratch + 16 * wave) -- no atomics; rdg_det_reduce_kernel then adds them up per Gaussian in a fixed order.
template <bool HAS_DEPTH, bool DET>
__device__ __forceinline__ void rdg_ring_flush(float (*ring)[16][RDG_RING_Q], int n, int lane, float scale, int foff,
                                               float* __restrict__ grow) {
    // LDS operations of one wave execute in program order; the fences only pin the compiler's ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int c = lane & 15;
    for (int e0 = 0; e0 < n; e0 += 4) {
        const int e = e0 + (lane >> 4);
        if (e < n && c < (HAS_DEPTH ? 10 : 9)) {
            const float* r = &ring[e][0][0] + foff;
            constexpr int G = 2 * RDG_RING_Q;   // floats per pixel row (quad pair)
            const float v = (((r[0] + r[G]) + (r[2 * G] + r[3 * G])) + ((r[4 * G] + r[5 * G]) + (r[6 * G] + r[7 * G]))) * scale;
#ifdef RDG_ABL_NOATOMIC   // ablation build (scripts/render_ablation.sh): results are wrong, only the timing is read
            if (v == 123.456f) atomicAdd(grow + (size_t)__float_as_uint(ring[e][1][4]) * RDG_GROW + c, v);
#else
            if (DET) grow[(size_t)__float_as_uint(ring[e][1][4]) * (4 * RDG_GROW) + c] = v;
            else if (v != 0.0f) atomicAdd(grow + (size_t)__float_as_uint(ring[e][1][4]) * RDG_GROW + c, v);
#endif
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------
// The backward's walk over one 64-slot word of staged splats (the scalar-instruction budget of rdg_fwd_walk applies).
// jthr: per lane, the slot bit index above which the list position lies below the pixel's last contributor.
// ALLBELOW: every list position of this word lies below the last contributor of every pixel of the quadrant (the usual case:
// pixels that never stopped early), so the per-lane position test drops out of the visit.
// HAS_NORMAL: an upstream gradient for the normal image.  The per-Gaussian normals are constants of the graph (the oracle
// detaches them), so the normal channels act like three more colour channels in dL/dalpha and nothing is accumulated for
// the normals themselves.
template <bool HAS_DEPTH, bool CAPPED, bool DET, bool ALLBELOW, bool HAS_NORMAL>
__device__ __forceinline__ void rdg_bwd_walk(unsigned long long mask, const unsigned long long cap, const int sbase,
                                             const char* sQ0, const char* sQ1, const char* sQ2, const char* sQ3,
                                             const float pixx, const float pixy, const int jthr, const float dLp0,
                                             const float dLp1, const float dLp2, const float dLd, const float dLn0,
                                             const float dLn1, const float dLn2, const int lane,
                                             float (*ring)[16][RDG_RING_Q], const float flush_scale, const int flush_off,
                                             float* __restrict__ grow, float& T, float& behind, int& ring_n) {
    while (mask) {
        const int jb = __builtin_ctzll(mask);
        asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(jb));
        // byte offset of slot j, pinned in one VGPR (the compiler otherwise re-materialises it from the scalar
        // for every array: a VALU instruction with a scalar source issues at half the rate of a plain one)
        int aj = (jb << 4) + sbase;
        asm volatile("" : "+v"(aj));
        const float4 q0 = *(const float4*)(sQ0 + aj);
        const float4 q1 = *(const float4*)(sQ1 + aj);
        const float dx = q0.x - pixx, dy = q0.y - pixy;
        float wsk;                                                  // dx + beta dy
        const float power = rdg_log2_gauss_w(q0.z, q0.w, q1.x, dx, dy, wsk);
        const float G = __builtin_amdgcn_exp2f(power);
        float alpha = q1.y * G;
        const bool capped = CAPPED && ((cap >> jb) & 1ull);
        if (capped) {
            asm volatile("; opacity above the cap" ::: "memory");   // keeps this a scalar branch (not min + select)
            alpha = fminf(RDG_ALPHA_CAP, alpha);
        }
        // "power <= 0 and alpha >= 1/255" as one unsigned compare (see rdg_fwd_walk); "list position below the pixel's
        // last contributor" as a compare of the slot's bit index with a per-lane bound formed once per word
        const uint32_t key = (__float_as_uint(power) & 0x80000000u) | __float_as_uint(alpha);
        const bool hit = ALLBELOW ? key >= RDG_THR_LIVE : ((jb > jthr) && key >= RDG_THR_LIVE);
        if (!rdg_any(hit)) continue;
#ifdef RDG_ABL_PREONLY   // ablation build: the walk + the blend test only
        if (G != 123.456f) continue;
#endif
        const float4 q2 = *(const float4*)(sQ2 + aj);
        // Branch-free per-pixel derivatives.  A lane that does not blend this splat runs the same instructions
        // with alpha_eff = 0, which leaves every piece of its state unchanged (T / (1 - 0) = T, behind += 0) and
        // zeroes its contributions, so no per-variable selects are needed.  The colour accumulated BEHIND the
        // current splat (back-to-front recurrence B <- alpha c + (1 - alpha) B, upstream's accum_rec evaluated
        // eagerly) only ever enters through its product with the pixel's dL/dpixel, a per-pixel constant: so the
        // recurrence runs on that ONE scalar, behind = B . dL/dpixel, with s = c . dL/dpixel per splat.
        const float Gm = hit ? G : 0.0f;                      // the ONE select of the visit
        float aeff = q1.y * Gm;                               // = alpha on the lanes that blend, 0 elsewhere
        if (capped) {
            asm volatile("; opacity above the cap" ::: "memory");
            aeff = fminf(RDG_ALPHA_CAP, aeff);
        }
        const float inv1ma = __builtin_amdgcn_rcpf(1.0f - aeff);
        T = T * inv1ma;
        const float dch = aeff * T;
        float s_ = q2.x * dLp0 + q2.y * dLp1 + q2.z * dLp2;
        if (HAS_DEPTH) s_ += q1.z * dLd;
        if (HAS_NORMAL) {
            const float4 q3 = *(const float4*)(sQ3 + aj);
            s_ += q3.x * dLn0 + q3.y * dLn1 + q3.z * dLn2;
        }
        const float e_ = s_ - behind;
        behind = fmaf(aeff, e_, behind);
        const float dL_dalpha = e_ * T;
        // What leaves the lane: the weight t0 = G dL/dalpha (= dL/dopacity; G dL/dG = opacity * t0, the opacity
        // being a per-splat constant applied by the per-Gaussian backward), its x-moments t1 = t0 w and
        // t2 = t0 w^2 ALONG THE CONIC'S SKEW AXIS, w = dx + beta dy (beta = conic_b / conic_a: the coordinate the
        // exponent was evaluated in, so it costs nothing here), and the colour terms.  Moments about (w, dy) instead of
        // (dx, dy): for a needle-shaped footprint sum(t0 dx) and beta sum(t0 dy) are large and cancel in
        // dL/dmean = -conic . moments; accumulated separately over pixels, waves and tiles, their rounding would come out
        // amplified by the footprint's condition number (1.5e-4 of the column's largest entry of dL/dmean on a
        // det / (a c) = 0.04 splat, sweep case 90000 / 159), where sum(t0 w) has nothing to cancel.  The per-Gaussian
        // backward turns the (w, dy) moments back into the (dx, dy) ones it needs (rdg_preprocess_bwd.hip).  The y-moments are NOT formed per pixel: the eight lanes of a
        // pixel row share dy, so sum(t0 dy) = dy sum(t0), sum(t0 dx dy) = dy sum(t1), sum(t0 dy^2) = dy^2 sum(t0)
        // are formed from the row totals (two multiplies per visit instead of three, and six values to reduce
        // instead of nine).  The constant factors of the conic derivatives (-0.5, -1, -0.5) are applied once per
        // flushed row total (rdg_ring_flush).
        const float t0 = Gm * dL_dalpha;
        const float t1 = t0 * wsk;
        const float t2 = t1 * wsk;
        const float c0 = dch * dLp0, c1 = dch * dLp1, c2 = dch * dLp2, cd = HAS_DEPTH ? dch * dLd : 0.0f;
        // Transposed reduction over the 8 lanes of a pixel row: DPP write masks work on quads (bank_mask), so the
        // step across the row's two quads comes first and folds two values into one register (quad A = lanes
        // 0-3 of the row keeps the pair sums of the first value, quad B of the second: one DPP add per bank
        // mask into the same register, which the compiler cannot express), then two plain quad butterflies.
        // 12 DPP adds (15 with depth) instead of 20 (21) for nine row totals over 16 lanes.  DPP needs two wait
        // states after the VALU write of the register it reads: the order below keeps at least two instructions
        // between every producer and its DPP consumer; s_nop 1 covers the inputs.
        // Result, in every lane of a quad:   quad A: r0 = sum t0, r1 = sum t2, r2 = sum c1
        //                                    quad B: r0 = sum t1, r1 = sum c0, r2 = sum c2     (r3 = sum cd)
        float r0, r1, r2, r3 = 0.0f;
#ifdef RDG_ABL_NODPP
        r0 = t0 + t1; r1 = t2 + c0; r2 = c1 + c2; r3 = cd;
        if (true) {
        } else if (HAS_DEPTH) {
#else
        if (HAS_DEPTH) {
#endif
            asm volatile(
                "s_nop 1\n\t"
                "v_add_f32_dpp %[r0], %[t0], %[t0] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %[r0], %[t1], %[t1] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %[r1], %[t2], %[t2] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %[r1], %[c0], %[c0] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %[r2], %[c1], %[c1] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %[r2], %[c2], %[c2] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %[r3], %[cd], %[cd] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r0], %[r0], %[r0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r1], %[r1], %[r1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r2], %[r2], %[r2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r3], %[r3], %[r3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r0], %[r0], %[r0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r1], %[r1], %[r1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r2], %[r2], %[r2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r3], %[r3], %[r3] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3)
                : [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [cd] "v"(cd));
        } else {
            asm volatile(
                "s_nop 1\n\t"
                "v_add_f32_dpp %[r0], %[t0], %[t0] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %[r0], %[t1], %[t1] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %[r1], %[t2], %[t2] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %[r1], %[c0], %[c0] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %[r2], %[c1], %[c1] row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                "v_add_f32_dpp %[r2], %[c2], %[c2] row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                "v_add_f32_dpp %[r0], %[r0], %[r0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r1], %[r1], %[r1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r2], %[r2], %[r2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r0], %[r0], %[r0] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r1], %[r1], %[r1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                "v_add_f32_dpp %[r2], %[r2], %[r2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2)
                : [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2));
        }
        // y-moments of the row from its totals (quad A: dy sum t0, dy^2 sum t0; quad B: dy sum t1)
        const float m1 = r0 * dy;
        const float m2 = m1 * dy;
        {
            // One lane per quad parks the quad's totals in this wave's private ring with PLAIN LDS stores (LDS
            // float atomics into a table shared by the 4 waves cost a third of the kernel); the 8 pixel-row
            // partials are added when the ring is flushed.
#if defined(RDG_PARK_ALL)
            {   // tuning build: every lane of a quad stores the quad's (identical) totals to the same address
                float* gr = &ring[ring_n][lane >> 2][0];
                gr[0] = r0; gr[1] = r1; gr[2] = r2; gr[3] = m1; gr[4] = ((lane >> 2) == 1) ? q2.w : m2;
                if (HAS_DEPTH) gr[5] = r3;
            }
#else
#ifdef RDG_ABL_NOPARK
            if (lane == 63 && m2 == 123.456f) {
#else
            if ((lane & 3) == 0) {
#endif
                float* gr = &ring[ring_n][lane >> 2][0];
                gr[0] = r0; gr[1] = r1; gr[2] = r2; gr[3] = m1; gr[4] = m2;
                if (HAS_DEPTH) gr[5] = r3;
                // the Gaussian's row index rides in a slot no component uses (quad B has no dy^2 moment); LDS
                // stores of a wave land in program order, so this one overrides m2 of lane 4
                if (lane == 4) gr[4] = q2.w;
            }
#endif
            if (++ring_n == RDG_RING) {
                rdg_ring_flush<HAS_DEPTH, DET>(ring, ring_n, lane, flush_scale, flush_off, grow);
                ring_n = 0;
            }
        }
    }
}

// The back-to-front walk of one workgroup over list positions [k_lo, k_top) of its tile (k_lo a multiple of 64): the
// whole-tile kernel calls it with (0, largest last contributor of the tile), the split path with one segment.
// m0..m3: per quadrant, the largest last contributor of its pixels.  T / behind: the pixel's transmittance after, and the
// (normalised) colour-gradient product behind, list position k_top - 1.
template <bool HAS_DEPTH, bool DET, bool HAS_NORMAL>
__device__ __forceinline__ void
rdg_bwd_composite(const int k_lo, const int k_top, const uint2 range, const int m0, const int m1, const int m2, const int m3,
                  const int last_contributor, const float pixx, const float pixy, const float dLp0, const float dLp1,
                  const float dLp2, const float dLd, const float dLn0, const float dLn1, const float dLn2,
                  const uint32_t* __restrict__ point_list,
                  const RdgRec* __restrict__ rec, const unsigned long long* __restrict__ hit, float4* sQ0, float4* sQ1,
                  float4* sQ2, float4* sQ3, unsigned long long (*sMask)[4], unsigned long long* sCap, float (*ring)[16][RDG_RING_Q],
                  const float flush_scale, const int flush_off, float* __restrict__ gdst, float& T, float& behind,
                  int& ring_n, const uint32_t* __restrict__ det_off, const uint4* __restrict__ rectd, const int tx, const int ty) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int rounds = (k_top - k_lo + RDG_BATCH - 1) / RDG_BATCH;
    for (int r = 0; r < rounds; ++r) {
        const int kbase = k_top - 1 - r * RDG_BATCH;  // list position of slot 0 of this batch
        uint32_t qbits = 0;
        bool over_cap = false;
        {
            const int k = kbase - tid;
            if (k >= k_lo) {
                // a quadrant whose pixels all stopped before this list position never needs the splat ...
                qbits = (uint32_t)(k < m0) | ((uint32_t)(k < m1) << 1) | ((uint32_t)(k < m2) << 2) |
                        ((uint32_t)(k < m3) << 3);
                // ... and neither does one in which the forward found no pixel that could blend it (same staged
                // values, same expression: every backward hit was a forward candidate)
                const unsigned long long* hw = hit + (size_t)(k >> 6) * 4;
                const int hb = k & 63;
                qbits &= (uint32_t)((hw[0] >> hb) & 1ull) | ((uint32_t)((hw[1] >> hb) & 1ull) << 1) |
                         ((uint32_t)((hw[2] >> hb) & 1ull) << 2) | ((uint32_t)((hw[3] >> hb) & 1ull) << 3);
                if (qbits) {   // nobody will look at the other slots: skip their record gathers
                    const uint32_t id = point_list[range.x + k];
                    const RdgRec* p = rec + id;
                    const float4 q0 = p->q0, q1 = p->q1, q2 = p->q2;
                    const RdgConicS cs = rdg_stage_conic(q0.z, q0.w, q2.w);
                    sQ0[tid] = make_float4(q0.x, q0.y, cs.hA, cs.beta);
                    sQ1[tid] = make_float4(cs.gam, q1.y, q1.z, 0.0f);
                    uint32_t row = id;
                    if (DET) {
                        // deterministic mode: the row of this (tile, Gaussian) instance in Gaussian-major order -- the
                        // Gaussian's first instance (det_off, exclusive scan of tiles_touched) + the tile's ordinal inside
                        // the Gaussian's rectangle (the rectangle the binning stage expanded, `rectd`): the
                        // reduction then reads a Gaussian's rows one after the other, no search
                        const uint4 rd = rectd[id];
                        const int x0 = (int)(rd.x & 0xffffu), y0 = (int)(rd.x >> 16), x1 = (int)(rd.y & 0xffffu);
                        row = det_off[id] + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
                    }
                    sQ2[tid] = make_float4(q2.x, q2.y, q2.z, __uint_as_float(row));
                    if (HAS_NORMAL) sQ3[tid] = p->q3;
                    over_cap = q1.y > RDG_ALPHA_CAP;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long m = __ballot((qbits >> q) & 1u);
            if (lane == 0) sMask[q][wv] = m;
        }
        {
            // alpha = min(0.99, opacity * G) with G <= 1 wherever the splat blends: the min is only ever active for a
            // splat whose opacity itself exceeds the cap -- a per-splat bit, tested by the scalar unit in the walk
            const unsigned long long m = __ballot(over_cap);
            if (lane == 0) sCap[wv] = m;
        }
        __syncthreads();
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            unsigned long long mask = rdg_uniform_u64(sMask[wv][s]);
            const unsigned long long cap = rdg_uniform_u64(sCap[s]);
#ifdef RDG_ABL_STAGEONLY   // ablation build: staging and barriers only
            if (T != 123.456f) mask = 0ull;
#endif
            // list position of slot bit jb of this word: kbase - 64 s - jb; below last_contributor <=> jb > jthr
            const int jthr = kbase - s * 64 - last_contributor;
#define RDG_WALK(CAPPED, ALLB)                                                                                          \
            rdg_bwd_walk<HAS_DEPTH, CAPPED, DET, ALLB, HAS_NORMAL>(mask, cap, s * 1024, (const char*)sQ0,                \
                                                       (const char*)sQ1, (const char*)sQ2, (const char*)sQ3, pixx, pixy,  \
                                                       jthr, dLp0, dLp1, dLp2, dLd, dLn0, dLn1, dLn2, lane,               \
                                                       ring, flush_scale, flush_off, gdst, T, behind, ring_n)
            if (rdg_all(jthr < 0)) { if (cap) RDG_WALK(true, true); else RDG_WALK(false, true); }
            else { if (cap) RDG_WALK(true, false); else RDG_WALK(false, false); }
#undef RDG_WALK
        }
        __syncthreads();   // every wave is done with this round's staged records
    }
}

// HAS_DEPTH = false: no upstream gradient for the depth image (photometric-only losses) -- the depth channel drops out
// of the per-pair arithmetic and of the reduction.
// SEG: the split path (see the forward): one work item = one segment of a long list; the pixel's transmittance after
// the segment is the one the forward stored, and what lies behind it is rebuilt from the later segments' partial sums.
template <bool HAS_DEPTH, bool DET, bool SEG, bool HAS_NORMAL>
__global__ void __launch_bounds__(256)
rdg_render_bwd_kernel(int W, int H, int gx, int n_tiles, const float* __restrict__ bg,
                      const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                      const RdgRec* __restrict__ rec, const float* __restrict__ final_T,
                      const uint32_t* __restrict__ n_contrib, const float* __restrict__ g_color,
                      const float* __restrict__ g_depth, const float* __restrict__ g_alpha,
                      const float* __restrict__ g_normal, float* __restrict__ grow, const unsigned long long* __restrict__ hitbits, int split_min,
                      const uint32_t* __restrict__ sp_header, const uint4* __restrict__ sp_work,
                      const float* __restrict__ seg_pix, const uint32_t* __restrict__ det_off,
                      const uint4* __restrict__ rectd) {
    __shared__ float4 sQ0[RDG_BATCH], sQ1[RDG_BATCH], sQ2[RDG_BATCH];   // sQ2.w = the Gaussian's row index (bits)
    __shared__ float4 sQ3[HAS_NORMAL ? RDG_BATCH : 1];                  // normals, only with a normal gradient
    __shared__ float sRing[4][RDG_RING][16][RDG_RING_Q];   // per wave: [entry][quad][slot] partial sums
    __shared__ unsigned long long sMask[4][4];
    __shared__ unsigned long long sCap[4];   // per staging wave: splats whose opacity exceeds the alpha cap
    __shared__ int sMax[4];
#ifdef RDG_ABL_BWD_LDSPAD   // ablation build: extra LDS per workgroup = fewer workgroups per CU (what a matrix-core row
                            // reduction with its per-wave transpose buffers would leave: profiles/r04_render_bwd_mfma_bound.txt)
    __shared__ float sPad[RDG_ABL_BWD_LDSPAD / 4];
    if (W == -12345) sPad[threadIdx.x] = 1.0f;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const size_t hw = (size_t)H * W;
    const int fc = lane & 15;
    // components 0, 1 are the raw first moments sum(G dL/dG dx), sum(G dL/dG dy): the per-Gaussian backward turns them
    // into dL/dmean2D with the conic it has anyway (two multiplies and two fused multiply-adds per pair less here)
    const float flush_scale = (fc == 2 || fc == 4) ? -0.5f : fc == 3 ? -1.0f : 1.0f;
    // where component fc sits in a pixel row's two quads (A = slots 0..5, B = slots 6..11): see the reduction below
    const int flush_off = (int)((0x5827049136ull >> (4 * fc)) & 15ull);
    float* __restrict__ const gdst = DET ? grow + wv * RDG_GROW : grow;   // DET: this wave's quarter of an instance row
    const uint32_t n_items = SEG ? sp_header[0] : 1u;
    for (uint32_t wi = SEG ? blockIdx.x : 0u; wi < n_items; wi += SEG ? gridDim.x : 1u) {
        uint4 item = make_uint4(0u, 0u, 0u, 0u);
        int tile;
        if (SEG) { item = sp_work[wi]; tile = (int)item.x; }
        else { tile = rdg_tile_of_block(blockIdx.x, n_tiles); if (tile >= n_tiles) return; }
        const int tx = tile % gx, ty = tile / gx;
        const int pxi = tx * RDG_TILE + (wv & 1) * 8 + (lane & 7);
        const int pyi = ty * RDG_TILE + (wv >> 1) * 8 + (lane >> 3);
        const bool inside = pxi < W && pyi < H;
        const float pixx = (float)pxi, pixy = (float)pyi;
        const uint2 range = ranges[tile];
        if (!SEG && (int)(range.y - range.x) > split_min) return;      // walked by the split path
        const size_t pid = (size_t)pyi * W + pxi;
        const unsigned long long* const hit = hitbits + ((size_t)(range.x >> 6) + (size_t)tile) * 4;
        const float T_final = inside ? final_T[pid] : 0.0f;
        const int last_contributor = inside ? (int)n_contrib[pid] : 0;
        float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLd = 0.f, dLa = 0.f, dLn0 = 0.f, dLn1 = 0.f, dLn2 = 0.f;
        if (inside) {
            if (g_color) { dLp0 = g_color[pid]; dLp1 = g_color[hw + pid]; dLp2 = g_color[2 * hw + pid]; }
            if (HAS_DEPTH) dLd = g_depth[pid];
            if (g_alpha) dLa = g_alpha[pid];
            if (HAS_NORMAL) { dLn0 = g_normal[pid]; dLn1 = g_normal[hw + pid]; dLn2 = g_normal[2 * hw + pid]; }
        }
        // What lies BEHIND the last splat of a pixel, in units of "colour . dL/dpixel": the background, and the alpha
        // output (alpha_out = 1 - T_final) as a colour of -dL/dalpha_out.  Starting the behind-value recurrence from it
        // makes dL/dalpha_k = T_k (s_k - behind_k) exact with no separate T_final term.
        const float bgdot = bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2;
        float T = T_final;
        float behind = bgdot - dLa;
        int k_lo = 0, k_hi = 0x7fffffff;
        if (SEG) {
            k_lo = (int)item.y * RDG_SPLIT_SEG;
            k_hi = min(k_lo + RDG_SPLIT_SEG, (int)(range.y - range.x));
            // The pixel takes part in this segment only if its last contributor lies beyond the segment's first slot.
            // Then: transmittance after the segment = what the forward stored; behind it = the later segments' partial
            // sums (already weighted by the global transmittance) + the background term, brought to this point's scale.
            if (last_contributor > k_lo) {
                float far = T_final * (bgdot - dLa);
                for (uint32_t s2 = item.y + 1; s2 < item.z; ++s2) {
                    const float* b2 = seg_pix + (size_t)(item.w + s2) * RDG_SEG_F * RDG_TILE_PIX + tid;
                    far += b2[(RDG_SEG_C + 0) * RDG_TILE_PIX] * dLp0 + b2[(RDG_SEG_C + 1) * RDG_TILE_PIX] * dLp1 +
                           b2[(RDG_SEG_C + 2) * RDG_TILE_PIX] * dLp2;
                    if (HAS_DEPTH) far += b2[(RDG_SEG_C + 3) * RDG_TILE_PIX] * dLd;
                    if (HAS_NORMAL)
                        far += b2[(RDG_SEG_C + 4) * RDG_TILE_PIX] * dLn0 + b2[(RDG_SEG_C + 5) * RDG_TILE_PIX] * dLn1 +
                               b2[(RDG_SEG_C + 6) * RDG_TILE_PIX] * dLn2;
                }
                T = seg_pix[((size_t)(item.w + item.y) * RDG_SEG_F + RDG_SEG_TOUT) * RDG_TILE_PIX + tid];
                behind = far / T;
            } else {
                T = 1.0f; behind = 0.0f;      // never blends here: finite values keep the branch-free arithmetic clean
            }
        }
        // wave / block maxima of last_contributor: splats at list positions >= max are skipped wholesale
        int wmax = last_contributor;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
        __syncthreads();                       // SEG: the previous work item is done with sMax
        if (lane == 0) sMax[wv] = wmax;
        __syncthreads();
        const int m0 = sMax[0], m1 = sMax[1], m2 = sMax[2], m3 = sMax[3];
        const int kmax = max(max(m0, m1), max(m2, m3));
        const int k_top = min(kmax, k_hi);
        int ring_n = 0;   // wave-uniform fill level of this wave's ring
        rdg_bwd_composite<HAS_DEPTH, DET, HAS_NORMAL>(k_lo, k_top, range, m0, m1, m2, m3, last_contributor, pixx, pixy, dLp0,
                                                      dLp1, dLp2, dLd, dLn0, dLn1, dLn2, point_list, rec, hit, sQ0, sQ1, sQ2,
                                                      sQ3, sMask, sCap, sRing[wv], flush_scale, flush_off, gdst, T, behind,
                                                      ring_n, det_off, rectd, tx, ty);
        rdg_ring_flush<HAS_DEPTH, DET>(sRing[wv], ring_n, lane, flush_scale, flush_off, gdst);
    }
}

// Deterministic mode, first step: det_off[g] = index of Gaussian g's first (tile, Gaussian) instance in Gaussian-major
// order = exclusive scan of tiles_touched (the per-block part is the binning stage's block_sums).
__global__ void __launch_bounds__(RDG_PRE_BLOCK)
rdg_det_offsets_kernel(int P, const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ block_sums,
                       uint32_t* __restrict__ det_off) {
    __shared__ uint32_t wsum[RDG_PRE_BLOCK / 64];
    const int i = blockIdx.x * RDG_PRE_BLOCK + threadIdx.x;
    const uint32_t mine = i < P ? tiles_touched[i] : 0u;
    const uint32_t inc = rdg_wave_scan_incl(mine);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = block_sums[blockIdx.x];
    for (uint32_t k = 0; k < w; ++k) base += wsum[k];
    if (i < P) det_off[i] = base + inc - mine;
}

// Deterministic mode, last step: one 16-lane group per Gaussian, lane c = component c of the 64-B gradient row.  The
// Gaussian's instances sit in rows det_off[g] .. + tiles_touched[g] in the row-major order of its tile rectangle (the
// staging of the compositing backward addressed them that way); the group adds the four per-wave partial rows of each
// in a fixed order.  Every float addition of the accumulation therefore happens in an order fixed by the geometry alone
// -- and the rows are read one after the other (the first form of this kernel found each instance by a binary search in
// its tile's sorted list: 9 dependent loads per instance, 1.7 ms at D = 4 M).
__global__ void __launch_bounds__(256)
rdg_det_reduce_kernel(int P, const uint32_t* __restrict__ tiles_touched, const uint32_t* __restrict__ det_off,
                      long long n_instances, const float* __restrict__ det, float* __restrict__ grow) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int g = (int)(t >> 4), c = (int)(t & 15);
    if (g >= P) return;
    // Compensated (two-sum) accumulation: a footprint hundreds of pixels long spans dozens of tiles, and its first moments
    // are signed sums that cancel to a fraction of their terms (strict sweep 410000 / 223: dL/dmean 1.05e-4 off).  This kernel
    // waits for gathers, not for its adds, so the exact rounding error of every addition is carried along for free -- the
    // hot kernel cannot afford the registers; here the order is fixed and the extra adds hide behind the loads.
    float acc = 0.0f, comp = 0.0f;
    const uint32_t n = tiles_touched[g], first = det_off[g];
    // (an overflowed frame has empty tile lists and meaningless offsets: nothing beyond the workspace is read)
    if ((long long)first + n <= n_instances) {
        for (uint32_t j = 0; j < n; ++j) {
            const float* row = det + (size_t)(first + j) * (4 * RDG_GROW) + c;
            const float a0 = row[0], a1 = row[RDG_GROW], a2 = row[2 * RDG_GROW], a3 = row[3 * RDG_GROW];
            // the four wave partials of the tile, then the tile's total into the running sum: Knuth's two-sum each time
            float s01 = a0 + a1, bb = s01 - a0;
            float e = (a0 - (s01 - bb)) + (a1 - bb);
            float s23 = a2 + a3; bb = s23 - a2;
            e += (a2 - (s23 - bb)) + (a3 - bb);
            float st = s01 + s23; bb = st - s01;
            e += (s01 - (st - bb)) + (s23 - bb);
            const float nw = acc + st; bb = nw - acc;
            e += (acc - (nw - bb)) + (st - bb);
            acc = nw;
            comp += e;
        }
    }
    grow[(size_t)g * RDG_GROW + c] = acc + comp;
}

// det != nullptr: deterministic mode -- `det` holds 4 * RDG_GROW floats per list position (zeroed by the caller)
int rdg_launch_render_bwd(const RdgDev& d, const float* bg, const void* geom_ws, const void* bin_ws,
                          int64_t capacity, const void* image_ws, const float* g_color, const float* g_depth,
                          const float* g_alpha, float* grow, hipStream_t s, float* det, const float* g_normal,
                          uint32_t* det_off, long long n_instances) {
    const RdgGeomLayout G = rdg_geom_layout(d.P);
    const int n_tiles = d.gx * d.gy;
    const RdgBinLayout B = rdg_bin_layout(capacity, n_tiles);
    const RdgImageLayout I = rdg_image_layout(d.H, d.W);
    const int npass = (rdg_key_bits(n_tiles) + RDG_SORT_BITS - 1) / RDG_SORT_BITS;
    const char* b = (const char*)bin_ws;
    const uint32_t* plist = (const uint32_t*)(b + ((npass & 1) ? B.vals_b : B.vals_a));
    const char* im = (const char*)image_ws;
    const int nblk = ((n_tiles + 7) / 8) * 8;
    const RdgSplitLayout SL = rdg_split_layout(capacity);
    const char* sp = b + B.split;
    const int split_min = (d.list_hints & 1) ? RDG_SPLIT_MIN : 0x7fffffff;
    const unsigned gseg = SL.max_seg < 2048u ? SL.max_seg : 2048u;
    // the split path's normal partial sums exist only if the forward composited normals
    if (g_normal && !d.render_normal) return rdg_set_error("render_bwd: a normal gradient needs render_normal in the forward");
#define RDG_BWD_LAUNCH4(DEPTH, DET, SEG, NRM, GRID, DST)                                                           \
    hipLaunchKernelGGL((rdg_render_bwd_kernel<DEPTH, DET, SEG, NRM>), dim3(GRID), dim3(256), 0, s, d.W, d.H, d.gx, n_tiles, bg, \
                       (const uint2*)(im + I.ranges), plist, (const RdgRec*)((const char*)geom_ws + G.rec),        \
                       (const float*)(im + I.final_T), (const uint32_t*)(im + I.n_contrib), g_color, g_depth,      \
                       g_alpha, g_normal, DST, (const unsigned long long*)(b + B.hit), split_min,                  \
                       (const uint32_t*)(sp + SL.header), (const uint4*)(sp + SL.work), (const float*)(sp + SL.seg_pix), \
                       (const uint32_t*)det_off, (const uint4*)((const char*)geom_ws + G.rectd))
#define RDG_BWD_LAUNCH(DEPTH, DET, SEG, GRID, DST)                                                                 \
    do { if (g_normal) RDG_BWD_LAUNCH4(DEPTH, DET, SEG, true, GRID, DST); else RDG_BWD_LAUNCH4(DEPTH, DET, SEG, false, GRID, DST); } while (0)
    if (det) {
        if (!det_off) return rdg_set_error("render_bwd: the deterministic mode needs the instance-offset buffer");
        if (d.P > 0)
            hipLaunchKernelGGL(rdg_det_offsets_kernel, dim3((d.P + RDG_PRE_BLOCK - 1) / RDG_PRE_BLOCK), dim3(RDG_PRE_BLOCK), 0, s,
                               d.P, (const uint32_t*)((const char*)geom_ws + G.tiles_touched),
                               (const uint32_t*)((const char*)geom_ws + G.block_sums), det_off);
        if (g_depth) RDG_BWD_LAUNCH(true, true, false, nblk, det); else RDG_BWD_LAUNCH(false, true, false, nblk, det);
        if ((d.list_hints & 1)) {
            if (g_depth) RDG_BWD_LAUNCH(true, true, true, gseg, det); else RDG_BWD_LAUNCH(false, true, true, gseg, det);
        }
        if (d.P > 0)
            hipLaunchKernelGGL(rdg_det_reduce_kernel, dim3((unsigned)(((long long)d.P * 16 + 255) / 256)), dim3(256), 0, s,
                               d.P, (const uint32_t*)((const char*)geom_ws + G.tiles_touched), (const uint32_t*)det_off,
                               n_instances, (const float*)det, grow);
    } else {
        if (g_depth) RDG_BWD_LAUNCH(true, false, false, nblk, grow); else RDG_BWD_LAUNCH(false, false, false, nblk, grow);
        if ((d.list_hints & 1)) {
            if (g_depth) RDG_BWD_LAUNCH(true, false, true, gseg, grow); else RDG_BWD_LAUNCH(false, false, true, gseg, grow);
        }
    }
#undef RDG_BWD_LAUNCH
#undef RDG_BWD_LAUNCH4
    return rdg_check_hip(hipGetLastError(), "render_bwd launch");
}
