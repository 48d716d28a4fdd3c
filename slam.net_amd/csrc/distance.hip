// distance.hip -- K1: batched CoreSLAM scan-to-map distance + arg-min (gfx950 only).
//
// Replaces CalculateDistanceSISD (CoreSLAM/CoreSLAMProcessor.cs:226-259) called from MonteCarloSearch
// (:624-653) on ParallelWorker threads (:674-710).  Arithmetic contract:
//   ix = (int)((px + c*X) - s*Y), iy = (int)((py + s*X) + c*Y) in binary32, one rounding per op,
//   no FMA (-ffp-contract=off), truncation toward zero; in-bounds pixels are summed as integers;
//   distance = (int)(sum*1024 / R) with R = ALL points (:253), int.MaxValue if none in bounds (:257).
// Integer sums make any evaluation order exact, so rays are processed in spatially compact blocks and
// candidates in theta-sorted order; the arg-min key (distance << 32 | flat index) restores the
// reference tie-break (first strictly smaller wins, :644,:700).
//
// Kernel design (k1_distance_tiled): a workgroup owns 1024 theta-consecutive candidates (256 lanes x 4
// candidates per lane) and a chunk of ray blocks.  For every ray block it bounds the end-point pixels of
// ALL its candidates by interval arithmetic on the very same float operations (rounding is monotone, so
// the box is rigorous), stages that HoleMap tile in LDS with coalesced 16-byte loads, and gathers from
// LDS: ~14 VALU + 1 ds_read_u16 per point evaluation and no bounds test (the box lies inside the map).
// When the 1024-candidate box does not fit the LDS budget (tails of the theta distribution, long rays)
// the four 256-candidate sub-batches get their own tiles; a sub-batch whose box still does not fit or
// that touches the map border falls back to bounds-checked global gathers.
#include "cs_internal.h"
#include "det_trig.h"
#include <stdlib.h>

#define K1_THREADS 256

// ---- shared pieces -----------------------------------------------------------------------------------
__device__ static inline void k1_coords(const float4 q, const float2 p, float &fx, float &fy)
{
    fx = q.x + q.z * p.x;  fx = fx - q.w * p.y;      // :240
    fy = q.y + q.w * p.x;  fy = fy + q.z * p.y;      // :241
}

// bounds-checked gather from the global map (also the NaN / overflow safe form when SAFE)
template <bool SAFE>
__device__ static inline void k1_gather_global(const uint16_t *__restrict__ map, int S, const float4 q, const float2 p,
                                               uint32_t &sum, uint32_t &cnt)
{
    float fx, fy;
    k1_coords(q, p, fx, fy);
    int ix, iy;
    if (SAFE) { ix = sh_f2i(fx); iy = sh_f2i(fy); }
    else      { ix = (int)fx;    iy = (int)fy; }         // |coords| < 1e9: v_cvt_i32_f32 saturates, never NaN
    const bool ok = ((unsigned)ix < (unsigned)S) & ((unsigned)iy < (unsigned)S);   // :244
    uint32_t v = 0;
    if (ok) v = map[(size_t)iy * S + ix];                // :246
    sum += v;
    cnt += ok ? 1u : 0u;
}

// distance + packed key of one candidate (:251-258)
__device__ static inline unsigned long long k1_finish(uint64_t sum, uint32_t cnt, int n_points, int flat,
                                                      int32_t *__restrict__ dist_out)
{
    const int32_t d = cnt > 0 ? (int32_t)((sum * 1024ull) / (uint64_t)n_points) : INT32_MAX;
    if (dist_out) dist_out[flat] = d;
    return ((unsigned long long)(uint32_t)d << 32) | (uint32_t)flat;
}

__device__ static inline void k1_wave_argmin(unsigned long long key, unsigned long long *__restrict__ key_out)
{
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(key, off, 64);
        key = o < key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key != ~0ull) atomicMin(key_out, key);
}

// ---- K1 main, global-gather form ---------------------------------------------------------------------------
// (a) whole-launch fallback for unsafe inputs or map sides that are not a multiple of 8 (plans == NULL);
// (b) TAIL companion of the tiled kernel: it evaluates exactly the (sub-batch, ray block) units whose plan kind
//     is GLOBAL -- candidates too sparse in theta, or rays too long, for an LDS tile to cover them.  Those are few,
//     and bounds-checked gathers with many independent waves in flight are the right tool for them.  A block
//     whose (sub-batch, chunk) has no such unit exits at once (tailmask).  Four independent gathers per iteration.
template <bool SAFE>
__global__ void __launch_bounds__(K1_THREADS)
k1_distance_global(const uint16_t *__restrict__ map, int S, const float2 *__restrict__ pts,
                   const int *__restrict__ rb_start, int n_rb, int blocks_per_chunk,
                   const float4 *__restrict__ pxcs, int count, uint2 *__restrict__ partial,
                   const int *__restrict__ plans, const int *__restrict__ tailmask, int nsub)
{
    const int chunk = blockIdx.y;
    if (tailmask && !tailmask[blockIdx.x * gridDim.y + chunk]) return;
    const int j = blockIdx.x * K1_THREADS + threadIdx.x;
    const int b0 = chunk * blocks_per_chunk;
    const int b1 = b0 + blocks_per_chunk < n_rb ? b0 + blocks_per_chunk : n_rb;
    const float4 q = pxcs[j < count ? j : count - 1];
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int b = b0; b < b1; b++) {
        if (plans) {       // sub-batch blockIdx.x = (group blockIdx.x / nsub, sub blockIdx.x % nsub)
            const int kind = plans[((size_t)(blockIdx.x / nsub) * n_rb + b) * 32 + (blockIdx.x % nsub) * 8 + 6];
            if (kind != 2) continue;
        }
        int r = rb_start[b];
        const int r1 = rb_start[b + 1];
        // eight gathers in flight per lane: these candidates have no locality, every gather is an L2/HBM round trip
        for (; r + 7 < r1; r += 8) {
            float fx[8], fy[8];
#pragma unroll
            for (int u = 0; u < 8; u++) k1_coords(q, pts[r + u], fx[u], fy[u]);
            uint32_t v[8]; bool ok[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                int ix, iy;
                if (SAFE) { ix = sh_f2i(fx[u]); iy = sh_f2i(fy[u]); } else { ix = (int)fx[u]; iy = (int)fy[u]; }
                ok[u] = ((unsigned)ix < (unsigned)S) & ((unsigned)iy < (unsigned)S);
                v[u] = map[ok[u] ? (size_t)iy * S + ix : 0];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
                s0 += ok[u] ? v[u] : 0u;         c0 += ok[u] ? 1u : 0u;
                s1 += ok[u + 1] ? v[u + 1] : 0u; c1 += ok[u + 1] ? 1u : 0u;
                s2 += ok[u + 2] ? v[u + 2] : 0u; c2 += ok[u + 2] ? 1u : 0u;
                s3 += ok[u + 3] ? v[u + 3] : 0u; c3 += ok[u + 3] ? 1u : 0u;
            }
        }
        for (; r < r1; r++) k1_gather_global<SAFE>(map, S, q, pts[r], s0, c0);
    }
    if (j < count) partial[(size_t)chunk * count + j] = make_uint2(s0 + s1 + s2 + s3, c0 + c1 + c2 + c3);
}

// ---- K1 main, LDS-tiled form ----------------------------------------------------------------------------
// A workgroup = 1024 lanes = 1024 theta-consecutive candidates ("group") = 4 sub-batches of 256.
// For every (group, ray block) the prep kernel has written a PLAN: either one shared tile that bounds the
// end points of all 1024 candidates, or one tile per sub-batch, or (per sub-batch) "global fallback".
#ifndef K1_WG
#define K1_WG 1024                     // lanes = candidates per workgroup ("group"); 512 or 1024
#endif
#define K1_SUB 256
#define K1_NSUB (K1_WG / K1_SUB)        // sub-batches per group
#define K1_NWAVES (K1_WG / 64)
#define K1_PF 5                        // prefetch registers (16-byte vectors) per lane
#define K1_PLAN_INTS 32                // 4 sub-batch records x 8 ints, each fully resolved
#define K1_KIND_OWN 0                  // the sub-batch has its own tile, staged by its 4 waves
#define K1_KIND_SHARED 1               // one tile for the whole group, staged by all 16 waves
#define K1_KIND_GLOBAL 2               // no tile fits (theta tail, long ray, map border): the unit goes to the tail kernel
#define K1_MAX_RB 2048                 // ray blocks the prep kernel can plan (beyond: whole-launch global path)
// sub-batch record: [0] x0a  [1] y0  [2] w8 (pitch, px)  [3] h  [4] lds byte offset  [5] shift = log2(lanes per row)
//                   [6] kind  [7] unused.
// Staging geometry: a wave-wide 16-byte load covers 64 >> shift tile rows of (1 << shift) vectors each
// (lanes beyond w8/8 vectors idle), so no division is needed to map lanes to tile vectors.

// byte address of pixel (ix,iy) in the staged tile: iy*pitch2 + 2*ix + kofs, as exactly two VALU ops
__device__ static inline unsigned k1_tile_addr(int ix, int iy, int pitch2, int kofs)
{
    unsigned t, a;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(t) : "v"(ix), "s"(kofs));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a) : "v"(iy), "s"(pitch2), "v"(t));
    return a;
}
// 16-bit LDS load at an absolute LDS byte address (saves the per-access `tile + offset` add)
typedef __attribute__((address_space(3))) const uint16_t k1_lds_u16;
__device__ static inline uint32_t k1_lds_load(unsigned addr) { return *(k1_lds_u16 *)(size_t)addr; }
// Ray points are broadcast from LDS with an ordinary ds_read_b64 whose address LOOKS lane-dependent to the
// compiler (zv is an opaque zero): measured on gfx950, v_readlane costs 4x and a VALU op with an SGPR source
// 2x the issue slots of a plain VGPR-operand VALU op, so the point must arrive in VGPRs.
typedef __attribute__((address_space(3))) const float k1_lds_f;
__device__ static inline float2 k1_point_lds(unsigned lds_addr)
{
    k1_lds_f *p = (k1_lds_f *)(size_t)lds_addr;
    return make_float2(p[0], p[1]);
}
#define K1_PTS_BYTES 512               // 64 rays x float2 at the start of dynamic LDS

// end-point pixel box of one ray over a candidate set, by interval arithmetic on the reference's own
// float operations: every rounding step is monotone, so [lo,hi] bounds every candidate's coordinate.
// b = {pxmin,pxmax,pymin,pymax,cmin,cmax,smin,smax}
__device__ static inline void k1_ray_box(const float *b, const float2 p, int &x0, int &y0, int &x1, int &y1)
{
    const float cx0 = b[4] * p.x, cx1 = b[5] * p.x, sy0 = b[6] * p.y, sy1 = b[7] * p.y;
    const float sx0 = b[6] * p.x, sx1 = b[7] * p.x, cy0 = b[4] * p.y, cy1 = b[5] * p.y;
    float xlo = b[0] + fminf(cx0, cx1);  xlo = xlo - fmaxf(sy0, sy1);
    float xhi = b[1] + fmaxf(cx0, cx1);  xhi = xhi - fminf(sy0, sy1);
    float ylo = b[2] + fminf(sx0, sx1);  ylo = ylo + fminf(cy0, cy1);
    float yhi = b[3] + fmaxf(sx0, sx1);  yhi = yhi + fmaxf(cy0, cy1);
    x0 = (int)xlo; x1 = (int)xhi; y0 = (int)ylo; y1 = (int)yhi;
}

// ---- prep + plan: one workgroup per candidate group -------------------------------------------------------
// MODE 0: pxcs given; 1: search_pose + jitter (:635-637); 2: poses.  (px,py,c,s) per :232-235.
#define K1_PLAN_BATCH 128
template <int MODE>
__global__ void __launch_bounds__(K1_WG)
k1_prep_plan(const float *__restrict__ src3, float bx, float by, float bth, float scale, float4 *__restrict__ pxcs,
             int count, unsigned long long *__restrict__ key, const float2 *__restrict__ pts,
             const int *__restrict__ rb_start, int n_rb, int S, int budget_shared, int budget_sub, int *__restrict__ plans,
             int *__restrict__ tailmask, int bpc_tail, int n_chunks_tail)
{
    __shared__ unsigned char kind_all[K1_MAX_RB * 4];            // plan kind per (ray block, sub-batch) of this group
    __shared__ int srb[K1_MAX_RB + 1];                           // rb_start staged once: every later use is an LDS read
    __shared__ float wred[K1_NWAVES][8];
    __shared__ float bnd[K1_NSUB + 1][8];                        // per sub-batch, then the whole group
    __shared__ int boxes[K1_PLAN_BATCH * (K1_NSUB + 1)][4];
    __shared__ float2 spts[K1_PLAN_BATCH * CS_RB_MAX / 4];       // the points of one batch of ray blocks (<= 2048)
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, g = blockIdx.x;
    const int j = g * K1_WG + t;
    if (j == 0) *key = ~0ull;
    if (plans) for (int i = t; i <= n_rb; i += K1_WG) srb[i] = rb_start[i];
    const int jc = j < count ? j : count - 1;
    float4 q;
    if (MODE == 0) {
        q = pxcs[jc];
    } else {
        float x, y, th;
        if (MODE == 1) { x = bx + src3[3 * jc]; y = by + src3[3 * jc + 1]; th = bth + src3[3 * jc + 2]; }
        else           { x = src3[3 * jc];      y = src3[3 * jc + 1];      th = src3[3 * jc + 2]; }
        float s, c;
        sh_det_sincosf(th, &s, &c);
        q.x = x * scale + 0.5f;
        q.y = y * scale + 0.5f;
        q.z = c * scale;
        q.w = s * scale;
        if (j < count) pxcs[j] = q;
    }
    // bounds: per wave -> per sub-batch (4 waves) -> whole group
    {
        float v[8] = { q.x, q.x, q.y, q.y, q.z, q.z, q.w, q.w };
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float x = v[k];
            for (int off = 32; off > 0; off >>= 1) {
                const float o = __shfl_xor(x, off, 64);
                x = (k & 1) ? fmaxf(x, o) : fminf(x, o);
            }
            if (lane == 0) wred[wid][k] = x;
        }
    }
    __syncthreads();
    if (t < K1_NSUB * 8) {
        const int sb = t >> 3, k = t & 7;
        float x = wred[sb * 4][k];
        for (int w = 1; w < 4; w++) x = (k & 1) ? fmaxf(x, wred[sb * 4 + w][k]) : fminf(x, wred[sb * 4 + w][k]);
        bnd[sb][k] = x;
    }
    __syncthreads();
    if (t < 8) {
        float x = bnd[0][t];
        for (int i = 1; i < K1_NSUB; i++) x = (t & 1) ? fmaxf(x, bnd[i][t]) : fminf(x, bnd[i][t]);
        bnd[K1_NSUB][t] = x;
    }
    __syncthreads();
    if (!plans) return;

    for (int base = 0; base < n_rb;) {
        int nb = n_rb - base < K1_PLAN_BATCH ? n_rb - base : K1_PLAN_BATCH;
        const int rbase = srb[base];
        while (srb[base + nb] - rbase > K1_PLAN_BATCH * CS_RB_MAX / 4) nb--;      // batch must fit spts (nb >= 1: a block has <= 64 rays)
        const int npts = srb[base + nb] - rbase;
        for (int i = t; i < npts; i += K1_WG) spts[i] = pts[rbase + i];
        __syncthreads();
        for (int pair = t; pair < nb * (K1_NSUB + 1); pair += K1_WG) {
            boxes[pair][0] = INT32_MAX; boxes[pair][1] = INT32_MAX; boxes[pair][2] = INT32_MIN; boxes[pair][3] = INT32_MIN;
        }
        __syncthreads();
        // one task = (ray block, candidate set, quarter of the block's rays); LDS atomics merge the quarters
        for (int task = t; task < nb * (K1_NSUB + 1) * 4; task += K1_WG) {
            const int pair = task >> 2, qr = task & 3;
            const int b = base + pair / (K1_NSUB + 1), set = pair % (K1_NSUB + 1);
            const int ra = srb[b] - rbase, rn = srb[b + 1] - srb[b];
            const int q0 = ra + (rn * qr) / 4, q1 = ra + (rn * (qr + 1)) / 4;
            float bb8[8];
#pragma unroll
            for (int k = 0; k < 8; k++) bb8[k] = bnd[set][k];
            int x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
            for (int r = q0; r < q1; r++) {
                int a0, b0, a1, b1;
                k1_ray_box(bb8, spts[r], a0, b0, a1, b1);
                x0 = min(x0, a0); y0 = min(y0, b0); x1 = max(x1, a1); y1 = max(y1, b1);
            }
            if (q1 > q0) {
                atomicMin(&boxes[pair][0], x0); atomicMin(&boxes[pair][1], y0);
                atomicMax(&boxes[pair][2], x1); atomicMax(&boxes[pair][3], y1);
            }
        }
        __syncthreads();
        if (t < nb * K1_NSUB) {
            // one thread per (ray block, sub-batch): the record is fully resolved, no second lookup in K1
            const int bb = t / K1_NSUB, sb = t % K1_NSUB;
            int rec[8] = { 0, 0, 8, 0, 0, 0, K1_KIND_GLOBAL, 0 };
            auto tile = [&](int set, int budget, int lds_off, int n_waves) -> bool {
                const int x0 = boxes[bb * (K1_NSUB + 1) + set][0], y0 = boxes[bb * (K1_NSUB + 1) + set][1];
                const int x1 = boxes[bb * (K1_NSUB + 1) + set][2], y1 = boxes[bb * (K1_NSUB + 1) + set][3];
                const bool inside = (x0 >= 0) & (y0 >= 0) & (x1 < S) & (y1 < S) & (x1 >= x0) & (y1 >= y0);
                if (!inside) return false;
                const int x0a = x0 & ~7;
                const int w8 = ((x1 - x0a + 1) + 7) & ~7;           // tile pitch in pixels (multiple of 8)
                const int h = y1 - y0 + 1;
                const int vpr = w8 >> 3;
                if (vpr > 64) return false;
                int shift = 0;
                while ((1 << shift) < vpr) shift++;
                const int rows_per_pass = n_waves * (64 >> shift);
                if ((long long)w8 * h * 2 > (long long)budget || h > K1_PF * rows_per_pass) return false;
                rec[0] = x0a; rec[1] = y0; rec[2] = w8; rec[3] = h; rec[4] = lds_off; rec[5] = shift;
                return true;
            };
            if (tile(K1_NSUB, budget_shared, 0, K1_NWAVES)) rec[6] = K1_KIND_SHARED;
            else if (tile(sb, budget_sub, sb * budget_sub, 4)) rec[6] = K1_KIND_OWN;
            kind_all[(base + bb) * 4 + sb] = (unsigned char)rec[6];
            int4 *dst = (int4 *)(plans + ((size_t)g * n_rb + base + bb) * K1_PLAN_INTS + sb * 8);
            dst[0] = make_int4(rec[0], rec[1], rec[2], rec[3]);
            dst[1] = make_int4(rec[4], rec[5], rec[6], rec[7]);
        }
        __syncthreads();
        base += nb;
    }
    // which (sub-batch, tail chunk) pairs hold at least one GLOBAL unit (only when a separate tail kernel is used)
    if (tailmask)
    for (int i = t; i < K1_NSUB * n_chunks_tail; i += K1_WG) {
        const int sb = i / n_chunks_tail, ct = i - sb * n_chunks_tail;
        const int c0 = ct * bpc_tail, c1 = c0 + bpc_tail < n_rb ? c0 + bpc_tail : n_rb;
        int any = 0;
        for (int b = c0; b < c1; b++) any |= (kind_all[b * 4 + sb] == K1_KIND_GLOBAL);
        tailmask[((size_t)g * K1_NSUB + sb) * n_chunks_tail + ct] = any;
    }
}

// ---- the distance kernel -------------------------------------------------------------------------------------
template <bool VERIFY, bool INLINE_GLOBAL>
__global__ void __launch_bounds__(K1_WG, 8)          // 8 waves / SIMD resident (<= 64 VGPRs): 32 waves per CU
k1_distance_tiled(const uint16_t *__restrict__ map, int S, const float2 *__restrict__ pts,
                  const int *__restrict__ rb_start, int n_rb, int blocks_per_chunk,
                  const float4 *__restrict__ pxcs, int count, const int *__restrict__ plans,
                  uint2 *__restrict__ partial,                      // [n_chunks][count]
                  unsigned int *__restrict__ verify_fail, int n_chunks_grid)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;   // LDS byte address

    const int t = threadIdx.x, lane = t & 63;
    int zv;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zv));                   // opaque zero (see k1_point_lds)
    const int sub = __builtin_amdgcn_readfirstlane(t >> 8);        // sub-batch of this wave
    // Workgroups are dispatched in blockIdx order; two orders, both measured on MI355X:
    //  - large launches (GLOBAL units go to the tail kernel): chunk-major, so the workgroups in flight work on the
    //    SAME ray blocks for neighbouring theta groups, whose tiles overlap almost completely (L2 reuse, ~1.5x);
    //  - small launches (GLOBAL units inline): group-major with the theta-extreme groups first (0, n-1, 1, n-2, ...):
    //    their global-gather units take longest, so they overlap the bulk instead of trailing it.
    const int ng_ = gridDim.x / (unsigned)n_chunks_grid;
    int g, chunk;
    if (INLINE_GLOBAL) {
        const int gi = blockIdx.x / (unsigned)n_chunks_grid;
        chunk = blockIdx.x - gi * n_chunks_grid;
        g = (gi & 1) ? ng_ - 1 - (gi >> 1) : (gi >> 1);
    } else {
        chunk = blockIdx.x / (unsigned)ng_;
        g = blockIdx.x - chunk * ng_;
    }
    const int j = g * K1_WG + t;
    const float4 q = pxcs[j < count ? j : count - 1];
    uint32_t sum = 0, cnt = 0;

    const int b0 = chunk * blocks_per_chunk;
    const int b1 = b0 + blocks_per_chunk < n_rb ? b0 + blocks_per_chunk : n_rb;

    // Tile staging through registers: the global loads of the NEXT ray block's tile are issued before the
    // current block is consumed and stay in flight meanwhile.  Exactly K1_PF loads are always issued (lanes
    // without work re-read row 0) so that the compiler can count them.
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);         // wave index in the workgroup
    uint4 R0, R1, R2, R3, R4;
    int d0, d1, d2, d3, d4;
#define K1_STAGE_ONE(Rk, dk, k)                                                                     \
    {                                                                                               \
        const int row = (wslot + (k) * nwv) * rpi + srow;                                           \
        const bool live = colok & (row < h_);                                                       \
        const int rr = live ? row : 0;                                                              \
        Rk = *(const uint4 *)(gbase + (size_t)rr * S);                                              \
        dk = live ? K1_PTS_BYTES + ldsb + rr * pitchb : -1;                                         \
    }
#define K1_PREFETCH(planp)                                                                          \
    {                                                                                               \
        const int4 pa_ = *(const int4 *)((planp) + sub * 8), pb_ = *(const int4 *)((planp) + sub * 8 + 4); \
        const int kind_ = pb_.z, shift_ = pb_.y, w8_ = pa_.z;                                       \
        const int h_ = kind_ == K1_KIND_GLOBAL ? 0 : pa_.w;                                         \
        const int rpi = 64 >> shift_;                           /* tile rows per wave-wide load */   \
        const int srow = lane >> shift_, scol = lane & ((1 << shift_) - 1);                         \
        const bool colok = scol < (w8_ >> 3);                                                       \
        const int cc = colok ? scol : 0;                                                            \
        const uint16_t *__restrict__ gbase = map + (size_t)pa_.y * S + pa_.x + (cc << 3);           \
        const int pitchb = w8_ << 1;                                                                \
        const int ldsb = pb_.x + (cc << 4);                                                         \
        const int wslot = kind_ == K1_KIND_SHARED ? wv : (wv & 3);                                  \
        const int nwv = kind_ == K1_KIND_SHARED ? K1_NWAVES : 4;                                    \
        K1_STAGE_ONE(R0, d0, 0) K1_STAGE_ONE(R1, d1, 1) K1_STAGE_ONE(R2, d2, 2)                     \
        K1_STAGE_ONE(R3, d3, 3) K1_STAGE_ONE(R4, d4, 4)                                             \
    }

    const int *__restrict__ plan = plans + ((size_t)g * n_rb + b0) * K1_PLAN_INTS;
    K1_PREFETCH(plan)
    // the ray block's points travel the same way: lane r of every wave holds ray r of the NEXT block
    int r0n = rb_start[b0], nrn = rb_start[b0 + 1] - r0n;
    float2 mypt_next = pts[r0n + (lane < nrn ? lane : 0)];
    for (int b = b0; b < b1; b++) {
        __syncthreads();                                           // the previous tile is no longer read
        if (d0 >= 0) *(uint4 *)(smem + d0) = R0;
        if (d1 >= 0) *(uint4 *)(smem + d1) = R1;
        if (d2 >= 0) *(uint4 *)(smem + d2) = R2;
        if (d3 >= 0) *(uint4 *)(smem + d3) = R3;
        if (d4 >= 0) *(uint4 *)(smem + d4) = R4;
        if (wv == 0) ((float2 *)smem)[lane] = mypt_next;           // this block's ray points (lanes >= nr: duplicates)
        const int nr = nrn;                                        // <= CS_RB_MAX (32) <= 64 lanes
        __syncthreads();
        const int *__restrict__ cur = plan;
        plan += K1_PLAN_INTS;
        {   // issue the next block's loads now; nothing in the compute loop below waits on vector memory
            const bool more = b + 1 < b1;
            const int *__restrict__ nplan = more ? plan : cur;
            K1_PREFETCH(nplan)
            const int bn = more ? b + 1 : b;
            r0n = rb_start[bn]; nrn = rb_start[bn + 1] - r0n;
            mypt_next = pts[r0n + (lane < nrn ? lane : 0)];
        }

        const int4 ca = *(const int4 *)(cur + sub * 8), cb = *(const int4 *)(cur + sub * 8 + 4);
        const int kind = cb.z;
        if (kind < K1_KIND_GLOBAL) {
            const int x0a = ca.x, y0 = ca.y, w8 = ca.z, lds = cb.x;
            const int pitch2 = w8 << 1;
            const int kofs = (int)smem_lds + K1_PTS_BYTES + lds - ((y0 * w8 + x0a) << 1);
            const unsigned pbase = smem_lds + (unsigned)zv;
            uint32_t sumb = 0;                                     // second accumulator: two gathers in flight
            int r = 0;
            for (; r + 1 < nr; r += 2) {
                const float2 pa = k1_point_lds(pbase + r * 8), pb = k1_point_lds(pbase + r * 8 + 8);
                float fxa, fya, fxb, fyb;
                k1_coords(q, pa, fxa, fya);
                k1_coords(q, pb, fxb, fyb);
                const int ixa = (int)fxa, iya = (int)fya, ixb = (int)fxb, iyb = (int)fyb;
                if (VERIFY) {
                    const int h = ca.w;
                    if (ixa < x0a || ixa >= x0a + w8 || iya < y0 || iya >= y0 + h ||
                        ixb < x0a || ixb >= x0a + w8 || iyb < y0 || iyb >= y0 + h) { atomicAdd(verify_fail, 1u); continue; }
                }
                const uint32_t va = k1_lds_load(k1_tile_addr(ixa, iya, pitch2, kofs));
                const uint32_t vb = k1_lds_load(k1_tile_addr(ixb, iyb, pitch2, kofs));
                if (VERIFY) {                                      // the staged tile must equal the map
                    if (va != map[(size_t)iya * S + ixa]) atomicAdd(verify_fail, 1u);
                    if (vb != map[(size_t)iyb * S + ixb]) atomicAdd(verify_fail, 1u);
                }
                sum += va;
                sumb += vb;
            }
            if (r < nr) {
                float fx, fy;
                k1_coords(q, k1_point_lds(pbase + r * 8), fx, fy);
                const int ix = (int)fx, iy = (int)fy;
                bool okv = true;
                if (VERIFY) {
                    const int h = ca.w;
                    if (ix < x0a || ix >= x0a + w8 || iy < y0 || iy >= y0 + h) { atomicAdd(verify_fail, 1u); okv = false; }
                }
                if (okv) {
                    const uint32_t v = k1_lds_load(k1_tile_addr(ix, iy, pitch2, kofs));
                    if (VERIFY) { if (v != map[(size_t)iy * S + ix]) atomicAdd(verify_fail, 1u); }
                    sum += v;
                }
            }
            sum += sumb;
            cnt += (uint32_t)nr;
            if (VERIFY && (t & (K1_SUB - 1)) == 0) atomicAdd(verify_fail + (kind == K1_KIND_SHARED ? 1 : 2), (unsigned)nr);
        } else if (INLINE_GLOBAL) {
            // No tile covers this unit (theta tail, long ray, map border): bounds-checked global gathers, four in
            // flight per lane -- these candidates have no locality, every gather is an L2 / MALL round trip; the
            // other waves of the CU keep computing from LDS meanwhile.  (Large launches leave these units to the
            // separate tail kernel instead: INLINE_GLOBAL == false.)
            const unsigned pbase = smem_lds + (unsigned)zv;
            int r = 0;
            for (; r + 3 < nr; r += 4) {
                uint32_t v[4]; bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    float fx, fy;
                    k1_coords(q, k1_point_lds(pbase + (r + u) * 8), fx, fy);
                    const int ix = (int)fx, iy = (int)fy;
                    ok[u] = ((unsigned)ix < (unsigned)S) & ((unsigned)iy < (unsigned)S);
                    v[u] = map[ok[u] ? (size_t)iy * S + ix : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) { sum += ok[u] ? v[u] : 0u; cnt += ok[u] ? 1u : 0u; }
            }
            for (; r < nr; r++) k1_gather_global<false>(map, S, q, k1_point_lds(pbase + r * 8), sum, cnt);
            if (VERIFY && (t & (K1_SUB - 1)) == 0) atomicAdd(verify_fail + 3, (unsigned)nr);
        }
    }
#undef K1_PREFETCH
#undef K1_STAGE_ONE

    // ---- epilogue: one (sum, in-bounds count) per candidate and chunk; K1r finishes --------------------------
    if (j < count) partial[(size_t)chunk * count + j] = make_uint2(sum, cnt);
}

// ---- K1r: per-candidate reduction of the chunk partials + arg-min ----------------------------------------
// A block works on 64 candidates at a time with its 4 waves splitting the partial rows; blocks grid-stride
// over the candidates and issue ONE atomicMin each (same-address device atomics serialise at ~12 ns each on
// MI355X, so per-wave atomics would dominate at K >= 1e5).
__global__ void __launch_bounds__(256)
k1_reduce(const uint2 *__restrict__ partial, int n_chunks, int n_chunks_tail, const int *__restrict__ tailmask,
          int count, int n_points, const int *__restrict__ ev_idx, int32_t *__restrict__ dist_out,
          unsigned long long *__restrict__ key_out)
{
    __shared__ uint32_t ssum[4][64], scnt[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long key = ~0ull;
    for (int base = blockIdx.x * 64; base < count; base += gridDim.x * 64) {
        const int j = base + lane;
        uint32_t sum = 0, cnt = 0;
        if (j < count) {
            // rows [0, n_chunks): tiled (or whole-launch global) kernel
            int c = w;
            for (; c + 12 < n_chunks; c += 16) {
                const uint2 p0 = partial[(size_t)c * count + j], p1 = partial[(size_t)(c + 4) * count + j];
                const uint2 p2 = partial[(size_t)(c + 8) * count + j], p3 = partial[(size_t)(c + 12) * count + j];
                sum += p0.x + p1.x + p2.x + p3.x; cnt += p0.y + p1.y + p2.y + p3.y;
            }
            for (; c < n_chunks; c += 4) {
                const uint2 p = partial[(size_t)c * count + j];
                sum += p.x; cnt += p.y;
            }
            // rows [n_chunks, n_chunks + n_chunks_tail): tail kernel, only where the sub-batch has GLOBAL units
            if (tailmask) {
                const int *tm = tailmask + (size_t)(j >> 8) * n_chunks_tail;
                for (int ct = w; ct < n_chunks_tail; ct += 4)
                    if (tm[ct]) {
                        const uint2 p = partial[(size_t)(n_chunks + ct) * count + j];
                        sum += p.x; cnt += p.y;
                    }
            }
        }
        ssum[w][lane] = sum; scnt[w][lane] = cnt;
        __syncthreads();
        if (w == 0 && j < count) {
            const uint64_t s = (uint64_t)ssum[0][lane] + ssum[1][lane] + ssum[2][lane] + ssum[3][lane];
            const uint32_t c = scnt[0][lane] + scnt[1][lane] + scnt[2][lane] + scnt[3][lane];
            const unsigned long long k = k1_finish(s, c, n_points, ev_idx ? ev_idx[j] : j, dist_out);
            key = k < key ? k : key;
        }
        __syncthreads();
    }
    if (w == 0) {
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_down(key, off, 64);
            key = o < key ? o : key;
        }
        if (lane == 0 && key != ~0ull) atomicMin(key_out, key);
    }
}

// ---- host side --------------------------------------------------------------------------------------
int32_t cs_alloc_candidates(slamhip_cs *cs, int count)
{
    if (count <= cs->cap_cand) return SLAMHIP_OK;
    int cap = count + (count >> 2) + 256;
    if (cs->d_ev_off) (void)hipFree(cs->d_ev_off);
    if (cs->d_ev_idx) (void)hipFree(cs->d_ev_idx);
    if (cs->d_pxcs) (void)hipFree(cs->d_pxcs);
    if (cs->d_dist) (void)hipFree(cs->d_dist);
    cs->d_ev_off = nullptr; cs->d_ev_idx = nullptr; cs->d_pxcs = nullptr; cs->d_dist = nullptr; cs->cap_cand = 0;
    SH_HIP(hipMalloc(&cs->d_ev_off, sizeof(float) * 3 * (size_t)cap));
    SH_HIP(hipMalloc(&cs->d_ev_idx, sizeof(int) * (size_t)cap));
    SH_HIP(hipMalloc(&cs->d_pxcs, sizeof(float4) * (size_t)cap));
    SH_HIP(hipMalloc(&cs->d_dist, sizeof(int32_t) * (size_t)cap));
    cs->cap_cand = cap;
    cs->shard_first = -1; cs->shard_count = -1;
    return SLAMHIP_OK;
}

static int32_t ensure_partial(slamhip_cs *cs, size_t need)
{
    if (need <= cs->cap_partial) return SLAMHIP_OK;
    if (cs->d_partial) (void)hipFree(cs->d_partial);
    cs->d_partial = nullptr; cs->cap_partial = 0;
    need += need / 4;
    SH_HIP(hipMalloc(&cs->d_partial, sizeof(uint2) * need));
    cs->cap_partial = need;
    return SLAMHIP_OK;
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

static int32_t ensure_plans(slamhip_cs *cs, size_t ints)
{
    if (ints <= cs->cap_plans) return SLAMHIP_OK;
    if (cs->d_plans) (void)hipFree(cs->d_plans);
    cs->d_plans = nullptr; cs->cap_plans = 0;
    ints += ints / 4;
    SH_HIP(hipMalloc(&cs->d_plans, sizeof(int) * ints));
    cs->cap_plans = ints;
    return SLAMHIP_OK;
}

// Candidate preparation (+ tile plans) followed by K1 (+ K1r) over `count` candidates in evaluation order
// (d_ev_idx maps to flat indices).  mode 0: d_pxcs already holds (px,py,c,s); 1: d_ev_off holds jitters added
// to `pose`; 2: d_ev_off holds poses.  The packed arg-min key is min-ed into key_dst (armed by the prep
// kernel).  Asynchronous on the context's stream.
int32_t cs_launch_distance(slamhip_cs *cs, int mode, const float pose[3], int count, bool want_dist, bool cand_sane,
                           uint64_t *key_dst)
{
    slamhip_ctx *ctx = cs->ctx;
    if (cs->n_points <= 0) SH_FAIL(SLAMHIP_ERR_STATE, "no scan set (slamhip_cs_set_scan)");
    static const int force_global = env_int("SLAMHIP_K1_GLOBAL", 0);
    static const int verify = env_int("SLAMHIP_K1_VERIFY", 0);
    static const int tile_kb = env_int("SLAMHIP_K1_TILE_KB", 60);
    static const int target_wgs = env_int("SLAMHIP_K1_TARGET_WGS", 768);
    const bool sane = cs->pts_sane && cand_sane;
    const bool tiled = sane && (cs->hs % 8 == 0) && !force_global && cs->n_rb <= K1_MAX_RB;
    const int n_rb = cs->n_rb;
    int32_t *dist = want_dist ? cs->d_dist : nullptr;
    unsigned long long *key = (unsigned long long *)key_dst;
    const float bx = pose ? pose[0] : 0.f, by = pose ? pose[1] : 0.f, bth = pose ? pose[2] : 0.f;
    const int n_groups = sh_div_up(count, K1_WG);

    int budget_shared = tile_kb * 1024;
    if (budget_shared > K1_WG * K1_PF * 16) budget_shared = K1_WG * K1_PF * 16;      // what 4 prefetch vectors/lane can stage
    int budget_sub = budget_shared / K1_NSUB;
    static const int sub_kb = env_int("SLAMHIP_K1_SUB_KB", 0);         // debugging: decouple the two budgets
    static const int no_shared = env_int("SLAMHIP_K1_NOSHARED", 0);
    if (sub_kb > 0) budget_sub = sub_kb * 1024;
    budget_sub &= ~15;
    if (budget_sub > K1_SUB * K1_PF * 16) budget_sub = K1_SUB * K1_PF * 16;
    const int budget_shared_eff = no_shared ? 0 : budget_shared;
    // tail kernel geometry: 256-lane blocks, ~32 rays per chunk
    int bpc_t = sh_div_up(32 * n_rb, cs->n_points > 0 ? cs->n_points : 1);
    if (bpc_t < 1) bpc_t = 1;
    if (bpc_t > n_rb) bpc_t = n_rb;
    const int n_chunks_t = sh_div_up(n_rb, bpc_t);
    if (tiled) SH_TRY(ensure_plans(cs, (size_t)n_groups * n_rb * K1_PLAN_INTS + (size_t)n_groups * K1_NSUB * n_chunks_t));
    // GLOBAL units (no LDS tile fits): small launches evaluate them inline in the tiled kernel, where the other waves
    // hide their latency; from ~48k candidates on, a separate many-wave tail kernel is faster (measured on MI355X)
    static const int tail_threshold = env_int("SLAMHIP_K1_TAIL_KERNEL_FROM", 49152);
    const bool use_tail_kernel = tiled && count >= tail_threshold;
    int *tailp = use_tail_kernel ? cs->d_plans + (size_t)n_groups * n_rb * K1_PLAN_INTS : nullptr;   // tailmask [sub-batch][tail chunk]
    {
        sh_timer t(ctx, SLAMHIP_K_CS_PREP);
        int *plans = tiled ? cs->d_plans : nullptr;
#define K1_PREP(M) hipLaunchKernelGGL(k1_prep_plan<M>, dim3(n_groups), dim3(K1_WG), 0, ctx->stream, (const float *)cs->d_ev_off, \
                       bx, by, bth, cs->hscale, cs->d_pxcs, count, key, (const float2 *)cs->d_pts_sorted, (const int *)cs->d_rb_start, \
                       n_rb, cs->hs, budget_shared_eff, budget_sub, plans, tailp, bpc_t, n_chunks_t)
        if (mode == 0) K1_PREP(0); else if (mode == 1) K1_PREP(1); else K1_PREP(2);
#undef K1_PREP
    }
    static const int dump = env_int("SLAMHIP_K1_DUMP", 0);
    if (dump && tiled) {                                           // debugging aid: histogram of plan kinds
        std::vector<int> h((size_t)n_groups * n_rb * K1_PLAN_INTS + (size_t)n_groups * K1_NSUB * n_chunks_t);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(h.data(), cs->d_plans, sizeof(int) * h.size(), hipMemcpyDeviceToHost);
        int kinds[4] = { 0, 0, 0, 0 }, tails = 0;
        for (size_t i = 0; i < (size_t)n_groups * n_rb * 4; i++) kinds[h[i * 8 + 6] & 3]++;

        fprintf(stderr, "[slamhip] K1 plans: count %d groups %d n_rb %d budgets %d/%d | own %d shared %d global %d (-) %d | tail (sub-batch,chunk) pairs %d\n",
                count, n_groups, n_rb, budget_shared_eff, budget_sub, kinds[0], kinds[1], kinds[2], kinds[3], tails);
        for (int b = 0; b < (n_rb < 3 ? n_rb : 3); b++)
            for (int sb = 0; sb < 4; sb++) {
                const int *r = &h[((size_t)0 * n_rb + b) * K1_PLAN_INTS + sb * 8];
                fprintf(stderr, "   g0 b%d sb%d: x0a %d y0 %d w8 %d h %d lds %d shift %d kind %d\n", b, sb, r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
            }
    }
    const int rblocks = sh_div_up(count, 64) < 512 ? sh_div_up(count, 64) : 512;
    if (tiled) {
        int bpc = (int)(((long long)n_groups * n_rb) / target_wgs);
        if (bpc < 1) bpc = 1;
        if (bpc > n_rb) bpc = n_rb;
        const int n_chunks = sh_div_up(n_rb, bpc);
        SH_TRY(ensure_partial(cs, (size_t)(n_chunks + (use_tail_kernel ? n_chunks_t : 0)) * count));
        const size_t lds = (size_t)K1_PTS_BYTES + (size_t)(budget_shared > K1_NSUB * budget_sub ? budget_shared : K1_NSUB * budget_sub);
        {
            sh_timer t(ctx, SLAMHIP_K_CS_DISTANCE);
            dim3 grid(n_groups * n_chunks);
#define K1_LAUNCH(V, I) hipLaunchKernelGGL((k1_distance_tiled<V, I>), grid, dim3(K1_WG), lds, ctx->stream, cs->d_hole, cs->hs, \
                       cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, cs->d_plans, (uint2 *)cs->d_partial, cs->d_verify, n_chunks)
            if (verify) { if (use_tail_kernel) K1_LAUNCH(true, false); else K1_LAUNCH(true, true); }
            else        { if (use_tail_kernel) K1_LAUNCH(false, false); else K1_LAUNCH(false, true); }
#undef K1_LAUNCH
            if (use_tail_kernel)
                hipLaunchKernelGGL(k1_distance_global<false>, dim3(n_groups * K1_NSUB, n_chunks_t), dim3(K1_THREADS), 0, ctx->stream,
                                   cs->d_hole, cs->hs, cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc_t, cs->d_pxcs, count,
                                   (uint2 *)cs->d_partial + (size_t)n_chunks * count, (const int *)cs->d_plans, (const int *)tailp, K1_NSUB);
        }
        {
            sh_timer t(ctx, SLAMHIP_K_CS_REDUCE);
            hipLaunchKernelGGL(k1_reduce, dim3(rblocks), dim3(256), 0, ctx->stream, (const uint2 *)cs->d_partial, n_chunks,
                               use_tail_kernel ? n_chunks_t : 0, (const int *)tailp, count, cs->n_points, cs->d_ev_idx, dist, key);
        }
    } else {
        int bpc = (int)(((long long)sh_div_up(count, K1_THREADS) * n_rb) / 4096);
        if (bpc < 1) bpc = 1;
        if (bpc > n_rb) bpc = n_rb;
        const int n_chunks = sh_div_up(n_rb, bpc);
        SH_TRY(ensure_partial(cs, (size_t)n_chunks * count));
        {
            sh_timer t(ctx, SLAMHIP_K_CS_DISTANCE);
            dim3 grid(sh_div_up(count, K1_THREADS), n_chunks);
            if (sane)
                hipLaunchKernelGGL(k1_distance_global<false>, grid, dim3(K1_THREADS), 0, ctx->stream, cs->d_hole, cs->hs,
                                   cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, (uint2 *)cs->d_partial,
                                   (const int *)nullptr, (const int *)nullptr, 1);
            else
                hipLaunchKernelGGL(k1_distance_global<true>, grid, dim3(K1_THREADS), 0, ctx->stream, cs->d_hole, cs->hs,
                                   cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, (uint2 *)cs->d_partial,
                                   (const int *)nullptr, (const int *)nullptr, 1);
        }
        {
            sh_timer t(ctx, SLAMHIP_K_CS_REDUCE);
            hipLaunchKernelGGL(k1_reduce, dim3(rblocks), dim3(256), 0, ctx->stream, (const uint2 *)cs->d_partial, n_chunks, 0,
                               (const int *)nullptr, count, cs->n_points, cs->d_ev_idx, dist, key);
        }
    }
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}
