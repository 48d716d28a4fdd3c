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
// (a) whole-launch fallback for unsafe inputs or map sides that are not a multiple of 8 (tail == NULL);
// (b) companion of the tiled kernel for theta-TAIL sub-batches: candidates so sparse in theta that no LDS tile
//     covers them; they are few, so plain bounds-checked gathers with many waves in flight are the right tool.
//     Blocks whose sub-batch is not flagged exit immediately.  Four independent gathers per iteration.
template <bool SAFE>
__global__ void __launch_bounds__(K1_THREADS)
k1_distance_global(const uint16_t *__restrict__ map, int S, const float2 *__restrict__ pts,
                   const int *__restrict__ rb_start, int n_rb, int blocks_per_chunk,
                   const float4 *__restrict__ pxcs, int count, uint2 *__restrict__ partial,
                   const int *__restrict__ tail)
{
    if (tail && !tail[blockIdx.x]) return;
    const int j = blockIdx.x * K1_THREADS + threadIdx.x;
    const int chunk = blockIdx.y;
    const int b0 = chunk * blocks_per_chunk;
    const int b1 = b0 + blocks_per_chunk < n_rb ? b0 + blocks_per_chunk : n_rb;
    const int r0 = rb_start[b0], r1 = rb_start[b1];
    const float4 q = pxcs[j < count ? j : count - 1];
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        k1_gather_global<SAFE>(map, S, q, pts[r], s0, c0);
        k1_gather_global<SAFE>(map, S, q, pts[r + 1], s1, c1);
        k1_gather_global<SAFE>(map, S, q, pts[r + 2], s2, c2);
        k1_gather_global<SAFE>(map, S, q, pts[r + 3], s3, c3);
    }
    for (; r < r1; r++) k1_gather_global<SAFE>(map, S, q, pts[r], s0, c0);
    if (j < count) partial[(size_t)chunk * count + j] = make_uint2(s0 + s1 + s2 + s3, c0 + c1 + c2 + c3);
}

// ---- K1 main, LDS-tiled form ----------------------------------------------------------------------------
// A workgroup = 1024 lanes = 1024 theta-consecutive candidates ("group") = 4 sub-batches of 256.
// For every (group, ray block) the prep kernel has written a PLAN: either one shared tile that bounds the
// end points of all 1024 candidates, or one tile per sub-batch, or (per sub-batch) "global fallback".
#define K1_WG 1024
#define K1_SUB 256
#define K1_PF 5                        // prefetch registers (16-byte vectors) per lane
#define K1_PLAN_INTS 32                // 4 sub-batch records x 8 ints, each fully resolved
#define K1_KIND_OWN 0                  // the sub-batch has its own tile, staged by its 4 waves
#define K1_KIND_SHARED 1               // one tile for the whole group, staged by all 16 waves
#define K1_KIND_GLOBAL 2               // no tile: bounds-checked global gathers
#define K1_KIND_SKIP 3                 // the whole sub-batch is a theta-tail: handled by k1_distance_tail instead
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
             int *__restrict__ tail, int n_points)
{
    __shared__ int gl_rays[4];                                   // rays per sub-batch that no tile covers
    __shared__ float wred[16][8];
    __shared__ float bnd[5][8];
    __shared__ int boxes[K1_PLAN_BATCH * 5][4];
    __shared__ float2 spts[K1_PLAN_BATCH * CS_RB_MAX / 4];       // the points of one batch of ray blocks (<= 2048)
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6, g = blockIdx.x;
    const int j = g * K1_WG + t;
    if (j == 0) *key = ~0ull;
    if (t < 4) gl_rays[t] = 0;
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
    if (t < 32) {
        const int sb = t >> 3, k = t & 7;
        float x = wred[sb * 4][k];
        for (int w = 1; w < 4; w++) x = (k & 1) ? fmaxf(x, wred[sb * 4 + w][k]) : fminf(x, wred[sb * 4 + w][k]);
        bnd[sb][k] = x;
    }
    __syncthreads();
    if (t < 8) {
        float x = bnd[0][t];
        for (int i = 1; i < 4; i++) x = (t & 1) ? fmaxf(x, bnd[i][t]) : fminf(x, bnd[i][t]);
        bnd[4][t] = x;
    }
    __syncthreads();
    if (!plans) return;

    for (int base = 0; base < n_rb;) {
        int nb = n_rb - base < K1_PLAN_BATCH ? n_rb - base : K1_PLAN_BATCH;
        const int rbase = rb_start[base];
        while (rb_start[base + nb] - rbase > K1_PLAN_BATCH * CS_RB_MAX / 4) nb--;      // batch must fit spts (nb >= 1: a block has <= 64 rays)
        const int npts = rb_start[base + nb] - rbase;
        for (int i = t; i < npts; i += K1_WG) spts[i] = pts[rbase + i];
        __syncthreads();
        for (int pair = t; pair < nb * 5; pair += K1_WG) {
            const int b = base + pair / 5, set = pair % 5;
            int x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
            for (int r = rb_start[b] - rbase; r < rb_start[b + 1] - rbase; r++) {
                int a0, b0, a1, b1;
                k1_ray_box(bnd[set], spts[r], a0, b0, a1, b1);
                x0 = min(x0, a0); y0 = min(y0, b0); x1 = max(x1, a1); y1 = max(y1, b1);
            }
            boxes[pair][0] = x0; boxes[pair][1] = y0; boxes[pair][2] = x1; boxes[pair][3] = y1;
        }
        __syncthreads();
        if (t < nb * 4) {
            // one thread per (ray block, sub-batch): the record is fully resolved, no second lookup in K1
            const int bb = t >> 2, sb = t & 3;
            int rec[8] = { 0, 0, 8, 0, 0, 0, K1_KIND_GLOBAL, 0 };
            auto tile = [&](int set, int budget, int lds_off, int n_waves) -> bool {
                const int x0 = boxes[bb * 5 + set][0], y0 = boxes[bb * 5 + set][1];
                const int x1 = boxes[bb * 5 + set][2], y1 = boxes[bb * 5 + set][3];
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
            if (tile(4, budget_shared, 0, 16)) rec[6] = K1_KIND_SHARED;
            else if (tile(sb, budget_sub, sb * budget_sub, 4)) rec[6] = K1_KIND_OWN;
            else atomicAdd(&gl_rays[sb], rb_start[base + bb + 1] - rb_start[base + bb]);
            int4 *dst = (int4 *)(plans + ((size_t)g * n_rb + base + bb) * K1_PLAN_INTS + sb * 8);
            dst[0] = make_int4(rec[0], rec[1], rec[2], rec[3]);
            dst[1] = make_int4(rec[4], rec[5], rec[6], rec[7]);
        }
        __syncthreads();
        base += nb;
    }
    // theta-tail sub-batches (more than a quarter of their rays uncovered) leave the tiled kernel altogether
    if (t < 4) tail[g * 4 + t] = (gl_rays[t] * 4 > n_points) ? 1 : 0;
    // (records of SHARED tiles stay: all 16 waves are needed to stage a shared tile; what a tail sub-batch
    //  computes there is simply ignored, K1r reads only its tail-kernel partials)
    for (int i = t; i < n_rb * 4; i += K1_WG) {
        const int sb = i & 3;
        int *kindp = plans + ((size_t)g * n_rb + (i >> 2)) * K1_PLAN_INTS + sb * 8 + 6;
        if (gl_rays[sb] * 4 > n_points && *kindp != K1_KIND_SHARED) *kindp = K1_KIND_SKIP;
    }
}

// ---- the distance kernel -------------------------------------------------------------------------------------
template <bool VERIFY>
__global__ void __launch_bounds__(K1_WG)
k1_distance_tiled(const uint16_t *__restrict__ map, int S, const float2 *__restrict__ pts,
                  const int *__restrict__ rb_start, int n_rb, int blocks_per_chunk,
                  const float4 *__restrict__ pxcs, int count, const int *__restrict__ plans,
                  uint2 *__restrict__ partial,                      // [n_chunks][count]
                  unsigned int *__restrict__ verify_fail)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;   // LDS byte address

    const int t = threadIdx.x, lane = t & 63;
    int zv;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zv));                   // opaque zero (see k1_point_lds)
    const int sub = __builtin_amdgcn_readfirstlane(t >> 8);        // sub-batch of this wave
    const int g = blockIdx.x, chunk = blockIdx.y;
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
        const int h_ = kind_ >= K1_KIND_GLOBAL ? 0 : pa_.w;                                         \
        const int rpi = 64 >> shift_;                           /* tile rows per wave-wide load */   \
        const int srow = lane >> shift_, scol = lane & ((1 << shift_) - 1);                         \
        const bool colok = scol < (w8_ >> 3);                                                       \
        const int cc = colok ? scol : 0;                                                            \
        const uint16_t *__restrict__ gbase = map + (size_t)pa_.y * S + pa_.x + (cc << 3);           \
        const int pitchb = w8_ << 1;                                                                \
        const int ldsb = pb_.x + (cc << 4);                                                         \
        const int wslot = kind_ == K1_KIND_SHARED ? wv : (wv & 3);                                  \
        const int nwv = kind_ == K1_KIND_SHARED ? 16 : 4;                                           \
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
        } else if (kind == K1_KIND_GLOBAL) {
            // box too large for LDS or touching the map border: bounds-checked global gathers
            const unsigned pbase = smem_lds + (unsigned)zv;
            for (int r = 0; r < nr; r++) k1_gather_global<false>(map, S, q, k1_point_lds(pbase + r * 8), sum, cnt);
            if (VERIFY && (t & (K1_SUB - 1)) == 0) atomicAdd(verify_fail + 3, (unsigned)nr);
        }
    }
#undef K1_PREFETCH
#undef K1_STAGE_ONE

    // ---- epilogue: one (sum, in-bounds count) per candidate and chunk; K1r finishes --------------------------
    if (j < count) partial[(size_t)chunk * count + j] = make_uint2(sum, cnt);
}

// ---- K1r: per-candidate reduction of the chunk partials + arg-min ----------------------------------------
// Grid-stride over candidates; ONE atomicMin per workgroup: same-address device atomics serialise at
// ~12 ns each on MI355X, so per-wave atomics would dominate at K >= 1e5.
__global__ void __launch_bounds__(256)
k1_reduce(const uint2 *__restrict__ partial, int n_chunks, int n_chunks_tail, const int *__restrict__ tail,
          int count, int n_points, const int *__restrict__ ev_idx, int32_t *__restrict__ dist_out,
          unsigned long long *__restrict__ key_out)
{
    __shared__ unsigned long long wkey[4];
    unsigned long long key = ~0ull;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < count; j += gridDim.x * 256) {
        uint32_t sum = 0, cnt = 0;
        // rows [0, n_chunks) come from the tiled (or whole-launch global) kernel, rows [n_chunks, +n_chunks_tail)
        // from the tail kernel; a candidate's sub-batch (j >> 8) tells which set holds its partials
        const bool is_tail = tail && tail[j >> 8];
        int c = is_tail ? n_chunks : 0;
        const int c_end = is_tail ? n_chunks + n_chunks_tail : n_chunks;
        for (; c + 3 < c_end; c += 4) {
            const uint2 p0 = partial[(size_t)c * count + j], p1 = partial[(size_t)(c + 1) * count + j];
            const uint2 p2 = partial[(size_t)(c + 2) * count + j], p3 = partial[(size_t)(c + 3) * count + j];
            sum += p0.x + p1.x + p2.x + p3.x; cnt += p0.y + p1.y + p2.y + p3.y;
        }
        for (; c < c_end; c++) {
            const uint2 p = partial[(size_t)c * count + j];
            sum += p.x; cnt += p.y;
        }
        const unsigned long long k = k1_finish(sum, cnt, n_points, ev_idx ? ev_idx[j] : j, dist_out);
        key = k < key ? k : key;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(key, off, 64);
        key = o < key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) wkey[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) key = wkey[w] < key ? wkey[w] : key;
        if (key != ~0ull) atomicMin(key_out, key);
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
    const bool tiled = sane && (cs->hs % 8 == 0) && !force_global;
    const int n_rb = cs->n_rb;
    int32_t *dist = want_dist ? cs->d_dist : nullptr;
    unsigned long long *key = (unsigned long long *)key_dst;
    const float bx = pose ? pose[0] : 0.f, by = pose ? pose[1] : 0.f, bth = pose ? pose[2] : 0.f;
    const int n_groups = sh_div_up(count, K1_WG);

    int budget_shared = tile_kb * 1024;
    if (budget_shared > K1_WG * K1_PF * 16) budget_shared = K1_WG * K1_PF * 16;      // what 4 prefetch vectors/lane can stage
    int budget_sub = budget_shared / 4;
    static const int sub_kb = env_int("SLAMHIP_K1_SUB_KB", 0);         // debugging: decouple the two budgets
    static const int no_shared = env_int("SLAMHIP_K1_NOSHARED", 0);
    if (sub_kb > 0) budget_sub = sub_kb * 1024;
    budget_sub &= ~15;
    if (budget_sub > K1_SUB * K1_PF * 16) budget_sub = K1_SUB * K1_PF * 16;
    const int budget_shared_eff = no_shared ? 0 : budget_shared;
    if (tiled) SH_TRY(ensure_plans(cs, (size_t)n_groups * n_rb * K1_PLAN_INTS + (size_t)n_groups * 4));
    int *tailp = tiled ? cs->d_plans + (size_t)n_groups * n_rb * K1_PLAN_INTS : nullptr;      // tail flag per sub-batch
    {
        sh_timer t(ctx, SLAMHIP_K_CS_PREP);
        int *plans = tiled ? cs->d_plans : nullptr;
#define K1_PREP(M) hipLaunchKernelGGL(k1_prep_plan<M>, dim3(n_groups), dim3(K1_WG), 0, ctx->stream, (const float *)cs->d_ev_off, \
                       bx, by, bth, cs->hscale, cs->d_pxcs, count, key, (const float2 *)cs->d_pts_sorted, (const int *)cs->d_rb_start, \
                       n_rb, cs->hs, budget_shared_eff, budget_sub, plans, tailp, cs->n_points)
        if (mode == 0) K1_PREP(0); else if (mode == 1) K1_PREP(1); else K1_PREP(2);
#undef K1_PREP
    }
    static const int dump = env_int("SLAMHIP_K1_DUMP", 0);
    if (dump && tiled) {                                           // debugging aid: histogram of plan kinds
        std::vector<int> h((size_t)n_groups * n_rb * K1_PLAN_INTS + (size_t)n_groups * 4);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(h.data(), cs->d_plans, sizeof(int) * h.size(), hipMemcpyDeviceToHost);
        int kinds[4] = { 0, 0, 0, 0 }, tails = 0;
        for (size_t i = 0; i < (size_t)n_groups * n_rb * 4; i++) kinds[h[i * 8 + 6] & 3]++;
        for (int i = 0; i < n_groups * 4; i++) tails += h[(size_t)n_groups * n_rb * K1_PLAN_INTS + i];
        fprintf(stderr, "[slamhip] K1 plans: count %d groups %d n_rb %d budgets %d/%d | own %d shared %d global %d skip %d | tail sub-batches %d\n",
                count, n_groups, n_rb, budget_shared_eff, budget_sub, kinds[0], kinds[1], kinds[2], kinds[3], tails);
        for (int b = 0; b < (n_rb < 3 ? n_rb : 3); b++)
            for (int sb = 0; sb < 4; sb++) {
                const int *r = &h[((size_t)0 * n_rb + b) * K1_PLAN_INTS + sb * 8];
                fprintf(stderr, "   g0 b%d sb%d: x0a %d y0 %d w8 %d h %d lds %d shift %d kind %d\n", b, sb, r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
            }
    }
    const int rblocks = sh_div_up(count, 256) < 256 ? sh_div_up(count, 256) : 256;
    if (tiled) {
        int bpc = (int)(((long long)n_groups * n_rb) / target_wgs);
        if (bpc < 1) bpc = 1;
        if (bpc > n_rb) bpc = n_rb;
        const int n_chunks = sh_div_up(n_rb, bpc);
        // tail kernel: 256-lane blocks, ~64 rays per chunk
        int bpc_t = sh_div_up(64 * n_rb, cs->n_points > 0 ? cs->n_points : 1);
        if (bpc_t < 1) bpc_t = 1;
        if (bpc_t > n_rb) bpc_t = n_rb;
        const int n_chunks_t = sh_div_up(n_rb, bpc_t);
        SH_TRY(ensure_partial(cs, (size_t)(n_chunks + n_chunks_t) * count));
        const size_t lds = (size_t)K1_PTS_BYTES + (size_t)(budget_shared > 4 * budget_sub ? budget_shared : 4 * budget_sub);
        {
            sh_timer t(ctx, SLAMHIP_K_CS_DISTANCE);
            dim3 grid(n_groups, n_chunks);
            if (verify)
                hipLaunchKernelGGL(k1_distance_tiled<true>, grid, dim3(K1_WG), lds, ctx->stream,
                                   cs->d_hole, cs->hs, cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, cs->d_plans,
                                   (uint2 *)cs->d_partial, cs->d_verify);
            else
                hipLaunchKernelGGL(k1_distance_tiled<false>, grid, dim3(K1_WG), lds, ctx->stream,
                                   cs->d_hole, cs->hs, cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, cs->d_plans,
                                   (uint2 *)cs->d_partial, cs->d_verify);
            hipLaunchKernelGGL(k1_distance_global<false>, dim3(sh_div_up(count, K1_THREADS), n_chunks_t), dim3(K1_THREADS), 0, ctx->stream,
                               cs->d_hole, cs->hs, cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc_t, cs->d_pxcs, count,
                               (uint2 *)cs->d_partial + (size_t)n_chunks * count, (const int *)tailp);
        }
        {
            sh_timer t(ctx, SLAMHIP_K_CS_REDUCE);
            hipLaunchKernelGGL(k1_reduce, dim3(rblocks), dim3(256), 0, ctx->stream, (const uint2 *)cs->d_partial, n_chunks, n_chunks_t,
                               (const int *)tailp, count, cs->n_points, cs->d_ev_idx, dist, key);
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
                                   cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, (uint2 *)cs->d_partial, (const int *)nullptr);
            else
                hipLaunchKernelGGL(k1_distance_global<true>, grid, dim3(K1_THREADS), 0, ctx->stream, cs->d_hole, cs->hs,
                                   cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, (uint2 *)cs->d_partial, (const int *)nullptr);
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
