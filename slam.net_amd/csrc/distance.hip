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
// One launch per search (k1_search_tiled).  A workgroup owns a "group" of 512 / 1024 / 2048 theta-consecutive
// candidates (by the size of the search) and a chunk = a range of the spatially sorted rays, cut into pieces at ray
// block boundaries:
//   prologue  (px,py,c,s) of its candidates (deterministic trigonometry), their min/max bounds, and -- by interval
//             arithmetic on the reference's very float operations (rounding is monotone, so the box is rigorous)
//             -- the pixel box every candidate's end points of a piece fall into; from the boxes a list of STEPS:
//             one HoleMap tile in LDS per piece when the box fits (SHARED), else the box, clipped to the map, cut
//             into row BANDs that are staged one after the other with a per-gather range test (theta tails, long
//             rays, boxes at the map border), else bounds-checked global gathers (GLOBAL);
//   steps     the tile is staged with coalesced 16-byte loads that were issued one step ahead (registers),
//             then ~14 VALU + 1 ds_read_u16 per point evaluation, no bounds test for SHARED;
//   epilogue  partial sums are added to per-candidate 64-bit accumulators (sum | in-map count | arrivals) with
//             agent-scope returning atomics; the lane whose add completes a candidate's arrival count holds its
//             total, finishes the distance and rests the word; the workgroup minima go through one atomic min, and
//             the workgroup that completes the finished-candidates counter publishes the overall arg-min.
// Inputs the fast path cannot take (NaN / huge coordinates, map sides that are not a multiple of 8) run the
// bounds-checked global-gather kernels (k1_prep_pxcs, k1_distance_global, k1_reduce).
#include "cs_internal.h"
#include "det_trig.h"
#include <stdlib.h>
#include <algorithm>

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
    // (sum * 1024) / R in 64 bits (:253).  The 64-bit integer division is a long software routine on this hardware and sits
    // on the critical path of the last arriver: take the quotient from one binary64 division (sum * 1024 < 2^50 is exact in
    // binary64, so the estimate is within one of the truth) and settle it with the exact remainder.
    uint64_t q = 0;
    if (cnt > 0) {
        const uint64_t num = sum * 1024ull, den = (uint64_t)n_points;
        q = (uint64_t)((double)num / (double)den);
        const int64_t rem = (int64_t)num - (int64_t)(q * den);
        if (rem < 0) q--; else if (rem >= (int64_t)den) q++;
    }
    const int32_t d = cnt > 0 ? (int32_t)q : INT32_MAX;
    if (dist_out) dist_out[flat] = d;
    return ((unsigned long long)(uint32_t)d << 32) | (uint32_t)flat;
}

// (px,py,c,s) of a candidate (:232-235).  MODE 1: search_pose + jitter (:635-637); MODE 2: pose.
template <int MODE, bool SMALL>
__device__ static inline float4 k1_candidate(const float c3[3], float bx, float by, float bth, float scale)
{
    float x, y, th;
    if (MODE == 1) { x = bx + c3[0]; y = by + c3[1]; th = bth + c3[2]; }
    else           { x = c3[0];      y = c3[1];      th = c3[2]; }
    float s, c;
    if (SMALL) sh_det_sincosf_small(th, &s, &c); else sh_det_sincosf(th, &s, &c);
    float4 q;
    q.x = x * scale + 0.5f;
    q.y = y * scale + 0.5f;
    q.z = c * scale;
    q.w = s * scale;
    return q;
}

// ---- fallback path: global gathers ---------------------------------------------------------------------------
// MODE 0: pxcs given; 1: search_pose + jitter; 2: poses.  Arms the arg-min key.
template <int MODE>
__global__ void __launch_bounds__(K1_THREADS)
k1_prep_pxcs(const float *__restrict__ src3, float bx, float by, float bth, float scale, float4 *__restrict__ pxcs,
             int count, unsigned long long *__restrict__ key)
{
    const int j = blockIdx.x * K1_THREADS + threadIdx.x;
    if (j == 0) *key = ~0ull;
    if (MODE != 0 && j < count) pxcs[j] = k1_candidate<MODE == 0 ? 1 : MODE, false>(src3 + 3 * (size_t)j, bx, by, bth, scale);
}

template <bool SAFE>
__global__ void __launch_bounds__(K1_THREADS)
k1_distance_global(const uint16_t *__restrict__ map, int S, const float2 *__restrict__ pts,
                   const int *__restrict__ rb_start, int n_rb, int blocks_per_chunk,
                   const float4 *__restrict__ pxcs, int count, uint2 *__restrict__ partial)
{
    const int chunk = blockIdx.y;
    const int j = blockIdx.x * K1_THREADS + threadIdx.x;
    const int b0 = chunk * blocks_per_chunk;
    const int b1 = b0 + blocks_per_chunk < n_rb ? b0 + blocks_per_chunk : n_rb;
    const float4 q = pxcs[j < count ? j : count - 1];
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int r = rb_start[b0];
    const int r1 = rb_start[b1];
    // eight gathers in flight per lane: every gather is an L2 / HBM round trip
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
    if (j < count) partial[(size_t)chunk * count + j] = make_uint2(s0 + s1 + s2 + s3, c0 + c1 + c2 + c3);
}

// per-candidate reduction of the chunk partials + arg-min: a block works on 64 candidates at a time with its 4
// waves splitting the partial rows and issues ONE atomicMin (same-address device atomics serialise at ~12 ns)
__global__ void __launch_bounds__(256)
k1_reduce(const uint2 *__restrict__ partial, int n_chunks, int count, int n_points, const int *__restrict__ ev_idx,
          int32_t *__restrict__ dist_out, unsigned long long *__restrict__ key_out)
{
    __shared__ uint32_t ssum[4][64], scnt[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long key = ~0ull;
    for (int base = blockIdx.x * 64; base < count; base += gridDim.x * 64) {
        const int j = base + lane;
        uint32_t sum = 0, cnt = 0;
        if (j < count)
            for (int c = w; c < n_chunks; c += 4) {
                const uint2 p = partial[(size_t)c * count + j];
                sum += p.x; cnt += p.y;
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

// ---- the tiled search kernel -----------------------------------------------------------------------------------
// Workgroup = one group of 1024 theta-consecutive candidates = LANES lanes x CPL candidates per lane (candidate
// k*LANES + t of the group lives in slot k of lane t).  Measured on MI355X: a wave costs ~0.15 us to launch and the
// waves of a CU start one after the other, so 16-wave workgroups (CPL 1) start over ~2.5 us, two of them per CU over
// ~5 us; with 4 waves (CPL 4: 2 waves / SIMD) the VALU starves, even with eight gathers in flight per lane.
// CPL 2 (512 lanes, 8 waves, 4 waves / SIMD) is the default.
#define K1_KIND_SHARED 1               // one tile holds the end points of the whole group: gathers need no test
#define K1_KIND_GLOBAL 2               // no tile (box wider than 512 px or more than K1_MAXBANDS bands): global gathers
#define K1_KIND_BAND 3                 // a row band of the group's box clipped to the map; gathers are range-tested
#define K1_MAXP 16                     // pieces (ray block fragments) per chunk
#define K1_MAXR 512                    // rays per chunk
#define K1_MAXBANDS 4
#define K1_MAXSTEPS (K1_MAXP * K1_MAXBANDS * 2)   // (a piece may be planned as two halves)
static_assert(K1_MAXBANDS == 4, "the step threads split t into (piece, band) with shifts");
#define K1_TABLE_G 64                  // groups with their own chunk count (the rest: one uniform count)
#define K1_TABLE_WGS 2048
#define K1_MAXCUT 596                  // entries of the cut table in the kernel arguments (ray ranges cut by cost: k1_balanced_cuts)
#define K1_ACC_INMAP 36                // accumulator fields: pixel sum below (R < 2^20 rays x 65535), workgroups with an in-map end point,
#define K1_ACC_ARRIVED 50              // workgroups arrived (chunks per group < 2^14)
#define K1_ZERO_OFS 0                  // dynamic LDS: a zero word (16 bytes), then the tile
#define K1_TILE_OFS 16
// step record: [0] x0a  [1] y0  [2] w8 (pitch, px)  [3] h  [4] shift = log2(lanes per row)  [5] kind
//              [6] piece: first ray (chunk-relative) | rays << 16   [7] -
// Staging geometry: a wave-wide 16-byte load covers 64 >> shift tile rows of (1 << shift) vectors each
// (lanes beyond w8/8 vectors idle), so no division is needed to map lanes to tile vectors.

struct k1_args {
    const uint16_t *map; int S;
    const float2 *pts;                 // spatially sorted rays
    const int4 *ray_blk;               // per ray: (first ray of its block, one past the last, block index, -)
    int n_rays;
    const float4 *pxcs;                // MODE 0
    const float *src3;                 // MODE 1: jitters, MODE 2: poses (evaluation order)
    float bx, by, bth, scale;
    int count, n_groups;
    int budget;                        // tile bytes
    int noden;                         // developer / test switch: tile addresses from the integer pixel coordinates (k1_tile_addr) everywhere
    int nopad;                         // tuning switch: tiles as wide as their box (no pitch padding against LDS bank conflicts)
    int nosplit;                       // tuning switch: a piece that does not fit one tile is never planned as two halves
    float band_stage;                  // cost of staging one band of a banded tile, in ray units (a large value: always band when it fits)
    unsigned long long *acc;           // [n_groups][K1_GROUP] per-candidate accumulators (sum | in-map | arrived); zero between launches
    unsigned long long *gmin;          // running minimum of the finished candidates' keys; all ones between launches
    unsigned *done;                    // finished candidates; zero between launches
    const int *ev_idx;
    int32_t *dist_out;
    unsigned long long *key_out;
    unsigned *verify;
    const float *offs_flat; float *best_pose;   // fused search + update: the winner's pose (theta normalised) for the map updates
    unsigned long long *sig; unsigned long long sig_val;   // sharded search: sig_val -> *sig (an HSA signal another stream waits for) once the key is out, or null
    unsigned *done_flag; unsigned done_val;     // blocking search: word 15 of the context's mailbox (pinned host memory, common.h) -- the launch's last act is the key into
                                                // words 0-1 and done_val into word 15 -- or null
    // Enqueue-only search (slamhip_cs_search_shard_enqueue): the result word is a slot of the handle's ring, all ones when the launch
    // starts -- the workgroups that finish candidates min their keys straight into it (no return, no count, no last finisher: the
    // END OF THE LAUNCH is the completion) -- and this launch leaves the NEXT slot all ones for the next one.  Null: key_out + the chain.
    unsigned long long *ring_slot, *ring_reset;
    // The PLAN of a search (k1_plan, launched beside the search on a stream of its own, round 6): every workgroup's step records
    // (bounds -> boxes -> tile steps: what its prologue used to work out before its first tile), left in device memory and STAMPED
    // with the plan's number: a workgroup that finds its record stamped plan_seq takes its steps from it, any other plans for
    // itself as before (a plan that is late, or was never made, costs time, never a result).  plan_rec: K1_PLAN_REC_WORDS words per
    // workgroup, every 8-word unit carrying its own stamp (see k1_plan).  Null: no plan.
    uint2 *plan_rec; uint32_t plan_seq;
    uint32_t *started; uint32_t launch_no;          // workgroup 0 stores launch_no here when the launch starts (pinned host word): everything before it in the stream has finished
    const uint32_t *scan_flag; uint32_t scan_seq;   // a launch that precedes its scan's tables (cs_search_and_update_prelaunched): wait here for scan_seq; bit 31 set: leave
    // launch layout: first the workgroups of the listed groups (expensive ones: more, smaller chunks), then the
    // groups [uni_g0, uni_g0 + uni_ng) with uni_nc chunks each, chunk-major (neighbouring groups work on the same
    // rays at the same time: their tiles overlap almost completely, L2 reuse)
    int n_tab_wgs, uni_g0, uni_ng, uni_nc;
    // ray ranges cut by cost (k1_balanced_cuts) instead of the equal-count formula: uni_cut -- range c of the uniform part is rays
    // [cut[c], cut[c + 1]); tab_cut > 0 -- listed workgroup w's range starts at cut[tab_cut + w] and ends where the workgroup of
    // the next range of its group starts (w + nbp), or with the scan
    int uni_cut, tab_cut;
    const float *grp_bounds;            // mode 1: per group {min dx, max dx, min dy, max dy, min dtheta, max dtheta, -, -} of the jitters, or null
    // One 8-byte record per dispatch position, every member at a multiple of its size.  (Separate byte / short / word
    // arrays indexed by the same position let the compiler form the scalar base "kernarg + p" for the byte loads and
    // reuse it for wider scalar loads, whose base the hardware aligns down to four bytes: seen with 64-bit members.)
    struct tab_rec {
        unsigned short group;
        unsigned short nbp;            // band parts: the chunks of a listed group come in sets of nbp that share a ray range and split its bands
        unsigned short first, nc;      // first workgroup of the position, workgroups
    } tab[K1_TABLE_G];
    alignas(4) unsigned char wg_pos[K1_TABLE_WGS];   // workgroup -> dispatch position (read a word at a time)
    alignas(4) unsigned short cut[K1_MAXCUT];       // (read a word at a time)
};
static_assert(sizeof(k1_args) <= 4096, "kernel arguments are limited to 4 KB");
static_assert(sizeof(k1_args::tab_rec) == 8 && offsetof(k1_args, tab) % 8 == 0, "table records are 8-byte aligned");
#define K1_PLAN_STEPS 15               // step records a workgroup's plan record holds (a workgroup with more plans for itself)
#define K1_PLAN_REC_WORDS 128          // a plan record: 16 x 8 words -- [0] header {stamp, steps (-1: too many), -, -, -, -, -, stamp}, [1 + i] step i {the step record's words 0 .. 6, stamp}

#ifdef K1_TIMES
// developer instrumentation (build with SLAMHIP_K1_TIMES=1): 100 MHz wall-clock stamps per workgroup and phase
__device__ unsigned long long g_k1_times[4096 * 16];
__device__ unsigned long long g_k1_wstart[4096 * 16];
__device__ unsigned long long g_k1_sub[4096 * 8];      // per workgroup: [0] staging (barrier, tile write, barrier) [1] prefetch issue [2] gather loops [3] steps [4] shader clocks of the compute phase
#define K1_SUB(k, v) { if (threadIdx.x == 0 && blockIdx.x < 4096) g_k1_sub[blockIdx.x * 8 + (k)] += (v); }
#define K1_STAMP(k) { if (threadIdx.x == 0 && blockIdx.x < 4096) g_k1_times[blockIdx.x * 16 + (k)] = wall_clock64(); }
#else
#define K1_STAMP(k) {}
#endif

// byte address of pixel (ix,iy) in the staged tile: iy*pitch2 + 2*ix + kofs, as exactly two VALU ops
__device__ static inline unsigned k1_tile_addr(int ix, int iy, int pitch2, int kofs)
{
    unsigned t, a;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(t) : "v"(ix), "s"(kofs));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a) : "v"(iy), "s"(pitch2), "v"(t));
    return a;
}
// The address of a tile step whose end points all lie in the tile, from the float coordinates: binary32 denormals are the integers 0 .. 2^23 - 1 in units of 2^-149 and
// their bit pattern IS that integer, the FMA unit handles them at full rate (the kernel runs with denormals on), and every term
// here -- 2 * ix, iy * pitch2, the offset c = tile base - 2 * first element, each partial sum -- is an integer of magnitude below
// 2^23, so the arithmetic is exact: truncation x 2 and two FMAs, no add, no mask.  two_d, pitch2_d, c_d are the integers 2,
// pitch2 and c as denormals (k1_den).  Valid while S * pitch2 + 2 * S + 2^18 < 2^23 (the caller checks; beyond, k1_tile_addr).
// (Six full-rate operations less per ray pair and candidate than truncation, FMA, the add of 2^22 + offset and the mantissa mask
// of the round-1 form; that one replaced two conversions and two integer multiply-adds at half rate.)
__device__ static inline float k1_den(int v) { return __uint_as_float(v < 0 ? (0x80000000u | (unsigned)(-v)) : (unsigned)v); }
__device__ static inline unsigned k1_tile_addr_d(float fx, float fy, float two_d, float pitch2_d, float c_d)
{
    const float g = fmaf(truncf(fy), pitch2_d, fmaf(truncf(fx), two_d, c_d));
    return __float_as_uint(g);
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

// two consecutive ray points with one LDS read (the list is 8-byte aligned: ds_read2_b64)
typedef float k1_f32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
__device__ static inline float4 k1_points2_lds(unsigned lds_addr)
{
    const k1_f32x4a8 v = *(__attribute__((address_space(3))) const k1_f32x4a8 *)(size_t)lds_addr;
    return make_float4(v.x, v.y, v.z, v.w);
}

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

// a / b for 0 <= a < 2^24, 1 <= b < 2^24: one v_rcp_f32 and an exact remainder fix-up instead of the ~35-instruction
// integer division sequence (the step records are made by a few lanes on the critical path of every workgroup)
__device__ static inline int k1_div(int a, int b)
{
    int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    const int r = a - q * b;
    if (r < 0) q--; else if (r >= b) q++;
    return q;
}

// wave-wide reduction (min / max) in six DPP steps, no LDS traffic: butterfly inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  The result is valid in lanes 48..63
// (row 3); lanes that a step's row mask excludes combine with themselves (min(x,x) = x).
template <int CTRL, int ROWS> __device__ static inline int k1_dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false); }
template <bool MAX> __device__ static inline int k1_wave_red(int x)
{
#define K1_MM(a, b) (MAX ? max(a, b) : min(a, b))
    x = K1_MM(x, (k1_dpp<0xB1, 0xf>(x)));       // quad_perm [1,0,3,2]
    x = K1_MM(x, (k1_dpp<0x4E, 0xf>(x)));       // quad_perm [2,3,0,1]
    x = K1_MM(x, (k1_dpp<0x124, 0xf>(x)));      // row_ror:4
    x = K1_MM(x, (k1_dpp<0x128, 0xf>(x)));      // row_ror:8
    x = K1_MM(x, (k1_dpp<0x142, 0xa>(x)));      // row_bcast:15 -> rows 1, 3
    x = K1_MM(x, (k1_dpp<0x143, 0xc>(x)));      // row_bcast:31 -> rows 2, 3
#undef K1_MM
    return x;
}
template <bool MAX> __device__ static inline float k1_wave_redf(float x)      // no NaNs (sane candidates)
{
#define K1_MM(a, b) (MAX ? fmaxf(a, b) : fminf(a, b))
    x = K1_MM(x, __int_as_float(k1_dpp<0xB1, 0xf>(__float_as_int(x))));
    x = K1_MM(x, __int_as_float(k1_dpp<0x4E, 0xf>(__float_as_int(x))));
    x = K1_MM(x, __int_as_float(k1_dpp<0x124, 0xf>(__float_as_int(x))));
    x = K1_MM(x, __int_as_float(k1_dpp<0x128, 0xf>(__float_as_int(x))));
    x = K1_MM(x, __int_as_float(k1_dpp<0x142, 0xa>(__float_as_int(x))));
    x = K1_MM(x, __int_as_float(k1_dpp<0x143, 0xc>(__float_as_int(x))));
#undef K1_MM
    return x;
}

// Which (candidate group, ray range) a workgroup of the search launch works on: the launch layout of k1_args (shared by the search
// kernel and the plan kernel, which must agree).
struct k1_wg_t { int g, nc, nbp, bp, rlo, rhi, chunk; };
__device__ __forceinline__ k1_wg_t k1_wg_decode(const k1_args &a, const int bid)
{
    int g, chunk, nc, nbp = 1;
    if ((int)bid < a.n_tab_wgs) {
        // (a byte load from the kernel arguments is a VECTOR load on this target -- a full memory round trip at the head of the
        // workgroup; the containing word is a scalar load)
        const unsigned pw = ((const unsigned *)a.wg_pos)[bid >> 2];
        const int p = (int)((pw >> ((bid & 3u) * 8u)) & 0xffu);
        const k1_args::tab_rec rec = a.tab[p];
        nc = rec.nc;
        chunk = bid - (int)rec.first;
        g = rec.group;
        nbp = rec.nbp;
    } else {
        // the uniform part runs chunk-major: the workgroups in flight work on the SAME rays for neighbouring theta groups,
        // whose tiles overlap almost completely (L2 reuse).  (Sending ray chunk c of every group to XCD c % 8 -- one
        // fabric fetch per tile instead of one per XCD -- was measured 1.2x to 2x SLOWER.)
        const int b = bid - a.n_tab_wgs;
        nc = a.uni_nc;
        chunk = b / (unsigned)a.uni_ng;
        g = a.uni_g0 + (b - chunk * a.uni_ng);
    }
    // the chunk = rays [rlo, rhi) of the sorted scan, cut into pieces at ray block boundaries (host: <= K1_MAXR
    // rays, <= K1_MAXP pieces)
    // (theta-tail groups: nbp workgroups share a ray range and take every nbp-th band of its banded tiles -- their
    // run time is the number of tile steps, which more ray ranges would not reduce)
    const int rc = chunk / nbp, bp = chunk - rc * nbp, nrc = nc / nbp;
    int rlo, rhi;
#define K1_CUT(i) ((int)((((const unsigned *)a.cut)[(i) >> 1] >> (((i) & 1) * 16)) & 0xffffu))
    if (a.uni_cut && (int)bid >= a.n_tab_wgs) {             // (uniform part: nbp = 1, rc = chunk)
        rlo = K1_CUT(chunk); rhi = K1_CUT(chunk + 1);
    } else if (a.tab_cut && (int)bid < a.n_tab_wgs) {
        const int i0 = a.tab_cut + (int)bid;
        rlo = K1_CUT(i0);
        rhi = rc + 1 < nrc ? K1_CUT(i0 + nbp) : a.n_rays;
    } else
#undef K1_CUT
    if (a.n_rays <= 46340) {                                       // (rc < nrc <= n_rays: the products fit 32 bits; the 64-bit division is a hundred scalar instructions)
        rlo = (int)(((unsigned)rc * (unsigned)a.n_rays) / (unsigned)nrc); rhi = (int)(((unsigned)(rc + 1) * (unsigned)a.n_rays) / (unsigned)nrc);
    } else { rlo = (int)(((long long)rc * a.n_rays) / nrc); rhi = (int)(((long long)(rc + 1) * a.n_rays) / nrc); }
    k1_wg_t W; W.g = g; W.nc = nc; W.nbp = nbp; W.bp = bp; W.rlo = rlo; W.rhi = rhi; W.chunk = chunk;
    return W;
}

// The bounds {pxmin, pxmax, pymin, pymax, cmin, cmax, smin, smax} of a group's candidates from the group's jitter bounds gb
// (k_gather_offsets) and the search pose of the launch; every lane computes them, lane 0 stores them.
__device__ __forceinline__ void k1_bounds_from_jitter(const k1_args &a, const float gb[6], float *bnd, const int lane)
{
    // px = (bx + dx) * scale + 0.5 and theta = btheta + dtheta are monotone in the jitter (every float operation
    // rounds monotonically), so the extreme jitters give the extreme px, py and theta of the group.  cos and sin
    // over [theta_lo, theta_hi] only need to be bounded, not reproduced: one reduction by a multiple of 2 pi in
    // double, the hardware sine / cosine at the end points, +-1 where the interval (with a margin) holds a
    // multiple of pi/2, and a pad of 1e-4 (a fifth of a pixel at 40 m) that covers the approximation.
    const float scale = a.scale;
    const float pxl = (a.bx + gb[0]) * scale + 0.5f, pxh = (a.bx + gb[1]) * scale + 0.5f;
    const float pyl = (a.by + gb[2]) * scale + 0.5f, pyh = (a.by + gb[3]) * scale + 0.5f;
    const float tlf = a.bth + gb[4], thf = a.bth + gb[5];
    const double n2 = rint((double)tlf * 0.15915494309189535) * 6.283185307179586;
    const float rl = (float)((double)tlf - n2), rh = (float)((double)thf - n2);      // rl in [-pi, pi], rh >= rl
    const float as = scale, pad = as * 1.0e-4f;                 // (scale = pixels per metre > 0)
    const float cl = __cosf(rl) * as, ch = __cosf(rh) * as, sl = __sinf(rl) * as, sh = __sinf(rh) * as;
    float clo = fminf(cl, ch) - pad, chi = fmaxf(cl, ch) + pad, slo = fminf(sl, sh) - pad, shi = fmaxf(sl, sh) + pad;
    const bool all = !(rh - rl < 6.2f);
    {   // multiples k of pi/2 in [rl - 1e-3, rh + 1e-3]: k_lo .. k_hi; residue m is among them iff (m - k_lo) mod 4 <= k_hi - k_lo
        // (this wave's arithmetic is on the critical path of the workgroup: a loop over the eleven possible k was 0.3 us)
        const int k_lo = (int)ceilf((rl - 1.0e-3f) * 0.636619772f), span = (int)floorf((rh + 1.0e-3f) * 0.636619772f) - k_lo;
        if (all || ((0 - k_lo) & 3) <= span) chi = as + pad;
        if (all || ((1 - k_lo) & 3) <= span) shi = as + pad;
        if (all || ((2 - k_lo) & 3) <= span) clo = -as - pad;
        if (all || ((3 - k_lo) & 3) <= span) slo = -as - pad;
    }
    const float4 qlo = make_float4(pxl, pyl, 0.f, 0.f), qhi = make_float4(pxh, pyh, 0.f, 0.f);
    if (lane == 0) {
        *(float4 *)&bnd[0] = make_float4(qlo.x, qhi.x, qlo.y, qhi.y);
        *(float4 *)&bnd[4] = make_float4(clo, chi, slo, shi);
    }
}

// developer experiments (SLAMHIP_K1_EXP=n at build time, WRONG RESULTS): parts of k1_search_tiled left out, so that the counters say
// what each costs (tools/k1_budget.sh) -- 1: the candidates' trigonometry (the jitters stand in for px, py, c, s), 2: the gather loops,
// 3: the tiles' staging (loads and LDS writes) as well, 4: the epilogue's accumulator adds and everything behind them
#ifndef K1_DMA0
#define K1_DMA0 0                      // 1: the first tile of a workgroup by LDS-DMA (see the steps; measured, no gain: DESIGN.md Appendix A)
#endif
#ifndef K1_EXP
#define K1_EXP 0
#endif
typedef unsigned int k1_u32x4 __attribute__((ext_vector_type(4)));   // a staging register quad (native vector: stays in VGPRs)

// LAT: the candidates of a lane share their heading, bit for bit (a heading lattice: slamhip_cs_generate_offsets_lattice) -- the
// four products c*X, s*Y, s*X, c*Y of a ray point are formed once per lane, from the lane's first candidate, and every candidate
// adds them to its own px, py in the reference's order (:240-241): the same floats as k1_coords, 2 (CPL = 2) or 3 (CPL = 4) of the
// ~12.75 VALU operations per candidate and ray less.
#ifndef K1_VGPR_CAP
#define K1_VGPR_CAP __attribute__((amdgpu_num_vgpr(104)))
#endif
template <int MODE, bool VERIFY, int CPL, int GROUP, bool LAT = false>
__global__ void __launch_bounds__(GROUP / CPL, GROUP / CPL / 64 / 2) K1_VGPR_CAP         // two workgroups per CU
k1_search_tiled(const k1_args a)
{
    constexpr int LANES = GROUP / CPL, NW = LANES / 64, PF = 65536 / (LANES * 16);      // PF staging vectors per lane: 64 KB per pass
    constexpr int RU = (K1_MAXR + LANES - 1) / LANES;                          // ray slots per lane in the prologue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;   // LDS byte address
    __shared__ __attribute__((aligned(16))) int stepbuf[K1_MAXSTEPS * 8];
    __shared__ __attribute__((aligned(16))) float2 cpts[K1_MAXR + 4];    // the chunk's ray points (+ slack: the point prefetch over-reads)
    __shared__ int2 pieces[K1_MAXP];                     // (first ray, rays), chunk-relative
    __shared__ __attribute__((aligned(16))) float wred[NW][8];
    __shared__ __attribute__((aligned(16))) float bnd[8];
    __shared__ unsigned long long wkey[NW];
    __shared__ unsigned wfin[NW];
    __shared__ int s_nsteps;
    __shared__ int s_plan_ok;
    const unsigned cpts_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)cpts;

    const uint16_t *__restrict__ map = a.map;
    const int S = a.S, count = a.count;
    const int t = threadIdx.x, lane = t & 63;
    int zv;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zv));                   // opaque zero (see k1_point_lds)
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);         // wave index in the workgroup
    const k1_wg_t W = k1_wg_decode(a, (int)blockIdx.x);
    const int g = W.g, nc = W.nc, nbp = W.nbp;
    if (a.ring_reset && blockIdx.x == 0 && t == 0)                 // (the slot of the NEXT ring launch; launches are ordered by the stream)
        __hip_atomic_store(a.ring_reset, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.started && blockIdx.x == 0 && t == 0)                    // (the host learns that everything before this launch in the stream has finished)
        __hip_atomic_store(a.started, a.launch_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    K1_STAMP(0)
#ifdef K1_TIMES
    if (t == 0 && blockIdx.x < 4096) { for (int k = 0; k < 8; k++) g_k1_sub[blockIdx.x * 8 + k] = 0; }
    if (t == 0 && blockIdx.x < 4096) { for (int k = 10; k < 16; k++) g_k1_times[blockIdx.x * 16 + k] = 0; g_k1_times[blockIdx.x * 16 + 14] = (unsigned long long)g; g_k1_times[blockIdx.x * 16 + 15] = (unsigned long long)nc; }
    if (lane == 0 && blockIdx.x < 4096) g_k1_wstart[blockIdx.x * 16 + (wv & 15)] = wall_clock64();
    if (t == 0 && blockIdx.x < 4096) {                                     // which CU runs this workgroup
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_k1_wstart[blockIdx.x * 16 + 15] = ((unsigned long long)(xcc & 15u) << 16) | ((hw >> 8) & 0xffu);   // xcc | se,sh,cu
    }
#endif
    // the chunk = rays [rlo, rhi) of the sorted scan, cut into pieces at ray block boundaries (host: <= K1_MAXR
    // rays, <= K1_MAXP pieces): k1_wg_decode
    const int bp = W.bp, rlo = W.rlo, rhi = W.rhi;
    const int nrays = rhi - rlo;
    if (a.scan_flag) {
        // The launch was put into the stream before its scan's tables existed (coreslam.hip, cs_search_and_update_prelaunched): the
        // host stores the scan's number behind the tables (through the BAR, fenced), and nothing of the scan is read before the
        // number is seen -- by every wavefront for itself.  The word lives in fine-grained memory and is loaded at system scope: never
        // from a cache.  The tables need no invalidate of their own -- the launch started with clean caches and no wavefront touches
        // a line of the scan before it has seen the number (an ACQUIRE here is a buffer_inv in every one of the launch's 4096
        // wavefronts as they start, one after the other, each emptying the L2 under the workgroups already at work: the search
        // took 50 us instead of 23).  The loads behind the loop depend on its exit.  Normally the word is there long before the
        // launch starts (it starts behind the previous scan's map update): one load.  Bit 31: the host found that the launch's assumptions do not hold for this scan -- leave without a trace (the
        // result word stays rested: the map update that decodes it falls back to the search pose, holemap.hip; the host searches
        // again).  A host that never answers: the same after ~10 s, with the self-check counter raised.
        // ONE wavefront watches the word and the verdict reaches the others through the LDS behind a barrier: a workgroup goes on or
        // leaves as a whole (wavefronts that gave up one by one would leave a workgroup half gone -- accumulators not at rest), and
        // the word is polled by an eighth of the wavefronts.
        uint32_t v = 0;
        if (wv == 0) {
            int spins = 0;
            for (;;) {
                v = __hip_atomic_load(a.scan_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((v & 0x7fffffffu) == a.scan_seq) break;
                if (++spins > (1 << 23)) { if (t == 0) atomicAdd(a.verify, 1u); v = 0x80000000u; break; }
                __builtin_amdgcn_s_sleep(32);
            }
            if (blockIdx.x == 0 && t == 0) { ((uint32_t *)a.scan_flag)[8] = (uint32_t)spins; ((uint32_t *)a.scan_flag)[9] += (uint32_t)spins; }   // (developer aid: how long the first workgroup waited, SLAMHIP_FUSED_TIMES)
            if (lane == 0) s_plan_ok = (int)v;
        }
        __syncthreads();
        v = (uint32_t)s_plan_ok;
        __syncthreads();                                           // (the word is used again below: the plan's verdict)
        if (__builtin_amdgcn_readfirstlane(v) & 0x80000000u) return;
    }
    // (the two block numbers are uniform, and the compiler would wait for them -- to move them into SGPRs -- before it issues
    // the loads below: a memory round trip at the head of every workgroup.  Loaded through an address that looks
    // lane-dependent they stay in VGPRs and are waited for where they are used)
    const int blk_first = a.ray_blk[rlo + zv].z;
    const int blk_last_v = a.ray_blk[rhi - 1 + zv].z;

    // ---- prologue: rays, candidates, bounds, boxes, steps ----------------------------------------------------------
    // every global load first, then the arithmetic -- and the unconditional loads before the masked ones: the merge after a
    // masked load makes the compiler wait for it, which would hold back whatever is issued after it
    // Search mode: the bounds of the group's (px, py, c, s) follow from the group's jitter bounds (left by
    // k_gather_offsets) and the search pose, so nothing before the first tile depends on the candidates: their
    // trigonometry runs later, under the tile's global loads.
    const bool pre = MODE == 1 && a.grp_bounds != nullptr;
    float gb[6];
    if (pre) {
#pragma unroll
        for (int k = 0; k < 6; k++) gb[k] = a.grp_bounds[8 * (size_t)g + k];
    }
    // The plan (k1_args): the workgroup's record -- wave 0, one 8-byte load per lane -- comes with the first round trip.
    const bool planned = MODE == 1 && !LAT && pre && a.plan_rec != nullptr;
    uint2 prw = make_uint2(0u, 0u);
    if (planned && wv == 0) prw = a.plan_rec[(size_t)blockIdx.x * (K1_PLAN_REC_WORDS / 2) + lane];
    float4 q[CPL];
    float c3[CPL][3];
#pragma unroll
    for (int k = 0; k < CPL; k++) {
        const int j = g * GROUP + k * LANES + t;
        const int jc = j < count ? j : count - 1;
        if (MODE == 0) q[k] = a.pxcs[jc];
        else { c3[k][0] = a.src3[3 * jc]; c3[k][1] = a.src3[3 * jc + 1]; c3[k][2] = a.src3[3 * jc + 2]; }
    }
    int4 rinfo[RU];
    float2 rpt[RU];
#pragma unroll
    for (int u = 0; u < RU; u++) {
        const int i = u * LANES + t;
        const int r = rlo + (i < nrays ? i : 0);
        rinfo[u] = make_int4(0, 0, 0, 0); rpt[u] = make_float2(0.f, 0.f);
        if (i < nrays) { rinfo[u] = a.ray_blk[r]; rpt[u] = a.pts[r]; }   // (lanes beyond the chunk issue nothing)
    }
    if (t == 0) { s_nsteps = 0; s_plan_ok = 0; *(unsigned *)(smem + K1_ZERO_OFS) = 0u; }
    if (MODE != 0 && !pre) {
#pragma unroll
        for (int k = 0; k < CPL; k++) q[k] = k1_candidate<MODE == 0 ? 1 : MODE, true>(c3[k], a.bx, a.by, a.bth, a.scale);
    }
#pragma unroll
    for (int u = 0; u < RU; u++) {
        const int i = u * LANES + t;
        if (i < nrays) {
            cpts[i] = rpt[u];
            if (i == 0 || rinfo[u].x == rlo + i)
                pieces[rinfo[u].z - blk_first] = make_int2(i, (rinfo[u].y < rhi ? rinfo[u].y : rhi) - (rlo + i));
        }
    }
    bool have_plan = false;
    if (planned) {
        if (wv == 0) {
            // the record is good when every one of its 8-word units carries the plan's stamp (and the header says the steps fit)
            const bool okl = ((lane & 3) != 3 || prw.y == a.plan_seq) && (lane != 0 || prw.x == a.plan_seq);
            const int ns = __builtin_amdgcn_readfirstlane((int)prw.y);             // (lane 0: the header's step count)
            const bool ok = __builtin_amdgcn_ballot_w64(okl) == ~0ull && ns >= 0 && ns <= K1_PLAN_STEPS;
            if (lane >= 4) *(int2 *)&stepbuf[2 * lane - 8] = make_int2((int)prw.x, (int)prw.y);
            if (lane == 0 && ok) { s_plan_ok = 1; s_nsteps = ns; }
        }
        __syncthreads();
        have_plan = __builtin_amdgcn_readfirstlane(s_plan_ok) != 0;
        if (VERIFY && t == 0) atomicAdd(a.verify + (have_plan ? 5 : 6), 1u);     // (self-check builds count the workgroups that found their record / did not)
    }
    K1_STAMP(1)
    if (!have_plan) {
    if (pre) {
        if (wv == 0) {
            k1_bounds_from_jitter(a, gb, bnd, lane);
        }
        // (Round 5, measured and rejected: the OTHER wavefronts making their candidates here, where they wait for the bounds, instead of
        // under the first tile's loads -- the K1_EXP = 1 build says the trigonometry, 107 binary64-heavy instructions of a wavefront's
        // 1472 that all sixteen resident wavefronts run in the same phase, costs the launch 1.7 us -- 17.2 - 17.4 us either way.)
        __syncthreads();
        K1_STAMP(2)
    } else
    {   // min / max of (px, py, c, s) over the group: the lane's candidates, the wave (DPP), the waves (LDS)
        float lo[4] = { q[0].x, q[0].y, q[0].z, q[0].w }, hi[4] = { q[0].x, q[0].y, q[0].z, q[0].w };
#pragma unroll
        for (int k = 1; k < CPL; k++) {
            lo[0] = fminf(lo[0], q[k].x); hi[0] = fmaxf(hi[0], q[k].x); lo[1] = fminf(lo[1], q[k].y); hi[1] = fmaxf(hi[1], q[k].y);
            lo[2] = fminf(lo[2], q[k].z); hi[2] = fmaxf(hi[2], q[k].z); lo[3] = fminf(lo[3], q[k].w); hi[3] = fmaxf(hi[3], q[k].w);
        }
        const float m0 = k1_wave_redf<false>(lo[0]), m1 = k1_wave_redf<true>(hi[0]);
        const float m2 = k1_wave_redf<false>(lo[1]), m3 = k1_wave_redf<true>(hi[1]);
        const float m4 = k1_wave_redf<false>(lo[2]), m5 = k1_wave_redf<true>(hi[2]);
        const float m6 = k1_wave_redf<false>(lo[3]), m7 = k1_wave_redf<true>(hi[3]);
        if (lane == 63) {
            *(float4 *)&wred[wv][0] = make_float4(m0, m1, m2, m3);
            *(float4 *)&wred[wv][4] = make_float4(m4, m5, m6, m7);
        }
        __syncthreads();
        K1_STAMP(2)
        if (t < 8) {
            float v[NW];
#pragma unroll
            for (int w = 0; w < NW; w++) v[w] = wred[w][t];
            float x = v[0];
#pragma unroll
            for (int w = 1; w < NW; w++) x = (t & 1) ? fmaxf(x, v[w]) : fminf(x, v[w]);
            bnd[t] = x;
        }
        __syncthreads();
    }
    K1_STAMP(3)
    const int npieces = __builtin_amdgcn_readfirstlane(blk_last_v - blk_first + 1);
    // the pieces' boxes and tile steps (k1_pieces.inc: the text is shared with the plan kernel -- as a FUNCTION the same statements
    // cost the search kernel 24 more vector registers and, at four candidates per lane, spills)
#define K1_PIECE_STAMP K1_STAMP(4)
#define K1_STEP_SLOT(i) (i)
#define K1_PIECE_RAY(i) (i)
#include "k1_pieces.inc"
#undef K1_PIECE_RAY
#undef K1_STEP_SLOT
#undef K1_PIECE_STAMP
    __syncthreads();
    }
#ifdef K1_TIMES
    else if (t == 0 && blockIdx.x < 4096) { const unsigned long long now_ = wall_clock64(); for (int k = 2; k <= 4; k++) g_k1_times[blockIdx.x * 16 + k] = now_; }
#endif
    const int nsteps = s_nsteps;
    K1_STAMP(5)

    // ---- steps ---------------------------------------------------------------------------------------------------
    // Tile staging through registers: the global loads of the NEXT step's tile are issued before the current
    // step is consumed and stay in flight meanwhile.  The lanes that have no tile vector are masked off: a masked lane
    // costs the address unit nothing, where a dummy load -- even of one address for the whole wave -- does (measured
    // 31.2 -> 29.3 us at 16 384 candidates, 164 -> 157 us at 262 144).  With loads under an exec mask the compiler's vmcnt
    // waits become vmcnt(0), and whenever an address temporary of the prefetch inside the step loop shared a register with a
    // load it could not prove complete, it put one between every two loads -- slower than no masking at all (33.5 us); which
    // registers are shared changes with unrelated edits.  The staging registers are therefore claimed once per step, outside
    // any branch, before the tile is written to LDS (see there).
    uint32_t sum[CPL], cnt[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) { sum[k] = 0; cnt[k] = 0; }
    if (LAT && !(MODE != 0 && a.grp_bounds != nullptr)) {          // (with the bounds the candidates are made under the first tile's loads: checked there)
#pragma unroll
        for (int k = 1; k < CPL; k++)
            if (g * GROUP + k * LANES + t < count && (__float_as_uint(q[k].z) != __float_as_uint(q[0].z) || __float_as_uint(q[k].w) != __float_as_uint(q[0].w)))
                atomicAdd(a.verify, 1u);                            // a lattice whose lane does not share its heading: slamhip_cs_selfcheck_failures
    }
    uint32_t cnt_all = 0;                                          // rays of the unchecked steps: in the map for every candidate
    if (nsteps > 0) {
        k1_u32x4 R[PF];
        int dst[PF];
#define K1_PREFETCH(step, enable)                                                                   \
        {                                                                                           \
            /* 32-bit byte offsets from the map base (a map holds less than 2^32 bytes): a load's address is the scalar   */ \
            /* base plus one VGPR offset, and from one load to the next the offsets advance by uniform strides (64-bit      */ \
            /* per-lane products were two quarter-rate multiplies per load).                                                */ \
            const int4 pa_ = *(const int4 *)&stepbuf[(step) * 8];                                   \
            const int shift_ = __builtin_amdgcn_readfirstlane(stepbuf[(step) * 8 + 4]);             \
            const int w8_ = __builtin_amdgcn_readfirstlane(pa_.z), h_ = (enable) ? __builtin_amdgcn_readfirstlane(pa_.w) : 0; \
            const int rpi = 64 >> shift_;                       /* tile rows per wave-wide load */   \
            const int srow = lane >> shift_, scol = lane & ((1 << shift_) - 1);                     \
            const bool colok = scol < (w8_ >> 3);                                                   \
            const int cc = colok ? scol : 0;                                                        \
            const unsigned S2_ = (unsigned)S * 2u;                                                  \
            const unsigned gofs = ((unsigned)__builtin_amdgcn_readfirstlane(pa_.y) * (unsigned)S + (unsigned)__builtin_amdgcn_readfirstlane(pa_.x)) * 2u; \
            const int pitchb = w8_ << 1;                                                            \
            const int row0 = wv * rpi + srow, rstep = NW * rpi;                                     \
            unsigned vofs = gofs + (unsigned)row0 * S2_ + ((unsigned)cc << 4);                      \
            int ldsd = K1_TILE_OFS + (cc << 4) + row0 * pitchb;                                     \
            const unsigned vstep = (unsigned)rstep * S2_;                                           \
            const int lstep = rstep * pitchb;                                                       \
            _Pragma("unroll") for (int k_ = 0; k_ < PF; k_++) {                                     \
                const bool live = colok & (row0 + k_ * rstep < h_);                                 \
                if (live && K1_EXP != 3) R[k_] = *(const k1_u32x4 *)((const char *)map + vofs);   /* (see above) */ \
                dst[k_] = live ? ldsd : -1;                                                         \
                vofs += vstep; ldsd += lstep;                                                       \
            }                                                                                       \
        }
#if K1_DMA0
        {
            // The FIRST tile by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes straight into 1 KiB of LDS, the base in M0):
            // nobody reads a previous tile, so the tile area itself can be the landing zone -- no staging registers, no
            // ds_write pass and one barrier less in front of the first gather.  A wave-wide load lands contiguously, so the tile's
            // 16-byte units are dealt in row-major order (unit e = row * ncols + col; the pitch stays w8 * 2).  Later tiles keep
            // the register form: their loads are in flight while the current tile is read, and the registers ARE the second buffer.
            const int4 pa_ = *(const int4 *)&stepbuf[0];
            const int w8_ = __builtin_amdgcn_readfirstlane(pa_.z), h_ = __builtin_amdgcn_readfirstlane(pa_.w);
            const int ncols = w8_ >> 3, nunits = ncols * h_;
            const unsigned S2_ = (unsigned)S * 2u;
            const unsigned gofs = ((unsigned)__builtin_amdgcn_readfirstlane(pa_.y) * (unsigned)S + (unsigned)__builtin_amdgcn_readfirstlane(pa_.x)) * 2u;
            const float rn = 1.0f / (float)ncols;
#pragma unroll
            for (int k_ = 0; k_ < PF; k_++) {
                const int e0 = (k_ * NW + wv) * 64;
                if (e0 < nunits && K1_EXP != 3) {
                    const int e = e0 + lane;
                    const int row = (int)(((float)e + 0.5f) * rn);      // e < 4096, ncols <= 64: (e + 0.5) / ncols is at least 1/128 away from an integer
                    const int col = e - row * ncols;
                    if (e < nunits)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)map + (gofs + (unsigned)row * S2_ + ((unsigned)col << 4))),
                                                         (__attribute__((address_space(3))) void *)(smem + K1_TILE_OFS + (e0 << 4)), 16, 0, 0);
                }
            }
#pragma unroll
            for (int k_ = 0; k_ < PF; k_++) dst[k_] = -1;
        }
#else
        K1_PREFETCH(0, true)
#endif
        if (MODE != 0 && pre) {                                    // (the first tile's loads are in flight)
            {
#pragma unroll
            for (int k = 0; k < CPL; k++) { if (K1_EXP == 1) q[k] = make_float4(a.bx * a.scale + c3[k][0], a.by * a.scale + c3[k][1], a.scale, c3[k][2]); else q[k] = k1_candidate<MODE == 0 ? 1 : MODE, true>(c3[k], a.bx, a.by, a.bth, a.scale); }
            }
            if (LAT) {
#pragma unroll
                for (int k = 1; k < CPL; k++)
                    if (g * GROUP + k * LANES + t < count && (__float_as_uint(q[k].z) != __float_as_uint(q[0].z) || __float_as_uint(q[k].w) != __float_as_uint(q[0].w)))
                        atomicAdd(a.verify, 1u);                    // a lattice whose lane does not share its heading: slamhip_cs_selfcheck_failures
            }
        }
#ifdef K1_TIMES
        const unsigned long long sclk0 = clock64();
#endif
        for (int s = 0; s < nsteps; s++) {
#ifdef K1_TIMES
            const unsigned long long ts0 = wall_clock64();
#endif
            if (K1_DMA0 && s == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wavefront's part of the first tile has landed
            else {
            __syncthreads();                                       // the previous tile is no longer read
            // (every staging register is claimed here, outside any branch: the loads that filled them sit under exec masks,
            // where the compiler cannot count them, and without this it guards the NEXT prefetch's address temporaries --
            // which share these registers -- with a vmcnt(0) between every two loads: the prefetch serialises, +4 us per launch)
#pragma unroll
            for (int k = 0; k < PF; k++) asm volatile("" : "+v"(R[k]));
#pragma unroll
            for (int k = 0; k < PF; k++) if (dst[k] >= 0 && K1_EXP != 3) *(k1_u32x4 *)(smem + dst[k]) = R[k];
            }
            __syncthreads();
            if (s == 0) K1_STAMP(6)
#ifdef K1_TIMES
            const unsigned long long ts1 = wall_clock64();
            if (s > 0) K1_SUB(0, ts1 - ts0)
#endif
            const int4 ca = *(const int4 *)&stepbuf[s * 8], cb = *(const int4 *)&stepbuf[s * 8 + 4];
            {   // issue the next step's loads now; nothing in the compute loops below waits on vector memory.  (After the last step
                // there is no next tile: every lane is idle and every load is skipped -- the last tile used to be fetched again.)
                const int sn = s + 1 < nsteps ? s + 1 : s;
                K1_PREFETCH(sn, s + 1 < nsteps)
            }
#ifdef K1_TIMES
            const unsigned long long ts2 = wall_clock64();
            K1_SUB(1, ts2 - ts1) K1_SUB(3, 1ull)
#endif
            const int kind = __builtin_amdgcn_readfirstlane(cb.y);             // (the record is uniform: keep it in SGPRs)
            const int nr = __builtin_amdgcn_readfirstlane(cb.z) >> 16;         // <= CS_RB_MAX = 64
#ifdef K1_TIMES
            if (t == 0 && blockIdx.x < 4096) g_k1_times[blockIdx.x * 16 + 10 + kind] += (unsigned long long)nr;
            if (t == 0 && blockIdx.x < 4096 && s < 3)            // (the first three steps' shapes: kind | tile width << 4 | tile rows << 16 | rays << 32)
                g_k1_sub[blockIdx.x * 8 + 5 + s] = (unsigned long long)kind | ((unsigned long long)__builtin_amdgcn_readfirstlane(ca.z) << 4) | ((unsigned long long)__builtin_amdgcn_readfirstlane(ca.w) << 16) | ((unsigned long long)nr << 32);
#endif
            const unsigned pbase = cpts_lds + (unsigned)zv + (unsigned)(__builtin_amdgcn_readfirstlane(cb.z) & 0xffff) * 8u;
            const int x0a = __builtin_amdgcn_readfirstlane(ca.x), y0 = __builtin_amdgcn_readfirstlane(ca.y);
            const int w8 = __builtin_amdgcn_readfirstlane(ca.z), h = __builtin_amdgcn_readfirstlane(ca.w);
            const int pitch2 = w8 << 1;
            const int kofs = (int)smem_lds + K1_TILE_OFS - ((y0 * w8 + x0a) << 1);
            const unsigned zaddr = smem_lds + K1_ZERO_OFS + (unsigned)zv;
            // (float form of the tile address, k1_tile_addr_d; its constants live in VGPRs -- an SGPR operand costs a VALU operation
            // 1.6x the issue time)
            const bool daddr = !a.noden && (unsigned)S * (unsigned)pitch2 + 2u * (unsigned)S + (1u << 18) < (1u << 23);   // k1_tile_addr_d is exact
            float two_v, p2d_v, cd_v;
            {
                const float two_s = k1_den(2), p2d_s = k1_den(pitch2), cd_s = k1_den(kofs);
                asm volatile("v_mov_b32 %0, %1" : "=v"(two_v) : "s"(two_s));
                asm volatile("v_mov_b32 %0, %1" : "=v"(p2d_v) : "s"(p2d_s));
                asm volatile("v_mov_b32 %0, %1" : "=v"(cd_v) : "s"(cd_s));
            }
            if (K1_EXP == 2 || K1_EXP == 3) { sum[0] += (uint32_t)nr; continue; }
            if (kind != K1_KIND_GLOBAL) {
                // SHARED: every end point of every candidate lies in the tile (the box is rigorous).  BAND: the band
                // holds rows [y0, y0+h) x columns [x0a, x0a+w8) of the map; an end point outside it (another band's,
                // or outside the map) reads the zero word instead, and every in-map end point is in exactly one band.
                // Software pipeline: the gathers of a ray pair are issued together and consumed one iteration later,
                // under the address arithmetic of the next pair.  LDS returns in order and the compiler's wait for the
                // prefetched points is the minimum over all paths into the loop of the number of younger LDS
                // operations: the pipeline is therefore primed with loads of the zero word, not with constants.
                const bool checked = kind == K1_KIND_BAND;
                uint32_t va[CPL], vb[CPL];
                int r = 0;
                float4 pn = k1_points2_lds(pbase);                                      // (cpts has room for the over-read)
                float2 pa_n = make_float2(pn.x, pn.y), pb_n = make_float2(pn.z, pn.w);
#pragma unroll
                for (int k = 0; k < CPL; k++) { va[k] = k1_lds_load(zaddr); vb[k] = k1_lds_load(zaddr); }
                __builtin_amdgcn_sched_barrier(0);
                if (!checked && daddr) {
                    // two ray pairs per iteration, the point registers of the pairs taking turns (a rotation of one set costs four
                    // moves per pair); the points of two rays come with one 16-byte LDS read
#define K1_DPAIR(P, NEXTREAD)                                                                       \
                    {                                                                               \
                        unsigned ada[CPL], adb[CPL];                                                \
                        float cxa_ = 0.f, sya_ = 0.f, sxa_ = 0.f, cya_ = 0.f, cxb_ = 0.f, syb_ = 0.f, sxb_ = 0.f, cyb_ = 0.f; \
                        if (LAT) {                                                                  \
                            cxa_ = q[0].z * (P).x; sya_ = q[0].w * (P).y; sxa_ = q[0].w * (P).x; cya_ = q[0].z * (P).y; \
                            cxb_ = q[0].z * (P).z; syb_ = q[0].w * (P).w; sxb_ = q[0].w * (P).z; cyb_ = q[0].z * (P).w; \
                        }                                                                           \
                        _Pragma("unroll") for (int k = 0; k < CPL; k++) {                           \
                            float fxa, fya, fxb, fyb;                                               \
                            if (LAT) {                                                              \
                                fxa = q[k].x + cxa_; fxa = fxa - sya_; fya = q[k].y + sxa_; fya = fya + cya_; \
                                fxb = q[k].x + cxb_; fxb = fxb - syb_; fyb = q[k].y + sxb_; fyb = fyb + cyb_; \
                            } else {                                                                \
                                k1_coords(q[k], make_float2((P).x, (P).y), fxa, fya);               \
                                k1_coords(q[k], make_float2((P).z, (P).w), fxb, fyb);               \
                            }                                                                       \
                            ada[k] = k1_tile_addr_d(fxa, fya, two_v, p2d_v, cd_v);                  \
                            adb[k] = k1_tile_addr_d(fxb, fyb, two_v, p2d_v, cd_v);                  \
                            if (VERIFY) {                                                           \
                                const int ixa = (int)fxa, iya = (int)fya, ixb = (int)fxb, iyb = (int)fyb; \
                                if (ixa < x0a || ixa >= x0a + w8 || iya < y0 || iya >= y0 + h ||    \
                                    ixb < x0a || ixb >= x0a + w8 || iyb < y0 || iyb >= y0 + h) atomicAdd(a.verify, 1u); \
                                else if (ada[k] != k1_tile_addr(ixa, iya, pitch2, kofs) || adb[k] != k1_tile_addr(ixb, iyb, pitch2, kofs) || \
                                         map[(size_t)iya * S + ixa] != *(const uint16_t *)(smem + (ada[k] - smem_lds)) || \
                                         map[(size_t)iyb * S + ixb] != *(const uint16_t *)(smem + (adb[k] - smem_lds))) \
                                    atomicAdd(a.verify, 1u);               /* the staged tile must equal the map */ \
                            }                                                                       \
                        }                                                                           \
                        __builtin_amdgcn_sched_barrier(0);                                          \
                        NEXTREAD;     /* (after the arithmetic: issued at the top, the wait for P would also wait for the gathers before it) */ \
                        _Pragma("unroll") for (int k = 0; k < CPL; k++) sum[k] += va[k] + vb[k];    \
                        _Pragma("unroll") for (int k = 0; k < CPL; k++) { va[k] = k1_lds_load(ada[k]); vb[k] = k1_lds_load(adb[k]); } \
                        __builtin_amdgcn_sched_barrier(0);                                          \
                    }
                    for (; r + 3 < nr; r += 4) {
                        const float4 p0 = pn;
                        float4 p1;
                        K1_DPAIR(p0, p1 = k1_points2_lds(pbase + r * 8 + 16))
                        K1_DPAIR(p1, pn = k1_points2_lds(pbase + r * 8 + 32))
                    }
                    if (r + 1 < nr) {
                        const float4 p0 = pn;
                        K1_DPAIR(p0, pn = k1_points2_lds(pbase + r * 8 + 16))
                        r += 2;
                    }
#undef K1_DPAIR
                    pa_n = make_float2(pn.x, pn.y);
                    cnt_all += (uint32_t)nr;
                } else if (!checked) {
                    for (; r + 1 < nr; r += 2) {
                        const float2 pa = pa_n, pb = pb_n;
                        pa_n = k1_point_lds(pbase + r * 8 + 16); pb_n = k1_point_lds(pbase + r * 8 + 24);
                        unsigned ada[CPL], adb[CPL];
#pragma unroll
                        for (int k = 0; k < CPL; k++) {
                            float fxa, fya, fxb, fyb;
                            k1_coords(q[k], pa, fxa, fya);
                            k1_coords(q[k], pb, fxb, fyb);
                            const int ixa = (int)fxa, iya = (int)fya, ixb = (int)fxb, iyb = (int)fyb;
                            ada[k] = k1_tile_addr(ixa, iya, pitch2, kofs);
                            adb[k] = k1_tile_addr(ixb, iyb, pitch2, kofs);
                            if (VERIFY) {
                                if (ixa < x0a || ixa >= x0a + w8 || iya < y0 || iya >= y0 + h ||
                                    ixb < x0a || ixb >= x0a + w8 || iyb < y0 || iyb >= y0 + h) atomicAdd(a.verify, 1u);
                                else if (map[(size_t)iya * S + ixa] != *(const uint16_t *)(smem + (ada[k] - smem_lds)) ||
                                         map[(size_t)iyb * S + ixb] != *(const uint16_t *)(smem + (adb[k] - smem_lds)))
                                    atomicAdd(a.verify, 1u);               // the staged tile must equal the map
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < CPL; k++) sum[k] += va[k] + vb[k];
#pragma unroll
                        for (int k = 0; k < CPL; k++) { va[k] = k1_lds_load(ada[k]); vb[k] = k1_lds_load(adb[k]); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    cnt_all += (uint32_t)nr;
                } else {
                    for (; r + 1 < nr; r += 2) {
                        const float2 pa = pa_n, pb = pb_n;
                        pa_n = k1_point_lds(pbase + r * 8 + 16); pb_n = k1_point_lds(pbase + r * 8 + 24);
                        unsigned ada[CPL], adb[CPL];
#pragma unroll
                        for (int k = 0; k < CPL; k++) {
                            float fxa, fya, fxb, fyb;
                            k1_coords(q[k], pa, fxa, fya);
                            k1_coords(q[k], pb, fxb, fyb);
                            const int ixa = (int)fxa, iya = (int)fya, ixb = (int)fxb, iyb = (int)fyb;
                            const bool oka = ((unsigned)(ixa - x0a) < (unsigned)w8) & ((unsigned)(iya - y0) < (unsigned)h);
                            const bool okb = ((unsigned)(ixb - x0a) < (unsigned)w8) & ((unsigned)(iyb - y0) < (unsigned)h);
                            const unsigned ta = k1_tile_addr(ixa, iya, pitch2, kofs), tb = k1_tile_addr(ixb, iyb, pitch2, kofs);
                            if (VERIFY) {
                                if (oka && map[(size_t)iya * S + ixa] != *(const uint16_t *)(smem + (ta - smem_lds))) atomicAdd(a.verify, 1u);
                                if (okb && map[(size_t)iyb * S + ixb] != *(const uint16_t *)(smem + (tb - smem_lds))) atomicAdd(a.verify, 1u);
                            }
                            ada[k] = oka ? ta : zaddr;
                            adb[k] = okb ? tb : zaddr;
                            cnt[k] += (oka ? 1u : 0u) + (okb ? 1u : 0u);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < CPL; k++) sum[k] += va[k] + vb[k];
#pragma unroll
                        for (int k = 0; k < CPL; k++) { va[k] = k1_lds_load(ada[k]); vb[k] = k1_lds_load(adb[k]); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (r < nr) {                                      // odd ray count: the last ray (range-tested in both kinds)
                    const float2 pa = pa_n;
#pragma unroll
                    for (int k = 0; k < CPL; k++) {
                        float fx, fy;
                        k1_coords(q[k], pa, fx, fy);
                        const int ix = (int)fx, iy = (int)fy;
                        const bool ok = ((unsigned)(ix - x0a) < (unsigned)w8) & ((unsigned)(iy - y0) < (unsigned)h);
                        if (VERIFY) {
                            if (!ok && !checked) atomicAdd(a.verify, 1u);
                            if (ok && map[(size_t)iy * S + ix] != *(const uint16_t *)(smem + (k1_tile_addr(ix, iy, pitch2, kofs) - smem_lds))) atomicAdd(a.verify, 1u);
                        }
                        sum[k] += k1_lds_load(ok ? k1_tile_addr(ix, iy, pitch2, kofs) : zaddr);
                        if (checked) cnt[k] += ok ? 1u : 0u;
                    }
                }
#pragma unroll
                for (int k = 0; k < CPL; k++) sum[k] += va[k] + vb[k];
                if (VERIFY && t == 0) atomicAdd(a.verify + (checked ? 4 : 1), (unsigned)(nr * 4));
            } else {
                // no tile covers this piece (box wider than 512 px or taller than K1_MAXBANDS bands: candidates spread
                // over a large part of the map): bounds-checked global gathers, several in flight per lane
                int r = 0;
                for (; r + 1 < nr; r += 2) {
                    const float2 pa = k1_point_lds(pbase + r * 8), pb = k1_point_lds(pbase + r * 8 + 8);
                    uint32_t ga[CPL], gb[CPL]; bool oka[CPL], okb[CPL];
#pragma unroll
                    for (int k = 0; k < CPL; k++) {
                        float fxa, fya, fxb, fyb;
                        k1_coords(q[k], pa, fxa, fya);
                        k1_coords(q[k], pb, fxb, fyb);
                        const int ixa = (int)fxa, iya = (int)fya, ixb = (int)fxb, iyb = (int)fyb;
                        oka[k] = ((unsigned)ixa < (unsigned)S) & ((unsigned)iya < (unsigned)S);
                        okb[k] = ((unsigned)ixb < (unsigned)S) & ((unsigned)iyb < (unsigned)S);
                        ga[k] = map[oka[k] ? (size_t)iya * S + ixa : 0];
                        gb[k] = map[okb[k] ? (size_t)iyb * S + ixb : 0];
                    }
#pragma unroll
                    for (int k = 0; k < CPL; k++) {
                        sum[k] += (oka[k] ? ga[k] : 0u) + (okb[k] ? gb[k] : 0u);
                        cnt[k] += (oka[k] ? 1u : 0u) + (okb[k] ? 1u : 0u);
                    }
                }
                if (r < nr) {
                    const float2 pa = k1_point_lds(pbase + r * 8);
#pragma unroll
                    for (int k = 0; k < CPL; k++) k1_gather_global<false>(map, S, q[k], pa, sum[k], cnt[k]);
                }
                if (VERIFY && t == 0) atomicAdd(a.verify + 3, (unsigned)(nr * 4));
            }
#ifdef K1_TIMES
            K1_SUB(2, wall_clock64() - ts2)
#endif
        }
#ifdef K1_TIMES
        K1_SUB(4, clock64() - sclk0)
#endif
#undef K1_PREFETCH
    }
    K1_STAMP(7)

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // Every candidate has one 64-bit accumulator: pixel sum (36 bits) | workgroups that saw an end point of it in the map (14) |
    // workgroups that have arrived (14).  A workgroup adds its share with one agent-scope atomic per candidate (performed at the
    // memory side: correct for any placement of the group's workgroups over CUs and XCDs) and gets the old value back: the lane
    // whose add completes the arrival count holds the candidate's total -- no drain, no ticket, no second read -- zeroes the
    // accumulator for the next launch and finishes the distance.  Workgroups that finished candidates min-reduce their keys into
    // one device word and add their number to a counter; the workgroup that completes the count reads the minimum back, resets both
    // words and delivers the result.  (Round 1 drained the adds, drew a ticket per group, had the last arriver swap the
    // accumulators back and the last group walk the group minima: five dependent round trips after the slowest workgroup's last
    // gather; now three.)  No fences, no spinning; accumulators, minimum (all ones) and counter (zero) are at rest between launches.
    typedef unsigned long long u64;
    // (slot k of lane t at k * LANES + t: the 64 adds of a wave instruction fall into four consecutive 128-byte lines.  The memory side
    // performs atomics a line at a time: with a lane's candidates adjacent -- every CPL-th word per instruction, 8 lines at two
    // candidates per lane, 16 at four -- the adds took twice / four times as long, and spaced 64 bytes apart, one line each, a
    // 16 384-candidate launch went from 25 to 38 us; measured)
    if (K1_EXP == 4) { if (sum[0] == 0x12345u && cnt[0] == 77u) a.verify[0] = 1u; return; }
    u64 *acc = a.acc + (size_t)g * GROUP + (size_t)t;
    u64 tot[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) {
        const u64 add = (u64)sum[k] | ((u64)((cnt[k] | cnt_all) ? 1u : 0u) << K1_ACC_INMAP) | (1ull << K1_ACC_ARRIVED);
        tot[k] = __hip_atomic_fetch_add(acc + k * LANES, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
    }
    u64 key = ~0ull;
    unsigned nfin = 0;
#pragma unroll
    for (int k = 0; k < CPL; k++) {
        if ((unsigned)(tot[k] >> K1_ACC_ARRIVED) == (unsigned)nc) {
            __hip_atomic_store(acc + k * LANES, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int j = g * GROUP + k * LANES + t;
            if (j < count) {
                const u64 kk = k1_finish(tot[k] & ((1ull << K1_ACC_INMAP) - 1), (uint32_t)(tot[k] >> K1_ACC_INMAP) & ((1u << (K1_ACC_ARRIVED - K1_ACC_INMAP)) - 1u),
                                         a.n_rays, a.ev_idx ? a.ev_idx[j] : j, a.dist_out);
                key = kk < key ? kk : key;
                nfin++;
            }
        }
    }
    K1_STAMP(8)
    if (__builtin_amdgcn_ballot_w64(nfin != 0) != 0) {             // (most workgroups complete nothing: no shuffles)
        for (int off = 32; off > 0; off >>= 1) {
            const u64 o = __shfl_down(key, off, 64);
            key = o < key ? o : key;
            nfin += (unsigned)__shfl_down((int)nfin, off, 64);
        }
    }
    if (a.ring_slot) {
        // ring mode: a wavefront that finished candidates contributes its minimum itself -- one atomic without return per such
        // wavefront (the finishing workgroup of a group: up to eight on one address, ~12 ns each at the memory side), no LDS round,
        // no barrier, nothing waits for anything: the launch's end is the completion.  Three dependent round trips shorter than
        // the chain below, which the slowest workgroup of the launch used to run after its last gather.
        if (lane == 0 && nfin != 0) (void)__hip_atomic_fetch_min(a.ring_slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        K1_STAMP(9)
        return;
    }
    if (lane == 0) { wkey[wv] = key; wfin[wv] = nfin; }
    __syncthreads();
    if (t != 0) return;
    for (int w = 1; w < NW; w++) { key = wkey[w] < key ? wkey[w] : key; nfin += wfin[w]; }
    if (nfin == 0) return;
    (void)__hip_atomic_fetch_min(a.gmin, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (with return: performed before the count below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned before = __hip_atomic_fetch_add(a.done, nfin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before + nfin != (unsigned)count) return;
    // the last finisher
    const u64 best = __hip_atomic_load(a.gmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.gmin, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the key at agent scope and acknowledged: the consumer behind the signal below is another kernel on this GPU -- the collective --
    // which then needs no system-scope release, i.e. no write-back of the L2, from this thread: ~3 us at the end of every launch)
    __hip_atomic_store(a.key_out, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.sig) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float bx = 0.0f, by = 0.0f, bthn = 0.0f;
    if (a.best_pose) {                                             // search_pose + offs[index - 1] (:635-637), theta normalised (:746)
        const uint32_t flat = (uint32_t)best;
        float th = a.bth;
        bx = a.bx; by = a.by;
        if (flat > 0) { bx = a.bx + a.offs_flat[3 * (flat - 1)]; by = a.by + a.offs_flat[3 * (flat - 1) + 1]; th = a.bth + a.offs_flat[3 * (flat - 1) + 2]; }
        bthn = sh_normalize_angle(th);
        a.best_pose[0] = bx; a.best_pose[1] = by; a.best_pose[2] = bthn;
        a.best_pose[3] = th;                                       // un-normalised, as MonteCarloSearch returns it
    }
    if (a.sig) __hip_atomic_store(a.sig, a.sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.done_flag) {                                             // blocking call: the key -- and the winner's pose when one is asked for -- into the mailbox, then its completion word
        *(unsigned long long *)(a.done_flag - 15) = best;
        if (a.best_pose) { float *mp = (float *)(a.done_flag - 13); mp[0] = bx; mp[1] = by; mp[2] = bthn; }
        __hip_atomic_store(a.done_flag, a.done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    K1_STAMP(9)
}


// ---- the plan kernel --------------------------------------------------------------------------------------------
// One launch per search, on a stream of its own beside the search launch (cs_launch_distance): what every workgroup of k1_search_tiled
// would otherwise work out in front of its first tile, made ONCE and left in device memory (k1_args: plan_rec).
//   blocks [0, n_wgs)       one wavefront per workgroup of the search launch: its ray range (k1_wg_decode -- the same layout), the group's
//                           bounds from the jitter bounds, the boxes and step records of its pieces (k1_pieces.inc -- the same text), and
//                           the record: 16 units of 8 words, [0] = {stamp, steps or -1, ...}, [1 + i] = step i, word 7 of every unit =
//                           stamp, all 512 bytes in ONE store instruction (a unit never exists without its stamp);
//   blocks [n_wgs, ...)     256 candidates each: (px, py, c, s) = k1_candidate (the deterministic trigonometry, bit for bit what the search
//                           kernel's lanes compute), stored THROUGH the L2 to memory (agent-scope stores: the L2 is per XCD) and
//                           waited for, THEN the stamps of the four 64-candidate runs.
// The search kernel trusts nothing without its stamp, so the launch may be late or missing; a stamp is the search's number, never reused.
__device__ static inline int k1_plan_slot(const int i) { return i <= K1_PLAN_STEPS ? i : K1_PLAN_STEPS + 1; }
template <int GROUP, int CPL>
__global__ void __launch_bounds__(64)
k1_plan(const k1_args a, const int n_wgs)
{
    constexpr int NW = 1, PF = 64;                                  // (k1_pieces.inc: PF * NW = the search workgroup's 64 staging vectors per lane and wave)
    const int lane = threadIdx.x, wv = 0;
    const uint32_t seq = a.plan_seq;
    if ((int)blockIdx.x >= n_wgs) return;
    // (little LDS and few registers: these wavefronts share the compute units with the search launch before this one, which leaves
    // ~23 KB of LDS and 96 registers per lane free -- the rays are read from memory where the search kernel keeps a copy in LDS,
    // and only the records that fit a plan record are kept)
    __shared__ __attribute__((aligned(16))) int stepbuf[(K1_PLAN_STEPS + 2) * 8];
    __shared__ int2 pieces[K1_MAXP];
    __shared__ __attribute__((aligned(16))) float bnd[8];
    __shared__ int s_nsteps;
    const int S = a.S;
    const k1_wg_t W = k1_wg_decode(a, (int)blockIdx.x);
    const int g = W.g, nbp = W.nbp, bp = W.bp, rlo = W.rlo, rhi = W.rhi, nrays = rhi - rlo;
    float gb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) gb[k] = a.grp_bounds[8 * (size_t)g + k];
    const int blk_first = a.ray_blk[rlo].z, blk_last = a.ray_blk[rhi - 1].z;
    if (lane == 0) s_nsteps = 0;
    for (int i = lane; i < nrays; i += 64) {
        const int4 ri = a.ray_blk[rlo + i];
        // (A plan launch that was kept waiting -- its queue behind another on the same compute pipe -- can run after its search has
        // finished, while the host stores the NEXT scan's tables into this very block: what it reads may be torn.  Its record will
        // carry a stamp nobody waits for; what it must not do is follow a torn index out of its arrays.)
        const int pidx = ri.z - blk_first;
        if ((i == 0 || ri.x == rlo + i) && (unsigned)pidx < (unsigned)K1_MAXP) pieces[pidx] = make_int2(i, (ri.y < rhi ? ri.y : rhi) - (rlo + i));
    }
    const float2 *cpts = a.pts + rlo;
    k1_bounds_from_jitter(a, gb, bnd, lane);
    __syncthreads();
    const int npieces = min(blk_last - blk_first + 1, K1_MAXP);   // (a plan that lags so far behind that its scan's block is being rewritten must still end)
#define K1_PIECE_STAMP
#define K1_STEP_SLOT(i) k1_plan_slot(i)                        /* (a function: the argument is an atomicAdd) */
#define K1_PIECE_RAY(i) min(max((i), 0), nrays - 1)            /* (see above: torn tables) */
#include "k1_pieces.inc"
#undef K1_PIECE_RAY
#undef K1_STEP_SLOT
#undef K1_PIECE_STAMP
    __syncthreads();
    const int nsteps = s_nsteps;
    const bool fits = nsteps <= K1_PLAN_STEPS;
    const int unit = lane >> 2, wp = lane & 3;                     // lane l holds words 2 l, 2 l + 1 of the record
    uint2 w = make_uint2(0u, 0u);
    if (unit > 0 && fits && unit - 1 < nsteps) { w.x = (uint32_t)stepbuf[(unit - 1) * 8 + 2 * wp]; w.y = (uint32_t)stepbuf[(unit - 1) * 8 + 2 * wp + 1]; }
    if (lane == 0) { w.x = seq; w.y = (uint32_t)(fits ? nsteps : -1); }
    if (wp == 3) w.y = seq;
    // (through the L2 to memory as well: a search launch that is already running may read the record before this launch ends)
    __hip_atomic_store((unsigned long long *)(a.plan_rec + (size_t)blockIdx.x * (K1_PLAN_REC_WORDS / 2) + lane), ((unsigned long long)w.y << 32) | w.x,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- host side --------------------------------------------------------------------------------------
int32_t cs_plan_drain(slamhip_cs *cs)
{
    if (cs->plan_stream) SH_HIP(hipStreamSynchronize(cs->plan_stream));
    return SLAMHIP_OK;
}

void cs_plan_free(slamhip_cs *cs)
{
    if (cs->plan_stream) (void)hipStreamSynchronize(cs->plan_stream);
    for (int i = 0; i < K1_PLAN_SLOTS; i++) {
        (void)hipFree(cs->d_plan_rec[i]);
        cs->d_plan_rec[i] = nullptr; cs->plan_slot_user[i] = 0;
    }
    cs->plan_cap_wgs = 0;
}

int32_t cs_alloc_candidates(slamhip_cs *cs, int count)
{
    if (count <= cs->cap_cand) return SLAMHIP_OK;
    SH_TRY(cs_plan_drain(cs));                                    // (a plan launch may still read the lists freed below)
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
    if (cs->d_grp_bounds) (void)hipFree(cs->d_grp_bounds);
    cs->d_grp_bounds = nullptr; cs->cap_grp = 0;
    SH_HIP(hipMalloc(&cs->d_grp_bounds, sizeof(float) * 8 * (size_t)(cap / K1_GROUP_SMALL + 2)));      // (the smallest groups: the most)
    cs->cap_grp = cap / K1_GROUP_SMALL + 2;
    cs->cap_cand = cap;
    cs->shard_first = -1; cs->shard_count = -1;
    return SLAMHIP_OK;
}

static int32_t ensure_partial(slamhip_cs *cs, size_t bytes)
{
    if (bytes <= cs->cap_partial) return SLAMHIP_OK;
    if (cs->d_partial) (void)hipFree(cs->d_partial);
    cs->d_partial = nullptr; cs->cap_partial = 0;
    bytes += bytes / 4;
    SH_HIP(hipMalloc(&cs->d_partial, bytes));
    cs->cap_partial = bytes;
    return SLAMHIP_OK;
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// Do nc equal ray ranges hold at most K1_MAXR rays and K1_MAXP block pieces each?  (Not monotone in nc: the cuts move.)
static bool k1_chunks_legal(const slamhip_cs *cs, int nc)
{
    const int R = cs->n_points, n_rb = cs->n_rb;
    const int *rb = cs->h_rb_start.data();
    if (nc < 1 || nc > R) return false;
    int b = 0;
    // the cuts floor(c R / nc), c = 0 .. nc, without a division per cut: quotient and remainder advance by R / nc and R % nc
    const int dq = R / nc, dr = R % nc;
    int rlo = 0, rem = 0;
    for (int c = 0; c < nc; c++) {
        int rhi = rlo + dq;
        rem += dr;
        if (rem >= nc) { rem -= nc; rhi++; }
        while (b + 1 < n_rb && rb[b + 1] <= rlo) b++;                  // block of ray rlo
        int e = b;
        while (e + 1 < n_rb && rb[e + 1] < rhi) e++;                   // block of ray rhi - 1
        if (e - b + 1 > K1_MAXP || rhi - rlo > K1_MAXR) return false;
        rlo = rhi;
    }
    return true;
}

// The slowest of nc equal ray ranges in ray units, a tile step (a ray block fragment) counted as `step` rays: a step costs a
// workgroup about a microsecond of staging whatever its size (barrier, tile write, barrier, the next tile's loads), eight rays'
// worth of gathers, and where the cuts fall on the block boundaries a range is one step -- one ray range more or less per group
// moved a launch by 4 us (1024^2 map, 26 against 27 ranges per group).  < 0: not a legal count.
static double k1_chunks_score(const slamhip_cs *cs, int nc, double step)
{
    const int R = cs->n_points, n_rb = cs->n_rb;
    const int *rb = cs->h_rb_start.data();
    if (nc < 1 || nc > R) return -1.0;
    int b = 0;
    double worst = 0.0;
    const int dq = R / nc, dr = R % nc;                                // (the cuts as in k1_chunks_legal)
    int rlo = 0, rem = 0;
    for (int c = 0; c < nc; c++) {
        int rhi = rlo + dq;
        rem += dr;
        if (rem >= nc) { rem -= nc; rhi++; }
        while (b + 1 < n_rb && rb[b + 1] <= rlo) b++;                  // block of ray rlo
        int e = b;
        while (e + 1 < n_rb && rb[e + 1] < rhi) e++;                   // block of ray rhi - 1
        if (e - b + 1 > K1_MAXP || rhi - rlo > K1_MAXR) return -1.0;
        worst = std::max(worst, (double)(rhi - rlo) + step * (double)(e - b + 1));
        rlo = rhi;
    }
    return worst;
}

// Smallest legal chunk count >= nc (one ray per chunk is always legal).
static int k1_legal_chunks(const slamhip_cs *cs, int nc)
{
    const int R = cs->n_points;
    if (nc < sh_div_up(R, K1_MAXR)) nc = sh_div_up(R, K1_MAXR);
    if (nc > R) nc = R;
    while (nc < R && !k1_chunks_legal(cs, nc)) nc++;
    return nc;
}

static thread_local double g_cut_t[4] = { 0, 0, 0, 0 };   // developer aid (SLAMHIP_K1_CUT_TIMES): host microseconds in weights / balanced cuts / banded check
static inline double k1_now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec * 1e6 + (double)t.tv_nsec * 1e-3; }
// Ray ranges cut by cost.  A workgroup's compute phase is its rays PLUS its tile steps -- one per ray block it touches -- and a
// step costs as much as a dozen or two rays of gathers (barrier, tile write, barrier, the next tile's loads; SLAMHIP_K1_TIMES at
// the headline size: 43-ray ranges of 2 steps take 8 us, of 5 steps 13 us, and the launch waits for the slowest).  Equal COUNTS
// of rays per range leave that to where the cuts happen to fall on the block boundaries; here the cuts minimise the largest
// rays + sum of the weights wb[] of the blocks touched, over at most nc ranges: binary search on the bound, greedy fill (a range
// takes whole blocks while they fit, then a fragment of the next one -- but no fragment shorter than half its block's weight: a
// step that short is all overhead).  A block's weight grows with its tile (k1_cut_weights).
// Every range holds <= K1_MAXR rays of <= K1_MAXP blocks.  Returns the number of ranges (<= nc; 0: no such cut), cuts[0 .. n].
static int k1_balanced_cuts(const slamhip_cs *cs, int nc, const std::vector<double> &wb, std::vector<int> &cuts)
{
    const int R = cs->n_points, n_rb = cs->n_rb;
    const int *rb = cs->h_rb_start.data();
    if (nc < 1 || R < 1 || n_rb < 1 || (int)wb.size() != n_rb) return 0;
    double wsum = 0.0, wmax = 0.0;
    for (int b = 0; b < n_rb; b++) { wsum += wb[(size_t)b]; wmax = std::max(wmax, wb[(size_t)b]); }
    auto fill = [&](double T, std::vector<int> *out) -> int {
        int r = 0, b = 0, n = 0;
        if (out) { out->clear(); out->push_back(0); }
        while (r < R) {
            while (b + 1 < n_rb && rb[b + 1] <= r) b++;                // block of ray r
            int e = r, bb = b, pieces = 0;
            double wacc = 0.0;
            while (e < R && pieces < K1_MAXP) {
                const int bend = rb[bb + 1];
                const double w = wb[(size_t)bb];
                const int min_frag = (int)(w * 0.5) > 1 ? (int)(w * 0.5) : 1;
                const double room = T - ((double)(e - r) + wacc + w);
                int take = bend - e;
                if (take > K1_MAXR - (e - r)) take = K1_MAXR - (e - r);
                const bool whole = take == bend - e && (double)take <= room;
                if (!whole) {
                    int part = room < (double)take ? (int)room : take;
                    if (part < 0) part = 0;
                    // a fragment: only if it is worth a step, and never one that leaves a shorter-than-worthwhile rest
                    if (pieces > 0 && (part < min_frag || (bend - e) - part < min_frag)) part = 0;
                    if (pieces == 0 && part < 1) return 1 << 30;       // the bound does not even admit one ray: infeasible
                    e += part;
                    if (part > 0) pieces++;
                    break;
                }
                e = bend; pieces++; bb++; wacc += w;
            }
            if (e == r) return 1 << 30;
            if (out) out->push_back(e);
            r = e; n++;
            if (n > nc) return n;
        }
        return n;
    };
    // (the bound lies between the mean cost of a range and the whole scan's: start near the mean, widen until a bound fits)
    // (this runs once per scan on the host's critical path in the per-scan flow: a handful of fills of ~0.3 us each)
    double lo = ((double)R + wsum) / (double)nc - 1.0, hi = lo + wmax + 3.0;
    const double top = (double)R + wsum + 1.0;
    if (lo < 1.0) lo = 1.0;
    for (int it = 0; it < 12 && hi < top && fill(hi, nullptr) > nc; it++) { lo = hi; hi = hi * 1.3 + 2.0; }
    if (hi > top) hi = top;
    if (hi >= top && fill(hi, nullptr) > nc) return 0;
    for (int it = 0; it < 24 && hi - lo > 1.5; it++) {
        const double mid = 0.5 * (lo + hi);
        if (fill(mid, nullptr) <= nc) hi = mid; else lo = mid;
    }
    const int n = fill(hi, &cuts);
    if (n < 1 || n > nc || cuts.back() != R) return 0;
    return n;
}

// The weight of a ray block in k1_balanced_cuts, in ray units: what one more tile step costs a workgroup.  Barriers, the tile's
// LDS writes and the issue of its loads all grow with the tile, so: w_fix + w_kb * (the tile's size in KB for the middle candidate
// group -- the block's bounding box turned into the map frame, grown by the group's translation spread and theta arc: the terms
// of k1_group_cost).  Without the groups' spreads (explicit pose lists): the box grown by a nominal 48 pixels.
static void k1_cut_weights(const slamhip_cs *cs, int n_groups, bool have_spread, double w_fix, double w_kb, int budget, std::vector<double> &wb)
{
    const int n_rb = cs->n_rb;
    std::vector<char> &cand = const_cast<slamhip_cs *>(cs)->k1_cut_cand;
    cand.assign((size_t)n_rb, 0);
    wb.assign((size_t)n_rb, w_fix);
    // the widest uniform group (the uniform part's outer groups): a block whose tile for THAT group may exceed the budget is marked
    // for the exact box test of the pieces (k1_cuts_banded_rays)
    double dth_w = 0.0, d_w = 48.0;
    if (have_spread && cs->k1_uni_ng > 0) {
        for (int g = cs->k1_uni_g0; g < cs->k1_uni_g0 + cs->k1_uni_ng && (size_t)g < cs->h_grp_dth.size(); g++) dth_w = std::max(dth_w, (double)cs->h_grp_dth[(size_t)g]);
        d_w = 4.0;
        for (int g = cs->k1_uni_g0; g < cs->k1_uni_g0 + cs->k1_uni_ng && (size_t)g < cs->h_grp_dxy.size(); g++) d_w = std::max(d_w, (double)cs->h_grp_dxy[(size_t)g] + 4.0);
    }
    const double th = have_spread ? (double)cs->k1_layout_theta : 0.0;
    const double c = fabs(cos(th)), s = fabs(sin(th));
    const int gm = n_groups / 2;
    const double dth = have_spread && (size_t)gm < cs->h_grp_dth.size() ? (double)cs->h_grp_dth[(size_t)gm] : 0.0;
    const double d = have_spread && (size_t)gm < cs->h_grp_dxy.size() ? (double)cs->h_grp_dxy[(size_t)gm] + 4.0 : 48.0;
    for (int b = 0; b < n_rb; b++) {
        const double ex = cs->h_rb_ex[(size_t)b], ey = cs->h_rb_ey[(size_t)b], mx = fabs(cs->h_rb_mx[(size_t)b]), my = fabs(cs->h_rb_my[(size_t)b]);
        const double w = ex * c + ey * s + d + (mx * s + my * c) * dth, h = ex * s + ey * c + d + (mx * c + my * s) * dth;
        wb[(size_t)b] = w_fix + w_kb * (2.0 * (w + 8.0) * h / 1024.0);
        // (a first, generous estimate for the widest uniform group picks the blocks worth the exact box below)
        const double ww = ex * c + ey * s + d_w + (mx * s + my * c) * dth_w, hw = ex * s + ey * c + d_w + (mx * c + my * s) * dth_w;
        static const double cand_f = getenv("SLAMHIP_K1_CUT_CAND") ? atof(getenv("SLAMHIP_K1_CUT_CAND")) : 0.9;
        cand[(size_t)b] = 2.0 * (ww + 8.0) * hw > cand_f * (double)budget || ww > 440.0;
    }
}

// Would rays [r0, r1) of the sorted scan, as ONE piece, be staged in bands (or gathered from memory) for candidate group g?
// The kernel's own decision on the host: the box of the end points by k1_ray_box's interval arithmetic from the group's jitter
// bounds (k1_search_tiled's prologue), against the tile budget and the staging pass.
struct k1_host_bounds { float pxl, pxh, pyl, pyh, cmin, cmax, smin, smax; };
static k1_host_bounds k1_group_bounds_host(const slamhip_cs *cs, int g, const float pose[3])
{
    const float *lh = &cs->h_grp_lohi[(size_t)g * 6];
    const float scale = cs->hscale;
    k1_host_bounds b;
    b.pxl = (pose[0] + lh[0]) * scale + 0.5f; b.pxh = (pose[0] + lh[1]) * scale + 0.5f;
    b.pyl = (pose[1] + lh[2]) * scale + 0.5f; b.pyh = (pose[1] + lh[3]) * scale + 0.5f;
    const double tl = (double)pose[2] + lh[4], thh = (double)pose[2] + lh[5];
    double cl = std::min(cos(tl), cos(thh)), ch = std::max(cos(tl), cos(thh)), sl = std::min(sin(tl), sin(thh)), sh = std::max(sin(tl), sin(thh));
    for (int k = (int)ceil(tl / 1.5707963267948966); (double)k * 1.5707963267948966 <= thh; k++) {
        switch (((k % 4) + 4) % 4) { case 0: ch = 1.0; break; case 1: sh = 1.0; break; case 2: cl = -1.0; break; default: sl = -1.0; break; }
    }
    const float pad = scale * 1.0e-4f;
    b.cmin = (float)cl * scale - pad; b.cmax = (float)ch * scale + pad; b.smin = (float)sl * scale - pad; b.smax = (float)sh * scale + pad;
    return b;
}
static int k1_nopad() { static const int v = env_int("SLAMHIP_K1_NOPAD", 0); return v; }     // (tuning: tiles without the pitch padding)
static bool k1_piece_banded(const slamhip_cs *cs, const k1_host_bounds &b, int r0, int r1, int budget)
{
    const float *pts = (const float *)((const char *)cs->h_scan_blob + (size_t)cs->cap_points * 24);     // the sorted rays (set_scan's staging block)
    const int S = cs->hs;
    float x0 = 1e30f, x1 = -1e30f, y0 = 1e30f, y1 = -1e30f;
    for (int r = r0; r < r1; r++) {
        const float X = pts[2 * (size_t)r], Y = pts[2 * (size_t)r + 1];
        const float cx0 = b.cmin * X, cx1 = b.cmax * X, sy0 = b.smin * Y, sy1 = b.smax * Y, sx0 = b.smin * X, sx1 = b.smax * X, cy0 = b.cmin * Y, cy1 = b.cmax * Y;
        x0 = fminf(x0, b.pxl + fminf(cx0, cx1) - fmaxf(sy0, sy1)); x1 = fmaxf(x1, b.pxh + fmaxf(cx0, cx1) - fminf(sy0, sy1));
        y0 = fminf(y0, b.pyl + fminf(sx0, sx1) + fminf(cy0, cy1)); y1 = fmaxf(y1, b.pyh + fmaxf(sx0, sx1) + fmaxf(cy0, cy1));
    }
    const int ix0 = std::max((int)x0, 0), iy0 = std::max((int)y0, 0), ix1 = std::min((int)x1, S - 1), iy1 = std::min((int)y1, S - 1);
    if (ix1 < ix0 || iy1 < iy0) return false;
    // (the kernel's pitch padding never changes the number of bands: the plain width decides)
    const int xa = ix0 & ~7, ww = ((ix1 - xa + 1) + 7) & ~7;
    const int vpr = ww >> 3;
    int shf = 0; while ((1 << shf) < vpr) shf++;
    const int hmax = std::min(budget / (ww * 2), 4096 >> shf), H = iy1 - iy0 + 1;
    return vpr > 64 || hmax < 1 || H > hmax;
}
// Do the cuts put a piece on a banded tile, for one of the two outer groups of the uniform part, that the equal-count ranges would
// not?  (A block's fragment can span as much of the map as the whole block -- the rays of a block are in Z order, not along the
// wall -- so this is asked of the pieces as cut, not estimated per block.)  Counts the rays on banded pieces.
static int k1_cuts_banded_rays(const slamhip_cs *cs, const std::vector<int> &cuts, const float pose[3], int budget, const std::vector<char> &cand)
{
    if (cs->k1_uni_ng <= 0 || cs->h_grp_lohi.size() < (size_t)(cs->k1_uni_g0 + cs->k1_uni_ng) * 6 || !cs->h_scan_blob) return 0;
    const int n_rb = cs->n_rb;
    const int *rb = cs->h_rb_start.data();
    int banded = 0;
    for (int e = 0; e < 2; e++) {
        const int g = e == 0 ? cs->k1_uni_g0 : cs->k1_uni_g0 + cs->k1_uni_ng - 1;
        if (e == 1 && g == cs->k1_uni_g0) break;
        const k1_host_bounds b = k1_group_bounds_host(cs, g, pose);
        int blk = 0;
        for (size_t c = 0; c + 1 < cuts.size(); c++) {
            const int lo = cuts[c], hi = cuts[c + 1];
            while (blk + 1 < n_rb && rb[blk + 1] <= lo) blk++;
            for (int bb = blk; bb < n_rb && rb[bb] < hi; bb++) {
                if (!cand[(size_t)bb]) continue;
                const int r0 = std::max(lo, rb[bb]), r1 = std::min(hi, rb[bb + 1]);
                if (r1 > r0 && k1_piece_banded(cs, b, r0, r1, budget)) banded += r1 - r0;
            }
        }
    }
    return banded;
}

// Cost of staging one band of a banded tile, in ray units (measured best on MI355X: 5; a huge value: multi-band tiles
// never, global gathers instead; a hugely negative one: bands whenever they fit)
static float k1_band_stage()
{
    static const float v = getenv("SLAMHIP_K1_BAND_STAGE") ? (float)atof(getenv("SLAMHIP_K1_BAND_STAGE")) : 5.0f;
    return v;
}

// Estimated cost of a group in ray units (tile steps cost 1 per ray): from the theta range and the translation spread
// of the group (ensure_shard), the bounding box of each ray block (set_scan) and the search pose's heading.  A box
// beyond the tile budget is staged in bands with range-tested gathers.  Only the balance of the launch depends on
// this estimate.
// (The terms of a block that do not depend on the group are made once per layout -- k1_block_terms -- and the sum keeps the
// order of the blocks: same doubles as the plain loop, a fifth of its time; the layout is remade for every scan, on the host's
// critical path between two scans.)
static void k1_block_terms(const slamhip_cs *cs, std::vector<k1_block_term> &t)
{
    const double c = fabs(cos((double)cs->k1_layout_theta)), s = fabs(sin((double)cs->k1_layout_theta));
    t.resize((size_t)cs->n_rb);
    for (int b = 0; b < cs->n_rb; b++) {
        // the block's bounding box and centre turned into the map frame (search pose theta)
        const double ex = cs->h_rb_ex[(size_t)b], ey = cs->h_rb_ey[(size_t)b], mx = fabs(cs->h_rb_mx[(size_t)b]), my = fabs(cs->h_rb_my[(size_t)b]);
        k1_block_term &e = t[(size_t)b];
        e.a = ex * c + ey * s; e.b = ex * s + ey * c; e.m = mx * s + my * c; e.n = mx * c + my * s;
        e.nr = cs->h_rb_start[(size_t)b + 1] - cs->h_rb_start[(size_t)b];
    }
}
static double k1_group_cost(const slamhip_cs *cs, const std::vector<k1_block_term> &t, int g, int budget)
{
    const double dth = cs->h_grp_dth[(size_t)g], d = cs->h_grp_dxy[(size_t)g] + 4.0;
    static const double f_band = getenv("SLAMHIP_K1_FBAND") ? atof(getenv("SLAMHIP_K1_FBAND")) : 1.9;
    static const double f_glob = getenv("SLAMHIP_K1_FGLOBAL") ? atof(getenv("SLAMHIP_K1_FGLOBAL")) : 3.0;
    const double stage = (double)k1_band_stage();
    double cost = 0.0;
    for (size_t b = 0; b < t.size(); b++) {
        // ... grown by the translation spread and by the arc the centre sweeps over the group's theta range
        const k1_block_term &e = t[b];
        const double w = e.a + d + e.m * dth, h = e.b + d + e.n * dth;
        const double bytes = 2.0 * (w + 8.0) * h;
        double f = 1.0;
        if (bytes > budget) {
            const double bands = ceil(bytes / budget);
            // (cost per ray relative to a plain tile step.  The kernel takes bands where they pay against 4.5 for a global gather
            // (k1_search_tiled) -- what a gather costs the workgroup that issues it.  In the balance of the launch a global-gather
            // step weighs less: it leaves the VALU and the LDS to the workgroup it shares the compute unit with.  Measured, three
            // runs each, 4.5 -> 3.0: 16 384 candidates 27.1 -> 26.1 us per launch with events, sigma_theta 20 degrees 36.0 -> 31.9,
            // 4096^2 map with 32 768 candidates 68.2 -> 51.5, 4096 candidates 23.8 -> 22.5, the other sizes unchanged; 3.5 and 2.5
            // each have sizes that lose 4-12 us to a second round of workgroups.)
            const double nr = e.nr;
            const bool pay = bands <= 1.0 || bands * (stage + f_band * nr) < f_glob * nr;
            f = bands <= K1_MAXBANDS && w <= 504.0 && pay ? f_band * bands + (bands > 1.0 ? bands * fmax(stage, 0.0) / fmax(nr, 1.0) : 0.0) : f_glob;
        }
        cost += f * e.nr;
    }
    return cost;
}

// Launch layout (cs->k1_*): the groups whose estimated cost per ray is well above a plain group's get their own,
// larger chunk counts (at most K1_TABLE_G groups, the most expensive first); the rest share one count.
static void k1_make_layout(slamhip_cs *cs, int n_groups, int target_wgs, int budget, bool have_spread, int band_parts)
{
    const int R = cs->n_points;
    cs->k1_tab_group.clear(); cs->k1_tab_nc.clear(); cs->k1_tab_nbp.clear();
    cs->k1_uni_g0 = 0; cs->k1_uni_ng = n_groups;
    int uni_want = (int)((double)target_wgs / n_groups + 0.5);
    if (have_spread) {
        // Groups in the middle of the theta order cost the same and form the uniform part; groups that cost clearly
        // more (theta tails: banded tiles) are listed with their own chunk counts.  With many groups only the outer
        // K1_TABLE_G / 2 on each side are examined.
        const int side = n_groups <= K1_TABLE_G ? n_groups : K1_TABLE_G / 2;
        std::vector<k1_block_term> &terms = cs->k1_terms;          // (kept between scans: no allocation per layout)
        k1_block_terms(cs, terms);
        const double ref = k1_group_cost(cs, terms, n_groups / 2, budget);
        std::vector<double> &cost = cs->k1_cost;
        cost.assign((size_t)n_groups, ref);
        for (int g = 0; g < n_groups; g++)
            if (g < side || g >= n_groups - side) cost[(size_t)g] = k1_group_cost(cs, terms, g, budget);
        int lo = n_groups / 2, hi = n_groups / 2 + 1;
        while (lo > 0 && cost[(size_t)lo - 1] <= 1.15 * ref && (n_groups <= K1_TABLE_G || lo - 1 >= 0)) lo--;
        while (hi < n_groups && cost[(size_t)hi] <= 1.15 * ref) hi++;
        if (lo > K1_TABLE_G / 2) lo = K1_TABLE_G / 2;              // (at most K1_TABLE_G listed groups)
        if (n_groups - hi > K1_TABLE_G / 2) hi = n_groups - K1_TABLE_G / 2;
        double total = (double)(hi - lo) * ref;
        for (int g = 0; g < lo; g++) total += cost[(size_t)g];
        for (int g = hi; g < n_groups; g++) total += cost[(size_t)g];
        double per_cost = (double)target_wgs / total;
        uni_want = (int)floor(ref * per_cost + 0.5);
        static const double step_cost = getenv("SLAMHIP_K1_STEPCOST") ? atof(getenv("SLAMHIP_K1_STEPCOST")) : 8.0;
        if (n_groups <= K1_TABLE_G && step_cost > 0.0 && uni_want >= 1) {
            // One round of workgroups: the uniform part's count of ray ranges is the one, near the proportional share, whose slowest
            // range -- steps included -- balances best against the slowest listed group with what is left (weighted 1.5: its
            // gathers miss the L2 on large maps).  Measured against the proportional shares, two runs each, us per launch with
            // events: 16 384 candidates 26.8 -> 25.7, 8192: 25.1 -> 23.3, sigma_theta 20 degrees 31.9 -> 30.4, 4096^2 map with
            // 16 384 / 32 768 candidates 46.6 -> 40.5 / 51.6 -> 50.8, 1024^2 23.9 -> 23.7; 360-ray scans lose 0.9 and 32 768
            // candidates 0.7.  The landscape is rough (one ray range more or less moves a launch by up to 4 us), the rule is a
            // compromise over these sizes.
            const double list_total = total - (double)(hi - lo) * ref;
            const int nl = lo + (n_groups - hi), min_legal = k1_legal_chunks(cs, 1);   // (a listed group gets a legal count too)
            int best_nc = -1; double best_t = 0.0;
            for (int nc = std::max(1, (int)(0.6 * uni_want)); nc <= (int)(1.6 * uni_want) + 1; nc++) {
                const long long left = (long long)target_wgs - (long long)(hi - lo) * nc;
                if (nl > 0 ? left < (long long)nl * std::max(2, min_legal) : left < 0) break;
                const double tu = k1_chunks_score(cs, nc, step_cost) * (ref / (double)R);
                if (tu < 0.0) continue;
                double tt = 0.0;
                if (nl > 0) {
                    // what the listed groups get out of `left` workgroups: their proportional shares, none below the smallest legal
                    // count, the largest trimmed until the sum fits -- and the slowest of them
                    std::vector<int> &share = cs->k1_share; std::vector<double> &lc = cs->k1_lc;
                    share.clear(); lc.clear();
                    long long sum = 0;
                    for (int g = 0; g < n_groups; g++) if (g < lo || g >= hi) {
                        const int v = std::max(min_legal, (int)floor(cost[(size_t)g] * (double)left / list_total + 0.5));
                        share.push_back(v); lc.push_back(cost[(size_t)g]); sum += v;
                    }
                    for (int guard = 0; sum > left && guard < 4096; guard++) {
                        size_t im = 0;
                        for (size_t i = 1; i < share.size(); i++) if (share[i] > share[im]) im = i;
                        if (share[im] <= min_legal) break;
                        share[im]--; sum--;
                    }
                    static const double tail_w = getenv("SLAMHIP_K1_TAILW") ? atof(getenv("SLAMHIP_K1_TAILW")) : 1.5;
                    for (size_t i = 0; i < share.size(); i++) tt = std::max(tt, tail_w * lc[i] / (double)share[i] + step_cost);
                }
                const double tm = std::max(tu, tt);
                if (best_nc < 0 || tm < best_t) { best_nc = nc; best_t = tm; }
            }
            if (best_nc > 0) {
                uni_want = best_nc;
                if (list_total > 0.0) per_cost = (double)(target_wgs - (long long)(hi - lo) * best_nc) / list_total;
            }
        }
        const int n_list = lo + (n_groups - hi);
        for (int p = 0; p < 2 * std::max(lo, n_groups - hi); p++) {    // dispatch order: theta extremes first
            const int k = p >> 1, g = (p & 1) ? n_groups - 1 - k : k;
            if ((p & 1) ? k >= n_groups - hi : k >= lo) continue;
            int v = (int)floor(cost[(size_t)g] * per_cost + 0.5);
            if (v < 1) v = 1;
            if (v > R / 4) v = R / 4 > 0 ? R / 4 : 1;
            int nbp = 1;                                           // banded tiles (two bands or more on average): chunks in sets
            static const double f_band = getenv("SLAMHIP_K1_FBAND") ? atof(getenv("SLAMHIP_K1_FBAND")) : 1.9;
            if (band_parts > 1 && cost[(size_t)g] >= 2.0 * f_band * R && v >= 2 * band_parts) nbp = band_parts;
            cs->k1_tab_group.push_back(g); cs->k1_tab_nc.push_back(v); cs->k1_tab_nbp.push_back(nbp);
        }
        (void)n_list;
        cs->k1_uni_g0 = lo; cs->k1_uni_ng = hi - lo;
    }
    if (uni_want < 1) uni_want = 1;
    // legal chunk counts (rays and pieces per chunk); identical requests share the search
    cs->k1_uni_nc = k1_legal_chunks(cs, uni_want);
    int req = -1, res = -1;
    for (size_t i = 0; i < cs->k1_tab_nc.size(); i++) {
        const int nbp = cs->k1_tab_nbp[i], want = cs->k1_tab_nc[i] / nbp > 0 ? cs->k1_tab_nc[i] / nbp : 1;
        if (want != req) { req = want; res = k1_legal_chunks(cs, req); }
        cs->k1_tab_nc[i] = res * nbp;
    }
    // one workgroup too many starts a second round on a full chip: take an excess over the target from the listed
    // groups with the most chunks (small launches only)
    long long tot = (long long)cs->k1_uni_ng * cs->k1_uni_nc;
    for (size_t i = 0; i < cs->k1_tab_nc.size(); i++) tot += cs->k1_tab_nc[i];
    if (n_groups <= K1_TABLE_G) {
        for (int guard = 0; tot > target_wgs && guard < 4096 && !cs->k1_tab_nc.empty(); guard++) {
            size_t im = 0;
            for (size_t i = 1; i < cs->k1_tab_nc.size(); i++) if (cs->k1_tab_nc[i] > cs->k1_tab_nc[im]) im = i;
            // the next smaller count of ray ranges that is legal itself (a count below a legal one need not be: tests/
            // fuzz_parity.py found a 2500-ray scan of one-ray blocks whose trimmed count put 18 pieces into a chunk)
            const int nbp = cs->k1_tab_nbp[im];
            int nrc = cs->k1_tab_nc[im] / nbp - 1;
            while (nrc >= 1 && !k1_chunks_legal(cs, nrc)) nrc--;
            if (nrc < 1) break;
            tot -= cs->k1_tab_nc[im] - nrc * nbp;
            cs->k1_tab_nc[im] = nrc * nbp;
        }
    }
    if (n_groups <= K1_TABLE_G) {
        // still more than one round (the listed groups' legal counts are sparse: trimming them stopped short): the uniform part
        // gives way, one legal count at a time
        for (int guard = 0; tot > target_wgs && cs->k1_uni_nc > 1 && guard < 4096; guard++) {
            int nc = cs->k1_uni_nc - 1;
            while (nc >= 1 && !k1_chunks_legal(cs, nc)) nc--;
            if (nc < 1) break;
            tot -= (long long)cs->k1_uni_ng * (cs->k1_uni_nc - nc);
            cs->k1_uni_nc = nc;
        }
    }
    long long tab = 0;
    for (size_t i = 0; i < cs->k1_tab_nc.size(); i++) tab += cs->k1_tab_nc[i];
    if (tab > K1_TABLE_WGS) {                                      // (huge scans: legal chunk counts alone overflow the table)
        cs->k1_tab_group.clear(); cs->k1_tab_nc.clear(); cs->k1_tab_nbp.clear();
        cs->k1_uni_g0 = 0; cs->k1_uni_ng = n_groups;
    }
}

// Are the layout's counts of ray ranges legal for the scan now set (k1_chunks_legal)?
static bool k1_layout_legal(const slamhip_cs *cs)
{
    if (cs->k1_uni_ng > 0 && !k1_chunks_legal(cs, cs->k1_uni_nc)) return false;
    int seen = -1;
    for (size_t i = 0; i < cs->k1_tab_nc.size(); i++) {
        const int nbp = cs->k1_tab_nbp[i] > 0 ? cs->k1_tab_nbp[i] : 1, nrc = cs->k1_tab_nc[i] / nbp;
        if (nrc * nbp != cs->k1_tab_nc[i]) return false;
        if (nrc != seen) { if (!k1_chunks_legal(cs, nrc)) return false; seen = nrc; }
    }
    return true;
}

// Is a cut -- ray ranges [cuts[c], cuts[c + 1]) -- legal for the scan now set: it covers the scan, and every range holds at most K1_MAXR
// rays of at most K1_MAXP ray blocks?
static bool k1_cuts_legal(const slamhip_cs *cs, const std::vector<int> &cuts)
{
    const int R = cs->n_points, n_rb = cs->n_rb;
    const int *rb = cs->h_rb_start.data();
    if (cuts.size() < 2 || cuts.front() != 0 || cuts.back() != R || n_rb < 1) return false;
    int b = 0;
    for (size_t c = 0; c + 1 < cuts.size(); c++) {
        const int lo = cuts[c], hi = cuts[c + 1];
        if (hi <= lo || hi - lo > K1_MAXR) return false;
        while (b + 1 < n_rb && rb[b + 1] <= lo) b++;
        int e = b;
        while (e + 1 < n_rb && rb[e + 1] < hi) e++;
        if (e - b + 1 > K1_MAXP) return false;
    }
    return true;
}

// The weight of a tile step in ray units for the cuts by cost (see cs_launch_distance): 26 where it was calibrated (2048^2 map = 51.2
// pixels per metre, two candidates per lane), less on coarser maps and in proportion to what a ray costs the workgroup.
static double k1_cut_wfix(const slamhip_cs *cs, int group)
{
    static const double cut_w20 = getenv("SLAMHIP_K1_CUT_WFIX") ? atof(getenv("SLAMHIP_K1_CUT_WFIX")) : 26.0;
    const int cpl_group = group == K1_GROUP_BIG ? 4 : group == K1_GROUP_SMALL ? 1 : 2;
    return cut_w20 * std::min(1.0, (double)cs->hscale / 51.2) * 2.0 / (double)cpl_group;
}

// One cut of the scan now set into nrc ranges by cost (k1_balanced_cuts with the block weights cs->k1_cut_wb, scaled down until the cut
// keeps nearly all the ranges asked for; dropped if it puts more rays on banded tiles than the equal-count ranges do).  c: empty = none.
static void k1_cuts_make(slamhip_cs *cs, int nrc, bool have_spread, const float pose3[3], int budget, std::vector<int> &c)
{
    static const int cut_keep = env_int("SLAMHIP_K1_CUT_KEEP", 90);   // per cent of the asked-for ranges a cut must keep
    c.clear();
    // The cut must keep (nearly) the asked-for count of ranges: where blocks are long in rays (coarse maps: a 64-ray block is
    // one tile) heavy weights end in one block per range -- 17 ranges for 25 at 1024^2 -- and the workgroups that are not
    // launched cost more than the steps that are saved (measured: 24 -> 33 us).  The weights are scaled down until it does.
    std::vector<double> &wsc = cs->k1_cut_wsc;
    double scale = 1.0;
    for (int tries = 0; tries < 6; tries++, scale *= 0.6) {
        wsc.resize(cs->k1_cut_wb.size());
        for (size_t i = 0; i < wsc.size(); i++) wsc[i] = cs->k1_cut_wb[i] * scale;
        const double tb0 = k1_now_us();
        const int n = k1_balanced_cuts(cs, nrc, wsc, c);
        g_cut_t[1] += k1_now_us() - tb0;
        if (n < 1) { c.clear(); break; }
        if (n * 100 >= nrc * cut_keep) break;
        c.clear();
    }
    if (!c.empty() && nrc == cs->k1_uni_nc && have_spread) {
        // a cut that puts more rays on banded tiles (for the uniform part's outer groups) than the equal-count ranges do is
        // dropped: a banded piece costs its workgroup more than twice a plain one, the launch waits for it (1024^2 map:
        // four workgroups at 19 us in a 21 us launch)
        const double tc0 = k1_now_us();
        const int banded = k1_cuts_banded_rays(cs, c, pose3, budget, cs->k1_cut_cand);
        g_cut_t[2] += k1_now_us() - tc0;
        if (banded > 0) {                                  // (seldom: the equal-count ranges are only looked at then)
            std::vector<int> eq((size_t)nrc + 1);
            for (int k = 0; k <= nrc; k++) eq[(size_t)k] = (int)(((long long)k * cs->n_points) / nrc);
            if (banded > k1_cuts_banded_rays(cs, eq, pose3, budget, cs->k1_cut_cand)) c.clear();
        }
    }
}

// Are the layout's counts of ray ranges -- and the cut the launch now in the stream took over from the previous scan -- legal for the
// scan now set?  (The search launched ahead of its scan's tables asks when the tables exist.)
bool cs_k1_layout_legal(const slamhip_cs *cs)
{
    if (!k1_layout_legal(cs)) return false;
    return !cs->k1_launch_prev_cuts || k1_cuts_legal(cs, cs->k1_prev_cuts);
}

// The layout for the scan now set, made while the host has nothing else to do (it waits for a search's result): the launch that
// just left used the previous scan's (cs_launch_distance) -- and, for the next scan's launch, the uniform part's ray ranges cut by
// cost (the per-scan flow sees every scan once: its launch takes the cut of the scan before, if legal).  Touches host state only.
void cs_layout_idle_refresh(slamhip_cs *cs)
{
    if (cs->k1_layout_dirty || cs->k1_scan_dirty || cs->n_points <= 0) return;
    if (cs->k1_layout_stale) {
        // (the layout follows the search heading: a launch-ahead search keeps the last layout only while its heading is within 0.1 rad
        // of the layout's -- cs_launch_distance -- and a robot that turns half a degree per scan used to lose the launch-ahead flow,
        // and 15 us, every dozen scans: 11 of 205 in bench.py's trajectory)
        if (cs->k1_last_valid && cs->k1_layout_spread) cs->k1_layout_theta = cs->k1_last_pose[2];
        k1_make_layout(cs, cs->k1_layout_groups, cs->k1_layout_target, cs->k1_layout_budget, cs->k1_layout_spread, cs->k1_layout_band_parts);
        cs->k1_layout_stale = false; cs->k1_layout_gen++;
    }
    static const int idle_cuts = env_int("SLAMHIP_K1_IDLE_CUTS", 0);      // (off: measured a LOSS, see below)
    if (!idle_cuts || !cs->k1_last_valid || cs->k1_uni_ng <= 0 || cs->n_points >= 65536 || !cs->h_scan_blob) return;
    if (cs->k1_prev_cuts_layout_gen == cs->k1_layout_gen && cs->k1_prev_cuts_points == cs->n_points && cs->k1_prev_cuts_nc == cs->k1_uni_nc &&
        cs->k1_cut_gen == cs->scan_gen && !cs->k1_prev_cuts.empty()) return;      // (made for this very scan and layout already)
    static const double cut_wkb = getenv("SLAMHIP_K1_CUT_WKB") ? atof(getenv("SLAMHIP_K1_CUT_WKB")) : 0.0;
    const double wfix = k1_cut_wfix(cs, cs->k1_last_group);
    if (!(wfix > 0.0 || cut_wkb > 0.0)) return;
    cs->k1_cut_cache.clear();
    cs->k1_cut_gen = cs->scan_gen; cs->k1_cut_layout_gen = cs->k1_layout_gen;
    k1_cut_weights(cs, cs->k1_layout_groups, cs->k1_layout_spread, wfix, cut_wkb, cs->k1_layout_budget, cs->k1_cut_wb);
    k1_cuts_make(cs, cs->k1_uni_nc, cs->k1_layout_spread, cs->k1_last_pose, cs->k1_layout_budget, cs->k1_prev_cuts);
    cs->k1_prev_cuts_nc = cs->k1_uni_nc; cs->k1_prev_cuts_layout_gen = cs->k1_layout_gen; cs->k1_prev_cuts_points = cs->n_points;
    if (!cs->k1_prev_cuts.empty()) cs->k1_cut_cache.emplace_back(cs->k1_uni_nc, cs->k1_prev_cuts);    // (a second search of THIS scan finds it too)
}

// K1 over `count` candidates in evaluation order (d_ev_idx maps to flat indices).  mode 0: d_pxcs already holds
// (px,py,c,s); 1: d_ev_off holds jitters added to `pose`; 2: d_ev_off holds poses.  The packed arg-min key of the
// launch is written to key_dst.  Asynchronous on the context's stream.
int32_t cs_launch_distance(slamhip_cs *cs, int mode, const float pose[3], int count, bool want_dist, bool cand_sane,
                           uint64_t *key_dst)
{
    slamhip_ctx *ctx = cs->ctx;
    if (cs->n_points <= 0) SH_FAIL(SLAMHIP_ERR_STATE, "no scan set (slamhip_cs_set_scan)");
    SH_TRY(cs_flush_scan(cs));
    cs->k1_launch_no = cs->launch_count;
    static const int force_global = env_int("SLAMHIP_K1_GLOBAL", 0);
    static const int verify = env_int("SLAMHIP_K1_VERIFY", 0);
    static const int tile_kb = env_int("SLAMHIP_K1_TILE_KB", 60);
    static const int target_wgs = env_int("SLAMHIP_K1_TARGET_WGS", 512);
    static const int target_wgs_uniform = env_int("SLAMHIP_K1_TARGET_WGS_UNIFORM", 768);
    static const int cpl_env = env_int("SLAMHIP_K1_CPL", 0);           // candidates per lane: 0 = by launch size
    static const int band_parts = env_int("SLAMHIP_K1_BAND_PARTS", 2);
    static const int no_table = env_int("SLAMHIP_K1_NOTABLE", 0);
    const bool sane = cs->pts_sane && cand_sane;
    const bool tiled = sane && (cs->hs % 8 == 0) && !force_global;
    if (cs->k1_prelaunch && !tiled) return CS_RC_NO_PRELAUNCH;      // (the fallback kernels read the scan's blocks on the host)
#ifdef K1_TIMES
    if (cs->k1_prelaunch) return CS_RC_NO_PRELAUNCH;               // (the developer build synchronises the stream inside this function: a prelaunched search would wait for a host that waits for it)
#endif
    const int n_rb = cs->n_rb;
    int32_t *dist = want_dist ? cs->d_dist : nullptr;
    unsigned long long *key = (unsigned long long *)key_dst;
    // a ring launch: the result word is the ring's current slot (all ones now), and the launch rests the next one
    unsigned long long *ring_slot = nullptr, *ring_reset = nullptr;
    const bool ring = cs->k1_ring_request;
    cs->k1_ring_request = false;
    if (ring) {
        if (!cs->d_k1_ring) {
            SH_HIP(hipMalloc(&cs->d_k1_ring, sizeof(uint64_t) * K1_RING_SLOTS));
            SH_HIP(hipMemsetAsync(cs->d_k1_ring, 0xFF, sizeof(uint64_t) * K1_RING_SLOTS, ctx->stream));
            cs->k1_ring_pos = 0;
        }
        ring_slot = (unsigned long long *)cs->d_k1_ring + cs->k1_ring_pos % K1_RING_SLOTS;
        ring_reset = (unsigned long long *)cs->d_k1_ring + (cs->k1_ring_pos + 1) % K1_RING_SLOTS;
        key = ring_slot;
    }
    const float bx = pose ? pose[0] : 0.f, by = pose ? pose[1] : 0.f, bth = pose ? pose[2] : 0.f;

    if (tiled) {
        const int group = mode == 1 && (cs->k1_group == K1_GROUP_BIG || cs->k1_group == K1_GROUP_SMALL) ? cs->k1_group : K1_GROUP;   // (explicit lists: always 1024)
        const int n_groups = sh_div_up(count, group);
        int budget = tile_kb * 1024;
        if (budget > 64 * 1024) budget = 64 * 1024;                // what the staging registers hold per pass
        const size_t lds = (size_t)K1_TILE_OFS + (size_t)budget;

        k1_args a;
        a.map = cs->d_hole; a.S = cs->hs; a.pts = cs->d_pts_sorted; a.ray_blk = cs->d_ray_blk; a.n_rays = cs->n_points;
        a.pxcs = cs->d_pxcs; a.src3 = cs->d_ev_off; a.bx = bx; a.by = by; a.bth = bth; a.scale = cs->hscale;
        a.count = count; a.n_groups = n_groups; a.budget = budget;
        a.band_stage = k1_band_stage();
        static const int noden = env_int("SLAMHIP_K1_NODEN", 0);
        a.noden = noden;
        a.nopad = k1_nopad();
        static const int nosplit = env_int("SLAMHIP_K1_NOSPLIT", 0);
        a.nosplit = nosplit;
        a.ev_idx = cs->d_ev_idx; a.dist_out = dist; a.key_out = key; a.verify = cs->d_verify;
        static const int no_bounds = env_int("SLAMHIP_K1_NOBOUNDS", 0);
        a.grp_bounds = (mode == 1 && !no_bounds) ? cs->d_grp_bounds : nullptr;
        a.offs_flat = cs->d_offs_flat; a.best_pose = (mode == 1 && cs->k1_want_pose) ? cs->d_best_pose : nullptr;
        cs->k1_pose_written = a.best_pose != nullptr;
        a.done_flag = cs->k1_done_flag; a.done_val = cs->k1_done_val;
        cs->k1_done_armed = a.done_flag != nullptr;
        a.sig = cs->k1_sig; a.sig_val = cs->k1_sig_val;
        cs->k1_sig_armed = a.sig != nullptr;
        a.ring_slot = ring_slot; a.ring_reset = ring_reset;
        a.scan_flag = cs->k1_prelaunch ? cs->d_scan_flag : nullptr; a.scan_seq = cs->scan_flag_seq;
        if (ring && (a.best_pose || a.done_flag || a.sig)) SH_FAIL(SLAMHIP_ERR_STATE, "a ring search delivers nothing but its key");

        // launch layout
        const bool have_spread = mode == 1 && !no_table && (int)cs->h_grp_dth.size() == n_groups;
        // a workgroup's prologue costs as much as ~25 rays of gathers: small searches get fewer, larger chunks (a dozen
        // rays or more each) rather than a full round of workgroups (measured at 4000 candidates x 400 rays: 21 -> 16 us)
        int target = n_groups <= K1_TABLE_G ? target_wgs : target_wgs_uniform;
        const long long by_work = (long long)n_groups * cs->n_points / 12;
        if (by_work < target) target = (int)(by_work > n_groups ? by_work : n_groups);
        g_cst.lap(8);
        bool remake = cs->k1_layout_dirty || cs->k1_layout_groups != n_groups || cs->k1_layout_budget != budget || cs->k1_layout_spread != have_spread ||
                      cs->k1_layout_target != target || (have_spread && !(fabsf(bth - cs->k1_layout_theta) < 0.1f));
        if (cs->k1_prelaunch) {
            // The launch precedes its scan's tables (cs_search_and_update_prelaunched): it keeps the last scan's layout, whose legality
            // for the new ray blocks the caller tests when they exist (and abandons the launch if it fails); a layout that has to be
            // remade for another reason needs the new blocks: no prelaunch.
            if (remake) { cs->k1_ring_request = ring; return CS_RC_NO_PRELAUNCH; }
            cs->k1_layout_stale = true;
        } else
        if (!remake && cs->k1_scan_dirty) {
            // A new scan under an unchanged candidate list (the per-scan flow): the layout made for the last scan serves this one
            // if its counts of ray ranges are legal for the new ray blocks -- only the balance of the launch depends on the layout,
            // and consecutive scans look alike -- and the one for THIS scan is made while the host waits for the search
            // (cs_layout_idle_refresh), for the next scan's launch: 8 us of estimate left the host's critical path between two
            // scans (`CoreSLAMProcessor.Update` 66.6 -> 60 us; SLAMHIP_K1_LAYOUT_SYNC=1 makes every scan's layout before its launch).
            static const int layout_sync = env_int("SLAMHIP_K1_LAYOUT_SYNC", 0);
            if (layout_sync || !k1_layout_legal(cs)) remake = true;
            else cs->k1_layout_stale = true;
        }
        cs->k1_scan_dirty = false;
        if (remake) {
            cs->k1_layout_theta = bth;
            k1_make_layout(cs, n_groups, target, budget, have_spread, band_parts);
            cs->k1_layout_dirty = false; cs->k1_layout_stale = false; cs->k1_layout_gen++;
            cs->k1_layout_groups = n_groups; cs->k1_layout_budget = budget; cs->k1_layout_spread = have_spread; cs->k1_layout_target = target;
            cs->k1_layout_band_parts = band_parts;
        }
        g_cst.lap(9);
        {   // the accumulators count up to 2^14 - 1 arrivals per candidate and sum up to 2^20 rays (pathological scans: fallback kernels)
            int nc_max = cs->k1_uni_nc;
            for (size_t i = 0; i < cs->k1_tab_nc.size(); i++) nc_max = std::max(nc_max, cs->k1_tab_nc[i]);
            if (nc_max >= (1 << (K1_ACC_ARRIVED - K1_ACC_INMAP)) || cs->n_points >= (1 << 20)) goto fallback;
        }
        static const int dump = env_int("SLAMHIP_K1_DUMP", 0);
        if (dump) {                                                // debugging aid: the launch layout and its cost estimates
            fprintf(stderr, "[slamhip] K1 layout: %d groups, %zu listed, uniform [%d, %d) x %d chunks (slowest range %.0f ray units with 8 per step; %d ray blocks)\n", n_groups, cs->k1_tab_group.size(),
                    cs->k1_uni_g0, cs->k1_uni_g0 + cs->k1_uni_ng, cs->k1_uni_nc, k1_chunks_score(cs, cs->k1_uni_nc, 8.0), cs->n_rb);
            for (size_t i = 0; i < cs->k1_tab_group.size(); i++) {
                const int g = cs->k1_tab_group[i];
                fprintf(stderr, "   group %3d: chunks %3d, dtheta %.4f rad, spread %.1f px, cost %.0f ray units\n", g, cs->k1_tab_nc[i],
                        have_spread ? cs->h_grp_dth[(size_t)g] : 0.f, have_spread ? cs->h_grp_dxy[(size_t)g] : 0.f,
                        have_spread ? k1_group_cost(cs, cs->k1_terms, g, budget) : 0.0);
            }
        }
        // Ray ranges cut by cost (k1_balanced_cuts), remade when the scan, a count of ranges or the weight changed: the uniform
        // part's and every listed group's (groups with the same count of ranges share the cut).  A cut may come out with fewer
        // ranges than asked for: the group then runs with that many workgroups.
        // The weight of a tile step in ray units: 26 where it was calibrated (2048^2 map = 51.2 pixels per metre, two candidates per
        // lane), less on coarser maps (smaller tiles, and the 64-ray cap makes blocks long in rays: 1024^2 wants about half, 256^2
        // none) and in proportion to what a ray costs the workgroup (four candidates per lane: a ray takes twice as long, the step
        // does not).  SLAMHIP_K1_CUT_WFIX = 0 and SLAMHIP_K1_CUT_WKB = 0: the equal-count formula everywhere.
        static const double cut_wkb = getenv("SLAMHIP_K1_CUT_WKB") ? atof(getenv("SLAMHIP_K1_CUT_WKB")) : 0.0;
        const double cut_wfix = k1_cut_wfix(cs, group);
        static const int cut_tab = env_int("SLAMHIP_K1_CUT_TAB", 0);      // (the listed groups too: measured slower)
        const int n_tab = (int)cs->k1_tab_group.size();
        const float pose3[3] = { bx, by, bth };
        // The cuts cost the host ~7 us (mostly the exact box test of the pieces, k1_cuts_banded_rays) and buy a launch 1 - 2 us.  They
        // are made (a) when a scan is searched for the SECOND time under one layout -- a list searched from many poses, a benchmark
        // loop -- and (b), round 6, for every scan of the per-scan flow while the host waits for that scan's pose
        // (cs_layout_idle_refresh: idle time), to be used by the NEXT scan's launch if they are legal for its ray blocks --
        // consecutive scans look alike, and a cut only balances the launch.  Never on the host's critical path between two scans
        // (measured there: CoreSLAMProcessor.Update 66 -> 77 us per scan).  SLAMHIP_K1_CUT_ALWAYS=1: always; SLAMHIP_K1_IDLE_CUTS=0: not (b).
        static const int cut_always = env_int("SLAMHIP_K1_CUT_ALWAYS", 0);
        const bool cut_repeat = cut_always || (cs->k1_cut_seen_scan == cs->scan_gen && cs->k1_cut_seen_layout == cs->k1_layout_gen);
        cs->k1_cut_seen_scan = cs->scan_gen; cs->k1_cut_seen_layout = cs->k1_layout_gen;
        const bool cuts_on = cut_repeat && (cut_wfix > 0.0 || cut_wkb > 0.0) && cs->n_points < 65536;
        if (cuts_on && (cs->k1_cut_gen != cs->scan_gen || cs->k1_cut_layout_gen != cs->k1_layout_gen)) {
            cs->k1_cut_cache.clear();
            cs->k1_cut_gen = cs->scan_gen; cs->k1_cut_layout_gen = cs->k1_layout_gen;
            const double tw0 = k1_now_us();
            k1_cut_weights(cs, n_groups, have_spread, cut_wfix, cut_wkb, budget, cs->k1_cut_wb);
            g_cut_t[0] += k1_now_us() - tw0;
        }
        cs->k1_cut_cache.reserve(K1_TABLE_G + 8);                      // (the entries' addresses are held below: no reallocation)
        auto cuts_for = [&](int nrc) -> const std::vector<int> * {       // nullptr: no cut for this count (the formula stays)
            if (!cuts_on || nrc < 2) return nullptr;
            for (auto &e : cs->k1_cut_cache) if (e.first == nrc) return e.second.empty() ? nullptr : &e.second;
            cs->k1_cut_cache.emplace_back(nrc, std::vector<int>());
            std::vector<int> &c = cs->k1_cut_cache.back().second;
            k1_cuts_make(cs, nrc, have_spread, pose3, budget, c);
            return c.empty() ? nullptr : &c;
        };
        static const int cut_times = env_int("SLAMHIP_K1_CUT_TIMES", 0);     // developer aid: host time of the cuts, printed every 64 makes
        timespec ct0; if (cut_times) clock_gettime(CLOCK_MONOTONIC, &ct0);
        a.uni_cut = 0; a.tab_cut = 0;
        int n_cut = 0;                                                 // entries of a.cut in use
        a.uni_g0 = cs->k1_uni_g0; a.uni_ng = cs->k1_uni_ng > 0 ? cs->k1_uni_ng : 1; a.uni_nc = cs->k1_uni_nc;
        cs->k1_launch_prev_cuts = false;
        if (cs->k1_uni_ng > 0) {
            const std::vector<int> *c = cuts_for(cs->k1_uni_nc);
            if (!c && !cuts_on && mode == 1 && !cs->k1_prev_cuts.empty() && cs->k1_prev_cuts_nc == cs->k1_uni_nc && cs->k1_prev_cuts_layout_gen == cs->k1_layout_gen &&
                cs->k1_prev_cuts_points == cs->n_points && (cs->k1_prelaunch || k1_cuts_legal(cs, cs->k1_prev_cuts))) {
                // the cut made for the scan before (idle refresh): legal for this scan's blocks -- or, for a launch that precedes its
                // scan's tables, tested when they exist (cs_k1_layout_legal: the launch is abandoned if it is not)
                c = &cs->k1_prev_cuts;
                cs->k1_launch_prev_cuts = true;
            }
            if (c && (int)c->size() <= K1_MAXCUT) {
                for (size_t i = 0; i < c->size(); i++) a.cut[i] = (unsigned short)(*c)[i];
                n_cut = (int)c->size();
                a.uni_cut = 1; a.uni_nc = (int)c->size() - 1;
            }
        }
        if (cut_times) {
            static thread_local double acc = 0.0; static thread_local int nacc = 0;   // (per host thread, like g_cut_t: a group drives one thread per GPU)
            timespec ct1; clock_gettime(CLOCK_MONOTONIC, &ct1);
            acc += (double)(ct1.tv_sec - ct0.tv_sec) * 1e6 + (double)(ct1.tv_nsec - ct0.tv_nsec) * 1e-3;
            if (++nacc == 64) {
                fprintf(stderr, "[slamhip] K1 cuts: %.2f us of host time per launch (uniform part): weights %.2f | balanced cuts %.2f | banded check %.2f\n", acc / nacc,
                        g_cut_t[0] / nacc, g_cut_t[1] / nacc, g_cut_t[2] / nacc);
                acc = 0.0; nacc = 0; g_cut_t[0] = g_cut_t[1] = g_cut_t[2] = 0.0;
            }
        }
        // the listed groups: all of them from the table, or none
        std::vector<const std::vector<int> *> tcut((size_t)n_tab, nullptr);
        bool tab_ok = cut_tab && n_tab > 0;
        long long tab_wgs = 0;
        for (int p = 0; p < n_tab && tab_ok; p++) {
            const int nbp = cs->k1_tab_nbp[(size_t)p] > 0 ? cs->k1_tab_nbp[(size_t)p] : 1;
            tcut[(size_t)p] = cuts_for(cs->k1_tab_nc[(size_t)p] / nbp);
            if (!tcut[(size_t)p]) tab_ok = false; else tab_wgs += (long long)(tcut[(size_t)p]->size() - 1) * nbp;
        }
        if (tab_ok && n_cut + tab_wgs > K1_MAXCUT) tab_ok = false;
        unsigned first = 0;
        for (int p = 0; p < n_tab; p++) {
            k1_args::tab_rec &rec = a.tab[p];
            const int nbp = cs->k1_tab_nbp[(size_t)p] > 0 ? cs->k1_tab_nbp[(size_t)p] : 1;
            const int nc_p = tab_ok ? (int)(tcut[(size_t)p]->size() - 1) * nbp : cs->k1_tab_nc[(size_t)p];
            rec.group = (unsigned short)cs->k1_tab_group[(size_t)p];
            rec.nbp = (unsigned short)cs->k1_tab_nbp[(size_t)p];
            rec.first = (unsigned short)first; rec.nc = (unsigned short)nc_p;
            if (tab_ok)
                for (int w = 0; w < nc_p; w++) a.cut[n_cut + (int)first + w] = (unsigned short)(*tcut[(size_t)p])[(size_t)(w / nbp)];
            first += (unsigned)nc_p;
        }
        if (tab_ok) a.tab_cut = n_cut;
        if (dump && a.uni_cut) {
            fprintf(stderr, "   uniform ranges cut by cost (%d of %d asked for; ray blocks start at", a.uni_nc, cs->k1_uni_nc);
            for (int b = 0; b <= cs->n_rb; b++) fprintf(stderr, " %d", cs->h_rb_start[(size_t)b]);
            fprintf(stderr, "):");
            for (int c = 0; c <= a.uni_nc; c++) fprintf(stderr, " %d", (int)a.cut[c]);
            fprintf(stderr, "\n");
        }
        for (int p = 0; p < n_tab; p++)
            for (unsigned w = a.tab[p].first; w < (unsigned)a.tab[p].first + a.tab[p].nc; w++) a.wg_pos[w] = (unsigned char)p;
        a.n_tab_wgs = (int)first;
        const int n_wgs = (int)first + a.uni_nc * cs->k1_uni_ng;
        // candidates per lane: 2 (512 lanes, 8 waves) measured best or equal from 16k to 256k candidates on MI355X;
        // 1 (16 waves: slow start) and 4 (4 waves: the VALU starves at 2 waves / SIMD) stay selectable for experiments
        const int cpl = cpl_env == 1 || cpl_env == 4 ? cpl_env : 2;
        if (sh_div_up(count, K1_GROUP) + 2 > cs->k1_cap_groups) {
            if (cs->d_k1_acc) (void)hipFree(cs->d_k1_acc);
            cs->d_k1_acc = nullptr; cs->k1_cap_groups = 0;
            const int cap = sh_div_up(count, K1_GROUP) + sh_div_up(count, K1_GROUP) / 4 + 16;
            SH_HIP(hipMalloc(&cs->d_k1_acc, sizeof(unsigned long long) * (size_t)cap * K1_GROUP));
            SH_HIP(hipMemsetAsync(cs->d_k1_acc, 0, sizeof(unsigned long long) * (size_t)cap * K1_GROUP, ctx->stream));
            cs->k1_cap_groups = cap;
        }
        if (!cs->d_k1_gmin) {
            // the running minimum (all ones at rest) and the count of finished candidates (zero at rest): the launch's last
            // finisher leaves them so
            SH_HIP(hipMalloc(&cs->d_k1_gmin, 16));
            SH_HIP(hipMemsetAsync(cs->d_k1_gmin, 0xff, 8, ctx->stream));
            SH_HIP(hipMemsetAsync((char *)cs->d_k1_gmin + 8, 0, 8, ctx->stream));
        }
        a.gmin = cs->d_k1_gmin; a.done = (unsigned *)((char *)cs->d_k1_gmin + 8); a.acc = cs->d_k1_acc;
        if (mode == 1) { cs->k1_last_pose[0] = bx; cs->k1_last_pose[1] = by; cs->k1_last_pose[2] = bth; cs->k1_last_group = group; cs->k1_last_valid = true; }
        static const int no_lat = env_int("SLAMHIP_K1_NO_LATTICE", 0);       // (a lattice list through the ordinary kernel: same results, for comparison)
        const bool lat2 = mode == 1 && !verify && !no_lat && cs->k1_lattice == 2 && group == K1_GROUP && cpl == 2;
        const bool lat4 = mode == 1 && !verify && !no_lat && cs->k1_lattice == 4 && group == K1_GROUP_BIG;
        g_cst.lap(1);
        SH_TRY(cs_side_join(cs));
        // ---- the plan (k1_plan): one launch per search on a stream of its own, beside the search ---------------------------------
        // Made for every search with jitter bounds (mode 1) whose inputs are known to be in memory: the plan launch is not ordered
        // behind the operator's stream, so after a candidate gather or a scan upload in that stream (plan_inputs_after) it waits
        // until a search launch behind them has STARTED (the started word).  A slot of the plan buffers is reused K1_PLAN_SLOTS
        // plans later, when the search that read it has finished -- i.e. a later search launch has started; the host waits for
        // that if it is that far ahead of the device (it then is in nobody's way).  SLAMHIP_K1_PLAN=0: never.
        static const int plan_env = env_int("SLAMHIP_K1_PLAN", 1);
        volatile uint32_t *h_started = (volatile uint32_t *)cs->h_key + 25;
        a.started = (uint32_t *)cs->h_key + 25; a.launch_no = ++cs->k1_launches;
        a.plan_rec = nullptr; a.plan_seq = 0;
        // Which launches get one: those that will WAIT in the stream -- the search launch before this one has not even started
        // (the started word), so this one is at least a whole search away from running and its plan has time to arrive: the
        // throughput forms (enqueue-only searches, the batched all-reduce form) once the host runs ahead of the device.  A search
        // that starts at once -- a blocking call, the first launches behind a synchronise, the per-scan flows, whose search follows
        // its scan -- gets none: its plan would arrive late and cost the host a launch (measured: a blocking search 27.8 -> 31.6 us
        // per call with one, the fused scan in the ordinary order 50 -> 54.5 us per scan).  Nor does a search launched ahead of its
        // scan's tables: the tile steps need the tables.  (Measured and dropped in round 6: the candidates' (px, py, c, s) made once per
        // search -- by the plan launch, or shared between a group's workgroups inside the search launch through memory -- instead of
        // by every workgroup: reading them back costs a launch what the trigonometry does, 15.2 us with or without at the headline size.)
        static const int plan_always = env_int("SLAMHIP_K1_PLAN_ALWAYS", 0);   // (tests: a plan for every eligible launch -- most arrive late, the race the stamps are for)
        const bool stream_busy = plan_always || (int32_t)(*h_started - (a.launch_no - 1)) < 0;
        bool plan_on = plan_env && mode == 1 && stream_busy && a.grp_bounds != nullptr && !cs->k1_prelaunch && !lat2 && !lat4 && !ctx->mail_off;
        const int n_plan_wgs = n_wgs;
        if (plan_on && cs->plan_inputs_after != 0) {
            if ((int32_t)(*h_started - cs->plan_inputs_after) >= 0) cs->plan_inputs_after = 0;
            else { plan_on = false; cs->plan_stats[3]++; }
        }
        if (plan_on) {
            if (n_wgs > cs->plan_cap_wgs) {
                SH_HIP(hipStreamSynchronize(ctx->stream));          // (searches in flight read the buffers)
                const int cw = std::max(cs->plan_cap_wgs, n_wgs + n_wgs / 4 + 64);
                cs_plan_free(cs);
                for (int i = 0; i < K1_PLAN_SLOTS; i++) {
                    SH_HIP(hipMalloc(&cs->d_plan_rec[i], sizeof(uint32_t) * K1_PLAN_REC_WORDS * (size_t)cw));
                    SH_HIP(hipMemsetAsync(cs->d_plan_rec[i], 0, sizeof(uint32_t) * K1_PLAN_REC_WORDS * (size_t)cw, cs->plan_stream));
                }
                SH_HIP(hipStreamSynchronize(cs->plan_stream));
                cs->plan_cap_wgs = cw;
            }
            const unsigned slot = cs->plan_count % K1_PLAN_SLOTS;
            const uint32_t user = cs->plan_slot_user[slot];
            if (user != 0 && (int32_t)(*h_started - (user + 1)) < 0) {
                // The host is K1_PLAN_SLOTS - 1 searches ahead of the device: it waits for the slot -- backpressure, the device has
                // work queued -- but never longer than 20 ms or the context's wait bound; past that the search goes without a plan
                // (nothing fails because of a plan).
                cs->plan_stats[2]++;
                const int64_t bound_us = 1000 * (ctx->wait_timeout_ms > 0 && ctx->wait_timeout_ms < 20 ? ctx->wait_timeout_ms : 20);
                const double t0w = k1_now_us();
                for (int spins = 0; (int32_t)(*h_started - (user + 1)) < 0; spins++) {
                    __builtin_ia32_pause();
                    if ((spins & 1023) == 1023 && k1_now_us() - t0w > (double)bound_us) break;
                }
                if ((int32_t)(*h_started - (user + 1)) < 0) { plan_on = false; cs->plan_stats[3]++; }
            }
        }
        if (plan_on) {
            const unsigned slot = cs->plan_count % K1_PLAN_SLOTS;
            if (++cs->plan_seq == 0) cs->plan_seq = 1;
            a.plan_rec = cs->d_plan_rec[slot]; a.plan_seq = cs->plan_seq;
            const dim3 pgrid((unsigned)n_plan_wgs);
            if (group == K1_GROUP_BIG) hipLaunchKernelGGL((k1_plan<K1_GROUP_BIG, 4>), pgrid, dim3(64), 0, cs->plan_stream, a, n_plan_wgs);
            else if (group == K1_GROUP_SMALL) hipLaunchKernelGGL((k1_plan<K1_GROUP_SMALL, 1>), pgrid, dim3(64), 0, cs->plan_stream, a, n_plan_wgs);
            else if (cpl == 4) hipLaunchKernelGGL((k1_plan<K1_GROUP, 4>), pgrid, dim3(64), 0, cs->plan_stream, a, n_plan_wgs);
            else if (cpl == 2) hipLaunchKernelGGL((k1_plan<K1_GROUP, 2>), pgrid, dim3(64), 0, cs->plan_stream, a, n_plan_wgs);
            else hipLaunchKernelGGL((k1_plan<K1_GROUP, 1>), pgrid, dim3(64), 0, cs->plan_stream, a, n_plan_wgs);
            SH_HIP(hipGetLastError());
            cs->plan_slot_user[slot] = a.launch_no;
            cs->plan_count++; cs->plan_stats[0]++;
        }
        if (!plan_on) cs->plan_stats[1]++;
        g_cst.lap(2);
        {
            sh_timer t(ctx, SLAMHIP_K_CS_DISTANCE);
#define K1_LAUNCH(M, V, C, G) hipLaunchKernelGGL((k1_search_tiled<M, V, C, G>), dim3(n_wgs), dim3(G / C), lds, ctx->stream, a)
#define K1_LAUNCH_LAT(C, G) hipLaunchKernelGGL((k1_search_tiled<1, false, C, G, true>), dim3(n_wgs), dim3(G / C), lds, ctx->stream, a)
#define K1_LAUNCH_C(M, V) { if (group == K1_GROUP_BIG) K1_LAUNCH(M, V, 4, K1_GROUP_BIG); else if (group == K1_GROUP_SMALL) K1_LAUNCH(M, V, 1, K1_GROUP_SMALL); else if (cpl == 4) K1_LAUNCH(M, V, 4, K1_GROUP); else if (cpl == 2) K1_LAUNCH(M, V, 2, K1_GROUP); else K1_LAUNCH(M, V, 1, K1_GROUP); }
            if (lat2) K1_LAUNCH_LAT(2, K1_GROUP);
            else if (lat4) K1_LAUNCH_LAT(4, K1_GROUP_BIG);
            else if (verify) { if (mode == 0) K1_LAUNCH_C(0, true) else if (mode == 1) K1_LAUNCH_C(1, true) else K1_LAUNCH_C(2, true) }
            else        { if (mode == 0) K1_LAUNCH_C(0, false) else if (mode == 1) K1_LAUNCH_C(1, false) else K1_LAUNCH_C(2, false) }
#undef K1_LAUNCH_C
#undef K1_LAUNCH_LAT
#undef K1_LAUNCH
        }
        SH_HIP(hipGetLastError());
        if (ring) { cs->k1_ring_last = (uint64_t *)ring_slot; cs->k1_ring_pos++; }
#ifdef K1_TIMES
        {
            static thread_local int calls = 0;
            if (++calls == 8) {
                (void)hipStreamSynchronize(ctx->stream);
                const int nw = n_wgs < 4096 ? n_wgs : 4096;
                std::vector<unsigned long long> h((size_t)nw * 16);
                (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_k1_times), sizeof(unsigned long long) * h.size());
                unsigned long long t0 = ~0ull, t1 = 0;
                auto endof = [&](int i) { unsigned long long e = h[i * 16 + 8]; if (h[i * 16 + 9] > e) e = h[i * 16 + 9]; return e; };
                for (int i = 0; i < nw; i++) { if (h[i * 16] < t0) t0 = h[i * 16]; if (endof(i) > t1) t1 = endof(i); }
                static const char *nm[9] = { "q", "wred", "bnd", "boxes", "steps", "tile", "compute", "publish", "tail" };
                double acc[9] = { 0 }; int nlast = 0;
                for (int i = 0; i < nw; i++) {
                    for (int k = 0; k < 8; k++) acc[k] += (double)(h[i * 16 + k + 1] - h[i * 16 + k]) * 0.01;
                    if (h[i * 16 + 9] > h[i * 16 + 8]) { acc[8] += (double)(h[i * 16 + 9] - h[i * 16 + 8]) * 0.01; nlast++; }
                }
                fprintf(stderr, "[k1 times] WGs %d (groups %d: %d listed, %d x %d uniform; %d candidates per lane) span %.2f us; mean per WG:", n_wgs, n_groups, n_tab, cs->k1_uni_ng, cs->k1_uni_nc, cpl, (double)(t1 - t0) * 0.01);
                for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.2f |", nm[k], acc[k] / nw);
                fprintf(stderr, " reducers (%d) %.2f us\n", nlast, nlast ? acc[8] / nlast : 0.0);
                {
                    std::vector<unsigned long long> ws((size_t)nw * 16);
                    (void)hipMemcpyFromSymbol(ws.data(), HIP_SYMBOL(g_k1_wstart), sizeof(unsigned long long) * ws.size());
                    double skew = 0, mx = 0;
                    for (int i = 0; i < nw; i++) {
                        unsigned long long lo = ~0ull, hi = 0;
                        for (int w = 0; w < (group != K1_GROUP ? 8 : K1_GROUP / cpl / 64); w++) { lo = std::min(lo, ws[i * 16 + w]); hi = std::max(hi, ws[i * 16 + w]); }
                        skew += (double)(hi - lo) * 0.01; mx = std::max(mx, (double)(hi - lo) * 0.01);
                    }
                    fprintf(stderr, "[k1 times] wave start skew inside a workgroup: mean %.2f us, max %.2f us\n", skew / nw, mx);
                }
                std::vector<int> order((size_t)nw);
                for (int i = 0; i < nw; i++) order[(size_t)i] = i;
                std::sort(order.begin(), order.end(), [&](int x, int y) { return endof(x) > endof(y); });
                for (int oi = 0; oi < nw; oi += (oi < 12 ? 1 : nw / 16 > 0 ? nw / 16 : 1)) {
                    const int i = order[(size_t)oi];
                    fprintf(stderr, "  wg %4d: start +%6.2f |", i, (double)(h[i * 16] - t0) * 0.01);
                    for (int k = 0; k < 8; k++) fprintf(stderr, " %s %5.2f", nm[k], (double)(h[i * 16 + k + 1] - h[i * 16 + k]) * 0.01);
                    fprintf(stderr, " tail %5.2f | end +%6.2f | g %2d nc %2d rays shared %d global %d band %d",
                            h[i * 16 + 9] > h[i * 16 + 8] ? (double)(h[i * 16 + 9] - h[i * 16 + 8]) * 0.01 : 0.0,
                            (double)(endof(i) - t0) * 0.01, (int)h[i * 16 + 14], (int)h[i * 16 + 15], (int)h[i * 16 + 11],
                            (int)h[i * 16 + 12], (int)h[i * 16 + 13]);
                    {
                        unsigned long long sbx[8];
                        (void)hipMemcpyFromSymbol(sbx, HIP_SYMBOL(g_k1_sub), sizeof(sbx), sizeof(unsigned long long) * (size_t)i * 8);
                        fprintf(stderr, " | steps");
                        for (int k = 5; k < 8; k++) if (sbx[k]) fprintf(stderr, " [kind %d tile %d x %d, %d rays]", (int)(sbx[k] & 15), (int)((sbx[k] >> 4) & 0xfff), (int)((sbx[k] >> 16) & 0xffff), (int)(sbx[k] >> 32));
                        fprintf(stderr, "\n");
                    }
                }
                {   // per CU: workgroups hosted and the time the last of them ends
                    std::vector<unsigned long long> ws((size_t)nw * 16);
                    (void)hipMemcpyFromSymbol(ws.data(), HIP_SYMBOL(g_k1_wstart), sizeof(unsigned long long) * ws.size());
                    std::vector<int> cnt(16 * 256, 0); std::vector<double> cend(16 * 256, 0.0), csum(16 * 256, 0.0);
                    for (int i = 0; i < nw; i++) {
                        const int cu = (int)(((ws[i * 16 + 15] >> 16) & 15) * 256 + (ws[i * 16 + 15] & 255));
                        cnt[cu]++; cend[cu] = std::max(cend[cu], (double)(endof(i) - t0) * 0.01);
                        csum[cu] += (double)(h[i * 16 + 7] - h[i * 16 + 6]) * 0.01;
                    }
                    int ncu = 0, hist[8] = { 0 }; double emean = 0, emax = 0, emin = 1e9;
                    for (int c = 0; c < 16 * 256; c++) if (cnt[c]) { ncu++; hist[cnt[c] < 7 ? cnt[c] : 7]++; emean += cend[c]; emax = std::max(emax, cend[c]); emin = std::min(emin, cend[c]); }
                    fprintf(stderr, "[k1 times] CUs used %d; workgroups per CU: 1:%d 2:%d 3:%d 4+:%d; CU end time mean %.2f min %.2f max %.2f us\n",
                            ncu, hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7], emean / std::max(ncu, 1), emin, emax);
                    // pairs sharing a CU (workgroups i and i + CUs): compute intervals and a solo estimate (half of the overlap each)
                    int same = 0;
                    for (int i = 0; i + 256 < nw; i++) if (ws[i * 16 + 15] == ws[(i + 256) * 16 + 15]) same++;
                    fprintf(stderr, "[k1 times] pairs (i, i+256) on the same CU: %d\n", same);
                    for (int i = 0; i + 256 < nw; i++) {
                        const int j = i + 256;
                        const double sa = (double)(h[i * 16 + 6] - t0) * 0.01, ea = (double)(h[i * 16 + 7] - t0) * 0.01;
                        const double sb = (double)(h[j * 16 + 6] - t0) * 0.01, eb = (double)(h[j * 16 + 7] - t0) * 0.01;
                        const double ov = std::max(0.0, std::min(ea, eb) - std::max(sa, sb));
                        fprintf(stderr, "PAIR %3d g %2d sh %3d gl %3d bd %3d comp %5.2f solo %5.2f | %3d g %2d sh %3d gl %3d bd %3d comp %5.2f solo %5.2f | end %5.2f\n",
                                i, (int)h[i * 16 + 14], (int)h[i * 16 + 11], (int)h[i * 16 + 12], (int)h[i * 16 + 13], ea - sa, ea - sa - ov / 2,
                                j, (int)h[j * 16 + 14], (int)h[j * 16 + 11], (int)h[j * 16 + 12], (int)h[j * 16 + 13], eb - sb, eb - sb - ov / 2,
                                std::max((double)(endof(i) - t0), (double)(endof(j) - t0)) * 0.01);
                    }
                }
                {   // inside the compute phase (wave 0 of every workgroup): staging of the steps after the first, prefetch issue, gather loops
                    std::vector<unsigned long long> sb((size_t)nw * 8);
                    (void)hipMemcpyFromSymbol(sb.data(), HIP_SYMBOL(g_k1_sub), sizeof(unsigned long long) * sb.size());
                    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, mhz = 0; int nm_ = 0;
                    for (int i = 0; i < nw; i++) {
                        a0 += (double)sb[i * 8] * 0.01; a1 += (double)sb[i * 8 + 1] * 0.01; a2 += (double)sb[i * 8 + 2] * 0.01; a3 += (double)sb[i * 8 + 3];
                        const double comp = (double)(h[i * 16 + 7] - h[i * 16 + 6]) * 0.01;
                        if (comp > 1.0) { mhz += (double)sb[i * 8 + 4] / (comp + (double)(h[i * 16 + 6] - h[i * 16 + 5]) * 0.01); nm_++; }
                    }
                    fprintf(stderr, "[k1 times] compute phase, mean per WG: steps %.2f | restaging %.2f us | prefetch issue %.2f us | gather loops %.2f us | shader clock ~%.0f MHz (steps incl. first staging)\n",
                            a3 / nw, a0 / nw, a1 / nw, a2 / nw, nm_ ? mhz / nm_ : 0.0);
                    for (int oi = 0; oi < nw; oi += nw / 24 > 0 ? nw / 24 : 1) {
                        const int i = order[(size_t)oi];
                        fprintf(stderr, "  wg %4d g %2d: steps %d restaging %5.2f prefetch %5.2f loops %5.2f | rays shared %d global %d band %d\n", i, (int)h[i * 16 + 14], (int)sb[i * 8 + 3],
                                (double)sb[i * 8] * 0.01, (double)sb[i * 8 + 1] * 0.01, (double)sb[i * 8 + 2] * 0.01, (int)h[i * 16 + 11], (int)h[i * 16 + 12], (int)h[i * 16 + 13]);
                    }
                }
                // per group: chunks, ray-steps per kind, mean / max compute time
                for (int g = 0; g < n_groups; g++) {
                    double cs_ = 0, cm = 0; int n = 0; long long k4[4] = { 0, 0, 0, 0 };
                    for (int i = 0; i < nw; i++) if ((int)h[i * 16 + 14] == g) {
                        const double c = (double)(h[i * 16 + 7] - h[i * 16 + 6]) * 0.01;
                        cs_ += c; cm = std::max(cm, c); n++;
                        for (int k = 0; k < 4; k++) k4[k] += (long long)h[i * 16 + 10 + k];
                    }
                    fprintf(stderr, "  group %2d: chunks %2d | ray-steps own %lld shared %lld global %lld band %lld | compute mean %.2f max %.2f us\n",
                            g, n, k4[0], k4[1], k4[2], k4[3], n ? cs_ / n : 0.0, cm);
                }
            }
        }
#endif
        return SLAMHIP_OK;
    }

fallback:
    // ---- fallback: candidate transform, bounds-checked global gathers, reduction -----------------------------------
    cs->k1_pose_written = false; cs->k1_done_armed = false; cs->k1_sig_armed = false;
    SH_TRY(cs_side_join(cs));
    {
        sh_timer t(ctx, SLAMHIP_K_CS_PREP);
        const dim3 grid(sh_div_up(count, K1_THREADS));
#define K1_PREP(M) hipLaunchKernelGGL(k1_prep_pxcs<M>, grid, dim3(K1_THREADS), 0, ctx->stream, (const float *)cs->d_ev_off, \
                       bx, by, bth, cs->hscale, cs->d_pxcs, count, key)
        if (mode == 0) K1_PREP(0); else if (mode == 1) K1_PREP(1); else K1_PREP(2);
#undef K1_PREP
    }
    int bpc = (int)(((long long)sh_div_up(count, K1_THREADS) * n_rb) / 4096);
    if (bpc < 1) bpc = 1;
    if (bpc > n_rb) bpc = n_rb;
    const int n_chunks = sh_div_up(n_rb, bpc);
    SH_TRY(ensure_partial(cs, sizeof(uint2) * (size_t)n_chunks * count));
    {
        sh_timer t(ctx, SLAMHIP_K_CS_DISTANCE);
        dim3 grid(sh_div_up(count, K1_THREADS), n_chunks);
        if (sane)
            hipLaunchKernelGGL(k1_distance_global<false>, grid, dim3(K1_THREADS), 0, ctx->stream, cs->d_hole, cs->hs,
                               cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, (uint2 *)cs->d_partial);
        else
            hipLaunchKernelGGL(k1_distance_global<true>, grid, dim3(K1_THREADS), 0, ctx->stream, cs->d_hole, cs->hs,
                               cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, (uint2 *)cs->d_partial);
    }
    {
        sh_timer t(ctx, SLAMHIP_K_CS_REDUCE);
        const int rblocks = sh_div_up(count, 64) < 512 ? sh_div_up(count, 64) : 512;
        hipLaunchKernelGGL(k1_reduce, dim3(rblocks), dim3(256), 0, ctx->stream, (const uint2 *)cs->d_partial, n_chunks,
                           count, cs->n_points, cs->d_ev_idx, dist, key);
    }
    SH_HIP(hipGetLastError());
    if (ring) {                                                    // (the fallback's first kernel arms its own key; the next slot is rested by a fill)
        SH_HIP(hipMemsetAsync(ring_reset, 0xFF, sizeof(uint64_t), ctx->stream));
        cs->k1_ring_last = (uint64_t *)ring_slot; cs->k1_ring_pos++;
    }
    return SLAMHIP_OK;
}
