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
#define K1_CPT 4
#define K1_GROUP (K1_THREADS * K1_CPT)
#define K1_CTRL_BYTES 1024            // control block at the start of dynamic LDS (bounds + boxes)
#define K1_MAX_RB 32

// ---- candidate preparation -------------------------------------------------------------------------
// pose_k = search_pose + offs_k (:635-637); (px,py,c,s) per :232-235 with deterministic trig.
// Thread 0 also arms the arg-min key.
__global__ void __launch_bounds__(256)
k1_prep_offsets(const float *__restrict__ ev_off, int count, float bx, float by, float bth, float scale,
                float4 *__restrict__ pxcs, unsigned long long *__restrict__ key)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) *key = ~0ull;
    if (j >= count) return;
    float x = bx + ev_off[3 * j + 0];
    float y = by + ev_off[3 * j + 1];
    float th = bth + ev_off[3 * j + 2];
    float s, c;
    sh_det_sincosf(th, &s, &c);
    float4 q;
    q.x = x * scale + 0.5f;
    q.y = y * scale + 0.5f;
    q.z = c * scale;
    q.w = s * scale;
    pxcs[j] = q;
}

__global__ void __launch_bounds__(256)
k1_prep_poses(const float *__restrict__ poses, int count, float scale, float4 *__restrict__ pxcs,
              unsigned long long *__restrict__ key)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) *key = ~0ull;
    if (j >= count) return;
    float s, c;
    sh_det_sincosf(poses[3 * j + 2], &s, &c);
    float4 q;
    q.x = poses[3 * j + 0] * scale + 0.5f;
    q.y = poses[3 * j + 1] * scale + 0.5f;
    q.z = c * scale;
    q.w = s * scale;
    pxcs[j] = q;
}

__global__ void k1_arm_key(unsigned long long *key) { *key = ~0ull; }

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

// ---- K1 main, global-gather form (fallback: unsafe inputs or map sides that are not a multiple of 8) ------
template <bool SAFE>
__global__ void __launch_bounds__(K1_THREADS)
k1_distance_global(const uint16_t *__restrict__ map, int S, const float2 *__restrict__ pts,
                   const int *__restrict__ rb_start, int n_rb, int blocks_per_chunk,
                   const float4 *__restrict__ pxcs, int count, uint2 *__restrict__ partial)
{
    const int j = blockIdx.x * K1_THREADS + threadIdx.x;
    const int chunk = blockIdx.y;
    const int b0 = chunk * blocks_per_chunk;
    const int b1 = b0 + blocks_per_chunk < n_rb ? b0 + blocks_per_chunk : n_rb;
    const int r0 = rb_start[b0], r1 = rb_start[b1];
    const float4 q = pxcs[j < count ? j : count - 1];
    uint32_t sum = 0, cnt = 0;
    for (int r = r0; r < r1; r++) k1_gather_global<SAFE>(map, S, q, pts[r], sum, cnt);
    if (j < count) partial[(size_t)chunk * count + j] = make_uint2(sum, cnt);
}

// ---- K1 main, LDS-tiled form ----------------------------------------------------------------------------
// byte offset of pixel (ix,iy) in the staged tile: iy*pitch2 + 2*ix + kofs, as exactly two VALU ops
__device__ static inline unsigned k1_tile_addr(int ix, int iy, int pitch2, int kofs)
{
    unsigned t, a;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(t) : "v"(ix), "s"(kofs));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(a) : "v"(iy), "s"(pitch2), "v"(t));
    return a;
}
// 16-bit LDS load at an absolute LDS byte address (saves the per-access `tile + offset` add)
typedef __attribute__((address_space(3))) const uint16_t k1_lds_u16;
__device__ static inline uint32_t k1_lds_load(unsigned addr)
{
    return *(k1_lds_u16 *)(size_t)addr;
}
// ray r of the block lives in lane (r - r0) of every wave: broadcast it without touching memory
__device__ static inline float2 k1_point(const float2 mine, int lane_idx)
{
    float2 p;
    p.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.x), lane_idx));
    p.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine.y), lane_idx));
    return p;
}

struct k1_ctrl {
    float bnd[5][8];      // [sub-batch 0..3, whole group][pxmin,pxmax,pymin,pymax,cmin,cmax,smin,smax]
    int box[5][4];        // [set][x0,y0,x1,y1] pixel box of the current ray block
    float wred[4][32];    // per-wave partial min/max
};

// end-point pixel box of one ray over a candidate set, by interval arithmetic on the reference's own
// float operations: every rounding step is monotone, so [lo,hi] bounds every candidate's coordinate.
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

template <bool VERIFY>
__global__ void __launch_bounds__(K1_THREADS)
k1_distance_tiled(const uint16_t *__restrict__ map, int S, const float2 *__restrict__ pts,
                  const int *__restrict__ rb_start, int n_rb, int blocks_per_chunk,
                  const float4 *__restrict__ pxcs, int count, int max_tile_bytes,
                  uint2 *__restrict__ partial,                      // [n_chunks][count], or NULL: finish in-kernel
                  int n_points, const int *__restrict__ ev_idx, int32_t *__restrict__ dist_out,
                  unsigned long long *__restrict__ key_out, unsigned int *__restrict__ verify_fail)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    k1_ctrl *ctl = (k1_ctrl *)smem;
    char *tile = smem + K1_CTRL_BYTES;
    const unsigned tile_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char *)tile;   // LDS byte address

    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int g0 = blockIdx.x * K1_GROUP;
    const int chunk = blockIdx.y;

    float4 q[K1_CPT];
    int jj[K1_CPT];
#pragma unroll
    for (int i = 0; i < K1_CPT; i++) {
        jj[i] = g0 + i * K1_THREADS + t;
        q[i] = pxcs[jj[i] < count ? jj[i] : count - 1];
    }

    // ---- candidate bounds per sub-batch and for the whole group -------------------------------------
#pragma unroll
    for (int i = 0; i < K1_CPT; i++) {
        float v[8] = { q[i].x, q[i].x, q[i].y, q[i].y, q[i].z, q[i].z, q[i].w, q[i].w };
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float x = v[k];
            for (int off = 32; off > 0; off >>= 1) {
                const float o = __shfl_xor(x, off, 64);
                x = (k & 1) ? fmaxf(x, o) : fminf(x, o);
            }
            if (lane == 0) ctl->wred[wid][i * 8 + k] = x;
        }
    }
    __syncthreads();
    if (t < 32) {
        const int k = t & 7;
        float x = ctl->wred[0][t];
        for (int w = 1; w < 4; w++) x = (k & 1) ? fmaxf(x, ctl->wred[w][t]) : fminf(x, ctl->wred[w][t]);
        ctl->bnd[t >> 3][k] = x;
    }
    __syncthreads();
    if (t < 8) {
        float x = ctl->bnd[0][t];
        for (int i = 1; i < 4; i++) x = (t & 1) ? fmaxf(x, ctl->bnd[i][t]) : fminf(x, ctl->bnd[i][t]);
        ctl->bnd[4][t] = x;
    }
    __syncthreads();

    uint32_t sum[K1_CPT] = { 0, 0, 0, 0 }, cnt[K1_CPT] = { 0, 0, 0, 0 };

    const int b0 = chunk * blocks_per_chunk;
    const int b1 = b0 + blocks_per_chunk < n_rb ? b0 + blocks_per_chunk : n_rb;
    for (int b = b0; b < b1; b++) {
        const int r0 = rb_start[b], r1 = rb_start[b + 1];
        const int nr = r1 - r0;                                    // <= CS_RB_MAX (32) <= 64 lanes
        const float2 mypt = pts[r0 + (lane < nr ? lane : 0)];

        // ---- pixel boxes of this ray block for the 5 candidate sets (wave 0, one lane per ray) -------
        if (wid == 0) {
            const float2 p = mypt;
#pragma unroll
            for (int s = 0; s < 5; s++) {
                int x0, y0, x1, y1;
                k1_ray_box(ctl->bnd[s], p, x0, y0, x1, y1);
                for (int off = 32; off > 0; off >>= 1) {
                    x0 = min(x0, __shfl_xor(x0, off, 64)); y0 = min(y0, __shfl_xor(y0, off, 64));
                    x1 = max(x1, __shfl_xor(x1, off, 64)); y1 = max(y1, __shfl_xor(y1, off, 64));
                }
                if (lane == 0) { ctl->box[s][0] = x0; ctl->box[s][1] = y0; ctl->box[s][2] = x1; ctl->box[s][3] = y1; }
            }
        }
        __syncthreads();

        // box geometry of candidate set `set` (uniform across the workgroup)
        auto geom = [&](int set, int &x0, int &y0, int &x1, int &y1, int &x0a, int &w8, int &h) -> bool {
            x0 = ctl->box[set][0]; y0 = ctl->box[set][1]; x1 = ctl->box[set][2]; y1 = ctl->box[set][3];
            x0a = x0 & ~7;
            w8 = ((x1 - x0a + 1) + 7) & ~7;                        // tile pitch in pixels (multiple of 8)
            h = y1 - y0 + 1;
            const bool inside = (x0 >= 0) & (y0 >= 0) & (x1 < S) & (y1 < S) & (x1 >= x0) & (y1 >= y0);
            return inside && ((long long)w8 * h * 2 <= (long long)max_tile_bytes);
        };
        // stage a tile: rows of 16-byte vectors, coalesced
        auto stage = [&](int x0a, int y0, int w8, int h) {
            const int vpr = w8 >> 3;                               // vectors per row
            const int nvec = vpr * h;
            const int dq = K1_THREADS / vpr, dr = K1_THREADS - dq * vpr;
            int row = t / vpr, col = t - row * vpr;
            for (int v = t; v < nvec; v += K1_THREADS) {
                const uint4 d = *(const uint4 *)(map + (size_t)(y0 + row) * S + x0a + (col << 3));
                *(uint4 *)(tile + ((size_t)(row * w8 + (col << 3)) << 1)) = d;
                row += dq; col += dr;
                if (col >= vpr) { col -= vpr; row++; }
            }
            __syncthreads();
        };

        int x0, y0, x1, y1, x0a, w8, h;
        if (geom(4, x0, y0, x1, y1, x0a, w8, h)) {
            // ---- the common case: one tile serves all four candidates of every lane ------------------------
            stage(x0a, y0, w8, h);
            const int pitch2 = w8 << 1;
            const int kofs = (int)tile_lds - ((y0 * w8 + x0a) << 1);
            for (int r = 0; r < nr; r++) {
                const float2 p = k1_point(mypt, r);
#pragma unroll
                for (int i = 0; i < K1_CPT; i++) {
                    float fx, fy;
                    k1_coords(q[i], p, fx, fy);
                    const int ix = (int)fx, iy = (int)fy;
                    if (VERIFY) { if (ix < x0 || ix > x1 || iy < y0 || iy > y1) { atomicAdd(verify_fail, 1u); continue; } }
                    sum[i] += k1_lds_load(k1_tile_addr(ix, iy, pitch2, kofs));
                }
            }
#pragma unroll
            for (int i = 0; i < K1_CPT; i++) cnt[i] += (uint32_t)nr;
            if (VERIFY && t == 0) atomicAdd(verify_fail + 1, (unsigned)nr * 4u);      // mode statistics (x256 lanes)
            __syncthreads();                                       // tile is overwritten by the next stage
        } else {
            // ---- tails of the theta distribution / long rays: one tile per 256-candidate sub-batch ------------
#pragma unroll
            for (int i = 0; i < K1_CPT; i++) {
                if (geom(i, x0, y0, x1, y1, x0a, w8, h)) {
                    stage(x0a, y0, w8, h);
                    const int pitch2 = w8 << 1;
                    const int kofs = (int)tile_lds - ((y0 * w8 + x0a) << 1);
                    for (int r = 0; r < nr; r++) {
                        float fx, fy;
                        k1_coords(q[i], k1_point(mypt, r), fx, fy);
                        const int ix = (int)fx, iy = (int)fy;
                        if (VERIFY) { if (ix < x0 || ix > x1 || iy < y0 || iy > y1) { atomicAdd(verify_fail, 1u); continue; } }
                        sum[i] += k1_lds_load(k1_tile_addr(ix, iy, pitch2, kofs));
                    }
                    cnt[i] += (uint32_t)nr;
                    if (VERIFY && t == 0) atomicAdd(verify_fail + 2, (unsigned)nr);
                    __syncthreads();
                } else {
                    if (VERIFY && t == 0) atomicAdd(verify_fail + 3, (unsigned)nr);
                    // box too large or touching the map border: bounds-checked global gathers
                    for (int r = 0; r < nr; r++) k1_gather_global<false>(map, S, q[i], k1_point(mypt, r), sum[i], cnt[i]);
                }
            }
        }
        __syncthreads();                                           // boxes are rewritten for the next ray block
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    if (partial) {
#pragma unroll
        for (int i = 0; i < K1_CPT; i++)
            if (jj[i] < count) partial[(size_t)chunk * count + jj[i]] = make_uint2(sum[i], cnt[i]);
    } else {
        unsigned long long key = ~0ull;
#pragma unroll
        for (int i = 0; i < K1_CPT; i++)
            if (jj[i] < count) {
                const unsigned long long k = k1_finish(sum[i], cnt[i], n_points, ev_idx ? ev_idx[jj[i]] : jj[i], dist_out);
                key = k < key ? k : key;
            }
        k1_wave_argmin(key, key_out);
    }
}

// ---- K1r: per-candidate reduction of the chunk partials + arg-min ----------------------------------------
// block = 64 candidates x 4 chunk slices
__global__ void __launch_bounds__(256)
k1_reduce(const uint2 *__restrict__ partial, int n_chunks, int count, int n_points,
          const int *__restrict__ ev_idx, int32_t *__restrict__ dist_out, unsigned long long *__restrict__ key_out)
{
    __shared__ uint32_t ssum[4][64], scnt[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    uint32_t sum = 0, cnt = 0;
    if (j < count)
        for (int c = w; c < n_chunks; c += 4) {
            const uint2 p = partial[(size_t)c * count + j];
            sum += p.x; cnt += p.y;
        }
    ssum[w][lane] = sum; scnt[w][lane] = cnt;
    __syncthreads();
    if (w != 0) return;
    unsigned long long key = ~0ull;
    if (j < count) {
        const uint64_t s = (uint64_t)ssum[0][lane] + ssum[1][lane] + ssum[2][lane] + ssum[3][lane];
        const uint32_t c = scnt[0][lane] + scnt[1][lane] + scnt[2][lane] + scnt[3][lane];
        key = k1_finish(s, c, n_points, ev_idx ? ev_idx[j] : j, dist_out);
    }
    k1_wave_argmin(key, key_out);
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

// Runs K1 (+ K1r) over d_pxcs[0..count) (evaluation order; d_ev_idx maps to flat indices).  The packed
// arg-min key is atomically min-ed into key_dst, which the prep kernel (or k1_arm_key) has armed.
// Asynchronous on the context's stream.
int32_t cs_launch_distance(slamhip_cs *cs, int count, bool want_dist, bool cand_sane, uint64_t *key_dst)
{
    slamhip_ctx *ctx = cs->ctx;
    if (cs->n_points <= 0) SH_FAIL(SLAMHIP_ERR_STATE, "no scan set (slamhip_cs_set_scan)");
    static const int force_global = env_int("SLAMHIP_K1_GLOBAL", 0);
    static const int verify = env_int("SLAMHIP_K1_VERIFY", 0);
    static const int tile_kb = env_int("SLAMHIP_K1_TILE_KB", 47);
    static const int target_wgs = env_int("SLAMHIP_K1_TARGET_WGS", 3072);
    const bool sane = cs->pts_sane && cand_sane;
    const bool tiled = sane && (cs->hs % 8 == 0) && !force_global && count >= 1;
    const int n_rb = cs->n_rb;
    int32_t *dist = want_dist ? cs->d_dist : nullptr;
    unsigned long long *key = (unsigned long long *)key_dst;

    if (tiled) {
        const int n_groups = sh_div_up(count, K1_GROUP);
        int bpc = (int)(((long long)n_groups * n_rb) / target_wgs);
        if (bpc < 1) bpc = 1;
        if (bpc > n_rb) bpc = n_rb;
        const int n_chunks = sh_div_up(n_rb, bpc);
        const bool in_kernel = n_chunks == 1 && cs->n_points <= 65536;
        if (!in_kernel) SH_TRY(ensure_partial(cs, (size_t)n_chunks * count));
        const int max_tile = tile_kb * 1024;
        const size_t lds = (size_t)K1_CTRL_BYTES + max_tile;
        static_assert(sizeof(k1_ctrl) <= K1_CTRL_BYTES, "control block too large");
        {
            sh_timer t(ctx, SLAMHIP_K_CS_DISTANCE);
            dim3 grid(n_groups, n_chunks);
            if (verify)
                hipLaunchKernelGGL(k1_distance_tiled<true>, grid, dim3(K1_THREADS), lds, ctx->stream,
                                   cs->d_hole, cs->hs, cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, max_tile,
                                   in_kernel ? (uint2 *)nullptr : (uint2 *)cs->d_partial, cs->n_points, cs->d_ev_idx, dist, key,
                                   (unsigned int *)cs->d_verify);
            else
                hipLaunchKernelGGL(k1_distance_tiled<false>, grid, dim3(K1_THREADS), lds, ctx->stream,
                                   cs->d_hole, cs->hs, cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, max_tile,
                                   in_kernel ? (uint2 *)nullptr : (uint2 *)cs->d_partial, cs->n_points, cs->d_ev_idx, dist, key,
                                   (unsigned int *)cs->d_verify);
        }
        if (!in_kernel) {
            sh_timer t(ctx, SLAMHIP_K_CS_REDUCE);
            hipLaunchKernelGGL(k1_reduce, dim3(sh_div_up(count, 64)), dim3(256), 0, ctx->stream,
                               (const uint2 *)cs->d_partial, n_chunks, count, cs->n_points, cs->d_ev_idx, dist, key);
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
                                   cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, (uint2 *)cs->d_partial);
            else
                hipLaunchKernelGGL(k1_distance_global<true>, grid, dim3(K1_THREADS), 0, ctx->stream, cs->d_hole, cs->hs,
                                   cs->d_pts_sorted, cs->d_rb_start, n_rb, bpc, cs->d_pxcs, count, (uint2 *)cs->d_partial);
        }
        {
            sh_timer t(ctx, SLAMHIP_K_CS_REDUCE);
            hipLaunchKernelGGL(k1_reduce, dim3(sh_div_up(count, 64)), dim3(256), 0, ctx->stream,
                               (const uint2 *)cs->d_partial, n_chunks, count, cs->n_points, cs->d_ev_idx, dist, key);
        }
    }
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}

void cs_launch_prep_offsets(slamhip_cs *cs, int count, const float pose[3], uint64_t *key_dst)
{
    sh_timer t(cs->ctx, SLAMHIP_K_CS_PREP);
    hipLaunchKernelGGL(k1_prep_offsets, dim3(sh_div_up(count, 256)), dim3(256), 0, cs->ctx->stream,
                       cs->d_ev_off, count, pose[0], pose[1], pose[2], cs->hscale, cs->d_pxcs, (unsigned long long *)key_dst);
}

void cs_launch_prep_poses(slamhip_cs *cs, const float *d_poses, int count, uint64_t *key_dst)
{
    sh_timer t(cs->ctx, SLAMHIP_K_CS_PREP);
    hipLaunchKernelGGL(k1_prep_poses, dim3(sh_div_up(count, 256)), dim3(256), 0, cs->ctx->stream,
                       d_poses, count, cs->hscale, cs->d_pxcs, (unsigned long long *)key_dst);
}

void cs_launch_arm_key(slamhip_cs *cs, uint64_t *key_dst)
{
    hipLaunchKernelGGL(k1_arm_key, dim3(1), dim3(1), 0, cs->ctx->stream, (unsigned long long *)key_dst);
}
