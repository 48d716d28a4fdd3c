// distance.hip -- K1: batched CoreSLAM scan-to-map distance + arg-min (gfx950 only).
//
// Replaces CalculateDistanceSISD (CoreSLAM/CoreSLAMProcessor.cs:226-259) called from MonteCarloSearch
// (:624-653) on ParallelWorker threads (:674-710).  Arithmetic contract:
//   ix = (int)((px + c*X) - s*Y), iy = (int)((py + s*X) + c*Y) in binary32, one rounding per op,
//   no FMA (-ffp-contract=off), truncation toward zero; in-bounds pixels are summed as integers;
//   distance = (int)(sum*1024 / R) with R = ALL points (:253), int.MaxValue if none in bounds (:257).
// Integer sums make any evaluation order exact, so rays are processed in spatially sorted blocks and
// candidates in theta-sorted order; the arg-min key (distance << 32 | flat index) restores the
// reference tie-break (first strictly smaller wins, :644,:700).
#include "cs_internal.h"
#include "det_trig.h"

#define K1_THREADS 256

// ---- candidate preparation -------------------------------------------------------------------------
// pose_k = search_pose + offs_k (:635-637); (px,py,c,s) per :232-235 with deterministic trig.
__global__ void __launch_bounds__(256)
k1_prep_offsets(const float *__restrict__ ev_off, int count, float bx, float by, float bth, float scale,
                float4 *__restrict__ pxcs)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
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
k1_prep_poses(const float *__restrict__ poses, int count, float scale, float4 *__restrict__ pxcs)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
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

// ---- K1 main, global-gather form ----------------------------------------------------------------------
// grid = (candidate groups, ray blocks).  One lane = one candidate; the ray block's points are
// wave-uniform (scalar loads).  SAFE adds the NaN / overflow handling of sh_f2i; the fast form relies
// on |coords| < 1e9 (checked on the host), where v_cvt_i32_f32 (truncating, saturating) == (int)f.
template <bool SAFE>
__global__ void __launch_bounds__(K1_THREADS)
k1_distance_global(const uint16_t *__restrict__ map, int S, const float2 *__restrict__ pts,
                   const int *__restrict__ rb_start, const float4 *__restrict__ pxcs, int count,
                   uint32_t *__restrict__ partial)
{
    const int j = blockIdx.x * K1_THREADS + threadIdx.x;
    const int rb = blockIdx.y;
    const int r0 = rb_start[rb], r1 = rb_start[rb + 1];
    float4 q = pxcs[j < count ? j : count - 1];
    uint32_t sum = 0, cnt = 0;
    for (int r = r0; r < r1; r++) {
        const float2 p = pts[r];
        float fx = q.x + q.z * p.x;  fx = fx - q.w * p.y;      // :240
        float fy = q.y + q.w * p.x;  fy = fy + q.z * p.y;      // :241
        int ix, iy;
        if (SAFE) { ix = sh_f2i(fx); iy = sh_f2i(fy); }
        else      { ix = (int)fx;    iy = (int)fy; }
        const bool ok = ((unsigned)ix < (unsigned)S) & ((unsigned)iy < (unsigned)S);   // :244
        uint32_t v = 0;
        if (ok) v = map[(size_t)iy * S + ix];                  // :246
        sum += v;
        cnt += ok ? 1u : 0u;
    }
    if (j < count) partial[(size_t)rb * count + j] = (cnt << CS_PART_SUM_BITS) | sum;
}

// ---- K1r: per-candidate reduction of the ray-block partials + arg-min ------------------------------------
__global__ void __launch_bounds__(256)
k1_reduce(const uint32_t *__restrict__ partial, int n_rb, int count, int n_points,
          const int *__restrict__ ev_idx, int32_t *__restrict__ dist_out, unsigned long long *__restrict__ key_out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long key = ~0ull;
    if (j < count) {
        uint64_t sum = 0; uint32_t cnt = 0;
        for (int rb = 0; rb < n_rb; rb++) {
            uint32_t p = partial[(size_t)rb * count + j];
            sum += p & CS_PART_SUM_MASK;
            cnt += p >> CS_PART_SUM_BITS;
        }
        int32_t d = cnt > 0 ? (int32_t)((sum * 1024ull) / (uint64_t)n_points) : INT32_MAX;   // :251-258
        const int flat = ev_idx ? ev_idx[j] : j;
        if (dist_out) dist_out[flat] = d;
        key = ((unsigned long long)(uint32_t)d << 32) | (uint32_t)flat;
    }
    // 64-lane wavefront min, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(key, off, 64);
        key = o < key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key != ~0ull) atomicMin(key_out, key);
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

static int32_t ensure_partial(slamhip_cs *cs, int count)
{
    size_t need = (size_t)cs->n_rb * (size_t)count;
    if (need <= cs->cap_partial) return SLAMHIP_OK;
    if (cs->d_partial) (void)hipFree(cs->d_partial);
    cs->d_partial = nullptr; cs->cap_partial = 0;
    need += need / 4;
    SH_HIP(hipMalloc(&cs->d_partial, sizeof(uint32_t) * need));
    cs->cap_partial = need;
    return SLAMHIP_OK;
}

// Runs K1 + K1r over d_pxcs[0..count) (evaluation order, d_ev_idx maps to flat indices); the packed
// arg-min key lands in cs->d_key.  Asynchronous on the context's stream.
int32_t cs_launch_distance(slamhip_cs *cs, int count, bool want_dist, bool cand_sane)
{
    slamhip_ctx *ctx = cs->ctx;
    if (cs->n_points <= 0) SH_FAIL(SLAMHIP_ERR_STATE, "no scan set (slamhip_cs_set_scan)");
    SH_TRY(ensure_partial(cs, count));
    SH_HIP(hipMemsetAsync(cs->d_key, 0xFF, sizeof(uint64_t), ctx->stream));
    {
        sh_timer t(ctx, SLAMHIP_K_CS_DISTANCE);
        dim3 grid(sh_div_up(count, K1_THREADS), cs->n_rb);
        if (cs->pts_sane && cand_sane)
            hipLaunchKernelGGL(k1_distance_global<false>, grid, dim3(K1_THREADS), 0, ctx->stream,
                               cs->d_hole, cs->hs, cs->d_pts_sorted, cs->d_rb_start, cs->d_pxcs, count, cs->d_partial);
        else
            hipLaunchKernelGGL(k1_distance_global<true>, grid, dim3(K1_THREADS), 0, ctx->stream,
                               cs->d_hole, cs->hs, cs->d_pts_sorted, cs->d_rb_start, cs->d_pxcs, count, cs->d_partial);
    }
    {
        sh_timer t(ctx, SLAMHIP_K_CS_REDUCE);
        hipLaunchKernelGGL(k1_reduce, dim3(sh_div_up(count, 256)), dim3(256), 0, ctx->stream,
                           cs->d_partial, cs->n_rb, count, cs->n_points, cs->d_ev_idx,
                           want_dist ? cs->d_dist : nullptr, (unsigned long long *)cs->d_key);
    }
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}

// exported for coreslam.hip
void cs_launch_prep_offsets(slamhip_cs *cs, int count, const float pose[3])
{
    sh_timer t(cs->ctx, SLAMHIP_K_CS_PREP);
    hipLaunchKernelGGL(k1_prep_offsets, dim3(sh_div_up(count, 256)), dim3(256), 0, cs->ctx->stream,
                       cs->d_ev_off, count, pose[0], pose[1], pose[2], cs->hscale, cs->d_pxcs);
}

void cs_launch_prep_poses(slamhip_cs *cs, const float *d_poses, int count)
{
    sh_timer t(cs->ctx, SLAMHIP_K_CS_PREP);
    hipLaunchKernelGGL(k1_prep_poses, dim3(sh_div_up(count, 256)), dim3(256), 0, cs->ctx->stream,
                       d_poses, count, cs->hscale, cs->d_pxcs);
}
