// coreslam.hip -- CoreSLAM operator-level entry points of include/slamhip.h (gfx950 only).
#include "cs_internal.h"
#include "det_trig.h"
#include "obstacle_dev.h"
#include <algorithm>
#include <numeric>
#include <math.h>
#include <stdlib.h>

// ---- small kernels ------------------------------------------------------------------------------------
__global__ void k_fill_u16(uint16_t *p, size_t n, uint16_t v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// HoleMap.GetPackedPixels (HoleMap.cs:44-55)
__global__ void k_pack_holemap(const uint16_t *__restrict__ pix, uint8_t *__restrict__ out, size_t n_bytes)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_bytes) return;
    out[i] = (uint8_t)(((pix[i * 2] >> 12) << 4) | (pix[i * 2 + 1] >> 12));      // :51
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3")
__device__ static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// Device replacement for FillRandomQueues (CoreSLAMProcessor.cs:599-612): jitter i is a pure function of
// (seed, stream, i).  dx,dy ~ N(0, sigma_xy) by Box-Muller; dtheta is STRATIFIED: the i-th of n equal-
// probability strata of N(0, sigma_theta), so the flat list is already sorted by theta (no sort on the
// search path) while every dtheta is still N(0, sigma_theta) distributed.
__device__ static inline void k_jitter(int i, int n, float sigma_xy, float sigma_theta, uint64_t seed, uint64_t stream, float o[3])
{
    uint32_t c[4] = { (uint32_t)i, 0u, (uint32_t)stream, (uint32_t)(stream >> 32) };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530718f * u2, &sn, &cs);
    // (the top stratum's quantile can round to 1.0f, whose normal quantile is +inf: keep it below)
    const float q = fminf(((float)i + u3) / (float)n, 0.99999994f);
    o[0] = sigma_xy * rad * cs;
    o[1] = sigma_xy * rad * sn;
    o[2] = sigma_theta * normcdfinvf(q);
}
// The heading lattice (opt-in: slamhip_cs_generate_offsets_lattice): the candidates that ONE LANE of the search kernel evaluates --
// evaluation positions t, t + lanes, ... of a candidate group -- share their dtheta, bit for bit, and differ in their translation.
// The kernel then forms the four products c*X, s*Y, s*X, c*Y of a ray point once per lane instead of once per candidate (the sums
// stay per candidate and in the reference's order, :240-241: every candidate's coordinates are the floats they always were).
// The headings are the strata of N(0, sigma_theta) as before, one per lane position: stratum u = group * lanes + lane of
// U = the number of lane positions in use; the stratum that holds the un-jittered pose has dtheta = 0 for all its members.
// lat = candidates per lane (2 or 4; 0: no lattice), grp = candidates per group, zero_pos = evaluation position of the un-jittered pose.
struct k_lattice { int cpl, grp, count, zero_pos; };
__device__ static inline int k_lat_stratum(const k_lattice L, int j) { const int lanes = L.grp / L.cpl; return (j / L.grp) * lanes + (j % lanes); }
__device__ static inline void k_jitter_lat(int i, int n, float sigma_xy, float sigma_theta, uint64_t seed, uint64_t stream, const k_lattice L, float o[3])
{
    k_jitter(i, n, sigma_xy, sigma_theta, seed, stream, o);              // dx, dy as ever (keyed by the jitter's index)
    const int j = i < L.zero_pos ? i : i + 1;                            // the jitter's evaluation position (flat i + 1)
    const int u = k_lat_stratum(L, j), u0 = k_lat_stratum(L, L.zero_pos);
    const int lanes = L.grp / L.cpl, full = L.count / L.grp, rest = L.count - full * L.grp;
    const int U = full * lanes + (rest < lanes ? rest : lanes);
    if (u == u0) { o[2] = 0.0f; return; }
    uint32_t c[4] = { (uint32_t)u, 0x4C415454u, (uint32_t)stream, (uint32_t)(stream >> 32) };      // (keyed by the stratum: "LATT")
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u3 = ((float)(c[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float q = fminf(((float)u + u3) / (float)U, 0.99999994f);
    o[2] = sigma_theta * normcdfinvf(q);
}
__global__ void k_generate_offsets(float *__restrict__ offs_flat, int n, float sigma_xy, float sigma_theta,
                                   uint64_t seed, uint64_t stream, const k_lattice L)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[3];
    if (L.cpl) k_jitter_lat(i, n, sigma_xy, sigma_theta, seed, stream, L, o);
    else k_jitter(i, n, sigma_xy, sigma_theta, seed, stream, o);
    offs_flat[3 * i + 0] = o[0]; offs_flat[3 * i + 1] = o[1]; offs_flat[3 * i + 2] = o[2];
}

// evaluation list: ev_idx[j] (or first + j) is a flat candidate index; flat 0 is the un-jittered pose
// (theta-sorted flat lists: the un-jittered pose, dtheta = 0, is evaluated at position zero_pos, between the
// negative and the positive dtheta, so that it does not widen the theta range of the first group).
// One workgroup per candidate group of K1_GROUP: it also leaves the group's jitter bounds {min dx, max dx, min dy,
// max dy, min dtheta, max dtheta} for K1, which turns them into the bounds of the candidates' (px, py, c, s) for the
// search pose of the launch without reading the candidates (k1_search_tiled).
// GEN: the whole device-generated list is evaluated (first = 0, count = n + 1), so the jitters are produced here, in
// evaluation order, and stored to the flat list as well: one launch instead of two per scan.
// Launched on the side stream (ensure_shard), the launch tells the HOST when everything it wrote is visible to a launch that has
// not started yet: every workgroup waits for its stores, writes its XCD's L2 back (agent-scope release) and arrives; the last
// arriver stores side_seq into a pinned word.  (The runtime's own completion tracking -- hipStreamQuery, or an event the other
// stream waits for -- was measured 10 - 16 us slower per scan than the launch saves.)
__device__ static inline void k_side_arrive(unsigned *__restrict__ arrive, uint32_t *__restrict__ flag, uint32_t seq)
{
    if (!flag) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned n = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n == gridDim.x - 1) {
            __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <bool GEN>
__global__ void __launch_bounds__(1024)
k_gather_offsets(float *__restrict__ offs_flat, const int *__restrict__ ev_idx_in, int first, int count, int zero_pos,
                 float *__restrict__ ev_off, int *__restrict__ ev_idx_out, float *__restrict__ grp_bounds, int grp,
                 int gen_n, float gen_sxy, float gen_sth, uint64_t gen_seed, uint64_t gen_stream,
                 const uint4 *__restrict__ up_src, uint4 *__restrict__ up_dst, int up_n16, uint32_t *__restrict__ up_flag, uint32_t up_seq,
                 unsigned *__restrict__ side_arrive, uint32_t *__restrict__ side_flag, uint32_t side_seq, const k_lattice lat)
{
    if (up_n16 > 0 && (int)blockIdx.x >= (int)gridDim.x - SH_UPLOAD_PARTS) {   // riding along: the scan upload (the two are independent, K1 needs both)
        sh_upload16_part(up_src, up_dst, up_n16, (int)blockIdx.x - ((int)gridDim.x - SH_UPLOAD_PARTS), up_flag, up_seq);
        k_side_arrive(side_arrive, side_flag, side_seq);
        return;
    }
    __shared__ float red[16][6];
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int jj = threadIdx.x; jj < grp; jj += 1024) {            // (grp = 1024 or 2048 candidates per group)
        const int j = blockIdx.x * grp + jj;
        if (j < count) {
            int flat = ev_idx_in ? ev_idx_in[j] : first + j;
            if (!ev_idx_in && zero_pos >= 0) flat = j < zero_pos ? j + 1 : j == zero_pos ? 0 : j;
            float ox = 0.f, oy = 0.f, ot = 0.f;
            if (flat > 0) {
                if (GEN) {
                    float o[3];
                    if (lat.cpl) k_jitter_lat(flat - 1, gen_n, gen_sxy, gen_sth, gen_seed, gen_stream, lat, o);
                    else k_jitter(flat - 1, gen_n, gen_sxy, gen_sth, gen_seed, gen_stream, o);
                    ox = o[0]; oy = o[1]; ot = o[2];
                    offs_flat[3 * (size_t)(flat - 1)] = ox; offs_flat[3 * (size_t)(flat - 1) + 1] = oy; offs_flat[3 * (size_t)(flat - 1) + 2] = ot;
                } else { ox = offs_flat[3 * (size_t)(flat - 1)]; oy = offs_flat[3 * (size_t)(flat - 1) + 1]; ot = offs_flat[3 * (size_t)(flat - 1) + 2]; }
            }
            ev_off[3 * (size_t)j] = ox; ev_off[3 * (size_t)j + 1] = oy; ev_off[3 * (size_t)j + 2] = ot;
            if (ev_idx_out) ev_idx_out[j] = flat;
            lo[0] = fminf(lo[0], ox); hi[0] = fmaxf(hi[0], ox); lo[1] = fminf(lo[1], oy); hi[1] = fmaxf(hi[1], oy);
            lo[2] = fminf(lo[2], ot); hi[2] = fmaxf(hi[2], ot);
        }
    }
    for (int m = 1; m < 64; m <<= 1)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], m)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], m)); }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; k++) { red[wv][2 * k] = lo[k]; red[wv][2 * k + 1] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = red[0][threadIdx.x];
        for (int w = 1; w < 16; w++) v = (threadIdx.x & 1) ? fmaxf(v, red[w][threadIdx.x]) : fminf(v, red[w][threadIdx.x]);
        grp_bounds[8 * (size_t)blockIdx.x + threadIdx.x] = v;          // (NaN jitters never reach the tiled kernel: sanity flags)
    }
    k_side_arrive(side_arrive, side_flag, side_seq);
}

// known-answer access to the generator's integer stream (slamhip_ctx_philox4x32_10): one Philox4x32-10 block on the device
__global__ void k_philox_kat(uint32_t *io)
{
    uint32_t c[4] = { io[0], io[1], io[2], io[3] };
    philox4x32_10(c, io[4], io[5]);
    io[6] = c[0]; io[7] = c[1]; io[8] = c[2]; io[9] = c[3];
}

extern "C" int32_t slamhip_ctx_philox4x32_10(slamhip_ctx *ctx, const uint32_t counter[4], const uint32_t key[2], uint32_t out[4])
{
    SH_CHECK_ARG(ctx && counter && key && out);
    SH_HIP(hipSetDevice(ctx->device));
    uint32_t h[10] = { counter[0], counter[1], counter[2], counter[3], key[0], key[1], 0, 0, 0, 0 };
    uint32_t *d = nullptr;
    SH_HIP(hipMalloc(&d, sizeof(h)));
    hipError_t e = hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_philox_kat, dim3(1), dim3(1), 0, ctx->stream, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    SH_HIP(e);
    for (int k = 0; k < 4; k++) out[k] = h[6 + k];
    return SLAMHIP_OK;
}

// winner pose from the packed key: search_pose + offs[index-1] (:635-637), theta normalised (:746)
__global__ void k_best_pose(const unsigned long long *__restrict__ key, const float *__restrict__ offs_flat,
                            float bx, float by, float bth, float *__restrict__ out_pose)
{
    const uint32_t flat = (uint32_t)(*key);
    float x = bx, y = by, th = bth;
    if (flat > 0) { x = bx + offs_flat[3 * (flat - 1)]; y = by + offs_flat[3 * (flat - 1) + 1]; th = bth + offs_flat[3 * (flat - 1) + 2]; }
    out_pose[0] = x; out_pose[1] = y; out_pose[2] = sh_normalize_angle(th);
    out_pose[3] = th;     // un-normalised, as MonteCarloSearch returns it
}

// ---- lifecycle -------------------------------------------------------------------------------------------
extern "C" int32_t slamhip_cs_destroy(slamhip_cs *cs)
{
    if (!cs) return SLAMHIP_OK;
    (void)hipSetDevice(cs->ctx->device);
    if (!cs->ctx->poisoned) (void)hipStreamSynchronize(cs->ctx->stream);     // (a poisoned context's stream may never drain: slamhip_ctx_destroy bounds that wait)
    cs_plan_free(cs);
    (void)hipFree(cs->d_hole); (void)hipFree(cs->d_obst);
    (void)hipFree(cs->d_scan_blob); if (cs->h_scan_blob) (void)hipHostFree(cs->h_scan_blob);
    (void)hipFree(cs->d_scan_flag);
    if (cs->ev_scan) (void)hipEventDestroy(cs->ev_scan);
    (void)hipFree(cs->d_offs_flat); (void)hipFree(cs->d_ev_off); (void)hipFree(cs->d_ev_idx);
    (void)hipFree(cs->d_pxcs); (void)hipFree(cs->d_partial); (void)hipFree(cs->d_dist);
    (void)hipFree(cs->d_key); (void)hipFree(cs->d_grp_bounds); (void)hipFree(cs->d_verify);
    (void)hipFree(cs->d_k1_gmin); (void)hipFree(cs->d_k1_acc); (void)hipFree(cs->d_k1_ring);
    if (cs->h_key) (void)hipHostFree(cs->h_key);
    // (the helper streams are the context's: what this object put into them is waited for, the streams stay)
    if (!cs->ctx->poisoned) { (void)hipStreamSynchronize(cs->mirror_stream); (void)hipStreamSynchronize(cs->side_stream); }
    (void)hipFree(cs->d_side_arrive);
    (void)hipFree(cs->spec_offs_flat); (void)hipFree(cs->spec_ev_off); (void)hipFree(cs->spec_ev_idx); (void)hipFree(cs->spec_grp_bounds);
    (void)hipFree(cs->cool_offs_flat); (void)hipFree(cs->cool_ev_off); (void)hipFree(cs->cool_ev_idx); (void)hipFree(cs->cool_grp_bounds);
    if (cs->ev_snap) (void)hipEventDestroy(cs->ev_snap);
    if (cs->ev_push) (void)hipEventDestroy(cs->ev_push);
    (void)hipFree(cs->d_hole_span); (void)hipFree(cs->d_hole_span_snap); (void)hipFree(cs->d_hole_shadow); (void)hipFree(cs->d_mirror_sum); (void)hipFree(cs->d_mirror_mask);
    if (cs->h_mirror_sum) (void)hipHostFree(cs->h_mirror_sum);
    (void)hipFree(cs->d_mirror_rows);
    if (cs->h_mirror_rows) (void)hipHostFree(cs->h_mirror_rows);
    if (cs->h_mirror_stage) (void)hipHostFree(cs->h_mirror_stage);
    if (cs->mirror_reg) (void)hipHostUnregister(cs->mirror_reg);
    cs_holemap_free(cs);
    cs_obstacle_free(cs);
    delete cs;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_create(slamhip_ctx *ctx, float physical, int32_t hole_size, int32_t obst_size, slamhip_cs **out)
{
    SH_CHECK_ARG(ctx && out);
    SH_CHECK_ARG(hole_size >= 2 && hole_size <= 32768 && obst_size >= 1 && obst_size <= 32768);
    SH_CHECK_ARG(physical > 0.0f);
    SH_HIP(hipSetDevice(ctx->device));
    slamhip_cs *cs = new slamhip_cs();
    cs->ctx = ctx;
    cs->plan_stream = ctx->plan_stream; cs->side_stream = ctx->side_stream; cs->mirror_stream = ctx->mirror_stream;   // (the context's: common.h)
    cs->physical = physical;
    cs->hs = hole_size; cs->hscale = (float)hole_size / physical;          // HoleMap.cs:19-20
    cs->os = obst_size; cs->oscale = (float)obst_size / physical;          // ObstacleMap.cs:19-20
    cs->shard_first = cs->shard_count = -1;
    cs->offs_theta_small = true; cs->k1_layout_dirty = true;
    int32_t rc = SLAMHIP_OK;
    do {
        if (hipMalloc(&cs->d_hole, sizeof(uint16_t) * (size_t)hole_size * hole_size) != hipSuccess ||
            hipMalloc(&cs->d_obst, (size_t)obst_size * obst_size) != hipSuccess ||
            hipMalloc(&cs->d_key, 64) != hipSuccess ||          // result block: key (8 B) | winner pose (16 B) | blended pixels (4 B)

            hipMalloc(&cs->d_verify, sizeof(unsigned int) * 8) != hipSuccess ||
            hipHostMalloc(&cs->h_key, 128, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { slamhip_set_error("device allocation failed"); rc = SLAMHIP_ERR_NOMEM; break; }
        cs->d_best_pose = (float *)cs->d_key + 2;
        memset(cs->h_key, 0, 128);
        if (hipMemset(cs->d_verify, 0, sizeof(unsigned int) * 8) != hipSuccess) { slamhip_set_error("hipMemset failed"); rc = SLAMHIP_ERR_HIP; break; }
        if ((rc = cs_holemap_alloc(cs)) != SLAMHIP_OK) break;
        if ((rc = cs_obstacle_alloc(cs)) != SLAMHIP_OK) break;
        if ((rc = slamhip_cs_reset(cs, -5)) != SLAMHIP_OK) break;          // CoreSLAMProcessor.cs:96,:140
    } while (0);
    if (rc != SLAMHIP_OK) { slamhip_cs_destroy(cs); return rc; }
    *out = cs;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_info(slamhip_cs *cs, int32_t *hs, float *hscale, int32_t *os, float *oscale)
{
    SH_CHECK_ARG(cs);
    if (hs) *hs = cs->hs;
    if (hscale) *hscale = cs->hscale;
    if (os) *os = cs->os;
    if (oscale) *oscale = cs->oscale;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_reset(slamhip_cs *cs, int32_t unmapped)
{
    SH_CHECK_ARG(cs);
    SH_CHECK_ARG(unmapped >= -128 && unmapped <= 127);
    SH_HIP(hipSetDevice(cs->ctx->device));
    const size_t n = (size_t)cs->hs * cs->hs;
    hipLaunchKernelGGL(k_fill_u16, dim3(1024), dim3(256), 0, cs->ctx->stream, cs->d_hole, n,
                       (uint16_t)((0 + 65500) / 2));                       // :169 (TS_OBSTACLE + TS_NO_OBSTACLE) / 2
    SH_TRY(cs_obstacle_flush(cs));
    SH_HIP(hipMemsetAsync(cs->d_obst, (int)(uint8_t)(int8_t)unmapped, (size_t)cs->os * cs->os, cs->ctx->stream)); // :170
    SH_TRY(cs_holemap_dirty_set(cs, true));
    SH_TRY(cs_holemap_span_set(cs, true));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    return SLAMHIP_OK;
}

// ---- map transfer ---------------------------------------------------------------------------------------
extern "C" int32_t slamhip_cs_holemap_upload(slamhip_cs *cs, const uint16_t *pix, size_t n)
{
    SH_CHECK_ARG(cs && pix && n == (size_t)cs->hs * cs->hs);
    SH_HIP(hipSetDevice(cs->ctx->device));
    SH_HIP(hipMemcpyAsync(cs->d_hole, pix, n * sizeof(uint16_t), hipMemcpyHostToDevice, cs->ctx->stream));
    SH_TRY(cs_holemap_dirty_set(cs, true));
    SH_TRY(cs_holemap_span_set(cs, true));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    return SLAMHIP_OK;
}
extern "C" int32_t slamhip_cs_holemap_download(slamhip_cs *cs, uint16_t *pix, size_t n)
{
    SH_CHECK_ARG(cs && pix && n == (size_t)cs->hs * cs->hs);
    SH_HIP(hipSetDevice(cs->ctx->device));
    SH_HIP(hipMemcpyAsync(pix, cs->d_hole, n * sizeof(uint16_t), hipMemcpyDeviceToHost, cs->ctx->stream));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    return SLAMHIP_OK;
}
// Live `HoleMap.Pixels` (HoleMap.cs:27; read directly by Simulation/MainWindow.xaml.cs:229) at the price of what changed: every
// HoleMap update leaves the bounding square of its scan in a device-side dirty rectangle; this call fetches the rectangle, copies
// only those rows and columns into the caller's full-size array and rests the rectangle.  `pix` must be the array the previous
// mirror call (or a full download) filled; the first call after create / reset / upload copies the whole map.
extern "C" int32_t slamhip_cs_holemap_mirror_wait(slamhip_cs *cs, int32_t out_rect[4], int64_t *out_pixels);

// Where may the device write?  Page-locking a caller's array (hipHostRegister) is only safe when the array OWNS its pages: an
// array in the middle of a heap shares its first and last page with other objects, and the runtime pins those for its own
// pageable copies -- read-only for a copy source.  Seen in the randomised soak (tests/fuzz_parity.py, small NumPy arrays): after a
// few hundred register / unregister cycles a later device write into such a page died with "write access to a read-only page".
// So: an array that starts on a page boundary and is a whole number of pages long (an aligned allocation: posix_memalign,
// NativeMemory.AlignedAlloc, mmap) is registered and written by the device directly; any other array is served through a pinned
// staging buffer of the library's own and copied by the HOST in the waiting call, row range by row range.
static bool mirror_owns_pages(const void *p, size_t bytes)
{
    static const bool never = getenv("SLAMHIP_MIRROR_NOREG") != nullptr;
    return !never && ((uintptr_t)p & 4095u) == 0 && (bytes & 4095u) == 0 && bytes >= 4096;
}
static int32_t mirror_stage(slamhip_cs *cs)
{
    if (cs->h_mirror_stage) return SLAMHIP_OK;
    SH_HIP(hipHostMalloc(&cs->h_mirror_stage, sizeof(uint16_t) * (size_t)cs->hs * cs->hs));
    return SLAMHIP_OK;
}
// the array the device may write for this caller array: the array itself (registered, `flags`) or the staging buffer
static int32_t mirror_target(slamhip_cs *cs, uint16_t *pix, size_t n, unsigned flags, uint16_t **out_dev, bool *out_direct)
{
    const size_t bytes = n * sizeof(uint16_t);
    if (mirror_owns_pages(pix, bytes)) {
        if (cs->mirror_reg != (void *)pix || cs->mirror_reg_bytes != bytes || cs->mirror_reg_flags != flags) {
            if (cs->mirror_reg) (void)hipHostUnregister(cs->mirror_reg);
            cs->mirror_reg = nullptr; cs->mirror_reg_bytes = 0; cs->mirror_dev_ptr = nullptr;
            // An array the caller page-locked itself (hipHostMalloc, a registered or pinned-tensor buffer) answers
            // hipErrorHostMemoryAlreadyRegistered, and any other refusal (a locked-memory limit) is no reason to fail the mirror
            // either: such an array is served through the staging buffer, like one that does not own its pages.
            const hipError_t er = hipHostRegister(pix, bytes, flags);
            bool ok = er == hipSuccess;
            if (ok && (flags & hipHostRegisterMapped) && hipHostGetDevicePointer(&cs->mirror_dev_ptr, pix, 0) != hipSuccess) {
                (void)hipHostUnregister(pix); cs->mirror_dev_ptr = nullptr; ok = false;
            }
            if (!ok) (void)hipGetLastError();                      // (the runtime's sticky error word)
            else { cs->mirror_reg = pix; cs->mirror_reg_bytes = bytes; cs->mirror_reg_flags = flags; }
        }
        if (cs->mirror_reg == (void *)pix) {
            *out_dev = (flags & hipHostRegisterMapped) ? (uint16_t *)cs->mirror_dev_ptr : pix;
            *out_direct = true;
            return SLAMHIP_OK;
        }
    }
    SH_TRY(mirror_stage(cs));
    *out_dev = cs->h_mirror_stage;
    *out_direct = false;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_holemap_mirror(slamhip_cs *cs, uint16_t *pix, size_t n, int32_t out_rect[4])
{
    SH_CHECK_ARG(cs && pix && n == (size_t)cs->hs * cs->hs);
    slamhip_ctx *ctx = cs->ctx;
    SH_HIP(hipSetDevice(ctx->device));
    if (cs->mirror_pending) SH_TRY(slamhip_cs_holemap_mirror_wait(cs, nullptr, nullptr));
    int r[4];
    {
        sh_mail_guard lock(ctx);
        SH_TRY(sh_publish(ctx, cs->d_hole_dirty, 4));
        SH_TRY(cs_holemap_dirty_set(cs, false));                   // (behind the publish on the stream; later updates are behind this)
        SH_TRY(sh_host_wait(ctx));
        memcpy(r, (const void *)ctx->mailbox, sizeof(r));
    }
    if (r[2] >= r[0] && r[3] >= r[1]) {
        // (a strided copy into pageable memory goes row by row through the runtime's staging buffer: 2.1 ms for a 1713 x 1713
        // rectangle, against 0.23 ms for the whole 2048^2 map in one piece -- so the copy lands in page-locked memory: the caller's
        // array when it owns its pages, else the library's staging buffer, from which the host copies the rows on)
        uint16_t *dst = nullptr; bool direct = false;
        SH_TRY(mirror_target(cs, pix, n, hipHostRegisterDefault, &dst, &direct));
        const size_t pitch = (size_t)cs->hs * sizeof(uint16_t);
        const size_t rows = (size_t)(r[3] - r[1] + 1), cols = (size_t)(r[2] - r[0] + 1);
        if (cols * 4 >= (size_t)cs->hs * 3) {                      // nearly full rows: whole rows in one linear copy
            const size_t ofs = (size_t)r[1] * cs->hs;
            SH_HIP(hipMemcpyAsync(dst + ofs, cs->d_hole + ofs, rows * pitch, hipMemcpyDeviceToHost, ctx->stream));
            r[0] = 0; r[2] = cs->hs - 1;
        } else {
            const size_t ofs = (size_t)r[1] * cs->hs + (size_t)r[0];
            SH_HIP(hipMemcpy2DAsync(dst + ofs, pitch, cs->d_hole + ofs, pitch, cols * sizeof(uint16_t), rows, hipMemcpyDeviceToHost, ctx->stream));
        }
        SH_HIP(hipStreamSynchronize(ctx->stream));
        if (!direct) {
            const size_t c2 = (size_t)(r[2] - r[0] + 1) * sizeof(uint16_t);
            for (int y = r[1]; y <= r[3]; y++) memcpy(pix + (size_t)y * cs->hs + r[0], dst + (size_t)y * cs->hs + r[0], c2);
        }
        // (what the caller's array holds no longer equals the asynchronous form's shadow -- in the staged form the staging buffer,
        // in the direct form the array itself was written behind the shadow's back: the next asynchronous request starts from everything)
        cs->mirror_user = nullptr;
    } else { r[0] = r[1] = 0; r[2] = r[3] = -1; }
    if (out_rect) memcpy(out_rect, r, sizeof(r));
    return SLAMHIP_OK;
}

// ---- asynchronous, span-exact host mirror -----------------------------------------------------------------------------------
// `HoleMap.Pixels` is read live by the reference's callers (HoleMap.cs:27; Simulation/MainWindow.xaml.cs:227-249), so a
// source-compatible shim keeps a host mirror.  The blocking form above stalls every scan for the bounding rectangle of the scan
// (7 of 8 MiB at 2048^2: 258 us per scan against 44 without).  This form moves what was DRAWN and does not stall the scan:
//   K2 keeps, per map row, the column span its rays crossed since the last snapshot (k2_row_spans, holemap.hip);
//   slamhip_cs_holemap_mirror_async enqueues, behind the updates on the operator's stream, ONE launch that copies those spans into a
//   shadow map (device to device: microseconds), takes the spans with it and rests them; on a copy stream, behind an event, a
//   second launch pushes the shadow's spans straight into the caller's array -- page-locked and mapped into the device's address
//   space on first use, so the stores travel over PCIe from the lanes, span by span, with no staging and no compaction -- and a
//   16-byte summary follows; the next search starts as soon as the snapshot is taken;
//   slamhip_cs_holemap_mirror_wait (the C# `Pixels` getter) waits for the push.
__global__ void __launch_bounds__(256) k_span_fill(int2 *__restrict__ span, int size, int full)
{
    const int y = blockIdx.x * 256 + threadIdx.x;
    if (y < size) span[y] = full ? make_int2(0, size - 1) : make_int2(size, -1);
}
// One wavefront per row: the row's span in 8-pixel (16-byte) units, map against shadow -- a unit that DIFFERS is copied into the
// shadow and marked in the row's bit mask (a span covers what the rays crossed; what they changed is less: a pixel that is blended
// towards the value it already has -- free space in a mapped area -- keeps it, and only what changed needs to travel).  The span
// moves to the snapshot table and is rested; the summary (bounding rectangle of the changed units, their pixels, rows) through
// wave-level atomics.  `all`: the shadow holds nothing yet (first request, another array): every unit of the span is news.
__global__ void __launch_bounds__(256) k_mirror_snapshot(const uint16_t *__restrict__ map, uint16_t *__restrict__ shadow, int2 *__restrict__ span,
                                                         int2 *__restrict__ snap, unsigned long long *__restrict__ mask, int chunks, int size, int all,
                                                         int lines, int *__restrict__ sum, int2 *__restrict__ rows)
{
    const int lane = threadIdx.x & 63, y = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (y >= size) return;
    const int2 sp = span[y];
    if (lane == 0) { snap[y] = sp; span[y] = make_int2(size, -1); rows[y] = make_int2(size, -1); }
    if (sp.y < sp.x) return;
    const size_t row = (size_t)y * size;
    const int upr = (size + 7) >> 3;                               // units per row (the last may be short: rows that are not whole units go pixel by pixel)
    const bool vec = size % 8 == 0;
    int n_changed = 0, ulo = upr, uhi = -1;
    for (int c = (sp.x >> 3) >> 6; c <= (sp.y >> 3) >> 6; c++) {
        const int u = c * 64 + lane;
        bool ch = false;
        if (u >= (sp.x >> 3) && u <= (sp.y >> 3) && u < upr) {
            if (vec) {
                const uint4 a = ((const uint4 *)(map + row))[u], b = ((const uint4 *)(shadow + row))[u];
                ch = all || a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w;
                if (ch) ((uint4 *)(shadow + row))[u] = a;
            } else {
                for (int x = u * 8; x < u * 8 + 8 && x < size; x++) {
                    const uint16_t a = map[row + x];
                    if (all || a != shadow[row + x]) { ch = true; shadow[row + x] = a; }
                }
            }
        }
        unsigned long long m = __builtin_amdgcn_ballot_w64(ch);
        if (lines && m) {
            // whole 64-byte lines travel: a lone 16-byte store to host memory is a partial-line write over PCIe (a transaction of
            // its own, a read-modify-write at the host's memory controller) -- the four units of a line that holds a changed one
            // are pushed together (the other three equal the shadow already: nothing to copy here)
            unsigned long long g = (m | (m >> 1) | (m >> 2) | (m >> 3)) & 0x1111111111111111ull;
            m = g | (g << 1) | (g << 2) | (g << 3);
        }
        if (lane == 0) mask[(size_t)y * chunks + c] = m;
        if (m) {
            n_changed += __builtin_popcountll(m);
            ulo = min(ulo, c * 64 + (int)__builtin_ctzll(m)); uhi = max(uhi, c * 64 + 63 - (int)__builtin_clzll(m));
        }
    }
    if (lane == 0 && n_changed > 0) {
        rows[y] = make_int2(ulo, uhi);                             // the units of this row that travel (first, last): the host's copy in _wait goes by them
        atomicMin(&sum[0], ulo * 8); atomicMin(&sum[1], y); atomicMax(&sum[2], min(uhi * 8 + 7, size - 1)); atomicMax(&sum[3], y);
        atomicAdd((unsigned long long *)&sum[4], (unsigned long long)n_changed * 8ull);
        atomicAdd(&sum[6], 1);
    }
}
// the shadow's marked units into the caller's array (host memory mapped into the device's address space), a wavefront per row at a time
__global__ void __launch_bounds__(256) k_mirror_push(const uint16_t *__restrict__ shadow, const int2 *__restrict__ snap, const unsigned long long *__restrict__ mask,
                                                     int chunks, uint16_t *__restrict__ host, int size, const int *__restrict__ sum, int *__restrict__ host_sum,
                                                     const int2 *__restrict__ rows, int2 *__restrict__ host_rows)
{
    const int lane = threadIdx.x & 63, wpb = 4;
    if (blockIdx.x == 0 && threadIdx.x < 8) host_sum[threadIdx.x] = sum[threadIdx.x];     // (the snapshot's summary: pinned host words, no copy of its own)
    if (host_rows)                                                 // (staged mirror: the rows' changed ranges for the host's copy)
        for (int y = (int)blockIdx.x * 256 + (int)threadIdx.x; y < size; y += (int)gridDim.x * 256) host_rows[y] = rows[y];
    const int upr = (size + 7) >> 3;
    const bool vec = size % 8 == 0 && ((size_t)host & 15) == 0;
    for (int y = (int)blockIdx.x * wpb + (int)(threadIdx.x >> 6); y < size; y += (int)gridDim.x * wpb) {
        const int2 sp = snap[y];
        if (sp.y < sp.x) continue;
        const size_t row = (size_t)y * size;
        for (int c = (sp.x >> 3) >> 6; c <= (sp.y >> 3) >> 6; c++) {
            const unsigned long long m = mask[(size_t)y * chunks + c];
            const int u = c * 64 + lane;
            if (!((m >> lane) & 1ull) || u >= upr) continue;
            if (vec) ((uint4 *)(host + row))[u] = ((const uint4 *)(shadow + row))[u];
            else for (int x = u * 8; x < u * 8 + 8 && x < size; x++) host[row + x] = shadow[row + x];
        }
    }
}

__global__ void k_mirror_sum_rest(int *__restrict__ sum)
{
    if (threadIdx.x < 8) sum[threadIdx.x] = threadIdx.x < 2 ? 0x7fffffff : threadIdx.x < 4 ? -1 : 0;
}

static int32_t mirror_resources(slamhip_cs *cs)
{
    if (cs->d_hole_span) return SLAMHIP_OK;
    const size_t npix = (size_t)cs->hs * cs->hs;
    SH_HIP(hipMalloc(&cs->d_hole_span, sizeof(int2) * (size_t)cs->hs));
    SH_HIP(hipMalloc(&cs->d_hole_span_snap, sizeof(int2) * (size_t)cs->hs));
    SH_HIP(hipMalloc(&cs->d_hole_shadow, sizeof(uint16_t) * npix));
    cs->mirror_chunks = (((cs->hs + 7) >> 3) + 63) >> 6;           // 64-unit chunks per row: one mask word each
    SH_HIP(hipMalloc(&cs->d_mirror_mask, sizeof(unsigned long long) * (size_t)cs->hs * cs->mirror_chunks));
    SH_HIP(hipMalloc(&cs->d_mirror_sum, sizeof(int) * 8));
    SH_HIP(hipMalloc(&cs->d_mirror_rows, sizeof(int2) * (size_t)cs->hs));
    SH_HIP(hipHostMalloc(&cs->h_mirror_rows, sizeof(int2) * (size_t)cs->hs));
    SH_HIP(hipHostMalloc(&cs->h_mirror_sum, sizeof(int) * 8));
    SH_HIP(hipEventCreateWithFlags(&cs->ev_snap, hipEventDisableTiming));
    SH_HIP(hipEventCreateWithFlags(&cs->ev_push, hipEventDisableTiming));
    return SLAMHIP_OK;
}

// every row's span := the whole row / empty (enqueued on the operator's stream); a no-op while no mirror is kept
int32_t cs_holemap_span_set(slamhip_cs *cs, bool all)
{
    if (!cs->mirror_on) return SLAMHIP_OK;
    hipLaunchKernelGGL(k_span_fill, dim3(sh_div_up(cs->hs, 256)), dim3(256), 0, cs->ctx->stream, cs->d_hole_span, cs->hs, all ? 1 : 0);
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_holemap_mirror_release(slamhip_cs *cs)
{
    SH_CHECK_ARG(cs);
    SH_HIP(hipSetDevice(cs->ctx->device));
    if (cs->mirror_stream) SH_HIP(hipStreamSynchronize(cs->mirror_stream));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    if (cs->mirror_pending) SH_TRY(slamhip_cs_holemap_mirror_wait(cs, nullptr, nullptr));
    if (cs->mirror_reg) { (void)hipHostUnregister(cs->mirror_reg); cs->mirror_reg = nullptr; cs->mirror_reg_bytes = 0; cs->mirror_dev_ptr = nullptr; }
    cs->mirror_user = nullptr;                                     // (the next request, whatever its array, starts from everything)
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_holemap_mirror_wait(slamhip_cs *cs, int32_t out_rect[4], int64_t *out_pixels)
{
    SH_CHECK_ARG(cs);
    SH_HIP(hipSetDevice(cs->ctx->device));
    const int *h = cs->h_mirror_sum;
    if (cs->mirror_pending) {
        SH_HIP(hipEventSynchronize(cs->ev_push));
        cs->mirror_pending = false;
        if (!cs->mirror_direct && cs->mirror_user && h[6] > 0) {
            // staged: the changed units of every row, from the library's pinned buffer into the caller's array
            uint16_t *pix = cs->mirror_user;
            const int S = cs->hs;
            for (int y = h[1]; y <= h[3]; y++) {
                const int2 ru = cs->h_mirror_rows[y];
                if (ru.y < ru.x) continue;
                const size_t x0 = (size_t)ru.x * 8, x1 = std::min((size_t)ru.y * 8 + 8, (size_t)S);
                memcpy(pix + (size_t)y * S + x0, cs->h_mirror_stage + (size_t)y * S + x0, (x1 - x0) * sizeof(uint16_t));
            }
        }
    }
    const bool any = h && h[6] > 0;
    if (out_rect) { out_rect[0] = any ? h[0] : 0; out_rect[1] = any ? h[1] : 0; out_rect[2] = any ? h[2] : -1; out_rect[3] = any ? h[3] : -1; }
    if (out_pixels) *out_pixels = any ? (int64_t)(((uint64_t)(uint32_t)h[5] << 32) | (uint32_t)h[4]) : 0;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_holemap_mirror_async(slamhip_cs *cs, uint16_t *pix, size_t n)
{
    SH_CHECK_ARG(cs && pix && n == (size_t)cs->hs * cs->hs);
    slamhip_ctx *ctx = cs->ctx;
    SH_HIP(hipSetDevice(ctx->device));
    SH_TRY(mirror_resources(cs));
    if (cs->mirror_pending) SH_TRY(slamhip_cs_holemap_mirror_wait(cs, nullptr, nullptr));    // (one push in flight: the shadow and its spans are single)
    // the first call, or another array than the last request's: everything the device holds is news to it
    bool fresh = !cs->mirror_on || cs->mirror_user != pix;
    uint16_t *dst = nullptr; bool direct = false;
    SH_TRY(mirror_target(cs, pix, n, hipHostRegisterMapped, &dst, &direct));
    cs->mirror_user = pix; cs->mirror_direct = direct;
    cs->mirror_on = true;
    if (fresh) SH_TRY(cs_holemap_span_set(cs, true));
    static const int mirror_lines = getenv("SLAMHIP_MIRROR_LINES") ? atoi(getenv("SLAMHIP_MIRROR_LINES")) : 1;
    hipLaunchKernelGGL(k_mirror_sum_rest, dim3(1), dim3(64), 0, ctx->stream, cs->d_mirror_sum);
    hipLaunchKernelGGL(k_mirror_snapshot, dim3(sh_div_up(cs->hs, 4)), dim3(256), 0, ctx->stream, (const uint16_t *)cs->d_hole, cs->d_hole_shadow,
                       cs->d_hole_span, cs->d_hole_span_snap, cs->d_mirror_mask, cs->mirror_chunks, cs->hs, fresh ? 1 : 0,
                       (mirror_lines && cs->hs % 32 == 0) ? 1 : 0, cs->d_mirror_sum, cs->d_mirror_rows);
    SH_HIP(hipGetLastError());
    SH_HIP(hipEventRecord(cs->ev_snap, ctx->stream));
    SH_HIP(hipStreamWaitEvent(cs->mirror_stream, cs->ev_snap, 0));
    static const int push_wgs = getenv("SLAMHIP_MIRROR_WGS") ? atoi(getenv("SLAMHIP_MIRROR_WGS")) : 32;
    hipLaunchKernelGGL(k_mirror_push, dim3(push_wgs > 0 ? push_wgs : 32), dim3(256), 0, cs->mirror_stream, (const uint16_t *)cs->d_hole_shadow,
                       (const int2 *)cs->d_hole_span_snap, (const unsigned long long *)cs->d_mirror_mask, cs->mirror_chunks, dst, cs->hs,
                       (const int *)cs->d_mirror_sum, cs->h_mirror_sum, (const int2 *)cs->d_mirror_rows, direct ? (int2 *)nullptr : cs->h_mirror_rows);
    SH_HIP(hipGetLastError());
    SH_HIP(hipEventRecord(cs->ev_push, cs->mirror_stream));
    cs->mirror_pending = true;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_holemap_download_packed(slamhip_cs *cs, uint8_t *packed, size_t n_bytes)
{
    SH_CHECK_ARG(cs && packed && n_bytes == ((size_t)cs->hs * cs->hs) / 2);
    SH_HIP(hipSetDevice(cs->ctx->device));
    uint8_t *d = nullptr;
    SH_HIP(hipMalloc(&d, n_bytes));
    hipLaunchKernelGGL(k_pack_holemap, dim3((unsigned)((n_bytes + 255) / 256)), dim3(256), 0, cs->ctx->stream, cs->d_hole, d, n_bytes);
    hipError_t e = hipMemcpyAsync(packed, d, n_bytes, hipMemcpyDeviceToHost, cs->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(cs->ctx->stream);
    (void)hipFree(d);
    SH_HIP(e);
    return SLAMHIP_OK;
}
extern "C" int32_t slamhip_cs_obstaclemap_upload(slamhip_cs *cs, const int8_t *pix, size_t n)
{
    SH_CHECK_ARG(cs && pix && n == (size_t)cs->os * cs->os);
    SH_HIP(hipSetDevice(cs->ctx->device));
    SH_TRY(cs_obstacle_flush(cs));
    SH_HIP(hipMemcpyAsync(cs->d_obst, pix, n, hipMemcpyHostToDevice, cs->ctx->stream));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    return SLAMHIP_OK;
}
extern "C" int32_t slamhip_cs_obstaclemap_download(slamhip_cs *cs, int8_t *pix, size_t n)
{
    SH_CHECK_ARG(cs && pix && n == (size_t)cs->os * cs->os);
    SH_HIP(hipSetDevice(cs->ctx->device));
    SH_TRY(cs_obstacle_flush(cs));                                 // (the fused call's cell pass trails one scan behind: obstacle_dev.h)
    SH_HIP(hipMemcpyAsync(pix, cs->d_obst, n, hipMemcpyDeviceToHost, cs->ctx->stream));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    return SLAMHIP_OK;
}

// ---- scan --------------------------------------------------------------------------------------------------
static inline uint32_t part1by1(uint32_t x)
{
    x &= 0x0000ffff;
    x = (x ^ (x << 8)) & 0x00ff00ff;
    x = (x ^ (x << 4)) & 0x0f0f0f0f;
    x = (x ^ (x << 2)) & 0x33333333;
    x = (x ^ (x << 1)) & 0x55555555;
    return x;
}

// slamhip_cs_set_scan in two parts: the blocks a scan lives in (capacity, which of the two device blocks, the staging block free
// again) and the scan's tables.  cs_search_and_update_prelaunched puts the search launch between them.
static int32_t cs_set_scan_begin(slamhip_cs *cs, int32_t n)
{
    g_sst.start();
    cs->n_points = 0;                                  // (stays "no scan" if anything below fails)
    cs->n_rb = 0;
    if (n > cs->cap_points) {
        // one device block and one pinned staging block for everything a scan uploads: rays (original order: K2/K3 are
        // ray-order dependent), rays sorted for K1, K1's per-ray block table, the block starts -- one copy per scan
        SH_HIP(hipStreamSynchronize(cs->ctx->stream));
        SH_TRY(cs_plan_drain(cs));
        (void)hipFree(cs->d_scan_blob); if (cs->h_scan_blob) (void)hipHostFree(cs->h_scan_blob);
        cs->d_scan_blob = nullptr; cs->h_scan_blob = nullptr; cs->cap_points = 0;
        const int cap = n + n / 4 + 64;
        const size_t bytes = (((size_t)cap * (16 + 8 + 8) + sizeof(int) * (size_t)(cap + 2)) + 15) & ~(size_t)15;      // (the upload moves 16-byte units)
        // (two device blocks, used in turn: the upload of scan n + 1 may run -- on the side stream, see ensure_shard -- while the
        // map updates of scan n still read theirs)
        SH_HIP(hipMalloc(&cs->d_scan_blob, 2 * bytes));
        SH_HIP(hipHostMalloc(&cs->h_scan_blob, bytes));
        if (!cs->ev_scan) SH_HIP(hipEventCreateWithFlags(&cs->ev_scan, hipEventDisableTiming));
        cs->scan_blob_bytes = bytes;
        cs->blob_use[0] = cs->blob_use[1] = 0;                  // (fresh blocks; the stream has been drained above)
        cs->cap_points = cap;
        cs->scan_in_flight = false;
    }
    {
        cs->scan_buf ^= 1;
        char *d = (char *)cs->d_scan_blob + (size_t)cs->scan_buf * cs->scan_blob_bytes;
        const size_t cap = (size_t)cs->cap_points;
        cs->d_scan_cur = d;
        cs->d_ray_blk = (int4 *)d;                     d += cap * 16;
        cs->d_pts = (float2 *)d;                       d += cap * 8;
        cs->d_pts_sorted = (float2 *)d;                d += cap * 8;
        cs->d_rb_start = (int *)d;
    }
    if (cs->upload_pending) cs->upload_pending = false; // (the staged scan was never consumed: nothing was launched, the block is ours)
    else if (cs->scan_in_flight) {                      // the previous copy has left the staging block
        if (!cs->ctx->mail_off) SH_TRY(sh_upload_wait(cs->ctx, (volatile uint32_t *)cs->h_key + 28, cs->upload_seq));
        else SH_HIP(hipEventSynchronize(cs->ev_scan));
        cs->scan_in_flight = false;
    }
    cs->k1_scan_dirty = true;
    cs->scan_gen++;
    g_sst.lap(0);
    return SLAMHIP_OK;
}

// force_bar: the search launch that reads this scan's tables is already in the stream (it waits on the device for cs->d_scan_flag):
// the tables are stored through the BAR whatever blob_use says -- the caller has made sure that nothing older reads the block
static int32_t cs_set_scan_finish(slamhip_cs *cs, const float *xy, int32_t n, bool force_bar)
{
    const int cap_ = cs->cap_points;
    int *h_rayblk = (int *)cs->h_scan_blob;
    float *h_pts = (float *)((char *)cs->h_scan_blob + (size_t)cap_ * 16);
    float *sorted = (float *)((char *)cs->h_scan_blob + (size_t)cap_ * 24);
    int *h_rb = (int *)((char *)cs->h_scan_blob + (size_t)cap_ * 32);
    memcpy(h_pts, xy, sizeof(float) * 2 * (size_t)n);
    // K1 sums integers, so it may visit the rays in any order: sort them along a Z-order curve at
    // 64-pixel granularity so that a ray block's end points stay close together in the map (the rigid
    // candidate transform preserves distances), then cut the sorted list into blocks of <= CS_RB_MAX.
    // (This function is the host's critical path between two scans: 12 us at 1080 points before round 4, of which two thirds
    // were libm calls -- fminf / fmaxf are not inlined without fast-math -- and a loop the compiler could not vectorise; now ~4.)
#define SH_MINF(a, b) ((a) < (b) ? (a) : (b))
#define SH_MAXF(a, b) ((a) > (b) ? (a) : (b))
    bool sane = true;
    std::vector<uint64_t> &keys = cs->h_sort_keys;                 // (kept between scans: no allocation per scan)
    static const float cell_px = getenv("SLAMHIP_RB_CELL") ? (float)atof(getenv("SLAMHIP_RB_CELL")) : 64.0f;   // (tuning override: the Z-order's cell, pixels)
    const float icell = cs->hscale / (cell_px >= 8.0f ? cell_px : 64.0f);         // 64-pixel cells per metre (ordering only: any monotone map of the coordinates does)
    // cell coordinates of every point (a branch-free loop over the 2 n floats: vectorised), their minima, and the sanity flag
    std::vector<int> &cellxy = cs->h_cell_xy;
    cellxy.resize((size_t)n * 2);
    int dmax = 0;                                   // (the largest distance of a cell coordinate from 32768: the codes' compression, below)
    {
        int bad = 0, cmin = 65535, cmax = 0;         // (min / max: reductions the loop vectorises with; the distance is V-shaped in the coordinate)
        int *cu = cellxy.data();
        for (int i = 0; i < 2 * n; i++) {
            const float v = xy[i];
            bad |= !(fabsf(v) < 1.0e9f);
            float g = v * icell + 32768.0f;
            g = g > 0.0f ? g : 0.0f;                 // (NaN -> 0)
            g = g < 65535.0f ? g : 65535.0f;
            const int c = (int)g;
            cu[i] = c;
            cmin = c < cmin ? c : cmin;
            cmax = c > cmax ? c : cmax;
        }
        sane = !bad;
        const int dlo = cmin >= 32768 ? cmin - 32768 : 32767 - cmin, dhi = cmax >= 32768 ? cmax - 32768 : 32767 - cmax;
        dmax = dlo > dhi ? dlo : dhi;
    }
    // Morton codes of the cell coordinates.  The order is that of the ABSOLUTE codes -- the scan frame's origin, the robot, is the
    // curve's major boundary: the scan splits into its four quadrants first, then recursively; measured at the headline size against
    // codes relative to the scan's first cell: 18.9 against 24.7 us per search, the blocks' shapes decide what fits a tile -- but
    // the codes are COMPRESSED: with every cell within 2^m of 32768 in both directions, bits m .. 15 of a coordinate all follow
    // bit 15, so subtracting 32768 - 2^m keeps every comparison and leaves 2 (m + 1) bits: two 8-bit passes for scans up to
    // 128 cells (160 m at 2048^2 / 40 m) instead of three 11-bit ones with their 2048-bin prefix sums.
    g_sst.lap(1);
    int mbits = 0;
    while ((1 << mbits) <= dmax) mbits++;                         // every cell in [32768 - 2^m, 32768 + 2^m)
    const bool compressed = mbits <= 10;
    const int shift_c = compressed ? 32768 - (1 << mbits) : 0;
    const int code_bits = compressed ? 2 * (mbits + 1) : 32;
    // The order is that of (code, ray index): unique keys, so any sorting algorithm gives the same permutation.  Scans of up to 2048
    // rays whose compressed code takes at most 21 bits (every scan within 1280 m at the headline scale) sort 32-bit keys -- half
    // the bytes of the general form; this loop is the largest single piece of the host's chain between two scans.
    const bool small = n <= 2048 && code_bits <= 21;
    std::vector<uint32_t> &order = cs->h_sort_order;               // ray index by sorted position
    order.resize((size_t)n);
    if (small) {
        std::vector<uint32_t> &k32 = cs->h_sort_k32;
        k32.resize(2 * (size_t)n);
        uint32_t *src = k32.data(), *dst = k32.data() + n;
        for (int i = 0; i < n; i++) {
            const uint32_t code = part1by1((uint32_t)(cellxy[2 * (size_t)i] - shift_c)) | (part1by1((uint32_t)(cellxy[2 * (size_t)i + 1] - shift_c)) << 1);
            src[i] = (code << 11) | (uint32_t)i;
        }
        g_sst.lap(2);
        // LSD radix sort on the code (stable 8-bit passes over the bits in use; ties keep ray order); the digits' histograms are
        // counted in ONE pass over the keys (three independent increments per key instead of one dependent chain per pass)
        unsigned cnt[3][256];
        memset(cnt, 0, sizeof(cnt));
        for (int i = 0; i < n; i++) { const uint32_t k = src[i] >> 11; cnt[0][k & 255u]++; cnt[1][(k >> 8) & 255u]++; cnt[2][(k >> 16) & 255u]++; }
        for (int dgt = 0; dgt * 8 < code_bits; dgt++) {
            const int shift = 11 + dgt * 8;
            unsigned *c = cnt[dgt];
            if (c[(src[0] >> shift) & 255u] == (unsigned)n) continue;        // every key has the same digit
            unsigned sum = 0;
            for (int k = 0; k < 256; k++) { const unsigned v = c[k]; c[k] = sum; sum += v; }
            for (int i = 0; i < n; i++) dst[c[(src[i] >> shift) & 255u]++] = src[i];
            uint32_t *t = src; src = dst; dst = t;
        }
        for (int j = 0; j < n; j++) order[(size_t)j] = src[j] & 2047u;
    } else {
        keys.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            const uint32_t code = part1by1((uint32_t)(cellxy[2 * (size_t)i] - shift_c)) | (part1by1((uint32_t)(cellxy[2 * (size_t)i + 1] - shift_c)) << 1);
            keys[i] = ((uint64_t)code << 32) | (uint32_t)i;
        }
        g_sst.lap(2);
        std::vector<uint64_t> &tmp = cs->h_sort_tmp;
        tmp.resize((size_t)n);
        uint64_t *src = keys.data(), *dst = tmp.data();
        unsigned cnt[256];
        for (int lo = 0; lo < code_bits; lo += 8) {
            const int shift = 32 + lo;
            memset(cnt, 0, sizeof(cnt));
            for (int i = 0; i < n; i++) cnt[(src[i] >> shift) & 255u]++;
            if (cnt[(src[0] >> shift) & 255u] == (unsigned)n) continue;      // every key has the same digit
            unsigned sum = 0;
            for (int k = 0; k < 256; k++) { const unsigned c = cnt[k]; cnt[k] = sum; sum += c; }
            for (int i = 0; i < n; i++) dst[cnt[(src[i] >> shift) & 255u]++] = src[i];
            uint64_t *t = src; src = dst; dst = t;
        }
        for (int j = 0; j < n; j++) order[(size_t)j] = (uint32_t)src[j];
    }
    g_sst.lap(3);
    std::vector<int> &rb = cs->h_rb_start;
    rb.clear();
    rb.push_back(0);
    int cur = 0;
    float bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
    // Block extent: the tile of a (candidate group, block) is about (extent + translation spread of the candidates + arc)
    // pixels square and should fit the 60 KB LDS tile (~173 px square) -- at fine map scales the spread (known from the
    // last candidate list, else the reference's default sigma of 0.1 m: ~0.7 m) takes a large part of that.  Measured on
    // MI355X at 4096^2 / 32768 candidates: 96 px blocks are 1.28x faster than 128 px ones.
    static const int ext_env = getenv("SLAMHIP_RB_EXTENT") ? atoi(getenv("SLAMHIP_RB_EXTENT")) : 0;   // (tuning override)
    float spread_px = 0.7f * cs->hscale;
    for (size_t g = 0; g < cs->h_grp_dxy.size(); g++) if (g == 0 || cs->h_grp_dxy[g] > spread_px) spread_px = g == 0 ? cs->h_grp_dxy[0] : cs->h_grp_dxy[g];
    float ext_px = 165.0f - spread_px;
    if (!(ext_px <= CS_RB_EXTENT_PX)) ext_px = CS_RB_EXTENT_PX;
    if (ext_px < 64.0f) ext_px = 64.0f;
    if (cs->n_offs + 1 >= 131072) ext_px = CS_RB_EXTENT_PX;   // (very many candidates: two bands of a big block beat more, smaller blocks -- measured)
    if (ext_env > 0) ext_px = (float)ext_env;
    const float ext = ext_px / cs->hscale;                     // block extent limit in metres
    // A block's tile is the box of its end points in the MAP frame, grown by the candidates' translation spread and the arc their
    // theta range sweeps: a block that is large in both directions there -- a corner of the room -- overflows the tile budget even
    // within the extent limit and is staged in bands at more than twice the cost, for every candidate group (seen at the headline
    // size: one such block kept twelve workgroups 17 us in a 19 us launch).  So the limits are tested on the points turned by the
    // last search heading (the layout's; 0 before the first search), and the box area, margins included, CAN be limited too
    // (SLAMHIP_RB_AREA = fraction of the tile budget).  Measured at the headline size, us per search: no area limit 18.9, 0.8 of
    // the budget 20.2 (more, smaller blocks: more tile steps) -- off by default; the ranges' cuts avoid banded pieces instead
    // (k1_cuts_banded_rays, distance.hip).
    static const float area_f = getenv("SLAMHIP_RB_AREA") ? (float)atof(getenv("SLAMHIP_RB_AREA")) : 0.0f;    // of the tile budget; <= 0: no area limit (the default: see below)
    const float rth = cs->k1_layout_theta;
    const float rc = cosf(rth), rs = sinf(rth);
    const float marg = (spread_px + 40.0f) / cs->hscale;       // metres: translation spread + a nominal arc
    const float amax = area_f > 0.0f ? area_f * (60.0f * 1024.0f / 2.0f) / (cs->hscale * cs->hscale) : 3.0e38f;   // square metres
    for (int j = 0; j < n; j++) {
        const int i = (int)order[(size_t)j];
        const float X = xy[2 * i], Y = xy[2 * i + 1];
        sorted[2 * j] = X; sorted[2 * j + 1] = Y;
        const float Xr = rc * X - rs * Y, Yr = rs * X + rc * Y;
        if (cur > 0) {
            const float nx0 = SH_MINF(bx0, Xr), nx1 = SH_MAXF(bx1, Xr), ny0 = SH_MINF(by0, Yr), ny1 = SH_MAXF(by1, Yr);
            const float ex = nx1 - nx0, ey = ny1 - ny0;
            if (cur == CS_RB_MAX || !(ex <= ext) || !(ey <= ext) || !((ex + marg) * (ey + marg) <= amax)) { rb.push_back(j); cur = 0; }
            else { bx0 = nx0; bx1 = nx1; by0 = ny0; by1 = ny1; }
        }
        if (cur == 0) { bx0 = bx1 = Xr; by0 = by1 = Yr; }
        cur++;
    }
    rb.push_back(n);
    g_sst.lap(4);
    cs->n_rb = (int)rb.size() - 1;
    cs->pts_sane = sane;
    // K1's view: per sorted ray its block (a workgroup's chunk is a ray range, cut into pieces at block
    // boundaries), and per block the figures the launch layout is balanced with
    const int n_rb = cs->n_rb;
    int *rayblk = h_rayblk;
    cs->h_rb_ex.resize((size_t)n_rb); cs->h_rb_ey.resize((size_t)n_rb); cs->h_rb_mx.resize((size_t)n_rb); cs->h_rb_my.resize((size_t)n_rb);
    for (int b = 0; b < n_rb; b++) {
        const int r0 = rb[(size_t)b], r1 = rb[(size_t)b + 1];
        float x0 = sorted[2 * (size_t)r0], x1 = x0, y0 = sorted[2 * (size_t)r0 + 1], y1 = y0;
        for (int r = r0; r < r1; r++) {
            const float X = sorted[2 * (size_t)r], Y = sorted[2 * (size_t)r + 1];
            x0 = SH_MINF(x0, X); x1 = SH_MAXF(x1, X); y0 = SH_MINF(y0, Y); y1 = SH_MAXF(y1, Y);
            ((int4 *)rayblk)[r] = make_int4(r0, r1, b, 0);
        }
        cs->h_rb_ex[(size_t)b] = (x1 - x0) * cs->hscale; cs->h_rb_ey[(size_t)b] = (y1 - y0) * cs->hscale;
        cs->h_rb_mx[(size_t)b] = 0.5f * (x0 + x1) * cs->hscale; cs->h_rb_my[(size_t)b] = 0.5f * (y0 + y1) * cs->hscale;
    }
#undef SH_MINF
#undef SH_MAXF
    memcpy(h_rb, rb.data(), sizeof(int) * rb.size());
    g_sst.lap(5);
    const size_t used = (size_t)cap_ * 32 + sizeof(int) * rb.size();
    // (a blocking call that returned through the mailbox leaves a stream the runtime has not yet seen finish; the copy takes
    // its immediate path only on a stream the runtime knows to be idle: one query lets it find out)
    if (cs->ctx->large_bar && (force_bar || cs->blob_use[cs->scan_buf] <= cs->launch_done)) {
        // The device block of this scan is idle (nothing that read it is still running: see launch_done) and the host can store
        // into device memory: the upload is a copy by the CPU through the PCIe aperture -- write-combined, 32 KB in ~1 us,
        // tools/ubench_bar.hip -- and the launches that follow find the data in memory (their doorbell is ordered behind the
        // posted writes; they start with the L2 invalidated).  No upload launch, no flags, nothing in flight.
        const size_t c_ = (size_t)cap_, n_ = (size_t)n;
        const char *hb = (const char *)cs->h_scan_blob;
        memcpy(cs->d_scan_cur, hb, n_ * 16);                                         // ray -> block table
        memcpy(cs->d_scan_cur + c_ * 16, hb + c_ * 16, n_ * 8);                      // rays, original order
        memcpy(cs->d_scan_cur + c_ * 24, hb + c_ * 24, n_ * 8);                      // rays, sorted
        memcpy(cs->d_scan_cur + c_ * 32, hb + c_ * 32, sizeof(int) * rb.size());     // block starts
        __builtin_ia32_sfence();
        cs->upload_pending = false;
    } else if (!cs->ctx->mail_off) {
        // The upload is a launch that pulls the staging block and then tells the host (h_key words 28 .. 31, one per workgroup of the upload) that it may be refilled.
        // It is left pending: the first launch that reads the scan issues it (cs_flush_scan) -- or the candidate gather of the
        // coming search carries it as an extra workgroup (ensure_shard): one launch and one launch boundary less per scan in the
        // processor's flow (upload -> gather -> K1 -> ...).
        cs->upload_pending = true;
        cs->upload_bytes = (used + 15) & ~(size_t)15;
    } else {
        SH_HIP(hipMemcpyAsync(cs->d_scan_cur, cs->h_scan_blob, used, hipMemcpyHostToDevice, cs->ctx->stream));
        cs_plan_inputs_pending(cs);
        SH_HIP(hipEventRecord(cs->ev_scan, cs->ctx->stream));
        cs->scan_in_flight = true;
    }
    cs->n_points = n;
    g_sst.lap(6); g_sst.done();
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_set_scan(slamhip_cs *cs, const float *xy, int32_t n)
{
    SH_CHECK_ARG(cs && n >= 0 && (xy || n == 0));
    SH_HIP(hipSetDevice(cs->ctx->device));
    if (n == 0) { cs->n_points = 0; cs->n_rb = 0; return SLAMHIP_OK; }
    SH_TRY(cs_set_scan_begin(cs, n));
    return cs_set_scan_finish(cs, xy, n, false);
}

// The search may be put into the operator's stream once the side stream's launch (ensure_shard) has COMPLETED: the host watches
// for that -- it has nothing else to do, the map updates of the previous scan are still running in the operator's stream -- rather
// than tying the streams with an event (measured: event record + cross-stream wait cost 10 us per scan, twice what the launch beside
// the updates saves).
thread_local cs_stage_times g_cst("fused scan");
thread_local cs_stage_times g_sst("set_scan");

int32_t cs_side_join(slamhip_cs *cs)
{
    if (!cs->side_join) return SLAMHIP_OK;
    cs->side_join = false;
    volatile uint32_t *flag = (volatile uint32_t *)cs->h_key + 24;
    for (int spins = 0; spins < 200000; spins++) { if (*flag == cs->side_seq) return SLAMHIP_OK; __builtin_ia32_pause(); }
    SH_HIP(hipStreamSynchronize(cs->side_stream));             // (far past any launch of this kind: let the runtime report what happened)
    if (*flag == cs->side_seq) return SLAMHIP_OK;
    SH_FAIL(SLAMHIP_ERR_HIP, "the side stream is idle but its completion word never arrived");
}

int32_t cs_flush_scan(slamhip_cs *cs)
{
    cs->blob_use[cs->scan_buf] = ++cs->launch_count;             // (called by every launcher that reads the scan)
    if (!cs->upload_pending) return SLAMHIP_OK;
    // (the upload state is committed once the launch that carries it is in the stream: on an error the scan stays pending)
    SH_TRY(sh_upload(cs->ctx, cs->h_scan_blob, cs->d_scan_cur, cs->upload_bytes, (uint32_t *)cs->h_key + 28, cs->upload_seq + 1));
    cs_plan_inputs_pending(cs);
    cs->upload_pending = false;
    cs->upload_seq++;
    cs->scan_in_flight = true;
    return SLAMHIP_OK;
}

// ---- distance ---------------------------------------------------------------------------------------------
static int32_t finish_distance(slamhip_cs *cs, int K, int32_t *out_dist, int32_t *out_best_index, int32_t *out_best_dist)
{
    slamhip_ctx *ctx = cs->ctx;
    sh_mail_guard lock(ctx);
    uint64_t key;
    if (out_dist) {
        SH_HIP(hipMemcpyAsync(cs->h_key, cs->d_key, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        SH_HIP(hipMemcpyAsync(out_dist, cs->d_dist, sizeof(int32_t) * (size_t)K, hipMemcpyDeviceToHost, ctx->stream));
        SH_HIP(hipStreamSynchronize(ctx->stream));
        key = *cs->h_key;
    } else {
        SH_TRY(sh_publish(ctx, cs->d_key, 2));
        SH_TRY(sh_host_wait(ctx));
        key = *(volatile uint64_t *)ctx->mailbox;
    }
    if (out_best_index) *out_best_index = (int32_t)(uint32_t)key;
    if (out_best_dist) *out_best_dist = (int32_t)(uint32_t)(key >> 32);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_distance_pxcs(slamhip_cs *cs, const float *pxcs, int32_t K, int32_t *out_dist,
                                            int32_t *out_best_index, int32_t *out_best_dist)
{
    SH_CHECK_ARG(cs && pxcs && K > 0);
    SH_HIP(hipSetDevice(cs->ctx->device));
    if (cs->n_points <= 0) SH_FAIL(SLAMHIP_ERR_STATE, "no scan set (slamhip_cs_set_scan)");
    SH_TRY(cs_alloc_candidates(cs, K));
    bool sane = true;
    for (size_t i = 0; i < (size_t)K * 4; i++) if (!(fabsf(pxcs[i]) < 1.0e9f)) { sane = false; break; }
    cs->shard_first = cs->shard_count = -1;        // evaluation buffers no longer hold the offset shard
    SH_HIP(hipMemcpyAsync(cs->d_pxcs, pxcs, sizeof(float) * 4 * (size_t)K, hipMemcpyHostToDevice, cs->ctx->stream));
    int *saved = cs->d_ev_idx; cs->d_ev_idx = nullptr;          // identity: evaluation order == flat order
    int32_t rc = cs_launch_distance(cs, 0, nullptr, K, out_dist != nullptr, sane, cs->d_key);
    cs->d_ev_idx = saved;
    SH_TRY(rc);
    return finish_distance(cs, K, out_dist, out_best_index, out_best_dist);
}

extern "C" int32_t slamhip_cs_distance_poses(slamhip_cs *cs, const float *poses, int32_t K, int32_t *out_dist,
                                             int32_t *out_best_index, int32_t *out_best_dist)
{
    SH_CHECK_ARG(cs && poses && K > 0);
    SH_HIP(hipSetDevice(cs->ctx->device));
    if (cs->n_points <= 0) SH_FAIL(SLAMHIP_ERR_STATE, "no scan set (slamhip_cs_set_scan)");
    SH_TRY(cs_alloc_candidates(cs, K));
    bool sane = true;
    for (size_t i = 0; i < (size_t)K * 3; i++) if (!(fabsf(poses[i]) < (i % 3 == 2 ? 6.0e4f : 1.0e6f))) { sane = false; break; }
    cs->shard_first = cs->shard_count = -1;
    // stage the poses in d_ev_off (same 3-float layout)
    SH_HIP(hipMemcpyAsync(cs->d_ev_off, poses, sizeof(float) * 3 * (size_t)K, hipMemcpyHostToDevice, cs->ctx->stream));
    int *saved = cs->d_ev_idx; cs->d_ev_idx = nullptr;
    int32_t rc = cs_launch_distance(cs, 2, nullptr, K, out_dist != nullptr, sane, cs->d_key);
    cs->d_ev_idx = saved;
    SH_TRY(rc);
    return finish_distance(cs, K, out_dist, out_best_index, out_best_dist);
}

// ---- offsets ------------------------------------------------------------------------------------------------
static int32_t ensure_offsets_capacity(slamhip_cs *cs, int n)
{
    SH_TRY(cs_alloc_candidates(cs, n + 1));
    if (cs->d_offs_flat && n <= cs->cap_offs) return SLAMHIP_OK;
    if (cs->d_offs_flat) { (void)hipFree(cs->d_offs_flat); cs->d_offs_flat = nullptr; cs->cap_offs = 0; }
    const int cap = n > 0 ? n + n / 8 : 1;
    SH_HIP(hipMalloc(&cs->d_offs_flat, sizeof(float) * 3 * (size_t)cap));
    cs->cap_offs = cap;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_set_offsets(slamhip_cs *cs, const float *offs, int32_t n)
{
    SH_CHECK_ARG(cs && n >= 0 && (offs || n == 0));
    SH_HIP(hipSetDevice(cs->ctx->device));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    SH_TRY(ensure_offsets_capacity(cs, n));
    cs->n_offs = n;
    cs->h_offs.assign(offs, offs + (size_t)n * 3);
    cs->offs_theta_small = true;
    for (size_t i = 0; i < (size_t)n * 3; i++) if (!(fabsf(offs[i]) < (i % 3 == 2 ? 1.0e4f : 1.0e6f))) { cs->offs_theta_small = false; break; }
    cs->offs_on_device_sorted = false; cs->gen_pending = false; cs->gen_lattice = false; cs->k1_lattice = 0;
    cs->spec_valid = false; cs->spec_base_ok = false;
    cs->shard_first = cs->shard_count = -1;
    if (n > 0) SH_HIP(hipMemcpy(cs->d_offs_flat, offs, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice));
    return SLAMHIP_OK;
}

// the lattice of a full-range search over n jitters (host mirror of what the kernels are told): none unless asked for
static k_lattice cs_lattice(const slamhip_cs *cs, int n, bool on)
{
    k_lattice L = { 0, 0, 0, 0 };
    if (!on || n + 1 <= 12288) return L;                          // (small searches run one candidate per lane -- 512-candidate groups, faster there: nothing to share)
    L.count = n + 1;
    L.grp = L.count >= 65536 ? K1_GROUP_BIG : K1_GROUP;            // (never the 512-candidate groups: a lane needs two candidates to share anything)
    L.cpl = L.grp == K1_GROUP_BIG ? 4 : 2;
    L.zero_pos = L.count - 1 < n / 2 ? L.count - 1 : n / 2;
    return L;
}

static int32_t cs_generate(slamhip_cs *cs, int32_t n, float sigma_xy, float sigma_theta, uint64_t seed, uint64_t stream, bool lattice);
extern "C" int32_t slamhip_cs_generate_offsets(slamhip_cs *cs, int32_t n, float sigma_xy, float sigma_theta,
                                               uint64_t seed, uint64_t stream)
{
    return cs_generate(cs, n, sigma_xy, sigma_theta, seed, stream, false);
}
extern "C" int32_t slamhip_cs_generate_offsets_lattice(slamhip_cs *cs, int32_t n, float sigma_xy, float sigma_theta,
                                                       uint64_t seed, uint64_t stream)
{
    return cs_generate(cs, n, sigma_xy, sigma_theta, seed, stream, true);
}
static int32_t cs_generate(slamhip_cs *cs, int32_t n, float sigma_xy, float sigma_theta, uint64_t seed, uint64_t stream, bool lattice)
{
    SH_CHECK_ARG(cs && n >= 0);
    SH_HIP(hipSetDevice(cs->ctx->device));
    if (n != cs->n_offs || !cs->d_offs_flat) {
        SH_HIP(hipStreamSynchronize(cs->ctx->stream));
        SH_TRY(ensure_offsets_capacity(cs, n));
    }
    // (bit patterns: the sigmas must be the very floats the prepared list was made from)
    const bool hit = cs->spec_valid && n == cs->spec_n && n == cs->n_offs && memcmp(&sigma_xy, &cs->spec_sxy, 4) == 0 &&
                     memcmp(&sigma_theta, &cs->spec_sth, 4) == 0 && seed == cs->spec_seed && stream == cs->spec_stream &&
                     cs->spec_cap_offs == cs->cap_offs && cs->spec_cap_cand == cs->cap_cand && cs->spec_cap_grp == cs->cap_grp &&
                     cs->offs_on_device_sorted && cs->shard_first == 0 && cs->shard_count == n + 1 && cs->k1_group == cs->spec_grp &&
                     lattice == cs->spec_lattice && lattice == cs->gen_lattice;
    cs->spec_valid = false;
    if (hit) {
        // the list asked for has been prepared (cs_speculate_next): swap the buffer sets -- the search that read the old set has
        // delivered its result -- and leave everything ensure_shard derived from (n, sigmas, group size) as it is
        // (THREE sets in rotation: the set just searched cools for one scan before a list is prepared into it again -- its
        // jitters are still read by the map update that decodes that scan's winner, and what proves THAT launch finished is a pose
        // delivered two scans later; the set a list is prepared into was last read two scans ago)
        { float *t_ = cs->d_offs_flat; cs->d_offs_flat = cs->spec_offs_flat; cs->spec_offs_flat = cs->cool_offs_flat; cs->cool_offs_flat = t_; }
        { float *t_ = cs->d_ev_off; cs->d_ev_off = cs->spec_ev_off; cs->spec_ev_off = cs->cool_ev_off; cs->cool_ev_off = t_; }
        { int *t_ = cs->d_ev_idx; cs->d_ev_idx = cs->spec_ev_idx; cs->spec_ev_idx = cs->cool_ev_idx; cs->cool_ev_idx = t_; }
        { float *t_ = cs->d_grp_bounds; cs->d_grp_bounds = cs->spec_grp_bounds; cs->spec_grp_bounds = cs->cool_grp_bounds; cs->cool_grp_bounds = t_; }
        cs->gen_stream = stream; cs->gen_pending = false; cs->spec_hits++;
        cs->h_offs.clear();                                        // (the host copy, if one was fetched, held the PREVIOUS list's jitters: host_offsets() must fetch again)
        cs->side_join = true;
        return cs_side_join(cs);                                  // (the side launch's word: there since the previous scan -- checked before anything reads the list)
    }
    cs->spec_base_ok = false;
    cs->n_offs = n;
    cs->h_offs.clear();
    cs->offs_theta_small = fabsf(sigma_theta) < 1.0e3f && fabsf(sigma_xy) < 1.0e5f;     // |normcdfinvf| < 6 for float quantiles
    cs->gen_sigma_xy = sigma_xy; cs->gen_sigma_theta = sigma_theta;
    cs->offs_on_device_sorted = true;
    cs->shard_first = cs->shard_count = -1;
    // produced on first use: a full-range search generates the list inside its gather launch (ensure_shard)
    cs->gen_pending = n > 0; cs->gen_seed = seed; cs->gen_stream = stream;
    cs->gen_lattice = lattice;
    return SLAMHIP_OK;
}

// The processor's flow asks for a fresh candidate list per scan -- slamhip_cs_generate_offsets(n, sigmas, seed, stream) with the
// stream counting up -- and the list depends on nothing else.  A fused scan that is about to wait for its pose therefore prepares
// the NEXT list (same parameters, stream + 1) into a second set of buffers, on a stream of its own, beside the search and the map
// updates in flight; when the next generate_offsets call asks for exactly that list the sets are swapped and the search finds
// its candidates in place: one launch (3.5 us of host time, 5 us in the operator's stream) less between two scans.  A call with
// other parameters simply drops the prepared list.  The side launch reports through a pinned word (k_side_arrive) which the
// search launch checks (cs_side_join); by then it has long arrived.
static void spec_free(slamhip_cs *cs)
{
    (void)hipFree(cs->spec_offs_flat); (void)hipFree(cs->spec_ev_off); (void)hipFree(cs->spec_ev_idx); (void)hipFree(cs->spec_grp_bounds);
    (void)hipFree(cs->cool_offs_flat); (void)hipFree(cs->cool_ev_off); (void)hipFree(cs->cool_ev_idx); (void)hipFree(cs->cool_grp_bounds);
    cs->spec_offs_flat = nullptr; cs->spec_ev_off = nullptr; cs->spec_ev_idx = nullptr; cs->spec_grp_bounds = nullptr;
    cs->cool_offs_flat = nullptr; cs->cool_ev_off = nullptr; cs->cool_ev_idx = nullptr; cs->cool_grp_bounds = nullptr;
    cs->spec_cap_offs = cs->spec_cap_cand = cs->spec_cap_grp = 0;
}

static int32_t cs_speculate_next(slamhip_cs *cs)
{
    slamhip_ctx *ctx = cs->ctx;
    static const bool off = getenv("SLAMHIP_NO_SPECULATION") != nullptr;
    const int n = cs->n_offs;
    cs->spec_valid = false;
    if (off || !cs->spec_base_ok || !cs->offs_on_device_sorted || cs->gen_pending || n <= 0 || cs->shard_first != 0 || cs->shard_count != n + 1 ||
        ctx->timing != 0 || ctx->mail_off) return SLAMHIP_OK;
    if (!cs->d_side_arrive) {
        SH_HIP(hipMalloc(&cs->d_side_arrive, 64));
        SH_HIP(hipMemset(cs->d_side_arrive, 0, 64));
    }
    if (cs->spec_cap_offs != cs->cap_offs || cs->spec_cap_cand != cs->cap_cand || cs->spec_cap_grp != cs->cap_grp) {
        SH_HIP(hipStreamSynchronize(cs->side_stream));
        SH_TRY(cs_plan_drain(cs));
        spec_free(cs);
        SH_HIP(hipMalloc(&cs->spec_offs_flat, sizeof(float) * 3 * (size_t)cs->cap_offs));
        SH_HIP(hipMalloc(&cs->spec_ev_off, sizeof(float) * 3 * (size_t)cs->cap_cand));
        SH_HIP(hipMalloc(&cs->spec_ev_idx, sizeof(int) * (size_t)cs->cap_cand));
        SH_HIP(hipMalloc(&cs->spec_grp_bounds, sizeof(float) * 8 * (size_t)cs->cap_grp));
        SH_HIP(hipMalloc(&cs->cool_offs_flat, sizeof(float) * 3 * (size_t)cs->cap_offs));
        SH_HIP(hipMalloc(&cs->cool_ev_off, sizeof(float) * 3 * (size_t)cs->cap_cand));
        SH_HIP(hipMalloc(&cs->cool_ev_idx, sizeof(int) * (size_t)cs->cap_cand));
        SH_HIP(hipMalloc(&cs->cool_grp_bounds, sizeof(float) * 8 * (size_t)cs->cap_grp));
        cs->spec_cap_offs = cs->cap_offs; cs->spec_cap_cand = cs->cap_cand; cs->spec_cap_grp = cs->cap_grp;
    }
    const int count = n + 1, grp = cs->k1_group, ng = sh_div_up(count, grp);
    const int zero_pos = count - 1 < n / 2 ? count - 1 : n / 2;
    hipLaunchKernelGGL(k_gather_offsets<true>, dim3(ng), dim3(1024), 0, cs->side_stream,
                       cs->spec_offs_flat, (const int *)nullptr, 0, count, zero_pos, cs->spec_ev_off, cs->spec_ev_idx, cs->spec_grp_bounds, grp,
                       n, cs->gen_sigma_xy, cs->gen_sigma_theta, cs->gen_seed, cs->gen_stream + 1, (const uint4 *)nullptr, (uint4 *)nullptr, 0,
                       (uint32_t *)nullptr, 0u, cs->d_side_arrive, (uint32_t *)cs->h_key + 24, cs->side_seq + 1, cs_lattice(cs, n, cs->gen_lattice));
    SH_HIP(hipGetLastError());
    cs->side_seq++; cs->spec_made++;
    cs->spec_valid = true; cs->spec_n = n; cs->spec_sxy = cs->gen_sigma_xy; cs->spec_sth = cs->gen_sigma_theta;
    cs->spec_seed = cs->gen_seed; cs->spec_stream = cs->gen_stream + 1; cs->spec_grp = grp; cs->spec_lattice = cs->gen_lattice;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_prepared_lists(slamhip_cs *cs, uint64_t *out_served, uint64_t *out_prepared)
{
    SH_CHECK_ARG(cs);
    if (out_served) *out_served = cs->spec_hits;
    if (out_prepared) *out_prepared = cs->spec_made;
    return SLAMHIP_OK;
}

int32_t cs_flush_generate(slamhip_cs *cs)
{
    if (!cs->gen_pending) return SLAMHIP_OK;
    cs->gen_pending = false;
    hipLaunchKernelGGL(k_generate_offsets, dim3(sh_div_up(cs->n_offs, 256)), dim3(256), 0, cs->ctx->stream,
                       cs->d_offs_flat, cs->n_offs, cs->gen_sigma_xy, cs->gen_sigma_theta, cs->gen_seed, cs->gen_stream,
                       cs_lattice(cs, cs->n_offs, cs->gen_lattice));
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_offsets_download(slamhip_cs *cs, float *offs, int32_t n)
{
    SH_CHECK_ARG(cs && offs && n == cs->n_offs);
    SH_HIP(hipSetDevice(cs->ctx->device));
    SH_TRY(cs_flush_generate(cs));
    if (n > 0) {
        SH_HIP(hipMemcpyAsync(offs, cs->d_offs_flat, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, cs->ctx->stream));
        SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    }
    return SLAMHIP_OK;
}

static int32_t host_offsets(slamhip_cs *cs)
{
    if (cs->h_offs.size() == (size_t)cs->n_offs * 3) return SLAMHIP_OK;
    cs->h_offs.resize((size_t)cs->n_offs * 3);
    SH_TRY(cs_flush_generate(cs));
    if (cs->n_offs > 0) {
        SH_HIP(hipMemcpyAsync(cs->h_offs.data(), cs->d_offs_flat, sizeof(float) * 3 * (size_t)cs->n_offs,
                              hipMemcpyDeviceToHost, cs->ctx->stream));
        SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    }
    return SLAMHIP_OK;
}

// quantile function of N(0,1) (Acklam's rational approximation, |error| < 1.2e-9): host-side layout heuristics only
static double host_normcdfinv(double p)
{
    static const double a[6] = { -3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                                 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00 };
    static const double b[5] = { -5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                                 6.680131188771972e+01, -1.328068155288572e+01 };
    static const double c[6] = { -7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                                 -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00 };
    static const double d[4] = { 7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00 };
    if (p < 0.02425) {
        const double q = sqrt(-2.0 * log(p));
        return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.0);
    }
    if (p > 1.0 - 0.02425) return -host_normcdfinv(1.0 - p);
    const double q = p - 0.5, r = q * q;
    return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
           (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0);
}

// materialise the theta-sorted evaluation list of flat candidates [first, first+count)
static int32_t ensure_shard(slamhip_cs *cs, int first, int count)
{
    if (cs->shard_first == first && cs->shard_count == count) return SLAMHIP_OK;
    slamhip_ctx *ctx = cs->ctx;
    // Candidates per group -- the candidates that share a tile: 1024 (512 lanes x 2); 2048 (512 lanes x 4) from 65 536 candidates
    // on -- a tile then serves twice the candidates, and the theta tails, whose groups widen, are a small part of the launch
    // (measured, 1024 -> 2048, us per launch: 1 M candidates 489 -> 395, 262 144: 152 -> 137, 98 304: 74 -> 64, 81 920: 66 ->
    // 55, 65 536: 51 -> 51 (4096^2 map: 83 -> 72), 49 152: 45.5 -> 44.6 (4096^2: 72 -> 77, 1024^2: 37.1 -> 38.5), 32 768: 34.7 -> 34.9; 16 384: 25 -> 33); 512 (512 lanes
    // x 1) up to 12 288 candidates -- the groups' theta ranges halve, which is worth more there than the tiles' reuse (1024 ->
    // 512: 4096 candidates 22.7 -> 18.8, the simulator's 4000 candidates on a 256^2 map with 400 rays 14.6 -> 13.3, 8192:
    // 23.2 -> 22.8 (1024^2 map: 22.0 -> 19.0), 12 288: 25.7 -> 24.0, but 16 384: 25.3 -> 26.6).
    static const int grp_env = getenv("SLAMHIP_K1_GROUP") ? atoi(getenv("SLAMHIP_K1_GROUP")) : 0;
    // (a heading lattice is laid out for the group size of its full-range search: cs_lattice)
    const k_lattice lat = cs_lattice(cs, cs->n_offs, cs->offs_on_device_sorted && cs->gen_lattice);
    const bool lat_list = lat.cpl != 0;
    const int grp = lat_list ? lat.grp
                    : grp_env == K1_GROUP || grp_env == K1_GROUP_BIG || grp_env == K1_GROUP_SMALL ? grp_env
                    : count >= 65536 ? K1_GROUP_BIG : count <= 12288 ? K1_GROUP_SMALL : K1_GROUP;
    cs->k1_group = grp;
    cs->k1_lattice = lat_list && first == 0 && count == cs->n_offs + 1 ? lat.cpl : 0;     // (the search kernel's lanes line up with the lattice)
    const int ng = sh_div_up(count, grp);
    // (the launch layout is made from these per-group figures: it stays valid when a new list leaves them as they were -- the
    // per-scan flow regenerates the list with the same sigmas -- see the end of this function)
    cs->h_grp_prev.clear();
    cs->h_grp_prev.insert(cs->h_grp_prev.end(), cs->h_grp_dth.begin(), cs->h_grp_dth.end());
    cs->h_grp_prev.insert(cs->h_grp_prev.end(), cs->h_grp_dxy.begin(), cs->h_grp_dxy.end());
    cs->h_grp_prev.insert(cs->h_grp_prev.end(), cs->h_grp_lohi.begin(), cs->h_grp_lohi.end());
    cs->h_grp_prev.push_back((float)cs->k1_group_prev);
    cs->k1_group_prev = grp;
    cs->h_grp_dth.assign((size_t)ng, 0.0f); cs->h_grp_dxy.assign((size_t)ng, 0.0f);
    cs->h_grp_lohi.assign((size_t)ng * 6, 0.0f);                   // per group: min / max of dx, dy, dtheta (host lists: exact; generated lists: from the quantiles)
    auto figures_changed = [cs, grp]() {
        const std::vector<float> &p = cs->h_grp_prev;
        const size_t a = cs->h_grp_dth.size(), b = cs->h_grp_dxy.size(), c = cs->h_grp_lohi.size();
        if (p.size() != a + b + c + 1 || p.back() != (float)grp) return true;
        return memcmp(p.data(), cs->h_grp_dth.data(), 4 * a) != 0 || memcmp(p.data() + a, cs->h_grp_dxy.data(), 4 * b) != 0 ||
               memcmp(p.data() + a + b, cs->h_grp_lohi.data(), 4 * c) != 0;
    };
    cs_plan_inputs_pending(cs);                                    // (every branch below launches the gather in the operator's stream)
    // a pending scan upload rides on the gather launch as one more workgroup
    const int up_wg = cs->upload_pending ? SH_UPLOAD_PARTS : 0;
    const uint4 *up_src = nullptr; uint4 *up_dst = nullptr; int up_n16 = 0; uint32_t *up_flag = nullptr; uint32_t up_seq = 0;
    if (up_wg) {                                                   // (committed below, once the gather launch is in the stream)
        up_src = (const uint4 *)cs->h_scan_blob; up_dst = (uint4 *)cs->d_scan_cur; up_n16 = (int)(cs->upload_bytes / 16);
        up_flag = (uint32_t *)cs->h_key + 28; up_seq = cs->upload_seq + 1;
    }
    if (cs->offs_on_device_sorted) {
        // flat candidates first .. first+count-1 = the un-jittered pose (flat 0) and jitters in ascending dtheta
        const int n = cs->n_offs;
        const int zero_pos = first == 0 ? (count - 1 < n / 2 ? count - 1 : n / 2) : -1;
        if (cs->gen_pending && first == 0 && count == n + 1) {
            cs->gen_pending = false;
            hipLaunchKernelGGL(k_gather_offsets<true>, dim3(ng + up_wg), dim3(1024), 0, ctx->stream,
                               cs->d_offs_flat, (const int *)nullptr, first, count, zero_pos, cs->d_ev_off, cs->d_ev_idx, cs->d_grp_bounds, grp,
                               n, cs->gen_sigma_xy, cs->gen_sigma_theta, cs->gen_seed, cs->gen_stream, up_src, up_dst, up_n16, up_flag, up_seq,
                               (unsigned *)nullptr, (uint32_t *)nullptr, 0u, lat);
            cs->spec_base_ok = true;                              // (the full generated list is in place: cs_speculate_next may prepare its successor)
        } else {
            SH_TRY(cs_flush_generate(cs));
            hipLaunchKernelGGL(k_gather_offsets<false>, dim3(ng + up_wg), dim3(1024), 0, ctx->stream,
                               cs->d_offs_flat, (const int *)nullptr, first, count, zero_pos, cs->d_ev_off, cs->d_ev_idx, cs->d_grp_bounds, grp,
                               0, 0.f, 0.f, (uint64_t)0, (uint64_t)0, up_src, up_dst, up_n16, up_flag, up_seq, (unsigned *)nullptr, (uint32_t *)nullptr, 0u, lat);
        }
        // the jitters are the strata of N(0, sigma): group ranges from the quantile function (layout balance only)
        for (int g = 0; g < ng; g++) {
            const int k0 = first + g * grp, k1 = (first + count < k0 + grp ? first + count : k0 + grp) - 1;
            double q0 = fmin(fmax((k0 - 0.5) / (n > 0 ? n : 1), 0.5 / (n + 1)), 1.0 - 0.5 / (n + 1));
            double q1 = fmin(fmax((k1 + 0.5) / (n > 0 ? n : 1), 0.5 / (n + 1)), 1.0 - 0.5 / (n + 1));
            if (cs->k1_lattice) {                                  // (a lattice group holds the strata of its lane positions, each cpl times)
                const int lanes = grp / lat.cpl, full = lat.count / grp, rest = lat.count - full * grp;
                const double U = (double)(full * lanes + (rest < lanes ? rest : lanes));
                q0 = fmin(fmax((double)g * lanes / U, 0.5 / (n + 1)), 1.0 - 0.5 / (n + 1));
                q1 = fmin(fmax((double)(g + 1) * lanes / U, 0.5 / (n + 1)), 1.0 - 0.5 / (n + 1));
            }
            cs->h_grp_dth[(size_t)g] = (float)(fabs(cs->gen_sigma_theta) * (host_normcdfinv(q1) - host_normcdfinv(q0)));
            cs->h_grp_dxy[(size_t)g] = 7.0f * fabsf(cs->gen_sigma_xy) * cs->hscale;
            float *lh = &cs->h_grp_lohi[(size_t)g * 6];
            lh[0] = lh[2] = -3.5f * fabsf(cs->gen_sigma_xy); lh[1] = lh[3] = 3.5f * fabsf(cs->gen_sigma_xy);
            lh[4] = (float)(fabs(cs->gen_sigma_theta) * host_normcdfinv(q0)); lh[5] = (float)(fabs(cs->gen_sigma_theta) * host_normcdfinv(q1));
        }
    } else {
        // theta = search_pose.Z + dtheta and float addition is monotone, so sorting by dtheta sorts by theta
        std::vector<int> perm((size_t)count);
        std::iota(perm.begin(), perm.end(), first);
        const float *o = cs->h_offs.data();
        std::stable_sort(perm.begin(), perm.end(), [o](int a, int b) {
            const float ta = a > 0 ? o[3 * (size_t)(a - 1) + 2] : 0.0f, tb = b > 0 ? o[3 * (size_t)(b - 1) + 2] : 0.0f;
            return ta < tb;
        });
        for (int g = 0; g < ng; g++) {
            float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
            for (int j = g * grp; j < count && j < (g + 1) * grp; j++) {
                const int flat = perm[(size_t)j];
                for (int k = 0; k < 3; k++) {
                    const float v = flat > 0 ? o[3 * (size_t)(flat - 1) + k] : 0.0f;
                    lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v);
                }
            }
            cs->h_grp_dth[(size_t)g] = hi[2] - lo[2];
            for (int k = 0; k < 3; k++) { cs->h_grp_lohi[(size_t)g * 6 + 2 * k] = lo[k]; cs->h_grp_lohi[(size_t)g * 6 + 2 * k + 1] = hi[k]; }
            cs->h_grp_dxy[(size_t)g] = fmaxf(hi[0] - lo[0], hi[1] - lo[1]) * cs->hscale;
        }
        SH_HIP(hipMemcpyAsync(cs->d_ev_idx, perm.data(), sizeof(int) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_gather_offsets<false>, dim3(ng + up_wg), dim3(1024), 0, ctx->stream,
                           cs->d_offs_flat, (const int *)cs->d_ev_idx, first, count, -1, cs->d_ev_off, (int *)nullptr, cs->d_grp_bounds, grp,
                           0, 0.f, 0.f, (uint64_t)0, (uint64_t)0, up_src, up_dst, up_n16, up_flag, up_seq,
                           (unsigned *)nullptr, (uint32_t *)nullptr, 0u, lat);
        const hipError_t le = hipGetLastError();
        if (le == hipSuccess && up_wg) { cs->upload_pending = false; cs->upload_seq = up_seq; cs->scan_in_flight = true; }
        SH_HIP(le);
        SH_HIP(hipStreamSynchronize(ctx->stream));      // perm dies here
        cs->shard_first = first; cs->shard_count = count;
        cs->k1_layout_dirty = true;
        return SLAMHIP_OK;
    }
    if (figures_changed()) cs->k1_layout_dirty = true;
    SH_HIP(hipGetLastError());
    if (up_wg) { cs->upload_pending = false; cs->upload_seq = up_seq; cs->scan_in_flight = true; }
    cs->shard_first = first; cs->shard_count = count;
    return SLAMHIP_OK;
}

static int32_t search_enqueue(slamhip_cs *cs, const float pose[3], int first, int count, uint64_t *key_dst)
{
    SH_CHECK_ARG(cs && pose);
    SH_CHECK_ARG(first >= 0 && count > 0 && first + count <= cs->n_offs + 1);
    SH_HIP(hipSetDevice(cs->ctx->device));
    if (cs->n_points <= 0) SH_FAIL(SLAMHIP_ERR_STATE, "no scan set (slamhip_cs_set_scan)");
    g_cst.start();
    SH_TRY(ensure_shard(cs, first, count));
    g_cst.lap(0);
    const bool sane = fabsf(pose[0]) < 1.0e6f && fabsf(pose[1]) < 1.0e6f && fabsf(pose[2]) < 1.0e4f && cs->offs_theta_small;
    return cs_launch_distance(cs, 1, pose, count, false, sane, key_dst);
}

extern "C" int32_t slamhip_cs_search_shard_async(slamhip_cs *cs, const float pose[3], int32_t first, int32_t count,
                                                 uint64_t *d_out_key)
{
    SH_CHECK_ARG(cs && d_out_key);
    return search_enqueue(cs, pose, first, count, d_out_key);     // prep arms the key, K1/K1r min into it
}

// Enqueue-only form whose result word the handle owns (a ring of K1_RING_SLOTS words): K1 needs no last finisher for it -- the
// finishing workgroups min straight into the word, which the previous ring launch left all ones (distance.hip, ring mode).
extern "C" int32_t slamhip_cs_search_shard_enqueue(slamhip_cs *cs, const float pose[3], int32_t first, int32_t count,
                                                   const uint64_t **d_key)
{
    SH_CHECK_ARG(cs && d_key);
    cs->k1_ring_request = true;
    const int32_t rc = search_enqueue(cs, pose, first, count, nullptr);
    cs->k1_ring_request = false;
    SH_TRY(rc);
    *d_key = cs->k1_ring_last;
    return SLAMHIP_OK;
}

// Waits for the handle's stream and reads one result word of the ring (or any 8-byte device word of this context).
extern "C" int32_t slamhip_cs_key_read(slamhip_cs *cs, const uint64_t *d_key, uint64_t *out_key)
{
    SH_CHECK_ARG(cs && d_key && out_key);
    SH_HIP(hipSetDevice(cs->ctx->device));
    SH_HIP(hipMemcpyAsync(cs->h_key + 8, d_key, sizeof(uint64_t), hipMemcpyDeviceToHost, cs->ctx->stream));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    *out_key = cs->h_key[8];
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_search_shard(slamhip_cs *cs, const float pose[3], int32_t first, int32_t count, uint64_t *out_key)
{
    SH_CHECK_ARG(cs && out_key);
    slamhip_ctx *ctx = cs->ctx;
    sh_mail_guard lock(ctx);
    if (!ctx->mail_off) {
        // the launch ends with the key and the completion word into the mailbox (tiled kernel), or a publish launch
        // follows the fallback kernels.  The sequence number is taken when the kernel is armed: the value K1 stores is the value
        // waited for, whatever publishes in between; the fallback's publish launch reuses it.
        const uint32_t seq = sh_mail_seq_next(ctx);
        cs->k1_done_flag = ctx->mailbox + 15; cs->k1_done_val = seq;
        const int32_t rc = search_enqueue(cs, pose, first, count, cs->d_key);
        cs->k1_done_flag = nullptr;
        SH_TRY(rc);
        if (!cs->k1_done_armed) SH_TRY(sh_publish_seq(ctx, cs->d_key, 2, seq));
        cs_layout_idle_refresh(cs);                                // (host work under the search)
        SH_TRY(sh_flag_wait(ctx, ctx->mailbox + 15, seq));
        *out_key = *(volatile uint64_t *)ctx->mailbox;
        return SLAMHIP_OK;
    }
    SH_TRY(search_enqueue(cs, pose, first, count, cs->d_key));
    SH_HIP(hipMemcpyAsync(cs->h_key, cs->d_key, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    SH_HIP(hipStreamSynchronize(ctx->stream));
    *out_key = *cs->h_key;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_pose_from_key(slamhip_cs *cs, const float pose[3], uint64_t key, float out_pose[3],
                                            int32_t *out_dist, int32_t *out_index)
{
    SH_CHECK_ARG(cs && pose);
    const uint32_t flat = (uint32_t)key;
    SH_CHECK_ARG(flat <= (uint32_t)cs->n_offs);
    SH_HIP(hipSetDevice(cs->ctx->device));
    SH_TRY(host_offsets(cs));
    if (out_pose) {
        out_pose[0] = pose[0]; out_pose[1] = pose[1]; out_pose[2] = pose[2];       // :626
        if (flat > 0) {
            const float *o = cs->h_offs.data() + 3 * (size_t)(flat - 1);
            out_pose[0] = pose[0] + o[0];                                          // :635
            out_pose[1] = pose[1] + o[1];                                          // :636
            out_pose[2] = pose[2] + o[2];                                          // :637
        }
    }
    if (out_dist) *out_dist = (int32_t)(uint32_t)(key >> 32);
    if (out_index) *out_index = (int32_t)flat;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_search(slamhip_cs *cs, const float pose[3], float out_pose[3], int32_t *out_dist, int32_t *out_index)
{
    SH_CHECK_ARG(cs);
    uint64_t key = 0;
    SH_TRY(slamhip_cs_search_shard(cs, pose, 0, cs->n_offs + 1, &key));
    return slamhip_cs_pose_from_key(cs, pose, key, out_pose, out_dist, out_index);
}

// ---- map updates ---------------------------------------------------------------------------------------------
static float4 pxcs_from_pose(const float pose[3], float scale)
{
    float s, c;
    sh_det_sincosf(pose[2], &s, &c);
    float4 q;
    q.x = pose[0] * scale + 0.5f;
    q.y = pose[1] * scale + 0.5f;
    q.z = c * scale;
    q.w = s * scale;
    return q;
}

static int32_t finish_holemap(slamhip_cs *cs)
{
    sh_mail_guard lock(cs->ctx);
    SH_TRY(sh_publish(cs->ctx, cs->d_k2_counters, 4));
    SH_TRY(sh_host_wait(cs->ctx));
    const int *m = (const int *)cs->ctx->mailbox;                           // [0] longest ray, [1] conflict pixels, [2] blended pixels
    cs->last_hole_pixels = m[2]; cs->hole_pixels_pending = false;
    static const bool stats = getenv("SLAMHIP_K2_STATS") != nullptr;        // developer aid
    if (stats) fprintf(stderr, "[slamhip] K2: reach %d px, %d blended pixels\n", m[0], m[2]);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_update_holemap_pxcs(slamhip_cs *cs, const float pxcs[4], float hole_width, int32_t quality)
{
    SH_CHECK_ARG(cs && pxcs && quality >= 0 && quality <= 256);
    SH_HIP(hipSetDevice(cs->ctx->device));
    cs->last_hole_pixels = 0; cs->hole_pixels_pending = false;
    if (cs->n_points <= 0) return SLAMHIP_OK;
    SH_TRY(cs_launch_holemap_update(cs, nullptr, make_float4(pxcs[0], pxcs[1], pxcs[2], pxcs[3]), make_float4(0, 0, 0, 0), hole_width, quality));
    return finish_holemap(cs);
}

extern "C" int32_t slamhip_cs_update_holemap(slamhip_cs *cs, const float pose[3], float hole_width, int32_t quality)
{
    SH_CHECK_ARG(cs && pose);
    const float4 q = pxcs_from_pose(pose, cs->hscale);                      // :499-502
    const float a[4] = { q.x, q.y, q.z, q.w };
    return slamhip_cs_update_holemap_pxcs(cs, a, hole_width, quality);
}

extern "C" int32_t slamhip_cs_update_obstaclemap_pxcs(slamhip_cs *cs, const float pxcs[4], int32_t max_hits)
{
    SH_CHECK_ARG(cs && pxcs && max_hits >= -128 && max_hits <= 127);
    SH_HIP(hipSetDevice(cs->ctx->device));
    if (cs->n_points <= 0) return SLAMHIP_OK;
    sh_mail_guard lock(cs->ctx);
    SH_TRY(cs_launch_obstacle_update(cs, nullptr, make_float4(pxcs[0], pxcs[1], pxcs[2], pxcs[3]), max_hits));
    SH_TRY(sh_publish(cs->ctx, nullptr, 0));
    return sh_host_wait(cs->ctx);
}

extern "C" int32_t slamhip_cs_update_obstaclemap(slamhip_cs *cs, const float pose[3], int32_t max_hits)
{
    SH_CHECK_ARG(cs && pose);
    const float4 q = pxcs_from_pose(pose, cs->oscale);                      // :545-548
    const float a[4] = { q.x, q.y, q.z, q.w };
    return slamhip_cs_update_obstaclemap_pxcs(cs, a, max_hits);
}

// both map updates from one pose, enqueued / completed separately so that a multi-GPU caller can overlap its replicas
int32_t cs_update_maps_enqueue(slamhip_cs *cs, const float pose[3], float hole_width, int32_t quality, int32_t max_hits)
{
    SH_CHECK_ARG(cs && pose && quality >= 0 && quality <= 256 && max_hits >= -128 && max_hits <= 127);
    SH_HIP(hipSetDevice(cs->ctx->device));
    cs->last_hole_pixels = 0; cs->hole_pixels_pending = false;
    if (cs->n_points <= 0) return SLAMHIP_OK;
    SH_TRY(cs_launch_holemap_update(cs, nullptr, pxcs_from_pose(pose, cs->hscale), make_float4(0, 0, 0, 0), hole_width, quality));   // :499-502
    SH_TRY(cs_launch_obstacle_update(cs, nullptr, pxcs_from_pose(pose, cs->oscale), max_hits));            // :545-548
    return SLAMHIP_OK;
}
int32_t cs_update_maps_finish(slamhip_cs *cs)
{
    SH_HIP(hipSetDevice(cs->ctx->device));
    if (cs->n_points <= 0) return SLAMHIP_OK;
    return finish_holemap(cs);
}

extern "C" int32_t slamhip_cs_last_holemap_pixels(slamhip_cs *cs, int64_t *out)
{
    SH_CHECK_ARG(cs && out);
    if (cs->hole_pixels_pending) {                                  // the fused call returned with the pose: fetch the count now
        SH_HIP(hipSetDevice(cs->ctx->device));
        int *h = (int *)(cs->h_key + 8);
        SH_HIP(hipMemcpyAsync(h, (const int *)cs->d_key + 6, sizeof(int), hipMemcpyDeviceToHost, cs->ctx->stream));
        SH_HIP(hipStreamSynchronize(cs->ctx->stream));
        cs->last_hole_pixels = *h;
        cs->hole_pixels_pending = false;
    }
    *out = cs->last_hole_pixels;
    return SLAMHIP_OK;
}

// Replica check: checksums of the two maps as they stand behind everything enqueued so far, into words 4
// (HoleMap) and 5 (ObstacleMap) of the result block; see sh_mix64 / k_checksum in common.h for the definition.
int32_t cs_maps_checksum_enqueue(slamhip_cs *cs)
{
    slamhip_ctx *ctx = cs->ctx;
    SH_TRY(cs_obstacle_flush(cs));                                  // (the fused call's cell pass trails one scan behind)
    unsigned long long *d = (unsigned long long *)cs->d_key + 4;
    SH_HIP(hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), ctx->stream));
    sh_checksum_launch<uint16_t>(ctx, cs->d_hole, (size_t)cs->hs * cs->hs, d);
    sh_checksum_launch<uint8_t>(ctx, cs->d_obst, (size_t)cs->os * cs->os, d + 1);
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}
extern "C" int32_t slamhip_cs_maps_checksum(slamhip_cs *cs, uint64_t out[2])
{
    SH_CHECK_ARG(cs && out);
    slamhip_ctx *ctx = cs->ctx;
    SH_HIP(hipSetDevice(ctx->device));
    sh_mail_guard lock(ctx);
    SH_TRY(cs_maps_checksum_enqueue(cs));
    SH_TRY(sh_publish(ctx, cs->d_key + 4, 4));
    SH_TRY(sh_host_wait(ctx));
    memcpy(out, (const void *)ctx->mailbox, 2 * sizeof(uint64_t));
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_prelaunch_stats(slamhip_cs *cs, uint64_t out[4])
{
    SH_CHECK_ARG(cs && out);
    for (int k = 0; k < 4; k++) out[k] = cs->pl_stats[k];
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_plan_stats(slamhip_cs *cs, uint64_t out[4])
{
    SH_CHECK_ARG(cs && out);
    for (int k = 0; k < 4; k++) out[k] = cs->plan_stats[k];
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_selfcheck_failures(slamhip_cs *cs, uint32_t *out)
{
    SH_CHECK_ARG(cs && out);
    SH_HIP(hipSetDevice(cs->ctx->device));
    unsigned int *h = (unsigned int *)(cs->h_key + 8);
    SH_HIP(hipMemcpyAsync(h, cs->d_verify, sizeof(unsigned int) * 8, hipMemcpyDeviceToHost, cs->ctx->stream));
    SH_HIP(hipStreamSynchronize(cs->ctx->stream));
    *out = h[0];
    if (getenv("SLAMHIP_K1_STATS"))
        fprintf(stderr, "[slamhip] K1 step kinds (ray x sub-batch units): group tile %u, sub-batch tiles %u, banded %u, global gathers %u; plan: workgroups with their record %u, without %u\n",
                h[1], h[2], h[4], h[3], h[5], h[6]);
    return SLAMHIP_OK;
}

// Both map updates of one pose from the caller's (px, py, c, s) at either scale (slamhip.h): enqueued back to back, one wait.
extern "C" int32_t slamhip_cs_update_maps_pxcs(slamhip_cs *cs, const float pxcs_hole[4], const float pxcs_obstacle[4], float hole_width,
                                               int32_t quality, int32_t max_obstacle_hits)
{
    SH_CHECK_ARG(cs && pxcs_hole && quality >= 0 && quality <= 256 && max_obstacle_hits >= -128 && max_obstacle_hits <= 127);
    SH_HIP(hipSetDevice(cs->ctx->device));
    cs->last_hole_pixels = 0; cs->hole_pixels_pending = false;
    if (cs->n_points <= 0) return SLAMHIP_OK;
    SH_TRY(cs_launch_holemap_update(cs, nullptr, make_float4(pxcs_hole[0], pxcs_hole[1], pxcs_hole[2], pxcs_hole[3]), make_float4(0, 0, 0, 0), hole_width, quality));   // :750
    if (pxcs_obstacle)
        SH_TRY(cs_launch_obstacle_update(cs, nullptr, make_float4(pxcs_obstacle[0], pxcs_obstacle[1], pxcs_obstacle[2], pxcs_obstacle[3]), max_obstacle_hits));     // :751
    return finish_holemap(cs);
}

// The fused scan with the caller's (px, py, c, s): slamhip.h.  A composition of the _pxcs operators -- what it adds is the contract
// (one call, the winner's rows taken from the caller's own arrays, the update rows being those of the NORMALISED pose), not speed:
// the candidates cross PCIe.
extern "C" int32_t slamhip_cs_search_and_update_pxcs(slamhip_cs *cs, const float *pxcs_search, const float *pxcs_update_hole,
                                                     const float *pxcs_update_obstacle, int32_t K,
                                                     float hole_width, int32_t quality, int32_t max_obstacle_hits,
                                                     int32_t *out_index, int32_t *out_dist)
{
    SH_CHECK_ARG(cs && pxcs_search && pxcs_update_hole && K > 0);
    SH_CHECK_ARG(quality >= 0 && quality <= 256 && max_obstacle_hits >= -128 && max_obstacle_hits <= 127);
    int32_t bi = 0, bd = 0;
    SH_TRY(slamhip_cs_distance_pxcs(cs, pxcs_search, K, nullptr, &bi, &bd));        // (:732; the packed key's minimum IS the first strict minimum in flat order)
    if (bi < 0 || bi >= K) SH_FAIL(SLAMHIP_ERR_STATE, "the search returned candidate %d of %d", (int)bi, (int)K);
    SH_TRY(slamhip_cs_update_maps_pxcs(cs, pxcs_update_hole + 4 * (size_t)bi, pxcs_update_obstacle ? pxcs_update_obstacle + 4 * (size_t)bi : nullptr,
                                       hole_width, quality, max_obstacle_hits));    // (:750-751)
    if (out_index) *out_index = bi;
    if (out_dist) *out_dist = bd;
    return SLAMHIP_OK;
}

// CoreSLAMProcessor.Update's scan (:723 set_scan, :732 search, :746-751 updates) with the SEARCH LAUNCH FIRST.  Between two scans the
// host's chain -- the pose comes back, the next scan is converted, sorted, cut into blocks and stored, the search is launched -- is a
// few microseconds longer than the map update the device is still busy with, and a launch that arrives when the queue has just run
// dry starts 5 us late (profiles/r05_timeline_csproc.txt).  The search's launch parameters do not depend on the new scan's tables
// -- the layout is the last scan's (cs_launch_distance), the buffers alternate, the ray count is known -- only its DATA does: so the
// launch goes into the stream first, behind the running map update, the host makes the tables while that update finishes, and the
// launch waits ON THE DEVICE for the word the host stores behind the tables (k1_search_tiled: a.scan_flag).  Afterwards the host
// tests what the launch assumed -- the layout legal for the new blocks, the points sane -- and otherwise tells the launch to leave
// (the word with bit 31 set) and searches again in the ordinary order.  *took = false: nothing was done.
int32_t cs_search_and_update_prelaunched(slamhip_cs *cs, const float *xy, int32_t n, const float pose[3], float hole_width, int32_t quality,
                                         int32_t max_hits, float out_pose[3], int32_t *out_dist, int32_t *out_index, bool *took)
{
    *took = false;
    static const bool on = getenv("SLAMHIP_PRELAUNCH") ? atoi(getenv("SLAMHIP_PRELAUNCH")) != 0 : true;
    static const bool wait_updates = getenv("SLAMHIP_FUSED_WAIT_UPDATES") != nullptr, k1_delivers = getenv("SLAMHIP_FUSED_K1_DELIVERS") != nullptr;
    slamhip_ctx *ctx = cs->ctx;
    if (g_cst.on) {                                               // (developer aid: why scans take the ordinary order)
        static thread_local unsigned why[8], calls;
        why[0] += n != cs->n_points; why[1] += cs->k1_layout_dirty; why[2] += !cs->pts_sane; why[3] += cs->upload_pending; why[4] += cs->scan_in_flight;
        why[5] += !(cs->blob_use[cs->scan_buf ^ 1] <= cs->launch_done); why[6] += !cs_holemap_one_launch(cs);
        if ((++calls & 255u) == 0) fprintf(stderr, "[slamhip] prelaunch refused in %u calls: ray count %u, layout dirty %u, points %u, upload pending %u, scan in flight %u, block busy %u, two-launch update %u\n",
                                           calls, why[0], why[1], why[2], why[3], why[4], why[5], why[6]);
    }
    if (!on || !xy || n <= 0 || n != cs->n_points || n > cs->cap_points || !ctx->large_bar || ctx->mail_off || ctx->timing != 0 || wait_updates || k1_delivers ||
        cs->k1_layout_dirty || !cs->pts_sane || cs->n_offs <= 0 || cs->upload_pending || cs->scan_in_flight || !cs_holemap_one_launch(cs)) { cs->pl_stats[3]++; return SLAMHIP_OK; }
    SH_CHECK_ARG(quality >= 0 && quality <= 256 && max_hits >= -128 && max_hits <= 127);
    SH_HIP(hipSetDevice(ctx->device));
    // the device block the new scan goes into must be idle BEFORE the search that reads it is launched (the launch marks it in use)
    if (!(cs->blob_use[cs->scan_buf ^ 1] <= cs->launch_done)) { cs->pl_stats[3]++; return SLAMHIP_OK; }
    if (!cs->d_scan_flag) {
        // (fine-grained device memory: the word is polled by a running launch while the host stores it through the BAR -- in an ordinary
        // allocation the poll is served from the L2 for tens of microseconds after the store has landed)
        SH_HIP(hipExtMallocWithFlags((void **)&cs->d_scan_flag, 64, hipDeviceMallocFinegrained));
        SH_HIP(hipMemsetAsync(cs->d_scan_flag, 0, 64, ctx->stream));
        SH_HIP(hipStreamSynchronize(ctx->stream));
        cs->scan_flag_seq = 0;
    }
    *took = true;
    SH_TRY(cs_set_scan_begin(cs, n));
    cs->n_points = n;                                              // (the launch's ray count; the tables follow)
    cs->scan_flag_seq = (cs->scan_flag_seq + 1) & 0x7fffffffu;
    if (cs->scan_flag_seq == 0) cs->scan_flag_seq = 1;
    // (the word through the BAR, fenced on both sides: behind the tables' stores, in front of whatever the host does next)
    auto answer = [cs](uint32_t v) { __builtin_ia32_sfence(); *(volatile uint32_t *)cs->d_scan_flag = v; __builtin_ia32_sfence(); };
    int32_t rc_f = SLAMHIP_OK;
    bool abandoned = false;
    {
        sh_mail_guard lock(ctx);
        const uint32_t seq = sh_mail_seq_next(ctx);
        cs->k1_ring_request = true; cs->k1_prelaunch = true;
        const int32_t rc_r = search_enqueue(cs, pose, 0, cs->n_offs + 1, cs->d_key);
        cs->k1_prelaunch = false;
        if (g_cst.on && rc_r == CS_RC_NO_PRELAUNCH) { static thread_local unsigned nn; if ((++nn & 15u) == 0) fprintf(stderr, "[slamhip] prelaunch: %u launches needed a new layout\n", nn); }
        if (rc_r != CS_RC_NO_PRELAUNCH) {
            cs->k1_ring_request = false;
            if (rc_r != SLAMHIP_OK) { cs->n_points = 0; return rc_r; }   // (nothing was launched)
            g_cst.lap(3);
            rc_f = cs_set_scan_finish(cs, xy, n, true);
            if (rc_f != SLAMHIP_OK || !cs->pts_sane || !cs_k1_layout_legal(cs)) {
                answer(cs->scan_flag_seq | 0x80000000u);           // the launch leaves; its result word stays rested
                abandoned = true;
                cs->pl_stats[1]++;
            }
            if (!abandoned) {
            answer(cs->scan_flag_seq);
            cs->pl_stats[0]++;
            if (g_cst.on && (cs->scan_flag_seq & 63u) == 0) fprintf(stderr, "[slamhip] prelaunched searches: spins of the first workgroup, last %u, sum %u\n", ((volatile uint32_t *)cs->d_scan_flag)[8], ((volatile uint32_t *)cs->d_scan_flag)[9]);
            cs->k1_scan_dirty = false;                             // (the launch is out; the layout for this scan is made in the idle refresh below)
            cs_k2_winner win;
            win.d_key = cs->k1_ring_last; win.d_offs_flat = cs->d_offs_flat; win.n_offs = cs->n_offs; win.bx = pose[0]; win.by = pose[1]; win.bth = pose[2];
            win.mail = ctx->mailbox; win.seq = seq;
            const int32_t rc_u = cs_launch_holemap_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), hole_width, quality, true, max_hits, &win);
            g_cst.lap(4);
            cs_layout_idle_refresh(cs);
            g_cst.lap(5);
            SH_TRY(rc_u);
            const int32_t rc_p = cs_speculate_next(cs);
            g_cst.lap(7);
            SH_TRY(sh_flag_wait(ctx, ctx->mailbox + 15, seq));
            g_cst.lap(6); g_cst.done();
            SH_TRY(rc_p);
            cs->hole_pixels_pending = true;
            cs->launch_done = cs->k1_launch_no;
            const volatile uint64_t *hk = (const volatile uint64_t *)ctx->mailbox;
            const float *hp = (const float *)(ctx->mailbox + 2);
            const uint64_t key = hk[0];
            if (key == ~0ull) {
                // The search launch left without searching: it gave up waiting for this scan's tables (a host stalled for ~10 s between
                // the launch and the answer above -- a debugger, a suspended process), its result word stayed at rest, and the map update
                // behind it decoded that to the un-searched search pose.  The maps are no longer what the reference would hold: this is
                // reported like any wait that passed its bound, and the context refuses further work.
                ctx->poisoned = true;
                SH_FAIL(SLAMHIP_ERR_TIMEOUT, "the search launched ahead of its scan's tables gave up waiting for them (host stalled); the maps were updated at the un-searched pose: the context is poisoned");
            }
            if (out_pose) { out_pose[0] = hp[0]; out_pose[1] = hp[1]; out_pose[2] = hp[2]; }
            if (out_dist) *out_dist = (int32_t)(uint32_t)(key >> 32);
            if (out_index) *out_index = (int32_t)(uint32_t)key;
            return SLAMHIP_OK;
            }
        }
        cs->k1_ring_request = false;
    }
    if (abandoned) {                                               // (the tables stand: the ordinary search over them; its layout is made first)
        if (rc_f != SLAMHIP_OK) cs->n_points = 0;                  // (... unless making them failed: no scan)
        SH_TRY(rc_f);
        return slamhip_cs_search_and_update(cs, pose, hole_width, quality, max_hits, out_pose, out_dist, out_index);
    }
    cs->pl_stats[2]++;
    // the layout has to be remade for the new scan: the ordinary order (nothing was launched -- the block was found idle above, whatever
    // the refused launch has noted in blob_use since)
    SH_TRY(cs_set_scan_finish(cs, xy, n, true));
    return slamhip_cs_search_and_update(cs, pose, hole_width, quality, max_hits, out_pose, out_dist, out_index);
}

extern "C" int32_t slamhip_cs_scan_search_and_update(slamhip_cs *cs, const float *xy, int32_t n, const float pose[3], float hole_width, int32_t quality,
                                                     int32_t max_hits, float out_pose[3], int32_t *out_dist, int32_t *out_index)
{
    SH_CHECK_ARG(cs && pose && n >= 0 && (xy || n == 0));
    bool took = false;
    SH_TRY(cs_search_and_update_prelaunched(cs, xy, n, pose, hole_width, quality, max_hits, out_pose, out_dist, out_index, &took));
    if (took) return SLAMHIP_OK;
    SH_TRY(slamhip_cs_set_scan(cs, xy, n));                            // :723
    return slamhip_cs_search_and_update(cs, pose, hole_width, quality, max_hits, out_pose, out_dist, out_index);   // :732, :746-751
}

extern "C" int32_t slamhip_cs_search_and_update(slamhip_cs *cs, const float pose[3], float hole_width, int32_t quality,
                                                int32_t max_hits, float out_pose[3], int32_t *out_dist, int32_t *out_index)
{
    SH_CHECK_ARG(cs && pose);
    SH_CHECK_ARG(quality >= 0 && quality <= 256 && max_hits >= -128 && max_hits <= 127);
    slamhip_ctx *ctx = cs->ctx;
    // search (:732) -- the launch also leaves the winner's pose, theta normalised (:746-747), on the device -- then both
    // map updates from that pose (:750-751).  The call returns when the POSE is known: K1's final arriver stores key and pose
    // into the mailbox, the updates are already enqueued behind it and run on while the caller prepares its next scan
    // (everything that reads the maps afterwards -- the next search, a download, an export -- is ordered behind them on the
    // operator's stream).  Without the mailbox, with per-kernel timing on, or when the search ran on the fallback kernels,
    // one result block comes back after the updates instead.
    static const bool wait_updates = getenv("SLAMHIP_FUSED_WAIT_UPDATES") != nullptr;      // (the former behaviour, for comparison)
    const bool early = !ctx->mail_off && ctx->timing == 0 && !wait_updates;
    sh_mail_guard lock(ctx);
    const uint32_t seq = sh_mail_seq_next(ctx);                    // (taken when K1 is armed; the result block's publish reuses it otherwise)
    // The winner decoded by the map update (round 4, late): the search runs in its result-ring form -- its workgroups min their keys
    // into a word and the end of the launch is the completion; no final arriver, whose chain (minimum acknowledged, count, minimum
    // read back, jitter loaded: four dependent round trips behind the launch's slowest workgroup) made the search 2.2 us longer in
    // this call than in the headline loop -- and every workgroup of the HoleMap update, which must start with the pose anyway,
    // decodes it from the key (two loads in a row where it had one, under the 1.3 us its sixteen wavefronts take to start); its
    // first workgroup stores the pose for later readers and delivers key + pose to the mailbox.  One-launch updates only
    // (scans of up to 2048 rays); SLAMHIP_FUSED_K1_DELIVERS=1 keeps the search's own delivery.
    static const bool k1_delivers = getenv("SLAMHIP_FUSED_K1_DELIVERS") != nullptr;
    const bool decode = early && !k1_delivers && cs_holemap_one_launch(cs) && cs->n_points > 0;
    if (decode) {
        cs->k1_ring_request = true;
        const int32_t rc_r = search_enqueue(cs, pose, 0, cs->n_offs + 1, cs->d_key);
        cs->k1_ring_request = false;                               // (a search that failed before its launch must not leave the request to the next one)
        SH_TRY(rc_r);
        g_cst.lap(3);
        cs_k2_winner win;
        win.d_key = cs->k1_ring_last; win.d_offs_flat = cs->d_offs_flat; win.n_offs = cs->n_offs; win.bx = pose[0]; win.by = pose[1]; win.bth = pose[2];
        win.mail = ctx->mailbox; win.seq = seq;
        const int32_t rc_u = cs_launch_holemap_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), hole_width, quality, true, max_hits, &win);
        g_cst.lap(4);
        cs_layout_idle_refresh(cs);                                // (host work under the search: the launch layout for the next scan)
        g_cst.lap(5);
        SH_TRY(rc_u);                                              // (no update launch, no delivery: nothing to wait for)
        const int32_t rc_p = cs_speculate_next(cs);                // (... and the next scan's candidates, on the side stream)
        g_cst.lap(7);
        SH_TRY(sh_flag_wait(ctx, ctx->mailbox + 15, seq));
        g_cst.lap(6); g_cst.done();
        SH_TRY(rc_p);
        cs->hole_pixels_pending = true;
        cs->launch_done = cs->k1_launch_no;                        // (the update that delivered runs behind the search: every launch before THAT has finished)
        const volatile uint64_t *hk = (const volatile uint64_t *)ctx->mailbox;
        const float *hp = (const float *)(ctx->mailbox + 2);
        const uint64_t key = hk[0];
        if (out_pose) { out_pose[0] = hp[0]; out_pose[1] = hp[1]; out_pose[2] = hp[2]; }
        if (out_dist) *out_dist = (int32_t)(uint32_t)(key >> 32);
        if (out_index) *out_index = (int32_t)(uint32_t)key;
        return SLAMHIP_OK;
    }
    if (early) { cs->k1_done_flag = ctx->mailbox + 15; cs->k1_done_val = seq; }
    cs->k1_want_pose = true;
    const int32_t rc_s = search_enqueue(cs, pose, 0, cs->n_offs + 1, cs->d_key);
    cs->k1_want_pose = false; cs->k1_done_flag = nullptr;
    SH_TRY(rc_s);
    g_cst.lap(3);
    const bool delivered = early && cs->k1_done_armed && cs->k1_pose_written;
    if (!cs->k1_pose_written)                                    // (fallback search kernels: decode the key in a launch of its own)
        hipLaunchKernelGGL(k_best_pose, dim3(1), dim3(1), 0, ctx->stream, (const unsigned long long *)cs->d_key,
                           cs->d_offs_flat, pose[0], pose[1], pose[2], cs->d_best_pose);
    // :750-751 -- the two maps are independent: the ObstacleMap update rides on the HoleMap update's ONE launch, inside its
    // wavefronts -- this scan's ray walks, and the cell pass of the previous scan (obstacle_dev.h); with per-kernel timing on,
    // each update keeps its own launches so that the timers mean what they say
    int32_t rc_u;
    if (ctx->timing == 0) {
        rc_u = cs_launch_holemap_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), hole_width, quality, true, max_hits);
    } else {
        rc_u = cs_launch_holemap_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), hole_width, quality);
        if (rc_u == SLAMHIP_OK) rc_u = cs_launch_obstacle_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), max_hits);
    }
    g_cst.lap(4);
    cs_layout_idle_refresh(cs);                                  // (host work under the search: the launch layout for the next scan)
    g_cst.lap(5);
    if (delivered) {
        const int32_t rc_p = cs_speculate_next(cs);              // (host work under the search: the next scan's candidates, on the side stream)
        g_cst.lap(7);
        SH_TRY(sh_flag_wait(ctx, ctx->mailbox + 15, seq));       // (the search's word arrives whatever became of the update launches)
        g_cst.lap(6); g_cst.done();
        SH_TRY(rc_p);
        SH_TRY(rc_u);
        cs->hole_pixels_pending = true;
        cs->launch_done = cs->k1_launch_no;                      // (the search has delivered: every launch before it has finished)
    } else {
        SH_TRY(rc_u);
        // (K1 armed but without the pose -- it then stored `seq` with the key alone: the result block follows under a fresh number)
        const uint32_t seq2 = early && cs->k1_done_armed ? sh_mail_seq_next(ctx) : seq;
        SH_TRY(sh_publish_seq(ctx, cs->d_key, 8, seq2));
        if (ctx->mail_off) SH_TRY(sh_host_wait(ctx)); else SH_TRY(sh_flag_wait(ctx, ctx->mailbox + 15, seq2));
        cs->last_hole_pixels = ((const int *)ctx->mailbox)[6]; cs->hole_pixels_pending = false;
    }
    const volatile uint64_t *hk = (const volatile uint64_t *)ctx->mailbox;
    const float *hp = (const float *)(ctx->mailbox + 2);
    const uint64_t key = hk[0];
    if (out_pose) { out_pose[0] = hp[0]; out_pose[1] = hp[1]; out_pose[2] = hp[2]; }
    if (out_dist) *out_dist = (int32_t)(uint32_t)(key >> 32);
    if (out_index) *out_index = (int32_t)(uint32_t)key;
    return SLAMHIP_OK;
}
