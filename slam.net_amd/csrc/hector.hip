// hector.hip -- K4 (Gauss-Newton scan matcher) and K5 (log-odds grid update) + HectorSLAM entry points.
//
// K4 replaces ScanMatcher.MatchData / EstimateTransformationLogLh / GetCompleteHessianDerivs /
// InterpMapValueWithDerivatives (HectorSLAM/Matcher/ScanMatcher.cs:41-249): all pyramid levels and all
// iterations of one match run in ONE persistent workgroup (the reference fans out to ParallelWorker
// threads once per iteration, :154); the nine sums are accumulated per lane in fp32, reduced across the
// workgroup in fp64 and the 3x3 system is solved on the device with the BCL's cofactor formulas.  A single match is 512
// lanes with the scan's points in LDS, one barrier and one sine / cosine per iteration (hs_hessian_block: round 5).
// Occupancy probabilities exp(v)/(exp(v)+1) (OccGridMap.GetCachedProbability, OccGridMap.cs:97-107) are READ from a dense
// per-level grid `prob` that every writer of the log-odds grid keeps current (K5 for the cells it touches, upload and
// reset for all of them) -- the device's form of the reference's per-cell cache, without its epochs: the value always is
// the current cell's probability, also across Reset, where the reference's cache can serve pre-reset values (deviation
// D5: DESIGN.md sec.3 and include/slamhip.h, slamhip_hs_probability).
// Float parity target: pose within 1e-4 m / 1e-4 rad (H6).
//
// K5 replaces OccGridMap.UpdateByScan and friends (HectorSLAM/Map/OccGridMap.cs:114-239) for every level of
// the pyramid (MapRepMultiMap.cs:73-77) in one launch.  The once-per-scan guards make a cell's new value depend only on
// (a) whether it is touched as free, (b) whether it is an end point, and (c) whether the first free touch
// precedes the first end-point touch in ray order.  No atomics and no per-cell scratch: lines are sorted
// by direction class and slope (raster.h, shared with the HoleMap update); a cell has ONE writer -- the wavefront of a
// cell near the begin cell, or beyond that the lane of the lowest line index among the lines that touch it (one lane
// per (line, step), closed-form Bresenham position) -- which finds the first "free" line and the first line that ends
// in the cell and replays the at most two state transitions literally: bit-exact fp32 cell values and update indices.
// The cells are stored as the reference stores them, LogOddsCell {UpdateIndex, Value} (LogOddsCell.cs:16-21): one 8-byte access.
#include "common.h"
#include "m3x2.h"
#include "raster.h"
#include <vector>
#include <atomic>
#include <chrono>
#include <stdlib.h>

#define HS_MAX_LEVELS 8
#define HS_NONE 0xFFFFFFFFu

struct hs_level {
    int w, h; float cell, stm;             // MapProperties: Dimensions, CellLength, ScaleToMap (MapProperties.cs:22-32)
    sh_m3x2 map_t_world, world_t_map;      // GridMap.cs:46-47
    slamhip_cell *d_cells;                 // mapArray (GridMap.cs:13) in the reference's own layout, LogOddsCell {UpdateIndex, Value} (LogOddsCell.cs:16-21): the grid
                                           // update reads and writes a cell with ONE 8-byte access (two arrays: 30.8 -> 26.9 us per update with the second one left out)
    float *d_prob;                         // GetCachedProbability of every cell (OccGridMap.cs:97-107), kept current by every writer of d_cells
    int curr_update_index;                 // OccGridMap.cs:20
    int iterations;                        // EstimateIterations (OccGridMap.cs:53)
};

struct hs_level_dev {                      // what the kernels need, by value
    int w, h; float cell, stm;
    sh_m3x2 map_t_world, world_t_map;
    const float *prob;                     // what the matcher's taps read: exp and divide happen when a cell changes, not per tap
    const slamhip_cell *cells;             // (HS_PROB_MODE 1 / 2, developer experiment: the taps read the cells and form the probabilities themselves)
    int iterations;
};
#ifndef HS_PROB_MODE
#define HS_PROB_MODE 0                     // 0: the probability grid, kept by every writer of the cells | 1: from the cells, exact expf and division per tap | 2: ... hardware exp and reciprocal
#endif

struct slamhip_hs {
    slamhip_ctx *ctx;
    int n_levels;
    hs_level lv[HS_MAX_LEVELS];
    float odds_free, odds_occ, lo_free, lo_occ;          // OccGridMap.cs:24-27
    int n_points, cap_points;
    float2 *d_pts; float origin[2];
    float2 *d_pts_base; int pts_buf; uint64_t launch_count, launch_done, pts_use[2], match_launch_no;   // two device blocks used in turn; which launches read which (see slamhip_cs_set_scan)
    float *h_pts; hipEvent_t ev_pts; bool pts_in_flight;   // pinned staging of the scan: one async copy (or upload launch), no wait in set_scan
    bool upload_pending; size_t upload_bytes;              // set_scan filled the staging block; the first launch that reads the points issues the upload (hs_flush_scan) -- a single match pulls the block itself
    uint32_t upload_seq;                                   // upload launches issued; the launch stores it behind the staged points (h_pts + 2 * cap) when it has read them
    float *d_io; float *h_io; int cap_io;                // hints in / poses out (floats)
    // K5 line tables, per level: lines by index, lines sorted by (direction class, slope bucket), bucket starts, header
    void *d_k5_byidx, *d_k5_cand; int *d_k5_start, *d_k5_hdr; int cap_lines;
    int *d_k5_sec; int k5_sec_parity; bool k5_toggle_pending;                        // [2][HS_MAX_LEVELS][K5_SEC] sector records of the cell kernel: an update reads the set the last one wrote
};

struct hs_levels_arg { hs_level_dev lv[HS_MAX_LEVELS]; int n; };

// ---- K4 device code ------------------------------------------------------------------------------------------
// OccGridMap.GetCachedProbability (:97-107)
__device__ static inline float hs_prob_v(float v)
{
    const float odds = expf(v);                                            // :101
    return odds / (odds + 1.0f);                                           // :102
}

__device__ static inline float hs_prob_tap(float v)
{
#if HS_PROB_MODE == 2
    const float odds = __expf(v);
    return __fdividef(odds, odds + 1.0f);
#else
    return hs_prob_v(v);
#endif
}

// one DPP step of a binary64 value (the wave partials' tree in hs_hessian_block; no LDS permutes: a ds_bpermute costs ~100
// cycles of latency).  Lanes a step's row mask excludes receive zero (update_dpp's `old`).
template <int CTRL, int ROWS> __device__ static inline double hs_dpp_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, ROWS, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWS, 0xf, false));
}

// one DPP step in binary32: the wave trees of the nine sums (butterfly inside each row of 16 lanes, then row_bcast:15 into rows
// 1 and 3 and row_bcast:31 into rows 2 and 3 -- the total is valid in lane 63).  (Nine binary64 wave sums per iteration -- two DPP
// moves and a double add per step, in dependent chains -- were half of the first matcher's run time; the reference itself sums
// these terms in binary32, sequentially per thread chunk, ScanMatcher.cs:166-180, so a binary32 tree over 64 lanes is at least
// as accurate as what it is compared with.  The wave partials are still added in binary64.)
template <int CTRL, int ROWS> __device__ static inline float hs_dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROWS, 0xf, false));
}

#ifdef K4_TIMES
// developer instrumentation (build with SLAMHIP_K4_TIMES=1): wall-clock ticks (100 MHz) per phase of an iteration,
// accumulated by thread 0 of workgroup 0
__device__ unsigned long long g_k4_times[16];
#define K4_STAMP(k) { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t_ = wall_clock64(); g_k4_times[k] += t_ - k4_last; k4_last = t_; } }
__device__ unsigned long long g_k4_last;
__device__ unsigned long long g_k4_pts[16];      // the points' phase per iteration of a match (coarse level first)
__device__ int g_k4_iter;
#define K4_STAMP_BEGIN unsigned long long k4_last = wall_clock64(); if (threadIdx.x == 0 && blockIdx.x == 0) { if (g_k4_last) g_k4_times[5] += k4_last - g_k4_last; }
#else
#define K4_STAMP(k) {}
#define K4_STAMP_BEGIN
#endif

// GetCompleteHessianDerivs (:135-204) for the whole workgroup; result (9 sums) in sums[], uniform in every wavefront.
// order: dTr.x, dTr.y, dTr.z, H11, H22, H33, H12, H13, H23
//
// Round 5: a single match is nine dependent iterations on ONE compute unit, and with 16 wavefronts (four per SIMD) it was
// bound by VALU issue, not by latency (profiles/r05_secondary_kernels.json: 475 VALU instructions per wavefront and iteration,
// 0.22 of a wavefront's cycles issuing VALU x 4 wavefronts per SIMD): every wavefront repeats the uniform part of an
// iteration -- Matrix3x2.CreateRotation's IEEERemainder, two binary64 sin/cos evaluations, the 3 x 3 inverse -- and three
// barriers.  Now
//  * FEW wavefronts with several points per lane (the single match runs 256 lanes x 5 points: one wavefront per SIMD, the
//    uniform part once per SIMD), the points of the scan in LDS, the taps of a lane's points requested together by a
//    branch-free interpolation (outside the grid: taps of cell 0, result selected to zero -- ScanMatcher.cs:216-219);
//  * ONE barrier per iteration: the wave partials go to one of two alternating LDS blocks, and after the barrier every
//    wavefront adds the partials itself -- lane k * NW + w loads partial w of sum k, a DPP row tree in binary64 (the same tree
//    and order as before), nine v_readlane;
//  * the rotation's sin/cos is the one the derivative needs (:145-146) whenever |angle| < pi (IEEERemainder returns its
//    argument there, exactly), so it is evaluated once.
// (Round 4, measured and rejected: a 4 x 4 window of probabilities per point kept in registers across a level's iterations, so
// that iterations 2 .. n read no memory -- 34.0 -> 37.1 us per match: the first iteration's 64 bytes per point in four unaligned
// 16-byte loads cost more than the later iterations' taps, which hit the L2 anyway.)
#define HS_LDS_PTS 2048                    // scan points kept in LDS (16 KB); longer scans are read from global memory

// Matrix3x2.CreateRotation (m3x2.h) given sin/cos of the SAME angle: valid for |radians| < pi, where IEEERemainder(radians,
// 2 pi) == radians
__device__ static inline sh_m3x2 hs_rotation_sc(float radians, float s_in, float c_in)
{
    const float pi = 3.14159274f;
    if (!(fabsf(radians) < pi)) return sh_m3x2_rotation(radians);
    const float epsilon = 0.001f * pi / 180.0f;
    float c = c_in, s = s_in;
    if (radians > -epsilon && radians < epsilon) { c = 1; s = 0; }
    else if (radians > pi / 2 - epsilon && radians < pi / 2 + epsilon) { c = 0; s = 1; }
    else if (radians < -pi + epsilon || radians > pi - epsilon) { c = -1; s = 0; }
    else if (radians > -pi / 2 - epsilon && radians < -pi / 2 + epsilon) { c = 0; s = -1; }
    sh_m3x2 r = { c, s, -s, c, 0.0f, 0.0f };
    return r;
}

template <int BDIM> struct hs_shape {
    static constexpr int NW = BDIM >> 6;                                   // wavefronts
    static constexpr int PU = BDIM >= 1024 ? 2 : BDIM >= 512 ? 3 : 5;      // points per lane and pass (1080 rays: one pass)
    static constexpr int RED = 9 * NW;                                     // doubles per reduction block
};

template <int BDIM, bool LDSP>
__device__ static __forceinline__ void hs_hessian_block(const hs_level_dev &L, const float2 *pts, int n, const float pose[3],
                                                        double *red /* [hs_shape::RED]: this iteration's block */, float sums[9])
{
    constexpr int NW = hs_shape<BDIM>::NW, PU = hs_shape<BDIM>::PU;
    K4_STAMP_BEGIN
    float s, c;
    sh_det_sincosf(pose[2], &s, &c);
    const sh_m3x2 t = sh_m3x2_mul(sh_m3x2_mul(hs_rotation_sc(pose[2], s, c),
                                              sh_m3x2_translation(pose[0] * L.cell, pose[1] * L.cell)),
                                  sh_m3x2_scale(L.stm));                   // :139-142
    const float sinRot = s * L.stm, cosRot = c * L.stm;                    // :145-146
    const float limx = (float)L.w - 2.0f, limy = (float)L.h - 2.0f;       // MapProperties.cs:42
    K4_STAMP(0)                                                            // transform + trigonometry
    float acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int base = 0; base < n; base += BDIM * PU) {
        float2 p[PU], r0[PU], r1[PU];
        float fx[PU], fy[PU];
        bool ok[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int i = base + (int)threadIdx.x + u * BDIM;
            p[u] = i < n ? pts[i] : make_float2(0.f, 0.f);
            float cx, cy;
            sh_v2_transform(p[u].x, p[u].y, t, &cx, &cy);                  // :161
            // InterpMapValueWithDerivatives (:211-249), MapProperties.cs:83-87
            ok[u] = i < n && !(!(cx == cx) || !(cy == cy) || cx < 0.0f || cx > limx || cy < 0.0f || cy > limy);
            const float fxx = floorf(cx), fyy = floorf(cy);                // :222
            const int ix = ok[u] ? (int)fxx : 0, iy = ok[u] ? (int)fyy : 0;
            fx[u] = cx - fxx; fy[u] = cy - fyy;                            // :225
            const int idx = iy * L.w + ix;                                 // :227
#if HS_PROB_MODE == 0
            __builtin_memcpy(&r0[u], L.prob + idx, sizeof(float2));        // (two adjacent taps: one 8-byte load)
            __builtin_memcpy(&r1[u], L.prob + idx + L.w, sizeof(float2));
#else
            int4 c0, c1;                                                   // (two adjacent cells {UpdateIndex, Value}: one 16-byte load)
            __builtin_memcpy(&c0, L.cells + idx, sizeof(int4));
            __builtin_memcpy(&c1, L.cells + idx + L.w, sizeof(int4));
            r0[u] = make_float2(hs_prob_tap(__int_as_float(c0.y)), hs_prob_tap(__int_as_float(c0.w)));
            r1[u] = make_float2(hs_prob_tap(__int_as_float(c1.y)), hs_prob_tap(__int_as_float(c1.w)));
#endif
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const float i0 = r0[u].x, i1 = r0[u].y, i2 = r1[u].x, i3 = r1[u].y;            // :230-233
            const float dx1 = i0 - i1, dx2 = i2 - i3, dy1 = i0 - i2, dy2 = i1 - i3;        // :235-239
            const float xi = 1.0f - fx[u], yi = 1.0f - fy[u];              // :241-242
            float P = ((i0 * xi + i1 * fx[u]) * yi) + ((i2 * xi + i3 * fx[u]) * fy[u]);   // :245-246
            float gx = -((dx1 * xi) + (dx2 * fx[u]));                      // :247
            float gy = -((dy1 * yi) + (dy2 * fy[u]));                      // :248
            if (!ok[u]) { P = 0.0f; gx = 0.0f; gy = 0.0f; }                // :216-219
            const float fun = 1.0f - P;                                    // :164
            const float rot = ((-sinRot * p[u].x - cosRot * p[u].y) * gx + (cosRot * p[u].x - sinRot * p[u].y) * gy);   // :169-170
            acc[0] += gx * fun;  acc[1] += gy * fun;  acc[2] += rot * fun; // :166,:167,:172
            acc[3] += gx * gx;   acc[4] += gy * gy;   acc[5] += rot * rot; // :174-176
            acc[6] += gx * gy;   acc[7] += gx * rot;  acc[8] += gy * rot;  // :178-180
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#ifdef K4_TIMES
    if (threadIdx.x == 0 && blockIdx.x == 0) { g_k4_pts[g_k4_iter & 15] += wall_clock64() - k4_last; g_k4_iter++; }
#endif
    K4_STAMP(1)                                                            // points: taps, interpolation, products
    // the nine trees step by step side by side (independent adds between the steps of one tree), then one store block
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] += hs_dpp_f32<0xB1, 0xf>(acc[k]);          // quad_perm [1,0,3,2]
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] += hs_dpp_f32<0x4E, 0xf>(acc[k]);          // quad_perm [2,3,0,1]
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] += hs_dpp_f32<0x124, 0xf>(acc[k]);         // row_ror:4
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] += hs_dpp_f32<0x128, 0xf>(acc[k]);         // row_ror:8
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] += hs_dpp_f32<0x142, 0xa>(acc[k]);         // row_bcast:15 -> rows 1, 3
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] += hs_dpp_f32<0x143, 0xc>(acc[k]);         // row_bcast:31 -> rows 2, 3
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 9; k++) red[k * NW + wid] = (double)acc[k];
    }
    K4_STAMP(2)                                                            // wave sums + store
    __syncthreads();
    K4_STAMP(3)                                                            // the barrier
    // every wavefront: nine sums over the NW wave partials; value v = k * NW + w sits in lane v & 63 of register v >> 6, so a
    // sum is one aligned group of NW lanes of a DPP row; after the tree the group's first lane holds it
    constexpr int NV = 9 * NW, NR = (NV + 63) >> 6;
    double d[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) d[r] = lane + 64 * r < NV ? red[lane + 64 * r] : 0.0;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        if (NW >= 2) d[r] += hs_dpp_f64<0xB1, 0xf>(d[r]);
        if (NW >= 4) d[r] += hs_dpp_f64<0x4E, 0xf>(d[r]);
        if (NW == 8) d[r] += hs_dpp_f64<0x141, 0xf>(d[r]);             // row_half_mirror: the other quad of the group of 8
        if (NW >= 16) d[r] += hs_dpp_f64<0x124, 0xf>(d[r]);
        if (NW >= 16) d[r] += hs_dpp_f64<0x128, 0xf>(d[r]);
    }
    float f[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) f[r] = (float)d[r];
#pragma unroll
    for (int k = 0; k < 9; k++)
        sums[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f[(k * NW) >> 6]), (k * NW) & 63));
    K4_STAMP(4)                                                            // totals
#ifdef K4_TIMES
    if (threadIdx.x == 0 && blockIdx.x == 0) g_k4_last = k4_last;          // (stamp 5: from here to the next iteration's start: the step, the level change)
#endif
}

// EstimateTransformationLogLh (:93-125) applied by every thread identically (uniform registers)
__device__ static inline void hs_step(const float sums[9], float est[3])
{
    const float H[9] = { sums[3], sums[6], sums[7],  sums[6], sums[4], sums[8],  sums[7], sums[8], sums[5] };  // :198-200
    if (H[0] != 0.0f && H[4] != 0.0f) {                                    // :97
        float R[9];
        if (!sh_invert_h(H, R)) return;                                    // :99-103
        const float d0 = sums[0], d1 = sums[1], d2 = sums[2];
        float sx = (d0 * R[0]) + (d1 * R[3]) + (d2 * R[6]) + 0.0f;         // :105 Vector3.Transform(dTr, iH)
        float sy = (d0 * R[1]) + (d1 * R[4]) + (d2 * R[7]) + 0.0f;
        float sz = (d0 * R[2]) + (d1 * R[5]) + (d2 * R[8]) + 0.0f;
        if (sz > 0.2f) sz = 0.2f;                                          // :107-111
        else if (sz < -0.2f) sz = -0.2f;                                   // :113-117
        est[0] += sx; est[1] += sy; est[2] += sz;                          // :119
    }
}

// MatchData(MapRepMultiMap) (:41-54): one workgroup per hint; levels coarse -> fine.
// only_level >= 0 restricts to one level with `iters_override` iterations (MatchData(OccGridMap), :64-84).
template <int BDIM>
__global__ void __launch_bounds__(BDIM)
k4_match(hs_levels_arg A, const float2 *__restrict__ pts, int n, const float *__restrict__ hints, float3 hint1,
         float *__restrict__ out, int only_level, int iters_override, uint32_t *mail, uint32_t mail_seq,
         const float2 *up_src, float2 *up_dst, uint32_t *up_flag, uint32_t up_seq, int n_helpers_from)
{
    __shared__ double red[2 * hs_shape<BDIM>::RED];
    __shared__ float2 pts_s[HS_LDS_PTS];
    const int b = blockIdx.x;
    if (n_helpers_from > 0 && b >= n_helpers_from) {
        // Round 6 -- the single match's HELPER workgroups.  In the per-scan flow the grid update has just rewritten the cached
        // probabilities from every XCD, and the first iteration on each level finds none of its taps in the L2 (the points' phase:
        // 14.0 us per match against 8.9 on a resting pyramid).  Requesting the finer levels' taps early from the matching
        // workgroup itself was measured a loss in round 5 (its own taps queue behind them).  Workgroups are dealt to the XCDs round
        // robin, so workgroup 8 of the launch shares its L2 with workgroup 0: it requests the lines of the FINER levels' taps at the
        // hint pose -- the match moves the pose by a cell or two, a line holds 32 -- while workgroup 0 iterates on the coarse level,
        // and leaves.  Workgroups 1 .. 7 (other XCDs) leave at once.  Nothing is written: a prefetch, never a result.
#ifndef K4_HELP_ALL
#define K4_HELP_ALL 0
#endif
        if ((!K4_HELP_ALL && (b & 7) != 0) || n <= 0 || only_level >= 0) return;
        // (The points are read from the DEVICE copy, never from the staging block: the host may refill -- or free and reallocate -- that
        // block as soon as workgroup 0 has read it, long before this workgroup runs; round 6's soak, seed 6105, a memory access fault.
        // On a freshly set scan the device copy still holds the previous scan, or a mixture while workgroup 0 stores the new one: end
        // points of consecutive scans fall on the same lines, and whatever floats are found there are range-tested like any point.)
        const float2 *src = pts;
        float acc = 0.f;
        for (int l = A.n - (K4_HELP_ALL == 2 ? 1 : 2); l >= 0; l--) {
            const hs_level_dev &L = A.lv[l];
            float est[3];
            sh_v2_transform(hint1.x, hint1.y, L.map_t_world, &est[0], &est[1]);
            est[2] = hint1.z;
            float s, c;
            sh_det_sincosf(est[2], &s, &c);
            const sh_m3x2 t = sh_m3x2_mul(sh_m3x2_mul(hs_rotation_sc(est[2], s, c), sh_m3x2_translation(est[0] * L.cell, est[1] * L.cell)), sh_m3x2_scale(L.stm));
            const float limx = (float)L.w - 2.0f, limy = (float)L.h - 2.0f;
            for (int i = threadIdx.x; i < n; i += BDIM) {
                const float2 p = src[i];
                float cx, cy;
                sh_v2_transform(p.x, p.y, t, &cx, &cy);
                const bool ok = !(!(cx == cx) || !(cy == cy) || cx < 0.0f || cx > limx || cy < 0.0f || cy > limy);
                const int idx = ok ? (int)floorf(cy) * L.w + (int)floorf(cx) : 0;
                acc += L.prob[idx] + L.prob[idx + L.w];
            }
        }
        asm volatile("" :: "v"(acc));                                       // (the loads are kept; their values are not)
        return;
    }
    float est_w[3] = { hint1.x, hint1.y, hint1.z };                         // :43 (a single hint travels in the launch arguments)
    if (hints) { est_w[0] = hints[3 * b]; est_w[1] = hints[3 * b + 1]; est_w[2] = hints[3 * b + 2]; }
#ifdef K4_TIMES
    if (threadIdx.x == 0 && blockIdx.x == 0) g_k4_iter = 0;
#endif
    const bool in_lds = n <= HS_LDS_PTS;
    if (up_src || in_lds) {
        // The scan's points into LDS, every lane's loads requested together (one memory round trip, not one per point).
        // up_src: a single match on a freshly set scan (one workgroup, n <= HS_LDS_PTS) reads them straight from the pinned
        // staging block -- this launch IS the scan upload -- and stores them to the device copy for the launches that follow
        // (grid update); the stores depend on the loads, so after the barrier the staging block has been read and the host may
        // refill it.
        constexpr int FU = HS_LDS_PTS / BDIM;
        const float2 *src = up_src ? up_src : pts;
        float2 v[FU];
#pragma unroll
        for (int u = 0; u < FU; u++) { const int i = threadIdx.x + u * BDIM; if (i < n) v[u] = src[i]; }
#pragma unroll
        for (int u = 0; u < FU; u++) {
            const int i = threadIdx.x + u * BDIM;
            if (i < n) { pts_s[i] = v[u]; if (up_src) up_dst[i] = v[u]; }
        }
        __syncthreads();
        if (up_src && threadIdx.x < SH_UPLOAD_PARTS) __hip_atomic_store(up_flag + threadIdx.x, up_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the words of sh_upload: common.h)
    }
    // (Round 5, measured and rejected: the finer levels' taps requested early -- one LDS-DMA word per point and row at the hint
    // pose, into a dump nobody reads, issued inside the first iteration so that their lines arrive while the coarse level
    // iterates.  In the per-scan flow the grid update has just rewritten the cached probabilities from every XCD and the points'
    // phase takes 14.0 us per match instead of 8.9 on a resting pyramid, but the requests cost more than the misses they avoid:
    // 23.6 -> 25.5 us per match stand-alone, 30.7 -> 34.0 us inside HectorSLAMProcessor.Update -- vector memory returns in
    // order, so the coarse level's own taps queue behind them, and twelve scattered 4-byte requests per lane are as much work
    // for the address unit as two iterations' taps.)
    if (n > 0) {                                                           // :66 (else: hint returned, :83)
        const int l_hi = only_level >= 0 ? only_level : A.n - 1;
        const int l_lo = only_level >= 0 ? only_level : 0;
        int par = 0;
        for (int l = l_hi; l >= l_lo; l--) {                               // :47
            const hs_level_dev &L = A.lv[l];
            float est[3];
            sh_v2_transform(est_w[0], est_w[1], L.map_t_world, &est[0], &est[1]);   // :68 GetMapCoordsPose
            est[2] = est_w[2];
            const int iters = only_level >= 0 ? iters_override : L.iterations;
            for (int it = 0; it < iters; it++) {                           // :70-73
                float sums[9];
                if (in_lds) hs_hessian_block<BDIM, true>(L, pts_s, n, est, red + par, sums);
                else hs_hessian_block<BDIM, false>(L, pts, n, est, red + par, sums);
                par ^= hs_shape<BDIM>::RED;                                // (a block is written again two barriers after it was read)
                hs_step(sums, est);
            }
            est[2] = sh_normalize_angle(est[2]);                           // :76
            sh_v2_transform(est[0], est[1], L.world_t_map, &est_w[0], &est_w[1]);   // :79 GetWorldCoordsPose
            est_w[2] = est[2];
        }
    }
    if (threadIdx.x == 0) {
        out[3 * b] = est_w[0]; out[3 * b + 1] = est_w[1]; out[3 * b + 2] = est_w[2];
        if (mail) {                                                        // a single blocking match: the pose and the completion word into the context's mailbox (common.h)
            ((float *)mail)[0] = est_w[0]; ((float *)mail)[1] = est_w[1]; ((float *)mail)[2] = est_w[2];
            __hip_atomic_store(mail + 15, mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void __launch_bounds__(256)
k4_hessian(hs_levels_arg A, int level, const float2 *__restrict__ pts, int n, const float *__restrict__ pose_in,
           float *__restrict__ out12)
{
    __shared__ double red[hs_shape<256>::RED];
    float pose[3] = { pose_in[0], pose_in[1], pose_in[2] };
    float sums[9];
    hs_hessian_block<256, false>(A.lv[level], pts, n, pose, red, sums);
    if (threadIdx.x == 0) {
        out12[0] = sums[3]; out12[1] = sums[6]; out12[2] = sums[7];
        out12[3] = sums[6]; out12[4] = sums[4]; out12[5] = sums[8];
        out12[6] = sums[7]; out12[7] = sums[8]; out12[8] = sums[5];
        out12[9] = sums[0]; out12[10] = sums[1]; out12[11] = sums[2];
    }
}

// ---- K5 device code --------------------------------------------------------------------------------------------
// One line of OccGridMap.UpdateByScan on one level: UpdateLineBresenhami (:155-190) + Bresenham2D (:220-239).
// The line has da "free" cells (steps i = 0..da-1, the end point excluded, :224-238) plus the occupied end cell; after
// i steps the walk has taken (e0 + i*db) / da minor steps, e0 = da / 2 (closed form of :228-235, db <= da;
// tests/test_closed_forms.py).  The update is CELL-centric (raster.h): a cell asks which lines draw it.  Lines are
// processed in index order by the reference, and a cell changes at most twice per update (BresenhamCellFree marks it,
// BresenhamCellOcc overrides the mark), so all a cell needs is the smallest index of a line that crosses it as "free",
// the smallest index of a line that ends in it, and their order -- no atomics, no per-cell scratch, coalesced rows.
struct k5_level { int w, h; sh_m3x2 t; slamhip_cell *cells; float *prob; int mark_free, mark_occ; int wg0, wgn; };
// a cell as one 8-byte word: update_index in the low half, the value's bits in the high half (slamhip_cell, include/slamhip.h)
__device__ static __forceinline__ void k5_load_cell(const slamhip_cell *c, float &v, int &u) { const int2 w = *(const int2 *)c; u = w.x; v = __int_as_float(w.y); }
__device__ static __forceinline__ void k5_store_cell(slamhip_cell *c, float v, int u) { *(int2 *)c = make_int2(u, __float_as_int(v)); }
struct k5_arg { k5_level lv[HS_MAX_LEVELS]; int n; };
// The update gated on the device (HectorSLAMProcessor's per-scan flow, slamhip_hsproc_update): the launch is enqueued right
// behind the match, before the host has the pose -- the kernel reads the matched pose the match left in device memory, applies
// the processor's own test (HectorSLAMProcessor.cs:107-109: moved more than min_dist or turned more than min_angle since the
// last update) with the very float operations the host applies to the pose it receives, and either returns at once or forms
// the level transforms (OccGridMap.cs:120-123) itself.  Without it the update waited for host round trip + launch: 11.5 us of
// idle device between the two kernels of a scan.
struct k5_gate { const float *d_pose; float last[3]; float min_dist, min_angle; float stm[HS_MAX_LEVELS]; int on; };
__host__ __device__ static inline float hs_deg_diff(float a, float b)      // MathEx.DegDiff (BaseSLAM/MathEx.cs:69-73)
{
    float d = ((a - b) + 180.0f) / 360.0f;
    return ((d - floorf(d)) * 360.0f) - 180.0f;
}
__host__ __device__ static inline bool hs_moved_enough(const float pose[3], const float last[3], float min_dist, float min_angle)
{
    const float ddx = pose[0] - last[0], ddy = pose[1] - last[1];
    const float dist2 = ddx * ddx + ddy * ddy;                            // Vector2.DistanceSquared :107
    return dist2 > min_dist * min_dist || hs_deg_diff(pose[2], last[2]) > min_angle;   // :108 (radians through DegDiff, as the reference does)
}
struct k5_line { int da, sdb, ray, flags; };      // major length, signed minor length, line index, valid | major_x << 1 | (smaj + 1) << 2
#define K5_ZONE 16                     // Chebyshev radius around the begin cell handled one wavefront per cell
#define K5_LDS_LINES 3072
#define K5_HDR 8                       // ints per level: [0] begin x, [1] begin y, [2] longest line, [3] valid lines, [4] first valid line

// does the line draw cell (major offset a >= 1, signed minor offset b)?  1: as a free cell, 2: as its end cell, 0: no
__device__ static inline int k5_hit(const k5_line c, int a, int b)
{
    if (a > c.da) return 0;
    const int B = b < 0 ? -b : b, db = c.sdb < 0 ? -c.sdb : c.sdb;
    if (B > 0 && (c.sdb == 0 || (b > 0) != (c.sdb > 0))) return 0;
    if (a == c.da) return B == db ? 2 : 0;                                 // the end cell (:187), excluded from the free steps
    const int e = c.da / 2 + a * db;                                       // minor steps = e / da (maps <= 32768 a side: < 2^31)
    return (e >= B * c.da && e < (B + 1) * c.da) ? 1 : 0;
}

// per level (blockIdx.x): the lines of the scan, counting-sorted by (direction class, slope bucket)
__global__ void __launch_bounds__(1024)
k5_prepare(k5_arg A, const float2 *__restrict__ pts, int n, float ox, float oy, int cap, k5_line *__restrict__ byidx_all,
           k5_line *__restrict__ cand_all, int *__restrict__ start_all, int *__restrict__ hdr_all)
{
    __shared__ int hist[4 * RS_NBUCK];
    __shared__ int wsum[16];
    __shared__ int s_R, s_nv, s_first;
    const k5_level &L = A.lv[blockIdx.x];
    k5_line *byidx = byidx_all + (size_t)blockIdx.x * cap, *cand = cand_all + (size_t)blockIdx.x * cap;
    int *start = start_all + (size_t)blockIdx.x * (4 * RS_NBUCK + 1), *hdr = hdr_all + blockIdx.x * K5_HDR;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    for (int i = t; i < 4 * RS_NBUCK; i += 1024) hist[i] = 0;
    if (t == 0) { s_R = 0; s_nv = 0; s_first = 0x7fffffff; }
    __syncthreads();
    float bxf, byf;
    sh_v2_transform(ox, oy, L.t, &bxf, &byf);                              // :126
    const int bx = sh_f2i(rintf(bxf)), by = sh_f2i(rintf(byf));            // :127 ToRoundPoint (banker's, VectorEx.cs:183-186)
    int my_R = 0, my_nv = 0, my_first = 0x7fffffff;
    k5_line keep[2];                                                       // a thread's first two lines stay in registers for the second pass
    keep[0].flags = 0; keep[1].flags = 0;
    for (int i = t, it = 0; i < n; i += 1024, it++) {
        float exf, eyf;
        sh_v2_transform(pts[i].x, pts[i].y, L.t, &exf, &eyf);              // :133
        const int ex = sh_f2i(rintf(exf)), ey = sh_f2i(rintf(eyf));        // :134
        const bool same = (bx == ex) & (by == ey);                         // :137
        const bool inside = (bx >= 0) & (by >= 0) & (bx < L.w) & (by < L.h) & (ex >= 0) & (ey >= 0) & (ex < L.w) & (ey < L.h);   // :158-161
        k5_line e; e.da = 0; e.sdb = 0; e.ray = i; e.flags = 0;
        if (!same && inside) {
            const int dx = ex - bx, dy = ey - by;
            const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
            const bool major_x = adx >= ady;                               // :175
            e.da = major_x ? adx : ady;
            e.sdb = major_x ? dy : dx;                                     // minor extent with its sign (:169-170)
            const int smaj = sh_sign(major_x ? dx : dy);
            e.flags = 1 | (major_x ? 2 : 0) | ((smaj + 1) << 2);
            atomicAdd(&hist[rs_class(major_x, smaj) * RS_NBUCK + rs_bucket((float)e.sdb / (float)e.da)], 1);
            my_R = max(my_R, e.da);
            my_nv++;
            my_first = min(my_first, i);
        }
        byidx[i] = e;
        if (it == 0) keep[0] = e; else if (it == 1) keep[1] = e;
    }
    for (int off = 32; off > 0; off >>= 1) {
        my_R = max(my_R, __shfl_down(my_R, off, 64)); my_nv += __shfl_down(my_nv, off, 64); my_first = min(my_first, __shfl_down(my_first, off, 64));
    }
    if (lane == 0) { atomicMax(&s_R, my_R); atomicAdd(&s_nv, my_nv); atomicMin(&s_first, my_first); }
    __syncthreads();
    {   // exclusive prefix over the 4096 bins: 4 consecutive bins per thread
        int v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = hist[4 * t + k]; sum += v[k]; }
        int incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        int base = incl - sum;
        for (int w = 0; w < wid; w++) base += wsum[w];
#pragma unroll
        for (int k = 0; k < 4; k++) { start[4 * t + k] = base; hist[4 * t + k] = base; base += v[k]; }
        if (t == 1023) start[4 * RS_NBUCK] = base;
    }
    __syncthreads();
    for (int i = t, it = 0; i < n; i += 1024, it++) {
        const k5_line e = it == 0 ? keep[0] : it == 1 ? keep[1] : byidx[i];      // (its own store: no other thread wrote byidx[i])
        if (e.flags & 1) {
            const int smaj = ((e.flags >> 2) & 3) - 1;
            const int pos = atomicAdd(&hist[rs_class((e.flags & 2) != 0, smaj) * RS_NBUCK + rs_bucket((float)e.sdb / (float)e.da)], 1);
            cand[pos] = e;
        }
    }
    if (t == 0) { hdr[0] = bx; hdr[1] = by; hdr[2] = s_R; hdr[3] = s_nv; hdr[4] = s_first; }
}

// the state transitions of one cell: BresenhamCellFree (:192-199) by the first line that crosses it, then
// BresenhamCellOcc (:201-218) by the first line that ends in it; a cell first touched by an end point is not
// marked free any more (the mark_occ update index is above mark_free)
__device__ static inline void k5_transition(const k5_level &L, float &v, int &u, int first_free, int first_occ, float lo_free, float lo_occ)
{
    if (first_free < first_occ && u < L.mark_free) { v += lo_free; u = L.mark_free; }     // :192-199
    if (first_occ != 0x7fffffff && u < L.mark_occ) {                       // :201-218
        if (u == L.mark_free) v -= lo_free;                                // :206-209
        if (v < 50.0f) v += lo_occ;                                        // :211-214
        u = L.mark_occ;                                                    // :216
    }
}

// wave-wide minimum by DPP (butterfly in rows of 16, row_bcast:15 / :31): valid in lane 63
template <int CTRL, int ROWS> __device__ static inline int k5_dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false); }
__device__ static inline int k5_wave_min(int x)
{
    x = min(x, (k5_dpp<0xB1, 0xf>(x))); x = min(x, (k5_dpp<0x4E, 0xf>(x)));
    x = min(x, (k5_dpp<0x124, 0xf>(x))); x = min(x, (k5_dpp<0x128, 0xf>(x)));
    x = min(x, (k5_dpp<0x142, 0xa>(x))); x = min(x, (k5_dpp<0x143, 0xc>(x)));
    return x;
}

// all levels in ONE launch.  BUILD (scans of up to K5_LDS_LINES points): every workgroup makes the line tables of ITS level
// itself, in LDS -- the transform of :133-134 per point, the counting sort by (direction class, slope bucket) -- instead of
// reading what a k5_prepare launch left in memory (5 us plus a launch boundary for a microsecond of arithmetic).  The order of
// the lines inside a bucket then differs from workgroup to workgroup (LDS atomics), so the work that is shared out between
// workgroups goes by LINE INDEX (byidx), never by table position.
#define K5_LDS_FIXED ((4 * RS_NBUCK + 4) * 4)
#ifndef K5_EXP
#define K5_EXP 0
#endif
#ifdef K5_TIMES
// developer instrumentation (build with SLAMHIP_K5_TIMES=1): 100 MHz wall-clock stamps per workgroup: start, tables, zone, end
__device__ unsigned long long g_k5_times[1024 * 4];
#define K5_STAMP(k) { if (threadIdx.x == 0 && blockIdx.x < 1024) g_k5_times[blockIdx.x * 4 + (k)] = wall_clock64(); }
__device__ unsigned long long g_k5_sub[1024 * 8];    // table phase, thread 0: behind the 1st barrier, the lines, the bins' prefix (2 stamps), the scatter
#define K5_SUB(k) { if (threadIdx.x == 0 && blockIdx.x < 1024) g_k5_sub[blockIdx.x * 8 + (k)] = wall_clock64(); }
#else
#define K5_STAMP(k) {}
#define K5_SUB(k) {}
#endif
#define K5_SEC 16                      // ints per level of the sector record: [0] the scan's line count, [1..9] the bounds of the eight sectors
// The sector bounds for the NEXT update, by the level's first workgroup when it has drawn its last cell (its tables still stand
// in LDS): a line's weight is its blocks of 64 steps beyond the zone (x 8) plus the fetch every line costs; an inclusive scan
// of the weights by line index (DPP wave scans, one wavefront for the wave totals); sector k starts behind the line in which the
// running weight passes k/8 of the total.  All 1024 threads of the workgroup call it.
template <int RPT>
__device__ static inline void k5_sector_bounds(const k5_line *__restrict__ byidx_s, int n_pts, int *s_wtot, int *s_wsum_all, int *s_bound,
                                               int *__restrict__ rec_out)
{
    const int t = threadIdx.x, lane_ = t & 63, wid = t >> 6;
    int wgt[RPT], wincl[RPT];
    if (t < 9) s_bound[t] = t == 0 ? 0 : n_pts;
#pragma unroll
    for (int it = 0; it < RPT; it++) {
        const int i = t + it * 1024;
        int w = 0;
        if (i < n_pts) {
            const k5_line ee = byidx_s[i];
            const int bl = ((ee.flags & 1) && ee.da >= K5_ZONE) ? (ee.da - K5_ZONE) / 64 + 1 : 0;
            w = 8 * bl + 2;
        }
        int incl = w;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);   // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);   // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);   // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);   // row_shr:8
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
        wgt[it] = w; wincl[it] = incl;
        if (lane_ == 63) s_wtot[it * 16 + wid] = incl;
    }
    __syncthreads();
    if (wid == 0) {        // the (at most 48) wave totals into their exclusive prefix; the total behind them
        constexpr int NT = RPT * 16;
        static_assert(NT <= 63, "the wave totals and their sum fit one wavefront");
        const int v = lane_ < NT ? s_wtot[lane_] : 0;
        int incl = v;
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);
        if (lane_ < NT) s_wtot[lane_] = incl - v;
        if (lane_ == 63) *s_wsum_all = incl;
    }
    __syncthreads();
    const int run = *s_wsum_all;
#pragma unroll
    for (int it = 0; it < RPT; it++) {
        const int i = t + it * 1024;
        if (i < n_pts) {
            const int incl = s_wtot[it * 16 + wid] + wincl[it], excl = incl - wgt[it];
#pragma unroll
            for (int k = 1; k < 8; k++) {
                const int ck = (k * run + 7) >> 3;
                if (excl < ck && ck <= incl) s_bound[k] = i + 1;           // (exactly one line per threshold: the weights are positive)
            }
        }
    }
    __syncthreads();
    if (t == 0) rec_out[0] = n_pts;
    if (t < 9) rec_out[1 + t] = s_bound[t];
}
static inline size_t k5_lds_bytes(bool build, int n) { return (size_t)K5_LDS_FIXED + (build ? (size_t)4 * RS_NBUCK * 4 + (size_t)32 * (size_t)((n + 3) & ~3) : 0); }
template <bool BUILD>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
k5_cells(k5_arg A, int cap, const float2 *__restrict__ pts, int n_pts, float ox, float oy,
         const k5_line *__restrict__ byidx_all, const k5_line *__restrict__ cand_all,
         const int *__restrict__ start_all, const int *__restrict__ hdr_all, float lo_free, float lo_occ,
         const int *__restrict__ sec_in, int *__restrict__ sec_out, const k5_gate gate)
{
    extern __shared__ __attribute__((aligned(16))) char k5_smem[];
    int *start = (int *)k5_smem;
    int *pos_s = (int *)(k5_smem + K5_LDS_FIXED);
    const int n4 = (n_pts + 3) & ~3;
    k5_line *cand_s = (k5_line *)(pos_s + (BUILD ? 4 * RS_NBUCK : 0));
    k5_line *byidx_s = cand_s + (BUILD ? n4 : 0);
    __shared__ __attribute__((aligned(16))) int wsum[16];
    __shared__ int s_R, s_nv, s_first;
    __shared__ int s_bound[9], s_rec[10], s_wsum_all, s_wtot[((K5_LDS_LINES + 1023) / 1024) * 16];   // BUILD: the sectors of phase 2 (below)
    // workgroups are shared out over the levels (host: wg0, wgn)
    int lvl = 0;
    for (int l = 1; l < A.n; l++) if ((int)blockIdx.x >= A.lv[l].wg0) lvl = l;
    const k5_level &L = A.lv[lvl];
    int bx, by, R, nv, first_line;
    K5_STAMP(0)
    sh_m3x2 T = L.t;
    if (BUILD && gate.on) {                                                 // (uniform: scalar loads, every wavefront the same answer)
        const float pose[3] = { gate.d_pose[0], gate.d_pose[1], gate.d_pose[2] };
        if (!hs_moved_enough(pose, gate.last, gate.min_dist, gate.min_angle)) return;
        T = sh_m3x2_mul(sh_m3x2_mul(sh_m3x2_rotation(pose[2]), sh_m3x2_translation(pose[0], pose[1])), sh_m3x2_scale(gate.stm[lvl]));   // OccGridMap.cs:120-123
    }
    if (BUILD) {
        const int t = threadIdx.x, lane_ = t & 63, wid = t >> 6;
        constexpr int RPT = (K5_LDS_LINES + 1023) / 1024;
        float2 p_next = make_float2(0.f, 0.f);
        if (t < n_pts) p_next = pts[t];
        for (int i = t; i < 4 * RS_NBUCK; i += 1024) start[i] = 0;          // (the histogram, then the bucket table)
        if (t == 0) { s_R = 0; s_nv = 0; s_first = 0x7fffffff; }
        int rec_v = -1;                                                     // (the sectors the level's first workgroup left last time: below;
        if (t < 10 && sec_in) rec_v = sec_in[lvl * K5_SEC + t];            //  requested here, stored behind the lines loop: no wait of its own)
        __syncthreads();
        K5_SUB(0)
        float bxf, byf;
        sh_v2_transform(ox, oy, T, &bxf, &byf);                            // :126
        bx = sh_f2i(rintf(bxf)); by = sh_f2i(rintf(byf));                  // :127 ToRoundPoint (banker's, VectorEx.cs:183-186)
        int bkt[RPT];
#pragma unroll
        for (int k = 0; k < RPT; k++) bkt[k] = -1;
        int my_R = 0, my_nv = 0, my_first = 0x7fffffff;
#pragma unroll 1
        for (int it = 0; it * 1024 < n_pts; it++) {
            const int i = t + it * 1024;
            int bb = -1;
            const float2 p = p_next;
            if (i + 1024 < n_pts) p_next = pts[i + 1024];
            if (i < n_pts) {
                float exf, eyf;
                sh_v2_transform(p.x, p.y, T, &exf, &eyf);                  // :133
                const int ex = sh_f2i(rintf(exf)), ey = sh_f2i(rintf(eyf));    // :134
                const bool same = (bx == ex) & (by == ey);                 // :137
                const bool inside = (bx >= 0) & (by >= 0) & (bx < L.w) & (by < L.h) & (ex >= 0) & (ey >= 0) & (ex < L.w) & (ey < L.h);   // :158-161
                k5_line e; e.da = 0; e.sdb = 0; e.ray = i; e.flags = 0;
                if (!same && inside) {
                    const int dx = ex - bx, dy = ey - by;
                    const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
                    const bool major_x = adx >= ady;                       // :175
                    e.da = major_x ? adx : ady;
                    e.sdb = major_x ? dy : dx;                             // minor extent with its sign (:169-170)
                    const int smaj = sh_sign(major_x ? dx : dy);
                    e.flags = 1 | (major_x ? 2 : 0) | ((smaj + 1) << 2);
                    bb = rs_class(major_x, smaj) * RS_NBUCK + rs_bucket((float)e.sdb / (float)e.da);
                    atomicAdd(&start[bb], 1);
                    my_R = max(my_R, e.da);
                    my_nv++;
                    my_first = min(my_first, i);
                }
                byidx_s[i] = e;
            }
#pragma unroll
            for (int k = 0; k < RPT; k++) if (k == it) bkt[k] = bb;
        }
        // (wave reductions by DPP, common.h: eighteen shuffles -- ds_bpermute, ~100 cycles each, on an LDS pipe that 32 wavefronts
        // of the compute unit use at once in this phase -- were a microsecond of it)
        my_R = sh_wave_max_to_lane63(my_R); my_nv = sh_wave_scan_incl(my_nv); my_first = sh_wave_min_all(my_first);
        if (lane_ == 63) { atomicMax(&s_R, my_R); atomicAdd(&s_nv, my_nv); atomicMin(&s_first, my_first); }
        if (t < 10) s_rec[t] = rec_v;
        __syncthreads();
        K5_SUB(1)
        // Phase 2's sectors: the lines go to the XCDs in eight ranges of consecutive indices (locality: see phase 2) that hold EQUAL
        // WORK, not equal counts -- with equal counts the sectors of the benchmark scan took 3.9 .. 13.6 us on level 0 (the long
        // corridor against the near wall; SLAMHIP_K5_TIMES) and the launch waited for the slowest.  The bounds are those the level's
        // first workgroup worked out during the LAST update (k5_sector_bounds at the end of this kernel; consecutive scans look
        // alike, and any partition is correct -- only the balance depends on it): making them here, in every workgroup, cost the
        // table phase 2 us (eight wavefronts per SIMD run that phase at once: an instruction more in it is 15 ns more).
        {   // exclusive prefix over the 4096 bins: 4 consecutive bins per thread
            int v[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = start[4 * t + k]; sum += v[k]; }
            const int incl = sh_wave_scan_incl(sum);
            if (lane_ == 63) wsum[wid] = incl;
            __syncthreads();                                               // (every thread has read its bins)
        K5_SUB(2)
            int base = incl - sum;
            {   // the wave totals in front of this one: four 16-byte reads, not up to fifteen dependent ones
                const int4 *w4 = (const int4 *)wsum;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int4 x = w4[q];
                    base += (4 * q + 0 < wid ? x.x : 0) + (4 * q + 1 < wid ? x.y : 0) + (4 * q + 2 < wid ? x.z : 0) + (4 * q + 3 < wid ? x.w : 0);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { start[4 * t + k] = base; pos_s[4 * t + k] = base; base += v[k]; }
            if (t == 1023) start[4 * RS_NBUCK] = base;
        }
        __syncthreads();
        K5_SUB(3)
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            const int i = t + it * 1024;
            if (i < n_pts && bkt[it] >= 0) cand_s[atomicAdd(&pos_s[bkt[it]], 1)] = byidx_s[i];      // (this thread's own store)
        }
        R = s_R; nv = s_nv; first_line = s_first;
        __syncthreads();
        K5_SUB(4)
        if (nv == 0) {
            if (sec_out && (int)blockIdx.x == L.wg0 && t == 0) sec_out[lvl * K5_SEC] = -1;      // (no record for the next update)
            return;
        }
        // From which step on is a line ALONE on its cells (round 5; K2's finding, holemap.hip)?  Step a of a line lies at minor offset
        // floor(a * slope + h), h = (da / 2) / da in [1/2 - 1/(2 da), 1/2] (:228-235), so two lines of a class -- signed slopes: a
        // cell of minor offset 0 is shared across the sign -- meet at major offset a only if a * |slope difference| < 1 + 1/(2 da):
        // beyond the zone (da >= 16) never from a = 1.0625 / g + 2 on, g the smallest slope difference to any other line of the class.
        // A thread per line looks at the twelve buckets either side of the line's own (no line in sight: g >= 10 bucket widths) and leaves
        // the step in bits 5 .. 17 of the line's flags; from there on the line's step lanes of phase 2 look nothing up (a cell on
        // the diagonal, which the quadrant's other class touches too, excepted).  No barrier: a lane that reads the word before it
        // is written finds zero = "not known" and takes the lookup -- slower, never wrong.
        // Only the lines of this workgroup's own sector (phase 2 below) are asked about.
        {
            const int wg_l_ = (int)blockIdx.x - L.wg0, xcd_ = wg_l_ & 7;
            const bool rec_ok_ = s_rec[0] == n_pts;
            const int c0_ = xcd_ == 0 ? 0 : rec_ok_ ? s_rec[1 + xcd_] : (int)(((long long)n_pts * xcd_) >> 3);
            const int c1_ = xcd_ == 7 ? n_pts : rec_ok_ ? s_rec[2 + xcd_] : (int)(((long long)n_pts * (xcd_ + 1)) >> 3);
            for (int i = c0_ + t; i < c1_; i += 1024) {
                const k5_line e = byidx_s[i];
                if (!(e.flags & 1) || e.da < K5_ZONE) continue;
                const float sl = (float)e.sdb * __builtin_amdgcn_rcpf((float)e.da);
                const int smaj = ((e.flags >> 2) & 3) - 1;
                const int cb = rs_class((e.flags & 2) != 0, smaj) * RS_NBUCK, bk = cb + rs_bucket(sl);
                const int w0 = start[max(bk - 12, cb)], w1 = start[min(bk + 12, cb + RS_NBUCK - 1) + 1];    // (the table's buckets come from the exact quotient: one bucket of slack)
                float g = 10.0f * (2.0f / (float)RS_NBUCK);
                int same = 0;                                              // (lines with this very slope: its own, and any other -> never alone)
                for (int ci = w0; ci < w1; ci++) {
                    const k5_line c = cand_s[ci];
                    const float d = fabsf((float)c.sdb * __builtin_amdgcn_rcpf((float)c.da) - sl);
                    same += d == 0.0f ? 1 : 0;
                    g = d > 0.0f && d < g ? d : g;
                }
                // (slopes by the hardware reciprocal: each within 2.5e-7 of the quotient; equal quotients that come out an ulp apart
                // make g tiny, i.e. "never alone")
                const int xa = same == 1 && g > 4.0e-6f ? min((int)(1.0625f * __builtin_amdgcn_rcpf(g - 1.0e-6f) * 1.0001f) + 2, 8191) : 8191;
                byidx_s[i].flags = e.flags | (xa << 5);
            }
        }
    } else {
        const int *start_g = start_all + (size_t)lvl * (4 * RS_NBUCK + 1), *hdr = hdr_all + lvl * K5_HDR;
        bx = hdr[0]; by = hdr[1]; R = hdr[2]; nv = hdr[3]; first_line = hdr[4];
        if (nv == 0) return;
        for (int i = threadIdx.x; i <= 4 * RS_NBUCK; i += 1024) start[i] = start_g[i];
        __syncthreads();
    }
    K5_STAMP(1)
#if K5_EXP == 1                          // developer experiment (wrong results): the launch with its table phase alone
    return;
#endif
    const k5_line *cand = BUILD ? cand_s : cand_all + (size_t)lvl * cap;
    const k5_line *byidx = BUILD ? byidx_s : byidx_all + (size_t)lvl * cap;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int gw = ((int)blockIdx.x - L.wg0) * 16 + wv, nw = L.wgn * 16;
    // (1) the zone around the begin cell, where a cell has many candidate lines: one wavefront per cell, one candidate
    //     per lane and trip, the smallest indices by wave reduction.  The begin cell itself is step 0 of every line.
    const int Z = K5_ZONE - 1 < R ? K5_ZONE - 1 : R;
    const int side = 2 * Z + 1;
    for (int item = gw; item < side * side; item += nw) {
        const int X = bx - Z + item % side, Y = by - Z + item / side;
        if (X < 0 || X >= L.w || Y < 0 || Y >= L.h) continue;              // wave-uniform
        const int dx = X - bx, dy = Y - by;
        const int cell = Y * L.w + X;
        float v; int u;
        k5_load_cell(L.cells + cell, v, u);                                // (requested now, needed after the search)
        int first_free = 0x7fffffff, first_occ = 0x7fffffff;
        if (dx == 0 && dy == 0) first_free = first_line;
        else {
            int cls[2], a[2], b[2];
            const int ncls = rs_classes(dx, dy, cls, a, b);
            for (int k = 0; k < ncls; k++) {
                int lo, hi;
                rs_range(start, cls[k], a[k], b[k], 0.5f, lo, hi);
                for (int ci = lo + lane; ci < hi; ci += 64) {
                    const k5_line c = cand[ci];
                    const int h = k5_hit(c, a[k], b[k]);
                    if (h == 1) first_free = min(first_free, c.ray);
                    else if (h == 2) first_occ = min(first_occ, c.ray);
                }
            }
            first_free = k5_wave_min(first_free);                          // valid in lane 63
            first_occ = k5_wave_min(first_occ);
        }
        if (lane == 63 && (first_free != 0x7fffffff || first_occ != 0x7fffffff)) {
            k5_transition(L, v, u, first_free, first_occ, lo_free, lo_occ);
            k5_store_cell(L.cells + cell, v, u);
#if HS_PROB_MODE == 0
            L.prob[cell] = hs_prob_v(v);
#endif
        }
    }
    // (2) beyond the zone: one lane per (line, step) -- work proportional to the cells the scan touches, not to the scan's
    //     bounding square (rounds 1-2 visited every cell of the square: ~5 M lanes for ~1.1 M touched cells over three levels of
    //     a 2048^2 pyramid).  Step i of a line lies at major offset i (its Chebyshev distance from the begin cell) and minor
    //     offset floor((da / 2 + i * db) / da) (:220-239; k5_hit is the same closed form), the end cell at i = da.  The lane
    //     asks, like a cell-centric lane would, which lines touch its cell -- one contiguous range of the slope-sorted table;
    //     nearly always the range holds the lane's own line and nothing else, and the cell is updated at once.  Otherwise the
    //     candidates are tested and the lane of the LOWEST line index among the touching lines owns the cell (every touching
    //     line has a lane on it, and all of them see the same candidates): it applies the transitions, the others drop it.
    //     Lines are dealt by index (a scan's points come in order of their angle), to the XCDs by sector: a line's cells share
    //     their 128-byte rows with its neighbours'.
    K5_STAMP(2)
#if K5_EXP == 2                          // developer experiment (wrong results): tables and the zone, no lines beyond it
    return;
#endif
    if (R < K5_ZONE) {
        if (BUILD && sec_out && (int)blockIdx.x == L.wg0 && threadIdx.x == 0) sec_out[lvl * K5_SEC] = -1;
        return;
    }
    const int wg_l = (int)blockIdx.x - L.wg0;                              // workgroup within the level
    const int xcd = wg_l & 7, wgs_x = (L.wgn - xcd + 7) >> 3, wg_x = wg_l >> 3;
    // (BUILD: the sectors hold equal work, and a sector's blocks end with ITS longest line -- see the tables above)
    const int nblk = (R - K5_ZONE) / 64 + 1;
    const bool rec_ok = BUILD && s_rec[0] == n_pts;                        // (a record of a scan with as many lines: its bounds are a partition of this one's)
    const int c0 = xcd == 0 ? 0 : rec_ok ? s_rec[1 + xcd] : (int)(((long long)n_pts * xcd) >> 3);
    const int n_sec = (xcd == 7 ? n_pts : rec_ok ? s_rec[2 + xcd] : (int)(((long long)n_pts * (xcd + 1)) >> 3)) - c0;
    const int items = nblk * n_sec;
    // (software pipeline: a cell's value and update index are requested when its item is fetched, two iterations before its
    // turn -- the cells and probabilities of a 2048^2 level are 48 MB, a microsecond or two away; two ahead against one: 32.6 -> 32.2 us,
    // and the kernel's 64 VGPRs leave no room for a third)
    struct k5_item { int cell, dx, dy, ray, end, xalone; float v; int u; };
#define K5_FETCH(it, item_)                                                                         \
    {                                                                                               \
        (it).cell = -1;                                                                             \
        if ((item_) < items) {                                                                      \
            const int blk_ = (item_) / n_sec, ci0_ = c0 + ((item_) - blk_ * n_sec);                 \
            const k5_line me_ = byidx[ci0_];               /* (uniform: a broadcast) */              \
            const int i_ = K5_ZONE + blk_ * 64 + lane;                                              \
            if ((me_.flags & 1) && i_ <= me_.da) {                                                                     \
                const int db_ = me_.sdb < 0 ? -me_.sdb : me_.sdb;                                   \
                const int e_ = me_.da / 2 + i_ * db_;      /* (maps <= 32768 a side: < 2^31) */      \
                int m_;                                                                             \
                if (L.w <= 2048 && L.h <= 2048) {          /* e < 2^24: the float estimate of e / da is within one; settled exactly */ \
                    m_ = (int)((float)e_ * __builtin_amdgcn_rcpf((float)me_.da));                   \
                    const int r_ = e_ - m_ * me_.da;                                                \
                    if (r_ < 0) m_--; else if (r_ >= me_.da) m_++;                                  \
                } else m_ = e_ / me_.da;                                                            \
                const int smaj_ = ((me_.flags >> 2) & 3) - 1;                                       \
                const int am_ = smaj_ < 0 ? -i_ : i_, bm_ = me_.sdb < 0 ? -m_ : m_;                 \
                (it).dx = (me_.flags & 2) ? am_ : bm_; (it).dy = (me_.flags & 2) ? bm_ : am_;       \
                (it).ray = me_.ray; (it).end = i_ == me_.da;                                        \
                { const int xa_ = (me_.flags >> 5) & 8191; (it).xalone = (xa_ == 0 || xa_ == 8191) ? 0x7fffffff : xa_; } \
                (it).cell = (by + (it).dy) * L.w + (bx + (it).dx);                                  \
                k5_load_cell(L.cells + (it).cell, (it).v, (it).u);                                  \
            }                                                                                       \
        }                                                                                           \
    }
    k5_item cur, nxt, nx2;
    cur.cell = -1; cur.dx = cur.dy = cur.ray = cur.end = cur.u = 0; cur.xalone = 0x7fffffff; cur.v = 0.f; nxt = cur; nx2 = cur;
    int item = wg_x * 16 + wv;
    K5_FETCH(cur, item)
    K5_FETCH(nxt, item + wgs_x * 16)
    for (; item < items; item += wgs_x * 16) {
        K5_FETCH(nx2, item + 2 * wgs_x * 16)
        if (cur.cell >= 0) {
            const int dx = cur.dx, dy = cur.dy;
            const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
            int first_free = 0x7fffffff, first_occ = 0x7fffffff;
            int lo = 0, hi = 2;
            const bool lone = K5_EXP == 3 || (K5_EXP != 4 && adx != ady && (adx > ady ? adx : ady) >= cur.xalone);     // (K5_EXP 3: every lane takes the lone path -- wrong results; 4: none does)
                // (beyond the step from which the line shares no cell: the table phase)
            if (adx != ady && !lone) {                                     // (a diagonal cell: the quadrant's other class touches it too)
                const bool xm = adx > ady;
                rs_range(start, xm ? (dx > 0 ? 0 : 1) : (dy > 0 ? 2 : 3), xm ? adx : ady, xm ? dy : dx, 0.5f, lo, hi);
            }
            bool mine = true;
            if (lone || hi - lo == 1) { if (cur.end) first_occ = cur.ray; else first_free = cur.ray; }
            else {
                int cls[2], a[2], b[2];
                const int ncls = rs_classes(dx, dy, cls, a, b);
                for (int k = 0; k < ncls; k++) {
                    rs_range(start, cls[k], a[k], b[k], 0.5f, lo, hi);
                    for (int ci = lo; ci < hi; ci++) {
                        const k5_line c = cand[ci];
                        const int h = k5_hit(c, a[k], b[k]);
                        if (h == 1) first_free = min(first_free, c.ray);
                        else if (h == 2) first_occ = min(first_occ, c.ray);
                    }
                }
                mine = min(first_free, first_occ) == cur.ray;              // else another line's lane owns this cell
            }
            if (mine) {
                float v = cur.v;
                int u = cur.u;
                k5_transition(L, v, u, first_free, first_occ, lo_free, lo_occ);
                k5_store_cell(L.cells + cur.cell, v, u);
#if HS_PROB_MODE == 0
                L.prob[cur.cell] = hs_prob_v(v);
#endif
            }
        }
        cur = nxt; nxt = nx2;
    }
#undef K5_FETCH
    if (BUILD && sec_out && (int)blockIdx.x == L.wg0) {                    // (uniform: the level's first workgroup)
        __syncthreads();
        k5_sector_bounds<(K5_LDS_LINES + 1023) / 1024>(byidx_s, n_pts, s_wtot, &s_wsum_all, s_bound, sec_out + lvl * K5_SEC);
    }
#ifdef K5_TIMES
    __syncthreads();                                                       // (the workgroup's last wavefront)
    K5_STAMP(3)
#endif
}

__global__ void k5_fill_cells(slamhip_cell *cells, float *prob, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float p0 = hs_prob_v(0.0f);
    for (; i < n; i += stride) { cells[i].value = 0.0f; cells[i].update_index = -1; prob[i] = p0; }   // LogOddsCell.Reset :38-42
}

// the cached probabilities of an uploaded mapArray
__global__ void k5_refresh_prob(const slamhip_cell *cells, float *prob, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) prob[i] = hs_prob_v(cells[i].value);
}
// GridMap.GetBitmapData (GridMap.cs:104-115)
__global__ void k5_bitmap(const slamhip_cell *cells, uint8_t *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = cells[i].value;
    const int sgn = (v > 0.0f) - (v < 0.0f);
    out[i] = (uint8_t)(127 - sgn * 127);                                   // :111
}
// GridMap.GetMapExtends (GridMap.cs:147-207): bounding rectangle of the cells whose Value != 0.  ext = {xMax, yMax, xMin, yMin},
// preset to {-1, -1, 10000, 10000} (:149-150 -- the reference's minima start at 10000 whatever the map size).
__global__ void k5_extends_init(int32_t *ext)
{
    if (threadIdx.x < 4) ext[threadIdx.x] = threadIdx.x < 2 ? -1 : 10000;
}
__global__ __launch_bounds__(256) void k5_extends(const slamhip_cell *cells, int w, int h, int32_t *ext)
{
    int xmax = -1, ymax = -1, xmin = 10000, ymin = 10000;
    for (int y = blockIdx.x; y < h; y += gridDim.x) {                      // one row per workgroup pass: coalesced reads
        const slamhip_cell *row = cells + (size_t)y * w;
        for (int x = threadIdx.x; x < w; x += 256)
            if (row[x].value != 0.0f) {                                          // :161 (a NaN cell counts, as in the reference)
                xmax = max(xmax, x); xmin = min(xmin, x);
                ymax = max(ymax, y); ymin = min(ymin, y);
            }
    }
    for (int m = 1; m < 64; m <<= 1) {
        xmax = max(xmax, __shfl_xor(xmax, m)); ymax = max(ymax, __shfl_xor(ymax, m));
        xmin = min(xmin, __shfl_xor(xmin, m)); ymin = min(ymin, __shfl_xor(ymin, m));
    }
    if ((threadIdx.x & 63) == 0 && xmax >= 0) {
        atomicMax(ext + 0, xmax); atomicMax(ext + 1, ymax);
        atomicMin(ext + 2, xmin); atomicMin(ext + 3, ymin);
    }
}
__global__ void k5_probability(const slamhip_cell *cells, const int32_t *idx, int n, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = hs_prob_v(cells[idx[i]].value);                   // OccGridMap.GetCachedProbability (:97-107)
}

// the two words of slamhip_hs_checksum (common.h: k_checksum's definition, per member of the cell): out[0] over the values' bit
// patterns, out[1] over the update indices
__global__ void __launch_bounds__(256) k5_checksum_cells(const slamhip_cell *__restrict__ cells, size_t n, unsigned long long *__restrict__ out)
{
    unsigned long long av = 0, au = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const slamhip_cell c = cells[i];
        av += sh_mix64(((unsigned long long)i << 32) | (unsigned long long)__float_as_uint(c.value));
        au += sh_mix64(((unsigned long long)i << 32) | (unsigned long long)(uint32_t)c.update_index);
    }
    for (int off = 32; off > 0; off >>= 1) { av += __shfl_down(av, off, 64); au += __shfl_down(au, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(out, av); atomicAdd(out + 1, au); }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static float prob_to_logodds(float prob) { const float odds = prob / (1.0f - prob); return logf(odds); }   // OccGridMap.cs:86-90

static hs_levels_arg levels_arg(slamhip_hs *hs)
{
    hs_levels_arg A;
    memset(&A, 0, sizeof(A));
    A.n = hs->n_levels;
    for (int l = 0; l < hs->n_levels; l++) {
        const hs_level &L = hs->lv[l];
        A.lv[l].w = L.w; A.lv[l].h = L.h; A.lv[l].cell = L.cell; A.lv[l].stm = L.stm;
        A.lv[l].map_t_world = L.map_t_world; A.lv[l].world_t_map = L.world_t_map;
        A.lv[l].prob = L.d_prob; A.lv[l].cells = L.d_cells; A.lv[l].iterations = L.iterations;
    }
    return A;
}

extern "C" int32_t slamhip_hs_destroy(slamhip_hs *hs)
{
    if (!hs) return SLAMHIP_OK;
    (void)hipSetDevice(hs->ctx->device);
    (void)hipStreamSynchronize(hs->ctx->stream);
    for (int l = 0; l < hs->n_levels; l++) {
        (void)hipFree(hs->lv[l].d_cells); (void)hipFree(hs->lv[l].d_prob);
    }
    (void)hipFree(hs->d_pts_base); (void)hipFree(hs->d_io);
    if (hs->h_pts) (void)hipHostFree(hs->h_pts);
    if (hs->ev_pts) (void)hipEventDestroy(hs->ev_pts);
    (void)hipFree(hs->d_k5_sec);
    (void)hipFree(hs->d_k5_byidx); (void)hipFree(hs->d_k5_cand); (void)hipFree(hs->d_k5_start); (void)hipFree(hs->d_k5_hdr);
    if (hs->h_io) (void)hipHostFree(hs->h_io);
    free(hs);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_reset(slamhip_hs *hs)
{
    SH_CHECK_ARG(hs);
    SH_HIP(hipSetDevice(hs->ctx->device));
    for (int l = 0; l < hs->n_levels; l++) {
        hs_level &L = hs->lv[l];
        hipLaunchKernelGGL(k5_fill_cells, dim3(1024), dim3(256), 0, hs->ctx->stream, L.d_cells, L.d_prob, (size_t)L.w * L.h);                   // GridMap.Reset :56-62
        L.curr_update_index = 0;                                           // OccGridMap.Reset :244-252
    }
    SH_HIP(hipStreamSynchronize(hs->ctx->stream));
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_create(slamhip_ctx *ctx, float cell_length, int32_t w, int32_t h, int32_t levels, slamhip_hs **out)
{
    SH_CHECK_ARG(ctx && out && levels >= 1 && levels <= HS_MAX_LEVELS && cell_length > 0.0f);
    SH_CHECK_ARG(w >= 2 && h >= 2 && w <= 32768 && h <= 32768 && (w >> (levels - 1)) >= 2 && (h >> (levels - 1)) >= 2);
    SH_HIP(hipSetDevice(ctx->device));
    slamhip_hs *hs = (slamhip_hs *)calloc(1, sizeof(slamhip_hs));
    if (!hs) SH_FAIL(SLAMHIP_ERR_NOMEM, "out of host memory");
    hs->ctx = ctx;
    hs->n_levels = levels;
    hs->odds_occ = 0.9f; hs->odds_free = 0.4f;                            // OccGridMap.cs:24-25
    hs->lo_free = prob_to_logodds(hs->odds_free);                         // :46
    hs->lo_occ = prob_to_logodds(hs->odds_occ);                           // :47
    float res = cell_length;
    int32_t rc = SLAMHIP_OK;
    for (int l = 0; l < levels && rc == SLAMHIP_OK; l++) {                // MapRepMultiMap.cs:49-57
        hs_level &L = hs->lv[l];
        L.w = w; L.h = h; L.cell = res; L.stm = 1.0f / res;               // MapProperties.cs:32
        L.iterations = 3;                                                 // OccGridMap.cs:53
        L.map_t_world = sh_m3x2_mul(sh_m3x2_scale(L.stm), sh_m3x2_translation(0.0f, 0.0f));   // GridMap.cs:46 (offset = 0)
        if (!sh_m3x2_invert(L.map_t_world, &L.world_t_map)) { slamhip_set_error("Map to world matrix is not invertible"); rc = SLAMHIP_ERR_INVALID; break; }  // :47-50
        const size_t n = (size_t)w * h;
        if (hipMalloc(&L.d_cells, sizeof(slamhip_cell) * n) != hipSuccess || hipMalloc(&L.d_prob, sizeof(float) * n) != hipSuccess) {
            slamhip_set_error("device allocation failed (level %d)", l); rc = SLAMHIP_ERR_NOMEM; break;
        }
        w /= 2; h /= 2;                                                   // :55
        res *= 2.0f;                                                      // :56
    }
    if (rc == SLAMHIP_OK) {
        hs->cap_io = 4096;
        if (hipMalloc(&hs->d_io, sizeof(float) * hs->cap_io) != hipSuccess || hipHostMalloc(&hs->h_io, sizeof(float) * hs->cap_io) != hipSuccess) {
            slamhip_set_error("device allocation failed"); rc = SLAMHIP_ERR_NOMEM;
        }
    }
    if (rc == SLAMHIP_OK) rc = slamhip_hs_reset(hs);
    if (rc != SLAMHIP_OK) { slamhip_hs_destroy(hs); return rc; }
    *out = hs;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_level_info(slamhip_hs *hs, int32_t level, int32_t *w, int32_t *h, float *cell)
{
    SH_CHECK_ARG(hs && level >= 0 && level < hs->n_levels);
    if (w) *w = hs->lv[level].w;
    if (h) *h = hs->lv[level].h;
    if (cell) *cell = hs->lv[level].cell;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_set_factors(slamhip_hs *hs, float free_f, float occ_f)
{
    SH_CHECK_ARG(hs);
    hs->odds_free = free_f; hs->lo_free = prob_to_logodds(free_f);        // OccGridMap.cs:58-66
    hs->odds_occ = occ_f;   hs->lo_occ = prob_to_logodds(occ_f);          // :71-79
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_set_iterations(slamhip_hs *hs, const int32_t *it)
{
    SH_CHECK_ARG(hs && it);
    for (int l = 0; l < hs->n_levels; l++) { SH_CHECK_ARG(it[l] >= 0 && it[l] <= 1000); hs->lv[l].iterations = it[l]; }
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_cells_upload(slamhip_hs *hs, int32_t level, const slamhip_cell *cells, size_t n)
{
    SH_CHECK_ARG(hs && cells && level >= 0 && level < hs->n_levels);
    hs_level &L = hs->lv[level];
    SH_CHECK_ARG(n == (size_t)L.w * L.h);
    SH_HIP(hipSetDevice(hs->ctx->device));
    // (the device holds the reference's own layout: a plain copy, then the cached probabilities)
    SH_HIP(hipMemcpyAsync(L.d_cells, cells, sizeof(slamhip_cell) * n, hipMemcpyHostToDevice, hs->ctx->stream));
    hipLaunchKernelGGL(k5_refresh_prob, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs->ctx->stream, (const slamhip_cell *)L.d_cells, L.d_prob, n);
    SH_HIP(hipStreamSynchronize(hs->ctx->stream));
    // keep the once-per-scan guards meaningful: the next scan's marks must exceed every stored index
    int mx = -1;
    for (size_t i = 0; i < n; i++) if (cells[i].update_index > mx) mx = cells[i].update_index;
    if (mx >= 0) {                      // marks of scan k are 3k+1 / 3k+2 (OccGridMap.cs:116-117,:144)
        const int need = (mx / 3 + 1) * 3;
        if (need > L.curr_update_index) L.curr_update_index = need;
    }
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_cells_download(slamhip_hs *hs, int32_t level, slamhip_cell *cells, size_t n)
{
    SH_CHECK_ARG(hs && cells && level >= 0 && level < hs->n_levels);
    hs_level &L = hs->lv[level];
    SH_CHECK_ARG(n == (size_t)L.w * L.h);
    SH_HIP(hipSetDevice(hs->ctx->device));
    SH_HIP(hipMemcpyAsync(cells, L.d_cells, sizeof(slamhip_cell) * n, hipMemcpyDeviceToHost, hs->ctx->stream));
    SH_HIP(hipStreamSynchronize(hs->ctx->stream));
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_bitmap_download(slamhip_hs *hs, int32_t level, uint8_t *out, size_t n)
{
    SH_CHECK_ARG(hs && out && level >= 0 && level < hs->n_levels);
    hs_level &L = hs->lv[level];
    SH_CHECK_ARG(n == (size_t)L.w * L.h);
    SH_HIP(hipSetDevice(hs->ctx->device));
    uint8_t *d = nullptr;
    SH_HIP(hipMalloc(&d, n));
    hipLaunchKernelGGL(k5_bitmap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hs->ctx->stream, (const slamhip_cell *)L.d_cells, d, n);
    hipError_t e = hipMemcpyAsync(out, d, n, hipMemcpyDeviceToHost, hs->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(hs->ctx->stream);
    (void)hipFree(d);
    SH_HIP(e);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_map_extends(slamhip_hs *hs, int32_t level, int32_t extends[4], int32_t *found)
{
    SH_CHECK_ARG(hs && extends && found && level >= 0 && level < hs->n_levels);
    hs_level &L = hs->lv[level];
    SH_HIP(hipSetDevice(hs->ctx->device));
    int32_t *d = nullptr;
    SH_HIP(hipMalloc(&d, 4 * sizeof(int32_t)));
    hipLaunchKernelGGL(k5_extends_init, dim3(1), dim3(64), 0, hs->ctx->stream, d);
    hipLaunchKernelGGL(k5_extends, dim3(L.h < 2048 ? L.h : 2048), dim3(256), 0, hs->ctx->stream, (const slamhip_cell *)L.d_cells, L.w, L.h, d);
    int32_t e4[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(e4, d, sizeof(e4), hipMemcpyDeviceToHost, hs->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(hs->ctx->stream);
    (void)hipFree(d);
    SH_HIP(e);
    // :186-205 -- all four must have moved off their start values, otherwise (false, 0, 0, 0, 0)
    const bool ok = e4[0] != -1 && e4[1] != -1 && e4[2] != 10000 && e4[3] != 10000;
    for (int i = 0; i < 4; i++) extends[i] = ok ? e4[i] : 0;
    *found = ok ? 1 : 0;
    return SLAMHIP_OK;
}

// Replica check (one match is too small to shard, the grids are replicas): checksums of a level's log-odds
// (bit patterns) and update indices behind everything enqueued so far; definition in common.h (sh_mix64 / k_checksum).
extern "C" int32_t slamhip_hs_checksum(slamhip_hs *hs, int32_t level, uint64_t out[2])
{
    SH_CHECK_ARG(hs && out && level >= 0 && level < hs->n_levels);
    hs_level &L = hs->lv[level];
    slamhip_ctx *ctx = hs->ctx;
    SH_HIP(hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    SH_HIP(hipMalloc(&d, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) {
        const size_t n = (size_t)L.w * L.h, want = (n + 2047) / 2048;
        hipLaunchKernelGGL(k5_checksum_cells, dim3((unsigned)(want < 1 ? 1 : want > 2048 ? 2048 : want)), dim3(256), 0, ctx->stream, (const slamhip_cell *)L.d_cells, n, d);
        e = hipMemcpyAsync(out, d, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    SH_HIP(e);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_probability(slamhip_hs *hs, int32_t level, const int32_t *indices, int32_t n, float *out)
{
    SH_CHECK_ARG(hs && indices && out && n > 0 && level >= 0 && level < hs->n_levels);
    hs_level &L = hs->lv[level];
    for (int i = 0; i < n; i++) SH_CHECK_ARG(indices[i] >= 0 && (size_t)indices[i] < (size_t)L.w * L.h);
    SH_HIP(hipSetDevice(hs->ctx->device));
    int32_t *di = nullptr; float *dout = nullptr;
    SH_HIP(hipMalloc(&di, sizeof(int32_t) * n));
    hipError_t e = hipMalloc(&dout, sizeof(float) * n);
    if (e == hipSuccess) e = hipMemcpyAsync(di, indices, sizeof(int32_t) * n, hipMemcpyHostToDevice, hs->ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k5_probability, dim3(sh_div_up(n, 256)), dim3(256), 0, hs->ctx->stream, (const slamhip_cell *)L.d_cells, di, n, dout);
        e = hipMemcpyAsync(out, dout, sizeof(float) * n, hipMemcpyDeviceToHost, hs->ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(hs->ctx->stream);
    (void)hipFree(di); (void)hipFree(dout);
    SH_HIP(e);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_set_scan(slamhip_hs *hs, const float *xy, int32_t n, const float origin[2])
{
    SH_CHECK_ARG(hs && n >= 0 && (xy || n == 0));
    SH_HIP(hipSetDevice(hs->ctx->device));
    hs->origin[0] = origin ? origin[0] : 0.0f;
    hs->origin[1] = origin ? origin[1] : 0.0f;
    hs->n_points = 0;                                  // (stays "no scan" if anything below fails)
    if (n == 0) return SLAMHIP_OK;
    if (n > hs->cap_points) {
        SH_HIP(hipStreamSynchronize(hs->ctx->stream));
        (void)hipFree(hs->d_pts_base); hs->d_pts_base = nullptr; hs->d_pts = nullptr; hs->cap_points = 0;
        if (hs->h_pts) { (void)hipHostFree(hs->h_pts); hs->h_pts = nullptr; }
        const int cap = (n + n / 4 + 64 + 1) & ~1;                         // (even: the upload launch moves 16-byte units)
        SH_HIP(hipMalloc(&hs->d_pts_base, sizeof(float2) * (size_t)cap * 2));
        hs->pts_use[0] = hs->pts_use[1] = 0;
        SH_HIP(hipHostMalloc(&hs->h_pts, sizeof(float2) * (size_t)cap + 64, hipHostMallocMapped | hipHostMallocCoherent));   // (+ the upload's completion word)
        memset(hs->h_pts + 2 * (size_t)cap, 0, 64);
        hs->upload_seq = 0;
        if (!hs->ev_pts) SH_HIP(hipEventCreateWithFlags(&hs->ev_pts, hipEventDisableTiming));
        hs->cap_points = cap;
        hs->pts_in_flight = false;
    }
    uint32_t *up_flag = (uint32_t *)(hs->h_pts + 2 * (size_t)hs->cap_points);
    if (hs->upload_pending) hs->upload_pending = false; // (the staged scan was never consumed: nothing was launched, the block is ours)
    else if (hs->pts_in_flight) {                       // the previous copy has left the staging block
        if (!hs->ctx->mail_off) SH_TRY(sh_upload_wait(hs->ctx, up_flag, hs->upload_seq));
        else SH_HIP(hipEventSynchronize(hs->ev_pts));
        hs->pts_in_flight = false;
    }
    memcpy(hs->h_pts, xy, sizeof(float) * 2 * (size_t)n);
    hs->pts_buf ^= 1;
    hs->d_pts = hs->d_pts_base + (size_t)hs->pts_buf * (size_t)hs->cap_points;
    if (hs->ctx->large_bar && hs->pts_use[hs->pts_buf] <= hs->launch_done) {
        // (the block is idle and the host can store into device memory: the upload is a copy by the CPU through the PCIe aperture,
        // see slamhip_cs_set_scan -- the match then reads its points from device memory instead of pulling them over PCIe)
        memcpy(hs->d_pts, xy, sizeof(float) * 2 * (size_t)n);
        __builtin_ia32_sfence();
        hs->upload_pending = false;
    } else if (!hs->ctx->mail_off) {                    // (see slamhip_cs_set_scan: the per-scan path is launches only, and the upload is left pending)
        hs->upload_pending = true;
        hs->upload_bytes = (sizeof(float) * 2 * (size_t)n + 15) & ~(size_t)15;
    } else {
        SH_HIP(hipMemcpyAsync(hs->d_pts, hs->h_pts, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, hs->ctx->stream));
        SH_HIP(hipEventRecord(hs->ev_pts, hs->ctx->stream));
        hs->pts_in_flight = true;
    }
    hs->n_points = n;
    return SLAMHIP_OK;
}

// launches the scan upload that slamhip_hs_set_scan left pending (every launch that reads the points calls it first)
static int32_t hs_flush_scan(slamhip_hs *hs)
{
    hs->pts_use[hs->pts_buf] = ++hs->launch_count;
    if (!hs->upload_pending) return SLAMHIP_OK;
    // (the upload state is committed once the launch that carries it is in the stream: on an error the scan stays pending)
    SH_TRY(sh_upload(hs->ctx, hs->h_pts, hs->d_pts, hs->upload_bytes, (uint32_t *)(hs->h_pts + 2 * (size_t)hs->cap_points), hs->upload_seq + 1));
    hs->upload_pending = false;
    hs->upload_seq++;
    hs->pts_in_flight = true;
    return SLAMHIP_OK;
}

static int32_t ensure_io(slamhip_hs *hs, int floats)
{
    if (floats <= hs->cap_io) return SLAMHIP_OK;
    (void)hipFree(hs->d_io); (void)hipHostFree(hs->h_io);
    hs->d_io = nullptr; hs->h_io = nullptr; hs->cap_io = 0;
    SH_HIP(hipMalloc(&hs->d_io, sizeof(float) * (size_t)floats * 2));
    SH_HIP(hipHostMalloc(&hs->h_io, sizeof(float) * (size_t)floats * 2));
    hs->cap_io = floats * 2;
    return SLAMHIP_OK;
}

// defer_seq: (single match through the mailbox only) return after the launch with the completion number in *defer_seq -- the
// caller holds the mailbox lock, enqueues what it wants behind the match and then calls match_collect
static int32_t match_collect(slamhip_hs *hs, uint32_t seq, float *out)
{
    slamhip_ctx *ctx = hs->ctx;
    SH_TRY(sh_flag_wait(ctx, ctx->mailbox + 15, seq));
    const volatile float *m = (const volatile float *)ctx->mailbox;
    out[0] = m[0]; out[1] = m[1]; out[2] = m[2];
    hs->launch_done = hs->match_launch_no;                                 // (the match has delivered: every launch before it has finished)
    return SLAMHIP_OK;
}
static int32_t run_match(slamhip_hs *hs, const float *hints, int B, float *out, int only_level, int iters, uint32_t *defer_seq = nullptr)
{
    SH_HIP(hipSetDevice(hs->ctx->device));
    slamhip_ctx *ctx = hs->ctx;
    sh_mail_guard lock(ctx);                                              // (the mailbox is the context's: common.h)
    SH_TRY(ensure_io(hs, 6 * B));
    float *d_in = hs->d_io, *d_out = hs->d_io + 3 * (size_t)B;
    const bool mail1 = B == 1 && !ctx->mail_off;                          // one match: the kernel itself delivers the pose to the host
    // ... and pulls a freshly set scan from the staging block itself (k4_match): no upload launch in the per-scan chain
    const bool pull = B == 1 && hs->upload_pending && hs->n_points > 0 && hs->n_points <= HS_LDS_PTS;
    const float2 *up_src = nullptr; float2 *up_dst = nullptr; uint32_t *up_flag = nullptr; uint32_t up_seq = 0;
    if (pull) {                                                           // (committed below, once the launch is in the stream)
        up_src = (const float2 *)hs->h_pts; up_dst = hs->d_pts; up_flag = (uint32_t *)(hs->h_pts + 2 * (size_t)hs->cap_points);
        up_seq = hs->upload_seq + 1;
        hs->pts_use[hs->pts_buf] = ++hs->launch_count;
    } else SH_TRY(hs_flush_scan(hs));
    hs->match_launch_no = hs->launch_count;
    if (B > 1) {
        memcpy(hs->h_io, hints, sizeof(float) * 3 * (size_t)B);
        SH_HIP(hipMemcpyAsync(d_in, hs->h_io, sizeof(float) * 3 * (size_t)B, hipMemcpyHostToDevice, ctx->stream));
    }
    uint32_t mail_seq = 0;
    if (defer_seq && !mail1) SH_FAIL(SLAMHIP_ERR_STATE, "a deferred match is a single match through the mailbox");
    {
        sh_timer t(ctx, SLAMHIP_K_HS_MATCH);
        // a single match is a latency chain on one compute unit, bound by VALU issue: 512 lanes (two wavefronts per SIMD, three
        // scan points per lane at 1080 rays; hs_hessian_block -- rocprofv3, 1080 rays, 3 levels: 1024 lanes 33.7 us, 512 23.9,
        // 256 25.9); batches run 256 lanes per hint (many workgroups per CU)
        static const int lanes1 = [] { const char *e = getenv("SLAMHIP_K4_LANES"); const int v = e ? atoi(e) : 0; return v == 256 || v == 1024 ? v : 512; }();
        const float *d_hints = B > 1 ? (const float *)d_in : (const float *)nullptr;
        const float3 h1 = make_float3(hints[0], hints[1], hints[2]);
        uint32_t *mb = mail1 ? ctx->mailbox : (uint32_t *)nullptr;
        if (mail1) mail_seq = sh_mail_seq_next(ctx);
        const int lanes = B <= 8 ? lanes1 : 256;
        // (a single full match in the per-scan flow brings helper workgroups: k4_match)
        static const int helpers_env = getenv("SLAMHIP_K4_HELPERS") ? atoi(getenv("SLAMHIP_K4_HELPERS")) : 1;
        const int helpers = B == 1 && only_level < 0 && hs->n_levels > 1 && hs->n_points > 0 && helpers_env > 0 ? 8 * helpers_env : 0;
#define K4_LAUNCH(BD) hipLaunchKernelGGL(k4_match<BD>, dim3(B + helpers), dim3(BD), 0, ctx->stream, levels_arg(hs), hs->d_pts, hs->n_points, d_hints, h1, d_out, \
                                         only_level, iters, mb, mail_seq, up_src, up_dst, up_flag, up_seq, helpers ? B : 0)
        if (lanes == 1024) K4_LAUNCH(1024);
        else if (lanes == 512) K4_LAUNCH(512);
        else K4_LAUNCH(256);
#undef K4_LAUNCH
    }
    SH_HIP(hipGetLastError());
    if (pull) { hs->upload_pending = false; hs->upload_seq = up_seq; hs->pts_in_flight = true; }
#ifdef K4_TIMES
    {
        static thread_local int calls = 0;
        if (B == 1 && ++calls == 20) {
            (void)hipStreamSynchronize(ctx->stream);
            unsigned long long h[16];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_k4_times), sizeof(h));
            static const char *nm[7] = { "transform+trig", "points", "wave sums", "barrier", "totals", "step..next (and the gap between matches)", "-" };
            double tot = 0;
            for (int k = 0; k < 7; k++) tot += (double)h[k];
            fprintf(stderr, "[k4 times] %d matches, thread 0 of the workgroup, us per match:", calls);
            for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.2f |", nm[k], (double)h[k] * 0.01 / calls);
            fprintf(stderr, " sum %.2f\n", tot * 0.01 / calls);
            unsigned long long hp[16];
            (void)hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_k4_pts), sizeof(hp));
            fprintf(stderr, "[k4 times] the points' phase per iteration, us:");
            for (int k = 0; k < 12; k++) fprintf(stderr, " %.2f", (double)hp[k] * 0.01 / calls);
            fprintf(stderr, "\n");
        }
    }
#endif
    if (mail1) {
        if (defer_seq) { *defer_seq = mail_seq; return SLAMHIP_OK; }
        return match_collect(hs, mail_seq, out);
    }
    SH_HIP(hipMemcpyAsync(hs->h_io + 3 * (size_t)B, d_out, sizeof(float) * 3 * (size_t)B, hipMemcpyDeviceToHost, ctx->stream));
    SH_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out, hs->h_io + 3 * (size_t)B, sizeof(float) * 3 * (size_t)B);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_match(slamhip_hs *hs, const float hint[3], float out[3])
{
    SH_CHECK_ARG(hs && hint && out);
    return run_match(hs, hint, 1, out, -1, 0);
}

extern "C" int32_t slamhip_hs_match_level(slamhip_hs *hs, int32_t level, const float hint[3], int32_t iterations, float out[3])
{
    SH_CHECK_ARG(hs && hint && out && level >= 0 && level < hs->n_levels && iterations >= 0);
    return run_match(hs, hint, 1, out, level, iterations);
}

extern "C" int32_t slamhip_hs_match_batch(slamhip_hs *hs, const float *hints, int32_t B, float *out)
{
    SH_CHECK_ARG(hs && hints && out && B > 0);
    return run_match(hs, hints, B, out, -1, 0);
}

extern "C" int32_t slamhip_hs_hessian(slamhip_hs *hs, int32_t level, const float pose_map[3], float H[9], float dTr[3])
{
    SH_CHECK_ARG(hs && pose_map && H && dTr && level >= 0 && level < hs->n_levels);
    SH_HIP(hipSetDevice(hs->ctx->device));
    slamhip_ctx *ctx = hs->ctx;
    SH_TRY(ensure_io(hs, 32));
    SH_TRY(hs_flush_scan(hs));
    memcpy(hs->h_io, pose_map, sizeof(float) * 3);
    SH_HIP(hipMemcpyAsync(hs->d_io, hs->h_io, sizeof(float) * 3, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k4_hessian, dim3(1), dim3(256), 0, ctx->stream, levels_arg(hs), level, hs->d_pts, hs->n_points,
                       (const float *)hs->d_io, hs->d_io + 16);
    SH_HIP(hipGetLastError());
    SH_HIP(hipMemcpyAsync(hs->h_io + 16, hs->d_io + 16, sizeof(float) * 12, hipMemcpyDeviceToHost, ctx->stream));
    SH_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(H, hs->h_io + 16, sizeof(float) * 9);
    memcpy(dTr, hs->h_io + 25, sizeof(float) * 3);
    return SLAMHIP_OK;
}

// the launches of UpdateByScan on the operator's stream; nothing comes back to the host
// gate_in: the device-gated form (k5_gate) -- `pose` is then only a stand-in, and the update indices are advanced by
// hs_update_commit once the host knows that the update took place
static void hs_update_commit(slamhip_hs *hs)
{
    for (int l = 0; l < hs->n_levels; l++) hs->lv[l].curr_update_index += 3;   // :144
    if (hs->k5_toggle_pending) hs->k5_sec_parity ^= 1;       // (the one-launch form wrote the other record set)
    hs->k5_toggle_pending = false;
}
static bool hs_update_gateable(slamhip_hs *hs)
{
    static const bool two_launch = getenv("SLAMHIP_K5_TWO_LAUNCHES") != nullptr, off = getenv("SLAMHIP_HS_NO_GATED_UPDATE") != nullptr;
    return !off && !two_launch && hs->n_points > 0 && hs->n_points <= K5_LDS_LINES && hs->ctx->timing == 0 && !hs->ctx->mail_off;
}
static int32_t hs_update_enqueue(slamhip_hs *hs, const float pose[3], const k5_gate *gate_in = nullptr)
{
    SH_CHECK_ARG(hs && pose);
    SH_HIP(hipSetDevice(hs->ctx->device));
    slamhip_ctx *ctx = hs->ctx;
    const int n = hs->n_points;
    k5_arg A;
    memset(&A, 0, sizeof(A));
    A.n = hs->n_levels;
    for (int l = 0; l < hs->n_levels; l++) {
        hs_level &L = hs->lv[l];
        A.lv[l].w = L.w; A.lv[l].h = L.h;
        A.lv[l].t = sh_m3x2_mul(sh_m3x2_mul(sh_m3x2_rotation(pose[2]), sh_m3x2_translation(pose[0], pose[1])),
                                sh_m3x2_scale(L.stm));                    // OccGridMap.cs:120-123
        A.lv[l].cells = L.d_cells; A.lv[l].prob = L.d_prob;
        A.lv[l].mark_free = L.curr_update_index + 1;                      // :116
        A.lv[l].mark_occ = L.curr_update_index + 2;                       // :117
    }
    if (n > 0) {
        SH_TRY(hs_flush_scan(hs));
        if (n > hs->cap_lines || !hs->d_k5_hdr) {
            (void)hipFree(hs->d_k5_byidx); (void)hipFree(hs->d_k5_cand); (void)hipFree(hs->d_k5_start); (void)hipFree(hs->d_k5_hdr);
            hs->d_k5_byidx = hs->d_k5_cand = nullptr; hs->d_k5_start = hs->d_k5_hdr = nullptr; hs->cap_lines = 0;
            const int cap = n + n / 4 + 64;
            SH_HIP(hipMalloc(&hs->d_k5_byidx, sizeof(k5_line) * (size_t)cap * HS_MAX_LEVELS));
            SH_HIP(hipMalloc(&hs->d_k5_cand, sizeof(k5_line) * (size_t)cap * HS_MAX_LEVELS));
            SH_HIP(hipMalloc(&hs->d_k5_start, sizeof(int) * (4 * RS_NBUCK + 1) * HS_MAX_LEVELS));
            SH_HIP(hipMalloc(&hs->d_k5_hdr, sizeof(int) * K5_HDR * HS_MAX_LEVELS));
            if (!hs->d_k5_sec) {
                SH_HIP(hipMalloc(&hs->d_k5_sec, sizeof(int) * 2 * HS_MAX_LEVELS * K5_SEC));
                SH_HIP(hipMemsetAsync(hs->d_k5_sec, 0xFF, sizeof(int) * 2 * HS_MAX_LEVELS * K5_SEC, ctx->stream));     // (no record: line count -1)
            }
            hs->cap_lines = cap;
        }
        int cgrid_x = 0;
        sh_timer t(ctx, SLAMHIP_K_HS_UPDATE);
        {   // ONE round of resident workgroups (two per CU: 512), shared out over the levels by the work they hold -- the cells a
            // scan touches, which halve from level to level (the zone around the begin cell is the same on every level: a floor
            // of 1/16 each).  (Round 2 shared them out by cell count with a floor of 1/8: 551 workgroups, i.e. a second round that
            // started when the first drained -- half of the kernel's 35 us.)
            // (every level needs a workgroup on each of the eight XCD sectors its lines are dealt to: below 8 workgroups per level
            // -- 8 levels x 16 as the floor is a sixteenth -- lines of the missing sectors would not be drawn; the override is clamped)
            static const int wgs_raw = getenv("SLAMHIP_K5_WGS") ? atoi(getenv("SLAMHIP_K5_WGS")) : 512;
            static const int wgs_env = wgs_raw < 128 ? 128 : wgs_raw;
            double tot = 0.0;
            for (int l = 0; l < hs->n_levels; l++) tot += (double)hs->lv[l].w + (double)hs->lv[l].h;
            int first = 0, left = wgs_env;
            for (int l = 0; l < hs->n_levels; l++) {
                const int floor_k = wgs_env / 16 > 0 ? wgs_env / 16 : 1;
                int k = (int)((double)wgs_env * ((double)hs->lv[l].w + (double)hs->lv[l].h) / tot);
                if (k < floor_k) k = floor_k;
                const int must_leave = (hs->n_levels - 1 - l) * floor_k;   // (the levels still to come keep their floor)
                if (k > left - must_leave) k = left - must_leave > 1 ? left - must_leave : 1;
                if (k < 8) k = 8;                                  // (one workgroup per XCD sector at least, whatever the shares)
                A.lv[l].wg0 = first; A.lv[l].wgn = k;
                first += k; left -= k;
            }
            cgrid_x = first;
        }
        const dim3 cgrid(cgrid_x);
        static const bool two_launch = getenv("SLAMHIP_K5_TWO_LAUNCHES") != nullptr;       // (tests: the large-scan path on ordinary scans)
        const bool build = n <= K5_LDS_LINES && !two_launch;
        static std::atomic<unsigned long long> attr_set{0};                                             // one bit per device (the attribute is the device's)
        if (!((attr_set.load(std::memory_order_acquire) >> (ctx->device & 63)) & 1ull)) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k5_cells<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)k5_lds_bytes(true, K5_LDS_LINES)); attr_set.fetch_or(1ull << (ctx->device & 63), std::memory_order_release); }
        if (!build)      // all levels in every launch (MapRepMultiMap.cs:76)
            hipLaunchKernelGGL(k5_prepare, dim3(hs->n_levels), dim3(1024), 0, ctx->stream, A, (const float2 *)hs->d_pts, n, hs->origin[0],
                               hs->origin[1], hs->cap_lines, (k5_line *)hs->d_k5_byidx, (k5_line *)hs->d_k5_cand, hs->d_k5_start, hs->d_k5_hdr);
        static const bool no_sectors = getenv("SLAMHIP_K5_EQUAL_SECTORS") != nullptr;       // (tuning: the sectors of phase 2 by count, as scans too large for the LDS tables have them)
        hs->k5_toggle_pending = build;
        k5_gate gate;
        memset(&gate, 0, sizeof(gate));
        if (gate_in) {
            if (!build) SH_FAIL(SLAMHIP_ERR_STATE, "the gated update needs the one-launch form");
            gate = *gate_in; gate.on = 1;
            for (int l = 0; l < hs->n_levels; l++) gate.stm[l] = hs->lv[l].stm;
        }
        if (build) {
            const int *sec_in = no_sectors ? nullptr : hs->d_k5_sec + (size_t)hs->k5_sec_parity * HS_MAX_LEVELS * K5_SEC;
            int *sec_out = no_sectors ? nullptr : hs->d_k5_sec + (size_t)(hs->k5_sec_parity ^ 1) * HS_MAX_LEVELS * K5_SEC;
            hipLaunchKernelGGL(k5_cells<true>, cgrid, dim3(1024), k5_lds_bytes(true, n), ctx->stream, A, hs->cap_lines, (const float2 *)hs->d_pts, n,
                               hs->origin[0], hs->origin[1], (const k5_line *)hs->d_k5_byidx, (const k5_line *)hs->d_k5_cand, (const int *)hs->d_k5_start,
                               (const int *)hs->d_k5_hdr, hs->lo_free, hs->lo_occ, sec_in, sec_out, gate);
        } else
            hipLaunchKernelGGL(k5_cells<false>, cgrid, dim3(1024), k5_lds_bytes(false, n), ctx->stream, A, hs->cap_lines, (const float2 *)hs->d_pts, n,
                               hs->origin[0], hs->origin[1], (const k5_line *)hs->d_k5_byidx, (const k5_line *)hs->d_k5_cand, (const int *)hs->d_k5_start,
                               (const int *)hs->d_k5_hdr, hs->lo_free, hs->lo_occ, (const int *)nullptr, (int *)nullptr, gate);
    }
    SH_HIP(hipGetLastError());
#ifdef K5_TIMES
    {
        static thread_local int calls = 0;
        if (n > 0 && ++calls == 12) {
            (void)hipStreamSynchronize(ctx->stream);
            std::vector<unsigned long long> h(1024 * 4);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_k5_times), sizeof(unsigned long long) * h.size());
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int i = 0; i < 1024; i++) if (h[i * 4] && h[i * 4 + 3] >= h[i * 4]) { t0 = std::min(t0, h[i * 4]); t1 = std::max(t1, h[i * 4 + 3]); }
            fprintf(stderr, "[k5 times] span %.2f us; per level, first thread of each workgroup, mean (max) us:\n", (double)(t1 - t0) * 0.01);
            std::vector<unsigned long long> hsub(1024 * 8);
            (void)hipMemcpyFromSymbol(hsub.data(), HIP_SYMBOL(g_k5_sub), sizeof(unsigned long long) * hsub.size());
            for (int l = 0; l < hs->n_levels; l++) {
                {   // the table phase in parts, mean over the level's workgroups: start -> barrier 1 -> lines -> prefix a -> prefix b -> scatter -> tables done
                    double part[6] = { 0, 0, 0, 0, 0, 0 }; int cc = 0;
                    for (int i = A.lv[l].wg0; i < A.lv[l].wg0 + A.lv[l].wgn && i < 1024; i++) if (h[i * 4] && hsub[i * 8 + 4] >= h[i * 4]) {
                        unsigned long long prev = h[i * 4];
                        for (int k = 0; k < 5; k++) { part[k] += (double)(hsub[i * 8 + k] - prev) * 0.01; prev = hsub[i * 8 + k]; }
                        part[5] += (double)(h[i * 4 + 1] - prev) * 0.01; cc++;
                    }
                    if (cc) fprintf(stderr, "   level %d table phase: zero+barrier %.2f | lines+reductions+barrier %.2f | prefix a %.2f | prefix b %.2f | scatter+barrier %.2f | alone pass + rest %.2f\n",
                                    l, part[0] / cc, part[1] / cc, part[2] / cc, part[3] / cc, part[4] / cc, part[5] / cc);
                }
                double acc[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 }, end = 0, endmx = 0, st = 0; int c = 0;
                for (int i = A.lv[l].wg0; i < A.lv[l].wg0 + A.lv[l].wgn && i < 1024; i++) if (h[i * 4] && h[i * 4 + 3] >= h[i * 4]) {
                    for (int k = 0; k < 3; k++) { const double d = (double)(h[i * 4 + k + 1] - h[i * 4 + k]) * 0.01; acc[k] += d; mx[k] = std::max(mx[k], d); }
                    const double e = (double)(h[i * 4 + 3] - t0) * 0.01; end += e; endmx = std::max(endmx, e); st += (double)(h[i * 4] - t0) * 0.01; c++;
                }
                {   // per XCD sector of the level (workgroup w of the level draws sector w % 8): mean time beyond the zone
                    fprintf(stderr, "   level %d, beyond + drain per sector:", l);
                    for (int x = 0; x < 8; x++) {
                        double a2 = 0; int c2 = 0;
                        for (int i = A.lv[l].wg0 + x; i < A.lv[l].wg0 + A.lv[l].wgn && i < 1024; i += 8) if (h[i * 4] && h[i * 4 + 3] >= h[i * 4]) { a2 += (double)(h[i * 4 + 3] - h[i * 4 + 2]) * 0.01; c2++; }
                        fprintf(stderr, " %.1f", c2 ? a2 / c2 : 0.0);
                    }
                    fprintf(stderr, "\n");
                }
                if (c) fprintf(stderr, "   level %d (%d workgroups): start +%.2f | tables %.2f (%.2f) | zone %.2f (%.2f) | beyond + drain %.2f (%.2f) | end +%.2f (%.2f)\n",
                               l, c, st / c, acc[0] / c, mx[0], acc[1] / c, mx[1], acc[2] / c, mx[2], end / c, endmx);
            }
        }
    }
#endif
    if (!gate_in) hs_update_commit(hs);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hs_update_by_scan(slamhip_hs *hs, const float pose[3])
{
    SH_CHECK_ARG(hs);
    sh_mail_guard lock(hs->ctx);
    SH_TRY(hs_update_enqueue(hs, pose));
    SH_TRY(sh_publish(hs->ctx, nullptr, 0));
    return sh_host_wait(hs->ctx);
}

// ---- HectorSLAMProcessor (Main/HectorSLAMProcessor.cs) ---------------------------------------------------------------
struct slamhip_hsproc {
    slamhip_hs *hs;
    float start_pose[3], match_pose[3], last_update_pose[3];
    float match_timing, update_timing;
    float min_dist, min_angle;
    unsigned upd_hist;                                     // the last scans' update decisions, newest in bit 0
};

static const float F_MIN = -3.40282347e+38f;       // float.MinValue

extern "C" int32_t slamhip_hsproc_create(slamhip_ctx *ctx, float res, int32_t w, int32_t h, const float start[3], int32_t depth,
                                         slamhip_hsproc **out)
{
    SH_CHECK_ARG(ctx && start && out);
    slamhip_hs *hs = nullptr;
    SH_TRY(slamhip_hs_create(ctx, res, w, h, depth, &hs));                // :71
    slamhip_hsproc *p = (slamhip_hsproc *)calloc(1, sizeof(*p));
    if (!p) { slamhip_hs_destroy(hs); SH_FAIL(SLAMHIP_ERR_NOMEM, "out of host memory"); }
    p->hs = hs;
    memcpy(p->start_pose, start, sizeof(float) * 3);
    memcpy(p->match_pose, start, sizeof(float) * 3);                      // :75
    p->last_update_pose[0] = p->last_update_pose[1] = p->last_update_pose[2] = F_MIN;   // :76
    p->min_dist = 0.3f; p->min_angle = 0.13f;                             // :51,:56
    *out = p;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hsproc_destroy(slamhip_hsproc *p)
{
    if (!p) return SLAMHIP_OK;
    slamhip_hs_destroy(p->hs);
    free(p);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hsproc_reset(slamhip_hsproc *p)
{
    SH_CHECK_ARG(p);
    SH_TRY(slamhip_hs_reset(p->hs));                                      // :133
    memcpy(p->match_pose, p->start_pose, sizeof(float) * 3);              // :136
    p->last_update_pose[0] = p->last_update_pose[1] = p->last_update_pose[2] = F_MIN;   // :137
    p->upd_hist = 0;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hsproc_update(slamhip_hsproc *p, const float *xy, int32_t n, const float origin[2],
                                         const float hint[3], int32_t map_without_matching, int32_t *out_updated)
{
    SH_CHECK_ARG(p && hint);
    SH_TRY(slamhip_hs_set_scan(p->hs, xy, n, origin));
    static const bool wait_update = getenv("SLAMHIP_HS_WAIT_UPDATE") != nullptr;
    // (worth it when the update does take place: a gated launch that returns at once still costs the stream ~15 us -- 512 workgroups
    // of 1024 lanes are dispatched to find that out -- so the flow is taken while the last two scans both updated the map: measured,
    // every scan updating 70 -> 66 us per scan; one scan in five, where it is never taken, 55 either way, 71 if it always were)
    if (!map_without_matching && !wait_update && (p->upd_hist & 3u) == 3u && hs_update_gateable(p->hs)) {
        // The per-scan flow on the device: match, then the grid update gated by the processor's own test (k5_gate) -- both enqueued
        // before the host has the pose, which it then takes from the mailbox and puts to the same test for its own books.
        slamhip_hs *hs = p->hs;
        sh_mail_guard lock(hs->ctx);
        auto t0 = std::chrono::steady_clock::now();
        float m[3];
        uint32_t seq = 0;
        SH_TRY(run_match(hs, hint, 1, m, -1, 0, &seq));                   // :93
        k5_gate g;
        memset(&g, 0, sizeof(g));
        g.d_pose = hs->d_io + 3;                                          // (the single match's result in device memory: run_match)
        memcpy(g.last, p->last_update_pose, sizeof(g.last));
        g.min_dist = p->min_dist; g.min_angle = p->min_angle;
        const int32_t rc_u = hs_update_enqueue(hs, hint, &g);
        auto t1 = std::chrono::steady_clock::now();
        SH_TRY(match_collect(hs, seq, m));
        SH_TRY(rc_u);
        memcpy(p->match_pose, m, sizeof(m));
        auto t2 = std::chrono::steady_clock::now();
        const float ms_u = std::chrono::duration<float, std::milli>(t1 - t0).count();      // (launches of match + update; the match's share is a few us)
        const float ms_m = std::chrono::duration<float, std::milli>(t2 - t0).count();
        p->match_timing = (3.0f * p->match_timing + ms_m) / 4.0f;         // :96
        int updated = 0;
        if (hs_moved_enough(p->match_pose, p->last_update_pose, p->min_dist, p->min_angle)) {   // :107-108, as the kernel decided
            hs_update_commit(hs);
            p->update_timing = (3.0f * p->update_timing + ms_u) / 4.0f;   // :115 (the time of the enqueue)
            memcpy(p->last_update_pose, p->match_pose, sizeof(float) * 3);    // :118
            updated = 1;                                                  // :122
        } else hs->k5_toggle_pending = false;
        p->upd_hist = (p->upd_hist << 1) | (unsigned)updated;
        if (out_updated) *out_updated = updated;
        return SLAMHIP_OK;
    }
    if (!map_without_matching) {                                          // :89
        auto t0 = std::chrono::steady_clock::now();
        float m[3];
        SH_TRY(slamhip_hs_match(p->hs, hint, m));                         // :93
        memcpy(p->match_pose, m, sizeof(m));
        const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        p->match_timing = (3.0f * p->match_timing + ms) / 4.0f;           // :96
    } else {
        memcpy(p->match_pose, hint, sizeof(float) * 3);                   // :100
    }
    int updated = 0;
    if (hs_moved_enough(p->match_pose, p->last_update_pose, p->min_dist, p->min_angle) ||   // :107-108
        map_without_matching) {                                           // :109
        // The grid update returns nothing to the host: it is enqueued and runs on while the caller prepares its next scan --
        // the next match, a download or an export is ordered behind it on the operator's stream (UpdateTiming :115 is then
        // the time of the enqueue; SLAMHIP_HS_WAIT_UPDATE=1 waits for the update as before).
        auto t0 = std::chrono::steady_clock::now();
        if (wait_update) { SH_TRY(slamhip_hs_update_by_scan(p->hs, p->match_pose)); }   // :112
        else { SH_TRY(hs_update_enqueue(p->hs, p->match_pose)); }
        const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        p->update_timing = (3.0f * p->update_timing + ms) / 4.0f;         // :115
        memcpy(p->last_update_pose, p->match_pose, sizeof(float) * 3);    // :118
        updated = 1;                                                      // :122
    }
    p->upd_hist = (p->upd_hist << 1) | (unsigned)updated;
    if (out_updated) *out_updated = updated;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hsproc_get(slamhip_hsproc *p, float match_pose[3], float last[3], float *mt, float *ut)
{
    SH_CHECK_ARG(p);
    if (match_pose) memcpy(match_pose, p->match_pose, sizeof(float) * 3);
    if (last) memcpy(last, p->last_update_pose, sizeof(float) * 3);
    if (mt) *mt = p->match_timing;
    if (ut) *ut = p->update_timing;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hsproc_set_thresholds(slamhip_hsproc *p, float min_dist, float min_angle)
{
    SH_CHECK_ARG(p);
    p->min_dist = min_dist; p->min_angle = min_angle;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_hsproc_hs(slamhip_hsproc *p, slamhip_hs **out)
{
    SH_CHECK_ARG(p && out);
    *out = p->hs;
    return SLAMHIP_OK;
}
