// obstacle.hip -- K3: ObstacleMap raster update, bit-exact (gfx950 only).
//
// Replaces UpdateObstacleMap (CoreSLAM/CoreSLAMProcessor.cs:540-593) and DrawLaserRayOnObstacleMap
// (:456-490).  Unlike the HoleMap blend this update is order-independent:
//   - traversed cells only set a per-scan flag (noHitMap, :483; idempotent plain stores);
//   - the endpoint cell does a saturating ++ below MaxObstacleHits (:474-477): k hits on a cell with
//     value v give v + min(k, max(0, Max - v)), so hits are counted with integer atomics and applied once;
//   - the decay pass (:576-592) runs after all rays, on the value that already includes the hits.
// One wavefront per (ray, 64 iterations of the walk), one lane per iteration: after i iterations of the Rosetta-style
// walk (:458-488) the position is i steps along the major axis and max(0, ceil((i*minor - major/2) / major)) along
// the minor one (tests/test_closed_forms.py checks this against the literal loop); the walk is monotone in x and
// y, so the cells inside the map are a prefix of it (:465-469 breaks at the first one outside).  One thread per cell
// then applies hits + decay and clears the per-scan scratch.
#include "cs_internal.h"
#include "det_trig.h"

__global__ void __launch_bounds__(256)
k3_rays(const float2 *__restrict__ pts, int n_points, int size, float scale, const float *d_pose, float4 h_pxcs,
        uint32_t *__restrict__ hits, uint8_t *__restrict__ nohit, int chunks_per_ray)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);             // (ray, chunk), wave-uniform
    const int ray = w / chunks_per_ray, chunk = w - ray * chunks_per_ray;
    if (ray >= n_points) return;
    float4 q = h_pxcs;
    if (d_pose) {
        float s, c;
        sh_det_sincosf(d_pose[2], &s, &c);
        q.x = d_pose[0] * scale + 0.5f;                                    // :545
        q.y = d_pose[1] * scale + 0.5f;                                    // :546
        q.z = c * scale;                                                   // :547
        q.w = s * scale;                                                   // :548
    }
    const int x1 = sh_f2i(q.x), y1 = sh_f2i(q.y);                          // :553-554
    if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) return;              // :557-560
    const float2 p = pts[ray];
    float fx = q.x + q.z * p.x;  fx = fx - q.w * p.y;                      // :566
    float fy = q.y + q.w * p.x;  fy = fy + q.z * p.y;                      // :567
    const int x2 = sh_f2i(fx), y2 = sh_f2i(fy);
    const int ddx = sh_wsub(x2, x1), ddy = sh_wsub(y2, y1);
    if (ddx == INT32_MIN || ddy == INT32_MIN) return;                      // Math.Abs overflow (throws in C#)
    const int dx = sh_abs(ddx), sx = sh_sign(ddx);                         // :458
    const int dy = sh_abs(ddy), sy = sh_sign(ddy);                         // :459
    const long long n = dx > dy ? dx : dy;                                 // iterations to the end point
    const long long i = (long long)chunk * 64 + (threadIdx.x & 63);
    // the part of the walk that can lie in the map is shorter than 2 * size iterations (host: chunks_per_ray)
    if (i > n) return;
    long long ax, ay;                                                      // steps taken along x / y before iteration i
    if (dx > dy) {
        const long long num = i * dy - dx / 2;                             // err0 = dx / 2 (:460)
        ax = i; ay = num <= 0 ? 0 : (num + dx - 1) / dx;
    } else {
        const long long num = i * dx - dy / 2;                             // err0 = -dy / 2 = -(dy / 2) in C#
        ay = i; ax = (num <= 0 || dy == 0) ? 0 : (num + dy - 1) / dy;
    }
    const long long X = x1 + sx * ax, Y = y1 + sy * ay;
    if (X < 0 || X >= size || Y < 0 || Y >= size) return;                  // :465-469 (everything after it is outside too)
    const int idx = (int)Y * size + (int)X;
    if (i == n) atomicAdd(&hits[idx], 1u);                                 // :471-477 (applied in k3_apply)
    else nohit[idx] = 1;                                                   // :483
}

__global__ void __launch_bounds__(256)
k3_apply(int8_t *__restrict__ map, int n_cells, uint32_t *__restrict__ hits, uint8_t *__restrict__ nohit, int max_hits)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cells) return;
    const uint32_t h = hits[i];
    const uint8_t nh = nohit[i];
    if (h == 0 && nh == 0) return;
    int v = map[i];
    if (h) {
        const int m = (int)(int8_t)max_hits;                               // sbyte MaxObstacleHits (:101)
        if (v < m) { const int room = m - v; v += (h < (uint32_t)room) ? (int)h : room; }   // :474-477, k times
        hits[i] = 0;
    }
    if (nh) {
        if (v < 0) v++;                                                    // :582-585
        else if (v > 0) v--;                                               // :586-589
        nohit[i] = 0;                                                      // next scan's ArrayEx.Fill(noHitMap,false) :542
    }
    map[i] = (int8_t)v;
}

int32_t cs_obstacle_alloc(slamhip_cs *cs)
{
    const size_t n = (size_t)cs->os * cs->os;
    SH_HIP(hipMalloc(&cs->d_o_hits, sizeof(uint32_t) * n));
    SH_HIP(hipMalloc(&cs->d_o_nohit, n));
    SH_HIP(hipMemsetAsync(cs->d_o_hits, 0, sizeof(uint32_t) * n, cs->ctx->stream));
    SH_HIP(hipMemsetAsync(cs->d_o_nohit, 0, n, cs->ctx->stream));
    return SLAMHIP_OK;
}

void cs_obstacle_free(slamhip_cs *cs)
{
    (void)hipFree(cs->d_o_hits);
    (void)hipFree(cs->d_o_nohit);
}

int32_t cs_launch_obstacle_update(slamhip_cs *cs, const float *d_pose, float4 h_pxcs, int max_hits)
{
    slamhip_ctx *ctx = cs->ctx;
    const int n = cs->n_points;
    if (n <= 0) return SLAMHIP_OK;
    const int cells = cs->os * cs->os;
    sh_timer t(ctx, SLAMHIP_K_CS_OBSTACLE);
    // a walk stays in the map for at most `size` iterations (one major-axis step each): iterations 0 .. size
    const int chunks_per_ray = sh_div_up(cs->os + 1, 64);
    hipLaunchKernelGGL(k3_rays, dim3(sh_div_up(n * chunks_per_ray, 4)), dim3(256), 0, ctx->stream,
                       cs->d_pts, n, cs->os, cs->oscale, d_pose, h_pxcs, cs->d_o_hits, cs->d_o_nohit, chunks_per_ray);
    hipLaunchKernelGGL(k3_apply, dim3(sh_div_up(cells, 256)), dim3(256), 0, ctx->stream,
                       cs->d_obst, cells, cs->d_o_hits, cs->d_o_nohit, max_hits);
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}
