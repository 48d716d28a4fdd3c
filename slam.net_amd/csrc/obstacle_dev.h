// obstacle_dev.h -- the two steps of the ObstacleMap update (K3) as device functions: obstacle.hip launches them as
// kernels of their own; the fused search + update rides them on the HoleMap update's ONE launch (extra workgroups of
// k2_pixels, holemap.hip): the ray walks of this scan, and the cell pass of the PREVIOUS scan -- the two steps of one scan
// need a launch boundary between them, so the per-scan scratch (hits, noHit) is double-buffered and the cell pass trails one
// scan behind; whoever reads or writes the ObstacleMap flushes it first (cs_obstacle_flush).  No launch of its own per scan.
#pragma once
#include "common.h"
#include "det_trig.h"

struct k3_ride {
    int ray_blocks, cell_blocks;       // extra 1024-thread workgroups: this scan's ray walks, the pending cell pass (0: none)
    const float2 *pts; int n_points, size; float scale; const float *d_pose; float4 h_pxcs;
    uint32_t *hits; uint8_t *nohit; int chunks_per_ray;                   // this scan's scratch
    int8_t *map; int n_cells; uint32_t *cell_hits; uint8_t *cell_nohit; int cell_max_hits;   // the pending pass: its scratch, its MaxObstacleHits
};

// one wavefront per (ray, 64 iterations of the walk): w = ray * chunks_per_ray + chunk, one lane per iteration
__device__ static inline void k3_rays_unit(int w, int lane, const float2 *__restrict__ pts, int n_points, int size, float scale,
                                           const float *d_pose, float4 h_pxcs, uint32_t *__restrict__ hits,
                                           uint8_t *__restrict__ nohit, int chunks_per_ray)
{
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
    const long long i = (long long)chunk * 64 + lane;
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

// one thread per cell: hits + decay, and the per-scan scratch is cleared for the next scan
__device__ static inline void k3_apply_cell(int i, int8_t *__restrict__ map, int n_cells, uint32_t *__restrict__ hits,
                                            uint8_t *__restrict__ nohit, int max_hits)
{
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
