// obstacle_dev.h -- the two steps of the ObstacleMap update (K3) as device functions: obstacle.hip launches them as
// kernels of their own; the fused search + update rides them on the HoleMap update's ONE launch (k2_pixels, holemap.hip: its
// wavefronts take a ray's walk and 64 cells each at their start, under the memory round trips the HoleMap tables wait for
// anyway): the ray walks of this scan, and the cell pass of the PREVIOUS scan -- the two steps of one scan need a launch
// boundary between them, so the per-scan scratch (hits, noHit) is double-buffered and the cell pass trails one scan behind;
// whoever reads or writes the ObstacleMap flushes it first (cs_obstacle_flush).  No launch of its own per scan.
#pragma once
#include "common.h"
#include "det_trig.h"

struct k3_ride {
    int on;                            // the ObstacleMap update of this scan rides on the launch
    const float2 *pts; int n_points, size; float scale; const float *d_pose; float4 h_pxcs;
    uint32_t *hits; uint8_t *nohit; int chunks_per_ray;                   // this scan's scratch
    int8_t *map; int n_cells; uint32_t *cell_hits; uint8_t *cell_nohit; int cell_max_hits;   // the pending pass (n_cells 0: none): its scratch, its MaxObstacleHits
};

// UpdateObstacleMap :545-548
__device__ static inline float4 k3_pxcs(const float *d_pose, float4 h_pxcs, float scale)
{
    float4 q = h_pxcs;
    if (d_pose) {
        float s, c;
        sh_det_sincosf(d_pose[2], &s, &c);
        q.x = d_pose[0] * scale + 0.5f;                                    // :545
        q.y = d_pose[1] * scale + 0.5f;                                    // :546
        q.z = c * scale;                                                   // :547
        q.w = s * scale;                                                   // :548
    }
    return q;
}

// a ray's walk (DrawLaserRayOnObstacleMap :456-490): its constants, then iteration i in closed form
struct k3_walk { int x1, y1, sx, sy, dx, dy; long long n; bool ok; };
__device__ static inline k3_walk k3_walk_setup(float2 p, float4 q, int size)
{
    k3_walk w;
    w.ok = false; w.n = -1; w.sx = w.sy = w.dx = w.dy = 0;
    w.x1 = sh_f2i(q.x); w.y1 = sh_f2i(q.y);                                // :553-554
    if (w.x1 < 0 || w.x1 >= size || w.y1 < 0 || w.y1 >= size) return w;   // :557-560
    float fx = q.x + q.z * p.x;  fx = fx - q.w * p.y;                      // :566
    float fy = q.y + q.w * p.x;  fy = fy + q.z * p.y;                      // :567
    const int x2 = sh_f2i(fx), y2 = sh_f2i(fy);
    const int ddx = sh_wsub(x2, w.x1), ddy = sh_wsub(y2, w.y1);
    if (ddx == INT32_MIN || ddy == INT32_MIN) return w;                    // Math.Abs overflow (throws in C#)
    w.dx = sh_abs(ddx); w.sx = sh_sign(ddx);                               // :458
    w.dy = sh_abs(ddy); w.sy = sh_sign(ddy);                               // :459
    w.n = w.dx > w.dy ? w.dx : w.dy;                                       // iterations to the end point
    w.ok = true;
    return w;
}
__device__ static inline void k3_walk_iter(const k3_walk &w, long long i, int size, uint32_t *__restrict__ hits, uint8_t *__restrict__ nohit)
{
    if (!w.ok || i > w.n) return;
    long long ax, ay;                                                      // steps taken along x / y before iteration i
    if ((w.dx | w.dy) < 16384 && i < 65536) {                              // (the ordinary case in 32-bit arithmetic: a 64-bit division is a hundred instructions)
        const int major = w.dx > w.dy ? w.dx : w.dy, minor = w.dx > w.dy ? w.dy : w.dx;
        const int num = (int)i * minor - major / 2;                        // err0 = dx / 2 (:460) resp. -(dy / 2)
        const int st = (num <= 0 || major == 0) ? 0 : (int)((unsigned)(num + major - 1) / (unsigned)major);
        if (w.dx > w.dy) { ax = i; ay = st; } else { ay = i; ax = st; }
    } else if (w.dx > w.dy) {
        const long long num = i * w.dy - w.dx / 2;                         // err0 = dx / 2 (:460)
        ax = i; ay = num <= 0 ? 0 : (num + w.dx - 1) / w.dx;
    } else {
        const long long num = i * w.dx - w.dy / 2;                         // err0 = -dy / 2 = -(dy / 2) in C#
        ay = i; ax = (num <= 0 || w.dy == 0) ? 0 : (num + w.dy - 1) / w.dy;
    }
    const long long X = w.x1 + w.sx * ax, Y = w.y1 + w.sy * ay;
    if (X < 0 || X >= size || Y < 0 || Y >= size) return;                  // :465-469 (everything after it is outside too)
    const int idx = (int)Y * size + (int)X;
    if (i == w.n) atomicAdd(&hits[idx], 1u);                               // :471-477 (applied in k3_apply)
    else nohit[idx] = 1;                                                   // :483
}

// one wavefront per (ray, 64 iterations of the walk): w = ray * chunks_per_ray + chunk, one lane per iteration
// (the part of a walk that can lie in the map is shorter than 2 * size iterations -- host: chunks_per_ray)
__device__ static inline void k3_rays_unit(int w, int lane, const float2 *__restrict__ pts, int n_points, int size, float scale,
                                           const float *d_pose, float4 h_pxcs, uint32_t *__restrict__ hits,
                                           uint8_t *__restrict__ nohit, int chunks_per_ray)
{
    const int ray = w / chunks_per_ray, chunk = w - ray * chunks_per_ray;
    if (ray >= n_points) return;
    const k3_walk wk = k3_walk_setup(pts[ray], k3_pxcs(d_pose, h_pxcs, scale), size);
    k3_walk_iter(wk, (long long)chunk * 64 + lane, size, hits, nohit);
}

// a cell of the cell pass whose three loads the caller issued earlier
__device__ static inline void k3_apply_loaded(int i, uint32_t h, uint8_t nh, int v, int8_t *__restrict__ map, uint32_t *__restrict__ hits,
                                              uint8_t *__restrict__ nohit, int max_hits)
{
    if (h == 0 && nh == 0) return;
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

// one thread per cell: hits + decay, and the per-scan scratch is cleared for the next scan
__device__ static inline void k3_apply_cell(int i, int8_t *__restrict__ map, int n_cells, uint32_t *__restrict__ hits,
                                            uint8_t *__restrict__ nohit, int max_hits)
{
    if (i >= n_cells) return;
    const uint32_t h = hits[i];
    const uint8_t nh = nohit[i];
    if (h == 0 && nh == 0) return;
    k3_apply_loaded(i, h, nh, map[i], map, hits, nohit, max_hits);
}
