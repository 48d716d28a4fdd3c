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
#include "obstacle_dev.h"

__global__ void __launch_bounds__(256)
k3_rays(const float2 *__restrict__ pts, int n_points, int size, float scale, const float *d_pose, float4 h_pxcs,
        uint32_t *__restrict__ hits, uint8_t *__restrict__ nohit, int chunks_per_ray)
{
    k3_rays_unit(blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63, pts, n_points, size, scale, d_pose, h_pxcs, hits, nohit, chunks_per_ray);
}

__global__ void __launch_bounds__(256)
k3_apply(int8_t *__restrict__ map, int n_cells, uint32_t *__restrict__ hits, uint8_t *__restrict__ nohit, int max_hits)
{
    k3_apply_cell(blockIdx.x * blockDim.x + threadIdx.x, map, n_cells, hits, nohit, max_hits);
}

// the same update riding on the HoleMap update's launch (holemap.hip): this scan's ray walks into the current scratch set, the
// pending cell pass out of the other one
void cs_obstacle_ride(slamhip_cs *cs, const float *d_pose, float4 h_pxcs, int max_hits, k3_ride *r)
{
    memset(r, 0, sizeof(*r));
    if (cs->n_points <= 0) return;
    const int b = cs->obst_buf;
    r->pts = cs->d_pts; r->n_points = cs->n_points; r->size = cs->os; r->scale = cs->oscale; r->d_pose = d_pose; r->h_pxcs = h_pxcs;
    r->hits = cs->d_o_hits[b]; r->nohit = cs->d_o_nohit[b]; r->chunks_per_ray = sh_div_up(cs->os + 1, 64);
    r->on = 1;
    r->map = cs->d_obst;
    if (cs->obst_pending) {
        r->cell_hits = cs->d_o_hits[cs->obst_pend_buf]; r->cell_nohit = cs->d_o_nohit[cs->obst_pend_buf]; r->cell_max_hits = cs->obst_pend_max_hits;
        r->n_cells = cs->os * cs->os;
    }
    (void)max_hits;
}

void cs_obstacle_ride_commit(slamhip_cs *cs, const k3_ride *r, int max_hits)
{
    if (!r->on) return;
    // (the pending pass, if any, went with the launch; this scan's pass is pending now)
    cs->obst_pending = true; cs->obst_pend_buf = cs->obst_buf; cs->obst_pend_max_hits = max_hits;
    cs->obst_buf ^= 1;
}

int32_t cs_obstacle_flush(slamhip_cs *cs)
{
    if (!cs->obst_pending) return SLAMHIP_OK;
    const int cells = cs->os * cs->os;
    hipLaunchKernelGGL(k3_apply, dim3(sh_div_up(cells, 256)), dim3(256), 0, cs->ctx->stream,
                       cs->d_obst, cells, cs->d_o_hits[cs->obst_pend_buf], cs->d_o_nohit[cs->obst_pend_buf], cs->obst_pend_max_hits);
    SH_HIP(hipGetLastError());
    cs->obst_pending = false;
    return SLAMHIP_OK;
}

int32_t cs_obstacle_alloc(slamhip_cs *cs)
{
    const size_t n = (size_t)cs->os * cs->os;
    for (int b = 0; b < 2; b++) {
        SH_HIP(hipMalloc(&cs->d_o_hits[b], sizeof(uint32_t) * n));
        SH_HIP(hipMalloc(&cs->d_o_nohit[b], n));
        SH_HIP(hipMemsetAsync(cs->d_o_hits[b], 0, sizeof(uint32_t) * n, cs->ctx->stream));
        SH_HIP(hipMemsetAsync(cs->d_o_nohit[b], 0, n, cs->ctx->stream));
    }
    return SLAMHIP_OK;
}

void cs_obstacle_free(slamhip_cs *cs)
{
    for (int b = 0; b < 2; b++) { (void)hipFree(cs->d_o_hits[b]); (void)hipFree(cs->d_o_nohit[b]); }
}

int32_t cs_launch_obstacle_update(slamhip_cs *cs, const float *d_pose, float4 h_pxcs, int max_hits)
{
    slamhip_ctx *ctx = cs->ctx;
    const int n = cs->n_points;
    if (n <= 0) return SLAMHIP_OK;
    SH_TRY(cs_flush_scan(cs));
    SH_TRY(cs_obstacle_flush(cs));
    const int cells = cs->os * cs->os;
    sh_timer t(ctx, SLAMHIP_K_CS_OBSTACLE);
    // a walk stays in the map for at most `size` iterations (one major-axis step each): iterations 0 .. size
    const int chunks_per_ray = sh_div_up(cs->os + 1, 64);
    const int b = cs->obst_buf;
    hipLaunchKernelGGL(k3_rays, dim3(sh_div_up(n * chunks_per_ray, 4)), dim3(256), 0, ctx->stream,
                       cs->d_pts, n, cs->os, cs->oscale, d_pose, h_pxcs, cs->d_o_hits[b], cs->d_o_nohit[b], chunks_per_ray);
    hipLaunchKernelGGL(k3_apply, dim3(sh_div_up(cells, 256)), dim3(256), 0, ctx->stream,
                       cs->d_obst, cells, cs->d_o_hits[b], cs->d_o_nohit[b], max_hits);
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}
