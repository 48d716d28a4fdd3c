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

// the same update as extra workgroups of the two HoleMap launches (holemap.hip)
void cs_obstacle_ride(slamhip_cs *cs, const float *d_pose, float4 h_pxcs, int max_hits, k3_ride *r)
{
    memset(r, 0, sizeof(*r));
    if (cs->n_points <= 0) return;
    r->pts = cs->d_pts; r->n_points = cs->n_points; r->size = cs->os; r->scale = cs->oscale; r->d_pose = d_pose; r->h_pxcs = h_pxcs;
    r->hits = cs->d_o_hits; r->nohit = cs->d_o_nohit; r->chunks_per_ray = sh_div_up(cs->os + 1, 64);
    r->map = cs->d_obst; r->n_cells = cs->os * cs->os; r->max_hits = max_hits;
    r->n_blocks = 1;                                               // (the launches size their extra workgroups from the fields above)
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
    SH_TRY(cs_flush_scan(cs));
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
