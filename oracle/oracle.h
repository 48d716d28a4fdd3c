/*
 * oracle.h -- CPU restatement of the mikkleini/slam.net hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity oracle: a plain-C restatement of the reference's C# algorithm,
 * one function per reference unit, each citing the reference file:line it follows.  It is used
 * by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the
 * product path (slam.net_amd/), which fails loudly when its HIP library is missing.
 *
 * PARITY UNPINNED: the reference (C#/.NET 6) cannot be built or run in this image (no dotnet /
 * mono) and ships no tests, golden vectors or fixtures.  The oracle is
 * therefore pinned only by (1) the hand-derived known answers of SURVEY.md sec.4, (2) an
 * independently written NumPy restatement (oracle/np_oracle.py) that must agree bit-for-bit
 * on all integer outputs, and (3) the golden fixtures generated from them (tests/golden/).
 *
 * Third-party arithmetic that is NOT in /root/reference and is restated here from its
 * published behaviour (.NET 6 BCL, runtime unpinned beyond "net6.0"):
 *   - MathF.Cos/Sin/Sqrt/Exp/Log/Floor/Round  -> C libm cosf/sinf/sqrtf/expf/logf/floorf/rintf
 *     (MathF.Round(x) is banker's rounding = rintf in the default rounding mode).
 *     Trig can be switched to the deterministic correctly-rounded-float variant (oracle_trig_mode)
 *     that the HIP path uses for device-side candidate generation; see det_trig.c.
 *   - System.Numerics Matrix3x2 / Matrix4x4 / Vector2 / Vector3 ops -> restated in hector_oracle.c
 *   - (float -> int) casts follow x64 cvttss2si: NaN / out-of-range -> INT_MIN.
 *   - Redzen 9.0.0 ZigguratGaussianSampler is NOT restated: the reference seeds it from entropy,
 *     so candidate offsets are an explicit input everywhere.
 */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- trig selection ------------------------------------------------------------------- */
enum { ORACLE_TRIG_LIBM = 0, ORACLE_TRIG_DET = 1 };
void  oracle_set_trig_mode(int mode);
int   oracle_get_trig_mode(void);
float oracle_cosf(float a);
float oracle_sinf(float a);
/* deterministic sincos: double Cody-Waite reduction + Taylor, no FMA, rounded once to float */
void  oracle_det_sincosf(float a, float *s, float *c);

/* ---- BaseSLAM/MathEx.cs ---------------------------------------------------------------- */
float oracle_normalize_angle(float angle);                 /* MathEx.cs:116-138 */
float oracle_deg_diff(float a, float b);                   /* MathEx.cs:69-73   */

/* ---- CoreSLAM ------------------------------------------------------------------------- */
#define ORACLE_TS_NO_OBSTACLE 65500   /* CoreSLAMProcessor.cs:21 */
#define ORACLE_TS_OBSTACLE    0       /* CoreSLAMProcessor.cs:22 */

float oracle_map_scale(int size_pixels, float size_meters); /* HoleMap.cs:20, ObstacleMap.cs:20 */

/* CoreSLAMProcessor.cs:232-235: (px,py,c,s) from a pose */
void oracle_cs_pose_to_pxcs(const float pose[3], float scale, float out_pxcs[4]);

/* CoreSLAMProcessor.cs:226-259 CalculateDistanceSISD with (px,py,c,s) already formed */
int32_t oracle_cs_distance_pxcs(const uint16_t *pixels, int size,
                                const float *xy, int n_points, const float pxcs[4]);
/* CoreSLAMProcessor.cs:226-259 from a pose */
int32_t oracle_cs_distance(const uint16_t *pixels, int size, float scale,
                           const float *xy, int n_points, const float pose[3]);

/* batch: K candidates given as K x 4 (px,py,c,s); writes K distances; returns arg-min with the
 * reference tie-break (first strictly smaller wins, CoreSLAMProcessor.cs:644,700) */
int32_t oracle_cs_distance_batch_pxcs(const uint16_t *pixels, int size, const float *xy, int n_points,
                                      const float *pxcs, int K, int32_t *out_dist, int32_t *out_best_dist);

/* CoreSLAMProcessor.cs:624-653 + :695-705 with an explicit flat candidate list:
 * candidate 0 is the un-jittered searchPose, candidate k>=1 is searchPose + offs[k-1]
 * (offs is (K-1) x 3: dx, dy, dtheta; draw order X,Y,theta :635-637).  Returns best index. */
int32_t oracle_cs_search(const uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                         const float search_pose[3], const float *offs, int n_offs,
                         float out_pose[3], int32_t *out_dist, int32_t *out_all_dist /* n_offs+1 or NULL */);

/* CoreSLAMProcessor.cs:320-345 */
int oracle_cs_clip_ray(int size, int *xyc, int *yxc, int xy, int yx);
/* CoreSLAMProcessor.cs:359-443; returns number of blended pixels, -1 if skipped */
int oracle_cs_draw_ray_holemap(uint16_t *pixels, int size, int x1, int y1, int x2, int y2,
                               int xp, int yp, int value, int alpha);
/* CoreSLAMProcessor.cs:496-534; returns total blended pixels (for the roofline byte count) */
int64_t oracle_cs_update_holemap(uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                                 const float pose[3], float hole_width, int quality);
/* same with explicit (px,py,c,s) (trig left to the caller) */
int64_t oracle_cs_update_holemap_pxcs(uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                                      const float pxcs[4], float hole_width, int quality);

/* CoreSLAMProcessor.cs:456-490 */
void oracle_cs_draw_ray_obstaclemap(int8_t *pixels, uint8_t *nohit, int size,
                                    int x1, int y1, int x2, int y2, int max_hits);
/* CoreSLAMProcessor.cs:540-593 */
void oracle_cs_update_obstaclemap(int8_t *pixels, uint8_t *nohit_scratch, int size, float scale,
                                  const float *xy, int n_points, const float pose[3], int max_hits);
void oracle_cs_update_obstaclemap_pxcs(int8_t *pixels, uint8_t *nohit_scratch, int size,
                                       const float *xy, int n_points, const float pxcs[4], int max_hits);

/* CoreSLAMProcessor.cs:187-207; segs: n_seg poses (x,y,theta), seg_start[n_seg+1] ray ranges,
 * rays: (angle, radius) pairs.  out_xy must hold total rays x 2 floats. */
void oracle_cs_segments_to_cloud(const float *seg_poses, const int *seg_start, int n_seg,
                                 const float *rays, const float odo_pose[3], float *out_xy);

/* HoleMap.cs:44-55 */
void oracle_cs_pack_holemap(const uint16_t *pixels, int n_pixels, uint8_t *out_packed);

/* Full CoreSLAMProcessor state machine (ctor :119-162, Reset :167-175, Update :717-752) with
 * explicit per-scan candidate offsets instead of the Redzen sampler. */
typedef struct oracle_csproc oracle_csproc;
oracle_csproc *oracle_csproc_create(float physical_map_size, int hole_size, int obst_size,
                                    const float start_pose[3]);
void  oracle_csproc_destroy(oracle_csproc *p);
void  oracle_csproc_reset(oracle_csproc *p);
void  oracle_csproc_set_params(oracle_csproc *p, int quality, float hole_width, int search_beginning,
                               int unmapped_hits, int max_hits);
/* offs: n_offs x 3 jitter list used if this scan searches (may be NULL/0 -> base pose only) */
void  oracle_csproc_update(oracle_csproc *p, const float *seg_poses, const int *seg_start, int n_seg,
                           const float *rays, const float *offs, int n_offs);
void  oracle_csproc_get_pose(const oracle_csproc *p, float out[3]);
uint16_t *oracle_csproc_holemap(oracle_csproc *p);
int8_t   *oracle_csproc_obstaclemap(oracle_csproc *p);

/* ---- HectorSLAM ----------------------------------------------------------------------- */
typedef struct { int32_t update_index; float value; } oracle_cell;   /* LogOddsCell.cs:16-21 */

typedef struct oracle_grid oracle_grid;
oracle_grid *oracle_grid_create(float cell_len, int w, int h, float off_x, float off_y); /* GridMap.cs:33-51, OccGridMap.cs:35-48 */
void   oracle_grid_destroy(oracle_grid *g);
void   oracle_grid_reset(oracle_grid *g);                       /* OccGridMap.cs:244-252 */
void   oracle_grid_set_factors(oracle_grid *g, float free_f, float occ_f); /* OccGridMap.cs:58-79 */
void   oracle_grid_get_logodds(const oracle_grid *g, float *lo_free, float *lo_occ);
oracle_cell *oracle_grid_cells(oracle_grid *g);
int    oracle_grid_w(const oracle_grid *g);
int    oracle_grid_h(const oracle_grid *g);
float  oracle_grid_prob(oracle_grid *g, int index);             /* OccGridMap.cs:97-107 as a function of the current value (deviation D5) */
float  oracle_grid_prob_literal(oracle_grid *g, int index);     /* OccGridMap.cs:97-107 with the literal cache: stale across Reset (D5) */
void   oracle_grid_map_pose(const oracle_grid *g, const float world[3], float out[3]);   /* GridMap.cs:133-137 */
void   oracle_grid_world_pose(const oracle_grid *g, const float map[3], float out[3]);   /* GridMap.cs:122-126 */
/* OccGridMap.cs:114-148 (+:155-239) */
void   oracle_grid_update_by_scan(oracle_grid *g, const float *xy, int n_points,
                                  const float scan_origin[2], const float pose_world[3]);
void   oracle_grid_bitmap(const oracle_grid *g, uint8_t *out);  /* GridMap.cs:104-115 */
int    oracle_grid_map_extends(const oracle_grid *g, int out[4]);  /* GridMap.cs:147-207: {xMax,yMax,xMin,yMin}, returns found */

/* ScanMatcher.cs:211-249: out = (P, dPdx, dPdy) */
void   oracle_hs_interp(oracle_grid *g, float cx, float cy, float out[3]);
/* ScanMatcher.cs:135-204 with T worker chunks; H as 9 floats row-major 3x3, dTr 3 floats */
void   oracle_hs_hessian(oracle_grid *g, const float *xy, int n_points, const float pose_map[3],
                         int n_threads, float H[9], float dTr[3]);
/* ScanMatcher.cs:93-125; returns 1 if the estimate was updated */
int    oracle_hs_estimate_step(oracle_grid *g, const float *xy, int n_points, float estimate[3], int n_threads);
/* ScanMatcher.cs:64-84 */
void   oracle_hs_match_grid(oracle_grid *g, const float *xy, int n_points, const float hint_world[3],
                            int iterations, int n_threads, float out_world[3]);
/* ScanMatcher.cs:41-54 over a pyramid (levels[0] finest); iterations per level */
void   oracle_hs_match_pyramid(oracle_grid **levels, int n_levels, const float *xy, int n_points,
                               const float hint_world[3], const int *iterations, int n_threads,
                               float out_world[3]);

/* ---- CPU baseline with the ParallelWorker structure (BaseSLAM/ParallelWorker.cs:34-117) - */
/* T persistent threads, broadcast one action, wait for all, serial arg-min on the caller
 * (CoreSLAMProcessor.cs:674-710).  Runs n_scans searches of T*iters candidates each over the
 * same map/scan and returns wall seconds; out_evals = total distance evaluations done. */
double oracle_cpu_baseline_search(const uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                                  const float search_pose[3], const float *offs /* T*iters x 3 */,
                                  int n_threads, int iters_per_thread, int n_scans,
                                  int64_t *out_evals, int32_t *out_best_index, int32_t *out_best_dist);
double oracle_cpu_baseline_search_timed(const uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                                  const float search_pose[3], const float *offs /* T*iters x 3 */,
                                  int n_threads, int iters_per_thread, int n_scans,
                                  int64_t *out_evals, int32_t *out_best_index, int32_t *out_best_dist, double *scan_secs);

#ifdef __cplusplus
}
#endif
#endif
