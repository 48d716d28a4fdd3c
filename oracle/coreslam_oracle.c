/*
 * coreslam_oracle.c -- plain-C restatement of the CoreSLAM hot path (TEST INFRASTRUCTURE ONLY).
 * PARITY UNPINNED (see oracle.h): the reference has no tests / golden vectors and cannot run here.
 *
 * Follows /root/reference/CoreSLAM/CoreSLAMProcessor.cs line by line; every function cites the
 * lines it restates.  Arithmetic conventions carried over from C# on x64 .NET 6:
 *   - all float expressions are binary32 with one rounding per operation, evaluated left to
 *     right, never contracted to FMA (compile with -ffp-contract=off);
 *   - (int)float truncates toward zero; NaN / out-of-range gives INT_MIN (cvttss2si);
 *   - int arithmetic is unchecked (wraps); '/' truncates toward zero;
 *   - (ushort) casts truncate modulo 65536; '>>' on int is arithmetic.
 * Documented deviations (reference behaviour there is an exception or platform-dependent):
 *   D1  a scan point whose extended/hit pixel coordinates are not representable (NaN/inf, e.g.
 *       a zero-range point: dist = 0 -> add = inf -> NaN, CoreSLAMProcessor.cs:524-530) is
 *       skipped in the HoleMap update instead of drawing the x64-specific garbage line;
 *   D2  Math.Abs(int.MinValue) / int.MinValue / -1 (OverflowException in C#) skip the ray;
 *   D3  a blend whose ptr leaves the pixel array (IndexOutOfRangeException in C#) is skipped
 *       and counted in oracle_cs_oob_blends (unreachable once D4 holds; kept as a guard);
 *   D4  a ray whose CLIPPED endpoint is still outside the map -- reachable only through int32
 *       overflow of the products in ClipRay (:329,:340) for endpoints hundreds of metres away,
 *       where C# would walk off the array and throw -- is skipped.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

long long oracle_cs_oob_blends = 0;

/* ---- C# arithmetic helpers -------------------------------------------------------------- */
static inline int32_t f2i(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t isign(int32_t a) { return (a > 0) - (a < 0); }

/* ---- BaseSLAM/MathEx.cs ----------------------------------------------------------------- */
/* MathEx.cs:116-121 NormalizeAnglePos, :128-138 NormalizeAngle.  C# '%' on float == fmodf. */
float oracle_normalize_angle(float angle)
{
    const float pi = 3.14159274f;              /* MathF.PI */
    float pi2 = pi * 2.0f;
    float a = fmodf(fmodf(angle, pi2) + pi2, pi2);
    if (a > pi) a -= 2.0f * pi;
    return a;
}

/* MathEx.cs:69-73 DegDiff(float,float) */
float oracle_deg_diff(float a, float b)
{
    float d = ((a - b) + 180.0f) / 360.0f;
    return ((d - floorf(d)) * 360.0f) - 180.0f;
}

/* HoleMap.cs:20 / ObstacleMap.cs:20: Scale = sizePixels / sizeMeters (int -> float, fp32 divide) */
float oracle_map_scale(int size_pixels, float size_meters)
{
    return (float)size_pixels / size_meters;
}

/* CoreSLAMProcessor.cs:232-235 */
void oracle_cs_pose_to_pxcs(const float pose[3], float scale, float out[4])
{
    out[0] = pose[0] * scale + 0.5f;
    out[1] = pose[1] * scale + 0.5f;
    out[2] = oracle_cosf(pose[2]) * scale;
    out[3] = oracle_sinf(pose[2]) * scale;
}

/* CoreSLAMProcessor.cs:226-259 CalculateDistanceSISD (the live path, dispatcher :215-218) */
int32_t oracle_cs_distance_pxcs(const uint16_t *pixels, int size, const float *xy, int n_points,
                                const float pxcs[4])
{
    int nb_points = 0;                                           /* :228 */
    int64_t sum = 0;                                             /* :229 */
    const float px = pxcs[0], py = pxcs[1], c = pxcs[2], s = pxcs[3];
    for (int i = 0; i < n_points; i++) {                         /* :238 */
        const float X = xy[2 * i], Y = xy[2 * i + 1];
        float tx = px + c * X;  tx = tx - s * Y;                 /* :240 */
        float ty = py + s * X;  ty = ty + c * Y;                 /* :241 */
        int x = f2i(tx), y = f2i(ty);
        if (x >= 0 && x < size && y >= 0 && y < size) {          /* :244 */
            sum += pixels[(size_t)y * size + x];                 /* :246 */
            nb_points++;
        }
    }
    if (nb_points > 0) return (int32_t)((sum * 1024) / n_points); /* :253 divides by ALL points */
    return INT32_MAX;                                            /* :257 */
}

int32_t oracle_cs_distance(const uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                           const float pose[3])
{
    float pxcs[4];
    oracle_cs_pose_to_pxcs(pose, scale, pxcs);
    return oracle_cs_distance_pxcs(pixels, size, xy, n_points, pxcs);
}

int32_t oracle_cs_distance_batch_pxcs(const uint16_t *pixels, int size, const float *xy, int n_points,
                                      const float *pxcs, int K, int32_t *out_dist, int32_t *out_best_dist)
{
    int32_t best = -1, best_d = 0;
    for (int k = 0; k < K; k++) {
        int32_t d = oracle_cs_distance_pxcs(pixels, size, xy, n_points, pxcs + 4 * (size_t)k);
        if (out_dist) out_dist[k] = d;
        if (best < 0 || d < best_d) { best = k; best_d = d; }    /* strict '<' :644,:700 */
    }
    if (out_best_dist) *out_best_dist = best_d;
    return best;
}

/* CoreSLAMProcessor.cs:624-653 MonteCarloSearch + :695-705 cross-thread arg-min, flat list */
int32_t oracle_cs_search(const uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                         const float search_pose[3], const float *offs, int n_offs,
                         float out_pose[3], int32_t *out_dist, int32_t *out_all_dist)
{
    float best_pose[3] = { search_pose[0], search_pose[1], search_pose[2] };      /* :626 */
    int32_t best_d = oracle_cs_distance(pixels, size, scale, xy, n_points, search_pose); /* :627 */
    int32_t best_i = 0;
    if (out_all_dist) out_all_dist[0] = best_d;
    for (int k = 0; k < n_offs; k++) {                                            /* :630 */
        float cur[3];
        cur[0] = search_pose[0] + offs[3 * k + 0];                                /* :635 */
        cur[1] = search_pose[1] + offs[3 * k + 1];                                /* :636 */
        cur[2] = search_pose[2] + offs[3 * k + 2];                                /* :637 */
        int32_t d = oracle_cs_distance(pixels, size, scale, xy, n_points, cur);   /* :641 */
        if (out_all_dist) out_all_dist[k + 1] = d;
        if (d < best_d) {                                                         /* :644 */
            best_d = d; best_i = k + 1;
            best_pose[0] = cur[0]; best_pose[1] = cur[1]; best_pose[2] = cur[2];
        }
    }
    if (out_pose) { out_pose[0] = best_pose[0]; out_pose[1] = best_pose[1]; out_pose[2] = best_pose[2]; }
    if (out_dist) *out_dist = best_d;
    return best_i;
}

/* CoreSLAMProcessor.cs:320-345 ClipRay */
int oracle_cs_clip_ray(int size, int *xyc, int *yxc, int xy, int yx)
{
    if (*xyc < 0) {                                                               /* :322 */
        if (*xyc == xy) return 0;                                                 /* :324 */
        int32_t num = wmul(wsub(*yxc, yx), wsub(0, *xyc));                        /* :329 */
        int32_t den = wsub(*xyc, xy);
        if (den == -1 && num == INT32_MIN) return 0;                              /* D2 */
        *yxc = wadd(*yxc, num / den);
        *xyc = 0;                                                                 /* :330 */
    }
    if (*xyc >= size) {                                                           /* :333 */
        if (*xyc == xy) return 0;                                                 /* :335 */
        int32_t num = wmul(wsub(*yxc, yx), wsub(wsub(size, 1), *xyc));            /* :340 */
        int32_t den = wsub(*xyc, xy);
        if (den == -1 && num == INT32_MIN) return 0;                              /* D2 */
        *yxc = wadd(*yxc, num / den);
        *xyc = size - 1;                                                          /* :341 */
    }
    return 1;
}

/* CoreSLAMProcessor.cs:359-443 DrawLaserRayOnHoleMap */
int oracle_cs_draw_ray_holemap(uint16_t *pixels, int size, int x1, int y1, int x2, int y2,
                               int xp, int yp, int value, int alpha)
{
    int x2c = x2, y2c = y2;                                                       /* :361-362 */
    if (!oracle_cs_clip_ray(size, &x2c, &y2c, x1, y1)) return -1;                 /* :365 */
    if (!oracle_cs_clip_ray(size, &y2c, &x2c, y1, x1)) return -1;                 /* :366 */
    if (x2c < 0 || x2c >= size || y2c < 0 || y2c >= size) return -1;              /* D4 */

    int32_t ddx = wsub(x2, x1), ddy = wsub(y2, y1);
    int32_t ddxc = wsub(x2c, x1), ddyc = wsub(y2c, y1);
    if (ddx == INT32_MIN || ddy == INT32_MIN || ddxc == INT32_MIN || ddyc == INT32_MIN) return -1; /* D2 */
    int32_t dx = abs(ddx), dy = abs(ddy);                                         /* :368-369 */
    int32_t dxc = abs(ddxc), dyc = abs(ddyc);                                     /* :370-371 */
    int32_t incptrx = isign(ddx);                                                 /* :372 */
    int32_t incptry = wmul(isign(ddy), size);                                     /* :373 */
    int32_t sincv = isign(value - ORACLE_TS_NO_OBSTACLE);                         /* :374 */
    int32_t derrorv;

    if (dx > dy) {                                                                /* :377 */
        int32_t t = wsub(xp, x2);
        if (t == INT32_MIN) return -1;                                            /* D2 */
        derrorv = abs(t);                                                         /* :379 */
    } else {
        int32_t t;
        dx = dy;                                                                  /* :383 */
        t = dxc; dxc = dyc; dyc = t;                                              /* :384 */
        t = incptrx; incptrx = incptry; incptry = t;                              /* :385 */
        t = wsub(yp, y2);
        if (t == INT32_MIN) return -1;                                            /* D2 */
        derrorv = abs(t);                                                         /* :386 */
    }
    if (derrorv == 0) return -1;                                                  /* :389-392 */

    int32_t error = wsub(wmul(2, dyc), dxc);                                      /* :394 */
    int32_t horiz = wmul(2, dyc);                                                 /* :395 */
    int32_t diago = wmul(2, wsub(dyc, dxc));                                      /* :396 */
    int32_t errorv = derrorv / 2;                                                 /* :397 */
    int32_t incv = (value - ORACLE_TS_NO_OBSTACLE) / derrorv;                     /* :398 */
    int32_t incerrorv = wsub(value - ORACLE_TS_NO_OBSTACLE, wmul(derrorv, incv)); /* :399 */
    int32_t ptr = wadd(wmul(y1, size), x1);                                       /* :401 */
    int32_t pixval = ORACLE_TS_NO_OBSTACLE;                                       /* :402 */
    const int32_t lim2 = wsub(dx, wmul(2, derrorv));
    const int32_t lim1 = wsub(dx, derrorv);
    const int64_t npix = (int64_t)size * size;
    int blended = 0;

    for (int32_t x = 0; x <= dxc; x++, ptr = wadd(ptr, incptrx)) {                /* :404 */
        if (x > lim2) {                                                           /* :406 */
            if (x <= lim1) {                                                      /* :408 */
                pixval = wadd(pixval, incv);
                errorv = wadd(errorv, incerrorv);
                if (errorv > derrorv) { pixval = wadd(pixval, sincv); errorv = wsub(errorv, derrorv); }
            } else {
                pixval = wsub(pixval, incv);
                errorv = wsub(errorv, incerrorv);
                if (errorv < 0) { pixval = wsub(pixval, sincv); errorv = wadd(errorv, derrorv); }
            }
        }
        if (ptr >= 0 && ptr < npix) {
            int32_t v = wadd(wmul(256 - alpha, (int32_t)pixels[ptr]), wmul(alpha, pixval)) >> 8;  /* :431 */
            pixels[ptr] = (uint16_t)v;
            blended++;
        } else {
            oracle_cs_oob_blends++;                                               /* D3 */
        }
        if (error > 0) { ptr = wadd(ptr, incptry); error = wadd(error, diago); }  /* :433-437 */
        else           { error = wadd(error, horiz); }                            /* :440 */
    }
    return blended;
}

/* CoreSLAMProcessor.cs:496-534 UpdateHoleMap with (px,py,c,s) from :499-502 formed by the caller */
int64_t oracle_cs_update_holemap_pxcs(uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                                      const float pxcs[4], float hole_width, int quality)
{
    const float px = pxcs[0], py = pxcs[1], c = pxcs[2], s = pxcs[3];
    int x1 = f2i(px), y1 = f2i(py);                                               /* :505-506 */
    if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) return 0;                   /* :509-512 */
    int64_t total = 0;
    for (int i = 0; i < n_points; i++) {                                          /* :517 */
        const float X = xy[2 * i], Y = xy[2 * i + 1];
        float x2p = c * X - s * Y;                                                /* :519 */
        float y2p = s * X + c * Y;                                                /* :520 */
        int xp = f2i(px + x2p);                                                   /* :521 */
        int yp = f2i(py + y2p);                                                   /* :522 */
        float dist = sqrtf(x2p * x2p + y2p * y2p);                                /* :524 */
        float add = hole_width * scale / 2.0f / dist;                             /* :525 */
        x2p *= (1.0f + add);                                                      /* :527 */
        y2p *= (1.0f + add);                                                      /* :528 */
        int x2 = f2i(px + x2p);                                                   /* :529 */
        int y2 = f2i(py + y2p);                                                   /* :530 */
        if (xp == INT32_MIN || yp == INT32_MIN || x2 == INT32_MIN || y2 == INT32_MIN) continue; /* D1 */
        int n = oracle_cs_draw_ray_holemap(pixels, size, x1, y1, x2, y2, xp, yp,
                                           ORACLE_TS_OBSTACLE, quality);          /* :532 */
        if (n > 0) total += n;
    }
    return total;
}

int64_t oracle_cs_update_holemap(uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                                 const float pose[3], float hole_width, int quality)
{
    float pxcs[4];
    oracle_cs_pose_to_pxcs(pose, scale, pxcs);                                    /* :499-502 */
    return oracle_cs_update_holemap_pxcs(pixels, size, scale, xy, n_points, pxcs, hole_width, quality);
}

/* CoreSLAMProcessor.cs:456-490 DrawLaserRayOnObstacleMap; pixels is [y][x] row-major (ObstacleMap.cs:31) */
void oracle_cs_draw_ray_obstaclemap(int8_t *pixels, uint8_t *nohit, int size,
                                    int x1, int y1, int x2, int y2, int max_hits)
{
    int32_t ddx = wsub(x2, x1), ddy = wsub(y2, y1);
    if (ddx == INT32_MIN || ddy == INT32_MIN) return;                             /* D2 */
    int32_t dx = abs(ddx), sx = isign(ddx);                                       /* :458 */
    int32_t dy = abs(ddy), sy = isign(ddy);                                       /* :459 */
    int32_t err = (dx > dy ? dx : -dy) / 2, e2;                                   /* :460 */
    for (;;) {                                                                    /* :462 */
        if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) break;                  /* :465-469 */
        else if (x1 == x2 && y1 == y2) {                                          /* :471 */
            size_t i = (size_t)y1 * size + x1;
            if (pixels[i] < (int8_t)max_hits) pixels[i]++;                        /* :474-477 */
            break;
        } else {
            nohit[(size_t)y1 * size + x1] = 1;                                    /* :483 */
        }
        e2 = err;                                                                 /* :486 */
        if (e2 > -dx) { err = wsub(err, dy); x1 = wadd(x1, sx); }                 /* :487 */
        if (e2 < dy)  { err = wadd(err, dx); y1 = wadd(y1, sy); }                 /* :488 */
    }
}

/* CoreSLAMProcessor.cs:540-593 UpdateObstacleMap, (px,py,c,s) at ObstacleMap scale from :545-548 */
void oracle_cs_update_obstaclemap_pxcs(int8_t *pixels, uint8_t *nohit, int size,
                                       const float *xy, int n_points, const float pxcs[4], int max_hits)
{
    const float px = pxcs[0], py = pxcs[1], c = pxcs[2], s = pxcs[3];
    memset(nohit, 0, (size_t)size * size);                                        /* :542 */
    int x1 = f2i(px), y1 = f2i(py);                                               /* :553-554 */
    if (x1 < 0 || x1 >= size || y1 < 0 || y1 >= size) return;                     /* :557-560 */
    for (int i = 0; i < n_points; i++) {                                          /* :563 */
        const float X = xy[2 * i], Y = xy[2 * i + 1];
        float tx = px + c * X;  tx = tx - s * Y;                                  /* :566 */
        float ty = py + s * X;  ty = ty + c * Y;                                  /* :567 */
        oracle_cs_draw_ray_obstaclemap(pixels, nohit, size, x1, y1, f2i(tx), f2i(ty), max_hits); /* :570 */
    }
    for (int y = 0; y < size; y++)                                                /* :576 */
        for (int x = 0; x < size; x++) {
            size_t i = (size_t)y * size + x;
            if (nohit[i]) {                                                       /* :580 */
                if (pixels[i] < 0) pixels[i]++;                                   /* :582-585 */
                else if (pixels[i] > 0) pixels[i]--;                              /* :586-589 */
            }
        }
}

void oracle_cs_update_obstaclemap(int8_t *pixels, uint8_t *nohit, int size, float scale,
                                  const float *xy, int n_points, const float pose[3], int max_hits)
{
    float pxcs[4];
    oracle_cs_pose_to_pxcs(pose, scale, pxcs);                                    /* :545-548 */
    oracle_cs_update_obstaclemap_pxcs(pixels, nohit, size, xy, n_points, pxcs, max_hits);
}

/* CoreSLAMProcessor.cs:187-207 ScanSegmentsToCloud */
void oracle_cs_segments_to_cloud(const float *seg_poses, const int *seg_start, int n_seg,
                                 const float *rays, const float odo_pose[3], float *out_xy)
{
    for (int sgm = 0; sgm < n_seg; sgm++) {                                       /* :191 */
        float px = seg_poses[3 * sgm + 0] - odo_pose[0];                          /* :194 */
        float py = seg_poses[3 * sgm + 1] - odo_pose[1];
        float pz = seg_poses[3 * sgm + 2] - odo_pose[2];
        for (int r = seg_start[sgm]; r < seg_start[sgm + 1]; r++) {               /* :196 */
            float angle = rays[2 * r], radius = rays[2 * r + 1];
            out_xy[2 * r + 0] = px + radius * oracle_cosf(angle + pz);            /* :200 */
            out_xy[2 * r + 1] = py + radius * oracle_sinf(angle + pz);            /* :201 */
        }
    }
}

/* HoleMap.cs:44-55 GetPackedPixels */
void oracle_cs_pack_holemap(const uint16_t *pixels, int n_pixels, uint8_t *out)
{
    for (int i = 0; i < n_pixels / 2; i++)
        out[i] = (uint8_t)(((pixels[i * 2] >> 12) << 4) | (pixels[i * 2 + 1] >> 12)); /* :51 */
}

/* ---- full processor state machine -------------------------------------------------------- */
struct oracle_csproc {
    float physical; int hole_size, obst_size; float hole_scale, obst_scale;
    float start_pose[3], pose[3], last_odo[3];
    int scan_count;
    int quality; float hole_width; int search_beginning; int unmapped_hits; int max_hits;
    uint16_t *hole; int8_t *obst; uint8_t *nohit;
};

/* ctor CoreSLAMProcessor.cs:119-162 (sampler/worker parts not restated: offsets are inputs) */
oracle_csproc *oracle_csproc_create(float physical, int hole_size, int obst_size, const float start_pose[3])
{
    oracle_csproc *p = (oracle_csproc *)calloc(1, sizeof(*p));
    p->physical = physical; p->hole_size = hole_size; p->obst_size = obst_size;
    p->hole_scale = oracle_map_scale(hole_size, physical);
    p->obst_scale = oracle_map_scale(obst_size, physical);
    memcpy(p->start_pose, start_pose, sizeof(float) * 3);
    p->quality = 50; p->hole_width = 0.6f; p->search_beginning = 5;              /* :80,:85,:90 */
    p->unmapped_hits = -5; p->max_hits = 10;                                      /* :96,:101 */
    p->hole = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)hole_size * hole_size);
    p->obst = (int8_t *)malloc((size_t)obst_size * obst_size);
    p->nohit = (uint8_t *)malloc((size_t)obst_size * obst_size);
    oracle_csproc_reset(p);
    return p;
}

void oracle_csproc_destroy(oracle_csproc *p)
{
    if (!p) return;
    free(p->hole); free(p->obst); free(p->nohit); free(p);
}

/* Reset CoreSLAMProcessor.cs:167-175 */
void oracle_csproc_reset(oracle_csproc *p)
{
    size_t nh = (size_t)p->hole_size * p->hole_size;
    for (size_t i = 0; i < nh; i++) p->hole[i] = (uint16_t)((ORACLE_TS_OBSTACLE + ORACLE_TS_NO_OBSTACLE) / 2); /* :169 */
    memset(p->obst, (int8_t)p->unmapped_hits, (size_t)p->obst_size * p->obst_size);                             /* :170 */
    memcpy(p->pose, p->start_pose, sizeof(float) * 3);                            /* :172 */
    p->last_odo[0] = p->last_odo[1] = p->last_odo[2] = 0.0f;                      /* :173 */
    p->scan_count = 0;                                                            /* :174 */
}

void oracle_csproc_set_params(oracle_csproc *p, int quality, float hole_width, int search_beginning,
                              int unmapped_hits, int max_hits)
{
    p->quality = quality; p->hole_width = hole_width; p->search_beginning = search_beginning;
    p->unmapped_hits = unmapped_hits; p->max_hits = max_hits;
}

/* Update CoreSLAMProcessor.cs:717-752 */
void oracle_csproc_update(oracle_csproc *p, const float *seg_poses, const int *seg_start, int n_seg,
                          const float *rays, const float *offs, int n_offs)
{
    const float *odo = seg_poses + 3 * (n_seg - 1);                               /* :719 */
    float odo_pose[3] = { odo[0], odo[1], odo[2] };
    int n_points = seg_start[n_seg];
    float *xy = (float *)malloc(sizeof(float) * 2 * (size_t)(n_points > 0 ? n_points : 1));
    float new_pose[3];
    oracle_cs_segments_to_cloud(seg_poses, seg_start, n_seg, rays, odo_pose, xy); /* :723 */
    if (p->scan_count >= p->search_beginning) {                                   /* :726 */
        float search[3];
        for (int i = 0; i < 3; i++) search[i] = p->pose[i] + (odo_pose[i] - p->last_odo[i]); /* :728 */
        oracle_cs_search(p->hole, p->hole_size, p->hole_scale, xy, n_points, search, offs, n_offs,
                         new_pose, NULL, NULL);                                   /* :732 */
    } else {
        p->scan_count++;                                                          /* :741 */
        memcpy(new_pose, odo_pose, sizeof(new_pose));                             /* :742 */
    }
    memcpy(p->last_odo, odo_pose, sizeof(odo_pose));                              /* :745 */
    new_pose[2] = oracle_normalize_angle(new_pose[2]);                            /* :746 */
    memcpy(p->pose, new_pose, sizeof(new_pose));                                  /* :747 */
    oracle_cs_update_holemap(p->hole, p->hole_size, p->hole_scale, xy, n_points, p->pose,
                             p->hole_width, p->quality);                          /* :750 */
    oracle_cs_update_obstaclemap(p->obst, p->nohit, p->obst_size, p->obst_scale, xy, n_points,
                                 p->pose, p->max_hits);                           /* :751 */
    free(xy);
}

void oracle_csproc_get_pose(const oracle_csproc *p, float out[3]) { memcpy(out, p->pose, sizeof(float) * 3); }
uint16_t *oracle_csproc_holemap(oracle_csproc *p) { return p->hole; }
int8_t *oracle_csproc_obstaclemap(oracle_csproc *p) { return p->obst; }
