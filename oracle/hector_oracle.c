/*
 * hector_oracle.c -- plain-C restatement of the HectorSLAM hot path (TEST INFRASTRUCTURE ONLY).
 * PARITY UNPINNED (see oracle.h).
 *
 * Follows /root/reference/HectorSLAM/{Matcher/ScanMatcher.cs, Map/OccGridMap.cs, Map/GridMap.cs,
 * Map/MapProperties.cs, Map/LogOddsCell.cs, Main/MapRepMultiMap.cs} and BaseSLAM/VectorEx.cs.
 *
 * The System.Numerics types the reference uses are NOT under /root/reference (they are the
 * .NET 6 BCL, runtime version unpinned).  Their arithmetic is restated below from the published
 * dotnet/runtime v6 scalar implementations (row-vector convention):
 *   Matrix3x2.CreateRotation   - IEEERemainder(theta, 2pi), snap to exact 0/90/180/270 deg within
 *                                 eps = 0.001*pi/180, else {c, s, -s, c, 0, 0}
 *   Matrix3x2.CreateTranslation, CreateScale(float), operator*, Invert (|det| < float.Epsilon fails)
 *   Vector2.Transform(v, M)    - (x*M11 + y*M21 + M31, x*M12 + y*M22 + M32)
 *   Matrix4x4.Invert           - cofactor expansion (software path), |det| < float.Epsilon fails
 *   Vector3.Transform(v, M44)  - row vector times matrix with w = 1
 * .NET's x64 SSE paths for Matrix4x4.Invert may differ from this in the last ulp, which is why
 * Hector parity is a tolerance (1e-4 m / 1e-4 rad), not bit-exactness.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

static inline int32_t f2i(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}

/* ---- System.Numerics restatement -------------------------------------------------------- */
typedef struct { float m11, m12, m21, m22, m31, m32; } m3x2;
static const float F_PI = 3.14159274f;
static const float F_EPSILON = 1.401298464e-45f;   /* float.Epsilon */

static m3x2 m3x2_rotation(float radians)
{
    m3x2 r;
    float c, s;
    const float epsilon = 0.001f * F_PI / 180.0f;
    radians = remainderf(radians, F_PI * 2);                 /* MathF.IEEERemainder */
    if (radians > -epsilon && radians < epsilon) { c = 1; s = 0; }
    else if (radians > F_PI / 2 - epsilon && radians < F_PI / 2 + epsilon) { c = 0; s = 1; }
    else if (radians < -F_PI + epsilon || radians > F_PI - epsilon) { c = -1; s = 0; }
    else if (radians > -F_PI / 2 - epsilon && radians < -F_PI / 2 + epsilon) { c = 0; s = -1; }
    else { c = oracle_cosf(radians); s = oracle_sinf(radians); }
    r.m11 = c; r.m12 = s; r.m21 = -s; r.m22 = c; r.m31 = 0; r.m32 = 0;
    return r;
}
static m3x2 m3x2_translation(float x, float y) { m3x2 r = { 1, 0, 0, 1, x, y }; return r; }
static m3x2 m3x2_scale(float s) { m3x2 r = { s, 0, 0, s, 0, 0 }; return r; }
static m3x2 m3x2_mul(m3x2 a, m3x2 b)
{
    m3x2 m;
    m.m11 = a.m11 * b.m11 + a.m12 * b.m21;
    m.m12 = a.m11 * b.m12 + a.m12 * b.m22;
    m.m21 = a.m21 * b.m11 + a.m22 * b.m21;
    m.m22 = a.m21 * b.m12 + a.m22 * b.m22;
    m.m31 = a.m31 * b.m11 + a.m32 * b.m21 + b.m31;
    m.m32 = a.m31 * b.m12 + a.m32 * b.m22 + b.m32;
    return m;
}
static int m3x2_invert(m3x2 m, m3x2 *r)
{
    float det = (m.m11 * m.m22) - (m.m21 * m.m12);
    if (fabsf(det) < F_EPSILON) return 0;
    float inv = 1.0f / det;
    r->m11 = m.m22 * inv;
    r->m12 = -m.m12 * inv;
    r->m21 = -m.m21 * inv;
    r->m22 = m.m11 * inv;
    r->m31 = (m.m21 * m.m32 - m.m31 * m.m22) * inv;
    r->m32 = (m.m31 * m.m12 - m.m11 * m.m32) * inv;
    return 1;
}
static void v2_transform(float x, float y, m3x2 m, float *ox, float *oy)
{
    *ox = x * m.m11 + y * m.m21 + m.m31;
    *oy = x * m.m12 + y * m.m22 + m.m32;
}

/* Matrix4x4.Invert, software path; M row-major m[r][c] */
static int m4_invert(const float M[4][4], float R[4][4])
{
    float a = M[0][0], b = M[0][1], c = M[0][2], d = M[0][3];
    float e = M[1][0], f = M[1][1], g = M[1][2], h = M[1][3];
    float i = M[2][0], j = M[2][1], k = M[2][2], l = M[2][3];
    float m = M[3][0], n = M[3][1], o = M[3][2], p = M[3][3];

    float kp_lo = k * p - l * o, jp_ln = j * p - l * n, jo_kn = j * o - k * n;
    float ip_lm = i * p - l * m, io_km = i * o - k * m, in_jm = i * n - j * m;

    float a11 = +(f * kp_lo - g * jp_ln + h * jo_kn);
    float a12 = -(e * kp_lo - g * ip_lm + h * io_km);
    float a13 = +(e * jp_ln - f * ip_lm + h * in_jm);
    float a14 = -(e * jo_kn - f * io_km + g * in_jm);

    float det = a * a11 + b * a12 + c * a13 + d * a14;
    if (fabsf(det) < F_EPSILON) return 0;
    float invDet = 1.0f / det;

    R[0][0] = a11 * invDet; R[1][0] = a12 * invDet; R[2][0] = a13 * invDet; R[3][0] = a14 * invDet;
    R[0][1] = -(b * kp_lo - c * jp_ln + d * jo_kn) * invDet;
    R[1][1] = +(a * kp_lo - c * ip_lm + d * io_km) * invDet;
    R[2][1] = -(a * jp_ln - b * ip_lm + d * in_jm) * invDet;
    R[3][1] = +(a * jo_kn - b * io_km + c * in_jm) * invDet;

    float gp_ho = g * p - h * o, fp_hn = f * p - h * n, fo_gn = f * o - g * n;
    float ep_hm = e * p - h * m, eo_gm = e * o - g * m, en_fm = e * n - f * m;
    R[0][2] = +(b * gp_ho - c * fp_hn + d * fo_gn) * invDet;
    R[1][2] = -(a * gp_ho - c * ep_hm + d * eo_gm) * invDet;
    R[2][2] = +(a * fp_hn - b * ep_hm + d * en_fm) * invDet;
    R[3][2] = -(a * fo_gn - b * eo_gm + c * en_fm) * invDet;

    float gl_hk = g * l - h * k, fl_hj = f * l - h * j, fk_gj = f * k - g * j;
    float el_hi = e * l - h * i, ek_gi = e * k - g * i, ej_fi = e * j - f * i;
    R[0][3] = -(b * gl_hk - c * fl_hj + d * fk_gj) * invDet;
    R[1][3] = +(a * gl_hk - c * el_hi + d * ek_gi) * invDet;
    R[2][3] = -(a * fl_hj - b * el_hi + d * ej_fi) * invDet;
    R[3][3] = +(a * fk_gj - b * ek_gi + c * ej_fi) * invDet;
    return 1;
}

/* ---- grid ------------------------------------------------------------------------------- */
struct oracle_grid {
    int w, h; float cell_len, off_x, off_y;
    oracle_cell *cells;                       /* GridMap.cs:13 mapArray */
    m3x2 map_t_world, world_t_map;            /* GridMap.cs:14-15 */
    int curr_update_index, curr_mark_occ, curr_mark_free;   /* OccGridMap.cs:20-22 */
    float odds_occ, odds_free, lo_occ, lo_free;             /* OccGridMap.cs:24-27 */
    /* the literal cache of OccGridMap.cs:16-19,38-42,97-107 -- kept ONLY for oracle_grid_prob_literal (deviation D5, below);
     * nothing else in the oracle reads it */
    float *cache_val; int *cache_idx; int curr_cache_index;
};

static float scale_to_map(const oracle_grid *g) { return 1.0f / g->cell_len; } /* MapProperties.cs:32 */

/* OccGridMap.cs:86-90 ProbToLogOdds */
static float prob_to_logodds(float prob) { float odds = prob / (1.0f - prob); return logf(odds); }

/* GridMap.cs:33-51 + OccGridMap.cs:35-48 */
oracle_grid *oracle_grid_create(float cell_len, int w, int h, float off_x, float off_y)
{
    oracle_grid *g = (oracle_grid *)calloc(1, sizeof(*g));
    g->w = w; g->h = h; g->cell_len = cell_len; g->off_x = off_x; g->off_y = off_y;
    g->cells = (oracle_cell *)malloc(sizeof(oracle_cell) * (size_t)w * h);
    g->cache_val = (float *)calloc((size_t)w * h, sizeof(float));
    g->cache_idx = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) g->cache_idx[i] = -1;   /* OccGridMap.cs:38-42: the ONLY place Index is set to -1 */
    g->curr_cache_index = 0;                                            /* :19 */
    g->map_t_world = m3x2_mul(m3x2_scale(scale_to_map(g)), m3x2_translation(off_x, off_y)); /* GridMap.cs:46 */
    if (!m3x2_invert(g->map_t_world, &g->world_t_map)) { free(g->cells); free(g->cache_val); free(g->cache_idx); free(g); return NULL; } /* :47-50 */
    g->odds_occ = 0.9f; g->odds_free = 0.4f;                 /* OccGridMap.cs:24-25 */
    g->lo_free = prob_to_logodds(g->odds_free);              /* :46 */
    g->lo_occ = prob_to_logodds(g->odds_occ);                /* :47 */
    oracle_grid_reset(g);
    return g;
}
void oracle_grid_destroy(oracle_grid *g) { if (g) { free(g->cells); free(g->cache_val); free(g->cache_idx); free(g); } }

/* GridMap.cs:56-62 + LogOddsCell.cs:38-42 + OccGridMap.cs:244-252 */
void oracle_grid_reset(oracle_grid *g)
{
    size_t n = (size_t)g->w * g->h;
    for (size_t i = 0; i < n; i++) { g->cells[i].value = 0.0f; g->cells[i].update_index = -1; }
    g->curr_update_index = 0; g->curr_mark_occ = -1; g->curr_mark_free = -1;
    g->curr_cache_index = 0;      /* :248 -- and cacheArray[i].Index is NOT touched (deviation D5) */
}
void oracle_grid_set_factors(oracle_grid *g, float free_f, float occ_f)   /* OccGridMap.cs:58-79 */
{
    g->odds_free = free_f; g->lo_free = prob_to_logodds(free_f);
    g->odds_occ = occ_f;   g->lo_occ = prob_to_logodds(occ_f);
}
void oracle_grid_get_logodds(const oracle_grid *g, float *lf, float *lo) { *lf = g->lo_free; *lo = g->lo_occ; }
oracle_cell *oracle_grid_cells(oracle_grid *g) { return g->cells; }
int oracle_grid_w(const oracle_grid *g) { return g->w; }
int oracle_grid_h(const oracle_grid *g) { return g->h; }

/* OccGridMap.cs:97-107 GetCachedProbability, as a function of the cell's CURRENT value.
 *
 * DEVIATION D5 (documented; oracle, library and tests agree on it).  The reference caches this expression per cell and
 * cache epoch: `currCacheIndex` is incremented by every UpdateByScan (:147), so between two Resets the cached value always
 * equals the expression (the cache is value-transparent).  It is NOT transparent across `Reset()`: :244-252 sets
 * currCacheIndex back to 0 but leaves every cacheArray[i].Index as it was (only the constructor writes -1, :38-42), so a
 * cell that was cached in epoch e before the Reset is served its PRE-RESET probability when it is queried in epoch e after
 * the Reset -- until the epochs differ again.  That is a reference bug (a matcher reading probabilities of a map that no
 * longer exists), it depends on which cells an earlier match happened to touch, and it is not reproduced: this function,
 * the device's probability grids (hector.hip: refreshed by every writer of the log-odds grid, reset included) and
 * slamhip_hs_cell_prob return the current value's probability.  oracle_grid_prob_literal below restates the C# cache
 * literally so that tests/test_oracle_kat.py can SHOW the reference's behaviour next to the chosen one. */
float oracle_grid_prob(oracle_grid *g, int index)
{
    float odds = expf(g->cells[index].value);               /* :101 */
    return odds / (odds + 1.0f);                            /* :102 */
}
/* OccGridMap.cs:97-107 literally, cache and all (D5: used by no product path and no other oracle function) */
float oracle_grid_prob_literal(oracle_grid *g, int index)
{
    if (g->cache_idx[index] != g->curr_cache_index) {       /* :99 */
        float odds = expf(g->cells[index].value);           /* :101 */
        g->cache_val[index] = odds / (odds + 1.0f);         /* :102 */
        g->cache_idx[index] = g->curr_cache_index;          /* :103 */
    }
    return g->cache_val[index];                             /* :106 */
}

/* GridMap.cs:133-137 */
void oracle_grid_map_pose(const oracle_grid *g, const float world[3], float out[3])
{
    v2_transform(world[0], world[1], g->map_t_world, &out[0], &out[1]);
    out[2] = world[2];
}
/* GridMap.cs:122-126 */
void oracle_grid_world_pose(const oracle_grid *g, const float map[3], float out[3])
{
    v2_transform(map[0], map[1], g->world_t_map, &out[0], &out[1]);
    out[2] = map[2];
}

/* OccGridMap.cs:192-199 */
static void cell_free(oracle_grid *g, int index)
{
    oracle_cell *c = &g->cells[index];
    if (c->update_index < g->curr_mark_free) { c->value += g->lo_free; c->update_index = g->curr_mark_free; }
}
/* OccGridMap.cs:201-218 */
static void cell_occ(oracle_grid *g, int index)
{
    oracle_cell *c = &g->cells[index];
    if (c->update_index < g->curr_mark_occ) {
        if (c->update_index == g->curr_mark_free) c->value -= g->lo_free;   /* :206-209 */
        if (c->value < 50.0f) c->value += g->lo_occ;                        /* :211-214 */
        c->update_index = g->curr_mark_occ;                                 /* :216 */
    }
}
/* OccGridMap.cs:220-239 */
static void bresenham2d(oracle_grid *g, int abs_da, int abs_db, int error_b, int offset_a, int offset_b, int offset)
{
    cell_free(g, offset);                                                   /* :222 */
    int end = abs_da - 1;                                                   /* :224 */
    for (int i = 0; i < end; ++i) {                                         /* :226 */
        offset += offset_a;
        error_b += abs_db;
        if (error_b >= abs_da) { offset += offset_b; error_b -= abs_da; }   /* :231-235 */
        cell_free(g, offset);                                               /* :237 */
    }
}
static int in_dims(const oracle_grid *g, int x, int y) { return x >= 0 && y >= 0 && x < g->w && y < g->h; } /* MapProperties.cs:94-97 */

/* OccGridMap.cs:155-190 */
static void update_line(oracle_grid *g, int bx, int by, int ex, int ey)
{
    if (!in_dims(g, bx, by) || !in_dims(g, ex, ey)) return;                 /* :158-161 */
    int dx = ex - bx, dy = ey - by;
    int abs_dx = abs(dx), abs_dy = abs(dy);
    int offset_dx = (dx > 0) - (dx < 0);
    int offset_dy = ((dy > 0) - (dy < 0)) * g->w;
    int start = by * g->w + bx;
    if (abs_dx >= abs_dy) bresenham2d(g, abs_dx, abs_dy, abs_dx / 2, offset_dx, offset_dy, start);   /* :175-179 */
    else                  bresenham2d(g, abs_dy, abs_dx, abs_dy / 2, offset_dy, offset_dx, start);   /* :180-185 */
    cell_occ(g, ey * g->w + ex);                                            /* :187-189 */
}

/* OccGridMap.cs:114-148 UpdateByScan */
void oracle_grid_update_by_scan(oracle_grid *g, const float *xy, int n_points,
                                const float scan_origin[2], const float pose[3])
{
    g->curr_mark_free = g->curr_update_index + 1;                           /* :116 */
    g->curr_mark_occ = g->curr_update_index + 2;                            /* :117 */
    m3x2 t = m3x2_mul(m3x2_mul(m3x2_rotation(pose[2]), m3x2_translation(pose[0], pose[1])),
                      m3x2_scale(scale_to_map(g)));                         /* :120-123 */
    float bxf, byf;
    v2_transform(scan_origin[0], scan_origin[1], t, &bxf, &byf);            /* :126 */
    int bx = f2i(rintf(bxf)), by = f2i(rintf(byf));                         /* :127 ToRoundPoint, VectorEx.cs:183-186 */
    for (int i = 0; i < n_points; i++) {                                    /* :130 */
        float exf, eyf;
        v2_transform(xy[2 * i], xy[2 * i + 1], t, &exf, &eyf);              /* :133 */
        int ex = f2i(rintf(exf)), ey = f2i(rintf(eyf));                     /* :134 */
        if (bx != ex || by != ey) update_line(g, bx, by, ex, ey);           /* :137-140 */
    }
    g->curr_update_index += 3;                                              /* :144 */
    g->curr_cache_index++;                                                  /* :147 */
}

/* GridMap.cs:104-115 */
void oracle_grid_bitmap(const oracle_grid *g, uint8_t *out)
{
    size_t n = (size_t)g->w * g->h;
    for (size_t i = 0; i < n; i++) {
        float v = g->cells[i].value;
        int sgn = (v > 0.0f) - (v < 0.0f);
        out[i] = (uint8_t)(127 - sgn * 127);                                /* :111 */
    }
}

/* GridMap.cs:147-207 GetMapExtends: out = {xMax, yMax, xMin, yMin}; the minima start at 10000 (:150), so on maps wider
 * than that a populated region entirely beyond column / row 10000 reports "nothing found" exactly as the reference does */
int oracle_grid_map_extends(const oracle_grid *g, int out[4])
{
    const int lower_start = -1, upper_start = 10000;                        /* :149-150 */
    int xmax = lower_start, ymax = lower_start, xmin = upper_start, ymin = upper_start;
    for (int x = 0; x < g->w; ++x)                                          /* :157-184 */
        for (int y = 0; y < g->h; ++y)
            if (g->cells[(size_t)y * g->w + x].value != 0.0f) {
                if (x > xmax) xmax = x;
                if (x < xmin) xmin = x;
                if (y > ymax) ymax = y;
                if (y < ymin) ymin = y;
            }
    if (xmax != lower_start && ymax != lower_start && xmin != upper_start && ymin != upper_start) {   /* :186-197 */
        out[0] = xmax; out[1] = ymax; out[2] = xmin; out[3] = ymin;
        return 1;
    }
    out[0] = out[1] = out[2] = out[3] = 0;                                  /* :199-205 */
    return 0;
}

/* ---- matcher ---------------------------------------------------------------------------- */
/* ScanMatcher.cs:211-249 InterpMapValueWithDerivatives */
void oracle_hs_interp(oracle_grid *g, float cx, float cy, float out[3])
{
    float limx = g->w - 2.0f, limy = g->h - 2.0f;                           /* MapProperties.cs:42 */
    if (isnan(cx) || isnan(cy) || cx < 0.0f || cx > limx || cy < 0.0f || cy > limy) { /* MapProperties.cs:83-87 */
        out[0] = out[1] = out[2] = 0.0f;                                    /* :216-219 */
        return;
    }
    int ix = f2i(floorf(cx)), iy = f2i(floorf(cy));                         /* :222 ToFloorPoint, VectorEx.cs:172-175 */
    float fx = cx - (float)ix, fy = cy - (float)iy;                         /* :225 */
    int sizeX = g->w;
    int index = iy * sizeX + ix;                                            /* :227 */
    float i0 = oracle_grid_prob(g, index);                                  /* :230 */
    float i1 = oracle_grid_prob(g, index + 1);                              /* :231 */
    float i2 = oracle_grid_prob(g, index + sizeX);                          /* :232 */
    float i3 = oracle_grid_prob(g, index + sizeX + 1);                      /* :233 */
    float dx1 = i0 - i1, dx2 = i2 - i3;                                     /* :235-236 */
    float dy1 = i0 - i2, dy2 = i1 - i3;                                     /* :238-239 */
    float xFacInv = 1.0f - fx, yFacInv = 1.0f - fy;                         /* :241-242 */
    out[0] = ((i0 * xFacInv + i1 * fx) * yFacInv) + ((i2 * xFacInv + i3 * fx) * fy);   /* :245-246 */
    out[1] = -((dx1 * xFacInv) + (dx2 * fx));                               /* :247 */
    out[2] = -((dy1 * yFacInv) + (dy2 * fy));                               /* :248 */
}

/* ScanMatcher.cs:135-204 GetCompleteHessianDerivs; H row-major 3x3 */
void oracle_hs_hessian(oracle_grid *g, const float *xy, int n_points, const float pose[3],
                       int n_threads, float H[9], float dTr[3])
{
    float cell = g->cell_len, stm = scale_to_map(g);
    m3x2 t = m3x2_mul(m3x2_mul(m3x2_rotation(pose[2]), m3x2_translation(pose[0] * cell, pose[1] * cell)),
                      m3x2_scale(stm));                                     /* :139-142 */
    float sinRot = oracle_sinf(pose[2]) * stm;                              /* :145 */
    float cosRot = oracle_cosf(pose[2]) * stm;                              /* :146 */
    if (n_threads < 1) n_threads = 1;
    int chunk = (n_points + n_threads - 1) / n_threads;                     /* :149 */
    float h11 = 0, h22 = 0, h33 = 0, h12 = 0, h13 = 0, h23 = 0, t0 = 0, t1 = 0, t2 = 0;   /* :188-189 */
    for (int th = 0; th < n_threads; th++) {                                /* thread order :191-195 */
        float l11 = 0, l22 = 0, l33 = 0, l12 = 0, l13 = 0, l23 = 0, lt0 = 0, lt1 = 0, lt2 = 0; /* :156-157 */
        int beg = th * chunk, end = beg + chunk;
        if (end > n_points) end = n_points;
        for (int i = beg; i < end; i++) {                                   /* :159 Skip/Take */
            float X = xy[2 * i], Y = xy[2 * i + 1], mx, my, d[3];
            v2_transform(X, Y, t, &mx, &my);                                /* :161 */
            oracle_hs_interp(g, mx, my, d);                                 /* :162 */
            float funVal = 1.0f - d[0];                                     /* :164 */
            lt0 += d[1] * funVal;                                           /* :166 */
            lt1 += d[2] * funVal;                                           /* :167 */
            float rotDeriv = ((-sinRot * X - cosRot * Y) * d[1] + (cosRot * X - sinRot * Y) * d[2]); /* :169-170 */
            lt2 += rotDeriv * funVal;                                       /* :172 */
            l11 += d[1] * d[1];                                             /* :174 */
            l22 += d[2] * d[2];                                             /* :175 */
            l33 += rotDeriv * rotDeriv;                                     /* :176 */
            l12 += d[1] * d[2];                                             /* :178 */
            l13 += d[1] * rotDeriv;                                         /* :179 */
            l23 += d[2] * rotDeriv;                                         /* :180 */
        }
        h11 += l11; h22 += l22; h33 += l33; h12 += l12; h13 += l13; h23 += l23;   /* :193 */
        t0 += lt0; t1 += lt1; t2 += lt2;                                    /* :194 */
    }
    H[0] = h11; H[1] = h12; H[2] = h13;
    H[3] = h12; H[4] = h22; H[5] = h23;                                     /* :198-200 symmetry */
    H[6] = h13; H[7] = h23; H[8] = h33;
    dTr[0] = t0; dTr[1] = t1; dTr[2] = t2;
}

/* ScanMatcher.cs:93-125 EstimateTransformationLogLh */
int oracle_hs_estimate_step(oracle_grid *g, const float *xy, int n_points, float estimate[3], int n_threads)
{
    float H[9], dTr[3];
    oracle_hs_hessian(g, xy, n_points, estimate, n_threads, H, dTr);        /* :95 */
    if (H[0] != 0.0f && H[4] != 0.0f) {                                     /* :97 */
        float M[4][4] = { { H[0], H[1], H[2], 0 }, { H[3], H[4], H[5], 0 }, { H[6], H[7], H[8], 0 },
                          { 0, 0, 0, 1.0f } };                              /* :203 M44 = 1 */
        float R[4][4];
        if (!m4_invert(M, R)) return 0;                                     /* :99-103 */
        float sd[3];                                                        /* :105 Vector3.Transform(dTr, iH) */
        sd[0] = (dTr[0] * R[0][0]) + (dTr[1] * R[1][0]) + (dTr[2] * R[2][0]) + R[3][0];
        sd[1] = (dTr[0] * R[0][1]) + (dTr[1] * R[1][1]) + (dTr[2] * R[2][1]) + R[3][1];
        sd[2] = (dTr[0] * R[0][2]) + (dTr[1] * R[1][2]) + (dTr[2] * R[2][2]) + R[3][2];
        if (sd[2] > 0.2f) sd[2] = 0.2f;                                     /* :107-111 */
        else if (sd[2] < -0.2f) sd[2] = -0.2f;                              /* :113-117 */
        estimate[0] += sd[0]; estimate[1] += sd[1]; estimate[2] += sd[2];   /* :119 */
        return 1;
    }
    return 0;                                                               /* :124 */
}

/* ScanMatcher.cs:64-84 MatchData(grid) */
void oracle_hs_match_grid(oracle_grid *g, const float *xy, int n_points, const float hint[3],
                          int iterations, int n_threads, float out[3])
{
    if (n_points > 0) {                                                     /* :66 */
        float est[3];
        oracle_grid_map_pose(g, hint, est);                                 /* :68 */
        for (int i = 0; i < iterations; i++)                                /* :70-73 */
            oracle_hs_estimate_step(g, xy, n_points, est, n_threads);
        est[2] = oracle_normalize_angle(est[2]);                            /* :76 */
        oracle_grid_world_pose(g, est, out);                                /* :79 */
        return;
    }
    out[0] = hint[0]; out[1] = hint[1]; out[2] = hint[2];                   /* :83 */
}

/* ScanMatcher.cs:41-54 MatchData(multiMap): coarsest level first */
void oracle_hs_match_pyramid(oracle_grid **levels, int n_levels, const float *xy, int n_points,
                             const float hint[3], const int *iterations, int n_threads, float out[3])
{
    float est[3] = { hint[0], hint[1], hint[2] };                           /* :43 */
    for (int idx = n_levels - 1; idx >= 0; idx--) {                         /* :47 */
        float next[3];
        oracle_hs_match_grid(levels[idx], xy, n_points, est, iterations[idx], n_threads, next); /* :49 */
        est[0] = next[0]; est[1] = next[1]; est[2] = next[2];
    }
    out[0] = est[0]; out[1] = est[1]; out[2] = est[2];
}
