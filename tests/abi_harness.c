/*
 * abi_harness.c -- a NON-Python caller of the C-ABI (TEST INFRASTRUCTURE; built and run by tests/test_abi_harness.py).
 *
 * What P/Invoke does, in C: dlopen("libslamhip.so"), resolve the entry points by NAME (the strings below are the
 * EntryPoint values of bindings/csharp/SlamHip/Native.cs), call them through plain function pointers with
 * caller-owned malloc'ed buffers, and compare with the golden vectors of tests/golden/ (written out as raw
 * little-endian arrays by the test: C has no .npz reader).  No header of the library is included on purpose --
 * the prototypes are restated here the way a foreign binding restates them, so a change of the ABI that the
 * header and the Python wrapper follow together still breaks THIS file.
 *
 *   abi_harness <libslamhip.so> <dir>
 *     <dir>/k1.meta  "size physical R K"             k1_pixels.u16 k1_xy.f32 k1_pxcs.f32 k1_base.f32 k1_offs.f32 k1_dist.i32
 *     <dir>/k2.meta  "size physical R scans hw q"     k2_xy.f32 k2_pxcs.f32 k2_after1.u16 k2_after_all.u16 k2_counts.i64
 *     <dir>/k3.meta  "obst_size physical R scans max_hits"  k3_xy.f32 k3_pxcs.f32 k3_after1.i8 k3_after_all.i8
 *     <dir>/k5.meta  "side cell_bits R scans"         k5_xy.f32 k5_poses.f32 k5_value.f32 k5_upd.i32
 *     <dir>/k4.meta  "R"   k4_xy.f32 k4_hint.f32 k4_pose.f32   (a match on the grid the k5 scans built; the expected pose within 1e-4)
 *   exit code 0 = every comparison bit-exact (the match: within its tolerance); otherwise the first failure is printed.
 *
 *   abi_harness <libslamhip.so> --group <dir>
 *     the multi-GPU entry points (slamhip_group_*) on a group of ONE GPU, from the k1 fixture: maps uploaded to the group, scan and
 *     jitters set on it, slamhip_group_search against the golden winner, slamhip_group_search_and_update against the same scan
 *     through a plain slamhip_cs (pose, distance, index and the HoleMap afterwards), slamhip_group_replicas_equal -- the path the
 *     driver's 8-GPU box runs first, driven by a caller that is not Python.
 *
 *   abi_harness <libslamhip.so> --bench-proc <hole_size> <rays> <candidates> <scans>
 *   abi_harness <libslamhip.so> --bench-hsproc <side> <levels> <rays> <scans>      HectorSLAMProcessor.Update, every scan updating the grids
 *     times slamhip_csproc_update (CoreSLAMProcessor.Update, CoreSLAMProcessor.cs:717-752) from a native caller: what a P/Invoke
 *     caller pays per scan, without an interpreter in the loop (bench.py's own figure goes through Python / ctypes).  The scans are
 *     the simulator's field (Simulation/Field.cs:45-71) seen from a slowly moving robot (ranges by ray / wall intersection, made here); prints one line
 *     "proc_us_per_scan <x>".  Timing only: parity of this path is tests/test_gpu_coreslam.py's business.
 *
 * Replaces nothing in the reference: it stands where Simulation/MainWindow.xaml.cs:69-72,145-146 would stand if it were a C
 * program ("all three callers go through the same extern "C" symbols").
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

typedef struct slamhip_ctx slamhip_ctx;
typedef struct slamhip_cs slamhip_cs;
typedef struct slamhip_hs slamhip_hs;
typedef struct { int32_t update_index; float value; } cell_t;          /* LogOddsCell.cs:16,21 sequential layout */

static const char *(*p_last_error)(void);
static int32_t (*p_ctx_create)(int32_t, slamhip_ctx **);
static int32_t (*p_ctx_destroy)(slamhip_ctx *);
static int32_t (*p_cs_create)(slamhip_ctx *, float, int32_t, int32_t, slamhip_cs **);
static int32_t (*p_cs_destroy)(slamhip_cs *);
static int32_t (*p_cs_info)(slamhip_cs *, int32_t *, float *, int32_t *, float *);
static int32_t (*p_cs_holemap_upload)(slamhip_cs *, const uint16_t *, size_t);
static int32_t (*p_cs_holemap_download)(slamhip_cs *, uint16_t *, size_t);
static int32_t (*p_cs_set_scan)(slamhip_cs *, const float *, int32_t);
static int32_t (*p_cs_distance_pxcs)(slamhip_cs *, const float *, int32_t, int32_t *, int32_t *, int32_t *);
static int32_t (*p_cs_set_offsets)(slamhip_cs *, const float *, int32_t);
static int32_t (*p_cs_search)(slamhip_cs *, const float *, float *, int32_t *, int32_t *);
static int32_t (*p_cs_update_holemap_pxcs)(slamhip_cs *, const float *, float, int32_t);
static int32_t (*p_cs_last_holemap_pixels)(slamhip_cs *, int64_t *);
static int32_t (*p_hs_create)(slamhip_ctx *, float, int32_t, int32_t, int32_t, slamhip_hs **);
static int32_t (*p_hs_destroy)(slamhip_hs *);
static int32_t (*p_hs_set_scan)(slamhip_hs *, const float *, int32_t, const float *);
static int32_t (*p_hs_update_by_scan)(slamhip_hs *, const float *);
static int32_t (*p_hs_cells_download)(slamhip_hs *, int32_t, cell_t *, size_t);

#define RESOLVE(var, name) do { *(void **)(&var) = dlsym(h, name); if (!var) { fprintf(stderr, "abi_harness: missing symbol %s\n", name); return 2; } } while (0)
#define CALL(expr) do { int32_t rc_ = (expr); if (rc_ != 0) { fprintf(stderr, "abi_harness: %s -> %d (%s)\n", #expr, (int)rc_, p_last_error()); return 3; } } while (0)
#define FAIL(...) do { fprintf(stderr, "abi_harness: " __VA_ARGS__); fprintf(stderr, "\n"); return 4; } while (0)

static void *slurp(const char *dir, const char *name, size_t want_bytes)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "abi_harness: cannot open %s\n", path); exit(5); }
    void *buf = malloc(want_bytes ? want_bytes : 1);
    size_t got = fread(buf, 1, want_bytes, f);
    int extra = fgetc(f);
    fclose(f);
    if (got != want_bytes || extra != EOF) { fprintf(stderr, "abi_harness: %s has the wrong size (want %zu bytes)\n", path, want_bytes); exit(5); }
    return buf;
}
static int read_meta(const char *dir, const char *name, double *v, int n)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    int k = 0;
    while (k < n && fscanf(f, "%lf", &v[k]) == 1) k++;
    fclose(f);
    return k == n;
}

/* K1: CalculateDistance over a candidate list given as (px,py,c,s) + the search over the jitter list (CoreSLAMProcessor.cs:226-259,:624-710) */
static int replay_k1(slamhip_ctx *ctx, const char *dir)
{
    double m[4];
    if (!read_meta(dir, "k1.meta", m, 4)) FAIL("k1.meta missing");
    const int size = (int)m[0], R = (int)m[2], K = (int)m[3];
    const float physical = (float)m[1];
    uint16_t *pixels = slurp(dir, "k1_pixels.u16", sizeof(uint16_t) * (size_t)size * size);
    float *xy = slurp(dir, "k1_xy.f32", sizeof(float) * 2 * (size_t)R);
    float *pxcs = slurp(dir, "k1_pxcs.f32", sizeof(float) * 4 * (size_t)K);
    float *base = slurp(dir, "k1_base.f32", sizeof(float) * 3);
    float *offs = slurp(dir, "k1_offs.f32", sizeof(float) * 3 * (size_t)(K - 1));
    int32_t *want = slurp(dir, "k1_dist.i32", sizeof(int32_t) * (size_t)K);
    slamhip_cs *cs = NULL;
    CALL(p_cs_create(ctx, physical, size, size / 4 > 0 ? size / 4 : 1, &cs));
    CALL(p_cs_holemap_upload(cs, pixels, (size_t)size * size));
    CALL(p_cs_set_scan(cs, xy, R));
    int32_t *dist = malloc(sizeof(int32_t) * (size_t)K), bi = -1, bd = -1;
    memset(dist, 0xAB, sizeof(int32_t) * (size_t)K);
    CALL(p_cs_distance_pxcs(cs, pxcs, K, dist, &bi, &bd));
    int wbi = 0;
    for (int k = 0; k < K; k++) {
        if (dist[k] != want[k]) FAIL("k1: distance %d is %d, golden %d", k, (int)dist[k], (int)want[k]);
        if (want[k] < want[wbi]) wbi = k;                        /* first strictly smaller wins (:644,:700) */
    }
    if (bi != wbi || bd != want[wbi]) FAIL("k1: arg-min (%d, %d), golden (%d, %d)", (int)bi, (int)bd, wbi, (int)want[wbi]);
    float pose[3] = { 0, 0, 0 };
    int32_t sd = -1, si = -1;
    CALL(p_cs_set_offsets(cs, offs, K - 1));
    CALL(p_cs_search(cs, base, pose, &sd, &si));
    if (si != wbi || sd != want[wbi]) FAIL("k1: search winner (%d, %d), golden (%d, %d)", (int)si, (int)sd, wbi, (int)want[wbi]);
    /* the winner's pose is search_pose + offs[index - 1] in binary32 (:635-637) */
    for (int c = 0; c < 3; c++) {
        const float w = wbi == 0 ? base[c] : base[c] + offs[3 * (wbi - 1) + c];
        if (memcmp(&w, &pose[c], 4) != 0) FAIL("k1: winner pose component %d is %.9g, expected %.9g", c, pose[c], w);
    }
    CALL(p_cs_destroy(cs));
    free(pixels); free(xy); free(pxcs); free(base); free(offs); free(want); free(dist);
    printf("k1 ok: %d candidates x %d rays on %d^2, winner %d distance %d\n", K, R, size, (int)si, (int)sd);
    return 0;
}

/* K2: UpdateHoleMap, ten scans in sequence (CoreSLAMProcessor.cs:496-534,:359-443) */
static int replay_k2(slamhip_ctx *ctx, const char *dir)
{
    double m[6];
    if (!read_meta(dir, "k2.meta", m, 6)) FAIL("k2.meta missing");
    const int size = (int)m[0], R = (int)m[2], scans = (int)m[3], q = (int)m[5];
    const float physical = (float)m[1], hw = (float)m[4];
    const size_t n = (size_t)size * size;
    float *xy = slurp(dir, "k2_xy.f32", sizeof(float) * 2 * (size_t)R * scans);
    float *pxcs = slurp(dir, "k2_pxcs.f32", sizeof(float) * 4 * (size_t)scans);
    uint16_t *after1 = slurp(dir, "k2_after1.u16", sizeof(uint16_t) * n);
    uint16_t *after_all = slurp(dir, "k2_after_all.u16", sizeof(uint16_t) * n);
    int64_t *counts = slurp(dir, "k2_counts.i64", sizeof(int64_t) * (size_t)scans);
    slamhip_cs *cs = NULL;
    CALL(p_cs_create(ctx, physical, size, size / 4 > 0 ? size / 4 : 1, &cs));
    uint16_t *got = malloc(sizeof(uint16_t) * n);
    for (int i = 0; i < scans; i++) {
        CALL(p_cs_set_scan(cs, xy + 2 * (size_t)R * i, R));
        CALL(p_cs_update_holemap_pxcs(cs, pxcs + 4 * i, hw, q));
        int64_t px = -1;
        CALL(p_cs_last_holemap_pixels(cs, &px));
        if (px != counts[i]) FAIL("k2: scan %d blended %lld pixels, golden %lld", i, (long long)px, (long long)counts[i]);
        if (i == 0) {
            CALL(p_cs_holemap_download(cs, got, n));
            if (memcmp(got, after1, sizeof(uint16_t) * n) != 0) FAIL("k2: HoleMap after the first scan differs from the golden image");
        }
    }
    CALL(p_cs_holemap_download(cs, got, n));
    for (size_t i = 0; i < n; i++)
        if (got[i] != after_all[i]) FAIL("k2: pixel %zu is %u, golden %u", i, (unsigned)got[i], (unsigned)after_all[i]);
    CALL(p_cs_destroy(cs));
    free(xy); free(pxcs); free(after1); free(after_all); free(counts); free(got);
    printf("k2 ok: %d scans x %d rays on %d^2\n", scans, R, size);
    return 0;
}

/* K5: OccGridMap.UpdateByScan, six scans (HectorSLAM/Map/OccGridMap.cs:114-239) */
static int replay_k5(slamhip_ctx *ctx, const char *dir)
{
    double m[4];
    if (!read_meta(dir, "k5.meta", m, 4)) FAIL("k5.meta missing");
    const int side = (int)m[0], R = (int)m[2], scans = (int)m[3];
    const uint32_t cell_bits = (uint32_t)m[1];
    float cell;
    memcpy(&cell, &cell_bits, 4);
    const size_t n = (size_t)side * side;
    float *xy = slurp(dir, "k5_xy.f32", sizeof(float) * 2 * (size_t)R * scans);
    float *poses = slurp(dir, "k5_poses.f32", sizeof(float) * 3 * (size_t)scans);
    float *value = slurp(dir, "k5_value.f32", sizeof(float) * n);
    int32_t *upd = slurp(dir, "k5_upd.i32", sizeof(int32_t) * n);
    slamhip_hs *hs = NULL;
    CALL(p_hs_create(ctx, cell, side, side, 1, &hs));
    const float origin[2] = { 0.0f, 0.0f };
    for (int i = 0; i < scans; i++) {
        CALL(p_hs_set_scan(hs, xy + 2 * (size_t)R * i, R, origin));
        CALL(p_hs_update_by_scan(hs, poses + 3 * i));
    }
    cell_t *cells = malloc(sizeof(cell_t) * n);
    CALL(p_hs_cells_download(hs, 0, cells, n));
    for (size_t i = 0; i < n; i++) {
        if (cells[i].update_index != upd[i]) FAIL("k5: cell %zu update index %d, golden %d", i, (int)cells[i].update_index, (int)upd[i]);
        if (memcmp(&cells[i].value, &value[i], 4) != 0) FAIL("k5: cell %zu value %.9g, golden %.9g", i, cells[i].value, value[i]);
    }
    CALL(p_hs_destroy(hs));
    free(xy); free(poses); free(value); free(upd); free(cells);
    printf("k5 ok: %d scans x %d rays on %d^2\n", scans, R, side);
    return 0;
}

/* K3: UpdateObstacleMap, ten scans in sequence (CoreSLAMProcessor.cs:540-593,:456-490) */
static int32_t (*p_cs_update_obstaclemap_pxcs)(slamhip_cs *, const float *, int32_t);
static int32_t (*p_cs_obstaclemap_download)(slamhip_cs *, int8_t *, size_t);
static int32_t (*p_cs_reset)(slamhip_cs *, int32_t);
static int replay_k3(slamhip_ctx *ctx, const char *dir)
{
    double m[5];
    if (!read_meta(dir, "k3.meta", m, 5)) FAIL("k3.meta missing");
    const int os = (int)m[0], R = (int)m[2], scans = (int)m[3], max_hits = (int)m[4];
    const float physical = (float)m[1];
    const size_t n = (size_t)os * os;
    float *xy = slurp(dir, "k3_xy.f32", sizeof(float) * 2 * (size_t)R * scans);
    float *pxcs = slurp(dir, "k3_pxcs.f32", sizeof(float) * 4 * (size_t)scans);
    int8_t *after1 = slurp(dir, "k3_after1.i8", n), *after_all = slurp(dir, "k3_after_all.i8", n);
    slamhip_cs *cs = NULL;
    CALL(p_cs_create(ctx, physical, 4 * os, os, &cs));
    CALL(p_cs_reset(cs, -5));                                     /* UnmappedObstacleHits (:96): the fixture's start value */
    int8_t *got = malloc(n);
    for (int i = 0; i < scans; i++) {
        CALL(p_cs_set_scan(cs, xy + 2 * (size_t)R * i, R));
        CALL(p_cs_update_obstaclemap_pxcs(cs, pxcs + 4 * i, max_hits));
        if (i == 0) {
            CALL(p_cs_obstaclemap_download(cs, got, n));
            if (memcmp(got, after1, n) != 0) FAIL("k3: ObstacleMap after the first scan differs from the golden image");
        }
    }
    CALL(p_cs_obstaclemap_download(cs, got, n));
    for (size_t i = 0; i < n; i++)
        if (got[i] != after_all[i]) FAIL("k3: cell %zu is %d, golden %d", i, (int)got[i], (int)after_all[i]);
    CALL(p_cs_destroy(cs));
    free(xy); free(pxcs); free(after1); free(after_all); free(got);
    printf("k3 ok: %d scans x %d rays on %d^2\n", scans, R, os);
    return 0;
}

/* K4: ScanMatcher.MatchData on the grid the K5 scans built (HectorSLAM/Matcher/ScanMatcher.cs:64-125); pose within 1e-4 m / rad */
static int32_t (*p_hs_match)(slamhip_hs *, const float *, float *);
static int replay_k4(slamhip_ctx *ctx, const char *dir)
{
    double m[4], m4[1];
    if (!read_meta(dir, "k5.meta", m, 4) || !read_meta(dir, "k4.meta", m4, 1)) FAIL("k4.meta / k5.meta missing");
    const int side = (int)m[0], R = (int)m[2], scans = (int)m[3], R4 = (int)m4[0];
    const uint32_t cell_bits = (uint32_t)m[1];
    float cell;
    memcpy(&cell, &cell_bits, 4);
    float *xy = slurp(dir, "k5_xy.f32", sizeof(float) * 2 * (size_t)R * scans);
    float *poses = slurp(dir, "k5_poses.f32", sizeof(float) * 3 * (size_t)scans);
    float *mxy = slurp(dir, "k4_xy.f32", sizeof(float) * 2 * (size_t)R4);
    float *hint = slurp(dir, "k4_hint.f32", sizeof(float) * 3), *want = slurp(dir, "k4_pose.f32", sizeof(float) * 3);
    slamhip_hs *hs = NULL;
    CALL(p_hs_create(ctx, cell, side, side, 1, &hs));
    const float origin[2] = { 0.0f, 0.0f };
    for (int i = 0; i < scans; i++) {
        CALL(p_hs_set_scan(hs, xy + 2 * (size_t)R * i, R, origin));
        CALL(p_hs_update_by_scan(hs, poses + 3 * i));
    }
    float pose[3] = { 0, 0, 0 };
    CALL(p_hs_set_scan(hs, mxy, R4, origin));
    CALL(p_hs_match(hs, hint, pose));
    for (int c = 0; c < 3; c++)
        if (!(fabsf(pose[c] - want[c]) <= 1.0e-4f)) FAIL("k4: matched pose component %d is %.7g, expected %.7g", c, pose[c], want[c]);
    CALL(p_hs_destroy(hs));
    free(xy); free(poses); free(mxy); free(hint); free(want);
    printf("k4 ok: match of %d rays on %d^2 -> %.5f %.5f %.5f\n", R4, side, pose[0], pose[1], pose[2]);
    return 0;
}

/* the multi-GPU entry points on a group of one GPU (the k1 fixture) */
typedef struct slamhip_group slamhip_group;
static int32_t (*p_group_create)(const int32_t *, int32_t, float, int32_t, int32_t, slamhip_group **);
static int32_t (*p_group_destroy)(slamhip_group *);
static int32_t (*p_group_size)(slamhip_group *, int32_t *);
static int32_t (*p_group_cs)(slamhip_group *, int32_t, slamhip_cs **);
static int32_t (*p_group_reset)(slamhip_group *, int32_t);
static int32_t (*p_group_holemap_upload)(slamhip_group *, const uint16_t *, size_t);
static int32_t (*p_group_set_scan)(slamhip_group *, const float *, int32_t);
static int32_t (*p_group_set_offsets)(slamhip_group *, const float *, int32_t);
static int32_t (*p_group_search)(slamhip_group *, const float *, float *, int32_t *, int32_t *);
static int32_t (*p_group_search_and_update)(slamhip_group *, const float *, float, int32_t, int32_t, float *, int32_t *, int32_t *);
static int32_t (*p_group_replicas_equal)(slamhip_group *, int32_t *);
static int32_t (*p_cs_search_and_update)(slamhip_cs *, const float *, float, int32_t, int32_t, float *, int32_t *, int32_t *);
static int run_group(void *h, const char *dir)
{
    RESOLVE(p_group_create, "slamhip_group_create"); RESOLVE(p_group_destroy, "slamhip_group_destroy"); RESOLVE(p_group_size, "slamhip_group_size");
    RESOLVE(p_group_cs, "slamhip_group_cs"); RESOLVE(p_group_reset, "slamhip_group_reset"); RESOLVE(p_group_holemap_upload, "slamhip_group_holemap_upload");
    RESOLVE(p_group_set_scan, "slamhip_group_set_scan"); RESOLVE(p_group_set_offsets, "slamhip_group_set_offsets"); RESOLVE(p_group_search, "slamhip_group_search");
    RESOLVE(p_group_search_and_update, "slamhip_group_search_and_update"); RESOLVE(p_group_replicas_equal, "slamhip_group_replicas_equal");
    RESOLVE(p_cs_search_and_update, "slamhip_cs_search_and_update"); RESOLVE(p_cs_reset, "slamhip_cs_reset");
    double m[4];
    if (!read_meta(dir, "k1.meta", m, 4)) FAIL("k1.meta missing");
    const int size = (int)m[0], R = (int)m[2], K = (int)m[3];
    const float physical = (float)m[1];
    const size_t n = (size_t)size * size;
    uint16_t *pixels = slurp(dir, "k1_pixels.u16", sizeof(uint16_t) * n);
    float *xy = slurp(dir, "k1_xy.f32", sizeof(float) * 2 * (size_t)R);
    float *base = slurp(dir, "k1_base.f32", sizeof(float) * 3);
    float *offs = slurp(dir, "k1_offs.f32", sizeof(float) * 3 * (size_t)(K - 1));
    int32_t *want = slurp(dir, "k1_dist.i32", sizeof(int32_t) * (size_t)K);
    int wbi = 0;
    for (int k = 0; k < K; k++) if (want[k] < want[wbi]) wbi = k;
    const int32_t dev0 = 0;
    slamhip_group *g = NULL;
    CALL(p_group_create(&dev0, 1, physical, size, size / 4 > 0 ? size / 4 : 1, &g));
    int32_t ng = 0;
    CALL(p_group_size(g, &ng));
    if (ng != 1) FAIL("group: size %d", (int)ng);
    CALL(p_group_reset(g, -5));
    CALL(p_group_holemap_upload(g, pixels, n));
    CALL(p_group_set_scan(g, xy, R));
    CALL(p_group_set_offsets(g, offs, K - 1));
    float pose[3] = { 0, 0, 0 }, fpose[3] = { 0, 0, 0 }, cpose[3] = { 0, 0, 0 };
    int32_t d = -1, i = -1;
    CALL(p_group_search(g, base, pose, &d, &i));                  /* block-sharded search + the exchange of the packed keys */
    if (i != wbi || d != want[wbi]) FAIL("group: search winner (%d, %d), golden (%d, %d)", (int)i, (int)d, wbi, (int)want[wbi]);
    CALL(p_group_search_and_update(g, base, 0.6f, 50, 10, fpose, &d, &i));   /* the whole scan on every GPU of the group */
    if (i != wbi || d != want[wbi]) FAIL("group: fused winner (%d, %d), golden (%d, %d)", (int)i, (int)d, wbi, (int)want[wbi]);
    /* the same scan through a plain handle on its own context */
    slamhip_ctx *ctx = NULL;
    slamhip_cs *cs = NULL, *cs0 = NULL;
    CALL(p_ctx_create(0, &ctx));
    CALL(p_cs_create(ctx, physical, size, size / 4 > 0 ? size / 4 : 1, &cs));
    CALL(p_cs_reset(cs, -5));
    CALL(p_cs_holemap_upload(cs, pixels, n));
    CALL(p_cs_set_scan(cs, xy, R));
    CALL(p_cs_set_offsets(cs, offs, K - 1));
    int32_t d2 = -1, i2 = -1;
    CALL(p_cs_search_and_update(cs, base, 0.6f, 50, 10, cpose, &d2, &i2));
    if (i2 != i || d2 != d || memcmp(cpose, fpose, sizeof cpose) != 0) FAIL("group: fused scan differs from the single handle's (%d %d | %d %d)", (int)i, (int)d, (int)i2, (int)d2);
    uint16_t *a = malloc(sizeof(uint16_t) * n), *b = malloc(sizeof(uint16_t) * n);
    CALL(p_group_cs(g, 0, &cs0));
    CALL(p_cs_holemap_download(cs0, a, n));
    CALL(p_cs_holemap_download(cs, b, n));
    if (memcmp(a, b, sizeof(uint16_t) * n) != 0) FAIL("group: the replica's HoleMap differs from the single handle's after the fused scan");
    if (memcmp(a, pixels, sizeof(uint16_t) * n) == 0) FAIL("group: the fused scan did not draw");
    int32_t eq = 0;
    CALL(p_group_replicas_equal(g, &eq));
    if (eq != 1) FAIL("group: replicas_equal says %d", (int)eq);
    CALL(p_cs_destroy(cs));
    CALL(p_ctx_destroy(ctx));
    CALL(p_group_destroy(g));
    free(pixels); free(xy); free(base); free(offs); free(want); free(a); free(b);
    printf("group ok: 1 GPU, winner %d distance %d, replica equals the single handle\n", (int)i, (int)d);
    return 0;
}

typedef struct slamhip_csproc slamhip_csproc;
static int32_t (*p_csproc_create)(slamhip_ctx *, float, int32_t, int32_t, const float *, float, float, int32_t, int32_t, slamhip_csproc **);
static int32_t (*p_csproc_destroy)(slamhip_csproc *);
static int32_t (*p_csproc_update)(slamhip_csproc *, const float *, const int32_t *, int32_t, const float *);
static int32_t (*p_csproc_get_pose)(slamhip_csproc *, float *);
static int32_t (*p_ctx_synchronize)(slamhip_ctx *);

/* range from (x, y) along angle a to the walls of the simulator's field -- the two closed polygons of Simulation/Field.cs:45-71 at
 * scale 30 m, offset (5, 5) in the 40 m world (Simulation/MainWindow.xaml.cs:97): the scene bench.py's headline scan is cast in
 * (SURVEY.md sec.8d), so that the search inside CoreSLAMProcessor.Update is comparable with the stand-alone one.  The robot stays
 * inside the outer polygon and outside the inner one: every ray hits. */
static const double FIELD_OUTER[12][2] = { {0.00, 0.0}, {1.00, 0.0}, {1.00, 0.2}, {0.80, 0.3}, {0.80, 0.5}, {1.00, 0.4},
                                           {1.00, 1.0}, {0.6, 1.0}, {0.6, 0.8}, {0.5, 0.8}, {0.5, 1.0}, {0.0, 1.0} };
static const double FIELD_INNER[4][2] = { {0.2, 0.3}, {0.3, 0.3}, {0.4, 0.7}, {0.3, 0.7} };
static double seg_hit(double x, double y, double c, double s, const double p0[2], const double p1[2])
{
    const double ax = 5.0 + 30.0 * p0[0], ay = 5.0 + 30.0 * p0[1], bx = 5.0 + 30.0 * p1[0], by = 5.0 + 30.0 * p1[1];
    const double ex = bx - ax, ey = by - ay, den = c * ey - s * ex;
    if (fabs(den) < 1e-12) return 1e9;
    const double t = ((ax - x) * ey - (ay - y) * ex) / den, u = ((ax - x) * s - (ay - y) * c) / den;
    return (t > 1e-9 && u >= 0.0 && u <= 1.0) ? t : 1e9;
}
static float room_range(double x, double y, double a)
{
    const double c = cos(a), s = sin(a);
    double t = 1e9;
    for (int i = 0; i < 12; i++) t = fmin(t, seg_hit(x, y, c, s, FIELD_OUTER[i], FIELD_OUTER[(i + 1) % 12]));
    for (int i = 0; i < 4; i++) t = fmin(t, seg_hit(x, y, c, s, FIELD_INNER[i], FIELD_INNER[(i + 1) % 4]));
    return (float)t;
}

static int bench_proc(void *h, int size, int R, int K, int scans)
{
    RESOLVE(p_csproc_create, "slamhip_csproc_create");
    RESOLVE(p_csproc_destroy, "slamhip_csproc_destroy");
    RESOLVE(p_csproc_update, "slamhip_csproc_update");
    RESOLVE(p_csproc_get_pose, "slamhip_csproc_get_pose");
    RESOLVE(p_ctx_synchronize, "slamhip_ctx_synchronize");
    slamhip_ctx *ctx = NULL;
    CALL(p_ctx_create(0, &ctx));
    const float start[3] = { 20.0f, 20.0f, 0.0f };
    slamhip_csproc *p = NULL;
    const int threads = 64, iters = (K - 1) / threads;          /* (K - 1) jitters + the base pose, as the C# constructor's T x iterations */
    CALL(p_csproc_create(ctx, 40.0f, size, size / 4, start, 0.1f, 0.17453292f, iters, threads, &p));
    const int n_distinct = 64, warm = 12;
    float *rays = malloc(sizeof(float) * 2 * (size_t)R * n_distinct), *poses = malloc(sizeof(float) * 3 * n_distinct);
    for (int k = 0; k < n_distinct; k++) {
        const double x = 20.0 + 0.04 * k, y = 20.0 + 0.015 * k, th = 0.004 * k;
        poses[3 * k] = (float)x; poses[3 * k + 1] = (float)y; poses[3 * k + 2] = (float)th;
        for (int i = 0; i < R; i++) {
            const double a = (double)i * 6.283185307179586 / R;
            rays[2 * ((size_t)k * R + i)] = (float)a;
            rays[2 * ((size_t)k * R + i) + 1] = room_range(x, y, a + th) + 0.002f * (float)((i * 7 + k * 3) % 11 - 5);
        }
    }
    const int32_t seg_start[2] = { 0, R };
    for (int k = 0; k < warm; k++) CALL(p_csproc_update(p, poses + 3 * k, seg_start, 1, rays + 2 * (size_t)k * R));
    CALL(p_ctx_synchronize(ctx));
    /* Three passes over the same scans: `scans` timed right away (a burst on a device at idle clocks), 5 x `scans` untimed (the
     * governor needs tens of milliseconds of sustained work), `scans` timed again -- the figure a scan loop that runs for seconds sees. */
    struct timespec t0, t1;
    double us[2] = { 0.0, 0.0 };
    int kk = 0;
    for (int pass = 0; pass < 3; pass++) {
        const int n = pass == 1 ? 5 * scans : scans;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int k = 0; k < n; k++, kk++) {
            const int j = warm + kk % (n_distinct - warm);
            CALL(p_csproc_update(p, poses + 3 * j, seg_start, 1, rays + 2 * (size_t)j * R));
        }
        if (pass != 1) CALL(p_ctx_synchronize(ctx));             /* (Update returns with the pose; the last scan's map updates belong to the figure) */
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (pass != 1) us[pass / 2] = ((double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3) / n;
    }
    float pose[3];
    CALL(p_csproc_get_pose(p, pose));
    printf("proc_us_per_scan %.3f  (sustained clocks; the first %d scans, from idle clocks: %.3f; %d^2 map, %d rays, %d candidates; last pose %.3f %.3f %.4f)\n",
           us[1], scans, us[0], size, R, threads * iters + 1, pose[0], pose[1], pose[2]);
    CALL(p_csproc_destroy(p));
    CALL(p_ctx_destroy(ctx));
    free(rays); free(poses);
    return 0;
}

/* HectorSLAMProcessor.Update (Main/HectorSLAMProcessor.cs:86-126) from a native caller: the same room, a hint = the last match,
 * thresholds at zero so that every scan updates the grids (the configuration of tools/exp.py hsproc ... 0). */
typedef struct slamhip_hsproc slamhip_hsproc;
static int32_t (*p_hsproc_create)(slamhip_ctx *, float, int32_t, int32_t, const float *, int32_t, slamhip_hsproc **);
static int32_t (*p_hsproc_destroy)(slamhip_hsproc *);
static int32_t (*p_hsproc_update)(slamhip_hsproc *, const float *, int32_t, const float *, const float *, int32_t, int32_t *);
static int32_t (*p_hsproc_get)(slamhip_hsproc *, float *, float *, float *, float *);
static int32_t (*p_hsproc_set_thresholds)(slamhip_hsproc *, float, float);
static int bench_hsproc(void *h, int side, int levels, int R, int scans)
{
    RESOLVE(p_hsproc_create, "slamhip_hsproc_create");
    RESOLVE(p_hsproc_destroy, "slamhip_hsproc_destroy");
    RESOLVE(p_hsproc_update, "slamhip_hsproc_update");
    RESOLVE(p_hsproc_get, "slamhip_hsproc_get");
    RESOLVE(p_hsproc_set_thresholds, "slamhip_hsproc_set_thresholds");
    RESOLVE(p_ctx_synchronize, "slamhip_ctx_synchronize");
    slamhip_ctx *ctx = NULL;
    CALL(p_ctx_create(0, &ctx));
    const float start[3] = { 20.0f, 20.0f, 0.0f }, origin[2] = { 0.0f, 0.0f };
    slamhip_hsproc *p = NULL;
    CALL(p_hsproc_create(ctx, 40.0f / (float)side, side, side, start, levels, &p));
    CALL(p_hsproc_set_thresholds(p, 0.0f, 0.0f));
    const int n_distinct = 64, warm = 12;
    float *xy = malloc(sizeof(float) * 2 * (size_t)R * n_distinct), *poses = malloc(sizeof(float) * 3 * n_distinct);
    for (int k = 0; k < n_distinct; k++) {
        const double x = 20.0 + 0.04 * k, y = 20.0 + 0.015 * k, th = 0.004 * k;
        poses[3 * k] = (float)x; poses[3 * k + 1] = (float)y; poses[3 * k + 2] = (float)th;
        for (int i = 0; i < R; i++) {                               /* the cloud in the robot frame (ScanCloud.Points) */
            const double a = (double)i * 6.283185307179586 / R;
            const double r = room_range(x, y, a + th) + 0.002 * (double)((i * 7 + k * 3) % 11 - 5);
            xy[2 * ((size_t)k * R + i)] = (float)(r * cos(a)); xy[2 * ((size_t)k * R + i) + 1] = (float)(r * sin(a));
        }
    }
    int32_t upd = 0, n_upd = 0;
    for (int k = 0; k < warm; k++) CALL(p_hsproc_update(p, xy + 2 * (size_t)k * R, R, origin, poses + 3 * k, 1, &upd));     /* (:179: the first loops map without matching) */
    CALL(p_ctx_synchronize(ctx));
    struct timespec t0, t1;
    double us[2] = { 0.0, 0.0 };
    float hint[3] = { poses[3 * (warm - 1)], poses[3 * (warm - 1) + 1], poses[3 * (warm - 1) + 2] };
    int kk = 0;
    for (int pass = 0; pass < 3; pass++) {
        const int n = pass == 1 ? 5 * scans : scans;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int k = 0; k < n; k++, kk++) {
            const int j = warm + kk % (n_distinct - warm);
            /* (the hint: the scan's true pose perturbed a little -- the lap jumps back every 52 scans, which a hint from the last match would not survive) */
            hint[0] = poses[3 * j] + 0.03f; hint[1] = poses[3 * j + 1] - 0.02f; hint[2] = poses[3 * j + 2] + 0.01f;
            CALL(p_hsproc_update(p, xy + 2 * (size_t)j * R, R, origin, hint, 0, &upd));
            n_upd += upd;
        }
        if (pass != 1) CALL(p_ctx_synchronize(ctx));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (pass != 1) us[pass / 2] = ((double)(t1.tv_sec - t0.tv_sec) * 1e6 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-3) / n;
    }
    float mp[3], lp[3], tm = 0, tu = 0;
    CALL(p_hsproc_get(p, mp, lp, &tm, &tu));
    printf("hsproc_us_per_scan %.3f  (sustained clocks; the first %d scans, from idle clocks: %.3f; %d^2 x %d levels, %d rays, %d of %d scans updated the grids; last match %.3f %.3f %.4f)\n",
           us[1], scans, us[0], side, levels, R, n_upd, 7 * scans, mp[0], mp[1], mp[2]);
    CALL(p_hsproc_destroy(p));
    CALL(p_ctx_destroy(ctx));
    free(xy); free(poses);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 7 && strcmp(argv[2], "--bench-proc") == 0) {
        void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "abi_harness: dlopen failed: %s\n", dlerror()); return 2; }
        RESOLVE(p_last_error, "slamhip_last_error");
        RESOLVE(p_ctx_create, "slamhip_ctx_create");
        RESOLVE(p_ctx_destroy, "slamhip_ctx_destroy");
        return bench_proc(h, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
    }
    if (argc == 7 && strcmp(argv[2], "--bench-hsproc") == 0) {
        void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "abi_harness: dlopen failed: %s\n", dlerror()); return 2; }
        RESOLVE(p_last_error, "slamhip_last_error");
        RESOLVE(p_ctx_create, "slamhip_ctx_create");
        RESOLVE(p_ctx_destroy, "slamhip_ctx_destroy");
        return bench_hsproc(h, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
    }
    if (argc == 4 && strcmp(argv[2], "--group") == 0) {
        void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "abi_harness: dlopen failed: %s\n", dlerror()); return 2; }
        RESOLVE(p_last_error, "slamhip_last_error");
        RESOLVE(p_ctx_create, "slamhip_ctx_create"); RESOLVE(p_ctx_destroy, "slamhip_ctx_destroy");
        RESOLVE(p_cs_create, "slamhip_cs_create"); RESOLVE(p_cs_destroy, "slamhip_cs_destroy");
        RESOLVE(p_cs_holemap_upload, "slamhip_cs_holemap_upload"); RESOLVE(p_cs_holemap_download, "slamhip_cs_holemap_download");
        RESOLVE(p_cs_set_scan, "slamhip_cs_set_scan"); RESOLVE(p_cs_set_offsets, "slamhip_cs_set_offsets");
        return run_group(h, argv[3]);
    }
    if (argc != 3) { fprintf(stderr, "usage: abi_harness <libslamhip.so> <fixture dir> | --group <fixture dir> | --bench-proc <size> <rays> <candidates> <scans>\n"); return 1; }
    void *h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "abi_harness: dlopen failed: %s\n", dlerror()); return 2; }
    RESOLVE(p_last_error, "slamhip_last_error");
    RESOLVE(p_ctx_create, "slamhip_ctx_create");
    RESOLVE(p_ctx_destroy, "slamhip_ctx_destroy");
    RESOLVE(p_cs_create, "slamhip_cs_create");
    RESOLVE(p_cs_destroy, "slamhip_cs_destroy");
    RESOLVE(p_cs_info, "slamhip_cs_info");
    RESOLVE(p_cs_holemap_upload, "slamhip_cs_holemap_upload");
    RESOLVE(p_cs_holemap_download, "slamhip_cs_holemap_download");
    RESOLVE(p_cs_set_scan, "slamhip_cs_set_scan");
    RESOLVE(p_cs_distance_pxcs, "slamhip_cs_distance_pxcs");
    RESOLVE(p_cs_set_offsets, "slamhip_cs_set_offsets");
    RESOLVE(p_cs_search, "slamhip_cs_search");
    RESOLVE(p_cs_update_holemap_pxcs, "slamhip_cs_update_holemap_pxcs");
    RESOLVE(p_cs_last_holemap_pixels, "slamhip_cs_last_holemap_pixels");
    RESOLVE(p_hs_create, "slamhip_hs_create");
    RESOLVE(p_hs_destroy, "slamhip_hs_destroy");
    RESOLVE(p_hs_set_scan, "slamhip_hs_set_scan");
    RESOLVE(p_hs_update_by_scan, "slamhip_hs_update_by_scan");
    RESOLVE(p_hs_cells_download, "slamhip_hs_cells_download");
    RESOLVE(p_cs_update_obstaclemap_pxcs, "slamhip_cs_update_obstaclemap_pxcs");
    RESOLVE(p_cs_obstaclemap_download, "slamhip_cs_obstaclemap_download");
    RESOLVE(p_cs_reset, "slamhip_cs_reset");
    RESOLVE(p_hs_match, "slamhip_hs_match");
    if (strcmp(argv[2], "--symbols-only") == 0) { printf("symbols ok\n"); return 0; }   /* (the CPU suite: no GPU, no compute) */
    slamhip_ctx *ctx = NULL;
    CALL(p_ctx_create(0, &ctx));
    int rc = replay_k1(ctx, argv[2]);
    if (rc == 0) rc = replay_k2(ctx, argv[2]);
    if (rc == 0) rc = replay_k3(ctx, argv[2]);
    if (rc == 0) rc = replay_k5(ctx, argv[2]);
    if (rc == 0) rc = replay_k4(ctx, argv[2]);
    CALL(p_ctx_destroy(ctx));
    if (rc == 0) printf("abi_harness: all golden replays bit-exact\n");
    return rc;
}
