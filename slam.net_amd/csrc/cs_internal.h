// cs_internal.h -- device state of the CoreSLAM operator object (slamhip_cs).
#pragma once
#include "common.h"
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <utility>

// Rays are grouped into blocks of at most CS_RB_MAX spatially close points (Z-order sorted, extent
// limited to CS_RB_EXTENT_PX map pixels); a workgroup of the distance kernel handles
// (1024 candidates) x (a chunk of ray blocks) and stages one HoleMap tile in LDS per ray block.
#define CS_RB_MAX 64
#define CS_RB_EXTENT_PX 128.0f         // upper limit; set_scan lowers it at fine map scales (coreslam.hip)
#define K1_GROUP 1024                  // theta-consecutive candidates per K1 workgroup ("group"): 512 lanes x 2 ...
#define K1_GROUP_BIG 2048              // ... or 512 lanes x 4 for large searches,
#define K1_RING_SLOTS 4                // result words of the enqueue-only search (valid until K1_RING_SLOTS - 1 further ring launches)
#define K1_GROUP_SMALL 512             // 512 lanes x 1 for small ones (slamhip_cs::k1_group, ensure_shard)
#define K1_PLAN_SLOTS 8                // plans of searches in flight (distance.hip: a slot is reused when the search that read it has finished)


// k1_make_layout (distance.hip): a ray block's terms of the cost estimate that do not depend on the candidate group
struct k1_block_term { double a, b, m, n, nr; };   // ex c + ey s, ex s + ey c, |mx| s + |my| c, |mx| c + |my| s, rays

struct slamhip_cs {
    slamhip_ctx *ctx;
    float physical;
    int hs; float hscale;         // HoleMap Size / Scale      (HoleMap.cs:19-20)
    int os; float oscale;         // ObstacleMap Size / Scale  (ObstacleMap.cs:19-20)
    uint16_t *d_hole;             // ushort[hs*hs] row-major   (HoleMap.cs:27)
    int8_t *d_obst;               // sbyte[os,os] [y,x]        (ObstacleMap.cs:31)

    // ---- scan -------------------------------------------------------------------------------
    int n_points, cap_points;
    void *d_scan_blob, *h_scan_blob;  // two device blocks used in turn / one pinned staging block for all per-scan uploads
    char *d_scan_cur; size_t scan_blob_bytes; int scan_buf;   // the block of the current scan
    // which launches read which block: a block may be refilled by the HOST (stores through the large BAR, set_scan) once the last
    // launch that read it is known to have finished -- launch_done is the number of the newest search launch whose result the host
    // has seen (everything before it in the stream has finished)
    uint64_t launch_count, launch_done, blob_use[2], k1_launch_no;
    hipStream_t side_stream; unsigned *d_side_arrive; uint32_t side_seq; bool side_join;
    // the next scan's candidate list, prepared ahead on the side stream (cs_speculate_next, coreslam.hip)
    float *spec_offs_flat, *spec_ev_off, *spec_grp_bounds; int *spec_ev_idx; int spec_cap_offs, spec_cap_cand, spec_cap_grp;
    float *cool_offs_flat, *cool_ev_off, *cool_grp_bounds; int *cool_ev_idx;   // the third set of the rotation: last scan's, not written yet (see cs_generate)
    uint64_t spec_hits, spec_made;
    bool spec_valid, spec_base_ok, spec_lattice; int spec_n, spec_grp; float spec_sxy, spec_sth; uint64_t spec_seed, spec_stream; // jitter generation + scan upload beside the previous scan's map updates (ensure_shard)
    hipEvent_t ev_scan; bool scan_in_flight;
    bool upload_pending; size_t upload_bytes;   // set_scan filled the staging block; the upload is launched by the first consumer (cs_flush_scan),
                                                // or rides on the candidate gather's launch when one comes first (ensure_shard)
    float2 *d_pts;                // original order: K2/K3 are ray-order dependent
    float2 *d_pts_sorted;         // spatially sorted copy for K1 (integer sum: any order is exact)
    int4 *d_ray_blk;              // per sorted ray: (first ray of its block, one past its last, block index, 0)
    std::vector<uint64_t> h_sort_keys, h_sort_tmp;   // set_scan's sort buffers
    std::vector<uint32_t> h_sort_k32, h_sort_order;   // ... the 32-bit form's, and the resulting order (ray index by sorted position)
    std::vector<int> h_cell_xy;                      // ... and the points' 64-pixel cell coordinates
    std::vector<int> h_rb_start;  // host copy of the block table
    std::vector<float> h_rb_ex, h_rb_ey, h_rb_mx, h_rb_my;   // per ray block, scan frame, pixels: bounding box size and centre
    int n_rb;                     // ray blocks over d_pts_sorted
    int *d_rb_start;              // [n_rb + 1]
    bool pts_sane;                // all |coords| < 1e9: fast kernels need no NaN/overflow handling

    // ---- candidates ---------------------------------------------------------------------------
    int n_offs;                   // jitters in the flat list; flat candidate count = n_offs + 1
    std::vector<float> h_offs;    // host copy (n_offs x 3) for shard sorting / pose_from_key
    bool offs_on_device_sorted;   // generated on the device: flat list already theta-sorted
    float *d_offs_flat;           // [n_offs x 3] flat order
    int cap_offs;                 // jitters d_offs_flat has room for
    bool gen_lattice; int k1_lattice;   // the generated list is a heading lattice (slamhip_cs_generate_offsets_lattice); candidates per lane the current shard's search may share products over (0: none)
    bool gen_pending; uint64_t gen_seed, gen_stream;   // device-generated list requested but not produced yet (see ensure_shard)
    int shard_first, shard_count; // evaluation list currently materialised
    float *d_ev_off;              // [cap_cand x 3] offsets in evaluation (theta-sorted) order
    int *d_ev_idx;                // [cap_cand] evaluation position -> flat index
    int cap_cand;
    float4 *d_pxcs;               // [cap_cand] (px,py,c,s) in evaluation order
    void *d_partial; size_t cap_partial;       // K1 partial rows (bytes)
    unsigned long long *d_k1_gmin;              // K1: [0] running minimum of the finished candidates' keys (all ones between launches), [1] their count (zero)
    unsigned long long *d_k1_acc;               // K1: per-candidate accumulators [groups][K1_GROUP] (zero between launches)
    int k1_cap_groups;                          // (in units of K1_GROUP candidates)
    int k1_group;                               // candidates per group of the materialised evaluation list (its jitter bounds are per group)
    // K1 launch layout: per group of 1024 evaluation-order candidates the theta range (rad) and translation spread
    // (pixels) -- from the offsets (ensure_shard) -- and the chunks per group derived from them and the scan
    std::vector<float> h_grp_dth, h_grp_dxy;
    std::vector<float> h_grp_lohi;              // per group: min, max of dx | dy | dtheta over its candidates' jitters (layout estimates only)
    std::vector<int> k1_tab_group, k1_tab_nc, k1_tab_nbp; int k1_uni_g0, k1_uni_ng, k1_uni_nc;
    std::vector<k1_block_term> k1_terms; std::vector<double> k1_cost, k1_lc; std::vector<int> k1_share;   // k1_make_layout's scratch (distance.hip)
    // ray ranges of the uniform part cut by COST (rays + a weight per ray block touched) instead of by count (k1_balanced_cuts):
    // cuts by count of ranges (cuts[0 .. n], n <= the count asked for; empty: none), for scan generation k1_cut_gen at weight k1_cut_w
    std::vector<std::pair<int, std::vector<int>>> k1_cut_cache; uint32_t k1_cut_gen, k1_cut_layout_gen;
    // ... and for the per-scan flow, where every launch sees a new scan: the uniform part's cut made for the PREVIOUS scan while the host
    // waited for its pose (cs_layout_idle_refresh), used for the next scan's launch if it is legal for that scan's ray blocks (a cut
    // only balances the launch: any legal one gives the same sums)
    std::vector<int> k1_prev_cuts; int k1_prev_cuts_nc; uint32_t k1_prev_cuts_layout_gen; int k1_prev_cuts_points;
    bool k1_launch_prev_cuts;                   // the search launch now in the stream took them (a prelaunched one: legality is tested when the tables exist)
    float k1_last_pose[3]; int k1_last_group; bool k1_last_valid;   // the last tiled mode-1 launch: what the idle refresh makes the next cut for
    uint32_t k1_cut_seen_scan, k1_cut_seen_layout;   // the (scan, layout) of the last tiled launch: cuts are made from the second launch of a pair on
    uint32_t k1_layout_gen;                     // layouts made so far (k1_make_layout)
    std::vector<char> k1_cut_cand;              // per ray block: its tile may exceed the budget (worth the exact box test)
    std::vector<double> k1_cut_wsc;             // (scratch)
    std::vector<double> k1_cut_wb;              // per ray block: the weight of one more tile step, in ray units (k1_cut_weights)
    uint32_t scan_gen;                          // scans set so far (slamhip_cs_set_scan)
    std::vector<float> h_grp_prev; int k1_group_prev;   // the groups' figures before the last ensure_shard (layout kept when unchanged)
    bool k1_layout_dirty, k1_layout_spread; int k1_layout_budget, k1_layout_groups; float k1_layout_theta;
    bool k1_scan_dirty;                         // a new scan since the layout was made (set_scan): it is kept if still legal, see cs_launch_distance
    bool k1_prelaunch;                          // the search launch now being made precedes its scan's tables (cs_search_and_update_prelaunched, coreslam.hip): the layout is the last scan's, unchecked
    uint64_t pl_stats[4];                       // slamhip_cs_prelaunch_stats: launched ahead of the tables | abandoned | a new layout was needed | refused (ordinary order)
    uint32_t *d_scan_flag; uint32_t scan_flag_seq;   // ... and waits on the device for this word: the host stores the scan's number there when the tables have landed (or the number | 2^31: abandon the launch)
    bool k1_layout_stale;                       // ... and the one for the scan now set is made in the host's next idle wait (cs_layout_idle_refresh)
    int k1_layout_target, k1_layout_band_parts;
    float gen_sigma_xy, gen_sigma_theta;        // offsets generated on the device: their distribution
    bool offs_theta_small;        // every |dtheta| <= 1e4: the tiled kernel's trigonometry needs no huge-angle branch
    unsigned int *d_verify;       // [8] SLAMHIP_K1_VERIFY=1: [0] tile self-check failures (must stay 0), [1..4] unit counts per kind
    int32_t *d_dist;              // [cap_cand] per-candidate distances in FLAT order (optional output)
    uint64_t *d_key;              // packed (dist << 32 | flat index) arg-min
    uint64_t *h_key;              // pinned
    float *d_grp_bounds; int cap_grp;          // per candidate group: min/max of px,py,c,s (8 floats)
    float *d_best_pose;           // winner's pose (theta normalised), device-resident for the fused path (inside d_key's block)
    bool k1_want_pose;            // the next search launch also writes d_best_pose (fused search + update)
    bool k1_pose_written;         // ... and it did (tiled kernel); the fallback kernels do not
    unsigned *k1_done_flag; unsigned k1_done_val;   // the next search launch ends with k1_done_val -> *k1_done_flag (pinned host word), if set
    bool k1_done_armed;           // ... and it will (tiled kernel)
    unsigned long long *k1_sig; unsigned long long k1_sig_val; bool k1_sig_armed;   // the same for an HSA signal (slamhip_comm: the collectives' stream waits for it)
    // enqueue-only searches (slamhip_cs_search_shard_enqueue): a ring of result words, all ones at rest; a ring launch mins into
    // slot k1_ring_pos and leaves slot k1_ring_pos + 1 all ones for the next ring launch (distance.hip)
    uint64_t *d_k1_ring; unsigned k1_ring_pos;
    bool k1_ring_request;         // the next search launch is a ring launch ...
    uint64_t *k1_ring_last;       // ... and this is the slot it used
    // the search's plan (distance.hip, k1_plan): K1_PLAN_SLOTS sets of buffers used in turn, filled on a stream of their own
    hipStream_t plan_stream;
    uint2 *d_plan_rec[K1_PLAN_SLOTS];
    int plan_cap_wgs;
    uint32_t plan_seq;            // stamps handed out (a stamp is never 0 and never reused while its slot holds it)
    uint32_t plan_count;          // plans launched (slot = plan_count % K1_PLAN_SLOTS)
    uint32_t k1_launches;         // tiled search launches so far: each stores its number into word 25 of h_key when it STARTS (k1_args::started)
    uint32_t plan_slot_user[K1_PLAN_SLOTS];   // the number of the search launch that read the slot last
    uint32_t plan_inputs_after;   // the plan kernel reads what launches in the operator's stream wrote (candidate gather, scan upload): it may only be
                                  // launched once the search launch of this number has started (0: nothing pending)
    uint64_t plan_stats[4];       // searches launched with a plan | without | host waits for a free slot | plans skipped because their inputs were still in flight
    uint32_t upload_seq;          // set_scan uploads issued (the upload's workgroups store it into words 28 .. 31 of h_key when they have read the staging block)

    // ---- K2 HoleMap update -----------------------------------------------------------------------------
    void *d_rays; int cap_rays;                 // rays by index (k2_byidx): clipped lengths, flags
    void *d_k2_cand;                            // rays as the pixel kernels test them, sorted by (direction class, slope bucket)
    void *d_k2_vprof;                           // V-profile parameters by ray index
    int *d_k2_start;                            // [4 x 1024 + 1] first table entry of every bucket
    int *d_k2_counters;           // [0] longest ray, [1] unused, [2] blended pixels, [3] x1, [4] y1
    void *mirror_reg; size_t mirror_reg_bytes;   // the caller's mirror array, page-locked on first use (slamhip_cs_holemap_mirror)
    // asynchronous host mirror (slamhip_cs_holemap_mirror_async): per-row column spans of what the updates since the last snapshot
    // may have changed (K2 keeps them while mirror_on), a shadow map the snapshot launch copies the spans into, the spans as
    // snapshotted, a copy stream on which a launch pushes the shadow's spans into the caller's (page-locked, device-mapped) array
    bool mirror_on, mirror_pending;
    int2 *d_hole_span, *d_hole_span_snap;       // [hs] (first column, last column); empty: (hs, -1)
    uint16_t *d_hole_shadow;                    // [hs * hs]
    unsigned long long *d_mirror_mask; int mirror_chunks;   // [hs][mirror_chunks]: per row and 64 units (of 8 pixels) the units the last snapshot found changed
    int *d_mirror_sum; int *h_mirror_sum;       // [8] x0, y0, x1, y1, pixels (low, high), rows, -: what the last snapshot holds (device; pinned host copy)
    hipStream_t mirror_stream; hipEvent_t ev_snap, ev_push;
    void *mirror_dev_ptr;                       // the device address of the registered host array
    unsigned mirror_reg_flags;                  // ... and the flags it was registered with
    uint16_t *mirror_user; bool mirror_direct;  // the caller's array of the last request; whether the device writes it directly (it owns its pages) or the staging buffer
    uint16_t *h_mirror_stage;                   // [hs * hs] pinned: what the device writes for arrays that do not own their pages
    int2 *d_mirror_rows, *h_mirror_rows;        // [hs] per row the first / last changed 8-pixel unit of the last snapshot (device; pinned host copy for the staged form)
    int *d_hole_dirty;            // [4] x0, y0, x1, y1 (inclusive): pixels the HoleMap updates may have changed since the last slamhip_cs_holemap_mirror
    int64_t last_hole_pixels;
    bool hole_pixels_pending;     // ... still on the device (d_key word 6): slamhip_cs_search_and_update returned with the pose, the updates run on

    // ---- K3 ObstacleMap update scratch ---------------------------------------------------------------
    uint32_t *d_o_hits[2];        // [os*os] endpoint hits of a scan; two sets: the fused path's cell pass trails one scan behind its ray walks (obstacle_dev.h)
    uint8_t *d_o_nohit[2];        // [os*os] noHitMap (CoreSLAMProcessor.cs:26,:133)
    int obst_buf;                 // the set the next ray walks use
    bool obst_pending; int obst_pend_buf, obst_pend_max_hits;   // a cell pass not yet applied to d_obst
};

// distance.hip
int32_t cs_alloc_candidates(slamhip_cs *cs, int count);
// waits for the plan stream (before anything a plan launch may still read is freed or rewritten by the host) / releases the plan buffers
int32_t cs_plan_drain(slamhip_cs *cs);
void    cs_plan_free(slamhip_cs *cs);
// a launch in the operator's stream writes what the next plan launches read (candidate gather, scan upload): see plan_inputs_after
static inline void cs_plan_inputs_pending(slamhip_cs *cs) { cs->plan_inputs_after = cs->k1_launches + 1; }
// launches the scan upload that slamhip_cs_set_scan left pending (every launch that reads the scan calls it first)
int32_t cs_flush_scan(slamhip_cs *cs);
int32_t cs_launch_distance(slamhip_cs *cs, int mode, const float pose[3], int count, bool want_dist, bool cand_sane,
                           uint64_t *key_dst);
bool cs_k1_layout_legal(const slamhip_cs *cs);   // distance.hip: are the layout's counts of ray ranges legal for the scan now set?
void cs_layout_idle_refresh(slamhip_cs *cs);   // host only: call between a search's enqueue and the wait for its result
// coreslam.hip: produces a device-generated jitter list that is still pending (slamhip_cs_generate_offsets)
int32_t cs_flush_generate(slamhip_cs *cs);
int32_t cs_side_join(slamhip_cs *cs);
#define CS_RC_NO_PRELAUNCH 77                   // cs_launch_distance, internal: a prelaunch would need a new layout -- nothing was launched
int32_t cs_search_and_update_prelaunched(slamhip_cs *cs, const float *xy, int32_t n, const float pose[3], float hole_width, int32_t quality,
                                         int32_t max_hits, float out_pose[3], int32_t *out_dist, int32_t *out_index, bool *took);   // coreslam.hip; *took = false: nothing done, the caller takes the ordinary order
// developer switch SLAMHIP_FUSED_TIMES=1: host clock between the stages of the per-scan calls (mean over 64 calls, stderr)
struct cs_stage_times {
    bool on; double acc[12], cur[12]; int n; timespec t; const char *what;
    explicit cs_stage_times(const char *w) : on(getenv("SLAMHIP_FUSED_TIMES") != nullptr), n(0), what(w) { for (double &a : acc) a = 0; for (double &a : cur) a = 0; }
    void start() { if (on) clock_gettime(CLOCK_MONOTONIC, &t); }
    void lap(int k) { if (!on) return; timespec u; clock_gettime(CLOCK_MONOTONIC, &u); const double d = (u.tv_sec - t.tv_sec) * 1e6 + (u.tv_nsec - t.tv_nsec) * 1e-3; acc[k] += d; cur[k] += d; t = u; }
    void done()
    {
        if (!on) return;
        {   // (a call that took over half a millisecond -- or SLAMHIP_SLOW_US -- : its own stages, at once)
            double sum = 0; for (double a : cur) sum += a;
            static const double slow_us = getenv("SLAMHIP_SLOW_US") ? atof(getenv("SLAMHIP_SLOW_US")) : 500.0;
            if (sum > slow_us) { fprintf(stderr, "[slamhip] SLOW call (%.0f us), %s:", sum, what); for (double a : cur) fprintf(stderr, " %.1f", a); fprintf(stderr, "\n"); }
            for (double &a : cur) a = 0;
        }
        if (++n < 64) return;
        fprintf(stderr, "[slamhip] host stages (us), %s:", what);
        for (int k = 0; k < 12; k++) fprintf(stderr, " %.2f", acc[k] / n);
        fprintf(stderr, "\n");
        for (double &a : acc) a = 0;
        n = 0;
    }
};
// g_cst, the fused scan: 0 shard / gather launch | 1 K1 host preparation | 2 side join | 3 K1 launch | 4 update launch | 5 idle refresh |
// 6 wait for the pose | 7 next candidates | 8 .. 11 inside the K1 preparation.  g_sst, set_scan: see the laps there.
extern thread_local cs_stage_times g_sst;
extern thread_local cs_stage_times g_cst;
// holemap.hip
int32_t cs_holemap_alloc(slamhip_cs *cs);
int32_t cs_holemap_dirty_set(slamhip_cs *cs, bool all);
int32_t cs_holemap_span_set(slamhip_cs *cs, bool all);    // coreslam.hip: the asynchronous mirror's row spans := whole rows / empty (no-op while no mirror is kept)   // the dirty rectangle := the whole map / empty (enqueued on the operator's stream)
void    cs_holemap_free(slamhip_cs *cs);
int32_t cs_update_maps_enqueue(slamhip_cs *cs, const float pose[3], float hole_width, int32_t quality, int32_t max_hits);
int32_t cs_update_maps_finish(slamhip_cs *cs);
struct k3_ride;
// with_obstacle: the ObstacleMap update of the same scan and pose rides along (or follows in launches of its own when it cannot)
// win: the fused scan's form (d_pose_or_null must then be a pose buffer to receive the decoded winner: the launch decodes the
// search's key itself and delivers key + pose to the mailbox; one-launch form only: cs_holemap_one_launch)
struct cs_k2_winner { const uint64_t *d_key; const float *d_offs_flat; int n_offs; float bx, by, bth; uint32_t *mail; uint32_t seq; };
bool cs_holemap_one_launch(const slamhip_cs *cs);
int32_t cs_launch_holemap_update(slamhip_cs *cs, const float *d_pose_or_null, float4 h_pxcs, float4 h_pxcs_obst, float hole_width, int quality,
                                 bool with_obstacle = false, int max_hits = 0, const cs_k2_winner *win = nullptr);
void cs_obstacle_ride(slamhip_cs *cs, const float *d_pose_or_null, float4 h_pxcs, int max_hits, k3_ride *out);
void cs_obstacle_ride_commit(slamhip_cs *cs, const k3_ride *ride, int max_hits);   // after the carrying launch is in the stream
int32_t cs_obstacle_flush(slamhip_cs *cs);                        // applies a pending cell pass (before anything reads or writes the ObstacleMap)
// coreslam.hip: checksums of HoleMap / ObstacleMap into words 4 / 5 of the result block d_key (enqueued on the operator's stream)
int32_t cs_maps_checksum_enqueue(slamhip_cs *cs);
// obstacle.hip
int32_t cs_obstacle_alloc(slamhip_cs *cs);
void    cs_obstacle_free(slamhip_cs *cs);
int32_t cs_launch_obstacle_update(slamhip_cs *cs, const float *d_pose_or_null, float4 h_pxcs, int max_hits);
