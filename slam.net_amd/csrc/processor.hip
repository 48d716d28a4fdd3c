// processor.hip -- CoreSLAMProcessor state machine on the host (C++), above the operator-level C-ABI.
//
// Mirrors the public C# class (CoreSLAM/CoreSLAMProcessor.cs): ctor :119-162, Reset :167-175,
// ScanSegmentsToCloud :187-207, Update :717-752, properties :40-106.  The Monte-Carlo search and both map
// updates are the HIP kernels; what stays on the host is exactly what stays in C# behind the P/Invoke
// shim: odometry bookkeeping, the polar -> cartesian conversion and the scan counter.
#include "cs_internal.h"
#include "det_trig.h"
#include <vector>
#include <chrono>
#include <stdlib.h>

// developer aid (SLAMHIP_PROC_TIMES=1): average host time of the stages of Update, printed every 64 searching scans
struct proc_times {
    bool on; double acc[6], cur[6]; int n;
    std::chrono::steady_clock::time_point t;
    proc_times() : on(getenv("SLAMHIP_PROC_TIMES") != nullptr), n(0) { for (double &a : acc) a = 0; for (double &a : cur) a = 0; }
    void start() { if (on) t = std::chrono::steady_clock::now(); }
    void lap(int k) { if (!on) return; auto u = std::chrono::steady_clock::now(); const double d = std::chrono::duration<double, std::micro>(u - t).count(); acc[k] += d; cur[k] += d; t = u; }
    void done()
    {
        if (!on) return;
        if (cur[0] + cur[1] + cur[2] + cur[3] > 500.0)     // (a slow call: its own stages, at once)
            fprintf(stderr, "[slamhip] SLOW Update: cloud %.1f | set_scan %.1f | candidates %.1f | search+update %.1f us\n", cur[0], cur[1], cur[2], cur[3]);
        for (double &a : cur) a = 0;
        if (++n < 64) return;
        fprintf(stderr, "[slamhip] Update host stages (us): cloud %.1f | set_scan %.1f | candidates %.1f | search+update (enqueue, wait, read back) %.1f\n",
                acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n);
        for (double &a : acc) a = 0;
        n = 0;
    }
};
static thread_local proc_times g_pt;

struct slamhip_csproc {
    slamhip_ctx *ctx;
    slamhip_cs *cs;
    float start_pose[3], pose[3], last_odo[3];            // :25, :106, :35
    int scan_count;                                       // :34
    float sigma_xy, sigma_theta; int iters, threads;      // :56-71
    int quality; float hole_width; int search_beginning, unmapped_hits, max_hits;   // :80-101
    uint64_t seed, scan_no;
    bool pinned;
    bool lattice;                                         // candidates as a heading lattice (slamhip_csproc_set_lattice)
    std::vector<float> cloud;
    // ScanSegmentsToCloud: a lidar's ray angles repeat from scan to scan, so the deterministic sine / cosine of ray r is kept with
    // the angle it was made for and reused while the angle is bit-identical (the same floats by construction; 6 -> 1.5 us per scan)
    std::vector<float> trig_angle, trig_s, trig_c;
};

extern "C" int32_t slamhip_csproc_destroy(slamhip_csproc *p)
{
    if (!p) return SLAMHIP_OK;
    slamhip_cs_destroy(p->cs);
    delete p;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_csproc_reset(slamhip_csproc *p)
{
    SH_CHECK_ARG(p);
    SH_TRY(slamhip_cs_reset(p->cs, p->unmapped_hits));                    // :169-170
    memcpy(p->pose, p->start_pose, sizeof(float) * 3);                    // :172
    p->last_odo[0] = p->last_odo[1] = p->last_odo[2] = 0.0f;              // :173
    p->scan_count = 0;                                                    // :174
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_csproc_create(slamhip_ctx *ctx, float physical, int32_t hole_size, int32_t obst_size,
                                         const float start_pose[3], float sigma_xy, float sigma_theta,
                                         int32_t iterations_per_thread, int32_t num_threads, slamhip_csproc **out)
{
    SH_CHECK_ARG(ctx && start_pose && out && iterations_per_thread >= 0);
    slamhip_cs *cs = nullptr;
    SH_TRY(slamhip_cs_create(ctx, physical, hole_size, obst_size, &cs));  // :131-133
    slamhip_csproc *p = new slamhip_csproc();
    p->ctx = ctx; p->cs = cs;
    memcpy(p->start_pose, start_pose, sizeof(float) * 3);                 // :124
    p->sigma_xy = sigma_xy; p->sigma_theta = sigma_theta;                 // :125-126
    p->iters = iterations_per_thread; p->threads = num_threads;           // :127-128
    p->quality = 50; p->hole_width = 0.6f; p->search_beginning = 5;       // :80,:85,:90
    p->unmapped_hits = -5; p->max_hits = 10;                              // :96,:101
    p->seed = 0x5EED5EEDull; p->scan_no = 0; p->pinned = false;
    int32_t rc = slamhip_csproc_reset(p);                                 // :140
    if (rc != SLAMHIP_OK) { slamhip_csproc_destroy(p); return rc; }
    *out = p;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_csproc_set_params(slamhip_csproc *p, int32_t quality, float hole_width, int32_t search_beginning,
                                             int32_t unmapped_hits, int32_t max_hits)
{
    SH_CHECK_ARG(p && quality >= 0 && quality <= 256 && unmapped_hits >= -128 && unmapped_hits <= 127 &&
                 max_hits >= -128 && max_hits <= 127);
    p->quality = quality; p->hole_width = hole_width; p->search_beginning = search_beginning;
    p->unmapped_hits = unmapped_hits; p->max_hits = max_hits;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_csproc_set_seed(slamhip_csproc *p, uint64_t seed)
{
    SH_CHECK_ARG(p);
    p->seed = seed;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_csproc_set_lattice(slamhip_csproc *p, int32_t on)
{
    SH_CHECK_ARG(p);
    p->lattice = on != 0;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_csproc_set_offsets(slamhip_csproc *p, const float *offs, int32_t n)
{
    SH_CHECK_ARG(p);
    SH_TRY(slamhip_cs_set_offsets(p->cs, offs, n));
    p->pinned = true;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_csproc_get_pose(slamhip_csproc *p, float out[3])
{
    SH_CHECK_ARG(p && out);
    memcpy(out, p->pose, sizeof(float) * 3);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_csproc_cs(slamhip_csproc *p, slamhip_cs **out)
{
    SH_CHECK_ARG(p && out);
    *out = p->cs;
    return SLAMHIP_OK;
}

// ScanSegmentsToCloud (:187-207) on its own: the arithmetic of slamhip_csproc_update's loop below, without the per-ray trigonometry cache
extern "C" int32_t slamhip_scan_segments_to_cloud(const float *seg_poses, const int32_t *seg_start, int32_t n_seg, const float *rays, float *out_xy)
{
    SH_CHECK_ARG(seg_poses && seg_start && n_seg >= 1 && seg_start[0] == 0);
    for (int sgm = 0; sgm < n_seg; sgm++) SH_CHECK_ARG(seg_start[sgm] <= seg_start[sgm + 1]);
    SH_CHECK_ARG(seg_start[n_seg] == 0 || (rays && out_xy));
    const float odo[3] = { seg_poses[3 * (n_seg - 1)], seg_poses[3 * (n_seg - 1) + 1], seg_poses[3 * (n_seg - 1) + 2] };  // :719
    for (int sgm = 0; sgm < n_seg; sgm++) {                               // :191
        const float px = seg_poses[3 * sgm] - odo[0];                     // :194
        const float py = seg_poses[3 * sgm + 1] - odo[1];
        const float pz = seg_poses[3 * sgm + 2] - odo[2];
        for (int r = seg_start[sgm]; r < seg_start[sgm + 1]; r++) {       // :196
            float s, c;
            sh_det_sincosf(rays[2 * r] + pz, &s, &c);                     // :200-201 angle + pose.Z
            out_xy[2 * (size_t)r] = px + rays[2 * r + 1] * c;             // :200
            out_xy[2 * (size_t)r + 1] = py + rays[2 * r + 1] * s;         // :201
        }
    }
    return SLAMHIP_OK;
}

// Update (:717-752)
extern "C" int32_t slamhip_csproc_update(slamhip_csproc *p, const float *seg_poses, const int32_t *seg_start,
                                         int32_t n_seg, const float *rays)
{
    SH_CHECK_ARG(p && seg_poses && seg_start && n_seg >= 1 && seg_start[0] == 0);
    const int n = seg_start[n_seg];
    SH_CHECK_ARG(n >= 0 && (rays || n == 0));
    for (int sgm = 0; sgm < n_seg; sgm++) SH_CHECK_ARG(seg_start[sgm] <= seg_start[sgm + 1]);
    const float odo[3] = { seg_poses[3 * (n_seg - 1)], seg_poses[3 * (n_seg - 1) + 1], seg_poses[3 * (n_seg - 1) + 2] };  // :719

    // ScanSegmentsToCloud (:187-207): polar -> cartesian in the robot frame, on the host
    g_pt.start();
    p->cloud.resize((size_t)n * 2);
    if (p->trig_angle.size() < (size_t)n) {
        const size_t old = p->trig_angle.size();
        p->trig_angle.resize((size_t)n); p->trig_s.resize((size_t)n); p->trig_c.resize((size_t)n);
        for (size_t k = old; k < (size_t)n; k++) { p->trig_angle[k] = NAN; p->trig_s[k] = 0.f; p->trig_c[k] = 0.f; }   // (NaN never compares equal: recomputed)
    }
    for (int sgm = 0; sgm < n_seg; sgm++) {                               // :191
        const float px = seg_poses[3 * sgm] - odo[0];                     // :194
        const float py = seg_poses[3 * sgm + 1] - odo[1];
        const float pz = seg_poses[3 * sgm + 2] - odo[2];
        for (int r = seg_start[sgm]; r < seg_start[sgm + 1]; r++) {       // :196
            float s, c;
            const float ang = rays[2 * r] + pz;                           // :200-201 angle + pose.Z
            if (ang == p->trig_angle[(size_t)r]) { s = p->trig_s[(size_t)r]; c = p->trig_c[(size_t)r]; }
            else { sh_det_sincosf(ang, &s, &c); p->trig_angle[(size_t)r] = ang; p->trig_s[(size_t)r] = s; p->trig_c[(size_t)r] = c; }
            p->cloud[2 * (size_t)r] = px + rays[2 * r + 1] * c;           // :200
            p->cloud[2 * (size_t)r + 1] = py + rays[2 * r + 1] * s;       // :201
        }
    }
    g_pt.lap(0);

    float new_pose[3];
    if (p->scan_count >= p->search_beginning && n > 0) {                  // :726
        float search[3];
        for (int i = 0; i < 3; i++) search[i] = p->pose[i] + (odo[i] - p->last_odo[i]);   // :728
        // (the candidates first: they do not depend on the scan, and the search launch may precede the scan's tables -- below)
        if (!p->pinned) {
            const int nj = (p->threads > 0 ? p->threads : 1) * p->iters;  // :143-160, :662-665
            if (p->lattice) SH_TRY(slamhip_cs_generate_offsets_lattice(p->cs, nj, p->sigma_xy, p->sigma_theta, p->seed, p->scan_no));
            else SH_TRY(slamhip_cs_generate_offsets(p->cs, nj, p->sigma_xy, p->sigma_theta, p->seed, p->scan_no));
        }
        g_pt.lap(2);
        // set_scan (:723), search (:732), NormalizeAngle (:746) and both map updates (:750-751): the search launch first where it can
        SH_TRY(slamhip_cs_scan_search_and_update(p->cs, p->cloud.data(), n, search, p->hole_width, p->quality, p->max_hits, new_pose, nullptr, nullptr));
        // (state moves only when the scan went through: a failed call leaves the odometry baseline and the list number where they
        // were, as the reference's exception would -- the next Update searches from the same baseline)
        p->scan_no++;
        memcpy(p->last_odo, odo, sizeof(odo));                            // :745
        memcpy(p->pose, new_pose, sizeof(new_pose));                      // :747
        g_pt.lap(3); g_pt.done();
        return SLAMHIP_OK;
    }
    SH_TRY(slamhip_cs_set_scan(p->cs, p->cloud.data(), n));               // :723
    g_pt.lap(1);
    if (p->scan_count < p->search_beginning) p->scan_count++;             // :741
    else if (n == 0) {
        // searching scan with an empty cloud: every distance is int.MaxValue, the base pose wins (:257,:626-628)
        for (int i = 0; i < 3; i++) new_pose[i] = p->pose[i] + (odo[i] - p->last_odo[i]);
        memcpy(p->last_odo, odo, sizeof(odo));
        new_pose[2] = sh_normalize_angle(new_pose[2]);
        memcpy(p->pose, new_pose, sizeof(new_pose));
        return SLAMHIP_OK;
    }
    memcpy(new_pose, odo, sizeof(odo));                                   // :742
    memcpy(p->last_odo, odo, sizeof(odo));                                // :745
    new_pose[2] = sh_normalize_angle(new_pose[2]);                        // :746
    memcpy(p->pose, new_pose, sizeof(new_pose));                          // :747
    SH_TRY(slamhip_cs_update_holemap(p->cs, p->pose, p->hole_width, p->quality));   // :750
    SH_TRY(slamhip_cs_update_obstaclemap(p->cs, p->pose, p->max_hits));             // :751
    return SLAMHIP_OK;
}
