/*
 * slamhip.h -- C-ABI of libslamhip.so: the MI355X-native CoreSLAM / HectorSLAM hot path.
 *
 * This is the drop-in boundary for mikkleini/slam.net.  The reference is 100 % managed C# with no
 * FFI seam of its own: its hot path is private methods of CoreSLAMProcessor and
 * ScanMatcher/OccGridMap.  The entry points below are what a P/Invoke shim behind the unchanged public
 * C# API (CoreSLAMProcessor / HoleMap / ObstacleMap / ScanMatcher / OccGridMap / MapRepMultiMap /
 * HectorSLAMProcessor) binds; each one cites the reference member it replaces
 * (paths relative to the reference repo).  INTEGRATION.md shows the C# [DllImport] stubs.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, blittable structs only; no callbacks, no exceptions.
 *   - every function returns int32 status: 0 = SLAMHIP_OK, < 0 = error; slamhip_last_error() returns
 *     a thread-local message.  Degenerate *data* (empty scan, robot outside the map, no point in
 *     bounds ...) is a silent no-op exactly as in the reference; negative status is reserved for
 *     API misuse and HIP / RCCL failures.
 *   - host pointers are only read/written during the call (pin with `fixed` / GCHandle); the
 *     library copies in/out and never retains them.  Device memory lives behind opaque handles.
 *   - a handle is single-caller (like the reference objects: ParallelWorker.Work is "blocking,
 *     non-reentrant", BaseSLAM/ParallelWorker.cs:95); different handles may be used from different
 *     threads, also handles that share one slamhip_ctx -- with per-kernel timing off
 *     (slamhip_ctx_timing_enable(ctx, 0), the default: the timers' event pool is not locked): the
 *     context's completion mailbox is guarded by a lock, so the blocking calls of such handles take
 *     turns (they share one HIP stream anyway); give each thread its own context if they should
 *     overlap or be timed.
 *   - calls block until their RESULT is on the host unless the name ends in _async.  Two calls
 *     return as soon as the result the caller reads is there while work that returns nothing is
 *     still running on the device: slamhip_cs_search_and_update / slamhip_csproc_update (back with
 *     the pose; the two map updates run on) and slamhip_hsproc_update (the grid update runs on).
 *     Every later call on the same context is ordered behind that work, so the caller sees the
 *     reference's sequential semantics.  A blocking wait polls a pinned word for up to ~300 us and
 *     then goes on polling it between 20 us sleeps, asking the stream for faults (it never waits for
 *     the stream itself: work queued behind the result is not waited for).
 *   - float poses are (x [m], y [m], theta [rad]) = System.Numerics.Vector3; points are
 *     System.Numerics.Vector2 (8 B); LogOddsCell is {int32 UpdateIndex; float Value} (8 B).
 *   - there is no CPU fallback: every compute entry point launches HIP kernels on gfx950.
 */
#ifndef SLAMHIP_H
#define SLAMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMHIP_OK            0
#define SLAMHIP_ERR_INVALID  (-1)   /* bad argument / API misuse */
#define SLAMHIP_ERR_HIP      (-2)   /* HIP runtime failure (message in slamhip_last_error) */
#define SLAMHIP_ERR_NOMEM    (-3)
#define SLAMHIP_ERR_STATE    (-4)   /* call order violated (e.g. search before set_scan) */
#define SLAMHIP_ERR_RCCL     (-5)
#define SLAMHIP_ERR_TIMEOUT  (-6)   /* a blocking wait passed its bound; the context is poisoned (slamhip_ctx_set_wait_timeout) */

typedef struct slamhip_ctx    slamhip_ctx;     /* one GPU + one HIP stream */
typedef struct slamhip_cs     slamhip_cs;      /* CoreSLAM device state: HoleMap + ObstacleMap + scan + candidates */
typedef struct slamhip_csproc slamhip_csproc;  /* CoreSLAMProcessor state machine on top of slamhip_cs */
typedef struct slamhip_hs     slamhip_hs;      /* HectorSLAM MapRepMultiMap pyramid + ScanMatcher */
typedef struct slamhip_hsproc slamhip_hsproc;  /* HectorSLAMProcessor state machine on top of slamhip_hs */
typedef struct slamhip_group  slamhip_group;   /* N GPUs in one process + RCCL communicator */
typedef struct slamhip_comm   slamhip_comm;    /* one rank (one process per GPU) of an RCCL communicator */

/* {int UpdateIndex; float Value} -- HectorSLAM/Map/LogOddsCell.cs:16-21 */
typedef struct { int32_t update_index; float value; } slamhip_cell;

/* ------------------------------------------------------------------------------------------------
 * Library / context
 * ---------------------------------------------------------------------------------------------- */
const char *slamhip_version(void);
const char *slamhip_last_error(void);
int32_t slamhip_device_count(int32_t *out_count);

/* Replaces `new ParallelWorker(numThreads)` (BaseSLAM/ParallelWorker.cs:34-56): the execution resource
 * the hot path runs on.  One context = one GPU ordinal + one non-blocking HIP stream, the OPERATOR's stream: everything a caller
 * can observe is ordered on it.  (The context also owns three helper streams -- plan launches, the next scan's candidate list,
 * host-mirror pushes -- created and bound to hardware queues together with it: create contexts early and keep them; a
 * process should not need more than one per GPU.) */
int32_t slamhip_ctx_create(int32_t device_ordinal, slamhip_ctx **out);
int32_t slamhip_ctx_destroy(slamhip_ctx *ctx);                     /* ParallelWorker.Dispose :122-139 */
int32_t slamhip_ctx_synchronize(slamhip_ctx *ctx);
int32_t slamhip_ctx_device(slamhip_ctx *ctx, int32_t *out_ordinal);
void   *slamhip_ctx_stream(slamhip_ctx *ctx);                      /* hipStream_t, for interop (RCCL / torch) */
/* Bound on every blocking wait of the context, in milliseconds (default: environment SLAMHIP_WAIT_TIMEOUT_MS, else 10000;
 * <= 0: unbounded).  ParallelWorker.Work waits on its AutoResetEvents without a bound (BaseSLAM/ParallelWorker.cs:106-116): a worker
 * that never signals hangs the reference's caller for ever; here a completion word that does not arrive in time ends the call with
 * SLAMHIP_ERR_TIMEOUT and POISONS the context -- the device's state is unknown, nothing is restarted or re-executed in-process, and
 * every later blocking call (every wait and every hand-over of a result to the host) on the context fails with SLAMHIP_ERR_TIMEOUT
 * at once; enqueue-only calls are not checked.  The bound counts from the start of the wait, work queued in front of the awaited
 * launch included: a caller that queues seconds of asynchronous work in front of a blocking call raises the bound first.  The caller
 * destroys its handles -- slamhip_ctx_destroy of a poisoned context waits for the stream once more, for at most the bound, and then
 * lets go of it (a kernel that never ends would otherwise move the hang into destroy); the process should then exit. */
int32_t slamhip_ctx_set_wait_timeout(slamhip_ctx *ctx, int64_t timeout_ms);
int32_t slamhip_ctx_poisoned(slamhip_ctx *ctx, int32_t *out_flag);
/* Test hook (no device involved): the wait loop of a blocking call on a caller-owned word -- returns SLAMHIP_OK once *flag has
 * reached `val` (wrap-safe), SLAMHIP_ERR_TIMEOUT after timeout_ms. */
int32_t slamhip_debug_flag_wait(volatile uint32_t *flag, uint32_t val, int64_t timeout_ms);

/* Kernel timing (the reference only has Stopwatch EMAs, HectorSLAMProcessor.cs:92-96,111-115).
 * When enabled, each kernel class is bracketed by HIP events on the context's stream. */
enum {
    SLAMHIP_K_CS_PREP = 0,      /* candidate transform (pose + offset -> px,py,c,s) */
    SLAMHIP_K_CS_DISTANCE = 1,  /* K1 batched distance (dominant kernel) */
    SLAMHIP_K_CS_REDUCE = 2,    /* K1r partial-sum + arg-min reduce */
    SLAMHIP_K_CS_HOLEMAP = 3,   /* K2 */
    SLAMHIP_K_CS_OBSTACLE = 4,  /* K3 */
    SLAMHIP_K_HS_MATCH = 5,     /* K4 */
    SLAMHIP_K_HS_UPDATE = 6,    /* K5 */
    SLAMHIP_K_COUNT = 7
};
/* mask: bit k enables kernel class k (e.g. 1 << SLAMHIP_K_CS_DISTANCE); 0 = off; -1 = all classes */
int32_t slamhip_ctx_timing_enable(slamhip_ctx *ctx, int32_t mask);
int32_t slamhip_ctx_timing_reset(slamhip_ctx *ctx);
/* total milliseconds and launch count accumulated for one kernel class since the last reset
 * (synchronises the stream) */
int32_t slamhip_ctx_timing_get(slamhip_ctx *ctx, int32_t which, double *out_ms, int64_t *out_launches);

/* ------------------------------------------------------------------------------------------------
 * CoreSLAM, operator level
 * ---------------------------------------------------------------------------------------------- */

/* Replaces `new HoleMap(holeMapSize, physicalMapSize)`, `new ObstacleMap(obstacleMapSize, ...)` and
 * the noHitMap (CoreSLAM/CoreSLAMProcessor.cs:131-133; HoleMap.cs:17-22; ObstacleMap.cs:17-22).
 * Scale = sizePixels / sizeMeters in binary32.  Maps are created in the Reset() state. */
int32_t slamhip_cs_create(slamhip_ctx *ctx, float physical_map_size, int32_t hole_map_size,
                          int32_t obstacle_map_size, slamhip_cs **out);
int32_t slamhip_cs_destroy(slamhip_cs *cs);
int32_t slamhip_cs_info(slamhip_cs *cs, int32_t *hole_size, float *hole_scale, int32_t *obst_size, float *obst_scale);

/* CoreSLAMProcessor.Reset map part (:169-170): HoleMap := 32750, ObstacleMap := unmapped_obstacle_hits */
int32_t slamhip_cs_reset(slamhip_cs *cs, int32_t unmapped_obstacle_hits);

/* HoleMap.Pixels (HoleMap.cs:27): ushort[Size*Size] row-major.  n_pixels must equal Size*Size. */
int32_t slamhip_cs_holemap_upload(slamhip_cs *cs, const uint16_t *pixels, size_t n_pixels);
int32_t slamhip_cs_holemap_download(slamhip_cs *cs, uint16_t *pixels, size_t n_pixels);
/* Live HoleMap.Pixels (HoleMap.cs:27; callers read the array directly, Simulation/MainWindow.xaml.cs:229) at the price of what
 * changed: copies into the caller's full-size array only the bounding rectangle of the scans drawn since the previous mirror call
 * (everything on the first call and after reset / upload).  out_rect (optional) = {x0, y0, x1, y1} inclusive, or {0, 0, -1, -1}
 * when nothing changed.  `pixels` must be the array the previous mirror call filled. */
int32_t slamhip_cs_holemap_mirror(slamhip_cs *cs, uint16_t *pixels, size_t n_pixels, int32_t out_rect[4]);
/* The same without stalling the scan (the reference's callers read HoleMap.Pixels live: HoleMap.cs:27,
 * Simulation/MainWindow.xaml.cs:227-249).  From the first call on the HoleMap updates keep, per map row, the span of columns their
 * rays crossed.  _async enqueues behind everything queued so far ONE launch that copies those spans into a shadow map on the
 * device and rests them, and -- on a copy stream, behind an event -- a launch that stores the shadow's spans straight into
 * `pixels`; it returns at once, and the next search starts as soon as the snapshot is taken.  _wait blocks until the last push
 * has landed; out_rect = the bounding rectangle x0, y0, x1, y1 of what it refreshed (x1 < x0: nothing), *out_pixels = the pixels
 * of its spans.  After _wait, `pixels` equals a full download taken at the moment of the _async call.
 * Contract: `pixels` must stay alive, and must not be read, between _async and _wait (the caller passes it once; _wait uses it).
 * WHERE THE DEVICE WRITES depends on the array: one that OWNS its pages -- it starts on a 4096-byte boundary and is a whole number
 * of pages long (posix_memalign, NativeMemory.AlignedAlloc, mmap; 8 MiB at 2048^2) -- is page-locked and mapped into the device's
 * address space on its first use (hipHostRegister) and written by the device directly; it stays registered until
 * slamhip_cs_holemap_mirror_release, slamhip_cs_destroy, or a call with another array.  Any other array (a managed array on the
 * pinned object heap, a NumPy array: they share their first and last page with other objects, which the runtime pins for its own
 * copies -- registering such pages ended in device write faults in the randomised soak) is served through a pinned staging buffer
 * of the library's, and _wait copies the changed row ranges into it on the HOST (a pass over what changed; measured at 2048^2 with
 * a moving robot: 225 us per scan against 164 with a request per scan).  One push is in flight at a time: _async waits for the
 * previous one first.  The blocking slamhip_cs_holemap_mirror follows the same rule for its copies. */
int32_t slamhip_cs_holemap_mirror_async(slamhip_cs *cs, uint16_t *pixels, size_t n_pixels);
int32_t slamhip_cs_holemap_mirror_wait(slamhip_cs *cs, int32_t out_rect[4], int64_t *out_pixels);
int32_t slamhip_cs_holemap_mirror_release(slamhip_cs *cs);   /* waits, then unregisters the array (either mirror form) */
/* HoleMap.GetPackedPixels (HoleMap.cs:44-55): 4-bit packing done on the device; n_bytes = Size*Size/2 */
int32_t slamhip_cs_holemap_download_packed(slamhip_cs *cs, uint8_t *packed, size_t n_bytes);
/* ObstacleMap.Pixels (ObstacleMap.cs:31): sbyte[Size,Size], [y,x] row-major */
int32_t slamhip_cs_obstaclemap_upload(slamhip_cs *cs, const int8_t *pixels, size_t n_pixels);
int32_t slamhip_cs_obstaclemap_download(slamhip_cs *cs, int8_t *pixels, size_t n_pixels);

/* The ScanCloud of the current Update (output of ScanSegmentsToCloud, CoreSLAMProcessor.cs:187-207):
 * n_points robot-frame points (x,y).  Kept on the device for the search and both map updates.  xy may be reused as soon as the
 * call returns.  (On a device with a large BAR the call stores the scan straight into one of two alternating device blocks --
 * CPU stores, no upload launch -- whenever it knows that block idle; otherwise the first launch that reads the scan uploads it.
 * SLAMHIP_NO_DIRECT_UPLOAD=1 always takes the second way.) */
int32_t slamhip_cs_set_scan(slamhip_cs *cs, const float *xy, int32_t n_points);

/* CalculateDistance (CoreSLAMProcessor.cs:215-259) for K candidates in one batched kernel.
 * pxcs is K x 4: px = x*Scale+0.5f, py = y*Scale+0.5f, c = cos(theta)*Scale, s = sin(theta)*Scale
 * (:232-235) computed by the caller, so the CRT trig stays on the host and every distance is
 * bit-exact with the reference for the same (px,py,c,s).  out_dist (K, may be NULL) receives every
 * distance; the arg-min uses the reference tie-break (first strictly smaller, :644,:700). */
int32_t slamhip_cs_distance_pxcs(slamhip_cs *cs, const float *pxcs, int32_t K, int32_t *out_dist,
                                 int32_t *out_best_index, int32_t *out_best_dist);
/* Same from poses (K x 3: x,y,theta); (px,py,c,s) are formed on the device with the deterministic
 * correctly-rounded float sin/cos (csrc/det_trig.h). */
int32_t slamhip_cs_distance_poses(slamhip_cs *cs, const float *poses, int32_t K, int32_t *out_dist,
                                  int32_t *out_best_index, int32_t *out_best_dist);

/* The pre-drawn jitter list: replaces FillRandomQueues + the Redzen samplers
 * (CoreSLAMProcessor.cs:136-137,:599-612).  offs is n x 3 (dx, dy, dtheta) in the reference draw order
 * X, Y, theta (:635-637), flat thread-major: entry t*iterations+i is thread t's i-th draw.
 * Like the reference's background refill (:692) this is off the search's critical path. */
int32_t slamhip_cs_set_offsets(slamhip_cs *cs, const float *offs, int32_t n);
/* Device-side generation: counter-based Philox4x32-10 + Box-Muller keyed by
 * (seed, stream, index), so a shard of the list has the same values on every GPU. */
int32_t slamhip_cs_generate_offsets(slamhip_cs *cs, int32_t n, float sigma_xy, float sigma_theta,
                                    uint64_t seed, uint64_t stream);
/* The per-scan flow (CoreSLAMProcessor.Update: a fresh list per scan, :662-665) asks for (n, sigmas, seed, stream), then
 * stream + 1, + 2, ...: a slamhip_cs_search_and_update on a generated list prepares the list of stream + 1 ahead, on a stream of
 * its own, while it waits for the pose, and the next slamhip_cs_generate_offsets that asks for exactly that list finds it in
 * place (any other request simply gets what it asks for).  Diagnostics: how many requests were served that way, and how many
 * lists were prepared, since the handle was created. */
int32_t slamhip_cs_prepared_lists(slamhip_cs *cs, uint64_t *out_served, uint64_t *out_prepared);
/* Opt-in, no reference counterpart (the reference draws every candidate's heading independently, :599-612): the same generator
 * with the headings on a LATTICE -- the candidates one lane of the search kernel evaluates (2 per lane below 65 536 candidates, 4
 * from there on) share their dtheta bit for bit and differ in their translation; the headings are the strata of
 * N(0, sigma_theta), one per lane position (n / 2 or n / 4 distinct headings instead of n), the translations N(0, sigma_xy) as
 * before.  A full-range search over such a list forms the products c*X, s*Y, s*X, c*Y of a ray point once per lane instead of
 * once per candidate: 2 (3) of the ~12.75 vector operations per candidate and ray less, every candidate's distance the float
 * arithmetic of :240-241 as ever (the list can be downloaded and handed to any checker).  Same (seed, stream, index) keying.
 * Up to 12 288 candidates the search evaluates one candidate per lane (smaller groups, faster there) and the call produces the
 * plain list of slamhip_cs_generate_offsets. */
int32_t slamhip_cs_generate_offsets_lattice(slamhip_cs *cs, int32_t n, float sigma_xy, float sigma_theta,
                                            uint64_t seed, uint64_t stream);
int32_t slamhip_cs_offsets_download(slamhip_cs *cs, float *offs, int32_t n);
/* Known-answer access to the generator's integer stream (what FillRandomQueues' Redzen samplers, CoreSLAMProcessor.cs:599-612, are
 * replaced by has no reference vectors of its own, so it is pinned to its published specification instead): ONE Philox4x32-10 block
 * computed on the device -- counter[4], key[2] -> out[4] -- to be compared with Random123's kat_vectors (tests/test_gpu_coreslam.py;
 * the same vectors pin the NumPy restatement, oracle/np_oracle.py: philox4x32_10, philox_jitters).  Jitter i of a generated list uses
 * counter (i, 0, stream low, stream high) and key (seed low, seed high). */
int32_t slamhip_ctx_philox4x32_10(slamhip_ctx *ctx, const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);

/* ParallelMonteCarloSearch / SingleMonteCarloSearch (CoreSLAMProcessor.cs:624-710) over the flat
 * candidate list: candidate 0 = search_pose itself (:626-628), candidate k = search_pose + offs[k-1].
 * Returns the winning pose, its distance and its flat index. */
int32_t slamhip_cs_search(slamhip_cs *cs, const float search_pose[3], float out_pose[3],
                          int32_t *out_dist, int32_t *out_index);
/* Shard of the same search: only flat candidates [first, first+count) are evaluated.  The result is
 * the packed key (uint64(distance) << 32) | flat_index; min over shards == the full search
 * (cross-thread arg-min :695-705; cross-GPU: one RCCL min all-reduce of this key). */
int32_t slamhip_cs_search_shard(slamhip_cs *cs, const float search_pose[3], int32_t first, int32_t count,
                                uint64_t *out_key);
/* Asynchronous form: enqueues on the context's stream and writes the key to DEVICE memory d_out_key
 * (8 bytes, e.g. a torch tensor fed to an RCCL all-reduce on the same stream). */
int32_t slamhip_cs_search_shard_async(slamhip_cs *cs, const float search_pose[3], int32_t first,
                                      int32_t count, uint64_t *d_out_key);
/* Enqueue-only form with a result word owned by the handle: *d_key receives the DEVICE address of the word that holds the packed
 * key once the stream reaches that point -- one of a ring of 4 words, valid until three further RING LAUNCHES on the handle have
 * been enqueued: calls of this function, fused scans (slamhip_cs_search_and_update and its _pxcs form) and the all-reduce forms
 * (slamhip_cs_search_allreduce*, which also overwrite the slot in place with the reduced key) all advance the same ring.  A caller
 * that keeps a word across such calls copies it first (slamhip_cs_key_read).  No caller memory is involved, so the kernel needs no final arriver: the workgroups that complete
 * candidates min their keys straight into the word (the previous call's launch left it all ones), and the end of the launch is
 * the completion (the cross-thread arg-min of :695-705 as fire-and-forget atomics).  slamhip_cs_key_read waits for the handle's
 * stream and copies one such word to the host.
 * Backpressure (round 6): every such launch is accompanied by its plan (slamhip_cs_plan_stats), whose buffers exist eight times, so
 * a caller that enqueues searches faster than the device runs them is held in this call until the search seven launches back has
 * started (the device still has six searches queued: nothing idles) -- at most 20 ms or the context's wait bound, whichever is
 * shorter; past that the search is launched without a plan. */
int32_t slamhip_cs_search_shard_enqueue(slamhip_cs *cs, const float search_pose[3], int32_t first, int32_t count,
                                        const uint64_t **d_key);
int32_t slamhip_cs_key_read(slamhip_cs *cs, const uint64_t *d_key, uint64_t *out_key);
/* Recompute the winner's pose from a (possibly all-reduced) key: search_pose + offs[index-1]. */
int32_t slamhip_cs_pose_from_key(slamhip_cs *cs, const float search_pose[3], uint64_t key,
                                 float out_pose[3], int32_t *out_dist, int32_t *out_index);

/* UpdateHoleMap + DrawLaserRayOnHoleMap + ClipRay (CoreSLAMProcessor.cs:320-443,:496-534) with the
 * current scan.  Bit-exact uint16 result including the ray-order dependence of the blend (:431). */
int32_t slamhip_cs_update_holemap(slamhip_cs *cs, const float pose[3], float hole_width, int32_t quality);
int32_t slamhip_cs_update_holemap_pxcs(slamhip_cs *cs, const float pxcs[4], float hole_width, int32_t quality);
/* UpdateObstacleMap + DrawLaserRayOnObstacleMap (:456-490,:540-593) */
int32_t slamhip_cs_update_obstaclemap(slamhip_cs *cs, const float pose[3], int32_t max_obstacle_hits);
int32_t slamhip_cs_update_obstaclemap_pxcs(slamhip_cs *cs, const float pxcs[4], int32_t max_obstacle_hits);
/* number of pixels blended by the last HoleMap update (4 algorithmic bytes each); after
 * slamhip_cs_search_and_update the figure is fetched here (this call then waits for the update) */
int32_t slamhip_cs_last_holemap_pixels(slamhip_cs *cs, int64_t *out_pixels);
/* Replica check (no reference counterpart; SURVEY.md sec.8e: on several GPUs the map updates run as replicas, kept
 * bit-identical by the integer-exact kernels): 64-bit checksums of the HoleMap (out[0]; HoleMap.cs:27 Pixels as uint16) and
 * the ObstacleMap (out[1]; ObstacleMap.cs:25 as bytes) behind everything enqueued so far:
 *     sum over i of mix64(i << 32 | element_i) mod 2^64,  mix64 = the SplitMix64 finaliser
 * (position-sensitive, order-independent; a host can recompute it from a download). */
int32_t slamhip_cs_maps_checksum(slamhip_cs *cs, uint64_t out[2]);

/* Diagnostics: with the environment variable SLAMHIP_K1_VERIFY=1 the distance kernel checks every end
 * point against the LDS tile box it derived by interval arithmetic and counts violations (always 0 when
 * the box reasoning holds; the parity tests assert it).  Returns the count accumulated so far. */
int32_t slamhip_cs_selfcheck_failures(slamhip_cs *cs, uint32_t *out_failures);

/* Diagnostics of slamhip_cs_scan_search_and_update / slamhip_csproc_update (CoreSLAMProcessor.cs:717-752): the search launch of a scan goes
 * into the stream BEFORE the scan's tables are made and waits for them on the device (DESIGN.md sec.4 "The per-scan flow").
 * out[0] scans searched that way, out[1] such launches abandoned (the last scan's launch layout did not serve the new scan: searched
 * again in the ordinary order), out[2] scans whose layout had to be remade first, out[3] scans refused (first scans, a changed ray
 * count, a new candidate list, SLAMHIP_PRELAUNCH=0 ...).  The results do not depend on the path taken. */
int32_t slamhip_cs_prelaunch_stats(slamhip_cs *cs, uint64_t out[4]);

/* Diagnostics of the search's PLAN (round 6; replaces nothing in the reference -- it takes the per-workgroup planning of the batched
 * CalculateDistanceSISD, CoreSLAMProcessor.cs:226-259, off the search launch's critical path): every search launch of the
 * tiled kernel in its enqueue-only form is accompanied by one small launch on a stream of its own that leaves each workgroup's tile
 * steps (bounds -> boxes -> steps: what its prologue would work out in front of its first tile) in device memory, stamped; a workgroup
 * uses the record that carries its stamp and plans for itself otherwise.
 * out[0] searches launched with a plan, out[1] without (blocking searches, explicit pose lists, lattice lists, prelaunched searches,
 * SLAMHIP_K1_PLAN=0), out[2] times the host waited for a free plan slot (it was seven searches ahead of the device), out[3] plans not launched because a
 * launch that writes their inputs (candidate gather, scan upload) had not been seen to finish.  Results never depend on the path. */
int32_t slamhip_cs_plan_stats(slamhip_cs *cs, uint64_t out[4]);

/* Fused configuration C3 (device boundary at CoreSLAMProcessor.cs:732,:750,:751): search, NormalizeAngle
 * (:746) and both map updates in one call; the winning pose never leaves the device between them.
 * Completion: the call returns when the winning pose is back on the host.  The two map updates are enqueued
 * behind the search on the operator's stream and may still be running; every later call that reads or writes
 * the maps (the next search, an update, a download, an export, destroy) is ordered behind them, so a caller
 * sees the reference's sequential semantics -- it only gets the pose ~40 us earlier and can prepare its next scan
 * meanwhile.  SLAMHIP_FUSED_WAIT_UPDATES=1 restores "return after the updates". */
int32_t slamhip_cs_search_and_update(slamhip_cs *cs, const float search_pose[3], float hole_width,
                                     int32_t quality, int32_t max_obstacle_hits, float out_pose[3],
                                     int32_t *out_dist, int32_t *out_index);
/* slamhip_cs_set_scan + slamhip_cs_search_and_update in ONE call -- a scan of CoreSLAMProcessor.Update from :723 to :751 (the candidate list
 * is set or generated before it: it does not depend on the scan).  Same results as the two calls.  What the one call can do that the
 * two cannot: put the search launch into the stream BEFORE the scan's tables are made (the launch's parameters do not depend on
 * them) and let it wait for them on the device, so that the device does not idle between the previous scan's map update and this
 * search while the host sorts (DESIGN.md sec.4 "The per-scan flow" 5; slamhip_cs_prelaunch_stats counts how often; SLAMHIP_PRELAUNCH=0:
 * never).  slamhip_csproc_update is built on it.  xy: n_points (x, y) pairs in the robot frame, valid during the call. */
int32_t slamhip_cs_scan_search_and_update(slamhip_cs *cs, const float *xy, int32_t n_points, const float search_pose[3], float hole_width,
                                          int32_t quality, int32_t max_obstacle_hits, float out_pose[3], int32_t *out_dist,
                                          int32_t *out_index);
/* The same scan with the CALLER's trigonometry, end to end: the reference forms c = MathF.Cos(theta) * Scale, s = MathF.Sin(theta) * Scale
 * with the platform CRT, in CalculateDistanceSISD (:232-235) and again for the two map updates of the winner (:499-502 at the
 * HoleMap's scale, :545-548 at the ObstacleMap's -- from the pose AFTER NormalizeAngle :746, whose float arithmetic moves theta by
 * up to an ulp of 2 pi even inside (-pi, pi]), and a .NET host that wants results IDENTICAL to its own MathF -- not to this
 * library's deterministic routine -- hands in what it computed.  Two forms:
 *   slamhip_cs_search_and_update_pxcs: the K candidates as K x 4 (px, py, c, s) at the HoleMap's scale for the search (flat order:
 *     index 0 is the un-jittered search pose, then MonteCarloSearch's draws thread by thread, :626-649), and the K candidates'
 *     rows for the updates -- from the NORMALISED pose -- at the HoleMap's and the ObstacleMap's scale (NULL: no ObstacleMap update).
 *     The call evaluates every candidate, keeps the first strict minimum in flat order (:644-648, :698-705; index 0 when no point
 *     of any candidate lies in the map, :257), and draws both updates from row `index` of the caller's update arrays (only that
 *     row is read: the arrays stay on the host);
 *   slamhip_cs_distance_pxcs, then slamhip_cs_update_maps_pxcs with the winner's two rows: for a host that would rather form the
 *     update rows of ONE pose after the search than of all K before it (what the C# shim's TrigMode.Host does).
 * The pose itself never crosses the boundary: the caller knows candidate `index`.  Blocking; every integer output -- distances,
 * index, both maps -- is bit-exact with the C# on any platform BY CONSTRUCTION. */
int32_t slamhip_cs_search_and_update_pxcs(slamhip_cs *cs, const float *pxcs_search, const float *pxcs_update_hole,
                                          const float *pxcs_update_obstacle, int32_t K,
                                          float hole_width, int32_t quality, int32_t max_obstacle_hits,
                                          int32_t *out_index, int32_t *out_dist);
/* UpdateHoleMap (:750) and UpdateObstacleMap (:751) of one pose given as its (px, py, c, s) at either scale (pxcs_obstacle NULL:
 * the HoleMap only): both enqueued, one wait. */
int32_t slamhip_cs_update_maps_pxcs(slamhip_cs *cs, const float pxcs_hole[4], const float pxcs_obstacle[4], float hole_width,
                                    int32_t quality, int32_t max_obstacle_hits);

/* ------------------------------------------------------------------------------------------------
 * CoreSLAM, processor level (host-side orchestration in C++, mirrors the public C# class)
 * ---------------------------------------------------------------------------------------------- */

/* new CoreSLAMProcessor(physicalMapSize, holeMapSize, obstacleMapSize, startPose, sigmaXY, sigmaTheta,
 * iterationsPerThread, numSearchThreads)  (CoreSLAMProcessor.cs:119-162).  The candidate list has
 * max(numSearchThreads,1) * iterationsPerThread jitters; it is device-generated from (seed, scan number)
 * unless slamhip_csproc_set_offsets pins it. */
int32_t slamhip_csproc_create(slamhip_ctx *ctx, float physical_map_size, int32_t hole_map_size,
                              int32_t obstacle_map_size, const float start_pose[3], float sigma_xy,
                              float sigma_theta, int32_t iterations_per_thread, int32_t num_search_threads,
                              slamhip_csproc **out);
int32_t slamhip_csproc_destroy(slamhip_csproc *p);                 /* Dispose :757-773 */
int32_t slamhip_csproc_reset(slamhip_csproc *p);                   /* Reset :167-175 */
/* Update(List<ScanSegment>) (:717-752).  Segment i has pose seg_poses[3i..3i+2] and rays
 * [seg_start[i], seg_start[i+1]) of (angle, radius) pairs (BaseSLAM/ScanSegment.cs, Ray.cs). */
int32_t slamhip_csproc_update(slamhip_csproc *p, const float *seg_poses, const int32_t *seg_start,
                              int32_t n_segments, const float *rays);
int32_t slamhip_csproc_get_pose(slamhip_csproc *p, float out_pose[3]);                /* Pose :106 */
/* Quality :80, HoleWidth :85, PositionSearchBeginning :90, UnmappedObstacleHits :96, MaxObstacleHits :101 */
int32_t slamhip_csproc_set_params(slamhip_csproc *p, int32_t quality, float hole_width,
                                  int32_t position_search_beginning, int32_t unmapped_obstacle_hits,
                                  int32_t max_obstacle_hits);
int32_t slamhip_csproc_set_seed(slamhip_csproc *p, uint64_t seed);
/* candidates per scan from slamhip_cs_generate_offsets_lattice instead of slamhip_cs_generate_offsets (default: off) */
int32_t slamhip_csproc_set_lattice(slamhip_csproc *p, int32_t on);
/* pin the jitter list used by the next searching Update (parity tests feed the oracle the same list) */
int32_t slamhip_csproc_set_offsets(slamhip_csproc *p, const float *offs, int32_t n);
/* the underlying operator-level object (HoleMap / ObstacleMap properties :45,:50) */
int32_t slamhip_csproc_cs(slamhip_csproc *p, slamhip_cs **out_cs);
/* ScanSegmentsToCloud (CoreSLAMProcessor.cs:187-207) on its own, as slamhip_csproc_update runs it on the host: every segment's pose relative to
 * the LAST segment's (the odometry pose, :719), every ray (angle, radius) -> pose.X + radius * cos(angle + pose.Z), pose.Y + radius *
 * sin(angle + pose.Z) with the library's deterministic cos / sin (the correctly rounded float; a C# host that wants its own MathF
 * forms the cloud itself).  seg_poses: n_seg x 3, seg_start: n_seg + 1 ray offsets, rays: (angle, radius) pairs, out_xy: one (x, y)
 * per ray.  No device is involved. */
int32_t slamhip_scan_segments_to_cloud(const float *seg_poses, const int32_t *seg_start, int32_t n_seg, const float *rays, float *out_xy);

/* ------------------------------------------------------------------------------------------------
 * HectorSLAM, operator level
 * ---------------------------------------------------------------------------------------------- */

/* new MapRepMultiMap(mapResolution, mapSize, numDepth, Vector2.Zero) (HectorSLAM/Main/MapRepMultiMap.cs:40-58;
 * OccGridMap ctor Map/OccGridMap.cs:35-48; GridMap ctor Map/GridMap.cs:33-51): level i has
 * (w >> i, h >> i) cells of cell_length * 2^i metres; levels are independent maps. */
int32_t slamhip_hs_create(slamhip_ctx *ctx, float cell_length, int32_t width, int32_t height, int32_t levels,
                          slamhip_hs **out);
int32_t slamhip_hs_destroy(slamhip_hs *hs);
int32_t slamhip_hs_reset(slamhip_hs *hs);                                   /* MapRepMultiMap.Reset :63-66; also resets the probabilities (deviation D5, slamhip_hs_probability) */
int32_t slamhip_hs_level_info(slamhip_hs *hs, int32_t level, int32_t *width, int32_t *height, float *cell_length);
/* SetUpdateFactorFree / SetUpdateFactorOccupied (:83-95; OccGridMap.cs:58-79) */
int32_t slamhip_hs_set_factors(slamhip_hs *hs, float update_free_factor, float update_occupied_factor);
/* OccGridMap.EstimateIterations per level (OccGridMap.cs:53; default 3) */
int32_t slamhip_hs_set_iterations(slamhip_hs *hs, const int32_t *iterations_per_level);
/* mapArray of one level (GridMap.cs:13): n_cells = width*height LogOddsCell structs */
int32_t slamhip_hs_cells_upload(slamhip_hs *hs, int32_t level, const slamhip_cell *cells, size_t n_cells);
int32_t slamhip_hs_cells_download(slamhip_hs *hs, int32_t level, slamhip_cell *cells, size_t n_cells);
/* GridMap.GetBitmapData (GridMap.cs:104-115) computed on the device */
int32_t slamhip_hs_bitmap_download(slamhip_hs *hs, int32_t level, uint8_t *out, size_t n_cells);
/* GridMap.GetMapExtends (GridMap.cs:147-207) reduced on the device: extends = {xMax, yMax, xMin, yMin} of the cells with
 * Value != 0, *found = 1; or all zeros and *found = 0 (also when a minimum never left the reference's start value 10000) */
int32_t slamhip_hs_map_extends(slamhip_hs *hs, int32_t level, int32_t extends[4], int32_t *found);
/* Replica check, as slamhip_cs_maps_checksum: out[0] over the level's log-odds (OccGridCell.Value, OccGridCell.cs, as its
 * binary32 bit pattern), out[1] over its update indices (OccGridCell.UpdateIndex as uint32) */
int32_t slamhip_hs_checksum(slamhip_hs *hs, int32_t level, uint64_t out[2]);
/* OccGridMap.GetCachedProbability (OccGridMap.cs:97-107) for a list of cell indices: exp(v)/(exp(v)+1) of each cell's
 * CURRENT value.  Deviation D5: the reference's cache is not invalidated by Reset (OccGridMap.cs:244-252 resets
 * currCacheIndex but not cacheArray[i].Index, which only the constructor sets to -1, :38-42), so a cell cached in epoch e
 * before a Reset is served its pre-reset probability in epoch e after it; here -- and in the matcher, which reads the same
 * grid -- a probability never outlives the value it was computed from. */
int32_t slamhip_hs_probability(slamhip_hs *hs, int32_t level, const int32_t *indices, int32_t n, float *out);

/* The ScanCloud handed to MatchData / UpdateByScan: points + scan.Pose.xy (ScanCloud.cs:15-20) */
int32_t slamhip_hs_set_scan(slamhip_hs *hs, const float *xy, int32_t n_points, const float scan_origin[2]);

/* ScanMatcher.MatchData(MapRepMultiMap, scan, hintPose) (HectorSLAM/Matcher/ScanMatcher.cs:41-54):
 * all levels x iterations in ONE persistent launch, 3x3 solve on the device. */
int32_t slamhip_hs_match(slamhip_hs *hs, const float hint_pose[3], float out_pose[3]);
/* ScanMatcher.MatchData(OccGridMap, scan, hintPose) (:64-84) on one level */
int32_t slamhip_hs_match_level(slamhip_hs *hs, int32_t level, const float hint_pose[3], int32_t iterations,
                               float out_pose[3]);
/* B independent hints against the same scan and maps in one launch (throughput form) */
int32_t slamhip_hs_match_batch(slamhip_hs *hs, const float *hint_poses, int32_t B, float *out_poses);
/* GetCompleteHessianDerivs (:135-204) at a map-coordinate pose: H row-major 3x3, dTr 3 */
int32_t slamhip_hs_hessian(slamhip_hs *hs, int32_t level, const float pose_map[3], float H[9], float dTr[3]);

/* MapRepMultiMap.UpdateByScan -> OccGridMap.UpdateByScan on every level (MapRepMultiMap.cs:73-77;
 * OccGridMap.cs:114-239), all levels in one launch sequence. */
int32_t slamhip_hs_update_by_scan(slamhip_hs *hs, const float robot_pose_world[3]);

/* ------------------------------------------------------------------------------------------------
 * HectorSLAM, processor level
 * ---------------------------------------------------------------------------------------------- */
/* new HectorSLAMProcessor(mapResolution, mapSize, startPose, numDepth, numThreads) (Main/HectorSLAMProcessor.cs:66-77) */
int32_t slamhip_hsproc_create(slamhip_ctx *ctx, float map_resolution, int32_t width, int32_t height,
                              const float start_pose[3], int32_t num_depth, slamhip_hsproc **out);
int32_t slamhip_hsproc_destroy(slamhip_hsproc *p);
int32_t slamhip_hsproc_reset(slamhip_hsproc *p);                                        /* :131-138 */
/* Update(scan, poseHintWorld, mapWithoutMatching) (:86-126); *out_map_updated = return value.  The match is waited
 * for (its pose gates the update); the grid update is enqueued and the call returns -- later calls that touch the
 * pyramid are ordered behind it on the operator's stream; UpdateTiming (:115) then reports the enqueue.
 * SLAMHIP_HS_WAIT_UPDATE=1 waits for the update.  While consecutive scans keep updating the map, the update is enqueued
 * behind the match BEFORE the pose is back: the kernel reads the matched pose from device memory and applies the test of
 * :107-109 itself (the same float operations the host then applies to the pose it receives); SLAMHIP_HS_NO_GATED_UPDATE=1
 * keeps the decision on the host. */
int32_t slamhip_hsproc_update(slamhip_hsproc *p, const float *xy, int32_t n_points, const float scan_origin[2],
                              const float pose_hint_world[3], int32_t map_without_matching,
                              int32_t *out_map_updated);
int32_t slamhip_hsproc_get(slamhip_hsproc *p, float match_pose[3], float last_map_update_pose[3],
                           float *match_timing_ms, float *update_timing_ms);            /* :31-46 */
/* MinDistanceDiffForMapUpdate :51, MinAngleDiffForMapUpdate :56 */
int32_t slamhip_hsproc_set_thresholds(slamhip_hsproc *p, float min_distance_diff, float min_angle_diff);
int32_t slamhip_hsproc_hs(slamhip_hsproc *p, slamhip_hs **out_hs);                       /* MapRep :26 */

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU (new; no reference counterpart: the reference's only parallelism is ParallelWorker threads)
 * ---------------------------------------------------------------------------------------------- */
/* One process, n GPUs: a context + CoreSLAM replica per device and one RCCL communicator. */
int32_t slamhip_group_create(const int32_t *device_ordinals, int32_t n, float physical_map_size,
                             int32_t hole_map_size, int32_t obstacle_map_size, slamhip_group **out);
int32_t slamhip_group_destroy(slamhip_group *g);
int32_t slamhip_group_size(slamhip_group *g, int32_t *out_n);
int32_t slamhip_group_cs(slamhip_group *g, int32_t rank, slamhip_cs **out_cs);
/* broadcast-by-replication helpers: apply the same call to every replica */
int32_t slamhip_group_reset(slamhip_group *g, int32_t unmapped_obstacle_hits);
int32_t slamhip_group_holemap_upload(slamhip_group *g, const uint16_t *pixels, size_t n_pixels);
int32_t slamhip_group_set_scan(slamhip_group *g, const float *xy, int32_t n_points);
int32_t slamhip_group_set_offsets(slamhip_group *g, const float *offs, int32_t n);
/* ... or generated on every GPU of the group (slamhip_cs_generate_offsets: the same list everywhere, keyed by seed, stream and index) */
int32_t slamhip_group_generate_offsets(slamhip_group *g, int32_t n, float sigma_xy, float sigma_theta, uint64_t seed, uint64_t stream);
/* Candidates block-sharded over the GPUs, one ncclAllReduce(min, uint64, count 1) of the packed key
 * over xGMI, winner pose recomputed locally. */
int32_t slamhip_group_search(slamhip_group *g, const float search_pose[3], float out_pose[3],
                             int32_t *out_dist, int32_t *out_index);
/* replicas apply the identical deterministic update (integer-exact kernels keep them bit-identical) */
/* One scan on every GPU of the group in one call (CoreSLAMProcessor.cs:732, :695-705, :750-751): search over the GPU's block,
 * ncclAllReduce(min), the winner decoded on each device, each replica's map updates queued behind -- as
 * slamhip_cs_search_allreduce_and_update, every rank on its own worker thread; returns with key and pose (theta normalised). */
int32_t slamhip_group_search_and_update(slamhip_group *g, const float search_pose[3], float hole_width, int32_t quality,
                                        int32_t max_obstacle_hits, float out_pose[3], int32_t *out_dist, int32_t *out_index);
int32_t slamhip_group_update_maps(slamhip_group *g, const float pose[3], float hole_width, int32_t quality,
                                  int32_t max_obstacle_hits);
/* *out_equal = 1 when slamhip_cs_maps_checksum agrees on every GPU of the group */
int32_t slamhip_group_replicas_equal(slamhip_group *g, int32_t *out_equal);


/* One process per GPU (torch.distributed.run, MPI, ...): this rank's end of an RCCL communicator.  The host framework
 * only carries the 128-byte id from rank 0 to the other ranks; the per-scan exchange -- the cross-thread arg-min of
 * CoreSLAMProcessor.cs:695-705 as ncclAllReduce(min, uint64, count 1) over xGMI -- is issued by the library on the
 * communicator's own stream, behind an event, so that the next search does not wait for the last collective. */
int32_t slamhip_comm_probe(void);                       /* every rank, before anything collective: can librccl be resolved here? */
int32_t slamhip_comm_unique_id(uint8_t out_id[128]);                                  /* rank 0 */
int32_t slamhip_comm_create(slamhip_ctx *ctx, const uint8_t id[128], int32_t rank, int32_t n_ranks, slamhip_comm **out);
int32_t slamhip_comm_destroy(slamhip_comm *comm);
int32_t slamhip_comm_info(slamhip_comm *comm, int32_t *out_rank, int32_t *out_n_ranks);
/* One sharded search step (asynchronous): flat candidates [first, first+count) on this rank; the keys of up to 16 consecutive
 * steps are min-all-reduced in one collective (every rank must issue the same steps and call slamhip_comm_wait at the same
 * places).  *d_out_key (optional) = device address where this step's reduced key will be once the collective of its batch has
 * run -- after slamhip_comm_wait, or behind a later batch on the communicator's stream -- valid for 64 further steps. */
int32_t slamhip_cs_search_allreduce_async(slamhip_cs *cs, slamhip_comm *comm, const float search_pose[3], int32_t first,
                                          int32_t count, uint64_t **d_out_key);
/* Issues the collective for the steps not yet covered by one, waits for every step issued so far; *out_key (optional) = the
 * reduced key of the last one.  (A host that needs every scan's winner before the next scan calls it after every step.) */
int32_t slamhip_comm_wait(slamhip_comm *comm, uint64_t *out_key);
/* Steps per collective of the asynchronous form (1 .. 32; default 16).  Waits for the steps issued so far; every rank calls it
 * at the same place. */
int32_t slamhip_comm_set_batch(slamhip_comm *comm, int32_t steps);
/* One sharded search step, BLOCKING -- the per-scan form: CoreSLAMProcessor.Update needs the winner (CoreSLAMProcessor.cs:732,
 * the arg-min of :695-705) before it updates the maps (:750-751).  K1 over this rank's block, ncclAllReduce(min, uint64, 1)
 * and the hand-over of the reduced key to the host sit on the operator's stream, one behind the other (no second stream, no
 * event).  *out_key = min over all ranks of (distance << 32 | flat index).  Every rank makes the same call. */
int32_t slamhip_cs_search_allreduce(slamhip_cs *cs, slamhip_comm *comm, const float search_pose[3], int32_t first,
                                    int32_t count, uint64_t *out_key);
/* One scan of the SLAM loop on every rank -- search (CoreSLAMProcessor.cs:732), exchange (:695-705 as ncclAllReduce(min, uint64, 1)),
 * both map updates from the winner's pose (:750-751) -- with NO host hop between the exchange and the updates: a one-thread launch
 * decodes the reduced key into the pose on the device (every rank holds the whole jitter list), the replicas' map updates are
 * enqueued behind it, and the call returns when key and pose have reached the host; the updates run on, and everything that touches
 * the maps afterwards is ordered behind them (as slamhip_cs_search_and_update).  out_pose: theta normalised (:746).  Every rank
 * makes the same call; a rank whose own part fails still joins the collective (with the neutral key) and reports afterwards. */
int32_t slamhip_cs_search_allreduce_and_update(slamhip_cs *cs, slamhip_comm *comm, const float search_pose[3], int32_t first,
                                               int32_t count, float hole_width, int32_t quality, int32_t max_obstacle_hits,
                                               float out_pose[3], int32_t *out_dist, int32_t *out_index);
/* Latency of the exchange step alone: `iters` 8-byte min all-reduces back to back; *out_us = device microseconds per
 * collective.  Every rank makes the same call (measurement aid for the scaling curve). */
int32_t slamhip_comm_allreduce_probe(slamhip_comm *comm, int32_t iters, float *out_us);
/* Replica check across the ranks: every rank checksums its maps (slamhip_cs_maps_checksum), one ncclAllReduce(min) and one
 * ncclAllReduce(max) of the two words; *out_equal = 1 when they agree, i.e. every rank holds bit-identical maps.  Every rank
 * makes the same call (a debug / health check: once per so many scans, not per scan). */
int32_t slamhip_comm_replicas_equal(slamhip_cs *cs, slamhip_comm *comm, int32_t *out_equal);

#ifdef __cplusplus
}
#endif
#endif /* SLAMHIP_H */
