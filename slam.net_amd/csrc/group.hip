// group.hip -- single-process multi-GPU CoreSLAM search: candidates block-sharded over the GPUs, one RCCL
// min all-reduce of the packed (distance << 32 | flat index) key over xGMI per scan.
//
// New design with no reference counterpart (the reference's only parallelism is ParallelWorker threads,
// BaseSLAM/ParallelWorker.cs:15-141): the cross-thread arg-min of CoreSLAMProcessor.cs:695-705 becomes
// ncclAllReduce(min, uint64, count = 1).  The 8-byte message makes the collective latency-bound; every rank
// then holds the winning key and recomputes the winning pose locally, and map updates are replicated
// (bit-exact integer kernels keep the replicas identical).
// RCCL is resolved with dlopen at group creation so that single-GPU users of libslamhip never load it
// (and a host process that already loaded its own librccl, e.g. PyTorch, shares that copy).
#include "cs_internal.h"
#include "det_trig.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <vector>
#include <string>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>

struct rccl_api {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};

static int32_t load_rccl(rccl_api *api)
{
    // The librccl that belongs to the HIP runtime this library is bound to -- the one in the same directory -- first.  A process
    // may hold two installations (PyTorch wheels bring their own libamdhip64 / libhsa-runtime64 / librccl, and whichever of
    // libslamhip and torch is loaded first decides which HIP runtime both run on): a librccl from the other installation opens
    // its own, uninitialised HSA runtime and fails in ncclCommInitRank ("no ROCm-capable device is detected"; seen when a test
    // loaded libslamhip before it imported torch).
    api->lib = nullptr;
    {
        Dl_info info;
        if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char *nm : { "librccl.so.1", "librccl.so" }) {
                    api->lib = dlopen((dir + nm).c_str(), RTLD_NOW | RTLD_LOCAL);
                    if (api->lib) break;
                }
            }
        }
    }
    const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char *nm : names) { if (api->lib) break; api->lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); }
    if (!api->lib) SH_FAIL(SLAMHIP_ERR_RCCL, "cannot load librccl: %s", dlerror());
#define SH_SYM(field, name) do { *(void **)(&api->field) = dlsym(api->lib, name); \
        if (!api->field) SH_FAIL(SLAMHIP_ERR_RCCL, "librccl lacks %s", name); } while (0)
    SH_SYM(GetUniqueId, "ncclGetUniqueId");
    SH_SYM(CommInitRank, "ncclCommInitRank");
    SH_SYM(CommInitAll, "ncclCommInitAll");
    SH_SYM(CommDestroy, "ncclCommDestroy");
    SH_SYM(AllReduce, "ncclAllReduce");
    SH_SYM(GroupStart, "ncclGroupStart");
    SH_SYM(GroupEnd, "ncclGroupEnd");
    SH_SYM(GetErrorString, "ncclGetErrorString");
#undef SH_SYM
    return SLAMHIP_OK;
}

#define SH_NCCL(g, expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
        slamhip_set_error("%s failed: %s", #expr, (g)->api.GetErrorString(r_)); return SLAMHIP_ERR_RCCL; } } while (0)

// One worker thread per GPU of a group: a job (the enqueue of a rank's search + its end of the collective + the wait for its
// stream; a rank's map update) runs on every rank AT THE SAME TIME instead of one rank after the other from the caller's thread --
// a launch costs the host ~5 us, the serial loop made an eight-GPU search step 8 x (enqueue + synchronise) long.  The workers
// bind their device once (the HIP device is per thread).  One-GPU groups run their jobs inline (SLAMHIP_GROUP_THREADS=1 forces
// the worker: the tests do, so that the machinery is exercised on a one-GPU box).
struct sh_group_workers {
    int n = 0;
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;                       // job generation: a worker runs the job once per increment
    int pending = 0;
    bool quit = false;
    std::function<int32_t(int)> job;
    std::vector<int32_t> rc;
    std::vector<std::string> err;
    void start(int n_, const std::vector<int> &devices)
    {
        n = n_; rc.assign((size_t)n, SLAMHIP_OK); err.assign((size_t)n, std::string());
        for (int r = 0; r < n; r++)
            th.emplace_back([this, r, devices]() {
                (void)hipSetDevice(devices[(size_t)r]);
                uint64_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(m);
                    cv_go.wait(lk, [&] { return quit || gen != seen; });
                    if (quit) return;
                    seen = gen;
                    lk.unlock();
                    const int32_t v = job(r);
                    std::string e = v != SLAMHIP_OK ? std::string(slamhip_last_error()) : std::string();
                    lk.lock();
                    rc[(size_t)r] = v; err[(size_t)r] = e;
                    if (--pending == 0) cv_done.notify_all();
                }
            });
    }
    // runs fn(rank) on every rank's worker, returns the first failure (its message becomes the caller's last error)
    int32_t run(const std::function<int32_t(int)> &fn)
    {
        {
            std::unique_lock<std::mutex> lk(m);
            job = fn; pending = n; gen++;
            cv_go.notify_all();
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        for (int r = 0; r < n; r++)
            if (rc[(size_t)r] != SLAMHIP_OK) { slamhip_set_error("rank %d: %s", r, err[(size_t)r].c_str()); return rc[(size_t)r]; }
        return SLAMHIP_OK;
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) if (t.joinable()) t.join();
        th.clear();
    }
};

struct slamhip_group {
    int n;
    rccl_api api;
    std::vector<slamhip_ctx *> ctx;
    std::vector<slamhip_cs *> cs;
    std::vector<ncclComm_t> comm;
    std::vector<uint64_t *> d_key;
    int n_offs;
    sh_group_workers *workers;              // null: jobs run inline on the caller's thread (one-GPU groups)
    uint64_t h_key;
};

// fn(rank) on every rank: in parallel on the workers, or inline
static int32_t group_run(slamhip_group *g, const std::function<int32_t(int)> &fn)
{
    if (g->workers) return g->workers->run(fn);
    for (int r = 0; r < g->n; r++) { SH_HIP(hipSetDevice(g->ctx[r]->device)); SH_TRY(fn(r)); }
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_group_destroy(slamhip_group *g)
{
    if (!g) return SLAMHIP_OK;
    if (g->workers) { g->workers->stop(); delete g->workers; g->workers = nullptr; }
    for (int r = 0; r < (int)g->comm.size(); r++) if (g->comm[r]) g->api.CommDestroy(g->comm[r]);
    for (int r = 0; r < (int)g->cs.size(); r++) {
        if (g->d_key.size() > (size_t)r && g->d_key[r]) { (void)hipSetDevice(g->ctx[r]->device); (void)hipFree(g->d_key[r]); }
        slamhip_cs_destroy(g->cs[r]);
    }
    for (auto c : g->ctx) slamhip_ctx_destroy(c);
    if (g->api.lib) dlclose(g->api.lib);
    delete g;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_group_create(const int32_t *devices, int32_t n, float physical, int32_t hole_size, int32_t obst_size,
                                        slamhip_group **out)
{
    SH_CHECK_ARG(devices && n >= 1 && n <= 64 && out);
    slamhip_group *g = new slamhip_group();
    g->n = n; g->n_offs = 0; g->workers = nullptr; g->h_key = 0;
    int32_t rc = load_rccl(&g->api);
    for (int r = 0; r < n && rc == SLAMHIP_OK; r++) {
        slamhip_ctx *c = nullptr; slamhip_cs *cs = nullptr; uint64_t *dk = nullptr;
        rc = slamhip_ctx_create(devices[r], &c);
        if (rc == SLAMHIP_OK) { g->ctx.push_back(c); rc = slamhip_cs_create(c, physical, hole_size, obst_size, &cs); }
        if (rc == SLAMHIP_OK) { g->cs.push_back(cs); if (hipMalloc(&dk, sizeof(uint64_t)) != hipSuccess) { slamhip_set_error("hipMalloc failed"); rc = SLAMHIP_ERR_NOMEM; } }
        if (rc == SLAMHIP_OK) g->d_key.push_back(dk);
    }
    if (rc == SLAMHIP_OK) {
        g->comm.assign((size_t)n, nullptr);
        std::vector<int> devs(devices, devices + n);
        ncclResult_t r_ = g->api.CommInitAll(g->comm.data(), n, devs.data());
        if (r_ != ncclSuccess) { slamhip_set_error("ncclCommInitAll failed: %s", g->api.GetErrorString(r_)); rc = SLAMHIP_ERR_RCCL; }
    }
    if (rc == SLAMHIP_OK && (n > 1 || getenv("SLAMHIP_GROUP_THREADS"))) {
        g->workers = new sh_group_workers();
        g->workers->start(n, std::vector<int>(devices, devices + n));
    }
    if (rc != SLAMHIP_OK) { slamhip_group_destroy(g); return rc; }
    *out = g;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_group_size(slamhip_group *g, int32_t *out)
{
    SH_CHECK_ARG(g && out);
    *out = g->n;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_group_cs(slamhip_group *g, int32_t rank, slamhip_cs **out)
{
    SH_CHECK_ARG(g && out && rank >= 0 && rank < g->n);
    *out = g->cs[rank];
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_group_reset(slamhip_group *g, int32_t unmapped)
{
    SH_CHECK_ARG(g);
    for (int r = 0; r < g->n; r++) SH_TRY(slamhip_cs_reset(g->cs[r], unmapped));
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_group_holemap_upload(slamhip_group *g, const uint16_t *pix, size_t n)
{
    SH_CHECK_ARG(g);
    return group_run(g, [g, pix, n](int r) -> int32_t { return slamhip_cs_holemap_upload(g->cs[r], pix, n); });
}

extern "C" int32_t slamhip_group_set_scan(slamhip_group *g, const float *xy, int32_t n)
{
    SH_CHECK_ARG(g);
    return group_run(g, [g, xy, n](int r) -> int32_t { return slamhip_cs_set_scan(g->cs[r], xy, n); });   // (per scan: every rank sorts and stages at the same time)
}

extern "C" int32_t slamhip_group_set_offsets(slamhip_group *g, const float *offs, int32_t n)
{
    SH_CHECK_ARG(g);
    for (int r = 0; r < g->n; r++) SH_TRY(slamhip_cs_set_offsets(g->cs[r], offs, n));
    g->n_offs = n;
    return SLAMHIP_OK;
}

// the generator is keyed by (seed, stream, index): every replica makes the very same list, and a rank's block of it is the same
// candidates whatever the number of ranks -- no list crosses the host
extern "C" int32_t slamhip_group_generate_offsets(slamhip_group *g, int32_t n, float sigma_xy, float sigma_theta, uint64_t seed, uint64_t stream)
{
    SH_CHECK_ARG(g && n >= 0);
    for (int r = 0; r < g->n; r++) SH_TRY(slamhip_cs_generate_offsets(g->cs[r], n, sigma_xy, sigma_theta, seed, stream));
    g->n_offs = n;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_group_search(slamhip_group *g, const float pose[3], float out_pose[3], int32_t *out_dist, int32_t *out_index)
{
    SH_CHECK_ARG(g && pose);
    const int K = g->n_offs + 1;                 // flat candidates, 0 = un-jittered pose
    const float p3[3] = { pose[0], pose[1], pose[2] };
    // contiguous blocks keep the flat (thread-major) indices, so the packed-key min reproduces the
    // reference tie-break across GPUs exactly as across threads (:695-705).
    // Every rank, on its own worker thread and its own communicator: the search over its block, its end of
    // ncclAllReduce(min, uint64, 1) behind it on the same stream, the wait for the stream.  A rank whose search cannot be
    // enqueued still joins the collective -- with the neutral key -- and reports afterwards: the others must not hang in it.
    const int32_t rc = group_run(g, [g, K, p3](int r) -> int32_t {
        const int first = (int)((long long)K * r / g->n), last = (int)((long long)K * (r + 1) / g->n);
        hipStream_t st = g->ctx[r]->stream;
        int32_t rc_local = SLAMHIP_OK;
        std::string err;
        if (last > first) {
            rc_local = slamhip_cs_search_shard_async(g->cs[r], p3, first, last - first, g->d_key[r]);
            if (rc_local != SLAMHIP_OK) err = slamhip_last_error();
        }
        if (last <= first || rc_local != SLAMHIP_OK) (void)hipMemsetAsync(g->d_key[r], 0xFF, sizeof(uint64_t), st);
        const ncclResult_t r_ = g->api.AllReduce(g->d_key[r], g->d_key[r], 1, ncclUint64, ncclMin, g->comm[r], st);
        if (r == 0) (void)hipMemcpyAsync(&g->h_key, g->d_key[0], sizeof(uint64_t), hipMemcpyDeviceToHost, st);
        const hipError_t e = hipStreamSynchronize(st);
        if (rc_local != SLAMHIP_OK) { slamhip_set_error("%s", err.c_str()); return rc_local; }
        if (r_ != ncclSuccess) { slamhip_set_error("ncclAllReduce failed: %s", g->api.GetErrorString(r_)); return SLAMHIP_ERR_RCCL; }
        SH_HIP(e);
        return SLAMHIP_OK;
    });
    SH_TRY(rc);
    SH_HIP(hipSetDevice(g->ctx[0]->device));
    return slamhip_cs_pose_from_key(g->cs[0], pose, g->h_key, out_pose, out_dist, out_index);
}

extern "C" int32_t slamhip_group_update_maps(slamhip_group *g, const float pose[3], float hole_width, int32_t quality, int32_t max_hits)
{
    SH_CHECK_ARG(g && pose);
    const float p3[3] = { pose[0], pose[1], pose[2] };
    // the replicas update concurrently: every rank's worker enqueues on its GPU's stream and waits for it
    return group_run(g, [g, p3, hole_width, quality, max_hits](int r) -> int32_t {
        SH_TRY(cs_update_maps_enqueue(g->cs[r], p3, hole_width, quality, max_hits));
        return cs_update_maps_finish(g->cs[r]);
    });
}

// Replica check: the replicas' maps after identical updates must be bit-identical; *out_equal = 1 when the
// checksums of HoleMap and ObstacleMap agree on every GPU of the group.
extern "C" int32_t slamhip_group_replicas_equal(slamhip_group *g, int32_t *out_equal)
{
    SH_CHECK_ARG(g && out_equal);
    uint64_t first[2] = { 0, 0 };
    int equal = 1;
    for (int r = 0; r < g->n; r++) {
        uint64_t c[2];
        SH_TRY(slamhip_cs_maps_checksum(g->cs[r], c));
        if (r == 0) { first[0] = c[0]; first[1] = c[1]; }
        else if (c[0] != first[0] || c[1] != first[1]) equal = 0;
    }
    *out_equal = equal;
    return SLAMHIP_OK;
}

// ---- one process per GPU: this rank's end of an RCCL communicator -------------------------------------------------
// The host framework (torch.distributed.run, MPI, ...) only carries the 128-byte RCCL id from rank 0 to the others;
// the per-scan exchange is issued by the library itself: K1 over this rank's block of the flat candidate list on the
// context's stream, then ncclAllReduce(min, uint64, 1) of the packed key on the communicator's OWN stream behind an
// event -- so the next search does not wait for the collective of the last one (a ring of key slots keeps them apart),
// and no interpreter or framework call sits between the kernel and the collective.
#define SH_COMM_SLOTS 64
#define SH_COMM_BLOCK 32                 // steps per completion event of the collectives' stream
#define SH_COMM_BATCH_MAX 32             // steps whose keys travel in one all-reduce (slamhip_comm::batch; divides SH_COMM_BLOCK)
struct slamhip_comm {
    slamhip_ctx *ctx;
    rccl_api api;
    ncclComm_t comm;
    int rank, n_ranks;
    hipStream_t stream;                            // the collectives' stream
    uint64_t *d_keys;                              // [SH_COMM_SLOTS] key slots (device)
    hipEvent_t ev_k1[SH_COMM_SLOTS], ev_ar[SH_COMM_SLOTS];
    bool ar_pending[SH_COMM_SLOTS];
    uint64_t step;
    int batch;                                     // steps per collective: the keys of `batch` consecutive steps are min-reduced in one call
    uint64_t flushed;                              // steps whose collective has been issued
    unsigned long long *d_sig;                     // HSA signal memory: the search kernel stores the step number, the collectives' stream waits for it
    bool by_value;
    uint64_t *h_key;                               // pinned
    uint64_t *d_sync_key;                          // the blocking step's key (slamhip_cs_search_allreduce)
    bool async_dirty;                              // asynchronous steps issued since the last slamhip_comm_wait
};

static_assert(sizeof(ncclUniqueId) == 128, "the id travels as 128 bytes");

// Can this process create a communicator at all (librccl resolvable, with the entry points used here)?  A collective
// ncclCommInitRank that only SOME ranks reach blocks the others for good, so hosts check this on every rank first.
extern "C" int32_t slamhip_comm_probe(void)
{
    rccl_api api;
    SH_TRY(load_rccl(&api));
    dlclose(api.lib);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_comm_unique_id(uint8_t out[128])
{
    SH_CHECK_ARG(out);
    rccl_api api;
    SH_TRY(load_rccl(&api));
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) { slamhip_set_error("ncclGetUniqueId failed: %s", api.GetErrorString(r)); dlclose(api.lib); return SLAMHIP_ERR_RCCL; }
    memcpy(out, &id, 128);
    dlclose(api.lib);                              // (reference counted: the communicator keeps its own handle)
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_comm_destroy(slamhip_comm *c)
{
    if (!c) return SLAMHIP_OK;
    (void)hipSetDevice(c->ctx->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) c->api.CommDestroy(c->comm);
    for (int i = 0; i < SH_COMM_SLOTS; i++) { if (c->ev_k1[i]) (void)hipEventDestroy(c->ev_k1[i]); if (c->ev_ar[i]) (void)hipEventDestroy(c->ev_ar[i]); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    (void)hipFree(c->d_keys); (void)hipFree(c->d_sync_key);
    if (c->d_sig) (void)hipFree(c->d_sig);
    if (c->h_key) (void)hipHostFree(c->h_key);
    if (c->api.lib) dlclose(c->api.lib);
    delete c;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_comm_create(slamhip_ctx *ctx, const uint8_t unique_id[128], int32_t rank, int32_t n_ranks, slamhip_comm **out)
{
    SH_CHECK_ARG(ctx && unique_id && out && n_ranks >= 1 && rank >= 0 && rank < n_ranks);
    SH_HIP(hipSetDevice(ctx->device));
    slamhip_comm *c = new slamhip_comm();
    c->ctx = ctx; c->rank = rank; c->n_ranks = n_ranks;
    {   // The collective is latency-bound on the device and costs the host ~15 us per call (stream wait + RCCL's enqueue path):
        // issued per step it makes a 21 us search step host-bound (25 us per step on one rank).  An asynchronous step therefore
        // only launches the search; every `batch` steps ONE all-reduce takes the keys of all of them (the slots of a batch are
        // consecutive), and slamhip_comm_wait flushes what is left.  Every rank issues the same steps and waits at the same
        // places (as any collective requires), so the batches agree.  SLAMHIP_COMM_BATCH=1: one collective per step.
        const char *b = getenv("SLAMHIP_COMM_BATCH");
        int v = b ? atoi(b) : 16;
        if (v < 1) v = 1;
        if (v > SH_COMM_BATCH_MAX) v = SH_COMM_BATCH_MAX;
        while (SH_COMM_BLOCK % v) v--;
        c->batch = v;
    }
    int32_t rc = load_rccl(&c->api);
    if (rc == SLAMHIP_OK) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&c->d_keys, sizeof(uint64_t) * SH_COMM_SLOTS);
        if (e == hipSuccess) e = hipMemset(c->d_keys, 0xFF, sizeof(uint64_t) * SH_COMM_SLOTS);
        if (e == hipSuccess) e = hipHostMalloc(&c->h_key, 64);
        if (e == hipSuccess) e = hipMalloc(&c->d_sync_key, 16);
        for (int i = 0; i < SH_COMM_SLOTS && e == hipSuccess; i++) {
            e = hipEventCreateWithFlags(&c->ev_k1[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ar[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) { slamhip_set_error("communicator resources: %s", hipGetErrorString(e)); rc = SLAMHIP_ERR_HIP; }
        // K1 -> collective dependency: an event pair costs the search stream ~9 us per step on this stack (measured); where
        // the device supports stream memory operations the kernel's last act is the step number into an HSA signal that the
        // collectives' stream waits for -- nothing is inserted into the search stream (SLAMHIP_COMM_DEP=event keeps events)
        // With the keys of a batch of steps in one collective the dependency is needed once per batch, and an event pair per
        // batch (~1 us per step at eight steps) beats the signal: the kernel's store into host-visible signal memory must
        // retire before the launch ends, +3 us per launch (measured on one rank: 25.1 us per step with the signal at one step
        // per collective, 26.3 at eight; 21.9 us without any collective).  SLAMHIP_COMM_DEP=signal keeps the signal.
        const char *dep = getenv("SLAMHIP_COMM_DEP");
        int can = 0;
        if (rc == SLAMHIP_OK && ((dep && strcmp(dep, "signal") == 0) || (c->batch == 1 && !(dep && strcmp(dep, "event") == 0))) &&
            hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ctx->device) == hipSuccess && can &&
            hipExtMallocWithFlags((void **)&c->d_sig, 8, hipMallocSignalMemory) == hipSuccess) {
            *c->d_sig = 0ull;                                     // (signal memory is host-visible)
            c->by_value = true;
        } else (void)hipGetLastError();
    }
    if (rc == SLAMHIP_OK) {
        ncclUniqueId id;
        memcpy(&id, unique_id, 128);
        const ncclResult_t r = c->api.CommInitRank(&c->comm, n_ranks, id, rank);
        if (r != ncclSuccess) { slamhip_set_error("ncclCommInitRank failed: %s", c->api.GetErrorString(r)); c->comm = nullptr; rc = SLAMHIP_ERR_RCCL; }
    }
    if (rc != SLAMHIP_OK) { slamhip_comm_destroy(c); return rc; }
    *out = c;
    return SLAMHIP_OK;
}

// The collective for the steps [flushed, step): behind the last of them on the collectives' stream, one min all-reduce over their
// (consecutive) key slots.
static int32_t comm_flush(slamhip_comm *c)
{
    if (c->flushed == c->step) return SLAMHIP_OK;
    const int first = (int)(c->flushed % SH_COMM_SLOTS), n = (int)(c->step - c->flushed);       // (a batch never wraps: batch divides the slot count)
    const int last = first + n - 1;
    if (c->by_value) SH_HIP(hipStreamWaitValue64(c->stream, c->d_sig, c->step, hipStreamWaitValueGte, ~0ull));
    else {                                                         // one event pair per batch, behind its last search
        SH_HIP(hipEventRecord(c->ev_k1[last], c->ctx->stream));
        SH_HIP(hipStreamWaitEvent(c->stream, c->ev_k1[last], 0));
    }
    SH_NCCL(c, c->api.AllReduce(c->d_keys + first, c->d_keys + first, (size_t)n, ncclUint64, ncclMin, c->comm, c->stream));
    if (last % SH_COMM_BLOCK == SH_COMM_BLOCK - 1) { SH_HIP(hipEventRecord(c->ev_ar[last / SH_COMM_BLOCK], c->stream)); c->ar_pending[last / SH_COMM_BLOCK] = true; }
    c->flushed = c->step;
    return SLAMHIP_OK;
}

// One sharded search step: returns at once; the reduced key of this step will be in *d_out_key (device memory owned by
// the communicator, valid until SH_COMM_SLOTS (64) further steps have been issued) once the collective of its batch has run:
// after slamhip_comm_wait, or behind a later step's batch on the collectives' stream.
extern "C" int32_t slamhip_cs_search_allreduce_async(slamhip_cs *cs, slamhip_comm *c, const float pose[3], int32_t first, int32_t count,
                                                     uint64_t **d_out_key)
{
    SH_CHECK_ARG(cs && c && pose && cs->ctx == c->ctx);
    SH_HIP(hipSetDevice(c->ctx->device));
    const int slot = (int)(c->step % SH_COMM_SLOTS);
    uint64_t *key = c->d_keys + slot;
    hipStream_t main = c->ctx->stream;
    // the slot's last collective (SH_COMM_SLOTS steps ago) must have read it: the collectives' stream records an event per
    // block of SH_COMM_BLOCK steps, and a block starts after a host-side wait -- normally a no-op -- for the event of the
    // block that used its slots last
    const int blk = (slot / SH_COMM_BLOCK);
    if (slot % SH_COMM_BLOCK == 0 && c->ar_pending[blk]) { SH_HIP(hipEventSynchronize(c->ev_ar[blk])); c->ar_pending[blk] = false; }
    bool signalled = false;
    if (count > 0) {
        if (c->by_value) { cs->k1_sig = c->d_sig; cs->k1_sig_val = c->step + 1; }
        const int32_t rc = slamhip_cs_search_shard_async(cs, pose, first, count, key);
        cs->k1_sig = nullptr;
        SH_TRY(rc);
        signalled = c->by_value && cs->k1_sig_armed;
    } else SH_HIP(hipMemsetAsync(key, 0xFF, sizeof(uint64_t), main));                                   // (a rank without candidates: the neutral key)
    if (c->by_value && !signalled) SH_HIP(hipStreamWriteValue64(main, c->d_sig, c->step + 1, 0));        // (fallback kernels, empty shard)
    c->step++;
    c->async_dirty = true;
    if (c->step % (uint64_t)c->batch == 0) SH_TRY(comm_flush(c));
    if (d_out_key) *d_out_key = key;
    return SLAMHIP_OK;
}

// Wait for every step issued so far; *out_key = the reduced key of the last one.
extern "C" int32_t slamhip_comm_wait(slamhip_comm *c, uint64_t *out_key)
{
    SH_CHECK_ARG(c);
    SH_HIP(hipSetDevice(c->ctx->device));
    SH_TRY(comm_flush(c));
    if (out_key) {
        if (c->step == 0) SH_FAIL(SLAMHIP_ERR_STATE, "no search step has been issued on this communicator");
        const int slot = (int)((c->step - 1) % SH_COMM_SLOTS);
        SH_HIP(hipMemcpyAsync(c->h_key, c->d_keys + slot, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    }
    SH_HIP(hipStreamSynchronize(c->stream));
    SH_HIP(hipStreamSynchronize(c->ctx->stream));
    c->async_dirty = false;
    if (out_key) *out_key = *c->h_key;
    return SLAMHIP_OK;
}

// Steps per collective of the asynchronous form (1 .. 32, rounded down to a divisor of 32; default 16 or SLAMHIP_COMM_BATCH).
// Waits for the steps issued so far first; every rank must make the same call at the same place.
extern "C" int32_t slamhip_comm_set_batch(slamhip_comm *c, int32_t steps)
{
    SH_CHECK_ARG(c && steps >= 1);
    SH_TRY(slamhip_comm_wait(c, nullptr));
    int v = steps > SH_COMM_BATCH_MAX ? SH_COMM_BATCH_MAX : steps;
    while (SH_COMM_BLOCK % v) v--;
    // (a batch never wraps the ring of key slots: restart the step count at a batch boundary -- every slot is idle after the wait)
    c->step = c->flushed = 0;
    for (int i = 0; i < SH_COMM_SLOTS; i++) c->ar_pending[i] = false;
    if (c->by_value) *c->d_sig = 0ull;
    c->batch = v;
    return SLAMHIP_OK;
}

// One sharded search step, BLOCKING: the per-scan form of the SLAM loop, which needs the winner before it can update the maps
// (CoreSLAMProcessor.cs:732 -> :750).  Nothing can overlap the exchange here, so everything sits on the operator's stream --
// K1 over this rank's block, ncclAllReduce(min, uint64, 1) behind it (no event, no second stream), then the publish launch that
// hands the reduced key to the host through the context's mailbox.  *out_key = the reduced key.  Every rank makes the same call.
extern "C" int32_t slamhip_cs_search_allreduce(slamhip_cs *cs, slamhip_comm *c, const float pose[3], int32_t first, int32_t count,
                                               uint64_t *out_key)
{
    SH_CHECK_ARG(cs && c && pose && out_key && cs->ctx == c->ctx && count >= 0);
    slamhip_ctx *ctx = c->ctx;
    SH_HIP(hipSetDevice(ctx->device));
    if (c->async_dirty) SH_TRY(slamhip_comm_wait(c, nullptr));    // (collectives of one communicator are issued from one stream at a time)
    sh_mail_guard lock(ctx);
    // A rank whose search cannot be enqueued still joins the collective -- with the neutral key -- and reports afterwards: the
    // other ranks are in it and would wait for this one for good.
    int32_t rc_local = SLAMHIP_OK;
    std::string err;
    // (the search delivers into a word of the handle's result ring -- no final arriver in the kernel, slamhip_cs_search_shard_enqueue --
    // and the collective reduces that word in place: the ring rests a slot only three launches later)
    uint64_t *d_key = c->d_sync_key;
    if (count > 0) {
        const uint64_t *slot = nullptr;
        rc_local = slamhip_cs_search_shard_enqueue(cs, pose, first, count, &slot);
        if (rc_local != SLAMHIP_OK) err = slamhip_last_error(); else d_key = (uint64_t *)slot;
    }
    if (count <= 0 || rc_local != SLAMHIP_OK) (void)hipMemsetAsync(d_key, 0xFF, sizeof(uint64_t), ctx->stream);       // (the neutral key)
    SH_NCCL(c, c->api.AllReduce(d_key, d_key, 1, ncclUint64, ncclMin, c->comm, ctx->stream));
    SH_TRY(sh_publish(ctx, d_key, 2));
    cs_layout_idle_refresh(cs);                                    // (host work under the search: cs_launch_distance)
    SH_TRY(sh_host_wait(ctx));
    *out_key = *(volatile uint64_t *)ctx->mailbox;
    if (rc_local != SLAMHIP_OK) { slamhip_set_error("%s", err.c_str()); return rc_local; }
    return SLAMHIP_OK;
}

// The reduced key decoded on the device: the winner's pose (search_pose + offs[index - 1], :635-637; theta normalised, :746) into
// the handle's result block for the map updates queued behind, key and pose into the mailbox for the host.
__global__ void k_winner_from_key(const unsigned long long *__restrict__ key, const float *__restrict__ offs_flat, int n_offs,
                                  float bx, float by, float bth, unsigned long long *__restrict__ result, uint32_t *__restrict__ mailbox, uint32_t seq)
{
    if (threadIdx.x != 0) return;
    const unsigned long long k = *key;
    const uint32_t flat = (uint32_t)k;
    float x = bx, y = by, th = bth;
    if (flat > 0 && flat <= (uint32_t)n_offs) { x = bx + offs_flat[3 * (size_t)(flat - 1)]; y = by + offs_flat[3 * (size_t)(flat - 1) + 1]; th = bth + offs_flat[3 * (size_t)(flat - 1) + 2]; }
    const float thn = sh_normalize_angle(th);
    float *pose = (float *)(result + 1);                           // (the result block: key | pose x, y, theta normalised, theta | ...)
    result[0] = k;
    pose[0] = x; pose[1] = y; pose[2] = thn; pose[3] = th;
    if (mailbox) {
        *(unsigned long long *)mailbox = k;
        float *mp = (float *)(mailbox + 2);
        mp[0] = x; mp[1] = y; mp[2] = thn;
        __hip_atomic_store(mailbox + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One scan of the SLAM loop on every rank (CoreSLAMProcessor.cs:732 -> :750-751), no host hop between the exchange and the map
// updates: K1 over this rank's block, ncclAllReduce(min, uint64, 1), a one-thread launch that decodes the reduced key into the
// winner's pose ON THE DEVICE (every rank holds the whole jitter list) and hands key + pose to the host, and behind it the
// replicas' map updates from that device-resident pose -- all on the operator's stream.  Returns when the pose is known; the
// updates run on (everything that touches the maps afterwards is ordered behind them).  Every rank makes the same call.
// (the body shared by the one-process-per-GPU form and the single-process group: `comm` is this rank's communicator, d_neutral a
// device word of the caller's for a rank without candidates)
static int32_t fused_allreduce_scan(slamhip_cs *cs, rccl_api &api, ncclComm_t comm, uint64_t *d_neutral, const float pose[3], int32_t first, int32_t count,
                                    float hole_width, int32_t quality, int32_t max_hits, float out_pose[3], int32_t *out_dist, int32_t *out_index)
{
    slamhip_ctx *ctx = cs->ctx;
    sh_mail_guard lock(ctx);
    int32_t rc_local = SLAMHIP_OK;
    std::string err;
    uint64_t *d_key = d_neutral;
    if (count > 0) {
        const uint64_t *slot = nullptr;
        rc_local = slamhip_cs_search_shard_enqueue(cs, pose, first, count, &slot);        // (into the handle's result ring: no final arriver in the kernel)
        if (rc_local != SLAMHIP_OK) err = slamhip_last_error(); else d_key = (uint64_t *)slot;
    } else rc_local = cs_flush_generate(cs);                      // (a rank without candidates still decodes the winner from the list)
    if (count <= 0 || rc_local != SLAMHIP_OK) (void)hipMemsetAsync(d_key, 0xFF, sizeof(uint64_t), ctx->stream);
    {
        const ncclResult_t r_ = api.AllReduce(d_key, d_key, 1, ncclUint64, ncclMin, comm, ctx->stream);
        if (r_ != ncclSuccess) { slamhip_set_error("ncclAllReduce failed: %s", api.GetErrorString(r_)); return SLAMHIP_ERR_RCCL; }
    }
    const uint32_t seq = sh_mail_seq_next(ctx);
    // (the reduced key is decoded by the HoleMap update itself where that is one launch -- every workgroup decodes, the first one
    // delivers key + pose to the mailbox: see slamhip_cs_search_and_update -- and by a launch of its own otherwise)
    static const bool k1_delivers = getenv("SLAMHIP_FUSED_K1_DELIVERS") != nullptr;
    const bool decode = !ctx->mail_off && ctx->timing == 0 && rc_local == SLAMHIP_OK && cs->n_points > 0 && cs_holemap_one_launch(cs) && !k1_delivers;
    int32_t rc_u = SLAMHIP_OK;
    if (decode) {
        cs_k2_winner win;
        win.d_key = d_key; win.d_offs_flat = cs->d_offs_flat; win.n_offs = cs->n_offs; win.bx = pose[0]; win.by = pose[1]; win.bth = pose[2]; win.mail = ctx->mailbox; win.seq = seq;
        rc_u = cs_launch_holemap_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), hole_width, quality, true, max_hits, &win);
        if (rc_u != SLAMHIP_OK) return rc_u;                       // (nothing was launched that would deliver)
    } else {
    hipLaunchKernelGGL(k_winner_from_key, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long *)d_key, (const float *)cs->d_offs_flat,
                       cs->n_offs, pose[0], pose[1], pose[2], (unsigned long long *)cs->d_key, ctx->mail_off ? (uint32_t *)nullptr : ctx->mailbox, seq);
    rc_u = hipGetLastError() == hipSuccess ? SLAMHIP_OK : SLAMHIP_ERR_HIP;
    }
    if (!decode && rc_u == SLAMHIP_OK && rc_local == SLAMHIP_OK && cs->n_points > 0) {
        if (ctx->timing == 0) rc_u = cs_launch_holemap_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), hole_width, quality, true, max_hits);
        else {
            rc_u = cs_launch_holemap_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), hole_width, quality);
            if (rc_u == SLAMHIP_OK) rc_u = cs_launch_obstacle_update(cs, cs->d_best_pose, make_float4(0, 0, 0, 0), max_hits);
        }
    }
    cs_layout_idle_refresh(cs);
    if (ctx->mail_off) {
        SH_HIP(hipMemcpyAsync(ctx->mailbox, cs->d_key, sizeof(uint32_t) * 8, hipMemcpyDeviceToHost, ctx->stream));
        SH_HIP(hipStreamSynchronize(ctx->stream));
    } else SH_TRY(sh_flag_wait(ctx, ctx->mailbox + 15, seq));
    cs->hole_pixels_pending = rc_u == SLAMHIP_OK && rc_local == SLAMHIP_OK && cs->n_points > 0;
    const uint64_t key = *(const volatile uint64_t *)ctx->mailbox;
    const float *hp = (const float *)(ctx->mailbox + 2);
    if (out_pose) { out_pose[0] = hp[0]; out_pose[1] = hp[1]; out_pose[2] = hp[2]; }
    if (out_dist) *out_dist = (int32_t)(uint32_t)(key >> 32);
    if (out_index) *out_index = (int32_t)(uint32_t)key;
    if (rc_local != SLAMHIP_OK) { slamhip_set_error("%s", err.c_str()); return rc_local; }
    SH_TRY(rc_u);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_cs_search_allreduce_and_update(slamhip_cs *cs, slamhip_comm *c, const float pose[3], int32_t first, int32_t count,
                                                          float hole_width, int32_t quality, int32_t max_hits,
                                                          float out_pose[3], int32_t *out_dist, int32_t *out_index)
{
    SH_CHECK_ARG(cs && c && pose && cs->ctx == c->ctx && count >= 0);
    SH_CHECK_ARG(quality >= 0 && quality <= 256 && max_hits >= -128 && max_hits <= 127);
    SH_HIP(hipSetDevice(c->ctx->device));
    if (c->async_dirty) SH_TRY(slamhip_comm_wait(c, nullptr));
    return fused_allreduce_scan(cs, c->api, c->comm, c->d_sync_key, pose, first, count, hole_width, quality, max_hits, out_pose, out_dist, out_index);
}

// The same for a single-process group: one scan on every GPU of the group at the same time (a worker thread each) -- its block of
// the flat list, its end of the collective, the winner decoded on its device, its replica's map updates queued behind; back with
// rank 0's copy of key and pose (every rank holds the same).
extern "C" int32_t slamhip_group_search_and_update(slamhip_group *g, const float pose[3], float hole_width, int32_t quality, int32_t max_hits,
                                                   float out_pose[3], int32_t *out_dist, int32_t *out_index)
{
    SH_CHECK_ARG(g && pose);
    SH_CHECK_ARG(quality >= 0 && quality <= 256 && max_hits >= -128 && max_hits <= 127);
    const int K = g->n_offs + 1;
    const float p3[3] = { pose[0], pose[1], pose[2] };
    std::vector<float> poses((size_t)g->n * 3, 0.0f);
    std::vector<int32_t> dist((size_t)g->n, 0), idx((size_t)g->n, 0);
    float *pp = poses.data(); int32_t *pd = dist.data(), *pi = idx.data();
    SH_TRY(group_run(g, [g, K, p3, hole_width, quality, max_hits, pp, pd, pi](int r) -> int32_t {
        const int first = (int)((long long)K * r / g->n), last = (int)((long long)K * (r + 1) / g->n);
        return fused_allreduce_scan(g->cs[r], g->api, g->comm[r], g->d_key[r], p3, first, last - first, hole_width, quality, max_hits,
                                    pp + 3 * r, pd + r, pi + r);
    }));
    for (int r = 1; r < g->n; r++)
        if (pd[r] != pd[0] || pi[r] != pi[0]) SH_FAIL(SLAMHIP_ERR_RCCL, "rank %d decoded another winner than rank 0 (%d/%d vs %d/%d)", r, pi[r], pd[r], pi[0], pd[0]);
    if (out_pose) { out_pose[0] = pp[0]; out_pose[1] = pp[1]; out_pose[2] = pp[2]; }
    if (out_dist) *out_dist = pd[0];
    if (out_index) *out_index = pi[0];
    return SLAMHIP_OK;
}

// Replica check across the ranks: every rank checksums its two maps, one min and one max all-reduce of the
// two words; *out_equal = 1 when min == max for both, i.e. the replicas are bit-identical.  Every rank makes the same call.
extern "C" int32_t slamhip_comm_replicas_equal(slamhip_cs *cs, slamhip_comm *c, int32_t *out_equal)
{
    SH_CHECK_ARG(cs && c && out_equal && cs->ctx == c->ctx);
    slamhip_ctx *ctx = c->ctx;
    SH_HIP(hipSetDevice(ctx->device));
    if (c->async_dirty) SH_TRY(slamhip_comm_wait(c, nullptr));    // (collectives of one communicator are issued from one stream at a time)
    sh_mail_guard lock(ctx);
    // (a rank whose checksums cannot be enqueued still joins the two collectives -- the others are in them -- and reports afterwards)
    const int32_t rc_local = cs_maps_checksum_enqueue(cs);
    const std::string err = rc_local != SLAMHIP_OK ? std::string(slamhip_last_error()) : std::string();
    uint64_t *d = cs->d_key + 4;                                   // words 4, 5: the checksums; 6, 7: their copies for the max
    SH_HIP(hipMemcpyAsync(d + 2, d, 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
    SH_NCCL(c, c->api.AllReduce(d, d, 2, ncclUint64, ncclMin, c->comm, ctx->stream));
    SH_NCCL(c, c->api.AllReduce(d + 2, d + 2, 2, ncclUint64, ncclMax, c->comm, ctx->stream));
    SH_TRY(sh_publish(ctx, d, 8));
    SH_TRY(sh_host_wait(ctx));
    uint64_t m[4];
    memcpy(m, (const void *)ctx->mailbox, sizeof(m));
    *out_equal = (m[0] == m[2] && m[1] == m[3]) ? 1 : 0;
    if (rc_local != SLAMHIP_OK) { *out_equal = 0; slamhip_set_error("%s", err.c_str()); return rc_local; }
    return SLAMHIP_OK;
}

// Latency of the exchange step alone: `iters` 8-byte min all-reduces back to back on the communicator's stream between two
// events; *out_us = microseconds per collective (device time, no search in front).  Every rank makes the same call.
extern "C" int32_t slamhip_comm_allreduce_probe(slamhip_comm *c, int32_t iters, float *out_us)
{
    SH_CHECK_ARG(c && iters >= 1 && out_us);
    SH_HIP(hipSetDevice(c->ctx->device));
    SH_TRY(slamhip_comm_wait(c, nullptr));
    hipEvent_t a = nullptr, b = nullptr;
    SH_HIP(hipEventCreate(&a));
    hipError_t e = hipEventCreate(&b);
    if (e != hipSuccess) { (void)hipEventDestroy(a); SH_HIP(e); }
    int32_t rc = SLAMHIP_OK;
    (void)hipMemsetAsync(c->d_sync_key, 0xFF, 16, c->stream);
    for (int w = 0; w < 3 && rc == SLAMHIP_OK; w++)                // (warm-up: RCCL sets up its channels on first use)
        if (c->api.AllReduce(c->d_sync_key, c->d_sync_key, 1, ncclUint64, ncclMin, c->comm, c->stream) != ncclSuccess) rc = SLAMHIP_ERR_RCCL;
    (void)hipEventRecord(a, c->stream);
    for (int i = 0; i < iters && rc == SLAMHIP_OK; i++)
        if (c->api.AllReduce(c->d_sync_key, c->d_sync_key, 1, ncclUint64, ncclMin, c->comm, c->stream) != ncclSuccess) rc = SLAMHIP_ERR_RCCL;
    (void)hipEventRecord(b, c->stream);
    e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (rc != SLAMHIP_OK) SH_FAIL(SLAMHIP_ERR_RCCL, "ncclAllReduce failed in the latency probe");
    SH_HIP(e);
    *out_us = ms * 1000.0f / (float)iters;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_comm_info(slamhip_comm *c, int32_t *rank, int32_t *n_ranks)
{
    SH_CHECK_ARG(c);
    if (rank) *rank = c->rank;
    if (n_ranks) *n_ranks = c->n_ranks;
    return SLAMHIP_OK;
}
