// context.hip -- library / context entry points of include/slamhip.h (gfx950 only).
#include "common.h"
#include <stdlib.h>
#include <time.h>

static thread_local char g_err[512] = "";

void slamhip_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *slamhip_version(void) { return "slamhip 0.1.0 (gfx950, HIP)"; }
extern "C" const char *slamhip_last_error(void) { return g_err; }

extern "C" int32_t slamhip_device_count(int32_t *out)
{
    SH_CHECK_ARG(out);
    int n = 0;
    SH_HIP(hipGetDeviceCount(&n));
    *out = n;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_create(int32_t device, slamhip_ctx **out)
{
    SH_CHECK_ARG(out);
    int n = 0;
    SH_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) SH_FAIL(SLAMHIP_ERR_INVALID, "device ordinal %d out of range (have %d)", device, n);
    SH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        SH_FAIL(SLAMHIP_ERR_INVALID, "device %d is %s; libslamhip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    slamhip_ctx *c = (slamhip_ctx *)calloc(1, sizeof(slamhip_ctx));
    if (!c) SH_FAIL(SLAMHIP_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        // (the highest priority the device offers for the plan stream: a plan is a few hundred short wavefronts that share the compute
        // units with a search launch in full flight -- at equal priority they trickle in as that launch's workgroups retire, ~10 us)
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        e = hipStreamCreateWithPriority(&c->plan_stream, hipStreamNonBlocking, prio_hi);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->mirror_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&c->d_touch, 256);
    if (e == hipSuccess) {
        // first use = hardware queue: in this order, see common.h
        hipStream_t order[4] = { c->stream, c->plan_stream, c->side_stream, c->mirror_stream };
        for (int i = 0; i < 4 && e == hipSuccess; i++) {
            e = hipMemsetAsync((char *)c->d_touch + 64 * i, 0, 64, order[i]);
            if (e == hipSuccess) e = hipStreamSynchronize(order[i]);
        }
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->mailbox, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) {
        if (c->d_touch) (void)hipFree(c->d_touch);
        if (c->mirror_stream) (void)hipStreamDestroy(c->mirror_stream);
        if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
        if (c->plan_stream) (void)hipStreamDestroy(c->plan_stream);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        free(c);
        SH_FAIL(SLAMHIP_ERR_HIP, "context creation failed (streams / pinned mailbox): %s", hipGetErrorString(e));
    }
    memset(c->mailbox, 0, 64);
    {
        pthread_mutexattr_t at;
        pthread_mutexattr_init(&at);
        pthread_mutexattr_settype(&at, PTHREAD_MUTEX_RECURSIVE);
        pthread_mutex_init(&c->mail_lock, &at);
        pthread_mutexattr_destroy(&at);
    }
    c->mail_off = getenv("SLAMHIP_NO_HOSTWAIT") && atoi(getenv("SLAMHIP_NO_HOSTWAIT"));
    c->wait_timeout_ms = getenv("SLAMHIP_WAIT_TIMEOUT_MS") ? atoll(getenv("SLAMHIP_WAIT_TIMEOUT_MS")) : 10000;
    {
        int lb = 0;
        if (hipDeviceGetAttribute(&lb, hipDeviceAttributeIsLargeBar, device) != hipSuccess) { lb = 0; (void)hipGetLastError(); }
        c->large_bar = lb != 0 && !c->mail_off && !getenv("SLAMHIP_NO_DIRECT_UPLOAD");
    }
    *out = c;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_destroy(slamhip_ctx *c)
{
    if (!c) return SLAMHIP_OK;
    (void)hipSetDevice(c->device);
    if (!c->poisoned) (void)hipStreamSynchronize(c->stream);
    else {
        // a wait on this context passed its bound: the stream may never drain -- give it the bound once more, then let go (the
        // process should end: device state is unknown, see slamhip.h)
        timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
        const int64_t bound = c->wait_timeout_ms > 0 ? c->wait_timeout_ms : 10000;
        while (hipStreamQuery(c->stream) == hipErrorNotReady) {
            timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000L + (t1.tv_nsec - t0.tv_nsec) / 1000000L >= bound) break;
            struct timespec nap = { 0, 200000 }; nanosleep(&nap, nullptr);
        }
    }
    for (int i = 0; i < c->n_pending; i++) { (void)hipEventDestroy(c->pending[i].a); (void)hipEventDestroy(c->pending[i].b); }
    for (int i = 0; i < c->n_pool; i++) (void)hipEventDestroy(c->pool[i]);
    free(c->pending); free(c->pool);
    if (!c->poisoned) { (void)hipStreamSynchronize(c->plan_stream); (void)hipStreamSynchronize(c->side_stream); (void)hipStreamSynchronize(c->mirror_stream); }
    (void)hipStreamDestroy(c->mirror_stream); (void)hipStreamDestroy(c->side_stream); (void)hipStreamDestroy(c->plan_stream);
    (void)hipStreamDestroy(c->stream);
    (void)hipFree(c->d_touch);
    if (c->mailbox) (void)hipHostFree(c->mailbox);
    pthread_mutex_destroy(&c->mail_lock);
    free(c);
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_synchronize(slamhip_ctx *c)
{
    SH_CHECK_ARG(c);
    SH_HIP(hipStreamSynchronize(c->stream));
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_set_wait_timeout(slamhip_ctx *c, int64_t timeout_ms)
{
    SH_CHECK_ARG(c);
    c->wait_timeout_ms = timeout_ms;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_poisoned(slamhip_ctx *c, int32_t *out)
{
    SH_CHECK_ARG(c && out);
    *out = c->poisoned ? 1 : 0;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_device(slamhip_ctx *c, int32_t *out)
{
    SH_CHECK_ARG(c && out);
    *out = c->device;
    return SLAMHIP_OK;
}

extern "C" void *slamhip_ctx_stream(slamhip_ctx *c) { return c ? (void *)c->stream : nullptr; }

// ---- mailbox: results and completion of a blocking call in pinned host memory ---------------------------------
__global__ void k_publish(const uint32_t *__restrict__ src, int n_words, uint32_t *__restrict__ mailbox, uint32_t seq)
{
    if ((int)threadIdx.x < n_words) mailbox[threadIdx.x] = src[threadIdx.x];
    __syncthreads();                                               // (one wavefront: orders the lanes' stores before lane 0's release)
    if (threadIdx.x == 0) __hip_atomic_store(mailbox + 15, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int32_t sh_publish(slamhip_ctx *ctx, const void *d_src, int n_words)
{
    return sh_publish_seq(ctx, d_src, n_words, sh_mail_seq_next(ctx));
}

int32_t sh_publish_seq(slamhip_ctx *ctx, const void *d_src, int n_words, uint32_t seq)
{
    SH_CHECK_ARG(n_words >= 0 && n_words <= 15 && (d_src || n_words == 0));
    if (ctx->poisoned) SH_FAIL(SLAMHIP_ERR_TIMEOUT, "the context was poisoned by a blocking wait that timed out; destroy it");
    if (ctx->mail_off) {                                           // (SLAMHIP_NO_HOSTWAIT=1: the copy + synchronise form)
        if (n_words > 0) SH_HIP(hipMemcpyAsync(ctx->mailbox, d_src, sizeof(uint32_t) * (size_t)n_words, hipMemcpyDeviceToHost, ctx->stream));
        return SLAMHIP_OK;
    }
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, ctx->stream, (const uint32_t *)d_src, n_words, ctx->mailbox, seq);
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}

__global__ void __launch_bounds__(1024)
k_upload16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16, uint32_t *__restrict__ flags, uint32_t seq)
{
    sh_upload16_part(src, dst, n16, (int)blockIdx.x, flags, seq);
}

int32_t sh_upload(slamhip_ctx *ctx, const void *h_src, void *d_dst, size_t bytes, uint32_t *h_flags, uint32_t seq)
{
    SH_CHECK_ARG(h_src && d_dst && h_flags && bytes % 16 == 0 && bytes / 16 < (size_t)INT32_MAX);
    hipLaunchKernelGGL(k_upload16, dim3(SH_UPLOAD_PARTS), dim3(1024), 0, ctx->stream, (const uint4 *)h_src, (uint4 *)d_dst, (int)(bytes / 16), h_flags, seq);
    SH_HIP(hipGetLastError());
    return SLAMHIP_OK;
}

// Waits until *flag has reached `val` (sequence numbers only grow: the comparison is wrap-safe, and a later number that
// landed first also ends the wait).  The host polls the pinned word for up to ~300 us -- a blocking call's device latency is
// tens of microseconds, and a stream synchronisation costs 15-20 us on this stack -- and then hands the wait to
// hipStreamSynchronize: a long-running or faulted stream no longer pins a core or is reported late.
static inline bool sh_flag_reached(volatile uint32_t *flag, uint32_t val)
{
    return (int32_t)(__atomic_load_n(flag, __ATOMIC_ACQUIRE) - val) >= 0;
}
static int32_t sh_flag_wait_bounded(slamhip_ctx *ctx, volatile uint32_t *flag, uint32_t val, int64_t timeout_ms);
int32_t sh_flag_wait(slamhip_ctx *ctx, volatile uint32_t *flag, uint32_t val)
{
    if (ctx->poisoned) SH_FAIL(SLAMHIP_ERR_TIMEOUT, "the context was poisoned by a blocking wait that timed out; destroy it");
    const int32_t rc = sh_flag_wait_bounded(ctx, flag, val, ctx->wait_timeout_ms);
    if (rc == SLAMHIP_ERR_TIMEOUT) ctx->poisoned = true;
    return rc;
}
// (ctx == nullptr: the wait has no stream to ask -- the CPU-side test hook slamhip_debug_flag_wait)
static int32_t sh_flag_wait_bounded(slamhip_ctx *ctx, volatile uint32_t *flag, uint32_t val, int64_t timeout_ms)
{
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long spins = 1;; spins++) {
        if (sh_flag_reached(flag, val)) return SLAMHIP_OK;
        if ((spins & 0x3ff) == 0) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 300000L) break;
        }
        __builtin_ia32_pause();
    }
    // Past the spin budget (a long search: very many candidates, a 4096^2 map): keep waiting for the WORD, not for the stream --
    // a call that returns with the pose while the map updates run on must not sit through those updates as well.  The stream is
    // only ASKED (hipStreamQuery: a fault is reported, an idle stream without the word is an error), between short sleeps.
    for (;;) {
        for (int k = 0; k < 64; k++) { if (sh_flag_reached(flag, val)) return SLAMHIP_OK; __builtin_ia32_pause(); }
        if (timeout_ms > 0) {
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000L + (t1.tv_nsec - t0.tv_nsec) / 1000000L >= timeout_ms)
                SH_FAIL(SLAMHIP_ERR_TIMEOUT, "a blocking wait passed its bound of %lld ms (SLAMHIP_WAIT_TIMEOUT_MS): the completion word never arrived", (long long)timeout_ms);
        }
        const hipError_t q = ctx ? hipStreamQuery(ctx->stream) : hipErrorNotReady;
        if (sh_flag_reached(flag, val)) return SLAMHIP_OK;
        if (q == hipSuccess) {                                     // idle: the word must be there (a store to pinned host memory, released at system scope)
            for (int k = 0; k < 1000; k++) { if (sh_flag_reached(flag, val)) return SLAMHIP_OK; __builtin_ia32_pause(); }
            SH_FAIL(SLAMHIP_ERR_HIP, "the stream is idle but the completion word never arrived");
        }
        if (q != hipErrorNotReady) SH_HIP(q);                      // a device fault
        timespec ts = { 0, 20000 };                                // 20 us
        nanosleep(&ts, nullptr);
    }
}

// CPU-side test hook: the very wait loop of a blocking call on a caller-owned word, with no stream behind it
extern "C" int32_t slamhip_debug_flag_wait(volatile uint32_t *flag, uint32_t val, int64_t timeout_ms)
{
    SH_CHECK_ARG(flag);
    return sh_flag_wait_bounded(nullptr, flag, val, timeout_ms);
}

int32_t sh_host_wait(slamhip_ctx *ctx)
{
    if (ctx->poisoned) SH_FAIL(SLAMHIP_ERR_TIMEOUT, "the context was poisoned by a blocking wait that timed out; destroy it");
    if (ctx->mail_off) { SH_HIP(hipStreamSynchronize(ctx->stream)); return SLAMHIP_OK; }
    return sh_flag_wait(ctx, ctx->mailbox + 15, ctx->mail_seq);
}

// ---- timing ------------------------------------------------------------------------------------
static hipEvent_t ev_get(slamhip_ctx *c)
{
    if (c->n_pool > 0) return c->pool[--c->n_pool];
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
static void ev_put(slamhip_ctx *c, hipEvent_t e)
{
    if (c->n_pool == c->cap_pool) {
        c->cap_pool = c->cap_pool ? c->cap_pool * 2 : 64;
        c->pool = (hipEvent_t *)realloc(c->pool, sizeof(hipEvent_t) * c->cap_pool);
    }
    c->pool[c->n_pool++] = e;
}

sh_timer::sh_timer(slamhip_ctx *c, int w) : ctx(c), which(w), a(nullptr)
{
    if (!(ctx->timing & (1u << which))) return;
    a = ev_get(ctx);
    if (a) (void)hipEventRecord(a, ctx->stream);
}
sh_timer::~sh_timer()
{
    if (!a) return;
    hipEvent_t b = ev_get(ctx);
    if (!b) { ev_put(ctx, a); return; }
    (void)hipEventRecord(b, ctx->stream);
    if (ctx->n_pending == ctx->cap_pending) {
        ctx->cap_pending = ctx->cap_pending ? ctx->cap_pending * 2 : 256;
        ctx->pending = (slamhip_ctx::TimedLaunch *)realloc(ctx->pending, sizeof(slamhip_ctx::TimedLaunch) * ctx->cap_pending);
    }
    ctx->pending[ctx->n_pending++] = { a, b, which };
}

int32_t sh_timing_collect(slamhip_ctx *c)
{
    SH_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->n_pending; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->pending[i].a, c->pending[i].b) == hipSuccess) {
            c->ms[c->pending[i].which] += ms;
            c->launches[c->pending[i].which] += 1;
        }
        ev_put(c, c->pending[i].a);
        ev_put(c, c->pending[i].b);
    }
    c->n_pending = 0;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_timing_enable(slamhip_ctx *c, int32_t mask)
{
    SH_CHECK_ARG(c);
    if (c->timing) SH_TRY(sh_timing_collect(c));
    c->timing = (uint32_t)mask;
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_timing_reset(slamhip_ctx *c)
{
    SH_CHECK_ARG(c);
    SH_TRY(sh_timing_collect(c));
    memset(c->ms, 0, sizeof(c->ms));
    memset(c->launches, 0, sizeof(c->launches));
    return SLAMHIP_OK;
}

extern "C" int32_t slamhip_ctx_timing_get(slamhip_ctx *c, int32_t which, double *out_ms, int64_t *out_launches)
{
    SH_CHECK_ARG(c && which >= 0 && which < SLAMHIP_K_COUNT);
    SH_TRY(sh_timing_collect(c));
    if (out_ms) *out_ms = c->ms[which];
    if (out_launches) *out_launches = c->launches[which];
    return SLAMHIP_OK;
}
