// common.h -- shared declarations for libslamhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <limits.h>
#include <pthread.h>
#include "../../include/slamhip.h"

void slamhip_set_error(const char *fmt, ...);

#define SH_FAIL(code, ...) do { slamhip_set_error(__VA_ARGS__); return (code); } while (0)
#define SH_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
        slamhip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
        return SLAMHIP_ERR_HIP; } } while (0)
#define SH_CHECK_ARG(cond) do { if (!(cond)) { slamhip_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
        return SLAMHIP_ERR_INVALID; } } while (0)
#define SH_TRY(expr) do { int32_t r_ = (expr); if (r_ != SLAMHIP_OK) return r_; } while (0)

struct slamhip_ctx {
    int device;
    hipStream_t stream;
    // The helper streams of every operator object of the context, made -- and USED once, which is when the runtime binds a stream to a
    // hardware queue -- at context creation, straight after the operator's stream: plan (the search's plan launches, highest priority),
    // side (the next scan's candidate list), mirror (host-mirror pushes).  Queues are dealt to the compute pipes in creation order,
    // and two queues of one pipe do not run side by side: a side stream made late in a process that had made other streams landed on
    // the operator's pipe, its launch waited until the operator's queue went idle, and a scan of the launch-ahead flow took 76 us
    // instead of 46 (round 6, bench.py's CoreSLAMProcessor section).  Four streams made back to back take four different pipes.
    hipStream_t plan_stream, side_stream, mirror_stream;
    void *d_touch;            // 256 B the creation-time launches write
    int num_cus;
    // timing
    uint32_t timing;          // bit mask of timed kernel classes
    struct TimedLaunch { hipEvent_t a, b; int which; };
    TimedLaunch *pending; int n_pending, cap_pending;
    hipEvent_t *pool; int n_pool, cap_pool;     // recycled events
    double ms[SLAMHIP_K_COUNT]; int64_t launches[SLAMHIP_K_COUNT];
    // mailbox: 64 B of pinned, device-visible host memory.  A blocking call ends its launches with sh_publish (copies up
    // to 15 result words here, then stores the call's sequence number into word 15) and waits in sh_host_wait for that
    // word: no device-to-host copy, no stream synchronisation (measured: 8 us less per call than copy + hipStreamSynchronize).
    // The mailbox and its sequence counter are shared by every operator object of the context: a call that uses them holds
    // `mail_lock` (recursive) from the moment it takes its sequence number until it has read its result words, so handles of
    // one context may be driven from different threads (slamhip.h) -- their blocking calls then take turns, as they do on the
    // context's one stream anyway.
    uint32_t *mailbox; uint32_t mail_seq; bool mail_off;
    bool large_bar;           // the host can store straight into device memory (hipDeviceAttributeIsLargeBar): per-scan uploads without a launch
    // A blocking wait is bounded (SLAMHIP_WAIT_TIMEOUT_MS, default 10000; slamhip_ctx_set_wait_timeout): a completion word that does
    // not arrive in that time -- a kernel that never ends keeps hipStreamQuery at NotReady for ever -- ends the call with
    // SLAMHIP_ERR_TIMEOUT and POISONS the context: nothing is restarted or re-executed in-process (device state is unknown), every
    // later blocking call or publish on the context fails with the same code at once, and the caller tears the context down.
    int64_t wait_timeout_ms;  // <= 0: unbounded
    bool poisoned;
    pthread_mutex_t mail_lock;
};
// RAII guard of slamhip_ctx::mail_lock (see there)
struct sh_mail_guard {
    slamhip_ctx *ctx;
    explicit sh_mail_guard(slamhip_ctx *c) : ctx(c) { pthread_mutex_lock(&ctx->mail_lock); }
    ~sh_mail_guard() { pthread_mutex_unlock(&ctx->mail_lock); }
    sh_mail_guard(const sh_mail_guard &) = delete;
    sh_mail_guard &operator=(const sh_mail_guard &) = delete;
};
int32_t sh_publish(slamhip_ctx *ctx, const void *d_src, int n_words);   // enqueue; returns after the launch
int32_t sh_publish_seq(slamhip_ctx *ctx, const void *d_src, int n_words, uint32_t seq);   // ... with a sequence number taken earlier (sh_mail_seq_next)
// Per-scan upload as a launch: SH_UPLOAD_PARTS workgroups pull `bytes` (a multiple of 16) from a pinned staging block over PCIe
// and each then stores `seq` into its own pinned word h_flags[part]; sh_upload_wait(h_flags, seq) tells the host that the staging
// block may be refilled.  (A hipMemcpyAsync right after a mailbox wait takes the runtime's slow path -- it has not seen the stream
// finish yet -- and the cross-engine dependency delays the first kernel: measured 8 us per scan; with this the per-scan path is
// launches only.)  Four workgroups, 16 KB per pass each: one workgroup pulls a scan's 46 KB in 5.5 us -- 2.6 us for the first
// 16 KB, 1.4 us per further pass -- four in 3.1 us (tools/ubench_pull.hip).  The words only say "the block has been READ" (a
// lane's stores depend on its loads, and the barrier collects the lanes): a relaxed store -- a release at system scope writes the
// L2 back first, 3 us.
#define SH_UPLOAD_PARTS 4
int32_t sh_upload(slamhip_ctx *ctx, const void *h_src, void *d_dst, size_t bytes, uint32_t *h_flags, uint32_t seq);
// (the work of one of the upload's workgroups of 1024 lanes: they also ride on another launch as extra workgroups, see k_gather_offsets)
__device__ static inline void sh_upload16_part(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16, int part,
                                               uint32_t *__restrict__ flags, uint32_t seq)
{
    for (int i = part * 1024 + (int)threadIdx.x; i < n16; i += SH_UPLOAD_PARTS * 1024) dst[i] = src[i];
    __syncthreads();                                               // every lane's loads have returned (its stores depend on them)
    if (threadIdx.x == 0) __hip_atomic_store(flags + part, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int32_t sh_flag_wait(slamhip_ctx *ctx, volatile uint32_t *h_flag, uint32_t seq);
static inline int32_t sh_upload_wait(slamhip_ctx *ctx, volatile uint32_t *h_flags, uint32_t seq)
{
    for (int p = 0; p < SH_UPLOAD_PARTS; p++) { const int32_t rc = sh_flag_wait(ctx, h_flags + p, seq); if (rc != SLAMHIP_OK) return rc; }
    return SLAMHIP_OK;
}
int32_t sh_host_wait(slamhip_ctx *ctx);                                  // until the last sh_publish of this context has landed
// (a kernel that is the last of its call may write the mailbox itself: words first, then sh_mail_seq_next() into word 15, released at system scope)
static inline uint32_t sh_mail_seq_next(slamhip_ctx *ctx) { return ++ctx->mail_seq; }

// RAII-ish helper: brackets a kernel class with events when timing is on.
struct sh_timer {
    slamhip_ctx *ctx; int which; hipEvent_t a;
    sh_timer(slamhip_ctx *c, int w);
    ~sh_timer();
};
int32_t sh_timing_collect(slamhip_ctx *ctx);

static inline int sh_div_up(int a, int b) { return (a + b - 1) / b; }

// C# (int)float on x64 (cvttss2si): truncate toward zero, NaN / out-of-range -> INT_MIN.
__host__ __device__ static inline int32_t sh_f2i(float f)
{
    if (!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN;
    return (int32_t)f;
}
// C# unchecked int arithmetic
__host__ __device__ static inline int32_t sh_wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__host__ __device__ static inline int32_t sh_wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__host__ __device__ static inline int32_t sh_wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
__host__ __device__ static inline int32_t sh_sign(int32_t a) { return (a > 0) - (a < 0); }
__host__ __device__ static inline int32_t sh_abs(int32_t a) { return a < 0 ? sh_wsub(0, a) : a; }

// One value for the whole wavefront, fetched by ONE lane per loop iteration (`EXPR` typically draws a work item from an LDS
// counter).  The lane is chosen through an opaque asm, once per evaluation: with `if (lane == 0) k = ...; k = readfirstlane(k);`
// at the head of a persistent loop whose body ends in another `if (lane == 0)` (a pixel's store), the compiler threads the tail's
// knowledge "lane != 0" into the next iteration's head and the structuriser then lets the lanes 1..63 run ahead of lane 0 -- the
// readfirstlane sees the wrong first lane (a zero), the wavefront re-draws the same item for ever (seen on gfx950 / ROCm 7.2
// after an unrelated edit of the loop's body; found with rocgdb).  The asm's result cannot be correlated with `lane`.
#define SH_WAVE_FETCH(dst, EXPR)                                                                    \
    {                                                                                               \
        int leader_;                                                                                \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(leader_)); \
        int v_ = 0;                                                                                 \
        if (leader_ == 0) v_ = (EXPR);                                                              \
        (dst) = __builtin_amdgcn_readfirstlane(v_);                                                 \
    }

// Wave-wide sums / maxima / inclusive scans of an int in six DPP steps each (no LDS permutes: a ds_bpermute -- what __shfl_up / __shfl_down
// compile to -- costs ~100 cycles of latency, and a reduction is six of them in a row).  Lanes whose source lies outside the row,
// or whose row a broadcast step does not address, receive the identity.
template <int CTRL, int ROWS> __device__ static __forceinline__ int sh_dpp_or0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, false); }
template <int CTRL, int ROWS> __device__ static __forceinline__ int sh_dpp_self(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false); }
__device__ static __forceinline__ int sh_wave_scan_incl(int v)         // lane l: the sum of lanes 0 .. l
{
    v += sh_dpp_or0<0x111, 0xf>(v);         // row_shr:1
    v += sh_dpp_or0<0x112, 0xf>(v);         // row_shr:2
    v += sh_dpp_or0<0x114, 0xf>(v);         // row_shr:4
    v += sh_dpp_or0<0x118, 0xf>(v);         // row_shr:8
    v += sh_dpp_or0<0x142, 0xa>(v);         // row_bcast:15 -> rows 1, 3
    v += sh_dpp_or0<0x143, 0xc>(v);         // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ static __forceinline__ int sh_wave_min_all(int v)            // in every lane (a readlane of lane 63: a uniform value)
{
    v = min(v, sh_dpp_self<0x111, 0xf>(v));
    v = min(v, sh_dpp_self<0x112, 0xf>(v));
    v = min(v, sh_dpp_self<0x114, 0xf>(v));
    v = min(v, sh_dpp_self<0x118, 0xf>(v));
    v = min(v, sh_dpp_self<0x142, 0xa>(v));
    v = min(v, sh_dpp_self<0x143, 0xc>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ static __forceinline__ int sh_wave_max_to_lane63(int v)     // valid in lane 63
{
    v = max(v, sh_dpp_self<0x111, 0xf>(v));
    v = max(v, sh_dpp_self<0x112, 0xf>(v));
    v = max(v, sh_dpp_self<0x114, 0xf>(v));
    v = max(v, sh_dpp_self<0x118, 0xf>(v));
    v = max(v, sh_dpp_self<0x142, 0xa>(v));
    v = max(v, sh_dpp_self<0x143, 0xc>(v));
    return v;
}

// Replica check (map updates run as replicas on every GPU; integer-exact kernels keep them bit-identical --
// this is how a host verifies it).  Position-sensitive, order-independent: sum over i of mix64(i << 32 | word_i) mod 2^64, where
// word_i is the element zero-extended from its own width (bit pattern for floats) and mix64 the SplitMix64 finaliser
// (Steele, Lea, Flood: "Fast splittable pseudorandom number generators").  *out must be zero before the launch.
__host__ __device__ static inline unsigned long long sh_mix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
template <typename U>                  // U: uint8_t / uint16_t / uint32_t view of the array
__global__ void __launch_bounds__(256) k_checksum(const U *__restrict__ a, size_t n, unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += sh_mix64(((unsigned long long)i << 32) | (unsigned long long)a[i]);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
template <typename U>
static inline void sh_checksum_launch(slamhip_ctx *ctx, const void *a, size_t n, unsigned long long *d_out)
{
    const size_t want = (n + 2047) / 2048;                          // ~8 elements per lane
    const unsigned grid = (unsigned)(want < 1 ? 1 : want > 2048 ? 2048 : want);
    hipLaunchKernelGGL(k_checksum<U>, dim3(grid), dim3(256), 0, ctx->stream, (const U *)a, n, d_out);
}
