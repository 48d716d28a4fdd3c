// micro-benchmark: can a dependent kernel be launched over the tail of its producer on gfx950?
//   A: 256 workgroups busy for ~30 us, the last one to finish publishes a sequence number (release, agent scope)
//   B: one workgroup that waits for the sequence number (acquire) -- bounded spin, so nothing can hang
// variants: B after A in the stream (barrier bit) | B with hipExtAnyOrderLaunch in the same stream | B in a second stream
// prints, in us from A's first stamp: A published, B started, B saw the number, and the host time per A+B pair.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>

__global__ void k_a(unsigned long long *st, unsigned *cnt, unsigned *flag, unsigned seq, int busy_ticks)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) st[0] = t0;
    while ((long long)(wall_clock64() - t0) < busy_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            *cnt = 0;
            st[1] = wall_clock64();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ void k_b(unsigned long long *st, const unsigned *flag, unsigned seq)
{
    if (threadIdx.x == 0) {
        st[2] = wall_clock64();
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq && spins < (1 << 20)) { __builtin_amdgcn_s_sleep(2); spins++; }
        st[3] = wall_clock64();
        st[4] = (unsigned long long)spins;
    }
}

int main()
{
    unsigned long long *st; unsigned *cnt, *flag;
    hipMalloc(&st, 64); hipMalloc(&cnt, 4); hipMalloc(&flag, 4);
    hipMemset(cnt, 0, 4); hipMemset(flag, 0, 4); hipMemset(st, 0, 64);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const char *names[3] = { "in stream (barrier bit)", "hipExtAnyOrderLaunch", "second stream" };
    unsigned seq = 0;
    for (int v = 0; v < 3; v++) {
        double host_us = 0;
        unsigned long long h[5] = { 0 };
        for (int rep = 0; rep < 2; rep++) {
            const int n = rep == 0 ? 3 : 50;
            hipDeviceSynchronize();
            const auto c0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) {
                seq++;
                hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, s1, st, cnt, flag, seq, 3000);
                if (v == 0) hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s1, st, (const unsigned *)flag, seq);
                else if (v == 1) hipExtLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, st, (const unsigned *)flag, seq);
                else hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s2, st, (const unsigned *)flag, seq);
                if (v == 2) hipStreamSynchronize(s2);      // (the next A must not overtake this B's stamps)
            }
            hipStreamSynchronize(s1); hipStreamSynchronize(s2);
            host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count() / n;
        }
        hipMemcpy(h, st, 40, hipMemcpyDeviceToHost);
        printf("%-26s A published %.2f us | B started %.2f | B saw it %.2f (%llu spins) | %.2f us per pair  err=%s\n", names[v],
               (double)(long long)(h[1] - h[0]) * 0.01, (double)(long long)(h[2] - h[0]) * 0.01, (double)(long long)(h[3] - h[0]) * 0.01, h[4], host_us,
               hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
