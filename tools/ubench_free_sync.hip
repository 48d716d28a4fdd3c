// Does hipFree / hipHostFree wait for work in flight on OTHER streams?  (The library's destroy paths rely on explicit stream
// synchronisation only; this measures what the runtime adds.)   hipcc --offload-arch=gfx950 -O2 -o /tmp/fs tools/ubench_free_sync.hip && /tmp/fs
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned *out) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} if (out) *out = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    unsigned *d1, *d2; void *h1;
    hipMalloc(&d1, 4096); hipMalloc(&d2, 4096); hipHostMalloc(&h1, 128, hipHostMallocMapped | hipHostMallocCoherent);
    printf("hipHostMalloc(128) = %p, hipMalloc = %p, a stack address = %p\n", h1, (void *)d1, (void *)&a);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 100ull, d1); hipStreamSynchronize(a);      // warm
    for (int which = 0; which < 3; which++) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 100000000ull / 20, d1);               // 50 ms at 100 MHz
        const double t0 = now();
        if (which == 0) hipFree(d2);
        else if (which == 1) hipHostFree(h1);
        else hipStreamSynchronize(b);
        const double t1 = now();
        hipStreamSynchronize(a);
        const double t2 = now();
        printf("%s returned after %.2f ms; the kernel on the other stream ended after %.2f ms\n", which == 0 ? "hipFree" : which == 1 ? "hipHostFree" : "hipStreamSynchronize(idle stream)", t1 - t0, t2 - t0);
    }
    return 0;
}
