// developer probe: can the host store straight into device memory (large BAR)?  usage: bar_test <mode>  (0 fine-grained, 1 plain hipMalloc, 2 uncached)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
__global__ void k_sum(const unsigned *p, int n, unsigned long long *out)
{
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    atomicAdd(out, s);
}
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int n = 8192;                                   // 32 KB
    unsigned *d = nullptr; unsigned long long *out = nullptr, *hout = nullptr;
    hipError_t e;
    if (mode == 0) e = hipExtMallocWithFlags((void **)&d, n * 4, hipDeviceMallocFinegrained);
    else if (mode == 2) e = hipExtMallocWithFlags((void **)&d, n * 4, hipDeviceMallocUncached);
    else e = hipMalloc((void **)&d, n * 4);
    printf("alloc mode %d: %s ptr %p\n", mode, hipGetErrorString(e), (void *)d);
    hipMalloc((void **)&out, 8); hipHostMalloc((void **)&hout, 8);
    hipPointerAttribute_t at; memset(&at, 0, sizeof(at));
    e = hipPointerGetAttributes(&at, d);
    printf("attr: %s type %d host %p dev %p managed %d\n", hipGetErrorString(e), (int)at.type, at.hostPointer, at.devicePointer, (int)at.isManaged);
    fflush(stdout);
    unsigned *src = (unsigned *)malloc(n * 4);
    for (int rep = 0; rep < 5; rep++) {
        unsigned long long want = 0;
        for (int i = 0; i < n; i++) { src[i] = (unsigned)(i * 2654435761u + rep); want += src[i]; }
        hipMemset(out, 0, 8);
        hipDeviceSynchronize();
        const double t0 = now();
        memcpy(d, src, n * 4);                            // <- host stores into device memory
        __sync_synchronize();
        const double t1 = now();
        hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, 0, d, n, out);
        hipMemcpy(hout, out, 8, hipMemcpyDeviceToHost);
        printf("rep %d: host write of %d KB took %.2f us; kernel saw %s\n", rep, n * 4 / 1024, t1 - t0, *hout == want ? "the data" : "STALE / WRONG data");
        fflush(stdout);
    }
    return 0;
}
