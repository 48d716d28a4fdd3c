// micro-benchmark: one launch pulls N bytes from pinned host memory over PCIe with W workgroups of 1024 lanes (16 bytes per lane and
// pass) -- how the scan upload (sh_upload16_unit, common.h) should be cut.   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_pull tools/ubench_pull.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(1024) pull(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16)
{
    for (int i = blockIdx.x * 1024 + threadIdx.x; i < n16; i += gridDim.x * 1024) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) pull256(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int n16)
{
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}
int main()
{
    const int bytes_list[] = { 16384, 36864, 46080, 131072 };
    hipStream_t st; hipStreamCreate(&st);
    uint4 *h, *d; hipHostMalloc(&h, 1 << 20); hipMalloc(&d, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int bytes : bytes_list) {
        const int n16 = bytes / 16;
        for (int mode = 0; mode < 2; mode++)
        for (int w : { 1, 2, 4, 8, 16, 32 }) {
            const int wg = mode ? w * 4 : w;
            if (mode == 0 && w > 16) continue;
            for (int i = 0; i < 20; i++) { if (mode) hipLaunchKernelGGL(pull256, dim3(wg), dim3(256), 0, st, h, d, n16); else hipLaunchKernelGGL(pull, dim3(wg), dim3(1024), 0, st, h, d, n16); }
            hipEventRecord(a, st);
            for (int i = 0; i < 200; i++) { if (mode) hipLaunchKernelGGL(pull256, dim3(wg), dim3(256), 0, st, h, d, n16); else hipLaunchKernelGGL(pull, dim3(wg), dim3(1024), 0, st, h, d, n16); }
            hipEventRecord(b, st); hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%6d bytes, %2d workgroups of %4d lanes: %.2f us per launch (back to back)\n", bytes, wg, mode ? 256 : 1024, ms * 1000.f / 200.f);
        }
    }
    return 0;
}
