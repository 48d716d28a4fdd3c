// developer probe: is __builtin_sqrtf (llvm.sqrt.f32 under -fno-fast-math) the correctly rounded binary32 root on this toolchain?
// Exhaustive over every non-negative finite binary32 value against (float)sqrt((double)x), which is (53 >= 2 * 24 + 2).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -o /tmp/sq tools/ubench_sqrt.hip && /tmp/sq
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long *bad, unsigned *first)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long nb = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < 0x7f800000ull; i += stride) {
        const float x = __uint_as_float((unsigned)i);
        const float a = __builtin_sqrtf(x);
        const float b = (float)sqrt((double)x);
        if (__float_as_uint(a) != __float_as_uint(b)) { nb++; atomicMin(first, (unsigned)i); }
    }
    if (nb) atomicAdd(bad, nb);
}
int main()
{
    unsigned long long *bad, h = 0; unsigned *first, hf = 0xffffffffu;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    hipMemcpy(bad, &h, 8, hipMemcpyHostToDevice); hipMemcpy(first, &hf, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, first);
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("__builtin_sqrtf differs from the correctly rounded root for %llu of 2139095040 inputs (first: 0x%08x)\n", h, hf);
    return 0;
}
