// micro-benchmark: issue cost of v_fma_f32 with denormal operands / results against normal ones (gfx950, 8 waves/SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 2048
template <int MODE>
__global__ void k_fma(float *out, float a, unsigned bbits, unsigned cbits)
{
    float x0, x1, x2, x3, x4, x5, x6, x7;
    const float b = __uint_as_float(bbits);
    x0 = x1 = x2 = x3 = x4 = x5 = x6 = x7 = __uint_as_float(cbits + threadIdx.x);
    for (int i = 0; i < N_IT; i++) {
        asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                     "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
static void run(const char *name, float a, unsigned bbits, unsigned cbits, float *d)
{
    int blocks = 2048, threads = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_fma<0><<<blocks, threads>>>(d, a, bbits, cbits); hipDeviceSynchronize();
    hipEventRecord(e0); k_fma<0><<<blocks, threads>>>(d, a, bbits, cbits); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winst = (double)blocks * threads / 64 * N_IT * 8;
    printf("%-28s %.3f ms  => %.2f cyc/wave-instr/SIMD @2.4GHz\n", name, ms, (1024.0 * 2.4e9) / (winst / (ms * 1e-3)));
}
int main()
{
    float *d; hipMalloc(&d, 4 * 2048 * 1024);
    run("normal * normal + normal", 1.0001f, 0x3f800100u, 0x3f800000u, d);
    run("normal * denormal + denormal", 3.0f, 2u, 16u, d);          // x += 6 units per step: stays denormal
    run("normal * normal + denormal->n", 1.0f, 0x00800000u, 16u, d); // result leaves the denormal range at once
    run("zero * denormal + denormal", 0.0f, 2u, 16u, d);
    return 0;
}
