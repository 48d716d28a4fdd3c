// micro-benchmark: issue rate of v_mul_f32 / v_pk_mul_f32 / v_cvt_i32_f32 / v_readlane / ds_read_u16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 4096
__global__ void k_mul(float *out, float a) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < N_IT; i++) {
        asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n"
                     "v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void k_pkmul(float *out, float a) {
    float2v x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float2v av = {a, a};
    for (int i = 0; i < N_IT; i++) {
        asm volatile("v_pk_mul_f32 %0, %8, %0\n v_pk_mul_f32 %1, %8, %1\n v_pk_mul_f32 %2, %8, %2\n v_pk_mul_f32 %3, %8, %3\n"
                     "v_pk_mul_f32 %4, %8, %4\n v_pk_mul_f32 %5, %8, %5\n v_pk_mul_f32 %6, %8, %6\n v_pk_mul_f32 %7, %8, %7\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(av));
    }
    float2v s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_cvt(float *out, float a) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; int y0 = 0, y1 = 0, y2 = 0, y3 = 0;
    for (int i = 0; i < N_IT; i++) {
        asm volatile("v_cvt_i32_f32 %4, %0\n v_cvt_i32_f32 %5, %1\n v_cvt_i32_f32 %6, %2\n v_cvt_i32_f32 %7, %3\n"
                     "v_cvt_i32_f32 %4, %0\n v_cvt_i32_f32 %5, %1\n v_cvt_i32_f32 %6, %2\n v_cvt_i32_f32 %7, %3\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = y0 + y1 + y2 + y3;
}
__global__ void k_mad24(float *out, float a) {
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; int b = (int)a;
    for (int i = 0; i < N_IT; i++) {
        asm volatile("v_mad_u32_u24 %0, %8, %0, %0\n v_mad_u32_u24 %1, %8, %1, %1\n v_mad_u32_u24 %2, %8, %2, %2\n v_mad_u32_u24 %3, %8, %3, %3\n"
                     "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_lds(float *out, float a) {
    __shared__ unsigned short tile[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = i;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    unsigned s = 0;
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { h = h * 1664525u + 1013904223u; s += tile[(h >> 10) & 16383]; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float run(F f, const char *name, double ops_per_thread_iter, int blocks, int threads, float *d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f<<<blocks, threads>>>(d, 1.0001f); hipDeviceSynchronize();
    hipEventRecord(a); f<<<blocks, threads>>>(d, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winst = (double)blocks * threads / 64 * N_IT * ops_per_thread_iter;
    printf("%-10s %.3f ms  %.1f G wave-instr/s  => %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, winst / ms / 1e6,
           (1024.0 * 2.4e9) / (winst / (ms * 1e-3)));
    return ms;
}
int main() {
    float *d; hipMalloc(&d, 4 * 2048 * 1024);
    int blocks = 2048, threads = 512;   // 8 waves/SIMD
    run(k_mul, "v_mul", 8, blocks, threads, d);
    run(k_pkmul, "v_pk_mul", 8, blocks, threads, d);
    run(k_cvt, "v_cvt", 8, blocks, threads, d);
    run(k_mad24, "mad24/lshl", 8, blocks, threads, d);
    run(k_lds, "lds_u16", 8, blocks, threads, d);
    return 0;
}
