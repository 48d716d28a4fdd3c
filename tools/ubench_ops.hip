// micro-benchmark: per-instruction issue cost (cycles per wave-instruction per SIMD) on gfx950, 8 waves/SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_IT 2048
#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define DEF_KERNEL(NAME, ASMSTR)                                                                      \
__global__ void NAME(float *out, float a) {                                                           \
    float x0 = threadIdx.x + 1.5f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    int s1 = 3;                                                                                       \
    asm volatile("s_mov_b32 %0, 3" : "=s"(s1));                                                       \
    for (int i = 0; i < N_IT; i++) {                                                                  \
        asm volatile(ASMSTR : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(s1)); \
    }                                                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;              \
}
#define R8(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define R8b(op) op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n"
#define R8c(op) op " %0, %0, %8, %0\n" op " %1, %1, %8, %1\n" op " %2, %2, %8, %2\n" op " %3, %3, %8, %3\n" op " %4, %4, %8, %4\n" op " %5, %5, %8, %5\n" op " %6, %6, %8, %6\n" op " %7, %7, %8, %7\n"
#define R8s(op) op " %0, %0, %9, %0\n" op " %1, %1, %9, %1\n" op " %2, %2, %9, %2\n" op " %3, %3, %9, %3\n" op " %4, %4, %9, %4\n" op " %5, %5, %9, %5\n" op " %6, %6, %9, %6\n" op " %7, %7, %9, %7\n"
DEF_KERNEL(k_mul, R8b("v_mul_f32"))
DEF_KERNEL(k_add, R8b("v_add_f32"))
DEF_KERNEL(k_fma, R8c("v_fma_f32"))
DEF_KERNEL(k_cvt_i32, R8("v_cvt_i32_f32"))
DEF_KERNEL(k_cvt_u32, R8("v_cvt_u32_f32"))
DEF_KERNEL(k_trunc, R8("v_trunc_f32"))
DEF_KERNEL(k_floor, R8("v_floor_f32"))
DEF_KERNEL(k_mov, R8("v_mov_b32"))
DEF_KERNEL(k_addu, R8b("v_add_u32"))
DEF_KERNEL(k_lshl, R8b("v_lshlrev_b32"))
DEF_KERNEL(k_mul24, R8b("v_mul_u32_u24"))
DEF_KERNEL(k_mad24, R8c("v_mad_u32_u24"))
DEF_KERNEL(k_mad24s, R8s("v_mad_u32_u24"))
DEF_KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n")
DEF_KERNEL(k_add3, R8c("v_add3_u32"))
DEF_KERNEL(k_readlane, "v_readlane_b32 s20, %0, %9\n v_readlane_b32 s21, %1, %9\n v_readlane_b32 s22, %2, %9\n v_readlane_b32 s23, %3, %9\n v_readlane_b32 s20, %4, %9\n v_readlane_b32 s21, %5, %9\n v_readlane_b32 s22, %6, %9\n v_readlane_b32 s23, %7, %9\n")
DEF_KERNEL(k_mulsgpr, "v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s21, %1\n v_mul_f32 %2, s22, %2\n v_mul_f32 %3, s23, %3\n v_mul_f32 %4, s20, %4\n v_mul_f32 %5, s21, %5\n v_mul_f32 %6, s22, %6\n v_mul_f32 %7, s23, %7\n")
DEF_KERNEL(k_cvtpk, "v_cvt_pk_u16_u32 %0, %0, %1\n v_cvt_pk_u16_u32 %1, %1, %2\n v_cvt_pk_u16_u32 %2, %2, %3\n v_cvt_pk_u16_u32 %3, %3, %4\n v_cvt_pk_u16_u32 %4, %4, %5\n v_cvt_pk_u16_u32 %5, %5, %6\n v_cvt_pk_u16_u32 %6, %6, %7\n v_cvt_pk_u16_u32 %7, %7, %0\n")

// pure LDS gather rate: addresses precomputed per lane (random within a 16 KB tile), 8 independent reads per iteration
template <int MODE>
__global__ void k_lds(float *out, float a) {
    __shared__ unsigned short tile[8192 * 2];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = i;
    __syncthreads();
    unsigned h = (threadIdx.x * 2654435761u + blockIdx.x * 40503u);
    unsigned ad[8];
    for (int k = 0; k < 8; k++) {
        h = h * 1664525u + 1013904223u;
        unsigned r = (h >> 9) & 16383;
        if (MODE == 1) r = (threadIdx.x & 63) * 2 + k * 128;          // conflict-free: consecutive halfwords
        if (MODE == 2) r = ((h >> 9) & 31) * 136 + ((h >> 20) & 31);  // jitter-like: 32 rows x 32 cols, pitch 136 px
        ad[k] = r * 2;
    }
    unsigned s = 0;
    const char *base = (const char *)tile;
    for (int i = 0; i < N_IT; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) s += *(const unsigned short *)(base + ad[k]);
        asm volatile("" : "+v"(s));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> void run(F f, const char *name, float *d) {
    int blocks = 2048, threads = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f<<<blocks, threads>>>(d, 1.0001f); hipDeviceSynchronize();
    hipEventRecord(a); f<<<blocks, threads>>>(d, 1.0001f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winst = (double)blocks * threads / 64 * N_IT * 8;
    printf("%-12s %.3f ms  => %.2f cyc/wave-instr/SIMD @2.4GHz  (%.2f per CU for LDS)\n", name, ms, (1024.0 * 2.4e9) / (winst / (ms * 1e-3)),
           (256.0 * 2.4e9) / (winst / (ms * 1e-3)));
}
int main() {
    float *d; hipMalloc(&d, 4 * 2048 * 1024);
    run(k_mul, "v_mul_f32", d); run(k_add, "v_add_f32", d); run(k_fma, "v_fma_f32", d); run(k_mulsgpr, "v_mul sgpr", d);
    run(k_cvt_i32, "cvt_i32_f32", d); run(k_cvt_u32, "cvt_u32_f32", d); run(k_trunc, "v_trunc_f32", d); run(k_floor, "v_floor_f32", d);
    run(k_mov, "v_mov_b32", d); run(k_addu, "v_add_u32", d); run(k_lshl, "v_lshlrev", d); run(k_mul24, "v_mul_u32_u24", d);
    run(k_mad24, "v_mad_u32_u24", d); run(k_mad24s, "mad24 sgpr", d); run(k_lshladd, "v_lshl_add", d); run(k_add3, "v_add3_u32", d);
    run(k_readlane, "v_readlane", d); run(k_cvtpk, "cvt_pk_u16", d);
    run(k_lds<0>, "lds rand", d); run(k_lds<1>, "lds linear", d); run(k_lds<2>, "lds jitter", d);
    return 0;
}
