// Microbenchmark (tools/ubench): issue cost per wave64 instruction on gfx950 by instruction KIND (inline asm so that the
// compiler cannot fuse or drop anything): v_fma_f32, v_add_f32, v_mul_f32, v_mov_b32, v_cndmask_b32 (vcc and SGPR-pair
// mask), v_cmp (to vcc), v_sub + v_fmac (a dependent pair), and the mix of one blend-backward trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITER = 2048;

#define BODY16(ASM)                                                                                               \
    float v[16];                                                                                                  \
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i;                                                          \
    for (int it = 0; it < ITER; it++) {                                                                           \
        _Pragma("unroll") for (int i = 0; i < 16; i++) ASM;                                                       \
    }                                                                                                             \
    float s = 0;                                                                                                  \
    for (int i = 0; i < 16; i++) s += v[i];                                                                       \
    out[blockIdx.x * 256 + threadIdx.x] = s;

__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) { BODY16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b))) }
__global__ __launch_bounds__(256) void k_fmac(float* out, float a, float b) { BODY16(asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b))) }
__global__ __launch_bounds__(256) void k_add(float* out, float a, float b) { BODY16(asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(a))) }
__global__ __launch_bounds__(256) void k_mul(float* out, float a, float b) { BODY16(asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(a))) }
__global__ __launch_bounds__(256) void k_mov(float* out, float a, float b) { BODY16(asm volatile("v_mov_b32_e32 %0, %1" : "+v"(v[i]) : "v"(a))) }
__global__ __launch_bounds__(256) void k_cnd(float* out, float a, float b) { BODY16(asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : "vcc")) }
__global__ __launch_bounds__(256) void k_cnd64(float* out, float a, float b)
{
    unsigned long long m = 0x5555555555555555ull;
    BODY16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "s"(m)))
}
__global__ __launch_bounds__(256) void k_cmp(float* out, float a, float b) { BODY16(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(v[i]), "v"(a) : "vcc")) }
__global__ __launch_bounds__(256) void k_mul_lit(float* out, float a, float b) { BODY16(asm volatile("v_mul_f32_e32 %0, 0x3fb8aa3b, %0" : "+v"(v[i]))) }
__global__ __launch_bounds__(256) void k_fma_sgpr(float* out, float a, float b) { BODY16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(a), "v"(b))) }
__global__ __launch_bounds__(256) void k_nop(float* out, float a, float b) { BODY16(asm volatile("v_add_f32_e32 %0, %0, %1\n s_nop 1" : "+v"(v[i]) : "v"(a))) }
// 1 chain per lane: fully dependent
__global__ __launch_bounds__(256) void k_dep(float* out, float a, float b)
{
    float x = threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <typename K>
static void run(K kern, float* out, int blocks, const char* name, double ops)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * ITER * ops / 1024.0;
    printf("%-26s waves/SIMD %2d  %8.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks * 4 / 1024, ms,
           ms * 1e6 / per_simd * 2.4);
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 8192 * 256 * 4);
    for (int blocks : {256, 1024, 1536}) {
        run(k_fma, out, blocks, "v_fma_f32 (VOP3)", 16);
        run(k_fmac, out, blocks, "v_fmac_f32_e32", 16);
        run(k_add, out, blocks, "v_add_f32_e32", 16);
        run(k_mul, out, blocks, "v_mul_f32_e32", 16);
        run(k_mul_lit, out, blocks, "v_mul_f32 literal", 16);
        run(k_fma_sgpr, out, blocks, "v_fma_f32 sgpr operand", 16);
        run(k_mov, out, blocks, "v_mov_b32", 16);
        run(k_cnd, out, blocks, "v_cndmask vcc", 16);
        run(k_cnd64, out, blocks, "v_cndmask_e64 sgpr mask", 16);
        run(k_cmp, out, blocks, "v_cmp_lt_f32 -> vcc", 16);
        run(k_nop, out, blocks, "v_add + s_nop 1", 16);
        run(k_dep, out, blocks, "v_fma dependent chain", 16);
    }
    return 0;
}
