// Microbenchmark (tools/ubench): does a wave64 VALU instruction on gfx950 cost less when half (or three quarters) of its
// lanes are switched off in EXEC?  (A SIMD-32 issues a wave64 instruction in two passes of 32 lanes.)  The answer decides
// whether concentrating the contributing pixels of a trip of the blend backward in one half of the wave is worth anything.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITER = 4096;

template <int ACTIVE>
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i;
    if ((int)(threadIdx.x & 63) < ACTIVE) {
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = __builtin_fmaf(v[i], a, b);
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// upper half only
__global__ __launch_bounds__(256) void k_fma_hi(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i;
    if ((int)(threadIdx.x & 63) >= 32) {
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = __builtin_fmaf(v[i], a, b);
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ACTIVE>
__global__ __launch_bounds__(256) void k_exp(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = (threadIdx.x + i) * 1e-3f;
    if ((int)(threadIdx.x & 63) < ACTIVE) {
        for (int it = 0; it < ITER; it++)
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = __builtin_amdgcn_exp2f(v[i]) * a;
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void run(K kern, float* out, int blocks, const char* name, double ops)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * ITER * ops / 1024.0;
    printf("%-22s blocks %5d  %8.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks, ms,
           ms * 1e6 / per_simd * 2.4);
}

int main()
{
    float* out;
    hipMalloc(&out, 8192 * 256 * 4);
    for (int blocks : {1024, 2048}) {
        run(k_fma<64>, out, blocks, "v_fma 64 lanes", 16);
        run(k_fma<32>, out, blocks, "v_fma lanes 0-31", 16);
        run(k_fma_hi, out, blocks, "v_fma lanes 32-63", 16);
        run(k_fma<16>, out, blocks, "v_fma lanes 0-15", 16);
        run(k_fma<1>, out, blocks, "v_fma lane 0", 16);
        run(k_exp<64>, out, blocks, "exp2+mul 64 lanes", 32);
        run(k_exp<32>, out, blocks, "exp2+mul lanes 0-31", 32);
        run(k_exp<16>, out, blocks, "exp2+mul lanes 0-15", 32);
    }
    return 0;
}
