// Microbenchmark (tools/ubench): issue cost of wave64 VALU instructions on gfx950, to decide how the blend
// kernels should be written.  Each kernel runs ITER iterations of 16 independent chains per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_fmaf(v[i], a, b);
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b)
{
    float2v v[16];
    for (int i = 0; i < 16; i++) v[i] = (float2v){(float)threadIdx.x + i, (float)i};
    const float2v av = {a, a}, bv = {b, b};
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_elementwise_fma(v[i], av, bv);
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_exp(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = (threadIdx.x + i) * 1e-3f;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_amdgcn_exp2f(v[i]) * a;
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_rcp(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = (threadIdx.x + i) * 1e-3f + 1.0f;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_amdgcn_rcpf(v[i]);
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_dpp(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < 16; i++)
            v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0xB1, 0xF, 0xF, true));
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_swap(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 1]), false, false);
            v[i] = __uint_as_float(r[0]);
            v[i + 1] = __uint_as_float(r[1]);
        }
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cndmask(float* out, float a, float b)
{
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < ITER; it++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = v[i] > a ? v[i] - b : v[i] + b;  // cmp + sub + add + cndmask-ish
    float s = 0;
    for (int i = 0; i < 16; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static double run(K kern, float* out, int blocks, const char* name, double ops_per_iter_per_lane)
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
    // waves per SIMD = blocks*4 / 1024
    const double wave_insts = (double)blocks * 4 * ITER * ops_per_iter_per_lane;  // wave-instructions in total
    const double per_simd = wave_insts / 1024.0;
    printf("%-10s blocks %5d  %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, blocks, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    return ms;
}

int main()
{
    float* out;
    hipMalloc(&out, 8192 * 256 * 4);
    for (int blocks : {256, 1024, 2048}) {
        run(k_fma, out, blocks, "v_fma", 16);
        run(k_pkfma, out, blocks, "v_pk_fma", 16);
        run(k_exp, out, blocks, "exp2+mul", 32);
        run(k_rcp, out, blocks, "v_rcp", 16);
        run(k_dpp, out, blocks, "dpp_add", 16);
        run(k_swap, out, blocks, "perm32swap", 8);
        run(k_cndmask, out, blocks, "cmp+sel", 48);
    }
    return 0;
}
