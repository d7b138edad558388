// Microbenchmark (tools/ubench): throughput of ds_add_f32 (no return) by how many lanes of a wave share an
// address.  Decides whether per-lane LDS accumulation of gradients is viable in the backward blend.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITER = 2048;

__global__ __launch_bounds__(256) void k_lds_add(float* out, int share, int stride)
{
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lanes [k*share, (k+1)*share) of a wave share one address; distinct groups go to different banks
    const int base = wave * 1024 + (lane / share) * stride;
    float v = 1.0f + lane;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int c = 0; c < 16; c++) atomicAdd(&s[base + c], v);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x];
}

int main()
{
    float* out;
    hipMalloc(&out, 4096 * 256 * 4);
    const int blocks = 1024;
    for (int share : {1, 4, 16, 64}) {
        for (int stride : {1, 17, 20}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(k_lds_add, dim3(blocks), dim3(256), 0, 0, out, share, stride);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_lds_add, dim3(blocks), dim3(256), 0, 0, out, share, stride);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts_per_cu = (double)blocks * 4 * ITER * 16 / 256.0;  // wave-instructions per CU
            printf("share %2d stride %2d: %8.3f ms  -> %.1f cycles per ds_add_f32 wave-instruction per CU (2.4 GHz)\n", share,
                   stride, ms, ms * 1e6 * 2.4 / insts_per_cu);
        }
    }
    return 0;
}
