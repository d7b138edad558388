// Microbenchmark (tools/ubench): LDS pipe cost of the accesses a "row walk" of the blend kernels would make
// (DESIGN.md §4.3): ds_read_b128 with one address per wave (today's broadcast record reads), one address per 16-lane
// row, or one per lane; and the non-atomic read-modify-write of 16 lanes (ds_read_b32 + ds_write_b32) next to
// ds_add_f32 of 16 lanes.  All four waves of every workgroup hammer the pipe; 4 workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int ITER = 4096;

__global__ __launch_bounds__(256) void k_read128(float* out, int mode)
{
    __shared__ float4 s[1280];
    for (int i = threadIdx.x; i < 1280; i += 256) s[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int j = mode == 0 ? wave : (mode == 1 ? wave * 4 + (lane >> 4) : lane + wave);
    float acc = 0.f;
    for (int it = 0; it < ITER; it++) {
        // five 16-byte pieces of an 80-byte record, like the blend kernels
        const float4 a = s[j * 5 + 0], b = s[j * 5 + 1], c = s[j * 5 + 2], d = s[j * 5 + 3], e = s[j * 5 + 4];
        acc += a.x + b.y + c.z + d.w + e.x;
        j = (j + 7 + (__float_as_int(acc) & 1)) & 127;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_rmw(float* out, int mode)
{
    __shared__ float s[4 * 64 * 20];
    for (int i = threadIdx.x; i < 4 * 64 * 20; i += 256) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int j = (lane >> 4) * 3;
    float v = 1.0f + lane;
    for (int it = 0; it < ITER; it++) {
        float* p = &s[(wave * 64 + j) * 20 + (lane & 15)];
        if (mode == 0) {  // one 16-lane float atomic per row, all rows in one instruction
            atomicAdd(p, v);
        } else if (mode == 1) {  // one plain RMW for the 64 lanes
            *p += v;
        } else {  // four row-masked RMWs in sequence
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if ((lane >> 4) == r) *p += v;
                __builtin_amdgcn_wave_barrier();
            }
        }
        j = (j + 5) & 63;
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x];
}

template <typename K>
static void run(const char* what, K kernel, float* out, int mode, int lds_insts_per_iter)
{
    const int blocks = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double iters_per_cu = (double)blocks * 4 * ITER / 256.0;  // wave iterations per CU
    printf("%-58s %8.3f ms -> %6.1f cycles per wave iteration per CU at 2.4 GHz (%d LDS instructions)\n", what, ms,
           ms * 1e6 * 2.4 / iters_per_cu, lds_insts_per_iter);
}

int main()
{
    float* out;
    hipMalloc(&out, 1024 * 256 * 4);
    run("5 x ds_read_b128, one address per wave", k_read128, out, 0, 5);
    run("5 x ds_read_b128, one address per 16-lane row", k_read128, out, 1, 5);
    run("5 x ds_read_b128, one address per lane", k_read128, out, 2, 5);
    run("ds_add_f32, 64 lanes (4 rows x 16 floats)", k_rmw, out, 0, 1);
    run("plain RMW, 64 lanes at once", k_rmw, out, 1, 2);
    run("plain RMW, four row-masked steps", k_rmw, out, 2, 8);
    return 0;
}
