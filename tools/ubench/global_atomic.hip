// Microbenchmark (tools/ubench): cost of global float atomics (no return) by how the lanes of one wave instruction
// are laid over the 80-byte gradient records -- decides whether the backward blend may send the per-row totals of its
// reduce-scatter straight to memory instead of combining them in LDS first.
//   pattern 0: 64 lanes = 64 consecutive floats from a record boundary (the workgroup flush of blend_bwd today)
//   pattern 1: 4 rows of 16 lanes, every row 16 consecutive floats of its own random record
//   pattern 2: 64 lanes, 64 random records, one float each
//   pattern 3: like 1, the 4 records of an instruction drawn from a window of 64 neighbouring records
// `work` = dependent FMA chains between two atomic instructions (0: atomics only); `noatomic` times the work alone.
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int RECORDS = 200000, REC = 20, ITER = 137;

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void k_atomic(float* acc, float* sink, int pattern, int work, int noatomic)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave_id = (blockIdx.x * 256 + threadIdx.x) >> 6;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = 1.0f + lane + i;
    for (int it = 0; it < ITER; it++) {
        for (int w = 0; w < work; w++) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = v[i] * 0.999f + 0.001f;
        }
        const uint32_t h = mix(wave_id * 977u + it);
        size_t off;
        if (pattern == 0) off = (size_t)(h % (RECORDS - 4)) * REC + lane;
        else if (pattern == 1) off = (size_t)(mix(h + (lane >> 4)) % RECORDS) * REC + (lane & 15);
        else if (pattern == 2) off = (size_t)(mix(h + lane) % RECORDS) * REC + (lane & 15);
        else off = (size_t)((h % (RECORDS - 64)) + (mix(h + (lane >> 4)) & 63)) * REC + (lane & 15);
        if (!noatomic) atomicAdd(acc + off, v[it & 7]);
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += v[i];
    if (s == 12345.678f) sink[0] = s;
}

int main()
{
    float *acc, *sink;
    hipMalloc(&acc, (size_t)RECORDS * REC * 4);
    hipMalloc(&sink, 64);
    hipMemset(acc, 0, (size_t)RECORDS * REC * 4);
    const int blocks = 2048;
    const double insts = (double)blocks * 4 * ITER;
    for (int work : {0, 18}) {
        for (int pattern = -1; pattern < 4; pattern++) {
            if (pattern < 0 && work == 0) continue;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            const int p = pattern < 0 ? 0 : pattern, na = pattern < 0;
            hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, acc, sink, p, work, na);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, acc, sink, p, work, na);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            printf("work %2d (%3d VALU/iter) pattern %2d: %8.1f us for %.2f M wave atomic instructions (%.0f M floats) -> %.2f ns each\n",
                   work, work * 8, pattern, ms * 1e3, insts / 1e6, insts * 64 / 1e6, ms * 1e6 / insts);
        }
    }
    return 0;
}
