// wave_utils.h -- wave64 / workgroup scan helpers shared by the per-surfel and binning kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace surfel {

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// Exclusive scan over a 256-thread workgroup (4 wave64); returns this thread's exclusive prefix and
// the workgroup total.  s_wave: 4 words of LDS; contains a barrier.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t t = s_wave[w];
        if (w < wave) base += t;
    }
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + inc - v;
}

}  // namespace surfel
