// Microbenchmark (tools/ubench): what does rocprofv3's FETCH_SIZE report for the blend kernels' access pattern?
// The guide (MI355X_MICROARCH.md, HBM) calibrates it for wide coalesced streaming reads only (exactly 1/2 of the bytes);
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The blend kernels
// stage their tile lists by GATHER: lane i loads the 112-byte record of a random surfel with seven 16-byte loads.
//   k_gather128  every lane loads one 128-byte-aligned 128-byte record (8 x 16 B) chosen by a PERMUTATION: every cache
//                line of the array is read exactly once -> known bytes = records x 128
//   k_gather112  the product's record: 112-byte stride, 7 x 16 B -- records straddle lines; known LINE bytes = the
//                array's size rounded to lines (every record is read once; a line shared by two records may be fetched
//                by two XCDs)
//   k_stream     lane i loads 16 B at consecutive addresses: the guide's calibrated case (expect 1/2)
// Arrays are 512 MB (beyond the 256 MiB Infinity Cache and every L2).  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- tools/ubench/fetch_calib
// and divide the known bytes by the counter (tools/fetch_calib.sh does both).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ __launch_bounds__(256) void k_gather128(const float4* __restrict__ a, const uint32_t* __restrict__ perm, size_t n, float4* sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* r = a + (size_t)perm[i] * 8;
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float4 v = r[k];
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    if (s.x == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_gather112(const float* __restrict__ a, const uint32_t* __restrict__ perm, size_t n, float4* sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4* r = reinterpret_cast<const float4*>(a + (size_t)perm[i] * 28);
    float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const float4 v = r[k];
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    if (s.x == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ a, size_t n16, float4* sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    const float4 v = a[i];
    if (v.x == 12345.678f) sink[0] = v;
}

int main()
{
    const size_t BYTES = 512ull << 20;
    float* a;
    float4* sink;
    hipMalloc(&a, BYTES);
    hipMalloc(&sink, 64);
    hipMemset(a, 0, BYTES);
    const size_t n128 = BYTES / 128, n112 = BYTES / 112;
    std::vector<uint32_t> p(n112);
    std::iota(p.begin(), p.end(), 0u);
    std::mt19937 rng(1);
    uint32_t *d128, *d112;
    hipMalloc(&d128, n128 * 4);
    hipMalloc(&d112, n112 * 4);
    std::shuffle(p.begin(), p.begin() + n128, rng);
    hipMemcpy(d128, p.data(), n128 * 4, hipMemcpyHostToDevice);
    std::iota(p.begin(), p.end(), 0u);
    std::shuffle(p.begin(), p.end(), rng);
    hipMemcpy(d112, p.data(), n112 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_gather128, dim3((n128 + 255) / 256), dim3(256), 0, 0, (const float4*)a, d128, n128, sink);
        hipLaunchKernelGGL(k_gather112, dim3((n112 + 255) / 256), dim3(256), 0, 0, a, d112, n112, sink);
        hipLaunchKernelGGL(k_stream, dim3((BYTES / 16 + 255) / 256), dim3(256), 0, 0, (const float4*)a, BYTES / 16, sink);
    }
    hipDeviceSynchronize();
    printf("known bytes per launch: k_gather128 %zu (+ %zu of indices), k_gather112 %zu (+ %zu), k_stream %zu\n", n128 * 128, n128 * 4,
           n112 * 112, n112 * 4, BYTES);
    return 0;
}
