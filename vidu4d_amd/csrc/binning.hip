// binning.hip -- on-device stable LSD radix sort of the (tile | depth) keys and tile-range
// identification.  gfx950 only.
//
// Replaces cub::DeviceRadixSort::SortPairs(keys, values, R, 0, 32 + bit) and identifyTileRanges
// (/root/reference/gs/submodules/diff-surfel-rasterization/cuda_rasterizer/rasterizer_impl.cu:
// 304-309, :116-138).  The result must be the *stable* ascending order on the low 32+bit key bits:
// ties (same tile, identical depth bits) keep emission order, i.e. ascending surfel id.
//
// Structure per 8-bit pass (three launches, all sized from the buffer capacity and guarded by the
// device-side element count so that no host sync is needed):
//   hist     each 256-thread workgroup counts the digits of its 4096-key tile in LDS
//   scan     one workgroup turns the [digit][workgroup] counts into global offsets (digit-major
//            exclusive scan)
//   scatter  each wave64 ranks its keys with ballot-based digit matching: for every 64 consecutive
//            keys the lanes holding the same digit are found with 8 ballots, the lane's rank inside
//            that group is a popcount of the lower lanes, and a per-wave LDS counter carries the
//            running count across the 16 rounds.  Waves own consecutive key ranges, so
//            offset(digit, workgroup) + sum(lower waves) + rank is the stable destination.
#include "surfel_state.h"

namespace surfel {

__device__ __forceinline__ uint32_t digit_of(uint64_t key, int shift) { return (uint32_t)(key >> shift) & (RADIX - 1); }

__global__ __launch_bounds__(SORT_BLOCK) void sort_hist_kernel(const uint32_t* num_ptr, int64_t capacity,
                                                               const uint64_t* keys, uint32_t* counts, int nblocks,
                                                               int shift)
{
    __shared__ uint32_t s_hist[RADIX];
    int64_t n = (int64_t)*num_ptr;
    if (n > capacity) n = 0;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll 4
    for (int k = 0; k < SORT_ITEMS; k++) {
        const int64_t i = base + (int64_t)k * SORT_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[digit_of(keys[i], shift)], 1u);
    }
    __syncthreads();
    counts[(size_t)threadIdx.x * nblocks + blockIdx.x] = s_hist[threadIdx.x];
}

// Exclusive scan of counts[RADIX * nblocks] in place (digit-major), one 1024-thread workgroup.
// Coalesced: the array is walked in 1024-wide strips, each strip scanned with wave64 shuffles +
// a 16-entry LDS step, and a running carry links the strips (total <= 256 * 1024 entries at 4M
// pairs, i.e. <= 256 strips).
__global__ __launch_bounds__(1024) void sort_scan_kernel(uint32_t* counts, int total)
{
    __shared__ uint32_t s_part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < total; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < total ? counts[i] : 0;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t t = s_part[w];
            if (w < wave) wbase += t;
            tot += t;
        }
        if (i < total) counts[i] = carry + wbase + inc - v;
        carry += tot;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SORT_BLOCK) void sort_scatter_kernel(const uint32_t* num_ptr, int64_t capacity,
                                                                  const uint64_t* keys_in, const uint32_t* vals_in,
                                                                  uint64_t* keys_out, uint32_t* vals_out,
                                                                  const uint32_t* offsets, int nblocks, int shift)
{
    __shared__ uint32_t s_cnt[4][RADIX];   // running / final per-wave digit counts
    __shared__ uint32_t s_base[4][RADIX];  // destination base per (wave, digit)
    int64_t n = (int64_t)*num_ptr;
    if (n > capacity) n = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * RADIX; i += SORT_BLOCK) (&s_cnt[0][0])[i] = 0;
    __syncthreads();

    // wave w owns keys [base + w*1024, base + (w+1)*1024), 64 consecutive keys per round
    const int64_t wbase = (int64_t)blockIdx.x * SORT_TILE + (int64_t)wave * (SORT_TILE / 4);
    uint64_t key[SORT_ITEMS];
    uint32_t val[SORT_ITEMS];
    uint32_t rank[SORT_ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const int64_t i = wbase + (int64_t)k * 64 + lane;
        const bool valid = i < n;
        key[k] = valid ? keys_in[i] : ~0ull;
        val[k] = valid ? vals_in[i] : 0u;
        const uint32_t d = digit_of(key[k], shift);
        // lanes holding the same digit (invalid lanes form their own group via the extra ballot)
        uint64_t peers = __ballot(valid);
        if (!valid) peers = ~peers;
#pragma unroll
        for (int b = 0; b < RADIX_BITS; b++) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t cnt = (uint32_t)__popcll(peers);
        uint32_t old = 0;
        if (valid) {
            old = s_cnt[wave][d];
            if (before == 0) s_cnt[wave][d] = old + cnt;  // group leader carries the count forward
        }
        rank[k] = old + before;
    }
    __syncthreads();
    {
        const int d = threadIdx.x;  // one digit per thread
        uint32_t run = offsets[(size_t)d * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            s_base[w][d] = run;
            run += s_cnt[w][d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; k++) {
        const int64_t i = wbase + (int64_t)k * 64 + lane;
        if (i < n) {
            const uint32_t d = digit_of(key[k], shift);
            const uint32_t pos = s_base[wave][d] + rank[k];
            keys_out[pos] = key[k];
            vals_out[pos] = val[k];
        }
    }
}

int launch_radix_sort(const GeomState& g, const BinState& b, int64_t capacity, int passes, hipStream_t stream)
{
    const int nb = b.sort_blocks;
    int side = 0;
    if (nb <= 0) return passes & 1;
    for (int p = 0; p < passes; p++) {
        const int shift = p * RADIX_BITS;
        hipLaunchKernelGGL(sort_hist_kernel, dim3(nb), dim3(SORT_BLOCK), 0, stream, &g.hdr->num_rendered, capacity,
                           b.keys[side], b.counts, nb, shift);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, stream, b.counts, RADIX * nb);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(nb), dim3(SORT_BLOCK), 0, stream, &g.hdr->num_rendered,
                           capacity, b.keys[side], b.vals[side], b.keys[side ^ 1], b.vals[side ^ 1], b.counts, nb,
                           shift);
        side ^= 1;
    }
    return side;
}

// rasterizer_impl.cu:116-138 (ranges were zeroed by the scan kernel).
__global__ void tile_ranges_kernel(const uint32_t* num_ptr, int64_t capacity, const uint64_t* keys, uint32_t* ranges)
{
    int64_t n = (int64_t)*num_ptr;
    if (n > capacity) n = 0;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const uint32_t cur = (uint32_t)(keys[idx] >> 32);
    if (idx == 0)
        ranges[2 * cur] = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[idx - 1] >> 32);
        if (cur != prev) {
            ranges[2 * prev + 1] = (uint32_t)idx;
            ranges[2 * cur] = (uint32_t)idx;
        }
    }
    if (idx == n - 1) ranges[2 * cur + 1] = (uint32_t)n;
}

void launch_tile_ranges(const GeomState& g, const uint64_t* sorted_keys, int64_t capacity, uint32_t* ranges,
                        hipStream_t stream)
{
    if (capacity <= 0) return;
    const int blocks = (int)((capacity + 255) / 256);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(blocks), dim3(256), 0, stream, &g.hdr->num_rendered, capacity,
                       sorted_keys, ranges);
}

}  // namespace surfel
