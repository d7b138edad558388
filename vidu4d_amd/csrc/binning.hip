// binning.hip -- tile binning and in-tile depth ordering of the (surfel, tile) pairs.  gfx950 only.
//
// What the reference computes (/root/reference/gs/submodules/diff-surfel-rasterization/
// cuda_rasterizer/rasterizer_impl.cu): InclusiveSum over tiles_touched (:278), duplicateWithKeys
// (:70-111) emitting 64-bit (tile << 32 | depth bits) keys in surfel-id order, a stable
// cub::DeviceRadixSort over 32+ceil(log2 tiles) bits (:304-309) and identifyTileRanges (:116-138).
// Result: point_list = surfel ids grouped by tile, ascending depth bits inside a tile, ties in
// ascending surfel id; ranges[tile] = [start, end).
//
// The same result, arranged for the MI355X (six 8-bit passes over a ~0.6M-pair list are ~18
// dependent launches of mostly launch latency, and every pass drags all pairs through HBM):
//   1. preprocess counts pairs per tile.  Device-scope atomics on MI355X are served at the memory
//      side of the fabric (the XCD L2s are not coherent), ~0.25 ns each at best and far worse on a hot
//      address, so the default "grouped" path uses none: a 1024-thread workgroup histograms the pairs
//      of its <= 4096 consecutive surfels in LDS and writes one row of per-(group, tile) counts; a
//      column prefix over the <= 256 groups then gives every group its private sub-range of every
//      tile segment.  (Images with more than 16k tiles fall back to sliced global atomics.);
//   2. tile_scan (one workgroup): exclusive scan over the tiles -> ranges, total -> num_rendered.
//      This IS the most-significant-digit pass of the reference's sort, done as a counting sort;
//   3. emit: every pair is written once, straight into its tile's segment, as (depth bits << 32 | id);
//      the order inside a segment is whatever the atomics give;
//   4. tile_sort: one workgroup per tile orders its segment by (depth bits, surfel id) with an LSD
//      radix sort held entirely in LDS (8-bit digits; ranks from wave64 ballot digit matching +
//      popcount prefix, as in a device-wide radix sort, but the ping-pong buffers are LDS arrays).
//      Passes whose digit is identical for the whole segment (upper id bytes, the depth exponent
//      byte) are detected from the histogram and skipped.  Segments longer than TILE_SORT_CAP run
//      the same code with the ping-pong in global memory.
// Ordering by (depth, id) equals the reference's stable sort because a surfel is emitted at most
// once per tile, in id order.  No pass ever moves a pair across tiles, so pairs cross HBM three
// times (emit, sort in, sort out) instead of thirteen.
#include <algorithm>

#include "surfel_state.h"
#include "wave_utils.h"

namespace surfel {

// Exclusive scan over the (tile, slice) counters in tile-major order, 16 per thread (one tile).
__global__ __launch_bounds__(1024) void tile_scan_kernel(GeomState g, ImageState img, int num_tiles)
{
    __shared__ uint32_t s_part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < num_tiles; base += 1024) {
        const int t = base + threadIdx.x;
        uint32_t cnt[TILE_SLICES];
        uint32_t c = 0;
        if (t < num_tiles) {
            const uint4* p = reinterpret_cast<const uint4*>(img.tile_count + (size_t)t * TILE_SLICES);
#pragma unroll
            for (int k = 0; k < TILE_SLICES / 4; k++) {
                const uint4 v = p[k];
                cnt[4 * k] = v.x;
                cnt[4 * k + 1] = v.y;
                cnt[4 * k + 2] = v.z;
                cnt[4 * k + 3] = v.w;
                c += v.x + v.y + v.z + v.w;
            }
        }
        const uint32_t inc = wave_inclusive_scan(c, lane);
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t v = s_part[w];
            if (w < wave) wbase += v;
            tot += v;
        }
        if (t < num_tiles) {
            const uint32_t start = carry + wbase + inc - c;
            // empty tiles keep the reference's memset value (0, 0) (rasterizer_impl.cu:311)
            img.ranges[2 * t] = c ? start : 0u;
            img.ranges[2 * t + 1] = c ? start + c : 0u;
            uint32_t run = start;
#pragma unroll
            for (int k = 0; k < TILE_SLICES; k++) {
                img.tile_base[(size_t)t * TILE_SLICES + k] = run;
                run += cnt[k];
                img.tile_count[(size_t)t * TILE_SLICES + k] = 0;  // becomes the emit cursor
            }
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        g.hdr->num_rendered = carry;
        g.hdr->overflow = 0;
    }
}

// Grouped path, step 2: exclusive scan of the tile totals -> ranges, total -> num_rendered
// (one 1024-thread workgroup).
__device__ __forceinline__ void tile_totals_scan(GeomState g, ImageState img, int num_tiles, uint32_t* s_part)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (int base = 0; base < num_tiles; base += 1024) {
        const int t = base + threadIdx.x;
        const uint32_t c = t < num_tiles ? img.tile_count[t] : 0;
        const uint32_t inc = wave_inclusive_scan(c, lane);
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t v = s_part[w];
            if (w < wave) wbase += v;
            tot += v;
        }
        if (t < num_tiles) {
            const uint32_t start = carry + wbase + inc - c;
            img.ranges[2 * t] = c ? start : 0u;  // empty tiles: the reference's memset (0, 0)
            img.ranges[2 * t + 1] = c ? start + c : 0u;
        }
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        g.hdr->num_rendered = carry;
        g.hdr->overflow = 0;
    }
}

// Longest-list-first schedule for the blend kernels.  Workgroups are dispatched in index order, so
// giving index b the tile with the b-th longest list lets the short tiles fill in behind the long
// ones instead of a long tile starting last and running alone (the lists of an object-centric frame
// differ by 10x and more).  One workgroup: bucket sort of the tile ids on 1024 length classes.
struct TileOrderShared {
    uint32_t bin[1024];
    uint32_t scan[16];
    uint32_t max;
    uint32_t any;
    uint32_t n_long;
    unsigned long long total_len;
    uint32_t n_paired;
    // XCD-local schedule (grouped_order): one histogram / queue per XCD group
    uint32_t bin8[8][1024];
    uint32_t scan8[8][16];
    uint32_t n8[8], na[8], nb[8];
};

// XCD group of a tile: the BxB-tile block of its frame it lies in, blocks dealt to the eight XCDs so that every block row
// and every block column of a frame holds all of them (3 is odd: bx + 3 by runs through all residues along either axis),
// i.e. every group gets its share of border tiles (short lists) and interior tiles (long ones).
__device__ __forceinline__ uint32_t xcd_group(int t, const ScheduleParams& sp)
{
    const int frame = t / sp.frame_tiles, r = t - frame * sp.frame_tiles;
    const int bx = (r % sp.grid_x) / sp.xcd_block, by = (r / sp.grid_x) / sp.xcd_block;
    return (uint32_t)(bx + 3 * by + 5 * frame) & 7u;
}

// XCD-local longest-first order (round 6).  Workgroup b of a launch runs on XCD b mod 8 (observed,
// MI355X_MICROARCH.md "Workgroup dispatch"), and each XCD has its own L2: with ONE longest-first queue neighbouring tiles
// land on different XCDs and a surfel's record -- gathered by every tile whose list holds it, ~3 of them -- is fetched by
// ~3 L2s (blend_bwd: L2 hit rate 21 %, 2.9 x its algorithmic bytes behind the L2, profiles/r05_pmc_tcc.csv).  Here every
// tile has a group g (xcd_group: its block's XCD) and the tiles of a group form their own longest-first queue (a bucket
// sort on `cls_of`, 1024 classes, as before); out[] interleaves the eight queues: while all of them hold a tile of rank r,
// position 8 r + g takes the rank-r tile of group g, so position mod 8 = group = XCD; shorter queues drop out of the rotation.
// Two regions: the tiles whose class lies below cls_min (the split tiles: they must be exactly the first positions,
// blend.hip find_work_recorded / find_work_split_ordered) are interleaved first, the others behind them with the rotation
// continued (so that position mod 8 = group holds there as well).  cls_min = 1024: one region.
template <class ClassFn>
__device__ __forceinline__ void grouped_order(uint32_t* __restrict__ out, int num_tiles, TileOrderShared& sh,
                                              const ScheduleParams& sp, ClassFn cls_of, uint32_t cls_min)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < 8; g++) sh.bin8[g][threadIdx.x] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < num_tiles; t += 1024) atomicAdd(&sh.bin8[xcd_group(t, sp)][cls_of(t)], 1u);
    __syncthreads();
    uint32_t c[8], inc[8];
#pragma unroll
    for (int g = 0; g < 8; g++) {
        c[g] = sh.bin8[g][threadIdx.x];
        inc[g] = wave_inclusive_scan(c[g], lane);
        if (lane == 63) sh.scan8[g][wave] = inc[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 8; g++) {
        uint32_t wbase = 0;
#pragma unroll
        for (int w = 0; w < 16; w++)
            if (w < wave) wbase += sh.scan8[g][w];
        sh.bin8[g][threadIdx.x] = wbase + inc[g] - c[g];  // rank of the class's first tile in its group's queue
        if (threadIdx.x == 1023) sh.n8[g] = wbase + inc[g];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const uint32_t n = sh.n8[threadIdx.x], a = cls_min < 1024u ? sh.bin8[threadIdx.x][cls_min] : n;
        sh.na[threadIdx.x] = a;
        sh.nb[threadIdx.x] = n - a;
    }
    __syncthreads();
    uint32_t na[8], nb[8], S = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        na[g] = sh.na[g];
        nb[g] = sh.nb[g];
        S += na[g];
    }
    __syncthreads();  // (the region counts are read: the cursors below overwrite bin8[.][cls_min])
    for (int t = threadIdx.x; t < num_tiles; t += 1024) {
        const uint32_t g = xcd_group(t, sp), cls = cls_of(t);
        uint32_t r = atomicAdd(&sh.bin8[g][cls], 1u), pos;
        if (cls < cls_min) {
            pos = 0;
#pragma unroll
            for (uint32_t h = 0; h < 8; h++) pos += min(na[h], r) + ((h < g && na[h] > r) ? 1u : 0u);
        } else {
            r -= na[g];
            const uint32_t ig = (g - S) & 7u;   // (the rotation continues behind the first region: position mod 8 = group)
            pos = S;
#pragma unroll
            for (uint32_t h = 0; h < 8; h++) pos += min(nb[h], r) + ((((h - S) & 7u) < ig && nb[h] > r) ? 1u : 0u);
        }
        out[pos] = (uint32_t)t;
    }
}

// by_class (recorded segments): a tile is split iff its length CLASS lies above the class of split_min -- the split tiles
// are then exactly the schedule positions [0, num_split_pos), which is what lets blend_bwd number the full segments and the
// remainders separately (blend.hip find_work_recorded); every split tile is longer than split_min.
__device__ __forceinline__ void tile_order(GeomState g, ImageState img, int num_tiles, TileOrderShared& sh, const ScheduleParams sp)
{
    const int seg_len = sp.seg_len, split_min = sp.split_min, by_class = sp.by_class;
    // (the XCD-local order keeps the split tiles in front only when they are split by length class)
    const bool xcd = sp.xcd_block > 0 && by_class && sp.grid_x > 0 && sp.frame_tiles > 0;
    uint32_t* s_bin = sh.bin;
    uint32_t* s_scan = sh.scan;
    uint32_t& s_max = sh.max;
    uint32_t& s_any = sh.any;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_max = 0;
    s_bin[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mx = 0, n_long = 0;
    unsigned long long total = 0;
    for (int t = threadIdx.x; t < num_tiles; t += 1024) {
        const uint32_t len = img.ranges[2 * t + 1] - img.ranges[2 * t];
        mx = max(mx, len);
        total += len;
        n_long += len > (uint32_t)SPLIT_MIN;
        img.seg_first[t] = SEG_NONE;
        img.live_count[t] = 0;   // (by schedule position; the paired workgroups of a tile raise it with atomicMax, blend.hip)
    }
    for (int off = 32; off; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    if (lane == 0) atomicMax(&s_max, mx);
    // (Header::num_long_tiles: tiles longer than SPLIT_MIN, whatever table this call builds -- the caller's split decision
    // reads THIS, not num_split_pos, whose meaning follows the mode of the call)
    for (int off = 32; off; off >>= 1) n_long += (uint32_t)__shfl_xor((int)n_long, off);
    if (threadIdx.x == 0) {
        sh.n_long = 0;
        sh.total_len = 0;
        sh.n_paired = 0;
    }
    __syncthreads();
    if (lane == 0 && n_long) atomicAdd(&sh.n_long, n_long);
    if (sp.pair_k) {   // (the launch's pair count: one shared-memory atomic per wave)
        for (int off = 32; off; off >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)total, off), hi = (uint32_t)__shfl_xor((int)(uint32_t)(total >> 32), off);
            total += ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0 && total) atomicAdd(&sh.total_len, total);
    }
    __syncthreads();
    const uint32_t max_len = s_max;
    const uint64_t denom = (uint64_t)max_len + 1;
    auto bucket = [&](int t) {
        const uint32_t len = img.ranges[2 * t + 1] - img.ranges[2 * t];
        return 1023u - (uint32_t)(((uint64_t)len << 10) / denom);
    };
    const uint32_t cls_min = 1023u - (uint32_t)(((uint64_t)min((uint32_t)split_min, max_len) << 10) / denom);
    if (xcd) {
        grouped_order(img.tile_order, num_tiles, sh, sp, bucket, cls_min);
    } else {
        for (int t = threadIdx.x; t < num_tiles; t += 1024) atomicAdd(&s_bin[bucket(t)], 1u);
        __syncthreads();
        const uint32_t c = s_bin[threadIdx.x];
        const uint32_t inc = wave_inclusive_scan(c, lane);
        if (lane == 63) s_scan[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int w = 0; w < 16; w++)
            if (w < wave) wbase += s_scan[w];
        s_bin[threadIdx.x] = wbase + inc - c;  // exclusive start of this length class
        __syncthreads();
        for (int t = threadIdx.x; t < num_tiles; t += 1024) img.tile_order[atomicAdd(&s_bin[bucket(t)], 1u)] = (uint32_t)t;
        if (sp.pair_k) {
            // Paired workgroups (blend.hip fwd_pair_walk): the tiles whose length class lies above the class of pair_k / 4 x
            // the mean list length -- a prefix of the schedule; s_bin[c] is now the END of class c, i.e. the number of tiles
            // of the classes 0 .. c (the longer ones)
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned long long thr = sh.total_len * (unsigned long long)sp.pair_k / (4ull * (unsigned long long)max(num_tiles, 1));
                uint32_t n = 0;
                if (sp.pair_k >= 15) n = (uint32_t)num_tiles;
                else if (thr < (unsigned long long)max_len) {
                    const uint32_t cls = 1023u - (uint32_t)((thr << 10) / denom);   // class of the threshold length: pair the classes in front of it
                    n = cls > 0u ? s_bin[cls - 1u] : 0u;
                }
                sh.n_paired = min(n, (uint32_t)PAIR_MAX);
            }
        }
    }
    __syncthreads();  // tile_order is read back below (same workgroup: the barrier orders the global accesses)

    // Segment table for the tiles longer than split_min (blend.hip; SPLIT_MIN for a segment-parallel forward, REC_MIN for
    // the recorded segments of a whole-tile forward): exclusive prefix of their segment
    // counts by schedule position.  The long tiles sit at the front of the schedule, so the walk stops
    // at the first chunk of 1024 positions that holds none (a length class may straddle SPLIT_MIN,
    // hence "none in a whole chunk" and not "the first short tile").
    uint32_t carry = 0, split_pos = 0;
    for (int base = 0; base < num_tiles; base += 1024) {
        const int pos = base + threadIdx.x;
        uint32_t nseg = 0, tile = 0;
        if (pos < num_tiles) {
            tile = img.tile_order[pos];
            const uint32_t len = img.ranges[2 * tile + 1] - img.ranges[2 * tile];
            const bool split = by_class ? (1023u - (uint32_t)(((uint64_t)len << 10) / denom)) < cls_min : len > (uint32_t)split_min;
            if (split) nseg = (len + (uint32_t)seg_len - 1) / (uint32_t)seg_len;
        }
        if (threadIdx.x == 0) s_any = 0;
        __syncthreads();
        const uint32_t sc = wave_inclusive_scan(nseg, lane);
        if (lane == 63) s_scan[wave] = sc;
        if (nseg) atomicMax(&s_any, (uint32_t)pos + 1);
        __syncthreads();
        uint32_t wb = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const uint32_t v = s_scan[w];
            if (w < wave) wb += v;
            tot += v;
        }
        const uint32_t any = s_any;
        if (pos < num_tiles) {
            const uint32_t first = carry + wb + sc - nseg;
            img.seg_prefix[pos] = first;
            if (nseg) img.seg_first[tile] = first;
        }
        carry += tot;
        if (any) split_pos = any;
        __syncthreads();
        if (!any) break;
    }
    if (by_class) {
        // Tails, largest first (blend.hip find_work_recorded): the remainder segment of a split tile or a whole unsplit
        // tile.  They are dispatched behind the full segments, so the launch drains on ever smaller units.  Same bucket
        // sort as above, on the tail's size.
        __syncthreads();  // (seg_first of every tile is final)
        s_bin[threadIdx.x] = 0;
        __syncthreads();
        const uint64_t denom2 = (uint64_t)max(max_len, (uint32_t)seg_len) + 1;
        auto tail_bucket = [&](int t) {
            const uint32_t len = img.ranges[2 * t + 1] - img.ranges[2 * t];
            const uint32_t tail = img.seg_first[t] == SEG_NONE ? len : len - ((len - 1u) / (uint32_t)seg_len) * (uint32_t)seg_len;
            return 1023u - (uint32_t)(((uint64_t)tail << 10) / denom2);
        };
        if (xcd) {
            // (the tails follow the full segments, whose count the recorded backward pads to a multiple of 8 -- bwd_prepare_kernel --
            // so tail j runs on XCD j mod 8: the same eight queues, on the tail's size)
            grouped_order(img.tail_order, num_tiles, sh, sp, tail_bucket, 1024u);
        } else {
            for (int t = threadIdx.x; t < num_tiles; t += 1024) atomicAdd(&s_bin[tail_bucket(t)], 1u);
            __syncthreads();
            const uint32_t c2 = s_bin[threadIdx.x];
            const uint32_t inc2 = wave_inclusive_scan(c2, lane);
            if (lane == 63) s_scan[wave] = inc2;
            __syncthreads();
            uint32_t wbase2 = 0;
#pragma unroll
            for (int w = 0; w < 16; w++)
                if (w < wave) wbase2 += s_scan[w];
            s_bin[threadIdx.x] = wbase2 + inc2 - c2;
            __syncthreads();
            for (int t = threadIdx.x; t < num_tiles; t += 1024) img.tail_order[atomicAdd(&s_bin[tail_bucket(t)], 1u)] = (uint32_t)t;
        }
    }
    if (threadIdx.x == 0) {
        img.seg_prefix[split_pos] = carry;  // closes the last split tile's interval
        g.hdr->max_tile_len = max_len;
        g.hdr->num_segments = carry;
        g.hdr->num_split_pos = split_pos;
        g.hdr->seg_len = (uint32_t)seg_len;
        g.hdr->split_min = (uint32_t)split_min;
        g.hdr->num_long_tiles = sh.n_long;
        g.hdr->split_used = 0;
        g.hdr->truncated = 0;
        g.hdr->xcd_block = xcd ? (uint32_t)sp.xcd_block : 0u;
        g.hdr->live_xcd = 0;
        g.hdr->num_paired = sh.n_paired;
    }
}

__global__ __launch_bounds__(1024) void tile_order_kernel(GeomState g, ImageState img, int num_tiles, ScheduleParams sp)
{
    __shared__ TileOrderShared sh;
    tile_order(g, img, num_tiles, sh, sp);
}

// Grouped path, ONE launch (round 1 used three dependent ones, 22 us of mostly launch latency for ~1 MB):
//   1. every workgroup turns 64 columns group_counts[*][t] into exclusive prefixes over the groups.  Wave w of
//      the 16 owns the groups [w G, (w + 1) G), lane = tile: a row segment is one coalesced 256-byte read, the
//      <= 16 counts of a thread stay in registers, the 16 partial sums per tile meet in LDS;
//   2. the workgroup that arrives LAST at the header's counter (agent-scope release / acquire around one relaxed
//      atomic, as the MI355X guide prescribes for inter-workgroup hand-offs) scans the tile totals into the
//      ranges and builds the longest-first schedule and segment table -- 12 KB of work for one workgroup.
// The counter is reset by the projection kernel of the same forward (the header is fresh memory every call).
constexpr int SCAN_TILES = 64;   // tiles per workgroup
constexpr int SCAN_MAX_PER_WAVE = BIN_MAX_GROUPS / 16;

__global__ __launch_bounds__(1024) void tile_scan_fused_kernel(GeomState g, ImageState img, int num_tiles, int groups)
{
    __shared__ uint32_t s_part[16][SCAN_TILES];
    __shared__ uint32_t s_scan16[16];
    __shared__ uint32_t s_last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * SCAN_TILES + lane;
    const int per = (groups + 15) / 16;
    const int g0 = wave * per, g1 = min(groups, g0 + per);
    uint32_t c[SCAN_MAX_PER_WAVE];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_MAX_PER_WAVE; i++) {
        c[i] = (t < num_tiles && g0 + i < g1) ? img.group_counts[(size_t)(g0 + i) * num_tiles + t] : 0u;
        sum += c[i];
    }
    s_part[wave][lane] = sum;
    __syncthreads();
    uint32_t run = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const uint32_t v = s_part[w][lane];
        if (w < wave) run += v;
        total += v;
    }
    if (t < num_tiles) {
#pragma unroll
        for (int i = 0; i < SCAN_MAX_PER_WAVE; i++) {
            if (g0 + i < g1) img.group_counts[(size_t)(g0 + i) * num_tiles + t] = run;
            run += c[i];
        }
        if (wave == 0) img.tile_count[t] = total;
    }
    // ---- hand-off to the last workgroup
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t arrived = __hip_atomic_fetch_add(&g.hdr->scan_arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = arrived == gridDim.x - 1;
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;
    tile_totals_scan(g, img, num_tiles, s_scan16);
    // (round 4: the longest-first schedule and the segment tables -- 20 us of single-workgroup work that only the sort and
    // the blend read -- are built by an extra workgroup of the key emit that follows, beside it: launch_emit_keys)
    if (threadIdx.x == 0) g.hdr->scan_arrivals = 0;
}

void launch_tile_scan(const GeomState& g, const ImageState& img, int num_tiles, int groups, hipStream_t stream)
{
    if (groups > 0) {
        hipLaunchKernelGGL(tile_scan_fused_kernel, dim3((num_tiles + SCAN_TILES - 1) / SCAN_TILES), dim3(1024), 0, stream,
                           g, img, num_tiles, groups);
        return;
    }
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, stream, g, img, num_tiles);
}

void launch_tile_order(const GeomState& g, const ImageState& img, int num_tiles, const ScheduleParams& sp, hipStream_t stream)
{
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, g, img, num_tiles, sp);
}

__global__ __launch_bounds__(PRE_BLOCK) void emit_keys_kernel(CameraParams cam, int P, const int32_t* radii,
                                                             GeomState g, ImageState img, uint64_t* entries,
                                                             int64_t capacity)
{
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    if (idx == 0) g.hdr->num_buckets = 0;  // (work list of the long-list sort that follows this launch)
    if ((int64_t)g.hdr->num_rendered > capacity) {  // binning buffer too small: render nothing, flag it
        if (idx == 0) g.hdr->overflow = 1;
        return;
    }
    if (idx >= P) return;
    const int radius = radii[idx];
    if (radius <= 0) return;
    const float4 q2 = reinterpret_cast<const float4*>(g.rec + (size_t)idx * REC_FLOATS)[2];
    const float4 q3 = reinterpret_cast<const float4*>(g.rec + (size_t)idx * REC_FLOATS)[3];
    int x0, y0, x1, y1;
    tile_rect(q2.y, q2.z, radius, cam.grid_x, cam.grid_y, x0, y0, x1, y1);
    const uint64_t entry = ((uint64_t)__float_as_uint(q3.w) << 32) | (uint32_t)idx;
    const int slice = blockIdx.x & (TILE_SLICES - 1);  // same slice the count came from (same launch geometry)
    const int tile0 = cam.frames > 1 ? (idx / cam.frame_surfels) * cam.grid_x * cam.grid_y : 0;  // (stacked frames)
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const size_t b = (size_t)(tile0 + y * cam.grid_x + x) * TILE_SLICES + slice;
            const uint32_t pos = img.tile_base[b] + atomicAdd(&img.tile_count[b], 1u);
            entries[pos] = entry;
        }
}

// Grouped path: the same 1024-thread groups as the projection; each group owns, in every tile's
// segment, the sub-range [ranges[t].x + prefix[group][t], +count[group][t]) and hands out its slots
// with LDS atomics.
// The LAST workgroup of the launch emits nothing: it builds the longest-first schedule and the segment tables
// (tile_order), which only the sort and the blend kernels behind this launch read -- beside the emit instead of in front of it.
template <bool COMPACT>
__global__ __launch_bounds__(BIN_THREADS) void emit_keys_grouped_kernel(CameraParams cam, int P, int iters,
                                                                       const int32_t* radii, GeomState g,
                                                                       ImageState img, uint64_t* entries,
                                                                       int64_t capacity, ScheduleParams sp)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_cur[];
    if (blockIdx.x == gridDim.x - 1) {
        tile_order(g, img, cam.grid_x * cam.grid_y * cam.frames, *reinterpret_cast<TileOrderShared*>(s_cur), sp);
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) g.hdr->num_buckets = 0;  // (work list of the long-list sort that follows)
    if ((int64_t)g.hdr->num_rendered > capacity) {
        if (blockIdx.x == 0 && threadIdx.x == 0) g.hdr->overflow = 1;
        return;
    }
    const int frame_tiles = cam.grid_x * cam.grid_y, num_tiles = frame_tiles * cam.frames;
    const uint32_t* row = img.group_counts + (size_t)blockIdx.x * num_tiles;
    for (int t = threadIdx.x; t < num_tiles; t += BIN_THREADS) s_cur[t] = img.ranges[2 * t] + row[t];
    __syncthreads();
    const int first = blockIdx.x * BIN_THREADS * iters;
    for (int it = 0; it < iters; it++) {
        const int idx = first + it * BIN_THREADS + threadIdx.x;
        if (idx >= P) continue;
        // COMPACT: centre, depth and radius as the projection kernel left them in one 16-byte slot per surfel
        // (preprocess.hip, staged path); else from the radius array and two strided pieces of the record
        int radius;
        float cx, cy, depth;
        if (COMPACT) {
            const float4 c = reinterpret_cast<const float4*>(g.colour)[idx];
            cx = c.x, cy = c.y, depth = c.z, radius = __float_as_int(c.w);
            if (radius <= 0) continue;
        } else {
            radius = radii[idx];
            if (radius <= 0) continue;
            const float4 q2 = reinterpret_cast<const float4*>(g.rec + (size_t)idx * REC_FLOATS)[2];
            const float4 q3 = reinterpret_cast<const float4*>(g.rec + (size_t)idx * REC_FLOATS)[3];
            cx = q2.y, cy = q2.z, depth = q3.w;
        }
        int x0, y0, x1, y1;
        tile_rect(cx, cy, radius, cam.grid_x, cam.grid_y, x0, y0, x1, y1);
        const uint64_t entry = ((uint64_t)__float_as_uint(depth) << 32) | (uint32_t)idx;
        uint32_t* cur = s_cur + (cam.frames > 1 ? (idx / cam.frame_surfels) * frame_tiles : 0);
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) entries[atomicAdd(&cur[y * cam.grid_x + x], 1u)] = entry;
    }
}

void launch_emit_keys(const CameraParams& cam, int P, const int32_t* radii, const GeomState& g, const ImageState& img,
                      const BinState& b, int64_t capacity, bool grouped, const ScheduleParams& sp, hipStream_t stream)
{
    if (P <= 0 || capacity <= 0) {   // nothing to emit: the schedule on its own (the blend kernels read it whatever happens)
        launch_tile_order(g, img, total_tiles(cam), sp, stream);
        return;
    }
    if (grouped) {
        auto kernel = preprocess_stages_records(total_tiles(cam)) ? &emit_keys_grouped_kernel<true> : &emit_keys_grouped_kernel<false>;
        const size_t lds = std::max((size_t)total_tiles(cam) * sizeof(uint32_t), sizeof(TileOrderShared));
        hipLaunchKernelGGL(kernel, dim3(bin_groups(P) + 1), dim3(BIN_THREADS), lds, stream,
                           cam, P, bin_iters(P), radii, g, img, b.entries, capacity, sp);
    } else {
        launch_tile_order(g, img, total_tiles(cam), sp, stream);
        hipLaunchKernelGGL(emit_keys_kernel, dim3(pre_blocks(P)), dim3(PRE_BLOCK), 0, stream, cam, P, radii, g, img,
                           b.entries, capacity);
    }
}

// One workgroup per tile.  Stable LSD radix sort of the tile's segment on the bytes of
// (depth bits << 32 | id) listed by the caller's id_bytes (id bytes 0..id_bytes-1, then depth bytes).
// <4, true>: 4 wave64, lists up to TILE_SORT_CAP are sorted inside LDS, longer ones ping-pong through
// global memory (unless `skip_long`: then the <16, false> launch takes them -- 16 wave64 per tile,
// global ping-pong only; the long tiles are the first positions of the longest-first schedule).
// CAP: list length the LDS ping-pong holds.  Two in-LDS launches share the tiles by length: <4, true, 1024>
// (16 KiB + counters: 8 workgroups per CU, all 1024 tiles of a 512^2 frame resident at once) takes the lists up
// to 1024 entries, <4, true, TILE_SORT_CAP> (56 KiB: 2 per CU) the longer ones; a workgroup whose tile belongs
// to the other launch exits at once.  (One launch with the large buffers ran the typical 600-entry lists of
// the headline scene in two rounds of 512 workgroups: 42 us instead of 25.)
// One list: src[0 .. n) -> sorted into entries[start ..) / point_list[start ..).  `depth_bits`: only the low depth_bits
// bits of the depth words differ inside the list (32: unknown) -- the byte passes above them are not even counted.
// long_src (global-memory path only): the list may start in `scratch` (after an MSD split) instead of `entries`.
template <int WAVES, bool IN_LDS, int CAP>
__device__ __forceinline__ void sort_one_list(const uint64_t* __restrict__ src, uint32_t start, int n, int depth_bits,
                                              uint64_t* entries, uint64_t* scratch, uint32_t* __restrict__ point_list,
                                              int id_bytes, uint64_t (*s_buf)[IN_LDS ? CAP : 1], uint32_t (*s_cnt)[256],
                                              uint32_t* s_scan, uint32_t* s_long_run)
{
    constexpr int THREADS = WAVES * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_lds = IN_LDS && n <= CAP;
    uint64_t* A;
    uint64_t* B;
    if (in_lds) {
        for (int i = threadIdx.x; i < n; i += THREADS) s_buf[0][i] = src[i];
        A = s_buf[0];
        B = s_buf[IN_LDS ? 1 : 0];
    } else {
        A = entries + start;
        B = scratch + start;
    }
    // wave w owns the consecutive keys [w*chunk, min(n, (w+1)*chunk)), walked 64 at a time
    const int chunk = ((n + THREADS - 1) / THREADS) << 6;
    const int w_lo = min(wave * chunk, n);
    const int w_hi = (w_lo + chunk < n) ? w_lo + chunk : n;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    // One stable counting pass on the byte at `shift` (skipped when the whole list shares that byte).
    auto radix_pass = [&](int shift) {
        for (int i = threadIdx.x; i < WAVES * 256; i += THREADS) (&s_cnt[0][0])[i] = 0;
        __syncthreads();  // also orders the previous pass's scatter (or the initial load) before the reads
        for (int i = w_lo + lane; i < w_hi; i += 64) atomicAdd(&s_cnt[wave][(uint32_t)(A[i] >> shift) & 255u], 1u);
        __syncthreads();
        const int d = threadIdx.x & 255;  // one digit per thread (threads 0..255)
        uint32_t tot = 0;
        if (threadIdx.x < 256)
            for (int w = 0; w < WAVES; w++) tot += s_cnt[w][d];
        if (__syncthreads_or(tot == (uint32_t)n)) return;  // whole segment shares this digit: nothing to move
        uint32_t total;
        const uint32_t dbase = block_exclusive_scan(tot, s_scan, total);  // (waves >= 4 contribute zeros)
        if (threadIdx.x < 256) {
            uint32_t run = dbase;
            for (int w = 0; w < WAVES; w++) {
                const uint32_t c = s_cnt[w][d];
                s_cnt[w][d] = run;
                run += c;
            }
        }
        __syncthreads();
        for (int base = w_lo; base < w_hi; base += 64) {
            const int i = base + lane;
            const bool valid = i < w_hi;
            const uint64_t key = valid ? A[i] : ~0ull;
            const uint32_t dg = (uint32_t)(key >> shift) & 255u;
            unsigned long long peers = __ballot(valid);
            if (!valid) peers = ~peers;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long m = __ballot((dg >> b) & 1u);
                peers &= ((dg >> b) & 1u) ? m : ~m;
            }
            const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
            if (valid) {
                const uint32_t old = s_cnt[wave][dg];
                if (before == 0) s_cnt[wave][dg] = old + (uint32_t)__popcll(peers);  // leader advances the cursor
                B[old + before] = key;
            }
        }
        uint64_t* t = A;
        A = B;
        B = t;
        __syncthreads();  // every wave is done with its cursors before the next pass clears them
    };

    // The order to reproduce is the reference's: stable sort by depth of entries emitted in id order, i.e. (depth, id)
    // ascending.  Sorting on all of (id bytes, depth bytes) costs 6-7 passes, and the id bytes are only there for exact
    // fp32 depth ties between surfels of one tile: rare in short lists, a few dozen pairs per list in 25 k-entry ones
    // (birthday statistics over ~2^21 depth values), whole clouds when densification has just cloned surfels.  So: the
    // depth bytes only (3 effective passes: the exponent byte is constant and skipped), then every run of equal depths
    // is put in id order in place by the thread that finds its start -- runs are disjoint, and permuting ids inside a
    // run changes no depth word, so the other threads' run-start tests are unaffected.  Only a run longer than
    // TIE_RUN_MAX (degenerate scenes: a plane of surfels at one depth) sends the list, as it stands, through the full
    // LSD sequence, which does not care about the order it starts from.
    constexpr int TIE_RUN_MAX = 32;
    for (int pass = 0; pass < 4; pass++)
        if (8 * pass < depth_bits) radix_pass(32 + 8 * pass);
    if (threadIdx.x == 0) *s_long_run = 0;
    __syncthreads();
    for (int i = threadIdx.x; i + 1 < n; i += THREADS) {
        const uint32_t d = (uint32_t)(A[i] >> 32);
        if ((uint32_t)(A[i + 1] >> 32) != d || (i > 0 && (uint32_t)(A[i - 1] >> 32) == d)) continue;  // not a run start
        int j = i + 1;
        while (j + 1 < n && j - i < TIE_RUN_MAX && (uint32_t)(A[j + 1] >> 32) == d) j++;
        if (j - i >= TIE_RUN_MAX) {
            *s_long_run = 1;
            continue;
        }
        for (int a = i + 1; a <= j; a++) {  // insertion sort of A[i..j] (equal depths: the keys order by id)
            const uint64_t key = A[a];
            int b = a - 1;
            while (b >= i && A[b] > key) {
                A[b + 1] = A[b];
                b--;
            }
            A[b + 1] = key;
        }
    }
    __syncthreads();
    if (*s_long_run)
        for (int pass = 0; pass < id_bytes + 4; pass++) radix_pass(8 * (pass < id_bytes ? pass : 4 + pass - id_bytes));
    for (int i = threadIdx.x; i < n; i += THREADS) {
        const uint64_t key = A[i];
        point_list[start + i] = (uint32_t)key;
        if (A != entries + start) entries[start + i] = key;
    }
    __syncthreads();  // (a persistent caller reuses the shared arrays for its next list)
}

template <int WAVES, bool IN_LDS, int CAP>
__global__ __launch_bounds__(WAVES * 64) void tile_sort_kernel(const uint32_t* __restrict__ tile_order,
                                                              const uint32_t* __restrict__ ranges,
                                                              const uint32_t* num_ptr, int64_t capacity,
                                                              uint64_t* entries, uint64_t* scratch,
                                                              uint32_t* __restrict__ point_list, int id_bytes,
                                                              int skip_long, int min_len, int take_long)
{
    __shared__ uint64_t s_buf[IN_LDS ? 2 : 1][IN_LDS ? CAP : 1];
    __shared__ uint32_t s_cnt[WAVES][256];  // per-wave digit counts, then per-wave destination cursors
    __shared__ uint32_t s_scan[16];
    __shared__ uint32_t s_long_run;
    if ((int64_t)*num_ptr > capacity) return;
    const uint32_t tile = tile_order[blockIdx.x];  // longest list first
    const uint32_t start = ranges[2 * tile];
    const int n = (int)(ranges[2 * tile + 1] - start);
    if (n == 0 || n < min_len) return;
#ifdef SURFEL_ABLATE_SORT_GLOBAL   // (timing experiment, results wrong: what do the lists beyond the LDS capacity cost?)
    if (n > CAP) return;
#endif
    // (take_long: this is the only sort launch -- a list beyond CAP goes through global memory here)
    if (IN_LDS ? ((skip_long || (CAP < TILE_SORT_CAP && !take_long)) && n > CAP) : n <= TILE_SORT_CAP) return;
    sort_one_list<WAVES, IN_LDS, CAP>(entries + start, start, n, 32, entries, scratch, point_list, id_bytes, s_buf, s_cnt,
                                      s_scan, &s_long_run);
}

// ---- Lists longer than TILE_SORT_CAP: MSD split, then in-LDS bucket sorts (round 3).
// A 16-wave workgroup per long tile finds the highest depth bit that differs inside the list, takes the 8 bits from there
// down as the digit, and moves the list -- one stable counting pass, entries -> scratch -- into up to 256 buckets in
// digit (= depth) order; every non-empty bucket is appended to a work list.  bucket_sort_kernel then sorts the buckets
// on their remaining low bits entirely inside LDS, one 4-wave workgroup per bucket at a time, and writes them back to
// `entries` / `point_list` in place: a 25 k-entry list crosses global memory twice instead of six to eight times, and
// its ~40-100 buckets are sorted by as many workgroups instead of one.  A list with a bucket beyond the LDS capacity
// (thousands of surfels within one 2^-8 slice of the list's depth range) takes the global-memory sort as before.
struct SortBucket {
    uint32_t start, count, low_bits, pad;
};
constexpr int MSD_MIN = 1024;
// lists longer than this are split (when the long pass runs at all)

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void tile_split_kernel(const uint32_t* __restrict__ tile_order,
                                                               const uint32_t* __restrict__ ranges, Header* hdr,
                                                               int64_t capacity, uint64_t* entries, uint64_t* scratch,
                                                               uint32_t* __restrict__ point_list, int id_bytes,
                                                               SortBucket* __restrict__ buckets, uint32_t max_buckets)
{
    constexpr int THREADS = WAVES * 64;
    __shared__ uint64_t s_buf[1][1];
    __shared__ uint32_t s_cnt[WAVES][256];
    __shared__ uint32_t s_scan[16];
    __shared__ uint32_t s_long_run, s_vary, s_slot;
    if ((int64_t)hdr->num_rendered > capacity) return;
    const uint32_t tile = tile_order[blockIdx.x];
    const uint32_t start = ranges[2 * tile];
    const int n = (int)(ranges[2 * tile + 1] - start);
    if (n <= MSD_MIN) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t* A = entries + start;
    // 1. which depth bits differ inside the list, and -- in the same pass -- the histogram of depth byte 2 (bits 16-23:
    //    the top mantissa bits when the whole list lies inside one binade, the usual case for an object in front of the
    //    camera).  Only a list whose depths differ above bit 23 (or only below bit 20) pays a second counting pass, on the
    //    8 bits below its highest differing bit.
    const int chunk = ((n + THREADS - 1) / THREADS) << 6;
    const int w_lo = min(wave * chunk, n);
    const int w_hi = (w_lo + chunk < n) ? w_lo + chunk : n;
    if (threadIdx.x == 0) s_vary = 0;
    for (int i = threadIdx.x; i < WAVES * 256; i += THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t first = (uint32_t)(A[0] >> 32);
    uint32_t vary = 0;
    for (int i = w_lo + lane; i < w_hi; i += 64) {
        const uint32_t dep = (uint32_t)(A[i] >> 32);
        vary |= dep ^ first;
        atomicAdd(&s_cnt[wave][(dep >> 16) & 255u], 1u);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) vary |= (uint32_t)__shfl_xor((int)vary, d, 64);
    if (lane == 0 && vary) atomicOr(&s_vary, vary);
    __syncthreads();
    vary = s_vary;
    const int msb = vary ? 31 - __builtin_clz(vary) : 0;
    // digit = (depth >> low) & 255 with nothing differing above bit low + 7: byte 2 as counted when the highest differing bit
    // is one of its upper four (>= 16 buckets' worth of spread), else the 8 bits below the highest differing bit
    const int low = (msb >= 20 && msb <= 23) ? 16 : (msb > 7 ? msb - 7 : 0);
    const int shift = 32 + low;
    if (low != 16) {  // (wave-uniform, workgroup-uniform)
        __syncthreads();
        for (int i = threadIdx.x; i < WAVES * 256; i += THREADS) (&s_cnt[0][0])[i] = 0;
        __syncthreads();
        for (int i = w_lo + lane; i < w_hi; i += 64) atomicAdd(&s_cnt[wave][(uint32_t)(A[i] >> shift) & 255u], 1u);
        __syncthreads();
    }
    const int d = threadIdx.x & 255;
    uint32_t tot = 0;
    if (threadIdx.x < 256)
        for (int w = 0; w < WAVES; w++) tot += s_cnt[w][d];
    const bool too_big = __syncthreads_or(tot > (uint32_t)TILE_SORT_CAP) || vary == 0;
    uint32_t nonempty_total;
    const uint32_t my_slot = block_exclusive_scan(threadIdx.x < 256 && tot ? 1u : 0u, s_scan, nonempty_total);
    __syncthreads();
    if (threadIdx.x == 0) s_slot = too_big ? 0xffffffffu : atomicAdd(&hdr->num_buckets, nonempty_total);
    __syncthreads();
    if (too_big || s_slot + nonempty_total > max_buckets) {
        // (the work list is sized for every long list; the test only guards a caller who passed a smaller buffer)
        sort_one_list<WAVES, false, 1>(A, start, n, 32, entries, scratch, point_list, id_bytes, s_buf, s_cnt, s_scan, &s_long_run);
        return;
    }
    uint32_t total;
    const uint32_t dbase = block_exclusive_scan(tot, s_scan, total);
    if (threadIdx.x < 256) {
        uint32_t run = dbase;
        for (int w = 0; w < WAVES; w++) {
            const uint32_t c = s_cnt[w][d];
            s_cnt[w][d] = run;
            run += c;
        }
        if (tot) buckets[s_slot + my_slot] = SortBucket{start + dbase, tot, (uint32_t)low, 0u};
    }
    __syncthreads();
    // 3. stable scatter into the buckets (entries -> scratch)
    uint64_t* B = scratch + start;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (int base = w_lo; base < w_hi; base += 64) {
        const int i = base + lane;
        const bool valid = i < w_hi;
        const uint64_t key = valid ? A[i] : ~0ull;
        const uint32_t dg = (uint32_t)(key >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
        if (!valid) peers = ~peers;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((dg >> b) & 1u);
            peers &= ((dg >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
        if (valid) {
            const uint32_t old = s_cnt[wave][dg];
            if (before == 0) s_cnt[wave][dg] = old + (uint32_t)__popcll(peers);
            B[old + before] = key;
        }
    }
}

// Buckets of at most SMALL_BUCKET entries: one WAVE per bucket, one key per lane, no barriers -- every lane ranks its key
// against all keys of the bucket, read back from a wave-private LDS copy with broadcast reads; the full 64-bit (depth, id)
// keys are distinct, so the rank IS the sorted position and no tie pass is needed.
constexpr int SMALL_BUCKET = 64;
__global__ __launch_bounds__(256) void bucket_sort_small_kernel(const Header* hdr, int64_t capacity,
                                                               const SortBucket* __restrict__ buckets, uint64_t* entries,
                                                               const uint64_t* __restrict__ scratch,
                                                               uint32_t* __restrict__ point_list)
{
    __shared__ uint64_t s_keys[4][SMALL_BUCKET];
    if ((int64_t)hdr->num_rendered > capacity) return;
    const uint32_t nb = hdr->num_buckets;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t* keys = s_keys[wave];
    for (uint32_t b = blockIdx.x * 4 + wave; b < nb; b += gridDim.x * 4) {
        const SortBucket k = buckets[b];
        const int n = (int)k.count;
        if (n > SMALL_BUCKET) continue;  // wave-uniform
        const uint64_t mine = lane < n ? scratch[k.start + lane] : ~0ull;
        if (lane < n) keys[lane] = mine;
        __builtin_amdgcn_wave_barrier();  // (LDS serves one wave's instructions in order: its reads see its writes)
        uint32_t rank = 0;
        for (int i = 0; i < n; i++) rank += keys[i] < mine ? 1u : 0u;
        if (lane < n) {
            entries[k.start + rank] = mine;
            point_list[k.start + rank] = (uint32_t)mine;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Two instances share the larger buckets by size, as the tile sorts do: CAP 1024 (16 KiB of LDS: 8 workgroups per CU) takes
// the buckets of (MIN, 1024] entries, CAP TILE_SORT_CAP the larger ones.
template <int CAP, int MIN>
__global__ __launch_bounds__(256) void bucket_sort_kernel(const Header* hdr, int64_t capacity, const SortBucket* __restrict__ buckets,
                                                         uint64_t* entries, uint64_t* scratch,
                                                         uint32_t* __restrict__ point_list, int id_bytes)
{
    __shared__ uint64_t s_buf[2][CAP];
    __shared__ uint32_t s_cnt[4][256];
    __shared__ uint32_t s_scan[16];
    __shared__ uint32_t s_long_run;
    if ((int64_t)hdr->num_rendered > capacity) return;
    const uint32_t nb = hdr->num_buckets;
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const SortBucket k = buckets[b];
        if ((int)k.count > CAP || (int)k.count <= MIN) continue;  // (wave-uniform: the other instance's bucket)
        sort_one_list<4, true, CAP>(scratch + k.start, k.start, (int)k.count, (int)k.low_bits, entries, scratch, point_list,
                                    id_bytes, s_buf, s_cnt, s_scan, &s_long_run);
    }
}

// Waves per workgroup of the in-LDS sort of the lists beyond 1024 entries (56 KiB of LDS: two workgroups per CU whatever
// their size, so the waves per tile are the waves per CU, and a list's passes -- count, scan, scatter between barriers --
// are a latency chain).  Dense Stage-3 ball (790 lists per frame beyond 320 entries, median 2 140, longest 5 240),
// tools/experiments/sort_waves_ab.sh: 4 waves 69-70 us, 8 waves 46-47 us, 16 waves 56-58 us per launch; a third instance
// for the lists of (1024, 2048] entries changed nothing (the launch lasts as long as its longest lists).  20 of the 46 us are
// the ~10 % of those lists beyond TILE_SORT_CAP, which ping-pong through global memory (the SURFEL_ABLATE_SORT_GLOBAL build);
// a 16-wave instance of their own on 128 KiB of launch-time LDS sorted them in 30 us -- one list's latency chain -- but as a
// launch in front of this one, which then took 28: 58 us for the two.  Not kept.
#ifndef SURFEL_SORT_LONG_WAVES
#define SURFEL_SORT_LONG_WAVES 8
#endif
constexpr int SORT_LONG_WAVES = SURFEL_SORT_LONG_WAVES;

void launch_tile_sort(const GeomState& g, const ImageState& img, const BinState& b, int num_tiles, int num_surfels,
                      int64_t capacity, LongListSort mode, hipStream_t stream)
{
    if (capacity <= 0 || num_tiles <= 0) return;
    int id_bytes = 1;
    while (id_bytes < 4 && ((uint64_t)(num_surfels > 0 ? num_surfels - 1 : 0) >> (8 * id_bytes))) id_bytes++;
    constexpr int SMALL_CAP = 1024;
    static_assert(MSD_MIN == SMALL_CAP, "the split takes every list the small in-LDS sort does not");
    // the long tiles lead the schedule, but it is sorted by length CLASS only (a long tile may sit behind shorter ones of
    // its class), so every position gets a workgroup; the ones a launch does not own exit at once
    if (mode == LongListSort::msd_split) {
        // (the bucket work list lives at the start of seg_data, which nobody touches before the blend)
        SortBucket* buckets = reinterpret_cast<SortBucket*>(b.seg_data);
        const uint32_t max_buckets = (uint32_t)std::min<int64_t>(256 * (capacity / MSD_MIN + 1), 0x7fffffff);
        hipLaunchKernelGGL((tile_split_kernel<16>), dim3(num_tiles), dim3(1024), 0, stream, img.tile_order, img.ranges, g.hdr,
                           capacity, b.entries, b.scratch, b.point_list, id_bytes, buckets, max_buckets);
        const int grid = (int)std::min<int64_t>(std::max<int64_t>(capacity / 512, 1), 2048);
        hipLaunchKernelGGL((bucket_sort_kernel<TILE_SORT_CAP, 1024>), dim3(grid), dim3(256), 0, stream, g.hdr, capacity, buckets,
                           b.entries, b.scratch, b.point_list, id_bytes);
        hipLaunchKernelGGL((bucket_sort_kernel<1024, SMALL_BUCKET>), dim3(grid), dim3(256), 0, stream, g.hdr, capacity, buckets,
                           b.entries, b.scratch, b.point_list, id_bytes);
        hipLaunchKernelGGL(bucket_sort_small_kernel, dim3(grid), dim3(256), 0, stream, g.hdr, capacity, buckets, b.entries,
                           b.scratch, b.point_list);
    } else {
        if (mode == LongListSort::one_workgroup)  // lists beyond the LDS capacity: 16 waves each, through global memory
            hipLaunchKernelGGL((tile_sort_kernel<16, false, 1>), dim3(num_tiles), dim3(1024), 0, stream, img.tile_order,
                               img.ranges, &g.hdr->num_rendered, capacity, b.entries, b.scratch, b.point_list, id_bytes, 0, 0, 0);
        if (mode != LongListSort::short_lists_expected)
            hipLaunchKernelGGL((tile_sort_kernel<SORT_LONG_WAVES, true, TILE_SORT_CAP>), dim3(num_tiles), dim3(SORT_LONG_WAVES * 64), 0, stream, img.tile_order,
                               img.ranges, &g.hdr->num_rendered, capacity, b.entries, b.scratch, b.point_list, id_bytes,
                               mode == LongListSort::one_workgroup ? 1 : 0, SMALL_CAP + 1, 0);
    }
    const int only = mode == LongListSort::short_lists_expected ? 1 : 0;
    hipLaunchKernelGGL((tile_sort_kernel<4, true, SMALL_CAP>), dim3(num_tiles), dim3(256), 0, stream, img.tile_order,
                       img.ranges, &g.hdr->num_rendered, capacity, b.entries, b.scratch, b.point_list, id_bytes, only ? 0 : 1, 0,
                       only);
}

}  // namespace surfel
