// blend.hip -- per-tile front-to-back alpha blending (forward) and the per-pixel gradient scatter
// (backward) of the surfel rasterizer.  gfx950 only.
//
// Reference behaviour: renderCUDA forward (/root/reference/gs/submodules/diff-surfel-rasterization/
// cuda_rasterizer/forward.cu:265-463) and renderCUDA backward (backward.cu:143-449).
//
// MI355X design (DESIGN.md "blend kernels"):
//   * one 256-thread workgroup (4 wave64) per 16x16 tile; each wave owns an 8x8 pixel quadrant
//     (compact footprint, 32-byte row segments on output);
//   * workgroup b takes tile tile_order[b]: longest list first (binning.hip builds the schedule); workgroups are
//     dispatched in index order and round-robin over the 8 XCDs, so the long tiles start first, spread over the
//     XCDs, and the short ones fill in behind them;
//   * the tile's depth-sorted surfel list is staged 256 (forward) / 128 (backward) entries at a
//     time into LDS as 80-byte records (five ds_write_b128 per lane, conflict-free at a 20-dword
//     stride) and read back with wave-uniform (broadcast) ds_read_b128;
//   * while staging, every thread tests its entry's footprint (the conic rho3d <= rc in the pixel plane and the rho2d
//     disc, surfel_math.h: conservative by its margins only) against the four quadrants' rectangles of pixel centres;
//     wave64 ballots turn that into one 64-bit mask per (quadrant, staging wave), and each
//     wave then iterates only over the set bits with scalar bit tricks (s_ff1 / s_and): list entries
//     that cannot reach a wave's 64 pixels are never evaluated.  The pair evaluation itself is
//     branch-free; the only branches in the loop are wave-uniform;
//   * backward: all 64 lanes of a wave work on the same list entry, so the 16 main gradient components are first
//     reduced across the wave with a reduce-scatter (gfx950 lane swaps across the 16-lane rows, DPP inside a row),
//     then combined across the 4 waves with LDS float atomics (16 distinct addresses), and one global atomic per
//     (entry, component) leaves the workgroup per 128-entry batch, lane-contiguous over the batch's 80-byte records.
//     The reference issues up to 16 global atomics per (pixel, entry).  Measured alternatives to the register
//     reduce-scatter (round 2): an LDS transposition, exact-f32 MFMA contraction of the pixel axis
//     (tools/experiments/blend_bwd_mfma_contraction.hip.txt), per-lane LDS atomics for sparse entries, pair
//     compaction, a 4x4-block row walk, a 2x2-block quad walk (round 3) -- none was faster;
//   * the longest tiles of a whole-tile forward get TWO workgroups, one per half of the tile's pixels, whose waves run the
//     transmittance recurrence over two consecutive entries per trip (fwd_pair_walk, round 6): the launch no longer waits
//     for one workgroup's walk of its longest list;
//   * long lists are blended segment-parallel (SPLIT instances: transmittance pre-pass, per-segment blend, in-order
//     combine); callers that read only colour + alpha plane get the LITE instances (aux_planes), which carry
//     nothing else and need no pre-pass: their segments are blended from T = 1 and the combine blends the one segment
//     a pixel saturates in again from the exact start.
#include <algorithm>
#include <cstdlib>

#include "surfel_state.h"
#include "wave_utils.h"

namespace surfel {

// SURFEL_WIDE_RECORD_READS=1 (experiment, round 5): the second half of a record (normal | rgb) read as two whole float4s
// instead of the ds_read_b96 the compiler narrows them to (8 LDS cycles per wave instruction against 4, MI355X_MICROARCH.md):
// measured SLOWER, blend_fwd 150 -> 154 us, blend_bwd 379 -> 384 us (profiles/r05_blend_micro_ab.txt) -- the LDS pipe is not
// what bounds these loops, and the two extra live registers cost the backward six more spills.  Off.
#ifndef SURFEL_WIDE_RECORD_READS
#define SURFEL_WIDE_RECORD_READS 0
#endif
constexpr int FWD_BATCH = 256;  // list entries staged per round (one per thread)
#ifndef SURFEL_BWD_BATCH
#define SURFEL_BWD_BATCH 128
#endif
constexpr int BWD_BATCH = SURFEL_BWD_BATCH;
// What the recorded segments (surfel_state.h) take for granted about these constants, all of which -D can override:
//  * blend_fwd_kernel leaves a record only when a batch STARTS at a segment boundary (`base == rec_next`) and names the
//    segment it is in base / rec_len: a segment is a whole number of forward batches -- and blend_bwd_kernel's "a walk that
//    stopped in segment `stop` has no contributor beyond its end" rests on the same;
//  * every tile the table lists has at least two segments (one prefix record and the final one in different slots);
//  * bwd_prepare_kernel scans REC_MAX_TILES / 256 positions per thread.
static_assert(REC_SEG_LEN % FWD_BATCH == 0, "a recorded segment must be a whole number of forward batches");
static_assert(REC_MIN >= REC_SEG_LEN, "a tile with recorded segments must have at least two of them");
static_assert(REC_MAX_TILES % 256 == 0, "bwd_prepare_kernel scans REC_MAX_TILES / 256 schedule positions per thread");
static_assert(BWD_BATCH % 64 == 0 && FWD_BATCH % 64 == 0, "cull masks are 64-entry words");

struct TileCoord {
    int tile, tx, ty;
    bool valid;
};

// Stacked frames (surfel_state.h): the tile grid handed to these kernels is `frames` grids of one frame on top of each
// other (grid_y = frames * ceil(H / 16)), image planes are (frames, H, W).  Makes tc.ty the row inside its frame and
// returns the offset of that frame in a plane; `plane` = distance between two planes.  One frame: 0 and H * W.
__device__ __forceinline__ size_t frame_of_tile(TileCoord& tc, int W, int H, int grid_y_total, size_t& plane)
{
    const int fgy = (H + TILE - 1) / TILE;
    const int frame = tc.ty / fgy;
    tc.ty -= frame * fgy;
    plane = (size_t)W * H * (size_t)(grid_y_total / fgy);
    return (size_t)frame * W * H;
}

// Stages one surfel record (q0..q4) into LDS slot `slot` and returns its footprint (q5, q6: surfel_math.h
// contribution_footprint) ready for the quadrant tests.  `NoFootprint()`: what a thread without an entry passes on.
__device__ __forceinline__ FootprintTest stage_record(float4* s_rec, int slot, const float* rec, uint32_t id)
{
    const float4* src = reinterpret_cast<const float4*>(rec + (size_t)id * REC_FLOATS);
    const float4 a = src[0], b = src[1], c = src[2], d = src[3], e = src[4], f0 = src[5], f1 = src[6];
    float4* dst = s_rec + slot * 5;
    dst[0] = a;
    dst[1] = b;
    dst[2] = c;
    dst[3] = d;
    dst[4] = e;
    const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    return footprint_test(f, c.y, c.z);
}

__device__ __forceinline__ FootprintTest no_footprint()
{
    const float f[8] = {0.f, 0.f, 0.f, -1.f, 0.f, 0.f, -1.f, 0.f};
    return footprint_test(f, 0.f, 0.f);
}

// The quadrant version of the cull masks below (four 8x8 quadrants, s_mask[q][w] = entries 64w..64w+63 relevant to quadrant
// q): rounds 1-3's walk, still what blend_combine_kernel's repair uses -- on the frames that need a repair (dense, saturating)
// the footprints cover whole quadrants and the half walk's bookkeeping buys nothing (measured, round 4: 148 -> 163 us).
__device__ __forceinline__ void publish_cull_masks(unsigned long long (*s_mask)[4], bool valid, const FootprintTest& foot,
                                                   int tile_x0, int tile_y0, int wave, int lane, bool no_cull)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float rx0 = (float)(tile_x0 + (q & 1) * 8) + 0.5f, ry0 = (float)(tile_y0 + (q >> 1) * 8) + 0.5f;
        const bool hit = valid && (no_cull || footprint_hits(foot, rx0, rx0 + 7.0f, ry0, ry0 + 7.0f));
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_mask[q][wave] = m;
    }
}

// Cull masks.  A thread that staged a list entry knows its footprint (the conic rho3d <= rc and the rho2d disc); for each
// 8x4 pixel block of the tile -- block 2q + h = rows 4h..4h+3 of the 8x8 quadrant q, i.e. the pixels of lanes 32h..32h+31 of
// wave q -- a wave64 ballot says which of the 64 entries the calling wave holds can reach a pixel centre of that block:
// s_mask8[b][word] for the blocks first..first+COUNT-1 and the mask word `word` the wave's entries belong to.  Afterwards
// each HALF of wave q walks only the set bits of its own block ("half walk", blend_bwd_kernel): entries that cannot
// contribute to its 32 pixels cost nothing.  (Rounds 1-3 and most of round 4 culled per quadrant, one entry per wave trip.)
//
// no_cull (Vidu4dSurfel*Args::debug_flags, VIDU4D_DEBUG_NO_CULL): every staged entry is handed to every block -- the
// reference's walk (forward.cu:359-405 evaluates every list entry for every pixel of the tile).  The culls only prune work;
// tests/test_gpu_round4.py holds the kernels to "bit-identical with and without".
template <int WORDS, int COUNT>
__device__ __forceinline__ void publish_block_masks(unsigned long long (*s_mask8)[WORDS], bool valid, const FootprintTest& foot,
                                                    int tile_x0, int tile_y0, int word, int first, int lane, bool no_cull)
{
#pragma unroll
    for (int i = 0; i < COUNT; i++) {
        const int b = first + i, q = b >> 1, h = b & 1;
        const float rx0 = (float)(tile_x0 + (q & 1) * 8) + 0.5f, ry0 = (float)(tile_y0 + (q >> 1) * 8 + 4 * h) + 0.5f;
        const bool hit = valid && (no_cull || footprint_hits(foot, rx0, rx0 + 7.0f, ry0, ry0 + 3.0f));
        const unsigned long long m = __ballot(hit);
        if (lane == 0) s_mask8[b][word] = m;
    }
}

// __any() on a predicate that is already a lane mask: hip's __any(int) widens the bool to an int per lane first (v_cndmask
// 0 / 1, v_cmp_ne: two VALU instructions of the expensive kind per call in the blend loops); the ballot builtin takes the mask.
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Where a workgroup of the blend kernels works.  Unsplit launch: workgroup b blends all of tile
// tile_order[b] -- longest list first (binning.hip, tile_order_kernel); consecutive workgroup ids go
// to consecutive XCDs, so the long tiles are also spread over the 8 XCDs.  Split launch: workgroups [0, num_segments) take one segment each of the tiles
// longer than SPLIT_MIN (found by bisection of the segment prefix over the schedule positions, which
// hold those tiles at the front), the following `tiles` workgroups take the remaining whole tiles.
struct WorkItem {
    TileCoord tc;
    int seg;       // -1: the whole tile
    uint32_t slot; // segment slot in seg_data
    bool valid;
};

// The largest schedule position p in [0, n) with prefix[p] <= idx (prefix non-decreasing, prefix[0] = 0): a search with 64
// keys per step, one per lane -- one load round trip per step instead of one per bisection step (2 steps for 2048 positions
// instead of 11; the bisection cost every workgroup of a segment-parallel launch 10-20 us before its first useful instruction).
// MINUS_POS: the keys are prefix[p] - p (the FULL segments in front of position p when prefix counts all segments and
// every position holds exactly one remainder: still non-decreasing).
template <bool MINUS_POS = false>
__device__ __forceinline__ uint32_t search_positions(const uint32_t* __restrict__ prefix, uint32_t n, uint32_t idx, int lane)
{
    uint32_t lo = 0;  // the answer lies in [lo, lo + n)
    while (n > 1) {
        const uint32_t step = (n + 63u) / 64u;
        const uint32_t p = lo + (uint32_t)lane * step;
        const bool in = (uint32_t)lane * step < n;
        const uint32_t key = in ? prefix[p] - (MINUS_POS ? p : 0u) : 0xffffffffu;
        const int cnt = __builtin_popcountll(__ballot(in && key <= idx));   // (keys ascend: a prefix of the lanes)
        const uint32_t first = (uint32_t)(cnt - 1) * step;
        lo += first;
        n = min(step, n - first);
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

// The same over the positions g, g + 8, g + 16, ... (n of them): the largest k in [0, n) with prefix[g + 8 k] <= idx.
__device__ __forceinline__ uint32_t search_positions_mod8(const uint32_t* __restrict__ prefix, uint32_t g, uint32_t n, uint32_t idx,
                                                          int lane)
{
    uint32_t lo = 0;
    while (n > 1) {
        const uint32_t step = (n + 63u) / 64u;
        const uint32_t k = lo + (uint32_t)lane * step;
        const bool in = (uint32_t)lane * step < n;
        const uint32_t key = in ? prefix[g + 8u * k] : 0xffffffffu;
        const int cnt = __builtin_popcountll(__ballot(in && key <= idx));
        const uint32_t first = (uint32_t)(cnt - 1) * step;
        lo += first;
        n = min(step, n - first);
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

template <bool SPLIT>
__device__ __forceinline__ WorkItem find_work(const Header* hdr, const ImageState& img, int grid_x, int grid_y,
                                              bool overflow, uint32_t pos = blockIdx.x)
{
    WorkItem w;
    w.seg = -1;
    w.slot = 0;
    w.valid = true;
    int tile;
    if (SPLIT) {
        const uint32_t nseg = hdr->num_segments;
        if (blockIdx.x < nseg) {
            if (overflow) {  // binning buffer too small (seg_data too): the whole-tile workgroups render the background
                w.valid = false;
                return w;
            }
            // largest position whose prefix is <= blockIdx.x
            const uint32_t lo = search_positions(img.seg_prefix, hdr->num_split_pos, blockIdx.x, threadIdx.x & 63);
            tile = (int)img.tile_order[lo];
            w.seg = (int)(blockIdx.x - img.seg_prefix[lo]);
            w.slot = blockIdx.x;
        } else {
            const uint32_t pos = blockIdx.x - nseg;
            if (pos >= (uint32_t)(grid_x * grid_y)) {
                w.valid = false;
                return w;
            }
            tile = (int)img.tile_order[pos];
            if (!overflow && img.seg_first[tile] != SEG_NONE) w.valid = false;  // blended by its segments
        }
    } else {
        tile = (int)img.tile_order[pos];   // (an unsplit launch: the schedule position, blend_fwd_kernel's paired workgroups in front)
    }
    w.tc.tile = tile;
    w.tc.valid = true;
    w.tc.tx = tile % grid_x;
    w.tc.ty = tile / grid_x;
    return w;
}

// Recorded segments (Header::split_used == 2; surfel_state.h): where a workgroup of blend_bwd works.  The split tiles are
// exactly the schedule positions [0, S) (tile_order: split by length class), tile at position p has n_p = its segment
// count, of which n_p - 1 are FULL segments (REC_SEG_LEN entries) and the last one the remainder; the forward's walk of
// the tile reached the first live_p of the full ones (ImageState::live_prefix: nothing behind them holds a contributor).
//   [0, F)        F = Header::num_live_full: the LIVE full segments, position by position -- equal units, dispatched first;
//   [F, F + T)    one "tail" per tile, largest first (ImageState::tail_order): the remainder segment of a split tile (if
//                 the walk got that far), or a whole unsplit tile -- the launch drains on its smallest units;
//   beyond        nothing (the host's grid is an upper bound) -- at the END of the dispatch order, where they delay nobody.
// A full segment finds its position by search_positions over live_prefix.
__device__ __forceinline__ WorkItem find_work_recorded(const Header* hdr, const ImageState& img, int grid_x, int grid_y,
                                                       const float* __restrict__ seg_data)
{
    WorkItem w;
    w.seg = -1;
    w.slot = 0;
    w.valid = true;
    const uint32_t S = hdr->num_split_pos, F = hdr->num_live_full;
    const uint32_t idx = blockIdx.x;
    const int lane = threadIdx.x & 63;
    int tile;
    if (idx < F && hdr->live_xcd) {
        // XCD-local schedule (round 6): workgroup 8 i + g runs on XCD g and takes the i-th live full segment of the schedule
        // positions = g mod 8 -- whose tiles are the ones dealt to XCD g (binning.hip grouped_order) -- so the segments of a
        // tile gather their records through the L2 its neighbours' segments use.  live_prefix holds the prefix per residue
        // class, F is 8 x the largest class total (bwd_prepare_kernel): a class with fewer segments leaves its last few
        // workgroups without work.
        const uint32_t g = idx & 7u, i = idx >> 3;
        if (S <= g) {
            w.valid = false;
            return w;
        }
        const uint32_t pos = g + 8u * search_positions_mod8(img.live_prefix, g, (S - g + 7u) / 8u, i, lane);
        const uint32_t seg = i - img.live_prefix[pos];
        if (seg >= img.live_count[pos]) {
            w.valid = false;
            return w;
        }
        tile = (int)img.tile_order[pos];
        w.seg = (int)seg;
        w.slot = img.seg_first[tile] + seg;
    } else if (idx < F) {
        // (positions without a live full segment repeat their neighbour's key: the search takes the last of equal keys)
        const uint32_t lo = search_positions(img.live_prefix, S, idx, lane);
        tile = (int)img.tile_order[lo];
        w.seg = (int)(idx - img.live_prefix[lo]);
        w.slot = img.seg_first[tile] + (uint32_t)w.seg;
    } else {
        const uint32_t j = idx - F;
        if (j >= (uint32_t)(grid_x * grid_y)) {
            w.valid = false;
            return w;
        }
        tile = (int)img.tail_order[j];
        const uint32_t first = img.seg_first[tile];
        if (first != SEG_NONE) {   // the remainder of a split tile: its last segment
            const uint32_t len = img.ranges[2 * tile + 1] - img.ranges[2 * tile];
            w.seg = (int)((len + hdr->seg_len - 1u) / hdr->seg_len) - 1;
            w.slot = first + (uint32_t)w.seg;
            // (the segment the forward's walk stopped in, as its final record says: short of this one, nothing to do here)
            const uint32_t stop = __float_as_uint(seg_data[((size_t)w.slot * REC_REC_FLOATS + RS_STOP) * 256]);
            if (stop < (uint32_t)w.seg) w.valid = false;
        }
    }
    w.tc.tile = tile;
    w.tc.valid = true;
    w.tc.tx = tile % grid_x;
    w.tc.ty = tile / grid_x;
    return w;
}

// The backward of a SEGMENT-PARALLEL forward (Header::split_used == 1), dispatched like the recorded one (round 5): the
// split tiles are the schedule positions [0, S) (by length class), position p holds n_p = seg_prefix[p + 1] - seg_prefix[p]
// segments of which n_p - 1 are FULL (SEG_LEN entries);
//   [0, F)      F = num_segments - S: the full segments, position by position -- equal units, dispatched first;
//   [F, F + T)  one tail per tile, largest first (ImageState::tail_order): the remainder segment of a split tile or a whole
//               unsplit tile (up to SPLIT_MIN entries: twice a segment -- dispatched in position order behind ALL segments,
//               as until round 5, the largest units of the launch started last: blend_bwd 427 us against 359 us for the
//               recorded backward of the same frames, dense Stage-3 ball);
//   beyond      nothing.
__device__ __forceinline__ WorkItem find_work_split_ordered(const Header* hdr, const ImageState& img, int grid_x, int grid_y)
{
    WorkItem w;
    w.seg = -1;
    w.slot = 0;
    w.valid = true;
    const uint32_t S = hdr->num_split_pos, F = hdr->num_segments - S;
    const uint32_t idx = blockIdx.x;
    int tile;
    if (idx < F) {
        const uint32_t lo = search_positions<true>(img.seg_prefix, S, idx, threadIdx.x & 63);
        tile = (int)img.tile_order[lo];
        w.seg = (int)(idx - (img.seg_prefix[lo] - lo));
        w.slot = img.seg_first[tile] + (uint32_t)w.seg;
    } else {
        const uint32_t j = idx - F;
        if (j >= (uint32_t)(grid_x * grid_y)) {
            w.valid = false;
            return w;
        }
        tile = (int)img.tail_order[j];
        const uint32_t first = img.seg_first[tile];
        if (first != SEG_NONE) {   // the remainder of a split tile: its last segment
            const uint32_t len = img.ranges[2 * tile + 1] - img.ranges[2 * tile];
            w.seg = (int)((len + hdr->seg_len - 1u) / hdr->seg_len) - 1;
            w.slot = first + (uint32_t)w.seg;
        }
    }
    w.tc.tile = tile;
    w.tc.valid = true;
    w.tc.tx = tile % grid_x;
    w.tc.ty = tile / grid_x;
    return w;
}

// Deepest list position blended by any pixel of this workgroup -> frame-wide maximum (a hint for the
// caller's split decision; the longest tiles run first, so most workgroups only read).
__device__ __forceinline__ void report_depth(uint32_t* depth_used, uint32_t last_contributor)
{
    if (!depth_used) return;
    uint32_t v = last_contributor;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o > v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v > *depth_used) atomicMax(depth_used, v);
}

// Smallest final transmittance of the frame (bits of a non-negative float order like unsigned integers).
__device__ __forceinline__ void report_min_T(Header* hdr, float T, bool inside)
{
    uint32_t v = inside ? __float_as_uint(fmaxf(T, 0.f)) : 0x3f800000u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o < v ? o : v;
    }
    if ((threadIdx.x & 63) == 0 && v < hdr->min_T_bits) atomicMin(&hdr->min_T_bits, v);
}

__device__ __forceinline__ void write_pixel(const FwdPixel& s, size_t HW, size_t pid, const float* bg, float* final_T,
                                            uint32_t* n_contrib, float* out_color, float* out_others)
{
    final_T[pid] = s.T;
    final_T[pid + HW] = s.dist1;
    final_T[pid + 2 * HW] = s.dist2;
    n_contrib[pid] = s.last_contributor;
    n_contrib[pid + HW] = s.median_contributor;
    for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pid] = s.C[ch] + s.T * bg[ch];
    out_others[pid] = s.D;
    out_others[pid + HW] = 1.0f - s.T;
    out_others[pid + 2 * HW] = s.N[0];
    out_others[pid + 3 * HW] = s.N[1];
    out_others[pid + 4 * HW] = s.N[2];
    out_others[pid + 5 * HW] = s.median_depth;
    out_others[pid + 6 * HW] = s.distortion;
    out_others[pid + 7 * HW] = s.median_weight;
}

// Pass 1 of the segment-parallel forward: the product of (1 - alpha) over one segment, per pixel
// (the same acceptance test and the same multiplication as the blend itself, minus everything else).
__global__ __launch_bounds__(256) void blend_seg_T_kernel(int W, int H, int grid_x, int grid_y, Header* hdr,
                                                         ImageState img, const uint32_t* __restrict__ point_list,
                                                         int64_t capacity, int max_seg, const float* __restrict__ rec,
                                                         float* __restrict__ seg_data, int flags)
{
    __shared__ float4 s_rec[(FWD_BATCH + 1) * 5];  // (+ the all-zero record of an idle half, see blend_bwd_kernel)
    __shared__ unsigned long long s_mask8[8][FWD_BATCH / 64];
    if (threadIdx.x < 5) s_rec[FWD_BATCH * 5 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool overflow = (int64_t)hdr->num_rendered > capacity;
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr->split_used = !overflow && hdr->num_segments > 0;  // for backward
    if (overflow || blockIdx.x >= hdr->num_segments) return;
    const WorkItem wk = find_work<true>(hdr, img, grid_x, grid_y, false);
    if (wk.seg >= max_seg) return;
    TileCoord tc = wk.tc;
    size_t plane_unused;
    (void)frame_of_tile(tc, W, H, grid_y, plane_unused);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tc.tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = tc.ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const float pixx = (float)px + 0.5f, pixy = (float)py + 0.5f;
    const uint32_t r0 = img.ranges[2 * tc.tile], r1 = img.ranges[2 * tc.tile + 1];
    const int begin = wk.seg * SEG_LEN;
    int todo = min((int)(r1 - r0) - begin, SEG_LEN);
    float T = 1.0f;
    for (int base = begin; todo > 0; base += FWD_BATCH, todo -= FWD_BATCH) {
        __syncthreads();
        const bool have = (int)threadIdx.x < todo;
        FootprintTest foot = no_footprint();
        if (have) foot = stage_record(s_rec, threadIdx.x, rec, point_list[r0 + base + threadIdx.x]);
        publish_block_masks<FWD_BATCH / 64, 8>(s_mask8, have, foot, tc.tx * TILE, tc.ty * TILE, wave, 0, lane, flags & FLAG_NO_CULL);
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < FWD_BATCH / 64; k++) {  // (half walk, as in blend_fwd_kernel)
            unsigned long long mA = uniform_u64(s_mask8[2 * wave][k]), mB = uniform_u64(s_mask8[2 * wave + 1][k]);
            while (mA | mB) {
                const int jA = mA ? k * 64 + __builtin_ctzll(mA) : FWD_BATCH, jB = mB ? k * 64 + __builtin_ctzll(mB) : FWD_BATCH;
                mA &= mA - 1;
                mB &= mB - 1;
                const int j = lane < 32 ? jA : jB;
                const float4 a0 = s_rec[j * 5 + 0], a1 = s_rec[j * 5 + 1], a2 = s_rec[j * 5 + 2];
                const float Tu[3] = {a0.x, a0.y, a0.z}, Tv[3] = {a0.w, a1.x, a1.y}, Tw[3] = {a1.z, a1.w, a2.x};
                PairEval e;
                if (eval_pair_flat(Tu, Tv, Tw, a2.y, a2.z, a2.w, pixx, pixy, e)) T = T * (1.0f - e.alpha);
            }
        }
    }
    seg_data[((size_t)wk.slot * SEG_FLOATS + SG_TSEG) * 256 + threadIdx.x] = T;
}

// One record of the recorded segments (surfel_state.h RecordedSlot): the sums of the segment the walk is leaving; the
// next segment's sums start from zero (the totals are put together from the records when the walk ends: the registers of
// seven more running sums would cost the kernel two of its seven waves per SIMD).
template <int MODE>
__device__ __forceinline__ void store_record(float* __restrict__ seg_data, uint32_t slot, FwdPixel& s, float first)
{
    float* d = seg_data + (size_t)slot * REC_REC_FLOATS * 256 + threadIdx.x;
    d[RS_T * 256] = first;
    for (int ch = 0; ch < 3; ch++) {
        d[(RS_C + ch) * 256] = s.C[ch];
        s.C[ch] = 0.f;
    }
    if (MODE != BLEND_LITE) {
        d[RS_D * 256] = s.D;
        s.D = 0.f;
        for (int ch = 0; ch < 3; ch++) {
            d[(RS_N + ch) * 256] = s.N[ch];
            s.N[ch] = 0.f;
        }
    }
    if (MODE == BLEND_FULL) {
        d[RS_M1 * 256] = s.dist1;
        d[RS_M2 * 256] = s.dist2;
    }
}

// ---------------------------------------------------------------------------------------------
// Long tiles on TWO workgroups ("paired workgroups", "two-entry walk").  A launch whose tiles all start at
// once (the dense Stage-3 ball: fewer long tiles than the chip has workgroup slots) lasts as long as its longest list takes
// ONE workgroup, and the second half of it runs on workgroups that are alone on their CU at one wave per SIMD
// (tools/fwd_trace.py: 35 % of the CU-time with fewer than two workgroups resident).  Cutting the LIST needs the start
// transmittance (a pre-pass, or speculation + repair walks: the segment-parallel instances); this cuts the PIXELS and
// shortens the walk:
//   * the tiles at the schedule positions [0, Header::num_paired) -- the longest, binning.hip tile_order -- get two
//     workgroups each; workgroup `half` of the pair owns the blocks 4 half .. 4 half + 3 (publish_block_masks: 8x4 pixels
//     each), stages every batch of the list itself and culls for its four blocks only.  Pixels are independent: the two
//     never talk to each other;
//   * wave w of such a workgroup owns ONE block and holds every pixel TWICE, in lane p and in lane p + 32;
//   * the two halves of the wave take CONSECUTIVE entries of the block's culled list (two each per trip in the colour +
//     alpha instance, as blend_fwd_kernel's halves; one in the instances held to 72 registers): both evaluate their pair (forward.cu:358-399, independent of the transmittance), exchange
//     alpha (and the mapped depth, full instance) with one v_permlane32_swap, and both run the transmittance recurrence
//     over the two entries in list order -- T1 = T (1 - a1), T2 = T1 (1 - a2), the same two roundings per entry, an entry
//     that was not accepted enters as alpha = 0 (T * 1 is exact) -- so T, the end of the walk (forward.cu:400-405) and, full
//     instance, the moments in front of every entry (forward.cu:407-414) and the median sample (:416-421) are
//     bit-identical to the one-entry walk;
//   * each half accumulates ITS entries' colour / depth / normal / distortion sums; the halves are added when a recorded
//     segment ends and when the walk ends -- these sums differ from the unpaired walk's by fp32 re-association (even
//     entries + odd entries), the transmittance, contributor counts and median sample do not.
// Per wave half the trips of the unpaired walk at ~1.3 x the instructions per entry (the exchange, the second recurrence,
// the selects; measured with EVERY tile on eight waves: headline blend_fwd 148 -> 210 us, which is why only the tail's tiles
// are paired) plus the second staging: the pair costs more VALU work than the one workgroup and ends in half the time --
// where the launch waits for it.  Records, outputs, schedule otherwise as the unpaired walk.
__device__ __forceinline__ void both_halves(float v, float& lo, float& hi)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    lo = __uint_as_float(r[0]);   // what lane (L & 31) holds, in both halves
    hi = __uint_as_float(r[1]);   // what lane (L & 31) + 32 holds
}
__device__ __forceinline__ void both_halves(uint32_t v, uint32_t& lo, uint32_t& hi)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    lo = r[0];
    hi = r[1];
}
__device__ __forceinline__ float halves_sum(float v)
{
    float lo, hi;
    both_halves(v, lo, hi);
    return lo + hi;
}

// One entry per half: the lower half's is the earlier one in the list.  Branch-free in the sums: a sample that does not
// count enters them with weight 0 (x * 0 + s == s bit for bit while x is finite: colours and normals of a record are, the
// depth and the mapped depth of a pair that failed its tests are replaced by 0) -- written with `if (live) { ... }` the
// compiler kept every running sum in two registers and copied them back and forth on both sides of the branch: 70 v_mov
// per trip, 1.8 x the unpaired walk's instructions.  The one branch is wave-uniform and rare: "a pixel of this wave ends its
// walk on this pair of entries" (once per pixel).
template <int MODE>
__device__ __forceinline__ void pair_step(FwdPixel& s, bool& done, bool upper, bool ok, const PairEval& e, const float4* r,
                                          uint32_t key)
{
#pragma clang fp contract(off)
    // (a finished pixel's entries enter as alpha = 0 in both halves: its transmittance stays where the walk ended)
    bool live = ok && !done;
    const float a_own = live ? e.alpha : 0.f;
    float a1, a2;
    both_halves(a_own, a1, a2);
    const float T0 = s.T;
    const float T1 = T0 * (1.0f - a1);   // THRESHOLD-EXACT (fwd_accumulate: the same two roundings per entry)
    const float T2 = T1 * (1.0f - a2);
    // (T >= T_EPS while a pixel is not done, so an entry that enters as alpha = 0 never stops the walk)
    const bool stop1 = T1 < T_EPS, stop2 = T2 < T_EPS;
    const float T_own = upper ? T1 : T0;
    const float4 q3 = r[3], q4 = r[4];
    float w = a_own * T_own;            // (0 for an entry that does not count: alpha * T otherwise, as fwd_accumulate)
    float w1 = a1 * T0, w2 = a2 * T1;   // (full instance: the two entries' weights as both halves see them)
    float Tn = T2;
    if (wave_any(stop1 || stop2)) {   // the walk of a pixel ends here: the entry that would take T below T_EPS does not count
        live = live && !stop1 && !(upper && stop2);
        w = live ? w : 0.f;
        w1 = stop1 ? 0.f : w1;
        w2 = (stop1 || stop2) ? 0.f : w2;
        Tn = stop1 ? T0 : (stop2 ? T1 : T2);
        done = done || stop1 || stop2;
    }
    if (MODE == BLEND_FULL) {
        float m1, m2;
        const float m = ok ? map_depth(e.depth) - s.m0 : 0.f;
        both_halves(m, m1, m2);
        // both halves carry the WHOLE moments: behind the first entry, then behind the second (forward.cu:413-414)
        const float M1_0 = s.dist1, M2_0 = s.dist2;
        const float M1a = fmaf(m1, w1, M1_0), M2a = fmaf(m1 * m1, w1, M2_0);
        s.dist1 = fmaf(m2, w2, M1a);
        s.dist2 = fmaf(m2 * m2, w2, M2a);
        // the half's own entry against the moments in front of it (forward.cu:407-411), the median sample (:416-421)
        const float A = 1.0f - T_own;
        const float M1 = upper ? M1a : M1_0, M2 = upper ? M2a : M2_0;
        const float error = fmaf(m * m, A, fmaf(-2.0f * m, M1, M2));
        s.distortion = fmaf(error, w, s.distortion);
        const bool med = live && T_own > 0.5f;
        s.median_depth = med ? e.depth : s.median_depth;
        s.median_weight = med ? w : s.median_weight;
        s.median_contributor = med ? key : s.median_contributor;
    }
    if (MODE != BLEND_LITE) {
        const float depth = live ? e.depth : 0.f;
        s.N[0] = fmaf(q3.x, w, s.N[0]);
        s.N[1] = fmaf(q3.y, w, s.N[1]);
        s.N[2] = fmaf(q3.z, w, s.N[2]);
        s.D = fmaf(depth, w, s.D);
    }
    s.C[0] = fmaf(q4.x, w, s.C[0]);
    s.C[1] = fmaf(q4.y, w, s.C[1]);
    s.C[2] = fmaf(q4.z, w, s.C[2]);
    s.last_contributor = live ? key : s.last_contributor;
    s.T = Tn;
}

// The sums the two halves hold of one pixel, added: afterwards both halves hold the pixel's sums.
template <int MODE>
__device__ __forceinline__ void pair_add_halves(FwdPixel& s)
{
    for (int ch = 0; ch < 3; ch++) s.C[ch] = halves_sum(s.C[ch]);
    if (MODE != BLEND_LITE) {
        for (int ch = 0; ch < 3; ch++) s.N[ch] = halves_sum(s.N[ch]);
        s.D = halves_sum(s.D);
    }
}

// store_record for a pixel index that is not the thread index; the caller has added the halves.
template <int MODE>
__device__ __forceinline__ void pair_store_record(float* __restrict__ seg_data, uint32_t slot, int pix, bool write, FwdPixel& s,
                                                  float first)
{
    float* d = seg_data + (size_t)slot * REC_REC_FLOATS * 256 + pix;
    if (write) {
        d[RS_T * 256] = first;
        for (int ch = 0; ch < 3; ch++) d[(RS_C + ch) * 256] = s.C[ch];
        if (MODE != BLEND_LITE) {
            d[RS_D * 256] = s.D;
            for (int ch = 0; ch < 3; ch++) d[(RS_N + ch) * 256] = s.N[ch];
        }
        if (MODE == BLEND_FULL) {
            d[RS_M1 * 256] = s.dist1;
            d[RS_M2 * 256] = s.dist2;
        }
    }
    for (int ch = 0; ch < 3; ch++) s.C[ch] = 0.f;
    if (MODE != BLEND_LITE) {
        s.D = 0.f;
        for (int ch = 0; ch < 3; ch++) s.N[ch] = 0.f;
    }
}

// One workgroup of a pair: the blocks 4 half .. 4 half + 3 of the tile at schedule position `pos`.
template <int MODE>
__device__ __forceinline__ int fwd_pair_walk(int W, int H, int grid_x, int grid_y, Header* hdr, const ImageState& img,
                                             const uint32_t* __restrict__ point_list, int64_t capacity,
                                             const float* __restrict__ rec, const float* __restrict__ bg,
                                             float* __restrict__ seg_data, float* __restrict__ out_color,
                                             float* __restrict__ out_others, uint32_t* depth_used, int flags, int rec_len,
                                             float4* s_rec, unsigned long long (*s_mask8)[FWD_BATCH / 64], uint32_t pos, int half)
{
#ifndef SURFEL_PAIR_ENTRIES_CAPPED
#define SURFEL_PAIR_ENTRIES_CAPPED 1
#endif
    constexpr int PAIR_ENTRIES = MODE == BLEND_LITE ? 2 : SURFEL_PAIR_ENTRIES_CAPPED;   // entries per half and trip
    const bool overflow = (int64_t)hdr->num_rendered > capacity;
    TileCoord tc;
    tc.tile = (int)img.tile_order[pos];
    tc.valid = true;
    tc.tx = tc.tile % grid_x;
    tc.ty = tc.tile / grid_x;
    size_t plane;
    const size_t frame_base = frame_of_tile(tc, W, H, grid_y, plane);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int block = 4 * half + wave;   // block 2q + h of publish_block_masks
    const bool upper = lane >= 32;
    const int p = lane & 31;
    const int pix = block * 32 + p;      // the pixel's thread index in the unpaired walk (records, segment data)
    const int px = tc.tx * TILE + ((block >> 1) & 1) * 8 + (p & 7);
    const int py = tc.ty * TILE + (block >> 2) * 8 + (block & 1) * 4 + (p >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px + 0.5f, pixy = (float)py + 0.5f;
    const uint32_t r0 = img.ranges[2 * tc.tile], r1 = img.ranges[2 * tc.tile + 1];
    int todo = overflow ? 0 : (int)(r1 - r0);
    const uint32_t rec_first = (rec_len && !overflow) ? img.seg_first[tc.tile] : SEG_NONE;
    int rec_next = rec_len;
    uint32_t rec_stop = 0;
    int walked = 0;

    FwdPixel s;
    if (MODE == BLEND_FULL && todo > 0) {
        s.m0 = map_depth(rec[(size_t)point_list[r0] * REC_FLOATS + R_DEPTH]);
        if (threadIdx.x == 0 && half == 0) img.tile_m0[tc.tile] = s.m0;
    }
    bool done = !inside;
    for (int base = 0; todo > 0; base += FWD_BATCH, todo -= FWD_BATCH) {
        if (rec_first != SEG_NONE && base == rec_next) {  // (wave-uniform) entries [0, base) are behind us
            rec_stop = (uint32_t)(base / rec_len);
            pair_add_halves<MODE>(s);
            pair_store_record<MODE>(seg_data, rec_first + rec_stop - 1u, pix, !upper, s, s.T);
            rec_next += rec_len;
        }
        if (__syncthreads_count(done) == 256) break;
        const int prio_len = (int)(r1 - r0);   // (issue priority by the fraction of the walk that is left: blend_fwd_kernel)
        if (3 * todo > 2 * prio_len) __builtin_amdgcn_s_setprio(3);
        else if (3 * todo > prio_len) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
        walked = base + min(todo, FWD_BATCH);
        const bool have = (int)threadIdx.x < todo;
        FootprintTest foot = no_footprint();
        if (have) foot = stage_record(s_rec, threadIdx.x, rec, point_list[r0 + base + threadIdx.x]);
        if (half == 0)   // (workgroup-uniform: the four blocks of this half)
            publish_block_masks<FWD_BATCH / 64, 4>(s_mask8, have, foot, tc.tx * TILE, tc.ty * TILE, wave, 0, lane, flags & FLAG_NO_CULL);
        else
            publish_block_masks<FWD_BATCH / 64, 4>(s_mask8, have, foot, tc.tx * TILE, tc.ty * TILE, wave, 4, lane, flags & FLAG_NO_CULL);
        __syncthreads();
        if (__all(done)) continue;
        const int key_base = (base + 1) * 80;   // (keys: blend_fwd_kernel)
#pragma unroll 1
        for (int k = 0; k < FWD_BATCH / 64; k++) {
            unsigned long long m = wave_any(!done) ? uniform_u64(s_mask8[block][k]) : 0ull;
            while (m) {
                const int o0 = (k * 64 + __builtin_ctzll(m)) * 80;
                m &= m - 1;
                const int o1 = (m ? k * 64 + __builtin_ctzll(m) : FWD_BATCH) * 80;
                m &= m - 1;
                const int oa = upper ? o1 : o0;
                const float4* ra = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + oa);
                const float4 a0 = ra[0], a1 = ra[1], a2 = ra[2];
                const float TuA[3] = {a0.x, a0.y, a0.z}, TvA[3] = {a0.w, a1.x, a1.y}, TwA[3] = {a1.z, a1.w, a2.x};
                PairEval ea;
                const bool okA = eval_pair_flat(TuA, TvA, TwA, a2.y, a2.z, a2.w, pixx, pixy, ea);
                if (PAIR_ENTRIES == 1) {   // (the instances held to 72 registers: one entry per half and trip)
                    pair_step<MODE>(s, done, upper, okA, ea, ra, (uint32_t)(key_base + oa));
                    continue;
                }
                const int o2 = (m ? k * 64 + __builtin_ctzll(m) : FWD_BATCH) * 80;
                m &= m - 1;
                const int o3 = (m ? k * 64 + __builtin_ctzll(m) : FWD_BATCH) * 80;
                m &= m - 1;
                const int ob = upper ? o3 : o2;
                const float4* rb = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + ob);
                const float4 b0 = rb[0], b1 = rb[1], b2 = rb[2];
                const float TuB[3] = {b0.x, b0.y, b0.z}, TvB[3] = {b0.w, b1.x, b1.y}, TwB[3] = {b1.z, b1.w, b2.x};
                PairEval eb;
                const bool okB = eval_pair_flat(TuB, TvB, TwB, b2.y, b2.z, b2.w, pixx, pixy, eb);
                pair_step<MODE>(s, done, upper, okA, ea, ra, (uint32_t)(key_base + oa));
                pair_step<MODE>(s, done, upper, okB, eb, rb, (uint32_t)(key_base + ob));
            }
        }
    }
    // the halves' sums, added; the later of their last contributors and median samples
    pair_add_halves<MODE>(s);
    {
        uint32_t lo, hi;
        both_halves(s.last_contributor, lo, hi);
        s.last_contributor = lo > hi ? lo : hi;
    }
    if (MODE == BLEND_FULL) {
        s.distortion = halves_sum(s.distortion);
        uint32_t lo, hi;
        both_halves(s.median_contributor, lo, hi);
        float d_lo, d_hi, w_lo, w_hi;
        both_halves(s.median_depth, d_lo, d_hi);
        both_halves(s.median_weight, w_lo, w_hi);
        s.median_contributor = lo > hi ? lo : hi;
        s.median_depth = lo > hi ? d_lo : d_hi;
        s.median_weight = lo > hi ? w_lo : w_hi;
    }
    s.last_contributor /= 80u;
    s.median_contributor /= 80u;
    if (rec_first != SEG_NONE) {  // the final sums, in the slot of the tile's last segment (blend_fwd_kernel)
        const uint32_t nseg = (r1 - r0 + (uint32_t)rec_len - 1u) / (uint32_t)rec_len;
        if (!upper) seg_data[((size_t)(rec_first + nseg - 1u) * REC_REC_FLOATS + RS_STOP) * 256 + pix] = __uint_as_float(rec_stop);
        pair_store_record<MODE>(seg_data, rec_first + nseg - 1u, pix, !upper, s, s.median_weight);
        if (!upper) {
            for (uint32_t k = 0; k <= rec_stop; k++) {
                const uint32_t q = k < rec_stop ? k : nseg - 1u;
                const float* e = seg_data + (size_t)(rec_first + q) * REC_REC_FLOATS * 256 + pix;
                for (int ch = 0; ch < 3; ch++) s.C[ch] += e[(RS_C + ch) * 256];
                if (MODE != BLEND_LITE) {
                    for (int ch = 0; ch < 3; ch++) s.N[ch] += e[(RS_N + ch) * 256];
                    s.D += e[RS_D * 256];
                }
            }
        }
    }
    if (inside && !upper)
        write_pixel(s, plane, frame_base + (size_t)py * W + px, bg, img.final_T, img.n_contrib, out_color, out_others);
    report_depth(depth_used, s.last_contributor);
    report_min_T(hdr, s.T, inside);
    if (rec_len && threadIdx.x == 0 && rec_first != SEG_NONE) {
        // the tile's live full segments: the farther of the two workgroups' walks (tile_order has zeroed the counts)
        const uint32_t nseg = (r1 - r0 + (uint32_t)rec_len - 1u) / (uint32_t)rec_len;
        atomicMax(&img.live_count[pos], rec_stop + 1u < nseg - 1u ? rec_stop + 1u : nseg - 1u);
    }
    return walked;
}

// The blend (pass 2 of the segment-parallel forward when SPLIT).  MODE (surfel_math.h): BLEND_LITE carries colour + alpha
// plane only, BLEND_GEOM colour + planes 0-4 (fwd_accumulate<MODE>); what an instance does not carry comes out as zeros.
//
// spec (SPLIT && MODE != BLEND_FULL only; Vidu4dSurfelForwardArgs::assume_unsaturated): no transmittance pre-pass ran.  A
// segment is blended from T = 1 -- colour, depth and normal sums are linear in the start transmittance and, as long as the
// pixel does not saturate, no decision depends on it (the full instance's median sample does: T > 0.5) -- and stores its sums
// and its transmittance PRODUCT; blend_combine_kernel scales by the product of the predecessors and blends the ONE segment
// in which a pixel comes near the saturation threshold again, in order, from the exact start.
//
// rec_len (!SPLIT only; 0: off): the walk leaves recorded segments for the backward (surfel_state.h): a prefix record after
// every rec_len-th entry of a tile the segment table lists, the final sums in the tile's last slot.
// Register budget: the whole-tile full instance is held to 72 VGPRs = 7 waves per SIMD (the LDS limit; the allocator takes
// 74 = 6 waves when left alone since the segment-local distortion moments joined the pixel state).
#ifdef SURFEL_FWD_TRACE   // (tools/fwd_trace.py: a variant build that timestamps every workgroup of blend_fwd; never in the product)
__device__ unsigned long long* g_fwd_trace = nullptr;
}  // namespace surfel
extern "C" int vidu4d_diag_set_forward_trace(void* buffer)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(surfel::g_fwd_trace), &buffer, sizeof(buffer));
}
namespace surfel {
#endif
#ifndef SURFEL_FWD_WAVES_PER_EU
#define SURFEL_FWD_WAVES_PER_EU 7
#endif
template <bool SPLIT, int MODE>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu((!SPLIT && MODE != BLEND_LITE) ? SURFEL_FWD_WAVES_PER_EU : 1, (!SPLIT && MODE != BLEND_LITE) ? SURFEL_FWD_WAVES_PER_EU : 8)))
void blend_fwd_kernel(int W, int H, int grid_x, int grid_y, Header* hdr,
                                                       ImageState img, const uint32_t* __restrict__ point_list,
                                                       int64_t capacity, int max_seg, const float* __restrict__ rec,
                                                       const float* __restrict__ bg, float* __restrict__ seg_data,
                                                       float* __restrict__ out_color, float* __restrict__ out_others,
                                                       uint32_t* depth_used, int spec, int flags, int rec_len)
{
    constexpr bool SPEC_OK = MODE != BLEND_FULL;
#ifdef SURFEL_FWD_TRACE
    struct TraceEnd {
        unsigned long long* p; unsigned long long t0; int entries;
        __device__ ~TraceEnd() {
            if (p && threadIdx.x == 0) {
                p[4 * blockIdx.x + 0] = t0;
                p[4 * blockIdx.x + 1] = t0;
                p[4 * blockIdx.x + 2] = wall_clock64();
                p[4 * blockIdx.x + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) |
                                        ((unsigned long long)(__builtin_amdgcn_s_getreg(0xF814) & 0xf) << 32) |
                                        ((unsigned long long)(entries & 0xffff) << 40);
            }
        }
    } trace_end{g_fwd_trace, (unsigned long long)wall_clock64(), 0};
#endif
    __shared__ float4 s_rec[(FWD_BATCH + 1) * 5];  // (+ an all-zero record: what an idle half of a wave evaluates)
    __shared__ unsigned long long s_mask8[8][FWD_BATCH / 64];
    if (threadIdx.x < 5) s_rec[FWD_BATCH * 5 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool overflow = (int64_t)hdr->num_rendered > capacity;
    if (SPLIT && SPEC_OK && spec && blockIdx.x == 0 && threadIdx.x == 0)  // (what blend_seg_T_kernel does when it runs)
        hdr->split_used = !overflow && hdr->num_segments > 0;
    if (!SPLIT && rec_len && blockIdx.x == 0 && threadIdx.x == 0)
        hdr->split_used = (!overflow && hdr->num_segments > 0) ? 2u : 0u;
    // Unsplit launch: the workgroups [0, 2 num_paired) are the pairs of the longest tiles (fwd_pair_walk), workgroup
    // num_paired + p takes schedule position p >= num_paired; the host's grid is tiles + its bound on num_paired.
    uint32_t pos = blockIdx.x;
    if (!SPLIT) {
        const uint32_t np = hdr->num_paired;
        if (blockIdx.x < 2u * np) {
            const int walked = fwd_pair_walk<MODE>(W, H, grid_x, grid_y, hdr, img, point_list, capacity, rec, bg, seg_data, out_color,
                                                   out_others, depth_used, flags, rec_len, s_rec, s_mask8, blockIdx.x >> 1,
                                                   (int)(blockIdx.x & 1u));
#ifdef SURFEL_FWD_TRACE
            trace_end.entries = walked;
#endif
            (void)walked;
            return;
        }
        pos = blockIdx.x - np;
        if (pos >= (uint32_t)(grid_x * grid_y)) return;
    }
    // (SPLIT: full segments first, then the remainders and the unsplit tiles by descending size -- find_work_split_ordered;
    // until round 5 in schedule-position order, the unsplit tiles of up to SPLIT_MIN entries, twice a segment, last)
    const WorkItem wk = (SPLIT && !overflow && !(flags & FLAG_POSITION_ORDER)) ? find_work_split_ordered(hdr, img, grid_x, grid_y)
                                                                              : find_work<SPLIT>(hdr, img, grid_x, grid_y, overflow, pos);
    if (!wk.valid || (SPLIT && wk.seg >= max_seg)) return;
    TileCoord tc = wk.tc;
    size_t plane;
    const size_t frame_base = frame_of_tile(tc, W, H, grid_y, plane);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tc.tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = tc.ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px + 0.5f, pixy = (float)py + 0.5f;
    const uint32_t r0 = img.ranges[2 * tc.tile], r1 = img.ranges[2 * tc.tile + 1];
    // binning buffer too small for this frame: nothing was emitted, render the background only (the
    // caller re-runs with a larger buffer)
    int todo = overflow ? 0 : (int)(r1 - r0);
    int begin = 0;
    // recorded segments: first slot of this tile (SEG_NONE: the table does not list it) and where the next record is due
    const uint32_t rec_first = (!SPLIT && rec_len && !overflow) ? img.seg_first[tc.tile] : SEG_NONE;
    int rec_next = rec_len;
    uint32_t rec_stop = 0;  // segment of the batch the walk is in

    FwdPixel s;
    if (MODE == BLEND_FULL && todo > 0) {
        // reference of the tile's distortion moments (surfel_math.h FwdPixel::m0): its first list entry's mapped depth
        // (every workgroup of a split tile computes the same value)
        s.m0 = map_depth(rec[(size_t)point_list[r0] * REC_FLOATS + R_DEPTH]);
        if (threadIdx.x == 0) img.tile_m0[tc.tile] = s.m0;
    }
    bool done = !inside;
    bool dead = false;
    if (SPLIT && wk.seg >= 0) {
        begin = wk.seg * SEG_LEN;
        todo = min(todo - begin, SEG_LEN);
        // transmittance left by the earlier segments of this tile; below T_EPS the pixel saturated
        // inside one of them (the running product only decreases), and this segment adds nothing
        if (!(SPEC_OK && spec)) {
            const uint32_t first = wk.slot - (uint32_t)wk.seg;
            for (int q = 0; q < wk.seg; q++)
                s.T = s.T * seg_data[((size_t)(first + q) * SEG_FLOATS + SG_TSEG) * 256 + threadIdx.x];
            dead = wk.seg > 0 && s.T < T_EPS;
            done = done || dead;
        }
    }
    bool sat_local = false;  // (spec: a sample was refused for saturation although the walk started from T = 1)
    for (int base = begin; todo > 0; base += FWD_BATCH, todo -= FWD_BATCH) {
        if (!SPLIT && rec_first != SEG_NONE && base == rec_next) {  // (wave-uniform) entries [0, base) are behind us
            rec_stop = (uint32_t)(base / rec_len);
            store_record<MODE>(seg_data, rec_first + rec_stop - 1u, s, s.T);
            rec_next += rec_len;
        }
        if (__syncthreads_count(done) == 256) break;
        // Issue priority by the fraction of the walk that is left.  A CU serves its oldest waves first, so of seven resident
        // workgroups with like lists the first dispatched finished after 93 us and the last after 154 (tools/fwd_trace.py), and
        // the last third of the launch ran on two or three workgroups per CU; with the ones that have most left in front they
        // finish together: 168 -> 157 us at the headline size.
        const int prio_len = (int)(r1 - r0);   // (against the frame's longest list instead -- the long walks ahead -- is worse: 431 -> 448 us on the dense ball)
        if (3 * todo > 2 * prio_len) __builtin_amdgcn_s_setprio(3);
        else if (3 * todo > prio_len) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
#ifdef SURFEL_FWD_TRACE
        trace_end.entries = base - begin + min(todo, FWD_BATCH);
#endif
        const bool have = (int)threadIdx.x < todo;
        FootprintTest foot = no_footprint();
        if (have) foot = stage_record(s_rec, threadIdx.x, rec, point_list[r0 + base + threadIdx.x]);
        publish_block_masks<FWD_BATCH / 64, 8>(s_mask8, have, foot, tc.tx * TILE, tc.ty * TILE, wave, 0, lane, flags & FLAG_NO_CULL);
        __syncthreads();
        if (__all(done)) continue;  // this wave's 64 pixels are saturated; it keeps helping to stage
        // Half walk (see blend_bwd_kernel): lanes 0-31 / 32-63 follow the cull masks of their own 8x4 block, two list
        // entries of each per trip; the accumulation stays in list order per pixel.
        const int key_base = (base + 1) * 80;   // (list positions stay below 2^31 / 80: the pair count is an int64 capacity, a tile's list a uint32 range)
#pragma unroll 1
        for (int k = 0; k < FWD_BATCH / 64; k++) {
            const unsigned long long alive = __ballot(!done);  // a half whose pixels are all finished walks nothing
            unsigned long long mA = (uint32_t)alive ? uniform_u64(s_mask8[2 * wave][k]) : 0ull;
            unsigned long long mB = (alive >> 32) ? uniform_u64(s_mask8[2 * wave + 1][k]) : 0ull;
            while (mA | mB) {
                // The slots travel as BYTE OFFSETS into s_rec (80 j; the scalar unit multiplies): `ja * 5` of a per-lane j
                // compiled to v_mul_lo_u32, a quarter-rate instruction, twice per trip.  An accepted sample's list position
                // is kept as the key 80 (base + j + 1) = offset + key_base and divided back once per pixel after the walk.
                const int oaA = (mA ? k * 64 + __builtin_ctzll(mA) : FWD_BATCH) * 80;
                mA &= mA - 1;
                const int obA = (mA ? k * 64 + __builtin_ctzll(mA) : FWD_BATCH) * 80;
                mA &= mA - 1;
                const int oaB = (mB ? k * 64 + __builtin_ctzll(mB) : FWD_BATCH) * 80;
                mB &= mB - 1;
                const int obB = (mB ? k * 64 + __builtin_ctzll(mB) : FWD_BATCH) * 80;
                mB &= mB - 1;
                const int oa = lane < 32 ? oaA : oaB, ob = lane < 32 ? obA : obB;
                const float4* ra = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + oa);
                const float4* rb = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + ob);
                const float4 a0 = ra[0], a1 = ra[1], a2 = ra[2];
                const float4 b0 = rb[0], b1 = rb[1], b2 = rb[2];
                const float TuA[3] = {a0.x, a0.y, a0.z}, TvA[3] = {a0.w, a1.x, a1.y}, TwA[3] = {a1.z, a1.w, a2.x};
                const float TuB[3] = {b0.x, b0.y, b0.z}, TvB[3] = {b0.w, b1.x, b1.y}, TwB[3] = {b1.z, b1.w, b2.x};
                PairEval ea, eb;
                bool okA = eval_pair_flat(TuA, TvA, TwA, a2.y, a2.z, a2.w, pixx, pixy, ea);
                bool okB = eval_pair_flat(TuB, TvB, TwB, b2.y, b2.z, b2.w, pixx, pixy, eb);
                // (no __any() around the masked blocks: the compiler's own "skip if no lane is left" -- s_and_saveexec,
                // s_cbranch_execz -- does that on the scalar unit; hip's __any(int) widened the predicate through a VGPR first,
                // v_cndmask 0 / 1 + v_cmp_ne, four VALU instructions of the expensive kind per trip for nothing)
                okA = okA && !done;
                if (okA) {
                    float4 q3 = ra[3], q4 = ra[4];
#if SURFEL_WIDE_RECORD_READS
                    asm volatile("" : "+v"(q3.w), "+v"(q4.w));   // (ds_read_b128, 4 LDS cycles, instead of ds_read_b96, 8: see blend_bwd_kernel)
#endif
                    const float nrm[3] = {q3.x, q3.y, q3.z}, rgb[3] = {q4.x, q4.y, q4.z};
                    if (!fwd_accumulate<MODE>(s, ea, nrm, rgb, (uint32_t)(key_base + oa))) done = sat_local = true;
                }
                okB = okB && !done;
                if (okB) {
                    float4 q3 = rb[3], q4 = rb[4];
#if SURFEL_WIDE_RECORD_READS
                    asm volatile("" : "+v"(q3.w), "+v"(q4.w));
#endif
                    const float nrm[3] = {q3.x, q3.y, q3.z}, rgb[3] = {q4.x, q4.y, q4.z};
                    if (!fwd_accumulate<MODE>(s, eb, nrm, rgb, (uint32_t)(key_base + ob))) done = sat_local = true;
                }
            }
        }
    }
    s.last_contributor /= 80u;   // (keys -> list positions, see the walk)
    s.median_contributor /= 80u;
    if (SPLIT && wk.seg >= 0) {
        // partial results of this segment; blend_combine_kernel adds the segments up in list order
        float* d = seg_data + (size_t)wk.slot * SEG_FLOATS * 256 + threadIdx.x;
        for (int ch = 0; ch < 3; ch++) d[(SG_C + ch) * 256] = s.C[ch];
        d[SG_TEND * 256] = dead ? -1.0f : s.T;
        d[SG_LAST * 256] = __uint_as_float(s.last_contributor);
        if (SPEC_OK && spec) d[SG_TSEG * 256] = sat_local ? 0.f : s.T;  // the segment's own product (0: must not be trusted)
        if (MODE != BLEND_LITE) {  // (LITE: nobody reads the other fields -- blend_combine_kernel, blend_bwd_kernel of that mode)
            for (int ch = 0; ch < 3; ch++) d[(SG_N + ch) * 256] = s.N[ch];
            d[SG_D * 256] = s.D;
        }
        if (MODE == BLEND_FULL) {
            d[SG_M1 * 256] = s.dist1;
            d[SG_M2 * 256] = s.dist2;
            d[SG_DIST * 256] = s.distortion;
            d[SG_MED_D * 256] = s.median_depth;
            d[SG_MED_W * 256] = s.median_weight;
            d[SG_MED_C * 256] = __uint_as_float(s.median_contributor);
        }
        return;
    }
    if (!SPLIT && rec_first != SEG_NONE) {  // the final sums, in the slot of the tile's last segment (which has nothing behind it)
        const uint32_t nseg = (r1 - r0 + (uint32_t)rec_len - 1u) / (uint32_t)rec_len;
        // (an early end -- every pixel saturated -- leaves the prefix records [rec_stop, nseg - 1) unwritten; nobody reads them:
        // no pixel's walk enters from behind them, and the distortion sums stop at rec_stop)
        seg_data[((size_t)(rec_first + nseg - 1u) * REC_REC_FLOATS + RS_STOP) * 256 + threadIdx.x] = __uint_as_float(rec_stop);
        store_record<MODE>(seg_data, rec_first + nseg - 1u, s, s.median_weight);
        // the totals the images get: the finished segments' sums in list order, then the last one (this thread's own
        // stores, read back: store_record has zeroed the running sums)
        for (uint32_t k = 0; k <= rec_stop; k++) {
            const uint32_t q = k < rec_stop ? k : nseg - 1u;
            const float* e = seg_data + (size_t)(rec_first + q) * REC_REC_FLOATS * 256 + threadIdx.x;
            for (int ch = 0; ch < 3; ch++) s.C[ch] += e[(RS_C + ch) * 256];
            if (MODE != BLEND_LITE) {
                for (int ch = 0; ch < 3; ch++) s.N[ch] += e[(RS_N + ch) * 256];
                s.D += e[RS_D * 256];
            }
        }
    }
    if (inside)
        write_pixel(s, plane, frame_base + (size_t)py * W + px, bg, img.final_T, img.n_contrib, out_color, out_others);
    report_depth(depth_used, s.last_contributor);
    report_min_T(hdr, s.T, inside);
    if (!SPLIT && rec_len && threadIdx.x == 0) {
        // Recorded segments: how many FULL segments of this tile the walk reached (nothing behind them holds a contributor:
        // the backward gives them no workgroup -- launch_bwd_prepare), by schedule position
        uint32_t live = 0;
        if (rec_first != SEG_NONE) {
            const uint32_t nseg = (r1 - r0 + (uint32_t)rec_len - 1u) / (uint32_t)rec_len;
            live = rec_stop + 1u < nseg - 1u ? rec_stop + 1u : nseg - 1u;
        }
        img.live_count[pos] = live;
    }
}

// Pass 3: adds the segments of a split tile up in list order.  The colour / depth / normal / moment
// partials were accumulated with the exact transmittance, so they simply add; the distortion needs
// the cross terms between a segment and the moments of everything in front of it:
//   sum_i w_i (m_i^2 A_i + M2_i - 2 m_i M1_i),  A_i = A_p + a_i, M1_i = M1_p + m1_i, M2_i = M2_p + m2_i
// where _p is the state at the segment start and a_i, m1_i, m2_i run inside the segment.  Pass 2 used
// the global accumulated alpha (1 - T) and segment-local moments, which leaves
//   M2_p * (sum w_i) - 2 M1_p * (sum w_i m_i)      with sum w_i = T_start - T_end.
// BLEND_LITE (colour + alpha plane only): the segments hold colour, end transmittance and last contributor, nothing else;
// BLEND_GEOM: depth and normal sums as well.
// PHASE (round 5; spec only): the pass as THREE launches, so that the repair walks of a tile -- one per segment some pixel
// of the tile saturates in -- run on workgroups of their own instead of one after the other inside the tile's workgroup
// (on the dense Stage-3 ball every covered pixel saturates, in two to four different segments per tile: the serial repairs
// were 150-310 us of a frame).  PHASE 1 ("scan") runs the first loop below, leaves what it found per
// pixel -- the segment to repair, the exact transmittance in front of it, the sums so far -- in fields of the tile's first
// three segment slots that the colour / planes-0-4 instances do not use (SG_M1 .. SG_MED_W: every split tile has at least
// three segments), and flags the segments that need a walk (SG_MED_C of the slot's pixel 0); blend_repair_kernel takes one
// flagged (tile, segment) per workgroup; PHASE 2 ("finalise") adds the repaired segment's sums and does everything behind
// the repair.  Same operations on the same operands in the same order as PHASE 0, which remains what runs without `spec`
// and under VIDU4D_DEBUG_SERIAL_REPAIR.
template <int MODE, int PHASE = 0>
__global__ __launch_bounds__(256) void blend_combine_kernel(int W, int H, int grid_x, int grid_y, Header* hdr,
                                                           ImageState img, const uint32_t* __restrict__ point_list,
                                                           const float* __restrict__ rec, int64_t capacity, int max_seg,
                                                           const float* __restrict__ bg,
                                                           float* __restrict__ seg_data,
                                                           float* __restrict__ out_color,
                                                           float* __restrict__ out_others, uint32_t* depth_used, int spec,
                                                           int flags)
{
    constexpr bool SPEC_OK = MODE != BLEND_FULL;
    constexpr bool GEOM = MODE == BLEND_GEOM;
    if ((int64_t)hdr->num_rendered > capacity || blockIdx.x >= hdr->num_split_pos) return;
    const int tile = (int)img.tile_order[blockIdx.x];
    const uint32_t first = img.seg_first[tile];
    if (first == SEG_NONE) return;
    const int nseg_all = (int)((img.ranges[2 * tile + 1] - img.ranges[2 * tile] + SEG_LEN - 1) / SEG_LEN);
    const int nseg = nseg_all < max_seg ? nseg_all : max_seg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TileCoord tc;
    tc.tile = tile;
    tc.tx = tile % grid_x;
    tc.ty = tile / grid_x;
    size_t plane;
    const size_t frame_base = frame_of_tile(tc, W, H, grid_y, plane);
    const int px = tc.tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = tc.ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    FwdPixel s;
    float T_raw = 1.0f;
    // spec: the segment in which this pixel saturates (or comes within 0.1 % of it), and what the segments before it leave
    int repair_q = -1;
    float repair_T = 1.0f;
    static_assert(PHASE == 0 || MODE != BLEND_FULL, "the three-launch form is the speculated (colour / planes 0-4) instances'");
    auto note = [&](int seg, int field) -> float& {   // per-pixel scratch of the three-launch form (see above)
        return seg_data[((size_t)(first + (uint32_t)seg) * SEG_FLOATS + field) * 256 + threadIdx.x];
    };
    for (int q = 0; PHASE != 2 && q < nseg; q++) {
        float* d = seg_data + (size_t)(first + q) * SEG_FLOATS * 256 + threadIdx.x;
        const float T_start = T_raw;
        if (SPEC_OK && spec) {
            // the segment was blended from T = 1: scale by what its predecessors leave.  A pixel that stays clear of the
            // saturation threshold to the segment's end never met it inside (the running product only decreases); one that
            // does not (or saturated inside a segment even from T = 1: product stored as 0) has this segment blended again
            // below, in order, from the exact start transmittance -- and everything behind it is dead.
            if (repair_q >= 0) {
                d[SG_TEND * 256] = -1.0f;
                continue;
            }
            const float T_end = T_start * d[SG_TSEG * 256];
            if (!(T_end >= T_EPS * 1.001f)) {
                repair_q = q;
                repair_T = T_start;
                continue;
            }
            for (int ch = 0; ch < 3; ch++) s.C[ch] = fmaf(T_start, d[(SG_C + ch) * 256], s.C[ch]);
            if (GEOM) {
                for (int ch = 0; ch < 3; ch++) s.N[ch] = fmaf(T_start, d[(SG_N + ch) * 256], s.N[ch]);
                s.D = fmaf(T_start, d[SG_D * 256], s.D);
            }
            d[SG_TEND * 256] = T_end;
            T_raw = s.T = T_end;
            const uint32_t last = __float_as_uint(d[SG_LAST * 256]);
            if (last) s.last_contributor = last;
            continue;
        }
        const float T_end = d[SG_TEND * 256];
        T_raw = T_raw * d[SG_TSEG * 256];
        if (T_end < 0.f) continue;  // saturated before this segment
        for (int ch = 0; ch < 3; ch++) s.C[ch] += d[(SG_C + ch) * 256];
        s.T = T_end;
        const uint32_t last = __float_as_uint(d[SG_LAST * 256]);
        if (last) s.last_contributor = last;
        if (MODE == BLEND_LITE) continue;
        for (int ch = 0; ch < 3; ch++) s.N[ch] += d[(SG_N + ch) * 256];
        s.D += d[SG_D * 256];
        if (GEOM) continue;
        const float m1 = d[SG_M1 * 256], m2 = d[SG_M2 * 256];
        s.distortion += d[SG_DIST * 256] + (s.dist2 * (T_start - T_end) - 2.0f * s.dist1 * m1);
        s.dist1 += m1;
        s.dist2 += m2;
        const uint32_t med = __float_as_uint(d[SG_MED_C * 256]);
        if (med) {
            s.median_contributor = med;
            s.median_depth = d[SG_MED_D * 256];
            s.median_weight = d[SG_MED_W * 256];
        }
    }
    if constexpr (PHASE == 1) {
        note(0, SG_M1) = __int_as_float(repair_q);
        note(0, SG_M2) = repair_T;
        note(0, SG_DIST) = T_raw;
        note(0, SG_MED_D) = s.T;
        note(0, SG_MED_W) = __uint_as_float(s.last_contributor);
        note(1, SG_M1) = s.C[0];
        note(1, SG_M2) = s.C[1];
        note(1, SG_DIST) = s.C[2];
        note(1, SG_MED_D) = s.D;
        note(1, SG_MED_W) = s.N[0];
        note(2, SG_M1) = s.N[1];
        note(2, SG_M2) = s.N[2];
        // which segments of this tile some pixel must be blended again in
        for (int q = threadIdx.x; q < nseg_all; q += 256)
            seg_data[((size_t)(first + (uint32_t)q) * SEG_FLOATS + SG_MED_C) * 256] = __uint_as_float(0u);
        __syncthreads();
        if (px < W && py < H && repair_q >= 0)
            seg_data[((size_t)(first + (uint32_t)repair_q) * SEG_FLOATS + SG_MED_C) * 256] = __uint_as_float(1u);
        return;
    }
    if constexpr (PHASE == 2) {
        repair_q = __float_as_int(note(0, SG_M1));
        T_raw = note(0, SG_DIST);
        s.T = note(0, SG_MED_D);
        s.last_contributor = __float_as_uint(note(0, SG_MED_W));
        s.C[0] = note(1, SG_M1);
        s.C[1] = note(1, SG_M2);
        s.C[2] = note(1, SG_DIST);
        s.D = note(1, SG_MED_D);
        s.N[0] = note(1, SG_MED_W);
        s.N[1] = note(2, SG_M1);
        s.N[2] = note(2, SG_M2);
        if (px < W && py < H && repair_q >= 0 && repair_q < nseg) {   // what blend_repair_kernel left in the segment's slot
            const float* d = seg_data + (size_t)(first + (uint32_t)repair_q) * SEG_FLOATS * 256 + threadIdx.x;
            for (int ch = 0; ch < 3; ch++) s.C[ch] += d[(SG_C + ch) * 256];
            if (GEOM) {
                for (int ch = 0; ch < 3; ch++) s.N[ch] += d[(SG_N + ch) * 256];
                s.D += d[SG_D * 256];
            }
            s.T = d[SG_TEND * 256];
            const uint32_t last = __float_as_uint(d[SG_LAST * 256]);
            if (last) s.last_contributor = last;
            T_raw = 0.f;  // (saturated: not cut short by a segment limit)
        }
    }
    if constexpr (SPEC_OK && PHASE == 0) if (spec) {
        // ---- the saturating segments, blended again for the pixels that saturate in them: the loop of blend_fwd_kernel,
        // restricted to the segment and to those pixels, from the exact transmittance (the product of the predecessors'
        // products in list order: what blend_seg_T_kernel + blend_fwd_kernel use for their start)
        __shared__ float4 s_rec[FWD_BATCH * 5];
        __shared__ unsigned long long s_mask[4][4];
        const bool inside = px < W && py < H;
        const float pixx = (float)px + 0.5f, pixy = (float)py + 0.5f;
        const uint32_t r0 = img.ranges[2 * tile], r1 = img.ranges[2 * tile + 1];
        for (int q = 0; q < nseg; q++) {
            const bool mine = inside && repair_q == q;
            if (!__syncthreads_or(mine)) continue;
            FwdPixel t;  // the segment's partial sums, from the exact start
            t.T = repair_T;
            bool done = !mine;
            const int begin = q * SEG_LEN;
            int todo = min((int)(r1 - r0) - begin, SEG_LEN);
            for (int base = begin; todo > 0; base += FWD_BATCH, todo -= FWD_BATCH) {
                if (__syncthreads_count(done) == 256) break;
                const bool have = (int)threadIdx.x < todo;
                FootprintTest foot = no_footprint();
                if (have) foot = stage_record(s_rec, threadIdx.x, rec, point_list[r0 + base + threadIdx.x]);
                publish_cull_masks(s_mask, have, foot, tc.tx * TILE, tc.ty * TILE, wave, lane, flags & FLAG_NO_CULL);
                __syncthreads();
                if (__all(done)) continue;
#pragma unroll 1
                for (int k = 0; k < 4; k++) {
                    unsigned long long m = uniform_u64(s_mask[wave][k]);
                    while (m) {
                        const int j = k * 64 + __builtin_ctzll(m);
                        m &= m - 1;
                        const float4 a0 = s_rec[j * 5 + 0], a1 = s_rec[j * 5 + 1], a2 = s_rec[j * 5 + 2];
                        const float Tu[3] = {a0.x, a0.y, a0.z}, Tv[3] = {a0.w, a1.x, a1.y}, Tw[3] = {a1.z, a1.w, a2.x};
                        PairEval e;
                        const bool ok = eval_pair_flat(Tu, Tv, Tw, a2.y, a2.z, a2.w, pixx, pixy, e) && !done;
                        if (!__any(ok)) continue;
                        if (ok) {
                            const float4 q3 = s_rec[j * 5 + 3], q4 = s_rec[j * 5 + 4];
                            const float nrm[3] = {q3.x, q3.y, q3.z}, rgb[3] = {q4.x, q4.y, q4.z};
                            if (!fwd_accumulate<MODE>(t, e, nrm, rgb, (uint32_t)(base + j + 1))) done = true;
                        }
                    }
                }
            }
            __syncthreads();  // (s_rec / s_mask are staged again for the next segment)
            if (mine) {
                float* d = seg_data + (size_t)(first + q) * SEG_FLOATS * 256 + threadIdx.x;
                for (int ch = 0; ch < 3; ch++) {
                    d[(SG_C + ch) * 256] = t.C[ch];  // (absolute, unlike the speculated segments': SG_TSEG says so)
                    s.C[ch] += t.C[ch];
                }
                if (GEOM) {
                    for (int ch = 0; ch < 3; ch++) {
                        d[(SG_N + ch) * 256] = t.N[ch];
                        s.N[ch] += t.N[ch];
                    }
                    d[SG_D * 256] = t.D;
                    s.D += t.D;
                }
                d[SG_TSEG * 256] = -2.0f;
                d[SG_TEND * 256] = t.T;
                d[SG_LAST * 256] = __uint_as_float(t.last_contributor);
                s.T = t.T;
                if (t.last_contributor) s.last_contributor = t.last_contributor;
                T_raw = 0.f;  // (saturated: not cut short by a segment limit)
            }
        }
    }
    if (px < W && py < H)
        write_pixel(s, plane, frame_base + (size_t)py * W + px, bg, img.final_T, img.n_contrib, out_color, out_others);

    report_depth(depth_used, s.last_contributor);
    // the caller's segment limit cut this tile short and this pixel had not saturated yet: its values are
    // incomplete -- tell the caller (who blends the frame again without the limit)
    if (nseg < nseg_all && px < W && py < H && T_raw >= T_EPS) hdr->truncated = 1;
    report_min_T(hdr, s.T, px < W && py < H);

    // For the segment-parallel backward: replace each segment's partials by the sums over the segments
    // BEHIND it (what the back-to-front recurrences of the backward have accumulated when they reach
    // the segment's last entry), and the median weight if the median sample lies behind it.
    float sufC[3] = {0, 0, 0}, sufN[3] = {0, 0, 0}, sufD = 0, sufM1 = 0, sufM2 = 0, med_w = 0;
    for (int q = nseg - 1; q >= 0; q--) {
        float* d = seg_data + (size_t)(first + q) * SEG_FLOATS * 256 + threadIdx.x;
        if (d[SG_TEND * 256] < 0.f) continue;
        if (SPEC_OK) {
            float c0 = d[(SG_C + 0) * 256], c1 = d[(SG_C + 1) * 256], c2 = d[(SG_C + 2) * 256];
            float n0 = 0, n1 = 0, n2 = 0, dd = 0;
            if (GEOM) n0 = d[(SG_N + 0) * 256], n1 = d[(SG_N + 1) * 256], n2 = d[(SG_N + 2) * 256], dd = d[SG_D * 256];
            if (spec && d[SG_TSEG * 256] != -2.0f) {  // (the stored sums are relative to the segment's start transmittance)
                const float T_start = q > 0 ? seg_data[((size_t)(first + q - 1) * SEG_FLOATS + SG_TEND) * 256 + threadIdx.x] : 1.0f;
                c0 *= T_start, c1 *= T_start, c2 *= T_start;
                n0 *= T_start, n1 *= T_start, n2 *= T_start, dd *= T_start;
            }
            d[(SG_C + 0) * 256] = sufC[0];
            d[(SG_C + 1) * 256] = sufC[1];
            d[(SG_C + 2) * 256] = sufC[2];
            sufC[0] += c0, sufC[1] += c1, sufC[2] += c2;
            if (GEOM) {
                d[(SG_N + 0) * 256] = sufN[0];
                d[(SG_N + 1) * 256] = sufN[1];
                d[(SG_N + 2) * 256] = sufN[2];
                d[SG_D * 256] = sufD;
                sufN[0] += n0, sufN[1] += n1, sufN[2] += n2;
                sufD += dd;
            }
            continue;
        }
        const float c0 = d[(SG_C + 0) * 256], c1 = d[(SG_C + 1) * 256], c2 = d[(SG_C + 2) * 256];
        const float n0 = d[(SG_N + 0) * 256], n1 = d[(SG_N + 1) * 256], n2 = d[(SG_N + 2) * 256];
        const float dd = d[SG_D * 256], m1 = d[SG_M1 * 256], m2 = d[SG_M2 * 256];
        const bool has_med = __float_as_uint(d[SG_MED_C * 256]) != 0 &&
                             __float_as_uint(d[SG_MED_C * 256]) == s.median_contributor;
        const float mw = d[SG_MED_W * 256];
        d[(SG_C + 0) * 256] = sufC[0];
        d[(SG_C + 1) * 256] = sufC[1];
        d[(SG_C + 2) * 256] = sufC[2];
        d[(SG_N + 0) * 256] = sufN[0];
        d[(SG_N + 1) * 256] = sufN[1];
        d[(SG_N + 2) * 256] = sufN[2];
        d[SG_D * 256] = sufD;
        d[SG_M1 * 256] = sufM1;
        d[SG_M2 * 256] = sufM2;
        d[SG_MED_W * 256] = med_w;
        sufC[0] += c0, sufC[1] += c1, sufC[2] += c2;
        sufN[0] += n0, sufN[1] += n1, sufN[2] += n2;
        sufD += dd, sufM1 += m1, sufM2 += m2;
        if (has_med) med_w = mw;
    }
}

// The repair walk of ONE (tile, segment) -- see blend_combine_kernel's PHASE note: workgroup b takes segment slot b (as
// blend_fwd_kernel<true> numbers them), leaves at once unless PHASE 1 flagged the slot, and blends the segment again for
// the pixels of the tile that saturate in it, from the exact transmittance PHASE 1 left per pixel; the sums go into the
// segment's slot as absolute values (SG_TSEG = -2 says so), where PHASE 2 and the backward read them.
template <int MODE>
__global__ __launch_bounds__(256) void blend_repair_kernel(int W, int H, int grid_x, int grid_y, Header* hdr, ImageState img,
                                                          const uint32_t* __restrict__ point_list,
                                                          const float* __restrict__ rec, int64_t capacity, int max_seg,
                                                          float* __restrict__ seg_data, int flags)
{
    static_assert(MODE != BLEND_FULL, "speculated instances only");
    constexpr bool GEOM = MODE == BLEND_GEOM;
    __shared__ float4 s_rec[FWD_BATCH * 5];
    __shared__ unsigned long long s_mask[4][4];
    if ((int64_t)hdr->num_rendered > capacity || blockIdx.x >= hdr->num_segments) return;
    const WorkItem wk = find_work<true>(hdr, img, grid_x, grid_y, false);
    if (wk.seg >= max_seg) return;
    if (__float_as_uint(seg_data[((size_t)wk.slot * SEG_FLOATS + SG_MED_C) * 256]) == 0u) return;   // (wave-uniform)
    TileCoord tc = wk.tc;
    const int tile = tc.tile;
    size_t plane_unused;
    (void)frame_of_tile(tc, W, H, grid_y, plane_unused);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tc.tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = tc.ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px + 0.5f, pixy = (float)py + 0.5f;
    const uint32_t first = wk.slot - (uint32_t)wk.seg;
    const int q = wk.seg;
    const int repair_q = __float_as_int(seg_data[((size_t)first * SEG_FLOATS + SG_M1) * 256 + threadIdx.x]);
    const float repair_T = seg_data[((size_t)first * SEG_FLOATS + SG_M2) * 256 + threadIdx.x];
    const bool mine = inside && repair_q == q;
    const uint32_t r0 = img.ranges[2 * tile], r1 = img.ranges[2 * tile + 1];
    FwdPixel t;  // the segment's partial sums, from the exact start
    t.T = repair_T;
    bool done = !mine;
    const int begin = q * SEG_LEN;
    int todo = min((int)(r1 - r0) - begin, SEG_LEN);
    for (int base = begin; todo > 0; base += FWD_BATCH, todo -= FWD_BATCH) {
        if (__syncthreads_count(done) == 256) break;
        const bool have = (int)threadIdx.x < todo;
        FootprintTest foot = no_footprint();
        if (have) foot = stage_record(s_rec, threadIdx.x, rec, point_list[r0 + base + threadIdx.x]);
        publish_cull_masks(s_mask, have, foot, tc.tx * TILE, tc.ty * TILE, wave, lane, flags & FLAG_NO_CULL);
        __syncthreads();
        if (__all(done)) continue;
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
            unsigned long long m = uniform_u64(s_mask[wave][k]);
            while (m) {
                const int j = k * 64 + __builtin_ctzll(m);
                m &= m - 1;
                const float4 a0 = s_rec[j * 5 + 0], a1 = s_rec[j * 5 + 1], a2 = s_rec[j * 5 + 2];
                const float Tu[3] = {a0.x, a0.y, a0.z}, Tv[3] = {a0.w, a1.x, a1.y}, Tw[3] = {a1.z, a1.w, a2.x};
                PairEval e;
                const bool ok = eval_pair_flat(Tu, Tv, Tw, a2.y, a2.z, a2.w, pixx, pixy, e) && !done;
                if (!__any(ok)) continue;
                if (ok) {
                    const float4 q3 = s_rec[j * 5 + 3], q4 = s_rec[j * 5 + 4];
                    const float nrm[3] = {q3.x, q3.y, q3.z}, rgb[3] = {q4.x, q4.y, q4.z};
                    if (!fwd_accumulate<MODE>(t, e, nrm, rgb, (uint32_t)(base + j + 1))) done = true;
                }
            }
        }
    }
    if (mine) {
        float* d = seg_data + (size_t)wk.slot * SEG_FLOATS * 256 + threadIdx.x;
        for (int ch = 0; ch < 3; ch++) d[(SG_C + ch) * 256] = t.C[ch];  // (absolute, unlike the speculated segments')
        if (GEOM) {
            for (int ch = 0; ch < 3; ch++) d[(SG_N + ch) * 256] = t.N[ch];
            d[SG_D * 256] = t.D;
        }
        d[SG_TSEG * 256] = -2.0f;
        d[SG_TEND * 256] = t.T;
        d[SG_LAST * 256] = __uint_as_float(t.last_contributor);
    }
}

template <bool SPLIT>
static auto pick_fwd(int mode)
{
    return mode == BLEND_LITE ? &blend_fwd_kernel<SPLIT, BLEND_LITE>
           : mode == BLEND_GEOM ? &blend_fwd_kernel<SPLIT, BLEND_GEOM>
                                : &blend_fwd_kernel<SPLIT, BLEND_FULL>;
}

// (experiments: VIDU4D_FWD_PAD_LDS=<bytes> of unused dynamic LDS per workgroup of the whole-tile forward caps the workgroups
// resident per CU)
static int fwd_pad_lds()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VIDU4D_FWD_PAD_LDS");
        v = e ? atoi(e) : 0;
    }
    return v;
}

void launch_blend_fwd(const CameraParams& cam, const GeomState& g, const ImageState& img, const BinState& b,
                      int64_t capacity, bool split, int max_seg, const float* background, float* out_color,
                      float* out_others, uint32_t* depth_used, int mode, bool assume_unsaturated, int flags, bool record,
                      hipStream_t stream)
{
    const int spec = (assume_unsaturated && mode != BLEND_FULL && split && capacity > 0) ? 1 : 0;
    const int tiles = total_tiles(cam), grid_y = cam.grid_y * cam.frames;  // (stacked frames: a taller tile grid)
    const uint32_t* point_list = capacity > 0 ? b.point_list : nullptr;
    if (!split || capacity <= 0) {
        const int rec_len = (record && capacity > 0) ? REC_SEG_LEN : 0;
        // (paired workgroups for the longest tiles -- fwd_pair_walk; the device knows how many: Header::num_paired)
        const int pair_cap = ((flags >> FLAG_PAIR_SHIFT) & 15) ? std::min(tiles, PAIR_MAX) : 0;
        hipLaunchKernelGGL(pick_fwd<false>(mode), dim3(tiles + pair_cap), dim3(256), fwd_pad_lds(), stream, cam.W, cam.H, cam.grid_x, grid_y, g.hdr, img,
                           point_list, capacity, 0, g.rec, background, b.seg_data, out_color, out_others, depth_used, 0, flags,
                           rec_len);
        return;
    }
    // upper bounds; the device knows the exact counts.  The combine runs over ALL schedule positions: the
    // schedule is sorted by length CLASS, so a split tile may sit behind unsplit ones of its class and the
    // number of positions that hold split tiles is not bounded by capacity / SPLIT_MIN.
    const int segs = (int)seg_capacity(capacity);
    const int split_tiles = tiles;
    if (!spec)
        hipLaunchKernelGGL(blend_seg_T_kernel, dim3(segs), dim3(256), 0, stream, cam.W, cam.H, cam.grid_x, grid_y,
                           g.hdr, img, point_list, capacity, max_seg, g.rec, b.seg_data, flags);
    hipLaunchKernelGGL(pick_fwd<true>(mode), dim3(segs + tiles), dim3(256), 0, stream, cam.W, cam.H, cam.grid_x, grid_y, g.hdr,
                       img, point_list, capacity, max_seg, g.rec, background, b.seg_data, out_color, out_others, depth_used, spec,
                       flags, 0);
    if (spec && !(flags & FLAG_SERIAL_REPAIR)) {
        // the combine as three launches: scan the segments' records, blend the saturating (tile, segment) pairs again on
        // workgroups of their own, add up (blend_combine_kernel's PHASE note)
        auto scan = mode == BLEND_LITE ? &blend_combine_kernel<BLEND_LITE, 1> : &blend_combine_kernel<BLEND_GEOM, 1>;
        auto repair = mode == BLEND_LITE ? &blend_repair_kernel<BLEND_LITE> : &blend_repair_kernel<BLEND_GEOM>;
        auto fin = mode == BLEND_LITE ? &blend_combine_kernel<BLEND_LITE, 2> : &blend_combine_kernel<BLEND_GEOM, 2>;
        hipLaunchKernelGGL(scan, dim3(split_tiles), dim3(256), 0, stream, cam.W, cam.H, cam.grid_x, grid_y, g.hdr, img, point_list,
                           g.rec, capacity, max_seg, background, b.seg_data, out_color, out_others, depth_used, spec, flags);
        hipLaunchKernelGGL(repair, dim3(segs), dim3(256), 0, stream, cam.W, cam.H, cam.grid_x, grid_y, g.hdr, img, point_list,
                           g.rec, capacity, max_seg, b.seg_data, flags);
        hipLaunchKernelGGL(fin, dim3(split_tiles), dim3(256), 0, stream, cam.W, cam.H, cam.grid_x, grid_y, g.hdr, img, point_list,
                           g.rec, capacity, max_seg, background, b.seg_data, out_color, out_others, depth_used, spec, flags);
        return;
    }
    auto combine = mode == BLEND_LITE ? &blend_combine_kernel<BLEND_LITE>
                   : mode == BLEND_GEOM ? &blend_combine_kernel<BLEND_GEOM>
                                        : &blend_combine_kernel<BLEND_FULL>;
    hipLaunchKernelGGL(combine, dim3(split_tiles), dim3(256), 0, stream, cam.W, cam.H, cam.grid_x, grid_y, g.hdr, img,
                       point_list, g.rec, capacity, max_seg, background, b.seg_data, out_color, out_others, depth_used, spec,
                       flags);
}

// ---------------------------------------------------------------------------------------------
// DPP lane exchanges (GFX9 encodings): all are involutions inside a row of 16 lanes.
constexpr int DPP_QUAD_XOR1 = 0xB1;    // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;    // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_MIRROR = 0x140;  // lane i <-> 15 - i
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i <-> 7 - i inside each 8

template <int CTRL>
__device__ __forceinline__ float dpp_xchg(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// One halving step of the reduce-scatter: `hi` lanes keep v[N/2..N), the others v[0..N/2); every
// kept value receives the partner lane's copy of it.
template <int CTRL, int N>
__device__ __forceinline__ void halve(float (&v)[16], bool hi)
{
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
        const float keep = hi ? v[i + N / 2] : v[i];
        const float send = hi ? v[i] : v[i + N / 2];
        v[i] = keep + dpp_xchg<CTRL>(send);
    }
}

// Reduce-scatter of 16 per-lane values over each HALF of the wave (32 lanes = two 16-lane rows = the pixels of one 8x4
// block) with the gfx950 lane-swap instruction: v_permlane16_swap exchanges the odd rows of one register with the even rows
// of another, so "swap, add" on the register pairs (k, k + 8) folds the two rows AND halves the values per lane without a
// select (even rows: values 0-7, odd rows: 8-15).  Three select-and-add DPP steps and a quad add finish the columns.
// Lane L then holds its half's total of value 8 ((L >> 4) & 1) + ((L >> 1) & 7) -- both lanes of a pair the same one.
// (Rounds 2-3 reduced over the whole wave, with a v_permlane32_swap step in front: same cost per trip, one entry per trip.)
__device__ __forceinline__ void swap_add16(float& a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Round 5: the two halving steps whose "upper" lanes are whole DPP banks (lane & 8: banks 2-3 of a row; lane & 4: banks 1
// and 3) are written as bank-masked DPP adds -- "the lower lanes add their partner's copy of v[i] into v[i]; the upper lanes
// put v[i + N/2] plus the partner's copy of it into v[i]" -- two instructions per kept value and NO select (the select
// version: two v_cndmask on an SGPR-pair mask, 4.7 issue cycles each, and one DPP add).  Same additions on the same operands
// (a + b against b + a): bit-identical.  Hand-scheduled: a DPP source must not have been written by the two instructions in
// front of it, and the compiler's hazard recogniser does not look into inline assembly -- hence the s_nop at both ends.
// SURFEL_ABLATE (timing experiments only, results are wrong): bit 0 replaces the reduce-scatter ladder by a per-lane sum,
// bit 1 drops the LDS atomics, bit 2 the global flush atomics.
#ifndef SURFEL_ABLATE
#define SURFEL_ABLATE 0
#endif
#ifndef SURFEL_MASKED_DPP_LADDER
#define SURFEL_MASKED_DPP_LADDER 1
#endif
__device__ __forceinline__ float half_reduce_scatter16(float (&v)[16], int lane)
{
#pragma unroll
    for (int k = 0; k < 8; k++) swap_add16(v[k], v[k + 8]);
#if SURFEL_MASKED_DPP_LADDER
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %6, %6 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %7, %7 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])
        : "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
#else
    halve<DPP_ROW_MIRROR, 8>(v, (lane & 8) != 0);
    halve<DPP_ROW_HALF_MIRROR, 4>(v, (lane & 4) != 0);
#endif
    halve<DPP_QUAD_XOR2, 2>(v, (lane & 2) != 0);
    float t = v[0];
    t += dpp_xchg<DPP_QUAD_XOR1>(t);
    return t;
}

// Same for 2 values: lane L holds the row total of component ((L >> 3) & 1).
__device__ __forceinline__ float row_reduce_scatter2(float a, float b, int lane)
{
    const bool hi = (lane & 8) != 0;
    float v = (hi ? b : a) + dpp_xchg<DPP_ROW_MIRROR>(hi ? a : b);
    v += dpp_xchg<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_xchg<DPP_QUAD_XOR2>(v);
    v += dpp_xchg<DPP_QUAD_XOR1>(v);
    return v;
}

// SPLIT: the tiles the segment table lists are walked segment-parallel.  The back-to-front recurrences of a segment start
// from what the segments behind it add up to -- stored by blend_combine_kernel after a segment-parallel forward
// (Header::split_used == 1), or final - prefix of the records a whole-tile forward left (== 2: recorded segments,
// surfel_state.h); everything else is the single-workgroup loop restricted to the segment.  The segment length is the one
// the table was built with (Header::seg_len).
// MODE: BLEND_LITE reads only dL/dcolour and dL/d(alpha plane), BLEND_GEOM those and the planes 0, 2-4 (the caller promised
// zeros elsewhere, aux_planes); the forward that filled the state may itself have run in that mode (no distortion moments,
// no median contributor: never read here).
// Register budget of the full instances: 80 VGPRs = 6 waves per SIMD (the allocator takes 88 = 5 waves when left alone;
// no spills at 80, 10-13 spilled registers at 72).  Round 3, stacked launch at the headline size: 509 -> 497 us; 4 waves
// per SIMD measure the same as 5 -- the kernel is bound by VALU issue, not by latency, the sixth wave only fills bubbles.
// The LITE instances need 64-66 registers and are left to the allocator (7 waves, the LDS limit).
#ifndef SURFEL_BWD_WAVES_PER_EU
#define SURFEL_BWD_WAVES_PER_EU 6
#endif
template <bool SPLIT, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE == BLEND_LITE ? 1 : SURFEL_BWD_WAVES_PER_EU, MODE == BLEND_LITE ? 8 : SURFEL_BWD_WAVES_PER_EU)))
void blend_bwd_kernel(int W, int H, int grid_x, int grid_y, const Header* hdr,
                                                       ImageState img, const uint32_t* __restrict__ point_list,
                                                       const float* __restrict__ rec, const float* __restrict__ bg,
                                                       const float* __restrict__ seg_data, int max_seg,
                                                       const float* __restrict__ dL_dcolor,
                                                       const float* __restrict__ dL_dothers, float* __restrict__ acc, int flags
#ifdef SURFEL_BWD_TRACE   // (tools/bwd_trace.py: a variant build that timestamps every workgroup; never in the product)
                                                       , unsigned long long* __restrict__ trace
#endif
                                                       )
{
    constexpr bool LITE = MODE == BLEND_LITE, FULL = MODE == BLEND_FULL;
#ifdef SURFEL_BWD_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    unsigned long long trace_t1 = trace_t0;
    struct TraceEnd {
        unsigned long long* p; unsigned long long t0; const unsigned long long* t1; int entries;
        __device__ ~TraceEnd() {
            if (p && threadIdx.x == 0) {
                p[4 * blockIdx.x + 0] = t0;
                p[4 * blockIdx.x + 1] = *t1;
                p[4 * blockIdx.x + 2] = wall_clock64();
                p[4 * blockIdx.x + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(0xF804) |
                                        ((unsigned long long)(__builtin_amdgcn_s_getreg(0xF814) & 0xf) << 32) |
                                        ((unsigned long long)(entries & 0xffff) << 40);
            }
        }
    } trace_end{trace, trace_t0, &trace_t1, 0};
#endif
    const uint32_t* __restrict__ ranges = img.ranges;
    const float* __restrict__ final_T = img.final_T;
    const uint32_t* __restrict__ n_contrib = img.n_contrib;
    __shared__ float4 s_rec[(BWD_BATCH + 1) * 5];  // (+ an all-zero record: what an idle half of a wave evaluates)
    __shared__ float s_acc[BWD_BATCH * ACC_FLOATS];
    __shared__ uint32_t s_id[BWD_BATCH];
    __shared__ unsigned long long s_mask8[8][BWD_BATCH / 64];
    __shared__ uint32_t s_max;
    // (a forward that left no segment state: every tile is then walked whole)
    const uint32_t split_used = SPLIT ? hdr->split_used : 0u;
    const WorkItem wk = (SPLIT && split_used == 2u) ? find_work_recorded(hdr, img, grid_x, grid_y, seg_data)
                        : (SPLIT && split_used == 1u && !(flags & FLAG_POSITION_ORDER)) ? find_work_split_ordered(hdr, img, grid_x, grid_y)
                                                    : find_work<SPLIT>(hdr, img, grid_x, grid_y, SPLIT && !split_used);
    if (!wk.valid || (SPLIT && wk.seg >= max_seg)) return;
    TileCoord tc = wk.tc;
    size_t HW;  // (distance between planes: frames * H * W)
    const size_t frame_base = frame_of_tile(tc, W, H, grid_y, HW);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tc.tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = tc.ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px + 0.5f, pixy = (float)py + 0.5f;
    const uint32_t r0 = ranges[2 * tc.tile];
    const size_t pid = frame_base + (size_t)py * W + px;
    const int seg_len = SPLIT ? (int)hdr->seg_len : 1;   // (!SPLIT: unused; 1 keeps the divisions below defined)
    const int seg_begin = (SPLIT && wk.seg >= 0) ? wk.seg * seg_len : 0;

    BwdPixel s;
    s.last_contributor = 0;
    s.median_contributor = 0;
    s.T_final = 0.f;
    s.final_D = s.final_D2 = 0.f;
    s.dL_ddepth = s.dL_daccum = s.dL_dreg = s.dL_dmedian_depth = s.dL_dmax_dweight = 0.f;
    for (int c = 0; c < 3; c++) s.dL_dpixel[c] = s.dL_dnormal2D[c] = 0.f;
    if (inside) {
        s.T_final = final_T[pid];
        s.last_contributor = n_contrib[pid];
        for (int c = 0; c < 3; c++) s.dL_dpixel[c] = dL_dcolor[c * HW + pid];
        s.dL_daccum = dL_dothers[pid + HW];
        if (!LITE) {
            s.dL_ddepth = dL_dothers[pid];
            for (int c = 0; c < 3; c++) s.dL_dnormal2D[c] = dL_dothers[pid + (2 + c) * HW];
        }
        if (FULL) {
            s.final_D = final_T[pid + HW];
            s.final_D2 = final_T[pid + 2 * HW];
            s.median_contributor = n_contrib[pid + HW];
            s.dL_dmedian_depth = dL_dothers[pid + 5 * HW];
            s.dL_dreg = dL_dothers[pid + 6 * HW];
            s.dL_dmax_dweight = dL_dothers[pid + 7 * HW];
        }
    }
    if (FULL) s.m0 = img.tile_m0[tc.tile];
    s.T = s.T_final;
    s.final_A = 1.0f - s.T_final;
    s.bg_dot_dpixel = bg[0] * s.dL_dpixel[0] + bg[1] * s.dL_dpixel[1] + bg[2] * s.dL_dpixel[2];
    if (SPLIT && wk.seg >= 0) {
        // the entries of this segment lie in [seg_begin, seg_begin + seg_len): clip the pixel's range
        const uint32_t seg_end = (uint32_t)(seg_begin + seg_len);
        if (s.last_contributor > seg_end) {
            // the walk enters from the segments behind: start from what they add up to
            float T_end, sufC[3], sufN[3] = {0.f, 0.f, 0.f}, sufD = 0.f, sufM1 = 0.f, sufM2 = 0.f, med_w = 0.f;
            if (split_used == 2u) {
                // recorded segments: what lies behind = the later segments' own sums, up to the one the forward's walk
                // stopped in (whose sums the final record, in the tile's last slot, holds) -- each summed from zero
                const uint32_t len = ranges[2 * tc.tile + 1] - r0;
                const uint32_t nseg = (len + (uint32_t)seg_len - 1u) / (uint32_t)seg_len;
                const float* d = seg_data + (size_t)wk.slot * REC_REC_FLOATS * 256 + threadIdx.x;
                const float* f = seg_data + (size_t)(wk.slot - (uint32_t)wk.seg + nseg - 1u) * REC_REC_FLOATS * 256 + threadIdx.x;
                T_end = d[RS_T * 256];
                const uint32_t stop = __builtin_amdgcn_readfirstlane(__float_as_uint(f[RS_STOP * 256]));
                for (int ch = 0; ch < 3; ch++) sufC[ch] = f[(RS_C + ch) * 256];
                if (!LITE) {
                    for (int ch = 0; ch < 3; ch++) sufN[ch] = f[(RS_N + ch) * 256];
                    sufD = f[RS_D * 256];
                }
                for (uint32_t q = (uint32_t)wk.seg + 1u; q < stop; q++) {
                    const float* e = seg_data + (size_t)(wk.slot - (uint32_t)wk.seg + q) * REC_REC_FLOATS * 256 + threadIdx.x;
                    for (int ch = 0; ch < 3; ch++) sufC[ch] += e[(RS_C + ch) * 256];
                    if (!LITE) {
                        for (int ch = 0; ch < 3; ch++) sufN[ch] += e[(RS_N + ch) * 256];
                        sufD += e[RS_D * 256];
                    }
                }
                if (FULL) {
                    // (the distortion moments: totals minus this segment's running totals -- about the tile's reference
                    // depth they are small numbers, and so is what the subtraction loses)
                    sufM1 = s.final_D - d[RS_M1 * 256];
                    sufM2 = s.final_D2 - d[RS_M2 * 256];
                    med_w = s.median_contributor > seg_end ? f[RS_T * 256] : 0.f;  // (the median sample lies behind)
                }
            } else {
                const float* d = seg_data + (size_t)wk.slot * SEG_FLOATS * 256 + threadIdx.x;
                T_end = d[SG_TEND * 256];
                for (int ch = 0; ch < 3; ch++) sufC[ch] = d[(SG_C + ch) * 256];
                if (!LITE) {
                    for (int ch = 0; ch < 3; ch++) sufN[ch] = d[(SG_N + ch) * 256];
                    sufD = d[SG_D * 256];
                }
                if (FULL) {
                    sufM1 = d[SG_M1 * 256];
                    sufM2 = d[SG_M2 * 256];
                    med_w = d[SG_MED_W * 256];
                }
            }
            const float inv = 1.0f / T_end;
            const float behind = T_end - s.T_final;  // sum of their blend weights
            for (int ch = 0; ch < 3; ch++) s.accum_rec[ch] = sufC[ch] * inv;
            s.accum_alpha_rec = behind * inv;
            if (!LITE) {
                for (int ch = 0; ch < 3; ch++) s.accum_normal_rec[ch] = sufN[ch] * inv;
                s.accum_depth_rec = sufD * inv;
            }
            // (distortion chain: sum over the samples behind of w (D2 + m^2 A - 2 m D); all moments about the tile's
            // reference depth, so the three terms are of the size of their sum)
            if (FULL)
                s.last_dL_dT = inv * (s.dL_dmax_dweight * med_w +
                                      s.dL_dreg * (s.final_D2 * behind + s.final_A * sufM2 - 2.0f * s.final_D * sufM1));
            s.T = T_end;
            s.last_contributor = seg_end;
        }
        if (s.last_contributor <= (uint32_t)seg_begin) s.last_contributor = 0;  // nothing of this pixel in here
    }

    // entries at or beyond every pixel's last contributor never matter: skip them wholesale
    // (per wave for the inner loop, per workgroup for the staging)
    if (threadIdx.x == 0) s_max = 0;
    if (threadIdx.x < 5) s_rec[BWD_BATCH * 5 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    uint32_t wave_last = s.last_contributor;
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)wave_last, d, 64);
        wave_last = o > wave_last ? o : wave_last;
    }
    const int half_last[2] = {(int)__builtin_amdgcn_readlane(wave_last, 0), (int)__builtin_amdgcn_readlane(wave_last, 32)};
    wave_last = (uint32_t)(half_last[0] > half_last[1] ? half_last[0] : half_last[1]);
    if (lane == 0) atomicMax(&s_max, wave_last);
    __syncthreads();
    const int n_used = (int)s_max;
#ifdef SURFEL_BWD_TRACE
    trace_t1 = wall_clock64();
    trace_end.entries = n_used > seg_begin ? n_used - seg_begin : 0;
#endif

    // accumulator slot handled by this lane after the row reduce-scatter
    // (after half_reduce_scatter16 the lane pair p of row r holds, for its half, value p + 8 (r & 1); the even lane adds it)
    const int c16 = ((lane >> 1) & 7) + 8 * ((lane >> 4) & 1);
    const bool adds16 = (lane & 1) == 0;
    const int slot16 = c16 < 9 ? c16 : (c16 == 9 ? A_OPAC : (c16 < 13 ? c16 + 2 : c16 + 3));
    const int slot2 = A_M2D + ((lane >> 3) & 1);
    // (this lane's two accumulator addresses of slot 0; a slot adds its byte offset)
    char* const acc16 = reinterpret_cast<char*>(s_acc + slot16);
    char* const acc2 = reinterpret_cast<char*>(s_acc + slot2);

    // one slot of the batch in LDS: an 80-byte record in s_rec, an 80-byte accumulator row in s_acc
    constexpr int SLOT_BYTES = 80;
    static_assert(ACC_FLOATS * 4 == SLOT_BYTES && sizeof(float4) * 5 == SLOT_BYTES, "record and accumulator row share the slot offset");
    for (int hi = n_used; hi > seg_begin; hi -= BWD_BATCH) {
        const int cnt = hi - seg_begin < BWD_BATCH ? hi - seg_begin : BWD_BATCH;
        // slot j of the batch <-> list entry hi - 1 - j.  This pixel's entries are those below its last contributor: slots
        // j > hi - 1 - last_contributor; its median sample is entry median_contributor - 1: slot hi - median_contributor (no
        // median: slot hi, which is not in the batch -- and whose offset no contributing lane holds: the idle slot BWD_BATCH
        // fails the pair test).  As byte offsets:
        const int off_last = (hi - 1 - (int)s.last_contributor) * SLOT_BYTES;
        const int off_median = (hi - (int)s.median_contributor) * SLOT_BYTES;
        // (issue priority by what is left of the unit, as in blend_fwd_kernel: 397 -> 381 us)
        if (hi - seg_begin > BWD_BATCH) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
        __syncthreads();
        {
            // threads 0..127 stage (back to front: slot t <-> list entry hi-1-t) and test their entry against all eight
            // blocks; all threads zero the accumulators
            const bool have = (int)threadIdx.x < cnt;
            if (wave < BWD_BATCH / 64) {
                FootprintTest foot = no_footprint();
                if (have) {
                    const uint32_t id = point_list[r0 + (uint32_t)(hi - 1 - (int)threadIdx.x)];
                    s_id[threadIdx.x] = id;
                    foot = stage_record(s_rec, threadIdx.x, rec, id);
                }
                publish_block_masks<BWD_BATCH / 64, 8>(s_mask8, have, foot, tc.tx * TILE, tc.ty * TILE, wave, 0, lane,
                                                       flags & FLAG_NO_CULL);
            }
            for (int i = threadIdx.x; i < BWD_BATCH * ACC_FLOATS / 4; i += 256)
                reinterpret_cast<float4*>(s_acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();

        // Half walk: lanes 0-31 and 32-63 (the upper and lower 8x4 block of the quadrant) each follow their OWN cull mask --
        // one trip evaluates the next entry of block A on the upper lanes and the next entry of block B on the lower ones,
        // so a 64-entry word costs max(|A|, |B|) trips instead of |A u B|.  Both bit streams are scalar; only the entry's
        // LDS address differs between the halves.
        // (the halves re-align at every 64-entry mask word; letting them run on independently over the batch saves 2 % of the
        // trips and costs more than that in scalar bookkeeping -- measured, round 4)
#pragma unroll 1
        for (int k = 0; k < BWD_BATCH / 64; k++) {
            unsigned long long mA = uniform_u64(s_mask8[2 * wave][k]), mB = uniform_u64(s_mask8[2 * wave + 1][k]);
            {
                // entries at or beyond a half's last contributor: slot j <-> list entry hi - 1 - j, i.e. the low bits
                const int dA = hi - half_last[0] - k * 64, dB = hi - half_last[1] - k * 64;
                mA = dA >= 64 ? 0ull : (dA > 0 ? mA & (~0ull << dA) : mA);
                mB = dB >= 64 ? 0ull : (dB > 0 ? mB & (~0ull << dB) : mB);
            }
            while (mA | mB) {
                // (a half whose word is exhausted evaluates the all-zero record in slot BWD_BATCH: no pixel passes its test)
                // Round 5: the slot travels as its BYTE OFFSET, 80 j -- the same for the record (s_rec, 80 bytes) and for the
                // accumulator row (s_acc, ACC_FLOATS floats) -- and the two tests on the entry's list position compare offsets
                // (off_last, off_median: per batch), so the per-lane select is the only vector instruction the walk's
                // bookkeeping costs a trip (it was select, multiply, subtract; the scalar unit does the multiplications)
                const int oA = (mA ? k * 64 + __builtin_ctzll(mA) : BWD_BATCH) * SLOT_BYTES;
                const int oB = (mB ? k * 64 + __builtin_ctzll(mB) : BWD_BATCH) * SLOT_BYTES;
                mA &= mA - 1;
                mB &= mB - 1;
                const int off = lane < 32 ? oA : oB;
                const float4* r = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + off);
                const float4 q0 = r[0], q1 = r[1], q2 = r[2];
                const float Tu[3] = {q0.x, q0.y, q0.z}, Tv[3] = {q0.w, q1.x, q1.y}, Tw[3] = {q1.z, q1.w, q2.x};
                PairEval e;
                // (contributor = hi - 1 - j < last_contributor  <=>  80 j > off_last; outside pixels have last_contributor 0)
                const bool ok = eval_pair_flat(Tu, Tv, Tw, q2.y, q2.z, q2.w, pixx, pixy, e) && off > off_last;
                if (!__any(ok)) continue;
                // (a pair has centre-offset terms only where the low-pass disc won, rho2d < rho3d: the branch bwd_pair_geometry
                // takes -- its compare, ANDed with `ok` on the scalar unit, replaces two compares of the products with zero.
                // Taken here, as a wave-uniform flag: a lane mask carried across the masked blocks below travels through a VGPR)
                const bool any2d = __builtin_amdgcn_ballot_w64(ok && !(e.rho3d <= e.rho2d)) != 0ull;   // (__any(int) widens the predicate in a VGPR)
                PairGrad pg;
                pg.w = pg.dL_dalpha = pg.dL_dz = 0.f;
                if (ok) {
                    float4 q3 = r[3], q4 = r[4];
#if SURFEL_WIDE_RECORD_READS
                    // (all four floats "used": a ds_read_b128 takes 4 LDS cycles per wave instruction, the ds_read_b96 the
                    // compiler narrows these to -- only x, y, z are read -- takes 8: MI355X_MICROARCH.md, LDS table)
                    asm volatile("" : "+v"(q3.w), "+v"(q4.w));
#endif
                    const float nrm[3] = {q3.x, q3.y, q3.z}, rgb[3] = {q4.x, q4.y, q4.z};
                    pg = bwd_pair_core<MODE>(s, e, nrm, rgb, off == off_median);
                }
                e.sanitise(ok);
                float g[ACC_FLOATS];
                bwd_pair_geometry<MODE>(s, e, pg, Tw, q2.w, pixx, pixy, g);
                float v[16] = {g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[A_OPAC],
                               g[A_NRM], g[A_NRM + 1], g[A_NRM + 2], g[A_RGB], g[A_RGB + 1], g[A_RGB + 2]};
#if SURFEL_ABLATE & 1   // (tools/gpu_r5_b.sh: what the ladder costs -- the lanes' values go nowhere in particular)
                float r16 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) r16 += v[i];
#else
                const float r16 = half_reduce_scatter16(v, lane);
#endif
#if SURFEL_ABLATE & 2   // (... and the LDS float atomics)
                if (r16 == 123.456f) s_acc[lane] = r16;
#else
                if (adds16 && r16 != 0.f) atomicAdd(reinterpret_cast<float*>(acc16 + off), r16);
#endif
                if (any2d) {
                    const float r2 = row_reduce_scatter2(g[A_M2D], g[A_M2D + 1], lane);
                    if ((lane & 7) == 0 && r2 != 0.f) atomicAdd(reinterpret_cast<float*>(acc2 + off), r2);
                }
            }
        }
        __syncthreads();
        // flush: consecutive lanes take consecutive floats of a record, so that the atomics of one wave instruction
        // fall into ~6 cache lines (3.2 records of 80 B) instead of 64
        for (int idx = threadIdx.x; idx < cnt * ACC_FLOATS; idx += 256) {
            const float val = s_acc[idx];
#if SURFEL_ABLATE & 4   // (... and the global flush)
            if (val == 123.456f) {
#else
            if (val != 0.f) {
#endif
                const int ent = idx / ACC_FLOATS;
                atomicAdd(acc + (size_t)s_id[ent] * ACC_FLOATS + (idx - ent * ACC_FLOATS), val);
            }
        }
    }
}

// Diagnostic (vidu4d_surfel_blend_stats): what the backward's walk looks like for the frame at hand, counted by walking
// every tile whole with the backward's own staging, cull masks and skip rules (no recurrences, no gradients).
//   [0] list entries staged (per workgroup batch, summed)          [1] wave trips (one entry per half wave) that evaluate
//   [2] trips in which some lane contributes                       [3] contributing lanes (pairs) in total
//   [4] 16-lane rows with a contributing lane, summed over [2]     [5..9] trips of [2] with <= 4, 8, 16, 32, 64 lanes
//   [10] trips of [1] in which no lane passes the pair test itself (the cull test's footprints reach the two 8x4 blocks,
//        the exact ones do not; the rest of [1] - [2] found every pixel they reach finished)
__global__ __launch_bounds__(256) void blend_bwd_stats_kernel(int W, int H, int grid_x, int grid_y, ImageState img,
                                                             const uint32_t* __restrict__ point_list,
                                                             const float* __restrict__ rec,
                                                             unsigned long long* __restrict__ out, int flags)
{
    __shared__ float4 s_rec[BWD_BATCH * 5];
    __shared__ unsigned long long s_mask8[8][BWD_BATCH / 64];
    __shared__ uint32_t s_max;
    TileCoord tc;
    tc.tile = (int)img.tile_order[blockIdx.x];
    tc.tx = tc.tile % grid_x;
    tc.ty = tc.tile / grid_x;
    size_t HW;
    const size_t frame_base = frame_of_tile(tc, W, H, grid_y, HW);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = tc.tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = tc.ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pixx = (float)px + 0.5f, pixy = (float)py + 0.5f;
    const uint32_t r0 = img.ranges[2 * tc.tile];
    const uint32_t last = inside ? img.n_contrib[frame_base + (size_t)py * W + px] : 0;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    uint32_t wave_last = last;
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)wave_last, d, 64);
        wave_last = o > wave_last ? o : wave_last;
    }
    const int half_last[2] = {(int)__builtin_amdgcn_readlane(wave_last, 0), (int)__builtin_amdgcn_readlane(wave_last, 32)};
    wave_last = (uint32_t)(half_last[0] > half_last[1] ? half_last[0] : half_last[1]);
    if (lane == 0) atomicMax(&s_max, wave_last);
    __syncthreads();
    const int n_used = (int)s_max;
    unsigned long long c_staged = 0, c_trips = 0, c_full = 0, c_lanes = 0, c_rows = 0, c_hist[5] = {0, 0, 0, 0, 0}, c_geo_empty = 0;
    for (int hi = n_used; hi > 0; hi -= BWD_BATCH) {
        const int cnt = hi < BWD_BATCH ? hi : BWD_BATCH;
        __syncthreads();
        const bool have = (int)threadIdx.x < cnt;
        if (wave < BWD_BATCH / 64) {
            FootprintTest foot = no_footprint();
            if (have) foot = stage_record(s_rec, threadIdx.x, rec, point_list[r0 + (uint32_t)(hi - 1 - (int)threadIdx.x)]);
            publish_block_masks<BWD_BATCH / 64, 8>(s_mask8, have, foot, tc.tx * TILE, tc.ty * TILE, wave, 0, lane, flags & FLAG_NO_CULL);
        }
        __syncthreads();
        if (wave == 0) c_staged += (unsigned long long)cnt;
        for (int k = 0; k < BWD_BATCH / 64; k++) {
            unsigned long long mA = uniform_u64(s_mask8[2 * wave][k]), mB = uniform_u64(s_mask8[2 * wave + 1][k]);
            const int dA = hi - half_last[0] - k * 64, dB = hi - half_last[1] - k * 64;
            mA = dA >= 64 ? 0ull : (dA > 0 ? mA & (~0ull << dA) : mA);
            mB = dB >= 64 ? 0ull : (dB > 0 ? mB & (~0ull << dB) : mB);
            while (mA | mB) {
                const int jA = mA ? __builtin_ctzll(mA) : 0, jB = mB ? __builtin_ctzll(mB) : 0;
                const bool liveA = mA != 0, liveB = mB != 0;
                mA &= mA - 1;
                mB &= mB - 1;
                const int j = k * 64 + (lane < 32 ? jA : jB);
                const bool live = lane < 32 ? liveA : liveB;
                const uint32_t contributor = (uint32_t)(hi - 1 - j);
                const float4 q0 = s_rec[j * 5 + 0], q1 = s_rec[j * 5 + 1], q2 = s_rec[j * 5 + 2];
                const float Tu[3] = {q0.x, q0.y, q0.z}, Tv[3] = {q0.w, q1.x, q1.y}, Tw[3] = {q1.z, q1.w, q2.x};
                PairEval e;
                const bool geo = eval_pair_flat(Tu, Tv, Tw, q2.y, q2.z, q2.w, pixx, pixy, e) && inside && live;
                const bool ok = geo && contributor < last;
                const unsigned long long b = __ballot(ok);
                c_trips++;
                if (!__ballot(geo)) c_geo_empty++;
                if (!b) continue;
                const int n = __builtin_popcountll(b);
                c_full++;
                c_lanes += (unsigned long long)n;
                c_rows += (unsigned long long)(((b & 0xffffull) != 0) + ((b & 0xffff0000ull) != 0) + ((b & 0xffff00000000ull) != 0) +
                                               ((b & 0xffff000000000000ull) != 0));
                c_hist[n <= 4 ? 0 : n <= 8 ? 1 : n <= 16 ? 2 : n <= 32 ? 3 : 4]++;
            }
        }
    }
    if (lane == 0) {
        atomicAdd(out + 0, c_staged);
        atomicAdd(out + 1, c_trips);
        atomicAdd(out + 2, c_full);
        atomicAdd(out + 3, c_lanes);
        atomicAdd(out + 4, c_rows);
        for (int i = 0; i < 5; i++) atomicAdd(out + 5 + i, c_hist[i]);
        atomicAdd(out + 10, c_geo_empty);
    }
}

void launch_blend_bwd_stats(const BackwardArgs& a, unsigned long long* counters, hipStream_t stream)
{
    const int tiles = total_tiles(a.cam), grid_y = a.cam.grid_y * a.cam.frames;
    hipLaunchKernelGGL(blend_bwd_stats_kernel, dim3(tiles), dim3(256), 0, stream, a.cam.W, a.cam.H, a.cam.grid_x, grid_y, a.img,
                       a.point_list, a.geom.rec, counters, a.flags);
}

// (experiments, tools/occ_probe.sh: VIDU4D_BWD_PAD_LDS=<bytes> of unused dynamic LDS per workgroup caps the workgroups
// resident per CU)
static int bwd_pad_lds()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VIDU4D_BWD_PAD_LDS");
        v = e ? atoi(e) : 0;
    }
    return v;
}

template <bool SPLIT>
static auto pick_bwd(int mode)
{
    return mode == BLEND_LITE ? &blend_bwd_kernel<SPLIT, BLEND_LITE>
           : mode == BLEND_GEOM ? &blend_bwd_kernel<SPLIT, BLEND_GEOM>
                                : &blend_bwd_kernel<SPLIT, BLEND_FULL>;
}

// What runs in front of blend_bwd, as one launch: the gradient accumulator is zeroed (workgroups [0, gridDim.x - 1)), and --
// recorded segments -- the last workgroup turns the forward's per-tile counts of live full segments (ImageState::live_count)
// into their exclusive prefix over the schedule positions and the total (live_prefix, Header::num_live_full), which is what
// blend_bwd's workgroups find their segment with.  On frames whose long lists saturate early most segments are dead: a
// workgroup per dead segment (3 us each until it had found that out, interleaved with the live ones in dispatch order)
// delayed the live ones by a quarter of the launch (tools/bwd_trace.py on a dense ball: 30 k dead against 4 k live).
__global__ __launch_bounds__(256) void bwd_prepare_kernel(uint4* __restrict__ acc, size_t n16, Header* hdr, ImageState img, int tiles)
{
    if (blockIdx.x > 0) {
        const size_t i = (size_t)(blockIdx.x - 1) * 256 + threadIdx.x;
        if (i < n16) acc[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    // (workgroup 0, dispatched first: the scan runs under the zero fill)
    if (hdr->split_used != 2u) return;
    constexpr int PER = REC_MAX_TILES / 256;  // positions per thread (recorded segments exist for at most REC_MAX_TILES tiles)
    __shared__ uint32_t s_wave[4];
    const uint32_t S = min(min(hdr->num_split_pos, (uint32_t)tiles), (uint32_t)REC_MAX_TILES);
    const uint32_t lo = threadIdx.x * PER;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        v[i] = lo + i < S ? img.live_count[lo + i] : 0u;
        sum += v[i];
    }
    uint32_t total;
    uint32_t run = block_exclusive_scan(sum, s_wave, total);
    if (hdr->xcd_block) {
        // XCD-local schedule: the prefix per residue class of the position (find_work_recorded), F = 8 x the largest class
        // total -- unless the classes are so uneven that the padding would outgrow the launch (the host's grid allows a
        // quarter more than the segments there can be, launch_blend_bwd): then the plain numbering below.
        static_assert(PER % 8 == 0, "a thread's positions start at residue 0");
        uint32_t sum8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, run8[8], most = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) sum8[i & 7] += v[i];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            uint32_t tot;
            __syncthreads();
            run8[g] = block_exclusive_scan(sum8[g], s_wave, tot);
            most = max(most, tot);
        }
        if (8u * most <= total + total / 4u + 64u) {
            if (threadIdx.x == 0) {
                hdr->num_live_full = 8u * most;
                hdr->live_xcd = 1u;
            }
#pragma unroll
            for (int i = 0; i < PER; i++) {
                if (lo + i < S) img.live_prefix[lo + i] = run8[i & 7];
                run8[i & 7] += v[i];
            }
            return;
        }
    }
    if (threadIdx.x == 0) {
        hdr->num_live_full = total;
        hdr->live_xcd = 0u;
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        if (lo + i < S) img.live_prefix[lo + i] = run;
        run += v[i];
    }
}

void launch_bwd_prepare(const BackwardArgs& a, size_t acc_bytes, hipStream_t stream)
{
    const size_t n16 = acc_bytes / 16;
    hipLaunchKernelGGL(bwd_prepare_kernel, dim3((unsigned)((n16 + 255) / 256) + 1u), dim3(256), 0, stream, (uint4*)a.acc, n16,
                       a.geom.hdr, a.img, total_tiles(a.cam));
}

void launch_blend_bwd(const BackwardArgs& a, hipStream_t stream)
{
    const int tiles = total_tiles(a.cam), grid_y = a.cam.grid_y * a.cam.frames;
    const int pad = bwd_pad_lds();
    if (a.split && a.seg_data) {
        // upper bound of the segment count (the device knows the exact one; the workgroups beyond it leave at once).
        // Recorded segments (find_work_recorded): the FULL segments, at most capacity / REC_SEG_LEN, then one tail per tile.
        // (+ a quarter and 64: the XCD-local numbering pads the full segments to 8 x the largest residue class, bwd_prepare_kernel)
        const int64_t rec_segs = a.capacity / REC_SEG_LEN + 1;
        const int64_t segs = a.recorded ? rec_segs + rec_segs / 4 + 64 : seg_capacity(a.capacity);
        hipLaunchKernelGGL(pick_bwd<true>(a.mode), dim3((int)segs + tiles), dim3(256), pad, stream, a.cam.W, a.cam.H,
                           a.cam.grid_x, grid_y, a.geom.hdr, a.img, a.point_list, a.geom.rec, a.background, a.seg_data,
                           a.max_seg, a.dL_dcolor, a.dL_dothers, a.acc, a.flags
#ifdef SURFEL_BWD_TRACE
                           , a.trace
#endif
                           );
    } else {
        hipLaunchKernelGGL(pick_bwd<false>(a.mode), dim3(tiles), dim3(256), pad, stream, a.cam.W, a.cam.H, a.cam.grid_x, grid_y,
                           a.geom.hdr, a.img, a.point_list, a.geom.rec, a.background, a.seg_data, 0, a.dL_dcolor, a.dL_dothers,
                           a.acc, a.flags
#ifdef SURFEL_BWD_TRACE
                           , a.trace
#endif
                           );
    }
}

}  // namespace surfel
