// surfel_state.h -- layout of the three opaque scratch buffers and the launch interfaces between
// the translation units of libvidu4d_surfel.so.
//
// The reference carves GeometryState / ImageState / BinningState out of byte buffers with a
// 128-byte aligned bump allocator (rasterizer_impl.h:21-27, rasterizer_impl.cu:155-194) and
// re-derives the typed pointers in backward from the same function.  We do the same (256-byte
// aligned), but the contents are laid out for gfx950: one 80-byte AoS record per surfel that the
// blend kernels gather with five 16-byte loads, instead of six separate arrays.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "surfel_math.h"

namespace surfel {

constexpr int PRE_BLOCK = 256;   // surfels per preprocess / emit workgroup
constexpr int TILE_SLICES = 16;     // (atomic path) sub-counters per tile: spreads same-address serialisation
// Grouped binning (the default): surfels are processed in <= BIN_MAX_GROUPS groups by 1024-thread
// workgroups that count / place their pairs with LDS atomics only; the per-(group, tile) counts are
// prefix-summed in between.  Needs one LDS word per tile.
constexpr int BIN_THREADS = 1024;
#ifndef SURFEL_BIN_MAX_GROUPS
#define SURFEL_BIN_MAX_GROUPS 256
#endif
constexpr int BIN_MAX_GROUPS = SURFEL_BIN_MAX_GROUPS;
constexpr int BIN_MAX_TILES = 16384;  // 64 KiB of LDS counters
constexpr int TILE_SORT_CAP = 3584;  // list entries a tile can sort entirely inside LDS (2 x 28 KiB ping-pong)

// Segment-parallel blending of long tile lists.  A tile whose list is longer than SPLIT_MIN entries
// is cut into segments of SEG_LEN entries, each blended by its own workgroup: front-to-back
// compositing is associative, so a segment needs from its predecessors only the transmittance they
// leave behind (one float per pixel).  See blend.hip for the three passes.
#ifndef SURFEL_SEG_LEN
#define SURFEL_SEG_LEN 512
#endif
#ifndef SURFEL_SPLIT_MIN
#define SURFEL_SPLIT_MIN 1024
#endif
constexpr int SEG_LEN = SURFEL_SEG_LEN;      // (a multiple of the blend kernels' batch sizes)
constexpr int SPLIT_MIN = SURFEL_SPLIT_MIN;
constexpr int SEG_FLOATS = 16;  // per (segment, pixel) values, slot-major: seg_data[(slot * 16 + k) * 256 + pixel]
enum SegSlot {
    SG_TSEG = 0,   // product of (1 - alpha) over the segment (pass 1)
    SG_C = 1,      // colour partial sum (3)          -- pass 2; after the combine: sum over LATER segments
    SG_D = 4,      // depth partial
    SG_N = 5,      // normal partial (3)
    SG_TEND = 8,   // transmittance at the end of the segment, -1 when the pixel was saturated before it
    SG_M1 = 9,     // sum w m  (distortion first moment)
    SG_M2 = 10,    // sum w m^2
    SG_DIST = 11,  // distortion partial (segment-local moments, global accumulated alpha)
    SG_MED_D = 12, SG_MED_W = 13, SG_MED_C = 14,  // median depth / weight / contributor seen in this segment
    SG_LAST = 15   // last contributor seen in this segment
};
constexpr uint32_t SEG_NONE = 0xFFFFFFFFu;
// segments of all split tiles: sum ceil(len / SEG_LEN) <= R / SEG_LEN + (#tiles longer than SPLIT_MIN)
inline int64_t seg_capacity(int64_t capacity) { return capacity / SEG_LEN + capacity / SPLIT_MIN + 2; }

// Recorded segments (round 4): the backward of a frame whose forward walked every tile WHOLE still runs segment-parallel.
// The forward is a sequential walk anyway; at every REC_SEG_LEN-th list entry of a tile longer than REC_MIN it stores, per
// pixel, the transmittance and the sums of the segment it has just finished (a record: REC_REC_FLOATS x 256 floats) and
// starts the next segment's sums from zero (its totals are base + running segment: a two-level sum); at the end it stores
// the sums of the segment it stopped in.  The backward's back-to-front recurrences of a segment start from "what lies
// behind the segment" = the sum of the later segments' records -- what blend_combine_kernel hands the backward of a
// segment-parallel forward -- each summed from zero, i.e. to the precision of the recurrence they replace (the first
// version stored running totals and took final - prefix: 5e-6 of the gradients' scale instead of 1e-6).  The serial
// chain of a backward workgroup drops from the tile's list to REC_SEG_LEN entries, and the units become small enough for
// the dispatcher to balance the CUs: the whole-tile launch spends a third of its CU-time with fewer than four workgroups
// per CU (tools/bwd_trace.py), this one a tenth.
#ifndef SURFEL_REC_SEG_LEN
#define SURFEL_REC_SEG_LEN 256
#endif
#ifndef SURFEL_REC_MIN
#define SURFEL_REC_MIN 320
#endif
#ifndef SURFEL_REC_MAX_TILES
#define SURFEL_REC_MAX_TILES 4096
#endif
constexpr int REC_MAX_TILES = SURFEL_REC_MAX_TILES;  // launches with more tiles (stacked frames counted) walk whole tiles in both directions
constexpr int REC_SEG_LEN = SURFEL_REC_SEG_LEN;  // (a multiple of the forward's batch, 256 entries)
constexpr int REC_MIN = SURFEL_REC_MIN;
constexpr int REC_REC_FLOATS = 11;  // per (segment, pixel) values of a record: seg_data[(slot * 11 + k) * 256 + pixel]
enum RecordedSlot {
    RS_T = 0,     // record of segment q (slot first + q, written at its end): transmittance after its last entry;
                  // FINAL record (the tile's last slot, written when the walk ends): the median weight
    RS_C = 1,     // sums over THIS segment only -- final record: over the segment the walk stopped in --: colour (3)
    RS_D = 4,     // depth
    RS_N = 5,     // normal (3)
    RS_M1 = 8,    // distortion moments sum w (m - m0), sum w (m - m0)^2 about the tile's reference (ImageState::tile_m0): the
    RS_M2 = 9,    // RUNNING TOTALS at the segment's end (the finals are ImageState::final_T planes 1, 2)
    RS_STOP = 10  // final record: index of the segment the walk stopped in (bits of a uint32)
};
inline int64_t rec_seg_capacity(int64_t capacity) { return capacity / REC_SEG_LEN + capacity / REC_MIN + 2; }
// floats of seg_data: the larger of the two uses
inline size_t seg_data_floats(int64_t capacity)
{
    const size_t a = (size_t)seg_capacity(capacity) * SEG_FLOATS * 256, b = (size_t)rec_seg_capacity(capacity) * REC_REC_FLOATS * 256;
    return a > b ? a : b;
}
// debug_flags of the ABI as the kernels see them
constexpr int FLAG_NO_CULL = 1;
constexpr int FLAG_XCD_SHIFT = 8;      // bits 8-11: VIDU4D_SCHED_XCD_BLOCK(B) (read by the forward's schedule builder)
constexpr int FLAG_POSITION_ORDER = 8;  // VIDU4D_DEBUG_POSITION_ORDER: the split backward's workgroups in schedule-position order (rounds 2-4)
constexpr int FLAG_PAIR_SHIFT = 12;     // bits 12-15: VIDU4D_SCHED_PAIR(K) (read by the forward: its schedule builder and its whole-tile launch)
constexpr int PAIR_MAX = 1024;          // at most this many tiles of a launch get a pair of workgroups (the host's grid bound)
constexpr int FLAG_SERIAL_REPAIR = 4;   // VIDU4D_DEBUG_SERIAL_REPAIR: the speculated combine as ONE launch (rounds 3-4)

struct Header {           // first 256 bytes of the geometry buffer
    uint32_t num_rendered;  // R, written by the tile scan
    uint32_t overflow;      // set by emit when R > capacity
    uint32_t max_tile_len;  // longest tile list (tile_order_kernel)
    uint32_t num_segments;  // segments over all tiles longer than SPLIT_MIN
    uint32_t num_split_pos; // schedule positions [0, num_split_pos) hold every split tile
    uint32_t split_used;    // 1: the forward blended long tiles segment-parallel (seg_data holds what the combine left);
                            // 2: it walked them whole and recorded prefix / final records per segment
    uint32_t truncated;     // a pixel was still unsaturated after the last segment the caller allowed (max_seg)
    uint32_t scan_arrivals; // workgroups of tile_scan_fused_kernel that have finished their columns (reset per forward)
    uint32_t min_T_bits;    // bits of the smallest final transmittance of the frame (atomic min; reset per forward)
    uint32_t num_buckets;   // buckets the MSD split of the long lists queued for bucket_sort_kernel (reset by the schedule)
    uint32_t seg_len;       // what the segment table was built with: entries per segment ...
    uint32_t split_min;     // ... of the tiles longer than this (SEG_LEN / SPLIT_MIN, or REC_SEG_LEN / REC_MIN: recorded segments)
    uint32_t depth_used;    // a home for Vidu4dSurfelForwardArgs::depth_used inside the buffer (word 12, zeroed by the projection
                            // kernel): a caller that reads the header back anyway points depth_used here and saves a counter
                            // tensor, its reset and its copy
    uint32_t num_live_full; // recorded segments: FULL segments the forward's walks reached (ImageState::live_prefix; written by
                            // the backward's preparation launch)
    uint32_t num_long_tiles; // word 14: tiles whose list is longer than SPLIT_MIN entries, whichever segment table the forward
                             // built (num_split_pos counts positions of the table at hand: tiles above REC_MIN's length class
                             // after a whole-tile forward, positions up to the last tile above SPLIT_MIN after a split one)
    uint32_t xcd_block;      // word 15 (round 6): 0, or the block size B (tiles) of the XCD-local schedule -- tile_order / tail_order hold
                             // eight longest-first queues interleaved, position p taking from the queue of the tiles whose BxB-tile
                             // block maps to XCD p mod 8 (binning.hip tile_order; ScheduleParams::xcd_block)
    uint32_t live_xcd;       // word 16: 1 when the recorded backward numbers its live full segments per residue class of the
                             // schedule position (workgroup 8 i + g takes the i-th live segment of the positions = g mod 8, so a
                             // tile's segments stay on its XCD); written by the backward's preparation launch
    uint32_t num_paired;     // word 17 (ABI 21): the tiles at the schedule positions [0, num_paired) are blended by TWO workgroups
                             // each in an unsplit forward (blend.hip fwd_pair_walk; binning.hip tile_order; ScheduleParams::pair_k)
    uint32_t pad[46];
};
static_assert(sizeof(Header) == 256, "the header is the first 256 bytes of the geometry buffer");

struct GeomState {
    Header* hdr;
    float* rec;              // [P][28]   (surfel_math.h RecSlot)
    uint32_t* tiles_touched; // [P]
    float* colour;           // [P][4]    rgb + clamp mask on their way into the records (preprocess.hip, staged path)
};

struct ImageState {
    float* final_T;        // [3][H*W]  T, dist1, dist2 (the distortion moments about ImageState::tile_m0)
    uint32_t* n_contrib;   // [2][H*W]  last contributor, median contributor
    uint32_t* ranges;      // [tiles][2]
    uint32_t* tile_count;  // [tiles][TILE_SLICES] pair counts (preprocess), then emit cursors (atomic path);
                           // [tiles] per-tile totals (grouped path)
    uint32_t* tile_base;   // [tiles][TILE_SLICES] start of each (tile, slice) sub-segment (atomic path)
    uint32_t* group_counts;  // [groups][tiles] pair counts per surfel group, then exclusive prefix over groups
    uint32_t* tile_order;  // [tiles] tile ids, longest list first: workgroup b of the blend kernels takes tile_order[b]
    uint32_t* seg_first;   // [tiles] first segment slot of a split tile, SEG_NONE otherwise
    uint32_t* seg_prefix;  // [tiles + 1] by schedule position: exclusive prefix of the segment counts
    uint32_t* tail_order;  // [tiles] recorded segments: tile ids by the size of their tail unit (remainder segment of a
                           // split tile, or the whole unsplit tile), largest first
    float* tile_m0;        // [tiles] reference mapped depth of the tile's distortion moments (surfel_math.h FwdPixel::m0):
                           // written by the full blend's forward, read by its backward
    uint32_t* live_count;  // [tiles] recorded segments, by schedule position: full segments the forward's walk of the tile
                           // reached (written by the tile's forward workgroup) ...
    uint32_t* live_prefix; // [tiles + 1] ... and their exclusive prefix (bwd_prepare_kernel): blend_bwd runs a workgroup per
                           // LIVE full segment
};

struct BinState {
    uint64_t* entries;     // [cap] (depth bits << 32 | surfel id), grouped by tile, sorted per tile
    uint64_t* scratch;     // [cap] ping-pong space for tiles too long for LDS
    uint32_t* point_list;  // [cap] sorted surfel ids == the reference's binningState.point_list
    float* seg_data;       // [seg_capacity(cap)][SEG_FLOATS][256] per-segment, per-pixel partial results, or
                           // [rec_seg_capacity(cap)][REC_REC_FLOATS][256] recorded segments
};

template <typename T>
inline void carve(char*& p, T*& out, size_t count)
{
    uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255);
    out = reinterpret_cast<T*>(a);
    p = reinterpret_cast<char*>(out + count);
}

__host__ __device__ inline int pre_blocks(int P) { return (P + PRE_BLOCK - 1) / PRE_BLOCK; }
// grouped binning geometry: each 1024-thread workgroup walks bin_iters(P) x 1024 consecutive surfels
inline int bin_iters(int P)
{
    const long long per = (long long)BIN_THREADS * BIN_MAX_GROUPS;
    const int k = (int)((P + per - 1) / per);
    return k > 1 ? k : 1;
}
inline int bin_groups(int P) { const int sb = BIN_THREADS * bin_iters(P); return (P + sb - 1) / sb; }
inline bool use_grouped_binning(int num_tiles) { return num_tiles <= BIN_MAX_TILES; }

inline size_t carve_geom(char* base, int P, GeomState& g)
{
    char* p = base;
    carve(p, g.hdr, 1);
    carve(p, g.rec, (size_t)P * REC_FLOATS);
    carve(p, g.tiles_touched, (size_t)P);
    carve(p, g.colour, (size_t)P * 4);
    return (size_t)(p - base) + 256;
}

inline size_t carve_image(char* base, int W, int H, ImageState& s, int frames = 1)
{
    char* p = base;
    const size_t hw = (size_t)W * H * frames;
    const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE) * frames;
    carve(p, s.final_T, 3 * hw);
    carve(p, s.n_contrib, 2 * hw);
    carve(p, s.ranges, 2 * tiles);
    carve(p, s.tile_count, tiles * TILE_SLICES);
    carve(p, s.tile_base, tiles * TILE_SLICES);
    carve(p, s.group_counts, tiles <= (size_t)BIN_MAX_TILES ? tiles * BIN_MAX_GROUPS : 0);
    carve(p, s.tile_order, tiles);
    carve(p, s.seg_first, tiles);
    carve(p, s.seg_prefix, tiles + 1);
    carve(p, s.tail_order, tiles);
    carve(p, s.tile_m0, tiles);
    carve(p, s.live_count, tiles);
    carve(p, s.live_prefix, tiles + 1);
    return (size_t)(p - base) + 256;
}

inline size_t carve_binning(char* base, int64_t capacity, BinState& b)
{
    char* p = base;
    const size_t cap = (size_t)(capacity > 0 ? capacity : 0);
    carve(p, b.entries, cap);
    carve(p, b.scratch, cap);
    carve(p, b.point_list, cap);
    carve(p, b.seg_data, cap ? seg_data_floats(capacity) : 0);
    return (size_t)(p - base) + 256;
}

// Camera as passed to kernels: scalars by value, view matrix and camera position stay in device
// memory (they are device tensors at the reference boundary) and are fetched with wave-uniform
// loads at kernel start -- no host round trip.
//
// Frame stacking ("frame || tile keys"): `frames` > 1 frames of one optimizer step are rasterized by ONE
// launch set.  Frame f owns the surfels [f * frame_surfels, (f + 1) * frame_surfels) of the per-surfel arrays (its
// warped centres / orientations; opacity, scale and SH rows are shared and indexed modulo frame_surfels) and the tiles
// [f * T, (f + 1) * T), T = grid_x * grid_y, of every per-tile array; W, H, grid_x, grid_y describe ONE frame, image planes
// are (frames, H, W).  Nothing moves between frames: the binning, the sort and the blend see a taller tile grid.
constexpr int MAX_STACKED_FRAMES = 8;
struct FrameCamera {
    const float* view;
    const float* campos;
    float focal_x, focal_y, tan_fovx, tan_fovy;
};
struct CameraParams {
    const float* view;    // (4,4) device   (frame 0)
    const float* campos;  // (3) device
    float focal_x, focal_y, cx, cy, tan_fovx, tan_fovy;
    int W, H, grid_x, grid_y, sh_degree, sh_coeffs;
    int frames, frame_surfels;
    FrameCamera fc[MAX_STACKED_FRAMES];
};

inline CameraParams make_camera_params(const float* view_dev, const float* campos_dev, int W, int H, float tfx,
                                       float tfy, int D, int M)
{
    CameraParams c;
    c.view = view_dev;
    c.campos = campos_dev;
    c.W = W;
    c.H = H;
    c.grid_x = (W + TILE - 1) / TILE;
    c.grid_y = (H + TILE - 1) / TILE;
    c.tan_fovx = tfx;
    c.tan_fovy = tfy;
    c.focal_y = H / (2.0f * tfy);  // rasterizer_impl.cu:223-224
    c.focal_x = W / (2.0f * tfx);
    c.cx = (float)((double)(float)W / 2.0);  // forward.cu:208
    c.cy = (float)((double)(float)H / 2.0);
    c.sh_degree = D;
    c.sh_coeffs = M;
    c.frames = 1;
    c.frame_surfels = 0;
    for (int f = 0; f < MAX_STACKED_FRAMES; f++) c.fc[f] = FrameCamera{view_dev, campos_dev, c.focal_x, c.focal_y, tfx, tfy};
    return c;
}

// frame f of a stacked launch: its own view matrix, camera centre and field of view
inline void set_frame_camera(CameraParams& c, int f, const float* view_dev, const float* campos_dev, float tfx, float tfy)
{
    c.fc[f] = FrameCamera{view_dev, campos_dev, c.W / (2.0f * tfx), c.H / (2.0f * tfy), tfx, tfy};
}
inline int total_tiles(const CameraParams& c) { return c.grid_x * c.grid_y * c.frames; }

#if defined(__HIPCC__)
__device__ __forceinline__ Camera load_camera(const CameraParams& p, int frame = 0)
{
    Camera c;
    const FrameCamera fc = p.fc[frame];
#pragma unroll
    for (int k = 0; k < 16; k++) c.view[k] = fc.view[k];
#pragma unroll
    for (int k = 0; k < 3; k++) c.campos[k] = fc.campos[k];
    c.focal_x = fc.focal_x;
    c.focal_y = fc.focal_y;
    c.cx = p.cx;
    c.cy = p.cy;
    c.tan_fovx = fc.tan_fovx;
    c.tan_fovy = fc.tan_fovy;
    c.W = p.W;
    c.H = p.H;
    c.grid_x = p.grid_x;
    c.grid_y = p.grid_y;
    c.sh_degree = p.sh_degree;
    c.sh_coeffs = p.sh_coeffs;
    return c;
}
#endif

// ---- launchers (each enqueues on `stream` and returns immediately) ----
struct PreprocessArgs {
    CameraParams cam;
    int P;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* opacities;
    const float* shs;
    const float* colors_precomp;
    const float* sh_dc;    // canonical SH rows in two tensors (P,1,3) + (P,15,3) instead of `shs` (both or neither)
    const float* sh_rest;
    int raw_params;        // scales = log-scales, opacities = logits: activated by the kernels
    int32_t* radii;
    GeomState geom;
    uint32_t* tile_count;    // atomic path: [tiles][TILE_SLICES], zeroed before the launch
    uint32_t* group_counts;  // grouped path: [groups][tiles]
    int iters;               // grouped path: surfel batches per workgroup
    int stage_records;       // grouped path: records staged per wave in LDS and stored lane-contiguously (set by the launcher)
};
void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t stream);
// counts -> ranges (exclusive scan over tiles), total -> header; atomic path: counts reset to 0 (they
// become cursors); grouped path (groups > 0): per-(group, tile) counts become prefixes over the groups
void launch_tile_scan(const GeomState& g, const ImageState& img, int num_tiles, int groups, hipStream_t stream);
// The longest-first schedule and the segment table (binning.hip tile_order): the table is built for tiles longer than
// split_min, in segments of seg_len entries; by_class: split by length class (recorded segments: the split tiles then lead
// the schedule without exception).  Built by launch_emit_keys -- by an extra workgroup of the emit launch where it can.
struct ScheduleParams {
    int seg_len, split_min, by_class;
    // XCD-local longest-first schedule (round 6): 0 = one longest-first queue over all tiles (rounds 1-5); B > 0 = the tiles
    // of a BxB-tile block of a frame share an XCD (workgroup b runs on XCD b mod 8: MI355X_MICROARCH.md, observed), i.e.
    // eight length-sorted queues interleaved over the schedule positions.  grid_x / frame_tiles: the tile grid of ONE frame.
    int xcd_block, grid_x, frame_tiles;
    // Paired workgroups for the longest tiles of an unsplit forward (round 6, blend.hip fwd_pair_walk): 0 off; K = 1..14: the
    // tiles whose length class lies above that of K / 4 x the mean list length of the launch; 15: every tile (tests).  At most
    // PAIR_MAX tiles; not together with the XCD-local schedule.
    int pair_k;
};
void launch_tile_order(const GeomState& g, const ImageState& img, int num_tiles, const ScheduleParams& sp, hipStream_t stream);
void launch_emit_keys(const CameraParams& cam, int P, const int32_t* radii, const GeomState& g, const ImageState& img,
                      const BinState& b, int64_t capacity, bool grouped, const ScheduleParams& sp, hipStream_t stream);
// per-tile stable radix sort of the (depth, id) entries; fills point_list
// How lists longer than the LDS capacity (TILE_SORT_CAP) are sorted (Vidu4dSurfelForwardArgs::long_list_sort):
//   in_lds_only   -- the long-tile machinery is off (segment_split == 0): the 4-wave kernel runs them through global memory;
//   msd_split     -- every list beyond 1024 entries is split on its leading differing depth bits, the buckets sorted in LDS;
//   one_workgroup -- a 16-wave workgroup per long list through global memory, the rest in LDS.
//   short_lists_expected -- as in_lds_only, but the caller expects no list beyond 1024 entries (max_list_hint): ONE launch,
//                   the in-LDS sort of the short lists, which sorts a longer list -- should one appear after all -- through
//                   global memory itself (correct whatever the hint; the launch of the 56 KiB instance, whose workgroups
//                   would all leave at once, is saved).
enum class LongListSort { in_lds_only, msd_split, one_workgroup, short_lists_expected };
constexpr int SHORT_LIST_HINT_MAX = 900;  // (an eighth of headroom below the 1024 entries of the small in-LDS sort)
// true when launch_preprocess_fwd stages its records (and leaves centre / depth / radius per surfel in GeomState::colour)
bool preprocess_stages_records(int num_tiles);
void launch_tile_sort(const GeomState& g, const ImageState& img, const BinState& b, int num_tiles, int num_surfels,
                      int64_t capacity, LongListSort mode, hipStream_t stream);
// split: blend tiles longer than SPLIT_MIN segment-parallel (three launches instead of one)
// max_seg: only the first max_seg segments of a split tile are blended (a caller that knows how deep the
// previous frames went saves the rest of pass 1); Header::truncated is set if that was not enough
// mode: BLEND_FULL / BLEND_LITE / BLEND_GEOM (surfel_math.h); flags: FLAG_*; record: an unsplit walk leaves recorded segments
void launch_blend_fwd(const CameraParams& cam, const GeomState& g, const ImageState& img, const BinState& b,
                      int64_t capacity, bool split, int max_seg, const float* background, float* out_color,
                      float* out_others, uint32_t* depth_used, int mode, bool assume_unsaturated, int flags, bool record,
                      hipStream_t stream);

struct BackwardArgs {
    CameraParams cam;
    int P;
    const float* background;
    const float* means3D;
    const int32_t* radii;
    const float* shs;
    const float* colors_precomp;
    const float* scales;
    const float* rotations;
    const float* dL_dcolor;
    const float* dL_dothers;
    GeomState geom;
    ImageState img;
    const uint32_t* point_list;
    const float* seg_data;  // per-segment state of the segment-parallel forward (or NULL)
    int64_t capacity;
    bool split;             // walk the split tiles segment-parallel (needs the forward to have run split)
    int max_seg;            // the forward's segment limit
    float* acc;  // [P][20] workspace
    float* dL_dmeans2D;
    float* dL_dcolors;
    float* dL_dopacity;
    float* dL_dmeans3D;
    float* dL_dtransMat;
    float* dL_dsh;
    float* dL_dscales;
    float* dL_drotations;
    const float* sh_dc;   // as in PreprocessArgs; then the SH gradient leaves through dL_dsh_dc / dL_dsh_rest
    const float* sh_rest;
    float* dL_dsh_dc;
    float* dL_dsh_rest;
    int raw_params;       // scales are log-scales; dL_dscales / dL_dopacity are w.r.t. log-scales / logits
    int mode;             // BLEND_LITE: only dL_dcolor and plane 1 of dL_dothers are live (aux_planes == alpha only);
                          // BLEND_GEOM: planes 5-7 are taken as zero; BLEND_FULL: everything
    int flags;            // FLAG_*
    bool recorded;        // (with split) the forward walked whole tiles and may have left recorded segments
#ifdef SURFEL_BWD_TRACE
    unsigned long long* trace;  // (variant build for tools/bwd_trace.py) 4 words per workgroup of blend_bwd, or NULL
#endif
};
void launch_bwd_prepare(const BackwardArgs& a, size_t acc_bytes, hipStream_t stream);  // zeroes acc (+ live-segment prefix)
void launch_blend_bwd(const BackwardArgs& a, hipStream_t stream);
void launch_blend_bwd_stats(const BackwardArgs& a, unsigned long long* counters, hipStream_t stream);  // diagnostic
void launch_preprocess_bwd(const BackwardArgs& a, hipStream_t stream);
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream);

}  // namespace surfel
