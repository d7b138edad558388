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

constexpr int PRE_BLOCK = 256;           // surfels per preprocess / emit workgroup
constexpr int SORT_ITEMS = 16;           // keys per thread in one radix-sort workgroup
constexpr int SORT_BLOCK = 256;
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;  // 4096 keys per workgroup
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

struct Header {           // first 256 bytes of the geometry buffer
    uint32_t num_rendered;  // R, written by the scan kernel
    uint32_t overflow;      // set by emit when R > capacity
    uint32_t pad[62];
};

struct GeomState {
    Header* hdr;
    float* rec;              // [P][20]   (surfel_math.h RecSlot)
    uint32_t* tiles_touched; // [P]
    uint32_t* block_sums;    // [ceil(P/256)]
    uint32_t* block_offsets; // [ceil(P/256)] exclusive scan of block_sums
};

struct ImageState {
    float* final_T;       // [3][H*W]  T, dist1, dist2
    uint32_t* n_contrib;  // [2][H*W]  last contributor, median contributor
    uint32_t* ranges;     // [tiles][2]
};

struct BinState {
    uint64_t* keys[2];    // ping-pong
    uint32_t* vals[2];
    uint32_t* counts;     // [RADIX][sort_blocks] per-pass digit histogram / offsets
    int sort_blocks;
};

template <typename T>
inline void carve(char*& p, T*& out, size_t count)
{
    uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255);
    out = reinterpret_cast<T*>(a);
    p = reinterpret_cast<char*>(out + count);
}

inline int pre_blocks(int P) { return (P + PRE_BLOCK - 1) / PRE_BLOCK; }

inline size_t carve_geom(char* base, int P, GeomState& g)
{
    char* p = base;
    carve(p, g.hdr, 1);
    carve(p, g.rec, (size_t)P * REC_FLOATS);
    carve(p, g.tiles_touched, (size_t)P);
    carve(p, g.block_sums, (size_t)pre_blocks(P) + 1);
    carve(p, g.block_offsets, (size_t)pre_blocks(P) + 1);
    return (size_t)(p - base) + 256;
}

inline size_t carve_image(char* base, int W, int H, ImageState& s)
{
    char* p = base;
    const size_t hw = (size_t)W * H;
    const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    carve(p, s.final_T, 3 * hw);
    carve(p, s.n_contrib, 2 * hw);
    carve(p, s.ranges, 2 * tiles);
    return (size_t)(p - base) + 256;
}

inline size_t carve_binning(char* base, int64_t capacity, BinState& b)
{
    char* p = base;
    const size_t cap = (size_t)(capacity > 0 ? capacity : 0);
    b.sort_blocks = (int)((cap + SORT_TILE - 1) / SORT_TILE);
    carve(p, b.keys[0], cap);
    carve(p, b.keys[1], cap);
    carve(p, b.vals[0], cap);
    carve(p, b.vals[1], cap);
    carve(p, b.counts, (size_t)RADIX * (size_t)(b.sort_blocks > 0 ? b.sort_blocks : 1));
    return (size_t)(p - base) + 256;
}

// Number of 8-bit radix passes for keys of (32 + tile bits) significant bits
// (rasterizer_impl.cu:301-309: SortPairs(..., 0, 32 + bit)).
inline int higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return (int)msb;
}
inline int sort_passes(int grid_x, int grid_y)
{
    const int bits = 32 + higher_msb((uint32_t)(grid_x * grid_y));
    return (bits + RADIX_BITS - 1) / RADIX_BITS;
}

// Camera as passed to kernels: scalars by value, view matrix and camera position stay in device
// memory (they are device tensors at the reference boundary) and are fetched with wave-uniform
// loads at kernel start -- no host round trip.
struct CameraParams {
    const float* view;    // (4,4) device
    const float* campos;  // (3) device
    float focal_x, focal_y, cx, cy, tan_fovx, tan_fovy;
    int W, H, grid_x, grid_y, sh_degree, sh_coeffs;
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
    return c;
}

#if defined(__HIPCC__)
__device__ __forceinline__ Camera load_camera(const CameraParams& p)
{
    Camera c;
#pragma unroll
    for (int k = 0; k < 16; k++) c.view[k] = p.view[k];
#pragma unroll
    for (int k = 0; k < 3; k++) c.campos[k] = p.campos[k];
    c.focal_x = p.focal_x;
    c.focal_y = p.focal_y;
    c.cx = p.cx;
    c.cy = p.cy;
    c.tan_fovx = p.tan_fovx;
    c.tan_fovy = p.tan_fovy;
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
    int32_t* radii;
    GeomState geom;
};
void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t stream);
void launch_scan_blocks(const GeomState& g, int P, uint32_t* ranges, int num_tiles, hipStream_t stream);
void launch_emit_keys(const CameraParams& cam, int P, const int32_t* radii, const GeomState& g, const BinState& b,
                      int64_t capacity, hipStream_t stream);
// Sorts keys[0]/vals[0] (num_rendered read on the device); returns which ping-pong side holds the result.
int launch_radix_sort(const GeomState& g, const BinState& b, int64_t capacity, int passes, hipStream_t stream);
void launch_tile_ranges(const GeomState& g, const uint64_t* sorted_keys, int64_t capacity, uint32_t* ranges,
                        hipStream_t stream);
void launch_blend_fwd(const CameraParams& cam, const GeomState& g, const ImageState& img, const uint32_t* point_list,
                      const float* background, float* out_color, float* out_others, hipStream_t stream);

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
    float* acc;  // [P][20] workspace
    float* dL_dmeans2D;
    float* dL_dcolors;
    float* dL_dopacity;
    float* dL_dmeans3D;
    float* dL_dtransMat;
    float* dL_dsh;
    float* dL_dscales;
    float* dL_drotations;
};
void launch_blend_bwd(const BackwardArgs& a, hipStream_t stream);
void launch_preprocess_bwd(const BackwardArgs& a, hipStream_t stream);
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream);

}  // namespace surfel
