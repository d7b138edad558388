// preprocess.hip -- per-surfel kernels: projection (forward), tile-count scan, key emission,
// per-surfel backward, frustum marking.  gfx950 only.
//
// Reference behaviour restated here (file:line in /root/reference/gs/submodules/
// diff-surfel-rasterization/cuda_rasterizer/): preprocessCUDA forward.cu:166-260; computeAABB +
// preprocessCUDA backward backward.cu:599-649, :533-597; checkFrustum rasterizer_impl.cu:54-66.
//
// MI355X notes: one thread per surfel, 256-thread workgroups (4 wave64).  The reference's
// per-surfel prefix sum + global 64-bit sort is replaced by tile binning (binning.hip): this kernel
// only counts, per tile, how many surfels touch it.
#include <cstdlib>

#include "surfel_state.h"
#include "wave_utils.h"

namespace surfel {

// Frame of a (stacked) surfel index and its row in the arrays the frames share.
struct FrameIndex {
    int frame, shared;
};
__device__ __forceinline__ FrameIndex frame_index(const CameraParams& cam, int idx)
{
    if (cam.frames <= 1) return FrameIndex{0, idx};
    const int f = idx / cam.frame_surfels;
    return FrameIndex{f, idx - f * cam.frame_surfels};
}

// Activations of the canonical parameters (raw_params; gs/scene/gaussian_model.py:47-57: exp, sigmoid), written as
// torch's elementwise kernels evaluate them -- exp(x) and 1 / (1 + exp(-x)) in fp32 -- so that the raw path gives the
// values torch.exp / torch.sigmoid hand to the activated path.
__device__ __forceinline__ float act_scale(float raw) { return expf(raw); }
__device__ __forceinline__ float act_opacity(float raw) { return 1.0f / (1.0f + expf(-raw)); }

// The [256][49] LDS tile of SH rows (row stride 49 words: the per-thread walk over a row is conflict free).
// Sources: one (P,16,3) tensor -- the 256 rows of a workgroup are one contiguous 48 KiB run -- or the canonical pair
// (P,1,3) + (P,15,3): two contiguous runs (3 KiB into columns 0-2, 45 KiB into columns 3-47).  All copies are
// coalesced 16-byte accesses; the runs start 16-byte aligned because block_first is a multiple of 256.
constexpr int SH_ROW = 48, SH_STRIDE = 49, SH_DC = 3, SH_REST = 45;

template <int WIDTH, int COL0>
__device__ __forceinline__ void tile_load_run(float* s_sh, const float* __restrict__ src, int rows)
{
    const int n = rows * WIDTH, n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(src);
    for (int i = threadIdx.x; i < n4; i += PRE_BLOCK) {
        const float4 v = g4[i];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 4 * i + j;
            s_sh[(k / WIDTH) * SH_STRIDE + COL0 + k % WIDTH] = e[j];
        }
    }
    for (int k = 4 * n4 + threadIdx.x; k < n; k += PRE_BLOCK) s_sh[(k / WIDTH) * SH_STRIDE + COL0 + k % WIDTH] = src[k];
}

template <int WIDTH, int COL0>
__device__ __forceinline__ void tile_store_run(const float* s_sh, float* __restrict__ dst, int rows)
{
    const int n = rows * WIDTH, n4 = n >> 2;
    float4* o4 = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < n4; i += PRE_BLOCK) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 4 * i + j;
            e[j] = s_sh[(k / WIDTH) * SH_STRIDE + COL0 + k % WIDTH];
        }
        o4[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
    for (int k = 4 * n4 + threadIdx.x; k < n; k += PRE_BLOCK) dst[k] = s_sh[(k / WIDTH) * SH_STRIDE + COL0 + k % WIDTH];
}

__device__ __forceinline__ void sh_tile_load(float* s_sh, const float* shs, const float* sh_dc, const float* sh_rest,
                                             int block_first, int rows)
{
    if (sh_dc) {
        tile_load_run<SH_DC, 0>(s_sh, sh_dc + (size_t)block_first * SH_DC, rows);
        tile_load_run<SH_REST, SH_DC>(s_sh, sh_rest + (size_t)block_first * SH_REST, rows);
    } else {
        tile_load_run<SH_ROW, 0>(s_sh, shs + (size_t)block_first * SH_ROW, rows);
    }
}

__device__ __forceinline__ void sh_tile_store(const float* s_sh, float* dL_dsh, float* dL_dsh_dc, float* dL_dsh_rest,
                                              int block_first, int rows)
{
    if (dL_dsh_dc) {
        tile_store_run<SH_DC, 0>(s_sh, dL_dsh_dc + (size_t)block_first * SH_DC, rows);
        tile_store_run<SH_REST, SH_DC>(s_sh, dL_dsh_rest + (size_t)block_first * SH_REST, rows);
    } else {
        tile_store_run<SH_ROW, 0>(s_sh, dL_dsh + (size_t)block_first * SH_ROW, rows);
    }
}

// Projection of surfel idx; writes record / radius / tile count and returns the tile rect.
// stage: NULL, or this lane's seven float4 slots of its wave's LDS staging buffer -- the record then goes there and the
// caller stores the wave's 64 records lane-contiguously (flush_staged_records): a record is 112 bytes, so the direct
// stores below are 16-byte pieces at a 112-byte lane stride.
__device__ __forceinline__ uint32_t preprocess_one(const PreprocessArgs& a, const Camera& cam, int idx, int shared,
                                                   Projected& o, float4* stage = nullptr)
{
    const float p_world[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
    float2 sc = reinterpret_cast<const float2*>(a.scales)[shared];
    if (a.raw_params) sc = make_float2(act_scale(sc.x), act_scale(sc.y));
    const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[idx];
    const float scale[2] = {sc.x, sc.y};
    const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
    uint32_t tiles = 0;
    int radius = 0;
    if (project_surfel(cam, p_world, quat, scale, o)) {
        tiles = o.tiles;
        radius = o.radius;
        // (q4 = colour and clamp mask is written by surfel_color_kernel)
        float4* rec = stage ? stage : reinterpret_cast<float4*>(a.geom.rec + (size_t)idx * REC_FLOATS);
        rec[0] = make_float4(o.T[0], o.T[1], o.T[2], o.T[3]);
        rec[1] = make_float4(o.T[4], o.T[5], o.T[6], o.T[7]);
        const float opacity = a.raw_params ? act_opacity(a.opacities[shared]) : a.opacities[shared];
        rec[2] = make_float4(o.T[8], o.center[0], o.center[1], opacity);
        rec[3] = make_float4(o.normal[0], o.normal[1], o.normal[2], o.depth);
        float f[8];
        contribution_footprint(o.T, o.center[0], o.center[1], opacity, f);
        rec[5] = make_float4(f[0], f[1], f[2], f[3]);
        rec[6] = make_float4(f[4], f[5], f[6], f[7]);
        // (staged path: the colour kernel, which ran before this one, left q4 in a compact array)
        if (stage) rec[4] = reinterpret_cast<const float4*>(a.geom.colour)[idx];
    }
    // (staged path: the compact slot now carries what emit_keys_grouped_kernel reads -- centre, depth bits, radius -- in one
    // lane-contiguous 16-byte load instead of three strided ones)
    if (stage) reinterpret_cast<float4*>(a.geom.colour)[idx] =
        make_float4(o.center[0], o.center[1], o.depth, __int_as_float(radius));
    a.radii[idx] = radius;
    a.geom.tiles_touched[idx] = tiles;
    return tiles;
}

// The 64 records a wave staged in LDS (7 float4 each), stored with consecutive lanes on consecutive 16-byte pieces.
// visible: ballot of the lanes that wrote one.
__device__ __forceinline__ void flush_staged_records(const float4* stage, float* rec, int idx0, unsigned long long visible,
                                                     int lane)
{
    // the lanes read each other's records: the per-lane LDS stores above must be complete (and not sunk below by the
    // compiler) before, and the reads done before the next batch's stores -- wave-level fences, as bucket_sort_small_kernel has
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (visible) {
        float4* dst = reinterpret_cast<float4*>(rec + (size_t)idx0 * REC_FLOATS);
        constexpr int PARTS = REC_USED_FLOATS / 4, STRIDE = REC_FLOATS / 4;  // (staged densely; a padded record keeps its gap)
#pragma unroll
        for (int i0 = 0; i0 < 64 * PARTS; i0 += 64) {
            const int i = i0 + lane, r = i / PARTS;
            if ((visible >> r) & 1ull) dst[PARTS == STRIDE ? i : r * STRIDE + (i - r * PARTS)] = stage[i];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Atomic path (more than BIN_MAX_TILES tiles): per-tile pair counts with global atomics.
__global__ __launch_bounds__(PRE_BLOCK) void preprocess_fwd_kernel(PreprocessArgs a)
{
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    if (idx == 0) {  // (the header is fresh memory)
        a.geom.hdr->min_T_bits = 0x3f800000u;  // 1.0f
        a.geom.hdr->depth_used = 0;
    }
    if (idx >= a.P) return;
    const FrameIndex fi = frame_index(a.cam, idx);
    const Camera cam = load_camera(a.cam, fi.frame);
    Projected o;
    if (preprocess_one(a, cam, idx, fi.shared, o)) {
        const int slice = blockIdx.x & (TILE_SLICES - 1);
        const int tile0 = fi.frame * cam.grid_x * cam.grid_y;
        for (int y = o.y0; y < o.y1; y++)
            for (int x = o.x0; x < o.x1; x++)
                atomicAdd(&a.tile_count[(size_t)(tile0 + y * cam.grid_x + x) * TILE_SLICES + slice], 1u);
    }
}

// Grouped path: a 1024-thread workgroup (16 wave64) projects a.iters x 1024 consecutive surfels and
// histograms their (surfel, tile) pairs in LDS; the row group_counts[group][*] is written once,
// coalesced.  No global atomics.
__global__ __launch_bounds__(BIN_THREADS) void preprocess_fwd_grouped_kernel(PreprocessArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];
    const int frame_tiles = a.cam.grid_x * a.cam.grid_y, num_tiles = frame_tiles * a.cam.frames;
    for (int t = threadIdx.x; t < num_tiles; t += BIN_THREADS) s_hist[t] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // (the header is fresh memory)
        a.geom.hdr->scan_arrivals = 0;
        a.geom.hdr->min_T_bits = 0x3f800000u;  // 1.0f
        a.geom.hdr->depth_used = 0;
    }
    __syncthreads();
    const int first = blockIdx.x * BIN_THREADS * a.iters;
    // (optional: per-wave staging of the records behind the histogram, launch_preprocess_fwd)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PARTS = REC_USED_FLOATS / 4;
    float4* stage = a.stage_records
                        ? reinterpret_cast<float4*>(s_hist + ((num_tiles + 3) & ~3)) + (size_t)wave * 64 * PARTS
                        : nullptr;
    for (int it = 0; it < a.iters; it++) {
        const int idx = first + it * BIN_THREADS + threadIdx.x;
        bool visible = false;
        if (idx < a.P) {
            const FrameIndex fi = frame_index(a.cam, idx);
            const Camera cam = load_camera(a.cam, fi.frame);
            Projected o;
            if (preprocess_one(a, cam, idx, fi.shared, o, stage ? stage + lane * PARTS : nullptr)) {
                visible = true;
                uint32_t* hist = s_hist + fi.frame * frame_tiles;
                for (int y = o.y0; y < o.y1; y++)
                    for (int x = o.x0; x < o.x1; x++) atomicAdd(&hist[y * cam.grid_x + x], 1u);
            }
        }
        if (stage) flush_staged_records(stage, a.geom.rec, first + it * BIN_THREADS + wave * 64, __ballot(visible), lane);
    }
    __syncthreads();
    uint32_t* row = a.group_counts + (size_t)blockIdx.x * num_tiles;
    for (int t = threadIdx.x; t < num_tiles; t += BIN_THREADS) row[t] = s_hist[t];
}

// View-dependent colour of every surfel (computeColorFromSH, forward.cu:20-71) -> record slot q4 (rgb, clamp
// mask).  A kernel of its own: the 192-byte SH rows are the bulk of the forward's input (38 of 46 MB at 200 k
// surfels), and a thread-per-surfel walk over them touches 64 cache lines per wave instruction (the projection
// kernel, which read them in round 1, ran at 1.6 TB/s).  Here the 256 rows of a workgroup -- one contiguous
// 48 KiB run -- are copied with coalesced 16-byte loads into an LDS tile with a 49-word row stride (odd: the
// per-thread walk is conflict free), as the per-surfel backward does.  SH_LDS needs 16 coefficients and a
// 16-byte aligned tensor; otherwise (and for colors_precomp) rows are read directly.
template <bool SH_LDS>
__global__ __launch_bounds__(PRE_BLOCK) void surfel_color_kernel(PreprocessArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_sh[];  // [256][49] when SH_LDS
    // a workgroup takes 256 consecutive rows of the arrays the frames share and evaluates them for EVERY frame of a stacked
    // launch: the SH tile is loaded once (round 3: a workgroup per (frame, 256 rows) read the 38 MB of SH rows per frame)
    const int N = a.cam.frames > 1 ? a.cam.frame_surfels : a.P;
    const int block_first = blockIdx.x * PRE_BLOCK;  // (row in the shared arrays)
    const int shared = block_first + threadIdx.x;
    const int rows = (N - block_first) < PRE_BLOCK ? (N - block_first) : PRE_BLOCK;
    if (SH_LDS) {
        sh_tile_load(s_sh, a.shs, a.sh_dc, a.sh_rest, block_first, rows);
        __syncthreads();
    }
    if (shared >= N) return;
    const int F = a.cam.frames > 1 ? a.cam.frames : 1;
    for (int frame = 0; frame < F; frame++) {
        const int idx = frame * N + shared;
        float rgb[3];
        uint32_t clamp_mask = 0;
        if (a.colors_precomp == nullptr) {
            const float p_world[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
            const float* cp = a.cam.fc[frame].campos;
            const float campos[3] = {cp[0], cp[1], cp[2]};
            const float* sh = SH_LDS ? s_sh + threadIdx.x * SH_STRIDE : a.shs + (size_t)shared * a.cam.sh_coeffs * 3;
            sh_forward(a.cam.sh_degree, p_world, campos, sh, rgb, clamp_mask);
        } else {
            rgb[0] = a.colors_precomp[3 * idx];
            rgb[1] = a.colors_precomp[3 * idx + 1];
            rgb[2] = a.colors_precomp[3 * idx + 2];
        }
        // (staged path: a compact array the projection kernel folds into the records it stores whole; else straight into q4)
        float4* dst = a.stage_records ? reinterpret_cast<float4*>(a.geom.colour) + idx
                                      : reinterpret_cast<float4*>(a.geom.rec + (size_t)idx * REC_FLOATS) + 4;
        *dst = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float(clamp_mask));
    }
}

static bool stage_records_enabled();
bool preprocess_stages_records(int num_tiles)
{
    static int big_lds = -1;  // (one attribute call; benign if raced)
    if (big_lds < 0)
        big_lds = hipFuncSetAttribute(reinterpret_cast<const void*>(&preprocess_fwd_grouped_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    const size_t hist = (size_t)((num_tiles + 3) & ~3) * sizeof(uint32_t);
    const size_t staged = hist + (size_t)(BIN_THREADS / 64) * 64 * REC_USED_FLOATS * sizeof(float);
    return use_grouped_binning(num_tiles) && big_lds == 1 && staged <= 160 * 1024 && stage_records_enabled();
}

// VIDU4D_STAGE_RECORDS=0: the projection kernel stores its records directly (A/B of the LDS staging)
static bool stage_records_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("VIDU4D_STAGE_RECORDS");
        v = e ? atoi(e) : 1;
    }
    return v != 0;
}

void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t stream)
{
    if (a.P <= 0) return;
    const int num_tiles = total_tiles(a.cam);
    // (the canonical pair sh_dc / sh_rest only exists on the LDS path; capi checks its alignment)
    const bool sh_lds = a.colors_precomp == nullptr && a.cam.sh_coeffs == 16 &&
                        (a.sh_dc != nullptr || (a.shs != nullptr && (reinterpret_cast<uintptr_t>(a.shs) & 15) == 0));
    const int color_blocks = a.cam.frames > 1 ? pre_blocks(a.cam.frame_surfels) : pre_blocks(a.P);
    // grouped path: histogram + (if it fits) a staging buffer of 64 records per wave -- the records then leave with
    // lane-contiguous stores, the colour kernel's q4 included (it writes a compact array the projection kernel reads)
    PreprocessArgs g = a;
    g.stage_records = 0;
    const size_t hist = (size_t)((num_tiles + 3) & ~3) * sizeof(uint32_t);
    const size_t staged = hist + (size_t)(BIN_THREADS / 64) * 64 * REC_USED_FLOATS * sizeof(float);
    g.stage_records = preprocess_stages_records(num_tiles) ? 1 : 0;
    if (sh_lds)
        hipLaunchKernelGGL(surfel_color_kernel<true>, dim3(color_blocks), dim3(PRE_BLOCK),
                           (size_t)PRE_BLOCK * SH_STRIDE * sizeof(float), stream, g);
    else
        hipLaunchKernelGGL(surfel_color_kernel<false>, dim3(color_blocks), dim3(PRE_BLOCK), 0, stream, g);
    if (use_grouped_binning(num_tiles))
        hipLaunchKernelGGL(preprocess_fwd_grouped_kernel, dim3(bin_groups(a.P)), dim3(BIN_THREADS),
                           g.stage_records ? staged : hist, stream, g);
    else
        hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(pre_blocks(a.P)), dim3(PRE_BLOCK), 0, stream, a);
}

// Per-surfel backward.  Reads the accumulator filled by the backward blend and writes every output
// gradient (zeros for culled surfels, which the reference gets from torch::zeros).
//
// SH_LDS (16 coefficients, 16-byte aligned tensors): the 192-byte SH rows of the workgroup's 256
// surfels are one contiguous 48 KiB run in global memory, but a thread-per-surfel access walks it
// with a 192-byte lane stride (every wave instruction touches 64 cache lines; measured 2.6x
// over-fetch and 2.5x over-write).  Instead the run is copied with coalesced 16-byte loads into an
// LDS tile with row stride 49 words (odd => the per-thread column walk is bank-conflict free), the
// gradients are written back into the same tile and leave with coalesced 16-byte stores.

// Stacked frames (cam.frames > 1): workgroups are laid out per frame (pre_blocks(N) each); gradients of what the
// frames share (opacity, scale, SH rows) are ADDED to arrays the caller zero-filled -- the SH tile leaves with
// lane-contiguous float atomics (merged per cache line by the memory side, as the blend backward's flush) -- while the
// per-frame outputs (centres, orientations, screen-space statistic, colours, homography) are plain stores.
template <bool SH_LDS>
__global__ __launch_bounds__(PRE_BLOCK) void preprocess_bwd_kernel(BackwardArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_sh[];  // [256][49] when SH_LDS
    const bool stacked = a.cam.frames > 1;
    const int N = stacked ? a.cam.frame_surfels : a.P;
    const int per_frame = pre_blocks(N);
    const int frame = blockIdx.x / per_frame;
    const int block_first = (blockIdx.x - frame * per_frame) * PRE_BLOCK;  // (row in the shared arrays)
    const int shared = block_first + threadIdx.x;
    const int idx = frame * N + shared;
    const Camera cam = load_camera(a.cam, frame);
    const int M = cam.sh_coeffs;
    const int rows = (N - block_first) < PRE_BLOCK ? (N - block_first) : PRE_BLOCK;
    if (SH_LDS) {
        sh_tile_load(s_sh, a.shs, a.sh_dc, a.sh_rest, block_first, rows);
        __syncthreads();
    }
    const bool has_sh = a.shs != nullptr || a.sh_dc != nullptr;
    const bool live = shared < N;
    const bool visible = live && a.radii[idx] > 0;
    float* sh_row = s_sh + threadIdx.x * SH_STRIDE;  // SH_LDS: this thread's private row (read, then overwritten)
    float* dsh = (a.dL_dsh && live) ? a.dL_dsh + (size_t)shared * M * 3 : nullptr;
    if (live && !visible) {
        for (int k = 0; k < 3; k++) {
            a.dL_dmeans3D[3 * idx + k] = 0.f;
            a.dL_dmeans2D[3 * idx + k] = 0.f;
            a.dL_dcolors[3 * idx + k] = 0.f;
        }
        for (int k = 0; k < 9; k++) a.dL_dtransMat[9 * idx + k] = 0.f;
        for (int k = 0; k < 4; k++) a.dL_drotations[4 * idx + k] = 0.f;
        if (!stacked) {
            a.dL_dopacity[shared] = 0.f;
            a.dL_dscales[2 * shared] = a.dL_dscales[2 * shared + 1] = 0.f;
        }
        if (SH_LDS) {
            for (int k = 0; k < SH_ROW; k++) sh_row[k] = 0.f;
        } else if (dsh && !stacked) {
            for (int k = 0; k < 3 * M; k++) dsh[k] = 0.f;
        }
    }
    if (visible) {
        float acc[ACC_FLOATS];
        {
            // (Round 4 tried to clear the row here and let the caller reuse the zero-filled workspace without the zero-fill
            // launch: the 32 MB of stores cost this kernel 13 us, the launch they replace 9.)
            const float4* p = reinterpret_cast<const float4*>(a.acc + (size_t)idx * ACC_FLOATS);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const float4 v = p[k];
                acc[4 * k] = v.x;
                acc[4 * k + 1] = v.y;
                acc[4 * k + 2] = v.z;
                acc[4 * k + 3] = v.w;
            }
        }
        float T[9];
        uint32_t clamp_mask;
        float opacity_act;  // (the activated opacity the forward stored in the record)
        {
            const float4* r = reinterpret_cast<const float4*>(a.geom.rec + (size_t)idx * REC_FLOATS);
            const float4 q0 = r[0], q1 = r[1], q2 = r[2], q4 = r[4];
            T[0] = q0.x; T[1] = q0.y; T[2] = q0.z; T[3] = q0.w;
            T[4] = q1.x; T[5] = q1.y; T[6] = q1.z; T[7] = q1.w;
            T[8] = q2.x;
            opacity_act = q2.w;
            clamp_mask = __float_as_uint(q4.w);
        }
        const float p_world[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
        float2 sc = reinterpret_cast<const float2*>(a.scales)[shared];
        if (a.raw_params) sc = make_float2(act_scale(sc.x), act_scale(sc.y));
        const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[idx];
        const float scale[2] = {sc.x, sc.y};
        const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
        SurfelGrads o;
        surfel_backward(cam, p_world, quat, scale, T, acc, o);
        if (a.raw_params) {  // chain through exp / sigmoid as torch's backward formulas do (grad * result; (grad * (1 - y)) * y)
            o.dscale[0] *= scale[0];
            o.dscale[1] *= scale[1];
            acc[A_OPAC] = (acc[A_OPAC] * (1.0f - opacity_act)) * opacity_act;
        }

        float dmean[3] = {o.dmean3D[0], o.dmean3D[1], o.dmean3D[2]};
        const float dcol[3] = {acc[A_RGB], acc[A_RGB + 1], acc[A_RGB + 2]};
        if (has_sh) {
            if (SH_LDS) {
                // in place: coefficient k is read before gradient k is stored over it
                sh_backward(cam.sh_degree, M, p_world, cam.campos, sh_row, clamp_mask, dcol, sh_row, dmean);
            } else if (!stacked) {
                sh_backward(cam.sh_degree, M, p_world, cam.campos, a.shs + (size_t)shared * M * 3, clamp_mask, dcol, dsh,
                            dmean);
            } else {
                float tmp[SH_ROW];
                for (int k = 0; k < 3 * M; k++) tmp[k] = 0.f;
                sh_backward(cam.sh_degree, M, p_world, cam.campos, a.shs + (size_t)shared * M * 3, clamp_mask, dcol, tmp,
                            dmean);
                for (int k = 0; k < 3 * M; k++)
                    if (tmp[k] != 0.f) atomicAdd(dsh + k, tmp[k]);
            }
        }
        for (int k = 0; k < 3; k++) {
            a.dL_dmeans3D[3 * idx + k] = dmean[k];
            a.dL_dmeans2D[3 * idx + k] = o.dmean2D[k];
            a.dL_dcolors[3 * idx + k] = dcol[k];
        }
        for (int k = 0; k < 9; k++) a.dL_dtransMat[9 * idx + k] = o.dT[k];
        for (int k = 0; k < 4; k++) a.dL_drotations[4 * idx + k] = o.drot[k];
        if (!stacked) {
            a.dL_dopacity[shared] = acc[A_OPAC];
            a.dL_dscales[2 * shared] = o.dscale[0];
            a.dL_dscales[2 * shared + 1] = o.dscale[1];
        } else {
            atomicAdd(a.dL_dopacity + shared, acc[A_OPAC]);
            atomicAdd(a.dL_dscales + 2 * shared, o.dscale[0]);
            atomicAdd(a.dL_dscales + 2 * shared + 1, o.dscale[1]);
        }
    }
    if (SH_LDS) {
        if (visible && !has_sh)
            for (int k = 0; k < SH_ROW; k++) sh_row[k] = 0.f;
        __syncthreads();
        if (!stacked) {
            sh_tile_store(s_sh, a.dL_dsh, a.dL_dsh_dc, a.dL_dsh_rest, block_first, rows);
        } else {
            float* out = a.dL_dsh + (size_t)block_first * SH_ROW;
            for (int i = threadIdx.x; i < rows * SH_ROW; i += PRE_BLOCK) {
                const float v = s_sh[(i / SH_ROW) * SH_STRIDE + i % SH_ROW];
                if (v != 0.f) atomicAdd(out + i, v);
            }
        }
    }
}

// Stacked frames with the LDS-staged SH rows: one thread per CANONICAL surfel walks the frames, so that the gradients of
// what the frames share (opacity, scale, SH row) are summed in registers in frame order and stored once -- no atomics,
// no zero fill; the per-frame outputs are written as in the single-frame kernel.
__global__ __launch_bounds__(PRE_BLOCK) void preprocess_bwd_stacked_kernel(BackwardArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float s_sh[];  // [256][49]
    const int N = a.cam.frame_surfels, F = a.cam.frames;
    const int block_first = blockIdx.x * PRE_BLOCK;
    const int shared = block_first + threadIdx.x;
    const int rows = (N - block_first) < PRE_BLOCK ? (N - block_first) : PRE_BLOCK;
    sh_tile_load(s_sh, a.shs, a.sh_dc, a.sh_rest, block_first, rows);
    __syncthreads();
    const bool live = shared < N;
    float* sh_row = s_sh + threadIdx.x * SH_STRIDE;
    float g_sh[SH_ROW];
#pragma unroll
    for (int k = 0; k < SH_ROW; k++) g_sh[k] = 0.f;
    float g_opacity = 0.f, g_scale0 = 0.f, g_scale1 = 0.f;
    float2 sc = live ? reinterpret_cast<const float2*>(a.scales)[shared] : make_float2(0.f, 0.f);
    if (a.raw_params) sc = make_float2(act_scale(sc.x), act_scale(sc.y));
    const float scale[2] = {sc.x, sc.y};
    float opacity_act = 0.f;  // (the activated opacity, from the record of any frame that sees the surfel)
#pragma unroll 1
    for (int f = 0; f < F && live; f++) {
        const int idx = f * N + shared;
        if (!(a.radii[idx] > 0)) {
            for (int k = 0; k < 3; k++) {
                a.dL_dmeans3D[3 * idx + k] = 0.f;
                a.dL_dmeans2D[3 * idx + k] = 0.f;
                a.dL_dcolors[3 * idx + k] = 0.f;
            }
            for (int k = 0; k < 9; k++) a.dL_dtransMat[9 * idx + k] = 0.f;
            for (int k = 0; k < 4; k++) a.dL_drotations[4 * idx + k] = 0.f;
            continue;
        }
        const Camera cam = load_camera(a.cam, f);
        float acc[ACC_FLOATS];
        {
            // (Round 4 tried to clear the row here and let the caller reuse the zero-filled workspace without the zero-fill
            // launch: the 32 MB of stores cost this kernel 13 us, the launch they replace 9.)
            const float4* p = reinterpret_cast<const float4*>(a.acc + (size_t)idx * ACC_FLOATS);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const float4 v = p[k];
                acc[4 * k] = v.x;
                acc[4 * k + 1] = v.y;
                acc[4 * k + 2] = v.z;
                acc[4 * k + 3] = v.w;
            }
        }
        float T[9];
        uint32_t clamp_mask;
        {
            const float4* r = reinterpret_cast<const float4*>(a.geom.rec + (size_t)idx * REC_FLOATS);
            const float4 q0 = r[0], q1 = r[1], q2 = r[2], q4 = r[4];
            T[0] = q0.x; T[1] = q0.y; T[2] = q0.z; T[3] = q0.w;
            T[4] = q1.x; T[5] = q1.y; T[6] = q1.z; T[7] = q1.w;
            T[8] = q2.x;
            opacity_act = q2.w;
            clamp_mask = __float_as_uint(q4.w);
        }
        const float p_world[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
        const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[idx];
        const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
        SurfelGrads o;
        surfel_backward(cam, p_world, quat, scale, T, acc, o);
        float dmean[3] = {o.dmean3D[0], o.dmean3D[1], o.dmean3D[2]};
        const float dcol[3] = {acc[A_RGB], acc[A_RGB + 1], acc[A_RGB + 2]};
        float tmp[SH_ROW];
        sh_backward(cam.sh_degree, 16, p_world, cam.campos, sh_row, clamp_mask, dcol, tmp, dmean);
#pragma unroll
        for (int k = 0; k < SH_ROW; k++) g_sh[k] += tmp[k];
        g_opacity += acc[A_OPAC];
        g_scale0 += o.dscale[0];
        g_scale1 += o.dscale[1];
        for (int k = 0; k < 3; k++) {
            a.dL_dmeans3D[3 * idx + k] = dmean[k];
            a.dL_dmeans2D[3 * idx + k] = o.dmean2D[k];
            a.dL_dcolors[3 * idx + k] = dcol[k];
        }
        for (int k = 0; k < 9; k++) a.dL_dtransMat[9 * idx + k] = o.dT[k];
        for (int k = 0; k < 4; k++) a.dL_drotations[4 * idx + k] = o.drot[k];
    }
    if (live) {
        if (a.raw_params) {  // the frames' sum through exp / sigmoid (torch: grad * result; (grad * (1 - y)) * y)
            g_scale0 *= scale[0];
            g_scale1 *= scale[1];
            g_opacity = (g_opacity * (1.0f - opacity_act)) * opacity_act;
        }
        a.dL_dopacity[shared] = g_opacity;
        a.dL_dscales[2 * shared] = g_scale0;
        a.dL_dscales[2 * shared + 1] = g_scale1;
    }
    __syncthreads();  // (every thread is done reading the coefficients of the tile)
#pragma unroll
    for (int k = 0; k < SH_ROW; k++) sh_row[k] = g_sh[k];
    __syncthreads();
    sh_tile_store(s_sh, a.dL_dsh, a.dL_dsh_dc, a.dL_dsh_rest, block_first, rows);
}

void launch_preprocess_bwd(const BackwardArgs& a, hipStream_t stream)
{
    if (a.P <= 0) return;
    const bool sh_lds = a.cam.sh_coeffs == 16 &&
                        (a.sh_dc != nullptr ||  // (canonical pair: always this path; capi checked the alignment)
                         (a.shs != nullptr && a.dL_dsh != nullptr && (reinterpret_cast<uintptr_t>(a.shs) & 15) == 0 &&
                          (reinterpret_cast<uintptr_t>(a.dL_dsh) & 15) == 0));
    if (a.cam.frames > 1 && sh_lds) {
        hipLaunchKernelGGL(preprocess_bwd_stacked_kernel, dim3(pre_blocks(a.cam.frame_surfels)), dim3(PRE_BLOCK),
                           (size_t)PRE_BLOCK * SH_STRIDE * sizeof(float), stream, a);
        return;
    }
    const int blocks = a.cam.frames > 1 ? a.cam.frames * pre_blocks(a.cam.frame_surfels) : pre_blocks(a.P);
    if (sh_lds)
        hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3(blocks), dim3(PRE_BLOCK),
                           (size_t)PRE_BLOCK * SH_STRIDE * sizeof(float), stream, a);
    else
        hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3(blocks), dim3(PRE_BLOCK), 0, stream, a);
}

__global__ void mark_visible_kernel(int P, const float* means3D, const float* vm, uint8_t* present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    float v[3];
    float view[16];
    for (int k = 0; k < 16; k++) view[k] = vm[k];
    to_view(view, p, v);
    present[idx] = !(v[2] <= 0.2f);
}

void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, viewmatrix,
                       present);
}

}  // namespace surfel
