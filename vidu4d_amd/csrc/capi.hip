// capi.hip -- the extern "C" boundary declared in include/vidu4d_surfel.h.
// Argument validation, scratch carving and launch orchestration; no device code here.
//
// Orchestration mirrors CudaRasterizer::Rasterizer::forward / ::backward
// (/root/reference/gs/submodules/diff-surfel-rasterization/cuda_rasterizer/rasterizer_impl.cu:
// 198-342, :346-448) with the host synchronisation taken out of the middle.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/vidu4d_surfel.h"
#include "../../include/vidu4d_surfel_diag.h"
#include "surfel_state.h"

using namespace surfel;

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(VIDU4D_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- optional per-stage timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: zero overhead (one branch) per stage.
enum Stage { ST_PREPROCESS = 0, ST_SCAN, ST_EMIT, ST_SORT, ST_BLEND_FWD, ST_BWD_ZERO, ST_BLEND_BWD,
             ST_PREPROCESS_BWD, ST_COUNT };
static const char* const kStageNames[ST_COUNT] = {"preprocess_fwd", "tile_scan", "emit_keys", "tile_sort",
                                                   "blend_fwd", "bwd_zero", "blend_bwd", "preprocess_bwd"};
struct StageSpan {
    hipEvent_t a, b;
    int stage;
};
static bool g_prof_on = false;
static std::vector<StageSpan> g_spans;      // recorded, not yet harvested
static std::vector<StageSpan> g_span_pool;  // reusable event pairs
static double g_stage_ms[ST_COUNT];
static long long g_stage_n[ST_COUNT];

struct StageTimer {
    StageSpan sp;
    hipStream_t stream;
    bool on;
    StageTimer(int stage, hipStream_t s) : stream(s), on(g_prof_on)
    {
        if (!on) return;
        if (!g_span_pool.empty()) {
            sp = g_span_pool.back();
            g_span_pool.pop_back();
        } else {
            (void)hipEventCreate(&sp.a);
            (void)hipEventCreate(&sp.b);
        }
        sp.stage = stage;
        (void)hipEventRecord(sp.a, stream);
    }
    ~StageTimer()
    {
        if (!on) return;
        (void)hipEventRecord(sp.b, stream);
        g_spans.push_back(sp);
    }
};

extern "C" int vidu4d_surfel_profile_enable(int on)
{
    g_prof_on = on != 0;
    return VIDU4D_OK;
}
// Diagnostic (vidu4d_surfel_diag.h): a device-to-device copy held to `workgroups` workgroups of 512 threads -- the footprint
// of a collective's kernel (RCCL runs a fixed number of channels, one workgroup each) -- for measuring what such a tenant
// costs a compute kernel it shares the GPU with (tools/contention_probe.py).
__global__ __launch_bounds__(512) void diag_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16, int repeat)
{
    for (int r = 0; r < repeat; r++)
        for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) dst[i] = src[i];
}
extern "C" int vidu4d_diag_copy(void* dst, const void* src, size_t bytes, int workgroups, int repeat, void* stream)
{
    if (!dst || !src || workgroups <= 0 || repeat <= 0 || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15))
        return fail(VIDU4D_E_INVALID, "diag_copy: bad arguments");
    hipLaunchKernelGGL(diag_copy_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, (uint4*)dst, (const uint4*)src,
                       bytes / 16, repeat);
    return VIDU4D_OK;
}
extern "C" int vidu4d_surfel_profile_stage_count(void) { return ST_COUNT; }
extern "C" const char* vidu4d_surfel_profile_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }
// Waits for the recorded events, adds them to the per-stage totals and returns the totals
// (milliseconds, launch-span counts).  reset != 0 clears the totals afterwards.
extern "C" int vidu4d_surfel_profile_read(double* total_ms, long long* count, int reset)
{
    for (const StageSpan& sp : g_spans) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            g_stage_ms[sp.stage] += ms;
            g_stage_n[sp.stage] += 1;
        }
        g_span_pool.push_back(sp);
    }
    g_spans.clear();
    for (int i = 0; i < ST_COUNT; i++) {
        if (total_ms) total_ms[i] = g_stage_ms[i];
        if (count) count[i] = g_stage_n[i];
        if (reset) {
            g_stage_ms[i] = 0;
            g_stage_n[i] = 0;
        }
    }
    return VIDU4D_OK;
}

// debug != 0 reproduces the reference's CHECK_CUDA(.., debug): synchronise + check after a stage.
#define STAGE_CHECK(debug, stream, what)                                                                    \
    do {                                                                                                    \
        hipError_t e_ = hipGetLastError();                                                                  \
        if (e_ == hipSuccess && (debug)) e_ = hipStreamSynchronize(stream);                                 \
        if (e_ != hipSuccess) return fail(VIDU4D_E_HIP, "stage '%s' failed: %s", what, hipGetErrorString(e_)); \
    } while (0)

extern "C" int vidu4d_surfel_abi_version(void) { return VIDU4D_SURFEL_ABI; }
extern "C" const char* vidu4d_last_error(void) { return g_err; }

extern "C" size_t vidu4d_surfel_geom_bytes(int P)
{
    GeomState g;
    return carve_geom(nullptr, P < 0 ? 0 : P, g);
}
extern "C" size_t vidu4d_surfel_image_bytes(int width, int height)
{
    ImageState s;
    return carve_image(nullptr, width < 0 ? 0 : width, height < 0 ? 0 : height, s);
}
extern "C" size_t vidu4d_surfel_image_bytes_frames(int width, int height, int frames)
{
    ImageState s;
    return carve_image(nullptr, width < 0 ? 0 : width, height < 0 ? 0 : height, s, frames < 1 ? 1 : frames);
}
extern "C" size_t vidu4d_surfel_binning_bytes(int64_t capacity)
{
    BinState b;
    return carve_binning(nullptr, capacity, b);
}
extern "C" size_t vidu4d_surfel_backward_workspace_bytes(int P)
{
    return (size_t)(P < 0 ? 0 : P) * ACC_FLOATS * sizeof(float) + 256;
}

// frames of a (possibly stacked) call, total surfel rows, camera with the per-frame views
template <typename Args>
static int frames_of(const Args* a) { return a->frames > 1 ? a->frames : 1; }
template <typename Args>
static CameraParams camera_of(const Args* a)
{
    const int F = frames_of(a);
    CameraParams c = make_camera_params(F > 1 ? a->frame_viewmatrix[0] : a->viewmatrix, F > 1 ? a->frame_campos[0] : a->campos,
                                        a->width, a->height, F > 1 ? a->frame_tan_fovx[0] : a->tan_fovx,
                                        F > 1 ? a->frame_tan_fovy[0] : a->tan_fovy, a->D, a->M);
    if (F > 1) {
        c.frames = F;
        c.frame_surfels = a->P;
        for (int f = 0; f < F; f++)
            set_frame_camera(c, f, a->frame_viewmatrix[f], a->frame_campos[f], a->frame_tan_fovx[f], a->frame_tan_fovy[f]);
    }
    return c;
}
template <typename Args>
static int check_frames(const Args* a)
{
    if (a->frames <= 1) return VIDU4D_OK;
    if (a->frames > MAX_STACKED_FRAMES) return fail(VIDU4D_E_INVALID, "at most %d stacked frames", MAX_STACKED_FRAMES);
    if ((int64_t)a->P * a->frames > 0x7fffffffll) return fail(VIDU4D_E_INVALID, "frames * P exceeds 2^31");
    for (int f = 0; f < a->frames; f++)
        if (!a->frame_viewmatrix[f] || !a->frame_campos[f] || !(a->frame_tan_fovx[f] > 0.f) || !(a->frame_tan_fovy[f] > 0.f))
            return fail(VIDU4D_E_INVALID, "camera of stacked frame %d is incomplete", f);
    return VIDU4D_OK;
}

// The canonical SH pair (sh_dc (P,1,3) + sh_rest (P,15,3)) replaces `shs`; it only exists on the LDS-staged kernels.
template <typename Args>
static int check_canonical_sh(const Args* a)
{
    if (!a->sh_dc && !a->sh_rest) return VIDU4D_OK;
    if (!a->sh_dc || !a->sh_rest) return fail(VIDU4D_E_INVALID, "sh_dc and sh_rest come together");
    if (a->shs) return fail(VIDU4D_E_INVALID, "sh_dc / sh_rest replace shs: pass shs = NULL");
    if (a->M != 16) return fail(VIDU4D_E_INVALID, "sh_dc / sh_rest need M = 16 (1 + 15 coefficients), got %d", a->M);
    if (((uintptr_t)a->sh_dc & 15) || ((uintptr_t)a->sh_rest & 15))
        return fail(VIDU4D_E_INVALID, "sh_dc / sh_rest must be 16-byte aligned");
    return VIDU4D_OK;
}

// aux_planes -> the blend instance that carries exactly what the caller reads (surfel_math.h)
static int blend_mode(int aux_planes)
{
    if (aux_planes == VIDU4D_AUX_ALPHA) return BLEND_LITE;
    if (aux_planes != 0 && (aux_planes & ~VIDU4D_AUX_GEOM) == 0) return BLEND_GEOM;  // some of the planes 0-4, nothing else
    return BLEND_FULL;
}
static int kernel_flags(int debug_flags)
{
    return ((debug_flags & VIDU4D_DEBUG_NO_CULL) ? FLAG_NO_CULL : 0) | ((debug_flags & VIDU4D_DEBUG_SERIAL_REPAIR) ? FLAG_SERIAL_REPAIR : 0) |
           ((debug_flags & VIDU4D_DEBUG_POSITION_ORDER) ? FLAG_POSITION_ORDER : 0) | (debug_flags & (15 << FLAG_PAIR_SHIFT));
}
// Does a whole-tile forward leave recorded segments for its backward (surfel_state.h)?  They pay by letting the dispatcher
// balance the CUs when a launch has about as many tiles as the chip has workgroup slots (256 CUs x 6: the headline's 2048
// tiles drain for a third of the launch); with several tiles per slot the whole-tile launch balances by itself and the
// records only cost (1080p, 16 320 tiles in two frames: blend_fwd +4 %, blend_bwd +1 %).
static bool records_segments(int segment_split, int tiles) { return segment_split == 0 && tiles <= REC_MAX_TILES; }

static int check_forward(const Vidu4dSurfelForwardArgs* a)
{
    if (!a) return fail(VIDU4D_E_INVALID, "args is NULL");
    if (a->P < 0 || a->width <= 0 || a->height <= 0) return fail(VIDU4D_E_INVALID, "bad sizes P=%d W=%d H=%d", a->P, a->width, a->height);
    if (a->transMat_precomp)
        return fail(VIDU4D_E_UNSUPPORTED,
                    "cov3D_precomp/transMat_precomp is not supported: the reference path leaves the normal "
                    "uninitialised (forward.cu:214-224) and its backward dereferences scales/rotations anyway");
    if (int rc = check_canonical_sh(a)) return rc;
    const bool has_sh = a->shs != nullptr || a->sh_dc != nullptr;
    if (has_sh == (a->colors_precomp != nullptr) && a->P > 0)
        return fail(VIDU4D_E_INVALID, "provide exactly one of shs / colors_precomp");
    if (has_sh && (a->M <= 0 || a->M > 16 || a->D < 0 || a->D > 3 || (a->D + 1) * (a->D + 1) > a->M))
        return fail(VIDU4D_E_INVALID, "bad SH configuration D=%d M=%d", a->D, a->M);
    const int F = frames_of(a);
    if (int rc = check_frames(a)) return rc;
    if (F == 1 && (!(a->tan_fovx > 0.f) || !(a->tan_fovy > 0.f))) return fail(VIDU4D_E_INVALID, "tan_fov must be > 0");
    if (!a->out_color || !a->out_others || !a->background) return fail(VIDU4D_E_INVALID, "output/background pointer is NULL");
    if (a->P > 0 && (!a->means3D || !a->opacities || !a->scales || !a->rotations || !a->radii ||
                     (F == 1 && (!a->viewmatrix || !a->campos))))
        return fail(VIDU4D_E_INVALID, "an input pointer is NULL");
    if (!a->geom_buffer || a->geom_bytes < vidu4d_surfel_geom_bytes(a->P * F)) return fail(VIDU4D_E_BUFFER, "geom_buffer too small");
    if (!a->image_buffer || a->image_bytes < vidu4d_surfel_image_bytes_frames(a->width, a->height, F))
        return fail(VIDU4D_E_BUFFER, "image_buffer too small");
    return VIDU4D_OK;
}

// Zero fill as a kernel rather than hipMemsetAsync: a memset node captured into a hipGraph is not
// ordered with the neighbouring kernel nodes on replay in this ROCm build (the backward's accumulator
// came back partly unzeroed; tools/hipgraph_replay_check.py) -- and a kernel costs the same.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* p, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ __launch_bounds__(256) void zero_fill_f32_kernel(float* p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}
static void zero_fill_f32(float* p, size_t n, hipStream_t stream)  // any alignment / count
{
    if (n) hipLaunchKernelGGL(zero_fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n);
}
static void zero_fill(void* p, size_t bytes, hipStream_t stream)  // p 16-byte aligned, bytes a multiple of 16
{
    const size_t n16 = bytes / 16;
    if (n16)
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream, (uint4*)p, n16);
}

extern "C" int vidu4d_surfel_forward_plan(const Vidu4dSurfelForwardArgs* a, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();  // a stale error of an unrelated earlier HIP call must not be blamed on us
    int rc = check_forward(a);
    if (rc) return rc;
    GeomState g;
    ImageState img;
    const int F = frames_of(a), P = a->P * F;  // (stacked frames: F * P surfel rows, F tile grids)
    carve_geom((char*)a->geom_buffer, P, g);
    carve_image((char*)a->image_buffer, a->width, a->height, img, F);
    PreprocessArgs pa;
    pa.cam = camera_of(a);
    pa.P = P;
    pa.means3D = a->means3D;
    pa.scales = a->scales;
    pa.rotations = a->rotations;
    pa.opacities = a->opacities;
    pa.shs = a->shs;
    pa.colors_precomp = a->colors_precomp;
    pa.sh_dc = a->sh_dc;
    pa.sh_rest = a->sh_rest;
    pa.raw_params = a->raw_params;
    pa.radii = a->radii;
    pa.geom = g;
    pa.tile_count = img.tile_count;
    pa.group_counts = img.group_counts;
    pa.iters = bin_iters(P);
    pa.stage_records = 0;  // (launch_preprocess_fwd decides)
    const int num_tiles = total_tiles(pa.cam);
    const bool grouped = use_grouped_binning(num_tiles);
    {
        StageTimer t(ST_PREPROCESS, stream);
        if (!grouped)
            zero_fill(img.tile_count, (size_t)num_tiles * TILE_SLICES * sizeof(uint32_t), stream);
        launch_preprocess_fwd(pa, stream);
    }
    STAGE_CHECK(a->debug, stream, "preprocess");
    {
        StageTimer t(ST_SCAN, stream);
        launch_tile_scan(g, img, num_tiles, grouped ? bin_groups(P) : 0, stream);
    }
    STAGE_CHECK(a->debug, stream, "tile_scan");
    return VIDU4D_OK;
}

extern "C" int vidu4d_surfel_num_rendered(const Vidu4dSurfelForwardArgs* a, void* stream_, int64_t* out)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!a || !out || !a->geom_buffer) return fail(VIDU4D_E_INVALID, "NULL argument");
    GeomState g;
    carve_geom((char*)a->geom_buffer, a->P * frames_of(a), g);
    uint32_t r = 0;
    HIP_TRY(hipMemcpyAsync(&r, &g.hdr->num_rendered, sizeof(r), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    *out = (int64_t)r;
    return VIDU4D_OK;
}

extern "C" int vidu4d_surfel_forward_run(const Vidu4dSurfelForwardArgs* a, void* binning, size_t binning_bytes,
                                         int64_t capacity, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();  // a stale error of an unrelated earlier HIP call must not be blamed on us
    int rc = check_forward(a);
    if (rc) return rc;
    if (capacity < 0) return fail(VIDU4D_E_INVALID, "negative capacity");
    if (capacity > 0 && (!binning || binning_bytes < vidu4d_surfel_binning_bytes(capacity)))
        return fail(VIDU4D_E_BUFFER, "binning_buffer too small for capacity %lld", (long long)capacity);
    if (capacity > 0xFFFFFFFFll) return fail(VIDU4D_E_INVALID, "capacity exceeds the 32-bit pair index of the reference");
    GeomState g;
    ImageState img;
    BinState b;
    const int F = frames_of(a), P = a->P * F;
    carve_geom((char*)a->geom_buffer, P, g);
    carve_image((char*)a->image_buffer, a->width, a->height, img, F);
    carve_binning((char*)binning, capacity, b);
    const CameraParams cam = camera_of(a);
    // the segment table: for a segment-parallel forward, or (whole-tile forward) for the recorded segments of its backward
    const bool record = records_segments(a->segment_split, total_tiles(cam));
    // (by length CLASS in both cases since round 5: the split tiles are then exactly the schedule positions [0, S), and the
    // schedule builder also orders the tails -- which the backward of a segment-parallel forward now dispatches by too)
    // (VIDU4D_SCHED_XCD_BLOCK(B) in debug_flags: the XCD-local longest-first schedule over BxB-tile blocks, binning.hip)
    const ScheduleParams sp = {record ? REC_SEG_LEN : SEG_LEN, record ? REC_MIN : SPLIT_MIN, 1,
                               (a->debug_flags >> FLAG_XCD_SHIFT) & 15, cam.grid_x, cam.grid_x * cam.grid_y,
                               (a->segment_split == 0 && !((a->debug_flags >> FLAG_XCD_SHIFT) & 15)) ? (a->debug_flags >> FLAG_PAIR_SHIFT) & 15 : 0};
    {
        StageTimer t(ST_EMIT, stream);
        launch_emit_keys(cam, P, a->radii, g, img, b, capacity, use_grouped_binning(total_tiles(cam)), sp, stream);
    }
    STAGE_CHECK(a->debug, stream, "emit_keys");
    if (capacity > 0) {
        {
            StageTimer t(ST_SORT, stream);
            launch_tile_sort(g, img, b, total_tiles(cam), P, capacity,
                             a->segment_split == 0 ? (a->max_list_hint > 0 && a->max_list_hint <= SHORT_LIST_HINT_MAX
                                                          ? LongListSort::short_lists_expected : LongListSort::in_lds_only)
                             : a->long_list_sort    ? LongListSort::one_workgroup
                                                    : LongListSort::msd_split,
                             stream);
        }
        STAGE_CHECK(a->debug, stream, "tile_sort");
    }
    {
        StageTimer t(ST_BLEND_FWD, stream);
        // segment_split: 0 off, 1 on, k > 1: on, at most k segments per tile (Header::truncated reports a miss)
        const int max_seg = a->segment_split > 1 ? a->segment_split : 0x7fffffff;
        launch_blend_fwd(cam, g, img, b, capacity, a->segment_split != 0, max_seg, a->background, a->out_color, a->out_others,
                         a->depth_used, blend_mode(a->aux_planes), a->assume_unsaturated != 0, kernel_flags(a->debug_flags),
                         record, stream);
    }
    STAGE_CHECK(a->debug, stream, "blend_forward");
    return VIDU4D_OK;
}

extern "C" int vidu4d_surfel_backward(const Vidu4dSurfelBackwardArgs* a, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();  // a stale error of an unrelated earlier HIP call must not be blamed on us
    if (!a) return fail(VIDU4D_E_INVALID, "args is NULL");
    if (a->P < 0 || a->width <= 0 || a->height <= 0) return fail(VIDU4D_E_INVALID, "bad sizes");
    if (a->transMat_precomp) return fail(VIDU4D_E_UNSUPPORTED, "transMat_precomp is not supported (see forward)");
    if (a->P == 0) return VIDU4D_OK;
    const int F = frames_of(a), P = a->P * F;
    if (int rc = check_frames(a)) return rc;
    if (!a->geom_buffer || !a->image_buffer || !a->workspace || !a->dL_dout_color || !a->dL_dout_others ||
        !a->means3D || !a->radii || !a->scales || !a->rotations || (F == 1 && (!a->viewmatrix || !a->campos)) || !a->background)
        return fail(VIDU4D_E_INVALID, "an input pointer is NULL");
    if (!a->dL_dmeans2D || !a->dL_dcolors || !a->dL_dopacity || !a->dL_dmeans3D || !a->dL_dtransMat ||
        !a->dL_dscales || !a->dL_drotations || (a->shs && !a->dL_dsh))
        return fail(VIDU4D_E_INVALID, "an output pointer is NULL");
    if (int rc = check_canonical_sh(a)) return rc;
    if (a->sh_dc && (!a->dL_dsh_dc || !a->dL_dsh_rest || ((uintptr_t)a->dL_dsh_dc & 15) || ((uintptr_t)a->dL_dsh_rest & 15)))
        return fail(VIDU4D_E_INVALID, "sh_dc / sh_rest need 16-byte aligned dL_dsh_dc / dL_dsh_rest");
    if (a->workspace_bytes < vidu4d_surfel_backward_workspace_bytes(P)) return fail(VIDU4D_E_BUFFER, "workspace too small");
    if (a->binning_capacity > 0 && !a->binning_buffer) return fail(VIDU4D_E_INVALID, "binning_buffer is NULL");

    BackwardArgs ba;
    ba.cam = camera_of(a);
    BinState b;
    carve_geom((char*)a->geom_buffer, P, ba.geom);
    carve_image((char*)a->image_buffer, a->width, a->height, ba.img, F);
    carve_binning((char*)a->binning_buffer, a->binning_capacity, b);
    ba.point_list = a->binning_capacity > 0 ? b.point_list : nullptr;
    ba.seg_data = a->binning_capacity > 0 ? b.seg_data : nullptr;
    ba.capacity = a->binning_capacity;
    // segment_split == 0: the forward walked whole tiles and (unless told not to) left recorded segments
    ba.recorded = records_segments(a->segment_split, total_tiles(ba.cam));
    ba.split = a->binning_capacity > 0 && (a->segment_split != 0 ||
                                           (ba.recorded && !(a->debug_flags & VIDU4D_DEBUG_WHOLE_TILE_BACKWARD)));
    ba.max_seg = a->segment_split > 1 ? a->segment_split : 0x7fffffff;
    ba.P = P;
    ba.background = a->background;
    ba.means3D = a->means3D;
    ba.radii = a->radii;
    ba.shs = a->shs;
    ba.colors_precomp = a->colors_precomp;
    ba.scales = a->scales;
    ba.rotations = a->rotations;
    ba.dL_dcolor = a->dL_dout_color;
    ba.dL_dothers = a->dL_dout_others;
    uintptr_t ws = ((uintptr_t)a->workspace + 255) & ~(uintptr_t)255;
    ba.acc = (float*)ws;
    ba.dL_dmeans2D = a->dL_dmeans2D;
    ba.dL_dcolors = a->dL_dcolors;
    ba.dL_dopacity = a->dL_dopacity;
    ba.dL_dmeans3D = a->dL_dmeans3D;
    ba.dL_dtransMat = a->dL_dtransMat;
    ba.dL_dsh = a->dL_dsh;
    ba.dL_dscales = a->dL_dscales;
    ba.dL_drotations = a->dL_drotations;
    ba.sh_dc = a->sh_dc;
    ba.sh_rest = a->sh_rest;
    ba.dL_dsh_dc = a->sh_dc ? a->dL_dsh_dc : nullptr;
    ba.dL_dsh_rest = a->sh_dc ? a->dL_dsh_rest : nullptr;
    ba.raw_params = a->raw_params;
    ba.mode = blend_mode(a->aux_planes);
    ba.flags = kernel_flags(a->debug_flags);

    {
        StageTimer t(ST_BWD_ZERO, stream);
        launch_bwd_prepare(ba, (size_t)P * ACC_FLOATS * sizeof(float), stream);  // (zeroes the accumulator)
        const bool sums_in_kernel = a->M == 16 && (a->sh_dc || (a->shs && a->dL_dsh && ((uintptr_t)a->shs & 15) == 0 &&
                                                               ((uintptr_t)a->dL_dsh & 15) == 0));  // (preprocess_bwd_stacked_kernel's condition)
        if (F > 1 && !sums_in_kernel) {  // the frames ADD their gradients of the shared parameters (preprocess_bwd_kernel)
            zero_fill_f32(a->dL_dopacity, (size_t)a->P, stream);
            zero_fill_f32(a->dL_dscales, (size_t)a->P * 2, stream);
            if (a->dL_dsh) zero_fill_f32(a->dL_dsh, (size_t)a->P * a->M * 3, stream);
        }
    }
    {
        StageTimer t(ST_BLEND_BWD, stream);
#ifdef SURFEL_BWD_TRACE   // (variant build for tools/bwd_trace.py: diag_walk_counters is blend_bwd's own trace buffer)
        ba.trace = (unsigned long long*)a->diag_walk_counters;
#endif
        launch_blend_bwd(ba, stream);
    }
#ifndef SURFEL_BWD_TRACE
    if (a->diag_walk_counters && ba.point_list)
        launch_blend_bwd_stats(ba, (unsigned long long*)a->diag_walk_counters, stream);  // (vidu4d_surfel_diag.h)
#endif
    STAGE_CHECK(a->debug, stream, "blend_backward");
    {
        StageTimer t(ST_PREPROCESS_BWD, stream);
        launch_preprocess_bwd(ba, stream);
    }
    STAGE_CHECK(a->debug, stream, "preprocess_backward");
    return VIDU4D_OK;
}

extern "C" int vidu4d_surfel_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                                          uint8_t* present, void* stream_)
{
    (void)projmatrix;
    if (P < 0) return fail(VIDU4D_E_INVALID, "negative P");
    if (P == 0) return VIDU4D_OK;
    if (!means3D || !viewmatrix || !present) return fail(VIDU4D_E_INVALID, "NULL pointer");
    (void)hipGetLastError();
    launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream_);
    STAGE_CHECK(0, (hipStream_t)stream_, "mark_visible");
    return VIDU4D_OK;
}

extern "C" int vidu4d_surfel_state_read(const Vidu4dSurfelForwardArgs* a, const void* binning, int64_t capacity,
                                        int what, void* dst, size_t dst_bytes, int64_t* count, void* stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!a || !dst || !count) return fail(VIDU4D_E_INVALID, "NULL argument");
    GeomState g;
    ImageState img;
    BinState b;
    const int F = frames_of(a);  // (stacked frames: every array is F times as long, frame-major)
    const size_t Ptot = (size_t)a->P * F;
    carve_geom((char*)a->geom_buffer, (int)Ptot, g);
    carve_image((char*)a->image_buffer, a->width, a->height, img, F);
    carve_binning((char*)binning, capacity, b);
    const int gx = (a->width + TILE - 1) / TILE, gy = ((a->height + TILE - 1) / TILE) * F;
    uint32_t R = 0;
    HIP_TRY(hipMemcpyAsync(&R, &g.hdr->num_rendered, sizeof(R), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const size_t hw = (size_t)a->width * a->height * F;
    const void* src = nullptr;
    size_t n = 0, esz = 4;
    switch (what) {
        case VIDU4D_STATE_NUM_RENDERED: src = &g.hdr->num_rendered; n = 1; break;
        case VIDU4D_STATE_RECORDS: src = g.rec; n = Ptot * REC_FLOATS; break;
        case VIDU4D_STATE_TILES_TOUCHED: src = g.tiles_touched; n = Ptot; break;
        case VIDU4D_STATE_POINT_LIST: src = b.point_list; n = R; break;
        case VIDU4D_STATE_SORTED_KEYS: src = b.entries; n = R; esz = 8; break;
        case VIDU4D_STATE_RANGES: src = img.ranges; n = (size_t)gx * gy * 2; break;
        case VIDU4D_STATE_FINAL_T: src = img.final_T; n = 3 * hw; break;
        case VIDU4D_STATE_N_CONTRIB: src = img.n_contrib; n = 2 * hw; break;
        case VIDU4D_STATE_TILE_ORDER: src = img.tile_order; n = (size_t)gx * gy; break;
        case VIDU4D_STATE_TAIL_ORDER: src = img.tail_order; n = (size_t)gx * gy; break;
        case VIDU4D_STATE_HEADER: src = g.hdr; n = sizeof(Header) / 4; break;
        default: return fail(VIDU4D_E_INVALID, "unknown state array %d", what);
    }
    if ((what == VIDU4D_STATE_POINT_LIST || what == VIDU4D_STATE_SORTED_KEYS) && (int64_t)R > capacity)
        return fail(VIDU4D_E_BUFFER, "num_rendered %u exceeds capacity %lld", R, (long long)capacity);
    *count = (int64_t)n;
    if (n * esz > dst_bytes) return fail(VIDU4D_E_BUFFER, "dst too small: need %zu bytes", n * esz);
    if (n) HIP_TRY(hipMemcpyAsync(dst, src, n * esz, hipMemcpyDefault, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (what == VIDU4D_STATE_SORTED_KEYS && n) {
        // stored per tile as (depth bits << 32 | id); hand back the reference's (tile << 32 | depth bits)
        std::vector<uint32_t> rng((size_t)gx * gy * 2);
        HIP_TRY(hipMemcpy(rng.data(), img.ranges, rng.size() * 4, hipMemcpyDeviceToHost));
        uint64_t* k = (uint64_t*)dst;  // dst must be host memory for this array
        for (size_t t = 0; t < (size_t)gx * gy; t++)
            for (uint32_t i = rng[2 * t]; i < rng[2 * t + 1]; i++) k[i] = ((uint64_t)t << 32) | (k[i] >> 32);
    }
    return VIDU4D_OK;
}
