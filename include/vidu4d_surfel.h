/*
 * vidu4d_surfel.h -- C ABI of the MI355X-native Gaussian-surfel rasterizer (libvidu4d_surfel.so).
 *
 * This is the drop-in boundary for the one hot path of yikaiw/Vidu4D Stage-3: every entry point
 * replaces one function of the reference's native extension `diff_surfel_rasterization._C`
 * (pybind module, /root/reference/gs/submodules/diff-surfel-rasterization/ext.cpp:15-19) or of
 * `lab4d/third_party/quaternion` (quaternion.cu), with torch::Tensor arguments restated as plain
 * device pointers + sizes.  No torch / HIP types appear in the signatures: `stream` is a
 * hipStream_t passed as void* (NULL = the legacy default stream, which is what the reference's
 * `<<<grid, block>>>` launches use).
 *
 * Ownership (same as the reference, rasterize_points.cu:87-103, :194-202): the CALLER allocates all
 * outputs and the three opaque scratch buffers (geometry / image / binning state) and keeps them
 * alive between forward and backward; this library never allocates device memory per call.  The
 * binning buffer's size depends on num_rendered, which is only known after the per-surfel pass --
 * the reference solves this with a std::function<char*(size_t)> resize callback and a blocking
 * cudaMemcpy (rasterizer_impl.cu:282-286).  Here the forward is split in two calls instead:
 *
 *     vidu4d_surfel_forward_plan(args, stream)          preprocess + per-tile count scan (tile ranges)
 *     vidu4d_surfel_num_rendered(args, stream, &R)      (optional) blocking read of num_rendered
 *     vidu4d_surfel_forward_run(args, binning, cap,...) pair emit + in-tile radix sort + blend
 *
 * A caller that wants the reference's exact behaviour calls all three (one host sync, exact
 * buffer).  A caller that wants no host sync passes a capacity guess straight to _run: every
 * kernel guards on the device-side count, and `vidu4d_surfel_num_rendered` later tells whether
 * the guess was large enough (if R > capacity nothing was rendered and _run must be repeated).
 *
 * All functions return VIDU4D_OK (0) or a negative error code; vidu4d_last_error() returns a
 * thread-local message for the last failure.  Shape errors are reported the way the reference's
 * AT_ERROR checks are (rasterize_points.cu:61-71): the host wrapper raises RuntimeError.
 *
 * Layout of all tensors is the reference's: fp32, C-contiguous; means3D (P,3); scales (P,2);
 * rotations (P,4) w-first; opacities (P,1); shs (P,M,3); out_color (3,H,W); out_others (8,H,W) with
 * planes {0 depth, 1 alpha, 2-4 normal, 5 median depth, 6 distortion, 7 median weight}
 * (auxiliary.h:25-30); radii (P) int32; viewmatrix/projmatrix 16 floats in the row-vector
 * convention the reference passes (viewmatrix = W^T).
 */
#ifndef VIDU4D_SURFEL_H_INCLUDED
#define VIDU4D_SURFEL_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIDU4D_OK 0
#define VIDU4D_E_INVALID (-1)      /* bad argument (NULL pointer, negative size, unsupported mode) */
#define VIDU4D_E_BUFFER (-2)       /* a scratch buffer is too small */
#define VIDU4D_E_HIP (-3)          /* a HIP runtime call failed (message has hipGetErrorString) */
#define VIDU4D_E_UNSUPPORTED (-4)  /* reference feature that is undefined upstream (see message) */

/* ABI version of this header; bumped on any struct change. */
#define VIDU4D_SURFEL_ABI 21
int vidu4d_surfel_abi_version(void);
const char* vidu4d_last_error(void);

/* ---- scratch sizing: replaces required<GeometryState/ImageState/BinningState>()
 *      (cuda_rasterizer/rasterizer_impl.h:64-71, rasterizer_impl.cu:155-194) ---- */
size_t vidu4d_surfel_geom_bytes(int P);
size_t vidu4d_surfel_image_bytes(int width, int height);
size_t vidu4d_surfel_binning_bytes(int64_t capacity /* max (surfel,tile) pairs */);
/* bytes of the fp32 gradient accumulator the backward needs (zero-filled by the backward itself) */
size_t vidu4d_surfel_backward_workspace_bytes(int P);

/* ---- forward: replaces RasterizeGaussiansCUDA / CudaRasterizer::Rasterizer::forward
 *      (rasterize_points.cu:39-141, rasterizer_impl.cu:198-342) ---- */
typedef struct Vidu4dSurfelForwardArgs {
    int P;           /* number of surfels */
    int D;           /* active SH degree (0..3) */
    int M;           /* SH coefficients per surfel in `shs` (0 when colors_precomp is used) */
    int width, height;
    float tan_fovx, tan_fovy;
    float scale_modifier;            /* accepted and ignored, as the reference does (forward.cu:95) */
    int prefiltered;                 /* accepted; culled surfels are skipped either way */
    int debug;                       /* !=0: synchronise and check after every launch (CHECK_CUDA) */
    const float* background;         /* (3) */
    const float* means3D;            /* (P,3) */
    const float* shs;                /* (P,M,3) or NULL */
    const float* colors_precomp;     /* (P,3) or NULL; exactly one of shs / colors_precomp */
    const float* opacities;          /* (P,1) */
    const float* scales;             /* (P,2) */
    const float* rotations;          /* (P,4) */
    const float* transMat_precomp;   /* must be NULL: upstream path is undefined (forward.cu:214-224) */
    const float* viewmatrix;         /* (4,4) */
    const float* projmatrix;         /* (4,4) accepted; only feeds a value upstream discards */
    const float* campos;             /* (3) */
    float* out_color;                /* (3,H,W) */
    float* out_others;               /* (8,H,W) */
    int32_t* radii;                  /* (P) */
    void* geom_buffer;               /* >= vidu4d_surfel_geom_bytes(P) */
    size_t geom_bytes;
    void* image_buffer;              /* >= vidu4d_surfel_image_bytes(W,H) */
    size_t image_bytes;
    int segment_split;               /* !=0: tiles whose list exceeds 1024 entries are blended segment-parallel
                                        (512-entry segments on separate workgroups; three launches instead of
                                        one).  Pays off when few tiles hold most pairs (object-centric frames);
                                        results agree with the unsplit blend to fp32 re-association.
                                        k > 1: additionally, only the first k segments of a tile are blended (a
                                        caller that knows how deep earlier frames went saves the rest of the
                                        transmittance pass); if some pixel was still unsaturated after them the
                                        frame is incomplete and word 6 of the geometry buffer (`truncated`) is
                                        set: blend it again with segment_split = 1. */
    uint32_t* depth_used;            /* optional device counter (or NULL): atomic max of the deepest list position
                                        any pixel of the frame blended, i.e. the serial chain length of the
                                        unsplit blend.  Callers use it to decide segment_split for later frames.  Word 12 of
                                        the geometry buffer is zeroed by every forward for this purpose: a caller that reads
                                        the buffer's first words back anyway may pass geom_buffer + 48. */
    /* ---- stacked frames (the reference renders the frames of a step one after the other,
     * lab4d/nnutils/deformable_gaussian.py:1175-1228).  frames > 1: ONE launch set rasterizes `frames` frames that
     * share opacities / scales / shs (P rows) but have their own centres and orientations -- means3D (frames,P,3),
     * rotations (frames,P,4), radii (frames,P) -- and cameras: frame f uses frame_viewmatrix[f], frame_campos[f],
     * frame_tan_fovx/y[f] (viewmatrix / campos / tan_fov* above are ignored).  Outputs are plane-major over the frames:
     * out_color (3,frames,H,W), out_others (8,frames,H,W).  geom_buffer must hold vidu4d_surfel_geom_bytes(frames * P),
     * image_buffer vidu4d_surfel_image_bytes_frames(W,H,frames).  Every per-frame result equals what a single-frame
     * call gives for that frame (same tile lists, same blend order).  colors_precomp, if used, is (frames,P,3).
     * frames <= 1: the fields below are ignored. */
    int frames;
    const float* frame_viewmatrix[8];
    const float* frame_campos[8];
    float frame_tan_fovx[8];
    float frame_tan_fovy[8];
    /* ---- canonical parameters as the optimizer holds them (extension).  Upstream activates and concatenates them in
     * torch before every rasterizer call (gs/scene/gaussian_model.py:98-118: exp of `_scaling`, sigmoid of `_opacity`,
     * cat of `_features_dc` / `_features_rest`) and differentiates back through those launches.
     * sh_dc (P,1,3) + sh_rest (P,15,3), both non-NULL and 16-byte aligned: used instead of `shs` (which must then be
     * NULL; M must be 16).  raw_params != 0: `scales` holds LOG-scales and `opacities` LOGITS; exp(x) and
     * 1 / (1 + exp(-x)) are applied by the per-surfel kernels. */
    const float* sh_dc;
    const float* sh_rest;
    int raw_params;
    /* ---- auxiliary planes the caller will read (extension).  Bit i = plane i of out_others; 0 means all (0xFF).
     * When only the alpha plane is named (aux_planes == VIDU4D_AUX_ALPHA: colour + silhouette losses, the Stage-3 loop
     * until its regularisers switch on, lab4d/engine/model.py:895-1012 with lambda_dist = lambda_normal = 0) the blend
     * carries colour, transmittance and the contributor count only: out_color and plane 1 are what the full blend
     * gives, bit for bit; the other planes come out as zeros, and the state kept for the backward holds no distortion
     * moments and no median contributor.  When only planes 0-4 are named (any non-empty subset of VIDU4D_AUX_GEOM other
     * than the alpha plane alone: depth, alpha, normal -- what the Stage-3 loop reads after step 8000 with the upstream
     * defaults lambda_dist = 0 and depth_ratio = 0, lab4d/config.py:181, gs/arguments/__init__.py:68,
     * lab4d/engine/model.py:817-842) the median sample and the distortion moments are not carried: out_color and planes
     * 0-4 are the full blend's bit for bit, planes 5-7 come out as zeros.  Any other value: everything is computed.
     * ("Bit for bit" compares the instances of ONE call configuration: same library build, same segment_split, same image
     * size and frame count.  A whole-tile forward that leaves recorded segments -- segment_split == 0 and at most 4096 tiles
     * in the launch, stacked frames counted -- sums colour, depth and normal of a tile longer than 320 entries in two levels,
     * per 256-entry segment and then across segments, in every instance alike; beyond 4096 tiles (-DSURFEL_REC_MAX_TILES, a
     * multiple of 256: a build with 256 is the A/B switch for any real image) it keeps ONE running sum per pixel as the
     * reference does, forward.cu:400-438.  The two differ in the last bits, so results do change in the last bits with the
     * image size / the number of stacked frames across that boundary, and against releases before ABI 17.) */
    int aux_planes;
    /* ---- segment-parallel blend without its transmittance pre-pass (extension; only with segment_split != 0 and
     * aux_planes naming nothing beyond planes 0-4, ignored otherwise: the full blend's median sample depends on the exact
     * start transmittance).  != 0: every segment is blended from T = 1 and scaled by the
     * product of its predecessors in the combine pass (colour is linear in the start transmittance; while a pixel stays
     * clear of the saturation threshold no decision depends on it), and the combine pass blends the one segment in which a
     * pixel comes within 0.1 % of the threshold (transmittance 1e-4) again, in list order, from the exact start: the
     * pixel's last contributor and final transmittance are those of the exact blend, its colour equals it up to fp32
     * re-association of the segments in front.  (The field is named after rounds 1-2, when such a frame was only reported --
     * word 6 of the geometry buffer, `truncated` -- and had to be blended again with assume_unsaturated = 0.)  Word 8 of
     * the geometry buffer holds the bits of the frame's smallest final transmittance (whatever the mode). */
    int assume_unsaturated;
    /* ---- how tile lists beyond the LDS capacity are sorted (extension; only with segment_split != 0; the sorted list is
     * the same either way).  0: MSD split on the leading differing depth bits + in-LDS bucket sorts (five small launches:
     * pays off from ~10 k entries per list, object-centric frames); 1: one 16-wave workgroup per long list through global
     * memory (two launches: cheaper while the longest lists hold a few thousand entries).  Callers decide from word 2 of the
     * geometry buffer (the longest list of earlier frames). */
    int long_list_sort;
    /* ---- debugging switches (ABI 17), bit mask, 0 in production:
     * VIDU4D_DEBUG_NO_CULL: the blend kernels' footprint culls are off -- every list entry of a tile is evaluated for every
     *   pixel of the tile, which is the reference's walk (forward.cu:359-405, backward.cu:282-323).  The culls only prune
     *   work: outputs are bit-identical with and without (tests/test_gpu_cull_ab.py); hand the same flag to the backward.
     * VIDU4D_DEBUG_WHOLE_TILE_BACKWARD (read by the backward only): after a forward with segment_split == 0 the backward
     *   walks every tile with one workgroup, as the reference does (backward.cu:143-449).  Otherwise it walks the tiles
     *   longer than 320 entries in 256-entry segments on separate workgroups, each starting from the per-pixel sums the
     *   forward's own walk stored at the segment boundaries: same gradients up to fp32 re-association of those sums. */
    int debug_flags;
    /* ---- longest tile list the caller expects (extension, ABI 17; 0 = unknown): e.g. word 2 of the geometry buffer after
     * an earlier frame of the same view.  With segment_split == 0 and 0 < max_list_hint <= 900 the in-LDS sort of the
     * lists up to 1024 entries is the only sort launch (a longer list, should one appear after all, is sorted by that
     * launch through global memory: the result never depends on the hint). */
    int max_list_hint;
} Vidu4dSurfelForwardArgs;
#define VIDU4D_AUX_ALPHA 0x02
#define VIDU4D_AUX_GEOM 0x1F   /* planes 0-4: depth, alpha, normal */
#define VIDU4D_DEBUG_NO_CULL 1
#define VIDU4D_DEBUG_WHOLE_TILE_BACKWARD 2
#define VIDU4D_DEBUG_POSITION_ORDER 8  /* (ABI 18, read by the backward) after a segment-parallel forward the backward's workgroups
                                        * take their units in schedule-position order (rounds 2-4) instead of full segments
                                        * first, then the tails by descending size (round 5).  Same results up to the order of
                                        * the float atomics. */
#define VIDU4D_DEBUG_SERIAL_REPAIR 4   /* (ABI 18) the pre-pass-free segment-parallel forward (assume_unsaturated) adds its
                                        * segments up in ONE launch that also blends the saturating segments of a tile again,
                                        * one after the other (rounds 3-4); default since round 5: three launches -- scan,
                                        * one workgroup per saturating (tile, segment), add up.  Same results. */
#define VIDU4D_SCHED_PAIR(K) (((K) & 15) << 12)        /* (ABI 21, read by the forward with segment_split == 0; carried in debug_flags like
                                        * VIDU4D_SCHED_XCD_BLOCK, which it excludes) the LONGEST tiles of the launch are blended
                                        * by TWO workgroups each -- one per half of the tile's pixels, every wave on one 8x4 block
                                        * with its two halves on consecutive list entries and the transmittance recurrence run
                                        * over both -- so that a launch whose tiles all start at once does not wait for one
                                        * workgroup's walk of its longest list.  K = 1..14: the tiles longer than about K / 4 x the
                                        * mean list length of the launch (at most 1024 of them); 15: every tile (tests); 0: off.
                                        * Same transmittances, contributor counts and median samples; colour / depth / normal /
                                        * distortion sums of the paired tiles up to fp32 re-association. */
#define VIDU4D_SCHED_XCD_BLOCK(B) (((B) & 15) << 8)   /* (ABI 20, read by the forward; not a debugging switch but carried in
                                        * debug_flags) the blend and sort launches' schedule as EIGHT longest-first queues,
                                        * one per XCD: the tiles of a BxB-tile block of a frame go to one XCD (workgroup b runs
                                        * on XCD b mod 8), so that a surfel's record is gathered through one L2 instead of ~3.
                                        * 0: one longest-first queue over all tiles (rounds 1-5).  Same results up to the order
                                        * of the backward's float atomics. */
#define VIDU4D_SURFEL_MAX_FRAMES 8
size_t vidu4d_surfel_image_bytes_frames(int width, int height, int frames);

int vidu4d_surfel_forward_plan(const Vidu4dSurfelForwardArgs* args, void* stream);
/* Blocking device->host read of num_rendered (the reference's cudaMemcpy, rasterizer_impl.cu:282). */
int vidu4d_surfel_num_rendered(const Vidu4dSurfelForwardArgs* args, void* stream, int64_t* num_rendered);
int vidu4d_surfel_forward_run(const Vidu4dSurfelForwardArgs* args, void* binning_buffer, size_t binning_bytes,
                              int64_t capacity, void* stream);

/* ---- backward: replaces RasterizeGaussiansBackwardCUDA / Rasterizer::backward
 *      (rasterize_points.cu:143-240, rasterizer_impl.cu:346-448).  All dL_* outputs are fully
 *      written (they need not be zero-filled by the caller). ---- */
typedef struct Vidu4dSurfelBackwardArgs {
    int P, D, M;
    int width, height;
    float tan_fovx, tan_fovy;
    float scale_modifier;
    int debug;
    const float* background;
    const float* means3D;
    const int32_t* radii;
    const float* shs;
    const float* colors_precomp;
    const float* scales;
    const float* rotations;
    const float* transMat_precomp;   /* must be NULL */
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    const float* dL_dout_color;      /* (3,H,W) */
    const float* dL_dout_others;     /* (8,H,W) */
    const void* geom_buffer;         /* as filled by the forward */
    const void* binning_buffer;
    int64_t binning_capacity;        /* the capacity the forward ran with */
    const void* image_buffer;
    void* workspace;                 /* >= vidu4d_surfel_backward_workspace_bytes(P) */
    size_t workspace_bytes;
    float* dL_dmeans2D;              /* (P,3) densification statistic, backward.cu:645-648 */
    float* dL_dcolors;               /* (P,3) (gradient of colors_precomp; always written) */
    float* dL_dopacity;              /* (P,1) */
    float* dL_dmeans3D;              /* (P,3) */
    float* dL_dtransMat;             /* (P,9) */
    float* dL_dsh;                   /* (P,M,3) or NULL when M == 0 */
    float* dL_dscales;               /* (P,2) */
    float* dL_drotations;            /* (P,4) */
    int segment_split;               /* !=0: walk long tile lists segment-parallel; takes effect only when the
                                        forward that filled the buffers ran with segment_split != 0 */
    /* ---- stacked frames, as in the forward.  Inputs dL_dout_color (3,frames,H,W), dL_dout_others (8,frames,H,W);
     * per-frame outputs dL_dmeans2D / dL_dmeans3D (frames,P,3), dL_drotations (frames,P,4), dL_dcolors (frames,P,3),
     * dL_dtransMat (frames,P,9); the gradients of what the frames share -- dL_dopacity (P,1), dL_dscales (P,2),
     * dL_dsh (P,M,3) -- are the SUMS over the frames.  workspace: vidu4d_surfel_backward_workspace_bytes(frames * P). */
    int frames;
    const float* frame_viewmatrix[8];
    const float* frame_campos[8];
    float frame_tan_fovx[8];
    float frame_tan_fovy[8];
    /* ---- canonical parameters, as in the forward (same values there and here).  With sh_dc / sh_rest the SH gradient
     * leaves as dL_dsh_dc (P,1,3) + dL_dsh_rest (P,15,3) (16-byte aligned; dL_dsh is ignored).  raw_params != 0:
     * dL_dscales is the gradient w.r.t. the log-scales (dL/ds * s) and dL_dopacity w.r.t. the logits
     * ((dL/do * (1 - o)) * o, torch's sigmoid_backward). */
    const float* sh_dc;
    const float* sh_rest;
    float* dL_dsh_dc;
    float* dL_dsh_rest;
    int raw_params;
    /* ---- planes of dL_dout_others that may be non-zero (0 = all).  VIDU4D_AUX_ALPHA: only dL_dout_color and plane 1 are
     * read -- the other planes are TAKEN as zero, whatever they hold -- and the depth / normal / median / distortion
     * chains they would multiply are skipped.  Must name every plane the forward's aux_planes did not compute: after a
     * forward with VIDU4D_AUX_ALPHA the backward must be given VIDU4D_AUX_ALPHA too.  Planes 0-4 only (see the forward):
     * planes 5-7 are taken as zero and the median / distortion chains are skipped; the gradients are the full backward's
     * for such an input, bit for bit. */
    int aux_planes;
    int debug_flags;                 /* as in the forward (same value there and here) */
    void* diag_walk_counters;        /* NULL in production; see vidu4d_surfel_diag.h */
} Vidu4dSurfelBackwardArgs;

int vidu4d_surfel_backward(const Vidu4dSurfelBackwardArgs* args, void* stream);

/* ---- replaces markVisible / checkFrustum (rasterize_points.cu:242-261, rasterizer_impl.cu:54-66) */
int vidu4d_surfel_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                               uint8_t* present /* (P) bool */, void* stream);

/* ---- inspection of the opaque state (used by the parity tests to compare every stage with the
 *      oracle; not needed by a normal caller).  `what` selects the array, it is copied to `dst`
 *      (device or host pointer, any hipMemcpy-able) and the element count is returned in *count. */
enum Vidu4dSurfelStateArray {
    VIDU4D_STATE_NUM_RENDERED = 0,  /* uint32[1] */
    VIDU4D_STATE_RECORDS = 1,       /* float[P][32], 28 used  (layout: vidu4d_amd/csrc/surfel_math.h); floats 28..31 of a
                                     * record are padding to the 128-byte line and come back as whatever the buffer held */
    VIDU4D_STATE_TILES_TOUCHED = 2, /* uint32[P] */
    VIDU4D_STATE_POINT_LIST = 3,    /* uint32[num_rendered] sorted surfel ids (binning.point_list) */
    VIDU4D_STATE_SORTED_KEYS = 4,   /* uint64[num_rendered] (tile << 32 | depth bits); dst must be HOST memory */
    VIDU4D_STATE_RANGES = 5,        /* uint32[tiles][2] */
    VIDU4D_STATE_FINAL_T = 6,       /* float[3][H*W]: T, dist1, dist2 (the distortion moments about a per-tile reference depth) */
    VIDU4D_STATE_N_CONTRIB = 7,     /* uint32[2][H*W]: last, median */
    VIDU4D_STATE_TILE_ORDER = 8,    /* uint32[tiles] (ABI 20): the blend / sort launches' schedule -- workgroup b takes tile [b] */
    VIDU4D_STATE_TAIL_ORDER = 9,    /* uint32[tiles] (ABI 20): the recorded backward's tail units, by schedule position */
    VIDU4D_STATE_HEADER = 10        /* uint32[64] (ABI 20): the geometry buffer's header words (vidu4d_amd/csrc/surfel_state.h) */
};
int vidu4d_surfel_state_read(const Vidu4dSurfelForwardArgs* args, const void* binning_buffer, int64_t capacity,
                             int what, void* dst, size_t dst_bytes, int64_t* count, void* stream);

/* (Diagnostics -- per-stage HIP-event timers, tile-walk counters -- are declared in vidu4d_surfel_diag.h.) */

/* ---- quaternion ops: replace lab4d/third_party/quaternion/src/quaternion.cu
 *      (quaternion_mul :28-63 / :324-335, backward :66-140, backward-backward :143-214,
 *      conjugate :289-304).  Da/Db are 3 (pure-vector quaternion) or 4; out is (B,4). ---- */
int vidu4d_quaternion_mul(int64_t B, const float* a, int Da, const float* b, int Db, float* out, void* stream);
int vidu4d_quaternion_mul_backward(int64_t B, const float* grad_out, const float* a, int Da, const float* b, int Db,
                                   float* grad_a, float* grad_b, void* stream);
int vidu4d_quaternion_mul_backward_backward(int64_t B, const float* gg_a, const float* gg_b, const float* grad_out,
                                            const float* a, int Da, const float* b, int Db, float* gg_out,
                                            float* g_a, float* g_b, void* stream);
int vidu4d_quaternion_conjugate(int64_t B, const float* q, float* out, void* stream);

/* ---- fused bob linear-blend-skinning apply (forward warp of Stage-3): replaces the chain
 *      dual_quaternion_skinning(return_qt=True) (lab4d/utils/geom_utils.py:48-92) ->
 *      apply_qt_to_gaussian (lab4d/nnutils/deformable_gaussian.py:1032-1046) -> field2cam apply
 *      (:1425-1430).  wT (B,N): softmax skinning weights, transposed; se3_qr/se3_qd (M,B,4): per-frame
 *      bone transforms as dual quaternions; xyz (N,3), rot (N,4): canonical surfels; cam_q (M,4),
 *      cam_t (M,3): object-to-camera.  Outputs out_xyz (M,N,3), out_rot (M,N,4).  The backward
 *      writes per-frame gradients g_wT (M,B,N), g_xyz (M,N,3), g_rot (M,N,4); bone and camera
 *      parameters are treated as constants. ---- */
int vidu4d_lbs_forward(int M, int N, int B, const float* wT, const float* se3_qr, const float* se3_qd,
                       const float* xyz, const float* rot, const float* cam_q, const float* cam_t, float* out_xyz,
                       float* out_rot, void* stream);
int vidu4d_lbs_backward(int M, int N, int B, const float* wT, const float* se3_qr, const float* se3_qd,
                        const float* xyz, const float* rot, const float* cam_q, const float* cam_t,
                        const float* g_out_xyz, const float* g_out_rot, float* g_wT, float* g_xyz, float* g_rot,
                        void* stream);

/* Skinning weights + blend + apply: replaces, on top of the above, SkinningField.forward's
 * Gaussian-bone distances, the relu * 0.1 of the delta-skin MLP output and the softmax
 * (lab4d/nnutils/skinning.py:89-142, lab4d/nnutils/warping.py:415-427).  Feature-major inputs: xbT (3B, N) bone
 * coordinates x_bone / gauss, rawT (B, N) raw delta-MLP output or NULL; one evaluation of the weights serves the
 * M <= 8 frames.  The backward writes g_xbT (3B, N), g_rawT (B, N) (NULL iff rawT is) and the gradients w.r.t. the
 * canonical centres (N, 3) and orientations (N, 4) already summed over the frames.  unit_rot != 0: the orientations
 * leave the kernel normalised (v / max(|v|, 1e-12): the rotation_activation that gs.gaussian_renderer.render applies to
 * them, gs/scene/gaussian_model.py:57 / gs/gaussian_renderer/__init__.py:73) and the backward goes through it.
 * bone_A (3B, 3) / bone_c (3B) with xbT == NULL: the bone coordinates are evaluated inside the kernels, x_bone = A xyz + c
 * (the rest pose's bone map, as vidu4d_skin_field_* takes it), and the backward adds A^T (d/d x_bone) to g_xyz instead of
 * writing g_xbT (which may be NULL): three quarters of the kernels' traffic.  Otherwise pass bone_A = bone_c = NULL.
 * frame_index (M device int64, or NULL; ABI 17): se3_qr / se3_qd / cam_q / cam_t are then TABLES over all frames of the
 * sequence ((frames,B,4), (frames,4), (frames,3): what frozen networks give once per run) and frame m of the call is their
 * row frame_index[m] -- the per-step row gathers happen inside the kernels.  table_rows (ABI 18): the number of rows of
 * those tables; an index outside [0, table_rows) reads the first / last row (the torch indexing this replaces raised).
 * g_params (backward, ABI 18; NULL = bones and cameras are constants, --gs_optim_warp=False): the gradients w.r.t. the
 * frames' bone dual quaternions and cameras, for networks that train (the reference's default, lab4d/config.py:157;
 * AdamW on them, lab4d/engine/trainer.py:592-598).  They are sums over all surfels; every workgroup of 256 surfels leaves
 * ONE row of partial sums, g_params = float[vidu4d_lbs_skin_param_rows(N)][M][8 B + 8]:
 *   [m][8 b .. 8 b + 3] d/d se3_qr[m][b]   [m][8 b + 4 .. 8 b + 7] d/d se3_qd[m][b]
 *   [m][8 B .. 8 B + 3] d/d cam_q[m]       [m][8 B + 4 .. 8 B + 6] d/d cam_t[m]      ([m][8 B + 7]: 0)
 * and the caller adds the rows up.  Needs the bone coordinates as an input (xbT != NULL, bone_A == NULL:
 * VIDU4D_E_UNSUPPORTED otherwise) -- the gradient of the bone map then flows through whatever produced xbT -- and
 * frame_index == NULL (per-frame rows).  The hemisphere signs and the anchor bone are piecewise constant, as in the
 * reference's graph (lab4d/utils/geom_utils.py:66-74). */
int vidu4d_lbs_skin_forward(int M, int N, int B, const float* xbT, const float* rawT, const float* se3_qr,
                            const float* se3_qd, const float* xyz, const float* rot, const float* cam_q,
                            const float* cam_t, float* out_xyz, float* out_rot, int unit_rot, const float* bone_A,
                            const float* bone_c, const int64_t* frame_index, int table_rows, void* stream);
int vidu4d_lbs_skin_backward(int M, int N, int B, const float* xbT, const float* rawT, const float* se3_qr,
                             const float* se3_qd, const float* xyz, const float* rot, const float* cam_q,
                             const float* cam_t, const float* g_out_xyz, const float* g_out_rot, float* g_xbT,
                             float* g_rawT, float* g_xyz, float* g_rot, int unit_rot, const float* bone_A,
                             const float* bone_c, const int64_t* frame_index, int table_rows, float* g_params,
                             void* stream);
int vidu4d_lbs_skin_param_rows(int N);

/* ---- (ABI 19) from the articulation network's heads to the tables vidu4d_lbs_skin_* reads, for networks that train:
 *      axis-angle and translation per bone -> unit dual quaternion (ArticulationFlatMLP.forward,
 *      lab4d/nnutils/pose.py:300-323; quat_transform.py axis_angle_to_quaternion :136-160,
 *      quaternion_translation_to_dual_quaternion :341-360), each frame's bones relative to the rest pose's
 *      (dual_quaternion_mul(t, dual_quaternion_inverse(rest)), lab4d/nnutils/warping.py:415-425, quat_transform.py:420-469),
 *      and the rest pose's object->bone rotation and translation with every row divided by the Gaussian bone's extent
 *      (gauss_mlp_skinning's bone coordinates, lab4d/nnutils/skinning.py:117-141; quaternion_to_matrix,
 *      quat_transform.py:221-255) -- ~75 elementwise launches forward and ~175 backward in the reference's graph, one here.
 *      so3_t / trans_t (M, B, 3): the heads' outputs for the M frames; so3_rest / trans_rest (B, 3): for the mean time code;
 *      inv_gauss (B, 3) = 1 / exp(log_gauss).  Outputs se3_qr / se3_qd (M, B, 4), bone_A (3B, 3) row 3 b + k =
 *      R_b[k][:] * inv_gauss[b][k], bone_c (3B) = t_b[k] * inv_gauss[b][k]  (bone_A / bone_c may both be NULL: skipped).
 *      The backward takes the gradients of the four outputs (any may be NULL = zero) and returns those of the five
 *      inputs (g_inv_gauss may be NULL); it evaluates the forward's own code on dual numbers, one direction at a time. ---- */
int vidu4d_bone_tables_forward(int M, int B, const float* so3_t, const float* trans_t, const float* so3_rest,
                               const float* trans_rest, const float* inv_gauss, float* se3_qr, float* se3_qd,
                               float* bone_A, float* bone_c, void* stream);
int vidu4d_bone_tables_backward(int M, int B, const float* so3_t, const float* trans_t, const float* so3_rest,
                                const float* trans_rest, const float* inv_gauss, const float* g_se3_qr,
                                const float* g_se3_qd, const float* g_bone_A, const float* g_bone_c, float* g_so3_t,
                                float* g_trans_t, float* g_so3_rest, float* g_trans_rest, float* g_inv_gauss, void* stream);

/* ---- (ABI 19) the camera network's tail (CameraMLP.get_vals, lab4d/nnutils/pose.py:120-150): cam_q[m] =
 *      normalize(raw_quat[m]) * normalize(base_quat[m]) (F.normalize: v / max(|v|, 1e-12); Hamilton product, w first), M rows
 *      of 4, one launch per direction instead of ~10 + ~30 elementwise ones; the backward on dual numbers like bone_tables. ---- */
int vidu4d_camera_tail_forward(int M, const float* raw_quat, const float* base_quat, float* cam_q, void* stream);
int vidu4d_camera_tail_backward(int M, const float* raw_quat, const float* base_quat, const float* g_cam_q,
                                float* g_raw_quat, float* g_base_quat, void* stream);

/* ---- (ABI 19) a stack of dense layers on a handful of rows, one launch per direction: the time-conditioned networks of
 *      the bob warp evaluated for the frames of a step (TimeMLP and its heads: lab4d/nnutils/time.py:11-133,
 *      pose.py:29-150 CameraMLP, :153-323 ArticulationFlatMLP; BaseMLP layers base.py:8-157).  A trunk of n_trunk layers
 *      (layer 0 reads x (rows, in[0])), then up to two heads that each start from the trunk's output.  Layer l:
 *      y = scale[l] * act(W[l] h + b[l]), W[l] (out[l], in[l]) row-major as torch.nn.Linear.weight, act = relu when
 *      relu[l] else identity; b[l] may be NULL.  rows <= 16, widths <= 256.
 *      forward: acts receives every layer's output, layer after layer, (rows, out[l]) each
 *               (vidu4d_dense_stack_acts_floats() floats in total); the heads' results are its last two blocks.
 *      backward: g_out_a / g_out_b (rows, out of the head's last layer; NULL = zero; without heads g_out_a is the trunk
 *               output's gradient) -> gW[l] / gb[l] (written, not accumulated; NULL = skipped) and g_x (rows, in[0]; may be
 *               NULL).  Needs the forward's x and acts, and a workspace of vidu4d_dense_stack_acts_floats() floats. ---- */
#define VIDU4D_DENSE_STACK_MAX_LAYERS 16
#define VIDU4D_DENSE_STACK_MAX_WIDTH 256
#define VIDU4D_DENSE_STACK_MAX_ROWS 16
typedef struct Vidu4dDenseStack {
    int rows, n_trunk, n_head_a, n_head_b;
    int in[VIDU4D_DENSE_STACK_MAX_LAYERS], out[VIDU4D_DENSE_STACK_MAX_LAYERS], relu[VIDU4D_DENSE_STACK_MAX_LAYERS];
    float scale[VIDU4D_DENSE_STACK_MAX_LAYERS];
    const float* W[VIDU4D_DENSE_STACK_MAX_LAYERS];
    const float* b[VIDU4D_DENSE_STACK_MAX_LAYERS];
    float* gW[VIDU4D_DENSE_STACK_MAX_LAYERS];   /* backward only */
    float* gb[VIDU4D_DENSE_STACK_MAX_LAYERS];
} Vidu4dDenseStack;
int vidu4d_dense_stack_acts_floats(const Vidu4dDenseStack* s);   /* -1: invalid description */
int vidu4d_dense_stack_forward(const Vidu4dDenseStack* s, const float* x, float* acts, void* stream);
int vidu4d_dense_stack_backward(const Vidu4dDenseStack* s, const float* x, const float* acts, const float* g_out_a,
                                const float* g_out_b, float* workspace, float* g_x, void* stream);

/* ---- the per-surfel part of the bob skinning field, once per optimizer step: Gaussian-bone coordinates of the rest
 *      pose and the delta-skin MLP on them (replaces gauss_mlp_skinning's bone transform and SkinningField.delta_field,
 *      lab4d/nnutils/skinning.py:89-142, as called by SkinningWarp.forward with the rest articulation and the mean
 *      time code, lab4d/nnutils/warping.py:415-427).  Feature-major outputs xbT (3B, N), rawT (B, N): the inputs of
 *      vidu4d_lbs_skin_*.  Bones and weights are constants (frozen networks); the backward returns d/d xyz only.
 *      All weight arrays are device pointers in the network's own row-major (out, in) layout, zero-padded to the
 *      sizes below:
 *        bone_A (3B, 3), bone_c (3B): x_bone = A xyz + c, 3 rows per bone;
 *        w_in (W, IN_MAX): first layer, coordinate columns; b_in (W): its bias plus its time- and instance-code
 *        columns applied to the step's code vector;
 *        w_hid (D-1, W, W), b_hid (D-1, W): further hidden layers;  w_out (OUT_MAX, W), b_out (OUT_MAX): output layer.
 *      ReLU after every hidden layer, none after the output (the caller's relu(.) * 0.1 is part of
 *      vidu4d_lbs_skin_*).  Runs on v_mfma_f32_32x32x2_f32 (exact fp32). ---- */
#define VIDU4D_SKIN_FIELD_WIDTH 64
#define VIDU4D_SKIN_FIELD_IN_MAX 96
#define VIDU4D_SKIN_FIELD_OUT_MAX 32
#define VIDU4D_SKIN_FIELD_MAX_HIDDEN 4
typedef struct Vidu4dSkinFieldArgs {
    int N, B, W, D;
    const float* xyz;
    const float* bone_A;
    const float* bone_c;
    const float* w_in;
    const float* b_in;
    const float* w_hid;
    const float* b_hid;
    const float* w_out;
    const float* b_out;
    float* xbT;          /* forward outputs */
    float* rawT;
    const float* g_xbT;  /* backward inputs: gradient w.r.t. xbT (may be NULL) and rawT */
    const float* g_rawT;
    float* g_xyz;        /* backward output (N, 3) */
    uint32_t* relu_masks; /* optional, D * 64 * ceil(N / 32) words: the forward records which hidden units are active,
                             the backward then skips its recomputation of the hidden layers (NULL: it recomputes) */
    /* optional (extension, ABI 16): the weights as vidu4d_skin_field_pack arranged them for the forward / backward
     * kernel (16-byte aligned).  The kernels then copy that image into LDS instead of gathering it from the arrays
     * above, which costs every launch ~20 us; b_in is still read from b_in.  NULL: gather. */
    const float* packed_fwd;
    const float* packed_bwd;
    /* optional (extension, ABI 19): networks that TRAIN (--gs_optim_warp=True, lab4d/config.py:157).  Feature-major arrays
     * from which the caller takes the weight gradients as contractions over the surfels:
     *   h_store  (forward output)  float[D][h_store_rows or W][N]   hidden layer l's activations (after the ReLU), rows 0..W-1
     *   g_store  (backward output) float[D][W][N]   the gradient w.r.t. hidden layer l's PRE-activation (needs relu_masks)
     *   gx_store (backward output) float[3B][N]     the whole gradient w.r.t. the bone coordinates (g_xbT + the MLP's)
     * d w_out = g_rawT h_store[D-1]^T, d w_hid[l-1] = g_store[l] h_store[l-1]^T, d w_in = g_store[0] xbT^T,
     * d bone_A = gx_store xyz, the biases the row sums.  g_store and gx_store come together; all NULL: frozen networks. */
    float* h_store;
    float* g_store;
    float* gx_store;
    int h_store_rows;   /* rows per layer of h_store: 0 = W; W + 1 leaves room for a row of ones behind each layer, which folds
                           the bias gradient into the weight gradient's contraction (d [w | b] = g [h; 1]^T) */
} Vidu4dSkinFieldArgs;
int vidu4d_skin_field_forward(const Vidu4dSkinFieldArgs* args, void* stream);
int vidu4d_skin_field_backward(const Vidu4dSkinFieldArgs* args, void* stream);
/* the kernels' weight image: vidu4d_skin_field_packed_floats(B, D, backward) floats (0: unsupported shape), written by
 * vidu4d_skin_field_pack from args' weight arrays (N, xyz, b_in and the output pointers are not read).  Valid while
 * the weights and the bone map stay what they were. */
int vidu4d_skin_field_packed_floats(int B, int D, int backward);
int vidu4d_skin_field_pack(const Vidu4dSkinFieldArgs* args, int backward, float* out, void* stream);

/* ---- mean squared distance of every point to its 3 nearest other points (exact): replaces
 *      simple-knn's distCUDA2 (gs/submodules/simple-knn/spatial.cu:15-25, simple_knn.cu:185-221), used
 *      once by GaussianModel.create_from_pcd (gs/scene/gaussian_model.py:134-136).  points (P,3),
 *      mean_dist2 (P).  Fewer than 4 points leave FLT_MAX terms in the mean, as upstream. ---- */
int vidu4d_knn_mean_dist2(int P, const float* points, float* mean_dist2, void* stream);
/* number of points (the query included) strictly closer than `radius` to each point: the neighbour count of
 * open3d's remove_radius_outlier, which the reference's Stage-3 loop runs on the CPU every 2000 steps
 * (lab4d/engine/trainer.py:573-588).  points (P,3), counts (P) int32. */
int vidu4d_radius_count(int P, const float* points, float radius, int32_t* counts, void* stream);

/* ---- fused post-processing of the auxiliary planes: what gs.gaussian_renderer.render computes after the
 *      blend (gs/gaussian_renderer/__init__.py:118-151, gs/utils/point_utils.py:9-37).  allmap (8,H,W);
 *      rays_d (H*W,3), rays_o (3): pixel rays of the camera; view3x3 (3,3) row-major = the block the
 *      rendered normals are multiplied by from the right; outputs rend_normal (3,H,W), depth_median /
 *      depth_expected / surf_depth (H,W), surf_normal (3,H,W).  The backward writes g_allmap (8,H,W)
 *      completely (planes 1 [through the division only], 6, 7 get what flows through these outputs: the
 *      caller adds the gradients of its own reads of alpha / distortion); any g_* input may be NULL. ---- */
int vidu4d_post_forward(int W, int H, const float* allmap, const float* rays_d, const float* rays_o,
                        const float* view3x3, float depth_ratio, float* rend_normal, float* depth_median,
                        float* depth_expected, float* surf_depth, float* surf_normal, void* stream);
int vidu4d_post_backward(int W, int H, const float* allmap, const float* surf_depth, const float* rays_d,
                         const float* rays_o, const float* view3x3, float depth_ratio, const float* g_rend_normal,
                         const float* g_depth_median, const float* g_depth_expected, const float* g_surf_depth,
                         const float* g_surf_normal, float* g_allmap, void* stream);

/* ---- the surfel optimizer's update: replaces the per-group launches of torch.optim.Adam(eps=1e-15) that
 *      Trainer.optimizer_init builds with one parameter group per surfel attribute (lab4d/engine/trainer.py:240-255)
 *      by ONE launch over all groups.  Update rule of torch/optim/adam.py (no amsgrad, no weight decay); the caller
 *      owns the step counts and passes 1 - beta1^t and sqrt(1 - beta2^t) per tensor; the betas are doubles so that
 *      1 - beta is rounded to fp32 once, as torch does.  grad_scale (device scalar or NULL): every gradient is multiplied
 *      by it on the way in -- the coefficient of torch.nn.utils.clip_grad_norm_ (lab4d/engine/trainer.py:861-869),
 *      which upstream applies in a pass of its own; zero_grads != 0: the gradient arrays are left zero-filled (the next
 *      step's zero_grad), `grad` is then written to.  `tensors` is a HOST array. ---- */
#define VIDU4D_ADAM_MAX_TENSORS 8
typedef struct Vidu4dAdamTensor {
    float* param;
    float* grad;        /* read; zero-filled afterwards when zero_grads is set */
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
    float lr;
    float bias_correction1;      /* 1 - beta1^step */
    float bias_correction2_sqrt; /* sqrt(1 - beta2^step) */
    /* optional (ABI 20): {lr, bias_correction1, bias_correction2_sqrt} read from DEVICE memory when the kernel runs, instead
     * of the three values above -- a launch captured into a hipGraph bakes its by-value arguments, these change every step
     * (the host writes them with one small async copy in front of the replay; lab4d/captured_step.py) */
    const float* device_scalars;
} Vidu4dAdamTensor;
int vidu4d_adam_step(int n, const Vidu4dAdamTensor* tensors, double beta1, double beta2, double eps,
                     const float* grad_scale, int zero_grads, void* stream);
/* (ABI 20) the same with a device-side guard: when *skip != 0 at run time the launch changes nothing (no update, no zero fill).
 * A fitting step replayed from a captured hipGraph cannot ask the host whether its forward outgrew the binning buffer (it then
 * rendered the background only): the step's verdict is a device word, and the update that would apply garbage gradients reads
 * it.  skip == NULL: vidu4d_adam_step. */
int vidu4d_adam_step_guarded(int n, const Vidu4dAdamTensor* tensors, double beta1, double beta2, double eps,
                             const float* grad_scale, int zero_grads, const uint32_t* skip, void* stream);
/* (ABI 21) AdamW for the warp / camera networks (reference: lab4d/engine/trainer.py:177-286, torch.optim.AdamW(betas (0.9,
 * 0.999), weight_decay 1e-4) over 66 small tensors): the update above with p <- p - lr * weight_decay * p in front (torch's
 * fused kernel, ATen/native/cuda/fused_adam_utils.cuh, ADAMW mode), n <= VIDU4D_ADAMW_MAX_TENSORS tensors per launch -- three
 * launches for the bob networks where torch's fused capturable form takes seven of ~24 us each. */
#define VIDU4D_ADAMW_MAX_TENSORS 32
int vidu4d_adamw_step_guarded(int n, const Vidu4dAdamTensor* tensors, double beta1, double beta2, double eps, double weight_decay,
                              const float* grad_scale, int zero_grads, const uint32_t* skip, void* stream);

/* ---- (ABI 21) contractions over the surfels: out_j (rows_a x rows_b, row-major, ADDED to: the caller zero-fills) +=
 *      sum_k a_j[r][k] * b_j[c][k] for up to VIDU4D_CONTRACT_MAX_JOBS pairs of strided (rows x K)
 *      operands of at most 96 rows each, in ONE launch -- the weight gradients of the skinning field's layers when the networks
 *      train (reference: autograd through lab4d/nnutils/skinning.py's delta MLP; here lab4d/lbs_fused._SkinFieldTrain.backward).
 *      fp32 matrix cores (an fmaf chain); partial sums leave through float atomics: not a fixed summation order.  `jobs` is
 *      a HOST array. ---- */
#define VIDU4D_CONTRACT_MAX_JOBS 4
typedef struct Vidu4dContractJob {
    const float* a;   /* element (r, k) at a[r * lda + k * sa] */
    const float* b;   /* element (c, k) at b[c * ldb + k * sb] (an (N, 4) array of points as four rows: ldb 1, sb 4) */
    float* out;       /* (rows_a, rows_b) */
    int rows_a, rows_b;
    int64_t lda, ldb, sa, sb;
} Vidu4dContractJob;
int vidu4d_contract_rows(int n, const Vidu4dContractJob* jobs, int64_t K, void* stream);

/* ---- (ABI 21) up to VIDU4D_COPY_MAX_JOBS strided 2-D copies of fp32 in one launch: dst[r * dst_ld + c] = src[r * src_ld +
 *      c * src_cs] for r < rows, c < cols (a transposing copy: src_ld 1, src_cs = the source's row length).  `jobs` is a HOST
 *      array.  Used by the TRAIN skinning field to repack its weights and the surfel centres every step. ---- */
#define VIDU4D_COPY_MAX_JOBS 8
typedef struct Vidu4dCopyJob {
    const float* src;
    float* dst;
    int64_t rows, cols, src_ld, src_cs, dst_ld;
} Vidu4dCopyJob;
int vidu4d_copy_strided(int n, const Vidu4dCopyJob* jobs, void* stream);

/* ---- the gradient clip's norm and coefficient: torch.nn.utils.clip_grad_norm_(params, max_norm) as Trainer.check_grad
 *      calls it (lab4d/engine/trainer.py:861-869) is a norm per tensor, a stack, a norm, an add, a division, a clamp and
 *      a multiply per tensor; here ONE launch reads the n gradient arrays (grads[i], numel[i]; HOST arrays, n <=
 *      VIDU4D_CLIP_MAX_TENSORS; 96 since ABI 20: the surfels' flat buffer and the bob networks' 66 tensors in one launch)
 *      and writes THREE floats: out[0] = the 2-norm over all of them, out[1] = min(1, max_norm / (norm + 1e-6)) -- the
 *      factor vidu4d_adam_step takes as grad_scale -- and (ABI 20) out[2] = 1 / out[1], the grad_scale of an optimizer
 *      that divides by it (torch's fused AdamW).  workspace: VIDU4D_CLIP_WORKSPACE_FLOATS device floats whose
 *      FIRST word is zero at entry (the kernel leaves it zero again); partial sums are added in a fixed order. ---- */
#define VIDU4D_CLIP_MAX_TENSORS 96
#define VIDU4D_CLIP_WORKSPACE_FLOATS 1056
int vidu4d_grad_clip_coef(int n, const float* const* grads, const int64_t* numel, float max_norm, float* workspace,
                          float* out, void* stream);

/* ---- densify_and_prune on the device: replaces GaussianModel.densify_and_clone / densify_and_split / prune_points
 *      and the optimizer surgery under them (gs/scene/gaussian_model.py:270-356, :384-448).
 *  plan:  per surfel, from grad_accum / denom (N), log-scales (N,2), opacity logits (N): counts (3,N) int32 =
 *         {the original survives, its clone exists and survives, its two split copies exist and survive}.
 *         dense_extent = percent_dense * extent; big_world = 0.1 * extent, or < 0 when the caller passes no
 *         max_screen_size (the world-size criterion is tied to it upstream, :443-446).
 *  apply: with the inclusive prefix sums of the three count rows and their totals, writes every attribute (and, where
 *         given, its two Adam moments: kept for surviving originals, zero for new rows) in the reference's row order
 *         [originals | clones | first split copies | second split copies].  Split copies get
 *         xyz + R(rotation) (draw * (sx, sy, 0)) and log(scale / 1.6); `draws` (2,N,3) holds unit normal draws per
 *         source surfel and copy (or, draws_are_scaled != 0, the already scaled samples).  src_row (rows) int32 and
 *         kind (rows) uint8 are scratch the caller provides; `attrs` is a HOST array. ---- */
#define VIDU4D_DENSIFY_MAX_ATTRS 8
typedef struct Vidu4dDensifyAttr {
    const float* src;   /* (N, width) */
    float* dst;         /* (rows, width) */
    const float* src_m; /* Adam exp_avg of src, or NULL (no optimizer state) */
    float* dst_m;
    const float* src_v; /* Adam exp_avg_sq */
    float* dst_v;
    int width;
} Vidu4dDensifyAttr;
int vidu4d_densify_plan(int N, const float* grad_accum, const float* denom, const float* scaling, const float* opacity,
                        float grad_threshold, float dense_extent, float min_opacity, float big_world, int32_t* counts,
                        void* stream);
int vidu4d_densify_apply(int N, const int32_t* inclusive_counts, int n_orig, int n_clone, int n_split, int n_attrs,
                         const Vidu4dDensifyAttr* attrs, int xyz_attr, int scaling_attr, int rotation_attr,
                         const float* draws, int draws_are_scaled, int32_t* src_row, uint8_t* kind, void* stream);

/* ---- Stage-3 loss terms straight from the rasterizer's planes, and their gradients straight into the planes the
 *      rasterizer's backward consumes: replaces the elementwise / reduction chain of dvr_model.compute_recon_loss,
 *      mask_losses, get_mask_balance_wt, compute_reg_loss (distortion term) and apply_loss_weights for the surfel
 *      field under --rgb_loss_only (lab4d/engine/model.py:586-693, :835-842, :895-1012) together with the learnable-
 *      background composite in front of it (lab4d/nnutils/deformable_gaussian.py:1216-1218).
 *      color[m] (3,H,W), allmap[m] (8,H,W): the rasterizer outputs of frame m (M <= 8); bkgd (3) learnable background
 *      or NULL; rgb (M,H,W,3), mask / vis2d (M,H,W,1) targets; det (M) 1/0 per frame or NULL.  losses (5) =
 *      {rgb, mask, dist, normal, their sum}, each term already multiplied by its weight.  sums (32 floats) and partials
 *      (VIDU4D_LOSS_BLOCKS * 16 floats) are scratch that must stay untouched between forward and backward.
 *      The backward takes g_losses (5, device: upstream gradients of the four terms and of their sum, the latter is
 *      added to each) and writes every plane of g_color[m] / g_allmap[m] and g_bkgd (3).
 *      normal_wt != 0 adds the normal-consistency term (compute_reg_loss, model.py:817-834, with its sum over the FRAME
 *      axis): lambda * mean_{H,W,3}(1 - sum_m rend_normal_m * surf_normal_m), where rend_normal = allmap[2:5] rotated by
 *      view3x3[m] (row-major 3x3, out_j = sum_i n_i M[i][j]) and surf_normal is what gs.gaussian_renderer.render derives
 *      from the depth planes (gs/gaussian_renderer/__init__.py:118-151, gs/utils/point_utils.py:9-37; depth_ratio as
 *      pipe.depth_ratio) -- evaluated inside the kernels from rays_d[m] (H*W,3) / rays_o[m] (3) with surf_depth
 *      (M*H*W floats) as the workspace between forward and backward, or, when surf_normal[m] (3,H,W) is given, read
 *      from there (its gradient then goes to g_surf_normal[m], if not NULL, instead of the depth planes). ---- */
#define VIDU4D_LOSS_MAX_FRAMES 32   /* (ABI 20; 8 before: a step of imgs_per_gpu > 4 -- two frames per image pair -- fell to the torch losses) */
#define VIDU4D_LOSS_BLOCKS 1024
#define VIDU4D_LOSS_SUMS_FLOATS 32
typedef struct Vidu4dStage3LossArgs {
    int M, H, W;
    const float* color[VIDU4D_LOSS_MAX_FRAMES];
    const float* allmap[VIDU4D_LOSS_MAX_FRAMES];
    const float* bkgd;
    const float* rgb;
    const float* mask;
    const float* vis2d;
    const float* det;
    float lambda_dssim, rgb_wt, mask_wt, dist_wt;
    float* sums;
    float* partials;
    float* losses;
    int64_t plane_stride;   /* floats between two planes of color[m] / allmap[m] and of the gradient planes; 0 = H*W.
                               M*H*W when the frames come from one stacked rasterizer call ((3,M,H,W) / (8,M,H,W)
                               tensors, color[m] = base + m*H*W) */
    float normal_wt, depth_ratio;
    const float* rays_d[VIDU4D_LOSS_MAX_FRAMES];
    const float* rays_o[VIDU4D_LOSS_MAX_FRAMES];
    const float* view3x3[VIDU4D_LOSS_MAX_FRAMES];
    const float* surf_normal[VIDU4D_LOSS_MAX_FRAMES];
    float* surf_depth;
} Vidu4dStage3LossArgs;
typedef struct Vidu4dStage3LossGrads {
    float* g_color[VIDU4D_LOSS_MAX_FRAMES];
    float* g_allmap[VIDU4D_LOSS_MAX_FRAMES];
    float* g_bkgd;
    float* g_surf_normal[VIDU4D_LOSS_MAX_FRAMES];
} Vidu4dStage3LossGrads;
int vidu4d_stage3_loss_forward(const Vidu4dStage3LossArgs* args, void* stream);
int vidu4d_stage3_loss_backward(const Vidu4dStage3LossArgs* args, const float* g_losses,
                                const Vidu4dStage3LossGrads* grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif
