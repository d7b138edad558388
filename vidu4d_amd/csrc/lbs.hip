// lbs.hip -- fused "bob" linear-blend-skinning apply for gfx950: per canonical surfel and frame,
// hemisphere-aligned dual-quaternion blend of the bone transforms, normalisation, conversion to
// (rotation, translation), application to the surfel centre and orientation, and the object-to-
// camera transform -- forward and backward in one kernel each.
//
// Replaces, for the Stage-3 forward warp, the chain dual_quaternion_skinning(return_qt=True)
// (/root/reference/lab4d/utils/geom_utils.py:48-92) -> apply_qt_to_gaussian
// (lab4d/nnutils/deformable_gaussian.py:1032-1046) -> field2cam apply (:1425-1430), which upstream
// runs as ~40 elementwise torch kernels per frame over (M,N,B,4) materialised copies of the bone
// dual quaternions.  The skinning weights (softmax of the Gaussian-bone logits + delta MLP) are an
// input: in the forward warp they do not depend on the frame (warping.py:415-425 uses the rest
// articulation and the mean time code), so the caller computes them once per step.
//
// One thread per (frame, surfel).  Bone dual quaternions and the BxB hemisphere-sign table of the
// frame live in LDS (wave-uniform broadcast reads); weights are read transposed (B,N) so that the
// 25 loads of a wave are coalesced.  Gradients are produced w.r.t. the weights, the canonical centre
// and the canonical orientation (per frame; the caller sums over frames); bone and camera
// parameters are treated as constants (--gs_optim_warp=False, the README's Stage-3 setting).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/vidu4d_surfel.h"
#include "wave_reduce.h"

namespace {

constexpr int MAX_BONES = 64;

struct Q {
    float w, x, y, z;
};
__device__ __forceinline__ Q qmul(Q a, Q b)
{
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q qconj(Q a) { return {a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ Q qadd(Q a, Q b) { return {a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Q qscale(Q a, float s) { return {a.w * s, a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float qdot(Q a, Q b) { return a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Q qvec(float x, float y, float z) { return {0.f, x, y, z}; }

// y = q (0,p) q*  and its adjoint w.r.t. q (as 4 free components, like autograd of the reference's
// quaternion_apply, quat_transform.py:259-276) and p.
__device__ __forceinline__ void rotate(Q q, Q p, Q& p1, Q& out)
{
    p1 = qmul(q, p);
    out = qmul(p1, qconj(q));
}
__device__ __forceinline__ void rotate_bwd(Q q, Q p, Q p1, Q g_out_vec, Q& g_q, Q& g_p)
{
    // out = p1 * conj(q); only the vector part of out is used downstream (g_out_vec.w == 0)
    const Q g_p1 = qmul(g_out_vec, q);               // g * conj(conj(q))
    const Q g_qc = qmul(qconj(p1), g_out_vec);        // gradient w.r.t. conj(q)
    g_q = qadd(qconj(g_qc), qmul(g_p1, qconj(p)));    // + through p1 = q * p
    g_p = qmul(qconj(q), g_p1);
}

struct Frame {
    Q qr[MAX_BONES];
    Q qd[MAX_BONES];
};

// Stages bone dual quaternions of frame m into LDS and builds the hemisphere-sign bit table.
__device__ __forceinline__ void stage_frame(const float* se3_qr, const float* se3_qd, int m, int B, float* s_q,
                                            unsigned long long* s_sign)
{
    for (int i = threadIdx.x; i < B * 4; i += blockDim.x) {
        s_q[i] = se3_qr[(size_t)m * B * 4 + i];
        s_q[MAX_BONES * 4 + i] = se3_qd[(size_t)m * B * 4 + i];
    }
    for (int a = threadIdx.x; a < B; a += blockDim.x) s_sign[a] = 0ull;
    __syncthreads();
    // bit b of s_sign[a]: bone b is in bone a's hemisphere (geom_utils.py:70-72).  One thread per (a, b) pair: with one
    // thread per ROW (B of the 256 threads walking B * 4 products each) this prologue was the longest dependent chain of
    // the kernel -- every workgroup paid ~B^2 * 8 LDS reads on 25 lanes before its surfels could start.
    for (int i = threadIdx.x; i < B * B; i += blockDim.x) {
        const int a = i / B, b = i - a * B;
        float d = 0.f;
        for (int k = 0; k < 4; k++) d += s_q[a * 4 + k] * s_q[b * 4 + k];
        if (d > 0.f) atomicOr(&s_sign[a], 1ull << b);
    }
    __syncthreads();
}

__device__ __forceinline__ Q ldq(const float* p) { return {p[0], p[1], p[2], p[3]}; }

template <bool BACKWARD>
__global__ __launch_bounds__(256) void lbs_kernel(int N, int B, const float* __restrict__ wT /*(B,N)*/,
                                                  const float* __restrict__ se3_qr, const float* __restrict__ se3_qd,
                                                  const float* __restrict__ xyz, const float* __restrict__ rot,
                                                  const float* __restrict__ cam_q, const float* __restrict__ cam_t,
                                                  float* __restrict__ out_xyz, float* __restrict__ out_rot,
                                                  const float* __restrict__ g_out_xyz,
                                                  const float* __restrict__ g_out_rot, float* __restrict__ g_wT,
                                                  float* __restrict__ g_xyz, float* __restrict__ g_rot)
{
    __shared__ float s_q[2 * MAX_BONES * 4];
    __shared__ unsigned long long s_sign[MAX_BONES];
    const int m = blockIdx.y;
    stage_frame(se3_qr, se3_qd, m, B, s_q, s_sign);
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;

    // skinning weights of this surfel, arg-max bone
    float w[MAX_BONES];
    int anchor = 0;
    float best = -1e30f;
#pragma unroll 5
    for (int b = 0; b < B; b++) {
        w[b] = wT[(size_t)b * N + n];
        if (w[b] > best) {  // first maximum, like torch.argmax
            best = w[b];
            anchor = b;
        }
    }
    const unsigned long long hemi = s_sign[anchor];
    Q Qr = {0, 0, 0, 0}, Qd = {0, 0, 0, 0};
    for (int b = 0; b < B; b++) {
        const float ws = ((hemi >> b) & 1ull) ? w[b] : -w[b];
        Qr = qadd(Qr, qscale(ldq(s_q + b * 4), ws));
        Qd = qadd(Qd, qscale(ldq(s_q + MAX_BONES * 4 + b * 4), ws));
    }
    const float inv = 1.0f / sqrtf(qdot(Qr, Qr));
    const Q q = qscale(Qr, inv), d = qscale(Qd, inv);
    const Q tq = qscale(qmul(d, qconj(q)), 2.0f);  // translation = vector part (quat_transform.py:346-352)
    const Q p = qvec(xyz[3 * n], xyz[3 * n + 1], xyz[3 * n + 2]);
    const Q r = ldq(rot + 4 * n);
    Q p1, px;
    rotate(q, p, p1, px);
    const Q xt = qvec(px.x + tq.x, px.y + tq.y, px.z + tq.z);
    const Q rt = qmul(q, r);
    const Q cq = ldq(cam_q + 4 * m);
    const float* ct = cam_t + 3 * m;
    Q c1, cx;
    rotate(cq, xt, c1, cx);
    if (!BACKWARD) {
        const size_t o = (size_t)m * N + n;
        out_xyz[3 * o] = cx.x + ct[0];
        out_xyz[3 * o + 1] = cx.y + ct[1];
        out_xyz[3 * o + 2] = cx.z + ct[2];
        const Q rc = qmul(cq, rt);
        out_rot[4 * o] = rc.w;
        out_rot[4 * o + 1] = rc.x;
        out_rot[4 * o + 2] = rc.y;
        out_rot[4 * o + 3] = rc.z;
        return;
    }
    // ---------------- backward
    const size_t o = (size_t)m * N + n;
    const Q g_xc = qvec(g_out_xyz[3 * o], g_out_xyz[3 * o + 1], g_out_xyz[3 * o + 2]);
    const Q g_rc = ldq(g_out_rot + 4 * o);
    // camera transform: xc = cq xt cq* + ct ; rc = cq * rt   (cq, ct constant)
    Q g_cq_unused, g_xt;
    rotate_bwd(cq, xt, c1, g_xc, g_cq_unused, g_xt);
    g_xt.w = 0.f;
    const Q g_rt = qmul(qconj(cq), g_rc);
    // xt = q p q* + t ; rt = q * r
    Q g_q, g_p;
    rotate_bwd(q, p, p1, g_xt, g_q, g_p);
    g_q = qadd(g_q, qmul(g_rt, qconj(r)));
    const Q g_r = qmul(qconj(q), g_rt);
    // t = 2 vec(d * conj(q))
    const Q g_tq = qscale(g_xt, 2.0f);                 // gradient w.r.t. the quaternion d*conj(q) (vector part)
    const Q g_d = qmul(g_tq, q);                       // g * conj(conj(q))
    g_q = qadd(g_q, qconj(qmul(qconj(d), g_tq)));      // through conj(q)
    // q = Qr / |Qr| ; d = Qd / |Qr|
    const float gq_q = qdot(g_q, q), gd_d = qdot(g_d, d);
    const Q g_Qr = qscale(qadd(g_q, qscale(q, -(gq_q + gd_d))), inv);
    const Q g_Qd = qscale(g_d, inv);
    // weights: Qr = sum_b s_b w_b qr_b (signs constant)
    for (int b = 0; b < B; b++) {
        const float s = ((hemi >> b) & 1ull) ? 1.0f : -1.0f;
        const float gw = s * (qdot(g_Qr, ldq(s_q + b * 4)) + qdot(g_Qd, ldq(s_q + MAX_BONES * 4 + b * 4)));
        g_wT[((size_t)m * B + b) * N + n] = gw;
    }
    g_xyz[3 * o] = g_p.x;
    g_xyz[3 * o + 1] = g_p.y;
    g_xyz[3 * o + 2] = g_p.z;
    g_rot[4 * o] = g_r.w;
    g_rot[4 * o + 1] = g_r.x;
    g_rot[4 * o + 2] = g_r.y;
    g_rot[4 * o + 3] = g_r.z;
}

// ---------------------------------------------------------------------------------------------------------
// Skinning weights + blend + apply in one kernel: per canonical surfel
//   logit_b = -(|x_bone_b|^2 + 0.1 relu(raw_b))      Gaussian-bone distance + delta-skin MLP output
//   w       = softmax_b(logit)                        (skinning.py:89-142, warping.py:415-427)
// followed, for every frame of the step, by the blend / apply / camera transform of lbs_kernel.  In the forward
// warp the weights do not depend on the frame, so ONE thread handles a surfel for all M frames: logits and softmax
// are evaluated once, and the backward accumulates the weight gradients of the frames in registers before it
// goes back through the softmax.  Inputs are feature-major -- xbT (3B, N) bone coordinates, rawT (B, N) raw MLP
// output (or NULL: no delta field) -- so that every load of a wave is coalesced; these are exactly the layouts
// the feature-major GEMMs of the delta MLP produce and consume (lab4d/lbs_fused.py).
constexpr int MAX_FRAMES = 8;

// ---- gradients w.r.t. the bone dual quaternions and the cameras (round 5: networks that train, --gs_optim_warp=True, the
// reference's default, lab4d/config.py:157) are sums over ALL surfels of a frame: per frame B x 8 + 7 numbers.  Each wave
// reduces eight of them at a time with a reduce-scatter in registers -- v_permlane32_swap folds the halves of the wave and
// halves the values per lane, v_permlane16_swap the row pairs, one bank-masked DPP step the half rows, three DPP adds the
// last eight lanes: 17 instructions for 8 sums over 64 lanes (a butterfly per value: 48) -- and the lane that ends up owning
// a value adds it to the workgroup's row in LDS; the workgroup stores its row, and the host adds the rows up (one
// torch.sum over ~800 rows of ~400 floats).  Value index owned by lane L: 4 (L >> 5) + 2 ((L >> 4) & 1) + ((L >> 3) & 1),
// on the lanes with (L & 7) == 0.
// BCAP: compile-time bound of the bone loops (32 for the bob field's 25 bones: fully unrolled, the per-bone weights and
// weight gradients stay in registers; the 64-bone instance indexes them dynamically, i.e. through scratch memory).
// XB_FROM_XYZ: the Gaussian-bone coordinates are not read from xbT but evaluated here, x_bone = A xyz + c with the
// (3B, 3) bone map of the rest pose staged in LDS (225 FMAs per surfel), and the backward folds A^T (d/d x_bone) into the
// centre's gradient instead of writing g_xbT: both kernels are bound by their traffic -- 100 feature-major floats per
// surfel read (twice in the backward) and written -- of which the coordinates are three quarters.
// BX: the bone count when it is known at compile time (25: the reference's "bob" skeleton), 0 = any B <= BCAP.  With BX the
// bone loops are straight-line code: every rawT / weight load of a loop can be in flight at once (the B-dependent `break`s of
// the generic instance put each load behind a branch; the backward then waited for ~80 global loads one after the other --
// 61 % of its wave cycles, rocprofv3 SQ_WAIT_ANY).
#ifndef LBS_BX_WAVES_PER_EU
#define LBS_BX_WAVES_PER_EU 1
#endif
// PIN(...): the named values are final HERE in program order (an empty asm the compiler must feed them through), so what
// went into them is dead behind this point.  Bounds the live ranges inside the straight-line bone loops of the BX
// instances: left alone the compiler defers the serial accumulations (blended quaternions, A^T d x_bone) and keeps the
// products of all 25 bones -- and the LDS rows they came from -- alive: 374 registers; pinned: 227 (backward), 72 (forward).
#define PIN3(a, b_, c)                                              \
    do {                                                            \
        if (BX) asm volatile("" : "+v"(a), "+v"(b_), "+v"(c)); \
    } while (0)
#define PINQ(q_)                                                                         \
    do {                                                                                 \
        if (BX) asm volatile("" : "+v"(q_.w), "+v"(q_.x), "+v"(q_.y), "+v"(q_.z)); \
    } while (0)
// PG (backward, generic instance only): also produce the gradients w.r.t. the frames' bone dual quaternions and cameras --
// per workgroup one row of `g_params`, (M, 8 B + 8) floats: bone b of frame m at [m][8 b .. 8 b + 7] = d/d qr (4), d/d qd (4);
// the camera at [m][8 B .. 8 B + 6] = d/d cam_q (4), d/d cam_t (3).  The hemisphere signs and the anchor bone are piecewise
// constant, as in the reference's torch graph (geom_utils.py:66-74: a comparison and a where).
template <bool BACKWARD, int BCAP, bool XB_FROM_XYZ, int BX, bool PG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BX && BACKWARD) ? LBS_BX_WAVES_PER_EU : 1, 8)))
void lbs_skin_kernel(int M, int N, int B, const float* __restrict__ xbT,
                                                       const float* __restrict__ rawT,
                                                       const float* __restrict__ se3_qr,
                                                       const float* __restrict__ se3_qd,
                                                       const float* __restrict__ xyz, const float* __restrict__ rot,
                                                       const float* __restrict__ cam_q,
                                                       const float* __restrict__ cam_t, float* __restrict__ out_xyz,
                                                       float* __restrict__ out_rot,
                                                       const float* __restrict__ g_out_xyz,
                                                       const float* __restrict__ g_out_rot,
                                                       float* __restrict__ g_xbT, float* __restrict__ g_rawT,
                                                       float* __restrict__ g_xyz, float* __restrict__ g_rot,
                                                       int unit_rot, const float* __restrict__ bone_A,
                                                       const float* __restrict__ bone_c,
                                                       const int64_t* __restrict__ frame_index, int table_rows,
                                                       float* __restrict__ g_params)
{
    static_assert(!PG || (BACKWARD && !BX), "parameter gradients: the generic backward instance");
    __shared__ float s_q[MAX_FRAMES][2 * MAX_BONES * 4];
    __shared__ unsigned long long s_sign[MAX_FRAMES][MAX_BONES];
    __shared__ float4 s_map[XB_FROM_XYZ ? 3 * BCAP : 1];  // row k of the bone map: (A[k][0..2], c[k])
    __shared__ float s_pg[PG ? MAX_FRAMES * (8 * BCAP + 8) : 1];
    const int pg_row = 8 * B + 8;
    if (PG)
        for (int i = threadIdx.x; i < M * pg_row; i += 256) s_pg[i] = 0.f;
    // (a frame id outside the tables -- negative, or beyond the sequence -- reads their first / last row instead of whatever
    // lies behind them: the caller's torch indexing used to raise for it)
    auto table_row = [&](int m) {
        if (!frame_index) return m;
        const long long f = frame_index[m];
        return (int)(f < 0 ? 0 : (f >= table_rows ? table_rows - 1 : f));
    };
    if (XB_FROM_XYZ)
        for (int k = threadIdx.x; k < 3 * B; k += 256)
            s_map[k] = make_float4(bone_A[3 * k], bone_A[3 * k + 1], bone_A[3 * k + 2], bone_c[k]);
    // frame_index: se3_* / cam_* are TABLES over all frames of the sequence, frame m of this call is their row frame_index[m]
    // (the per-step row gathers of the frozen-network tables -- four launches -- happen here instead)
    for (int m = 0; m < M; m++) stage_frame(se3_qr, se3_qd, table_row(m), B, s_q[m], s_sign[m]);
    int n = blockIdx.x * 256 + threadIdx.x;
    // (PG: every thread stays for the workgroup's reductions; one beyond the last surfel works on the last one with weight 0)
    const bool live_thread = n < N;
    if (!PG && !live_thread) return;
    if (PG && !live_thread) n = N - 1;
    const float pg_live = live_thread ? 1.0f : 0.0f;

    const float cx_ = xyz[3 * n], cy_ = xyz[3 * n + 1], cz_ = xyz[3 * n + 2];
    int map_off = 0;  // (made opaque before the backward's last loop: rows re-read from LDS, not kept -- see sq_off below)
    auto bone_coord = [&](int k) {
        if (XB_FROM_XYZ) {
            const float4 r = s_map[k + map_off];
            return fmaf(r.x, cx_, fmaf(r.y, cy_, fmaf(r.z, cz_, r.w)));
        }
        return xbT[(uint32_t)k * (uint32_t)N + (uint32_t)n];
    };
    constexpr int NB = BX ? BX : BCAP;
    float w[BCAP];
    // bit b: the delta-skin logit of bone b is positive (the relu of the forward).  One word per 32 bones: the 64-bone
    // instance needs two (a 32-bit shift by b >= 32 is undefined and aliases bit b - 32 on this hardware)
    using RawMask = typename std::conditional<(NB > 32), uint64_t, uint32_t>::type;
    RawMask raw_pos = 0;
    int anchor = 0;
    float best = -3.0e38f;
    float rawv[BX ? BX : 1];
    if (BX)  // (all global loads of the loop below first)
        _Pragma("unroll") for (int b = 0; b < NB; b++) rawv[b] = rawT ? rawT[(uint32_t)b * (uint32_t)N + (uint32_t)n] : 0.f;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (!BX && b >= B) break;
        const float x0 = bone_coord(3 * b), x1 = bone_coord(3 * b + 1), x2 = bone_coord(3 * b + 2);
        const float raw = BX ? rawv[BX ? b : 0] : (rawT ? rawT[(uint32_t)b * (uint32_t)N + (uint32_t)n] : 0.f);
        raw_pos |= raw > 0.f ? (RawMask(1) << b) : RawMask(0);
        w[b] = -((x0 * x0 + x1 * x1 + x2 * x2) + 0.1f * fmaxf(raw, 0.f));
        if (w[b] > best) {  // first maximum, like torch.argmax (the softmax keeps the order)
            best = w[b];
            anchor = b;
        }
        if (BX) asm volatile("" : "+v"(w[b]), "+v"(best));
    }
    float sum = 0.f;
    _Pragma("unroll") for (int b = 0; b < NB; b++) {
        if (!BX && b >= B) break;
        w[b] = __expf(w[b] - best);
        sum += w[b];
    }
    const float isum = 1.0f / sum;
    _Pragma("unroll") for (int b = 0; b < NB; b++)
        if (BX || b < B) w[b] *= isum;

    const Q p = qvec(cx_, cy_, cz_);
    const Q r = ldq(rot + 4 * n);
    float gw[BCAP];
    Q acc_p = {0, 0, 0, 0}, acc_r = {0, 0, 0, 0};
    if (BACKWARD)
        _Pragma("unroll") for (int b = 0; b < NB; b++) gw[b] = 0.f;

    for (int m = 0; m < M; m++) {
        const float* sq = s_q[m];
        const unsigned long long hemi = s_sign[m][anchor];
        Q Qr = {0, 0, 0, 0}, Qd = {0, 0, 0, 0};
        _Pragma("unroll") for (int b = 0; b < NB; b++) {
            if (!BX && b >= B) break;
            const float ws = ((hemi >> b) & 1ull) ? w[b] : -w[b];
            Qr = qadd(Qr, qscale(ldq(sq + b * 4), ws));
            Qd = qadd(Qd, qscale(ldq(sq + MAX_BONES * 4 + b * 4), ws));
            PINQ(Qr);
            PINQ(Qd);
        }
        const float inv = 1.0f / sqrtf(qdot(Qr, Qr));
        const Q q = qscale(Qr, inv), d = qscale(Qd, inv);
        const Q tq = qscale(qmul(d, qconj(q)), 2.0f);
        Q p1, px;
        rotate(q, p, p1, px);
        const Q xt = qvec(px.x + tq.x, px.y + tq.y, px.z + tq.z);
        const Q rt = qmul(q, r);
        const int mf = table_row(m);
        const Q cq = ldq(cam_q + 4 * mf);
        const float* ct = cam_t + 3 * mf;
        Q c1, cx;
        rotate(cq, xt, c1, cx);
        const size_t o = (size_t)m * N + n;
        if (!BACKWARD) {
            out_xyz[3 * o] = cx.x + ct[0];
            out_xyz[3 * o + 1] = cx.y + ct[1];
            out_xyz[3 * o + 2] = cx.z + ct[2];
            Q rc = qmul(cq, rt);
            if (unit_rot) rc = qscale(rc, 1.0f / fmaxf(sqrtf(qdot(rc, rc)), 1e-12f));  // F.normalize (eps = 1e-12)
            out_rot[4 * o] = rc.w;
            out_rot[4 * o + 1] = rc.x;
            out_rot[4 * o + 2] = rc.y;
            out_rot[4 * o + 3] = rc.z;
            continue;
        }
        const Q g_xc = qvec(g_out_xyz[3 * o], g_out_xyz[3 * o + 1], g_out_xyz[3 * o + 2]);
        Q g_rc = ldq(g_out_rot + 4 * o);
        if (unit_rot) {  // back through v / max(|v|, eps)
            const Q rc = qmul(cq, rt);
            const float nrm = sqrtf(qdot(rc, rc));
            const float invn = 1.0f / fmaxf(nrm, 1e-12f);
            const Q u = qscale(rc, invn);
            g_rc = nrm > 1e-12f ? qscale(qadd(g_rc, qscale(u, -qdot(u, g_rc))), invn) : qscale(g_rc, invn);
        }
        Q g_cq, g_xt;
        rotate_bwd(cq, xt, c1, g_xc, g_cq, g_xt);
        g_xt.w = 0.f;
        const Q g_rt = qmul(qconj(cq), g_rc);
        if (PG) {
            // the camera of frame m: xc = cq xt cq* + ct, rc = cq rt  ->  d/d cq gets the rotation's adjoint plus
            // g_rc conj(rt); d/d ct = g_xc.  Reduced over the wave, added to the workgroup's row.
            g_cq = qadd(g_cq, qmul(g_rc, qconj(rt)));
            float v[8] = {g_cq.w * pg_live, g_cq.x * pg_live, g_cq.y * pg_live, g_cq.z * pg_live,
                          g_xc.x * pg_live, g_xc.y * pg_live, g_xc.z * pg_live, 0.f};
            const float tot = wave_reduce_scatter8(v);
            const int lane = threadIdx.x & 63;
            if ((lane & 7) == 0) atomicAdd(&s_pg[m * pg_row + 8 * B + 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1)], tot);
        }
        Q g_q, g_p;
        rotate_bwd(q, p, p1, g_xt, g_q, g_p);
        g_q = qadd(g_q, qmul(g_rt, qconj(r)));
        acc_r = qadd(acc_r, qmul(qconj(q), g_rt));
        acc_p = qadd(acc_p, g_p);
        const Q g_tq = qscale(g_xt, 2.0f);
        const Q g_d = qmul(g_tq, q);
        g_q = qadd(g_q, qconj(qmul(qconj(d), g_tq)));
        const float gq_q = qdot(g_q, q), gd_d = qdot(g_d, d);
        const Q g_Qr = qscale(qadd(g_q, qscale(q, -(gq_q + gd_d))), inv);
        const Q g_Qd = qscale(g_d, inv);
        // (the bone quaternions are read from LDS AGAIN here: at offsets it can see through the compiler keeps the 200 floats
        // the blend loop above loaded alive across the whole frame body instead.  An opaque OFFSET, not an opaque pointer:
        // that would lose its address space and turn the reads into flat loads)
        int sq_off = 0;
        asm volatile("" : "+v"(sq_off));
        const float* sq2 = sq + sq_off;
        _Pragma("unroll") for (int b = 0; b < NB; b++) {
            if (!BX && b >= B) break;
            const float sgn = ((hemi >> b) & 1ull) ? 1.0f : -1.0f;
            gw[b] += sgn * (qdot(g_Qr, ldq(sq2 + b * 4)) + qdot(g_Qd, ldq(sq2 + MAX_BONES * 4 + b * 4)));
            if (BX && BACKWARD) asm volatile("" : "+v"(gw[b]));
            if (PG) {
                // Qr = sum_b s_b w_b qr_b, Qd likewise: d/d qr_b = s_b w_b g_Qr, d/d qd_b = s_b w_b g_Qd, summed over the surfels
                const float ws = sgn * w[b] * pg_live;
                float v[8] = {ws * g_Qr.w, ws * g_Qr.x, ws * g_Qr.y, ws * g_Qr.z, ws * g_Qd.w, ws * g_Qd.x, ws * g_Qd.y, ws * g_Qd.z};
                const float tot = wave_reduce_scatter8(v);
                const int lane = threadIdx.x & 63;
                if ((lane & 7) == 0) atomicAdd(&s_pg[m * pg_row + 8 * b + 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1)], tot);
            }
        }
    }
    if (PG) {
        __syncthreads();
        float* row = g_params + (size_t)blockIdx.x * (size_t)(M * pg_row);
        for (int i = threadIdx.x; i < M * pg_row; i += 256) row[i] = s_pg[i];
        if (!live_thread) return;
    }
    if (!BACKWARD) return;
    // back through the softmax and the logits.  (Row offsets of the feature-major outputs from an N the compiler cannot
    // see through: the same expressions as the loads' at the top, whose 64-bit addresses it otherwise keeps alive -- two
    // registers per row -- for the whole kernel)
    uint32_t Ns = (uint32_t)N;
    asm volatile("" : "+s"(Ns));
    asm volatile("" : "+v"(map_off));
    float dot = 0.f;
    _Pragma("unroll") for (int b = 0; b < NB; b++)
        if (BX || b < B) dot += w[b] * gw[b];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        if (!BX && b >= B) break;
        const float g_logit = w[b] * (gw[b] - dot);
        const float x0 = bone_coord(3 * b), x1 = bone_coord(3 * b + 1), x2 = bone_coord(3 * b + 2);
        const float g0 = -2.0f * x0 * g_logit, g1 = -2.0f * x1 * g_logit, g2 = -2.0f * x2 * g_logit;
        if (XB_FROM_XYZ) {  // d xyz += A^T d x_bone
            const float4 r0 = s_map[3 * b + map_off], r1 = s_map[3 * b + 1 + map_off], r2 = s_map[3 * b + 2 + map_off];
            acc_p.x += r0.x * g0 + r1.x * g1 + r2.x * g2;
            acc_p.y += r0.y * g0 + r1.y * g1 + r2.y * g2;
            acc_p.z += r0.z * g0 + r1.z * g1 + r2.z * g2;
        } else {
            g_xbT[(uint32_t)(3 * b) * Ns + (uint32_t)n] = g0;
            g_xbT[(uint32_t)(3 * b + 1) * Ns + (uint32_t)n] = g1;
            g_xbT[(uint32_t)(3 * b + 2) * Ns + (uint32_t)n] = g2;
        }
        if (g_rawT) g_rawT[(uint32_t)b * Ns + (uint32_t)n] = ((raw_pos >> b) & RawMask(1)) ? -0.1f * g_logit : 0.f;
        PIN3(acc_p.x, acc_p.y, acc_p.z);
    }
    g_xyz[3 * n] = acc_p.x;
    g_xyz[3 * n + 1] = acc_p.y;
    g_xyz[3 * n + 2] = acc_p.z;
    g_rot[4 * n] = acc_r.w;
    g_rot[4 * n + 1] = acc_r.x;
    g_rot[4 * n + 2] = acc_r.y;
    g_rot[4 * n + 3] = acc_r.z;
}

template <bool BACKWARD, typename... Args>
void launch_lbs_skin(int M, int N, int B, bool from_xyz, bool param_grads, hipStream_t stream, Args... args)
{
    const dim3 grid((N + 255) / 256), block(256);
    if constexpr (BACKWARD) {
        if (param_grads) {   // (bone coordinates from memory: their map's gradient is torch's, through the GEMM that made them)
            if (B <= 32) hipLaunchKernelGGL((lbs_skin_kernel<true, 32, false, 0, true>), grid, block, 0, stream, M, N, B, args...);
            else hipLaunchKernelGGL((lbs_skin_kernel<true, MAX_BONES, false, 0, true>), grid, block, 0, stream, M, N, B, args...);
            return;
        }
    }
#ifndef LBS_BX_FORWARD
#define LBS_BX_FORWARD 0
#endif
    if (B == 25 && from_xyz && (BACKWARD || LBS_BX_FORWARD)) {
        // the reference's "bob" skeleton, frozen bones: straight-line bone loops in the backward (round 3: 76 -> 69 us as
        // they were, 374 registers = one wave per SIMD; 44 us with the accumulators pinned per bone -- 227 registers, two
        // waves).  The forward measures the same either way (25 / 26 us) and keeps the generic instance.
        hipLaunchKernelGGL((lbs_skin_kernel<BACKWARD, 32, true, 25>), grid, block, 0, stream, M, N, B, args...);
    } else if (B <= 32) {
        if (from_xyz) hipLaunchKernelGGL((lbs_skin_kernel<BACKWARD, 32, true, 0>), grid, block, 0, stream, M, N, B, args...);
        else hipLaunchKernelGGL((lbs_skin_kernel<BACKWARD, 32, false, 0>), grid, block, 0, stream, M, N, B, args...);
    } else {
        if (from_xyz) hipLaunchKernelGGL((lbs_skin_kernel<BACKWARD, MAX_BONES, true, 0>), grid, block, 0, stream, M, N, B, args...);
        else hipLaunchKernelGGL((lbs_skin_kernel<BACKWARD, MAX_BONES, false, 0>), grid, block, 0, stream, M, N, B, args...);
    }
}

int check(int M, int N, int B)
{
    if (M < 0 || N < 0 || B <= 0 || B > MAX_BONES) return VIDU4D_E_INVALID;
    // feature-major rows are addressed as (uint32_t)k * N + n with k < 3 B: must stay inside 32 bits
    if ((int64_t)3 * B * (int64_t)N > 0xffffffffll) return VIDU4D_E_INVALID;
    return VIDU4D_OK;
}

}  // namespace

extern "C" int vidu4d_lbs_forward(int M, int N, int B, const float* wT, const float* se3_qr, const float* se3_qd,
                                  const float* xyz, const float* rot, const float* cam_q, const float* cam_t,
                                  float* out_xyz, float* out_rot, void* stream)
{
    if (check(M, N, B)) return VIDU4D_E_INVALID;
    if (M == 0 || N == 0) return VIDU4D_OK;
    if (!wT || !se3_qr || !se3_qd || !xyz || !rot || !cam_q || !cam_t || !out_xyz || !out_rot) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    hipLaunchKernelGGL(lbs_kernel<false>, dim3((N + 255) / 256, M), dim3(256), 0, (hipStream_t)stream, N, B, wT, se3_qr,
                       se3_qd, xyz, rot, cam_q, cam_t, out_xyz, out_rot, nullptr, nullptr, nullptr, nullptr, nullptr);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_lbs_backward(int M, int N, int B, const float* wT, const float* se3_qr, const float* se3_qd,
                                   const float* xyz, const float* rot, const float* cam_q, const float* cam_t,
                                   const float* g_out_xyz, const float* g_out_rot, float* g_wT /*(M,B,N)*/,
                                   float* g_xyz /*(M,N,3)*/, float* g_rot /*(M,N,4)*/, void* stream)
{
    if (check(M, N, B)) return VIDU4D_E_INVALID;
    if (M == 0 || N == 0) return VIDU4D_OK;
    if (!wT || !se3_qr || !se3_qd || !xyz || !rot || !cam_q || !cam_t || !g_out_xyz || !g_out_rot || !g_wT || !g_xyz ||
        !g_rot)
        return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    hipLaunchKernelGGL(lbs_kernel<true>, dim3((N + 255) / 256, M), dim3(256), 0, (hipStream_t)stream, N, B, wT, se3_qr,
                       se3_qd, xyz, rot, cam_q, cam_t, nullptr, nullptr, g_out_xyz, g_out_rot, g_wT, g_xyz, g_rot);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_lbs_skin_forward(int M, int N, int B, const float* xbT, const float* rawT, const float* se3_qr,
                                       const float* se3_qd, const float* xyz, const float* rot, const float* cam_q,
                                       const float* cam_t, float* out_xyz, float* out_rot, int unit_rot,
                                       const float* bone_A, const float* bone_c, const int64_t* frame_index, int table_rows,
                                       void* stream)
{
    if (check(M, N, B) || M > MAX_FRAMES) return VIDU4D_E_INVALID;
    if (M == 0 || N == 0) return VIDU4D_OK;
    if ((bone_A == nullptr) != (bone_c == nullptr) || (bone_A != nullptr) == (xbT != nullptr)) return VIDU4D_E_INVALID;
    if (!se3_qr || !se3_qd || !xyz || !rot || !cam_q || !cam_t || !out_xyz || !out_rot) return VIDU4D_E_INVALID;
    if (frame_index && table_rows <= 0) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    launch_lbs_skin<false>(M, N, B, bone_A != nullptr, false, (hipStream_t)stream, xbT, rawT, se3_qr, se3_qd, xyz, rot, cam_q, cam_t,
                           out_xyz, out_rot, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                           (float*)nullptr, (float*)nullptr, unit_rot, bone_A, bone_c, frame_index, table_rows, (float*)nullptr);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_lbs_skin_backward(int M, int N, int B, const float* xbT, const float* rawT, const float* se3_qr,
                                        const float* se3_qd, const float* xyz, const float* rot, const float* cam_q,
                                        const float* cam_t, const float* g_out_xyz, const float* g_out_rot,
                                        float* g_xbT /*(3B,N)*/, float* g_rawT /*(B,N) or NULL*/, float* g_xyz /*(N,3)*/,
                                        float* g_rot /*(N,4)*/, int unit_rot, const float* bone_A, const float* bone_c,
                                        const int64_t* frame_index, int table_rows, float* g_params, void* stream)
{
    if (check(M, N, B) || M > MAX_FRAMES) return VIDU4D_E_INVALID;
    if (M == 0 || N == 0) return VIDU4D_OK;
    if ((bone_A == nullptr) != (bone_c == nullptr) || (bone_A != nullptr) == (xbT != nullptr)) return VIDU4D_E_INVALID;
    if (!se3_qr || !se3_qd || !xyz || !rot || !cam_q || !cam_t || !g_out_xyz || !g_out_rot || (xbT && !g_xbT) || !g_xyz ||
        !g_rot || (rawT && !g_rawT))
        return VIDU4D_E_INVALID;
    if (frame_index && table_rows <= 0) return VIDU4D_E_INVALID;
    // parameter gradients: with the bone coordinates as an input (their map's gradient then flows through whatever made them)
    if (g_params && bone_A) return VIDU4D_E_UNSUPPORTED;
    (void)hipGetLastError();
    launch_lbs_skin<true>(M, N, B, bone_A != nullptr, g_params != nullptr, (hipStream_t)stream, xbT, rawT, se3_qr, se3_qd, xyz, rot,
                          cam_q, cam_t, (float*)nullptr, (float*)nullptr, g_out_xyz, g_out_rot, g_xbT, g_rawT, g_xyz, g_rot, unit_rot,
                          bone_A, bone_c, frame_index, table_rows, g_params);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_lbs_skin_param_rows(int N) { return (N + 255) / 256; }
