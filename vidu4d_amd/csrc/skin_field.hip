// skin_field.hip -- the per-surfel part of the bob skinning field, evaluated once per optimizer step for all surfels:
//
//     x_bone = A xyz + c                      Gaussian-bone coordinates of the rest pose, 3 per bone
//     raw    = MLP(x_bone | time, instance)   the delta-skin network: D hidden layers of width 64, B outputs
//
// (reference: lab4d/nnutils/skinning.py:89-142 -- SkinningField.forward: gauss_mlp_skinning's bone transform and the
// CondMLP `delta_field` on the bone coordinates; lab4d/nnutils/warping.py:415-427 calls it with the rest articulation and
// the mean time code, so the time / instance part of the first layer's input is one vector for the whole step and
// arrives here folded into that layer's bias.)  Outputs are feature-major -- xbT (3B, N), rawT (B, N) -- which is what
// csrc/lbs.hip (distances, softmax, blend, apply, camera) reads with coalesced loads.
//
// Upstream this is five library GEMMs + activations forward (every hidden activation, 51 MB each at 200k surfels,
// goes through HBM) and as many again backward.  Here one thread carries one surfel through the whole network:
//   * the weights are wave-uniform, so they are fetched with scalar loads and enter v_fma_f32 as the SGPR operand: the
//     inner loops are pure FMA issue (fp32 has the same peak on the vector and the matrix pipes of CDNA4, so MFMA
//     would buy nothing here and would need a layout shuffle between layers);
//   * the hidden vector of a thread lives in a PRIVATE LDS column between layers (the next layer walks it with a
//     uniform index, which registers cannot do): no barrier anywhere;
//   * HBM traffic is the inputs and outputs only (12 B in, 4 (3B + B) B out per surfel).
// The backward recomputes the forward (cheaper than storing it), keeps the ReLU masks as bits and returns the
// gradient with respect to the canonical centres only: bones and network weights are constants of Stage-3
// (--gs_optim_warp=False); a caller that trains them uses the torch path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"

namespace {

constexpr int W = VIDU4D_SKIN_FIELD_WIDTH;       // hidden width
constexpr int IN_MAX = VIDU4D_SKIN_FIELD_IN_MAX;  // padded 3B
constexpr int OUT_MAX = VIDU4D_SKIN_FIELD_OUT_MAX;  // padded B
constexpr int MAX_HIDDEN = VIDU4D_SKIN_FIELD_MAX_HIDDEN;
constexpr int WAVE = 64;

// Weights, biases and bone matrices are read-only for the whole launch and every index into them is wave-uniform:
// viewed through the constant address space they are fetched by the scalar unit (s_load_dwordx8/x16 through the
// scalar cache) and enter the FMAs as SGPR operands, instead of 64 lanes loading the same address.
typedef const float __attribute__((address_space(4))) * UniformPtr;
__device__ __forceinline__ UniformPtr uniform(const float* p) { return (UniformPtr)(uintptr_t)p; }

__device__ __forceinline__ float relu_bit(float v, uint64_t& mask, int j)
{
    const bool on = v > 0.f;
    mask |= on ? (1ull << j) : 0ull;
    return on ? v : 0.f;
}

// acc[j] += sum_k wT[k][j] * lds[k]   (k-major weights, `rows` uniform)
template <int J>
__device__ __forceinline__ void layer_from_lds(float (&acc)[J], const float* w, const float* lds_col, int rows)
{
    UniformPtr wT = uniform(w);
#pragma unroll 1
    for (int k = 0; k < rows; k++) {
        const float hk = lds_col[k * WAVE];
        UniformPtr wk = wT + (size_t)k * J;
#pragma unroll
        for (int j = 0; j < J; j++) acc[j] = fmaf(wk[j], hk, acc[j]);
    }
}

// hidden layers 1..D of the forward; on return the last hidden vector is in the thread's LDS column.
// masks[l] = which units of hidden layer l are active.
__device__ __forceinline__ void hidden_forward(const Vidu4dSkinFieldArgs& a, float x, float y, float z, float* lds_col,
                                               uint64_t (&masks)[MAX_HIDDEN], float* xbT_out, int n)
{
    float acc[W];
    UniformPtr b_in = uniform(a.b_in), A = uniform(a.bone_A), c = uniform(a.bone_c), w_in_T = uniform(a.w_in_T);
#pragma unroll
    for (int j = 0; j < W; j++) acc[j] = b_in[j];
#pragma unroll 1
    for (int k = 0; k < 3 * a.B; k++) {
        const float xb = fmaf(A[3 * k], x, fmaf(A[3 * k + 1], y, fmaf(A[3 * k + 2], z, c[k])));
        if (xbT_out) xbT_out[(size_t)k * a.N + n] = xb;
        UniformPtr wk = w_in_T + (size_t)k * W;
#pragma unroll
        for (int j = 0; j < W; j++) acc[j] = fmaf(wk[j], xb, acc[j]);
    }
    masks[0] = 0;
#pragma unroll
    for (int j = 0; j < W; j++) lds_col[j * WAVE] = relu_bit(acc[j], masks[0], j);
#pragma unroll 1
    for (int l = 1; l < a.D; l++) {
#pragma unroll
        for (int j = 0; j < W; j++) acc[j] = uniform(a.b_hid)[(l - 1) * W + j];
        layer_from_lds<W>(acc, a.w_hid_T + (size_t)(l - 1) * W * W, lds_col, W);
        uint64_t m = 0;
#pragma unroll
        for (int j = 0; j < W; j++) lds_col[j * WAVE] = relu_bit(acc[j], m, j);
        masks[l] = m;
    }
}

__global__ __launch_bounds__(WAVE) void skin_field_fwd_kernel(Vidu4dSkinFieldArgs a)
{
    __shared__ float lds[W * WAVE];
    const int n = blockIdx.x * WAVE + threadIdx.x;
    const bool live = n < a.N;
    const int nn = live ? n : a.N - 1;  // (idle lanes of the last wave repeat its last surfel and store nothing)
    float* lds_col = lds + threadIdx.x;
    const float x = a.xyz[3 * nn], y = a.xyz[3 * nn + 1], z = a.xyz[3 * nn + 2];
    uint64_t masks[MAX_HIDDEN];
    hidden_forward(a, x, y, z, lds_col, masks, live ? a.xbT : nullptr, nn);
    float out[OUT_MAX];
#pragma unroll
    for (int j = 0; j < OUT_MAX; j++) out[j] = uniform(a.b_out)[j];
    layer_from_lds<OUT_MAX>(out, a.w_out_T, lds_col, W);
    if (live) {
#pragma unroll
        for (int j = 0; j < OUT_MAX; j++)
            if (j < a.B) a.rawT[(size_t)j * a.N + n] = out[j];
    }
}

__global__ __launch_bounds__(WAVE) void skin_field_bwd_kernel(Vidu4dSkinFieldArgs a)
{
    __shared__ float lds[W * WAVE];
    const int n = blockIdx.x * WAVE + threadIdx.x;
    const bool live = n < a.N;
    const int nn = live ? n : a.N - 1;
    float* lds_col = lds + threadIdx.x;
    const float x = a.xyz[3 * nn], y = a.xyz[3 * nn + 1], z = a.xyz[3 * nn + 2];
    uint64_t masks[MAX_HIDDEN];
    hidden_forward(a, x, y, z, lds_col, masks, nullptr, nn);

    // d raw -> d h_D : rows of w_out (B, W)
    float g[W];
#pragma unroll
    for (int k = 0; k < W; k++) g[k] = 0.f;
#pragma unroll 1
    for (int j = 0; j < a.B; j++) {
        const float gj = a.g_rawT[(size_t)j * a.N + nn];
        UniformPtr wj = uniform(a.w_out) + (size_t)j * W;
#pragma unroll
        for (int k = 0; k < W; k++) g[k] = fmaf(wj[k], gj, g[k]);
    }
#pragma unroll 1
    for (int l = a.D - 1; l >= 1; l--) {
        const uint64_t m = masks[l];
#pragma unroll
        for (int k = 0; k < W; k++) lds_col[k * WAVE] = ((m >> k) & 1ull) ? g[k] : 0.f;
#pragma unroll
        for (int k = 0; k < W; k++) g[k] = 0.f;
        layer_from_lds<W>(g, a.w_hid + (size_t)(l - 1) * W * W, lds_col, W);  // rows j of w_hid (W, W): sum_j w[j][k] g_j
    }
    {
        const uint64_t m = masks[0];
#pragma unroll
        for (int k = 0; k < W; k++) lds_col[k * WAVE] = ((m >> k) & 1ull) ? g[k] : 0.f;
    }
    // d h_1 -> d x_bone (rows of w_in (W, IN_MAX)), plus what the skinning kernel passes for x_bone itself
    float gx[IN_MAX];
#pragma unroll
    for (int k = 0; k < IN_MAX; k++) gx[k] = (k < 3 * a.B && a.g_xbT) ? a.g_xbT[(size_t)k * a.N + nn] : 0.f;
    layer_from_lds<IN_MAX>(gx, a.w_in, lds_col, W);
    // x_bone = A xyz + c
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    UniformPtr A = uniform(a.bone_A);
#pragma unroll
    for (int k = 0; k < IN_MAX; k++) {
        g0 = fmaf(A[3 * k], gx[k], g0);
        g1 = fmaf(A[3 * k + 1], gx[k], g1);
        g2 = fmaf(A[3 * k + 2], gx[k], g2);
    }
    if (live) {
        a.g_xyz[3 * n] = g0;
        a.g_xyz[3 * n + 1] = g1;
        a.g_xyz[3 * n + 2] = g2;
    }
}

int check_args(const Vidu4dSkinFieldArgs* a, bool backward)
{
    if (!a || a->N < 0 || a->B <= 0 || a->B > OUT_MAX || 3 * a->B > IN_MAX || a->W != W || a->D < 1 || a->D > MAX_HIDDEN)
        return VIDU4D_E_INVALID;
    if (a->N == 0) return VIDU4D_OK;
    if (!a->xyz || !a->bone_A || !a->bone_c || !a->w_in_T || !a->b_in || !a->w_out_T || !a->b_out) return VIDU4D_E_INVALID;
    if (a->D > 1 && (!a->w_hid_T || !a->b_hid)) return VIDU4D_E_INVALID;
    if (!backward) return (a->xbT && a->rawT) ? VIDU4D_OK : VIDU4D_E_INVALID;
    if (!a->w_in || !a->w_out || (a->D > 1 && !a->w_hid) || !a->g_rawT || !a->g_xyz) return VIDU4D_E_INVALID;
    return VIDU4D_OK;
}

}  // namespace

extern "C" int vidu4d_skin_field_forward(const Vidu4dSkinFieldArgs* a, void* stream)
{
    const int rc = check_args(a, false);
    if (rc != VIDU4D_OK || a->N == 0) return rc;
    (void)hipGetLastError();
    hipLaunchKernelGGL(skin_field_fwd_kernel, dim3((a->N + WAVE - 1) / WAVE), dim3(WAVE), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_skin_field_backward(const Vidu4dSkinFieldArgs* a, void* stream)
{
    const int rc = check_args(a, true);
    if (rc != VIDU4D_OK || a->N == 0) return rc;
    (void)hipGetLastError();
    hipLaunchKernelGGL(skin_field_bwd_kernel, dim3((a->N + WAVE - 1) / WAVE), dim3(WAVE), 0, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}
