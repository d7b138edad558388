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
// Upstream this is five library GEMMs + activations forward (every hidden activation, 51 MB each at 200k surfels, goes
// through HBM) and as many again backward.  This is the one GEMM-shaped piece of the hot path, so it runs on the matrix
// cores: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: bitwise an fmaf chain, no precision trade).
//
//   * A wave owns a tile of 32 surfels = the 32 columns of every MFMA; a layer's 64 outputs are two 32-row blocks.
//   * The C/D layout of this MFMA puts row (v&3) + 8 (v>>2) of a block into register v of lanes 0-31 and that row + 4
//     into the same register of lanes 32-63 -- exactly the shape of a B operand (lanes 0-31: k0, lanes 32-63: k1) for
//     the k-pair (row, row + 4).  The contraction order is free, so the NEXT layer simply contracts over these pairs:
//     the accumulator registers of one layer (after bias + ReLU) ARE the B operands of the next.  No transposition, no
//     LDS round trip, no cross-lane traffic between layers, forward or backward.
//   * The weights are the A operands.  Each workgroup stages them once into LDS already arranged per MFMA ("stream"
//     m holds, for lane l, the weight of output row l&31 and the contraction index that lane half l>>5 supplies), so an
//     A operand is one conflict-free ds_read_b32; the 16 waves of a workgroup share them and loop over tiles.
//   * HBM traffic is inputs and outputs only: 12 B in, 4 (3B + B) B out per surfel.
// (A first version carried one surfel per thread with scalar-loaded weights as SGPR FMA operands; at 2.5 waves per SIMD
// its 40 KB weight stream per wave through the scalar cache held it to 18 TFLOP/s -- 244 us forward at 200k surfels.)
//
// The backward recomputes the forward (cheaper than storing it), keeps the ReLU masks as bits and returns the gradient
// with respect to the canonical centres only: bones and network weights are constants of Stage-3 (--gs_optim_warp=False);
// (round 5) a caller that trains them gets TRAIN instances, which additionally leave every hidden layer's activations (forward:
// h_store) and masked pre-activation gradients (backward: g_store) and the whole gradient w.r.t. the bone coordinates
// (gx_store) in feature-major arrays, from which the weight gradients are contractions over the surfels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"

namespace {

constexpr int W = VIDU4D_SKIN_FIELD_WIDTH;          // hidden width: two row blocks of 32
constexpr int IN_MAX = VIDU4D_SKIN_FIELD_IN_MAX;    // padded 3B: three row blocks
constexpr int OUT_MAX = VIDU4D_SKIN_FIELD_OUT_MAX;  // padded B: one row block
constexpr int MAX_HIDDEN = VIDU4D_SKIN_FIELD_MAX_HIDDEN;
// waves per workgroup share the staged weights: 16 (4 per SIMD, <= 128 registers each) forward, 12 (3 per SIMD, <= 168
// registers: no spills since the row addresses stopped being hoisted) backward
#ifndef SKIN_BWD_THREADS
#define SKIN_BWD_THREADS 768
#endif
template <bool BACKWARD>
constexpr int threads_of() { return BACKWARD ? SKIN_BWD_THREADS : 1024; }
static_assert(W == 64 && IN_MAX == 96 && OUT_MAX == 32, "row-block structure");

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int rowv(int v, int half) { return (v & 3) + 8 * (v >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Schedule hint for the unrolled layer loops: N LDS reads (the A operands of the next step), then N MFMAs, repeated.
// Without it the scheduler hoists all 64-96 operand reads of a layer to its top and spills the accumulators.  (Reading two
// steps ahead instead of one measures the same: a lone wave's MFMA pair already covers the LDS round trip.)
template <int N>
__device__ __forceinline__ void interleave_reads_and_mfmas()
{
    __builtin_amdgcn_sched_group_barrier(0x100, N, 0);  // DS read
    __builtin_amdgcn_sched_group_barrier(0x008, N, 0);  // MFMA
}

// LDS plan (floats).  Streams are [m][64 lanes].
struct Plan {
    int T1, T3;            // k-pairs of the first layer (ceil(3B / 2)) and of the output gradient (ceil(B / 2))
    int s_in, s_hid, s_out;          // forward streams
    int t_out, t_hid, t_in;          // backward streams
    int bias, bone;                  // (D + 1) * 64 bias floats; IN_MAX float4 bone rows
    int total;
};

__host__ __device__ inline Plan make_plan(int B, int D, bool backward)
{
    Plan p;
    p.T1 = (3 * B + 1) / 2;
    p.T3 = (B + 1) / 2;
    int o = 0;
    p.s_in = o;
    o += p.T1 * 2 * 64;
    p.s_hid = o;
    o += (D - 1) * 64 * 64;
    p.s_out = o;
    o += backward ? 0 : 32 * 64;
    p.t_out = o;
    o += backward ? p.T3 * 2 * 64 : 0;
    p.t_hid = o;
    o += backward ? (D - 1) * 64 * 64 : 0;
    p.t_in = o;
    o += backward ? 32 * 3 * 64 : 0;
    p.bias = o;
    o += (D + 1) * 64;
    p.bone = o;
    o += IN_MAX * 4;
    p.total = o;
    return p;
}

// Fills `lds` (LDS, or -- vidu4d_skin_field_pack -- a global buffer that later launches copy into LDS linearly: the
// gather below costs a workgroup ~20 us, a third of a forward at 200k surfels, and the weights are constants).
template <bool BACKWARD>
__device__ void stage(const Vidu4dSkinFieldArgs& a, const Plan& p, float* lds)
{
    constexpr int THREADS = threads_of<BACKWARD>();
    const int B3 = 3 * a.B;
    for (int e = threadIdx.x; e < p.T1 * 2 * 64; e += THREADS) {
        const int l = e & 63, m = e >> 6, ob = m & 1, t = m >> 1;
        const int k = t + p.T1 * (l >> 5);
        lds[p.s_in + e] = k < B3 ? a.w_in[(32 * ob + (l & 31)) * IN_MAX + k] : 0.f;
    }
    for (int e = threadIdx.x; e < (a.D - 1) * 64 * 64; e += THREADS) {
        const int l = e & 63, m = (e >> 6) & 63, layer = e >> 12, ob = m & 1, sv = m >> 1;
        const int src = 32 * (sv >> 4) + rowv(sv & 15, l >> 5), out = 32 * ob + (l & 31);
        const float* w = a.w_hid + (size_t)layer * W * W;
        lds[p.s_hid + e] = w[out * W + src];
        if (BACKWARD) lds[p.t_hid + e] = w[src * W + out];
    }
    if (!BACKWARD) {
        for (int e = threadIdx.x; e < 32 * 64; e += THREADS) {
            const int l = e & 63, sv = e >> 6;
            lds[p.s_out + e] = a.w_out[(l & 31) * W + 32 * (sv >> 4) + rowv(sv & 15, l >> 5)];  // (rows >= B are zero)
        }
    } else {
        for (int e = threadIdx.x; e < p.T3 * 2 * 64; e += THREADS) {
            const int l = e & 63, m = e >> 6, ob = m & 1, t = m >> 1;
            const int j = t + p.T3 * (l >> 5);
            lds[p.t_out + e] = j < a.B ? a.w_out[j * W + 32 * ob + (l & 31)] : 0.f;
        }
        for (int e = threadIdx.x; e < 32 * 3 * 64; e += THREADS) {
            const int l = e & 63, m = e >> 6, ob = m % 3, sv = m / 3;
            lds[p.t_in + e] = a.w_in[(32 * (sv >> 4) + rowv(sv & 15, l >> 5)) * IN_MAX + 32 * ob + (l & 31)];
        }
    }
    for (int e = threadIdx.x; e < (a.D + 1) * 64; e += THREADS) {
        const int layer = e >> 6, j = e & 63;
        lds[p.bias + e] = layer == 0 ? (a.b_in ? a.b_in[j] : 0.f)
                                     : (layer < a.D ? a.b_hid[(layer - 1) * W + j] : (j < OUT_MAX ? a.b_out[j] : 0.f));
    }
    for (int e = threadIdx.x; e < IN_MAX; e += THREADS) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < B3) r = make_float4(a.bone_A[3 * e], a.bone_A[3 * e + 1], a.bone_A[3 * e + 2], a.bone_c[e]);
        reinterpret_cast<float4*>(lds + p.bone)[e] = r;
    }
    __syncthreads();
}

// Row offsets of the feature-major arrays are row * N + surfel.  N is a kernel argument and the rows of a lane are
// compile-time patterns, so the compiler hoists every row's 64-bit address out of the tile loop -- 32 registers for the
// forward's output rows, 96 for the backward's 48 gradient rows -- and spills.  An N it cannot see through is multiplied
// again per tile (scalar unit) and the addresses stay 32-bit offsets until they are used.
__device__ __forceinline__ uint32_t opaque_uniform(uint32_t v)
{
    asm volatile("" : "+s"(v));
    return v;
}

// acc = bias of `layer` in the D layout
__device__ __forceinline__ void load_bias(f32x16 (&acc)[2], const float* lds, const Plan& p, int layer, int half)
{
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[ob][v] = lds[p.bias + layer * 64 + 32 * ob + rowv(v, half)];
}

__device__ __forceinline__ uint32_t relu_tiles(f32x16 (&h)[2])
{
    uint32_t mask = 0;
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int v = 0; v < 16; v++) {
            const bool on = h[ob][v] > 0.f;
            mask |= on ? (1u << (16 * ob + v)) : 0u;
            h[ob][v] = on ? h[ob][v] : 0.f;
        }
    return mask;
}

// Networks that train (TRAIN instances): a layer's 64 rows of this tile to a feature-major (layers, 64, N) array -- the
// activations from the forward, the pre-activation gradients from the backward -- from which the weight gradients are taken
// as contractions over the surfels (lab4d/lbs_fused.py).  Lanes 0-31 write 128 contiguous bytes of a row, lanes 32-63 of
// the row four further down.
__device__ __forceinline__ void store_rows(float* dst, int first_row, const f32x16 (&t)[2], int half, uint32_t Ns, int n)
{
#pragma unroll
    for (int ob = 0; ob < 2; ob++)
#pragma unroll
        for (int v = 0; v < 16; v++)
            dst[(uint32_t)(first_row + 32 * ob + rowv(v, half)) * (size_t)Ns + (uint32_t)n] = t[ob][v];
}

// Hidden layers of the forward for one tile; h = last hidden activations (D layout), masks[l] = active units of layer l.
__device__ __forceinline__ void hidden_forward(const Vidu4dSkinFieldArgs& a, const Plan& p, const float* lds, int lane,
                                               int n, uint32_t Ns, bool valid, float x, float y, float z, float* xbT_out,
                                               f32x16 (&h)[2], uint32_t (&masks)[MAX_HIDDEN], uint32_t* mask_out,
                                               size_t mask_stride, float* h_store = nullptr)
{
    const int half = lane >> 5;
    load_bias(h, lds, p, 0, half);
    // (bone rows and first-layer weights beyond 3B are staged as zeros, rows up to IN_MAX exist: no conditions on k.  Two
    // k-pairs per trip, all eight LDS reads of a trip in flight before the first use -- a trip per k-pair waited for two
    // LDS round trips in sequence and made the 76 MFMAs of this layer take as long as the whole tile's matrix time)
    const float4* bone = reinterpret_cast<const float4*>(lds + p.bone);
    const int B3 = 3 * a.B;
    for (int t = 0; t < p.T1; t += 2) {
        const bool two = t + 1 < p.T1;
        const int ta = t, tb = two ? t + 1 : t;
        const int ka = ta + p.T1 * half, kb = tb + p.T1 * half;
        const float4 ba = bone[ka], bb = bone[kb];
        const float* sa = lds + p.s_in + (ta * 2) * 64 + lane;
        const float* sb = lds + p.s_in + (tb * 2) * 64 + lane;
        const float wa0 = sa[0], wa1 = sa[64], wb0 = sb[0], wb1 = sb[64];
        const float xa = fmaf(ba.x, x, fmaf(ba.y, y, fmaf(ba.z, z, ba.w)));
        const float xb = two ? fmaf(bb.x, x, fmaf(bb.y, y, fmaf(bb.z, z, bb.w))) : 0.f;
        if (xbT_out && valid) {
            if (ka < B3) xbT_out[(uint32_t)ka * Ns + (uint32_t)n] = xa;
            if (two && kb < B3) xbT_out[(uint32_t)kb * Ns + (uint32_t)n] = xb;
        }
        h[0] = mfma(wa0, xa, h[0]);
        h[1] = mfma(wa1, xa, h[1]);
        h[0] = mfma(wb0, xb, h[0]);
        h[1] = mfma(wb1, xb, h[1]);
    }
    masks[0] = relu_tiles(h);
    if (mask_out) mask_out[0] = masks[0];
    // (h_store_rows: rows per layer of the caller's array, W or more -- e.g. W + 1 with a row of ones behind each layer)
    const int h_rows = a.h_store_rows > 0 ? a.h_store_rows : W;
    if (h_store && valid) store_rows(h_store, 0, h, half, Ns, n);
#pragma unroll 1
    for (int layer = 1; layer < a.D; layer++) {
        f32x16 acc[2];
        load_bias(acc, lds, p, layer, half);
        const float* s = lds + p.s_hid + (layer - 1) * 64 * 64 + lane;
#pragma unroll
        for (int sb = 0; sb < 2; sb++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const float b = h[sb][v];
                acc[0] = mfma(s[((sb * 16 + v) * 2) * 64], b, acc[0]);
                acc[1] = mfma(s[((sb * 16 + v) * 2 + 1) * 64], b, acc[1]);
                interleave_reads_and_mfmas<2>();
            }
        masks[layer] = relu_tiles(acc);
        if (mask_out) mask_out[layer * mask_stride] = masks[layer];
        h[0] = acc[0];
        h[1] = acc[1];
        if (h_store && valid) store_rows(h_store, layer * h_rows, h, half, Ns, n);
    }
}

template <bool BACKWARD>
__global__ __launch_bounds__(threads_of<BACKWARD>()) void skin_field_pack_kernel(Vidu4dSkinFieldArgs a, float* out)
{
    const Plan p = make_plan(a.B, a.D, BACKWARD);
    stage<BACKWARD>(a, p, out);
}

template <bool BACKWARD, bool TRAIN>
__global__ __launch_bounds__(threads_of<BACKWARD>()) void skin_field_kernel(Vidu4dSkinFieldArgs a)
{
    constexpr int THREADS = threads_of<BACKWARD>();
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Plan p = make_plan(a.B, a.D, BACKWARD);
    const float* packed = BACKWARD ? a.packed_bwd : a.packed_fwd;
    if (packed) {
        // the image vidu4d_skin_field_pack made of everything but the step's first-layer bias (every section of the plan
        // is a multiple of 4 floats)
        const int bias4 = p.bias / 4;
        for (int e = threadIdx.x; e < p.total / 4; e += THREADS) {
            float4 v = reinterpret_cast<const float4*>(packed)[e];
            if (e >= bias4 && e < bias4 + W / 4) {
                const float* b = a.b_in + 4 * (e - bias4);
                v = make_float4(b[0], b[1], b[2], b[3]);
            }
            reinterpret_cast<float4*>(lds)[e] = v;
        }
        __syncthreads();
    } else {
        stage<BACKWARD>(a, p, lds);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int tiles = (a.N + 31) / 32;
    // (tile = (round * waves + wave) * workgroups + workgroup: a partial last round is spread over all CUs and SIMDs)
    for (int tile = wave * gridDim.x + blockIdx.x; tile < tiles; tile += gridDim.x * (THREADS / 64)) {
        const int n = tile * 32 + (lane & 31);
        const bool valid = n < a.N;
        const int nn = valid ? n : a.N - 1;
        const uint32_t Ns = opaque_uniform((uint32_t)a.N);  // (rows * N < 2^32: check_args)
        const float x = a.xyz[3 * nn], y = a.xyz[3 * nn + 1], z = a.xyz[3 * nn + 2];
        f32x16 h[2];
        uint32_t masks[MAX_HIDDEN];
        // relu_masks[layer][tile][lane]: with them the backward needs no forward at all
        uint32_t* mask_slot = a.relu_masks ? a.relu_masks + (size_t)tile * 64 + lane : nullptr;
        const size_t mask_layer_stride = (size_t)tiles * 64;
        if (BACKWARD && mask_slot) {
            for (int l = 0; l < a.D; l++) masks[l] = mask_slot[l * mask_layer_stride];
        } else {
            hidden_forward(a, p, lds, lane, n, Ns, valid, x, y, z, BACKWARD ? nullptr : a.xbT, h, masks,
                           BACKWARD ? nullptr : mask_slot, mask_layer_stride, (TRAIN && !BACKWARD) ? a.h_store : nullptr);
        }
        if (!BACKWARD) {
            f32x16 out;
#pragma unroll
            for (int v = 0; v < 16; v++) out[v] = lds[p.bias + a.D * 64 + rowv(v, half)];
            const float* s = lds + p.s_out + lane;
#pragma unroll
            for (int sb = 0; sb < 2; sb++)
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    out = mfma(s[(sb * 16 + v) * 64], h[sb][v], out);
                    interleave_reads_and_mfmas<1>();
                }
            if (valid) {
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const int j = rowv(v, half);
                    if (j < a.B) a.rawT[(uint32_t)j * Ns + (uint32_t)n] = out[v];
                }
            }
            continue;
        }
        // ---- backward: d raw -> d h_D
        f32x16 g[2];
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int v = 0; v < 16; v++) g[ob][v] = 0.f;
        {
            // (the upstream gradients of the tile first, all loads in flight: one global round trip, not T3 of them)
            constexpr int T3_MAX = (OUT_MAX + 1) / 2;
            float gj[T3_MAX];
#pragma unroll
            for (int t = 0; t < T3_MAX; t++) {
                const int j = t + p.T3 * half;
                gj[t] = (t < p.T3 && j < a.B) ? a.g_rawT[(uint32_t)j * Ns + (uint32_t)nn] : 0.f;
            }
            const float* s = lds + p.t_out + lane;
#pragma unroll
            for (int t = 0; t < T3_MAX; t++) {
                if (t < p.T3) {  // (uniform)
                    g[0] = mfma(s[(t * 2) * 64], gj[t], g[0]);
                    g[1] = mfma(s[(t * 2 + 1) * 64], gj[t], g[1]);
                }
            }
        }
#pragma unroll 1
        for (int layer = a.D - 1; layer >= 1; layer--) {
            const uint32_t m = masks[layer];
            f32x16 acc[2];
#pragma unroll
            for (int ob = 0; ob < 2; ob++)
#pragma unroll
                for (int v = 0; v < 16; v++) acc[ob][v] = 0.f;
            const float* s = lds + p.t_hid + (layer - 1) * 64 * 64 + lane;
            if (TRAIN) {   // (the masked gradient once, in place: what the MFMAs below contract is what is stored)
#pragma unroll
                for (int sb = 0; sb < 2; sb++)
#pragma unroll
                    for (int v = 0; v < 16; v++) g[sb][v] = ((m >> (16 * sb + v)) & 1u) ? g[sb][v] : 0.f;
                if (valid) store_rows(a.g_store, layer * W, g, half, Ns, n);
            }
#pragma unroll
            for (int sb = 0; sb < 2; sb++)
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const float b = ((m >> (16 * sb + v)) & 1u) ? g[sb][v] : 0.f;
                    acc[0] = mfma(s[((sb * 16 + v) * 2) * 64], b, acc[0]);
                    acc[1] = mfma(s[((sb * 16 + v) * 2 + 1) * 64], b, acc[1]);
                    interleave_reads_and_mfmas<2>();
                }
            g[0] = acc[0];
            g[1] = acc[1];
        }
        // ---- d h_1 -> d x_bone (three row blocks), starting from what the skinning kernel passes for x_bone itself
        f32x16 gx[3];
#pragma unroll
        for (int ob = 0; ob < 3; ob++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int k = 32 * ob + rowv(v, half);
                gx[ob][v] = (a.g_xbT && k < 3 * a.B) ? a.g_xbT[(uint32_t)k * Ns + (uint32_t)nn] : 0.f;
            }
        {
            const uint32_t m = masks[0];
            const float* s = lds + p.t_in + lane;
            if (TRAIN) {
#pragma unroll
                for (int sb = 0; sb < 2; sb++)
#pragma unroll
                    for (int v = 0; v < 16; v++) g[sb][v] = ((m >> (16 * sb + v)) & 1u) ? g[sb][v] : 0.f;
                if (valid) store_rows(a.g_store, 0, g, half, Ns, n);
            }
#pragma unroll
            for (int sb = 0; sb < 2; sb++)
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const float b = ((m >> (16 * sb + v)) & 1u) ? g[sb][v] : 0.f;
#pragma unroll
                    for (int ob = 0; ob < 3; ob++) gx[ob] = mfma(s[((sb * 16 + v) * 3 + ob) * 64], b, gx[ob]);
                    interleave_reads_and_mfmas<3>();
                }
        }
        if (TRAIN && valid) {   // the whole gradient w.r.t. the bone coordinates: d/dA = gx xyz^T, d/dc = sum gx
#pragma unroll
            for (int ob = 0; ob < 3; ob++)
#pragma unroll
                for (int v = 0; v < 16; v++) {
                    const int k = 32 * ob + rowv(v, half);
                    if (k < 3 * a.B) a.gx_store[(uint32_t)k * (size_t)Ns + (uint32_t)n] = gx[ob][v];
                }
        }
        // ---- x_bone = A xyz + c: d xyz = A^T d x_bone; this lane holds 48 of the surfel's rows, its partner the others
        const float4* bone = reinterpret_cast<const float4*>(lds + p.bone);
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int ob = 0; ob < 3; ob++)
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const float4 bc = bone[32 * ob + rowv(v, half)];  // (rows >= 3B are zero)
                g0 = fmaf(bc.x, gx[ob][v], g0);
                g1 = fmaf(bc.y, gx[ob][v], g1);
                g2 = fmaf(bc.z, gx[ob][v], g2);
            }
        g0 += __shfl_xor(g0, 32, 64);
        g1 += __shfl_xor(g1, 32, 64);
        g2 += __shfl_xor(g2, 32, 64);
        if (valid && half == 0) {
            a.g_xyz[3 * n] = g0;
            a.g_xyz[3 * n + 1] = g1;
            a.g_xyz[3 * n + 2] = g2;
        }
    }
}

int check_args(const Vidu4dSkinFieldArgs* a, bool backward)
{
    if (!a || a->N < 0 || (int64_t)a->N * IN_MAX > 0x7fffffffll || a->B <= 0 || a->B > OUT_MAX || 3 * a->B > IN_MAX || a->W != W || a->D < 1 || a->D > MAX_HIDDEN)
        return VIDU4D_E_INVALID;
    if (a->N == 0) return VIDU4D_OK;
    if (!a->xyz || !a->bone_A || !a->bone_c || !a->w_in || !a->b_in || !a->w_out || !a->b_out) return VIDU4D_E_INVALID;
    if (a->D > 1 && (!a->w_hid || !a->b_hid)) return VIDU4D_E_INVALID;
    if (a->h_store_rows != 0 && a->h_store_rows < W) return VIDU4D_E_INVALID;
    if (!backward) return a->rawT ? VIDU4D_OK : VIDU4D_E_INVALID;  // (xbT may be NULL: a caller that evaluates the bone map itself)
    if ((a->g_store != nullptr) != (a->gx_store != nullptr)) return VIDU4D_E_INVALID;
    if (a->g_store && !a->relu_masks) return VIDU4D_E_INVALID;   // (the stored gradients are the masked ones)
    return (a->g_rawT && a->g_xyz) ? VIDU4D_OK : VIDU4D_E_INVALID;
}

template <bool BACKWARD, bool TRAIN>
int launch_instance(const Vidu4dSkinFieldArgs* a, void* stream);

template <bool BACKWARD>
int launch(const Vidu4dSkinFieldArgs* a, void* stream)
{
    const int rc = check_args(a, BACKWARD);
    if (rc != VIDU4D_OK || a->N == 0) return rc;
    const bool train = BACKWARD ? a->g_store != nullptr : a->h_store != nullptr;
    return train ? launch_instance<BACKWARD, true>(a, stream) : launch_instance<BACKWARD, false>(a, stream);
}

template <bool BACKWARD, bool TRAIN>
int launch_instance(const Vidu4dSkinFieldArgs* a, void* stream)
{
    const Plan p = make_plan(a->B, a->D, BACKWARD);
    const size_t bytes = (size_t)p.total * sizeof(float);
    static bool attr_set = false;  // (one attribute call per template instance; idempotent, benign if raced)
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&skin_field_kernel<BACKWARD, TRAIN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return VIDU4D_E_HIP;
        attr_set = true;
    }
    if (bytes > 160 * 1024) return VIDU4D_E_UNSUPPORTED;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    constexpr int THREADS = threads_of<BACKWARD>();
    const int tiles = (a->N + 31) / 32, per_wg = THREADS / 64;
    int grid = (tiles + per_wg - 1) / per_wg;
    if (grid > cus) grid = cus;  // one resident workgroup per CU, waves loop over tiles
    (void)hipGetLastError();
    hipLaunchKernelGGL((skin_field_kernel<BACKWARD, TRAIN>), dim3(grid), dim3(THREADS), bytes, (hipStream_t)stream, *a);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

template <bool BACKWARD>
int pack(const Vidu4dSkinFieldArgs* a, float* out, void* stream)
{
    if (!a || a->B <= 0 || a->B > OUT_MAX || 3 * a->B > IN_MAX || a->W != W || a->D < 1 || a->D > MAX_HIDDEN || !out)
        return VIDU4D_E_INVALID;
    if (!a->bone_A || !a->bone_c || !a->w_in || !a->w_out || !a->b_out || (a->D > 1 && (!a->w_hid || !a->b_hid)))
        return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    hipLaunchKernelGGL(skin_field_pack_kernel<BACKWARD>, dim3(1), dim3(threads_of<BACKWARD>()), 0, (hipStream_t)stream, *a, out);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

}  // namespace

extern "C" int vidu4d_skin_field_packed_floats(int B, int D, int backward)
{
    if (B <= 0 || B > OUT_MAX || 3 * B > IN_MAX || D < 1 || D > MAX_HIDDEN) return 0;
    return make_plan(B, D, backward != 0).total;
}

extern "C" int vidu4d_skin_field_pack(const Vidu4dSkinFieldArgs* a, int backward, float* out, void* stream)
{
    return backward ? pack<true>(a, out, stream) : pack<false>(a, out, stream);
}

extern "C" int vidu4d_skin_field_forward(const Vidu4dSkinFieldArgs* a, void* stream) { return launch<false>(a, stream); }

extern "C" int vidu4d_skin_field_backward(const Vidu4dSkinFieldArgs* a, void* stream) { return launch<true>(a, stream); }
