// optim.hip -- what the Stage-3 loop does to the canonical surfel set between two rasterizer calls, on the device:
//
//   * the surfel optimizer's update: torch.optim.Adam(eps = 1e-15) with one parameter group per attribute
//     (reference: lab4d/engine/trainer.py:240-255) is 8 multi-tensor launches of ~40 us each for 200k surfels because
//     every group has its own learning rate; here all groups are updated by ONE launch (12M elements, 7 streams of
//     4 bytes each: ~330 MB -> HBM-bound, ~50 us);
//   * densify_and_prune (reference: gs/scene/gaussian_model.py:384-448, with the optimizer surgery :270-356):
//     clone / split / prune decisions per surfel, then ONE gather that writes the new parameter set and the new Adam
//     moments of all attributes in the reference's row order
//         [surviving originals | surviving clones | first split copies | second split copies]
//     instead of ~150 indexing / cat launches and ~10 host round trips (boolean-mask indexing waits for its count).
//
// Upstream behaviours kept on purpose (all pinned by tests/golden/refpy_densify.npz through the Python path these
// kernels are tested against): the screen-size criterion never fires because densification_postfix has just zeroed
// max_radii2D (:371-373 before :443); clones are never split (their padded gradient is 0, :386-387); a clone /
// split copy starts with zero moments while a surviving original keeps its moments (:336-356).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"

namespace {

// ------------------------------------------------------------------------------------------------ Adam
template <int MAXT>
struct AdamLaunchT {
    Vidu4dAdamTensor t[MAXT];
    unsigned first_block[MAXT + 1];
    int n;
    float beta2, one_minus_beta1, one_minus_beta2, eps;
    float weight_decay;     // AdamW's decoupled decay (vidu4d_adamw_step_guarded); 0: Adam
    const float* grad_scale;
    int zero_grads;
    const uint32_t* skip;   // device word: non-zero = this launch changes nothing (vidu4d_adam_step_guarded)
};
using AdamLaunch = AdamLaunchT<VIDU4D_ADAM_MAX_TENSORS>;

constexpr int ADAM_PER_THREAD = 4;
constexpr int ADAM_PER_BLOCK = 256 * ADAM_PER_THREAD;

// The update of torch's Adam (torch/optim/adam.py _single_tensor_adam, non-amsgrad, no weight decay):
//   m <- lerp(m, g, 1 - beta1);  v <- beta2 v + (1 - beta2) g g;  p <- p - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// AdamW (vidu4d_adamw_step_guarded; torch's fused kernel, ATen/native/cuda/fused_adam_utils.cuh adam_math in ADAMW mode): the
// same with p <- p - lr * weight_decay * p in front.  The networks of the warp / camera fields are 66 small tensors: their
// launch struct holds 32.
template <int MAXT>
__global__ __launch_bounds__(256) void adam_kernel(AdamLaunchT<MAXT> a)
{
    int k = 0;
    if (MAXT <= 8) {
#pragma unroll
        for (int i = 1; i < MAXT; i++)
            if (i < a.n && blockIdx.x >= a.first_block[i]) k = i;
    } else {   // (first_block ascends: bisection)
        int lo = 0, hi = a.n;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (blockIdx.x >= a.first_block[mid]) lo = mid;
            else hi = mid;
        }
        k = lo;
    }
    if (a.skip && *a.skip) return;   // (wave-uniform: the step's forward did not fit its buffers, its gradients are garbage)
    const Vidu4dAdamTensor t = a.t[k];
    // (captured launches read the step's scalars from device memory: same float values the host would have passed by value)
    const float lr = t.device_scalars ? t.device_scalars[0] : t.lr;
    const float bc1 = t.device_scalars ? t.device_scalars[1] : t.bias_correction1;
    const float bc2_sqrt = t.device_scalars ? t.device_scalars[2] : t.bias_correction2_sqrt;
    const float step_size = lr / bc1;
    const float w = a.one_minus_beta1, w2 = a.one_minus_beta2;
    const float gs = a.grad_scale ? *a.grad_scale : 1.0f;
    const int64_t base = (int64_t)(blockIdx.x - a.first_block[k]) * ADAM_PER_BLOCK + threadIdx.x;
    float g[ADAM_PER_THREAD], m[ADAM_PER_THREAD], v[ADAM_PER_THREAD], p[ADAM_PER_THREAD];
#pragma unroll
    for (int j = 0; j < ADAM_PER_THREAD; j++) {
        const int64_t e = base + j * 256;
        if (e < t.numel) {
            g[j] = a.grad_scale ? t.grad[e] * gs : t.grad[e];
            m[j] = t.exp_avg[e];
            v[j] = t.exp_avg_sq[e];
            p[j] = t.param[e];
        }
    }
#pragma unroll
    for (int j = 0; j < ADAM_PER_THREAD; j++) {
        const int64_t e = base + j * 256;
        if (e < t.numel) {
            const float mn = fmaf(w, g[j] - m[j], m[j]);  // lerp with weight < 0.5 (ATen/native/Lerp.h)
            const float vn = fmaf(w2 * g[j], g[j], a.beta2 * v[j]);
            const float denom = sqrtf(vn) / bc2_sqrt + a.eps;
            if (a.zero_grads) t.grad[e] = 0.f;
            t.exp_avg[e] = mn;
            t.exp_avg_sq[e] = vn;
            const float pd = a.weight_decay != 0.f ? p[j] - lr * a.weight_decay * p[j] : p[j];
            t.param[e] = pd - step_size * (mn / denom);
        }
    }
}

// ------------------------------------------------------------------------------------------------ gradient clip
constexpr int CLIP_BLOCKS = 512, CLIP_THREADS = 512;
constexpr int CLIP_L1 = 16;          // first-level arrival counters (CLIP_BLOCKS / CLIP_L1 blocks each), one cache line apart
constexpr int CLIP_PARTIALS = 32 + CLIP_L1 * 16;  // workspace offset of the block partials
struct ClipLaunch {
    const float* g[VIDU4D_CLIP_MAX_TENSORS];
    int64_t numel[VIDU4D_CLIP_MAX_TENSORS];
    int n;
    float max_norm;
    float* workspace;  // [0] final arrival counter, [32 + 16 i] first-level counters (as unsigned), then the block partials
    float* out;
};

// Every block strides over every tensor; the block that arrives last adds the CLIP_BLOCKS partials in index order (in
// double), so the result does not depend on the arrival order.  Arrival is counted in two levels: device-scope atomics
// on ONE address are served one after the other at the memory side (~35 ns each on the MI355X: with 1024 blocks on one
// counter this kernel took 37-44 us whatever the width of its loads), so blocks first meet in CLIP_L1 groups on separate
// cache lines and only the last of each group goes on to the final counter: 32 + 16 serialised atomics instead of 1024.
__global__ __launch_bounds__(CLIP_THREADS) void clip_kernel(ClipLaunch a)
{
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int64_t stride = (int64_t)CLIP_BLOCKS * CLIP_THREADS;
    for (int k = 0; k < a.n; k++) {
        const float* __restrict__ g = a.g[k];
        const int64_t n = a.numel[k];
        int64_t e = (int64_t)blockIdx.x * CLIP_THREADS + threadIdx.x;
        if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
            // 16-byte loads, eight of them in flight per thread (four left 47 MB at 1.7 TB/s: three dependent rounds of
            // loads per thread and nothing else to hide them behind)
            const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
            const int64_t n4 = n >> 2;
            for (; e + 7 * stride < n4; e += 8 * stride) {
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = g4[e + i * stride];
#pragma unroll
                for (int i = 0; i < 8; i += 4) {
                    s0 = fmaf(v[i].w, v[i].w, fmaf(v[i].z, v[i].z, fmaf(v[i].y, v[i].y, fmaf(v[i].x, v[i].x, s0))));
                    s1 = fmaf(v[i + 1].w, v[i + 1].w, fmaf(v[i + 1].z, v[i + 1].z, fmaf(v[i + 1].y, v[i + 1].y, fmaf(v[i + 1].x, v[i + 1].x, s1))));
                    s2 = fmaf(v[i + 2].w, v[i + 2].w, fmaf(v[i + 2].z, v[i + 2].z, fmaf(v[i + 2].y, v[i + 2].y, fmaf(v[i + 2].x, v[i + 2].x, s2))));
                    s3 = fmaf(v[i + 3].w, v[i + 3].w, fmaf(v[i + 3].z, v[i + 3].z, fmaf(v[i + 3].y, v[i + 3].y, fmaf(v[i + 3].x, v[i + 3].x, s3))));
                }
            }
            {   // the remaining (up to seven) strides: all loads first here too
                float4 v[7];
#pragma unroll
                for (int i = 0; i < 7; i++) v[i] = e + i * stride < n4 ? g4[e + i * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    float& acc = (i & 3) == 0 ? s0 : ((i & 3) == 1 ? s1 : ((i & 3) == 2 ? s2 : s3));
                    acc = fmaf(v[i].w, v[i].w, fmaf(v[i].z, v[i].z, fmaf(v[i].y, v[i].y, fmaf(v[i].x, v[i].x, acc))));
                }
            }
            if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // (the last n mod 4 elements)
                const float v = g[4 * n4 + threadIdx.x];
                s1 = fmaf(v, v, s1);
            }
            continue;
        }
        for (; e + 3 * stride < n; e += 4 * stride) {
            const float v0 = g[e], v1 = g[e + stride], v2 = g[e + 2 * stride], v3 = g[e + 3 * stride];
            s0 = fmaf(v0, v0, s0);
            s1 = fmaf(v1, v1, s1);
            s2 = fmaf(v2, v2, s2);
            s3 = fmaf(v3, v3, s3);
        }
        for (; e < n; e += stride) s0 = fmaf(g[e], g[e], s0);
    }
    float s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    __shared__ float s_part[CLIP_THREADS / 64];
    __shared__ bool s_last;
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    unsigned* counter = reinterpret_cast<unsigned*>(a.workspace);
    float* partial = a.workspace + CLIP_PARTIALS;
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < CLIP_THREADS / 64; w++) t += s_part[w];
        partial[blockIdx.x] = t;
        __threadfence();
        bool last = atomicAdd(counter + 32 + 16 * (blockIdx.x % CLIP_L1), 1u) == (unsigned)(CLIP_BLOCKS / CLIP_L1 - 1);
        if (last) {
            __threadfence();
            last = atomicAdd(counter, 1u) == (unsigned)(CLIP_L1 - 1);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double t = 0.0;
    for (int i = threadIdx.x; i < CLIP_BLOCKS; i += CLIP_THREADS)
        t += (double)__hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // fixed-shape tree in double (xor butterflies inside a wave, then the waves in index order): the result depends on
    // neither the arrival order nor on who adds.  (Thread 0 adding the CLIP_THREADS values one after the other out of LDS
    // was 14 of this kernel's 19 us of fixed cost.)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d, 64);
    __shared__ double s_t[CLIP_THREADS / 64];
    if ((threadIdx.x & 63) == 0) s_t[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x < 32 + CLIP_L1 * 16 && (threadIdx.x == 0 || (threadIdx.x >= 32 && (threadIdx.x & 15) == 0)))
        counter[threadIdx.x] = 0u;  // (the counters are ready for the next launch)
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < CLIP_THREADS / 64; i++) tot += s_t[i];
        const float norm = (float)sqrt(tot);
        a.out[0] = norm;
        const float coef = fminf(a.max_norm / (norm + 1e-6f), 1.0f);
        a.out[1] = coef;
        a.out[2] = 1.0f / coef;   // (ABI 20: what a fused AdamW that DIVIDES by its grad_scale takes, torch/optim/adamw.py)
    }
}

// ------------------------------------------------------------------------------------------------ densify
__device__ __forceinline__ float sigmoid_like_torch(float x) { return 1.0f / (1.0f + expf(-x)); }

// counts (3, N): [0] the original survives, [1] its clone exists and survives, [2] its two split copies exist and survive
__global__ __launch_bounds__(256) void densify_plan_kernel(int N, const float* __restrict__ grad_accum,
                                                          const float* __restrict__ denom,
                                                          const float* __restrict__ scaling,
                                                          const float* __restrict__ opacity, float grad_threshold,
                                                          float dense_extent, float min_opacity, float big_world,
                                                          int32_t* __restrict__ counts)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float g = grad_accum[i] / denom[i];
    if (g != g) g = 0.f;  // grads[grads.isnan()] = 0 (:435-436)
    const float s0 = scaling[2 * i], s1 = scaling[2 * i + 1];
    const float smax = fmaxf(expf(s0), expf(s1));
    const bool clone = fabsf(g) >= grad_threshold && smax <= dense_extent;
    const bool split = g >= grad_threshold && smax > dense_extent;
    const bool faint = sigmoid_like_torch(opacity[i]) < min_opacity;
    const bool prune_same = faint || (big_world >= 0.f && smax > big_world);
    // a split copy's extent: exp(log(exp(s) / 1.6)), the division done as torch's scalar division (times reciprocal)
    const float r = 1.0f / 1.6f;
    const float cmax = fmaxf(expf(logf(expf(s0) * r)), expf(logf(expf(s1) * r)));
    const bool prune_copy = faint || (big_world >= 0.f && cmax > big_world);
    counts[i] = (!split && !prune_same) ? 1 : 0;
    counts[N + i] = (clone && !prune_same) ? 1 : 0;
    counts[2 * (size_t)N + i] = (split && !prune_copy) ? 1 : 0;
}

// source row and kind (0 original, 1 clone, 2 / 3 first / second split copy) of every output row
__global__ __launch_bounds__(256) void densify_index_kernel(int N, const int32_t* __restrict__ inc, int n_orig,
                                                           int n_clone, int n_split, int32_t* __restrict__ src_row,
                                                           uint8_t* __restrict__ kind)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int32_t a = inc[i], b = inc[N + i], c = inc[2 * (size_t)N + i];
    const int32_t pa = i ? inc[i - 1] : 0, pb = i ? inc[N + i - 1] : 0, pc = i ? inc[2 * (size_t)N + i - 1] : 0;
    if (a != pa) {
        src_row[a - 1] = i;
        kind[a - 1] = 0;
    }
    if (b != pb) {
        src_row[n_orig + b - 1] = i;
        kind[n_orig + b - 1] = 1;
    }
    if (c != pc) {
        const int r = n_orig + n_clone + c - 1;
        src_row[r] = i;
        kind[r] = 2;
        src_row[r + n_split] = i;
        kind[r + n_split] = 3;
    }
}

struct DensifyLaunch {
    Vidu4dDensifyAttr a[VIDU4D_DENSIFY_MAX_ATTRS];
    int first_col[VIDU4D_DENSIFY_MAX_ATTRS + 1];
    int n_attrs, total_width;
    int xyz_attr, scaling_attr, rotation_attr;
    int N, rows;
    const int32_t* src_row;
    const uint8_t* kind;
    const float* draws;  // (2, N, 3)
    int draws_are_scaled;
};

// One thread per (output row, column of the concatenated attributes): coalesced along the columns of a row.
__global__ __launch_bounds__(256) void densify_gather_kernel(DensifyLaunch L)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)L.rows * L.total_width) return;
    const int r = (int)(e / L.total_width), col = (int)(e - (int64_t)r * L.total_width);
    int k = 0;
#pragma unroll
    for (int i = 1; i < VIDU4D_DENSIFY_MAX_ATTRS; i++)
        if (i < L.n_attrs && col >= L.first_col[i]) k = i;
    const Vidu4dDensifyAttr at = L.a[k];
    const int c = col - L.first_col[k];
    const int s = L.src_row[r];
    const int kd = L.kind[r];
    const size_t si = (size_t)s * at.width + c, di = (size_t)r * at.width + c;
    float val = at.src[si];
    if (kd >= 2) {
        if (k == L.scaling_attr) {
            val = logf(expf(val) * (1.0f / 1.6f));  // scaling_inverse_activation(get_scaling / (0.8 * 2)) (:403-405)
        } else if (k == L.xyz_attr) {
            // xyz + R(rotation) (draw * (sx, sy, 0)) (:393-402); R as general_utils.build_rotation of the raw quaternion
            const float* q = L.a[L.rotation_attr].src + (size_t)s * 4;
            const float* sc = L.a[L.scaling_attr].src + (size_t)s * 2;
            const float* z = L.draws + ((size_t)(kd - 2) * L.N + s) * 3;
            const float d0 = L.draws_are_scaled ? z[0] : z[0] * expf(sc[0]);
            const float d1 = L.draws_are_scaled ? z[1] : z[1] * expf(sc[1]);
            const float d2 = L.draws_are_scaled ? z[2] : 0.f;
            const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            const float qr = q[0] / n, qx = q[1] / n, qy = q[2] / n, qz = q[3] / n;
            float R0, R1, R2;
            if (c == 0) {
                R0 = 1.f - 2.f * (qy * qy + qz * qz);
                R1 = 2.f * (qx * qy - qr * qz);
                R2 = 2.f * (qx * qz + qr * qy);
            } else if (c == 1) {
                R0 = 2.f * (qx * qy + qr * qz);
                R1 = 1.f - 2.f * (qx * qx + qz * qz);
                R2 = 2.f * (qy * qz - qr * qx);
            } else {
                R0 = 2.f * (qx * qz - qr * qy);
                R1 = 2.f * (qy * qz + qr * qx);
                R2 = 1.f - 2.f * (qx * qx + qy * qy);
            }
            val = (R0 * d0 + R1 * d1 + R2 * d2) + val;
        }
    }
    at.dst[di] = val;
    if (at.dst_m) {
        at.dst_m[di] = kd == 0 ? at.src_m[si] : 0.f;
        at.dst_v[di] = kd == 0 ? at.src_v[si] : 0.f;
    }
}

}  // namespace

extern "C" int vidu4d_adam_step(int n, const Vidu4dAdamTensor* tensors, double beta1, double beta2, double eps,
                                const float* grad_scale, int zero_grads, void* stream)
{
    return vidu4d_adam_step_guarded(n, tensors, beta1, beta2, eps, grad_scale, zero_grads, nullptr, stream);
}

template <int MAXT>
static int adam_launch(int n, const Vidu4dAdamTensor* tensors, double beta1, double beta2, double eps, double weight_decay,
                       const float* grad_scale, int zero_grads, const uint32_t* skip, void* stream)
{
    if (n < 0 || n > MAXT || (n && !tensors)) return VIDU4D_E_INVALID;
    AdamLaunchT<MAXT> a;
    a.n = 0;
    a.weight_decay = (float)weight_decay;
    a.beta2 = (float)beta2;
    a.one_minus_beta1 = (float)(1.0 - beta1);
    a.one_minus_beta2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.grad_scale = grad_scale;
    a.zero_grads = zero_grads;
    a.skip = skip;
    unsigned blocks = 0;
    for (int i = 0; i < n; i++) {
        const Vidu4dAdamTensor& t = tensors[i];
        if (t.numel < 0 || (t.numel && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq))) return VIDU4D_E_INVALID;
        if (!t.device_scalars && (!(t.bias_correction1 > 0.f) || !(t.bias_correction2_sqrt > 0.f))) return VIDU4D_E_INVALID;
        if (t.numel == 0) continue;
        a.t[a.n] = t;
        a.first_block[a.n] = blocks;
        blocks += (unsigned)((t.numel + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK);
        a.n++;
    }
    a.first_block[a.n] = blocks;
    if (!blocks) return VIDU4D_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(adam_kernel<MAXT>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_adam_step_guarded(int n, const Vidu4dAdamTensor* tensors, double beta1, double beta2, double eps,
                                        const float* grad_scale, int zero_grads, const uint32_t* skip, void* stream)
{
    return adam_launch<VIDU4D_ADAM_MAX_TENSORS>(n, tensors, beta1, beta2, eps, 0.0, grad_scale, zero_grads, skip, stream);
}

extern "C" int vidu4d_adamw_step_guarded(int n, const Vidu4dAdamTensor* tensors, double beta1, double beta2, double eps,
                                         double weight_decay, const float* grad_scale, int zero_grads, const uint32_t* skip,
                                         void* stream)
{
    return adam_launch<VIDU4D_ADAMW_MAX_TENSORS>(n, tensors, beta1, beta2, eps, weight_decay, grad_scale, zero_grads, skip, stream);
}

extern "C" int vidu4d_grad_clip_coef(int n, const float* const* grads, const int64_t* numel, float max_norm,
                                     float* workspace, float* out, void* stream)
{
    static_assert(VIDU4D_CLIP_WORKSPACE_FLOATS >= CLIP_PARTIALS + CLIP_BLOCKS, "clip workspace");
    if (n < 0 || n > VIDU4D_CLIP_MAX_TENSORS || (n && (!grads || !numel)) || !workspace || !out) return VIDU4D_E_INVALID;
    ClipLaunch a;
    a.n = 0;
    for (int i = 0; i < n; i++) {
        if (numel[i] < 0 || (numel[i] && !grads[i])) return VIDU4D_E_INVALID;
        if (numel[i] == 0) continue;
        a.g[a.n] = grads[i];
        a.numel[a.n] = numel[i];
        a.n++;
    }
    a.max_norm = max_norm;
    a.workspace = workspace;
    a.out = out;
    (void)hipGetLastError();
    hipLaunchKernelGGL(clip_kernel, dim3(CLIP_BLOCKS), dim3(CLIP_THREADS), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_densify_plan(int N, const float* grad_accum, const float* denom, const float* scaling,
                                   const float* opacity, float grad_threshold, float dense_extent, float min_opacity,
                                   float big_world, int32_t* counts, void* stream)
{
    if (N < 0) return VIDU4D_E_INVALID;
    if (N == 0) return VIDU4D_OK;
    if (!grad_accum || !denom || !scaling || !opacity || !counts) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    hipLaunchKernelGGL(densify_plan_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, grad_accum,
                       denom, scaling, opacity, grad_threshold, dense_extent, min_opacity, big_world, counts);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_densify_apply(int N, const int32_t* inclusive_counts, int n_orig, int n_clone, int n_split,
                                    int n_attrs, const Vidu4dDensifyAttr* attrs, int xyz_attr, int scaling_attr,
                                    int rotation_attr, const float* draws, int draws_are_scaled, int32_t* src_row,
                                    uint8_t* kind, void* stream)
{
    if (N < 0 || n_orig < 0 || n_clone < 0 || n_split < 0 || n_attrs <= 0 || n_attrs > VIDU4D_DENSIFY_MAX_ATTRS)
        return VIDU4D_E_INVALID;
    const int64_t rows = (int64_t)n_orig + n_clone + 2 * (int64_t)n_split;
    if (rows > INT32_MAX) return VIDU4D_E_INVALID;
    if (N == 0 || rows == 0) return VIDU4D_OK;
    if (!inclusive_counts || !attrs || !src_row || !kind || (n_split && !draws)) return VIDU4D_E_INVALID;
    auto ok_attr = [&](int k, int width) { return k >= 0 && k < n_attrs && attrs[k].width == width; };
    if (!ok_attr(xyz_attr, 3) || !ok_attr(scaling_attr, 2) || !ok_attr(rotation_attr, 4)) return VIDU4D_E_INVALID;
    DensifyLaunch L;
    int col = 0;
    for (int i = 0; i < n_attrs; i++) {
        const Vidu4dDensifyAttr& a = attrs[i];
        if (a.width <= 0 || !a.src || !a.dst) return VIDU4D_E_INVALID;
        if ((a.dst_m != nullptr) != (a.dst_v != nullptr) || (a.dst_m && (!a.src_m || !a.src_v))) return VIDU4D_E_INVALID;
        L.a[i] = a;
        L.first_col[i] = col;
        col += a.width;
    }
    L.first_col[n_attrs] = col;
    L.n_attrs = n_attrs;
    L.total_width = col;
    L.xyz_attr = xyz_attr;
    L.scaling_attr = scaling_attr;
    L.rotation_attr = rotation_attr;
    L.N = N;
    L.rows = (int)rows;
    L.src_row = src_row;
    L.kind = kind;
    L.draws = draws;
    L.draws_are_scaled = draws_are_scaled;
    (void)hipGetLastError();
    hipLaunchKernelGGL(densify_index_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N,
                       inclusive_counts, n_orig, n_clone, n_split, src_row, kind);
    const int64_t elems = rows * col;
    hipLaunchKernelGGL(densify_gather_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}
