// loss.hip -- from the rasterizer's planes to the Stage-3 loss terms and back, without the ~80 elementwise / reduction
// launches (and the permute + stack copies feeding them) that the torch expression of the same arithmetic costs per step.
//
// What is computed (reference: lab4d/engine/model.py, field type "fg" under --rgb_loss_only):
//   r        = color + (1 - alpha) * learnable_bkgd                   deformable_gaussian.py:1216-1218
//   l1       = mean over ALL entries of |r - rgb| where vis2d > 0 (else 0), times (1 - lambda_dssim)     :653-667
//   rgb map  = l1 * (mask * vis2d)                                     mask_losses :895-978 (one scalar spread out)
//   w        = get_mask_balance_wt(mask, vis2d, is_detected)           :586-611
//   mask map = (alpha - mask)^2 * w * vis2d * is_detected              :640-651, :945-952
//   term     = mean over the entries > 0 (over everything when there is none), times its weight   apply_loss_weights :980-1012
//   dist     = weight * mean(distortion plane)                         compute_reg_loss :835-842
//   normal   = weight * mean over (H, W, 3) of (1 - SUM OVER THE FRAMES of rend_normal * surf_normal)   :817-834 -- the
//              `.sum(dim=0)` upstream wrote for the (3,H,W) layout of 2DGS runs over the frame axis of the (M,H,W,3)
//              maps here; reproduced.  rend_normal = allmap[2:5] rotated by the view block, surf_normal = the
//              depth-to-normal stencil of gs.gaussian_renderer.render (post_math.h), evaluated IN these kernels from
//              the depth / alpha planes (or read from planes the caller supplies): the seven maps render() derives
//              per frame and their backward never reach HBM.  Switched on by normal_wt != 0 (step > 8000 upstream).
// The torch statement of this (vidu4d_amd/lab4d/stage3.py::compute_losses) is pinned against the imported reference by
// tests/golden/refpy_losses.npz; tests compare these kernels with it, values and gradients, on those cases.
//
// Structure: the sums that other sums depend on force four small passes forward -- per-block partials of the first-level
// sums, their fixed-order reduction + derived scalars, per-block partials of the mask-map sums, final reduction -- and one
// pass backward that writes the gradient planes the rasterizer's backward consumes (colour 3 planes, all 8 auxiliary
// planes) plus per-block partials of the background gradient.  Fixed block count and fixed-order second stages: the
// result does not depend on scheduling.  inf / NaN propagate as in the torch expression (0/0 weights, empty masks).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"
#include "post_math.h"

namespace {

using post::V3;

constexpr int BLOCKS = VIDU4D_LOSS_BLOCKS;
constexpr int THREADS = 256;
// first-level sums
enum { S_L1, S_VIS, S_POSW, S_NEGW, S_MF, S_NMF, S_RGB_N, S_RGB_SUM, S_MV, S_DIST, S_NRM, N1 };
// second-level sums
enum { S_MASK_POS, S_MASK_N, S_MASK_ALL, N2 };
// derived scalars stored after the raw sums in `sums`
enum { D_L1 = 16, D_POSWT, D_NEGWT, D_BOTH, D_RGB_COEF, D_MASK_INV, D_MASK_USE_POS, SUMS_FLOATS = 32 };
static_assert(N1 <= 12 && N1 + N2 <= 16 && SUMS_FLOATS == VIDU4D_LOSS_SUMS_FLOATS, "layout of the sums buffer");

template <int K>
__device__ __forceinline__ void block_partials(float (&v)[K], float* out /* [K] of this block */)
{
    __shared__ float s[THREADS / 64][K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        float x = v[k];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float x = 0.f;
        for (int w = 0; w < THREADS / 64; w++) x += s[w][threadIdx.x];
        out[threadIdx.x] = x;
    }
}

struct Pixel {
    float r[3], t[3], a, mf, v, det, dist;
};

__device__ __forceinline__ Pixel load_pixel(const Vidu4dStage3LossArgs& a, size_t HW, size_t e)
{
    const int m = (int)(e / HW);
    const size_t p = e - (size_t)m * HW;
    Pixel q;
    const float* col = a.color[m];
    const float* aux = a.allmap[m];
    const size_t PS = a.plane_stride ? (size_t)a.plane_stride : HW;
    q.a = aux[PS + p];
    q.dist = aux[6 * PS + p];
    for (int c = 0; c < 3; c++) {
        q.r[c] = col[c * PS + p] + (1.0f - q.a) * (a.bkgd ? a.bkgd[c] : 0.f);
        q.t[c] = a.rgb[3 * e + c];
    }
    q.mf = a.mask[e];
    q.v = a.vis2d[e];
    q.det = a.det ? a.det[m] : 1.0f;
    return q;
}

// rend_normal of pixel p of frame m: allmap[2:5] rotated by the frame's view block (n_out_j = sum_i n_i M[i][j]).
__device__ __forceinline__ V3 rend_normal_at(const Vidu4dStage3LossArgs& a, int m, size_t PS, size_t p)
{
    const float* aux = a.allmap[m];
    const float* M = a.view3x3[m];
    const float n0 = aux[2 * PS + p], n1 = aux[3 * PS + p], n2 = aux[4 * PS + p];
    return {n0 * M[0] + n1 * M[3] + n2 * M[6], n0 * M[1] + n1 * M[4] + n2 * M[7], n0 * M[2] + n1 * M[5] + n2 * M[8]};
}

__global__ __launch_bounds__(THREADS) void loss_stats1_kernel(Vidu4dStage3LossArgs a, float* partial)
{
    const size_t HW = (size_t)a.H * a.W, total = HW * a.M;
    float s[N1];
#pragma unroll
    for (int k = 0; k < N1; k++) s[k] = 0.f;
    for (size_t e = (size_t)blockIdx.x * THREADS + threadIdx.x; e < total; e += (size_t)BLOCKS * THREADS) {
        const Pixel q = load_pixel(a, HW, e);
        if (q.v > 0.f)
            for (int c = 0; c < 3; c++) s[S_L1] += fabsf(q.r[c] - q.t[c]);
        const float vis = q.v * q.det;
        const float seen = vis > 0.f ? 1.f : 0.f;
        s[S_VIS] += vis;
        s[S_POSW] += q.mf * seen;
        s[S_NEGW] += (1.f - q.mf) * seen;
        s[S_MF] += q.mf;
        s[S_NMF] += 1.f - q.mf;
        const float mv = q.mf * q.v;
        s[S_RGB_N] += mv > 0.f ? 1.f : 0.f;
        s[S_RGB_SUM] += mv > 0.f ? mv : 0.f;
        s[S_MV] += mv;
        s[S_DIST] += q.dist;
        if (a.normal_wt != 0.f) {
            const int m = (int)(e / HW);
            const size_t p = e - (size_t)m * HW;
            const size_t PS = a.plane_stride ? (size_t)a.plane_stride : HW;
            const V3 rn = rend_normal_at(a, m, PS, p);
            V3 sn;
            if (a.surf_normal[m]) {
                const float* t = a.surf_normal[m];
                sn = {t[p], t[HW + p], t[2 * HW + p]};
            } else {
                const float* aux = a.allmap[m];
                const float ratio = a.depth_ratio;
                a.surf_depth[e] = post::surf_depth_at(aux, PS, p, ratio);   // (the backward's stencil reads the plane)
                sn = post::surf_normal_at(a.W, a.H, (int)(p / a.W), (int)(p % a.W), q.a, a.rays_d[m], a.rays_o[m],
                                          [&](size_t n) { return post::surf_depth_at(aux, PS, n, ratio); });
            }
            s[S_NRM] += rn.x * sn.x + rn.y * sn.y + rn.z * sn.z;
        }
    }
    block_partials<N1>(s, partial + blockIdx.x * 16);
}

__device__ __forceinline__ float ordered_sum(const float* partial, int k)
{
    // one wave, fixed order: lane i adds blocks i, i + 64, ...; then a fixed butterfly
    float x = 0.f;
    for (int b = (threadIdx.x & 63); b < BLOCKS; b += 64) x += partial[b * 16 + k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// The K second-stage sums of a reduce kernel, one WAVE per sum (the kernel runs K waves): the same fixed order per sum as a
// single wave doing them one after the other, a quarter of its latency (16 -> 5 us for the 11 first-level sums).
template <int K>
__device__ __forceinline__ void ordered_sums(const float* partial, float (&s)[K])
{
    __shared__ float sh[K];
    const int wave = threadIdx.x >> 6;
    if (wave < K) {
        const float x = ordered_sum(partial, wave);
        if ((threadIdx.x & 63) == 0) sh[wave] = x;
    }
    __syncthreads();
    for (int k = 0; k < K; k++) s[k] = sh[k];
}

__global__ __launch_bounds__(64 * N1) void loss_reduce1_kernel(Vidu4dStage3LossArgs a, const float* partial)
{
    float s[N1];
    ordered_sums<N1>(partial, s);
    if (threadIdx.x != 0) return;
    const float numel = (float)((size_t)a.H * a.W * a.M);
    float* S = a.sums;
    for (int k = 0; k < N1; k++) S[k] = s[k];
    const float l1 = s[S_L1] / (3.0f * numel) * (1.0f - a.lambda_dssim);
    S[D_L1] = l1;
    S[D_POSWT] = s[S_VIS] / s[S_POSW];
    S[D_NEGWT] = s[S_VIS] / s[S_NEGW];
    S[D_BOTH] = (s[S_MF] > 0.f && s[S_NMF] > 0.f) ? 1.f : 0.f;
    // rgb term = mean_of_positive(l1 * mv) * rgb_wt; positives exist iff l1 > 0 and some mv > 0
    const bool pos = l1 > 0.f && s[S_RGB_N] > 0.f;
    const float coef = pos ? s[S_RGB_SUM] / s[S_RGB_N] : s[S_MV] / numel;  // d(term / rgb_wt) / d l1
    S[D_RGB_COEF] = coef;
    a.losses[0] = l1 * coef * a.rgb_wt;
    a.losses[2] = a.dist_wt != 0.f ? a.dist_wt * s[S_DIST] / numel : 0.f;
    // mean over the (1, H, W, 3) map left by the sum over the frame axis: 3 H W entries, whatever M is
    a.losses[3] = a.normal_wt != 0.f ? a.normal_wt * (1.0f - s[S_NRM] / (3.0f * (float)((size_t)a.H * a.W))) : 0.f;
}

__device__ __forceinline__ float mask_map(const Pixel& q, const float* S, float& dmap_da)
{
    const float w = S[D_BOTH] != 0.f ? 0.5f * S[D_POSWT] * q.mf + 0.5f * S[D_NEGWT] * (1.f - q.mf) : 1.f;
    const float k = w * q.v * q.det;
    const float d = q.a - q.mf;
    dmap_da = 2.f * d * k;
    return d * d * k;
}

__global__ __launch_bounds__(THREADS) void loss_stats2_kernel(Vidu4dStage3LossArgs a, float* partial)
{
    const size_t HW = (size_t)a.H * a.W, total = HW * a.M;
    float s[N2] = {0.f, 0.f, 0.f};
    for (size_t e = (size_t)blockIdx.x * THREADS + threadIdx.x; e < total; e += (size_t)BLOCKS * THREADS) {
        const Pixel q = load_pixel(a, HW, e);
        float unused;
        const float d = mask_map(q, a.sums, unused);
        s[S_MASK_POS] += d > 0.f ? d : 0.f;
        s[S_MASK_N] += d > 0.f ? 1.f : 0.f;
        s[S_MASK_ALL] += d;
    }
    block_partials<N2>(s, partial + blockIdx.x * 16);
}

__global__ __launch_bounds__(64 * N2) void loss_reduce2_kernel(Vidu4dStage3LossArgs a, const float* partial)
{
    float s[N2];
    ordered_sums<N2>(partial, s);
    if (threadIdx.x != 0) return;
    const float numel = (float)((size_t)a.H * a.W * a.M);
    float* S = a.sums;
    for (int k = 0; k < N2; k++) S[N1 + k] = s[k];
    const bool pos = s[S_MASK_N] > 0.f;
    S[D_MASK_USE_POS] = pos ? 1.f : 0.f;
    S[D_MASK_INV] = pos ? 1.0f / s[S_MASK_N] : 1.0f / numel;
    const float mask_term = (pos ? s[S_MASK_POS] / s[S_MASK_N] : s[S_MASK_ALL] / numel) * a.mask_wt;
    a.losses[1] = mask_term;
    // the sum the trainer back-propagates, in the order of its dict (rgb, mask, normal_loss, dist_loss; [0], [2], [3]: reduce1)
    a.losses[4] = ((a.losses[0] + mask_term) + a.losses[3]) + a.losses[2];
}

// g (5): upstream gradients of the four terms and, g[4], of their sum losses[4] (added to each).  Writes g_color[m] (3,H,W),
// g_allmap[m] (8,H,W) completely (and g_surf_normal[m] when the caller supplied the surf_normal planes) and the per-block
// partials of d / d learnable_bkgd.
__global__ __launch_bounds__(THREADS) void loss_backward_kernel(Vidu4dStage3LossArgs a, const float* g, Vidu4dStage3LossGrads o,
                                                               float* partial)
{
    const size_t HW = (size_t)a.H * a.W, total = HW * a.M;
    const size_t PS = a.plane_stride ? (size_t)a.plane_stride : HW;
    const float numel = (float)total;
    const float* S = a.sums;
    const float g_rgb = (g[0] + g[4]) * a.rgb_wt * S[D_RGB_COEF] * (1.0f - a.lambda_dssim) / (3.0f * numel);  // per |r - t| entry
    const float g_mask = (g[1] + g[4]) * a.mask_wt * S[D_MASK_INV];
    const bool only_pos = S[D_MASK_USE_POS] != 0.f;
    const float g_dist = a.dist_wt != 0.f ? (g[2] + g[4]) * a.dist_wt / numel : 0.f;
    // d normal_loss / d (rend_normal . surf_normal summed over everything) = -weight / (3 H W)
    const float g_nrm = a.normal_wt != 0.f ? -(g[3] + g[4]) * a.normal_wt / (3.0f * (float)HW) : 0.f;
    float bg[3] = {0.f, 0.f, 0.f};
    for (size_t e = (size_t)blockIdx.x * THREADS + threadIdx.x; e < total; e += (size_t)BLOCKS * THREADS) {
        const int m = (int)(e / HW);
        const size_t p = e - (size_t)m * HW;
        const Pixel q = load_pixel(a, HW, e);
        float da;
        const float d = mask_map(q, S, da);
        float ga = (only_pos && !(d > 0.f)) ? 0.f : g_mask * da;
        float* gc = o.g_color[m];
        for (int c = 0; c < 3; c++) {
            const float diff = q.r[c] - q.t[c];
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);  // (torch: d|x|/dx = sign(x), 0 at 0; NaN stays NaN)
            const float gr = (q.v > 0.f) ? g_rgb * (diff != diff ? diff : sgn) : 0.f;
            gc[c * PS + p] = gr;
            if (a.bkgd) {
                ga -= gr * a.bkgd[c];
                bg[c] += gr * (1.0f - q.a);
            }
        }
        float* gm = o.g_allmap[m];
        float g_depth = 0.f, g_median = 0.f, g_n[3] = {0.f, 0.f, 0.f};
        if (a.normal_wt != 0.f) {
            const int W = a.W, H = a.H, i = (int)(p / W), j = (int)(p % W);
            const float* aux = a.allmap[m];
            const float* M = a.view3x3[m];
            V3 sn;   // d / d rend_normal(p) = g_nrm * surf_normal(p)
            if (a.surf_normal[m]) {
                const float* t = a.surf_normal[m];
                sn = {t[p], t[HW + p], t[2 * HW + p]};
                if (o.g_surf_normal[m]) {   // d / d surf_normal(p) = g_nrm * rend_normal(p)
                    const V3 rn = rend_normal_at(a, m, PS, p);
                    float* gs = o.g_surf_normal[m];
                    gs[p] = g_nrm * rn.x, gs[HW + p] = g_nrm * rn.y, gs[2 * HW + p] = g_nrm * rn.z;
                }
            } else {
                const float* sd = a.surf_depth + (size_t)m * HW;
                sn = post::surf_normal_at(W, H, i, j, q.a, a.rays_d[m], a.rays_o[m], [&](size_t n) { return sd[n]; });
                // the depth of this pixel enters the surf_normal of its four neighbours (whose upstream gradient is
                // g_nrm * their rend_normal), then the expected / median depth, then planes 0, 1 and 5
                const float gd = post::depth_grad_through_normals(
                    W, H, i, j, sd, a.rays_d[m], a.rays_o[m], [&](int qi, int qj, float& alpha_q) {
                        const size_t n = (size_t)qi * W + qj;
                        alpha_q = aux[PS + n];
                        const V3 rn = rend_normal_at(a, m, PS, n);
                        return V3{g_nrm * rn.x, g_nrm * rn.y, g_nrm * rn.z};
                    });
                const float ge = gd * (1.0f - a.depth_ratio), gmed = gd * a.depth_ratio;
                const float a0 = aux[p];
                const float gr = post::is_finite(a0 / q.a) ? ge : 0.f;  // nan_to_num passes the gradient where its input is finite
                g_depth = gr / q.a;                                      // (0 / 0 where nothing was blended, as torch: never read)
                ga += -gr * a0 / (q.a * q.a);
                g_median = post::is_finite(aux[5 * PS + p]) ? gmed : 0.f;
            }
            const float r0 = g_nrm * sn.x, r1 = g_nrm * sn.y, r2 = g_nrm * sn.z;
            for (int c = 0; c < 3; c++) g_n[c] = M[3 * c] * r0 + M[3 * c + 1] * r1 + M[3 * c + 2] * r2;
        }
        gm[p] = g_depth;
        gm[PS + p] = ga;
        for (int k = 0; k < 3; k++) gm[(2 + k) * PS + p] = g_n[k];
        gm[5 * PS + p] = g_median;
        gm[6 * PS + p] = g_dist;
        gm[7 * PS + p] = 0.f;
    }
    block_partials<3>(bg, partial + blockIdx.x * 16);
}

__global__ __launch_bounds__(192) void loss_reduce_bg_kernel(const float* partial, float* g_bkgd)
{
    const float x = ordered_sum(partial, threadIdx.x >> 6);   // one wave per channel
    if ((threadIdx.x & 63) == 0) g_bkgd[threadIdx.x >> 6] = x;
}

int check(const Vidu4dStage3LossArgs* a)
{
    if (!a || a->M <= 0 || a->M > VIDU4D_LOSS_MAX_FRAMES || a->H <= 0 || a->W <= 0) return VIDU4D_E_INVALID;
    if (!a->rgb || !a->mask || !a->vis2d || !a->sums || !a->losses || !a->partials) return VIDU4D_E_INVALID;
    for (int m = 0; m < a->M; m++)
        if (!a->color[m] || !a->allmap[m]) return VIDU4D_E_INVALID;
    if (a->normal_wt != 0.f)
        for (int m = 0; m < a->M; m++) {
            if (!a->view3x3[m]) return VIDU4D_E_INVALID;
            if (!a->surf_normal[m] && (!a->rays_d[m] || !a->rays_o[m] || !a->surf_depth)) return VIDU4D_E_INVALID;
        }
    return VIDU4D_OK;
}

}  // namespace

extern "C" int vidu4d_stage3_loss_forward(const Vidu4dStage3LossArgs* a, void* stream)
{
    const int rc = check(a);
    if (rc != VIDU4D_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    hipLaunchKernelGGL(loss_stats1_kernel, dim3(BLOCKS), dim3(THREADS), 0, s, *a, a->partials);
    hipLaunchKernelGGL(loss_reduce1_kernel, dim3(1), dim3(64 * N1), 0, s, *a, a->partials);
    hipLaunchKernelGGL(loss_stats2_kernel, dim3(BLOCKS), dim3(THREADS), 0, s, *a, a->partials);
    hipLaunchKernelGGL(loss_reduce2_kernel, dim3(1), dim3(64 * N2), 0, s, *a, a->partials);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_stage3_loss_backward(const Vidu4dStage3LossArgs* a, const float* g_losses,
                                           const Vidu4dStage3LossGrads* grads, void* stream)
{
    const int rc = check(a);
    if (rc != VIDU4D_OK) return rc;
    if (!g_losses || !grads || (a->bkgd && !grads->g_bkgd)) return VIDU4D_E_INVALID;
    for (int m = 0; m < a->M; m++)
        if (!grads->g_color[m] || !grads->g_allmap[m]) return VIDU4D_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();
    hipLaunchKernelGGL(loss_backward_kernel, dim3(BLOCKS), dim3(THREADS), 0, s, *a, g_losses, *grads, a->partials);
    if (a->bkgd) hipLaunchKernelGGL(loss_reduce_bg_kernel, dim3(1), dim3(192), 0, s, a->partials, grads->g_bkgd);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}
