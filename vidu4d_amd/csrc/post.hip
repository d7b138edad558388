// post.hip -- the per-frame post-processing of the rasterizer's auxiliary planes that
// gs.gaussian_renderer.render does after the blend (reference: gs/gaussian_renderer/__init__.py:118-151
// and gs/utils/point_utils.py:9-37), fused into one kernel per direction:
//   rend_normal  = allmap[2:5] rotated by the view matrix block M (n_out_j = sum_i n_i M[i][j])
//   depth_median = nan_to_num(allmap[5]),   depth_expected = nan_to_num(allmap[0] / allmap[1])
//   surf_depth   = (1 - ratio) depth_expected + ratio depth_median
//   surf_normal  = normalize( (P[i+1,j] - P[i-1,j]) x (P[i,j+1] - P[i,j-1]) ) * alpha   (0 on the border),
//                  P = surf_depth * ray_d + ray_o, alpha = allmap[1] taken as a constant
// Upstream this is ~25 elementwise launches per frame and ~45 in the backward; the depth-to-normal
// stencil is the only part with neighbours, and since surf_depth of a pixel needs three planes of
// that pixel only, the forward recomputes the four neighbours instead of making a second pass.
// One thread per pixel, planes are H*W apart (coalesced).  The backward is a gather: the point of
// pixel p enters the normals of its four neighbours, whose cross-product / normalisation vjp is
// recomputed from the saved surf_depth plane.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"
#include "post_math.h"

namespace {

using namespace post;

__global__ __launch_bounds__(256) void post_fwd_kernel(int W, int H, const float* __restrict__ allmap,
                                                      const float* __restrict__ rays_d,
                                                      const float* __restrict__ rays_o, const float* __restrict__ M,
                                                      float ratio, float* __restrict__ rend_normal,
                                                      float* __restrict__ depth_median,
                                                      float* __restrict__ depth_expected,
                                                      float* __restrict__ surf_depth, float* __restrict__ surf_normal)
{
    const size_t HW = (size_t)W * H;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int i = (int)(p / W), j = (int)(p % W);
    const float alpha = allmap[HW + p];
    const float n0 = allmap[2 * HW + p], n1 = allmap[3 * HW + p], n2 = allmap[4 * HW + p];
    for (int c = 0; c < 3; c++) rend_normal[c * HW + p] = n0 * M[c] + n1 * M[3 + c] + n2 * M[6 + c];
    const float expd = finite_or_zero(allmap[p] / alpha);
    const float med = finite_or_zero(allmap[5 * HW + p]);
    depth_expected[p] = expd;
    depth_median[p] = med;
    surf_depth[p] = expd * (1.0f - ratio) + ratio * med;
    const V3 n = surf_normal_at(W, H, i, j, alpha, rays_d, rays_o,
                                [&](size_t q) { return surf_depth_at(allmap, HW, q, ratio); });
    surf_normal[p] = n.x;
    surf_normal[HW + p] = n.y;
    surf_normal[2 * HW + p] = n.z;
}

__global__ __launch_bounds__(256) void post_bwd_kernel(int W, int H, const float* __restrict__ allmap,
                                                      const float* __restrict__ sd, const float* __restrict__ rays_d,
                                                      const float* __restrict__ rays_o, const float* __restrict__ M,
                                                      float ratio, const float* __restrict__ g_rn,
                                                      const float* __restrict__ g_med, const float* __restrict__ g_exp,
                                                      const float* __restrict__ g_sd, const float* __restrict__ g_sn,
                                                      float* __restrict__ g_allmap)
{
    const size_t HW = (size_t)W * H;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int i = (int)(p / W), j = (int)(p % W);
    // normals
    const float r0 = g_rn ? g_rn[p] : 0.f, r1 = g_rn ? g_rn[HW + p] : 0.f, r2 = g_rn ? g_rn[2 * HW + p] : 0.f;
    for (int c = 0; c < 3; c++) g_allmap[(2 + c) * HW + p] = M[3 * c] * r0 + M[3 * c + 1] * r1 + M[3 * c + 2] * r2;
    // depth: direct terms + the point's part in the four neighbouring normals
    float gd = g_sd ? g_sd[p] : 0.f;
    if (g_sn)
        gd += depth_grad_through_normals(W, H, i, j, sd, rays_d, rays_o, [&](int qi, int qj, float& alpha_q) {
            const size_t q = (size_t)qi * W + qj;
            alpha_q = allmap[HW + q];
            return V3{g_sn[q], g_sn[HW + q], g_sn[2 * HW + q]};
        });
    const float ge = (g_exp ? g_exp[p] : 0.f) + gd * (1.0f - ratio);
    const float gm = (g_med ? g_med[p] : 0.f) + gd * ratio;
    const float a0 = allmap[p], alpha = allmap[HW + p];
    const float r = a0 / alpha;
    const float gr = is_finite(r) ? ge : 0.f;  // nan_to_num passes the gradient where its input is finite
    g_allmap[p] = gr / alpha;
    g_allmap[HW + p] = -gr * a0 / (alpha * alpha);
    g_allmap[5 * HW + p] = is_finite(allmap[5 * HW + p]) ? gm : 0.f;
    g_allmap[6 * HW + p] = 0.f;
    g_allmap[7 * HW + p] = 0.f;
}

}  // namespace

extern "C" int vidu4d_post_forward(int W, int H, const float* allmap, const float* rays_d, const float* rays_o,
                                   const float* view3x3, float depth_ratio, float* rend_normal, float* depth_median,
                                   float* depth_expected, float* surf_depth, float* surf_normal, void* stream)
{
    if (W <= 0 || H <= 0) return VIDU4D_E_INVALID;
    if (!allmap || !rays_d || !rays_o || !view3x3 || !rend_normal || !depth_median || !depth_expected || !surf_depth ||
        !surf_normal)
        return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    const size_t HW = (size_t)W * H;
    hipLaunchKernelGGL(post_fwd_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, H,
                       allmap, rays_d, rays_o, view3x3, depth_ratio, rend_normal, depth_median, depth_expected,
                       surf_depth, surf_normal);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

extern "C" int vidu4d_post_backward(int W, int H, const float* allmap, const float* surf_depth, const float* rays_d,
                                    const float* rays_o, const float* view3x3, float depth_ratio,
                                    const float* g_rend_normal, const float* g_depth_median,
                                    const float* g_depth_expected, const float* g_surf_depth,
                                    const float* g_surf_normal, float* g_allmap, void* stream)
{
    if (W <= 0 || H <= 0) return VIDU4D_E_INVALID;
    if (!allmap || !surf_depth || !rays_d || !rays_o || !view3x3 || !g_allmap) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    const size_t HW = (size_t)W * H;
    hipLaunchKernelGGL(post_bwd_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, H,
                       allmap, surf_depth, rays_d, rays_o, view3x3, depth_ratio, g_rend_normal, g_depth_median,
                       g_depth_expected, g_surf_depth, g_surf_normal, g_allmap);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}
