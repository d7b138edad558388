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

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float finite_or_zero(float v) { return (v == v && fabsf(v) <= 3.402823466e38f) ? v : 0.f; }
__device__ __forceinline__ bool is_finite(float v) { return v == v && fabsf(v) <= 3.402823466e38f; }

__device__ __forceinline__ float surf_depth_at(const float* __restrict__ allmap, size_t HW, size_t p, float ratio)
{
    const float expd = finite_or_zero(allmap[p] / allmap[HW + p]);
    const float med = finite_or_zero(allmap[5 * HW + p]);
    return expd * (1.0f - ratio) + ratio * med;
}

__device__ __forceinline__ V3 point_at(const float* __restrict__ rays_d, const float* __restrict__ rays_o, size_t p,
                                       float depth)
{
    return {depth * rays_d[3 * p] + rays_o[0], depth * rays_d[3 * p + 1] + rays_o[1],
            depth * rays_d[3 * p + 2] + rays_o[2]};
}

constexpr float NORM_EPS = 1e-12f;  // torch.nn.functional.normalize

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
    V3 n = {0.f, 0.f, 0.f};
    if (i > 0 && i < H - 1 && j > 0 && j < W - 1) {
        const size_t up = p - W, dn = p + W, lf = p - 1, rt = p + 1;
        const V3 dx = point_at(rays_d, rays_o, dn, surf_depth_at(allmap, HW, dn, ratio)) -
                      point_at(rays_d, rays_o, up, surf_depth_at(allmap, HW, up, ratio));
        const V3 dy = point_at(rays_d, rays_o, rt, surf_depth_at(allmap, HW, rt, ratio)) -
                      point_at(rays_d, rays_o, lf, surf_depth_at(allmap, HW, lf, ratio));
        const V3 c = cross(dx, dy);
        const float inv = alpha / fmaxf(sqrtf(dot(c, c)), NORM_EPS);
        n = {c.x * inv, c.y * inv, c.z * inv};
    }
    surf_normal[p] = n.x;
    surf_normal[HW + p] = n.y;
    surf_normal[2 * HW + p] = n.z;
}

// vjp of the normal at interior pixel q w.r.t. its two difference vectors
__device__ __forceinline__ void normal_vjp(int W, int H, const float* __restrict__ allmap,
                                           const float* __restrict__ sd, const float* __restrict__ rays_d,
                                           const float* __restrict__ rays_o, const float* __restrict__ g_sn, int qi,
                                           int qj, V3& g_dx, V3& g_dy)
{
    g_dx = g_dy = {0.f, 0.f, 0.f};
    if (qi <= 0 || qi >= H - 1 || qj <= 0 || qj >= W - 1) return;
    const size_t HW = (size_t)W * H, q = (size_t)qi * W + qj;
    const V3 dx = point_at(rays_d, rays_o, q + W, sd[q + W]) - point_at(rays_d, rays_o, q - W, sd[q - W]);
    const V3 dy = point_at(rays_d, rays_o, q + 1, sd[q + 1]) - point_at(rays_d, rays_o, q - 1, sd[q - 1]);
    const V3 c = cross(dx, dy);
    const float len = sqrtf(dot(c, c));
    const float alpha = allmap[HW + q];
    const V3 g = {g_sn[q] * alpha, g_sn[HW + q] * alpha, g_sn[2 * HW + q] * alpha};  // d/d(unit normal)
    V3 g_c;
    if (len > NORM_EPS) {  // n = c / len
        const float inv = 1.0f / len;
        const V3 nh = {c.x * inv, c.y * inv, c.z * inv};
        const float proj = dot(nh, g);
        g_c = {(g.x - nh.x * proj) * inv, (g.y - nh.y * proj) * inv, (g.z - nh.z * proj) * inv};
    } else {                // n = c / eps
        g_c = {g.x / NORM_EPS, g.y / NORM_EPS, g.z / NORM_EPS};
    }
    g_dx = cross(dy, g_c);  // c = dx x dy
    g_dy = cross(g_c, dx);
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
    if (g_sn) {
        V3 a, b, gp = {0.f, 0.f, 0.f};
        normal_vjp(W, H, allmap, sd, rays_d, rays_o, g_sn, i - 1, j, a, b);  // P(p) is the "+" end of its dx
        gp = {gp.x + a.x, gp.y + a.y, gp.z + a.z};
        normal_vjp(W, H, allmap, sd, rays_d, rays_o, g_sn, i + 1, j, a, b);  // "-" end
        gp = {gp.x - a.x, gp.y - a.y, gp.z - a.z};
        normal_vjp(W, H, allmap, sd, rays_d, rays_o, g_sn, i, j - 1, a, b);  // "+" end of its dy
        gp = {gp.x + b.x, gp.y + b.y, gp.z + b.z};
        normal_vjp(W, H, allmap, sd, rays_d, rays_o, g_sn, i, j + 1, a, b);  // "-" end
        gp = {gp.x - b.x, gp.y - b.y, gp.z - b.z};
        gd += gp.x * rays_d[3 * p] + gp.y * rays_d[3 * p + 1] + gp.z * rays_d[3 * p + 2];
    }
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
