// post_math.h -- per-pixel pieces of the depth / normal post-processing of the rasterizer's auxiliary planes, shared by
// post.hip (one kernel per direction behind gs.gaussian_renderer.render) and loss.hip (the same arithmetic inside the
// Stage-3 loss kernels, where the maps never reach HBM).  Reference: gs/gaussian_renderer/__init__.py:118-151,
// gs/utils/point_utils.py:9-37.
//   depth_median = nan_to_num(allmap[5]),   depth_expected = nan_to_num(allmap[0] / allmap[1])
//   surf_depth   = (1 - ratio) depth_expected + ratio depth_median
//   surf_normal  = normalize( (P[i+1,j] - P[i-1,j]) x (P[i,j+1] - P[i,j-1]) ) * alpha   (0 on the border),
//                  P = surf_depth * ray_d + ray_o, alpha = allmap[1] taken as a constant
// `PS` is the distance between two planes of `allmap` in floats (H*W, or frames*H*W for the planes of a stacked call).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace post {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float finite_or_zero(float v) { return (v == v && fabsf(v) <= 3.402823466e38f) ? v : 0.f; }
__device__ __forceinline__ bool is_finite(float v) { return v == v && fabsf(v) <= 3.402823466e38f; }

__device__ __forceinline__ float surf_depth_at(const float* __restrict__ allmap, size_t PS, size_t p, float ratio)
{
    const float expd = finite_or_zero(allmap[p] / allmap[PS + p]);
    const float med = finite_or_zero(allmap[5 * PS + p]);
    return expd * (1.0f - ratio) + ratio * med;
}

__device__ __forceinline__ V3 point_at(const float* __restrict__ rays_d, const float* __restrict__ rays_o, size_t p,
                                       float depth)
{
    return {depth * rays_d[3 * p] + rays_o[0], depth * rays_d[3 * p + 1] + rays_o[1],
            depth * rays_d[3 * p + 2] + rays_o[2]};
}

constexpr float NORM_EPS = 1e-12f;  // torch.nn.functional.normalize

// surf_normal of pixel (i, j) from the surf_depth of its four neighbours, given by `depth_of(pixel index)`.
template <typename DepthOf>
__device__ __forceinline__ V3 surf_normal_at(int W, int H, int i, int j, float alpha, const float* __restrict__ rays_d,
                                             const float* __restrict__ rays_o, DepthOf depth_of)
{
    if (!(i > 0 && i < H - 1 && j > 0 && j < W - 1)) return {0.f, 0.f, 0.f};
    const size_t p = (size_t)i * W + j;
    const size_t up = p - W, dn = p + W, lf = p - 1, rt = p + 1;
    const V3 dx = point_at(rays_d, rays_o, dn, depth_of(dn)) - point_at(rays_d, rays_o, up, depth_of(up));
    const V3 dy = point_at(rays_d, rays_o, rt, depth_of(rt)) - point_at(rays_d, rays_o, lf, depth_of(lf));
    const V3 c = cross(dx, dy);
    const float inv = alpha / fmaxf(sqrtf(dot(c, c)), NORM_EPS);
    return {c.x * inv, c.y * inv, c.z * inv};
}

// vjp of the normal at interior pixel (qi, qj) w.r.t. its two difference vectors; `g` = d L / d surf_normal(q), alpha =
// allmap[1] at q, `sd` the saved surf_depth plane of the frame.
__device__ __forceinline__ void normal_vjp(int W, int H, float alpha, V3 g, const float* __restrict__ sd,
                                           const float* __restrict__ rays_d, const float* __restrict__ rays_o, int qi,
                                           int qj, V3& g_dx, V3& g_dy)
{
    const size_t q = (size_t)qi * W + qj;
    const V3 dx = point_at(rays_d, rays_o, q + W, sd[q + W]) - point_at(rays_d, rays_o, q - W, sd[q - W]);
    const V3 dy = point_at(rays_d, rays_o, q + 1, sd[q + 1]) - point_at(rays_d, rays_o, q - 1, sd[q - 1]);
    const V3 c = cross(dx, dy);
    const float len = sqrtf(dot(c, c));
    g = {g.x * alpha, g.y * alpha, g.z * alpha};  // d/d(unit normal)
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

__device__ __forceinline__ bool interior(int W, int H, int i, int j) { return i > 0 && i < H - 1 && j > 0 && j < W - 1; }

// d L / d surf_depth(i, j) through the normals of the four neighbours: the point of pixel p is the "+" end of the dx of
// the pixel above it, the "-" end of the one below, the "+" end of the dy of its left neighbour, the "-" end of its
// right one.  `g_normal_of(qi, qj, alpha_out)` returns d L / d surf_normal at an INTERIOR pixel and its alpha.
template <typename GradOf>
__device__ __forceinline__ float depth_grad_through_normals(int W, int H, int i, int j, const float* __restrict__ sd,
                                                            const float* __restrict__ rays_d,
                                                            const float* __restrict__ rays_o, GradOf g_normal_of)
{
    V3 a, b, gp = {0.f, 0.f, 0.f};
    float alpha;
    if (interior(W, H, i - 1, j)) {
        const V3 g = g_normal_of(i - 1, j, alpha);
        normal_vjp(W, H, alpha, g, sd, rays_d, rays_o, i - 1, j, a, b);
        gp = {gp.x + a.x, gp.y + a.y, gp.z + a.z};
    }
    if (interior(W, H, i + 1, j)) {
        const V3 g = g_normal_of(i + 1, j, alpha);
        normal_vjp(W, H, alpha, g, sd, rays_d, rays_o, i + 1, j, a, b);
        gp = {gp.x - a.x, gp.y - a.y, gp.z - a.z};
    }
    if (interior(W, H, i, j - 1)) {
        const V3 g = g_normal_of(i, j - 1, alpha);
        normal_vjp(W, H, alpha, g, sd, rays_d, rays_o, i, j - 1, a, b);
        gp = {gp.x + b.x, gp.y + b.y, gp.z + b.z};
    }
    if (interior(W, H, i, j + 1)) {
        const V3 g = g_normal_of(i, j + 1, alpha);
        normal_vjp(W, H, alpha, g, sd, rays_d, rays_o, i, j + 1, a, b);
        gp = {gp.x - b.x, gp.y - b.y, gp.z - b.z};
    }
    const size_t p = (size_t)i * W + j;
    return gp.x * rays_d[3 * p] + gp.y * rays_d[3 * p + 1] + gp.z * rays_d[3 * p + 2];
}

}  // namespace post
