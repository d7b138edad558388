// TEST INFRASTRUCTURE.  C entry points around the reference's CudaRasterizer::Rasterizer, compiled
// together with the reference's own sources (read in place from /root/reference) by build_ref.py.
// The three scratch buffers are hipMalloc'ed here the way rasterize_points.cu:31-37 resizes torch
// tensors; typed views of them are re-derived with the reference's own fromChunk functions.
#include <functional>
#include <stdint.h>
#include <string.h>

#include "rasterizer.h"
#include "rasterizer_impl.h"

namespace {
struct Buf {
    char* p = nullptr;
    size_t n = 0;
    char* get(size_t want)
    {
        if (want > n) {
            if (p) (void)hipFree(p);
            (void)hipMalloc((void**)&p, want);
            n = want;
        }
        return p;
    }
};
Buf g_geom, g_bin, g_img;
int g_R = 0, g_P = 0, g_W = 0, g_H = 0;
}  // namespace

extern "C" int ref_forward(int P, int D, int M, const float* bg, int W, int H, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* view, const float* proj, const float* campos,
                           float tan_fovx, float tan_fovy, float* out_color, float* out_others, int* radii)
{
    std::function<char*(size_t)> fg = [](size_t n) { return g_geom.get(n); };
    std::function<char*(size_t)> fb = [](size_t n) { return g_bin.get(n); };
    std::function<char*(size_t)> fi = [](size_t n) { return g_img.get(n); };
    g_P = P; g_W = W; g_H = H;
    g_R = CudaRasterizer::Rasterizer::forward(fg, fb, fi, P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities,
                                              scales, 1.0f, rotations, nullptr, view, proj, campos, tan_fovx, tan_fovy,
                                              false, out_color, out_others, radii, false);
    (void)hipDeviceSynchronize();
    return g_R;
}

extern "C" void ref_backward(int P, int D, int M, const float* bg, int W, int H, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* scales, const float* rotations,
                             const float* view, const float* proj, const float* campos, float tan_fovx, float tan_fovy,
                             const int* radii, const float* dL_dpix, const float* dL_depths, float* dL_dmean2D,
                             float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                             float* dL_dtransMat, float* dL_dsh, float* dL_dscale, float* dL_drot)
{
    CudaRasterizer::Rasterizer::backward(P, D, M, g_R, bg, W, H, means3D, shs, colors_precomp, scales, 1.0f, rotations,
                                         nullptr, view, proj, campos, tan_fovx, tan_fovy, radii, g_geom.p, g_bin.p,
                                         g_img.p, dL_dpix, dL_depths, dL_dmean2D, dL_dnormal, dL_dopacity, dL_dcolor,
                                         dL_dmean3D, dL_dtransMat, dL_dsh, dL_dscale, dL_drot, false);
    (void)hipDeviceSynchronize();
}

// what: 0 point_list (u32[R]), 1 sorted keys (u64[R]), 2 ranges (u32[tiles*2]), 3 n_contrib (u32[2*HW]),
//       4 final_T (f32[3*HW]), 5 transMat (f32[P*9]), 6 means2D (f32[P*2]), 7 depths (f32[P]), 8 rgb (f32[P*3]),
//       9 normal_opacity (f32[P*4]), 10 tiles_touched (u32[P])
extern "C" long ref_state(int what, void* host_dst, size_t dst_bytes)
{
    char* gp = g_geom.p;
    char* bp = g_bin.p;
    char* ip = g_img.p;
    auto geom = CudaRasterizer::GeometryState::fromChunk(gp, g_P);
    auto bin = CudaRasterizer::BinningState::fromChunk(bp, g_R);
    auto img = CudaRasterizer::ImageState::fromChunk(ip, (size_t)g_W * g_H);
    const size_t HW = (size_t)g_W * g_H;
    const size_t tiles = (size_t)((g_W + 15) / 16) * ((g_H + 15) / 16);
    const void* src = nullptr;
    size_t bytes = 0;
    switch (what) {
        case 0: src = bin.point_list; bytes = (size_t)g_R * 4; break;
        case 1: src = bin.point_list_keys; bytes = (size_t)g_R * 8; break;
        case 2: src = img.ranges; bytes = tiles * 8; break;
        case 3: src = img.n_contrib; bytes = 2 * HW * 4; break;
        case 4: src = img.accum_alpha; bytes = 3 * HW * 4; break;
        case 5: src = geom.transMat; bytes = (size_t)g_P * 36; break;
        case 6: src = geom.means2D; bytes = (size_t)g_P * 8; break;
        case 7: src = geom.depths; bytes = (size_t)g_P * 4; break;
        case 8: src = geom.rgb; bytes = (size_t)g_P * 12; break;
        case 9: src = geom.normal_opacity; bytes = (size_t)g_P * 16; break;
        case 10: src = geom.tiles_touched; bytes = (size_t)g_P * 4; break;
        default: return -1;
    }
    if (bytes > dst_bytes) return -2;
    if (bytes) (void)hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost);
    return (long)bytes;
}
