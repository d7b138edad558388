// host_emul.cpp -- TEST INFRASTRUCTURE.  Compiles the product's arithmetic header
// (vidu4d_amd/csrc/surfel_math.h) for the host and runs it through sequential loops that stand in
// for the kernels' thread grids, so that the formulas can be checked against the oracle on a box
// without a GPU.  It is never loaded by the product package; it is not a fallback.
#include <algorithm>
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../vidu4d_amd/csrc/surfel_math.h"

using namespace surfel;

static Camera make_cam(const float* view, const float* campos, int W, int H, float tfx, float tfy, int D, int M)
{
    Camera c;
    memcpy(c.view, view, sizeof(c.view));
    memcpy(c.campos, campos, sizeof(c.campos));
    c.W = W; c.H = H;
    c.grid_x = (W + TILE - 1) / TILE; c.grid_y = (H + TILE - 1) / TILE;
    c.tan_fovx = tfx; c.tan_fovy = tfy;
    c.focal_y = H / (2.0f * tfy); c.focal_x = W / (2.0f * tfx);
    c.cx = (float)((double)(float)W / 2.0); c.cy = (float)((double)(float)H / 2.0);
    c.sh_degree = D; c.sh_coeffs = M;
    return c;
}

extern "C" void emul_preprocess(int P, int D, int M, const float* means3D, const float* scales, const float* rotations,
                                const float* opacities, const float* shs, const float* colors_precomp,
                                const float* view, const float* campos, int W, int H, float tfx, float tfy,
                                int32_t* radii, uint32_t* tiles, float* rec)
{
    const Camera cam = make_cam(view, campos, W, H, tfx, tfy, D, M);
    for (int i = 0; i < P; i++) {
        Projected o;
        radii[i] = 0; tiles[i] = 0;
        for (int k = 0; k < REC_FLOATS; k++) rec[(size_t)i * REC_FLOATS + k] = 0.f;
        if (!project_surfel(cam, means3D + 3 * i, rotations + 4 * i, scales + 2 * i, o)) continue;
        float rgb[3]; uint32_t mask = 0;
        if (colors_precomp) { for (int c = 0; c < 3; c++) rgb[c] = colors_precomp[3 * i + c]; }
        else sh_forward(D, means3D + 3 * i, cam.campos, shs + (size_t)i * M * 3, rgb, mask);
        float* r = rec + (size_t)i * REC_FLOATS;
        for (int k = 0; k < 9; k++) r[k] = o.T[k];
        r[R_CX] = o.center[0]; r[R_CY] = o.center[1]; r[R_OPAC] = opacities[i];
        for (int k = 0; k < 3; k++) { r[R_NX + k] = o.normal[k]; r[R_RGB + k] = rgb[k]; }
        r[R_DEPTH] = o.depth;
        memcpy(&r[R_CLAMP], &mask, 4);
        contribution_footprint(o.T, o.center[0], o.center[1], opacities[i], r + R_FOOT);
        radii[i] = o.radius; tiles[i] = o.tiles;
    }
}

// For every visible surfel and every 8x8 quadrant of the image: does footprint_hits say the quadrant's rectangle of pixel
// centres can be reached, and does some pixel of it pass the exact pair test (eval_pair) / the per-pixel footprint test?
// counts[0] quadrants the rectangle test keeps, [1] quadrants with a pixel that passes eval_pair, [2] quadrants with such a
// pixel that the rectangle test DROPS (must be 0: not conservative), [3] quadrants the rectangle test keeps although no
// pixel centre passes even the per-pixel footprint test (slack of the rectangle test over its own per-pixel form: the
// footprint passing between pixel centres), [4] quadrants the bounding box of the footprint's per-pixel hits reaches
// (what a box test keeps at best).
extern "C" void emul_footprint_scan(int P, int W, int H, const int32_t* radii, const float* rec, long long* counts)
{
    for (int k = 0; k < 5; k++) counts[k] = 0;
    const int qx = (W + 7) / 8, qy = (H + 7) / 8;
    std::vector<unsigned char> pass(qx * qy), foot(qx * qy);
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        const float* r = rec + (size_t)i * REC_FLOATS;
        const FootprintTest ft = footprint_test(r + R_FOOT, r[R_CX], r[R_CY]);
        std::fill(pass.begin(), pass.end(), 0);
        std::fill(foot.begin(), foot.end(), 0);
        int bx0 = qx, bx1 = -1, by0 = qy, by1 = -1;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float pixx = (float)x + 0.5f, pixy = (float)y + 0.5f;
                PairEval e;
                if (eval_pair(r + R_TU, r + R_TV, r + R_TW, r[R_CX], r[R_CY], r[R_OPAC], pixx, pixy, e)) pass[(y / 8) * qx + x / 8] = 1;
                if (footprint_hits(ft, pixx, pixx, pixy, pixy)) {
                    foot[(y / 8) * qx + x / 8] = 1;
                    bx0 = std::min(bx0, x / 8); bx1 = std::max(bx1, x / 8);
                    by0 = std::min(by0, y / 8); by1 = std::max(by1, y / 8);
                }
            }
        for (int b = 0; b < qy; b++)
            for (int a = 0; a < qx; a++) {
                const float x0 = (float)(a * 8) + 0.5f, y0 = (float)(b * 8) + 0.5f;
                const bool hit = footprint_hits(ft, x0, x0 + 7.0f, y0, y0 + 7.0f);
                counts[0] += hit;
                counts[1] += pass[b * qx + a];
                counts[2] += pass[b * qx + a] && !hit;
                counts[3] += hit && !foot[b * qx + a];
                counts[4] += a >= bx0 && a <= bx1 && b >= by0 && b <= by1;
            }
    }
}

// (debug aid of the fuzz tool) the first `cap` (surfel, x, y) whose pixel passes eval_pair but not the per-pixel footprint test
extern "C" int emul_footprint_misses(int P, int W, int H, const int32_t* radii, const float* rec, int cap, int32_t* out)
{
    int n = 0;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        const float* r = rec + (size_t)i * REC_FLOATS;
        const FootprintTest ft = footprint_test(r + R_FOOT, r[R_CX], r[R_CY]);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float pixx = (float)x + 0.5f, pixy = (float)y + 0.5f;
                PairEval e;
                if (eval_pair(r + R_TU, r + R_TV, r + R_TW, r[R_CX], r[R_CY], r[R_OPAC], pixx, pixy, e) &&
                    !footprint_hits(ft, pixx, pixx, pixy, pixy)) {
                    if (n < cap) { out[3 * n] = i; out[3 * n + 1] = x; out[3 * n + 2] = y; }
                    n++;
                }
            }
    }
    return n;
}

extern "C" void emul_render_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* rec,
                                const float* bg, float* final_T, uint32_t* n_contrib, float* out_color,
                                float* out_others, int cull, int lite)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)W * H;
    for (int tile = 0; tile < gx * gy; tile++)
        for (int l = 0; l < TILE * TILE; l++) {
            const int px = (tile % gx) * TILE + l % TILE, py = (tile / gx) * TILE + l / TILE;
            if (px >= W || py >= H) continue;
            const float pixx = px + 0.5f, pixy = py + 0.5f;
            FwdPixel s;
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            for (uint32_t i = r0; i < r1; i++) {
                const float* r = rec + (size_t)point_list[i] * REC_FLOATS;
                PairEval e;
                if (cull && !footprint_hits(footprint_test(r + R_FOOT, r[R_CX], r[R_CY]), pixx, pixx, pixy, pixy)) continue;
                if (!eval_pair_flat(r, r + 3, r + 6, r[R_CX], r[R_CY], r[R_OPAC], pixx, pixy, e)) continue;
                // (lite: the colour + alpha-plane instance the blend kernels run for aux_planes == VIDU4D_AUX_ALPHA)
                if (!(lite ? fwd_accumulate<true>(s, e, r + R_NX, r + R_RGB, i - r0 + 1)
                           : fwd_accumulate<false>(s, e, r + R_NX, r + R_RGB, i - r0 + 1))) break;
            }
            const size_t pid = (size_t)py * W + px;
            final_T[pid] = s.T; final_T[pid + HW] = s.dist1; final_T[pid + 2 * HW] = s.dist2;
            n_contrib[pid] = s.last_contributor; n_contrib[pid + HW] = s.median_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[ch * HW + pid] = s.C[ch] + s.T * bg[ch];
            out_others[pid] = s.D; out_others[pid + HW] = 1.0f - s.T;
            for (int ch = 0; ch < 3; ch++) out_others[pid + (2 + ch) * HW] = s.N[ch];
            out_others[pid + 5 * HW] = s.median_depth; out_others[pid + 6 * HW] = s.distortion;
            out_others[pid + 7 * HW] = s.median_weight;
        }
}

extern "C" void emul_render_bwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* rec,
                                const float* bg, const float* final_T, const uint32_t* n_contrib,
                                const float* dL_dcolor, const float* dL_dothers, double* acc /*[P][20]*/, int cull,
                                int lite)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const size_t HW = (size_t)W * H;
    for (int tile = 0; tile < gx * gy; tile++)
        for (int l = 0; l < TILE * TILE; l++) {
            const int px = (tile % gx) * TILE + l % TILE, py = (tile / gx) * TILE + l / TILE;
            if (px >= W || py >= H) continue;
            const float pixx = px + 0.5f, pixy = py + 0.5f;
            const size_t pid = (size_t)py * W + px;
            BwdPixel s;
            s.T_final = final_T[pid]; s.final_D = final_T[pid + HW]; s.final_D2 = final_T[pid + 2 * HW];
            s.last_contributor = n_contrib[pid]; s.median_contributor = n_contrib[pid + HW];
            for (int c = 0; c < 3; c++) s.dL_dpixel[c] = dL_dcolor[c * HW + pid];
            s.dL_ddepth = dL_dothers[pid]; s.dL_daccum = dL_dothers[pid + HW];
            for (int c = 0; c < 3; c++) s.dL_dnormal2D[c] = dL_dothers[pid + (2 + c) * HW];
            s.dL_dmedian_depth = dL_dothers[pid + 5 * HW]; s.dL_dreg = dL_dothers[pid + 6 * HW];
            s.dL_dmax_dweight = dL_dothers[pid + 7 * HW];
            if (lite) {  // (as blend_bwd_kernel<*, true>: the dead planes and the state behind them are not read)
                s.final_D = s.final_D2 = 0.f; s.median_contributor = 0;
                s.dL_ddepth = s.dL_dmedian_depth = s.dL_dreg = s.dL_dmax_dweight = 0.f;
                for (int c = 0; c < 3; c++) s.dL_dnormal2D[c] = 0.f;
            }
            s.T = s.T_final; s.final_A = 1.0f - s.T_final;
            s.bg_dot_dpixel = bg[0] * s.dL_dpixel[0] + bg[1] * s.dL_dpixel[1] + bg[2] * s.dL_dpixel[2];
            const uint32_t r0 = ranges[2 * tile];
            for (int ci = (int)s.last_contributor - 1; ci >= 0; ci--) {
                const uint32_t id = point_list[r0 + ci];
                const float* r = rec + (size_t)id * REC_FLOATS;
                PairEval e;
                if (cull && !footprint_hits(footprint_test(r + R_FOOT, r[R_CX], r[R_CY]), pixx, pixx, pixy, pixy)) continue;
                if (!eval_pair_flat(r, r + 3, r + 6, r[R_CX], r[R_CY], r[R_OPAC], pixx, pixy, e)) continue;
                float g[ACC_FLOATS];
                if (lite) {
                    const PairGrad pg = bwd_pair_core<true>(s, e, r + R_NX, r + R_RGB, false);
                    bwd_pair_geometry<true>(s, e, pg, r + 6, r[R_OPAC], pixx, pixy, g);
                } else {
                    bwd_pair(s, e, r + 6, r[R_OPAC], r + R_NX, r + R_RGB, pixx, pixy,
                             (uint32_t)ci + 1 == s.median_contributor, g);
                }
                for (int k = 0; k < ACC_FLOATS; k++) acc[(size_t)id * ACC_FLOATS + k] += (double)g[k];
            }
        }
}

extern "C" void emul_preprocess_bwd(int P, int D, int M, const float* means3D, const float* scales,
                                    const float* rotations, const float* shs, const float* view, const float* campos,
                                    int W, int H, float tfx, float tfy, const int32_t* radii, const float* rec,
                                    const float* acc, float* dmeans3D, float* dmeans2D, float* dcolors, float* dopacity,
                                    float* dtransMat, float* dsh, float* dscales, float* drot)
{
    const Camera cam = make_cam(view, campos, W, H, tfx, tfy, D, M);
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;  // outputs pre-zeroed by the caller
        const float* r = rec + (size_t)i * REC_FLOATS;
        const float* a = acc + (size_t)i * ACC_FLOATS;
        SurfelGrads o;
        surfel_backward(cam, means3D + 3 * i, rotations + 4 * i, scales + 2 * i, r, a, o);
        float dmean[3] = {o.dmean3D[0], o.dmean3D[1], o.dmean3D[2]};
        const float dcol[3] = {a[A_RGB], a[A_RGB + 1], a[A_RGB + 2]};
        uint32_t mask; memcpy(&mask, &r[R_CLAMP], 4);
        if (shs) sh_backward(D, M, means3D + 3 * i, cam.campos, shs + (size_t)i * M * 3, mask, dcol, dsh + (size_t)i * M * 3, dmean);
        for (int k = 0; k < 3; k++) { dmeans3D[3 * i + k] = dmean[k]; dmeans2D[3 * i + k] = o.dmean2D[k]; dcolors[3 * i + k] = dcol[k]; }
        dopacity[i] = a[A_OPAC];
        for (int k = 0; k < 9; k++) dtransMat[9 * i + k] = o.dT[k];
        dscales[2 * i] = o.dscale[0]; dscales[2 * i + 1] = o.dscale[1];
        for (int k = 0; k < 4; k++) drot[4 * i + k] = o.drot[k];
    }
}
