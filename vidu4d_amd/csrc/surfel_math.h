// surfel_math.h -- per-surfel and per-(pixel,surfel) arithmetic of the MI355X surfel rasterizer.
//
// Every function is __host__ __device__ so that the same source is (a) inlined into the gfx950
// kernels and (b) compiled for the host by tests/host_emul to check the formulas on the CPU before
// GPU time is spent.  Semantics follow the reference (file:line = /root/reference/gs/submodules/
// diff-surfel-rasterization/cuda_rasterizer/...); the code is written from the maths (SURVEY.md
// Appendix A), not translated.
//
// Numeric contract (DESIGN.md "Numeric conventions"): functions marked EXACT feed the integer
// binning outputs (radii, tile rects, sort keys) and are compiled with fp contraction off and a
// fixed operation order, so they agree bit-for-bit with oracle/surfel_oracle.c.  Everything else
// may be FMA-contracted; it is compared with a tolerance.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SURFEL_HD __host__ __device__ __forceinline__
#else
#define SURFEL_HD inline
#endif

namespace surfel {

constexpr int TILE = 16;              // config.h:15-16 (BLOCK_X, BLOCK_Y)
constexpr float NEAR_PLANE = 0.2f;    // auxiliary.h:35
constexpr float FAR_PLANE = 100.0f;   // auxiliary.h:36
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_EPS = 0.0001f;

// Per-surfel record written by preprocess and gathered by the blend kernels (7 x float4 = 112 B, at a 128-byte stride).
//   q0 = Tu.x Tu.y Tu.z Tv.x | q1 = Tv.y Tv.z Tw.x Tw.y | q2 = Tw.z cx cy opacity
//   q3 = n.x n.y n.z depth   | q4 = r g b clampmask(bits)
//   q5, q6 = the footprint outside which the surfel cannot contribute (contribution_footprint below)
// Stride of a record: 32 floats = ONE 128-byte cache line (round 4; the first 28 are used, the rest is never written).
// With the dense 112-byte stride of round 3 a record straddled line boundaries three times in four and the blend kernels'
// gathers pulled ~1.9 lines per record (tools/ubench/fetch_calib.hip, profiles/r04_fetch_calib.txt); a line per record
// takes a fifth off their HBM traffic and costs the projection kernel 2 us of extra stores: a wash on the clock.
#ifndef SURFEL_REC_FLOATS
#define SURFEL_REC_FLOATS 32
#endif
constexpr int REC_FLOATS = SURFEL_REC_FLOATS;
constexpr int REC_USED_FLOATS = 28;
constexpr int REC_LDS_FLOATS = 20;  // q0..q4 are staged in LDS; q5, q6 only feed the per-wave cull masks
enum RecSlot {
    R_TU = 0, R_TV = 3, R_TW = 6, R_CX = 9, R_CY = 10, R_OPAC = 11, R_NX = 12, R_DEPTH = 15, R_RGB = 16, R_CLAMP = 19,
    R_FOOT = 20  // 8 floats: contribution_footprint
};

// Per-surfel gradient accumulator filled by the backward blend (20 floats = 80 B):
//   0..8 dL_dT (Tu,Tv,Tw) | 9,10 dL_dmean2D (filter branch) | 11 dL_dopacity | 12..14 dL_dnormal
//   15 unused | 16..18 dL_dcolor | 19 unused
constexpr int ACC_FLOATS = 20;
enum AccSlot { A_T = 0, A_M2D = 9, A_OPAC = 11, A_NRM = 12, A_RGB = 16 };

struct Camera {
    float view[16];  // row-vector convention as handed over by the caller (= W^T), forward.cu:79-84
    float campos[3];
    float focal_x, focal_y;  // rasterizer_impl.cu:223-224
    float cx, cy;            // forward.cu:208 (W/2, H/2)
    float tan_fovx, tan_fovy;
    int W, H, grid_x, grid_y;
    int sh_degree, sh_coeffs;  // D, M
};

SURFEL_HD int f2i_sat(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)v;  // v_cvt_i32_f32 saturates and maps NaN to 0
#else
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
#endif
}
SURFEL_HD int imin(int a, int b) { return a < b ? a : b; }
SURFEL_HD int imax(int a, int b) { return a > b ? a : b; }

// EXACT. Tile rectangle of a surfel (auxiliary.h:64-74).
SURFEL_HD void tile_rect(float px, float py, int radius, int grid_x, int grid_y, int& x0, int& y0, int& x1, int& y1)
{
#pragma clang fp contract(off)
    const float r = (float)radius;
    x0 = imin(grid_x, imax(0, f2i_sat((px - r) / (float)TILE)));
    y0 = imin(grid_y, imax(0, f2i_sat((py - r) / (float)TILE)));
    x1 = imin(grid_x, imax(0, f2i_sat((px + r + (float)TILE - 1.0f) / (float)TILE)));
    y1 = imin(grid_y, imax(0, f2i_sat((py + r + (float)TILE - 1.0f) / (float)TILE)));
}

// EXACT. Rotation matrix of a (not necessarily unit) quaternion, row-major (auxiliary.h:188-210).
SURFEL_HD void quat_to_rotmat(const float q[4], float R[9])
{
#pragma clang fp contract(off)
    const float inv = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[3] = 2.f * (x * y + w * z);
    R[6] = 2.f * (x * z - w * y);
    R[1] = 2.f * (x * y - w * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[7] = 2.f * (y * z + w * x);
    R[2] = 2.f * (x * z + w * y);
    R[5] = 2.f * (y * z - w * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

// EXACT. out = W v, W = rotation part of the view matrix.
SURFEL_HD void view_rot(const float* vm, const float v[3], float out[3])
{
#pragma clang fp contract(off)
    for (int r = 0; r < 3; r++) out[r] = vm[r] * v[0] + vm[4 + r] * v[1] + vm[8 + r] * v[2];
}
// out = W^T v
SURFEL_HD void view_rot_t(const float* vm, const float v[3], float out[3])
{
    for (int r = 0; r < 3; r++) out[r] = vm[4 * r + 0] * v[0] + vm[4 * r + 1] * v[1] + vm[4 * r + 2] * v[2];
}
// EXACT. Camera-space position (auxiliary.h:76-84); .z is the sort depth.
SURFEL_HD void to_view(const float* vm, const float p[3], float out[3])
{
#pragma clang fp contract(off)
    for (int r = 0; r < 3; r++) out[r] = vm[r] * p[0] + vm[4 + r] * p[1] + vm[8 + r] * p[2] + vm[12 + r];
}

struct Projected {
    float T[9];       // Tu, Tv, Tw
    float normal[3];  // sign-flipped towards the camera
    float center[2];
    float extent[2];
    float depth;
    int radius;
    int x0, y0, x1, y1;  // tile rect
    uint32_t tiles;
};

// EXACT. Projection of one surfel: homography T, normal, screen AABB, radius, tile rect
// (forward.cu:75-128, :133-163, :198-244).  Returns false if the surfel is culled.
SURFEL_HD bool project_surfel(const Camera& cam, const float p_world[3], const float quat[4], const float scale[2],
                              Projected& o)
{
#pragma clang fp contract(off)
    float pv[3];
    to_view(cam.view, p_world, pv);
    if (pv[2] <= 0.2f) return false;  // auxiliary.h:175
    float R[9];
    quat_to_rotmat(quat, R);
    float p_view[3];
    {
        float t[3];
        view_rot(cam.view, p_world, t);
        for (int r = 0; r < 3; r++) p_view[r] = t[r] + cam.view[12 + r];
    }
    const float rs0[3] = {R[0] * scale[0], R[3] * scale[0], R[6] * scale[0]};
    const float rs1[3] = {R[1] * scale[1], R[4] * scale[1], R[7] * scale[1]};
    const float r2[3] = {R[2], R[5], R[8]};
    float M0[3], M1[3], tn[3];
    view_rot(cam.view, rs0, M0);
    view_rot(cam.view, rs1, M1);
    view_rot(cam.view, r2, tn);
    const float cosv = -tn[0] * p_view[0] + -tn[1] * p_view[1] + -tn[2] * p_view[2];
    if (cosv == 0.0f) return false;
    const float mult = cosv > 0 ? 1.f : -1.f;
    const float fx = cam.focal_x, fy = cam.focal_y, cx = cam.cx, cy = cam.cy;
    float* T = o.T;
    T[0] = fx * M0[0] + cx * M0[2];
    T[1] = fx * M1[0] + cx * M1[2];
    T[2] = fx * p_view[0] + cx * p_view[2];
    T[3] = fy * M0[1] + cy * M0[2];
    T[4] = fy * M1[1] + cy * M1[2];
    T[5] = fy * p_view[1] + cy * p_view[2];
    T[6] = M0[2];
    T[7] = M1[2];
    T[8] = p_view[2];
    o.normal[0] = tn[0] * mult;
    o.normal[1] = tn[1] * mult;
    o.normal[2] = tn[2] * mult;
    o.depth = pv[2];

    const float* Tu = T;
    const float* Tv = T + 3;
    const float* Tw = T + 6;
    const float d = Tw[0] * Tw[0] + Tw[1] * Tw[1] + -1.0f * (Tw[2] * Tw[2]);
    if (d == 0.0f) return false;
    const float r = 1.0f / d;
    const float f0 = r, f1 = r, f2 = -1.0f * r;
    const float px = f0 * (Tu[0] * Tw[0]) + f1 * (Tu[1] * Tw[1]) + f2 * (Tu[2] * Tw[2]);
    const float py = f0 * (Tv[0] * Tw[0]) + f1 * (Tv[1] * Tw[1]) + f2 * (Tv[2] * Tw[2]);
    const float h0x = px * px - (f0 * (Tu[0] * Tu[0]) + f1 * (Tu[1] * Tu[1]) + f2 * (Tu[2] * Tu[2]));
    const float h0y = py * py - (f0 * (Tv[0] * Tv[0]) + f1 * (Tv[1] * Tv[1]) + f2 * (Tv[2] * Tv[2]));
    o.center[0] = px;
    o.center[1] = py;
    o.extent[0] = sqrtf(h0x > 0.0f ? h0x : 0.0f);
    o.extent[1] = sqrtf(h0y > 0.0f ? h0y : 0.0f);
    // forward.cu:237-239: FilterSize is a double literal, so the reference evaluates this in fp64
    const float emax = o.extent[0] > o.extent[1] ? o.extent[0] : o.extent[1];
    const double em = (double)emax > 0.7071067811865476 ? (double)emax : 0.7071067811865476;
    const float radius = (float)ceil(3.0 * em);
    o.radius = f2i_sat(radius);
    tile_rect(px, py, o.radius, cam.grid_x, cam.grid_y, o.x0, o.y0, o.x1, o.y1);
    o.tiles = (uint32_t)(o.y1 - o.y0) * (uint32_t)(o.x1 - o.x0);
    return o.tiles != 0;
}

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

// View direction used by the SH evaluation (forward.cu:25-27).
SURFEL_HD void sh_dir(const float pos[3], const float campos[3], float dir_orig[3], float dir[3])
{
    for (int c = 0; c < 3; c++) dir_orig[c] = pos[c] - campos[c];
    const float len = sqrtf(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
    for (int c = 0; c < 3; c++) dir[c] = dir_orig[c] / len;
}

// The 16 real SH basis values (with the reference's signs) for direction (x,y,z); b[0] is the
// constant.  RGB = sum_k b[k] * sh[k] + 0.5 (forward.cu:30-63).
SURFEL_HD void sh_basis(int deg, float x, float y, float z, float b[16])
{
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -SH_C1 * y;
        b[2] = SH_C1 * z;
        b[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2_0 * xy;
            b[5] = SH_C2_1 * yz;
            b[6] = SH_C2_2 * (2.0f * zz - xx - yy);
            b[7] = SH_C2_3 * xz;
            b[8] = SH_C2_4 * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3_0 * y * (3.0f * xx - yy);
                b[10] = SH_C3_1 * xy * z;
                b[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
                b[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
                b[14] = SH_C3_5 * z * (xx - yy);
                b[15] = SH_C3_6 * x * (xx - 3.0f * yy);
            }
        }
    }
}
SURFEL_HD int sh_count(int deg) { return (deg + 1) * (deg + 1); }

// d(basis)/d(dir) for the view-direction gradient (backward.cu:58-122), component-wise.
SURFEL_HD void sh_basis_grad(int deg, float x, float y, float z, float bx[16], float by[16], float bz[16])
{
    for (int k = 0; k < 16; k++) bx[k] = by[k] = bz[k] = 0.f;
    if (deg > 0) {
        by[1] = -SH_C1;
        bz[2] = SH_C1;
        bx[3] = -SH_C1;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            bx[4] = SH_C2_0 * y;
            by[4] = SH_C2_0 * x;
            by[5] = SH_C2_1 * z;
            bz[5] = SH_C2_1 * y;
            bx[6] = SH_C2_2 * 2.f * -x;
            by[6] = SH_C2_2 * 2.f * -y;
            bz[6] = SH_C2_2 * 2.f * 2.f * z;
            bx[7] = SH_C2_3 * z;
            bz[7] = SH_C2_3 * x;
            bx[8] = SH_C2_4 * 2.f * x;
            by[8] = SH_C2_4 * 2.f * -y;
            if (deg > 2) {
                bx[9] = SH_C3_0 * 3.f * 2.f * xy;
                by[9] = SH_C3_0 * 3.f * (xx - yy);
                bx[10] = SH_C3_1 * yz;
                by[10] = SH_C3_1 * xz;
                bz[10] = SH_C3_1 * xy;
                bx[11] = SH_C3_2 * -2.f * xy;
                by[11] = SH_C3_2 * (-3.f * yy + 4.f * zz - xx);
                bz[11] = SH_C3_2 * 4.f * 2.f * yz;
                bx[12] = SH_C3_3 * -3.f * 2.f * xz;
                by[12] = SH_C3_3 * -3.f * 2.f * yz;
                bz[12] = SH_C3_3 * 3.f * (2.f * zz - xx - yy);
                bx[13] = SH_C3_4 * (-3.f * xx + 4.f * zz - yy);
                by[13] = SH_C3_4 * -2.f * xy;
                bz[13] = SH_C3_4 * 4.f * 2.f * xz;
                bx[14] = SH_C3_5 * 2.f * xz;
                by[14] = SH_C3_5 * -2.f * yz;
                bz[14] = SH_C3_5 * (xx - yy);
                bx[15] = SH_C3_6 * 3.f * (xx - yy);
                by[15] = SH_C3_6 * -3.f * 2.f * xy;
            }
        }
    }
}

// auxiliary.h:125-135
SURFEL_HD void dnormvdv3(const float v[3], const float dv[3], float out[3])
{
    const float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    out[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * invsum32;
    out[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * invsum32;
    out[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * invsum32;
}

// SH -> RGB of one surfel (forward.cu:20-71).  `sh` points at this surfel's [M][3] coefficients.
SURFEL_HD void sh_forward(int deg, const float p_world[3], const float campos[3], const float* sh, float rgb[3],
                          uint32_t& clamp_mask)
{
    float dir_orig[3], dir[3], b[16];
    sh_dir(p_world, campos, dir_orig, dir);
    sh_basis(deg, dir[0], dir[1], dir[2], b);
    const int n = sh_count(deg);
    float r = 0.f, g = 0.f, bl = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < n) {
            r += b[k] * sh[3 * k + 0];
            g += b[k] * sh[3 * k + 1];
            bl += b[k] * sh[3 * k + 2];
        }
    }
    r += 0.5f;
    g += 0.5f;
    bl += 0.5f;
    clamp_mask = (r < 0 ? 1u : 0u) | (g < 0 ? 2u : 0u) | (bl < 0 ? 4u : 0u);
    rgb[0] = fmaxf(r, 0.f);
    rgb[1] = fmaxf(g, 0.f);
    rgb[2] = fmaxf(bl, 0.f);
}

// SH backward of one surfel (backward.cu:20-139): writes dsh[M][3] (zeros beyond the active
// degree) and adds the view-direction term to dmean.
// SURFEL_PIN3: the three values are final here in program order (device code: an empty asm they are fed through), so what
// went into them is dead -- bounds live ranges in unrolled accumulation loops (see lbs.hip).
#if defined(__HIP_DEVICE_COMPILE__)
#define SURFEL_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#else
#define SURFEL_PIN3(a, b, c) ((void)0)
#endif

SURFEL_HD void sh_backward(int deg, int M, const float p_world[3], const float campos[3], const float* sh,
                           uint32_t clamp_mask, const float dcol[3], float* dsh, float dmean[3])
{
    float dir_orig[3], dir[3], b[16], bx[16], by[16], bz[16];
    sh_dir(p_world, campos, dir_orig, dir);
    sh_basis(deg, dir[0], dir[1], dir[2], b);
    sh_basis_grad(deg, dir[0], dir[1], dir[2], bx, by, bz);
    const float dRGB[3] = {(clamp_mask & 1u) ? 0.f : dcol[0], (clamp_mask & 2u) ? 0.f : dcol[1],
                           (clamp_mask & 4u) ? 0.f : dcol[2]};
    const int n = sh_count(deg);
    float ddir[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < M) {
            if (k < n) {
                const float s0 = sh[3 * k], s1 = sh[3 * k + 1], s2 = sh[3 * k + 2];  // (dsh may alias sh)
                const float dot = s0 * dRGB[0] + s1 * dRGB[1] + s2 * dRGB[2];
                ddir[0] += bx[k] * dot;
                ddir[1] += by[k] * dot;
                ddir[2] += bz[k] * dot;
                dsh[3 * k] = b[k] * dRGB[0];
                dsh[3 * k + 1] = b[k] * dRGB[1];
                dsh[3 * k + 2] = b[k] * dRGB[2];
                SURFEL_PIN3(ddir[0], ddir[1], ddir[2]);
            } else {
                dsh[3 * k] = dsh[3 * k + 1] = dsh[3 * k + 2] = 0.f;
            }
        }
    }
    float dm[3];
    dnormvdv3(dir_orig, ddir, dm);
    dmean[0] += dm[0];
    dmean[1] += dm[1];
    dmean[2] += dm[2];
}

// ---------------------------------------------------------------------------------------------
// Per-(pixel, surfel) evaluation shared by the forward and backward blend (forward.cu:358-399,
// backward.cu:282-323).
struct PairEval {
    float sx, sy, pz, ipz;
    float kx, ky, kz, lx, ly, lz;
    float dx, dy;
    float rho3d, rho2d;
    float depth, G, alpha;
    // a pair that failed the tests may hold inf / NaN here (pz == 0, rho = NaN): make the factors the gradient
    // products use finite, so that zero recurrence outputs give zero contributions
    SURFEL_HD void sanitise(bool ok)
    {
        sx = ok ? sx : 0.f;
        sy = ok ? sy : 0.f;
        ipz = ok ? ipz : 0.f;
        G = ok ? G : 0.f;
        // (takes the homography branch of bwd_pair_geometry, whose entries are all products with the above -- rho2d is a
        // sum of squares of finite numbers, so 0 <= rho2d holds without touching it; the other branch would be just as
        // safe, but a lane without a contribution must not be the reason a wave enters it)
        rho3d = ok ? rho3d : 0.f;
    }
};

SURFEL_HD float fast_exp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return expf(x);
#endif
}

SURFEL_HD float fast_rcp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);  // v_rcp_f32, 1 ulp
#else
    return 1.0f / x;
#endif
}

// Tu,Tv,Tw: homography rows; (cx,cy): projected centre; returns false when the pair is skipped.
//
// THRESHOLD-EXACT: the ray/splat intersection is a cross product of two large, nearly parallel
// plane vectors, so a different rounding of its inputs (FMA contraction or not) moves rho3d by
// ~1e-5 relative and flips the alpha >= 1/255, rho3d <= rho2d and T < 1e-4 decisions for a few
// pixels per image.  To keep the GPU and the CPU oracle on the same side of every threshold the
// operation sequence below is fixed -- explicit fmaf, contraction off -- and oracle/surfel_oracle.c
// (eval_pair) states the identical sequence.  Only rcp and exp differ (<= 2 ulp).
SURFEL_HD bool eval_pair(const float Tu[3], const float Tv[3], const float Tw[3], float cx, float cy, float opacity,
                         float pixx, float pixy, PairEval& e)
{
#pragma clang fp contract(off)
    e.kx = fmaf(pixx, Tw[0], -Tu[0]);
    e.ky = fmaf(pixx, Tw[1], -Tu[1]);
    e.kz = fmaf(pixx, Tw[2], -Tu[2]);
    e.lx = fmaf(pixy, Tw[0], -Tv[0]);
    e.ly = fmaf(pixy, Tw[1], -Tv[1]);
    e.lz = fmaf(pixy, Tw[2], -Tv[2]);
    const float px = fmaf(e.ky, e.lz, -(e.kz * e.ly));
    const float py = fmaf(e.kz, e.lx, -(e.kx * e.lz));
    const float pz = fmaf(e.kx, e.ly, -(e.ky * e.lx));
    if (pz == 0.0f) return false;
    e.pz = pz;
    const float ipz = fast_rcp(pz);
    e.ipz = ipz;
    e.sx = px * ipz;
    e.sy = py * ipz;
    e.rho3d = fmaf(e.sx, e.sx, e.sy * e.sy);
    e.dx = cx - pixx;
    e.dy = cy - pixy;
    e.rho2d = 2.0f * fmaf(e.dx, e.dx, e.dy * e.dy);  // FilterInvSquare == 2 (auxiliary.h:20-21)
    const float rho = fminf(e.rho3d, e.rho2d);
    e.depth = (e.rho3d <= e.rho2d) ? fmaf(e.sx, Tw[0], fmaf(e.sy, Tw[1], Tw[2])) : Tw[2];
    if (e.depth < NEAR_PLANE) return false;
    const float power = -0.5f * rho;
    if (power > 0.0f) return false;
    e.G = fast_exp(power);
    e.alpha = fminf(ALPHA_MAX, opacity * e.G);
    return e.alpha >= ALPHA_MIN;
}

// Same arithmetic as eval_pair (bit-identical values), without early returns: every lane of a wave
// runs the full sequence and the skip conditions come back as one predicate.  The conditions are
// the negations the reference writes (`if (x) continue`), so NaNs take the same side.
SURFEL_HD bool eval_pair_flat(const float Tu[3], const float Tv[3], const float Tw[3], float cx, float cy,
                              float opacity, float pixx, float pixy, PairEval& e)
{
#pragma clang fp contract(off)
    e.kx = fmaf(pixx, Tw[0], -Tu[0]);
    e.ky = fmaf(pixx, Tw[1], -Tu[1]);
    e.kz = fmaf(pixx, Tw[2], -Tu[2]);
    e.lx = fmaf(pixy, Tw[0], -Tv[0]);
    e.ly = fmaf(pixy, Tw[1], -Tv[1]);
    e.lz = fmaf(pixy, Tw[2], -Tv[2]);
    const float px = fmaf(e.ky, e.lz, -(e.kz * e.ly));
    const float py = fmaf(e.kz, e.lx, -(e.kx * e.lz));
    const float pz = fmaf(e.kx, e.ly, -(e.ky * e.lx));
    e.pz = pz;
    const float ipz = fast_rcp(pz);
    e.ipz = ipz;
    e.sx = px * ipz;
    e.sy = py * ipz;
    e.rho3d = fmaf(e.sx, e.sx, e.sy * e.sy);
    e.dx = cx - pixx;
    e.dy = cy - pixy;
    e.rho2d = 2.0f * fmaf(e.dx, e.dx, e.dy * e.dy);
    const float rho = fminf(e.rho3d, e.rho2d);
    e.depth = (e.rho3d <= e.rho2d) ? fmaf(e.sx, Tw[0], fmaf(e.sy, Tw[1], Tw[2])) : Tw[2];
    const float power = -0.5f * rho;
    e.G = fast_exp(power);
    e.alpha = fminf(ALPHA_MAX, opacity * e.G);
    // (`power > 0` -- forward.cu:390 -- can never hold: rho3d and rho2d are sums of squares, so rho = fminf(...) is >= +0 or
    // NaN, and power = -0.5 rho is <= -0 or NaN, neither of which compares greater than zero.  eval_pair keeps the test as the
    // reference writes it; here it was a VOPC compare and a scalar AND per pair evaluation for nothing.)
    (void)power;
    return (pz != 0.0f) & !(e.depth < NEAR_PLANE) & !(e.alpha < ALPHA_MIN);
}

// Which pixels a surfel can contribute to.  A pair contributes only if alpha = min(0.99, o*exp(-rho/2)) >= 1/255 with
// rho = min(rho3d, rho2d), i.e. only if rho3d <= rc or rho2d <= rc with rc = 2 ln(255 o).  {rho2d <= rc} is a disc of
// radius sqrt(rc/2) around the projected centre; {rho3d <= rc} is the projection of the splat-space disc of radius
// sqrt(rc).  Margins: BOX_MARGIN_PX on lengths, 0.01 % + 1e-4 on rc, 1 % on the conic's form.
#ifndef SURFEL_BOX_MARGIN_PX
#define SURFEL_BOX_MARGIN_PX 0.02f
#endif
constexpr float BOX_MARGIN_PX = SURFEL_BOX_MARGIN_PX;

// Conservative pixel-space box of the same region (rounds 1-2's cull; round 3: the fallback where fp32 cannot hold the
// conic to the margin): {rho2d <= rc} is a disc around the projected centre, {rho3d <= rc} the projection of the
// splat-space disc of radius sqrt(rc), whose exact screen AABB follows from the homography like the reference's 1-sigma
// box (forward.cu:133-163) with the first two columns scaled by sqrt(rc) -- evaluated relative to the projected centre.
// If that conic is not an ellipse in front of the camera plane the box is unbounded.
SURFEL_HD void contribution_box(const float T[9], float cx, float cy, float opacity, float box[4])
{
    const float BIG = 3.0e38f;
    const float oa = opacity * 255.0f;
    if (!(oa >= 1.0f)) {  // can never reach alpha >= 1/255 (also catches NaN)
        box[0] = box[1] = BIG;
        box[2] = box[3] = -BIG;
        return;
    }
    const float rc = 2.0f * logf(oa) * 1.0001f + 1e-4f;
    const float r2 = sqrtf(0.5f * rc);
    float x0 = cx - r2, x1 = cx + r2, y0 = cy - r2, y1 = cy + r2;
    // centred homography rows: screen coordinates relative to (cx, cy)
    const float Tw0 = T[6], Tw1 = T[7], Tw2 = T[8];
    const float Ux = T[0] - cx * Tw0, Uy = T[1] - cx * Tw1, Uz = T[2] - cx * Tw2;
    const float Vx = T[3] - cy * Tw0, Vy = T[4] - cy * Tw1, Vz = T[5] - cy * Tw2;
    const float d = rc * (Tw0 * Tw0 + Tw1 * Tw1) - Tw2 * Tw2;
    if (d < -1e-3f * Tw2 * Tw2) {
        const float f = 1.0f / d;
        const float ex = f * (rc * (Ux * Tw0 + Uy * Tw1) - Uz * Tw2);
        const float ey = f * (rc * (Vx * Tw0 + Vy * Tw1) - Vz * Tw2);
        const float hx2 = ex * ex - f * (rc * (Ux * Ux + Uy * Uy) - Uz * Uz);
        const float hy2 = ey * ey - f * (rc * (Vx * Vx + Vy * Vy) - Vz * Vz);
        const float hx = sqrtf(fmaxf(hx2, 0.f)), hy = sqrtf(fmaxf(hy2, 0.f));
        if (hx == hx && hy == hy && ex == ex && ey == ey) {
            x0 = fminf(x0, cx + ex - hx);
            x1 = fmaxf(x1, cx + ex + hx);
            y0 = fminf(y0, cy + ey - hy);
            y1 = fmaxf(y1, cy + ey + hy);
        } else {
            x0 = y0 = -BIG;
            x1 = y1 = BIG;
        }
    } else {
        x0 = y0 = -BIG;
        x1 = y1 = BIG;
    }
    const float mx = BOX_MARGIN_PX + 2e-3f * (x1 - x0), my = BOX_MARGIN_PX + 2e-3f * (y1 - y0);
    box[0] = x0 - mx;
    box[1] = y0 - my;
    box[2] = x1 + mx;
    box[3] = y1 + my;
}

// The footprint for the blend kernels' per-quadrant culls (round 3; rounds 1-2 used its bounding box).
// rho3d <= rc is a conic in the pixel plane: with k = x Tw - Tu, l = y Tw - Tv the intersection is s = (n_x, n_y) / n_z,
// n = k x l = Tu x Tv + x (Tv x Tw) + y (Tw x Tu) -- LINEAR in the pixel -- so rho3d <= rc <=> n_x^2 + n_y^2 - rc n_z^2 <= 0,
// a quadratic Q(x, y).  Where that is an ellipse it is stored by its centre and its form in offsets from the centre,
// scaled so that the inside is q(du, dv) = A du^2 + 2 B du dv + C dv^2 <= 1:
//   f[0..2] = A, B, C   f[3] = squared radius of the rho2d disc around the projected centre (with the margin)
//   f[4], f[5] = conic centre minus projected centre   f[6] = limit q is compared with: 1 + margin; -1 (and
//   f[3] = -1) where alpha >= 1/255 is out of reach: nothing hits; -2 where the conic is not an ellipse that fp32 holds
//   to the margin (thin, far away, or not an ellipse at all): f[0..3] are then the contribution box x0 y0 x1 y1 above
//   (absolute pixel coordinates; it covers the disc too).
// Everything is evaluated relative to the projected centre (fp32 cancellation far below the margins: 1 % on q, i.e.
// 0.5 % on the axes, and BOX_MARGIN_PX on the rectangle).  This only prunes work, like the box.
SURFEL_HD void contribution_footprint(const float T[9], float cx, float cy, float opacity, float f[8])
{
    const float BIG = 3.0e38f;
    for (int k = 0; k < 8; k++) f[k] = 0.f;
    const float oa = opacity * 255.0f;
    if (!(oa >= 1.0f)) {  // can never reach alpha >= 1/255 (also catches NaN)
        f[3] = -1.0f;
        f[6] = -1.0f;
        return;
    }
    const float rc = 2.0f * logf(oa) * 1.0001f + 1e-4f;
    const float rd = sqrtf(0.5f * rc) + BOX_MARGIN_PX;
    f[3] = rd * rd;
    {   // until the ellipse below is accepted: the box
        float box[4];
        contribution_box(T, cx, cy, opacity, box);
        f[0] = box[0];
        f[1] = box[1];
        f[2] = box[2];
        f[3] = box[3];
        f[6] = -2.0f;
    }
    const float* Tu = T;
    const float* Tv = T + 3;
    const float* Tw = T + 6;
    const float c0[3] = {Tu[1] * Tv[2] - Tu[2] * Tv[1], Tu[2] * Tv[0] - Tu[0] * Tv[2], Tu[0] * Tv[1] - Tu[1] * Tv[0]};
    const float c1[3] = {Tv[1] * Tw[2] - Tv[2] * Tw[1], Tv[2] * Tw[0] - Tv[0] * Tw[2], Tv[0] * Tw[1] - Tv[1] * Tw[0]};
    const float c2[3] = {Tw[1] * Tu[2] - Tw[2] * Tu[1], Tw[2] * Tu[0] - Tw[0] * Tu[2], Tw[0] * Tu[1] - Tw[1] * Tu[0]};
    const float m0[3] = {c0[0] + cx * c1[0] + cy * c2[0], c0[1] + cx * c1[1] + cy * c2[1], c0[2] + cx * c1[2] + cy * c2[2]};
    // Q(u, v) = A u^2 + 2 B u v + C v^2 + 2 D u + 2 E v + F in offsets (u, v) from the projected centre
    const float A = c1[0] * c1[0] + c1[1] * c1[1] - rc * c1[2] * c1[2];
    const float B = c1[0] * c2[0] + c1[1] * c2[1] - rc * c1[2] * c2[2];
    const float C = c2[0] * c2[0] + c2[1] * c2[1] - rc * c2[2] * c2[2];
    const float D = m0[0] * c1[0] + m0[1] * c1[1] - rc * m0[2] * c1[2];
    const float E = m0[0] * c2[0] + m0[1] * c2[1] - rc * m0[2] * c2[2];
    const float F = m0[0] * m0[0] + m0[1] * m0[1] - rc * m0[2] * m0[2];
    const float det = A * C - B * B;
    if (!(A > 0.f && C > 0.f && det > 1e-5f * A * C)) return;  // not an ellipse: the box
    const float uc = (E * B - D * C) / det, vc = (D * B - E * A) / det;
    // -Q at the centre (> 0 inside), from n(centre) itself: F + D uc + E vc is the same number, but with a conic centre
    // thousands of pixels from the projected centre (huge, strongly foreshortened surfels) its terms are 1e5 times their
    // sum and fp32 left K 7 % off -- found by tools/fuzz_footprint_cpu.py, two scenes in sixty
    const float nc[3] = {m0[0] + uc * c1[0] + vc * c2[0], m0[1] + uc * c1[1] + vc * c2[1], m0[2] + uc * c1[2] + vc * c2[2]};
    const float K = rc * nc[2] * nc[2] - (nc[0] * nc[0] + nc[1] * nc[1]);
    (void)F;
    const float ik = 1.0f / K;
    const float a = A * ik, b = B * ik, c = C * ik;
    if (!(K > 0.f) || !(a == a && b == b && c == c && uc == uc && vc == vc) || !(a < BIG && c < BIG)) return;
    // can fp32 hold the 1 % margin?  b2: squared minor semi-axis (>= 1 / trace of the form).  The centre comes out of a
    // system with condition A C / det, so it is off by ~ 2e-7 cond |centre| px, which moves q at the boundary by twice that
    // over the minor semi-axis; and where q ~ 1 along the length of a rotated ellipse its three terms are ~ (a / b)^2 =
    // trace^2 / det of the form each.  Thin or far-away ellipses (a third fuzz scene: axes 370 : 1, centre 3e5 px away) fail
    // these: the box
    const float b2 = 1.0f / (a + c);
    const float cdist = fmaxf(fabsf(uc), fabsf(vc));
    const float centre_err = 2e-7f * (A * C / det) * (cdist + 1.0f);
    if (!(centre_err * centre_err < 4e-6f * b2) || !((a + c) * (a + c) < 1.5e4f * (a * c - b * b))) return;
    f[0] = a;
    f[1] = b;
    f[2] = c;
    f[3] = rd * rd;
    f[4] = uc;
    f[5] = vc;
    f[6] = 1.01f;
}

// Can the surfel with footprint f and projected centre (cx, cy) contribute to a pixel centre inside
// [x0, x1] x [y0, y1]?  Exact for the rectangle up to the margins: q is convex with its minimum (0) at the conic centre,
// so its minimum over a rectangle that does not hold the centre lies on an edge facing the centre; on the edge
// u = ue it is at v = -B ue / C clamped to the edge, and likewise for v = ve.
struct FootprintTest {
    float A, B, C, rd2, lim, cx, cy, ecx, ecy, rBC, rBA;
};

SURFEL_HD FootprintTest footprint_test(const float f[8], float cx, float cy)
{
    FootprintTest t;
    t.A = f[0];
    t.B = f[1];
    t.C = f[2];
    t.rd2 = f[3];
    t.lim = f[6];
    t.cx = cx;
    t.cy = cy;
    t.ecx = cx + f[4];
    t.ecy = cy + f[5];
    t.rBC = t.C > 0.f ? -t.B * fast_rcp(t.C) : 0.f;
    t.rBA = t.A > 0.f ? -t.B * fast_rcp(t.A) : 0.f;
    return t;
}

SURFEL_HD bool footprint_hits(const FootprintTest& t, float x0, float x1, float y0, float y1)
{
    if (t.lim == -2.0f) return !(t.C < x0 || t.A > x1 || t.rd2 < y0 || t.B > y1);  // box mode: A B C rd2 = x0 y0 x1 y1
    const float ddx = fmaxf(fmaxf(x0 - t.cx, t.cx - x1), 0.f), ddy = fmaxf(fmaxf(y0 - t.cy, t.cy - y1), 0.f);
    const float u0 = x0 - t.ecx - BOX_MARGIN_PX, u1 = x1 - t.ecx + BOX_MARGIN_PX;
    const float v0 = y0 - t.ecy - BOX_MARGIN_PX, v1 = y1 - t.ecy + BOX_MARGIN_PX;
    const float ue = fmaxf(u0, fminf(0.f, u1)), ve = fmaxf(v0, fminf(0.f, v1));  // the rectangle's point nearest the centre, per axis
    const float vs = fmaxf(v0, fminf(t.rBC * ue, v1)), us = fmaxf(u0, fminf(t.rBA * ve, u1));
    const float q1 = ue * (t.A * ue + 2.0f * t.B * vs) + t.C * vs * vs;
    const float q2 = us * (t.A * us + 2.0f * t.B * ve) + t.C * ve * ve;
    return (ddx * ddx + ddy * ddy <= t.rd2) | (fminf(q1, q2) <= t.lim);
}

SURFEL_HD float map_depth(float depth)
{
#pragma clang fp contract(off)
    return (FAR_PLANE * depth - FAR_PLANE * NEAR_PLANE) * fast_rcp((FAR_PLANE - NEAR_PLANE) * depth);
}

// Forward per-pixel state (forward.cu:311-327).
struct FwdPixel {
    float T = 1.0f;
    float C[3] = {0, 0, 0};
    float D = 0, N[3] = {0, 0, 0};
    float dist1 = 0, dist2 = 0, distortion = 0;
    float median_depth = 0, median_weight = 0;
    uint32_t last_contributor = 0, median_contributor = 0;
    // Reference mapped depth of the tile (wave-uniform; the blend kernels set it to the mapped depth of the tile's first
    // list entry, 0 elsewhere): dist1, dist2 are the moments of m - m0.  Every use of them -- the distortion
    // sum_{j<i} w_j (m_i - m_j)^2 here, dL/dweight and dL/dm in the backward -- is a function of DIFFERENCES of mapped
    // depths, written by the reference (forward.cu:400-407, backward.cu:343-362) as m^2 A + M2 - 2 m M1 with
    // A = sum w, M1 = sum w m, M2 = sum w m^2: three terms of size m^2 ~ 0.9 that cancel to (depth spread)^2 ~ 1e-4.  About
    // a reference inside the tile's depth range the same three terms are of the size of the result.
    float m0 = 0;
};

// One accepted sample (forward.cu:400-438).  Returns false (and leaves the state untouched) when
// the pixel saturates on this sample.
//
// Which planes a blend instance carries (Vidu4dSurfelForwardArgs::aux_planes):
//   BLEND_FULL  everything the reference computes;
//   BLEND_LITE  colour + alpha plane (aux_planes names only the alpha plane: colour + silhouette losses, Stage-3 before the
//               regularisers switch on): depth, normal, median and distortion accumulations are not carried, their planes
//               come out as zeros;
//   BLEND_GEOM  colour + planes 0-4 (depth, alpha, normal: Stage-3 after step 8000 with the upstream defaults lambda_dist = 0,
//               depth_ratio = 0 -- lab4d/config.py:181, gs/arguments/__init__.py:68 -- where only these planes are read,
//               lab4d/engine/model.py:817-842): the median sample and the distortion moments are not carried, planes 5-7
//               come out as zeros.
// What an instance does carry is the same operations on the same operands as in the full version: bit-identical planes.
constexpr int BLEND_FULL = 0, BLEND_LITE = 1, BLEND_GEOM = 2;

template <int MODE = BLEND_FULL>
SURFEL_HD bool fwd_accumulate(FwdPixel& s, const PairEval& e, const float normal[3], const float rgb[3],
                              uint32_t contributor)
{
#pragma clang fp contract(off)
    const float test_T = s.T * (1.0f - e.alpha);  // THRESHOLD-EXACT (same two roundings as the oracle)
    if (test_T < T_EPS) return false;
    const float w = e.alpha * s.T;
    if (MODE == BLEND_FULL) {
        const float A = 1.0f - s.T;
        const float m = map_depth(e.depth) - s.m0;
        const float error = fmaf(m * m, A, fmaf(-2.0f * m, s.dist1, s.dist2));
        s.distortion = fmaf(error, w, s.distortion);
        if (s.T > 0.5f) {
            s.median_depth = e.depth;
            s.median_weight = w;
            s.median_contributor = contributor;
        }
        s.dist1 = fmaf(m, w, s.dist1);
        s.dist2 = fmaf(m * m, w, s.dist2);
    }
    if (MODE != BLEND_LITE) {
        for (int ch = 0; ch < 3; ch++) s.N[ch] = fmaf(normal[ch], w, s.N[ch]);
        s.D = fmaf(e.depth, w, s.D);
    }
    for (int ch = 0; ch < 3; ch++) s.C[ch] = fmaf(rgb[ch], w, s.C[ch]);
    s.T = test_T;
    s.last_contributor = contributor;
    return true;
}

// Backward per-pixel state (backward.cu:192-244).  The reference carries last_alpha / last_color /
// last_depth / last_normal and folds the previous sample into the accum_*_rec running values at the
// START of the next processed sample (backward.cu:337, :371, :375, :380).  Here the same update
// (same operands, same operations, hence the same bits) is applied at the END of the sample that
// produced them, which removes eight carried registers and their copies from the inner loop.
struct BwdPixel {
    float T, T_final;
    float dL_dpixel[3];
    float dL_ddepth, dL_daccum, dL_dreg, dL_dnormal2D[3], dL_dmedian_depth, dL_dmax_dweight;
    float final_D, final_D2, final_A;
    float bg_dot_dpixel;
    float accum_rec[3] = {0, 0, 0};
    float accum_depth_rec = 0, accum_alpha_rec = 0, accum_normal_rec[3] = {0, 0, 0};
    float last_dL_dT = 0;
    uint32_t last_contributor, median_contributor;
    float m0 = 0;  // the tile's reference mapped depth (FwdPixel): final_D, final_D2 are moments of m - m0
};

// The part of one (pixel, surfel) pair's backward that runs through the pixel's back-to-front recurrences
// (backward.cu:325-400): advances the pixel state and returns the blend weight w = alpha * T, dL/dalpha
// (final), dL/dz so far (median + distortion + depth terms).  `is_median` == (contributor == median_contributor - 1).
struct PairGrad {
    float w, dL_dalpha, dL_dz;
};

// BLEND_LITE: only dL/dcolour and dL/dalpha-plane are live (every other upstream gradient plane is zero by the caller's
// promise, aux_planes): the depth / normal / median / distortion chains, which would multiply by those zeros, are left out.
// BLEND_GEOM: the planes 5-7 are zero by the caller's promise: the median and distortion chains are left out; what remains
// is what the full version computes with those zeros (x + 0 and 0 * finite are exact: the same bits).
template <int MODE = BLEND_FULL>
SURFEL_HD PairGrad bwd_pair_core(BwdPixel& s, const PairEval& e, const float normal[3], const float rgb[3],
                                 bool is_median)
{
    const float alpha = e.alpha, c_d = e.depth;
    const float one_m_alpha = 1.f - alpha;
    const float inv_1ma = fast_rcp(one_m_alpha);
    s.T = s.T * inv_1ma;
    const float w = alpha * s.T;
    float dL_dalpha = 0.0f;
    for (int ch = 0; ch < 3; ch++) dL_dalpha += (rgb[ch] - s.accum_rec[ch]) * s.dL_dpixel[ch];
    if (MODE == BLEND_LITE) {
        dL_dalpha += (1.0f - s.accum_alpha_rec) * s.dL_daccum;
        dL_dalpha *= s.T;
        dL_dalpha += (-s.T_final * inv_1ma) * s.bg_dot_dpixel;
        for (int ch = 0; ch < 3; ch++) s.accum_rec[ch] = fmaf(alpha, rgb[ch] - s.accum_rec[ch], s.accum_rec[ch]);
        s.accum_alpha_rec = fmaf(alpha, 1.0f - s.accum_alpha_rec, s.accum_alpha_rec);
        PairGrad r;
        r.w = w;
        r.dL_dalpha = dL_dalpha;
        r.dL_dz = 0.f;
        return r;
    }
    float dL_dz = 0.0f;
    if (MODE == BLEND_FULL) {
        float dL_dweight = 0.0f;
        // m_d = far (d - near) / ((far - near) d) and its derivative far near / ((far - near) d^2) from ONE reciprocal
        const float inv_d = fast_rcp((FAR_PLANE - NEAR_PLANE) * c_d);
        const float m_d = (FAR_PLANE * c_d - FAR_PLANE * NEAR_PLANE) * inv_d - s.m0;
        const float dmd_dd = (FAR_PLANE * NEAR_PLANE * (FAR_PLANE - NEAR_PLANE)) * inv_d * inv_d;
        if (is_median) {
            dL_dz += s.dL_dmedian_depth;
            dL_dweight += s.dL_dmax_dweight;
        }
        dL_dweight += (s.final_D2 + m_d * m_d * s.final_A - 2.0f * m_d * s.final_D) * s.dL_dreg;
        dL_dalpha += dL_dweight - s.last_dL_dT;
        s.last_dL_dT = dL_dweight * alpha + one_m_alpha * s.last_dL_dT;
        const float dL_dmd = 2.0f * w * (m_d * s.final_A - s.final_D) * s.dL_dreg;
        dL_dz += dL_dmd * dmd_dd;
    }

    dL_dalpha += (c_d - s.accum_depth_rec) * s.dL_ddepth;
    dL_dalpha += (1.0f - s.accum_alpha_rec) * s.dL_daccum;
    for (int ch = 0; ch < 3; ch++) dL_dalpha += (normal[ch] - s.accum_normal_rec[ch]) * s.dL_dnormal2D[ch];
    dL_dalpha *= s.T;
    dL_dalpha += (-s.T_final * inv_1ma) * s.bg_dot_dpixel;

    // fold this sample into the running "what lies behind" values (see the struct comment)
    // x <- x + alpha (c - x): the differences are the ones dL_dalpha was just built from, so each running value costs one
    // FMA instead of a multiplication and an FMA (8 of the ~210 VALU instructions of a pair evaluation: blend_bwd -2.4 %).
    // backward.cu:337-380 writes alpha c + (1 - alpha) x; the two round differently by an ulp of x.
    for (int ch = 0; ch < 3; ch++) {
        s.accum_rec[ch] = fmaf(alpha, rgb[ch] - s.accum_rec[ch], s.accum_rec[ch]);
        s.accum_normal_rec[ch] = fmaf(alpha, normal[ch] - s.accum_normal_rec[ch], s.accum_normal_rec[ch]);
    }
    s.accum_depth_rec = fmaf(alpha, c_d - s.accum_depth_rec, s.accum_depth_rec);
    s.accum_alpha_rec = fmaf(alpha, 1.0f - s.accum_alpha_rec, s.accum_alpha_rec);
    dL_dz += w * s.dL_ddepth;
    PairGrad r;
    r.w = w;
    r.dL_dalpha = dL_dalpha;
    r.dL_dz = dL_dz;
    return r;
}

// Gradient contributions of one (pixel, surfel) pair, in AccSlot order (backward.cu:325-446), from the pair's
// recurrence outputs `pg`.  Every entry is linear in (pg.w, pg.dL_dalpha, pg.dL_dz): a pair that does not
// contribute passes pg = 0 and `e` with finite sx, sy, ipz, G (PairEval::sanitise) and gets exact zeros without
// a branch.  The six dL/dTu, dL/dTv sums are accumulated with the OPPOSITE sign (+dk, +dl: the negations are
// seven VALU instructions per pair); surfel_backward flips them back (exact).
template <int MODE = BLEND_FULL>
SURFEL_HD void bwd_pair_geometry(const BwdPixel& s, const PairEval& e, const PairGrad& pg, const float Tw[3],
                                 float opacity, float pixx, float pixy, float g[ACC_FLOATS])
{
    constexpr bool LITE = MODE == BLEND_LITE;
    const float G = e.G, dL_dz = LITE ? 0.f : pg.dL_dz;
    for (int ch = 0; ch < 3; ch++) {
        g[A_RGB + ch] = pg.w * s.dL_dpixel[ch];
        g[A_NRM + ch] = LITE ? 0.f : pg.w * s.dL_dnormal2D[ch];
    }
    const float dL_dG = opacity * pg.dL_dalpha;  // straight-through the 0.99 clamp (backward.cu:400)
    g[A_OPAC] = G * pg.dL_dalpha;
    g[15] = 0.f;
    g[19] = 0.f;
    if (LITE) {  // (dL_dz == 0: the same expressions without the terms it multiplies)
        if (e.rho3d <= e.rho2d) {
            const float dL_dsx = dL_dG * -G * e.sx, dL_dsy = dL_dG * -G * e.sy;
            const float dpx = dL_dsx * e.ipz, dpy = dL_dsy * e.ipz, dpz = -(dpx * e.sx + dpy * e.sy);
            const float dkx = e.ly * dpz - e.lz * dpy, dky = e.lz * dpx - e.lx * dpz, dkz = e.lx * dpy - e.ly * dpx;
            const float dlx = dpy * e.kz - dpz * e.ky, dly = dpz * e.kx - dpx * e.kz, dlz = dpx * e.ky - dpy * e.kx;
            g[A_T + 0] = dkx;
            g[A_T + 1] = dky;
            g[A_T + 2] = dkz;
            g[A_T + 3] = dlx;
            g[A_T + 4] = dly;
            g[A_T + 5] = dlz;
            g[A_T + 6] = pixx * dkx + pixy * dlx;
            g[A_T + 7] = pixx * dky + pixy * dly;
            g[A_T + 8] = pixx * dkz + pixy * dlz;
            g[A_M2D + 0] = 0.f;
            g[A_M2D + 1] = 0.f;
        } else {
            for (int k = 0; k < 9; k++) g[A_T + k] = 0.f;
            g[A_M2D + 0] = dL_dG * (-G * 2.0f * e.dx);
            g[A_M2D + 1] = dL_dG * (-G * 2.0f * e.dy);
        }
        return;
    }
    if (e.rho3d <= e.rho2d) {
        const float dL_dsx = dL_dG * -G * e.sx + dL_dz * Tw[0];
        const float dL_dsy = dL_dG * -G * e.sy + dL_dz * Tw[1];
        const float dsx_pz = dL_dsx * e.ipz, dsy_pz = dL_dsy * e.ipz;
        const float dpx = dsx_pz, dpy = dsy_pz, dpz = -(dsx_pz * e.sx + dsy_pz * e.sy);
        // dL_dk = l x dL_dp ; dL_dl = dL_dp x k
        const float dkx = e.ly * dpz - e.lz * dpy, dky = e.lz * dpx - e.lx * dpz, dkz = e.lx * dpy - e.ly * dpx;
        const float dlx = dpy * e.kz - dpz * e.ky, dly = dpz * e.kx - dpx * e.kz, dlz = dpx * e.ky - dpy * e.kx;
        g[A_T + 0] = dkx;  // (sign: see above)
        g[A_T + 1] = dky;
        g[A_T + 2] = dkz;
        g[A_T + 3] = dlx;
        g[A_T + 4] = dly;
        g[A_T + 5] = dlz;
        g[A_T + 6] = pixx * dkx + pixy * dlx + dL_dz * e.sx;
        g[A_T + 7] = pixx * dky + pixy * dly + dL_dz * e.sy;
        g[A_T + 8] = pixx * dkz + pixy * dlz + dL_dz;
        g[A_M2D + 0] = 0.f;
        g[A_M2D + 1] = 0.f;
    } else {
        for (int k = 0; k < 8; k++) g[A_T + k] = 0.f;
        g[A_T + 8] = dL_dz;
        g[A_M2D + 0] = dL_dG * (-G * 2.0f * e.dx);
        g[A_M2D + 1] = dL_dG * (-G * 2.0f * e.dy);
    }
}

SURFEL_HD void bwd_pair(BwdPixel& s, const PairEval& e, const float Tw[3], float opacity, const float normal[3],
                        const float rgb[3], float pixx, float pixy, bool is_median, float g[ACC_FLOATS])
{
    const PairGrad pg = bwd_pair_core(s, e, normal, rgb, is_median);
    bwd_pair_geometry(s, e, pg, Tw, opacity, pixx, pixy, g);
}

// ---------------------------------------------------------------------------------------------
// Per-surfel backward: AABB-centre chain + densification statistic (backward.cu:599-649), then
// homography / normal vjp (backward.cu:451-529, auxiliary.h:213-257).
struct SurfelGrads {
    float dmean3D[3];
    float dscale[2];
    float drot[4];
    float dmean2D[3];  // the densification statistic, not a gradient (backward.cu:645-648)
    float dT[9];       // dL_dtransMat after the AABB chain
};

SURFEL_HD void surfel_backward(const Camera& cam, const float p_world[3], const float quat[4], const float scale[2],
                               const float T[9], const float acc[ACC_FLOATS], SurfelGrads& o)
{
    // ---- AABB centre chain
    const float* T0 = T;
    const float* T1 = T + 3;
    const float* T3 = T + 6;
    const float gx = acc[A_M2D], gy = acc[A_M2D + 1];
    const float d = T3[0] * T3[0] + T3[1] * T3[1] + -1.0f * (T3[2] * T3[2]);
    const float r = 1.0f / d;
    const float f[3] = {r, r, -r};
    const float sgn[3] = {1.0f, 1.0f, -1.0f};
    float dT0[3], dT1[3], dT3[3], df[3];
    for (int c = 0; c < 3; c++) {
        dT0[c] = gx * f[c] * T3[c];
        dT1[c] = gy * f[c] * T3[c];
        dT3[c] = gx * f[c] * T0[c] + gy * f[c] * T1[c];
        df[c] = (gx * T0[c] * T3[c]) + (gy * T1[c] * T3[c]);
    }
    const float dL_dd = (df[0] * f[0] + df[1] * f[1] + df[2] * f[2]) * (-1.0f / d);
    for (int c = 0; c < 3; c++) dT3[c] += dL_dd * (sgn[c] * T3[c] * 2.0f);
    for (int c = 0; c < 3; c++) {
        o.dT[c] = dT0[c] - acc[A_T + c];  // (the blend kernel accumulates these six with the opposite sign)
        o.dT[3 + c] = dT1[c] - acc[A_T + 3 + c];
        o.dT[6 + c] = acc[A_T + 6 + c] + dT3[c];
    }
    const float z = T[8];
    o.dmean2D[0] = o.dT[2] * z * (cam.focal_x * cam.tan_fovx);
    o.dmean2D[1] = o.dT[5] * z * (cam.focal_y * cam.tan_fovy);
    o.dmean2D[2] = 0.f;

    // ---- homography vjp
    const float* dT = o.dT;
    const float fx = cam.focal_x, fy = cam.focal_y;
    const float cx = cam.focal_x * cam.tan_fovx, cy = cam.focal_y * cam.tan_fovy;  // backward.cu:570
    float R[9];
    quat_to_rotmat(quat, R);
    float p_view[3];
    {
        float t[3];
        view_rot(cam.view, p_world, t);
        for (int k = 0; k < 3; k++) p_view[k] = t[k] + cam.view[12 + k];
    }
    float dM[3][3];
    for (int j = 0; j < 3; j++) {
        dM[j][0] = fx * dT[j];
        dM[j][1] = fy * dT[3 + j];
        dM[j][2] = cx * dT[j] + cy * dT[3 + j] + dT[6 + j];
    }
    float dRS0[3], dRS1[3], dpw[3], dtn[3];
    view_rot_t(cam.view, dM[0], dRS0);
    view_rot_t(cam.view, dM[1], dRS1);
    view_rot_t(cam.view, dM[2], dpw);
    const float dN[3] = {acc[A_NRM], acc[A_NRM + 1], acc[A_NRM + 2]};
    view_rot_t(cam.view, dN, dtn);
    {
        const float r2[3] = {R[2], R[5], R[8]};
        float tn[3];
        view_rot(cam.view, r2, tn);
        const float cosv = -tn[0] * p_view[0] + -tn[1] * p_view[1] + -tn[2] * p_view[2];
        const float mult = cosv > 0 ? 1.f : -1.f;
        for (int c = 0; c < 3; c++) dtn[c] *= mult;
    }
    float vR[3][3];  // column-major v_R[c][r]
    for (int k = 0; k < 3; k++) {
        vR[0][k] = dRS0[k] * scale[0];
        vR[1][k] = dRS1[k] * scale[1];
        vR[2][k] = dtn[k];
    }
    {
        const float inv = 1.0f / sqrtf(quat[3] * quat[3] + quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2]);
        const float w = quat[0] * inv, x = quat[1] * inv, y = quat[2] * inv, z2 = quat[3] * inv;
        o.drot[0] = 2.f * (x * (vR[1][2] - vR[2][1]) + y * (vR[2][0] - vR[0][2]) + z2 * (vR[0][1] - vR[1][0]));
        o.drot[1] = 2.f * (-2.f * x * (vR[1][1] + vR[2][2]) + y * (vR[0][1] + vR[1][0]) + z2 * (vR[0][2] + vR[2][0]) +
                           w * (vR[1][2] - vR[2][1]));
        o.drot[2] = 2.f * (x * (vR[0][1] + vR[1][0]) - 2.f * y * (vR[0][0] + vR[2][2]) + z2 * (vR[1][2] + vR[2][1]) +
                           w * (vR[2][0] - vR[0][2]));
        o.drot[3] = 2.f * (x * (vR[0][2] + vR[2][0]) + y * (vR[1][2] + vR[2][1]) - 2.f * z2 * (vR[0][0] + vR[1][1]) +
                           w * (vR[0][1] - vR[1][0]));
    }
    o.dscale[0] = dRS0[0] * R[0] + dRS0[1] * R[3] + dRS0[2] * R[6];
    o.dscale[1] = dRS1[0] * R[1] + dRS1[1] * R[4] + dRS1[2] * R[7];
    o.dmean3D[0] = dpw[0];
    o.dmean3D[1] = dpw[1];
    o.dmean3D[2] = dpw[2];
}

}  // namespace surfel
