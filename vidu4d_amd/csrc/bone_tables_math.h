// bone_tables_math.h -- the arithmetic of bone_tables.hip, free of any launch syntax: hipcc compiles it into the kernels,
// and tests/test_bone_tables_cpu.py compiles the SAME header with g++ to check values and dual-number gradients against
// torch autograd without a GPU.  See bone_tables.hip for what it replaces.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VIDU4D_HD __device__ __forceinline__
#else
#define VIDU4D_HD inline
struct float4 {
    float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
#endif

namespace bone_tables {

// ---- first-order dual number: value and ONE tangent ------------------------------------------------------------------
struct Dual {
    float v, d;
};
VIDU4D_HD Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
VIDU4D_HD Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
VIDU4D_HD Dual operator-(Dual a) { return {-a.v, -a.d}; }
VIDU4D_HD Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
VIDU4D_HD Dual operator*(float a, Dual b) { return {a * b.v, a * b.d}; }
VIDU4D_HD Dual operator/(Dual a, Dual b)
{
    const float q = a.v / b.v;
    return {q, (a.d - q * b.d) / b.v};
}
VIDU4D_HD Dual operator/(float a, Dual b)
{
    const float q = a / b.v;
    return {q, -q * b.d / b.v};
}
VIDU4D_HD Dual operator-(float a, Dual b) { return {a - b.v, -b.d}; }

VIDU4D_HD float value(float a) { return a; }
VIDU4D_HD float value(Dual a) { return a.v; }
VIDU4D_HD float m_sqrt(float a) { return sqrtf(a); }
// (|x| at x = 0: torch's norm backward masks the 0/0 to zero)
VIDU4D_HD Dual m_sqrt(Dual a)
{
    const float r = sqrtf(a.v);
    return {r, r > 0.f ? 0.5f * a.d / r : 0.f};
}
VIDU4D_HD float m_sin(float a) { return sinf(a); }
VIDU4D_HD Dual m_sin(Dual a) { return {sinf(a.v), cosf(a.v) * a.d}; }
VIDU4D_HD float m_cos(float a) { return cosf(a); }
VIDU4D_HD Dual m_cos(Dual a) { return {cosf(a.v), -sinf(a.v) * a.d}; }
VIDU4D_HD float m_select(bool c, float a, float b) { return c ? a : b; }
VIDU4D_HD Dual m_select(bool c, Dual a, Dual b) { return c ? a : b; }
VIDU4D_HD float lift(float a, float) { return a; }   // constant of type T from a float
VIDU4D_HD Dual lift(float a, Dual) { return {a, 0.f}; }

template <class T>
struct Quat {
    T w, x, y, z;
};
template <class T>
VIDU4D_HD Quat<T> qmul(Quat<T> a, Quat<T> b)
{
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
template <class T>
VIDU4D_HD Quat<T> qconj(Quat<T> a)
{
    return {a.w, -a.x, -a.y, -a.z};
}

// axis-angle (3) and translation (3) -> unit dual quaternion (real, dual); quat_transform.axis_angle_to_quaternion's
// small-angle series included.
template <class T>
VIDU4D_HD void head_to_dual_quaternion(const T* p, Quat<T>& qr, Quat<T>& qd)
{
    const T angle = m_sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    const T half = 0.5f * angle;
    const bool small = fabsf(value(angle)) < 1e-6f;
    const T one = lift(1.f, angle);
    const T k = m_select(small, lift(0.5f, angle) - (1.f / 48.f) * (angle * angle), m_sin(half) / m_select(small, one, angle));
    qr = {m_cos(half), p[0] * k, p[1] * k, p[2] * k};
    const Quat<T> t = {lift(0.f, angle), p[3], p[4], p[5]};
    const Quat<T> h = qmul(t, qr);
    qd = {0.5f * h.w, 0.5f * h.x, 0.5f * h.y, 0.5f * h.z};
}

// se3 (8): the frame's bone transform relative to the rest pose's, t * rest^-1 (conjugate of both parts).
template <class T>
VIDU4D_HD void relative_to_rest(const T* frame6, const T* rest6, T* out8)
{
    Quat<T> tr, td, rr, rd;
    head_to_dual_quaternion(frame6, tr, td);
    head_to_dual_quaternion(rest6, rr, rd);
    const Quat<T> cr = qconj(rr), cd = qconj(rd);
    const Quat<T> r = qmul(tr, cr);
    const Quat<T> a = qmul(tr, cd), b = qmul(td, cr);
    out8[0] = r.w, out8[1] = r.x, out8[2] = r.y, out8[3] = r.z;
    out8[4] = a.w + b.w, out8[5] = a.x + b.x, out8[6] = a.y + b.y, out8[7] = a.z + b.z;
}

// tab (12): rows k = 0..2 of R (object -> bone, the 2 / |q|^2 form) times inv_gauss[k] (9), then t[k] * inv_gauss[k] (3).
template <class T>
VIDU4D_HD void rest_bone_map(const T* rest6, const T* ig3, T* tab12)
{
    Quat<T> rr, rd;
    head_to_dual_quaternion(rest6, rr, rd);
    const Quat<T> q = qconj(rr);
    const Quat<T> h = qmul(qconj(rd), rr);   // (inverse's dual part) * conj(inverse's real part)
    const T t[3] = {2.f * h.x, 2.f * h.y, 2.f * h.z};
    const T s = 2.f / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    const T R[9] = {1.f - s * (q.y * q.y + q.z * q.z), s * (q.x * q.y - q.z * q.w), s * (q.x * q.z + q.y * q.w),
                    s * (q.x * q.y + q.z * q.w), 1.f - s * (q.x * q.x + q.z * q.z), s * (q.y * q.z - q.x * q.w),
                    s * (q.x * q.z - q.y * q.w), s * (q.y * q.z + q.x * q.w), 1.f - s * (q.x * q.x + q.y * q.y)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = 0; j < 3; ++j) tab12[3 * k + j] = R[3 * k + j] * ig3[k];
        tab12[9 + k] = t[k] * ig3[k];
    }
}

// so3 / trans are the heads' outputs, (rows, B, 3) each; row m < M a frame, the rest pose its own (B, 3) pair.
VIDU4D_HD void load6(const float* so3, const float* trans, int64_t row, float* p)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = so3[row * 3 + k], p[3 + k] = trans[row * 3 + k];
}

VIDU4D_HD void bone_tables_fwd_body(int i, int M, int B, const float* so3_t, const float* trans_t, const float* so3_r,
                                       const float* trans_r, const float* inv_gauss, float* se3_qr, float* se3_qd,
                                       float* bone_A, float* bone_c)
{
    if (i >= (M + 1) * B) return;
    const int m = i / B, b = i - m * B;
    float rest[6];
    load6(so3_r, trans_r, b, rest);
    if (m < M) {
        float f[6], o[8];
        load6(so3_t, trans_t, (int64_t)m * B + b, f);
        relative_to_rest(f, rest, o);
        reinterpret_cast<float4*>(se3_qr)[(int64_t)m * B + b] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4*>(se3_qd)[(int64_t)m * B + b] = make_float4(o[4], o[5], o[6], o[7]);
    } else if (bone_A) {
        float tab[12];
        const float ig[3] = {inv_gauss[b * 3], inv_gauss[b * 3 + 1], inv_gauss[b * 3 + 2]};
        rest_bone_map(rest, ig, tab);
#pragma unroll
        for (int k = 0; k < 9; ++k) bone_A[b * 9 + k] = tab[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) bone_c[b * 3 + k] = tab[9 + k];
    }
}

// One thread per (bone, input direction): 6 M directions of the frames' inputs (one evaluation each), 6 of the rest
// pose's (M evaluations of the relative transform + one of the bone map: every thread owns its output, no atomics), 3 of
// the inverse extents.  (One thread per bone with the directions in a loop measured 50 us on the backward graph's critical
// path: 33 evaluations in sequence.)
#if defined(__HIPCC__)
__host__
#endif
VIDU4D_HD int bone_tables_bwd_dirs(int M) { return 6 * M + 9; }

VIDU4D_HD float relative_tangent_dot(const float* f, const float* rest, int seed_f, int seed_r, const float* g_qr,
                                     const float* g_qd, int64_t row)
{
    Dual f_d[6], r_d[6], o[8];
#pragma unroll
    for (int k = 0; k < 6; ++k) f_d[k] = {f[k], k == seed_f ? 1.f : 0.f}, r_d[k] = {rest[k], k == seed_r ? 1.f : 0.f};
    relative_to_rest(f_d, r_d, o);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (g_qr) acc += g_qr[row * 4 + k] * o[k].d;
        if (g_qd) acc += g_qd[row * 4 + k] * o[4 + k].d;
    }
    return acc;
}

VIDU4D_HD void bone_tables_bwd_body(int idx, int M, int B, const float* so3_t, const float* trans_t, const float* so3_r,
                                    const float* trans_r, const float* inv_gauss, const float* g_qr, const float* g_qd,
                                    const float* g_A, const float* g_c, float* g_so3_t, float* g_trans_t,
                                    float* g_so3_r, float* g_trans_r, float* g_inv_gauss)
{
    const int ndir = bone_tables_bwd_dirs(M);
    if (idx >= B * ndir) return;
    const int b = idx / ndir, dir = idx - b * ndir;
    float rest[6];
    load6(so3_r, trans_r, b, rest);
    if (dir < 6 * M) {   // a frame's input
        const int m = dir / 6, k = dir - 6 * m;
        const int64_t row = (int64_t)m * B + b;
        float f[6];
        load6(so3_t, trans_t, row, f);
        (k < 3 ? g_so3_t : g_trans_t)[row * 3 + k % 3] = relative_tangent_dot(f, rest, k, -1, g_qr, g_qd, row);
        return;
    }
    const int k = dir - 6 * M;   // 0-5: the rest pose's six inputs, 6-8: the inverse extents
    if (k >= 6 && !g_inv_gauss) return;
    float acc = 0.f;
    if (k < 6) {
        for (int m = 0; m < M; ++m) {
            const int64_t row = (int64_t)m * B + b;
            float f[6];
            load6(so3_t, trans_t, row, f);
            acc += relative_tangent_dot(f, rest, -1, k, g_qr, g_qd, row);
        }
    }
    if (g_A) {
        Dual r_d[6], ig_d[3], tab[12];
#pragma unroll
        for (int j = 0; j < 6; ++j) r_d[j] = {rest[j], j == k ? 1.f : 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j) ig_d[j] = {inv_gauss[b * 3 + j], j == k - 6 ? 1.f : 0.f};
        rest_bone_map(r_d, ig_d, tab);
#pragma unroll
        for (int j = 0; j < 9; ++j) acc += g_A[b * 9 + j] * tab[j].d;
        if (g_c) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc += g_c[b * 3 + j] * tab[9 + j].d;
        }
    }
    if (k < 6)
        (k < 3 ? g_so3_r : g_trans_r)[b * 3 + k % 3] = acc;
    else
        g_inv_gauss[b * 3 + k - 6] = acc;
}

// ---- the camera network's tail (CameraMLP.get_vals, reference lab4d/nnutils/pose.py:120-150): the rotation head's raw
// quaternion and the video's learnable base quaternion, both normalised (F.normalize: v / max(|v|, 1e-12)), composed.
template <class T>
VIDU4D_HD Quat<T> normalised(const T* v)
{
    const T n = m_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    const bool tiny = !(value(n) > 1e-12f);
    const T inv = 1.f / m_select(tiny, lift(1e-12f, n), n);
    return {v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
}
template <class T>
VIDU4D_HD void camera_rotation(const T* raw4, const T* base4, T* out4)
{
    const Quat<T> q = qmul(normalised(raw4), normalised(base4));
    out4[0] = q.w, out4[1] = q.x, out4[2] = q.y, out4[3] = q.z;
}

VIDU4D_HD void camera_tail_fwd_body(int m, int M, const float* raw, const float* base, float* out)
{
    if (m >= M) return;
    float o[4];
    camera_rotation(raw + 4 * m, base + 4 * m, o);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * m + k] = o[k];
}

// one thread per (row, input direction): 0-3 the raw quaternion's components, 4-7 the base quaternion's
VIDU4D_HD void camera_tail_bwd_body(int idx, int M, const float* raw, const float* base, const float* g_out, float* g_raw,
                                    float* g_base)
{
    if (idx >= 8 * M) return;
    const int m = idx >> 3, dir = idx & 7;
    Dual r[4], b[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = {raw[4 * m + k], k == dir ? 1.f : 0.f}, b[k] = {base[4 * m + k], k + 4 == dir ? 1.f : 0.f};
    camera_rotation(r, b, o);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += g_out[4 * m + k] * o[k].d;
    (dir < 4 ? g_raw : g_base)[4 * m + (dir & 3)] = acc;
}

}  // namespace bone_tables
