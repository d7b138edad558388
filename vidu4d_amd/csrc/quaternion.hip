// quaternion.hip -- Hamilton product, its VJP and double-VJP, and the conjugate, for gfx950.
//
// Replaces the CUDA helpers behind `quaternion.quaternion_mul` / `quaternion_conjugate`
// (/root/reference/lab4d/third_party/quaternion/src/quaternion.cu:28-63 forward, :66-140 backward,
// :143-214 backward-backward, :289-304 conjugate).  Operands have 4 components (w first) or 3
// (pure-vector quaternion, w = 0); gradients w.r.t. a 3-component operand have 3 components.
//
// All four are bilinear in quaternion algebra:
//   y   = a * b
//   ga  = g * conj(b)            gb  = conj(a) * g
//   ggy = da * b + a * db        g_a = g * conj(db)        g_b = conj(da) * g
// One thread handles a whole row and moves 16 bytes per operand (the reference uses one thread per
// component for the backward kernels, re-reading each row four times).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"

namespace {

struct Quat {
    float w, x, y, z;
};

__device__ __forceinline__ Quat load_q(const float* p, int64_t row, int D)
{
    if (D == 4) {
        const float4 v = reinterpret_cast<const float4*>(p)[row];
        return {v.x, v.y, v.z, v.w};
    }
    const float* q = p + row * 3;
    return {0.f, q[0], q[1], q[2]};
}
__device__ __forceinline__ void store_q(float* p, int64_t row, int D, Quat q)
{
    if (D == 4) {
        reinterpret_cast<float4*>(p)[row] = make_float4(q.w, q.x, q.y, q.z);
    } else {
        float* o = p + row * 3;
        o[0] = q.x;
        o[1] = q.y;
        o[2] = q.z;
    }
}
__device__ __forceinline__ Quat qmul(Quat a, Quat b)
{
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Quat qconj(Quat a) { return {a.w, -a.x, -a.y, -a.z}; }
__device__ __forceinline__ Quat qadd(Quat a, Quat b) { return {a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }

__global__ void qmul_kernel(int64_t B, const float* a, int Da, const float* b, int Db, float* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    store_q(out, i, 4, qmul(load_q(a, i, Da), load_q(b, i, Db)));
}

__global__ void qmul_bwd_kernel(int64_t B, const float* g, const float* a, int Da, const float* b, int Db, float* ga,
                                float* gb)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const Quat G = load_q(g, i, 4), A = load_q(a, i, Da), Bq = load_q(b, i, Db);
    store_q(ga, i, Da, qmul(G, qconj(Bq)));
    store_q(gb, i, Db, qmul(qconj(A), G));
}

__global__ void qmul_bwd_bwd_kernel(int64_t B, const float* da, const float* db, const float* g, const float* a,
                                    int Da, const float* b, int Db, float* ggy, float* g_a, float* g_b)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const Quat G = load_q(g, i, 4), A = load_q(a, i, Da), Bq = load_q(b, i, Db);
    const Quat dA = load_q(da, i, Da), dB = load_q(db, i, Db);
    store_q(ggy, i, 4, qadd(qmul(dA, Bq), qmul(A, dB)));
    store_q(g_a, i, Da, qmul(G, qconj(dB)));
    store_q(g_b, i, Db, qmul(qconj(dA), G));
}

__global__ void qconj_kernel(int64_t B, const float* q, float* out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    store_q(out, i, 4, qconj(load_q(q, i, 4)));
}

int bad_dims(int Da, int Db) { return !((Da == 3 || Da == 4) && (Db == 3 || Db == 4)); }
int done(void)
{
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}
unsigned blocks(int64_t B)
{
    (void)hipGetLastError();  // drop stale errors of unrelated earlier HIP calls
    return (unsigned)((B + 255) / 256);
}

}  // namespace

extern "C" int vidu4d_quaternion_mul(int64_t B, const float* a, int Da, const float* b, int Db, float* out, void* stream)
{
    if (B < 0 || bad_dims(Da, Db)) return VIDU4D_E_INVALID;
    if (B == 0) return VIDU4D_OK;
    if (!a || !b || !out) return VIDU4D_E_INVALID;
    hipLaunchKernelGGL(qmul_kernel, dim3(blocks(B)), dim3(256), 0, (hipStream_t)stream, B, a, Da, b, Db, out);
    return done();
}

extern "C" int vidu4d_quaternion_mul_backward(int64_t B, const float* grad_out, const float* a, int Da, const float* b,
                                              int Db, float* grad_a, float* grad_b, void* stream)
{
    if (B < 0 || bad_dims(Da, Db)) return VIDU4D_E_INVALID;
    if (B == 0) return VIDU4D_OK;
    if (!grad_out || !a || !b || !grad_a || !grad_b) return VIDU4D_E_INVALID;
    hipLaunchKernelGGL(qmul_bwd_kernel, dim3(blocks(B)), dim3(256), 0, (hipStream_t)stream, B, grad_out, a, Da, b, Db,
                       grad_a, grad_b);
    return done();
}

extern "C" int vidu4d_quaternion_mul_backward_backward(int64_t B, const float* gg_a, const float* gg_b,
                                                       const float* grad_out, const float* a, int Da, const float* b,
                                                       int Db, float* gg_out, float* g_a, float* g_b, void* stream)
{
    if (B < 0 || bad_dims(Da, Db)) return VIDU4D_E_INVALID;
    if (B == 0) return VIDU4D_OK;
    if (!gg_a || !gg_b || !grad_out || !a || !b || !gg_out || !g_a || !g_b) return VIDU4D_E_INVALID;
    hipLaunchKernelGGL(qmul_bwd_bwd_kernel, dim3(blocks(B)), dim3(256), 0, (hipStream_t)stream, B, gg_a, gg_b,
                       grad_out, a, Da, b, Db, gg_out, g_a, g_b);
    return done();
}

extern "C" int vidu4d_quaternion_conjugate(int64_t B, const float* q, float* out, void* stream)
{
    if (B < 0) return VIDU4D_E_INVALID;
    if (B == 0) return VIDU4D_OK;
    if (!q || !out) return VIDU4D_E_INVALID;
    hipLaunchKernelGGL(qconj_kernel, dim3(blocks(B)), dim3(256), 0, (hipStream_t)stream, B, q, out);
    return done();
}
