// bone_tables.hip -- from the articulation network's heads to what the fused warp reads, in one launch per direction.
//
// With networks that train (--gs_optim_warp=True, /root/reference/lab4d/config.py:157) every fitting step evaluates, for
// the B bones of the M frames of the batch and of the rest pose,
//     axis-angle, translation -> unit dual quaternion          (lab4d/nnutils/pose.py:300-323,
//                                                                lab4d/utils/quat_transform.py:341-360, axis_angle 136-160)
//     bone transform relative to the rest pose  t * rest^-1    (lab4d/nnutils/warping.py:415-425, quat_transform.py:420-469)
//     the rest pose's object->bone rotation matrix and translation, scaled by the Gaussian bones' inverse extents
//                                                              (lab4d/nnutils/skinning.py:117-141; quat_transform.py:221-255)
// as ~75 elementwise torch launches forward and ~175 backward on (M, B, 4) tensors: latency, not work.
// Here one thread carries one (frame, bone) through the chain.  The backward does not restate the chain's adjoint by
// hand: the same templated function is evaluated on first-order dual numbers, one input direction at a time
// (6 M + 9 directions of ~150 flops per bone, one thread each), and the gradient is the contraction of those tangents
// with the incoming gradients -- one source of truth for values and derivatives.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"
#include "bone_tables_math.h"

namespace {

__global__ void bone_tables_fwd_kernel(int M, int B, const float* so3_t, const float* trans_t, const float* so3_r,
                                       const float* trans_r, const float* inv_gauss, float* se3_qr, float* se3_qd,
                                       float* bone_A, float* bone_c)
{
    bone_tables::bone_tables_fwd_body(blockIdx.x * blockDim.x + threadIdx.x, M, B, so3_t, trans_t, so3_r, trans_r, inv_gauss,
                                      se3_qr, se3_qd, bone_A, bone_c);
}

__global__ void bone_tables_bwd_kernel(int M, int B, const float* so3_t, const float* trans_t, const float* so3_r,
                                       const float* trans_r, const float* inv_gauss, const float* g_qr, const float* g_qd,
                                       const float* g_A, const float* g_c, float* g_so3_t, float* g_trans_t,
                                       float* g_so3_r, float* g_trans_r, float* g_inv_gauss)
{
    bone_tables::bone_tables_bwd_body(blockIdx.x * blockDim.x + threadIdx.x, M, B, so3_t, trans_t, so3_r, trans_r, inv_gauss,
                                      g_qr, g_qd, g_A, g_c, g_so3_t, g_trans_t, g_so3_r, g_trans_r, g_inv_gauss);
}

__global__ void camera_tail_fwd_kernel(int M, const float* raw, const float* base, float* out)
{
    bone_tables::camera_tail_fwd_body(blockIdx.x * blockDim.x + threadIdx.x, M, raw, base, out);
}

__global__ void camera_tail_bwd_kernel(int M, const float* raw, const float* base, const float* g_out, float* g_raw,
                                       float* g_base)
{
    bone_tables::camera_tail_bwd_body(blockIdx.x * blockDim.x + threadIdx.x, M, raw, base, g_out, g_raw, g_base);
}

int done()
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

}  // namespace

extern "C" int vidu4d_bone_tables_forward(int M, int B, const float* so3_t, const float* trans_t, const float* so3_rest,
                                          const float* trans_rest, const float* inv_gauss, float* se3_qr, float* se3_qd,
                                          float* bone_A, float* bone_c, void* stream)
{
    if (M < 0 || B < 0) return VIDU4D_E_INVALID;
    if (B == 0) return VIDU4D_OK;
    if (!so3_rest || !trans_rest || (M > 0 && (!so3_t || !trans_t || !se3_qr || !se3_qd))) return VIDU4D_E_INVALID;
    if ((bone_A != nullptr) != (bone_c != nullptr) || (bone_A && !inv_gauss)) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    const int n = (M + 1) * B;
    hipLaunchKernelGGL(bone_tables_fwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, M, B, so3_t, trans_t,
                       so3_rest, trans_rest, inv_gauss, se3_qr, se3_qd, bone_A, bone_c);
    return done();
}

extern "C" int vidu4d_bone_tables_backward(int M, int B, const float* so3_t, const float* trans_t, const float* so3_rest,
                                           const float* trans_rest, const float* inv_gauss, const float* g_se3_qr,
                                           const float* g_se3_qd, const float* g_bone_A, const float* g_bone_c,
                                           float* g_so3_t, float* g_trans_t, float* g_so3_rest, float* g_trans_rest,
                                           float* g_inv_gauss, void* stream)
{
    if (M < 0 || B < 0) return VIDU4D_E_INVALID;
    if (B == 0) return VIDU4D_OK;
    if (!so3_rest || !trans_rest || !g_so3_rest || !g_trans_rest) return VIDU4D_E_INVALID;
    if (M > 0 && (!so3_t || !trans_t || !g_so3_t || !g_trans_t)) return VIDU4D_E_INVALID;
    if (g_bone_A && !inv_gauss) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    const int n = B * bone_tables::bone_tables_bwd_dirs(M);
    hipLaunchKernelGGL(bone_tables_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, M, B, so3_t, trans_t,
                       so3_rest, trans_rest, inv_gauss, g_se3_qr, g_se3_qd, g_bone_A, g_bone_c, g_so3_t, g_trans_t,
                       g_so3_rest, g_trans_rest, g_inv_gauss);
    return done();
}

extern "C" int vidu4d_camera_tail_forward(int M, const float* raw_quat, const float* base_quat, float* cam_q, void* stream)
{
    if (M < 0) return VIDU4D_E_INVALID;
    if (M == 0) return VIDU4D_OK;
    if (!raw_quat || !base_quat || !cam_q) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    hipLaunchKernelGGL(camera_tail_fwd_kernel, dim3((M + 63) / 64), dim3(64), 0, (hipStream_t)stream, M, raw_quat, base_quat,
                       cam_q);
    return done();
}

extern "C" int vidu4d_camera_tail_backward(int M, const float* raw_quat, const float* base_quat, const float* g_cam_q,
                                           float* g_raw_quat, float* g_base_quat, void* stream)
{
    if (M < 0) return VIDU4D_E_INVALID;
    if (M == 0) return VIDU4D_OK;
    if (!raw_quat || !base_quat || !g_cam_q || !g_raw_quat || !g_base_quat) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    hipLaunchKernelGGL(camera_tail_bwd_kernel, dim3((8 * M + 63) / 64), dim3(64), 0, (hipStream_t)stream, M, raw_quat,
                       base_quat, g_cam_q, g_raw_quat, g_base_quat);
    return done();
}
