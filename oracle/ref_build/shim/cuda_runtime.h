/* TEST INFRASTRUCTURE (oracle/ref_build): lets the reference's own CUDA sources under /root/reference
 * compile UNMODIFIED with hipcc for gfx950, so that the CPU oracle and the product can be checked
 * against the real reference running on an MI355X.  Never part of the product; the product is
 * written directly for HIP and includes none of this. */
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaMemcpy hipMemcpy
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemset hipMemset
#ifndef __trap
#define __trap() __builtin_trap()
#endif
#ifdef REF_STRICT_RSQRT
/* "strict" variant (build_ref.py): the approximate v_rsq_f32 behind rsqrtf is replaced by the correctly
 * rounded 1/sqrt, so that the build evaluates exactly the fp32 operations the source names. */
#define rsqrtf(x) (1.0f / sqrtf(x))
#endif
