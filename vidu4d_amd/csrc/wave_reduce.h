// wave_reduce.h -- sums of eight per-lane values over the 64 lanes of a wave in 17 instructions (gfx950).
// v_permlane32_swap folds the halves of the wave and halves the values per lane, v_permlane16_swap the row pairs, one
// bank-masked DPP step the half rows, three DPP adds the last eight lanes (a butterfly per value: 48 cross-lane operations,
// and `__shfl_xor` is an LDS-crossbar `ds_bpermute` each).  Afterwards lane L holds the total of value
//     4 (L >> 5) + 2 ((L >> 4) & 1) + ((L >> 3) & 1)
// (the eight lanes of a group agree; callers act on the lanes with (L & 7) == 0).  Used by lbs.hip (bone / camera gradients)
// and dense_stack.hip (a layer's dot products).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void swap_add32(float& a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_add16(float& a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_reduce_scatter8(float (&v)[8])
{
#pragma unroll
    for (int k = 0; k < 4; k++) swap_add32(v[k], v[k + 4]);
    swap_add16(v[0], v[2]);
    swap_add16(v[1], v[3]);
    float t = v[0];
    // lanes 0-7 of a row: v0 + its mirror lane's v0; lanes 8-15: v1 + the mirror lane's v1 (bank-masked DPP writes)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(t)
        : "v"(v[1]));
    return t;
}
