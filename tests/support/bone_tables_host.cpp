// Host-side driver of vidu4d_amd/csrc/bone_tables_math.h for tests/test_bone_tables_cpu.py: the kernels' per-thread
// bodies in plain loops (test infrastructure; the product launches them from bone_tables.hip).
#include "bone_tables_math.h"

extern "C" void bt_forward(int M, int B, const float* so3_t, const float* trans_t, const float* so3_r, const float* trans_r,
                           const float* inv_gauss, float* se3_qr, float* se3_qd, float* bone_A, float* bone_c)
{
    for (int i = 0; i < (M + 1) * B; ++i)
        bone_tables::bone_tables_fwd_body(i, M, B, so3_t, trans_t, so3_r, trans_r, inv_gauss, se3_qr, se3_qd, bone_A, bone_c);
}

extern "C" void bt_backward(int M, int B, const float* so3_t, const float* trans_t, const float* so3_r, const float* trans_r,
                            const float* inv_gauss, const float* g_qr, const float* g_qd, const float* g_A, const float* g_c,
                            float* g_so3_t, float* g_trans_t, float* g_so3_r, float* g_trans_r, float* g_inv_gauss)
{
    for (int i = 0; i < B * bone_tables::bone_tables_bwd_dirs(M); ++i)
        bone_tables::bone_tables_bwd_body(i, M, B, so3_t, trans_t, so3_r, trans_r, inv_gauss, g_qr, g_qd, g_A, g_c, g_so3_t,
                                          g_trans_t, g_so3_r, g_trans_r, g_inv_gauss);
}

extern "C" void ct_forward(int M, const float* raw, const float* base, float* out)
{
    for (int m = 0; m < M; ++m) bone_tables::camera_tail_fwd_body(m, M, raw, base, out);
}

extern "C" void ct_backward(int M, const float* raw, const float* base, const float* g_out, float* g_raw, float* g_base)
{
    for (int i = 0; i < 8 * M; ++i) bone_tables::camera_tail_bwd_body(i, M, raw, base, g_out, g_raw, g_base);
}
