// contract.hip -- weight gradients of the skinning field's layers as contractions over the surfels, all of them in ONE launch.
//
// With networks that train (--gs_optim_warp=True) the TRAIN instances of skin_field.hip leave every layer's masked
// pre-activation gradients G (O, N) and input activations X (I, N) in feature-major arrays; the weight gradient of a layer is
//     dW[o][i] = sum_n G[o][n] X[i][n]          (the bias gradient is the column of X's constant-1 row)
// a product of two short, very wide matrices (O, I <= 96; N = 200 000 surfels) that the library runs as a handful of
// workgroups down the whole K.  lab4d/bob_warp.contract_over_columns cut K into chunks for a batched GEMM: per contraction a
// bmm (36 us), a sum over the chunks (9), a GEMM for the remainder (13) and an add (5) -- four contractions per step, 250 us
// of a 2.2 ms step.  Here: grid (chunks of K, contraction); a workgroup stages 32 columns of both operands at a time into LDS
// (coalesced 128-byte row pieces, the next stage's loads in flight under the current one's MFMAs), its four waves own up to
// three 32x32 output tiles each on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: an fmaf chain, bit for bit), and the
// partial products leave with one float atomic per output and workgroup.  ~350 MB of operands per step: memory-bound.
// Another summation order than the library's (and not a fixed one: float atomics), 1e-6 relative.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KS = 32;          // columns per stage
constexpr int MAXR = 96;        // rows of either operand (three 32-row blocks)
constexpr int LD = KS + 1;      // LDS row stride: rows 32 apart at one column fall into 32 different banks

#ifndef CONTRACT_BUFS
#define CONTRACT_BUFS 2
#endif
#ifndef CONTRACT_WAVES
#define CONTRACT_WAVES 3
#endif
#ifndef CONTRACT_WG_PER_CU
#define CONTRACT_WG_PER_CU 3
#endif
struct __attribute__((packed, aligned(4))) Quad {
    float x, y, z, w;
};

struct ContractLaunch {
    Vidu4dContractJob j[VIDU4D_CONTRACT_MAX_JOBS];
    int64_t K;
    int64_t k_per_block;   // a multiple of KS
};

// Three workgroups per CU (50 KB of LDS each, <= 168 registers): 54 KB of operand rows in flight per CU, which is what it
// takes to cover the memory latency at this bandwidth (one workgroup's 18 KB stage at a time, four-byte loads: 266 us for the
// step's four contractions, against ~90 for their 350 MB).
// WIDE: every operand row has unit stride along k: thread t fetches columns 4 (t & 7) .. + 3 of the rows (t >> 3) + 32 i, one
// stage ahead, into registers (24 dwords in flight per thread).  Otherwise (an operand strided along k, e.g. an (N, 4) point
// array read as four rows): four-byte loads straight into the other LDS buffer -- correct, not fast.
template <bool WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CONTRACT_WAVES, CONTRACT_WAVES))) void contract_rows_kernel(ContractLaunch a)
{
    __shared__ float s_a[CONTRACT_BUFS][MAXR * LD], s_b[CONTRACT_BUFS][MAXR * LD];
    const Vidu4dContractJob job = a.j[blockIdx.y];
    const int M = job.rows_a, C = job.rows_b;
    const int MT = (M + 31) >> 5, NT = (C + 31) >> 5, tiles = MT * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t k_begin = (int64_t)blockIdx.x * a.k_per_block;
    const int64_t k_end = k_begin + a.k_per_block < a.K ? k_begin + a.k_per_block : a.K;
    if (k_begin >= k_end) return;
    const int rows = (MT + NT) * 32;             // staged rows: A's blocks, then B's

    // WIDE: this thread's (up to six) row pieces: where they start at k = 0 (nullptr: a padding row) and where they go in LDS
    const int col4 = 4 * (threadIdx.x & 7), row4 = threadIdx.x >> 3;
    const float* src[6];
    int dst[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int r = row4 + 32 * i;
        src[i] = nullptr;
        dst[i] = -1;
        if (r < rows) {
            const bool in_a = r < MT * 32;
            const int rr = in_a ? r : r - MT * 32;
            dst[i] = (in_a ? 0 : MAXR * LD) + rr * LD + col4;          // (s_b follows s_a's buffer 0 by 2 * MAXR * LD floats: below)
            if (rr < (in_a ? M : C)) src[i] = (in_a ? job.a + (int64_t)rr * job.lda : job.b + (int64_t)rr * job.ldb) + col4;
        }
    }
    // (every four-column group of every row 16-byte aligned: one 16-byte load per group -- workgroup-uniform)
    const bool aligned16 = ((job.lda | job.ldb | k_begin) & 3) == 0 &&
                           ((reinterpret_cast<uintptr_t>(job.a) | reinterpret_cast<uintptr_t>(job.b)) & 15) == 0;
    float4 pre[6];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t k = k0 + col4;
            if (src[i] && k < k_end) {
                const float* p = src[i] + k0;
                if (k + 3 < k_end) {
                    if (aligned16) v = *reinterpret_cast<const float4*>(p);
                    else {
                        const Quad q = *reinterpret_cast<const Quad*>(p);   // (four-byte aligned: rows of any length)
                        v = make_float4(q.x, q.y, q.z, q.w);
                    }
                }
                else {   // (the last one to three columns of K)
                    v.x = p[0];
                    if (k + 1 < k_end) v.y = p[1];
                    if (k + 2 < k_end) v.z = p[2];
                }
            }
            pre[i] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 6; i++)
            if (dst[i] >= 0) {
                float* d = (dst[i] < MAXR * LD ? &s_a[buf][dst[i]] : &s_b[buf][dst[i] - MAXR * LD]);
                d[0] = pre[i].x, d[1] = pre[i].y, d[2] = pre[i].z, d[3] = pre[i].w;
            }
    };
    // !WIDE: a stage straight into LDS, (row, column) = (t >> 5) + 8 i, t & 31
    auto stage_narrow = [&](int buf, int64_t k0) {
        const int col = threadIdx.x & 31;
        const int64_t k = k0 + col;
#pragma unroll 4
        for (int r = threadIdx.x >> 5; r < rows; r += 8) {
            float v = 0.f;
            const bool in_a = r < MT * 32;
            const int rr = in_a ? r : r - MT * 32;
            if (k < k_end && rr < (in_a ? M : C))
                v = in_a ? job.a[(int64_t)rr * job.lda + k * job.sa] : job.b[(int64_t)rr * job.ldb + k * job.sb];
            (in_a ? s_a[buf] : s_b[buf])[rr * LD + col] = v;
        }
    };

    f32x16 acc[3];   // wave w owns the tiles w, w + 4, w + 8 (half tiles dealt round-robin -- 3, 3, 3, 3 units for the 64 x 76
                     // product instead of 2, 2, 1, 1 tiles -- measured the same: the matrix cores are not what the launch waits for)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[i][v] = 0.f;
    if (WIDE) {
        fetch(k_begin);
        stash(0);
    } else {
        stage_narrow(0, k_begin);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += KS) {
        const bool more = k0 + KS < k_end;
        if (WIDE && more) fetch(k0 + KS);                 // (in flight under the MFMAs below)
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int t = wave + 4 * i;
            if (t < tiles) {                              // (wave-uniform)
                const int tm = t / NT, tn = t - tm * NT;
                const float* pa = &s_a[buf][(tm * 32 + (lane & 31)) * LD + (lane >> 5)];
                const float* pb = &s_b[buf][(tn * 32 + (lane & 31)) * LD + (lane >> 5)];
#pragma unroll 4
                for (int kp = 0; kp < KS / 2; kp++)   // (unrolled all the way the 32 operand reads of a tile are hoisted: spills)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * kp], pb[2 * kp], acc[i], 0, 0, 0);
            }
        }
        if (CONTRACT_BUFS == 1) __syncthreads();          // (every wave is done reading the one buffer)
        if (more) {
            if (WIDE) stash(CONTRACT_BUFS == 1 ? 0 : buf ^ 1);
            else stage_narrow(CONTRACT_BUFS == 1 ? 0 : buf ^ 1, k0 + KS);
        }
        __syncthreads();
        if (CONTRACT_BUFS == 2) buf ^= 1;
    }
    // C/D layout of the 32x32x2 MFMA: register v of lane l holds row (v & 3) + 8 (v >> 2) + 4 (l >> 5), column l & 31
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int t = wave + 4 * i;
        if (t < tiles) {
            const int tm = t / NT, tn = t - tm * NT;
            const int c = tn * 32 + (lane & 31);
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int r = tm * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
                if (r < M && c < C && acc[i][v] != 0.f) atomicAdd(job.out + (int64_t)r * C + c, acc[i][v]);
            }
        }
    }
}

}  // namespace

extern "C" int vidu4d_contract_rows(int n, const Vidu4dContractJob* jobs, int64_t K, void* stream)
{
    if (n < 0 || n > VIDU4D_CONTRACT_MAX_JOBS || (n && !jobs) || K < 0) return VIDU4D_E_INVALID;
    if (n == 0 || K == 0) return VIDU4D_OK;
    ContractLaunch wide, narrow;
    int n_wide = 0, n_narrow = 0;
    for (int i = 0; i < n; i++) {
        const Vidu4dContractJob& j = jobs[i];
        if (j.rows_a <= 0 || j.rows_a > MAXR || j.rows_b <= 0 || j.rows_b > MAXR || !j.a || !j.b || !j.out || j.sa < 1 || j.sb < 1)
            return VIDU4D_E_INVALID;
        const bool w = j.sa == 1 && j.sb == 1;   // (rows contiguous along k: four columns per thread and stage, a stage ahead)
        if (w) wide.j[n_wide++] = j;
        else narrow.j[n_narrow++] = j;
    }
    (void)hipGetLastError();
    const int64_t stages = (K + KS - 1) / KS;
    auto launch = [&](ContractLaunch& a, int count, bool is_wide) {
        if (!count) return;
        // three workgroups per compute unit over the contractions of the launch, at least eight stages each
        const int64_t want = 256 * CONTRACT_WG_PER_CU / count;
        int64_t blocks = stages / 8 < want ? (stages / 8 > 0 ? stages / 8 : 1) : want;
        a.K = K;
        a.k_per_block = ((stages + blocks - 1) / blocks) * KS;
        blocks = (K + a.k_per_block - 1) / a.k_per_block;
        if (is_wide)
            hipLaunchKernelGGL(contract_rows_kernel<true>, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(contract_rows_kernel<false>, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
    };
    launch(wide, n_wide, true);
    launch(narrow, n_narrow, false);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

// ---- several small strided 2-D copies in one launch: the TRAIN skinning field repacks its weights into the kernels' padded
// arrays and the surfel centres into a feature-major array every step -- six copies of 64 bytes .. 2.4 MB, ~6 us each as
// launches of their own (lab4d/lbs_fused._SkinFieldTrain.forward).
namespace {
struct CopyLaunch {
    Vidu4dCopyJob j[VIDU4D_COPY_MAX_JOBS];
    unsigned first_block[VIDU4D_COPY_MAX_JOBS + 1];
    int n;
};
constexpr int COPY_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void copy_strided_kernel(CopyLaunch a)
{
    int k = 0;
#pragma unroll
    for (int i = 1; i < VIDU4D_COPY_MAX_JOBS; i++)
        if (i < a.n && blockIdx.x >= a.first_block[i]) k = i;
    const Vidu4dCopyJob job = a.j[k];
    const int64_t total = (int64_t)job.rows * job.cols;
    const int64_t base = (int64_t)(blockIdx.x - a.first_block[k]) * COPY_PER_BLOCK + threadIdx.x;
#pragma unroll
    for (int i = 0; i < COPY_PER_BLOCK / 256; i++) {
        const int64_t e = base + 256 * i;
        if (e < total) {
            const int64_t r = e / job.cols, c = e - r * job.cols;
            job.dst[r * job.dst_ld + c] = job.src[r * job.src_ld + c * job.src_cs];
        }
    }
}
}  // namespace

extern "C" int vidu4d_copy_strided(int n, const Vidu4dCopyJob* jobs, void* stream)
{
    if (n < 0 || n > VIDU4D_COPY_MAX_JOBS || (n && !jobs)) return VIDU4D_E_INVALID;
    CopyLaunch a;
    a.n = 0;
    unsigned blocks = 0;
    for (int i = 0; i < n; i++) {
        const Vidu4dCopyJob& j = jobs[i];
        if (j.rows < 0 || j.cols < 0 || ((int64_t)j.rows * j.cols && (!j.src || !j.dst))) return VIDU4D_E_INVALID;
        if ((int64_t)j.rows * j.cols == 0) continue;
        a.j[a.n] = j;
        a.first_block[a.n] = blocks;
        blocks += (unsigned)(((int64_t)j.rows * j.cols + COPY_PER_BLOCK - 1) / COPY_PER_BLOCK);
        a.n++;
    }
    a.first_block[a.n] = blocks;
    if (!blocks) return VIDU4D_OK;
    (void)hipGetLastError();
    hipLaunchKernelGGL(copy_strided_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}
