// dense_stack.hip -- a time-conditioned network's dense layers on a HANDFUL of rows, one launch per direction.
//
// The articulation and camera networks of the bob warp (TimeMLP: five 256-wide layers + a final one, then two heads
// 256 -> 128 -> out; /root/reference/lab4d/nnutils/time.py:11-133, pose.py:29-150 / :153-323, base.py:8-157) are evaluated
// per fitting step on the step's frames (2-3 rows).  As library calls that is 10 addmm + 8 relu forward and 40 launches
// backward per network, each ~6 us of latency on ONE compute unit's worth of work and ~3 us of host time even inside a
// captured graph: 0.35 ms of device time and 0.2 ms of host time per network and step, for 1.7 MB of weights.
// Here a single workgroup of 16 waves walks the whole stack:
//   forward   a wave owns an output feature: lanes stride the input (coalesced 256-byte row reads of W), the rows'
//             activations sit in registers, one butterfly reduction per row; every layer's output goes to `acts`
//             (the backward's only saved state; the workgroup reads it back through its own L1)
//   backward  what is sequential -- g_pre = g * relu' * scale and dX[r][i] = sum_o g_pre[r][o] W[o][i], layer after layer,
//             last to first -- is one workgroup's chain (the output range cut into 1024 / in parts that add into the next
//             LDS buffer: ds_add_f32, 4-8 adders per address); it leaves every layer's g_pre in a workspace, from which a
//             second launch takes dW[o][i] = sum_r g_pre[r][o] h[r][i] and db on a workgroup per 16 outputs of a layer.
// Rows <= 16, widths <= 256.  Float32 throughout; sums in another order than the library's (1e-6 relative).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vidu4d_surfel.h"
#include "wave_reduce.h"

namespace {

constexpr int THREADS = 1024;
constexpr int WAVES = THREADS / 64;
constexpr int MAXW = VIDU4D_DENSE_STACK_MAX_WIDTH;   // 256
constexpr int MAXL = VIDU4D_DENSE_STACK_MAX_LAYERS;  // 16

struct Layout {
    int acts_off[MAXL];   // offset (floats) of layer l's output in `acts`
    int src[MAXL];        // layer whose output is layer l's input, -1 = x
};

__device__ __forceinline__ Layout layout_of(const Vidu4dDenseStack& s)
{
    Layout L;
    int off = 0;
    const int n = s.n_trunk + s.n_head_a + s.n_head_b;
    for (int l = 0; l < MAXL; ++l) {
        L.acts_off[l] = off;
        if (l < n) off += s.rows * s.out[l];
        int src = l - 1;
        if (l == s.n_trunk || l == s.n_trunk + s.n_head_a) src = s.n_trunk - 1;   // a head starts from the trunk's output
        L.src[l] = src;
    }
    return L;
}

template <int RT>
__global__ __launch_bounds__(THREADS) void dense_stack_fwd_kernel(Vidu4dDenseStack s, const float* __restrict__ x,
                                                                  float* __restrict__ acts)
{
    const Layout L = layout_of(s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int R = s.rows, n = s.n_trunk + s.n_head_a + s.n_head_b;
    for (int l = 0; l < n; ++l) {
        const int in = s.in[l], out = s.out[l];
        const float* __restrict__ W = s.W[l];
        const float* __restrict__ b = s.b[l];
        const float* h_in = L.src[l] < 0 ? x : acts + L.acts_off[L.src[l]];
        float* h_out = acts + L.acts_off[l];
        float h[RT][MAXW / 64];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int j = 0; j < MAXW / 64; ++j) {
                const int i = lane + 64 * j;
                h[r][j] = (r < R && i < in) ? h_in[r * in + i] : 0.f;
            }
        // Eight dot products at a time: U outputs' weight rows are fetched back to back (a single workgroup has no other way
        // to keep loads in flight), and eight per-lane partial sums -- eight outputs of one row (RT = 4) or eight rows of one
        // output (RT = 16) -- are reduced over the wave together (wave_reduce.h).  (One row per trip and a `__shfl_xor`
        // butterfly per value measured 200 us for the ten layers: load latency, then the LDS crossbar.)
        constexpr int U = RT <= 4 ? 8 : 2;
        const int owner = 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1);
        for (int o0 = wave; o0 < out; o0 += WAVES * U) {
            float w[U][MAXW / 64];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int o = o0 + u * WAVES;
#pragma unroll
                for (int j = 0; j < MAXW / 64; ++j) {
                    const int i = lane + 64 * j;
                    w[u][j] = (o < out && i < in) ? W[(int64_t)o * in + i] : 0.f;
                }
            }
            if (RT <= 4) {
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        float a = 0.f;
#pragma unroll
                        for (int j = 0; j < MAXW / 64; ++j) a = fmaf(w[u % U][j], h[r][j], a);
                        v[u] = a;
                    }
                    const float tot = wave_reduce_scatter8(v);
                    const int o = o0 + owner * WAVES;
                    if ((lane & 7) == 0 && r < R && o < out) {
                        float y = tot + (b ? b[o] : 0.f);
                        if (s.relu[l]) y = fmaxf(y, 0.f);
                        h_out[r * out + o] = y * s.scale[l];
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int o = o0 + u * WAVES;
#pragma unroll
                    for (int rb = 0; rb < RT / 8; ++rb) {
                        float v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            float a = 0.f;
#pragma unroll
                            for (int j = 0; j < MAXW / 64; ++j) a = fmaf(w[u][j], h[(8 * rb + k) % RT][j], a);
                            v[k] = a;
                        }
                        const float tot = wave_reduce_scatter8(v);
                        const int r = 8 * rb + owner;
                        if ((lane & 7) == 0 && r < R && o < out) {
                            float y = tot + (b ? b[o] : 0.f);
                            if (s.relu[l]) y = fmaxf(y, 0.f);
                            h_out[r * out + o] = y * s.scale[l];
                        }
                    }
                }
            }
        }
        __syncthreads();   // (workgroup-scope fence + barrier: the next layer reads h_out through this CU's L1)
    }
}

// One link of the backward chain: g (R x out, LDS: gradient w.r.t. the layer's output) becomes g_pre = g * relu' * scale
// (kept in LDS and written to `gpre` for the weight-gradient kernel), and the gradient w.r.t. the layer's input is ADDED into
// g_next (LDS, R x in; zeroed or holding the other head's share) unless g_next == nullptr.  The chain is all that is
// sequential in the backward: it reads the weights once; dW and db are the other kernel's, on as many workgroups as it likes.
template <int RT>
__device__ __forceinline__ void chain_link(const Vidu4dDenseStack& s, int l, int R, float* g, float* g_next,
                                           const float* __restrict__ h_out, float* __restrict__ gpre)
{
    const int in = s.in[l], out = s.out[l], t = threadIdx.x;
    const float scale = s.scale[l];
    for (int e = t; e < R * out; e += THREADS) {
        float v = g[e] * scale;
        if (s.relu[l] && !(h_out[e] > 0.f)) v = 0.f;   // (scale is 1 on the layers that carry a relu)
        g[e] = v;
        gpre[e] = v;
    }
    __syncthreads();
    if (g_next) {
        const float* __restrict__ W = s.W[l];
        if (THREADS % in == 0) {
            // thread: input column i, a contiguous range of outputs; U weight loads in flight (a single workgroup has no
            // other way to hide their latency), the rows' g_pre as 16-byte LDS reads where the range allows
            const int i = t % in, parts = THREADS / in, per = ((out + parts - 1) / parts + 3) & ~3;
            const int lo = (t / in) * per, hi = min(out, lo + per);
            float acc[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = 0.f;
            constexpr int U = 16;
            for (int ob = lo; ob < hi; ob += U) {
                float w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = ob + u < hi ? W[(int64_t)(ob + u) * in + i] : 0.f;
                if ((out & 3) == 0 && ob + U <= hi) {
#pragma unroll
                    for (int r = 0; r < RT; ++r)
                        if (r < R) {
#pragma unroll
                            for (int u = 0; u < U; u += 4) {
                                const float4 gv = *reinterpret_cast<const float4*>(&g[r * out + ob + u]);
                                acc[r] = fmaf(gv.x, w[u], acc[r]);
                                acc[r] = fmaf(gv.y, w[u + 1], acc[r]);
                                acc[r] = fmaf(gv.z, w[u + 2], acc[r]);
                                acc[r] = fmaf(gv.w, w[u + 3], acc[r]);
                            }
                        }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (ob + u < hi) {
#pragma unroll
                            for (int r = 0; r < RT; ++r)
                                if (r < R) acc[r] = fmaf(g[r * out + ob + u], w[u], acc[r]);
                        }
                }
            }
            if (lo < hi) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    if (r < R) atomicAdd(&g_next[r * in + i], acc[r]);
            }
        } else {
            for (int e = t; e < R * in; e += THREADS) {
                const int r = e / in, i = e - r * in;
                float a = 0.f;
                for (int o = 0; o < out; ++o) a = fmaf(g[r * out + o], W[(int64_t)o * in + i], a);
                atomicAdd(&g_next[e], a);
            }
        }
    }
    __syncthreads();
}

template <int RT>
__global__ __launch_bounds__(THREADS) void dense_stack_chain_kernel(Vidu4dDenseStack s, const float* __restrict__ acts,
                                                                    const float* __restrict__ g_out_a,
                                                                    const float* __restrict__ g_out_b,
                                                                    float* __restrict__ gpre, float* __restrict__ g_x)
{
    __shared__ __align__(16) float buf[3][RT * MAXW];
    const Layout L = layout_of(s);
    const int R = s.rows, t = threadIdx.x;
    float* feat = buf[2];
    const bool headless = s.n_head_a == 0 && s.n_head_b == 0;   // (then g_out_a is the gradient of the trunk's output)
    for (int e = t; e < RT * MAXW; e += THREADS)
        feat[e] = (headless && g_out_a && e < R * s.out[s.n_trunk - 1]) ? g_out_a[e] : 0.f;
    __syncthreads();
    // ---- the heads, each from its last layer down to its first, whose input gradient adds into `feat`
    for (int head = 0; head < 2; ++head) {
        const int first = head == 0 ? s.n_trunk : s.n_trunk + s.n_head_a;
        const int count = head == 0 ? s.n_head_a : s.n_head_b;
        if (count == 0) continue;
        const float* g_out = head == 0 ? g_out_a : g_out_b;
        int cur = 0;
        const int last = first + count - 1;
        for (int e = t; e < RT * MAXW; e += THREADS) buf[0][e] = (g_out && e < R * s.out[last]) ? g_out[e] : 0.f;
        __syncthreads();
        for (int l = last; l >= first; --l) {
            float* next = l == first ? feat : buf[cur ^ 1];
            if (l != first) {
                for (int e = t; e < RT * MAXW; e += THREADS) next[e] = 0.f;
                __syncthreads();
            }
            chain_link<RT>(s, l, R, buf[cur], next, acts + L.acts_off[l], gpre + L.acts_off[l]);
            cur ^= 1;
        }
    }
    // ---- the trunk; its first layer's input gradient is the stack's (g_x, optional)
    float* g = feat;
    int spare = 0;
    for (int l = s.n_trunk - 1; l >= 0; --l) {
        float* next = (l > 0 || g_x) ? buf[spare] : nullptr;
        if (next) {
            for (int e = t; e < RT * MAXW; e += THREADS) next[e] = 0.f;
            __syncthreads();
        }
        chain_link<RT>(s, l, R, g, next, acts + L.acts_off[l], gpre + L.acts_off[l]);
        g = next;
        spare ^= 1;
    }
    if (g_x && g)
        for (int e = t; e < R * s.in[0]; e += THREADS) g_x[e] = g[e];
}

// dW[o][i] = sum_r g_pre[r][o] h_in[r][i], db[o] = sum_r g_pre[r][o]: a workgroup per (layer, DW_CHUNK outputs), a thread per
// input column (coalesced row writes; g_pre's address is uniform over the wave: scalar loads).
constexpr int DW_CHUNK = 16;
__host__ __device__ inline int dw_chunks(const Vidu4dDenseStack& s, int l) { return (s.out[l] + DW_CHUNK - 1) / DW_CHUNK; }

template <int RT>
__global__ __launch_bounds__(MAXW) void dense_stack_dw_kernel(Vidu4dDenseStack s, const float* __restrict__ x,
                                                              const float* __restrict__ acts, const float* __restrict__ gpre)
{
    const Layout L = layout_of(s);
    const int n = s.n_trunk + s.n_head_a + s.n_head_b, R = s.rows;
    int l = 0, c = blockIdx.x;
    while (l < n && c >= dw_chunks(s, l)) c -= dw_chunks(s, l++);
    if (l >= n) return;
    const int in = s.in[l], out = s.out[l], i = threadIdx.x;
    const float* __restrict__ h_in = L.src[l] < 0 ? x : acts + L.acts_off[L.src[l]];
    const float* __restrict__ gp = gpre + L.acts_off[l];
    const int o_lo = c * DW_CHUNK, o_hi = min(out, o_lo + DW_CHUNK);
    if (s.gb[l] && i < o_hi - o_lo) {
        float a = 0.f;
        for (int r = 0; r < R; ++r) a += gp[r * out + o_lo + i];
        s.gb[l][o_lo + i] = a;
    }
    if (!s.gW[l] || i >= in) return;
    float hv[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) hv[r] = r < R ? h_in[r * in + i] : 0.f;
    for (int o = o_lo; o < o_hi; ++o) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < RT; ++r)
            if (r < R) a = fmaf(gp[r * out + o], hv[r], a);
        s.gW[l][(int64_t)o * in + i] = a;
    }
}

int check(const Vidu4dDenseStack* s)
{
    if (!s || s->rows < 1 || s->rows > VIDU4D_DENSE_STACK_MAX_ROWS || s->n_trunk < 1 || s->n_head_a < 0 || s->n_head_b < 0)
        return VIDU4D_E_INVALID;
    const int n = s->n_trunk + s->n_head_a + s->n_head_b;
    if (n > MAXL) return VIDU4D_E_INVALID;
    for (int l = 0; l < n; ++l) {
        if (!s->W[l] || s->in[l] < 1 || s->out[l] < 1 || s->in[l] > MAXW || s->out[l] > MAXW) return VIDU4D_E_INVALID;
        const int src = (l == s->n_trunk || l == s->n_trunk + s->n_head_a) ? s->n_trunk - 1 : l - 1;
        if (src >= 0 && s->in[l] != s->out[src]) return VIDU4D_E_INVALID;
    }
    return VIDU4D_OK;
}

int done()
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? VIDU4D_OK : VIDU4D_E_HIP;
}

}  // namespace

extern "C" int vidu4d_dense_stack_acts_floats(const Vidu4dDenseStack* s)
{
    if (check(s) != VIDU4D_OK) return -1;
    int n = 0;
    for (int l = 0; l < s->n_trunk + s->n_head_a + s->n_head_b; ++l) n += s->rows * s->out[l];
    return n;
}

extern "C" int vidu4d_dense_stack_forward(const Vidu4dDenseStack* s, const float* x, float* acts, void* stream)
{
    const int rc = check(s);
    if (rc != VIDU4D_OK) return rc;
    if (!x || !acts) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    if (s->rows <= 4)
        hipLaunchKernelGGL(dense_stack_fwd_kernel<4>, dim3(1), dim3(THREADS), 0, (hipStream_t)stream, *s, x, acts);
    else
        hipLaunchKernelGGL(dense_stack_fwd_kernel<16>, dim3(1), dim3(THREADS), 0, (hipStream_t)stream, *s, x, acts);
    return done();
}

extern "C" int vidu4d_dense_stack_backward(const Vidu4dDenseStack* s, const float* x, const float* acts,
                                           const float* g_out_a, const float* g_out_b, float* workspace, float* g_x,
                                           void* stream)
{
    const int rc = check(s);
    if (rc != VIDU4D_OK) return rc;
    if (!x || !acts || !workspace) return VIDU4D_E_INVALID;
    (void)hipGetLastError();
    int chunks = 0;
    for (int l = 0; l < s->n_trunk + s->n_head_a + s->n_head_b; ++l) chunks += dw_chunks(*s, l);
    if (s->rows <= 4) {
        hipLaunchKernelGGL(dense_stack_chain_kernel<4>, dim3(1), dim3(THREADS), 0, (hipStream_t)stream, *s, acts, g_out_a,
                           g_out_b, workspace, g_x);
        hipLaunchKernelGGL(dense_stack_dw_kernel<4>, dim3(chunks), dim3(MAXW), 0, (hipStream_t)stream, *s, x, acts, workspace);
    } else {
        hipLaunchKernelGGL(dense_stack_chain_kernel<16>, dim3(1), dim3(THREADS), 0, (hipStream_t)stream, *s, acts, g_out_a,
                           g_out_b, workspace, g_x);
        hipLaunchKernelGGL(dense_stack_dw_kernel<16>, dim3(chunks), dim3(MAXW), 0, (hipStream_t)stream, *s, x, acts, workspace);
    }
    return done();
}
