// preprocess.hip -- per-surfel kernels: projection (forward), tile-count scan, key emission,
// per-surfel backward, frustum marking.  gfx950 only.
//
// Reference behaviour restated here (file:line in /root/reference/gs/submodules/
// diff-surfel-rasterization/cuda_rasterizer/): preprocessCUDA forward.cu:166-260; InclusiveSum
// rasterizer_impl.cu:278; duplicateWithKeys rasterizer_impl.cu:70-111; computeAABB +
// preprocessCUDA backward backward.cu:599-649, :533-597; checkFrustum rasterizer_impl.cu:54-66.
//
// MI355X notes: one thread per surfel, 256-thread workgroups (4 wave64).  The tile-count prefix sum
// is split so that it costs one tiny extra launch: the projection kernel reduces its workgroup's
// counts (wave64 shuffle + LDS), one single-workgroup kernel scans the <= 4k workgroup sums, and the
// emit kernel redoes the in-workgroup scan on the fly instead of reading back a materialised
// offsets array.
#include "surfel_state.h"

namespace surfel {

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// Exclusive scan over a 256-thread workgroup; returns this thread's exclusive prefix and the total.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t t = s_wave[w];
        if (w < wave) base += t;
    }
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return base + inc - v;
}

__global__ __launch_bounds__(PRE_BLOCK) void preprocess_fwd_kernel(PreprocessArgs a)
{
    __shared__ uint32_t s_wave[4];
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const Camera cam = load_camera(a.cam);
    uint32_t tiles = 0;
    if (idx < a.P) {
        const float p_world[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
        const float2 sc = reinterpret_cast<const float2*>(a.scales)[idx];
        const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[idx];
        const float scale[2] = {sc.x, sc.y};
        const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
        Projected o;
        int radius = 0;
        if (project_surfel(cam, p_world, quat, scale, o)) {
            tiles = o.tiles;
            radius = o.radius;
            float rgb[3];
            uint32_t clamp_mask = 0;
            if (a.colors_precomp == nullptr) {
                sh_forward(cam.sh_degree, p_world, cam.campos, a.shs + (size_t)idx * cam.sh_coeffs * 3, rgb,
                           clamp_mask);
            } else {
                rgb[0] = a.colors_precomp[3 * idx];
                rgb[1] = a.colors_precomp[3 * idx + 1];
                rgb[2] = a.colors_precomp[3 * idx + 2];
            }
            float4* rec = reinterpret_cast<float4*>(a.geom.rec + (size_t)idx * REC_FLOATS);
            rec[0] = make_float4(o.T[0], o.T[1], o.T[2], o.T[3]);
            rec[1] = make_float4(o.T[4], o.T[5], o.T[6], o.T[7]);
            rec[2] = make_float4(o.T[8], o.center[0], o.center[1], a.opacities[idx]);
            rec[3] = make_float4(o.normal[0], o.normal[1], o.normal[2], o.depth);
            rec[4] = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float(clamp_mask));
            float box[4];
            contribution_box(o.T, o.center[0], o.center[1], a.opacities[idx], box);
            rec[5] = make_float4(box[0], box[1], box[2], box[3]);
        }
        a.radii[idx] = radius;
        a.geom.tiles_touched[idx] = tiles;
    }
    uint32_t total;
    block_exclusive_scan(tiles, s_wave, total);
    if (threadIdx.x == 0) a.geom.block_sums[blockIdx.x] = total;
}

void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t stream)
{
    if (a.P <= 0) return;
    hipLaunchKernelGGL(preprocess_fwd_kernel, dim3(pre_blocks(a.P)), dim3(PRE_BLOCK), 0, stream, a);
}

// One workgroup: exclusive scan of the per-workgroup tile counts, total -> header, zero the tile
// ranges (the reference's cudaMemset, rasterizer_impl.cu:311).
__global__ __launch_bounds__(1024) void scan_blocks_kernel(GeomState g, int nblocks, uint32_t* ranges, int num_tiles)
{
    __shared__ uint32_t s_part[16];
    __shared__ uint32_t s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? g.block_sums[i] : 0;
        const uint32_t inc = wave_inclusive_scan(v, lane);
        if (lane == 63) s_part[wave] = inc;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int w = 0; w < 16; w++) {
            const uint32_t t = s_part[w];
            if (w < wave) wbase += t;
            tot += t;
        }
        const uint32_t carry = s_carry;
        if (i < nblocks) g.block_offsets[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        g.hdr->num_rendered = s_carry;
        g.hdr->overflow = 0;
    }
    for (int i = threadIdx.x; i < 2 * num_tiles; i += 1024) ranges[i] = 0;
}

void launch_scan_blocks(const GeomState& g, int P, uint32_t* ranges, int num_tiles, hipStream_t stream)
{
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, g, pre_blocks(P), ranges, num_tiles);
}

// (tile | depth) key / surfel-id value pairs in the reference's emission order: surfel id ascending,
// tiles row-major inside the rect (rasterizer_impl.cu:98-109).  The stable sort relies on it.
__global__ __launch_bounds__(PRE_BLOCK) void emit_keys_kernel(CameraParams cam, int P, const int32_t* radii, GeomState g,
                                                             uint64_t* keys, uint32_t* vals, int64_t capacity)
{
    __shared__ uint32_t s_wave[4];
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const uint32_t tiles = idx < P ? g.tiles_touched[idx] : 0;
    uint32_t total;
    uint32_t off = block_exclusive_scan(tiles, s_wave, total) + g.block_offsets[blockIdx.x];
    if ((int64_t)g.hdr->num_rendered > capacity) {  // binning buffer too small: render nothing, flag it
        if (idx == 0) g.hdr->overflow = 1;
        return;
    }
    if (tiles == 0) return;
    const float4 q2 = reinterpret_cast<const float4*>(g.rec + (size_t)idx * REC_FLOATS)[2];
    const float4 q3 = reinterpret_cast<const float4*>(g.rec + (size_t)idx * REC_FLOATS)[3];
    int x0, y0, x1, y1;
    tile_rect(q2.y, q2.z, radii[idx], cam.grid_x, cam.grid_y, x0, y0, x1, y1);
    const uint64_t dbits = (uint64_t)__float_as_uint(q3.w);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const uint64_t key = ((uint64_t)(uint32_t)(y * cam.grid_x + x) << 32) | dbits;
            keys[off] = key;
            vals[off] = (uint32_t)idx;
            off++;
        }
}

void launch_emit_keys(const CameraParams& cam, int P, const int32_t* radii, const GeomState& g, const BinState& b,
                      int64_t capacity, hipStream_t stream)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(emit_keys_kernel, dim3(pre_blocks(P)), dim3(PRE_BLOCK), 0, stream, cam, P, radii, g,
                       b.keys[0], b.vals[0], capacity);
}

// Per-surfel backward.  Reads the accumulator filled by the backward blend and writes every output
// gradient (zeros for culled surfels, which the reference gets from torch::zeros).
__global__ __launch_bounds__(PRE_BLOCK) void preprocess_bwd_kernel(BackwardArgs a)
{
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    if (idx >= a.P) return;
    const Camera cam = load_camera(a.cam);
    const int M = cam.sh_coeffs;
    float* dsh = a.dL_dsh ? a.dL_dsh + (size_t)idx * M * 3 : nullptr;
    if (!(a.radii[idx] > 0)) {
        for (int k = 0; k < 3; k++) {
            a.dL_dmeans3D[3 * idx + k] = 0.f;
            a.dL_dmeans2D[3 * idx + k] = 0.f;
            a.dL_dcolors[3 * idx + k] = 0.f;
        }
        a.dL_dopacity[idx] = 0.f;
        for (int k = 0; k < 9; k++) a.dL_dtransMat[9 * idx + k] = 0.f;
        a.dL_dscales[2 * idx] = a.dL_dscales[2 * idx + 1] = 0.f;
        for (int k = 0; k < 4; k++) a.dL_drotations[4 * idx + k] = 0.f;
        if (dsh)
            for (int k = 0; k < 3 * M; k++) dsh[k] = 0.f;
        return;
    }
    float acc[ACC_FLOATS];
    {
        const float4* p = reinterpret_cast<const float4*>(a.acc + (size_t)idx * ACC_FLOATS);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float4 v = p[k];
            acc[4 * k] = v.x;
            acc[4 * k + 1] = v.y;
            acc[4 * k + 2] = v.z;
            acc[4 * k + 3] = v.w;
        }
    }
    float T[9];
    uint32_t clamp_mask;
    {
        const float4* r = reinterpret_cast<const float4*>(a.geom.rec + (size_t)idx * REC_FLOATS);
        const float4 q0 = r[0], q1 = r[1], q2 = r[2], q4 = r[4];
        T[0] = q0.x; T[1] = q0.y; T[2] = q0.z; T[3] = q0.w;
        T[4] = q1.x; T[5] = q1.y; T[6] = q1.z; T[7] = q1.w;
        T[8] = q2.x;
        clamp_mask = __float_as_uint(q4.w);
    }
    const float p_world[3] = {a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]};
    const float2 sc = reinterpret_cast<const float2*>(a.scales)[idx];
    const float4 q4 = reinterpret_cast<const float4*>(a.rotations)[idx];
    const float scale[2] = {sc.x, sc.y};
    const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
    SurfelGrads o;
    surfel_backward(cam, p_world, quat, scale, T, acc, o);

    float dmean[3] = {o.dmean3D[0], o.dmean3D[1], o.dmean3D[2]};
    const float dcol[3] = {acc[A_RGB], acc[A_RGB + 1], acc[A_RGB + 2]};
    if (a.shs != nullptr)
        sh_backward(cam.sh_degree, M, p_world, cam.campos, a.shs + (size_t)idx * M * 3, clamp_mask, dcol, dsh, dmean);
    for (int k = 0; k < 3; k++) {
        a.dL_dmeans3D[3 * idx + k] = dmean[k];
        a.dL_dmeans2D[3 * idx + k] = o.dmean2D[k];
        a.dL_dcolors[3 * idx + k] = dcol[k];
    }
    a.dL_dopacity[idx] = acc[A_OPAC];
    for (int k = 0; k < 9; k++) a.dL_dtransMat[9 * idx + k] = o.dT[k];
    a.dL_dscales[2 * idx] = o.dscale[0];
    a.dL_dscales[2 * idx + 1] = o.dscale[1];
    for (int k = 0; k < 4; k++) a.dL_drotations[4 * idx + k] = o.drot[k];
}

void launch_preprocess_bwd(const BackwardArgs& a, hipStream_t stream)
{
    if (a.P <= 0) return;
    hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(pre_blocks(a.P)), dim3(PRE_BLOCK), 0, stream, a);
}

__global__ void mark_visible_kernel(int P, const float* means3D, const float* vm, uint8_t* present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float p[3] = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
    float v[3];
    float view[16];
    for (int k = 0; k < 16; k++) view[k] = vm[k];
    to_view(view, p, v);
    present[idx] = !(v[2] <= 0.2f);
}

void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, means3D, viewmatrix,
                       present);
}

}  // namespace surfel
