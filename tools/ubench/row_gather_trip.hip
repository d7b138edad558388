// Microbenchmark (tools/ubench): what would ONE TRIP of a "row-gather" blend backward cost against a
// trip of the shipped half walk?  Not a rasterizer: two loops with the same arithmetic payload (160 dependent-ish FMAs per
// trip standing in for pair evaluation + recurrences + gradient products) and the two ACCUMULATION / STATE models around it:
//
//   pinned (the product, csrc/blend.hip blend_bwd_kernel): pixels pinned to lanes, per-pixel state in registers; per trip
//     3 ds_read_b128 of the entry's record (two distinct addresses: one entry per 32-lane half), the payload, the half
//     reduce-scatter of 16 values (8 v_permlane16_swap + adds, 12 bank-masked DPP adds, 2 selects, 2 DPP adds), one
//     ds_add_f32 with 32 active lanes + the second small one.  ~21.5 KB of LDS per workgroup, 6 workgroups per CU.
//   row-gather (DESIGN.md "Next", the decomposition the verdict asked to be built): per-pixel state in LDS (12 floats per
//     pixel, 48-byte rows), each 16-lane row of a wave takes its OWN list entry and gathers the <= 16 pixels of that entry's
//     footprint wherever they lie in the wave's 8x8 quadrant: per trip 3 ds_read_b128 of the record at FOUR distinct
//     addresses, per-lane pixel address arithmetic, 3 ds_read_b128 + 3 ds_write_b128 of pixel state at per-lane addresses
//     (a 4x4 block at a pseudo-random offset: the bank conflicts a footprint has), the payload, a 16-lane reduce-scatter of
//     16 values (15 DPP adds + selects, no lane swap), one ds_add_f32 with 64 active lanes.  48 KB of pixel state + the
//     batch per workgroup: 3 workgroups per CU.
// Pencilled for the row-gather walk: 0.64 x the trips of the half walk -- it pays if its trip costs less than 1.56 x.
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int TRIPS = 4096;
constexpr int PAYLOAD_ROUNDS = 20;   // x 8 independent chains = 160 v_fma per trip

template <int CTRL>
__device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void swap_add16(float& a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ void payload(float (&acc)[8], float a, float b)
{
#pragma unroll
    for (int r = 0; r < PAYLOAD_ROUNDS; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
}

// ---- the product's trip
__global__ __launch_bounds__(256) void k_pinned(float* out, int trips)
{
    __shared__ float4 s_rec[129 * 5];
    __shared__ float s_acc[128 * 20];
    for (int i = threadIdx.x; i < 129 * 5; i += 256) s_rec[i] = make_float4(0.001f * i, 0.5f, 0.25f, 1.0f);
    for (int i = threadIdx.x; i < 128 * 20; i += 256) s_acc[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc[8];
    for (int i = 0; i < 8; i++) acc[i] = 1.0f + 0.01f * (lane + i);
    unsigned j = threadIdx.x >> 6;
    for (int t = 0; t < trips; t++) {
        j = (j * 5u + 3u) & 127u;                               // (the walk's next entries: one per half)
        const unsigned ja = j, jb = (j + 17u) & 127u;
        const int off = (lane < 32 ? (int)ja : (int)jb) * 5;
        const float4 q0 = s_rec[off], q1 = s_rec[off + 1], q2 = s_rec[off + 2];
        payload(acc, q0.x + q1.y, q2.z);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = acc[i & 7] * (1.0f + 0.125f * i);
#pragma unroll
        for (int k = 0; k < 8; k++) swap_add16(v[k], v[k + 8]);
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %1, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %2, %6, %6 row_mirror row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %3, %7, %7 row_mirror row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
            "s_nop 1"
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])
            : "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        const bool hi = (lane & 2) != 0;
        float r = (hi ? v[1] : v[0]) + dpp<0x4E>(hi ? v[0] : v[1]);
        r += dpp<0xB1>(r);
        if ((lane & 1) == 0) atomicAdd(&s_acc[(lane < 32 ? ja : jb) * 20 + ((lane >> 1) & 15)], r);
        acc[0] += r * 1e-9f;
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + s_acc[threadIdx.x];
}

// ---- a row-gather trip
constexpr int STATE_FLOATS = 12;   // T, 3 colour, alpha, 3 normal, depth, dL/dT chain, 2 distortion: three float4 per pixel
__global__ __launch_bounds__(256) void k_row_gather(float* out, int trips)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* s_rec = reinterpret_cast<float4*>(smem);                         // 129 * 5 float4
    float* s_acc = reinterpret_cast<float*>(smem + 129 * 80);                // 128 * 20 floats
    float4* s_state = reinterpret_cast<float4*>(smem + 129 * 80 + 128 * 80); // 256 pixels * 3 float4
    for (int i = threadIdx.x; i < 129 * 5; i += 256) s_rec[i] = make_float4(0.001f * i, 0.5f, 0.25f, 1.0f);
    for (int i = threadIdx.x; i < 128 * 20; i += 256) s_acc[i] = 0.f;
    for (int i = threadIdx.x; i < 256 * 3; i += 256) s_state[i] = make_float4(1.0f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, l16 = lane & 15;
    float acc[8];
    for (int i = 0; i < 8; i++) acc[i] = 1.0f + 0.01f * (lane + i);
    unsigned j = wave;
    for (int t = 0; t < trips; t++) {
        j = (j * 5u + 3u) & 127u;
        // four entries per trip, one per 16-lane row (disjoint footprints: the rows' pixel sets do not overlap)
        const unsigned je = (j + 17u * (unsigned)row) & 127u;
        const float4 q0 = s_rec[je * 5], q1 = s_rec[je * 5 + 1], q2 = s_rec[je * 5 + 2];
        // the entry's footprint inside the wave's 8x8 quadrant: a 4x4 block at an offset that depends on the entry
        const unsigned fx = (je * 7u + (unsigned)row * 3u) & 3u, fy = ((je >> 2) + (unsigned)row) & 1u;
        const unsigned px = fx + (l16 & 3u), py = 4u * fy + (l16 >> 2) ;       // (rows take disjoint y bands here)
        const unsigned pix = (unsigned)wave * 64u + ((py * 8u + px) & 63u);
        float4 s0 = s_state[pix * 3], s1 = s_state[pix * 3 + 1], s2 = s_state[pix * 3 + 2];
        payload(acc, q0.x + q1.y + s0.x, q2.z + s1.y);
        s0.x = s0.x * 0.999f + acc[0] * 1e-9f;
        s1.y += acc[1] * 1e-9f;
        s2.z += acc[2] * 1e-9f;
        s_state[pix * 3] = s0;
        s_state[pix * 3 + 1] = s1;
        s_state[pix * 3 + 2] = s2;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = acc[i & 7] * (1.0f + 0.125f * i);
        // reduce-scatter of 16 values over the 16 lanes of a row: four select-and-add DPP halvings (8 + 4 + 2 + 1 adds)
        {
            const bool h8 = (lane & 8) != 0, h4 = (lane & 4) != 0, h2 = (lane & 2) != 0, h1 = (lane & 1) != 0;
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = (h8 ? v[i + 8] : v[i]) + dpp<0x140>(h8 ? v[i] : v[i + 8]);
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = (h4 ? v[i + 4] : v[i]) + dpp<0x141>(h4 ? v[i] : v[i + 4]);
#pragma unroll
            for (int i = 0; i < 2; i++) v[i] = (h2 ? v[i + 2] : v[i]) + dpp<0x4E>(h2 ? v[i] : v[i + 2]);
            v[0] = (h1 ? v[1] : v[0]) + dpp<0xB1>(h1 ? v[0] : v[1]);
        }
        atomicAdd(&s_acc[je * 20 + l16], v[0]);                  // 64 active lanes: 4 entries x 16 values
        acc[0] += v[0] * 1e-9f;
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + s_acc[threadIdx.x] + s_state[threadIdx.x].x;
}

template <typename K>
static double run(K kern, float* out, int blocks, size_t lds)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, TRIPS);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, TRIPS);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 4096 * 256 * 4);
    const size_t lds_rg = 129 * 80 + 128 * 80 + 256 * 48;   // record batch + accumulators + pixel state = 32.8 KB
    (void)hipFuncSetAttribute((const void*)k_row_gather, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    printf("one trip = 160 v_fma + the accumulation / state traffic of the model; chip-wide wave trips per microsecond\n");
    for (int per_cu : {1, 2, 3, 4, 6}) {
        const int blocks = 256 * per_cu;
        const double a = run(k_pinned, out, blocks, 0);
        // (the row-gather workgroup holds 48 KB of pixel state in the full design -- FULL instance: 12 floats x 256 pixels x ...;
        // here 32.8 KB + padding to 52 KB so that at most 3 fit a CU, as DESIGN.md's pencil says)
        const double b = run(k_row_gather, out, blocks, per_cu <= 3 ? 52 * 1024 : lds_rg);
        const double trips = (double)blocks * 4 * TRIPS;
        printf("workgroups per CU %d: pinned %.3f ms = %.0f wave trips/us (one entry per HALF: %.0f half-entries/us) | row-gather%s "
               "%.3f ms = %.0f wave trips/us (one entry per ROW: %.0f row-entries/us) | trip cost ratio %.2f\n",
               per_cu, a, trips / (a * 1e3), 2 * trips / (a * 1e3), per_cu <= 3 ? " (52 KB LDS)" : " (32.8 KB LDS)", b,
               trips / (b * 1e3), 4 * trips / (b * 1e3), b / a);
    }
    return 0;
}
