// Bare MFMA loops for the energy table: the SAME 128 x 64 output tile per wave, the SAME LDS fragment reads
// (24 ds_read_b128 per 64-deep K-tile and wave = the 192 KiB per K-tile of the 256 x 256 GEMM kernel's eight waves), on
// v_mfma_f32_16x16x32_bf16 (64 per K-tile) or v_mfma_f32_32x32x16_bf16 (32 per K-tile), and the 16x16x32 loop at BK = 32
// (two barriers per 64 of K instead of one).  No global traffic in the loop: what differs is the matrix instruction (and the
// barrier count).  Built and driven by tools/probe/mfma_shape.py on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// LDS image: A [256 rows][128 B], B [256 rows][128 B]; 16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
template <int VARIANT>   // 0: 16x16x32, BK 64   1: 32x32x16, BK 64   2: 16x16x32, BK 32 (barrier per half K-tile)
__global__ __launch_bounds__(512, 2) void mfma_loop(const uint4* __restrict__ init, float* __restrict__ out, int ktiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 2 * 256 * 8; i += 512) ((uint4*)smem)[i] = init[(i + blockIdx.x * 17) & 4095];
    __syncthreads();
    const int wr = w >> 2, wc = w & 3;
    const char* sa = smem + wr * 128 * 128;
    const char* sb = smem + 256 * 128 + wc * 64 * 128;
    float total = 0.f;
    if constexpr (VARIANT == 1) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int r = lane & 31, h = lane >> 5;
#pragma unroll 1
        for (int t = 0; t < ktiles; ++t) {
            asm volatile("" ::: "memory");     // the fragments are re-read from LDS every K-tile, as in the GEMM
            bf16x8 fa[4][4], fb[2][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int row = i * 32 + r; fa[i][ks] = *(const bf16x8*)(sa + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4)); }
#pragma unroll
                for (int j = 0; j < 2; ++j) { const int row = j * 32 + r; fb[j][ks] = *(const bf16x8*)(sb + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4)); }
            }
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks], fb[j][ks], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) total += acc[i][j][e];
    } else {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int r = lane & 15, q = lane >> 4;
#pragma unroll 1
        for (int t = 0; t < ktiles; ++t) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[8], fb[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) { const int row = i * 16 + r; fa[i] = *(const bf16x8*)(sa + row * 128 + (((ks * 4 + q) ^ ((row >> 1) & 7)) << 4)); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int row = j * 16 + r; fb[j] = *(const bf16x8*)(sb + row * 128 + (((ks * 4 + q) ^ ((row >> 1) & 7)) << 4)); }
                if (VARIANT == 2 || ks == 0) __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    out[blockIdx.x * 512 + tid] = total;
}

extern "C" int mfma_shape_run(int variant, const void* init, float* out, int blocks, int ktiles, void* stream) {
    const size_t lds = 2 * 256 * 128;
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(512), lds, s, (const uint4*)init, out, ktiles);
    else if (variant == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(512), lds, s, (const uint4*)init, out, ktiles);
    else hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(512), lds, s, (const uint4*)init, out, ktiles);
    return (int)hipGetLastError();
}
