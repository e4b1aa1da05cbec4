// Issue rates of the instructions an attention tile is made of, per SIMD, in matrix-pipe terms (round 6: why the forward's tile loop
// tops out near half of the MFMA peak at head dimension 64 -- DESIGN.md section 5, "attention ceiling").
//
// One 16x16x32 bf16 MFMA is 16,384 FLOP.  A softmax-attention forward spends 4 * hd = 256 FLOP per score at hd = 64 (q.k and p.v), so
// ONE MFMA's worth of matrix work comes with 64 scores = ONE wave-wide instruction of every per-score VALU step: one v_exp_f32, one
// scale-and-subtract, one running-sum add, one max, half a v_cvt_pk_bf16_f32, ...  This probe measures how long the SIMD needs for each
// of those next to how long it needs for the MFMA, alone and mixed (same wave / other waves of the SIMD).
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Prints one JSON line per (case, waves per SIMD): wave-instructions per SIMD per microsecond and the time per instruction relative to
// the lone MFMA's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

enum { C_MFMA, C_EXP, C_FMA, C_PKFMA, C_MAX, C_CVT, C_PERM, C_MFMA_EXP, C_MFMA_EXP4, C_MFMA_FMA4, C_ATTN_MIX, C_N };
static const char* NAMES[C_N] = {"mfma_16x16x32_bf16", "v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_max_f32", "v_cvt_pk_bf16_f32",
                                 "v_permlane32_swap", "mfma + 1 v_exp_f32 (same wave)", "mfma + 1 v_exp_f32 + 3 v_fma_f32 (same wave)",
                                 "mfma + 4 v_fma_f32 (same wave)", "mfma + exp + fma + add + max + cvt/2 (one tile's mix, same wave)"};
// wave-instructions of the counted kind per loop iteration
static const int PER_ITER[C_N] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};

template <int CASE>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, int iters, float seed) {
    // 16 independent values: no instruction waits for its predecessor's result
    float v[16];
    f32x2 pv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = seed + 0.001f * (float)(threadIdx.x + i); pv[i] = f32x2{v[i], v[i] + 1.f}; }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (bf16_t)(seed + j); b[j] = (bf16_t)(seed - j); }
    const float c0 = seed * 0.5f, c1 = seed * 0.25f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (CASE == C_MFMA || CASE >= C_MFMA_EXP) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
            if (CASE == C_EXP || CASE == C_MFMA_EXP || CASE == C_MFMA_EXP4 || CASE == C_ATTN_MIX) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (CASE == C_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
            if (CASE == C_MFMA_EXP4 || CASE == C_MFMA_FMA4) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 1) & 15]) : "v"(c0), "v"(c1));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 2) & 15]) : "v"(c0), "v"(c1));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 3) & 15]) : "v"(c0), "v"(c1));
                if (CASE == C_MFMA_FMA4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 4) & 15]) : "v"(c0), "v"(c1));
            }
            if (CASE == C_ATTN_MIX) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + 1) & 15]) : "v"(c0), "v"(c1));     // score * scale - max
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(i + 2) & 15]) : "v"(c0));                    // running sum
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[(i + 3) & 15]) : "v"(c1));                    // running max
                if (i & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[(i + 4) & 15]) : "v"(c1)); // two probabilities per instruction
            }
            if (CASE == C_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[i]) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1}));
            if (CASE == C_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
            if (CASE == C_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
            if (CASE == C_PERM) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i + 8) & 15]));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i] + pv[i][0] + pv[i][1] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;     // never true: keeps the work alive
}

template <int CASE>
static int run(float* dout, int cus, double mfma_ns[3]) {
    const int waves_per_simd[3] = {1, 2, 4};
    for (int wi = 0; wi < 3; ++wi) {
        const int threads = 256 * waves_per_simd[wi];      // one workgroup per CU: 4 SIMDs x waves
        const int iters = 20000;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rate_kernel<CASE>, dim3(cus), dim3(threads), 0, 0, dout, 2000, 1.0f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate_kernel<CASE>, dim3(cus), dim3(threads), 0, 0, dout, iters, 1.0f);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        // per SIMD: waves x iters x PER_ITER instructions of the counted kind
        const double n = (double)waves_per_simd[wi] * iters * PER_ITER[CASE];
        const double ns_per = ms * 1e6 / n;
        if (CASE == C_MFMA) mfma_ns[wi] = ns_per;
        printf("{\"case\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_wave_instruction_group_per_simd\": %.4f, \"relative_to_lone_mfma\": %.3f, "
               "\"mfma_share_of_peak_if_this_were_the_tile_loop\": %.3f}\n",
               NAMES[CASE], waves_per_simd[wi], ms, ns_per, ns_per / mfma_ns[wi],
               (CASE == C_MFMA || CASE >= C_MFMA_EXP) ? (16384.0 * 4 * cus / ns_per) / 2.5e6 : 0.0);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    float* dout;
    CHK(hipMalloc(&dout, 4096));
    double mfma_ns[3] = {1, 1, 1};
    if (run<C_MFMA>(dout, cus, mfma_ns)) return 1;
    if (run<C_EXP>(dout, cus, mfma_ns)) return 1;
    if (run<C_FMA>(dout, cus, mfma_ns)) return 1;
    if (run<C_PKFMA>(dout, cus, mfma_ns)) return 1;
    if (run<C_MAX>(dout, cus, mfma_ns)) return 1;
    if (run<C_CVT>(dout, cus, mfma_ns)) return 1;
    if (run<C_PERM>(dout, cus, mfma_ns)) return 1;
    if (run<C_MFMA_EXP>(dout, cus, mfma_ns)) return 1;
    if (run<C_MFMA_EXP4>(dout, cus, mfma_ns)) return 1;
    if (run<C_MFMA_FMA4>(dout, cus, mfma_ns)) return 1;
    if (run<C_ATTN_MIX>(dout, cus, mfma_ns)) return 1;
    return 0;
}
