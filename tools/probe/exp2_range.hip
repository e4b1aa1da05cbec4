// What does the raw v_exp_f32 (__builtin_amdgcn_exp2f, no range reduction) return below -126, and what do the bf16 conversion and the
// bf16 MFMA make of it?  (the attention forward's non-finite row, round 6: a lane whose 16 scores all sit more than 128 below the row maximum)
//   hipcc --offload-arch=gfx950 -O3 tools/probe/exp2_range.hip -o /tmp/exp2_range && /tmp/exp2_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const float* x, float* e, unsigned short* b, float* mf, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        const float v = __builtin_amdgcn_exp2f(x[i]);
        e[i] = v;
        const bf16_t h = (bf16_t)v;
        unsigned short u; __builtin_memcpy(&u, &h, 2);
        b[i] = u;
    }
    // MFMA: A = ones, B = the converted value in every k slot of column (lane & 15): D = 32 * value
    bf16x8 ones, bv;
    const float v = __builtin_amdgcn_exp2f(x[(threadIdx.x & 15) < n ? (threadIdx.x & 15) : 0]);
    for (int j = 0; j < 8; ++j) { ones[j] = (bf16_t)1.0f; bv[j] = (bf16_t)v; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bv, acc, 0, 0, 0);
    if (threadIdx.x < 16) mf[threadIdx.x] = acc[0];
}
int main() {
    const int n = 16;
    float hx[n] = {-100.f, -120.f, -125.5f, -126.f, -126.5f, -127.f, -127.8f, -128.f, -128.5f, -129.07f, -130.f, -137.97f, -140.f, -149.f, -150.f, -200.f};
    float *dx, *de, *dm; unsigned short* db;
    hipMalloc(&dx, n * 4); hipMalloc(&de, n * 4); hipMalloc(&dm, 16 * 4); hipMalloc(&db, n * 2);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, de, db, dm, n);
    float he[n], hm[16]; unsigned short hb[n];
    hipMemcpy(he, de, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hm, dm, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        unsigned u; memcpy(&u, &he[i], 4);
        printf("x = %8.2f  v_exp_f32 -> %.6e (bits %08x)  bf16 bits %04x  mfma(ones, bf16 value) / 32 = %.6e\n", hx[i], he[i], u, hb[i], hm[i] / 32.f);
    }
    return 0;
}
