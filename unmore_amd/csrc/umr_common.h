// Shared device/host helpers for the unmore_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/umr.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define UMR_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define UMR_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// 16-byte async global -> LDS copy: LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(UMR_GLOBAL_PTR(gsrc), UMR_LDS_PTR(lds_wave_base), 16, 0, 0);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 4 consecutive elements <-> registers
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void store(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
        bf16x4 t = *(const bf16x4*)p;
        f32x4 r = {(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
        return r;
    }
    static __device__ __forceinline__ void store(bf16_t* p, f32x4 v) {
        bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *(bf16x4*)p = t;
    }
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Throughput-mode (bf16 storage) forms: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution) on
// one v_rcp_f32 + one v_exp_f32 + six FMAs instead of libm's erff (~40 instructions with two divergent branches); the fp32
// parity mode keeps erff.  The derivative shares the exponential: e^{-z^2} with z = x/sqrt(2) is sqrt(2 pi) * pdf(x).
__device__ __forceinline__ void erf_exp_fast(float x, float& erf_z, float& e) {   // erf(x/sqrt2), exp(-x^2/2)
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);   // exp(-x^2/2) = 2^(-x^2 * log2(e)/2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * e;
    erf_z = x < 0.f ? -r : r;
}
template <typename T> __device__ __forceinline__ float gelu_sel(float x) {
    if constexpr (sizeof(T) == 2) {
        float ez, e;
        erf_exp_fast(x, ez, e);
        return 0.5f * x * (1.0f + ez);
    } else {
        return gelu_erf(x);
    }
}
template <typename T> __device__ __forceinline__ float dgelu_sel(float x) {
    if constexpr (sizeof(T) == 2) {
        float ez, e;
        erf_exp_fast(x, ez, e);
        return 0.5f * (1.0f + ez) + x * 0.39894228040143267794f * e;
    } else {
        return dgelu_erf(x);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// zero page: source address for masked LDS-DMA lanes (one copy per translation unit;
// the library is built without relocatable device code)
static __device__ uint4 umr_zero_page[16];

// Raise a kernel's dynamic-LDS limit once per process.  Thread-safe: the flag is published AFTER the attribute call (release /
// acquire), so a thread that sees it set launches with the limit in place; two threads racing both make the (idempotent) call.
#include <atomic>
#define UMR_SET_MAX_LDS_ONCE(fn, bytes)                                                                             \
    do {                                                                                                            \
        static std::atomic<bool> done_{false};                                                                      \
        if (!done_.load(std::memory_order_acquire)) {                                                               \
            (void)hipFuncSetAttribute((const void*)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));      \
            done_.store(true, std::memory_order_release);                                                           \
        }                                                                                                           \
    } while (0)

// environment switch read ONCE per process, thread-safely:  static const int v = umr_env_int("NAME", dflt);
#include <stdlib.h>
static inline int umr_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

// Debug / A-B options that can change WHILE the process runs (tests and probes switch them between launches): include/umr.h,
// umr_set_debug_option.  Each is read from the environment once, when the library is loaded; on the launch path an option costs one
// relaxed atomic load -- no getenv per launch (round-5 review: getenv is not thread-safe against setenv and made the library's
// behaviour depend on process-global state read 650-950 times per step).  umr_opt(): the value (an integer option: atoi of its
// text; a letter option such as UMR_NT_ORDER=n|m: the first character's code) or UMR_OPT_UNSET.
#include <limits.h>
enum {
    UMR_OPT_GEMM_TILE = 0, UMR_OPT_NT_SPLITK, UMR_OPT_SPLITK_FENCE, UMR_OPT_NT_ORDER, UMR_OPT_NT256_PERSIST, UMR_OPT_NT256_BM,
    UMR_OPT_X3_TRACE, UMR_OPT_NT256_PH2, UMR_OPT_ATTN_BWD_FUSED, UMR_OPT_BILINEAR_GY, UMR_OPT_HEAD_OUT_BWD_GENERIC, UMR_OPT_COUNT
};
#define UMR_OPT_UNSET INT_MIN
int umr_opt(int id);                                    // umr_api.hip
static inline int umr_opt_or(int id, int dflt) { const int v = umr_opt(id); return v == UMR_OPT_UNSET ? dflt : v; }

// host-side error plumbing (umr_api.hip)
int umr_set_error(int code, const char* msg);
int umr_f32_mode_now();   // umr_api.hip: UMR_F32_EXACT / UMR_F32_X3
int umr_cu_budget_now();  // umr_api.hip: 0 = every CU, n = the persistent GEMM grids occupy at most n CUs
#define UMR_CHECK_ARG(cond, msg) do { if (!(cond)) return umr_set_error(UMR_ERR_INVALID, msg); } while (0)
#define UMR_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return umr_set_error(UMR_ERR_HIP - (int)e_, hipGetErrorString(e_)); } while (0)
