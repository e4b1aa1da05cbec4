// Fused GEMM epilogue shared by the NT kernels (see include/umr.h for the order of operations).
#pragma once
#include "umr_common.h"

template <typename T>
__device__ __forceinline__ void epilogue_store(const umr_gemm_desc& p, int m_logical, int n, f32x4 v) {
    // v = 4 consecutive columns n..n+3 of logical row m_logical; C-shaped operands use the remapped row m
    int m = m_logical;
    if (p.c_rows_in > 0) m = (m_logical / p.c_rows_in) * p.c_rows_out + p.c_row_off + (m_logical % p.c_rows_in);
    const int m_aux = p.aux_mod > 0 ? (m_logical % p.aux_mod) : m;
    const int nv = p.N - n;  // >0 guaranteed by caller
    const bool full = nv >= 4 && ((p.N & 3) == 0);
    if (p.flags & UMR_EPI_BIAS) {
        if (full) { f32x4 b = *(const f32x4*)(p.bias + n); v += b; }
        else { for (int j = 0; j < 4; ++j) if (j < nv) v[j] += p.bias[n + j]; }
    }
    if (p.flags & UMR_EPI_ROWBIAS) {
        const float* rb = p.rowbias + (int64_t)(m_logical / p.rows_per_batch) * p.N + n;
        if (full) { f32x4 b = *(const f32x4*)rb; v += b; }
        else { for (int j = 0; j < 4; ++j) if (j < nv) v[j] += rb[j]; }
    }
    if (p.flags & (UMR_EPI_ADD_AUX | UMR_EPI_MASK_RELU | UMR_EPI_MASK_DGELU)) {
        const T* ap = (const T*)p.aux + (int64_t)m_aux * p.ldaux + n;
        f32x4 a;
        if (full && ((p.ldaux & 3) == 0)) a = Vec4<T>::load(ap);
        else { for (int j = 0; j < 4; ++j) a[j] = (j < nv) ? to_f32<T>(ap[j]) : 0.f; }
        if (p.flags & UMR_EPI_ADD_AUX) v += a;
        else if (p.flags & UMR_EPI_MASK_RELU) { for (int j = 0; j < 4; ++j) v[j] = a[j] > 0.f ? v[j] : 0.f; }
        else { for (int j = 0; j < 4; ++j) v[j] *= dgelu_sel<T>(a[j]); }
    }
    if (p.flags & UMR_EPI_ADD_AUX2) {
        const T* ap = (const T*)p.aux2 + (int64_t)m * p.ldaux2 + n;
        f32x4 a;
        if (full && ((p.ldaux2 & 3) == 0)) a = Vec4<T>::load(ap);
        else { for (int j = 0; j < 4; ++j) a[j] = (j < nv) ? to_f32<T>(ap[j]) : 0.f; }
        v += a;
    }
    if (p.c2_mode == 2) {
        T* cp = (T*)p.C2 + (int64_t)m * p.ldc2 + n;
        if (full && ((p.ldc2 & 3) == 0)) Vec4<T>::store(cp, v);
        else { for (int j = 0; j < 4; ++j) if (j < nv) cp[j] = from_f32<T>(v[j]); }
    }
    if (p.act == UMR_ACT_RELU) { for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f); }
    else if (p.act == UMR_ACT_GELU) { for (int j = 0; j < 4; ++j) v[j] = gelu_sel<T>(v[j]); }
    else if (p.act == UMR_ACT_TANH) { for (int j = 0; j < 4; ++j) v[j] = tanhf(v[j]); }
    else if (p.act == UMR_ACT_SIGMOID) { for (int j = 0; j < 4; ++j) v[j] = 1.f / (1.f + expf(-v[j])); }
    if (p.flags & UMR_EPI_OUT_F32) {
        float* cp = (float*)p.C + (int64_t)m * p.ldc + n;
        if (full && ((p.ldc & 3) == 0)) *(f32x4*)cp = v;
        else { for (int j = 0; j < 4; ++j) if (j < nv) cp[j] = v[j]; }
    } else {
        T* cp = (T*)p.C + (int64_t)m * p.ldc + n;
        if (full && ((p.ldc & 3) == 0)) Vec4<T>::store(cp, v);
        else { for (int j = 0; j < 4; ++j) if (j < nv) cp[j] = from_f32<T>(v[j]); }
    }
    if (p.c2_mode == 1) {
        T* cp = (T*)p.C2 + (int64_t)m * p.ldc2 + n;
        f32x4 r = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        if (full && ((p.ldc2 & 3) == 0)) Vec4<T>::store(cp, r);
        else { for (int j = 0; j < 4; ++j) if (j < nv) cp[j] = from_f32<T>(r[j]); }
    }
}

// 8 consecutive elements <-> two f32x4
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, f32x4& a, f32x4& b) { a = *(const f32x4*)p; b = *(const f32x4*)(p + 4); }
    static __device__ __forceinline__ void store(float* p, f32x4 a, f32x4 b) { *(f32x4*)p = a; *(f32x4*)(p + 4) = b; }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, f32x4& a, f32x4& b) {
        const bf16x8 t = *(const bf16x8*)p;
        a = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
        b = f32x4{(float)t[4], (float)t[5], (float)t[6], (float)t[7]};
    }
    static __device__ __forceinline__ void store(bf16_t* p, f32x4 a, f32x4 b) {
        bf16x8 t;
        t[0] = (bf16_t)a[0]; t[1] = (bf16_t)a[1]; t[2] = (bf16_t)a[2]; t[3] = (bf16_t)a[3];
        t[4] = (bf16_t)b[0]; t[5] = (bf16_t)b[1]; t[6] = (bf16_t)b[2]; t[7] = (bf16_t)b[3];
        *(bf16x8*)p = t;
    }
};

// vector form of epilogue_store: 8 consecutive columns n..n+7 of logical row m_logical (all in range, all
// row strides multiples of 8 elements)
template <typename T>
__device__ __forceinline__ void epilogue_store8(const umr_gemm_desc& p, int m_logical, int n, f32x4 v0, f32x4 v1) {
    int m = m_logical;
    if (p.c_rows_in > 0) m = (m_logical / p.c_rows_in) * p.c_rows_out + p.c_row_off + (m_logical % p.c_rows_in);
    const int m_aux = p.aux_mod > 0 ? (m_logical % p.aux_mod) : m;
    if (p.flags & UMR_EPI_BIAS) { v0 += *(const f32x4*)(p.bias + n); v1 += *(const f32x4*)(p.bias + n + 4); }
    if (p.flags & UMR_EPI_ROWBIAS) {
        const float* rb = p.rowbias + (int64_t)(m_logical / p.rows_per_batch) * p.N + n;
        v0 += *(const f32x4*)rb; v1 += *(const f32x4*)(rb + 4);
    }
    if (p.flags & (UMR_EPI_ADD_AUX | UMR_EPI_MASK_RELU | UMR_EPI_MASK_DGELU)) {
        f32x4 a0, a1;
        Vec8<T>::load((const T*)p.aux + (int64_t)m_aux * p.ldaux + n, a0, a1);
        if (p.flags & UMR_EPI_ADD_AUX) { v0 += a0; v1 += a1; }
        else if (p.flags & UMR_EPI_MASK_RELU) {
            for (int j = 0; j < 4; ++j) { v0[j] = a0[j] > 0.f ? v0[j] : 0.f; v1[j] = a1[j] > 0.f ? v1[j] : 0.f; }
        } else {
            for (int j = 0; j < 4; ++j) { v0[j] *= dgelu_sel<T>(a0[j]); v1[j] *= dgelu_sel<T>(a1[j]); }
        }
    }
    if (p.flags & UMR_EPI_ADD_AUX2) {
        f32x4 a0, a1;
        Vec8<T>::load((const T*)p.aux2 + (int64_t)m * p.ldaux2 + n, a0, a1);
        v0 += a0; v1 += a1;
    }
    if (p.c2_mode == 2) Vec8<T>::store((T*)p.C2 + (int64_t)m * p.ldc2 + n, v0, v1);
    if (p.act == UMR_ACT_RELU) { for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(v0[j], 0.f); v1[j] = fmaxf(v1[j], 0.f); } }
    else if (p.act == UMR_ACT_GELU) { for (int j = 0; j < 4; ++j) { v0[j] = gelu_sel<T>(v0[j]); v1[j] = gelu_sel<T>(v1[j]); } }
    else if (p.act == UMR_ACT_TANH) { for (int j = 0; j < 4; ++j) { v0[j] = tanhf(v0[j]); v1[j] = tanhf(v1[j]); } }
    else if (p.act == UMR_ACT_SIGMOID) { for (int j = 0; j < 4; ++j) { v0[j] = 1.f / (1.f + expf(-v0[j])); v1[j] = 1.f / (1.f + expf(-v1[j])); } }
    if (p.flags & UMR_EPI_OUT_F32) Vec8<float>::store((float*)p.C + (int64_t)m * p.ldc + n, v0, v1);
    else Vec8<T>::store((T*)p.C + (int64_t)m * p.ldc + n, v0, v1);
    if (p.c2_mode == 1) {
        for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(v0[j], 0.f); v1[j] = fmaxf(v1[j], 0.f); }
        Vec8<T>::store((T*)p.C2 + (int64_t)m * p.ldc2 + n, v0, v1);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Epilogue of the UMR_BF16X3 (fp32-grade, bf16-plane operand) GEMMs, shared by the persistent kernel's copy-out loop and by the
// split-K finish kernel (gemm_nt256p.hip).  f32 arithmetic throughout; every tensor operand is either f32 [M][N] or three bf16
// planes [M][h(N) | m(N) | l(N)] whose sum IS the f32 value (h + m is exact in f32, + l restores all 24 bits).
// Order as documented in include/umr.h: (+bias) (+rowbias); mask / GELU' / add aux; add aux2; C2 (mode 2) = v; act; C; C2 (mode 1) = relu.
__device__ __forceinline__ void x3_planes_load8(const bf16_t* q, int N, f32x4& a, f32x4& b) {
    const bf16x8 h = *(const bf16x8*)q, m = *(const bf16x8*)(q + N), l = *(const bf16x8*)(q + 2 * (int64_t)N);
    a = f32x4{((float)h[0] + (float)m[0]) + (float)l[0], ((float)h[1] + (float)m[1]) + (float)l[1],
              ((float)h[2] + (float)m[2]) + (float)l[2], ((float)h[3] + (float)m[3]) + (float)l[3]};
    b = f32x4{((float)h[4] + (float)m[4]) + (float)l[4], ((float)h[5] + (float)m[5]) + (float)l[5],
              ((float)h[6] + (float)m[6]) + (float)l[6], ((float)h[7] + (float)m[7]) + (float)l[7]};
}
__device__ __forceinline__ void x3_planes_store8(bf16_t* q, int N, const f32x4& a, const f32x4& b) {
    bf16x8 h, mm, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? a[e] : b[e - 4];
        const bf16_t hh = (bf16_t)x;
        const float r1 = x - (float)hh;
        const bf16_t m2 = (bf16_t)r1;
        h[e] = hh; mm[e] = m2; l[e] = (bf16_t)(r1 - (float)m2);
    }
    *(bf16x8*)q = h;
    *(bf16x8*)(q + N) = mm;
    *(bf16x8*)(q + 2 * (int64_t)N) = l;
}

// 8 consecutive columns n..n+7 (all < N, N % 8 == 0) of logical row m.  ADD_BIAS: the GEMM kernel starts its accumulators from
// the bias and passes false; the finish kernel adds it here.
template <bool ADD_BIAS>
__device__ __forceinline__ void x3_epilogue_store8(const umr_gemm_desc& p, int m_logical, int n, f32x4 v0, f32x4 v1) {
    int m = m_logical;
    if (p.c_rows_in > 0) m = (m_logical / p.c_rows_in) * p.c_rows_out + p.c_row_off + (m_logical % p.c_rows_in);
    if (ADD_BIAS && (p.flags & UMR_EPI_BIAS)) { v0 += *(const f32x4*)(p.bias + n); v1 += *(const f32x4*)(p.bias + n + 4); }
    if (p.flags & UMR_EPI_ROWBIAS) {
        const float* rb = p.rowbias + (int64_t)(m_logical / p.rows_per_batch) * p.N + n;
        v0 += *(const f32x4*)rb; v1 += *(const f32x4*)(rb + 4);
    }
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};   // argument of the erf evaluation (GELU: v, GELU' mask: aux)
    if (p.flags & (UMR_EPI_ADD_AUX | UMR_EPI_MASK_RELU | UMR_EPI_MASK_DGELU)) {
        const int m_aux = p.aux_mod > 0 ? (m_logical % p.aux_mod) : m;
        f32x4 a0, a1;
        if (p.flags & UMR_EPI_AUX_X3) {
            const bf16_t* q = (const bf16_t*)p.aux + (int64_t)m_aux * p.ldaux + n;
            if (p.flags & UMR_EPI_MASK_RELU) {   // the sign of the value is the sign of its leading plane
                const bf16x8 h = *(const bf16x8*)q;
                a0 = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                a1 = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
            } else {
                x3_planes_load8(q, p.N, a0, a1);
            }
        } else {
            const float* q = (const float*)p.aux + (int64_t)m_aux * p.ldaux + n;
            a0 = *(const f32x4*)q; a1 = *(const f32x4*)(q + 4);
        }
        if (p.flags & UMR_EPI_ADD_AUX) { v0 += a0; v1 += a1; }
        else if (p.flags & UMR_EPI_MASK_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = a0[e] > 0.f ? v0[e] : 0.f; v1[e] = a1[e] > 0.f ? v1[e] : 0.f; }
        } else {
            // GELU'-masked gradient (no aux2 / C2 / activation with it: the dispatcher checks): shares the ONE inlined erf
            // evaluation below with the GELU activation -- two copies of erff x 8 elements doubled this epilogue's code
            g0 = a0; g1 = a1;
        }
    }
    if (p.flags & UMR_EPI_ADD_AUX2) {
        f32x4 a0, a1;
        if (p.flags & UMR_EPI_AUX2_X3) x3_planes_load8((const bf16_t*)p.aux2 + (int64_t)m * p.ldaux2 + n, p.N, a0, a1);
        else { const float* q = (const float*)p.aux2 + (int64_t)m * p.ldaux2 + n; a0 = *(const f32x4*)q; a1 = *(const f32x4*)(q + 4); }
        v0 += a0; v1 += a1;
    }
    if (p.c2_mode == 2) {
        if (p.flags & UMR_EPI_C2_X3) x3_planes_store8((bf16_t*)p.C2 + (int64_t)m * p.ldc2 + n, p.N, v0, v1);
        else { float* q = (float*)p.C2 + (int64_t)m * p.ldc2 + n; *(f32x4*)q = v0; *(f32x4*)(q + 4) = v1; }
    }
    const bool dgelu = (p.flags & UMR_EPI_MASK_DGELU) != 0;
    if (p.act == UMR_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
    } else if (p.act == UMR_ACT_GELU || dgelu) {
        if (!dgelu) { g0 = v0; g1 = v1; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = e < 4 ? g0[e] : g1[e - 4];
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            float r;
            if (dgelu) r = (e < 4 ? v0[e] : v1[e - 4]) * (cdf + x * (0.39894228040143267794f * __expf(-0.5f * x * x)));   // = v * dgelu_erf(x)
            else r = x * cdf;                                                                                             // = gelu_erf(x)
            if (e < 4) v0[e] = r; else v1[e - 4] = r;
        }
    }
    if (p.flags & UMR_EPI_OUT_X3) x3_planes_store8((bf16_t*)p.C + (int64_t)m * p.ldc + n, p.N, v0, v1);
    else { float* q = (float*)p.C + (int64_t)m * p.ldc + n; *(f32x4*)q = v0; *(f32x4*)(q + 4) = v1; }
    if (p.c2_mode == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
        if (p.flags & UMR_EPI_C2_X3) x3_planes_store8((bf16_t*)p.C2 + (int64_t)m * p.ldc2 + n, p.N, v0, v1);
        else { float* q = (float*)p.C2 + (int64_t)m * p.ldc2 + n; *(f32x4*)q = v0; *(f32x4*)(q + 4) = v1; }
    }
}
