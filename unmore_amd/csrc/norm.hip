// LayerNorm forward / backward over rows of D channels (timm LayerNorm, eps 1e-6,
// biased variance; vit.py:196-199 via timm Block).  HBM-bound: one wave per row,
// 16-byte vector loads, statistics in f32, two-pass variance in registers.
#include "umr_common.h"
#include <stdlib.h>

namespace {

constexpr int LN_MAXV = 8;  // up to 8 x 4 elements per lane -> D <= 2048

// PLANES (T = float): y is written as three bf16 planes per row [h(D) | m(D) | l(D)] (UMR_BF16X3, include/umr.h) -- the operand
// format of the fp32-grade plane GEMMs; the normalised rows feed only GEMMs (qkv / fc1 and their weight gradients)
template <typename T, bool PLANES = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                     float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + (int64_t)row * D;
    f32x4 v[LN_MAXV];
    const int nv = D >> 2;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) { v[i] = Vec4<T>::load(xr + c * 4); s += v[i][0] + v[i][1] + v[i][2] + v[i][3]; }
    }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) { for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mu; q += d * d; } }
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    T* yr = y + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            const f32x4 g = *(const f32x4*)(gamma + c * 4), b = *(const f32x4*)(beta + c * 4);
            f32x4 o;
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[j] + b[j];
            if constexpr (PLANES) {
                bf16x4 h, m2, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16_t hh = (bf16_t)o[j];
                    const float r1 = o[j] - (float)hh;
                    const bf16_t mm = (bf16_t)r1;
                    h[j] = hh; m2[j] = mm; l[j] = (bf16_t)(r1 - (float)mm);
                }
                bf16_t* yp = (bf16_t*)y + (int64_t)row * 3 * D + c * 4;
                *(bf16x4*)yp = h; *(bf16x4*)(yp + D) = m2; *(bf16x4*)(yp + 2 * D) = l;
            } else {
                Vec4<T>::store(yr + c * 4, o);
            }
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// bf16 fast path (D % 8 == 0): 16-byte loads (8 channels per lane and chunk) and TWO rows per wave, so that twice the bytes
// are in flight per wave and the two rows' reductions interleave (the 4-channel form measured 1.6 TB/s)

__device__ __forceinline__ void ld8(const bf16_t* p, f32x4& a, f32x4& b) {
    const bf16x8 t = *(const bf16x8*)p;
    a = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    b = f32x4{(float)t[4], (float)t[5], (float)t[6], (float)t[7]};
}
__device__ __forceinline__ void st8(bf16_t* p, f32x4 a, f32x4 b) {
    bf16x8 t;
    t[0] = (bf16_t)a[0]; t[1] = (bf16_t)a[1]; t[2] = (bf16_t)a[2]; t[3] = (bf16_t)a[3];
    t[4] = (bf16_t)b[0]; t[5] = (bf16_t)b[1]; t[6] = (bf16_t)b[2]; t[7] = (bf16_t)b[3];
    *(bf16x8*)p = t;
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_bf16x8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= M) return;
    const int nc = D >> 3;
    f32x4 va[2][NCH], vb[2][NCH];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const bool rok = row0 + r < M;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 64 + lane;
            va[r][i] = vb[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rok && c < nc) ld8(x + (int64_t)(row0 + r) * D + c * 8, va[r][i], vb[r][i]);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            s[r] += (va[r][i][0] + va[r][i][1] + va[r][i][2] + va[r][i][3]) + (vb[r][i][0] + vb[r][i][1] + vb[r][i][2] + vb[r][i][3]);
    float mu[2], rs[2];
    mu[0] = wave_sum(s[0]) / D;
    mu[1] = wave_sum(s[1]) / D;
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = i * 64 + lane;
            if (c < nc) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d0 = va[r][i][j] - mu[r], d1 = vb[r][i][j] - mu[r]; q[r] += d0 * d0 + d1 * d1; }
            }
        }
    rs[0] = rsqrtf(wave_sum(q[0]) / D + eps);
    rs[1] = rsqrtf(wave_sum(q[1]) / D + eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 64 + lane;
        if (c < nc) {
            const f32x4 g0 = *(const f32x4*)(gamma + c * 8), g1 = *(const f32x4*)(gamma + c * 8 + 4);
            const f32x4 b0 = *(const f32x4*)(beta + c * 8), b1 = *(const f32x4*)(beta + c * 8 + 4);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (row0 + r >= M) continue;
                f32x4 o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) { o0[j] = (va[r][i][j] - mu[r]) * rs[r] * g0[j] + b0[j]; o1[j] = (vb[r][i][j] - mu[r]) * rs[r] * g1[j] + b1[j]; }
                st8(y + (int64_t)(row0 + r) * D + c * 8, o0, o1);
            }
        }
    }
    if (lane == 0) {
        mean[row0] = mu[0]; rstd[row0] = rs[0];
        if (row0 + 1 < M) { mean[row0 + 1] = mu[1]; rstd[row0 + 1] = rs[1]; }
    }
}

// dx = rstd * (dy*g - mean(dy*g) - xhat*mean(dy*g*xhat)); optional residual gradient add.
// Per-block partial dgamma/dbeta go to workspace [gridDim.x][2][D] (reduced by ln_bwd_reduce).
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const T* __restrict__ dres,
                                                     T* __restrict__ dx, float* __restrict__ part, int M, int D,
                                                     int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nv = D >> 2;
    f32x4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) { ag[i] = f32x4{0, 0, 0, 0}; ab[i] = f32x4{0, 0, 0, 0}; }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    for (int row = r0 + wv; row < r1; row += 4) {
        const T* xr = x + (int64_t)row * D;
        const T* dr = dy + (int64_t)row * D;
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[LN_MAXV], dg[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                const f32x4 xv = Vec4<T>::load(xr + c * 4), dv = Vec4<T>::load(dr + c * 4);
                const f32x4 g = *(const f32x4*)(gamma + c * 4);
                for (int j = 0; j < 4; ++j) {
                    xh[i][j] = (xv[j] - mu) * rs;
                    dg[i][j] = dv[j] * g[j];
                    s1 += dg[i][j];
                    s2 += dg[i][j] * xh[i][j];
                    ag[i][j] += dv[j] * xh[i][j];
                    ab[i][j] += dv[j];
                }
            }
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
        T* ox = dx + (int64_t)row * D;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = i * 64 + lane;
            if (c < nv) {
                f32x4 o;
                for (int j = 0; j < 4; ++j) o[j] = rs * (dg[i][j] - s1 - xh[i][j] * s2);
                if (dres != nullptr) { const f32x4 r = Vec4<T>::load(dres + (int64_t)row * D + c * 4); o += r; }
                Vec4<T>::store(ox + c * 4, o);
            }
        }
    }
    // block reduce of the 4 waves' partial sums through LDS, fixed order
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = i * 64 + lane;
        if (c < nv) {
            *(f32x4*)(sh + (wv * 2 + 0) * D + c * 4) = ag[i];
            *(f32x4*)(sh + (wv * 2 + 1) * D + c * 4) = ab[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c / D, col = c - which * D;
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += sh[(k * 2 + which) * D + col];
        part[((int64_t)blockIdx.x * 2 + which) * D + col] = s;
    }
}

// bf16 fast path of the backward (see ln_fwd_bf16x8_kernel): 16-byte loads, two rows per wave and iteration
#ifndef LN_BWD_ROWS
#define LN_BWD_ROWS 1   // rows per wave and iteration: 1 measured 6 % faster than 2 at 36928 x 768 (fewer registers), 4 is 60 % slower
#endif
template <int NCH, int R = LN_BWD_ROWS>
__global__ __launch_bounds__(256) void ln_bwd_bf16x8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                            bf16_t* __restrict__ dx, float* __restrict__ part, int M, int D,
                                                            int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [4 waves][2][D]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nc = D >> 3;
    f32x4 ag[NCH][2], ab[NCH][2], gm[NCH][2];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 64 + lane;
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            ag[i][hlf] = ab[i][hlf] = gm[i][hlf] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nc) gm[i][hlf] = *(const f32x4*)(gamma + c * 8 + hlf * 4);
        }
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    for (int row = r0 + R * wv; row < r1; row += 4 * R) {
        f32x4 xh[R][NCH][2], dg[R][NCH][2];
        float s1[R], s2[R], rsv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) s1[r] = s2[r] = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool rok = row + r < r1;
            const float mu = rok ? mean[row + r] : 0.f;
            rsv[r] = rok ? rstd[row + r] : 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = i * 64 + lane;
                f32x4 xv[2], dv[2];
                xv[0] = xv[1] = dv[0] = dv[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (rok && c < nc) {
                    ld8(x + (int64_t)(row + r) * D + c * 8, xv[0], xv[1]);
                    ld8(dy + (int64_t)(row + r) * D + c * 8, dv[0], dv[1]);
                }
#pragma unroll
                for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xhv = (rok && c < nc) ? (xv[hlf][j] - mu) * rsv[r] : 0.f;
                        const float dgv = dv[hlf][j] * gm[i][hlf][j];
                        xh[r][i][hlf][j] = xhv;
                        dg[r][i][hlf][j] = dgv;
                        s1[r] += dgv;
                        s2[r] += dgv * xhv;
                        ag[i][hlf][j] += dv[hlf][j] * xhv;
                        ab[i][hlf][j] += dv[hlf][j];
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { s1[r] = wave_sum(s1[r]) / D; s2[r] = wave_sum(s2[r]) / D; }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (row + r >= r1) continue;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = i * 64 + lane;
                if (c < nc) {
                    f32x4 o[2];
#pragma unroll
                    for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[hlf][j] = rsv[r] * (dg[r][i][hlf][j] - s1[r] - xh[r][i][hlf][j] * s2[r]);
                    if (dres != nullptr) {
                        f32x4 q0, q1;
                        ld8(dres + (int64_t)(row + r) * D + c * 8, q0, q1);
                        o[0] += q0; o[1] += q1;
                    }
                    st8(dx + (int64_t)(row + r) * D + c * 8, o[0], o[1]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = i * 64 + lane;
        if (c < nc) {
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                *(f32x4*)(sh + (wv * 2 + 0) * D + c * 8 + hlf * 4) = ag[i][hlf];
                *(f32x4*)(sh + (wv * 2 + 1) * D + c * 8 + hlf * 4) = ab[i][hlf];
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * D; c += 256) {
        const int which = c / D, col = c - which * D;
        float s_ = 0.f;
        for (int k = 0; k < 4; ++k) s_ += sh[(k * 2 + which) * D + col];
        part[((int64_t)blockIdx.x * 2 + which) * D + col] = s_;
    }
}

// fixed-order sum of the per-block partials (bitwise reproducible)
__global__ __launch_bounds__(1024) void ln_bwd_reduce(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      int nblocks, int D, int accumulate) {
    // 64 columns x 16 slab phases per workgroup, 8 loads in flight per thread; phases added in index order (reproducible).
    // (4 phases x 4 chains walked 512 slabs in 32 dependent steps: 13 us for 3 MB)
    __shared__ float sh[16][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < 2 * D) {
        const int which = c / D, col = c - which * D;
        const float* q = part + (int64_t)which * D + col;
        const int64_t st = (int64_t)2 * D;
        int b = sl;
        for (; b + 16 * 7 < nblocks; b += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = q[(int64_t)(b + 16 * u) * st];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; b < nblocks; b += 16) acc += q[(int64_t)b * st];
    }
    sh[sl][cl] = acc;
    __syncthreads();
    if (sl == 0 && c < 2 * D) {
        float s = sh[0][cl];
#pragma unroll
        for (int y = 1; y < 16; ++y) s += sh[y][cl];
        const int which = c / D, col = c - which * D;
        float* o = which == 0 ? dgamma : dbeta;
        o[col] = accumulate ? o[col] + s : s;
    }
}

}  // namespace

extern "C" int umr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                 int M, int D, float eps, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
    UMR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0 && D <= 256 * LN_MAXV, "layernorm_fwd: D must be a multiple of 8, <= 2048");
    hipStream_t s = (hipStream_t)stream;
    dim3 g((M + 3) / 4), b(256);
    if (dtype == UMR_BF16 && D % 8 == 0 && D <= 1024)
        hipLaunchKernelGGL(ln_fwd_bf16x8_kernel<2>, dim3((M + 7) / 8), b, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps);
    else if (dtype == UMR_BF16 && D % 8 == 0 && D <= 2048)
        hipLaunchKernelGGL(ln_fwd_bf16x8_kernel<4>, dim3((M + 7) / 8), b, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps);
    else if (dtype == UMR_BF16) hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps);
    else if (dtype == UMR_F32) hipLaunchKernelGGL(ln_fwd_kernel<float>, g, b, 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, eps);
    else if (dtype == UMR_BF16X3) hipLaunchKernelGGL((ln_fwd_kernel<float, true>), g, b, 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, eps);
    else return umr_set_error(UMR_ERR_INVALID, "layernorm_fwd: dtype");
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

// partial slabs: one per workgroup; a workgroup takes >= 8 rows (one pass of its four waves x two rows).  64 rows per
// workgroup left a 1300-token batch (the reference's 128^2 recipe) on 21 of 256 CUs: 44 us for 5 MB
static int ln_bwd_blocks(int M) {
    int nb = (M + 7) / 8;
    if (nb > 1024) nb = 1024;
    return nb;
}

extern "C" int64_t umr_layernorm_bwd_workspace(int M, int D) {
    int nb = ln_bwd_blocks(M);
    return (int64_t)nb * 2 * D * 4;
}

// workgroups of the row pass and rows per workgroup (the parameter pass reduces exactly that many partial sums)
static void ln_bwd_plan(int M, int* nb_out, int* rpb_out) {
    int nb = ln_bwd_blocks(M);
    {
        // ONE round of workgroups: the bf16 kernel needs 214 VGPRs (two workgroups per CU), so more than 2 x CUs blocks run as a
        // second, nearly empty round -- 577 blocks at the cfg2 token count took two rounds for 1.13 rounds of work
        static const int cus = [] {
            int dev = 0;
            hipDeviceProp_t prop;
            return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                       ? prop.multiProcessorCount : 256;
        }();
        // UMR_LN_BWD_WG_PER_CU: experiment hook (register use decides how many are co-resident); initialised once, thread-safe
        static const int wg_per_cu = [] { const char* e = getenv("UMR_LN_BWD_WG_PER_CU"); const int v = e ? atoi(e) : 2; return v < 1 ? 2 : v; }();
        if (nb > wg_per_cu * cus) nb = wg_per_cu * cus;
    }
    const int rpb = (M + nb - 1) / nb;
    *nb_out = (M + rpb - 1) / rpb;
    *rpb_out = rpb;
}

// row pass: dx (+ dres) and, per workgroup, the partial sums of dgamma / dbeta over its rows -> workspace [nb][2][D]
extern "C" int umr_layernorm_bwd_rows(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                      const void* dres, void* dx, void* workspace, int64_t workspace_bytes, int M, int D, int dtype,
                                      umr_stream_t stream) {
    UMR_CHECK_ARG(dy && x && gamma && mean && rstd && dx && workspace, "layernorm_bwd: null pointer");
    UMR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0 && D <= 256 * LN_MAXV, "layernorm_bwd: D must be a multiple of 8, <= 2048");
    UMR_CHECK_ARG(workspace_bytes >= umr_layernorm_bwd_workspace(M, D), "layernorm_bwd: workspace too small");
    int nb, rpb;
    ln_bwd_plan(M, &nb, &rpb);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)8 * D * 4;
    if (dtype == UMR_BF16 && D % 8 == 0 && D <= 1024)
        hipLaunchKernelGGL(ln_bwd_bf16x8_kernel<2>, dim3(nb), dim3(256), lds, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                           (const bf16_t*)dres, (bf16_t*)dx, (float*)workspace, M, D, rpb);
    else if (dtype == UMR_BF16 && D % 8 == 0 && D <= 2048)
        hipLaunchKernelGGL(ln_bwd_bf16x8_kernel<4>, dim3(nb), dim3(256), lds, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                           (const bf16_t*)dres, (bf16_t*)dx, (float*)workspace, M, D, rpb);
    else if (dtype == UMR_BF16)
        hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, dim3(nb), dim3(256), lds, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                           (const bf16_t*)dres, (bf16_t*)dx, (float*)workspace, M, D, rpb);
    else if (dtype == UMR_F32)
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nb), dim3(256), lds, s, (const float*)dy, (const float*)x, gamma, mean, rstd,
                           (const float*)dres, (float*)dx, (float*)workspace, M, D, rpb);
    else return umr_set_error(UMR_ERR_INVALID, "layernorm_bwd: dtype");
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

// parameter pass: dgamma / dbeta (+=, if accumulate) = sum of the row pass's partial sums.  A weight gradient: the caller may run it on
// another stream than the data-gradient chain (behind the row pass), as long as `workspace` stays untouched until it has run.
extern "C" int umr_layernorm_bwd_params(const void* workspace, int64_t workspace_bytes, float* dgamma, float* dbeta, int accumulate,
                                        int M, int D, umr_stream_t stream) {
    UMR_CHECK_ARG(workspace && dgamma && dbeta, "layernorm_bwd: null pointer");
    UMR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0 && D <= 256 * LN_MAXV, "layernorm_bwd: D must be a multiple of 8, <= 2048");
    UMR_CHECK_ARG(workspace_bytes >= umr_layernorm_bwd_workspace(M, D), "layernorm_bwd: workspace too small");
    int nb, rpb;
    ln_bwd_plan(M, &nb, &rpb);
    hipLaunchKernelGGL(ln_bwd_reduce, dim3((2 * D + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, dgamma, dbeta, nb, D,
                       accumulate);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, float* dgamma, float* dbeta, int accumulate, void* workspace,
                                 int64_t workspace_bytes, int M, int D, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(dgamma && dbeta, "layernorm_bwd: null pointer");
    const int rc = umr_layernorm_bwd_rows(dy, x, gamma, mean, rstd, dres, dx, workspace, workspace_bytes, M, D, dtype, stream);
    if (rc != UMR_OK) return rc;
    return umr_layernorm_bwd_params(workspace, workspace_bytes, dgamma, dbeta, accumulate, M, D, stream);
}
