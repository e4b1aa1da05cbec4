// LayerNorm forward / backward over rows of D channels (timm LayerNorm, eps 1e-6,
// biased variance; vit.py:196-199 via timm Block).  HBM-bound: one wave per row,
// 16-byte vector loads, statistics in f32, two-pass variance in registers.
#include "umr_common.h"

namespace {

constexpr int LN_MAXV = 8;  // up to 8 x 4 elements per lane -> D <= 2048

template <typename T>
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
            Vec4<T>::store(yr + c * 4, o);
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
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

// fixed-order sum of the per-block partials: 64 columns x 4 interleaved block slices per workgroup, four independent
// accumulators per thread, slices combined in index order (bitwise reproducible)
__global__ __launch_bounds__(256) void ln_bwd_reduce(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int nblocks, int D, int accumulate) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < 2 * D) {
        const int which = c / D, col = c - which * D;
        const float* q = part + (int64_t)which * D + col;
        const int64_t st = (int64_t)2 * D;
        int b = sl;
        for (; b + 12 < nblocks; b += 16) {
            s0 += q[(int64_t)b * st];
            s1 += q[(int64_t)(b + 4) * st];
            s2 += q[(int64_t)(b + 8) * st];
            s3 += q[(int64_t)(b + 12) * st];
        }
        for (; b < nblocks; b += 4) s0 += q[(int64_t)b * st];
    }
    sh[sl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && c < 2 * D) {
        const float s = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
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
    if (dtype == UMR_BF16) hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, eps);
    else if (dtype == UMR_F32) hipLaunchKernelGGL(ln_fwd_kernel<float>, g, b, 0, s, (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D, eps);
    else return umr_set_error(UMR_ERR_INVALID, "layernorm_fwd: dtype");
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int64_t umr_layernorm_bwd_workspace(int M, int D) {
    int nb = (M + 63) / 64;
    if (nb > 1024) nb = 1024;
    return (int64_t)nb * 2 * D * 4;
}

extern "C" int umr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                 const void* dres, void* dx, float* dgamma, float* dbeta, int accumulate, void* workspace,
                                 int64_t workspace_bytes, int M, int D, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, "layernorm_bwd: null pointer");
    UMR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0 && D <= 256 * LN_MAXV, "layernorm_bwd: D must be a multiple of 8, <= 2048");
    UMR_CHECK_ARG(workspace_bytes >= umr_layernorm_bwd_workspace(M, D), "layernorm_bwd: workspace too small");
    int nb = (M + 63) / 64;
    if (nb > 1024) nb = 1024;
    const int rpb = (M + nb - 1) / nb;
    nb = (M + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)8 * D * 4;
    if (dtype == UMR_BF16)
        hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, dim3(nb), dim3(256), lds, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd,
                           (const bf16_t*)dres, (bf16_t*)dx, (float*)workspace, M, D, rpb);
    else if (dtype == UMR_F32)
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nb), dim3(256), lds, s, (const float*)dy, (const float*)x, gamma, mean, rstd,
                           (const float*)dres, (float*)dx, (float*)workspace, M, D, rpb);
    else return umr_set_error(UMR_ERR_INVALID, "layernorm_bwd: dtype");
    UMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_bwd_reduce, dim3((2 * D + 63) / 64), dim3(256), 0, s, (const float*)workspace, dgamma, dbeta, nb, D, accumulate);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
