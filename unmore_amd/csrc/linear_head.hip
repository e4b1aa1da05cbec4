// Collapsed form of a head WITHOUT non-linearities between its convolutions (the boundary-distance head's
// 'tanh' / 'sine' / None variants, models/objectness_net.py:119-142): 1x1 (C->512), 3x3 (512->512, zero padding),
// 1x1 (512->1024), 1x1 (1024->1) compose to ONE 3x3 convolution C->1 plus a border-dependent bias
//   z(p) = sum_{taps t with p+t inside the image} ( K[t] . x(p+t) + s[t] ) + c0,
//   K[t] = W1^T W2[t]^T W3^T W4^T,  s[t] = b1 . (W2[t]^T W3^T W4^T),  c0 = W4 W3 b2 + W4 b3 + b4
// (the first conv's bias does not propagate through the zero padding of the 3x3, hence the per-tap s[t]).
// Opt-in (engine `sdf_head_mode="collapsed"`): exact in real arithmetic, ~1300x fewer FLOPs for that head; every
// factored parameter still gets its exact gradient from three pixel-level reductions (this file) and a few tiny
// matrix products (engine).  HBM/L2-bound streaming kernels, coalesced over the channel dimension.
#include "umr_common.h"

namespace {

__device__ __forceinline__ float act_apply(float z, int act) { return act == UMR_ACT_TANH ? tanhf(z) : (act == 4 ? sinf(z) : z); }
// derivative expressed through the saved OUTPUT y (tanh only; identity otherwise)
__device__ __forceinline__ float act_grad_from_out(float y, int act) { return act == UMR_ACT_TANH ? (1.f - y * y) : 1.f; }

// ---- forward: one wave per output pixel, lanes split the C = 256 channels (4 each)
template <typename T>
__global__ __launch_bounds__(256) void lh_fwd_kernel(const T* __restrict__ x, const float* __restrict__ kw /* [9][C] */,
                                                     const float* __restrict__ tapbias /* [9] + constant */, float* __restrict__ out,
                                                     int B, int H, int W, int C, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t M = (int64_t)B * H * W;
    f32x4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = *(const f32x4*)(kw + t * C + lane * 4);
    for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
        const int64_t bhw = m / W;
        const int px = (int)(m - bhw * W), py = (int)(bhw % H);
        float acc = 0.f, sb = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const f32x4 v = Vec4<T>::load(x + (m + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * C + lane * 4);
                acc += v[0] * k[t][0] + v[1] * k[t][1] + v[2] * k[t][2] + v[3] * k[t][3];
                sb += tapbias[t];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) out[m] = act_apply(acc + sb + tapbias[9], act);
    }
}

// ---- backward, data: dx[p][c] (+)= sum_t K[t][c] * g(p - t),  g = dout * act'(y)
template <typename T>
__global__ __launch_bounds__(256) void lh_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ yout,
                                                          const float* __restrict__ kw, T* __restrict__ dx, int B, int H, int W, int C,
                                                          int act, int accumulate) {
    const int cv = C >> 2;
    const int64_t total = (int64_t)B * H * W * cv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv);
        const int64_t m = idx / cv;
        const int64_t bhw = m / W;
        const int px = (int)(m - bhw * W), py = (int)(bhw % H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // output pixel q = p - tap offset uses x(q + t) = x(p)
            const int qy = py - (t / 3 - 1), qx = px - (t % 3 - 1);
            if ((unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W) {
                const int64_t q = m - (int64_t)(t / 3 - 1) * W - (t % 3 - 1);
                const float g = dout[q] * act_grad_from_out(yout ? yout[q] : 0.f, act);
                acc += *(const f32x4*)(kw + t * C + c * 4) * g;
            }
        }
        T* o = dx + m * C + c * 4;
        if (accumulate) acc += Vec4<T>::load(o);
        Vec4<T>::store(o, acc);
    }
}

// ---- backward, weights: G[t][c] = sum_p g(p) x(p+t)[c];  n[t] = sum_{p: p+t inside} g(p);  D = sum_p g(p)
// block partials [gridDim.x][9*C + 16] -> lh_reduce_kernel
template <typename T>
__global__ __launch_bounds__(256) void lh_bwd_weight_kernel(const T* __restrict__ x, const float* __restrict__ dout,
                                                            const float* __restrict__ yout, float* __restrict__ part, int B, int H, int W,
                                                            int C, int act, int64_t pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [4 waves][9*C + 16]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t M = (int64_t)B * H * W;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < M ? p0 + pix_per_block : M;
    f32x4 acc[9];
    float nt[9];
    float dsum = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; nt[t] = 0.f; }
    for (int64_t m = p0 + wv; m < p1; m += 4) {
        const int64_t bhw = m / W;
        const int px = (int)(m - bhw * W), py = (int)(bhw % H);
        const float g = dout[m] * act_grad_from_out(yout ? yout[m] : 0.f, act);
        dsum += g;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const f32x4 v = Vec4<T>::load(x + (m + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * C + lane * 4);
                acc[t] += v * g;
                nt[t] += g;
            }
        }
    }
    const int stride = 9 * C + 16;
    float* mine = sh + wv * stride;
#pragma unroll
    for (int t = 0; t < 9; ++t) *(f32x4*)(mine + t * C + lane * 4) = acc[t];
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) mine[9 * C + t] = nt[t];
        mine[9 * C + 9] = dsum;
    }
    __syncthreads();
    float* o = part + (int64_t)blockIdx.x * stride;
    for (int i = threadIdx.x; i < 9 * C + 10; i += 256) o[i] = (sh[i] + sh[stride + i]) + (sh[2 * stride + i] + sh[3 * stride + i]);
}

__global__ void lh_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nblocks, int n, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(int64_t)b * stride + i];
    out[i] = s;
}

// tiny strided f32 matrix product for the weight algebra of the collapsed head (all operands <= a few MB):
// C[i*sc_m + j*sc_n] (=|+=) sum_k A[i*sa_m + k*sa_k] * B[k*sb_k + j*sb_n]
struct SmallGemm { int M, N, K; int64_t sa_m, sa_k, sb_k, sb_n, sc_m, sc_n; int accumulate; };
__global__ void small_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, SmallGemm g) {
    const int64_t total = (int64_t)g.M * g.N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / g.N), j = (int)(idx - (int64_t)i * g.N);
        const float* a = A + i * g.sa_m;
        const float* b = B + j * g.sb_n;
        float s = 0.f;
        for (int k = 0; k < g.K; ++k) s += a[k * g.sa_k] * b[k * g.sb_k];
        float* c = C + i * g.sc_m + j * g.sc_n;
        *c = g.accumulate ? *c + s : s;
    }
}

int lh_blocks(int64_t M) { int64_t nb = (M + 4095) / 4096; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1; return (int)nb; }

}  // namespace

#define LH_DISPATCH(dtype, CALL)                                  \
    if ((dtype) == UMR_BF16) { typedef bf16_t T; CALL; }          \
    else if ((dtype) == UMR_F32) { typedef float T; CALL; }       \
    else return umr_set_error(UMR_ERR_INVALID, "dtype");

extern "C" int umr_linear_head_fwd(const void* x, const float* kw, const float* tapbias10, float* out, int B, int H, int W, int C,
                                   int act, int dtype, umr_stream_t stream) {
    const float* tapbias = tapbias10;
    UMR_CHECK_ARG(x && kw && tapbias && out && B > 0 && H > 0 && W > 0, "linear_head_fwd: bad arguments");
    if (C != 256) return umr_set_error(UMR_ERR_UNSUPPORTED, "linear_head: C must be 256");
    const int64_t M = (int64_t)B * H * W;
    int64_t g = (M + 3) / 4;
    if (g > 16384) g = 16384;
    LH_DISPATCH(dtype, hipLaunchKernelGGL(lh_fwd_kernel<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const T*)x, kw, tapbias, out, B, H, W, C, act));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_linear_head_bwd_data(const float* dout, const float* yout, const float* kw, void* dx, int B, int H, int W, int C,
                                        int act, int accumulate, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(dout && kw && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "linear_head_bwd_data: bad arguments");
    UMR_CHECK_ARG(act != UMR_ACT_TANH || yout, "linear_head_bwd_data: tanh needs the forward output");
    if (act == 4) return umr_set_error(UMR_ERR_UNSUPPORTED, "linear_head: sine backward is not implemented");
    const int64_t total = (int64_t)B * H * W * (C / 4);
    int64_t g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    LH_DISPATCH(dtype, hipLaunchKernelGGL(lh_bwd_data_kernel<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dout, yout, kw, (T*)dx, B, H, W, C, act, accumulate));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int64_t umr_linear_head_bwd_weight_workspace(int64_t M, int C) { return (int64_t)lh_blocks(M) * (9 * C + 16) * 4; }

// out = [G (9*C) | n (9) | D (1)] f32
extern "C" int umr_linear_head_bwd_weight(const void* x, const float* dout, const float* yout, float* out, void* workspace,
                                          int64_t workspace_bytes, int B, int H, int W, int C, int act, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(x && dout && out && workspace && B > 0 && H > 0 && W > 0, "linear_head_bwd_weight: bad arguments");
    if (C != 256) return umr_set_error(UMR_ERR_UNSUPPORTED, "linear_head: C must be 256");
    UMR_CHECK_ARG(act != UMR_ACT_TANH || yout, "linear_head_bwd_weight: tanh needs the forward output");
    if (act == 4) return umr_set_error(UMR_ERR_UNSUPPORTED, "linear_head: sine backward is not implemented");
    const int64_t M = (int64_t)B * H * W;
    UMR_CHECK_ARG(workspace_bytes >= umr_linear_head_bwd_weight_workspace(M, C), "linear_head_bwd_weight: workspace too small");
    int nb = lh_blocks(M);
    const int64_t ppb = (M + nb - 1) / nb;
    nb = (int)((M + ppb - 1) / ppb);
    const int stride = 9 * C + 16;
    const size_t lds = (size_t)4 * stride * 4;
    hipStream_t s = (hipStream_t)stream;
    LH_DISPATCH(dtype, hipLaunchKernelGGL(lh_bwd_weight_kernel<T>, dim3(nb), dim3(256), lds, s, (const T*)x, dout, yout, (float*)workspace, B, H, W, C, act, ppb));
    UMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(lh_reduce_kernel, dim3((9 * C + 10 + 255) / 256), dim3(256), 0, s, (const float*)workspace, out, nb, 9 * C + 10, stride);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_small_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t sa_m, int64_t sa_k,
                                  int64_t sb_k, int64_t sb_n, int64_t sc_m, int64_t sc_n, int accumulate, umr_stream_t stream) {
    UMR_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "small_gemm: bad arguments");
    SmallGemm g{M, N, K, sa_m, sa_k, sb_k, sb_n, sc_m, sc_n, accumulate};
    int64_t nb = ((int64_t)M * N + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, A, B, C, g);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
