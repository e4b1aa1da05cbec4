// Kernels specific to the existence classifier (`Binary_Classifier` = torchvision
// ResNet-50 + Linear(1000,1) + sigmoid, models/objectness_net.py:205-223, called in eval mode on [<=128,3,128,128]
// crops by object_reasoning.py:491-523 / object_scoring.py:123-140).  Everything MFMA-shaped runs on umr_gemm_nt
// (1x1 convs as plain GEMMs, 3x3 convs as implicit GEMMs, the 7x7 stem through the im2col below); what is left is
// HBM-bound data movement:
//   umr_im2col_nchw   stem input: NCHW f32 image -> [B*Ho*Wo][ldk] rows, K order (c, ky, kx) = Conv2d weight order
//   umr_maxpool3x3s2  nn.MaxPool2d(3, stride 2, padding 1) on NHWC
//   umr_bn_fold       eval-mode BatchNorm folded into the preceding conv: w' = w * g/sqrt(var+eps), b' = beta - mean * g/sqrt(var+eps)
#include "umr_common.h"

namespace {

template <typename T> __device__ __forceinline__ T out_cvt(float v);
template <> __device__ __forceinline__ float out_cvt<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t out_cvt<bf16_t>(float v) { return (bf16_t)v; }

// one thread per (output pixel, c, ky): KW contiguous source pixels of one image row
template <typename T>
__global__ void im2col_nchw_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int C, int H, int W, int KH, int KW,
                                   int stride, int pad, int Ho, int Wo, int ldk) {
    const int64_t total = (int64_t)B * Ho * Wo * C * KH;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % Wo);
    int64_t r = idx / Wo;
    const int ky = (int)(r % KH); r /= KH;
    const int c = (int)(r % C); r /= C;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const int iy = oy * stride - pad + ky;
    const int64_t row = ((int64_t)b * Ho + oy) * Wo + ox;
    T* dst = out + row * ldk + (c * KH + ky) * KW;
    const bool yok = (unsigned)iy < (unsigned)H;
    const float* src = img + (((int64_t)b * C + c) * H + (yok ? iy : 0)) * W;
    for (int kx = 0; kx < KW; ++kx) {
        const int ix = ox * stride - pad + kx;
        dst[kx] = out_cvt<T>((yok && (unsigned)ix < (unsigned)W) ? src[ix] : 0.f);
    }
    if (c == C - 1 && ky == KH - 1)
        for (int k = C * KH * KW; k < ldk; ++k) out[row * ldk + k] = out_cvt<T>(0.f);
}

// NHWC, 4 channels per thread
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    const int c4 = C / 4;
    const int64_t total = (int64_t)B * Ho * Wo * c4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4) * 4;
        int64_t r = idx / c4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = Vec4<T>::load(x + (((int64_t)b * H + iy) * W + ix) * C + c);
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        Vec4<T>::store(y + (((int64_t)b * Ho + oy) * Wo + ox) * C + c, m);
    }
}

// one block per output channel; w [Co][K] f32 (any K), w_out [Co][ldk] T (tail zero), b_out [Co] f32
template <typename T>
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps, T* __restrict__ w_out,
                               float* __restrict__ b_out, int K, int ldk) {
    const int co = blockIdx.x;
    const float s = gamma[co] / sqrtf(var[co] + eps);
    for (int k = threadIdx.x; k < ldk; k += blockDim.x) w_out[(int64_t)co * ldk + k] = out_cvt<T>(k < K ? w[(int64_t)co * K + k] * s : 0.f);
    if (threadIdx.x == 0) b_out[co] = beta[co] - mean[co] * s;
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                   \
    if ((dtype) == UMR_BF16) { typedef bf16_t T; CALL; }          \
    else if ((dtype) == UMR_F32) { typedef float T; CALL; }       \
    else return umr_set_error(UMR_ERR_INVALID, "dtype");

extern "C" int umr_im2col_nchw(const float* images, void* out, int B, int C, int H, int W, int KH, int KW, int stride, int pad,
                               int ldk, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(images && out, "im2col: null pointer");
    UMR_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && ldk >= C * KH * KW, "im2col: bad geometry");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    UMR_CHECK_ARG(Ho > 0 && Wo > 0, "im2col: empty output");
    const int64_t total = (int64_t)B * Ho * Wo * C * KH;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL(im2col_nchw_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, images, (T*)out, B, C, H,
                                         W, KH, KW, stride, pad, Ho, Wo, ldk));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool3x3s2: bad arguments");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    int64_t g = (total + 255) / 256;
    if (g > 65536) g = 65536;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool3x3s2_kernel<T>, dim3((unsigned)g), dim3(256), 0, s, (const T*)x, (T*)y, B, H, W, C, Ho, Wo));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_bn_fold(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                           void* w_out, float* b_out, int Co, int K, int ldk, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(w && gamma && beta && mean && var && w_out && b_out && Co > 0 && K > 0 && ldk >= K, "bn_fold: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_fold_kernel<T>, dim3(Co), dim3(256), 0, s, w, gamma, beta, mean, var, eps, (T*)w_out, b_out, K, ldk));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
