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

// ---- backward kernels: ONE pass over the [M, C] tensor each.
// Work unit = a run of up to 64 consecutive pixels of one image row, handled by one wave.  The nine gradient values a
// pixel needs, g(q - t) (g = dout * act'(y), zero outside the image), are kept per run as nine registers whose lane l holds
// the value for pixel x0 + l (three rows x three column shifts, coalesced 4-byte loads of the tiny [M] maps); inside the
// run they are broadcast by v_readlane.  Each lane owns 4 of the C = 256 channels, so a pixel's channel vector is one
// coalesced 512-byte (bf16) access and every element of the big tensor is touched once (the first version walked the 9
// taps per output pixel and pulled every channel vector nine times through L1/L2: 10.6 + 6.6 ms at cfg2, against the
// 1.2 + 2.4 ms of the HBM traffic).
#ifndef LH_WEIGHT_UNROLL
#define LH_WEIGHT_UNROLL 4
#endif
#ifndef LH_DATA_UNROLL
#define LH_DATA_UNROLL 1
#endif
struct RunG { float g[3][3]; };   // [row dy = -1,0,1][column shift dx = -1,0,1], lane l <-> pixel x0 + l

__device__ __forceinline__ RunG load_run_g(const float* __restrict__ dout, const float* __restrict__ yout, int64_t rowbase, int py, int x0,
                                            int H, int W, int act, int lane) {
    RunG r;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = py + dy - 1;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = x0 + lane + dx - 1;
            float v = 0.f;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                const int64_t q = rowbase + (int64_t)(dy - 1) * W + xx;
                v = dout[q] * act_grad_from_out(yout ? yout[q] : 0.f, act);
            }
            r.g[dy][dx] = v;
        }
    }
    return r;
}

// dx[p][c] (+)= sum_t K[t][c] * g(p - off_t),  off_t = (t/3 - 1, t%3 - 1)
template <typename T>
__global__ __launch_bounds__(256) void lh_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ yout,
                                                          const float* __restrict__ kw, T* __restrict__ dx, int B, int H, int W, int C,
                                                          int act, int accumulate, int runs_per_row, int64_t total_runs) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = *(const f32x4*)(kw + t * C + lane * 4);
    for (int64_t run = (int64_t)blockIdx.x * 4 + wv; run < total_runs; run += (int64_t)gridDim.x * 4) {
        const int64_t row = run / runs_per_row;            // b * H + y
        const int x0 = (int)(run - row * runs_per_row) * 64;
        const int py = (int)(row % H);
        const int64_t rowbase = row * W;
        const int n = (W - x0) < 64 ? (W - x0) : 64;
        const RunG r = load_run_g(dout, yout, rowbase, py, x0, H, W, act, lane);
        T* o = dx + (rowbase + x0) * C + lane * 4;
        // blocked by hand for the same reason as lh_bwd_weight_kernel (convergent v_readlane blocks "#pragma unroll N")
        int i = 0;
        for (; i + LH_DATA_UNROLL <= n; i += LH_DATA_UNROLL) {
            f32x4 acc[LH_DATA_UNROLL];
#pragma unroll
            for (int u = 0; u < LH_DATA_UNROLL; ++u) {
                acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (accumulate) acc[u] = Vec4<T>::load(o + (int64_t)(i + u) * C);
            }
#pragma unroll
            for (int u = 0; u < LH_DATA_UNROLL; ++u) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    // tap t of output pixel q = p - off_t reads x(p): g at row py - (t/3 - 1), column px - (t%3 - 1)
                    const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.g[2 - t / 3][2 - t % 3]), i + u));
                    acc[u] += k[t] * g;
                }
                Vec4<T>::store(o + (int64_t)(i + u) * C, acc[u]);
            }
        }
        for (; i < n; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (accumulate) acc = Vec4<T>::load(o + (int64_t)i * C);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.g[2 - t / 3][2 - t % 3]), i));
                acc += k[t] * g;
            }
            Vec4<T>::store(o + (int64_t)i * C, acc);
        }
    }
}

// G[t][c] = sum_p g(p) x(p + off_t)[c] = sum_q x(q)[c] g(q - off_t);  n[t] = sum_q [q inside] g(q - off_t);  D = sum_p g(p)
// block partials [gridDim.x][9*C + 16] -> lh_reduce_kernel
template <typename T>
__global__ __launch_bounds__(256) void lh_bwd_weight_kernel(const T* __restrict__ x, const float* __restrict__ dout,
                                                            const float* __restrict__ yout, float* __restrict__ part, int B, int H, int W,
                                                            int C, int act, int runs_per_row, int64_t total_runs, int64_t runs_per_block) {
    extern __shared__ __attribute__((aligned(16))) float sh[];  // [9*C + 16]: the four waves add their partials in turn
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r0 = (int64_t)blockIdx.x * runs_per_block;
    const int64_t r1 = r0 + runs_per_block < total_runs ? r0 + runs_per_block : total_runs;
    f32x4 acc[9];
    float nt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; nt[t] = 0.f; }
    for (int64_t run = r0 + wv; run < r1; run += 4) {
        const int64_t row = run / runs_per_row;
        const int x0 = (int)(run - row * runs_per_row) * 64;
        const int py = (int)(row % H);
        const int64_t rowbase = row * W;
        const int n = (W - x0) < 64 ? (W - x0) : 64;
        const RunG r = load_run_g(dout, yout, rowbase, py, x0, H, W, act, lane);
        const T* xi = x + (rowbase + x0) * C + lane * 4;
        // n[t]: every lane l < n is one pixel q of the run; summed over lanes at the end
        if (lane < n) {
#pragma unroll
            for (int t = 0; t < 9; ++t) nt[t] += r.g[2 - t / 3][2 - t % 3];
        }
        // v_readlane is convergent, so the compiler refuses "#pragma unroll N" on a run-time trip count (a remainder loop
        // would duplicate it under new control flow) and the loop ran with ONE 512-byte load in flight per wave: 2.9 TB/s.
        // Blocked by hand: LH_WEIGHT_UNROLL loads issued back to back, then their 9 x U broadcasts and FMAs.
        int i = 0;
        for (; i + LH_WEIGHT_UNROLL <= n; i += LH_WEIGHT_UNROLL) {
            f32x4 v[LH_WEIGHT_UNROLL];
#pragma unroll
            for (int u = 0; u < LH_WEIGHT_UNROLL; ++u) v[u] = Vec4<T>::load(xi + (int64_t)(i + u) * C);
#pragma unroll
            for (int u = 0; u < LH_WEIGHT_UNROLL; ++u) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.g[2 - t / 3][2 - t % 3]), i + u));
                    acc[t] += v[u] * g;
                }
            }
        }
        for (; i < n; ++i) {
            const f32x4 v = Vec4<T>::load(xi + (int64_t)i * C);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r.g[2 - t / 3][2 - t % 3]), i));
                acc[t] += v * g;
            }
        }
    }
    // one slab of LDS (9.3 KB instead of 37 KB, so the block count is bounded by registers: 8 waves per SIMD keep twice the
    // bytes in flight of the four-slab version) -- waves 0..3 add in a fixed order, so the result is reproducible
    const int stride = 9 * C + 16;
    float ns[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) ns[t] = wave_sum(nt[t]);
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                f32x4* d = (f32x4*)(sh + t * C + lane * 4);
                *d = w ? *d + acc[t] : acc[t];
                if (lane == 0) sh[9 * C + t] = w ? sh[9 * C + t] + ns[t] : ns[t];
            }
        }
        __syncthreads();
    }
    float* o = part + (int64_t)blockIdx.x * stride;
    for (int i = threadIdx.x; i < 9 * C + 9; i += 256) o[i] = sh[i];
    if (threadIdx.x == 0) o[9 * C + 9] = sh[9 * C + 4];   // D = sum_p g(p): the centre tap is inside for every pixel
}

// The nine shifted gradient maps of a pixel as a 16-channel NHWC map: s9[q][t] = g(q - off_t) (t < 9; zero outside the image and for
// t >= 9), g = dout * act'(yout).  With them the two big reductions of the head's backward become linear maps of ONE small tensor:
//   G[t][c] = sum_q s9[q][t] x(q)[c],   dx(q)[c] = sum_t s9[q][t] K[t][c]
// and when x is itself a bilinear resize of a smaller map (x = U y: the head reads the x2-interpolated feature map, models.py:70-72),
// both move to the small map through the resize's adjoint applied to s9 (16 channels instead of 256):
//   G[t] = (U^T s9)[:, t]^T y,   dy = (U^T s9) K.
// n[t] = sum_q s9[q][t] and D = n[4] are summed here in f32 (block partials [gridDim.x][16] -> lh_reduce_kernel).
template <typename T>
__global__ __launch_bounds__(256) void lh_shift9_kernel(const float* __restrict__ dout, const float* __restrict__ yout, T* __restrict__ s9,
                                                        float* __restrict__ part, int B, int H, int W, int act) {
    __shared__ float sh[4][9];
    const int64_t M = (int64_t)B * H * W;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float nt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) nt[t] = 0.f;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < M; q += (int64_t)gridDim.x * 256) {
        const int64_t row = q / W;
        const int px = (int)(q - row * W), py = (int)(row % H);
        float g[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) g[t] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = py - (t / 3 - 1), xx = px - (t % 3 - 1);
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                const int64_t r = q - (int64_t)(t / 3 - 1) * W - (t % 3 - 1);
                g[t] = dout[r] * act_grad_from_out(yout ? yout[r] : 0.f, act);
            }
            nt[t] += g[t];
        }
        T* o = s9 + q * 16;
#pragma unroll
        for (int v = 0; v < 4; ++v) Vec4<T>::store(o + 4 * v, f32x4{g[4 * v], g[4 * v + 1], g[4 * v + 2], g[4 * v + 3]});
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float s_ = wave_sum(nt[t]);
        if (lane == 0) sh[wv][t] = s_;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int t = threadIdx.x;
        const int src = t < 9 ? t : 4;      // slot 9 = D = sum_p g(p) = n[4] (the centre tap is inside for every pixel)
        part[(int64_t)blockIdx.x * 16 + t] = (t <= 9) ? ((sh[0][src] + sh[1][src]) + (sh[2][src] + sh[3][src])) : 0.f;
    }
}

// Forward of the collapsed head when x is a bilinear resize of a smaller map (x = U y; the head reads the x2-interpolated feature map,
// models.py:70-72): kw[t] . x(p) = (U (y kw^T))(p)[t] -- a 1x1 product with the nine tap vectors commutes with the resize -- so the
// nine tap products are taken on the small map (a 16-column GEMM), resized as a 16-channel map T, and summed here over the taps'
// shifted positions:  out(q) = act( sum_{t: q + off_t inside} (T[q + off_t][t] + tapbias[t]) + tapbias[9] ).  The mirror of
// lh_shift9_kernel; the interpolated 256-channel map is never formed.
__global__ __launch_bounds__(256) void lh_gather9_kernel(const float* __restrict__ T, int64_t ldt, const float* __restrict__ tapbias,
                                                         float* __restrict__ out, int B, int H, int W, int act) {
    const int64_t M = (int64_t)B * H * W;
    float tb[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) tb[t] = tapbias[t];
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < M; q += (int64_t)gridDim.x * 256) {
        const int64_t row = q / W;
        const int px = (int)(q - row * W), py = (int)(row % H);
        float acc = 0.f, sb = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                acc += T[(q + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * ldt + t];
                sb += tb[t];
            }
        }
        out[q] = act_apply(acc + sb + tb[9], act);
    }
}

// block = 64 columns x 16 slab phases; phase y sums slabs y, y+16, ... (8 loads in flight), the 16 phase sums are added in
// a fixed order through LDS (one thread per column walking all slabs serially took 0.29 ms for 9.5 MB)
__global__ __launch_bounds__(1024) void lh_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nblocks, int n, int stride) {
    __shared__ float sh[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (i < n) {
        int b = ty;
        for (; b + 16 * 7 < nblocks; b += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + 16 * u) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblocks; b += 16) s += part[(int64_t)b * stride + i];
    }
    sh[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n) {
        float r = sh[0][tx];
#pragma unroll
        for (int y = 1; y < 16; ++y) r += sh[y][tx];
        out[i] = r;
    }
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

// long contractions (K >= 64: u = W4 W3, Vc = u W2, gu = W2 gv, ...): one wave per output, lanes split k -- a thread per
// output walks K serially at 18-512 threads in flight (0.68 ms for the two 512 x 4608 products)
__global__ __launch_bounds__(256) void small_gemm_wave_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, SmallGemm g) {
    const int lane = threadIdx.x & 63;
    const int64_t total = (int64_t)g.M * g.N;
    for (int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); idx < total; idx += (int64_t)gridDim.x * 4) {
        const int i = (int)(idx / g.N), j = (int)(idx - (int64_t)i * g.N);
        const float* a = A + i * g.sa_m;
        const float* b = B + j * g.sb_n;
        float s = 0.f;
        for (int k = lane; k < g.K; k += 64) s += a[k * g.sa_k] * b[k * g.sb_k];
        s = wave_sum(s);
        if (lane == 0) {
            float* c = C + i * g.sc_m + j * g.sc_n;
            *c = g.accumulate ? *c + s : s;
        }
    }
}

#ifndef LH_BLOCKS_CAP
#define LH_BLOCKS_CAP 2048
#endif
int lh_blocks(int64_t M) { int64_t nb = (M + 4095) / 4096; if (nb > LH_BLOCKS_CAP) nb = LH_BLOCKS_CAP; if (nb < 1) nb = 1; return (int)nb; }   // upper bound on partial slabs (8 blocks per CU)

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
    if (C != 256) return umr_set_error(UMR_ERR_UNSUPPORTED, "linear_head: C must be 256");
    const int rpr = (W + 63) / 64;
    const int64_t runs = (int64_t)B * H * rpr;
    int64_t g = (runs + 3) / 4;
    if (g > 8192) g = 8192;
    LH_DISPATCH(dtype, hipLaunchKernelGGL(lh_bwd_data_kernel<T>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, dout, yout, kw, (T*)dx, B, H, W, C, act, accumulate, rpr, runs));
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
    const int rpr = (W + 63) / 64;
    const int64_t runs = (int64_t)B * H * rpr;
    int nb = lh_blocks(M);
    if (nb > runs) nb = (int)runs;
    const int64_t rpb = (runs + nb - 1) / nb;
    nb = (int)((runs + rpb - 1) / rpb);
    const int stride = 9 * C + 16;
    const size_t lds = (size_t)stride * 4;
    hipStream_t s = (hipStream_t)stream;
    LH_DISPATCH(dtype, hipLaunchKernelGGL(lh_bwd_weight_kernel<T>, dim3(nb), dim3(256), lds, s, (const T*)x, dout, yout, (float*)workspace, B, H, W, C, act, rpr, runs, rpb));
    UMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(lh_reduce_kernel, dim3((9 * C + 10 + 63) / 64), dim3(1024), 0, s, (const float*)workspace, out, nb, 9 * C + 10, stride);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

static int lh_shift9_blocks(int64_t M) { const int64_t b = (M + 255) / 256; return (int)(b > 2048 ? 2048 : b); }
extern "C" int64_t umr_linear_head_shift9_workspace(int64_t M) { return (int64_t)lh_shift9_blocks(M) * 16 * 4; }

// s9: [B*H*W][16] of dtype (the nine shifted gradient maps, see lh_shift9_kernel); nd: [16] f32 = n[0..8], D, zeros
extern "C" int umr_linear_head_shift9(const float* dout, const float* yout, void* s9, float* nd, void* workspace, int64_t workspace_bytes,
                                      int B, int H, int W, int act, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(dout && s9 && nd && workspace && B > 0 && H > 0 && W > 0, "linear_head_shift9: bad arguments");
    UMR_CHECK_ARG(act != UMR_ACT_TANH || yout, "linear_head_shift9: tanh needs the forward output");
    if (act == 4) return umr_set_error(UMR_ERR_UNSUPPORTED, "linear_head: sine backward is not implemented");
    const int64_t M = (int64_t)B * H * W;
    UMR_CHECK_ARG(workspace_bytes >= umr_linear_head_shift9_workspace(M), "linear_head_shift9: workspace too small");
    const int nb = lh_shift9_blocks(M);
    hipStream_t s = (hipStream_t)stream;
    LH_DISPATCH(dtype, hipLaunchKernelGGL(lh_shift9_kernel<T>, dim3(nb), dim3(256), 0, s, dout, yout, (T*)s9, (float*)workspace, B, H, W, act));
    UMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(lh_reduce_kernel, dim3(1), dim3(1024), 0, s, (const float*)workspace, nd, nb, 16, 16);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_linear_head_gather9(const float* taps, int64_t ldt, const float* tapbias10, float* out, int B, int H, int W, int act,
                                       umr_stream_t stream) {
    UMR_CHECK_ARG(taps && tapbias10 && out && B > 0 && H > 0 && W > 0 && ldt >= 9, "linear_head_gather9: bad arguments");
    const int64_t M = (int64_t)B * H * W;
    int64_t g = (M + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(lh_gather9_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, taps, ldt, tapbias10, out, B, H, W, act);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_small_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t sa_m, int64_t sa_k,
                                  int64_t sb_k, int64_t sb_n, int64_t sc_m, int64_t sc_n, int accumulate, umr_stream_t stream) {
    UMR_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "small_gemm: bad arguments");
    SmallGemm g{M, N, K, sa_m, sa_k, sb_k, sb_n, sc_m, sc_n, accumulate};
    if (K >= 64) {
        int64_t nb = ((int64_t)M * N + 3) / 4;
        if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL(small_gemm_wave_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, A, B, C, g);
    } else {
        int64_t nb = ((int64_t)M * N + 255) / 256;
        if (nb > 8192) nb = 8192;
        hipLaunchKernelGGL(small_gemm_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, A, B, C, g);
    }
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
