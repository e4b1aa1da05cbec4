// HBM-bound helper kernels of the ObjectnessNet path: patch extraction, bilinear
// resize (forward + exact adjoint), ConvTranspose pixel (un)shuffle, stride-2 zero
// stuffing, weight packing (permute / flip / cast), small reductions, the 1024->{1,2}
// head output layer, the fused 4-term loss (value + gradient in one pass), Adam.
// All are coalesced over the channel (innermost) dimension with 16-byte accesses
// where the layout allows; no atomics (every reduction is a fixed-order two-stage sum).
#include "umr_common.h"
#include "gemm_epilogue.h"   // x3_planes_store8

namespace {

template <typename T> __device__ __forceinline__ T cvt_out(float v) { return from_f32<T>(v); }

// ---------------------------------------------------------------- patchify
// images [B,3,H,W] f32 (NCHW) -> rows [B*gh*gw][ldk] T, K order (c, py, px); tail (ldk > 3*p*p) zero.
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int p, int gh, int gw,
                                int ldk) {
    const int64_t total = (int64_t)B * gh * gw * 3 * p;  // one thread per (patch, c, py): p contiguous pixels
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // order: gx fastest among patches so that a wave reads consecutive p-pixel segments of one image row
    const int gx = (int)(idx % gw);
    int64_t r = idx / gw;
    const int py = (int)(r % p); r /= p;
    const int c = (int)(r % 3); r /= 3;
    const int gy = (int)(r % gh);
    const int b = (int)(r / gh);
    const float* src = img + (((int64_t)b * 3 + c) * H + gy * p + py) * W + gx * p;
    T* dst = out + ((int64_t)(b * gh + gy) * gw + gx) * ldk + (c * p + py) * p;
    for (int x = 0; x < p; ++x) dst[x] = cvt_out<T>(src[x]);
    if (c == 2 && py == p - 1) for (int k = 3 * p * p; k < ldk; ++k) out[((int64_t)(b * gh + gy) * gw + gx) * ldk + k] = cvt_out<T>(0.f);
}

// ---------------------------------------------------------------- bilinear resize, NHWC
__device__ __forceinline__ void src_index(int o, float scale, int align, int in, int& i0, int& i1, float& l0, float& l1) {
    float s = align ? scale * (float)o : fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}
__device__ __forceinline__ float area_scale(int in, int out, int align) {
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

// NV f32x4 vectors (4*NV channels) per thread: NV = 2 gives 16-byte bf16 accesses and half the index arithmetic
// ldx / ldy: elements between consecutive pixels of x / y (>= C: the maps may be column slices of wider row-major buffers);
// relu: max(., 0) on the result; PLANES (T = float, NV = 2): the result is written as three bf16 planes [h(C) | m(C) | l(C)] per pixel
// (ldy counts bf16 elements), the operand format of the plane GEMMs
template <typename T, int NV, bool PLANES = false>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, void* __restrict__ yv, int B, int Hi, int Wi, int Ho, int Wo, int C, int align,
                                    int64_t ldx, int64_t ldy, int relu) {
    // grid.y = (b, oy) pairs (strided), grid.x x 256 threads = the (ox, channel-vector) pairs of one output row: the
    // per-thread index arithmetic is one 32-bit division (the flat 64-bit form spent more time dividing than loading)
    const int cv = C / (4 * NV);
    const int rowlen = Wo * cv;
    const float sh = area_scale(Hi, Ho, align), sw = area_scale(Wi, Wo, align);
    // optional ReLU as a compare-and-select: NaN activations stay NaN (fmaxf(NaN, x) returns x and would hide a diverged run)
    const bool relu_ = relu != 0;
    for (int by = blockIdx.y; by < B * Ho; by += gridDim.y) {
        const int b = by / Ho, oy = by - b * Ho;
        int y0, y1;
        float ly0, ly1;
        src_index(oy, sh, align, Hi, y0, y1, ly0, ly1);
        const T* r0 = x + ((int64_t)b * Hi + y0) * Wi * ldx;
        const T* r1 = x + ((int64_t)b * Hi + y1) * Wi * ldx;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rowlen; i += gridDim.x * blockDim.x) {
            const int ox = i / cv, c = (i - ox * cv) * 4 * NV;
            int x0, x1;
            float lx0, lx1;
            src_index(ox, sw, align, Wi, x0, x1, lx0, lx1);
            const T* p00 = r0 + (int64_t)x0 * ldx + c;
            const T* p01 = r0 + (int64_t)x1 * ldx + c;
            const T* p10 = r1 + (int64_t)x0 * ldx + c;
            const T* p11 = r1 + (int64_t)x1 * ldx + c;
            f32x4 o[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const f32x4 v00 = Vec4<T>::load(p00 + 4 * v), v01 = Vec4<T>::load(p01 + 4 * v);
                const f32x4 v10 = Vec4<T>::load(p10 + 4 * v), v11 = Vec4<T>::load(p11 + 4 * v);
                for (int j = 0; j < 4; ++j)
                    {
                    const float r_ = ly0 * (lx0 * v00[j] + lx1 * v01[j]) + ly1 * (lx0 * v10[j] + lx1 * v11[j]);
                    o[v][j] = (relu_ && r_ < 0.f) ? 0.f : r_;
                }
            }
            if constexpr (PLANES) {
                static_assert(NV == 2, "plane output works on 8 channels per thread");
                x3_planes_store8((bf16_t*)yv + ((int64_t)by * Wo + ox) * ldy + c, C, o[0], o[1]);
            } else {
                T* yr = (T*)yv + ((int64_t)by * Wo + ox) * ldy + c;
#pragma unroll
                for (int v = 0; v < NV; ++v) Vec4<T>::store(yr + 4 * v, o[v]);
            }
        }
    }
}

// exact adjoint by gathering: for input pixel (iy,ix) visit the few output pixels whose taps touch it.  Output o reads at
// s(o) = scale * o (align_corners) or scale * (o + 0.5) - 0.5 clamped at 0; it touches input i when s lies in (i - 1, i + 1) -- plus,
// without align_corners, every o whose s was clamped to the first / last input pixel (round 4: the half-pixel offsets were missing
// from this range and strongly up-scaling resizes without align_corners lost contributors; tests/test_commute_gpu.py)
__device__ __forceinline__ void out_range(int i, float scale, int in, int out, int align, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
    const float off = align ? 0.f : 0.5f;
    lo = (int)floorf(((float)i - 1.f + off) / scale - off) - 1;
    hi = (int)ceilf(((float)i + 1.f + off) / scale - off) + 1;
    if (lo < 0 || i == 0) lo = 0;
    if (hi > out - 1 || i == in - 1) hi = out - 1;
}

template <typename T, int NV>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int align,
                                    int64_t lddy, int64_t lddx) {
    // same 2-D decomposition as the forward: grid.y = (b, iy), grid.x covers the (ix, channel-vector) pairs of one input row
    const int cv = C / (4 * NV);
    const int rowlen = Wi * cv;
    const float sh = area_scale(Hi, Ho, align), sw = area_scale(Wi, Wo, align);
    for (int by = blockIdx.y; by < B * Hi; by += gridDim.y) {
        const int b = by / Hi, iy = by - b * Hi;
        int ylo, yhi;
        out_range(iy, sh, Hi, Ho, align, ylo, yhi);
        const T* base = dy + (int64_t)b * Ho * Wo * lddy;
        T* xr = dx + (int64_t)by * Wi * lddx;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rowlen; i += gridDim.x * blockDim.x) {
            const int ix = i / cv, c = (i - ix * cv) * 4 * NV;
            int xlo, xhi;
            out_range(ix, sw, Wi, Wo, align, xlo, xhi);
            f32x4 acc[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int oy = ylo; oy <= yhi; ++oy) {
                int y0, y1; float ly0, ly1;
                src_index(oy, sh, align, Hi, y0, y1, ly0, ly1);
                const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
                if (wy == 0.f) continue;
                for (int ox = xlo; ox <= xhi; ++ox) {
                    int x0, x1; float lx0, lx1;
                    src_index(ox, sw, align, Wi, x0, x1, lx0, lx1);
                    const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
                    if (wx == 0.f) continue;
                    const T* gp = base + ((int64_t)oy * Wo + ox) * lddy + c;
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v] += Vec4<T>::load(gp + 4 * v) * (wy * wx);
                }
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) Vec4<T>::store(xr + (int64_t)ix * lddx + c + 4 * v, acc[v]);
        }
    }
}

// bf16 with C % 8 == 0: the same two kernels on 16-byte accesses (one bf16x8 per tap instead of two bf16x4: half the vector-memory
// instructions; the generic forms above reach 3.4 / 2.9 TB/s on the final x2 resize of the feature map, tools/bilinear_bench.py)
__device__ __forceinline__ void ld_bf16x8(const bf16_t* p, float (&f)[8]) {
    const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)t[j];
}
// forward, streaming over a run of output rows (round 4): a thread owns the column (ox, 8 channels) and keeps the x-interpolated vectors
// of the two input rows the current output row blends; when the upper row moves on by one (every second output row at a x2 resize) the
// lower vector becomes the upper one and ONE new input row is needed -- its two taps were requested when the previous row was
// taken.  Two loads per new input row instead of sixteen per four output rows (most of them repeats that cost load-unit slots): the
// kernel is left with its stores.  Same expression as the gather form (x first, then y).
__global__ __launch_bounds__(256) void bilinear_fwd_bf16x8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int Hi, int Wi, int Ho,
                                                                int Wo, int C, int align, int64_t ldx, int64_t ldy, int relu, int RUN) {
    // optional ReLU as a compare-and-select: NaN activations stay NaN (fmaxf(NaN, x) returns x and would hide a diverged run)
    const bool relu_ = relu != 0;
    const int cv = C >> 3;
    const int rowlen = Wo * cv;
    const float sh = area_scale(Hi, Ho, align), sw = area_scale(Wi, Wo, align);
    const int runs_per_image = (Ho + RUN - 1) / RUN;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rowlen; i += gridDim.x * blockDim.x) {
        const int ox = i / cv, c = (i - ox * cv) * 8;
        int x0, x1;
        float lx0, lx1;
        src_index(ox, sw, align, Wi, x0, x1, lx0, lx1);
        const int64_t o0 = (int64_t)x0 * ldx + c, o1 = (int64_t)x1 * ldx + c;
        for (int run = blockIdx.y; run < B * runs_per_image; run += gridDim.y) {
            const int b = run / runs_per_image, oy0 = (run - b * runs_per_image) * RUN;
            const int oy1 = min(Ho, oy0 + RUN);
            const bf16_t* xb = x + (int64_t)b * Hi * Wi * ldx;
            auto row_taps = [&](int yy, bf16x8& a, bf16x8& bb) {
                const bf16_t* r = xb + (int64_t)(yy < Hi ? yy : Hi - 1) * Wi * ldx;
                a = *(const bf16x8*)(r + o0);
                bb = *(const bf16x8*)(r + o1);
            };
            auto xin = [&](const bf16x8& a, const bf16x8& bb, float (&v)[8]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = lx0 * (float)a[j] + lx1 * (float)bb[j];
            };
            float vA[8], vB[8];
            bf16x8 na, nb;            // taps of input row cury + 2, in flight
            int cury = -4;
            for (int oy = oy0; oy < oy1; ++oy) {
                int y0, y1;
                float l0, l1;
                src_index(oy, sh, align, Hi, y0, y1, l0, l1);
                if (y0 != cury) {     // block-uniform
                    if (y0 == cury + 1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) vA[j] = vB[j];
                        xin(na, nb, vB);                       // input row cury + 2 = y0 + 1 (clamped = y1)
                    } else {
                        bf16x8 a0, b0, a1, b1;
                        row_taps(y0, a0, b0);
                        row_taps(y0 + 1, a1, b1);
                        xin(a0, b0, vA);
                        xin(a1, b1, vB);
                    }
                    cury = y0;
                    row_taps(y0 + 2, na, nb);
                }
                // y1 == y0 only at the clamped last row, where vB was built from the same (clamped) row: the same value either way
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float r_ = l0 * vA[j] + l1 * vB[j];
                    o[j] = (bf16_t)((relu_ && r_ < 0.f) ? 0.f : r_);
                }
                *(bf16x8*)(y + (((int64_t)b * Ho + oy) * Wo + ox) * ldy + c) = o;
            }
        }
    }
}
// adjoint: the output pixels that touch input pixel i lie in [ (i-1)/s , (i+1)/s ) -- at most 2/s + 2 candidates per axis (8 for
// s >= 0.4; the host picks this kernel only then).  Their weights are computed ONCE per thread and axis (the generic kernel
// re-derives the x weights for every candidate row and scans 7 x 7 candidates at s = 0.5: index arithmetic, not memory, bound it
// at 2.9 TB/s).
// NCX = x candidates per output row (6 covers scales >= 0.49: <= 5 contributors + the one-early window start; 8 down to 0.4).
// The candidate rows are skipped by a block-uniform branch (iy is the block's); inside a row ALL NCX taps are loaded
// unconditionally (zero weight / clamped address where a tap does not contribute) so that the NCX loads go out back to back --
// a per-tap `continue` left one load in flight per thread (3.3 TB/s on the final x2 resize).
template <int NCX>
__global__ __launch_bounds__(256) void bilinear_bwd_bf16x8_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B, int Hi, int Wi, int Ho,
                                                                int Wo, int C, int align, int64_t lddy, int64_t lddx, int RUN) {
    constexpr int NC = 8;
    const int cv = C >> 3;
    const int rowlen = Wi * cv;
    const float sh = area_scale(Hi, Ho, align), sw = area_scale(Wi, Wo, align);
    auto weights = [&](int i, float sc, int in, int out, int& lo, float (&wv)[NC]) {
        lo = (int)floorf(((float)i - 1.f) / sc - 1e-3f);
        if (lo < 0) lo = 0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int o = lo + k;
            int i0, i1; float l0, l1;
            src_index(o < out ? o : out - 1, sc, align, in, i0, i1, l0, l1);
            wv[k] = (o < out) ? ((i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f)) : 0.f;
        }
    };
    // a thread's column (ix, 8 channels) is the same for every row it visits: the x weights are computed once per thread
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rowlen; i += gridDim.x * blockDim.x) {
        const int ix = i / cv, c = (i - ix * cv) * 8;
        int xlo;
        float wx[NC];
        weights(ix, sw, Wi, Wo, xlo, wx);
        int xoff[NCX];       // element offsets of the NCX candidate columns, clamped inside the row
#pragma unroll
        for (int kx = 0; kx < NCX; ++kx) xoff[kx] = ((xlo + kx < Wo ? xlo + kx : Wo - 1) - xlo) * (int)lddy;
        // Streaming over the OUTPUT rows (round 4).  An output row touches two neighbouring input rows, so gathering per input row
        // read every output row twice (three candidate rows per input row at a x2 resize) -- and the second read came from HBM again:
        // a CU's resident blocks keep ~0.6 MB of rows in flight, 19 MB per XCD against 4 MB of L2 (the 512-channel adjoint of the head
        // layer fetched 14.4 GB for a 9.7-GB operand).  Here a block owns a run of input rows [iy0, iy1) of one image and walks the
        // output rows that touch it ONCE, in order: row oy adds l0 * v to input row y0(oy) and l1 * v to y0 + 1 (v = the x-gathered
        // row), y0 never decreases, so two accumulators are live and a finished input row is stored when y0 moves past it.  All
        // control flow is block-uniform.  Only the run's first and last output rows are read by the neighbouring run as well.
        const int runs_per_image = (Hi + RUN - 1) / RUN;
        for (int run = blockIdx.y; run < B * runs_per_image; run += gridDim.y) {
            const int b = run / runs_per_image, iy0 = (run - b * runs_per_image) * RUN;
            const int iy1 = min(Hi, iy0 + RUN);
            // output rows whose y0 lies in [iy0 - 1, iy1 - 1]
            int oy_lo = (sh > 0.f) ? (int)floorf(((float)iy0 - 1.f) / sh - 1e-3f) - 1 : 0;
            int oy_hi = (sh > 0.f) ? (int)ceilf((float)iy1 / sh + 1e-3f) + 1 : Ho - 1;
            if (oy_lo < 0) oy_lo = 0;
            if (oy_hi > Ho - 1) oy_hi = Ho - 1;
            const bf16_t* base = dy + (int64_t)b * Ho * Wo * lddy + (int64_t)xlo * lddy + c;
            bf16_t* xcol = dx + ((int64_t)b * Hi * Wi + ix) * lddx + c;
            float accA[8], accB[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { accA[j] = 0.f; accB[j] = 0.f; }
            int cur = iy0 - 1;                       // the input row accA belongs to (accB: cur + 1)
            auto retire = [&]() {                    // accA is complete: store it if the row is ours, then shift
                if (cur >= iy0 && cur < iy1) {
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)accA[j];
                    *(bf16x8*)(xcol + (int64_t)cur * Wi * lddx) = o;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { accA[j] = accB[j]; accB[j] = 0.f; }
                ++cur;
            };
            bf16x8 t[NCX], tn[NCX];
#pragma unroll
            for (int kx = 0; kx < NCX; ++kx) tn[kx] = *(const bf16x8*)(base + (int64_t)oy_lo * Wo * lddy + xoff[kx]);
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
#pragma unroll
                for (int kx = 0; kx < NCX; ++kx) t[kx] = tn[kx];
                if (oy < oy_hi) {                    // the next row's taps go out before this row is used
                    const bf16_t* rowp = base + (int64_t)(oy + 1) * Wo * lddy;
#pragma unroll
                    for (int kx = 0; kx < NCX; ++kx) tn[kx] = *(const bf16x8*)(rowp + xoff[kx]);
                }
                int y0, y1;
                float l0, l1;
                src_index(oy, sh, align, Hi, y0, y1, l0, l1);
                if (y0 < cur) continue;              // a row before the run (block-uniform)
                while (y0 > cur && cur < iy1) retire();
                if (cur >= iy1) break;               // everything of the run is stored
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
                for (int kx = 0; kx < NCX; ++kx) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += (float)t[kx][j] * wx[kx];
                }
                if (y1 == y0) l0 += l1;              // clamped at the last input row: both taps are that row
#pragma unroll
                for (int j = 0; j < 8; ++j) { accA[j] += l0 * v[j]; if (y1 != y0) accB[j] += l1 * v[j]; }
            }
            while (cur < iy1) retire();
        }
    }
}

// ---------------------------------------------------------------- pixel shuffle (ConvTranspose k == stride)
// src [B*H*W][s*s*C] (column = (i*s+j)*C + c)  <->  dst [B, H*s, W*s, C];  inverse=1 gathers dst -> src layout
template <typename T>
__global__ void pixel_shuffle_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int s, int C, int inverse) {
    const int cv = C >> 2;
    const int64_t total = (int64_t)B * H * W * s * s * cv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv);
        int64_t r = idx / cv;
        const int ij = (int)(r % (s * s)); r /= (s * s);
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        const int i = ij / s, j = ij - i * s;
        const int64_t a = idx * 4;  // [m][(i,j,c)] layout
        const int64_t d = ((((int64_t)b * H * s + y * s + i) * W * s) + x * s + j) * C + c * 4;
        if (inverse) Vec4<T>::store(dst + a, Vec4<T>::load(src + d));
        else Vec4<T>::store(dst + d, Vec4<T>::load(src + a));
    }
}

// dY [B,Ho,Wo,C] -> zero-stuffed [B,H,W,C] with out[2oy,2ox] = dY[oy,ox] (stride-2 conv data gradient)
template <typename T>
__global__ void zero_stuff2_kernel(const T* __restrict__ dy, T* __restrict__ out, int B, int H, int W, int Ho, int Wo, int C) {
    const int cv = C >> 2;
    const int64_t total = (int64_t)B * H * W * cv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv);
        int64_t r = idx / cv;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(y & 1) && !(x & 1) && (y >> 1) < Ho && (x >> 1) < Wo)
            v = Vec4<T>::load(dy + (((int64_t)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c * 4);
        Vec4<T>::store(out + idx * 4, v);
    }
}

// ---------------------------------------------------------------- generic 4-D permute (+flip) with cast
struct PermDesc { int d[4]; int64_t sstride[4]; int64_t soff; };  // dst dims, src stride per dst dim (may be negative), src offset
template <typename TI, typename TO>
__global__ void permute4_kernel(const TI* __restrict__ src, TO* __restrict__ dst, PermDesc pd, int accumulate) {
    const int64_t total = (int64_t)pd.d[0] * pd.d[1] * pd.d[2] * pd.d[3];
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx;
        const int i3 = (int)(r % pd.d[3]); r /= pd.d[3];
        const int i2 = (int)(r % pd.d[2]); r /= pd.d[2];
        const int i1 = (int)(r % pd.d[1]);
        const int i0 = (int)(r / pd.d[1]);
        const float v = to_f32<TI>(src[pd.soff + i0 * pd.sstride[0] + i1 * pd.sstride[1] + i2 * pd.sstride[2] + i3 * pd.sstride[3]]);
        if (accumulate) dst[idx] = from_f32<TO>(to_f32<TO>(dst[idx]) + v);
        else dst[idx] = from_f32<TO>(v);
    }
}

// Many permutes in ONE launch (the per-step refresh of every kernel-layout weight copy after the optimizer step: ~180 launches
// of a few microseconds each otherwise).  table[e] describes one permute and the first block that works on it; a block finds
// its entry by binary search.  Two block shapes: linear (e[3] == 0: 2048 consecutive destination elements, for permutes whose
// innermost destination dimension is contiguous in the source too -- casts) and TILED (a hyper-rectangle e[0..3] of the
// destination index space, <= 4608 elements): the tile is read in SOURCE-address order (ord[] = dimensions by ascending
// |source stride|) into LDS at its destination-linear position and written out in destination order, so that both the
// gather and the store are runs of >= 128 bytes.  The conv-weight transposes ([co,ci,3,3] -> [co][ky][kx][ci] and the flipped
// [ci][ky][kx][co] data-gradient form) read single floats 36 B / 18 KB apart in the linear shape: 11.4 GB of HBM-side traffic
// per refresh for 0.7 GB of weights (1.7 ms); tiled: see DESIGN.md section 5.
struct PermEntry { const void* src; void* dst; int32_t d[4]; int64_t sstride[4]; int64_t soff; int32_t dtype_in, dtype_out; int64_t blk_start;
                   int32_t e[4]; int32_t ord[4]; int64_t rowlen; };
// destination element di of an entry: f32 / bf16 at dst[di], or -- dtype_out UMR_BF16X3, the weights of the fp32-grade plane GEMMs --
// the three bf16 planes of the value in row di / rowlen of [rows][h(rowlen) | m(rowlen) | l(rowlen)]
__device__ __forceinline__ void perm_store(const PermEntry& e, int64_t di, float v) {
    if (e.dtype_out == UMR_F32) ((float*)e.dst)[di] = v;
    else if (e.dtype_out == UMR_BF16) ((bf16_t*)e.dst)[di] = (bf16_t)v;
    else {
        const int64_t row = di / e.rowlen;
        bf16_t* q = (bf16_t*)e.dst + row * 3 * e.rowlen + (di - row * e.rowlen);
        const bf16_t h = (bf16_t)v;
        const float r1 = v - (float)h;
        const bf16_t m = (bf16_t)r1;
        q[0] = h; q[e.rowlen] = m; q[2 * e.rowlen] = (bf16_t)(r1 - (float)m);
    }
}
static_assert(sizeof(PermEntry) == sizeof(umr_perm_entry), "umr_perm_entry layout");
constexpr int PERM_TILE_MAX = 4608;
constexpr int PERM_LDS_FLOATS = PERM_TILE_MAX;   // tile rows of e[3] floats are padded by one: e[0] e[1] e[2] (e[3] + 1) <= 4608
__global__ __launch_bounds__(256) void permute4_batched_kernel(const PermEntry* __restrict__ table, int n, const int32_t* __restrict__ blk_entry) {
    __shared__ float tile[PERM_LDS_FLOATS];
    const int64_t b = blockIdx.x;
    int lo = 0;
    if (blk_entry) {
        lo = blk_entry[b];     // one load instead of a dependent chain of ~8 (a block moves only 8-18 KB: the search was a third of its time)
    } else {
        int hi = n - 1;
        while (lo < hi) {   // last entry with blk_start <= b
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid].blk_start <= b) lo = mid; else hi = mid - 1;
        }
    }
    const PermEntry e = table[lo];
    const int64_t lb = b - e.blk_start;
    if (e.e[3] == 0) {
        const int64_t total = (int64_t)e.d[0] * e.d[1] * e.d[2] * e.d[3];
        // a plain cast (the Linear weights: most of the bytes): 16-byte loads
        const bool vec = e.d[0] == 1 && e.d[1] == 1 && e.d[2] == 1 && e.sstride[3] == 1 && e.dtype_in == UMR_F32 && (e.soff & 3) == 0 &&
                         (((uintptr_t)e.src | (uintptr_t)e.dst) & 15) == 0;
        if (vec && e.dtype_out == UMR_BF16 && (lb * 4 + 4) * 2048 <= total) {
            // a whole block of a cast to bf16 (most of a refresh's bytes): all eight 16-byte loads in flight before the first store
            // (the loop below waits for each repetition's two loads in turn: four memory round trips per block)
            const float* sp = (const float*)e.src + e.soff + lb * 8192;
            bf16_t* dp = (bf16_t*)e.dst + lb * 8192;
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(sp + (j * 256 + threadIdx.x) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x4 t = {(bf16_t)v[j][0], (bf16_t)v[j][1], (bf16_t)v[j][2], (bf16_t)v[j][3]};
                *(bf16x4*)(dp + (j * 256 + threadIdx.x) * 4) = t;
            }
            return;
        }
#pragma unroll 1
        for (int rep = 0; rep < 4; ++rep) {      // 4 x 2048 consecutive destination elements per block
            const int64_t base = (lb * 4 + rep) * 2048;
            if (base >= total) break;
            if (vec && base + 2048 <= total) {
                const float* sp = (const float*)e.src + e.soff + base;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int o = (j * 256 + threadIdx.x) * 4;
                    const f32x4 v = *(const f32x4*)(sp + o);
                    if (e.dtype_out == UMR_F32) *(f32x4*)((float*)e.dst + base + o) = v;
                    else if (e.dtype_out == UMR_BF16) { bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]}; *(bf16x4*)((bf16_t*)e.dst + base + o) = t; }
                    else if ((e.rowlen & 3) == 0) {
                        const int64_t di = base + o, row = di / e.rowlen;
                        bf16_t* q = (bf16_t*)e.dst + row * 3 * e.rowlen + (di - row * e.rowlen);
                        bf16x4 h, m2, l;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const bf16_t hh = (bf16_t)v[c];
                            const float r1 = v[c] - (float)hh;
                            const bf16_t mm = (bf16_t)r1;
                            h[c] = hh; m2[c] = mm; l[c] = (bf16_t)(r1 - (float)mm);
                        }
                        *(bf16x4*)q = h; *(bf16x4*)(q + e.rowlen) = m2; *(bf16x4*)(q + 2 * e.rowlen) = l;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) perm_store(e, base + o + c, v[c]);
                    }
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t idx = base + j * 256 + threadIdx.x;
                if (idx >= total) break;
                int64_t r = idx;
                const int i3 = (int)(r % e.d[3]); r /= e.d[3];
                const int i2 = (int)(r % e.d[2]); r /= e.d[2];
                const int i1 = (int)(r % e.d[1]);
                const int i0 = (int)(r / e.d[1]);
                const int64_t si = e.soff + i0 * e.sstride[0] + i1 * e.sstride[1] + i2 * e.sstride[2] + i3 * e.sstride[3];
                const float v = e.dtype_in == UMR_F32 ? ((const float*)e.src)[si] : (float)((const bf16_t*)e.src)[si];
                perm_store(e, idx, v);
            }
        }
        return;
    }
    if (e.e[3] < 0) {
        // 2-D transpose (the Linear weights' data-gradient form, [N,K] f32 -> [K,N] bf16: most of the transposed bytes): 64 x 64
        // tiles, shifts instead of the generic tile's run-time divisions (three per element, twice), 16 loads in flight per
        // thread.  dst[r][c] = src[soff + r * sstride[2] + c * sstride[3]], sstride[2] == 1: reads run along r, writes along c.
        const int R = e.d[2], C = e.d[3];
        // r (the source's contiguous direction) fastest: neighbouring blocks read adjacent 256-byte pieces of the same 64 source rows
        const int ntr = (R + 63) >> 6;
        const int r0 = (int)(lb % ntr) << 6, c0 = (int)(lb / ntr) << 6;
        // f32 -> bf16 with 16-byte accesses on both sides: four f32x4 loads per thread along r, the tile transposed through LDS, two
        // 8 x bf16 stores per thread along c (the scalar form below moves 4-byte loads and 2-byte stores)
        if (e.dtype_in == UMR_F32 && e.dtype_out == UMR_BF16 && (R & 3) == 0 && (C & 7) == 0 && (e.sstride[3] & 3) == 0 && (e.soff & 3) == 0 &&
            (((uintptr_t)e.src | (uintptr_t)e.dst) & 15) == 0) {
            const int t = threadIdx.x;
            f32x4 v4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cc = c0 + k * 16 + (t >> 4), rr = r0 + (t & 15) * 4;
                v4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cc < C && rr < R) v4[k] = *(const f32x4*)((const float*)e.src + e.soff + (int64_t)rr + (int64_t)cc * e.sstride[3]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[(k * 16 + (t >> 4)) * 65 + (t & 15) * 4 + j] = v4[k][j];     // tile[c local][r local]
            __syncthreads();
            const int rl = t >> 2;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cl = (t & 3) * 8 + h * 32;
                if (r0 + rl < R && c0 + cl < C) {
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16_t)tile[(cl + j) * 65 + rl];
                    *(bf16x8*)((bf16_t*)e.dst + (int64_t)(r0 + rl) * C + c0 + cl) = o;
                }
            }
            return;
        }
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int cc = c0 + k * 4 + ty, rr = r0 + tx;
            v[k] = 0.f;
            if (cc < C && rr < R) {
                const int64_t si = e.soff + (int64_t)rr + (int64_t)cc * e.sstride[3];
                v[k] = e.dtype_in == UMR_F32 ? ((const float*)e.src)[si] : (float)((const bf16_t*)e.src)[si];
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) tile[(k * 4 + ty) * 65 + tx] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int rr = k * 4 + ty, cc = tx;
            if (r0 + rr < R && c0 + cc < C) {
                const float o = tile[cc * 65 + rr];
                const int64_t di = (int64_t)(r0 + rr) * C + c0 + cc;
                perm_store(e, di, o);
            }
        }
        return;
    }
    // tiled: tile coordinates (dimension 3 fastest), then its origin
    int nt[4], org[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) nt[k] = (e.d[k] + e.e[k] - 1) / e.e[k];
    int64_t r = lb;
    org[3] = (int)(r % nt[3]) * e.e[3]; r /= nt[3];
    org[2] = (int)(r % nt[2]) * e.e[2]; r /= nt[2];
    org[1] = (int)(r % nt[1]) * e.e[1]; r /= nt[1];
    org[0] = (int)r * e.e[0];
    const int T = e.e[0] * e.e[1] * e.e[2] * e.e[3];
    const int o0 = e.ord[0], o1 = e.ord[1], o2 = e.ord[2], o3 = e.ord[3];   // o0 = the dimension that runs fastest in the source
    for (int idx = threadIdx.x; idx < T; idx += 256) {
        int l[4];
        int q = idx;
        l[o0] = q % e.e[o0]; q /= e.e[o0];
        l[o1] = q % e.e[o1]; q /= e.e[o1];
        l[o2] = q % e.e[o2];
        l[o3] = q / e.e[o2];
        const bool in = org[0] + l[0] < e.d[0] && org[1] + l[1] < e.d[1] && org[2] + l[2] < e.d[2] && org[3] + l[3] < e.d[3];
        if (in) {
            const int64_t si = e.soff + (int64_t)(org[0] + l[0]) * e.sstride[0] + (int64_t)(org[1] + l[1]) * e.sstride[1] +
                               (int64_t)(org[2] + l[2]) * e.sstride[2] + (int64_t)(org[3] + l[3]) * e.sstride[3];
            // row pitch e[3] + 1: the gather walks the tile along a dimension other than 3, i.e. at a stride of e[3] floats -- 64
            // floats = 256 B = every lane on the same LDS bank without the pad (the transposes ran at 1.2 TB/s because of it)
            tile[((l[0] * e.e[1] + l[1]) * e.e[2] + l[2]) * (e.e[3] + 1) + l[3]] =
                e.dtype_in == UMR_F32 ? ((const float*)e.src)[si] : (float)((const bf16_t*)e.src)[si];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T; idx += 256) {
        int q = idx;
        const int l3 = q % e.e[3]; q /= e.e[3];
        const int l2 = q % e.e[2]; q /= e.e[2];
        const int l1 = q % e.e[1];
        const int l0 = q / e.e[1];
        const int i0 = org[0] + l0, i1 = org[1] + l1, i2 = org[2] + l2, i3 = org[3] + l3;
        if (i0 < e.d[0] && i1 < e.d[1] && i2 < e.d[2] && i3 < e.d[3]) {
            const int64_t di = (((int64_t)i0 * e.d[1] + i1) * e.d[2] + i2) * e.d[3] + i3;
            const float v = tile[((l0 * e.e[1] + l1) * e.e[2] + l2) * (e.e[3] + 1) + l3];
            perm_store(e, di, v);
        }
    }
}

// ---------------------------------------------------------------- small reductions / fills
// out[r][c] (f32 or T) = sum_{k<reps} x[(r*reps + k)*ld_rep + c]   (segment sum over `reps` consecutive row blocks)
template <typename T, typename TO>
__global__ void segsum_kernel(const T* __restrict__ x, TO* __restrict__ out, int R, int reps, int64_t rep_stride, int64_t seg_stride,
                              int C, int accumulate) {
    const int64_t total = (int64_t)R * C;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const int r = (int)(idx / C);
        const T* p = x + (int64_t)r * seg_stride + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains (the loop is latency-bound)
        int k = 0;
        for (; k + 4 <= reps; k += 4) {
            s0 += to_f32<T>(p[(int64_t)(k + 0) * rep_stride]);
            s1 += to_f32<T>(p[(int64_t)(k + 1) * rep_stride]);
            s2 += to_f32<T>(p[(int64_t)(k + 2) * rep_stride]);
            s3 += to_f32<T>(p[(int64_t)(k + 3) * rep_stride]);
        }
        for (; k < reps; ++k) s0 += to_f32<T>(p[(int64_t)k * rep_stride]);
        float s = (s0 + s1) + (s2 + s3);
        if (accumulate) s += to_f32<TO>(out[idx]);
        out[idx] = from_f32<TO>(s);
    }
}

// tokens[b, 0, :] = cls + pos[0, :]
template <typename T>
__global__ void fill_cls_kernel(T* __restrict__ tokens, const float* __restrict__ cls, const float* __restrict__ pos0, int B, int64_t bstride, int D) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, c = idx - b * D;
    tokens[(int64_t)b * bstride + c] = from_f32<T>(cls[c] + pos0[c]);
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int64_t n, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = from_f32<TO>(to_f32<TI>(src[i]) * scale);
}

// ---------------------------------------------------------------- f32 -> three bf16 planes per row (UMR_BF16X3 operands)
// one thread = 4 consecutive k of one row: 16 B read, 3 x 8 B written (a wave covers 256 consecutive k: 1 KiB read, 3 x 512 B written)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int K,
                                                     int64_t ld_src, int64_t ld_dst, int rows_in, int rows_out, int row_off) {
    const int k4 = K >> 2;
    const int64_t total = rows * k4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / k4;
        const int k = (int)(i - r * k4) * 4;
        const int64_t rs = rows_in > 0 ? (r / rows_in) * rows_out + row_off + (r % rows_in) : r;   // gather of source rows (token rows without the class tokens)
        const f32x4 x = *(const f32x4*)(src + rs * ld_src + k);
        bf16x4 h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bf16_t hh = (bf16_t)x[e];
            const float r1 = x[e] - (float)hh;
            const bf16_t mm = (bf16_t)r1;
            h[e] = hh; m[e] = mm; l[e] = (bf16_t)(r1 - (float)mm);
        }
        bf16_t* d = dst + r * ld_dst + k;
        *(bf16x4*)d = h;
        *(bf16x4*)(d + K) = m;
        *(bf16x4*)(d + 2 * K) = l;
    }
}

// planes -> f32: x = (h + m) + l, exact (the inverse of split3 for every value split3 represents)
__global__ __launch_bounds__(256) void unsplit3_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t rows, int K,
                                                       int64_t ld_src, int64_t ld_dst) {
    const int k4 = K >> 2;
    const int64_t total = rows * k4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / k4;
        const int k = (int)(i - r * k4) * 4;
        const bf16_t* q = src + r * ld_src + k;
        const bf16x4 h = *(const bf16x4*)q, m = *(const bf16x4*)(q + K), l = *(const bf16x4*)(q + 2 * K);
        f32x4 x;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = ((float)h[e] + (float)m[e]) + (float)l[e];
        *(f32x4*)(dst + r * ld_dst + k) = x;
    }
}

// ---------------------------------------------------------------- head output layer: 1024 -> {1,2} (+ activation), NCHW f32 out
// one wave per pixel row: lanes split K, wave reduction.  out[b][c][hw]
template <typename T>
__global__ __launch_bounds__(256) void head_out_fwd_kernel(const T* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ out, int64_t M, int K, int Cout, int HW, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t m = wave; m < M; m += nw) {
        const T* row = h + m * K;
        float s0 = 0.f, s1 = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
            const f32x4 v = Vec4<T>::load(row + k);
            const f32x4 w0 = *(const f32x4*)(w + k);
            s0 += v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3];
            if (Cout == 2) {
                const f32x4 w1 = *(const f32x4*)(w + K + k);
                s1 += v[0] * w1[0] + v[1] * w1[1] + v[2] * w1[2] + v[3] * w1[3];
            }
        }
        s0 = wave_sum(s0);
        if (Cout == 2) s1 = wave_sum(s1);
        if (lane == 0) {
            const int64_t b = m / HW, hw = m - b * HW;
            float v0 = s0 + bias[0];
            if (act == UMR_ACT_TANH) v0 = tanhf(v0); else if (act == 4) v0 = sinf(v0);
            out[(b * Cout) * HW + hw] = v0;
            if (Cout == 2) {
                float v1 = s1 + bias[1];
                if (act == UMR_ACT_TANH) v1 = tanhf(v1); else if (act == 4) v1 = sinf(v1);
                out[(b * Cout + 1) * HW + hw] = v1;
            }
        }
    }
}

// second half of the fused output layer: the producing GEMM left per-256-column-tile partial sums
// part[t][m][c] (umr_gemm_desc.red_out); add them in tile order, then bias + activation, NCHW f32 out
__global__ __launch_bounds__(256) void head_out_finish_kernel(const float* __restrict__ part, int nparts, const float* __restrict__ bias,
                                                              float* __restrict__ out, int64_t M, int Cout, int HW, int act) {
    const int64_t total = M * Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / Cout;
        const int c = (int)(i - m * Cout);
        float v = 0.f;
        for (int t = 0; t < nparts; ++t) v += part[(int64_t)t * total + i];
        v += bias[c];
        if (act == UMR_ACT_TANH) v = tanhf(v); else if (act == 4) v = sinf(v);
        const int64_t b = m / HW, hw = m - b * HW;
        out[(b * Cout + c) * HW + hw] = v;
    }
}

// backward: g[m][c] = dout[b][c][hw] * act'(.);  dh[m][k] = sum_c g[m][c] w[c][k] (optionally * (h>0): fused ReLU
// backward of the producer);  partial dW[c][k], db[c] per block -> workspace, reduced by head_out_reduce.
template <typename T>
__global__ __launch_bounds__(256) void head_out_bwd_kernel(const T* __restrict__ h, const float* __restrict__ w, const float* __restrict__ dout,
                                                           const float* __restrict__ yout, T* __restrict__ dh, float* __restrict__ part,
                                                           int64_t M, int K, int Cout, int HW, int act, int relu_mask, int rows_per_block) {
    // thread t owns columns k = 4t..4t+3 (K <= 1024); loops rows of this block
    const int t = threadIdx.x;
    const int k = t * 4;
    const bool kok = k < K;
    f32x4 w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0}, a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    if (kok) { w0 = *(const f32x4*)(w + k); if (Cout == 2) w1 = *(const f32x4*)(w + K + k); }
    float b0 = 0.f, b1 = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (int64_t m = r0; m < r1; ++m) {
        const int64_t b = m / HW, hw = m - b * HW;
        float g0 = dout[(b * Cout) * HW + hw], g1 = 0.f;
        // tanh: yout is the output y (1 - y^2); sine (act 4, objectness_net.py:30-35,126): yout is the PRE-activation z (cos z)
        if (act == UMR_ACT_TANH) { const float y = yout[(b * Cout) * HW + hw]; g0 *= (1.f - y * y); }
        else if (act == 4) g0 *= cosf(yout[(b * Cout) * HW + hw]);
        if (Cout == 2) {
            g1 = dout[(b * Cout + 1) * HW + hw];
            if (act == UMR_ACT_TANH) { const float y = yout[(b * Cout + 1) * HW + hw]; g1 *= (1.f - y * y); }
            else if (act == 4) g1 *= cosf(yout[(b * Cout + 1) * HW + hw]);
        }
        if (kok) {
            const f32x4 hv = Vec4<T>::load(h + m * K + k);
            f32x4 d = w0 * g0 + w1 * g1;
            if (relu_mask) for (int j = 0; j < 4; ++j) d[j] = hv[j] > 0.f ? d[j] : 0.f;
            Vec4<T>::store(dh + m * K + k, d);
            a0 += hv * g0;
            a1 += hv * g1;
        }
        b0 += g0;
        b1 += g1;
    }
    float* p = part + (int64_t)blockIdx.x * (2 * K + 2);
    if (kok) { *(f32x4*)(p + k) = a0; *(f32x4*)(p + K + k) = a1; }
    if (t == 0) { p[2 * K] = b0; p[2 * K + 1] = b1; }
}

// K = 1024, bf16 (the centre head's 1024-channel layer: 19.3 GB in, 19.3 GB out at cfg2).  The generic kernel above walks its
// rows one at a time behind a chain of scalar loads of dout / yout (one 8-byte load in flight per lane: 5.1 TB/s).  Here a wave
// owns one 512-column half of a run of 64 consecutive rows: lane l first fetches the gradient pair of row l (coalesced), the row
// loop broadcasts it with v_readlane and keeps HOB_UNROLL 16-byte loads per lane in flight.  Waves (half, stream) = (wv & 1,
// wv >> 1); the two row streams of a block are added in a fixed order through LDS, so the partial slab keeps its layout.
#ifndef HOB_UNROLL
#define HOB_UNROLL 2
#endif
#ifndef HOB_NT
#define HOB_NT 0
#endif
__global__ __launch_bounds__(256) void head_out_bwd_k1024_kernel(const bf16_t* __restrict__ h, const float* __restrict__ w,
                                                                 const float* __restrict__ dout, const float* __restrict__ yout,
                                                                 bf16_t* __restrict__ dh, float* __restrict__ part, int64_t M, int Cout,
                                                                 int HW, int act, int relu_mask, int rows_per_block) {
    constexpr int K = 1024;
    __shared__ float sh[2 * K + 2];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch = wv & 1, rs = wv >> 1;
    const int col = ch * 512 + lane * 8;
    float w0[8], w1[8], a0[8], a1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w0[j] = w[col + j]; w1[j] = Cout == 2 ? w[K + col + j] : 0.f; a0[j] = 0.f; a1[j] = 0.f; }
    float bs0 = 0.f, bs1 = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (int64_t base = r0 + rs * 64; base < r1; base += 128) {
        const int n = (int)(r1 - base < 64 ? r1 - base : 64);
        float g0v = 0.f, g1v = 0.f;
        if (lane < n) {
            const int64_t m = base + lane;
            const int64_t b = m / HW, hw = m - b * HW;
            const int64_t o0 = (b * Cout) * HW + hw;
            g0v = dout[o0];
            if (act == UMR_ACT_TANH) { const float y = yout[o0]; g0v *= (1.f - y * y); }
            if (Cout == 2) {
                g1v = dout[o0 + HW];
                if (act == UMR_ACT_TANH) { const float y = yout[o0 + HW]; g1v *= (1.f - y * y); }
            }
        }
        bs0 += g0v;
        bs1 += g1v;
        const bf16_t* hp = h + base * K + col;
        bf16_t* dp = dh + base * K + col;
        auto row = [&](const bf16x8 t, int i) {
            const float g0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g0v), i));
            const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g1v), i));
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float hv = (float)t[j];
                float d = w0[j] * g0 + w1[j] * g1;
                if (relu_mask) d = hv > 0.f ? d : 0.f;
                o[j] = (bf16_t)d;
                a0[j] += hv * g0;
                a1[j] += hv * g1;
            }
#if HOB_NT & 2
            __builtin_nontemporal_store(__builtin_bit_cast(f32x4, o), (f32x4*)(dp + (int64_t)i * K));
#else
            *(bf16x8*)(dp + (int64_t)i * K) = o;
#endif
        };
        int i = 0;
        for (; i + HOB_UNROLL <= n; i += HOB_UNROLL) {
            bf16x8 t[HOB_UNROLL];
#pragma unroll
            for (int u = 0; u < HOB_UNROLL; ++u) {
#if HOB_NT & 1
                t[u] = __builtin_bit_cast(bf16x8, __builtin_nontemporal_load((const f32x4*)(hp + (int64_t)(i + u) * K)));
#else
                t[u] = *(const bf16x8*)(hp + (int64_t)(i + u) * K);
#endif
            }
#pragma unroll
            for (int u = 0; u < HOB_UNROLL; ++u) row(t[u], i + u);
        }
        for (; i < n; ++i) row(*(const bf16x8*)(hp + (int64_t)i * K), i);
    }
    bs0 = wave_sum(bs0);
    bs1 = wave_sum(bs1);
    for (int s = 0; s < 2; ++s) {
        if (rs == s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sh[col + j] = s ? sh[col + j] + a0[j] : a0[j];
                sh[K + col + j] = s ? sh[K + col + j] + a1[j] : a1[j];
            }
            if (ch == 0 && lane == 0) {
                sh[2 * K] = s ? sh[2 * K] + bs0 : bs0;
                sh[2 * K + 1] = s ? sh[2 * K + 1] + bs1 : bs1;
            }
        }
        __syncthreads();
    }
    float* p = part + (int64_t)blockIdx.x * (2 * K + 2);
    for (int i = threadIdx.x; i < 2 * K + 2; i += 256) p[i] = sh[i];
}

// fixed-order sum of the per-block partials: 64 entries x 4 interleaved block slices per workgroup, four independent
// accumulation chains per thread (the one-thread-per-entry loop over 2048 blocks took 0.7 ms)
__global__ __launch_bounds__(256) void head_out_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                              int nblocks, int K, int Cout) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cl;
    const int per = 2 * K + 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < per) {
        const float* q = part + idx;
        int b = sl;
        for (; b + 12 < nblocks; b += 16) {
            s0 += q[(int64_t)b * per];
            s1 += q[(int64_t)(b + 4) * per];
            s2 += q[(int64_t)(b + 8) * per];
            s3 += q[(int64_t)(b + 12) * per];
        }
        for (; b < nblocks; b += 4) s0 += q[(int64_t)b * per];
    }
    sh[sl][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && idx < per) {
        const float s = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
        if (idx < K) dw[idx] = s;
        else if (idx < 2 * K) { if (Cout == 2) dw[idx] = s; }
        else if (idx == 2 * K) db[0] = s;
        else if (Cout == 2) db[1] = s;
    }
}

// ---------------------------------------------------------------- fused loss (train_objectness_net.py:215-254)
// partial sums of the four terms per block -> part[block][4]; gradients w.r.t. the predictions in the same pass.
struct LossCfg { int B, H, W; int center_l2, sdf_l2, use_grad, use_bce; float gscale; };

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }

__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ pc, const float* __restrict__ ps, const float* __restrict__ gc,
                                                   const float* __restrict__ gs, const float* __restrict__ sal, float* __restrict__ dpc,
                                                   float* __restrict__ dps, float* __restrict__ part, LossCfg cfg) {
    __shared__ float red[4][4];
    const int H = cfg.H, W = cfg.W;
    const int64_t HW = (int64_t)H * W, total = (int64_t)cfg.B * HW;
    const float n_c = 1.f / (float)(total * 2), n_s = 1.f / (float)total;
    const float n_g = (H > 1 && W > 1) ? 1.f / (float)((int64_t)cfg.B * 2 * (H - 1) * (W - 1)) : 0.f;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = idx / HW, hw = idx - b * HW;
        const int y = (int)(hw / W), x = (int)(hw - (int64_t)y * W);
        // center term
        for (int c = 0; c < 2; ++c) {
            const int64_t o = (b * 2 + c) * HW + hw;
            const float d = pc[o] - gc[o];
            t0 += cfg.center_l2 ? d * d : fabsf(d);
            if (dpc) dpc[o] = cfg.gscale * n_c * (cfg.center_l2 ? 2.f * d : sgn(d));
        }
        // sdf term
        const int64_t o = b * HW + hw;
        const float p = ps[o], d = p - gs[o];
        t1 += cfg.sdf_l2 ? d * d : fabsf(d);
        float g = n_s * (cfg.sdf_l2 ? 2.f * d : sgn(d));
        // gradient-map term: e = (gt diff) - (pred diff) on the (H-1)x(W-1) interior
        if (cfg.use_grad) {
            const float* P = ps + b * HW;
            const float* G = gs + b * HW;
            auto f = [&](float e) { return cfg.sdf_l2 ? 2.f * e : sgn(e); };
            if (y < H - 1 && x < W - 1) {
                const float ey = (G[hw + W] - G[hw]) - (P[hw + W] - p);
                const float ex = (G[hw + 1] - G[hw]) - (P[hw + 1] - p);
                t2 += cfg.sdf_l2 ? ey * ey + ex * ex : fabsf(ey) + fabsf(ex);
                g += n_g * (f(ey) + f(ex));  // d e / d p[y,x] = +1 for both
            }
            if (y > 0 && x < W - 1) {  // dy at (y-1, x): -(p[y,x] - p[y-1,x])
                const float ey = (G[hw] - G[hw - W]) - (p - P[hw - W]);
                g -= n_g * f(ey);
            }
            if (x > 0 && y < H - 1) {  // dx at (y, x-1)
                const float ex = (G[hw] - G[hw - 1]) - (p - P[hw - 1]);
                g -= n_g * f(ex);
            }
        }
        if (cfg.use_bce) {
            const float t = sal[o];
            const float q = 1.f / (1.f + expf(-p));
            const float lq = fmaxf(logf(q), -100.f), l1q = fmaxf(logf(1.f - q), -100.f);
            t3 += -(t * lq + (1.f - t) * l1q);
            const float gb = (q - t) / fmaxf((1.f - q) * q, 1e-12f);
            g += n_s * gb * q * (1.f - q);
        }
        if (dps) dps[o] = cfg.gscale * g;
    }
    t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2); t3 = wave_sum(t3);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = t0; red[wv][1] = t1; red[wv][2] = t2; red[wv][3] = t3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const float nrm = threadIdx.x == 0 ? n_c : (threadIdx.x == 2 ? n_g : n_s);
        part[(int64_t)blockIdx.x * 4 + threadIdx.x] = s * nrm;
    }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ part, float* __restrict__ out5, int nblocks) {
    // out5 = [total, center, sdf, grad, bce].  Fixed-order tree (four threads walking all blocks serially took 82 us)
    __shared__ f32x4 sh[256];
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblocks; b += 256) a += *(const f32x4*)(part + (int64_t)b * 4);
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const f32x4 t = sh[0];
        out5[1] = t[0]; out5[2] = t[1]; out5[3] = t[2]; out5[4] = t[3];
        out5[0] = ((t[0] + t[1]) + t[2]) + t[3];
    }
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam defaults; train_objectness_net.py:96)
// ONE update expression for every Adam kernel of this file, written with the round-to-nearest intrinsics: the compiler neither contracts
// them into other fused multiply-adds nor reassociates them, so the plain, the device-scalar and the copy-writing launches give the same
// bits whatever code surrounds the call (the copy-writing kernel's unrolled form first came out one ulp away from the plain one's)
__device__ __forceinline__ void adam_update1(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float bc1,
                                             float bc2_sqrt, float gscale) {
    const float gi = __fmul_rn(g, gscale);
    const float mi = __fmaf_rn(m, b1, __fmul_rn(gi, __fsub_rn(1.f, b1)));
    const float vi = __fmaf_rn(v, b2, __fmul_rn(__fmul_rn(gi, gi), __fsub_rn(1.f, b2)));
    m = mi;
    v = vi;
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);
    p = __fsub_rn(p, __fmul_rn(__fdiv_rn(lr, bc1), __fdiv_rn(mi, denom)));
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                            float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_update1(pi, g[i], mi, vi, lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// ---------------------------------------------------------------- Adam that writes the kernel-layout bf16 weight copies itself
// Round 6 (small-step regime: the reference recipe's 347.6 M parameters per 20-ms step): the optimizer pass already holds every updated
// weight in registers; writing the bf16 copies the GEMMs read -- the [N][K] forward operand and the [K][N] data-gradient operand of a
// Linear weight -- from there saves the refresh pass that re-read the f32 weights (umr_permute4_batched: 12 B per parameter and copy pair
// on top of Adam's 28; now 4).  One launch covers a whole stage of the flat parameter buffer: `plain` entries are ranges updated as
// umr_adam_step_hyper does, `weight` entries are 2-D [N][K] weights walked in 64 x 64 tiles (16-byte accesses on every stream; the
// transposed copy goes through LDS).  Same update expression as adam_hyper_kernel, same rounding of the copies as the refresh pass's
// casts: bit-identical weights and copies (tests/test_train_gpu.py::test_adam_pack_equals_adam_then_refresh).
struct AdamPackEntry { float* p; const float* g; float* m; float* v; void* dst_lin; void* dst_t; int64_t n; int32_t N, K; int64_t blk_start; };
static_assert(sizeof(AdamPackEntry) == sizeof(umr_adam_pack_entry), "umr_adam_pack_entry layout");

__device__ __forceinline__ void adam_update4(f32x4& p, const f32x4& g, f32x4& m, f32x4& v, float lr, float b1, float b2, float eps, float bc1,
                                             float bc2_sqrt, float gscale) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float pc = p[c], mc = m[c], vc = v[c];
        adam_update1(pc, g[c], mc, vc, lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
        p[c] = pc; m[c] = mc; v[c] = vc;
    }
}

__global__ __launch_bounds__(256) void adam_pack_kernel(const AdamPackEntry* __restrict__ table, const int32_t* __restrict__ blk_entry,
                                                        const float* __restrict__ hyper) {
    __shared__ float tile[64 * 65];
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2_sqrt = hyper[5], gscale = hyper[6];
    const AdamPackEntry e = table[blk_entry[blockIdx.x]];
    const int64_t lb = (int64_t)blockIdx.x - e.blk_start;
    const int t = threadIdx.x;
    if (e.N == 0) {
        // plain range: 4096 elements per block, four 16-byte accesses per thread and stream; a tail shorter than four goes scalar
        const int64_t base = lb * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = base + (int64_t)(j * 256 + t) * 4;
            if (i + 4 <= e.n) {
                f32x4 p = *(const f32x4*)(e.p + i), m = *(const f32x4*)(e.m + i), v = *(const f32x4*)(e.v + i);
                const f32x4 g = *(const f32x4*)(e.g + i);
                adam_update4(p, g, m, v, lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
                *(f32x4*)(e.p + i) = p; *(f32x4*)(e.m + i) = m; *(f32x4*)(e.v + i) = v;
            } else {
                for (int64_t q = i; q < e.n; ++q) {
                    float pq = e.p[q], mq = e.m[q], vq = e.v[q];
                    adam_update1(pq, e.g[q], mq, vq, lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
                    e.p[q] = pq; e.m[q] = mq; e.v[q] = vq;
                }
            }
        }
        return;
    }
    // weight [N][K]: tile rows n0.., columns k0.. (k fastest over the blocks: neighbouring blocks touch adjacent 256-byte pieces of the same rows)
    const int ntk = (e.K + 63) >> 6;
    const int k0 = (int)(lb % ntk) << 6, n0 = (int)(lb / ntk) << 6;
    const int kc = k0 + (t & 15) * 4;
    f32x4 pw[4];
    bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j * 16 + (t >> 4);
        live[j] = n < e.N && kc < e.K;          // K % 4 == 0: a chunk is inside or outside as a whole
        pw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live[j]) {
            const int64_t i = (int64_t)n * e.K + kc;
            f32x4 m = *(const f32x4*)(e.m + i), v = *(const f32x4*)(e.v + i);
            const f32x4 g = *(const f32x4*)(e.g + i);
            pw[j] = *(const f32x4*)(e.p + i);
            adam_update4(pw[j], g, m, v, lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
            *(f32x4*)(e.p + i) = pw[j]; *(f32x4*)(e.m + i) = m; *(f32x4*)(e.v + i) = v;
            if (e.dst_lin) {
                const bf16x4 o = {(bf16_t)pw[j][0], (bf16_t)pw[j][1], (bf16_t)pw[j][2], (bf16_t)pw[j][3]};
                *(bf16x4*)((bf16_t*)e.dst_lin + i) = o;
            }
        }
    }
    if (!e.dst_t) return;
    // transposed copy dst_t[k][n]: tile[n local][k local] through LDS, then 16-byte stores along n
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) tile[(j * 16 + (t >> 4)) * 65 + (t & 15) * 4 + c] = pw[j][c];
    __syncthreads();
    const int kl = t >> 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int nl = (t & 3) * 8 + h * 32;
        if (k0 + kl < e.K && n0 + nl < e.N) {      // N % 8 == 0: eight rows are inside or outside as a whole
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)tile[(nl + j) * 65 + kl];
            *(bf16x8*)((bf16_t*)e.dst_t + (int64_t)(k0 + kl) * e.N + n0 + nl) = o;
        }
    }
}

// same update with its seven scalars read from device memory (a captured HIP graph replays this launch every step with new values)
__global__ void adam_hyper_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                  const float* __restrict__ hyper) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2_sqrt = hyper[5], gscale = hyper[6];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_update1(pi, g[i], mi, vi, lr, b1, b2, eps, bc1, bc2_sqrt, gscale);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

inline int grid_for(int64_t total, int block = 256, int cap = 8192) {
    int64_t g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                   \
    if ((dtype) == UMR_BF16) { typedef bf16_t T; CALL; }          \
    else if ((dtype) == UMR_F32) { typedef float T; CALL; }       \
    else return umr_set_error(UMR_ERR_INVALID, "dtype");

extern "C" int umr_patchify(const float* images, void* out, int B, int H, int W, int patch, int ldk, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(images && out, "patchify: null pointer");
    UMR_CHECK_ARG(B > 0 && patch > 0 && H >= patch && W >= patch && ldk >= 3 * patch * patch, "patchify: bad geometry");
    const int gh = H / patch, gw = W / patch;
    const int64_t total = (int64_t)B * gh * gw * 3 * patch;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL(patchify_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, images, (T*)out, B, H, W, patch, gh, gw, ldk));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

// rows of blocks of the resize kernels: a block loops over (image, row) pairs at stride gridDim.y.  UMR_BILINEAR_GY (read per
// launch) caps it: one output row per block is 1.2 M blocks of one element per thread at the cfg2 feature map
static int bilinear_gy_cap() {
    const int v = umr_opt_or(UMR_OPT_BILINEAR_GY, 0);
    return v > 0 ? v : 65535;
}

// flags: UMR_BILINEAR_RELU = max(., 0) on the result; UMR_BILINEAR_OUT_X3 (dtype f32, C % 8 == 0) = the result as three bf16 planes per
// pixel, ldy in bf16 elements (>= 3 C).  ldx / ldy: elements between consecutive pixels (0 = C: dense maps)
extern "C" int umr_bilinear_fwd_ex(const void* x, int64_t ldx, void* y, int64_t ldy, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                   int align_corners, int flags, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, "bilinear_fwd: bad arguments");
    const bool planes = (flags & UMR_BILINEAR_OUT_X3) != 0;
    const int relu = (flags & UMR_BILINEAR_RELU) ? 1 : 0;
    if (ldx == 0) ldx = C;
    if (ldy == 0) ldy = planes ? 3 * (int64_t)C : C;
    UMR_CHECK_ARG(ldx >= C && ldy >= (planes ? 3 * (int64_t)C : C) && ldx % 4 == 0 && ldy % 4 == 0, "bilinear_fwd: ldx / ldy smaller than a pixel or not a multiple of 4");
    UMR_CHECK_ARG(!planes || (dtype == UMR_F32 && C % 8 == 0 && ldy % 8 == 0), "bilinear_fwd: plane output needs f32 input, C and ldy multiples of 8");
    hipStream_t s = (hipStream_t)stream;
    {
        const int nv = (C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0) ? 2 : 1;
        const int rowlen = Wo * (C / (4 * nv));
        int64_t gy = (int64_t)B * Ho;
        if (gy > bilinear_gy_cap()) gy = bilinear_gy_cap();
        const dim3 g((unsigned)((rowlen + 255) / 256), (unsigned)gy);
        if (planes) {
            hipLaunchKernelGGL((bilinear_fwd_kernel<float, 2, true>), g, dim3(256), 0, s, (const float*)x, y, B, Hi, Wi, Ho, Wo, C, align_corners, ldx, ldy, relu);
        } else if (nv == 2 && dtype == UMR_BF16) {
            // rows of a run (see the kernel): 16 where that still leaves >= 2048 blocks, fewer on small maps
            int run = 16;
            while (run > 2 && (int64_t)B * ((Ho + run - 1) / run) * g.x < 2048) run = (run + 1) / 2;
            int64_t runs = (int64_t)B * ((Ho + run - 1) / run);
            if (runs > bilinear_gy_cap()) runs = bilinear_gy_cap();
            hipLaunchKernelGGL(bilinear_fwd_bf16x8_kernel, dim3(g.x, (unsigned)runs), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, Hi, Wi, Ho, Wo, C, align_corners, ldx, ldy, relu, run);
        }
        else if (nv == 2) { DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_fwd_kernel<T, 2>), g, dim3(256), 0, s, (const T*)x, y, B, Hi, Wi, Ho, Wo, C, align_corners, ldx, ldy, relu)); }
        else { DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_fwd_kernel<T, 1>), g, dim3(256), 0, s, (const T*)x, y, B, Hi, Wi, Ho, Wo, C, align_corners, ldx, ldy, relu)); }
    }
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_bilinear_fwd(const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, int dtype,
                                umr_stream_t stream) {
    return umr_bilinear_fwd_ex(x, 0, y, 0, B, Hi, Wi, Ho, Wo, C, align_corners, 0, dtype, stream);
}

// lddy / lddx: elements between consecutive pixels of dy / dx (0 = C)
extern "C" int umr_bilinear_bwd_ex(const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                   int align_corners, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0, "bilinear_bwd: bad arguments");
    if (lddy == 0) lddy = C;
    if (lddx == 0) lddx = C;
    UMR_CHECK_ARG(lddy >= C && lddx >= C && lddy % 4 == 0 && lddx % 4 == 0 && (int64_t)Wo * lddy < ((int64_t)1 << 31),
                  "bilinear_bwd: lddy / lddx smaller than a pixel, not a multiple of 4, or a dy row of 2^31 elements");
    hipStream_t s = (hipStream_t)stream;
    {
        const int nv = (C % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0) ? 2 : 1;
        const int rowlen = Wi * (C / (4 * nv));
        int64_t gy = (int64_t)B * Hi;
        if (gy > bilinear_gy_cap()) gy = bilinear_gy_cap();
        dim3 g((unsigned)((rowlen + 255) / 256), (unsigned)gy);
        const float sh_ = align_corners ? (Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f) : (float)Hi / (float)Ho;
        const float sw_ = align_corners ? (Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f) : (float)Wi / (float)Wo;
        if (nv == 2 && dtype == UMR_BF16 && sh_ >= 0.4f && sw_ >= 0.4f)    // <= 8 candidate outputs per axis (see the kernel)
        {
            // rows of a run (see the kernel): 12 where that still leaves >= 2048 blocks, fewer on small maps
            int run = 12;
            while (run > 2 && (int64_t)B * ((Hi + run - 1) / run) * g.x < 2048) run = (run + 1) / 2;
            int64_t runs = (int64_t)B * ((Hi + run - 1) / run);
            if (runs > bilinear_gy_cap()) runs = bilinear_gy_cap();
            g.y = (unsigned)runs;
            // align_corners: contributors of input column i are the outputs o with o * sw in (i - 1, i + 1): at most lo + 5 for sw >= 0.49
            if (align_corners && sw_ >= 0.49f) hipLaunchKernelGGL(bilinear_bwd_bf16x8_kernel<6>, g, dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, B, Hi, Wi, Ho, Wo, C, align_corners, lddy, lddx, run);
            else hipLaunchKernelGGL(bilinear_bwd_bf16x8_kernel<8>, g, dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, B, Hi, Wi, Ho, Wo, C, align_corners, lddy, lddx, run);
        }
        else if (nv == 2) { DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_bwd_kernel<T, 2>), g, dim3(256), 0, s, (const T*)dy, (T*)dx, B, Hi, Wi, Ho, Wo, C, align_corners, lddy, lddx)); }
        else { DISPATCH_T(dtype, hipLaunchKernelGGL((bilinear_bwd_kernel<T, 1>), g, dim3(256), 0, s, (const T*)dy, (T*)dx, B, Hi, Wi, Ho, Wo, C, align_corners, lddy, lddx)); }
    }
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_bilinear_bwd(const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, int dtype,
                                umr_stream_t stream) {
    return umr_bilinear_bwd_ex(dy, 0, dx, 0, B, Hi, Wi, Ho, Wo, C, align_corners, dtype, stream);
}

extern "C" int umr_pixel_shuffle(const void* src, void* dst, int B, int H, int W, int s_, int C, int inverse, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && s_ > 0 && C > 0 && C % 4 == 0, "pixel_shuffle: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W * s_ * s_ * (C / 4);
    DISPATCH_T(dtype, hipLaunchKernelGGL(pixel_shuffle_kernel<T>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, s, (const T*)src, (T*)dst, B, H, W, s_, C, inverse));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_zero_stuff2(const void* dy, void* out, int B, int H, int W, int Ho, int Wo, int C, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(dy && out && B > 0 && C > 0 && C % 4 == 0, "zero_stuff2: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W * (C / 4);
    DISPATCH_T(dtype, hipLaunchKernelGGL(zero_stuff2_kernel<T>, dim3(grid_for(total, 256, 65536)), dim3(256), 0, s, (const T*)dy, (T*)out, B, H, W, Ho, Wo, C));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

// dst[i0,i1,i2,i3] (dims = dst_dims) = src[soff + sum_k i_k * sstride_k]; dtype_in/out in {F32,BF16}
extern "C" int umr_permute4(const void* src, void* dst, const int32_t* dst_dims, const int64_t* src_strides, int64_t src_offset,
                            int dtype_in, int dtype_out, int accumulate, umr_stream_t stream) {
    UMR_CHECK_ARG(src && dst && dst_dims && src_strides, "permute4: null pointer");
    PermDesc pd;
    int64_t total = 1;
    for (int i = 0; i < 4; ++i) { pd.d[i] = dst_dims[i]; pd.sstride[i] = src_strides[i]; total *= dst_dims[i]; UMR_CHECK_ARG(dst_dims[i] > 0, "permute4: dims"); }
    pd.soff = src_offset;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for(total, 256, 65536)), b(256);
    if (dtype_in == UMR_F32 && dtype_out == UMR_F32) hipLaunchKernelGGL((permute4_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, pd, accumulate);
    else if (dtype_in == UMR_F32 && dtype_out == UMR_BF16) hipLaunchKernelGGL((permute4_kernel<float, bf16_t>), g, b, 0, s, (const float*)src, (bf16_t*)dst, pd, accumulate);
    else if (dtype_in == UMR_BF16 && dtype_out == UMR_F32) hipLaunchKernelGGL((permute4_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, (float*)dst, pd, accumulate);
    else if (dtype_in == UMR_BF16 && dtype_out == UMR_BF16) hipLaunchKernelGGL((permute4_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, pd, accumulate);
    else return umr_set_error(UMR_ERR_INVALID, "permute4: dtype");
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_permute4_batched(const umr_perm_entry* table_dev, int n, int64_t total_blocks, const int32_t* blk_entry_dev, umr_stream_t stream) {
    UMR_CHECK_ARG(table_dev && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "permute4_batched: bad arguments");
    hipLaunchKernelGGL(permute4_batched_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PermEntry*)table_dev, n,
                       blk_entry_dev);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

// out[r][c] = sum_{k<reps} x[r*seg_stride + k*rep_stride + c]
extern "C" int umr_segsum(const void* x, void* out, int R, int reps, int64_t rep_stride, int64_t seg_stride, int C, int dtype_in,
                          int out_f32, int accumulate, umr_stream_t stream) {
    UMR_CHECK_ARG(x && out && R > 0 && reps > 0 && C > 0, "segsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for((int64_t)R * C)), b(256);
    if (dtype_in == UMR_BF16 && out_f32) hipLaunchKernelGGL((segsum_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)x, (float*)out, R, reps, rep_stride, seg_stride, C, accumulate);
    else if (dtype_in == UMR_BF16) hipLaunchKernelGGL((segsum_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)x, (bf16_t*)out, R, reps, rep_stride, seg_stride, C, accumulate);
    else if (dtype_in == UMR_F32) hipLaunchKernelGGL((segsum_kernel<float, float>), g, b, 0, s, (const float*)x, (float*)out, R, reps, rep_stride, seg_stride, C, accumulate);
    else return umr_set_error(UMR_ERR_INVALID, "segsum: dtype");
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_fill_cls(void* tokens, const float* cls, const float* pos0, int B, int64_t batch_stride, int D, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(tokens && cls && pos0 && B > 0 && D > 0, "fill_cls: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL(fill_cls_kernel<T>, dim3((B * D + 255) / 256), dim3(256), 0, s, (T*)tokens, cls, pos0, B, batch_stride, D));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

namespace {
__global__ void scale_dev_kernel(const float* __restrict__ src, const float* __restrict__ scalar, float* __restrict__ dst, int64_t n) {
    const float sc = *scalar;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i] * sc;
}
}  // namespace

extern "C" int umr_scale_by_device_scalar(const float* src, const float* scalar, float* dst, int64_t n, umr_stream_t stream) {
    UMR_CHECK_ARG(src && scalar && dst && n > 0, "scale_by_device_scalar: bad arguments");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, (hipStream_t)stream, src, scalar, dst, n);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_cast(const void* src, void* dst, int64_t n, float scale, int dtype_in, int dtype_out, umr_stream_t stream) {
    UMR_CHECK_ARG(src && dst && n > 0, "cast: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for(n, 256, 16384)), b(256);
    if (dtype_in == UMR_F32 && dtype_out == UMR_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, s, (const float*)src, (bf16_t*)dst, n, scale);
    else if (dtype_in == UMR_BF16 && dtype_out == UMR_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, (float*)dst, n, scale);
    else if (dtype_in == UMR_F32 && dtype_out == UMR_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, n, scale);
    else if (dtype_in == UMR_BF16 && dtype_out == UMR_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, n, scale);
    else return umr_set_error(UMR_ERR_INVALID, "cast: dtype");
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_split3_rows(const float* src, void* dst, int64_t rows, int K, int64_t ld_src, int64_t ld_dst, int rows_in, int rows_out,
                               int row_off, umr_stream_t stream) {
    UMR_CHECK_ARG(src && dst && rows > 0 && K > 0 && K % 4 == 0 && ld_src >= K && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_dst >= 3 * (int64_t)K && rows_in >= 0,
                  "split3: bad arguments (K, ld_src, ld_dst multiples of 4; ld_dst >= 3K)");
    hipLaunchKernelGGL(split3_kernel, dim3(grid_for(rows * (K / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, rows, K,
                       ld_src, ld_dst, rows_in, rows_out, row_off);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_unsplit3(const void* src, float* dst, int64_t rows, int K, int64_t ld_src, int64_t ld_dst, umr_stream_t stream) {
    UMR_CHECK_ARG(src && dst && rows > 0 && K > 0 && K % 4 == 0 && ld_src >= 3 * (int64_t)K && ld_src % 4 == 0 && ld_dst >= K && ld_dst % 4 == 0,
                  "unsplit3: bad arguments (K, ld_src, ld_dst multiples of 4; ld_src >= 3K)");
    hipLaunchKernelGGL(unsplit3_kernel, dim3(grid_for(rows * (K / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, rows, K,
                       ld_src, ld_dst);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_split3(const float* src, void* dst, int64_t rows, int K, int64_t ld_src, int64_t ld_dst, umr_stream_t stream) {
    return umr_split3_rows(src, dst, rows, K, ld_src, ld_dst, 0, 0, 0, stream);
}

extern "C" int umr_head_out_fwd(const void* h, const float* w, const float* bias, float* out, int64_t M, int K, int Cout, int HW, int act,
                                int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(h && w && bias && out && M > 0 && K > 0 && K % 4 == 0 && (Cout == 1 || Cout == 2) && HW > 0 && M % HW == 0, "head_out_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, hipLaunchKernelGGL(head_out_fwd_kernel<T>, dim3(grid_for((M + 3) / 4, 1, 16384)), dim3(256), 0, s, (const T*)h, w, bias, out, M, K, Cout, HW, act));
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_head_out_finish(const float* partials, int nparts, const float* bias, float* out, int64_t M, int Cout, int HW, int act,
                                   umr_stream_t stream) {
    UMR_CHECK_ARG(partials && bias && out && nparts > 0 && M > 0 && (Cout == 1 || Cout == 2) && HW > 0 && M % HW == 0, "head_out_finish: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(head_out_finish_kernel, dim3(grid_for(M * Cout, 256, 8192)), dim3(256), 0, s, partials, nparts, bias, out, M, Cout, HW, act);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

static int head_out_blocks(int64_t M) { int64_t nb = (M + 255) / 256; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1; return (int)nb; }

extern "C" int64_t umr_head_out_bwd_workspace(int64_t M, int K) { return (int64_t)head_out_blocks(M) * (2 * K + 2) * 4; }

extern "C" int umr_head_out_bwd(const void* h, const float* w, const float* dout, const float* yout, void* dh, float* dw, float* db,
                                void* workspace, int64_t workspace_bytes, int64_t M, int K, int Cout, int HW, int act, int relu_mask,
                                int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(h && w && dout && dh && dw && db && workspace, "head_out_bwd: null pointer");
    UMR_CHECK_ARG(M > 0 && K > 0 && K % 4 == 0 && K <= 1024 && (Cout == 1 || Cout == 2) && HW > 0 && M % HW == 0, "head_out_bwd: bad arguments");
    UMR_CHECK_ARG(act == UMR_ACT_NONE || yout, "head_out_bwd: activation needs the forward output");
    UMR_CHECK_ARG(workspace_bytes >= umr_head_out_bwd_workspace(M, K), "head_out_bwd: workspace too small");
    int nb = head_out_blocks(M);
    const int rpb = (int)((M + nb - 1) / nb);
    nb = (int)((M + rpb - 1) / rpb);
    hipStream_t s = (hipStream_t)stream;
    const bool generic_only = umr_opt(UMR_OPT_HEAD_OUT_BWD_GENERIC) != UMR_OPT_UNSET;   // A/B hook (tests compare both forms: umr_set_debug_option)
    if (dtype == UMR_BF16 && K == 1024 && act != 4 && !generic_only) {
        hipLaunchKernelGGL(head_out_bwd_k1024_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)h, w, dout, yout, (bf16_t*)dh, (float*)workspace, M, Cout, HW, act, relu_mask, rpb);
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL(head_out_bwd_kernel<T>, dim3(nb), dim3(256), 0, s, (const T*)h, w, dout, yout, (T*)dh, (float*)workspace, M, K, Cout, HW, act, relu_mask, rpb));
    }
    UMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(head_out_reduce_kernel, dim3((2 * K + 2 + 63) / 64), dim3(256), 0, s, (const float*)workspace, dw, db, nb, K, Cout);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int64_t umr_loss_workspace(void) { return 1024 * 4 * 4; }

extern "C" int umr_objectness_loss(const float* pred_center, const float* pred_sdf, const float* gt_center, const float* gt_sdf,
                                   const float* gt_saliency, float* d_center, float* d_sdf, float* out5, void* workspace, int B, int H,
                                   int W, int center_l2, int sdf_l2, int use_grad, int use_bce, float grad_scale, umr_stream_t stream) {
    UMR_CHECK_ARG(pred_center && pred_sdf && gt_center && gt_sdf && out5 && workspace, "loss: null pointer");
    UMR_CHECK_ARG(!use_bce || gt_saliency, "loss: bce term needs the saliency mask");
    UMR_CHECK_ARG(B > 0 && H > 0 && W > 0, "loss: empty");
    LossCfg cfg{B, H, W, center_l2, sdf_l2, use_grad, use_bce, grad_scale};
    const int64_t total = (int64_t)B * H * W;
    const int nb = grid_for(total, 256, 1024);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_kernel, dim3(nb), dim3(256), 0, s, pred_center, pred_sdf, gt_center, gt_sdf, gt_saliency, d_center, d_sdf, (float*)workspace, cfg);
    UMR_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, out5, nb);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

static void adam_hyper(float lr, float beta1, float beta2, float eps, int step, float grad_scale, float* hyper7) {
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hyper7[0] = lr; hyper7[1] = beta1; hyper7[2] = beta2; hyper7[3] = eps;
    hyper7[4] = (float)bc1; hyper7[5] = (float)sqrt(bc2); hyper7[6] = grad_scale;
}

namespace {
struct Hyper7 { float v[7]; };
__global__ void adam_set_hyper_kernel(float* dst, Hyper7 h) { if (threadIdx.x < 7) dst[threadIdx.x] = h.v[threadIdx.x]; }
}

extern "C" int umr_adam_set_hyper(float* hyper7_dev, float lr, float beta1, float beta2, float eps, int step, float grad_scale, umr_stream_t stream) {
    UMR_CHECK_ARG(hyper7_dev && step >= 1, "adam_set_hyper: bad arguments");
    Hyper7 h;
    adam_hyper(lr, beta1, beta2, eps, step, grad_scale, h.v);
    hipLaunchKernelGGL(adam_set_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper7_dev, h);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             int step, float grad_scale, umr_stream_t stream) {
    UMR_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam: bad arguments");
    float h[7];
    adam_hyper(lr, beta1, beta2, eps, step, grad_scale, h);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, s, p, g, m, v, n, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_adam_pack_step(const umr_adam_pack_entry* table_dev, int n_entries, int64_t total_blocks, const int32_t* blk_entry_dev,
                                  const float* hyper7_dev, umr_stream_t stream) {
    UMR_CHECK_ARG(table_dev && blk_entry_dev && hyper7_dev && n_entries > 0 && total_blocks > 0 && total_blocks < (1ll << 31), "adam_pack_step: bad arguments");
    hipLaunchKernelGGL(adam_pack_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const AdamPackEntry*)table_dev, blk_entry_dev,
                       hyper7_dev);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int umr_adam_step_hyper(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper7_dev, umr_stream_t stream) {
    UMR_CHECK_ARG(p && g && m && v && n > 0 && hyper7_dev, "adam_step_hyper: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_hyper_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, s, p, g, m, v, n, hyper7_dev);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
