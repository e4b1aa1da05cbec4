// NT GEMM / implicit 3x3 convolution with fused epilogue, gfx950 MFMA.
//   C[M,N] = epi(A[M,K] . B[N,K]^T)
// 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave),
// K-tiles of 128 bytes per row (64 bf16 / 32 f32) staged global -> LDS by
// 16-byte LDS-DMA (global_load_lds_dwordx4), double buffered, one barrier per K-tile.
// LDS image: [128 rows][8 x 16-B chunks]; chunk c of row r is stored at chunk position
// c ^ ((r>>1)&7) (pre-swizzled on the SOURCE address, same XOR on the ds_read_b128),
// which makes the 16-lane ds_read_b128 groups conflict-free.
// Operand roles are swapped (MFMA "A" = weight rows, "B" = activation rows) so that a
// lane's 4 accumulator registers are 4 consecutive output columns of one output row:
// 8/16-byte epilogue stores, bias/aux loads vectorised the same way.
// Implicit conv: the A row of output pixel m for tap (ky,kx) is the input pixel row
// (iy,ix) = (oy*s+ky-1, ox*s+kx-1); out-of-image rows, K tails and M/N tails read a zero page.
#include "umr_common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;
constexpr int TILE_BYTES = BM * ROWB;       // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES; // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // double buffered: 64 KiB -> 2 workgroups / CU

template <typename T> struct Tr;
template <> struct Tr<bf16_t> { static constexpr int EPC = 8, BK = 64; };
template <> struct Tr<float> { static constexpr int EPC = 4, BK = 32; };

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
        else { for (int j = 0; j < 4; ++j) v[j] *= dgelu_erf(a[j]); }
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
    else if (p.act == UMR_ACT_GELU) { for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]); }
    else if (p.act == UMR_ACT_TANH) { for (int j = 0; j < 4; ++j) v[j] = tanhf(v[j]); }
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

template <typename T, int CONV>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const umr_gemm_desc p, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int EPC = Tr<T>::EPC, BK = Tr<T>::BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware, bijective block -> tile map: blocks that share an XCD (id % 8) take a
    // contiguous run of tiles; consecutive tiles share the A row-panel (L2 reuse).
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const char* zero = (const char*)umr_zero_page;
    const int lrow = lane >> 3, lchk = lane & 7;
    const char* a_ptr[4];
    const char* b_ptr[4];
    bool a_ok[4], b_ok[4];
    int a_y[4], a_x[4], gch[4];
    const int stride = (CONV == 2) ? 2 : 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (w * 4 + i) * 8 + lrow;
        gch[i] = lchk ^ (((i & 1) << 2) + (lane >> 4));
        const int m = m0 + r;
        a_ok[i] = m < p.M;
        if (CONV == 0) {
            int ar = m;
            if (p.a_rows_in > 0) ar = (m / p.a_rows_in) * p.a_rows_out + p.a_row_off + (m % p.a_rows_in);
            a_ptr[i] = (const char*)p.A + (int64_t)ar * p.lda * (int64_t)sizeof(T);
            a_y[i] = a_x[i] = 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_ptr[i] = (const char*)p.A + (int64_t)b * p.H * p.W * p.Cin * (int64_t)sizeof(T);
            a_y[i] = oy * stride - 1;
            a_x[i] = ox * stride - 1;
        }
        const int n = n0 + r;
        b_ok[i] = n < p.N;
        b_ptr[i] = (const char*)p.B + (int64_t)n * p.ldb * (int64_t)sizeof(T);
    }

    const int ktiles_per_tap = (CONV == 0) ? 0 : (p.Cin + BK - 1) / BK;
    const int nt = (CONV == 0) ? (p.K + BK - 1) / BK : 9 * ktiles_per_tap;

    auto stage = [&](int t, int buf) {
        char* sa = smem + buf * STAGE_BYTES + w * 4096;
        char* sb = sa + TILE_BYTES;
        int tap = 0, c0 = t * BK, ky = 0, kx = 0;
        if (CONV != 0) {
            tap = t / ktiles_per_tap;
            c0 = (t - tap * ktiles_per_tap) * BK;
            ky = tap / 3;
            kx = tap - ky * 3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = c0 + gch[i] * EPC;
            const char* srcA;
            const char* srcB;
            if (CONV == 0) {
                const bool okk = kk < p.K;
                srcA = (a_ok[i] && okk) ? a_ptr[i] + (int64_t)kk * sizeof(T) : zero;
                srcB = (b_ok[i] && okk) ? b_ptr[i] + (int64_t)kk * sizeof(T) : zero;
            } else {
                const bool okc = kk < p.Cin;
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                const bool oka = a_ok[i] && okc && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                srcA = oka ? a_ptr[i] + ((int64_t)(iy * p.W + ix) * p.Cin + kk) * (int64_t)sizeof(T) : zero;
                srcB = (b_ok[i] && okc) ? b_ptr[i] + (int64_t)(tap * p.Cin + kk) * sizeof(T) : zero;
            }
            glds16(srcA, sa + i * 1024);
            glds16(srcB, sb + i * 1024);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wr = w >> 1, wc = w & 1;
    const int frow = lane & 15, fq = lane >> 4;
    // row byte offsets and swizzle keys of this lane's fragment rows
    int a_off[4], b_off[4], a_sw[4], b_sw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + frow, rb = wc * 64 + i * 16 + frow;
        a_off[i] = ra * ROWB;
        a_sw[i] = (ra >> 1) & 7;
        b_off[i] = TILE_BYTES + rb * ROWB;
        b_sw[i] = (rb >> 1) & 7;
    }

    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
        const char* sbuf = smem + (t & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int q = ks * 4 + fq;
            if constexpr (sizeof(T) == 2) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = *(const bf16x8*)(sbuf + a_off[i] + ((q ^ a_sw[i]) << 4));
                    bfr[i] = *(const bf16x8*)(sbuf + b_off[i] + ((q ^ b_sw[i]) << 4));
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int ntl = 0; ntl < 4; ++ntl)
                        acc[mt][ntl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ntl], af[mt], acc[mt][ntl], 0, 0, 0);
            } else {
                f32x4 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = *(const f32x4*)(sbuf + a_off[i] + ((q ^ a_sw[i]) << 4));
                    bfr[i] = *(const f32x4*)(sbuf + b_off[i] + ((q ^ b_sw[i]) << 4));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int ntl = 0; ntl < 4; ++ntl)
                            acc[mt][ntl] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[ntl][j], af[mt][j], acc[mt][ntl], 0, 0, 0);
            }
        }
    }

    // epilogue: lane holds row m = .. + (lane&15), columns n = .. + (lane>>4)*4 + {0..3}
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wr * 64 + mt * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) {
            const int n = n0 + wc * 64 + ntl * 16 + fq * 4;
            if (n >= p.N) continue;
            epilogue_store<T>(p, m, n, acc[mt][ntl]);
        }
    }
}

}  // namespace

extern "C" int umr_gemm_nt(const umr_gemm_desc* d, umr_stream_t stream) {
    UMR_CHECK_ARG(d != nullptr, "gemm_nt: null descriptor");
    UMR_CHECK_ARG(d->A && d->B && d->C, "gemm_nt: null operand");
    UMR_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gemm_nt: empty problem");
    UMR_CHECK_ARG(d->dtype == UMR_F32 || d->dtype == UMR_BF16, "gemm_nt: dtype");
    const int epc = d->dtype == UMR_BF16 ? 8 : 4;
    UMR_CHECK_ARG(d->conv >= 0 && d->conv <= 2, "gemm_nt: conv mode");
    if (d->conv == 0) {
        UMR_CHECK_ARG(d->K % epc == 0 && d->lda % epc == 0 && d->ldb % epc == 0, "gemm_nt: K/lda/ldb must be multiples of 16 bytes");
    } else {
        UMR_CHECK_ARG(d->Cin % epc == 0 && d->K == 9 * d->Cin && d->ldb % epc == 0, "gemm_nt: conv needs Cin % 16B == 0 and K == 9*Cin");
        UMR_CHECK_ARG((int64_t)d->nb * d->Ho * d->Wo == d->M, "gemm_nt: conv M != nb*Ho*Wo");
        const int s = d->conv == 2 ? 2 : 1;
        UMR_CHECK_ARG(d->Ho == (d->H - 1) / s + 1 && d->Wo == (d->W - 1) / s + 1, "gemm_nt: conv output size");
    }
    UMR_CHECK_ARG(!(d->flags & UMR_EPI_BIAS) || d->bias, "gemm_nt: bias flag without pointer");
    UMR_CHECK_ARG(!(d->flags & (UMR_EPI_ADD_AUX | UMR_EPI_MASK_RELU | UMR_EPI_MASK_DGELU)) || d->aux, "gemm_nt: aux flag without pointer");
    UMR_CHECK_ARG(!(d->flags & UMR_EPI_ADD_AUX2) || d->aux2, "gemm_nt: aux2 flag without pointer");
    UMR_CHECK_ARG(!(d->flags & UMR_EPI_ROWBIAS) || (d->rowbias && d->rows_per_batch > 0), "gemm_nt: rowbias");
    UMR_CHECK_ARG(d->c2_mode == 0 || d->C2, "gemm_nt: c2_mode without C2");
    const int tiles_m = (d->M + BM - 1) / BM, tiles_n = (d->N + BN - 1) / BN;
    const int64_t grid = (int64_t)tiles_m * tiles_n;
    UMR_CHECK_ARG(grid < (1ll << 31), "gemm_nt: grid too large");
    hipStream_t s = (hipStream_t)stream;
    dim3 g((unsigned)grid), b(256);
#define LAUNCH(T, CV) hipLaunchKernelGGL((gemm_nt_kernel<T, CV>), g, b, LDS_BYTES, s, *d, tiles_n)
    if (d->dtype == UMR_BF16) {
        if (d->conv == 0) LAUNCH(bf16_t, 0); else if (d->conv == 1) LAUNCH(bf16_t, 1); else LAUNCH(bf16_t, 2);
    } else {
        if (d->conv == 0) LAUNCH(float, 0); else if (d->conv == 1) LAUNCH(float, 1); else LAUNCH(float, 2);
    }
#undef LAUNCH
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
