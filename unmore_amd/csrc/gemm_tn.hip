// Weight-gradient GEMM ("TN", reduction over rows), gfx950 MFMA.
//   dW[N,K] (f32) = sum_m dY[m,N]^T . X[m,K]        (+ dbias[N] = sum_m dY[m,N])
// Both operands are row-major with the REDUCTION index m as the row, so neither can
// be read as an MFMA fragment directly.  Tiles are staged as they lie in memory
// ([rows m][128 columns], 16-byte LDS-DMA) and transposed on the way to the registers:
//   bf16: ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group -> column-major),
//         LDS rows of 256 B with chunk XOR ((r&3)<<2 | (r>>2)&3) on source + read;
//   f32 : ds_read_b32 (one element per lane, v_mfma_f32_16x16x4_f32).
// 128 (n) x 128 (k) outputs per 256-thread workgroup, 4 waves as 2x2; the row range is
// split over `splits` workgroups which write f32 slabs reduced in fixed order by a
// second kernel (bitwise reproducible, no float atomics).
// Implicit 3x3 conv: X row for (m, tap) = input pixel (oy*s+ky-1, ox*s+kx-1); the
// K index is (tap, ci), so dW comes out as [N][3][3][Cin].
// dbias rides along on workgroups with k-tile 0: one extra MFMA against an all-ones
// fragment per n-tile and k-step.
#include "umr_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int TN_BN = 128, TN_BK = 128;
constexpr int TN_LDS = 65536;

template <typename T> struct TnTr;
template <> struct TnTr<bf16_t> { static constexpr int ROWS = 64, ROWB = 256, RPI = 4; };  // rows/stage, bytes/row, rows per glds
template <> struct TnTr<float> { static constexpr int ROWS = 32, ROWB = 512, RPI = 2; };

__device__ __forceinline__ int tn_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// three-way bf16 split of 8 f32 values (see gemm_nt.hip: f32 products on the bf16 matrix cores)
__device__ __forceinline__ void tn_split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const bf16_t hh = (bf16_t)x[e];
        const float r = x[e] - (float)hh;
        const bf16_t mm = (bf16_t)r;
        h[e] = hh; m[e] = mm; l[e] = (bf16_t)(r - (float)mm);
    }
}

template <typename T, int CONV, bool X3 = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const umr_gemm_tn_desc p, int tiles_k, int rows_per_split,
                                                         float* slab, float* bslab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = TnTr<T>::ROWS, ROWB = TnTr<T>::ROWB, RPI = TnTr<T>::RPI;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int TILE = ROWS * ROWB;  // 16 KiB
    constexpr int STAGE = 2 * TILE;
    constexpr int CPR = ROWB / 16;  // chunks per row: 16 (bf16) / 32 (f32)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, split = blockIdx.y;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * TN_BN, k0 = tk * TN_BK;
    const int m_begin = split * rows_per_split;
    const int m_end = min(p.M, m_begin + rows_per_split);
    const bool do_bias = (p.dbias != nullptr) && (tk == 0);

    // ---- staging by buffer LDS-DMA (see gemm_nt.hip): per stage each wave issues 4 loads per operand;
    // instruction i of wave w covers rows (w*4+i)*RPI + lane/CPR, chunk position lane%CPR.  Descriptor bases are
    // advanced per stage with scalar arithmetic; per-lane voffsets are loop constants; row tails fall out of the
    // descriptor's num_records and conv halos / column tails use an out-of-range voffset (LDS gets zeros).
    constexpr unsigned OOB = 0x80000000u;
    constexpr int SZ = (int)sizeof(T);
    const int lrow = lane / CPR, lchk = lane % CPR;
    const int stride = (CONV == 2) ? 2 : 1;
    int rr[4], gch[4], k_tap[4], k_ci[4], tky[4], tkx[4];
    unsigned y_vo[4], x_vo[4];
    bool k_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (w * 4 + i) * RPI + lrow;
        rr[i] = r;
        gch[i] = (sizeof(T) == 2) ? (lchk ^ tn_swz(r)) : lchk;
        const bool n_ok = (n0 + gch[i] * EPC) < p.N;
        y_vo[i] = n_ok ? (unsigned)(((int64_t)r * p.lddy) * SZ + gch[i] * 16) : OOB;
        const int kc = k0 + gch[i] * EPC;
        k_ok[i] = kc < p.K;
        k_tap[i] = 0;
        k_ci[i] = kc;
        tky[i] = tkx[i] = 0;
        if (CONV != 0 && k_ok[i]) {
            k_tap[i] = kc / p.Cin;
            k_ci[i] = kc - k_tap[i] * p.Cin;
            tky[i] = k_tap[i] / 3;
            tkx[i] = k_tap[i] - tky[i] * 3 - 1;
            tky[i] -= 1;
        }
        if (CONV == 0) x_vo[i] = k_ok[i] ? (unsigned)(((int64_t)r * p.ldx) * SZ + gch[i] * 16) : OOB;
        else x_vo[i] = k_ok[i] ? (unsigned)(((int64_t)(r + (tky[i] + 1) * p.W + (tkx[i] + 1)) * p.Cin + k_ci[i]) * SZ) : OOB;
    }
    // fast conv path: stride 1 and every ROWS-row stage lies inside one image row (Wo % ROWS == 0)
    const bool conv_fast = (CONV == 1) && (p.Wo % ROWS == 0);
    const bool remap = (p.dy_rows_in > 0) || (CONV == 0 && p.x_rows_in > 0);
    // scalar pixel position of the stage's first row (conv)
    int sb = 0, soy = 0, sox = 0;
    if (CONV != 0) {
        const int hw = p.Ho * p.Wo;
        sb = m_begin / hw;
        const int rem = m_begin - sb * hw;
        soy = rem / p.Wo;
        sox = rem - soy * p.Wo;
    }
    // general conv path: per-lane pixel coordinates advanced incrementally (float-reciprocal carries)
    int cb[4], coy[4], cox[4];
    const float inv_wo = (CONV != 0) ? 1.0f / (float)p.Wo : 0.f, inv_ho = (CONV != 0) ? 1.0f / (float)p.Ho : 0.f;
    auto carry = [](int& x, int d, float inv) -> int {  // x < 2^22: q = x / d, x %= d
        int q = (int)((float)x * inv);
        int r = x - q * d;
        if (r >= d) { r -= d; ++q; } else if (r < 0) { r += d; --q; }
        x = r;
        return q;
    };
    if (CONV != 0 && !conv_fast) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cb[i] = sb; coy[i] = soy; cox[i] = sox + rr[i];
            coy[i] += carry(cox[i], p.Wo, inv_wo);
            cb[i] += carry(coy[i], p.Ho, inv_ho);
        }
    }

    auto stage = [&](int mbase, int buf) {
        char* sa = smem + buf * STAGE + w * 4096;  // dY tile
        char* sb_ = sa + TILE;                     // X tile
        const int rows_left = min(m_end - mbase, ROWS);
        unsigned vy[4], vx[4];
        const char* ybase;
        const char* xbase;
        unsigned yrec = 0x7FFFFFFFu, xrec = 0x7FFFFFFFu;
        if (!remap) {
            ybase = (const char*)p.dY + ((int64_t)mbase * p.lddy + n0) * SZ;
            yrec = (unsigned)(rows_left * p.lddy * SZ);
#pragma unroll
            for (int i = 0; i < 4; ++i) vy[i] = y_vo[i];
        } else {
            ybase = (const char*)p.dY + (int64_t)n0 * SZ;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mbase + rr[i];
                int my = m;
                if (p.dy_rows_in > 0) my = (m / p.dy_rows_in) * p.dy_rows_out + p.dy_row_off + (m % p.dy_rows_in);
                vy[i] = (m < m_end && y_vo[i] != OOB) ? (unsigned)(((int64_t)my * p.lddy) * SZ + gch[i] * 16) : OOB;
            }
        }
        if (CONV == 0) {
            if (!remap) {
                xbase = (const char*)p.X + ((int64_t)mbase * p.ldx + k0) * SZ;
                xrec = (unsigned)(rows_left * p.ldx * SZ);
#pragma unroll
                for (int i = 0; i < 4; ++i) vx[i] = x_vo[i];
            } else {
                xbase = (const char*)p.X + (int64_t)k0 * SZ;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mbase + rr[i];
                    int mx = m;
                    if (p.x_rows_in > 0) mx = (m / p.x_rows_in) * p.x_rows_out + p.x_row_off + (m % p.x_rows_in);
                    vx[i] = (m < m_end && k_ok[i]) ? (unsigned)(((int64_t)mx * p.ldx) * SZ + gch[i] * 16) : OOB;
                }
            }
        } else if (conv_fast) {
            // all ROWS rows of this stage: image sb, output row soy, columns sox .. sox+ROWS-1
            xbase = (const char*)p.X + ((((int64_t)sb * p.H + soy) * p.W + sox) - (p.W + 1)) * p.Cin * SZ;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (unsigned)(soy + tky[i]) < (unsigned)p.H && (unsigned)(sox + rr[i] + tkx[i]) < (unsigned)p.W &&
                                rr[i] < rows_left;
                vx[i] = ok ? x_vo[i] : OOB;
            }
            sox += ROWS;
            if (sox >= p.Wo) { sox = 0; if (++soy >= p.Ho) { soy = 0; ++sb; } }
        } else {
            // general path: per-lane coordinates, addressed relative to the stage's first input pixel (minus one
            // halo row + column) so that the 32-bit voffset stays small for tensors of any size
            const int64_t pix_base = ((int64_t)sb * p.H + soy * stride) * p.W + sox * stride - (p.W + 1);
            xbase = (const char*)p.X + pix_base * p.Cin * SZ;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iy = coy[i] * stride + tky[i], ix = cox[i] * stride + tkx[i];
                const bool ok = k_ok[i] && rr[i] < rows_left && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const int64_t pix = ((int64_t)cb[i] * p.H + iy) * p.W + ix;
                vx[i] = ok ? (unsigned)(((pix - pix_base) * p.Cin + k_ci[i]) * SZ) : OOB;
                cox[i] += ROWS;
                coy[i] += carry(cox[i], p.Wo, inv_wo);
                cb[i] += carry(coy[i], p.Ho, inv_ho);
            }
            sox += ROWS;
            while (sox >= p.Wo) { sox -= p.Wo; ++soy; }
            while (soy >= p.Ho) { soy -= p.Ho; ++sb; }
        }
        const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, yrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, xrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, UMR_LDS_PTR(sa + i * 1024), 16, vy[i], 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, UMR_LDS_PTR(sb_ + i * 1024), 16, vx[i], 0, 0, 0);
        }
    };

    f32x4 acc[4][4];
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int wn = w >> 1, wk = w & 1;
    const int g = lane >> 4, li = lane & 15;

    const int nstages = (m_end > m_begin) ? (m_end - m_begin + ROWS - 1) / ROWS : 0;
    if (nstages > 0) stage(m_begin, 0);
    for (int t = 0; t < nstages; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < nstages) stage(m_begin + (t + 1) * ROWS, (t + 1) & 1);
        const char* sA = smem + (t & 1) * STAGE;
        const char* sB = sA + TILE;
        if constexpr (sizeof(T) == 2) {
            // lane 4q+pp of a 16-lane group addresses row q, columns 4pp..4pp+3 of the block
            const int q = li >> 2, pp = li & 3;
            // The transposed reads are inline asm (see gemm_tn256.hip): through the builtin the compiler orders them
            // behind the LDS-DMA of the NEXT stage issued just above (s_waitcnt vmcnt(0)), which serialises the
            // prefetch.  All 32 reads of the stage are issued, then consumed in two halves under counted lgkmcnt.
            const unsigned lbase = (unsigned)(uintptr_t)UMR_LDS_PTR(sA);
            u32x2 ra[2][4][2], rb[2][4][2];
            auto issue_reads = [&](auto kstag) {
                constexpr int ks = decltype(kstag)::value;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int r = ks * 32 + g * 8 + half * 4 + q;
                    const int sw = tn_swz(r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ca = (wn * 64 + i * 16) / 8 + (pp >> 1);
                        const int cb = (wk * 64 + i * 16) / 8 + (pp >> 1);
                        const unsigned pa = lbase + r * ROWB + ((ca ^ sw) << 4) + ((pp & 1) << 3);
                        const unsigned pb = lbase + TILE + r * ROWB + ((cb ^ sw) << 4) + ((pp & 1) << 3);
                        u32x2 ta, tb;
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ta) : "v"(pa));
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tb) : "v"(pb));
                        ra[ks][i][half] = ta;
                        rb[ks][i][half] = tb;
                    }
                }
            };
            auto mfmas = [&](auto kstag) {
                constexpr int ks = decltype(kstag)::value;
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 ua = {ra[ks][i][0][0], ra[ks][i][0][1], ra[ks][i][1][0], ra[ks][i][1][1]};
                    const u32x4 ub = {rb[ks][i][0][0], rb[ks][i][0][1], rb[ks][i][1][0], rb[ks][i][1][1]};
                    fa[i] = __builtin_bit_cast(bf16x8, ua);
                    fb[i] = __builtin_bit_cast(bf16x8, ub);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kt], fa[nt], acc[nt][kt], 0, 0, 0);
                if (do_bias) {
                    bf16x8 ones;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[nt], accb[nt], 0, 0, 0);
                }
            };
            // reads(ks0); wait; reads(ks1) fly behind MFMAs(ks0); wait; MFMAs(ks1)   (lgkmcnt is a 4-bit counter)
            issue_reads(std::integral_constant<int, 0>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            issue_reads(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            mfmas(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mfmas(std::integral_constant<int, 1>{});
        } else if constexpr (X3) {
            // f32 operands, products on the bf16 matrix cores (three-way splits, six MFMAs per block: gemm_nt.hip).  The 32 rows
            // of a stage are ONE k-step: lane group g, element e <-> stage row 8 g + e, the same map for both operands.
            bf16x8 ah[4], am[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = *(const float*)(sA + (g * 8 + e) * ROWB + (wn * 64 + i * 16 + li) * 4);
                tn_split3(v, ah[i], am[i], al[i]);
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = *(const float*)(sB + (g * 8 + e) * ROWB + (wk * 64 + kt * 16 + li) * 4);
                bf16x8 bh, bm, bl;
                tn_split3(v, bh, bm, bl);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4 c = acc[nt][kt];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am[nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah[nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am[nt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[nt], c, 0, 0, 0);
                    acc[nt][kt] = c;
                }
            }
            if (do_bias) {
                bf16x8 ones;
#pragma unroll
                for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, al[nt], accb[nt], 0, 0, 0);
                    accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, am[nt], accb[nt], 0, 0, 0);
                    accb[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ah[nt], accb[nt], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < ROWS / 4; ++ks) {
                const int r = ks * 4 + g;
                float fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = *(const float*)(sA + r * ROWB + (wn * 64 + i * 16 + li) * 4);
                    fb[i] = *(const float*)(sB + r * ROWB + (wk * 64 + i * 16 + li) * 4);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[kt], fa[nt], acc[nt][kt], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        accb[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, fa[nt], accb[nt], 0, 0, 0);
                }
            }
        }
    }

    // D[i = k_local][j = n_local]: lane holds n = li, k = 4*g + reg
    float* out = slab + (int64_t)split * p.N * p.K;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + li;
        if (n >= p.N) continue;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k = k0 + wk * 64 + kt * 16 + g * 4;
            if (k >= p.K) continue;
            float* o = out + (int64_t)n * p.K + k;
            if (k + 3 < p.K && (p.K & 3) == 0) *(f32x4*)o = acc[nt][kt];
            else { for (int e = 0; e < 4; ++e) if (k + e < p.K) o[e] = acc[nt][kt][e]; }
        }
        if (do_bias && wk == 0 && g == 0) bslab[(int64_t)split * p.N + n] = accb[nt][0];
    }
}

// Fixed-order sum of the split slabs (deterministic).  VEC = 4: four consecutive k per thread as 16-byte accesses
// (K, lddw multiples of 4); the split loop keeps four independent loads in flight.
template <int VEC>
__global__ void tn_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dW, int64_t lddw, int N, int K,
                                 int splits, int accumulate, const float* __restrict__ bslab, float* __restrict__ dbias, int bias_parts) {
    const int64_t total = (int64_t)N * K;
    const int64_t nvec = total / VEC;
    for (int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; iv < nvec; iv += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = iv * VEC;
        const int n = (int)(i / K), k = (int)(i - (int64_t)n * K);
        float* o = dW + (int64_t)n * lddw + k;
        if (VEC == 4) {
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
            const float* q = slab + i;
            int sp = 0;
            for (; sp + 4 <= splits; sp += 4) {
                s0 += *(const f32x4*)(q + (int64_t)(sp + 0) * total);
                s1 += *(const f32x4*)(q + (int64_t)(sp + 1) * total);
                s2 += *(const f32x4*)(q + (int64_t)(sp + 2) * total);
                s3 += *(const f32x4*)(q + (int64_t)(sp + 3) * total);
            }
            for (; sp < splits; ++sp) s0 += *(const f32x4*)(q + (int64_t)sp * total);
            f32x4 r = (s0 + s1) + (s2 + s3);
            if (accumulate) r += *(const f32x4*)o;
            *(f32x4*)o = r;
        } else {
            float sacc = 0.f;
            for (int sp = 0; sp < splits; ++sp) sacc += slab[(int64_t)sp * total + i];
            *o = accumulate ? (*o + sacc) : sacc;
        }
    }
    if (dbias != nullptr) {
        for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;   // four independent chains: the loop is latency-bound
            int sp = 0;
            for (; sp + 4 <= bias_parts; sp += 4) {
                b0 += bslab[(int64_t)(sp + 0) * N + n];
                b1 += bslab[(int64_t)(sp + 1) * N + n];
                b2 += bslab[(int64_t)(sp + 2) * N + n];
                b3 += bslab[(int64_t)(sp + 3) * N + n];
            }
            for (; sp < bias_parts; ++sp) b0 += bslab[(int64_t)sp * N + n];
            const float sacc = (b0 + b1) + (b2 + b3);
            dbias[n] = accumulate ? (dbias[n] + sacc) : sacc;
        }
    }
}

struct TnPlan { int tiles_n, tiles_k, splits, rows_per_split; int64_t ws; bool big; };

}  // namespace
bool umr_tn256_eligible(const umr_gemm_tn_desc* d, bool force);                                                  // gemm_tn256.hip
void umr_tn256_plan(const umr_gemm_tn_desc* d, int* splits, int* rows_per_split);
int umr_launch_gemm_tn256(const umr_gemm_tn_desc* d, int splits, int rows_per_split, float* slab, float* bslab, hipStream_t s);
namespace {

static int tn_tile_override() {  // UMR_GEMM_TILE=128|256 forces a tile size (benchmarking, tests), as umr_gemm_nt does
    return umr_opt_or(UMR_OPT_GEMM_TILE, 0);
}

TnPlan tn_plan(const umr_gemm_tn_desc* d) {
    TnPlan pl;
    pl.big = tn_tile_override() != 128 && umr_tn256_eligible(d, tn_tile_override() == 256);
    if (pl.big) {
        pl.tiles_n = (d->N + 255) / 256;
        pl.tiles_k = (d->K + 255) / 256;
        umr_tn256_plan(d, &pl.splits, &pl.rows_per_split);
        pl.ws = ((int64_t)pl.splits * d->N * d->K + (int64_t)pl.splits * pl.tiles_k * d->N) * 4;  // bias partials per (split, k-tile)
        return pl;
    }
    pl.tiles_n = (d->N + TN_BN - 1) / TN_BN;
    pl.tiles_k = (d->K + TN_BK - 1) / TN_BK;
    const int64_t tiles = (int64_t)pl.tiles_n * pl.tiles_k;
    const int rows = d->dtype == UMR_BF16 ? 64 : 32;
    // 256 CUs x 2 resident workgroups = 512 slots: aim just under 8 full rounds so the last round is full
    int64_t want = 4096 / tiles;
    int64_t max_by_rows = ((int64_t)d->M + rows * 8 - 1) / (rows * 8);  // >= 8 stages per split
    // Short reductions (M <= 4096: the reference recipe's 1300 tokens): about ONE workgroup per CU in all, splits of >= 4 stages.  The
    // weight-gradient lane of that step is bound by its traffic and by the chip it shares with the data-gradient chain, not by one
    // launch's latency: three splits of every weight (the rule above) wrote and re-read 8 GB of slabs per step; with ~256 workgroups
    // per launch the 192- and 256-tile weights take one split (and no reduce pass, below), the 64-tile ones four
    // (same box: 922.5 -> 945 images/s; UMR_TN_PLAN=<workgroups> to try another target, =-1 the old rule)
    static const int plan_env = [] { const char* e = getenv("UMR_TN_PLAN"); return e ? atoi(e) : 0; }();
    if (plan_env >= 0 && d->M <= 4096) {
        const int target = plan_env > 0 ? plan_env : 256;
        want = (target + tiles / 2) / tiles;
        max_by_rows = ((int64_t)d->M + rows * 4 - 1) / (rows * 4);
    }
    if (want > max_by_rows) want = max_by_rows;
    if (want < 1) want = 1;
    if (want > 4096) want = 4096;
    int64_t rps = ((int64_t)d->M + want - 1) / want;
    rps = (rps + rows - 1) / rows * rows;
    pl.rows_per_split = (int)rps;
    pl.splits = (int)(((int64_t)d->M + rps - 1) / rps);
    pl.ws = ((int64_t)pl.splits * d->N * d->K + (int64_t)pl.splits * d->N) * 4;
    return pl;
}

}  // namespace

extern "C" int64_t umr_gemm_tn_workspace(const umr_gemm_tn_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    return tn_plan(d).ws;
}

extern "C" int umr_gemm_tn(const umr_gemm_tn_desc* d, umr_stream_t stream) {
    UMR_CHECK_ARG(d != nullptr, "gemm_tn: null descriptor");
    UMR_CHECK_ARG(d->dY && d->X && d->dW && d->workspace, "gemm_tn: null operand");
    UMR_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gemm_tn: empty problem");
    UMR_CHECK_ARG(d->dtype == UMR_F32 || d->dtype == UMR_BF16 || d->dtype == UMR_BF16X3, "gemm_tn: dtype");
    const int epc = d->dtype == UMR_F32 ? 4 : 8;
    if (d->dtype == UMR_BF16X3) {
        // f32 values as three bf16 planes per row (include/umr.h): the 256x256 kernel only
        if (d->dy_rows_in > 0 || d->x_rows_in > 0 || d->conv == 2)
            return umr_set_error(UMR_ERR_UNSUPPORTED, "gemm_tn (BF16X3): no row remaps, no stride-2 conv");
        UMR_CHECK_ARG(d->lddy >= 3 * (int64_t)d->N && (d->conv != 0 || d->ldx >= 3 * (int64_t)d->K), "gemm_tn (BF16X3): rows hold three planes (lddy >= 3N, ldx >= 3K)");
    }
    UMR_CHECK_ARG(d->N % epc == 0 && d->lddy % epc == 0, "gemm_tn: N/lddy must be multiples of 16 bytes");
    UMR_CHECK_ARG(d->conv >= 0 && d->conv <= 2, "gemm_tn: conv mode");
    if (d->conv == 0) {
        UMR_CHECK_ARG(d->K % epc == 0 && d->ldx % epc == 0, "gemm_tn: K/ldx must be multiples of 16 bytes");
    } else {
        UMR_CHECK_ARG(d->Cin % epc == 0 && d->K == 9 * d->Cin, "gemm_tn: conv needs Cin % 16B == 0 and K == 9*Cin");
        UMR_CHECK_ARG((int64_t)d->nb * d->Ho * d->Wo == d->M, "gemm_tn: conv M != nb*Ho*Wo");
        const int s = d->conv == 2 ? 2 : 1;
        UMR_CHECK_ARG(d->Ho == (d->H - 1) / s + 1 && d->Wo == (d->W - 1) / s + 1, "gemm_tn: conv output size");
    }
    {   // paths that address with absolute 32-bit offsets from the tensor base
        const int rows = d->dtype == UMR_F32 ? 32 : 64;
        const int64_t sz = d->dtype == UMR_F32 ? 4 : 2;
        const bool remap = d->dy_rows_in > 0 || (d->conv == 0 && d->x_rows_in > 0);
        const bool conv_general = d->conv != 0 && !(d->conv == 1 && d->Wo % rows == 0);
        (void)conv_general;  // both conv paths address relative to a per-stage base: no size limit
        if (remap) {
            const int64_t ymax = d->dy_rows_in > 0 ? ((int64_t)d->M / d->dy_rows_in + 1) * d->dy_rows_out : d->M;
            const int64_t xmax = d->x_rows_in > 0 ? ((int64_t)d->M / d->x_rows_in + 1) * d->x_rows_out : d->M;
            if (ymax * d->lddy * sz >= (1ll << 31) || (d->conv == 0 && xmax * d->ldx * sz >= (1ll << 31)))
                return umr_set_error(UMR_ERR_UNSUPPORTED, "gemm_tn: row-remapped operands must be < 2 GiB");
        }
    }
    const TnPlan pl = tn_plan(d);
    UMR_CHECK_ARG(d->workspace_bytes >= pl.ws, "gemm_tn: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* slab = (float*)d->workspace;
    float* bslab = slab + (int64_t)pl.splits * d->N * d->K;
    // one split of the 128x128 kernel, nothing to add to: its "slab" IS dW (and its bias partials dbias) -- no reduce pass
    const bool direct = !pl.big && pl.splits == 1 && !d->accumulate && d->lddw == d->K &&
                        ((uintptr_t)d->dW & 15) == 0;   // the kernel stores f32x4 (the reduce pass it replaces checks the same)
    if (direct) { slab = d->dW; bslab = d->dbias; }
    if (pl.big) {
        const int st = umr_launch_gemm_tn256(d, pl.splits, pl.rows_per_split, slab, bslab, s);
        if (st != UMR_OK) return st;
    } else {
    dim3 g((unsigned)(pl.tiles_n * pl.tiles_k), (unsigned)pl.splits), b(256);
#define LAUNCH(T, CV) hipLaunchKernelGGL((gemm_tn_kernel<T, CV>), g, b, TN_LDS, s, *d, pl.tiles_k, pl.rows_per_split, slab, bslab)
#define LAUNCH_X3(CV) hipLaunchKernelGGL((gemm_tn_kernel<float, CV, true>), g, b, TN_LDS, s, *d, pl.tiles_k, pl.rows_per_split, slab, bslab)
    const int f32_x3 = umr_f32_mode_now() != UMR_F32_EXACT;   // include/umr.h: umr_set_f32_mode (default X3; UMR_F32_X3=0 selects the f32 MFMA)
    if (d->dtype == UMR_BF16) {
        if (d->conv == 0) LAUNCH(bf16_t, 0); else if (d->conv == 1) LAUNCH(bf16_t, 1); else LAUNCH(bf16_t, 2);
    } else if (f32_x3) {
        if (d->conv == 0) LAUNCH_X3(0); else if (d->conv == 1) LAUNCH_X3(1); else LAUNCH_X3(2);
    } else {
        if (d->conv == 0) LAUNCH(float, 0); else if (d->conv == 1) LAUNCH(float, 1); else LAUNCH(float, 2);
    }
#undef LAUNCH_X3
#undef LAUNCH
    UMR_LAUNCH_CHECK();
    }
    if (direct) return UMR_OK;
    const int64_t total = (int64_t)d->N * d->K;
    const bool v4 = (d->K % 4 == 0) && (d->lddw % 4 == 0) && (((uintptr_t)d->dW & 15) == 0);
    int rb = (int)((total / (v4 ? 4 : 1) + 255) / 256);
    if (rb > 8192) rb = 8192;
    if (rb < 1) rb = 1;
    if (v4)
        hipLaunchKernelGGL(tn_reduce_kernel<4>, dim3(rb), dim3(256), 0, s, slab, d->dW, d->lddw, d->N, d->K, pl.splits,
                           d->accumulate, bslab, d->dbias, pl.big ? pl.splits * pl.tiles_k : pl.splits);
    else
        hipLaunchKernelGGL(tn_reduce_kernel<1>, dim3(rb), dim3(256), 0, s, slab, d->dW, d->lddw, d->N, d->K, pl.splits,
                           d->accumulate, bslab, d->dbias, pl.big ? pl.splits * pl.tiles_k : pl.splits);
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
