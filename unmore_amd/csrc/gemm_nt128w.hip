// 128x256-tile bf16 NT GEMM, TWO workgroups per CU (the epilogue-overlap form for short-K GEMMs).
//
// Why: the persistent 256x256 kernel (gemm_nt256p.hip) owns a whole CU with one 512-thread workgroup; at K <= 1024 its
// epilogue (6.7 k cycles, nothing overlaps it: both waves of every SIMD are in it together) is 17-30 % of a tile -- the ViT
// GEMMs (K = 768) and the heads' 1x1 layers (K = 256 / 512 / 1024) run at 38-46 % of the matrix peak although their K loop
// keeps the pipe 74 % busy.  Here a workgroup is 256 threads = ONE wave per SIMD with a 128x256 tile and <= 80 KiB of LDS,
// so two workgroups share a CU: the waves of a SIMD belong to different workgroups, are not coupled by barriers, and one
// workgroup's epilogue, barrier waits and first-load latency sit beside the other's MFMAs.
//
// Geometry: 4 waves as 1 (M) x 4 (N); a wave owns 128x64 = 8x4 tiles of v_mfma_f32_16x16x32_bf16 (128 accumulator VGPRs,
// the same per-wave tile as the 256x256 kernels).  BK = 32 (a 64-wide K-tile would need 2 x 48 KiB of LDS): one MFMA k-step per
// stage, 32 MFMAs per wave and barrier.  LDS: 3-stage ring of (A 128 rows + B 256 rows) x 64 B = 24 KiB per stage = 72 KiB;
// rows are 64 B, 16-byte chunk c of row r is stored at slot c ^ T[(r >> 2) & 3], T = {0,3,2,1}: the four lane groups of a
// ds_read_b128 (MI355X_MICROARCH.md, LDS) then hit 16 distinct 16-byte slots of the 256-byte bank row (checked by hand for all
// four groups).  The swizzle is applied on the GLOBAL side of the LDS-DMA (the LDS side of buffer_load ... lds is lane-linear).
//
// Pipeline (per wave): fragments are double-buffered in registers (BK = 32 makes a fragment set 48 VGPRs: 128 + 2 x 48 fit),
// the LDS-DMA runs three stages ahead:
//     iteration t:  s_waitcnt vmcnt(6)      stage t+1 has landed (this wave's part; stage t+2's six DMAs may fly)
//                   s_waitcnt lgkmcnt(0)    fragments of stage t have arrived (read during iteration t-1)
//                   s_barrier               everyone's part of stage t+1 is in LDS; everyone is done reading slot t % 3
//                   12 ds_read_b128         fragments of stage t+1 into the other register set
//                   32 MFMAs on stage t, the six DMA instructions of stage t+3 (into slot t % 3) spread between them
// One barrier per stage; no wait inside the loop ever drains the DMA queue.  Stages past the end of K are issued too (the
// descriptor bounds make them read zeros or dead padding; nobody consumes them) so that the counted wait is uniform.
//
// MEASURED (round 3, tools/nt128w_bench.py, profiles/r03_nt128w_ab.txt, interleaved rounds in one process): bit-identical to the
// persistent kernel and 11-23 % SLOWER on every short-K GEMM of the step (heads 1x1: 256->512 -19 %, 512->1024 + fused output -17 %
// storing / -11 % not storing, masked 1024->512 -18 %; ViT qkv -14 %, proj -13 %, fc1 dgrad -23 %).  A stage takes ~2400 cycles per
// workgroup for 512 cycles of MFMA work per wave.  Reason: a 128x256 tile needs 1 byte of LDS fill per 85 FLOP against 1 per 128 for
// 256x256, and the L2 -> LDS fill (~70 GB/s = 29 B/clk per CU, MI355X_MICROARCH.md "Indexed rows: gather into LDS") is what bounds
// the 256x256 kernel's K loop already (32 B/clk needed at full MFMA rate): two co-resident workgroups need 47 B/clk.  What the
// partner workgroup hides (epilogue, barrier skew) is less than what the smaller tile costs.  The kernel therefore is OFF by default
// (UMR_NT128W=2 selects it; tests/test_gemm_gpu.py keeps it correct); an epilogue-overlap form has to keep 256x256 tiles.
//
// Epilogue = the fast class of gemm_nt256p.hip (bias as the accumulators' start value, ReLU, residual add / ReLU mask applied
// in the copy-out layout, fused 1024 -> {1,2} row reduction, no_store): the whole 128x256 bf16 tile is staged at once in the
// (drained) ring, two barriers per tile.
#include "umr_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int WBM = 128, WBN = 256, WBK = 32;
constexpr int WROWB = 64;                       // bytes per LDS row (32 bf16)
constexpr int WA_BYTES = WBM * WROWB;           // 8 KiB
constexpr int WB_BYTES = WBN * WROWB;           // 16 KiB
constexpr int WSTAGE = WA_BYTES + WB_BYTES;     // 24 KiB
constexpr int WNST = 3;
constexpr int WLDS = WNST * WSTAGE;             // 72 KiB (the epilogue's 64-KiB staging aliases it)

typedef bf16_t T2;

__device__ __forceinline__ unsigned pos_mask_bf16x2_w(unsigned a) {
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const s16x2 one = {1, 1}, zero = {0, 0};
    s16x2 v = __builtin_bit_cast(s16x2, a);
    v = __builtin_elementwise_min(v, one);
    v = __builtin_elementwise_max(v, zero);
    v = zero - v;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned add_bf16x2_w(unsigned x, unsigned y) {
    const float lo = __builtin_bit_cast(float, x << 16) + __builtin_bit_cast(float, y << 16);
    const float hi = __builtin_bit_cast(float, x & 0xFFFF0000u) + __builtin_bit_cast(float, y & 0xFFFF0000u);
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    bf16x2_ r;
    r[0] = (bf16_t)lo; r[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, r);
}

// AUXM: 0 none, 1 residual add, 2 ReLU mask.  RED: fused row reduction (umr_gemm_desc.red_*).
template <int AUXM, bool RED>
__global__ __launch_bounds__(256, 2) void gemm_nt128w_kernel(const umr_gemm_desc p, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SZ = 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile id (bijective for any grid size): consecutive tiles -- the N tiles of one M tile, then the next M tile --
    // run on one XCD and share its L2
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * WBM, n0 = tn * WBN;
    auto clamp31 = [](int64_t v) -> int { return v > 0x7FFFFFFFll ? 0x7FFFFFFF : (v < 0 ? 0 : (int)v); };
    const int rows_a = (p.M - m0 < WBM) ? (p.M - m0) : WBM;
    const int rows_b = (p.N - n0 < WBN) ? (p.N - n0) : WBN;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + (int64_t)m0 * p.lda * SZ), 0,
                                                                         clamp31((int64_t)rows_a * p.lda * SZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + (int64_t)n0 * p.ldb * SZ), 0,
                                                                         clamp31((int64_t)rows_b * p.ldb * SZ), 0x00020000);

    // ---- staging: DMA instruction i of wave w covers 16 rows x 64 B (1 KiB, lane-linear in LDS); lane -> (row = lane / 4, slot = lane % 4),
    // the slot holds chunk slot ^ T[(row >> 2) & 3]  ((row >> 2) & 3 == (lane >> 4) & 3: the instruction's first row is a multiple of 16)
    const int srow = lane >> 2;
    const int swz_l = (0x6C >> (2 * ((lane >> 4) & 3))) & 3;      // T = {0,3,2,1} packed two bits each: 0b01'10'11'00
    const unsigned gchunk = (unsigned)(((lane & 3) ^ swz_l) * 16);
    unsigned voA[2], voB[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) voA[i] = (unsigned)(((int64_t)((w * 2 + i) * 16 + srow) * p.lda) * SZ) + gchunk;
#pragma unroll
    for (int i = 0; i < 4; ++i) voB[i] = (unsigned)(((int64_t)((w * 4 + i) * 16 + srow) * p.ldb) * SZ) + gchunk;
    int st_issue = 0;   // next stage to issue
    auto dma = [&](auto itag) {   // one of the six DMA instructions of stage st_issue: 0,1 = A, 2..5 = B
        constexpr int I = decltype(itag)::value;
        char* dst = smem + (st_issue % WNST) * WSTAGE;
        const unsigned so = (unsigned)(st_issue * WBK * SZ);
        if (I < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, UMR_LDS_PTR(dst + (w * 2 + I) * 1024), 16, voA[I < 2 ? I : 0], so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, UMR_LDS_PTR(dst + WA_BYTES + (w * 4 + (I - 2)) * 1024), 16, voB[I >= 2 ? I - 2 : 0], so, 0, 0);
        if (I == 5) ++st_issue;
    };
#define WDMA(I) dma(std::integral_constant<int, I>{})

    // ---- fragment reads: A block i: rows i*16 + frow, chunk fq; B block j: rows w*64 + j*16 + frow
    const int frow = lane & 15, fq = lane >> 4;
    const int swz_f = (0x6C >> (2 * ((frow >> 2) & 3))) & 3;
    const int a_off = frow * WROWB + ((fq ^ swz_f) << 4);
    const int b_off = WA_BYTES + (w * 64 + frow) * WROWB + ((fq ^ swz_f) << 4);

    f32x4 acc[8][4];
    {
        // bias = the accumulators' start value (column n = n0 + w*64 + ntl*16 + fq*4 + e)
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) {
            const int n = n0 + w * 64 + ntl * 16 + fq * 4;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if ((p.flags & UMR_EPI_BIAS) && n < p.N) b = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt][ntl] = b;
        }
    }
    bf16x8 fa[2][8], fb[2][4];
    const int nk = p.K / WBK;   // even (K % 64 == 0, checked by the launcher)

    // prologue: three stages in flight, fragments of stage 0
    WDMA(0); WDMA(1); WDMA(2); WDMA(3); WDMA(4); WDMA(5);
    WDMA(0); WDMA(1); WDMA(2); WDMA(3); WDMA(4); WDMA(5);
    WDMA(0); WDMA(1); WDMA(2); WDMA(3); WDMA(4); WDMA(5);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[0][i] = *(const bf16x8*)(smem + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = *(const bf16x8*)(smem + b_off + j * 1024);

#define WMFMA(ACC, BF, AF) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF, AF, ACC, 0, 0, 0)
    auto body = [&](auto ctag, int t) {
        constexpr int C = decltype(ctag)::value;          // register set holding stage t
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {   // fragments of stage t+1 (past the end: dead data, never multiplied)
            const char* sb = smem + ((t + 1) % WNST) * WSTAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[C ^ 1][j] = *(const bf16x8*)(sb + b_off + j * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[C ^ 1][i] = *(const bf16x8*)(sb + a_off + i * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int ntl = 0; ntl < 4; ++ntl) WMFMA(acc[mt][ntl], fb[C][ntl], fa[C][mt]);
            // the six DMA instructions of stage t+3 sit between the MFMA groups (their ~60-100-cycle issue hides behind queued matrix work)
            if (mt < 6) {
                __builtin_amdgcn_sched_barrier(0);
                if (mt == 0) WDMA(0); else if (mt == 1) WDMA(1); else if (mt == 2) WDMA(2); else if (mt == 3) WDMA(3); else if (mt == 4) WDMA(4); else WDMA(5);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
#pragma unroll 1
    for (int t = 0; t < nk; t += 2) {
        body(std::integral_constant<int, 0>{}, t);
        body(std::integral_constant<int, 1>{}, t + 1);
    }
    // drain: trailing DMAs landed, everyone done with the ring
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue (fast class)
    const float relu_floor = (p.act == UMR_ACT_RELU) ? 0.f : -INFINITY;
    const int m_end = (p.M - m0 < WBM) ? p.M : m0 + WBM;
    f32x4 rw[2][4];
    if (RED) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ntl = 0; ntl < 4; ++ntl) {
                const int n = n0 + w * 64 + ntl * 16 + fq * 4;
                rw[c][ntl] = (c < p.red_c && n < p.N) ? *(const f32x4*)(p.red_w + (int64_t)c * p.N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    char* stb = smem;   // [128 rows][256 bf16] = 512 B per row, 16-byte chunk c16 of row r at (c16 ^ (r & 15))
    const bool store = !p.no_store;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const int lr = mt * 16 + frow;
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) {
            f32x4 v = acc[mt][ntl];
            if (AUXM == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], relu_floor);
            }
            bf16x4 t;
            t[0] = (bf16_t)v[0]; t[1] = (bf16_t)v[1]; t[2] = (bf16_t)v[2]; t[3] = (bf16_t)v[3];
            if (store) {
                const int c16 = w * 8 + ntl * 2 + (fq >> 1);
                *(bf16x4*)(stb + lr * 512 + ((c16 ^ (lr & 15)) << 4) + (fq & 1) * 8) = t;
            }
            if (RED) {   // dot products with the values AS STORED (bf16-rounded), as in the 256x256 kernel
#pragma unroll
                for (int e = 0; e < 4; ++e) { rs0 += (float)t[e] * rw[0][ntl][e]; rs1 += (float)t[e] * rw[1][ntl][e]; }
            }
        }
        if (RED) {
            rs0 += __shfl_xor(rs0, 16, 64); rs0 += __shfl_xor(rs0, 32, 64);
            rs1 += __shfl_xor(rs1, 16, 64); rs1 += __shfl_xor(rs1, 32, 64);
            const int m = m0 + lr;
            if (fq == 0 && m < m_end && n0 + w * 64 < p.N) {
                float* ro = p.red_out + ((int64_t)(tn * 4 + w) * p.M + m) * p.red_c;
                ro[0] = rs0;
                if (p.red_c == 2) ro[1] = rs1;
            }
        }
    }
    if (store) {
        __syncthreads();
        // copy-out: iteration j handles rows j*8 + tid/32, 16-byte chunk tid % 32; aux loads run four iterations ahead
        const int r8 = tid >> 5, c16 = tid & 31;
        const int n = n0 + c16 * 8;
        u32x4 ax[4];
        auto load_aux = [&](int j) -> u32x4 {
            const int m = m0 + j * 8 + r8;
            u32x4 a = {0u, 0u, 0u, 0u};
            if (AUXM != 0 && m < m_end && n < p.N) a = *(const u32x4*)((const T2*)p.aux + (int64_t)m * p.ldaux + n);
            return a;
        };
        if (AUXM != 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ax[j] = load_aux(j);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int lr = j * 8 + r8;
            const int m = m0 + lr;
            u32x4 o = *(const u32x4*)(stb + lr * 512 + ((c16 ^ (lr & 15)) << 4));
            if (AUXM != 0) {
                const u32x4 a = ax[j & 3];
                if (j + 4 < 16) ax[j & 3] = load_aux(j + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (AUXM == 2) ? (o[e] & pos_mask_bf16x2_w(a[e])) : add_bf16x2_w(o[e], a[e]);
            }
            if (m < m_end && n < p.N) *(u32x4*)((T2*)p.C + (int64_t)m * p.ldc + n) = o;
        }
    }
#undef WMFMA
#undef WDMA
}

}  // namespace

bool umr_nt256p_fast_epilogue(const umr_gemm_desc* d);   // gemm_nt256p.hip
bool umr_nt256p_plain_epilogue(const umr_gemm_desc* d);

// would umr_gemm_nt hand d to this kernel?  (bf16 plain GEMM, fast epilogue class, short K, enough tiles to fill the chip twice)
bool umr_nt128w_eligible(const umr_gemm_desc* d) {
    // UMR_NT128W: 0 = never (DEFAULT: measured 11-23 % SLOWER than the persistent 256x256 kernel on every short-K GEMM of the step,
    // see the header), 1 = by the rule below, 2 = whenever the kernel can run the problem; read per launch (tests A/B it)
    const char* e = getenv("UMR_NT128W");
    const int mode = e ? atoi(e) : 0;
    if (mode == 0) return false;
    if (d->dtype != UMR_BF16 || d->conv != 0 || d->a_rows_in > 0 || (d->K % 64) != 0 || (d->lda % 8) != 0 || (d->ldb % 8) != 0) return false;
    if (!umr_nt256p_fast_epilogue(d)) return false;
    if ((d->red_w || d->no_store) && !umr_nt256p_plain_epilogue(d)) return false;
    if (mode == 2) return true;
    const int64_t tiles = (int64_t)((d->M + WBM - 1) / WBM) * ((d->N + WBN - 1) / WBN);
    return d->K <= 1024 && tiles >= 1024;
}

int umr_launch_gemm_nt128w(const umr_gemm_desc* d, hipStream_t s) {
    const int tiles_m = (d->M + WBM - 1) / WBM, tiles_n = (d->N + WBN - 1) / WBN;
    const int64_t total = (int64_t)tiles_m * tiles_n;
    if (total >= (1ll << 31)) return umr_set_error(UMR_ERR_INVALID, "gemm_nt: grid too large");
    dim3 g((unsigned)total), b(256);
    const int auxm = (d->flags & UMR_EPI_ADD_AUX) ? 1 : (d->flags & UMR_EPI_MASK_RELU) ? 2 : 0;
#define LW(AX, RD)                                                                                                             \
    do {                                                                                                                       \
        static bool set_ = false;                                                                                              \
        if (!set_) {                                                                                                           \
            (void)hipFuncSetAttribute((const void*)gemm_nt128w_kernel<AX, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS); \
            set_ = true;                                                                                                       \
        }                                                                                                                      \
        hipLaunchKernelGGL((gemm_nt128w_kernel<AX, RD>), g, b, WLDS, s, *d, tiles_n);                                           \
    } while (0)
    if (d->red_w) LW(0, true);
    else if (auxm == 0) LW(0, false);
    else if (auxm == 1) LW(1, false);
    else LW(2, false);
#undef LW
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
