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
// partner workgroup hides (epilogue, barrier skew) is less than what the smaller tile costs.  The 128x512 form (NW = 8, one workgroup
// per CU, A read once for N = 512) was built on the hypothesis that the masked 1024 -> 512 data gradient is HBM-bound by its doubled
// A read (56 GB per launch for 39 GB algorithmic): it is 17 % slower there too (12.6 vs 10.4 ms) and 14-45 % slower elsewhere -- this
// simple one-barrier-per-stage BK = 32 pipeline reaches 0.79-0.89 PFLOP/s where the four-phase BK = 64 schedule of the persistent
// kernel reaches 0.95-1.08 on the same problems.  Both forms therefore are OFF by default (UMR_NT128W=2 / 3 select them;
// tests/test_gemm_gpu.py keeps them bit-identical to the persistent kernel); an epilogue-overlap form has to keep 256x256 tiles AND
// the persistent kernel's staging schedule.
//
// Epilogue = the fast class of gemm_nt256p.hip (bias as the accumulators' start value, ReLU, residual add / ReLU mask applied
// in the copy-out layout, fused 1024 -> {1,2} row reduction, no_store): the whole 128x256 bf16 tile is staged at once in the
// (drained) ring, two barriers per tile.
#include "umr_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// NW = waves per workgroup = 64-column blocks of the tile: 4 (128x256, two workgroups per CU) or 8 (128x512, one workgroup per CU:
// the N = 512 GEMMs then read their A operand ONCE -- with 256-wide tiles the two N-tiles of an M-tile run on neighbouring CUs but
// drift apart by more than the ~3 us a line lives in the 4-MiB L2, so A comes from HBM twice: the masked 1024 -> 512 data gradient
// of the centre head moves 56 GB per launch for 39 GB algorithmic and is HBM-bound at 5.5 TB/s).
constexpr int WBM = 128, WBK = 32;
constexpr int WROWB = 64;                       // bytes per LDS row (32 bf16)
constexpr int WA_BYTES = WBM * WROWB;           // 8 KiB
constexpr int WNST = 3;
template <int NW> struct WG {
    static constexpr int BN = 64 * NW;
    static constexpr int B_BYTES = BN * WROWB;              // 16 / 32 KiB
    static constexpr int STAGE = WA_BYTES + B_BYTES;        // 24 / 40 KiB
    static constexpr int LDS = WNST * STAGE;                // 72 / 120 KiB (the epilogue's 64-KiB staging aliases it)
    static constexpr int NDMA_A = 8 / NW;                   // DMA instructions per wave and stage: A 2 / 1, B 4
    static constexpr int NDMA = NDMA_A + 4;
    static constexpr int PASSES = NW / 4;                   // epilogue staging passes (64 KiB each)
};

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
template <int AUXM, bool RED, int NW>
__global__ __launch_bounds__(512, 2) void gemm_nt128w_kernel(const umr_gemm_desc p, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SZ = 2;
    constexpr int WBN = WG<NW>::BN, WSTAGE = WG<NW>::STAGE, NDMA_A = WG<NW>::NDMA_A, NDMA = WG<NW>::NDMA;
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
    unsigned voA[2], voB[4];   // (fixed bound: an array of template-dependent size captured by the generic lambda below makes hipcc drop the kernel on the host side without a diagnostic)
#pragma unroll
    for (int i = 0; i < NDMA_A; ++i) voA[i] = (unsigned)(((int64_t)((w * NDMA_A + i) * 16 + srow) * p.lda) * SZ) + gchunk;
#pragma unroll
    for (int i = 0; i < 4; ++i) voB[i] = (unsigned)(((int64_t)((w * 4 + i) * 16 + srow) * p.ldb) * SZ) + gchunk;
    int st_issue = 0;   // next stage to issue
    auto dma = [&](auto itag) {   // one of the NDMA instructions of stage st_issue: the first NDMA_A are A rows, then four of B
        constexpr int I = decltype(itag)::value;
        if constexpr (I < NDMA) {
            char* dst = smem + (st_issue % WNST) * WSTAGE;
            const unsigned so = (unsigned)(st_issue * WBK * SZ);
            if constexpr (I < NDMA_A)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, UMR_LDS_PTR(dst + (w * NDMA_A + I) * 1024), 16, voA[I], so, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, UMR_LDS_PTR(dst + WA_BYTES + (w * 4 + (I - NDMA_A)) * 1024), 16, voB[I - NDMA_A], so, 0, 0);
            if (I == NDMA - 1) ++st_issue;
        }
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
    WDMA(0); WDMA(1); WDMA(2); WDMA(3); WDMA(4); WDMA(5);      // (WDMA(5) is empty when a stage has five instructions)
    WDMA(0); WDMA(1); WDMA(2); WDMA(3); WDMA(4); WDMA(5);
    WDMA(0); WDMA(1); WDMA(2); WDMA(3); WDMA(4); WDMA(5);
    if (NDMA == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[0][i] = *(const bf16x8*)(smem + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[0][j] = *(const bf16x8*)(smem + b_off + j * 1024);

#define WMFMA(ACC, BF, AF) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF, AF, ACC, 0, 0, 0)
    auto body = [&](auto ctag, int t) {
        constexpr int C = decltype(ctag)::value;          // register set holding stage t
        if (NDMA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
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
    constexpr int PASSES = WG<NW>::PASSES, MT_PER_PASS = 8 / PASSES, ROWS_PER_PASS = WBM / PASSES;
    constexpr int ROWBYTES = WBN * 2;      // staged row: WBN bf16; 16-byte chunk c16 of row r at (c16 ^ (r & 15))
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
    char* stb = smem;
    const bool store = !p.no_store;
    // copy-out roles: iteration j of a pass handles rows j*8 + tid / (8 NW), 16-byte chunk tid % (8 NW)
    const int r8 = tid / (8 * NW), c16o = tid % (8 * NW);
    const int n_out = n0 + c16o * 8;
    auto pass = [&](auto ptag) {
        constexpr int PS = decltype(ptag)::value;
        if (PS > 0) __syncthreads();    // the previous pass has been copied out
#pragma unroll
        for (int ml = 0; ml < MT_PER_PASS; ++ml) {
            constexpr int dummy = 0; (void)dummy;
            const int mt = PS * MT_PER_PASS + ml;
            const int lr = ml * 16 + frow;           // row inside the pass
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
                    *(bf16x4*)(stb + lr * ROWBYTES + ((c16 ^ (lr & 15)) << 4) + (fq & 1) * 8) = t;
                }
                if (RED) {   // dot products with the values AS STORED (bf16-rounded), as in the 256x256 kernel
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rs0 += (float)t[e] * rw[0][ntl][e]; rs1 += (float)t[e] * rw[1][ntl][e]; }
                }
            }
            if (RED) {
                rs0 += __shfl_xor(rs0, 16, 64); rs0 += __shfl_xor(rs0, 32, 64);
                rs1 += __shfl_xor(rs1, 16, 64); rs1 += __shfl_xor(rs1, 32, 64);
                const int m = m0 + mt * 16 + frow;
                if (fq == 0 && m < m_end && n0 + w * 64 < p.N) {
                    float* ro = p.red_out + ((int64_t)(tn * NW + w) * p.M + m) * p.red_c;
                    ro[0] = rs0;
                    if (p.red_c == 2) ro[1] = rs1;
                }
            }
        }
        if (store) {
            __syncthreads();
            constexpr int NIT = ROWS_PER_PASS / 8;
            u32x4 ax[4];
            auto load_aux = [&](int j) -> u32x4 {
                const int m = m0 + PS * ROWS_PER_PASS + j * 8 + r8;
                u32x4 a = {0u, 0u, 0u, 0u};
                if (AUXM != 0 && m < m_end && n_out < p.N) a = *(const u32x4*)((const T2*)p.aux + (int64_t)m * p.ldaux + n_out);
                return a;
            };
            if (AUXM != 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ax[j] = load_aux(j);
            }
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int lr = j * 8 + r8;
                const int m = m0 + PS * ROWS_PER_PASS + lr;
                u32x4 o = *(const u32x4*)(stb + lr * ROWBYTES + ((c16o ^ (lr & 15)) << 4));
                if (AUXM != 0) {
                    const u32x4 a = ax[j & 3];
                    if (j + 4 < NIT) ax[j & 3] = load_aux(j + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (AUXM == 2) ? (o[e] & pos_mask_bf16x2_w(a[e])) : add_bf16x2_w(o[e], a[e]);
                }
                if (m < m_end && n_out < p.N) *(u32x4*)((T2*)p.C + (int64_t)m * p.ldc + n_out) = o;
            }
        }
    };
    pass(std::integral_constant<int, 0>{});
    if constexpr (PASSES == 2) pass(std::integral_constant<int, 1>{});
#undef WMFMA
#undef WDMA
}

}  // namespace

bool umr_nt256p_fast_epilogue(const umr_gemm_desc* d);   // gemm_nt256p.hip
bool umr_nt256p_plain_epilogue(const umr_gemm_desc* d);

// Which form (0 = none, 4 = 128x256 / two workgroups per CU, 8 = 128x512 / one workgroup per CU) would umr_gemm_nt use for d?
// UMR_NT128W (read per launch; tests A/B it): 0 = never (DEFAULT -- both forms measured slower than the persistent 256x256 kernel
// on every GEMM of the step, see the header and profiles/r03_nt128w_ab.txt); 2 = the 128x256 form whenever it can run the
// problem; 3 = the 128x512 form whenever it can.
static int nt128w_form(const umr_gemm_desc* d) {
    const char* e = getenv("UMR_NT128W");
    const int mode = e ? atoi(e) : 0;
    if (mode != 2 && mode != 3) return 0;
    if (d->dtype != UMR_BF16 || d->conv != 0 || d->a_rows_in > 0 || (d->K % 64) != 0 || (d->lda % 8) != 0 || (d->ldb % 8) != 0) return 0;
    if (!umr_nt256p_fast_epilogue(d)) return 0;
    if ((d->red_w || d->no_store) && !umr_nt256p_plain_epilogue(d)) return 0;
    return mode == 2 ? 4 : 8;
}
bool umr_nt128w_eligible(const umr_gemm_desc* d) { return nt128w_form(d) != 0; }

int umr_launch_gemm_nt128w(const umr_gemm_desc* d, hipStream_t s) {
    const int nw = nt128w_form(d);
    const int bn = 64 * nw;
    const int tiles_m = (d->M + WBM - 1) / WBM, tiles_n = (d->N + bn - 1) / bn;
    const int64_t total = (int64_t)tiles_m * tiles_n;
    if (nw == 0 || total >= (1ll << 31)) return umr_set_error(UMR_ERR_INVALID, "gemm_nt: grid too large");
    dim3 g((unsigned)total), b(64 * nw);
    const int auxm = (d->flags & UMR_EPI_ADD_AUX) ? 1 : (d->flags & UMR_EPI_MASK_RELU) ? 2 : 0;
#define LW(AX, RD, NWV)                                                                                                             \
    do {                                                                                                                            \
        UMR_SET_MAX_LDS_ONCE((gemm_nt128w_kernel<AX, RD, NWV>), WG<NWV>::LDS);                                                                                                                           \
        hipLaunchKernelGGL((gemm_nt128w_kernel<AX, RD, NWV>), g, b, WG<NWV>::LDS, s, *d, tiles_n);                                   \
    } while (0)
#define LWN(AX, RD) do { if (nw == 4) LW(AX, RD, 4); else LW(AX, RD, 8); } while (0)
    if (d->red_w) LWN(0, true);
    else if (auxm == 0) LWN(0, false);
    else if (auxm == 1) LWN(1, false);
    else LWN(2, false);
#undef LWN
#undef LW
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
