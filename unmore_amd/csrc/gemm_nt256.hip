// 256x256-tile NT GEMM / implicit 3x3 conv for bf16 (the throughput path of umr_gemm_nt).
//
// Why a second kernel: at 128x128 the LDS staging traffic is 64 flop/byte -- ~17 TB/s of L2->LDS at 1.1 PFLOP/s,
// half of the chip's aggregate L2 bandwidth.  A 256x256 tile halves that (128 flop/byte) but, with one
// 512-thread workgroup per CU, only pays with a software pipeline that keeps LDS-DMA in flight across barriers.
//
// Geometry: 8 waves as 2 (M) x 4 (N); a wave owns 128x64 = 8x4 tiles of v_mfma_f32_16x16x32_bf16
// (128 accumulator VGPRs).  BK = 64; LDS = 2 buffers x (A 256x128 B + B 256x128 B) = 128 KiB, XOR-swizzled as in
// gemm_nt.hip.  Each K-tile is 4 PHASES; a phase is one 64x32 output quadrant x K=64 (16 MFMAs):
//     Q0 = (m-half 0, n-half 0)   reads A(mh0) [8 ds_read_b128] + B(nh0) [4]
//     Q1 = (m-half 0, n-half 1)   reads B(nh1) [4]
//     Q2 = (m-half 1, n-half 1)   reads A(mh1) [8]
//     Q3 = (m-half 1, n-half 0)   reads nothing (B(nh0) kept in registers)
// The tile is STAGED in four 16-KiB groups that are exactly those read sets (A rows {0-63,128-191}, B rows
// {wc*64+0..31}, B rows {wc*64+32..63}, A rows {64-127,192-255}), one group per phase, two tiles ahead:
//     group      last read   re-staged for tile t+2 in   first read again in   lead
//     A(mh0)     (t,  Q0)    (t,  Q2)                    (t+2, Q0)             6 phases
//     B(nh0)     (t,  Q0)    (t,  Q3)                    (t+2, Q0)             5
//     B(nh1)     (t,  Q1)    (t+1,Q0)                    (t+2, Q1)             5
//     A(mh1)     (t,  Q2)    (t+1,Q1)                    (t+2, Q2)             5
// so a region is rewritten >= 2 phases (two barriers) after its last ds_read (WAR) and every group has >= 5
// phases (~5 x 256..512 MFMA cycles) to land.  Each phase: issue 2 LDS-DMA, ds_reads, `s_waitcnt vmcnt(8)`
// (everything but the 4 youngest groups has landed: exactly what the NEXT phase reads), one raw s_barrier,
// lgkmcnt(0), 16 MFMAs.  The DMA count per phase is constant (tiles past the end are issued out-of-range and
// write zeros into dead regions), so the counted wait is uniform.  MFMAs run asynchronously behind the next
// phase's DMA issue and ds_reads.
#include "umr_common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int ROWB2 = 128;
constexpr int TILE2 = 256 * ROWB2;   // 32 KiB per operand
constexpr int BUF2 = 2 * TILE2;      // 64 KiB per K-tile
constexpr int LDS2 = 2 * BUF2;       // 128 KiB

typedef bf16_t T2;

template <int CONV>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(const umr_gemm_desc p, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SZ = 2;
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;

    // ---- staging roles.  Group-local row lr = (w*2+i)*8 + lane/8 (0..127), chunk position lane%8.
    // tile row of (group, lr): A(mh): (lr>>6)*128 + mh*64 + (lr&63);  B(nh): (lr>>5)*64 + nh*32 + (lr&31).
    const int lrow = lane >> 3, lchk = lane & 7;
    const int stride = (CONV == 2) ? 2 : 1;
    // [group 0..3 = A0,B0,B1,A1][i]
    unsigned vo[4][2];   // per-lane voffset (or OOB) -- constants of the K loop
    unsigned a_eff[2][2];  // A groups after the per-tap halo mask (index 0: A0, 1: A1)
    unsigned a_tapmask[2][2];  // conv: bit t set <=> tap t of this row lies inside the image
    int lds_row[4][2];   // tile row written by (group, i): LDS dest = row*128 (wave-uniform part) ...
    const char* a_base;
    {
        int64_t origin;
        int ar0 = m0;
        int64_t pix0 = 0;
        const int hw = (CONV != 0) ? p.Ho * p.Wo : 1;
        if (CONV == 0) {
            if (p.a_rows_in > 0) ar0 = (m0 / p.a_rows_in) * p.a_rows_out + p.a_row_off + (m0 % p.a_rows_in);
            origin = (int64_t)ar0 * p.lda;
        } else {
            const int b0 = m0 / hw, rem0 = m0 - b0 * hw;
            const int oy0 = rem0 / p.Wo, ox0 = rem0 - oy0 * p.Wo;
            pix0 = ((int64_t)b0 * p.H + oy0 * stride) * p.W + ox0 * stride;
            origin = (pix0 - (p.W + 1)) * p.Cin;
        }
        a_base = (const char*)p.A + origin * SZ;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {      // A groups: gi = m-half
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int lr = (w * 2 + i) * 8 + lrow;
                const int trow = (lr >> 6) * 128 + gi * 64 + (lr & 63);
                const int g = gi == 0 ? 0 : 3;
                lds_row[g][i] = (((w * 2 + i) * 8) >> 6) * 128 + gi * 64 + (((w * 2 + i) * 8) & 63);
                const int gch = lchk ^ ((trow >> 1) & 7);
                const int m = m0 + trow;
                unsigned v = OOB;
                a_tapmask[gi][i] = 0x1FFu;
                if (CONV == 0) {
                    int ar = m;
                    if (p.a_rows_in > 0) ar = (m / p.a_rows_in) * p.a_rows_out + p.a_row_off + (m % p.a_rows_in);
                    if (m < p.M) v = (unsigned)(((int64_t)(ar - ar0) * p.lda) * SZ + gch * 16);
                } else {
                    const int b = m / hw, rem = m - b * hw;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    const int64_t pix = ((int64_t)b * p.H + oy * stride) * p.W + ox * stride;
                    if (m < p.M) v = (unsigned)((pix - pix0) * p.Cin * SZ + gch * 16);
                    unsigned mask = 0;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int iy = oy * stride - 1 + tap / 3, ix = ox * stride - 1 + tap % 3;
                        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mask |= 1u << tap;
                    }
                    a_tapmask[gi][i] = mask;
                }
                vo[g][i] = v;
                a_eff[gi][i] = v;
            }
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {      // B groups: gi = n-half
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int lr = (w * 2 + i) * 8 + lrow;
                const int trow = (lr >> 5) * 64 + gi * 32 + (lr & 31);
                const int g = 1 + gi;
                lds_row[g][i] = (((w * 2 + i) * 8) >> 5) * 64 + gi * 32 + (((w * 2 + i) * 8) & 31);
                const int gch = lchk ^ ((trow >> 1) & 7);
                vo[g][i] = (n0 + trow < p.N) ? (unsigned)(((int64_t)trow * p.ldb) * SZ + gch * 16) : OOB;
            }
        }
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + (int64_t)n0 * p.ldb * SZ), 0, 0x7FFFFFFF, 0x00020000);

    const int ktiles_per_tap = (CONV == 0) ? 0 : (p.Cin + BK2 - 1) / BK2;
    const int nt = (CONV == 0) ? (p.K + BK2 - 1) / BK2 : 9 * ktiles_per_tap;
    // (K % 64 == 0, resp. Cin % 64 == 0, is guaranteed by the dispatcher: no K-tail masking here)

    // staging cursor: groups are issued in the order A0,B0,B1,A1 of tile 0, then of tile 1, ...
    int st_tile = 0, st_tap = 0, st_ci = 0;
    unsigned soffA = 0, soffB = 0;
    // called when group 0 (A0) of a new tile is about to be issued: K offsets, and the halo mask on a tap change
    auto stage_prep = [&]() {
        if (CONV == 0) {
            soffA = soffB = (unsigned)(st_tile * BK2 * SZ);
        } else {
            const int c0 = st_ci * BK2;
            const int ky = st_tap / 3, kx = st_tap - ky * 3;
            {   // the tap changes every K-tile (channel-chunk-major order)
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        a_eff[gi][i] = ((a_tapmask[gi][i] >> st_tap) & 1u) ? vo[gi == 0 ? 0 : 3][i] : OOB;
            }
            soffA = (unsigned)(((ky * p.W + kx) * p.Cin + c0) * SZ);
            soffB = (unsigned)((st_tap * p.Cin + c0) * SZ);
            if (++st_tap == 9) { st_tap = 0; ++st_ci; }
        }
    };
    // one LDS-DMA instruction (i = 0/1) of group G of the cursor's tile; G = 3 completes the tile
    auto stage_issue = [&](auto gtag, auto itag) {
        constexpr int G = decltype(gtag)::value, I = decltype(itag)::value;
        unsigned v = (G == 0) ? a_eff[0][I] : (G == 3) ? a_eff[1][I] : vo[G][I];
        if (st_tile >= nt) v = OOB;  // past the end: zero fill into a dead region, keeps the DMA count uniform
        char* dst = smem + (st_tile & 1) * BUF2;
        if (G == 0 || G == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, UMR_LDS_PTR(dst + lds_row[G][I] * ROWB2), 16, v, soffA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, UMR_LDS_PTR(dst + TILE2 + lds_row[G][I] * ROWB2), 16, v, soffB, 0, 0);
        if (G == 3 && I == 1) ++st_tile;
    };
#define STAGE_DMA(G, I) stage_issue(std::integral_constant<int, G>{}, std::integral_constant<int, I>{})

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wr = w >> 2, wc = w & 3;
    const int frow = lane & 15, fq = lane >> 4;
    // fragment addresses: the swizzle key ((row>>1)&7) of row = base + 16*i + frow does not depend on i, so tile i
    // of an operand is a compile-time +i*2048 bytes from one per-(operand, k-step) base register
    int a_ad[2], b_ad[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int q = ks * 4 + fq;
        const int sw = (frow >> 1) & 7;
        a_ad[ks] = (wr * 128 + frow) * ROWB2 + ((q ^ sw) << 4);
        b_ad[ks] = TILE2 + (wc * 64 + frow) * ROWB2 + ((q ^ sw) << 4);
    }
#define A_FRAG(ks, i) (*(const bf16x8*)(sbuf + a_ad[ks] + (i) * 16 * ROWB2))
#define B_FRAG(ks, i) (*(const bf16x8*)(sbuf + b_ad[ks] + (i) * 16 * ROWB2))

    bf16x8 fa[2][4], fb0[2][2], fb1[2][2];  // [ks][tile]: A of the current m-half, B(nh0), B(nh1)

    // One phase = ds_reads, counted wait + barrier, then a 16-MFMA quadrant with this phase's two LDS-DMA
    // instructions placed INSIDE the MFMA cluster (after 4 and after 10 MFMAs): the ~100-cycle DMA issue then
    // hides behind matrix work that is already queued instead of stalling both waves of a SIMD at once.
    // At the wait, the groups issued in earlier phases number "all but this phase's", so vmcnt(6) leaves the
    // three youngest groups in flight -- the same lead as issuing before the wait with vmcnt(8).
#define PHASE_SYNC()                                               \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");               \
    __builtin_amdgcn_s_barrier();                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);
#define MFMA(ACC, BF, AF) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF, AF, ACC, 0, 0, 0)
    // quadrant (M0 = first m-tile of the acc block, N0 = first n-tile), B fragments FB, staging group G
#define QUADRANT(M0, N0, FB, G)                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                  \
    MFMA(acc[M0 + 0][N0 + 0], FB[0][0], fa[0][0]); MFMA(acc[M0 + 0][N0 + 1], FB[0][1], fa[0][0]);   \
    MFMA(acc[M0 + 1][N0 + 0], FB[0][0], fa[0][1]); MFMA(acc[M0 + 1][N0 + 1], FB[0][1], fa[0][1]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    STAGE_DMA(G, 0);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MFMA(acc[M0 + 2][N0 + 0], FB[0][0], fa[0][2]); MFMA(acc[M0 + 2][N0 + 1], FB[0][1], fa[0][2]);   \
    MFMA(acc[M0 + 3][N0 + 0], FB[0][0], fa[0][3]); MFMA(acc[M0 + 3][N0 + 1], FB[0][1], fa[0][3]);   \
    MFMA(acc[M0 + 0][N0 + 0], FB[1][0], fa[1][0]); MFMA(acc[M0 + 0][N0 + 1], FB[1][1], fa[1][0]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    STAGE_DMA(G, 1);                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MFMA(acc[M0 + 1][N0 + 0], FB[1][0], fa[1][1]); MFMA(acc[M0 + 1][N0 + 1], FB[1][1], fa[1][1]);   \
    MFMA(acc[M0 + 2][N0 + 0], FB[1][0], fa[1][2]); MFMA(acc[M0 + 2][N0 + 1], FB[1][1], fa[1][2]);   \
    MFMA(acc[M0 + 3][N0 + 0], FB[1][0], fa[1][3]); MFMA(acc[M0 + 3][N0 + 1], FB[1][1], fa[1][3]);   \
    __builtin_amdgcn_s_setprio(0);

    // With 6 groups pre-issued, phase p of tile t issues group (6 + 4t + p) % 4 of tile (6 + 4t + p) / 4:
    //   Q0 -> B1(t+1), Q1 -> A1(t+1), Q2 -> A0(t+2), Q3 -> B0(t+2)   (the table in the header)
    auto tile_body = [&](const char* sbuf) {
        // ---- phase 0: Q0 = (mh0, nh0)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fb0[ks][i] = B_FRAG(ks, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, i);
        }
        PHASE_SYNC();
        QUADRANT(0, 0, fb0, 2)
        // ---- phase 1: Q1 = (mh0, nh1)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) fb1[ks][i] = B_FRAG(ks, 2 + i);
        PHASE_SYNC();
        QUADRANT(0, 2, fb1, 3)
        // ---- phase 2: Q2 = (mh1, nh1)
        stage_prep();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, 4 + i);
        PHASE_SYNC();
        QUADRANT(4, 2, fb1, 0)
        // ---- phase 3: Q3 = (mh1, nh0)
        PHASE_SYNC();
        QUADRANT(4, 0, fb0, 1)
    };

    // prologue: the six groups the steady-state schedule has already issued when tile 0 starts
    // (A0,B0,B1,A1 of tile 0; A0,B0 of tile 1), then make A0(0), B0(0) visible.
    stage_prep();
    STAGE_DMA(0, 0); STAGE_DMA(0, 1); STAGE_DMA(1, 0); STAGE_DMA(1, 1);
    STAGE_DMA(2, 0); STAGE_DMA(2, 1); STAGE_DMA(3, 0); STAGE_DMA(3, 1);
    stage_prep();
    STAGE_DMA(0, 0); STAGE_DMA(0, 1); STAGE_DMA(1, 0); STAGE_DMA(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int t = 0; t < nt; ++t) tile_body(smem + (t & 1) * BUF2);
#undef QUADRANT
#undef MFMA
#undef PHASE_SYNC
#undef STAGE_DMA
#undef A_FRAG
#undef B_FRAG

    // ---- epilogue through LDS: 4 passes of 64 tile rows x 256 columns
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    constexpr int EP_LD = 260;
    float* stg = (float*)smem;
    const bool vec_ok = ((p.N & 7) == 0) && ((p.ldc & 7) == 0) && ((p.ldc2 & 7) == 0) && ((p.ldaux & 7) == 0) &&
                        ((p.ldaux2 & 7) == 0);
    // one copy of the store code in runtime loops: fully unrolled and inlined the generic epilogue is ~100 KiB of
    // instructions per kernel and runs out of the I-cache (the persistent kernel, gemm_nt256p.hip, has the fast path)
    auto stage_pass = [&](auto ptag) {
        constexpr int PASS = decltype(ptag)::value;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int lr = wr * 32 + mh * 16 + frow;
#pragma unroll
            for (int ntl = 0; ntl < 4; ++ntl)
                *(f32x4*)(stg + lr * EP_LD + wc * 64 + ntl * 16 + fq * 4) = acc[PASS * 2 + mh][ntl];
        }
    };
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        __syncthreads();
        switch (pass) {
            case 0: stage_pass(std::integral_constant<int, 0>{}); break;
            case 1: stage_pass(std::integral_constant<int, 1>{}); break;
            case 2: stage_pass(std::integral_constant<int, 2>{}); break;
            default: stage_pass(std::integral_constant<int, 3>{}); break;
        }
        __syncthreads();
#pragma unroll 1
        for (int task = tid; task < 64 * 32; task += 512) {
            const int lr = task >> 5, c8 = (task & 31) * 8;
            const int trow = (lr >> 5) * 128 + pass * 32 + (lr & 31);
            const int m = m0 + trow, n = n0 + c8;
            if (m >= p.M || n >= p.N) continue;
            const f32x4 v0 = *(const f32x4*)(stg + lr * EP_LD + c8), v1 = *(const f32x4*)(stg + lr * EP_LD + c8 + 4);
            if (vec_ok) {
                epilogue_store8<T2>(p, m, n, v0, v1);
            } else {
                epilogue_store<T2>(p, m, n, v0);
                if (n + 4 < p.N) epilogue_store<T2>(p, m, n + 4, v1);
            }
        }
    }
}

}  // namespace

int umr_launch_gemm_nt256p(const umr_gemm_desc* d, hipStream_t s);  // gemm_nt256p.hip (persistent form)
bool umr_nt256p_fast_epilogue(const umr_gemm_desc* d);
bool umr_nt256p_plain_epilogue(const umr_gemm_desc* d);

// would umr_launch_gemm_nt256 hand d to the persistent kernel's fast epilogue?
bool umr_nt256_rowreduce_path(const umr_gemm_desc* d) {
    if (umr_opt_or(UMR_OPT_NT256_PERSIST, 1) == 0) return false;
    return ((d->conv == 0 && d->a_rows_in <= 0) || d->conv == 1) && umr_nt256p_plain_epilogue(d);
}

// launched from umr_gemm_nt (gemm_nt.hip) for bf16 problems large enough to fill the chip with 256x256 tiles
int umr_launch_gemm_nt256(const umr_gemm_desc* d, hipStream_t s) {
    {   // the persistent kernel covers plain GEMMs without A-row remap and stride-1 convs; UMR_NT256_PERSIST=0 disables it
        const int persist = umr_opt_or(UMR_OPT_NT256_PERSIST, 1);
        if (persist && ((d->conv == 0 && d->a_rows_in <= 0) || d->conv == 1)) return umr_launch_gemm_nt256p(d, s);
    }
    if (d->red_w || d->no_store) return umr_set_error(UMR_ERR_UNSUPPORTED, "gemm_nt: fused row reduction / no_store is only implemented by the persistent 256x256 path");
    const int tiles_m = (d->M + BM2 - 1) / BM2, tiles_n = (d->N + BN2 - 1) / BN2;
    const int64_t grid = (int64_t)tiles_m * tiles_n;
    dim3 g((unsigned)grid), b(512);
#define L256(CV)                                                                                                       \
    do {                                                                                                               \
        UMR_SET_MAX_LDS_ONCE((gemm_nt256_kernel<CV>), LDS2); \
        hipLaunchKernelGGL((gemm_nt256_kernel<CV>), g, b, LDS2, s, *d, tiles_n);                                        \
    } while (0)
    if (d->conv == 0) L256(0);
    else if (d->conv == 1) L256(1);
    else L256(2);
#undef L256
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
