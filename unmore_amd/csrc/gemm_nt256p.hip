// Persistent form of the 256x256-tile bf16 NT GEMM / implicit 3x3 conv (see gemm_nt256.hip for the tile
// geometry, the four-phase K-tile and the staging-group schedule, which are unchanged here).
//
// Why: the one-tile-per-workgroup kernel measures  T(tile) = 12 us + 1.56 us x K-tiles  (tools/kbench4.py): with one
// 512-thread workgroup per CU nothing overlaps a tile's first-load latency, its epilogue and the launch of the next
// workgroup, so the 1x1 head GEMMs (4..16 K-tiles) spend about half their time outside the MFMA loop and run at
// the SUM of their HBM time and their MFMA time instead of the max.
//
// Here one workgroup per CU walks a strided sequence of output tiles and treats their K-tiles as ONE stream: the
// staging cursor (two K-tiles ahead of the MFMAs) rolls over from the last K-tile of output tile i into the
// first K-tiles of output tile i+1, so those loads are in flight during -- and have landed by the end of -- the
// epilogue of tile i.  The epilogue therefore cannot borrow the staging buffers: it goes through a separate
// 32-KiB LDS region (fast / GELU classes: 4 passes of 64 rows x 256 bf16 columns; generic class: 8 passes of 32 rows of
// f32; XOR-swizzled 16-B chunks), and its global stores drain behind the next tile's first phases.
//
// Addressing is arranged so that a tile switch costs scalar work only: per-lane voffsets are tile-independent
// (rows relative to the tile origin) and row/column validity comes from the buffer descriptor's num_records
// (out-of-range rows read as zeros); only the conv halo masks are recomputed per tile.
#include "umr_common.h"
#include "gemm_epilogue.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int ROWB2 = 128;
constexpr int TILE2 = 256 * ROWB2;   // 32 KiB per operand
constexpr int BUF2 = 2 * TILE2;      // 64 KiB per K-tile
constexpr int STG_OFF = 2 * BUF2;    // epilogue staging behind the two K-tile buffers
constexpr int STG_BYTES = 32 * 256 * 4;
constexpr int LDS2P = STG_OFF + STG_BYTES;  // 160 KiB

typedef bf16_t T2;

// two packed bf16 gradients times GELU'(two packed bf16 pre-activations)
__device__ __forceinline__ unsigned mul_dgelu_bf16x2(unsigned g, unsigned x) {
    const float lo = __builtin_bit_cast(float, g << 16) * dgelu_sel<bf16_t>(__builtin_bit_cast(float, x << 16));
    const float hi = __builtin_bit_cast(float, g & 0xFFFF0000u) * dgelu_sel<bf16_t>(__builtin_bit_cast(float, x & 0xFFFF0000u));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = (bf16_t)lo; r[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, r);
}

// two packed bf16 + two packed bf16, each sum in f32, rounded once
__device__ __forceinline__ unsigned add_bf16x2(unsigned x, unsigned y) {
    const float lo = __builtin_bit_cast(float, x << 16) + __builtin_bit_cast(float, y << 16);
    const float hi = __builtin_bit_cast(float, x & 0xFFFF0000u) + __builtin_bit_cast(float, y & 0xFFFF0000u);
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 r;
    r[0] = (bf16_t)lo; r[1] = (bf16_t)hi;
    return __builtin_bit_cast(unsigned, r);
}

// per 16-bit half: 0xFFFF where the bf16 is > 0 (its bits, read as a signed 16-bit integer, are positive), else 0
__device__ __forceinline__ unsigned pos_mask_bf16x2(unsigned a) {
    typedef __attribute__((ext_vector_type(2))) short s16x2;
    const s16x2 one = {1, 1}, zero = {0, 0};
    s16x2 v = __builtin_bit_cast(s16x2, a);
    v = __builtin_elementwise_min(v, one);
    v = __builtin_elementwise_max(v, zero);
    v = zero - v;
    return __builtin_bit_cast(unsigned, v);
}

// EPI: 1 = generic epilogue, 3 = fast class (AUXM: 0 none, 1 residual add, 2 ReLU mask; RED: fused row reduction), 4 = GELU class.
// The fast class is specialised at compile time: as runtime flag tests its 32 values per thread and pass cost ~1000
// issue cycles per wave (4 branches per 4 values), and an epilogue pass is pure issue time on 2 waves per SIMD.
//
// X3 (dtype UMR_BF16X3, EPI 5): fp32-grade products from operands that arrive PRE-SPLIT into three bf16 planes per value
// (x = h + m + l, exact for f32 data; umr_split3).  Operand rows hold [h(K) | m(K) | l(K)] (conv A: [h(Cin) | m(Cin) | l(Cin)]
// per pixel; conv B: [plane][tap][ci]).  The K loop walks, for every 64-wide K-tile, the six plane pairs
// (h,h) (h,m) (h,l) (m,h) (m,m) (l,h) -- the terms of (ah+am+al)(bh+bm+bl) above 2^-26 of the product -- as six ordinary bf16
// K-tiles accumulated in f32: the main loop is the bf16 kernel unchanged (no split arithmetic in it; the 128x128 fp32 kernel of
// gemm_nt.hip spends more time splitting fragments than multiplying), only the staging offsets differ.  Per f32 product: 6 MFMA
// products = 96 matrix-pipe cycles per 16x16x32 block against 256 for the f32 MFMA.
template <int CONV, int EPI, int AUXM, bool RED, bool X3 = false, bool P2 = true>
__global__ __launch_bounds__(512, 2) void gemm_nt256p_kernel(const umr_gemm_desc p, int tiles_n, int total_tiles, int stagger, int bm, int npairs = 6,
                                                             int kt_per_arg = 0) {
    // X3 only -- split-K along gridDim.y: the K range of every output tile is cut into gridDim.y contiguous runs of `kt_per_arg`
    // K-tiles; the workgroups of grid row sp walk the tiles as usual, but only run sp of each, and store their raw f32 accumulators
    // to slab sp (p.C = the slab array [gridDim.y][M][ldc], plain f32 store, no bias); x3_splitk_finish_kernel adds the slabs in
    // order and runs the real epilogue.  The transformer's GEMMs of the reference recipe (1300 tokens: 24-96 tiles of 96-384
    // plane-pair steps) and the DPT convolutions on 4x4 ... 64x64 maps (5-320 tiles of 216+ steps) leave most CUs idle otherwise.
    // The run is a constant of the workgroup: the persistent loop itself is unchanged.
    const int sp = (X3 && kt_per_arg > 0) ? (int)blockIdx.y : 0;
    static_assert(X3 == (EPI == 5 || EPI == 6), "the plane-pair K loop and the f32 / plane epilogues (EPI 5, 6) go together");
    // P2: two-phase K-tile (2 barriers instead of 4) with a static priority for waves 4-7; !P2: four phases with priority flips around
    // the MFMA clusters.  The conv always runs two-phase (+2.4 %).  Plain GEMMs (round 4): two-phase from 8 K-tiles per output tile on
    // -- 2-8 % faster on the head 1x1 layers, the ViT GEMMs and their plane (fp32-grade) forms, 11 % at K = 8192 -- four-phase below
    // (the 256 -> 512 layers, 4 K-tiles, are 3 % SLOWER two-phase): the host picks (profiles/r04_plain_gemm_two_phase_ab.txt).
    constexpr bool PH2 = P2;
    static_assert(CONV == 0 || P2, "the conv is instantiated in its two-phase form only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SZ = 2;
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Stagger (MI355X_MICROARCH.md, "Two waves per SIMD", item 9).  Waves w and w+4 share a SIMD.  When all eight waves run
    // the same [read fragments, wait, barrier, MFMA] program they reach their LDS read bursts (128 KiB per phase = 512 LDS
    // cycles) and the barrier together, and the matrix pipe idles meanwhile: 2.78 k cycles per K-tile for 2.05 k of MFMA work.
    // Waves 0-3 therefore run ONE PHASE AHEAD of waves 4-7 in their MFMA work: per barrier interval they read the phase's
    // fragments right AFTER the barrier that publishes them and multiply at once ([barrier] read, MFMA), while waves 4-7
    // keep the original order ([barrier] MFMA of the previous phase, read).  Inside an interval one half's LDS burst and
    // barrier wait sits beside the other half's MFMAs.  Both halves read a phase's data in the same interval and issue the
    // same LDS-DMA groups in the same interval (the ahead half carries in its MFMA block the groups the other half issues
    // in the block it runs meanwhile), so LDS lifetimes, counted vmcnt waits and barrier counts are unchanged.
#ifdef UMR_EXP_AHEAD_HI
    const bool ahead = (stagger != 0) && (w >= 4);
#else
    const bool ahead = (stagger != 0) && (w < 4);
#endif

    // tile sequence of this workgroup: virtual ids pw, pw + G, pw + 2G, ...; workgroups of one XCD (blockIdx % 8)
    // own a contiguous run of G/8 ids per round, so neighbouring tiles share that XCD's L2
    const int G = gridDim.x;
#ifdef UMR_EXP_NO_XCD_REMAP   // experiment (tools/probe/xcd_remap_fetch.sh): consecutive tiles on consecutive workgroups = on different XCDs
    const int pw = (int)blockIdx.x;
#else
    const int pw = ((G & 7) == 0) ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#endif
    if (pw >= total_tiles) return;
    const int n_my = (total_tiles - pw + G - 1) / G;

    const int lrow = lane >> 3, lchk = lane & 7;
    const int hw = (CONV != 0) ? p.Ho * p.Wo : 1;
    const int apix = X3 ? 3 * p.Cin : p.Cin;   // conv: bf16 elements per input pixel
    // ---- per-lane staging constants (tile independent).  Group-local row lr = (w*2+i)*8 + lane/8.
    unsigned vo[4][2];         // [group A0,B0,B1,A1][i]
    int trow_a[2][2];          // tile row of A group gi, instruction i (for the conv halo masks)
    unsigned a_tapmask[2][2];  // conv: bit t set <=> tap t of this row lies inside the image (0 for rows >= M)
    unsigned a_eff[2][2];
    int lds_row[4][2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (w * 2 + i) * 8 + lrow;
            {   // A group gi = m-half
                const int trow = (lr >> 6) * 128 + gi * 64 + (lr & 63);
                const int g = gi == 0 ? 0 : 3;
                lds_row[g][i] = (((w * 2 + i) * 8) >> 6) * 128 + gi * 64 + (((w * 2 + i) * 8) & 63);
                const int gch = lchk ^ ((trow >> 1) & 7);
                trow_a[gi][i] = trow;
                vo[g][i] = (CONV == 0) ? (unsigned)(((int64_t)trow * p.lda) * SZ + gch * 16)
                                       : (unsigned)((int64_t)trow * apix * SZ + gch * 16);
                a_eff[gi][i] = vo[g][i];
                a_tapmask[gi][i] = 0x1FFu;
            }
            {   // B group gi = n-half
                const int trow = (lr >> 5) * 64 + gi * 32 + (lr & 31);
                const int g = 1 + gi;
                lds_row[g][i] = (((w * 2 + i) * 8) >> 5) * 64 + gi * 32 + (((w * 2 + i) * 8) & 31);
                const int gch = lchk ^ ((trow >> 1) & 7);
                vo[g][i] = (unsigned)(((int64_t)trow * p.ldb) * SZ + gch * 16);
            }
        }
    }

    const int ktiles_per_tap = (CONV == 0) ? 0 : p.Cin / BK2;
    // X3: npairs = 6 (fp32-grade) or 3 (UMR_F32_X3_FAST: only (m,h) (h,m) (h,h), products to 2^-16)
    const int kt_total = (CONV == 0) ? p.K / BK2 : 9 * ktiles_per_tap;   // K % 64 == 0 guaranteed by the dispatcher
    const int kb = sp * kt_per_arg;                                      // first K-tile of this workgroup's run (0 without a split)
    const int kt_mine = (X3 && kt_per_arg > 0) ? ((kt_total - kb < kt_per_arg) ? kt_total - kb : kt_per_arg) : kt_total;
    // X3: npairs = 6 (fp32-grade) or 3 (UMR_F32_X3_FAST: only (m,h) (h,m) (h,h), products to 2^-16)
    const int nt = kt_mine * (X3 ? npairs : 1);

    // ---- staging side state
    __amdgpu_buffer_rsrc_t rsA, rsB;
    int s_it = 0;                     // tile iteration the cursor is in
    int st_tile = 0, st_tap = 0, st_ci = 0, st_par = 0;
    int st_pp = 0, st_k = 0;          // X3: plane pair within the K-tile, K-tile index (plain)
    unsigned soffA = 0, soffB = 0;
    auto clamp31 = [](int64_t v) -> int { return v > 0x7FFFFFFFll ? 0x7FFFFFFF : (v < 0 ? 0 : (int)v); };
    auto stage_setup = [&](int it) {
        const bool live = it < n_my;
        const int v = live ? it * G + pw : 0;
        const int tm = v / tiles_n, tn = v - tm * tiles_n;
        const int m0 = tm * bm, n0 = tn * BN2;
        // the cursor starts at the first K-tile of the workgroup's run (conv K order: channel chunk major, tap minor)
        st_tile = 0; st_pp = 0; st_k = kb;
        st_ci = (CONV == 0) ? 0 : kb / 9; st_tap = (CONV == 0) ? 0 : kb - st_ci * 9;
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + (int64_t)n0 * p.ldb * SZ), 0,
                                                live ? clamp31((int64_t)(p.N - n0) * p.ldb * SZ) : 0, 0x00020000);
        if (CONV == 0) {
            rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + (int64_t)m0 * p.lda * SZ), 0,
                                                    live ? clamp31((int64_t)((p.M - m0 < bm) ? (p.M - m0) : bm) * p.lda * SZ) : 0, 0x00020000);
        } else {
            // stride-1 'same' conv: input pixel index == output row index; origin = tap (-1,-1) of tile row 0
            rsA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + ((int64_t)m0 - (p.W + 1)) * apix * SZ), 0,
                                                    live ? 0x7FFFFFFF : 0, 0x00020000);
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + trow_a[gi][i];
                    const int rem = m % hw;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    unsigned mask = 0;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int iy = oy - 1 + tap / 3, ix = ox - 1 + tap % 3;
                        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) mask |= 1u << tap;
                    }
                    a_tapmask[gi][i] = (live && m < p.M) ? mask : 0u;
                }
        }
    };
    // called when group 0 (A0) of a new K-tile is about to be issued
    auto stage_prep = [&]() {
        if (st_tile == nt) {   // roll over into the next output tile of this workgroup
            ++s_it;
            stage_setup(s_it);
        }
        // X3: planes of this K-tile's pair, 2 bits per pair
        // pairs ordered so that equal A planes are consecutive -- (h,h) (h,m) (h,l) (m,h) (m,m) (l,h): the second and third
        // read of an A-plane tile come 1.5 us after the first and hit L2 (+1.8 % on the head conv against the order that
        // alternated planes, tools/probe/x3_order_ab.sh); three-term mode: (h,h) (h,m) (m,h)
        const int pa = X3 ? (((npairs == 6 ? 0x940 : 0x010) >> (2 * st_pp)) & 3) : 0;   // A planes 0,0,0,1,1,2  |  0,0,1
        const int pb = X3 ? (((npairs == 6 ? 0x124 : 0x004) >> (2 * st_pp)) & 3) : 0;   // B planes 0,1,2,0,1,0  |  0,1,0
        const bool next_k = !X3 || st_pp == npairs - 1;
        if (X3) st_pp = next_k ? 0 : st_pp + 1;
        if (CONV == 0) {
            const int kt = X3 ? st_k : st_tile;
            soffA = (unsigned)((kt * BK2 + pa * p.K) * SZ);
            soffB = (unsigned)((kt * BK2 + pb * p.K) * SZ);
            if (X3 && next_k) ++st_k;
        } else {
            const int c0 = st_ci * BK2;
            const int ky = st_tap / 3, kx = st_tap - ky * 3;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a_eff[gi][i] = ((a_tapmask[gi][i] >> st_tap) & 1u) ? vo[gi == 0 ? 0 : 3][i] : OOB;
            soffA = (unsigned)(((ky * p.W + kx) * apix + pa * p.Cin + c0) * SZ);
            soffB = (unsigned)((pb * 9 * p.Cin + st_tap * p.Cin + c0) * SZ);
            if (next_k) { if (++st_tap == 9) { st_tap = 0; ++st_ci; } }
        }
    };
    auto stage_issue = [&](auto gtag, auto itag) {
        constexpr int Gp = decltype(gtag)::value, I = decltype(itag)::value;
        const unsigned v = (Gp == 0) ? a_eff[0][I] : (Gp == 3) ? a_eff[1][I] : vo[Gp][I];
        char* dst = smem + st_par * BUF2;
        if (Gp == 0 || Gp == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, UMR_LDS_PTR(dst + lds_row[Gp][I] * ROWB2), 16, v, soffA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, UMR_LDS_PTR(dst + TILE2 + lds_row[Gp][I] * ROWB2), 16, v, soffB, 0, 0);
        if (Gp == 3 && I == 1) { ++st_tile; st_par ^= 1; }
    };
#define STAGE_DMA(Gp, I) stage_issue(std::integral_constant<int, Gp>{}, std::integral_constant<int, I>{})

    f32x4 acc[8][4];
    const int wr = w >> 2, wc = w & 3;
    const int dead_blocks = (CONV == 0 && wr == 1) ? ((BM2 - bm) >> 4) : 0;   // wave-uniform: 0, 2 or 4
    const int frow = lane & 15, fq = lane >> 4;
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

    bf16x8 fa[2][4], fb0[2][2], fb1[2][2];

#define PHASE_SYNC_N(N)                                            \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");          \
    __builtin_amdgcn_s_barrier();                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);
#define MFMA(ACC, BF, AF) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF, AF, ACC, 0, 0, 0)
#ifndef UMR_EXP_PRIO_MODE
#define UMR_EXP_PRIO_MODE 1
#endif
    // priority policy of the MFMA clusters: 0 = raise around every cluster (flips); 1 / 2 = static priority 1 for waves 4-7 /
    // waves 0-3 and no flips; 3 = none.  Measured on the head conv (two-phase K-tile, same box, tools/probe/prio_ab.sh): 30.9 /
    // 30.2 / 31.1 / 31.2 ms -- the guide's "static priority for the younger half" (Two waves per SIMD, item 4).  The four-phase
    // plain GEMMs lose 4-10 % without the flips (their LDS-DMA pieces are issued inside the raised cluster) and keep them.
    constexpr int PRIO_MODE = P2 ? UMR_EXP_PRIO_MODE : 0;
#define QPRIO(x) if (PRIO_MODE == 0) __builtin_amdgcn_s_setprio(x);
#define QUADRANT_D(M0, N0, FB, DMA_A, DMA_B)                                                         \
    QPRIO(1)                                                                                        \
    MFMA(acc[M0 + 0][N0 + 0], FB[0][0], fa[0][0]); MFMA(acc[M0 + 0][N0 + 1], FB[0][1], fa[0][0]);   \
    MFMA(acc[M0 + 1][N0 + 0], FB[0][0], fa[0][1]); MFMA(acc[M0 + 1][N0 + 1], FB[0][1], fa[0][1]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    DMA_A;                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MFMA(acc[M0 + 2][N0 + 0], FB[0][0], fa[0][2]); MFMA(acc[M0 + 2][N0 + 1], FB[0][1], fa[0][2]);   \
    MFMA(acc[M0 + 3][N0 + 0], FB[0][0], fa[0][3]); MFMA(acc[M0 + 3][N0 + 1], FB[0][1], fa[0][3]);   \
    MFMA(acc[M0 + 0][N0 + 0], FB[1][0], fa[1][0]); MFMA(acc[M0 + 0][N0 + 1], FB[1][1], fa[1][0]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    DMA_B;                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MFMA(acc[M0 + 1][N0 + 0], FB[1][0], fa[1][1]); MFMA(acc[M0 + 1][N0 + 1], FB[1][1], fa[1][1]);   \
    MFMA(acc[M0 + 2][N0 + 0], FB[1][0], fa[1][2]); MFMA(acc[M0 + 2][N0 + 1], FB[1][1], fa[1][2]);   \
    MFMA(acc[M0 + 3][N0 + 0], FB[1][0], fa[1][3]); MFMA(acc[M0 + 3][N0 + 1], FB[1][1], fa[1][3]);   \
    QPRIO(0)
#define QUADRANT(M0, N0, FB, Gp) QUADRANT_D(M0, N0, FB, STAGE_DMA(Gp, 0), STAGE_DMA(Gp, 1))
    // bm < 256: the waves that own tile rows 128..255 (wr == 1) have `dead_blocks` (2 at bm 224, 4 at bm 192) 16-row blocks without
    // rows at the end of their 128: the upper quadrants (blocks 4..7) run their first two blocks only, or only their DMA slots
#define QUADRANT_S(M0, N0, FB, Gp)                                                                  \
    QPRIO(1)                                                                                        \
    MFMA(acc[M0 + 0][N0 + 0], FB[0][0], fa[0][0]); MFMA(acc[M0 + 0][N0 + 1], FB[0][1], fa[0][0]);   \
    MFMA(acc[M0 + 1][N0 + 0], FB[0][0], fa[0][1]); MFMA(acc[M0 + 1][N0 + 1], FB[0][1], fa[0][1]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    STAGE_DMA(Gp, 0);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MFMA(acc[M0 + 0][N0 + 0], FB[1][0], fa[1][0]); MFMA(acc[M0 + 0][N0 + 1], FB[1][1], fa[1][0]);   \
    MFMA(acc[M0 + 1][N0 + 0], FB[1][0], fa[1][1]); MFMA(acc[M0 + 1][N0 + 1], FB[1][1], fa[1][1]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    STAGE_DMA(Gp, 1);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    QPRIO(0)
#define QUADRANT_HI(N0, FB, Gp)                                                  \
    if (dead_blocks == 0) { QUADRANT(4, N0, FB, Gp) }                            \
    else if (dead_blocks == 2) { QUADRANT_S(4, N0, FB, Gp) }                     \
    else { STAGE_DMA(Gp, 0); STAGE_DMA(Gp, 1); }

    // the same for the two-phase bodies (explicit DMA slots): all four blocks, the first two, or the DMA slots only
#define QUADRANT_SD(M0, N0, FB, DMA_A, DMA_B)                                                        \
    QPRIO(1)                                                                                        \
    MFMA(acc[M0 + 0][N0 + 0], FB[0][0], fa[0][0]); MFMA(acc[M0 + 0][N0 + 1], FB[0][1], fa[0][0]);   \
    MFMA(acc[M0 + 1][N0 + 0], FB[0][0], fa[0][1]); MFMA(acc[M0 + 1][N0 + 1], FB[0][1], fa[0][1]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    DMA_A;                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MFMA(acc[M0 + 0][N0 + 0], FB[1][0], fa[1][0]); MFMA(acc[M0 + 0][N0 + 1], FB[1][1], fa[1][0]);   \
    MFMA(acc[M0 + 1][N0 + 0], FB[1][0], fa[1][1]); MFMA(acc[M0 + 1][N0 + 1], FB[1][1], fa[1][1]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    DMA_B;                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    QPRIO(0)
#define QUADRANT_HID(N0, FB, DMA_A, DMA_B)                                       \
    if (dead_blocks == 0) { QUADRANT_D(4, N0, FB, DMA_A, DMA_B) }                \
    else if (dead_blocks == 2) { QUADRANT_SD(4, N0, FB, DMA_A, DMA_B) }          \
    else { DMA_A; DMA_B; }

#define PHASE_SYNC() PHASE_SYNC_N(6)
    // Two-phase form of the same K-tile (PH2): phases (Q0,Q1) and (Q2,Q3) merged -- 2 barriers instead of 4.  All waves
    // of the workgroup move in lock-step (read, wait, barrier, MFMA), and a wave cannot run far ahead of its queued
    // MFMAs, so the matrix pipe idles for the read + barrier latency of every phase; longer phases halve that share.
    // The counted wait of a phase publishes what the NEXT phase reads.  P0(t) reads A0,B0,B1 of tile t, P1(t) reads A1:
    //   P1(t) issues A0,B0,B1 of tile t+2 (their regions were last read in P0(t)),  P0(t+1) issues A1 of tile t+2;
    //   wait of P1(t): A0,B0,B1(t+1) landed, A1(t+1) may fly -> vmcnt(2);  wait of P0(t): A1(t) landed, A0,B0,B1(t+1) may
    //   fly -> vmcnt(6).  Every group has one K-tile (two phases) to land.
#ifdef UMR_NT256P_TIMESTAMPS
    const bool dbg = (p.rows_per_batch == -9) && blockIdx.x == 0 && tid == 0;
    unsigned long long* dbgp = (unsigned long long*)p.rowbias;
    bool ph_rec = false;   // stamps inside one K-tile (4-phase form): slots 128..135
    // slots 136..139: shader-clock and 100-MHz real-time stamps around the whole kernel -> the clock the chip holds under
    // this load = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6; tools/power_probe.py)
    if (dbg) { dbgp[136] = __builtin_readcyclecounter(); dbgp[137] = __builtin_amdgcn_s_memrealtime(); }
#define PT(k) do { if (dbg && ph_rec) dbgp[128 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PT(k) do { } while (0)
#endif
    auto tile_body2 = [&](const char* sbuf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fb0[ks][i] = B_FRAG(ks, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, i);
#pragma unroll
            for (int i = 0; i < 2; ++i) fb1[ks][i] = B_FRAG(ks, 2 + i);
        }
        PHASE_SYNC_N(6);
        QUADRANT_D(0, 0, fb0, STAGE_DMA(3, 0), STAGE_DMA(3, 1))
        QUADRANT_D(0, 2, fb1, (void)0, (void)0)
        stage_prep();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, 4 + i);
        PHASE_SYNC_N(2);
        QUADRANT_HID(2, fb1, STAGE_DMA(0, 0); STAGE_DMA(0, 1), STAGE_DMA(1, 0); STAGE_DMA(1, 1))
        QUADRANT_HID(0, fb0, STAGE_DMA(2, 0), STAGE_DMA(2, 1))
    };
    auto tile_body = [&](const char* sbuf) {
        if (PH2) { tile_body2(sbuf); return; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fb0[ks][i] = B_FRAG(ks, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, i);
        }
        PHASE_SYNC();
        PT(0);
        QUADRANT(0, 0, fb0, 2)
        PT(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) fb1[ks][i] = B_FRAG(ks, 2 + i);
        PHASE_SYNC();
        PT(2);
        QUADRANT(0, 2, fb1, 3)
        PT(3);
        stage_prep();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, 4 + i);
        PHASE_SYNC();
        PT(4);
        QUADRANT_HI(2, fb1, 0)
        PT(5);
        PHASE_SYNC();
        PT(6);
        QUADRANT_HI(0, fb0, 1)
        PT(7);
    };

#define READ_WAIT()                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);
#define END_SYNC_N(N)                                              \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");          \
    __builtin_amdgcn_s_barrier();                                  \
    __builtin_amdgcn_sched_barrier(0);
    // ---- waves 0-3 when staggered: per phase [read, MFMA (+ the DMA groups the other half issues meanwhile), wait, barrier]
    auto tile_body2_ahead = [&](const char* sbuf) {
        stage_prep();                       // K-tile t+1 (all four groups are issued during this body)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fb0[ks][i] = B_FRAG(ks, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, i);
#pragma unroll
            for (int i = 0; i < 2; ++i) fb1[ks][i] = B_FRAG(ks, 2 + i);
        }
        READ_WAIT();
        QUADRANT_D(0, 0, fb0, STAGE_DMA(0, 0); STAGE_DMA(0, 1), STAGE_DMA(1, 0); STAGE_DMA(1, 1))
        QUADRANT_D(0, 2, fb1, STAGE_DMA(2, 0), STAGE_DMA(2, 1))
        END_SYNC_N(6);                      // A1(t) landed (own part); A0,B0,B1(t+1) may fly
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, 4 + i);
        READ_WAIT();
        QUADRANT_HID(2, fb1, STAGE_DMA(3, 0), STAGE_DMA(3, 1))
        QUADRANT_HID(0, fb0, (void)0, (void)0)
        END_SYNC_N(2);                      // A0,B0,B1(t+1) landed; A1(t+1) may fly
    };
    auto tile_body4_ahead = [&](const char* sbuf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fb0[ks][i] = B_FRAG(ks, i);
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, i);
        }
        READ_WAIT();
        QUADRANT(0, 0, fb0, 1)              // B0 of K-tile t+1 (the other half: in its Q(4,0) of K-tile t-1)
        END_SYNC_N(6);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) fb1[ks][i] = B_FRAG(ks, 2 + i);
        READ_WAIT();
        QUADRANT(0, 2, fb1, 2)
        END_SYNC_N(6);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[ks][i] = A_FRAG(ks, 4 + i);
        READ_WAIT();
        QUADRANT(4, 2, fb1, 3)
        END_SYNC_N(6);
        stage_prep();                       // K-tile t+2
        QUADRANT(4, 0, fb0, 0)
        END_SYNC_N(6);
    };

    if (PRIO_MODE == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
    if (PRIO_MODE == 2 && w < 4) __builtin_amdgcn_s_setprio(1);
    // prologue: the six groups the steady-state schedule has already issued when the first K-tile starts
    stage_setup(0);
    stage_prep();
    STAGE_DMA(0, 0); STAGE_DMA(0, 1); STAGE_DMA(1, 0); STAGE_DMA(1, 1);
    STAGE_DMA(2, 0); STAGE_DMA(2, 1); STAGE_DMA(3, 0); STAGE_DMA(3, 1);
    if (ahead) {
        // the ahead half issues inside its first MFMA block what the other half pre-issues here
        if (PH2) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                     // A0,B0,B1 of K-tile 0 landed
        } else {
            stage_prep();
            STAGE_DMA(0, 0); STAGE_DMA(0, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                     // A0,B0 of K-tile 0 landed
        }
    } else {
        stage_prep();
        STAGE_DMA(0, 0); STAGE_DMA(0, 1); STAGE_DMA(1, 0); STAGE_DMA(1, 1);
        if (PH2) { STAGE_DMA(2, 0); STAGE_DMA(2, 1); }   // two-phase schedule: B1 of tile 1 is pre-issued as well
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    // Epilogue of the fast class (round 5): every wave turns ITS 16 x 64 blocks of the tile from the fragment layout into row-contiguous
    // 16-byte chunks through a private 2-KiB staging block -- no workgroup barrier anywhere in the epilogue (LDS serves one wave's
    // accesses in order, so a block's writes, its reads and the next block's writes need no wait between them).  Copy-out: lane ->
    // rows er, er + 8 of the block, chunk ec of the wave's 128-byte row segment (a store instruction writes 8 full 128-byte lines).
    // Byte offsets into C / aux relative to the tile origin are constants of the kernel; chunks are XOR-swizzled with (row >> 1) & 7.
    const int er = lane >> 3, ec = lane & 7;
    const unsigned co_lane = ((unsigned)((w >> 2) * 128 + er) * (unsigned)p.ldc + (unsigned)((w & 3) * 64 + ec * 8)) * 2u;
    const unsigned ax_lane = ((unsigned)((w >> 2) * 128 + er) * (unsigned)p.ldaux + (unsigned)((w & 3) * 64 + ec * 8)) * 2u;
    constexpr int WST_BYTES = 2048;              // per wave: 16 rows x 128 bytes
    constexpr int RW_OFF = 8 * WST_BYTES;        // fused row reduction: its weights, 2 tiles x 2 rows x 1 KiB, behind the waves' blocks
    float* stg = (float*)(smem + STG_OFF);
    // one epilogue pass writes the 32 rows {wr*128 + mt*16 + 0..15} x 256 columns of the tile into the staging region
    auto stage_rows = [&](auto mtag) {
        constexpr int MT = decltype(mtag)::value;
        const int lr = wr * 16 + frow;
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) {
            const int ch = (wc * 16 + ntl * 4 + fq) ^ frow;
            *(f32x4*)(stg + lr * 256 + ch * 4) = acc[MT][ntl];
        }
    };
    int c_par = 0;
    // Fast / GELU classes: the accumulators start from the bias instead of zero.  The bias vectors of the NEXT tile are
    // fetched at the start of each epilogue (first tile: here): a load issued where it is needed would be waited for at once,
    // and vmcnt retires in order -- that wait drained the next tile's prefetched K-tiles (~1900 cycles per tile, measured
    // with s_memtime stamps).  Fetched there, everything issued before it has long landed when the epilogue ends.
    constexpr bool BIAS_INIT = (EPI == 3 || EPI == 4 || EPI == 5 || EPI == 6);
    f32x4 bqn[4];
    auto fetch_bias = [&](int it_) {
        const int v_ = it_ * G + pw;
        const int tn_ = v_ - (v_ / tiles_n) * tiles_n;
#pragma unroll
        for (int ntl = 0; ntl < 4; ++ntl) {
            const int n = tn_ * BN2 + wc * 64 + ntl * 16 + fq * 4;
            bqn[ntl] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((p.flags & UMR_EPI_BIAS) && it_ < n_my && n < p.N) bqn[ntl] = *(const f32x4*)(p.bias + n);
        }
    };
    if (BIAS_INIT) fetch_bias(0);
    // -DUMR_NT256P_TIMESTAMPS (tools/probe/ts_probe.py builds its own library with it): workgroup 0 / thread 0 writes
    // s_memtime stamps of the first 16 tiles to the rowbias pointer when rows_per_batch == -9 -- how the per-tile budget was taken apart
#ifdef UMR_NT256P_TIMESTAMPS
#define TS(slot) do { if (dbg && it < 16) dbgp[it * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define TS(slot) do { } while (0)
#endif
#pragma unroll 1
    for (int it = 0; it < n_my; ++it) {
        TS(0);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = BIAS_INIT ? bqn[j] : f32x4{0.f, 0.f, 0.f, 0.f};
        TS(1);
        if (RED && w < 2) {
            // reduction weights of this tile's 256 columns: row c = w of red_w as one 1-KiB LDS-DMA (columns >= N and a
            // missing second row read as zeros)
            const int v_ = it * G + pw;
            const int n0_ = (v_ - (v_ / tiles_n) * tiles_n) * BN2;
            const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(p.red_w + (int64_t)w * p.N + n0_), 0, (w < p.red_c) ? clamp31((int64_t)(p.N - n0_) * 4) : 0, 0x00020000);
            // (fast class: double-buffered by tile parity -- its epilogue has no barrier, a wave may start the next tile while another
            // still reads this tile's weights; the copy two tiles on is separated from those reads by that tile's K-loop barriers)
            const int rw_off = (EPI == 3) ? RW_OFF + (it & 1) * 2048 : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, UMR_LDS_PTR(smem + STG_OFF + rw_off + w * 1024), 16, (unsigned)(lane * 16), 0, 0, 0);
        }
        if (ahead) {
#pragma unroll 1
            for (int t = 0; t < nt; ++t) {
                if (PH2) tile_body2_ahead(smem + c_par * BUF2); else tile_body4_ahead(smem + c_par * BUF2);
                c_par ^= 1;
            }
        } else {
#pragma unroll 1
            for (int t = 0; t < nt; ++t) {
#ifdef UMR_NT256P_TIMESTAMPS
                ph_rec = (it == 4 && t == 5);
#endif
                tile_body(smem + c_par * BUF2);
                c_par ^= 1;
#ifdef UMR_NT256P_TIMESTAMPS
                if (t == 0) TS(2);
                if (t == 1) TS(3);
#endif
            }
        }
        TS(4);
        // ---- epilogue of output tile `it`; the next tile's first K-tiles are already in flight
        const int v = it * G + pw;
        const int tm = v / tiles_n, tn = v - tm * tiles_n;
        const int m0 = tm * bm, n0 = tn * BN2;
        const int m_end = (p.M - m0 < bm) ? p.M : m0 + bm;   // rows of this tile that exist
        if (EPI == 3) {
            // fast class: (bias already in the accumulators), aux add / ReLU mask, ReLU.  The math runs on the accumulators in
            // their fragment layout, the results are staged as bf16 -- 64 tile rows per pass, 4 passes / 8 barriers, half the
            // LDS traffic of an f32 staging -- and leave as plain 16-byte copies.
            fetch_bias(it + 1);
            // ReLU without a branch, on the bf16 pair AFTER rounding: as 16-bit integers positive bf16 values are positive and everything
            // with the sign bit is negative, so max with (0, 0) is ReLU (rounding keeps the sign: same bits as ReLU before rounding), max
            // with (-32768, -32768) the identity.  One v_pk_max_i16 per two values instead of two v_max_f32.
            typedef __attribute__((ext_vector_type(2))) short s16x2;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const short relu_lo = (p.act == UMR_ACT_RELU) ? (short)0 : (short)-32768;
            const s16x2 relu16 = {relu_lo, relu_lo};
            auto pack4 = [&](const f32x4& v) -> u32x2 {
                // (as asm: with the integer max behind it the compiler no longer pairs the two conversions into one instruction)
                unsigned lo, hi;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
                s16x2 a = __builtin_bit_cast(s16x2, lo), b = __builtin_bit_cast(s16x2, hi);
                if (AUXM == 0) {   // (an aux operand excludes ReLU: umr_nt256p_fast_epilogue)
                    a = __builtin_elementwise_max(a, relu16);
                    b = __builtin_elementwise_max(b, relu16);
                }
                return u32x2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
            };
            // fused row reduction (umr_gemm_desc.red_*): this lane's 16 columns of the reduction weights; the partial dot
            // products of a row are summed over the 4 lanes that share it (fq) and written per 64-column wave slice
            // They were brought into the (idle) staging region by LDS-DMA at the start of the tile -- a global load here would
            // be waited for on the spot, behind the next tile's prefetched K-tiles -- and are read out before pass 0 reuses it.
            // Round 4: the dot products run on the MATRIX pipe.  Per 16-row block the wave's 16 x 64 slice of C (bf16, as stored) is the
            // B operand of two 16x16x32 MFMAs -- a lane holds row frow, columns ntl*16 + 4 fq + e: eight values per pair of ntl, a
            // permutation of the k index that the weight fragment simply shares -- against A = the reduction weights, row c = lane & 15
            // (rows >= red_c are zero), as hi + lo bf16 halves (16 significant bits).  D[c][row] lands in the lanes fq == 0, registers
            // 0 / 1 = outputs 0 / 1: no shuffles.  32 MFMAs per wave and tile instead of ~100 conversions + FMAs per lane and pass:
            // the VALU form made this epilogue 26.5 k cycles of a 55 k-cycle tile (s_memtime stamps, profiles/r04_ts_probe_head_1x1.txt).
            bf16x8 rwh[2], rwl[2];
            float* ro_lane = nullptr;     // this lane's slot of the partial sums for row block 0 (lanes fq == 0 of a live column slice)
            if (RED) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
                    if (frow < 2) {
                        const char* rwp = smem + STG_OFF + RW_OFF + (it & 1) * 2048 + frow * 1024;
                        a = *(const f32x4*)(rwp + (wc * 64 + (2 * pr) * 16 + fq * 4) * 4);
                        b = *(const f32x4*)(rwp + (wc * 64 + (2 * pr + 1) * 16 + fq * 4) * 4);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = e < 4 ? a[e] : b[e - 4];
                        const bf16_t hi = (bf16_t)x;
                        rwh[pr][e] = hi;
                        rwl[pr][e] = (bf16_t)(x - (float)hi);
                    }
                }
                if (fq == 0 && n0 + wc * 64 < p.N)
                    ro_lane = p.red_out + ((int64_t)(tn * 4 + wc) * p.M + (m0 + wr * 128 + frow)) * p.red_c;
            }
            // partial dot products of TWO 16-row blocks (mt_, mt_ + 1): a01 / a23 = the lane's bf16 values of column blocks (0,1) / (2,3)
            // of the first, b01 / b23 of the second.  Each block's four MFMAs depend on each other (same order as ever: bit-identical);
            // the two chains are interleaved so that no MFMA waits for its predecessor's result.  (Round 5: one chain per block ended in
            // `s_nop 7`, and the address of every 4-byte result took two quarter-rate 32-bit multiplies.)
            auto row_reduce2 = [&](const bf16x8& a01, const bf16x8& a23, const bf16x8& b01, const bf16x8& b23, int mt_) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                f32x4 ra = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rwh[0], a01, z, 0, 0, 0);
                f32x4 rb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rwh[0], b01, z, 0, 0, 0);
                ra = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rwl[0], a01, ra, 0, 0, 0);
                rb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rwl[0], b01, rb, 0, 0, 0);
                ra = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rwh[1], a23, ra, 0, 0, 0);
                rb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rwh[1], b23, rb, 0, 0, 0);
                ra = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rwl[1], a23, ra, 0, 0, 0);
                rb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rwl[1], b23, rb, 0, 0, 0);
                const int m = m0 + wr * 128 + mt_ * 16 + frow;
                if (ro_lane != nullptr) {
                    float* ro = ro_lane + (int64_t)(mt_ * 16) * p.red_c;
                    if (p.red_c == 2) {
                        if (m < m_end) *(f32x2*)ro = f32x2{ra[0], ra[1]};
                        if (m + 16 < m_end) *(f32x2*)(ro + 32) = f32x2{rb[0], rb[1]};
                    } else {
                        if (m < m_end) ro[0] = ra[0];
                        if (m + 16 < m_end) ro[16] = rb[0];
                    }
                }
            };
            // The aux operand is applied to the staged bf16 values in the COPY-OUT layout: 16-byte loads of whole 128-byte row segments
            // (in the fragment layout a lane reads 8 bytes at a row stride: 4x the cache lines per instruction, and the epilogue of a
            // residual-add GEMM took 25 k cycles against 6.7 k without aux).  Loads are issued two blocks ahead (their first touch is
            // an HBM round trip).  ReLU mask (AUXM 2): masking commutes with the rounding -- bit-identical.  Residual add (AUXM 1):
            // bf16(bf16(acc) + aux), i.e. the GEMM result is rounded to the storage type before the residual is added, as a separate
            // Linear + add in bf16 would do.
            // C and aux are addressed through buffer descriptors of the TILE (origin = its first row and column, num_records = its live
            // rows): rows past the end are dropped / read as zeros by the bounds check, a lane whose 8 columns lie past N gets an
            // out-of-range offset, and the row of a block is a scalar added to the lane offset (in the VECTOR offset: the
            // instruction's scalar offset is excluded from the bounds check).
            const bool col_live = n0 + wc * 64 + ec * 8 < p.N;
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((T2*)p.C + (int64_t)m0 * p.ldc + n0), 0, p.no_store ? 0 : clamp31((int64_t)(m_end - m0) * p.ldc * 2), 0x00020000);
            const unsigned co_vo = col_live ? co_lane : OOB;
            __amdgpu_buffer_rsrc_t rsX = rsC;
            unsigned ax_vo = OOB;
            if (AUXM != 0) {
                rsX = __builtin_amdgcn_make_buffer_rsrc((void*)((T2*)p.aux + (int64_t)m0 * p.ldaux + n0), 0,
                                                        clamp31((int64_t)(m_end - m0) * p.ldaux * 2), 0x00020000);
                ax_vo = col_live ? ax_lane : OOB;
            }
            u32x4 axc[4][2];   // ring over row blocks; loads run two blocks ahead of their use
            auto load_auxc = [&](auto btag) {
                constexpr int MB = decltype(btag)::value;
                if (MB < 8 && AUXM != 0) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        axc[MB & 3][k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ax_vo + (unsigned)((MB * 16 + k * 8) * p.ldaux * 2), 0, 0);
                }
            };
            load_auxc(std::integral_constant<int, 0>{});
            load_auxc(std::integral_constant<int, 1>{});
            char* wst = smem + STG_OFF + w * WST_BYTES;
            const int wsw = (frow >> 1) & 7;
            char* wst_w = wst + frow * 128 + (fq & 1) * 8;                       // + (((ntl * 2 + (fq >> 1)) ^ wsw) << 4)
            const char* wst_r0 = wst + er * 128 + ((ec ^ ((er >> 1) & 7)) << 4);
            const char* wst_r1 = wst + (er + 8) * 128 + ((ec ^ (((er >> 1) + 4) & 7)) << 4);
            // row blocks MB, MB + 1 of the wave
            auto blocks2 = [&](auto btag) {
                constexpr int MB = decltype(btag)::value;
                load_auxc(std::integral_constant<int, MB + 2>{});
                load_auxc(std::integral_constant<int, MB + 3>{});
                u32x2 pk[2][4];      // the values AS STORED (bf16-rounded)
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int ntl = 0; ntl < 4; ++ntl) pk[mh][ntl] = pack4(acc[MB + mh][ntl]);
                if (RED) {
                    bf16x8 tq[2][2];
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            tq[mh][h] = __builtin_bit_cast(bf16x8, u32x4{pk[mh][2 * h][0], pk[mh][2 * h][1], pk[mh][2 * h + 1][0], pk[mh][2 * h + 1][1]});
                    row_reduce2(tq[0][0], tq[0][1], tq[1][0], tq[1][1], MB);
                }
#pragma unroll
                for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
                    for (int ntl = 0; ntl < 4; ++ntl) *(u32x2*)(wst_w + (((ntl * 2 + (fq >> 1)) ^ wsw) << 4)) = pk[mh][ntl];
                    asm volatile("" ::: "memory");     // (compiler only: the reads below see these writes, the next block's writes follow the reads)
                    u32x4 o[2];
                    o[0] = *(const u32x4*)wst_r0;
                    o[1] = *(const u32x4*)wst_r1;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if (AUXM == 2) {
                            // keep a bf16 where the mask operand is > 0, i.e. where its 16 bits read as a positive integer:
                            // min(a, 1) -> max(.., 0) is 1 or 0 per half, 0 - that is 0xFFFF or 0
                            const u32x4 a = axc[(MB + mh) & 3][k];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[k][e] &= pos_mask_bf16x2(a[e]);
                        }
                        if (AUXM == 1) {
                            const u32x4 a = axc[(MB + mh) & 3][k];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[k][e] = add_bf16x2(o[k][e], a[e]);
                        }
                        const unsigned so = (unsigned)(((MB + mh) * 16 + k * 8) * p.ldc * 2);
#ifdef UMR_EXP_NT_STORE   // experiment (tools/energy_probe.py): non-temporal C stores -- C is re-read only by a much later kernel
                        __builtin_amdgcn_raw_buffer_store_b128(o[k], rsC, co_vo + so, 0, 2);
#else
                        __builtin_amdgcn_raw_buffer_store_b128(o[k], rsC, co_vo + so, 0, 0);
#endif
                    }
                }
            };
            if (RED && p.no_store) {
                // inference / algebraic-backward form of the fused output layer: C itself is never stored, so nothing is staged
                // through LDS -- the dot products are taken on the bf16-rounded values in registers
#pragma unroll
                for (int mt = 0; mt < 8; mt += 2) {
                    bf16x8 tq[2][2];
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                        for (int ntl = 0; ntl < 4; ++ntl) {
                            const bf16x4 t = __builtin_bit_cast(bf16x4, pack4(acc[mt + mh][ntl]));
#pragma unroll
                            for (int e = 0; e < 4; ++e) tq[mh][ntl >> 1][(ntl & 1) * 4 + e] = t[e];
                        }
                    row_reduce2(tq[0][0], tq[0][1], tq[1][0], tq[1][1], mt);
                }
            } else {
                blocks2(std::integral_constant<int, 0>{}); TS(5); blocks2(std::integral_constant<int, 2>{});
                blocks2(std::integral_constant<int, 4>{}); blocks2(std::integral_constant<int, 6>{});
            }
            TS(6);
            TS(7);
        } else if (EPI == 4) {
            // GELU class (the transformer MLP): act == GELU with the pre-activation optionally saved to C2 (c2_mode 2), or
            // the GELU'-masked gradient (MASK_DGELU).  Same bf16 staging as the fast class, in its own instantiation so that
            // the erf code does not sit in the instruction stream of the conv / 1x1 kernels.  The GELU' factor is applied in
            // the copy-out layout (16-byte coalesced loads of the pre-activation, two passes ahead), to the stored bf16 value.
            fetch_bias(it + 1);
            constexpr bool dgelu = (AUXM == 2);   // instantiated as <0, 4, 0> (GELU forward) and <0, 4, 2> (GELU'-masked gradient)
            const bool two_out = !dgelu && p.c2_mode == 2;
            // Round 5: the fast class's barrier-free form (see there) -- wave-private 2-KiB staging blocks, per-tile buffer descriptors
            // for C / C2 / the pre-activation, row-contiguous 16-byte chunks.  Before: four passes with 2 (4 with C2) workgroup barriers each.
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            const bool col_live = n0 + wc * 64 + ec * 8 < p.N;
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
                (void*)((T2*)p.C + (int64_t)m0 * p.ldc + n0), 0, clamp31((int64_t)(m_end - m0) * p.ldc * 2), 0x00020000);
            const unsigned co_vo = col_live ? co_lane : OOB;
            __amdgpu_buffer_rsrc_t rsC2 = rsC, rsX = rsC;
            unsigned c2_vo = OOB, ax_vo = OOB;
            if (two_out) {
                rsC2 = __builtin_amdgcn_make_buffer_rsrc((void*)((T2*)p.C2 + (int64_t)m0 * p.ldc2 + n0), 0,
                                                         clamp31((int64_t)(m_end - m0) * p.ldc2 * 2), 0x00020000);
                c2_vo = col_live ? ((unsigned)(wr * 128 + er) * (unsigned)p.ldc2 + (unsigned)(wc * 64 + ec * 8)) * 2u : OOB;
            }
            if (dgelu) {
                rsX = __builtin_amdgcn_make_buffer_rsrc((void*)((T2*)p.aux + (int64_t)m0 * p.ldaux + n0), 0,
                                                        clamp31((int64_t)(m_end - m0) * p.ldaux * 2), 0x00020000);
                ax_vo = col_live ? ax_lane : OOB;
            }
            u32x4 axc[4][2];   // ring over row blocks; loads run two blocks ahead of their use
            auto load_auxc = [&](auto btag) {
                constexpr int MB = decltype(btag)::value;
                if (MB < 8 && dgelu) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        axc[MB & 3][k] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ax_vo + (unsigned)((MB * 16 + k * 8) * p.ldaux * 2), 0, 0);
                }
            };
            load_auxc(std::integral_constant<int, 0>{});
            load_auxc(std::integral_constant<int, 1>{});
            char* wst = smem + STG_OFF + w * WST_BYTES;
            const int wsw = (frow >> 1) & 7;
            char* wst_w = wst + frow * 128 + (fq & 1) * 8;                       // + (((ntl * 2 + (fq >> 1)) ^ wsw) << 4)
            const char* wst_r0 = wst + er * 128 + ((ec ^ ((er >> 1) & 7)) << 4);
            const char* wst_r1 = wst + (er + 8) * 128 + ((ec ^ (((er >> 1) + 4) & 7)) << 4);
            auto pack4g = [&](const f32x4& v) -> u32x2 {
                unsigned lo, hi;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
                return u32x2{lo, hi};
            };
            auto block4 = [&](auto btag) {
                constexpr int MB = decltype(btag)::value;
                load_auxc(std::integral_constant<int, MB + 2>{});
                if (two_out) {   // the pre-activation, as the backward pass wants it
#pragma unroll
                    for (int ntl = 0; ntl < 4; ++ntl) *(u32x2*)(wst_w + (((ntl * 2 + (fq >> 1)) ^ wsw) << 4)) = pack4g(acc[MB][ntl]);
                    asm volatile("" ::: "memory");
                    const u32x4 o0 = *(const u32x4*)wst_r0, o1 = *(const u32x4*)wst_r1;
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_raw_buffer_store_b128(o0, rsC2, c2_vo + (unsigned)((MB * 16) * p.ldc2 * 2), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(o1, rsC2, c2_vo + (unsigned)((MB * 16 + 8) * p.ldc2 * 2), 0, 0);
                }
#pragma unroll
                for (int ntl = 0; ntl < 4; ++ntl) {
                    f32x4 v = acc[MB][ntl];   // bias included (accumulator start value)
                    if (!dgelu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_sel<T2>(v[e]);
                    }
                    *(u32x2*)(wst_w + (((ntl * 2 + (fq >> 1)) ^ wsw) << 4)) = pack4g(v);
                }
                asm volatile("" ::: "memory");
                u32x4 o[2];
                o[0] = *(const u32x4*)wst_r0;
                o[1] = *(const u32x4*)wst_r1;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (dgelu) {
                        const u32x4 a = axc[MB & 3][k];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[k][e] = mul_dgelu_bf16x2(o[k][e], a[e]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o[k], rsC, co_vo + (unsigned)((MB * 16 + k * 8) * p.ldc * 2), 0, 0);
                }
            };
            block4(std::integral_constant<int, 0>{}); block4(std::integral_constant<int, 1>{});
            block4(std::integral_constant<int, 2>{}); block4(std::integral_constant<int, 3>{});
            block4(std::integral_constant<int, 4>{}); block4(std::integral_constant<int, 5>{});
            block4(std::integral_constant<int, 6>{}); block4(std::integral_constant<int, 7>{});
        } else if (EPI == 5 || EPI == 6) {
            // X3 classes: (bias already in the accumulators); f32-staged, 8 passes of 32 rows in a runtime loop (one copy of the
            // store code): the K loop is six times longer than the bf16 kernel's, the epilogue's share is small.
            // EPI 5 (the head layers, every K-split work item): ReLU, optional ReLU mask (f32 tensor or the leading plane of a plane
            // tensor), output either as f32 rows (UMR_EPI_OUT_F32) or again as three bf16 planes [h(N) | m(N) | l(N)] per row
            // (UMR_EPI_OUT_X3: lossless for f32 values, and what the next X3 layer stages directly).
            // EPI 6: everything else of include/umr.h (x3_epilogue_store8) in its own instantiation -- inlined into the lean
            // class it doubled the kernel's code and cost the head conv 7 % (I-cache, register pressure around the K loop).
            fetch_bias(it + 1);
            const float relu_floor = (p.act == UMR_ACT_RELU) ? 0.f : -INFINITY;
            const bool planes = (p.flags & UMR_EPI_OUT_X3) != 0;
            const bool maskf = (p.flags & UMR_EPI_MASK_RELU) != 0;
            const bool maskp = (p.flags & UMR_EPI_AUX_X3) != 0;
            if (RED) {
                // fused output layer (1024 -> {1,2}) of a head at inference: dot products of the f32 values with the reduction
                // weights (brought into the staging region by LDS-DMA at the start of the tile), C itself is never stored
                f32x4 rw[2][4];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ntl = 0; ntl < 4; ++ntl)
                        rw[c][ntl] = *(const f32x4*)(smem + STG_OFF + c * 1024 + (wc * 64 + ntl * 16 + fq * 4) * 4);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
                    for (int ntl = 0; ntl < 4; ++ntl) {
                        const f32x4 v = acc[mt][ntl];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = fmaxf(v[e], relu_floor);
                            rs0 += t * rw[0][ntl][e];
                            rs1 += t * rw[1][ntl][e];
                        }
                    }
                    rs0 += __shfl_xor(rs0, 16, 64); rs0 += __shfl_xor(rs0, 32, 64);
                    rs1 += __shfl_xor(rs1, 16, 64); rs1 += __shfl_xor(rs1, 32, 64);
                    const int m = m0 + wr * 128 + mt * 16 + frow;
                    if (fq == 0 && m < m_end && n0 + wc * 64 < p.N) {
                        float* ro = p.red_out + ((int64_t)(tn * 4 + wc) * p.M + m) * p.red_c;
                        ro[0] = rs0;
                        if (p.red_c == 2) ro[1] = rs1;
                    }
                }
            } else {
#pragma unroll 1
            for (int mt = 0; mt < 8; ++mt) {
                if (mt > 0) __syncthreads();
                switch (mt) {
                    case 0: stage_rows(std::integral_constant<int, 0>{}); break;
                    case 1: stage_rows(std::integral_constant<int, 1>{}); break;
                    case 2: stage_rows(std::integral_constant<int, 2>{}); break;
                    case 3: stage_rows(std::integral_constant<int, 3>{}); break;
                    case 4: stage_rows(std::integral_constant<int, 4>{}); break;
                    case 5: stage_rows(std::integral_constant<int, 5>{}); break;
                    case 6: stage_rows(std::integral_constant<int, 6>{}); break;
                    default: stage_rows(std::integral_constant<int, 7>{}); break;
                }
                __syncthreads();
                constexpr int UNROLL_J = (EPI == 6) ? 1 : 2;
#pragma clang loop unroll_count(UNROLL_J)
                for (int j = 0; j < 2; ++j) {
                    const int lr = (tid >> 5) + j * 16, cg = tid & 31;
                    const int m = m0 + (lr >> 4) * 128 + mt * 16 + (lr & 15), n = n0 + cg * 8;
                    if (m >= m_end || n >= p.N) continue;
                    const int sw = lr & 15;
                    f32x4 v0 = *(const f32x4*)(stg + lr * 256 + (((2 * cg) ^ sw) << 2));
                    f32x4 v1 = *(const f32x4*)(stg + lr * 256 + (((2 * cg + 1) ^ sw) << 2));
                    if constexpr (EPI == 6) {
                        x3_epilogue_store8<false>(p, m, n, v0, v1);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], relu_floor); v1[e] = fmaxf(v1[e], relu_floor); }
                        if (maskf) {   // ReLU-masked data gradient: aux = the activation whose sign decides (objectness_net.py:111-115)
                            f32x4 a0, a1;
                            if (maskp) {
                                const bf16x8 h = *(const bf16x8*)((const T2*)p.aux + (int64_t)m * p.ldaux + n);
                                a0 = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                                a1 = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
                            } else {
                                const float* ap = (const float*)p.aux + (int64_t)m * p.ldaux + n;
                                a0 = *(const f32x4*)ap; a1 = *(const f32x4*)(ap + 4);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v0[e] = a0[e] > 0.f ? v0[e] : 0.f; v1[e] = a1[e] > 0.f ? v1[e] : 0.f; }
                        }
                        const int64_t mo = m + (int64_t)sp * p.M;   // a split-K workgroup writes its raw sums to slab sp
                        if (planes) x3_planes_store8((T2*)p.C + mo * p.ldc + n, p.N, v0, v1);
                        else { float* cp = (float*)p.C + mo * p.ldc + n; *(f32x4*)cp = v0; *(f32x4*)(cp + 4) = v1; }
                    }
                }
            }
            }
        } else {
            // generic epilogue (every flag / activation / remap of include/umr.h): ONE copy of the store code in a
            // runtime loop over the passes -- unrolled it is ~100 KiB of instructions and runs out of the I-cache
            const bool vec_ok = ((p.N & 7) == 0) && ((p.ldc & 7) == 0) && ((p.ldc2 & 7) == 0) && ((p.ldaux & 7) == 0) &&
                                ((p.ldaux2 & 7) == 0);
#pragma unroll 1
            for (int mt = 0; mt < 8; ++mt) {
                if (mt > 0) __syncthreads();
                switch (mt) {
                    case 0: stage_rows(std::integral_constant<int, 0>{}); break;
                    case 1: stage_rows(std::integral_constant<int, 1>{}); break;
                    case 2: stage_rows(std::integral_constant<int, 2>{}); break;
                    case 3: stage_rows(std::integral_constant<int, 3>{}); break;
                    case 4: stage_rows(std::integral_constant<int, 4>{}); break;
                    case 5: stage_rows(std::integral_constant<int, 5>{}); break;
                    case 6: stage_rows(std::integral_constant<int, 6>{}); break;
                    default: stage_rows(std::integral_constant<int, 7>{}); break;
                }
                __syncthreads();
#pragma unroll 1
                for (int j = 0; j < 2; ++j) {
                    const int lr = (tid >> 5) + j * 16, cg = tid & 31;
                    const int m = m0 + (lr >> 4) * 128 + mt * 16 + (lr & 15), n = n0 + cg * 8;
                    if (m >= m_end || n >= p.N) continue;
                    const int sw = lr & 15;
                    const f32x4 v0 = *(const f32x4*)(stg + lr * 256 + (((2 * cg) ^ sw) << 2));
                    const f32x4 v1 = *(const f32x4*)(stg + lr * 256 + (((2 * cg + 1) ^ sw) << 2));
                    if (vec_ok) {
                        epilogue_store8<T2>(p, m, n, v0, v1);
                    } else {
                        epilogue_store<T2>(p, m, n, v0);
                        if (n + 4 < p.N) epilogue_store<T2>(p, m, n + 4, v1);
                    }
                }
            }
        }
        __syncthreads();
    }
    // the trailing (zero-fill) LDS-DMA groups must land before the LDS allocation is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef UMR_NT256P_TIMESTAMPS
    if (dbg) { dbgp[138] = __builtin_readcyclecounter(); dbgp[139] = __builtin_amdgcn_s_memrealtime(); }
#endif
#undef QUADRANT
#undef QUADRANT_S
#undef QUADRANT_HI
#undef QUADRANT_HID
#undef QUADRANT_SD
#undef TS
#undef PT
#undef QUADRANT_D
#undef MFMA
#undef PHASE_SYNC
#undef PHASE_SYNC_N
#undef STAGE_DMA
#undef READ_WAIT
#undef END_SYNC_N
#undef A_FRAG
#undef B_FRAG
}

// second half of a split-K plane GEMM: C = epilogue(sum over the ksplit slabs, in slab order) -- bitwise reproducible, and the
// one place where bias / aux / activation / output format of such a GEMM are applied (x3_epilogue_store8).
// One thread = 8 consecutive columns of one row.
__global__ __launch_bounds__(256) void x3_splitk_finish_kernel(const umr_gemm_desc p, const float* __restrict__ slabs, int ksplit, int64_t slab_stride) {
    const int n8 = p.N >> 3;
    const int64_t total = (int64_t)p.M * n8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / n8);
        const int n = (int)(idx - (int64_t)m * n8) * 8;
        const float* q = slabs + (int64_t)m * p.N + n;
        f32x4 v0 = *(const f32x4*)q, v1 = *(const f32x4*)(q + 4);
        for (int s_ = 1; s_ < ksplit; ++s_) {
            const float* qs = q + s_ * slab_stride;
            v0 += *(const f32x4*)qs; v1 += *(const f32x4*)(qs + 4);
        }
        x3_epilogue_store8<true>(p, m, n, v0, v1);
    }
}

int num_cus() {
    static const int n = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
    }();
    return n;
}

}  // namespace

// the epilogue class of the fast (EPI 3) instantiations
bool umr_nt256p_fast_epilogue(const umr_gemm_desc* d) {
    const bool vec_ok = ((d->N & 7) == 0) && ((d->ldc & 7) == 0) && ((d->ldaux & 7) == 0);
    const bool mask = (d->flags & UMR_EPI_MASK_RELU) != 0, add = (d->flags & UMR_EPI_ADD_AUX) != 0;
    return vec_ok && d->c2_mode == 0 && d->c_rows_in <= 0 && d->aux_mod <= 0 &&
           !(d->flags & (UMR_EPI_ROWBIAS | UMR_EPI_ADD_AUX2 | UMR_EPI_OUT_F32 | UMR_EPI_MASK_DGELU)) && !(mask && add) &&
           (d->act == UMR_ACT_NONE || (d->act == UMR_ACT_RELU && !mask && !add));
}
// the epilogue class that implements red_* / no_store (the reduction only without an aux operand)
bool umr_nt256p_plain_epilogue(const umr_gemm_desc* d) {
    return umr_nt256p_fast_epilogue(d) && (!d->red_w || (d->conv == 0 && !(d->flags & (UMR_EPI_ADD_AUX | UMR_EPI_MASK_RELU))));
}

// the GELU class (EPI 4): GELU (optionally saving the pre-activation) or the GELU'-masked gradient
static bool umr_nt256p_gelu_epilogue(const umr_gemm_desc* d) {
    const bool vec_ok = ((d->N & 7) == 0) && ((d->ldc & 7) == 0) && ((d->ldaux & 7) == 0) && ((d->ldc2 & 7) == 0);
    const bool dg = (d->flags & UMR_EPI_MASK_DGELU) != 0;
    return vec_ok && (d->c2_mode == 0 || (d->c2_mode == 2 && !dg)) && d->c_rows_in <= 0 && d->aux_mod <= 0 && !d->red_w && !d->no_store &&
           !(d->flags & (UMR_EPI_ROWBIAS | UMR_EPI_ADD_AUX2 | UMR_EPI_OUT_F32 | UMR_EPI_MASK_RELU | UMR_EPI_ADD_AUX)) &&
           ((d->act == UMR_ACT_GELU && !dg) || (d->act == UMR_ACT_NONE && dg));
}

// eligibility: plain NT GEMM without A-row remap, or stride-1 3x3 conv (checked by the caller, gemm_nt256.hip)
// Tile height and K-split of a plane GEMM (see the kernel), chosen together.  Cost model in plane-pair steps of a 256-row tile
// (~1.5 us): a round of work items on the chip takes (steps per item x bm / 256 + ~8 for its first loads and its store); splitting adds
// the finish launch (~8) and its slab traffic (ks x M x N x 4 bytes at ~3 TB/s).  Checked against a sweep of forced caps at the
// reference recipe's shapes (profiles/r04_x3_ksplit_sweep.txt): within 4 % of the best cap in every column.
static void x3_plan(const umr_gemm_desc* d, int cus, int npairs, int64_t ws_bytes, int* bm_out, int* ks_out) {
    const int tiles_n = (d->N + BN2 - 1) / BN2;
    const int kt = d->conv == 0 ? d->K / BK2 : 9 * (d->Cin / BK2);
    static const int forced = umr_env_int("UMR_X3_KSPLIT", 0);   // 0 = cost model; n = at most n runs (1 disables)
    const int bm_env = umr_opt_or(UMR_OPT_NT256_BM, 0);          // tests switch it inside one process (umr_set_debug_option)
    const int64_t slab = (int64_t)d->M * d->N * 4;
    int best_ks = 1, best_bm = BM2;
    double best_cost = 1e300;
    for (int bm = BM2; bm >= (d->conv == 0 ? 192 : BM2); bm -= 32) {
        if (d->conv == 0 && (bm_env == 256 || bm_env == 224 || bm_env == 192) && bm != bm_env) continue;
        const int64_t tiles = (int64_t)((d->M + bm - 1) / bm) * tiles_n;
        for (int ks = 1; ks <= 32 && ks * 2 <= kt + 1; ++ks) {
            if (d->red_w && ks > 1) break;
            if (forced > 0 && ks > forced) break;
            if (ks > 1 && ((int64_t)ks * slab > ws_bytes)) break;
            const int per = (kt + ks - 1) / ks;
            if ((kt + per - 1) / per != ks) continue;            // no empty run
            const int64_t rounds = (tiles * ks + cus - 1) / cus;
            const double cost = (double)rounds * (per * npairs * (bm / 256.0) + 8.0) + (ks > 1 ? 8.0 + (double)ks * slab / 4.5e6 : 0.0);
            // ties go to the larger tile; MORE runs have to buy 10 % (measured: the 64x64-map conv is flat within 3 % from 2 to 8
            // runs, and 3 runs -- 1.33 channel chunks each -- are 20 % slower than 2)
            if (cost < best_cost * (ks > best_ks ? 0.90 : 0.98)) { best_cost = cost; best_ks = ks; best_bm = bm; }
        }
    }
    *bm_out = best_bm;
    *ks_out = best_ks;
}

extern "C" int64_t umr_gemm_nt_x3_workspace(const umr_gemm_desc* d) {
    if (d == nullptr || d->dtype != UMR_BF16X3 || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    int bm, ks;
    x3_plan(d, num_cus(), umr_f32_mode_now() == UMR_F32_X3_FAST ? 3 : 6, (int64_t)1 << 40, &bm, &ks);
    return ks > 1 ? (int64_t)ks * d->M * d->N * 4 : 0;
}

// the number of K runs umr_launch_gemm_nt256p_ws would use for this plane problem with `ws_bytes` of slab space
int umr_x3_ksplit_of(const umr_gemm_desc* d, int64_t ws_bytes) {
    if (ws_bytes <= 0) return 1;
    int bm, ks;
    x3_plan(d, num_cus(), umr_f32_mode_now() == UMR_F32_X3_FAST ? 3 : 6, ws_bytes, &bm, &ks);
    return ks;
}

int umr_launch_gemm_nt256p_ws(const umr_gemm_desc* d, void* ws, int64_t ws_bytes, hipStream_t s);
int umr_launch_gemm_nt256p(const umr_gemm_desc* d, hipStream_t s) { return umr_launch_gemm_nt256p_ws(d, nullptr, 0, s); }

int umr_launch_gemm_nt256p_ws(const umr_gemm_desc* d, void* ws, int64_t ws_bytes, hipStream_t s) {
    const int tiles_n = (d->N + BN2 - 1) / BN2;
    const int cus = num_cus();
    // rows per tile (see the kernel): for plain GEMMs of a few rounds, the bm in {256, 224, 192} with the smallest
    // rounds x tile time (tile time ~ a fixed quarter -- epilogue, first-load latency -- plus the K loop, which scales with bm);
    // the waves that skip blocks are the non-ahead half, so the stagger has to be on.  UMR_NT256_BM forces a value.
    static const int stagger = umr_env_int("UMR_NT256_STAGGER", 1);   // A/B switch (0 = all waves in lock-step)
    const int bm_env = umr_opt_or(UMR_OPT_NT256_BM, 0);   // tests switch it inside one process (umr_set_debug_option)
    int bm = BM2;
    if (d->conv == 0 && d->dtype == UMR_BF16) {
        if (bm_env == 256 || bm_env == 224 || bm_env == 192) bm = bm_env;
        else if ((int64_t)((d->M + BM2 - 1) / BM2) * tiles_n <= 16ll * cus) {
            double best = 1e300;
            for (int c = 256; c >= 192; c -= 32) {
                const int64_t t = (int64_t)((d->M + c - 1) / c) * tiles_n;
                const double cost = (double)((t + cus - 1) / cus) * (0.25 + 0.75 * c / 256.0);
                if (cost < best - 1e-9) { best = cost; bm = c; }
            }
        }
    }
    const int npairs_x3 = umr_f32_mode_now() == UMR_F32_X3_FAST ? 3 : 6;
    int ksplit = 1;
    if (d->dtype == UMR_BF16X3) x3_plan(d, cus, npairs_x3, ws != nullptr ? ws_bytes : 0, &bm, &ksplit);
    if (d->dtype == UMR_BF16X3 && umr_opt(UMR_OPT_X3_TRACE) != UMR_OPT_UNSET)   // one line per plane launch: shape and plan (tools/probe/x3_shapes.py sums them up)
        fprintf(stderr, "x3 M=%d N=%d K=%d conv=%d Cin=%d ks=%d bm=%d flags=%u act=%d red=%d\n", d->M, d->N, d->K, d->conv, d->Cin, ksplit, bm,
                (unsigned)d->flags, d->act, d->red_w != nullptr);
    const int tiles_m = (d->M + bm - 1) / bm;
    const int64_t total = (int64_t)tiles_m * tiles_n;
    // One workgroup fits a CU (160 KiB LDS).  The grid is a small multiple of the CU count, not exactly the CU count: if
    // some CUs are busy when the kernel starts (an RCCL all-reduce of the previous gradient bucket runs beside backward),
    // a one-workgroup-per-CU launch would leave the workgroups that did not get a CU to run a second full round after the
    // others -- up to 2x the kernel time.  With several shorter workgroups per CU the dispatcher balances them itself; a
    // workgroup still walks >= 32 tiles, so the cross-tile prefetch keeps its value, and workgroups that run together on
    // one XCD still own neighbouring tiles (pw in the kernel).  UMR_NT256_WG_PER_CU overrides the factor.
    static const int wg_per_cu = umr_env_int("UMR_NT256_WG_PER_CU", 0);
    int64_t kf = wg_per_cu > 0 ? wg_per_cu : total / ((int64_t)cus * 32);
    if (kf < 1) kf = 1;
    if (kf > 8) kf = 8;
    // CU budget (umr_set_cu_budget): exactly `budget` workgroups, one per CU -- the other CUs stay free for a collective's kernels.
    // (the tile height and K-split above were planned on the device's CU count: results do not depend on the budget)
    const int budget = umr_cu_budget_now();
    const int cub = (budget > 0 && budget < cus) ? budget : cus;
    if (cub < cus) kf = 1;
    int64_t grid64 = cub * kf;
    if (ksplit > 1) grid64 = (cub + ksplit - 1) / ksplit;   // split-K: gridDim.y = ksplit rows of workgroups, about one workgroup per CU in all
    if (total < grid64) grid64 = total;
    const int grid = (int)grid64;
    dim3 g((unsigned)grid, (unsigned)ksplit), b(512);
    // fast class = bias / aux add / ReLU mask / ReLU with bf16 output and 16-B aligned strides
    const bool fast_ep = umr_nt256p_fast_epilogue(d);
    // K-tile form of a plain GEMM (see the kernel): two-phase from 8 K-tiles (X3: plane-pair steps) per output tile on.
    // UMR_NT256_PH2=0|1 forces a form (A/B, tests: umr_set_debug_option)
    const int ph_e = umr_opt(UMR_OPT_NT256_PH2);
    const int kt_steps = (d->conv == 0 ? d->K / BK2 : 9 * (d->Cin / BK2)) * (d->dtype == UMR_BF16X3 ? npairs_x3 : 1) / (ksplit > 1 ? ksplit : 1);
    const bool two = ph_e != UMR_OPT_UNSET ? (ph_e != 0) : (kt_steps >= 8);
#define L256P(CV, EP, AX, RD)                                                                                          \
    do {                                                                                                               \
        if (CV == 1 || two) {                                                                                          \
            UMR_SET_MAX_LDS_ONCE((gemm_nt256p_kernel<CV, EP, AX, RD, false, true>), LDS2P);                            \
            hipLaunchKernelGGL((gemm_nt256p_kernel<CV, EP, AX, RD, false, true>), g, b, LDS2P, s, *d, tiles_n, (int)total, stagger, bm); \
        } else {                                                                                                       \
            UMR_SET_MAX_LDS_ONCE((gemm_nt256p_kernel<0, EP, AX, RD, false, false>), LDS2P);                            \
            hipLaunchKernelGGL((gemm_nt256p_kernel<0, EP, AX, RD, false, false>), g, b, LDS2P, s, *d, tiles_n, (int)total, stagger, bm); \
        }                                                                                                              \
    } while (0)
    // EPI 3: the fast class (bias / aux add / ReLU mask / ReLU, bf16-staged; one instantiation per aux mode, plus the fused
    // row reduction); EPI 4: the GELU class (plain GEMM only); EPI 1: everything else
    if ((d->red_w || d->no_store) && !umr_nt256p_plain_epilogue(d))
        return umr_set_error(UMR_ERR_UNSUPPORTED, "gemm_nt: fused row reduction / no_store needs the fast epilogue class (no aux operand with the reduction)");
    const int auxm = (d->flags & UMR_EPI_ADD_AUX) ? 1 : (d->flags & UMR_EPI_MASK_RELU) ? 2 : 0;
#define L256PX(CV, EP)                                                                                                 \
    do {                                                                                                               \
        if (CV == 1 || two) {                                                                                          \
            UMR_SET_MAX_LDS_ONCE((gemm_nt256p_kernel<CV, EP, 0, false, true, true>), LDS2P);                           \
            hipLaunchKernelGGL((gemm_nt256p_kernel<CV, EP, 0, false, true, true>), g, b, LDS2P, s, *d, tiles_n, (int)total, stagger, bm, npairs, 0); \
        } else {                                                                                                       \
            UMR_SET_MAX_LDS_ONCE((gemm_nt256p_kernel<0, EP, 0, false, true, false>), LDS2P);                           \
            hipLaunchKernelGGL((gemm_nt256p_kernel<0, EP, 0, false, true, false>), g, b, LDS2P, s, *d, tiles_n, (int)total, stagger, bm, npairs, 0); \
        }                                                                                                              \
    } while (0)
    if (d->dtype == UMR_BF16X3) {   // eligibility checked by umr_gemm_nt (gemm_nt.hip)
        const int npairs = npairs_x3;
        if (d->red_w) {
            if (two) {
                UMR_SET_MAX_LDS_ONCE((gemm_nt256p_kernel<0, 5, 0, true, true, true>), LDS2P);
                hipLaunchKernelGGL((gemm_nt256p_kernel<0, 5, 0, true, true, true>), g, b, LDS2P, s, *d, tiles_n, (int)total, stagger, bm, npairs, 0);
            } else {
                UMR_SET_MAX_LDS_ONCE((gemm_nt256p_kernel<0, 5, 0, true, true, false>), LDS2P);
                hipLaunchKernelGGL((gemm_nt256p_kernel<0, 5, 0, true, true, false>), g, b, LDS2P, s, *d, tiles_n, (int)total, stagger, bm, npairs, 0);
            }
        } else if (ksplit > 1) {
            // work items write raw f32 sums to their slab; the finish kernel applies the caller's epilogue
            umr_gemm_desc sd = *d;
            sd.C = ws; sd.ldc = d->N; sd.C2 = nullptr; sd.c2_mode = 0; sd.bias = nullptr; sd.aux = nullptr; sd.aux2 = nullptr; sd.rowbias = nullptr;
            sd.flags = UMR_EPI_OUT_F32; sd.act = UMR_ACT_NONE; sd.c_rows_in = 0; sd.aux_mod = 0;
            const umr_gemm_desc* dd = &sd;
            const int kt_all = d->conv == 0 ? d->K / BK2 : 9 * (d->Cin / BK2);
            const int kt_per = (kt_all + ksplit - 1) / ksplit;
#define L256PXS(CV)                                                                                                    \
    do {                                                                                                               \
        if (CV == 1 || two) {                                                                                          \
            UMR_SET_MAX_LDS_ONCE((gemm_nt256p_kernel<CV, 5, 0, false, true, true>), LDS2P);                            \
            hipLaunchKernelGGL((gemm_nt256p_kernel<CV, 5, 0, false, true, true>), g, b, LDS2P, s, *dd, tiles_n, (int)total, stagger, bm, npairs, kt_per); \
        } else {                                                                                                       \
            UMR_SET_MAX_LDS_ONCE((gemm_nt256p_kernel<0, 5, 0, false, true, false>), LDS2P);                            \
            hipLaunchKernelGGL((gemm_nt256p_kernel<0, 5, 0, false, true, false>), g, b, LDS2P, s, *dd, tiles_n, (int)total, stagger, bm, npairs, kt_per); \
        }                                                                                                              \
    } while (0)
            if (d->conv == 0) L256PXS(0); else L256PXS(1);
#undef L256PXS
            UMR_LAUNCH_CHECK();
            const int64_t tasks = (int64_t)d->M * (d->N >> 3);
            int64_t fg = (tasks + 255) / 256;
            if (fg > 8192) fg = 8192;
            hipLaunchKernelGGL(x3_splitk_finish_kernel, dim3((unsigned)fg), dim3(256), 0, s, *d, (const float*)ws, ksplit, (int64_t)d->M * d->N);
        } else {
            // lean class: bias / ReLU / ReLU mask, one output (the head layers); everything else: the generic class
            const bool lean = !(d->flags & ~(UMR_EPI_BIAS | UMR_EPI_OUT_F32 | UMR_EPI_OUT_X3 | UMR_EPI_MASK_RELU | UMR_EPI_AUX_X3)) && d->c2_mode == 0 &&
                              d->c_rows_in <= 0 && d->aux_mod <= 0 && d->act != UMR_ACT_GELU;
            if (d->conv == 0) { if (lean) L256PX(0, 5); else L256PX(0, 6); }
            else { if (lean) L256PX(1, 5); else L256PX(1, 6); }
        }
        UMR_LAUNCH_CHECK();
        return UMR_OK;
    }
#undef L256PX
    if (d->conv == 0) {
        if (fast_ep) {
            if (d->red_w) L256P(0, 3, 0, true);
            else if (auxm == 0) L256P(0, 3, 0, false);
            else if (auxm == 1) L256P(0, 3, 1, false);
            else L256P(0, 3, 2, false);
        } else if (umr_nt256p_gelu_epilogue(d)) { if (d->flags & UMR_EPI_MASK_DGELU) L256P(0, 4, 2, false); else L256P(0, 4, 0, false); }
        else L256P(0, 1, 0, false);
    } else {
        if (fast_ep) {
            if (auxm == 0) L256P(1, 3, 0, false);
            else if (auxm == 1) L256P(1, 3, 1, false);
            else L256P(1, 3, 2, false);
        } else L256P(1, 1, 0, false);
    }
#undef L256P
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
