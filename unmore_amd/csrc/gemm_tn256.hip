// 256x256-tile weight-gradient GEMM ("TN") for bf16: the throughput path of umr_gemm_tn.
//   dW[N,K] (f32 slabs) = sum_m dY[m,N]^T . X[m,K]        (+ dbias[N])
// Same software pipeline as gemm_nt256.hip (4 phases per 64-row stage, one 16-MFMA 64x32 quadrant each, LDS-DMA
// groups issued two stages ahead inside the MFMA clusters, counted vmcnt, one raw barrier per phase), with the
// TN specifics of gemm_tn.hip: operands are staged as they lie in memory ([rows m][columns]) and transposed on
// the LDS->register path by ds_read_b64_tr_b16.
//
// Geometry: 8 waves as 2 (n) x 4 (k); a wave owns 128 n x 64 k = 8 x 4 MFMA tiles.  A stage is 64 rows of m.
// LDS per stage: four 16-KiB sub-tiles [64 rows][128 columns] (256-B rows, chunk XOR as gemm_tn.hip):
//     Y0 / Y1 : dY columns of n-half 0 / 1 of both wave rows   (column c' = wn*64 + n_local)
//     X0 / X1 : X  columns of k-half 0 / 1 of the four wave columns (c' = wk*32 + k_local)
// which are exactly the read sets of the quadrants Q0=(nh0,kh0) Q1=(nh0,kh1) Q2=(nh1,kh1) Q3=(nh1,kh0), so the
// staging table of gemm_nt256.hip applies verbatim with A -> Y and B -> X.
// Descriptors are per split (loop invariant), the stage is the DMA's scalar offset, per-lane voffsets are loop constants;
// the conv path (stride 1, Wo >= 64: a stage straddles at most one image-row end) masks halo lanes once per stage --
// a bit test against two scalar conditions when Wo % 64 == 0.
// Row tails of the split fall out of num_records (plain) or are masked with the halo lanes (conv).
// dbias: workgroups of k-tile 0 add up their dY fragments on the VALU (one f32 per n-tile and lane).
#include "umr_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int TSUB = 64 * 256;       // 16 KiB sub-tile
constexpr int TBUF = 4 * TSUB;       // 64 KiB per stage: Y0, X0, X1, Y1
constexpr int TLDS = 2 * TBUF;       // 128 KiB

__device__ __forceinline__ int tn_swz2(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// sub-tile order inside a buffer = staging group order: 0 = Y0, 1 = X0, 2 = X1, 3 = Y1
// WF (conv only): Wo % 64 == 0 -- a 64-row stage never runs over the end of an image row, so the halo test of a stage is
// two scalar conditions and a bit test per lane instead of a per-lane pixel walk
// X3 (dtype UMR_BF16X3): both operands are f32 values held as three bf16 planes per row -- dY [M][h(N) | m(N) | l(N)], X
// [M][h(K) | m(K) | l(K)] (conv: per pixel [h(Cin) | m(Cin) | l(Cin)]) -- and every 64-row stage is walked once per plane pair
// (h,h) (h,m) (h,l) (m,h) (m,m) (l,h), the six terms of the fp32-grade product (gemm_nt256p.hip): the stage loop is the bf16 loop
// unchanged, only the staging column offsets differ.  WF: 0 = a stage runs over at most one image-row end (Wo >= 64), 1 = over
// none (Wo % 64 == 0), 2 = any map size (per-lane pixel arithmetic; the DPT maps of 4x4 ... 32x32).
template <int CONV, bool PH2, int WF, bool X3 = false>
__global__ __launch_bounds__(512, 2) void gemm_tn256_kernel(const umr_gemm_tn_desc p, int tiles_k, int ntiles, int rows_per_split,
                                                            float* slab, float* bslab, int mapmode, int npairs_arg = 1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SZ = 2;
    const int npairs = X3 ? npairs_arg : 1;
    const int xpix = X3 ? 3 * p.Cin : p.Cin;          // conv: bf16 elements per input pixel
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware bijective remap: the workgroups that share an XCD (id % 8) take a contiguous run of
    // (split, tile) pairs, so the tiles of one split -- which all stream the same dY / X rows -- share one L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q_ = nwg >> 3, r_ = nwg & 7;
        const int base = (xcd < r_) ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
        bid = base + (bid >> 3);
    }
    const int split = bid / ntiles, tile = bid - split * ntiles;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;
    const int m_begin = split * rows_per_split;
    const int m_end = min(p.M, m_begin + rows_per_split);
    const int nst = (m_end - m_begin + 63) / 64;
    // dbias: the column sums of dY are spread over the k-tiles of a split (k-tile tk adds up the stages t with
    // t % tiles_k == tk), so no workgroup carries the whole VALU cost and becomes the straggler of its round
    const bool do_bias = (p.dbias != nullptr);

    // ---- staging roles: instruction i (0/1) of wave w covers sub-tile rows (w*2+i)*4 + lane/16, chunk position lane%16
    unsigned vo[4][2];      // per-lane voffsets (or OOB) of the four groups, relative to the stage's descriptor bases
    unsigned x_eff[2][2];   // conv: X groups after the per-stage halo mask
    int x_r[2][2], x_t[2][2];  // conv: row within the stage, packed tap offsets (tky+1) | (tkx+1) << 2
    unsigned x_bits[2][2];     // conv, WF: 1 << ky | (row 0 and kx == 0) << 3 | (row 63 and kx == 2) << 4
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (w * 2 + i) * 4 + (lane >> 4);
        const int gch = (lane & 15) ^ tn_swz2(r);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // Y_h: chunk gch -> stripe (wave row) gch>>3, 8-column chunk gch&7
            const int ncol = (mapmode & 2) ? n0 + h * 128 + gch * 8 : n0 + (gch >> 3) * 128 + h * 64 + (gch & 7) * 8;
            vo[h == 0 ? 0 : 3][i] = (ncol < p.N) ? (unsigned)(((int64_t)r * p.lddy + (ncol - n0)) * SZ) : OOB;
            // X_h: stripe (wave column) gch>>2, chunk gch&3
            const int kcol = (mapmode & 1) ? k0 + h * 128 + gch * 8 : k0 + (gch >> 2) * 64 + h * 32 + (gch & 3) * 8;
            unsigned v = OOB;
            x_r[h][i] = r;
            x_t[h][i] = 0;
            x_bits[h][i] = 0;
            if (kcol < p.K) {
                if (CONV == 0) {
                    v = (unsigned)(((int64_t)r * p.ldx + (kcol - k0)) * SZ);
                } else {
                    const int tap = kcol / p.Cin, ci = kcol - tap * p.Cin;
                    const int ky = tap / 3, kx = tap - ky * 3;  // offsets ky-1, kx-1 relative to the output pixel
                    x_t[h][i] = ky | (kx << 2);
                    x_bits[h][i] = (1u << ky) | ((r == 0 && kx == 0) ? 8u : 0u) | ((r == 63 && kx == 2) ? 16u : 0u);
                    v = (unsigned)(((int64_t)(r + ky * p.W + kx) * xpix + ci) * SZ);
                }
            }
            vo[1 + h][i] = v;
            x_eff[h][i] = v;
        }
    }

    // ---- stage cursor (groups are issued Y0,X0,X1,Y1 of stage 0 -- X3: of its first plane pair, its second, ... --, then of stage 1, ...)
    int st_tile = 0;               // 64-row stage of the cursor
    int st_pp = 0, st_par = 0;     // X3: plane pair within the stage; LDS buffer of the step being issued
    int sb = 0, soy = 0, sox = 0;  // conv: pixel position of the cursor stage's first row
    const int hw = (CONV != 0) ? p.Ho * p.Wo : 1;
    if (CONV != 0) {
        sb = m_begin / hw;
        const int rem = m_begin - sb * hw;
        soy = rem / p.Wo;
        sox = rem - soy * p.Wo;
    }
    // Descriptors are per SPLIT and never change (so they live in SGPRs); the stage is selected by the scalar offset of the
    // DMA instruction.  (Per-stage descriptor bases cost ~60 scalar instructions per stage to rebuild -- carried across the
    // loop as 128-bit values the compiler parks them in VGPRs and wraps every LDS-DMA in a readfirstlane waterfall loop.)
    // The host plan keeps rows_per_split * row bytes below 2^31.  Row tails: dY / plain X rows >= m_end lie beyond
    // num_records; conv X masks them with the halo lanes in the last stage.
    const int rows_split = m_end - m_begin;
    auto nrec = [](int64_t v) -> unsigned { return v > 0x7FFFFFFFll ? 0x7FFFFFFFu : (v < 0 ? 0u : (unsigned)v); };
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.dY + ((int64_t)m_begin * p.lddy + n0) * SZ), 0, nrec(((int64_t)rows_split * p.lddy - n0) * SZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = (CONV == 0)
        ? __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.X + ((int64_t)m_begin * p.ldx + k0) * SZ), 0,
                                            nrec(((int64_t)rows_split * p.ldx - k0) * SZ), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.X + ((int64_t)m_begin - (p.W + 1)) * xpix * SZ), 0,
                                            rows_split > 0 ? 0x7FFFFFFFu : 0u, 0x00020000);   // stride-1 'same' conv: pixel index == row index
    const unsigned y_step = (unsigned)(64 * p.lddy * SZ), x_step = (unsigned)(64 * (CONV == 0 ? p.ldx : xpix) * SZ);
    unsigned soffY = 0, soffX = 0;
    auto stage_prep = [&]() {
        // X3: planes of this step's pair (same order as gemm_nt256p.hip: equal dY planes consecutive)
        const int pa = X3 ? (((npairs == 6 ? 0x940 : 0x010) >> (2 * st_pp)) & 3) : 0;   // dY planes 0,0,0,1,1,2  |  0,0,1
        const int pb = X3 ? (((npairs == 6 ? 0x124 : 0x004) >> (2 * st_pp)) & 3) : 0;   // X  planes 0,1,2,0,1,0  |  0,1,0
        soffY = (unsigned)st_tile * y_step + (unsigned)(pa * p.N * SZ);
        soffX = (unsigned)st_tile * x_step + (unsigned)(pb * (CONV == 0 ? p.K : p.Cin) * SZ);
        const bool first_pair = !X3 || st_pp == 0;
        if (X3) { if (++st_pp == npairs) st_pp = 0; }
        if (CONV != 0 && first_pair) {     // the halo mask of a stage serves all of its plane pairs
            const int rows_left = rows_split - st_tile * 64;   // may be <= 0 for the cursor's run-ahead stages
            if (WF == 2) {
                // any map size: a 64-row stage may cover several image rows and images; addresses are linear in the pixel
                // index, only the validity of each lane's tap is position dependent
                const int lin0 = soy * p.Wo + sox;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int ky = x_t[h][i] & 3, kx = x_t[h][i] >> 2;
                        const int rem = (lin0 + x_r[h][i]) % hw;
                        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                        const bool ok = (unsigned)(oy + ky - 1) < (unsigned)p.H && (unsigned)(ox + kx - 1) < (unsigned)p.W && x_r[h][i] < rows_left;
                        x_eff[h][i] = ok ? vo[1 + h][i] : OOB;
                    }
                const int lin1 = (lin0 + 64) % hw;
                soy = lin1 / p.Wo;
                sox = lin1 - soy * p.Wo;
            } else if (WF == 1) {
                // vertical: rows above / below the image (scalar); horizontal: only pixel 0 of the first stage and pixel
                // Wo-1 of the last stage of an image row can fall outside
                const unsigned sw = (soy == 0 ? 1u : 0u) | (soy == p.Ho - 1 ? 4u : 0u) | (sox == 0 ? 8u : 0u) | (sox == p.Wo - 64 ? 16u : 0u);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i) x_eff[h][i] = (x_bits[h][i] & sw) ? OOB : vo[1 + h][i];
                if (rows_left < 64) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (x_r[h][i] >= rows_left) x_eff[h][i] = OOB;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int ky = x_t[h][i] & 3, kx = x_t[h][i] >> 2;
                        // pixel of this lane's row: the 64-row stage may run over the end of an image row (Wo >= 64: once)
                        int ox = sox + x_r[h][i], oy = soy;
                        if (ox >= p.Wo) { ox -= p.Wo; ++oy; }
                        if (oy >= p.Ho) oy = 0;   // first row of the next image (addresses are linear in the pixel index)
                        const bool ok = (unsigned)(oy + ky - 1) < (unsigned)p.H && (unsigned)(ox + kx - 1) < (unsigned)p.W && x_r[h][i] < rows_left;
                        x_eff[h][i] = ok ? vo[1 + h][i] : OOB;
                    }
            }
            if (WF != 2) {
                sox += 64;
                if (sox >= p.Wo) { sox -= p.Wo; if (++soy >= p.Ho) { soy = 0; ++sb; } }
            }
        }
    };
    auto stage_issue = [&](auto gtag, auto itag) {
        constexpr int G = decltype(gtag)::value, I = decltype(itag)::value;
        char* dst = smem + st_par * TBUF + G * TSUB + (w * 2 + I) * 1024;
        if constexpr (G == 0 || G == 3) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, UMR_LDS_PTR(dst), 16, vo[G][I], soffY, 0, 0);
        } else {
            const unsigned v = (CONV == 0) ? vo[G][I] : x_eff[G - 1][I];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, UMR_LDS_PTR(dst), 16, v, soffX, 0, 0);
        }
        if (G == 3 && I == 1) { st_par ^= 1; if (!X3 || st_pp == 0) ++st_tile; }   // (st_pp was advanced by stage_prep: 0 = the stage's last pair)
    };
#define STAGE_DMA(G, I) stage_issue(std::integral_constant<int, G>{}, std::integral_constant<int, I>{})

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = 0.f;

    const int wn = w >> 2, wk = w & 3;
    const int g = lane >> 4, li = lane & 15;
    const int q = li >> 2, pp = li & 3;
    // transposed-read addresses.  Block row r = ks*32 + g*8 + half*4 + q; swizzle key = (q<<2) | ((2g+half)&3); the
    // 16-column tile t of a wave starts at chunk cbase + 2t.  One base per (tile, half); ks adds 32 rows (+8192 B).
    int y_ad[4][2], x_ad[2][2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int r = g * 8 + half * 4 + q;
        const int sw = tn_swz2(r);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int c = wn * 8 + 2 * t + (pp >> 1);
            y_ad[t][half] = r * 256 + ((c ^ sw) << 4) + ((pp & 1) << 3);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c = wk * 4 + 2 * t + (pp >> 1);
            x_ad[t][half] = r * 256 + ((c ^ sw) << 4) + ((pp & 1) << 3);
        }
    }
    // (ks*32 rows keep the swizzle key: (32>>2)&3 == 0 and 32&3 == 0)
    // Transposed reads are issued as inline asm: through the builtin the compiler cannot tell them from the in-flight
    // LDS-DMA writes and puts `s_waitcnt vmcnt(0)` in front of every group of reads, which drains the two-stage
    // prefetch each phase.  Ordering is explicit instead: the reads of a phase are issued before PHASE_SYNC, whose
    // counted vmcnt + barrier published the data one phase earlier and whose lgkmcnt(0) precedes their first use.
    const unsigned lds0 = (unsigned)(uintptr_t)UMR_LDS_PTR(smem);
    unsigned par_off = lds0;  // LDS address of the current stage buffer
    auto read_frag = [&](auto offtag, int ad0, int ad1) -> bf16x8 {
        constexpr int OFF = decltype(offtag)::value;
        u32x2 v0, v1;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v0) : "v"(par_off + (unsigned)ad0), "n"(OFF));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v1) : "v"(par_off + (unsigned)ad1), "n"(OFF));
        const u32x4 f = {v0[0], v0[1], v1[0], v1[1]};
        return __builtin_bit_cast(bf16x8, f);
    };
#define RF(SUBI, KS, AD) read_frag(std::integral_constant<int, (SUBI) * TSUB + (KS) * 8192>{}, AD[0], AD[1])

    bf16x8 fy[2][4], fx0[2][2], fx1[2][2];  // [ks][tile]: dY of the current n-half, X(kh0), X(kh1)

#define PHASE_SYNC_N(N)                                            \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");          \
    __builtin_amdgcn_s_barrier();                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);
    // D[i = k_local][j = n_local] = sum_m X[m,k] dY[m,n]: first operand = X fragment, second = dY fragment
#define MFMA(ACC, XF, YF) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(XF, YF, ACC, 0, 0, 0)
#ifndef UMR_EXP_TN_PRIO_MODE
#define UMR_EXP_TN_PRIO_MODE 1   // conv weight gradient, same box: 33.7 / 33.2 / 33.4 ms for modes 0 / 1 / 3
#endif
    // static priority for waves 4-7 in the two-phase form, as in gemm_nt256p.hip; round 4: for the plain weight gradients as well
    // (about -1 %: 1x1 head 7.81 -> 7.73 ms, ViT-B fc1 182.6 -> 174.0 us; profiles/r04_plain_gemm_two_phase_ab.txt)
    constexpr int PRIO_MODE = PH2 ? UMR_EXP_TN_PRIO_MODE : 0;
#define QPRIO(x) if (PRIO_MODE == 0) __builtin_amdgcn_s_setprio(x);
#define QUADRANT_D(N0, K0, FX, DMA_A, DMA_B)                                                         \
    QPRIO(1)                                                                                        \
    MFMA(acc[N0 + 0][K0 + 0], FX[0][0], fy[0][0]); MFMA(acc[N0 + 0][K0 + 1], FX[0][1], fy[0][0]);   \
    MFMA(acc[N0 + 1][K0 + 0], FX[0][0], fy[0][1]); MFMA(acc[N0 + 1][K0 + 1], FX[0][1], fy[0][1]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    DMA_A;                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MFMA(acc[N0 + 2][K0 + 0], FX[0][0], fy[0][2]); MFMA(acc[N0 + 2][K0 + 1], FX[0][1], fy[0][2]);   \
    MFMA(acc[N0 + 3][K0 + 0], FX[0][0], fy[0][3]); MFMA(acc[N0 + 3][K0 + 1], FX[0][1], fy[0][3]);   \
    MFMA(acc[N0 + 0][K0 + 0], FX[1][0], fy[1][0]); MFMA(acc[N0 + 0][K0 + 1], FX[1][1], fy[1][0]);   \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    DMA_B;                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    MFMA(acc[N0 + 1][K0 + 0], FX[1][0], fy[1][1]); MFMA(acc[N0 + 1][K0 + 1], FX[1][1], fy[1][1]);   \
    MFMA(acc[N0 + 2][K0 + 0], FX[1][0], fy[1][2]); MFMA(acc[N0 + 2][K0 + 1], FX[1][1], fy[1][2]);   \
    MFMA(acc[N0 + 3][K0 + 0], FX[1][0], fy[1][3]); MFMA(acc[N0 + 3][K0 + 1], FX[1][1], fy[1][3]);   \
    QPRIO(0)
#define QUADRANT(N0, K0, FX, G) QUADRANT_D(N0, K0, FX, STAGE_DMA(G, 0), STAGE_DMA(G, 1))
#define PHASE_SYNC() PHASE_SYNC_N(6)
#define BIAS_ACC(N0)                                                                                \
    if (do_bias && bias_turn && wk == 0) {                                                          \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                             \
            float s_ = 0.f;                                                                         \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) s_ += (float)fy[ks][t][e];            \
            bsum[N0 + t] += s_;                                                                     \
        }                                                                                           \
    }

    bool bias_turn = false;
    // two-phase form (PH2; see gemm_nt256p.hip): P0 = (Q0,Q1) reads X0,Y0,X1 and issues Y1 of stage t+1; P1 = (Q2,Q3) reads
    // Y1 and issues Y0,X0,X1 of stage t+2; waits vmcnt(6) / vmcnt(2)
    auto stage_body2 = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) { fx0[0][t] = RF(1, 0, x_ad[t]); fx0[1][t] = RF(1, 1, x_ad[t]); }
#pragma unroll
        for (int t = 0; t < 4; ++t) { fy[0][t] = RF(0, 0, y_ad[t]); fy[1][t] = RF(0, 1, y_ad[t]); }
#pragma unroll
        for (int t = 0; t < 2; ++t) { fx1[0][t] = RF(2, 0, x_ad[t]); fx1[1][t] = RF(2, 1, x_ad[t]); }
        PHASE_SYNC_N(6);
        BIAS_ACC(0)
        QUADRANT_D(0, 0, fx0, STAGE_DMA(3, 0), STAGE_DMA(3, 1))
        QUADRANT_D(0, 2, fx1, (void)0, (void)0)
        stage_prep();
#pragma unroll
        for (int t = 0; t < 4; ++t) { fy[0][t] = RF(3, 0, y_ad[t]); fy[1][t] = RF(3, 1, y_ad[t]); }
        PHASE_SYNC_N(2);
        BIAS_ACC(4)
        QUADRANT_D(4, 2, fx1, STAGE_DMA(0, 0); STAGE_DMA(0, 1), STAGE_DMA(1, 0); STAGE_DMA(1, 1))
        QUADRANT_D(4, 0, fx0, STAGE_DMA(2, 0), STAGE_DMA(2, 1))
    };
    auto stage_body = [&]() {
        if (PH2) { stage_body2(); return; }
        // ---- phase 0: Q0 = (nh0, kh0)
#pragma unroll
        for (int t = 0; t < 2; ++t) { fx0[0][t] = RF(1, 0, x_ad[t]); fx0[1][t] = RF(1, 1, x_ad[t]); }
#pragma unroll
        for (int t = 0; t < 4; ++t) { fy[0][t] = RF(0, 0, y_ad[t]); fy[1][t] = RF(0, 1, y_ad[t]); }
        PHASE_SYNC();
        BIAS_ACC(0)
        QUADRANT(0, 0, fx0, 2)
        // ---- phase 1: Q1 = (nh0, kh1)
#pragma unroll
        for (int t = 0; t < 2; ++t) { fx1[0][t] = RF(2, 0, x_ad[t]); fx1[1][t] = RF(2, 1, x_ad[t]); }
        PHASE_SYNC();
        QUADRANT(0, 2, fx1, 3)
        // ---- phase 2: Q2 = (nh1, kh1)
        stage_prep();
#pragma unroll
        for (int t = 0; t < 4; ++t) { fy[0][t] = RF(3, 0, y_ad[t]); fy[1][t] = RF(3, 1, y_ad[t]); }
        PHASE_SYNC();
        BIAS_ACC(4)
        QUADRANT(4, 2, fx1, 0)
        // ---- phase 3: Q3 = (nh1, kh0)
        PHASE_SYNC();
        QUADRANT(4, 0, fx0, 1)
    };

    if (PRIO_MODE == 1 && w >= 4) __builtin_amdgcn_s_setprio(1);
    // prologue: Y0,X0,X1,Y1 of stage 0 and Y0,X0 of stage 1
    stage_prep();
    STAGE_DMA(0, 0); STAGE_DMA(0, 1); STAGE_DMA(1, 0); STAGE_DMA(1, 1);
    STAGE_DMA(2, 0); STAGE_DMA(2, 1); STAGE_DMA(3, 0); STAGE_DMA(3, 1);
    stage_prep();
    STAGE_DMA(0, 0); STAGE_DMA(0, 1); STAGE_DMA(1, 0); STAGE_DMA(1, 1);
    if (PH2) { STAGE_DMA(2, 0); STAGE_DMA(2, 1); }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int bias_next = tk;   // stages tk, tk + tiles_k, ... carry this workgroup's share of the dbias sums
    int c_pp = 0, c_m = 0;
    const int nsteps = nst * npairs;
#pragma unroll 1
    for (int t = 0; t < nsteps; ++t) {
        par_off = lds0 + (unsigned)((t & 1) * TBUF);
        // X3: a stage's column sums are taken once per dY plane -- in the pairs whose X plane is h (0, 3, 5 | 0, 2)
        const bool sum_pair = !X3 || ((((npairs == 6 ? 0x124 : 0x004) >> (2 * c_pp)) & 3) == 0);
        bias_turn = (c_m == bias_next) && sum_pair;
        stage_body();
        if (++c_pp == npairs) { c_pp = 0; if (c_m == bias_next) bias_next += tiles_k; ++c_m; }
    }
#undef BIAS_ACC
#undef QUADRANT
#undef QUADRANT_D
#undef PHASE_SYNC_N
#undef MFMA
#undef PHASE_SYNC
#undef STAGE_DMA
#undef RF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- results: lane holds n = li, k = 4*g + reg of each tile
    float* out = slab + (int64_t)split * p.N * p.K;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        const int n = (mapmode & 2) ? n0 + (nt >> 2) * 128 + wn * 64 + (nt & 3) * 16 + li : n0 + wn * 128 + nt * 16 + li;
        if (n >= p.N) continue;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k = (mapmode & 1) ? k0 + (kt >> 1) * 128 + wk * 32 + (kt & 1) * 16 + g * 4 : k0 + wk * 64 + kt * 16 + g * 4;
            if (k >= p.K) continue;
            float* o = out + (int64_t)n * p.K + k;
            if (k + 3 < p.K && (p.K & 3) == 0) *(f32x4*)o = acc[nt][kt];
            else { for (int e = 0; e < 4; ++e) if (k + e < p.K) o[e] = acc[nt][kt][e]; }
        }
        if (do_bias && wk == 0) {
            float s_ = bsum[nt];
            s_ += __shfl_xor(s_, 16, 64);
            s_ += __shfl_xor(s_, 32, 64);
            if (g == 0) bslab[((int64_t)split * tiles_k + tk) * p.N + n] = s_;
        }
    }
}

}  // namespace

// plan shared with gemm_tn.hip through these two helpers
bool umr_tn256_eligible(const umr_gemm_tn_desc* d, bool force) {
    if (d->dtype != UMR_BF16 && d->dtype != UMR_BF16X3) return false;
    if (d->dy_rows_in > 0 || d->x_rows_in > 0) return false;
    if (d->conv == 2) return false;
    if (d->dtype == UMR_BF16X3) return true;        // the only kernel for plane operands (any map size: WF 2)
    if (d->conv == 1 && d->Wo < 64) return false;   // a 64-row stage may straddle one image-row end, not two
    if (force) return true;  // structurally supported (tails are masked); the rest is a performance heuristic
    if (d->N < 192 || d->K < 192) return false;
    const int64_t tiles = (int64_t)((d->N + 255) / 256) * ((d->K + 255) / 256);
    // enough rows that one workgroup per CU gets >= 16 stages
    return (int64_t)d->M >= 256 * 64 * 16 / (tiles < 1 ? 1 : tiles) && d->M >= 64 * 64;
}

void umr_tn256_plan(const umr_gemm_tn_desc* d, int* splits, int* rows_per_split) {
    const int64_t tiles = (int64_t)((d->N + 255) / 256) * ((d->K + 255) / 256);
    static const int rounds_env = umr_env_int("UMR_TN256_ROUNDS", -1);
    // Rounds of one workgroup per CU: every split writes a 256 KiB fp32 partial per tile and the reduction reads it back,
    // so few-row problems (the transformer weight gradients: 37 k tokens) want ONE round of long splits -- 4 rounds cost
    // +50 % there (tools/vit_block_bench.py) -- while the pixel-sized ones (>= 64 stages per split even at 4 rounds) keep
    // 4 rounds for the tail balance.
    const int x3 = d->dtype == UMR_BF16X3 ? 6 : 1;   // plane pairs: steps per 64-row stage
    int64_t rounds = (int64_t)d->M * tiles * x3 / (256ll * 4096);
    if (rounds < 1) rounds = 1;
    if (rounds > 4) rounds = 4;
    if (rounds_env > 0) rounds = rounds_env;
    int64_t want = rounds * 256 / tiles;
    if (want < 1) want = 1;
    const int64_t min_stages = x3 == 6 ? 2 : 16;      // >= 16 (X3: 12) steps per split
    const int64_t max_by_rows = ((int64_t)d->M + 64 * min_stages - 1) / (64 * min_stages);
    if (want > max_by_rows) want = max_by_rows;
    int64_t rps = ((int64_t)d->M + want - 1) / want;
    rps = (rps + 63) / 64 * 64;
    {   // the kernel addresses a split with 32-bit offsets from its first row
        const int64_t cpix = (int64_t)d->Cin * (d->dtype == UMR_BF16X3 ? 3 : 1);
        const int64_t rowb = 2 * (int64_t)(d->conv ? (cpix > d->lddy ? cpix : d->lddy) : (d->ldx > d->lddy ? d->ldx : d->lddy));
        const int64_t halo = d->conv ? (2 * (int64_t)d->W + 4) * cpix * 2 : 0;
        int64_t cap = ((1ll << 31) - halo - 65536) / rowb / 64 * 64;
        if (cap < 64) cap = 64;
        if (rps > cap) rps = cap;
    }
    *rows_per_split = (int)rps;
    *splits = (int)(((int64_t)d->M + rps - 1) / rps);
}

int umr_launch_gemm_tn256(const umr_gemm_tn_desc* d, int splits, int rows_per_split, float* slab, float* bslab, hipStream_t s) {
    const int tiles_n = (d->N + 255) / 256, tiles_k = (d->K + 255) / 256;
    dim3 g((unsigned)(tiles_n * tiles_k * splits)), b(512);
    static const int mapmode = umr_env_int("UMR_TN_MAP", 3);
    static const int ph2 = umr_env_int("UMR_TN256_PH2", 1);   // two-phase stage: +1-2 % (tools/kbench.py)
#define LT(CV, WFV)                                                                                                    \
    do {                                                                                                               \
        UMR_SET_MAX_LDS_ONCE((gemm_tn256_kernel<CV, false, WFV>), TLDS); \
        UMR_SET_MAX_LDS_ONCE((gemm_tn256_kernel<CV, true, WFV>), TLDS);                                                                                                              \
        if (ph2) hipLaunchKernelGGL((gemm_tn256_kernel<CV, true, WFV>), g, b, TLDS, s, *d, tiles_k, tiles_n * tiles_k, rows_per_split, slab, bslab, mapmode, 1); \
        else hipLaunchKernelGGL((gemm_tn256_kernel<CV, false, WFV>), g, b, TLDS, s, *d, tiles_k, tiles_n * tiles_k, rows_per_split, slab, bslab, mapmode, 1);   \
    } while (0)
#define LTX(CV, WFV)                                                                                                   \
    do {                                                                                                               \
        UMR_SET_MAX_LDS_ONCE((gemm_tn256_kernel<CV, true, WFV, true>), TLDS);                                          \
        hipLaunchKernelGGL((gemm_tn256_kernel<CV, true, WFV, true>), g, b, TLDS, s, *d, tiles_k, tiles_n * tiles_k, rows_per_split, slab, bslab, mapmode, npairs); \
    } while (0)
    if (d->dtype == UMR_BF16X3) {
        const int npairs = umr_f32_mode_now() == UMR_F32_X3_FAST ? 3 : 6;
        if (d->conv == 0) LTX(0, 0);
        else if (d->Wo % 64 == 0) LTX(1, 1);
        else if (d->Wo >= 64) LTX(1, 0);
        else LTX(1, 2);
    } else if (d->conv == 0) LT(0, 0);
    else if (d->Wo % 64 == 0) LT(1, 1);
    else LT(1, 0);
#undef LTX
#undef LT
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
