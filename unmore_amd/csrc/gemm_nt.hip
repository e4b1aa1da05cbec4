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
#include <type_traits>
#include "gemm_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int UMR_SPLITK_COUNTERS = 4096;   // int32 tile counters at the head of the split-K workspace (16 KiB), slabs behind
constexpr int ROWB = 128;
constexpr int TILE_BYTES = BM * ROWB;       // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES; // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // double buffered: 64 KiB -> 2 workgroups / CU

template <typename T> struct Tr;
template <> struct Tr<bf16_t> { static constexpr int EPC = 8, BK = 64; };
template <> struct Tr<float> { static constexpr int EPC = 4, BK = 32; };


template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> {
    typedef bf16x8 type;
    static __device__ __forceinline__ type load(const bf16_t* q) { return *(const bf16x8*)q; }
    static __device__ __forceinline__ void cvt(type t, f32x4& a, f32x4& b) {
        a = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
        b = f32x4{(float)t[4], (float)t[5], (float)t[6], (float)t[7]};
    }
};
template <> struct Raw8<float> {
    struct type { f32x4 a, b; };
    static __device__ __forceinline__ type load(const float* q) { type t; t.a = *(const f32x4*)q; t.b = *(const f32x4*)(q + 4); return t; }
    static __device__ __forceinline__ void cvt(type t, f32x4& a, f32x4& b) { a = t.a; b = t.b; }
};

// three-way bf16 split of 8 f32 values: x = h + m + l with |m| <= 2^-8 |x|, |l| <= 2^-16 |x| (round-to-nearest each time)
__device__ __forceinline__ void split3_bf16(const f32x4& x0, const f32x4& x1, bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? x0[e] : x1[e - 4];
        const bf16_t hh = (bf16_t)x;
        const float r = x - (float)hh;
        const bf16_t mm = (bf16_t)r;
        const float r2 = r - (float)mm;
        h[e] = hh; m[e] = mm; l[e] = (bf16_t)r2;
    }
}

// X3 (f32 only): the products of the f32 GEMM are formed on the bf16 matrix cores from three-way bf16 splits of both operands,
//   a b ~= ah bh + ah bm + am bh + ah bl + al bh + am bm      (dropped: am bl, al bm, al bl <= 2^-23 |a b|),
// six v_mfma_f32_16x16x32_bf16 (96 matrix-pipe cycles per 16x16x32 block) instead of eight v_mfma_f32_16x16x4_f32 (256 cycles),
// accumulated in f32 as before.  Every product is within ~2.4e-7 of the f32 product (f32's own rounding is 6e-8) -- fp32-grade
// results (measured: every fp32 parity test unchanged), not the bit-exact f32 FMA chain of the plain path; +35 % on the fp32 head conv
// (48.3 -> 35.8 ms at 50 crops: the splitting VALU work, not the matrix pipe, bounds it).  Default; UMR_F32_X3=0 restores the f32 MFMA.
// (A four-deep K-tile ring for grids of at most one workgroup per CU was built and measured: no change -- 43.0 vs 43.5 us at
// 1300 x 1024 x 4096.  A lone workgroup takes 0.68 us per K-tile: not the load round trip but the CU's L2 -> LDS fill rate,
// 32 KiB per K-tile at ~29 B/clk = 0.46 us; what helps small grids is the K range on more CUs: split-K below.)
template <typename T, int CONV, int EPI, bool X3 = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const umr_gemm_desc p, int tiles_n, int splits_arg, float* __restrict__ skws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // splits_arg < 0: the split-K hand-over in its textbook form (agent-scope release / acquire fences) -- chosen at run time
    // (umr_set_debug_option("UMR_SPLITK_FENCE", "1"); the environment variable is read once, at load) or at build time (-DUMR_SPLITK_FENCE: libumr_fence.so)
#ifdef UMR_SPLITK_FENCE
    const bool fenced = true;
#else
    const bool fenced = splits_arg < 0;
#endif
    const int splits = splits_arg < 0 ? -splits_arg : splits_arg;
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
    // split-K (few tiles and a long K, plain or conv: umr_gemm_nt_ws): the `splits` workgroups of a tile are neighbours in
    // the remapped order (same XCD, except where a tile straddles two XCD runs), each takes a contiguous range of K-tiles
    int sk = 0, tile_id = bid;
    if (splits > 1) { tile_id = bid / splits; sk = bid - tile_id * splits; }
    // Tile order inside an XCD's run.  tiles_n > 0: n fastest -- the run covers a few M-rows of tiles and (nearly) every N-tile: the XCD
    // streams its own A row panels once and ALL of B.  tiles_n < 0 (= -tiles_m): m fastest -- the run covers a few N-columns of tiles and
    // every M-tile: its own B panels once and all of A.  The host picks the order that makes the operand every XCD has to fetch whole
    // the SMALLER one (umr_gemm_nt: B = the weights is the larger one whenever N > M, i.e. at the reference recipe's 1300 tokens):
    // profiles/r05_small_gemm_xcd_order.txt -- 77 MB fetched per launch for 11 MB of operands with the n-fastest order.
    int tm, tn;
    if (tiles_n < 0) { tn = tile_id / (-tiles_n); tm = tile_id - tn * (-tiles_n); }
    else { tm = tile_id / tiles_n; tn = tile_id - tm * tiles_n; }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging by buffer LDS-DMA (buffer_load_dwordx4 ... lds): address = descriptor base (tile-local,
    // scalar) + per-lane 32-bit voffset (constant for the whole K loop) + scalar soffset (the K-tile / tap
    // offset).  Out-of-range lanes make the DMA write ZEROS to LDS (measured: tools/probe/oob_lds.hip), which
    // gives conv halos, M/N tails and K tails for free: a masked lane's voffset is simply OOB.  Per K-tile the
    // staging path issues no VALU address arithmetic at all (it was ~4 VALU per MFMA with flat addresses).
    constexpr unsigned OOB = 0x80000000u;
    constexpr int SZ = (int)sizeof(T);
    const int lrow = lane >> 3, lchk = lane & 7;
    const int stride = (CONV == 2) ? 2 : 1;
    unsigned a_vo[4], a_eff[4], b_vo[4];
    int a_y[4], a_x[4], gch[4];
    const char* a_base;
    {
        int64_t origin;  // element index of the tile-local origin in A
        if (CONV == 0) {
            int ar0 = m0;
            if (p.a_rows_in > 0) ar0 = (m0 / p.a_rows_in) * p.a_rows_out + p.a_row_off + (m0 % p.a_rows_in);
            origin = (int64_t)ar0 * p.lda;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (w * 4 + i) * 8 + lrow;
                gch[i] = lchk ^ (((i & 1) << 2) + (lane >> 4));
                const int m = m0 + r;
                int ar = m;
                if (p.a_rows_in > 0) ar = (m / p.a_rows_in) * p.a_rows_out + p.a_row_off + (m % p.a_rows_in);
                a_vo[i] = (m < p.M) ? (unsigned)(((int64_t)(ar - ar0) * p.lda) * SZ + gch[i] * 16) : OOB;
                a_y[i] = a_x[i] = 0;
            }
        } else {
            const int hw = p.Ho * p.Wo;
            const int b0 = m0 / hw, rem0 = m0 - b0 * hw;
            const int oy0 = rem0 / p.Wo, ox0 = rem0 - oy0 * p.Wo;
            const int64_t pix0 = ((int64_t)b0 * p.H + oy0 * stride) * p.W + ox0 * stride;
            origin = (pix0 - (p.W + 1)) * p.Cin;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (w * 4 + i) * 8 + lrow;
                gch[i] = lchk ^ (((i & 1) << 2) + (lane >> 4));
                const int m = m0 + r;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int64_t pix = ((int64_t)b * p.H + oy * stride) * p.W + ox * stride;
                a_vo[i] = (m < p.M) ? (unsigned)((pix - pix0) * p.Cin * SZ + gch[i] * 16) : OOB;
                a_y[i] = oy * stride - 1;
                a_x[i] = ox * stride - 1;
            }
        }
        a_base = (const char*)p.A + origin * SZ;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (w * 4 + i) * 8 + lrow;
        b_vo[i] = (n0 + r < p.N) ? (unsigned)(((int64_t)r * p.ldb) * SZ + gch[i] * 16) : OOB;
        a_eff[i] = a_vo[i];
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.B + (int64_t)n0 * p.ldb * SZ), 0, 0x7FFFFFFF, 0x00020000);

    const int ktiles_per_tap = (CONV == 0) ? 0 : (p.Cin + BK - 1) / BK;
    const int nt = (CONV == 0) ? (p.K + BK - 1) / BK : 9 * ktiles_per_tap;
    const bool k_tail = (CONV == 0) ? (p.K % BK) != 0 : (p.Cin % BK) != 0;

    // sequential staging state (stage() is called for t = 0, 1, 2, ... in order)
    int st_tap = 0, st_ci = 0;
    auto stage = [&](int t, int buf) {
        char* sa = smem + buf * STAGE_BYTES + w * 4096;
        char* sb = sa + TILE_BYTES;
        unsigned soffA, soffB;
        int c0;
        if (CONV == 0) {
            c0 = t * BK;
            soffA = soffB = (unsigned)(c0 * SZ);
        } else {
            c0 = st_ci * BK;
            const int ky = st_tap / 3, kx = st_tap - ky * 3;
            {   // the tap changes every K-tile (channel-chunk-major order): rows of this lane inside the image
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = (unsigned)(a_y[i] + ky) < (unsigned)p.H && (unsigned)(a_x[i] + kx) < (unsigned)p.W;
                    a_eff[i] = ok ? a_vo[i] : OOB;
                }
            }
            soffA = (unsigned)(((ky * p.W + kx) * p.Cin + c0) * SZ);
            soffB = (unsigned)((st_tap * p.Cin + c0) * SZ);
            if (++st_tap == 9) { st_tap = 0; ++st_ci; }
        }
        const int klim = (CONV == 0) ? p.K : p.Cin;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned va = a_eff[i], vb = b_vo[i];
            if (k_tail && (c0 + gch[i] * EPC >= klim)) { va = OOB; vb = OOB; }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, UMR_LDS_PTR(sa + i * 1024), 16, va, soffA, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, UMR_LDS_PTR(sb + i * 1024), 16, vb, soffB, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wr = w >> 1, wc = w & 1;
    const int frow = lane & 15, fq = lane >> 4;
    // LDS byte addresses of this lane's fragments (buffer 0), one per (k-step, tile): loop invariant
    int a_addr[2][4], b_addr[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + frow, rb = wc * 64 + i * 16 + frow;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int q = ks * 4 + fq;
            a_addr[ks][i] = ra * ROWB + ((q ^ ((ra >> 1) & 7)) << 4);
            b_addr[ks][i] = TILE_BYTES + rb * ROWB + ((q ^ ((rb >> 1) & 7)) << 4);
        }
    }

    auto compute = [&](const char* sbuf) {
        if constexpr (sizeof(T) == 4 && X3) {
            // one K-tile = 32 f32 per row = ONE k-step of the bf16 MFMA: lane group fq, element e <-> k = (e < 4 ? 0 : 16) + 4 fq + (e & 3)
            // (any bijection works as long as both operands use the same one)
            bf16x8 ah[4], am[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                split3_bf16(*(const f32x4*)(sbuf + a_addr[0][i]), *(const f32x4*)(sbuf + a_addr[1][i]), ah[i], am[i], al[i]);
#pragma unroll
            for (int ntl = 0; ntl < 4; ++ntl) {
                bf16x8 bh, bm, bl;
                split3_bf16(*(const f32x4*)(sbuf + b_addr[0][ntl]), *(const f32x4*)(sbuf + b_addr[1][ntl]), bh, bm, bl);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    f32x4 c = acc[mt][ntl];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am[mt], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[mt], c, 0, 0, 0);
                    acc[mt][ntl] = c;
                }
            }
            return;
        }
#ifndef UMR_NT_FRAGS_PER_KSTEP   // (A/B hook: the round-4 form below, fragments read per 32-wide k-step)
        if constexpr (sizeof(T) == 2) {
            // bf16: all 16 fragment reads of the K-tile are issued before its first MFMA -- one exposed LDS latency per K-tile instead of one per
            // group of reads the compiler's schedule waited for (+32 VGPRs, still two workgroups per CU).  With two workgroups on a CU
            // -7...-9 % per launch (qkv / fc1 at 1300 tokens: 18.4 -> 16.7, 21.9 -> 20.3 us), -3 % for lone workgroups at K = 1024, nothing at
            // K = 4096 (profiles/r05_small_gemm_xcd_order.txt)
            bf16x8 af[2][4], bfr[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[ks][i] = *(const bf16x8*)(sbuf + a_addr[ks][i]);
                    bfr[ks][i] = *(const bf16x8*)(sbuf + b_addr[ks][i]);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int ntl = 0; ntl < 4; ++ntl)
                        acc[mt][ntl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][ntl], af[ks][mt], acc[mt][ntl], 0, 0, 0);
            return;
        }
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[i] = *(const bf16x8*)(sbuf + a_addr[ks][i]);
                    bfr[i] = *(const bf16x8*)(sbuf + b_addr[ks][i]);
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
                    af[i] = *(const f32x4*)(sbuf + a_addr[ks][i]);
                    bfr[i] = *(const f32x4*)(sbuf + b_addr[ks][i]);
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
    };

    int t_beg = 0, t_end = nt;
    if (splits > 1) {
        const int per = (nt + splits - 1) / splits;     // the launcher chose splits so that no range is empty
        t_beg = sk * per;
        t_end = t_beg + per < nt ? t_beg + per : nt;
        if (CONV != 0) { st_ci = t_beg / 9; st_tap = t_beg - st_ci * 9; }   // conv K order: channel chunk major, tap minor
    }
    stage(t_beg, 0);
    for (int t = t_beg; t < t_end; t += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < t_end) stage(t + 1, 1);
        compute(smem);
        if (t + 1 >= t_end) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < t_end) stage(t + 2, 0);
        compute(smem + STAGE_BYTES);
    }

    if (splits > 1) {
        // Every workgroup leaves its partial accumulators in its slab (register-major: 16 x [256 threads] x 16 B, coalesced);
        // the one that arrives last at the tile's counter adds the slabs IN SPLIT ORDER (so the result does not depend on
        // arrival order: bitwise reproducible), resets the counter for the next launch and runs the epilogue.
        //
        // Hand-over (default build).  What it rests on, by name -- cdna_hip_programming.md section 5 "Projection GEMM at M = 256",
        // item 2 ("Equally valid and cheaper per episode") and section 6 Guideline 16 R1:
        //   * slab stores are agent-scope atomic stores, i.e. `global_store ... sc1`: WRITE-THROUGH to the memory side shared by
        //     all eight XCD L2s, and their vmcnt credit returns only when that write is acknowledged -- after
        //     `s_waitcnt vmcnt(0)` this wave's partial sums are visible to any agent-scope (sc1) load, whichever XCD issues it;
        //   * the barrier makes that true for every wave of the workgroup before lane 0 draws its ticket; the ticket is an
        //     agent-scope RMW (performed at the same memory side), relaxed: it orders nothing by itself, the vmcnt waits do;
        //   * the reducer reads the slabs with agent-scope atomic loads, i.e. `global_load ... sc1`, EVERY one of them: an sc1
        //     load is served from the memory side, never from a stale line of the reducer's own XCD L2.
        // So no cache write-back / invalidate is needed: the agent-scope FENCE of the textbook protocol is correct too, but on
        // this chip it writes back and invalidates the XCD's whole L2 -- twice per workgroup: measured 27.5 -> 31.7 ms on the
        // reference recipe's step, slower than not splitting at all.  This is hardware behaviour documented in the guide, not
        // something the HIP memory model promises for relaxed atomics; -DUMR_SPLITK_FENCE builds the textbook release / acquire
        // form (libumr_fence.so), and tests/test_gemm_gpu.py::test_split_k_equals_the_fence_build requires bit-identical results.
        // The workgroup-scope fences around the ticket cost nothing (no cache action) and keep the COMPILER from moving memory
        // operations across it.
        int* counters = (int*)skws;
        unsigned long long* slabs = (unsigned long long*)(skws + UMR_SPLITK_COUNTERS);
        constexpr int64_t SLAB = (int64_t)BM * BN / 2;      // in 8-byte units
        unsigned long long* mine = slabs + ((int64_t)tile_id * splits + sk) * SLAB + tid * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 v = acc[r >> 2][r & 3];
            __hip_atomic_store(mine + r * 512, __builtin_bit_cast(unsigned long long, f32x2{v[0], v[1]}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + r * 512 + 1, __builtin_bit_cast(unsigned long long, f32x2{v[2], v[3]}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // every thread's slab stores are acknowledged; the K loop's LDS reads are done
        if (tid == 0) {
            if (fenced) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            }
            const int old = __hip_atomic_fetch_add(counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // a counter that was not zero on first use / was left by an aborted launch.  No trap (it would kill the context of every
            // stream of the process; include/umr.h: no entry point aborts): the event is counted in the workspace's error word
            // (umr_gemm_nt_ws_status), the counter is healed for the next launch, and this workgroup stands down -- the tile's
            // output is NOT written and must not be trusted until the status call returns zero bad tickets
            const bool bad = old < 0 || old >= splits;
            if (bad) {
                __hip_atomic_fetch_add(counters + (UMR_SPLITK_COUNTERS - 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (fenced) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            const int last = !bad && old == splits - 1;
            if (last) __hip_atomic_store(counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *(int*)smem = last;
        }
        __syncthreads();
        if (*(const int*)smem == 0) return;
        const unsigned long long* base = slabs + (int64_t)tile_id * splits * SLAB + tid * 2;
        auto ld4 = [&](const unsigned long long* q) {
            const f32x2 lo = __builtin_bit_cast(f32x2, __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const f32x2 hi = __builtin_bit_cast(f32x2, __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            return f32x4{lo[0], lo[1], hi[0], hi[1]};
        };
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r >> 2][r & 3] = ld4(base + r * 512);
        for (int q = 1; q < splits; ++q) {      // 32 loads in flight per split, added in split order
            const unsigned long long* bq = base + (int64_t)q * SLAB;
            f32x4 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = ld4(bq + r * 512);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r >> 2][r & 3] += v[r];
        }
    }

    // ---- epilogue.  Accumulators (lane: row (lane&15), 4 consecutive columns) are staged through LDS in two
    // passes of 64 tile rows ([64][132] f32, padded against ds_write_b128 conflicts), then every thread handles
    // (row, 8 consecutive columns) tasks: bias / aux / C accesses are 16-byte vectors and each 128-column tile row
    // is written as one contiguous 256-byte (bf16) run -- full HBM lines instead of 32-byte fragments.
    // Two instantiations keep the instruction footprint small (the fully inlined generic epilogue is ~50 KiB and
    // runs out of the I-cache):
    //   EPI 0  bias / aux add / ReLU mask / aux2 add / ReLU / C2 copies with 16-B aligned strides: unrolled, every
    //          global LOAD issued before the first store (vmcnt retires in order -- a load behind a store waits for
    //          the store's acknowledgement);
    //   EPI 1  everything else (GELU / dGELU / tanh, row bias, f32 output, row remaps, odd strides): one copy of
    //          the generic store code in runtime loops.
    constexpr int EP_LD = 132;
    float* stg = (float*)smem;
    auto stage_pass = [&](auto ptag) {
        constexpr int PASS = decltype(ptag)::value;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int lr = wr * 32 + mh * 16 + frow;  // staging row 0..63
#pragma unroll
            for (int ntl = 0; ntl < 4; ++ntl)
                *(f32x4*)(stg + lr * EP_LD + wc * 64 + ntl * 16 + fq * 4) = acc[PASS * 2 + mh][ntl];
        }
    };
    if constexpr (EPI == 0) {
        const int c8 = (tid & 15) * 8, n = n0 + c8;
        const bool n_ok = n < p.N;
        const bool use_aux = (p.flags & (UMR_EPI_ADD_AUX | UMR_EPI_MASK_RELU | UMR_EPI_MASK_DGELU)) != 0;
        const bool use_aux2 = (p.flags & UMR_EPI_ADD_AUX2) != 0;
        f32x4 bias0 = {0.f, 0.f, 0.f, 0.f}, bias1 = {0.f, 0.f, 0.f, 0.f};
        if ((p.flags & UMR_EPI_BIAS) && n_ok) { bias0 = *(const f32x4*)(p.bias + n); bias1 = *(const f32x4*)(p.bias + n + 4); }
        typename Raw8<T>::type ra[8], rb2[8];
        auto row_of = [&](int pass, int k) { const int lr = (tid >> 4) + k * 16; return m0 + (lr >> 5) * 64 + pass * 32 + (lr & 31); };
        auto prefetch = [&](auto ptag) {
            constexpr int PASS = decltype(ptag)::value;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = row_of(PASS, k);
                if (m < p.M && n_ok) {
                    if (use_aux) ra[PASS * 4 + k] = Raw8<T>::load((const T*)p.aux + (int64_t)m * p.ldaux + n);
                    if (use_aux2) rb2[PASS * 4 + k] = Raw8<T>::load((const T*)p.aux2 + (int64_t)m * p.ldaux2 + n);
                }
            }
        };
        auto pass = [&](auto ptag) {
            constexpr int PASS = decltype(ptag)::value;
            __syncthreads();
            stage_pass(ptag);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int lr = (tid >> 4) + k * 16;
                const int m = row_of(PASS, k);
                if (m >= p.M || !n_ok) continue;
                f32x4 v0 = *(const f32x4*)(stg + lr * EP_LD + c8), v1 = *(const f32x4*)(stg + lr * EP_LD + c8 + 4);
                v0 += bias0; v1 += bias1;
                if (use_aux) {
                    f32x4 a0, a1;
                    Raw8<T>::cvt(ra[PASS * 4 + k], a0, a1);
                    if (p.flags & UMR_EPI_ADD_AUX) { v0 += a0; v1 += a1; }
                    else if (p.flags & UMR_EPI_MASK_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = a0[e] > 0.f ? v0[e] : 0.f; v1[e] = a1[e] > 0.f ? v1[e] : 0.f; }
                    } else if constexpr (sizeof(T) == 2) {   // GELU'-mask (16-bit mode only: the cheap erf form)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] *= dgelu_sel<T>(a0[e]); v1[e] *= dgelu_sel<T>(a1[e]); }
                    }
                }
                if (use_aux2) {
                    f32x4 a0, a1;
                    Raw8<T>::cvt(rb2[PASS * 4 + k], a0, a1);
                    v0 += a0; v1 += a1;
                }
                if (p.c2_mode == 2) Vec8<T>::store((T*)p.C2 + (int64_t)m * p.ldc2 + n, v0, v1);
                if (p.act == UMR_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                } else if (p.act == UMR_ACT_GELU) {
                    if constexpr (sizeof(T) == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = gelu_sel<T>(v0[e]); v1[e] = gelu_sel<T>(v1[e]); }
                    }
                }
                Vec8<T>::store((T*)p.C + (int64_t)m * p.ldc + n, v0, v1);
                if (p.c2_mode == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                    Vec8<T>::store((T*)p.C2 + (int64_t)m * p.ldc2 + n, v0, v1);
                }
            }
        };
        prefetch(std::integral_constant<int, 0>{});
        if constexpr (sizeof(T) == 2) prefetch(std::integral_constant<int, 1>{});   // 16-bit: both passes fit in registers
        pass(std::integral_constant<int, 0>{});
        if constexpr (sizeof(T) != 2) prefetch(std::integral_constant<int, 1>{});
        pass(std::integral_constant<int, 1>{});
    } else {
        const bool vec_ok = ((p.N & 7) == 0) && ((p.ldc & 7) == 0) && ((p.ldc2 & 7) == 0) && ((p.ldaux & 7) == 0) &&
                            ((p.ldaux2 & 7) == 0);
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            if (pass == 0) stage_pass(std::integral_constant<int, 0>{});
            else stage_pass(std::integral_constant<int, 1>{});
            __syncthreads();
#pragma unroll 1
            for (int task = tid; task < 64 * 16; task += 256) {
                const int lr = task >> 4, c8 = (task & 15) * 8;
                // staging row -> tile row: rows [wr*32, wr*32+32) of this pass are tile rows wr*64 + pass*32 + ..
                const int trow = (lr >> 5) * 64 + pass * 32 + (lr & 31);
                const int m = m0 + trow, n = n0 + c8;
                if (m >= p.M || n >= p.N) continue;
                const f32x4 v0 = *(const f32x4*)(stg + lr * EP_LD + c8), v1 = *(const f32x4*)(stg + lr * EP_LD + c8 + 4);
                if (vec_ok) {
                    epilogue_store8<T>(p, m, n, v0, v1);
                } else {
                    epilogue_store<T>(p, m, n, v0);
                    if (n + 4 < p.N) epilogue_store<T>(p, m, n + 4, v1);
                }
            }
        }
    }
}

}  // namespace

int umr_launch_gemm_nt256(const umr_gemm_desc* d, hipStream_t s);  // gemm_nt256.hip
int umr_launch_gemm_nt256p(const umr_gemm_desc* d, hipStream_t s);  // gemm_nt256p.hip
int umr_launch_gemm_nt256p_ws(const umr_gemm_desc* d, void* ws, int64_t ws_bytes, hipStream_t s);
bool umr_nt256_rowreduce_path(const umr_gemm_desc* d);

static int tile_override() {  // UMR_GEMM_TILE=128|256 forces a tile size (benchmarking, tests); umr_set_debug_option switches it at run time
    return umr_opt_or(UMR_OPT_GEMM_TILE, 0);
}

static bool uses_256(const umr_gemm_desc* d) {
    if (d->dtype != UMR_BF16) return false;
    const int64_t t256 = (int64_t)((d->M + 255) / 256) * ((d->N + 255) / 256);
    const int ov = tile_override();
    const bool kfit = d->conv == 0 ? (d->K % 64 == 0) : (d->Cin % 64 == 0);  // the 256 kernel has no K-tail path
    // >= 8 full rounds of one 256x256 tile per CU (no tail, overheads amortised); plain GEMMs already win from 1.5 rounds on
    // when K >= 512 or there are >= 4 rounds (the transformer's GEMMs at 37 k tokens: tools/vit_block_bench.py)
    // UMR_NT256_MIN_TILES: smallest plain-GEMM problem (in 256x256 tiles) given to the 256x256 kernel
    static const int min_tiles = umr_env_int("UMR_NT256_MIN_TILES", 300);   // 384 -> 300: +2 % on the ViT-L/14 step (proj / fc2 at 344 tiles), neutral at cfg2
    // smallest 3x3-conv problem on the 256x256 kernel: its K loop is 9 x Cin long, so two to three rounds of tiles already beat
    // the 128x128 kernel (cfg1's head convs, 784 tiles: 542 -> 577 images/s; cfg2 / cfg4 / ref / the sweep unchanged)
    static const int conv_min_tiles = umr_env_int("UMR_NT256_CONV_MIN_TILES", 512);
    const bool big = kfit && d->N >= 192 &&
                     (t256 >= 2048 || (d->conv == 1 && t256 >= conv_min_tiles) || (d->conv == 0 && d->a_rows_in <= 0 && t256 >= min_tiles && (d->K >= 512 || t256 >= 1024)));
    return kfit && (ov == 256 || (ov == 0 && big));
}

extern "C" int umr_gemm_nt_rowreduce_ok(const umr_gemm_desc* d) {
    if (d == nullptr || d->M <= 0 || d->N <= 0 || d->K <= 0) return 0;
    return (uses_256(d) && umr_nt256_rowreduce_path(d)) ? 1 : 0;
}

static int gemm_nt_impl(const umr_gemm_desc* d, void* workspace, int64_t workspace_bytes, umr_stream_t stream);

extern "C" int umr_gemm_nt(const umr_gemm_desc* d, umr_stream_t stream) { return gemm_nt_impl(d, nullptr, 0, stream); }

// The one diagnostic entry point that synchronises (include/umr.h): how many split-K tickets were out of range since the word was
// last read (each one = an output tile that was not written), and the word is cleared.  The ticket counters of a launch use at most
// the first 256 of the UMR_SPLITK_COUNTERS ints (pick_splits: tiles x splits <= 512, splits >= 2); the LAST int is this error word.
extern "C" int umr_gemm_nt_ws_status(void* workspace, umr_stream_t stream, int* bad_tickets) {
    UMR_CHECK_ARG(workspace != nullptr && bad_tickets != nullptr, "gemm_nt_ws_status: null argument");
    hipStream_t s = (hipStream_t)stream;
    int* word = (int*)workspace + (UMR_SPLITK_COUNTERS - 1);
    int v = 0;
    hipError_t e_ = hipMemcpyAsync(&v, word, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e_ == hipSuccess) e_ = hipStreamSynchronize(s);
    if (e_ == hipSuccess && v != 0) e_ = hipMemsetAsync(word, 0, sizeof(int), s);
    if (e_ != hipSuccess) return umr_set_error(UMR_ERR_HIP - (int)e_, hipGetErrorString(e_));
    *bad_tickets = v;
    return UMR_OK;
}

extern "C" int64_t umr_gemm_nt_workspace(void) { return (int64_t)UMR_SPLITK_COUNTERS * 4 + (int64_t)512 * BM * BN * 4; }   // 16 KiB + 32 MiB

static int pick_splits(const umr_gemm_desc* d, int64_t tiles, bool have_ws);
static bool uses_256(const umr_gemm_desc* d);
int umr_x3_ksplit_of(const umr_gemm_desc* d, int64_t ws_bytes);   // gemm_nt256p.hip

extern "C" int umr_gemm_nt_splits(const umr_gemm_desc* d, int64_t workspace_bytes) {
    if (d == nullptr || d->M <= 0 || d->N <= 0 || d->K <= 0 || workspace_bytes <= 0) return 1;
    if (d->dtype == UMR_BF16X3) return umr_x3_ksplit_of(d, workspace_bytes - (int64_t)UMR_SPLITK_COUNTERS * 4);
    if (d->dtype != UMR_BF16 && d->dtype != UMR_F32) return 1;
    if (uses_256(d)) return 1;
    return pick_splits(d, (int64_t)((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN), true);
}

extern "C" int umr_gemm_nt_ws(const umr_gemm_desc* d, void* workspace, int64_t workspace_bytes, umr_stream_t stream) {
    UMR_CHECK_ARG(workspace == nullptr || (workspace_bytes >= umr_gemm_nt_workspace() && ((uintptr_t)workspace & 15) == 0),
                  "gemm_nt_ws: workspace smaller than umr_gemm_nt_workspace() or not 16-byte aligned");
    return gemm_nt_impl(d, workspace, workspace_bytes, stream);
}

// Split-K for the 128x128 kernel: plain GEMMs whose tiles fill less than ~a third of the chip's 2 x CUs workgroup slots and
// whose K loop is long (the transformer's projections at a few thousand tokens: 88 tiles x 64 K-tiles on 256 CUs).  Every
// range gets >= 6 K-tiles; all tiles x splits workgroups are co-resident.  UMR_NT_SPLITK=0 disables, =n forces n (tests).
static int pick_splits(const umr_gemm_desc* d, int64_t tiles, bool have_ws) {
    if (!have_ws || tiles > UMR_SPLITK_COUNTERS) return 1;
    const int bk = d->dtype == UMR_BF16 ? 64 : 32;
    const int nt = d->conv == 0 ? (d->K + bk - 1) / bk : 9 * ((d->Cin + bk - 1) / bk);   // K-tiles, as the kernel counts them
    int want;
    const int forced = umr_opt(UMR_OPT_NT_SPLITK);
    if (forced != UMR_OPT_UNSET) {
        want = forced;
    } else {
        // measured (tools/probe/small_gemm.py, 1300 tokens): bf16 pays ~10 us for the slab round trip -- two ranges from
        // K = 2048 on (43 -> 33 us at K = 4096), never more; the f32 forms do 6-8x the matrix work per K-tile -- up to five
        // ranges of >= 16 K-tiles (235 -> 92 us at K = 4096)
        if (tiles > 170) return 1;
        // a handful of tiles (3x3 convs on 7^2 ... 28^2 maps): the slab traffic is negligible, only the ~10 us hand-over counts --
        // bf16 29 -> 20 us with four ranges, f32 118 -> 34 us with eight
        if (d->dtype == UMR_BF16) {
            want = tiles <= 32 ? (nt / 8 < 4 ? nt / 8 : 4) : (nt >= 32 ? 2 : 1);
        } else {
            want = (int)(448 / tiles);
            const int per_min = tiles <= 32 ? 8 : 16;
            if (want > nt / per_min) want = nt / per_min;
            if (want > 8) want = 8;
        }
    }
    if (want > nt) want = nt;
    if (want < 2 || tiles * want > 512) return 1;
    const int per = (nt + want - 1) / want;
    return (nt + per - 1) / per;      // no empty range
}

template <typename T, int CV, int EPI, bool X3>
static void launch_nt(dim3 g, hipStream_t s, const umr_gemm_desc* d, int tiles_n, int splits, float* skws) {
    // UMR_SPLITK_FENCE=1 (umr_set_debug_option): the textbook agent-scope release / acquire hand-over without a rebuild
    const int splits_arg = (splits > 1 && umr_opt_or(UMR_OPT_SPLITK_FENCE, 0) != 0) ? -splits : splits;
    hipLaunchKernelGGL((gemm_nt_kernel<T, CV, EPI, X3>), g, dim3(256), LDS_BYTES, s, *d, tiles_n, splits_arg, skws);
}

static int gemm_nt_impl(const umr_gemm_desc* d, void* workspace, int64_t workspace_bytes, umr_stream_t stream) {
    UMR_CHECK_ARG(d != nullptr, "gemm_nt: null descriptor");
    UMR_CHECK_ARG(d->A && d->B && (d->C || d->no_store), "gemm_nt: null operand");
    UMR_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, "gemm_nt: empty problem");
    if (d->dtype == UMR_BF16X3) {
        // f32 values as three bf16 planes: persistent 256x256 kernel only (include/umr.h)
        UMR_CHECK_ARG(d->conv == 0 || d->conv == 1, "gemm_nt (BF16X3): plain GEMM or stride-1 3x3 conv only");
        const int fl = d->flags;
        const bool out_f32 = (fl & UMR_EPI_OUT_F32) != 0, out_x3 = (fl & UMR_EPI_OUT_X3) != 0;
        const int kk = d->conv == 0 ? d->K : d->Cin;
        const bool red = d->red_w != nullptr;   // fused row reduction: inference form only (C is not stored), plain GEMM
        const int n_aux = ((fl & UMR_EPI_ADD_AUX) != 0) + ((fl & UMR_EPI_MASK_RELU) != 0) + ((fl & UMR_EPI_MASK_DGELU) != 0);
        auto ld_ok = [&](int64_t ld, bool planes) { return planes ? (ld % 8 == 0 && ld >= 3 * (int64_t)d->N) : (ld % 4 == 0 && ld >= d->N); };
        bool ok = (kk % 64 == 0) && (d->N % 8 == 0) && d->a_rows_in <= 0 && (d->ldb % 8 == 0) && (d->ldb >= 3 * (int64_t)d->K) &&
                  (d->act == UMR_ACT_NONE || d->act == UMR_ACT_RELU || d->act == UMR_ACT_GELU) && n_aux <= 1 &&
                  (!(fl & UMR_EPI_MASK_DGELU) || (d->act == UMR_ACT_NONE && d->c2_mode == 0 && !(fl & UMR_EPI_ADD_AUX2))) &&
                  (d->conv == 1 ? (d->K == 9 * d->Cin && (int64_t)d->nb * d->Ho * d->Wo == d->M && d->Ho == d->H && d->Wo == d->W && d->c_rows_in <= 0)
                                : (d->lda % 8 == 0 && d->lda >= 3 * (int64_t)d->K));
        if (red) {
            ok = ok && d->no_store && !out_f32 && !out_x3 && d->conv == 0 && d->red_out && (d->red_c == 1 || d->red_c == 2) && d->c2_mode == 0 &&
                 !(fl & ~UMR_EPI_BIAS) && d->act != UMR_ACT_GELU && d->c_rows_in <= 0 && d->aux_mod <= 0;
        } else {
            ok = ok && !d->no_store && d->C && (out_f32 != out_x3) && ld_ok(d->ldc, out_x3) &&
                 (d->c2_mode == 0 || ((d->c2_mode == 1 || d->c2_mode == 2) && d->C2 && ld_ok(d->ldc2, (fl & UMR_EPI_C2_X3) != 0))) &&
                 (n_aux == 0 || (d->aux && ld_ok(d->ldaux, (fl & UMR_EPI_AUX_X3) != 0))) &&
                 (!(fl & UMR_EPI_ADD_AUX2) || (d->aux2 && ld_ok(d->ldaux2, (fl & UMR_EPI_AUX2_X3) != 0))) &&
                 (!(fl & UMR_EPI_ROWBIAS) || (d->rowbias && d->rows_per_batch > 0)) && (d->aux_mod <= 0 || n_aux == 1);
        }
        if (!ok) return umr_set_error(UMR_ERR_UNSUPPORTED, "gemm_nt (BF16X3): needs K (conv: Cin) % 64 == 0, N % 8 == 0, no A-row remap, act in {none, ReLU, GELU}, "
                                                           "exactly one of OUT_F32 / OUT_X3 (or red_w with no_store and a bias / ReLU epilogue), strides that "
                                                           "fit the operand formats (include/umr.h)");
        UMR_CHECK_ARG(!(fl & UMR_EPI_BIAS) || d->bias, "gemm_nt: bias flag without pointer");
        UMR_CHECK_ARG((int64_t)((d->M + 191) / 192) * ((d->N + 255) / 256) * 32 < (1ll << 31), "gemm_nt: grid too large");
        // the split-K slabs lie behind the 16 KiB of tile counters of the 128x128 kernel's split-K
        char* xws = workspace ? (char*)workspace + (int64_t)UMR_SPLITK_COUNTERS * 4 : nullptr;
        const int64_t xws_bytes = workspace ? workspace_bytes - (int64_t)UMR_SPLITK_COUNTERS * 4 : 0;
        return umr_launch_gemm_nt256p_ws(d, xws, xws_bytes, (hipStream_t)stream);
    }
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
    int tiles_n_arg = tiles_n;
    const int64_t grid = (int64_t)tiles_m * tiles_n;
    UMR_CHECK_ARG(grid < (1ll << 31), "gemm_nt: grid too large");
    hipStream_t s = (hipStream_t)stream;
    if (d->red_w || d->no_store) {
        UMR_CHECK_ARG(umr_gemm_nt_rowreduce_ok(d) == 1, "gemm_nt: fused row reduction / no_store requested on a path that does not implement it (umr_gemm_nt_rowreduce_ok)");
        UMR_CHECK_ARG(!d->red_w || (d->red_out && (d->red_c == 1 || d->red_c == 2)), "gemm_nt: red_out / red_c");
    }
    if (uses_256(d)) return umr_launch_gemm_nt256(d, s);
    const int splits = pick_splits(d, grid, workspace != nullptr);
    float* skws = (float*)workspace;
    // tile order per XCD (see the kernel): m fastest when the B operand is the larger one.  UMR_NT_ORDER=n|m forces an order (A/B; umr_set_debug_option)
    {
        const int oe = umr_opt(UMR_OPT_NT_ORDER);
        const bool mfast = oe != UMR_OPT_UNSET ? (oe == 'm') : (d->conv == 0 && d->a_rows_in <= 0 && (int64_t)d->N > (int64_t)d->M);
        if (mfast) tiles_n_arg = -tiles_m;
    }
    dim3 g((unsigned)(grid * splits)), b(256);
    const bool fast_ep = ((d->N & 7) == 0) && ((d->ldc & 7) == 0) && (d->c2_mode == 0 || (d->ldc2 & 7) == 0) &&
                         (!(d->flags & (UMR_EPI_ADD_AUX | UMR_EPI_MASK_RELU | UMR_EPI_MASK_DGELU)) || (d->ldaux & 7) == 0) &&
                         (!(d->flags & UMR_EPI_ADD_AUX2) || (d->ldaux2 & 7) == 0) && d->c_rows_in <= 0 && d->aux_mod <= 0 &&
                         !(d->flags & (UMR_EPI_ROWBIAS | UMR_EPI_OUT_F32)) &&
                         (!(d->flags & UMR_EPI_MASK_DGELU) || d->dtype == UMR_BF16) &&   // GELU forms: 16-bit mode only (cheap erf)
                         (d->act == UMR_ACT_NONE || d->act == UMR_ACT_RELU || (d->act == UMR_ACT_GELU && d->dtype == UMR_BF16));
#define LAUNCH(T, CV)                                                                   \
    do {                                                                                \
        if (fast_ep) launch_nt<T, CV, 0, false>(g, s, d, tiles_n_arg, splits, skws);  \
        else launch_nt<T, CV, 1, false>(g, s, d, tiles_n_arg, splits, skws);          \
    } while (0)
#define LAUNCH_X3(CV)                                                                      \
    do {                                                                                   \
        if (fast_ep) launch_nt<float, CV, 0, true>(g, s, d, tiles_n_arg, splits, skws);  \
        else launch_nt<float, CV, 1, true>(g, s, d, tiles_n_arg, splits, skws);          \
    } while (0)
    const int f32_x3 = umr_f32_mode_now() != UMR_F32_EXACT;   // include/umr.h: umr_set_f32_mode (default X3; UMR_F32_X3=0 selects the f32 MFMA)
    if (d->dtype == UMR_BF16) {
        if (d->conv == 0) LAUNCH(bf16_t, 0); else if (d->conv == 1) LAUNCH(bf16_t, 1); else LAUNCH(bf16_t, 2);
    } else if (f32_x3 && !((d->conv == 0 ? d->K : d->Cin) % 32)) {   // whole 32-wide K-tiles only (the tail path zero-fills, which is fine, but keep it simple)
        if (d->conv == 0) LAUNCH_X3(0); else if (d->conv == 1) LAUNCH_X3(1); else LAUNCH_X3(2);
    } else {
        if (d->conv == 0) LAUNCH(float, 0); else if (d->conv == 1) LAUNCH(float, 1); else LAUNCH(float, 2);
    }
#undef LAUNCH_X3
#undef LAUNCH
    {
        const hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) {
            // a launch that did not happen leaves the tile counters as they were (zero); one that was enqueued and then failed may
            // not: put them back so that the NEXT launch on this workspace does not pick a wrong last arriver
            if (splits > 1) (void)hipMemsetAsync(workspace, 0, (size_t)UMR_SPLITK_COUNTERS * 4, s);
            return umr_set_error(UMR_ERR_HIP - (int)e_, hipGetErrorString(e_));
        }
    }
    return UMR_OK;
}
