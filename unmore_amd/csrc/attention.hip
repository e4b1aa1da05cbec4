// Fused scaled-dot-product attention for the ViT blocks (timm Attention: softmax(q k^T
// * hd^-0.5) v; models/dpt/vit.py:196-197 via timm Block), head dim 64, forward and backward,
// flash-style (scores never reach HBM), MFMA 16x16 (bf16 x32 / exact f32 x4).
//
// Layout: qkv is the qkv-GEMM output [B*N, 3*D] (q | k | v, heads contiguous inside each),
// the output is [B*N, D]; gradients come back in the same [B*N, 3*D] layout.
//
// Forward / dQ kernels: one wave = 16 query rows, workgroup = 64 query rows; K/V tiles of
// 32 keys staged in LDS.  Scores are computed TRANSPOSED (S^T = K Q^T) so that a lane owns
// one query column: the online-softmax row reduction is 7 in-register max/adds plus two
// shuffles, and the f32 score accumulator is already the B operand of the next MFMA
// (O^T += V^T P^T) -- no LDS round trip for P.  V^T / K^T fragments come from the row-major
// LDS tile by ds_read_b64_tr_b16 (bf16) or per-lane ds_read_b32 (f32).
// dK/dV kernel: one wave = 16 keys (K, V fragments live in registers), loops over 32-query
// tiles of Q and dO in LDS; S = Q K^T has the query on the accumulator rows, so P and dS
// feed dV^T += dO^T P and dK^T += Q^T dS directly.  dQ and dK/dV are separate kernels
// (no atomics: bitwise reproducible).
// hd^-0.5 = 1/8 is folded into Q (K in the dK/dV kernel): exact in bf16 and f32.
#include "umr_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int HD = 64;
constexpr int KT = 32;  // keys (or queries) per LDS tile

template <typename T> struct AT;
template <> struct AT<bf16_t> { static constexpr int ROWB = 128, NF = 2; typedef bf16x8 frag_t; };
template <> struct AT<float> { static constexpr int ROWB = 256, NF = 4; typedef f32x4 frag_t; };

template <typename T> __device__ __forceinline__ int kswz(int r) { return sizeof(T) == 2 ? ((r >> 1) & 7) : (r & 15); }

template <typename T> struct RowFrag { typename AT<T>::frag_t v[AT<T>::NF]; };

// fragment of one row (64 head-dim values) for lane group g: chunks (4*i + g)
template <typename T>
__device__ __forceinline__ RowFrag<T> rowfrag_global(const T* row, int g, float scale) {
    RowFrag<T> f;
#pragma unroll
    for (int i = 0; i < AT<T>::NF; ++i) {
        if constexpr (sizeof(T) == 2) {
            bf16x8 t;
            for (int e = 0; e < 8; ++e) t[e] = (bf16_t)0.f;
            if (row) t = *(const bf16x8*)(row + (4 * i + g) * 8);
            for (int e = 0; e < 8; ++e) f.v[i][e] = (bf16_t)((float)t[e] * scale);
        } else {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (row) t = *(const f32x4*)(row + (4 * i + g) * 4);
            f.v[i] = t * scale;
        }
    }
    return f;
}

template <typename T>
__device__ __forceinline__ RowFrag<T> rowfrag_lds(const char* tile, int r, int g) {
    RowFrag<T> f;
    const int sw = kswz<T>(r);
#pragma unroll
    for (int i = 0; i < AT<T>::NF; ++i)
        f.v[i] = *(const typename AT<T>::frag_t*)(tile + r * AT<T>::ROWB + (((4 * i + g) ^ sw) << 4));
    return f;
}

// acc(16x16) += A(rows) . B(rows)^T over the 64 head-dim values; D[i = A row][j = B row]
template <typename T>
__device__ __forceinline__ f32x4 mma_rows(f32x4 acc, const RowFrag<T>& a, const RowFrag<T>& b) {
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[i], b.v[i], acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[i][e], b.v[i][e], acc, 0, 0, 0);
    }
    return acc;
}

// acc[dt](16 d x 16 cols) += X^T[d][r] . P[r][col], r over the 32 tile rows.  p0/p1 are the
// accumulators of the two 16-row sub-tiles (lane group g holds rows 4g..4g+3 of each).
template <typename T>
__device__ __forceinline__ void mma_trans(f32x4 (&acc)[4], const char* tile, f32x4 p0, f32x4 p1, int lane) {
    const int g = lane >> 4, li = lane & 15;
    if constexpr (sizeof(T) == 2) {
        bf16x8 pb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pb[e] = (bf16_t)p0[e]; pb[4 + e] = (bf16_t)p1[e]; }
        const int q = li >> 2, pp = li & 3;
        const int r0 = 4 * g + q, r1 = 16 + 4 * g + q;
        const int s0 = kswz<T>(r0), s1 = kswz<T>(r1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int c = dt * 2 + (pp >> 1);
            const char* a0 = tile + r0 * 128 + ((c ^ s0) << 4) + ((pp & 1) << 3);
            const char* a1 = tile + r1 * 128 + ((c ^ s1) << 4) + ((pp & 1) << 3);
            bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a0);
            bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a1);
            bf16x8 va;
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = v0[e]; va[4 + e] = v1[e]; }
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, acc[dt], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int r = 16 * sub + 4 * g + s;
                const int sw = kswz<T>(r);
                const float pv = sub == 0 ? p0[s] : p1[s];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int d = dt * 16 + li;
                    const float a = *(const float*)(tile + r * 256 + (((d >> 2) ^ sw) << 4) + ((d & 3) << 2));
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pv, acc[dt], 0, 0, 0);
                }
            }
    }
}

// stage a [32 rows][64] tile (rows r0.. of one head) into LDS, swizzled; rows >= nrows are zero
template <typename T>
__device__ __forceinline__ void load_tile(char* tile, const T* base, int64_t ld, int r0, int nrows, int tid) {
    constexpr int CPR = AT<T>::ROWB / 16;
    constexpr int EPC = 16 / sizeof(T);
    for (int c = tid; c < KT * CPR; c += 256) {
        const int r = c / CPR, ch = c - r * CPR;
        uint4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < nrows) v = *(const uint4*)(base + (int64_t)(r0 + r) * ld + ch * EPC);
        *(uint4*)(tile + r * AT<T>::ROWB + ((ch ^ kswz<T>(r)) << 4)) = v;
    }
}

template <typename T> __device__ __forceinline__ float fexp(float x) { return sizeof(T) == 2 ? __expf(x) : expf(x); }

// XCD-aware block map.  The dispatcher deals consecutive workgroups round-robin to the eight XCDs (id % 8), each with its own L2.
// The gx workgroups of one (batch, head) all read that head's K and V (forward, dQ) or Q and dO (dK / dV): as blockIdx.x of a 2-D
// grid they were consecutive ids, i.e. on gx DIFFERENT XCDs, and every XCD fetched the head's rows again from beyond its L2 --
// 0.62 GB per forward launch at the cfg2 shape for a 0.17-GB qkv tensor, 5.9 TB/s for 0.105 ms: the kernels ran at the fabric's
// rate, not the matrix pipe's (profiles/r05_attention_xcd_map_ab.txt).  Launched 1-D; id -> (XCD, slot) -> a virtual id such
// that an XCD owns a contiguous run of virtual ids (bijective for any grid size, as in gemm_nt.hip): the workgroups of a head are
// neighbours on one XCD, dispatched within a few slots of each other.  gx_arg < 0: the old order (A/B).
__device__ __forceinline__ void attn_block_of(int bid, int nwg, int gx_arg, int& bx, int& by) {
    int gx = gx_arg;
    if (gx_arg > 0) {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        bid = base + (bid >> 3);
    } else {
        gx = -gx_arg;
    }
    by = bid / gx;
    bx = bid - by * gx;
}
__device__ __forceinline__ void attn_block(int gx_arg, int& bx, int& by) { attn_block_of((int)blockIdx.x, (int)gridDim.x, gx_arg, bx, by); }

// ------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ lse,
                                                       int N, int heads, int gx_arg) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KT * AT<T>::ROWB];
    char* sK = smem;
    char* sV = smem + KT * AT<T>::ROWB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int bx_, bh;
    attn_block(gx_arg, bx_, bh);   // XCD-aware: the workgroups of one (batch, head) share an XCD's L2
    const int b = bh / heads, h = bh - b * heads;
    const int D = heads * HD;
    const int64_t ld = 3 * (int64_t)D;
    const T* qb = qkv + (int64_t)b * N * ld + h * HD;
    const T* kb = qb + D;
    const T* vb = qb + 2 * D;
    const int q = bx_ * 64 + w * 16 + li;
    const RowFrag<T> qf = rowfrag_global<T>(q < N ? qb + (int64_t)q * ld : nullptr, g, 0.125f);

    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < N; k0 += KT) {
        __syncthreads();
        load_tile<T>(sK, kb, ld, k0, N, tid);
        load_tile<T>(sV, vb, ld, k0, N, tid);
        __syncthreads();
        f32x4 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const RowFrag<T> kf = rowfrag_lds<T>(sK, 16 * sub + li, g);
            s[sub] = mma_rows<T>(f32x4{0.f, 0.f, 0.f, 0.f}, kf, qf);  // D[i = key][j = query]
        }
        float mx = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (k0 + 16 * sub + 4 * g + e >= N) s[sub][e] = -INFINITY;
                mx = fmaxf(mx, s[sub][e]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fexp<T>(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[sub][e] = fexp<T>(s[sub][e] - m_new); ps += s[sub][e]; }
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] *= alpha;
        mma_trans<T>(o, sV, s[0], s[1], lane);  // O^T[d][query] += V^T[d][key] P^T[key][query]
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (q < N) {
        const float inv = 1.0f / l_run;
        T* orow = out + ((int64_t)b * N + q) * D + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) Vec4<T>::store(orow + dt * 16 + 4 * g, o[dt] * inv);
        if (g == 0 && lse) lse[(int64_t)bh * N + q] = m_run + logf(l_run);
    }
}

// ------------------------------------------------------------------ forward, bf16 throughput form
// Same mapping as above (one wave = 16 query columns of S^T, P^T is directly the B operand of O^T += V^T P^T), rebuilt
// around the fact that at head dim 64 the kernel is bound by vector-instruction issue, not by the matrix pipe: the loop
// above spends ~160 VALU / LDS instructions per 8 MFMAs.  Here, per 64-key tile:
//   * K/V arrive by LDS-DMA into a two-deep ring (no VGPR round trip, no address VALU in the loop; rows >= N read as
//     zeros through the buffer descriptor), one barrier per tile;
//   * exp(s - m) is ONE v_fma + ONE v_exp (log2 e folded into the fma, in f32 -- Q keeps its exact 1/8 prescale);
//   * the row sum l rides on the matrix pipe (an all-ones A fragment: one extra MFMA per 32 keys instead of 8 adds +
//     cross-lane sums, and it sums the bf16-rounded P the numerator uses);
//   * the running max is only raised when a tile exceeds it by more than 2^8 in the exponent (deferred rescale): after the
//     first tile the O rescale -- 16 accumulator registers through the VALU -- is almost never executed;
//   * key masking only in the last tile.
// (built with -fno-honor-nans: fmaxf on MFMA results otherwise costs a canonicalising v_max per operand.  Not inline asm:
// the compiler does not see an asm statement's reads when it pads the MFMA -> VALU read hazard.)
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float max2f(float a, float b) { return fmaxf(a, b); }
// max over lanes l, l^16 (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second) and l, l^32.
// ROUND 6: the second result of the swap goes through an empty asm before the max.  hipcc (ROCm 7.2, clang 22) folds
// fmax(extract 0, extract 1) of llvm.amdgcn.permlane{16,32}.swap to extract 0 at -O3 (tools/probe: the max instruction is simply absent from
// the ISA, with or without -fno-honor-nans), so the "maximum over the four lane groups of a query column" used to be lane group 0's own
// maximum.  Softmax does not care which reference is subtracted -- every test passed -- until the true maximum sits more than 128 (log2
// units) above that group's 16 scores: exp2 overflows, the row sum becomes +inf and the output row NaN.  Found in a 2000-step soak of the
// reference recipe (step 921 / 1145 / 2913 by trajectory; tests/golden/attn_wide_score_range_n65.npz is that operand).
__device__ __forceinline__ float xor16_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("" : "+v"(r1));
    return max2f(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
}
__device__ __forceinline__ float xor32_max(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("" : "+v"(r1));
    return max2f(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
}

template <int OFF>
__device__ __forceinline__ u32x2 ds_tr_b64(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int QB>   // 16-query blocks per wave: a workgroup covers 64 * QB queries; K / V^T fragments are read once per QB MFMAs
#ifndef UMR_ATTN_FWD_MIN_WAVES
#define UMR_ATTN_FWD_MIN_WAVES 1   // experiment hook (tools/probe/build_exp_lib.sh attention.hip -DUMR_ATTN_FWD_MIN_WAVES=4)
#endif
#ifndef UMR_ATTN_DKV_MIN_WAVES
#define UMR_ATTN_DKV_MIN_WAVES 1
#endif
__global__ __launch_bounds__(256, UMR_ATTN_FWD_MIN_WAVES) void attn_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int N, int heads, int gx_arg) {
    constexpr int TK = 64;                 // keys per tile
    constexpr int OPB = TK * 128;          // one operand tile: 8 KiB
    constexpr int STB = 2 * OPB;           // K | V
#ifdef UMR_ATTN_PIPE
    constexpr int RING = 3;                // experiment: QK^T of tile j+1 is issued before the softmax of tile j (needs K[j+1] beside V[j])
#else
    constexpr int RING = 2;
#endif
    __shared__ __attribute__((aligned(16))) char smem[RING * STB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    int bx_, bh;
    attn_block(gx_arg, bx_, bh);   // XCD-aware: the workgroups of one (batch, head) share an XCD's L2
    const int b = bh / heads, h = bh - b * heads;
    const int D = heads * HD;
    const int64_t ld = 3 * (int64_t)D;
    const bf16_t* qb = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* kb = qb + D;
    const bf16_t* vb = qb + 2 * D;
    const int q0 = bx_ * (64 * QB) + w * (16 * QB) + li;   // query of block qi: q0 + 16 qi
    RowFrag<bf16_t> qf[QB];
    // Q carries hd^-0.5 AND log2(e): the scores come off the matrix pipe in log2 units, and the running maximum is subtracted by
    // STARTING the score accumulators from -m (the C operand of the first QK^T MFMA) -- exp2 applies to the accumulator as it is.
    // One fused multiply-add per score (16 per query block and tile, of ~52 vector instructions) disappears from a loop whose
    // issue port is the bound (profiles/r03_attention_sq_counters.txt: 5.6 VALU per MFMA, 97 % busy).  The price is a second bf16
    // rounding of Q (q * 0.18034 instead of the exact q / 8): 2^-9 relative on the scores, the size of Q's own bf16 rounding.
    constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) qf[qi] = rowfrag_global<bf16_t>(q0 + 16 * qi < N ? qb + (int64_t)(q0 + 16 * qi) * ld : nullptr, g, 0.125f * LOG2E);

    // ---- staging: wave w, instruction i fills LDS rows (w*2+i)*8 .. +8 of a tile (1 KiB); lane -> (row, slot), and
    // the swizzle is applied on the global side (slot s of row r holds chunk s ^ kswz(r))
    const unsigned rec = (unsigned)(((int64_t)(N - 1) * ld + HD) * 2);   // rows >= N are out of range -> zeros
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, rec, 0x00020000);
    unsigned voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (w * 2 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ kswz<bf16_t>(r);
        voff[i] = (unsigned)(r * (int)ld * 2 + c * 16);
    }
    const unsigned tile_stride = (unsigned)(TK * (int)ld * 2);
    auto issue_tile = [&](int j) {
        char* dst = smem + (RING == 2 ? (j & 1) : (j % RING)) * STB + w * 2048;
        const unsigned so = (unsigned)j * tile_stride;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, UMR_LDS_PTR(dst), 16, voff[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, UMR_LDS_PTR(dst + 1024), 16, voff[1], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, UMR_LDS_PTR(dst + OPB), 16, voff[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, UMR_LDS_PTR(dst + OPB + 1024), 16, voff[1], so, 0, 0);
    };

    // ---- read side, lane constants.  K fragment of key row 16*sub + li: chunks (4 i + g) ^ kswz; kswz(16 sub + li + 32 half)
    // == kswz(li), so sub / half / stage are immediate offsets.  V^T by transposing reads: rows 4 g + (li >> 2) (+16),
    // chunk pair dt, same argument for the swizzle.
    const int swk = kswz<bf16_t>(li);
    const char* kad0 = smem + li * 128 + (((0 + g) ^ swk) << 4);
    const char* kad1 = smem + li * 128 + (((4 + g) ^ swk) << 4);
    const int vr = 4 * g + (li >> 2), pp = li & 3;
    const int swv = kswz<bf16_t>(vr) ^ (pp >> 1);
    unsigned vad[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        vad[dt] = (unsigned)(uintptr_t)UMR_LDS_PTR(smem) + OPB + vr * 128 + (((dt * 2) ^ swv) << 4) + ((pp & 1) << 3);

    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

    f32x4 o[QB][4], lacc[QB];
    float m_run[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        lacc[qi] = f32x4{0.f, 0.f, 0.f, 0.f};
        m_run[qi] = 0.f;         // log2 units; the first tile always sets it (see below)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[qi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr float DEFER = 8.0f;           // log2 units: raise the running max only when a tile tops it by 2^8

    const int ntiles = (N + TK - 1) / TK;
#ifdef UMR_ATTN_PIPE
    // ---- experiment: in-wave pipeline.  Iteration j: [tile j+1 landed; barrier] issue tile j+2; S(j+1) = Q K[j+1]^T on the matrix pipe WHILE the
    // vector unit runs the softmax of S(j); then O += V[j] P(j).  The wave's chain per tile is max(QK^T, softmax) + PV instead of their sum.
    auto qk_tile = [&](int j, f32x4 (&s)[QB][4]) {
        const int sb = (j % RING) * STB;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const bf16x8 k0 = *(const bf16x8*)(kad0 + sb + sub * 2048);
            const bf16x8 k1 = *(const bf16x8*)(kad1 + sb + sub * 2048);
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                s[qi][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qi].v[0], f32x4{-m_run[qi], -m_run[qi], -m_run[qi], -m_run[qi]}, 0, 0, 0);
                s[qi][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qi].v[1], s[qi][sub], 0, 0, 0);
            }
        }
        if (j == ntiles - 1) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j * TK + 16 * sub + 4 * g + e >= N) {
#pragma unroll
                        for (int qi = 0; qi < QB; ++qi) s[qi][sub][e] = -INFINITY;
                    }
        }
    };
    f32x4 s[QB][4], sn[QB][4];
    issue_tile(0);
    if (ntiles > 1) issue_tile(1);
    if (ntiles > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    qk_tile(0, s);
    for (int j = 0; j < ntiles; ++j) {
        const bool more = j + 1 < ntiles;
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // tile j+1 landed for everyone; everyone is done with tile j-1's buffer
            if (j + 2 < ntiles) issue_tile(j + 2);
            qk_tile(j + 1, sn);                // independent of everything below until the end of the iteration
        }
        const int sb = (j % RING) * STB;
        bool any_up = false;
        float mxs[QB];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            float mx = max3f(s[qi][0][0], s[qi][0][1], s[qi][0][2]);
            mx = max3f(mx, s[qi][0][3], s[qi][1][0]);
            mx = max3f(mx, s[qi][1][1], s[qi][1][2]);
            mx = max3f(mx, s[qi][1][3], s[qi][2][0]);
            mx = max3f(mx, s[qi][2][1], s[qi][2][2]);
            mx = max3f(mx, s[qi][2][3], s[qi][3][0]);
            mx = max3f(mx, s[qi][3][1], s[qi][3][2]);
            mx = max2f(mx, s[qi][3][3]);
            mx = xor16_max(mx);
            mx = xor32_max(mx);
            mxs[qi] = mx;
            any_up = any_up || (mx > DEFER);
        }
        if (j == 0 || __builtin_amdgcn_ballot_w64(any_up) != 0ull) {
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                const float d = (j == 0 || mxs[qi] > DEFER) ? mxs[qi] : 0.f;
                const float alpha = j == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
                m_run[qi] += d;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[qi][i] *= alpha;
                lacc[qi] *= alpha;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) { s[qi][sub] -= d; if (more) sn[qi][sub] -= d; }   // S(j+1) was started from the old reference
            }
        }
        const unsigned vb0 = (unsigned)sb;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x2 t0[4], t1[4];
            if (half == 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { t0[dt] = ds_tr_b64<0>(vad[dt] + vb0); t1[dt] = ds_tr_b64<2048>(vad[dt] + vb0); }
            } else {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { t0[dt] = ds_tr_b64<4096>(vad[dt] + vb0); t1[dt] = ds_tr_b64<4096 + 2048>(vad[dt] + vb0); }
            }
            bf16x8 pb[QB];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pb[qi][e] = (bf16_t)__builtin_amdgcn_exp2f(s[qi][2 * half][e]);
                    pb[qi][4 + e] = (bf16_t)__builtin_amdgcn_exp2f(s[qi][2 * half + 1][e]);
                }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3])
                         :: "memory");
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x4 f = {t0[dt][0], t0[dt][1], t1[dt][0], t1[dt][1]};
#pragma unroll
                for (int qi = 0; qi < QB; ++qi)
                    o[qi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f), pb[qi], o[qi][dt], 0, 0, 0);
            }
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) lacc[qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[qi], lacc[qi], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) s[qi][sub] = sn[qi][sub];
        }
    }
#else
    issue_tile(0);
#ifdef UMR_ATTN_BARE
    // ceiling probe (tools/probe/attn_ceiling.py; never in the product build): both ring buffers are filled ONCE, the tile loop below
    // then issues no global or LDS-DMA traffic at all -- what is left is the loop's own arithmetic on LDS-resident operands (QK^T MFMAs,
    // the max3 chain and lane swaps, exp2, bf16 conversion, V^T transposing reads, PV and row-sum MFMAs, one barrier per tile)
    if (ntiles > 1) issue_tile(1);
#endif
#ifdef UMR_ATTN_PEEL_LAST
    // experiment hook (profiles/r05_attention_experiments.txt): the tile body twice, the key mask only in the copy that runs the last tile
    auto tile_body = [&](int j, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
#else
    for (int j = 0; j < ntiles; ++j) {
        const bool LAST = j == ntiles - 1;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // tile j landed for everyone; everyone is done reading buffer (j+1)&1
#ifndef UMR_ATTN_BARE
        if (!LAST) issue_tile(j + 1);
#endif
        const int sb = (j & 1) * STB;
        // S^T tiles: 4 x (16 keys x 16 queries) per query block
        f32x4 s[QB][4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const bf16x8 k0 = *(const bf16x8*)(kad0 + sb + sub * 2048);
            const bf16x8 k1 = *(const bf16x8*)(kad1 + sb + sub * 2048);
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                // scores relative to the running maximum: s = q.k * log2(e) / 8 - m_run
                s[qi][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qi].v[0], f32x4{-m_run[qi], -m_run[qi], -m_run[qi], -m_run[qi]}, 0, 0, 0);
                s[qi][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qi].v[1], s[qi][sub], 0, 0, 0);
            }
        }
        if (LAST) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j * TK + 16 * sub + 4 * g + e >= N) {
#pragma unroll
                        for (int qi = 0; qi < QB; ++qi) s[qi][sub][e] = -INFINITY;
                    }
        }
        bool any_up = false;
        float mxs[QB];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            // 16 scores -> 8 v_max3, then the four lane groups of a query column by two row swaps instead of two
            // ds_bpermute round trips
            float mx = max3f(s[qi][0][0], s[qi][0][1], s[qi][0][2]);
            mx = max3f(mx, s[qi][0][3], s[qi][1][0]);
            mx = max3f(mx, s[qi][1][1], s[qi][1][2]);
            mx = max3f(mx, s[qi][1][3], s[qi][2][0]);
            mx = max3f(mx, s[qi][2][1], s[qi][2][2]);
            mx = max3f(mx, s[qi][2][3], s[qi][3][0]);
            mx = max3f(mx, s[qi][3][1], s[qi][3][2]);
            mx = max2f(mx, s[qi][3][3]);
            mx = xor16_max(mx);
            mx = xor32_max(mx);
            mxs[qi] = mx;                                    // the tile's maximum RELATIVE to the running one
            any_up = any_up || (mx > DEFER);
        }
        if (j == 0 || __builtin_amdgcn_ballot_w64(any_up) != 0ull) {
            // the rare path (always the first tile: its maximum becomes the reference, whatever its sign): move the reference by d,
            // rescale what was accumulated against the old one, and shift this tile's scores -- after which the common path below
            // exponentiates the accumulators as they are
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                const float d = (j == 0 || mxs[qi] > DEFER) ? mxs[qi] : 0.f;
                const float alpha = j == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);   // (nothing accumulated yet on the first tile)
                m_run[qi] += d;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[qi][i] *= alpha;
                lacc[qi] *= alpha;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) s[qi][sub] -= d;
            }
        }
        const unsigned vb0 = (unsigned)sb;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x2 t0[4], t1[4];
            if (half == 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { t0[dt] = ds_tr_b64<0>(vad[dt] + vb0); t1[dt] = ds_tr_b64<2048>(vad[dt] + vb0); }
            } else {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { t0[dt] = ds_tr_b64<4096>(vad[dt] + vb0); t1[dt] = ds_tr_b64<4096 + 2048>(vad[dt] + vb0); }
            }
            bf16x8 pb[QB];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pb[qi][e] = (bf16_t)__builtin_amdgcn_exp2f(s[qi][2 * half][e]);
                    pb[qi][4 + e] = (bf16_t)__builtin_amdgcn_exp2f(s[qi][2 * half + 1][e]);
                }
            // the wait names the registers the transposing reads fill, so that no consumer can be scheduled above it
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3])
                         :: "memory");
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x4 f = {t0[dt][0], t0[dt][1], t1[dt][0], t1[dt][1]};
#pragma unroll
                for (int qi = 0; qi < QB; ++qi)
                    o[qi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f), pb[qi], o[qi][dt], 0, 0, 0);
            }
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) lacc[qi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[qi], lacc[qi], 0, 0, 0);
        }
#ifdef UMR_ATTN_PEEL_LAST
    };
    for (int j = 0; j < ntiles - 1; ++j) tile_body(j, std::false_type{});
    tile_body(ntiles - 1, std::true_type{});
#else
    }
#endif
#endif   // UMR_ATTN_PIPE
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        const int q = q0 + 16 * qi;
        if (q < N) {
            const float l_run = lacc[qi][0];
            const float inv = 1.0f / l_run;
            bf16_t* orow = out + ((int64_t)b * N + q) * D + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Vec4<bf16_t>::store(orow + dt * 16 + 4 * g, o[qi][dt] * inv);
            if (g == 0 && lse) lse[(int64_t)bh * N + q] = m_run[qi] * 0.69314718055994531f + logf(l_run);   // m_run is in log2 units
        }
    }
}

// ------------------------------------------------------------------ backward prep: Dq = rowsum(dO * O)
template <typename T>
__global__ void attn_bwd_prep_kernel(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ dq_sum, int B, int N,
                                     int heads) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, n, h)
    const int64_t total = (int64_t)B * N * heads;
    if (idx >= total) return;
    const int h = (int)(idx % heads);
    const int64_t bn = idx / heads;
    const int b = (int)(bn / N), n = (int)(bn - (int64_t)b * N);
    const T* po = o + bn * heads * HD + h * HD;
    const T* pd = dout + bn * heads * HD + h * HD;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
        const f32x4 a = Vec4<T>::load(po + c * 4), d = Vec4<T>::load(pd + c * 4);
        s += a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
    }
    dq_sum[((int64_t)b * heads + h) * N + n] = s;
}

// ------------------------------------------------------------------ backward, bf16 throughput form
// Same recipe as the bf16 forward: 64-row tiles by LDS-DMA into a two-deep ring, one barrier per tile, one fma + one exp per
// recomputed probability (the prep kernel stores -lse * log2 e), fragments read once per two 16-row blocks of the wave.
// No masking anywhere: rows >= N of K / V / Q / dO arrive as zeros (buffer bounds), so out-of-range keys contribute
// K^T dS = 0 to dQ and out-of-range queries contribute Q^T dS = 0, dO^T P = 0 to dK, dV whatever finite P they get.
// Workspace (f32): per (batch, head) [2][Npad], Npad = N rounded up to 64: row 0 = -lse * log2 e, row 1 = rowsum(dO * O);
// the padding (zeros) makes the per-tile f32x4 loads of the dK/dV kernel unconditional.
__global__ void attn_bwd_prep_bf16_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                          float* __restrict__ ws, int B, int N, int Npad, int heads) {
    // 8 lanes per (b, n, h) row of 64 values, one 16-byte chunk each: consecutive lanes read consecutive memory
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = idx >> 3;                       // (b, n_pad, h), h fastest
    const int c = (int)(idx & 7);
    const int64_t total = (int64_t)B * Npad * heads;
    if (row >= total) return;
    const int h = (int)(row % heads);
    const int64_t bn = row / heads;
    const int b = (int)(bn / Npad), n = (int)(bn - (int64_t)b * Npad);
    float ds = 0.f;
    if (n < N) {
        const int64_t off = (((int64_t)b * N + n) * heads + h) * HD + c * 8;
        const bf16x8 a = *(const bf16x8*)(o + off), d = *(const bf16x8*)(dout + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) ds += (float)a[e] * (float)d[e];
    }
    ds += __shfl_xor(ds, 1, 64);
    ds += __shfl_xor(ds, 2, 64);
    ds += __shfl_xor(ds, 4, 64);
    if (c == 0) {
        const int64_t bh = (int64_t)b * heads + h;
        ws[(bh * 2) * Npad + n] = n < N ? -lse[bh * N + n] * 1.4426950408889634f : 0.f;
        ws[(bh * 2 + 1) * Npad + n] = ds;
    }
}

// OWN: the workgroup takes its rows' statistics (-lse * log2 e and rowsum(dO * O)) from `lse`, `dout` and `out` itself instead of
// from the prep kernel's workspace -- the form of the single-launch backward of short sequences (attn_bwd_small_bf16_kernel)
template <int QB, bool OWN>
__device__ __forceinline__ void attn_bwd_dq_bf16_body(char* smem, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                      const bf16_t* __restrict__ out, const float* __restrict__ lse,
                                                      const float* __restrict__ ws, bf16_t* __restrict__ dqkv, int N, int Npad, int heads,
                                                      int bx_, int bh) {
    constexpr int TK = 64, OPB = TK * 128, STB = 2 * OPB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = bh / heads, h = bh - b * heads;
    const int D = heads * HD;
    const int64_t ld = 3 * (int64_t)D;
    const bf16_t* qb = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* kb = qb + D;
    const bf16_t* vb = qb + 2 * D;
    const float* nl2_b = OWN ? nullptr : ws + (int64_t)(bh * 2) * Npad;
    const float* dsum_b = OWN ? nullptr : nl2_b + Npad;
    const int q0 = bx_ * (64 * QB) + w * (16 * QB) + li;
    RowFrag<bf16_t> qf[QB], dof[QB];
    float nl2q[QB], d_q[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        const int q = q0 + 16 * qi;
        const bool qok = q < N;
        qf[qi] = rowfrag_global<bf16_t>(qok ? qb + (int64_t)q * ld : nullptr, g, 0.125f * 1.4426950408889634f);   // as in the forward kernel
        dof[qi] = rowfrag_global<bf16_t>(qok ? dout + ((int64_t)b * N + q) * D + h * HD : nullptr, g, 1.0f);
        if (OWN) {
            // the row's 64 products dO * O: 16 in this lane, the rest in the three other lane groups of the row
            const RowFrag<bf16_t> of = rowfrag_global<bf16_t>(qok ? out + ((int64_t)b * N + q) * D + h * HD : nullptr, g, 1.0f);
            float ds = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) ds += (float)dof[qi].v[i][e] * (float)of.v[i][e];
            ds += __shfl_xor(ds, 16, 64);
            ds += __shfl_xor(ds, 32, 64);
            nl2q[qi] = qok ? -lse[(int64_t)bh * N + q] * 1.4426950408889634f : 0.f;
            d_q[qi] = ds;
        } else {
            nl2q[qi] = qok ? nl2_b[q] : 0.f;
            d_q[qi] = qok ? dsum_b[q] : 0.f;
        }
    }
    const unsigned rec = (unsigned)(((int64_t)(N - 1) * ld + HD) * 2);
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, rec, 0x00020000);
    unsigned voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (w * 2 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ kswz<bf16_t>(r);
        voff[i] = (unsigned)(r * (int)ld * 2 + c * 16);
    }
    const unsigned tile_stride = (unsigned)(TK * (int)ld * 2);
    auto issue_tile = [&](int j) {
        char* dst = smem + (j & 1) * STB + w * 2048;
        const unsigned so = (unsigned)j * tile_stride;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, UMR_LDS_PTR(dst), 16, voff[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, UMR_LDS_PTR(dst + 1024), 16, voff[1], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, UMR_LDS_PTR(dst + OPB), 16, voff[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, UMR_LDS_PTR(dst + OPB + 1024), 16, voff[1], so, 0, 0);
    };
    const int swk = kswz<bf16_t>(li);
    const char* kad0 = smem + li * 128 + (((0 + g) ^ swk) << 4);
    const char* kad1 = smem + li * 128 + (((4 + g) ^ swk) << 4);
    const int vr = 4 * g + (li >> 2), pp = li & 3;
    const int swv = kswz<bf16_t>(vr) ^ (pp >> 1);
    unsigned tad[4];   // K^T by transposing reads of the K tile
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        tad[dt] = (unsigned)(uintptr_t)UMR_LDS_PTR(smem) + vr * 128 + (((dt * 2) ^ swv) << 4) + ((pp & 1) << 3);

    f32x4 acc[QB][4];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[qi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr float LOG2E = 1.4426950408889634f;
    const int ntiles = (N + TK - 1) / TK;
    issue_tile(0);
    for (int j = 0; j < ntiles; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (j + 1 < ntiles) issue_tile(j + 1);
        const int sb = (j & 1) * STB;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 s[QB][2], dp[QB][2];
#pragma unroll
            for (int su = 0; su < 2; ++su) {
                const int off = sb + (2 * half + su) * 2048;
                const bf16x8 k0 = *(const bf16x8*)(kad0 + off), k1 = *(const bf16x8*)(kad1 + off);
                const bf16x8 v0 = *(const bf16x8*)(kad0 + off + OPB), v1 = *(const bf16x8*)(kad1 + off + OPB);
#pragma unroll
                for (int qi = 0; qi < QB; ++qi) {
                    // log2-unit scores starting from -lse * log2 e: the accumulator IS log2 of the probability
                    s[qi][su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qi].v[0], f32x4{nl2q[qi], nl2q[qi], nl2q[qi], nl2q[qi]}, 0, 0, 0);
                    s[qi][su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qi].v[1], s[qi][su], 0, 0, 0);      // S^T[key][query]
                    dp[qi][su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, dof[qi].v[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    dp[qi][su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, dof[qi].v[1], dp[qi][su], 0, 0, 0);   // dP^T[key][query]
                }
            }
            u32x2 t0[4], t1[4];
            const unsigned tb = (unsigned)sb;
            if (half == 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { t0[dt] = ds_tr_b64<0>(tad[dt] + tb); t1[dt] = ds_tr_b64<2048>(tad[dt] + tb); }
            } else {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { t0[dt] = ds_tr_b64<4096>(tad[dt] + tb); t1[dt] = ds_tr_b64<4096 + 2048>(tad[dt] + tb); }
            }
            // Keys past N (zero-filled rows of the last tile) are NOT harmless when a query's scores all lie far below zero: their
            // "probability" is exp2(0 - lse * log2 e), +inf once lse < -128 log2 units, and inf * (0 - D) * 0 is NaN in dQ.  Found in a
            // 6000-step soak of the reference recipe (round 6: block 1, a head whose scores had drifted to -130 .. -170).  Masked as in
            // the forward kernel, in the last tile only (wave-uniform branch).
            if (j == ntiles - 1 && (N & (TK - 1)) != 0) {
#pragma unroll
                for (int su = 0; su < 2; ++su)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j * TK + (2 * half + su) * 16 + 4 * g + e >= N) {
#pragma unroll
                            for (int qi = 0; qi < QB; ++qi) s[qi][su][e] = -INFINITY;
                        }
            }
            bf16x8 dsb[QB];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int su = 0; su < 2; ++su)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(s[qi][su][e]);
                        dsb[qi][4 * su + e] = (bf16_t)(pe * (dp[qi][su][e] - d_q[qi]));
                    }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3])
                         :: "memory");
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {   // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
                const u32x4 f = {t0[dt][0], t0[dt][1], t1[dt][0], t1[dt][1]};
#pragma unroll
                for (int qi = 0; qi < QB; ++qi)
                    acc[qi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f), dsb[qi], acc[qi][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        const int q = q0 + 16 * qi;
        if (q < N) {
            bf16_t* orow = dqkv + ((int64_t)b * N + q) * ld + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Vec4<bf16_t>::store(orow + dt * 16 + 4 * g, acc[qi][dt] * 0.125f);
        }
    }
}

template <int QB>
__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ ws, bf16_t* __restrict__ dqkv, int N,
                                                               int Npad, int heads, int gx_arg) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * 128];
    int bx_, bh;
    attn_block(gx_arg, bx_, bh);   // XCD-aware: the workgroups of one (batch, head) share an XCD's L2
    attn_bwd_dq_bf16_body<QB, false>(smem, qkv, dout, nullptr, nullptr, ws, dqkv, N, Npad, heads, bx_, bh);
}

template <int KB, bool OWN>
__device__ __forceinline__ void attn_bwd_dkv_bf16_body(char* smem, const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                       const bf16_t* __restrict__ out, const float* __restrict__ lse,
                                                       const float* __restrict__ ws, bf16_t* __restrict__ dqkv, int N, int Npad, int heads,
                                                       int bx_, int bh) {
    constexpr int TQ = 64, OPB = TQ * 128, STB = 2 * OPB;
    // smem: [stage][Q | dO][64 rows][128 B], then per stage 1 KiB: 64 x -lse*log2e | 64 x dsum | unused (zero-filled by the DMA)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = bh / heads, h = bh - b * heads;
    const int D = heads * HD;
    const int64_t ld = 3 * (int64_t)D;
    const bf16_t* qb = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* kb = qb + D;
    const bf16_t* vb = qb + 2 * D;
    const bf16_t* dob = dout + (int64_t)b * N * D + h * HD;
    const float* nl2_b = OWN ? nullptr : ws + (int64_t)(bh * 2) * Npad;
    const int key0 = bx_ * (64 * KB) + w * (16 * KB) + li;
    RowFrag<bf16_t> kf[KB], vf[KB];
#pragma unroll
    for (int ki = 0; ki < KB; ++ki) {
        const int key = key0 + 16 * ki;
        // K as it is: hd^-0.5 and log2 e are on Q, as in the forward and dQ kernels (round 6).  This kernel used to put the scale on K
        // instead -- bf16(k c) . q against the forward's bf16(q c) . k: two different roundings of the same score, 2^-9 RELATIVE apart.
        // At |score| ~ 10 log2 units that is 1 % in every probability; at the -140 a soak of the reference recipe reached, 0.19 log2
        // units = 14 % in dK and dV against an lse that came from the other rounding (tools/probe/attn_extreme.py)
        kf[ki] = rowfrag_global<bf16_t>(key < N ? kb + (int64_t)key * ld : nullptr, g, 1.0f);
        vf[ki] = rowfrag_global<bf16_t>(key < N ? vb + (int64_t)key * ld : nullptr, g, 1.0f);
    }
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)qb, 0, (unsigned)(((int64_t)(N - 1) * ld + HD) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)dob, 0, (unsigned)(((int64_t)(N - 1) * D + HD) * 2), 0x00020000);
    unsigned voffq[2], voffo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (w * 2 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ kswz<bf16_t>(r);
        voffq[i] = (unsigned)(r * (int)ld * 2 + c * 16);
        voffo[i] = (unsigned)(r * D * 2 + c * 16);
    }
    // the tile's 64 + 64 row statistics ride on the same DMA stream (wave 0, one instruction: lanes 0-15 -lse*log2e,
    // 16-31 dsum, the rest out of range): no VMEM load in the loop has to be waited for behind the next tile's DMA
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)nl2_b, 0, OWN ? 0u : (unsigned)(2 * Npad * 4), 0x00020000);
    // OWN: the tile's 64 + 64 row statistics are computed here, one tile ahead like the DMA: thread t takes a quarter (16 values) of row
    // t / 4 of dO and O straight from global memory (loads issued BEFORE the tile's DMA, so that their wait does not drain it), the four
    // quarters are summed by shuffles, and the results go to the stage's statistics block after the tile's matrix work
    const bf16_t* ob = OWN ? out + (int64_t)b * N * D + h * HD : nullptr;
    bf16x8 so_d[2], so_o[2];
    float so_l = 0.f;
    auto stats_load = [&](int j) {
        const int r = j * TQ + (tid >> 2), c = tid & 3;
        const bool ok = r < N;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x8 zd, zo;
#pragma unroll
            for (int e = 0; e < 8; ++e) { zd[e] = (bf16_t)0.f; zo[e] = (bf16_t)0.f; }
            if (ok) {
                zd = *(const bf16x8*)(dob + (int64_t)r * D + c * 16 + i * 8);
                zo = *(const bf16x8*)(ob + (int64_t)r * D + c * 16 + i * 8);
            }
            so_d[i] = zd; so_o[i] = zo;
        }
        so_l = (ok && c == 0) ? lse[(int64_t)bh * N + r] : 0.f;
    };
    auto stats_store = [&](int j) {
        float ds = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) ds += (float)so_d[i][e] * (float)so_o[i][e];
        ds += __shfl_xor(ds, 1, 64);
        ds += __shfl_xor(ds, 2, 64);
        if ((tid & 3) == 0) {
            float* st = (float*)(smem + 2 * STB + (j & 1) * 1024);
            st[tid >> 2] = -so_l * 1.4426950408889634f;
            st[64 + (tid >> 2)] = ds;
        }
    };
    const unsigned voffs = lane < 16 ? (unsigned)(lane * 16) : lane < 32 ? (unsigned)(Npad * 4 + (lane - 16) * 16) : 0x80000000u;
    const unsigned strq = (unsigned)(TQ * (int)ld * 2), stro = (unsigned)(TQ * D * 2);
    auto issue_tile = [&](int j) {
        char* dst = smem + (j & 1) * STB + w * 2048;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, UMR_LDS_PTR(dst), 16, voffq[0], (unsigned)j * strq, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, UMR_LDS_PTR(dst + 1024), 16, voffq[1], (unsigned)j * strq, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, UMR_LDS_PTR(dst + OPB), 16, voffo[0], (unsigned)j * stro, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, UMR_LDS_PTR(dst + OPB + 1024), 16, voffo[1], (unsigned)j * stro, 0, 0);
        if (!OWN && w == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, UMR_LDS_PTR(smem + 2 * STB + (j & 1) * 1024), 16, voffs, (unsigned)j * 256u, 0, 0);
    };
    const int swk = kswz<bf16_t>(li);
    const char* rad0 = smem + li * 128 + (((0 + g) ^ swk) << 4);
    const char* rad1 = smem + li * 128 + (((4 + g) ^ swk) << 4);
    const int vr = 4 * g + (li >> 2), pp = li & 3;
    const int swv = kswz<bf16_t>(vr) ^ (pp >> 1);
    unsigned tad[4];   // transposing reads; + OPB for the dO tile
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        tad[dt] = (unsigned)(uintptr_t)UMR_LDS_PTR(smem) + vr * 128 + (((dt * 2) ^ swv) << 4) + ((pp & 1) << 3);

    f32x4 accK[KB][4], accV[KB][4];
#pragma unroll
    for (int ki = 0; ki < KB; ++ki)
#pragma unroll
        for (int i = 0; i < 4; ++i) { accK[ki][i] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[ki][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    constexpr float LOG2E = 1.4426950408889634f;
    const int ntiles = (N + TQ - 1) / TQ;
    if (OWN) { stats_load(0); stats_store(0); }
    issue_tile(0);
    for (int j = 0; j < ntiles; ++j) {
        if (OWN) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the statistics block was written with ds_write)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        {
            // Q' = bf16(q * hd^-0.5 * log2 e), the forward kernel's own rounding, applied in place to the two 1-KiB pieces of the Q tile
            // THIS wave's DMA wrote (lane-linear: piece base + 16 lane), before the barrier that publishes the tile
            char* qt = smem + (j & 1) * STB + w * 2048 + lane * 16;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x8 t = *(const bf16x8*)(qt + i * 1024);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] = (bf16_t)((float)t[e] * (0.125f * 1.4426950408889634f));
                *(bf16x8*)(qt + i * 1024) = t;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (OWN && j + 1 < ntiles) stats_load(j + 1);
        if (j + 1 < ntiles) issue_tile(j + 1);
        const int sb = (j & 1) * STB;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 s[KB][2], dp[KB][2], nl2v[2], dsv[2];
#pragma unroll
            for (int su = 0; su < 2; ++su) {
                const char* st = smem + 2 * STB + (j & 1) * 1024 + (16 * (2 * half + su) + 4 * g) * 4;
                nl2v[su] = *(const f32x4*)st;
                dsv[su] = *(const f32x4*)(st + 256);
            }
#pragma unroll
            for (int su = 0; su < 2; ++su) {
                const int off = sb + (2 * half + su) * 2048;
                const bf16x8 a0 = *(const bf16x8*)(rad0 + off), a1 = *(const bf16x8*)(rad1 + off);
                const bf16x8 o0 = *(const bf16x8*)(rad0 + off + OPB), o1 = *(const bf16x8*)(rad1 + off + OPB);
#pragma unroll
                for (int ki = 0; ki < KB; ++ki) {
                    // (the query is on the accumulator rows here: the four rows' -lse * log2 e are exactly the C operand)
                    s[ki][su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, kf[ki].v[0], nl2v[su], 0, 0, 0);
                    s[ki][su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, kf[ki].v[1], s[ki][su], 0, 0, 0);      // S[query][key]
                    dp[ki][su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o0, vf[ki].v[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    dp[ki][su] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o1, vf[ki].v[1], dp[ki][su], 0, 0, 0);    // dP[query][key]
                }
            }
            u32x2 t0[4], t1[4];
            const unsigned tb = (unsigned)sb;
            if (half == 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { t0[dt] = ds_tr_b64<OPB>(tad[dt] + tb); t1[dt] = ds_tr_b64<OPB + 2048>(tad[dt] + tb); }
            } else {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { t0[dt] = ds_tr_b64<OPB + 4096>(tad[dt] + tb); t1[dt] = ds_tr_b64<OPB + 4096 + 2048>(tad[dt] + tb); }
            }
            bf16x8 pb[KB], dsb[KB];
#pragma unroll
            for (int ki = 0; ki < KB; ++ki)
#pragma unroll
                for (int su = 0; su < 2; ++su)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(s[ki][su][e]);
                        pb[ki][4 * su + e] = (bf16_t)pe;
                        dsb[ki][4 * su + e] = (bf16_t)(pe * (dp[ki][su][e] - dsv[su][e]));
                    }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3])
                         :: "memory");
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {   // dV^T[d][key] += dO^T[d][query] P[query][key]
                const u32x4 f = {t0[dt][0], t0[dt][1], t1[dt][0], t1[dt][1]};
#pragma unroll
                for (int ki = 0; ki < KB; ++ki)
                    accV[ki][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f), pb[ki], accV[ki][dt], 0, 0, 0);
            }
            u32x2 u0[4], u1[4];
            if (half == 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { u0[dt] = ds_tr_b64<0>(tad[dt] + tb); u1[dt] = ds_tr_b64<2048>(tad[dt] + tb); }
            } else {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { u0[dt] = ds_tr_b64<4096>(tad[dt] + tb); u1[dt] = ds_tr_b64<4096 + 2048>(tad[dt] + tb); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(u0[0]), "+v"(u0[1]), "+v"(u0[2]), "+v"(u0[3]), "+v"(u1[0]), "+v"(u1[1]), "+v"(u1[2]), "+v"(u1[3])
                         :: "memory");
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {   // dK^T[d][key] += Q^T[d][query] dS[query][key]
                const u32x4 f = {u0[dt][0], u0[dt][1], u1[dt][0], u1[dt][1]};
#pragma unroll
                for (int ki = 0; ki < KB; ++ki)
                    accK[ki][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f), dsb[ki], accK[ki][dt], 0, 0, 0);
            }
        }
        if (OWN && j + 1 < ntiles) stats_store(j + 1);
    }
#pragma unroll
    for (int ki = 0; ki < KB; ++ki) {
        const int key = key0 + 16 * ki;
        if (key < N) {
            bf16_t* krow = dqkv + ((int64_t)b * N + key) * ld + D + h * HD;
            bf16_t* vrow = krow + D;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                Vec4<bf16_t>::store(krow + dt * 16 + 4 * g, accK[ki][dt] * 0.69314718055994531f);   // dK = hd^-0.5 dS^T q = dS^T Q' / log2 e
                Vec4<bf16_t>::store(vrow + dt * 16 + 4 * g, accV[ki][dt]);
            }
        }
    }
}

template <int KB>
__global__ __launch_bounds__(256, UMR_ATTN_DKV_MIN_WAVES) void attn_bwd_dkv_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ ws, bf16_t* __restrict__ dqkv, int N,
                                                                int Npad, int heads, int gx_arg) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * 128 + 2 * 1024];
    int bx_, bh;
    attn_block(gx_arg, bx_, bh);   // XCD-aware: the workgroups of one (batch, head) share an XCD's L2
    attn_bwd_dkv_bf16_body<KB, false>(smem, qkv, dout, nullptr, nullptr, ws, dqkv, N, Npad, heads, bx_, bh);
}

// Backward of SHORT sequences in one launch (the reference recipe: 65 tokens per crop, 320 (batch, head) pairs; three launches of 7-16 us
// each were launch boundaries more than work): workgroups [0, half) are the dQ workgroups, [half, 2 half) the dK / dV workgroups, each
// with the XCD-aware map over its own half; both take their rows' statistics themselves (OWN), so there is no prep launch and no
// workspace.  Same arithmetic per output as the three-launch form except the order of the 64 products in rowsum(dO * O).
__global__ __launch_bounds__(256) void attn_bwd_small_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                  const bf16_t* __restrict__ out, const float* __restrict__ lse,
                                                                  bf16_t* __restrict__ dqkv, int N, int heads, int gx_arg, int half) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 64 * 128 + 2 * 1024];
    int bid = (int)blockIdx.x;
    const bool dkv = bid >= half;     // (workgroup-uniform)
    if (dkv) bid -= half;
    int bx_, bh;
    attn_block_of(bid, half, gx_arg, bx_, bh);
    if (dkv) attn_bwd_dkv_bf16_body<1, true>(smem, qkv, dout, out, lse, nullptr, dqkv, N, 0, heads, bx_, bh);
    else attn_bwd_dq_bf16_body<1, true>(smem, qkv, dout, out, lse, nullptr, dqkv, N, 0, heads, bx_, bh);
}

// ------------------------------------------------------------------ backward: dQ
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ dsum,
                                                          T* __restrict__ dqkv, int N, int heads, int gx_arg) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KT * AT<T>::ROWB];
    char* sK = smem;
    char* sV = smem + KT * AT<T>::ROWB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int bx_, bh;
    attn_block(gx_arg, bx_, bh);   // XCD-aware: the workgroups of one (batch, head) share an XCD's L2
    const int b = bh / heads, h = bh - b * heads;
    const int D = heads * HD;
    const int64_t ld = 3 * (int64_t)D;
    const T* qb = qkv + (int64_t)b * N * ld + h * HD;
    const T* kb = qb + D;
    const T* vb = qb + 2 * D;
    const int q = bx_ * 64 + w * 16 + li;
    const bool qok = q < N;
    const RowFrag<T> qf = rowfrag_global<T>(qok ? qb + (int64_t)q * ld : nullptr, g, 0.125f);
    const RowFrag<T> dof = rowfrag_global<T>(qok ? dout + ((int64_t)b * N + q) * D + h * HD : nullptr, g, 1.0f);
    const float lse_q = qok ? lse[(int64_t)bh * N + q] : 0.f;
    const float d_q = qok ? dsum[(int64_t)bh * N + q] : 0.f;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < N; k0 += KT) {
        __syncthreads();
        load_tile<T>(sK, kb, ld, k0, N, tid);
        load_tile<T>(sV, vb, ld, k0, N, tid);
        __syncthreads();
        f32x4 ds[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const RowFrag<T> kf = rowfrag_lds<T>(sK, 16 * sub + li, g);
            const RowFrag<T> vf = rowfrag_lds<T>(sV, 16 * sub + li, g);
            const f32x4 s = mma_rows<T>(f32x4{0.f, 0.f, 0.f, 0.f}, kf, qf);    // S^T[key][query]
            const f32x4 dp = mma_rows<T>(f32x4{0.f, 0.f, 0.f, 0.f}, vf, dof);  // dP^T[key][query]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool kok = (k0 + 16 * sub + 4 * g + e) < N;
                const float p = kok ? fexp<T>(s[e] - lse_q) : 0.f;
                ds[sub][e] = p * (dp[e] - d_q);
            }
        }
        mma_trans<T>(acc, sK, ds[0], ds[1], lane);  // dQ^T[d][query] += K^T[d][key] dS^T[key][query]
    }
    if (qok) {
        T* orow = dqkv + ((int64_t)b * N + q) * ld + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) Vec4<T>::store(orow + dt * 16 + 4 * g, acc[dt] * 0.125f);
    }
}

// ------------------------------------------------------------------ backward: dK, dV
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           T* __restrict__ dqkv, int N, int heads, int gx_arg) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KT * AT<T>::ROWB];
    char* sQ = smem;
    char* sO = smem + KT * AT<T>::ROWB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int bx_, bh;
    attn_block(gx_arg, bx_, bh);   // XCD-aware: the workgroups of one (batch, head) share an XCD's L2
    const int b = bh / heads, h = bh - b * heads;
    const int D = heads * HD;
    const int64_t ld = 3 * (int64_t)D;
    const T* qb = qkv + (int64_t)b * N * ld + h * HD;
    const T* kb = qb + D;
    const T* vb = qb + 2 * D;
    const T* dob = dout + (int64_t)b * N * D + h * HD;
    const int key = bx_ * 64 + w * 16 + li;
    const bool kok = key < N;
    const RowFrag<T> kf = rowfrag_global<T>(kok ? kb + (int64_t)key * ld : nullptr, g, 0.125f);
    const RowFrag<T> vf = rowfrag_global<T>(kok ? vb + (int64_t)key * ld : nullptr, g, 1.0f);
    const float* lse_b = lse + (int64_t)bh * N;
    const float* dsum_b = dsum + (int64_t)bh * N;

    f32x4 accK[4], accV[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { accK[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int q0 = 0; q0 < N; q0 += KT) {
        __syncthreads();
        load_tile<T>(sQ, qb, ld, q0, N, tid);
        load_tile<T>(sO, dob, (int64_t)D, q0, N, tid);
        __syncthreads();
        f32x4 p[2], ds[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const RowFrag<T> qf = rowfrag_lds<T>(sQ, 16 * sub + li, g);
            const RowFrag<T> dof = rowfrag_lds<T>(sO, 16 * sub + li, g);
            const f32x4 s = mma_rows<T>(f32x4{0.f, 0.f, 0.f, 0.f}, qf, kf);    // S[query][key]
            const f32x4 dp = mma_rows<T>(f32x4{0.f, 0.f, 0.f, 0.f}, dof, vf);  // dP[query][key]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int qq = q0 + 16 * sub + 4 * g + e;
                const bool qok = qq < N;
                const float pe = qok ? fexp<T>(s[e] - lse_b[qok ? qq : 0]) : 0.f;
                p[sub][e] = pe;
                ds[sub][e] = pe * (dp[e] - (qok ? dsum_b[qq] : 0.f));
            }
        }
        mma_trans<T>(accV, sO, p[0], p[1], lane);    // dV^T[d][key] += dO^T[d][query] P[query][key]
        mma_trans<T>(accK, sQ, ds[0], ds[1], lane);  // dK^T[d][key] += Q^T[d][query] dS[query][key]
    }
    if (kok) {
        T* krow = dqkv + ((int64_t)b * N + key) * ld + D + h * HD;
        T* vrow = krow + D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            Vec4<T>::store(krow + dt * 16 + 4 * g, accK[dt] * 0.125f);
            Vec4<T>::store(vrow + dt * 16 + 4 * g, accV[dt]);
        }
    }
}

}  // namespace

extern "C" int umr_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int heads, int head_dim, int dtype,
                                 umr_stream_t stream) {
    UMR_CHECK_ARG(qkv && out, "attention_fwd: null pointer");
    UMR_CHECK_ARG(B > 0 && N > 0 && heads > 0, "attention_fwd: empty problem");
    if (head_dim != HD) return umr_set_error(UMR_ERR_UNSUPPORTED, "attention: head_dim must be 64");
    hipStream_t s = (hipStream_t)stream;
    // 1-D grids of gx x (B * heads) workgroups; the kernels map their id XCD-aware (attn_block).  UMR_ATTN_XCD=0: the old order
    static const int xcd_map = umr_env_int("UMR_ATTN_XCD", 1);
    const int gx1 = (N + 63) / 64, gx2 = (N + 127) / 128;
    UMR_CHECK_ARG((int64_t)gx1 * B * heads < (1ll << 31), "attention_fwd: grid too large");
    dim3 g((unsigned)(gx1 * B * heads)), b(256);
    const int a1 = xcd_map ? gx1 : -gx1, a2 = xcd_map ? gx2 : -gx2;
    static const int fast_fwd = umr_env_int("UMR_ATTN_FAST", 1);   // 0: the generic kernel for bf16 too (A/B)
    if (dtype == UMR_BF16 && fast_fwd) {
        if (N >= 128 && fast_fwd != 2) {   // 32 queries per wave: half the LDS reads per MFMA
            dim3 g2((unsigned)(gx2 * B * heads));
#ifdef UMR_ATTN_BARE
            // probe build only: extra dynamic LDS per workgroup caps the resident workgroups per CU (32 KiB static: 0 -> 4 per CU by
            // registers, 24576 -> 2, 98304 -> 1): the bare loop at 4, 2 and 1 waves per SIMD
            static const int pad = umr_env_int("UMR_ATTN_BARE_LDS", 0);
            if (pad > 0) (void)hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, pad);
            hipLaunchKernelGGL(attn_fwd_bf16_kernel<2>, g2, b, pad, s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, heads, a2);
#else
            hipLaunchKernelGGL(attn_fwd_bf16_kernel<2>, g2, b, 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, heads, a2);
#endif
        } else {
            hipLaunchKernelGGL(attn_fwd_bf16_kernel<1>, g, b, 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, heads, a1);
        }
    }
    else if (dtype == UMR_BF16) hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)qkv, (bf16_t*)out, lse, N, heads, a1);
    else if (dtype == UMR_F32) hipLaunchKernelGGL(attn_fwd_kernel<float>, g, b, 0, s, (const float*)qkv, (float*)out, lse, N, heads, a1);
    else return umr_set_error(UMR_ERR_INVALID, "attention_fwd: dtype");
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}

extern "C" int64_t umr_attention_bwd_workspace(int B, int N, int heads) {
    if (B <= 0 || N <= 0 || heads <= 0) return 0;
    return (int64_t)B * heads * 2 * ((N + 63) / 64 * 64) * 4;
}

extern "C" int umr_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dsum_ws, void* dqkv,
                                 int B, int N, int heads, int head_dim, int dtype, umr_stream_t stream) {
    UMR_CHECK_ARG(qkv && out && dout && lse && dsum_ws && dqkv, "attention_bwd: null pointer");
    UMR_CHECK_ARG(B > 0 && N > 0 && heads > 0, "attention_bwd: empty problem");
    if (head_dim != HD) return umr_set_error(UMR_ERR_UNSUPPORTED, "attention: head_dim must be 64");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * N * heads;
    static const int xcd_map = umr_env_int("UMR_ATTN_XCD", 1);
    const int gx1 = (N + 63) / 64, gx2 = (N + 127) / 128;
    UMR_CHECK_ARG((int64_t)gx1 * B * heads < (1ll << 31), "attention_bwd: grid too large");
    const int a1 = xcd_map ? gx1 : -gx1, a2 = xcd_map ? gx2 : -gx2;
    dim3 gp((unsigned)((total + 255) / 256)), g((unsigned)(gx1 * B * heads)), b(256);
    static const int fast_bwd = umr_env_int("UMR_ATTN_FAST", 1);   // 0: the generic kernels for bf16 too (A/B)
    // short sequences: one launch (dQ and dK / dV workgroups side by side, no prep pass).  UMR_ATTN_BWD_FUSED=0: the three launches (A/B;
    // umr_set_debug_option)
    if (dtype == UMR_BF16 && fast_bwd && N < 128 && umr_opt_or(UMR_OPT_ATTN_BWD_FUSED, 1) != 0) {
        const int half = gx1 * B * heads;
        UMR_CHECK_ARG(2ll * half < (1ll << 31), "attention_bwd: grid too large");
        hipLaunchKernelGGL(attn_bwd_small_bf16_kernel, dim3((unsigned)(2 * half)), b, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, (const bf16_t*)out,
                           lse, (bf16_t*)dqkv, N, heads, a1, half);
    } else if (dtype == UMR_BF16 && fast_bwd) {
        const int Npad = (N + 63) / 64 * 64;
        const int64_t tp = (int64_t)B * heads * Npad * 8;
        hipLaunchKernelGGL(attn_bwd_prep_bf16_kernel, dim3((unsigned)((tp + 255) / 256)), b, 0, s, (const bf16_t*)out, (const bf16_t*)dout, lse,
                           dsum_ws, B, N, Npad, heads);
        if (N >= 128 && fast_bwd != 2) {
            dim3 g2((unsigned)(gx2 * B * heads));
            hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<2>, g2, b, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, dsum_ws, (bf16_t*)dqkv, N, Npad, heads, a2);
            hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel<2>, g2, b, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, dsum_ws, (bf16_t*)dqkv, N, Npad, heads, a2);
        } else {
            hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel<1>, g, b, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, dsum_ws, (bf16_t*)dqkv, N, Npad, heads, a1);
            hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel<1>, g, b, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, dsum_ws, (bf16_t*)dqkv, N, Npad, heads, a1);
        }
    } else if (dtype == UMR_BF16) {
        hipLaunchKernelGGL(attn_bwd_prep_kernel<bf16_t>, gp, b, 0, s, (const bf16_t*)out, (const bf16_t*)dout, dsum_ws, B, N, heads);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, dsum_ws, (bf16_t*)dqkv, N, heads, a1);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<bf16_t>, g, b, 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, dsum_ws, (bf16_t*)dqkv, N, heads, a1);
    } else if (dtype == UMR_F32) {
        hipLaunchKernelGGL(attn_bwd_prep_kernel<float>, gp, b, 0, s, (const float*)out, (const float*)dout, dsum_ws, B, N, heads);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<float>, g, b, 0, s, (const float*)qkv, (const float*)dout, lse, dsum_ws, (float*)dqkv, N, heads, a1);
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<float>, g, b, 0, s, (const float*)qkv, (const float*)dout, lse, dsum_ws, (float*)dqkv, N, heads, a1);
    } else return umr_set_error(UMR_ERR_INVALID, "attention_bwd: dtype");
    UMR_LAUNCH_CHECK();
    return UMR_OK;
}
