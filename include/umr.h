/* unmore_amd C-ABI: hand-written gfx950 (MI355X) kernels for unMORE's stage-1
 * ObjectnessNet hot path.  Plain pointers + extents + a HIP stream; the caller
 * (PyTorch, or any host) owns every buffer; nothing here allocates, synchronises
 * or throws.  Every entry point returns 0 on success, a negative umr_status
 * otherwise (umr_last_error_string() has the text).
 *
 * The reference (pure PyTorch, no FFI) has no C interface to mirror, so each
 * entry point cites the reference Python op(s) it replaces (paths relative to the
 * reference root).  Activations are NHWC / [rows, channels] row-major; weights
 * are passed pre-packed as documented per entry point.
 */
#ifndef UMR_H
#define UMR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* umr_stream_t; /* hipStream_t */

enum umr_status { UMR_OK = 0, UMR_ERR_INVALID = -1, UMR_ERR_UNSUPPORTED = -2, UMR_ERR_HIP = -1000 };
/* UMR_BF16X3 (umr_gemm_nt only): f32 VALUES held as three bf16 planes per value, x = h + m + l (exact for every finite f32 in
 * bf16's exponent range; written by umr_split3 or by a GEMM with UMR_EPI_OUT_X3).  A row of K logical elements is
 * [h(K) | m(K) | l(K)], 3K bf16; see umr_gemm_desc. */
enum umr_dtype { UMR_F32 = 0, UMR_BF16 = 1, UMR_BF16X3 = 2 };

int umr_version(void);
const char* umr_last_error_string(void);

/* ---- how UMR_F32 GEMMs (umr_gemm_nt / umr_gemm_tn with dtype UMR_F32; the reference computes in fp32,
 * object_reasoning.py:74) form their products.  Process-wide, may be changed between launches (not per stream).
 *   UMR_F32_EXACT: v_mfma_f32_*_f32 -- IEEE f32 multiply-adds; inf / NaN / denormal operands behave as in an f32 FMA chain.
 *   UMR_F32_X3 (default; env UMR_F32_X3=0 selects EXACT at load): every operand is split into three bf16 terms x = h + m + l
 *     and a product is six bf16 MFMAs accumulated in f32 (hh + hm + mh + hl + lh + mm; the dropped terms are <= 2^-26 of the
 *     product).  fp32-GRADE, not bit-identical to an f32 FMA chain: rms error vs float64 equals the f32 MFMA's for FINITE
 *     operands with 2^-100 < |x| < 2^126.  Outside that range it is NOT IEEE: an inf operand gives NaN (inf - inf in the split),
 *     |x| within 2^-8 of FLT_MAX rounds its leading term to inf -> NaN, and the m / l terms of |x| < 2^-110 underflow in
 *     bf16 (the product then keeps only 8 / 16 significant bits -- of a value that is itself ~1e-33).  A NaN operand gives NaN
 *     in both modes.  tests/test_gemm_gpu.py::test_f32_x3_* pins this behaviour.
 *   UMR_F32_X3_FAST (opt-in, never a default): as X3, but the UMR_BF16X3 plane GEMMs keep only the three leading terms
 *     (hh + hm + mh): every product to 2^-16 instead of 2^-24, at half the matrix work.  Meant for inference
 *     under a 1e-4 output tolerance (measured field error and peak-index agreement: bench.py --workload cfg5, alt_fp32_3term);
 *     the non-plane f32 kernels behave as in X3. */
enum umr_f32_mode { UMR_F32_EXACT = 0, UMR_F32_X3 = 1, UMR_F32_X3_FAST = 2 };
int umr_set_f32_mode(int mode);   /* returns UMR_OK or UMR_ERR_INVALID */
int umr_get_f32_mode(void);

/* CU budget of the persistent GEMM grids (umr_gemm_nt on the 256x256 kernel: one workgroup owns a CU's whole LDS).  0 (default;
 * env UMR_CU_BUDGET) = the grid is a small multiple of the CU count and the dispatcher balances it; n > 0 = the grid is exactly
 * min(n, CUs) workgroups, so the other CUs stay free for kernels that must run BESIDE it -- RCCL's all-reduce of the previous
 * gradient bucket during backward (data-parallel training, train_objectness_net.py has no such step: build-defined).  Tile shape
 * and K-split are planned on the device's CU count either way, and every output element's K sum is the same instruction
 * sequence in any workgroup: results are bit-identical across budgets.  The weight-gradient kernel (umr_gemm_tn) launches
 * (split, tile) workgroups whose count does not depend on the CU count and needs no budget.  Process-wide, run-time. */
int umr_set_cu_budget(int cus);   /* returns UMR_OK or UMR_ERR_INVALID */
int umr_get_cu_budget(void);

/* Debug / A-B options (forced tile sizes, split-K factors, kernel variants: the UMR_* names csrc/umr_common.h lists).  Process-wide.
 * The environment variable of the same name is read ONCE, when the library is loaded -- no entry point calls getenv() on its launch
 * path, so the library's behaviour never depends on environment changes made while it runs and is safe beside a host that calls
 * setenv(); tests and probes switch an option between launches with this call instead.  value: the text the environment variable
 * would hold ("256", "0", "m"), NULL = unset (the library's own default).  Unknown names return UMR_ERR_INVALID.  None of these
 * options changes WHAT is computed beyond rounding order where the option's test says so; they exist to compare forms. */
int umr_set_debug_option(const char* name, const char* value);
int umr_get_debug_option(const char* name, int* value, int* is_set);   /* letter options report the character code */

/* ---- GEMM "NT" with implicit 3x3 convolution and fused epilogue ------------
 * C[M,N] = epi(A[M,K] . B[N,K]^T), fp32 accumulate on MFMA.
 * Replaces torch.nn.Linear / 1x1 nn.Conv2d (models/dpt/vit.py:84,263-327,
 * models/dpt/blocks.py:347-355, models/objectness_net.py:110,114), timm
 * Attention.qkv/proj, Mlp.fc1/fc2, and -- with conv != 0 -- 3x3 nn.Conv2d
 * forward / data-gradient (models/dpt/blocks.py:80-115,262-280,
 * models/dpt/vit.py:329-335, models/objectness_net.py:112).
 * epilogue order: v = acc (+bias[n]) (+rowbias[m/rows_per_batch][n]);
 *   MASK_RELU: v *= (aux>0) | MASK_DGELU: v *= gelu'(aux) | ADD_AUX: v += aux;
 *   ADD_AUX2: v += aux2;  c2_mode==2: C2 = v;  v = act(v);  C = v;
 *   c2_mode==1: C2 = relu(v).
 * Rounding: fp32 (parity mode) applies every step in f32 and rounds once.  On the bf16 large-tile path the aux steps
 * (ADD_AUX, MASK_RELU, MASK_DGELU) act on the value already rounded to bf16 -- bf16(bf16(acc + bias) op aux), what separate
 * bf16 layers compute; for MASK_RELU that is bit-identical to rounding last. */
enum umr_epi_flags {
    UMR_EPI_BIAS = 1, UMR_EPI_ADD_AUX = 2, UMR_EPI_MASK_RELU = 4, UMR_EPI_MASK_DGELU = 8,
    UMR_EPI_ADD_AUX2 = 16, UMR_EPI_OUT_F32 = 32, UMR_EPI_ROWBIAS = 64,
    UMR_EPI_OUT_X3 = 128,  /* dtype UMR_BF16X3 only: C is written as three bf16 planes [h(N) | m(N) | l(N)] per row (ldc >= 3N) */
    /* dtype UMR_BF16X3 only: the operand is given as / written as bf16 planes ([M][3N], row stride >= 3N) instead of f32 [M][N] */
    UMR_EPI_AUX_X3 = 256, UMR_EPI_AUX2_X3 = 512, UMR_EPI_C2_X3 = 1024
};
enum umr_act { UMR_ACT_NONE = 0, UMR_ACT_RELU = 1, UMR_ACT_GELU = 2, UMR_ACT_TANH = 3, UMR_ACT_SIGMOID = 5 /* 4 = sine, head_out only */ };

typedef struct umr_gemm_desc {
    const void* A;        /* [M,K] rows (lda) or NHWC input when conv != 0 */
    const void* B;        /* [N,K] rows (ldb); conv: [N][3][3][Cin] */
    void* C;              /* [M,N] (ldc), dtype or f32 with UMR_EPI_OUT_F32 */
    void* C2;             /* optional second output (ldc2), dtype */
    const float* bias;    /* [N] f32 */
    const void* aux;      /* [M,N] (ldaux), dtype */
    const void* aux2;     /* [M,N] (ldaux2), dtype */
    const float* rowbias; /* [ceil(M/rows_per_batch), N] f32 */
    int64_t lda, ldb, ldc, ldc2, ldaux, ldaux2;
    int32_t M, N, K;
    int32_t dtype;        /* umr_dtype of A, B, aux, aux2, C2 (and C) */
    int32_t flags, act, c2_mode, rows_per_batch;
    int32_t conv;         /* 0 plain; 1 = 3x3 stride 1 pad 1; 2 = 3x3 stride 2 pad 1 */
    int32_t nb, H, W, Cin, Ho, Wo; /* conv geometry: A is [nb,H,W,Cin]; M = nb*Ho*Wo; K = 9*Cin */
    /* optional row remaps (0 = identity): logical row m -> (m / rows_in) * rows_out + row_off + m % rows_in.
     * a_*: rows of A (plain mode); c_*: rows of C, C2, aux, aux2 (token buffers with a class-token row per image,
     * models/dpt/vit.py:87-88,188-193).  aux_mod > 0: aux row = m % aux_mod (broadcast over images: pos-embed add, vit.py:193) */
    int32_t a_rows_in, a_rows_out, a_row_off;
    int32_t c_rows_in, c_rows_out, c_row_off;
    int32_t aux_mod;
    /* optional fused row reduction (the 1024 -> {1,2} output layer of a head, objectness_net.py:116,133, folded into the
     * epilogue of the layer that produces its input): for each 64-column slice t of the output,
     *   red_out[t][m][c] = sum_{n in slice t} C[m,n] * red_w[c][n]      (C as stored: after bias / activation / rounding)
     * red_w: [red_c][N] f32, red_c in {1,2}; red_out: [ceil(N/64)][M][red_c] f32 partial sums, summed in fixed order by
     * umr_head_out_finish.  no_store = 1 skips the store of C itself (inference: the activation is not needed again).
     * Only the persistent 256x256 bf16 path with a bias / ReLU-only epilogue implements it: call
     * umr_gemm_nt_rowreduce_ok first; umr_gemm_nt fails (UMR_ERR_UNSUPPORTED / UMR_ERR_INVALID) rather than silently
     * ignoring the request. */
    const float* red_w;
    float* red_out;
    int32_t red_c;
    int32_t no_store;
} umr_gemm_desc;

/* dtype UMR_BF16X3 -- the fast form of the fp32 parity mode (the reference runs in fp32, object_reasoning.py:74,
 * train_objectness_net.py:81): both operands are f32 values pre-split into bf16 planes; the product is the six-term sum of
 * UMR_F32_X3 (same accuracy and the same range caveats) computed as six bf16 K-tiles per logical K-tile on the persistent 256x256
 * kernel, with no split arithmetic in the loop.  A: [M][3K] bf16 (lda >= 3K; no A-row remap), or NHWC with 3*Cin bf16 per pixel
 * [h(Cin) | m(Cin) | l(Cin)] when conv == 1; B: [N][3K] bf16 (ldb >= 3K; conv: K = 9*Cin ordered (ky,kx,ci) inside each plane).
 * K (conv: Cin) must be a multiple of 64, N of 8.  Epilogue: everything umr_gemm_desc describes except tanh / sigmoid, in f32
 * (GELU = the exact erf form): bias, rowbias, one of ADD_AUX / MASK_RELU / MASK_DGELU, ADD_AUX2, c2_mode 1 / 2, act none / ReLU /
 * GELU, C-row remap and aux_mod (plain GEMM).  Operand FORMATS: C is f32 [M][N] (UMR_EPI_OUT_F32) or planes [M][3N]
 * (UMR_EPI_OUT_X3) -- exactly one of the two; aux / aux2 / C2 are f32 [M][N] unless UMR_EPI_AUX_X3 / AUX2_X3 / C2_X3 say planes
 * (a plane operand is read back as h + m + l = the f32 value; the ReLU mask reads only h, whose sign is the value's).  Or, for a
 * plain GEMM at inference, the fused row reduction red_* with no_store = 1 (bias / ReLU only; dot products of the f32 values, C
 * never stored).  Anything else returns UMR_ERR_UNSUPPORTED.
 * Small problems (the transformer at a few thousand tokens, 3x3 convs on the 4x4 ... 64x64 DPT maps: fewer tiles than CUs, or a
 * ragged last round) are cut along K into work items whose f32 partial sums go to workspace slabs, added in slab order by a
 * second launch that applies the epilogue (bitwise reproducible): pass umr_gemm_nt_ws a workspace of
 * umr_gemm_nt_workspace() + umr_gemm_nt_x3_workspace(d) bytes (the latter 0 when the problem is not split). */
int umr_gemm_nt(const umr_gemm_desc* d, umr_stream_t stream);
/* Same, with a scratch buffer that lets small problems use split-K: plain GEMMs with few 128x128 tiles and a long K (the
 * transformer's projections at a few thousand tokens -- the reference's own recipe trains on 128x128 images, batch 20 = 1300
 * tokens, README.md:148-155, train_objectness_net.py:783-788,815-817) run as tiles x splits workgroups; partial accumulators go to the workspace and the
 * workgroup that arrives last at a tile adds them in split order (bitwise reproducible) and runs the epilogue.  workspace:
 * umr_gemm_nt_workspace() bytes of device memory, 16-byte aligned, whose first 16 KiB are ZERO on first use (tile counters; every
 * launch leaves them zero) and which no other stream uses concurrently.  workspace == NULL is umr_gemm_nt. */
/* HARD PRECONDITIONS of the workspace (the split-K hand-over is a ticket counter per tile; csrc/gemm_nt.hip documents the memory
 * ordering it relies on): (1) its first 16 KiB are zero before the first launch that uses it; (2) it belongs to ONE stream -- two
 * launches that may run concurrently must not share it (results would be silently wrong); (3) a launch that was enqueued and then
 * aborted (device fault, reset) leaves counters undefined: zero them again before reuse (umr_gemm_nt_ws does so itself when the
 * launch call reports an error).  A ticket outside [0, splits) -- i.e. a violated precondition -- does NOT trap (no entry point of
 * this library aborts the process or the device context): the kernel counts it in the workspace's error word (the last of the 4096
 * counter ints), heals the counter and leaves that output tile unwritten; umr_gemm_nt_ws_status reads and clears the word.  The
 * textbook agent-scope release / acquire hand-over -- the slow reference form the default is tested against, bit for bit -- is
 * available at run time (umr_set_debug_option("UMR_SPLITK_FENCE", "1"); the environment variable of that name is read once, when the
 * library is loaded) and as a build (libumr_fence.so, make fence).  The default form's reliance on sc1 write-through
 * stores and sc1 loads is the chip's documented behaviour, not a promise of the HIP memory model: DESIGN.md section 4 quotes the guide.
 * umr_gemm_nt_splits: the number of K ranges umr_gemm_nt_ws would use for d with a workspace of that size (1 = not split). */
int64_t umr_gemm_nt_workspace(void);
int umr_gemm_nt_splits(const umr_gemm_desc* d, int64_t workspace_bytes);
int64_t umr_gemm_nt_x3_workspace(const umr_gemm_desc* d);   /* extra bytes a UMR_BF16X3 problem wants for its K-split slabs */
int umr_gemm_nt_ws(const umr_gemm_desc* d, void* workspace, int64_t workspace_bytes, umr_stream_t stream);
/* Diagnostic, and the ONE entry point that synchronises (it waits for `stream`): *bad_tickets = the number of out-of-range split-K
 * tickets seen on this workspace since the last call (0 = every split GEMM that used it wrote all of its tiles); clears the count. */
int umr_gemm_nt_ws_status(void* workspace, umr_stream_t stream, int* bad_tickets);
/* rows x K f32 (row stride ld_src elements) -> rows x [h(K) | m(K) | l(K)] bf16 (row stride ld_dst >= 3K elements):
 * h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), round-to-nearest-even each.  K % 4 == 0. */
int umr_split3(const float* src, void* dst, int64_t rows, int K, int64_t ld_src, int64_t ld_dst, umr_stream_t stream);
/* the inverse: rows x [h(K) | m(K) | l(K)] bf16 -> rows x K f32, x = (h + m) + l (exact) */
int umr_unsplit3(const void* src, float* dst, int64_t rows, int K, int64_t ld_src, int64_t ld_dst, umr_stream_t stream);
/* the same with a gather of source rows: destination row r is source row (r / rows_in) * rows_out + row_off + r % rows_in (rows_in
 * == 0: identity) -- the token rows of every image without its class-token row (models/dpt/vit.py:87-88) as a compact operand */
int umr_split3_rows(const float* src, void* dst, int64_t rows, int K, int64_t ld_src, int64_t ld_dst, int rows_in, int rows_out,
                    int row_off, umr_stream_t stream);
/* 1 if umr_gemm_nt would run d (ignoring red_*, no_store) on the path that implements the fused row reduction */
int umr_gemm_nt_rowreduce_ok(const umr_gemm_desc* d);

/* ---- weight-gradient GEMM "TN" (reduction over rows) -----------------------
 * dW[N,K] (f32) = sum_m dY[m,N]^T . X[m,K]   (X rows shifted per tap when conv != 0,
 * K = 9*Cin, dW laid out [N][3][3][Cin]).  Replaces the autograd weight
 * gradients of the modules listed above (train_objectness_net.py:259).
 * Split over `splits` row ranges into workspace slabs [splits][N][K] f32, then
 * reduced in fixed order (bitwise reproducible).  workspace_bytes must be
 * >= umr_gemm_tn_workspace(d).
 * dtype UMR_BF16X3: dY [M][h(N) | m(N) | l(N)] and X [M][h(K) | m(K) | l(K)] (conv == 1: NHWC pixels of [h(Cin) | m(Cin) |
 * l(Cin)]) hold f32 values as bf16 planes (lddy >= 3N, ldx >= 3K); dW / dbias are the fp32-grade six-term products / sums (see
 * umr_gemm_nt).  N, K (conv: Cin) multiples of 8; any map size; no row remaps, no stride-2 conv (UMR_ERR_UNSUPPORTED). */
typedef struct umr_gemm_tn_desc {
    const void* dY;       /* [M,N] (lddy) */
    const void* X;        /* [M,K] (ldx) or NHWC input when conv != 0 */
    float* dW;            /* [N,K] f32 (lddw) */
    float* dbias;         /* optional [N] f32: column sums of dY */
    void* workspace;
    int64_t workspace_bytes;
    int64_t lddy, ldx, lddw;
    int32_t M, N, K;
    int32_t dtype;
    int32_t accumulate;   /* dW += instead of = */
    int32_t conv, nb, H, W, Cin, Ho, Wo;
    /* optional row remaps (0 = identity), as in umr_gemm_desc: dy_* for dY rows, x_* for X rows (plain mode) */
    int32_t dy_rows_in, dy_rows_out, dy_row_off;
    int32_t x_rows_in, x_rows_out, x_row_off;
} umr_gemm_tn_desc;

int64_t umr_gemm_tn_workspace(const umr_gemm_tn_desc* d);
int umr_gemm_tn(const umr_gemm_tn_desc* d, umr_stream_t stream);

/* ---- LayerNorm over rows of D (timm LayerNorm eps 1e-6 inside Block; vit.py:196-199) ----
 * fwd with dtype UMR_BF16X3: x is f32, y is written as bf16 planes [M][h(D) | m(D) | l(D)] (the operand of the plane GEMMs).
 * bwd: dx = LN'(dy) (+ dres if non-null: residual-branch gradient), dgamma/dbeta (f32, =/+= per `accumulate`). */
int umr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int M, int D, float eps, int dtype, umr_stream_t stream);
int64_t umr_layernorm_bwd_workspace(int M, int D);
int umr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                      const void* dres, void* dx, float* dgamma, float* dbeta, int accumulate, void* workspace,
                      int64_t workspace_bytes, int M, int D, int dtype, umr_stream_t stream);
/* The same in two steps.  _rows: dx (+ dres) and the per-workgroup partial sums of dgamma / dbeta -> workspace; _params: dgamma / dbeta
 * from those partial sums.  The second step is a weight-gradient computation: the caller may run it on the stream that carries the weight
 * gradients, behind the first, provided `workspace` is left alone until it has run (umr_layernorm_bwd = both on one stream). */
int umr_layernorm_bwd_rows(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                           const void* dres, void* dx, void* workspace, int64_t workspace_bytes, int M, int D, int dtype,
                           umr_stream_t stream);
int umr_layernorm_bwd_params(const void* workspace, int64_t workspace_bytes, float* dgamma, float* dbeta, int accumulate,
                             int M, int D, umr_stream_t stream);

/* ---- fused attention (timm Attention / F.scaled_dot_product_attention; scale head_dim^-0.5) ----
 * qkv [B*N, 3*heads*64] as produced by the qkv GEMM; out [B*N, heads*64]; lse f32 [B*heads*N].
 * bwd writes dqkv in the qkv layout; dsum_ws is scratch of umr_attention_bwd_workspace(B, N, heads) bytes
 * (per batch-head: -lse*log2(e) and rowsum(dO*O), rows padded to 64 queries). head_dim must be 64. */
int64_t umr_attention_bwd_workspace(int B, int N, int heads);
int umr_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int heads, int head_dim, int dtype,
                      umr_stream_t stream);
int umr_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dsum_ws, void* dqkv,
                      int B, int N, int heads, int head_dim, int dtype, umr_stream_t stream);

/* ---- data movement / small ops ------------------------------------------------------------
 * patchify: NCHW f32 image -> patch rows [B*gh*gw][ldk] (K order c,py,px = Conv2d weight order; vit.py:179)
 * bilinear: NHWC resize, align_corners as given (blocks.py:155-172,377-379; vit.py:157) and its exact adjoint
 * pixel_shuffle: ConvTranspose2d with kernel == stride as GEMM + scatter (vit.py:270-302); inverse gathers
 * zero_stuff2: scatter for the data gradient of the stride-2 3x3 conv (vit.py:329-335)
 * permute4: weight packing / unpacking (4-D permutation with arbitrary, possibly negative, source strides)
 * segsum / fill_cls / cast: reductions over images, class-token row init (vit.py:188-193), dtype casts */
int umr_patchify(const float* images, void* out, int B, int H, int W, int patch, int ldk, int dtype, umr_stream_t stream);
int umr_bilinear_fwd(const void* x, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, int dtype, umr_stream_t stream);
int umr_bilinear_bwd(const void* dy, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners, int dtype, umr_stream_t stream);
/* The same resizes on maps that are column slices of wider row-major buffers: ldx / ldy (lddy / lddx) = elements between consecutive
 * pixels, 0 = C.  flags (forward): UMR_BILINEAR_RELU = max(., 0) on the result (a ReLU that follows the resize: the heads' first
 * layer runs BEFORE the final x2 interpolation of models.py:70-72 -- a 1x1 convolution and a bilinear resize commute exactly -- and
 * objectness_net.py:111's ReLU after it); UMR_BILINEAR_OUT_X3 (dtype f32, C and ldy multiples of 8) = the result as three bf16
 * planes per pixel [h(C) | m(C) | l(C)], ldy in bf16 elements (>= 3 C): the operand format of the UMR_BF16X3 GEMMs. */
#define UMR_BILINEAR_RELU 1
#define UMR_BILINEAR_OUT_X3 2
int umr_bilinear_fwd_ex(const void* x, int64_t ldx, void* y, int64_t ldy, int B, int Hi, int Wi, int Ho, int Wo, int C, int align_corners,
                        int flags, int dtype, umr_stream_t stream);
int umr_bilinear_bwd_ex(const void* dy, int64_t lddy, void* dx, int64_t lddx, int B, int Hi, int Wi, int Ho, int Wo, int C,
                        int align_corners, int dtype, umr_stream_t stream);
int umr_pixel_shuffle(const void* src, void* dst, int B, int H, int W, int s, int C, int inverse, int dtype, umr_stream_t stream);
int umr_zero_stuff2(const void* dy, void* out, int B, int H, int W, int Ho, int Wo, int C, int dtype, umr_stream_t stream);
int umr_permute4(const void* src, void* dst, const int32_t* dst_dims, const int64_t* src_strides, int64_t src_offset,
                 int dtype_in, int dtype_out, int accumulate, umr_stream_t stream);
/* many permutes in one launch (the per-step refresh of the kernel-layout weight copies after the optimizer step).
 * table_dev: n entries IN DEVICE MEMORY, sorted by blk_start; entry e covers blocks [blk_start_e, blk_start_{e+1}), blk_start_0 = 0;
 * total_blocks = the sum.  Same element semantics as umr_permute4 (dst[i0,i1,i2,i3] = src[soff + sum i_k * sstride[k]], cast to
 * dtype_out; no accumulate).  Block shape per entry: e[3] == 0 -> linear, ceil(elements / 8192) blocks of 8192 consecutive
 * destination elements; e[3] == -1 -> 2-D transpose (needs d[0] == d[1] == 1, sstride[2] == 1): ceil(d[2] / 64) * ceil(d[3] / 64)
 * blocks of 64 x 64, tile coordinate of dimension 3 fastest; otherwise TILED: a block handles the hyper-rectangle e[0] x e[1] x e[2] x e[3] (e[0] e[1] e[2] (e[3] + 1) <= 4608) of the
 * destination index space, read in source-address order (ord[] = the four dimensions by ascending |sstride|) through LDS --
 * prod_k ceil(d[k] / e[k]) blocks, tile coordinate of dimension 3 fastest.  Use tiles when the innermost destination dimension
 * is strided in the source (transposes). */
typedef struct umr_perm_entry {
    const void* src; void* dst; int32_t d[4]; int64_t sstride[4]; int64_t soff; int32_t dtype_in, dtype_out; int64_t blk_start;
    int32_t e[4]; int32_t ord[4];
    int64_t rowlen;   /* dtype_out == UMR_BF16X3: the destination is [rows][h(rowlen) | m(rowlen) | l(rowlen)] bf16 planes of the f32 values,
                         row = destination element index / rowlen (the weights of the fp32-grade plane GEMMs); otherwise ignored */
} umr_perm_entry;
/* blk_entry_dev: optional int32[total_blocks] in device memory, the entry index of every block (saves the per-block search) */
int umr_permute4_batched(const umr_perm_entry* table_dev, int n, int64_t total_blocks, const int32_t* blk_entry_dev, umr_stream_t stream);
int umr_segsum(const void* x, void* out, int R, int reps, int64_t rep_stride, int64_t seg_stride, int C, int dtype_in,
               int out_f32, int accumulate, umr_stream_t stream);
int umr_fill_cls(void* tokens, const float* cls, const float* pos0, int B, int64_t batch_stride, int D, int dtype, umr_stream_t stream);
int umr_cast(const void* src, void* dst, int64_t n, float scale, int dtype_in, int dtype_out, umr_stream_t stream);
/* dst = src * (*scalar), scalar on the device: chains the incoming gradient of the scalar loss (loss.backward(), train_objectness_net.py:259)
 * into the loss kernel's gradient maps without a host read of that scalar */
int umr_scale_by_device_scalar(const float* src, const float* scalar, float* dst, int64_t n, umr_stream_t stream);

/* ---- head output layer 1024 -> {1,2} (+tanh / sine=4), NCHW f32 output (objectness_net.py:116,133-134) ---- */
int umr_head_out_fwd(const void* h, const float* w, const float* bias, float* out, int64_t M, int K, int Cout, int HW, int act,
                     int dtype, umr_stream_t stream);
/* second half of the same layer when its dot products were folded into the producing GEMM (umr_gemm_desc.red_*):
 * out = act(bias + sum_t partials[t][m][c]), tiles added in index order (bitwise reproducible) */
int umr_head_out_finish(const float* partials, int nparts, const float* bias, float* out, int64_t M, int Cout, int HW, int act,
                        umr_stream_t stream);
int64_t umr_head_out_bwd_workspace(int64_t M, int K);
int umr_head_out_bwd(const void* h, const float* w, const float* dout, const float* yout, void* dh, float* dw, float* db,
                     void* workspace, int64_t workspace_bytes, int64_t M, int K, int Cout, int HW, int act, int relu_mask,
                     int dtype, umr_stream_t stream);

/* ---- fused 4-term loss, value + gradient (train_objectness_net.py:215-254) -----------------
 * all maps NCHW f32; out5 = [total, center, sdf, sdf-gradient, bce]; d_center / d_sdf may be NULL (value only). */
int64_t umr_loss_workspace(void);
int umr_objectness_loss(const float* pred_center, const float* pred_sdf, const float* gt_center, const float* gt_sdf,
                        const float* gt_saliency, float* d_center, float* d_sdf, float* out5, void* workspace, int B, int H,
                        int W, int center_l2, int sdf_l2, int use_grad, int use_bce, float grad_scale, umr_stream_t stream);

/* ---- Adam on a flat f32 parameter buffer (torch.optim.Adam defaults; train_objectness_net.py:96,260) ---- */
int umr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  int step, float grad_scale, umr_stream_t stream);
/* The same update with its scalars in DEVICE memory, for a train step captured in a HIP graph and replayed with a new learning
 * rate / step count each iteration (the per-iteration MultiStepLR of train_objectness_net.py:261 changes lr between replays).
 * umr_adam_set_hyper writes hyper7 = [lr, beta1, beta2, eps, 1 - beta1^step, sqrt(1 - beta2^step), grad_scale] (one tiny launch,
 * arguments by value: no host buffer has to outlive the call); umr_adam_step_hyper is bit-identical to umr_adam_step with the
 * same arguments. */
int umr_adam_set_hyper(float* hyper7_dev, float lr, float beta1, float beta2, float eps, int step, float grad_scale, umr_stream_t stream);
int umr_adam_step_hyper(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper7_dev, umr_stream_t stream);
/* The optimizer step of a whole STAGE of the flat parameter buffer in one launch, writing the kernel-layout bf16 copies of its Linear
 * weights itself (no refresh pass that re-reads the f32 weights: -8 B per parameter and copy pair).  table_dev[i]:
 *   N == 0  a plain range: p / g / m / v of n elements, updated exactly as umr_adam_step_hyper does (4096 elements per block);
 *   N  > 0  a 2-D weight [N][K] (f32, contiguous; K % 4 == 0, N % 8 == 0, 16-byte aligned), walked in 64 x 64 tiles
 *           (ceil(N/64) * ceil(K/64) blocks, k fastest); dst_lin (may be NULL) receives bf16(p) as [N][K], dst_t (may be NULL) as [K][N].
 * blk_start = the first block of the entry, blk_entry_dev[b] = the entry of block b (int32[total_blocks]).  The update is the same
 * expression as umr_adam_step_hyper's, the copies round as umr_cast does: weights and copies are bit-identical to
 * umr_adam_step_hyper followed by umr_permute4_batched. */
typedef struct umr_adam_pack_entry {
    float* p; const float* g; float* m; float* v; void* dst_lin; void* dst_t; int64_t n; int32_t N, K; int64_t blk_start;
} umr_adam_pack_entry;
int umr_adam_pack_step(const umr_adam_pack_entry* table_dev, int n_entries, int64_t total_blocks, const int32_t* blk_entry_dev,
                       const float* hyper7_dev, umr_stream_t stream);

/* ---- object-reasoning glue around the net ("next" rows f1/f2) -----------------------
 * crop_resize: proposal crops [x1,y1,x2,y2) of a [3,H,W] f32 image -> [N,3,S,S], bilinear, no antialias
 *   (object_reasoning.py:311-323,398-410: torchvision Resize on tensors == F.interpolate(align_corners=False)).
 * center_peaks: union mask (sigmoid(sdf)>0.5 | ||center||>0.5), `erode_rounds` x (k x k) erosion, float64 5x5
 *   anti-centre correlation / 24, zero `border`, per-map max and FIRST flat argmax
 *   (object_reasoning.py:360-377,528-550; utils/misc.py:10-20).  filter50 = the 2x5x5 float64 taps (device).
 *   score_out may be NULL.  Maps up to H*W = 76,800 pixels.
 * boundary_deltas: [delta_x1, delta_y1, delta_x2, delta_y2] per map (object_reasoning.py:139-174). */
int umr_crop_resize_bilinear(const float* image, const int32_t* boxes, float* out, int N, int H, int W, int S, umr_stream_t stream);
int umr_center_peaks(const float* sdf_maps, const float* center_fields, const double* filter50, double* score_out,
                     double* max_values, int64_t* argmax, int B, int H, int W, int border, int erode_kernel, int erode_rounds,
                     umr_stream_t stream);
/* center_peaks with an argmax certificate: the same max / first flat argmax, plus certified[b] = 1 when the argmax is PROVABLY the
 * one that ANY pair of fields within `eps` (max-norm) of these would give -- the peak survives in erode(union & ~F), beats every
 * other pixel of erode(union | F) by more than 2 sqrt(2) eps (F = the pixels whose mask decision such a perturbation can flip) and
 * stays on its side of `singular_threshold` (object_reasoning.py:541); or, for a map without a positive score, no pixel of the
 * largest mask can reach one.  Lets a sweep run in a cheaper arithmetic mode and re-run only what it cannot certify
 * (unmore_amd/reasoning.py::sweep_proposals; the argument is oracle/objectness_oracle.py::peak_certificate's).  Maps up to
 * H*W = 25,600 pixels (six mask planes in LDS). */
int umr_center_peaks_certified(const float* sdf_maps, const float* center_fields, const double* filter50, double* max_values,
                               int64_t* argmax, int32_t* certified, int B, int H, int W, int border, int erode_kernel, int erode_rounds,
                               float eps, double singular_threshold, umr_stream_t stream);
int umr_boundary_deltas(const float* sdf_maps, float* deltas, int B, int H, int W, umr_stream_t stream);
/* nms: torchvision.ops.nms as object_reasoning.py:661 calls it -- `order` = the boxes' indices by descending score (the caller sorts;
 * equal scores -- the reference passes its labels, all ones -- keep their input order), greedy: a box is kept unless an earlier kept one
 * overlaps it with IoU > iou_threshold (f32: inter / (area_a + area_b - inter)).  keep[0 .. *n_keep) = the kept boxes' indices in rank
 * order; keep has room for n, n_keep is one int32 on the device.  n <= 4096; workspace: umr_nms_workspace(n) bytes, 8-byte aligned. */
/* object scoring (object_scoring.py:172-272).  Per proposal, from its S x S fields: the two binary masks (||center|| > 0.5,
 * sigmoid(sdf) > 0.5), each resized to the proposal's box (integer corners, already clipped to the image) as torchvision's Resize
 * does for integer tensors -- bilinear, align_corners=False, then round half to even -- and OR-ed (:196-228).
 * mask_paste_stats: stats[n] = {x_min, y_min, x_max + 1, y_max + 1, area} of that pasted union mask in image coordinates (the tight
 *   box of :231-235; all zero for an empty mask), maxima[n] = {max ||center||, max sdf} over the crop (:189-193).  Nothing image-sized
 *   is written.
 * mask_paste: the pasted union masks of the K selected proposals, masks [K,H,W] u8 (what survives NMS, :238-240). */
int umr_mask_paste_stats(const float* sdf_maps, const float* center_fields, const int32_t* boxes, int N, int S, int H, int W,
                         int32_t* stats, float* maxima, umr_stream_t stream);
int umr_mask_paste(const float* sdf_maps, const float* center_fields, const int32_t* boxes, const int64_t* select, int K, int S, int H, int W,
                   uint8_t* masks, umr_stream_t stream);
/* mask_components: 8-connected components of every map's union mask (sigmoid(sdf) > 0.5 | ||center|| > 0.5) in scipy.ndimage.label's
 * order (by first pixel in raster order) -- object_reasoning.py:206-257, the --analyze_cc branch of center_reasoning (README.md:176).
 * counts[b] = the number of components of map b; boxes[b][i] = [x1, y1, x2, y2) of component i for i < min(counts[b], max_components),
 * zeros beyond.  S * S * 8 + max_components * 16 bytes of LDS (<= 150 KiB). */
int umr_mask_components(const float* sdf_maps, const float* center_fields, int B, int S, int max_components, int32_t* counts, int32_t* boxes,
                        umr_stream_t stream);
int64_t umr_nms_workspace(int n);
int umr_nms(const float* boxes, const int64_t* order, int n, float iou_threshold, void* workspace, int64_t workspace_bytes,
            int64_t* keep, int32_t* n_keep, umr_stream_t stream);

/* ---- collapsed linear head (opt-in; SURVEY.md section 7 "the sdf head has no nonlinearity before tanh") ------
 * A head whose four convolutions have no activation between them (objectness_net.py:119-142) equals ONE 3x3 conv
 * C -> 1 plus a border-dependent bias: out(p) = act( sum_{taps t inside the image at p} (kw[t].x(p+t) + tapbias10[t])
 * + tapbias10[9] ).  x is NHWC [B,H,W,C=256]; out / dout / yout are [B,1,H,W] f32; kw is [9][C] f32.
 * bwd_data: dx (+)= sum_t kw[t] * g(p-t), g = dout*act'(yout).  bwd_weight: out = [G(9*C) | n(9) | D(1)] with
 * G[t] = sum_p g(p) x(p+t), n[t] = sum_{p: p+t inside} g(p), D = sum_p g(p): all the factored weights' gradients
 * follow from these by small matrix products (umr_small_gemm_f32).  act: UMR_ACT_NONE / UMR_ACT_TANH (4 = sine, fwd only). */
int umr_linear_head_fwd(const void* x, const float* kw, const float* tapbias10, float* out, int B, int H, int W, int C, int act,
                        int dtype, umr_stream_t stream);
int umr_linear_head_bwd_data(const float* dout, const float* yout, const float* kw, void* dx, int B, int H, int W, int C, int act,
                             int accumulate, int dtype, umr_stream_t stream);
int64_t umr_linear_head_bwd_weight_workspace(int64_t M, int C);
int umr_linear_head_bwd_weight(const void* x, const float* dout, const float* yout, float* out, void* workspace,
                               int64_t workspace_bytes, int B, int H, int W, int C, int act, int dtype, umr_stream_t stream);
/* shift9: s9[q][t] = g(q - off_t) (t < 9; zero outside the image and for 9 <= t < 16), a [B*H*W][16] map of `dtype`; nd[16] f32 =
 * n[0..8], D, zeros.  G[t] = sum_q s9[q][t] x(q) and dx(q) = sum_t s9[q][t] kw[t] are then ordinary TN / NT products -- and when x is
 * a bilinear resize of a smaller map (models.py:70-72), products on THAT map after umr_bilinear_bwd_ex of s9 (16 channels). */
int64_t umr_linear_head_shift9_workspace(int64_t M);
int umr_linear_head_shift9(const float* dout, const float* yout, void* s9, float* nd, void* workspace, int64_t workspace_bytes,
                           int B, int H, int W, int act, int dtype, umr_stream_t stream);
/* gather9 (forward mirror of shift9): with taps[q][t] = kw[t] . x(q) given as a [B*H*W] map of >= 9 f32 channels (pixel stride ldt) --
 * for x = resize(y) that is the resize of the 16-column product y kw^T taken on the small map --
 * out(q) = act( sum_{t: q + off_t inside the image} (taps[q + off_t][t] + tapbias10[t]) + tapbias10[9] ), out [B,1,H,W] f32. */
int umr_linear_head_gather9(const float* taps, int64_t ldt, const float* tapbias10, float* out, int B, int H, int W, int act,
                            umr_stream_t stream);
/* C[i*sc_m + j*sc_n] (=|+=) sum_k A[i*sa_m + k*sa_k] * B[k*sb_k + j*sb_n]  (tiny f32 products, arbitrary strides) */
int umr_small_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t sa_m, int64_t sa_k, int64_t sb_k,
                       int64_t sb_n, int64_t sc_m, int64_t sc_n, int accumulate, umr_stream_t stream);

/* ---- existence classifier: torchvision ResNet-50 + Linear(1000,1) + sigmoid in eval mode
 * (models/objectness_net.py:205-223; callers object_reasoning.py:491-523, object_scoring.py:123-140).  Convolutions and
 * Linear layers run on umr_gemm_nt; these are the remaining pieces.
 * im2col_nchw: NCHW f32 image -> rows [B*Ho*Wo][ldk], K order (c, ky, kx) (= Conv2d weight order), tail zero: the 7x7 s2 p3 stem
 * maxpool3x3s2: nn.MaxPool2d(kernel 3, stride 2, padding 1), NHWC
 * bn_fold: eval-mode BatchNorm2d folded into the preceding conv's packed weight rows: w_out[co][k] = w[co][k]*s, b_out[co] =
 *          beta - mean*s, s = gamma / sqrt(var + eps) */
int umr_im2col_nchw(const float* images, void* out, int B, int C, int H, int W, int KH, int KW, int stride, int pad, int ldk,
                    int dtype, umr_stream_t stream);
int umr_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, umr_stream_t stream);
int umr_bn_fold(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps, void* w_out,
                float* b_out, int Co, int K, int ldk, int dtype, umr_stream_t stream);

/* ---- ground-truth synthesis on the device (datasets.py:158-159,171-222 without random crop) -------
 * mask [B,H,W] u8 (non-zero = object) at the training resolution; center_xy [B,2] f32 (x, y) object centres in the same
 * pixel coordinates, or NULL = bounding-box centre of each mask ((min+max)/2, datasets.py:158-159).
 * sdf = DT(mask)/max - DT(1-mask)/max (second term only with use_bg_sdf) where DT is cv2.distanceTransform(u8, DIST_L2, 3)
 * (OpenCV's 3x3 chamfer in 16.16 fixed point, a = 0.955, b = 1.3693); center_field [B,2,H,W] = normalize(mask *
 * normalize((i - c_y, j - c_x))); saliency = mask > 0.  Empty masks give all-zero labels (datasets.py:128-138). */
int64_t umr_label_synthesis_workspace(int B, int H, int W);
int umr_label_synthesis(const uint8_t* mask, const float* center_xy, float* center_field, float* saliency, float* sdf,
                        void* workspace, int64_t workspace_bytes, int B, int H, int W, int use_bg_sdf, umr_stream_t stream);

/* ---- the random-crop branch of the training item (datasets.py:144-145,161-190: the default training path, :109) ----
 * crop_resize_batch: src [B,C,H,W] with one box [x1,y1,x2,y2) per item -> dst [B,C,Ho,Wo]; f32 bilinear (torchvision tensor
 *   Resize without antialias == F.interpolate(align_corners=False), :99,103) or, with nearest_u8, u8 nearest
 *   (F.interpolate(mode="nearest"), :100,104).  Serves transforms.Resize (box = whole image) and crop + Resize.
 * distance_transform: cv2.distanceTransform(u8, DIST_L2, 3) of each mask, divided by its maximum when normalize (:162-163).
 * label_synthesis_cropped: as label_synthesis, but the foreground field was computed before the crop and arrives resized
 *   (fg_sdf [B,H,W]), the object centres are given, only the background transform is taken from `mask`, and an empty
 *   mask is not short-circuited (the reference tests emptiness before the crop, :146-157). */
int umr_crop_resize_batch(const void* src, const int32_t* boxes, void* dst, int B, int C, int H, int W, int Ho, int Wo,
                          int nearest_u8, umr_stream_t stream);
int64_t umr_distance_transform_workspace(int B, int H, int W);
int umr_distance_transform(const uint8_t* mask, float* out, void* workspace, int64_t workspace_bytes, int B, int H, int W,
                           int normalize, umr_stream_t stream);
int umr_label_synthesis_cropped(const uint8_t* mask, const float* center_xy, const float* fg_sdf, float* center_field,
                                float* saliency, float* sdf, void* workspace, int64_t workspace_bytes, int B, int H, int W,
                                int use_bg_sdf, umr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
