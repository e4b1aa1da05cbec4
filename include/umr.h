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
enum umr_dtype { UMR_F32 = 0, UMR_BF16 = 1 };

int umr_version(void);
const char* umr_last_error_string(void);

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
 *   c2_mode==1: C2 = relu(v). */
enum umr_epi_flags {
    UMR_EPI_BIAS = 1, UMR_EPI_ADD_AUX = 2, UMR_EPI_MASK_RELU = 4, UMR_EPI_MASK_DGELU = 8,
    UMR_EPI_ADD_AUX2 = 16, UMR_EPI_OUT_F32 = 32, UMR_EPI_ROWBIAS = 64
};
enum umr_act { UMR_ACT_NONE = 0, UMR_ACT_RELU = 1, UMR_ACT_GELU = 2, UMR_ACT_TANH = 3 };

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
} umr_gemm_desc;

int umr_gemm_nt(const umr_gemm_desc* d, umr_stream_t stream);

/* ---- weight-gradient GEMM "TN" (reduction over rows) -----------------------
 * dW[N,K] (f32) = sum_m dY[m,N]^T . X[m,K]   (X rows shifted per tap when conv != 0,
 * K = 9*Cin, dW laid out [N][3][3][Cin]).  Replaces the autograd weight
 * gradients of the modules listed above (train_objectness_net.py:259).
 * Split over `splits` row ranges into workspace slabs [splits][N][K] f32, then
 * reduced in fixed order (bitwise reproducible).  workspace_bytes must be
 * >= umr_gemm_tn_workspace(d). */
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
} umr_gemm_tn_desc;

int64_t umr_gemm_tn_workspace(const umr_gemm_tn_desc* d);
int umr_gemm_tn(const umr_gemm_tn_desc* d, umr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
