"""Thin tensor-level wrappers over the C-ABI (include/umr.h).

PyTorch here is plumbing only: it owns device memory and the current HIP stream;
every computation below is a hand-written gfx950 kernel in unmore_amd/csrc.
All wrappers raise if the tensor is not on a GPU or the library is missing --
there is no CPU / eager fallback."""
import ctypes

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("unmore_amd ops need GPU tensors (no CPU fallback in the product path)")


def _rowmajor2d(t):
    assert t.stride(-1) == 1, "innermost dimension must be contiguous"
    return t


_ws_cache = {}


def _workspace(nbytes, device):
    key = (device.index,)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def gemm_nt(A, B, bias=None, *, out=None, out2=None, aux=None, aux2=None, rowbias=None, rows_per_batch=0,
            act=L.ACT_NONE, mask_relu=False, mask_dgelu=False, c2_mode=0, out_f32=False,
            conv=0, conv_geom=None, M=None, lda=None):
    """C[M,N] = epi(A[M,K] . B[N,K]^T).  A: [M,K] (2-D, row stride lda) or NHWC
    [nb,H,W,Cin] when conv != 0 (B then is [N, 9*Cin] packed (ky,kx,ci))."""
    _need_gpu(A, B)
    dt = _DT[A.dtype]
    assert B.dtype == A.dtype
    d = L.GemmDesc()
    N, K = B.shape[0], B.shape[1]
    if conv:
        nb, H, W, Cin = A.shape
        assert A.is_contiguous() and K == 9 * Cin
        s = 2 if conv == 2 else 1
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        M_ = nb * Ho * Wo
        d.nb, d.H, d.W, d.Cin, d.Ho, d.Wo = nb, H, W, Cin, Ho, Wo
        d.lda = Cin
    else:
        A2 = A if A.dim() == 2 else A.reshape(-1, A.shape[-1])
        M_ = A2.shape[0] if M is None else M
        d.lda = A2.stride(0) if lda is None else lda
        assert A2.stride(1) == 1
    odt = torch.float32 if out_f32 else A.dtype
    if out is None:
        out = torch.empty((M_, N), dtype=odt, device=A.device)
    assert out.dtype == odt and out.stride(-1) == 1
    flags = 0
    if bias is not None:
        assert bias.dtype == torch.float32
        flags |= L.EPI_BIAS
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.is_contiguous()
        flags |= L.EPI_ROWBIAS
    if aux is not None:
        assert aux.dtype == A.dtype
        flags |= L.EPI_MASK_RELU if mask_relu else (L.EPI_MASK_DGELU if mask_dgelu else L.EPI_ADD_AUX)
    if aux2 is not None:
        assert aux2.dtype == A.dtype
        flags |= L.EPI_ADD_AUX2
    if out_f32:
        flags |= L.EPI_OUT_F32
    if c2_mode:
        if out2 is None:
            out2 = torch.empty((M_, N), dtype=A.dtype, device=A.device)
        assert out2.dtype == A.dtype
    o2d = out.reshape(-1, N) if out.is_contiguous() else out
    d.A, d.B, d.C, d.C2 = _p(A), _p(B), _p(out), _p(out2)
    d.bias, d.aux, d.aux2, d.rowbias = _p(bias), _p(aux), _p(aux2), _p(rowbias)
    d.ldb = B.stride(0)
    d.ldc = o2d.stride(0) if o2d.dim() == 2 else N
    d.ldc2 = (out2.reshape(-1, N).stride(0) if out2 is not None else 0)
    d.ldaux = (aux.reshape(-1, N).stride(0) if aux is not None else 0)
    d.ldaux2 = (aux2.reshape(-1, N).stride(0) if aux2 is not None else 0)
    d.M, d.N, d.K, d.dtype = M_, N, K, dt
    d.flags, d.act, d.c2_mode, d.rows_per_batch, d.conv = flags, act, c2_mode, rows_per_batch, conv
    L.check(L.lib().umr_gemm_nt(ctypes.byref(d), _stream()), "umr_gemm_nt")
    return (out, out2) if c2_mode else out


def gemm_tn(dY, X, *, dW=None, dbias=None, accumulate=False, conv=0):
    """dW[N,K] f32 = sum_m dY[m,N]^T X[m,K]; X is NHWC [nb,H,W,Cin] when conv != 0
    (dY then is [nb*Ho*Wo, N] and dW is [N, 9*Cin] packed (ky,kx,ci))."""
    _need_gpu(dY, X)
    dt = _DT[X.dtype]
    assert dY.dtype == X.dtype
    d = L.GemmTnDesc()
    dY2 = dY.reshape(-1, dY.shape[-1])
    M, N = dY2.shape
    if conv:
        nb, H, W, Cin = X.shape
        assert X.is_contiguous()
        s = 2 if conv == 2 else 1
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        assert M == nb * Ho * Wo
        K = 9 * Cin
        d.nb, d.H, d.W, d.Cin, d.Ho, d.Wo = nb, H, W, Cin, Ho, Wo
        d.ldx = Cin
    else:
        X2 = X.reshape(-1, X.shape[-1])
        assert X2.shape[0] == M
        K = X2.shape[1]
        d.ldx = X2.stride(0)
    if dW is None:
        dW = torch.empty((N, K), dtype=torch.float32, device=X.device)
        assert not accumulate
    assert dW.dtype == torch.float32 and dW.stride(-1) == 1
    if dbias is not None:
        assert dbias.dtype == torch.float32 and dbias.numel() == N
    d.dY, d.X, d.dW, d.dbias = _p(dY2), _p(X), _p(dW), _p(dbias)
    d.lddy, d.lddw = dY2.stride(0), (dW.stride(0) if dW.dim() == 2 else K)
    d.M, d.N, d.K, d.dtype, d.accumulate, d.conv = M, N, K, dt, int(accumulate), conv
    need = L.lib().umr_gemm_tn_workspace(ctypes.byref(d))
    ws = _workspace(need, X.device)
    d.workspace, d.workspace_bytes = _p(ws), ws.numel()
    L.check(L.lib().umr_gemm_tn(ctypes.byref(d), _stream()), "umr_gemm_tn")
    return dW
