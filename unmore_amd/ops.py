"""Thin tensor-level wrappers over the C-ABI (include/umr.h).

PyTorch here is plumbing only: it owns device memory and the current HIP stream;
every computation below is a hand-written gfx950 kernel in unmore_amd/csrc.
All wrappers raise if the tensor is not on a GPU or the library is missing --
there is no CPU / eager fallback."""
import collections
import ctypes

import torch

from . import _lib as L
from . import graphs

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16}


# the current device / stream as raw handles: torch.cuda.current_stream() builds a Stream object through several Python layers
# (~6 us), and a step asks for it ~1500 times -- a third of the host time of the reference recipe's step
_cur_dev = torch._C._cuda_getDevice
_raw_stream = torch._C._cuda_getCurrentRawStream


def _stream_id(device_index=None):
    return _raw_stream(_cur_dev() if device_index is None else device_index)


def _stream():
    return ctypes.c_void_p(_raw_stream(_cur_dev()))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("unmore_amd ops need GPU tensors (no CPU fallback in the product path)")


def _rowmajor2d(t):
    assert t.stride(-1) == 1, "innermost dimension must be contiguous"
    return t


def set_f32_mode(mode):
    """How fp32 (parity-mode) GEMMs form their products (include/umr.h, umr_set_f32_mode): 'x3' (default) = three-way bf16
    splits on the bf16 matrix cores, fp32-grade for finite operands; 'exact' = the f32 MFMA (IEEE behaviour for inf / huge /
    denormal operands); 'x3_fast' (opt-in) = as 'x3' with three instead of six terms in the plane GEMMs (products to 2^-16; inference
    under a 1e-4 tolerance).  Process-wide.  Returns the previous mode."""
    prev = get_f32_mode()
    L.check(L.lib().umr_set_f32_mode({"exact": L.F32_EXACT, "x3": L.F32_X3, "x3_fast": L.F32_X3_FAST}[mode]), "umr_set_f32_mode")
    return prev


def get_f32_mode():
    return {L.F32_EXACT: "exact", L.F32_X3: "x3", L.F32_X3_FAST: "x3_fast"}[L.lib().umr_get_f32_mode()]


def set_cu_budget(cus):
    """CUs the persistent GEMM grids occupy (include/umr.h, umr_set_cu_budget): 0 = all; n = exactly min(n, CUs) workgroups, one per
    CU, the rest left to kernels that run beside them (RCCL's bucket all-reduces during backward).  Bit-identical results.
    Returns the previous budget."""
    prev = int(L.lib().umr_get_cu_budget())
    L.check(L.lib().umr_set_cu_budget(int(cus)), "umr_set_cu_budget")
    return prev


def set_debug_option(name, value):
    """A debug / A-B option of the library (include/umr.h, umr_set_debug_option; names: csrc/umr_common.h): value = the text the
    environment variable of that name would hold, None = unset.  The library reads the environment once, when it is loaded; this is
    how tests and probes switch an option between launches.  Returns the previous value (int, or None if it was unset)."""
    import ctypes
    v, isset = ctypes.c_int(0), ctypes.c_int(0)
    L.check(L.lib().umr_get_debug_option(name.encode(), ctypes.byref(v), ctypes.byref(isset)), "umr_get_debug_option")
    L.check(L.lib().umr_set_debug_option(name.encode(), None if value is None else str(value).encode()), "umr_set_debug_option")
    return v.value if isset.value else None


def splitk_bad_tickets(device=None):
    """Diagnostic (synchronises the current stream): out-of-range split-K tickets counted on the current stream's GEMM workspace
    since the last call (include/umr.h, umr_gemm_nt_ws_status); 0 = every split GEMM wrote all of its tiles."""
    import ctypes
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    ws = _splitk_workspace(dev)
    n = ctypes.c_int(0)
    L.check(L.lib().umr_gemm_nt_ws_status(ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(_stream_id(dev.index)), ctypes.byref(n)), "umr_gemm_nt_ws_status")
    return n.value


_ws_cache = {}


_WS_MAX_ENTRIES = 8


def _workspace(nbytes, device):
    # one scratch buffer per (device, stream): kernels of different streams may run concurrently (reasoning.sweep_proposals).
    # Streams come and go (their raw handles are the keys), so the cache is a small LRU: an evicted buffer goes back to the
    # allocator pool of the stream it was allocated (and only ever used) on, which orders its reuse after the kernels that
    # scribbled on it.
    if graphs.capturing():
        # a captured graph owns its scratch (allocated from the capture's private pool, kept alive by the Captured object): two
        # graphs recorded on the same internal capture stream may be replayed on different streams at the same time
        # ... and one PER STREAM of the capture: the weight-gradient branch of a train step (engine.WgradStream) may run beside the
        # main branch in a replay, as it does eagerly.  A buffer that was outgrown stays alive with the capture: launches recorded
        # earlier still point at it, and the pool must not hand it to a tensor of the other branch.
        store = graphs.capture_store()
        key = ("ws", _stream_id(device.index), graphs.lane())
        buf = store.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                store.setdefault("outgrown", []).append(buf)
            buf = store[key] = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        return buf
    key = (device.index, _stream_id(device.index))
    buf = _ws_cache.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    _ws_cache[key] = buf   # re-inserted last = most recently used
    while len(_ws_cache) > _WS_MAX_ENTRIES:
        _ws_cache.pop(next(iter(_ws_cache)))
    return buf


_sk_cache = collections.OrderedDict()


def _splitk_workspace(device):
    """Per-(device, stream) scratch of umr_gemm_nt_ws: tile counters (zeroed once here; every launch leaves them zero) + slabs."""
    if graphs.capturing():
        store = graphs.capture_store()   # see _workspace; the zeroing of the counters is recorded too: every replay starts clean
        key = ("sk", _stream_id(device.index), graphs.lane())
        buf = store.get(key)
        if buf is None:
            buf = store[key] = torch.empty(int(L.lib().umr_gemm_nt_workspace()), dtype=torch.uint8, device=device)
            buf[:16384].zero_()
        return buf
    key = (device.index, _stream_id(device.index))
    buf = _sk_cache.pop(key, None)
    if buf is None:
        buf = torch.empty(int(L.lib().umr_gemm_nt_workspace()), dtype=torch.uint8, device=device)
        buf[:16384].zero_()
    _sk_cache[key] = buf
    while len(_sk_cache) > _WS_MAX_ENTRIES:
        _sk_cache.pop(next(iter(_sk_cache)))
    return buf


def gemm_nt(A, B, bias=None, *, out=None, out2=None, aux=None, aux2=None, rowbias=None, rows_per_batch=0,
            act=L.ACT_NONE, mask_relu=False, mask_dgelu=False, c2_mode=0, out_f32=False,
            conv=0, M=None, lda=None, a_remap=None, c_remap=None, aux_mod=0, red_w=None, no_store=False, query_rowreduce=False,
            query_splits=False, _stamps=None):
    """C[M,N] = epi(A[M,K] . B[N,K]^T).  A: [M,K] (2-D, row stride lda) or NHWC
    [nb,H,W,Cin] when conv != 0 (B then is [N, 9*Cin] packed (ky,kx,ci)).
    red_w ([c, N] f32, c in {1,2}): fused row reduction (umr_gemm_desc.red_*) -> returns (C, partials [ceil(N/64), M, c]);
    with no_store=True C is neither allocated nor written (returned as None).  query_rowreduce=True only asks the
    library whether this call would run on the path that implements the reduction (returns bool, launches nothing)."""
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
    skip_c = no_store or query_rowreduce or query_splits
    if out is None and not skip_c:
        out = torch.empty((M_, N), dtype=odt, device=A.device)
    assert skip_c or (out.dtype == odt and out.stride(-1) == 1)
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
    o2d = None if out is None else (out.reshape(-1, N) if out.is_contiguous() else out)
    d.A, d.B, d.C, d.C2 = _p(A), _p(B), _p(out), _p(out2)
    d.bias, d.aux, d.aux2, d.rowbias = _p(bias), _p(aux), _p(aux2), _p(rowbias)
    d.ldb = B.stride(0)
    d.ldc = o2d.stride(0) if (o2d is not None and o2d.dim() == 2) else N
    d.ldc2 = (out2.reshape(-1, N).stride(0) if out2 is not None else 0)
    d.ldaux = (aux.reshape(-1, N).stride(0) if aux is not None else 0)
    d.ldaux2 = (aux2.reshape(-1, N).stride(0) if aux2 is not None else 0)
    d.M, d.N, d.K, d.dtype = M_, N, K, dt
    d.flags, d.act, d.c2_mode, d.rows_per_batch, d.conv = flags, act, c2_mode, rows_per_batch, conv
    if a_remap is not None:
        d.a_rows_in, d.a_rows_out, d.a_row_off = a_remap
    if c_remap is not None:
        d.c_rows_in, d.c_rows_out, d.c_row_off = c_remap
    d.aux_mod = aux_mod
    if query_rowreduce:
        return bool(L.lib().umr_gemm_nt_rowreduce_ok(ctypes.byref(d)))
    if query_splits:   # how many K ranges this call would run as (1 = not split); launches nothing
        return int(L.lib().umr_gemm_nt_splits(ctypes.byref(d), int(L.lib().umr_gemm_nt_workspace()) if _wants_splitk_ws(d) else 0))
    partials = None
    if red_w is not None:
        assert red_w.dtype == torch.float32 and red_w.is_contiguous() and red_w.shape[1] == N and red_w.shape[0] in (1, 2)
        partials = torch.empty(((N + 63) // 64, M_, red_w.shape[0]), dtype=torch.float32, device=A.device)
        d.red_w, d.red_out, d.red_c = _p(red_w), _p(partials), red_w.shape[0]
    d.no_store = 1 if no_store else 0
    if _stamps is not None:   # instrumented library only (tools/probe/ts_probe.py): int64 [16, 8] cycle stamps
        d.rowbias, d.rows_per_batch = _p(_stamps), -9
    L.check(_timed_call(d), "umr_gemm_nt")
    if red_w is not None:
        return out, partials
    return (out, out2) if c2_mode else out


def split3(x, out=None, remap=None):
    """f32 [..., K] (rows contiguous) -> bf16 planes [..., 3K] = [h(K) | m(K) | l(K)] per row (include/umr.h, UMR_BF16X3).
    remap = (rows_in, rows_out, row_off) with a 2-D x: output row r is source row (r // rows_in) * rows_out + row_off + r % rows_in
    (the token rows without the class-token rows), rows = x.shape[0] // rows_out * rows_in."""
    _need_gpu(x)
    assert x.dtype == torch.float32 and x.stride(-1) == 1
    K = x.shape[-1]
    x2 = x.reshape(-1, K) if x.is_contiguous() else x
    assert x2.dim() == 2
    rows = x2.shape[0]
    ri, ro, off = (0, 0, 0)
    if remap is not None:
        ri, ro, off = remap
        assert rows % ro == 0 and off + ri <= ro
        rows = rows // ro * ri
        if out is None:
            out = torch.empty((rows, 3 * K), dtype=torch.bfloat16, device=x.device)
    if out is None:
        out = torch.empty(x.shape[:-1] + (3 * K,), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().umr_split3_rows(_p(x2), _p(out), rows, K, x2.stride(0), 3 * K, ri, ro, off, _stream()), "umr_split3")
    return out


def unsplit3(pl):
    """bf16 planes [..., 3K] -> f32 [..., K] (exact inverse of split3)"""
    _need_gpu(pl)
    assert pl.dtype == torch.bfloat16 and pl.is_contiguous() and pl.shape[-1] % 3 == 0
    K = pl.shape[-1] // 3
    out = torch.empty(pl.shape[:-1] + (K,), dtype=torch.float32, device=pl.device)
    rows = pl.numel() // (3 * K)
    L.check(L.lib().umr_unsplit3(_p(pl), _p(out), rows, K, 3 * K, K, _stream()), "umr_unsplit3")
    return out


def _x3_operand(t, N):
    """(pointer, row stride, is_planes) of an [M, N] epilogue operand given as f32 [M, N] or as bf16 planes [M, 3N]"""
    if t.dtype == torch.bfloat16:
        t2 = t.reshape(-1, 3 * N)
        assert t2.stride(1) == 1
        return t2, t2.stride(0), True
    assert t.dtype == torch.float32
    t2 = t.reshape(-1, N)
    assert t2.stride(1) == 1
    return t2, t2.stride(0), False


_x3_ws_cache = collections.OrderedDict()


def _x3_workspace(d, device):
    """split-K scratch of a plane GEMM: the 128x128 kernel's counters + slabs (umr_gemm_nt_workspace) followed by this problem's
    K-split slabs; one grow-only buffer per (device, stream), private to a capture in progress"""
    extra = int(L.lib().umr_gemm_nt_x3_workspace(ctypes.byref(d)))
    if extra == 0:
        return None
    need = int(L.lib().umr_gemm_nt_workspace()) + extra
    if graphs.capturing():
        store = graphs.capture_store()
        key = ("x3ws", _stream_id(device.index), graphs.lane())
        buf = store.get(key)
        if buf is None or buf.numel() < need:
            if buf is not None:
                store.setdefault("outgrown", []).append(buf)
            buf = store[key] = torch.empty(need, dtype=torch.uint8, device=device)
        return buf
    key = (device.index, _stream_id(device.index))
    buf = _x3_ws_cache.pop(key, None)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 64 << 20), dtype=torch.uint8, device=device)
    _x3_ws_cache[key] = buf
    while len(_x3_ws_cache) > _WS_MAX_ENTRIES:
        _x3_ws_cache.pop(next(iter(_x3_ws_cache)))
    return buf


def gemm_nt_x3(Ap, Bp, bias=None, *, act=L.ACT_NONE, conv=0, out_planes=False, out=None, red_w=None, mask=None, dgelu=None, aux=None,
               aux2=None, rowbias=None, rows_per_batch=0, c2_mode=0, c2_planes=False, c_remap=None, aux_mod=0):
    """fp32-grade C = epi(A . B^T) from operands held as three bf16 planes per f32 value (split3): Ap [M, 3K] (or NHWC
    [nb,H,W,3*Cin] when conv == 1), Bp [N, 3K].  Persistent 256x256 bf16 kernel, six plane pairs per K-tile (csrc/gemm_nt256p.hip,
    X3), split along K for small problems.  Epilogue as umr_gemm_desc (include/umr.h), all in f32: bias, rowbias, ONE of
    mask (keep where > 0) / dgelu (times GELU'(.)) / aux (add), aux2 (add), c2_mode 1 (second output = relu) / 2 (second output =
    pre-activation), act.  Every [M, N] operand may be f32 [M, N] or planes [M, 3N] (told apart by dtype); C is f32, or planes
    with out_planes=True; the second output is planes with c2_planes=True.  Returns C, or (C, C2), or the row-reduction
    partials [ceil(N/64), M, c] with red_w (inference form: C is not stored)."""
    _need_gpu(Ap, Bp)
    assert Ap.dtype == torch.bfloat16 and Bp.dtype == torch.bfloat16 and Bp.dim() == 2 and Bp.is_contiguous()
    d = L.GemmDesc()
    N, K = Bp.shape[0], Bp.shape[1] // 3
    if conv:
        assert conv == 1
        nb, H, W, C3 = Ap.shape
        Cin = C3 // 3
        assert Ap.is_contiguous() and K == 9 * Cin
        M_ = nb * H * W
        d.nb, d.H, d.W, d.Cin, d.Ho, d.Wo = nb, H, W, Cin, H, W
        d.lda = C3
    else:
        A2 = Ap if Ap.dim() == 2 else Ap.reshape(-1, Ap.shape[-1])
        assert A2.stride(1) == 1 and A2.shape[1] == 3 * K
        M_ = A2.shape[0]
        d.lda = A2.stride(0)
    rows_out = M_
    if c_remap is not None:
        d.c_rows_in, d.c_rows_out, d.c_row_off = c_remap
        rows_out = None   # the caller supplies `out`
    partials, out2 = None, None
    flags = 0
    if red_w is not None:
        # fused row reduction (the 1024 -> {1,2} output layer of a head): returns partials [ceil(N/64), M, c] for head_out_finish
        assert red_w.dtype == torch.float32 and red_w.is_contiguous() and red_w.shape[1] == N and red_w.shape[0] in (1, 2) and not conv
        partials = torch.empty(((N + 63) // 64, M_, red_w.shape[0]), dtype=torch.float32, device=Ap.device)
        d.red_w, d.red_out, d.red_c, d.no_store = _p(red_w), _p(partials), red_w.shape[0], 1
        out = None
    else:
        if out is None:
            assert rows_out is not None
            out = torch.empty((rows_out, 3 * N if out_planes else N), dtype=torch.bfloat16 if out_planes else torch.float32, device=Ap.device)
        else:
            out_planes = out.dtype == torch.bfloat16
        flags |= L.EPI_OUT_X3 if out_planes else L.EPI_OUT_F32
        o2 = out if out.dim() == 2 else out.reshape(-1, out.shape[-1])
        assert o2.stride(1) == 1 and o2.shape[1] == (3 * N if out_planes else N)
        d.ldc = o2.stride(0)
    if bias is not None:
        assert bias.dtype == torch.float32
        flags |= L.EPI_BIAS
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.is_contiguous() and rows_per_batch > 0
        flags |= L.EPI_ROWBIAS
        d.rowbias, d.rows_per_batch = _p(rowbias), rows_per_batch
    a_t, a_flag = None, 0
    for t, fl in ((mask, L.EPI_MASK_RELU), (dgelu, L.EPI_MASK_DGELU), (aux, L.EPI_ADD_AUX)):
        if t is not None:
            assert a_t is None, "one of mask / dgelu / aux"
            a_t, a_flag = t, fl
    keep = []
    if a_t is not None:
        t2, ld, pl = _x3_operand(a_t, N)
        flags |= a_flag | (L.EPI_AUX_X3 if pl else 0)
        d.aux, d.ldaux = _p(t2), ld
        keep.append(t2)
    if aux2 is not None:
        t2, ld, pl = _x3_operand(aux2, N)
        flags |= L.EPI_ADD_AUX2 | (L.EPI_AUX2_X3 if pl else 0)
        d.aux2, d.ldaux2 = _p(t2), ld
        keep.append(t2)
    if c2_mode:
        assert red_w is None and c_remap is None
        out2 = torch.empty((M_, 3 * N if c2_planes else N), dtype=torch.bfloat16 if c2_planes else torch.float32, device=Ap.device)
        flags |= L.EPI_C2_X3 if c2_planes else 0
        d.C2, d.ldc2, d.c2_mode = _p(out2), out2.stride(0), c2_mode
    d.aux_mod = aux_mod
    d.A, d.B, d.C, d.bias = _p(Ap), _p(Bp), _p(out), _p(bias)
    d.ldb = Bp.stride(0)
    d.M, d.N, d.K, d.dtype = M_, N, K, L.BF16X3
    d.flags, d.act, d.conv = flags, act, conv
    L.check(_timed_call(d), "umr_gemm_nt")
    if red_w is not None:
        return partials
    return (out, out2) if c2_mode else out


def gemm_tn(dY, X, *, dW=None, dbias=None, accumulate=False, conv=0, M=None, dy_remap=None, x_remap=None, lddy=None, ldx=None, x3=False):
    """dW[N,K] f32 = sum_m dY[m,N]^T X[m,K]; X is NHWC [nb,H,W,Cin] when conv != 0
    (dY then is [nb*Ho*Wo, N] and dW is [N, 9*Cin] packed (ky,kx,ci)).
    x3=True: both operands are f32 values as bf16 planes (split3): dY [M, 3N], X [M, 3K] / NHWC [nb,H,W,3*Cin] -- fp32-grade."""
    _need_gpu(dY, X)
    dt = L.BF16X3 if x3 else _DT[X.dtype]
    assert dY.dtype == X.dtype and (not x3 or X.dtype == torch.bfloat16)
    pl = 3 if x3 else 1
    d = L.GemmTnDesc()
    dY2 = dY.reshape(-1, dY.shape[-1]) if dY.is_contiguous() else dY
    N = dY2.shape[-1] // pl
    M = dY2.shape[0] if M is None else M
    if conv:
        nb, H, W, Cin = X.shape
        Cin //= pl
        assert X.is_contiguous()
        s = 2 if conv == 2 else 1
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        assert M == nb * Ho * Wo
        K = 9 * Cin
        d.nb, d.H, d.W, d.Cin, d.Ho, d.Wo = nb, H, W, Cin, Ho, Wo
        d.ldx = Cin * pl
    else:
        X2 = X.reshape(-1, X.shape[-1]) if X.is_contiguous() else X
        K = X2.shape[-1] // pl
        d.ldx = X2.stride(0) if ldx is None else ldx
    if dW is None:
        dW = torch.empty((N, K), dtype=torch.float32, device=X.device)
        assert not accumulate
    assert dW.dtype == torch.float32 and dW.stride(-1) == 1
    if dbias is not None:
        assert dbias.dtype == torch.float32 and dbias.numel() == N
    d.dY, d.X, d.dW, d.dbias = _p(dY2), _p(X), _p(dW), _p(dbias)
    d.lddy, d.lddw = (dY2.stride(0) if lddy is None else lddy), (dW.stride(0) if dW.dim() == 2 else K)
    if dy_remap is not None:
        d.dy_rows_in, d.dy_rows_out, d.dy_row_off = dy_remap
    if x_remap is not None:
        d.x_rows_in, d.x_rows_out, d.x_row_off = x_remap
    d.M, d.N, d.K, d.dtype, d.accumulate, d.conv = M, N, K, dt, int(accumulate), conv
    need = L.lib().umr_gemm_tn_workspace(ctypes.byref(d))
    ws = _workspace(need, X.device)
    d.workspace, d.workspace_bytes = _p(ws), ws.numel()
    L.check(L.lib().umr_gemm_tn(ctypes.byref(d), _stream()), "umr_gemm_tn")
    return dW


def layernorm_fwd(x, gamma, beta, eps=1e-6, planes=False):
    """planes=True (x f32): the normalised rows are written as bf16 planes [M, 3D] (the operand format of the plane GEMMs)"""
    _need_gpu(x)
    D = x.shape[-1]
    x2 = x.reshape(-1, D)
    M = x2.shape[0]
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    if planes:
        assert x.dtype == torch.float32
        y = torch.empty((M, 3 * D), dtype=torch.bfloat16, device=x.device)
        L.check(L.lib().umr_layernorm_fwd(_p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D, eps, L.BF16X3, _stream()),
                "umr_layernorm_fwd")
        return y, mean, rstd
    y = torch.empty_like(x2)
    L.check(L.lib().umr_layernorm_fwd(_p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), M, D, eps, _DT[x.dtype], _stream()),
            "umr_layernorm_fwd")
    return y.view(x.shape), mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, accumulate=False, params_via=None):
    """params_via(fn, *tensors): run the parameter pass (dgamma / dbeta from the row pass's partial sums -- a weight gradient) through
    the caller's weight-gradient lane (engine.WgradStream.run) instead of behind the row pass on the current stream; the partial sums
    then live in a buffer of their own (the shared scratch would be rewritten by the next call on this stream)."""
    _need_gpu(dy, x)
    D = x.shape[-1]
    M = x.numel() // D
    dx = torch.empty_like(x)
    need = L.lib().umr_layernorm_bwd_workspace(M, D)
    if params_via is not None:
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        L.check(L.lib().umr_layernorm_bwd_rows(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(ws), ws.numel(), M, D,
                                               _DT[x.dtype], _stream()), "umr_layernorm_bwd_rows")
        params_via(lambda: L.check(L.lib().umr_layernorm_bwd_params(_p(ws), ws.numel(), _p(dgamma), _p(dbeta), int(accumulate), M, D, _stream()),
                                   "umr_layernorm_bwd_params"), ws)
        return dx
    ws = _workspace(need, x.device)
    L.check(L.lib().umr_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dgamma), _p(dbeta),
                                      int(accumulate), _p(ws), ws.numel(), M, D, _DT[x.dtype], _stream()), "umr_layernorm_bwd")
    return dx


def attention_fwd(qkv, B, N, heads, need_lse=True):
    _need_gpu(qkv)
    D = heads * 64
    assert qkv.is_contiguous() and qkv.numel() == B * N * 3 * D
    out = torch.empty((B * N, D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B * heads * N, dtype=torch.float32, device=qkv.device) if need_lse else None
    L.check(L.lib().umr_attention_fwd(_p(qkv), _p(out), _p(lse), B, N, heads, 64, _DT[qkv.dtype], _stream()), "umr_attention_fwd")
    return out, lse


def attention_bwd(qkv, out, dout, lse, B, N, heads):
    _need_gpu(qkv, out, dout)
    assert dout.is_contiguous() and out.is_contiguous()
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty(L.lib().umr_attention_bwd_workspace(B, N, heads) // 4, dtype=torch.float32, device=qkv.device)
    L.check(L.lib().umr_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(dsum), _p(dqkv), B, N, heads, 64, _DT[qkv.dtype],
                                      _stream()), "umr_attention_bwd")
    return dqkv


def patchify(images, patch, dtype, ldk=None):
    _need_gpu(images)
    assert images.dtype == torch.float32 and images.is_contiguous()
    B, C, H, W = images.shape
    assert C == 3
    gh, gw = H // patch, W // patch
    k = 3 * patch * patch
    ldk = k if ldk is None else ldk
    out = torch.empty((B * gh * gw, ldk), dtype=dtype, device=images.device)
    L.check(L.lib().umr_patchify(_p(images), _p(out), B, H, W, patch, ldk, _DT[dtype], _stream()), "umr_patchify")
    return out


def _pixel_view(t, what):
    """(pointer tensor, B, H, W, C, pixel stride) of a [B, H, W, C] map whose pixels are rows of a (possibly wider) row-major buffer"""
    B, H, W, C = t.shape
    ld = t.stride(2)
    assert t.stride(3) == 1 and t.stride(1) == W * ld and (B == 1 or t.stride(0) == H * W * ld) and ld >= C, \
        f"{what}: [B,H,W,C] with contiguous channels and one pixel stride expected, got strides {t.stride()}"
    return B, H, W, C, ld


BILINEAR_RELU, BILINEAR_OUT_X3 = 1, 2


def bilinear_fwd(x, Ho, Wo, align_corners, relu=False, planes=False, out=None):
    """NHWC resize (blocks.py:155-172).  x / out may be column slices of wider row-major buffers.  relu: max(., 0) on the result;
    planes (x f32): the result as three bf16 planes per pixel [B, Ho, Wo, 3C] (the operand format of the plane GEMMs)."""
    _need_gpu(x)
    B, Hi, Wi, C, ldx = _pixel_view(x, "bilinear_fwd input")
    if out is None:
        out = torch.empty((B, Ho, Wo, 3 * C if planes else C), dtype=(torch.bfloat16 if planes else x.dtype), device=x.device)
    Bo, Ho_, Wo_, Co, ldy = _pixel_view(out, "bilinear_fwd output")
    assert (Bo, Ho_, Wo_) == (B, Ho, Wo) and Co == (3 * C if planes else C) and out.dtype == (torch.bfloat16 if planes else x.dtype)
    flags = (BILINEAR_RELU if relu else 0) | (BILINEAR_OUT_X3 if planes else 0)
    L.check(L.lib().umr_bilinear_fwd_ex(_p(x), ldx, _p(out), ldy, B, Hi, Wi, Ho, Wo, C, int(align_corners), flags, _DT[x.dtype], _stream()),
            "umr_bilinear_fwd_ex")
    return out


def bilinear_bwd(dy, Hi, Wi, align_corners, out=None):
    """exact adjoint of bilinear_fwd; dy / out may be column slices of wider row-major buffers"""
    _need_gpu(dy)
    B, Ho, Wo, C, lddy = _pixel_view(dy, "bilinear_bwd input")
    if out is None:
        out = torch.empty((B, Hi, Wi, C), dtype=dy.dtype, device=dy.device)
    Bo, Hi_, Wi_, Co, lddx = _pixel_view(out, "bilinear_bwd output")
    assert (Bo, Hi_, Wi_, Co) == (B, Hi, Wi, C) and out.dtype == dy.dtype
    L.check(L.lib().umr_bilinear_bwd_ex(_p(dy), lddy, _p(out), lddx, B, Hi, Wi, Ho, Wo, C, int(align_corners), _DT[dy.dtype], _stream()),
            "umr_bilinear_bwd_ex")
    return out


def pixel_shuffle(src, B, H, W, s, C, inverse=False):
    """forward: src [B*H*W, s*s*C] -> [B,H*s,W*s,C]; inverse: src [B,H*s,W*s,C] -> [B*H*W, s*s*C]"""
    _need_gpu(src)
    assert src.is_contiguous()
    if inverse:
        dst = torch.empty((B * H * W, s * s * C), dtype=src.dtype, device=src.device)
    else:
        dst = torch.empty((B, H * s, W * s, C), dtype=src.dtype, device=src.device)
    L.check(L.lib().umr_pixel_shuffle(_p(src), _p(dst), B, H, W, s, C, int(inverse), _DT[src.dtype], _stream()), "umr_pixel_shuffle")
    return dst


def zero_stuff2(dy, H, W):
    _need_gpu(dy)
    B, Ho, Wo, C = dy.shape
    out = torch.empty((B, H, W, C), dtype=dy.dtype, device=dy.device)
    L.check(L.lib().umr_zero_stuff2(_p(dy), _p(out), B, H, W, Ho, Wo, C, _DT[dy.dtype], _stream()), "umr_zero_stuff2")
    return out


# recorder for engine.PackCache: while a list is installed here, every permute4 / cast is appended to it as a replayable recipe
# (src tensor, dst tensor, dims, strides, offset) -- the per-step refresh of all packed weight copies replays them in ONE launch
_pack_recorder = None


_PERM_TILE_MAX = 4608


def _perm_tile(dims, strides):
    """Block shape for one permute of umr_permute4_batched: None = linear (innermost destination dimension contiguous in the
    source, or a tiny tensor), else (extents e[4], order ord[4]): a destination hyper-rectangle of <= 4608 elements whose
    innermost extent is >= 64 (stores in runs of >= 128 B) and which grows along the dimensions of smallest source stride first
    (gathers in long runs)."""
    total = dims[0] * dims[1] * dims[2] * dims[3]
    if abs(strides[3]) == 1 or dims[3] == 1 or total < 4096:
        return None
    if dims[0] == 1 and dims[1] == 1 and strides[2] == 1 and strides[3] > 0:
        return "t2d"                             # plain 2-D transpose: the kernel's 64 x 64 shift-indexed form
    order = sorted(range(4), key=lambda k: (abs(strides[k]) if dims[k] > 1 else 0, k))   # size-1 dimensions first (free)
    e = [1, 1, 1, 1]
    e[3] = min(dims[3], 64)
    rows_max = _PERM_TILE_MAX // (e[3] + 1)      # the kernel pads every row of e[3] floats by one (LDS bank spread)
    rows = 1
    for k in order:
        if k == 3:
            continue
        e[k] = max(1, min(dims[k], rows_max // rows))
        rows *= e[k]
    return e, order


def permute4_batched(recipes):
    """Replays recorded permutes (list of (src, dst, dims, strides, offset)) in one launch.  Returns a callable that launches
    it again on the current stream (the device-side table is built once)."""
    import numpy as np
    n = len(recipes)
    arr = (L.PermEntry * n)()
    blk = 0
    counts = []
    for i, rec in enumerate(recipes):
        src, dst, dims, strides, off = rec[:5]
        rowlen = rec[5] if len(rec) > 5 else 0     # > 0: dst holds bf16 PLANES of the f32 values, rows of `rowlen` values
        e = arr[i]
        e.src, e.dst = src.data_ptr(), dst.data_ptr()
        for k in range(4):
            e.d[k], e.sstride[k] = dims[k], strides[k]
        e.soff, e.dtype_in, e.blk_start = off, _DT[src.dtype], blk
        e.dtype_out, e.rowlen = (L.BF16X3, rowlen) if rowlen else (_DT[dst.dtype], 0)
        tile = _perm_tile(dims, strides)
        if tile is None:
            for k in range(4):
                e.e[k], e.ord[k] = 0, k
            blk += (dims[0] * dims[1] * dims[2] * dims[3] + 8191) // 8192
        elif tile == "t2d":
            for k in range(4):
                e.e[k], e.ord[k] = (-1 if k == 3 else 0), k
            blk += ((dims[2] + 63) // 64) * ((dims[3] + 63) // 64)
        else:
            ext, order = tile
            nb = 1
            for k in range(4):
                e.e[k], e.ord[k] = ext[k], order[k]
                nb *= (dims[k] + ext[k] - 1) // ext[k]
            blk += nb
        counts.append(blk - e.blk_start)
    host = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy())
    table = host.to(recipes[0][1].device)
    blk_entry = torch.from_numpy(np.repeat(np.arange(n, dtype=np.int32), counts)).to(recipes[0][1].device)
    keep = [r[0] for r in recipes] + [r[1] for r in recipes]   # the table holds raw pointers: keep the tensors alive with it

    def launch():
        L.check(L.lib().umr_permute4_batched(_p(table), n, blk, _p(blk_entry), _stream()), "umr_permute4_batched")
    launch.keep = keep
    return launch


def adam_pack(entries, hyper):
    """One optimizer launch for a stage of the flat parameter buffer that also writes the bf16 kernel-layout copies of its Linear
    weights (include/umr.h, umr_adam_pack_step).  entries: list of
        ("plain", p, g, m, v)                   1-D f32 views of equal length
        ("weight", p, g, m, v, dst_lin, dst_t)  p .. v: [N, K] f32 contiguous views; dst_lin [N, K] / dst_t [K, N] bf16 or None
    hyper: the device-side scalars of umr_adam_set_hyper.  Returns a callable that launches it on the current stream."""
    import numpy as np
    n = len(entries)
    arr = (L.AdamPackEntry * n)()
    blk, counts, keep = 0, [], [hyper]
    for i, ent in enumerate(entries):
        e = arr[i]
        p, g, m, v = ent[1:5]
        assert all(t.dtype == torch.float32 and t.is_contiguous() for t in (p, g, m, v)) and p.shape == g.shape == m.shape == v.shape
        e.p, e.g, e.m, e.v = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
        e.blk_start = blk
        if ent[0] == "plain":
            e.n, e.N, e.K, e.dst_lin, e.dst_t = p.numel(), 0, 0, None, None
            nb = (p.numel() + 4095) // 4096
        else:
            dl, dt_ = ent[5], ent[6]
            N, K = p.shape
            assert K % 4 == 0 and N % 8 == 0 and all(t.data_ptr() % 16 == 0 for t in (p, g, m, v))
            for d, shp in ((dl, (N, K)), (dt_, (K, N))):
                assert d is None or (d.dtype == torch.bfloat16 and d.is_contiguous() and tuple(d.shape) == shp and d.data_ptr() % 16 == 0)
            e.n, e.N, e.K = p.numel(), N, K
            e.dst_lin, e.dst_t = (dl.data_ptr() if dl is not None else None), (dt_.data_ptr() if dt_ is not None else None)
            nb = ((N + 63) // 64) * ((K + 63) // 64)
            keep += [t for t in (dl, dt_) if t is not None]
        keep += [p, g, m, v]
        blk += nb
        counts.append(nb)
    dev = entries[0][1].device
    table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
    blk_entry = torch.from_numpy(np.repeat(np.arange(n, dtype=np.int32), counts)).to(dev)

    def launch():
        L.check(L.lib().umr_adam_pack_step(_p(table), n, blk, _p(blk_entry), _p(hyper), _stream()), "umr_adam_pack_step")
    launch.keep = keep
    return launch


def permute4(src, dst, dst_dims, src_strides, src_offset=0, accumulate=False):
    """dst[i0,i1,i2,i3] = src.flat[src_offset + sum i_k*src_strides[k]] (cast to dst dtype)."""
    _need_gpu(src, dst)
    if _pack_recorder is not None and not accumulate:
        _pack_recorder.append((src, dst, tuple(dst_dims), tuple(src_strides), src_offset))
    dims = (ctypes.c_int32 * 4)(*dst_dims)
    strides = (ctypes.c_int64 * 4)(*src_strides)
    L.check(L.lib().umr_permute4(_p(src), _p(dst), ctypes.cast(dims, ctypes.c_void_p), ctypes.cast(strides, ctypes.c_void_p),
                                 src_offset, _DT[src.dtype], _DT[dst.dtype], int(accumulate), _stream()), "umr_permute4")
    return dst


def segsum(x, R, reps, rep_stride, seg_stride, C, out=None, out_f32=True, accumulate=False):
    _need_gpu(x)
    if out is None:
        out = torch.empty((R, C), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    L.check(L.lib().umr_segsum(_p(x), _p(out), R, reps, rep_stride, seg_stride, C, _DT[x.dtype], int(out.dtype == torch.float32),
                               int(accumulate), _stream()), "umr_segsum")
    return out


def fill_cls(tokens, cls, pos0, B, batch_stride, D):
    L.check(L.lib().umr_fill_cls(_p(tokens), _p(cls), _p(pos0), B, batch_stride, D, _DT[tokens.dtype], _stream()), "umr_fill_cls")


def cast(src, dtype, scale=1.0, out=None):
    _need_gpu(src)
    assert src.is_contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    if _pack_recorder is not None and scale == 1.0:
        _pack_recorder.append((src, out, (1, 1, 1, src.numel()), (0, 0, 0, 1), 0))
    L.check(L.lib().umr_cast(_p(src), _p(out), src.numel(), scale, _DT[src.dtype], _DT[out.dtype], _stream()), "umr_cast")
    return out


def scale_by_device_scalar(src, scalar):
    """src * scalar with the scalar read on the device (no host synchronisation)"""
    _need_gpu(src, scalar)
    assert src.dtype == torch.float32 and src.is_contiguous() and scalar.dtype == torch.float32 and scalar.numel() == 1
    out = torch.empty_like(src)
    L.check(L.lib().umr_scale_by_device_scalar(_p(src), _p(scalar), _p(out), src.numel(), _stream()), "umr_scale_by_device_scalar")
    return out


ACT_SINE = 4


def head_out_finish(partials, bias, B, H, W, act):
    """act(bias + sum over column tiles of the fused row-reduction partials) -> NCHW f32 [B, c, H, W]"""
    nparts, M, Cout = partials.shape
    assert M == B * H * W and partials.is_contiguous() and bias.dtype == torch.float32
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=partials.device)
    L.check(L.lib().umr_head_out_finish(_p(partials), nparts, _p(bias), _p(out), M, Cout, H * W, act, _stream()), "umr_head_out_finish")
    return out


def head_out_fwd(h, w, bias, B, H, W, act):
    """h [B*H*W, K] -> NCHW f32 [B, Cout, H, W]; w f32 [Cout, K]."""
    _need_gpu(h)
    M, K = h.shape
    Cout = w.shape[0]
    out = torch.empty((B, Cout, H, W), dtype=torch.float32, device=h.device)
    L.check(L.lib().umr_head_out_fwd(_p(h), _p(w), _p(bias), _p(out), M, K, Cout, H * W, act, _DT[h.dtype], _stream()), "umr_head_out_fwd")
    return out


def head_out_bwd(h, w, dout, yout, act, relu_mask, dw, db):
    _need_gpu(h)
    M, K = h.shape
    Cout = w.shape[0]
    HW = dout.shape[-1] * dout.shape[-2]
    dh = torch.empty_like(h)
    need = L.lib().umr_head_out_bwd_workspace(M, K)
    ws = _workspace(need, h.device)
    L.check(L.lib().umr_head_out_bwd(_p(h), _p(w), _p(dout), _p(yout), _p(dh), _p(dw), _p(db), _p(ws), ws.numel(), M, K, Cout, HW,
                                     act, int(relu_mask), _DT[h.dtype], _stream()), "umr_head_out_bwd")
    return dh


def objectness_loss(pc, ps, gc, gs, sal, center_l2=True, sdf_l2=False, use_grad=True, use_bce=True, need_grad=True, grad_scale=1.0):
    _need_gpu(pc, ps, gc, gs)
    B, _, H, W = pc.shape
    for t in (pc, ps, gc, gs):
        assert t.dtype == torch.float32 and t.is_contiguous()
    out5 = torch.empty(5, dtype=torch.float32, device=pc.device)
    dpc = torch.empty_like(pc) if need_grad else None
    dps = torch.empty_like(ps) if need_grad else None
    ws = _workspace(L.lib().umr_loss_workspace(), pc.device)
    L.check(L.lib().umr_objectness_loss(_p(pc), _p(ps), _p(gc), _p(gs), _p(sal), _p(dpc), _p(dps), _p(out5), _p(ws), B, H, W,
                                        int(center_l2), int(sdf_l2), int(use_grad), int(use_bce), grad_scale, _stream()),
            "umr_objectness_loss")
    return out5, dpc, dps


def adam_step(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _need_gpu(p, g, m, v)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous()
    L.check(L.lib().umr_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, grad_scale, _stream()),
            "umr_adam_step")


def adam_set_hyper(hyper, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """writes Adam's seven scalars for this step into the device buffer `hyper` (f32 [7]); see adam_step_hyper"""
    _need_gpu(hyper)
    assert hyper.dtype == torch.float32 and hyper.numel() >= 7 and hyper.is_contiguous()
    L.check(L.lib().umr_adam_set_hyper(_p(hyper), lr, beta1, beta2, eps, step, grad_scale, _stream()), "umr_adam_set_hyper")


def adam_step_hyper(p, g, m, v, hyper):
    """adam_step with its scalars read from device memory (a captured graph replays this launch with new values every step)"""
    _need_gpu(p, g, m, v, hyper)
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous()
    L.check(L.lib().umr_adam_step_hyper(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _stream()), "umr_adam_step_hyper")


# ---------------------------------------------------------------- kernel timing hook (bench.py roofline leg)
# HIP events recorded on the launch stream around selected umr_gemm_nt launches.
_timer = {"select": None, "events": [], "modes": []}


def set_kernel_timer(select):
    """select(desc: GemmDesc) -> bool chooses which gemm_nt launches to time; None disables."""
    _timer["select"] = select
    _timer["events"] = []
    _timer["modes"] = []


def kernel_timer_results_ms(with_modes=False):
    """durations of the timed launches; with_modes: (ms, f32 product mode at launch) pairs -- a certificate-driven sweep mixes
    three-term and six-term launches of the same kernel (reasoning.sweep_proposals)"""
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in _timer["events"]]
    return list(zip(ms, _timer["modes"])) if with_modes else ms


_raw_gemm_nt_call = None


def _wants_splitk_ws(d):
    return d.K >= 768 and ((d.M + 127) // 128) * ((d.N + 127) // 128) <= 170 and d.dtype in (L.BF16, L.F32)


def _gemm_nt_call(d):
    if d.dtype == L.BF16X3:
        ws = _x3_workspace(d, torch.device("cuda", _cur_dev()))
        if ws is not None:
            return L.lib().umr_gemm_nt_ws(ctypes.byref(d), _p(ws), ws.numel(), _stream())
        return L.lib().umr_gemm_nt(ctypes.byref(d), _stream())
    # few 128x128 tiles and a long K: hand the library its split-K scratch (include/umr.h: umr_gemm_nt_ws); the library decides
    if _wants_splitk_ws(d):
        ws = _splitk_workspace(torch.device("cuda", _cur_dev()))
        return L.lib().umr_gemm_nt_ws(ctypes.byref(d), _p(ws), ws.numel(), _stream())
    return L.lib().umr_gemm_nt(ctypes.byref(d), _stream())


def _timed_call(d):
    sel = _timer["select"]
    if sel is not None and sel(d):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        st = _gemm_nt_call(d)
        b.record()
        _timer["events"].append((a, b))
        _timer["modes"].append(get_f32_mode())
        return st
    return _gemm_nt_call(d)


# ---------------------------------------------------------------- collapsed linear head (opt-in)
def small_gemm(A, B, C, M, N, K, sa, sb, sc, accumulate=False):
    """C[i*sc[0] + j*sc[1]] (=|+=) sum_k A[i*sa[0] + k*sa[1]] * B[k*sb[0] + j*sb[1]]; f32 device tensors, element strides."""
    _need_gpu(A, B, C)
    assert A.dtype == B.dtype == C.dtype == torch.float32
    L.check(L.lib().umr_small_gemm_f32(_p(A), _p(B), _p(C), M, N, K, sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], int(accumulate), _stream()),
            "umr_small_gemm_f32")
    return C


def linear_head_fwd(x, kw, tapbias10, act):
    """x NHWC [B,H,W,256] -> [B,1,H,W] f32."""
    _need_gpu(x)
    B, H, W, C = x.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    L.check(L.lib().umr_linear_head_fwd(_p(x), _p(kw), _p(tapbias10), _p(out), B, H, W, C, act, _DT[x.dtype], _stream()), "umr_linear_head_fwd")
    return out


def linear_head_bwd_data(dout, yout, kw, dx, act, accumulate):
    B, H, W, C = dx.shape
    L.check(L.lib().umr_linear_head_bwd_data(_p(dout), _p(yout), _p(kw), _p(dx), B, H, W, C, act, int(accumulate), _DT[dx.dtype], _stream()),
            "umr_linear_head_bwd_data")
    return dx


def linear_head_bwd_weight(x, dout, yout, act):
    """-> f32 [9*C + 10] = [G (9,C) | n (9) | D (1)]"""
    B, H, W, C = x.shape
    out = torch.empty(9 * C + 10, dtype=torch.float32, device=x.device)
    ws = _workspace(L.lib().umr_linear_head_bwd_weight_workspace(B * H * W, C), x.device)
    L.check(L.lib().umr_linear_head_bwd_weight(_p(x), _p(dout), _p(yout), _p(out), _p(ws), ws.numel(), B, H, W, C, act, _DT[x.dtype],
                                               _stream()), "umr_linear_head_bwd_weight")
    return out


def linear_head_shift9(dout, yout, act, dtype):
    """-> (s9 [B, H, W, 16] of `dtype`: the nine shifted gradient maps g(q - off_t) of every pixel, zero padded to 16 channels;
    nd f32 [16] = n[0..8], D, zeros) -- csrc/linear_head.hip"""
    _need_gpu(dout)
    B, H, W = dout.shape[0], dout.shape[-2], dout.shape[-1]
    assert dout.dtype == torch.float32 and dout.is_contiguous() and dout.numel() == B * H * W
    s9 = torch.empty((B, H, W, 16), dtype=dtype, device=dout.device)
    nd = torch.empty(16, dtype=torch.float32, device=dout.device)
    ws = _workspace(L.lib().umr_linear_head_shift9_workspace(B * H * W), dout.device)
    L.check(L.lib().umr_linear_head_shift9(_p(dout), _p(yout), _p(s9), _p(nd), _p(ws), ws.numel(), B, H, W, act, _DT[dtype], _stream()),
            "umr_linear_head_shift9")
    return s9, nd


def linear_head_gather9(taps, tapbias10, act):
    """taps [B, H, W, >= 9] f32 (taps[q][t] = kw[t] . x(q); channels contiguous, one pixel stride) -> [B,1,H,W] f32:
    act(sum over the taps inside the image of (taps[q + off_t][t] + tapbias10[t]) + tapbias10[9]) -- csrc/linear_head.hip"""
    _need_gpu(taps, tapbias10)
    B, H, W, C, ldt = _pixel_view(taps, "linear_head_gather9 input")
    assert taps.dtype == torch.float32 and C >= 9 and tapbias10.dtype == torch.float32 and tapbias10.numel() >= 10
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=taps.device)
    L.check(L.lib().umr_linear_head_gather9(_p(taps), ldt, _p(tapbias10), _p(out), B, H, W, act, _stream()), "umr_linear_head_gather9")
    return out


# ---- existence classifier pieces (csrc/classifier.hip)
def im2col_nchw(images, KH, KW, stride, pad, ldk, dtype):
    """NCHW f32 [B,C,H,W] -> rows [B*Ho*Wo, ldk] (K order c,ky,kx; zero tail) for the 7x7 stem conv."""
    _need_gpu(images)
    assert images.dtype == torch.float32 and images.is_contiguous()
    B, C, H, W = images.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    out = torch.empty((B * Ho * Wo, ldk), dtype=dtype, device=images.device)
    L.check(L.lib().umr_im2col_nchw(_p(images), _p(out), B, C, H, W, KH, KW, stride, pad, ldk, _DT[dtype], _stream()), "umr_im2col_nchw")
    return out, Ho, Wo


def maxpool3x3s2(x):
    """nn.MaxPool2d(3, 2, 1) on NHWC [B,H,W,C]."""
    _need_gpu(x)
    assert x.is_contiguous()
    B, H, W, C = x.shape
    y = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
    L.check(L.lib().umr_maxpool3x3s2(_p(x), _p(y), B, H, W, C, _DT[x.dtype], _stream()), "umr_maxpool3x3s2")
    return y


def bn_fold(w2d, gamma, beta, mean, var, eps, ldk, dtype):
    """Fold eval-mode BatchNorm into packed conv weight rows w2d [Co,K] f32 -> ([Co,ldk] dtype, bias [Co] f32)."""
    _need_gpu(w2d)
    assert w2d.dtype == torch.float32 and w2d.is_contiguous() and all(t.dtype == torch.float32 for t in (gamma, beta, mean, var))
    Co, K = w2d.shape
    w_out = torch.empty((Co, ldk), dtype=dtype, device=w2d.device)
    b_out = torch.empty((Co,), dtype=torch.float32, device=w2d.device)
    L.check(L.lib().umr_bn_fold(_p(w2d), _p(gamma), _p(beta), _p(mean), _p(var), eps, _p(w_out), _p(b_out), Co, K, ldk, _DT[dtype], _stream()),
            "umr_bn_fold")
    return w_out, b_out
