"""The fp32-grade plane kernels beyond the head forward (round 4): every epilogue of the plane class of umr_gemm_nt, its K-split
work items + finish launch, the plane-operand weight-gradient kernel (plain, 3x3 conv on any map size, column sums), LayerNorm
with plane output, the row-gathering split, plane output of the batched weight refresh -- each against float64."""
import pytest
import torch
import torch.nn.functional as F

from test_gemm_gpu import _dev, _planes_to_f64, _rel_rms, _rnd, f32_mode_restored  # noqa: F401

pytestmark = pytest.mark.gpu


def _both(t, N, as_planes):
    """an [M, N] f32 epilogue operand in the requested format"""
    from unmore_amd import ops
    return ops.split3(t) if as_planes else t


@pytest.mark.parametrize("M,N,K", [(300, 264, 128), (1300, 1024, 1024), (5000, 512, 256)])
@pytest.mark.parametrize("planes_in", [False, True])
def test_gemm_nt_x3_epilogues(M, N, K, planes_in, monkeypatch):
    """(1300, 1024, 1024) is the attention projection of the reference recipe (dpt_large at 20 x 128^2: README.md:148-155) and
    runs as K-split work items + the finish launch; the other two run the epilogue inside the GEMM kernel."""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    A = _rnd((M, K), torch.float32, dev, 1)
    B = _rnd((N, K), torch.float32, dev, 2, K ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 3)
    aux = _rnd((M, N), torch.float32, dev, 4)
    aux2 = _rnd((M, N), torch.float32, dev, 5)
    Ap, Bp = ops.split3(A), ops.split3(B)
    lin = A.double() @ B.double().t()
    tol = dict(atol=2e-5, rtol=2e-5)
    ax, ax2 = _both(aux, N, planes_in), _both(aux2, N, planes_in)
    # residual add (+ second addend) + relu copy, both output formats
    for op in (False, True):
        out, r = ops.gemm_nt_x3(Ap, Bp, bias, aux=ax, aux2=ax2, c2_mode=1, out_planes=op, c2_planes=not op)
        o = _planes_to_f64(out, N) if op else out.double()
        rr = r.double() if op else _planes_to_f64(r, N)
        ref = lin + bias.double() + aux.double() + aux2.double()
        torch.testing.assert_close(o, ref, **tol)
        assert torch.equal(rr, F.relu(o))
    # GELU + saved pre-activation (timm Mlp.fc1 forward, the readout projection)
    out, pre = ops.gemm_nt_x3(Ap, Bp, bias, act=L.ACT_GELU, c2_mode=2, out_planes=True)
    torch.testing.assert_close(pre.double(), lin + bias.double(), **tol)
    torch.testing.assert_close(_planes_to_f64(out, N), F.gelu(lin + bias.double()), **tol)
    # ReLU-masked and GELU'-masked data gradients
    out = ops.gemm_nt_x3(Ap, Bp, None, mask=ax)
    torch.testing.assert_close(out.double(), lin * (aux.double() > 0), **tol)
    x = aux.double().requires_grad_(True)
    F.gelu(x).sum().backward()
    out = ops.gemm_nt_x3(Ap, Bp, None, dgelu=ax, out_planes=True)
    torch.testing.assert_close(_planes_to_f64(out, N), lin * x.grad, **tol)
    # row bias per image (the class-token half of the readout, models/dpt/vit.py:86-90)
    rpb = 100 if M % 100 == 0 else M
    rb = _rnd((M // rpb, N), torch.float32, dev, 6)
    out = ops.gemm_nt_x3(Ap, Bp, None, rowbias=rb, rows_per_batch=rpb)
    torch.testing.assert_close(out.double(), lin + rb.double().repeat_interleave(rpb, 0), **tol)
    # fp32 grade, not merely 2e-5: against the exact f32 MFMA path
    prev = ops.get_f32_mode()
    try:
        plain = ops.gemm_nt_x3(Ap, Bp, bias)
        ops.set_f32_mode("exact")
        exact = ops.gemm_nt(A, B, bias)
    finally:
        ops.set_f32_mode(prev)
    e_x3, e_ex = _rel_rms(plain, lin + bias.double()), _rel_rms(exact, lin + bias.double())
    assert e_x3 < 2.0 * e_ex + 1e-8 and e_x3 < 1e-6, (e_x3, e_ex)


def test_gemm_nt_x3_token_remap_and_broadcast_aux():
    """patch-embed form (models/dpt/vit.py:179-193): rows written past a class-token row per image, position embedding added per patch"""
    from unmore_amd import ops
    dev = _dev()
    Bn, g, D, K = 3, 64, 128, 768
    A = _rnd((Bn * g, K), torch.float32, dev, 11)
    W = _rnd((D, K), torch.float32, dev, 12, K ** -0.5)
    bias = _rnd((D,), torch.float32, dev, 13)
    pos = _rnd((g, D), torch.float32, dev, 14)
    tokens = torch.zeros((Bn * (g + 1), D), dtype=torch.float32, device=dev)
    ops.gemm_nt_x3(ops.split3(A), ops.split3(W), bias, out=tokens, aux=pos, aux_mod=g, c_remap=(g, g + 1, 1))
    ref = (A.double() @ W.double().t() + bias.double()).view(Bn, g, D) + pos.double()
    torch.testing.assert_close(tokens.view(Bn, g + 1, D)[:, 1:].double(), ref, atol=2e-5, rtol=2e-5)
    assert torch.equal(tokens.view(Bn, g + 1, D)[:, 0], torch.zeros((Bn, D), device=dev))


@pytest.mark.parametrize("nb,H,W,Cin,N", [(20, 8, 8, 256, 256), (20, 16, 16, 256, 256), (3, 32, 32, 128, 256), (20, 4, 4, 1024, 256)])
def test_conv3x3_x3_small_maps_split_k(nb, H, W, Cin, N):
    """the DPT fusion convs on the reference recipe's 4x4 ... 32x32 maps (models/dpt/blocks.py:80-115,290-313): a handful of
    tiles with 36-144 K-tiles, run as K-split work items; masked data-gradient epilogue with a plane mask, residual epilogue"""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    x = _rnd((nb, H, W, Cin), torch.float32, dev, 41)
    w = _rnd((N, 3, 3, Cin), torch.float32, dev, 42, (9 * Cin) ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 43)
    aux = _rnd((nb * H * W, N), torch.float32, dev, 44)
    conv = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, N)
    xp, wp = ops.split3(x), ops.split3(w.reshape(N, 9 * Cin))
    out, r = ops.gemm_nt_x3(xp, wp, bias, conv=1, aux=aux, c2_mode=1, c2_planes=True)
    torch.testing.assert_close(out.double(), conv + aux.double(), atol=2e-5, rtol=2e-5)
    assert torch.equal(_planes_to_f64(r, N), F.relu(out.double()))
    out = ops.gemm_nt_x3(xp, wp, bias, conv=1, act=L.ACT_RELU, out_planes=True)
    torch.testing.assert_close(_planes_to_f64(out, N), F.relu(conv), atol=2e-5, rtol=2e-5)
    out = ops.gemm_nt_x3(xp, wp, None, conv=1, mask=ops.split3(aux), out_planes=True)
    torch.testing.assert_close(_planes_to_f64(out, N), (conv - bias.double()) * (aux.double() > 0), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("M,N,K", [(1300, 384, 256), (70000, 512, 256), (333, 1024, 3072), (64 * 20, 96, 768)])
def test_gemm_tn_x3_plain(M, N, K, f32_mode_restored):
    from unmore_amd import ops
    dev = _dev()
    dY = _rnd((M, N), torch.float32, dev, 51)
    X = _rnd((M, K), torch.float32, dev, 52)
    ref = dY.double().t() @ X.double()
    db = torch.empty(N, dtype=torch.float32, device=dev)
    dW = ops.gemm_tn(ops.split3(dY), ops.split3(X), dbias=db, x3=True)
    assert dW.shape == (N, K)
    s = M ** 0.5
    torch.testing.assert_close(dW.double(), ref, atol=3e-6 * s, rtol=2e-5)
    torch.testing.assert_close(db.double(), dY.double().sum(0), atol=3e-6 * s, rtol=2e-5)
    ops.set_f32_mode("exact")
    exact = ops.gemm_tn(dY, X)
    e_x3, e_ex = _rel_rms(dW, ref), _rel_rms(exact, ref)
    assert e_x3 < 2.0 * e_ex + 1e-8 and e_x3 < 1e-6, (e_x3, e_ex)
    # accumulate into an existing gradient, strided destination (the two halves of the readout weight, vit.py:84)
    big = torch.ones((N, 2 * K), dtype=torch.float32, device=dev)
    ops.gemm_tn(ops.split3(dY), ops.split3(X), dW=big[:, K:], accumulate=True, x3=True)
    torch.testing.assert_close(big[:, K:].double(), ref + 1.0, atol=3e-6 * s, rtol=2e-5)
    assert torch.equal(big[:, :K], torch.ones((N, K), device=dev))


@pytest.mark.parametrize("nb,H,W,Cin,N", [(20, 8, 8, 256, 256), (20, 4, 4, 64, 72), (2, 16, 16, 128, 256), (3, 64, 64, 64, 128),
                                          (2, 37, 96, 64, 64), (1, 128, 128, 128, 256), (2, 19, 23, 64, 64)])
def test_gemm_tn_x3_conv_any_map_size(nb, H, W, Cin, N):
    """weight gradient of the 3x3 convs from plane operands: maps narrower than a 64-row stage (general halo arithmetic), rows of
    exactly 64 / 128 pixels (scalar halo test), and rows a stage straddles once"""
    from unmore_amd import ops
    dev = _dev()
    x = _rnd((nb, H, W, Cin), torch.float32, dev, 61)
    dy = _rnd((nb, H, W, N), torch.float32, dev, 62)
    xt = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros((N, Cin, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
    F.conv2d(xt, w, padding=1).backward(dy.double().permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(N, 9 * Cin)          # [co][ky][kx][ci]
    db = torch.empty(N, dtype=torch.float32, device=dev)
    dW = ops.gemm_tn(ops.split3(dy.reshape(-1, N)), ops.split3(x), dbias=db, conv=1, x3=True)
    s = (nb * H * W) ** 0.5
    torch.testing.assert_close(dW.double(), ref, atol=3e-6 * s, rtol=2e-5)
    torch.testing.assert_close(db.double(), dy.double().sum((0, 1, 2)), atol=3e-6 * s, rtol=2e-5)


def test_layernorm_planes_and_split3_row_gather():
    from unmore_amd import ops
    dev = _dev()
    M, D = 1300, 1024
    x = _rnd((M, D), torch.float32, dev, 71) * 3 + 0.5
    g, b = _rnd((D,), torch.float32, dev, 72), _rnd((D,), torch.float32, dev, 73)
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    yp, mean2, rstd2 = ops.layernorm_fwd(x, g, b, planes=True)
    assert yp.shape == (M, 3 * D) and torch.equal(_planes_to_f64(yp, D), y.double())
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    # tokens without the class-token rows (vit.py:87-88)
    Bn, Nt = 20, 65
    tok = _rnd((Bn * Nt, D), torch.float32, dev, 74)
    pl = ops.split3(tok, remap=(Nt - 1, Nt, 1))
    assert pl.shape == (Bn * (Nt - 1), 3 * D)
    assert torch.equal(_planes_to_f64(pl, D), tok.view(Bn, Nt, D)[:, 1:].reshape(-1, D).double())


def test_batched_refresh_writes_planes():
    """the per-step weight refresh (umr_permute4_batched) with plane destinations: Linear weights (a cast), their transposes,
    conv weights in the [co][ky][kx][ci] and flipped data-gradient layouts -- each equals split3 of the f32 pack"""
    from unmore_amd import engine, ops
    dev = _dev()
    wl = _rnd((96, 200), torch.float32, dev, 81)
    wc = _rnd((72, 40, 3, 3), torch.float32, dev, 82)
    recipes, want = [], []
    for build, src, K in ((lambda: engine._pack_linear_t(wl, torch.float32), wl, 96), (lambda: engine._pack_conv3(wc, torch.float32), wc, 360),
                          (lambda: engine._pack_conv3_dgrad(wc, torch.float32), wc, 648)):
        rec = []
        ops._pack_recorder = rec
        try:
            f32 = build()
        finally:
            ops._pack_recorder = None
        assert len(rec) == 1
        dst = torch.zeros((f32.shape[0], 3 * K), dtype=torch.bfloat16, device=dev)
        recipes.append((rec[0][0], dst, rec[0][2], rec[0][3], rec[0][4], K))
        want.append(ops.split3(f32))
    dst = torch.zeros((96, 600), dtype=torch.bfloat16, device=dev)
    recipes.append((wl, dst, (1, 1, 1, 96 * 200), (0, 0, 0, 1), 0, 200))
    want.append(ops.split3(wl))
    ops.permute4_batched(recipes)()
    for r, w in zip(recipes, want):
        assert torch.equal(r[1], w)
