"""GPU parity of the MFMA GEMM / implicit-conv kernels against PyTorch fp64
references of the same op (the oracle's building blocks).  Runs on the MI355X."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch.device("cuda:0")


def _tol(dtype):
    return dict(atol=2e-5, rtol=2e-5) if dtype == torch.float32 else dict(atol=3e-2, rtol=3e-2)


def _rnd(shape, dtype, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 136), (1, 8, 8), (577 * 2, 2304, 768), (70, 96, 1536)])
def test_gemm_nt_plain(dtype, M, N, K):
    from unmore_amd import ops, _lib as L
    dev = _dev()
    A = _rnd((M, K), dtype, dev, 1)
    B = _rnd((N, K), dtype, dev, 2, K ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 3)
    aux = _rnd((M, N), dtype, dev, 4)
    ref = A.double() @ B.double().t() + bias.double()
    out = ops.gemm_nt(A, B, bias)
    torch.testing.assert_close(out.double(), ref, **_tol(dtype))
    # gelu + second output (pre-activation)
    out, pre = ops.gemm_nt(A, B, bias, act=L.ACT_GELU, c2_mode=2)
    torch.testing.assert_close(pre.double(), ref, **_tol(dtype))
    torch.testing.assert_close(out.double(), F.gelu(ref), **_tol(dtype))
    # residual add + relu copy
    out, r = ops.gemm_nt(A, B, bias, aux=aux, c2_mode=1)
    torch.testing.assert_close(out.double(), ref + aux.double(), **_tol(dtype))
    torch.testing.assert_close(r.double(), F.relu(out.double()), **_tol(dtype))
    # relu-mask and dgelu-mask epilogues (backward forms), f32 output
    out = ops.gemm_nt(A, B, None, aux=aux, mask_relu=True, out_f32=True)
    torch.testing.assert_close(out.double(), (A.double() @ B.double().t()) * (aux.double() > 0), **_tol(dtype))
    x = aux.double().requires_grad_(True)
    F.gelu(x).sum().backward()
    out = ops.gemm_nt(A, B, None, aux=aux, mask_dgelu=True)
    torch.testing.assert_close(out.double(), (A.double() @ B.double().t()) * x.grad, **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nt_rowbias_and_strided(dtype):
    from unmore_amd import ops
    dev = _dev()
    Bn, Ntok, D = 3, 17, 128
    tok = _rnd((Bn, Ntok, D), dtype, dev, 5)
    Wt = _rnd((D, D), dtype, dev, 6, D ** -0.5)
    rb = _rnd((Bn, D), torch.float32, dev, 7)
    A = tok.reshape(-1, D)
    out = ops.gemm_nt(A, Wt, None, rowbias=rb, rows_per_batch=Ntok)
    ref = A.double() @ Wt.double().t() + rb.double().repeat_interleave(Ntok, 0)
    torch.testing.assert_close(out.double(), ref, **_tol(dtype))
    # strided A (lda > K) and strided output
    big = _rnd((40, 3 * D), dtype, dev, 8)
    outbig = torch.zeros((40, 2 * D), dtype=dtype, device=dev)
    ops.gemm_nt(big[:, D:2 * D], Wt, None, out=outbig[:, D:])
    torch.testing.assert_close(outbig[:, D:].double(), big[:, D:2 * D].double() @ Wt.double().t(), **_tol(dtype))
    assert outbig[:, :D].abs().sum() == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nb,H,W,Cin,Cout,stride", [(2, 8, 8, 64, 128, 1), (1, 13, 9, 96, 40, 1), (2, 6, 4, 32, 256, 1),
                                                     (2, 12, 12, 128, 128, 2), (1, 7, 5, 64, 64, 2), (1, 24, 24, 256, 256, 1)])
def test_conv3x3_fwd(dtype, nb, H, W, Cin, Cout, stride):
    from unmore_amd import ops
    dev = _dev()
    x = _rnd((nb, H, W, Cin), dtype, dev, 11)
    w = _rnd((Cout, Cin, 3, 3), dtype, dev, 12, (9 * Cin) ** -0.5)
    bias = _rnd((Cout,), torch.float32, dev, 13)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous()
    out = ops.gemm_nt(x, wp, bias, conv=2 if stride == 2 else 1)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), stride=stride, padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    torch.testing.assert_close(out.double(), ref, **_tol(dtype))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (1000, 200, 136), (5000, 768, 768), (33, 8, 16)])
def test_gemm_tn_plain(dtype, M, N, K):
    from unmore_amd import ops
    dev = _dev()
    dY = _rnd((M, N), dtype, dev, 21)
    X = _rnd((M, K), dtype, dev, 22)
    db = torch.empty(N, dtype=torch.float32, device=dev)
    dW = ops.gemm_tn(dY, X, dbias=db)
    ref = dY.double().t() @ X.double()
    tol = dict(atol=2e-4 * M ** 0.5, rtol=1e-4) if dtype == torch.float32 else dict(atol=2e-2 * M ** 0.5, rtol=2e-2)
    torch.testing.assert_close(dW.double(), ref, **tol)
    torch.testing.assert_close(db.double(), dY.double().sum(0), **tol)
    dW2 = ops.gemm_tn(dY, X, dW=dW.clone(), accumulate=True)
    torch.testing.assert_close(dW2.double(), 2 * ref, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nb,H,W,Cin,Cout,stride", [(2, 8, 8, 64, 128, 1), (1, 13, 9, 96, 40, 1), (2, 12, 12, 128, 128, 2),
                                                     (3, 24, 24, 256, 256, 1)])
def test_conv3x3_wgrad(dtype, nb, H, W, Cin, Cout, stride):
    from unmore_amd import ops
    dev = _dev()
    x = _rnd((nb, H, W, Cin), dtype, dev, 31)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = _rnd((nb, Ho, Wo, Cout), dtype, dev, 32)
    db = torch.empty(Cout, dtype=torch.float32, device=dev)
    dW = ops.gemm_tn(dy, x, dbias=db, conv=2 if stride == 2 else 1)
    xr = x.double().permute(0, 3, 1, 2)
    wr = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
    out = F.conv2d(xr, wr, None, stride=stride, padding=1)
    out.backward(dy.double().permute(0, 3, 1, 2))
    ref = wr.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)
    M = nb * Ho * Wo
    tol = dict(atol=2e-4 * M ** 0.5, rtol=1e-4) if dtype == torch.float32 else dict(atol=2e-2 * M ** 0.5, rtol=2e-2)
    torch.testing.assert_close(dW.double(), ref, **tol)
    torch.testing.assert_close(db.double(), dy.double().sum((0, 1, 2)), **tol)


def test_invalid_arguments_raise():
    from unmore_amd import ops
    dev = _dev()
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.zeros((4, 10), device=dev), torch.zeros((8, 10), device=dev))
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.zeros((4, 8)), torch.zeros((8, 8)))  # CPU tensors: no fallback


@pytest.mark.parametrize("M,C,relu", [(2 * 256 * 256, 2, True), (2 * 256 * 256 + 100, 1, False)])
def test_gemm_nt_fused_row_reduction(M, C, relu):
    """umr_gemm_desc.red_*: the head's 1024 -> {1,2} output layer folded into the producing GEMM's epilogue
    (objectness_net.py:116,133).  Partials must reproduce (stored C) @ red_w^T; C itself is unchanged by the fusion."""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    N, K = 1024, 512
    A = _rnd((M // 64 + 1, K), torch.bfloat16, dev, 21).repeat(64, 1)[:M].contiguous()
    A[::7] *= 0.5
    B = _rnd((N, K), torch.bfloat16, dev, 22, K ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 23)
    rw = _rnd((C, N), torch.float32, dev, 24, N ** -0.5)
    act = L.ACT_RELU if relu else L.ACT_NONE
    assert ops.gemm_nt(A, B, bias, act=act, query_rowreduce=True), "shape chosen to run on the persistent 256x256 path"
    ref_c = ops.gemm_nt(A, B, bias, act=act)
    out, parts = ops.gemm_nt(A, B, bias, act=act, red_w=rw)
    assert torch.equal(out, ref_c)
    assert parts.shape == (N // 64, M, C)
    ref = ref_c.double() @ rw.double().t()
    torch.testing.assert_close(parts.sum(0).double(), ref, atol=1e-4, rtol=1e-4)
    # every 64-column slice separately
    for t in range(N // 64):
        sl = slice(t * 64, (t + 1) * 64)
        torch.testing.assert_close(parts[t].double(), ref_c[:, sl].double() @ rw[:, sl].double().t(), atol=1e-4, rtol=1e-4)
    # inference form: C is not written at all
    none, parts2 = ops.gemm_nt(A, B, bias, act=act, red_w=rw, no_store=True)
    assert none is None and torch.equal(parts2, parts)
    # finish kernel: fixed-order sum + bias + activation, NCHW
    b4 = _rnd((C,), torch.float32, dev, 25)
    if M % (256 * 256) == 0:
        Bn = M // (256 * 256)
        z = ops.head_out_finish(parts, b4, Bn, 256, 256, L.ACT_TANH)
        zr = torch.tanh(ref + b4.double()).view(Bn, 256, 256, C).permute(0, 3, 1, 2)
        torch.testing.assert_close(z.double(), zr, atol=1e-4, rtol=1e-4)
    # a shape that runs on another path must refuse the request instead of ignoring it
    small = _rnd((300, K), torch.bfloat16, dev, 26)
    assert not ops.gemm_nt(small, B, bias, query_rowreduce=True)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(small, B, bias, red_w=rw)


# ---- the large-tile (256x256) kernels only run on problems with >= 2048 tiles / many rows: parity at such sizes against
# ---- torch fp32 ops of the same bf16 operands, computed on the GPU (the shapes are too large for a CPU fp64 reference)
@pytest.mark.parametrize("M,N,K", [(270000, 512, 512), (2 * 256 * 256, 1024, 256), (530001, 200, 64)])
def test_large_tile_nt_gemm(M, N, K):
    from unmore_amd import ops, _lib as L
    dev = _dev()
    A = _rnd((M // 16 + 1, K), torch.bfloat16, dev, 41).repeat(16, 1)[:M].contiguous()
    A[::5] *= -0.5
    B = _rnd((N, K), torch.bfloat16, dev, 42, K ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 43)
    aux = _rnd((M, N), torch.bfloat16, dev, 44)
    ref = A.float() @ B.float().t() + bias
    assert ops.gemm_nt(A, B, bias, query_rowreduce=True), "shape chosen to run on the persistent 256x256 kernel"
    torch.testing.assert_close(ops.gemm_nt(A, B, bias, act=L.ACT_RELU).float(), torch.relu(ref), atol=3e-2, rtol=3e-2)
    torch.testing.assert_close(ops.gemm_nt(A, B, None, aux=aux, mask_relu=True).float(), (ref - bias) * (aux.float() > 0), atol=3e-2, rtol=3e-2)
    out, pre = ops.gemm_nt(A, B, bias, act=L.ACT_GELU, c2_mode=2)   # GELU epilogue class
    torch.testing.assert_close(pre.float(), ref, atol=3e-2, rtol=3e-2)
    torch.testing.assert_close(out.float(), F.gelu(ref), atol=3e-2, rtol=3e-2)
    out = ops.gemm_nt(A, B, bias, act=L.ACT_TANH)                   # generic epilogue class
    torch.testing.assert_close(out.float(), torch.tanh(ref), atol=3e-2, rtol=3e-2)


def test_transformer_shape_gemms_on_large_tiles():
    """The ViT-B GEMMs at the cfg2 token count (64 x 577 rows: a ragged last row tile) run on the 256x256 kernels: the GELU
    epilogue class (GELU + saved pre-activation, GELU'-masked gradient), residual add, and the one-round split weight gradient."""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    M, D = 64 * 577, 768
    x = _rnd((M, D), torch.bfloat16, dev, 61)
    x4 = _rnd((M, 4 * D), torch.bfloat16, dev, 62)
    w1 = _rnd((4 * D, D), torch.bfloat16, dev, 63, D ** -0.5)
    w2 = _rnd((D, 4 * D), torch.bfloat16, dev, 64, (4 * D) ** -0.5)
    b4 = _rnd((4 * D,), torch.float32, dev, 65)
    b1 = _rnd((D,), torch.float32, dev, 66)
    assert ops.gemm_nt(x, w1, b4, query_rowreduce=True) and ops.gemm_nt(x4, w2, b1, query_rowreduce=True), "expected the 256x256 path"
    ref = x.float() @ w1.float().t() + b4
    h, pre = ops.gemm_nt(x, w1, b4, act=L.ACT_GELU, c2_mode=2)
    torch.testing.assert_close(pre.float(), ref, atol=3e-2, rtol=3e-2)
    torch.testing.assert_close(h.float(), F.gelu(ref), atol=3e-2, rtol=3e-2)
    h1 = ops.gemm_nt(x, w1, b4, act=L.ACT_GELU)
    assert torch.equal(h1, h)
    # GELU'-masked gradient: dpre = (dy . W2) * gelu'(pre)
    g = ops.gemm_nt(x, w2.t().contiguous(), None, aux=pre, mask_dgelu=True)
    p32 = pre.float()
    dg = 0.5 * (1 + torch.erf(p32 / 2 ** 0.5)) + p32 * torch.exp(-0.5 * p32 * p32) / (2 * torch.pi) ** 0.5
    torch.testing.assert_close(g.float(), (x.float() @ w2.float()) * dg, atol=3e-2, rtol=3e-2)
    # residual add
    y = ops.gemm_nt(x4, w2, b1, aux=x)
    torch.testing.assert_close(y.float(), x4.float() @ w2.float().t() + b1 + x.float(), atol=5e-2, rtol=3e-2)
    # weight gradient + bias gradient (split over rows, one round)
    dW = torch.empty((4 * D, D), dtype=torch.float32, device=dev)
    db = torch.empty((4 * D,), dtype=torch.float32, device=dev)
    ops.gemm_tn(x4, x, dW=dW, dbias=db)
    rw = x4.float().t() @ x.float()
    torch.testing.assert_close(dW, rw, atol=2e-3 * float(rw.abs().max()), rtol=0)
    rb = x4.float().sum(0)
    torch.testing.assert_close(db, rb, atol=2e-3 * float(rb.abs().max()), rtol=0)


@pytest.mark.parametrize("nb,H,W,Cin,N", [(4, 256, 256, 64, 512), (9, 100, 300, 128, 320), (40, 128, 128, 64, 256)])
def test_large_tile_conv3x3_fwd(nb, H, W, Cin, N):
    from unmore_amd import ops, _lib as L
    dev = _dev()
    x = _rnd((nb, H, W, Cin), torch.bfloat16, dev, 31)
    w = _rnd((N, Cin, 3, 3), torch.bfloat16, dev, 32, (9 * Cin) ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 33)
    wp = w.permute(0, 2, 3, 1).reshape(N, 9 * Cin).contiguous()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, N)
    out = ops.gemm_nt(x, wp, bias, conv=1)
    torch.testing.assert_close(out.float(), ref, atol=3e-2, rtol=3e-2)
    o4, r4 = out.float().view(nb, H, W, N), ref.view(nb, H, W, N)   # image borders: where the halo masks act
    for a, b in ((o4[:, 0], r4[:, 0]), (o4[:, -1], r4[:, -1]), (o4[:, :, 0], r4[:, :, 0]), (o4[:, :, -1], r4[:, :, -1])):
        torch.testing.assert_close(a, b, atol=3e-2, rtol=3e-2)
    aux = _rnd((nb * H * W, N), torch.bfloat16, dev, 34)
    out = ops.gemm_nt(x, wp, None, conv=1, aux=aux, mask_relu=True)
    torch.testing.assert_close(out.float(), (ref - bias) * (aux.float() > 0), atol=3e-2, rtol=3e-2)
    # the mask is applied to the stored bf16 values: bit-identical to masking the unmasked output
    plain = ops.gemm_nt(x, wp, None, conv=1)
    assert torch.equal(out, torch.where(aux > 0, plain, torch.zeros_like(plain)))
    # residual add (fusion blocks: conv2 + skip): bf16(bf16(conv + bias) + aux)
    out = ops.gemm_nt(x, wp, bias, conv=1, aux=aux)
    with_bias = ops.gemm_nt(x, wp, bias, conv=1)
    assert torch.equal(out, (with_bias.float() + aux.float()).to(torch.bfloat16))


def test_large_tile_tn_plain_and_conv():
    from unmore_amd import ops
    dev = _dev()
    # plain: dW = dY^T X, dbias = column sums
    M, N, K = 300000, 512, 256
    dY = _rnd((M // 8, N), torch.bfloat16, dev, 51).repeat(8, 1)
    X = _rnd((M // 8, K), torch.bfloat16, dev, 52).repeat(8, 1)
    X[::3] *= -1.0
    dW = torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty((N,), dtype=torch.float32, device=dev)
    ops.gemm_tn(dY, X, dW=dW, dbias=db)
    ref = dY.float().t() @ X.float()
    scale = float(ref.abs().max())
    torch.testing.assert_close(dW, ref, atol=2e-3 * scale, rtol=0)
    refb = dY.float().sum(0)
    torch.testing.assert_close(db, refb, atol=2e-3 * float(refb.abs().max()), rtol=0)
    # accumulate form
    ops.gemm_tn(dY, X, dW=dW, dbias=db, accumulate=True)
    torch.testing.assert_close(dW, 2 * ref, atol=4e-3 * scale, rtol=0)
    # conv weight gradient (stride 1, Wo % 64 == 0: the conv fast path of the 256 kernel)
    nb, H, W, Cin, Co = 6, 256, 256, 64, 256
    x = _rnd((nb, H, W, Cin), torch.bfloat16, dev, 53)
    dy = _rnd((nb * H * W, Co), torch.bfloat16, dev, 54)
    dwp = ops.gemm_tn(dy, x, conv=1)                                    # [Co][ky][kx][ci]
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (Co, Cin, 3, 3),
                                      dy.float().view(nb, H, W, Co).permute(0, 3, 1, 2), padding=1)   # [Co,Cin,3,3]
    got = dwp.view(Co, 3, 3, Cin).permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref, atol=2e-3 * float(ref.abs().max()), rtol=0)
    # Wo % 64 != 0: 64-row stages straddle image-row ends; M % 64 != 0: ragged last stage
    for nb, H, W in ((50, 96, 96), (37, 70, 110)):
        x = _rnd((nb, H, W, Cin), torch.bfloat16, dev, 55)
        dy = _rnd((nb * H * W, Co), torch.bfloat16, dev, 56)
        db = torch.empty((Co,), dtype=torch.float32, device=dev)
        dwp = ops.gemm_tn(dy, x, conv=1, dbias=db)
        ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (Co, Cin, 3, 3),
                                          dy.float().view(nb, H, W, Co).permute(0, 3, 1, 2), padding=1)
        torch.testing.assert_close(dwp.view(Co, 3, 3, Cin).permute(0, 3, 1, 2), ref, atol=2e-3 * float(ref.abs().max()), rtol=0)
        refb = dy.float().sum(0)
        torch.testing.assert_close(db, refb, atol=2e-3 * float(refb.abs().max()), rtol=0)


# ------------------------------------------------------------------------------------------------ fp32 product modes
@pytest.fixture
def f32_mode_restored():
    from unmore_amd import ops
    prev = ops.get_f32_mode()
    yield
    ops.set_f32_mode(prev)


def _rel_rms(x, ref):
    return ((x.double() - ref).norm() / ref.norm()).item()


@pytest.mark.parametrize("kind", ["nt", "tn", "conv"])
def test_f32_x3_is_fp32_grade_on_finite_operands(kind, f32_mode_restored):
    """UMR_F32_X3 (three-way bf16 splits on the bf16 matrix cores, the default of the fp32 parity mode) against the exact f32
    MFMA path and float64, switched inside ONE process (umr_set_f32_mode).  Operands span 60 binary orders of magnitude per
    row / column (scaled so that every dot product mixes them): the relative rms error of X3 must be within 1.5x of the exact
    path's, and both within 2e-6."""
    from unmore_amd import ops
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(11)
    if kind == "conv":
        x = torch.randn((2, 12, 10, 64), generator=g)
        w = torch.randn((96, 9 * 64), generator=g) * 0.05
        sx = torch.exp2(torch.randint(-30, 30, (1, 1, 1, 64), generator=g).float())
        A, B = (x * sx).to(dev), (w / sx.reshape(1, 1, 64).expand(1, 9, 64).reshape(1, -1)).to(dev)
        ref = F.conv2d(A.double().permute(0, 3, 1, 2), B.double().reshape(96, 3, 3, 64).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(-1, 96)
        run = lambda: ops.gemm_nt(A, B, None, conv=1)
    else:
        M, N, K = 700, 200, 512
        a = torch.randn((M, K), generator=g)
        b = torch.randn((N, K), generator=g)
        sk = torch.exp2(torch.randint(-30, 30, (1, K), generator=g).float())
        if kind == "nt":
            A, B = (a * sk).to(dev), (b / sk).to(dev)
            ref = A.double() @ B.double().t()
            run = lambda: ops.gemm_nt(A, B, None)
        else:   # dW[N,K'] = dY[M,N]^T X[M,K']: the reduction runs over rows, so the scales sit on rows
            sm = torch.exp2(torch.randint(-30, 30, (M, 1), generator=g).float())
            dY, X = (a[:, :N] * sm).to(dev), (torch.randn((M, 160), generator=g) / sm).to(dev)
            ref = dY.double().t() @ X.double()
            run = lambda: ops.gemm_tn(dY, X)
    ops.set_f32_mode("exact")
    e_exact = _rel_rms(run(), ref)
    ops.set_f32_mode("x3")
    e_x3 = _rel_rms(run(), ref)
    print(f"{kind}: relative rms error vs float64 -- exact f32 MFMA {e_exact:.2e}, X3 {e_x3:.2e}")
    assert e_exact < 2e-6 and e_x3 < 2e-6 and e_x3 < 1.5 * e_exact + 1e-8


def test_f32_x3_documented_behaviour_outside_its_range(f32_mode_restored):
    """What include/umr.h promises for non-finite / extreme operands, pinned: the EXACT mode behaves like an f32 FMA chain
    (inf stays inf, FLT_MAX-sized and denormal operands are handled); X3 turns an inf or near-FLT_MAX operand into NaN and
    loses the low-order terms of operands below ~2^-110 -- so a caller with such data selects the exact mode."""
    from unmore_amd import ops
    dev = _dev()
    K = 64
    A = torch.zeros((4, K), device=dev)
    B = torch.zeros((8, K), device=dev)
    B[:, 0] = 1.0
    A[0, 0] = float("inf")
    A[1, 0] = 3.4e38           # rounds to inf as a bf16
    A[2, 0] = 2.0e-38          # normal in f32, but its m / l split terms are bf16 denormals
    A[3, 0] = 1.2345678
    ops.set_f32_mode("exact")
    ex = ops.gemm_nt(A, B, None).cpu()
    ops.set_f32_mode("x3")
    x3 = ops.gemm_nt(A, B, None).cpu()
    assert torch.isinf(ex[0]).all() and (ex[0] > 0).all()
    assert torch.allclose(ex[1], torch.full((8,), 3.4e38)) and torch.allclose(ex[3], torch.full((8,), 1.2345678))
    assert (ex[2] == A[2, 0].item()).all()
    assert torch.isnan(x3[0]).all(), "X3: inf operand -> NaN (inf - inf in the split), as documented"
    assert not torch.isfinite(x3[1]).any(), "X3: |x| within 2^-8 of FLT_MAX is out of range, as documented"
    assert torch.allclose(x3[3], torch.full((8,), 1.2345678), rtol=3e-7, atol=0)
    assert ((x3[2] - ex[2]).abs() <= 2.0 ** -8 * ex[2].abs()).all()   # at least the leading bf16 term survives
    # NaN propagates in both modes
    A[3, 1] = float("nan")
    B[:, 1] = 1.0
    for mode in ("exact", "x3"):
        ops.set_f32_mode(mode)
        assert torch.isnan(ops.gemm_nt(A, B, None)[3]).all()


# ------------------------------------------------------------------------------------------------ f32 values as bf16 planes
def _planes_to_f64(pl, K):
    return pl[..., :K].double() + pl[..., K:2 * K].double() + pl[..., 2 * K:].double()


def test_split3_is_lossless_for_f32():
    from unmore_amd import ops
    dev = _dev()
    x = _rnd((777, 200), torch.float32, dev, 21) * torch.exp2(torch.randint(-40, 40, (777, 1), generator=torch.Generator().manual_seed(3)).float()).to(dev)
    pl = ops.split3(x)
    assert pl.shape == (777, 600) and pl.dtype == torch.bfloat16
    assert torch.equal(_planes_to_f64(pl, 200), x.double())          # h + m + l == x exactly
    assert torch.equal(pl[:, :200], x.to(torch.bfloat16))             # h = round-to-nearest bf16 of x


@pytest.mark.parametrize("M,N,K,relu", [(1000, 264, 192, True), (257, 8, 64, False), (70001, 512, 256, True)])
def test_gemm_nt_x3_planes(M, N, K, relu, f32_mode_restored):
    """umr_gemm_nt with dtype UMR_BF16X3 (operands = f32 values as three bf16 planes, six plane pairs per K-tile on the persistent
    256x256 kernel): fp32-grade vs float64 -- error within 2x of the exact f32 MFMA path's --, f32 and plane outputs agree
    exactly, tails in M and N."""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    A = _rnd((M, K), torch.float32, dev, 31)
    B = _rnd((N, K), torch.float32, dev, 32, K ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 33)
    ref = A.double() @ B.double().t() + bias.double()
    if relu:
        ref = F.relu(ref)
    act = L.ACT_RELU if relu else L.ACT_NONE
    Ap, Bp = ops.split3(A), ops.split3(B)
    out = ops.gemm_nt_x3(Ap, Bp, bias, act=act)
    outp = ops.gemm_nt_x3(Ap, Bp, bias, act=act, out_planes=True)
    assert out.dtype == torch.float32 and out.shape == (M, N) and outp.shape == (M, 3 * N)
    assert torch.equal(_planes_to_f64(outp, N), out.double())
    ops.set_f32_mode("exact")
    exact = ops.gemm_nt(A, B, bias, act=act)
    e_x3, e_ex = _rel_rms(out, ref), _rel_rms(exact, ref)
    print(f"planes GEMM {M}x{N}x{K}: relative rms error vs float64 {e_x3:.2e} (exact f32 MFMA path {e_ex:.2e})")
    assert e_x3 < 2.0 * e_ex + 1e-8 and e_x3 < 1e-6
    torch.testing.assert_close(out.double(), ref, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("nb,H,W,Cin,N", [(2, 20, 24, 64, 128), (1, 37, 19, 128, 72), (5, 128, 128, 64, 256)])
def test_conv3x3_x3_planes(nb, H, W, Cin, N, f32_mode_restored, umr_opts):
    from unmore_amd import ops, _lib as L
    dev = _dev()
    umr_opts.setenv("UMR_NT_SPLITK", "0")   # the yardstick is ONE f32 accumulation chain per output (split-K shortens the chains)
    x = _rnd((nb, H, W, Cin), torch.float32, dev, 41)
    w = _rnd((N, 3, 3, Cin), torch.float32, dev, 42, (9 * Cin) ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 43)
    ref = F.relu(F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), bias.double(), padding=1)).permute(0, 2, 3, 1).reshape(-1, N)
    xp, wp = ops.split3(x), ops.split3(w.reshape(N, 9 * Cin))
    out = ops.gemm_nt_x3(xp, wp, bias, act=L.ACT_RELU, conv=1)
    outp = ops.gemm_nt_x3(xp, wp, bias, act=L.ACT_RELU, conv=1, out_planes=True)
    assert torch.equal(_planes_to_f64(outp, N), out.double())
    ops.set_f32_mode("exact")
    exact = ops.gemm_nt(x, w.reshape(N, 9 * Cin), bias, act=L.ACT_RELU, conv=1)
    e_x3, e_ex = _rel_rms(out, ref), _rel_rms(exact, ref)
    print(f"planes conv {nb}x{H}x{W}x{Cin}->{N}: relative rms error vs float64 {e_x3:.2e} (exact f32 MFMA path {e_ex:.2e})")
    assert e_x3 < 2.0 * e_ex + 1e-8 and e_x3 < 1e-6   # measured 6.3e-7 vs 4.1e-7: the six-term product is ~1.5x f32's own rounding
    torch.testing.assert_close(out.double(), ref, atol=2e-5, rtol=2e-5)


def test_gemm_nt_x3_fused_row_reduction():
    """the 1024 -> {1,2} head output layer folded into the planes GEMM (inference: C never stored) vs float64"""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    M, N, K = 40000 + 37, 1024, 512
    A = _rnd((M, K), torch.float32, dev, 51)
    B = _rnd((N, K), torch.float32, dev, 52, K ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 53)
    for c in (1, 2):
        rw = _rnd((c, N), torch.float32, dev, 54 + c, N ** -0.5)
        b4 = _rnd((c,), torch.float32, dev, 57)
        parts = ops.gemm_nt_x3(ops.split3(A), ops.split3(B), bias, act=L.ACT_RELU, red_w=rw)
        out = ops.head_out_finish(parts, b4, 1, 1, M, L.ACT_NONE)          # [1, c, 1, M]
        ref = F.relu(A.double() @ B.double().t() + bias.double()) @ rw.double().t() + b4.double()
        torch.testing.assert_close(out[0, :, 0, :].t().double(), ref, atol=2e-5, rtol=2e-5)


def test_gemm_nt_x3_three_term_mode(f32_mode_restored):
    """opt-in UMR_F32_X3_FAST: the plane GEMMs keep hh + hm + mh only -- relative rms error ~2^-17..2^-16 instead of ~2^-24"""
    from unmore_amd import ops
    dev = _dev()
    M, N, K = 5000, 256, 512
    A = _rnd((M, K), torch.float32, dev, 91)
    B = _rnd((N, K), torch.float32, dev, 92, K ** -0.5)
    ref = A.double() @ B.double().t()
    Ap, Bp = ops.split3(A), ops.split3(B)
    e6 = _rel_rms(ops.gemm_nt_x3(Ap, Bp), ref)
    ops.set_f32_mode("x3_fast")
    assert ops.get_f32_mode() == "x3_fast"
    e3 = _rel_rms(ops.gemm_nt_x3(Ap, Bp), ref)
    print(f"plane GEMM relative rms error vs float64: six terms {e6:.2e}, three terms {e3:.2e}")
    assert e6 < 1e-6 and 1e-6 < e3 < 2e-5


def test_gemm_nt_x3_refuses_what_it_does_not_implement():
    from unmore_amd import ops
    dev = _dev()
    Ap = ops.split3(_rnd((64, 96), torch.float32, dev, 1))      # K = 96 is not a multiple of 64
    Bp = ops.split3(_rnd((16, 96), torch.float32, dev, 2))
    with pytest.raises(RuntimeError, match="BF16X3"):
        ops.gemm_nt_x3(Ap, Bp)


@pytest.mark.parametrize("M", [64 * 577, 1000, 224 * 3 + 5])
def test_tile_height_does_not_change_results(M, umr_opts):
    """The persistent 256x256 kernel picks 256, 224 or 192 output rows per tile for plain GEMMs so that the tile count fills whole
    rounds of the chip (csrc/gemm_nt256p.hip, `bm`).  Every output element's K sum is the same instruction sequence whatever the
    tile it lands in, so all epilogue classes must give BIT-IDENTICAL results for the three heights -- and match torch."""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    D = 768
    x = _rnd((M, D), torch.bfloat16, dev, 71)
    w = _rnd((D, D), torch.bfloat16, dev, 72, D ** -0.5)
    w4 = _rnd((4 * D, D), torch.bfloat16, dev, 73, D ** -0.5)
    bias = _rnd((D,), torch.float32, dev, 74)
    b4 = _rnd((4 * D,), torch.float32, dev, 75)
    aux = _rnd((M, D), torch.bfloat16, dev, 76)
    redw = _rnd((2, 4 * D), torch.float32, dev, 77)
    umr_opts.setenv("UMR_GEMM_TILE", "256")   # small M would otherwise go to the 128x128 kernel

    def run():
        outs = [ops.gemm_nt(x, w, bias, act=L.ACT_RELU),                      # fast class
                ops.gemm_nt(x, w, bias, aux=aux),                             # residual add
                ops.gemm_nt(x, w, None, aux=aux, mask_relu=True),             # ReLU mask
                ops.gemm_nt(x, w, None, aux=aux, mask_dgelu=True),            # GELU' class
                ops.gemm_nt(x, w, bias, out_f32=True)]                        # generic class
        outs += list(ops.gemm_nt(x, w4, b4, act=L.ACT_GELU, c2_mode=2))       # GELU class, two outputs
        outs += list(ops.gemm_nt(x, w4, b4, act=L.ACT_RELU, red_w=redw))      # fused row reduction
        outs.append(ops.gemm_nt(x, w4, b4, act=L.ACT_RELU, red_w=redw, no_store=True)[1])
        torch.cuda.synchronize()
        return outs

    res = {}
    for bm in (256, 224, 192):
        umr_opts.setenv("UMR_NT256_BM", str(bm))
        res[bm] = run()
    umr_opts.delenv("UMR_NT256_BM")
    res[0] = run()   # the library's own choice
    for bm in (224, 192, 0):
        for a, b in zip(res[256], res[bm]):
            assert torch.equal(a, b), bm
    ref = F.relu(x.float() @ w.float().t() + bias)
    torch.testing.assert_close(res[224][0].float(), ref, atol=3e-2, rtol=3e-2)


def test_randomized_large_tile_vs_small_tile_kernels(umr_opts):
    """40 random plain-GEMM problems (ragged M and N, K a multiple of 64, every epilogue class) through the persistent 256x256
    kernel -- every tile height it may choose -- against the 128x128 kernel (the one the fixture tests exercise): bf16 results
    within two output ulps of each other and of torch fp32; the fp32 plane kernel against the in-register-split kernel to 2e-6."""
    import random
    from unmore_amd import ops, _lib as L
    dev = _dev()
    rng = random.Random(1234)
    for case in range(40):
        M = rng.choice([1, 17, 255, 256, 257, 1000, 4097, 20000 + rng.randrange(512)])
        N = 8 * rng.randrange(24, 130)
        K = 64 * rng.randrange(1, 9)
        kind = rng.choice(["bias_relu", "residual", "mask", "gelu2", "dgelu", "f32out", "red", "x3"])
        x = _rnd((M, K), torch.bfloat16, dev, 1000 + case)
        w = _rnd((N, K), torch.bfloat16, dev, 2000 + case, K ** -0.5)
        bias = _rnd((N,), torch.float32, dev, 3000 + case)
        aux = _rnd((M, N), torch.bfloat16, dev, 4000 + case)
        redw = _rnd((2, N), torch.float32, dev, 5000 + case, N ** -0.5)

        def run():
            if kind == "bias_relu":
                return [ops.gemm_nt(x, w, bias, act=L.ACT_RELU)]
            if kind == "residual":
                return [ops.gemm_nt(x, w, bias, aux=aux)]
            if kind == "mask":
                return [ops.gemm_nt(x, w, None, aux=aux, mask_relu=True)]
            if kind == "gelu2":
                return list(ops.gemm_nt(x, w, bias, act=L.ACT_GELU, c2_mode=2))
            if kind == "dgelu":
                return [ops.gemm_nt(x, w, None, aux=aux, mask_dgelu=True)]
            if kind == "f32out":
                return [ops.gemm_nt(x, w, bias, out_f32=True)]
            return []

        if kind == "x3":
            xf, wf = x.float() * 1.37, w.float() * 0.91
            a = ops.gemm_nt_x3(ops.split3(xf), ops.split3(wf), bias, act=L.ACT_RELU)
            b = ops.gemm_nt(xf, wf, bias, act=L.ACT_RELU)
            torch.testing.assert_close(a, b, atol=2e-6 * float(b.abs().max() + 1), rtol=0)
            continue
        if kind == "red":
            umr_opts.setenv("UMR_GEMM_TILE", "256")
            h, parts = ops.gemm_nt(x, w, bias, act=L.ACT_RELU, red_w=redw)
            out = ops.head_out_finish(parts, torch.zeros(2, device=dev), 1, 1, M, L.ACT_NONE)[0, :, 0, :].t()
            ref = h.float() @ redw.t()
            torch.testing.assert_close(out, ref, atol=2e-3, rtol=2e-3)
            continue
        umr_opts.setenv("UMR_GEMM_TILE", "128")
        small = run()
        for bm in ("0", "256", "224", "192"):
            umr_opts.setenv("UMR_GEMM_TILE", "256")
            if bm == "0":
                umr_opts.delenv("UMR_NT256_BM", raising=False)
            else:
                umr_opts.setenv("UMR_NT256_BM", bm)
            big = run()
            for a, b in zip(small, big):
                # bf16: the large-tile path rounds acc + bias to bf16 before a residual add / mask (documented in include/umr.h): two output ulps
                tol = 1e-5 if a.dtype == torch.float32 else 2.0 ** -6
                assert ((a.float() - b.float()).abs() <= tol * (b.float().abs() + 1)).all(), (case, kind, M, N, K, bm)
        umr_opts.delenv("UMR_NT256_BM", raising=False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,force", [(1300, 1024, 4096, None), (1300, 1024, 1024, None), (300, 200, 1096, "3"), (129, 136, 776, "8"),
                                         (700, 3072, 1024, "2")])
def test_gemm_nt_split_k(dtype, M, N, K, force, f32_mode_restored, umr_opts):
    """Split-K of the 128x128 kernel (umr_gemm_nt_ws: few tiles, long K -- the reference recipe's 1300-token projections): against
    fp64, against the unsplit launch, twice in a row (counters are left zero; the fixed-order slab sum is bitwise reproducible),
    through every epilogue class (bias, residual, ReLU mask, GELU + saved pre-activation), with ragged M / N and a K tail."""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    A = _rnd((M, K), dtype, dev, 1)
    B = _rnd((N, K), dtype, dev, 2, K ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 3)
    aux = _rnd((M, N), dtype, dev, 4)
    ref = A.double() @ B.double().t() + bias.double()

    def run():
        o1 = ops.gemm_nt(A, B, bias)
        o2, pre = ops.gemm_nt(A, B, bias, act=L.ACT_GELU, c2_mode=2) if dtype == torch.bfloat16 else (None, None)
        o3 = ops.gemm_nt(A, B, bias, aux=aux)
        o4 = ops.gemm_nt(A, B, None, aux=aux, mask_relu=True)
        return [t for t in (o1, o2, pre, o3, o4) if t is not None]

    if force:
        umr_opts.setenv("UMR_NT_SPLITK", force)
    else:
        umr_opts.delenv("UMR_NT_SPLITK", raising=False)
    for mode in (("x3", "exact") if dtype == torch.float32 else (None,)):
        if mode:
            ops.set_f32_mode(mode)
        # the launch really runs split: the library reports the number of K ranges it will use
        nsplit = ops.gemm_nt(A, B, bias, query_splits=True)
        if force:
            assert 1 < nsplit <= int(force), (nsplit, force)      # (no range is left empty: 13 K-tiles in 8 ranges run as 7)
        elif K >= 4096 or dtype == torch.float32:
            assert nsplit > 1, nsplit
        first = run()
        second = run()
        for a, b in zip(first, second):
            assert torch.equal(a, b)
        umr_opts.setenv("UMR_NT_SPLITK", "0")
        unsplit = run()
        if force:
            umr_opts.setenv("UMR_NT_SPLITK", force)
        else:
            umr_opts.delenv("UMR_NT_SPLITK", raising=False)
        torch.testing.assert_close(first[0].double(), ref, **_tol(dtype))
        torch.testing.assert_close(first[-2].double(), ref + aux.double(), **_tol(dtype))
        torch.testing.assert_close(first[-1].double(), (ref - bias.double()) * (aux.double() > 0), **_tol(dtype))
        if dtype == torch.bfloat16:
            torch.testing.assert_close(first[2].double(), ref, **_tol(dtype))
            torch.testing.assert_close(first[1].double(), F.gelu(ref), **_tol(dtype))
        for a, b in zip(first, unsplit):     # same products, another f32 summation order: a rounding apart at most
            torch.testing.assert_close(a.float(), b.float(), atol=2e-5 if dtype == torch.float32 else 2e-2, rtol=2 ** -7 if dtype == torch.bfloat16 else 2e-5)
    assert ops._sk_cache, "the split-K workspace was never requested"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_split_k_back_to_back_with_different_operands(dtype, umr_opts):
    """Two split launches in a row on the SAME workspace with different A, each checked against float64: a last arriver that read a
    slab of the previous launch (stale line, counter not back at zero) would reproduce the first result or mix the two."""
    from unmore_amd import ops
    dev = _dev()
    umr_opts.setenv("UMR_NT_SPLITK", "4")
    M, N, K = 300, 264, 2048
    B = _rnd((N, K), dtype, dev, 2, K ** -0.5)
    outs, refs = [], []
    As = [_rnd((M, K), dtype, dev, 10 + i) for i in range(6)]
    assert ops.gemm_nt(As[0], B, query_splits=True) == 4
    for A in As:          # enqueued back to back, no synchronisation in between
        outs.append(ops.gemm_nt(A, B))
    for A, o in zip(As, outs):
        torch.testing.assert_close(o.double(), A.double() @ B.double().t(), **_tol(dtype))
    assert not torch.equal(outs[0], outs[1])


def test_split_k_bad_ticket_is_counted_not_trapped(umr_opts):
    """A violated workspace precondition (a tile counter that is not zero on first use) used to end in __builtin_trap(), i.e. in the
    loss of the process's device context.  Now (include/umr.h): the kernel counts the out-of-range ticket in the workspace's error word,
    heals the counter and leaves that ONE tile unwritten; umr_gemm_nt_ws_status reads and clears the count; the next launch is clean."""
    from unmore_amd import ops
    umr_opts.setenv("UMR_GEMM_TILE", "128")
    umr_opts.setenv("UMR_NT_SPLITK", "2")
    dev = _dev()
    A, B = _rnd((300, 2048), torch.bfloat16, dev, 1), _rnd((200, 2048), torch.bfloat16, dev, 2, 2048 ** -0.5)
    assert ops.gemm_nt(A, B, None, query_splits=True) == 2
    ref = ops.gemm_nt(A, B, None)
    assert ops.splitk_bad_tickets() == 0
    ws = ops._splitk_workspace(A.device)
    ws[:4].view(torch.int32)[0] = 1000                      # counter of one tile: garbage, as after an aborted launch
    out = ops.gemm_nt(A, B, None)
    torch.cuda.synchronize()                                # the context survives
    assert ops.splitk_bad_tickets() in (1, 2)               # one or both workgroups of the tile drew a ticket before the counter was healed
    assert ops.splitk_bad_tickets() == 0                    # read-and-clear
    bad_tiles = sum(int(not torch.equal(out[r:r + 128, c:c + 128], ref[r:r + 128, c:c + 128])) for r in range(0, 300, 128) for c in range(0, 200, 128))
    assert bad_tiles <= 1                                   # only the poisoned tile may be unwritten
    assert int(ws[:16384].view(torch.int32).abs().sum()) == 0   # every counter healed, error word cleared
    assert torch.equal(ops.gemm_nt(A, B, None), ref)
    assert ops.splitk_bad_tickets() == 0


def test_split_k_equals_the_fence_build(tmp_path):
    """The default hand-over (sc1 slab traffic + acknowledged stores + relaxed ticket: hardware ordering, csrc/gemm_nt.hip) against
    libumr_fence.so (agent-scope release / acquire fences around the ticket: the textbook protocol, and a trap on a bad ticket),
    run in a child process on the same seeded problems: results must be bit-identical."""
    import os
    import subprocess
    import sys
    from unmore_amd import _lib as L
    fence = os.path.join(os.path.dirname(L.LIB_PATH), "libumr_fence.so")
    assert os.path.exists(fence), "libumr_fence.so missing: build() makes it (make -C unmore_amd/csrc)"
    script = r"""
import sys, torch
sys.path.insert(0, %r)
from unmore_amd import ops
import os
def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda().to(dtype)
outs = []
for dtype in (torch.float32, torch.bfloat16):
    for (M, N, K, force) in ((1300, 1024, 4096, None), (300, 200, 1096, "3"), (129, 136, 776, "8")):
        if force: ops.set_debug_option("UMR_NT_SPLITK", force)
        else: ops.set_debug_option("UMR_NT_SPLITK", None)
        A, B, bias, aux = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5), rnd((N,), torch.float32, 3), rnd((M, N), dtype, 4)
        assert ops.gemm_nt(A, B, bias, query_splits=True) > 1
        for rep in range(3):
            outs.append(ops.gemm_nt(A, B, bias).float().cpu())
            outs.append(ops.gemm_nt(A, B, None, aux=aux, mask_relu=True).float().cpu())
    ops.set_debug_option("UMR_NT_SPLITK", "5")
    x, w = rnd((2, 14, 9, 128), dtype, 5), rnd((200, 9 * 128), dtype, 6, 0.03)
    outs.append(ops.gemm_nt(x, w, None, conv=1).float().cpu())
torch.save(outs, sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(L.__file__)))
    res = {}
    # ... and the same switch at RUN time: the default library with UMR_SPLITK_FENCE=1 takes the fenced hand-over without a rebuild
    for tag, lib, fe in (("default", L.LIB_PATH, None), ("fence", fence, None), ("runtime", L.LIB_PATH, "1")):
        env = dict(os.environ, UMR_LIB=lib)
        env.pop("UMR_NT_SPLITK", None)
        env.pop("UMR_SPLITK_FENCE", None)
        if fe:
            env["UMR_SPLITK_FENCE"] = fe
        out = tmp_path / f"{tag}.pt"
        r = subprocess.run([sys.executable, "-c", script, str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = torch.load(out)
    assert len(res["default"]) == len(res["fence"]) == len(res["runtime"]) > 30
    for a, b, c in zip(res["default"], res["fence"], res["runtime"]):
        assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nb,H,W,Cin,N,stride,force", [(2, 7, 7, 256, 256, 1, None), (2, 14, 9, 128, 200, 1, "5"), (3, 12, 12, 128, 128, 2, "3"),
                                                        (20, 16, 16, 256, 256, 1, None)])
def test_conv3x3_split_k(dtype, nb, H, W, Cin, N, stride, force, f32_mode_restored, umr_opts):
    """Split-K of the implicit 3x3 conv on small maps (a K range starts in the middle of the tap sequence): against torch conv2d in
    fp64, twice (bitwise reproducible), and against the unsplit launch."""
    from unmore_amd import ops
    dev = _dev()
    x = _rnd((nb, H, W, Cin), dtype, dev, 1)
    w = _rnd((N, Cin, 3, 3), dtype, dev, 2, (9 * Cin) ** -0.5)
    bias = _rnd((N,), torch.float32, dev, 3)
    wp = w.permute(0, 2, 3, 1).reshape(N, 9 * Cin).contiguous()
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, N)
    if force:
        umr_opts.setenv("UMR_NT_SPLITK", force)
    else:
        umr_opts.delenv("UMR_NT_SPLITK", raising=False)
    for mode in (("x3", "exact") if dtype == torch.float32 else (None,)):
        if mode:
            ops.set_f32_mode(mode)
        a = ops.gemm_nt(x, wp, bias, conv=stride)
        b = ops.gemm_nt(x, wp, bias, conv=stride)
        assert torch.equal(a, b)
        torch.testing.assert_close(a.double().reshape(-1, N), ref, **_tol(dtype))
        umr_opts.setenv("UMR_NT_SPLITK", "0")
        c = ops.gemm_nt(x, wp, bias, conv=stride)
        if force:
            umr_opts.setenv("UMR_NT_SPLITK", force)
        else:
            umr_opts.delenv("UMR_NT_SPLITK", raising=False)
        torch.testing.assert_close(a.float(), c.float(), atol=2e-5 if dtype == torch.float32 else 2e-2, rtol=2 ** -7 if dtype == torch.bfloat16 else 2e-5)


def test_cu_budget_does_not_change_results(umr_opts):
    """umr_set_cu_budget (include/umr.h): the persistent 256x256 grids launch exactly `budget` workgroups, leaving the other CUs to
    kernels that run beside them (RCCL's bucket all-reduces during a data-parallel backward).  Tile height and K-split are planned on
    the device's CU count, every output element's K sum is the same instruction sequence in any workgroup: plain GEMMs of every
    epilogue class, the 3x3 conv forms, the plane (fp32-grade) GEMMs incl. a K-split one, and a whole train step must be
    BIT-IDENTICAL for budgets 0 (all), 208, 64 and 7 (a grid that is not a multiple of the 8 XCDs)."""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    D, M = 768, 2500
    x = _rnd((M, D), torch.bfloat16, dev, 171)
    w = _rnd((D, D), torch.bfloat16, dev, 172, D ** -0.5)
    w4 = _rnd((4 * D, D), torch.bfloat16, dev, 173, D ** -0.5)
    bias = _rnd((D,), torch.float32, dev, 174)
    b4 = _rnd((4 * D,), torch.float32, dev, 175)
    aux = _rnd((M, D), torch.bfloat16, dev, 176)
    redw = _rnd((2, 4 * D), torch.float32, dev, 177)
    img = _rnd((2, 40, 56, 256), torch.bfloat16, dev, 178)
    wc = _rnd((256, 9 * 256), torch.bfloat16, dev, 179, (9 * 256) ** -0.5)
    cmask = _rnd((2 * 40 * 56, 256), torch.bfloat16, dev, 180)
    xf = _rnd((1300, 1024), torch.float32, dev, 181)
    wf = _rnd((1024, 1024), torch.float32, dev, 182, 1024 ** -0.5)
    xp, wp = ops.split3(xf), ops.split3(wf)
    umr_opts.setenv("UMR_GEMM_TILE", "256")

    def run():
        outs = [ops.gemm_nt(x, w, bias, act=L.ACT_RELU), ops.gemm_nt(x, w, bias, aux=aux), ops.gemm_nt(x, w, None, aux=aux, mask_relu=True),
                ops.gemm_nt(x, w, None, aux=aux, mask_dgelu=True), ops.gemm_nt(x, w, bias, out_f32=True)]
        outs += list(ops.gemm_nt(x, w4, b4, act=L.ACT_GELU, c2_mode=2))
        outs += list(ops.gemm_nt(x, w4, b4, act=L.ACT_RELU, red_w=redw))
        outs.append(ops.gemm_nt(img, wc, bias[:256].contiguous(), conv=1, act=L.ACT_RELU))
        outs.append(ops.gemm_nt(img, wc, None, conv=1, aux=cmask, mask_relu=True))
        outs.append(ops.gemm_nt_x3(xp, wp, None))                           # plane GEMM at the reference recipe's token count (K-split)
        outs.append(ops.gemm_nt_x3(xp, wp, None, out_planes=True))
        torch.cuda.synchronize()
        return outs

    res = {}
    try:
        for budget in (0, 208, 64, 7):
            ops.set_cu_budget(budget)
            assert L.lib().umr_get_cu_budget() == budget
            res[budget] = run()
    finally:
        ops.set_cu_budget(0)
    for budget in (208, 64, 7):
        for i, (a, b) in enumerate(zip(res[0], res[budget])):
            assert torch.equal(a, b), (budget, i)
    torch.testing.assert_close(res[64][0].float(), F.relu(x.float() @ w.float().t() + bias), atol=3e-2, rtol=3e-2)
    torch.testing.assert_close(res[7][-2], xf @ wf.t(), atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_xcd_tile_order_does_not_change_results(dtype, umr_opts):
    """The 128x128 NT kernel walks an XCD's run of tiles n-fastest or m-fastest (csrc/gemm_nt.hip: the operand every XCD has to fetch
    whole should be the smaller one; UMR_NT_ORDER forces an order: umr_set_debug_option).  Which workgroup computes a tile never changes the
    tile's arithmetic: both orders and the library's own choice are bit-identical, with and without split-K, ragged edges included."""
    from unmore_amd import ops, _lib as L
    dev = _dev()
    umr_opts.setenv("UMR_GEMM_TILE", "128")
    for (M, N, K, force) in ((1300, 4096, 1024, None), (1300, 1024, 4096, None), (300, 200, 1096, "3"), (129, 136, 776, None), (70, 1544, 136, None)):
        if force:
            umr_opts.setenv("UMR_NT_SPLITK", force)
        else:
            umr_opts.delenv("UMR_NT_SPLITK", raising=False)
        A, B = _rnd((M, K), dtype, dev, 11), _rnd((N, K), dtype, dev, 12, K ** -0.5)
        bias, aux = _rnd((N,), torch.float32, dev, 13), _rnd((M, N), dtype, dev, 14)
        res = {}
        for order in ("n", "m", None):
            if order:
                umr_opts.setenv("UMR_NT_ORDER", order)
            else:
                umr_opts.delenv("UMR_NT_ORDER", raising=False)
            res[order] = (ops.gemm_nt(A, B, bias, act=L.ACT_RELU), ops.gemm_nt(A, B, None, aux=aux, mask_relu=True), ops.gemm_nt(A, B, bias, out_f32=True))
            torch.cuda.synchronize()
        for order in ("m", None):
            for a, b in zip(res["n"], res[order]):
                assert torch.equal(a, b), (M, N, K, order)
        ref = torch.relu(A.float() @ B.float().t() + bias)
        torch.testing.assert_close(res["m"][0].float(), ref, **_tol(dtype))
