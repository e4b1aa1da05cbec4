"""The algebra behind the inference form of the boundary-distance head, in float64 on the CPU (no kernels): the reference's four
convolutions without a non-linearity between them (models/objectness_net.py:128-135: 1x1 256->512, 3x3 512->512 zero-padded,
1x1 512->1024, 1x1 1024->1, then tanh) equal ONE 3x3 convolution 256->1 plus a border-dependent bias, and that convolution's nine
tap products commute with the bilinear x2 resize in front of the head (models/dpt/models.py:70-72) -- the two identities
unmore_amd/engine.py::_linear_head_weights and _linear_head_forward_lowres + csrc/linear_head.hip::lh_gather9_kernel rest on."""
import torch
import torch.nn.functional as F


def _weights(C, C1, C3, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    return dict(W1=r(C1, C, 1, 1) * C ** -0.5, b1=r(C1), W2=r(C1, C1, 3, 3) * (9 * C1) ** -0.5, b2=r(C1),
                W3=r(C3, C1, 1, 1) * C1 ** -0.5, b3=r(C3), W4=r(1, C3, 1, 1) * C3 ** -0.5, b4=r(1))


def _factored(x, w):
    h = F.conv2d(x, w["W1"], w["b1"])
    h = F.conv2d(h, w["W2"], w["b2"], padding=1)
    h = F.conv2d(h, w["W3"], w["b3"])
    return torch.tanh(F.conv2d(h, w["W4"], w["b4"]))


def _collapse(w):
    """the engine's formulas (engine.Engine._linear_head_weights), element for element"""
    W1, W2, W3, W4 = w["W1"][:, :, 0, 0], w["W2"], w["W3"][:, :, 0, 0], w["W4"][:, :, 0, 0]
    u = (W4 @ W3)[0]                                            # [C1]
    Vc = torch.einsum("o,oit->it", u, W2.reshape(W2.shape[0], W2.shape[1], 9))   # [ci][t]
    Kw = torch.einsum("it,ic->tc", Vc, W1)                      # [9][C]
    tb = torch.zeros(10, dtype=torch.float64)
    tb[:9] = torch.einsum("it,i->t", Vc, w["b1"])
    tb[9] = u @ w["b2"] + (W4 @ w["b3"])[0] + w["b4"][0]
    return Kw, tb


def _gather9(taps, tb):
    """lh_gather9_kernel: out(q) = tanh(sum over taps t whose position q + off_t lies inside the image of (taps[q + off_t][t] + tb[t]) + tb[9])"""
    B, H, W, _ = taps.shape
    out = torch.full((B, H, W), float(tb[9]), dtype=torch.float64)
    for t in range(9):
        dy, dx = t // 3 - 1, t % 3 - 1
        ys, xs = slice(max(0, -dy), H - max(0, dy)), slice(max(0, -dx), W - max(0, dx))
        yd, xd = slice(max(0, dy), H + min(0, dy)), slice(max(0, dx), W + min(0, dx))
        out[:, ys, xs] += taps[:, yd, xd, t] + tb[t]
    return torch.tanh(out).unsqueeze(1)


def test_four_convolutions_equal_one_3x3_plus_border_bias():
    w = _weights(16, 24, 40, 0)
    x = torch.randn(2, 16, 7, 9, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    Kw, tb = _collapse(w)
    taps = torch.einsum("bchw,tc->bhwt", x, Kw)                 # taps[q][t] = Kw[t] . x(q)
    ref, got = _factored(x, w), _gather9(taps, tb)
    assert (ref - got).abs().max().item() < 1e-12
    # the border bias matters: with the first conv's bias propagated everywhere (a plain 3x3 conv with ONE bias) the border rows differ
    w33 = Kw.reshape(3, 3, 16).permute(2, 0, 1).unsqueeze(0).contiguous()       # [1, C, ky, kx]
    plain = torch.tanh(F.conv2d(x, w33, None, padding=1) + tb[:9].sum() + tb[9])
    assert (ref - plain)[:, :, 1:-1, 1:-1].abs().max().item() < 1e-12 and (ref - plain).abs().max().item() > 1e-6


def test_tap_products_commute_with_the_resize():
    """out = gather9(resize(path . Kw^T)): the 16-column GEMM on the map BEFORE the x2 resize (align_corners=True, models.py:70-72)"""
    w = _weights(16, 24, 40, 2)
    path = torch.randn(2, 16, 5, 6, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    Kw, tb = _collapse(w)
    x = F.interpolate(path, scale_factor=2, mode="bilinear", align_corners=True)
    ref = _factored(x, w)
    taps_small = torch.einsum("bchw,tc->bthw", path, Kw)        # nine channels on the small map
    taps = F.interpolate(taps_small, scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    assert (ref - _gather9(taps, tb)).abs().max().item() < 1e-12
