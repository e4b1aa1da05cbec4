"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain-PyTorch (CPU, fp32, NCHW) restatement of unMORE's stage-1 ObjectnessNet
hot path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this file; the product (`unmore_amd/`) never does.

Parity status: PINNED for everything that lives in the reference tree (DPT
readout / reassemble / fusion / heads, `ObjectnessNet.forward`) -- checked against
fixtures under `tests/golden/` produced by running the reference's own modules in
the build container (`tests/golden/make_golden.py`).  UNPINNED at two third-party
boundaries that are absent from /root/reference and from this image:
  * timm==1.0.15 `VisionTransformer` Block/Attention/Mlp (requirements.txt:13;
    call sites models/dpt/vit.py:196-199,518,534) -- restated in `vit_block`;
  * torchmetrics==1.5.2 `functional.image_gradients` (requirements.txt:15; call
    sites train_objectness_net.py:236,239) -- restated in `image_gradients`.
The reference ships no tests / golden vectors of its own.

Every function cites the reference file:line it follows (paths relative to the
reference root).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# configurations (models/dpt/models.py:43-47, models/dpt/blocks.py:24-54,
# models/dpt/vit.py:515-543).  'dpt_large' is the only live reference config;
# the others are the BASELINE.json extensions.
# --------------------------------------------------------------------------
CONFIGS = {
    "dpt_large": dict(D=1024, depth=24, heads=16, patch=16, pos_grid=24,
                      hooks=[5, 11, 17, 23], features=[256, 512, 1024, 1024]),
    "dpt_base": dict(D=768, depth=12, heads=12, patch=16, pos_grid=24,
                     hooks=[2, 5, 8, 11], features=[96, 192, 384, 768]),
    "dpt_small": dict(D=384, depth=12, heads=6, patch=16, pos_grid=24,
                      hooks=[2, 5, 8, 11], features=[48, 96, 192, 384]),
    "dpt_large14": dict(D=1024, depth=24, heads=16, patch=14, pos_grid=37,
                        hooks=[5, 11, 17, 23], features=[256, 512, 1024, 1024]),
    # test-only miniature (same wiring as vitb16_384, head dim 64)
    "dpt_tiny": dict(D=128, depth=4, heads=2, patch=16, pos_grid=24,
                     hooks=[0, 1, 2, 3], features=[32, 64, 128, 128]),
}
SCRATCH = 256          # models/objectness_net.py:66 (features=256)
NUM_CLASSES = 1000     # timm ViT head (kept for state-dict key parity)


def head_layout(args_use_bg_sdf=True, sdf_activation="tanh"):
    """Sequential indices of the conv layers and the activation kinds.

    models/objectness_net.py:109-117 (center head: convs at 0,2,4,6 with ReLU
    between) and :119-164 (sdf head variants)."""
    center = dict(conv_idx=[0, 2, 4, 6], relu=True, final=None)
    if args_use_bg_sdf:
        if sdf_activation == "sine":
            sdf = dict(conv_idx=[0, 1, 2, 3], relu=False, final="sine")
        elif sdf_activation == "tanh":
            sdf = dict(conv_idx=[0, 1, 2, 3], relu=False, final="tanh")
        elif sdf_activation is None:
            sdf = dict(conv_idx=[0, 1, 2, 3], relu=False, final=None)
        elif sdf_activation == "relu":
            sdf = dict(conv_idx=[0, 2, 4, 6], relu=True, final=None)
        else:
            raise NotImplementedError
    else:
        sdf = dict(conv_idx=[0, 2, 4, 6], relu=True, final=None)
    return center, sdf


def state_dict_spec(cfg, use_bg_sdf=True, sdf_activation="tanh"):
    """name -> shape, in the reference's registration order."""
    D, depth, p, g = cfg["D"], cfg["depth"], cfg["patch"], cfg["pos_grid"]
    Fs = cfg["features"]
    s = OrderedDict()
    m = "backbone.pretrained.model."
    s[m + "cls_token"] = (1, 1, D)
    s[m + "pos_embed"] = (1, 1 + g * g, D)
    s[m + "patch_embed.proj.weight"] = (D, 3, p, p)
    s[m + "patch_embed.proj.bias"] = (D,)
    for i in range(depth):
        b = m + f"blocks.{i}."
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.qkv.bias"] = (3 * D,)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (4 * D, D)
        s[b + "mlp.fc1.bias"] = (4 * D,)
        s[b + "mlp.fc2.weight"] = (D, 4 * D)
        s[b + "mlp.fc2.bias"] = (D,)
    s[m + "norm.weight"] = (D,)
    s[m + "norm.bias"] = (D,)
    s[m + "head.weight"] = (NUM_CLASSES, D)
    s[m + "head.bias"] = (NUM_CLASSES,)
    pp = "backbone.pretrained."
    for k in range(4):
        a = pp + f"act_postprocess{k + 1}."
        s[a + "0.project.0.weight"] = (D, 2 * D)
        s[a + "0.project.0.bias"] = (D,)
        s[a + "3.weight"] = (Fs[k], D, 1, 1)
        s[a + "3.bias"] = (Fs[k],)
        if k == 0:
            s[a + "4.weight"] = (Fs[0], Fs[0], 4, 4)
            s[a + "4.bias"] = (Fs[0],)
        elif k == 1:
            s[a + "4.weight"] = (Fs[1], Fs[1], 2, 2)
            s[a + "4.bias"] = (Fs[1],)
        elif k == 3:
            s[a + "4.weight"] = (Fs[3], Fs[3], 3, 3)
            s[a + "4.bias"] = (Fs[3],)
    sc = "backbone.scratch."
    for k in range(4):
        s[sc + f"layer{k + 1}_rn.weight"] = (SCRATCH, Fs[k], 3, 3)
    for k in (1, 2, 3, 4):
        r = sc + f"refinenet{k}."
        s[r + "out_conv.weight"] = (SCRATCH, SCRATCH, 1, 1)
        s[r + "out_conv.bias"] = (SCRATCH,)
        for u in (1, 2):
            for c in (1, 2):
                s[r + f"resConfUnit{u}.conv{c}.weight"] = (SCRATCH, SCRATCH, 3, 3)
                s[r + f"resConfUnit{u}.conv{c}.bias"] = (SCRATCH,)
    center, sdf = head_layout(use_bg_sdf, sdf_activation)
    chans = [(512, SCRATCH, 1), (512, 512, 3), (1024, 512, 1)]
    for name, lay, cout in (("center_field_prediction_head", center, 2),
                            ("sdf_prediction_head", sdf, 1)):
        for li, idx in enumerate(lay["conv_idx"]):
            if li < 3:
                co, ci, k = chans[li]
            else:
                co, ci, k = cout, 1024, 1
            s[f"{name}.{idx}.weight"] = (co, ci, k, k)
            s[f"{name}.{idx}.bias"] = (co,)
    return s


# --------------------------------------------------------------------------
# third-party restatements
# --------------------------------------------------------------------------
def vit_block(x, sd, b, heads):
    """timm==1.0.15 `Block.forward` (pre-norm; LayerScale/DropPath identity):
    x += proj(SDPA(qkv(LN1(x)))); x += fc2(GELU_erf(fc1(LN2(x)))).
    LayerNorm eps 1e-6; scale head_dim**-0.5.  (vit.py:196-197)."""
    B, N, D = x.shape
    hd = D // heads
    h = F.layer_norm(x, (D,), sd[b + "norm1.weight"], sd[b + "norm1.bias"], eps=1e-6)
    qkv = F.linear(h, sd[b + "attn.qkv.weight"], sd[b + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * (hd ** -0.5)) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, D)
    x = x + F.linear(o, sd[b + "attn.proj.weight"], sd[b + "attn.proj.bias"])
    h = F.layer_norm(x, (D,), sd[b + "norm2.weight"], sd[b + "norm2.bias"], eps=1e-6)
    h = F.gelu(F.linear(h, sd[b + "mlp.fc1.weight"], sd[b + "mlp.fc1.bias"]))
    x = x + F.linear(h, sd[b + "mlp.fc2.weight"], sd[b + "mlp.fc2.bias"])
    return x


def image_gradients(img):
    """torchmetrics==1.5.2 functional.image_gradients: forward differences,
    dy[i,j]=x[i+1,j]-x[i,j] (last row 0), dx[i,j]=x[i,j+1]-x[i,j] (last col 0)."""
    dy = torch.zeros_like(img)
    dx = torch.zeros_like(img)
    dy[..., :-1, :] = img[..., 1:, :] - img[..., :-1, :]
    dx[..., :, :-1] = img[..., :, 1:] - img[..., :, :-1]
    return dy, dx


# --------------------------------------------------------------------------
# reference restatement
# --------------------------------------------------------------------------
def resize_pos_embed(posemb, gs_h, gs_w):
    """models/dpt/vit.py:148-162 (bilinear, align_corners=False, grid part only)."""
    tok, grid = posemb[:, :1], posemb[0, 1:]
    gs_old = int(math.sqrt(len(grid)))
    grid = grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(gs_h, gs_w), mode="bilinear")
    grid = grid.permute(0, 2, 3, 1).reshape(1, gs_h * gs_w, -1)
    return torch.cat([tok, grid], dim=1)


def forward_flex(sd, x, cfg, inter=None):
    """models/dpt/vit.py:165-201; returns the 4 hooked block outputs (:234-237)."""
    m = "backbone.pretrained.model."
    p = cfg["patch"]
    B, _, h, w = x.shape
    pos = resize_pos_embed(sd[m + "pos_embed"], h // p, w // p)
    t = F.conv2d(x, sd[m + "patch_embed.proj.weight"], sd[m + "patch_embed.proj.bias"], stride=p)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd[m + "cls_token"].expand(B, -1, -1), t), dim=1)
    t = t + pos
    if inter is not None:
        inter["tokens0"] = t
    acts = []
    for i in range(cfg["depth"]):
        t = vit_block(t, sd, m + f"blocks.{i}.", cfg["heads"])
        if i in cfg["hooks"]:
            acts.append(t)
    # final norm (vit.py:199) is computed by the reference and discarded (:107).
    return acts


def project_readout(x, w, b):
    """models/dpt/vit.py:86-90: GELU(Linear(cat(tok[:,1:], cls.expand)))."""
    readout = x[:, 0].unsqueeze(1).expand_as(x[:, 1:])
    feats = torch.cat((x[:, 1:], readout), -1)
    return F.gelu(F.linear(feats, w, b))


def reassemble(sd, acts, cfg, h, w, inter=None):
    """models/dpt/vit.py:104-145 with act_postprocess1..4 (:259-336)."""
    pp = "backbone.pretrained."
    p = cfg["patch"]
    outs = []
    for k in range(4):
        a = pp + f"act_postprocess{k + 1}."
        y = project_readout(acts[k], sd[a + "0.project.0.weight"], sd[a + "0.project.0.bias"])
        y = y.transpose(1, 2)
        y = y.unflatten(2, (h // p, w // p))
        y = F.conv2d(y, sd[a + "3.weight"], sd[a + "3.bias"])
        if k == 0:
            y = F.conv_transpose2d(y, sd[a + "4.weight"], sd[a + "4.bias"], stride=4)
        elif k == 1:
            y = F.conv_transpose2d(y, sd[a + "4.weight"], sd[a + "4.bias"], stride=2)
        elif k == 3:
            y = F.conv2d(y, sd[a + "4.weight"], sd[a + "4.bias"], stride=2, padding=1)
        outs.append(y)
        if inter is not None:
            inter[f"layer_{k + 1}"] = y
    return outs


# ---- ReLU sites.  Every ReLU of the path has a name (RCU: "<refinenetK.resConfUnitJ.>relu_in" / "relu_mid"; heads:
# "<head>.relu0..2").  Tests can run the oracle with the DECISIONS of another implementation imposed at every site
# (`with relu_hook(fn)`: fn(x, site) replaces F.relu) -- the network then is the same piecewise-linear function on the same
# piece, and gradients can be compared without the noise of masks decided differently within rounding of zero.
_RELU_HOOK = None


class relu_hook:
    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _RELU_HOOK
        self.prev, _RELU_HOOK = _RELU_HOOK, self.fn

    def __exit__(self, *exc):
        global _RELU_HOOK
        _RELU_HOOK = self.prev


def _relu(x, site):
    return F.relu(x) if _RELU_HOOK is None else _RELU_HOOK(x, site)


def rcu(x, sd, r):
    """models/dpt/blocks.py:290-313 (bn=False, ReLU out of place)."""
    out = _relu(x, r + "relu_in")
    out = F.conv2d(out, sd[r + "conv1.weight"], sd[r + "conv1.bias"], padding=1)
    out = _relu(out, r + "relu_mid")
    out = F.conv2d(out, sd[r + "conv2.weight"], sd[r + "conv2.bias"], padding=1)
    return out + x


def _up2(x, size=None):
    if size is None:
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


def fusion(sd, r, x0, x1=None, up_size=None):
    """models/dpt/blocks.py:362-383."""
    out = x0
    if x1 is not None:
        out = out + rcu(x1, sd, r + "resConfUnit1.")
    out = rcu(out, sd, r + "resConfUnit2.")
    out = _up2(out, up_size)
    return F.conv2d(out, sd[r + "out_conv.weight"], sd[r + "out_conv.bias"])


def dpt_forward(sd, x, cfg, inter=None):
    """models/dpt/models.py:74-94.  For patch sizes whose pyramid is not an
    exact power-of-two chain (dpt_large14, an extension: SURVEY section 9) every
    fusion block upsamples to the next skip's size and the last upsample goes to
    the input size; for the reference's patch-16 configs these coincide with x2."""
    sc = "backbone.scratch."
    B, _, h, w = x.shape
    acts = forward_flex(sd, x, cfg, inter)
    l1, l2, l3, l4 = reassemble(sd, acts, cfg, h, w, inter)
    l1 = F.conv2d(l1, sd[sc + "layer1_rn.weight"], padding=1)
    l2 = F.conv2d(l2, sd[sc + "layer2_rn.weight"], padding=1)
    l3 = F.conv2d(l3, sd[sc + "layer3_rn.weight"], padding=1)
    l4 = F.conv2d(l4, sd[sc + "layer4_rn.weight"], padding=1)
    exact = cfg["patch"] == 16
    p4 = fusion(sd, sc + "refinenet4.", l4, None, None if exact else l3.shape[-2:])
    p3 = fusion(sd, sc + "refinenet3.", p4, l3, None if exact else l2.shape[-2:])
    p2 = fusion(sd, sc + "refinenet2.", p3, l2, None if exact else l1.shape[-2:])
    p1 = fusion(sd, sc + "refinenet1.", p2, l1, None)
    if inter is not None:
        inter.update(path_4=p4, path_3=p3, path_2=p2, path_1=p1)
    out = _up2(p1, None if exact else (h, w))
    if inter is not None:
        inter["feat"] = out
    return out


def head_forward(sd, name, feat, layout):
    """models/objectness_net.py:109-117 / :128-135."""
    idx = layout["conv_idx"]
    y = F.conv2d(feat, sd[f"{name}.{idx[0]}.weight"], sd[f"{name}.{idx[0]}.bias"])
    if layout["relu"]:
        y = _relu(y, f"{name}.relu0")
    y = F.conv2d(y, sd[f"{name}.{idx[1]}.weight"], sd[f"{name}.{idx[1]}.bias"], padding=1)
    if layout["relu"]:
        y = _relu(y, f"{name}.relu1")
    y = F.conv2d(y, sd[f"{name}.{idx[2]}.weight"], sd[f"{name}.{idx[2]}.bias"])
    if layout["relu"]:
        y = _relu(y, f"{name}.relu2")
    y = F.conv2d(y, sd[f"{name}.{idx[3]}.weight"], sd[f"{name}.{idx[3]}.bias"])
    if layout["final"] == "tanh":
        y = torch.tanh(y)
    elif layout["final"] == "sine":
        y = torch.sin(y)
    return y


def forward(sd, images, cfg, use_bg_sdf=True, sdf_activation="tanh", inter=None):
    """models/objectness_net.py:167-183."""
    center, sdf = head_layout(use_bg_sdf, sdf_activation)
    feat = dpt_forward(sd, images, cfg, inter)
    return {
        "center_fields": head_forward(sd, "center_field_prediction_head", feat, center),
        "sdf_maps": head_forward(sd, "sdf_prediction_head", feat, sdf),
    }


def loss_terms(out, gt_center, gt_sdf, gt_sal, center_loss="l2", sdf_loss="l1",
               use_grad_loss=True, use_bce_loss=True):
    """train_objectness_net.py:215-254; returns (total, [terms])."""
    pc, ps = out["center_fields"], out["sdf_maps"]
    terms = []
    d = pc - gt_center
    terms.append(((d ** 2) if center_loss == "l2" else d.abs()).mean())
    d = ps - gt_sdf
    terms.append(((d ** 2) if sdf_loss == "l2" else d.abs()).mean())
    if use_grad_loss:
        dy, dx = image_gradients(gt_sdf)
        g_gt = torch.cat((dy, dx), dim=1)[:, :, 0:-1, 0:-1]
        dy, dx = image_gradients(ps)
        g_pr = torch.cat((dy, dx), dim=1)[:, :, 0:-1, 0:-1]
        d = g_gt - g_pr
        terms.append(((d ** 2) if sdf_loss == "l2" else d.abs()).mean())
    if use_bce_loss:
        terms.append(F.binary_cross_entropy(torch.sigmoid(ps), gt_sal, reduction="mean"))
    total = terms[0]
    for t in terms[1:]:
        total = total + t
    return total, terms


def adam_update(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (train_objectness_net.py:96: lr only; defaults otherwise),
    in place; `step` is the 1-based step count."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# synthetic labels (datasets.py:158-222 formulas)
# --------------------------------------------------------------------------
def _edt(mask):
    from scipy import ndimage
    return ndimage.distance_transform_edt(mask).astype(np.float32)


def synth_labels(masks):
    """masks: [B,H,W] {0,1} numpy -> center_field [B,2,H,W], sdf [B,H,W], saliency.

    datasets.py:158-159,200-213 (center = bbox centre, channel 0 = row offset,
    L2-normalised inside the mask, 0 outside) and :187-197 (fg DT/max - bg DT/max,
    --use_bg_sdf).  cv2.distanceTransform(DIST_L2, 3x3/5x5 mask) is approximated by
    the exact Euclidean DT here: these are synthetic *inputs*, not a parity claim."""
    B, H, W = masks.shape
    cf = np.zeros((B, 2, H, W), np.float32)
    sdf = np.zeros((B, H, W), np.float32)
    for b in range(B):
        m = masks[b] > 0
        ys, xs = np.nonzero(m)
        cy = (ys.min() + ys.max()) / 2.0
        cx = (xs.min() + xs.max()) / 2.0
        yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        oy, ox = yy - cy, xx - cx
        n = np.sqrt(oy ** 2 + ox ** 2) + 1e-12
        cf[b, 0] = np.where(m, oy / n, 0)
        cf[b, 1] = np.where(m, ox / n, 0)
        fg = _edt(m)
        bg = _edt(~m)
        sdf[b] = fg / max(fg.max(), 1e-6) - bg / max(bg.max(), 1e-6)
    return cf, sdf, masks.astype(np.float32)


# --------------------------------------------------------------------------
# peak picking (object_reasoning.py:360-377,528-550, utils/misc.py:10-20)
# --------------------------------------------------------------------------
def batch_erode(binary_masks, kernel_size=9, num_round=3):
    """utils/misc.py:10-20 (float64 box conv, keep where sum >= k*k)."""
    m = binary_masks.unsqueeze(1)
    kernel = torch.ones(1, 1, kernel_size, kernel_size, dtype=torch.float64)
    for _ in range(num_round):
        conved = F.conv2d(m.double(), kernel, padding=(kernel_size - 1) // 2)[:, 0]
        m = torch.where(conved >= kernel_size * kernel_size, 1, 0).unsqueeze(1)
    return m.squeeze(1)


def anti_center_map(vote_maps, kernel_size=5):
    """object_reasoning.py:360-377: float64 correlation with normalize((c-i, c-j))."""
    xv, yv = torch.meshgrid([torch.arange(kernel_size), torch.arange(kernel_size)], indexing="ij")
    grid = torch.stack((xv, yv), 2).view((1, kernel_size, kernel_size, 2)).float()
    c = int(kernel_size / 2)
    filt = -grid.permute(0, 3, 1, 2) + torch.tensor([c, c]).view(1, 2, 1, 1)
    filt = F.normalize(filt, dim=1).double()
    s = F.conv2d(vote_maps.double(), filt, padding=(kernel_size - 1) // 2)[:, 0]
    return s / (kernel_size ** 2 - 1)


def peak_pick(sdf_maps, center_fields, border=10):
    """object_reasoning.py:528-550: returns (score maps f64, amax, flat argmax)."""
    sdf_bin = torch.where(torch.sigmoid(sdf_maps) > 0.5, 1, 0)
    cen_bin = torch.where(torch.norm(center_fields, dim=1) > 0.5, 1, 0)
    union = torch.where((cen_bin + sdf_bin) > 0, 1, 0)
    eroded = batch_erode(union, 9, 3)
    score = anti_center_map(center_fields, 5) * eroded
    score[:, 0:border, :] = 0
    score[:, -border:, :] = 0
    score[:, :, 0:border] = 0
    score[:, :, -border:] = 0
    B = score.shape[0]
    flat = score.reshape(B, -1)
    return score, flat.amax(dim=1), flat.argmax(dim=1)


def peak_certificate(sdf_maps, center_fields, eps, thres=0.009, border=10, certify_empty=False):
    """Is the flat argmax of `peak_pick` provably unchanged by ANY perturbation of the fields below `eps` (max-norm)?
    (tests/golden/make_golden_r2.py computes the same certificate for the committed peak fixtures.)  Erosion is monotone, so
    with F = the pixels such a perturbation can move across a threshold of the union mask (object_reasoning.py:528-533), every
    reachable eroded mask lies between erode(union & ~F) and erode(union | F); a score moves by at most sqrt(2)*eps (24
    unit-vector taps / 24).  Certified when the peak survives in the smallest mask, beats every pixel of the largest mask by
    more than 2*sqrt(2)*eps and amax stays on its side of the singularity threshold (:541).
    certify_empty (round 6; the committed fixtures were made without it): a map WITHOUT a positive score -- amax == 0 at flat index 0, a
    border pixel whose score is exactly zero under any perturbation -- is certified too when no pixel of the largest mask can reach a
    positive score (all of them below -2*sqrt(2)*eps, or that mask empty inside the border): the checker of the device-side
    certificate unmore_amd/csrc/reasoning.hip::center_peaks_cert_kernel, which needs it because most sweep proposals have no peak.
    Returns (amax [B] f64, flat argmax [B] i64, certified [B] bool)."""
    sdf_bin = torch.where(torch.sigmoid(sdf_maps) > 0.5, 1, 0)
    cnorm = torch.norm(center_fields, dim=1)
    cen_bin = torch.where(cnorm > 0.5, 1, 0)
    union = torch.where((cen_bin + sdf_bin) > 0, 1, 0)
    score, amax, arg = peak_pick(sdf_maps, center_fields, border)
    d_s, d_c = sdf_maps.abs(), (cnorm - 0.5).abs()
    both = (sdf_bin == 1) & (cen_bin == 1)
    flip = torch.where(both, torch.maximum(d_s, d_c), torch.where(sdf_bin == 1, d_s, torch.where(cen_bin == 1, d_c, torch.minimum(d_s, d_c))))
    Fm = flip < eps
    er_min = batch_erode(torch.where(Fm, 0, union), 9, 3)
    er_max = batch_erode(torch.where(Fm, 1, union), 9, 3)
    fg_max = anti_center_map(center_fields, 5) * er_max
    fg_max[:, 0:border, :] = 0
    fg_max[:, -border:, :] = 0
    fg_max[:, :, 0:border] = 0
    fg_max[:, :, -border:] = 0
    B = sdf_maps.shape[0]
    bound = 2.0 * math.sqrt(2.0) * eps
    cert = torch.zeros(B, dtype=torch.bool)
    for b in range(B):
        p_ = int(arg[b])
        if float(amax[b]) <= 0:
            if certify_empty and float(amax[b]) == 0.0 and p_ == 0 and border >= 1 and abs(thres) > bound:
                inside = torch.zeros_like(er_max[b], dtype=torch.bool)
                inside[border:-border, border:-border] = True
                live = (er_max[b] == 1) & inside
                cert[b] = (not bool(live.any())) or bool(float(fg_max[b][live].max()) < -bound)
            continue
        if int(er_min[b].reshape(-1)[p_]) != 1:
            continue
        others = fg_max[b].reshape(-1).clone()
        others[p_] = -1e300
        cert[b] = bool(float(amax[b]) - float(others.max()) > bound) and abs(float(amax[b]) - thres) > bound
    return amax, arg, cert


def crop_resize(image, boxes, size=128):
    """object_reasoning.py:311-323: per box floor/ceil, crop, torchvision tensor Resize((size,size), BILINEAR)
    (torchvision 0.14: no antialias == F.interpolate(mode='bilinear', align_corners=False))."""
    outs = []
    for box in boxes:
        x1, y1, x2, y2 = [float(v) for v in box]
        x1, y1, x2, y2 = int(math.floor(x1)), int(math.floor(y1)), int(math.ceil(x2)), int(math.ceil(y2))
        crop = image[:, y1:y2, x1:x2]
        outs.append(F.interpolate(crop[None], size=(size, size), mode="bilinear", align_corners=False)[0])
    return torch.stack(outs, 0)


def update_bbox_with_boundary_fields(sdf_maps):
    """object_reasoning.py:139-174 (image_gradients restated as above)."""
    dy, dx = image_gradients(sdf_maps.unsqueeze(1))
    g = torch.cat((dy, dx), dim=1)[:, :, 0:-1, 0:-1]
    s = sdf_maps[:, 0:-1, 0:-1]
    gn = torch.norm(g, dim=1)
    fg = torch.sigmoid(s)
    bg = 1 - fg
    avg_fg = (fg * gn).sum(-1).sum(-1) / (fg.sum(-1).sum(-1) + 1e-8)
    avg_bg = (bg * gn).sum(-1).sum(-1) / (bg.sum(-1).sum(-1) + 1e-8)
    step_fg = 1 / (avg_fg + 1e-10)
    step_bg = 1 / (avg_bg + 1e-10)
    step = step_fg[:, None, None] * fg + step_bg[:, None, None] * bg
    mv = step * s
    return (-torch.amax(mv[:, :, 0], dim=1), -torch.amax(mv[:, 0, :], dim=1), torch.amax(mv[:, :, -1], dim=1),
            torch.amax(mv[:, -1, :], dim=1))


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms (object_reasoning.py:661; torchvision is absent from this image: restated from its documented algorithm and
    its CPU kernel's arithmetic -- UNPINNED boundary): boxes by descending score (stable: equal scores keep their input order), greedily
    keep a box unless a kept one overlaps it with IoU > threshold; IoU in float32 as inter / (area_a + area_b - inter).
    numpy in, kept indices (int64, rank order) out."""
    b = np.asarray(boxes, dtype=np.float32)
    order = np.argsort(-np.asarray(scores, dtype=np.float64), kind="stable")
    keep, removed = [], np.zeros(len(b), dtype=bool)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    for r, i in enumerate(order):
        if removed[r]:
            continue
        keep.append(int(i))
        for r2 in range(r + 1, len(order)):
            j = order[r2]
            w = np.float32(max(np.float32(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0])), np.float32(0)))
            h = np.float32(max(np.float32(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1])), np.float32(0)))
            inter = np.float32(w * h)
            with np.errstate(divide="ignore", invalid="ignore"):
                iou = np.float32(inter / np.float32(np.float32(area[i] + area[j]) - inter))
            if iou > np.float32(iou_threshold):
                removed[r2] = True
    return np.asarray(keep, dtype=np.int64)
