"""Round-2 golden fixtures (SURVEY.md section 8c classes (ii), (iii), (iv)), made by running the REFERENCE's own code on CPU.

Run in the build container only (needs /root/reference; never shipped):
    python tests/golden/make_golden_r2.py [loss] [peaks] [full]

What executes verbatim from the reference:
  (ii)  loss   -- the source LINES train_objectness_net.py:215-254 (the 4-term loss; it has no callable entry point: the block
                  sits inside the 140-line train loop), read from the reference file at run time, dedented and exec'd in a
                  namespace that supplies `self.args`, `self.device`, `out_dict`, the three label tensors, `torch`, `nn` and
                  a `torchmetrics` placeholder; gradients w.r.t. the predictions by autograd.
  (iv)  peaks  -- utils/misc.py:10-20 `batch_erode`; object_reasoning.py:360-377 `center_field_to_anti_center_map`, :139-174
                  `update_bbox_with_boundary_fields`, :198-204 `unravel_index` and the whole of :525-580 `center_reasoning`
                  (threshold / union / erode / score / border / amax / argmax / split boxes), called as unbound functions of
                  the imported `Object_Discovery` class with a stub `self`; the maps fed to them come from the reference
                  modules' own forward (ViT-B/16 wiring `DPT("vitb16_384")` and the miniature config, as make_golden.py).
  (iii) full   -- ObjectnessNet forward at the benchmark's size (ViT-B/16 wiring, 384x384, B=1) through the reference modules.
What cannot: timm (TimmContractViT of make_golden.py), torchmetrics.functional.image_gradients (absent from this image:
restated below as documented -- forward differences, last row / column zero), and import-only modules of
object_reasoning.py (cv2, seaborn, skimage, pycocotools, torchvision: inert placeholders, never called).

Only DATA is written: inputs that cannot be regenerated bit-exactly from unmore_amd.hashrng / unmore_amd.synth, and the
reference's outputs.
"""
import os
import sys
import textwrap
import types
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"
sys.path.insert(0, REF)

import make_golden as mg  # noqa: E402  (TimmContractViT, hash weights)
from unmore_amd import synth  # noqa: E402
from unmore_amd.hashrng import uniform, uniform01  # noqa: E402

# documented weight edits that make a randomly initialised net produce masks that survive the 3 x (9x9) erosion and centre
# fields whose norm crosses 0.5 (with plain hash weights every score map is identically zero -- a vacuous argmax):
#   shift: added to the last bias of sdf_prediction_head (pre-tanh); scale: multiplies weight and bias of the last layer of
#   center_field_prediction_head.  Chosen per weight set so that the eight blob images give eroded masks of varied size.
EDITS = {"base": dict(shift=0.5, scale=1.5), "tiny": dict(shift=0.05, scale=2.0)}
CERT_EPS = 2e-4            # field perturbation the argmax certificate below is computed for (2x the 1e-4 parity bar)


# --------------------------------------------------------------------------- placeholders
class _Inert(types.ModuleType):
    """import-only placeholder: any attribute is another inert object; calling it is an error we want to see."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Inert(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        raise RuntimeError(f"placeholder {self.__name__} was called: the fixture would not be the reference's result")


def image_gradients(img):
    """torchmetrics==1.5.2 functional.image_gradients (requirements.txt:15), restated from its documentation: dy[i,j] =
    x[i+1,j]-x[i,j], dx[i,j] = x[i,j+1]-x[i,j], last row (dy) / last column (dx) zero; returns (dy, dx).  UNPINNED boundary."""
    dy = torch.zeros_like(img)
    dx = torch.zeros_like(img)
    dy[..., :-1, :] = img[..., 1:, :] - img[..., :-1, :]
    dx[..., :, :-1] = img[..., :, 1:] - img[..., :, :-1]
    return dy, dx


def install_placeholders():
    mg._install_placeholders()  # timm (contract ViT), torchvision, torchvision.transforms
    for name in ("cv2", "seaborn", "skimage", "skimage.morphology", "skimage.draw", "pycocotools", "pycocotools.mask",
                 "pycocotools.coco", "torchvision.utils"):
        if name not in sys.modules:
            sys.modules[name] = _Inert(name)
    for name in ("torchvision", "torchvision.transforms"):
        m = _Inert(name)
        sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    tm = types.ModuleType("torchmetrics")
    tmf = types.ModuleType("torchmetrics.functional")
    tmf.image_gradients = image_gradients
    tm.functional = tmf
    sys.modules["torchmetrics"] = tm
    sys.modules["torchmetrics.functional"] = tmf
    return tm


# --------------------------------------------------------------------------- (ii) loss
def reference_loss_block():
    with open(os.path.join(REF, "train_objectness_net.py")) as f:
        lines = f.readlines()
    src = textwrap.dedent("".join(lines[214:254]))  # 1-based lines 215..254
    assert src.lstrip().startswith("loss = torch.tensor(0.0)") and "bce_loss(pred_sdf_binary_mask, gt_saliency_maps)" in src
    return compile(src, "train_objectness_net.py:215-254", "exec")


def make_loss(tm):
    code = reference_loss_block()
    B, H, W = 2, 12, 10
    pc = torch.from_numpy(uniform("loss:pc", (B, 2, H, W), -1.2, 1.2))
    ps = torch.from_numpy(uniform("loss:ps", (B, 1, H, W), -0.98, 0.98))   # tanh range
    gc = torch.from_numpy(uniform("loss:gc", (B, 2, H, W), -1.0, 1.0))
    gs = torch.from_numpy(uniform("loss:gs", (B, 1, H, W), -1.0, 1.0))
    sal = torch.from_numpy((uniform01("loss:sal", (B, 1, H, W)) > 0.5).astype(np.float32))
    save = {}
    for cl in ("l2", "l1"):
        for sl in ("l1", "l2"):
            for ug in (0, 1):
                for ub in (0, 1):
                    p_c = pc.clone().requires_grad_(True)
                    p_s = ps.clone().requires_grad_(True)
                    ns = dict(self=Namespace(args=Namespace(center_field_loss_type=cl, sdf_loss_type=sl, use_sdf_gradient_loss=bool(ug),
                                                            use_sdf_binary_mask_loss=bool(ub)), device="cpu"),
                              out_dict={"center_fields": p_c, "sdf_maps": p_s}, gt_center_fields=gc, gt_sdf_maps=gs,
                              gt_saliency_maps=sal, torch=torch, nn=nn, torchmetrics=tm)
                    exec(code, ns)
                    loss = ns["loss"]
                    loss.backward()
                    key = f"{cl}_{sl}_g{ug}_b{ub}"
                    save[key + "_loss"] = np.float32(loss.item())
                    save[key + "_dpc"] = p_c.grad.numpy()
                    save[key + "_dps"] = p_s.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "loss_terms.npz"), **save)
    print("loss fixtures:", len(save) // 3, "flag combinations; e.g. l2_l1_g1_b1 =", save["l2_l1_g1_b1_loss"])


# --------------------------------------------------------------------------- reference nets
def build_reference_nets():
    from models.objectness_net import ObjectnessNet
    from models.dpt.models import DPT, _make_fusion_block
    from models.dpt.blocks import _make_scratch, Interpolate
    from models.dpt.vit import _make_vit_b16_backbone
    args = Namespace(use_bg_sdf=True, sdf_activation="tanh")
    net = ObjectnessNet(device="cpu", image_size=128, backbone_type="dpt_large", args=args).eval()
    heads = (net.center_field_prediction_head, net.sdf_prediction_head)

    def shell(backbone):
        n = ObjectnessNet.__new__(ObjectnessNet)
        nn.Module.__init__(n)
        n.image_size, n.device, n.backbone_type, n.args = 128, "cpu", "dpt_large", args
        n.backbone = backbone
        n.center_field_prediction_head, n.sdf_prediction_head = heads
        return n.eval()

    def base():
        return shell(DPT(head=None, features=256, backbone="vitb16_384", readout="project", channels_last=False, use_bn=False,
                         enable_attention_hooks=False))

    def tiny():
        D, depth, nh, Fs, hooks = 128, 4, 2, [32, 64, 128, 128], [0, 1, 2, 3]
        vit = mg.TimmContractViT(D, depth, nh)
        pre = _make_vit_b16_backbone(vit, features=Fs, size=[384, 384], hooks=hooks, vit_features=D, use_readout="project")
        scratch = _make_scratch(Fs, 256, groups=1, expand=False)
        for k in (1, 2, 3, 4):
            setattr(scratch, f"refinenet{k}", _make_fusion_block(256, False))
        scratch.output_conv = nn.Sequential(Interpolate(scale_factor=2, mode="bilinear", align_corners=True))
        dpt = DPT.__new__(DPT)
        nn.Module.__init__(dpt)
        dpt.channels_last = False
        dpt.pretrained, dpt.scratch = pre, scratch
        return shell(dpt)

    def small():
        # BASELINE configs[0]: ViT-S/16 (an extension of the reference's backbone switch) assembled by the
        # reference's OWN builders with the DPT-small convention: D 384, 12 blocks, 6 heads, hooks [2,5,8,11], features [48,96,192,384]
        D, depth, nh, Fs, hooks = 384, 12, 6, [48, 96, 192, 384], [2, 5, 8, 11]
        vit = mg.TimmContractViT(D, depth, nh)
        pre = _make_vit_b16_backbone(vit, features=Fs, size=[384, 384], hooks=hooks, vit_features=D, use_readout="project")
        scratch = _make_scratch(Fs, 256, groups=1, expand=False)
        for k in (1, 2, 3, 4):
            setattr(scratch, f"refinenet{k}", _make_fusion_block(256, False))
        scratch.output_conv = nn.Sequential(Interpolate(scale_factor=2, mode="bilinear", align_corners=True))
        dpt = DPT.__new__(DPT)
        nn.Module.__init__(dpt)
        dpt.channels_last = False
        dpt.pretrained, dpt.scratch = pre, scratch
        return shell(dpt)

    return base, tiny, small


def edit_for_peaks(net, wtag):
    """the documented weight edits (EDITS above); mirrored by tests/peaks_common.py on the product side"""
    e = EDITS[wtag]
    with torch.no_grad():
        net.sdf_prediction_head[3].bias += e["shift"]
        net.center_field_prediction_head[6].weight *= e["scale"]
        net.center_field_prediction_head[6].bias *= e["scale"]


# --------------------------------------------------------------------------- (iv) peaks
def reference_peak_chain(OD, batch_erode, sdf_maps, center_fields, thres=0.009):
    """Runs the reference functions on maps ([B,H,W] f32, [B,2,H,W] f32); returns everything the tests compare."""
    B, H, W = sdf_maps.shape
    # the chain, step by step, with the reference's own functions (object_reasoning.py:528-539)
    sdf_bin = torch.where(torch.sigmoid(sdf_maps) > 0.5, 1, 0)
    cen_bin = torch.where(torch.norm(center_fields, dim=1) > 0.5, 1, 0)
    union = torch.where((cen_bin + sdf_bin) > 0, 1, 0)
    eroded = batch_erode(union, kernel_size=9, num_round=3)
    score = OD.center_field_to_anti_center_map(None, center_fields, kernel_size=5)
    fg = score * eroded
    fg[:, 0:10, :] = 0
    fg[:, -10:, :] = 0
    fg[:, :, 0:10] = 0
    fg[:, :, -10:] = 0
    amax = torch.amax(fg, dim=(1, 2))
    flat = fg.reshape(B, -1)
    arg = flat.argmax(dim=1)
    top2 = flat.topk(2, dim=1).values
    # ... and the method itself, verbatim, with a stub self: its split boxes encode (y, x) of the argmax for every map whose
    # maximum exceeds the threshold (proposal box = the whole 0..W x 0..H crop, so x_center = left box's x2, y_center = top box's y2)
    stub = types.SimpleNamespace(args=Namespace(center_score_max_thres=thres, analyze_cc=False))
    stub.get_prediction_with_proposals = lambda proposals, image: (sdf_maps, center_fields)
    stub.center_field_to_anti_center_map = lambda vm, kernel_size=5: OD.center_field_to_anti_center_map(stub, vm, kernel_size)
    stub.unravel_index = OD.unravel_index
    props = torch.tensor([[0.0, 0.0, float(W), float(H)]] * B)
    out = OD.center_reasoning(stub, None, props)
    fail = (amax > thres).nonzero().flatten().tolist()
    yx = np.full((B, 2), -1, np.int64)
    sp = out["splited_new_proposals"]
    assert len(sp) == 4 * len(fail)
    for j, b in enumerate(fail):
        left, top = sp[4 * j], sp[4 * j + 2]
        yx[b] = (int(round(float(top[3]))), int(round(float(left[2]))))
        assert yx[b, 0] * W + yx[b, 1] == int(arg[b]), "center_reasoning and the step-by-step chain disagree"
    # margin of the union mask to a threshold flip (smallest field change that alters any mask pixel)
    d_s = sdf_maps.abs()
    d_c = (torch.norm(center_fields, dim=1) - 0.5).abs()
    both = (sdf_bin == 1) & (cen_bin == 1)
    flip = torch.where(both, torch.maximum(d_s, d_c), torch.where(sdf_bin == 1, d_s, torch.where(cen_bin == 1, d_c, torch.minimum(d_s, d_c))))
    flip = flip.contiguous()
    # certificate: is the argmax provably unchanged by ANY perturbation of the fields below CERT_EPS?  Erosion is monotone,
    # so with F = pixels that such a perturbation can flip, every reachable eroded mask lies between erode(union & ~F) and
    # erode(union | F); a score changes by at most sqrt(2)*eps (24 unit-vector taps / 24).  Certified when the peak survives
    # in the smallest mask, beats every pixel of the largest mask by more than 2*sqrt(2)*eps, and amax stays on its side of
    # the singularity threshold.
    F_ = flip < CERT_EPS
    er_min = batch_erode(torch.where(F_, 0, union), kernel_size=9, num_round=3)
    er_max = batch_erode(torch.where(F_, 1, union), kernel_size=9, num_round=3)
    fg_max = score * er_max
    fg_max[:, 0:10, :] = 0
    fg_max[:, -10:, :] = 0
    fg_max[:, :, 0:10] = 0
    fg_max[:, :, -10:] = 0
    cert = np.zeros(B, bool)
    bound = 2.0 * np.sqrt(2.0) * CERT_EPS
    for b in range(B):
        p_ = int(arg[b])
        if float(amax[b]) <= 0 or int(er_min[b].reshape(-1)[p_]) != 1:
            continue
        others = fg_max[b].reshape(-1).clone()
        others[p_] = -1e300
        cert[b] = bool(float(amax[b]) - float(others.max()) > bound) and abs(float(amax[b]) - thres) > bound
    dx1, dy1, dx2, dy2 = OD.update_bbox_with_boundary_fields(sdf_maps)
    return dict(argmax_certified=cert, flippable_pixels=F_.reshape(B, -1).sum(1).numpy().astype(np.int64),
                eroded_bits=np.packbits(eroded.numpy().astype(np.uint8).reshape(B, -1), axis=1),
                eroded_count=eroded.reshape(B, -1).sum(1).numpy().astype(np.int64),
                score_support=(fg != 0).reshape(B, -1).sum(1).numpy().astype(np.int64),
                amax=amax.numpy(), argmax=arg.numpy().astype(np.int64), top2_margin=(top2[:, 0] - top2[:, 1]).numpy(),
                peak_yx=yx, mask_flip_margin=flip.reshape(B, -1).amin(1).numpy(),
                score_at_rows=fg[:, H // 2, :].numpy(),  # one full score row per map (float64), for value-level comparison
                deltas=torch.stack([dx1, dy1, dx2, dy2], 1).numpy())


def make_peaks():
    import object_reasoning as orz
    from utils.misc import batch_erode
    OD = orz.Object_Discovery
    base, tiny, _ = build_reference_nets()
    save = {}
    # (a) kernel-level: synthetic object-like fields (regenerated bit-exactly by unmore_amd.synth) through the reference functions
    for tag, (B, H, W, seed) in {"syn128": (6, 128, 128, 0), "syn96x160": (4, 96, 160, 1)}.items():
        sdf, cen = synth.object_like_fields(B, H, W, seed)
        r = reference_peak_chain(OD, batch_erode, torch.from_numpy(sdf), torch.from_numpy(cen))
        for k, v in r.items():
            save[f"{tag}_{k}"] = v
        print(tag, "amax", np.round(r["amax"], 4), "eroded", r["eroded_count"], "margin", r["top2_margin"])
    # (b) end-to-end: reference net forward (hash weights + documented edits) on blob images -> the same chain
    for tag, mk, wtag, B in (("e2e_base128", base, "base", 8), ("e2e_tiny128", tiny, "tiny", 8)):
        net = mk()
        mg._load_hash_weights(net, wtag)
        edit_for_peaks(net, wtag)
        x = torch.from_numpy(synth.blob_images(B, 128, 128, seed=7))
        with torch.no_grad():
            out = net(images=x)
        sdf, cen = out["sdf_maps"].squeeze(1).contiguous(), out["center_fields"].contiguous()
        r = reference_peak_chain(OD, batch_erode, sdf, cen)
        for k, v in r.items():
            save[f"{tag}_{k}"] = v
        # the reference net's maps themselves for the first 3 images (exact f32): kernel-level inputs that are real net outputs
        save[f"{tag}_sdf_maps"] = sdf[:3].numpy()
        save[f"{tag}_center_fields"] = cen[:3].numpy()
        # sampled field values of every image (forward parity at the fixture's own inputs)
        idx = (uniform01(f"peaks:{tag}:idx", (512,)) * (128 * 128)).astype(np.int64)
        save[f"{tag}_sample_idx"] = idx
        save[f"{tag}_sdf_samples"] = sdf.reshape(B, -1)[:, idx].numpy()
        save[f"{tag}_center_samples"] = cen.reshape(B, 2, -1)[:, :, idx].numpy()
        print(tag, "amax", np.round(r["amax"], 4), "eroded", r["eroded_count"], "margin", r["top2_margin"], "flip", r["mask_flip_margin"],
              "flippable", r["flippable_pixels"], "certified", r["argmax_certified"])
        assert (r["amax"] > 0).sum() >= B // 2 and r["argmax_certified"].sum() >= 3, "fixture would be close to vacuous"
        save[f"{tag}_meta_shift_scale"] = np.float32([EDITS[wtag]["shift"], EDITS[wtag]["scale"]])
    save["meta_cert_eps"] = np.float64(CERT_EPS)
    np.savez_compressed(os.path.join(HERE, "peaks.npz"), **save)


# --------------------------------------------------------------------------- (iii) full size
def make_full():
    base, _, small = build_reference_nets()
    net = base()
    mg._load_hash_weights(net, "base")
    H = W = 384
    x = torch.from_numpy(synth.blob_images(1, H, W, seed=11))
    inter = {}
    hooks = [net.backbone.scratch.refinenet1.register_forward_hook(lambda m, i, o: inter.__setitem__("path_1", o.detach())),
             net.backbone.scratch.output_conv.register_forward_hook(lambda m, i, o: inter.__setitem__("feat", o.detach()))]
    with torch.no_grad():
        out = net(images=x)
    for h in hooks:
        h.remove()
    idx = (uniform01("full384:idx", (4096,)) * (H * W)).astype(np.int64)
    cen, sdf = out["center_fields"][0], out["sdf_maps"][0]
    feat = inter["feat"][0]          # [256, 384, 384]
    save = dict(sample_idx=idx, center_samples=cen.reshape(2, -1)[:, idx].numpy(), sdf_samples=sdf.reshape(1, -1)[:, idx].numpy(),
                center_mean=cen.mean(dim=(1, 2)).numpy(), center_absmax=cen.abs().amax(dim=(1, 2)).numpy(),
                sdf_mean=sdf.mean(dim=(1, 2)).numpy(), sdf_absmax=sdf.abs().amax(dim=(1, 2)).numpy(),
                feat_samples=feat.reshape(256, -1)[:, idx[:256]].numpy(), feat_mean=feat.mean(dim=(1, 2)).numpy(),
                feat_absmax=feat.abs().amax(dim=(1, 2)).numpy())
    np.savez_compressed(os.path.join(HERE, "fwd_dpt_base_384_sampled.npz"), **save)
    print("full 384:", {k: (v.shape, float(np.abs(v).max())) for k, v in save.items()})
    # cfg1 shape: ViT-S/16, 224x224, batch 2
    net = small()
    mg._load_hash_weights(net, "dpt_small")
    x = torch.from_numpy(synth.blob_images(2, 224, 224, seed=12))
    with torch.no_grad():
        out = net(images=x)
    idx = (uniform01("small224:idx", (2048,)) * (224 * 224)).astype(np.int64)
    cen, sdf = out["center_fields"], out["sdf_maps"]
    save = dict(sample_idx=idx, center_samples=cen.reshape(2, 2, -1)[:, :, idx].numpy(), sdf_samples=sdf.reshape(2, 1, -1)[:, :, idx].numpy(),
                center_mean=cen.mean(dim=(0, 2, 3)).numpy(), sdf_mean=sdf.mean(dim=(0, 2, 3)).numpy(),
                center_absmax=cen.abs().amax(dim=(0, 2, 3)).numpy(), sdf_absmax=sdf.abs().amax(dim=(0, 2, 3)).numpy())
    np.savez_compressed(os.path.join(HERE, "fwd_dpt_small_224_sampled.npz"), **save)
    with open(os.path.join(HERE, "schema_dpt_small.txt"), "w") as f:
        for k, v in net.state_dict().items():
            f.write(f"{k} {' '.join(map(str, v.shape))}\n")
    print("small 224:", {k: (v.shape, float(np.abs(v).max())) for k, v in save.items()})


def main():
    what = set(sys.argv[1:]) or {"loss", "peaks", "full"}
    tm = install_placeholders()
    torch.manual_seed(0)
    if "loss" in what:
        make_loss(tm)
    torch.set_grad_enabled(False)
    if "peaks" in what:
        make_peaks()
    if "full" in what:
        make_full()


if __name__ == "__main__":
    main()
