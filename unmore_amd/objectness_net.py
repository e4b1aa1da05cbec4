"""Drop-in `ObjectnessNet` (reference: models/objectness_net.py:37-203).

Same constructor, attributes, `forward(images)` / `get_prediction(images)` contract
and `state_dict()` schema (378 keys for 'dpt_large') as the
reference module, so `object_reasoning.py` / `object_scoring.py` /
`train_objectness_net.py` can construct it, `load_state_dict(strict=True)` a released
checkpoint and call it unchanged.  The sub-modules below only HOLD parameters under the
reference's names; all arithmetic runs in the hand-written HIP kernels via
`unmore_amd.engine.Engine`.  There is no CPU path: calling it with CPU tensors raises.
"""
import os

import torch
from torch import nn

from . import graphs, ops
from .engine import CONFIGS, Engine


_SDF_HEAD_MODE = os.environ.get("UMR_SDF_HEAD_MODE", "auto")    # default of set_sdf_head_mode (A/B switch for benchmarking)


def head_layout(use_bg_sdf, sdf_activation):
    """Conv indices / activations of the two heads (objectness_net.py:109-164)."""
    center = dict(conv_idx=[0, 2, 4, 6], relu=True, final=None)
    if use_bg_sdf:
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


class SinActivation(nn.Module):  # objectness_net.py:30-35 (placeholder in the Sequential; never called)
    def forward(self, x):
        return torch.sin(x)


class _Attention(nn.Module):
    def __init__(self, D, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(D, 3 * D, bias=True)
        self.proj = nn.Linear(D, D)


class _Mlp(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.fc1 = nn.Linear(D, 4 * D)
        self.fc2 = nn.Linear(4 * D, D)


class _Block(nn.Module):
    def __init__(self, D, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(D, eps=1e-6)
        self.attn = _Attention(D, heads)
        self.norm2 = nn.LayerNorm(D, eps=1e-6)
        self.mlp = _Mlp(D)


class _PatchEmbed(nn.Module):
    def __init__(self, D, p):
        super().__init__()
        self.proj = nn.Conv2d(3, D, kernel_size=p, stride=p)


class _ViT(nn.Module):
    """Parameter holder with timm VisionTransformer's names (vit.py:165-201 touches these)."""

    def __init__(self, D, depth, heads, patch, grid):
        super().__init__()
        self.patch_embed = _PatchEmbed(D, patch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.randn(1, 1 + grid * grid, D) * 0.02)
        self.blocks = nn.Sequential(*[_Block(D, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(D, eps=1e-6)  # computed-and-discarded in the reference (vit.py:107,199)
        self.head = nn.Linear(D, 1000)         # never used; kept for checkpoint key parity
        self.patch_size = [patch, patch]
        self.start_index = 1


class _ProjectReadout(nn.Module):  # vit.py:79-90
    def __init__(self, D):
        super().__init__()
        self.project = nn.Sequential(nn.Linear(2 * D, D), nn.GELU())


def _postprocess(D, F_, kind):  # vit.py:259-336
    mods = [_ProjectReadout(D), nn.Identity(), nn.Identity(), nn.Conv2d(D, F_, 1)]
    if kind == "up4":
        mods.append(nn.ConvTranspose2d(F_, F_, kernel_size=4, stride=4))
    elif kind == "up2":
        mods.append(nn.ConvTranspose2d(F_, F_, kernel_size=2, stride=2))
    elif kind == "down2":
        mods.append(nn.Conv2d(F_, F_, kernel_size=3, stride=2, padding=1))
    return nn.Sequential(*mods)


class _RCU(nn.Module):  # blocks.py:247-313
    def __init__(self, f):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, padding=1)
        self.conv2 = nn.Conv2d(f, f, 3, padding=1)


class _Fusion(nn.Module):  # blocks.py:318-383
    def __init__(self, f):
        super().__init__()
        self.out_conv = nn.Conv2d(f, f, 1)
        self.resConfUnit1 = _RCU(f)
        self.resConfUnit2 = _RCU(f)


class DPT(nn.Module):  # models/dpt/models.py:26-94
    def __init__(self, cfg, features=256):
        super().__init__()
        D, Fs = cfg["D"], cfg["features"]
        self.pretrained = nn.Module()
        self.pretrained.model = _ViT(D, cfg["depth"], cfg["heads"], cfg["patch"], cfg["pos_grid"])
        for k, kind in enumerate(("up4", "up2", "same", "down2")):
            setattr(self.pretrained, f"act_postprocess{k + 1}", _postprocess(D, Fs[k], kind))
        self.scratch = nn.Module()
        for k in range(4):
            setattr(self.scratch, f"layer{k + 1}_rn", nn.Conv2d(Fs[k], features, 3, padding=1, bias=False))
        for k in (1, 2, 3, 4):
            setattr(self.scratch, f"refinenet{k}", _Fusion(features))


def _head(feat_dim, cout, layout):
    convs = [nn.Conv2d(feat_dim, 512, 1), nn.Conv2d(512, 512, 3, padding=1), nn.Conv2d(512, 1024, 1), nn.Conv2d(1024, cout, 1)]
    mods = []
    for c in convs[:-1]:
        mods.append(c)
        if layout["relu"]:
            mods.append(nn.ReLU())
    mods.append(convs[-1])
    if layout["final"] == "tanh":
        mods.append(nn.Tanh())
    elif layout["final"] == "sine":
        mods.append(SinActivation())
    return nn.Sequential(*mods)


class _NetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward / backward are the engine's
    hand-scheduled kernel sequences."""

    @staticmethod
    def forward(ctx, images, net, names, *params):
        P = dict(zip(names, params))
        center, sdf, S = net._engine().forward(P, images, save=True)
        ctx.net, ctx.names, ctx.S, ctx.P = net, names, S, P
        return center, sdf

    @staticmethod
    def backward(ctx, d_center, d_sdf):
        net, names, S, P = ctx.net, ctx.names, ctx.S, ctx.P
        if S is None or not S:
            raise RuntimeError("ObjectnessNet backward called twice (activations already released)")
        B, H, W = S["B"], S["H"], S["W"]
        dev = P[names[0]].device
        if d_center is None:
            d_center = torch.zeros((B, 2, H, W), dtype=torch.float32, device=dev)
        if d_sdf is None:
            d_sdf = torch.zeros((B, 1, H, W), dtype=torch.float32, device=dev)
        skip = net.nograd_names()
        G = {n: torch.empty_like(P[n]) for n in names if n not in skip}
        net._engine().backward(P, S, d_center.float().contiguous(), d_sdf.float().contiguous(), G)
        ctx.S = None
        return (None, None, None) + tuple(G.get(n) for n in names)


class ObjectnessNet(nn.Module):
    def __init__(self, device, image_size, backbone_type, args=None, compute_dtype=None):
        super().__init__()
        self.image_size = image_size
        self.device = device
        self.backbone_type = backbone_type
        self.args = args
        if backbone_type not in CONFIGS:
            # 'resnet50' / 'dpt_hybrid' cannot execute in the reference either (undefined names,
            # objectness_net.py:51-61,74-105); unknown names raise as the reference does (:106-107)
            raise NotImplementedError
        self.cfg = CONFIGS[backbone_type]
        self.backbone = DPT(self.cfg, features=256)
        feat_dim = 256
        self._layouts = head_layout(self.args.use_bg_sdf, self.args.sdf_activation)
        self.center_field_prediction_head = _head(feat_dim, 2, self._layouts[0])
        self.sdf_prediction_head = _head(feat_dim, 1, self._layouts[1])
        dt = compute_dtype if compute_dtype is not None else getattr(args, "compute_dtype", None)
        self.compute_dtype = dt if dt is not None else torch.float32
        self._eng = None

    # ---- engine plumbing
    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        self._eng = None
        return self

    def set_sdf_head_mode(self, mode):
        """How a head WITHOUT non-linearities between its convs (the boundary-distance head's tanh / sine / None variants,
        objectness_net.py:119-142,128-135) is evaluated.  Its four convolutions compose to ONE 3x3 conv 256 -> 1 plus a
        border-dependent bias (csrc/linear_head.hip).
        'auto' (default): inference calls (torch.no_grad() / no parameter requires grad -- object_reasoning.py:326,351,413) use
        that collapsed form (round 5; qualified against every reference-made forward fixture at the 1e-4 contract and through the
        peak chain, tests/test_collapsed_head_gpu.py), and so do training steps whose backward of that head is the algebraic one
        (the default, set_linear_head_backward): that backward reads the head's output only, so the four convolutions' 512 / 512 /
        1024-channel maps would be computed and thrown away (round 6; loss and every gradient against the float64 oracle at the
        suite's bars, tests/test_collapsed_train_gpu.py).  'sine' in training and every ReLU variant run the four convolutions.
        The weights stay factored: same state_dict schema, same eight gradient tensors.
        'factored': always the four convolutions as the reference runs them (the A/B switch; bench.py's alt_factored_sdf_head).
        'collapsed': the collapsed form wherever the head allows it, whatever the backward mode."""
        assert mode in ("auto", "factored", "collapsed")
        self.sdf_head_mode = mode
        self._eng = None
        return self

    def set_linear_head_backward(self, mode):
        """How the gradients of a head WITHOUT non-linearities between its convs (tanh / None variants of the boundary-distance
        head, objectness_net.py:119-142) are computed; its forward always runs the four convolutions as the reference does.
        'algebraic' (default): exact gradients of all eight factored tensors from three pixel reductions over the shared
        feature map -- the 512/1024-channel activations are neither stored nor multiplied in backward.  'gemm': the
        layer-by-layer data- and weight-gradient GEMMs."""
        assert mode in ("algebraic", "gemm")
        self.linear_head_bwd_mode = mode
        self._eng = None
        return self

    def _engine(self):
        if self._eng is None or self._eng.dt != self.compute_dtype:
            self._eng = Engine(self.cfg, self._layouts, self.compute_dtype,
                               collapse_linear_heads={"auto": "auto", "factored": False, "collapsed": True}[
                                   getattr(self, "sdf_head_mode", None) or _SDF_HEAD_MODE],
                               linear_head_backward=getattr(self, "linear_head_bwd_mode", None))
        return self._eng

    def nograd_names(self):
        """Parameters the reference never back-propagates into."""
        m = "backbone.pretrained.model."
        out = {m + "norm.weight", m + "norm.bias", m + "head.weight", m + "head.bias"}
        for u in ("conv1", "conv2"):
            for t in ("weight", "bias"):
                out.add(f"backbone.scratch.refinenet4.resConfUnit1.{u}.{t}")
        last = max(self.cfg["hooks"])
        for n, _ in self.named_parameters():
            if n.startswith(m + "blocks."):
                if int(n[len(m + "blocks."):].split(".")[0]) > last:
                    out.add(n)
        return out

    _HEAD_OF = {"center_fields": "center_field_prediction_head", "sdf_maps": "sdf_prediction_head"}

    def _run(self, images, heads=None):
        if self.backbone_type not in CONFIGS:
            raise NotImplementedError
        if heads is not None:
            heads = tuple(heads)
            if not heads or any(h not in self._HEAD_OF for h in heads):
                raise ValueError(f"heads: a non-empty subset of {tuple(self._HEAD_OF)}, got {heads}")
        skip = () if heads is None else tuple(sorted(m for k, m in self._HEAD_OF.items() if k not in heads))
        if not images.is_cuda:
            raise RuntimeError("unmore_amd.ObjectnessNet runs on the MI355X only (no CPU fallback); move the model and inputs to the GPU")
        in_dtype = images.dtype
        if images.shape[0] == 0:
            # an empty batch (the train loop's filter can drop every image, train_objectness_net.py:190-207): the reference's convs
            # return empty maps; so do we, without a launch
            B0, _, H0, W0 = images.shape
            empty = {"center_fields": images.new_zeros((0, 2, H0, W0)), "sdf_maps": images.new_zeros((0, 1, H0, W0))}
            return {k: v for k, v in empty.items() if heads is None or k in heads}
        x = images.float()
        named = list(self.named_parameters())
        names = tuple(n for n, _ in named)
        params = tuple(p for _, p in named)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            if skip:
                raise RuntimeError("get_prediction(heads=...): inference only (call under torch.no_grad() / with frozen parameters)")
            center, sdf = _NetFunction.apply(x, self, names, *params)
        else:
            center, sdf = self._forward_nograd(x, names, params, skip)
        out_dict = {}
        if center is not None:
            out_dict["center_fields"] = center.to(in_dtype)  ## [B, 2, H, W]
        if sdf is not None:
            out_dict["sdf_maps"] = sdf.to(in_dtype)  ## [B, 1, H, W]
        return out_dict

    def set_graph_mode(self, mode):
        """Inference calls (no autograd) of small batches -- 'auto': B*H*W <= 2^20 pixels, the reference's [<= 50, 3, 128, 128] crops
        (object_reasoning.py:301-337) -- are captured into a HIP graph on their third call with the same shape and replayed
        afterwards (graphs.py); 'on' / 'off' force it.  A change of any parameter drops the captures.
        Stream contract: a capture belongs to the stream it was recorded for (the key holds the stream) and a replay is ordered like
        any launch on that stream.  The weight copies it reads were packed or refreshed by work enqueued BEFORE the capture was made
        (the two eager warm-up calls, on this stream) or are refreshed in place by a train step whose caller has joined its streams
        (TrainStep does before it returns); a caller that updates parameters on ANOTHER stream must order that stream before the
        next call itself, exactly as for an eager call.  Every net keeps at most graphs.MAX_CAPTURES captures (each with its own
        pool of temporaries); beyond that all are dropped and re-captured as their shapes recur."""
        assert mode in ("auto", "on", "off")
        self.graph_mode = mode
        self._inf_graphs = {}
        return self

    def _forward_nograd(self, x, names, params, skip=()):
        eng = self._engine()

        def both(outs):       # the captured call returns the evaluated heads only; put them back in (center, sdf) order
            it = iter(outs)
            return tuple(None if m in skip else next(it) for m in ("center_field_prediction_head", "sdf_prediction_head"))

        P = dict(zip(names, params))
        B, _, H, W = x.shape
        mode = getattr(self, "graph_mode", None)
        if graphs.wanted(mode, B * H * W) and ops._timer["select"] is None and not graphs.capturing():
            # the captures read the packed weight copies and, in fp32 mode, the parameters themselves: any in-place update bumps a
            # version counter, a re-homed parameter changes its address
            sig = (sum(p._version for p in params), hash(tuple(p.data_ptr() for p in params)), id(eng))
            store = self.__dict__.setdefault("_inf_graphs", {})
            if store.get("sig") != sig:
                store.clear()
                store["sig"] = sig
            key = (tuple(x.shape), ops.get_f32_mode(), torch.cuda.current_stream(x.device).cuda_stream, skip)
            ent = store.get(key)
            if isinstance(ent, graphs.Captured):
                if ent.valid():
                    return both(t.clone() for t in ent.replay(x))
                if ent.failed is None:
                    ent = None
            if not isinstance(ent, graphs.Captured):
                n = (ent or 0) + 1
                store[key] = n
                if n > graphs.WARMUP_CALLS:
                    # every capture keeps its own pool of temporaries alive: a caller that cycles through many shapes gets at most
                    # MAX_CAPTURES of them (then all are dropped and the current shapes are captured again as they recur)
                    if sum(isinstance(v, graphs.Captured) for v in store.values()) >= graphs.MAX_CAPTURES:
                        for k in [k for k, v in store.items() if isinstance(v, graphs.Captured)]:
                            del store[k]
                        graphs.release_dropped()
                    cap = graphs.Captured(lambda xs: tuple(t for t in eng.forward(P, xs, save=False, skip=skip)[:2] if t is not None), (x,),
                                          generation_of=eng.cache.generation, on_fail=eng.cache.purge_capture)
                    store[key] = cap
                    if cap.failed is None:
                        return both(t.clone() for t in cap.replay(x))
        center, sdf, _ = eng.forward(P, x, save=False, skip=skip)
        return center, sdf

    def forward(self, images):
        return self._run(images)

    def get_prediction(self, images, heads=None):
        """objectness_net.py:188-203.  heads (extension, inference only): a subset of ('center_fields', 'sdf_maps') -- only those heads are
        evaluated and only those keys returned (object_reasoning.py:379-487 reads 'sdf_maps' alone, fifty rounds per image)."""
        return self._run(images, heads)
