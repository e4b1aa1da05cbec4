"""Drop-in `Binary_Classifier` (reference: models/objectness_net.py:205-223) -- SURVEY.md section 8f row f3.

torchvision ResNet-50 (v1.5: the stride sits on each bottleneck's 3x3 conv) + Linear(1000, 1) + sigmoid, as the
reference builds it, with the same constructor, the same 322-key `state_dict()` schema
(`classifier_backbone.*` in torchvision's names + `binary_classification_head.*`) and the same `forward(images) -> [B, 1]`
probabilities, so `object_reasoning.py:64-90,491-523` / `object_scoring.py:65-90,123-140` can construct it,
`load_state_dict(strict=True)` a released checkpoint and call it unchanged.

Only the path those callers use is implemented: eval mode (BatchNorm with running statistics, folded into the convs), no
gradients.  The sub-modules only HOLD parameters and buffers; the arithmetic runs on the HIP kernels (umr_gemm_nt for all
53 convolutions and both Linear layers, csrc/classifier.hip for the stem im2col / max-pool / BN folding).  There is no CPU
path and no training path: both raise.
"""
import torch
from torch import nn

from . import _lib as L
from . import ops

_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # (planes, blocks, stride of the first block)
_BN_EPS = 1e-5


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.stride = stride
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None


class _ResNet50(nn.Module):
    """Parameter / buffer holder with torchvision.models.resnet50's attribute names."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (planes, blocks, stride) in enumerate(_LAYERS):
            mods = []
            for bi in range(blocks):
                mods.append(_Bottleneck(inplanes, planes, stride if bi == 0 else 1, downsample=(bi == 0)))
                inplanes = planes * 4
            setattr(self, f"layer{li + 1}", nn.Sequential(*mods))
        self.fc = nn.Linear(2048, 1000)


class Binary_Classifier(nn.Module):
    def __init__(self, device, image_size, args=None, compute_dtype=None):
        super().__init__()
        self.image_size = image_size
        self.device = device
        self.args = args
        self.classifier_backbone = _ResNet50()
        self.binary_classification_head = torch.nn.Linear(1000, 1)
        self.sigmoid = torch.nn.Sigmoid()
        dt = compute_dtype if compute_dtype is not None else getattr(args, "compute_dtype", None)
        self.compute_dtype = dt if dt is not None else torch.float32
        self._packed = None

    def set_compute_dtype(self, dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        self._packed = None
        return self

    # ---- weight preparation: BatchNorm folded into packed conv rows, cached until a parameter / buffer changes
    def _signature(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers())) + (self.compute_dtype,)

    def _fold(self, conv, bn):
        dt = self.compute_dtype
        w = conv.weight.detach()
        co, ci, kh, kw = w.shape
        if kh == 1:
            w2d = w.reshape(co, ci).contiguous()
        elif kh == 3:  # [co,ci,3,3] -> [co][ky][kx][ci] (the implicit-GEMM K order of umr_gemm_nt)
            st = w.stride()
            w2d = torch.empty((co, 9 * ci), dtype=torch.float32, device=w.device)
            ops.permute4(w, w2d, (co, 3, 3, ci), (st[0], st[2], st[3], st[1]))
        else:          # 7x7 stem: K order (c, ky, kx) as im2col_nchw produces it
            w2d = w.reshape(co, ci * kh * kw).contiguous()
        K = w2d.shape[1]
        ldk = (K + 7) // 8 * 8
        return ops.bn_fold(w2d, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, _BN_EPS, ldk, dt)

    def _pack(self):
        sig = self._signature()
        if self._packed is not None and self._packed[0] == sig:
            return self._packed[1]
        rb, dt = self.classifier_backbone, self.compute_dtype
        P = {"stem": self._fold(rb.conv1, rb.bn1), "blocks": []}
        for li in range(4):
            for blk in getattr(rb, f"layer{li + 1}"):
                e = {"c1": self._fold(blk.conv1, blk.bn1), "c2": self._fold(blk.conv2, blk.bn2), "c3": self._fold(blk.conv3, blk.bn3),
                     "stride": blk.stride, "down": None}
                if blk.downsample is not None:
                    e["down"] = self._fold(blk.downsample[0], blk.downsample[1])
                P["blocks"].append(e)
        P["fc"] = (ops.cast(rb.fc.weight.detach(), dt), rb.fc.bias.detach().float())
        P["head"] = (ops.cast(self.binary_classification_head.weight.detach(), dt), self.binary_classification_head.bias.detach().float())
        self._packed = (sig, P)
        return P

    def forward(self, images):
        if not images.is_cuda:
            raise RuntimeError("unmore_amd.Binary_Classifier runs on the MI355X only (no CPU fallback); move the model and inputs to the GPU")
        if self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("unmore_amd.Binary_Classifier implements the reference's inference path only: call .eval() "
                                      "and run under torch.no_grad() (object_reasoning.py:86-90,505-506)")
        in_dtype = images.dtype
        dt = self.compute_dtype
        P = self._pack()
        x = images.float().contiguous()
        B = x.shape[0]
        # stem: 7x7 s2 p3 conv (+BN+ReLU) as im2col + GEMM, then max-pool 3x3 s2 p1
        w, b = P["stem"]
        cols, H, W = ops.im2col_nchw(x, 7, 7, 2, 3, w.shape[1], dt)
        h = ops.gemm_nt(cols, w, b, act=L.ACT_RELU)
        del cols
        h = ops.maxpool3x3s2(h.view(B, H, W, 64))
        for e in P["blocks"]:
            _, H, W, C = h.shape
            x2d = h.view(-1, C)
            a = ops.gemm_nt(x2d, e["c1"][0], e["c1"][1], act=L.ACT_RELU)
            planes = a.shape[1]
            s = e["stride"]
            bmid = ops.gemm_nt(a.view(B, H, W, planes), e["c2"][0], e["c2"][1], conv=(2 if s == 2 else 1), act=L.ACT_RELU)
            Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
            if e["down"] is not None:
                if s == 2:  # 1x1 stride-2 conv = 1x1 conv of the subsampled map
                    xs = torch.empty((B, Ho, Wo, C), dtype=dt, device=h.device)
                    ops.permute4(h, xs, (B, Ho, Wo, C), (H * W * C, 2 * W * C, 2 * C, 1))
                    x2d = xs.view(-1, C)
                idt = ops.gemm_nt(x2d, e["down"][0], e["down"][1])
            else:
                idt = x2d
            h = ops.gemm_nt(bmid, e["c3"][0], e["c3"][1], aux=idt, act=L.ACT_RELU).view(B, Ho, Wo, planes * 4)
        _, H, W, C = h.shape
        pooled = ops.segsum(h, B, H * W, C, H * W * C, C, out_f32=True)        # AdaptiveAvgPool2d((1,1)): sum over pixels ...
        pooled = ops.cast(pooled, dt, scale=1.0 / (H * W))                       # ... times 1/HW
        logits = ops.gemm_nt(pooled, P["fc"][0], P["fc"][1])
        pred = ops.gemm_nt(logits, P["head"][0], P["head"][1], act=L.ACT_SIGMOID, out_f32=True)
        return pred.to(in_dtype)  ## [B, 1]
