"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (existence classifier).

Plain-PyTorch (CPU, NCHW) restatement of the reference's `Binary_Classifier` in the mode its callers use it:
`sigmoid(Linear(1000,1)(resnet50(images)))` in eval mode (models/objectness_net.py:205-223; object_reasoning.py:64-90,
491-523; object_scoring.py:65-90,123-140).  Only `tests/` (and `tests/golden/make_golden_classifier.py`) may import this
file; the product (`unmore_amd/`) never does.

Parity status:
  * PINNED for what lives in the reference tree -- the composition backbone -> Linear(1000,1) -> sigmoid and the
    `classifier_backbone.` / `binary_classification_head.` key prefixes: `tests/golden/clf_*.npz|txt` were produced by
    running the reference's own `Binary_Classifier` class in the build container.
  * UNPINNED at the torchvision boundary: `torchvision.models.resnet50` (torchvision 0.14.1 per README.md:25) is absent
    from /root/reference and from this image, and the reference ships no tests or vectors for it.  `resnet50_forward`
    restates its published architecture (He et al. 2015 "v1.5": stride on the bottleneck's 3x3 conv; BatchNorm eps 1e-5;
    MaxPool 3x3 s2 p1; AdaptiveAvgPool(1); fc 2048->1000) and its state-dict key names; the golden generator had to
    supply the same restatement as the `torchvision.models.resnet50` stand-in.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # planes, blocks, stride of the first block
BN_EPS = 1e-5


def resnet50_spec(prefix=""):
    """Ordered (name, shape) list of torchvision resnet50's state dict (320 entries)."""
    out = []

    def bn(name, c):
        out.extend([(name + ".weight", (c,)), (name + ".bias", (c,)), (name + ".running_mean", (c,)), (name + ".running_var", (c,)),
                    (name + ".num_batches_tracked", ())])

    out.append((prefix + "conv1.weight", (64, 3, 7, 7)))
    bn(prefix + "bn1", 64)
    inplanes = 64
    for li, (planes, blocks, _) in enumerate(LAYERS):
        for bi in range(blocks):
            p = f"{prefix}layer{li + 1}.{bi}."
            out.append((p + "conv1.weight", (planes, inplanes, 1, 1)))
            bn(p + "bn1", planes)
            out.append((p + "conv2.weight", (planes, planes, 3, 3)))
            bn(p + "bn2", planes)
            out.append((p + "conv3.weight", (planes * 4, planes, 1, 1)))
            bn(p + "bn3", planes * 4)
            if bi == 0:
                out.append((p + "downsample.0.weight", (planes * 4, inplanes, 1, 1)))
                bn(p + "downsample.1", planes * 4)
            inplanes = planes * 4
    out.append((prefix + "fc.weight", (1000, 2048)))
    out.append((prefix + "fc.bias", (1000,)))
    return out


def state_dict_spec():
    """The reference `Binary_Classifier`'s 322-key schema (objectness_net.py:216-217)."""
    return resnet50_spec("classifier_backbone.") + [("binary_classification_head.weight", (1, 1000)), ("binary_classification_head.bias", (1,))]


def hash_state(tag, hash_uniform):
    """Deterministic synthetic checkpoint.  hash_uniform(name, shape, lo, hi) -> float32 ndarray (unmore_amd.hashrng.uniform).
    Scales keep activations O(1) through the 16 residual blocks: variance-preserving conv weights, running_var in
    [0.5, 1.5], the last BatchNorm of each residual branch scaled by 0.25."""
    sd = OrderedDict()
    for name, shape in state_dict_spec():
        key = f"{tag}:{name}"
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_var"):
            sd[name] = torch.from_numpy(hash_uniform(key, shape, 0.5, 1.5))
        elif name.endswith("running_mean"):
            sd[name] = torch.from_numpy(hash_uniform(key, shape, -0.2, 0.2))
        elif name.endswith(".bias"):
            sd[name] = torch.from_numpy(hash_uniform(key, shape, -0.1, 0.1))
        elif len(shape) == 1:  # BatchNorm scale
            last = (".bn3." in name) or (".downsample.1." in name)
            sd[name] = torch.from_numpy(hash_uniform(key, shape, 0.9, 1.1)) * (0.25 if ".bn3." in name else (0.7 if last else 1.0))
        else:
            fan_in = int(np.prod(shape[1:]))
            b = (3.0 / fan_in) ** 0.5 * (1.4 if len(shape) == 4 else 1.0)  # convs feed a ReLU: He-style gain
            sd[name] = torch.from_numpy(hash_uniform(key, shape, -b, b))
    return sd


def _bn(x, sd, name):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"], False, 0.0, BN_EPS)


def resnet50_forward(sd, x, prefix=""):
    """torchvision.models.resnet50(...).eval()(x): [B,3,H,W] -> [B,1000] (restated; see the header)."""
    x = F.conv2d(x, sd[prefix + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(x, sd, prefix + "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, (planes, blocks, stride) in enumerate(LAYERS):
        for bi in range(blocks):
            p = f"{prefix}layer{li + 1}.{bi}."
            s = stride if bi == 0 else 1
            idt = x
            out = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1"))
            out = F.relu(_bn(F.conv2d(out, sd[p + "conv2.weight"], None, stride=s, padding=1), sd, p + "bn2"))
            out = _bn(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3")
            if bi == 0:
                idt = _bn(F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=s), sd, p + "downsample.1")
            x = F.relu(out + idt)
    x = F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)
    return F.linear(x, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])


def forward(sd, images):
    """`Binary_Classifier.forward` (objectness_net.py:220-223): [B,3,H,W] -> [B,1] probabilities."""
    logits = resnet50_forward(sd, images, "classifier_backbone.")
    return torch.sigmoid(F.linear(logits, sd["binary_classification_head.weight"], sd["binary_classification_head.bias"]))


def existence_scores(sd, image, proposals, crop_resize):
    """`existence_checking` (object_reasoning.py:491-523): crops -> 128x128 bilinear -> classifier, batches of 128.
    `crop_resize(image, boxes)` is the oracle's proposal crop (objectness_oracle.crop_resize)."""
    scores = []
    for i in range(0, len(proposals), 128):
        crops = crop_resize(image, proposals[i:i + 128])
        scores.append(forward(sd, crops.float()))
    return torch.cat(scores, 0).squeeze(1)
