"""unmore_amd.object_scoring.Object_Scoring on the GPU against the REFERENCE's own code run on the CPU (tests/golden/scoring.npz, made by
tests/golden/make_golden_r6_scoring.py from object_scoring.py:112-157 and the source lines :182-245 of main_object_scoring), with the
stand-in networks of tests/discovery_stubs.py on both sides: field maxima, the tight box and the area of every proposal's pasted union
mask (BEFORE NMS: all 149 / 130 proposals, among them the whole image and an empty one), the boxes NMS keeps, their masks pixel for
pixel, the four factors and the final score."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from discovery_stubs import FieldsFromCrop, ObjectFraction

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "scoring.npz"))
SCENES = {"a": (240, 320, 0, 4), "b": (200, 288, 5, 6)}
PIX = 1.0 / (128 * 128)


@pytest.mark.parametrize("tag", list(SCENES))
def test_scoring_equals_the_reference(tag):
    from unmore_amd import _lib as L, synth
    from unmore_amd.object_scoring import Object_Scoring
    from unmore_amd.ops import _p, _stream
    H, W, seed, nobj = SCENES[tag]
    image = torch.from_numpy(synth.reasoning_scene(H, W, seed, nobj)).to(DEV)
    raw = G[f"{tag}_raw_proposals"]
    osc = Object_Scoring(Namespace(), DEV, objectness_model=FieldsFromCrop(), binary_classifier_model=ObjectFraction())
    # ---- every proposal, before NMS: the launch that replaces the two paste loops, the union and the pycocotools boxes (:189-235)
    pred = osc.get_prediction_with_proposals(image, torch.from_numpy(raw))
    assert float((pred["pred_existence_scores"].cpu() - torch.from_numpy(G[f"{tag}_existence"])).abs().max()) <= 4 * PIX
    N = len(raw)
    ib = torch.from_numpy(np.stack([np.floor(raw[:, 0]), np.floor(raw[:, 1]), np.ceil(raw[:, 2]), np.ceil(raw[:, 3])], 1).astype(np.int32)).to(DEV)
    stats = torch.empty((N, 5), dtype=torch.int32, device=DEV)
    maxima = torch.empty((N, 2), dtype=torch.float32, device=DEV)
    sdf, cen = pred["pred_boundary_fields"].contiguous(), pred["pred_center_fields"].contiguous()
    L.check(L.lib().umr_mask_paste_stats(_p(sdf), _p(cen), _p(ib), N, 128, H, W, _p(stats), _p(maxima), _stream()), "umr_mask_paste_stats")
    st = stats.cpu().numpy()
    assert float(np.abs(maxima[:, 0].cpu().numpy() - G[f"{tag}_max_center"]).max()) <= 2e-6
    assert float(np.abs(maxima[:, 1].cpu().numpy() - G[f"{tag}_max_boundary"]).max()) <= 2e-6
    # a crop pixel that sits on a mask threshold may fall on the other side (the crops agree with F.interpolate's to the last bits, not
    # bit for bit): areas to a handful of pixels, tight boxes to one pixel, and exactly for all but a few proposals
    ref_area, ref_tight = G[f"{tag}_union_area"], G[f"{tag}_tight"]
    d_area = np.abs(st[:, 4] - ref_area)
    d_box = np.abs(st[:, :4] - ref_tight).max(axis=1)
    print(f"scene {tag}: {N} proposals; areas exact for {int((d_area == 0).sum())}, worst {int(d_area.max())} px of {int(ref_area.max())}; "
          f"tight boxes exact for {int((d_box == 0).sum())}, worst {d_box.max():.0f} px; empty masks {int((st[:, 4] == 0).sum())}")
    assert int((st[:, 4] == 0).sum()) == int((ref_area == 0).sum()) >= 1 and np.array_equal(st[ref_area == 0], np.zeros((int((ref_area == 0).sum()), 5)))
    assert d_area.max() <= 4 and (d_area == 0).mean() >= 0.98          # (measured: every area and every box exact)
    assert d_box.max() <= 1 and (d_box == 0).mean() >= 0.98
    # ---- the method: NMS, masks of the survivors, scores (:238-255)
    out = osc.score_image(image, raw.tolist())
    assert out["keep"].cpu().numpy().tolist() == G[f"{tag}_nms"].tolist()
    shape = tuple(G[f"{tag}_final_masks_shape"])
    ref_masks = np.unpackbits(G[f"{tag}_final_masks_packed"])[:int(np.prod(shape))].reshape(shape)
    got_masks = out["masks"].cpu().numpy()
    assert got_masks.shape == shape and got_masks.dtype == np.uint8
    assert int((got_masks != ref_masks).sum()) <= 4 * shape[0]
    assert float(np.abs(out["tight_bboxes"].cpu().numpy() - G[f"{tag}_tight"][G[f"{tag}_nms"]]).max()) <= 1
    assert out["score"].dtype == np.float64 and out["score"].shape == G[f"{tag}_score"].shape
    assert np.allclose(out["score"], G[f"{tag}_score"], rtol=2e-3, atol=1e-6)
    ann = osc.annotations(17, out)
    assert len(ann) == shape[0] and ann[0]["image_id"] == 17 and ann[0]["category_id"] == 1 and len(ann[0]["bbox"]) == 4
    assert osc.main_object_scoring([(17, image), (18, image)], {"17": raw.tolist()})[0]["score"] == ann[0]["score"]


def test_pasted_mask_equals_torch_resize_of_a_random_mask():
    """the paste arithmetic on its own: random fields, boxes of odd sizes (upscaling, downscaling, one pixel wide, the whole image) --
    every pixel of the pasted union mask against torch: round(F.interpolate(mask.float(), bilinear, align_corners=False)) of each mask,
    OR-ed (what torchvision's Resize does to an integer tensor)"""
    import torch.nn.functional as F
    from unmore_amd import _lib as L
    from unmore_amd.ops import _p, _stream
    g = torch.Generator().manual_seed(5)
    H, W, S = 150, 210, 128
    boxes = torch.tensor([[0, 0, W, H], [10, 20, 74, 84], [5, 7, 6, 140], [30, 40, 200, 43], [100, 3, 209, 149], [17, 90, 81, 122], [50, 50, 178, 178 - 28]], dtype=torch.int32)
    N = len(boxes)
    sdf = (torch.randn(N, S, S, generator=g) * 0.7).contiguous()
    cen = (torch.randn(N, 2, S, S, generator=g) * 0.4).contiguous()
    # smooth them a little so that masks have structure at several scales
    sdf = F.avg_pool2d(sdf[:, None], 5, 1, 2)[:, 0].contiguous()
    cen = F.avg_pool2d(cen, 3, 1, 1).contiguous()
    sel = torch.arange(N, dtype=torch.int64, device=DEV)
    masks = torch.empty((N, H, W), dtype=torch.uint8, device=DEV)
    stats = torch.empty((N, 5), dtype=torch.int32, device=DEV)
    maxima = torch.empty((N, 2), dtype=torch.float32, device=DEV)
    sd, cd, bd = sdf.to(DEV), cen.to(DEV), boxes.to(DEV)
    L.check(L.lib().umr_mask_paste(_p(sd), _p(cd), _p(bd), _p(sel), N, S, H, W, _p(masks), _stream()), "umr_mask_paste")
    L.check(L.lib().umr_mask_paste_stats(_p(sd), _p(cd), _p(bd), N, S, H, W, _p(stats), _p(maxima), _stream()), "umr_mask_paste_stats")
    mc = (torch.norm(cen, dim=1) > 0.5).to(torch.int64)
    mb = (torch.sigmoid(sdf) > 0.5).to(torch.int64)
    bad = 0
    for n, (x1, y1, x2, y2) in enumerate(boxes.tolist()):
        ref = torch.zeros((H, W), dtype=torch.int64)
        for m in (mc[n], mb[n]):
            r = torch.round(F.interpolate(m[None, None].float(), size=(y2 - y1, x2 - x1), mode="bilinear", align_corners=False))[0, 0].to(torch.int64)
            ref[y1:y2, x1:x2] += r
        ref = (ref > 0).to(torch.uint8)
        got = masks[n].cpu()
        bad += int((got != ref).sum())
        ys, xs = torch.nonzero(ref, as_tuple=True)
        want = [int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1, int(ref.sum())] if len(ys) else [0] * 5
        if torch.equal(got, ref):
            assert stats[n].cpu().tolist() == want, n
    print("pasted pixels differing from torch's resize:", bad, "of", N * H * W)
    assert bad <= 4        # values exactly on 0.5 (a 2x upscale puts many there) round the same way; only last-bit cases may differ


def test_scoring_with_the_real_networks_runs():
    """Object_Scoring around unmore_amd's own ObjectnessNet and Binary_Classifier (hash-initialised: the scores mean nothing, the plumbing is
    what runs): shapes, dtypes, finite scores, masks inside their tight boxes"""
    from unmore_amd import synth
    from unmore_amd.binary_classifier import Binary_Classifier
    from unmore_amd.hashrng import hash_init
    from unmore_amd.object_scoring import Object_Scoring
    from unmore_amd.objectness_net import ObjectnessNet
    args = Namespace(use_bg_sdf=True, sdf_activation="tanh")
    net = ObjectnessNet(DEV, 128, "dpt_base", args)
    spec = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.peak_edited_state_dict(spec, "base").items()}, strict=True)
    osc = Object_Scoring(args, DEV, objectness_model=net.to(DEV), binary_classifier_model=Binary_Classifier(DEV, 128, args).to(DEV))
    image = torch.from_numpy(synth.blob_images(1, 160, 224, seed=4)[0]).to(DEV)
    raw = [[10.3, 12.7, 90.2, 80.9], [100.0, 20.0, 220.5, 150.1], [0.0, 0.0, 224.0, 160.0], [30.0, 90.0, 60.0, 120.0]]
    out = osc.score_image(image, raw)
    K = len(out["keep"])
    assert 1 <= K <= len(raw) and out["masks"].shape == (K, 160, 224) and out["tight_bboxes"].shape == (K, 4)
    assert np.isfinite(out["score"]).all() and out["score"].dtype == np.float64
    for k in range(K):
        ys, xs = torch.nonzero(out["masks"][k], as_tuple=True)
        x1, y1, x2, y2 = out["tight_bboxes"][k].tolist()
        if len(ys):
            assert [int(xs.min()), int(ys.min()), int(xs.max()) + 1, int(ys.max()) + 1] == [int(x1), int(y1), int(x2), int(y2)]
    assert osc.score_image(image, []) is None
