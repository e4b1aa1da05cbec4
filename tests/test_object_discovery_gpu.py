"""unmore_amd.object_discovery.Object_Discovery on the GPU against the REFERENCE's `Object_Discovery` (object_reasoning.py:43-665) run on
the CPU (tests/golden/discovery.npz, made by tests/golden/make_golden_r6_discovery.py): the same stand-in "networks" on both sides
(tests/discovery_stubs.py read object-like fields back out of the crop), so what is compared is everything AROUND the networks --
crops, existence threshold, union mask / erosion / anti-centre score / peak, the box splits, the boundary rounds with their filters and
label rules -- on two scenes.  Then the pieces the reference run cannot pin: NMS against the oracle's restatement, the sdf-only
forward against the full forward, and the fixed-point carry against evaluating every box in every round."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from discovery_stubs import FieldsFromCrop, ObjectFraction

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "discovery.npz"))
SCENES = {"a": (240, 320, 0, 4), "b": (200, 288, 5, 6)}      # tests/golden/make_golden_r6_discovery.py::SCENES
# the crops differ from F.interpolate's in the last bits (another order of the same four products); a pixel that sits on a threshold
# may fall on the other side: scores that are pixel counts are compared to a few pixels, box coordinates to a hundredth of a pixel
PIX = 1.0 / (128 * 128)


def _od():
    from unmore_amd.object_discovery import Object_Discovery
    args = Namespace()                                        # the reference's defaults (object_reasoning.py:701-710)
    return Object_Discovery(args, DEV, objectness_model=FieldsFromCrop(), binary_classifier_model=ObjectFraction())


def _image(tag):
    from unmore_amd import synth
    H, W, seed, nobj = SCENES[tag]
    return torch.from_numpy(synth.reasoning_scene(H, W, seed, nobj)).to(DEV)


@pytest.mark.parametrize("tag", list(SCENES))
def test_existence_and_centre_reasoning_equal_the_reference(tag):
    od = _od()
    image = _image(tag)
    od.height, od.width = image.shape[-2], image.shape[-1]
    p0 = torch.tensor(od.generate_random_proposal(height=od.height, width=od.width)).to(DEV)
    ex = od.existence_checking(image, p0)["existence_scores"]
    ref_ex = torch.from_numpy(G[f"{tag}_existence0"])
    assert float((ex - ref_ex).abs().max()) <= 4 * PIX
    thr = od.args.class_score_thres
    assert float((ref_ex - thr).abs().min()) > 8 * PIX, "fixture: a score sits on the existence threshold"
    keep = (ex >= thr)
    assert torch.equal(keep, ref_ex >= thr)
    cr = od.center_reasoning(image, p0[keep.to(DEV)])
    passed, split = cr["proposals_pass_singularity"], cr["splited_new_proposals"]
    # which boxes pass, and where the others are cut (the peak index decides the cut: bit-exact or a different box)
    assert passed.dtype == torch.float64 and np.array_equal(passed.cpu().numpy(), G[f"{tag}_pass1"])
    assert split.dtype == torch.float64 and split.shape == G[f"{tag}_split1"].shape
    assert np.array_equal(split.cpu().numpy(), G[f"{tag}_split1"])
    # second pass on the split boxes (main_object_discovery :632-638)
    ex2 = od.existence_checking(image, split)["existence_scores"]
    ref_ex2 = torch.from_numpy(G[f"{tag}_existence1"])
    assert float((ex2 - ref_ex2).abs().max()) <= 4 * PIX
    sure = (ref_ex2 - thr).abs() > 8 * PIX
    assert torch.equal((ex2 >= thr)[sure], (ref_ex2 >= thr)[sure])
    props2 = split[(ref_ex2 >= thr).to(DEV)]
    cr2 = od.center_reasoning(image, props2)
    assert np.array_equal(cr2["proposals_pass_singularity"].cpu().numpy(), G[f"{tag}_pass2"])
    got2, ref2 = cr2["splited_new_proposals"].cpu().numpy(), G[f"{tag}_split2"]
    assert got2.shape == ref2.shape
    # the split boxes of this pass include slivers a few pixels high stretched to 128 rows: neighbouring rows of their score maps agree
    # to 1e-10, and which of two such rows is "the" maximum hangs on the crop's last bit.  Rows must be identical wherever the peak
    # leads by more than 1e-6; where it does not, the reference's cut must be one of the tied pixels
    from unmore_amd import reasoning
    sdf, cen = od.get_prediction_with_proposals(props2, image)
    mx, am, sc = reasoning.center_peaks(sdf, cen, return_scores=True)
    fail = torch.nonzero(mx > od.args.center_score_max_thres).flatten()
    top2 = torch.topk(sc[fail].flatten(1), 2, dim=1).values
    tied = ((top2[:, 0] - top2[:, 1]) <= 1e-6).cpu().numpy()
    differs = (got2 != ref2).any(axis=1).reshape(-1, 4).any(axis=1)
    assert not bool((differs & ~tied).any()), np.nonzero(differs & ~tied)[0]
    assert int(differs.sum()) <= 0.05 * len(differs)
    for b in np.nonzero(differs)[0]:
        box = props2[fail[b]].cpu().numpy()
        xr = (ref2[4 * b, 2] - box[0]) / (box[2] - box[0]) * 128            # the reference's peak, back on the 128 x 128 map
        yr = (ref2[4 * b + 2, 3] - box[1]) / (box[3] - box[1]) * 128
        assert abs(xr - round(xr)) < 1e-6 and abs(yr - round(yr)) < 1e-6
        assert float(top2[b, 0] - sc[fail[b], int(round(yr)), int(round(xr))]) <= 1e-6


@pytest.mark.parametrize("tag", list(SCENES))
def test_boundary_rounds_equal_the_reference(tag):
    od = _od()
    image = _image(tag)
    cur = torch.from_numpy(G[f"{tag}_boundary_in"]).to(DEV)
    labels = torch.zeros(len(cur), device=DEV)
    for r in range(3):                                   # round by round, as boundary_reasoning does (:598-606)
        cur, labels = od.filter_small_proposal(cur, labels)
        out = od.optimize_one_image_single_round(image, cur, labels)
        cur, labels = out["updated_bboxes"], out["labels"]
        ref_b, ref_l = G[f"{tag}_round{r}_boxes"], G[f"{tag}_round{r}_labels"]
        assert cur.dtype == torch.float32 and cur.shape == ref_b.shape
        assert np.array_equal(labels.cpu().numpy(), ref_l), (r, int((labels.cpu().numpy() != ref_l).sum()))
        assert float(np.abs(cur.cpu().numpy() - ref_b).max()) <= 1e-2, r
    res = od.boundary_reasoning(image, torch.from_numpy(G[f"{tag}_boundary_in"]).to(DEV), n_round=od.args.n_round)
    ref_b, ref_l = G[f"{tag}_final_boxes"], G[f"{tag}_final_labels"]
    assert res["proposals"].shape == ref_b.shape and np.array_equal(res["labels"].cpu().numpy(), ref_l)
    err = float(np.abs(res["proposals"].cpu().numpy() - ref_b).max())
    print(f"scene {tag}: {len(ref_l)} boxes after {od.args.n_round} rounds ({int((ref_l == 1).sum())} good), max box error {err:.2e} px; "
          f"rounds evaluated {od.stats['boundary_rounds']}, crops {od.stats['boundary_crops']} (the reference evaluates {od.args.n_round} x all)")
    assert err <= 2e-2


@pytest.mark.parametrize("tag", list(SCENES))
def test_fixed_point_carry_changes_nothing(tag):
    """boxes that a round labels good and leaves where they were are not evaluated again (object_discovery.py): bit-identical to
    evaluating every box in every round, with fewer crops"""
    od = _od()
    image = _image(tag)
    start = torch.from_numpy(G[f"{tag}_boundary_in"]).to(DEV)
    a = od.boundary_reasoning(image, start)
    s_carry = dict(od.stats)
    od.carry_fixed_points = False
    b = od.boundary_reasoning(image, start)
    s_full = dict(od.stats)
    assert torch.equal(a["proposals"], b["proposals"]) and torch.equal(a["labels"], b["labels"])
    assert s_full["boundary_rounds"] == od.args.n_round and s_carry["boundary_crops"] < 0.5 * s_full["boundary_crops"], (s_carry, s_full)
    # sharing within a round only (no memory of earlier rounds): the same again
    od.carry_fixed_points, od.remember_crops = True, False
    d = od.boundary_reasoning(image, start)
    s_round = dict(od.stats)
    od.remember_crops = True
    assert torch.equal(a["proposals"], d["proposals"]) and torch.equal(a["labels"], d["labels"])
    assert s_carry["boundary_distinct_crops"] < s_round["boundary_distinct_crops"] <= s_round["boundary_crops"]
    # and with every box cropped on its own (no sharing between boxes with equal integer corners): the same again
    od.carry_fixed_points, od.share_equal_crops = True, False
    c = od.boundary_reasoning(image, start)
    s_noshare = dict(od.stats)
    assert torch.equal(a["proposals"], c["proposals"]) and torch.equal(a["labels"], c["labels"])
    assert s_noshare["boundary_distinct_crops"] == s_noshare["boundary_crops"] and s_carry["boundary_distinct_crops"] <= s_carry["boundary_crops"]
    print(f"scene {tag}: crops evaluated -- reference flow {s_full['boundary_crops']}, fixed points carried {s_carry['boundary_crops']}, "
          f"equal crops shared within a round {s_round['boundary_distinct_crops']}, within the image {s_carry['boundary_distinct_crops']}")


def test_discover_image_end_to_end_and_nms():
    """the whole per-image sequence (main_object_discovery :619-662): the boxes that reach NMS are the fixture's good boxes, and the
    kept ones are what the oracle's NMS keeps of them"""
    from oracle import objectness_oracle as orc
    od = _od()
    boxes = od.discover_image(_image("a"))
    ref_in = G["a_final_boxes"][G["a_final_labels"] == 1]
    keep = orc.nms(ref_in, np.ones(len(ref_in)), 0.5)
    assert boxes is not None and boxes.dtype == torch.float32 and len(boxes) == len(keep)
    assert float(np.abs(boxes.cpu().numpy() - ref_in[keep]).max()) <= 2e-2
    out = od.main_object_discovery([(7, _image("a")), (9, torch.full((3, 64, 64), -1.0))])     # second image: nothing exists -> skipped
    assert list(out) == [7] and np.array_equal(out[7], boxes.cpu().numpy())


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 300, 1500])
def test_nms_kernel_equals_the_oracle(n):
    from oracle import objectness_oracle as orc
    from unmore_amd import reasoning
    rng = np.random.default_rng(n)
    c = rng.uniform(0, 200, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 60, (n, 2)).astype(np.float32)
    b = np.concatenate([c - wh / 2, c + wh / 2], axis=1).astype(np.float32)
    if n >= 4:
        b[n // 2] = b[0]                                  # an exact duplicate
        b[n // 3, 2:] = b[n // 3, :2]                     # a degenerate (zero-area) box
    for scores in (np.ones(n, np.float32), rng.uniform(0, 1, n).astype(np.float32), np.round(rng.uniform(0, 1, n) * 4).astype(np.float32) / 4):
        for thr in (0.3, 0.5):
            got = reasoning.nms(torch.from_numpy(b).to(DEV), torch.from_numpy(scores).to(DEV), thr)
            ref = orc.nms(b, scores, thr)
            assert got.dtype == torch.int64 and got.cpu().numpy().tolist() == ref.tolist(), (n, thr)
    assert reasoning.nms(torch.zeros((0, 4), device=DEV), torch.zeros(0, device=DEV), 0.5).numel() == 0


@pytest.mark.parametrize("backbone,dtype_name", [("dpt_tiny", "float32"), ("dpt_base", "float32"), ("dpt_base", "bfloat16")])
def test_single_head_prediction_equals_the_full_prediction(backbone, dtype_name):
    """get_prediction(heads=("sdf_maps",)) -- what the boundary rounds call -- returns the full call's map bit for bit (eager and
    replayed from its own graph), and only that key; likewise for the centre field"""
    from unmore_amd.hashrng import hash_init
    from unmore_amd.objectness_net import ObjectnessNet
    net = ObjectnessNet(DEV, 128, backbone, Namespace(use_bg_sdf=True, sdf_activation="tanh"))
    tag = {"dpt_tiny": "tiny", "dpt_base": "base"}[backbone]
    net.load_state_dict({k: torch.from_numpy(hash_init(k, tuple(v.shape), tag)) for k, v in net.state_dict().items()}, strict=True)
    net = net.to(DEV).eval()
    net.set_compute_dtype(getattr(torch, dtype_name))
    x = torch.rand(6, 3, 128, 128, device=DEV)
    with torch.no_grad():
        full = net.get_prediction(x)
        for _ in range(4):                                # eager, eager, capture + replay, replay
            only_s = net.get_prediction(x, heads=("sdf_maps",))
            only_c = net.get_prediction(x, heads=("center_fields",))
            assert list(only_s) == ["sdf_maps"] and list(only_c) == ["center_fields"]
            assert torch.equal(only_s["sdf_maps"], full["sdf_maps"]) and torch.equal(only_c["center_fields"], full["center_fields"])
        both = net.get_prediction(x, heads=("center_fields", "sdf_maps"))
        assert torch.equal(both["sdf_maps"], full["sdf_maps"]) and torch.equal(both["center_fields"], full["center_fields"])
        with pytest.raises(ValueError):
            net.get_prediction(x, heads=("sdf",))
    for p in net.parameters():
        p.requires_grad = True
    with pytest.raises(RuntimeError):
        net.get_prediction(x, heads=("sdf_maps",))


def test_discovery_with_the_real_networks_runs():
    """Object_Discovery around unmore_amd's own ObjectnessNet and Binary_Classifier (hash-initialised: the boxes mean nothing, the
    plumbing -- crops on the device, sdf-only rounds, graphs per batch shape, NMS -- is what runs)"""
    from unmore_amd.binary_classifier import Binary_Classifier
    from unmore_amd.hashrng import hash_init
    from unmore_amd.object_discovery import Object_Discovery
    from unmore_amd.objectness_net import ObjectnessNet
    from unmore_amd import synth
    args = Namespace(use_bg_sdf=True, sdf_activation="tanh", n_round=4, class_score_thres=0.0, max_sdf_thres=-2.0)
    net = ObjectnessNet(DEV, 128, "dpt_tiny", args)
    net.load_state_dict({k: torch.from_numpy(hash_init(k, tuple(v.shape), "tiny")) for k, v in net.state_dict().items()}, strict=True)
    clf = Binary_Classifier(DEV, 128, args)
    od = Object_Discovery(args, DEV, objectness_model=net.to(DEV), binary_classifier_model=clf.to(DEV))
    image = torch.from_numpy(synth.blob_images(1, 96, 128, seed=3)[0]).to(DEV)
    boxes = od.discover_image(image)
    assert boxes is None or (boxes.dim() == 2 and boxes.shape[1] == 4 and bool(torch.isfinite(boxes).all()))
    assert od.stats.get("boundary_rounds", 0) <= 4


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
def test_centre_reasoning_through_the_pipelined_sweep_equals_the_plain_path(dtype_name):
    """with unmore_amd's own net, center_reasoning runs the proposals through reasoning.sweep_proposals (three streams; fp32:
    certificate-driven precision): the boxes that pass and the split boxes are those of the plain batch-by-batch path"""
    from unmore_amd import synth
    from unmore_amd.hashrng import hash_init
    from unmore_amd.object_discovery import Object_Discovery
    from unmore_amd.objectness_net import ObjectnessNet
    args = Namespace(use_bg_sdf=True, sdf_activation="tanh")
    net = ObjectnessNet(DEV, 128, "dpt_base", args)
    spec = {k: tuple(v.shape) for k, v in net.state_dict().items()}          # hash weights + the documented edits that give peaks
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.peak_edited_state_dict(spec, "base").items()}, strict=True)
    net = net.to(DEV)
    net.set_compute_dtype(getattr(torch, dtype_name))
    od = Object_Discovery(args, DEV, objectness_model=net, binary_classifier_model=ObjectFraction())
    image = torch.from_numpy(synth.blob_images(1, 240, 320, seed=11)[0]).to(DEV)
    props = torch.tensor(od.generate_random_proposal(240, 320)).to(DEV)
    a = od.center_reasoning(image, props)
    od.pipelined_center_sweep = False
    b = od.center_reasoning(image, props)
    n_pass, n_split = len(a["proposals_pass_singularity"]), len(a["splited_new_proposals"])
    print(f"{dtype_name}: {len(props)} proposals -> {n_pass} pass, {n_split // 4} split")
    assert n_pass > 0 and n_split > 0, "the fixture should exercise both outcomes"
    assert torch.equal(a["proposals_pass_singularity"], b["proposals_pass_singularity"])
    assert torch.equal(torch.as_tensor(a["splited_new_proposals"]), torch.as_tensor(b["splited_new_proposals"]))


def test_analyze_cc_branch_equals_the_reference():
    """--analyze_cc (object_reasoning.py:562-573): the connected components (scipy, on the host, as in the reference) of the union masks of
    the boxes that pass, enlarged 1.5x and appended to the split boxes -- rows and dtype of the reference's result"""
    od = _od()
    od.args.analyze_cc = True
    image = _image("a")
    od.height, od.width = image.shape[-2], image.shape[-1]
    props = torch.from_numpy(G["a_proposals0"])[torch.from_numpy(G["a_existence0"]) >= od.args.class_score_thres].to(DEV)
    cr = od.center_reasoning(image, props)
    got, ref = cr["splited_new_proposals"].cpu().numpy(), G["a_cc_split"]
    assert np.array_equal(cr["proposals_pass_singularity"].cpu().numpy(), G["a_cc_pass"])
    assert got.dtype == ref.dtype and got.shape == ref.shape
    n_peak = G["a_split1"].shape[0]
    assert np.array_equal(got[:n_peak], ref[:n_peak])
    # the component boxes come from masks whose threshold pixels may differ in the last bit of a crop: all but a few rows identical
    same = (got[n_peak:] == ref[n_peak:]).all(axis=1)
    print(f"analyze_cc: {len(ref) - n_peak} component boxes, {int(same.sum())} identical")
    assert same.mean() >= 0.95 and float(np.abs(got[n_peak:] - ref[n_peak:]).max()) <= 3


def test_connected_components_kernel_equals_scipy_label():
    """umr_mask_components against scipy.ndimage.label + find_objects (what object_reasoning.py:206-257 calls; here only the checker):
    counts, numbering and boxes on blobs, noise, a spiral whose ends are ~4000 steps apart, a full and an empty mask, 1024 isolated
    pixels (the recorded maximum) -- and the documented failure beyond it"""
    from scipy.ndimage import find_objects, label
    from unmore_amd import reasoning
    from unmore_amd.object_discovery import Object_Discovery as OD
    S = 128
    rng = np.random.default_rng(0)
    masks = []
    yy, xx = np.mgrid[0:S, 0:S]
    blobs = np.zeros((S, S), bool)
    for _ in range(9):
        cy, cx, r = rng.uniform(10, 118), rng.uniform(10, 118), rng.uniform(3, 14)
        blobs |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    masks.append(blobs)
    masks.append(rng.uniform(size=(S, S)) > 0.6)                       # noise: hundreds of components, diagonal contacts
    masks.append(rng.uniform(size=(S, S)) > 0.45)                      # denser: a few big tangled ones
    spiral = np.zeros((S, S), bool)
    y = x = 0
    dy, dx, seg = 0, 1, S - 1
    lo, hi = 0, S - 1
    while hi - lo >= 2:                                                # a square spiral of one-pixel walls, two pixels apart
        spiral[lo, lo:hi + 1] = True; spiral[lo:hi + 1, hi] = True     # noqa: E702
        spiral[hi, lo + 2:hi + 1] = True; spiral[lo + 2:hi + 1, lo + 2] = True   # noqa: E702
        lo += 2; hi -= 2                                               # noqa: E702
        spiral[lo, lo] = True
    masks.append(spiral)
    masks.append(np.ones((S, S), bool))
    masks.append(np.zeros((S, S), bool))
    grid = np.zeros((S, S), bool)
    grid[::4, ::4] = True                                              # 32 x 32 = 1024 isolated pixels
    masks.append(grid)
    m = torch.from_numpy(np.stack(masks)).to(DEV)
    sdf = torch.where(m, 1.0, -1.0)
    counts, boxes = reasoning.mask_components(sdf, torch.zeros((len(masks), 2, S, S), device=DEV))
    counts, boxes = counts.cpu().numpy(), boxes.cpu().numpy()
    for b, mk in enumerate(masks):
        lab, n = label(mk, np.ones((3, 3), dtype=int))
        ref = [[sl[1].start, sl[0].start, sl[1].stop, sl[0].stop] for sl in find_objects(lab)]
        assert counts[b] == n, (b, counts[b], n)
        assert boxes[b, :n].tolist() == ref, b
        assert not boxes[b, n:].any()
    print("components per mask:", counts.tolist())
    cc, single = OD.separate_connected_components(m.to(torch.int64))
    assert single == [int(c == 1) for c in counts] and len(cc["multi"]) == int(sum(c for c in counts if c > 1))
    # the union of the two field masks, not just the sdf's: a centre field switches on what the sdf leaves off
    cen = torch.zeros((1, 2, S, S), device=DEV)
    cen[0, 0, 100:110, 100:110] = 0.8
    c2, b2 = reasoning.mask_components(sdf[:1], cen)
    lab, n = label(masks[0] | (cen[0, 0].cpu().numpy() > 0.5), np.ones((3, 3), dtype=int))
    assert int(c2[0]) == n
    dense = np.zeros((S, S), bool)
    dense[::2, ::2] = True                                             # 4096 isolated pixels: more than the kernel records
    with pytest.raises(RuntimeError, match="more than"):
        OD.separate_connected_components(torch.from_numpy(dense)[None].to(DEV))


def test_single_rounds_on_two_images_do_not_share_remembered_crops():
    """the per-image memory of crop results lives inside boundary_reasoning: optimize_one_image_single_round on image B right after the
    same boxes on image A returns B's results"""
    od = _od()
    a_img, b_img = _image("a"), torch.flip(_image("a"), dims=[-1]).contiguous()
    boxes = torch.from_numpy(G["a_boundary_in"]).to(DEV)
    lab = torch.zeros(len(boxes), device=DEV)
    ra = od.optimize_one_image_single_round(a_img, boxes, lab)
    rb = od.optimize_one_image_single_round(b_img, boxes, lab)
    fresh = _od().optimize_one_image_single_round(b_img, boxes, lab)
    assert torch.equal(rb["updated_bboxes"], fresh["updated_bboxes"]) and torch.equal(rb["labels"], fresh["labels"])
    assert not torch.equal(ra["updated_bboxes"], rb["updated_bboxes"])
    # and two boundary_reasoning calls on different images: each starts with an empty memory
    fa = od.boundary_reasoning(a_img, boxes)
    fb = od.boundary_reasoning(b_img, boxes)
    fb2 = _od().boundary_reasoning(b_img, boxes)
    assert torch.equal(fb["proposals"], fb2["proposals"]) and torch.equal(fb["labels"], fb2["labels"]) and not torch.equal(fa["proposals"], fb["proposals"])


def test_images_in_lock_step_equal_images_one_at_a_time():
    """discover_images / boundary_reasoning_many: the boundary rounds of several images share their net calls; per image the result is
    bit for bit the one-at-a-time result (images of different sizes, one of them without any object)"""
    od = _od()
    imgs = [_image("a"), _image("b"), torch.full((3, 96, 128), -1.0, device=DEV), torch.flip(_image("a"), dims=[-1]).contiguous()]
    one = [od.discover_image(im) for im in imgs]
    many = od.discover_images(imgs)
    assert one[2] is None and many[2] is None
    for o, m in zip(one, many):
        assert (o is None) == (m is None)
        if o is not None:
            assert torch.equal(o, m)
    starts = [torch.from_numpy(G["a_boundary_in"]).to(DEV), torch.from_numpy(G["b_boundary_in"]).to(DEV)]
    sep = [od.boundary_reasoning(imgs[0], starts[0]), od.boundary_reasoning(imgs[1], starts[1])]
    tog = od.boundary_reasoning_many(imgs[:2], starts)
    for s_, t_ in zip(sep, tog):
        assert torch.equal(s_["proposals"], t_["proposals"]) and torch.equal(s_["labels"], t_["labels"])
    res = od.main_object_discovery(list(enumerate(imgs)), images_in_lock_step=3)
    assert sorted(res) == [0, 1, 3] and all(np.array_equal(res[i], one[i].cpu().numpy()) for i in res)


def test_discover_image_with_analyze_cc_runs_end_to_end():
    """the documented command line has --analyze_cc (README.md:176): the whole per-image sequence with the component boxes in the split list
    (both centre-reasoning passes), in lock-step with a second image"""
    od = _od()
    od.args.analyze_cc = True
    a, b = od.discover_images([_image("a"), _image("b")])
    for boxes, tag in ((a, "a"), (b, "b")):
        H, W = SCENES[tag][0], SCENES[tag][1]
        assert boxes is not None and boxes.dim() == 2 and boxes.shape[1] == 4 and len(boxes) >= 1 and bool(torch.isfinite(boxes).all())
        assert float(boxes[:, 0].min()) >= 0 and float(boxes[:, 1].min()) >= 0 and float(boxes[:, 2].max()) <= W and float(boxes[:, 3].max()) <= H
    one = od.discover_image(_image("a"))
    assert torch.equal(one, a)


def test_construction_from_checkpoints_as_the_reference_does(tmp_path):
    """Object_Discovery(args, device) / Object_Scoring(args, device) without models handed in: both networks are built from `args` and
    restored from args.objectness_resume / args.binary_classifier_resume ({'model_state_dict': ...}, strict), fp32, eval, frozen
    (object_reasoning.py:58-88, object_scoring.py:59-90)"""
    from unmore_amd.binary_classifier import Binary_Classifier
    from unmore_amd.hashrng import hash_init
    from unmore_amd.object_discovery import Object_Discovery
    from unmore_amd.object_scoring import Object_Scoring
    from unmore_amd.objectness_net import ObjectnessNet
    base = Namespace(use_bg_sdf=True, sdf_activation="tanh")
    net = ObjectnessNet("cpu", 128, "dpt_tiny", base)
    sd = {k: torch.from_numpy(hash_init(k, tuple(v.shape), "tiny")) for k, v in net.state_dict().items()}
    torch.save({"model_state_dict": sd, "iter": 7}, tmp_path / "obj.ckpt")
    clf = Binary_Classifier("cpu", 128, base)
    torch.save({"model_state_dict": clf.state_dict()}, tmp_path / "clf.ckpt")
    args = Namespace(use_bg_sdf=True, sdf_activation="tanh", image_size=128, backbone_type="dpt_tiny", objectness_resume=str(tmp_path / "obj.ckpt"),
                     binary_classifier_resume=str(tmp_path / "clf.ckpt"), n_round=2)
    for cls in (Object_Discovery, Object_Scoring):
        o = cls(args, DEV)
        assert isinstance(o.objectness_model, ObjectnessNet) and isinstance(o.binary_classifier_model, Binary_Classifier)
        assert not o.objectness_model.training and not any(p.requires_grad for p in o.objectness_model.parameters())
        got = o.objectness_model.state_dict()
        assert all(torch.equal(got[k].cpu(), sd[k]) for k in sd) and next(o.objectness_model.parameters()).is_cuda
    x = torch.rand(3, 3, 128, 128, device=DEV)
    out = o.objectness_model.get_prediction(x)
    assert out["sdf_maps"].shape == (3, 1, 128, 128) and out["center_fields"].dtype == torch.float32
    bad = Namespace(**{**vars(args), "backbone_type": "resnet"})
    with pytest.raises((NotImplementedError, KeyError)):
        Object_Discovery(bad, DEV)
