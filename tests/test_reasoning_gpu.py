"""f1/f2 rows: crop+resize, centre peak picking (bit-exact integer results) and boundary deltas vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import objectness_oracle as orc

pytestmark = pytest.mark.gpu


def _fields(B, H, W, seed):
    """smooth, object-like fields so the masks have structure (random noise would erode to nothing)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    sdf = torch.zeros(B, H, W)
    cen = torch.zeros(B, 2, H, W)
    for b in range(B):
        for _ in range(2):
            cy, cx = (0.25 + 0.5 * torch.rand(2, generator=g)) * torch.tensor([H, W])
            r = (0.15 + 0.2 * torch.rand(1, generator=g)) * min(H, W)
            d = torch.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
            sdf[b] = torch.maximum(sdf[b], torch.tanh((r - d) / 8))
            inside = d < r
            n = d + 1e-6
            sign = -1.0 if b % 2 == 0 else 1.0  # converging fields give positive anti-centre peaks, diverging ones none
            cen[b, 0] = torch.where(inside, sign * (yy - cy) / n, cen[b, 0])
            cen[b, 1] = torch.where(inside, sign * (xx - cx) / n, cen[b, 1])
        sdf[b] = sdf[b] * 2 - 0.3
    sdf += 0.05 * torch.randn(B, H, W, generator=g)
    cen += 0.05 * torch.randn(B, 2, H, W, generator=g)
    return sdf, cen


@pytest.mark.parametrize("B,H,W", [(5, 128, 128), (3, 96, 160)])
def test_center_peaks_bit_exact(B, H, W):
    from unmore_amd import reasoning
    sdf, cen = _fields(B, H, W, 0)
    score_r, max_r, arg_r = orc.peak_pick(sdf, cen)
    mx, am, sc = reasoning.center_peaks(sdf.cuda(), cen.cuda(), return_scores=True)
    mx, am, sc = mx.cpu(), am.cpu(), sc.cpu()
    # the integer side (masks, erosion, border) is exact: identical support of the score map
    assert torch.equal(sc != 0, score_r != 0)
    torch.testing.assert_close(sc, score_r, atol=1e-12, rtol=0)
    assert (max_r > 0).any(), "fixture has no peak: test would be vacuous"
    for b in range(B):
        if am[b] != arg_r[b]:  # only a genuine float64 tie may differ; report it
            gap = abs(score_r[b].flatten()[am[b]] - score_r[b].flatten()[arg_r[b]]).item()
            assert gap < 1e-13, f"map {b}: argmax {int(am[b])} vs {int(arg_r[b])} gap {gap}"
    assert torch.equal(am, arg_r)
    torch.testing.assert_close(mx, max_r, atol=1e-12, rtol=0)


def test_boundary_deltas():
    from unmore_amd import reasoning
    sdf, _ = _fields(6, 128, 128, 1)
    ref = orc.update_bbox_with_boundary_fields(sdf)
    out = reasoning.update_bbox_with_boundary_fields(sdf.cuda())
    for a, b in zip(out, ref):
        torch.testing.assert_close(a.cpu(), b, atol=1e-4, rtol=1e-4)


def test_crop_resize_matches_torchvision_semantics():
    from unmore_amd import reasoning
    g = torch.Generator().manual_seed(2)
    img = torch.rand(3, 480, 640, generator=g)
    boxes = torch.tensor([[0.0, 0.0, 640.0, 480.0], [10.3, 20.7, 200.2, 150.9], [300.0, 100.0, 340.5, 460.0], [600.2, 400.1, 640.0, 480.0],
                          [5.0, 5.0, 9.0, 8.0]])
    ref = orc.crop_resize(img, boxes, 128)
    out, on_edge = reasoning.crop_resize(img.cuda(), boxes, 128)
    torch.testing.assert_close(out.cpu(), ref, atol=2e-6, rtol=0)
    assert on_edge[0].tolist() == [True, True, True, True] and on_edge[1].tolist() == [False] * 4
    assert on_edge[3].tolist() == [False, False, True, True]


def test_pipeline_crops_to_peaks_fp32():
    """crop -> net (fp32) -> peaks on device equals the same chain with the oracle's post-processing on the SAME device
    fields (bit-exact integer side), on inputs whose score maps are not empty (structured image, the peak fixtures' weight
    edits -- see tests/peaks_common.py; the chain against the REFERENCE's own functions is tests/test_parity_r2_gpu.py)."""
    from argparse import Namespace
    import peaks_common as pc
    from unmore_amd import reasoning, synth
    from unmore_amd.objectness_net import ObjectnessNet
    sd = pc.edited_state_dict(orc.state_dict_spec(orc.CONFIGS["dpt_tiny"]), "tiny", 0.05, 2.0)
    net = ObjectnessNet("cuda:0", 128, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
    net.load_state_dict(sd)
    net = net.to("cuda:0").eval()
    img = torch.from_numpy(synth.blob_images(1, 240, 320, seed=3, n_blobs=5)[0]).cuda()
    boxes = torch.tensor([[0, 0, 320, 240], [40, 30, 200, 180], [100, 60, 300, 220.0], [10, 10, 150, 230]])
    crops, _ = reasoning.crop_resize(img, boxes, 128)
    with torch.no_grad():
        out = net.get_prediction(crops)
    mx, am = reasoning.center_peaks(out["sdf_maps"].squeeze(1), out["center_fields"])
    s_r, m_r, a_r = orc.peak_pick(out["sdf_maps"].squeeze(1).cpu(), out["center_fields"].cpu())
    assert (m_r > 0).sum() >= 2, "vacuous: no score map has a peak"
    assert torch.equal(am.cpu(), a_r)
    torch.testing.assert_close(mx.cpu(), m_r, atol=1e-12, rtol=0)


def test_sweep_over_two_streams_equals_sequential():
    """reasoning.sweep_proposals deals the independent 50-crop batches to two HIP streams: same peaks / deltas as one stream."""
    from argparse import Namespace
    import peaks_common as pc
    from unmore_amd import reasoning, synth
    from unmore_amd.objectness_net import ObjectnessNet
    sd = pc.edited_state_dict(orc.state_dict_spec(orc.CONFIGS["dpt_tiny"]), "tiny", 0.05, 2.0)
    net = ObjectnessNet("cuda:0", 128, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
    net.load_state_dict(sd)
    net = net.to("cuda:0").eval()
    img = torch.from_numpy(synth.blob_images(1, 240, 320, seed=4, n_blobs=6)[0]).cuda()
    g = torch.Generator().manual_seed(0)
    x1 = torch.rand(130, generator=g) * 200
    y1 = torch.rand(130, generator=g) * 140
    boxes = torch.stack([x1, y1, x1 + 40 + torch.rand(130, generator=g) * 80, y1 + 40 + torch.rand(130, generator=g) * 60], 1)
    a = reasoning.sweep_proposals(net, img, boxes, 50, n_streams=1)
    b = reasoning.sweep_proposals(net, img, boxes, 50, n_streams=2)
    torch.cuda.synchronize()
    assert (a[0] > 0).any()
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_sweep_on_three_streams_with_a_cold_pack_cache_equals_a_sequential_net():
    """The packed (kernel-layout) weight copies are built lazily by whichever stream touches a weight first.  With a COLD cache
    (freshly loaded net, first call ever) batch 0 packs on stream 0 while batches 1 and 2 hit the cache from streams 1 and 2:
    they must wait for the pack kernels (engine.PackCache records an event per entry).  Compared with a second, separately
    constructed net swept sequentially; the same again after the weights change in place (every entry is rebuilt)."""
    from argparse import Namespace
    import peaks_common as pc
    from unmore_amd import reasoning, synth
    from unmore_amd.objectness_net import ObjectnessNet
    sd = pc.edited_state_dict(orc.state_dict_spec(orc.CONFIGS["dpt_base"]), "base", 0.05, 2.0)
    img = torch.from_numpy(synth.blob_images(1, 240, 320, seed=4, n_blobs=6)[0]).cuda()
    g = torch.Generator().manual_seed(1)
    x1 = torch.rand(150, generator=g) * 200
    y1 = torch.rand(150, generator=g) * 140
    boxes = torch.stack([x1, y1, x1 + 40 + torch.rand(150, generator=g) * 80, y1 + 40 + torch.rand(150, generator=g) * 60], 1)

    def fresh():
        net = ObjectnessNet("cuda:0", 128, "dpt_base", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
        net.load_state_dict(sd)
        return net.to("cuda:0").eval()

    cold, seq = fresh(), fresh()
    torch.cuda.synchronize()
    b = reasoning.sweep_proposals(cold, img, boxes, 50, n_streams=3)     # very first call of this net: cold cache, 3 streams
    a = reasoning.sweep_proposals(seq, img, boxes, 50, n_streams=1)
    torch.cuda.synchronize()
    assert (a[0] > 0).any()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    # in-place weight change: every packed copy is stale and rebuilt, again first touched from three streams
    with torch.no_grad():
        for net in (cold, seq):
            for p in net.parameters():
                p.mul_(1.0009765625)
    b = reasoning.sweep_proposals(cold, img, boxes, 50, n_streams=3)
    a = reasoning.sweep_proposals(seq, img, boxes, 50, n_streams=1)
    torch.cuda.synchronize()
    for u, v in zip(a, b):
        assert torch.equal(u, v)
