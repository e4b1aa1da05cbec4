"""Host-side pieces of unmore_amd.object_discovery.Object_Discovery against what the reference's own methods returned
(tests/golden/discovery.npz, made by tests/golden/make_golden_r6_discovery.py from object_reasoning.py), and the oracle's NMS."""
import os

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "discovery.npz"))
SCENES = {"a": (240, 320, 0, 4), "b": (200, 288, 5, 6)}


def test_proposal_grid_equals_the_reference_grid():
    """object_reasoning.py:109-137, bit for bit: two scene sizes and the benchmark's 640 x 480 (1 225 boxes)"""
    from unmore_amd.object_discovery import Object_Discovery as OD
    for tag, (H, W, _, _) in SCENES.items():
        got = OD.generate_random_proposal(height=H, width=W)
        assert got.dtype == np.float64 and np.array_equal(got, G[f"{tag}_proposals0"])
    got = OD.generate_random_proposal(height=480, width=640)
    assert got.shape == (1225, 4) and np.array_equal(got, G["proposals_640x480"])


def test_box_update_and_enlargement_equal_the_reference():
    """post_process_bbox_update (:176-197) with float64 and float32 boxes (the first boundary round and the later ones), enlarge_proposals"""
    from unmore_amd.object_discovery import Object_Discovery as OD
    boxes, deltas = torch.from_numpy(G["ppbu_boxes"]), torch.from_numpy(G["ppbu_deltas"])
    o64 = OD.post_process_bbox_update(boxes, deltas)
    o32 = OD.post_process_bbox_update(boxes.to(torch.float32), deltas)
    assert o64.dtype == torch.float64 and np.array_equal(o64.numpy(), G["ppbu_out64"])
    assert o32.dtype == torch.float32 and np.array_equal(o32.numpy(), G["ppbu_out32"])
    got = OD.enlarge_proposals([[10, 20, 74, 52], [0, 0, 320, 240], [300, 200, 318, 238]], (240, 320), ratio=1.5)
    assert np.array_equal(np.array(got), G["enlarge_out"])
    assert OD.unravel_index(torch.tensor(130), (128, 128)) == (torch.tensor(1), torch.tensor(2))


def test_oracle_nms_known_answers():
    from oracle import objectness_oracle as orc
    b = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10], [21, 21, 29, 29]], np.float32)
    # equal scores: input order decides (box 0 suppresses 1 (IoU 0.68) and its duplicate 3; box 2 suppresses 4 (IoU 0.64))
    assert orc.nms(b, np.ones(5), 0.5).tolist() == [0, 2]
    # scores reorder the ranks: the highest-scored box of a cluster survives
    assert orc.nms(b, np.array([0.1, 0.9, 0.2, 0.3, 0.8]), 0.5).tolist() == [1, 4]
    assert orc.nms(b, np.ones(5), 0.7).tolist() == [0, 1, 2, 4]          # IoU 0.68 and 0.64 pass a 0.7 threshold, the duplicate does not
    assert orc.nms(b[:1], np.ones(1), 0.5).tolist() == [0]
