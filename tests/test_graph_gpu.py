"""HIP-graph replay of the engine's launch lists (unmore_amd/graphs.py): a replayed train step / inference call produces
bit-identical results to the eager launch list -- same kernels, same arguments, same order -- across changing inputs, a
learning-rate milestone, an in-place parameter change, a reloaded state dict, another input shape and several streams."""
from argparse import Namespace

import pytest
import torch

from unmore_amd import synth
from unmore_amd.hashrng import hash_init

pytestmark = pytest.mark.gpu
ARGS = Namespace(use_bg_sdf=True, sdf_activation="tanh")


def _net(backbone="dpt_tiny", tag="tiny", dtype=torch.float32, size=64):
    from unmore_amd.objectness_net import ObjectnessNet
    net = ObjectnessNet("cuda:0", size, backbone, ARGS)
    sd = {k: torch.from_numpy(hash_init(k, tuple(v.shape), tag)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    net = net.to("cuda:0")
    net.set_compute_dtype(dtype)
    return net, sd


def _batch(B, H, W, seed):
    return tuple(torch.from_numpy(a).cuda() for a in synth.make_batch(B, H, W, seed=seed))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_trainstep_replay_is_bit_identical_to_eager(dtype):
    from unmore_amd.trainer import TrainStep
    B, H, W = 2, 64, 64
    net_e, _ = _net(dtype=dtype)
    net_g, _ = _net(dtype=dtype)
    # a milestone inside the run: the learning rate and Adam's bias corrections reach the replay through device memory
    step_e = TrainStep(net_e, lr=1e-3, lr_milestones=(4,), lr_gamma=0.1).set_graph_mode("off")
    step_g = TrainStep(net_g, lr=1e-3, lr_milestones=(4,), lr_gamma=0.1).set_graph_mode("on")
    for it in range(7):
        batch = _batch(B, H, W, seed=10 + it)       # new tensors every step: the replay reads its own input buffers
        le = step_e.step(*batch)
        lg = step_g.step(*batch)
        assert torch.equal(le, lg), (it, le.tolist(), lg.tolist())
        assert torch.equal(step_e.flat_g, step_g.flat_g), it
        assert torch.equal(step_e.flat_p, step_g.flat_p), it
    assert step_g.graph_replays == 7 - 2 and step_e.graph_replays == 0     # two eager warm-up steps, then capture + replays
    assert torch.equal(step_e.m, step_g.m) and torch.equal(step_e.v, step_g.v)
    # another shape: eager while it warms up, results still equal; the first shape's capture is kept
    b2 = _batch(3, 64, 96, seed=99)
    assert torch.equal(step_e.step(*b2), step_g.step(*b2))
    b1 = _batch(B, H, W, seed=100)
    assert torch.equal(step_e.step(*b1), step_g.step(*b1))
    assert step_g.graph_replays == 6
    assert torch.equal(step_e.flat_p, step_g.flat_p)


def test_trainstep_recaptures_after_state_dict_reload():
    from unmore_amd.trainer import TrainStep
    net_e, sd = _net()
    net_g, _ = _net()
    step_e = TrainStep(net_e, lr=1e-3).set_graph_mode("off")
    step_g = TrainStep(net_g, lr=1e-3).set_graph_mode("on")
    batch = _batch(2, 64, 64, seed=1)
    for _ in range(4):
        assert torch.equal(step_e.step(*batch), step_g.step(*batch))
    sd2 = {k: v * 0.5 for k, v in sd.items()}
    for net, st in ((net_e, step_e), (net_g, step_g)):
        net.load_state_dict(sd2, strict=True)
        st.sync_from_model()
    n0 = step_g.graph_replays
    for _ in range(4):
        assert torch.equal(step_e.step(*batch), step_g.step(*batch))
    assert step_g.graph_replays == n0 + 2            # two eager steps (the packed weights were rebuilt), then a new capture
    assert torch.equal(step_e.flat_p, step_g.flat_p)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_inference_replay_is_bit_identical_and_follows_parameter_changes(dtype):
    net_e, _ = _net(dtype=dtype)
    net_g, _ = _net(dtype=dtype)
    net_e.set_graph_mode("off").eval()
    net_g.set_graph_mode("on").eval()
    outs = []
    with torch.no_grad():
        for it in range(5):
            x = _batch(3, 64, 64, seed=it)[0]
            oe, og = net_e.get_prediction(x), net_g.get_prediction(x)
            for k in ("center_fields", "sdf_maps"):
                assert torch.equal(oe[k], og[k]), (it, k)
            outs.append(og["sdf_maps"])
        assert not torch.equal(outs[3], outs[4])         # results are the caller's own tensors, not the capture's output buffer
        from unmore_amd import graphs
        caps = [v for v in net_g._inf_graphs.values() if isinstance(v, graphs.Captured)]
        assert len(caps) == 1 and caps[0].failed is None
        # an in-place parameter change: the capture read the packed copy of the old weight and must not be replayed
        for net in (net_e, net_g):
            net.sdf_prediction_head[1].weight.mul_(1.25)
            net.backbone.pretrained.model.blocks[0].mlp.fc1.weight.add_(0.01)
        x = _batch(3, 64, 64, seed=7)[0]
        oe, og = net_e.get_prediction(x), net_g.get_prediction(x)
        for k in ("center_fields", "sdf_maps"):
            assert torch.equal(oe[k], og[k]), k
        assert not torch.equal(og["sdf_maps"], outs[-1])


def test_sweep_on_three_streams_with_replays_equals_eager():
    """reasoning.sweep_proposals deals 50-crop batches to three streams: every stream replays its own capture (own scratch)."""
    import numpy as np
    from unmore_amd import reasoning
    net_e, _ = _net("dpt_tiny", "tiny", size=128)
    net_g, _ = _net("dpt_tiny", "tiny", size=128)
    net_e.set_graph_mode("off").eval()
    net_g.set_graph_mode("on").eval()
    image = torch.from_numpy(synth.blob_images(1, 240, 320, seed=3)[0]).cuda()
    rng = np.random.default_rng(0)
    x1, y1 = rng.integers(0, 200, 600), rng.integers(0, 140, 600)
    props = torch.from_numpy(np.stack([x1, y1, x1 + rng.integers(16, 120, 600), y1 + rng.integers(16, 100, 600)], 1).astype(np.float64))
    for rep in range(2):
        re = reasoning.sweep_proposals(net_e, image, props, 50, n_streams=3)
        rg = reasoning.sweep_proposals(net_g, image, props, 50, n_streams=3)
        for a, b in zip(re, rg):
            assert torch.equal(a, b), rep
    from unmore_amd import graphs
    caps = [v for v in net_g._inf_graphs.values() if isinstance(v, graphs.Captured)]
    assert len(caps) == 3 and all(c.failed is None for c in caps)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("graph", ["off", "on", "on-forked"])
def test_weight_gradient_stream_does_not_change_results(dtype, graph, monkeypatch):
    """engine.WgradStream: the weight-gradient GEMMs of a small problem run on a second stream beside the data-gradient chain --
    eagerly, as a chain of per-stage graphs replayed on two streams (graphs.StagedCaptured, the default capture of such a step), or
    as fork / join edges of one graph ('on-forked').  Same kernels on the same operands: bit-identical to the one-stream schedule,
    step after step."""
    from unmore_amd import engine, graphs
    from unmore_amd.trainer import TrainStep
    res = {}
    if graph == "on-forked":
        monkeypatch.setattr(graphs, "STAGED", False)
        graph = "on"
    for mode in ("0", "1"):
        monkeypatch.setattr(engine, "_WGRAD_STREAM", mode)
        net, _ = _net(dtype=dtype)
        step = TrainStep(net, lr=1e-3).set_graph_mode(graph)
        losses = []
        for it in range(5):
            losses.append(step.step(*_batch(2, 64, 96, seed=40 + it)))
        torch.cuda.synchronize()
        res[mode] = (torch.stack(losses), step.flat_p.clone(), step.flat_g.clone())
        if graph == "on":
            assert step.graph_replays == 3
            caps = [v for v in step._graphs.values() if isinstance(v, graphs.CAPTURE_TYPES)]
            assert len(caps) == 1 and caps[0].failed is None
            if mode == "1" and graphs.STAGED:
                lanes = [lane for lane, _ in caps[0].segments]
                # heads, refine, reassemble, 4 blocks, embed: a side graph per stage except the reassemble stage's update (main lane)
                assert isinstance(caps[0], graphs.StagedCaptured) and lanes.count("side") >= 7 and lanes[0] == "main", lanes
                # the join of the weight-gradient lane at the end of backward is part of the chain: what follows it on the main lane
                # (Adam for the stages updated there) reads gradients the side lane wrote
                assert lanes.count("join") == 1 and lanes.index("join") > max(i for i, l in enumerate(lanes) if l == "side"), lanes
                assert lanes[-1] == "main"
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)


def test_staged_capture_follows_reload_and_lr_milestones():
    """the chain-of-graphs train step (graphs.StagedCaptured) across a learning-rate milestone, a reloaded state dict (recaptured) and
    changing inputs: bit-identical to the eager one-stream step throughout"""
    from unmore_amd import graphs
    from unmore_amd.trainer import TrainStep
    net_e, sd = _net(dtype=torch.bfloat16)
    net_g, _ = _net(dtype=torch.bfloat16)
    step_e = TrainStep(net_e, lr=1e-3, lr_milestones=(3,), lr_gamma=0.1).set_graph_mode("off")
    step_g = TrainStep(net_g, lr=1e-3, lr_milestones=(3,), lr_gamma=0.1).set_graph_mode("on")
    import os
    os.environ["UMR_WGRAD_STREAM"] = "auto"
    for it in range(6):
        batch = _batch(2, 64, 96, seed=60 + it)
        assert torch.equal(step_e.step(*batch), step_g.step(*batch)), it
        assert torch.equal(step_e.flat_p, step_g.flat_p) and torch.equal(step_e.flat_g, step_g.flat_g), it
    assert step_g.graph_replays == 4
    assert any(isinstance(v, graphs.StagedCaptured) for v in step_g._graphs.values())
    sd2 = {k: v * 0.5 for k, v in sd.items()}
    for net, st in ((net_e, step_e), (net_g, step_g)):
        net.load_state_dict(sd2, strict=True)
        st.sync_from_model()
    for it in range(4):
        batch = _batch(2, 64, 96, seed=80 + it)
        assert torch.equal(step_e.step(*batch), step_g.step(*batch)), it
    assert step_g.graph_replays == 6 and torch.equal(step_e.flat_p, step_g.flat_p)
    assert torch.equal(step_e.m, step_g.m) and torch.equal(step_e.v, step_g.v)


def test_capture_scratch_is_per_stream_and_outgrown_buffers_stay_alive(monkeypatch):
    """ops._workspace / _splitk_workspace / the plane GEMMs' K-split scratch inside a capture: one buffer per STREAM of the capture (the
    weight-gradient branch of a train step runs beside the main branch in a replay -- sharing one scratch made
    test_weight_gradient_stream_does_not_change_results[on-float32] fail about one run in three), and a buffer that a later, larger
    request replaces is kept: launches recorded earlier still point at it."""
    from unmore_amd import graphs, ops
    dev = torch.device("cuda:0")
    store = {}
    monkeypatch.setitem(graphs._state, "capturing", True)
    monkeypatch.setitem(graphs._state, "store", store)
    side = torch.cuda.Stream(device=dev)
    a = ops._workspace(1 << 20, dev)
    assert ops._workspace(1 << 10, dev) is a                       # reused on the same stream
    sk = ops._splitk_workspace(dev)
    with torch.cuda.stream(side):
        b = ops._workspace(1 << 20, dev)
        skb = ops._splitk_workspace(dev)
    assert b.data_ptr() != a.data_ptr() and skb.data_ptr() != sk.data_ptr()
    big = ops._workspace(8 << 20, dev)                              # outgrows a: a new buffer, the old one stays referenced
    assert big.data_ptr() != a.data_ptr() and any(t is a for t in store["outgrown"])
    assert (sk[:16384] == 0).all() and (skb[:16384] == 0).all()     # tile counters start at zero on either stream


def test_failed_capture_leaves_no_unwritten_packs(monkeypatch):
    """A capture that raises has only RECORDED the packs it built (PackCache.get inside the capture): their buffers hold nothing.
    graphs.Captured's failure path purges them (PackCache.purge_capture), so the eager path that takes over re-packs instead of
    reading uninitialised copies under a matching version."""
    from unmore_amd import engine, graphs
    net, _ = _net()
    eng = net._engine()
    w = net.center_field_prediction_head[0].weight
    key = ("probe", "lin_t", torch.float32)
    x = _batch(2, 64, 64, seed=3)[0]

    def fn(xs):
        eng.cache.get(key, w, lambda: engine._pack_linear_t(w.detach().reshape(w.shape[0], -1), torch.float32))   # built INSIDE the capture
        assert key in eng.cache._c or key in eng.cache._o
        raise RuntimeError("forced failure after a pack")
    gen0 = eng.cache.generation()
    with pytest.warns(UserWarning, match="capture failed"):
        cap = graphs.Captured(fn, (x,), generation_of=eng.cache.generation, on_fail=eng.cache.purge_capture)
    assert cap.failed is not None and not cap.valid()
    assert key not in eng.cache._c and key not in eng.cache._o
    assert eng.cache.generation() != gen0                                  # other captures that read the set are invalidated
    good = eng.cache.get(key, w, lambda: engine._pack_linear_t(w.detach().reshape(w.shape[0], -1), torch.float32))
    torch.cuda.synchronize()
    assert torch.equal(good, w.detach().reshape(w.shape[0], -1).t())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_trainstep_survives_a_failed_capture(dtype, monkeypatch):
    """the capturing call of a train step fails late (Adam's launch raises under capture): the step falls back to the eager launch
    list in the same call and stays bit-identical to a step that never tried to capture"""
    from unmore_amd import graphs, ops
    from unmore_amd.trainer import TrainStep
    net_e, _ = _net(dtype=dtype)
    net_g, _ = _net(dtype=dtype)
    step_e = TrainStep(net_e, lr=1e-3).set_graph_mode("off")
    step_g = TrainStep(net_g, lr=1e-3).set_graph_mode("on")
    real = ops.adam_step_hyper

    def adam(*a, **k):
        if graphs.capturing():
            raise RuntimeError("forced failure at the end of the captured step")
        return real(*a, **k)
    monkeypatch.setattr(ops, "adam_step_hyper", adam)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(5):
            batch = _batch(2, 64, 64, seed=20 + it)
            assert torch.equal(step_e.step(*batch), step_g.step(*batch)), it
            assert torch.equal(step_e.flat_p, step_g.flat_p), it
    assert step_g.graph_replays == 0
    assert any(isinstance(v, graphs.CAPTURE_TYPES) and v.failed for v in step_g._graphs.values())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_staged_capture_at_the_reference_recipe_shape(dtype):
    """where the chain of per-stage graphs is the default: the reference's own recipe (dpt_large, 20 crops of 128 x 128,
    train_objectness_net.py:783-788,815-817) -- 24 transformer blocks = 28 stages, split-K GEMMs and their per-lane scratch, the
    fp32 mode's plane kernels with their K-split workspaces on both lanes.  Default graph mode ('auto') against eager, five steps,
    bit for bit."""
    from unmore_amd import graphs
    from unmore_amd.trainer import TrainStep
    net_e, _ = _net("dpt_large", "large", dtype=dtype, size=128)
    net_g, _ = _net("dpt_large", "large", dtype=dtype, size=128)
    step_e = TrainStep(net_e, lr=1e-4).set_graph_mode("off")
    step_g = TrainStep(net_g, lr=1e-4)                      # 'auto'
    assert step_g.graph_mode == graphs.DEFAULT_MODE
    for it in range(5):
        batch = _batch(20, 128, 128, seed=300 + it)
        le, lg = step_e.step(*batch), step_g.step(*batch)
        assert torch.equal(le, lg), (it, le.tolist(), lg.tolist())
    torch.cuda.synchronize()
    assert torch.equal(step_e.flat_p, step_g.flat_p) and torch.equal(step_e.m, step_g.m) and torch.equal(step_e.v, step_g.v)
    if graphs.DEFAULT_MODE == "auto":
        caps = [v for v in step_g._graphs.values() if isinstance(v, graphs.StagedCaptured)]
        assert len(caps) == 1 and caps[0].failed is None and step_g.graph_replays == 3
        lanes = [lane for lane, _ in caps[0].segments]
        assert lanes.count("side") >= 27 and lanes.count("main") >= 27 and lanes.count("join") == 1, (lanes.count("main"), lanes.count("side"))


@pytest.mark.parametrize("eval_graphs", ["off", "on"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_evaluation_between_replayed_train_steps_sees_the_new_weights(dtype, eval_graphs):
    """The reference's loop evaluates between training steps (train_objectness_net.py:356, under torch.no_grad()).  An evaluation call
    caches derived weights (the collapsed boundary-distance head's tap matrix, engine._linear_head_weights_cached) and may itself be
    replayed from a graph; a REPLAYED train step updates the parameters on the device without touching their torch version counters.
    Every evaluation must see the weights of the step before it: compared, call by call, with a twin trained eagerly."""
    from unmore_amd.trainer import TrainStep
    net_e, _ = _net(dtype=dtype)
    net_g, _ = _net(dtype=dtype)
    step_e = TrainStep(net_e, lr=2e-4).set_graph_mode("off")
    step_g = TrainStep(net_g, lr=2e-4).set_graph_mode("on")
    net_e.set_graph_mode("off")
    net_g.set_graph_mode(eval_graphs)      # 'off': the evaluation's cached packs are built outside any capture (PackCache.refreshed_by_replay drops them)
    xe = _batch(3, 64, 64, seed=7)[0]
    prev = None
    for it in range(8):
        batch = _batch(2, 64, 64, seed=700 + it)
        assert torch.equal(step_e.step(*batch), step_g.step(*batch)), it
        with torch.no_grad():
            oe, og = net_e.get_prediction(xe), net_g.get_prediction(xe)
        for k in ("center_fields", "sdf_maps"):
            assert torch.equal(oe[k], og[k]), (it, k)
        if prev is not None:                                       # the weights moved: so did both maps
            assert not torch.equal(prev[0], og["sdf_maps"]) and not torch.equal(prev[1], og["center_fields"]), it
        prev = (og["sdf_maps"].clone(), og["center_fields"].clone())
    assert step_g.graph_replays == 6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_second_evaluation_shape_after_replayed_train_steps_sees_the_new_weights(dtype):
    """Round-5 advisor: once train steps replay, an evaluation capture builds the collapsed head's weights INSIDE itself (the copies
    of its warm-up calls were dropped by the replays), and that entry used to stay in the host-side cache under an unchanged version
    key.  An eager evaluation of ANOTHER shape right after the next replayed step -- and, on its third call, a second capture -- then
    hit it and read the weights the first graph's last replay had written: one step old.  Every evaluation of both shapes is compared
    with a twin that trains and evaluates eagerly; the second shape is evaluated FIRST after each step (before the first graph's replay
    rewrites its buffer)."""
    from unmore_amd.trainer import TrainStep
    net_e, _ = _net(dtype=dtype)
    net_g, _ = _net(dtype=dtype)
    step_e = TrainStep(net_e, lr=2e-4).set_graph_mode("off")
    step_g = TrainStep(net_g, lr=2e-4).set_graph_mode("on")
    net_e.set_graph_mode("off")
    net_g.set_graph_mode("on")
    xa, xb = _batch(3, 64, 64, seed=7)[0], _batch(2, 64, 96, seed=8)[0]
    for it in range(11):
        batch = _batch(2, 64, 64, seed=900 + it)
        assert torch.equal(step_e.step(*batch), step_g.step(*batch)), it
        with torch.no_grad():
            for tag, x in (("b", xb), ("a", xa)) if it >= 5 else (("a", xa),):
                oe, og = net_e.get_prediction(x), net_g.get_prediction(x)
                for k in ("center_fields", "sdf_maps"):
                    assert torch.equal(oe[k], og[k]), (it, tag, k)
    assert step_g.graph_replays >= 8
    from unmore_amd import graphs
    assert sum(isinstance(v, graphs.Captured) for v in net_g._inf_graphs.values()) >= 1


def test_dropped_captures_release_their_pools():
    """A loop whose batch size keeps changing (the reference's batch filter, train_objectness_net.py:190-207) captures every shape on
    its third step and holds at most graphs.MAX_CAPTURES captures; every capture owns private memory pools, and a new capture never
    draws from the allocator's cache -- dropped captures must hand their pools back (graphs.release_dropped), or reserved memory
    grows by a pool per capture (measured before the fix: +33 GiB per 14 shapes of dpt_base 128^2)."""
    from unmore_amd import graphs
    from unmore_amd.trainer import TrainStep
    net, _ = _net(dtype=torch.bfloat16)
    step = TrainStep(net, lr=1e-4).set_graph_mode("on")
    reserved = []
    for rnd in range(3):
        for B in range(2, 2 + graphs.MAX_CAPTURES + 4):           # more shapes than captures are held
            batch = _batch(B, 64, 64, seed=B)
            for _ in range(4):
                step.step(*batch)
        torch.cuda.synchronize()
        held = sum(isinstance(v, graphs.CAPTURE_TYPES) for v in step._graphs.values())
        assert 1 <= held <= graphs.MAX_CAPTURES
        reserved.append(torch.cuda.memory_reserved())
    assert step.graph_replays == 3 * (graphs.MAX_CAPTURES + 4) * 2
    assert reserved[2] <= 1.25 * reserved[0] + (64 << 20), [r / 2 ** 20 for r in reserved]


@pytest.mark.parametrize("batched_repack", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_staged_replay_with_a_lagging_side_lane(dtype, batched_repack, monkeypatch):
    """Dependencies of the main lane on the side lane must be in the chain, not in the timing: every side graph of the replay is held back
    by ~1 ms of spinning (StagedCaptured.debug_side_delay), so the weight gradients, the per-stage Adam updates and weight refreshes
    finish long after the main lane has moved on.  The step must stay bit-identical to the eager one -- with the whole Adam update on
    the main lane (lazy re-pack, batched_repack=False: it reads every gradient the side lane writes) and with the default schedule."""
    from unmore_amd import graphs, trainer
    monkeypatch.setattr(trainer, "_BATCHED_REPACK", batched_repack)
    net_e, _ = _net(dtype=dtype)
    net_g, _ = _net(dtype=dtype)
    step_e = trainer.TrainStep(net_e, lr=1e-3).set_graph_mode("off")
    step_g = trainer.TrainStep(net_g, lr=1e-3).set_graph_mode("on")
    for it in range(7):
        batch = _batch(2, 64, 64, seed=900 + it)
        le, lg = step_e.step(*batch), step_g.step(*batch)
        assert torch.equal(le, lg), (it, le.tolist(), lg.tolist())
        assert torch.equal(step_e.flat_p, step_g.flat_p), it
        for cap in step_g._graphs.values():
            if isinstance(cap, graphs.StagedCaptured):
                cap.debug_side_delay = 2_500_000
    assert step_g.graph_replays == 5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_staged_replay_with_a_lagging_main_lane(dtype):
    """The mirror image: the MAIN lane is held back ~1 ms before every segment (StagedCaptured.debug_main_delay), so side graph B_k --
    stage k's weight gradients, its Adam update and the refresh of its packed weight copies -- has long finished when the main lane
    runs A_k+1, A_k+2, ...  Any later main-lane kernel that still read stage k's weights (the hazard trainer.py keeps the reassemble
    stage off the side lane for: the readout projections are read again when the transformer's backward reaches a hooked block,
    models/dpt/vit.py:86-90) would read UPDATED weights and the step would differ from the eager one.  dpt_tiny has all four hooks
    (blocks 0-3) and the two-input fusion blocks, i.e. every such re-read the wiring has."""
    from unmore_amd import graphs, trainer
    net_e, _ = _net(dtype=dtype)
    net_g, _ = _net(dtype=dtype)
    step_e = trainer.TrainStep(net_e, lr=1e-3).set_graph_mode("off")
    step_g = trainer.TrainStep(net_g, lr=1e-3).set_graph_mode("on")
    for it in range(7):
        batch = _batch(2, 64, 64, seed=950 + it)
        le, lg = step_e.step(*batch), step_g.step(*batch)
        assert torch.equal(le, lg), (it, le.tolist(), lg.tolist())
        assert torch.equal(step_e.flat_p, step_g.flat_p), it
        for cap in step_g._graphs.values():
            if isinstance(cap, graphs.StagedCaptured):
                cap.debug_main_delay = 2_500_000
    assert step_g.graph_replays == 5
