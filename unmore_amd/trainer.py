"""Native training step for the stage-1 ObjectnessNet (reference loop:
train_objectness_net.py:181-261): forward, fused 4-term loss, backward, gradient
all-reduce (data parallel, one process per GPU over RCCL), Adam and the per-iteration
MultiStepLR schedule -- all on the HIP kernels, no autograd graph, one host sync at most
(only if the caller reads the loss).

Parameters that receive gradients are re-homed into ONE flat fp32 buffer laid out in
backward-completion order (heads, refinenets, reassemble, blocks last..first, embeddings)
so that (a) the gradient buffer is exchanged in a few large contiguous buckets that
overlap the remaining backward, and (b) Adam is a single kernel launch over the flat
buffers.  `model.state_dict()` is unaffected (parameters are views)."""
import torch

import os

from . import graphs, ops
from .parallel import BucketedAllReduce

_BATCHED_REPACK = os.environ.get("UMR_BATCHED_REPACK", "1") != "0"   # A/B switch: 0 = drop the packed copies, re-pack lazily (round 2)
# A/B switch: 0 = per stage, the Adam launch and then the batched refresh of its packed copies (round 5); 1 = the Adam launch writes the
# bf16 copies of the stage's Linear weights itself (round 6: ops.adam_pack; bit-identical)
_DP_ADAM_LAG = max(1, int(os.environ.get("UMR_DP_ADAM_LAG", "2")))   # stages between a bucket's all-reduce and its optimizer update
_ADAM_PACK = os.environ.get("UMR_ADAM_PACK", "1") != "0"


def _stage_of(name, cfg):
    if name.startswith("center_field_prediction_head") or name.startswith("sdf_prediction_head"):
        return "heads"
    if name.startswith("backbone.scratch.refinenet"):
        return "refine"
    if name.startswith("backbone.scratch.layer") or name.startswith("backbone.pretrained.act_postprocess"):
        return "reassemble"
    m = "backbone.pretrained.model.blocks."
    if name.startswith(m):
        return "block" + name[len(m):].split(".")[0]
    return "embed"


def flat_layout(net):
    """The flat gradient / parameter buffer of a net: {name: element offset}, bucket boundaries [0, ..., total] and
    {stage: bucket index}, in backward-completion order (heads, refinenets, reassemble, blocks last..first, embeddings); every
    view 256-byte aligned.  Device-independent (bench.py --rehearse builds the same 16 buckets on the CPU)."""
    cfg = net.cfg
    named = dict(net.named_parameters())
    nograd = net.nograd_names()
    order = ["heads", "refine", "reassemble"] + [f"block{i}" for i in range(max(cfg["hooks"]), -1, -1)] + ["embed"]
    by_stage = {s: [] for s in order}
    for n in named:
        if n not in nograd:
            by_stage[_stage_of(n, cfg)].append(n)
    offs, bounds, off, stage_bucket = {}, [0], 0, {}
    for bi, s in enumerate(order):
        for n in by_stage[s]:
            offs[n] = off
            off += (named[n].numel() + 63) // 64 * 64
        bounds.append(off)
        stage_bucket[s] = bi
    return offs, bounds, stage_bucket


def filter_batch(images, gt_center_fields, gt_sdf_maps, gt_saliency_maps):
    """The reference's per-step batch filter (train_objectness_net.py:190-207): drop images whose pseudo-mask is all
    background or all foreground, so every remaining image has both.  Host-side boolean indexing (tensor plumbing,
    one device sync for the data-dependent batch size), exactly as the reference does it."""
    s = gt_saliency_maps.reshape(gt_saliency_maps.shape[0], -1)
    keep = (s.sum(1) > 0) & ((1 - s).sum(1) != 0)
    return images[keep], gt_center_fields[keep], gt_sdf_maps[keep], gt_saliency_maps[keep]


class TrainStep:
    def __init__(self, net, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, center_field_loss_type="l2", sdf_loss_type="l1",
                 use_sdf_gradient_loss=True, use_sdf_binary_mask_loss=True, lr_milestones=(), lr_gamma=1.0, group=None,
                 grad_wire_dtype=None):
        """grad_wire_dtype (data-parallel runs only): None / env UMR_DP_WIRE unset = the f32 gradient buckets are all-reduced as they
        are; torch.bfloat16 / UMR_DP_WIRE=bf16 = bf16 copies are exchanged (parallel.BucketedAllReduce: half the bytes per link)."""
        self.net = net
        self.lr0, self.betas, self.eps = lr, betas, eps
        self.milestones, self.gamma = tuple(lr_milestones), lr_gamma
        self.loss_cfg = (center_field_loss_type == "l2", sdf_loss_type == "l2", bool(use_sdf_gradient_loss),
                         bool(use_sdf_binary_mask_loss))
        self.iter = 0
        named = dict(net.named_parameters())
        dev = next(iter(named.values())).device
        assert dev.type == "cuda", "TrainStep needs the model on the GPU"
        offs, bounds, self.stage_bucket = flat_layout(net)
        off = bounds[-1]
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.G = {}
        with torch.no_grad():
            for n, o in offs.items():
                p = named[n]
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self.G[n] = self.flat_g[o:o + p.numel()].view(p.shape)
        self.P = {n: p for n, p in net.named_parameters()}
        self._offs = offs
        self._stage_params = {}     # stage -> [(name, offset, numel, shape)] in buffer order (PackCache.adam_and_refresh)
        for n, o in sorted(offs.items(), key=lambda t: t[1]):
            self._stage_params.setdefault(_stage_of(n, net.cfg), []).append((n, o, named[n].numel(), tuple(named[n].shape)))
        if grad_wire_dtype is None and os.environ.get("UMR_DP_WIRE", "") == "bf16":
            grad_wire_dtype = torch.bfloat16
        self.comm = BucketedAllReduce(self.flat_g, bounds, group, wire_dtype=grad_wire_dtype)
        self._hyper = torch.zeros(8, dtype=torch.float32, device=dev)   # Adam's per-step scalars (ops.adam_set_hyper)
        self.graph_mode = graphs.DEFAULT_MODE
        self._graphs = {}           # (input shape, dtype, f32 mode, stream) -> eager-call count, then graphs.Captured
        self.graph_replays = 0
        self.poisoned = None        # set when a step raised in mid-flight (step()); cleared by load_optimizer_state_dict
        net._engine().cache.clear()

    def current_lr(self):
        lr = self.lr0
        for ms in self.milestones:
            if self.iter >= ms:
                lr *= self.gamma
        return lr

    def _body(self, images, gt_center_fields, gt_sdf_maps, gt_saliency_maps):
        """forward, loss, backward, gradient exchange, Adam (scalars from self._hyper), weight-copy refresh: the launch list of one
        step.  Runs eagerly or under a HIP-graph capture (graphs.py)."""
        eng = self.net._engine()
        center, sdf, S = eng.forward(self.P, images, save=True)
        out5, dpc, dps = ops.objectness_loss(center, sdf, gt_center_fields, gt_sdf_maps, gt_saliency_maps, *self.loss_cfg)
        bounds = self.comm.bounds
        updated = set()

        def update(k):
            lo, hi = bounds[k], bounds[k + 1]
            if hi > lo:
                ops.adam_step_hyper(self.flat_p[lo:hi], self.flat_g[lo:hi], self.m[lo:hi], self.v[lo:hi], self._hyper)

        def update_and_refresh(stage, k):
            """stage k's optimizer step and the refresh of its packed weight copies: one launch that writes the Linear weights' bf16
            copies itself where the stage has such copies (bf16 mode), the two-launch form otherwise"""
            lo, hi = bounds[k], bounds[k + 1]
            sel = lambda key, stage=stage: _stage_of(key[0], self.net.cfg) == stage
            if _ADAM_PACK and hi > lo and eng.cache.adam_and_refresh(stage, sel, self._stage_params.get(stage, []), lo, hi,
                                                                     (self.flat_p, self.flat_g, self.m, self.v), self._hyper):
                return
            update(k)
            eng.cache.refresh(tag=stage, select=sel)

        # a staged capture in progress (graphs.StagedCaptured): collectives are not captured -- the chain is cut at the bucket
        # boundaries it already has, and a replay issues each bucket's all-reduce between two graph launches (on the side lane,
        # behind the stage's weight gradients; the main lane waits for the exchange only in finish(), before the optimizer update)
        staged = graphs.staged() if self.comm.enabled else None

        lagq = []      # data-parallel small problems: stages whose all-reduce is in flight; updated _DP_ADAM_LAG stages later

        def stage_done(stage, wg):
            k = self.stage_bucket[stage]
            if staged is not None:
                staged.defer_host(lambda k=k: self.comm.ready(k))
            else:
                self.comm.ready(k)
            if not (wg.on and _BATCHED_REPACK):
                return
            # (not the reassemble stage: the readout projections are read again when the transformer's backward reaches a hooked
            # block -- the token gradient goes through them, models/dpt/vit.py:86-90 -- so they are updated after backward)
            if not self.comm.enabled:
                if stage != "reassemble":
                    # small problems: this stage's Adam update and the refresh of its packed weight copies go behind its weight
                    # gradients on the second stream -- HBM-bound work beside the (latency-bound) rest of backward.  Nothing later in
                    # this step reads the stage's weights again; the final join of backward orders the next step after them.
                    def launch():
                        update_and_refresh(stage, k)
                    wg.run(launch)
                    updated.add(k)
                return
            # data-parallel: the same, _DP_ADAM_LAG stages LATER -- once the bucket's all-reduce, issued above, has had that many
            # stages of backward to complete, the weight-gradient lane waits for it (BucketedAllReduce.wait: a stream dependency, the
            # host does not block) and updates the stage.  Without this the whole optimizer pass (9.7 GB of traffic for ViT-L) sits
            # exposed behind finish(): +2.1 ms on the reference recipe's 18.5-ms step (profiles/r06_rccl_one_rank.jsonl).  The last
            # _DP_ADAM_LAG stages and the reassemble stage are updated after finish(), on the main lane.
            if stage != "reassemble":
                lagq.append((stage, k))
            if len(lagq) > _DP_ADAM_LAG:
                ps, pk = lagq.pop(0)
                if staged is not None:
                    staged.defer_host(lambda pk=pk: self.comm.wait(pk))
                    wg.run(lambda ps=ps, pk=pk: update_and_refresh(ps, pk))
                else:                              # eager (the warm-up steps before the capture, --graphs off): on the main stream,
                    self.comm.wait(pk)             # which joins the weight-gradient stream at every stage in this form
                    update_and_refresh(ps, pk)
                updated.add(pk)

        eng.backward(self.P, S, dpc, dps, self.G, stage_cb=stage_done, join_at_stages=self.comm.enabled and staged is None)
        if staged is not None:
            staged.host_call(self.comm.finish)
        else:
            self.comm.finish()
        if not updated:
            ops.adam_step_hyper(self.flat_p, self.flat_g, self.m, self.v, self._hyper)
            # the packed (kernel-layout) weight copies are stale after the in-place update: refreshed in one launch
            if _BATCHED_REPACK:
                eng.cache.refresh()
            else:
                eng.cache.clear()
        else:
            for stage, k in self.stage_bucket.items():
                if k not in updated:
                    update_and_refresh(stage, k)
            eng.cache.refresh_done()
            if not graphs.capturing():
                eng.cache.synced_with(torch.cuda.current_stream(images.device))
        return (out5,)

    def set_graph_mode(self, mode):
        """'auto' (default; env UMR_GRAPHS): steps of small problems (B*H*W <= 2^20 pixels: the reference's recipe) are captured after
        two eager steps of the same shape as a chain of per-stage HIP graphs and replayed on two streams (graphs.StagedCaptured:
        bit-identical to the eager step, host enqueue 18.7 -> 1.0 ms, 873 -> 904 images/s on the reference recipe); large ones run
        eagerly (GPU-bound, < 3 % of the step is host time).  'on' captures any size, 'off' nothing.  Data-parallel runs
        (world > 1) replay the same chain where the chain form applies (small problems): the gradient all-reduces are not captured
        but issued between the chain's graph launches at the bucket boundaries it is cut at (round 6); larger data-parallel steps
        run eagerly."""
        assert mode in ("auto", "on", "off")
        self.graph_mode = mode
        self._graphs.clear()
        return self

    def step(self, images, gt_center_fields, gt_sdf_maps, gt_saliency_maps):
        """One optimisation step; returns the [total, center, sdf, grad, bce] loss tensor (device, f32)."""
        if images.shape[0] == 0:
            # the reference would take the mean of empty maps (NaN) and push NaN gradients into Adam; refuse instead
            raise ValueError("TrainStep.step: empty batch (every image was filtered out); skip this iteration")
        # the step count and the schedule advance only once the step's launches are enqueued: a body that raises leaves `iter` (and
        # with it the checkpoint's 'iter' and the learning-rate schedule) where the last COMPLETE update put it -- and poisons this
        # object, because finished stages of the failed step may already have been updated (a failed CAPTURE does not raise: it
        # falls back to the eager body within the same call)
        if self.poisoned is not None:
            raise RuntimeError("TrainStep: an earlier step raised after part of its optimizer update was enqueued (" + self.poisoned +
                               "); weights and Adam state no longer match `iter` -- reload a checkpoint (model.load_state_dict, "
                               "load_optimizer_state_dict, sync_from_model) before the next step")
        it = self.iter + 1
        ops.adam_set_hyper(self._hyper, it, self.lr_of_step(it), self.betas[0], self.betas[1], self.eps, self.comm.grad_scale)
        try:
            out5 = self._launch_step(images, gt_center_fields, gt_sdf_maps, gt_saliency_maps)
        except BaseException as e:
            # an eager two-stream body that raises in mid-backward has already enqueued the Adam updates of the stages it finished
            # (with step index `it`): a retried step would apply that index to those stages a second time.  Refuse to continue.
            self.poisoned = f"{type(e).__name__}: {e}"
            raise
        self.iter = it
        return out5

    def _launch_step(self, images, gt_center_fields, gt_sdf_maps, gt_saliency_maps):
        ins = (images, gt_center_fields, gt_sdf_maps, gt_saliency_maps)
        eng = self.net._engine()
        B, _, H, W = images.shape
        from .engine import WgradStream
        two = WgradStream.wanted(B * H * W)
        # data-parallel steps replay only as a chain of per-stage graphs (its cuts are where the collectives go); a single graph cannot
        # hold them and such a step stays eager
        if ((not self.comm.enabled or (graphs.STAGED and two)) and graphs.wanted(self.graph_mode, B * H * W, train=True, two_streams=two)
                and ops._timer["select"] is None):
            key = (tuple(images.shape), eng.dt, ops.get_f32_mode(), torch.cuda.current_stream(images.device).cuda_stream)
            ent = self._graphs.get(key)
            if isinstance(ent, graphs.CAPTURE_TYPES):
                if ent.valid():
                    (out5,) = ent.replay(*ins)
                    eng.cache.refreshed_by_replay(images.device)
                    self.graph_replays += 1
                    return out5.clone()
                if ent.failed is None:
                    ent = None            # the packed weights moved (state dict reloaded): warm up and capture again
            if not isinstance(ent, graphs.CAPTURE_TYPES):
                n = (ent or 0) + 1
                self._graphs[key] = n
                if n > graphs.WARMUP_CALLS:
                    # the warm-up steps have packed every weight, built the batched refresh and sized the workspaces
                    # small problems (the two-stream regime, engine.WgradStream): a chain of per-stage graphs on two streams; large
                    # ones: one graph.  Every capture keeps its own pools of temporaries: a loop whose batch size keeps changing (the
                    # reference's batch filter, train_objectness_net.py:190-207) holds at most MAX_CAPTURES of them
                    if sum(isinstance(v, graphs.CAPTURE_TYPES) for v in self._graphs.values()) >= graphs.MAX_CAPTURES:
                        for k_ in [k_ for k_, v in self._graphs.items() if isinstance(v, graphs.CAPTURE_TYPES)]:
                            del self._graphs[k_]
                        graphs.release_dropped()
                    kind = graphs.StagedCaptured if (graphs.STAGED and two) else graphs.Captured
                    cap = kind(self._body, ins, generation_of=eng.cache.generation, on_fail=eng.cache.purge_capture)
                    self._graphs[key] = cap
                    if cap.failed is None:
                        # the capture only RECORDED the step: run it
                        (out5,) = cap.replay(*ins)
                        eng.cache.refreshed_by_replay(images.device)
                        self.graph_replays += 1
                        return out5.clone()
        return self._body(*ins)[0]

    # ---- checkpoint / resume (train_objectness_net.py:118-123,268-275: {'model_state_dict', 'optimizer_state_dict', 'iter'})
    def optimizer_state_dict(self):
        """The optimizer half of the reference's checkpoint in torch.optim.Adam's own format (parameter index = position in
        model.parameters(); parameters that never receive a gradient have no state entry, as in torch), so a checkpoint
        written here resumes in the reference loop and vice versa."""
        names = [n for n, _ in self.net.named_parameters()]
        state = {}
        if self.iter > 0:
            for i, n in enumerate(names):
                if n in self._offs:
                    o, p = self._offs[n], self.P[n]
                    state[i] = {"step": torch.tensor(float(self.iter)),
                                "exp_avg": self.m[o:o + p.numel()].view(p.shape).clone(),
                                "exp_avg_sq": self.v[o:o + p.numel()].view(p.shape).clone()}
        group = {"lr": self.current_lr(), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "initial_lr": self.lr0, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd, iteration=None):
        """Restore Adam's moments (and the step count) from `optimizer_state_dict()` / a torch.optim.Adam state dict of the
        same model.  iteration: the checkpoint's 'iter' (defaults to the stored Adam step); the learning-rate schedule here
        is a function of that absolute iteration."""
        names = [n for n, _ in self.net.named_parameters()]
        assert len(sd["param_groups"]) == 1 and len(sd["param_groups"][0]["params"]) == len(names), "optimizer state of another model"
        self.m.zero_()
        self.v.zero_()
        step = 0
        for i, st in sd["state"].items():
            n = names[int(i)]
            if n not in self._offs:
                continue
            o, p = self._offs[n], self.P[n]
            self.m[o:o + p.numel()].view(p.shape).copy_(st["exp_avg"])
            self.v[o:o + p.numel()].view(p.shape).copy_(st["exp_avg_sq"])
            step = max(step, int(float(st["step"])))
        self.iter = int(iteration) if iteration is not None else step
        assert self.iter == step or not sd["state"], "Adam step count and checkpoint iteration disagree"
        self.poisoned = None

    def sync_from_model(self):
        """Call after model.load_state_dict(): the flat parameter buffer is the storage of the parameters, so loading writes
        through; only the packed kernel-layout weight copies have to be dropped (captured steps notice through the cache's
        generation and are captured again after two eager steps)."""
        self.net._engine().cache.clear()

    def lr_of_step(self, k):
        # torch's MultiStepLR.step() runs after optimizer.step(): step k (1-based) uses the lr of k-1 completed steps
        lr = self.lr0
        for ms in self.milestones:
            if k - 1 >= ms:
                lr *= self.gamma
        return lr

    def current_lr_for_step(self):
        return self.lr_of_step(self.iter)
