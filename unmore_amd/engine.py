"""Hand-scheduled forward / backward of the ObjectnessNet hot path on the HIP kernels.

This is the host-side "graph": an explicit list of kernel launches (no tracing
compiler, no autograd inside).  Activations are NHWC / [rows, channels]; weights
are repacked from the reference's PyTorch layouts into the layouts the kernels want
(cached per parameter version).  Reference op order and formulas:
models/dpt/vit.py:165-201 (forward_flex), :86-90 (ProjectReadout), :104-145 +
:259-336 (reassemble), models/dpt/blocks.py:290-313 (RCU), :362-383 (fusion),
models/dpt/models.py:74-94 (DPT.forward), models/objectness_net.py:109-135,167-183
(heads), timm Block semantics as restated in SURVEY.md section 8c.
"""
import os

import torch

from . import _lib as L
from . import graphs
from . import ops
from .engine_x3 import X3Path

CONFIGS = {
    # reference wiring: models/dpt/models.py:43-47, blocks.py:24-54, vit.py:515-543
    "dpt_large": dict(D=1024, depth=24, heads=16, patch=16, pos_grid=24, hooks=[5, 11, 17, 23], features=[256, 512, 1024, 1024]),
    "dpt_base": dict(D=768, depth=12, heads=12, patch=16, pos_grid=24, hooks=[2, 5, 8, 11], features=[96, 192, 384, 768]),
    # extensions named by BASELINE.json
    "dpt_small": dict(D=384, depth=12, heads=6, patch=16, pos_grid=24, hooks=[2, 5, 8, 11], features=[48, 96, 192, 384]),
    "dpt_large14": dict(D=1024, depth=24, heads=16, patch=14, pos_grid=37, hooks=[5, 11, 17, 23], features=[256, 512, 1024, 1024]),
    "dpt_tiny": dict(D=128, depth=4, heads=2, patch=16, pos_grid=24, hooks=[0, 1, 2, 3], features=[32, 64, 128, 128]),
}

_ACT = {None: L.ACT_NONE, "tanh": L.ACT_TANH, "sine": ops.ACT_SINE}


class ParamGroup:
    """Several parameters seen by PackCache.get as one: the key of a pack derived from all of them (the collapsed head's weight
    algebra reads eight tensors) -- its version is the tuple of theirs, so a change of any one misses."""

    def __init__(self, params):
        self.ps = tuple(params)
        self.is_cuda, self.device = self.ps[0].is_cuda, self.ps[0].device

    @property
    def _version(self):
        return tuple(p._version for p in self.ps)

    def data_ptr(self):
        return tuple(p.data_ptr() for p in self.ps)


class PackCache:
    """Kernel-layout copies of parameters, rebuilt when the parameter changes.

    Stream-safe: an entry is built by kernels enqueued on whatever stream touches it first and is published to this host-side
    dict at once, so a consumer on ANOTHER stream (reasoning.sweep_proposals deals batches to several streams) could launch
    a GEMM that reads the packed buffer before the pack kernel has run.  Every entry therefore carries an event recorded
    right after its build; a hit from a stream that has not yet ordered itself after that event waits on it first (once per
    stream and entry -- afterwards the stream's own order covers it)."""

    def __init__(self):
        self._c = {}           # key -> [version, value, event, synced streams, recipe, param]: replayable packs
        self._o = {}           # same without a recipe: rebuilt lazily after refresh()
        self._replay = {}      # tag -> (launcher, keys) of a batched refresh (None: all entries), built on first use
        # A captured graph holds the ADDRESSES of the copies it read: gen_c counts changes of the replayable set (and of the batched
        # refresh's table), gen_o changes of the rest.  A capture depends on gen_o only if it read such an entry that it did not
        # build itself (a capture rebuilds its own on every replay): generation(store)
        self.gen_c = 0
        self.gen_o = 0
        self._epoch = None     # (event, synced streams) of the last refresh that ran inside a graph replay (graphs.py)

    def get(self, key, param, build, recipe_fn=None):
        """recipe_fn(recorded launches, value) -> replay recipe or None: for packs whose replayable form is not the launch that
        built them (the bf16-plane weights: built as f32 pack + split, refreshed as one permute straight into planes)"""
        ver = (param._version, param.data_ptr())
        hit = self._c.get(key) or self._o.get(key)
        cap = graphs.capturing()
        sid = ops._stream_id(param.device.index) if param.is_cuda else None    # raw handle: no Stream object on the hit path
        if hit is not None and hit[0] == ver:
            if cap and key in self._o and hit[6] is not graphs.capture_store():
                graphs.capture_store()["hit_o"] = True
            # (inside a capture nothing from before it is pending -- graphs.Captured synchronises first -- and an event wait on
            # work outside the capture must not be recorded into it)
            if sid is not None and not cap:
                if hit[2] is not None and sid not in hit[3]:
                    torch.cuda.current_stream(param.device).wait_event(hit[2])
                    hit[3].add(sid)
                if self._epoch is not None and sid not in self._epoch[1]:
                    torch.cuda.current_stream(param.device).wait_event(self._epoch[0])
                    self._epoch[1].add(sid)
            return hit[1]
        st = torch.cuda.current_stream(param.device) if param.is_cuda else None
        # record what the build launches: a pack that is exactly ONE permute / cast into the returned tensor can be replayed by
        # refresh() (every pack helper below is); anything else is rebuilt lazily after a refresh
        rec = []
        prev, ops._pack_recorder = ops._pack_recorder, rec
        try:
            val = build()
        finally:
            ops._pack_recorder = prev
        # replayable only if the recorded source IS the parameter's storage: a pack helper that had to make a temporary copy first
        # (reshape of a non-contiguous parameter) would be re-packed from that stale temporary forever
        if recipe_fn is not None:
            recipe = recipe_fn(rec, val)
            if recipe is not None and recipe[0].untyped_storage().data_ptr() != param.untyped_storage().data_ptr():
                recipe = None
        else:
            recipe = rec[0] if (len(rec) == 1 and torch.is_tensor(val) and rec[0][1].data_ptr() == val.data_ptr()
                                and rec[0][0].untyped_storage().data_ptr() == param.untyped_storage().data_ptr()) else None
        ev = None
        if st is not None and not cap:
            ev = torch.cuda.Event()
            ev.record(st)
        entry = [ver, val, ev, {sid}, recipe, param, (graphs.capture_store() if cap else None)]
        if self._c.pop(key, None) is not None:
            self._replay = {}         # the batched refreshes were built over the dropped entry
            self.gen_c += 1
        if self._o.pop(key, None) is not None:
            self.gen_o += 1
        if recipe is not None and st is not None:
            self._c[key] = entry
            self._replay = {}
            self.gen_c += 1
        else:
            self._o[key] = entry
            self.gen_o += 1
        return val

    def generation(self, store=None):
        """validity stamp of a capture whose scratch dict is `store` (graphs.Captured)"""
        return (self.gen_c, self.gen_o if (store is None or store.get("hit_o")) else None)

    def clear(self):
        self._c.clear()
        self._o.clear()
        self._replay = {}
        self._epoch = None
        self.gen_c += 1
        self.gen_o += 1

    def purge_capture(self, store):
        """A HIP-graph capture whose scratch dict is `store` FAILED: the packs it built were only recorded, never executed -- their
        buffers (in the capture's private pool) hold nothing, yet they sit in the cache under the parameters' current versions.
        Drop them, so the eager path that takes over re-packs (graphs.Captured calls this from its failure path)."""
        dead_c = [k for k, e in self._c.items() if e[6] is store]
        dead_o = [k for k, e in self._o.items() if e[6] is store]
        for k in dead_c:
            del self._c[k]
        for k in dead_o:
            del self._o[k]
        if dead_c:
            self._replay = {}
            self.gen_c += 1
        if dead_o:
            self.gen_o += 1
        return len(dead_c) + len(dead_o)

    def refresh(self, tag=None, select=None):
        """The parameters were updated IN PLACE by a kernel torch does not see (TrainStep's Adam launch): re-run every pack into
        its existing destination in ONE launch (umr_permute4_batched) instead of dropping the copies and re-packing ~180
        weights one launch each during the next step.  Entries that are not a single permute are dropped (rebuilt lazily).
        tag / select(key): refresh only the entries select() accepts (one launch per tag: TrainStep updates and refreshes stage
        by stage, beside the rest of backward); the caller ends the round of partial refreshes with refresh_done()."""
        if tag is None:
            self.refresh_done()
        if not self._c:
            return
        if tag not in self._replay:
            assert not graphs.capturing(), "PackCache.refresh: the batched refresh must be built before a capture (warm-up steps)"
            keys = [k for k in self._c if select is None or select(k)]
            self._replay[tag] = (ops.permute4_batched([self._c[k][4] for k in keys]) if keys else None, keys)
        launch, keys = self._replay[tag]
        if launch is None:
            return
        if any(self._c[k][5].data_ptr() != self._c[k][0][1] for k in keys):   # a parameter's storage moved: the recipes are stale
            self.clear()
            return
        launch()
        if graphs.capturing():
            return                     # the replaying caller publishes the refresh with refreshed_by_replay()
        st = torch.cuda.current_stream(self._c[keys[0]][5].device)
        ev = torch.cuda.Event()
        ev.record(st)
        for k in keys:
            e = self._c[k]
            e[0] = (e[5]._version, e[5].data_ptr())
            e[2], e[3] = ev, {st.cuda_stream}

    def adam_and_refresh(self, tag, select, stage_params, lo, hi, bufs, hyper):
        """Optimizer step of the flat-buffer slice [lo, hi) AND the refresh of its packed copies, with the copies of its Linear
        weights written by the optimizer launch itself (ops.adam_pack / umr_adam_pack_step: the refresh pass that re-read those f32
        weights is gone; the other packs of the stage -- conv layouts, plane forms -- keep the batched permute).
        stage_params: [(name, element offset, numel, shape)] of the slice in buffer order; bufs = (flat p, g, m, v); hyper: device
        scalars of umr_adam_set_hyper.  Returns False (nothing launched) when no weight of the stage has a bf16 [N,K] / [K,N] copy
        -- the caller then runs the two-launch form.  Bit-identical to it (tests/test_train_gpu.py)."""
        rk = ("adam", tag)
        if rk not in self._replay:
            assert not graphs.capturing(), "PackCache.adam_and_refresh: the tables must be built before a capture (warm-up steps)"
            keys = [k for k in self._c if select is None or select(k)]
            by_name = {}
            for k in keys:
                by_name.setdefault(k[0], {})[k[1]] = k
            flat_p, flat_g, flat_m, flat_v = bufs
            entries, fused, cur = [], set(), lo
            for name, off, numel, shape in stage_params:
                kinds = by_name.get(name, {})
                ok = (len(shape) == 2 and shape[0] % 8 == 0 and shape[1] % 4 == 0 and off % 4 == 0 and kinds and set(kinds) <= {"lin", "lin_t"}
                      and all(torch.is_tensor(self._c[k][1]) and self._c[k][1].dtype == torch.bfloat16 and self._c[k][1].is_contiguous()
                              for k in kinds.values()))
                if ok:
                    dl = self._c[kinds["lin"]][1] if "lin" in kinds else None
                    dt_ = self._c[kinds["lin_t"]][1] if "lin_t" in kinds else None
                    ok = (dl is None or tuple(dl.shape) == tuple(shape)) and (dt_ is None or tuple(dt_.shape) == (shape[1], shape[0]))
                if not ok:
                    continue
                if cur < off:
                    entries.append(("plain",) + tuple(b[cur:off] for b in bufs))
                entries.append(("weight",) + tuple(b[off:off + numel].view(shape) for b in bufs) + (dl, dt_))
                fused.update(kinds.values())
                cur = off + numel
            if not fused:
                self._replay[rk] = None
            else:
                if cur < hi:
                    entries.append(("plain",) + tuple(b[cur:hi] for b in bufs))
                rest = [k for k in keys if k not in fused]
                self._replay[rk] = (ops.adam_pack(entries, hyper), ops.permute4_batched([self._c[k][4] for k in rest]) if rest else None, keys)
        rec = self._replay[rk]
        if rec is None:
            return False
        launch_adam, launch_rest, keys = rec
        if any(self._c[k][5].data_ptr() != self._c[k][0][1] for k in keys):   # a parameter's storage moved: the tables are stale
            self.clear()
            return False
        launch_adam()
        if launch_rest is not None:
            launch_rest()
        if graphs.capturing():
            return True                # the replaying caller publishes the refresh with refreshed_by_replay()
        st = torch.cuda.current_stream(self._c[keys[0]][5].device)
        ev = torch.cuda.Event()
        ev.record(st)
        for k in keys:
            e = self._c[k]
            e[0] = (e[5]._version, e[5].data_ptr())
            e[2], e[3] = ev, {st.cuda_stream}
        return True

    def refresh_done(self):
        """after the last (partial) refresh of a round: the copies that cannot be replayed are dropped (rebuilt on next use).
        Inside a capture, entries the capture built itself go silently (each of its replays rebuilds them); every OTHER dropped entry
        -- e.g. the collapsed head's weights an evaluation call cached between a TrainStep's warm-up and its capturing step -- counts
        as a change of the set, so an inference capture that read it (hit_o) is invalidated instead of replaying from freed memory."""
        if self._o:
            cur = graphs.capture_store() if graphs.capturing() else None
            foreign = cur is None or any(e[6] is not cur for e in self._o.values())
            self._o.clear()
            if foreign:
                self.gen_o += 1

    def synced_with(self, stream):
        """`stream` has waited for the stream(s) the refreshes ran on (a join): its later launches need no per-entry event wait"""
        sid = stream.cuda_stream
        for e in self._c.values():
            e[3].add(sid)

    def refreshed_by_replay(self, device):
        """A graph replay on the current stream has just re-run the optimizer step and the refresh: a consumer on another stream
        orders itself after it (one event for the whole cache instead of one per entry).  The replay updated the parameters on
        the device without running this class's host code, so what refresh_done() does after an eager step is done here: EVERY copy
        that cannot be replayed is stale in this host-side dict now and is dropped -- those built outside a capture (the collapsed head's
        weights an evaluation call cached between two training steps) and those an inference capture built inside itself: that capture
        holds the addresses and rewrites them on each of its replays, but a cache HIT by anybody else (an eager call of another shape,
        a second capture) would read what the first graph's LAST replay wrote, i.e. weights one or more steps old (round-5 advisor)."""
        if self._o:
            self._o.clear()
            self.gen_o += 1
        st = torch.cuda.current_stream(device)
        ev = torch.cuda.Event()
        ev.record(st)
        self._epoch = (ev, {st.cuda_stream})


def _pack_linear(w, dt):  # [N,K] -> [N,K] T
    return ops.cast(w.detach().reshape(w.shape[0], -1), dt)


def _pack_linear_t(w2d, dt):  # [N,K] (possibly strided rows) -> [K,N] T
    N, K = w2d.shape
    out = torch.empty((K, N), dtype=dt, device=w2d.device)
    return ops.permute4(w2d, out, (1, 1, K, N), (0, 0, w2d.stride(1), w2d.stride(0)), src_offset=0)


def _pack_conv3(w, dt):  # [co,ci,3,3] -> [co][ky][kx][ci]
    co, ci = w.shape[0], w.shape[1]
    st = w.stride()
    out = torch.empty((co, 9 * ci), dtype=dt, device=w.device)
    return ops.permute4(w.detach(), out, (co, 3, 3, ci), (st[0], st[2], st[3], st[1]))


def _pack_conv3_dgrad(w, dt):  # [co,ci,3,3] -> [ci][2-ky][2-kx][co]
    co, ci = w.shape[0], w.shape[1]
    st = w.stride()
    out = torch.empty((ci, 9 * co), dtype=dt, device=w.device)
    return ops.permute4(w.detach(), out, (ci, 3, 3, co), (st[1], -st[2], -st[3], st[0]), src_offset=2 * st[2] + 2 * st[3])


def _unpack_conv3_grad(dwp, grad_out):  # [co][ky][kx][ci] f32 -> [co,ci,3,3] f32
    co, ci = grad_out.shape[0], grad_out.shape[1]
    return ops.permute4(dwp, grad_out, (co, ci, 3, 3), (9 * ci, 1, 3 * ci, ci))


def _pack_convT(w, dt):  # ConvTranspose2d [ci,co,s,s] -> [(i,j,co)][ci]
    ci, co, s, _ = w.shape
    st = w.stride()
    out = torch.empty((s * s * co, ci), dtype=dt, device=w.device)
    return ops.permute4(w.detach(), out, (s, s, co, ci), (st[2], st[3], st[1], st[0]))


def _pack_convT_dgrad(w, dt):  # -> [ci][(i,j,co)]
    ci, co, s, _ = w.shape
    st = w.stride()
    out = torch.empty((ci, s * s * co), dtype=dt, device=w.device)
    return ops.permute4(w.detach(), out, (ci, s, s, co), (st[0], st[2], st[3], st[1]))


def _rep_bias(b, reps):
    out = torch.empty(reps * b.numel(), dtype=torch.float32, device=b.device)
    return ops.permute4(b.detach(), out, (1, 1, reps, b.numel()), (0, 0, 0, 1))


_FUSE_HEAD_OUT = os.environ.get("UMR_FUSE_HEAD_OUT", "1") != "0"  # A/B switch for benchmarking
_X3_HEADS = os.environ.get("UMR_X3_HEADS", "1") != "0"            # A/B switch: fp32-mode inference heads on the bf16-plane kernel
_X3_ALL = os.environ.get("UMR_X3_ALL", "1") != "0"            # A/B switch: 0 = fp32 mode as in round 3 (only the heads on the plane kernels)
_MERGE_DFEAT = os.environ.get("UMR_MERGE_DFEAT", "1") != "0"    # A/B switch: one GEMM for the feature-map gradient of both heads
# A 1x1 convolution and a bilinear resize commute exactly (both linear, one per pixel across channels, the other per channel across
# pixels with weights that sum to 1, so the bias commutes too): conv1x1(resize(x)) == resize(conv1x1(x)).  The fusion blocks'
# out_conv (blocks.py:377-381: interpolate x2, then out_conv) and the first layer of both heads (objectness_net.py:110,121 on the
# x2-interpolated feature map, models.py:70-72) therefore run on the map BEFORE the resize -- a quarter of the rows in the GEMM, its
# weight gradient and its data gradient -- and the resize moves the GEMM's output.  0 = the reference's order (A/B switch).
_COMMUTE_RESIZE = os.environ.get("UMR_COMMUTE_RESIZE", "1") != "0"
# Backward of a head without non-linearities between its convs (objectness_net.py:119-142): "algebraic" (default) = exact
# gradients of all eight factored tensors from three pixel reductions (no 512/1024-channel tensor is stored, read or
# multiplied in backward); "gemm" = the layer-by-layer data/weight-gradient GEMMs (A/B switch, the round-1 form).
_LINEAR_HEAD_BWD = os.environ.get("UMR_LINEAR_HEAD_BWD", "algebraic")


_WGRAD_STREAM = os.environ.get("UMR_WGRAD_STREAM", "auto")   # weight gradients on a second stream: auto (small problems) | 0 | 1
# LayerNorm's parameter gradients (the reduction pass of layernorm_bwd) ride on the weight-gradient lane when there is one (A/B switch)
_LN_PARAMS_SIDE = os.environ.get("UMR_LN_PARAMS_SIDE", "1") != "0"
_side_streams = {}


class WgradStream:
    """Weight gradients are leaves of the backward pass: nothing in it reads them.  For small problems (the reference's recipe:
    20 crops of 128x128, README.md:148-155 -- GEMMs of 24-264 tiles on 256 CUs) they run on a second HIP stream, beside the data-
    gradient chain instead of inside it; under a HIP-graph capture the fork and join are graph edges and cost the host nothing.
    Same kernels, same operands, same order per stream: results are bit-identical to the one-stream schedule.  Large problems
    fill the chip with every launch and keep one stream (and their memory: a tensor read on the side stream cannot be reused by
    the allocator until that stream has passed it)."""

    def __init__(self, dev, on):
        self.on = bool(on)
        # a staged capture in progress (graphs.StagedCaptured): the side-stream work is not launched here but deferred to the capture,
        # which records it as the stage's own graph at the next stage boundary and replays that graph on its second stream
        self.staged = graphs.staged() if self.on else None
        if self.on and self.staged is None:
            self.main = torch.cuda.current_stream(dev)
            key = (dev.index, self.main.cuda_stream)
            if key not in _side_streams:
                if len(_side_streams) > 16:
                    _side_streams.clear()
                _side_streams[key] = torch.cuda.Stream(device=dev)
            self.side = _side_streams[key]

    def run(self, fn, *used):
        """fn: launches on the current stream; used: tensors (allocated on the main stream) those launches read"""
        if not self.on:
            return fn()
        if self.staged is not None:
            return self.staged.defer(fn, used)
        self.side.wait_stream(self.main)       # the producers of `used` are enqueued on main
        graphs.note_fork(self.main, self.side)
        with torch.cuda.stream(self.side):
            fn()
        for t in used:
            if t is not None:
                t.record_stream(self.side)

    def stage_end(self):
        """a stage of backward is complete (engine.backward's cb): a staged capture closes the stage's graphs here"""
        if self.staged is not None:
            self.staged.boundary()

    def join(self):
        if self.on and self.staged is None:
            self.main.wait_stream(self.side)
        elif self.staged is not None:
            self.staged.join()

    @staticmethod
    def wanted(pixels):
        if _WGRAD_STREAM in ("0", "1"):
            return _WGRAD_STREAM == "1"
        return pixels <= graphs.AUTO_MAX_PIXELS


class Engine(X3Path):
    def __init__(self, cfg, head_layouts, compute_dtype=torch.float32, collapse_linear_heads=False, linear_head_backward=None):
        self.cfg = cfg
        self.linear_head_backward = linear_head_backward or _LINEAR_HEAD_BWD
        assert self.linear_head_backward in ("algebraic", "gemm")
        self.center_layout, self.sdf_layout = head_layouts
        self.dt = compute_dtype
        self.cache = PackCache()
        # algebraic fast path for heads without non-linearities between their convs: "auto" (the net's
        # default) = such a head is evaluated as ONE 3x3 convolution at inference and, since round 6, in training whenever its
        # backward is the algebraic one (_collapse); True = always; False = never (the four convolutions as the reference runs them)
        assert collapse_linear_heads in (False, True, "auto")
        self.collapse_linear_heads = collapse_linear_heads

    # ------------------------------------------------------------------ collapsed linear head (opt-in)
    def _linear_head_weights(self, P, name, idx, dev):
        """The collapsed form of a linear head W4 W3 (W2 * (W1 x + b1) + b2) + b3) + b4 = one 3x3 conv 256 -> 1 plus a
        border-dependent bias.  All weight algebra is f32 on the master weights (element strides of the PyTorch layouts, no
        packing):  u = W4 W3 [512];  Vc[ci][t] = sum_co u[co] W2[co,ci,t];  Kw[t][c] = sum_ci Vc[ci][t] W1[ci][c];
        tb[t] = sum_ci Vc[ci][t] b1[ci] (t < 9);  tb[9] = u.b2 + W4.b3 + b4."""
        W1, b1 = self._f32(P, f"{name}.{idx[0]}.weight"), self._f32(P, f"{name}.{idx[0]}.bias")
        W2, b2 = self._f32(P, f"{name}.{idx[1]}.weight"), self._f32(P, f"{name}.{idx[1]}.bias")
        W3, b3 = self._f32(P, f"{name}.{idx[2]}.weight"), self._f32(P, f"{name}.{idx[2]}.bias")
        W4, b4 = self._f32(P, f"{name}.{idx[3]}.weight"), self._f32(P, f"{name}.{idx[3]}.bias")
        C, C1, C3 = W1.shape[1], W1.shape[0], W3.shape[0]  # 256, 512, 1024
        assert W4.shape[0] == 1 and W2.shape == (C1, C1, 3, 3) and all(t.is_contiguous() for t in (W1, W2, W3, W4))
        f = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
        u, Vc, Kw, tb = f(C1), f(C1 * 9), f(9 * C), f(10)
        ops.small_gemm(W4, W3, u, 1, C1, C3, (0, 1), (C1, 1), (0, 1))
        ops.small_gemm(u, W2, Vc, 1, C1 * 9, C1, (0, 1), (C1 * 9, 1), (0, 1))
        ops.small_gemm(Vc, W1, Kw, 9, C, C1, (1, 9), (C, 1), (C, 1))
        ops.small_gemm(Vc, b1, tb, 9, 1, C1, (1, 9), (1, 0), (1, 0))
        ops.cast(b4, torch.float32, out=tb[9:10])
        ops.small_gemm(u, b2, tb[9:10], 1, 1, C1, (0, 1), (1, 0), (0, 0), accumulate=True)
        ops.small_gemm(W4, b3, tb[9:10], 1, 1, C3, (0, 1), (1, 0), (0, 0), accumulate=True)
        return u, Vc, Kw, tb

    def _collapse(self, lay, save):
        """does this call evaluate the head `lay` in its collapsed form?  (a head with ReLUs never; 'sine' is not invertible from
        its value, so its training step keeps the factored form whose backward gets the pre-activation)"""
        if lay["relu"] or (save and lay["final"] == "sine"):
            return False
        if self.collapse_linear_heads == "auto":
            # inference: always.  Training: when the head's backward is the algebraic one (the default) -- that backward reads the
            # head's OUTPUT only, so the 512 / 512 / 1024-channel maps of the four-convolution forward would be computed and thrown
            # away (qualified by tests/test_collapsed_train_gpu.py).  With the layer-by-layer GEMM
            # backward (set_linear_head_backward('gemm')) the forward keeps the four convolutions whose activations it reads.
            return (not save) or self.linear_head_backward == "algebraic"
        return bool(self.collapse_linear_heads)

    def _linear_head_weights_cached(self, P, name, idx, dev):
        """inference: the collapsed weights (and the 16-row tap matrix in the compute dtype) per version of the head's eight
        tensors (PackCache; dropped with the other non-replayable packs after an optimizer step of TrainStep)"""
        ps = [P[f"{name}.{i}.{t}"] for i in idx for t in ("weight", "bias")]

        def build():
            u, Vc, Kw, tb = self._linear_head_weights(P, name, idx, dev)
            return u, Vc, Kw, tb, self._tap_matrix(Kw)
        return self.cache.get((name, "collapsed", self.dt), ParamGroup(ps), build)

    def _tap_matrix(self, Kw):
        """[16, C] in the compute dtype: rows 0..8 = the nine tap vectors, the rest zero (the N = 16 operand of the tap GEMM)"""
        C = Kw.numel() // 9
        k16 = torch.zeros((16, C), dtype=torch.float32, device=Kw.device)
        k16[:9].copy_(Kw.view(9, C))
        return k16 if self.dt == torch.float32 else ops.cast(k16, self.dt)

    def _linear_head_forward_lowres(self, P, name, idx, path, H, W, act, save):
        """Collapsed forward of a head that reads the x2-interpolated feature map (models.py:70-72, objectness_net.py:128-135), taken
        BEFORE the resize: the nine tap products kw[t] . x are 1x1 products and commute with the resize (_COMMUTE_RESIZE), so they
        are one 16-column GEMM on the small map `path` [nb, ph, pw, 256]; the resize moves 16 f32 channels instead of 256, and
        lh_gather9_kernel sums every pixel's taps at their shifted positions (csrc/linear_head.hip).  The backward of a training
        step in this form is the algebraic one (it needs the output only)."""
        if save:
            u, Vc, Kw, tb = self._linear_head_weights(P, name, idx, path.device)
            k16 = self._tap_matrix(Kw)
        else:
            u, Vc, Kw, tb, k16 = self._linear_head_weights_cached(P, name, idx, path.device)
        nb, ph, pw, C = path.shape
        taps = ops.gemm_nt(path.reshape(-1, C), k16, None, out_f32=(self.dt != torch.float32))                    # [Ml, 16] f32
        taps = ops.bilinear_fwd(taps.view(nb, ph, pw, 16), H, W, True)                        # [B, H, W, 16] f32
        out = ops.linear_head_gather9(taps, tb, act)
        return out, dict(algebraic=True, act=act, out=out, u=u, Vc=Vc, Kw=Kw)

    def _linear_head_forward(self, P, name, idx, feat, act):
        """opt-in collapsed forward (csrc/linear_head.hip): the head as ONE streaming 3x3 conv 256 -> 1"""
        u, Vc, Kw, tb = self._linear_head_weights(P, name, idx, feat.device)
        out = ops.linear_head_fwd(feat, Kw, tb, act)
        return out, dict(collapsed=True, u=u, Vc=Vc, Kw=Kw, act=act)

    def _linear_head_backward(self, P, name, idx, feat, hs, dout, dfeat, G):
        """Gradients of every factored parameter from the three pixel reductions R = [G | n | D] (csrc/linear_head.hip)."""
        W1, b1 = self._f32(P, f"{name}.{idx[0]}.weight"), self._f32(P, f"{name}.{idx[0]}.bias")
        W2, b2 = self._f32(P, f"{name}.{idx[1]}.weight"), self._f32(P, f"{name}.{idx[1]}.bias")
        W3, b3 = self._f32(P, f"{name}.{idx[2]}.weight"), self._f32(P, f"{name}.{idx[2]}.bias")
        W4 = self._f32(P, f"{name}.{idx[3]}.weight")
        C, C1, C3 = W1.shape[1], W1.shape[0], W3.shape[0]
        dev = feat.device
        if "u" not in hs:   # factored forward, algebraic backward: the weight algebra is done here
            hs["u"], hs["Vc"], hs["Kw"], _ = self._linear_head_weights(P, name, idx, dev)
        act, u, Vc, Kw = hs["act"], hs["u"], hs["Vc"], hs["Kw"]
        dout = dout.contiguous()
        R = ops.linear_head_bwd_weight(feat, dout, hs["out"], act)
        Gm, n, D = R[:9 * C], R[9 * C:9 * C + 9], R[9 * C + 9:]
        # data gradient (accumulates into the centre head's dfeat when there is one)
        if dfeat is None:
            dfeat = torch.empty_like(feat)
            ops.linear_head_bwd_data(dout, hs["out"], Kw, dfeat, act, False)
        else:
            ops.linear_head_bwd_data(dout, hs["out"], Kw, dfeat.view(feat.shape), act, True)
        self._linear_head_algebra(P, name, idx, hs, Gm, n, D, G)
        return dfeat

    def _linear_head_algebra(self, P, name, idx, hs, Gm, n, D, G):
        """the factored parameters' gradients from G [9*C] (tap major), n [9], D [1] (all f32)"""
        W1, b1 = self._f32(P, f"{name}.{idx[0]}.weight"), self._f32(P, f"{name}.{idx[0]}.bias")
        W2, b2 = self._f32(P, f"{name}.{idx[1]}.weight"), self._f32(P, f"{name}.{idx[1]}.bias")
        W3, b3 = self._f32(P, f"{name}.{idx[2]}.weight"), self._f32(P, f"{name}.{idx[2]}.bias")
        W4 = self._f32(P, f"{name}.{idx[3]}.weight")
        C, C1, C3 = W1.shape[1], W1.shape[0], W3.shape[0]
        dev = Gm.device
        u, Vc = hs["u"], hs["Vc"]
        g = lambda k: (G[f"{name}.{idx[k]}.weight"], G[f"{name}.{idx[k]}.bias"])
        (gW1, gb1), (gW2, gb2), (gW3, gb3), (gW4, gb4) = g(0), g(1), g(2), g(3)
        gvc = torch.empty(C1 * 9, dtype=torch.float32, device=dev)
        gu = torch.empty(C1, dtype=torch.float32, device=dev)
        # gv[ci][t] = sum_c W1[ci][c] G[t][c] + n[t] b1[ci]
        ops.small_gemm(W1, Gm, gvc, C1, 9, C, (C, 1), (1, C), (9, 1))
        ops.small_gemm(b1, n, gvc, C1, 9, 1, (1, 0), (0, 1), (9, 1), accumulate=True)
        ops.small_gemm(Vc, Gm, gW1, C1, C, 9, (9, 1), (C, 1), (C, 1))           # dW1 = Vc G
        ops.small_gemm(Vc, n, gb1, C1, 1, 9, (9, 1), (1, 0), (1, 0))            # db1 = Vc n
        ops.small_gemm(u, gvc, gW2, C1, C1 * 9, 1, (1, 0), (0, 1), (C1 * 9, 1))  # dW2[co,ci,t] = u[co] gv[ci][t]
        ops.small_gemm(W2, gvc, gu, C1, 1, C1 * 9, (C1 * 9, 1), (1, 0), (1, 0))  # gu = W2 gv (+ D b2)
        ops.small_gemm(b2, D, gu, C1, 1, 1, (1, 0), (0, 0), (1, 0), accumulate=True)
        ops.small_gemm(u, D, gb2, C1, 1, 1, (1, 0), (0, 0), (1, 0))             # db2 = D u
        ops.small_gemm(gu, W3, gW4, 1, C3, C1, (0, 1), (1, C1), (0, 1))         # dW4 = gu W3^T (+ D b3)
        ops.small_gemm(D, b3, gW4, 1, C3, 1, (0, 0), (0, 1), (0, 1), accumulate=True)
        ops.small_gemm(W4, gu, gW3, C3, C1, 1, (1, 0), (0, 1), (C1, 1))         # dW3 = W4^T gu
        ops.small_gemm(W4, D, gb3, C3, 1, 1, (1, 0), (0, 0), (1, 0))            # db3 = D W4^T
        ops.cast(D, torch.float32, out=gb4.view(1))                              # db4 = D

    # ------------------------------------------------------------------ helpers
    def _w(self, P, name, kind):
        p = P[name]
        if kind == "lin" and self.dt == torch.float32 and p.dtype == torch.float32 and p.is_contiguous():
            return p.detach().reshape(p.shape[0], -1)    # fp32 mode: the [N, K] kernel layout IS the parameter's layout -- no copy

        return self.cache.get((name, kind, self.dt), p, lambda: {
            "lin": lambda: _pack_linear(p, self.dt),
            "lin_t": lambda: _pack_linear_t(p.detach().reshape(p.shape[0], -1), self.dt),
            "c3": lambda: _pack_conv3(p, self.dt),
            "c3_d": lambda: _pack_conv3_dgrad(p, self.dt),
            "ct": lambda: _pack_convT(p, self.dt),
            "ct_d": lambda: _pack_convT_dgrad(p, self.dt),
        }[kind]())

    def _wx3(self, P, name, kind):
        """f32 weights as three bf16 planes per value ([rows, 3K]; conv: K = (ky,kx,ci) inside each plane) for ops.gemm_nt_x3.
        kinds: the layouts of _w plus 'lin_a' / 'lin_b' (/ 'lin_t_a' / 'lin_t_b'): the token / class-token column halves of the
        readout projection [D, 2D] (models/dpt/vit.py:84-88).  Refreshed after an optimizer step by the batched permute, straight
        from the parameter into the planes (PackCache.refresh)."""
        p = P[name]

        def pack_f32():
            w = p.detach()
            if kind == "lin":
                return w.reshape(w.shape[0], -1).contiguous()
            if kind == "lin_t":
                return _pack_linear_t(w.reshape(w.shape[0], -1), torch.float32)
            if kind in ("lin_a", "lin_b", "lin_t_a", "lin_t_b"):
                D = w.shape[1] // 2
                half = w[:, :D] if kind.endswith("a") else w[:, D:]
                if kind.startswith("lin_t"):
                    return _pack_linear_t(half, torch.float32)
                out = torch.empty((w.shape[0], D), dtype=torch.float32, device=w.device)
                return ops.permute4(w, out, (1, 1, w.shape[0], D), (0, 0, w.stride(0), 1), src_offset=(0 if kind.endswith("a") else D))
            return {"c3": _pack_conv3, "c3_d": _pack_conv3_dgrad, "ct": _pack_convT, "ct_d": _pack_convT_dgrad}[kind](w, torch.float32)

        def build():
            return ops.split3(pack_f32())

        def recipe(rec, val):
            rowlen = val.shape[1] // 3
            if kind == "lin":          # no launch recorded: the pack is the parameter itself
                return (p.detach(), val, (1, 1, 1, p.numel()), (0, 0, 0, 1), 0, rowlen) if p.is_contiguous() else None
            return (rec[0][0], val, rec[0][2], rec[0][3], rec[0][4], rowlen) if len(rec) == 1 else None

        return self.cache.get((name, kind + "_x3"), p, build, recipe)

    def _f32(self, P, name):
        p = P[name].detach()
        assert p.dtype == torch.float32, "parameters must be fp32 (call .to(torch.float32))"
        return p

    # ------------------------------------------------------------------ forward
    def forward(self, P, images, save, skip=()):
        """P: dict name -> fp32 parameter tensor on the GPU (reference state-dict names).
        images: [B,3,H,W] fp32 on the GPU.  Returns (center [B,2,H,W] f32, sdf [B,1,H,W] f32, saved).
        skip (inference only): head module names that are not evaluated (their output is None) -- the boundary-reasoning rounds of
        object_reasoning.py:379-487 read the boundary-distance map alone, and the centre head is most of a 128x128 crop's forward."""
        cfg, dt = self.cfg, self.dt
        assert not (skip and save), "skip: inference only"
        if _X3_ALL and dt == torch.float32 and ops.get_f32_mode() in ("x3", "x3_fast"):
            return self.forward_x3(P, images, save, skip)      # fp32 parity mode on the bf16-plane kernels (engine_x3.py)
        assert images.is_cuda and images.dtype == torch.float32 and images.dim() == 4 and images.shape[1] == 3
        images = images.contiguous()
        B, _, H, W = images.shape
        p, D, heads = cfg["patch"], cfg["D"], cfg["heads"]
        gh, gw = H // p, W // p
        assert gh >= 1 and gw >= 1
        g, Nt = gh * gw, gh * gw + 1
        m = "backbone.pretrained.model."
        S = {"B": B, "H": H, "W": W, "gh": gh, "gw": gw} if save else None

        # ---- patch embed + cls + pos (vit.py:168-193)
        G = cfg["pos_grid"]
        pos = self._f32(P, m + "pos_embed")[0]  # [1+G*G, D]
        if (gh, gw) != (G, G):
            pos_grid = ops.bilinear_fwd(pos[1:].reshape(1, G, G, D), gh, gw, False).reshape(g, D)
        else:
            pos_grid = pos[1:]
        pos_t = pos_grid if dt == torch.float32 else ops.cast(pos_grid.contiguous(), dt)
        K = 3 * p * p
        ldk = (K + 7) // 8 * 8
        patches = ops.patchify(images, p, dt, ldk)
        wp = self._w(P, m + "patch_embed.proj.weight", "lin")
        if ldk != K:
            wpad = torch.zeros((D, ldk), dtype=dt, device=images.device)
            wpad[:, :K] = wp
            wp = wpad
        tokens = torch.empty((B * Nt, D), dtype=dt, device=images.device)
        ops.gemm_nt(patches, wp, self._f32(P, m + "patch_embed.proj.bias"), out=tokens, aux=pos_t.contiguous(), aux_mod=g,
                    c_remap=(g, Nt, 1))
        ops.fill_cls(tokens, self._f32(P, m + "cls_token").reshape(-1), pos[0].contiguous(), B, Nt * D, D)
        if save:
            S["patches"] = patches

        # ---- transformer blocks (timm Block; only up to the last hooked block: later ones feed nothing, vit.py:107)
        x = tokens
        acts = []
        blocks = []
        for i in range(max(cfg["hooks"]) + 1):
            b = m + f"blocks.{i}."
            ln1, mean1, rstd1 = ops.layernorm_fwd(x, self._f32(P, b + "norm1.weight"), self._f32(P, b + "norm1.bias"))
            qkv = ops.gemm_nt(ln1, self._w(P, b + "attn.qkv.weight", "lin"), self._f32(P, b + "attn.qkv.bias"))
            att, lse = ops.attention_fwd(qkv, B, Nt, heads, need_lse=save)
            x1 = ops.gemm_nt(att, self._w(P, b + "attn.proj.weight", "lin"), self._f32(P, b + "attn.proj.bias"), aux=x)
            ln2, mean2, rstd2 = ops.layernorm_fwd(x1, self._f32(P, b + "norm2.weight"), self._f32(P, b + "norm2.bias"))
            if save:
                h, hpre = ops.gemm_nt(ln2, self._w(P, b + "mlp.fc1.weight", "lin"), self._f32(P, b + "mlp.fc1.bias"),
                                      act=L.ACT_GELU, c2_mode=2)
            else:
                h, hpre = ops.gemm_nt(ln2, self._w(P, b + "mlp.fc1.weight", "lin"), self._f32(P, b + "mlp.fc1.bias"), act=L.ACT_GELU), None
            x2 = ops.gemm_nt(h, self._w(P, b + "mlp.fc2.weight", "lin"), self._f32(P, b + "mlp.fc2.bias"), aux=x1)
            if save:
                blocks.append(dict(x=x, mean1=mean1, rstd1=rstd1, ln1=ln1, qkv=qkv, att=att, lse=lse, x1=x1, mean2=mean2,
                                   rstd2=rstd2, ln2=ln2, hpre=hpre, h=h))
            x = x2
            if i in cfg["hooks"]:
                acts.append(x)
        if save:
            S["blocks"] = blocks
            S["acts"] = acts

        # ---- readout + reassemble (vit.py:86-90,104-145,259-336)
        pp = "backbone.pretrained."
        Fs = cfg["features"]
        layers = []
        re_saved = []
        for k in range(4):
            a = pp + f"act_postprocess{k + 1}."
            wname = a + "0.project.0.weight"
            w_full = self.cache.get((wname, "lin", dt), P[wname], lambda: _pack_linear(P[wname], dt))  # [D, 2D]
            tok = acts[k]
            rb = ops.gemm_nt(tok, w_full[:, D:], self._f32(P, a + "0.project.0.bias"), M=B, lda=Nt * D, out_f32=True)  # cls part + bias
            if save:
                r, rpre = ops.gemm_nt(tok, w_full[:, :D], None, rowbias=rb, rows_per_batch=g, act=L.ACT_GELU, c2_mode=2, M=B * g,
                                      a_remap=(g, Nt, 1))
            else:
                r, rpre = ops.gemm_nt(tok, w_full[:, :D], None, rowbias=rb, rows_per_batch=g, act=L.ACT_GELU, M=B * g,
                                      a_remap=(g, Nt, 1)), None
            f = ops.gemm_nt(r, self._w(P, a + "3.weight", "lin"), self._f32(P, a + "3.bias"))  # [B*g, F]
            F_ = Fs[k]
            if k in (0, 1):
                s = 4 if k == 0 else 2
                bname = a + "4.bias"
                brep = self.cache.get((bname, "rep", s), P[bname], lambda: _rep_bias(P[bname], s * s))
                y = ops.gemm_nt(f, self._w(P, a + "4.weight", "ct"), brep)
                lay = ops.pixel_shuffle(y, B, gh, gw, s, F_)
            elif k == 2:
                lay = f.view(B, gh, gw, F_)
            else:
                lay = ops.gemm_nt(f.view(B, gh, gw, F_), self._w(P, a + "4.weight", "c3"), self._f32(P, a + "4.bias"), conv=2)
                lay = lay.view(B, (gh - 1) // 2 + 1, (gw - 1) // 2 + 1, F_)
            layers.append(lay)
            if save:
                re_saved.append(dict(r=r, rpre=rpre, f=f))
        if save:
            S["re"] = re_saved
            S["layers"] = layers

        # ---- scratch convs + refinenets (models.py:80-91, blocks.py:290-383)
        sc = "backbone.scratch."
        rn, rn_relu = [], []
        for k in range(4):
            lay = layers[k]
            o, orl = ops.gemm_nt(lay, self._w(P, sc + f"layer{k + 1}_rn.weight", "c3"), None, conv=1, c2_mode=1)
            rn.append(o.view(lay.shape[0], lay.shape[1], lay.shape[2], 256))
            rn_relu.append(orl.view_as(rn[-1]))
        fus_saved = {}
        path = None
        for k in (4, 3, 2, 1):
            r_ = sc + f"refinenet{k}."
            x1_, x1_relu = rn[k - 1], rn_relu[k - 1]
            nb, hh, ww, _ = x1_.shape
            fs = {}
            if path is None:
                s_, s_relu = x1_, x1_relu
            else:
                assert path.shape == x1_.shape, "fusion skip/size mismatch"
                t1 = ops.gemm_nt(x1_relu, self._w(P, r_ + "resConfUnit1.conv1.weight", "c3"),
                                 self._f32(P, r_ + "resConfUnit1.conv1.bias"), conv=1, act=L.ACT_RELU).view_as(x1_)
                s_, s_relu = ops.gemm_nt(t1, self._w(P, r_ + "resConfUnit1.conv2.weight", "c3"),
                                         self._f32(P, r_ + "resConfUnit1.conv2.bias"), conv=1, aux=x1_, aux2=path, c2_mode=1)
                s_, s_relu = s_.view_as(x1_), s_relu.view_as(x1_)
                fs.update(x1_relu=x1_relu, t1=t1)
            t2 = ops.gemm_nt(s_relu, self._w(P, r_ + "resConfUnit2.conv1.weight", "c3"),
                             self._f32(P, r_ + "resConfUnit2.conv1.bias"), conv=1, act=L.ACT_RELU).view_as(x1_)
            u = ops.gemm_nt(t2, self._w(P, r_ + "resConfUnit2.conv2.weight", "c3"),
                            self._f32(P, r_ + "resConfUnit2.conv2.bias"), conv=1, aux=s_).view_as(x1_)
            if k > 1 and cfg["patch"] != 16:
                nxt = rn[k - 2].shape
                Ho, Wo = nxt[1], nxt[2]  # patch-14 extension: resize to the next skip's size
            else:
                # the reference's wiring: exactly x2 (blocks.py:377-379); a token grid that does not survive the stride-2 conv and
                # the doublings (e.g. an odd grid) then fails at the skip addition, as the reference does (blocks.py:372)
                Ho, Wo = 2 * hh, 2 * ww
            if _COMMUTE_RESIZE:
                # out_conv before the resize (see _COMMUTE_RESIZE): its backward reads u, not the 4x larger resized map
                ul = ops.gemm_nt(u.view(-1, 256), self._w(P, r_ + "out_conv.weight", "lin"), self._f32(P, r_ + "out_conv.bias"))
                path = ops.bilinear_fwd(ul.view(nb, hh, ww, 256), Ho, Wo, True)
                up = None
                del ul
            else:
                up = ops.bilinear_fwd(u, Ho, Wo, True)
                path = ops.gemm_nt(up.view(-1, 256), self._w(P, r_ + "out_conv.weight", "lin"), self._f32(P, r_ + "out_conv.bias"))
                path = path.view(nb, Ho, Wo, 256)
            if save:
                fs.update(s_relu=s_relu, t2=t2, in_hw=(hh, ww))
                fs.update(dict(u=u) if up is None else dict(up=up))
                fus_saved[k] = fs
        if cfg["patch"] == 16:
            # models.py:70-72: Interpolate(scale_factor=2) -- the maps have 32 * (grid // 2 ...) = 16 * grid pixels per side, which is
            # the input size whenever that is a multiple of 16 (every documented use); otherwise smaller, exactly as in the reference
            H, W = 2 * path.shape[1], 2 * path.shape[2]
            if save:
                S["H"], S["W"] = H, W
        x3_ok = _X3_HEADS and dt == torch.float32 and ops.get_f32_mode() in ("x3", "x3_fast")
        # the heads' first layer before the final resize (_COMMUTE_RESIZE): the interpolated 256-channel feature map is never
        # formed, and the backward of that layer -- weight gradient, data gradient, the algebraic head's reductions -- runs on
        # the quarter-size map
        lowres = _COMMUTE_RESIZE and not x3_ok
        feat = ops.bilinear_fwd(path, H, W, True) if not lowres else None
        if save:
            S["fus"] = fus_saved
            S["rn_in"] = layers
            S["path1_hw"] = (path.shape[1], path.shape[2])
            S["feat"] = feat
            if lowres:
                S["path"] = path

        # ---- heads (objectness_net.py:109-135)
        outs = []
        heads_saved = []
        # fp32 (parity) mode at inference: the heads' three big layers run on the persistent 256x256 bf16 kernel with every f32
        # value held as three bf16 planes (six plane pairs per K-tile, csrc/gemm_nt256p.hip X3) -- the same six-term products as
        # the 128x128 fp32 kernel's in-register split (UMR_F32_X3), without the split arithmetic in the loop.  Training keeps
        # f32 activations (its backward reads them), and the exact-f32 mode keeps the f32 MFMA.
        featp = None
        for name, lay in (("center_field_prediction_head", self.center_layout), ("sdf_prediction_head", self.sdf_layout)):
            if name in skip:
                outs.append(None)
                continue
            idx = lay["conv_idx"]
            # in training the plane form serves the heads that keep no activation for their backward (algebraic backward)
            alg_ = save and not lay["relu"] and lay["final"] != "sine" and self.linear_head_backward == "algebraic"
            collapse = self._collapse(lay, save)
            x3_heads = x3_ok and (not save or alg_) and not collapse
            if x3_heads and featp is None:
                featp = ops.split3(feat.view(-1, 256))
            if collapse:
                if lowres:
                    out, cs = self._linear_head_forward_lowres(P, name, idx, path, H, W, _ACT[lay["final"]], save)
                else:
                    out, cs = self._linear_head_forward(P, name, idx, feat, _ACT[lay["final"]])
                    cs["out"] = out
                outs.append(out)
                if save:
                    heads_saved.append(cs)
                continue
            act = L.ACT_RELU if lay["relu"] else L.ACT_NONE
            if x3_heads:
                b_ = lambda k: self._f32(P, f"{name}.{idx[k]}.bias")
                h1p = ops.gemm_nt_x3(featp, self._wx3(P, f"{name}.{idx[0]}.weight", "lin"), b_(0), act=act, out_planes=True)
                h2p = ops.gemm_nt_x3(h1p.view(B, H, W, -1), self._wx3(P, f"{name}.{idx[1]}.weight", "c3"), b_(1), act=act, conv=1, out_planes=True)
                del h1p
                # the 1024 -> {1,2} output layer rides in the epilogue of the layer that produces its input: h3 is never stored
                w4 = self._f32(P, f"{name}.{idx[3]}.weight")
                parts = ops.gemm_nt_x3(h2p, self._wx3(P, f"{name}.{idx[2]}.weight", "lin"), b_(2), act=act,
                                       red_w=w4.reshape(w4.shape[0], -1).contiguous())
                del h2p
                out = ops.head_out_finish(parts, b_(3), B, H, W, _ACT[lay["final"]])
                del parts
                outs.append(out)
                if save:
                    heads_saved.append(dict(algebraic=True, act=_ACT[lay["final"]], out=out))
                continue
            # a head that is linear up to its output activation needs none of its 512/1024-channel activations in backward
            # (exact gradients from three pixel reductions over feat, _linear_head_backward); sin is not invertible from its value
            algebraic = save and not lay["relu"] and lay["final"] != "sine" and self.linear_head_backward == "algebraic"
            keep = save and not algebraic
            if x3_ok and keep and lay["final"] != "sine":
                # fp32 training, a head whose backward reads its activations: the three big layers on the plane kernel with f32
                # outputs (saved for the backward as before), a split pass between them
                if featp is None:
                    featp = ops.split3(feat.view(-1, 256))
                b_ = lambda k: self._f32(P, f"{name}.{idx[k]}.bias")
                h1 = ops.gemm_nt_x3(featp, self._wx3(P, f"{name}.{idx[0]}.weight", "lin"), b_(0), act=act)
                h2 = ops.gemm_nt_x3(ops.split3(h1).view(B, H, W, -1), self._wx3(P, f"{name}.{idx[1]}.weight", "c3"), b_(1), act=act, conv=1)
                h3 = ops.gemm_nt_x3(ops.split3(h2), self._wx3(P, f"{name}.{idx[2]}.weight", "lin"), b_(2), act=act)
                w4 = self._f32(P, f"{name}.{idx[3]}.weight")
                out = ops.head_out_fwd(h3, w4.reshape(w4.shape[0], -1), b_(3), B, H, W, _ACT[lay["final"]])
                outs.append(out)
                heads_saved.append(dict(h1=h1, h2=h2, h3=h3, out=out, x3=True))
                del h1, h2, h3
                continue
            if lowres:
                h1l = ops.gemm_nt(path.view(-1, 256), self._w(P, f"{name}.{idx[0]}.weight", "lin"), self._f32(P, f"{name}.{idx[0]}.bias"))
                h1 = ops.bilinear_fwd(h1l.view(path.shape[0], path.shape[1], path.shape[2], -1), H, W, True, relu=lay["relu"]).view(B * H * W, -1)
                del h1l
            else:
                h1 = ops.gemm_nt(feat.view(-1, 256), self._w(P, f"{name}.{idx[0]}.weight", "lin"), self._f32(P, f"{name}.{idx[0]}.bias"), act=act)
            h2 = ops.gemm_nt(h1.view(B, H, W, 512), self._w(P, f"{name}.{idx[1]}.weight", "c3"), self._f32(P, f"{name}.{idx[1]}.bias"),
                             conv=1, act=act)
            w3, b3 = self._w(P, f"{name}.{idx[2]}.weight", "lin"), self._f32(P, f"{name}.{idx[2]}.bias")
            w4 = self._f32(P, f"{name}.{idx[3]}.weight")
            w4 = w4.reshape(w4.shape[0], -1)
            b4 = self._f32(P, f"{name}.{idx[3]}.bias")
            if _FUSE_HEAD_OUT and ops.gemm_nt(h2, w3, b3, act=act, query_rowreduce=True):
                # the 1024 -> {1,2} output layer rides in the epilogue of the GEMM that produces its input (one read of
                # h3 saved); without saved activations (inference) h3 is not written at all
                h3, parts = ops.gemm_nt(h2, w3, b3, act=act, red_w=w4.contiguous(), no_store=not keep)
                out = ops.head_out_finish(parts, b4, B, H, W, _ACT[lay["final"]])
                # sin is not invertible from its value: the backward pass of the 'sine' variant gets the pre-activation
                zpre = ops.head_out_finish(parts, b4, B, H, W, L.ACT_NONE) if (save and lay["final"] == "sine") else None
                del parts
            else:
                h3 = ops.gemm_nt(h2, w3, b3, act=act)
                out = ops.head_out_fwd(h3, w4, b4, B, H, W, _ACT[lay["final"]])
                zpre = ops.head_out_fwd(h3, w4, b4, B, H, W, L.ACT_NONE) if (save and lay["final"] == "sine") else None
            outs.append(out)
            if algebraic:
                heads_saved.append(dict(algebraic=True, act=_ACT[lay["final"]], out=out))
            elif save:
                heads_saved.append(dict(h1=h1, h2=h2, h3=h3, out=(zpre if zpre is not None else out)))
            del h1, h2, h3
        if save:
            S["heads"] = heads_saved
        return outs[0], outs[1], S

    def _heads_backward_fullres(self, P, S, d_center, d_sdf, G, wgrad_lin, wgrad_c3):
        """backward of both heads on the interpolated feature map (the reference's order of operations); returns the gradient of the
        map before the final resize"""
        dt = self.dt
        B, H, W = S["B"], S["H"], S["W"]
        dev = d_center.device
        dfeat = None
        feat = S["feat"]
        # Both heads factored: their layer-1 input gradients dh1 go side by side into one [M, 2*C1] buffer and the gradient of
        # the shared feature map is ONE GEMM over K = 2*C1 (instead of a GEMM plus a second one that re-reads and re-writes
        # the [M, 256] result to accumulate into it).
        merge_dfeat = _MERGE_DFEAT and all(not (hs_.get("collapsed") or hs_.get("algebraic")) for hs_ in S["heads"])
        dh1cat, w1cat = None, []
        for hi, (name, lay, dout) in enumerate((("center_field_prediction_head", self.center_layout, d_center),
                                               ("sdf_prediction_head", self.sdf_layout, d_sdf))):
            hs = S["heads"][hi]
            idx = lay["conv_idx"]
            if hs.get("collapsed") or hs.get("algebraic"):
                dfeat = self._linear_head_backward(P, name, idx, feat, hs, dout, dfeat, G)
                continue
            relu = lay["relu"]
            w4 = self._f32(P, f"{name}.{idx[3]}.weight")
            dh3 = ops.head_out_bwd(hs["h3"], w4.reshape(w4.shape[0], -1), dout.contiguous(), hs["out"], _ACT[lay["final"]], relu,
                                   G[f"{name}.{idx[3]}.weight"].view(w4.shape[0], -1), G[f"{name}.{idx[3]}.bias"])
            hs["h3"] = None
            wgrad_lin(f"{name}.{idx[2]}.weight", dh3, hs["h2"], f"{name}.{idx[2]}.bias")
            x3b = bool(hs.get("x3")) and dt == torch.float32
            if x3b:
                dh2 = ops.gemm_nt_x3(ops.split3(dh3), self._wx3(P, f"{name}.{idx[2]}.weight", "lin_t"), None, mask=(hs["h2"] if relu else None))
            else:
                dh2 = ops.gemm_nt(dh3, self._w(P, f"{name}.{idx[2]}.weight", "lin_t"), None, aux=(hs["h2"] if relu else None), mask_relu=relu)
            del dh3
            h1 = hs["h1"].view(B, H, W, 512)
            wgrad_c3(f"{name}.{idx[1]}.weight", dh2, h1, f"{name}.{idx[1]}.bias")
            hs["h2"] = None
            c1 = hs["h1"].shape[-1]
            if merge_dfeat and dh1cat is None:
                dh1cat = torch.empty((B * H * W, 2 * c1), dtype=dt, device=dev)
            if x3b and not merge_dfeat:
                dh1 = ops.gemm_nt_x3(ops.split3(dh2).view(B, H, W, -1), self._wx3(P, f"{name}.{idx[1]}.weight", "c3_d"), None, conv=1,
                                     mask=(hs["h1"] if relu else None))
            else:
                dh1 = ops.gemm_nt(dh2.view(B, H, W, 512), self._w(P, f"{name}.{idx[1]}.weight", "c3_d"), None, conv=1,
                                  aux=(hs["h1"] if relu else None), mask_relu=relu,
                                  out=(dh1cat[:, hi * c1:(hi + 1) * c1] if merge_dfeat else None))
            del dh2
            hs["h1"] = None
            wgrad_lin(f"{name}.{idx[0]}.weight", dh1, feat.view(-1, 256), f"{name}.{idx[0]}.bias")
            if merge_dfeat:
                w1cat.append(self._w(P, f"{name}.{idx[0]}.weight", "lin_t"))
            elif dfeat is None:
                dfeat = ops.gemm_nt(dh1, self._w(P, f"{name}.{idx[0]}.weight", "lin_t"), None)
            else:
                ops.gemm_nt(dh1, self._w(P, f"{name}.{idx[0]}.weight", "lin_t"), None, aux=dfeat, out=dfeat)
            del dh1
        if merge_dfeat:
            dfeat = ops.gemm_nt(dh1cat, torch.cat(w1cat, dim=1), None)
            del dh1cat
        S["feat"] = None
        ph, pw = S["path1_hw"]
        dpath = ops.bilinear_bwd(dfeat.view(B, H, W, 256), ph, pw, True)
        del dfeat
        return dpath

    def _heads_backward_lowres(self, P, S, d_center, d_sdf, G, wgrad_lin, wgrad_c3):
        """Backward of both heads when their first layer ran before the final resize (_COMMUTE_RESIZE).  The gradient of that layer's
        OUTPUT goes through the resize's adjoint -- the 512 channels of a factored head, the 16-channel map of shifted output
        gradients of an algebraic head (csrc/linear_head.hip, lh_shift9_kernel) -- side by side into ONE [Ml, K] buffer on the small
        map; the layer's weight gradients, the algebraic head's tap reductions and the gradient of the small map (one GEMM over K)
        are taken there.  Returns that gradient [nb, ph, pw, 256]."""
        dt = self.dt
        B, H, W = S["B"], S["H"], S["W"]
        path = S["path"]
        nb, ph, pw, C = path.shape
        pl = path.view(-1, C)
        Ml = pl.shape[0]
        dev = path.device
        widths = [64 if hs_.get("algebraic") else hs_["h1"].shape[-1] for hs_ in S["heads"]]
        K = sum(widths)
        dlow = torch.empty((Ml, K), dtype=dt, device=dev)
        bcat = torch.zeros((C, K), dtype=dt, device=dev)          # [256, K]: dpath = dlow . bcat^T
        c0 = 0
        for hi, (name, lay, dout) in enumerate((("center_field_prediction_head", self.center_layout, d_center),
                                               ("sdf_prediction_head", self.sdf_layout, d_sdf))):
            hs = S["heads"][hi]
            idx = lay["conv_idx"]
            if hs.get("algebraic"):     # (also a head whose FORWARD ran collapsed: _linear_head_forward_lowres)
                if "u" not in hs:
                    hs["u"], hs["Vc"], hs["Kw"], _ = self._linear_head_weights(P, name, idx, dev)
                s9, nd = ops.linear_head_shift9(dout.contiguous(), hs["out"], hs["act"], dt)
                ops.bilinear_bwd(s9, ph, pw, True, out=dlow[:, c0:c0 + 16].unflatten(0, (nb, ph, pw)))
                del s9
                dlow[:, c0 + 16:c0 + 64].zero_()
                gm = ops.gemm_tn(dlow[:, c0:c0 + 16], pl)           # [16, 256]: G[t][c] = sum_q (U^T s9)[q][t] path(q)[c]
                self._linear_head_algebra(P, name, idx, hs, gm[:9].reshape(-1), nd[:9], nd[9:10], G)
                bcat[:, c0:c0 + 9].copy_(hs["Kw"].view(9, C).t())
                c0 += 64
                continue
            relu = lay["relu"]
            w4 = self._f32(P, f"{name}.{idx[3]}.weight")
            dh3 = ops.head_out_bwd(hs["h3"], w4.reshape(w4.shape[0], -1), dout.contiguous(), hs["out"], _ACT[lay["final"]], relu,
                                   G[f"{name}.{idx[3]}.weight"].view(w4.shape[0], -1), G[f"{name}.{idx[3]}.bias"])
            hs["h3"] = None
            wgrad_lin(f"{name}.{idx[2]}.weight", dh3, hs["h2"], f"{name}.{idx[2]}.bias")
            dh2 = ops.gemm_nt(dh3, self._w(P, f"{name}.{idx[2]}.weight", "lin_t"), None, aux=(hs["h2"] if relu else None), mask_relu=relu)
            del dh3
            c1 = hs["h1"].shape[-1]
            wgrad_c3(f"{name}.{idx[1]}.weight", dh2, hs["h1"].view(B, H, W, c1), f"{name}.{idx[1]}.bias")
            hs["h2"] = None
            dh1 = ops.gemm_nt(dh2.view(B, H, W, -1), self._w(P, f"{name}.{idx[1]}.weight", "c3_d"), None, conv=1,
                              aux=(hs["h1"] if relu else None), mask_relu=relu)
            del dh2
            hs["h1"] = None
            ops.bilinear_bwd(dh1.view(B, H, W, c1), ph, pw, True, out=dlow[:, c0:c0 + c1].unflatten(0, (nb, ph, pw)))
            del dh1
            wgrad_lin(f"{name}.{idx[0]}.weight", dlow[:, c0:c0 + c1], pl, f"{name}.{idx[0]}.bias")
            bcat[:, c0:c0 + c1].copy_(self._w(P, f"{name}.{idx[0]}.weight", "lin_t"))
            c0 += c1
        dpath = ops.gemm_nt(dlow, bcat, None).view(nb, ph, pw, C)
        S["path"] = None
        S["feat"] = None
        return dpath

    # ------------------------------------------------------------------ backward
    def backward(self, P, S, d_center, d_sdf, G, stage_cb=None, join_at_stages=False):
        """G: dict name -> preallocated fp32 gradient tensor (parameter shape) to fill.
        Parameters that receive no gradient are left untouched.
        stage_cb(name, wg) is called when a stage's gradients are complete or enqueued behind wg (the WgradStream of this pass):
        the data-parallel exchange launches its bucket there, a single-GPU step enqueues the stage's Adam update behind wg;
        join_at_stages: the caller reads the gradients inside stage_cb (so the weight-gradient stream is joined before each call)."""
        cfg, dt = self.cfg, self.dt
        if S.get("x3"):
            return self.backward_x3(P, S, d_center, d_sdf, G, stage_cb, join_at_stages)
        B, H, W, gh, gw = S["B"], S["H"], S["W"], S["gh"], S["gw"]
        D, heads, p = cfg["D"], cfg["heads"], cfg["patch"]
        g, Nt = gh * gw, gh * gw + 1
        dev = d_center.device
        wg = WgradStream(dev, WgradStream.wanted(B * H * W))
        # LayerNorm's dgamma / dbeta are weight gradients too: in a chain-of-graphs step their reduction pass leaves the data-gradient chain
        # for the weight-gradient lane (48 launches of the reference recipe's step; +0.7 %).  Not in the eager two-stream schedule, whose
        # host is the slower side in backward: a hand-over costs it more than the 4-us kernel costs the GPU.
        ln_via = wg.run if (wg.staged is not None and _LN_PARAMS_SIDE) else None

        def cb(name):
            if join_at_stages:
                wg.join()
            if stage_cb is not None:
                stage_cb(name, wg)
            wg.stage_end()

        def wgrad_lin(name, dy, x, bias_name=None, **kw):
            wg.run(lambda: ops.gemm_tn(dy, x, dW=G[name].view(G[name].shape[0], -1), dbias=(G[bias_name] if bias_name else None), **kw), dy, x)

        def wgrad_c3(name, dy, x_nhwc, bias_name=None, conv=1):
            co = G[name].shape[0]

            def launch():
                dwp = ops.gemm_tn(dy.reshape(-1, co), x_nhwc, dbias=(G[bias_name] if bias_name else None), conv=conv)
                _unpack_conv3_grad(dwp, G[name])
            wg.run(launch, dy, x_nhwc)

        # ---- heads
        heads_bwd = self._heads_backward_lowres if S.get("path") is not None else self._heads_backward_fullres
        dpath = heads_bwd(P, S, d_center, d_sdf, G, wgrad_lin, wgrad_c3)
        cb("heads")

        # ---- refinenets + scratch convs
        sc = "backbone.scratch."
        d_rn = {}
        for k in (1, 2, 3, 4):
            r_ = sc + f"refinenet{k}."
            fs = S["fus"][k]
            hh, ww = fs["in_hw"]
            nb = dpath.shape[0]
            if "u" in fs:    # out_conv ran before the resize (_COMMUTE_RESIZE)
                dul = ops.bilinear_bwd(dpath, hh, ww, True).view(-1, 256)
                wgrad_lin(r_ + "out_conv.weight", dul, fs["u"].view(-1, 256), r_ + "out_conv.bias")
                du = ops.gemm_nt(dul, self._w(P, r_ + "out_conv.weight", "lin_t"), None).view(nb, hh, ww, 256)
                del dul
            else:
                wgrad_lin(r_ + "out_conv.weight", dpath.reshape(-1, 256), fs["up"].view(-1, 256), r_ + "out_conv.bias")
                dup = ops.gemm_nt(dpath.reshape(-1, 256), self._w(P, r_ + "out_conv.weight", "lin_t"), None)
                du = ops.bilinear_bwd(dup.view(nb, dpath.shape[1], dpath.shape[2], 256), hh, ww, True)
                del dup
            # RCU2: u = conv2(relu(conv1(relu(s)))) + s
            wgrad_c3(r_ + "resConfUnit2.conv2.weight", du, fs["t2"], r_ + "resConfUnit2.conv2.bias")
            dt2 = ops.gemm_nt(du, self._w(P, r_ + "resConfUnit2.conv2.weight", "c3_d"), None, conv=1, aux=fs["t2"], mask_relu=True)
            dt2 = dt2.view(nb, hh, ww, 256)
            wgrad_c3(r_ + "resConfUnit2.conv1.weight", dt2, fs["s_relu"], r_ + "resConfUnit2.conv1.bias")
            ds = ops.gemm_nt(dt2, self._w(P, r_ + "resConfUnit2.conv1.weight", "c3_d"), None, conv=1, aux=fs["s_relu"], mask_relu=True,
                             aux2=du).view(nb, hh, ww, 256)
            del dt2, du
            if "t1" in fs:
                # s = path_prev + RCU1(x1)
                wgrad_c3(r_ + "resConfUnit1.conv2.weight", ds, fs["t1"], r_ + "resConfUnit1.conv2.bias")
                dt1 = ops.gemm_nt(ds, self._w(P, r_ + "resConfUnit1.conv2.weight", "c3_d"), None, conv=1, aux=fs["t1"], mask_relu=True)
                dt1 = dt1.view(nb, hh, ww, 256)
                wgrad_c3(r_ + "resConfUnit1.conv1.weight", dt1, fs["x1_relu"], r_ + "resConfUnit1.conv1.bias")
                dx1 = ops.gemm_nt(dt1, self._w(P, r_ + "resConfUnit1.conv1.weight", "c3_d"), None, conv=1, aux=fs["x1_relu"],
                                  mask_relu=True, aux2=ds).view(nb, hh, ww, 256)
                del dt1
                d_rn[k] = dx1
                dpath = ds  # gradient of the previous (coarser) path
            else:
                d_rn[k] = ds
            S["fus"][k] = None

        cb("refine")
        # ---- layerK_rn + reassemble + readout; token gradients collected per hook
        pp = "backbone.pretrained."
        Fs = cfg["features"]
        d_hook = [None] * 4  # (d_rpre [B*g, D], sB) applied to the token gradient when the block is reached
        for k in range(4):
            lay_in = S["rn_in"][k]
            dr = d_rn.pop(k + 1)
            wgrad_c3(sc + f"layer{k + 1}_rn.weight", dr, lay_in, None)
            dl = ops.gemm_nt(dr, self._w(P, sc + f"layer{k + 1}_rn.weight", "c3_d"), None, conv=1)
            del dr
            a = pp + f"act_postprocess{k + 1}."
            F_ = Fs[k]
            rs = S["re"][k]
            f = rs["f"]
            if k in (0, 1):
                s = 4 if k == 0 else 2
                dyu = ops.pixel_shuffle(dl.view(B, gh * s, gw * s, F_), B, gh, gw, s, F_, inverse=True)  # [B*g, s*s*F]
                brep = torch.empty(s * s * F_, dtype=torch.float32, device=dev)
                dwp = ops.gemm_tn(dyu, f, dbias=brep)  # [(i,j,co)][ci]
                gw_ = G[a + "4.weight"]  # [ci, co, s, s]
                ops.permute4(dwp, gw_, (F_, F_, s, s), (1, F_, s * F_ * F_, F_ * F_))
                ops.segsum(brep, 1, s * s, F_, 0, F_, out=G[a + "4.bias"].view(1, F_))
                df = ops.gemm_nt(dyu, self._w(P, a + "4.weight", "ct_d"), None)
                del dyu
            elif k == 2:
                df = dl.view(-1, F_)
            else:
                ho, wo = (gh - 1) // 2 + 1, (gw - 1) // 2 + 1
                wgrad_c3(a + "4.weight", dl, f.view(B, gh, gw, F_), a + "4.bias", conv=2)
                stuffed = ops.zero_stuff2(dl.view(B, ho, wo, F_), gh, gw)
                df = ops.gemm_nt(stuffed, self._w(P, a + "4.weight", "c3_d"), None, conv=1)
                del stuffed
            del dl
            wgrad_lin(a + "3.weight", df, rs["r"], a + "3.bias")
            d_rpre = ops.gemm_nt(df, self._w(P, a + "3.weight", "lin_t"), None, aux=rs["rpre"], mask_dgelu=True)  # [B*g, D]
            del df
            tok = S["acts"][k]
            gfull = G[a + "0.project.0.weight"]  # [D, 2D]
            ops.gemm_tn(d_rpre, tok, dW=gfull[:, :D], x_remap=(g, Nt, 1), M=B * g)
            sB32 = ops.segsum(d_rpre, B, g, D, g * D, D)  # [B, D] f32: sum over patches
            ops.segsum(sB32, 1, B, D, 0, D, out=G[a + "0.project.0.bias"].view(1, D))
            sBt = sB32 if dt == torch.float32 else ops.cast(sB32, dt)
            ops.gemm_tn(sBt, tok, dW=gfull[:, D:], M=B, ldx=Nt * D)
            d_hook[k] = (d_rpre, sBt, a)
            S["re"][k] = None

        cb("reassemble")

        def add_hook_grad(k, dx):
            d_rpre, sBt, a = d_hook[k]
            wname = a + "0.project.0.weight"
            w = P[wname].detach()
            wa_t = self.cache.get((wname, "lin_t_a", dt), P[wname], lambda: _pack_linear_t(w[:, :D], dt))
            wb_t = self.cache.get((wname, "lin_t_b", dt), P[wname], lambda: _pack_linear_t(w[:, D:], dt))
            if dx is None:
                dx = torch.zeros((B * Nt, D), dtype=dt, device=dev)
            ops.gemm_nt(d_rpre, wa_t, None, out=dx, aux=dx, c_remap=(g, Nt, 1))
            cls_rows = dx.view(B, Nt * D)[:, :D]  # token 0 of every image: row stride Nt*D
            ops.gemm_nt(sBt, wb_t, None, out=cls_rows, aux=cls_rows)
            d_hook[k] = None
            return dx

        # ---- transformer blocks
        m = "backbone.pretrained.model."
        dx = None
        hooks = cfg["hooks"]
        for i in range(max(hooks), -1, -1):
            if i in hooks:
                dx = add_hook_grad(hooks.index(i), dx)
            b = m + f"blocks.{i}."
            bs = S["blocks"][i]
            wgrad_lin(b + "mlp.fc2.weight", dx, bs["h"], b + "mlp.fc2.bias")
            dhp = ops.gemm_nt(dx, self._w(P, b + "mlp.fc2.weight", "lin_t"), None, aux=bs["hpre"], mask_dgelu=True)
            wgrad_lin(b + "mlp.fc1.weight", dhp, bs["ln2"], b + "mlp.fc1.bias")
            dln2 = ops.gemm_nt(dhp, self._w(P, b + "mlp.fc1.weight", "lin_t"), None)
            del dhp
            dx1 = ops.layernorm_bwd(dln2, bs["x1"], self._f32(P, b + "norm2.weight"), bs["mean2"], bs["rstd2"],
                                    G[b + "norm2.weight"], G[b + "norm2.bias"], dres=dx, params_via=ln_via)
            del dln2
            wgrad_lin(b + "attn.proj.weight", dx1, bs["att"], b + "attn.proj.bias")
            datt = ops.gemm_nt(dx1, self._w(P, b + "attn.proj.weight", "lin_t"), None)
            dqkv = ops.attention_bwd(bs["qkv"], bs["att"], datt, bs["lse"], B, Nt, heads)
            del datt
            wgrad_lin(b + "attn.qkv.weight", dqkv, bs["ln1"], b + "attn.qkv.bias")
            dln1 = ops.gemm_nt(dqkv, self._w(P, b + "attn.qkv.weight", "lin_t"), None)
            del dqkv
            dx = ops.layernorm_bwd(dln1, bs["x"], self._f32(P, b + "norm1.weight"), bs["mean1"], bs["rstd1"],
                                   G[b + "norm1.weight"], G[b + "norm1.bias"], dres=dx1, params_via=ln_via)
            del dln1, dx1
            S["blocks"][i] = None
            cb(f"block{i}")

        # ---- embeddings (vit.py:179-193)
        dpos = ops.segsum(dx, Nt, B, Nt * D, D, D)  # [Nt, D] f32, sum over images
        G[m + "cls_token"].view(-1).copy_(dpos[0])
        gpos = G[m + "pos_embed"]
        Gd = cfg["pos_grid"]
        gpos[0, 0].copy_(dpos[0])
        if (gh, gw) == (Gd, Gd):
            gpos[0, 1:].copy_(dpos[1:])
        else:
            gpos[0, 1:].copy_(ops.bilinear_bwd(dpos[1:].reshape(1, gh, gw, D).contiguous(), Gd, Gd, False).view(Gd * Gd, D))
        K = 3 * p * p
        gwp = G[m + "patch_embed.proj.weight"].view(D, K)
        patches = S["patches"]
        if patches.shape[1] == K:
            ops.gemm_tn(dx, patches, dW=gwp, dbias=G[m + "patch_embed.proj.bias"], dy_remap=(g, Nt, 1), M=B * g)
        else:
            tmp = ops.gemm_tn(dx, patches, dbias=G[m + "patch_embed.proj.bias"], dy_remap=(g, Nt, 1), M=B * g)
            gwp.copy_(tmp[:, :K])
        cb("embed")
        wg.join()
        S.clear()
