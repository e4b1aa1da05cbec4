"""HIP-graph capture of the engine's launch lists.

`engine.Engine.forward / backward` are static lists of libumr launches for a given input shape: the library neither allocates
nor synchronises, every launch goes to torch's current stream, and every temporary is a torch tensor.  Capturing such a list
on a stream (torch.cuda.graph: HIP stream capture + a private allocator pool for the temporaries) and replaying it removes the
host side of ~1000 launches per step.  That host side is what bounds the reference's own regime -- 128x128 crops, batch 20 /
50 (README.md:148-155, object_reasoning.py:301-337): ~10 us per launch back to back against kernels of a few microseconds.

Rules the captured region keeps:
  * no host read of device data, no synchronisation, no event with timing (ops.set_kernel_timer is refused while capturing);
  * per-call scalars that change between replays live in device memory (Adam's step count / learning rate: ops.adam_set_hyper);
  * inputs are copied into buffers owned by the capture; outputs are buffers owned by the capture (callers get clones);
  * the packed weight copies (engine.PackCache) must all exist before the capture starts (two eager warm-up calls), and the
    capture is dropped when the cache's generation changes (a parameter was reassigned / reloaded and re-packed elsewhere).
A capture that fails for any reason leaves the caller on the eager path (`Captured.failed`).
One capture at a time per process: the "capture in progress" state below is module-global, as the reference's loops are
single-threaded (train_objectness_net.py, object_reasoning.py; DataLoader workers are processes that never touch the model)."""
import os
import warnings

import torch

_state = {"capturing": False, "store": None, "forks": [], "staged": None, "lane": None}

# "auto": capture small problems (their step is launch-bound), run large ones eagerly; "on" / "off" force it.  Environment
# default for every net / TrainStep that is not told otherwise.
DEFAULT_MODE = os.environ.get("UMR_GRAPHS", "auto")
# auto threshold, in output pixels per call (B*H*W): the reference's regime (20 x 128^2 = 0.33 M, 50 x 128^2 = 0.82 M) is below,
# the 384^2 / 518^2 batches (9.4 M / 4.3 M: < 0.1 % of the step is launch gaps) are above
AUTO_MAX_PIXELS = int(os.environ.get("UMR_GRAPHS_AUTO_MAX_PIXELS", str(1 << 20)))
WARMUP_CALLS = 2
# train steps of small problems: capture as a chain of per-stage graphs replayed on two streams (StagedCaptured); 0 = one graph whose
# weight-gradient branches are fork / join edges (the round-4 form: the runtime executes them one after the other)
STAGED = os.environ.get("UMR_GRAPH_STAGED", "1") != "0"
MAX_CAPTURES = 8     # per net (inference): shapes x streams


def capturing():
    return _state["capturing"]


def capture_store():
    """scratch buffers that belong to the capture in progress (ops._workspace): kept alive with the Captured object"""
    return _state["store"]


def staged():
    """the StagedCaptured whose capture is in progress (engine.WgradStream defers its side-stream work to it), or None"""
    return _state["staged"]


def lane():
    """'main' / 'side' inside a staged capture (its segments are all recorded on ONE capture stream, but replayed on two: scratch
    buffers are keyed by lane as well as by stream, ops._workspace); None otherwise"""
    return _state["lane"]


def note_fork(main, side):
    """a side stream was forked into the capture in progress (engine.WgradStream): if the captured call raises before its own join,
    Captured joins it back, so that ending the capture is legal and the side stream leaves capture mode"""
    if _state["capturing"] and all(side is not s for _, s in _state["forks"]):
        _state["forks"].append((main, side))


def wanted(mode, pixels, train=False, two_streams=True):
    """'auto': small problems are captured -- inference calls as one graph; train steps as a chain of per-stage graphs replayed on two
    streams (StagedCaptured), where the eager schedule would put the weight gradients on a second stream (`two_streams`:
    engine.WgradStream.wanted).  Measured on the reference recipe (dpt_large, 20 x 128^2, bf16): eager one
    stream 794 images/s, one graph 797 (GPU-bound: the step's kernels add up to the step time), eager two streams 873 (host-bound
    in backward: ~21 launches per transformer block), ONE graph with fork / join edges 779-806 (the runtime executes the branches
    one after the other), chain of per-stage graphs on two streams 904.  'on' captures any size, 'off' nothing."""
    mode = mode or DEFAULT_MODE
    if mode == "on":
        return True
    if mode == "off":
        return False
    if train:
        return STAGED and two_streams and pixels <= AUTO_MAX_PIXELS
    return pixels <= AUTO_MAX_PIXELS


def release_dropped():
    """after captures were dropped: their private pools only return to the allocator's cache once the graph objects are gone (collect
    reference cycles first), and a NEW capture's pool never draws from that cache -- without handing the cached blocks back to the
    driver the reserved memory of a loop that keeps meeting new shapes grows by a pool per capture (measured: +33 GiB per 14 shapes)"""
    import gc
    gc.collect()
    torch.cuda.empty_cache()


class Captured:
    """One captured call: `fn(*static_inputs)` -> tuple of output tensors, recorded once, replayed with fresh input values."""

    def __init__(self, fn, example_inputs, generation_of=None, on_fail=None):
        """on_fail(store): called when the capture raised -- whatever the capture built for later use (packed weight copies in
        engine.PackCache) was only RECORDED, never executed, and must not be served to the eager path that takes over"""
        self.failed = None
        self.graph = None
        self._gen_of = generation_of
        dev = example_inputs[0].device
        self.static_in = [torch.empty_like(t) for t in example_inputs]
        for s, t in zip(self.static_in, example_inputs):
            s.copy_(t)
        torch.cuda.synchronize(dev)          # everything enqueued before the capture is complete: no cross-capture dependencies
        g = torch.cuda.CUDAGraph()
        self.store = {}
        _state["capturing"], _state["store"], _state["forks"] = True, self.store, []
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # (see StagedCaptured._begin)
                try:
                    outs = fn(*self.static_in)
                except BaseException:
                    # a capture may only end once every stream forked into it has rejoined its origin; a call that raised in
                    # mid-backward has not joined its weight-gradient stream -- without this the runtime refuses to end the capture
                    # and leaves the side stream in capture mode, where the eager fallback's next launch on it fails
                    for main, side in _state["forks"]:
                        main.wait_stream(side)
                    raise
            self.graph, self.outs = g, tuple(outs)
        except Exception as e:   # noqa: BLE001 -- whatever went wrong, the eager path still works
            self.failed = f"{type(e).__name__}: {e}"
            warnings.warn(f"unmore_amd: HIP-graph capture failed, staying on the eager path ({self.failed})")
            if on_fail is not None:
                on_fail(self.store)
        finally:
            _state["capturing"], _state["store"], _state["forks"] = False, None, []
        self.generation = generation_of(self.store) if generation_of else None
        torch.cuda.synchronize(dev)

    def valid(self):
        return self.graph is not None and (self._gen_of is None or self._gen_of(self.store) == self.generation)

    def replay(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            s.copy_(t)
        self.graph.replay()
        return self.outs


class _HostCall:
    def __init__(self, fn):
        self.fn = fn


class StagedCaptured:
    """A train step of a small problem captured as a CHAIN of graphs instead of one graph with forks.

    Why: in the reference's regime (128x128 crops, batch 20: README.md:148-155, train_objectness_net.py:783-788) the backward pass
    enqueues ~21 launches per transformer block (11 on the data-gradient chain, the weight gradients + Adam + weight refresh beside
    it) and the host needs about as long for them as the GPU needs to run the chain: the eager two-stream schedule starves the
    main stream (profiles/r05_ref_step_anatomy.txt: ~100 us of idle main stream per block boundary, 4-5 ms per 23-ms step), while
    ONE captured graph with fork / join edges costs the host nothing but is executed branch after branch by the runtime (measured
    in round 4: 790 images/s against 870 eager).  So the step is cut at the stage boundaries of backward (heads, refinenets,
    reassemble, every transformer block, embeddings: the same cuts the data-parallel buckets use): main-lane graph A_k holds the
    data-gradient chain of stage k, side-lane graph B_k the stage's weight gradients, its Adam update and its weight-copy refresh
    (engine.WgradStream defers them here instead of launching them on the second stream).  A replay enqueues A_0, A_1, ... on the
    caller's stream and every B_k on a second stream behind an event recorded after A_k: B_k runs beside A_k+1.., the host's part
    is ~60 graph launches per step, and the main stream never waits for the host.

    Memory rules (the two lanes run concurrently in a replay, but are RECORDED one after the other on one capture stream):
      * the lanes allocate from separate private pools, so a temporary freed inside a side graph is only ever reused by a later
        side graph (which runs after it on the same stream), never by the main lane;
      * every main-lane tensor a deferred launch reads (`used`, the tensors the eager schedule record_stream()s) is kept alive
        until the whole capture has ended: its block is not handed to a later main-lane allocation while the side lane may
        still read it;
      * scratch buffers are per lane (ops._workspace);
      * where the eager schedule joins its second stream (WgradStream.join at the end of backward) the chain has a join marker: the
        replay makes the main stream wait for the side lane before the next main graph -- the tail of the step (Adam for the
        stages that are updated on the main lane) reads gradients the side lane wrote.
    Same kernels on the same operands in the same per-lane order as the eager schedule: bit-identical results (tests/test_graph_gpu.py)."""

    def __init__(self, fn, example_inputs, generation_of=None, on_fail=None):
        self.failed = None
        self.debug_side_delay = 0
        self.debug_main_delay = 0
        self.segments = []          # (lane, CUDAGraph) in issue order
        self._gen_of = generation_of
        dev = example_inputs[0].device
        self.dev = dev
        self.static_in = [torch.empty_like(t) for t in example_inputs]
        for s, t in zip(self.static_in, example_inputs):
            s.copy_(t)
        torch.cuda.synchronize(dev)
        self.store = {}
        self._pending, self._keep = [], []
        self._cur = None
        self._pools = {"main": torch.cuda.graph_pool_handle(), "side": torch.cuda.graph_pool_handle()}
        self._cap_stream = torch.cuda.Stream(device=dev)
        self._side = torch.cuda.Stream(device=dev)
        _state.update(capturing=True, store=self.store, forks=[], staged=self, lane=None)
        try:
            with torch.cuda.stream(self._cap_stream):
                try:
                    self._begin("main")
                    outs = fn(*self.static_in)
                    self.boundary()                  # anything still deferred goes into a last side graph
                    self._end()
                except BaseException:
                    if self._cur is not None:
                        self._cur[1].capture_end()   # leave capture mode before anything else touches the stream
                        self._cur = None
                    raise
            self.outs = tuple(outs)
        except Exception as e:   # noqa: BLE001 -- whatever went wrong, the eager path still works
            self.failed = f"{type(e).__name__}: {e}"
            self.segments = []
            warnings.warn(f"unmore_amd: staged HIP-graph capture failed, staying on the eager path ({self.failed})")
            if on_fail is not None:
                on_fail(self.store)
        finally:
            _state.update(capturing=False, store=None, forks=[], staged=None, lane=None)
            self._pending, self._keep = [], []       # the captures have ended: the kept tensors' blocks stay reserved in the graphs' pools
        self.generation = generation_of(self.store) if generation_of else None
        self._events = [torch.cuda.Event() for lane_, _ in self.segments if lane_ in ("side", "scall")]
        torch.cuda.synchronize(dev)

    # ---- capture side (called through engine.WgradStream while fn runs)
    def _begin(self, lane_):
        g = torch.cuda.CUDAGraph()
        # (thread_local: only THIS thread's calls are policed while the stream records -- a data-parallel process has RCCL's watchdog
        # thread polling its work events beside the capture)
        g.capture_begin(pool=self._pools[lane_], capture_error_mode="thread_local")
        self._cur = (lane_, g, _launches[0])
        _state["lane"] = lane_

    def _end(self):
        lane_, g, n0 = self._cur
        # a cut right after a cut records nothing: torch warns about the empty graph, and that warning -- not the count of libumr
        # launches -- is what says a segment may be dropped: a segment holding only torch kernels (.copy_ / .zero_ / torch.zeros of
        # engine.backward) is work too and is replayed like any other
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            g.capture_end()
        empty = _launches[0] == n0 and any("empty" in str(w.message).lower() for w in caught)
        for w in caught:
            if "empty" not in str(w.message).lower():
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        self._cur = None
        _state["lane"] = None
        if not empty:
            self.segments.append((lane_, g))

    def defer(self, fn, used):
        self._pending.append(fn)
        self._keep.extend(t for t in used if t is not None)

    def defer_host(self, fn):
        """a HOST action that belongs behind the side-lane work deferred so far -- a collective on the gradient bucket those launches
        complete (data-parallel steps: trainer.TrainStep, parallel.BucketedAllReduce.ready).  Collectives are not captured: the chain is
        cut here, and a replay calls fn() between two graph launches with the SIDE stream current, so the collective's own stream
        orders itself behind the stage's weight gradients and the main lane never waits for it."""
        self._pending.append(_HostCall(fn))

    def host_call(self, fn):
        """a host action on the MAIN lane between two graphs of the chain (BucketedAllReduce.finish: the main stream waits for every
        collective before the optimizer update)"""
        self.boundary()
        self._end()
        self.segments.append(("call", fn))
        self._begin("main")

    def boundary(self):
        """end of a stage of backward: what was deferred since the last boundary becomes the stage's side graph (host actions deferred
        among it cut that graph: launches before -- graph -- host call -- launches after)"""
        if not self._pending:
            return
        self._end()
        pend, self._pending = self._pending, []
        self._begin("side")
        for fn in pend:
            if isinstance(fn, _HostCall):
                self._end()
                self.segments.append(("scall", fn.fn))
                self._begin("side")
            else:
                fn()
        self._end()
        self._begin("main")

    def join(self):
        """the caller's join of its weight-gradient stream (engine.WgradStream.join at the end of backward: what follows on the main
        lane -- the Adam update and weight refresh of the stages that are not updated on the side lane, e.g. the reassemble stage --
        READS what the side lane wrote): flush what is pending, cut the main lane here, and make the replay wait for the side lane
        before it enqueues the next main graph"""
        self.boundary()
        self._end()
        self.segments.append(("join", None))
        self._begin("main")

    # ---- replay side
    def valid(self):
        return bool(self.segments) and (self._gen_of is None or self._gen_of(self.store) == self.generation)

    def replay(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            s.copy_(t)
        main = torch.cuda.current_stream(self.dev)
        side = self._side
        k = 0
        for lane_, g in self.segments:
            if lane_ == "main":
                if self.debug_main_delay:              # test hook, the mirror image of debug_side_delay: the main lane is held back before
                    torch.cuda._sleep(self.debug_main_delay)   # every segment, so the side lane's updates of stage k are long done when the
                g.replay()                             # main lane runs stage k+1.. -- which must therefore not read stage k's weights again
            elif lane_ == "join":
                main.wait_stream(side)
            elif lane_ == "call":
                g()                                    # host action with the main stream current (BucketedAllReduce.finish)
            elif lane_ == "scall":
                ev = self._events[k]                   # host action with the side stream current, behind everything the main lane has
                k += 1                                 # enqueued so far (a collective on a bucket whose last producers may be main-lane kernels)
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    g()
            else:
                ev = self._events[k]
                k += 1
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    if self.debug_side_delay:          # test hook: hold the side lane back (spin cycles) so that a missing
                        torch.cuda._sleep(self.debug_side_delay)   # dependency of the main lane on it cannot hide behind timing
                    g.replay()
        main.wait_stream(side)
        return self.outs


from ._lib import _count as _launches     # libumr launches so far (counted by _lib.check): a staged capture skips empty segments  # noqa: E402


CAPTURE_TYPES = (Captured, StagedCaptured)
