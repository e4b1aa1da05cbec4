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
A capture that fails for any reason leaves the caller on the eager path (`Captured.failed`)."""
import os
import warnings

import torch

_state = {"capturing": False, "store": None, "forks": []}

# "auto": capture small problems (their step is launch-bound), run large ones eagerly; "on" / "off" force it.  Environment
# default for every net / TrainStep that is not told otherwise.
DEFAULT_MODE = os.environ.get("UMR_GRAPHS", "auto")
# auto threshold, in output pixels per call (B*H*W): the reference's regime (20 x 128^2 = 0.33 M, 50 x 128^2 = 0.82 M) is below,
# the 384^2 / 518^2 batches (9.4 M / 4.3 M: < 0.1 % of the step is launch gaps) are above
AUTO_MAX_PIXELS = int(os.environ.get("UMR_GRAPHS_AUTO_MAX_PIXELS", str(1 << 20)))
WARMUP_CALLS = 2
MAX_CAPTURES = 8     # per net (inference): shapes x streams


def capturing():
    return _state["capturing"]


def capture_store():
    """scratch buffers that belong to the capture in progress (ops._workspace): kept alive with the Captured object"""
    return _state["store"]


def note_fork(main, side):
    """a side stream was forked into the capture in progress (engine.WgradStream): if the captured call raises before its own join,
    Captured joins it back, so that ending the capture is legal and the side stream leaves capture mode"""
    if _state["capturing"] and all(side is not s for _, s in _state["forks"]):
        _state["forks"].append((main, side))


def wanted(mode, pixels, train=False):
    """'auto': inference calls of small batches are captured; train steps are NOT -- measured on the reference recipe (dpt_large,
    20 x 128^2, bf16): eager 794 images/s, replay 797 (the step is GPU-bound: its 1060 kernels add up to the step time, the
    host enqueues them in 18 of the 25 ms), eager with the weight gradients on a second stream (engine.WgradStream) 852 -- and
    a replay runs the captured fork / join branches one after the other (779 with them, 789 without), so the eager two-stream
    schedule is the faster default.  'on' captures either."""
    mode = mode or DEFAULT_MODE
    if mode == "on":
        return True
    if mode == "off" or train:
        return False
    return pixels <= AUTO_MAX_PIXELS


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
            with torch.cuda.graph(g):
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
