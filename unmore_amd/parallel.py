"""Data-parallel gradient exchange: one process per GPU, RCCL (backend "nccl" on ROCm)
over xGMI.  The reference has no data parallelism on this path (single process,
train_objectness_net.py:836); this is new functionality defined by the north-star
(SURVEY.md section 8e): shard the image batch, all-reduce the flat gradient buffer in
buckets laid out in backward-completion order so the exchange overlaps the rest of
backward, then scale by 1/world inside the Adam kernel.

Device-agnostic on purpose (tested with gloo on CPU, world_size 2)."""
import os
import time

import torch
import torch.distributed as dist


class BucketedAllReduce:
    def __init__(self, flat, boundaries, group=None, wire_dtype=None, force=None):
        """flat: 1-D gradient buffer; boundaries: ascending element offsets [0, ..., flat.numel()];
        bucket k = flat[boundaries[k]:boundaries[k+1]] becomes ready when `ready(k)` is called.
        wire_dtype: None = exchange the f32 gradients as they are (default).  torch.bfloat16 = exchange bf16 copies (half the bytes
        per link -- what counts where the step is short against its gradient: the reference recipe's 1.39 GB per 20-ms step, DESIGN.md
        section 6): every rank rounds its gradient, pre-scaled by 1/world, to bf16; the ranks' bf16 values are summed by the
        collective (bf16 arithmetic: one more rounding per addition); finish() widens the result back into `flat` and returns
        scale 1.0.  Error of the exchanged gradient against the f32 exchange: ~2^-9 relative per element from the rounding of the
        inputs plus <= 2^-9 per addition (tests/test_parallel_cpu.py asserts relative L2 <= 6e-3 at world 2); it is the same order
        as the bf16 step's own gradient error against fp32 (relative L2 0.02-0.03).
        force (default: env UMR_DP_FORCE=1): exchange even in a process group of ONE rank.  A builder with one GPU cannot run RCCL at
        world > 1 (it refuses two ranks on one device); at world 1 the "nccl" backend still takes every bucket through
        ProcessGroupNCCL -- its own stream ordered behind the issuing lane, asynchronous work handles, the watchdog thread beside a
        graph capture -- which is the part of the path gloo's host-side transport does not exercise (tests/test_dp_gpu.py)."""
        if force is None:
            force = os.environ.get("UMR_DP_FORCE", "0") == "1"
        assert flat.dim() == 1 and boundaries[0] == 0 and boundaries[-1] == flat.numel()
        assert wire_dtype in (None, torch.float32, torch.bfloat16)
        self.flat, self.bounds, self.group = flat, list(boundaries), group
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or bool(force))
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.wire = None
        if self.enabled and wire_dtype == torch.bfloat16:
            self.wire = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
        self._pending = {}          # bucket -> work handle of its all-reduce, in issue order
        self._sent = []
        # trace (bench.py switches it on for ONE untimed step when world > 1): per bucket, when its all-reduce was issued behind the
        # backward kernels that produce it and when it completed, on the device's timeline (events) for GPU buffers, on the host
        # clock for CPU buffers (the gloo rehearsal) -- the first 8-GPU record shows the overlap with backward directly
        self.trace = False
        self._trace = []
        self._t0 = None
        self._probe = None

    @property
    def grad_scale(self):
        """what the optimizer multiplies the exchanged gradient by: 1/world for the f32 exchange (the collective sums), 1 for the bf16
        wire (1/world is applied before the rounding)"""
        return 1.0 if self.wire is not None else 1.0 / self.world

    @property
    def num_buckets(self):
        return len(self.bounds) - 1

    def ready(self, k):
        """Launch the (asynchronous) all-reduce of bucket k; call in backward-completion order."""
        if not self.enabled:
            return
        lo, hi = self.bounds[k], self.bounds[k + 1]
        if hi <= lo:
            return
        if self.wire is not None:
            buf = self.wire[lo:hi]
            # pre-scaled by 1/world (the sum of the bf16 values stays in range), rounded to nearest even, on the current stream
            if self.flat.is_cuda:
                from . import ops                       # one HIP launch (umr_cast with a scale); the f32 buffer is left as it is
                ops.cast(self.flat[lo:hi], torch.bfloat16, scale=1.0 / self.world, out=buf)
            else:                                       # (CPU tensors: the gloo rehearsal / tests)
                buf.copy_(self.flat[lo:hi] * (1.0 / self.world))
            self._sent.append(k)
        else:
            buf = self.flat[lo:hi]
        if not self.trace:
            self._pending[k] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return
        gpu = self.flat.is_cuda
        if self._t0 is None:       # time zero: the first bucket's issue point
            if gpu:
                self._t0 = torch.cuda.Event(enable_timing=True)
                self._t0.record()
                self._probe = self._probe or torch.cuda.Stream(device=self.flat.device)
            else:
                self._t0 = time.perf_counter()
        if gpu:
            ev_issue = torch.cuda.Event(enable_timing=True)
            ev_issue.record()                      # on the compute stream: the collective's stream waits for this point
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            ev_done = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self._probe):   # a side stream waits for the collective and stamps its completion: the compute
                work.wait()                        # stream is not held up
                ev_done.record()
            self._trace.append([k, (hi - lo) * buf.element_size(), ev_issue, ev_done])
        else:
            t_issue = time.perf_counter()
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._trace.append([k, (hi - lo) * buf.element_size(), t_issue, None, work])
        self._pending[k] = work

    def wait(self, k):
        """Make the CURRENT stream wait for bucket k's exchange (bf16 wire: and widen the sums back into `flat` there): what follows on
        this stream may read the bucket's exchanged gradient -- the optimizer update of its stage, a few stages of backward later
        (trainer.TrainStep: small problems update stage by stage on the weight-gradient lane instead of after finish()).  A bucket
        that was not sent, or was waited for already, is a no-op; finish() covers whatever is left."""
        w = self._pending.pop(k, None)
        if w is None:
            return
        w.wait()
        if self.wire is not None and k in self._sent:
            lo, hi = self.bounds[k], self.bounds[k + 1]
            self.flat[lo:hi].copy_(self.wire[lo:hi])
            self._sent.remove(k)

    def finish(self):
        """Wait for every launched bucket; returns the scale (1/world) the optimizer must apply."""
        if self.trace and self._t0 is not None:
            if self.flat.is_cuda:
                self._end = torch.cuda.Event(enable_timing=True)
                self._end.record()                 # the compute stream has enqueued all of backward
            else:
                self._end = time.perf_counter()
        for w in self._pending.values():
            w.wait()
        if self.trace and not self.flat.is_cuda:
            for e in self._trace:
                if e[3] is None:
                    e[4].wait()
                    e[3] = time.perf_counter()
        self._pending = {}
        if self.wire is not None:
            for k in self._sent:       # widen the exchanged bf16 sums back into the f32 gradient buffer the optimizer reads
                lo, hi = self.bounds[k], self.bounds[k + 1]
                self.flat[lo:hi].copy_(self.wire[lo:hi])
            self._sent = []
            return 1.0                 # 1/world was applied before the rounding
        return 1.0 / self.world

    def trace_report(self):
        """-> {'buckets': [{'bucket', 'mbytes', 'issue_ms', 'done_ms'}...], 'backward_end_ms', 'exposed_ms'} of the traced step, times
        relative to the first bucket's issue; exposed_ms = how long the last all-reduce ran past the end of backward (what the step
        waits for before Adam).  Synchronises the device.  Clears the trace."""
        if not self._trace:
            return None
        gpu = self.flat.is_cuda
        if gpu:
            torch.cuda.synchronize(self.flat.device)
            rel = lambda ev: self._t0.elapsed_time(ev)
        else:
            rel = lambda t: 1e3 * (t - self._t0)
        rows = [{"bucket": e[0], "mbytes": round(e[1] / 2 ** 20, 2), "issue_ms": round(rel(e[2]), 3), "done_ms": round(rel(e[3]), 3)}
                for e in self._trace]
        end = rel(self._end)
        out = {"buckets": rows, "backward_end_ms": round(end, 3), "exposed_ms": round(max(0.0, max(r["done_ms"] for r in rows) - end), 3),
               "clock": "device events" if gpu else "host clock"}
        self._trace, self._t0 = [], None
        return out
