"""Data-parallel gradient exchange: one process per GPU, RCCL (backend "nccl" on ROCm)
over xGMI.  The reference has no data parallelism on this path (single process,
train_objectness_net.py:836); this is new functionality defined by the north-star
(SURVEY.md section 8e): shard the image batch, all-reduce the flat gradient buffer in
buckets laid out in backward-completion order so the exchange overlaps the rest of
backward, then scale by 1/world inside the Adam kernel.

Device-agnostic on purpose (tested with gloo on CPU, world_size 2)."""
import time

import torch
import torch.distributed as dist


class BucketedAllReduce:
    def __init__(self, flat, boundaries, group=None):
        """flat: 1-D gradient buffer; boundaries: ascending element offsets [0, ..., flat.numel()];
        bucket k = flat[boundaries[k]:boundaries[k+1]] becomes ready when `ready(k)` is called."""
        assert flat.dim() == 1 and boundaries[0] == 0 and boundaries[-1] == flat.numel()
        self.flat, self.bounds, self.group = flat, list(boundaries), group
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.enabled else 1
        self._pending = []
        # trace (bench.py switches it on for ONE untimed step when world > 1): per bucket, when its all-reduce was issued behind the
        # backward kernels that produce it and when it completed, on the device's timeline (events) for GPU buffers, on the host
        # clock for CPU buffers (the gloo rehearsal) -- the first 8-GPU record shows the overlap with backward directly
        self.trace = False
        self._trace = []
        self._t0 = None
        self._probe = None

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
        if not self.trace:
            self._pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
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
            work = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            ev_done = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self._probe):   # a side stream waits for the collective and stamps its completion: the compute
                work.wait()                        # stream is not held up
                ev_done.record()
            self._trace.append([k, (hi - lo) * self.flat.element_size(), ev_issue, ev_done])
        else:
            t_issue = time.perf_counter()
            work = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._trace.append([k, (hi - lo) * self.flat.element_size(), t_issue, None, work])
        self._pending.append(work)

    def finish(self):
        """Wait for every launched bucket; returns the scale (1/world) the optimizer must apply."""
        if self.trace and self._t0 is not None:
            if self.flat.is_cuda:
                self._end = torch.cuda.Event(enable_timing=True)
                self._end.record()                 # the compute stream has enqueued all of backward
            else:
                self._end = time.perf_counter()
        for w in self._pending:
            w.wait()
        if self.trace and not self.flat.is_cuda:
            for e in self._trace:
                if e[3] is None:
                    e[4].wait()
                    e[3] = time.perf_counter()
        self._pending = []
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
