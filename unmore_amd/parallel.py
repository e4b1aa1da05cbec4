"""Data-parallel gradient exchange: one process per GPU, RCCL (backend "nccl" on ROCm)
over xGMI.  The reference has no data parallelism on this path (single process,
train_objectness_net.py:836); this is new functionality defined by the north-star
(SURVEY.md section 8e): shard the image batch, all-reduce the flat gradient buffer in
buckets laid out in backward-completion order so the exchange overlaps the rest of
backward, then scale by 1/world inside the Adam kernel.

Device-agnostic on purpose (tested with gloo on CPU, world_size 2)."""
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

    @property
    def num_buckets(self):
        return len(self.bounds) - 1

    def ready(self, k):
        """Launch the (asynchronous) all-reduce of bucket k; call in backward-completion order."""
        if not self.enabled:
            return
        lo, hi = self.bounds[k], self.bounds[k + 1]
        if hi > lo:
            self._pending.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for every launched bucket; returns the scale (1/world) the optimizer must apply."""
        for w in self._pending:
            w.wait()
        self._pending = []
        return 1.0 / self.world
