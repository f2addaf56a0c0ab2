"""Data-parallel sharding of collocation batches: one process per MI355X, RCCL over xGMI via torch.distributed.

The reference is single-device.  Here every rank samples the SAME batch (same CPU seed -> bit-identical points),
processes the contiguous slice ``[r*N/R, (r+1)*N/R)`` of it through the fused kernels (whose loss / adjoint seeds are
already normalised by the GLOBAL point count), accumulates over ``n_batches`` locally like solvers.py:360-419, and
then takes part in ONE all-reduce (sum, fp32) of the flat ``[all parameter gradients | per-batch losses]`` vector
per optimizer step.  The message is 4.7 KB (C2) ... 103 KB (C5): latency-bound, so a single fused message is the
whole point -- no bucketing.  Replicas stay bit-identical because every rank applies the same optimizer step to the
same reduced gradient."""
import torch
import torch.distributed as dist


class BatchSharding:
    def __init__(self, rank=None, world_size=None, group=None, presharded=False):
        """presharded: every rank's generator already yields only its own shard (weak-scaling runs with resident
        batches); ``bounds`` is then the whole local batch and the global size is ``world_size`` times the local one."""
        self.group = group
        self.presharded = presharded
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self._flat = None

    def bounds(self, n):
        """Row range of this rank's shard of an n-point batch (contiguous, sizes differ by at most one)."""
        if self.presharded:
            return 0, n
        base, rem = divmod(n, self.world_size)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)

    def global_n(self, n):
        return n * self.world_size if self.presharded else n

    def all_reduce_flat(self, flat):
        """In-place sum of one contiguous fp32 vector ([gradient | loss] of a single-network system)."""
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce(self, system, n_batches, train=True):
        """Sum gradients (``system.flat[k].grad``) and the first ``n_batches`` loss slots over all ranks, in place."""
        grads = [fp.grad for fp in system.flat] if train else []
        loss = system.loss_buf[:n_batches]
        total = sum(g.numel() for g in grads) + n_batches
        if self._flat is None or self._flat.numel() != total or self._flat.device != loss.device:
            self._flat = torch.empty(total, dtype=torch.float32, device=loss.device)
        off = 0
        for t in grads + [loss]:
            self._flat[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for t in grads + [loss]:
            t.reshape(-1).copy_(self._flat[off:off + t.numel()])
            off += t.numel()
