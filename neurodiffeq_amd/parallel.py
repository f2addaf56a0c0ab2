"""Data-parallel sharding of collocation batches: one process per MI355X, RCCL over xGMI via torch.distributed.

The reference is single-device.  Here every rank samples the SAME batch (same CPU seed -> bit-identical points),
processes the contiguous slice ``[r*N/R, (r+1)*N/R)`` of it through the fused kernels (whose loss / adjoint seeds are
already normalised by the GLOBAL point count), accumulates over ``n_batches`` locally like solvers.py:360-419, and
then takes part in ONE all-reduce (sum, fp32) of the flat ``[all parameter gradients | per-batch losses]`` vector
per optimizer step.  The message is 4.7 KB (C2) ... 103 KB (C5): latency-bound, so a single fused message is the
whole point -- no bucketing.  Replicas stay bit-identical because every rank applies the same optimizer step to the
same reduced gradient."""
import ctypes
import os
import warnings

import torch
import torch.distributed as dist


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]          # ncclUniqueId


class DirectRccl:
    """A RCCL communicator of our own, driven through ctypes, so that the per-step all-reduce can be enqueued by the
    native step itself (``ndq_fused_step_run``) on the compute stream: no Python/c10d dispatch, no hop onto
    ProcessGroupNCCL's side stream and back (the message is 4.7 KB -- all of its cost is such overhead).

    The library is the ``librccl.so`` torch itself has loaded; the unique id travels through the existing process
    group.  After construction every rank has verified one all-reduce against the known answer; if any rank failed,
    ALL ranks fall back to ``torch.distributed`` (the decision itself is an all-reduce over the regular group)."""

    def __init__(self, rank, world_size, group, device):
        self.ok, self.comm, self.fn, self.lib = False, None, None, None
        err = None
        uid = _UniqueId()
        # phase 1 (local): load the library, rank 0 draws the unique id.  Nothing here may skip the collectives below.
        try:
            lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
            lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
            lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
            lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
            lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            if rank == 0 and lib.ncclGetUniqueId(ctypes.byref(uid)) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
        except Exception as e:                      # noqa: BLE001
            err, lib = e, None
        # phase 2 (collective, unconditional): the id -- or None if rank 0 could not produce one -- reaches every rank
        payload = [bytes(uid) if (rank == 0 and err is None) else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(payload, src=src, group=group, device=device)
        # phase 3 (collective inside RCCL, entered only if EVERY rank can): agree first
        ready = torch.tensor([1.0 if (err is None and payload[0] is not None) else 0.0], device=device)
        dist.all_reduce(ready, op=dist.ReduceOp.MIN, group=group)
        if ready.item() == 1.0:
            try:
                ctypes.memmove(ctypes.byref(uid), payload[0], 128)
                comm = ctypes.c_void_p()
                with torch.cuda.device(device):
                    rc = lib.ncclCommInitRank(ctypes.byref(comm), world_size, uid, rank)
                    if rc != 0:
                        raise RuntimeError(f"ncclCommInitRank returned {rc}")
                    probe = torch.full((8,), float(rank + 1), dtype=torch.float32, device=device)
                    stream = torch.cuda.current_stream(device).cuda_stream
                    rc = lib.ncclAllReduce(probe.data_ptr(), probe.data_ptr(), 8, 7, 0, comm, ctypes.c_void_p(stream))
                    torch.cuda.synchronize(device)
                    want = world_size * (world_size + 1) / 2.0
                    if rc != 0 or not bool((probe == want).all()):
                        raise RuntimeError(f"self-test all-reduce failed (rc={rc}, got {probe[0].item()}, want {want})")
                self.lib, self.comm = lib, comm
                self.fn = ctypes.cast(lib.ncclAllReduce, ctypes.c_void_p).value
            except Exception as e:                  # noqa: BLE001
                err = e
        elif err is None:
            err = RuntimeError("another rank could not set up the communicator")
        # phase 4 (collective, unconditional): use it only if the self-test passed everywhere
        flag = torch.tensor([0.0 if (err is not None or self.comm is None) else 1.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        self.ok = bool(flag.item() == 1.0)
        if not self.ok and err is not None:
            warnings.warn(f"direct RCCL communicator unavailable ({err}); using torch.distributed for the all-reduce")

    def close(self):
        if self.comm is not None:
            torch.cuda.synchronize()
            self.lib.ncclCommDestroy(self.comm)
            self.comm, self.ok = None, False

    def all_reduce(self, tensor):
        stream = torch.cuda.current_stream(tensor.device).cuda_stream
        rc = self.lib.ncclAllReduce(tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), 7, 0, self.comm,
                                    ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce returned {rc}")


class BatchSharding:
    def __init__(self, rank=None, world_size=None, group=None, presharded=False):
        """presharded: every rank's generator already yields only its own shard (weak-scaling runs with resident
        batches); ``bounds`` is then the whole local batch and the global size is ``world_size`` times the local one."""
        self.group = group
        self.presharded = presharded
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world_size = dist.get_world_size(group) if world_size is None else world_size
        self._flat = None
        self._direct = None         # DirectRccl, created on first use on a GPU under the nccl backend

    def agree(self, flag, device):
        """True iff ``flag`` is true on every rank (one tiny MIN all-reduce over the regular group)."""
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item() == 1.0)

    def direct(self, device):
        """(address of ncclAllReduce, ncclComm_t) for the native step, or None (CPU / gloo / NDQ_RCCL_DIRECT=0 /
        communicator unavailable -> ``torch.distributed``)."""
        if self._direct is None:
            device = torch.device(device)
            usable = (device.type == "cuda" and dist.is_initialized() and dist.get_backend(self.group) == "nccl"
                      and os.environ.get("NDQ_RCCL_DIRECT", "1") != "0")
            self._direct = DirectRccl(self.rank, self.world_size, self.group, device) if usable else False
        d = self._direct
        return (d.fn, d.comm.value) if d and d.ok else None

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
        self._sum(flat)

    def close(self):
        """Release the direct communicator (before ``destroy_process_group``)."""
        if self._direct:
            self._direct.close()
        self._direct = False

    def _sum(self, t):
        if self.direct(t.device) is not None:
            self._direct.all_reduce(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

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
        self._sum(self._flat)
        off = 0
        for t in grads + [loss]:
            t.reshape(-1).copy_(self._flat[off:off + t.numel()])
            off += t.numel()
