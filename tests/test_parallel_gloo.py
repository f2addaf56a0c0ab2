"""world_size-2 gloo test (CPU) of the data-parallel contract: shard bounds, GLOBAL-N normalisation and the single
fused all-reduce reproduce the unsharded loss and gradient.  Per-shard compute here is the oracle closure (tests
only) standing in for the HIP kernels, which have their own parity tests."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from neurodiffeq_amd.parallel import BatchSharding


class _FP:
    def __init__(self, g):
        self.grad = g


class _System:
    def __init__(self, grads, n_slots):
        self.flat = [_FP(g) for g in grads]
        self.loss_buf = torch.zeros(n_slots, dtype=torch.float32)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import autograd_ref as R
    torch.manual_seed(0)
    cfg = R.build_config("c2", 12)
    torch.manual_seed(1)
    coords = cfg["sampler"]()
    n = coords[0].numel()
    sh = BatchSharding()
    lo, hi = sh.bounds(n)
    # shard closure with GLOBAL normalisation: loss_shard = sum r^2 / (N * n_eq)
    batch = [c[lo:hi].reshape(-1, 1).requires_grad_(True) for c in coords]
    funcs = [e(net, *batch) for net, e in zip(cfg["nets"], cfg["enforcers"])]
    res = torch.cat(cfg["pde"](*funcs, *batch), dim=1)
    loss = (res ** 2).sum() / (n * res.shape[1])
    loss.backward()
    sys_ = _System([R.get_flat_grad(cfg["nets"]).clone()], 1)
    sys_.loss_buf[0] = loss.detach()
    sh.all_reduce(sys_, 1, train=True)
    if rank == 0:
        for net in cfg["nets"]:
            net.zero_grad()
        full = R.closure(cfg["nets"], cfg["enforcers"], cfg["pde"], coords)
        np.savez(out, grad=sys_.flat[0].grad.numpy(), loss=sys_.loss_buf.numpy(),
                 grad_full=R.get_flat_grad(cfg["nets"]).numpy(), loss_full=full["loss"].numpy(), bounds=np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_all_reduce_equals_unsharded(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    assert np.allclose(r["loss"][0], r["loss_full"], rtol=1e-5)
    assert np.linalg.norm(r["grad"] - r["grad_full"]) <= 1e-5 * np.linalg.norm(r["grad_full"])


def _worker_presharded(rank, world, port, out):
    """Weak scaling as bench.py runs it: every rank owns a batch of its own (presharded), seeds are normalised by the
    GLOBAL point count world * n; the reduced [gradient | loss] equals one closure over the concatenated batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import autograd_ref as R
    torch.manual_seed(0)
    cfg = R.build_config("c2", 8)
    shards = []
    for r in range(world):
        torch.manual_seed(10 + r)
        shards.append(cfg["sampler"]())
    mine = shards[rank]
    n = mine[0].numel()
    sh = BatchSharding(presharded=True)
    assert sh.bounds(n) == (0, n) and sh.global_n(n) == world * n and sh.direct("cpu") is None
    batch = [c.reshape(-1, 1).requires_grad_(True) for c in mine]
    funcs = [e(net, *batch) for net, e in zip(cfg["nets"], cfg["enforcers"])]
    res = torch.cat(cfg["pde"](*funcs, *batch), dim=1)
    loss = (res ** 2).sum() / (sh.global_n(n) * res.shape[1])
    loss.backward()
    flat = torch.cat([R.get_flat_grad(cfg["nets"]), loss.detach().reshape(1)])     # [grad | loss]: ONE message
    sh.all_reduce_flat(flat)
    if rank == 0:
        for net in cfg["nets"]:
            net.zero_grad()
        whole = [torch.cat([s[i] for s in shards]) for i in range(len(mine))]
        full = R.closure(cfg["nets"], cfg["enforcers"], cfg["pde"], whole)
        np.savez(out, flat=flat.numpy(), grad_full=R.get_flat_grad(cfg["nets"]).numpy(), loss_full=full["loss"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_presharded_weak_scaling_equals_one_big_batch(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "res2.npz")
    mp.spawn(_worker_presharded, args=(2, port, out), nprocs=2, join=True)
    r = np.load(out)
    assert np.allclose(r["flat"][-1], r["loss_full"], rtol=1e-5)
    assert np.linalg.norm(r["flat"][:-1] - r["grad_full"]) <= 1e-5 * np.linalg.norm(r["grad_full"])


def test_bounds_partition():
    for n in (1, 7, 64, 65536, 1000):
        for world in (1, 2, 3, 8):
            edges = [BatchSharding(rank=r, world_size=world).bounds(n) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_sharding_without_a_process_group_has_no_native_all_reduce():
    """a BatchSharding built from an explicit rank / world size with no process group initialised answers
    "torch.distributed" / None instead of raising out of dist.get_backend()."""
    assert not dist.is_initialized()
    sh = BatchSharding(rank=0, world_size=2)
    assert sh.direct("cpu") is None and sh.direct("cuda") is None
    assert sh.allreduce_kind("cuda") == "torch.distributed"


def _agree_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = BatchSharding()
    res = [sh.agree(True, "cpu"), sh.agree(rank == 0, "cpu"), sh.agree(False, "cpu")]
    if rank == 1:
        np.save(out, np.array(res))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_a_flag_only_if_all_do(tmp_path):
    """BatchSharding.agree (the closure kernel's self-check outcome must send every rank down the same launch path):
    true only when true everywhere."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "agree.npy")
    mp.spawn(_agree_worker, args=(2, port, out), nprocs=2, join=True)
    assert np.load(out).tolist() == [True, False, False]
