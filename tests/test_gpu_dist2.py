"""World size 2 on ONE MI355X: two processes share cuda:0 and exchange through gloo (RCCL refuses two ranks on one
device, so the collective itself is not what is tested here) -- everything else of the data-parallel training step
is: contiguous shards of one global batch, seeds normalised by the GLOBAL point count, the native closure + local sums,
the all-reduce of [gradient | loss], the device-side tail on every rank, bit-identical replicas.  Compared with one
process training on the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _train(name, size, epochs, sharding):
    from tests import configs
    from neurodiffeq_amd.parallel import BatchSharding
    torch.manual_seed(0)
    solver, cfg = configs.make_solver(name, size)
    solver.fused = "require"
    if sharding:
        solver.dist = BatchSharding()
    torch.manual_seed(1)                      # every rank samples the same global batch (same CPU seed)
    for _ in range(epochs):
        solver.run_train_epoch()
    params = torch.cat([p.detach().reshape(-1) for n in cfg["nets"] for p in n.parameters()]).cpu().numpy()
    return np.array(solver.metrics_history["train_loss"]), params


def _worker(rank, world, port, name, size, epochs, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hist, params = _train(name, size, epochs, sharding=True)
    np.savez(out + f".{rank}.npz", hist=hist, params=params)
    dist.barrier()
    dist.destroy_process_group()


def _worker_rccl(rank, world, port, name, size, epochs, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from tests import configs
    from neurodiffeq_amd.parallel import BatchSharding
    torch.manual_seed(0)
    solver, cfg = configs.make_solver(name, size)
    solver.fused = "require"
    solver.dist = BatchSharding()
    torch.manual_seed(1)
    for _ in range(epochs):
        solver.run_train_epoch()
    direct = solver.dist.direct("cuda") is not None
    params = torch.cat([p.detach().reshape(-1) for n in cfg["nets"] for p in n.parameters()]).cpu().numpy()
    np.savez(out + ".rccl.npz", hist=np.array(solver.metrics_history["train_loss"]), params=params, direct=direct)
    solver.dist.close()
    dist.destroy_process_group()


def test_direct_rccl_communicator_world_size_one(tmp_path):
    """The communicator parallel.DirectRccl sets up through ctypes (unique id over the process group, ncclCommInitRank,
    known-answer self-test) and the all-reduce the native step enqueues with it -- at the one world size this box has."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp")
    mp.spawn(_worker_rccl, args=(1, port, "c2", 32, 4, out), nprocs=1, join=True)
    r = np.load(out + ".rccl.npz")
    assert bool(r["direct"])
    hist, params = _train("c2", 32, 4, sharding=False)
    assert np.allclose(r["hist"], hist, rtol=2e-5) and np.linalg.norm(r["params"] - params) <= 2e-5 * np.linalg.norm(params)


@pytest.mark.parametrize("name,size", [("c2", 32), ("c1", 250)])
def test_two_ranks_equal_one_process_on_the_whole_batch(tmp_path, name, size):
    epochs = 4
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp")
    mp.spawn(_worker, args=(2, port, name, size, epochs, out), nprocs=2, join=True)
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["params"], r1["params"]) and np.array_equal(r0["hist"], r1["hist"])      # replicas identical
    hist, params = _train(name, size, epochs, sharding=False)
    assert np.allclose(r0["hist"], hist, rtol=2e-5), (r0["hist"], hist)
    assert np.linalg.norm(r0["params"] - params) <= 2e-5 * np.linalg.norm(params)
