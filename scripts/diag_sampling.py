"""GPU diagnostic: does the C2 solver's DeviceGenerator draw get deferred into the closure kernel, and what does a
step cost with (a) the separate sampler kernel, (b) in-kernel sampling, (c) a resident batch."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import configs
from neurodiffeq_amd.generators import DeviceGenerator, SamplerGenerator, ResidentBatchGenerator
from neurodiffeq_amd import solvers

torch.manual_seed(0)
solver, cfg = configs.make_solver("c2", 256)
solver.fused = "require"
launches = [0]
orig = DeviceGenerator._launch
def counted(self, draw):
    launches[0] += 1
    return orig(self, draw)
DeviceGenerator._launch = counted

def run(label, k=300):
    for _ in range(20):
        solver.run_train_epoch()
    torch.cuda.synchronize()
    launches[0] = 0
    t0 = time.perf_counter()
    for _ in range(k):
        solver.run_train_epoch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print(f"{label}: {dt * 1e6:.2f} us/step, sampler launches per step {launches[0] / k:.2f}", flush=True)

solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(cfg["gen"], 8, "cuda"))
run("resident")
run("resident")
solver.generator["train"] = SamplerGenerator(DeviceGenerator(cfg["gen"], seed=2))
run("device, in-kernel allowed")
defer = solvers.BaseSolver._maybe_defer_sampling
solvers.BaseSolver._maybe_defer_sampling = lambda self, key: None
run("device, separate sampler kernel")
solvers.BaseSolver._maybe_defer_sampling = defer
run("device, in-kernel allowed")
g = solver.generator["train"].generator
print("pending", g.pending, "draw", g.draw, "loss", solver.metrics_history["train_loss"][-1] if solver.metrics_history["train_loss"] else None)
