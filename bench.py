#!/usr/bin/env python
"""Headline benchmark: collocation-points/sec of one training step (forward + residual + loss + backward + Adam)
of the 2D Laplace system -- BASELINE.json config C2: Solver2D, FCNN(2-32-32-1, tanh), DirichletBVP2D,
Generator2D 256x256 = 65 536 noisy-grid points per batch per GPU, fp32 -- through ``Solver2D.run_train_epoch()``.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cX] [--scaling weak|strong]

N > 1 is launched by the driver with torch.distributed.run (one rank per MI355X); scaling is WEAK by default: every rank
trains on its own 65 536-point shard of a global N*65 536-point batch and joins one all-reduce of the flat
[gradients | loss] vector per step (one-shot exchange through HIP-IPC inboxes, RCCL fallback).  A "step" is one
``run_train_epoch()`` with ``n_batches_train=1``, ``n_batches_valid=0``: two kernel launches -- the single-launch closure
kernel (forward streams + traced pointwise stage + reverse pass) and the second-stage sums / [exchange] / device-side
epoch tail (loss history, best-network snapshot, fused Adam) -- and no host synchronisation.

Timing: W untimed warm-up steps, then windows of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize()
on both sides (max over ranks), repeated until >= 0.25 s and >= 3 windows have been timed; the MEDIAN window is reported.
Inputs of the headline ``value`` are pre-sampled (reference RNG order) and resident in HBM before the timed region
(``ResidentBatchGenerator``); the figure with host sampling + PCIe upload inside the step is reported separately as
``with_host_sampling`` (like for like with the CPU baseline's full step), the one with a fresh batch drawn by the
device-side Philox sampler every step as ``with_device_sampling``.

Rank 0 prints ONE JSON line (contract in the task description) including
  roofline      -- dominant kernel (fused closure kernel): algorithmic GEMM flops / HIP-event launch time vs the fp32
                   MFMA peak, plus the HBM bytes per launch MEASURED IN THIS RUN (two child runs of this script under
                   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, outside the timed region; when rocprofv3 is missing the last
                   committed figure of profiles/traffic_c2.json is reported and labelled stale-able),
  in_fit, cold_start -- the headline's epochs issued by fit(k) (one native call); seconds to the first training step of a
                   PDE this installation has never seen (trace + hipcc + first-use self-check),
  kernels       -- the three-kernel pipeline (forward / pointwise / backward) timed the same way,
  configs       -- C1, C3, C4, C5 at their BASELINE sizes: ms per step, algorithmic TFLOP/s, fraction of the peak,
  roofline_pointwise_large -- the standalone pointwise residual kernel at 1 M / 4 M points vs the HBM roofline,
  cpu_baseline  -- the oracle's port of the reference step (torch CPU autograd) timed on this host: with sampling
                   (``value``), on a pre-sampled batch, and in fp64; ``speedup_vs_cpu_baseline`` pairs like with like.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

GRID = 256
N_POINTS = GRID * GRID
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CUs @ 2.4 GHz
HBM_PEAK_GBS = 8000.0
# algorithmic GEMM flops per point, C2: forward 10 688, adjoint = 2 x forward
FWD_FLOP_PER_PT = 2 * (32 * 2 * 1 + 32 * 32 * 5 + 32 * 1 * 5)
BWD_FLOP_PER_PT = 2 * FWD_FLOP_PER_PT


def time_launches(fn, iters=1000, warm=200):
    """Average duration (seconds) of one launch of ``fn`` on torch's current stream, by HIP events."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def fused_breakdown(system, batch):
    """Launch time of the single-launch fused closure kernel (forward + pointwise + reverse) on a resident batch."""
    from neurodiffeq_amd.engine import _c_vp, _ptr
    b, n = system.upload(batch)
    stream = _c_vp(torch.cuda.current_stream().cuda_stream)
    system.step(batch, train=True)
    fp = system.flat[0]
    seed = 1.0 / n

    def closure_only():
        b["fusedk"].lib.ndq_fused_launch(system._coord_ptr(b, 0), b["ld"], n, _ptr(fp.flat), _ptr(b["fused_partials"]),
                                           _ptr(b["fused_loss_partials"]), None, None, b["ld"], seed, 1, stream)

    def reduce_only():
        system.L.ndq_reduce_partials(_ptr(b["fused_partials"]), b["fused_blocks"], fp.numel, _ptr(fp.grad), 0, 1.0, stream)

    t, tr = time_launches(closure_only), time_launches(reduce_only)
    return dict(fused_closure=dict(us=t * 1e6, tflops=(FWD_FLOP_PER_PT + BWD_FLOP_PER_PT) * n / t / 1e12,
                                   blocks=b["fused_blocks"]),
                reduce_partials=dict(us=tr * 1e6))


def kernel_breakdown(system, batch):
    """Per-kernel launch time of the three-kernel pipeline on a resident batch -> roofline figures."""
    from neurodiffeq_amd.engine import _c_vp
    b, n = system.upload(batch)
    stream = _c_vp(torch.cuda.current_stream().cuda_stream)
    system.step(batch, train=True)
    t_fwd = time_launches(lambda: system.forward(b, n, stream))
    t_pw = time_launches(lambda: system.pointwise(b, n, stream, True, n))
    L, k, fp = system.L, 0, system.flat[0]
    d = system.descs[0]
    coords = system._coord_ptr(b, system.coord0[0])

    def bwd_only():
        L.ndq_mlp_jet_bwd(ctypes.byref(d), coords, b["ld"], n, fp.flat.data_ptr(), b["gbar"][0].data_ptr(), b["ld"],
                          b["partials"][0].data_ptr(), stream)

    def reduce_only():
        L.ndq_reduce_partials(b["partials"][0].data_ptr(), b["bwd_blocks"][0], fp.numel, fp.grad.data_ptr(), 0, 1.0, stream)

    t_bwd = time_launches(bwd_only)
    t_red = time_launches(reduce_only)
    pw_bytes = system.program.bytes_per_point(train=True) * n
    return dict(
        mlp_jet_fwd=dict(us=t_fwd * 1e6, tflops=FWD_FLOP_PER_PT * n / t_fwd / 1e12),
        pointwise=dict(us=t_pw * 1e6, gbs=pw_bytes / t_pw / 1e9, bytes_per_point=pw_bytes // n),
        mlp_jet_bwd=dict(us=t_bwd * 1e6, tflops=BWD_FLOP_PER_PT * n / t_bwd / 1e12),
        reduce_partials=dict(us=t_red * 1e6),
    )


def traffic_child(steps=300):
    """``bench.py --traffic-child``: the headline step a few hundred times, nothing else -- what the rocprofv3 --pmc
    passes of ``measure_traffic`` run over.  ``--trace-child``: 8 000 steps (0.2 s: an MI355X needs tens of ms of sustained work
    to reach its clocks -- a 300-step trace shows the kernel 10 % slower than the 40 000-launch trace under profiles/), for
    ``measure_in_situ``."""
    from neurodiffeq_amd.generators import Generator2D, ResidentBatchGenerator, SamplerGenerator
    from tests import configs
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    solver, cfg = configs.make_solver("c2", GRID)
    solver.fused = "require"
    torch.manual_seed(1)
    gen = Generator2D((GRID, GRID), (0, 0), (1, 1), "equally-spaced-noisy")
    solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(gen, 4, "cuda"))
    for _ in range(steps):
        solver.run_train_epoch()
    torch.cuda.synchronize()


def measure_traffic(timeout_s=150):
    """HBM-side bytes per launch of the headline closure kernel, measured IN THIS RUN: two rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE cannot share a pass -- MI355X_MICROARCH.md, PMC slots) over ``--traffic-child``, mean per
    dispatch of the closure kernel.  Units / correction as the guide prescribes and as calibrated on this package's
    access patterns (profiles/archive/r02/r02v_pmc_c2_summary.txt: counters in KiB, a streaming read reports exactly half its
    bytes in FETCH_SIZE, a streaming write reports WRITE_SIZE exactly).  None if rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ndq_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc",
                                "--", sys.executable, os.path.abspath(__file__), "--traffic-child"],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "fused_closure" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                        rows.append(float(row["Counter_Value"]))
            if len(rows) < 40:
                return None
            rows = rows[len(rows) // 4:]                 # skip warm-up dispatches
            vals[counter] = sum(rows) / len(rows)
        except (subprocess.TimeoutExpired, OSError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return dict(FETCH_SIZE_KiB=vals["FETCH_SIZE"], WRITE_SIZE_KiB=vals["WRITE_SIZE"],
                hbm_bytes=vals["FETCH_SIZE"] * 1024 * 2.0 + vals["WRITE_SIZE"] * 1024,
                algorithmic_bytes=N_POINTS * 8 + 256 * 1185 * 4 + 256 * 4)


def measure_in_situ(timeout_s=150):
    """Average duration of the headline closure kernel INSIDE the training step -- launched between the sums / tail kernels of
    successive epochs by run_train_epoch(), not back to back -- from a plain rocprofv3 --kernel-trace pass over
    ``--traffic-child`` (no counters: they serialise dispatches).  This is the figure the committed
    ``profiles/*_kernel_stats.md`` tables hold and what ``roofline.frac`` is priced on; the HIP-event figure of 1 000
    back-to-back launches is kept beside it (2 % kinder).  None if rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    d = tempfile.mkdtemp(prefix="ndq_trace_", dir="/tmp")
    try:
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "trace",
                            "--", sys.executable, os.path.abspath(__file__), "--trace-child"],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
        if r.returncode != 0:
            return None
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row.get("Kernel_Name", "")
                key = "closure" if "fused_closure" in name else ("tail" if "reduce_tail" in name else None)
                if key:
                    per.setdefault(key, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        if len(per.get("closure", ())) < 40:
            return None
        out = {}
        for key, spans in per.items():
            spans = sorted(spans)[len(spans) // 2:]          # the second half: clocks up, caches warm
            out[key + "_us"] = sum(e - s for s, e in spans) / len(spans) * 1e-3
            out[key + "_launches"] = len(spans)
        spans = sorted(per["closure"])[len(per["closure"]) // 2:]
        gaps = sorted(b[0] - a[0] for a, b in zip(spans[:-1], spans[1:]))
        out["step_us_median"] = gaps[len(gaps) // 2] * 1e-3      # closure start to next closure start
        return out
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cold_start():
    """Time to the first training step of a PDE this installation has never seen: trace + code generation
    + hipcc (pointwise kernel and single-launch closure kernel, built concurrently) + first-use self-check, then the
    second solver of the same system (everything cached).  The PDE carries a fresh constant so that no cache can know it."""
    import random
    from neurodiffeq_amd import diff
    from neurodiffeq_amd.conditions import NoCondition
    from neurodiffeq_amd.generators import Generator2D
    from neurodiffeq_amd.networks import FCNN
    from neurodiffeq_amd.solvers import Solver2D
    c = 1.0 + random.SystemRandom().random()
    pde = lambda u, x, y: [diff(u, x, order=2) + diff(u, y, order=2) - c * u * u]
    out = {}
    for tag in ("cold_start_s", "warm_start_s"):
        torch.manual_seed(0)
        t0 = time.perf_counter()
        s = Solver2D(pde, [NoCondition()], nets=[FCNN(2, 1, hidden_units=(32, 32))],
                     train_generator=Generator2D((32, 32), (0, 0), (1, 1)), valid_generator=Generator2D((32, 32), (0, 0), (1, 1)),
                     n_batches_valid=0)
        s.fused = "require"
        s.run_train_epoch()
        _ = s.metrics_history["train_loss"][-1]
        torch.cuda.synchronize()
        out[tag] = time.perf_counter() - t0
    out["what"] = ("Solver2D of a never-seen nonlinear Poisson problem: constructor + first run_train_epoch() incl. trace, code "
                   "generation, hipcc of the pointwise and closure kernels (concurrent), assembly fix-up, first-use self-check")
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_time(fn, min_runs, budget_s, max_runs=200):
    times, t_start = [], time.perf_counter()
    while len(times) < min_runs or (time.perf_counter() - t_start < budget_s and len(times) < max_runs):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > 3 * budget_s:
            break
    times.sort()
    return times[len(times) // 2], len(times)


def cpu_baseline(budget_s=8.0):
    """The reference's training step restated on torch CPU autograd (oracle/autograd_ref.py, pinned to the
    reference's golden vectors; kind = "port": the unmodified reference cannot travel to the GPU box), same C2 config,
    timed on this host's cores three ways: the step as BASELINE.md section 2.3 defines it (sample + fwd + diff + loss +
    bwd + Adam; ``value``), the same step on a pre-sampled batch (``presampled``: what the GPU headline times), and the
    library-default fp64 (``fp64``)."""
    from oracle import autograd_ref as R
    torch.manual_seed(0)
    cfg = R.build_config("c2", GRID)
    loop = R.TrainLoop(cfg["nets"], cfg["enforcers"], cfg["pde"], cfg["sampler"])
    # ATen's intra-op pool defaults to one thread per logical cpu, which is far past the sweet spot of these small
    # ops on a many-core host: pick the fastest thread count first (that IS the baseline a CPU user would tune to).
    ncpu = os.cpu_count() or 1
    best = None
    for t in sorted({1, 4, 8, 16, 32, 64, min(128, ncpu)}):
        if t > ncpu:
            continue
        torch.set_num_threads(t)
        loop.epoch()
        t0 = time.perf_counter()
        loop.epoch(); loop.epoch()
        dt = (time.perf_counter() - t0) / 2
        if best is None or dt < best[0]:
            best = (dt, t)
    torch.set_num_threads(best[1])
    for _ in range(2):
        loop.epoch()
    med, runs = _median_time(loop.epoch, 8, budget_s)
    fixed = cfg["sampler"]()
    pre = R.TrainLoop(cfg["nets"], cfg["enforcers"], cfg["pde"], lambda: fixed)
    pre.epoch()
    med_pre, runs_pre = _median_time(pre.epoch, 5, budget_s / 2)
    torch.manual_seed(0)
    cfg64 = R.build_config("c2", GRID, dtype=torch.float64)
    samp64 = cfg64["sampler"]
    loop64 = R.TrainLoop(cfg64["nets"], cfg64["enforcers"], cfg64["pde"], samp64)
    loop64.epoch()
    med64, runs64 = _median_time(loop64.epoch, 3, budget_s / 2)
    port = dict(value=N_POINTS / med, unit="collocation-points/s", cores=torch.get_num_threads(), kind="port",
                ms_per_step=med * 1e3, cpu_model=cpu_model(), logical_cpus=os.cpu_count(),
                presampled=dict(value=N_POINTS / med_pre, ms_per_step=med_pre * 1e3, runs=runs_pre,
                                note="same step without the generator draw: what the GPU headline `value` times"),
                fp64=dict(value=N_POINTS / med64, ms_per_step=med64 * 1e3, runs=runs64,
                          note="library default precision (neurodiffeq/__init__.py:22), with sampling"),
                sample=f"{runs} timed run_train_epoch-equivalent steps (sample+fwd+diff+loss+bwd+Adam) of the "
                       f"same C2 config, fp32, torch {torch.__version__} CPU, {os.cpu_count()} logical cpus, "
                       f"{torch.get_num_threads()} threads (fastest of 1/4/8/16/32/64/128)")
    ref = reference_baseline(budget_s)
    if ref is None:
        return port
    # the UNMODIFIED reference, timed on this box (oracle/_ref, oracle/make_ref.sh); the port stays beside it
    return dict(value=ref["value"], unit="collocation-points/s", cores=ref["threads"], kind="reference",
                ms_per_step=ref["ms_per_step"], cpu_model=cpu_model(), logical_cpus=os.cpu_count(),
                presampled=dict(ref["presampled"], note="same step on a pre-sampled batch (PredefinedGenerator): what the GPU "
                                                        "headline `value` times"),
                fp64=dict(ref["fp64"], note="library default precision (neurodiffeq/__init__.py:22), with sampling"),
                reference_revision=ref["reference_revision"],
                port=dict(value=port["value"], ms_per_step=port["ms_per_step"], cores=port["cores"],
                          presampled=port["presampled"]["value"], fp64=port["fp64"]["value"]),
                port_ratio=port["value"] / ref["value"],
                sample=f"{ref['runs']} timed Solver2D.run_train_epoch() calls of the UNMODIFIED reference (oracle/_ref = "
                       f"/root/reference/neurodiffeq, set_tensor_type('cpu', 32)): sample+fwd+diff+loss+bwd+Adam on the C2 "
                       f"config (256 x 256 noisy grid, FCNN 2-32-32-1), torch {ref['torch']} CPU, {ref['logical_cpus']} logical "
                       f"cpus, {ref['threads']} threads (fastest of 1/4/8/16/32/64/128); port_ratio = oracle port / reference")


def reference_baseline(budget_s):
    """Time the unmodified reference in a child process (importing it changes torch's global defaults); None when
    oracle/_ref is absent (oracle/make_ref.sh was never run where /root/reference exists) or the child fails."""
    import subprocess
    script = os.path.join(ROOT, "oracle", "ref_bench.py")
    if not os.path.isdir(os.path.join(ROOT, "oracle", "_ref", "neurodiffeq")):
        return None
    try:
        out = subprocess.run([sys.executable, script, "--grid", str(GRID), "--budget", str(budget_s)], capture_output=True,
                             text=True, timeout=60 * budget_s + 120, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001 -- the bench line then carries the port and says so
        print(f"bench.py: reference baseline failed ({type(e).__name__}: {e}); reporting the oracle port", file=sys.stderr)
        return None


# algorithmic GEMM flop per point of a training step, SURVEY.md 8(d) table
ALGO_FLOP_PER_PT = {"c1": 25728, "c2": 32064, "c3": 198912, "c4": 33024, "c5": 646272,
                    # networks wider than 64 units (round 4), same formula (3 x sum_l 2 in out S_l):
                    "w16": 21504,        # README.md:125 FCNN(2, 1, (512,)) on C2's problem, 5 streams
                    "w17": 52224,        # 2 -> 512 -> 3, lid-driven cavity on one network, 5 streams x 3 outputs
                    "w18": 791040,       # 2 -> 128 -> 128 -> 128 -> 1, Burgers, 4 streams
                    "w19": 1992192}      # 2 -> 256 -> 256 -> 3 (the RE100 notebook's FCNN), 5 streams x 3 outputs
WIDE_RECORDS = {"readme_laplace_512": ("w16", 256), "cavity_single_net_512x1": ("w17", 256), "burgers_128x3": ("w18", 256),
                "cavity_single_net_256x2": ("w19", 256)}


def timed_windows(step, k, barrier, reduce_max=None, min_total_s=0.25, max_windows=2000):
    """Time windows of EXACTLY ``k`` steps, each bracketed by barrier + synchronize on both sides, until at least
    ``min_total_s`` of timed work has been seen (a 20-step window of the headline is 0.5 ms: one of them is noise);
    every rank runs the same number of windows (decided from the first one, maximum over ranks).  Returns the sorted
    per-window times (seconds; max over ranks)."""
    def window():
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        barrier()
        return time.perf_counter() - t0
    first = window()
    n = int(min(max_windows, max(3, -(-min_total_s // max(first, 1e-9)))))       # >= 3: the first window of a cold box is slow
    if reduce_max is not None:
        n = int(reduce_max([float(n)])[0])
    times = [first] + [window() for _ in range(n - 1)]
    if reduce_max is not None:
        times = reduce_max(times)
    return sorted(times)


def config_record(name, size=None):
    """One other BASELINE config at its stated size through run_train_epoch() on resident pre-sampled batches:
    ms per step (median of >= 0.25 s of windows), points/s, algorithmic TFLOP/s and fraction of the fp32 MFMA peak."""
    from tests import configs
    from neurodiffeq_amd.generators import ResidentBatchGenerator, SamplerGenerator
    torch.manual_seed(0)
    solver, cfg = configs.make_solver(name, size)
    solver.fused = "require"
    torch.manual_seed(1)
    solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(cfg["gen"], 2, "cuda"))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5):
            solver.run_train_epoch()
        torch.cuda.synchronize()
    k = 5 if name in ("c5", "w18", "w19") else 50
    times = timed_windows(solver.run_train_epoch, k, torch.cuda.synchronize)
    dt_epoch = times[(len(times) - 1) // 2] / k
    # the same training epochs the way a user runs them: fit(k) -- whole chunks of epochs per native call (solvers.fit,
    # ndq_fused_fit_run), no Python between the epochs.  For the launch-bound configs this is the step time that counts.
    kf = 10 if name in ("c5", "w18", "w19") else 500
    solver.fit(kf, tqdm_file=None)
    times_fit = timed_windows(lambda: solver.fit(kf, tqdm_file=None), 1, torch.cuda.synchronize)
    dt_fit = times_fit[(len(times_fit) - 1) // 2] / kf
    dt = min(dt_epoch, dt_fit)
    n = cfg["n_points"]
    tf = ALGO_FLOP_PER_PT[name] * n / dt / 1e12
    sysm = solver._fused_sys
    return dict(points=n, ms_per_step=dt * 1e3, points_per_s=n / dt, algorithmic_flop_per_point=ALGO_FLOP_PER_PT[name],
                ms_per_step_run_train_epoch=dt_epoch * 1e3, ms_per_step_in_fit=dt_fit * 1e3,
                step_definition="one training epoch (n_batches_train = 1, no validation); ms_per_step = the faster of "
                                "run_train_epoch() in a Python loop and fit(k) (multi-epoch native call), both listed",
                algorithmic_tflops=tf, frac_of_fp32_mfma_peak=tf / FP32_MFMA_PEAK_TFLOPS, windows=len(times), steps_per_window=k,
                single_launch=sysm.fusedk is not None, launches_per_step=sysm.launches_per_step(),
                final_loss=solver.metrics_history["train_loss"][-1])


def fp64_record():
    """C2 at its stated size in the reference's DEFAULT precision (neurodiffeq/__init__.py:22: float64): fp64 networks on
    the fused path in double (the closure kernel compiled for fp64 on the f64 MFMA, fixed-order fp64 sums,
    device-side epoch tail in double), batch resident in HBM, through run_train_epoch()."""
    from tests import configs
    torch.manual_seed(0)
    solver, cfg = configs.make_solver("c2")
    for net in cfg["nets"]:
        net.double()
    solver.fused = "require"
    torch.manual_seed(1)
    batch = [c.detach().double().cuda().reshape(-1, 1) for c in cfg["gen"].get_examples()]
    solver.generator["train"].get_examples = lambda: batch
    for _ in range(20):
        solver.run_train_epoch()
    torch.cuda.synchronize()
    k = 20
    times = timed_windows(solver.run_train_epoch, k, torch.cuda.synchronize)
    dt = times[(len(times) - 1) // 2] / k
    n = cfg["n_points"]
    assert solver.fused_active and solver._fused_sys.f64
    return dict(points=n, dtype="f64", ms_per_step=dt * 1e3, points_per_s=n / dt, windows=len(times), steps_per_window=k,
                optimizer="FusedAdam on the device: ndq64_epoch_tail (history, best snapshot, Adam in double; no host synchronisation per epoch)",
                final_loss=solver.metrics_history["train_loss"][-1])


def reference_defaults_record(name="c2", size=None):
    """What importing the reference sets up -- cuda default device AND float64 default dtype (neurodiffeq/__init__.py:22,
    utils.py:10-41) -- then a plain Solver2D with its noisy 256 x 256 generator, nothing else changed: a fresh batch every
    epoch from the Philox kernel (fp32 draws handed out as doubles), the closure kernel in double, bookkeeping on the device."""
    from tests import configs
    from neurodiffeq_amd.utils import set_tensor_type
    try:
        set_tensor_type(device="cuda", float_bits=64)
        torch.manual_seed(0)
        solver, cfg = configs.make_solver(name, size)
        solver.fused = "require"
        assert next(cfg["nets"][0].parameters()).dtype == torch.float64
        for _ in range(20):
            solver.run_train_epoch()
        torch.cuda.synchronize()
        k = 20
        times = timed_windows(solver.run_train_epoch, k, torch.cuda.synchronize)
        dt = times[(len(times) - 1) // 2] / k
        n = cfg["n_points"]
        assert solver.fused_active and solver._fused_sys.f64
        return dict(points=n, dtype="f64", ms_per_step=dt * 1e3, points_per_s=n / dt, windows=len(times), steps_per_window=k,
                    generator=type(solver.generator["train"].generator).__name__,
                    single_launch=solver._fused_sys.fusedk is not None,
                    note="set_tensor_type('cuda', 64) as `import neurodiffeq` does, plain Solver2D + Generator2D: sampling, "
                         "closure in double, Adam and history all on the device",
                    final_loss=solver.metrics_history["train_loss"][-1])
    finally:
        set_tensor_type(device="cpu", float_bits=32)


def pointwise_large():
    """The standalone generated pointwise residual kernel (HBM-bound: reads coordinates + network streams, writes the
    adjoint streams) at sizes where HBM speed, not launch latency, decides: C2's at 1 M and 4 M points, C5's at 1 M."""
    from tests import configs
    from neurodiffeq_amd.engine import FusedSystem
    out = {}
    for name, size in (("c2", 1024), ("c2", 2048), ("c5", 1024)):
        torch.manual_seed(0)
        cfg = configs.make(name, size)
        for net in cfg["nets"]:
            net.to("cuda")
        system = FusedSystem(cfg["nets"], cfg["conds"], cfg["pde"], 2, "cuda", single_kernel=False)
        torch.manual_seed(1)
        batch = [c.detach().cuda() for c in cfg["gen"].get_examples()]
        b, n = system.upload(batch)
        stream = system._stream()
        system.step(batch, train=True)
        t = time_launches(lambda: system.pointwise(b, n, stream, True, n), iters=200, warm=50)
        bpp = system.program.bytes_per_point(train=True)
        out[f"{name}_{n}"] = dict(points=n, us=t * 1e6, bytes_per_point=bpp, gbs=bpp * n / t / 1e9,
                                  frac=bpp * n / t / 1e9 / HBM_PEAK_GBS)
        del system, b
        torch.cuda.empty_cache()
    return out


def scaling_run(args, world, rank, use_dist, dist):
    """``--config cX [--scaling strong|weak]``: one BASELINE config through run_train_epoch() on resident pre-sampled
    batches, data-parallel over the ranks.  strong: the config's batch (C3 262 144, C5 1 048 576 points ...) is cut into
    ``world`` contiguous shards, one per rank; weak: every rank keeps the whole batch of its own draw.  One all-reduce
    of [gradients | loss] per step either way.  Same JSON contract as the headline; no roofline / CPU legs here."""
    from neurodiffeq_amd.generators import ResidentBatchGenerator, SamplerGenerator
    from neurodiffeq_amd.parallel import BatchSharding
    from tests import configs
    name = args.config
    torch.manual_seed(0)
    solver, cfg = configs.make_solver(name)
    solver.fused = "require"
    n_all = cfg["n_points"]
    if args.scaling == "strong":
        base, rem = divmod(n_all, world)
        assert rem == 0, f"{n_all} points do not split evenly over {world} ranks"
        lo, hi = rank * base, (rank + 1) * base
        torch.manual_seed(1)                       # every rank draws the SAME global batches and keeps its shard
    else:
        lo, hi = 0, n_all
        torch.manual_seed(1 + rank)
    solver.generator["train"] = SamplerGenerator(ResidentBatchGenerator.presample(cfg["gen"], 2, "cuda", lo=lo, hi=hi))
    if use_dist:
        solver.dist = BatchSharding(presharded=True)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(vals):
        t = torch.tensor(vals, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()
    for _ in range(args.warmup):
        solver.run_train_epoch()
    windows = timed_windows(solver.run_train_epoch, args.steps, barrier, reduce_max if use_dist else None)
    dt = windows[(len(windows) - 1) // 2]
    per_rank = hi - lo
    total = per_rank * world
    if rank == 0:
        step_s = dt / args.steps
        tf = ALGO_FLOP_PER_PT[name] * total / step_s / 1e12
        print(json.dumps({
            "metric": f"collocation-points/sec (residual+bwd), config {name.upper()}, {args.scaling} scaling",
            "value": total / step_s, "unit": "collocation-points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3,
            "timing": {"windows": len(windows), "steps_per_window": args.steps, "statistic": "median window"},
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name.upper()} of BASELINE.json at its stated size through run_train_epoch()",
                       "points_per_gpu": per_rank, "global_batch": total,
                       "parallelism": f"dp{world} (shard-by-batch, one all-reduce of [P+1] fp32 per step)",
                       "inputs": "pre-sampled in the reference's RNG order, resident in HBM"},
            "algorithmic_tflops": tf, "frac_of_fp32_mfma_peak_per_gpu": tf / world / FP32_MFMA_PEAK_TFLOPS,
            "single_launch": solver._fused_sys.fusedk is not None,
            "final_loss": solver.metrics_history["train_loss"][-1]}), flush=True)
    if use_dist:
        dist.barrier()
        solver.dist.close()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: windows of ~55 ms, repeated until >= 0.25 s (and >= 3 windows) have been timed; the median window counts
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config sub-records (C1, C3, C4, C5)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--trace-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes for roofline.traffic")
    ap.add_argument("--cold-start", action="store_true", help="(default at N = 1 since round 3; kept for old command lines)")
    ap.add_argument("--no-cold-start", action="store_true",
                    help="skip timing the first step of a never-seen PDE (trace + hipcc + self-check, ~1.5 s)")
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE config to run (the driver's headline is c2; c3 / c5 are the sizes where strong scaling "
                         "has work to share: SURVEY.md 8e)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank trains the config's full batch (global batch = N x that); strong: the "
                         "config's batch is split over the ranks")
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child()
    if args.trace_child:
        return traffic_child(8000)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    local_rank = int(os.environ.get("NDQ_BENCH_DEVICE", local_rank))     # debugging aid: several ranks on one device
    torch.cuda.set_device(local_rank)
    torch.set_num_threads(min(16, os.cpu_count() or 1))   # host-side torch ops (sampling leg); ATen's default of one
    # thread per logical cpu is far past the sweet spot for 65k-element ops
    import torch.distributed as dist
    # under torch.distributed.run (RANK set) the RCCL path is exercised even for a single rank, so that the 1-GPU box
    # can validate exactly the code the multi-GPU scaling runs use
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("NDQ_BENCH_BACKEND", "nccl")   # "gloo": dry runs of the N > 1 path with several ranks on ONE GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from neurodiffeq_amd.generators import Generator2D, ResidentBatchGenerator
    from neurodiffeq_amd.parallel import BatchSharding
    from tests import configs

    if args.config != "c2" or args.scaling != "weak":
        return scaling_run(args, world, rank, use_dist, dist)

    # identical weights on every rank (same seed); the global batch is a (256*world) x 256 noisy grid of which each
    # rank keeps its contiguous 65 536-point shard resident.
    torch.manual_seed(0)
    solver, cfg = configs.make_solver("c2", GRID)
    solver.fused = "require"
    torch.manual_seed(1)
    global_gen = Generator2D((GRID * world, GRID), (0, 0), (1, 1), "equally-spaced-noisy")
    n_pool = 8
    resident = ResidentBatchGenerator.presample(global_gen, n_pool, "cuda", lo=rank * N_POINTS, hi=(rank + 1) * N_POINTS)
    from neurodiffeq_amd.generators import SamplerGenerator
    solver.generator["train"] = SamplerGenerator(resident)
    if use_dist:
        solver.dist = BatchSharding(presharded=True)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # W untimed warm-up steps (the closure kernel's first-use self-check, buffer allocation and the host path warm up
    # here), then windows of EXACTLY K steps, each bracketed by barrier + synchronize; a window of the driver's K = 20
    # is 0.5 ms, so windows are repeated until >= 0.25 s has been timed and the MEDIAN window is reported (an MI355X
    # also needs tens of ms of sustained work to reach its clocks: the first windows of a cold box are slower).
    for _ in range(args.warmup):
        solver.run_train_epoch()

    def reduce_max(vals):
        t = torch.tensor(vals, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()
    windows = timed_windows(solver.run_train_epoch, args.steps, barrier, reduce_max if use_dist else None)
    dt = windows[(len(windows) - 1) // 2]
    assert solver.fused_active
    # the headline is the single-launch closure kernel: if its first-use self-check (engine.verify_fused) rejected it,
    # the numbers below would silently be the three-kernel pipeline's -- refuse instead
    assert solver._fused_sys is not None and solver._fused_sys.fusedk is not None, \
        f"single-launch closure kernel not in use: {getattr(solver._fused_sys, 'fused_check', None)}"
    ms = dt / args.steps * 1e3
    value = N_POINTS * world / (dt / args.steps)

    out = None
    if rank == 0:
        out = {
            "metric": "collocation-points/sec (residual+bwd), 2D Laplace 65k pts, 1/2/4/8 GPU",
            "value": value, "unit": "collocation-points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms,
            "timing": {"windows": len(windows), "steps_per_window": args.steps, "statistic": "median window",
                       "window_ms_min": windows[0] * 1e3, "window_ms_median": dt * 1e3, "window_ms_max": windows[-1] * 1e3,
                       "timed_total_s": sum(windows)}, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: Solver2D 2D Laplace, DirichletBVP2D, FCNN(2-32-32-1, tanh), Generator2D "
                                   "256x256 = 65536 noisy-grid points per GPU per step, Adam(1e-3), "
                                   "run_train_epoch() with n_batches_train=1, n_batches_valid=0",
                       "points_per_gpu": N_POINTS, "global_batch": N_POINTS * world,
                       "parallelism": f"dp{world} (shard-by-batch, one all-reduce of [P+1] fp32 per step)",
                       "allreduce": ("none (single process)" if not use_dist else solver.dist.allreduce_kind("cuda")),
                       "inputs": "pre-sampled in the reference's RNG order, resident in HBM"},
            "final_loss": solver.metrics_history["train_loss"][-1],
        }
        if use_dist and hasattr(solver.dist._direct, "status"):
            out["allreduce_flag_timeouts"] = solver.dist._direct.status()      # one-shot exchange: must be 0
    if world == 1 and not use_dist:
        # the same training epochs inside fit(): whole chunks of epochs per native call (ndq_fused_fit_run), no Python in
        # between -- at this size the step is GPU-bound either way; listed so that the two routes can be compared
        kf = max(args.steps, 200)
        solver.fit(kf, tqdm_file=None)
        tf_ = timed_windows(lambda: solver.fit(kf, tqdm_file=None), 1, barrier)
        dtf = tf_[(len(tf_) - 1) // 2] / kf
        out["in_fit"] = {"value": N_POINTS / dtf, "ms_per_step": dtf * 1e3, "epochs_per_call": kf,
                         "note": "fit(k) with n_batches_valid = 0: the headline's training epochs issued k per native call"}
    if rank == 0 and world == 1 and not use_dist:
        system = solver._fused_sys
        batch = solver._generate_batch("train")
        from neurodiffeq_amd.engine import FusedSystem
        pipeline = FusedSystem(solver.nets, solver.conditions, solver.diff_eqs, 2, "cuda", single_kernel=False)
        kb = kernel_breakdown(pipeline, batch)
        if system.fusedk is not None:
            kb.update(fused_breakdown(system, batch))
            # HIP events on the stream the kernel runs on, 1 000 launches back to back (rocprofv3's kernel trace of the same
            # command is committed under profiles/ and must agree)
            threads = system.fused_variant(N_POINTS).threads
            out["roofline"] = {"kernel": "fused_closure_kernel<Cfg<2,1,5,2,2,tanh>, PW, train> (fwd + pointwise + bwd), "
                                         f"{threads} threads per workgroup",
                               "bound": "mfma", "achieved": kb["fused_closure"]["tflops"],
                               "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": kb["fused_closure"]["tflops"] / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                               "algorithmic_flop_per_point": FWD_FLOP_PER_PT + BWD_FLOP_PER_PT,
                               "avg_launch_us": kb["fused_closure"]["us"],
                               # the kernel carries u_xx + u_yy as ONE "Laplacian" stream when the tracer proves the
                               # residual only needs the sum, i.e. it executes 4 streams instead of SURVEY's 5
                               "executed_streams": system.program.streams[0].n_streams,
                               # every GEMM of the kernel runs as bf16 plane products on the bf16 matrix core: 6 per
                               # fp32-class product in the forward GEMMs, 4 in the reverse ones, 3 in the weight gradients
                               # since round 6 (profiles/r06_headline_ab.md); the yardstick stays the fp32 MFMA peak the
                               # earlier rounds were priced against, the dense bf16 peak over the average 13 / 3 products
                               # is given beside it
                               "peak_note": "fp32 MFMA peak (157.3 TFLOP/s); the kernel's GEMMs are split-bf16 (6 / 4 / 3 bf16 MFMA "
                                            "products per fp32-class product in the forward / reverse / weight-gradient GEMMs): "
                                            "ceiling of that mix 2500 / (13 / 3) = 576.9 TFLOP/s",
                               "frac_of_bf16x3_ceiling": kb["fused_closure"]["tflops"] / (2500.0 * 3.0 / 13.0),
                               "executed_gemm_flop_per_point":
                                   3 * 2 * (32 * 2 + 32 * 32 * system.program.streams[0].n_streams
                                            + 32 * system.program.streams[0].n_streams)}
        else:
            out["roofline"] = {"kernel": "mlp_jet_bwd_kernel<Cfg<2,1,5,2,2,tanh>>", "bound": "mfma",
                               "achieved": kb["mlp_jet_bwd"]["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": kb["mlp_jet_bwd"]["tflops"] / FP32_MFMA_PEAK_TFLOPS,
                               "traffic": None, "algorithmic_flop_per_point": BWD_FLOP_PER_PT,
                               "avg_launch_us": kb["mlp_jet_bwd"]["us"]}
        out["kernels"] = kb
        out["roofline_pointwise"] = {"kernel": "ndq_pw_kernel (generated)", "bound": "hbm",
                                     "achieved": kb["pointwise"]["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": kb["pointwise"]["gbs"] / HBM_PEAK_GBS, "traffic": None,
                                     "algorithmic_bytes_per_point": kb["pointwise"]["bytes_per_point"]}
        situ = None if (args.no_traffic or system.fusedk is None) else measure_in_situ()
        if situ is not None:
            # priced on what the kernel takes INSIDE the step (rocprofv3 kernel trace of this run); the back-to-back HIP-event
            # figure stays beside it
            rf = out["roofline"]
            rf["avg_launch_us_back_to_back_hip_events"], rf["frac_back_to_back"] = rf["avg_launch_us"], rf["frac"]
            tfl = (FWD_FLOP_PER_PT + BWD_FLOP_PER_PT) * N_POINTS / (situ["closure_us"] * 1e-6) / 1e12
            rf.update(achieved=tfl, frac=tfl / FP32_MFMA_PEAK_TFLOPS, avg_launch_us=situ["closure_us"],
                      frac_of_bf16x3_ceiling=tfl / (2500.0 * 3.0 / 13.0), in_situ=situ,
                      timing_note="avg_launch_us / achieved / frac: rocprofv3 --kernel-trace of THIS run over 8 000 training "
                                  "steps, second half (the kernel between the tails of successive epochs); *_back_to_back: HIP events "
                                  "around 1 000 launches of the kernel alone")
        tpath = os.path.join(ROOT, "profiles", "traffic_c2.json")
        live = None if (args.no_traffic or system.fusedk is None) else measure_traffic()
        if live is not None:           # HBM bytes per launch measured in THIS run (two rocprofv3 --pmc passes)
            out["roofline"]["traffic"] = live["hbm_bytes"]
            out["roofline"]["traffic_note"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
                                               "300 steps, mean per dispatch; FETCH_SIZE x 2 per the guide and this package's "
                                               "calibration (profiles/archive/r02/r02v_pmc_c2_summary.txt)")
            out["roofline"]["traffic_detail"] = live
        if os.path.exists(tpath):      # the last committed rocprofv3 --pmc measurement (may be stale: see its source field)
            tr = json.load(open(tpath))
            key = "fused_closure" if system.fusedk is not None else None
            if live is None and key and key in tr["kernels"]:
                out["roofline"]["traffic"] = tr["kernels"][key]["hbm_bytes"]
                out["roofline"]["traffic_note"] = "NOT measured in this run (stale-able): " + tr["source"]
        # (the standalone pointwise kernel's counter traffic is not measured by this run: `traffic` stays null rather than
        # citing an old PMC file; its algorithmic bytes are exact by construction -- codegen.PointwiseProgram.bytes_per_point)
        # the same step with host sampling (CPU RNG, bit-exact with the reference) + PCIe upload inside it
        torch.manual_seed(2)
        solver.generator["train"] = SamplerGenerator(cfg["gen"])
        for _ in range(3):
            solver.run_train_epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k2 = min(max(10, args.steps // 10), 200)
        for _ in range(k2):
            solver.run_train_epoch()
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / k2
        out["with_host_sampling"] = {"value": N_POINTS / dt2, "ms_per_step": dt2 * 1e3}
        # ... and with a fresh batch drawn ON the device every step (generators.DeviceGenerator: same distribution,
        # Philox stream instead of the host RNG -> not the reference's numbers, hence not the headline); prefetch=True:
        # the next batch is drawn by extra workgroups of the step's own sums + tail kernel, no sampler launch
        from neurodiffeq_amd.generators import DeviceGenerator
        solver.generator["train"] = SamplerGenerator(DeviceGenerator(cfg["gen"], seed=2, prefetch=True))
        for _ in range(max(10, args.warmup)):
            solver.run_train_epoch()
        torch.cuda.synchronize()
        # (the headline's protocol: median of >= 0.25 s of K-step windows -- one 20-step window is 0.5 ms of noise)
        w3 = timed_windows(solver.run_train_epoch, args.steps, torch.cuda.synchronize)
        dt3 = w3[(len(w3) - 1) // 2] / args.steps
        out["with_device_sampling"] = {"value": N_POINTS / dt3, "ms_per_step": dt3 * 1e3, "windows": len(w3)}
        # ... and what an UNCHANGED user script gets when torch's default device is cuda (the reference's import default,
        # neurodiffeq/__init__.py:22; its generators draw on the default device, generators.py:152,264): a fresh solver on
        # the same config -- its noisy training grid is drawn on the MI355X automatically (generators.on_default_device),
        # one sampler launch + the two launches of the step, nothing resident, nothing opted into
        torch.set_default_device("cuda")
        try:
            torch.manual_seed(0)
            dsolver, dcfg = configs.make_solver("c2", GRID)
            dsolver.fused = "require"
            for _ in range(max(10, args.warmup)):
                dsolver.run_train_epoch()
            torch.cuda.synchronize()
            w4 = timed_windows(dsolver.run_train_epoch, args.steps, torch.cuda.synchronize)
            dt4 = w4[(len(w4) - 1) // 2] / args.steps
            out["default_generator_cuda"] = {"value": N_POINTS / dt4, "ms_per_step": dt4 * 1e3, "windows": len(w4),
                                             "generator": type(dsolver.generator["train"].generator).__name__,
                                             "note": "torch.set_default_device('cuda'), plain Solver2D + Generator2D: noise drawn "
                                                     "by the Philox kernel every step, each generator seeded from torch's cuda generator"}
            del dsolver
        finally:
            torch.set_default_device(None)       # (None removes torch's global device mode; "cpu" would leave one installed)
        if not args.no_configs:
            # the other BASELINE configs at their stated sizes (parity-tested at those sizes in tests/test_gpu_parity.py)
            del pipeline
            solver.generator["train"] = None
            torch.cuda.empty_cache()
            out["configs"] = {name: config_record(name) for name in ("c1", "c3", "c4", "c5")}
            # networks wider than 64 units on C2's grid: the reference's README network first (csrc/ndq_wide.h: no GEMM at
            # all, VALU-bound; csrc/ndq_deep.h: layer by layer through HBM, per-point GEMMs as bf16x3 on the bf16 matrix core)
            for label, (name, size) in WIDE_RECORDS.items():
                out["configs"][label] = dict(config_record(name, size), golden=name,
                                             kernels="csrc/ndq_wide.h" if name in ("w16", "w17") else "csrc/ndq_deep.h")
            out["roofline_pointwise_large"] = pointwise_large()
            out["c2_fp64"] = fp64_record()
            try:
                out["c2_reference_defaults_cuda_float64"] = reference_defaults_record()
            except Exception as e:                      # a side figure must not cost the bench line
                out["c2_reference_defaults_cuda_float64"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            try:        # ... and the same two import defaults with the README's own network, FCNN(2, 1, hidden_units=(512,))
                out["readme_512_reference_defaults_cuda_float64"] = reference_defaults_record("w16", GRID)
            except Exception as e:
                out["readme_512_reference_defaults_cuda_float64"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cold_start and (args.cold_start or not args.no_configs):
            try:
                out["cold_start"] = cold_start()
                out["cold_start_s"] = out["cold_start"]["cold_start_s"]
            except Exception as e:                      # e.g. no hipcc on the box: the figure is missing, the bench line is not
                out["cold_start"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_cpu_baseline:
            cb = out["cpu_baseline"] = cpu_baseline()
            # like for like: resident inputs on both sides / generator draw inside the step on both sides (the host draw +
            # PCIe upload is then what the GPU step waits for; its numbers are the reference's bit for bit)
            if "c2_fp64" in out:
                out["c2_fp64"]["speedup_vs_cpu_fp64"] = out["c2_fp64"]["points_per_s"] / cb["fp64"]["value"]
            out["speedup_vs_cpu_baseline"] = {
                "presampled_inputs_both_sides": value / cb["presampled"]["value"],
                "host_sampling_both_sides": out["with_host_sampling"]["value"] / cb["value"],
                "note": f"headline `value` (inputs resident in HBM) vs the CPU baseline (kind = {cb['kind']}) on a pre-sampled "
                        "batch; `with_host_sampling` (reference RNG draw + upload inside the step) vs its full step"}
    if rank == 0:
        # ONE JSON line; the bulky side records first, the contract keys and the compact headline figures (roofline,
        # cpu_baseline, in_fit, with_*_sampling, default_generator_cuda) LAST: whoever keeps only the tail of this process's
        # stdout still sees the headline
        bulky = ("kernels", "configs", "cold_start", "roofline_pointwise_large", "c2_fp64", "c2_reference_defaults_cuda_float64",
                 "readme_512_reference_defaults_cuda_float64")
        out = {**{k: out[k] for k in bulky if k in out}, **{k: v for k, v in out.items() if k not in bulky}}
        # ... and, as the LAST key, one compact line of fractions of the fp32 MFMA peak per config (step-level, algorithmic
        # flops; c2 = the headline kernel in situ) so that the tail of the line carries them
        fr = {}
        if "roofline" in out:
            fr["c2_kernel"] = round(out["roofline"]["frac"], 4)
            fr["c2_step"] = round((FWD_FLOP_PER_PT + BWD_FLOP_PER_PT) * N_POINTS / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        for name, rec in (out.get("configs") or {}).items():
            if isinstance(rec, dict) and "frac_of_fp32_mfma_peak" in rec:
                fr[name] = round(rec["frac_of_fp32_mfma_peak"], 4)
        out["fracs"] = fr
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        solver.dist.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
