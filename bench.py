#!/usr/bin/env python
"""bench.py -- MPC control steps/s of the batched convex-MPC contact-force solve on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch of synthetic input: every robot of the batch does
one ``compute_contact_forces`` (QP build + OSQP-equivalent solve, mpc_osqp.cc:578-796) on the GPU.
Workload at N=1: BASELINE.json configs[1] -- 4096 Aliengo, trot, horizon 10, flat terrain.  The K+W
input batches are a seeded sequence: step 0 is the cold "osqp_setup" solve, the
following ones advance the gait and perturb the state, so the timed steps are warm-started solves,
as in the reference's control loop.  Inputs are resident in HBM before the timed region.
Multi-GPU: robots shard across ranks (weak scaling, 4096 per GPU, no data-path collective).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (vector)"; SURVEY.md 8(d) prices against it
FP64_VECTOR_PEAK_TFLOPS = 78.6    # datasheet fp64 vector rate (half the fp32 rate); the kernel computes in fp64


def algorithmic_flops(h, contact, iters, nfact):
    """SURVEY.md 8(d) minimal-algorithm flop count per control step, summed over the batch, split by kernel:
    (assembly kernel: A^k B, P recursion, q;  solve kernel: F factorisations + I ADMM iterations).
    n_r = 3 * (stance leg-steps in the horizon); I = ADMM iterations executed; F = factorisations."""
    n_r = 3.0 * contact.reshape(len(contact), -1).sum(1)
    fixed = 4056.0 * (h - 1) + 3900.0 * h * (h + 1) / 2 + 2.0 * 13 * h * (13 + 12 * h)
    per = nfact * n_r ** 3 / 3.0 + iters * (2.0 * n_r ** 2 + 40.0 * n_r)
    return float(fixed * len(n_r)), float(per.sum())


def executed_flops(h, iters, nfact, polished):
    """fp64 operations the kernel actually executes per batch: the OSQP-faithful algorithm keeps all
    n = 12 h variables.  Per robot: nfact_K full symmetric sweeps (n pivots x MT tiles x (36 FMA + 6 mul)), the masked
    polish sweep counted as half a sweep, `iters` ADMM iterations (tile mat-vec 2 x 36 FMA per tile + ~45 flops per
    variable + ~15 per constraint row), 10 Ruiz passes (4 ops per tile entry) and three P_s products (the assembly kernel's
    work is not counted here)."""
    n, m, mt = 12 * h, 20 * h, h * (2 * h + 1)
    sweep = n * mt * (72.0 + 6.0)
    it = mt * 144.0 + 45.0 * n + 15.0 * m
    fixed = 10 * mt * 144.0 + 3 * mt * 144.0
    n_k = nfact - polished                       # factorisations of K (the polish one is counted in info[4])
    return float((n_k * sweep + polished * 0.5 * sweep + iters * it + fixed).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--robots", type=int, default=4096, help="robots per GPU")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-control-loop", action="store_true", help="skip the secondary controller.run leg (profiling runs)")
    args = ap.parse_args()

    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd import layout as L
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    n, h, K, W = args.robots, args.horizon, args.steps, args.warmup
    # this rank's shard of the global batch: robots [rank*n, (rank+1)*n) of a world*n batch
    wl = make_solver_workload(n, h=h, seed=1000 + rank, config=2)
    batches = []
    w = wl
    for s in range(K + W):
        batches.append(w.inputs)
        w = perturb_workload(w, 7000 + 131 * s + rank)
    d_in = [torch.from_numpy(b).to(dev) for b in batches]          # resident in HBM before timing
    inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
    solver = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, device=dev)
    solver.enable_timing()                       # HIP events inside the library, around each kernel, on the launch stream
    infos = [torch.zeros((n, 8), dtype=torch.int32, device=dev) for _ in range(K)]

    for s in range(W):
        solver.solve(d_in[s])
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    first_out = torch.zeros((n, 12 * h), dtype=torch.float64, device=dev)
    for s in range(K):
        ev[s][0].record()                      # HIP events on the stream the kernel is launched on
        solver.solve(d_in[W + s], forces=first_out if s == 0 else None, info=infos[s])
        ev[s][1].record()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        from rl_mpc_locomotion_amd.sharding import max_over_ranks
        elapsed = max_over_ranks(elapsed, dev)

    step_ms = np.array([a.elapsed_time(b) for a, b in ev])        # assembly + solve + dispatch-order kernels of one step
    kt = min(K, 64)
    assemble_ms, kernel_ms = (a.astype(np.float64) for a in solver.kernel_times(kt))   # the dominant kernel (mpc_solve_kernel) alone
    first_forces = first_out.cpu().numpy()
    info = torch.stack(infos).cpu().numpy()                         # [K, n, 8]
    solved = int((info[..., 1] == 1).sum())
    flops = asm_flops = 0.0
    for s in range(K):
        contact = batches[W + s][:, L.IN_CONTACT:L.IN_CONTACT + 4 * h]
        fa, fs = algorithmic_flops(h, contact, info[s, :, 0].astype(np.float64), info[s, :, 4].astype(np.float64))
        flops += fs; asm_flops += fa
    flops_per_launch = flops / K                                     # of the dominant kernel (mpc_solve_kernel)
    exec_flops = sum(executed_flops(h, info[s, :, 0].astype(np.float64), info[s, :, 4].astype(np.float64),
                                    (info[s, :, 2] != 0).astype(np.float64)) for s in range(K)) / K
    achieved_tflops = flops_per_launch / (kernel_ms.mean() * 1e-3) / 1e12

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # HBM traffic of the solve kernel is a rocprofv3 PMC measurement taken offline on this same command
    # (tools/pmc_passes.sh -> profiles/rNN_pmc_summary.json); bench.py cannot run the profiler on itself.
    traffic = None
    try:
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
        if pm and n == 4096 and h == 10:
            traffic = float(json.load(open(pm[-1]))["hbm_traffic_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        traffic = None
    value = world * n * K / elapsed
    out = {
        "metric": "MPC control steps/sec (whole node) @ horizon=10, 4096 robots; max |GRF| err vs OSQP",
        "value": value,
        "unit": "control steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{n} Aliengo/GPU, trot, horizon={h}, flat terrain, 1 compute_contact_forces per robot per step "
                               "(BASELINE configs[1]); warm-started seeded sequence, SURVEY.md 8(d)",
                   "robots_per_gpu": n, "horizon": h, "parallelism": f"robot-sharded x{world}"},
        "solved_fraction": solved / float(K * n),
        "mean_admm_iters": float(info[..., 0].mean()),
        "mean_factorisations": float(info[..., 4].mean()),
        "roofline": {"bound": "mfma", "achieved": achieved_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_note": "HBM-side bytes per launch from rocprofv3 PMC (profiles/*_pmc_summary.json, measured offline on this command)",
                     "note": "vector-FP bound, no MFMA/HBM roofline applies (SURVEY 8d): algorithmic flops (n_r formula) / "
                             "mean duration of mpc_solve_kernel from HIP events; kernel arithmetic is fp64 (peak 78.6 TF)",
                     "kernel": "mpc_solve_kernel", "kernel_ms": float(kernel_ms.mean()), "assemble_kernel_ms": float(assemble_ms.mean()),
                     "step_ms_all_kernels": float(step_ms.mean()), "flops_per_launch": flops_per_launch,
                     "assemble_kernel_flops_per_launch": asm_flops / K,
                     "frac_fp64_peak": achieved_tflops / FP64_VECTOR_PEAK_TFLOPS,
                     "executed": {"flops_per_launch": exec_flops, "tflops": exec_flops / (kernel_ms.mean() * 1e-3) / 1e12,
                                  "frac_fp64_peak": exec_flops / (kernel_ms.mean() * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                  "note": "operations the OSQP-faithful kernel executes (all 12h variables kept; bench.py executed_flops), "
                                          "for orientation only -- `achieved` / `frac` above use the SURVEY 8(d) minimal-algorithm count"}},
    }
    if not args.no_control_loop and world == 1:      # secondary legs: single-GPU runs only
        out["control_loop"] = control_loop_leg(n, h, dev)
        out["policy"] = policy_leg(n, dev)
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(wl, batches, W, h, gpu_first_forces=first_forces)
        out["max_grf_err_vs_osqp"] = out["cpu_baseline"].pop("_gpu_err", None)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def control_loop_leg(n, h, dev, ticks=40, warm=10):
    """Secondary figure (NOT `value`): robot-ticks/s of the whole controller.run seam on device tensors --
    state estimator + leg kinematics + gait / foot placement + the MPC solve on every second tick (the
    reference's cadence, RobotRunnerMin.py:21-22) + swing / stance commands + joint torques."""
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.synthetic import TickStream
    ts = TickStream(n, seed=4242, config=2)
    ctl = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=h, device=dev)
    ins = [tuple(torch.from_numpy(a).to(dev) for a in ts.tick(k)) for k in range(warm + ticks)]
    for k in range(warm):
        ctl.run(*ins[k])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(warm, warm + ticks):
        ctl.run(*ins[k])
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    solved = float((ctl.solver_info()[:, 1] == 1).mean())
    return {"robot_ticks_per_s": n * ticks / dt, "ms_per_tick": dt / ticks * 1e3, "ticks": ticks, "mpc_every_n_ticks": 2,
            "solved_fraction_last_mpc": solved,
            "note": "controller.run for every robot per tick; the MPC solve runs on every 2nd tick, so this is ~2x the control-step rate by construction"}


def policy_leg(n, dev, steps=50, warm=5):
    """Secondary figure: the weight policy in front of the controller (observations -> 48-512-256-128-12 ELU actor ->
    MPC weights -> command record) for all robots, random-init parameters of the reference architecture."""
    import torch
    from rl_mpc_locomotion_amd.weight_policy import WeightPolicy
    rng = np.random.default_rng(99)
    dims = [48, 512, 256, 128, 12]
    layers = [((rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32), np.zeros(dims[i + 1], np.float32)) for i in range(4)]
    pol = WeightPolicy(layers, device=dev)
    t = lambda *shape: torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dev)
    dof, est, nrm, cmd, act = t(n, 12, 2), t(n, 18), t(n, 3), t(n, 3), t(n, 12)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(warm + steps):
        if k == warm:
            ev0.record()
        obs = pol.compute_observations(dof, est, nrm, cmd, act)
        w = pol.step(obs)
        pol.pack_commands(cmd, w)
    ev1.record()
    torch.cuda.synchronize(dev)
    ms = ev0.elapsed_time(ev1) / steps
    flops = 2.0 * n * sum(dims[i] * dims[i + 1] for i in range(4))
    return {"ms_per_step": ms, "robots": n, "tflops_fp32": flops / (ms * 1e-3) / 1e12,
            "note": "three launches per step (observations, fused MLP on the fp32 MFMA pipe, command packing); not part of `value`"}


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (the GPU boxes expose 256 logical CPUs but grant a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(wl, batches, W, h, sample=2048, steps=4, gpu_first_forces=None):
    """The reference path (oracle/_ref: restated mpc_osqp.cc assembly + the vendored OSQP) timed on the
    host cores on a bounded sample of the same workload: the first `sample` robots, cold solve +
    `steps` timed warm solves (the same batches the GPU warmed up / timed on)."""
    from oracle.refmpc import RefBatch
    cores = usable_cores()
    sample = min(sample, len(wl.mass))
    ref = RefBatch(wl.mass[:sample], wl.inertia_diag[:sample], h, wl.dt_mpc, wl.alpha)
    for s in range(W):
        ref.solve(batches[s][:sample], nthreads=cores)
    t0 = time.perf_counter()
    fr0 = None
    for s in range(steps):
        fr = ref.solve(batches[W + s][:sample], nthreads=cores)
        if s == 0:
            fr0 = fr.copy()
    dt = time.perf_counter() - t0
    err = None
    if gpu_first_forces is not None:
        ok = ~np.isnan(fr0[:, 0])
        g = gpu_first_forces[:sample]
        err = float((np.abs(g[ok, :12] - fr0[ok, :12]).max(1) / np.maximum(np.abs(fr0[ok, :12]).max(1), 1.0)).max())
    return {"_gpu_err": err, "value": sample * steps / dt, "unit": "control steps/s", "cores": cores, "kind": "reference",
            "sample": f"first {sample} robots of the workload, {W} warm-up + {steps} timed warm-started solves each, "
                      f"one OSQP workspace per robot, static partition over {cores} threads"}


if __name__ == "__main__":
    main()
