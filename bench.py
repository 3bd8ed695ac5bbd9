#!/usr/bin/env python
"""bench.py -- MPC control steps/s of the batched convex-MPC contact-force solve on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

A "step" = one pass of the hot path over one batch of synthetic input = SURVEY.md 8(d)'s unit of work for every robot of the batch: one
``controller.run(dof_states, body_states, commands) -> torques`` call (the reference's batch seam, RL_Environment/tasks/aliengo.py:246-256) on
which EVERY robot is due for its MPC update -- state estimator, leg kinematics, gait / foot placement, one ``compute_contact_forces``
(QP build + OSQP-equivalent solve, mpc_osqp.cc:578-796), swing / stance commands and the leg-torque map (LegController.updateCommand, a22).
(`--seam ctrl`, the default for configs 2 and 3.  `--seam solver` times the bare ``compute_contact_forces`` batch, a2-a13 without the
controller around it: the `secondary.solver_seam` line, and the default seam of configs 4 / 5, whose terrain normals are a solver-input
specification.)

Workloads (BASELINE.json configs):
  --config 2 (default, the configuration the metric is quoted on): 4096 Aliengo per GPU, trot, horizon 10, flat terrain; weak scaling.
  --config 3: 4096 robots per GPU, {Go1, A1, Aliengo} mixed, trot / walk / bound CYCLING every 50 steps (use --steps > 40 to time a switch;
              `secondary.config3` of the default run is 110 steps of it), horizon 10; weak scaling.
  --config 4: 32768 Aliengo in total, horizon 16, random terrain normals, sharded over the N ranks; strong scaling.
  --config 5: 65536 Aliengo in total, horizon 20, sharded over the N ranks; strong scaling.
The K+W input batches are a seeded sequence: step 0 is the cold "osqp_setup" solve, the following ones advance the gait and
perturb the state, so the timed steps are warm-started solves, as in the reference's control loop.  Inputs are resident in HBM
before the timed region.

Multi-GPU: one process per GPU, robots shard across ranks, no data-path collective.  With N > 1 and no
torch.distributed environment, this script launches itself under ``python -m torch.distributed.run`` (one node, 127.0.0.1).
A separate, separately reported leg times the optional RCCL all-gather of the per-robot torques on a side stream.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (vector)"; SURVEY.md 8(d) prices against it
FP64_VECTOR_PEAK_TFLOPS = 78.6    # datasheet fp64 vector rate (half the fp32 rate); the kernels compute in fp64

CONFIGS = {
    2: dict(h=10, robots_total=None, robots_per_gpu=4096, scaling="weak",
            what="Aliengo, trot, horizon=10, flat terrain (BASELINE configs[1])"),
    3: dict(h=10, robots_total=None, robots_per_gpu=4096, scaling="weak",
            what="{Go1, A1, Aliengo} mixed, trot / walk / bound cycling every 50 steps, horizon=10 (BASELINE configs[2])"),
    4: dict(h=16, robots_total=32768, robots_per_gpu=None, scaling="strong",
            what="Aliengo, horizon=16, random terrain normals, 32768 robots sharded over the ranks (BASELINE configs[3])"),
    5: dict(h=20, robots_total=65536, robots_per_gpu=None, scaling="strong",
            what="Aliengo, horizon=20, 65536 robots sharded over the ranks (BASELINE configs[4]; fp64 arithmetic, see DESIGN.md)"),
}


def algorithmic_flops(h, contact, iters, nfact):
    """SURVEY.md 8(d) minimal-algorithm flop count per control step, summed over the batch, split by kernel:
    (prep kernel: A^k B, P recursion, q;  solve kernel: F factorisations + I ADMM iterations).
    n_r = 3 * (stance leg-steps in the horizon); I = ADMM iterations executed; F = factorisations."""
    n_r = 3.0 * contact.reshape(len(contact), -1).sum(1)
    fixed = 4056.0 * (h - 1) + 3900.0 * h * (h + 1) / 2 + 2.0 * 13 * h * (13 + 12 * h)
    per = nfact * n_r ** 3 / 3.0 + iters * (2.0 * n_r ** 2 + 40.0 * n_r)
    return float(fixed * len(n_r)), float(per.sum())


def executed_flops(h, iters, nfact):
    """fp64 operations the solve kernel executes per batch: the OSQP iteration on all 12 h variables, with the KKT
    solve carried through the 6 h x 6 h wrench-space core.  Per robot: `nfact` factorisations (6 h pivots x MT tiles x (36 FMA + 6 mul), or at h = 10, where two pivots are swept per phase,
    3 h pairs x MT tiles x (84 FMA + 12 mul);
    + the tile build 2 x 216 FMA + 72 mul per tile + ~900 flops per foot), `iters` ADMM iterations (tile mat-vec 2 x 36 FMA per tile
    + ~230 flops per foot), and ~4 products with Theta per termination check / polish.  (The prep kernel's work -- ten Ruiz passes
    over the dense P, 4 ops per entry -- is not counted here.)"""
    nw, nf, mt = 6 * h, 4 * h, h * (h + 1) // 2
    sweep = nw * mt * (90.0 if h == 10 else 78.0)
    build = mt * (2 * 432.0 + 72.0) + nf * 900.0
    it = mt * 144.0 + nf * 230.0
    return float((nfact * (sweep + build) + iters * it + (iters / 25.0 + 3.0) * mt * 290.0).sum())


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10, help="untimed steps (default: one gait period, which also fills the dispatch-order history)")
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--robots", type=int, default=None, help="robots per GPU (overrides the configuration's size)")
    ap.add_argument("--horizon", type=int, default=None, help="(overrides the configuration's horizon)")
    ap.add_argument("--seam", default=None, choices=["ctrl", "solver"], help="ctrl: controller.run with every robot due (SURVEY 8(d)'s unit incl. the torque map; default for "
                                                                             "configs 2, 3); solver: the bare compute_contact_forces batch (default for configs 4, 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-control-loop", action="store_true", help="skip the secondary legs (profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary single-GPU lines of the other configurations")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps each; `value` is the MEDIAN block (one block of ~20 steps is an 18 ms sample)")
    ap.add_argument("--robots-total", type=int, default=None, help="(strong-scaling configurations: overrides the total that is sharded over the ranks)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of the N > 1 launch (nccl = RCCL; gloo only with --emulate)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "peer", "both"],
                    help="N > 1: which torque exchange the separately reported gather leg times -- the RCCL all-gather (default), the one-shot direct peer writes "
                         "(sharding.PeerExchange: hipIpc + flags; unmeasured across GPUs, hence opt-in), or both for an A/B")
    ap.add_argument("--emulate", action="store_true",
                    help="CPU dry run of this script's control flow (self-launch, sharding, barriers, gather leg, rank-0 JSON) with the host emulation of the "
                         "kernels (tests/emu) standing in for the HIP library: for tests/test_bench_multirank.py, the numbers mean nothing")
    return ap.parse_args()


class _EmulatedSolver:
    """--emulate: the host emulation of the kernels behind BatchedConvexMpc's interface (CPU tensors)."""

    def __init__(self, wl, h):
        from tests.emu.emu import EmuBatch
        self.e = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)

    def enable_timing(self):
        pass

    def reset(self):
        self.e.state[:] = 0.0

    def solve(self, d_in, forces=None, info=None):
        import torch
        f = self.e.solve(d_in.numpy(), nthreads=2)
        if forces is not None:
            forces.copy_(torch.from_numpy(np.nan_to_num(f)))
        if info is not None:
            info.copy_(torch.from_numpy(self.e.info))

    def kernel_times(self, k):
        return np.full(k, 1e-3, np.float32), np.full(k, 1e-3, np.float32)


class _EmulatedLocomotion:
    """--emulate: the host emulation of the controller kernels behind BatchedLocomotion's interface (CPU tensors)."""

    def __init__(self, cs, h, controller_dt):
        from tests.emu.emu import EmuLocomotion
        self.e = EmuLocomotion(cs.robot_type, cs.gait_id, horizon=h, controller_dt=controller_dt, nthreads=2)

    def enable_timing(self):
        pass

    def reset(self):
        self.e.reset()

    def set_iteration(self, it):
        self.e.set_iteration(it)

    def set_gait(self, gait_id):
        self.e.set_gait(gait_id.numpy() if hasattr(gait_id, "numpy") else gait_id)

    def run(self, dof, body, cmd):
        import torch
        return torch.from_numpy(self.e.run(dof.numpy(), body.numpy(), cmd.numpy()))

    def kernel_times(self, k):
        return np.full(k, 1e-3, np.float32), np.full(k, 1e-3, np.float32)

    def solver_info(self):
        return self.e.solver_info()

    def solver_record(self):
        return self.e.solver_record()

    def solver_forces(self):
        return np.nan_to_num(self.e.solver_forces())


class _Model:
    """what cpu_baseline / exact_leg need to know about the robots of a leg"""

    def __init__(self, robot_type, dt_mpc, alpha):
        from rl_mpc_locomotion_amd.quadruped import ROBOT_TABLE64, COL_MASS, COL_INERTIA
        self.mass = ROBOT_TABLE64[robot_type, COL_MASS]
        self.inertia_diag = ROBOT_TABLE64[robot_type, COL_INERTIA:COL_INERTIA + 3]
        self.dt_mpc, self.alpha = dt_mpc, alpha


CTRL_DT = 0.02      # controller.run period of the ctrl seam: iterationsBetweenMPC = int(27 / (1000 * 0.02)) = 1 (RobotRunnerMin.py:21-22), so EVERY call is
                    # an MPC update of every robot, with the reference's dtMPC = 0.02 (its own 0.01 x 2, ConvexMPCLocomotion.py:58)


def run_leg_ctrl(cfg_id, n, h, K, W, dev, rank, world, dist, repeats=1, emulate=False):
    """`value`'s leg: K timed controller.run calls (every robot due on every call) after W warm-up calls, `repeats` blocks, each bracketed by
    barrier + synchronize; then ONE untimed replay of the same block that fetches, per step, what the solver was handed and what it did
    (records, info) -- the solves are deterministic, so the replay's are the timed block's."""
    import torch
    from rl_mpc_locomotion_amd import layout as L
    from rl_mpc_locomotion_amd.synthetic import ControlStepStream

    R = max(1, repeats)
    cs = ControlStepStream(n, h=h, seed=1000 + rank, config=cfg_id)            # this rank's shard: its own seeded robots
    host = [cs.step(s) for s in range(W + K)]
    ins = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in t) for t in host]     # resident in HBM before timing
    if emulate:
        ctl = _EmulatedLocomotion(cs, h, CTRL_DT)
        sync = lambda: None
    else:
        from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
        ctl = BatchedLocomotion(cs.robot_type, cs.gait_id, horizon=h, controller_dt=CTRL_DT, device=dev)
        assert ctl.iterations_between_mpc == 1
        sync = lambda: torch.cuda.synchronize(dev)
    ctl.enable_timing()
    # config 3: Parameters.cmpc_gait CYCLES Trot -> Walk -> Bound every 50 steps; the ids of every switch inside the sequence are resident in HBM
    # before timing and handed over as device tensors (mpc_ctrl_set_gait_device: stream-ordered, no host round trip), inside the timed region when it falls there
    gaits = {s: torch.from_numpy(cs.gait_at(s)).to(dev) for s in range(W + K) if s == 0 or cs.gait_switch_at(s)} if cs.gait_at(0) is not None else {}
    block_s, prep_ms, solve_ms = [], [], []
    for r in range(R + 1):                       # block R is the untimed replay
        replay = r == R
        ctl.reset()
        ctl.set_iteration(cs.iteration0)
        recs, infos, first_forces = [], [], None
        for s in range(W):
            if s in gaits:
                ctl.set_gait(gaits[s])
            ctl.run(*ins[s])
            if replay:
                recs.append(ctl.solver_record())
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for s in range(K):
            if W + s in gaits:
                ctl.set_gait(gaits[W + s])
            ctl.run(*ins[W + s])
            if replay:
                recs.append(ctl.solver_record()); infos.append(ctl.solver_info())
                if s == 0:
                    first_forces = ctl.solver_forces()
        sync()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if replay:
            break
        if dist is not None:
            from rl_mpc_locomotion_amd.sharding import max_over_ranks
            elapsed = max_over_ranks(elapsed, dev)
        block_s.append(elapsed)
        a, b = ctl.kernel_times(min(K, 64))
        prep_ms.append(a.astype(np.float64)); solve_ms.append(b.astype(np.float64))
    prep_ms, solve_ms = np.concatenate(prep_ms), np.concatenate(solve_ms)
    info = np.stack(infos)                                           # [K, n, 8]
    flops = asm_flops = exec_flops = 0.0
    for j in range(K):
        contact = recs[W + j][:, L.IN_CONTACT:L.IN_CONTACT + 4 * h]
        it, nf = info[j, :, 0].astype(np.float64), info[j, :, 4].astype(np.float64)
        fa, fs = algorithmic_flops(h, contact, it, nf)
        flops += fs; asm_flops += fa
        exec_flops += executed_flops(h, it, nf)
    return dict(gait_switches_timed=sum(1 for s in gaits if s >= W and s > 0), wl=_Model(cs.robot_type, CTRL_DT, 1e-5), batches=recs, solver=ctl, block_s=np.array(block_s), elapsed=float(np.median(block_s)), prep_ms=prep_ms, first_dof=host[W][0], robot_type=cs.robot_type,
                solve_ms=solve_ms, info=info, first_forces=first_forces, flops_per_launch=flops / K, prep_flops_per_launch=asm_flops / K, exec_flops=exec_flops / K)


def run_leg(cfg_id, n, h, K, W, dev, rank, world, dist, repeats=1, emulate=False):
    """Warm up, then time `repeats` blocks of exactly K steps, each bracketed by barrier + synchronize; returns the raw measurements
    of this rank (block times already reduced to the slowest rank)."""
    import torch
    from rl_mpc_locomotion_amd import layout as L
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload

    R = max(1, repeats)
    wl = make_solver_workload(n, h=h, seed=1000 + rank, config=cfg_id)   # this rank's shard: its own seeded robots
    batches = []
    w = wl
    for s in range(W + K):
        batches.append(w.inputs)
        w = perturb_workload(w, 7000 + 131 * s + rank)
    d_in = [torch.from_numpy(b).to(dev) for b in batches]          # resident in HBM before timing
    if emulate:
        solver = _EmulatedSolver(wl, h)
        sync = lambda: None
    else:
        from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
        inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
        solver = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, device=dev)
        sync = lambda: torch.cuda.synchronize(dev)
    solver.enable_timing()                       # HIP events inside the library, around each kernel, on the launch stream
    infos = [torch.zeros((n, 8), dtype=torch.int32, device=dev) for _ in range(K)]
    first_out = torch.zeros((n, 12 * h), dtype=torch.float64, device=dev)

    # Every block is the SAME experiment -- cold start, the W warm-up steps, then the K timed steps on the same seeded sequence -- so the
    # blocks are repeated samples of one quantity and their median is meaningful (a longer sequence would drift: the states random-walk).
    block_s, prep_ms, solve_ms = [], [], []
    for r in range(R):
        if r:
            solver.reset()
        for s in range(W):
            solver.solve(d_in[s])
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for s in range(K):
            solver.solve(d_in[W + s], forces=first_out if s == 0 else None, info=infos[s])
        sync()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            from rl_mpc_locomotion_amd.sharding import max_over_ranks
            elapsed = max_over_ranks(elapsed, dev)
        block_s.append(elapsed)
        a, b = solver.kernel_times(min(K, 64))
        prep_ms.append(a.astype(np.float64)); solve_ms.append(b.astype(np.float64))
    prep_ms, solve_ms = np.concatenate(prep_ms), np.concatenate(solve_ms)
    info = torch.stack(infos).cpu().numpy()                         # [K, n, 8] (the last block's; every block solves the same problems)
    flops = asm_flops = exec_flops = 0.0
    for j in range(K):
        contact = batches[W + j][:, L.IN_CONTACT:L.IN_CONTACT + 4 * h]
        it, nf = info[j, :, 0].astype(np.float64), info[j, :, 4].astype(np.float64)
        fa, fs = algorithmic_flops(h, contact, it, nf)
        flops += fs; asm_flops += fa
        exec_flops += executed_flops(h, it, nf)
    return dict(wl=wl, batches=batches, solver=solver, block_s=np.array(block_s), elapsed=float(np.median(block_s)), prep_ms=prep_ms, solve_ms=solve_ms, info=info,
                first_forces=first_out.cpu().numpy(), flops_per_launch=flops / K, prep_flops_per_launch=asm_flops / K, exec_flops=exec_flops / K)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launch ourselves as one process per GPU (the driver's own launch line); rank 0 prints the JSON line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import rl_mpc_locomotion_amd  # noqa: F401

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.backend == "gloo" and not args.emulate:
        raise SystemExit("bench.py: --backend gloo is the CPU dry run, it needs --emulate (the product path has no CPU fallback)")
    if not args.emulate and (not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank):
        raise SystemExit(f"bench.py: rank {rank} has no GPU {local_rank} ({torch.cuda.device_count()} visible)")
    dev = torch.device("cpu") if args.emulate else torch.device(f"cuda:{local_rank}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")
        assert dist.get_world_size() == args.gpus, "the process group has a different number of ranks than --gpus"
    if not args.emulate:
        torch.cuda.set_device(dev)

    cfg = CONFIGS[args.config]
    h = args.horizon or cfg["h"]
    K, W = args.steps, args.warmup
    robots_total = args.robots_total or cfg["robots_total"]
    if args.robots:
        n = args.robots
    elif cfg["robots_per_gpu"]:
        n = cfg["robots_per_gpu"]
    else:
        from rl_mpc_locomotion_amd.sharding import shard_bounds
        lo, hi = shard_bounds(robots_total, rank, world)
        n = hi - lo
    n_total = n * world if (args.robots or cfg["robots_per_gpu"]) else robots_total

    # configs 4 / 5 specify the solver's terrain-normal argument directly, so their default seam is the solver's; `--seam ctrl` runs them
    # through controller.run as well (16- / 20-segment gaits, the ground normal from the estimator's foot-contact fit: `secondary.config4_ctrl` / `config5_ctrl`)
    seam = args.seam or ("ctrl" if args.config in (2, 3) else "solver")
    clock0 = device_state(local_rank) if not args.emulate and rank == 0 else None
    leg = run_leg_ctrl if seam == "ctrl" else run_leg
    m = leg(args.config, n, h, K, W, dev, rank, world, dist, repeats=args.repeats, emulate=args.emulate)
    clock1 = device_state(local_rank, smi=False) if not args.emulate and rank == 0 else None
    gather = all_gather_leg(n, n_total, dev, dist) if dist is not None else None
    if dist is not None and args.exchange in ("peer", "both") and not args.emulate:
        gather["peer_write"] = peer_write_leg(n, n_total, rank, world, dev, dist)
    if dist is not None and seam == "ctrl":
        gather["sharded_loop"] = sharded_loop_leg(args.config, n_total, h, dev, dist, emulate=args.emulate)

    if rank != 0:
        dist.destroy_process_group()
        return

    info, solve_ms, prep_ms = m["info"], m["solve_ms"], m["prep_ms"]
    achieved_tflops = m["flops_per_launch"] / (solve_ms.mean() * 1e-3) / 1e12
    # HBM-side traffic is a rocprofv3 PMC measurement of THIS command, taken in separate --pmc passes (tools/pmc_passes.sh ->
    # profiles/<TRAFFIC_PROFILE>_pmc_summary[_prep]_h10.json; bench.py cannot run the profiler on itself).  The files are named here, not
    # searched for, and quoted only if they were measured on the kernel sources of this tree (their kernel_source_sha256).
    traffic, traffic_parts = None, None
    if n == 4096 and h == 10 and args.config == 2 and not args.emulate:
        traffic, traffic_parts = traffic_from_profile()
    value = n_total * K / m["elapsed"]
    out = {
        "metric": f"MPC control steps/sec (whole node) @ horizon={h}, {n} robots; max |GRF| err vs OSQP",   # BASELINE.json's metric at the default configuration
        "value": value,
        "unit": "control steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": m["elapsed"] / K * 1e3,
        "repeats": len(m["block_s"]),
        "timed": f"median of {len(m['block_s'])} blocks; a block = cold start, {W} untimed warm-up steps, then exactly {K} timed steps bracketed by barrier + synchronize "
                 "(max over ranks per block); every block runs the same seeded sequence",
        "block_ms_per_step": [float(b) / K * 1e3 for b in m["block_s"]],
        "higher_is_better": True,
        "scaling": cfg["scaling"],
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"config {args.config}: {n} robots/GPU x {world} GPU(s), {cfg['what']}; " + (
                       "one controller.run(dof_states, body_states, commands) -> torques call per step with EVERY robot due for its MPC update: state estimator, leg "
                       "kinematics, gait / foot placement, compute_contact_forces (QP build + solve), swing / stance commands, leg-torque map a22 = SURVEY.md 8(d)'s "
                       "unit of work; warm-started seeded sequence at 8(d)'s input distributions" if seam == "ctrl" else
                       "one compute_contact_forces (QP build + solve) per robot per step, warm-started seeded sequence, SURVEY.md 8(d); the bare solver batch "
                       "(a2-a13), no controller around it"),
                   "seam": seam, "robots_per_gpu": n, "robots_total": n_total, "horizon": h, "parallelism": f"robot-sharded x{world}"},
        "solved_fraction": float((info[..., 1] == 1).mean()),
        "mean_admm_iters": float(info[..., 0].mean()),
        "mean_factorisations": float(info[..., 4].mean()),
        "roofline": {"bound": "vector_fp64", "achieved": achieved_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_parts": traffic_parts,
                     # "and rocprof-measured HBM GB/s / 8 TB/s, side by side": the fabric-side bytes of a step over the step's two solver kernels
                     "hbm": None if traffic is None else {"gb_per_s": traffic / ((prep_ms + solve_ms).mean() * 1e-3) / 1e9, "peak_gb_per_s": 8000.0,
                                                           "frac": traffic / ((prep_ms + solve_ms).mean() * 1e-3) / 8e12,
                                                           "algorithmic_gb_per_s": 47.7e6 * (n / 4096.0) / ((prep_ms + solve_ms).mean() * 1e-3) / 1e9,
                                                           "note": "<< 1 by design (SURVEY 8d: the path is not HBM-bound); the bytes are an upper bound on HBM traffic (Infinity-Cache hits included)"},
                     "traffic_note": "HBM-side bytes per step, BOTH kernels (prep + solve), from rocprofv3 PMC (profiles/*_pmc_summary*_h10.json, measured offline on this command); "
                                     "algorithmic bytes per step: input 384 B + forces 960 B + state 2 x 5136 B per robot = 47.7 MB",
                     "note": "vector-FP bound, no MFMA / HBM roofline applies (SURVEY 8d): algorithmic flops (n_r formula, solve-kernel share) / "
                             "mean duration of the solve kernel (mpc_solve_jobs_kernel) from HIP events on the launch stream; `peak` is the FP32 vector rate SURVEY 8(d) "
                             "prescribes, the kernel's arithmetic is fp64 (frac_fp64_peak, peak 78.6 TF)",
                     "kernel": "mpc_solve_jobs_kernel<10> (persistent: ADMM and polish jobs of every robot)", "kernel_ms": float(solve_ms.mean()), "prep_kernel_ms": float(prep_ms.mean()),
                     "step_ms_all_kernels": float((prep_ms + solve_ms).mean()), "step_ms_outside_the_two_solver_kernels": float(m["elapsed"] / K * 1e3 - (prep_ms + solve_ms).mean()),
                     "flops_per_launch": m["flops_per_launch"],
                     "prep_kernel_flops_per_launch": m["prep_flops_per_launch"],
                     "frac_fp64_peak": achieved_tflops / FP64_VECTOR_PEAK_TFLOPS,
                     "mfma": {"utilisation_of_this_kernel": 0.0,
                              "note": "by design: the one GEMM-shaped contraction of the reference's assembly (B_qp^T Q B_qp) is removed algebraically (P = alpha I + BB^T Theta BB, "
                                      "DESIGN.md 3.1) and fp64 MFMA issues at the vector rate on gfx950; the matrix pipe is used where it cuts INSTRUCTIONS: the weight policy's "
                                      "actor (v_mfma_f32_32x32x2_f32, `policy`) and the exact mode's seeded Gram inverse (v_mfma_f64_16x16x4_f64, `secondary.exact`)"},
                     "executed": {"flops_per_launch": m["exec_flops"], "tflops": m["exec_flops"] / (solve_ms.mean() * 1e-3) / 1e12,
                                  "frac_fp64_peak": m["exec_flops"] / (solve_ms.mean() * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                                  "note": "operations the solve kernel executes (bench.py executed_flops: OSQP on all 12 h variables through the 6 h x 6 h "
                                          "wrench-space core), for orientation only -- `achieved` / `frac` use the SURVEY 8(d) minimal-algorithm count"}},
    }
    if clock0 is not None:
        out["device_state"] = {"before": clock0, "after": clock1,
                               "note": "shader_clock_ghz: mpc_device_clock -- shader cycles / HIP-event time of ~20 ms of dependent fp64 FMAs, one wave per SIMD on every CU "
                                       "(the solve kernel's regime); boxes of the pool that sustain ~10 % less run every kernel of this line ~10 % slower"}
    if args.emulate:
        out["data"] = "synthetic; EMULATED kernels on the CPU (control-flow dry run: the numbers mean nothing)"
    if gather is not None:
        out["all_gather_torques"] = gather
    if args.emulate:
        print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return
    if world == 1 and not args.no_secondary and args.config == 2 and not args.robots:
        out["secondary"] = secondary_lines(dev)          # the other BASELINE configurations at their per-GPU sizes (N = 1 only)
    if world == 1 and not args.no_secondary and args.config == 2 and not args.robots:
        out["secondary"]["exact"] = exact_leg(m["wl"], m["batches"], W, h, dev)
        if seam == "ctrl":      # the bare compute_contact_forces batch on the same configuration (what `value` was before round 4)
            ms = run_leg(args.config, n, h, K, W, dev, 0, 1, None, repeats=3)
            out["secondary"]["solver_seam"] = {"robots": n, "horizon": h, "steps": K, "control_steps_per_s": n * K / ms["elapsed"], "ms_per_step": ms["elapsed"] / K * 1e3,
                                               "prep_kernel_ms": float(ms["prep_ms"].mean()), "solve_kernel_ms": float(ms["solve_ms"].mean()),
                                               "solved_fraction": float((ms["info"][..., 1] == 1).mean()), "mean_admm_iters": float(ms["info"][..., 0].mean()),
                                               "what": "config 2 through the bare solver seam: one compute_contact_forces batch per step (a2-a13), no estimator / controller / torque map"}
            del ms
    if not args.no_control_loop and world == 1:      # secondary legs: single-GPU runs only
        out["control_loop"] = control_loop_leg(n, h, dev)
        out["control_loop_with_resets"] = control_loop_leg(n, h, dev, reset_every=37)
        out["control_loop_mixed_phases"] = control_loop_leg(n, h, dev, reset_every=37, mixed=True)
        out["control_loop_exact"] = control_loop_leg(n, h, dev, solver="exact")      # the reference AS SHIPPED passes mpc.QPOASES (ConvexMPCLocomotion.py:108)
        out["control_loop_exact"]["note"] = "the same loop with the controllers' ConvexMpc objects in the exact-optimum mode (the reference's qpOASES branch, what its Python selects)"
        out["policy"] = policy_leg(n, dev)
        # what a training loop / the policy runner actually call (RL_Environment/tasks/aliengo.py:237-258, 321-334; robot_runner/RobotRunnerPolicy.py:62-92)
        out["env_bridge_loop"] = env_bridge_loop_leg(n, h, dev)
        out["env_bridge_loop"]["vs_control_loop"] = out["env_bridge_loop"]["robot_ticks_per_s"] / out["control_loop"]["robot_ticks_per_s"]
        out["env_bridge_loop_with_resets"] = env_bridge_loop_leg(n, h, dev, reset_every=37)
        out["env_bridge_loop_with_resets"]["vs_control_loop_with_resets"] = out["env_bridge_loop_with_resets"]["robot_ticks_per_s"] / out["control_loop_with_resets"]["robot_ticks_per_s"]
        out["runner_policy_loop"] = runner_policy_loop_leg(n, h, dev)
        # the same unit at the reference's own cadence (controller.run every 10 ms, MPC update on every 2nd call): two controller ticks per control step
        out["control_steps_per_s_incl_torque_map"] = out["control_loop"]["control_steps_per_s_incl_torque_map"]
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(m["wl"], m["batches"], W, h, gpu_first_forces=m["first_forces"], first_dof=m.get("first_dof"), robot_type=m.get("robot_type"))
        out["max_grf_err_vs_osqp"] = out["cpu_baseline"].pop("_gpu_err", None)
        out["max_abs_dtau_vs_osqp_Nm"] = out["cpu_baseline"].pop("_gpu_dtau", None)      # the torque map J^T (f_gpu - f_osqp) of the first timed step, sampled robots
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


TRAFFIC_PROFILE = "r06"      # profiles/r06_pmc_summary_h10.json + ..._prep_h10.json: the PMC passes that `roofline.traffic` quotes


def traffic_from_profile():
    """(bytes per step, parts) from the named PMC summaries, or (None, why): refused when a file is missing or was measured on other kernel sources."""
    from rl_mpc_locomotion_amd import _lib
    files = [os.path.join(ROOT, "profiles", f"{TRAFFIC_PROFILE}_pmc_summary_h10.json"), os.path.join(ROOT, "profiles", f"{TRAFFIC_PROFILE}_pmc_summary_prep_h10.json")]
    try:
        ds = [json.load(open(f)) for f in files]
    except (OSError, ValueError) as e:
        return None, {"refused": f"no PMC summary for this round ({e.__class__.__name__}: {files[0]})"}
    here = _lib.kernel_source_hash()
    if any(d.get("kernel_source_sha256") != here for d in ds):
        return None, {"refused": "the PMC summaries were measured on other kernel sources than this tree's (kernel_source_sha256 differs): re-run tools/pmc_passes.sh",
                      "from": [os.path.basename(f) for f in files]}
    t_solve, t_prep = (float(d["hbm_traffic_bytes_per_launch"]) for d in ds)
    return t_solve + t_prep, {"solve_kernel": t_solve, "prep_kernel": t_prep, "from": [os.path.basename(f) for f in files], "round": TRAFFIC_PROFILE,
                              "kernel_source_sha256": here}


def device_state(device, smi=True):
    """Shader clock under the solve kernel's regime (mpc_device_clock) and, best effort, what rocm-smi says about clocks / power / temperature."""
    from rl_mpc_locomotion_amd import _lib
    out = {}
    try:
        ghz, ms = _lib.device_clock(device, 20)
        out["shader_clock_ghz"] = ghz
        out["probe_ms"] = ms
    except Exception as e:      # never fail the bench line on the probe
        out["error"] = repr(e)
    if smi:
        try:
            r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20)
            j = json.loads(r.stdout)
            card = next(iter(j.values()))
            keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "power", "temperature (sensor junction)", "performance level"))}
            out["rocm_smi"] = keep
        except Exception as e:
            out["rocm_smi"] = {"unavailable": repr(e)[:120]}
    return out


def secondary_lines(dev, steps=5, warm=2):
    """Single-GPU lines of the other BASELINE configurations at their per-GPU shard sizes (config 3: 4096; config 4: 32768 / 8;
    config 5: 65536 / 8), so that the driver's N = 1 record carries them too.  Not `value`."""
    out = {}
    # config 3 AS BASELINE STATES IT: three robot types, Trot / Walk / Bound cycling every 50 steps -- 110 timed controller.run steps after 10 warm-up steps, i.e.
    # across the switches at steps 50 and 100 (parity of exactly this: tests/test_controller.py on controller_h10_cycling, minted from the unmodified reference)
    m = run_leg_ctrl(3, 4096, 10, 110, 10, dev, 0, 1, None)
    out["config3"] = {"robots": 4096, "horizon": 10, "steps": 110, "warmup": 10, "gait_switches_inside_the_timed_region": m["gait_switches_timed"], "seam": "ctrl",
                      "control_steps_per_s": 4096 * 110 / m["elapsed"], "ms_per_step": m["elapsed"] / 110 * 1e3,
                      "prep_kernel_ms": float(m["prep_ms"].mean()), "solve_kernel_ms": float(m["solve_ms"].mean()), "solved_fraction": float((m["info"][..., 1] == 1).mean()),
                      "mean_admm_iters": float(m["info"][..., 0].mean()),
                      "what": CONFIGS[3]["what"] + ": controller.run with every robot due, gait = (idx div 3 + step div 50) mod 3 over {TROT, WALK, BOUND} (SURVEY 8(d)), the "
                              "ids of each switch handed over as a device tensor (mpc_ctrl_set_gait_device)"}
    del m
    for cid, n in ((3, 4096), (4, 4096), (5, 8192)):
        h = CONFIGS[cid]["h"]
        m = run_leg(cid, n, h, steps, warm, dev, 0, 1, None)
        if cid == 3:
            out["config3_solver_seam"] = {"robots": n, "horizon": h, "steps": steps, "control_steps_per_s": n * steps / m["elapsed"], "ms_per_step": m["elapsed"] / steps * 1e3,
                                          "what": "config 3's robot / gait mix through the bare compute_contact_forces batch (no controller, no gait switch)"}
            del m
            continue
        out[f"config{cid}"] = {"robots": n, "horizon": h, "steps": steps, "control_steps_per_s": n * steps / m["elapsed"],
                               "ms_per_step": m["elapsed"] / steps * 1e3, "prep_kernel_ms": float(m["prep_ms"].mean()),
                               "solve_kernel_ms": float(m["solve_ms"].mean()), "solved_fraction": float((m["info"][..., 1] == 1).mean()),
                               "mean_admm_iters": float(m["info"][..., 0].mean()), "what": CONFIGS[cid]["what"]}
        del m
    # configs 4 / 5 through the controller.run seam too (a15-a23 with 16- / 20-segment gaits, ground normal from the estimator's fit, a22 included);
    # parity of that path: tests/test_controller.py on the controller_h16_* / controller_h20_* goldens
    for cid, n in ((4, 4096), (5, 8192)):
        h = CONFIGS[cid]["h"]
        m = run_leg_ctrl(cid, n, h, steps, warm, dev, 0, 1, None)
        out[f"config{cid}_ctrl"] = {"robots": n, "horizon": h, "steps": steps, "control_steps_per_s": n * steps / m["elapsed"],
                                    "ms_per_step": m["elapsed"] / steps * 1e3, "prep_kernel_ms": float(m["prep_ms"].mean()),
                                    "solve_kernel_ms": float(m["solve_ms"].mean()), "solved_fraction": float((m["info"][..., 1] == 1).mean()),
                                    "mean_admm_iters": float(m["info"][..., 0].mean()),
                                    "what": f"config {cid}'s robots and horizon through controller.run (every robot due): estimator, {h}-segment trot, foot placement, "
                                            "compute_contact_forces with the ground normal of the estimator's foot-contact fit, swing / stance commands, torque map"}
        del m
    return out


def exact_leg(wl, batches, W, h, dev, steps=5, sample=256):
    """The exact-optimum mode (MPC_SOLVER_EXACT: what the reference AS SHIPPED asks for, mpc.QPOASES, ConvexMPCLocomotion.py:108) on the
    headline workload: steps/s, kernel times, and on a sample of robots the error against the oracle's exact optimum (vendored OSQP, cold,
    eps 1e-9, polish) and the KKT conditions of the oracle-assembled QP.  Not `value`."""
    import torch
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    n = len(wl.mass)
    inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
    sv = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, device=dev, solver="exact")
    sv.enable_timing()
    d_in = [torch.from_numpy(batches[W + s]).to(dev) for s in range(steps)]
    for s in range(max(0, W - 2), W):      # the calls before the timed ones, in sequence (the active-set method starts from the previous call's working set)
        sv.solve(torch.from_numpy(batches[s]).to(dev))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in range(steps):
        f, info = sv.solve(d_in[s])
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    prep, solve = sv.kernel_times(steps)
    f, info = f.cpu().numpy(), info.cpu().numpy()
    out = {"robots": n, "horizon": h, "steps": steps, "control_steps_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3, "prep_kernel_ms": float(prep.mean()),
           "solve_kernels_ms": float(solve.mean()), "solved_fraction": float((info[:, 1] == 1).mean()), "working_set_changes_mean": float(info[:, 0].mean()),
           "working_set_changes_max": int(info[:, 0].max()),
           "what": "MPC_SOLVER_EXACT: dual active-set method + verified polish (first launch), ADMM route for robots it does not certify (second launch); the result is the "
                   "unique optimum whatever the start, the method's working set is seeded from the previous call's (MPC_EXACT_WARM=0: starts empty)"}
    try:
        from oracle.refmpc import RefConvexMpc
        from tests.helpers import kkt_certificate
        pick = np.arange(0, n, max(1, n // sample))[:sample]
        errs, kp, ks = [], 0.0, 0.0
        for r in pick:
            ref = RefConvexMpc(wl.mass[r], list(inertia9[r]), 4, h, wl.dt_mpc, wl.alpha)
            fx = ref.solve_exact(batches[W + steps - 1][r])
            errs.append(float(np.abs(f[r] - fx).max() / max(np.abs(fx).max(), 1.0)))
            if len(errs) <= 32:
                P, q, l, u, cone = ref.qp()
                pv, sr = kkt_certificate(P, q, cone, l, u, -f[r])
                kp, ks = max(kp, pv), max(ks, sr)
        out.update({"max_rel_err_vs_oracle_optimum": float(np.max(errs)), "sample": len(errs), "kkt_max_primal_violation": kp, "kkt_max_stationarity": ks})
    except Exception as e:      # the checker is optional here; never fail the bench line on it
        out["check_error"] = repr(e)
    return out


def all_gather_leg(n, n_total, dev, dist, reps=50, warm=5):
    """The optional exchange of SURVEY 8(e): RCCL all-gather of the per-robot torques ([n_local, 12] float32 per rank) on a side
    stream, timed with HIP events on that stream; reported separately, never part of `value`.  (CPU dry run: same calls over gloo,
    wall-clock timed.)"""
    import torch
    from rl_mpc_locomotion_amd.sharding import all_gather_torques
    local = torch.randn((n, 12), dtype=torch.float32, device=dev)
    if dev.type != "cuda":
        for _ in range(2):
            out = all_gather_torques(local, n_total)
        t0 = time.perf_counter()
        for _ in range(3):
            out = all_gather_torques(local, n_total)
        ms = (time.perf_counter() - t0) / 3 * 1e3
    else:
        side = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(side):
            for _ in range(warm):
                all_gather_torques(local, n_total)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps):
                out = all_gather_torques(local, n_total)
            e1.record(side)
        side.synchronize()
        ms = e0.elapsed_time(e1) / reps
    ok = bool(tuple(out.shape) == (n_total, 12))
    return {"ms": ms, "bytes_per_rank": n * 48, "robots_total": n_total, "shape_ok": ok,
            "note": "one all_gather_into_tensor of [n_local, 12] float32 per rank over RCCL / xGMI on a side stream; latency-bound"}


def peer_write_leg(n, n_total, rank, world, dev, dist, reps=50, warm=5):
    """--exchange peer | both: the same exchange as one-shot direct peer writes (sharding.PeerExchange), timed like all_gather_leg: put + wait per repetition on a side
    stream, HIP events on that stream, max over ranks.  Equal shards only (a block must start on a 16-byte boundary of the batch)."""
    import torch
    from rl_mpc_locomotion_amd.sharding import PeerExchange, max_over_ranks
    try:
        px = PeerExchange(n_total, rank * n, n, 12, device=dev)
        local = torch.randn((n, 12), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(side):
            for _ in range(warm):
                px.put(local); out = px.wait()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(reps):
                px.put(local); out = px.wait()
            e1.record(side)
        side.synchronize()
        ms = max_over_ranks(e0.elapsed_time(e1) / reps, dev)
        ok = bool(torch.equal(out[rank * n:(rank + 1) * n], local)) and px.timeouts() == 0
        dist.barrier()
        return {"ms": ms, "bytes_per_rank": n * 48, "own_block_ok_and_no_timeouts": ok,
                "note": "one put kernel (this rank's rows into every rank's receive region + epoch flag) and one wait kernel (flags of all ranks, copy out) per exchange"}
    except Exception as e:      # never fail the bench line on the optional variant
        return {"error": repr(e)[:300]}


def sharded_loop_leg(cfg_id, n_total, h, dev, dist, ticks=8, warm=3, emulate=False):
    """N > 1 only, NOT `value`: the product-level sharded stepper (sharding.ShardedLocomotion: per-rank controller on its block of ONE env batch,
    asynchronous all-gather of the torques on a side stream, read one tick later) -- ms per controller.run tick without and with the exchange,
    max over ranks.  The difference is what the all-gather costs when it overlaps the next tick's estimator / ctrl_pre / prep kernels."""
    import torch
    from rl_mpc_locomotion_amd.sharding import ShardedLocomotion, max_over_ranks
    from rl_mpc_locomotion_amd.synthetic import ControlStepStream
    cs = ControlStepStream(n_total, h=h, seed=2000, config=cfg_id)             # the WHOLE batch, identical on every rank; each rank runs its block
    kw = dict(controller_dt=CTRL_DT)
    if emulate:
        class Emu(_EmulatedLocomotion):
            def __init__(self, robot_type, gait_id, horizon=10, controller_dt=CTRL_DT):
                from tests.emu.emu import EmuLocomotion
                self.e = EmuLocomotion(robot_type, gait_id, horizon=horizon, controller_dt=controller_dt, nthreads=2)
                self.device = "cpu"
        kw["controller_factory"] = Emu
    else:
        kw["device"] = dev
    sl = ShardedLocomotion(cs.robot_type, cs.gait_id, horizon=h, **kw)
    ins = [tuple(torch.from_numpy(np.ascontiguousarray(a[sl.lo:sl.hi] if a.shape[0] == n_total else a)).to(dev) for a in cs.step(k)) for k in range(warm + ticks)]
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
    out = {}
    for label, with_gather in (("ms_per_tick_no_exchange", False), ("ms_per_tick_with_exchange", True)):
        sl.reset()
        got = None
        for k in range(warm + ticks):
            if k == warm:
                sync(); dist.barrier(); sync()
                t0 = time.perf_counter()
            if with_gather and k > 0:
                got = sl.torques_all()                      # last tick's exchange, read while this tick is being launched
            sl.run(*ins[k])
            if with_gather:
                sl.start_gather()
        sync(); dist.barrier()
        out[label] = max_over_ranks(time.perf_counter() - t0, dev) / ticks * 1e3
        if with_gather:
            out["shape_ok"] = bool(got is not None and tuple(got.shape) == (n_total, 12))
    # SURVEY 8(e)'s scaling check: the N-rank torques of the last tick, gathered over the collective, against the SAME env batch run by one
    # process on rank 0's device -- bit for bit; and what every rank held (a hash of its block, its device) so a record shows N devices took part
    import hashlib
    last = sl.torques_all()                                 # the exchange started after the last tick
    mine = last[sl.lo:sl.hi].contiguous().cpu().numpy()
    report = {"rank": sl.rank, "robots": [sl.lo, sl.hi], "torque_block_sha256": hashlib.sha256(mine.tobytes()).hexdigest()[:16],
              "device": (torch.cuda.get_device_name(dev) + f" #{dev.index}") if dev.type == "cuda" else "cpu (emulated kernels)"}
    if dev.type == "cuda":
        try:
            from rl_mpc_locomotion_amd import _lib
            report["shader_clock_ghz"] = _lib.device_clock(dev.index, 5)[0]
        except Exception as e:
            report["shader_clock_ghz"] = repr(e)[:80]
    reports = [None] * dist.get_world_size()
    dist.all_gather_object(reports, report)
    out["ranks"] = reports
    out["collective"] = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}
    if sl.rank == 0:
        if emulate:
            one = kw["controller_factory"](cs.robot_type, cs.gait_id, horizon=h, controller_dt=CTRL_DT)
        else:
            from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
            one = BatchedLocomotion(cs.robot_type, cs.gait_id, horizon=h, controller_dt=CTRL_DT, device=dev)
        cs.rewind()
        for k in range(warm + ticks):
            ref = one.run(*(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in cs.step(k)))
        same = bool(torch.equal(ref.to(last.device), last))
        out["bit_identical_to_single_process"] = same
        out["torque_blocks_sha256_single_process"] = [hashlib.sha256(ref[lo:hi].contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
                                                       for lo, hi in ((r["robots"][0], r["robots"][1]) for r in reports)]
        if not same:
            raise SystemExit(f"bench.py: the {dist.get_world_size()}-rank torques differ from the single-process batch (SURVEY 8(e) demands bit identity): {json.dumps(out)}")
    dist.barrier()
    out["robots_total"], out["ticks"] = n_total, ticks
    out["note"] = "ShardedLocomotion: one env batch over the ranks, all-gather of [n_local, 12] float32 on a side stream overlapped with the next tick; every robot due on every tick"
    return out


def control_loop_leg(n, h, dev, ticks=40, warm=10, reset_every=0, solver="osqp", mixed=False):
    """Secondary figure (NOT `value`): robot-ticks/s of the whole controller.run seam on device tensors --
    state estimator + leg kinematics + gait / foot placement + the MPC solve on every second tick (the
    reference's cadence, RobotRunnerMin.py:21-22) + swing / stance commands + joint torques.
    reset_every > 0: every that many ticks 1/64 of the robots are reset through a DEVICE tensor of indices, as VecTask.reset_idx does
    (RL_Environment/tasks/aliengo.py:321-334) -- after which the robots' MPC phases are no longer aligned and every tick has solves due.
    mixed: a random HALF of the robots is reset after an odd tick of the warm-up, so the two MPC phases hold ~n/2 robots each -- where a long training
    run with resets at arbitrary steps ends up (the 1/64 case above is the unfavourable start of that drift: a 64-robot job list costs a tick one robot's whole
    solve latency, a 2048-robot list costs it half a full launch)."""
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.synthetic import TickStream
    ts = TickStream(n, seed=4242, config=2)
    ctl = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=h, device=dev, solver=solver)
    ins = [tuple(torch.from_numpy(a).to(dev) for a in ts.tick(k)) for k in range(warm + ticks)]
    rng = np.random.default_rng(5)
    ids = [torch.from_numpy(rng.choice(n, max(1, n // 64), replace=False).astype(np.int32)).to(dev) for _ in range(warm + ticks)]
    for k in range(warm):
        ctl.run(*ins[k])
        if reset_every and (k + 1) % 7 == 0:
            ctl.reset(ids[k])
        if mixed and k == 4:
            ctl.reset(torch.from_numpy(rng.choice(n, n // 2, replace=False).astype(np.int32)).to(dev))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(warm, warm + ticks):
        ctl.run(*ins[k])
        if reset_every and (k + 1) % reset_every == 0:
            ctl.reset(ids[k])
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    solved = float((ctl.solver_info()[:, 1] == 1).mean())
    out = {"robot_ticks_per_s": n * ticks / dt, "ms_per_tick": dt / ticks * 1e3, "ticks": ticks, "mpc_every_n_ticks": 2,
           "control_steps_per_s_incl_torque_map": n * ticks / dt / 2,
           "solved_fraction_last_mpc": solved,
           "note": "controller.run for every robot per tick (estimator, gait, foot placement, MPC solve on every 2nd tick, swing / stance "
                   "commands, leg-torque map a22); robot_ticks_per_s is ~2x the control-step rate by construction"}
    if reset_every:
        out["reset_every_ticks"] = reset_every
        out["note"] = ("the same loop with reset(env_ids on the device) of n/64 robots every %d ticks (and a few during warm-up, so the robots' MPC phases "
                       "are mixed: solves are due on every tick)" % reset_every)
    if mixed:
        out["note"] += "; the two MPC phases hold ~half of the robots each (a random half was reset after an odd tick of the warm-up)"
    return out


def env_bridge_loop_leg(n, h, dev, ticks=40, warm=10, reset_every=0):
    """Secondary figure (NOT `value`): what an RL training loop calls per simulator step -- MpcEnvBridge.pre_physics_step(actions, dof_state, root_states, commands)
    (RL_Environment/tasks/aliengo.py:237-258: rescale of the policy's actions, command record, controller.run for every env) and, with reset_every > 0,
    reset_idx(env_ids on the device) of n/64 envs every that many ticks (:321-334).  Same tick stream and cadence as control_loop_leg (MPC on every 2nd tick); the
    actions are the ones whose rescale gives control_loop_leg's weights (so the two legs solve the same problems and differ by the glue only), everything resident in HBM."""
    import torch
    from rl_mpc_locomotion_amd.env_bridge import MpcEnvBridge
    from rl_mpc_locomotion_amd.synthetic import TickStream
    ts = TickStream(n, seed=4242, config=2)
    br = MpcEnvBridge(ts.robot_type, ts.gait_id, horizon=h, device=dev)
    from rl_mpc_locomotion_amd.weight_policy import MPC_PARAM_CONST, MPC_PARAM_SCALE
    rng = np.random.default_rng(5)
    actions = torch.from_numpy(((ts.w - np.array(MPC_PARAM_CONST, np.float32)) / np.array(MPC_PARAM_SCALE, np.float32)).astype(np.float32)).to(dev)
    ins = []
    for k in range(warm + ticks):
        dof, body, cmd = ts.tick(k)
        ins.append((actions, torch.from_numpy(dof.reshape(n * 12, 2)).to(dev),
                    torch.from_numpy(body).to(dev), torch.from_numpy(np.ascontiguousarray(cmd[:, :3])).to(dev)))
    ids = [torch.from_numpy(rng.choice(n, max(1, n // 64), replace=False).astype(np.int32)).to(dev) for _ in range(warm + ticks)]
    for k in range(warm):
        br.pre_physics_step(*ins[k])
        if reset_every and (k + 1) % 7 == 0:
            br.reset_idx(ids[k])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(warm, warm + ticks):
        br.pre_physics_step(*ins[k])
        if reset_every and (k + 1) % reset_every == 0:
            br.reset_idx(ids[k])
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    out = {"robot_ticks_per_s": n * ticks / dt, "ms_per_tick": dt / ticks * 1e3, "ticks": ticks, "mpc_every_n_ticks": 2,
           "note": "MpcEnvBridge.pre_physics_step per tick: one fused rescale + pack kernel (actions * MPC_param_scale + MPC_param_const, command record), then controller.run"}
    if reset_every:
        out["reset_every_ticks"] = reset_every
        out["note"] += "; reset_idx(device env_ids) of n/64 envs every %d ticks (and a few during warm-up: the MPC phases are mixed, compare control_loop_with_resets)" % reset_every
    return out


def runner_policy_loop_leg(n, h, dev, ticks=40, warm=10):
    """Secondary figure (NOT `value`): the batched RobotRunnerPolicy.run (robot_runner/RobotRunnerPolicy.py:62-92) per tick -- StateEstimator.update, observations from
    the fresh estimate and the previous weights, the 48-512-256-128-12 actor (random-init parameters of the reference architecture), command record, the control FSM in
    LOCOMOTION (MPC on every 2nd tick) -- BatchedLocomotion.run_policy."""
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.quadruped import ROBOT_TABLE64
    from rl_mpc_locomotion_amd.synthetic import TickStream
    from rl_mpc_locomotion_amd.weight_policy import WeightPolicy
    rng = np.random.default_rng(99)
    dims = [48, 512, 256, 128, 12]
    layers = [((rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32), np.zeros(dims[i + 1], np.float32)) for i in range(4)]
    pol = WeightPolicy(layers, device=dev)
    ts = TickStream(n, seed=4242, config=2)
    ctl = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=h, device=dev)
    ctl.fsm_init(np.full(n, BatchedLocomotion.LOCOMOTION), operating_mode=1, check_safety=True)
    req = torch.full((n,), BatchedLocomotion.LOCOMOTION, dtype=torch.int32, device=dev)
    ins = []
    for k in range(warm + ticks):
        dof, body, cmd = ts.tick(k)
        ins.append((torch.from_numpy(dof).to(dev), torch.from_numpy(body).to(dev), torch.from_numpy(np.ascontiguousarray(cmd[:, :3])).to(dev)))
    w = torch.from_numpy(np.tile(ROBOT_TABLE64[0, 12:24].astype(np.float32), (n, 1))).to(dev)      # Quadruped._mpc_weights[:-1] (RobotRunnerPolicy.py:44)
    for k in range(warm):
        _, w = ctl.run_policy(pol, *ins[k], w, req)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(warm, warm + ticks):
        _, w = ctl.run_policy(pol, *ins[k], w, req)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"robot_ticks_per_s": n * ticks / dt, "ms_per_tick": dt / ticks * 1e3, "ticks": ticks, "mpc_every_n_ticks": 2,
            "in_locomotion": float((ctl.fsm_state()[:, 0] == BatchedLocomotion.LOCOMOTION).mean()),
            "note": "BatchedLocomotion.run_policy per tick: estimator, observations (straight from the controller's estimate), fused MLP on the fp32 MFMA pipe, command "
                    "packing, fsm_pre, the solver kernels, fsm_post"}


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


def cpu_baseline(wl, batches, W, h, sample=2048, steps=4, gpu_first_forces=None, first_dof=None, robot_type=None):
    """The reference path (oracle/_ref: restated mpc_osqp.cc assembly + the vendored OSQP) timed on the
    host cores on a bounded sample of the same workload: the first `sample` robots, cold solve +
    `steps` timed warm solves (the same batches the GPU warmed up / timed on); all granted cores, then one core
    on a quarter of the sample."""
    from oracle.refmpc import RefBatch
    cores = usable_cores()
    sample = min(sample, len(wl.mass))
    steps = max(1, min(steps, len(batches) - W))      # (--steps smaller than the default sample length)
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
    err = dtau_max = None
    if gpu_first_forces is not None:
        ok = ~np.isnan(fr0[:, 0])
        g = gpu_first_forces[:sample]
        err = float((np.abs(g[ok, :12] - fr0[ok, :12]).max(1) / np.maximum(np.abs(fr0[ok, :12]).max(1), 1.0)).max())
        if first_dof is not None:      # "also report max-abs delta tau on the 12 torques": the stance legs' torque is J^T f_ff (LegController.py:108-132)
            from rl_mpc_locomotion_amd.synthetic import leg_jacobian
            J = leg_jacobian(np.asarray(first_dof)[:sample, :, 0].reshape(sample, 4, 3), np.asarray(robot_type)[:sample])
            df = (g[:, :12] - fr0[:, :12]).reshape(sample, 4, 3)
            dtau = np.einsum("nlij,nli->nlj", J, df)
            dtau_max = float(np.abs(dtau[ok]).max())
    # one core: SURVEY 8(d)(i)
    s1 = max(64, sample // 8)
    ref1 = RefBatch(wl.mass[:s1], wl.inertia_diag[:s1], h, wl.dt_mpc, wl.alpha)
    for s in range(W):
        ref1.solve(batches[s][:s1], nthreads=1)
    t1 = time.perf_counter()
    for s in range(steps):
        ref1.solve(batches[W + s][:s1], nthreads=1)
    dt1 = time.perf_counter() - t1
    out = {"_gpu_err": err, "_gpu_dtau": dtau_max, "value": sample * steps / dt, "unit": "control steps/s", "cores": cores, "kind": "reference",
           "sample": f"first {sample} robots of the workload, {W} warm-up + {steps} timed warm-started solves each, "
                     f"one OSQP workspace per robot, static partition over {cores} threads",
           "one_core": {"value": s1 * steps / dt1, "cores": 1, "sample": f"first {s1} robots, same sequence, one thread"}}
    tick = python_tick_baseline()
    if tick is not None:
        out["python_tick"] = tick
    return out


def python_tick_baseline(ticks=200):
    """the real Python path of BASELINE configs[0] -- one Aliengo, RobotRunnerMin.run, trot, h = 10 -- timed per tick with
    the oracle behind the reference's mpc_osqp seam.  Only where the reference tree is present (not on the GPU boxes)."""
    if not os.path.isdir("/root/reference/MPC_Controller"):
        # the GPU boxes have no reference tree: quote the record taken in the build container (profiles/r06_python_tick.json, which says where it was measured)
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "r06_python_tick.json")))
            rec["quoted_from"] = "profiles/r06_python_tick.json (measured in the build container, not on this box's host cores)"
            return rec
        except (OSError, ValueError):
            return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_golden_controller as mg
        return mg.time_reference_tick(ticks)
    except Exception as e:      # the baseline is optional; never fail the bench line on it
        return {"error": repr(e)}


if __name__ == "__main__":
    main()
