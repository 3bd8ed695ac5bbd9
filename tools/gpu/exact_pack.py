"""Exact mode: per-robot cycles of the first launch against the launch time (how much the one-workgroup-per-robot launch loses to packing)."""
import os, sys, heapq
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rl_mpc_locomotion_amd  # noqa
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
n, h = 4096, 10
wl = make_solver_workload(n, h=h, seed=1000, config=2)
inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
sv = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, solver="exact")
sv.enable_timing()
w = wl
hist = []
for s in range(14):
    f, info = sv.solve(torch.from_numpy(w.inputs).cuda()); torch.cuda.synchronize()
    cyc = np.abs(sv.get_profile()[:, 15]).astype(np.float64)
    ms = float(sv.kernel_times(1)[-1][-1])
    slots = 1024
    def lpt(order):
        hp = [0.0] * slots; heapq.heapify(hp)
        for r in order: heapq.heappush(hp, heapq.heappop(hp) + cyc[r])
        return max(hp)
    ghz = 2.09
    line = f"step {s}: kernel {ms:.3f} ms | work/slot {cyc.sum() / slots / ghz / 1e6:.3f} ms | longest robot {cyc.max() / ghz / 1e6:.3f} ms | clairvoyant LPT {lpt(np.argsort(-cyc)) / ghz / 1e6:.3f} | robot order {lpt(np.arange(n)) / ghz / 1e6:.3f}"
    if hist:
        line += f" | LPT by the previous call {lpt(np.argsort(-hist[-1])) / ghz / 1e6:.3f}"
        line += f" | by the max of the last 3 {lpt(np.argsort(-np.max(hist[-3:], axis=0))) / ghz / 1e6:.3f} | of the last 10 {lpt(np.argsort(-np.max(hist[-10:], axis=0))) / ghz / 1e6:.3f}"
    print(line)
    hist.append(cyc)
    w = perturb_workload(w, 7000 + 131 * s)
