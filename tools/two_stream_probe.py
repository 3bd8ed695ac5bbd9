"""Probe: one 4096-robot batch as one launch vs two 2048-robot halves on two streams (tail / assembly overlap)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_mpc_locomotion_amd  # noqa
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload

n, h, K, W = 4096, 10, 20, 3
wl = make_solver_workload(n, h=h, seed=1000, config=2)
batches = [wl.inputs]
w = wl
for s in range(K + W - 1):
    w = perturb_workload(w, 7000 + s); batches.append(w.inputs)
inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
def one():
    sv = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha)
    d = [torch.from_numpy(b).cuda() for b in batches]
    for s in range(W): sv.solve(d[s])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(K): sv.solve(d[W + s])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
def split(parts, join=False):
    m = n // parts
    svs = [BatchedConvexMpc(wl.mass[i*m:(i+1)*m], inertia9[i*m:(i+1)*m], h, wl.dt_mpc, wl.alpha) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    d = [[torch.from_numpy(np.ascontiguousarray(b[i*m:(i+1)*m])).cuda() for b in batches] for i in range(parts)]
    outs = [[None] * (K + W) for _ in range(parts)]
    def step(s):
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                outs[i][s] = svs[i].solve(d[i][s])
        if join:   # a closed-loop step: nothing of step s + 1 may start before all of step s is done (GPU-side join, no host sync)
            evs = [st.record_event() for st in streams]
            for st in streams:
                for e in evs: st.wait_event(e)
    torch.cuda.synchronize()
    for s in range(W): step(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(K): step(W + s)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
print("one launch: %.3f ms/step; two halves on two streams: %.3f; four quarters: %.3f; two halves joined after every step: %.3f; four joined: %.3f" % (one(), split(2), split(4), split(2, True), split(4, True)))
