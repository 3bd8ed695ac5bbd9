"""Host-core scaling of the reference path (oracle/_ref) -- picks the thread count for cpu_baseline."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl_mpc_locomotion_amd  # noqa
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
from oracle.refmpc import RefBatch
n = 1024
wl = make_solver_workload(n, h=10, seed=1000, config=2)
w2 = perturb_workload(wl, 1); w3 = perturb_workload(w2, 2)
print("affinity cores", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for th in (1, 8, 32, 64, 128, 256):
    ref = RefBatch(wl.mass, wl.inertia_diag, 10, wl.dt_mpc, wl.alpha)
    ref.solve(wl.inputs, nthreads=th)
    t0 = time.perf_counter(); ref.solve(w2.inputs, nthreads=th); ref.solve(w3.inputs, nthreads=th); dt = time.perf_counter() - t0
    print(f"threads {th}: {2*n/dt:.0f} solves/s")
