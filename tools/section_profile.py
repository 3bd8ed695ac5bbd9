"""Per-section shader-cycle breakdown of the solve kernel (in-kernel s_memtime laps), 4096 robots.

The section counters are compiled out of the product library; build and select an instrumented copy first:
  tools/build_variant.sh prof -DMPC_SECTION_PROFILE
  MPC_LIB_PATH=rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_prof.so python tools/section_profile.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_mpc_locomotion_amd  # noqa
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
h = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = make_solver_workload(n, h=h, seed=1000, config={10: 2, 16: 4, 20: 5}[h])
inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
solver = sys.argv[3] if len(sys.argv) > 3 else "osqp"      # or "exact"
sv = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, solver=solver)
names = ["load", "dyn", "qP", "sc-load", "sc-loop", "sc-store", "Kform", "sweep", "admm", "r-mulP", "r-rest", "p-setup", "p-H", "p-refine", "p-fin", "total"]
if os.environ.get("PROF_NAMES"): names = os.environ["PROF_NAMES"].split(",")
w = wl
for step in range(4):
    d = torch.from_numpy(w.inputs).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f, info = sv.solve(d)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = np.abs(sv.get_profile()).astype(np.float64); i = info.cpu().numpy()
    print(f"step {step}: {dt*1e3:.2f} ms  iters {i[:,0].mean():.1f} nfact {i[:,4].mean():.2f} | mean kcycles/robot: " +
          " ".join(f"{nm}={p[:,k].mean()/1e3:.0f}" for k, nm in enumerate(names)) +
          f" | per-iter admm {p[:,8].sum()/i[:,0].sum():.0f} cyc, per-sweep {p[:,7].sum()/np.maximum(i[:,4]-1,1).sum():.0f} cyc")
    w = perturb_workload(w, 7000 + step)
