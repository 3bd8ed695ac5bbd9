import sys, os, numpy as np, torch, time
sys.path.insert(0, os.getcwd())
import rl_mpc_locomotion_amd
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
n, h = 4096, 10
wl = make_solver_workload(n, h=h, seed=1000, config=2)
inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
batches=[]; w=wl
for s in range(40):
    batches.append(torch.from_numpy(w.inputs).cuda()); w=perturb_workload(w, 7000+131*s)
for sync in (True, False):
    sv = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, solver="exact"); sv.enable_timing()
    for s in range(40):
        f, info = sv.solve(batches[s])
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    a, c = sv.kernel_times(40)
    print("sync", sync, "solve ms per step:", np.round(c, 2).tolist())
