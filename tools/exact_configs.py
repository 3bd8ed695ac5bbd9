import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rl_mpc_locomotion_amd
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
CASES = {3: (3, 10, 4096), 4: (4, 16, 4096), 5: (5, 20, 4096)}
for cfg, h, n in [CASES[int(a)] for a in (sys.argv[1:] or ["3", "4", "5"])]:
    wl = make_solver_workload(n, h=h, seed=1000, config=cfg)
    inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
    s = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, solver="exact")
    s.enable_timing()
    w = wl
    for k in range(8):
        x = torch.from_numpy(w.inputs).cuda(); torch.cuda.synchronize(); t0 = time.perf_counter()
        f, info = s.solve(x); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        w = perturb_workload(w, 7000 + 131 * k)
        ii = info.cpu().numpy()
        print("cfg", cfg, "step", k, "wall %.2f ms | solve kernels %.3f ms prep %.3f" % (wall, s.kernel_times(1)[1][-1], s.kernel_times(1)[0][-1]), "passes mean %.1f max %d" % (ii[:, 0].mean(), ii[:, 0].max()), "admm route", int((ii[:, 0] > 8 * 4 * h).sum() + (ii[:,3] > 0).sum()), "not solved", int((ii[:, 1] != 1).sum()))
