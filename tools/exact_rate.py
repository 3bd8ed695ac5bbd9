import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import rl_mpc_locomotion_amd
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
n, h = 4096, 10
wl = make_solver_workload(n, h=h, seed=1000, config=2)
inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
for mode in ("osqp", "exact"):
    s = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, solver=mode)
    s.enable_timing()
    w = wl
    for k in range(8):
        f, info = s.solve(torch.from_numpy(w.inputs).cuda()); w = perturb_workload(w, 7000 + 131 * k)
        torch.cuda.synchronize()
        ii = info.cpu().numpy()
        print("   step", k, "solve kernel %.3f ms" % s.kernel_times(1)[1][-1], "iters mean %.1f max %d" % (ii[:, 0].mean(), ii[:, 0].max()), "unsolved", int((ii[:, 1] != 1).sum()), "rho updates max", int(ii[:, 3].max()), "nfact max", int(ii[:, 4].max()))
    a, c = s.kernel_times(5)
    info = info.cpu().numpy()
    print(mode, "prep %.3f solve %.3f ms" % (a.mean(), c.mean()), "iters mean %.1f max %d" % (info[:, 0].mean(), info[:, 0].max()), "polish ok %.3f" % (info[:, 2] == 1).mean(), "nfact %.2f" % info[:, 4].mean(), "-> %.2f M steps/s" % (n / (a.mean() + c.mean()) / 1e3))
