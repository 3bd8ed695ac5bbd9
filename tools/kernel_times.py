import sys, numpy as np, torch, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl_mpc_locomotion_amd
from rl_mpc_locomotion_amd import _lib
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
h = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg = {10: 2, 16: 4, 20: 5}[h]
dev = torch.device("cuda:0")
wl = make_solver_workload(n, h=h, seed=1000, config=cfg)
K, W = 10, 3
batches = []; w = wl
for s in range(K + W):
    batches.append(w.inputs); w = perturb_workload(w, 7000 + 131*s)
d_in = [torch.from_numpy(b).to(dev) for b in batches]
inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
solver = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, device=dev)
solver.enable_timing()
for s in range(K + W): solver.solve(d_in[s])
torch.cuda.synchronize()
a, c = solver.kernel_times(K)
p = lambda x: x.ctypes.data_as(C.c_void_p)

print(f"h={h} n={n} prep {a.mean():.4f} ms  solve {c.mean():.4f} ms  (min {a.min():.4f} {c.min():.4f})  -> {n / (a.mean() + c.mean()) / 1e3:.3f} M steps/s kernel-time")

