"""does eliminating the swing-leg variables (n_r = 3 x stance leg-steps, the SURVEY 8(d) "minimal
algorithm") keep the forces of the reference's OSQP solve?  CPU only.  For every robot of the parity-sweep workloads: the vendored
OSQP on the full QP (the reference, cold call) against the vendored OSQP -- same settings -- on the reduced QP.

    python tools/swing_elimination.py [robots per config]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl_mpc_locomotion_amd  # noqa: E402,F401
from rl_mpc_locomotion_amd.synthetic import make_solver_workload  # noqa: E402
from oracle.refmpc import RefConvexMpc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
out = {}
for name, cfg, h in (("config2_h10", 2, 10), ("config3_h10_mixed", 3, 10), ("config4_h16", 4, 16), ("config5_h20", 5, 20)):
    nn = n if h == 10 else max(n // 4, 32)
    wl = make_solver_workload(nn, h=h, seed=0, config=cfg)
    err, flips, its_full, its_red = [], 0, [], []
    for r in range(nn):
        d = wl.inertia_diag[r]
        ref = RefConvexMpc(wl.mass[r], [d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]], 4, h, wl.dt_mpc, wl.alpha)
        ff = ref.solve_flat(wl.inputs[r])
        fr = ref.solve_reduced(wl.inputs[r])
        if ff is None:
            continue
        err.append(np.abs(fr[:12] - ff[:12]).max() / max(np.abs(ff[:12]).max(), 1.0))
        flips += int((ref.info[:4] != ref.reduced_info[:4]).any())
        its_full.append(int(ref.info[0])); its_red.append(int(ref.reduced_info[0]))
    err = np.array(err)
    out[name] = dict(robots=len(err), frac_over_1e_3=float((err > 1e-3).mean()), frac_over_1e_2=float((err > 1e-2).mean()), median_rel_err=float(np.median(err)),
                     max_rel_err=float(err.max()), frac_decisions_differ=flips / len(err), mean_iters_full=float(np.mean(its_full)), mean_iters_reduced=float(np.mean(its_red)))
    print(name, json.dumps(out[name]), flush=True)
print("SWING_JSON " + json.dumps(out))
