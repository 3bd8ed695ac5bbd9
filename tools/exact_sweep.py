"""Exact-optimum mode (the reference's qpOASES branch) on the GPU over the SURVEY 8(d) workloads: n robots per config, two
seeds, four consecutive calls (the RESULT must not depend on the call before; the active-set method behind it starts from the previous call's working set), against
  (i) the KKT conditions of the oracle-assembled QP (tests/helpers.py kkt_certificate: no second solver involved), and
  (ii) the oracle's "exact" optimum (vendored OSQP, cold, eps 1e-9, polish -- itself only ~1e-8 .. 1e-6 accurate on this QP).
Prints one JSON record.   usage: python tools/exact_sweep.py [n_robots]"""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_mpc_locomotion_amd  # noqa
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
from oracle.refmpc import RefConvexMpc
from tests.helpers import kkt_certificate

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
out = {}
for name, cfg, h in (("config2_h10", 2, 10), ("config3_h10_mixed", 3, 10), ("config4_h16_normals", 4, 16), ("config5_h20", 5, 20)):
    errs, t0, unsolved, kkt_p, kkt_s, passes, second, p_first, p_later = [], time.time(), 0, [], [], [], 0, [], []
    for seed in (0, 1):
        wl = make_solver_workload(n, h=h, seed=seed, config=cfg)
        inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
        gpu = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha, solver="exact")
        refs = [RefConvexMpc(wl.mass[r], list(inertia9[r]), 4, h, wl.dt_mpc, wl.alpha) for r in range(n)]
        for step in range(4):
            f, info = gpu.solve(torch.from_numpy(wl.inputs).cuda()); torch.cuda.synchronize()
            f, info = f.cpu().numpy(), info.cpu().numpy()
            unsolved += int((info[:, 1] != 1).sum())
            with ThreadPoolExecutor(16) as ex:       # (the ctypes call releases the GIL)
                fx = np.array(list(ex.map(lambda r: refs[r].solve_exact(wl.inputs[r]), range(n))))
            errs.append(np.abs(f - fx).max(1) / np.maximum(np.abs(fx).max(1), 1.0))
            for r in range(0, n, max(1, n // 64)):      # (the certificate on a sample: it is a numpy / scipy loop)
                P, q, l, u, cone = refs[r].qp()
                pv, sr = kkt_certificate(P, q, cone, l, u, -f[r])
                kkt_p.append(pv); kkt_s.append(sr)
            (p_first if step == 0 else p_later).append(info[:, 0])
            passes.append(info[:, 0]); second += int((info[:, 3] > 0).sum() + (info[:, 0] > 8 * 4 * h).sum())
            wl = perturb_workload(wl, 9000 + 17 * step + seed)
    e = np.concatenate(errs)
    out[name] = dict(n=n, h=h, solves=int(e.size), unsolved_gpu=unsolved, max_rel_err=float(e.max()), p999_rel_err=float(np.percentile(e, 99.9)),
                     frac_below_1e_6=float((e < 1e-6).mean()), kkt_max_primal_violation=float(np.max(kkt_p)), kkt_max_stationarity=float(np.max(kkt_s)),
                     active_set_passes_mean=float(np.concatenate(passes).mean()), passes_first_call_mean=float(np.concatenate(p_first).mean()),
                     passes_seeded_calls_mean=float(np.concatenate(p_later).mean()), passes_seeded_calls_max=int(np.concatenate(p_later).max()), active_set_passes_max=int(np.concatenate(passes).max()),
                     robots_on_the_admm_route=second, seconds=time.time() - t0)
    print(name, json.dumps(out[name]), flush=True)
print("EXACT_JSON " + json.dumps(out))
