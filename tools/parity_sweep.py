"""GRF parity sweep on the GPU: the HIP solver vs the oracle (restated mpc_osqp.cc assembly + vendored OSQP)
over the SURVEY.md 8(d) workloads -- seeds 0..4 per config, cold solve + 2 warm-started solves.
Prints one JSON record (copied to profiles/ by the caller)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_mpc_locomotion_amd  # noqa
from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
from oracle.refmpc import RefBatch

cases = [("config2_h10_aliengo_trot", 2, 10, 4096), ("config3_h10_mixed", 3, 10, 4096), ("config4_h16_normals", 4, 16, 1024),
         ("config5_h20_normals", 5, 20, 256)]
seeds = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4]
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
out = {}
for name, cfg, h, n in cases:
    rec = dict(n=n, h=h, seeds=seeds, solves=0, polished=0, unsolved_ref=0, decision_mismatch=0, max_rel_err=0.0, max_rel_err_polished=0.0,
               max_rel_err_unpolished=0.0, frac_below_1e_6=0.0)
    errs = []
    t0 = time.time()
    for seed in seeds:
        wl = make_solver_workload(n, h=h, seed=seed, config=cfg)
        inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
        gpu = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha)
        ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
        for step in range(3):
            f, info = gpu.solve(torch.from_numpy(wl.inputs).cuda()); torch.cuda.synchronize()
            f = f.cpu().numpy(); info = info.cpu().numpy()
            fr = ref.solve(wl.inputs, nthreads=threads)
            ok = ref.info[:, 1] == 1
            rec["unsolved_ref"] += int((~ok).sum())
            rec["decision_mismatch"] += int((info[:, :4] != ref.info[:, :4]).any(1).sum())
            e = np.abs(f[ok, :12] - fr[ok, :12]).max(1) / np.maximum(np.abs(fr[ok, :12]).max(1), 1.0)
            pol = ref.info[ok, 2] == 1
            errs.append(e)
            rec["solves"] += n; rec["polished"] += int(pol.sum())
            if pol.any(): rec["max_rel_err_polished"] = max(rec["max_rel_err_polished"], float(e[pol].max()))
            if (~pol).any(): rec["max_rel_err_unpolished"] = max(rec["max_rel_err_unpolished"], float(e[~pol].max()))
            wl = perturb_workload(wl, 9000 + 17 * step + seed)
        del gpu, ref
    e = np.concatenate(errs)
    rec["max_rel_err"] = float(e.max()); rec["frac_below_1e_6"] = float((e < 1e-6).mean()); rec["p999_rel_err"] = float(np.percentile(e, 99.9))
    rec["seconds"] = time.time() - t0
    out[name] = rec
    print(name, json.dumps(rec), flush=True)
print("PARITY_JSON " + json.dumps(out))
