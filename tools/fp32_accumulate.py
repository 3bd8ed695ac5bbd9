"""BASELINE configs[4] asks for "fp16 state / fp32 accumulate QP".  The kernels store fp16 / fp32 records but compute in fp64; this script is the evidence for
that choice, regenerated on the current tree: OSQP's algorithm in the dense scalar restatement (oracle/osqp_dense_port.c), built once with REAL = double and once with
REAL = float (the SAME source: every product, sum and division of the iteration, the factorisation, the scaling and the polish in float32), against the reference's vendored
OSQP (oracle/_ref/libosqp_ref.so) on SURVEY 8(d)'s workloads -- cold solve + two warm-started solves per robot.  Reported per configuration: the fraction of solves whose
first-step forces miss BASELINE's bar (1e-3 relative to max(|f_osqp|_inf, 1 N)), the fraction whose OSQP decisions (iterations, status, polish, rho updates) differ, and
the same for the double build (which must be ~0: it is the control).  CPU only:  python tools/fp32_accumulate.py  -> profiles/r06_fp32_accumulate.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rl_mpc_locomotion_amd  # noqa: E402,F401
from oracle.port import PortBatch  # noqa: E402
from oracle.refmpc import RefBatch  # noqa: E402
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload  # noqa: E402

out = {}
for cfg, h, n in ((2, 10, 256), (4, 16, 128), (5, 20, 128)):
    wl = make_solver_workload(n, h=h, seed=2024 + cfg, config=cfg)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ports = {p: PortBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha, precision=p) for p in ("f64", "f32")}
    stats = {p: dict(solves=0, over=0, differ=0, errs=[]) for p in ports}
    w = wl
    for step in range(3):
        rec = w.inputs
        if cfg == 5:
            rec = rec.astype(np.float16).astype(np.float32)      # "fp16 state": every solver and the reference see the same float16-valued records
        fr = ref.solve(rec, nthreads=8)
        ok = ref.info[:, 1] == 1
        for p, pb in ports.items():
            f = pb.solve(rec, nthreads=8)
            err = np.abs(f[:, :12] - fr[:, :12]).max(1) / np.maximum(np.abs(fr[:, :12]).max(1), 1.0)
            dec = (pb.info[:, :4] != ref.info[:, :4]).any(1)
            st = stats[p]
            st["solves"] += int(ok.sum()); st["over"] += int((err[ok] > 1e-3).sum()); st["differ"] += int(dec[ok].sum()); st["errs"].extend(err[ok].tolist())
        w = perturb_workload(w, 31 + step)
    out[f"config{cfg}_h{h}"] = {p: {"solves": st["solves"], "frac_over_1e_3": st["over"] / st["solves"], "frac_decisions_differ": st["differ"] / st["solves"],
                                      "max_rel_err": float(np.max(st["errs"])), "median_rel_err": float(np.median(st["errs"]))} for p, st in stats.items()}
    print(f"config{cfg}_h{h}", json.dumps(out[f"config{cfg}_h{h}"]))
out["what"] = ("oracle/osqp_dense_port.c as REAL = float (f32: fp32 accumulate) and REAL = double (f64: the control) against the reference's vendored OSQP, first-step GRF error "
               "relative to max(|f|_inf, 1 N); cold + two warm-started solves per robot; config 5's records rounded to float16 first")
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_fp32_accumulate.json"), "w"), indent=1)
