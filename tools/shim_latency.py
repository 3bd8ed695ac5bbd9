import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import rl_mpc_locomotion_amd
from rl_mpc_locomotion_amd import mpc_osqp as mpc
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
from rl_mpc_locomotion_amd import layout as L
wl = make_solver_workload(1, h=10, seed=3, config=2)
def args(rec, h=10):
    r = rec.astype(np.float64)
    o = [r[0:13], r[13:16], r[16:19], r[19:22], r[22:25], r[25:28], r[28:28+4*h]]
    p = 28+4*h
    o += [r[p:p+12], r[p+12:p+16], r[p+16:p+19], r[p+19:p+22], r[p+22:p+25], r[p+25:p+28]]
    return [list(map(float, a)) for a in o]
for name, which in (("OSQP", mpc.OSQP), ("QPOASES (exact-optimum mode)", mpc.QPOASES)):
  m = mpc.ConvexMpc(float(wl.mass[0]), [float(wl.inertia_diag[0,0]),0,0,0,float(wl.inertia_diag[0,1]),0,0,0,float(wl.inertia_diag[0,2])], 4, 10, float(wl.dt_mpc), float(wl.alpha), which)
  w = wl
  ts = []
  for k in range(110):
    a = args(w.inputs[0])
    t0 = time.perf_counter(); f = m.compute_contact_forces(*a); ts.append(time.perf_counter() - t0)
    assert len(f) == 120
    w = perturb_workload(w, 50 + k)
  ts = np.array(ts[10:]) * 1e3
  print("mpc_osqp shim, %s, 1 robot per call: median %.3f ms, mean %.3f, p90 %.3f ms" % (name, np.median(ts), ts.mean(), np.percentile(ts, 90)))
