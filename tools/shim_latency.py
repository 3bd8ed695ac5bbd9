"""Per-call latency of the per-robot plugin seam (one robot per `compute_contact_forces`, host pointers) on the GPU box, with a breakdown:
  total            wall time of compute_contact_forces (ctypes module and pybind11 module)
  pack             the Python side of the ctypes module (13 arguments -> the float64 record)
  library call     mpc_batch_solve_host_f64 alone (two synchronous H2D copies, the launches, two synchronous D2H copies)
  kernels          prep + solve kernels from HIP events inside the library
  copies + sync    library call - kernels
usage: python tools/shim_latency.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rl-mpc-locomotion_amd", "pybind"))
import rl_mpc_locomotion_amd  # noqa: E402
from rl_mpc_locomotion_amd import _lib, layout as L, mpc_osqp as shim  # noqa: E402
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload  # noqa: E402
import mpc_osqp as pyb  # noqa: E402  (the pybind11 extension module)

wl = make_solver_workload(1, h=10, seed=3, config=2)
inertia = [float(wl.inertia_diag[0, 0]), 0, 0, 0, float(wl.inertia_diag[0, 1]), 0, 0, 0, float(wl.inertia_diag[0, 2])]


def args(rec, h=10):
    return [list(map(float, a)) for a in L.unpack_args(h, rec.astype(np.float64))]


for name, which_s, which_p in (("OSQP", shim.OSQP, pyb.OSQP), ("QPOASES (exact-optimum mode)", shim.QPOASES, pyb.QPOASES)):
    m = shim.ConvexMpc(float(wl.mass[0]), inertia, 4, 10, float(wl.dt_mpc), float(wl.alpha), which_s)
    p = pyb.ConvexMpc(float(wl.mass[0]), inertia, 4, 10, float(wl.dt_mpc), float(wl.alpha), which_p)
    _lib.check(_lib.lib().mpc_batch_enable_timing(m._handle), "timing")
    w = wl
    t_tot, t_pyb, t_pack, t_lib, t_k = [], [], [], [], []
    ka, kb = np.zeros(1, np.float32), np.zeros(1, np.float32)
    for k in range(110):
        a = args(w.inputs[0])
        t0 = time.perf_counter(); f = m.compute_contact_forces(*a); t_tot.append(time.perf_counter() - t0)
        assert len(f) == 120
        _lib.check(_lib.lib().mpc_batch_kernel_times(m._handle, 1, ka.ctypes.data, kb.ctypes.data), "times")
        t_k.append((float(ka[0]) + float(kb[0])) * 1e-3)
        t0 = time.perf_counter(); L.pack_args(10, *a, out=m._rec); t_pack.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        _lib.lib().mpc_batch_solve_host_f64(m._handle, m._rec.ctypes.data, m._out.ctypes.data, m.info.ctypes.data)
        t_lib.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); g = p.compute_contact_forces(*a); t_pyb.append(time.perf_counter() - t0)
        assert len(g) == 120
        w = perturb_workload(w, 50 + k)
    med = lambda v: float(np.median(np.array(v[10:]) * 1e3))
    print(f"{name}: compute_contact_forces median {med(t_tot):.3f} ms (ctypes module) / {med(t_pyb):.3f} ms (pybind11 module); "
          f"pack {med(t_pack):.3f}, library call {med(t_lib):.3f} = kernels {med(t_k):.3f} + copies / sync {med(t_lib) - med(t_k):.3f}")
