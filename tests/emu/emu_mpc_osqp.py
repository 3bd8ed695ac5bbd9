"""TEST ONLY: a module with the reference's `mpc_osqp` interface (mpc_osqp.cc:952-983) served by the HOST EMULATION of the
device kernels (tests/emu/emu.cpp) -- what lets the CPU test-suite run the UNMODIFIED reference Python on the kernel's
algorithm through the very seam INTEGRATION.md describes (sys.modules["mpc_osqp"] = this module).  The product's own module
with this interface is rl_mpc_locomotion_amd.mpc_osqp (HIP library, GPU only)."""
import numpy as np

from rl_mpc_locomotion_amd.layout import in_len, pack_args
from tests.emu.emu import EmuBatch

import os

OSQP, QPOASES = 0, 1
__version__ = "dev"
TEST = 42


class ConvexMpc:
    def __init__(self, mass, inertia, num_legs, planning_horizon, timestep, alpha=1e-5, qp_solver_name=QPOASES):
        assert num_legs == 4
        inertia = np.asarray(inertia, dtype=np.float64).reshape(3, 3)
        self._h = int(planning_horizon)
        self._emu = EmuBatch(np.array([float(mass)]), np.array([[inertia[0, 0], inertia[1, 1], inertia[2, 2]]]), self._h, float(timestep), float(alpha))
        self._rec = np.zeros((1, in_len(self._h)), dtype=np.float32)
        # qp_solver_name selects the result like in the product module (rl_mpc_locomotion_amd.mpc_osqp): QPOASES -> the exact-optimum mode.
        # EMU_SHIM_SOLVER=osqp forces the OSQP branch (the goldens minted from the vendored OSQP behind the seam).
        self._exact = int(qp_solver_name) == QPOASES and os.environ.get("EMU_SHIM_SOLVER", "") != "osqp"

    def compute_contact_forces(self, *args):
        pack_args(self._h, *args, out=self._rec[0])
        f = self._emu.solve(self._rec, nthreads=1, exact=self._exact)
        st = int(self._emu.info[0, 1])
        if self._exact:
            return [] if st == -7 else f[0].tolist()       # that branch returns its vector whatever the status (mpc_osqp.cc:906-947)
        return [] if st != 1 else f[0].tolist()

    def reset_solver(self):
        pass
