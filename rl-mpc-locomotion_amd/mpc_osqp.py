"""Drop-in module for the reference's pybind11 extension ``mpc_osqp`` (mpc_osqp.cc:952-983).

Same names, argument meaning and error behaviour as the reference module, served by the HIP solver:

    import rl_mpc_locomotion_amd.mpc_osqp as mpc            # instead of `import mpc_osqp as mpc`
    cpp_mpc = mpc.ConvexMpc(mass, inertia9, 4, horizon, dt_mpc, alpha, mpc.QPOASES)
    forces = cpp_mpc.compute_contact_forces(w, pos, vel, rpy, normal, omega, table, feet, mu, dpos, dvel, drpy, domega)

``forces`` is a list of 12*h floats ([step][leg][xyz], negated like mpc_osqp.cc:789-790) or ``[]`` when
the solver does not report OSQP_SOLVED (mpc_osqp.cc:781-794).  ``qp_solver_name`` selects what the reference's
two branches return: ``OSQP`` -> OSQP 0.6.0's warm-started iterates at eps 1e-3 with polish (mpc_osqp.cc:690-796, BASELINE.json's
comparator); ``QPOASES`` -> the QP's optimum, cold on every call (mpc_osqp.cc:797-947 -- qpOASES is an empty submodule in
the reference, so its result, the unique optimum of the strictly convex QP, is reproduced rather than its iterations).
To serve the *unmodified* reference Python, put this module on ``sys.modules['mpc_osqp']`` before
importing ``MPC_Controller.convex_MPC.ConvexMPCLocomotion``.
"""
import ctypes as C
from enum import IntEnum

import numpy as np

from . import _lib
from .layout import in_len, pack_args


class QPSolverName(IntEnum):
    OSQP = 0
    QPOASES = 1


OSQP = QPSolverName.OSQP          # py::enum_::export_values() (mpc_osqp.cc:964-967)
QPOASES = QPSolverName.QPOASES
__version__ = "dev"               # mpc_osqp.cc:976-980
TEST = 42                         # mpc_osqp.cc:982


class ConvexMpc:
    def __init__(self, mass, inertia, num_legs, planning_horizon, timestep, alpha=1e-5, qp_solver_name=QPOASES):
        if num_legs != 4:
            raise ValueError("only quadrupeds (num_legs == 4) are supported")
        inertia = np.ascontiguousarray(inertia, dtype=np.float64).reshape(-1)
        if inertia.size != 9:
            raise ValueError("inertia must have 9 elements")   # assert at mpc_osqp.cc:556
        self._h = int(planning_horizon)
        self._handle = C.c_void_p()
        m = np.array([float(mass)])
        _lib.check(_lib.lib().mpc_batch_create(C.byref(self._handle), 1, self._h, float(timestep), float(alpha),
                                               m.ctypes.data, inertia.ctypes.data), "mpc_batch_create")
        self._exact = int(qp_solver_name) == int(QPOASES)
        _lib.check(_lib.lib().mpc_batch_set_solver(self._handle, 1 if self._exact else 0), "mpc_batch_set_solver")
        self._rec = np.zeros(in_len(self._h), dtype=np.float64)     # pybind11 widens every argument to double (mpc_osqp.cc:578-591)
        self._out = np.zeros(12 * self._h, dtype=np.float64)
        self.info = np.zeros(8, dtype=np.int32)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.lib().mpc_batch_destroy(h)
            except Exception:      # interpreter shutdown: the loader module may be gone already
                pass
            self._handle = None

    def compute_contact_forces(self, qp_weights, com_position, com_velocity, com_roll_pitch_yaw, ground_normal_vec,
                               com_angular_velocity, foot_contact_states, foot_positions_body_frame,
                               foot_friction_coeffs, desired_com_position, desired_com_velocity,
                               desired_com_roll_pitch_yaw, desired_com_angular_velocity):
        pack_args(self._h, qp_weights, com_position, com_velocity, com_roll_pitch_yaw, ground_normal_vec,
                  com_angular_velocity, foot_contact_states, foot_positions_body_frame, foot_friction_coeffs,
                  desired_com_position, desired_com_velocity, desired_com_roll_pitch_yaw,
                  desired_com_angular_velocity, out=self._rec)
        _lib.check(_lib.lib().mpc_batch_solve_host_f64(self._handle, self._rec.ctypes.data, self._out.ctypes.data,
                                                       self.info.ctypes.data), "mpc_batch_solve_host_f64")
        if self._exact:            # the qpOASES branch returns its vector whatever the solver's status (mpc_osqp.cc:906-947); only a
            return [] if self.info[1] == -7 else self._out.tolist()      # non-finite / non-convex problem has nothing to return
        if self.info[1] != 1:      # OSQP branch: not OSQP_SOLVED -> empty vector (mpc_osqp.cc:781-794)
            return []
        return self._out.tolist()

    def reset_solver(self):
        """mpc_osqp.cc:576: flips a flag that nothing reads -- a no-op there and here."""
