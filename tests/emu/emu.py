"""ctypes wrapper of the host emulation of the device solver (TEST ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
        _LIB = C.CDLL(os.path.join(_HERE, "_build", "libmpc_emu.so"))
        _LIB.emu_batch_solve.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return _LIB


class EmuBatch:
    def __init__(self, mass, inertia_diag, h, dt, alpha):
        n = len(mass)
        self.n, self.h, self.dt, self.alpha = n, h, dt, alpha
        self.model = np.zeros((n, 10))
        self.model[:, 0] = mass
        self.model[:, 1] = inertia_diag[:, 0]
        self.model[:, 5] = inertia_diag[:, 1]
        self.model[:, 9] = inertia_diag[:, 2]
        self.state = np.zeros((n, lib().emu_state_len(h)))
        self.info = np.zeros((n, 8), dtype=np.int32)
        self.phases = np.zeros(n, dtype=np.int64)
        self.seed = np.zeros((n, 4 * h), dtype=np.int32)      # exact mode: the working sets the previous call ended on (mpc_batch's d_seed)
        self.warm_sets = True

    def solve(self, records, reverse=False, nthreads=8, exact=False):
        """exact: the exact-optimum mode (mpc_batch_set_solver(MPC_SOLVER_EXACT)): cold on every call (its RESULT; the working set of the
        previous call seeds the active-set method unless warm_sets is False)."""
        if exact:
            self.state[:] = 0.0
            lib().emu_set_seed_buffer.argtypes = [C.c_void_p, C.c_int]
            lib().emu_set_seed_buffer(self.seed.ctypes.data_as(C.c_void_p) if self.warm_sets else None, 4 * self.h)
        rec = np.ascontiguousarray(records, dtype=np.float32)
        out = np.full((self.n, 12 * self.h), np.nan)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib().emu_batch_solve(self.h, self.n, p(self.model), self.dt, self.alpha, p(rec), p(self.state), p(out),
                                   p(self.info), int(reverse) | (2 if exact else 0), nthreads, p(self.phases))
        assert rc == 0
        return out


def ctrl_replay(robot_type, gait_id, flat_ground, dof, est, cmd, dt=0.01, iters_between_mpc=2, alpha=1e-5, horizon=10):
    """Host emulation of ctrl_pre -> solve -> ctrl_post over a recorded input sequence."""
    from rl_mpc_locomotion_amd.gait import gait_arrays
    from rl_mpc_locomotion_amd.quadruped import ROBOT_TABLE64
    L = lib()
    T, n = dof.shape[0], dof.shape[1]
    h = int(horizon)
    off, dur = gait_arrays(h)
    tab = np.ascontiguousarray(ROBOT_TABLE64, dtype=np.float64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rt = np.ascontiguousarray(robot_type, dtype=np.int32); gi = np.ascontiguousarray(gait_id, dtype=np.int32)
    off = np.ascontiguousarray(off, dtype=np.int32); dur = np.ascontiguousarray(dur, dtype=np.int32)
    dof = np.ascontiguousarray(dof, dtype=np.float32); est = np.ascontiguousarray(est, dtype=np.float32); cmd = np.ascontiguousarray(cmd, dtype=np.float32)
    tau = np.zeros((T, n, 12), np.float32); rec = np.zeros((T, n, 56 + 4 * h), np.float32); fff = np.zeros((T, n, 12), np.float32)
    L.emu_ctrl_replay.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_double, C.c_int, C.c_double] + [C.c_void_p] * 6
    rc = L.emu_ctrl_replay(h, n, T, p(tab), p(rt), p(gi), p(off), p(dur), int(flat_ground), dt, iters_between_mpc, alpha,
                           p(dof), p(est), p(cmd), p(tau), p(rec), p(fff))
    assert rc == 0
    return tau, rec, fff


class EmuLocomotion:
    """The library's per-tick controller (mpc_ctrl_create / _run / _reset: BatchedLocomotion) on the host emulation, as a stepping object:
    run(dof [n,12,2], body [n,13], cmd [n,16]) -> torques [n,12] (numpy float32)."""

    def __init__(self, robot_type, gait_id, horizon=10, controller_dt=0.01, alpha=1e-5, flat_ground=False, solver="osqp", iterations_between_mpc=None, nthreads=8):
        import rl_mpc_locomotion_amd  # noqa: F401
        from rl_mpc_locomotion_amd.gait import gait_arrays
        from rl_mpc_locomotion_amd.quadruped import ROBOT_TABLE64
        L = lib()
        self.n, self.h, self.exact, self.nthreads = len(robot_type), int(horizon), int(solver == "exact"), nthreads
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        off, dur = gait_arrays(self.h)
        off = np.ascontiguousarray(off, dtype=np.int32); dur = np.ascontiguousarray(dur, dtype=np.int32)
        tab = np.ascontiguousarray(ROBOT_TABLE64, dtype=np.float64)
        rt = np.ascontiguousarray(robot_type, dtype=np.int32); gi = np.ascontiguousarray(gait_id, dtype=np.int32)
        iters = int(27 / (1000.0 * controller_dt)) if iterations_between_mpc is None else int(iterations_between_mpc)
        L.emu_ctrl_open.restype = C.c_void_p
        L.emu_ctrl_open.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_double, C.c_int, C.c_double]
        L.emu_ctrl_run.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int]
        L.emu_ctrl_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.emu_ctrl_set_iteration.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_ctrl_get.argtypes = [C.c_void_p] * 4
        L.emu_ctrl_close.argtypes = [C.c_void_p]
        self._h = L.emu_ctrl_open(self.n, self.h, p(tab), p(rt), p(gi), p(off), p(dur), int(bool(flat_ground)), float(controller_dt), iters, float(alpha))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().emu_ctrl_close(self._h)
            self._h = None

    def run(self, dof, body, cmd):
        dof = np.ascontiguousarray(dof, dtype=np.float32); body = np.ascontiguousarray(body, dtype=np.float32); cmd = np.ascontiguousarray(cmd, dtype=np.float32)
        assert dof.size == self.n * 24 and body.size == self.n * 13 and cmd.size == self.n * 16
        tau = np.zeros((self.n, 12), np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        assert lib().emu_ctrl_run(self._h, p(dof), p(body), p(cmd), p(tau), self.exact, self.nthreads) == 0
        return tau

    def reset(self, env_ids=None):
        if env_ids is None:
            lib().emu_ctrl_reset(self._h, None, 0)
        else:
            ids = np.ascontiguousarray(env_ids, dtype=np.int32)
            lib().emu_ctrl_reset(self._h, ids.ctypes.data_as(C.c_void_p), len(ids))

    def set_iteration(self, it):
        it = np.ascontiguousarray(it, dtype=np.int32)
        lib().emu_ctrl_set_iteration(self._h, it.ctypes.data_as(C.c_void_p))

    def set_gait(self, gait_id):
        gi = np.ascontiguousarray(gait_id, dtype=np.int32)
        assert gi.shape == (self.n,)
        lib().emu_ctrl_set_gait.argtypes = [C.c_void_p, C.c_void_p]
        lib().emu_ctrl_set_gait(self._h, gi.ctypes.data_as(C.c_void_p))

    def solver_info(self):
        out = np.zeros((self.n, 8), np.int32)
        lib().emu_ctrl_get(self._h, out.ctypes.data_as(C.c_void_p), None, None)
        return out

    def solver_forces(self):
        out = np.zeros((self.n, 12 * self.h))
        lib().emu_ctrl_get(self._h, None, out.ctypes.data_as(C.c_void_p), None)
        return out

    def solver_record(self):
        out = np.zeros((self.n, 56 + 4 * self.h), np.float32)
        lib().emu_ctrl_get(self._h, None, None, out.ctypes.data_as(C.c_void_p))
        return out

    def estimate(self):
        """(ground_normal_yaw [n,3], foot contact history [n,4,3], CoM height [n]) after the last run (StateEstimator.py:99-143)."""
        nrm = np.zeros((self.n, 3), np.float32); hist = np.zeros((self.n, 4, 3), np.float32); z = np.zeros(self.n, np.float32)
        lib().emu_ctrl_get_estimate.argtypes = [C.c_void_p] * 4
        lib().emu_ctrl_get_estimate(self._h, nrm.ctypes.data_as(C.c_void_p), hist.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p))
        return nrm, hist, z


def estimator_update(body, normal):
    L = lib()
    body = np.ascontiguousarray(body, dtype=np.float32); normal = np.ascontiguousarray(normal, dtype=np.float32)
    n = body.shape[0]
    est = np.zeros((n, 18), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.emu_estimator_update.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.emu_estimator_update(n, p(body), p(normal), p(est))
    return est


def fsm_replay(robot_type, gait_id, init_mode, dof, body, cmd, request, flat_ground=False, dt=0.01, iters_between_mpc=2, alpha=1e-5,
               check_safety=True, op_mode=1):
    """RobotRunnerFSM.run for every robot and tick (host emulation, horizon 10).
    dof [T,n,12,2], body [T,n,13], cmd [T,n,16], request [T,n] -> torques [T,n,12], fsm [T,n,3] (state, op mode, recovery flag)."""
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.gait import gait_arrays
    from rl_mpc_locomotion_amd.quadruped import ROBOT_TABLE64
    L = lib()
    T, n = dof.shape[0], dof.shape[1]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    tab = np.ascontiguousarray(ROBOT_TABLE64, dtype=np.float64)
    rt = np.ascontiguousarray(robot_type, dtype=np.int32); gi = np.ascontiguousarray(gait_id, dtype=np.int32)
    im = np.ascontiguousarray(init_mode, dtype=np.int32); rq = np.ascontiguousarray(request, dtype=np.int32)
    off, dur = gait_arrays(10)
    off = np.ascontiguousarray(off, dtype=np.int32); dur = np.ascontiguousarray(dur, dtype=np.int32)
    dof = np.ascontiguousarray(dof, dtype=np.float32); body = np.ascontiguousarray(body, dtype=np.float32); cmd = np.ascontiguousarray(cmd, dtype=np.float32)
    tau = np.zeros((T, n, 12), np.float32); fsm = np.zeros((T, n, 3), np.int32)
    L.emu_fsm_replay.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 7
    rc = L.emu_fsm_replay(n, T, p(tab), p(rt), p(gi), p(off), p(dur), int(flat_ground), dt, iters_between_mpc, alpha, int(check_safety), int(op_mode),
                          p(im), p(dof), p(body), p(cmd), p(rq), p(tau), p(fsm))
    assert rc == 0
    return tau, fsm
