"""Seeded synthetic workloads for BASELINE.json's configs (SURVEY.md 8(d) "Synthetic inputs").

Produces the float32 ``[N, 56+4h]`` solver-input records (layout.py) plus the per-robot model
constants, exactly as the reference's Python layer would marshal them at
``ConvexMPCLocomotion.py:128-185`` for the sampled robot states.  Used by bench.py and the tests; it
contains no solver code.
"""
import numpy as np

from . import layout as L
from .gait import mpc_table
from .quadruped import (COL_ABAD, COL_HEIGHT, COL_HIP, COL_HIPLOC, COL_KNEE, COL_MU, ROBOT_TABLE, ROBOT_TABLE64,
                        COL_MASS, COL_INERTIA, SIDE_SIGN, RobotType)

# Parameters.py:25-33 (action -> weight rescale of the RL bridge)
MPC_PARAM_SCALE = np.array([4, 4, 4, 20, 20, 20, 1, 1, 1, 1, 1, 1], dtype=np.float32)
MPC_PARAM_CONST = np.array([5, 5, 5, 50, 50, 50, 1, 1, 1, 1, 1, 1], dtype=np.float32)


def leg_fk(q, robot_type):
    """Foot position in the hip frame, LegController.computeLegJacobianAndPosition (LegController.py:135-154).

    q: [N,4,3] joint angles; robot_type: [N] int.  The reference evaluates this in Python floats
    (double) and stores float32.
    """
    q = np.asarray(q, dtype=np.float64)
    rt = np.asarray(robot_type)
    dy = ROBOT_TABLE64[rt, COL_ABAD][:, None] * SIDE_SIGN[None, :].astype(np.float64)   # link lengths are Python floats there
    dz1 = -ROBOT_TABLE64[rt, COL_HIP][:, None]
    dz2 = -ROBOT_TABLE64[rt, COL_KNEE][:, None]
    s1, s2, s3 = np.sin(q[..., 0]), np.sin(q[..., 1]), np.sin(q[..., 2])
    c1, c2, c3 = np.cos(q[..., 0]), np.cos(q[..., 1]), np.cos(q[..., 2])
    c23 = c2 * c3 - s2 * s3
    s23 = s2 * c3 + c2 * s3
    p = np.stack([dz2 * s23 + dz1 * s2,
                  dy * c1 - dz1 * c2 * s1 - dz2 * s1 * c23,
                  dy * s1 + dz1 * c1 * c2 + dz2 * c1 * c23], axis=-1)
    return p.astype(np.float32)


def leg_jacobian(q, robot_type):
    """[N,4,3,3] leg Jacobians, LegController.computeLegJacobianAndPosition (LegController.py:155-171): the torque map of
    LegController.updateCommand is tau_leg = J^T (f_ff + ...) (LegController.py:108-132)."""
    q = np.asarray(q, dtype=np.float64)
    rt = np.asarray(robot_type)
    dy = ROBOT_TABLE64[rt, COL_ABAD][:, None] * SIDE_SIGN[None, :].astype(np.float64)
    dz1 = -ROBOT_TABLE64[rt, COL_HIP][:, None]
    dz2 = -ROBOT_TABLE64[rt, COL_KNEE][:, None]
    s1, s2, s3 = np.sin(q[..., 0]), np.sin(q[..., 1]), np.sin(q[..., 2])
    c1, c2, c3 = np.cos(q[..., 0]), np.cos(q[..., 1]), np.cos(q[..., 2])
    c23 = c2 * c3 - s2 * s3
    s23 = s2 * c3 + c2 * s3
    J = np.zeros(q.shape[:2] + (3, 3))
    J[..., 1, 0] = -dy * s1 - dz2 * c1 * c23 - dz1 * c1 * c2
    J[..., 2, 0] = -dz2 * s1 * c23 + dy * c1 - dz1 * c2 * s1
    J[..., 0, 1] = dz2 * c23 + dz1 * c2
    J[..., 1, 1] = dz2 * s1 * s23 + dz1 * s1 * s2
    J[..., 2, 1] = -dz2 * c1 * s23 - dz1 * c1 * s2
    J[..., 0, 2] = dz2 * c23
    J[..., 1, 2] = dz2 * s1 * s23
    J[..., 2, 2] = -dz2 * c1 * s23
    return J


def hip_locations(robot_type):
    """[N,4,3] float32 hip locations, Quadruped.getHipLocation (Quadruped.py:96-107)."""
    rt = np.asarray(robot_type)
    loc = ROBOT_TABLE[rt, COL_HIPLOC:COL_HIPLOC + 3]                  # [N,3]
    sx = np.array([1, 1, -1, -1], dtype=np.float32)
    sy = np.array([1, -1, 1, -1], dtype=np.float32)
    out = np.empty((len(rt), 4, 3), dtype=np.float32)
    out[..., 0] = loc[:, None, 0] * sx
    out[..., 1] = loc[:, None, 1] * sy
    out[..., 2] = loc[:, None, 2]
    return out


class Workload:
    """Solver-boundary workload: records + per-robot model constants."""

    def __init__(self, h, inputs, robot_type, gait_id, iteration_counter, dt_mpc, alpha):
        self.h = h
        self.inputs = inputs                  # float32 [N, 56+4h]
        self.robot_type = robot_type          # int32 [N]
        self.gait_id = gait_id                # int32 [N]
        self.iteration_counter = iteration_counter
        self.dt_mpc = dt_mpc
        self.alpha = alpha
        self.mass = ROBOT_TABLE64[robot_type, COL_MASS]                         # float64 [N]
        self.inertia_diag = ROBOT_TABLE64[robot_type, COL_INERTIA:COL_INERTIA + 3]  # float64 [N,3]

    @property
    def n(self):
        return self.inputs.shape[0]


def make_solver_workload(n, h=10, seed=0, config=2, iterations_between_mpc=2, controller_dt=0.01,
                         alpha=1e-5, step_index=0):
    """per-robot random state -> the 13 arguments of compute_contact_forces.

    config 1/2: Aliengo, trot, normal=(0,0,1).  config 3: robot type = idx mod 3 over
    {Go1,A1,Aliengo}, gait = (idx div 3 + step_index div 50) mod 3 over {TROT,WALK,BOUND}.
    config 4/5: Aliengo trot with random ground normals (use h=16 / h=20).
    """
    rng = np.random.default_rng(seed)
    idx = np.arange(n)
    if config == 3:
        robot_type = np.array([RobotType.GO1, RobotType.A1, RobotType.ALIENGO], dtype=np.int32)[idx % 3]
        gait_id = np.array([0, 6, 1], dtype=np.int32)[(idx // 3 + step_index // 50) % 3]
    else:
        robot_type = np.full(n, int(RobotType.ALIENGO), dtype=np.int32)
        gait_id = np.zeros(n, dtype=np.int32)
    H = ROBOT_TABLE[robot_type, COL_HEIGHT]

    rpy = np.stack([rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n), rng.uniform(-np.pi, np.pi, n)], -1)
    rpy = rpy.astype(np.float16).astype(np.float32)        # orientation_tools.py:13 -> float16 rpy
    pos = np.zeros((n, 3), dtype=np.float32)
    pos[:, 2] = H * rng.uniform(0.9, 1.05, n)
    omega = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    vel = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.1, 0.1, n)], -1).astype(np.float32)
    q = np.array([0.0, 0.8, -1.6])[None, None, :] + rng.uniform(-0.2, 0.2, (n, 4, 3))
    foot = hip_locations(robot_type) + leg_fk(q, robot_type)          # ConvexMPCLocomotion.py:248-249
    cmd = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(-1, 1, n), rng.uniform(-2.5, 2.5, n)], -1).astype(np.float32)
    w = np.zeros((n, 13), dtype=np.float32)
    w[:, :12] = MPC_PARAM_CONST + rng.uniform(-1, 1, (n, 12)).astype(np.float32) * MPC_PARAM_SCALE
    it = rng.integers(0, 2 * h, n).astype(np.int32)
    if config in (4, 5):
        nv = np.stack([rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), np.ones(n)], -1)
        normal = (nv / np.linalg.norm(nv, axis=1, keepdims=True)).astype(np.float32)
    else:
        normal = np.tile(np.array([0, 0, 1], dtype=np.float32), (n, 1))

    rec = np.zeros((n, L.in_len(h)), dtype=np.float32)
    rec[:, L.IN_WEIGHTS:L.IN_WEIGHTS + 13] = w
    rec[:, L.IN_COM_POS:L.IN_COM_POS + 3] = pos
    rec[:, L.IN_COM_VEL:L.IN_COM_VEL + 3] = vel
    rec[:, L.IN_RPY:L.IN_RPY + 3] = rpy
    rec[:, L.IN_NORMAL:L.IN_NORMAL + 3] = normal
    rec[:, L.IN_ANGVEL:L.IN_ANGVEL + 3] = omega
    rec[:, L.IN_CONTACT:L.IN_CONTACT + 4 * h] = mpc_table(gait_id, it, iterations_between_mpc, h)
    rec[:, L.in_footpos(h):L.in_footpos(h) + 12] = foot.reshape(n, 12)
    rec[:, L.in_friction(h):L.in_friction(h) + 4] = ROBOT_TABLE[robot_type, COL_MU][:, None]
    rec[:, L.in_des_pos(h) + 2] = H                                   # ConvexMPCLocomotion.py:151
    rec[:, L.in_des_vel(h):L.in_des_vel(h) + 2] = cmd[:, :2]          # :152
    rec[:, L.in_des_angvel(h) + 2] = cmd[:, 2]                        # :154
    return Workload(h, rec, robot_type, gait_id, it, controller_dt * iterations_between_mpc, alpha)


def perturb_workload(wl, seed, scale=0.05):
    """A follow-up solve for the same robots (exercises the warm-start path): advance the gait
    counter by iterations_between_mpc and jitter the state a little."""
    rng = np.random.default_rng(seed)
    h, n = wl.h, wl.n
    rec = wl.inputs.copy()
    for off, k, s in ((L.IN_COM_VEL, 3, 0.1), (L.IN_ANGVEL, 3, 0.05), (L.in_footpos(h), 12, 0.01)):
        rec[:, off:off + k] += (rng.standard_normal((n, k)) * s * scale / 0.05).astype(np.float32)
    rpy = rec[:, L.IN_RPY:L.IN_RPY + 3] + (rng.standard_normal((n, 3)) * 0.01).astype(np.float32)
    rec[:, L.IN_RPY:L.IN_RPY + 3] = rpy.astype(np.float16).astype(np.float32)
    it = (wl.iteration_counter + 2).astype(np.int32)
    rec[:, L.IN_CONTACT:L.IN_CONTACT + 4 * h] = mpc_table(wl.gait_id, it, 2, h)
    return Workload(h, rec, wl.robot_type, wl.gait_id, it, wl.dt_mpc, wl.alpha)


GAIT_CYCLE_STEPS = 50      # "gait = (idx div 3 + step div 50) mod 3 over {TROT, WALK, BOUND}"


def config3_gait(n, step):
    """Per-robot ``Parameters.cmpc_gait`` of BASELINE configs[2] at control step `step`: Trot / Walk / Bound CYCLING every 50 steps (the reference re-reads the
    parameter on every tick, ConvexMPCLocomotion.py:224-244; ids TROT 0, WALK 6, BOUND 1, utils.py:17-24)."""
    return np.array([0, 6, 1], dtype=np.int32)[(np.arange(n) // 3 + int(step) // GAIT_CYCLE_STEPS) % 3]


class _GaitSchedule:
    def gait_at(self, step):
        """the gait ids controller.run must see at `step` (None: the gait of this configuration never changes)"""
        return config3_gait(self.n, step) if self.config == 3 else None

    def gait_switch_at(self, step):
        """True when the ids of gait_at(step) differ from those of step - 1 (the caller hands them to set_gait before that step's run)"""
        return self.config == 3 and step > 0 and step % GAIT_CYCLE_STEPS == 0


class TickStream(_GaitSchedule):
    """Seeded, smooth open-loop (dof_states, body_states, commands) signals for N robots, shaped like the RL
    bridge's per-tick inputs (RL_Environment/tasks/aliengo.py:246-256).  Same construction as
    tests/golden/make_golden_controller.py, vectorised."""

    def __init__(self, n, seed=0, config=2):
        rng = np.random.default_rng(seed)
        idx = np.arange(n)
        if config == 3:
            self.robot_type = np.array([RobotType.GO1, RobotType.A1, RobotType.ALIENGO], dtype=np.int32)[idx % 3]
            self.gait_id = config3_gait(n, 0)
        else:
            self.robot_type = np.full(n, int(RobotType.ALIENGO), dtype=np.int32)
            self.gait_id = np.zeros(n, dtype=np.int32)
        self.config = config
        self.n = n
        self.phase = rng.uniform(0, 2 * np.pi, (n, 21))
        self.amp = rng.uniform(0.02, 0.15, (n, 21))
        self.yaw0 = rng.uniform(-3, 3, n)
        self.H = ROBOT_TABLE[self.robot_type, COL_HEIGHT] * rng.uniform(0.9, 1.05, n)
        self.v0 = rng.uniform(-0.5, 0.5, (n, 3)) * np.array([1, 0.4, 0.1])
        self.cmd = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-1.0, 1.0, n)], -1)
        self.w = MPC_PARAM_CONST + rng.uniform(-1, 1, (n, 12)).astype(np.float32) * MPC_PARAM_SCALE

    def tick(self, k, dt=0.01):
        t, ph, amp, n = dt * k, self.phase, self.amp, self.n
        q = np.tile([0.0, 0.8, -1.6], 4)[None] + amp[:, :12] * np.sin(2 * np.pi * 1.3 * t + ph[:, :12])
        qd = amp[:, :12] * 2 * np.pi * 1.3 * np.cos(2 * np.pi * 1.3 * t + ph[:, :12])
        dof = np.stack([q, qd], axis=2).astype(np.float32)
        rpy = 0.12 * np.sin(2 * np.pi * 0.7 * t + ph[:, 12:15])
        rpy[:, 2] += self.yaw0 + 0.4 * t
        cy, sy, cp, sp, cr, sr = (np.cos(rpy[:, 2] / 2), np.sin(rpy[:, 2] / 2), np.cos(rpy[:, 1] / 2), np.sin(rpy[:, 1] / 2),
                                  np.cos(rpy[:, 0] / 2), np.sin(rpy[:, 0] / 2))
        body = np.zeros((n, 13), dtype=np.float32)
        body[:, 0] = 0.3 * t
        body[:, 2] = self.H
        body[:, 3:7] = np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                                 cr * cp * cy + sr * sp * sy], -1)
        body[:, 7:10] = self.v0 + 0.2 * np.sin(2 * np.pi * 0.5 * t + ph[:, 15:18])
        body[:, 10:13] = 0.3 * np.sin(2 * np.pi * 0.9 * t + ph[:, 18:21])
        cmd = np.zeros((n, 16), dtype=np.float32)
        cmd[:, 0:3] = self.cmd
        cmd[:, 3:15] = self.w
        return dof, body, cmd


class ControlStepStream(_GaitSchedule):
    """SURVEY.md 8(d)'s synthetic robot states at the BATCH seam -- (dof_states, body_states, commands) of ``controller.run``
    (RL_Environment/tasks/aliengo.py:246-256) instead of the 13 solver arguments: the same distributions as make_solver_workload
    (roll / pitch U(-0.15, 0.15), yaw U(-pi, pi), height H U(0.9, 1.05), omega U(-0.5, 0.5)^3, v (U(-1.5, 1.5), U(-0.5, 0.5), U(-0.1, 0.1)),
    joints = stand pose + U(-0.2, 0.2), commands (U(-2.5, 2.5), U(-1, 1), U(-2.5, 2.5)), weights = MPC_param_const + U(-1, 1) MPC_param_scale,
    gait counter U{0 .. h-1} MPC steps), and from step to step the random walk of perturb_workload.  The controller derives the solver's
    arguments itself (state estimator, leg kinematics, gait table), as the reference's does.  step(s) must be called for s = 0, 1, 2 ... in
    order (or rewind() first)."""

    def __init__(self, n, h=10, seed=0, config=2):
        rng = np.random.default_rng(seed)
        idx = np.arange(n)
        if config == 3:
            self.robot_type = np.array([RobotType.GO1, RobotType.A1, RobotType.ALIENGO], dtype=np.int32)[idx % 3]
            self.gait_id = config3_gait(n, 0)
        else:
            self.robot_type = np.full(n, int(RobotType.ALIENGO), dtype=np.int32)
            self.gait_id = np.zeros(n, dtype=np.int32)
        self.config = config
        self.n, self.h, self.seed = n, h, seed
        H = ROBOT_TABLE[self.robot_type, COL_HEIGHT]
        self.rpy0 = np.stack([rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n), rng.uniform(-np.pi, np.pi, n)], -1)
        self.z = H * rng.uniform(0.9, 1.05, n)
        self.omega0 = rng.uniform(-0.5, 0.5, (n, 3))
        self.vel0 = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(-0.1, 0.1, n)], -1)
        self.q0 = np.tile([0.0, 0.8, -1.6], 4)[None] + rng.uniform(-0.2, 0.2, (n, 12))
        self.cmd = np.zeros((n, 16), dtype=np.float32)
        self.cmd[:, 0:3] = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(-1, 1, n), rng.uniform(-2.5, 2.5, n)], -1)
        self.cmd[:, 3:15] = MPC_PARAM_CONST + rng.uniform(-1, 1, (n, 12)).astype(np.float32) * MPC_PARAM_SCALE
        self.iteration0 = rng.integers(0, h, n).astype(np.int32)      # ConvexMPCLocomotion.iterationCounter at the first step (one MPC step per tick)
        self.rewind()

    def rewind(self):
        self._s = 0
        self._rpy, self._omega, self._vel, self._q = self.rpy0.copy(), self.omega0.copy(), self.vel0.copy(), self.q0.copy()

    def step(self, s):
        assert s == self._s, "ControlStepStream.step: steps in order (rewind() to start again)"
        n = self.n
        if s > 0:
            rng = np.random.default_rng((self.seed + 1) * 100003 + s)
            self._vel += rng.standard_normal((n, 3)) * 0.1
            self._omega += rng.standard_normal((n, 3)) * 0.05
            self._q += rng.standard_normal((n, 12)) * 0.03
            self._rpy += rng.standard_normal((n, 3)) * 0.01
        self._s += 1
        rng2 = np.random.default_rng((self.seed + 7) * 7919 + s)
        dof = np.stack([self._q, rng2.standard_normal((n, 12)) * 0.5], axis=2).astype(np.float32)
        r, p, y = self._rpy[:, 0], self._rpy[:, 1], self._rpy[:, 2]
        cy, sy, cp, sp, cr, sr = np.cos(y / 2), np.sin(y / 2), np.cos(p / 2), np.sin(p / 2), np.cos(r / 2), np.sin(r / 2)
        body = np.zeros((n, 13), dtype=np.float32)
        body[:, 2] = self.z
        body[:, 3:7] = np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], -1)
        body[:, 7:10] = self._vel
        body[:, 10:13] = self._omega
        return dof, body, self.cmd
