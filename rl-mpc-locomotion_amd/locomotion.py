"""Batched per-tick controller: N ``RobotRunnerMin``-style controllers behind one handle.

Host-side mirror of the reference's runner for the part of ``RobotRunnerMin.run``
(robot_runner/RobotRunnerMin.py:54-75) that follows ``StateEstimator.update``:
``LegController.updateData`` -> ``ConvexMPCLocomotion.run`` (with the MPC solve every
``iterationsBetweenMPC``-th tick) -> ``LegController.updateCommand``.  Per-robot gait and robot type replace
the process-global ``Parameters.cmpc_gait`` / one-runner-per-type of the reference.
"""
import ctypes as C

import numpy as np

from . import _lib, gym_states
from .gait import gait_arrays
from .quadruped import ROBOT_TABLE64


class BatchedLocomotion:
    def __init__(self, robot_type, gait_id, horizon=10, controller_dt=0.01, alpha=1e-5, flat_ground=False, device=None, solver="osqp",
                 iterations_between_mpc=None):
        """solver: "osqp" (the reference's OSQP branch, BASELINE's comparator) or "exact" (its qpOASES branch -- what the shipped
        Python passes, ConvexMPCLocomotion.py:108 -- the QP's optimum, cold every call); see BatchedConvexMpc.
        iterations_between_mpc: the second constructor argument of ConvexMPCLocomotion (ConvexMPCLocomotion.py:58); None = what
        RobotRunnerMin passes, int(27 / (1000 controller_dt)) (RobotRunnerMin.py:21-22: 2 at the reference's controller_dt = 0.01)."""
        import torch
        if not torch.cuda.is_available():
            raise _lib.MpcLibraryError("BatchedLocomotion needs a GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        rt = np.ascontiguousarray(robot_type, dtype=np.int32)
        gi = np.ascontiguousarray(gait_id, dtype=np.int32)
        self.n, self.h = len(rt), int(horizon)
        iters = int(27 / (1000.0 * controller_dt)) if iterations_between_mpc is None else int(iterations_between_mpc)   # RobotRunnerMin.py:21-22
        self.iterations_between_mpc = iters
        off, dur = gait_arrays(self.h)
        off = np.ascontiguousarray(off, dtype=np.int32); dur = np.ascontiguousarray(dur, dtype=np.int32)
        tab = np.ascontiguousarray(ROBOT_TABLE64, dtype=np.float64)
        self._handle = C.c_void_p()
        _lib.check(_lib.lib().mpc_ctrl_create(C.byref(self._handle), self.n, self.h, float(controller_dt), iters, float(alpha),
                                              int(bool(flat_ground)), rt.ctypes.data, gi.ctypes.data, tab.shape[0], tab.ctypes.data,
                                              off.ctypes.data, dur.ctypes.data), "mpc_ctrl_create")
        if solver not in ("osqp", "exact"):
            raise ValueError("solver must be 'osqp' or 'exact'")
        _lib.check(_lib.lib().mpc_ctrl_set_solver(self._handle, 1 if solver == "exact" else 0), "mpc_ctrl_set_solver")
        self.torques = torch.zeros((self.n, 12), dtype=torch.float32, device=self.device)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h and _lib is not None and _lib._LIB is not None:
            _lib._LIB.mpc_ctrl_destroy(h)
            self._handle = None

    def step(self, dof_states, est, commands, torques=None):
        """dof_states [N,12,2] (or [N*12,2]), est [N,18], commands [N,16]: contiguous cuda float32.
        Returns torques [N,12] float32 (FL FR RL RR x hip, thigh, calf)."""
        import torch
        commands = self._full_commands(commands)
        for name, t, numel in (("dof_states", dof_states, self.n * 24), ("est", est, self.n * 18), ("commands", commands, self.n * 16)):
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != numel:
                raise ValueError(f"{name} must be a contiguous cuda float32 tensor with {numel} elements")
        torques = self.torques if torques is None else torques
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().mpc_ctrl_step(self._handle, dof_states.data_ptr(), est.data_ptr(), commands.data_ptr(),
                                            torques.data_ptr(), stream), "mpc_ctrl_step")
        return torques

    def run(self, dof_states, body_states, commands, torques=None):
        """The batched ``controller.run(dof_states, body_states, commands)`` of the reference loop
        (RL_Environment/tasks/aliengo.py:252-256): dof_states [N,12,2], body_states [N,13] (pos3, quat xyzw,
        lin vel3, ang vel3, world frame), commands [N,16]; returns torques [N,12].  Also takes the interactive runners'
        container (RL_MPC_Locomotion.py:96-101): Isaac Gym's structured host arrays `dof_states["pos" / "vel"]`,
        `body_states["pose"]["r"]` ... and a [3] / [N,3] command (gym_states.py)."""
        import torch
        dof_states, body_states, commands = self._inputs(dof_states, body_states, commands)
        commands = self._full_commands(commands)
        for name, t, numel in (("dof_states", dof_states, self.n * 24), ("body_states", body_states, self.n * 13), ("commands", commands, self.n * 16)):
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != numel:
                raise ValueError(f"{name} must be a contiguous cuda float32 tensor with {numel} elements")
        torques = self.torques if torques is None else torques
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().mpc_ctrl_run(self._handle, dof_states.data_ptr(), body_states.data_ptr(), commands.data_ptr(),
                                           torques.data_ptr(), stream), "mpc_ctrl_run")
        return torques

    def _inputs(self, dof_states, body_states, commands):
        """The reference's two input containers (gym_states.py): device tensors in the RL-bridge layout pass through; Isaac Gym's structured
        host arrays (`dof_states["pos"]`, `body_states["pose"]["r"]` ..., the interactive runners) and host float arrays are converted and
        uploaded."""
        import torch
        if gym_states.is_structured(dof_states):
            dof_states = gym_states.dof_states_to_array(dof_states)
        if body_states is not None and gym_states.is_structured(body_states):
            body_states = gym_states.body_states_to_array(body_states)
        up = lambda x: x if (x is None or hasattr(x, "is_cuda")) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(self.device)
        commands = up(commands)
        if commands.dim() == 1:
            commands = commands.reshape(1, -1).expand(self.n, -1).contiguous()      # one command for every robot (the viewer loop's `commands`)
        return up(dof_states), up(body_states), commands

    def _full_commands(self, commands):
        """[N, 3] commands (vx, vy, yaw rate: the interactive runners, mpc_weights None) -> [N, 16] with NaN weights, which the
        controller replaces by the robot type's Quadruped._mpc_weights (ConvexMPCLocomotion.py:132-135)."""
        import torch
        if commands.dim() == 2 and commands.shape[1] == 3:
            full = torch.full((self.n, 16), float("nan"), dtype=torch.float32, device=commands.device)
            full[:, :3] = commands
            return full
        return commands

    def reset(self, env_ids=None):
        import torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if env_ids is None:
            _lib.check(_lib.lib().mpc_ctrl_reset(self._handle, None, 0, stream), "mpc_ctrl_reset")
            return
        if hasattr(env_ids, "is_cuda") and env_ids.is_cuda:      # an env_ids tensor on the device (VecTask.reset_idx): no host round trip
            d_ids = env_ids.to(device=self.device, dtype=torch.int32).contiguous()
            _lib.check(_lib.lib().mpc_ctrl_reset_device(self._handle, d_ids.data_ptr(), d_ids.numel(), stream), "mpc_ctrl_reset_device")
            return
        ids = np.ascontiguousarray(env_ids.detach().cpu().numpy() if hasattr(env_ids, "detach") else env_ids, dtype=np.int32)
        _lib.check(_lib.lib().mpc_ctrl_reset(self._handle, ids.ctypes.data, len(ids), stream), "mpc_ctrl_reset")

    def set_gait(self, gait_id):
        """``Parameters.cmpc_gait`` (Parameters.py:17) per robot, effective from the next ``run`` -- which re-reads it on every tick like the
        reference's (ConvexMPCLocomotion.py:224-244), so the gait may change DURING a run: iterationCounter, firstSwing, swingTimeRemaining and
        the swing trajectories carry over.  A cuda int32 tensor [n] is taken as it is, stream-ordered (no host round trip, no synchronisation:
        BASELINE configs[2] cycles Trot / Walk / Bound every 50 steps); a host array is validated, copied and waited for."""
        import torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if hasattr(gait_id, "is_cuda") and gait_id.is_cuda:
            if gait_id.dtype != torch.int32 or not gait_id.is_contiguous() or gait_id.numel() != self.n:
                raise ValueError("gait_id on the device must be a contiguous int32 tensor with one entry per robot")
            _lib.check(_lib.lib().mpc_ctrl_set_gait_device(self._handle, gait_id.data_ptr(), stream), "mpc_ctrl_set_gait_device")
            return
        gi = np.ascontiguousarray(gait_id.cpu().numpy() if hasattr(gait_id, "cpu") else gait_id, dtype=np.int32)
        if len(gi) != self.n:
            raise ValueError("gait_id must have one entry per robot")
        _lib.check(_lib.lib().mpc_ctrl_set_gait(self._handle, gi.ctypes.data, stream), "mpc_ctrl_set_gait")
        torch.cuda.current_stream(self.device).synchronize()      # (`gi` is a host buffer)

    # ---- control FSM (RobotRunnerFSM) --------------------------------------------------------------------
    PASSIVE, LOCOMOTION, RECOVERY_STAND = 0, 4, 6          # FSM_StateName (MPC_Controller/utils.py:26-30)

    def fsm_init(self, control_mode, operating_mode=1, check_safety=True):
        """``RobotRunnerFSM.init`` for every robot: fresh controller objects and ``ControlFSM.initialize`` into
        ``control_mode[r]`` (per-robot ``Parameters.control_mode``); operating_mode 0 TEST / 1 NORMAL."""
        import torch
        cm = np.ascontiguousarray(control_mode, dtype=np.int32)
        if len(cm) != self.n:
            raise ValueError("control_mode must have one entry per robot")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().mpc_ctrl_fsm_init(self._handle, cm.ctypes.data, int(operating_mode), int(bool(check_safety)), stream), "mpc_ctrl_fsm_init")

    def run_fsm(self, dof_states, body_states, commands, request, torques=None, estimated=False):
        """The batched ``RobotRunnerFSM.run(dof_states, body_states, commands)`` (robot_runner/RobotRunnerFSM.py:44-71);
        ``request`` [N] cuda int32 is the control mode requested for each robot this tick.  estimated: ``update_estimate(body_states)`` has been
        called on the same body_states this tick (run_policy): StateEstimator.update is not run a second time."""
        import torch
        dof_states, body_states, commands = self._inputs(dof_states, body_states, commands)
        commands = self._full_commands(commands)
        for name, t, numel in (("dof_states", dof_states, self.n * 24), ("body_states", body_states, self.n * 13), ("commands", commands, self.n * 16)):
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != numel:
                raise ValueError(f"{name} must be a contiguous cuda float32 tensor with {numel} elements")
        if request.dtype != torch.int32 or not request.is_cuda or not request.is_contiguous() or request.numel() != self.n:
            raise ValueError("request must be a contiguous cuda int32 tensor with one entry per robot")
        torques = self.torques if torques is None else torques
        stream = torch.cuda.current_stream(self.device).cuda_stream
        fn = _lib.lib().mpc_ctrl_run_fsm_estimated if estimated else _lib.lib().mpc_ctrl_run_fsm
        _lib.check(fn(self._handle, dof_states.data_ptr(), body_states.data_ptr(), commands.data_ptr(), request.data_ptr(), torques.data_ptr(), stream), "mpc_ctrl_run_fsm")
        return torques

    def fsm_reset(self, env_ids=None, control_mode=None):
        """``RobotRunnerFSM.reset`` (= ``ControlFSM.initialize``) for the given robots (all if None)."""
        import torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        cm = None if control_mode is None else np.ascontiguousarray(control_mode, dtype=np.int32)
        if cm is not None and cm.shape != (self.n,):
            raise ValueError(f"fsm_reset: control_mode must have one entry per robot ({self.n}), got shape {cm.shape}")
        ids = None
        if env_ids is not None and cm is None and hasattr(env_ids, "is_cuda") and env_ids.is_cuda:      # an env_ids tensor on the device: no host round trip
            d_ids = env_ids.to(device=self.device, dtype=torch.int32).contiguous()
            _lib.check(_lib.lib().mpc_ctrl_fsm_reset_device(self._handle, d_ids.data_ptr(), d_ids.numel(), stream), "mpc_ctrl_fsm_reset_device")
            return
        if env_ids is not None:
            ids = np.ascontiguousarray(env_ids.detach().cpu().numpy() if hasattr(env_ids, "detach") else env_ids, dtype=np.int32)
        _lib.check(_lib.lib().mpc_ctrl_fsm_reset(self._handle, None if ids is None else ids.ctypes.data, 0 if ids is None else len(ids),
                                                 None if cm is None else cm.ctypes.data, stream), "mpc_ctrl_fsm_reset")

    def fsm_state(self):
        """[N, 4] int32: FSM state name, operating mode, RecoveryStand flag, unsafe flag."""
        out = np.zeros((self.n, 4), dtype=np.int32)
        _lib.check(_lib.lib().mpc_ctrl_fsm_state(self._handle, out.ctypes.data), "mpc_ctrl_fsm_state")
        return out

    def run_policy(self, policy, dof_states, body_states, commands3, prev_weights, request):
        """The batched ``RobotRunnerPolicy.run`` (robot_runner/RobotRunnerPolicy.py:62-92): StateEstimator.update, the weight
        policy on the fresh estimate (its "previous actions" observation slot carries the previous WEIGHTS, as there:
        ``compute_observations(dof, result, commands, self.weights)``), then the control FSM with those weights.
        ``policy`` is a ``weight_policy.WeightPolicy``; returns (torques [N,12], weights [N,12])."""
        import torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        dof_states, body_states, commands3 = self._inputs(dof_states, body_states, commands3)
        if prev_weights is not None and not hasattr(prev_weights, "is_cuda"):
            prev_weights = torch.from_numpy(np.ascontiguousarray(prev_weights, dtype=np.float32)).to(self.device)
        if body_states.dtype != torch.float32 or not body_states.is_cuda or not body_states.is_contiguous() or body_states.numel() != self.n * 13:
            raise ValueError("body_states must be a contiguous cuda float32 tensor with %d elements" % (self.n * 13))
        _lib.check(_lib.lib().mpc_ctrl_update_estimate(self._handle, body_states.data_ptr(), stream), "mpc_ctrl_update_estimate")
        # the observations straight from the controller's estimate and ground_normal_yaw (no copies), and the FSM tick without a second StateEstimator.update
        obs = policy.observations_from(self, dof_states, commands3, prev_weights)
        weights = policy.step(obs)
        return self.run_fsm(dof_states, body_states, policy.pack_commands(commands3, weights), request, estimated=True), weights

    def estimate(self):
        """(est [n,18], ground_normal_yaw [n,3]) of the last ``run``: the StateEstimate the reference passes to
        ``WeightPolicy.compute_observations`` (vBody, omegaBody, rpyBody, ground_R_body_frame; StateEstimator.py:99-143)."""
        import torch
        est = torch.empty((self.n, 18), dtype=torch.float32, device=self.device)
        nrm = torch.empty((self.n, 3), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().mpc_ctrl_estimate(self._handle, est.data_ptr(), nrm.data_ptr(), stream), "mpc_ctrl_estimate")
        return est, nrm

    def solver_record(self):
        """[N, 56+4h] float32: the 13 arguments of each robot's last compute_contact_forces call as the controller marshalled them."""
        out = np.zeros((self.n, 56 + 4 * self.h), dtype=np.float32)
        _lib.check(_lib.lib().mpc_ctrl_solver_record(self._handle, out.ctypes.data), "mpc_ctrl_solver_record")
        return out

    def solver_forces(self):
        """[N, 12h] float64: each robot's force vector of its last solve (what compute_contact_forces returned)."""
        out = np.zeros((self.n, 12 * self.h), dtype=np.float64)
        _lib.check(_lib.lib().mpc_ctrl_solver_forces(self._handle, out.ctypes.data), "mpc_ctrl_solver_forces")
        return out

    def set_iteration(self, iteration):
        """``cMPC.iterationCounter = iteration[r]`` for every robot (ConvexMPCLocomotion.py:62): gait phase and MPC cadence follow from it."""
        import torch
        it = np.ascontiguousarray(iteration, dtype=np.int32)
        if it.shape != (self.n,):
            raise ValueError("iteration must have one entry per robot")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().mpc_ctrl_set_iteration(self._handle, it.ctypes.data, stream), "mpc_ctrl_set_iteration")

    def enable_timing(self):
        """HIP events around the two solver kernels of every launch of this controller's solver (mpc_batch_enable_timing)."""
        _lib.check(_lib.lib().mpc_batch_enable_timing(_lib.lib().mpc_ctrl_solver(self._handle)), "mpc_batch_enable_timing")

    def kernel_times(self, last_k):
        """(prep_ms [k], solve_ms [k]) of the last k solver launches of this controller."""
        a = np.zeros(last_k, dtype=np.float32); b = np.zeros(last_k, dtype=np.float32)
        _lib.check(_lib.lib().mpc_batch_kernel_times(_lib.lib().mpc_ctrl_solver(self._handle), int(last_k), a.ctypes.data, b.ctypes.data), "mpc_batch_kernel_times")
        return a, b

    def solver_info(self):
        out = np.zeros((self.n, 8), dtype=np.int32)
        _lib.check(_lib.lib().mpc_ctrl_solver_info(self._handle, out.ctypes.data), "mpc_ctrl_solver_info")
        return out
