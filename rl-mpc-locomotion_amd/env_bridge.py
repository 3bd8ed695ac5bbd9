"""The MPC part of the RL task's step, on device tensors.

Mirror of what ``VecTask.pre_physics_step`` / ``reset_idx`` do when ``Parameters.bridge_MPC_to_RL`` is set
(RL_Environment/tasks/aliengo.py:227-263 and :321-334, same in a1.py / go1.py): rescale the policy's actions to MPC weights,
run ``controller.run`` for every environment, and reset the controllers of the environments that are being reset.  The
reference copies four tensors to the host and loops over Python controllers; here nothing leaves the GPU.  The simulator
calls themselves (``gym.set_dof_actuation_force_tensor`` ...) stay with the caller: Isaac Gym is not part of this package.
"""
from .locomotion import BatchedLocomotion
from .weight_policy import MPC_PARAM_CONST, MPC_PARAM_SCALE


class MpcEnvBridge:
    def __init__(self, robot_type, gait_id, horizon=10, controller_dt=0.01, flat_ground=False, device=None,
                 param_scale=MPC_PARAM_SCALE, param_const=MPC_PARAM_CONST):
        import numpy as np
        import torch
        self.ctl = BatchedLocomotion(robot_type, gait_id, horizon=horizon, controller_dt=controller_dt, flat_ground=flat_ground, device=device)
        self.device, self.n = self.ctl.device, self.ctl.n
        self._scale = np.ascontiguousarray(param_scale, dtype=np.float32)       # Parameters.MPC_param_scale
        self._const = np.ascontiguousarray(param_const, dtype=np.float32)       # Parameters.MPC_param_const
        if self._scale.shape != (12,) or self._const.shape != (12,):
            raise ValueError("param_scale / param_const: twelve entries each (Parameters.py:25-33)")
        self._cmd = torch.zeros((self.n, 16), dtype=torch.float32, device=self.device)

    def pre_physics_step(self, actions, dof_state, root_states, commands):
        """actions [N,12] in [-1,1], dof_state [N*12,2] (or [N,12,2]), root_states [N,13], commands [N,3] -> torques [N,12]
        (aliengo.py:237-258).  The weights are ``actions * scale + const`` exactly as there (torch.mul(...).add(...): a float32 product, then a
        float32 sum), formed and packed with the commands into the controllers' command record by ONE kernel (mpc_pack_commands_scaled; five torch
        launches until round 5 -- tests/test_controller.py::test_env_bridge_equals_manual_composition holds the two bit-identical)."""
        import torch
        from . import _lib
        actions = actions.to(self.device, torch.float32).contiguous()
        commands = commands.to(self.device, torch.float32).contiguous()
        if actions.numel() != self.n * 12 or commands.numel() != self.n * 3:
            raise ValueError("actions [N, 12] and commands [N, 3] expected")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().mpc_pack_commands_scaled(self.n, commands.data_ptr(), actions.data_ptr(), self._scale.ctypes.data, self._const.ctypes.data,
                                                       self._cmd.data_ptr(), stream), "mpc_pack_commands_scaled")
        return self.ctl.run(dof_state.reshape(self.n, 12, 2).contiguous(), root_states.contiguous(), self._cmd)

    def reset_idx(self, env_ids):
        """``for idx in env_ids: self.controllers[idx].reset()`` (aliengo.py:330-334)."""
        if len(env_ids):
            self.ctl.reset(env_ids)
