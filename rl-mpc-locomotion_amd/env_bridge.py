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
        import torch
        self.ctl = BatchedLocomotion(robot_type, gait_id, horizon=horizon, controller_dt=controller_dt, flat_ground=flat_ground, device=device)
        self.device, self.n = self.ctl.device, self.ctl.n
        self._scale = torch.tensor(param_scale, dtype=torch.float, device=self.device)       # Parameters.MPC_param_scale
        self._const = torch.tensor(param_const, dtype=torch.float, device=self.device)       # Parameters.MPC_param_const
        self._cmd = torch.zeros((self.n, 16), dtype=torch.float32, device=self.device)

    def pre_physics_step(self, actions, dof_state, root_states, commands):
        """actions [N,12] in [-1,1], dof_state [N*12,2] (or [N,12,2]), root_states [N,13], commands [N,3] -> torques [N,12]
        (aliengo.py:237-258).  The weights are ``actions * scale + const`` exactly as there (torch.mul(...).add(...))."""
        import torch
        actions_rescale = torch.mul(actions.to(self.device, torch.float), self._scale).add(self._const)
        self._cmd[:, 0:3] = commands
        self._cmd[:, 3:15] = actions_rescale
        self._cmd[:, 15] = 0.0                                          # np.concatenate((commands, actions, [0.0]))
        return self.ctl.run(dof_state.reshape(self.n, 12, 2).contiguous(), root_states.contiguous(), self._cmd)

    def reset_idx(self, env_ids):
        """``for idx in env_ids: self.controllers[idx].reset()`` (aliengo.py:330-334)."""
        if len(env_ids):
            self.ctl.reset(env_ids)
