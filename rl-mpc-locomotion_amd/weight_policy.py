"""Batched weight policy: observations -> actor MLP -> MPC weights, on the GPU.

Host-side mirror of the reference's ``WeightPolicy`` (RL_Environment/WeightPolicy.py:33-139) for N robots at once:
``compute_observations`` + ``step`` keep their meaning, tensors replace the per-robot numpy arrays.  The network is the
actor of rsl_rl's ActorCritic (Linear/ELU stack 48-512-256-128-12, RL_Environment/tasks/legged_config_ppo.py:5-9);
``from_state_dict`` takes the ``model_state_dict`` of a checkpoint exactly as ``WeightPolicy.__init__`` loads it (:75-77).
"""
import ctypes as C

import numpy as np

from . import _lib

MPC_PARAM_SCALE = (4, 4, 4, 20, 20, 20, 1, 1, 1, 1, 1, 1)     # MPC_Controller/Parameters.py:25-28
MPC_PARAM_CONST = (5, 5, 5, 50, 50, 50, 1, 1, 1, 1, 1, 1)     # MPC_Controller/Parameters.py:30-33


class WeightPolicy:
    def __init__(self, layers, scale=MPC_PARAM_SCALE, const=MPC_PARAM_CONST, obs_scales=(1.0, 1.0, 1.0, 1.0), device=None):
        """layers: [(weight [out, in], bias [out]), ...] float32 (torch Linear layout).
        obs_scales = (linearVelocityScale, angularVelocityScale, dofPositionScale, dofVelocityScale), cfg/task/*.yaml."""
        import torch
        if not torch.cuda.is_available():
            raise _lib.MpcLibraryError("WeightPolicy needs a GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w, _ in layers]
        bs = [np.ascontiguousarray(b, dtype=np.float32) for _, b in layers]
        dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        for l, (w, b) in enumerate(zip(ws, bs)):
            if w.shape != (dims[l + 1], dims[l]) or b.shape != (dims[l + 1],):
                raise ValueError(f"layer {l}: expected weight {(dims[l + 1], dims[l])} and bias {(dims[l + 1],)}")
        self.dims = dims
        self.num_obs, self.num_actions = dims[0], dims[-1]
        sc = np.ascontiguousarray(scale, dtype=np.float32); ct = np.ascontiguousarray(const, dtype=np.float32)
        if len(sc) != self.num_actions or len(ct) != self.num_actions:
            raise ValueError("scale / const need one entry per action")
        self._scales = np.ascontiguousarray(obs_scales, dtype=np.float32)
        L = len(ws)
        d = (C.c_int * (L + 1))(*dims)
        wp = (C.c_void_p * L)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * L)(*[b.ctypes.data for b in bs])
        self._handle = C.c_void_p()
        _lib.check(_lib.lib().mpc_policy_create(C.byref(self._handle), L, C.cast(d, C.c_void_p), C.cast(wp, C.c_void_p),
                                                C.cast(bp, C.c_void_p), sc.ctypes.data, ct.ctypes.data), "mpc_policy_create")

    @classmethod
    def from_state_dict(cls, state_dict, **kw):
        idx = sorted({int(k.split(".")[1]) for k in state_dict if k.startswith("actor.") and k.endswith(".weight")})
        layers = []
        for i in idx:
            w, b = state_dict[f"actor.{i}.weight"], state_dict[f"actor.{i}.bias"]
            layers.append((w.detach().cpu().numpy() if hasattr(w, "detach") else w, b.detach().cpu().numpy() if hasattr(b, "detach") else b))
        return cls(layers, **kw)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h and _lib is not None and _lib._LIB is not None:
            _lib._LIB.mpc_policy_destroy(h)
            self._handle = None

    def _stream(self):
        import torch
        return torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _chk(name, t, numel):
        import torch
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != numel:
            raise ValueError(f"{name} must be a contiguous cuda float32 tensor with {numel} elements")

    def step(self, obs, return_actions=False):
        """obs [n, 48] -> MPC weights [n, 12] (and the raw actor output if asked)."""
        import torch
        n = obs.shape[0]
        self._chk("obs", obs, n * self.num_obs)
        weights = torch.empty((n, self.num_actions), dtype=torch.float32, device=self.device)
        actions = torch.empty_like(weights) if return_actions else None
        _lib.check(_lib.lib().mpc_policy_step(self._handle, n, obs.data_ptr(), actions.data_ptr() if return_actions else None,
                                              weights.data_ptr(), self._stream()), "mpc_policy_step")
        return (weights, actions) if return_actions else weights

    def compute_observations(self, dof_states, est, ground_normal_yaw, commands, actions):
        """dof_states [n,12,2], est [n,18] (vBody, omegaBody, rpy, R), ground_normal_yaw [n,3], commands [n,3],
        actions [n,12] (previous policy output) -> obs [n,48]."""
        import torch
        n = commands.shape[0]
        for name, t, k in (("dof_states", dof_states, 24), ("est", est, 18), ("ground_normal_yaw", ground_normal_yaw, 3),
                           ("commands", commands, 3), ("actions", actions, 12)):
            self._chk(name, t, n * k)
        obs = torch.empty((n, 48), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().mpc_policy_observations(n, dof_states.data_ptr(), est.data_ptr(), ground_normal_yaw.data_ptr(), commands.data_ptr(),
                                                      actions.data_ptr(), self._scales.ctypes.data, obs.data_ptr(), self._stream()),
                   "mpc_policy_observations")
        return obs

    def observations_from(self, ctl, dof_states, commands, actions):
        """compute_observations with the StateEstimate taken where the controller keeps it (``ctl``: a BatchedLocomotion whose
        ``update_estimate`` / ``run`` has just been called): RobotRunnerPolicy.run's ``compute_observations(dof_states, result, commands,
        self.weights)`` (RobotRunnerPolicy.py:72-78) without copying the estimate out first."""
        import torch
        n = ctl.n
        for name, t, k in (("dof_states", dof_states, 24), ("commands", commands, 3), ("actions", actions, 12)):
            self._chk(name, t, n * k)
        obs = torch.empty((n, 48), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().mpc_ctrl_policy_observations(ctl._handle, dof_states.data_ptr(), commands.data_ptr(), actions.data_ptr(), self._scales.ctypes.data,
                                                           obs.data_ptr(), self._stream()), "mpc_ctrl_policy_observations")
        return obs

    def pack_commands(self, commands, weights):
        """[n,3] velocity commands + [n,12] MPC weights -> the [n,16] command record of BatchedLocomotion.run/step."""
        import torch
        n = commands.shape[0]
        self._chk("commands", commands, n * 3); self._chk("weights", weights, n * 12)
        out = torch.empty((n, 16), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().mpc_pack_commands(n, commands.data_ptr(), weights.data_ptr(), out.data_ptr(), self._stream()), "mpc_pack_commands")
        return out
