"""Batched contact-force solver: N robots' ``ConvexMpc`` objects behind one handle.

Host-side mirror of the reference's per-robot plugin object (``mpc_osqp.ConvexMpc``,
mpc_osqp.cc:952-983) for a whole batch: construction = N constructor calls
(ConvexMPCLocomotion.py:102-108), ``solve`` = N ``compute_contact_forces`` calls
(ConvexMPCLocomotion.py:171-185, looped at RL_Environment/tasks/aliengo.py:252-256),
``reset(env_ids)`` = re-construction for those robots (aliengo.py:333-334).  torch is used for device
memory and streams only.
"""
import ctypes as C

import numpy as np

from . import _lib
from .layout import in_len

STATUS_SOLVED = 1


class BatchedConvexMpc:
    def __init__(self, mass, inertia9, planning_horizon, timestep, alpha=1e-5, device=None, solver="osqp"):
        """mass: [N] floats; inertia9: [N, 9] row-major 3x3 body inertias (host arrays).
        solver: "osqp" = the reference's OSQP branch (BASELINE's comparator), "exact" = its qpOASES branch (the QP's optimum, cold every call)."""
        import torch
        if not torch.cuda.is_available():
            raise _lib.MpcLibraryError("BatchedConvexMpc needs a GPU (torch.cuda.is_available() is False); no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        mass = np.ascontiguousarray(mass, dtype=np.float64).reshape(-1)
        inertia9 = np.ascontiguousarray(inertia9, dtype=np.float64).reshape(len(mass), 9)
        self.n, self.h = len(mass), int(planning_horizon)
        self.in_len = in_len(self.h)
        self._handle = C.c_void_p()
        L = _lib.lib()
        _lib.check(L.mpc_batch_create(C.byref(self._handle), self.n, self.h, float(timestep), float(alpha),
                                      mass.ctypes.data, inertia9.ctypes.data), "mpc_batch_create")
        if solver not in ("osqp", "exact"):
            raise ValueError("solver must be 'osqp' or 'exact'")
        _lib.check(L.mpc_batch_set_solver(self._handle, 1 if solver == "exact" else 0), "mpc_batch_set_solver")
        self.forces = torch.zeros((self.n, 12 * self.h), dtype=torch.float64, device=self.device)
        self.info = torch.zeros((self.n, 8), dtype=torch.int32, device=self.device)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h and _lib is not None and _lib._LIB is not None:   # module globals may be gone at interpreter exit
            _lib._LIB.mpc_batch_destroy(h)
            self._handle = None

    def solve(self, inputs, forces=None, info=None):
        """inputs: cuda [N, 56+4h] (layout.py) -- float32 (what the reference's Python holds), float64 (what pybind11 widens it to) or
        float16 (BASELINE configs[4]: the state stored in half precision); the arithmetic is fp64 whatever the storage type.
        Returns (forces f64 [N,12h], info i32 [N,8]).
        Rows whose info[:,1] != 1 (not OSQP_SOLVED) keep their previous forces (the reference returns
        an empty list there, mpc_osqp.cc:781-794)."""
        import torch
        entry = {torch.float32: "mpc_batch_solve", torch.float64: "mpc_batch_solve_f64", torch.float16: "mpc_batch_solve_f16"}.get(inputs.dtype)
        if entry is None or not inputs.is_cuda or not inputs.is_contiguous() or tuple(inputs.shape) != (self.n, self.in_len):
            raise ValueError(f"inputs must be a contiguous cuda float32 / float64 / float16 tensor of shape {(self.n, self.in_len)}")
        forces = self.forces if forces is None else forces
        info = self.info if info is None else info
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(getattr(_lib.lib(), entry)(self._handle, inputs.data_ptr(), forces.data_ptr(), info.data_ptr(), stream), entry)
        return forces, info

    def set_max_iter(self, max_iter):
        """OSQP's max_iter setting (the reference keeps the default 4000): a positive multiple of 25."""
        _lib.check(_lib.lib().mpc_batch_set_max_iter(self._handle, int(max_iter)), "mpc_batch_set_max_iter")

    def reset(self, env_ids=None):
        import torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if env_ids is None:
            _lib.check(_lib.lib().mpc_batch_reset(self._handle, None, 0, stream), "mpc_batch_reset")
            return
        if hasattr(env_ids, "is_cuda") and env_ids.is_cuda:      # an env_ids tensor on the device (VecTask.reset_idx): no host round trip
            d_ids = env_ids.to(device=self.device, dtype=torch.int32).contiguous()
            _lib.check(_lib.lib().mpc_batch_reset_device(self._handle, d_ids.data_ptr(), d_ids.numel(), stream), "mpc_batch_reset_device")
            return
        ids = np.ascontiguousarray(env_ids.detach().cpu().numpy() if hasattr(env_ids, "detach") else env_ids, dtype=np.int32)
        _lib.check(_lib.lib().mpc_batch_reset(self._handle, ids.ctypes.data, len(ids), stream), "mpc_batch_reset")

    def device_bytes(self):
        return _lib.lib().mpc_batch_device_bytes(self._handle)

    def get_state(self):
        out = np.zeros((self.n, _lib.lib().mpc_batch_state_len(self._handle)))
        _lib.check(_lib.lib().mpc_batch_get_state(self._handle, out.ctypes.data), "mpc_batch_get_state")
        return out

    def get_qp(self):
        """[N, qp_len] float64: the QP record of the last launch (see include/mpc_batch.h)."""
        out = np.zeros((self.n, _lib.lib().mpc_batch_qp_len(self._handle)))
        _lib.check(_lib.lib().mpc_batch_get_qp(self._handle, out.ctypes.data), "mpc_batch_get_qp")
        return out

    def get_scale(self):
        """[N, scale_len] float64: the scale record (OSQP's Ruiz equilibration) of the last launch."""
        out = np.zeros((self.n, _lib.lib().mpc_batch_scale_len(self._handle)))
        _lib.check(_lib.lib().mpc_batch_get_scale(self._handle, out.ctypes.data), "mpc_batch_get_scale")
        return out

    def enable_timing(self):
        _lib.check(_lib.lib().mpc_batch_enable_timing(self._handle), "mpc_batch_enable_timing")

    def kernel_times(self, last_k):
        """(assembly_ms [k], solve_ms [k]) of the last k launches, from HIP events recorded on the launch stream."""
        a = np.zeros(last_k, dtype=np.float32); b = np.zeros(last_k, dtype=np.float32)
        _lib.check(_lib.lib().mpc_batch_kernel_times(self._handle, int(last_k), a.ctypes.data, b.ctypes.data), "mpc_batch_kernel_times")
        return a, b

    def get_profile(self):
        """Shader-clock cycles of the last solve per robot, 16 sections (csrc/mpc_core.h kProfLen)."""
        out = np.zeros((self.n, 16), dtype=np.int64)
        _lib.check(_lib.lib().mpc_batch_get_profile(self._handle, out.ctypes.data), "mpc_batch_get_profile")
        return out

    def set_state(self, state):
        st = np.ascontiguousarray(state, dtype=np.float64)
        _lib.check(_lib.lib().mpc_batch_set_state(self._handle, st.ctypes.data), "mpc_batch_set_state")
