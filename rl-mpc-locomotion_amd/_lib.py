"""ctypes loader of the C-ABI library (include/mpc_batch.h).  There is NO fallback: if the HIP
library is missing or no GPU is usable, every entry point raises -- the product path never routes
through a CPU implementation."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPC_LIB_PATH", os.path.join(_HERE, "csrc", "libmpc_batch.so"))   # override: kernel-variant experiments
_LIB = None

MPC_OK = 0
SYMBOLS = [
    "mpc_input_len", "mpc_supported_horizons", "mpc_batch_create", "mpc_batch_destroy", "mpc_batch_solve",
    "mpc_batch_set_solver", "mpc_batch_set_max_iter", "mpc_batch_solve_f64", "mpc_batch_solve_f16", "mpc_batch_reset", "mpc_batch_reset_device", "mpc_batch_solve_host", "mpc_batch_solve_host_f64", "mpc_batch_size", "mpc_batch_horizon", "mpc_batch_device_bytes",
    "mpc_batch_state_len", "mpc_batch_get_state", "mpc_batch_set_state", "mpc_batch_qp_len", "mpc_batch_scale_len", "mpc_batch_get_qp", "mpc_batch_get_scale", "mpc_batch_get_profile", "mpc_batch_enable_timing",
    "mpc_batch_kernel_times", "mpc_last_error",
    "mpc_ctrl_create", "mpc_ctrl_destroy", "mpc_ctrl_step", "mpc_ctrl_run", "mpc_ctrl_reset", "mpc_ctrl_reset_device", "mpc_ctrl_set_gait", "mpc_ctrl_set_gait_device", "mpc_ctrl_set_solver", "mpc_ctrl_solver_info", "mpc_ctrl_solver_record", "mpc_ctrl_solver_forces", "mpc_ctrl_solver", "mpc_ctrl_set_iteration", "mpc_device_clock",
    "mpc_ctrl_fsm_init", "mpc_ctrl_run_fsm", "mpc_ctrl_fsm_reset", "mpc_ctrl_fsm_reset_device", "mpc_ctrl_fsm_state",
    "mpc_policy_create", "mpc_policy_destroy", "mpc_policy_step", "mpc_policy_observations", "mpc_ctrl_estimate", "mpc_ctrl_update_estimate", "mpc_pack_commands", "mpc_pack_commands_scaled", "mpc_ctrl_policy_observations", "mpc_ctrl_run_fsm_estimated",
    "mpc_peer_create", "mpc_peer_handle", "mpc_peer_connect", "mpc_peer_put", "mpc_peer_wait", "mpc_peer_timeouts", "mpc_peer_destroy", "mpc_peer_last_error",
]


class MpcLibraryError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MpcLibraryError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # The Python host hands torch device tensors to the library, so both must sit on ONE HIP runtime: torch ships its own
        # libamdhip64 and has to be loaded first -- a library that pulled in /opt/rocm's copy before `import torch` ends up
        # with a second runtime that sees no device ("mpc_batch_create: no HIP device").
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        L.mpc_input_len.argtypes = [ci]; L.mpc_input_len.restype = ci
        L.mpc_supported_horizons.argtypes = [vp, ci]; L.mpc_supported_horizons.restype = ci
        L.mpc_batch_create.argtypes = [C.POINTER(vp), ci, ci, cd, cd, vp, vp]; L.mpc_batch_create.restype = ci
        L.mpc_batch_destroy.argtypes = [vp]; L.mpc_batch_destroy.restype = None
        L.mpc_batch_solve.argtypes = [vp, vp, vp, vp, vp]; L.mpc_batch_solve.restype = ci
        L.mpc_batch_reset.argtypes = [vp, vp, ci, vp]; L.mpc_batch_reset.restype = ci
        L.mpc_batch_reset_device.argtypes = [vp, vp, ci, vp]; L.mpc_batch_reset_device.restype = ci
        L.mpc_batch_solve_host.argtypes = [vp, vp, vp, vp]; L.mpc_batch_solve_host.restype = ci
        L.mpc_batch_solve_host_f64.argtypes = [vp, vp, vp, vp]; L.mpc_batch_solve_host_f64.restype = ci
        L.mpc_batch_solve_f64.argtypes = [vp, vp, vp, vp, vp]; L.mpc_batch_solve_f64.restype = ci
        L.mpc_batch_solve_f16.argtypes = [vp, vp, vp, vp, vp]; L.mpc_batch_solve_f16.restype = ci
        L.mpc_batch_set_solver.argtypes = [vp, ci]; L.mpc_batch_set_solver.restype = ci
        L.mpc_batch_set_max_iter.argtypes = [vp, ci]; L.mpc_batch_set_max_iter.restype = ci
        L.mpc_batch_size.argtypes = [vp]; L.mpc_batch_size.restype = ci
        L.mpc_batch_horizon.argtypes = [vp]; L.mpc_batch_horizon.restype = ci
        L.mpc_batch_device_bytes.argtypes = [vp]; L.mpc_batch_device_bytes.restype = C.c_longlong
        L.mpc_batch_state_len.argtypes = [vp]; L.mpc_batch_state_len.restype = ci
        L.mpc_batch_get_state.argtypes = [vp, vp]; L.mpc_batch_get_state.restype = ci
        L.mpc_batch_set_state.argtypes = [vp, vp]; L.mpc_batch_set_state.restype = ci
        L.mpc_batch_get_profile.argtypes = [vp, vp]; L.mpc_batch_get_profile.restype = ci
        L.mpc_batch_qp_len.argtypes = [vp]; L.mpc_batch_qp_len.restype = ci
        L.mpc_batch_scale_len.argtypes = [vp]; L.mpc_batch_scale_len.restype = ci
        L.mpc_batch_get_qp.argtypes = [vp, vp]; L.mpc_batch_get_qp.restype = ci
        L.mpc_batch_get_scale.argtypes = [vp, vp]; L.mpc_batch_get_scale.restype = ci
        L.mpc_batch_enable_timing.argtypes = [vp]; L.mpc_batch_enable_timing.restype = ci
        L.mpc_batch_kernel_times.argtypes = [vp, ci, vp, vp]; L.mpc_batch_kernel_times.restype = ci
        L.mpc_ctrl_create.argtypes = [C.POINTER(vp), ci, ci, cd, ci, cd, ci, vp, vp, ci, vp, vp, vp]; L.mpc_ctrl_create.restype = ci
        L.mpc_ctrl_destroy.argtypes = [vp]; L.mpc_ctrl_destroy.restype = None
        L.mpc_ctrl_step.argtypes = [vp, vp, vp, vp, vp, vp]; L.mpc_ctrl_step.restype = ci
        L.mpc_ctrl_run.argtypes = [vp, vp, vp, vp, vp, vp]; L.mpc_ctrl_run.restype = ci
        L.mpc_ctrl_reset.argtypes = [vp, vp, ci, vp]; L.mpc_ctrl_reset.restype = ci
        L.mpc_ctrl_reset_device.argtypes = [vp, vp, ci, vp]; L.mpc_ctrl_reset_device.restype = ci
        L.mpc_ctrl_set_gait.argtypes = [vp, vp, vp]; L.mpc_ctrl_set_gait.restype = ci
        L.mpc_ctrl_set_gait_device.argtypes = [vp, vp, vp]; L.mpc_ctrl_set_gait_device.restype = ci
        L.mpc_ctrl_set_solver.argtypes = [vp, ci]; L.mpc_ctrl_set_solver.restype = ci
        L.mpc_ctrl_solver_info.argtypes = [vp, vp]; L.mpc_ctrl_solver_info.restype = ci
        L.mpc_ctrl_solver_record.argtypes = [vp, vp]; L.mpc_ctrl_solver_record.restype = ci
        L.mpc_ctrl_solver_forces.argtypes = [vp, vp]; L.mpc_ctrl_solver_forces.restype = ci
        L.mpc_ctrl_solver.argtypes = [vp]; L.mpc_ctrl_solver.restype = vp
        L.mpc_ctrl_set_iteration.argtypes = [vp, vp, vp]; L.mpc_ctrl_set_iteration.restype = ci
        L.mpc_device_clock.argtypes = [ci, ci, vp, vp]; L.mpc_device_clock.restype = ci
        L.mpc_ctrl_fsm_init.argtypes = [vp, vp, ci, ci, vp]; L.mpc_ctrl_fsm_init.restype = ci
        L.mpc_ctrl_run_fsm.argtypes = [vp, vp, vp, vp, vp, vp, vp]; L.mpc_ctrl_run_fsm.restype = ci
        L.mpc_ctrl_fsm_reset.argtypes = [vp, vp, ci, vp, vp]; L.mpc_ctrl_fsm_reset.restype = ci
        L.mpc_ctrl_fsm_reset_device.argtypes = [vp, vp, ci, vp]; L.mpc_ctrl_fsm_reset_device.restype = ci
        L.mpc_ctrl_fsm_state.argtypes = [vp, vp]; L.mpc_ctrl_fsm_state.restype = ci
        L.mpc_policy_create.argtypes = [C.POINTER(vp), ci, vp, vp, vp, vp, vp]; L.mpc_policy_create.restype = ci
        L.mpc_policy_destroy.argtypes = [vp]; L.mpc_policy_destroy.restype = None
        L.mpc_policy_step.argtypes = [vp, ci, vp, vp, vp, vp]; L.mpc_policy_step.restype = ci
        L.mpc_policy_observations.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp]; L.mpc_policy_observations.restype = ci
        L.mpc_ctrl_update_estimate.argtypes = [vp, vp, vp]; L.mpc_ctrl_update_estimate.restype = ci
        L.mpc_ctrl_estimate.argtypes = [vp, vp, vp, vp]; L.mpc_ctrl_estimate.restype = ci
        L.mpc_pack_commands.argtypes = [ci, vp, vp, vp, vp]; L.mpc_pack_commands.restype = ci
        L.mpc_pack_commands_scaled.argtypes = [ci, vp, vp, vp, vp, vp, vp]; L.mpc_pack_commands_scaled.restype = ci
        L.mpc_ctrl_policy_observations.argtypes = [vp, vp, vp, vp, vp, vp, vp]; L.mpc_ctrl_policy_observations.restype = ci
        L.mpc_ctrl_run_fsm_estimated.argtypes = [vp, vp, vp, vp, vp, vp, vp]; L.mpc_ctrl_run_fsm_estimated.restype = ci
        L.mpc_peer_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci]; L.mpc_peer_create.restype = ci
        L.mpc_peer_handle.argtypes = [vp, vp]; L.mpc_peer_handle.restype = ci
        L.mpc_peer_connect.argtypes = [vp, vp]; L.mpc_peer_connect.restype = ci
        L.mpc_peer_put.argtypes = [vp, vp, ci, ci, vp]; L.mpc_peer_put.restype = ci
        L.mpc_peer_wait.argtypes = [vp, vp, vp]; L.mpc_peer_wait.restype = ci
        L.mpc_peer_timeouts.argtypes = [vp, vp]; L.mpc_peer_timeouts.restype = ci
        L.mpc_peer_destroy.argtypes = [vp]; L.mpc_peer_destroy.restype = None
        L.mpc_peer_last_error.argtypes = []; L.mpc_peer_last_error.restype = C.c_char_p
        L.mpc_last_error.argtypes = []; L.mpc_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def kernel_source_hash():
    """sha256 over the kernel sources (csrc/*.h, *.hip, the Makefile): what a measurement of the library is a measurement OF.  Profiles record
    it (tools/pmc_passes.sh) and bench.py quotes a profile's numbers only when it equals the tree's."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def device_clock(device=0, busy_ms=20):
    """(GHz, ms): the shader clock `device` sustains under one wave of dependent fp64 FMAs per SIMD (mpc_device_clock)."""
    ghz, ms = C.c_double(0.0), C.c_double(0.0)
    check(lib().mpc_device_clock(int(device), int(busy_ms), C.addressof(ghz), C.addressof(ms)), "mpc_device_clock")
    return ghz.value, ms.value


def check(rc, what):
    if rc != MPC_OK:
        raise MpcLibraryError(f"{what} failed ({rc}): {lib().mpc_last_error().decode()}")
