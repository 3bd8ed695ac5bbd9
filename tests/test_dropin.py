"""The drop-in claim of INTEGRATION.md, executed.

CPU (build container, reference tree present): the UNMODIFIED reference Python -- MPC_Controller.robot_runner.RobotRunnerMin with
its ConvexMPCLocomotion, Gait, LegController, StateEstimator -- runs with `sys.modules["mpc_osqp"]` replaced, exactly as
INTEGRATION.md says, by a module of the reference's interface that is served by the host emulation of the device kernels, and
reproduces the golden torques of BASELINE configs[0] (minted with the vendored OSQP behind the same seam).

GPU (no reference tree there): every compute_contact_forces call the reference made in that run -- recorded with its 13
arguments by tests/golden/make_golden_shim_calls.py -- is replayed through the product's module of that interface,
rl_mpc_locomotion_amd.mpc_osqp.ConvexMpc (HIP library), and must return the lists the reference got."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from tests.helpers import GOLDEN, HAVE_REFERENCE, ROOT

CHILD = r'''
import sys, types, numpy as np
sys.path.insert(0, "{root}"); sys.path.insert(0, "/root/reference")
import rl_mpc_locomotion_amd
import tests.emu.emu_mpc_osqp as shim
sys.modules["mpc_osqp"] = shim                     # <- the one line of INTEGRATION.md
from MPC_Controller.Parameters import Parameters
from MPC_Controller.utils import GaitType
Parameters.bridge_MPC_to_RL = True
from MPC_Controller.robot_runner.RobotRunnerMin import RobotRunnerMin
from MPC_Controller.common.Quadruped import RobotType
g = np.load("{gold}")
Parameters.flat_ground = bool(g["flat_ground"]); Parameters.cmpc_gait = GaitType.TROT
runner = RobotRunnerMin(); runner.init(RobotType.ALIENGO)
T = {ticks}
err = 0.0
for k in range(T):
    tau = runner.run(g["dof"][k, 0], g["body"][k, 0], g["cmd"][k, 0])
    ref = g["torque"][k, 0]
    err = max(err, float(np.abs(tau - ref).max() / max(np.abs(ref).max(), 1.0)))
print("DROPIN_MAX_RELERR", err)
'''


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference tree (/root/reference)")
def test_unmodified_reference_runs_on_the_kernel_algorithm():
    ticks = 400
    code = CHILD.format(root=ROOT, gold=os.path.join(GOLDEN, "controller_h10_config1.npz"), ticks=ticks)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    err = float([l for l in out.stdout.splitlines() if l.startswith("DROPIN_MAX_RELERR")][0].split()[1])
    assert err < 5e-5, err


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["OSQP", "QPOASES"])
def test_recorded_reference_calls_through_the_hip_module(solver):
    """solver = OSQP: the lists the reference's OSQP branch returned (warm-started call sequence); QPOASES -- what the shipped Python
    passes (ConvexMPCLocomotion.py:108) -- the QP's exact optimum per call (oracle: vendored OSQP, cold, eps 1e-9, polish)."""
    import rl_mpc_locomotion_amd  # noqa: F401
    import rl_mpc_locomotion_amd.mpc_osqp as mpc
    from rl_mpc_locomotion_amd import layout as L
    g = np.load(os.path.join(GOLDEN, "shim_calls_config1.npz"))
    h = int(g["horizon"][0])
    cpp_mpc = mpc.ConvexMpc(float(g["mass"][0]), g["inertia"].tolist(), int(g["num_legs"][0]), h, float(g["timestep"][0]), float(g["alpha"][0]),
                            getattr(mpc, solver))                                                    # ConvexMPCLocomotion.py:102-108
    wants = g["out"] if solver == "OSQP" else g["out_exact"]
    worst, t_call = 0.0, []
    for rec, want, ok in zip(g["rec"], wants, g["ok"]):
        args = L.unpack_args(h, rec)                                                               # the 13 positional arguments
        t0 = time.perf_counter()
        got = cpp_mpc.compute_contact_forces(*args)                                                  # ConvexMPCLocomotion.py:171-185
        t_call.append(time.perf_counter() - t0)
        assert isinstance(got, list) and (len(got) == 12 * h) == (bool(ok) or solver == "QPOASES")
        if len(got):
            worst = max(worst, float(np.abs(np.array(got) - want).max() / max(np.abs(want).max(), 1.0)))
    assert worst < (1e-5 if solver == "OSQP" else 1e-6), worst
    print(f"compute_contact_forces({solver}) through the HIP module: median {np.median(t_call) * 1e3:.3f} ms per call (one robot, host buffers)")


@pytest.mark.gpu
def test_exact_solver_batch_matches_the_unique_optimum():
    """MPC_SOLVER_EXACT on a batch (mixed robots / gaits, and the edge cases): within 1e-6 of the oracle's exact optimum, all
    horizon steps, swing feet at zero."""
    import torch
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    from oracle.refmpc import RefConvexMpc
    from tests.helpers import load_golden, inertia9_from_diag
    for name in ("solver_h10_cfg3", "solver_h10_edge", "solver_h16_cfg4"):
        g = load_golden(name)
        h, n = int(g["h"]), min(len(g["mass"]), 12)
        gpu = BatchedConvexMpc(g["mass"][:n], inertia9_from_diag(g["inertia_diag"][:n]), h, float(g["dt_mpc"]), float(g["alpha"]), device="cuda:0", solver="exact")
        for s in range(2):                                  # the second call must not be warm-started
            f, info = gpu.solve(torch.from_numpy(g[f"inputs_{s}"][:n]).cuda())
            torch.cuda.synchronize()
            f = f.cpu().numpy(); info = info.cpu().numpy()
            assert (info[:, 1] == 1).all() and (info[:, 5] == 1).all()
            for r in range(n):
                d = g["inertia_diag"][r]
                ref = RefConvexMpc(g["mass"][r], [d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]], 4, h, float(g["dt_mpc"]), float(g["alpha"]))
                fx = ref.solve_exact(g[f"inputs_{s}"][r])
                assert np.abs(f[r] - fx).max() < 1e-6 * max(np.abs(fx).max(), 1.0), (name, s, r, np.abs(f[r] - fx).max())
                swing = np.repeat(g[f"inputs_{s}"][r, 28:28 + 4 * h] == 0, 3)
                assert np.abs(f[r][swing]).max(initial=0.0) < 1e-6


def test_exact_solver_on_the_host_emulation_matches_the_unique_optimum():
    """The exact-optimum mode of the kernel code (staged ADMM + verified active-set polish, mpc_wrench.h run<true>) on the host
    emulation against the oracle's exact optimum; the second call of a pair must start cold."""
    from oracle.refmpc import RefConvexMpc
    from tests.emu.emu import EmuBatch
    from tests.helpers import load_golden
    for name, n in (("solver_h10_cfg3", 8), ("solver_h10_edge", 10), ("solver_h16_cfg4", 3)):
        g = load_golden(name)
        h = int(g["h"])
        emu = EmuBatch(g["mass"][:n], g["inertia_diag"][:n], h, float(g["dt_mpc"]), float(g["alpha"]))
        for s in range(2):
            f = emu.solve(g[f"inputs_{s}"][:n], exact=True)
            assert (emu.info[:, 1] == 1).all() and (emu.info[:, 5] == 1).all()
            for r in range(n):
                d = g["inertia_diag"][r]
                ref = RefConvexMpc(g["mass"][r], [d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]], 4, h, float(g["dt_mpc"]), float(g["alpha"]))
                fx = ref.solve_exact(g[f"inputs_{s}"][r])
                assert np.abs(f[r] - fx).max() < 1e-6 * max(np.abs(fx).max(), 1.0), (name, s, r)
