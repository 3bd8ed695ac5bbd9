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
import sys, types, os, numpy as np
sys.path.insert(0, "{root}"); sys.path.insert(0, "/root/reference")
os.environ["EMU_SHIM_SOLVER"] = "{force}"
import rl_mpc_locomotion_amd
import tests.emu.emu_mpc_osqp as shim
sys.modules["mpc_osqp"] = shim                     # <- the one line of INTEGRATION.md
from MPC_Controller.Parameters import Parameters
from MPC_Controller.utils import GaitType
Parameters.bridge_MPC_to_RL = True
from MPC_Controller.robot_runner.RobotRunnerMin import RobotRunnerMin
from MPC_Controller.common.Quadruped import RobotType
g = np.load("{gold}")
gt = g["torque"][:, 0] if "{force}" == "osqp" else np.load("{gold_exact}")["torque_exact"]
Parameters.flat_ground = bool(g["flat_ground"]); Parameters.cmpc_gait = GaitType.TROT
runner = RobotRunnerMin(); runner.init(RobotType.ALIENGO)
T = {ticks}
err = 0.0
for k in range(T):
    tau = runner.run(g["dof"][k, 0], g["body"][k, 0], g["cmd"][k, 0])
    ref = gt[k]
    err = max(err, float(np.abs(tau - ref).max() / max(np.abs(ref).max(), 1.0)))
print("DROPIN_MAX_RELERR", err)
'''


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference tree (/root/reference)")
@pytest.mark.parametrize("branch", ["osqp", "as shipped"])
def test_unmodified_reference_runs_on_the_kernel_algorithm(branch):
    """branch = osqp: the seam is forced onto the OSQP branch (BASELINE's comparator; golden: the vendored OSQP behind the seam).
    as shipped: the shim honours the solver the unmodified reference passes, mpc.QPOASES (ConvexMPCLocomotion.py:108) -> the exact-optimum
    mode (golden: the oracle's exact optimum behind the seam, torque_exact of shim_calls_config1.npz)."""
    ticks = 400
    code = CHILD.format(root=ROOT, gold=os.path.join(GOLDEN, "controller_h10_config1.npz"), gold_exact=os.path.join(GOLDEN, "shim_calls_config1.npz"),
                        ticks=ticks, force="osqp" if branch == "osqp" else "")
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


def test_pybind11_module_has_the_reference_interface():
    """The pybind11 extension `mpc_osqp` (rl-mpc-locomotion_amd/pybind, built by __graft_entry__.build()) exposes what mpc_osqp.cc:952-983
    does: QPSolverName with exported values, ConvexMpc with the 7-argument constructor, compute_contact_forces, reset_solver,
    __version__, TEST; and it fails loudly, not silently, where there is no GPU."""
    pyb = os.path.join(ROOT, "rl-mpc-locomotion_amd", "pybind")
    code = (f"import sys; sys.path.insert(0, {pyb!r}); import mpc_osqp as mpc\n"
            "assert int(mpc.OSQP) == 0 and int(mpc.QPOASES) == 1 and mpc.QPSolverName.QPOASES == mpc.QPOASES and mpc.TEST == 42 and mpc.__version__ == 'dev'\n"
            "assert all(hasattr(mpc.ConvexMpc, a) for a in ('compute_contact_forces', 'reset_solver'))\n"
            "import torch\n"
            "try:\n"
            "    c = mpc.ConvexMpc(18.0, [0.03, 0, 0, 0, 0.16, 0, 0, 0, 0.17], 4, 10, 0.02, 1e-5, mpc.QPOASES)\n"
            "    print('HAVE_GPU' if torch.cuda.is_available() else 'SILENT_FALLBACK')\n"
            "except RuntimeError as e:\n"
            "    print('LOUD', e)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "SILENT_FALLBACK" not in out.stdout and ("LOUD" in out.stdout or "HAVE_GPU" in out.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["OSQP", "QPOASES"])
def test_recorded_reference_calls_through_the_pybind11_module(solver):
    """The same replay as test_recorded_reference_calls_through_the_hip_module through the pybind11 extension module (the binding
    north_star words: a thin pybind11 layer over the C ABI), in a fresh interpreter that imports it as `mpc_osqp`."""
    pyb = os.path.join(ROOT, "rl-mpc-locomotion_amd", "pybind")
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {pyb!r})
import mpc_osqp as mpc
import rl_mpc_locomotion_amd
from rl_mpc_locomotion_amd import layout as L
g = np.load({os.path.join(GOLDEN, "shim_calls_config1.npz")!r})
h = int(g["horizon"][0])
cpp_mpc = mpc.ConvexMpc(float(g["mass"][0]), g["inertia"].tolist(), int(g["num_legs"][0]), h, float(g["timestep"][0]), float(g["alpha"][0]), mpc.{solver})
wants = g["out"] if "{solver}" == "OSQP" else g["out_exact"]
worst = 0.0
for rec, want, ok in zip(g["rec"], wants, g["ok"]):
    got = cpp_mpc.compute_contact_forces(*[list(map(float, a)) for a in L.unpack_args(h, rec)])
    assert isinstance(got, list) and (len(got) == 12 * h) == (bool(ok) or "{solver}" == "QPOASES")
    if len(got):
        worst = max(worst, float(np.abs(np.array(got) - want).max() / max(np.abs(want).max(), 1.0)))
print("PYBIND_WORST", worst)
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    worst = float([l for l in out.stdout.splitlines() if l.startswith("PYBIND_WORST")][0].split()[1])
    assert worst < (1e-5 if solver == "OSQP" else 1e-6), worst


def _check_exact(solve, name, n, steps=2):
    """The exact mode's result against (i) the KKT conditions of the oracle-assembled QP -- a certificate that needs no second solver --
    and (ii) the oracle's 'exact' solve (vendored OSQP, cold, eps 1e-9, polish), which at the long horizons is the LESS accurate side
    (this QP is only alpha = 1e-5 convex along internal forces: a 1e-9 residual is a ~1e-6 error)."""
    from oracle.refmpc import RefConvexMpc
    from tests.helpers import kkt_certificate, load_golden
    g = load_golden(name)
    h = int(g["h"])
    n = min(n, len(g["mass"]))
    for s in range(steps):                                  # the second call must not be warm-started
        f, info = solve(g, n, s)
        assert (info[:, 1] == 1).all() and (info[:, 5] == 1).all(), (name, s, info[:, :6])
        for r in range(n):
            d = g["inertia_diag"][r]
            ref = RefConvexMpc(g["mass"][r], [d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]], 4, h, float(g["dt_mpc"]), float(g["alpha"]))
            fx = ref.solve_exact(g[f"inputs_{s}"][r])
            P, q, l, u, cone = ref.qp()
            pv, sr = kkt_certificate(P, q, cone, l, u, -f[r])
            assert pv < 1e-9 and sr < 1e-8, (name, s, r, pv, sr)
            assert np.abs(f[r] - fx).max() < (1e-6 if h <= 10 else 1e-5) * max(np.abs(fx).max(), 1.0), (name, s, r, np.abs(f[r] - fx).max())
            swing = np.repeat(g[f"inputs_{s}"][r, 28:28 + 4 * h] == 0, 3)
            # eliminated feet (mpc_osqp.cc:838-856): `qp_sol = 0.0f` (:926) NEGATED on the way out (:940-942) -- exact zeros with the sign bit set
            assert (f[r][swing] == 0.0).all() and np.signbit(f[r][swing]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["solver_h10_cfg3", "solver_h10_edge", "solver_h16_cfg4", "solver_h20_cfg5"])
def test_exact_solver_batch_is_the_unique_optimum(name):
    """MPC_SOLVER_EXACT on a batch (mixed robots / gaits, the edge cases, every compiled horizon)."""
    import torch
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    from tests.helpers import inertia9_from_diag
    handle = {}

    def solve(g, n, s):
        if "gpu" not in handle:
            handle["gpu"] = BatchedConvexMpc(g["mass"][:n], inertia9_from_diag(g["inertia_diag"][:n]), int(g["h"]), float(g["dt_mpc"]), float(g["alpha"]), device="cuda:0", solver="exact")
        f, info = handle["gpu"].solve(torch.from_numpy(g[f"inputs_{s}"][:n]).cuda())
        torch.cuda.synchronize()
        return f.cpu().numpy(), info.cpu().numpy()
    _check_exact(solve, name, 12)


@pytest.mark.parametrize("name,n,route", [("solver_h10_cfg3", 8, 0), ("solver_h10_edge", 10, 0), ("solver_h16_cfg4", 3, 0), ("solver_h20_cfg5", 2, 0),
                                          ("solver_h10_cfg3", 4, 1), ("solver_h16_cfg4", 2, 1), ("solver_h10_cfg3", 6, 3), ("solver_h16_cfg4", 2, 3)])
def test_exact_solver_on_the_host_emulation_is_the_unique_optimum(name, n, route):
    """The exact-optimum mode of the kernel code on the host emulation.  route 0: as the product runs it (the dual active-set method of
    mpc_wrench.h active_set, its iterate checked for optimality, the verifying polish refining it only if that check fails; the ADMM route
    only if that fails too); route 1: the ADMM route alone (run<true>: staged ADMM + verified active-set polish), which is the product's
    second launch; route 3: route 0 with the direct check forced to fail (every robot takes the factorise-and-refine rounds)."""
    import os
    from tests.emu.emu import EmuBatch, lib
    handle = {}
    if route == 3:
        os.environ["EMU_FORCE_DIRECT_FAIL"] = "1"

    def solve(g, n, s):
        if "emu" not in handle:
            handle["emu"] = EmuBatch(g["mass"][:n], g["inertia_diag"][:n], int(g["h"]), float(g["dt_mpc"]), float(g["alpha"]))
        f = handle["emu"].solve(g[f"inputs_{s}"][:n], exact=True)
        return f, handle["emu"].info.copy()
    try:
        lib().emu_set_exact_route(0 if route == 3 else route)
        _check_exact(solve, name, n)
        if route == 3:
            assert (handle["emu"].info[:, 6] >= 2).all(), handle["emu"].info[:, 6]      # polish rounds: the direct check + at least one refinement round
    finally:
        lib().emu_set_exact_route(0)
        os.environ.pop("EMU_FORCE_DIRECT_FAIL", None)


def test_active_set_method_alone_certifies_nearly_every_robot():
    """The first launch on its own (no ADMM route behind it): the dual active-set method must end on a certified set for (nearly) every
    robot of the workloads -- it is what makes the exact mode fast; a robot it gives up on costs a second launch."""
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload
    from tests.emu.emu import EmuBatch, lib
    try:
        lib().emu_set_exact_route(2)
        for h, cfg, n in ((10, 2, 32), (10, 3, 32), (16, 4, 8), (20, 5, 4)):
            wl = make_solver_workload(n, h=h, seed=77, config=cfg)
            emu = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
            emu.solve(wl.inputs, exact=True)
            assert (emu.info[:, 1] == 1).mean() >= 0.9, (h, cfg, emu.info[:, 1])
            assert emu.info[:, 0].max() <= 3 * emu.info[:, 0].mean()            # passes (adds + drops): no long tail
    finally:
        lib().emu_set_exact_route(0)


# ---- exact mode: the working set of the previous call seeds the active-set method (mpc_wrench.h seed_working_set) -------------------------
def _warm_sequence(make, h, cfg, n, steps=4):
    """`make(wl)` -> (solve(records) -> (forces, info), reset()).  Consecutive calls of a controller: the gait moves on by one MPC step."""
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
    from oracle.refmpc import RefConvexMpc
    from tests.helpers import kkt_certificate
    wl = make_solver_workload(n, h=h, seed=31, config=cfg)
    solve, reset = make(wl)
    w, passes = wl, []
    for s in range(steps):
        f, info = solve(w.inputs)
        assert (info[:, 1] == 1).all(), (h, cfg, s, info[:, 1])
        passes.append(info[:, 0].mean())
        for r in range(0, n, max(1, n // 4)):      # the optimum, whatever the method started from: the KKT conditions of the oracle-assembled QP
            d = wl.inertia_diag[r]
            ref = RefConvexMpc(wl.mass[r], [d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]], 4, h, wl.dt_mpc, wl.alpha)
            ref.assemble_only(w.inputs[r])
            P, q, l, u, cone = ref.qp()
            pv, sr = kkt_certificate(P, q, cone, l, u, -f[r])
            assert pv < 1e-9 and sr < 1e-8, (h, cfg, s, r, pv, sr)
        w = perturb_workload(w, 900 + s)
    assert max(passes[1:]) < 0.5 * passes[0], passes          # seeded starts: a fraction of the cold start's working-set changes
    reset()
    f, info = solve(w.inputs)                                  # a new ConvexMpc object knows no working set either
    assert info[:, 0].mean() > 0.7 * passes[0], (info[:, 0].mean(), passes)
    return passes


@pytest.mark.parametrize("h,cfg,n,scalar", [(10, 2, 16, False), (10, 3, 12, False), (16, 4, 4, False), (20, 5, 3, False), (10, 3, 8, True), (8, 2, 6, False), (3, 2, 4, False)])
def test_exact_mode_warm_working_set_on_the_host_emulation(h, cfg, n, scalar, monkeypatch):
    """scalar: the seeded Gram matrix inverted by the scalar sweep also where the workgroup is one wavefront (h <= 10: otherwise the blocked inverse on the
    matrix pipe, seed_inverse_mfma, which hands over to the scalar sweep only for a dependent row)."""
    from tests.emu.emu import EmuBatch
    if scalar:
        monkeypatch.setenv("EMU_SEED_SCALAR", "1")

    def make(wl):
        e = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
        cold = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
        cold.warm_sets = False

        def solve(rec):
            f, fc = e.solve(rec, exact=True), cold.solve(rec, exact=True)
            # the same optimum as the method started empty reaches (the returned point is the method's iterate: rounding of its path, not more)
            assert np.abs(f - fc).max() <= 1e-9 * max(np.abs(fc).max(), 1.0)
            return f, e.info.copy()

        def reset():
            e.seed[:] = 0
        return solve, reset
    _warm_sequence(make, h, cfg, n)


@pytest.mark.parametrize("h,cfg,n,trial", [(10, 2, 10, 0), (10, 3, 9, 1), (16, 4, 3, 2), (20, 5, 2, 3), (6, 2, 5, 4)])
def test_exact_mode_is_immune_to_a_wrong_seed(h, cfg, n, trial):
    """The stored working set is a hint, never an input: with the rows of a valid seed record scrambled (rows the optimum does not hold, wrong sides, rows that
    are linearly dependent on each other, too many rows -- but the record's fixed-foot flags intact, so that it is accepted as a seed), the call still ends on the
    certified optimum the method reaches from the empty set."""
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
    from tests.emu.emu import EmuBatch
    rng = np.random.default_rng(100 + trial)
    wl = make_solver_workload(n, h=h, seed=40 + trial, config=cfg)
    e = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    cold = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    cold.warm_sets = False
    w = wl
    e.solve(w.inputs, exact=True)
    for rep in range(4):
        good = e.seed.copy()
        assert ((good >> 11) & 1).all()                              # every foot left a valid code
        rows = rng.integers(0, 3, size=good.shape + (5,))            # 0 free, 1 at its lower, 2 at its upper bound -- at random
        if rep == 1: rows[...] = 2                                   # every row at its upper bound: five rows for three variables
        if rep == 2: rows[...] = 1
        scr = (good & ~0x3ff)
        for r in range(5): scr |= rows[..., r] << (2 * r)
        keep = rng.random(good.shape) < (0.5 if rep == 3 else 0.0)   # (rep 3: half of the feet keep their true rows)
        e.seed[:] = np.where(keep | (((good >> 10) & 1) == 1), good, scr)
        w = perturb_workload(w, 500 + rep) if rep % 2 == 0 else w     # the seed is applied "moved by a step" or "as it is"
        f, fc = e.solve(w.inputs, exact=True), cold.solve(w.inputs, exact=True)
        assert (e.info[:, 1] == 1).all() and (cold.info[:, 1] == 1).all()
        assert np.abs(f - fc).max() <= 1e-8 * max(np.abs(fc).max(), 1.0), (rep, np.abs(f - fc).max())


@pytest.mark.parametrize("h,cfg,n", [(10, 3, 8), (16, 4, 3), (20, 5, 2), (8, 2, 4)])
def test_exact_mode_seed_gram_matrix_entry_by_entry_equals_the_one_by_columns(h, cfg, n, monkeypatch, capfd):
    """The seed's Gram matrix N^T H^-1 N is formed entry by entry from the held tiles of the factorisation (mpc_wrench.h seed_working_set); the debug build
    of the emulation also forms it the long way -- one application of H^-1 per seeded row -- and reports the largest difference."""
    import re
    from tests.emu.emu import EmuBatch
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
    monkeypatch.setenv("EMU_GRAM_CHECK", "1")
    wl = make_solver_workload(n, h=h, seed=77, config=cfg)
    e = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    w = wl
    for s in range(3):
        e.solve(w.inputs, exact=True)
        w = perturb_workload(w, 300 + s)
    lines = re.findall(r"seed Gram matrix: K (\d+) direct vs by columns: max \|diff\| (\S+) \(largest diagonal entry (\S+)\)", capfd.readouterr().err)
    assert len(lines) >= n, lines                                   # the seeded calls (the second and third of every robot) report
    assert all(int(k) > 0 and float(d) <= 1e-12 * float(big) for k, d, big in lines), lines


@pytest.mark.gpu
@pytest.mark.parametrize("h,cfg,n", [(10, 2, 512), (10, 3, 96), (16, 4, 64), (20, 5, 32)])
def test_exact_mode_warm_working_set(h, cfg, n):
    import torch
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    from tests.helpers import inertia9_from_diag

    def make(wl):
        gpu = BatchedConvexMpc(wl.mass, inertia9_from_diag(wl.inertia_diag), h, wl.dt_mpc, wl.alpha, device="cuda:0", solver="exact")

        def solve(rec):
            f, info = gpu.solve(torch.from_numpy(rec).cuda())
            torch.cuda.synchronize()
            return f.cpu().numpy().copy(), info.cpu().numpy().copy()
        return solve, gpu.reset
    _warm_sequence(make, h, cfg, n)


def test_exact_mode_seed_survives_changes_of_the_contact_pattern():
    """Calls whose contact pattern is neither the previous one nor the previous one moved by a step (another gait, another phase, a batch of other robots'
    records) must not be hurt by the stored working set: every call ends on the certified optimum, equal to what the method reaches from the empty set."""
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
    from tests.emu.emu import EmuBatch
    n, h = 9, 10
    a = make_solver_workload(n, h=h, seed=5, config=3)
    b = make_solver_workload(n, h=h, seed=6, config=3, step_index=50)      # (the gait assignment rotates with step_index: other contact tables)
    b.inputs[:, :] = make_solver_workload(n, h=h, seed=6, config=3, step_index=50).inputs
    e = EmuBatch(a.mass, a.inertia_diag, h, a.dt_mpc, a.alpha)
    cold = EmuBatch(a.mass, a.inertia_diag, h, a.dt_mpc, a.alpha)
    cold.warm_sets = False
    seq = [a.inputs, perturb_workload(a, 1).inputs, b.inputs, a.inputs, perturb_workload(perturb_workload(a, 1), 2).inputs, b.inputs[::-1].copy()]
    for rec in seq:
        f, fc = e.solve(rec, exact=True), cold.solve(rec, exact=True)
        assert (e.info[:, 1] == 1).all() and (cold.info[:, 1] == 1).all()
        assert np.abs(f - fc).max() <= 1e-8 * max(np.abs(fc).max(), 1.0)
