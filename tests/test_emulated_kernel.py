"""CPU tests of the HIP kernel's algorithm through the host emulation of mpc_core.h (tests/emu):
every barrier-separated phase is executed for all emulated threads, forwards and backwards."""
import numpy as np
import pytest

import rl_mpc_locomotion_amd  # noqa: F401
from tests.emu.emu import EmuBatch
from tests.helpers import GRF_RTOL, grf_rtol, grf_relerr, load_golden


@pytest.mark.parametrize("name", ["solver_h10_cfg2", "solver_h10_cfg3", "solver_h16_cfg4", "solver_h20_cfg5", "solver_h10_stress", "solver_h10_edge", "solver_h16_polish"])
def test_emulated_kernel_matches_golden(name):
    g = load_golden(name)
    h, n = int(g["h"]), len(g["mass"])
    emu = EmuBatch(g["mass"], g["inertia_diag"], h, float(g["dt_mpc"]), float(g["alpha"]))
    for s in range(int(g["steps"])):
        f = emu.solve(g[f"inputs_{s}"])
        gi = g[f"info_{s}"]
        assert np.array_equal(emu.info[:, :4], gi), f"step {s}: OSQP decisions differ"
        ok = gi[:, 1] == 1
        assert ok.all()
        assert grf_relerr(f, g[f"forces_{s}"], first_step_only=False).max() < grf_rtol(name)


def test_thread_order_independence():
    """Forward and reverse thread order inside every phase must give bit-identical results
    (an intra-phase data race would break this)."""
    g = load_golden("solver_h10_cfg3")
    sel = slice(0, 12)
    a = EmuBatch(g["mass"][sel], g["inertia_diag"][sel], 10, float(g["dt_mpc"]), float(g["alpha"]))
    b = EmuBatch(g["mass"][sel], g["inertia_diag"][sel], 10, float(g["dt_mpc"]), float(g["alpha"]))
    for s in range(2):
        fa = a.solve(g[f"inputs_{s}"][sel], reverse=False)
        fb = b.solve(g[f"inputs_{s}"][sel], reverse=True)
        assert np.array_equal(fa, fb) and np.array_equal(a.state, b.state) and np.array_equal(a.info, b.info)


def test_reset_semantics_cold_equals_fresh():
    """Zeroing a robot's state record makes its next solve the cold 'osqp_setup' solve."""
    g = load_golden("solver_h10_cfg2")
    sel = slice(0, 8)
    a = EmuBatch(g["mass"][sel], g["inertia_diag"][sel], 10, float(g["dt_mpc"]), float(g["alpha"]))
    f0 = a.solve(g["inputs_0"][sel]).copy()
    a.solve(g["inputs_1"][sel])
    a.state[:] = 0
    f2 = a.solve(g["inputs_0"][sel])
    assert np.array_equal(f0, f2)


def test_no_read_of_unwritten_registers_or_lds():
    """A workgroup's registers and LDS start undefined on the device, while the emulation starts from zeros.  With everything but
    the tile registers (which the kernels zero) poisoned with NaN / -1 patterns the results must not move by a bit -- in both solver
    modes, all horizons, and on the edge cases."""
    from tests.emu.emu import lib
    for name, n in (("solver_h10_cfg3", 8), ("solver_h10_edge", 10), ("solver_h16_cfg4", 2), ("solver_h20_cfg5", 1)):
        g = load_golden(name)
        h = int(g["h"])
        runs = []
        try:
            for poison in (0, 1):
                lib().emu_set_poison(poison)
                emu = EmuBatch(g["mass"][:n], g["inertia_diag"][:n], h, float(g["dt_mpc"]), float(g["alpha"]))
                out = [emu.solve(g[f"inputs_{s}"][:n]).copy() for s in range(2)]
                ex = EmuBatch(g["mass"][:n], g["inertia_diag"][:n], h, float(g["dt_mpc"]), float(g["alpha"]))
                out.append(ex.solve(g["inputs_0"][:n], exact=True).copy())
                runs.append((out, emu.info.copy(), ex.info.copy()))
        finally:
            lib().emu_set_poison(0)
        for a, b in zip(runs[0][0], runs[1][0]):
            assert np.array_equal(a, b, equal_nan=True), name
        assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2]), name


def test_non_solved_statuses_match_osqp():
    """info[:, :4] against the vendored OSQP on robots that do NOT end SOLVED: primal infeasible QPs (OSQP's certificate,
    auxil.c:364-424, at the check where OSQP finds it), and ordinary QPs cut off by a small max_iter -- MAX_ITER_REACHED or, when the
    last check passes at ten times the tolerances, SOLVED_INACCURATE (osqp.c:563-568).  Neither returns forces (mpc_osqp.cc:788-794)."""
    from oracle.refmpc import RefBatch
    from rl_mpc_locomotion_amd.synthetic import perturb_workload
    from tests.emu.emu import lib
    from tests.helpers import infeasible_workload
    wl, inp = infeasible_workload()
    n, h = len(wl.mass), 10
    emu = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    for step in range(3):
        f = emu.solve(inp)
        fr = ref.solve(inp, nthreads=4)
        assert np.array_equal(emu.info[:, :4], ref.info[:, :4]), step
        assert (ref.info[:n // 3, 1] == -3).all() and (ref.info[n // 3:, 1] == 1).all()
        assert np.isnan(f[:n // 3]).all()                       # (rows the emulation wrapper pre-fills with NaN: no forces written)
        ok = ref.info[:, 1] == 1
        assert grf_relerr(f[ok], fr[ok]).max() < GRF_RTOL
    try:
        lib().emu_set_max_iter(25)
        emu = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
        ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
        ref.set_max_iter(25)
        seen = set()
        w = wl
        for step in range(3):
            emu.solve(w.inputs)
            ref.solve(w.inputs, nthreads=4)
            assert np.array_equal(emu.info[:, :4], ref.info[:, :4]), step
            seen |= set(ref.info[:, 1].tolist())
            w = perturb_workload(w, 3 + step)
        assert {-2, 2} <= seen                                   # MAX_ITER_REACHED and SOLVED_INACCURATE both occurred
    finally:
        lib().emu_set_max_iter(0)


def test_invalid_bounds_are_refused():
    """l > u (friction coefficient below -1): OSQP refuses the data (validate_data) and the reference is left without a workspace;
    here the robot reports NON_CVX, writes no forces and restarts cold."""
    from rl_mpc_locomotion_amd import layout as L
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload
    wl = make_solver_workload(6, h=10, seed=3, config=2)
    inp = wl.inputs.copy()
    inp[:2, L.in_friction(10):L.in_friction(10) + 4] = -2.0
    emu = EmuBatch(wl.mass, wl.inertia_diag, 10, wl.dt_mpc, wl.alpha)
    f = emu.solve(inp)
    assert (emu.info[:2, 1] == -7).all() and (emu.info[2:, 1] == 1).all() and np.isnan(f[:2]).all() and (emu.state[:2] == 0).all()


def test_job_split_equals_the_single_pass():
    """The persistent kernel runs a solve as two jobs (ADMM part; polish on a freshly started workgroup that re-loads the problem):
    results must not move by a bit against the single pass, with the second workgroup's registers and LDS poisoned."""
    from tests.emu.emu import lib
    for name, n in (("solver_h10_cfg3", 10), ("solver_h16_polish", 3), ("solver_h20_cfg5", 1)):
        g = load_golden(name)
        h = int(g["h"])
        a = EmuBatch(g["mass"][:n], g["inertia_diag"][:n], h, float(g["dt_mpc"]), float(g["alpha"]))
        b = EmuBatch(g["mass"][:n], g["inertia_diag"][:n], h, float(g["dt_mpc"]), float(g["alpha"]))
        for s in range(min(2, int(g["steps"]))):
            fa = a.solve(g[f"inputs_{s}"][:n])
            try:
                lib().emu_set_split(1); lib().emu_set_poison(1)
                fb = b.solve(g[f"inputs_{s}"][:n])
            finally:
                lib().emu_set_split(0); lib().emu_set_poison(0)
            assert np.array_equal(fa, fb) and np.array_equal(a.info, b.info) and np.array_equal(a.state, b.state), name


@pytest.mark.parametrize("h", [2, 3, 5, 7, 8, 9, 11, 12, 13, 14, 15, 17, 18, 19])
def test_other_planning_horizons_match_osqp(h):
    """ConvexMpc accepts any planning_horizon (mpc_osqp.cc:186-190, 508-574); the library ships every horizon from 2 to 20
    (mpc_supported_horizons; one translation unit each, csrc/mpc_horizon.hip).  The horizons beyond BASELINE's 10 / 16 / 20 against the live
    oracle -- odd ones (the even row stride of the partial products), the shortest (the set-up scratch, the exact mode's slot count), the
    ones whose workgroups are two and three wavefronts: decisions identical, forces within the tolerance, both modes."""
    from oracle.refmpc import RefBatch
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
    wl = make_solver_workload(6 if h > 12 else 10, h=h, seed=3, config=2)
    emu = EmuBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    for s in range(3):
        f = emu.solve(wl.inputs)
        fr = ref.solve(wl.inputs, nthreads=4)
        assert np.array_equal(emu.info[:, :4], ref.info[:, :4]), (h, s)
        assert grf_relerr(f, fr, first_step_only=False).max() < GRF_RTOL
        wl = perturb_workload(wl, 4 + s)
    emu.solve(wl.inputs, exact=True)
    assert (emu.info[:, 1] == 1).all()
