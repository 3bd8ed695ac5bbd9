"""GPU parity tests proper: the HIP path, called through the C ABI, against the committed golden
vectors and against the live oracle (vendored OSQP) on seeded inputs."""
import numpy as np
import pytest

import rl_mpc_locomotion_amd  # noqa: F401
from rl_mpc_locomotion_amd.synthetic import make_solver_workload, perturb_workload
from tests.helpers import GRF_RTOL, grf_rtol, grf_relerr, inertia9_from_diag, load_golden

pytestmark = pytest.mark.gpu


def _gpu(mass, inertia_diag, h, dt, alpha):
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    return BatchedConvexMpc(mass, inertia9_from_diag(inertia_diag), h, dt, alpha, device="cuda:0")


def _solve(gpu, rec):
    import torch
    f, info = gpu.solve(torch.from_numpy(np.ascontiguousarray(rec, dtype=np.float32)).to("cuda:0"))
    torch.cuda.synchronize()
    return f.cpu().numpy().copy(), info.cpu().numpy().copy()


@pytest.mark.parametrize("name", ["solver_h10_cfg2", "solver_h10_cfg3", "solver_h16_cfg4", "solver_h20_cfg5", "solver_h10_stress", "solver_h10_edge", "solver_h16_polish"])
def test_hip_matches_golden(name):
    g = load_golden(name)
    gpu = _gpu(g["mass"], g["inertia_diag"], int(g["h"]), float(g["dt_mpc"]), float(g["alpha"]))
    for s in range(int(g["steps"])):
        f, info = _solve(gpu, g[f"inputs_{s}"])
        assert np.array_equal(info[:, :4], g[f"info_{s}"]), f"step {s}: OSQP decisions differ"
        assert grf_relerr(f, g[f"forces_{s}"], first_step_only=False).max() < grf_rtol(name)


@pytest.mark.parametrize("config,n", [(2, 1024), (3, 768)])
def test_hip_matches_live_oracle(config, n):
    from oracle.refmpc import RefBatch
    h = 10
    wl = make_solver_workload(n, h=h, seed=100 + config, config=config)
    gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    for s in range(3):
        f, info = _solve(gpu, wl.inputs)
        fr = ref.solve(wl.inputs, nthreads=8)
        assert np.array_equal(info[:, :4], ref.info[:, :4].astype(np.int32))
        ok = ref.info[:, 1] == 1
        assert grf_relerr(f[ok], fr[ok]).max() < GRF_RTOL
        wl = perturb_workload(wl, 500 + s)


@pytest.mark.parametrize("h", [2, 3, 5, 7, 8, 9, 11, 12, 13, 14, 15, 17, 18, 19])
def test_other_planning_horizons_match_live_oracle(h):
    """planning_horizon beyond BASELINE's 10 / 16 / 20: ConvexMpc takes any (mpc_osqp.cc:186-190, 508-574), the library ships 2 .. 20
    (mpc_supported_horizons).  Cold + two warm-started solves against the vendored OSQP: decisions equal, forces within the tolerance."""
    from oracle.refmpc import RefBatch
    from rl_mpc_locomotion_amd import _lib
    import ctypes as C
    hs = (C.c_int * 32)()
    cnt = _lib.lib().mpc_supported_horizons(hs, 32)
    assert list(hs[:cnt]) == list(range(2, 21))
    n = 64
    wl = make_solver_workload(n, h=h, seed=40 + h, config=2)
    gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    for s in range(3):
        f, info = _solve(gpu, wl.inputs)
        fr = ref.solve(wl.inputs, nthreads=8)
        assert np.array_equal(info[:, :4], ref.info[:, :4].astype(np.int32)), (h, s)
        ok = ref.info[:, 1] == 1
        assert grf_relerr(f[ok], fr[ok], first_step_only=False).max() < GRF_RTOL
        wl = perturb_workload(wl, 600 + s)


def test_full_size_properties_4096():
    """BASELINE configs[1] at full size: size-independent properties instead of the (slow) oracle --
    determinism across two handles, swing-leg forces vanish, stance forces inside the friction
    pyramid and force limits, warm start not slower than cold."""
    n, h = 4096, 10
    wl = make_solver_workload(n, h=h, seed=0, config=2)
    a = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    b = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    fa, ia = _solve(a, wl.inputs)
    fb, ib = _solve(b, wl.inputs)
    assert np.array_equal(fa, fb) and np.array_equal(ia, ib)          # bit-identical per robot
    assert (ia[:, 1] == 1).all()
    from rl_mpc_locomotion_amd import layout as L
    c = wl.inputs[:, L.IN_CONTACT:L.IN_CONTACT + 4 * h].astype(bool)
    f = -fa.reshape(n, 4 * h, 3)                                       # back to the QP variable x
    fz_max = (wl.mass * 9.8 * 10)[:, None]
    tol = 2e-3 * fz_max                                                 # eps_abs/rel = 1e-3 ADMM accuracy
    assert (np.abs(f[~c]) < tol.max()).all()
    mu = 0.4
    assert ((np.abs(f[..., 0]) <= mu * f[..., 2] + tol) | ~c).all()
    assert ((np.abs(f[..., 1]) <= mu * f[..., 2] + tol) | ~c).all()
    assert ((f[..., 2] <= fz_max + tol) | ~c).all() and ((f[..., 2] >= 0.1 * fz_max / 10 - tol) | ~c).all()
    wl2 = perturb_workload(wl, 1)
    _, iw = _solve(a, wl2.inputs)
    assert iw[:, 0].mean() < ia[:, 0].mean()


def test_reset_subset_is_cold_start():
    n, h = 64, 10
    wl = make_solver_workload(n, h=h, seed=9, config=2)
    gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    f0, i0 = _solve(gpu, wl.inputs)
    wl2 = perturb_workload(wl, 2)
    _solve(gpu, wl2.inputs)
    ids = np.arange(0, n, 2, dtype=np.int32)
    gpu.reset(ids)
    f2, i2 = _solve(gpu, wl.inputs)
    assert np.array_equal(f2[ids], f0[ids]) and np.array_equal(i2[ids], i0[ids])
    assert (i2[1::2, 5] == 0).all() and (i2[ids, 5] == 1).all()      # first_run flag


def test_mpc_osqp_shim_signature():
    """The per-robot plugin seam: same 7 + 13 positional arguments as mpc_osqp.ConvexMpc."""
    from rl_mpc_locomotion_amd import mpc_osqp as mpc
    from rl_mpc_locomotion_amd import layout as L
    g = load_golden("solver_h10_cfg2")
    h = 10
    d = g["inertia_diag"][0]
    obj = mpc.ConvexMpc(float(g["mass"][0]), [d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]], 4, h, float(g["dt_mpc"]),
                        float(g["alpha"]), mpc.OSQP)      # the OSQP branch: BASELINE's comparator (QPOASES = the exact optimum: tests/test_dropin.py)
    r = g["inputs_0"][0]
    out = obj.compute_contact_forces(
        list(r[0:13]), r[13:16], r[16:19], r[19:22], r[22:25], r[25:28], r[28:28 + 4 * h],
        r[L.in_footpos(h):L.in_footpos(h) + 12], r[L.in_friction(h):L.in_friction(h) + 4],
        r[L.in_des_pos(h):L.in_des_pos(h) + 3], r[L.in_des_vel(h):L.in_des_vel(h) + 3],
        r[L.in_des_rpy(h):L.in_des_rpy(h) + 3], r[L.in_des_angvel(h):L.in_des_angvel(h) + 3])
    assert isinstance(out, list) and len(out) == 12 * h
    ref = g["forces_0"][0]
    assert np.abs(np.array(out) - ref).max() / max(np.abs(ref).max(), 1.0) < GRF_RTOL
    assert mpc.TEST == 42 and mpc.OSQP == mpc.QPSolverName.OSQP


def test_non_finite_input_fails_cleanly():
    """A robot fed NaNs must terminate (bounded iterations), report a non-SOLVED status and leave its
    force row untouched (the reference returns [] there, mpc_osqp.cc:781-794); its neighbours are unaffected."""
    n, h = 8, 10
    wl = make_solver_workload(n, h=h, seed=3, config=2)
    gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    good, _ = _solve(gpu, wl.inputs)
    bad_in = wl.inputs.copy()
    bad_in[2, 16:19] = np.nan                      # com_velocity of robot 2
    gpu2 = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    import torch
    sentinel = torch.full((n, 12 * h), 123.0, dtype=torch.float64, device="cuda:0")
    f, info = gpu2.solve(torch.from_numpy(bad_in).cuda(), forces=sentinel)
    torch.cuda.synchronize()
    f = f.cpu().numpy(); info = info.cpu().numpy()
    assert info[2, 1] != 1 and (f[2] == 123.0).all()
    keep = np.arange(n) != 2
    assert (info[keep, 1] == 1).all() and np.array_equal(f[keep], good[keep])
    # the NEXT call: a failed robot's record is cleared, so it solves again as a cold robot (the vendored OSQP itself reports SOLVED
    # with NaN iterates on NaN data and stays poisoned until the object is rebuilt: tests/test_oracle.py::test_reference_is_poisoned_by_nan)
    f2, info2 = gpu2.solve(torch.from_numpy(wl.inputs).cuda())
    torch.cuda.synchronize()
    f2 = f2.cpu().numpy(); info2 = info2.cpu().numpy()
    assert info2[2, 1] == 1 and info2[2, 5] == 1                       # solved, as an "osqp_setup" call
    assert np.array_equal(f2[2], good[2]) and np.array_equal(info2[2, :5], _solve(_gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha), wl.inputs)[1][2, :5])


def test_reset_with_device_ids_equals_host_ids():
    """env_ids as a device tensor (VecTask.reset_idx) takes the stream-ordered device path; same result as host ids."""
    import torch
    n, h = 12, 10
    wl = make_solver_workload(n, h=h, seed=9, config=3)
    a = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    b = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    _solve(a, wl.inputs); _solve(b, wl.inputs)
    ids = [1, 4, 7, 11]
    a.reset(ids)
    b.reset(torch.tensor(ids, dtype=torch.int64, device="cuda:0"))
    wl2 = perturb_workload(wl, 3)
    fa, ia = _solve(a, wl2.inputs); fb, ib = _solve(b, wl2.inputs)
    assert np.array_equal(fa, fb) and np.array_equal(ia, ib)
    assert ia[ids, 5].tolist() == [1, 1, 1, 1] and ia[[0, 2, 3], 5].tolist() == [0, 0, 0]


def test_single_robot_batch_and_state_roundtrip():
    """n = 1 works, and the warm-start state can be saved / restored (determinism across handles)."""
    h = 10
    wl = make_solver_workload(1, h=h, seed=5, config=2)
    a = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    _solve(a, wl.inputs)
    st = a.get_state()
    wl2 = perturb_workload(wl, 1)
    fa, ia = _solve(a, wl2.inputs)
    b = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    b.set_state(st)
    fb, ib = _solve(b, wl2.inputs)
    assert np.array_equal(fa, fb) and np.array_equal(ia, ib) and ia[0, 5] == 0
    assert a.device_bytes() > 0


def test_bad_arguments_raise():
    import torch
    from rl_mpc_locomotion_amd import _lib
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    for bad_h in (1, 21, 0, -3):
        with pytest.raises(_lib.MpcLibraryError):
            BatchedConvexMpc([18.0], [[0.03, 0, 0, 0, 0.16, 0, 0, 0, 0.17]], bad_h, 0.02)    # outside the shipped range 2 .. 20 (MPC_E_HORIZON)
    g = _gpu(np.array([18.0]), np.array([[0.03, 0.16, 0.17]]), 10, 0.02, 1e-5)
    with pytest.raises(ValueError):
        g.solve(torch.zeros((1, 95), dtype=torch.float32, device="cuda:0"))              # wrong record length
    with pytest.raises(ValueError):
        g.solve(torch.zeros((1, 96), dtype=torch.int32, device="cuda:0"))                # wrong dtype (float32 / float64 / float16 records only)


@pytest.mark.gpu
def test_dispatch_order_does_not_change_results():
    """After the first launch the workgroup -> robot map is re-sorted by solve time (longest first, order_block in the assembly launch; two
    counting-sort passes above 8192 robots).  Per-robot results must not depend on it: a robot solved inside a large batch
    equals the same robot solved alone, bit for bit, over several warm-started steps."""
    import torch
    n, h = 8200, 10
    wl = make_solver_workload(n, h=h, seed=21, config=3)
    big = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    pick = [0, 17, 4095, 8191, 8199]
    solo = [_gpu(wl.mass[i:i + 1], wl.inertia_diag[i:i + 1], h, wl.dt_mpc, wl.alpha) for i in pick]
    w = wl
    for step in range(3):
        f, info = _solve(big, w.inputs)
        assert (info[:, 1] == 1).all()
        for i, s in zip(pick, solo):
            fs, infos = _solve(s, w.inputs[i:i + 1])
            assert np.array_equal(fs[0], f[i]) and np.array_equal(infos[0], info[i])
        w = perturb_workload(w, 300 + step)


@pytest.mark.parametrize("config,h,n", [(4, 16, 256), (5, 20, 256)])
def test_warm_started_long_horizons_match_live_oracle(config, h, n):
    """BASELINE configs[3] / configs[4] (h = 16 with random ground normals, h = 20) over a cold and two WARM-STARTED solves against the
    vendored OSQP, one workspace per robot: equal iterations / status / polish / rho updates per robot, forces within the tolerance over
    the whole horizon."""
    from oracle.refmpc import RefBatch
    wl = make_solver_workload(n, h=h, seed=700 + config, config=config)
    gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    for s in range(3):
        f, info = _solve(gpu, wl.inputs)
        fr = ref.solve(wl.inputs, nthreads=8)
        assert np.array_equal(info[:, :4], ref.info[:, :4].astype(np.int32)), (h, s)
        if s:
            assert (info[:, 5] == 0).all()          # not a first ("osqp_setup") call: warm-started
        ok = ref.info[:, 1] == 1
        assert ok.mean() > 0.95 and grf_relerr(f[ok], fr[ok], first_step_only=False).max() < GRF_RTOL, (h, s)
        wl = perturb_workload(wl, 710 + s)


def test_fp16_state_inputs_h20():
    """BASELINE configs[4]: "65536 Aliengo, horizon = 20, fp16 state".  The input records are ROUNDED TO FLOAT16 (SURVEY 8(d) rounds rpy that
    way -- the reference's own com_roll_pitch_yaw is numpy.float16 -- here every one of the 13 arguments) and handed to
    mpc_batch_solve_f16 as a torch.float16 tensor.  The oracle is fed the same rounded record (float16 -> float32 is exact): decisions
    equal, forces within the tolerance, over a cold and two warm-started solves; and the float64 entry on the widened values gives the
    same bits (the storage type changes nothing but the load)."""
    import torch
    from oracle.refmpc import RefBatch
    n, h = 256, 20
    wl = make_solver_workload(n, h=h, seed=905, config=5)
    gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    g64 = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    for s in range(3):
        half = wl.inputs.astype(np.float16)
        assert np.isfinite(half.astype(np.float32)).all()
        f, info = gpu.solve(torch.from_numpy(half).to("cuda:0"))
        f2, info2 = g64.solve(torch.from_numpy(half.astype(np.float64)).to("cuda:0"))
        torch.cuda.synchronize()
        f, info = f.cpu().numpy().copy(), info.cpu().numpy().copy()
        assert np.array_equal(f, f2.cpu().numpy()) and np.array_equal(info, info2.cpu().numpy())
        fr = ref.solve(half.astype(np.float32), nthreads=8)
        assert np.array_equal(info[:, :4], ref.info[:, :4].astype(np.int32)), s
        ok = ref.info[:, 1] == 1
        assert ok.mean() > 0.95 and grf_relerr(f[ok], fr[ok], first_step_only=False).max() < GRF_RTOL, s
        wl = perturb_workload(wl, 910 + s)


@pytest.mark.parametrize("config,h,n", [(4, 16, 4096), (5, 20, 8192)])
def test_full_size_properties_long_horizons(config, h, n):
    """BASELINE configs[3] / configs[4] at their per-GPU size (32768 / 8 and 65536 / 8 robots): permutation equivariance
    (robot i's result does not depend on where it sits in the batch, bit for bit), every solve OSQP_SOLVED, swing forces
    vanish, and a 24-robot sample agrees with the oracle."""
    from oracle.refmpc import RefBatch
    wl = make_solver_workload(n, h=h, seed=31, config=config)
    a = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    fa, ia = _solve(a, wl.inputs)
    assert (ia[:, 1] == 1).all()
    perm = np.random.default_rng(0).permutation(n)
    b = _gpu(wl.mass[perm], wl.inertia_diag[perm], h, wl.dt_mpc, wl.alpha)
    fb, ib = _solve(b, wl.inputs[perm])
    assert np.array_equal(fb, fa[perm]) and np.array_equal(ib, ia[perm])
    from rl_mpc_locomotion_amd import layout as L
    c = wl.inputs[:, L.IN_CONTACT:L.IN_CONTACT + 4 * h].astype(bool)
    f = fa.reshape(n, 4 * h, 3)
    assert (np.abs(f[~c]) < 2e-3 * (wl.mass * 98.0).max()).all()
    pick = np.arange(0, n, n // 24)[:24]
    ref = RefBatch(wl.mass[pick], wl.inertia_diag[pick], h, wl.dt_mpc, wl.alpha)
    fr = ref.solve(wl.inputs[pick], nthreads=8)
    assert np.array_equal(ref.info[:, :4], ia[pick][:, :4])
    assert grf_relerr(fa[pick], fr, first_step_only=False).max() < GRF_RTOL


@pytest.mark.parametrize("name", ["solver_h10_cfg3", "solver_h10_edge", "solver_h16_cfg4", "solver_h20_cfg5"])
def test_assembly_and_scaling_records_match_the_oracle(name):
    """the QP the prep kernel builds -- q, l, u, the cone block, and P through its wrench form
    P = BB^T Theta BB + alpha I (B6, th1, th2 of the QP record) -- against the oracle's restated mpc_osqp.cc assembly, element by
    element; and its Ruiz scaling (D, E, c, cold call) against the vendored OSQP's own scaling vectors."""
    import torch
    from oracle.refmpc import RefConvexMpc
    g = load_golden(name)
    h, n = int(g["h"]), min(len(g["mass"]), 10)
    N, M = 12 * h, 20 * h
    gpu = _gpu(g["mass"][:n], g["inertia_diag"][:n], h, float(g["dt_mpc"]), float(g["alpha"]))
    gpu.solve(torch.from_numpy(g["inputs_0"][:n]).cuda())
    qp, sc = gpu.get_qp(), gpu.get_scale()
    dt = float(g["dt_mpc"])
    for r in range(n):
        d = g["inertia_diag"][r]
        ref = RefConvexMpc(g["mass"][r], [d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]], 4, h, dt, float(g["alpha"]))
        ref.solve_flat(g["inputs_0"][r])
        P, q, l, u, cone = ref.qp()
        st = ref.state()
        rec = qp[r]
        np.testing.assert_allclose(rec[:N], q, rtol=1e-12, atol=1e-12 * np.abs(q).max())
        np.testing.assert_array_equal(rec[N:N + M], l)
        np.testing.assert_array_equal(rec[N + M:N + 2 * M], u)
        np.testing.assert_array_equal(rec[N + 2 * M:N + 2 * M + 15].reshape(5, 3), cone)
        o = N + 2 * M + 16
        B6, th1, th2 = rec[o:o + 72].reshape(6, 12), rec[o + 72:o + 108].reshape(6, 6), rec[o + 108:o + 114]
        Th = np.zeros((6 * h, 6 * h))
        for i in range(h):
            for j in range(h):
                mm, dd = h - max(i, j), abs(i - j)
                Th[6 * i:6 * i + 6, 6 * j:6 * j + 6] = (mm * (4 * mm * mm - 1) / 12.0 + dd * mm * mm / 2.0) * th1 + mm * np.diag(th2)
        BB = np.kron(np.eye(h), B6)
        P2 = BB.T @ Th @ BB + float(g["alpha"]) * np.eye(N)
        assert np.abs(P2 - P).max() <= 1e-13 * np.abs(P).max()
        np.testing.assert_allclose(sc[r, :N], st["D"], rtol=1e-11)
        np.testing.assert_allclose(sc[r, N:N + M], st["E"], rtol=1e-11)
        assert abs(sc[r, 2 * N + 3 * M + 60 * h] / st["c"] - 1) < 1e-11       # D[N] E[M] q_s[N] A_s[15 * 4 h] l_s[M] u_s[M] c 1/c job[2]
        # ... and what the solve left for the next call (a13): OSQP's scaled iterates and rho.  This bounds the drift of the iteration against
        # the vendored library -- the scaled cone block is formed as (a E) D from the final scalings here, ten per-pass roundings there
        # (mpc_core.h: scaled_cone_entry) -- on top of the equal decisions: a few ulp in A_s stay a few 1e-9 in the iterates
        mine = gpu.get_state()[r]
        assert abs(mine[2 * N + 2 * M] / st["rho"] - 1) < 1e-6      # (an adopted rho is a ratio of residual norms: equal to ~1e-10, not bit for bit)
        for name_, lo, hi in (("x", 0, N), ("z", N, N + M), ("y", N + M, N + 2 * M)):
            ref_v = st[name_]
            assert np.abs(mine[lo:hi] - ref_v).max() <= 2e-6 * max(np.abs(ref_v).max(), 1e-12), (name_, r)


def test_non_solved_statuses_match_osqp():
    """info[:, :4] of robots that do not end SOLVED against the vendored OSQP: primal infeasible QPs (status -3 at the check where OSQP
    finds its certificate, auxil.c:364-424) and a small max_iter (MAX_ITER_REACHED / SOLVED_INACCURATE, osqp.c:563-568); their force
    rows stay untouched (the reference returns [], mpc_osqp.cc:788-794)."""
    from oracle.refmpc import RefBatch
    from tests.helpers import infeasible_workload
    wl, inp = infeasible_workload(n=48)
    n, h = len(wl.mass), 10
    gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    import torch
    for step in range(3):
        gpu.forces.fill_(777.0)
        f, info = _solve(gpu, inp)
        fr = ref.solve(inp, nthreads=8)
        assert np.array_equal(info[:, :4], ref.info[:, :4].astype(np.int32)), step
        assert (info[:n // 3, 1] == -3).all() and (f[:n // 3] == 777.0).all()
        ok = ref.info[:, 1] == 1
        assert grf_relerr(f[ok], fr[ok]).max() < GRF_RTOL
    gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    gpu.set_max_iter(25)
    ref = RefBatch(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
    ref.set_max_iter(25)
    seen = set()
    w = wl
    for step in range(3):
        f, info = _solve(gpu, w.inputs)
        ref.solve(w.inputs, nthreads=8)
        assert np.array_equal(info[:, :4], ref.info[:, :4].astype(np.int32)), step
        seen |= set(info[:, 1].tolist())
        w = perturb_workload(w, 3 + step)
    assert {-2, 2} <= seen


def test_one_infeasible_robot_does_not_hold_the_launch():
    """4096 robots, one of them with an infeasible QP: OSQP's certificate ends it after 25-50 iterations (it ran to max_iter = 4000,
    ~7 launches' worth of time, before the certificates were evaluated); the launch takes what it takes without that robot."""
    import torch
    n, h = 4096, 10
    wl = make_solver_workload(n, h=h, seed=0, config=2)
    from rl_mpc_locomotion_amd import layout as L
    bad = wl.inputs.copy()
    bad[17, L.in_friction(h):L.in_friction(h) + 4] = -0.4
    times = {}
    for name, inp in (("clean", wl.inputs), ("one infeasible", bad)):
        gpu = _gpu(wl.mass, wl.inertia_diag, h, wl.dt_mpc, wl.alpha)
        gpu.enable_timing()
        d = torch.from_numpy(inp).to("cuda:0")
        for _ in range(3):
            f, info = gpu.solve(d)
        torch.cuda.synchronize()
        times[name] = float(gpu.kernel_times(1)[1][-1])
        st = info.cpu().numpy()
        assert (np.delete(st[:, 1], 17) == 1).all()
        if name != "clean":
            assert st[17, 1] == -3 and st[17, 0] <= 50
    assert times["one infeasible"] < 1.25 * times["clean"], times


@pytest.mark.gpu
def test_bench_contract_line():
    """`python bench.py` prints ONE JSON line with the keys the driver reads (small sizes; the CPU baseline leg uses the oracle)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--robots", "256", "--steps", "3", "--warmup", "2", "--no-secondary", "--no-control-loop"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    b = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["metric"].startswith("MPC control steps/sec (whole node) @ horizon=10, 256 robots") and b["n_gpus"] == 1 and b["steps"] == 3 and b["warmup"] == 2 and b["dtype"] == "f64" and b["scaling"] == "weak" and b["vs_baseline"] is None
    assert b["value"] > 0 and abs(b["value"] - 256 * 3 / (b["ms_per_step"] * 3e-3)) < 1e-6 * b["value"]
    r = b["roofline"]
    assert r["bound"] == "vector_fp64" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = b["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["value"] > 0 and c["one_core"]["cores"] == 1
    assert b["solved_fraction"] == 1.0 and b["max_grf_err_vs_osqp"] < 1e-5
