"""Per-tick controller (ctrl_pre -> solve -> ctrl_post) against golden torques recorded from the UNMODIFIED
reference Python (tests/golden/make_golden_controller.py; the h = 16 / 20 goldens: with horizonLength patched).  CPU: host emulation; GPU: the HIP kernels
through the C ABI."""
import numpy as np
import pytest

import rl_mpc_locomotion_amd  # noqa: F401
from tests.helpers import load_golden

# The controller's float32 arithmetic is the reference's operation for operation -- the ground-normal least squares included
# (csrc/gelsd43.h walks LAPACK's SGELSD as scipy runs it; the normal is bit-identical on every tick of every golden), so on identical
# inputs every argument of every compute_contact_forces call is bit-identical to the reference's and the torques differ only by what
# the fp64 solve differs from the vendored OSQP (tests/helpers.py:GRF_RTOL).  Observed: <= 3.2e-7 relative to max(|tau|_inf, 1 Nm).
TAU_RTOL = 5e-6
GOLDENS_H10 = ["controller_h10_flat", "controller_h10_slope", "controller_h10_config1"]
# controller_h10_gaits: all seven gaits of the reference's dispatch x three robot types (63 robots); controller_h16_* / controller_h20_*:
# BASELINE configs[3] / [4] -- the reference with horizonLength patched (tests/golden/make_golden_controller.py says how)
GOLDENS_NEW = ["controller_h10_gaits", "controller_h16_flat", "controller_h16_slope", "controller_h20_flat", "controller_h20_slope"]
GOLDENS = GOLDENS_H10 + GOLDENS_NEW
# Every operation of StateEstimator.update is reproduced, numpy's float32 transcendentals included: on the AVX-512 machine that minted the goldens np.arccos
# (orientation_tools.py:94) and np.arctan2 (quat_to_rpy, :120-133) on float32 are Intel SVML's __svml_acosf16 / __svml_atan2f16, restated in csrc/svml_acosf.h and checked
# against numpy on every float32 of [-1, 1] and on 1.3 G pairs (tools/acosf/pin.py, profiles/r06_acosf_pinning.txt).  Until round 5 those calls went through libm's / OCML's
# acosf / atan2f (up to 2 ulp away: 3 of the 11 512 estimator samples of the goldens differed after the float16 rounding of the yaw -- rounds 4-5 blamed the arccos; it
# was the arctan2 -- and the tests let such a robot leave the torque comparison); now EVERY sample of every golden is compared: the bookkeeping must find compared == 1.0.


def _relerr(a, b):
    return np.abs(a - b).max(-1) / np.maximum(np.abs(b).max(-1), 1.0)


def _horizon(g):
    return int(g["horizon"]) if "horizon" in g.files else 10


def _normal_prev(g):
    T, n = g["body"].shape[:2]
    return np.concatenate([np.tile(np.array([0, 0, 1], np.float32), (1, n, 1)), g["normal"][:-1]], axis=0)     # estimate of the previous tick


@pytest.mark.parametrize("name", GOLDENS)
def test_emulated_controller_matches_reference_python(name):
    from tests.emu.emu import ctrl_replay
    g = load_golden(name)
    tau, rec, fff = ctrl_replay(g["robot_type"], g["gait_id"], int(g["flat_ground"]), g["dof"], g["est"], g["cmd"], horizon=_horizon(g))
    assert _relerr(tau, g["torque"]).max() < TAU_RTOL
    assert _relerr(fff, g["f_ff"]).max() < TAU_RTOL
    # the MPC ran on every second tick (iterationsBetweenMPC = 2, ConvexMPCLocomotion.py:217-220)
    ran = np.abs(rec).sum(-1) > 0
    assert np.array_equal(ran, g["solved"].astype(bool))
    if "record" in g.files:      # the 13 arguments of every compute_contact_forces call: bit-identical to what the reference passed
        assert np.array_equal(rec[ran], g["record"][ran])


@pytest.mark.parametrize("name", GOLDENS)
def test_emulated_estimator_matches_reference_python(name):
    """StateEstimator.update restated with explicit float16/float32 semantics: every output -- the float16 ones (rpyBody,
    ground_R_body_frame) and the float32 ones (vBody, omegaBody, numpy's float16 @ float32 product through OpenBLAS) -- bit-identical
    to the reference's on every sample of every golden."""
    from tests.emu.emu import estimator_update
    g = load_golden(name)
    T, n = g["body"].shape[:2]
    est = estimator_update(g["body"].reshape(T * n, 13), _normal_prev(g).reshape(T * n, 3)).reshape(T, n, 18)
    bad = (est != g["est"]).any(-1)
    print(f"{name}: {int(bad.sum())} of {T * n} estimator samples differ")
    assert not bad.any()


def _full_run(g, make_ctl, est_of):
    """controller.run over a golden: returns (torque errors of the compared samples, compared [T, n], ticks x robots whose ground normal is not
    bit-identical to the reference's).  A robot would leave the comparison at an estimator sample that differs from the reference's: the callers assert that none does."""
    T, n = g["dof"].shape[:2]
    ctl = make_ctl(g)
    ok = np.ones(n, bool)
    errs, compared, normal_bad = [], np.zeros((T, n), bool), 0
    for k in range(T):
        tau, est, nrm = est_of(ctl, g, k)
        ok &= (est == g["est"][k]).all(-1)
        compared[k] = ok
        errs.append(_relerr(tau, g["torque"][k])[ok])
        if not bool(g["flat_ground"]):
            normal_bad += int((nrm != g["normal"][k]).any(-1).sum())
    return np.concatenate(errs), compared, normal_bad


@pytest.mark.parametrize("name", GOLDENS)
def test_emulated_full_run_matches_reference_python(name):
    """The whole controller.run seam (estimator + ground-normal fit + controller + solve) on the host emulation: ground normal bit-identical
    and every estimator sample bit-identical on every tick, torques inside TAU_RTOL."""
    from tests.emu.emu import EmuLocomotion, estimator_update

    def step(ctl, g, k):
        prev = g["normal"][k - 1] if k else np.tile(np.array([0, 0, 1], np.float32), (g["body"].shape[1], 1))
        tau = ctl.run(g["dof"][k], g["body"][k], g["cmd"][k])
        return tau, estimator_update(g["body"][k], prev), ctl.estimate()[0]
    g = load_golden(name)
    errs, compared, normal_bad = _full_run(g, lambda g: EmuLocomotion(g["robot_type"], g["gait_id"], horizon=_horizon(g), flat_ground=bool(g["flat_ground"])), step)
    print(f"{name}: compared {compared.mean():.4f} of the samples, max torque error {errs.max():.2e}")
    assert normal_bad == 0
    assert errs.max() < TAU_RTOL
    assert compared.mean() == 1.0                       # no estimator sample differs from the reference's any more (svml_acosf.h)


class _Live(dict):
    """a live run of the reference in the shape of a loaded golden"""
    files = property(lambda self: list(self.keys()))


@pytest.mark.reference
@pytest.mark.parametrize("horizon,seed", [(10, 101), (16, 102), (20, 103)])
def test_emulated_full_run_matches_the_live_reference(horizon, seed):
    """Volume beyond the committed fixtures, where the reference tree is mounted (the build container): the reference Python itself is run
    here -- 84 robots (seven gaits x three robot types x four), 24 ticks, sloped-ground estimate in the loop, horizonLength patched for
    h = 16 / 20 exactly as for the goldens (tests/golden/make_golden_controller.py) -- and the host emulation of controller.run is held to it
    like to a golden: ground normal and estimator outputs bit-identical on every tick, torques inside TAU_RTOL."""
    import os
    import sys
    if not os.path.isdir("/root/reference/MPC_Controller"):
        pytest.skip("reference not mounted")
    from tests.emu.emu import EmuLocomotion, estimator_update
    argv, sys.argv = sys.argv, ["make_golden_controller"]
    try:
        from tests.golden import make_golden_controller as M
        g = _Live(M.run_case(None, 84, 24, seed, False, horizon=horizon, gait_cycle=(0, 1, 2, 3, 5, 6, 7)))
    finally:
        sys.argv = argv
    g = _Live({k: np.asarray(v) for k, v in g.items()})

    def step(ctl, g, k):
        prev = g["normal"][k - 1] if k else np.tile(np.array([0, 0, 1], np.float32), (g["body"].shape[1], 1))
        tau = ctl.run(g["dof"][k], g["body"][k], g["cmd"][k])
        return tau, estimator_update(g["body"][k], prev), ctl.estimate()[0]
    errs, compared, normal_bad = _full_run(g, lambda g: EmuLocomotion(g["robot_type"], g["gait_id"], horizon=horizon, flat_ground=False), step)
    print(f"live reference, h = {horizon}: compared {compared.mean():.4f} of {compared.size} samples, max torque error {errs.max():.2e}")
    assert normal_bad == 0
    assert errs.max() < TAU_RTOL
    assert compared.mean() == 1.0


def test_gait_tables_match_reference_definition():
    """gait.py's tables against ConvexMPCLocomotion.py:30-56 (restated literally here)."""
    from rl_mpc_locomotion_amd.gait import GAIT_TABLE_10, gait_arrays, mpc_table
    ref = {0: ([0, 5, 5, 0], [5] * 4), 1: ([5, 5, 0, 0], [4] * 4), 2: ([0] * 4, [4] * 4), 3: ([5, 0, 5, 0], [5] * 4),
           5: ([0, 2, 7, 9], [4] * 4), 6: ([0, 3, 5, 8], [5] * 4), 7: ([0, 5, 5, 0], [4] * 4)}
    assert GAIT_TABLE_10 == ref
    off, dur = gait_arrays(16)
    assert off[0].tolist() == [0, 8, 8, 0] and dur[0].tolist() == [8] * 4
    t = mpc_table([0], [0], 2, 10).reshape(10, 4)
    assert t[:, 0].tolist() == [1, 1, 1, 1, 0, 0, 0, 0, 0, 1]                    # trot, leg FL, counter 0


@pytest.mark.reference
def test_gait_and_fk_match_reference_modules():
    import os
    import sys
    if not os.path.isdir("/root/reference/MPC_Controller"):
        pytest.skip("reference not mounted")
    sys.path.insert(0, "/root/reference")
    from MPC_Controller.convex_MPC.Gait import OffsetDurationGait
    from MPC_Controller.common.Quadruped import Quadruped, RobotType as RefType
    from MPC_Controller.common.LegController import LegController
    from rl_mpc_locomotion_amd.gait import GAIT_TABLE_10, mpc_table
    from rl_mpc_locomotion_amd.synthetic import leg_fk, leg_jacobian
    for gid, (o, d) in GAIT_TABLE_10.items():
        g = OffsetDurationGait(10, np.array(o, dtype=np.float32), np.array(d, dtype=np.float32), "x")
        for it in range(0, 23):
            g.setIterations(2, it)
            assert np.array_equal(np.array(g.getMpcTable(), dtype=np.float32), mpc_table([gid], [it], 2, 10)[0])
    rng = np.random.default_rng(0)
    for ours, ref in ((0, RefType.ALIENGO), (1, RefType.A1), (2, RefType.GO1)):
        lc = LegController(Quadruped(ref))
        q = rng.uniform(-1, 1, (4, 3))
        for leg in range(4):
            lc.datas[leg].q[:, 0] = q[leg]
            lc.computeLegJacobianAndPosition(leg)
            assert np.array_equal(lc.datas[leg].p[:, 0], leg_fk(q[None].astype(np.float32), [ours])[0, leg])
            assert np.allclose(lc.datas[leg].J, leg_jacobian(q[None].astype(np.float32), [ours])[0, leg], atol=1e-6)      # (bench.py's torque-map error figure)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDENS)
def test_hip_controller_matches_reference_python(name):
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g = load_golden(name)
    T, n = g["dof"].shape[:2]
    ctl = BatchedLocomotion(g["robot_type"], g["gait_id"], horizon=_horizon(g), flat_ground=bool(g["flat_ground"]), device="cuda:0")
    worst = 0.0
    for k in range(T):
        tau = ctl.step(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["est"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda())
        torch.cuda.synchronize()
        worst = max(worst, float(_relerr(tau.cpu().numpy(), g["torque"][k]).max()))
        if (k + 1) % 2 == 0:
            assert (ctl.solver_info()[:, 1] == 1).all()
            if "record" in g.files:      # every argument of the tick's compute_contact_forces calls and OSQP's decisions: the reference's
                assert np.array_equal(ctl.solver_record(), g["record"][k])
                assert np.array_equal(ctl.solver_info()[:, :4], g["decisions"][k])
    print(f"{name}: max torque error {worst:.2e}")
    assert worst < TAU_RTOL


def test_estimator_matmul_rule_matches_numpy():
    """The rule estimator_update uses for rBody @ vWorld (csrc/controller.h) against numpy itself on random data: rows 0 and 1
    plain float32 products and sums, row 2 fma(a2, b2, fma(a0, b0, a1 * b1)) -- what OpenBLAS' sgemv does with a 3 x 3 matrix on
    this class of CPU.  (If a numpy / OpenBLAS build ever rounds differently this test says so before the torque tests do.)"""
    rng = np.random.default_rng(0)
    f32 = np.float32
    fma = lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
    for _ in range(200):
        A16 = rng.uniform(-1, 1, (3, 3)).astype(np.float16)
        x = rng.uniform(-2, 2, (3, 1)).astype(f32)
        rB = np.array(A16.tolist(), dtype=np.float16).T            # like orientation_tools.quat_to_rot
        got = (rB @ x).flatten()
        a = rB.astype(f32); b = x[:, 0]
        want = np.array([(a[0, 0] * b[0] + a[0, 1] * b[1]) + a[0, 2] * b[2], (a[1, 0] * b[0] + a[1, 1] * b[1]) + a[1, 2] * b[2],
                         fma(a[2, 2], b[2], fma(a[2, 0], b[0], a[2, 1] * b[1]))], dtype=f32)
        assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDENS)
def test_hip_full_run_matches_reference_python(name):
    """The complete controller.run seam (estimator + ground-normal fit + controller + solve) on the GPU against the reference's torques,
    horizons 10, 16 and 20, all seven gaits: the ground normal bit-identical on every tick (gelsd43.h on the device), every compared
    (tick, robot) sample inside TAU_RTOL, every estimator sample bit-identical (compared fraction asserted to be 1.0)."""
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion

    def step(ctl, g, k):
        tau = ctl.run(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda())
        est, nrm = ctl.estimate()
        return tau.cpu().numpy(), est.cpu().numpy(), nrm.cpu().numpy()
    g = load_golden(name)
    errs, compared, normal_bad = _full_run(g, lambda g: BatchedLocomotion(g["robot_type"], g["gait_id"], horizon=_horizon(g), flat_ground=bool(g["flat_ground"]), device="cuda:0"), step)
    print(f"{name}: compared {compared.mean():.4f} of the samples, max torque error {errs.max():.2e}")
    assert normal_bad == 0
    assert errs.max() < TAU_RTOL, (float((errs < TAU_RTOL).mean()), float(errs.max()))
    assert compared.mean() == 1.0


@pytest.mark.gpu
def test_hip_full_run_as_shipped_matches_reference_python():
    """The reference AS SHIPPED passes mpc.QPOASES (ConvexMPCLocomotion.py:108): BatchedLocomotion(solver="exact") through the whole
    controller.run seam against the torques of the unmodified reference Python with the exact optimum behind its mpc_osqp seam
    (shim_calls_config1.npz: torque_exact; BASELINE configs[0], 1000 ticks)."""
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g = load_golden("controller_h10_config1")
    want = load_golden("shim_calls_config1")["torque_exact"]
    T = min(len(want), g["dof"].shape[0])
    ctl = BatchedLocomotion(g["robot_type"], g["gait_id"], horizon=10, flat_ground=bool(g["flat_ground"]), device="cuda:0", solver="exact")
    errs = []
    for k in range(T):
        tau = ctl.run(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda())
        errs.append(_relerr(tau.cpu().numpy(), want[k][None]))
    errs = np.concatenate(errs)
    assert errs.max() < TAU_RTOL, (float((errs < TAU_RTOL).mean()), float(errs.max()))


# ---- BASELINE configs[2] as SURVEY 8(d) writes it: the gait changes DURING the run ----------------------------------------------------
# controller_h10_cycling: 27 robots (3 types), Parameters.cmpc_gait cycling TROT -> WALK -> BOUND every 50 ticks (the last nine robots every 25: switches on
# ticks with and without an MPC update), 124 ticks, minted from the unmodified reference (ConvexMPCLocomotion.py:224-244 re-reads the parameter on every tick).
def _cycling_run(g, ctl, run, set_gait):
    """controller.run over the golden with the gait schedule applied before each tick; returns what test_*_full_run compare, plus per-tick records / decisions."""
    T, n = g["dof"].shape[:2]
    sched = g["gait_sched"]
    ok = np.ones(n, bool)
    errs, compared, normal_bad, rec_bad, dec_bad, solves = [], np.zeros((T, n), bool), 0, 0, 0, 0
    for k in range(T):
        if k == 0 or (sched[k] != sched[k - 1]).any():
            set_gait(ctl, sched[k])
        tau, est, nrm, rec, dec = run(ctl, g, k)
        ok &= (est == g["est"][k]).all(-1)
        compared[k] = ok
        errs.append(_relerr(tau, g["torque"][k])[ok])
        normal_bad += int((nrm != g["normal"][k]).any(-1).sum())
        due = g["solved"][k].astype(bool) & ok
        if due.any():
            solves += int(due.sum())
            rec_bad += int((rec[due] != g["record"][k][due]).any(-1).sum())
            dec_bad += int((dec[due, :4] != g["decisions"][k][due]).any(-1).sum())
    return np.concatenate(errs), compared, normal_bad, rec_bad, dec_bad, solves


def _check_cycling(name, out):
    errs, compared, normal_bad, rec_bad, dec_bad, solves = out
    print(f"{name}: compared {compared.mean():.4f} of the samples, {solves} solves, max torque error {errs.max():.2e}")
    assert normal_bad == 0
    assert rec_bad == 0          # every argument of every compute_contact_forces call bit-identical ACROSS the switches (contact tables of the new gait included)
    assert dec_bad == 0          # ... and OSQP's decisions on every one of them
    assert errs.max() < TAU_RTOL
    assert compared.mean() == 1.0
    assert compared[101:].all()                                     # both switches (ticks 50 and 100) lie inside the compared stretch


def test_emulated_gait_cycling_matches_reference_python():
    from tests.emu.emu import EmuLocomotion, estimator_update
    g = load_golden("controller_h10_cycling")
    assert len(np.unique(g["gait_sched"][[0, 50, 100], 0])) == 3 and (g["gait_sched"][24] != g["gait_sched"][25]).any()      # TROT / WALK / BOUND; a switch before an odd tick

    def run(ctl, g, k):
        prev = g["normal"][k - 1] if k else np.tile(np.array([0, 0, 1], np.float32), (g["body"].shape[1], 1))
        tau = ctl.run(g["dof"][k], g["body"][k], g["cmd"][k])
        return tau, estimator_update(g["body"][k], prev), ctl.estimate()[0], ctl.solver_record(), ctl.solver_info()
    ctl = EmuLocomotion(g["robot_type"], g["gait_id"], horizon=10, flat_ground=False)
    _check_cycling("controller_h10_cycling (emulation)", _cycling_run(g, ctl, run, lambda c, gi: c.set_gait(gi)))


@pytest.mark.gpu
@pytest.mark.parametrize("device_ids", [True, False])
def test_hip_gait_cycling_matches_reference_python(device_ids):
    """BASELINE configs[2] through controller.run on the GPU: the gait ids arrive as a DEVICE tensor (mpc_ctrl_set_gait_device, stream-ordered, no
    synchronisation) or as a host array (mpc_ctrl_set_gait)."""
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g = load_golden("controller_h10_cycling")

    def run(ctl, g, k):
        tau = ctl.run(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda())
        est, nrm = ctl.estimate()
        return tau.cpu().numpy(), est.cpu().numpy(), nrm.cpu().numpy(), ctl.solver_record(), ctl.solver_info()
    ctl = BatchedLocomotion(g["robot_type"], g["gait_id"], horizon=10, flat_ground=False, device="cuda:0")
    set_gait = (lambda c, gi: c.set_gait(torch.from_numpy(np.ascontiguousarray(gi)).cuda())) if device_ids else (lambda c, gi: c.set_gait(gi))
    _check_cycling("controller_h10_cycling (hip)", _cycling_run(g, ctl, run, set_gait))


@pytest.mark.gpu
def test_hip_controller_reset_and_gait_switch():
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g = load_golden("controller_h10_flat")
    T, n = g["dof"].shape[:2]
    a = BatchedLocomotion(g["robot_type"], g["gait_id"], flat_ground=True, device="cuda:0")
    run = lambda ctl, k: ctl.step(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["est"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda()).cpu().numpy().copy()
    first = [run(a, k) for k in range(6)]
    a.reset(np.array([0, 2], dtype=np.int32))          # robots 0 and 2 start over, the others continue
    again = [run(a, k) for k in range(6)]
    b = BatchedLocomotion(g["robot_type"], g["gait_id"], flat_ground=True, device="cuda:0")
    cont = [run(b, k) for k in range(6)] + [run(b, k) for k in range(6)]
    for k in range(6):
        # reset robots: same as a fresh controller except for the carried f_ff / last swing p, v (reference semantics),
        # which only matter before the first solve / first swing -> compare from the first MPC tick on
        if k >= 1:
            np.testing.assert_allclose(again[k][[0, 2]], first[k][[0, 2]], rtol=0, atol=2e-3 * np.abs(first[k]).max())
        np.testing.assert_array_equal(again[k][[1, 3, 4, 5]], cont[6 + k][[1, 3, 4, 5]])
    a.set_gait(np.full(n, 1, dtype=np.int32))
    out = run(a, 6)
    assert np.isfinite(out).all()      # (parity of a gait switch against the reference: test_hip_gait_cycling_matches_reference_python)
    with pytest.raises(ValueError):
        a.set_gait(torch.zeros(n, dtype=torch.int64, device="cuda:0"))


@pytest.mark.gpu
def test_env_bridge_equals_manual_composition():
    """MpcEnvBridge.pre_physics_step (one fused rescale + pack kernel, then controller.run) against the composition it replaced -- torch.mul(...).add(...),
    three slice copies into the command record, BatchedLocomotion.run (RL_Environment/tasks/aliengo.py:237-258 statement by statement): torques bit-identical
    on every tick, device-side reset_idx included."""
    import torch
    from rl_mpc_locomotion_amd.env_bridge import MpcEnvBridge
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.synthetic import TickStream
    from rl_mpc_locomotion_amd.weight_policy import MPC_PARAM_CONST, MPC_PARAM_SCALE
    n = 96
    ts = TickStream(n, seed=31, config=3)
    br = MpcEnvBridge(ts.robot_type, ts.gait_id, horizon=10, flat_ground=False)
    ctl = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=10, flat_ground=False)
    scale = torch.tensor(MPC_PARAM_SCALE, dtype=torch.float, device="cuda"); const = torch.tensor(MPC_PARAM_CONST, dtype=torch.float, device="cuda")
    rng = np.random.default_rng(3)
    cmd16 = torch.zeros((n, 16), dtype=torch.float32, device="cuda")
    for k in range(14):
        dof, body, cmd = ts.tick(k)
        actions = torch.from_numpy(rng.uniform(-1, 1, (n, 12)).astype(np.float32)).cuda()
        commands = torch.from_numpy(np.ascontiguousarray(cmd[:, :3])).cuda()
        dof_t, body_t = torch.from_numpy(dof.reshape(n * 12, 2)).cuda(), torch.from_numpy(body).cuda()
        a = br.pre_physics_step(actions, dof_t, body_t, commands).clone()
        actions_rescale = torch.mul(actions, scale).add(const)
        cmd16[:, 0:3] = commands; cmd16[:, 3:15] = actions_rescale; cmd16[:, 15] = 0.0
        b = ctl.run(dof_t.reshape(n, 12, 2).contiguous(), body_t, cmd16)
        assert torch.equal(a, b), k
        assert torch.equal(br._cmd, cmd16)
        if k == 6:
            ids = torch.tensor([3, 17, 40], dtype=torch.int32, device="cuda")
            br.reset_idx(ids); ctl.reset(ids)


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["aliengo", "a1", "go1"])
def test_env_bridge_matches_reference_glue(task):
    """MpcEnvBridge against the reference's own pre_physics_step / reset_idx of all three task files (RL_Environment/tasks/aliengo.py,
    a1.py, go1.py :227-263, :321-349 -- the same glue around another RobotType, :201), executed unmodified by
    tests/golden/make_golden_bridge.py: rescaled actions, command record, controller.run for every env, device env_ids in reset_idx."""
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.env_bridge import MpcEnvBridge
    g = load_golden("bridge_h10_" + task)
    T, n = g["actions"].shape[:2]
    assert (g["robot_type"] == {"aliengo": 0, "a1": 1, "go1": 2}[task]).all()
    br = MpcEnvBridge(g["robot_type"], np.zeros(n, np.int32), horizon=10, flat_ground=False)              # four robots of the task's type, trot
    agree = np.ones(n, bool)          # the robot's solver decisions have equalled the reference's on every solve so far
    errs, counted = [], 0
    for k in range(T):
        if k == int(g["reset_at"]):
            br.reset_idx(torch.tensor(g["reset_ids"], dtype=torch.long, device="cuda"))                 # env_ids stay on the device
            agree[g["reset_ids"]] = True                                                                  # a fresh ConvexMpc: cold start on both sides
        tau = br.pre_physics_step(torch.from_numpy(g["actions"][k]).cuda(), torch.from_numpy(g["dof_state"][k]).cuda(),
                                  torch.from_numpy(g["root_states"][k]).cuda(), torch.from_numpy(g["commands"][k]).cuda())
        dec = g["decisions"][k]
        solved = dec[:, 0] > 0
        info = br.ctl.solver_info()[:, :4]
        agree &= ~solved | (info == dec).all(axis=1)
        e = _relerr(tau.cpu().numpy(), g["torques"][k])
        errs.append(np.where(agree, e, 0.0))
        counted += int(agree.sum())
    # The torques hinge on OSQP's discrete decisions (a polish accepted there and rejected here moves the forces by 1e-2), and those on every
    # bit of the solver's arguments.  With the ground-normal fit walked in LAPACK's own arithmetic (gelsd43.h) the arguments are the
    # reference's: observed on all three tasks, every robot's decisions equal the reference's on every solve (agree fraction 1.0) and the
    # torques are within 3.2e-7.  With the estimator's float32 arccos / arctan2 restated too (svml_acosf.h) nothing is left that could make a robot take another
    # decision: the bookkeeping must find every sample compared.
    frac = counted / (T * n)
    print(f"bridge_{task}: decisions agree on {frac:.4f} of the (tick, robot) samples, max torque error {np.max(errs):.2e}")
    assert np.max(errs) < TAU_RTOL, np.max(errs)
    assert frac == 1.0, (counted, T * n)


@pytest.mark.gpu
def test_env_bridge_equals_manual_composition():
    """MpcEnvBridge.pre_physics_step = rescale + command record + controller.run (aliengo.py:237-258), reset_idx = per-robot reset."""
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.env_bridge import MpcEnvBridge
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.synthetic import TickStream
    n = 96
    ts = TickStream(n, seed=9, config=3)
    br = MpcEnvBridge(ts.robot_type, ts.gait_id, horizon=10)
    ref = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=10)
    rng = np.random.default_rng(4)
    scale = np.array([4, 4, 4, 20, 20, 20, 1, 1, 1, 1, 1, 1], np.float32); const = np.array([5, 5, 5, 50, 50, 50, 1, 1, 1, 1, 1, 1], np.float32)
    for k in range(6):
        dof, body, cmd16 = ts.tick(k)
        act = rng.uniform(-1, 1, (n, 12)).astype(np.float32)
        cmd = cmd16.copy(); cmd[:, 3:15] = act * scale + const; cmd[:, 15] = 0
        t_ref = ref.run(torch.from_numpy(dof).cuda(), torch.from_numpy(body).cuda(), torch.from_numpy(cmd).cuda()).clone()
        t_br = br.pre_physics_step(torch.from_numpy(act).cuda(), torch.from_numpy(dof.reshape(n * 12, 2)).cuda(), torch.from_numpy(body).cuda(),
                                   torch.from_numpy(np.ascontiguousarray(cmd16[:, :3])).cuda())
        assert torch.equal(t_ref, t_br)
        if k == 2:
            ids = torch.tensor([1, 5, 40], dtype=torch.int64, device="cuda")
            br.reset_idx(ids); ref.reset(ids)


@pytest.mark.gpu
def test_long_run_with_random_resets_stays_healthy():
    """Soak: 2048 robots (3 robot types x 3 gaits), 600 ticks of controller.run with random per-robot resets every 37 ticks:
    every torque finite and bounded, every MPC solve of every robot reported OSQP_SOLVED."""
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.synthetic import TickStream
    n = 2048
    ts = TickStream(n, seed=77, config=3)
    ctl = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=10)
    rng = np.random.default_rng(5)
    worst = 0.0
    for k in range(600):
        dof, body, cmd = (torch.from_numpy(a).cuda() for a in ts.tick(k))
        tau = ctl.run(dof, body, cmd)
        if k % 37 == 36:
            ctl.reset(torch.from_numpy(rng.choice(n, size=64, replace=False).astype(np.int64)).cuda())
        if k % 50 == 49:
            assert torch.isfinite(tau).all()
            worst = max(worst, float(tau.abs().max()))
            assert (ctl.solver_info()[:, 1] == 1).all()
    assert worst < 1e4


@pytest.mark.gpu
def test_controller_solver_accessors_and_iteration_counter():
    """mpc_ctrl_set_iteration / mpc_ctrl_solver / mpc_ctrl_solver_forces / mpc_device_clock (the accessors bench.py's `value` leg uses): with
    controller_dt = 0.02 every robot is due on every run (iterationsBetweenMPC = 1); the counter sets the gait phase; the solver's forces and record are
    those of the launch; the solver handle's kernel timing works through the controller."""
    import torch
    from rl_mpc_locomotion_amd import _lib
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.synthetic import ControlStepStream
    n, h = 64, 10
    cs = ControlStepStream(n, h=h, seed=3, config=3)
    ctl = BatchedLocomotion(cs.robot_type, cs.gait_id, horizon=h, controller_dt=0.02, device="cuda:0")
    assert ctl.iterations_between_mpc == 1
    ctl.enable_timing()
    ctl.set_iteration(cs.iteration0)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for s in range(3):
        tau = ctl.run(*(up(a) for a in cs.step(s)))
    torch.cuda.synchronize()
    info, rec, f = ctl.solver_info(), ctl.solver_record(), ctl.solver_forces()
    assert (info[:, 1] == 1).all() and np.isfinite(tau.cpu().numpy()).all()
    # the contact table of the record is the gait table at the counter the robots were given (+ the three runs)
    from rl_mpc_locomotion_amd.gait import mpc_table
    from rl_mpc_locomotion_amd import layout as L
    tables = [mpc_table(cs.gait_id, cs.iteration0 + k, 1, h) for k in (2, 3)]      # (the counter is read before / after its increment: ConvexMPCLocomotion.py:217-231)
    got = rec[:, L.IN_CONTACT:L.IN_CONTACT + 4 * h]
    assert any(np.array_equal(got, w) for w in tables)
    want = tables[0] if np.array_equal(got, tables[0]) else tables[1]
    swing = np.repeat(want == 0, 3, axis=1)
    assert np.abs(f[swing]).max() < 1e-2 * np.abs(f).max() and (np.abs(f).max(1) > 1.0).all()      # forces of the last solve: zero to ADMM accuracy (eps 1e-3) on swing feet
    prep, solve = ctl.kernel_times(3)
    assert (prep > 0).all() and (solve > 0).all()
    ghz, ms = _lib.device_clock(0, 5)
    assert 0.5 < ghz < 3.5 and ms > 1.0
    with pytest.raises(_lib.MpcLibraryError):
        _lib.check(_lib.lib().mpc_ctrl_set_iteration(ctl._handle, np.full(n, -1, np.int32).ctypes.data, None), "mpc_ctrl_set_iteration")


@pytest.mark.gpu
@pytest.mark.parametrize("h,n", [(10, 4096), (16, 4096), (20, 8192)])
def test_controller_run_full_size_properties(h, n):
    """controller.run at BASELINE's per-GPU sizes (configs[1] / [3] / [4]: 4096 x h = 10, 4096 x h = 16, 8192 x h = 20), through properties that need
    no oracle: (a) a robot's torques do not depend on the batch around it -- the 64 robots whose inputs are the golden-sized batch's come out bit-identical
    from the full batch and from a batch of 64; (b) duplicated robots (second half = first half) give bit-identical halves on every tick, a per-robot reset
    of one copy included; (c) every MPC solve of every robot reports OSQP_SOLVED and the torques are finite."""
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.synthetic import TickStream
    half = n // 2
    ts = TickStream(half, seed=31 + h, config=3)
    rt = np.concatenate([ts.robot_type, ts.robot_type]); gi = np.concatenate([ts.gait_id, ts.gait_id])
    big = BatchedLocomotion(rt, gi, horizon=h, device="cuda:0")
    small = BatchedLocomotion(ts.robot_type[:64], ts.gait_id[:64], horizon=h, device="cuda:0")
    ids = np.array([3, 17, 40], dtype=np.int32)
    for k in range(7):
        dof, body, cmd = ts.tick(k)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        if k == 4:      # both copies of three robots start over, in all three controllers
            big.reset(torch.from_numpy(np.concatenate([ids, ids + half])).cuda()); small.reset(ids)
        tau = big.run(up(np.concatenate([dof, dof])), up(np.concatenate([body, body])), up(np.concatenate([cmd, cmd]))).clone()
        ref = small.run(up(dof[:64]), up(body[:64]), up(cmd[:64]))
        assert torch.equal(tau[:half], tau[half:]), k
        assert torch.equal(tau[:64], ref), k
        assert torch.isfinite(tau).all()
        if (k + 1) % 2 == 0:
            assert (big.solver_info()[:, 1] == 1).all()
    assert float(tau.abs().max()) > 1.0
