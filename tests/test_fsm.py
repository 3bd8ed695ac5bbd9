"""Control FSM: the batched RobotRunnerFSM against tests/golden/fsm_h10.npz, which was minted by
the UNMODIFIED reference Python (tests/golden/make_golden_fsm.py): Passive / RecoveryStand (StandUp, FoldLegs, RollOver) /
Locomotion, commanded and safety-triggered transitions, a refused PASSIVE -> LOCOMOTION request.
CPU: host emulation of the same C++ (csrc/controller.h).  GPU: through the C ABI."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fsm_h10.npz")
TAU_RTOL = 5e-5          # locomotion ticks (float32 controller around the fp64 solve); joint-PD ticks are bit-exact


def _check(g, tau, fsm):
    np.testing.assert_array_equal(fsm[:, :, 0], g["state"])
    np.testing.assert_array_equal(fsm[:, :, 1], g["op_mode"])
    np.testing.assert_array_equal(fsm[:, :, 2], g["rs_flag"])
    ref = g["torque"]
    loco = (g["state"] == 4)
    # ticks on which no locomotion step ran (other states, or the two transition ticks): pure float32 joint PD -> exact
    prev_loco = np.vstack([loco[:1], loco[:-1]])
    pd = ~loco & ~prev_loco
    np.testing.assert_array_equal(tau[pd], ref[pd])
    scale = np.maximum(np.abs(ref).max(axis=2, keepdims=True), 1.0)
    assert (np.abs(tau - ref) / scale).max() < TAU_RTOL
    assert set(np.unique(g["state"])) == {0, 4, 6} and set(np.unique(g["rs_flag"])) == {0, 1, 2}     # the fixture covers everything


def test_fsm_emulation_matches_reference():
    from tests.emu.emu import fsm_replay
    g = np.load(GOLD)
    n = g["dof"].shape[1]
    tau, fsm = fsm_replay(g["robot_type"], np.zeros(n, np.int32), g["init_mode"], g["dof"], g["body"], g["cmd"], g["request"])
    _check(g, tau, fsm)


@pytest.mark.gpu
def test_fsm_gpu_matches_reference():
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g = np.load(GOLD)
    T, n = g["dof"].shape[0], g["dof"].shape[1]
    ctl = BatchedLocomotion(g["robot_type"], np.zeros(n, np.int32), horizon=10)
    ctl.fsm_init(g["init_mode"], operating_mode=1, check_safety=True)
    tau = np.zeros((T, n, 12), np.float32); fsm = np.zeros((T, n, 3), np.int32)
    for k in range(T):
        t = ctl.run_fsm(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda(),
                        torch.from_numpy(g["request"][k]).cuda())
        tau[k] = t.cpu().numpy()
        fsm[k] = ctl.fsm_state()[:, :3]
    _check(g, tau, fsm)
    assert ctl.fsm_state()[:, 3].tolist() == [1, 0, 0, 0, 1, 0]          # the two robots with a roll excursion were flagged unsafe


@pytest.mark.gpu
def test_fsm_reset_and_bad_arguments():
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd import _lib
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g = np.load(GOLD)
    n = g["dof"].shape[1]
    ctl = BatchedLocomotion(g["robot_type"], np.zeros(n, np.int32), horizon=10)
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device="cuda")
    with pytest.raises(_lib.MpcLibraryError):
        ctl.run_fsm(z(n, 12, 2), z(n, 13), z(n, 16), torch.zeros(n, dtype=torch.int32, device="cuda"))      # fsm_init not called
    with pytest.raises(_lib.MpcLibraryError):
        ctl.fsm_init(np.full(n, 5))                                                                       # not a state name
    ctl.fsm_init(np.full(n, BatchedLocomotion.RECOVERY_STAND))
    req = torch.full((n,), BatchedLocomotion.LOCOMOTION, dtype=torch.int32, device="cuda")
    for k in range(4):
        ctl.run_fsm(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda(), req)
    assert (ctl.fsm_state()[:, 0] == BatchedLocomotion.LOCOMOTION).all()
    ctl.fsm_reset(env_ids=[1, 3], control_mode=np.full(n, BatchedLocomotion.PASSIVE))
    st = ctl.fsm_state()
    assert st[[1, 3], 0].tolist() == [0, 0] and (st[[0, 2, 4, 5], 0] == 4).all()


@pytest.mark.gpu
def test_fsm_reset_into_locomotion_cold_starts_the_solver():
    """RobotRunnerFSM.reset -> ControlFSM.initialize -> FSM_State_Locomotion.onEnter -> cMPC.initialize builds a NEW ConvexMpc
    (FSM_State_Locomotion.py:32-42, ConvexMPCLocomotion.py:102-108): the reset robots' next solve is the cold "osqp_setup" solve
    (x = y = z = 0, rho = 0.1), the others stay warm."""
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g = np.load(GOLD)
    n = g["dof"].shape[1]
    ctl = BatchedLocomotion(g["robot_type"], np.zeros(n, np.int32), horizon=10)
    ctl.fsm_init(np.full(n, BatchedLocomotion.LOCOMOTION))
    req = torch.full((n,), BatchedLocomotion.LOCOMOTION, dtype=torch.int32, device="cuda")
    run = lambda k: ctl.run_fsm(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda(), req)
    for k in range(6):
        run(k)
    assert (ctl.solver_info()[:, 5] == 0).all()                 # everybody is warm by now
    ctl.fsm_reset(env_ids=[1, 4], control_mode=np.full(n, BatchedLocomotion.LOCOMOTION))
    for k in range(6, 8):                                        # the next MPC update of every robot (every 2nd tick)
        run(k)
    first = ctl.solver_info()[:, 5]
    assert first[[1, 4]].tolist() == [1, 1] and (first[[0, 2, 3, 5]] == 0).all()
    with pytest.raises(ValueError):
        ctl.fsm_reset(env_ids=[1], control_mode=[BatchedLocomotion.LOCOMOTION])       # one mode per ROBOT, not per id


@pytest.mark.gpu
def test_fsm_reset_with_device_ids_equals_host_ids():
    """mpc_ctrl_fsm_reset_device (an env_ids tensor on the GPU: stream-ordered, no host round trip) = mpc_ctrl_fsm_reset with the same ids
    on the host: two controllers on the same inputs, one reset each way, give the same torques, FSM states and solver records."""
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g = np.load(GOLD)
    n = g["dof"].shape[1]
    ctls = [BatchedLocomotion(g["robot_type"], np.zeros(n, np.int32), horizon=10) for _ in range(2)]
    req = torch.full((n,), BatchedLocomotion.LOCOMOTION, dtype=torch.int32, device="cuda")
    for c in ctls:
        c.fsm_init(np.full(n, BatchedLocomotion.LOCOMOTION))
    out = [[], []]
    for k in range(14):
        if k == 7:
            ctls[0].fsm_reset(env_ids=[0, 3, 5])
            ctls[1].fsm_reset(env_ids=torch.tensor([0, 3, 5], dtype=torch.int64, device="cuda"))
        for i, c in enumerate(ctls):
            t = c.run_fsm(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda(), req)
            out[i].append((t.cpu().numpy().copy(), c.fsm_state().copy(), c.solver_info().copy()))
    for a, b in zip(*out):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    firsts = np.stack([out[1][k][2][:, 5] for k in (7, 8, 9)])
    assert firsts[:, [0, 3, 5]].max(0).tolist() == [1, 1, 1] and firsts[:, [1, 2, 4]].max() == 0      # the reset robots' next MPC update was a cold first call
