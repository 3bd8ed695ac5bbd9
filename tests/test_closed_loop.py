"""Closed-loop rollouts (8(d) config 1 "closed-loop on a toy integrator"): this repository's controller and the
unmodified reference Python each drive their OWN copy of a toy rigid-body simulator (tests/toy_sim.py) from the same initial state -- the
torques a controller returns decide what it sees next, so a force difference is fed back instead of being replayed away.

Golden: tests/golden/closed_loop_h10.npz (make_golden_closed_loop.py: RobotRunnerMin + the oracle's OSQP behind its seam, up to 1000
ticks; trot on flat ground for the three robot types, trot on a slope with the ground-normal estimate in the loop, walk, bound).  What
"stays together" can mean is bounded by the REFERENCE ITSELF: OSQP at eps 1e-3 takes discrete decisions (iterations in steps of 25, rho
updates, polish acceptance), so a 1e-6 relative change of its inputs -- below float32 resolution -- already moves the reference's own closed
loop.  The golden therefore also holds three runs of the reference with that noise on body_states (`pert_*`):
  * trot, flat (3 robot types): the perturbed reference keeps the contact schedule for all 1000 ticks and stays within 1.4e-5 m / 2.1e-5 m;
    measured here: identical contact flags on every tick, |dpos| <= 7e-6 m (emulation and GPU) -- asserted: identical flags, <= 1e-4 m;
  * trot on the slope / walk: the perturbed reference leaves the golden's contact schedule at tick 81 / 56-110 (a 200-iteration solve with
    four rho updates at tick 65 amplifies 1e-5 to 1 N m) and falls in the toy in one of three runs; this controller leaves it at tick 81 / 68.
    Asserted: not before the perturbed reference does (10 ticks of slack), and up to there no further away than 5 x the perturbed
    reference's own distance (floor 1e-4 m);
  * bound: 16 ticks (the reference itself then sinks in this toy: with two hind legs in stance its MPC asks for the minimum force).
"""
import os

import numpy as np
import pytest

import rl_mpc_locomotion_amd  # noqa: F401
from rl_mpc_locomotion_amd.quadruped import ROBOT_TABLE64
from tests.helpers import ROOT
from tests.toy_sim import ToyRobot

GOLD = os.path.join(ROOT, "tests", "golden", "closed_loop_h10.npz")


def _cases():
    g = np.load(GOLD)
    return g, sorted({k.split("/")[0] for k in g.files})


def _rollout(g, names, make_ctrl):
    """One controller handle for `names` (same flat_ground), each robot in its own simulator copy; returns per case (dpos [T], first tick at
    which the contact flags differ from the golden's or the robot has fallen (T if never))."""
    meta = [g[n + "/meta"] for n in names]
    rt = [int(m[0]) for m in meta]
    ctrl = make_ctrl(rt, [int(m[1]) for m in meta], bool(meta[0][2]))
    toys = [ToyRobot(ROBOT_TABLE64[r], yaw0=m[5], slope=(m[3], m[4])) for r, m in zip(rt, meta)]
    ticks = [int(m[6]) for m in meta]
    cmd = np.stack([g[n + "/cmd"] for n in names])
    dpos = [np.full(t, np.nan) for t in ticks]
    first = list(ticks)
    last = [np.zeros(12, np.float32) for _ in names]
    for k in range(max(ticks)):
        obs = [t.observe() for t in toys]
        for i, n in enumerate(names):
            if k >= ticks[i] or toys[i].fell:
                first[i] = min(first[i], k)
                continue
            dpos[i][k] = np.abs(obs[i][1][:3] - g[n + "/body"][k, :3]).max()
            if (toys[i].contact != g[n + "/contact"][k]).any():
                first[i] = min(first[i], k)
        body = np.nan_to_num(np.stack([o[1] for o in obs]), nan=0.0, posinf=0.0, neginf=0.0)    # (a fallen toy may hold non-finite numbers)
        tau = ctrl(np.stack([o[0] for o in obs]), body, cmd)
        for i in range(len(names)):
            if k < ticks[i] and not toys[i].fell:
                toys[i].step(tau[i])
    return {n: (dpos[i], first[i]) for i, n in enumerate(names)}


def _check(g, name, dpos, first):
    ticks = int(g[name + "/meta"][6])
    pd, pc = g[name + "/pert_dpos"], g[name + "/pert_contact"]
    ref_first = []
    for s in range(pd.shape[0]):
        bad = (pc[s] != g[name + "/contact"]).any(1) | np.isnan(pd[s])
        ref_first.append(int(np.argmax(bad)) if bad.any() else ticks)
    stable = min(ref_first) == ticks
    if stable:
        assert first == ticks, f"{name}: contact schedule left the reference's at tick {first} (the perturbed reference never does)"
        assert np.nanmax(dpos) <= 1e-4, f"{name}: |dpos| {np.nanmax(dpos):.2e} m"
        return
    assert first >= min(ref_first) - 10, f"{name}: left the reference's contact schedule at tick {first}, the perturbed reference at {ref_first}"
    w = min(first, min(ref_first))
    ours = np.maximum.accumulate(np.nan_to_num(dpos[:w]))
    ref = np.maximum.accumulate(np.nan_to_num(pd[:, :w]), axis=1).max(0)
    worst = int(np.argmax(ours - np.maximum(1e-4, 5 * ref)))
    assert (ours <= np.maximum(1e-4, 5 * ref)).all(), f"{name}: tick {worst}: |dpos| {ours[worst]:.2e} m vs the perturbed reference's {ref[worst]:.2e}"


def _groups(names, g):
    flat = [n for n in names if g[n + "/meta"][2] == 1]
    return [grp for grp in (flat, [n for n in names if n not in flat]) if grp]


def test_emulated_controller_closed_loop_tracks_the_reference():
    from tests.emu import emu
    g, names = _cases()
    for grp in _groups(names, g):
        def make(rt, gait, flat):
            c = emu.EmuLocomotion(rt, gait, flat_ground=flat, nthreads=len(rt))
            return lambda dof, body, cmd: c.run(dof, body, cmd)
        for n, (dpos, first) in _rollout(g, grp, make).items():
            _check(g, n, dpos, first)


@pytest.mark.gpu
def test_hip_controller_closed_loop_tracks_the_reference():
    import torch
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    g, names = _cases()
    for grp in _groups(names, g):
        def make(rt, gait, flat):
            c = BatchedLocomotion(rt, gait, horizon=10, flat_ground=flat, device="cuda:0")
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda:0")
            return lambda dof, body, cmd: c.run(up(dof), up(body), up(cmd)).cpu().numpy().copy()
        for n, (dpos, first) in _rollout(g, grp, make).items():
            _check(g, n, dpos, first)
