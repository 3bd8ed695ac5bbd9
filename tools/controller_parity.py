"""Parity figures of the controller.run seam on the GPU against the goldens minted from the reference Python (tests/golden/controller_*.npz,
bridge_*.npz): per golden the number of (tick, robot) samples, how many estimator samples differ from the reference's bit for bit (none since round 6:
csrc/svml_acosf.h), on how many ticks the ground normal differs (expected: none), which fraction of the samples is compared and the
largest torque error among them; through the step seam (golden estimator outputs) whether every compute_contact_forces argument record and every
OSQP decision equals the reference's.  Prints one line: CONTROLLER_PARITY_JSON {...}   (copied to profiles/ by the caller)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import rl_mpc_locomotion_amd  # noqa: E402,F401
from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion  # noqa: E402
from rl_mpc_locomotion_amd.env_bridge import MpcEnvBridge  # noqa: E402
from tests.helpers import load_golden  # noqa: E402
from tests.test_controller import GOLDENS, _relerr, _horizon, _full_run, _cycling_run  # noqa: E402

out = {}
for name in GOLDENS:
    g = load_golden(name)
    T, n = g["dof"].shape[:2]
    h = _horizon(g)

    def step(ctl, g, k):
        tau = ctl.run(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda())
        est, nrm = ctl.estimate()
        return tau.cpu().numpy(), est.cpu().numpy(), nrm.cpu().numpy()
    est_bad = [0]

    def step_counting(ctl, g, k):
        tau, est, nrm = step(ctl, g, k)
        est_bad[0] += int((est != g["est"][k]).any(-1).sum())
        return tau, est, nrm
    errs, compared, normal_bad = _full_run(g, lambda g: BatchedLocomotion(g["robot_type"], g["gait_id"], horizon=h, flat_ground=bool(g["flat_ground"]), device="cuda:0"), step_counting)
    rec = dict(horizon=h, robots=n, ticks=T, samples=T * n, flat_ground=bool(g["flat_ground"]), gaits=sorted(set(int(x) for x in g["gait_id"])),
               estimator_samples_not_bit_identical=est_bad[0], ground_normal_samples_not_bit_identical=normal_bad,
               compared_fraction=float(compared.mean()), max_torque_rel_err_compared=float(errs.max()))
    # step seam: the reference's estimator outputs in, every solver argument and decision out
    ctl = BatchedLocomotion(g["robot_type"], g["gait_id"], horizon=h, flat_ground=bool(g["flat_ground"]), device="cuda:0")
    worst, rec_same, dec_same, solves = 0.0, 0, 0, 0
    for k in range(T):
        tau = ctl.step(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["est"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda())
        worst = max(worst, float(_relerr(tau.cpu().numpy(), g["torque"][k]).max()))
        if (k + 1) % 2 == 0 and "record" in g.files:
            solves += n
            rec_same += int((ctl.solver_record() == g["record"][k]).all(-1).sum())
            dec_same += int((ctl.solver_info()[:, :4] == g["decisions"][k]).all(-1).sum())
    rec["step_seam"] = dict(max_torque_rel_err=worst, solves=solves, argument_records_bit_identical=rec_same, osqp_decisions_equal=dec_same)
    out[name] = rec
    print(name, json.dumps(rec), flush=True)
# BASELINE configs[2] as stated: the gait changes DURING the run (controller_h10_cycling: TROT / WALK / BOUND every 50 ticks, set through a device tensor)
g = load_golden("controller_h10_cycling")


def run_c(ctl, g, k):
    tau = ctl.run(torch.from_numpy(g["dof"][k]).cuda(), torch.from_numpy(g["body"][k]).cuda(), torch.from_numpy(g["cmd"][k]).cuda())
    est, nrm = ctl.estimate()
    return tau.cpu().numpy(), est.cpu().numpy(), nrm.cpu().numpy(), ctl.solver_record(), ctl.solver_info()


ctl = BatchedLocomotion(g["robot_type"], g["gait_id"], horizon=10, flat_ground=False, device="cuda:0")
errs, compared, normal_bad, rec_bad, dec_bad, solves = _cycling_run(g, ctl, run_c, lambda c, gi: c.set_gait(torch.from_numpy(np.ascontiguousarray(gi)).cuda()))
out["controller_h10_cycling"] = dict(horizon=10, robots=int(g["dof"].shape[1]), ticks=int(g["dof"].shape[0]), samples=int(compared.size), gait_switches="every 50 ticks (the last nine robots: every 25)",
                                     compared_fraction=float(compared.mean()), ground_normal_samples_not_bit_identical=normal_bad, solves=solves,
                                     argument_records_not_bit_identical=rec_bad, osqp_decisions_not_equal=dec_bad, max_torque_rel_err_compared=float(errs.max()))
print("controller_h10_cycling", json.dumps(out["controller_h10_cycling"]), flush=True)
for task in ("aliengo", "a1", "go1"):
    g = load_golden("bridge_h10_" + task)
    T, n = g["actions"].shape[:2]
    br = MpcEnvBridge(g["robot_type"], np.zeros(n, np.int32), horizon=10, flat_ground=False)
    agree, counted, worst = np.ones(n, bool), 0, 0.0
    for k in range(T):
        if k == int(g["reset_at"]):
            br.reset_idx(torch.tensor(g["reset_ids"], dtype=torch.long, device="cuda"))
            agree[g["reset_ids"]] = True
        tau = br.pre_physics_step(torch.from_numpy(g["actions"][k]).cuda(), torch.from_numpy(g["dof_state"][k]).cuda(), torch.from_numpy(g["root_states"][k]).cuda(),
                                  torch.from_numpy(g["commands"][k]).cuda())
        dec = g["decisions"][k]
        agree &= ~(dec[:, 0] > 0) | (br.ctl.solver_info()[:, :4] == dec).all(axis=1)
        e = _relerr(tau.cpu().numpy(), g["torques"][k])
        worst = max(worst, float(e[agree].max()) if agree.any() else 0.0)
        counted += int(agree.sum())
    out["bridge_h10_" + task] = dict(samples=T * n, decisions_agree_fraction=counted / (T * n), max_torque_rel_err_agreeing=worst)
    print("bridge", task, json.dumps(out["bridge_h10_" + task]), flush=True)
print("CONTROLLER_PARITY_JSON " + json.dumps(out))
