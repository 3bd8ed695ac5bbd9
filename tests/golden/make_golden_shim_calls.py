"""Golden fixture for the per-robot plugin seam (B1): every `compute_contact_forces` call the UNMODIFIED reference Python makes
while RobotRunnerMin replays BASELINE configs[0] (one Aliengo, trot, h = 10; the inputs of controller_h10_config1.npz), with the
13 arguments exactly as the reference passes them (packed into the 56 + 4 h record, float64) and the list the oracle returned.

    python tests/golden/make_golden_shim_calls.py        (build container only: needs /root/reference)

The GPU test replays these calls through rl_mpc_locomotion_amd.mpc_osqp.ConvexMpc -- the module INTEGRATION.md puts on
sys.modules["mpc_osqp"] -- and compares the returned lists (tests/test_dropin.py)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import rl_mpc_locomotion_amd  # noqa: E402,F401
from rl_mpc_locomotion_amd.layout import in_len, pack_args  # noqa: E402
from oracle.refmpc import RefConvexMpc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CALLS = {"ctor": None, "rec": [], "out": [], "ok": [], "exact": []}
RETURN_EXACT = [False]      # second replay: the seam returns the exact optimum (what the reference's own choice, mpc.QPOASES, asks for)


class Recording(RefConvexMpc):
    def __init__(self, *a):
        CALLS["ctor"] = [np.asarray(x, dtype=np.float64).reshape(-1) for x in a[:6]] + [int(a[6])]
        super().__init__(*a)

    def compute_contact_forces(self, *args):
        rec = np.zeros(in_len(self.h), dtype=np.float64)
        pack_args(self.h, *args, out=rec)
        if RETURN_EXACT[0]:
            return self.solve_exact(rec).tolist()
        CALLS["exact"].append(self.solve_exact(rec).copy())       # what the qpOASES branch would return (oracle/README.md)
        out = super().compute_contact_forces(*args)
        CALLS["rec"].append(rec)
        CALLS["ok"].append(len(out) > 0)
        CALLS["out"].append(np.asarray(out, dtype=np.float64) if len(out) else np.full(12 * self.h, np.nan))
        return out


m = types.ModuleType("mpc_osqp")
m.ConvexMpc = Recording
m.OSQP, m.QPOASES = 0, 1
sys.modules["mpc_osqp"] = m
from MPC_Controller.Parameters import Parameters  # noqa: E402
from MPC_Controller.utils import GaitType  # noqa: E402
Parameters.bridge_MPC_to_RL = True
from MPC_Controller.robot_runner.RobotRunnerMin import RobotRunnerMin  # noqa: E402
from MPC_Controller.common.Quadruped import RobotType  # noqa: E402


def main(ticks=1000):
    g = np.load(os.path.join(HERE, "controller_h10_config1.npz"))
    Parameters.flat_ground = bool(g["flat_ground"])
    Parameters.cmpc_gait = GaitType.TROT
    runner = RobotRunnerMin()
    runner.init(RobotType.ALIENGO)
    tau = np.zeros((ticks, 12), np.float32)
    for k in range(ticks):
        tau[k] = runner.run(g["dof"][k, 0], g["body"][k, 0], g["cmd"][k, 0])
    assert np.array_equal(tau, g["torque"][:ticks, 0]), "replay does not reproduce the controller golden"
    # the same replay with the exact optimum behind the seam: the torques of the reference AS SHIPPED (it passes mpc.QPOASES)
    RETURN_EXACT[0] = True
    runner = RobotRunnerMin()
    runner.init(RobotType.ALIENGO)
    tau_exact = np.zeros((ticks, 12), np.float32)
    for k in range(ticks):
        tau_exact[k] = runner.run(g["dof"][k, 0], g["body"][k, 0], g["cmd"][k, 0])
    c = CALLS["ctor"]
    np.savez_compressed(os.path.join(HERE, "shim_calls_config1.npz"), mass=c[0], inertia=c[1], num_legs=c[2], horizon=c[3], timestep=c[4],
                        alpha=c[5], solver=c[6], rec=np.array(CALLS["rec"]), out=np.array(CALLS["out"]), ok=np.array(CALLS["ok"]), out_exact=np.array(CALLS["exact"]), ticks=ticks, torque_exact=tau_exact)
    print("shim_calls_config1 written:", len(CALLS["rec"]), "calls, ctor", [x.tolist() if hasattr(x, "tolist") else x for x in c])


if __name__ == "__main__":
    main()
