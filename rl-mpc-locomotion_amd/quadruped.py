"""Per-robot constant table of the batched stepper.

Host-side mirror of the reference's ``MPC_Controller/common/Quadruped.py:16-92`` (constants) and
``:96-107`` (hip location sign convention), restated as one float32 row per robot type so the
device kernels can index it by ``robot_type``.  Values are the reference's; nothing is tuned here.
"""
from enum import IntEnum

import numpy as np


class RobotType(IntEnum):
    # order of the reference's RobotType enum (Quadruped.py:6-10, auto() starts at 1) minus one
    ALIENGO = 0
    A1 = 1
    GO1 = 2


# column layout of ROBOT_TABLE (float32)
COL_ABAD, COL_HIP, COL_KNEE = 0, 1, 2          # link lengths            Quadruped.py:17-19,39-41,61-63
COL_HIPLOC = 3                                 # abad location x,y,z     Quadruped.py:21,43,65
COL_MASS = 6                                   # body mass               Quadruped.py:23,45,67
COL_INERTIA = 7                                # body inertia diag xx,yy,zz (off-diagonals are 0)
COL_HEIGHT = 10                                # body height             Quadruped.py:27,49,71
COL_MU = 11                                    # friction coeff (x4)     Quadruped.py:28,50,72
COL_WEIGHTS = 12                               # 13 default MPC weights  Quadruped.py:30-34,52-56,76
ROBOT_COLS = 25

_W_ALIENGO = [1.0, 1.5, 0.0, 0.0, 0.0, 50, 0.0, 0.0, 0.1, 1.0, 1.0, 0.1, 0.0]


def _row(abad, hip, knee, loc, mass, inertia, height, mu, w):
    r = np.zeros(ROBOT_COLS, dtype=np.float64)
    r[COL_ABAD], r[COL_HIP], r[COL_KNEE] = abad, hip, knee
    r[COL_HIPLOC:COL_HIPLOC + 3] = loc
    r[COL_MASS] = mass
    r[COL_INERTIA:COL_INERTIA + 3] = inertia
    r[COL_HEIGHT] = height
    r[COL_MU] = mu
    r[COL_WEIGHTS:COL_WEIGHTS + 13] = w
    return r


# float64 master copy (mass/inertia go to the C++ ConvexMpc constructor as Python floats in the
# reference: ConvexMPCLocomotion.py:102-108); the float32 view is what numpy-side code there uses.
ROBOT_TABLE64 = np.stack([
    _row(0.083, 0.25, 0.25, [0.2399, 0.051, 0.0], 9.041 * 2,
         [0.033260231, 0.16117211, 0.17460442], 0.35, np.float32(0.4), np.asarray(_W_ALIENGO, np.float32)),
    _row(0.08505, 0.2, 0.2, [0.183, 0.047, 0.0], 8.5 * 3,
         np.array([0.017, 0.057, 0.064]) * 10, 0.26, np.float32(0.4),
         np.asarray([0.25, 0.25, 10, 2, 2, 50, 0, 0, 0.3, 0.5, 0.5, 0.1, 0], np.float32)),
    _row(0.08, 0.213, 0.213, [0.1881, 0.04675, 0.0], 5.204 * 2,
         np.array([0.0168128557, 0.063009565, 0.0716547275]) * 5, 0.26, np.float32(0.4),
         np.asarray(_W_ALIENGO, np.float32) * np.float32(10)),
])
ROBOT_TABLE = ROBOT_TABLE64.astype(np.float32)

SIDE_SIGN = np.array([1, -1, 1, -1], dtype=np.float32)       # utils.py:7  FL, FR, RL, RR


def hip_location(robot_type: int, leg: int) -> np.ndarray:
    """Quadruped.getHipLocation (Quadruped.py:96-107): +x for front legs (0,1), +y for left legs (0,2)."""
    loc = ROBOT_TABLE[robot_type, COL_HIPLOC:COL_HIPLOC + 3]
    return np.array([loc[0] if leg in (0, 1) else -loc[0], loc[1] if leg in (0, 2) else -loc[1], loc[2]],
                    dtype=np.float32)


def body_inertia9(robot_type: int) -> list:
    """Row-major 3x3 body inertia as the reference passes it (ConvexMPCLocomotion.py:103)."""
    d = ROBOT_TABLE64[robot_type, COL_INERTIA:COL_INERTIA + 3]
    return [d[0], 0, 0, 0, d[1], 0, 0, 0, d[2]]
