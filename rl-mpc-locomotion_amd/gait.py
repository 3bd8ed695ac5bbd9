"""Offset/duration gait tables (host mirror).

Restates ``MPC_Controller/convex_MPC/Gait.py:10-93`` (OffsetDurationGait) and the gait table of
``ConvexMPCLocomotion.py:30-56,229-241`` as vectorised numpy over a batch of robots.  The device
kernels evaluate the same formulas per robot; this module is the host-side definition the tests and
the synthetic workload use.  Gait ids follow ``Parameters.cmpc_gait.value`` (utils.py:17-24 plus the
commented-out ids the ``run`` dispatch still honours, ConvexMPCLocomotion.py:229-241).
"""
import numpy as np

# id -> (offsets, durations) in 10-segment units, legs FL FR RL RR.  Unknown ids fall back to trot
# exactly like the if/elif chain of ConvexMPCLocomotion.py:229-241.
GAIT_TABLE_10 = {
    0: ([0, 5, 5, 0], [5, 5, 5, 5]),   # trotting
    1: ([5, 5, 0, 0], [4, 4, 4, 4]),   # bounding
    2: ([0, 0, 0, 0], [4, 4, 4, 4]),   # pronking
    3: ([5, 0, 5, 0], [5, 5, 5, 5]),   # pacing
    5: ([0, 2, 7, 9], [4, 4, 4, 4]),   # galloping
    6: ([0, 3, 5, 8], [5, 5, 5, 5]),   # walking
    7: ([0, 5, 5, 0], [4, 4, 4, 4]),   # trot running
}
NUM_GAIT_IDS = 8


def gait_arrays(n_segments: int = 10):
    """(offsets[8,4], durations[8,4]) int32 for an n_segments horizon.

    The reference hard-codes 10 segments (ConvexMPCLocomotion.py:27).  For the longer horizons of
    BASELINE configs 4-5 the tables are rescaled as SURVEY.md 8(d) prescribes (trot at h=16:
    offsets [0,8,8,0], durations [8]*4): value * n_segments / 10, rounded half up.
    """
    off = np.zeros((NUM_GAIT_IDS, 4), dtype=np.int32)
    dur = np.zeros((NUM_GAIT_IDS, 4), dtype=np.int32)
    for gid in range(NUM_GAIT_IDS):
        o, d = GAIT_TABLE_10.get(gid, GAIT_TABLE_10[0])
        off[gid] = np.floor(np.asarray(o) * n_segments / 10.0 + 0.5)
        dur[gid] = np.floor(np.asarray(d) * n_segments / 10.0 + 0.5)
    return off, dur


def mpc_table(gait_id, iteration_counter, iterations_between_mpc: int, n_segments: int = 10):
    """Gait.getMpcTable (Gait.py:69-84) for a batch: returns float32 [N, n_segments*4], [step][leg].

    ``iteration`` follows Gait.setIterations (Gait.py:26-28): true division, so it is fractional when
    the counter is not a multiple of iterations_between_mpc; the comparison chain is reproduced in
    floating point as written.
    """
    gait_id = np.atleast_1d(np.asarray(gait_id))
    it = np.atleast_1d(np.asarray(iteration_counter, dtype=np.float64))
    off, dur = gait_arrays(n_segments)
    o = off[gait_id].astype(np.float64)   # [N,4]
    d = dur[gait_id].astype(np.float64)
    iteration = (it / iterations_between_mpc) % n_segments            # Gait.py:27
    i = np.arange(n_segments, dtype=np.float64)[None, :, None]        # [1,S,1]
    itr = (i + iteration[:, None, None] + 1) % n_segments             # Gait.py:73
    prog = itr - o[:, None, :]
    prog = np.where(prog < 0, prog + n_segments, prog)
    tab = (prog < d[:, None, :]).astype(np.float32)
    return tab.reshape(len(gait_id), n_segments * 4)
