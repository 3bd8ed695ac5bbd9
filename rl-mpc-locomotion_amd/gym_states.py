"""The reference's INTERACTIVE input container (``Parameters.bridge_MPC_to_RL = False``, the default of MPC_Controller/Parameters.py:19).

``RL_MPC_Locomotion.py:96-101`` hands every controller what Isaac Gym's ``get_actor_dof_states`` / ``get_actor_rigid_body_states``
return -- numpy STRUCTURED arrays: ``dof_states["pos"] / ["vel"]`` (12 joints; read at ``LegController.py:99-101`` and by
``WeightPolicy.compute_observations``, WeightPolicy.py:131-132) and ``body_states["pose"]["p" / "r"]``, ``["vel"]["linear" / "angular"]``
(one rigid body; ``StateEstimator.py:70-79``).  The RL bridge uses plain arrays instead: ``dof_states[12, 2]`` and ``body_states[13]``
= pos3, quat xyzw, linear velocity3, angular velocity3 (``LegController.py:96-98``, ``StateEstimator.py:58-69``).

The batched stepper works on the second form; this module maps the first onto it (host arrays: the interactive seam is a per-robot
viewer loop, its data is on the host already).  The dtypes are Isaac Gym's ``gymapi.DofState.dtype`` / ``RigidBodyState.dtype``.
"""
import numpy as np

VEC3 = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4")])
QUAT = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("w", "f4")])
DOF_STATE = np.dtype([("pos", "f4"), ("vel", "f4")])
BODY_STATE = np.dtype([("pose", [("p", VEC3), ("r", QUAT)]), ("vel", [("linear", VEC3), ("angular", VEC3)])])


def is_structured(x):
    """True for a structured numpy array / record, or a list of them (one per robot)."""
    if isinstance(x, (list, tuple)) and len(x):
        x = x[0]
    return isinstance(x, (np.ndarray, np.void)) and x.dtype.names is not None


def _stack(x):
    return np.stack([np.asarray(e) for e in x]) if isinstance(x, (list, tuple)) else np.asarray(x)


def dof_states_to_array(dof_states):
    """structured [N, 12] (or [12], or a list of N [12]) with fields pos, vel -> float32 [N, 12, 2]."""
    d = _stack(dof_states)
    if d.ndim == 1:
        d = d[None]
    if d.ndim != 2 or d.shape[1] != 12:
        raise ValueError("dof_states: expected 12 joints per robot")
    return np.ascontiguousarray(np.stack([d["pos"], d["vel"]], axis=-1), dtype=np.float32)


def body_states_to_array(body_states):
    """structured [N] (or a scalar record, or a list of N records) with fields pose.p, pose.r, vel.linear, vel.angular -> float32 [N, 13]
    = pos3, quat xyzw, linear velocity3, angular velocity3 (what the RL bridge passes, StateEstimator.py:58-69)."""
    b = _stack(body_states)
    b = b.reshape(-1)
    out = np.zeros((len(b), 13), dtype=np.float32)
    for i, k in enumerate("xyz"):
        out[:, i] = b["pose"]["p"][k]
        out[:, 7 + i] = b["vel"]["linear"][k]
        out[:, 10 + i] = b["vel"]["angular"][k]
    for i, k in enumerate("xyzw"):
        out[:, 3 + i] = b["pose"]["r"][k]
    return out


def to_structured(dof, body):
    """The inverse (tests, examples): float arrays [N, 12, 2] / [N, 13] -> (DOF_STATE [N, 12], BODY_STATE [N])."""
    dof = np.asarray(dof, dtype=np.float32).reshape(-1, 12, 2)
    body = np.asarray(body, dtype=np.float32).reshape(-1, 13)
    d = np.zeros((len(dof), 12), dtype=DOF_STATE)
    d["pos"], d["vel"] = dof[:, :, 0], dof[:, :, 1]
    b = np.zeros(len(body), dtype=BODY_STATE)
    for i, k in enumerate("xyz"):
        b["pose"]["p"][k] = body[:, i]; b["vel"]["linear"][k] = body[:, 7 + i]; b["vel"]["angular"][k] = body[:, 10 + i]
    for i, k in enumerate("xyzw"):
        b["pose"]["r"][k] = body[:, 3 + i]
    return d, b
