"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's weight policy, numpy float32.

Follows RL_Environment/WeightPolicy.py:
  * compute_observations (:120-139)  -> ``observations``
  * step (:94-118)                   -> ``step``: actions = ActorCritic.act_inference(obs), clamp to [-1, 1]
    (``_rescale_actions(-1, 1, .)`` is the identity), then ``actions * MPC_param_scale + MPC_param_const``
    (MPC_Controller/Parameters.py:25-33).
The network itself is third-party code that is absent from /root/reference: rsl_rl (commit 2ad79cf, README.md:33),
``rsl_rl.modules.ActorCritic``: ``actor = nn.Sequential(Linear(num_obs, h0), act, Linear(h0, h1), act, ..., Linear(h_last,
num_actions))`` and ``act_inference(obs) = actor(obs)``; hidden sizes and activation from
RL_Environment/tasks/legged_config_ppo.py:5-9 ([512, 256, 128], 'elu').  Its state_dict keys are ``actor.0.weight``,
``actor.0.bias``, ``actor.2.weight``, ... (every second index is the parameter-free activation).
Pinned by (i) tests/golden/runner_policy_h10.npz -- minted by executing the REFERENCE'S OWN `compute_observations` / `step` and
`RobotRunnerPolicy.run` (taken from the source files by AST, tests/golden/make_golden_runner_policy.py; only the rsl_rl network and the
hydra config are stand-ins) -- and (ii) tests/golden/policy_mlp.npz, minted with torch.nn.Sequential of the same structure
(tests/golden/make_golden_policy.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np

MPC_PARAM_SCALE = np.array([4, 4, 4, 20, 20, 20, 1, 1, 1, 1, 1, 1], dtype=np.float32)   # Parameters.py:25-28
MPC_PARAM_CONST = np.array([5, 5, 5, 50, 50, 50, 1, 1, 1, 1, 1, 1], dtype=np.float32)   # Parameters.py:30-33
ACTOR_DIMS = (48, 512, 256, 128, 12)                                                    # legged_config_ppo.py:7


def actor_params_from_state_dict(sd):
    """[(W, b), ...] of the actor's Linear layers, in order, from an ActorCritic state_dict (torch tensors or arrays)."""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("actor.") and k.endswith(".weight")})
    return [(np.asarray(sd[f"actor.{i}.weight"], dtype=np.float32), np.asarray(sd[f"actor.{i}.bias"], dtype=np.float32)) for i in idx]


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(np.float32)


def act_inference(params, obs):
    h = np.asarray(obs, dtype=np.float32)
    for l, (W, b) in enumerate(params):
        h = (h @ W.T + b).astype(np.float32)
        if l + 1 < len(params):
            h = elu(h)
    return h


def step(params, obs, scale=MPC_PARAM_SCALE, const=MPC_PARAM_CONST):
    """-> (raw actions, MPC weights [n, 12])"""
    a = act_inference(params, obs)
    return a, (np.clip(a, -1.0, 1.0) * scale + const).astype(np.float32)


def observations(dof_states, v_body, omega_body, ground_normal_yaw, commands, actions, lin=1.0, ang=1.0, dof_pos=1.0, dof_vel=1.0):
    """dof_states [n,12,2]; the scale defaults are RL_Environment/cfg/task/Aliengo.yaml:72-75 (all 1)"""
    d = np.asarray(dof_states, dtype=np.float32).reshape(-1, 12, 2)
    return np.concatenate((np.asarray(v_body, np.float32) * np.float32(lin), np.asarray(omega_body, np.float32) * np.float32(ang),
                           -np.asarray(ground_normal_yaw, np.float32),
                           np.asarray(commands, np.float32) * np.array([lin, lin, ang], dtype=np.float32),
                           d[:, :, 0] * np.float32(dof_pos), d[:, :, 1] * np.float32(dof_vel),
                           np.asarray(actions, np.float32)), axis=1).astype(np.float32)
