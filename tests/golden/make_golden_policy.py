"""Mint tests/golden/policy_mlp.npz: the actor of rsl_rl's ActorCritic (nn.Sequential Linear/ELU stack,
48-512-256-128-12, RL_Environment/tasks/legged_config_ppo.py:5-9) evaluated by torch on the CPU in float32, plus the
WeightPolicy.step post-processing (RL_Environment/WeightPolicy.py:94-118).  rsl_rl is not installed here, so the
Sequential is rebuilt with the same structure and state_dict key names (actor.0, actor.2, ...).
Run from the repo root:  python tests/golden/make_golden_policy.py"""
import os
import numpy as np
import torch

torch.manual_seed(20240924)
dims = [48, 512, 256, 128, 12]
mods = []
for i in range(4):
    mods.append(torch.nn.Linear(dims[i], dims[i + 1]))
    if i < 3:
        mods.append(torch.nn.ELU())
actor = torch.nn.Sequential(*mods).float().eval()
with torch.no_grad():
    for m in actor:
        if isinstance(m, torch.nn.Linear):   # trained-policy-sized weights: outputs spread over and beyond [-1, 1]
            m.weight.mul_(2.0)
rng = np.random.default_rng(7)
n = 77                                   # not a multiple of the kernel's 32-robot tile
obs = rng.normal(0, 1.0, (n, 48)).astype(np.float32)
obs[5] = 0.0
obs[6] *= 10.0
with torch.no_grad():
    act = actor(torch.from_numpy(obs)).numpy()
scale = torch.tensor([4, 4, 4, 20, 20, 20, 1, 1, 1, 1, 1, 1], dtype=torch.float)    # Parameters.py:25-33
const = torch.tensor([5, 5, 5, 50, 50, 50, 1, 1, 1, 1, 1, 1], dtype=torch.float)
weights = torch.mul(torch.clamp(torch.from_numpy(act), -1.0, 1.0), scale).add(const).numpy()
sd = {f"actor.{k}": v.numpy() for k, v in actor.state_dict().items()}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "policy_mlp.npz")
np.savez_compressed(out, obs=obs, actions=act, weights=weights, **{k.replace(".", "__"): v for k, v in sd.items()})
print(out, os.path.getsize(out), "bytes; |actions| > 1 fraction", float((np.abs(act) > 1).mean()))
