"""bench.py's N > 1 control flow on the CPU: `python bench.py --gpus 2 --backend gloo --emulate` launches itself under
torch.distributed.run exactly as the driver's multi-GPU line does (self-launch, the world == --gpus check, robot sharding, barriers
and max-over-ranks timing, the all-gather leg, JSON from rank 0 only) with the host emulation of the kernels (tests/emu) standing in
for the HIP library -- so the first contact with an 8-GPU node cannot die in control flow.  The numbers mean nothing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, gpus=2):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--backend", "gloo", "--emulate", "--steps", "2", "--warmup", "1",
                          "--repeats", "2", *flags], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout          # rank 0 only
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_line():
    b = _run("--robots", "5")
    assert b["n_gpus"] == 2 and b["steps"] == 2 and b["warmup"] == 1 and b["repeats"] == 2 and b["scaling"] == "weak"
    assert b["config"]["robots_per_gpu"] == 5 and b["config"]["robots_total"] == 10
    assert b["value"] > 0 and abs(b["value"] - 10 * 2 / (b["ms_per_step"] * 2e-3)) < 1e-6 * b["value"]
    assert b["all_gather_torques"]["shape_ok"] and b["all_gather_torques"]["robots_total"] == 10
    assert b["solved_fraction"] == 1.0 and "EMULATED" in b["data"]
    assert b["config"]["seam"] == "ctrl"                 # `value` goes through controller.run (SURVEY 8(d)'s unit incl. the torque map)
    sl = b["all_gather_torques"]["sharded_loop"]           # the product-level sharded stepper with its overlapped exchange
    assert sl["shape_ok"] and sl["robots_total"] == 10 and sl["ms_per_tick_with_exchange"] > 0
    # SURVEY 8(e)'s scaling check inside the bench: the 2-rank torques equal the single-process batch bit for bit; one report per rank
    assert sl["bit_identical_to_single_process"] is True and sl["collective"] == {"backend": "gloo", "ranks": 2}
    assert [r["rank"] for r in sl["ranks"]] == [0, 1] and [r["robots"] for r in sl["ranks"]] == [[0, 5], [5, 10]]
    assert [r["torque_block_sha256"] for r in sl["ranks"]] == sl["torque_blocks_sha256_single_process"]


@pytest.mark.parametrize("flags,total", [(("--robots", "2"), 16), (("--config", "4", "--robots-total", "19"), 19), (("--config", "5", "--robots-total", "17"), 17)])
def test_eight_ranks_dry_run(flags, total):
    """The full node: `bench.py --gpus 8` exactly as the driver launches it, on the CPU (gloo, emulated kernels) for configs 2 / 4 / 5 --
    even and uneven shards, barriers, max-over-ranks, gather leg, JSON from rank 0 only.  Unmeasured on hardware until a SCALE record exists."""
    b = _run(*flags, gpus=8)
    assert b["n_gpus"] == 8 and b["config"]["robots_total"] == total and b["value"] > 0
    assert b["all_gather_torques"]["shape_ok"] and b["all_gather_torques"]["robots_total"] == total


@pytest.mark.parametrize("config,h", [(4, 16), (5, 20)])
def test_strong_scaling_configs_shard_unevenly(config, h):
    """configs 4 / 5 shard a fixed total over the ranks; an odd total gives the ranks different shard sizes (the padded all-gather path)."""
    b = _run("--config", str(config), "--robots-total", "5")
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["config"]["horizon"] == h
    assert b["config"]["robots_total"] == 5 and b["config"]["robots_per_gpu"] == 3            # rank 0 holds 3 of the 5
    assert b["all_gather_torques"]["shape_ok"] and b["all_gather_torques"]["robots_total"] == 5


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--emulate"], capture_output=True, text=True, cwd=ROOT, env=env)
    assert out.returncode != 0 and "--gpus 4 but the launcher started 2" in out.stderr


def test_gloo_needs_emulate():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode != 0 and "needs --emulate" in out.stderr
