"""Multi-GPU path on CPU: world_size-2 gloo processes run the sharding / gather / timing helpers that
bench.py --gpus N and a sharded env use (the per-robot compute is stood in for by the host emulation of
the kernel, which is what makes 'same batch on 1 vs k ranks is bit-identical per robot' checkable here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rl_mpc_locomotion_amd  # noqa: F401
from rl_mpc_locomotion_amd.sharding import all_gather_torques, max_over_ranks, shard_bounds, shard_sizes
from tests.helpers import load_golden


def test_shard_bounds_cover_batch_exactly():
    for n in (0, 1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            bounds = [shard_bounds(n, r, world) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
            assert max(shard_sizes(n, world)) - min(shard_sizes(n, world)) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu import EmuBatch
    g = load_golden("solver_h10_cfg3")
    lo, hi = shard_bounds(n_total, rank, world)
    emu = EmuBatch(g["mass"][lo:hi], g["inertia_diag"][lo:hi], 10, float(g["dt_mpc"]), float(g["alpha"]))
    outs = []
    for s in range(2):   # a cold and a warm solve: the shard keeps its own warm-start state
        f = emu.solve(g[f"inputs_{s}"][lo:hi], nthreads=1)
        local = torch.from_numpy(f[:, :12].astype(np.float32))
        outs.append(all_gather_torques(local, n_total).numpy())
    slowest = max_over_ranks(1.0 + rank, torch.device("cpu"))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), a=outs[0], b=outs[1], slowest=slowest)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [12, 13])
def test_two_rank_shards_match_single_process(tmp_path, n_total):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    from tests.emu.emu import EmuBatch
    g = load_golden("solver_h10_cfg3")
    emu = EmuBatch(g["mass"][:n_total], g["inertia_diag"][:n_total], 10, float(g["dt_mpc"]), float(g["alpha"]))
    ref = [emu.solve(g[f"inputs_{s}"][:n_total], nthreads=1)[:, :12].astype(np.float32) for s in range(2)]
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["a"], ref[0]) and np.array_equal(z["b"], ref[1])   # bit-identical per robot
        assert float(z["slowest"]) == 2.0


def _nccl_worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))     # RCCL
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    from tests.helpers import inertia9_from_diag
    g = load_golden("solver_h10_cfg3")
    lo, hi = shard_bounds(n_total, rank, world)
    gpu = BatchedConvexMpc(g["mass"][lo:hi], inertia9_from_diag(g["inertia_diag"][lo:hi]), 10, float(g["dt_mpc"]), float(g["alpha"]), device=f"cuda:{rank}")
    outs = []
    for s in range(2):
        f, _ = gpu.solve(torch.from_numpy(g[f"inputs_{s}"][lo:hi]).to(f"cuda:{rank}"))
        outs.append(all_gather_torques(f[:, :12].to(torch.float32), n_total).cpu().numpy())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), a=outs[0], b=outs[1])
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_gpu_shards_match_one_gpu(tmp_path):
    """The N > 1 path on hardware: two ranks over RCCL, each solving its shard on its own GPU, all-gather of the per-robot forces;
    bit-identical per robot to the single-GPU batch.  Needs two visible devices (the round-end 8-GPU node; the 1-GPU boxes skip)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    n_total, world, port = 13, 2, _free_port()
    mp.spawn(_nccl_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    from tests.helpers import inertia9_from_diag
    g = load_golden("solver_h10_cfg3")
    gpu = BatchedConvexMpc(g["mass"][:n_total], inertia9_from_diag(g["inertia_diag"][:n_total]), 10, float(g["dt_mpc"]), float(g["alpha"]), device="cuda:0")
    ref = [gpu.solve(torch.from_numpy(g[f"inputs_{s}"][:n_total]).cuda())[0][:, :12].to(torch.float32).cpu().numpy().copy() for s in range(2)]
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(z["a"], ref[0]) and np.array_equal(z["b"], ref[1])
