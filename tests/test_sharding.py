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


# ---- ShardedLocomotion: one env batch over the ranks of a node ------------------------------------------------------------
class _EmuController:
    """the host emulation of the controller kernels behind BatchedLocomotion's interface (CPU tensors), for the gloo tests"""

    def __init__(self, robot_type, gait_id, horizon=10, **kw):
        from tests.emu.emu import EmuLocomotion
        self.e = EmuLocomotion(robot_type, gait_id, horizon=horizon, nthreads=1, **kw)
        self.device = "cpu"

    def run(self, dof, body, cmd):
        tau = torch.from_numpy(self.e.run(dof.numpy(), body.numpy(), cmd.numpy()))
        if getattr(self, "torques", None) is None:
            self.torques = torch.zeros_like(tau)
        self.torques.copy_(tau)            # one persistent output buffer, overwritten by every tick -- like BatchedLocomotion.torques
        return self.torques

    def reset(self, env_ids=None):
        self.e.reset(env_ids)


def _tick_inputs(n_total, ticks):
    from rl_mpc_locomotion_amd.synthetic import TickStream
    ts = TickStream(n_total, seed=77, config=3)
    return ts, [tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in ts.tick(k)) for k in range(ticks)]


def _sharded_worker(rank, world, port, n_total, ticks, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rl_mpc_locomotion_amd.sharding import ShardedLocomotion
    ts, ins = _tick_inputs(n_total, ticks)
    sl = ShardedLocomotion(ts.robot_type, ts.gait_id, horizon=10, controller_factory=_EmuController)
    assert (sl.lo, sl.hi) == shard_bounds(n_total, rank, world)
    outs = []
    for k in range(ticks):
        if k == 3:
            sl.reset(np.array([0, n_total - 1, n_total // 2]))       # global ids: every rank resets what it owns
        local = sl.run(*ins[k])                                       # whole-batch tensors: sliced to this rank's block
        assert tuple(local.shape) == (sl.n_local, 12)
        sl.start_gather()
        outs.append(sl.torques_all().numpy().copy())
    np.save(os.path.join(out_dir, f"sharded{rank}.npy"), np.stack(outs))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 7), (8, 20)])
def test_sharded_locomotion_matches_single_process(tmp_path, world, n_total):
    """ShardedLocomotion on `world` gloo ranks (uneven shards, global-id resets, the asynchronous all-gather) returns, on every rank, the
    torques of the same batch run in one process -- bit for bit.  world = 8: the dry run of a full node."""
    ticks, port = 6, _free_port()
    mp.spawn(_sharded_worker, args=(world, port, n_total, ticks, str(tmp_path)), nprocs=world, join=True)
    ts, ins = _tick_inputs(n_total, ticks)
    one = _EmuController(ts.robot_type, ts.gait_id)
    ref = []
    for k in range(ticks):
        if k == 3:
            one.reset(np.array([0, n_total - 1, n_total // 2], dtype=np.int32))
        ref.append(one.run(*ins[k]).numpy().copy())
    ref = np.stack(ref)
    assert np.abs(ref).max() > 1.0
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"sharded{r}.npy"), ref), f"rank {r}"


def _overlap_worker(rank, world, port, n_total, ticks, out_dir):
    """the documented overlap: run(k) -> start_gather() -> run(k + 1) -> torques_all() must return tick k's torques"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rl_mpc_locomotion_amd.sharding import ShardedLocomotion
    ts, ins = _tick_inputs(n_total, ticks)
    sl = ShardedLocomotion(ts.robot_type, ts.gait_id, horizon=10, controller_factory=_EmuController)
    outs = []
    sl.run(*ins[0]); sl.start_gather()
    for k in range(1, ticks):
        sl.run(*ins[k])                        # overwrites the controller's torque buffer while tick k - 1's exchange is in flight
        got = sl.torques_all()                 # tick k - 1
        sl.start_gather()                      # tick k; must not disturb what was just returned
        outs.append(got.numpy().copy())
        assert np.array_equal(outs[-1], got.numpy())
    outs.append(sl.torques_all().numpy().copy())
    np.save(os.path.join(out_dir, f"overlap{rank}.npy"), np.stack(outs))
    dist.destroy_process_group()


def test_overlapped_gather_returns_the_tick_it_was_started_for(tmp_path):
    world, n_total, ticks, port = 2, 7, 5, _free_port()
    mp.spawn(_overlap_worker, args=(world, port, n_total, ticks, str(tmp_path)), nprocs=world, join=True)
    ts, ins = _tick_inputs(n_total, ticks)
    one = _EmuController(ts.robot_type, ts.gait_id)
    ref = np.stack([one.run(*ins[k]).numpy().copy() for k in range(ticks)])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"overlap{r}.npy"), ref), f"rank {r}"


def _one_rank_rccl_worker(rank, world, port, n_total, ticks, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)               # RCCL, a communicator of one rank
    from rl_mpc_locomotion_amd.sharding import ShardedLocomotion, all_gather_torques, max_over_ranks
    ts, ins = _tick_inputs(n_total, ticks)
    sl = ShardedLocomotion(ts.robot_type, ts.gait_id, horizon=10, device=dev)
    assert sl.world == 1 and dist.is_initialized()
    outs = []
    sl.run(*(a.to(dev) for a in ins[0])); sl.start_gather()
    for k in range(1, ticks):
        sl.run(*(a.to(dev) for a in ins[k]))
        got = sl.torques_all()
        sl.start_gather()
        outs.append(got.cpu().numpy().copy())
    outs.append(sl.torques_all().cpu().numpy().copy())
    direct = all_gather_torques(torch.from_numpy(outs[-1]).to(dev), n_total).cpu().numpy()
    assert np.array_equal(direct, outs[-1]) and max_over_ranks(1.5, dev) == 1.5
    np.save(os.path.join(out_dir, "rccl1.npy"), np.stack(outs))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_exchange_runs_on_one_gpu(tmp_path):
    """What a 1-GPU box can execute of the N > 1 path: a torch.distributed process group over RCCL (backend "nccl") with ONE rank,
    ShardedLocomotion's asynchronous all_gather_into_tensor on its side stream in the documented overlap (run(k) -> start_gather() ->
    run(k + 1) -> torques_all()), all_gather_torques and max_over_ranks -- against BatchedLocomotion, bit for bit.  Communicators of two
    and eight ranks stay unmeasured on hardware until a multi-GPU node runs the two tests below."""
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    n_total, ticks, port = 33, 5, _free_port()
    mp.spawn(_one_rank_rccl_worker, args=(1, port, n_total, ticks, str(tmp_path)), nprocs=1, join=True)
    ts, ins = _tick_inputs(n_total, ticks)
    one = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=10, device="cuda:0")
    ref = np.stack([one.run(*(a.cuda() for a in ins[k])).cpu().numpy().copy() for k in range(ticks)])
    assert np.array_equal(np.load(tmp_path / "rccl1.npy"), ref)


@pytest.mark.gpu
def test_sharded_locomotion_single_gpu_equals_batched():
    """world = 1 (no process group): ShardedLocomotion is BatchedLocomotion plus a no-op exchange; device-side global-id reset included."""
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    from rl_mpc_locomotion_amd.sharding import ShardedLocomotion
    n, ticks = 33, 6
    ts, ins = _tick_inputs(n, ticks)
    sl = ShardedLocomotion(ts.robot_type, ts.gait_id, horizon=10, device="cuda:0")
    one = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=10, device="cuda:0")
    ids = torch.tensor([0, n - 1, n // 2], dtype=torch.int32, device="cuda:0")
    for k in range(ticks):
        d = tuple(a.cuda() for a in ins[k])
        if k == 3:
            sl.reset(ids); one.reset(ids)
        a = sl.run(*d); sl.start_gather()
        b = one.run(*d)
        assert torch.equal(sl.torques_all(), b) and torch.equal(a, b)
    assert float(b.abs().max()) > 1.0


# ---- the exchange as one-shot direct peer writes (sharding.PeerExchange, include/mpc_batch.h mpc_peer_*) ------------------------------------------------
def _peer_worker(rank, world, port, n_total, ticks, out_dir, devices):
    """`world` processes; rank r on GPU devices[r] (the 1-GPU boxes: every rank on GPU 0 -- the hipIpc handles, the flags and the double buffering are the
    real ones, only the link is not xGMI).  The host side (handle exchange, barriers) runs over gloo: RCCL refuses two ranks on one device."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(devices[rank])
    dev = torch.device(f"cuda:{devices[rank]}")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rl_mpc_locomotion_amd.sharding import ShardedLocomotion
    ts, ins = _tick_inputs(n_total, ticks)
    sl = ShardedLocomotion(ts.robot_type, ts.gait_id, horizon=10, device=dev, exchange="peer")
    outs, prev = [], None
    for k in range(ticks):
        if k == 3:
            sl.reset(torch.tensor([0, n_total - 1, n_total // 2], dtype=torch.int32, device=dev))    # global ids on the device
        if prev is not None:
            outs.append(sl.torques_all().cpu().numpy().copy())        # last tick's exchange, read after this tick's inputs were staged
        sl.run(*(a.to(dev) for a in ins[k]))
        sl.start_gather()
        prev = k
    outs.append(sl.torques_all().cpu().numpy().copy())
    assert sl._peer.timeouts() == 0
    np.save(os.path.join(out_dir, f"peer{rank}.npy"), np.stack(outs))
    dist.barrier()
    del sl
    dist.destroy_process_group()


def _peer_reference(n_total, ticks):
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    ts, ins = _tick_inputs(n_total, ticks)
    one = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=10, device="cuda:0")
    ref = []
    for k in range(ticks):
        if k == 3:
            one.reset(np.array([0, n_total - 1, n_total // 2], dtype=np.int32))
        ref.append(one.run(*(a.cuda() for a in ins[k])).cpu().numpy().copy())
    return np.stack(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_peer_write_exchange_between_processes_of_one_gpu(tmp_path, world):
    """ShardedLocomotion(exchange="peer") -- hipIpc receive regions, one put kernel, one wait kernel, two parities -- in the documented overlap pattern,
    against one BatchedLocomotion of the whole batch: bit-identical torques on every rank, no wait kernel timed out.  world = 2: two PROCESSES sharing
    GPU 0 (what a 1-GPU box can execute of the cross-process path); world = 1: the degenerate group.  n_total = 16 puts rank 1's block on a 16-byte boundary."""
    n_total, ticks, port = 16, 6, _free_port()
    mp.spawn(_peer_worker, args=(world, port, n_total, ticks, str(tmp_path), [0] * world), nprocs=world, join=True)
    ref = _peer_reference(n_total, ticks)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"peer{r}.npy"), ref), f"rank {r}"


@pytest.mark.gpu
def test_peer_write_exchange_two_gpus(tmp_path):
    """The same across two GPUs (xGMI peer writes): needs two visible devices (1-GPU boxes skip) -- so far unmeasured on hardware."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    n_total, ticks, port = 16, 6, _free_port()
    mp.spawn(_peer_worker, args=(2, port, n_total, ticks, str(tmp_path), [0, 1]), nprocs=2, join=True)
    ref = _peer_reference(n_total, ticks)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"peer{r}.npy"), ref), f"rank {r}"


def _sharded_nccl_worker(rank, world, port, n_total, ticks, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device(f"cuda:{rank}")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # RCCL
    from rl_mpc_locomotion_amd.sharding import ShardedLocomotion
    ts, ins = _tick_inputs(n_total, ticks)
    sl = ShardedLocomotion(ts.robot_type, ts.gait_id, horizon=10, device=dev)
    outs, prev = [], None
    for k in range(ticks):
        if k == 3:
            sl.reset(torch.tensor([0, n_total - 1, n_total // 2], dtype=torch.int32, device=dev))    # global ids on the device
        if prev is not None:
            outs.append(sl.torques_all().cpu().numpy().copy())        # last tick's exchange, read after this tick's inputs were staged
        sl.run(*(a.to(dev) for a in ins[k]))
        sl.start_gather()
        prev = k
    outs.append(sl.torques_all().cpu().numpy().copy())
    np.save(os.path.join(out_dir, f"sharded{rank}.npy"), np.stack(outs))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_locomotion_two_gpus_match_one(tmp_path):
    """ShardedLocomotion over RCCL on two GPUs (uneven shards, overlapped exchange) against one GPU's batch: bit-identical torques on both ranks.
    Needs two visible devices (1-GPU boxes skip)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from rl_mpc_locomotion_amd.locomotion import BatchedLocomotion
    n_total, world, ticks, port = 13, 2, 6, _free_port()
    mp.spawn(_sharded_nccl_worker, args=(world, port, n_total, ticks, str(tmp_path)), nprocs=world, join=True)
    ts, ins = _tick_inputs(n_total, ticks)
    one = BatchedLocomotion(ts.robot_type, ts.gait_id, horizon=10, device="cuda:0")
    ref = []
    for k in range(ticks):
        if k == 3:
            one.reset(np.array([0, n_total - 1, n_total // 2], dtype=np.int32))
        ref.append(one.run(*(a.cuda() for a in ins[k])).cpu().numpy().copy())
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"sharded{r}.npy"), np.stack(ref)), f"rank {r}"
