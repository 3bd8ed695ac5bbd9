"""Robot sharding across the GPUs of one node.

Robots are independent -- there is no cross-robot term anywhere on the path -- so the batch is cut into
contiguous blocks of robot indices, one per rank (one process per GPU), each rank keeping its robots'
persistent state.  No data-path collective is needed; the only optional exchange is an all-gather of the
per-robot torques ([n_local, 12] float32 per rank) when one consumer wants the whole batch on every
device.  ``torch.distributed`` with backend "nccl" is RCCL on ROCm; the CPU tests run the same code over
"gloo".
"""


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; the first n_total % world ranks get one extra robot."""
    if not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_total: int, world: int):
    return [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]


def all_gather_torques(local, n_total: int, group=None):
    """All-gather the per-rank torque blocks into a [n_total, 12] tensor on every rank (one collective;
    the message is 48 B per robot, i.e. latency-bound on xGMI -- SURVEY.md 5)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    if len(set(sizes)) == 1:
        out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # uneven shards: pad every block to the largest one (collectives want equal sizes), then drop the padding
    mx = max(sizes)
    padded = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def max_over_ranks(seconds: float, device, group=None) -> float:
    """bench.py's timing rule: the job is as slow as its slowest rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class PeerExchange:
    """The same exchange as one-shot direct peer writes (include/mpc_batch.h mpc_peer_*; SURVEY.md 5 recommends it for a 24 KB message): every rank's
    rows go straight into every rank's receive region over xGMI in ONE kernel, flags instead of a collective.  The hipIpc handles travel once, at
    construction, through torch.distributed.all_gather_object (any backend).  put(local) / wait() -> [n_total, width] are stream-ordered on the current
    stream; a rank must wait() for an exchange before it starts the next one.  UNMEASURED across GPUs (tests: two processes on one GPU, a one-rank group)."""

    def __init__(self, n_total, lo, n_local, width=12, group=None, device=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _lib
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.n_total, self.lo, self.n_local, self.width = int(n_total), int(lo), int(n_local), int(width)
        if (self.lo * self.width * 4) % 16:
            raise ValueError("PeerExchange: a rank's block must start on a 16-byte boundary of the batch (lo * width * 4 bytes)")
        with torch.cuda.device(self.device):
            self._h = C.c_void_p()
            self._check(_lib.lib().mpc_peer_create(C.byref(self._h), self.rank, self.world, self.n_total, self.width * 4), "mpc_peer_create")
            mine = (C.c_ubyte * 64)()
            self._check(_lib.lib().mpc_peer_handle(self._h, C.cast(mine, C.c_void_p)), "mpc_peer_handle")
            handles = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(handles, bytes(mine), group=group)
                blob = (C.c_ubyte * (64 * self.world)).from_buffer_copy(b"".join(handles))
                self._check(_lib.lib().mpc_peer_connect(self._h, C.cast(blob, C.c_void_p)), "mpc_peer_connect")
            else:
                self._check(_lib.lib().mpc_peer_connect(self._h, None), "mpc_peer_connect")
        if self.world > 1:
            dist.barrier(group=group)      # every rank has opened every region before anybody writes

    @staticmethod
    def _check(rc, what):
        from . import _lib
        if rc != 0:
            raise _lib.MpcLibraryError(f"{what} failed ({rc}): {_lib.lib().mpc_peer_last_error().decode()}")

    def put(self, local):
        import torch
        from . import _lib
        if local.dtype != torch.float32 or not local.is_cuda or not local.is_contiguous() or local.numel() != self.n_local * self.width:
            raise ValueError("PeerExchange.put: a contiguous cuda float32 [n_local, width] tensor")
        self._check(_lib.lib().mpc_peer_put(self._h, local.data_ptr(), self.lo, self.n_local, torch.cuda.current_stream(self.device).cuda_stream), "mpc_peer_put")

    def wait(self):
        import torch
        from . import _lib
        out = torch.empty((self.n_total, self.width), dtype=torch.float32, device=self.device)
        self._check(_lib.lib().mpc_peer_wait(self._h, out.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream), "mpc_peer_wait")
        return out

    def timeouts(self):
        import ctypes as C
        from . import _lib
        n = C.c_int(0)
        self._check(_lib.lib().mpc_peer_timeouts(self._h, C.addressof(n)), "mpc_peer_timeouts")
        return n.value

    def __del__(self):
        from . import _lib
        h = getattr(self, "_h", None)
        if h and _lib is not None and _lib._LIB is not None:
            _lib._LIB.mpc_peer_destroy(h)
            self._h = None


class ShardedLocomotion:
    """One env batch of `n_total` robots over the ranks of a node (one process per GPU, `torch.distributed` initialised by
    the caller): this rank owns the contiguous block [lo, hi) of robot indices and their persistent state -- warm starts, gait counters,
    swing trajectories -- in its own ``BatchedLocomotion``; nothing migrates and the data path has NO collective.

        sl = ShardedLocomotion(robot_type, gait_id, horizon=10)              # per-robot arrays of the WHOLE batch
        tau_local = sl.run(dof_states, body_states, commands)                # this rank's block (or the whole batch: sliced here) -> [n_local, 12]
        sl.start_gather()                                                    # only when one consumer needs every robot's torques on every device:
        ...                                                                  #   the all-gather runs on a side stream, next to whatever is launched now
        tau_all = sl.torques_all()                                           #   (the next tick's estimator / ctrl_pre / prep kernels) -> [n_total, 12]

    The exchange is one RCCL all-gather of 48 B per robot (24.6 KB per GPU at 4096 robots: latency-bound on xGMI), so it is issued once per tick,
    from a snapshot of the local block taken on the current stream, and waited for only when its result is read.  Uneven shards (the
    first n_total % world ranks hold one robot more) are padded to the largest block for the collective.
    `controller_factory(robot_type, gait_id, **kw)` builds the per-rank controller (default BatchedLocomotion on this rank's GPU; the CPU tests pass
    the host emulation and run over gloo)."""

    def __init__(self, robot_type, gait_id, horizon=10, group=None, device=None, controller_factory=None, exchange="rccl", **kw):
        """exchange: "rccl" -- torch.distributed's all_gather_into_tensor (RCCL on GPUs, gloo in the CPU tests) -- or "peer": one-shot direct peer writes
        (PeerExchange; GPUs only), behind the same start_gather() / torques_all()."""
        import numpy as np
        import torch
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        robot_type, gait_id = np.asarray(robot_type), np.asarray(gait_id)
        self.n_total = len(robot_type)
        self.lo, self.hi = shard_bounds(self.n_total, self.rank, self.world)
        self.n_local = self.hi - self.lo
        self.sizes = shard_sizes(self.n_total, self.world)
        if controller_factory is None:
            from .locomotion import BatchedLocomotion
            controller_factory = BatchedLocomotion
            kw.setdefault("device", device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.local = controller_factory(robot_type[self.lo:self.hi], gait_id[self.lo:self.hi], horizon=horizon, **kw)
        self.device = torch.device(getattr(self.local, "device", "cpu"))
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        mx = max(self.sizes)
        self._stage = torch.zeros((mx, 12), dtype=torch.float32, device=self.device)          # snapshot of the local block (padded)
        self._all = torch.zeros((self.world * mx, 12), dtype=torch.float32, device=self.device)
        self._work = None
        self._tau = None
        if exchange not in ("rccl", "peer"):
            raise ValueError("exchange must be 'rccl' or 'peer'")
        self._peer = None
        if exchange == "peer":
            if self.device.type != "cuda":
                raise ValueError("exchange='peer' needs GPUs (hipIpc peer writes)")
            self._peer = PeerExchange(self.n_total, self.lo, self.n_local, 12, group=group, device=self.device)
            self._peer_pending = False

    def _mine(self, t):
        """a [n_total, ...] batch tensor -> this rank's block; a [n_local, ...] tensor passes through"""
        return t[self.lo:self.hi].contiguous() if t.shape[0] == self.n_total and self.n_total != self.n_local else t

    def run(self, dof_states, body_states, commands):
        cmd = commands if commands.dim() == 1 else self._mine(commands)
        dof = dof_states.reshape(-1, 12, 2) if dof_states.dim() == 2 else dof_states
        self._tau = self.local.run(self._mine(dof), self._mine(body_states), cmd)
        return self._tau

    def start_gather(self):
        """Issue the all-gather of the last run's torques (asynchronously: on the side stream on a GPU, as an async collective on the CPU).
        The local block is snapshot ON THE CURRENT STREAM first -- `run` hands out the controller's persistent torque buffer, which the next
        tick's ctrl_post overwrites on that stream -- so run(k) -> start_gather() -> run(k + 1) -> torques_all() returns tick k's torques.
        A process group of ONE rank (a 1-GPU box) takes the same path: the collective still runs on the side stream."""
        import torch
        import torch.distributed as dist
        if self._tau is None:
            raise RuntimeError("ShardedLocomotion.start_gather: run() first")
        if self._peer is not None:
            if self._peer_pending:                                               # (a rank waits for an exchange before it starts the next one: PeerExchange)
                self._peer.wait()
            self._side.wait_stream(torch.cuda.current_stream(self.device))      # the torques of this tick are complete
            with torch.cuda.stream(self._side):
                self._peer.put(self._tau)                                        # reads the controller's buffer on the side stream ...
            torch.cuda.current_stream(self.device).wait_stream(self._side)      # ... before the next tick's ctrl_post may overwrite it (the put is one short kernel)
            self._peer_pending = True
            return
        if self.world == 1 and not dist.is_initialized():
            self._all[: self.n_local].copy_(self._tau)
            return
        if self._work is not None:
            # a gather nobody read yet still sends from the staging buffer: an async collective runs on the backend's own stream (RCCL) or thread (gloo), so only
            # work.wait() orders the next write of the buffer behind it (wait_stream on the side stream does not)
            self._work.wait()
            self._work = None
            if self._side is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._side)
        self._stage[: self.n_local].copy_(self._tau)                             # ordered before the next run() by the stream itself
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))      # the snapshot is complete
            with torch.cuda.stream(self._side):
                self._work = dist.all_gather_into_tensor(self._all, self._stage, group=self.group, async_op=True)
        else:
            self._work = dist.all_gather_into_tensor(self._all, self._stage, group=self.group, async_op=True)

    def torques_all(self):
        """[n_total, 12] on this rank's device, a fresh tensor (the next gather reuses the receive buffer): waits for the gather started last
        (the current stream waits; the host does not block on a GPU)."""
        import torch
        if self._peer is not None:
            if not self._peer_pending:
                raise RuntimeError("ShardedLocomotion.torques_all: start_gather() first")
            self._peer_pending = False
            return self._peer.wait()                                             # a fresh [n_total, 12] tensor, filled on the current stream once every rank's flag is up
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        mx = max(self.sizes)
        if len(set(self.sizes)) == 1:
            return self._all[: self.n_total].clone()
        return torch.cat([self._all[r * mx: r * mx + self.sizes[r]] for r in range(self.world)], dim=0)

    def reset(self, env_ids=None):
        """``reset_idx(env_ids)`` with GLOBAL robot indices (host array or device tensor): every rank resets the ones it owns.  Device ids are
        shifted by -lo and handed on as they are -- the reset kernels ignore indices outside [0, n_local) -- so there is no host round trip."""
        import numpy as np
        if env_ids is None:
            return self.local.reset()
        if hasattr(env_ids, "is_cuda") and env_ids.is_cuda:
            return self.local.reset(env_ids - self.lo)
        ids = np.asarray(env_ids.cpu() if hasattr(env_ids, "cpu") else env_ids, dtype=np.int64)
        ids = ids[(ids >= self.lo) & (ids < self.hi)] - self.lo
        if len(ids):
            self.local.reset(ids.astype(np.int32))
