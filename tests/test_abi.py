"""The C-ABI library builds, loads and exports exactly what include/mpc_batch.h declares (no compute)."""
import ctypes
import os
import re

import pytest

import rl_mpc_locomotion_amd  # noqa: F401
from rl_mpc_locomotion_amd import _lib
from tests.helpers import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "mpc_batch.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpc_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    import __graft_entry__ as g
    g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mpc_batch.h but not exported"
    assert sorted(_lib.SYMBOLS) == names


def test_input_len_and_horizons():
    L = _lib.lib()
    assert L.mpc_input_len(10) == 96 and L.mpc_input_len(16) == 120
    buf = (ctypes.c_int * 32)()
    k = L.mpc_supported_horizons(buf, 32)
    assert list(buf[:k]) == list(range(2, 21))      # ConvexMpc takes any planning_horizon (mpc_osqp.cc:186-190): every h up to 20 ships
    small = (ctypes.c_int * 4)()
    assert L.mpc_supported_horizons(small, 4) == k and list(small) == [2, 3, 4, 5]      # the count is returned whatever the capacity


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    with pytest.raises(_lib.MpcLibraryError):
        BatchedConvexMpc([18.0], [[0.03, 0, 0, 0, 0.16, 0, 0, 0, 0.17]], 10, 0.02)
    from rl_mpc_locomotion_amd import mpc_osqp
    with pytest.raises(_lib.MpcLibraryError):
        mpc_osqp.ConvexMpc(18.0, [0.03, 0, 0, 0, 0.16, 0, 0, 0, 0.17], 4, 10, 0.02, 1e-5, mpc_osqp.QPOASES)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "rl-mpc-locomotion_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt.replace("oracle/README", ""), f


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """Round 6's entry points (mpc_ctrl_set_gait_device, mpc_pack_commands_scaled, mpc_ctrl_policy_observations, mpc_ctrl_run_fsm_estimated, mpc_peer_*) validate
    their arguments before they touch the device: MPC_E_ARG (-1) and a message, on a box without a GPU too."""
    L = _lib.lib()
    MPC_E_ARG = -1
    z = ctypes.c_void_p(0)
    assert L.mpc_ctrl_set_gait_device(z, z, z) == MPC_E_ARG and b"mpc_ctrl_set_gait_device" in L.mpc_last_error()
    assert L.mpc_pack_commands_scaled(0, z, z, z, z, z, z) == MPC_E_ARG and b"mpc_pack_commands_scaled" in L.mpc_last_error()
    assert L.mpc_ctrl_policy_observations(z, z, z, z, z, z, z) == MPC_E_ARG
    assert L.mpc_ctrl_run_fsm_estimated(z, z, z, z, z, z, z) == MPC_E_ARG
    h = ctypes.c_void_p()
    assert L.mpc_peer_create(ctypes.byref(h), 3, 2, 16, 48) == MPC_E_ARG and b"mpc_peer_create" in L.mpc_peer_last_error()      # rank outside the group
    assert L.mpc_peer_create(ctypes.byref(h), 0, 17, 16, 48) == MPC_E_ARG                                                        # more than 16 ranks
    assert L.mpc_peer_put(z, z, 0, 0, z) == MPC_E_ARG and L.mpc_peer_wait(z, z, z) == MPC_E_ARG and L.mpc_peer_timeouts(z, z) == MPC_E_ARG
    L.mpc_peer_destroy(z)      # a null handle is ignored


def test_sharded_locomotion_rejects_the_peer_exchange_without_gpus():
    import numpy as np
    from rl_mpc_locomotion_amd.sharding import ShardedLocomotion

    class Ctl:      # stands for the per-rank controller of the CPU tests
        device = "cpu"

        def __init__(self, robot_type, gait_id, horizon=10):
            pass
    with pytest.raises(ValueError):
        ShardedLocomotion(np.zeros(4, np.int32), np.zeros(4, np.int32), controller_factory=Ctl, exchange="peer")
    with pytest.raises(ValueError):
        ShardedLocomotion(np.zeros(4, np.int32), np.zeros(4, np.int32), controller_factory=Ctl, exchange="carrier pigeon")
