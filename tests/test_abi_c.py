"""The C ABI from C: tests/abi_c/abi_smoke.c is compiled with gcc against include/mpc_batch.h only and driven here.
CPU: the library loads and exports every documented symbol.  GPU: a solve through the C program equals the solve
through the Python host layer bit for bit (same library, same inputs)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi_c", "abi_smoke.c")
LIB = os.path.join(ROOT, "rl-mpc-locomotion_amd", "csrc", "libmpc_batch.so")


def _build(tmp_path):
    exe = os.path.join(str(tmp_path), "abi_smoke")
    subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-ldl", "-o", exe], check=True)
    return exe


def test_c_consumer_resolves_every_symbol(tmp_path):
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd import _lib
    if not os.path.exists(LIB):
        pytest.skip("library not built (run __graft_entry__.build())")
    exe = _build(tmp_path)
    out = subprocess.run([exe, LIB, "check"], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok")
    # the C program's symbol list is the same list the Python loader binds
    listed = set(open(SRC).read().split('kSymbols[] = {')[1].split('};')[0].replace('"', '').replace('\n', '').replace(' ', '').split(','))
    assert listed == set(_lib.SYMBOLS)


@pytest.mark.gpu
def test_c_consumer_solve_equals_python_path(tmp_path):
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload
    n, h = 16, 10
    wl = make_solver_workload(n, h=h, seed=11, config=3)
    inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
    exe = _build(tmp_path)
    fin, fout = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<iidd", n, h, float(wl.dt_mpc), float(wl.alpha)))
        f.write(np.ascontiguousarray(wl.mass, np.float64).tobytes()); f.write(np.ascontiguousarray(inertia9, np.float64).tobytes())
        f.write(np.ascontiguousarray(wl.inputs, np.float32).tobytes())
    subprocess.run([exe, LIB, "solve", fin, fout], check=True)
    raw = open(fout, "rb").read()
    info_c = np.frombuffer(raw[:n * 8 * 4], dtype=np.int32).reshape(n, 8)
    f_c = np.frombuffer(raw[n * 8 * 4:], dtype=np.float64).reshape(n, 12 * h)
    gpu = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha)
    f_py, info_py = gpu.solve(torch.from_numpy(wl.inputs).cuda())
    torch.cuda.synchronize()
    assert (info_c[:, 1] == 1).all()
    np.testing.assert_array_equal(info_c, info_py.cpu().numpy())
    np.testing.assert_array_equal(f_c, f_py.cpu().numpy())


@pytest.mark.gpu
def test_cpp_consumer_with_the_reference_argument_types(tmp_path):
    """A C++ caller holding the reference's thirteen std::vector<double> per robot (mpc_osqp.cc:578-591) drives the float64 host entry
    point for a whole batch; results equal the Python path on the float32 record bit for bit (float32 -> float64 widening is exact)."""
    import torch
    import rl_mpc_locomotion_amd  # noqa: F401
    from rl_mpc_locomotion_amd.batched import BatchedConvexMpc
    from rl_mpc_locomotion_amd.synthetic import make_solver_workload
    n, h = 24, 10
    wl = make_solver_workload(n, h=h, seed=13, config=3)
    inertia9 = np.zeros((n, 9)); inertia9[:, 0], inertia9[:, 4], inertia9[:, 8] = wl.inertia_diag.T
    exe = os.path.join(str(tmp_path), "abi_cpp_consumer")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_c", "abi_cpp_consumer.cpp"), "-ldl", "-o", exe], check=True)
    fin, fout = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<iidd", n, h, float(wl.dt_mpc), float(wl.alpha)))
        f.write(np.ascontiguousarray(wl.mass, np.float64).tobytes()); f.write(np.ascontiguousarray(inertia9, np.float64).tobytes())
        f.write(np.ascontiguousarray(wl.inputs, np.float64).tobytes())
    subprocess.run([exe, LIB, fin, fout], check=True)
    raw = open(fout, "rb").read()
    info_c = np.frombuffer(raw[:n * 8 * 4], dtype=np.int32).reshape(n, 8)
    f_c = np.frombuffer(raw[n * 8 * 4:], dtype=np.float64).reshape(n, 12 * h)
    gpu = BatchedConvexMpc(wl.mass, inertia9, h, wl.dt_mpc, wl.alpha)
    f_py, info_py = gpu.solve(torch.from_numpy(wl.inputs).cuda())
    torch.cuda.synchronize()
    assert (info_c[:, 1] == 1).all()
    np.testing.assert_array_equal(info_c, info_py.cpu().numpy())
    np.testing.assert_array_equal(f_c, f_py.cpu().numpy())
