"""csrc/gelsd43.h -- the ground-normal least squares in LAPACK SGELSD's own arithmetic -- against the reference's call
(scipy.linalg.lstsq on float32, StateEstimator.py:130): bit for bit, on committed known-answer vectors and, where the scipy at hand is
the build the vectors were minted with, on fresh random matrices.  (The fit inside controller.run -- history update, this solve, the two
normalisations -- is held bit-identical to the reference's ground_normal_yaw on every tick of every golden by
tests/test_controller.py::test_emulated_full_run_matches_reference_python and its -m gpu twin.)"""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import load_golden


def _solve(A):
    from tests.emu.emu import lib
    A = np.ascontiguousarray(A, dtype=np.float32)
    n = A.shape[0]
    x = np.zeros((n, 3), np.float32); rank = np.zeros(n, np.int32)
    lib().emu_gelsd43.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib().emu_gelsd43(n, A.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), rank.ctypes.data_as(C.c_void_p))
    return x, rank


def test_gelsd43_matches_the_known_answer_vectors():
    g = load_golden("gelsd43_vectors")
    x, rank = _solve(g["A"])
    assert np.array_equal(x, g["x"])
    assert (rank == 3).all()


def test_gelsd43_matches_scipy_live():
    linalg = pytest.importorskip("scipy.linalg")
    from tests.golden.make_golden_gelsd43 import corename, matrices
    g = load_golden("gelsd43_vectors")
    if corename() != str(g["core"]):
        pytest.skip(f"scipy's OpenBLAS runs its {corename()} kernels here; the reference's arithmetic was pinned on {g['core']}")
    A = matrices(6000, seed=123)
    x, _ = _solve(A)
    want = np.stack([linalg.lstsq(a, np.ones(4, dtype=np.float32))[0] for a in A])
    assert np.array_equal(x, want)


@pytest.mark.gpu
def test_gelsd43_on_the_device_matches_the_known_answers():
    """The same header compiled for gfx950 (hipcc, -ffp-contract=off as csrc/Makefile compiles the controller) and run on the GPU over the
    known-answer matrices: bit-identical to the reference's answers, with and without the stage records (tests/device/run_gelsd43_device.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "device", "run_gelsd43_device.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "device vs reference answers: 0 of" in out.stdout, out.stdout
    assert "with / without the stage record: 0 entries differ" in out.stdout, out.stdout
