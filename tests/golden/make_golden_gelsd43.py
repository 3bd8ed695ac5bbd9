"""Known-answer vectors for csrc/gelsd43.h: scipy.linalg.lstsq(A, ones(4)) on float32 4 x 3 matrices -- the call of
MPC_Controller/common/StateEstimator.py:130 -- executed here by the scipy the reference runs on (scipy 1.15.3, OpenBLAS 0.3.28,
SkylakeX kernels; the core name is stored with the vectors).

    python tests/golden/make_golden_gelsd43.py

Matrix families: uniform random; a constant third column (the first-run contact history: every foot at -body_height); foot layouts of a
standing quadruped with small perturbations; columns scaled apart by 1e3; nearly rank-deficient layouts (four feet almost on a line).
"""
import ctypes
import glob
import os

import numpy as np
import scipy
from scipy import linalg

HERE = os.path.dirname(os.path.abspath(__file__))


def matrices(n, seed=0):
    rng = np.random.default_rng(seed)
    stand = np.array([[0.24, 0.13, -0.3], [0.24, -0.13, -0.3], [-0.24, 0.13, -0.3], [-0.24, -0.13, -0.3]])
    out = np.zeros((n, 4, 3), np.float32)
    for t in range(n):
        k = t % 6
        A = rng.uniform(-1, 1, (4, 3))
        if k == 1:
            A[:, 2] = -rng.uniform(0.2, 0.4)
        elif k == 2:
            A = stand + 0.02 * rng.uniform(-1, 1, (4, 3))
        elif k == 3:
            A = stand.copy(); A[:, :2] += 0.05 * rng.uniform(-1, 1, (4, 2))
        elif k == 4:
            A = A * np.array([1e2, 1.0, 1e-1])
        elif k == 5:
            A = np.outer(rng.uniform(-1, 1, 4), rng.uniform(-1, 1, 3)) + np.array([0, 0, -0.3]) + 1e-3 * rng.uniform(-1, 1, (4, 3))
        out[t] = A.astype(np.float32)
    return out


def corename():
    so = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas-*.so"))
    if not so:
        return "unknown"
    f = ctypes.CDLL(so[0]).scipy_openblas_get_corename
    f.restype = ctypes.c_char_p
    return f().decode()


if __name__ == "__main__":
    A = matrices(3000)
    x = np.stack([linalg.lstsq(a, np.ones(4, dtype=np.float32))[0] for a in A])
    assert x.dtype == np.float32
    np.savez_compressed(os.path.join(HERE, "gelsd43_vectors.npz"), A=A, x=x, scipy=scipy.__version__, core=corename())
    print("gelsd43_vectors written:", A.shape, "core", corename())
