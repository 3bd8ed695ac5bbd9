"""How csrc/gelsd43.h was pinned to scipy.linalg.lstsq (LAPACK SGELSD in the OpenBLAS scipy ships), and how to re-pin it after a scipy / OpenBLAS change.

    python tools/gelsd43/pin.py            (needs scipy; the library's LAPACK / BLAS entry points are called through ctypes)

1. BLAS kernel facts: the summation order of sgemv('T') for the shapes SGELSD uses on a 4 x 3 matrix, of sgemm('T','N',3,1,3), of numpy's float32 dot / norm and of
   the (4,3) x (3,3)^T product of the CoM height, found by brute force over evaluation orders (every permutation, fused or not) against the library.
2. LAPACK level: SLARTG / SLAS2 / SLASV2 / SBDSQR of gelsd43.h against the library's own routines on random inputs, then the whole solve stage by stage
   (SGEQR2, SORM2R, SGEBD2, SORM2R, SLALSD) and end to end against scipy.linalg.lstsq.  Every count printed must be 0.
"""
import ctypes as C
import glob
import itertools
import os
import subprocess
import sys
import tempfile

import numpy as np
import scipy
from scipy import linalg

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
so = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas-*.so"))[0]
L = C.CDLL(so)
tmp = tempfile.mkdtemp()
subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "rl-mpc-locomotion_amd", "csrc"), os.path.join(ROOT, "tools", "gelsd43", "harness.cpp"),
                "-o", os.path.join(tmp, "libh.so")], check=True)
H = C.CDLL(os.path.join(tmp, "libh.so"))
f32 = np.float32
p = lambda a: a.ctypes.data_as(C.c_void_p)
ci = lambda v: C.byref(C.c_int(v))
cf = lambda v: C.byref(C.c_float(v))
rng = np.random.default_rng(1)
rnd = lambda *s: rng.uniform(-1, 1, s).astype(f32)
fma = lambda a, b, c: f32(np.float64(a) * np.float64(b) + np.float64(c))
core = L.scipy_openblas_get_corename; core.restype = C.c_char_p
print("scipy", scipy.__version__, "OpenBLAS kernels:", core().decode())
H.g_lartg.argtypes = [C.c_float, C.c_float, C.c_void_p]; H.g_las2.argtypes = [C.c_float] * 3 + [C.c_void_p]; H.g_lasv2.argtypes = [C.c_float] * 3 + [C.c_void_p]


# ---- 1. BLAS kernel facts -------------------------------------------------------------------------------------------------------------------------
def orders(m):
    return [(perm, mode) for perm in itertools.permutations(range(m)) for mode in itertools.product((0, 1), repeat=m - 1)]


def evaluate(c, a, x):
    perm, mode = c
    acc = f32(a[perm[0]] * x[perm[0]])
    for k, i in enumerate(perm[1:]):
        acc = fma(a[i], x[i], acc) if mode[k] else f32(acc + f32(a[i] * x[i]))
    return acc


def surviving(m, sample):
    cands = orders(m)
    alive = set(range(len(cands)))
    for _ in range(300):
        a, x, got = sample()
        alive = {k for k in alive if evaluate(cands[k], a, x) == got}
    return [cands[k] for k in sorted(alive)]


def sgemv_t(m, n, lda):
    def sample():
        A = np.asfortranarray(rnd(lda, n)); x = rnd(m); y = rnd(n)
        L.scipy_sgemv_(C.c_char_p(b"T"), ci(m), ci(n), cf(1.0), p(A), ci(lda), p(x), ci(1), cf(0.0), p(y), ci(1), C.c_long(1))
        return A[:m, n - 1], x, y[n - 1]
    return sample


print("sgemv('T') summation orders (permutation of the terms, 1 = fused with the running sum):")
for m, n in ((2, 1), (3, 1), (3, 2), (4, 1)):
    print("   m = %d, n = %d:" % (m, n), surviving(m, sgemv_t(m, n, 4)))
cnt = 0
for _ in range(2000):
    A = np.asfortranarray(rnd(4, 2)); x = rnd(4); y = rnd(2)
    L.scipy_sgemv_(C.c_char_p(b"T"), ci(4), ci(2), cf(1.0), p(A), ci(4), p(x), ci(1), cf(0.0), p(y), ci(1), C.c_long(1))
    pr = (A[:, 0] * x).astype(f32)
    cnt += y[0] != f32(f32(pr[0] + pr[1]) + f32(pr[2] + pr[3]))
print("   m = 4, n = 2: (p0 + p1) + (p2 + p3), unfused -- mismatches", cnt)


def sgemm_sample():
    W = np.asfortranarray(rnd(3, 3)); B = rnd(4); out = np.zeros(3, f32)
    L.scipy_sgemm_(C.c_char_p(b"T"), C.c_char_p(b"N"), ci(3), ci(1), ci(3), cf(1.0), p(W), ci(3), p(B), ci(4), cf(0.0), p(out), ci(3), C.c_long(1), C.c_long(1))
    return W[:, 0], B, out[0]


print("sgemm('T','N',3,1,3):", surviving(3, sgemm_sample))


def dot43_sample():
    fp = rng.uniform(-0.4, 0.4, (4, 3, 1)).astype(f32); gR = rng.uniform(-1, 1, (3, 3)).astype(np.float16)
    out = fp.reshape((4, 3)).dot(gR.T)
    return fp[1, :, 0], gR.astype(f32)[2], out[1, 2]


print("numpy (4,3) . (3,3)^T float32 (the CoM height, StateEstimator.py:113):", surviving(3, dot43_sample))
bad = 0
for _ in range(20000):
    x = rng.uniform(-4, 4, 3).astype(f32); pr = (x * x).astype(f32)
    bad += np.linalg.norm(x) != np.sqrt(f32(np.float64(pr[0]) + np.float64(pr[1]) + np.float64(pr[2])))
print("np.linalg.norm (float32, 3 entries) = sqrtf(float(sum in double of the float32 products)): mismatches", bad)
L.scipy_snrm2_.restype = C.c_float
bad = 0
for _ in range(20000):
    x = rnd(3)
    bad += f32(L.scipy_snrm2_(ci(3), p(x), ci(1))) != f32(np.sqrt(np.sum(x.astype(np.float64) ** 2)))
print("snrm2 = float(sqrt(sum of squares in double)): mismatches", bad)

# ---- 2. LAPACK level --------------------------------------------------------------------------------------------------------------------------------
bad = 0
for t in range(20000):
    f, g = rnd(2) * f32(10.0 ** rng.integers(-3, 3))
    if t % 50 == 0: g = f32(0)
    if t % 71 == 0: f = f32(0)
    o = np.zeros(3, f32); H.g_lartg(f, g, p(o))
    c, s, r = C.c_float(), C.c_float(), C.c_float()
    L.scipy_slartg_(cf(f), cf(g), C.byref(c), C.byref(s), C.byref(r))
    bad += not np.array_equal(o, np.array([c.value, s.value, r.value], f32))
print("SLARTG mismatches", bad)
bad = 0
for t in range(20000):
    f, g, h = rnd(3) * f32(10.0 ** rng.integers(-2, 2))
    o = np.zeros(2, f32); H.g_las2(f, g, h, p(o))
    a, b = C.c_float(), C.c_float()
    L.scipy_slas2_(cf(f), cf(g), cf(h), C.byref(a), C.byref(b))
    bad += not np.array_equal(o, np.array([a.value, b.value], f32))
print("SLAS2 mismatches", bad)
bad = 0
for t in range(20000):
    f, g, h = rnd(3) * f32(10.0 ** rng.integers(-2, 2))
    if t % 40 == 0: g = f32(0)
    o = np.zeros(6, f32); H.g_lasv2(f, g, h, p(o))
    v = [C.c_float() for _ in range(6)]
    L.scipy_slasv2_(cf(f), cf(g), cf(h), *[C.byref(x) for x in v])
    bad += not np.array_equal(o, np.array([x.value for x in v], f32))
print("SLASV2 mismatches", bad)
bad = 0
for t in range(50000):
    d = rnd(3); e = rnd(2); k = t % 10
    if k == 1: e *= f32(1e-4)
    if k == 2: d[2] *= f32(1e-5)
    if k == 3: e[0] = 0
    if k == 5: d[0] *= f32(1e-3)
    if k == 6: e[1] *= f32(1e-7)
    mx = max(np.abs(d).max(), np.abs(e).max()); d = (d / mx).astype(f32); e = (e / mx).astype(f32)
    cc = rnd(3)
    d1, e1, c1, vt1 = d.copy(), e.copy(), cc.copy(), np.eye(3, dtype=f32).flatten()
    d2, e2, c2, vt2 = d.copy(), e.copy(), cc.copy(), np.eye(3, dtype=f32).flatten()
    i1 = H.g_bdsqr3(p(d1), p(e1), p(vt1), p(c1))
    work = np.zeros(12, f32); info = C.c_int(0); u = np.zeros(1, f32)
    L.scipy_sbdsqr_(C.c_char_p(b"U"), ci(3), ci(3), ci(0), ci(1), p(d2), p(e2), p(vt2), ci(3), p(u), ci(1), p(c2), ci(3), p(work), C.byref(info), C.c_long(1))
    bad += not (np.array_equal(d1, d2) and np.array_equal(vt1, vt2) and np.array_equal(c1, c2) and i1 == info.value)
print("SBDSQR (n = 3, VT and one right-hand side) mismatches", bad)
cnt = {}
for t in range(3000):
    A = rnd(4, 3); Af = np.asfortranarray(A)
    x = np.zeros(3, f32); dbg = np.zeros(64, f32); H.g_solve_dbg(p(np.ascontiguousarray(A)), p(x), p(dbg))
    a = Af.copy(order="F"); tau = np.zeros(3, f32); work = np.zeros(64, f32); info = C.c_int()
    L.scipy_sgeqr2_(ci(4), ci(3), p(a), ci(4), p(tau), p(work), C.byref(info))
    b = np.ones(4, f32)
    L.scipy_sorm2r_(C.c_char_p(b"L"), C.c_char_p(b"T"), ci(4), ci(1), ci(3), p(a), ci(4), p(tau), p(b), ci(4), p(work), C.byref(info), C.c_long(1), C.c_long(1))
    ok = {"SGEQR2": np.array_equal(dbg[:12], a.flatten(order="F")) and np.array_equal(dbg[12:15], tau), "SORM2R (Q^T b)": np.array_equal(dbg[15:19], b)}
    a[1, 0] = a[2, 0] = a[2, 1] = 0
    d = np.zeros(3, f32); e = np.zeros(2, f32); tq = np.zeros(3, f32); tp = np.zeros(3, f32)
    L.scipy_sgebd2_(ci(3), ci(3), p(a), ci(4), p(d), p(e), p(tq), p(tp), p(work), C.byref(info))
    ok["SGEBD2"] = np.array_equal(dbg[20:32], a.flatten(order="F")) and np.array_equal(dbg[32:35], d) and np.array_equal(dbg[35:37], e) and np.array_equal(dbg[37:40], tq) and np.array_equal(dbg[40:43], tp)
    L.scipy_sorm2r_(C.c_char_p(b"L"), C.c_char_p(b"T"), ci(3), ci(1), ci(3), p(a), ci(4), p(tq), p(b), ci(4), p(work), C.byref(info), C.c_long(1), C.c_long(1))
    ok["SORM2R (Q_b^T b)"] = np.array_equal(dbg[43:46], b[:3])
    rank = C.c_int(); iwork = np.zeros(64, np.int32); work2 = np.zeros(256, f32)
    L.scipy_slalsd_(C.c_char_p(b"U"), ci(25), ci(3), ci(1), p(d), p(e), p(b), ci(4), cf(1.1920929e-07), C.byref(rank), p(work2), p(iwork), C.byref(info), C.c_long(1))
    ok["SLALSD"] = np.array_equal(dbg[46:49], b[:3])
    for k, v in ok.items():
        cnt[k] = cnt.get(k, 0) + int(not v)
print("stage by stage against the library's routines, mismatches:", cnt)
bad = 0
for t in range(50000):
    k = t % 8
    A = rnd(4, 3)
    if k == 1: A[:, 2] = f32(-0.3)
    if k == 2: A = (np.array([[0.24, 0.13, -0.3], [0.24, -0.13, -0.3], [-0.24, 0.13, -0.3], [-0.24, -0.13, -0.3]]) + 0.02 * rng.uniform(-1, 1, (4, 3))).astype(f32)
    x = np.zeros(3, f32); H.g_solve(p(np.ascontiguousarray(A)), p(x))
    bad += not np.array_equal(x, linalg.lstsq(A, np.ones(4, dtype=f32))[0])
print("end to end against scipy.linalg.lstsq, 50 000 matrices: mismatches", bad)
