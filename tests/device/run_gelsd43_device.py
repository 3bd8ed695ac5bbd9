"""Builds tests/device/gelsd43_device.hip with hipcc, runs it on the GPU over the known-answer matrices and compares with the reference's answers
and, stage by stage, with the host build of the same header (tests/emu)."""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
g = np.load(os.path.join(ROOT, "tests", "golden", "gelsd43_vectors.npz"))
A, want = np.ascontiguousarray(g["A"]), g["x"]
n = len(A)
d = tempfile.mkdtemp()
exe = os.path.join(d, "gelsd43_device")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "rl-mpc-locomotion_amd", "csrc"),
                os.path.join(ROOT, "tests", "device", "gelsd43_device.hip"), "-o", exe] + sys.argv[1:], check=True)
A.tofile(os.path.join(d, "A.bin"))
print(subprocess.run([exe, os.path.join(d, "A.bin"), str(n), os.path.join(d, "x.bin"), os.path.join(d, "dbg.bin")], capture_output=True, text=True).stdout.strip())
x = np.fromfile(os.path.join(d, "x.bin"), np.float32).reshape(n, 3)
dbg = np.fromfile(os.path.join(d, "dbg.bin"), np.float32).reshape(n, 64)
bad = (x != want).any(1)
print("device vs reference answers: %d of %d matrices differ" % (bad.sum(), n))
# host stage records
subprocess.run(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "rl-mpc-locomotion_amd", "csrc"), "-x", "c++", "-", "-o", os.path.join(d, "h.so")], check=True, text=True,
               input='#include "gelsd43.h"\nextern "C" void hs(int n, const float *A, float *x, float *dbg) { for (int i = 0; i < n; ++i) mpc::gelsd43::solve_ones(A + 12 * i, x + 3 * i, dbg + 64 * i); }\n')
H = C.CDLL(os.path.join(d, "h.so"))
hx = np.zeros((n, 3), np.float32); hd = np.zeros((n, 64), np.float32)
H.hs(n, A.ctypes.data_as(C.c_void_p), hx.ctypes.data_as(C.c_void_p), hd.ctypes.data_as(C.c_void_p))
print("host vs reference answers: %d differ" % (hx != want).any(1).sum())
stages = {"qr a": (0, 12), "qr tau": (12, 15), "Q^T b": (15, 19), "bd a": (20, 32), "bd d": (32, 35), "bd e": (35, 37), "tauq": (37, 40), "taup": (40, 43), "Qb^T b": (43, 46), "lalsd": (46, 49)}
for k, (a, b) in stages.items():
    m = (dbg[:, a:b] != hd[:, a:b]).any(1)
    print("  stage %-7s device != host on %d matrices" % (k, m.sum()), ("first: %s vs %s" % (dbg[m][0, a:b], hd[m][0, a:b])) if m.any() else "")
