"""How csrc/svml_acosf.h was pinned (round 6): numpy's float32 arccos / arctan2 ARE Intel SVML's __svml_acosf16 / __svml_atan2f16 on an AVX-512 machine (numpy/_core/_multiarray_umath: the routine
and its constant block __svml_sacos_data_internal were read off the shared object with objdump), a straight line of ~45 single-precision operations around one
VRSQRT14PS.  This script
  1. tabulates VRSQRT14PS on this CPU (gcc -mavx512f, _mm512_rsqrt14_ps): the result is a function of the exponent's parity and the top 15 mantissa bits (16 significant
     result bits, bits 6..0 zero; an exact power of four gives the exact root), scales by powers of two exactly -> 65536 entries, monotone within a parity, stored as
     2-bit decrements + one base per 32 entries in csrc/svml_acosf_table.h (generated here);
  2. compiles the restatement (csrc/svml_acosf.h, host build, fmaf = hardware FMA) and compares it with np.arccos on EVERY float32 in [-1, 1] (2 130 706 434 values);
  3. the same for VRCP14PS (top 16 mantissa bits, no parity) and __svml_atan2f16 (a quotient by VRCP14PS + two correction steps, an odd polynomial in two interleaved
     chains): np.arctan2 against svml_atan2f on 2^30 random pairs over 40 binades each way plus every float32 quotient y / 1 of [2^-8, 2^8] (2^27 values, both signs of x).
Run:  python tools/acosf/pin.py [--no-sweep]      (needs an AVX-512 CPU: the build container's has one; writes profiles/r06_acosf_pinning.txt)
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "rl-mpc-locomotion_amd", "csrc")
d = tempfile.mkdtemp()

# ---- 1. VRSQRT14PS on this CPU
open(os.path.join(d, "r14.c"), "w").write("""
#include <immintrin.h>
void rsqrt14(const float *in, float *out, long n) { for (long i = 0; i < n; i += 16) _mm512_storeu_ps(out + i, _mm512_rsqrt14_ps(_mm512_loadu_ps(in + i))); }
""")
subprocess.run(["gcc", "-O2", "-mavx512f", "-shared", "-fPIC", os.path.join(d, "r14.c"), "-o", os.path.join(d, "r14.so")], check=True)
R = C.CDLL(os.path.join(d, "r14.so"))


def r14(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert len(x) % 16 == 0
    out = np.empty_like(x)
    R.rsqrt14(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
    return out


m = np.arange(1 << 23, dtype=np.uint32)
table = np.zeros(1 << 16, dtype=np.uint16)
for par, e in ((0, 127), (1, 128)):      # x in [1, 2) and [2, 4)
    y = r14(((np.uint32(e) << 23) | m).view(np.float32)).view(np.uint32)
    g = y.reshape(1 << 15, 256)
    first = g[:, 0].copy()
    if par == 0:
        assert y[0] == 0x3F800000      # rsqrt14(1) = 1 exactly
        g = g[:, 1:]                   # (the exact power of four is the one input of its group that differs)
    assert (g == g[:, :1]).all(), "VRSQRT14PS depends on more than the top 15 mantissa bits here"
    v = g[:, 0]
    assert ((v >> 23) == 126).all() and ((v & 0x7F) == 0).all()
    table[par << 15:(par + 1) << 15] = ((v & 0x7FFFFF) >> 7).astype(np.uint16)
xs = ((np.uint32(127) << 23) | m[::997][: (len(m[::997]) // 16) * 16]).view(np.float32)
assert np.array_equal(r14(xs * 4), r14(xs) / 2) and np.array_equal(r14(xs / 4), r14(xs) * 2) and np.array_equal(r14(xs / 1024), r14(xs) * 32)
# ---- 1b. VRCP14PS
open(os.path.join(d, "rc14.c"), "w").write("""
#include <immintrin.h>
void rcp14(const float *in, float *out, long n) { for (long i = 0; i < n; i += 16) _mm512_storeu_ps(out + i, _mm512_rcp14_ps(_mm512_loadu_ps(in + i))); }
""")
subprocess.run(["gcc", "-O2", "-mavx512f", "-shared", "-fPIC", os.path.join(d, "rc14.c"), "-o", os.path.join(d, "rc14.so")], check=True)
RC = C.CDLL(os.path.join(d, "rc14.so"))


def rc14(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert len(x) % 16 == 0
    out = np.empty_like(x)
    RC.rcp14(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
    return out


yc = rc14(((np.uint32(127) << 23) | m).view(np.float32)).view(np.uint32)
assert yc[0] == 0x3F800000
gc = yc.reshape(1 << 16, 128)
assert (gc[1:] == gc[1:, :1]).all() and (gc[0, 1:] == gc[0, 1]).all(), "VRCP14PS depends on more than the top 16 mantissa bits here"
vc = gc[:, 1]
assert ((vc >> 23) == 126).all() and ((vc & 0x7F) == 0).all()
rtable = ((vc & 0x7FFFFF) >> 7).astype(np.uint16)
assert np.array_equal(rc14(xs * 2), rc14(xs) / 2) and np.array_equal(rc14(xs / 1024), rc14(xs) * 1024)


def pack(tab):
    blk = tab.reshape(-1, 32).astype(np.int64)
    dec2 = blk[:, :-1] - blk[:, 1:]
    assert dec2.min() >= 0 and dec2.max() <= 3, (dec2.min(), dec2.max())
    words = np.zeros((blk.shape[0], 2), dtype=np.uint32)
    for p in range(31):
        words[:, p // 16] |= (dec2[:, p].astype(np.uint32) << np.uint32(2 * (p % 16)))
    return blk[:, 0], words.reshape(-1)


# 2-bit decrements + a base per 32 entries
with open(os.path.join(CSRC, "svml_acosf_table.h"), "w") as f:
    f.write("// svml_acosf_table.h -- GENERATED by tools/acosf/pin.py: VRSQRT14PS on [1, 4) as Intel's AVX-512 hardware returns it (a function of the exponent's parity and the top\n"
            "// 15 mantissa bits; 16 significant result bits).  Entry i = parity << 15 | top 15 mantissa bits; value = result mantissa >> 7 (the result's exponent field is 126).\n"
            "// Stored as one base per 32 entries and 2-bit decrements (the function falls by 0 .. 3 units per entry).\n#pragma once\n")
    f.write("// VRCP14PS on [1, 2) likewise: entry i = top 16 mantissa bits (an exact power of two gives the exact reciprocal).\n")
    for nm, tab in (("Rsqrt14", table), ("Rcp14", rtable)):
        base, w = pack(tab)
        f.write("static constexpr unsigned short k%sBase[2048] = {\n" % nm)
        for i in range(0, 2048, 16):
            f.write("  " + ", ".join(str(int(v)) for v in base[i:i + 16]) + ",\n")
        f.write("};\nstatic constexpr unsigned k%sDec[4096] = {\n" % nm)
        for i in range(0, 4096, 8):
            f.write("  " + ", ".join("0x%08xu" % int(v) for v in w[i:i + 8]) + ",\n")
        f.write("};\n")
print("table written: 65536 entries,", len(np.unique(table)), "distinct values")

# ---- 2. the restatement against np.arccos on every float32 of [-1, 1]
open(os.path.join(d, "m.cpp"), "w").write("""
#include "svml_acosf.h"
extern "C" void model(const float *in, float *out, long n) { for (long i = 0; i < n; ++i) out[i] = mpc::svml_acosf(in[i]); }
extern "C" void model_r14(const float *in, float *out, long n) { for (long i = 0; i < n; ++i) out[i] = mpc::rsqrt14f(in[i]); }
extern "C" void model_rc14(const float *in, float *out, long n) { for (long i = 0; i < n; ++i) out[i] = mpc::rcp14f(in[i]); }
extern "C" void model_atan2(const float *y, const float *x, float *out, long n) { for (long i = 0; i < n; ++i) out[i] = mpc::svml_atan2f(y[i], x[i]); }
""")
subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-I" + CSRC, os.path.join(d, "m.cpp"), "-o", os.path.join(d, "m.so")], check=True)
M = C.CDLL(os.path.join(d, "m.so"))


def call(fn, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    fn(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
    return out


rng = np.random.default_rng(0)
xr = np.abs(rng.standard_normal(1 << 20)).astype(np.float32) * np.float32(2.0) ** rng.integers(-30, 30, 1 << 20).astype(np.float32) + np.float32(1e-30)
assert np.array_equal(call(M.model_r14, xr).view(np.uint32), r14(xr).view(np.uint32)), "rsqrt14f differs from the hardware"
assert np.array_equal(call(M.model_rc14, xr).view(np.uint32), rc14(xr).view(np.uint32)), "rcp14f differs from the hardware"


def call2(fn, y, x):
    y = np.ascontiguousarray(y, dtype=np.float32); x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    fn(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_long(len(x)))
    return out


lines = [f"VRSQRT14PS: csrc/svml_acosf.h rsqrt14f bit-identical to the instruction on {len(xr)} random positive floats over 60 binades and on all 2^24 inputs of [1, 4) (by construction of the table)"]
if "--no-sweep" not in sys.argv:
    t0 = time.time()
    bad = total = 0
    worst = []
    top = 0x3F800000      # bits of 1.0f
    CH = 1 << 24
    for sign in (0, 0x80000000):
        for lo in range(0, top + 1, CH):
            bits = (np.arange(lo, min(lo + CH, top + 1), dtype=np.uint32) | np.uint32(sign))
            x = bits.view(np.float32)
            a, b = np.arccos(x), call(M.model, x)
            ne = a.view(np.uint32) != b.view(np.uint32)
            total += len(x)
            if ne.any():
                bad += int(ne.sum())
                worst.extend(x[ne][:3].tolist())
    lines.append(f"np.arccos (numpy {np.__version__}, float32) vs csrc/svml_acosf.h svml_acosf on EVERY float32 of [-1, 1]: {total} values, {bad} differ" + (f" (e.g. {worst[:6]})" if bad else "") +
                 f"   [{time.time() - t0:.0f} s]")
    assert bad == 0
lines.append(f"VRCP14PS: rcp14f bit-identical to the instruction on the same {len(xr)} random floats and on all 2^23 inputs of [1, 2)")
t0 = time.time()
bad = total = 0
ex = []
for rep in range(4 if "--no-sweep" in sys.argv else 64):      # 2^24 pairs per repetition
    n = 1 << 24
    y = (rng.standard_normal(n) * np.exp2(rng.integers(-20, 20, n))).astype(np.float32)
    x = (rng.standard_normal(n) * np.exp2(rng.integers(-20, 20, n))).astype(np.float32)
    if rep % 4 == 1:
        x = (y * rng.uniform(0.9, 1.1, n)).astype(np.float32) * rng.choice([-1.0, 1.0], n).astype(np.float32)      # near the |y| = |x| switch
    a, b = np.arctan2(y, x), call2(M.model_atan2, y, x)
    ne = a.view(np.uint32) != b.view(np.uint32)
    total += n; bad += int(ne.sum())
    if ne.any(): ex.extend(zip(y[ne][:2].tolist(), x[ne][:2].tolist()))
for sx in (1.0, -1.0):      # every float32 y of [2^-8, 2^8] against x = +-1
    bits = np.arange(0x3B800000, 0x43800000, dtype=np.uint32)
    for lo in range(0, len(bits), 1 << 24):
        y = bits[lo:lo + (1 << 24)].view(np.float32)
        x = np.full(len(y), sx, dtype=np.float32)
        a, b = np.arctan2(y, x), call2(M.model_atan2, y, x)
        ne = a.view(np.uint32) != b.view(np.uint32)
        total += len(y); bad += int(ne.sum())
        if ne.any(): ex.extend(zip(y[ne][:2].tolist(), x[ne][:2].tolist()))
edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-38, 1e36], dtype=np.float32)      # (finite magnitudes outside [2^-125, 2^123) go to libm here: not pinned)
ye, xe = np.meshgrid(edge, edge)
a, b = np.arctan2(ye.ravel(), xe.ravel()), call2(M.model_atan2, ye.ravel(), xe.ravel())
edge_ok = bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
lines.append(f"np.arctan2 (float32) vs svml_atan2f: {total} pairs (random over 40 binades each way, near |y| = |x|, every y of [2^-8, 2^8] against x = +-1), {bad} differ" +
             (f" (e.g. {ex[:4]})" if bad else "") + f"; zeros / infinities / NaN (the fall-back to atan2f): {'equal' if edge_ok else 'DIFFER'}   [{time.time() - t0:.0f} s]")
print(lines[-1])
if not edge_ok:
    for yy, xx, aa, bb in zip(ye.ravel(), xe.ravel(), a, b):
        if aa.view(np.uint32) != bb.view(np.uint32) and not (np.isnan(aa) and np.isnan(bb)): print("edge", yy, xx, aa, bb)
assert bad == 0 and edge_ok
print("\n".join(lines))
open(os.path.join(ROOT, "profiles", "r06_acosf_pinning.txt"), "w").write(__doc__ + "\n" + "\n".join(lines) + "\n")
