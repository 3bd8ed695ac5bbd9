// svml_acosf.h -- numpy's float32 arccos, bit for bit.  The reference's state estimator calls np.arccos on a float32 (MPC_Controller/math_utils/orientation_tools.py:94,
// from StateEstimator.py:88-92); numpy 2.x on an AVX-512 machine -- the build container that minted every golden -- runs Intel SVML's __svml_acosf16 for it, up to 2 ulp
// from libm's / OCML's acosf, and the angle's cos / sin are rounded to float16 right after, so the difference showed in ~4 of 10 000 estimator samples (rounds 4-5 counted
// them and let the robot leave the comparison).  This is that routine restated operation for operation (read off numpy's shared object; tools/acosf/pin.py says how and
// checks it against np.arccos on EVERY float32 of [-1, 1]): |x| < 1/2: pi/2 - (x + x R P(R)), R = x^2; else 2 sqrt(y) (1 + R P(R)), R = y = (1 - |x|) / 2, reflected for
// x < 0 -- with sqrt(y) from the VRSQRT14PS instruction (architecturally defined: svml_acosf_table.h) and one correction step.  Arguments outside [-1, 1] / NaN take SVML's
// scalar fall-back there; here they return NaN (the estimator's argument is a dot product of unit vectors the reference never checks either).
// Compile with -ffp-contract=off: fused multiply-adds only where written as fmaf.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "mpc_core.h"
#include "svml_acosf_table.h"

namespace mpc {

MPC_HD uint32_t f32_bits(float v) { uint32_t b; memcpy(&b, &v, 4); return b; }
MPC_HD float bits_f32(uint32_t b) { float v; memcpy(&v, &b, 4); return v; }

MPC_HD uint32_t svml_tab16(const unsigned short *base, const unsigned *decw, uint32_t idx);
// VRSQRT14PS for a positive normal float
MPC_HD float rsqrt14f(float x) {
  const uint32_t b = f32_bits(x), m = b & 0x7FFFFFu;
  const int E = (int)(b >> 23) - 127, par = E & 1, q = (E - par) / 2;      // x = 4^q * [1, 4)
  if (par == 0 && m == 0) return bits_f32((uint32_t)(127 - q) << 23);     // an exact power of four: the exact root
  return bits_f32(((uint32_t)(126 - q) << 23) | (svml_tab16(kRsqrt14Base, kRsqrt14Dec, ((uint32_t)par << 15) | (m >> 8)) << 7));
}

// 2-bit-decrement table look-up shared by the two instructions
MPC_HD uint32_t svml_tab16(const unsigned short *base, const unsigned *decw, uint32_t idx) {
  const uint32_t blk = idx >> 5, pos = idx & 31;
  const unsigned long long w = (unsigned long long)decw[2 * blk] | ((unsigned long long)decw[2 * blk + 1] << 32);
  const unsigned long long wm = pos ? (w & (~0ull >> (64 - 2 * pos))) : 0ull;
  return (uint32_t)base[blk] - (uint32_t)(__builtin_popcountll(wm & 0x5555555555555555ull) + 2 * __builtin_popcountll(wm & 0xAAAAAAAAAAAAAAAAull));
}

// VRCP14PS for a positive normal float whose reciprocal is normal
MPC_HD float rcp14f(float x) {
  const uint32_t b = f32_bits(x), m = b & 0x7FFFFFu;
  const int E = (int)(b >> 23) - 127;
  if (m == 0) return bits_f32((uint32_t)(127 - E) << 23);      // an exact power of two: the exact reciprocal
  return bits_f32(((uint32_t)(126 - E) << 23) | (svml_tab16(kRcp14Base, kRcp14Dec, m >> 7) << 7));
}

// numpy's float32 arctan2 = SVML's __svml_atan2f16 (quat_to_rpy's yaw, orientation_tools.py:120-133): q = min / max of |y|, |x| by VRCP14PS and two correction steps,
// atan as an odd polynomial (two interleaved chains in q^4), pi / 2 - . when |y| >= |x|, reflected for x <= 0, y's sign.  Zeros, infinities, NaN and magnitudes outside
// [2^-125, 2^123) take SVML's scalar path there and libm's atan2f here (the same values for zeros and infinities; tools/acosf/pin.py checks them).
MPC_HD float svml_atan2f(float y, float x) {
  const uint32_t xb = f32_bits(x), yb = f32_bits(y), ax = xb & 0x7FFFFFFFu, ay = yb & 0x7FFFFFFFu;
  if (ax - 0x01000000u >= 0x7C000000u || ay - 0x01000000u >= 0x7C000000u) return atan2f(y, x);
  const float fax = bits_f32(ax), fay = bits_f32(ay);
  const bool lt = fay < fax;
  const float num = lt ? fay : -fax, den = lt ? fax : fay, base = lt ? 0.0f : bits_f32(0x3FC90FDBu);
  const float r0 = rcp14f(den);
  const float e0 = fmaf(-r0, den, 1.0f);
  const float r1 = fmaf(e0, r0, r0);
  const float q0 = num * r1;
  const float rem = fmaf(-q0, den, num);
  const float q = fmaf(rem, r1, q0);
  const float s = q * q, s2 = s * s;
  float A = fmaf(bits_f32(0x3B322CC0u), s2, bits_f32(0x3D2BC384u));
  float B = fmaf(bits_f32(0xBC7F2631u), s2, bits_f32(0xBD987629u));
  A = fmaf(s2, A, bits_f32(0x3DD96474u));
  B = fmaf(s2, B, bits_f32(0xBE1161F8u));
  A = fmaf(s2, A, bits_f32(0x3E4CB79Fu));
  B = fmaf(s2, B, bits_f32(0xBEAAAA49u));
  A = fmaf(s2, A, 1.0f);
  const float P = fmaf(s, B, A);
  float res = fmaf(q, P, base);
  res = bits_f32(f32_bits(res) | (xb & 0x80000000u));
  if (x <= 0.0f) res = res + bits_f32(0x40490FDBu);
  return bits_f32(f32_bits(res) | (yb & 0x80000000u));
}

MPC_HD float svml_acosf(float x) {
  const uint32_t xb = f32_bits(x), sgn = xb & 0x80000000u;
  const float nax = bits_f32(xb | 0x80000000u);                 // -|x|
  if (!(nax >= -1.0f)) return bits_f32(0x7FC00000u);            // |x| > 1 or NaN (SVML's rare path)
  const float y = fmaf(0.5f, nax, 0.5f);                        // (1 - |x|) / 2
  const float x2 = nax * nax;
  const float r = y < bits_f32(0x2F800000u) ? 0.0f : rsqrt14f(y);
  const float R = x2 < y ? x2 : y;                              // (VMINPS)
  const float y2 = y + y, R2 = R * R, r2 = r * r, S = y2 * r;
  const bool big = !(R < y), neg = x < R;
  const float E = fmaf(r2, y2, -2.0f);
  const float pA = fmaf(bits_f32(0x3D3A9AB4u), R, bits_f32(0x3D997C12u));
  const float sc = fmaf(bits_f32(0xBDC00004u), E, bits_f32(0x3E800001u));
  const float pB = fmaf(bits_f32(0x3D2EDC07u), R, bits_f32(0x3CC32A6Bu));
  const float SE = S * E;
  float p = fmaf(R2, pB, pA);
  const float Sq = fmaf(-SE, sc, S);
  p = fmaf(R, p, bits_f32(0x3E2AAAFFu));
  const float RP = p * R;
  const float t = bits_f32(f32_bits(big ? Sq : nax) ^ sgn);
  const float tail = fmaf(t, RP, t);
  const float base = big ? (neg ? bits_f32(0x40490FDBu) : 0.0f) : bits_f32(0x3FC90FDBu);
  return base + tail;
}

}  // namespace mpc
