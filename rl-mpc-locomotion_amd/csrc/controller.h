// controller.h -- the per-tick controller around the contact-force solve: one robot per call on the host and in the FSM kernels,
// one leg per lane in ctrl_pre_kernel / ctrl_post_kernel (the functions take a leg range).
//
// Restates, in the reference's own arithmetic types (numpy float32 "f", Python float "d"), what
//   MPC_Controller/common/LegController.py:89-106,135-171   updateData (leg FK, Jacobian, foot velocity)
//   MPC_Controller/convex_MPC/ConvexMPCLocomotion.py:222-378 run (gait, estimator sub-steps, foot placement,
//                                                            MPC marshalling, swing / stance leg commands)
//   MPC_Controller/convex_MPC/Gait.py:26-93                   OffsetDurationGait
//   MPC_Controller/common/FootSwingTrajectory.py:54-70, math_utils/interplation.py:4-26   swing Bezier
//   MPC_Controller/common/StateEstimator.py:99-143            contact history, CoM height, ground normal (gelsd43.h)
//   MPC_Controller/common/LegController.py:108-132            updateCommand (12 joint torques)
// do for one robot and one control tick.  Split in two halves around the solver launch:
//   ctrl_pre  : everything up to the call of compute_contact_forces (writes the solver input record)
//   ctrl_post : f_ff <- first-step forces, leg commands, torques
// StateEstimator.update (quaternion -> body-frame velocities, float16 rpy) runs before this stage; its
// outputs arrive in the `est` record.
#pragma once

#include <math.h>

#include "mpc_core.h"
#include "gelsd43.h"
#include "svml_acosf.h"

namespace mpc {

constexpr int kEstLen = 18;     // vBody[3] omegaBody[3] rpyBody[3] ground_R_body_frame[9] (fp16-valued)
constexpr int kNumGaitIds = 8;

struct RobotConst {   // MPC_Controller/common/Quadruped.py:16-92 (one row per robot type)
  double abad, hip, knee;        // link lengths (Python floats)
  float hiploc[3];               // _abadLocation (float32)
  double body_height;
  float mu;
  float weights[13];
};

struct GaitTable {    // ConvexMPCLocomotion.py:30-56 rescaled to n_seg segments (gait.py)
  int n_seg;
  float offsets[kNumGaitIds][4], durations[kNumGaitIds][4];
};

struct CtrlParams {
  double dt;                     // Parameters.controller_dt                       (Parameters.py:44)
  int iters_between_mpc;         // int(27 / (1000 dt)) = 2                        (RobotRunnerMin.py:21-22)
  double dt_mpc;                 // dt * iterationsBetweenMPC                      (ConvexMPCLocomotion.py:58)
  int horizon;
  int flat_ground;               // Parameters.flat_ground                          (Parameters.py:21)
};

struct CtrlState {    // per-robot persistent controller state
  int iter, first_run, first_swing[4];
  int gait_id, robot_type;
  double swing_time_remaining[4];
  float swing_times[4];
  float f_ff[12];
  float p0[12], pf[12], tp[12], tv[12];          // FootSwingTrajectory _p0 _pf _p _v per foot
  float pos_z, normal[3], contact_phase[4], hist[12];
  // scratch carried from ctrl_pre to ctrl_post within one tick
  float q[12], qd[12], p[12], v[12], J[36];
  float foot_positions[12], pfoot[12];
  float contact_states[4], swing_states[4];
  float vbody[3], posz_tick;
  int do_solve;
};

// RobotRunnerMin.reset (RobotRunnerMin.py:49-52): cMPC.initialize (ConvexMPCLocomotion.py:89-114) resets
// the counter and the firstRun / firstSwing flags, StateEstimator.reset (:41-48) the estimate.  f_ff and the
// swing trajectories' last p / v are NOT touched by the reference's reset -- kept here as well.
MPC_HD void ctrl_reset(CtrlState &s, const RobotConst &rc) {
  s.iter = 0; s.first_run = 1;
  for (int i = 0; i < 4; ++i) { s.first_swing[i] = 1; s.contact_phase[i] = 0.f; }
  for (int i = 0; i < 12; ++i) s.hist[i] = 0.f;
  s.pos_z = (float)rc.body_height;                             // StateEstimator.py:39
  s.normal[0] = 0.f; s.normal[1] = 0.f; s.normal[2] = 1.f;     // StateEstimator.py:21-22
  s.do_solve = 0;
}
// RobotRunnerMin.init: fresh objects (zeros everywhere)
MPC_HD void ctrl_init(CtrlState &s, const RobotConst &rc, int robot_type, int gait_id) {
  for (int i = 0; i < 4; ++i) { s.swing_time_remaining[i] = 0.0; s.swing_times[i] = 0.f; s.contact_states[i] = s.swing_states[i] = 0.f; }
  for (int i = 0; i < 12; ++i) { s.f_ff[i] = 0.f; s.p0[i] = s.pf[i] = s.tp[i] = s.tv[i] = 0.f; s.q[i] = s.qd[i] = s.p[i] = s.v[i] = 0.f; s.foot_positions[i] = s.pfoot[i] = 0.f; }
  for (int i = 0; i < 36; ++i) s.J[i] = 0.f;
  s.vbody[0] = s.vbody[1] = s.vbody[2] = 0.f; s.posz_tick = 0.f;
  s.gait_id = gait_id; s.robot_type = robot_type;
  ctrl_reset(s, rc);
}

// LegController.computeLegJacobianAndPosition (LegController.py:135-171): Python-float math, float32 storage.
MPC_HD void leg_kinematics(const RobotConst &rc, int leg, const float *q, float *p, float *J) {
  const double side = (leg == 0 || leg == 2) ? 1.0 : -1.0;      // utils.py:7 SIDE_SIGN
  const double dy = rc.abad * side, dz1 = -rc.hip, dz2 = -rc.knee;
  const double s1 = sin((double)q[0]), s2 = sin((double)q[1]), s3 = sin((double)q[2]);
  const double c1 = cos((double)q[0]), c2 = cos((double)q[1]), c3 = cos((double)q[2]);
  const double c23 = c2 * c3 - s2 * s3, s23 = s2 * c3 + c2 * s3;
  p[0] = (float)(dz2 * s23 + dz1 * s2);
  p[1] = (float)(dy * c1 - dz1 * c2 * s1 - dz2 * s1 * c23);
  p[2] = (float)(dy * s1 + dz1 * c1 * c2 + dz2 * c1 * c23);
  J[0] = 0.f;
  J[3] = (float)(-dy * s1 - dz2 * c1 * c23 - dz1 * c1 * c2);
  J[6] = (float)(-dz2 * s1 * c23 + dy * c1 - dz1 * c2 * s1);
  J[1] = (float)(dz2 * c23 + dz1 * c2);
  J[4] = (float)(dz2 * s1 * s23 + dz1 * s1 * s2);
  J[7] = (float)(-dz2 * c1 * s23 - dz1 * c1 * s2);
  J[2] = (float)(dz2 * c23);
  J[5] = (float)(dz2 * s1 * s23);
  J[8] = (float)(-dz2 * c1 * s23);
}

MPC_HD float round_to_half(float x) {   // numpy float16 rounding (round-to-nearest-even), returned as float
#if defined(__HIP_DEVICE_COMPILE__)
  return (float)(_Float16)x;
#else
  union { float f; unsigned u; } c;
  c.f = x;
  const unsigned ex = (c.u >> 23) & 0xffu;
  if (ex == 255u) return x;
  const int e = (int)ex - 127;
  if (e >= -14) {                       // normal half (or overflow): round the 13 dropped mantissa bits to nearest even
    c.u += 0xFFFu + ((c.u >> 13) & 1u);
    c.u &= ~0x1FFFu;
    if (fabsf(c.f) > 65504.f) return c.f > 0 ? INFINITY : -INFINITY;
    return c.f;
  }
  return nearbyintf(x * 16777216.f) / 16777216.f;   // subnormal half: multiples of 2^-24
#endif
}

MPC_HD void hip_location(const RobotConst &rc, int leg, float *h) {   // Quadruped.getHipLocation (Quadruped.py:96-107)
  h[0] = (leg == 0 || leg == 1) ? rc.hiploc[0] : -rc.hiploc[0];
  h[1] = (leg == 0 || leg == 2) ? rc.hiploc[1] : -rc.hiploc[1];
  h[2] = rc.hiploc[2];
}


// ================================================================================================
// StateEstimator.update (MPC_Controller/common/StateEstimator.py:57-97) with the arithmetic types the
// reference's numpy code really uses (numpy 2, NEP 50): the quaternion fields are np.float32 scalars, so
// quat_to_rot / quat_to_rpy (math_utils/orientation_tools.py:120-149) evaluate in float32 and store
// float16; rot_to_quat (:159-196) mixes Python floats with np.float16 scalars, which demotes every
// mixed operation to float16.  A float16 value is carried as a float here; hadd/hsub/hmul/hdiv round each
// operation to half like numpy's half loops (compute in float, round to half).
// ================================================================================================
MPC_HD float round_to_half_d(double d) {   // Python float -> np.float16 (one rounding, ties to even)
  if (!(d == d) || d == 0.0) return (float)d;
  const double ad = fabs(d);
  if (ad >= 65520.0) return d > 0 ? INFINITY : -INFINITY;
  int e = ilogb(ad);
  if (e < -14) e = -14;                       // subnormal halfs share the quantum 2^-24
  const double scale = ldexp(1.0, 10 - e);
  return (float)(nearbyint(d * scale) / scale);
}
MPC_HD float hadd(float a, float b) { return round_to_half(a + b); }
MPC_HD float hsub(float a, float b) { return round_to_half(a - b); }
MPC_HD float hmul(float a, float b) { return round_to_half(a * b); }
MPC_HD float hdiv(float a, float b) { return round_to_half(a / b); }

struct PyOrHalf {   // a quaternion component that is either a Python float or an np.float16 scalar
  double py; float h; bool is_py;
};
MPC_HD float as_half(const PyOrHalf &v) { return v.is_py ? round_to_half_d(v.py) : v.h; }
MPC_HD PyOrHalf qmul(const PyOrHalf &a, const PyOrHalf &b) {
  PyOrHalf r;
  if (a.is_py && b.is_py) { r.is_py = true; r.py = a.py * b.py; r.h = 0.f; }
  else { r.is_py = false; r.py = 0.0; r.h = hmul(as_half(a), as_half(b)); }
  return r;
}
MPC_HD float qadd(const PyOrHalf &a, const PyOrHalf &b) { return hadd(as_half(a), as_half(b)); }   // at most one side is Python
MPC_HD float qsub(const PyOrHalf &a, const PyOrHalf &b) { return hsub(as_half(a), as_half(b)); }
MPC_HD PyOrHalf from_half(float h) { PyOrHalf r; r.is_py = false; r.py = 0.0; r.h = h; return r; }

// 3x3 float16 matmul: float accumulation, one rounding to half (numpy HALF matmul)
MPC_HD void hmatmul(const float *A, const float *B, float *C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = round_to_half((A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j]);
}

// body: 13 floats (pos3, quat xyzw, lin vel world 3, ang vel world 3); normal: ground_normal_yaw of the
// previous tick; est out: vBody[3], omegaBody[3], rpyBody[3] (half valued), ground_R_body_frame[9].
MPC_HD void estimator_update(const float *body, const float *normal, float *est) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float x = body[3], y = body[4], z = body[5], w = body[6];
  const float e0 = w, e1 = x, e2 = y, e3 = z;
  // quat_to_rot (orientation_tools.py:135-149): listed row-major, then transposed
  float Ml[9];
  Ml[0] = round_to_half(1.f - 2.f * (e2 * e2 + e3 * e3)); Ml[1] = round_to_half(2.f * (e1 * e2 - e0 * e3)); Ml[2] = round_to_half(2.f * (e1 * e3 + e0 * e2));
  Ml[3] = round_to_half(2.f * (e1 * e2 + e0 * e3)); Ml[4] = round_to_half(1.f - 2.f * (e1 * e1 + e3 * e3)); Ml[5] = round_to_half(2.f * (e2 * e3 - e0 * e1));
  Ml[6] = round_to_half(2.f * (e1 * e3 - e0 * e2)); Ml[7] = round_to_half(2.f * (e2 * e3 + e0 * e1)); Ml[8] = round_to_half(1.f - 2.f * (e1 * e1 + e2 * e2));
  float R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[3 * i + j] = Ml[3 * j + i];
  // float16 matrix @ float32 vector -> float32: numpy hands this to OpenBLAS (sgemv on the C-ordered float32 copy), whose
  // kernel takes the rows in pairs -- rows 0 and 1 vectorised, products and sums rounded one by one -- and the odd last row in a
  // scalar tail compiled with FMA contraction: fma(a2, b2, fma(a0, b0, a1 * b1)).  Probed with crafted inputs
  // (tests/test_controller.py::test_estimator_matmul_rule_matches_numpy) and bit-exact on every recorded sample of the goldens.
  for (int i = 0; i < 3; ++i) {
    if (i < 2) {
      est[i] = (R[3 * i] * body[7] + R[3 * i + 1] * body[8]) + R[3 * i + 2] * body[9];
      est[3 + i] = (R[3 * i] * body[10] + R[3 * i + 1] * body[11]) + R[3 * i + 2] * body[12];
    } else {
      est[i] = fmaf(R[3 * i + 2], body[9], fmaf(R[3 * i], body[7], R[3 * i + 1] * body[8]));
      est[3 + i] = fmaf(R[3 * i + 2], body[12], fmaf(R[3 * i], body[10], R[3 * i + 1] * body[11]));
    }
  }
  // quat_to_rpy (:120-133) -- only the yaw of the world-frame rpy is used
  const float yaw = round_to_half(svml_atan2f(2.f * (x * y + w * z), w * w + x * x - y * y - z * z));      // np.arctan2 on float32 = SVML's routine (svml_acosf.h)
  const double th = (double)yaw;
  const float cz = round_to_half_d(cos(th)), sz = round_to_half_d(sin(th));
  const float ZT[9] = {cz, -sz, 0.f, sz, cz, 0.f, 0.f, 0.f, 1.f};          // world_R_yaw_frame^T
  // get_rot_from_normals / axis_angle_to_rot (:88-107), axis un-normalised, zz term uses k1*k1 (as written there)
  const float k0 = 0.f * normal[2] - 1.f * normal[1], k1 = 1.f * normal[0] - 0.f * normal[2], k2 = 0.f * normal[1] - 0.f * normal[0];
  const float theta = svml_acosf((0.f * normal[0] + 0.f * normal[1]) + 1.f * normal[2]);      // np.arccos on float32 = SVML's routine, bit for bit (svml_acosf.h)
  const float c_t = (float)cos((double)theta), s_t = (float)sin((double)theta), v_t = (float)(1.0 - cos((double)theta));
  float E[9];   // R_axis_angle as listed (= yaw_R_ground_frame^T)
  E[0] = round_to_half(k0 * k0 * v_t + c_t); E[1] = round_to_half(k0 * k1 * v_t - k2 * s_t); E[2] = round_to_half(k0 * k2 * v_t + k1 * s_t);
  E[3] = round_to_half(k0 * k1 * v_t + k2 * s_t); E[4] = round_to_half(k1 * k1 * v_t + c_t); E[5] = round_to_half(k1 * k2 * v_t - k0 * s_t);
  E[6] = round_to_half(k0 * k2 * v_t - k1 * s_t); E[7] = round_to_half(k1 * k2 * v_t + k0 * s_t); E[8] = round_to_half(k1 * k1 * v_t + c_t);
  float T1[9], G[9];
  hmatmul(R, ZT, T1);
  hmatmul(T1, E, G);            // ground_R_body_frame
  for (int k = 0; k < 9; ++k) est[9 + k] = G[k];
  // rot_to_rpy = quat_to_rpy(rot_to_quat(G))  (:159-199)
  float r[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[3 * i + j] = G[3 * j + i];   // r = rot.T
  const float tr = round_to_half((r[0] + r[4]) + r[8]);
  PyOrHalf qw, qx, qy, qz;
  double S;
  if (tr > 0.f) {
    S = sqrt((double)hadd(tr, 1.f)) * 2.0;
    const float Sh = round_to_half_d(S);
    qw.is_py = true; qw.py = 0.25 * S; qw.h = 0.f;
    qx = from_half(hdiv(hsub(r[7], r[5]), Sh)); qy = from_half(hdiv(hsub(r[2], r[6]), Sh)); qz = from_half(hdiv(hsub(r[3], r[1]), Sh));
  } else if (r[0] > r[4] && r[0] > r[8]) {
    S = sqrt((double)hsub(hsub(hadd(1.f, r[0]), r[4]), r[8])) * 2.0;
    const float Sh = round_to_half_d(S);
    qw = from_half(hdiv(hsub(r[7], r[5]), Sh));
    qx.is_py = true; qx.py = 0.25 * S; qx.h = 0.f;
    qy = from_half(hdiv(hadd(r[1], r[3]), Sh)); qz = from_half(hdiv(hadd(r[2], r[6]), Sh));
  } else if (r[4] > r[8]) {
    S = sqrt((double)hsub(hsub(hadd(1.f, r[4]), r[0]), r[8])) * 2.0;
    const float Sh = round_to_half_d(S);
    qw = from_half(hdiv(hsub(r[2], r[6]), Sh)); qx = from_half(hdiv(hadd(r[1], r[3]), Sh));
    qy.is_py = true; qy.py = 0.25 * S; qy.h = 0.f;
    qz = from_half(hdiv(hadd(r[5], r[7]), Sh));
  } else {
    S = sqrt((double)hsub(hsub(hadd(1.f, r[8]), r[0]), r[4])) * 2.0;
    const float Sh = round_to_half_d(S);
    qw = from_half(hdiv(hsub(r[3], r[1]), Sh)); qx = from_half(hdiv(hadd(r[2], r[6]), Sh)); qy = from_half(hdiv(hadd(r[5], r[7]), Sh));
    qz.is_py = true; qz.py = 0.25 * S; qz.h = 0.f;
  }
  const float as_h = hmul(-2.f, qsub(qmul(qx, qz), qmul(qw, qy)));
  const double as_ = fmin((double)as_h, 0.99999);
  const float roll = round_to_half(atan2f(hmul(2.f, qadd(qmul(qy, qz), qmul(qw, qx))),
                                          hadd(hsub(qsub(qmul(qw, qw), qmul(qx, qx)), as_half(qmul(qy, qy))), as_half(qmul(qz, qz)))));
  const float pitch = round_to_half_d(asin(as_));
  const float yawb = round_to_half(atan2f(hmul(2.f, qadd(qmul(qx, qy), qmul(qw, qz))),
                                          hsub(hsub(qadd(qmul(qw, qw), qmul(qx, qx)), as_half(qmul(qy, qy))), as_half(qmul(qz, qz)))));
  est[6] = roll; est[7] = pitch; est[8] = yawb;
}

// LegController.updateData (LegController.py:89-106): joint state, leg FK / Jacobian, foot velocity
MPC_HD void leg_update_data(CtrlState &s, const RobotConst &rc, const float *dof, int l0 = 0, int l1 = 4) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  for (int leg = l0; leg < l1; ++leg) {
    for (int j = 0; j < 3; ++j) { s.q[3 * leg + j] = dof[2 * (3 * leg + j)]; s.qd[3 * leg + j] = dof[2 * (3 * leg + j) + 1]; }
    leg_kinematics(rc, leg, s.q + 3 * leg, s.p + 3 * leg, s.J + 9 * leg);
    for (int r = 0; r < 3; ++r) {
      const float *Jr = s.J + 9 * leg + 3 * r, *qd = s.qd + 3 * leg;
      s.v[3 * leg + r] = Jr[0] * qd[0] + Jr[1] * qd[1] + Jr[2] * qd[2];
    }
  }
}

// ---- first half of the tick -----------------------------------------------------------------
// dof: [12][2] (pos, vel) leg-major; est: kEstLen floats; cmd: 16 floats (vx vy yaw_rate w[13]).
// rec: solver input record [56 + 4h] (written only when s.do_solve).
// The tick's first half comes in two parts so that the device can give every leg its own lane (mpc_batch.hip ctrl_pre_kernel): the legs
// [l0, l1) of this call do their own work, everything that concerns the whole robot is computed by every caller alike (and stored by the
// one with lead = true).  ctrl_pre_legs: joint data, kinematics, foot positions.  Between the two parts the lanes of a robot exchange
// foot_positions; ctrl_pre_rest needs those of all four legs.  The host and the FSM path call ctrl_pre, i.e. both parts for all legs.
MPC_HD void ctrl_pre_legs(CtrlState &s, const RobotConst &rc, const float *dof, int l0 = 0, int l1 = 4) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  leg_update_data(s, rc, dof, l0, l1);
  // foot positions (ConvexMPCLocomotion.py:248-250)
  for (int i = l0; i < l1; ++i) {
    float h[3];
    hip_location(rc, i, h);
    for (int c = 0; c < 3; ++c) s.foot_positions[3 * i + c] = h[c] + s.p[3 * i + c];
    s.pfoot[3 * i] = s.foot_positions[3 * i] + 0.f;
    s.pfoot[3 * i + 1] = s.foot_positions[3 * i + 1] + 0.f;
    s.pfoot[3 * i + 2] = s.foot_positions[3 * i + 2] + s.pos_z;
  }
}
// StateEstimator._compute_ground_normal_and_com_position without the CoM height (StateEstimator.py:120-143): the contact history takes the
// positions of the feet that were in contact, the ground normal is the normalised least-squares solution of  history n = 1.
MPC_HD void ground_normal_update(float *hist, float *normal, const float *contact_phase, const float *foot_positions) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  for (int i = 0; i < 4; ++i)
    if (contact_phase[i] != 0.f)
      for (int c = 0; c < 3; ++c) hist[3 * i + c] = foot_positions[3 * i + c];
  // least squares H n = 1: scipy.linalg.lstsq on float32 = LAPACK SGELSD there, walked operation by operation here (gelsd43.h) --
  // bit-identical to the reference's normal on every tick of the goldens
  float n[3];
  gelsd43::solve_ones(hist, n);
  for (int pass = 0; pass < 2; ++pass) {           // normalised twice (StateEstimator.py:134,140)
    // np.linalg.norm = sqrt(x.dot(x)); OpenBLAS' sdot rounds each product to float32, sums them in double and rounds once
    const float nn = sqrtf((float)(((double)(n[0] * n[0]) + (double)(n[1] * n[1])) + (double)(n[2] * n[2])));
    n[0] /= nn; n[1] /= nn; n[2] /= nn;
    if (pass == 0 && n[2] < 0.f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
  }
  normal[0] = n[0]; normal[1] = n[1]; normal[2] = n[2];
}
// ... and the first-run initialisation of the history (ConvexMPCLocomotion.py:257-263 -> StateEstimator.py:99-101)
MPC_HD void contact_history_init(float *hist, const float *foot_positions, double body_height) {
  for (int i = 0; i < 4; ++i) {
    hist[3 * i] = foot_positions[3 * i]; hist[3 * i + 1] = foot_positions[3 * i + 1];
    hist[3 * i + 2] = (float)(-body_height);
  }
}
MPC_HD void ctrl_pre_rest(CtrlState &s, const RobotConst &rc, const GaitTable &gt, const CtrlParams &cp,
                          const float *est, const float *cmd, float *rec, int l0 = 0, int l1 = 4, bool lead = true, bool do_normal = true) {
  // do_normal = false: the contact history and the ground normal are somebody else's (ctrl_pre_fused_kernel: another wavefront updates them
  // and writes the normal into the record)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const int nseg = gt.n_seg;
  const float *vBody = est, *omegaBody = est + 3, *rpyBody = est + 6, *gRb = est + 9;
  const float x_vel_des = cmd[0], y_vel_des = cmd[1], yaw_rate = cmd[2];   // ConvexMPCLocomotion.py:119-126
  const float *off = gt.offsets[s.gait_id], *dur = gt.durations[s.gait_id];
  // Gait.setIterations (Gait.py:26-28), called with the counter BEFORE the increment
  const int per = cp.iters_between_mpc * nseg;
  const double iteration = fmod((double)s.iter / (double)cp.iters_between_mpc, (double)nseg);
  const double phase = (double)(s.iter % per) / (double)per;

  if (s.first_run) {   // :257-263
    s.first_run = 0;
    contact_history_init(s.hist, s.foot_positions, rc.body_height);      // (the foot history of ALL legs: the ground-normal fit below reads it)
    for (int i = l0; i < l1; ++i)
      for (int c = 0; c < 3; ++c) { s.p0[3 * i + c] = s.pfoot[3 * i + c]; s.pf[3 * i + c] = s.pfoot[3 * i + c]; }
  }
  // StateEstimator._update_com_position_ground_frame (StateEstimator.py:109-118)
  {
    const float *cph = s.contact_phase;
    const float csum = ((cph[0] + cph[1]) + cph[2]) + cph[3];
    if (csum != 0.f) {
      float acc = 0.f;
      for (int i = 0; i < 4; ++i) {
        const float *fp = s.foot_positions + 3 * i;
        const float z = fmaf(fp[2], gRb[8], fmaf(fp[1], gRb[7], fp[0] * gRb[6]));   // (foot . gRb^T)[:, 2]: numpy's (4,3) x (3,3) float32 product is OpenBLAS sgemm, which fuses the k = 1, 2 terms
        acc = (i == 0) ? (-z) * cph[0] : acc + (-z) * cph[i];
      }
      s.pos_z = acc / csum;
    }
  }
  if (!cp.flat_ground && do_normal) ground_normal_update(s.hist, s.normal, s.contact_phase, s.foot_positions);

  // foot placement (ConvexMPCLocomotion.py:270-311)
  const float swing_seg = (float)nseg - dur[0], stance_seg = dur[0];        // Gait.py:22-23
  const float swing_time = (float)cp.dt_mpc * swing_seg;                      // getCurrentSwingTime
  const float stance_time = (float)cp.dt_mpc * stance_seg;                    // getCurrentStanceTime
  for (int l = l0; l < l1; ++l) s.swing_times[l] = swing_time;
  const float posz = s.pos_z;
  // coordinateRotation(Z, -yaw_rate * stance_time / 2): float16 matrix (orientation_tools.py:13,20-37) -- the same for the four legs
  const float theta = -yaw_rate * stance_time / 2.f;
  const float cz = round_to_half((float)cos((double)theta)), sz = round_to_half((float)sin((double)theta));
  const float msz = round_to_half((float)(-sin((double)theta)));
  for (int i = l0; i < l1; ++i) {
    if (s.first_swing[i]) s.swing_time_remaining[i] = (double)s.swing_times[i];
    else s.swing_time_remaining[i] -= cp.dt;
    const float side = (i == 0 || i == 2) ? 1.f : -1.f;
    float pr[3];
    hip_location(rc, i, pr);
    pr[1] = pr[1] + (float)((double)side * rc.abad);
    const float pyc[3] = {cz * pr[0] + sz * pr[1] + 0.f * pr[2], msz * pr[0] + cz * pr[1] + 0.f * pr[2], 0.f * pr[0] + 0.f * pr[1] + 1.f * pr[2]};
    const float str = (float)s.swing_time_remaining[i];
    float Pf[3] = {0.f + (pyc[0] + x_vel_des * str), 0.f + (pyc[1] + y_vel_des * str), posz + (pyc[2] + 0.f * str)};
    float pfx = vBody[0] * 0.5f * stance_time + 0.03f * (vBody[0] - x_vel_des) + (0.5f * posz / 9.81f) * (vBody[1] * yaw_rate);
    float pfy = vBody[1] * 0.5f * stance_time * (float)cp.dt_mpc + 0.03f * (vBody[1] - y_vel_des) + (0.5f * posz / 9.81f) * (-vBody[0] * yaw_rate);
    pfx = fminf(fmaxf(pfx, -0.3f), 0.3f);
    pfy = fminf(fmaxf(pfy, -0.3f), 0.3f);
    Pf[0] += pfx; Pf[1] += pfy; Pf[2] = -0.003f;
    s.pf[3 * i] = Pf[0]; s.pf[3 * i + 1] = Pf[1]; s.pf[3 * i + 2] = Pf[2];
  }
  s.iter += 1;   // :314

  // gait states (Gait.py:30-67) -- phase set before the increment
  for (int i = l0; i < l1; ++i) {
    const float offf = off[i] / (float)nseg, durf = dur[i] / (float)nseg;
    float pc = (float)phase - offf;
    if (pc < 0.f) pc += 1.0f;
    s.contact_states[i] = (pc > durf) ? 0.f : pc / durf;
    float so = offf + durf;
    if (so > 1.f) so -= 1.0f;
    const float sd = 1.f - durf;
    float ps = (float)phase - so;
    if (ps < 0.f) ps += 1.0f;
    s.swing_states[i] = (ps > sd) ? 0.f : (sd == 0.f ? 0.f : ps / sd);
  }
  s.vbody[0] = vBody[0]; s.vbody[1] = vBody[1]; s.vbody[2] = vBody[2];
  s.posz_tick = s.pos_z;

  // updateMPCIfNeeded / solveDenseMPC marshalling (ConvexMPCLocomotion.py:128-185, 217-220)
  s.do_solve = (s.iter % cp.iters_between_mpc) == 0;
  if (s.do_solve) {
    const int h = cp.horizon;
    // weights from the command (DesiredStateCommand.py:24-28), or Quadruped._mpc_weights when the command carries none
    // (mpc_weights is None, ConvexMPCLocomotion.py:132-135: the 3-entry commands of the interactive runners) -- here: cmd[3 .. 15] all NaN.
    // The reference asserts w >= 0 (DesiredStateCommand.py:21,27); a negative weight makes this robot's record non-finite, so its
    // solve reports NON_CVX and its previous forces stay in place.
    if (lead) {
      bool dflt = true;                      // "no weights" = ALL thirteen entries NaN (what BatchedLocomotion sends for a 3-entry command); a NaN
      for (int k = 0; k < 13; ++k) dflt = dflt && (cmd[3 + k] != cmd[3 + k]);   // among real weights (a diverged policy) poisons the record -> NON_CVX
      bool neg = false;
      for (int k = 0; k < 13; ++k) { const float w = dflt ? rc.weights[k] : cmd[3 + k]; neg = neg || (w < 0.f); rec[IN_W + k] = w; }
      if (neg) for (int k = 0; k < 13; ++k) rec[IN_W + k] = __builtin_nanf("");
    }
    if (lead) {
      rec[IN_POS] = 0.f; rec[IN_POS + 1] = 0.f; rec[IN_POS + 2] = s.pos_z;
      for (int k = 0; k < 3; ++k) {
        rec[IN_VEL + k] = vBody[k];
        rec[IN_RPY + k] = rpyBody[k];
        if (cp.flat_ground) rec[IN_NRM + k] = k == 2 ? 1.f : 0.f;
        else if (do_normal) rec[IN_NRM + k] = s.normal[k];
        rec[IN_ANG + k] = omegaBody[k];
      }
    }
    for (int i = 0; i < h; ++i) {                                             // Gait.getMpcTable (Gait.py:69-84)
      const double it = fmod((double)i + iteration + 1.0, (double)nseg);
      for (int j = l0; j < l1; ++j) {
        float pg = (float)it - off[j];
        if (pg < 0.f) pg += (float)nseg;
        rec[IN_CONTACT + 4 * i + j] = (pg < dur[j]) ? 1.f : 0.f;
      }
    }
    const int o_foot = 28 + 4 * h, o_fric = 40 + 4 * h, o_dpos = 44 + 4 * h, o_dvel = 47 + 4 * h, o_drpy = 50 + 4 * h, o_dang = 53 + 4 * h;
    for (int k = 3 * l0; k < 3 * l1; ++k) rec[o_foot + k] = s.foot_positions[k];
    for (int k = l0; k < l1; ++k) rec[o_fric + k] = rc.mu;
    if (lead) {
      rec[o_dpos] = 0.f; rec[o_dpos + 1] = 0.f; rec[o_dpos + 2] = (float)rc.body_height;
      rec[o_dvel] = x_vel_des; rec[o_dvel + 1] = y_vel_des; rec[o_dvel + 2] = 0.f;
      rec[o_drpy] = rec[o_drpy + 1] = rec[o_drpy + 2] = 0.f;
      rec[o_dang] = 0.f; rec[o_dang + 1] = 0.f; rec[o_dang + 2] = yaw_rate;
    }
  }
}
MPC_HD void ctrl_pre(CtrlState &s, const RobotConst &rc, const GaitTable &gt, const CtrlParams &cp, const float *dof,
                     const float *est, const float *cmd, float *rec) {
  ctrl_pre_legs(s, rc, dof);
  ctrl_pre_rest(s, rc, gt, cp, est, cmd, rec);
}

// interplation.py:4-26
MPC_HD float bez(float x) { return x * x * x + 3.0f * (x * x * (1.0f - x)); }
MPC_HD float bez_d(float x) { return 6.0f * x * (1.0f - x); }

// ---- second half of the tick ------------------------------------------------------------------
// forces: this robot's solver output (fp64 [12h], first 12 used) -- read only when the solve ran and
// reported OSQP_SOLVED.  torques: 12 floats, FL FR RL RR x (hip, thigh, calf).
MPC_HD void ctrl_post(CtrlState &s, const RobotConst &rc, const double *forces, int solved, float *torques, int l0 = 0, int l1 = 4) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  if (s.do_solve && solved)
    for (int k = 3 * l0; k < 3 * l1; ++k) s.f_ff[k] = (float)forces[k];      // ConvexMPCLocomotion.py:186-187
  const float height = (float)(rc.body_height / 3.0);                          // :287
  for (int foot = l0; foot < l1; ++foot) {
    const float swing = s.swing_states[foot];
    float hloc[3];
    hip_location(rc, foot, hloc);
    float kp[3] = {0.f, 0.f, 0.f}, kd[3] = {7.f, 7.f, 7.f}, ff[3] = {0.f, 0.f, 0.f}, kdj = 0.f;
    if (swing > 0.f) {   // :327-348
      if (s.first_swing[foot]) {
        s.first_swing[foot] = 0;
        for (int c = 0; c < 3; ++c) s.p0[3 * foot + c] = s.pfoot[3 * foot + c];
      }
      const float T = (float)(double)s.swing_times[foot];                       // swingTimes[foot].item()
      const float *p0 = s.p0 + 3 * foot, *pf = s.pf + 3 * foot;
      float *tp = s.tp + 3 * foot, *tv = s.tv + 3 * foot;
      const float b = bez(swing), bd = bez_d(swing);
      for (int c = 0; c < 3; ++c) { tp[c] = p0[c] + b * (pf[c] - p0[c]); tv[c] = bd * (pf[c] - p0[c]) / T; }
      if (swing < 0.5f) {   // FootSwingTrajectory.py:59-62
        const float x = swing * 2.f, top = p0[2] + height;
        tp[2] = p0[2] + bez(x) * (top - p0[2]);
        tv[2] = bez_d(x) * (top - p0[2]) * 2.f / T;
      } else {
        const float x = swing * 2.f - 1.f, top = p0[2] + height;
        tp[2] = top + bez(x) * (pf[2] - top);
        tv[2] = bez_d(x) * (pf[2] - top) * 2.f / T;
      }
      kp[0] = 700.f; kp[1] = 700.f; kp[2] = 150.f;                            // :82-83
      s.contact_phase[foot] = 0.f;
    } else {             // stance :350-376
      s.first_swing[foot] = 1;
      for (int c = 0; c < 3; ++c) ff[c] = s.f_ff[3 * foot + c];
      kdj = 0.2f;
      s.contact_phase[foot] = s.contact_states[foot];                          // setContactPhase (:378)
    }
    // pDesLeg = (pDesFoot - position) - hip ; vDesLeg = vDesFoot - vBody
    float pdes[3], vdes[3];
    const float pos[3] = {0.f, 0.f, s.posz_tick};
    for (int c = 0; c < 3; ++c) { pdes[c] = (s.tp[3 * foot + c] - pos[c]) - hloc[c]; vdes[c] = s.tv[3 * foot + c] - s.vbody[c]; }
    // LegController.updateCommand (LegController.py:108-132)
    float force[3];
    for (int c = 0; c < 3; ++c)
      force[c] = ff[c] + kp[c] * (pdes[c] - s.p[3 * foot + c]) + kd[c] * (vdes[c] - s.v[3 * foot + c]);
    const float *J = s.J + 9 * foot;
    for (int j = 0; j < 3; ++j) {
      float tau = 0.f + (J[j] * force[0] + J[3 + j] * force[1] + J[6 + j] * force[2]);   // tauFeedForward + J^T f
      tau += 0.f * (0.f - s.q[3 * foot + j]);                                              // kpJoint = 0
      tau += kdj * (0.f - s.qd[3 * foot + j]);                                             // kdJoint (qdDes = 0)
      torques[3 * foot + j] = tau;
    }
  }
}


// ============================================================================================================
// Control FSM (MPC_Controller/FSM_states/ControlFSM.py:71-124 runFSM; FSM_State_Passive.py, FSM_State_RecoveryStand.py,
// FSM_State_Locomotion.py; driven by robot_runner/RobotRunnerFSM.py:44-71), one robot per thread.  The reference's
// process-global Parameters.control_mode becomes a per-robot request, Parameters.locomotionUnsafe a per-robot flag.
// ============================================================================================================
enum { kFsmPassive = 0, kFsmLocomotion = 4, kFsmRecoveryStand = 6 };          // FSM_StateName (utils.py:26-30)
enum { kOpTest = 0, kOpNormal = 1, kOpTransitioning = 2 };                    // FSM_OperatingMode (utils.py:32-35)
enum { kRsStandUp = 0, kRsFoldLegs = 1, kRsRollOver = 2 };                    // FSM_State_RecoveryStand.py:8-10

struct FsmParams {
  int check_safety;                                      // Parameters.FSM_check_safety
  int fold_ramp, fold_settle, standup_ramp, standup_settle, roll_ramp, roll_settle;   // int(k / (controller_dt * 100)), RecoveryStand :35-57
};
struct FsmState {
  int cur, next_state, op_mode;
  int passive_iter, loco_iter, rs_iter, rs_state_iter, rs_motion_start, rs_flag;
  float rs_initial[12];
  int unsafe;                     // set when locomotionSafe() failed (Parameters.locomotionUnsafe)
  float last_rb22;                // rBody[2,2] of the estimator's current result (0 after StateEstimator.reset)
  // this tick's outcome
  int run_loco, entered_loco;     // ConvexMPCLocomotion.run is due / LOCOMOTION was entered (new ConvexMpc object: cold solver)
  float qdes[12], kpj, kdj;       // joint PD command (kp = kd = 0 after zeroCommand when no state ran)
};

MPC_HD FsmParams fsm_params(double controller_dt, int check_safety) {
  FsmParams P;
  const double d = controller_dt * 100.0;
  P.check_safety = check_safety;
  P.fold_ramp = (int)(45 / d); P.fold_settle = (int)(75 / d); P.standup_ramp = (int)(30 / d); P.standup_settle = (int)(30 / d);
  P.roll_ramp = (int)(13 / d); P.roll_settle = (int)(15 / d);
  return P;
}
// rBody[2,2] of quat_to_rot (orientation_tools.py:135-149; float32 arithmetic, float16 storage) -- _UpsideDown reads its sign
MPC_HD float rbody22(const float *body) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float e1 = body[3], e2 = body[4];
  return round_to_half(1.f - 2.f * (e1 * e1 + e2 * e2));
}
// roll and pitch of quat_to_rpy (orientation_tools.py:120-133) as the estimator stores them (float16)
MPC_HD void world_roll_pitch(const float *body, float *roll, float *pitch) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float x = body[3], y = body[4], z = body[5], w = body[6];
  *roll = round_to_half(svml_atan2f(2.f * (y * z + w * x), w * w - x * x - y * y + z * z));
  *pitch = round_to_half_d(asin(fmin((double)(-2.f * (x * z - w * y)), 0.99999)));
}

MPC_HD void fsm_on_enter(FsmState &f, int state, CtrlState &s, const RobotConst &rc, float rb22) {
  f.next_state = state;
  if (state == kFsmLocomotion) {            // FSM_State_Locomotion.onEnter (:32-42): cMPC.initialize, command reset, estimator reset
    ctrl_reset(s, rc);
    f.entered_loco = 1;
    f.last_rb22 = 0.f;                      // stateEstimator.reset(): fresh StateEstimate (rBody = 0)
  } else if (state == kFsmRecoveryStand) {  // FSM_State_RecoveyrStand.onEnter (:65-92)
    f.rs_iter = 0; f.rs_state_iter = 0;
    for (int k = 0; k < 12; ++k) f.rs_initial[k] = s.q[k];
    const float h = s.pos_z;                // stateEstimator.getResult().position[2]
    f.rs_flag = kRsFoldLegs;
    if (!(rb22 < 0.f) && 0.2f < h && h < 0.45f) f.rs_flag = kRsStandUp;
    f.rs_motion_start = 0;
  }
}

// ControlFSM.__init__ / initialize (:28-78): fresh states, enter the state Parameters.control_mode names
MPC_HD void fsm_init(FsmState &f, int control_mode, int op_mode, CtrlState &s, const RobotConst &rc, float rb22) {
  f.passive_iter = 0; f.loco_iter = 0; f.rs_iter = 0; f.rs_state_iter = 0; f.rs_motion_start = 0; f.rs_flag = kRsFoldLegs;
  for (int k = 0; k < 12; ++k) { f.rs_initial[k] = 0.f; f.qdes[k] = 0.f; }
  f.unsafe = 0; f.run_loco = 0; f.entered_loco = 0; f.kpj = 0.f; f.kdj = 0.f; f.last_rb22 = rb22;
  f.cur = control_mode;
  fsm_on_enter(f, control_mode, s, rc, rb22);
  f.op_mode = op_mode;
}
// ControlFSM.initialize again (RobotRunnerFSM.reset, :41-42): the state objects and their counters persist
MPC_HD void fsm_reinit(FsmState &f, int control_mode, int op_mode, CtrlState &s, const RobotConst &rc, float rb22) {
  f.cur = control_mode;
  fsm_on_enter(f, control_mode, s, rc, rb22);
  f.op_mode = op_mode;
}

MPC_HD void fsm_joint_pd(FsmState &f, int leg, const float *qdes) {   // FSM_State.jointPDControl (:44-66): kp = 80 I, kd = I
  for (int j = 0; j < 3; ++j) f.qdes[3 * leg + j] = qdes[j];
  f.kpj = 80.f; f.kdj = 1.f;
}
MPC_HD void fsm_interp(FsmState &f, int curr, int max_iter, const float *fin12) {   // _SetJPosInterPts (:167-181), all legs
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  double a = 0.0, b = 1.0;
  if (curr <= max_iter) { b = (double)curr / (double)max_iter; a = 1.0 - b; }
  const float af = (float)a, bf = (float)b;                 // Python floats enter the float32 array arithmetic as float32
  for (int leg = 0; leg < 4; ++leg) {
    float q[3];
    for (int j = 0; j < 3; ++j) q[j] = af * f.rs_initial[3 * leg + j] + bf * fin12[3 * leg + j];
    fsm_joint_pd(f, leg, q);
  }
}

MPC_HD void fsm_run_state(FsmState &f, CtrlState &s, const RobotConst &rc, const FsmParams &P, float rb22) {
  const float fold[12] = {0.0f, 1.4f, -2.7f, -0.0f, 1.4f, -2.7f, 0.0f, 1.4f, -2.7f, -0.0f, 1.4f, -2.7f};        // :37-41
  const float stand[12] = {0.f, 0.8f, -1.6f, 0.f, 0.8f, -1.6f, 0.f, 0.8f, -1.6f, 0.f, 0.8f, -1.6f};            // :47-51
  const float rolling[12] = {1.3f, 3.1f, -2.77f, 0.0f, 1.6f, -2.77f, 1.3f, 3.1f, -2.77f, 0.0f, 1.6f, -2.77f};  // :57-61
  if (f.cur == kFsmPassive) {                 // FSM_State_Passive.run (:31-42)
    if (f.passive_iter < 10) {
      const float q[3] = {0.0f, 0.01f, 0.01f};
      for (int leg = 0; leg < 4; ++leg) fsm_joint_pd(f, leg, q);
    }
  } else if (f.cur == kFsmLocomotion) {       // LocomotionControlStep (:138-139)
    f.run_loco = 1;
  } else {                                    // FSM_State_RecoveyrStand.run (:94-106)
    const int curr = f.rs_state_iter - f.rs_motion_start;
    const bool upside = rb22 < 0.f;
    if (f.rs_flag == kRsStandUp) {            // _StandUp (:184-205)
      const bool wrong = upside || rc.body_height < 0.1;
      if (curr > (int)floor(P.standup_ramp * 0.7) && wrong) {
        for (int k = 0; k < 12; ++k) f.rs_initial[k] = s.q[k];
        f.rs_flag = kRsFoldLegs;
        f.rs_motion_start = f.rs_state_iter + 1;
      } else {
        fsm_interp(f, curr, P.standup_ramp, stand);
      }
    } else if (f.rs_flag == kRsFoldLegs) {    // _FoldLegs (:209-224) -- ramps over rollover_ramp_iter, as written there
      fsm_interp(f, curr, P.roll_ramp, fold);
      if (curr >= P.fold_ramp + P.fold_settle) {
        f.rs_flag = upside ? kRsRollOver : kRsStandUp;
        for (int k = 0; k < 12; ++k) f.rs_initial[k] = fold[k];
        f.rs_motion_start = f.rs_state_iter + 1;
      }
    } else {                                  // _RollOver (:226-235)
      fsm_interp(f, curr, P.roll_ramp, rolling);
      if (curr > P.roll_ramp + P.roll_settle) {
        f.rs_flag = kRsFoldLegs;
        for (int k = 0; k < 12; ++k) f.rs_initial[k] = rolling[k];
        f.rs_motion_start = f.rs_state_iter + 1;
      }
    }
    f.rs_state_iter += 1;
  }
}

// FSM_State_Locomotion.locomotionSafe (:104-136), comparisons in the types numpy uses there: float16 angles against the
// float16-rounded limit, float32 leg positions against float32 constants; the roll test has no abs (as written)
MPC_HD bool fsm_locomotion_safe(const CtrlState &s, const FsmParams &P, const float *body) {
  if (!P.check_safety) return true;
  float roll, pitch;
  world_roll_pitch(body, &roll, &pitch);
  const float lim = round_to_half_d(40.0 * 3.14159265358979323846 / 180.0);
  if (roll > lim) return false;
  if (fabsf(pitch) > lim) return false;
  for (int leg = 0; leg < 4; ++leg) {
    if (s.p[3 * leg + 2] > 0.f) return false;
    if (s.p[3 * leg + 1] > 0.18f) return false;
  }
  return true;
}

MPC_HD int fsm_check_transition(FsmState &f, const CtrlState &s, const FsmParams &P, int request, const float *body) {
  if (f.cur == kFsmPassive) {                 // FSM_State_Passive.checkTransition (:52-74)
    f.next_state = f.cur;
    f.passive_iter += 1;
    if (request == kFsmRecoveryStand) f.next_state = kFsmRecoveryStand;     // anything else but PASSIVE: refused
  } else if (f.cur == kFsmRecoveryStand) {    // FSM_State_RecoveyrStand.checkTransition (:115-140)
    f.next_state = f.cur;
    f.rs_iter += 1;
    if (request == kFsmLocomotion || request == kFsmPassive) f.next_state = request;
  } else {                                    // FSM_State_Locomotion.checkTransition (:54-84)
    f.loco_iter += 1;
    if (fsm_locomotion_safe(s, P, body)) {
      if (request == kFsmPassive || request == kFsmRecoveryStand) f.next_state = request;
    } else {
      f.next_state = kFsmRecoveryStand;
      f.unsafe = 1;
    }
  }
  return f.next_state;
}

// One tick of ControlFSM.runFSM after updateData / zeroCommand / StateEstimator.update.  On return f.run_loco tells whether
// ConvexMPCLocomotion.run is due (the caller then runs ctrl_pre / solve / ctrl_post), otherwise f.qdes / kpj / kdj hold the
// joint PD command of the tick.
MPC_HD void fsm_tick(FsmState &f, CtrlState &s, const RobotConst &rc, const FsmParams &P, const float *dof, const float *body, int request) {
  leg_update_data(s, rc, dof);
  for (int k = 0; k < 12; ++k) f.qdes[k] = 0.f;
  f.kpj = 0.f; f.kdj = 0.f; f.run_loco = 0; f.entered_loco = 0;
  const float rb22 = rbody22(body);
  f.last_rb22 = rb22;
  if (f.op_mode == kOpTest) {
    fsm_run_state(f, s, rc, P, rb22);
  } else if (f.op_mode == kOpNormal) {
    const int nxt = fsm_check_transition(f, s, P, request, body);
    if (nxt != f.cur) { f.op_mode = kOpTransitioning; f.next_state = nxt; }
    else fsm_run_state(f, s, rc, P, rb22);
  } else {   // TRANSITIONING: every transition() of the three states completes at once (:86-105 / :142-164 / :76-85)
    if (f.cur == kFsmLocomotion) f.loco_iter = 0;       // FSM_State_Locomotion.onExit (:50-51)
    f.cur = f.next_state;
    fsm_on_enter(f, f.cur, s, rc, rb22);
    f.op_mode = kOpNormal;
  }
}

// LegController.updateCommand (LegController.py:108-132) when only the joint PD part of the command is set
MPC_HD void fsm_joint_torques(const FsmState &f, const CtrlState &s, float *torques) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  for (int k = 0; k < 12; ++k) {
    float tau = 0.f;
    tau += f.kpj * (f.qdes[k] - s.q[k]);
    tau += f.kdj * (0.f - s.qd[k]);
    torques[k] = tau;
  }
}

}  // namespace mpc
