/*
 * oracle/convex_mpc_assembly.h -- TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked,
 * imported or executed by the product path (rl-mpc-locomotion_amd/); only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * Plain-C restatement (no Eigen) of the QP assembly of the reference's
 *   MPC_Controller/convex_MPC/mpc_osqp.cc   (class ConvexMpc, OSQP branch)
 * Every function cites the reference lines it follows.  All reference quirks are kept on purpose:
 *   - feet are rotated with Rx*Ry*Rz (mpc_osqp.cc:606-617) while the inertia uses Rz*Ry*Rx (:283-291,669-671)
 *   - A_qp's last 13x13 block stays zero (loop bound `i < horizon - 1`, mpc_osqp.cc:361, zero-init :568)
 *   - the 4-vector of "foot friction coeffs" is used as the four cone-row coefficients of EVERY leg
 *     (mpc_osqp.cc:441-446) and only element [0] enters the bounds (:687)
 *   - P is built with the reference's block recursion, in the reference's order (mpc_osqp.cc:387-434)
 * The matrix exponential (mpc_osqp.cc:338-351, Eigen Pade) is replaced by its exact closed form:
 * the 25x25 augmented matrix M is nilpotent of index 3, so exp(M) = I + M + M^2/2 exactly
 * (tests/test_oracle.py::test_exponential_closed_form_matches_expm checks this against scipy.linalg.expm to 1e-15).
 *
 * The arithmetic type is AREAL (double unless overridden).  The reference computes in double; the
 * float instantiation exists only so the CPU can predict what an fp32 device assembly does.
 *
 * PARITY PIN: the reference ships no test or golden vector for this path ("parity unpinned",
 * SURVEY.md 8c); the pin is the vendored OSQP itself, built from /root/reference by oracle/Makefile.
 */
#ifndef CONVEX_MPC_ASSEMBLY_H
#define CONVEX_MPC_ASSEMBLY_H

#include <math.h>
#include <string.h>

#ifndef AREAL
#define AREAL double
#endif

#define MPC_STATE_DIM 13          /* mpc_osqp.cc:180 kStateDim */
#define MPC_NUM_LEGS 4
#define MPC_ACTION_DIM 12         /* num_legs * 3 */
#define MPC_CONSTRAINT_DIM 5      /* mpc_osqp.cc:184 kConstraintDim */
#define MPC_MAX_H 20
#define MPC_GRAVITY 9.8           /* mpc_osqp.cc:54 */
#define MPC_MAX_SCALE 10.0        /* mpc_osqp.cc:55 */
#define MPC_MIN_SCALE 0.1         /* mpc_osqp.cc:56 */

/* Flat input record of one compute_contact_forces() call: the 13 positional arguments of
 * mpc_osqp.cc:578-591 concatenated in call order.  Length 56 + 4*h. */
enum {
  MPC_IN_WEIGHTS = 0,   /* qp_weights[13] */
  MPC_IN_COM_POS = 13,  /* com_position[3] */
  MPC_IN_COM_VEL = 16,  /* com_velocity[3] */
  MPC_IN_RPY = 19,      /* com_roll_pitch_yaw[3] */
  MPC_IN_NORMAL = 22,   /* ground_normal_vec[3] */
  MPC_IN_ANGVEL = 25,   /* com_angular_velocity[3] */
  MPC_IN_CONTACT = 28   /* foot_contact_states[4*h], row-major [step][leg]; the rest follows */
};
static inline int mpc_in_footpos(int h) { return 28 + 4 * h; }      /* foot_positions_body_frame[12] */
static inline int mpc_in_friction(int h) { return 40 + 4 * h; }     /* foot_friction_coeffs[4] */
static inline int mpc_in_des_pos(int h) { return 44 + 4 * h; }      /* desired_com_position[3] */
static inline int mpc_in_des_vel(int h) { return 47 + 4 * h; }      /* desired_com_velocity[3] */
static inline int mpc_in_des_rpy(int h) { return 50 + 4 * h; }      /* desired_com_roll_pitch_yaw[3] */
static inline int mpc_in_des_angvel(int h) { return 53 + 4 * h; }   /* desired_com_angular_velocity[3] */
static inline int mpc_in_len(int h) { return 56 + 4 * h; }

typedef struct {
  int h;                 /* planning_horizon */
  AREAL mass, inv_mass;  /* mpc_osqp.cc:513-514 */
  AREAL inv_inertia[9];  /* body-frame inverse inertia, row-major (mpc_osqp.cc:515-516) */
  AREAL dt;              /* timestep_ */
  AREAL alpha;           /* alpha_single_ = alpha * I (mpc_osqp.cc:525-526) */
} MpcModel;

/* mpc_osqp.cc:515-516: inertia_(inertia.data()) then .inverse().  Eigen reads the 9 numbers
 * column-major, i.e. the transpose of the row-major list Python passes; the inertia is symmetric
 * so this is the same matrix.  General 3x3 inverse by cofactors. */
static void mpc_model_init(MpcModel *m, double mass, const double inertia9[9], int h, double dt, double alpha) {
  double a[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) a[r * 3 + c] = inertia9[c * 3 + r];
  double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  double inv[9] = {c00 / det, (a[2] * a[7] - a[1] * a[8]) / det, (a[1] * a[5] - a[2] * a[4]) / det,
                   c01 / det, (a[0] * a[8] - a[2] * a[6]) / det, (a[2] * a[3] - a[0] * a[5]) / det,
                   c02 / det, (a[1] * a[6] - a[0] * a[7]) / det, (a[0] * a[4] - a[1] * a[3]) / det};
  m->h = h;
  m->mass = (AREAL)mass;
  m->inv_mass = (AREAL)(1.0 / mass);
  for (int i = 0; i < 9; ++i) m->inv_inertia[i] = (AREAL)inv[i];
  m->dt = (AREAL)dt;
  m->alpha = (AREAL)alpha;
}

static void mpc_mat3_mul(const AREAL *a, const AREAL *b, AREAL *c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      AREAL s = 0;
      for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
      c[i * 3 + j] = s;
    }
}
static void mpc_rot_x(AREAL t, AREAL *r) {
  AREAL c = cos(t), s = sin(t);
  AREAL m[9] = {1, 0, 0, 0, c, -s, 0, s, c};
  memcpy(r, m, sizeof m);
}
static void mpc_rot_y(AREAL t, AREAL *r) {
  AREAL c = cos(t), s = sin(t);
  AREAL m[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
  memcpy(r, m, sizeof m);
}
static void mpc_rot_z(AREAL t, AREAL *r) {
  AREAL c = cos(t), s = sin(t);
  AREAL m[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  memcpy(r, m, sizeof m);
}

/* Dense work area of one assembly.  Sizes for MPC_MAX_H. */
typedef struct {
  AREAL x0[MPC_STATE_DIM];                       /* state_ (mpc_osqp.cc:630-633) */
  AREAL xref[MPC_STATE_DIM * MPC_MAX_H];         /* desired_states_ (:635-659) */
  AREAL a_mat[MPC_STATE_DIM * MPC_STATE_DIM];    /* a_mat_ */
  AREAL b_mat[MPC_STATE_DIM * MPC_ACTION_DIM];   /* b_mat_ */
  AREAL a_exp[MPC_STATE_DIM * MPC_STATE_DIM];
  AREAL b_exp[MPC_STATE_DIM * MPC_ACTION_DIM];
  AREAL a_qp[MPC_MAX_H * MPC_STATE_DIM * MPC_STATE_DIM]; /* block i = rows 13i..13i+12 */
  AREAL anb[MPC_MAX_H * MPC_STATE_DIM * MPC_ACTION_DIM]; /* anb_aux_: block k = A_exp^k B_exp */
  AREAL sdiff[MPC_STATE_DIM * MPC_MAX_H];        /* state_diff (:681) */
} MpcWork;

/* mpc_osqp.cc:299-322 CalculateAMat */
static void mpc_calc_a_mat(const AREAL rpy[3], const AREAL normal[3], AREAL *a) {
  memset(a, 0, sizeof(AREAL) * MPC_STATE_DIM * MPC_STATE_DIM);
  const AREAL cy = cos(rpy[2]), sy = sin(rpy[2]), cp = cos(rpy[1]), tp = tan(rpy[1]);
  const AREAL t[9] = {cy / cp, sy / cp, 0, -sy, cy, 0, cy * tp, sy * tp, 1};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) a[r * 13 + 6 + c] = t[r * 3 + c];
  a[3 * 13 + 9] = 1;
  a[4 * 13 + 10] = 1;
  a[5 * 13 + 11] = 1;
  a[9 * 13 + 12] = normal[0];
  a[10 * 13 + 12] = normal[1];
  a[11 * 13 + 12] = normal[2];
}

/* mpc_osqp.cc:324-336 CalculateBMat; foot_world is [leg][xyz] */
static void mpc_calc_b_mat(AREAL inv_mass, const AREAL *inv_inertia_w, const AREAL *foot_world, AREAL *b) {
  memset(b, 0, sizeof(AREAL) * MPC_STATE_DIM * MPC_ACTION_DIM);
  for (int i = 0; i < MPC_NUM_LEGS; ++i) {
    const AREAL *v = foot_world + 3 * i;
    /* mpc_osqp.cc:293-297 ConvertToSkewSymmetric */
    const AREAL skew[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    AREAL blk[9];
    mpc_mat3_mul(inv_inertia_w, skew, blk);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) b[(6 + r) * 12 + 3 * i + c] = blk[r * 3 + c];
    b[9 * 12 + 3 * i] = inv_mass;
    b[10 * 12 + 3 * i + 1] = inv_mass;
    b[11 * 12 + 3 * i + 2] = inv_mass;
  }
}

/* mpc_osqp.cc:338-351 CalculateExponentials, closed form (see header): with M = [[A dt, B dt],[0,0]],
 * M^3 = 0, hence A_exp = I + A dt + (A dt)^2/2 and B_exp = B dt + (A dt)(B dt)/2. */
static void mpc_calc_exponentials(const AREAL *a, const AREAL *b, AREAL dt, AREAL *a_exp, AREAL *b_exp) {
  AREAL adt[169], bdt[156];
  for (int i = 0; i < 169; ++i) adt[i] = a[i] * dt;
  for (int i = 0; i < 156; ++i) bdt[i] = b[i] * dt;
  for (int r = 0; r < 13; ++r)
    for (int c = 0; c < 13; ++c) {
      AREAL s = 0;
      for (int k = 0; k < 13; ++k) s += adt[r * 13 + k] * adt[k * 13 + c];
      a_exp[r * 13 + c] = (r == c ? (AREAL)1 : (AREAL)0) + adt[r * 13 + c] + s / 2;
    }
  for (int r = 0; r < 13; ++r)
    for (int c = 0; c < 12; ++c) {
      AREAL s = 0;
      for (int k = 0; k < 13; ++k) s += adt[r * 13 + k] * bdt[k * 12 + c];
      b_exp[r * 12 + c] = bdt[r * 12 + c] + s / 2;
    }
}

/* C(13 x nc) = A(13x13) * B(13 x nc) */
static void mpc_mul13(const AREAL *a, const AREAL *b, int nc, AREAL *c) {
  for (int r = 0; r < 13; ++r)
    for (int j = 0; j < nc; ++j) {
      AREAL s = 0;
      for (int k = 0; k < 13; ++k) s += a[r * 13 + k] * b[k * nc + j];
      c[r * nc + j] = s;
    }
}

/* blk(12x12) = X(13x12)^T * diag(w) * Y(13x12) */
static void mpc_xtqy(const AREAL *x, const AREAL *w, const AREAL *y, AREAL *blk) {
  for (int i = 0; i < 12; ++i)
    for (int j = 0; j < 12; ++j) {
      AREAL s = 0;
      for (int k = 0; k < 13; ++k) s += x[k * 12 + i] * w[k] * y[k * 12 + j];
      blk[i * 12 + j] = s;
    }
}

/*
 * Full assembly of one call: in[] is the flat record above (as doubles: pybind11 converts every
 * argument to std::vector<double>, mpc_osqp.cc:578-591).
 * Outputs: P (n x n row-major, full symmetric), q (n), cone[5*3] the single 5x3 constraint block
 * (same for every (step, leg), mpc_osqp.cc:437-447), l/u (m), n = 12h, m = 20h.
 */
static void mpc_assemble(const MpcModel *mdl, const double *in, MpcWork *wk, AREAL *P, AREAL *q, AREAL *cone,
                         AREAL *lb, AREAL *ub) {
  const int h = mdl->h, n = 12 * h;
  AREAL w[13], rpy[3], normal[3], com_pos[3], com_vel[3], ang_vel[3];
  AREAL des_pos[3], des_vel[3], des_rpy[3], des_ang[3], fric[4];
  for (int i = 0; i < 13; ++i) w[i] = (AREAL)in[MPC_IN_WEIGHTS + i];
  for (int i = 0; i < 3; ++i) {
    com_pos[i] = (AREAL)in[MPC_IN_COM_POS + i];
    com_vel[i] = (AREAL)in[MPC_IN_COM_VEL + i];
    rpy[i] = (AREAL)in[MPC_IN_RPY + i];
    normal[i] = (AREAL)in[MPC_IN_NORMAL + i];
    ang_vel[i] = (AREAL)in[MPC_IN_ANGVEL + i];
    des_pos[i] = (AREAL)in[mpc_in_des_pos(h) + i];
    des_vel[i] = (AREAL)in[mpc_in_des_vel(h) + i];
    des_rpy[i] = (AREAL)in[mpc_in_des_rpy(h) + i];
    des_ang[i] = (AREAL)in[mpc_in_des_angvel(h) + i];
  }
  for (int i = 0; i < 4; ++i) fric[i] = (AREAL)in[mpc_in_friction(h) + i];
  const double *contact = in + MPC_IN_CONTACT; /* [step][leg], mpc_osqp.cc:598-601 */
  const double *foot_body = in + mpc_in_footpos(h);

  /* mpc_osqp.cc:606-617: com_rotation = Rx(roll) * Ry(pitch) * Rz(yaw); foot_world = com_rotation * foot_base */
  AREAL rx[9], ry[9], rz[9], tmp[9], rot_xyz[9], foot_world[12];
  mpc_rot_x(rpy[0], rx);
  mpc_rot_y(rpy[1], ry);
  mpc_rot_z(rpy[2], rz);
  mpc_mat3_mul(rx, ry, tmp);
  mpc_mat3_mul(tmp, rz, rot_xyz);
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 3; ++r) {
      AREAL s = 0;
      for (int k = 0; k < 3; ++k) s += rot_xyz[r * 3 + k] * (AREAL)foot_body[3 * i + k];
      foot_world[3 * i + r] = s;
    }

  /* mpc_osqp.cc:630-633 state_ */
  AREAL *x0 = wk->x0;
  x0[0] = rpy[0]; x0[1] = rpy[1]; x0[2] = rpy[2];
  x0[3] = com_pos[0]; x0[4] = com_pos[1]; x0[5] = com_pos[2];
  x0[6] = ang_vel[0]; x0[7] = ang_vel[1]; x0[8] = ang_vel[2];
  x0[9] = com_vel[0]; x0[10] = com_vel[1]; x0[11] = com_vel[2];
  x0[12] = (AREAL)(-MPC_GRAVITY);

  /* mpc_osqp.cc:635-659 desired_states_ */
  for (int i = 0; i < h; ++i) {
    AREAL *d = wk->xref + 13 * i;
    d[0] = des_rpy[0];
    d[1] = des_rpy[1];
    d[2] = rpy[2] + mdl->dt * (i + 1) * des_ang[2];
    d[3] = mdl->dt * (i + 1) * des_vel[0] + com_pos[0];
    d[4] = mdl->dt * (i + 1) * des_vel[1] + com_pos[1];
    d[5] = des_pos[2];
    d[6] = des_ang[0]; d[7] = des_ang[1]; d[8] = des_ang[2];
    d[9] = des_vel[0]; d[10] = des_vel[1];
    d[11] = 0;
    d[12] = (AREAL)(-MPC_GRAVITY);
  }

  mpc_calc_a_mat(rpy, normal, wk->a_mat); /* :667 */

  /* mpc_osqp.cc:669-671: rotation_ = Rz*Ry*Rx; inv_inertia_world = R * inv_inertia * R^T */
  AREAL rot_zyx[9], rt[9], iw[9];
  mpc_mat3_mul(rz, ry, tmp);
  mpc_mat3_mul(tmp, rx, rot_zyx);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) rt[r * 3 + c] = rot_zyx[c * 3 + r];
  mpc_mat3_mul(rot_zyx, mdl->inv_inertia, tmp);
  mpc_mat3_mul(tmp, rt, iw);

  mpc_calc_b_mat(mdl->inv_mass, iw, foot_world, wk->b_mat);                 /* :673 */
  mpc_calc_exponentials(wk->a_mat, wk->b_mat, mdl->dt, wk->a_exp, wk->b_exp); /* :675 */

  /* mpc_osqp.cc:353-373 CalculateQpMats: A_qp blocks (last one left zero), anb_aux */
  memset(wk->a_qp, 0, sizeof(AREAL) * h * 169);
  memcpy(wk->a_qp, wk->a_exp, sizeof(AREAL) * 169);
  for (int i = 1; i < h - 1; ++i) mpc_mul13(wk->a_exp, wk->a_qp + (i - 1) * 169, 13, wk->a_qp + i * 169);
  memcpy(wk->anb, wk->b_exp, sizeof(AREAL) * 156);
  for (int i = 1; i < h; ++i) mpc_mul13(wk->a_exp, wk->anb + (i - 1) * 156, 12, wk->anb + i * 156);

  /* mpc_osqp.cc:387-434 P by block recursion.  Block (I,J) of P occupies rows 12I.., cols 12J.. */
#define PBLK(I, J, r, c) P[(size_t)(12 * (I) + (r)) * n + 12 * (J) + (c)]
  AREAL blk[144];
  for (int i = h - 1; i >= 0; --i) {
    mpc_xtqy(wk->anb + (h - i - 1) * 156, w, wk->b_exp, blk);
    for (int r = 0; r < 12; ++r)
      for (int c = 0; c < 12; ++c) PBLK(i, h - 1, r, c) = blk[r * 12 + c];
    if (i != h - 1)
      for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) PBLK(h - 1, i, r, c) = blk[c * 12 + r];
  }
  for (int i = h - 2; i >= 0; --i) {
    mpc_xtqy(wk->anb + (h - i - 1) * 156, w, wk->anb + (h - i - 1) * 156, blk);
    for (int r = 0; r < 12; ++r)
      for (int c = 0; c < 12; ++c) PBLK(i, i, r, c) = PBLK(i + 1, i + 1, r, c) + blk[r * 12 + c];
    for (int j = i + 1; j < h - 1; ++j) {
      mpc_xtqy(wk->anb + (h - i - 1) * 156, w, wk->anb + (h - j - 1) * 156, blk);
      for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) PBLK(i, j, r, c) = PBLK(i + 1, j + 1, r, c) + blk[r * 12 + c];
      for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) PBLK(j, i, r, c) = PBLK(i, j, c, r);
    }
  }
  for (size_t k = 0; k < (size_t)n * n; ++k) P[k] *= (AREAL)2.0;        /* :430 */
  for (int k = 0; k < n; ++k) P[(size_t)k * n + k] += mdl->alpha;       /* :431-434 */
#undef PBLK

  /* mpc_osqp.cc:681-683: state_diff = A_qp x0 - x_ref ; q = 2 B_qp^T (Q state_diff),
   * B_qp block (i,j) = anb[i-j] for j <= i (mpc_osqp.cc:375-385). */
  for (int i = 0; i < h; ++i)
    for (int r = 0; r < 13; ++r) {
      AREAL s = 0;
      for (int k = 0; k < 13; ++k) s += wk->a_qp[i * 169 + r * 13 + k] * x0[k];
      wk->sdiff[13 * i + r] = s - wk->xref[13 * i + r];
    }
  for (int j = 0; j < h; ++j)
    for (int c = 0; c < 12; ++c) {
      AREAL s = 0;
      for (int i = j; i < h; ++i) {
        const AREAL *bk = wk->anb + (i - j) * 156;
        for (int r = 0; r < 13; ++r) s += bk[r * 12 + c] * (w[r] * wk->sdiff[13 * i + r]);
      }
      q[12 * j + c] = 2 * s;
    }

  /* mpc_osqp.cc:685-688 + 449-477 CalculateConstraintBounds */
  const AREAL fz_max = mdl->mass * (AREAL)MPC_GRAVITY * (AREAL)MPC_MAX_SCALE;
  const AREAL fz_min = mdl->mass * (AREAL)MPC_GRAVITY * (AREAL)MPC_MIN_SCALE;
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < 4; ++j) {
      const int row = (i * 4 + j) * 5;
      const AREAL c = (AREAL)contact[i * 4 + j];
      const AREAL fub = (fric[0] + 1) * fz_max * c;
      lb[row] = lb[row + 1] = lb[row + 2] = lb[row + 3] = 0;
      lb[row + 4] = fz_min * c;
      ub[row] = ub[row + 1] = ub[row + 2] = ub[row + 3] = fub;
      ub[row + 4] = fz_max * c;
    }

  /* mpc_osqp.cc:437-447 UpdateConstraintsMatrix: one 5x3 block, repeated on the block diagonal */
  const AREAL cb[15] = {-1, 0, fric[0], 1, 0, fric[1], 0, -1, fric[2], 0, 1, fric[3], 0, 0, 1};
  memcpy(cone, cb, sizeof cb);
}

#endif /* CONVEX_MPC_ASSEMBLY_H */
