// mpc_wrench.h -- the OSQP iteration of one robot's convex-MPC QP, carried out in the space of the net body wrenches.
//
// Reference semantics (unchanged): extern/osqp/src auxil.c:164-228 (ADMM iteration), :243-362, 684-793 (residuals,
// termination), :13-77 (rho adaptation), polish.c (polish), osqp.c:354-519 (driver) on the QP that
// mpc_osqp.cc:578-796 builds, with the reference's settings (:705-712).  What is new is the LINEAR ALGEBRA behind
// OSQP's KKT solve  x~ = K^{-1} (sigma x - q + A^T (R z - y)),  K = P_s + sigma I + A_s^T R A_s  (n = 12 h).
//
// The single-rigid-body model only feels the NET WRENCH of the four foot forces of a step:
//     A_exp^k B_exp = Gamma_k B6,   B6 = [I_w^-1 [r_i]x ; I / m]  (6 x 12, the same for every step),
//     Gamma_k = [dt^2 (k + 1/2) That ; dt I6]
// so the QP Hessian is  P = alpha I + BB^T Theta BB  with BB = blockdiag(B6) (6 h x 12 h) and the 6 h x 6 h matrix
//     Theta_{jj'} = sum_{i >= max(j,j')} 2 Gamma_{i-j}^T Q Gamma_{i-j'} = s2(j,j') th1 + n(j,j') diag(th2)
// (th1, th2: QP record of the assembly kernel; s2, n: numbers that depend on the step indices only).  In OSQP's scaled
// variables  K = S + W^T (c Theta) W  with  W = BB D  and  S = c alpha D^2 + sigma I + A_s^T R A_s,  which is block diagonal
// with one 3 x 3 block per (step, foot).  Woodbury, in the form that stays accurate for every rho in [1e-6, 1e6]:
//     Z = W S^-1 W^T = blockdiag(Z_k) = L L^T  (6 x 6 per step),  T = L^-1,  G = T W  (per foot 6 x 3),
//     M = I + L^T (c Theta) L,     K^-1 b = v - S^-1 G^T (I - M^-1) (G v),  v = S^-1 b.
// M is well conditioned whatever rho is (the textbook form (c Theta + Z^-1)^-1 is not: Z is nearly singular along torques that
// only swing feet can produce, and the error of K^-1 b grows like rho).  The only dense object is the 6 h x 6 h matrix M:
// 60 x 60 at h = 10 instead of 120 x 120 -- an eighth of the factorisation work and a quarter of the matrix-vector work per
// ADMM iteration, exact (no swing-leg elimination, any gait), and small enough that ONE WAVEFRONT holds it: I - M^-1 lives as
// the lower triangle of an h x h grid of 6 x 6 register tiles, 55 tiles at h = 10 -> one tile per lane, no workgroup barrier
// anywhere in the solve (136 tiles + 64 foot lanes in 256 threads at h = 16, 210 tiles in 256 threads at h = 20).  Everything that belongs to one (step, foot)
// -- three force variables, five cone rows, their iterates x, z, y, the scaled cone block, bounds, S^-1, G_f -- lives in the
// registers of one "foot lane"; the wrench space is the only communication between lanes: six numbers per foot out (G_f v_f),
// six per step back.
//
// Polish (polish.c) is the same algebra on the reduced Hessian of the active set: H^-1 restricted to the null space of the
// active rows is  Xi - Xi G^T (I - M^-1) G Xi  with Xi = N (delta I + c alpha N^T D^2 N)^-1 N^T per foot in place of S^-1.
// There Z is only positive SEMI-definite (two point feet cannot produce a torque about the line through them): the L D L^T
// of a step drops its zero pivots (zero columns of L, zero rows of T), which leaves identity rows in M.
#pragma once

#include "mpc_core.h"
#include <type_traits>
#ifdef MPC_EMU_DEBUG
#include <vector>
#include <cstdio>
#include <cstdlib>
#endif

namespace mpc {

constexpr double kEpsAdmmFloor = 1e-9;   // exact mode: the tightest ADMM stage (what its first version ran from the start)

// The registers of a thread's role.  Where tile lanes and foot lanes are different threads (Cfg::FOOT0 != 0: h = 12, 16) a thread is one or
// the other, and the tile and the foot's iterate share their registers -- at the 256-register cap of the multi-wave workgroups the sum of
// both was what spilled.  (Every write to the foot members sits behind `t.foot`, every use of Mx behind `t.mact`.)
template <int TE, bool SHARED>
struct RoleRegs {
  double Mx[TE];
  double x[3], px[3], z[5], y[5], b[3], xt[3];   // iterates, P_s x, the right-hand side, x~
  double q[3], Si[6];                            // scaled q, S^{-1}   (cone block and bounds: Shared::fa)
};
template <int TE>
struct RoleRegs<TE, true> {
  union {
    double Mx[TE];
    struct {
      double x[3], px[3], z[5], y[5], b[3], xt[3];
      double q[3], Si[6];
    };
  };
};

template <int H>
struct WThread : RoleRegs<Cfg<H>::TE, (Cfg<H>::FOOT0 != 0 && MPC_SHARE_ROLE_REGS)> {
  using C = Cfg<H>;
  int tid;
  // ---- tile lane (tid < MTW): my tile of the wrench grid
  int ti, tj;
  bool mact, dia;
  // ---- foot lane: threads FOOT0 .. FOOT0 + NF - 1 (Cfg::FOOT0: 0 where a thread is tile lane and foot lane at once)
  int fid;                                       // foot index: step fid / 4, foot fid % 4 (the four lanes of a step are a hardware quad)
  bool foot;
  double w6[6];                                  // wrench exchange: my contribution out, the step's six numbers back
  double gq[2], yq[2], dlq[2];                   // rows j and j + 4 of my step (j = tid & 3): g, y, diag(M)^-1/2
  double zq[21];                                 // factorisation: W_f X_f W_f^T (packed lower triangle), then the step's sum
  int tyb;                                       // row types, two bits per row: 0 loose, 1 inequality, 2 equality (auxil.c:79-96)
  int sig;                                       // exact mode: my rows' active-set guess at the last check (base-3 code)
  // polish (foot lane)
  int act[5];
  int gslot[5], gfix, gbrow, gbside;            // exact mode (active_set): slot of each of my rows in the working set (-1: none); all my rows are equalities; my best candidate row
  double gx[3], gv[3], gbviol;                   //   the dual method's primal iterate, H^-1 n_p of my variables, my best candidate's violation
  double grn[5], gtol[5], gax[5], gred[1];       //   1 / |row|, violation tolerance and A x of my rows; operand of the workgroup reductions
  int gidx;                                      //   their index result
  static constexpr int kGaccT = C::TW <= 64 ? 3 : 0;      //   seed_inverse_mfma: up to 3 x 3 tiles of 16 x 16 (single-wavefront workgroups only)
  double gacc[kGaccT ? 4 * kGaccT * kGaccT : 1], gz[kGaccT ? kGaccT : 1], gnr[kGaccT ? kGaccT : 1];      //   my four elements of each tile; my element of L^-1 A_K,J / B^-1 A_K,J per tile column
  double pG[9], pC[9], pu0[3], pg[3], pr[3], pt[3], pxN[3], pPu[3], pw[3];
  double pQ[18], pR[21], pv[3], pn[6];           // orthogonalisation of the step's wrench columns (polish)
  double xp[3], zp[5], yp[5];
  MPC_HD void init(int id) {
    tid = id;
    fid = id - C::FOOT0;
    foot = fid >= 0 && fid < C::NF;
    mact = id < C::MTW;
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= id) ++r;
    ti = r; tj = id - r * (r + 1) / 2; dia = ti == tj;
  }
};

#ifndef MPC_STABLE_CHECKS         // exact mode: consecutive checks with an unchanged active-set guess before the polish is tried
#define MPC_STABLE_CHECKS 2
#endif
#ifndef MPC_EXACT_RHO_UPDATES
#define MPC_EXACT_RHO_UPDATES 10
#endif
#ifndef MPC_EPS_EXACT            // exact mode: optimality tolerance of an accepted active-set step (relative, OSQP's termination test)
#define MPC_EPS_EXACT 1e-10   // (the polished point of the right active set has a dual residual of ~1e-10 of the norms; 1e-11 rejects it, 1e-9 lets 0.04 % of the robots end 1e-6 off)
#endif
#ifndef MPC_EXACT_DIRECT          // exact mode: the active-set method's iterate is tested for optimality before any polish refinement
#define MPC_EXACT_DIRECT 1
#endif
#ifndef MPC_PAIR_SWEEP_MAXT     // the largest workgroup that sweeps two pivots per phase (see sweep_all)
#define MPC_PAIR_SWEEP_MAXT 64
#endif
#ifndef MPC_PAIR_SWEEP
#define MPC_PAIR_SWEEP(T) ((T) <= MPC_PAIR_SWEEP_MAXT)
#endif
#ifndef MPC_LOCKSTEP
#define MPC_LOCKSTEP 0
#endif
#ifndef MPC_ADMM_BATCH_LOADS   // which workgroup sizes request every LDS constant of the foot phase in one batch up front (~90 more registers)
#define MPC_ADMM_BATCH_LOADS(T) ((T) <= 64)
#endif
// LDS of the exact mode's active-set phase (Solver::active_set): the working set of the dual method and the inverse of its
// Gram matrix N^T H^-1 N, one slot per working constraint.
template <int H>
struct GiShared {
  static constexpr int NF = 4 * H;
  // slots (the optimum's active stance rows: 24 / 32 / 40 / 51 on average for configs 2 / 3 / 4 / 5, up to 34 / 44 / 54 / 67; a working set
  // holds linearly independent rows only, so 3 NF bounds it at the shortest horizons)
  static constexpr int NWMAX = H <= 3 ? 8 * ((3 * NF + 7) / 8) : H <= 6 ? 40 : H <= 10 ? 56 : (H <= 16 ? 72 : 88);
  static constexpr int TL = NWMAX > NF ? NWMAX : NF;
  alignas(16) double ci[NWMAX * (NWMAX + 1) / 2];   // packed lower triangle, (i, j <= i) at i (i + 1) / 2 + j; zero rows / columns at free slots
  // (the method's vectors -- d = N^T H^-1 n_p, the dual direction r, the multipliers, two scratch rows -- live in Shared::fr, which only
  // the ADMM iteration uses: with them here four workgroups would not fit a CU's 160 KB)
  int owner[NWMAX];                                 // foot * 8 + row of the slot's constraint, -1: free
  unsigned long long freem[2];                      // bit i: slot i is free
  int hi, p_foot, p_row, p_side, p_slot, k1, converged, fail, passes, adds, drops;
  int seed_shift, seed_same, seed_k, seeded;        // seeding (Solver::seed_working_set): which of the previous call's sets fits, its size, rows kept
  double p_viol, gamma, zeta, t1, lam_p, tstep;
};

#ifndef MPC_EXACT_REFINE
#define MPC_EXACT_REFINE 8    // refinement steps of the exact mode's polish before its first optimality test (83 % of the certified sets pass it at
                               // h = 10; the rest go round again, MPC_EXACT_ROUND_STEPS more steps each time; x 1.5 at the long horizons)
#endif
#ifndef MPC_EXACT_ROUND_STEPS
#define MPC_EXACT_ROUND_STEPS 4
#endif
#ifndef MPC_EXACT_ROUNDS
#define MPC_EXACT_ROUNDS 6
#endif
#ifndef MPC_GI_DELTA
#define MPC_GI_DELTA 0.0
#endif
#ifndef MPC_SEED_PIVOT            // exact mode, seeding: a seeded row whose pivot in the Gram matrix falls below this fraction of its diagonal entry is dropped as dependent
#define MPC_SEED_PIVOT 1e-8
#endif
#ifndef MPC_SEED_DIRECT_GRAM      // ... the seed's Gram matrix entry by entry from the held tiles (0: a column per seeded row, K applications of H^-1)
#define MPC_SEED_DIRECT_GRAM 1
#endif
#ifndef MPC_SEED_LAMBDA           // ... and one whose multiplier is below this fraction of the largest is dropped as not (clearly) active
#define MPC_SEED_LAMBDA 1e-6
#endif
#define MPC_V alignas(16) double
template <int H>
struct Shared {
  using C = Cfg<H>;
  static constexpr int RW = ((C::NF + 1) & ~1);                         // row stride of the residual scratch
  static constexpr int NRED = 21;                                       // residual / certificate reductions (Solver::residuals)
  static constexpr int PARTLEN_A0 = C::GW * C::NPW, PARTLEN_A1 = C::NW * (((C::GW + 1) & ~1) + MPC_PART_PAD);   // [slot][row] / [row][slot] (even row stride) partials
  static constexpr int PARTLEN_A = PARTLEN_A0 > PARTLEN_A1 ? PARTLEN_A0 : PARTLEN_A1, PARTLEN_B = NRED * RW;
  static constexpr int PARTLEN = PARTLEN_A > PARTLEN_B ? PARTLEN_A : PARTLEN_B;
  MPC_V B6[72]; MPC_V th1[36]; MPC_V th2[8];
  double c, cinv, rho, calpha;
  double rho3[4], rinv3[4];                             // rho and 1 / rho of a loose / inequality / equality row (index type + 1)
  MPC_V gh[C::NW + 2]; MPC_V dl[C::NW + 2];             // the tile product's input (dl g, or g for Theta products); dl = diag(M)^-1/2
  // per-foot constants in LDS, element k of foot f at [((k >> 1) NF + f) 2 + (k & 1)]: consecutive lanes read consecutive 16-byte
  // pairs (one conflict-free ds_read_b128 per pair)
  MPC_V fa[C::NF * 16];                                 // 0-8: the non-zeros of the scaled cone block, 9: l of row 4, 10-14: u of the five rows
  static constexpr int FR_GI = 3 * GiShared<H>::NWMAX + 2 * GiShared<H>::TL;           // the exact mode's vectors live here too (Solver::active_set)
  MPC_V fr[C::NF * 10 > FR_GI ? C::NF * 10 : FR_GI];    // 0-8: the cone block times rho of its row (factorisation)
  MPC_V Gf[C::NF * 18];                                 // per foot: G_f = T_k W_f (6 x 3) of the current factorisation
  MPC_V dxy[C::NF * 8];                                 // per foot: delta_x (0-2) and delta_y (3-7) of the last iteration before a check (auxil.c:187-228)
  // the published pivot rows are double buffered -- except where the workgroup is a single wavefront in lock step (MPC_LOCKSTEP:
  // the device build at h = 10), whose LDS instructions execute in program order: every lane has read the pair before any lane
  // publishes the next one
  static constexpr int NBUF = (MPC_LOCKSTEP && C::TW <= 64) ? 1 : 2;
  MPC_V prow_raw[NBUF][2][C::NW + 2];                   // [buffer][pivot of the pair][column]
  MPC_HD double *prow(int b, int r) { return prow_raw[b & (NBUF - 1)][r] + MPC_PROW_SKEW; }
  MPC_V piv_raw[NBUF][4];                               // [buffer]: the inverse of the pair's 2 x 2 pivot block (B00, B01, B11)
  MPC_HD double *piv(int b) { return piv_raw[b & (NBUF - 1)]; }
  union {
    MPC_V part[PARTLEN];                                // [slot][row] partial products of the tile mat-vec; residual scratch [14][RW]
    struct { MPC_V Lk[H * 36]; MPC_V Tk[H * 36]; };     // factorisation only: per step Z_k = L L^T, T = L^-1 (zero columns / rows at dropped pivots)
  };
  unsigned long long red[24];
  double dua_last;                                      // exact mode: the dual residual of the previous polish round (Solver::polish)
  int first, iter, status, status_polish, rho_updates, nfact, done, bad, pol_ok, pol_near, pol_rounds, sig_changed, loose_ok, dual_cand;
  double pri_res, dua_res, rho_new;
};
#undef MPC_V

#ifdef MPC_EMU_DEBUG
static long g_checks = 0, g_dual_cands = 0;   // host emulation only: how often the dual certificate's expensive part runs
#endif
template <int H, class Exec>
struct Solver {
  using C = Cfg<H>;
  using Th = WThread<H>;
  using Sh = Shared<H>;
  static constexpr int N = C::N, M = C::M, NF = C::NF, NW = C::NW, T = C::TW, TS = C::TS, G = C::GW, TE = C::TE, NP = C::NPW;

  Exec &ex;
  Sh &s;
  const RobotModel &mdl;
  double *state;       // [state_len<H>()]
  const double *qp;    // [QP_LEN]  q, bounds, cone, wrench description from the assembly kernel
  const double *sc;    // [SC_LEN]  D, E, q_s, c from the scaling kernel
  double *forces;      // [N]   out: -D x (all horizon steps), untouched on failure
  int *info;           // [kInfoLen]
  long long *prof;     // [kProfLen] shader-clock cycles per section (may be null)
  // Solver settings.  Defaults = the reference's OSQP call (mpc_osqp.cc:705-712).  exact(): the QP's optimum to working accuracy,
  // i.e. what the reference's qpOASES branch returns (mpc_osqp.cc:797-947; the caller clears the warm-start record: that branch
  // never warm-starts, :906-919).  The optimum of this strictly convex QP is unique, so the route to it is free (run<true>): ADMM
  // towards 1e-9, and as soon as OSQP's active-set guess (polish.c:36-52) has not changed over kStableChecks consecutive checks
  // (and the iterate passes the 1e-3 test) the polish is tried as an ACTIVE-SET step -- the equality-constrained solve on the guessed
  // set, ten refinement steps -- and accepted only if the polished point satisfies the optimality conditions to eps_exact (its z / y
  // are built by projection, so a wrong-signed multiplier or a violated inactive row shows up as a residual).  A rejected step costs a
  // polish and a re-factorisation of K and is not repeated until the guess has changed.  Rho updates are capped: ADMM run straight
  // to 1e-9 (the first version: ~600 iterations per robot) let rho oscillate for ever on an occasional robot.
  double eps_abs = kEpsAbs, eps_rel = kEpsRel, eps_exact = 0.0;
  int max_iter = kMaxIter, polish_refine = kPolishRefine, max_rho_updates = 1 << 30;
  bool polish_must_verify = false;   // (set around an early polish of the exact mode)
  bool act_given = false;            // polish(): the active set is in t.act already (active_set) instead of OSQP's guess from (z, y)
  bool xn_given = false;             // polish<true>(): Shared::dxy holds the candidate optimum of that set (active_set's iterate): checked first, refined only if the check fails
  GiShared<H> *gi = nullptr;         // LDS of the exact mode's active-set phase (null in the OSQP mode)
  int *seedrec = nullptr;            // [NF] exact mode: the working set the previous call of this robot ended on (one code per foot, seed_code), or null: start empty
  static constexpr int kStableChecks = MPC_STABLE_CHECKS;
  MPC_HD void exact() { eps_exact = MPC_EPS_EXACT; eps_abs = eps_rel = kEpsAdmmFloor; max_iter = 5 * kMaxIter; polish_refine = H > 10 ? MPC_EXACT_REFINE + MPC_EXACT_REFINE / 2 : MPC_EXACT_REFINE; max_rho_updates = MPC_EXACT_RHO_UPDATES; }   // then run<true>()
#ifdef MPC_EMU_DEBUG
  double *dbg = nullptr;   // host emulation only: per foot 20 doubles of the first polish application (tests/emu)
#endif
  using Tv = TileView;
  long long tc[kProfLen] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = 0;
#ifndef MPC_SECTION_PROFILE
  MPC_HD void lap(int) {}
#else
  MPC_HD void lap(int k) { const long long now = MPC_CLOCK(); tc[k] += now - tlast; tlast = now; }
#endif

  static MPC_HD double fast_recip(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);            // v_rcp_f64 + two Newton steps (full double accuracy, not IEEE-rounded)
    r = r * (2.0 - d * r);
    r = r * (2.0 - d * r);
    return r;
#else
    return 1.0 / d;
#endif
  }
  // Theta_{ti,tj} = s2 th1 + nn diag(th2):  s2 = sum_{i < m} (i + 1/2)(i + d + 1/2) = m (4 m^2 - 1) / 12 + d m^2 / 2,  nn = m = H - ti,  d = ti - tj
  static MPC_HD double th_nn(const Th &t) { return (double)(H - t.ti); }
  static MPC_HD double th_s2(const Th &t) { const double m = (double)(H - t.ti), d = (double)(t.ti - t.tj); return m * (4.0 * m * m - 1.0) / 12.0 + d * (m * m) * 0.5; }
  MPC_HD double Dat(const Th &t, int c) const { return sc[C::SC_D + 3 * t.fid + c]; }   // D of my variables (scale record; not worth registers)
  // rho / 1 / rho of row r of my foot (three uniform values, selected by the row's type code)
  MPC_HD double rho_at(const Th &t, int r) const {
    const int ty = (t.tyb >> (2 * r)) & 3;
    const double r0 = s.rho3[0], r1 = s.rho3[1], r2 = s.rho3[2];
    return ty == 2 ? r2 : (ty == 1 ? r1 : r0);
  }
  MPC_HD double rinv_at(const Th &t, int r) const {
    const int ty = (t.tyb >> (2 * r)) & 3;
    const double r0 = s.rinv3[0], r1 = s.rinv3[1], r2 = s.rinv3[2];
    return ty == 2 ? r2 : (ty == 1 ? r1 : r0);
  }
  // ---- the scaled cone block of a foot: rows (a0, 0, a1) (a2, 0, a3) (0, a4, a5) (0, a6, a7) (0, 0, a8)  (mpc_osqp.cc:437-447) ----
  static MPC_HD void a_mul(const double *a, const double *v, double *out) {      // out[5] = A_f v
    out[0] = a[0] * v[0] + a[1] * v[2];
    out[1] = a[2] * v[0] + a[3] * v[2];
    out[2] = a[4] * v[1] + a[5] * v[2];
    out[3] = a[6] * v[1] + a[7] * v[2];
    out[4] = a[8] * v[2];
  }
  static MPC_HD void at_mul(const double *a, const double *w, double *out) {     // out[3] = A_f^T w
    out[0] = a[0] * w[0] + a[2] * w[1];
    out[1] = a[4] * w[2] + a[6] * w[3];
    out[2] = (((a[1] * w[0] + a[3] * w[1]) + a[5] * w[2]) + a[7] * w[3]) + a[8] * w[4];
  }
  static MPC_HD void sym3_mul(const double *m, const double *v, double *out) {   // packed (00 01 02 11 12 22)
    out[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    out[1] = m[1] * v[0] + m[3] * v[1] + m[4] * v[2];
    out[2] = m[2] * v[0] + m[4] * v[1] + m[5] * v[2];
  }
  // inverse of a symmetric positive definite 3 x 3 matrix (packed) by LDL^T
  static MPC_HD void sym3_inv(const double *m, double *inv) {
    const double i0 = fast_recip(m[0]);
    const double l10 = m[1] * i0, l20 = m[2] * i0;
    const double d1 = m[3] - l10 * m[1];
    const double i1 = fast_recip(d1);
    const double t21 = m[4] - l20 * m[1];
    const double l21 = t21 * i1;
    const double d2 = m[5] - l20 * m[2] - l21 * t21;
    const double i2 = fast_recip(d2);
    // L^-1 = [1 0 0; -l10 1 0; l10 l21 - l20, -l21, 1]
    const double k20 = l10 * l21 - l20;
    inv[5] = i2;
    inv[4] = -l21 * i2;
    inv[2] = k20 * i2;
    inv[3] = i1 + l21 * l21 * i2;
    inv[1] = -l10 * i1 - l21 * k20 * i2;
    inv[0] = i0 + l10 * l10 * i1 + k20 * k20 * i2;
  }
  // my foot's wrench map W_f = B6[:, 3 j .. 3 j + 2] diag(D)  (6 x 3, row-major in w[18])
  MPC_HD void foot_w(const Th &t, double *w) const {
    const int j = t.fid & 3;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) w[3 * r + c] = s.B6[12 * r + 3 * j + c] * Dat(t, c);
  }
  // t.w6 = W_f v  (my contribution to the step's wrench)
  MPC_HD void put_wrench(Th &t, const double *v) const {
    double w[18];
    foot_w(t, w);
#pragma unroll
    for (int r = 0; r < 6; ++r) t.w6[r] = w[3 * r] * v[0] + w[3 * r + 1] * v[1] + w[3 * r + 2] * v[2];
  }
  // out[3] = W_f^T y_k   (y of my step: t.w6 after recv)
  MPC_HD void get_wrench(const Th &t, double *out) const {
    double w[18];
    foot_w(t, w);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double acc = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) acc += w[3 * r + c] * t.w6[r];
      out[c] = acc;
    }
  }
  // out = T_k w  (T lower triangular, w 6 x 3)
  MPC_HD void mul_tk(const Th &t, const double *w, double *out) const {
    const double *tk = s.Tk + 36 * (t.fid >> 2);
    double o[18];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double v = 0;
#pragma unroll
        for (int k = 0; k <= r; ++k) v += tk[6 * r + k] * w[3 * k + c];
        o[3 * r + c] = v;
      }
#pragma unroll
    for (int k = 0; k < 18; ++k) out[k] = o[k];
  }
  // the same with the factorisation's G_f = T_k W_f: t.w6 = G_f v,  out = G_f^T y_k.  G_f sits in LDS and is fetched where it
  // is used (volatile 64-bit loads, see foot_a: 36 VGPRs the iteration loop does not have)
  static constexpr MPC_HD int pidx(int k, int f) { return ((k >> 1) * NF + f) * 2 + (k & 1); }   // pair layout (see Shared::fa)
  MPC_HD void load_g(const Th &t, double *gf) const {
#pragma unroll
    for (int k = 0; k < 18; k += 2) MPC_LDS_LOAD128(s.Gf + pidx(k, t.fid), gf[k], gf[k + 1]);
  }
  MPC_HD void put_g(Th &t, const double *v) const {
    double gf[18];
    load_g(t, gf);
#pragma unroll
    for (int r = 0; r < 6; ++r) t.w6[r] = gf[3 * r] * v[0] + gf[3 * r + 1] * v[1] + gf[3 * r + 2] * v[2];
  }
  MPC_HD void get_g(const Th &t, double *out) const {
    double gf[18];
    load_g(t, gf);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double acc = 0;
#pragma unroll
      for (int r = 0; r < 6; ++r) acc += gf[3 * r + c] * t.w6[r];
      out[c] = acc;
    }
  }

  // ---- wrench-space products.  The four feet of a step are the four lanes of a quad, so the step's sums and broadcasts are
  // register operations (DPP quad permutes, Exec::quad_allsum / quad_gather6): no LDS round trip.  Lane j of the quad owns rows
  // j and j + 4 (< 6) of its step.
  //   send:  t.w6 (per foot) -> g_k = sum over the quad; the row owners publish the tile product's input to LDS
  //   tile_product: part <- tile partials                                   (the only phase with an LDS hand-over)
  //   recv:  the row owners combine their rows, the quad gathers all six -> t.w6 = (result)_k
  enum { kHeld = 0, kTheta = 1 };
  // part <- partial products of the tile with v, both orientations (tile (i, j), j < i, stands for itself and its transpose)
  template <int KIND>
  MPC_HD void tile_product() {
    const double *v = s.gh;
    ex.par([&](Th &t) {
      if (t.mact) {
        const double s2v = th_s2(t), nnv = th_nn(t);
        double vc[TS], vr[TS], ar[TS], ac[TS];
#pragma unroll
        for (int bb = 0; bb < TS; ++bb) { vc[bb] = v[TS * t.tj + bb]; vr[bb] = v[TS * t.ti + bb]; ar[bb] = 0; ac[bb] = 0; }
#pragma unroll
        for (int aa = 0; aa < TS; ++aa)
#pragma unroll
          for (int bb = 0; bb < TS; ++bb) {
            double m;
            if (KIND == kHeld) m = t.Mx[aa * TS + bb];
            else m = s.c * (s2v * s.th1[aa * TS + bb] + (aa == bb ? nnv * s.th2[aa] : 0.0));
            ar[aa] += m * vc[bb];
            ac[bb] += m * vr[aa];
          }
        if constexpr (kPartRowMajor) {
          // [row][slot]: the G partials of a row are contiguous, and sum_parts fetches them as G / 2 ds_read_b128 (the [slot][row] layout
          // costs the reader G ds_read_b64; the writer's stride-G stores pair up as ds_write2_b64 either way)
          double *pd = s.part + (TS * t.ti) * GP + t.tj, *pt = s.part + (TS * t.tj) * GP + t.ti;
#pragma unroll
          for (int aa = 0; aa < TS; ++aa) pd[aa * GP] = ar[aa];
          if (!t.dia) {
#pragma unroll
            for (int bb = 0; bb < TS; ++bb) pt[bb * GP] = ac[bb];
          }
        } else {
          double *pd = s.part + t.tj * NP + TS * t.ti, *pt = s.part + t.ti * NP + TS * t.tj;
#pragma unroll
          for (int aa = 0; aa < TS; ++aa) pd[aa] = ar[aa];
          if (!t.dia) {
#pragma unroll
            for (int bb = 0; bb < TS; ++bb) pt[bb] = ac[bb];
          }
        }
      }
    });
  }
  // The partial products of the tile mat-vec in LDS: [row][slot] in the single-wavefront kernels (h <= 10), [slot][row] in the multi-wave ones --
  // at G = 16 a row of the [row][slot] layout is 128 bytes, the readers' 16-byte loads of neighbouring rows fall on the same banks (38 % of
  // the h = 16 kernel's LDS cycles were bank conflicts) and the plain layout is 6 % faster end to end (profiles/r05_ab_long_horizon_lds_layout.txt)
  static constexpr bool kPartRowMajor = MPC_PART_ROWMAJOR(T);
  static constexpr int GP = ((G + 1) & ~1) + MPC_PART_PAD;   // row stride of the [row][slot] layout: even, so that a row starts on a 16-byte boundary
  static_assert(NW * GP <= Sh::PARTLEN, "the partial products must fit Shared::part");
  static MPC_HD double sum_parts(const Sh &s, int row) {   // fixed pairwise order
    double v[GP];
    if constexpr (kPartRowMajor) {
#pragma unroll
      for (int k = 0; k < ((G + 1) & ~1); k += 2) MPC_LDS_LOAD128(s.part + row * GP + k, v[k], v[k + 1]);
    } else {
#pragma unroll
      for (int k = 0; k < G; ++k) v[k] = MPC_LDS_LOAD64(s.part + k * NP + row);
    }
#pragma unroll
    for (int w = 1; w < G; w *= 2)
#pragma unroll
      for (int k = 0; k + w < G; k += 2 * w) v[k] = v[k] + v[k + w];
    return v[0];
  }
  // KIND = kHeld: the tiles hold -Mh^-1 (+2 on the diagonal, see sweep_all) of the unit-diagonal Mh = dl M dl, and the result is
  // (I - M^-1) g = g - dl Mh^-1 (dl g).  KIND = kTheta: c Theta g.
  template <int KIND>
  MPC_HD void send() {
#if MPC_QUAD_SCATTER
    ex.quad_scatter6([](Th &t) { return t.w6; }, [](Th &t) { return t.gq; });
#else
    ex.template quad_allsum<6>([](Th &t) { return t.w6; });
#endif
    ex.par([&](Th &t) {
      if (t.foot) {
        const int k = t.fid >> 2, j = t.fid & 3;
#if MPC_QUAD_SCATTER
        const double g0 = t.gq[0], g1 = t.gq[1];
#else
        const double g0 = j == 0 ? t.w6[0] : (j == 1 ? t.w6[1] : (j == 2 ? t.w6[2] : t.w6[3]));
        const double g1 = j == 0 ? t.w6[4] : t.w6[5];
        t.gq[0] = g0; t.gq[1] = g1;
#endif
        s.gh[6 * k + j] = KIND == kHeld ? t.dlq[0] * g0 : g0;
        if (j < 2) s.gh[6 * k + 4 + j] = KIND == kHeld ? t.dlq[1] * g1 : g1;
      }
    });
  }
  template <int KIND>
  MPC_HD void recv() {
    ex.seq([&](Th &t) {
      if (t.foot) {
        const int k = t.fid >> 2, j = t.fid & 3;
        const double s0 = sum_parts(s, 6 * k + j), s1 = sum_parts(s, 6 * k + 4 + (j & 1));   // (lanes 2, 3 have no second row: ignored)
        if (KIND == kHeld) {
          t.yq[0] = t.gq[0] - t.dlq[0] * (2.0 * (t.dlq[0] * t.gq[0]) - s0);
          t.yq[1] = t.gq[1] - t.dlq[1] * (2.0 * (t.dlq[1] * t.gq[1]) - s1);
        } else { t.yq[0] = s0; t.yq[1] = s1; }
      }
    });
    ex.quad_gather6([](Th &t) { return t.yq; }, [](Th &t) { return t.w6; });
  }
  // t.w6 <- (I - M^-1) (sum of the foot contributions t.w6)   /   c Theta (...)
  template <int KIND>
  MPC_HD void product() {
    send<KIND>();
    tile_product<KIND>();
    recv<KIND>();
  }

  // ================================ 1. load ======================================================================
  // (The state record crosses between the two jobs of a solve with device-coherent accesses, MPC_GLD / MPC_GST, which go past this
  // XCD's L2: it is moved with lane-contiguous addresses -- whole lines per instruction -- through an LDS stage (s.part, free at both
  // ends of a job); a foot lane fetching its own 3 + 5 + 5 values straight from HBM touched every line of the record several times.)
  static constexpr int SL = 2 * N + 2 * M + 2;   // state record: x[N] z[M] y[M] q_old[N] rho flag
  static_assert(SL + N <= Sh::PARTLEN, "the state / force stage must fit Shared::part");
  MPC_HD void load() {
    ex.par([&](Th &t) {
      for (int i = t.tid; i < N + 2 * M; i += T) s.part[i] = MPC_GLD(state + i);      // x, z, y (the previous q in between is the prep kernel's business)
      if (t.tid < 2) s.part[2 * N + 2 * M + t.tid] = MPC_GLD(state + 2 * N + 2 * M + t.tid);      // rho, flag
      for (int i = t.tid; i < 72; i += T) s.B6[i] = qp[C::QP_B6 + i];
      for (int i = t.tid; i < 36; i += T) s.th1[i] = qp[C::QP_TH1 + i];
      for (int i = t.tid; i < 6; i += T) s.th2[i] = qp[C::QP_TH2 + i];
      if (t.foot) {
        const int f = t.fid;
#pragma unroll
        for (int c = 0; c < 3; ++c) t.q[c] = sc[C::SC_QS + 3 * f + c];
#pragma unroll
        for (int k = 0; k < 9; ++k) s.fa[pidx(k, f)] = C::scaled_cone_entry(qp, sc, f, k);
        s.fa[pidx(15, f)] = 0.0;
        const double *bnd = qp + C::QP_BND + 3 * f;
        int tyb = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          // l_s = E l, u_s = E u (scaling.c:152-153) from the foot's three bound values (l of rows 0-3 is 0)
          const double e = sc[C::SC_E + 5 * f + r];
          const double lo = e * (r < 4 ? 0.0 : bnd[0]), hi = e * (r < 4 ? bnd[1] : bnd[2]);
          s.fa[pidx(10 + r, f)] = hi;
          if (r == 4) s.fa[pidx(9, f)] = lo;
          // set_rho_vec / update_rho_vec (auxil.c:79-141): the row type is a function of the scaled bounds
          const int ty = (lo < -kInfty * kMinScaling && hi > kInfty * kMinScaling) ? 0 : (hi - lo < kRhoTol ? 2 : 1);
          tyb |= ty << (2 * r);
          if (lo > hi) tyb |= 1 << 10;     // OSQP refuses such data (validate_data, auxil.c:795-830; the reference then has no workspace)
        }
        t.tyb = tyb;
      }
      if (t.tid == 0) {
        s.c = sc[C::SC_C]; s.cinv = sc[C::SC_C + 1]; s.calpha = sc[C::SC_C] * mdl.alpha;
        s.status = kStUnsolved; s.status_polish = 0; s.rho_updates = 0; s.nfact = 0; s.iter = 0; s.done = 0; s.bad = 0; s.pol_ok = 0; s.sig_changed = 0; s.loose_ok = 0;
      }
    });
    ex.par([&](Th &t) {
      if (t.foot) {
        const int f = t.fid;
#pragma unroll
        for (int c = 0; c < 3; ++c) t.x[c] = s.part[3 * f + c];      // scaled iterates of the previous call; zeros on the first call
#pragma unroll
        for (int r = 0; r < 5; ++r) { t.z[r] = s.part[N + 5 * f + r]; t.y[r] = s.part[N + M + 5 * f + r]; }
        if (t.tyb >> 10) s.bad = 1;
      }
      if (t.tid == 0) {
        const bool first = s.part[2 * N + 2 * M + 1] == 0.0;
        s.first = first;
        s.rho = first ? kRho0 : s.part[2 * N + 2 * M];
      }
    });
    ex.par([](Th &) {});      // (the stage is read before anything else writes s.part)
    lap(0);
  }
  // my foot's constants, fetched where they are used (volatile 64-bit LDS loads: the compiler neither hoists them out of the
  // iteration loop nor keeps them in registers across it -- 30 VGPRs the loop does not have)
  MPC_HD void foot_a(const Th &t, double *a) const {
    double dummy;
#pragma unroll
    for (int k = 0; k < 8; k += 2) MPC_LDS_LOAD128(s.fa + pidx(k, t.fid), a[k], a[k + 1]);
    MPC_LDS_LOAD128(s.fa + pidx(8, t.fid), a[8], dummy);
  }
  MPC_HD void foot_ar(const Th &t, double *a) const {   // the cone block times rho of its row
    double dummy;
#pragma unroll
    for (int k = 0; k < 8; k += 2) MPC_LDS_LOAD128(s.fr + pidx(k, t.fid), a[k], a[k + 1]);
    MPC_LDS_LOAD128(s.fr + pidx(8, t.fid), a[8], dummy);
  }
  MPC_HD void foot_bounds(const Th &t, double *lo, double *up) const {   // rows 0-3 have l = 0 (mpc_osqp.cc:449-477)
    double dummy;
#pragma unroll
    for (int r = 0; r < 4; ++r) lo[r] = 0.0;
    MPC_LDS_LOAD128(s.fa + pidx(8, t.fid), dummy, lo[4]);
    MPC_LDS_LOAD128(s.fa + pidx(10, t.fid), up[0], up[1]);
    MPC_LDS_LOAD128(s.fa + pidx(12, t.fid), up[2], up[3]);
    MPC_LDS_LOAD128(s.fa + pidx(14, t.fid), up[4], dummy);
  }

  MPC_HD void set_rho_vec() {   // rho per row type (auxil.c:79-96, osqp.c:1267-1310)
    ex.par([&](Th &t) {
      if (t.tid < 3) {
        const double rv = t.tid == 0 ? kRhoMin : (t.tid == 2 ? kRhoEqOverIneq * s.rho : s.rho);
        s.rho3[t.tid] = rv;
        s.rinv3[t.tid] = 1.0 / rv;
      }
    });
  }

  // ---- 6 x 6 helpers of the step lanes (packed lower triangle: index r (r + 1) / 2 + c, c <= r; all indices static) --------
  static constexpr MPC_HD int pk(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
  // L D L^T of a symmetric positive semi-definite 6 x 6 (unit lower L in l[], pivots in d[]).  A pivot that is <= tol times its
  // own diagonal entry -- row j lies in the span of the rows before it (sin^2 of the angle <= tol) -- is dropped: d = 0, column
  // of L = 0.  (The test must be relative to the row's OWN diagonal: a wrench component that is small for every foot is not
  // dependent, and dropping it would discard off-diagonal entries of relative size sqrt(tol).)
  static MPC_HD void ldl6(const double *zz, double *l, double *d, double tol) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double dj = zz[pk(j, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) dj -= l[pk(j, k)] * l[pk(j, k)] * d[k];
      const bool ok = dj > tol * zz[pk(j, j)];
      d[j] = ok ? dj : 0.0;
      const double inv = ok ? fast_recip(dj) : 0.0;
      l[pk(j, j)] = 1.0;
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        double v = zz[pk(i, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) v -= l[pk(i, k)] * l[pk(j, k)] * d[k];
        l[pk(i, j)] = v * inv;
      }
    }
  }

  // per foot: t.zq <- w X w^T for a symmetric 3 x 3 X (packed) and the 6 x 3 map w = W_f
  MPC_HD void put_zf(Th &t, const double *X, const double *w) const {
    double v[18];
#pragma unroll
    for (int r = 0; r < 6; ++r) sym3_mul(X, w + 3 * r, v + 3 * r);   // V = W X (X symmetric)
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) t.zq[pk(r, c)] = v[3 * r] * w[3 * c] + v[3 * r + 1] * w[3 * c + 1] + v[3 * r + 2] * w[3 * c + 2];
  }

  // ================================ 2. factorisation: Mx <- -Mh^-1,  Mh = dl (I + L^T (c Theta) L) dl =================
  // xs(t): the symmetric 3 x 3 matrix X_f of the foot (S_f^-1 for the ADMM system, Xi_f for polish); Z_k = sum_f W_f X_f W_f^T
  static MPC_HD double fast_rsqrt(double d) {   // 1 / sqrt(d), d > 0 finite: v_rsq_f64 + two Newton steps
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(d);
    const double hh = 0.5 * d;
    r = r * (1.5 - hh * r * r);
    r = r * (1.5 - hh * r * r);
    return r;
#else
    return 1.0 / sqrt(d);
#endif
  }
  // Lane 0 of every quad: Z_k = sum of the feet's t.zq = L_u D L_u^T;  L = L_u D^1/2 -> Lk,  T = D^-1/2 L_u^-1 -> Tk.
  // (Z_k of the ADMM system is positive definite; a row that is numerically dependent all the same is dropped.)
  MPC_HD void step_factor() {
    ex.template quad_allsum<21>([](Th &t) { return t.zq; });
    ex.par([&](Th &t) {
      if (t.foot && (t.fid & 3) == 0) {
        const int kk = t.fid >> 2;
        const double *zz = t.zq;
        double l[21], d[6], li[21];
        double mxd = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) mxd = dmax(mxd, zz[pk(j, j)]);
        if (!(mxd < kInfty)) s.bad = 1;   // (NaN / inf inputs)
        ldl6(zz, l, d, 1e-13);
#pragma unroll
        for (int j = 0; j < 6; ++j) {   // li = L_u^-1 (unit lower)
          li[pk(j, j)] = 1.0;
#pragma unroll
          for (int i = j + 1; i < 6; ++i) {
            double v = l[pk(i, j)];
#pragma unroll
            for (int k = j + 1; k < i; ++k) v += l[pk(i, k)] * li[pk(k, j)];
            li[pk(i, j)] = -v;
          }
        }
        double sd[6], si[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const bool ok = d[j] > 0;
          si[j] = ok ? fast_rsqrt(ok ? d[j] : 1.0) : 0.0;
          sd[j] = ok ? d[j] * si[j] : 0.0;
        }
        double *ol = s.Lk + 36 * kk, *ot = s.Tk + 36 * kk;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            ol[6 * i + j] = i >= j ? l[pk(i, j)] * sd[j] : 0.0;
            ot[6 * i + j] = i >= j ? li[pk(i, j)] * si[i] : 0.0;
          }
      }
    });
  }
  // The ADMM system: X_f = S_f^-1.  Gram-matrix orthogonalisation (L D L^T of Z_k = F^T F) loses eps cond(Z_k) in the range of
  // F; with |S^-1| W^T Theta W of order 1e3 at most that is harmless here (tests/test_emulated_kernel.py: K^-1 b agrees with a
  // dense solve to 1e-15 for rho from 1e-5 to 1e3).  Polish, where 1 / delta stands in for S^-1, orthogonalises F itself.
  template <class XS>
  MPC_HD void factor_core(XS &&xs) {
    ex.seq([&](Th &t) {
      if (t.foot) {
        double w[18];
        foot_w(t, w);
        put_zf(t, xs(t), w);
      }
    });
    step_factor();
    ex.par([&](Th &t) {
      if (t.foot) {   // G_f = T_k W_f
        double w[18], gf[18];
        foot_w(t, w);
        mul_tk(t, w, gf);
        // G_f X_f is what the iteration uses:  x~ = X b - (G X)^T y_w,  the wrench of the next right-hand side = (G X) b
        double hs[18];
#pragma unroll
        for (int r = 0; r < 6; ++r) sym3_mul(xs(t), gf + 3 * r, hs + 3 * r);
#pragma unroll
        for (int k = 0; k < 18; ++k) s.Gf[pidx(k, t.fid)] = hs[k];
      }
    });
    factor_tail();
  }
  // from L_k (LDS) to the held tiles: dl = diag(M)^-1/2 of my rows, the unit-diagonal tile of Mh, the sweep
  MPC_HD void factor_tail() {
    ex.par([&](Th &t) {
      if (t.foot) {   // M_rr = 1 + l_r^T (c Theta_kk) l_r  (l_r: column r of L_k)
        const int k = t.fid >> 2, j = t.fid & 3;
        const double mm = (double)(H - k), s2kk = mm * (4.0 * mm * mm - 1.0) / 12.0;   // sum_{i < m} (i + 1/2)^2
        const double *lk = s.Lk + 36 * k;
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
          const int r = slot == 0 ? j : 4 + (j & 1);
          double lc[TS];
#pragma unroll
          for (int a = 0; a < TS; ++a) lc[a] = lk[6 * a + r];
          double acc = 0;
#pragma unroll
          for (int a = 0; a < TS; ++a) {
            double row = 0;
#pragma unroll
            for (int bb = 0; bb < TS; ++bb) row += (s2kk * s.th1[a * TS + bb] + (a == bb ? mm * s.th2[a] : 0.0)) * lc[bb];
            acc += lc[a] * row;
          }
          const double dv = fast_rsqrt(1.0 + s.c * acc);
          t.dlq[slot] = dv;
          if (slot == 0 || j < 2) s.dl[6 * k + r] = dv;
        }
      }
    });
    ex.par([&](Th &t) {
      if (t.mact) {   // Mh tile = dl_i ([i == j] I + L_i^T (c Theta_ij) L_j) dl_j   (L lower triangular); unit diagonal
        const double s2v = th_s2(t), nnv = th_nn(t);
        const double *li = s.Lk + 36 * t.ti, *lj = s.Lk + 36 * t.tj, *di = s.dl + 6 * t.ti, *dj = s.dl + 6 * t.tj;
        double t1[TE];
#pragma unroll
        for (int aa = 0; aa < TS; ++aa)
#pragma unroll
          for (int bb = 0; bb < TS; ++bb) {
            double v = 0;
#pragma unroll
            for (int k = bb; k < TS; ++k) v += (s.c * (s2v * s.th1[aa * TS + k] + (aa == k ? nnv * s.th2[aa] : 0.0))) * lj[k * TS + bb];
            t1[aa * TS + bb] = v;
          }
#pragma unroll
        for (int aa = 0; aa < TS; ++aa)
#pragma unroll
          for (int bb = 0; bb < TS; ++bb) {
            double v = (t.dia && aa == bb) ? 1.0 : 0.0;
#pragma unroll
            for (int k = aa; k < TS; ++k) v += li[k * TS + aa] * t1[k * TS + bb];
            t.Mx[aa * TS + bb] = (di[aa] * v) * dj[bb];
          }
      }
    });
    lap(6);
    sweep_all();
    ex.par([&](Th &t) { if (t.tid == 0) s.nfact++; });
    lap(7);
  }
  MPC_HD void factor() {
    ex.par([&](Th &t) {
      if (t.foot) {   // S_f = c alpha D^2 + sigma I + A_f^T R A_f  ->  S_f^-1
        double rv[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) rv[r] = rho_at(t, r);
        double a[9];
        foot_a(t, a);
        double S[6];
        S[0] = (s.calpha * Dat(t, 0) * Dat(t, 0) + kSigma) + (rv[0] * a[0] * a[0] + rv[1] * a[2] * a[2]);
        S[1] = 0.0;
        S[2] = rv[0] * a[0] * a[1] + rv[1] * a[2] * a[3];
        S[3] = (s.calpha * Dat(t, 1) * Dat(t, 1) + kSigma) + (rv[2] * a[4] * a[4] + rv[3] * a[6] * a[6]);
        S[4] = rv[2] * a[4] * a[5] + rv[3] * a[6] * a[7];
        S[5] = (s.calpha * Dat(t, 2) * Dat(t, 2) + kSigma) + ((((rv[0] * a[1] * a[1] + rv[1] * a[3] * a[3]) + rv[2] * a[5] * a[5]) + rv[3] * a[7] * a[7]) + rv[4] * a[8] * a[8]);
        sym3_inv(S, t.Si);
        // the cone block times rho of its row, for the iteration's A^T R (.)
        const double ar[9] = {a[0] * rv[0], a[1] * rv[0], a[2] * rv[1], a[3] * rv[1], a[4] * rv[2], a[5] * rv[2], a[6] * rv[3], a[7] * rv[3], a[8] * rv[4]};
#pragma unroll
        for (int k = 0; k < 9; ++k) s.fr[pidx(k, t.fid)] = ar[k];
        s.fr[pidx(9, t.fid)] = 0.0;
      }
    });
    factor_core([](Th &t) { return t.Si; });
  }

  // Symmetric sweep of every pivot, two pivots at a time: after all pivots the matrix equals -inverse.  Per pair K = {k, k+1} with
  // B = A_KK^-1:  A_ij -= A_iK B A_Kj (i, j outside K);  A_iK -> A_iK B;  A_KK -> -B.  The matrix stays symmetric, so both A_iK
  // and A_Kj are read from the two published pivot rows, and only the lower-triangle tiles are updated.  The published rows carry
  // A_KK - I in the slots of K, which makes the generic update  a_ij -= [u_i v_i] B [u_j v_j]^T  produce A_iK B on the pair's
  // columns and B A_Kj on its rows with no per-element select ((A_KK - I) B = I - B); the diagonal elements of a swept pair take
  // the generic update too and end up as (true value + 2), the off-diagonal one as -B01 exactly.
  // One LDS round trip and one reciprocal (of the block's determinant) per pair: a wave that runs alone on its SIMD has nobody
  // to hide that latency behind, and it was most of the 540 cycles per pivot of the one-pivot form.
  // The loop over the pairs of a tile row is unrolled so that the pair's position inside its tile is static.
  static constexpr bool kPairSweep = MPC_PAIR_SWEEP(T);
  MPC_HD void sweep_all() {
    static_assert(TS % 2 == 0, "pairs must not straddle tiles");
    if constexpr (!kPairSweep) { sweep_all_single(); return; }
    int buf = 0;
    ex.par([&](Th &t) { if (t.mact) publish<0>(t, 0, 0); });
    for (int kt = 0; kt < G; ++kt) sweep_pairs<0>(kt, buf);
  }
  template <int A>
  MPC_HD void sweep_pairs(int kt, int &buf) {
    if constexpr (A < TS) {
      sweep_pair<A>(kt, buf);
      buf ^= 1;
      sweep_pairs<A + 2>(kt, buf);
    }
  }
  template <int A>
  MPC_HD void sweep_pair(int kt, int buf) {
    constexpr int AN = (A + 2) % TS;                 // next pair's position inside its tile
    const int ktn = (A + 2 < TS) ? kt : kt + 1;      // tile row (= column) of the next pair
    const bool pub = ktn < G;
    ex.par([&](Th &t) {
      if (t.mact) {
        double u[TS], v[TS], x[TS], y[TS];
        const double *bi = s.piv(buf);
        const double b00 = bi[0], b01 = bi[1], b11 = bi[2];
        const double *r0 = s.prow(buf, 0), *r1 = s.prow(buf, 1);
#pragma unroll
        for (int a = 0; a < TS; ++a) {
          u[a] = r0[TS * t.ti + a];
          v[a] = r1[TS * t.ti + a];
          const double uj = r0[TS * t.tj + a], vj = r1[TS * t.tj + a];
          x[a] = b00 * uj + b01 * vj;
          y[a] = b01 * uj + b11 * vj;
        }
        auto upd = [&](int a, int b) {   // two fused multiply-adds
          double m = t.Mx[a * TS + b];
          m -= u[a] * x[b];
          m -= v[a] * y[b];
          t.Mx[a * TS + b] = m;
        };
        // the cross through the next pair first: it is what the next phase waits for
#pragma unroll
        for (int a = AN; a < AN + 2; ++a)
#pragma unroll
          for (int b = 0; b < TS; ++b) upd(a, b);
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
          for (int b = AN; b < AN + 2; ++b)
            if (a != AN && a != AN + 1) upd(a, b);
        if (pub) publish<AN>(t, buf ^ 1, ktn);
        MPC_SCHED_FENCE();
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
          for (int b = 0; b < TS; ++b)
            if (a != AN && a != AN + 1 && b != AN && b != AN + 1) upd(a, b);
      }
    });
  }
  // Rows k = 6 kt + A and k + 1 of the matrix -> prow[b][0 / 1]: the tiles of tile row kt hold their part left of (and on) the
  // diagonal as their rows A, A + 1, the tiles of tile column kt hold the rest as their columns A, A + 1.  The 2 x 2 pivot block
  // goes out minus the identity, and piv[b] = its inverse.
  template <int A>
  MPC_HD void publish(const Th &t, int b, int kt) {
    if (t.ti == kt) {
      double *p0 = s.prow(b, 0) + TS * t.tj, *p1 = s.prow(b, 1) + TS * t.tj;
      const double one = t.dia ? 1.0 : 0.0;
#pragma unroll
      for (int bb = 0; bb < TS; ++bb) {
        p0[bb] = bb == A ? t.Mx[A * TS + bb] - one : t.Mx[A * TS + bb];
        p1[bb] = bb == A + 1 ? t.Mx[(A + 1) * TS + bb] - one : t.Mx[(A + 1) * TS + bb];
      }
      if (t.dia) {
        const double pp = t.Mx[A * TS + A], qq = t.Mx[(A + 1) * TS + A], ss = t.Mx[(A + 1) * TS + A + 1];
        const double det = pp * ss - qq * qq, rd = fast_recip(det);
        double *bo = s.piv(b);
        bo[0] = ss * rd;
        bo[1] = -(qq * rd);
        bo[2] = pp * rd;
        if (!(pp > 0) || !(det > 0)) s.bad = 1;      // not positive definite
      }
    } else if (t.tj == kt) {
      double *p0 = s.prow(b, 0) + TS * t.ti, *p1 = s.prow(b, 1) + TS * t.ti;
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        MPC_LDS_STORE64(p0 + a, t.Mx[a * TS + A]);
        MPC_LDS_STORE64(p1 + a, t.Mx[a * TS + A + 1]);
      }
    }
  }

  // ---- the one-pivot form of the same sweep (the multi-wave kernels of the long horizons: at their 256-register cap the pair form's
  // two extra 6-vectors go to scratch inside the loop -- measured h = 16 2.35 -> 2.72 ms, h = 20 2.83 -> 3.11 ms per 4096 robots).
  // Per step k:  p = a_kk;  a_ij -= a_ik a_kj / p;  a_ik -> a_ik / p;  a_kk -> -1/p (+2, as above).  The published row carries
  // (p - 1) in slot k; piv = {p, 1 / p}.
  MPC_HD void sweep_all_single() {
    int buf = 0;
    ex.par([&](Th &t) { if (t.mact) publish1<0>(t, 0, 0); });
    for (int kt = 0; kt < G; ++kt) sweep_steps1<0>(kt, buf);
  }
  template <int A>
  MPC_HD void sweep_steps1(int kt, int &buf) {
    if constexpr (A < TS) {
      sweep_step1<A>(kt, buf);
      buf ^= 1;
      sweep_steps1<A + 1>(kt, buf);
    }
  }
  template <int A>
  MPC_HD void sweep_step1(int kt, int buf) {
    constexpr int AN = (A + 1) % TS;                 // next pivot's position inside its tile
    const int ktn = (A + 1 < TS) ? kt : kt + 1;      // tile row (= column) of the next pivot
    const bool pub = ktn < G;
    ex.par([&](Th &t) {
      if (t.mact) {
        double g[TS], pc[TS];
        const double p = s.piv(buf)[0], pinv = s.piv(buf)[1];
        const double *pr = s.prow(buf, 0);
#pragma unroll
        for (int a = 0; a < TS; ++a) { g[a] = pr[TS * t.ti + a] * pinv; pc[a] = pr[TS * t.tj + a]; }
#pragma unroll
        for (int b = 0; b < TS; ++b) t.Mx[AN * TS + b] -= g[AN] * pc[b];
#pragma unroll
        for (int a = 0; a < TS; ++a)
          if (a != AN) t.Mx[a * TS + AN] -= g[a] * pc[AN];
        if (t.tid == 0 && !(p > 0)) s.bad = 1;     // not positive definite
        if (pub) publish1<AN>(t, buf ^ 1, ktn);
        MPC_SCHED_FENCE();
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
          for (int b = 0; b < TS; ++b)
            if (a != AN && b != AN) t.Mx[a * TS + b] -= g[a] * pc[b];
      }
    });
  }
  // Row k = 6 kt + A of the matrix -> prow[b]: the tiles of tile row kt hold its part left of (and on) the
  // diagonal as their row A, the tiles of tile column kt hold the rest as their column A.  Slot k itself
  // gets (pivot - 1), and piv[b] = {pivot, 1 / pivot}.
  template <int A>
  MPC_HD void publish1(const Th &t, int b, int kt) {
    if (t.ti == kt) {
      double *pn = s.prow(b, 0) + TS * t.tj;
#pragma unroll
      for (int bb = 0; bb < TS; ++bb)
        if (bb != A) pn[bb] = t.Mx[A * TS + bb];
      const double pivot = t.Mx[A * TS + A];
      pn[A] = t.dia ? pivot - 1.0 : pivot;
      if (t.dia) {
        s.piv(b)[0] = pivot;
        s.piv(b)[1] = fast_recip(pivot);
      }
    } else if (t.tj == kt) {
      double *pn = s.prow(b, 0) + TS * t.ti;
#pragma unroll
      for (int a = 0; a < TS; ++a) MPC_LDS_STORE64(pn + a, t.Mx[a * TS + A]);
    }
  }

  // ================================ 3. ADMM (auxil.c:164-228) ====================================================
  // b = sigma x - q + A^T (R z - y);  v = S^-1 b;  gp = G v
  MPC_HD void foot_rhs(Th &t) {
    double tm[5], acc[3], v[3], a[9];
    foot_a(t, a);
#pragma unroll
    for (int r = 0; r < 5; ++r) tm[r] = rho_at(t, r) * t.z[r] - t.y[r];
    at_mul(a, tm, acc);
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = kSigma * t.x[c] - t.q[c] + acc[c];
    sym3_mul(t.Si, v, t.b);       // t.b holds S^-1 b
    put_g(t, v);                  // (G S^-1) b
  }
  MPC_HD void admm_prepare() {
    ex.seq([&](Th &t) { if (t.foot) foot_rhs(t); });
    send<kHeld>();
  }
  // One ADMM iteration = two phases: the tile product, and the foot phase, which finishes the KKT solve
  //   x~ = S^-1 (b - G^T y_w),  z~ = A x~,  updates x, z, y (relaxation 1.6), prepares the next
  // right-hand side and hands its wrench contribution to the quad.
  // Inside a block of iterations t.y holds yh = y / rho (per row): OSQP's  z = clip(z_r + y / rho),  y <- y + rho (z_r - z)  becomes
  // z = clip(z_r + yh),  yh <- (z_r + yh) - z,  and  rho z - y = rho (z - yh)  takes its rho from the pre-multiplied cone block (fr).
  MPC_HD void y_scaled(bool to) {
    ex.seq([&](Th &t) {
      if (t.foot) {
#pragma unroll
        for (int r = 0; r < 5; ++r) t.y[r] *= to ? rinv_at(t, r) : rho_at(t, r);
      }
    });
  }
  static constexpr bool kBatchLoads = MPC_ADMM_BATCH_LOADS(T);
  // LAST: the iteration a termination check follows.  OSQP keeps delta_x = x - x_prev and delta_y = rho (alpha z~ + (1 - alpha) z_prev - z)
  // of every iteration (auxil.c:187-228); only the infeasibility certificates of the next check read them (auxil.c:364-515), so
  // only this variant forms them, and hands them over through LDS (the check is far away in registers).
  template <bool LAST = false>
  MPC_HD void admm_iter() {
    tile_product<kHeld>();
    recv<kHeld>();
    ex.seq([&](Th &t) {
      if (t.foot) {
        // every LDS constant of the phase is requested up front, as one batch behind a single wait (a wave that runs alone on its
        // SIMD has nobody to hide five separate round trips behind; the registers are there: 512 per lane)
        double gf[18], a[9], ar[9], lo[5], up[5];
        load_g(t, gf);
        foot_a(t, a);
        if constexpr (kBatchLoads) {
          foot_bounds(t, lo, up);
          foot_ar(t, ar);
        }
        double tt[3], zt[5], dd[5], acc[3], v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double wy = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r) wy += gf[3 * r + c] * t.w6[r];
          t.xt[c] = t.b[c] - wy;      // S^-1 b - (G S^-1)^T y_w
        }
        a_mul(a, t.xt, zt);
        if constexpr (!kBatchLoads) foot_bounds(t, lo, up);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const double zr = kAlphaRelax * zt[r] + (1.0 - kAlphaRelax) * t.z[r];
          const double tq = zr + t.y[r];
          const double zn = clampd(tq, lo[r], up[r]);
          const double yn = tq - zn;
          if constexpr (LAST) s.dxy[pidx(3 + r, t.fid)] = rho_at(t, r) * (zr - zn);
          t.z[r] = zn;
          t.y[r] = yn;
          dd[r] = zn - yn;
        }
        if constexpr (!kBatchLoads) foot_ar(t, ar);
        at_mul(ar, dd, acc);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double xn = kAlphaRelax * t.xt[c] + (1.0 - kAlphaRelax) * t.x[c];
          if constexpr (LAST) s.dxy[pidx(c, t.fid)] = xn - t.x[c];
          t.x[c] = xn;
          v[c] = kSigma * xn - t.q[c] + acc[c];
        }
        sym3_mul(t.Si, v, t.b);
#pragma unroll
        for (int r = 0; r < 6; ++r) t.w6[r] = gf[3 * r] * v[0] + gf[3 * r + 1] * v[1] + gf[3 * r + 2] * v[2];
      }
    });
    send<kHeld>();
  }

  // P_s v for a per-foot vector given by sel(t): out = c alpha D^2 v + W^T (c Theta (W v)).  `in` and `out` are members of Th.
  template <class In, class Out>
  MPC_HD void mul_P(In &&in, Out &&out) {
    ex.seq([&](Th &t) { if (t.foot) put_wrench(t, in(t)); });
    product<kTheta>();
    ex.seq([&](Th &t) {
      if (t.foot) {
        double wy[3];
        get_wrench(t, wy);
        const double *v = in(t);
        double *o = out(t);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (s.calpha * Dat(t, c) * Dat(t, c)) * v[c] + wy[c];
      }
    });
  }

  // 1 / E, 1 / D of the unscaled residual norms: they feed tolerance tests and the rho estimate, nothing that must be rounded like a
  // division -- v_rcp_f64 + two Newton steps (to the last ulp or two) where the IEEE sequence costs a dozen dependent instructions.
  // (Single-wave kernels only: at the 256-register cap of the long horizons the shorter sequence moves the allocator's spills into the
  // sweep loops -- h = 16: 2.24 -> 2.47 ms.)
  static MPC_HD double norm_recip(double d) {
    if constexpr (T <= 64) return fast_recip(d);
    else return 1.0 / d;
  }
  // residuals of (x, z, y) (auxil.c:243-306, 563-629) + the norms termination and rho need.
  // red[] slots: 0 pri_res 1 ||Einv z|| 2 ||Einv Ax|| 3 ||rp|| 4 ||z|| 5 ||Ax||
  //              6 ||Dinv rd|| 7 ||Dinv q|| 8 ||Dinv Aty|| 9 ||Dinv Px|| 10 ||rd|| 11 ||q|| 12 ||Aty|| 13 ||Px||
  // INF (the checks of the ADMM loop): the cheap parts of OSQP's infeasibility certificates on the last iteration's delta_x, delta_y
  // (Shared::dxy; is_primal_infeasible / is_dual_infeasible, auxil.c:364-515):
  //              14 ||E dy||   15 SUM u max(dy, 0) + l min(dy, 0)   16 ||Dinv A^T dy||        (dy projected on the recession cone's polar)
  //              17 ||D dx||   18 SUM q dx   19 max(0, Einv A dx) over rows with finite u   20 max(0, -Einv A dx) over rows with finite l
  // Every foot lane forms the maxima / sums over its five rows and three variables; NRED lanes finish.
  template <bool INF = false, class X, class Z, class Y, class PX>
  MPC_HD void residuals(X &&xs, Z &&zs, Y &&ys, PX &&pxs) {
    constexpr int RW = Sh::RW, NR = INF ? Sh::NRED : 14;
    ex.par([&](Th &t) {
      if (t.foot) {
        const double *x = xs(t), *z = zs(t), *y = ys(t), *px = pxs(t);
        double mx[NR], ax[5], aty[3], a[9];
        foot_a(t, a);
#pragma unroll
        for (int k = 0; k < NR; ++k) mx[k] = 0;
        a_mul(a, x, ax);
        at_mul(a, y, aty);
        double ev[5], ei[5], di[3];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const double rr = ax[r] - z[r];
          ev[r] = sc[C::SC_E + 5 * t.fid + r];
          ei[r] = norm_recip(ev[r]);
          mx[0] = dmax(mx[0], fabs(ei[r] * rr)); mx[1] = dmax(mx[1], fabs(ei[r] * z[r])); mx[2] = dmax(mx[2], fabs(ei[r] * ax[r]));
          mx[3] = dmax(mx[3], fabs(rr)); mx[4] = dmax(mx[4], fabs(z[r])); mx[5] = dmax(mx[5], fabs(ax[r]));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double qv = t.q[c], rr = qv + px[c] + aty[c];
          di[c] = norm_recip(Dat(t, c));
          mx[6] = dmax(mx[6], fabs(di[c] * rr)); mx[7] = dmax(mx[7], fabs(di[c] * qv)); mx[8] = dmax(mx[8], fabs(di[c] * aty[c]));
          mx[9] = dmax(mx[9], fabs(di[c] * px[c])); mx[10] = dmax(mx[10], fabs(rr)); mx[11] = dmax(mx[11], fabs(qv));
          mx[12] = dmax(mx[12], fabs(aty[c])); mx[13] = dmax(mx[13], fabs(px[c]));
        }
        if constexpr (INF) {
          double lo[5], up[5], dx[3], dy[5], adx[5], atdy[3];
          foot_bounds(t, lo, up);
#pragma unroll
          for (int c = 0; c < 3; ++c) dx[c] = s.dxy[pidx(c, t.fid)];
#pragma unroll
          for (int r = 0; r < 5; ++r) {
            double v = s.dxy[pidx(3 + r, t.fid)];
            const bool uinf = up[r] > kInfty * kMinScaling, linf = lo[r] < -kInfty * kMinScaling;   // (auxil.c:377-391)
            v = uinf ? (linf ? 0.0 : dmin(v, 0.0)) : (linf ? dmax(v, 0.0) : v);
            dy[r] = v;
            mx[14] = dmax(mx[14], fabs(ev[r] * v));                                                // E dy
            mx[15] += up[r] * dmax(v, 0.0) + lo[r] * dmin(v, 0.0);
          }
          at_mul(a, dy, atdy);
          a_mul(a, dx, adx);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            mx[16] = dmax(mx[16], fabs(di[c] * atdy[c]));
            mx[17] = dmax(mx[17], fabs(Dat(t, c) * dx[c]));                                         // D dx
            mx[18] += t.q[c] * dx[c];
          }
#pragma unroll
          for (int r = 0; r < 5; ++r) {
            const double v = ei[r] * adx[r];
            if (up[r] < kInfty * kMinScaling) mx[19] = dmax(mx[19], v);
            if (lo[r] > -kInfty * kMinScaling) mx[20] = dmax(mx[20], -v);
          }
        }
#pragma unroll
        for (int k = 0; k < NR; ++k) s.part[k * RW + t.fid] = mx[k];
      }
    });
    ex.par([&](Th &t) {
      if (t.tid < NR) {
        // (both folds for every lane, one select at the end: a per-lane choice inside the loop makes the compiler branch around every load)
        const double *p = s.part + t.tid * RW;
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        static_assert(NF % 4 == 0, "four feet per step");
        // (unrolled in the multi-wave kernels it costs them registers they do not have: tools/isa_census.py)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll T <= 64 ? NF / 4 : 1
#endif
        for (int k = 0; k < NF; k += 4) {
          const double v0 = p[k], v1 = p[k + 1], v2 = p[k + 2], v3 = p[k + 3];
          m0 = dmax(m0, v0); m1 = dmax(m1, v1); m2 = dmax(m2, v2); m3 = dmax(m3, v3);
          if constexpr (INF) { a0 += v0; a1 += v1; a2 += v2; a3 += v3; }
        }
        const double mv = dmax(dmax(m0, m1), dmax(m2, m3)), av = (a0 + a1) + (a2 + a3);
        s.red[t.tid] = dbits(INF && (t.tid == 15 || t.tid == 18) ? av : mv);
      }
    });
  }

  // check_termination (auxil.c:684-793) + adapt_rho decision (auxil.c:13-77).  One thread decides.  APPROX: the second look
  // OSQP takes when max_iter is reached, every tolerance times ten (osqp.c:563-568) -> the *_INACCURATE statuses.
  // The dual certificate's expensive part (P dx, the recession directions) is evaluated only when its cheap part holds: dual_cand.
  static constexpr double kEpsPrimInf = 1e-4, kEpsDualInf = 1e-4;   // constants.h:64-65 (the reference keeps the defaults)
  template <bool APPROX = false>
  MPC_HD void check_and_adapt(int iter) {
    ex.par([&](Th &t) {
      if (t.tid == 0) {
        const double pri = bitsd(s.red[0]), dua = s.cinv * bitsd(s.red[6]);
        const double tf = APPROX ? 10.0 : 1.0;
        const double ea = tf * eps_abs, er = tf * eps_rel, epi = tf * kEpsPrimInf, edi = tf * kEpsDualInf;
        s.pri_res = pri; s.dua_res = dua; s.iter = iter; s.rho_new = 0; s.dual_cand = 0;
        if (!(pri <= kInfty) || !(dua <= kInfty)) { s.status = kStNonCvx; s.done = 1; }
        else {
          const double eps_prim = ea + er * dmax(bitsd(s.red[1]), bitsd(s.red[2]));
          const double eps_dual = ea + er * s.cinv * dmax(dmax(bitsd(s.red[7]), bitsd(s.red[8])), bitsd(s.red[9]));
          s.sig_changed = 0;
          s.loose_ok = pri < kEpsAbs + (eps_prim - ea) * (kEpsRel / er) && dua < kEpsAbs + (eps_dual - ea) * (kEpsRel / er);   // the 1e-3 test
          const bool prim_ok = pri < eps_prim, dual_ok = dua < eps_dual;
          if (prim_ok && dual_ok) { s.status = APPROX ? kStSolvedInaccurate : kStSolved; s.done = 1; }
          else {
            bool prim_inf = false;
            if (!prim_ok) {                       // is_primal_infeasible (auxil.c:364-424)
              const double ndy = bitsd(s.red[14]);
              if (ndy > epi && bitsd(s.red[15]) < -epi * ndy) prim_inf = bitsd(s.red[16]) < epi * ndy;
            }
            if (prim_inf) { s.status = APPROX ? kStPrimInfInaccurate : kStPrimInf; s.done = 1; }
            else {
              if (!dual_ok) {                     // is_dual_infeasible, first two tests (auxil.c:426-460)
                const double ndx = bitsd(s.red[17]);
                if (ndx > edi && bitsd(s.red[18]) < -s.c * edi * ndx) s.dual_cand = 1;
              }
              if (!APPROX) {
                double pr = bitsd(s.red[3]) / (dmax(bitsd(s.red[4]), bitsd(s.red[5])) + 1e-10);
                double dr = bitsd(s.red[10]) / (dmax(dmax(bitsd(s.red[11]), bitsd(s.red[12])), bitsd(s.red[13])) + 1e-10);
                double rn = s.rho * sqrt(pr / (dr + 1e-10));
                rn = clampd(rn, kRhoMin, kRhoMax);
                if ((rn > s.rho * kAdaptTol || rn < s.rho / kAdaptTol) && s.rho_updates < max_rho_updates) s.rho_new = rn;
              }
            }
          }
        }
      }
    });
#ifdef MPC_EMU_DEBUG
    ++g_checks; if (s.dual_cand) ++g_dual_cands;
#endif
    if (s.dual_cand) dual_certificate<APPROX>();
  }
  // the rest of is_dual_infeasible (auxil.c:461-505): ||Dinv P dx|| < c eps ||D dx||, and A dx inside the recession cone of [l, u]
  template <bool APPROX>
  MPC_HD void dual_certificate() {
    ex.seq([&](Th &t) {
      if (t.foot) {
#pragma unroll
        for (int c = 0; c < 3; ++c) t.xt[c] = s.dxy[pidx(c, t.fid)];   // (x~ is dead between a check and the next iteration)
      }
    });
    mul_P([](Th &t) { return t.xt; }, [](Th &t) { return t.px; });
    constexpr int RW = Sh::RW;
    ex.par([&](Th &t) {
      if (t.foot) {
        double m = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) m = dmax(m, fabs(t.px[c] * norm_recip(Dat(t, c))));
        s.part[t.fid] = m;
      }
    });
    ex.par([&](Th &t) {
      if (t.tid == 0) {
        double m = 0;
        for (int k = 0; k < NF; ++k) m = dmax(m, s.part[k]);
        const double edi = (APPROX ? 10.0 : 1.0) * kEpsDualInf, ndx = bitsd(s.red[17]);
        if (m < s.c * edi * ndx && !(bitsd(s.red[19]) > edi * ndx) && !(bitsd(s.red[20]) > edi * ndx)) {
          s.status = APPROX ? kStDualInfInaccurate : kStDualInf; s.done = 1; s.rho_new = 0;
        }
      }
    });
    (void)RW;
  }

  // ================================ 4. polish (polish.c) =========================================================
  // Active set from (z, y) (polish.c:36-52); the delta-regularised KKT solve with three refinement steps (polish.c:102-160)
  // is carried out in the null space of the active rows, foot by foot, in FORCE coordinates: with N_f the orthonormal null
  // basis of a foot's active rows, Hd = N^T (P_s + delta I) N and Omega = N Hd^-1 N^T,
  //     w_{k+1} = w_k + Omega r_k,   r_{k+1} = delta Omega r_k        (H Hd^-1 = I - delta Hd^-1: no further products with P),
  //     Omega r = Xi r - Xi G^T (I - M^-1) (G Xi r),  Xi = N (delta I + c alpha N^T D^2 N)^-1 N^T  (3 x 3 per foot),
  // with G, M of factor_core(Xi).
  MPC_HD void omega_apply() {   // in: t.w6 = V_f^T u from the foot lanes, pt = u = C^T r;  out: pw = Omega r = C (u - V_f (I - M^-1) V^T u)
    product<kHeld>();
    ex.seq([&](Th &t) {
      if (t.foot) {
        double vy[3], d[3];
        get_g(t, vy);
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = t.pt[k] - vy[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) t.pw[c] = t.pC[3 * c] * d[0] + t.pC[3 * c + 1] * d[1] + t.pC[3 * c + 2] * d[2];
      }
    });
  }
  static MPC_HD void ct_mul(const double *C, const double *r, double *u) {   // u = C^T r
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] = C[k] * r[0] + C[3 + k] * r[1] + C[6 + k] * r[2];
  }
  // Polish factorisation.  H^-1 restricted to the null space of the active rows is  C (I - V V^T + V M^-1 V^T) C^T  with
  // Xi = C C^T per foot and V an ORTHONORMAL basis of the range of F = C^T W^T (per step: the 12 x 6 stack of its feet).
  // |Xi| ~ 1 / delta multiplies whatever I - V V^T fails to annihilate, so the range must be accurate to ~1e-11: Gram-matrix
  // orthogonalisation (the L D L^T of Z = F^T F that the ADMM system uses) loses eps cond(Z) and is not enough; the columns
  // are orthogonalised directly -- classical Gram-Schmidt, twice per column, the inner products summed over the quad -- which
  // loses eps cond(F) only.  A column whose remainder is below 1e-12 of its length lies in the span of the earlier ones
  // (two point feet give no torque about the line through them) and is dropped.  F = V R, so Z = R^T R: L = R^T.
  template <int J>
  MPC_HD void orth_col() {
    if constexpr (J < 6) {
      ex.seq([&](Th &t) {
        if (t.foot) {
#pragma unroll
          for (int k = 0; k < 3; ++k) t.pv[k] = t.zq[3 * J + k];
#pragma unroll
          for (int i = 0; i < J; ++i) t.pR[pk(J, i)] = 0.0;
        }
      });
      for (int rep = 0; rep < 2; ++rep) {
        ex.seq([&](Th &t) {
#pragma unroll
          for (int i = 0; i < 6; ++i) t.w6[i] = (i < J && t.foot) ? t.pQ[3 * i] * t.pv[0] + t.pQ[3 * i + 1] * t.pv[1] + t.pQ[3 * i + 2] * t.pv[2] : 0.0;
        });
        ex.template quad_allsum<6>([](Th &t) { return t.w6; });
        ex.seq([&](Th &t) {
          if (t.foot) {
#pragma unroll
            for (int i = 0; i < J; ++i) {
#pragma unroll
              for (int k = 0; k < 3; ++k) t.pv[k] -= t.w6[i] * t.pQ[3 * i + k];
              t.pR[pk(J, i)] += t.w6[i];
            }
          }
        });
      }
      ex.seq([&](Th &t) {
#pragma unroll
        for (int i = 0; i < 6; ++i) t.w6[i] = 0.0;
        if (t.foot) t.w6[0] = t.pv[0] * t.pv[0] + t.pv[1] * t.pv[1] + t.pv[2] * t.pv[2];
      });
      ex.template quad_allsum<6>([](Th &t) { return t.w6; });
      ex.seq([&](Th &t) {
        if (t.foot) {
          const double n2 = t.w6[0];
          const bool keep = n2 > 1e-24 * t.pn[J];
          const double inv = keep ? fast_rsqrt(keep ? n2 : 1.0) : 0.0;
          t.pR[pk(J, J)] = keep ? n2 * inv : 0.0;
#pragma unroll
          for (int k = 0; k < 3; ++k) t.pQ[3 * J + k] = t.pv[k] * inv;
        }
      });
      orth_col<J + 1>();
    }
  }
  MPC_HD void polish_factor() {
    ex.seq([&](Th &t) {
#pragma unroll
      for (int r = 0; r < 6; ++r) t.w6[r] = 0.0;
      if (t.foot) {   // F_f[k][r] = sum_c C[c][k] W[r][c]  ->  zq[3 r + k]; squared column lengths
        double w[18];
        foot_w(t, w);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          double n2 = 0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double v = t.pC[k] * w[3 * r] + t.pC[3 + k] * w[3 * r + 1] + t.pC[6 + k] * w[3 * r + 2];
            t.zq[3 * r + k] = v;
            n2 += v * v;
          }
          t.w6[r] = n2;
        }
      }
    });
    ex.template quad_allsum<6>([](Th &t) { return t.w6; });
    ex.seq([&](Th &t) {
#pragma unroll
      for (int r = 0; r < 6; ++r) t.pn[r] = t.w6[r];
    });
    orth_col<0>();
    ex.par([&](Th &t) {
      if (t.foot) {
#pragma unroll
        for (int k = 0; k < 18; ++k) s.Gf[pidx(k, t.fid)] = t.pQ[k];          // "G_f" = V_f^T (6 x 3)
        if ((t.fid & 3) == 0) {   // L = R^T (pR[pk(j, i)] = R[i][j], i <= j)
          double *ol = s.Lk + 36 * (t.fid >> 2);
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) ol[6 * i + j] = i >= j ? t.pR[pk(i, j)] : 0.0;
        }
      }
    });
    factor_tail();
  }

  // ROUNDS (the exact mode's calls): extra refinement rounds for a near miss of the optimality test, see below
  template <bool ROUNDS = false>
  MPC_HD void polish() {
    ex.par([&](Th &t) {
      if (t.foot) {
        double a[9], lo[5], up[5];
        foot_a(t, a);
        foot_bounds(t, lo, up);
        // rows of the scaled cone block (static indexing)
        const double A[15] = {a[0], 0, a[1], a[2], 0, a[3], 0, a[4], a[5], 0, a[6], a[7], 0, 0, a[8]};
#pragma unroll
        for (int r = 0; r < 5; ++r) if (!act_given) t.act[r] = (t.z[r] - lo[r] < -t.y[r]) ? -1 : ((up[r] - t.z[r] < t.y[r]) ? 1 : 0);
        // orthonormal basis Q of the active rows (rank r), null basis Nn (rows, 3 - r of them); rows of Q beyond the
        // current rank are zero, so projecting on all three rows equals projecting on the first r of them
        double Q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Nn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int r = 0;
#pragma unroll
        for (int row = 0; row < 5; ++row) {
          const bool cand = r < 3 && t.act[row];
          double v[3] = {A[3 * row], A[3 * row + 1], A[3 * row + 2]};
          const double n02 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
#pragma unroll
          for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const double d = v[0] * Q[3 * k] + v[1] * Q[3 * k + 1] + v[2] * Q[3 * k + 2];
              v[0] -= d * Q[3 * k]; v[1] -= d * Q[3 * k + 1]; v[2] -= d * Q[3 * k + 2];
            }
          // (one reciprocal square root per row -- the test is on the squares -- where two square roots and three divisions stood:
          // a foot lane's set-up is one long dependent chain, and nothing else of the wave runs beside it)
          const double nr2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
          const bool take = cand && nr2 > 1e-12 * n02;
          const double ninv = fast_rsqrt(take ? nr2 : 1.0);
          const double q0 = v[0] * ninv, q1 = v[1] * ninv, q2 = v[2] * ninv;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const bool here = take && r == k;
            Q[3 * k] = here ? q0 : Q[3 * k]; Q[3 * k + 1] = here ? q1 : Q[3 * k + 1]; Q[3 * k + 2] = here ? q2 : Q[3 * k + 2];
          }
          r += take ? 1 : 0;
        }
        if (r == 0) { Nn[0] = 1; Nn[4] = 1; Nn[8] = 1; }
        else if (r == 1) {
          const int imin = fabs(Q[0]) <= fabs(Q[1]) ? (fabs(Q[0]) <= fabs(Q[2]) ? 0 : 2) : (fabs(Q[1]) <= fabs(Q[2]) ? 1 : 2);
          const double e[3] = {imin == 0 ? 1.0 : 0.0, imin == 1 ? 1.0 : 0.0, imin == 2 ? 1.0 : 0.0};
          const double d = imin == 0 ? Q[0] : imin == 1 ? Q[1] : Q[2];
          double v[3] = {e[0] - d * Q[0], e[1] - d * Q[1], e[2] - d * Q[2]};
          const double ninv = fast_rsqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);      // (>= 2 / 3: e is the axis Q leans on least)
          Nn[0] = v[0] * ninv; Nn[1] = v[1] * ninv; Nn[2] = v[2] * ninv;
          Nn[3] = Q[1] * Nn[2] - Q[2] * Nn[1]; Nn[4] = Q[2] * Nn[0] - Q[0] * Nn[2]; Nn[5] = Q[0] * Nn[1] - Q[1] * Nn[0];
        } else if (r == 2) {
          Nn[0] = Q[1] * Q[5] - Q[2] * Q[4]; Nn[1] = Q[2] * Q[3] - Q[0] * Q[5]; Nn[2] = Q[0] * Q[4] - Q[1] * Q[3];
          const double ninv = fast_rsqrt(Nn[0] * Nn[0] + Nn[1] * Nn[1] + Nn[2] * Nn[2]);    // (= 1 up to rounding: the cross product of two orthonormal rows)
          Nn[0] *= ninv; Nn[1] *= ninv; Nn[2] *= ninv;
        }
        const int nn = 3 - r;
        // Gamma = Q (Q^T B Q)^{-1} Q^T with B = A_act^T A_act
        double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Gq[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Gi[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
        for (int row = 0; row < 5; ++row)
          if (t.act[row])
#pragma unroll
            for (int c1 = 0; c1 < 3; ++c1)
#pragma unroll
              for (int c2 = 0; c2 < 3; ++c2) B[3 * c1 + c2] += A[3 * row + c1] * A[3 * row + c2];
#pragma unroll
        for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
          for (int k2 = 0; k2 < 3; ++k2) {
            if (k1 >= r || k2 >= r) continue;
            double tt = 0;
#pragma unroll
            for (int c1 = 0; c1 < 3; ++c1)
#pragma unroll
              for (int c2 = 0; c2 < 3; ++c2) tt += Q[3 * k1 + c1] * B[3 * c1 + c2] * Q[3 * k2 + c2];
            Gq[3 * k1 + k2] = tt;
          }
        // Gauss-Jordan on the (identity padded) 3x3
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const double d = fast_recip(Gq[3 * p + p]);
#pragma unroll
          for (int j = 0; j < 3; ++j) { Gq[3 * p + j] *= d; Gi[3 * p + j] *= d; }
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            if (i == p) continue;
            const double fc = Gq[3 * i + p];
#pragma unroll
            for (int j = 0; j < 3; ++j) { Gq[3 * i + j] -= fc * Gq[3 * p + j]; Gi[3 * i + j] -= fc * Gi[3 * p + j]; }
          }
        }
#pragma unroll
        for (int c1 = 0; c1 < 3; ++c1)
#pragma unroll
          for (int c2 = 0; c2 < 3; ++c2) {
            double tt = 0;
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
              for (int k2 = 0; k2 < 3; ++k2) {
                if (k1 >= r || k2 >= r) continue;
                tt += Q[3 * k1 + c1] * Gi[3 * k1 + k2] * Q[3 * k2 + c2];
              }
            t.pG[3 * c1 + c2] = tt;
          }
        // u0 = Gamma A_act^T b_act  (the point satisfying the active rows)
        double vb[3] = {0, 0, 0};
#pragma unroll
        for (int row = 0; row < 5; ++row)
          if (t.act[row]) {
            const double bd = t.act[row] < 0 ? lo[row] : up[row];
#pragma unroll
            for (int c = 0; c < 3; ++c) vb[c] += A[3 * row + c] * bd;
          }
#pragma unroll
        for (int c = 0; c < 3; ++c) t.pu0[c] = t.pG[3 * c] * vb[0] + t.pG[3 * c + 1] * vb[1] + t.pG[3 * c + 2] * vb[2];
        // Xi = N (delta I + c alpha N^T D^2 N)^-1 N^T  on the nn null coordinates (identity padding keeps the inverse well defined)
        double Sp[6];
        {
          double nd[9];
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) nd[3 * k + c] = Nn[3 * k + c] * Dat(t, c);
          const double ca = s.calpha;
          Sp[0] = 0 < nn ? ca * (nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]) + kDelta : 1.0;
          Sp[3] = 1 < nn ? ca * (nd[3] * nd[3] + nd[4] * nd[4] + nd[5] * nd[5]) + kDelta : 1.0;
          Sp[5] = 2 < nn ? ca * (nd[6] * nd[6] + nd[7] * nd[7] + nd[8] * nd[8]) + kDelta : 1.0;
          Sp[1] = 1 < nn ? ca * (nd[0] * nd[3] + nd[1] * nd[4] + nd[2] * nd[5]) : 0.0;
          Sp[2] = 2 < nn ? ca * (nd[0] * nd[6] + nd[1] * nd[7] + nd[2] * nd[8]) : 0.0;
          Sp[4] = 2 < nn ? ca * (nd[3] * nd[6] + nd[4] * nd[7] + nd[5] * nd[8]) : 0.0;
        }
        // Xi = C C^T with C = N^T Ls^-T (Sp = Ls Ls^T): the factor is what the orthogonalisation below works with
        {
          const double i00 = fast_rsqrt(Sp[0]), l00 = Sp[0] * i00;
          const double l10 = Sp[1] * i00, l20 = Sp[2] * i00;
          const double d1 = Sp[3] - l10 * l10;
          const double i11 = fast_rsqrt(d1), l11 = d1 * i11;
          const double l21 = (Sp[4] - l20 * l10) * i11;
          const double d2 = Sp[5] - l20 * l20 - l21 * l21;
          const double i22 = fast_rsqrt(d2);
          (void)l00; (void)l11;
          const double i10 = -l10 * i00 * i11, i21 = -l21 * i11 * i22, i20 = -(l20 * i00 + l21 * i10) * i22;
          const double U[9] = {i00, i10, i20, 0, i11, i21, 0, 0, i22};   // Ls^-T, upper triangular
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              double v = 0;
#pragma unroll
              for (int mm = 0; mm <= k; ++mm) v += Nn[3 * mm + c] * U[3 * mm + k];
              t.pC[3 * c + k] = v;
            }
        }
        put_wrench(t, t.pu0);
      }
    });
    product<kTheta>();
    ex.seq([&](Th &t) {
      if (t.foot) {   // P_s u0;  g = -q - P_s u0;  t = Xi g
        double wy[3];
        get_wrench(t, wy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          t.pPu[c] = (s.calpha * Dat(t, c) * Dat(t, c)) * t.pu0[c] + wy[c];
          t.pg[c] = -t.q[c] - t.pPu[c];
          t.pxN[c] = 0;
        }
        ct_mul(t.pC, t.pg, t.pt);                         // u = C^T g
      }
    });
    lap(11);
    // (Exact mode, after the active-set method: its iterate t.gx is the optimum of the working set already -- the method's H^-1 carries
    // no regularisation -- so the first round only checks it: xN = gx - u0, one Theta product, the finish and the optimality test; the
    // reduced system is factorised, and refined from that xN, only if the test fails.)
    // w_0 = Omega g, then kPolishRefine steps of iterative refinement against the UN-regularised system (polish.c:102-160):
    //   w <- w + Omega (g - P_s w).   With an exact Omega the residual obeys r_{k+1} = delta Omega r_k and needs no product with
    // P; Omega is only accurate to ~1e-10 |Xi|, so the true residual is formed (one Theta product per step) -- which is also
    // what OSQP does, and what keeps the refinement self-correcting.  The last product is the P_s xN the finish needs anyway.
    // (Exact mode: a polish that must pass the optimality test and misses the dual test by less than 1e4 x -- the right set, the
    // refinement not yet converged -- goes round again with MPC_EXACT_ROUND_STEPS more steps, at most MPC_EXACT_ROUNDS times: most sets
    // pass after the first eight steps, and the test costs less than two steps.)
    for (int round = 0;; ++round) {
    const bool direct = ROUNDS && xn_given && round == 0;
    const bool first = round == 0 || (ROUNDS && xn_given && round == 1);      // the first round that refines
    const int nsteps = first ? polish_refine : MPC_EXACT_ROUND_STEPS;
    if (first && !direct) {
      polish_factor();
      if (round == 0) ex.seq([&](Th &t) { if (t.foot) put_g(t, t.pt); });
      lap(12);
    }
    if (direct) {
      ex.seq([&](Th &t) {
        if (t.foot) {
#pragma unroll
          for (int c = 0; c < 3; ++c) t.pxN[c] = s.dxy[pidx(c, t.fid)] - t.pu0[c];      // (the method's iterate, parked by run_active_set)
          put_wrench(t, t.pxN);
        }
      });
      product<kTheta>();
    }
    if (ROUNDS && round > 0) {      // the residual of the current xN, as the loop's last iteration would have left it
      ex.seq([&](Th &t) { if (t.foot) put_wrench(t, t.pxN); });
      product<kTheta>();
      ex.seq([&](Th &t) {
        if (t.foot) {
          double wy[3];
          get_wrench(t, wy);
#pragma unroll
          for (int c = 0; c < 3; ++c) t.pr[c] = t.pg[c] - ((s.calpha * Dat(t, c) * Dat(t, c)) * t.pxN[c] + wy[c]);
          ct_mul(t.pC, t.pr, t.pt);
          put_g(t, t.pt);
        }
      });
    }
    for (int it = round == 0 ? 0 : 1; it <= nsteps && !direct; ++it) {
      omega_apply();
#ifdef MPC_EMU_DEBUG
      if (dbg && it == 0) ex.par([&](Th &t) {
        if (t.foot) {
          double *o = dbg + 38 * t.fid;
          for (int r = 0; r < 5; ++r) o[r] = t.act[r];
          for (int c1 = 0, e = 0; c1 < 3; ++c1) for (int c2 = c1; c2 < 3; ++c2, ++e) o[5 + e] = t.pC[3 * c1] * t.pC[3 * c2] + t.pC[3 * c1 + 1] * t.pC[3 * c2 + 1] + t.pC[3 * c1 + 2] * t.pC[3 * c2 + 2];
          for (int c = 0; c < 3; ++c) { o[11 + c] = t.pg[c]; o[14 + c] = t.pC[3 * c] * t.pt[0] + t.pC[3 * c + 1] * t.pt[1] + t.pC[3 * c + 2] * t.pt[2]; o[17 + c] = t.pw[c]; }
          for (int k = 0; k < 18; ++k) o[20 + k] = s.Gf[pidx(k, t.fid)];
        }
      });
#endif
      ex.seq([&](Th &t) {
        if (t.foot) {
#pragma unroll
          for (int c = 0; c < 3; ++c) t.pxN[c] += t.pw[c];
          put_wrench(t, t.pxN);
        }
      });
      product<kTheta>();                                  // c Theta W xN  (-> P_s xN)
      if (it < nsteps) {
        ex.seq([&](Th &t) {
          if (t.foot) {
            double wy[3];
            get_wrench(t, wy);
#pragma unroll
            for (int c = 0; c < 3; ++c) t.pr[c] = t.pg[c] - ((s.calpha * Dat(t, c) * Dat(t, c)) * t.pxN[c] + wy[c]);
            ct_mul(t.pC, t.pr, t.pt);
            put_g(t, t.pt);
          }
        });
      }
    }
    lap(13);
    // x = u + xN ; y = A Gamma (g - P xN) on active rows ; z = A x ; normal-cone projection (proj.c:17-31)
    ex.par([&](Th &t) {
      if (t.foot) {
        double wy[3], pxn[3], gg[3], rwv[3], ax[5], ay[5], a[9], lo[5], up[5];
        foot_a(t, a);
        foot_bounds(t, lo, up);
        get_wrench(t, wy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          pxn[c] = (s.calpha * Dat(t, c) * Dat(t, c)) * t.pxN[c] + wy[c];
          t.xp[c] = t.pu0[c] + t.pxN[c];
          gg[c] = t.pg[c] - pxn[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rwv[c] = t.pG[3 * c] * gg[0] + t.pG[3 * c + 1] * gg[1] + t.pG[3 * c + 2] * gg[2];
        a_mul(a, rwv, ay);
        a_mul(a, t.xp, ax);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const double yv = t.act[r] ? ay[r] : 0.0;
          const double tt = ax[r] + yv;
          const double zc = clampd(tt, lo[r], up[r]);
          t.zp[r] = zc;
          t.yp[r] = tt - zc;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // P_s x_pol (with ROUNDS in its own array: P_s u0 is needed again)
          if constexpr (ROUNDS) t.px[c] = t.pPu[c] + pxn[c];
          else t.pPu[c] += pxn[c];
        }
      }
    });
    // residuals at the polished point, acceptance (polish.c:306-345)
    const double pri0 = s.pri_res, dua0 = s.dua_res;
    residuals([](Th &t) { return t.xp; }, [](Th &t) { return t.zp; }, [](Th &t) { return t.yp; }, [](Th &t) { return ROUNDS ? t.px : t.pPu; });
    ex.par([&](Th &t) {
      if (t.tid == 0) {
        const double pri = bitsd(s.red[0]), dua = s.cinv * bitsd(s.red[6]);
        const bool ok = !s.bad && ((pri < pri0 && dua < dua0) || (pri < pri0 && dua0 < 1e-10) || (dua < dua0 && pri0 < 1e-10));
#ifdef MPC_EMU_DEBUG
        if (dbg) { dbg[38 * NF] = pri0; dbg[38 * NF + 1] = dua0; dbg[38 * NF + 2] = pri; dbg[38 * NF + 3] = dua; }
#endif
        // exact mode: is the polished point the optimum?  (the termination test of auxil.c:684-793 at eps_exact)
        const double ep = eps_exact + eps_exact * dmax(bitsd(s.red[1]), bitsd(s.red[2]));
        const double ed = eps_exact + eps_exact * s.cinv * dmax(dmax(bitsd(s.red[7]), bitsd(s.red[8])), bitsd(s.red[9]));
        bool verified = !s.bad && pri < ep && dua < ed;
        if constexpr (ROUNDS) {
          // The active-set method's own set, refined to where rounding stops it: a dual residual that a round of refinement no longer
          // lowers and that misses the test by less than 10 x is the floor of this problem's arithmetic (multipliers of 1e5 against
          // forces of 1e2), not a wrong set -- those miss by orders of magnitude.
          if (act_given && !verified && !s.bad && round > 0 && pri < ep && dua < 10.0 * ed && dua > 0.5 * s.dua_last) verified = true;
          // The method's iterate itself (the direct check): what it misses the dual test by is the rounding of ~30 rank-one steps, 1.0 to
          // 1.6 x the tolerance on one robot in thirty (a round of refinement takes it down 100-1000 x: the set is right); 2 x is accepted.
          if (direct && !verified && !s.bad && pri < ep && dua < 2.0 * ed) verified = true;
          s.dua_last = dua;
        }
#ifdef MPC_EMU_DEBUG
        if (direct && getenv("EMU_FORCE_DIRECT_FAIL")) verified = false;
#endif
        s.pol_near = polish_must_verify && !verified && !s.bad && pri < ep && dua < 1e4 * ed;
        s.pol_rounds = round + 1;
#ifdef MPC_EMU_DEBUG
        if (getenv("EMU_GI_TRACE")) fprintf(stderr, "  polish: pri %.3e (tol %.3e) dua %.3e (tol %.3e) bad %d\n", pri, ep, dua, ed, (int)s.bad);
#endif
        // an early polish of the exact mode is taken only when it is verified (an unverified one would bend the ADMM trajectory
        // that the final polish relies on); everything else follows OSQP: take it when it improves the residuals
        const bool take = polish_must_verify ? verified : ok;
        s.pol_ok = verified ? 1 : 0;
        s.status_polish = take ? 1 : -1;
        if (take) { s.pri_res = pri; s.dua_res = dua; }
#ifndef MPC_X_KEEPBAD
        s.bad = 0;     // a breakdown inside the polish (non-positive pivot of the reduced system) only fails the polish (polish.c:263-273);
                       // bad inputs were caught by the ADMM system's factorisation before any polish is tried
#endif
      }
    });
    if (!(ROUNDS && (s.pol_near || (direct && !s.pol_ok)) && round < MPC_EXACT_ROUNDS + (xn_given ? 1 : 0))) break;      // (the direct check is not one of the refinement rounds)
    }
    ex.par([&](Th &t) {
      if (s.status_polish == 1 && t.foot) {
#pragma unroll
        for (int c = 0; c < 3; ++c) t.x[c] = t.xp[c];
#pragma unroll
        for (int r = 0; r < 5; ++r) { t.z[r] = t.zp[r]; t.y[r] = t.yp[r]; }
      }
    });
  }
  static constexpr MPC_HD int pk3(int r, int c) {   // packed symmetric 3 x 3: 00 01 02 11 12 22
    return r <= c ? (r == 0 ? c : (r == 1 ? 2 + c : 5)) : (c == 0 ? r : (c == 1 ? 2 + r : 5));
  }

  // ================================ 5. exact mode: the optimal active set by a dual active-set method =================================
  // The reference's qpOASES branch (mpc_osqp.cc:797-947) returns THE optimum of the strictly convex QP; which route leads there is
  // free.  Goldfarb-Idnani's dual method on the problem with the fixed feet eliminated (rows with l = u: the swing feet, which that
  // branch eliminates as well, :838-856): start at the unconstrained minimum x = -H^-1 q with an empty working set; while some row
  // is violated, take the most violated one, n_p, and move along  z = H^-1 (n_p - N r),  r = (N^T H^-1 N)^-1 N^T H^-1 n_p  (N: the
  // working set's normals) until the row is satisfied -- add it -- or a multiplier of the working set reaches zero -- drop that row
  // and go on towards n_p.  Every iterate is optimal for the rows it holds, the dual objective rises strictly, so the method ends at
  // the optimum after about as many steps as it has active rows (25-60 here; ADMM needs 250+ iterations to pin the same set).
  //   H^-1 is the polish's operator for "no row active on a free foot, every row active on a fixed one" (polish_factor / omega_apply:
  //   Omega = C (I - V V^T + V M^-1 V^T) C^T with C C^T = (c alpha D^2 + delta)^-1 per free foot): ONE factorisation for the whole
  //   method, two applications per added row.  The delta keeps |C C^T| <= 1e6 like in the polish; it bends the iterates by ~1e-6 but
  //   not the set they end on, and the polish that follows (on that set, refined against the unregularised system and verified
  //   against the optimality conditions at 1e-10) is what produces the returned point -- or rejects the set, in which case the
  //   robot takes the ADMM route of run<true>().
  //   The inverse of the Gram matrix N^T H^-1 N is kept explicitly (packed, one lane per slot): bordering adds a row, a rank-one
  //   downdate drops one, free slots keep zero rows -- no triangular solves, nothing to compact.
  // Lane roles: foot lanes own x, their rows' status (t.act: -1 lower, +1 upper) and slots; lane i < NWMAX also serves slot i.
  static constexpr int kGiMaxPass = 8 * NF;        // adds + drops (observed: <= 1.3 x the final set)
  static constexpr int kSeedStep0 = 8;             // seed_working_set: the first slot of every step, in the second scratch row from here on
  static constexpr bool kSeedDirectGram = MPC_SEED_DIRECT_GRAM && 12 * GiShared<H>::NWMAX <= Sh::PARTLEN && 3 * GiShared<H>::NWMAX <= (int)(sizeof(Sh::dxy) / sizeof(double)) &&
                                          kSeedStep0 + H + 1 <= GiShared<H>::TL;
  MPC_HD double gi_ci(int i, int j) const { return i >= j ? gi->ci[i * (i + 1) / 2 + j] : gi->ci[j * (j + 1) / 2 + i]; }
  // t.pw <- H^-1 r for the per-foot vector r given by rv(t) (zero on fixed feet whatever rv says: C = 0 there)
  template <class RV>
  MPC_HD void gi_apply(RV &&rv) {
    ex.seq([&](Th &t) {
      if (t.foot) {
        double r3[3];
        rv(t, r3);
        ct_mul(t.pC, r3, t.pt);
        put_g(t, t.pt);
      }
    });
    omega_apply();
  }
  // row r of the scaled cone block (see a_mul).  (Masked sums, not a chain of selects: with a run-time r the compiler turns select chains
  // over array elements into an indexed load, which puts the array into scratch memory.)
  static MPC_HD double gi_pick(const double *v, int r) {   // v[r], r = 0 .. 4
    double o = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) o += (k == r) ? v[k] : 0.0;
    return o;
  }
  static MPC_HD void gi_row(const double *a, int r, double *n) {
    const double m0 = r == 0 ? 1.0 : 0.0, m1 = r == 1 ? 1.0 : 0.0, m2 = r == 2 ? 1.0 : 0.0, m3 = r == 3 ? 1.0 : 0.0, m4 = r == 4 ? 1.0 : 0.0;
    n[0] = m0 * a[0] + m1 * a[2];
    n[1] = m2 * a[4] + m3 * a[6];
    n[2] = (m0 * a[1] + m1 * a[3]) + (m2 * a[5] + m3 * a[7]) + m4 * a[8];
  }
  // ci <- ci + alpha u u^T on the slots below hi, except row / column `skip` (the slot being added or dropped, which the caller writes):
  // the packed lower triangle in 8 x 8 blocks, one entry per lane -- every lane has work in every block, where a row-per-lane
  // form runs as long as its longest row.
  MPC_HD void gi_rank1(const Th &t, const double *u, double alpha, int hi, int skip) {
    const int l = t.tid & 63, li = l >> 3, lj = l & 7, w = t.tid >> 6;
    constexpr int NWAVE = T / 64;
    const int nb = (hi + 7) >> 3;
    for (int bi = 0, b = 0; bi < nb; ++bi)
      for (int bj = 0; bj <= bi; ++bj, ++b) {
        if (NWAVE > 1 && b % NWAVE != w) continue;
        const int i = 8 * bi + li, j = 8 * bj + lj;
        if (j <= i && i < hi && i != skip && j != skip) {
          double *e = gi->ci + (i * (i + 1) / 2 + j);
          *e += (alpha * u[i]) * u[j];
        }
      }
  }
  // the empty working set (first: also what a foot lane derives once per call -- row norms, tolerances, the C of H^-1)
  MPC_HD void gi_clear(bool first) {
    GiShared<H> &g = *gi;
    constexpr int NW = GiShared<H>::NWMAX;
    double *const gd = s.fr, *const gr = s.fr + NW, *const glam = s.fr + 2 * NW;
    ex.par([&](Th &t) {
      for (int i = t.tid; i < NW * (NW + 1) / 2; i += T) g.ci[i] = 0.0;
      if (t.tid < NW) { g.owner[t.tid] = -1; glam[t.tid] = 0.0; gd[t.tid] = 0.0; gr[t.tid] = 0.0; }
      if (t.tid == 0) {
        g.hi = 0; g.converged = 0; g.fail = 0; g.passes = 0; g.adds = 0; g.drops = 0;
        g.freem[0] = NW >= 64 ? ~0ull : ((1ull << (NW & 63)) - 1); g.freem[1] = NW > 64 ? ((1ull << (NW - 64)) - 1) : 0ull;
      }
      if (t.foot) {
        const int fixed = ((t.tyb & 0x3ff) == 0x2aa);                 // all five rows are equalities (set_rho_vec's test: u - l < 1e-4)
        t.gfix = fixed;
        if (!first) {
#pragma unroll
          for (int r = 0; r < 5; ++r) { t.act[r] = fixed ? -1 : 0; t.gslot[r] = -1; }
        } else {
        double a[9], lo[5], up[5];
        foot_a(t, a);
        foot_bounds(t, lo, up);
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          double n[3];
          gi_row(a, r, n);
          t.act[r] = fixed ? -1 : 0; t.gslot[r] = -1;
          t.grn[r] = fast_rsqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
          t.gtol[r] = 1e-9 * dmax(1.0, dmax(fabs(lo[r]), up[r] < kInfty * kMinScaling ? fabs(up[r]) : 0.0));
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) t.pC[k] = 0.0;
        if (!fixed) {
#pragma unroll
          for (int c = 0; c < 3; ++c) t.pC[4 * c] = fast_rsqrt(s.calpha * Dat(t, c) * Dat(t, c) + MPC_GI_DELTA);
        }
        }
      }
    });
  }
  // ---- a warm start of the working set.  The branch this mode stands for is cold on every call (mpc_osqp.cc:906-919), and so is the RESULT here
  // -- the optimum is unique, whatever set the method starts from --, but consecutive calls of a controller end on working sets that share most
  // of their rows once the previous one is moved by one horizon step (23 of 24 rows seeded, 3 of them dropped again, 5 passes left of 31:
  // profiles/r04_active_set_persistence.txt).  seedrec holds, per foot, the set the previous call ended on (seed_code); the start is
  //   x = x0 + H^-1 N lam,  lam = (N^T H^-1 N)^-1 (b - N^T x0) >= 0  on the seeded rows N  (x0 = -H^-1 q):
  // one application of H^-1 per seeded row (its column of the Gram matrix N^T H^-1 N, which is also what a regular pass computes for its
  // row: step B), the Gram matrix inverted in place by symmetric sweeps, rows with a negative multiplier dropped one at a time (the rank-one
  // downdate of a regular drop) -- then every invariant of the dual method holds (x optimal on its rows, multipliers >= 0) and the regular
  // passes take over.  A dependent seed (tiny pivot) abandons the warm start: the method starts empty, as before.
  // x += H^-1 N v  (v: one number per slot), A x with it
  MPC_HD void gi_move(int hi, const double *v) {
    (void)hi;
    gi_apply([&](Th &t, double *r3) {
      double a[9], w[5];
      foot_a(t, a);
#pragma unroll
      for (int r = 0; r < 5; ++r) w[r] = t.gslot[r] >= 0 ? (t.act[r] < 0 ? 1.0 : -1.0) * v[t.gslot[r]] : 0.0;
      at_mul(a, w, r3);
    });
    ex.par([&](Th &t) {
      if (t.foot) {
        double a[9];
        foot_a(t, a);
#pragma unroll
        for (int c = 0; c < 3; ++c) t.gx[c] += t.pw[c];
        a_mul(a, t.gx, t.gax);
      }
    });
  }
  // Put the iterate back ON its working rows:  d lam = G^-1 (b - N^T x),  x += H^-1 N d lam,  lam += d lam.  The seeded inverse comes out of up to 50 sweeps
  // of a matrix whose condition reaches 1e8, so steps taken with it leave the rows by ~1e-9 -- and an iterate that is off its rows stays there: the passes move
  // along the rows, and the polish that tests the final point refines inside their null space only (it reports a dual residual of |A| x 1e-9 for ever).
  MPC_HD void gi_on_rows(int hi, double *gd, double *gr, double *glam) {
    GiShared<H> &g = *gi;
    constexpr int NW = GiShared<H>::NWMAX;
    ex.par([&](Th &t) {
      if (t.foot && !t.gfix) {
        double lo[5], up[5];
        foot_bounds(t, lo, up);
#pragma unroll
        for (int r = 0; r < 5; ++r)
          if (t.gslot[r] >= 0) gd[t.gslot[r]] = t.act[r] < 0 ? lo[r] - t.gax[r] : t.gax[r] - up[r];
      }
    });
    ex.par([&](Th &t) {
      if (t.tid < NW && t.tid < hi) {
        double acc = 0.0;
        for (int j = 0; j < hi; ++j) acc += gi_ci(t.tid, j) * gd[j];
        gr[t.tid] = g.owner[t.tid] >= 0 ? acc : 0.0;
      }
    });
    gi_move(hi, gr);
    ex.par([&](Th &t) {
      if (t.tid < NW && t.tid < hi) {
        if (g.owner[t.tid] >= 0) glam[t.tid] = dmax(glam[t.tid] + gr[t.tid], 0.0);
        gd[t.tid] = 0.0; gr[t.tid] = 0.0;
      }
    });
  }
  static MPC_HD int seed_code(const int *act, int fixed) {      // rows: 2 bits each (0 free, 1 at its lower, 2 at its upper bound); bit 10: fixed foot; bit 11: valid
    int c = (1 << 11) | (fixed ? 1 << 10 : 0);
#pragma unroll
    for (int r = 0; r < 5; ++r) c |= (act[r] < 0 ? 1 : (act[r] > 0 ? 2 : 0)) << (2 * r);
    return c;
  }
  // ---- the seeded Gram matrix inverted on the matrix pipe (single-wavefront workgroups, K <= 48).  The symmetric sweep of seed_working_set in blocks of
  // FOUR pivots:  A_ij -= A_iK B A_Kj,  A_Kj -> B A_Kj (and its mirror),  A_KK -> -B,  B = A_KK^-1 (4 x 4, by Cholesky) -- after all blocks A = -G^-1.  The matrix is held
  // as TT x TT tiles of 16 x 16 in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane l: rows (l >> 4) + 4 r, r = 0 .. 3, column l & 15; padded with the
  // identity), and that layout makes the update ONE instruction per tile and block: the four pivot rows 4 q + (l >> 4) of a tile row are register q of every
  // lane, which is exactly the B-operand layout (B[k = l >> 4][col = l & 15]) -- and, the matrix being symmetric, also the A-operand layout of the transposed
  // block A_iK (A[row = l & 15][k = l >> 4]) of the tile in the mirrored position.  With B = L^-T L^-1 both operands are Z = L^-1 A_K,: , so
  //     acc(I, J) += (-Z_I)^T-as-A x Z_J-as-B.
  // Per block: the pivot rows go through LDS once (every lane needs the four rows of its column to form its element of Z and of B A_K,:), the 4 x 4
  // Cholesky is done redundantly by every lane, the new pivot rows go through LDS once more to be mirrored into the pivot columns.  ~200 instructions per four
  // pivots against ~500 per pivot of the scalar sweep (profiles/r04_exact_warm_start.txt).  A pivot below MPC_SEED_PIVOT of its original diagonal entry gives the
  // matrix back untouched (false): the scalar path, which can drop a dependent row, takes over.
  template <int N, int I = 0, class F>
  static MPC_HD void static_for(F &&f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, I + 1>(f); }
  }
  template <int TT, int GQ>
  MPC_HD bool mfma_groups(int K, double *rowsA, double *rowsB, double *coef) {
    if constexpr (GQ >= 4 * TT) return true;
    else {
      GiShared<H> &g = *gi;
      constexpr int G = GQ / 4, q = GQ % 4, base = 4 * GQ;
      if (base >= K) return true;      // (the rest is the identity padding)
      ex.par([&](Th &t) {      // the four pivot rows of every tile column
        const int gl = t.tid >> 4, c = t.tid & 15;
        static_for<TT>([&](auto J) { rowsA[(J.value * 4 + gl) * 16 + c] = t.gacc[(G * TT + J.value) * 4 + q]; });
      });
      ex.par([&](Th &t) {      // B = P^-1 of the 4 x 4 pivot block P = L L^T (every lane the same arithmetic; lane 0 hands out L^-1 and B)
        double P[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) P[a][b] = rowsA[(G * 4 + a) * 16 + 4 * q + b];
        bool ok = true;
        double L[4][4], m[4][4], ri[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          double d = P[a][a];
#pragma unroll
          for (int k = 0; k < a; ++k) d -= L[a][k] * L[a][k];
          const double d0 = base + a < K ? gi_ci(base + a, base + a) : 1.0;      // the diagonal as it was: the scale of the pivot test
          ok = ok && d > MPC_SEED_PIVOT * d0;
          ri[a] = fast_rsqrt(ok ? d : 1.0);
          L[a][a] = d * ri[a];
#pragma unroll
          for (int b = a + 1; b < 4; ++b) {
            double v = P[b][a];
#pragma unroll
            for (int k = 0; k < a; ++k) v -= L[b][k] * L[a][k];
            L[b][a] = v * ri[a];
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {      // m = L^-1 (lower)
          m[a][a] = ri[a];
#pragma unroll
          for (int b = 0; b < a; ++b) {
            double v = 0.0;
#pragma unroll
            for (int k = b; k < a; ++k) v += L[a][k] * m[k][b];
            m[a][b] = -v * ri[a];
          }
        }
        if (t.tid == 0) {
          g.fail = ok ? 0 : 1;
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              coef[4 * a + b] = b <= a ? m[a][b] : 0.0;
              double v = 0.0;      // B = m^T m
#pragma unroll
              for (int k = (a > b ? a : b); k < 4; ++k) v += m[k][a] * m[k][b];
              coef[16 + 4 * a + b] = v;
            }
        }
      });
      if (g.fail) return false;
      ex.par([&](Th &t) {      // my element of Z_J = L^-1 A_K,J and of the new pivot rows B A_K,J
        const int gl = t.tid >> 4, c = t.tid & 15;
        double lin[4], bin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { lin[k] = coef[4 * gl + k]; bin[k] = coef[16 + 4 * gl + k]; }
        static_for<TT>([&](auto J) {
          double z = 0.0, nr = 0.0;
#pragma unroll
          for (int k = 0; k < 4; ++k) { const double v = rowsA[(J.value * 4 + k) * 16 + c]; z += lin[k] * v; nr += bin[k] * v; }
          if (J.value == G && (c >> 2) == q) nr = -((c & 3) == 0 ? bin[0] : ((c & 3) == 1 ? bin[1] : ((c & 3) == 2 ? bin[2] : bin[3])));      // the pivot block: -B
          t.gz[J.value] = z; t.gnr[J.value] = nr;
          rowsB[(J.value * 4 + gl) * 16 + c] = nr;
        });
      });
      static_for<TT * TT>([&](auto IJ) {
        constexpr int I = IJ.value / TT, J = IJ.value % TT;
        ex.mfma16([](Th &t) { return -t.gz[I]; }, [](Th &t) { return t.gz[J]; }, [](Th &t) { return t.gacc + 4 * (I * TT + J); });
      });
      ex.seq([&](Th &t) {      // the pivot rows and, mirrored, the pivot columns
        const int gl = t.tid >> 4, c = t.tid & 15;
        static_for<TT>([&](auto J) { t.gacc[(G * TT + J.value) * 4 + q] = t.gnr[J.value]; });
        if ((c >> 2) == q) {
          static_for<TT>([&](auto I) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (!(I.value == G && r == q)) t.gacc[(I.value * TT + G) * 4 + r] = rowsB[(I.value * 4 + (c & 3)) * 16 + gl + 4 * r];
          });
        }
      });
      ex.par([](Th &) {});      // (rowsA / rowsB are rewritten by the next block)
      return mfma_groups<TT, GQ + 1>(K, rowsA, rowsB, coef);
    }
  }
  template <int TT>
  MPC_HD bool seed_inverse_mfma(int K) {
    if constexpr (Th::kGaccT < TT || (int)(sizeof(s.part) / sizeof(double)) < 2 * 64 * TT + 32) return false;
    else {
      GiShared<H> &g = *gi;
      double *rowsA = s.part, *rowsB = s.part + 64 * TT, *coef = s.part + 128 * TT;      // (the tile product's partials: free between two applications of H^-1)
      ex.seq([&](Th &t) {
        const int gl = t.tid >> 4, c = t.tid & 15;
        static_for<TT * TT>([&](auto IJ) {
          constexpr int I = IJ.value / TT, J = IJ.value % TT;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int R = 16 * I + gl + 4 * r, Cc = 16 * J + c;
            t.gacc[4 * IJ.value + r] = (R < K && Cc < K) ? gi_ci(R, Cc) : (R == Cc ? 1.0 : 0.0);
          }
        });
      });
      if (!mfma_groups<TT, 0>(K, rowsA, rowsB, coef)) return false;
      ex.par([&](Th &t) {      // G^-1 = -(the swept matrix): the packed lower triangle for the passes that follow
        const int gl = t.tid >> 4, c = t.tid & 15;
        static_for<TT * TT>([&](auto IJ) {
          constexpr int I = IJ.value / TT, J = IJ.value % TT;
          if constexpr (J <= I) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int R = 16 * I + gl + 4 * r, Cc = 16 * J + c;
              if (Cc <= R && R < K) g.ci[R * (R + 1) / 2 + Cc] = -t.gacc[4 * IJ.value + r];
            }
          }
        });
      });
      return true;
    }
  }
  // returns the number of slots in use (0: nothing seeded); hi / free masks / adds are the caller's loop variables
  MPC_HD int seed_working_set(int &hi, unsigned long long &free0, unsigned long long &free1, double *gd, double *gr, double *glam, double *gtmp, double *gtmp2) {
    GiShared<H> &g = *gi;
    constexpr int NW = GiShared<H>::NWMAX;
    ex.par([&](Th &t) { if (t.tid == 0) { g.seed_shift = 1; g.seed_same = 1; g.seed_k = 0; g.seeded = 0; } });
    ex.par([&](Th &t) {      // does the previous call's set fit this call's contact pattern -- moved by one horizon step, or as it is?
      if (t.foot) {
        const int k = t.fid >> 2, leg = t.fid & 3, last = k + 1 >= H;
        const int cs = seedrec[(last ? k : k + 1) * 4 + leg], cu = seedrec[t.fid];
        if (!((cs >> 11) & 1) || (!last && ((cs >> 10) & 1) != t.gfix)) g.seed_shift = 0;
        if (!((cu >> 11) & 1) || ((cu >> 10) & 1) != t.gfix) g.seed_same = 0;
        t.gbrow = cs; t.gbside = cu;
      }
    });
    const int mode = g.seed_shift ? 1 : (g.seed_same ? 2 : 0);
    if (!mode) return 0;
    MPC_SUBLAP(9, 9);      // (-DMPC_PROFILE_SUB=9: the seeding's parts in slots 1 .. 5, what came before it in slot 9)
    ex.par([&](Th &t) {      // my seeded rows (at most three per foot: its three variables), their number
      if (t.foot) {
        const int code = mode == 1 ? t.gbrow : t.gbside;
        double lo[5], up[5];
        foot_bounds(t, lo, up);
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          const int c = t.gfix ? 0 : (code >> (2 * r)) & 3;
          const bool take = c != 0 && cnt < 3 && (c == 1 ? lo[r] > -kInfty * kMinScaling : up[r] < kInfty * kMinScaling);
          t.act[r] = take ? (c == 1 ? -1 : 1) : (t.gfix ? -1 : 0);
          cnt += take ? 1 : 0;
        }
        gtmp[t.fid] = (double)cnt;
      }
    });
    ex.par([&](Th &t) {      // slots in foot order
      if (t.foot) {
        int base = 0, total = 0;
        for (int j = 0; j < NF; ++j) { const int c = (int)gtmp[j]; base += j < t.fid ? c : 0; total += c; }
        if ((t.fid & 3) == 0) gtmp2[kSeedStep0 + (t.fid >> 2)] = (double)(base < NW ? base : NW);      // first slot of my step (slots are in foot order)
        if (t.fid == 0) gtmp2[kSeedStep0 + H] = (double)(total < NW ? total : NW);
        if (!t.gfix) {
#pragma unroll
          for (int r = 0; r < 5; ++r)
            if (t.act[r] != 0) {
              if (base < NW) { t.gslot[r] = base; g.owner[base] = t.fid * 8 + r; } else t.act[r] = 0;
              ++base;
            }
        }
        if (t.fid == 0) g.seed_k = total < NW ? total : NW;
      }
    });
    const int K = g.seed_k;
    if (K == 0) return 0;
    MPC_SUBLAP(9, 1);
    // The Gram matrix G = N^T H^-1 N.  H^-1 = C (I - V (I - M^-1) V^T) C^T (omega_apply) and a normal touches one foot, so with u_i = C^T n_i
    // (three numbers), v_i = V_f^T u_i (six, in the wrench space of the row's step k_i) and vt_i = dl o v_i
    //     G_ij = [same foot] u_i^T u_j  -  [same step] (v_i^T v_j - 2 vt_i^T vt_j)  -  vt_i^T Mx(k_i, k_j) vt_j
    // with Mx the held tile (2 I - Mh^-1, see product<kHeld>): every entry is a 6 x 6 quadratic form of ONE tile, formed by the lane that holds
    // it -- no application of H^-1 at all, where a column per seeded row costs K of them (71 k of a seeded solve's 311 k cycles at h = 10).
    if constexpr (kSeedDirectGram) {
      double *const rvt = s.part, *const rv = s.part + 6 * NW, *const ru = s.dxy;      // per slot: vt, v, u
      ex.par([&](Th &t) {
        if (t.foot && !t.gfix) {
          double a[9], gf[18];
          foot_a(t, a);
          load_g(t, gf);
          const double *dl = s.dl + 6 * (t.fid >> 2);
#pragma unroll
          for (int r = 0; r < 5; ++r)
            if (t.gslot[r] >= 0) {
              double n[3], u[3];
              gi_row(a, r, n);
              const double sg = t.act[r] < 0 ? 1.0 : -1.0;
#pragma unroll
              for (int c = 0; c < 3; ++c) n[c] *= sg;
              ct_mul(t.pC, n, u);
              double *ov = rv + 6 * t.gslot[r], *ovt = rvt + 6 * t.gslot[r], *ou = ru + 3 * t.gslot[r];
#pragma unroll
              for (int q = 0; q < 6; ++q) {
                const double v = gf[3 * q] * u[0] + gf[3 * q + 1] * u[1] + gf[3 * q + 2] * u[2];
                ov[q] = v;
                ovt[q] = dl[q] * v;
              }
#pragma unroll
              for (int c = 0; c < 3; ++c) ou[c] = u[c];
            }
        }
      });
      ex.par([&](Th &t) {
        if (t.mact) {
          const int i0 = (int)gtmp2[kSeedStep0 + t.ti], i1 = (int)gtmp2[kSeedStep0 + 1 + t.ti], j0 = (int)gtmp2[kSeedStep0 + t.tj], j1 = (int)gtmp2[kSeedStep0 + 1 + t.tj];
          for (int j = j0; j < j1; ++j) {
            double vj[6], w[6], vrj[6], uj[3];
#pragma unroll
            for (int q = 0; q < 6; q += 2) MPC_LDS_LOAD128(rvt + 6 * j + q, vj[q], vj[q + 1]);
#pragma unroll
            for (int q = 0; q < 6; ++q) { vrj[q] = 0.0; if (q < 3) uj[q] = 0.0; }
            if (t.dia) {
#pragma unroll
              for (int q = 0; q < 6; q += 2) MPC_LDS_LOAD128(rv + 6 * j + q, vrj[q], vrj[q + 1]);
#pragma unroll
              for (int c = 0; c < 3; ++c) uj[c] = ru[3 * j + c];
            }
#pragma unroll
            for (int aa = 0; aa < TS; ++aa) {
              double acc = t.dia ? -2.0 * vj[aa] : 0.0;      // (Mx - 2 I) vt_j on the diagonal tile: -Mh^-1 vt_j
#pragma unroll
              for (int bb = 0; bb < TS; ++bb) acc += t.Mx[aa * TS + bb] * vj[bb];
              w[aa] = acc;
            }
            const int fj = g.owner[j] >> 3;
            for (int i = t.dia ? j : i0; i < i1; ++i) {
              double vi[6];
#pragma unroll
              for (int q = 0; q < 6; q += 2) MPC_LDS_LOAD128(rvt + 6 * i + q, vi[q], vi[q + 1]);
              double acc = 0.0;
#pragma unroll
              for (int q = 0; q < 6; ++q) acc -= vi[q] * w[q];
              if (t.dia) {
                double vri[6];
#pragma unroll
                for (int q = 0; q < 6; q += 2) MPC_LDS_LOAD128(rv + 6 * i + q, vri[q], vri[q + 1]);
#pragma unroll
                for (int q = 0; q < 6; ++q) acc -= vri[q] * vrj[q];
                if ((g.owner[i] >> 3) == fj) acc += (ru[3 * i] * uj[0] + ru[3 * i + 1] * uj[1]) + ru[3 * i + 2] * uj[2];
              }
              g.ci[i * (i + 1) / 2 + j] = acc;
            }
          }
        }
      });
    }
    bool by_columns = !kSeedDirectGram;
#ifdef MPC_EMU_DEBUG
    std::vector<double> dbgDirect;
    if (kSeedDirectGram && getenv("EMU_GRAM_CHECK")) { dbgDirect.assign(g.ci, g.ci + K * (K + 1) / 2); by_columns = true; }
#endif
    // (horizons whose LDS stages are too small for the slot records: a column per seeded row,  y = H^-1 n_j,  G_ij = n_i^T y)
    if (by_columns)
    for (int j = 0; j < K; ++j) {
      const int pf = g.owner[j] >> 3, pr = g.owner[j] & 7;
      gi_apply([&](Th &t, double *r3) {
        double a[9];
        foot_a(t, a);
        gi_row(a, pr, r3);
        int side = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) side += r == pr ? t.act[r] : 0;
        const double sg = (t.foot && t.fid == pf) ? (side < 0 ? 1.0 : -1.0) : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) r3[c] *= sg;
      });
      ex.par([&](Th &t) {
        if (t.foot && !t.gfix) {
          double a[9], av[5];
          foot_a(t, a);
          a_mul(a, t.pw, av);
#pragma unroll
          for (int r = 0; r < 5; ++r)
            if (t.gslot[r] >= j) g.ci[t.gslot[r] * (t.gslot[r] + 1) / 2 + j] = (t.act[r] < 0 ? 1.0 : -1.0) * av[r];
        }
      });
    }
#ifdef MPC_EMU_DEBUG
    if (!dbgDirect.empty()) {
      double worst = 0, big = 0;
      for (int i = 0; i < K; ++i) big = dmax(big, fabs(gi_ci(i, i)));
      for (int i = 0; i < K * (K + 1) / 2; ++i) worst = dmax(worst, fabs(dbgDirect[i] - g.ci[i]));
      fprintf(stderr, "  seed Gram matrix: K %d direct vs by columns: max |diff| %.3e (largest diagonal entry %.3e)\n", K, worst, big);
    }
#endif
    MPC_SUBLAP(9, 2);
    // its inverse, in place: a symmetric sweep per pivot (a -> -a^-1 after all of them), then the sign
#ifdef MPC_EMU_DEBUG
    std::vector<double> dbgG((size_t)K * K);
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) dbgG[(size_t)i * K + j] = gi_ci(i, j);
#endif
    bool inverted = false;
    if constexpr (T <= 64) {      // on the matrix pipe where the workgroup is one wavefront; false: a dependent row, or a seed beyond 48 rows -- the scalar sweep below
      bool try_mfma = true;
#ifdef MPC_EMU_DEBUG
      if (getenv("EMU_SEED_SCALAR")) try_mfma = false;      // (tests: the scalar sweep, which the single-wavefront workgroups otherwise reach only through a dependent row)
#endif
      if (try_mfma) inverted = K <= 16 ? seed_inverse_mfma<1>(K) : (K <= 32 ? seed_inverse_mfma<2>(K) : (K <= 48 ? seed_inverse_mfma<3>(K) : false));
      ex.par([&](Th &t) { if (t.tid == 0) g.fail = 0; });
    }
#ifdef MPC_EMU_DEBUG
    if (getenv("EMU_GI_TRACE") && inverted) {
      double worst = 0;
      for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) {
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += dbgG[(size_t)i * K + k] * gi_ci(k, j);
        worst = dmax(worst, fabs(acc - (i == j ? 1.0 : 0.0)));
      }
      fprintf(stderr, "  seeded inverse on the matrix pipe: K %d |G Ginv - I| %.3e\n", K, worst);
    }
#endif
    ex.par([&](Th &t) { if (t.tid < K) gr[t.tid] = gi_ci(t.tid, t.tid); });      // the diagonal as it was: the scale of the pivot test
    int live = K;
    for (int p = 0; p < K && !inverted; ++p) {
      ex.par([&](Th &t) { if (t.tid < K) gtmp[t.tid] = gi_ci(t.tid, p); });
      const double piv = gtmp[p];
      if (!(piv > MPC_SEED_PIVOT * gr[p])) {      // row p is (nearly) a combination of the rows before it: it leaves the seed -- its row and column
        ex.par([&](Th &t) {                        // become a free slot's zeros, which is the matrix without it (the sweeps so far never used it as a pivot)
          if (t.tid < K) g.ci[t.tid >= p ? t.tid * (t.tid + 1) / 2 + p : p * (p + 1) / 2 + t.tid] = 0.0;
          if (t.tid == 0) g.owner[p] = -1;
          if (t.foot) {
#pragma unroll
            for (int r = 0; r < 5; ++r) if (t.gslot[r] == p) { t.act[r] = 0; t.gslot[r] = -1; }
          }
        });
        if (p < 64) free0 |= 1ull << p; else free1 |= 1ull << (p - 64);
        --live;
        continue;
      }
      const double rp = fast_recip(piv);
      ex.par([&](Th &t) {
        gi_rank1(t, gtmp, -rp, K, p);
        if (t.tid < K) g.ci[t.tid >= p ? t.tid * (t.tid + 1) / 2 + p : p * (p + 1) / 2 + t.tid] = t.tid == p ? -rp : gtmp[t.tid] * rp;
      });
    }
    MPC_SUBLAP(9, 3);
    ex.par([&](Th &t) {
      if (!inverted) for (int i = t.tid; i < K * (K + 1) / 2; i += T) g.ci[i] = -g.ci[i];
      if (t.foot && !t.gfix) {      // b - N^T x0: what each seeded row is violated by at x0
        double lo[5], up[5];
        foot_bounds(t, lo, up);
#pragma unroll
        for (int r = 0; r < 5; ++r)
          if (t.gslot[r] >= 0) gd[t.gslot[r]] = t.act[r] < 0 ? lo[r] - t.gax[r] : t.gax[r] - up[r];
      }
    });
    ex.par([&](Th &t) {      // lam = G^-1 (b - N^T x0)
      if (t.tid < K) {
        double acc = 0.0;
        for (int j = 0; j < K; ++j) acc += gi_ci(t.tid, j) * gd[j];
        glam[t.tid] = acc;
      }
    });
    hi = K;
    // Rows the new problem does not hold at this point leave: the smallest multiplier goes, the rest follow it (the rank-one downdate of a
    // regular drop), until every multiplier is CLEARLY positive -- a seeded row is only as good as its computed multiplier, and one whose true
    // multiplier is -1e-8 of the largest would stay in the set for good (the regular passes only ever add violated rows and drop rows whose
    // multiplier they drive to zero); a row dropped here that the optimum does hold is violated again and comes back by a regular pass.
    ex.seq([&](Th &t) { t.gred[0] = (t.tid < K && g.owner[t.tid < NW ? t.tid : 0] >= 0) ? glam[t.tid < NW ? t.tid : 0] : 0.0; });
    ex.wg_argmax([](Th &t) { return t.gred; }, [](Th &t) -> int & { return t.gidx; }, gtmp);
    const double lam_floor = MPC_SEED_LAMBDA * ex.first().gred[0];
    for (int rep = 0; rep < K; ++rep) {
      ex.seq([&](Th &t) { t.gred[0] = (t.tid < K && g.owner[t.tid < NW ? t.tid : 0] >= 0) ? -glam[t.tid < NW ? t.tid : 0] : -kInfty; });
      ex.wg_argmax([](Th &t) { return t.gred; }, [](Th &t) -> int & { return t.gidx; }, gtmp);
      if (!(ex.first().gred[0] > -lam_floor)) break;
      const int k1 = ex.first().gidx;
      ex.par([&](Th &t) {
        if (t.tid < K) gtmp[t.tid] = t.tid == k1 ? 0.0 : gi_ci(t.tid, k1);
        if (t.tid == 0) { gtmp2[0] = fast_recip(gi_ci(k1, k1)); gtmp2[1] = glam[k1]; }
      });
      ex.par([&](Th &t) {
        gi_rank1(t, gtmp, -gtmp2[0], K, k1);
        if (t.tid < K) {
          const int i = t.tid;
          if (g.owner[i] >= 0 && i != k1) glam[i] -= gtmp2[1] * gtmp2[0] * gtmp[i];
          g.ci[i >= k1 ? i * (i + 1) / 2 + k1 : k1 * (k1 + 1) / 2 + i] = 0.0;
          if (i == k1) { g.owner[k1] = -1; glam[k1] = 0.0; gd[k1] = 0.0; }
        }
        if (t.foot) {
#pragma unroll
          for (int r = 0; r < 5; ++r) if (t.gslot[r] == k1) { t.act[r] = 0; t.gslot[r] = -1; }
        }
      });
      if (k1 < 64) free0 |= 1ull << k1; else free1 |= 1ull << (k1 - 64);
      --live;
    }
    MPC_SUBLAP(9, 4);
    // x = x0 + H^-1 N lam, then once more with the correction for what that left on the seeded rows (gi_on_rows)
    gi_move(K, glam);
    gi_on_rows(K, gd, gr, glam);
    ex.par([&](Th &t) { if (t.tid == 0) g.seeded = live; });
    MPC_SUBLAP(9, 5);
    ex.par([&](Th &t) { if (t.tid < K) { gd[t.tid] = 0.0; gr[t.tid] = 0.0; } });
    return K;
  }
  MPC_HD bool active_set() {
    GiShared<H> &g = *gi;
    constexpr int NW = GiShared<H>::NWMAX, TL = GiShared<H>::TL;
    static_assert(3 * NW + 2 * TL <= (int)(sizeof(s.fr) / sizeof(double)), "the method's vectors must fit Shared::fr");
    double *const gd = s.fr, *const gr = s.fr + NW, *const glam = s.fr + 2 * NW, *const gtmp = s.fr + 3 * NW, *const gtmp2 = s.fr + 3 * NW + TL;
    gi_clear(true);
    polish_factor();
    if (s.bad) return false;
    gi_apply([&](Th &t, double *r3) {
#pragma unroll
      for (int c = 0; c < 3; ++c) r3[c] = -t.q[c];
    });
    ex.seq([&](Th &t) {
      if (t.foot) {
        double a[9];
        foot_a(t, a);
#pragma unroll
        for (int c = 0; c < 3; ++c) t.gx[c] = t.pw[c];
        a_mul(a, t.gx, t.gax);
      }
    });
    // The method's scalars (slots in use, free-slot mask, multiplier of row p, step lengths) are kept by every thread in registers -- the
    // same values everywhere, from workgroup-wide reductions -- not in LDS, where one thread's read-modify-write chains would cost a
    // round trip each.
    bool new_p = true, converged = false, fail = false;
    int hi = 0, passes = 0, adds = 0, drops = 0;
    unsigned long long free0 = NW >= 64 ? ~0ull : ((1ull << (NW & 63)) - 1), free1 = NW > 64 ? ((1ull << (NW - 64)) - 1) : 0ull;
    double lam_p = 0.0;
    int seeded_k = 0;
    if (seedrec) {
      unsigned long long rel0 = 0, rel1 = 0;      // slots the seeding took and released again
      const int K = seeded_k = seed_working_set(hi, rel0, rel1, gd, gr, glam, gtmp, gtmp2);
      if (K > 0) {      // slots 0 .. K - 1 are in use, except the released ones
        const unsigned long long low0 = K >= 64 ? ~0ull : ((1ull << K) - 1), low1 = K > 64 ? ((1ull << (K - 64)) - 1) : 0ull;
        free0 = (free0 & ~low0) | rel0; free1 = (free1 & ~low1) | rel1;
      }
    }
    for (int pass = 0; pass < kGiMaxPass; ++pass) {
      if (new_p) {
        // ---- A. the most violated row outside the working set (violation over the row's norm)
        ex.seq([&](Th &t) {
          double best = -1.0;
          t.gbrow = 0; t.gbside = 0;
          if (t.foot && !t.gfix) {
            double lo[5], up[5];
            foot_bounds(t, lo, up);
#pragma unroll
            for (int r = 0; r < 5; ++r) {
              const double vl = lo[r] - t.gax[r], vu = t.gax[r] - up[r];
              const double v = dmax(vl, vu), sv = v * t.grn[r];
              if (t.act[r] == 0 && v > t.gtol[r] && sv > best) { best = sv; t.gbrow = r; t.gbside = vl >= vu ? -1 : 1; }
            }
          }
          t.gred[0] = best;
        });
        ex.wg_argmax([](Th &t) { return t.gred; }, [](Th &t) -> int & { return t.gidx; }, gtmp);
        if (!(ex.first().gred[0] > 0.0)) { converged = true; break; }
        ex.par([&](Th &t) { if (t.tid == t.gidx) { g.p_foot = t.fid; g.p_row = t.gbrow; g.p_side = t.gbside; } });
        lam_p = 0.0;
        MPC_SUBLAP(7, 1);
        // ---- B. v = H^-1 n_p;  d = N^T v,  gamma = n_p^T v      (n = side * row: the normal of "row >= l" is +row, of "row <= u" is -row)
        gi_apply([&](Th &t, double *r3) {
          double a[9];
          foot_a(t, a);
          gi_row(a, g.p_row, r3);
          const double sg = (t.foot && t.fid == g.p_foot) ? (g.p_side < 0 ? 1.0 : -1.0) : 0.0;
#pragma unroll
          for (int c = 0; c < 3; ++c) r3[c] *= sg;
        });
        ex.par([&](Th &t) {
          if (t.foot) {
            double a[9], av[5];
            foot_a(t, a);
#pragma unroll
            for (int c = 0; c < 3; ++c) t.gv[c] = t.pw[c];
            a_mul(a, t.gv, av);
#pragma unroll
            for (int r = 0; r < 5; ++r)
              if (t.gslot[r] >= 0) gd[t.gslot[r]] = (t.act[r] < 0 ? 1.0 : -1.0) * av[r];
            if ((t.foot && t.fid == g.p_foot)) g.gamma = (g.p_side < 0 ? 1.0 : -1.0) * gi_pick(av, g.p_row);
          }
        });
        MPC_SUBLAP(7, 2);
      }
      // ---- C. r = (N^T H^-1 N)^-1 d;  zeta = gamma - d^T r = n_p^T z;  the largest dual step t1 that keeps every multiplier >= 0
      ex.par([&](Th &t) {
        double dr = 0.0, ratio = kInfty;
        if (t.tid < NW) {
          const int i = t.tid, rowb = i * (i + 1) / 2;
          double acc = 0.0;
          for (int j0 = 0; j0 < hi; j0 += 8) {     // (zero rows at free slots; blocks of eight: all loads of a block in flight together)
            double cv[8], dv[8];
#pragma unroll
            for (int u = 0; u < 8; u += 2) MPC_LDS_LOAD128(gd + j0 + u, dv[u], dv[u + 1]);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + u < NW ? j0 + u : NW - 1; cv[u] = MPC_LDS_LOAD64(g.ci + (j <= i ? rowb + j : j * (j + 1) / 2 + i)); }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (j0 + u < hi ? cv[u] : 0.0) * dv[u];
          }
          const bool live = i < hi && g.owner[i] >= 0;
          gr[i] = live ? acc : 0.0;
          if (live) { dr = gd[i] * acc; if (acc > 0.0) ratio = glam[i] * fast_recip(acc); }
        }
        t.gred[0] = dr;
        t.gbviol = ratio;
      });
      ex.wg_sum([](Th &t) { return t.gred; }, gtmp);
      const double zeta = g.gamma - ex.first().gred[0];
      ex.seq([&](Th &t) { t.gred[0] = -t.gbviol; });
      ex.wg_argmax([](Th &t) { return t.gred; }, [](Th &t) -> int & { return t.gidx; }, gtmp);
      const double t1 = -ex.first().gred[0];
      const int k1 = ex.first().gidx;
      MPC_SUBLAP(7, 3);
      const bool moves = zeta > 1e-10 * g.gamma;      // else n_p is a combination of the working set's normals: z = 0
      // ---- D. z = H^-1 (n_p - N r)
      if (moves) {
        gi_apply([&](Th &t, double *r3) {
          double a[9], w[5];
          foot_a(t, a);
#pragma unroll
          for (int r = 0; r < 5; ++r) {     // the combination of my rows: +-1 for row p, -(+-r_slot) for my working rows
            double c = ((t.foot && t.fid == g.p_foot) && r == g.p_row) ? (g.p_side < 0 ? 1.0 : -1.0) : 0.0;
            if (t.gslot[r] >= 0) c -= (t.act[r] < 0 ? 1.0 : -1.0) * gr[t.gslot[r]];
            w[r] = c;
          }
          at_mul(a, w, r3);
        });
      }
      MPC_SUBLAP(7, 4);
      // ---- E. the step: t2 = (violation of row p now) / zeta brings the row to its bound; t = min(t1, t2)
      ex.par([&](Th &t) {      // (row p's foot lane publishes the violation: one LDS round trip, where a reduction takes six exchanges)
        if ((t.foot && t.fid == g.p_foot)) {
          double lo[5], up[5];
          foot_bounds(t, lo, up);
          const double lo_p = gi_pick(lo, g.p_row), up_p = gi_pick(up, g.p_row), ax_p = gi_pick(t.gax, g.p_row);
          g.p_viol = dmax(g.p_side < 0 ? lo_p - ax_p : ax_p - up_p, 0.0);
        }
      });
      const double zinv = moves ? fast_recip(zeta) : 0.0;
      const double t2 = moves ? g.p_viol * zinv : kInfty;
      const double ts = dmin(t1, t2);
      if (!(ts < kInfty)) { fail = true; break; }      // (an infeasible QP: not this path's business)
      const bool add = t2 <= t1;
      lam_p += ts; ++passes;
      int ps = -1;
      if (add) {   // the lowest free slot
        ps = free0 ? mpc_ffs64(free0) : (free1 ? 64 + mpc_ffs64(free1) : -1);
        if (ps < 0) { fail = true; break; }
        if (ps < 64) free0 &= ~(1ull << ps); else free1 &= ~(1ull << (ps - 64));
        ++adds;
      } else {
        if (k1 < 64) free0 |= 1ull << k1; else free1 |= 1ull << (k1 - 64);
        ++drops;
      }
      const int hi_new = add && hi < ps + 1 ? ps + 1 : hi;
      if (!add) ex.par([&](Th &t) {   // column k of the inverse, before the rows change
        if (t.tid < NW && t.tid < hi) gtmp[t.tid] = t.tid == k1 ? 0.0 : gi_ci(t.tid, k1);
        if (t.tid == 0) gtmp2[0] = fast_recip(gi_ci(k1, k1));
      });
      ex.par([&](Th &t) {
        if (t.foot && moves) {
          double a[9], az[5];
          foot_a(t, a);
          a_mul(a, t.pw, az);
#pragma unroll
          for (int c = 0; c < 3; ++c) t.gx[c] += ts * t.pw[c];
#pragma unroll
          for (int r = 0; r < 5; ++r) t.gax[r] += ts * az[r];
        }
        // bordering (add): the rest gains r r^T / zeta, then the new row -r / zeta -- element (ps, i) or (i, ps) by the lane that holds
        // r_i -- and 1 / zeta.  Drop slot k: a rank-one downdate with column k, then row / column k are zero again and the slot is free.
        if (add) gi_rank1(t, gr, zinv, hi_new, ps);
        else gi_rank1(t, gtmp, -gtmp2[0], hi, k1);
        if (t.tid < NW && t.tid < hi_new) {
          const int i = t.tid;
          const bool live = i < hi && g.owner[i] >= 0;
          if (live) glam[i] = dmax(glam[i] - ts * gr[i], 0.0);
          if (add) {
            const double zi = zinv, ri = gr[i] * zi;
            if (i == ps) { g.ci[ps * (ps + 1) / 2 + ps] = zi; g.owner[ps] = g.p_foot * 8 + g.p_row; glam[ps] = lam_p; gd[ps] = 0.0; }
            else g.ci[i > ps ? i * (i + 1) / 2 + ps : ps * (ps + 1) / 2 + i] = -ri;
          } else {
            g.ci[i >= k1 ? i * (i + 1) / 2 + k1 : k1 * (k1 + 1) / 2 + i] = 0.0;
            if (i == k1) { g.owner[k1] = -1; glam[k1] = 0.0; gd[k1] = 0.0; }
          }
        }
        if (t.foot) {
          if (add) {
            if ((t.foot && t.fid == g.p_foot)) {
#pragma unroll
              for (int r = 0; r < 5; ++r) if (r == g.p_row) { t.act[r] = g.p_side; t.gslot[r] = ps; }
            }
          } else {
#pragma unroll
            for (int r = 0; r < 5; ++r) if (t.gslot[r] == k1) { t.act[r] = 0; t.gslot[r] = -1; }
          }
        }
      });
      hi = hi_new;
      new_p = add;
      MPC_SUBLAP(7, 5);
    }
    if (seeded_k > 0 && converged && !fail) gi_on_rows(hi, gd, gr, glam);      // (see there; a cold start keeps its rows to rounding by itself)
    ex.par([&](Th &t) { if (t.tid == 0) { g.passes = passes; g.adds = adds; g.drops = drops; g.hi = hi; g.converged = converged; g.fail = fail; } });
#ifdef MPC_EMU_DEBUG
    if (getenv("EMU_GI_TRACE")) {
      double mn = kInfty, mx = 0; int cnt = 0;
      for (int i = 0; i < hi; ++i) if (g.owner[i] >= 0) { ++cnt; mn = dmin(mn, glam[i]); mx = dmax(mx, glam[i]); }
      fprintf(stderr, "  working set %d rows, multipliers min %.3e max %.3e\n", cnt, mn, mx);
    }
#endif
    return converged && !fail && !s.bad;
  }

  // ================================ driver ======================================================================
  // kCheck iterations between termination checks (osqp.c:417-517 checks when iter % 25 == 0).  P_s x, which only the dual
  // residual needs, is formed at the check (one Theta product) instead of being carried through every iteration; that product
  // uses the exchange registers of the iteration, so the right-hand side is handed over again afterwards (admm_prepare
  // recomputes exactly the values the last iteration left).
  // Returns true when the exact mode wants an early polish (the loop is left with the iterate unfinished).
  template <bool EXACT>
  MPC_HD bool admm_until_done(int &iter, int &stable) {
    while (!s.done && !s.bad && iter < max_iter) {
      y_scaled(true);
      for (int k = 0; k < kCheck - 1; ++k) admm_iter();
      admm_iter<true>();
      y_scaled(false);
      iter += kCheck;
      lap(8);
      mul_P([](Th &t) { return t.x; }, [](Th &t) { return t.px; });
      MPC_SUBLAP(6, 1);
      residuals<true>([](Th &t) { return t.x; }, [](Th &t) { return t.z; }, [](Th &t) { return t.y; }, [](Th &t) { return t.px; });
      MPC_SUBLAP(6, 2);
      check_and_adapt(iter);
      if (MPC_PROFILE_SUB == 6) lap(3); else
      lap(10);
      if (!s.done) {
        if constexpr (EXACT) {
          ex.par([&](Th &t) {   // the active-set guess of my rows (polish.c:36-52), as a base-3 code; did it change since the last check?
            if (t.foot) {
              double lo[5], up[5];
              foot_bounds(t, lo, up);
              int sig = 0;
#pragma unroll
              for (int r = 4; r >= 0; --r) sig = 3 * sig + ((t.z[r] - lo[r] < -t.y[r]) ? 0 : ((up[r] - t.z[r] < t.y[r]) ? 2 : 1));
              if (sig != t.sig) s.sig_changed = 1;
              t.sig = sig;
            }
          });
          stable = s.sig_changed ? 0 : stable + 1;
          if (stable >= kStableChecks && s.loose_ok) return true;
        }
        if (s.rho_new > 0) {          // osqp_update_rho: new rho_vec, refactor
          ex.par([&](Th &t) { if (t.tid == 0) { s.rho = s.rho_new; s.rho_updates++; } });
          set_rho_vec();
          factor();
        }
        admm_prepare();
        lap(8);
      }
    }
    return false;
  }
  // The force the reference returns for variable c of my foot.  OSQP branch: `-x` of the unscaled solution written as 0 - D x
  // (mpc_osqp.cc:789-790).  qpOASES branch (exact mode): an eliminated foot -- every row an equality with l = u = 0, the swing feet,
  // mpc_osqp.cc:838-856 -- is `qp_sol = 0.0f`, and the copy-out NEGATES it (:926, 940-942): the sign bit is set, -0.0; every other
  // variable is the negation of the solver's value.
  MPC_HD double force_out(const Th &t, int c) const {
    if (eps_exact > 0) return ((t.tyb & 0x3ff) == 0x2aa) ? -0.0 : -(Dat(t, c) * t.x[c]);
    return 0.0 - Dat(t, c) * t.x[c];
  }
  // outputs + persistent state (store_solution, auxil.c:528-561; mpc_osqp.cc:788-790: forces = -x).  A non-convex / non-finite
  // problem has no solution: OSQP cold-starts the iterates (auxil.c:539-563); here the whole record is cleared, so that the
  // robot's next call is the cold "osqp_setup" call on clean data (with NaN inputs the vendored OSQP itself stays poisoned).
  MPC_HD void store(long long t0) {
    tc[15] = MPC_CLOCK() - t0;
    if (s.first && eps_exact == 0.0) tc[15] = -tc[15];   // a cold solve is no predictor of the robot's next (warm) one: negative = ignored by the dispatch order (order_block).
                                                         // (The exact mode clears the record on every call, so every solve is "first": there the last solves do order the launch --
                                                         //  with a seeded working set a robot that needed 30 passes last time tends to need them again.)
    const bool failed = s.bad || s.status == kStNonCvx;
    // (the reference's qpOASES branch returns its vector whatever the solver's status, mpc_osqp.cc:906-947: in the exact mode an iterate that
    // ran out of iterations is written too, with its status)
    const bool solved = (s.status == kStSolved || (eps_exact > 0 && (s.status == kStSolvedInaccurate || s.status == kStMaxIter))) && !failed;
    ex.par([&](Th &t) {      // the record and the forces into the LDS stage (see load) ...
      if (t.foot) {
        const int f = t.fid;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          s.part[SL + 3 * f + c] = force_out(t, c);
          s.part[3 * f + c] = failed ? 0.0 : t.x[c];
          s.part[N + 2 * M + 3 * f + c] = failed ? 0.0 : qp[C::QP_Q + 3 * f + c];
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) { s.part[N + 5 * f + r] = failed ? 0.0 : t.z[r]; s.part[N + M + 5 * f + r] = failed ? 0.0 : t.y[r]; }
      }
      if (t.tid == 0) { s.part[2 * N + 2 * M] = failed ? 0.0 : s.rho; s.part[2 * N + 2 * M + 1] = failed ? 0.0 : 1.0; }
    });
    ex.par([&](Th &t) {      // ... and out with lane-contiguous addresses
      for (int i = t.tid; i < SL; i += T) MPC_GST(state + i, s.part[i]);
      if (solved) for (int i = t.tid; i < N; i += T) MPC_GST(forces + i, s.part[SL + i]);
      if (t.tid == 0) {
        const int iv[8] = {s.iter, s.bad ? kStNonCvx : s.status, s.status_polish, s.rho_updates, s.nfact, s.first, eps_exact > 0 ? s.pol_rounds : 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) MPC_GST(info + k, iv[k]);
        if (prof) for (int k = 0; k < kProfLen; ++k) if ((k < 1 || k > 5 || (MPC_PROFILE_SUB >= 6 && MPC_PROFILE_SUB != 8)) && !(MPC_PROFILE_SUB && MPC_PROFILE_SUB <= 4 && k >= 9 && k <= 13)) MPC_GST(prof + k, tc[k]);   // (1 .. 5, and 9 .. 13 of a prep sub-profile: the prep kernel's)
      }
    });
  }
  // OSQP mode up to the polish: load, factor, ADMM until a check ends it (osqp.c:354-568)
  MPC_HD void admm_part() {
    load();
    set_rho_vec();
    factor();
    lap(9);
    admm_prepare();
    lap(8);
    static_assert(kMaxIter % kCheck == 0, "the check falls on the last iteration");
    int iter = 0, stable = 0;
    admm_until_done<false>(iter, stable);
    if (!s.done && !s.bad) {   // max_iter reached (osqp.c:563-568): a second look at the last check's residuals with every tolerance
      check_and_adapt<true>(iter);   // times ten (-> *_INACCURATE), else MAX_ITER_REACHED; only SOLVED counts for the reference
      ex.par([&](Th &t) { if (t.tid == 0 && !s.done) s.status = kStMaxIter; });
    }
  }
  // (the body of run() is kept in one piece rather than built from admm_part() + store(): at the 256-register cap of the long horizons
  // the allocator's spill decisions inside the hot loops move with the code shape -- tools/isa_census.py, tests/test_isa_budget.py)
  template <bool EXACT = false>
  MPC_HD void run() {
    const long long t0 = MPC_CLOCK();
    tlast = t0;
    load();
    set_rho_vec();
    factor();
    lap(9);
    admm_prepare();
    lap(8);
    static_assert(kMaxIter % kCheck == 0, "the check falls on the last iteration");
    int iter = 0, stable = 0;
    if constexpr (!EXACT) {
      admm_until_done<false>(iter, stable);
      if (!s.done && !s.bad) {   // max_iter reached (osqp.c:563-568): a second look at the last check's residuals with every tolerance
        check_and_adapt<true>(iter);   // times ten (-> *_INACCURATE), else MAX_ITER_REACHED; only SOLVED counts for the reference
        ex.par([&](Th &t) { if (t.tid == 0 && !s.done) s.status = kStMaxIter; });
      }
      if (s.status == kStSolved && !s.bad) polish();
    } else {
      ex.seq([&](Th &t) { t.sig = -1; });
      for (;;) {
        const bool early = admm_until_done<true>(iter, stable);
        if (s.bad) break;
        if (!early) {                 // converged at the ADMM floor (then OSQP's own polish rule), or out of iterations
          if (!s.done) ex.par([&](Th &t) { if (t.tid == 0) s.status = s.loose_ok ? kStSolvedInaccurate : kStMaxIter; });   // (out of iterations: the iterate is the result, see store)
          else if (s.status == kStSolved) polish();
          break;
        }
        polish_must_verify = true;
        polish<true>();
        polish_must_verify = false;
        if (s.bad) break;
        if (s.pol_ok) {
          ex.par([&](Th &t) { if (t.tid == 0) s.status = kStSolved; });
          break;
        }
        // rejected: on with ADMM from the untouched iterate (the tiles hold the polish's factorisation: K again, with the pending rho
        // if there is one); no further attempt until the guess has changed and settled again
        stable = -(1 << 20);
        if (s.rho_new > 0) {
          ex.par([&](Th &t) { if (t.tid == 0) { s.rho = s.rho_new; s.rho_updates++; } });
          set_rho_vec();
        }
        factor();
        admm_prepare();
      }
    }
    lap(14);
    tc[15] = MPC_CLOCK() - t0;
    if (s.first && eps_exact == 0.0) tc[15] = -tc[15];   // a cold solve is no predictor of the robot's next (warm) one: negative = ignored by the dispatch order (order_block).
                                                         // (The exact mode clears the record on every call, so every solve is "first": there the last solves do order the launch --
                                                         //  with a seeded working set a robot that needed 30 passes last time tends to need them again.)
    // outputs + persistent state (store_solution, auxil.c:528-561; mpc_osqp.cc:788-790: forces = -x): see store()
    ex.par([&](Th &t) {
      const bool failed = s.bad || s.status == kStNonCvx;
      // (the reference's qpOASES branch returns its vector whatever the solver's status, mpc_osqp.cc:906-947: in the exact mode an iterate that
      // ran out of iterations is written too, with its status)
      const bool solved = (s.status == kStSolved || (eps_exact > 0 && (s.status == kStSolvedInaccurate || s.status == kStMaxIter))) && !failed;
      if (t.foot) {
        const int f = t.fid;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (solved) forces[3 * f + c] = force_out(t, c);
          state[3 * f + c] = failed ? 0.0 : t.x[c];
          state[N + 2 * M + 3 * f + c] = failed ? 0.0 : qp[C::QP_Q + 3 * f + c];
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) { state[N + 5 * f + r] = failed ? 0.0 : t.z[r]; state[N + M + 5 * f + r] = failed ? 0.0 : t.y[r]; }
      }
      if (t.tid == 0) {
        const bool failed = s.bad || s.status == kStNonCvx;
        state[2 * N + 2 * M] = failed ? 0.0 : s.rho;
        state[2 * N + 2 * M + 1] = failed ? 0.0 : 1.0;
        info[0] = s.iter; info[1] = s.bad ? kStNonCvx : s.status; info[2] = s.status_polish; info[3] = s.rho_updates;
        info[4] = s.nfact; info[5] = s.first; info[6] = 0; info[7] = 0;
        if (prof) for (int k = 0; k < kProfLen; ++k) if ((k < 1 || k > 5 || (MPC_PROFILE_SUB >= 6 && MPC_PROFILE_SUB != 8)) && !(MPC_PROFILE_SUB && MPC_PROFILE_SUB <= 4 && k >= 9 && k <= 13)) prof[k] = tc[k];   // (1 .. 5, and 9 .. 13 of a prep sub-profile: the prep kernel's)
      }
    });
  }

  // Exact mode, first attempt: the active-set method, then the polish on its set as the returned point -- if that point passes the
  // optimality test (eps_exact; after exact()).  false: nothing was written, the robot takes run<true>() (a second launch).
  MPC_HD bool run_active_set() {
    const long long t0 = MPC_CLOCK();
    tlast = t0;
    load();
    lap(9);
    const bool found = active_set();
    if (seedrec) ex.par([&](Th &t) {      // what the next call of this robot may start from (before the polish, which re-uses t.act)
      if (t.foot) seedrec[t.fid] = found ? seed_code(t.act, t.gfix) : 0;
    });
    lap(8);
    bool ok = false;
    if (found) {
      ex.par([&](Th &t) { if (t.tid == 0) { s.pri_res = kInfty; s.dua_res = kInfty; s.status = kStSolved; } });
      act_given = true; polish_must_verify = true; xn_given = MPC_EXACT_DIRECT;
      if (xn_given) ex.par([&](Th &t) {      // (through LDS, not registers: nothing of the method stays live across the polish set-up)
        if (t.foot) {
#pragma unroll
          for (int c = 0; c < 3; ++c) s.dxy[pidx(c, t.fid)] = t.gx[c];
        }
      });
      polish<true>();
      act_given = false; polish_must_verify = false; xn_given = false;
      ok = s.pol_ok && s.status_polish == 1;
    }
    lap(14);
#ifdef MPC_EMU_DEBUG
    if (!ok && getenv("EMU_GI_DUMP")) {   // debugging: the dual method's iterate and working set of a robot whose set was rejected
      ex.par([&](Th &t) {
        if (t.foot) {
          for (int c = 0; c < 3; ++c) forces[3 * t.fid + c] = 0.0 - Dat(t, c) * t.gx[c];
          for (int r = 0; r < 5; ++r) state[N + 5 * t.fid + r] = t.act[r];
        }
      });
    }
#endif
    if (!ok) return false;
    ex.par([&](Th &t) { if (t.tid == 0) { s.iter = gi->passes; s.rho_updates = 0; } });
    store(t0);
    return true;
  }

  // ---- the OSQP-mode solve as two jobs of a persistent wave (mpc_batch.hip: mpc_solve_jobs_kernel).  The ADMM part of a solve takes
  // 25 ... 250+ iterations, the polish that follows is the same work for every robot and needs nothing of the ADMM part but its
  // result (x, z, y and the two residuals its acceptance test compares with, polish.c:306-345): a separate job that any wave can
  // run, which is what fills the tail of a launch.  admm_job leaves the complete result of a solve whose polish "has not
  // happened yet" (status_polish 0); polish_job re-loads the problem, polishes and, if OSQP would take the polished point,
  // overwrites x, z, y and the forces.  Returns true when a polish job has to follow.
  long long t_start = 0;   // (profiling builds: when the job started)
  MPC_HD bool admm_job() {
    const long long t0 = MPC_CLOCK();
    tlast = t0; if (MPC_PROFILE_SUB == 8) t_start = t0;
    admm_part();
    lap(14);
    store(t0);
    const bool pol = s.status == kStSolved && !s.bad;
    if (pol) ex.par([&](Th &t) { if (t.tid == 0) { MPC_GST(jobrec, s.pri_res); MPC_GST(jobrec + 1, s.dua_res); } });
    return pol;
  }
  MPC_HD void polish_job() {
    const long long t0 = MPC_CLOCK();
    tlast = t0; if (MPC_PROFILE_SUB == 8) t_start = t0;
    load();          // (x, z, y of the state record are the ADMM part's result)
    ex.par([&](Th &t) {
      if (t.tid == 0) { s.pri_res = MPC_GLD(jobrec); s.dua_res = MPC_GLD(jobrec + 1); s.status = kStSolved; s.nfact = MPC_GLD(info + 4); }
    });
    lap(9);
    polish();
    lap(14);
    ex.par([&](Th &t) {
      if (s.status_polish == 1 && t.foot) {      // OSQP takes the polished point: x, z, y and the forces again, through the stage (see load)
        const int f = t.fid;
#pragma unroll
        for (int c = 0; c < 3; ++c) { s.part[SL + 3 * f + c] = 0.0 - Dat(t, c) * t.x[c]; s.part[3 * f + c] = t.x[c]; }
#pragma unroll
        for (int r = 0; r < 5; ++r) { s.part[N + 5 * f + r] = t.z[r]; s.part[N + M + 5 * f + r] = t.y[r]; }
      }
    });
    ex.par([&](Th &t) {
      if (s.status_polish == 1) {
        for (int i = t.tid; i < N + 2 * M; i += T) MPC_GST(state + i, s.part[i]);
        for (int i = t.tid; i < N; i += T) MPC_GST(forces + i, s.part[SL + i]);
      }
      if (t.tid == 0) {
        MPC_GST(info + 2, s.status_polish); MPC_GST(info + 4, s.nfact);
        if (prof) {   // (slot 0: the polish job as a whole; the sections add to the ADMM job's under MPC_SECTION_PROFILE)
          for (int k = 6; k < 15; ++k) if (k != 8 && k != 10) MPC_GST(prof + k, MPC_GLD(prof + k) + tc[k]);
          MPC_GST(prof, MPC_GLD(prof) + (MPC_CLOCK() - t0));
        }
      }
    });
  }
  double *jobrec = nullptr;   // [2] the ADMM part's residuals, handed to the polish job (scale record tail)
};

}  // namespace mpc
