// mpc_core.h -- one robot's convex-MPC contact-force solve, written once as sequences of
// barrier-separated phases over the threads of one workgroup: Assembler (QP assembly, its own kernel) and
// Solver (the OSQP algorithm).
//
//   Device build (mpc_batch.hip): Exec::par(f) = { f(thread); __syncthreads(); } -- the per-thread
//   state lives in VGPRs, Shared<H> in LDS.
//   Host emulation (tests/emu): Exec::par(f) runs f for every emulated thread (in forward or reverse
//   order, to expose intra-phase races); used only by the CPU tests.
//
// What is computed (reference boundary: MPC_Controller/convex_MPC/mpc_osqp.cc:578-796,
// ConvexMpc::ComputeContactForces, OSQP branch):
//   1. single-rigid-body QP assembly (mpc_osqp.cc:606-688): x0, x_ref, A/B, exact exp, A^k B, q, P
//      (P by cumulative diagonal sums = the reference's block recursion :387-434 in the same order)
//   2. the OSQP 0.6.0 algorithm the reference calls on it (extern/osqp/src): Ruiz scaling
//      (scaling.c:44-156), ADMM (auxil.c:164-228), residuals/termination (auxil.c:243-362,684-793),
//      rho adaptation (auxil.c:13-77), polish (polish.c) -- restated for dense algebra:
//        KKT solve      -> x~ = Kinv (sigma x - q + A^T(R z - y)),  K = P + sigma I + A^T R A,  z~ = A x~
//        Kinv           -> explicit inverse by symmetric sweeps on register tiles
//        polish         -> delta-regularised refinement in the null space of the active rows
//   All arithmetic is fp64: an fp32 ADMM does not reproduce OSQP's iterates (oracle/README).
//
// Thread layout: every n x n matrix on the path (P_s, K, -K^{-1}, H, -H^{-1}; n = 12 H) is SYMMETRIC and is
// held as the lower triangle of a G x G grid (G = 2 H) of 6 x 6 register tiles (2 feet x 2 feet), one tile
// per thread (four at h = 20): tile index ti (ti + 1) / 2 + tj holds tile (ti, tj), tj <= ti; thread tid owns tiles
// tid, tid + MTH, ...; a diagonal tile is stored in full.  An off-diagonal tile stands for itself and for its transpose, so
//   * a symmetric sweep step costs 36 FMAs per thread (half of a full-matrix update) for 12 LDS reads,
//   * a matrix-vector product uses every tile twice (T v_cols -> rows, T^T v_rows -> cols),
//   * P_s in HBM is one contiguous 288-byte run per thread (tile-major), half the bytes of the full matrix.
// Vector phases use tid < n and the constraint rows tid, tid + T, ... < m (Solver::for_rows).
//
// Per-horizon tuning knobs (Cfg: NT, kPinMask, kLoopExitFence, kColumnStore64, kQInLds, kRhoPerType) exist because the kernel
// lives at the edge of the register file: they do not change any arithmetic, only how the compiler allocates registers, and
// are set from measurements (tools/isa_census.py; tests/test_isa_budget.py guards the outcome).
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MPC_HD __host__ __device__ __forceinline__
#else
#define MPC_HD inline
#endif
// Opaque identity on an int: stops the optimiser from hoisting comparisons against it out of
// unrolled loops (which would pin dozens of 64-bit lane masks in SGPRs).
#if defined(__HIP_DEVICE_COMPILE__)
#define MPC_LAUNDER(x) asm volatile("" : "+v"(x))
// The slice loops are fully unrolled (static register indices); without a fence every few columns the
// scheduler hoists all 60 LDS/HBM loads above the FMAs and the live set overflows the register file.
#define MPC_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// A 64-bit LDS store that the load/store vectoriser leaves alone (volatile, address space 3 spelled out so that it stays a ds_write)
#define MPC_LDS_STORE64(p, v) (*(volatile __attribute__((address_space(3))) double *)(p) = (v))
// A 64-bit LDS load that is not paired into ds_read2_b64: the pairs' 8-bit offsets (2 KB reach) force one base register per
// pair for the 20 slots of a partial-sum row, whereas single loads take 16-bit immediates off one base.
#define MPC_LDS_LOAD64(p) (*(const volatile __attribute__((address_space(3))) double *)(p))
#else
#define MPC_LDS_STORE64(p, v) (*(p) = (v))
#define MPC_LDS_LOAD64(p) (*(p))
#define MPC_LAUNDER(x) ((void)0)
#define MPC_SCHED_FENCE() ((void)0)
#endif
// Tuning hooks for tools/build_variant.sh experiments (the product build uses the defaults; see Cfg for what they mean):
//   MPC_NT_H16 / MPC_NT_H20   tiles per thread at h = 16 / 20          MPC_EXIT_FENCE_UPTO  largest h with the loop-exit fence
//   MPC_PIN_MASK              live-range split points (all horizons)   MPC_COLUMN_STORE64   0 / 1 for all horizons
//   MPC_PROW_SKEW             1: pivot rows 8 bytes off the 16-byte grid (ds_write2_b64 instead of ds_write_b128)
#ifndef MPC_EXIT_FENCE_UPTO
#define MPC_EXIT_FENCE_UPTO 16
#endif
#ifndef MPC_PROW_SKEW
#define MPC_PROW_SKEW 0
#endif
#ifndef MPC_NT_H20
#define MPC_NT_H20 4
#endif
#ifndef MPC_NT_H16
#define MPC_NT_H16 1
#endif
#define MPC_CHUNK 10   // columns between scheduling fences
// A value that is the same in every lane, moved to a scalar register (so that branches on it are scalar branches)
#if defined(__HIP_DEVICE_COMPILE__)
#define MPC_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane(x)
#else
#define MPC_UNIFORM_INT(x) (x)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MPC_CLOCK() ((long long)__builtin_readcyclecounter())
#else
#define MPC_CLOCK() (0LL)
#endif

namespace mpc {

// ---- OSQP constants (extern/osqp/include/constants.h:59-88) and the reference's settings -------
constexpr double kRho0 = 0.1, kSigma = 1e-6, kAlphaRelax = 1.6;
constexpr double kEpsAbs = 1e-3, kEpsRel = 1e-3;           // mpc_osqp.cc:711-712
constexpr int kMaxIter = 4000, kCheck = 25;                // CHECK_TERMINATION; adaptive_rho_interval (mpc_osqp.cc:710)
constexpr double kRhoMin = 1e-6, kRhoMax = 1e6, kRhoEqOverIneq = 1e3, kRhoTol = 1e-4;
constexpr int kScalingIters = 10;
constexpr double kMinScaling = 1e-4, kMaxScaling = 1e4, kAdaptTol = 5.0;
constexpr double kInfty = 1e30, kDelta = 1e-6;
constexpr int kPolishRefine = 3;
constexpr double kGravity = 9.8, kMaxScale = 10.0, kMinScale = 0.1;  // mpc_osqp.cc:54-56

// OSQP status values (constants.h:17-31)
constexpr int kStSolved = 1, kStSolvedInaccurate = 2, kStMaxIter = -2, kStNonCvx = -7, kStUnsolved = -10;

template <int H>
struct Cfg {
  static constexpr int N = 12 * H, M = 20 * H, NF = 4 * H;
  static constexpr int TS = 6;                           // register tile side (2 feet)
  static constexpr int G = N / TS;                       // tile grid G x G, lower triangle stored
  static constexpr int MT = G * (G + 1) / 2;             // lower-triangle tiles
  // Tiles per thread.  Up to h = 16 every thread holds one tile.  The longest horizon has 820 tiles: with one or two per
  // thread the workgroup runs 2-4 waves per SIMD, i.e. at most 256 / 128 registers per lane, and the tile spills to scratch
  // in every hot loop (measured: spill traffic, not arithmetic, bounded the kernel).  Four tiles per thread make it a
  // 256-thread workgroup, one wave per SIMD, with the full 512-register budget (256 VGPRs + 256 AGPRs as spill space).
  static constexpr int NT = H > 16 ? MPC_NT_H20 : (H > 12 ? MPC_NT_H16 : 1);
  static constexpr int MTH = (MT + NT - 1) / NT;         // threads that hold tiles
  static constexpr int TE = TS * TS;                     // tile elements per thread
  static constexpr int PG_LEN = MT * TE;                 // doubles of P_s scratch per robot (tile-major)
  static constexpr int T = (((MTH > N ? MTH : N) + 63) / 64) * 64;   // solve-kernel workgroup: a thread per tile slot and per variable
  static constexpr int MR = (M + T - 1) / T;             // constraint rows per thread (Solver::for_rows): 1, or 2 at h = 20
  static constexpr int TA = ((M > 256 ? M : 256) + 63) / 64 * 64;   // assembly-kernel workgroup (>= M threads)
  static constexpr int IN_LEN = 56 + 4 * H;
  static constexpr int NTASK2 = 21 + (H - 1) * 36;       // P assembly tasks (d, 2 x 2 block of (a, b))
  static_assert(T <= 1024, "workgroup too large");
  static constexpr int NP = N + 2;                       // row stride of part[] (doubles)
  static constexpr int MEVEN = (M + 1) & ~1;
  static constexpr int PARTLEN = (NP * G > 14 * 64 + 2 * MEVEN) ? NP * G : 14 * 64 + 2 * MEVEN;   // part[] doubles as the reduction scratch
                                                         // [0, 14 * 64) and, above it, holds z_pol / y_pol at the end of polish
  // LDS diet of the short horizon (three robots per CU need <= 54.6 KB each): q stays in the HBM record and rho per row
  // is a three-way select on the row type.  h = 16 keeps both in LDS -- it runs one robot per CU whatever its LDS size,
  // and the leaner forms cost it 9 % each (measured; they lengthen live ranges in a kernel that is at its register cap).
  // h = 20 needs the per-type rho to fit 160 KB.
  static constexpr bool kQInLds = H > 12;
  static constexpr bool kRhoPerType = H <= 12 || H > 16;
  static constexpr bool kLoopExitFence = H <= MPC_EXIT_FENCE_UPTO;   // scheduling fence after the ADMM loop (see Solver::run)
  // publish() stores a tile column as six 64-bit LDS stores instead of three 128-bit ones (no v_mov packing: -9 % VALU
  // instructions per sweep step; +2..3 %).
#ifdef MPC_COLUMN_STORE64
  static constexpr bool kColumnStore64 = MPC_COLUMN_STORE64;
#else
  static constexpr bool kColumnStore64 = true;
#endif
  // Live-range split points of the tile registers (Solver::pin_tiles): bit 0 / 1 before / after a sweep, 2 / 3 around the
  // 25 ADMM iterations, 4 inside the sweep loop, 5 / 6 before / after the Ruiz passes, 7 inside them.  They mattered by
  // factors while the kernel carried ~20 hoisted LDS base registers (see MPC_LDS_LOAD64); since those are gone h = 10 / 16 are
  // within 2 % for every mask tried, and h = 20 still prefers 3 (246 k steps/s against 220-235 k).
#ifdef MPC_PIN_MASK
  static constexpr int kPinMask = MPC_PIN_MASK;
#else
  static constexpr int kPinMask = H > 16 ? 3 : 17;
#endif
  // The QP record the assembly kernel hands to the solve kernel (doubles per robot): q[N] l[M] u[M] cone[15] pad
  static constexpr int QP_Q = 0, QP_L = N, QP_U = N + M, QP_CONE = N + 2 * M, QP_LEN = N + 2 * M + 16;
};

// Flat input record offsets (include/mpc_batch.h, layout.py)
constexpr int IN_W = 0, IN_POS = 13, IN_VEL = 16, IN_RPY = 19, IN_NRM = 22, IN_ANG = 25, IN_CONTACT = 28;
template <int H> constexpr int in_foot() { return 28 + 4 * H; }
template <int H> constexpr int in_fric() { return 40 + 4 * H; }
template <int H> constexpr int in_dpos() { return 44 + 4 * H; }
template <int H> constexpr int in_dvel() { return 47 + 4 * H; }
template <int H> constexpr int in_drpy() { return 50 + 4 * H; }
template <int H> constexpr int in_dang() { return 53 + 4 * H; }

// Per-robot persistent solver state in HBM (one contiguous record of doubles per robot):
//   x[N] z[M] y[M] q_old[N] rho flags      flags: 0 = cold (next call is the "osqp_setup" call)
template <int H> constexpr int state_len() { return 2 * Cfg<H>::N + 2 * Cfg<H>::M + 2; }

// Per-robot info record (ints): iter, status, status_polish, rho_updates, n_factor, first_run, 0, 0
constexpr int kInfoLen = 8;
// Per-robot profile record (shader cycles), 16 sections: 0 load 1 dynamics 2 q+P 3 scale-load 4 scale-loop 5 scale-store
// 6 K-form 7 sweep 8 admm 9 resid-mulP 10 resid-rest+check 11 polish-setup 12 polish-H 13 polish-refine 14 polish-finish 15 total
constexpr int kProfLen = 16;

struct RobotModel {       // constructor arguments of ConvexMpc (mpc_osqp.cc:508-527)
  double mass, inv_mass, inv_inertia[9], dt, alpha;
};

// Every vector in LDS starts on a 16-byte boundary so that runs of doubles can move as ds_read_b128 /
// ds_write_b128 with an immediate address (no per-access address register).
#define MPC_V alignas(16) double
template <int H>
struct Shared {
  using C = Cfg<H>;
  // ---- alive for the whole solve -------------------------------------------------------------
  // (LDS is what limits the robots per CU at h = 10 -- three fit in 160 KB below 54.6 KB each -- so nothing is stored that
  //  is cheap to re-derive: the unscaled q stays in the HBM record, 1/D and 1/E are divided out where the residuals need
  //  them (every 25 iterations), rho per row is a three-way select on the row type.)
  MPC_V q[C::kQInLds ? C::N : 2];                       // unscaled q (becomes q_old of the next call), unless it is re-read from HBM
  MPC_V qs[C::N]; MPC_V ls[C::M]; MPC_V us[C::M]; MPC_V As[C::NF * 15];   // scaled problem
  MPC_V D[C::N]; MPC_V E[C::M];
  double c, cinv, rho, ctmp;
  MPC_V rho_vec[C::kRhoPerType ? 2 : C::M]; MPC_V rho_inv[C::kRhoPerType ? 2 : C::M];   // per row, or unused: rho3 / rinv3
  double rho3[4], rinv3[4];                             // rho and 1/rho of a loose / inequality / equality row (index type + 1)
  signed char ctype[C::M];                              // -1 loose, 0 inequality, 1 equality (auxil.c:79-96)
  MPC_V x[C::N]; MPC_V xt[C::N]; MPC_V Px[C::N];        // Px = P_s x, carried through the ADMM iterations
  MPC_V zz[1][C::M]; MPC_V yy[1][C::M]; MPC_V rr[1][C::N];   // z, y, rhs
  // sweep pivot row (double buffered).  With MPC_PROW_SKEW = 1 the row starts 8 bytes off the 16-byte grid, so that publish()
  // stores it as ds_write2_b64 (two independent 64-bit sources) instead of ds_write_b128, whose 128-bit source ties pairs of
  // tile elements into register quads.
  MPC_V prow_raw[2][C::N + 2];
  MPC_HD double *prow(int b) { return prow_raw[b] + MPC_PROW_SKEW; }
  MPC_V piv[2][2];                                      // current pivot and its reciprocal (double buffered)
  unsigned long long red[16];                           // max-reductions (bit pattern of doubles >= 0)
  int first, iter, status, status_polish, rho_updates, nfact, done, bad;   // control (uniform)
  double pri_res, dua_res, rho_new;
  // ---- phase-local storage: scaling, then polish share the same LDS ---------------------------------------------
  union {
    struct {
      MPC_V cone[16];
      MPC_V l[C::M]; MPC_V u[C::M];                     // unscaled bounds
      MPC_V dt_[C::N]; MPC_V et_[C::M]; MPC_V cn_[C::N];   // Ruiz pass temporaries
    };
    struct {
      signed char act[C::M];
      MPC_V Nb[C::NF * 9]; MPC_V Gm[C::NF * 9];         // per foot: null basis rows (3 x 3, zero padded), Gamma
      int nnull[C::NF], isnull[C::N], rowmask[C::G];      // rowmask: isnull of a tile row's 6 coordinates, one bit each
      MPC_V u0[C::N]; MPC_V Pu[C::N]; MPC_V g[C::N]; MPC_V xN[C::N]; MPC_V PxN[C::N];
      // (the other polish vectors reuse storage that is dead by then: Solver::wv / rw / zpol / ypol)
    };
  };
  // (last: the big arrays sit above the statically addressable 64 KB, the small hot ones below)
  union {
    MPC_V part[C::PARTLEN];                             // [slot][row] partial sums / maxima of the tile products
    struct { MPC_V tm[C::M]; MPC_V rzt[C::M]; };        // R z - y and R z~ of the current ADMM iteration (part is dead then)
  };
};
// LDS of the assembly kernel (one workgroup per robot, its own launch: see Assembler)
template <int H>
struct AsmShared {
  using C = Cfg<H>;
  MPC_V in[C::IN_LEN];
  MPC_V x0[13]; MPC_V xref[13 * H]; MPC_V sdiff[13 * H]; MPC_V xk[13 * H];
  MPC_V a_dt[169]; MPC_V b_dt[156]; MPC_V a_exp[169]; MPC_V b_exp[156];
  MPC_V anb[H * 156]; MPC_V wanb[H * 156];              // A^k B and diag(w) A^k B
};
#undef MPC_V

template <int H>
struct Thread {
  using C = Cfg<H>;
  int tid;                        // thread id
  int ti[C::NT], tj[C::NT];       // my tiles: tile row / tile column (tj <= ti); tile u has index tid + u * MTH
  bool mact[C::NT], dia[C::NT];   // tile u exists; it sits on the diagonal
  double Mx[C::NT * C::TE];       // my tiles of the current symmetric n x n matrix (P_s, K, -Kinv, H, -Hinv), row-major 6 x 6 each
  MPC_HD void init(int id) {
    tid = id;
    for (int u = 0; u < C::NT; ++u) {
      const int tile = id + u * C::MTH;
      mact[u] = id < C::MTH && tile < C::MT;
      int r = 0;
      while ((r + 1) * (r + 2) / 2 <= tile) ++r;
      ti[u] = r; tj[u] = tile - r * (r + 1) / 2; dia[u] = ti[u] == tj[u];
    }
  }
};

// One tile of a thread, as the tile routines see it (built by Solver::for_tiles with a static tile slot, so Mx stays a
// statically indexed register array)
struct TileView {
  int ti, tj, index;
  bool dia;
  double *Mx;
};

MPC_HD double limit_scaling(double v) {  // scaling.c:7-14
  v = v < kMinScaling ? 1.0 : v;
  return v > kMaxScaling ? kMaxScaling : v;
}
MPC_HD double dmax(double a, double b) { return a > b ? a : b; }
MPC_HD double dmin(double a, double b) { return a < b ? a : b; }
// max / min as the bare instructions.  fmax() / fmin() first canonicalise every operand the compiler cannot prove to be a quiet
// number -- a `v_max_f64 x, x` for each value that comes from memory: 126 of the 309 v_max of a Ruiz pass, two per row in the
// z-update.  Nothing on this path produces a signalling NaN (values are results of arithmetic or converted floats), and for
// quiet NaNs the instruction returns the other operand exactly as fmax / fmin do.
MPC_HD double raw_max(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return fmax(a, b);
#endif
}
MPC_HD double raw_min(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return fmin(a, b);
#endif
}
// (equal to the c_max / c_min selects of OSQP whenever lo and hi are not NaN)
MPC_HD double clampd(double v, double lo, double hi) { return raw_min(raw_max(v, lo), hi); }

MPC_HD unsigned long long dbits(double v) {
  union { double d; unsigned long long u; } c;
  c.d = v;
  return c.u;
}
MPC_HD double bitsd(unsigned long long u) {
  union { double d; unsigned long long u; } c;
  c.u = u;
  return c.d;
}

// ------------------------------------------------------------------------------------------------
// The solver.  `Exec` provides: par(f), amax(&slot, value) (LDS atomic max on a double >= 0).
// `Pg` is this robot's n*n fp64 scratch in HBM (holds P, then the scaled P_s).
// ------------------------------------------------------------------------------------------------
// Diagnostic builds (-DMPC_PROFILE_SUB=<section>) split one section into slots 9..13 of the profile record:
// 1 = dynamics, 2 = one scaling pass, 3 = polish set-up, 4 = A dt / B dt set-up, 5 = the four phases of an ADMM iteration.
#ifndef MPC_PROFILE_SUB
#define MPC_PROFILE_SUB 0
#endif
#define MPC_SUBLAP(sec, k) do { if (MPC_PROFILE_SUB == (sec)) lap(k); } while (0)

// ============================================================================================================
// 1. Assembly (mpc_osqp.cc:606-688), a kernel of its own: one workgroup per robot builds q, the bounds, the cone block
// and P (tile-major, unscaled) and leaves them in HBM for the solve kernel.  It needs 38 KB of LDS that the solver does
// not (four robots per CU instead of two) and is bound by LDS latency / bandwidth, not by fp64 issue.
// ============================================================================================================
template <int H, class Exec>
struct Assembler {
  using C = Cfg<H>;
  using Th = Thread<H>;
  static constexpr int N = C::N, M = C::M, NF = C::NF, T = C::TA, TS = C::TS, TE = C::TE;

  Exec &ex;
  AsmShared<H> &s;
  const RobotModel &mdl;
  const float *in;     // [IN_LEN]
  double *Pg;          // [PG_LEN]   out: P, lower-triangle tiles
  double *qp;          // [QP_LEN]   out: q, l, u, cone
  long long *prof;     // [kProfLen] slots 1 (dynamics) and 2 (q + P) are written here (may be null)
  long long tc[3] = {0, 0, 0};
  long long tlast = 0;
#ifndef MPC_SECTION_PROFILE   // per-section counters cost ~30 SGPRs (and push uniform values into VGPRs): opt-in, tools/section_profile.py
  MPC_HD void lap(int) {}
#else
  MPC_HD void lap(int k) { const long long now = MPC_CLOCK(); tc[k] += now - tlast; tlast = now; }
#endif

  static MPC_HD double mat3e(const double *a, const double *b, int e) {   // entry e = 3 i + j of the 3 x 3 product a b
    const int i = e / 3, j = e - 3 * i;
    return a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  }
  static MPC_HD void mat3(const double *a, const double *b, double *c) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  }

  static MPC_HD size_t pg_index(int r, int c) {   // offset of entry (r, c) in the tile-major store; needs r / 6 >= c / 6
    const int I = r / TS, J = c / TS;
    return (size_t)(I * (I + 1) / 2 + J) * TE + (r - TS * I) * TS + (c - TS * J);
  }

  // ================================ 1. assembly =================================================
  MPC_HD void run() {
    tlast = MPC_CLOCK();
    ex.par([&](Th &t) {
      for (int i = t.tid; i < C::IN_LEN; i += T) s.in[i] = (double)in[i];
      for (int i = t.tid; i < 169; i += T) s.a_dt[i] = 0;
      for (int i = t.tid; i < 156; i += T) s.b_dt[i] = 0;
    });
    // ---- A dt, B dt (mpc_osqp.cc:299-336, 606-617, 661-673): every 3 x 3 product is one entry per thread, all
    // operands in LDS (a thread-local array indexed by a runtime entry number would live in scratch memory).
    // The intermediates use the not yet used xk / sdiff areas:
    //   xk:    tan(pitch); m1 = Rx Ry; m2 = Rz Ry; rxyz; rzyx; fw[12]; t2 = rzyx I^-1; iw
    //   sdiff: [26..53) Rx, Ry, Rz; [53..62) I^-1 (body)
    double *const tp_ = s.xk, *const m1 = s.xk + 7, *const m2 = s.xk + 16, *const rxyz = s.xk + 25, *const rzyx = s.xk + 34,
                 *const fw = s.xk + 43, *const t2 = s.xk + 55, *const iw = s.xk + 64;
    double *const rxm = s.sdiff + 26, *const rym = s.sdiff + 35, *const rzm = s.sdiff + 44, *const iib = s.sdiff + 53;
    static_assert(13 * H >= 73, "xk / sdiff too small for the set-up scratch");
    ex.par([&](Th &t) {
      if (t.tid < 27) {   // entry k of rotation `which` about x / y / z: 0, 1, cos, sin or -sin of its angle
        const int which = t.tid / 9, k = t.tid - 9 * which;
        const int ax = which, u = (ax + 1) % 3, v = (ax + 2) % 3, r = k / 3, c = k - 3 * r;
        const double ang = s.in[IN_RPY + which];
        double val;
        if (r == ax || c == ax) val = (r == c) ? 1.0 : 0.0;
        else if (r == c) val = cos(ang);
        else val = (r == v && c == u) ? sin(ang) : -sin(ang);     // R[u][v] = -sin, R[v][u] = +sin
        rxm[t.tid] = val;
      } else if (t.tid == 64) {
        tp_[0] = tan(s.in[IN_RPY + 1]);
      } else if (t.tid >= 96 && t.tid < 105) {
        iib[t.tid - 96] = mdl.inv_inertia[t.tid - 96];
      }
    });
    MPC_SUBLAP(4, 9);
    ex.par([&](Th &t) {   // m1 = Rx Ry (feet, :606-609), m2 = Rz Ry (inertia, :283-291)
      if (t.tid < 18) {
        const int e = t.tid % 9;
        (t.tid < 9 ? m1 : m2)[e] = mat3e(t.tid < 9 ? rxm : rzm, rym, e);
      }
      // x0 (:630-633)
      if (t.tid >= 32 && t.tid < 45) {
        const int i = t.tid - 32;
        s.x0[i] = i < 3 ? s.in[IN_RPY + i] : i < 6 ? s.in[IN_POS + i - 3] : i < 9 ? s.in[IN_ANG + i - 6] : i < 12 ? s.in[IN_VEL + i - 9] : -kGravity;
      }
      // bounds (:449-477, 685-688, 720-721)
      if (t.tid < M) {
        const int i = t.tid, f = i / 5, r = i - 5 * f;
        const double cst = s.in[IN_CONTACT + f];
        const double fzmax = mdl.mass * kGravity * kMaxScale, fzmin = mdl.mass * kGravity * kMinScale;
        const double mu0 = s.in[in_fric<H>()];
        qp[C::QP_L + i] = dmax(r < 4 ? 0.0 : fzmin * cst, -kInfty);
        qp[C::QP_U + i] = dmin(r < 4 ? (mu0 + 1) * fzmax * cst : fzmax * cst, kInfty);
      }
    });
    MPC_SUBLAP(4, 10);
    ex.par([&](Th &t) {   // rxyz = (Rx Ry) Rz, rzyx = (Rz Ry) Rx
      if (t.tid < 18) {
        const int e = t.tid % 9;
        (t.tid < 9 ? rxyz : rzyx)[e] = mat3e(t.tid < 9 ? m1 : m2, t.tid < 9 ? rzm : rxm, e);
      }
      // x_ref (:635-659): row r of step i is base_r + dt (i + 1) slope_r  (slope 0 for the constant rows)
      for (int k = t.tid; k < 13 * H; k += T) {
        const int i = k / 13, r = k - 13 * i;
        const double tt = mdl.dt * (i + 1);
        const int drpy = in_drpy<H>(), dvel = in_dvel<H>(), dang = in_dang<H>(), dpos = in_dpos<H>();
        const int bi = r < 2 ? drpy + r : r == 2 ? IN_RPY + 2 : r < 5 ? IN_POS + r - 3 : r == 5 ? dpos + 2 : r < 9 ? dang + r - 6 : dvel + (r < 11 ? r - 9 : 0);
        const int si = r == 2 ? dang + 2 : dvel + (r == 4 ? 1 : 0);
        const double base = s.in[bi], slope = s.in[si];
        const double v = (r == 2 || r == 3 || r == 4) ? tt * slope + base : base;
        s.xref[k] = r == 11 ? 0.0 : r == 12 ? -kGravity : v;
      }
    });
    MPC_SUBLAP(4, 11);
    ex.par([&](Th &t) {   // feet in the world frame; t2 = rzyx I^-1 (:670)
      if (t.tid < 12) {
        const int i = t.tid / 3, r = t.tid - 3 * i;
        const double *fb = s.in + in_foot<H>();
        fw[t.tid] = rxyz[3 * r] * fb[3 * i] + rxyz[3 * r + 1] * fb[3 * i + 1] + rxyz[3 * r + 2] * fb[3 * i + 2];
      } else if (t.tid < 21) {
        t2[t.tid - 12] = mat3e(rzyx, iib, t.tid - 12);
      }
    });
    MPC_SUBLAP(4, 12);
    ex.par([&](Th &t) {   // iw = t2 rzyx^T (:671)
      if (t.tid < 9) {
        const int i = t.tid / 3, j = t.tid - 3 * i;
        iw[t.tid] = t2[3 * i] * rzyx[3 * j] + t2[3 * i + 1] * rzyx[3 * j + 1] + t2[3 * i + 2] * rzyx[3 * j + 2];
      }
    });
    ex.par([&](Th &t) {
      const double dt = mdl.dt;
      if (t.tid < 36) {   // B rows 6-8: I_w^-1 [r_i]x (:324-336); [v]x = {0, -v2, v1; v2, 0, -v0; -v1, v0, 0}
        const int i = t.tid / 9, e = t.tid - 9 * i, r = e / 3, c = e - 3 * r;
        const double *v = fw + 3 * i;
        double acc = 0;
        for (int k = 0; k < 3; ++k) {   // same left-to-right sum as the 3 x 3 product, with skew[k][c] formed on the fly
          const double sk = (k == c) ? 0.0 : (((c - k + 3) % 3 == 1) ? -v[3 - k - c] : v[3 - k - c]);
          const double term = iw[3 * r + k] * sk;
          acc = k == 0 ? term : acc + term;
        }
        s.b_dt[(6 + r) * 12 + 3 * i + c] = acc * dt;
      } else if (t.tid < 48) {   // B rows 9-11: I / m
        const int k = t.tid - 36, i = k / 3, r = k - 3 * i;
        s.b_dt[(9 + r) * 12 + 3 * i + r] = mdl.inv_mass * dt;
      } else if (t.tid < 57) {   // A rows 0-2: omega -> rpy rates (:311-312): {cy/cp, sy/cp, 0; -sy, cy, 0; cy tp, sy tp, 1}
        const int e = t.tid - 48, r = e / 3, c = e - 3 * r;
        const double cp = rym[0], cy = rzm[0], sy = rzm[3], tp = tp_[0];
        const double num = c == 0 ? cy : sy;
        double val;
        if (c == 2) val = r == 2 ? 1.0 : 0.0;
        else if (r == 0) val = num / cp;
        else if (r == 1) val = c == 0 ? -sy : cy;
        else val = num * tp;
        s.a_dt[r * 13 + 6 + c] = val * dt;
      } else if (t.tid < 60) {
        const int r = t.tid - 57;
        s.a_dt[(3 + r) * 13 + 9 + r] = dt;
        s.a_dt[(9 + r) * 13 + 12] = s.in[IN_NRM + r] * dt;
      } else if (t.tid == 60) {
        const double *fr = s.in + in_fric<H>();
        const double cb[15] = {-1, 0, fr[0], 1, 0, fr[1], 0, -1, fr[2], 0, 1, fr[3], 0, 0, 1};   // :437-447
        for (int k = 0; k < 15; ++k) qp[C::QP_CONE + k] = cb[k];
      }
    });
    MPC_SUBLAP(1, 9);
    MPC_SUBLAP(4, 13);
    // exact exponential (mpc_osqp.cc:338-351; M^3 = 0): A_exp = I + A dt + (A dt)^2/2, B_exp = B dt + (A dt)(B dt)/2.
    // A dt is nonzero only at rows 0-2 x cols 6-8, (3+i, 9+i) and rows 9-11 x col 12; the dense products of the
    // reference add exact zeros elsewhere, so only the nonzero terms are formed (same order, same values).
    ex.par([&](Th &t) {
      for (int k = t.tid; k < 169 + 156; k += T) {
        if (k < 169) {
          const int r = k / 13, c = k - 13 * r;
          const double acc = (r >= 3 && r < 6 && c == 12) ? s.a_dt[r * 13 + r + 6] * s.a_dt[(r + 6) * 13 + 12] : 0.0;
          s.a_exp[k] = (r == c ? 1.0 : 0.0) + s.a_dt[k] + acc / 2;
        } else {
          const int kk = k - 169, r = kk / 12, c = kk - 12 * r;
          double acc = 0;
          if (r < 3) { for (int j = 6; j < 9; ++j) acc += s.a_dt[r * 13 + j] * s.b_dt[j * 12 + c]; }
          else if (r < 6) acc += s.a_dt[r * 13 + r + 6] * s.b_dt[(r + 6) * 12 + c];
          s.b_exp[kk] = s.b_dt[kk] + acc / 2;
        }
      }
    });
    MPC_SUBLAP(1, 10);
    // A^k B (:368-373) and the free response A^{i+1} x0, i < H-1 (:360-364; the last A_qp block stays 0).
    // A dt is nilpotent, so A_exp^k = exp(k A dt) = I + k A dt + k^2 (A dt)^2 / 2 exactly, and (A dt)^2 B_exp = 0
    // (its only column, 12, meets the zero row 12 of B_exp):  A_exp^k B_exp = B_exp + k U,  U = (A dt) B_exp,
    // which is nonzero in rows 0-5 only.  All k are formed at once (the reference multiplies k times; the two
    // agree to rounding).  U overwrites b_dt, (A dt) x0 and (A dt)^2 x0 go to the first 26 slots of sdiff.
    ex.par([&](Th &t) {
      if (t.tid < 72) {
        const int r = t.tid / 12, c = t.tid - 12 * r;
        double acc = 0;
        if (r < 3) { for (int j = 6; j < 9; ++j) acc += s.a_dt[r * 13 + j] * s.b_exp[j * 12 + c]; }
        else acc = s.a_dt[r * 13 + r + 6] * s.b_exp[(r + 6) * 12 + c];
        s.b_dt[t.tid] = acc;
      } else if (t.tid < 72 + 13) {
        const int r = t.tid - 72;
        double a1 = 0, a2 = 0;
        if (r < 3) { for (int j = 6; j < 9; ++j) a1 += s.a_dt[r * 13 + j] * s.x0[j]; }
        else if (r < 6) { a1 = s.a_dt[r * 13 + r + 6] * s.x0[r + 6]; a2 = (s.a_dt[r * 13 + r + 6] * s.a_dt[(r + 6) * 13 + 12]) * s.x0[12]; }
        else if (r >= 9 && r < 12) a1 = s.a_dt[r * 13 + 12] * s.x0[12];
        s.sdiff[r] = a1; s.sdiff[13 + r] = a2;
      }
    });
    ex.par([&](Th &t) {
      for (int e = t.tid; e < H * 156; e += T) {
        const int k = e / 156, rc = e - 156 * k, r = rc / 12;
        const double v = r < 6 ? s.b_exp[rc] + (double)k * s.b_dt[rc] : s.b_exp[rc];
        s.anb[e] = v;
        s.wanb[e] = s.in[IN_W + r] * v;
      }
      for (int e = t.tid; e < 13 * (H - 1); e += T) {   // (this overwrites the set-up scratch, which is dead by now)
        const int i = e / 13, r = e - 13 * i;
        const double kk = i + 1;
        s.xk[e] = s.x0[r] + kk * s.sdiff[r] + (kk * kk / 2) * s.sdiff[13 + r];
      }
    });
    MPC_SUBLAP(1, 11);
    ex.par([&](Th &t) {   // state_diff (:681)
      for (int k = t.tid; k < 13 * H; k += T) s.sdiff[k] = (k < 13 * (H - 1) ? s.xk[k] : 0.0) - s.xref[k];
    });
    lap(1);
    // q (:683) and P (:387-434) -> Pg (unscaled, lower-triangle tiles)
    ex.par([&](Th &t) {
      if (t.tid < N) {
        const int j = t.tid / 12, c = t.tid - 12 * j;
        double acc = 0;
        for (int i = j; i < H; ++i) {
          const double *bk = s.wanb + (i - j) * 156 + c, *sd = s.sdiff + 13 * i;
          double e = 0, o = 0;   // two FMA chains per horizon step
#pragma unroll
          for (int r = 0; r < 13; ++r) {
            if (r & 1) o += bk[r * 12] * sd[r];
            else e += bk[r * 12] * sd[r];
          }
          acc += e + o;
        }
        qp[C::QP_Q + t.tid] = 2 * acc;
      }
      // P: 2 x 2 register blocks of (a, b) -- four outputs share two 16-byte loads per term (one LDS byte per flop
      // instead of two: with four robots per CU this phase is LDS-bandwidth bound).  Tasks: for d = 0 the block pairs
      // a2 <= b2 (21), for d >= 1 all 36; d-major, so a thread's second task is a short one.
      for (int task = t.tid; task < C::NTASK2; task += T) {
        int d, a2, b2;
        if (task < 21) {
          d = 0;
          int k = task; a2 = 0;
          while (k >= 6 - a2) { k -= 6 - a2; ++a2; }
          b2 = a2 + k;
        } else {
          const int k = task - 21;
          d = 1 + k / 36;
          const int ab = k - (d - 1) * 36;
          a2 = ab / 6; b2 = ab - 6 * a2;
        }
        const int a0 = 2 * a2, b0 = 2 * b2, ah = a0 / TS, bh = b0 / TS, ar = a0 - TS * ah, br = b0 - TS * bh;
        // entry (12 I + a, 12 J + b), I = J - d <= J, lives at row 12 J + b, column 12 I + a of the lower triangle:
        // tile (2 J + b / 6, 2 I + a / 6), position (b % 6, a % 6); a diagonal tile also takes the mirrored entry
        const bool dtile = d == 0 && ah == bh;
        const double *xa = s.wanb + d * 156 + a0, *yb = s.anb + b0;
        double acc[4] = {0, 0, 0, 0};
        const int ns = H - d;
        for (int sidx = 0; sidx < ns; ++sidx) {
          const double *x = xa + sidx * 156, *y = yb + sidx * 156;
          double e[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};   // even / odd terms: eight independent FMA chains
#pragma unroll
          for (int r = 0; r < 13; ++r) {
            const double x0 = x[r * 12], x1 = x[r * 12 + 1], y0 = y[r * 12], y1 = y[r * 12 + 1];
            double *w = (r & 1) ? o : e;
            w[0] += x0 * y0; w[1] += x1 * y0; w[2] += x0 * y1; w[3] += x1 * y1;      // index ia + 2 ib
          }
          const int J = H - 1 - sidx, I = J - d;
          const int tr = 2 * J + bh;
          double *tile = Pg + (size_t)(tr * (tr + 1) / 2 + 2 * I + ah) * TE;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ia = q & 1, ib = q >> 1;
            acc[q] += e[q] + o[q];
            if (d == 0 && a0 + ia > b0 + ib) continue;       // below the diagonal of a diagonal block: its mirror is computed
            double v = 2.0 * acc[q];
            if (d == 0 && a0 + ia == b0 + ib) v += mdl.alpha;
            tile[(br + ib) * TS + ar + ia] = v;
            if (dtile && a0 + ia != b0 + ib) tile[(ar + ia) * TS + br + ib] = v;
          }
        }
      }
    });
    lap(2);
    if (prof) {
      ex.par([&](Th &t) { if (t.tid == 0) { prof[1] = tc[1]; prof[2] = tc[2]; } });
    }
  }
};

template <int H, class Exec>
struct Solver {
  using C = Cfg<H>;
  using Th = Thread<H>;
  using Sh = Shared<H>;
  static constexpr int N = C::N, M = C::M, NF = C::NF, T = C::T, TS = C::TS, G = C::G, TE = C::TE, NP = C::NP;

  Exec &ex;
  Sh &s;
  const RobotModel &mdl;
  double *state;       // [state_len<H>()]
  double *Pg;          // [PG_LEN]  P (unscaled) from the assembly kernel; P_s after scaling
  const double *qp;    // [QP_LEN]  q, l, u, cone from the assembly kernel
  double *forces;      // [N]   out: -D x (all horizon steps), untouched on failure
  int *info;           // [kInfoLen]
  long long *prof;     // [kProfLen] shader-clock cycles per section (may be null)
  int pp = 0;          // which half of the z / y / rhs ping-pong buffers is current (uniform)
  MPC_HD double *cz() { return s.zz[pp]; }
  MPC_HD double *cy() { return s.yy[pp]; }
  MPC_HD double *crhs() { return s.rr[pp]; }
  // polish vectors in storage that is dead during polish: the ADMM right-hand side, P_s x (re-derived by the next call), and
  // the upper part of part[] -- z_pol / y_pol live from after the last tile product to the acceptance test, and residuals()
  // in between uses part[0 .. 14 * 64) only.
  MPC_HD double *rw() { return s.rr[0]; }
  MPC_HD double *wv() { return s.Px; }
  MPC_HD double *zpol() { return s.part + 14 * 64; }
  MPC_HD double *ypol() { return s.part + 14 * 64 + C::MEVEN; }
  using Tv = TileView;
  template <class F>
  MPC_HD void for_tiles(Th &t, F &&f) {
#pragma unroll
    for (int u = 0; u < C::NT; ++u)
      if (t.mact[u]) {
        Tv v{t.ti[u], t.tj[u], t.tid + u * C::MTH, t.dia[u], t.Mx + u * TE};
        f(v, u);
      }
  }
  // The constraint rows of a thread: tid, tid + T, ... (one row per thread unless the workgroup is smaller than M).
  template <class F>
  MPC_HD void for_rows(const Th &t, F &&f) {
#pragma unroll
    for (int p = 0; p < C::MR; ++p) {
      const int i = t.tid + p * T;
      if (i < M) f(i);
    }
  }
  // A register-allocation hint, no code: every tile element passes through an empty asm, which ends its live range and
  // starts a new one.  The tile lives from load() to polish(); without such split points the allocator treats a
  // tile register pair as one range over the whole kernel and, once some phase is over budget, spills it in the hot
  // loops as well (h = 16: 12 of the 36 elements went through scratch on every sweep step with 70 VGPRs idle).
  MPC_HD void pin_tiles(int site) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!((C::kPinMask >> site) & 1)) return;
#pragma unroll
    for (int e = 0; e < C::NT * TE; ++e) MPC_LAUNDER(ex.th.Mx[e]);
#endif
  }
  MPC_HD double q_at(int i) const {
    if constexpr (C::kQInLds) return s.q[i];
    else return qp[C::QP_Q + i];
  }
  MPC_HD double rho_at(int i) const {
    if constexpr (C::kRhoPerType) {
      // (three uniform loads issued together with the row type; the empty asm keeps the compiler from sinking them into
      //  branches on the type, which would make them a second, dependent LDS round trip in every ADMM iteration)
      double r0 = s.rho3[0], r1 = s.rho3[1], r2 = s.rho3[2];
      const int ty = s.ctype[i];
      MPC_LAUNDER(r0); MPC_LAUNDER(r1); MPC_LAUNDER(r2);
      return ty == 1 ? r2 : (ty == 0 ? r1 : r0);
    } else return s.rho_vec[i];
  }
  MPC_HD double rinv_at(int i) const {
    if constexpr (C::kRhoPerType) {
      double r0 = s.rinv3[0], r1 = s.rinv3[1], r2 = s.rinv3[2];
      const int ty = s.ctype[i];
      MPC_LAUNDER(r0); MPC_LAUNDER(r1); MPC_LAUNDER(r2);
      return ty == 1 ? r2 : (ty == 0 ? r1 : r0);
    } else return s.rho_inv[i];
  }
  long long tc[kProfLen] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = 0;
#ifndef MPC_SECTION_PROFILE   // per-section counters cost ~30 SGPRs (and push uniform values into VGPRs): opt-in, tools/section_profile.py
  MPC_HD void lap(int) {}
#else
  MPC_HD void lap(int k) { const long long now = MPC_CLOCK(); tc[k] += now - tlast; tlast = now; }
#endif

  // ---- helpers valid inside a phase ------------------------------------------------------------
  static MPC_HD double a_row_dot(const Sh &s, int i, const double *v) {  // row i of scaled A times v
    const int f = i / 5, r = i - 5 * f;
    const double *a = s.As + 15 * f + 3 * r;
    return a[0] * v[3 * f] + a[1] * v[3 * f + 1] + a[2] * v[3 * f + 2];
  }
  static MPC_HD double at_col_dot(const Sh &s, int j, const double *v) {  // column j of scaled A times v
    const int f = j / 3, c = j - 3 * f;
    const double *a = s.As + 15 * f + c;
    double t = 0;
    for (int r = 0; r < 5; ++r) t += a[3 * r] * v[5 * f + r];
    return t;
  }
  // Partial results of the tile products live in part[slot * NP + row] (NP = N + 2: the pad spreads the six-double runs
  // that consecutive lanes store into different slots over the LDS banks): row i of tile row I gets slot J from the
  // tile (I, J) itself (J <= I) and slot J > I from the transpose of tile (J, I) -- G slots per row, each written
  // by exactly one thread, six consecutive doubles per thread and slot.
  // part <- partial products of Mx v, both orientations of the tile (Mx holds the NEGATED inverse: inv_combine flips the sign)
  MPC_HD void tile_matvec(const Tv &t, const double *v) {
    double vc[TS], vr[TS], ar[TS], ac[TS];
#pragma unroll
    for (int b = 0; b < TS; ++b) { vc[b] = v[TS * t.tj + b]; vr[b] = v[TS * t.ti + b]; ar[b] = 0; ac[b] = 0; }
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
      for (int b = 0; b < TS; ++b) {   // twelve independent accumulation chains
        const double m = t.Mx[a * TS + b];
        ar[a] += m * vc[b];
        ac[b] += m * vr[a];
      }
    double *pd = s.part + t.tj * NP + TS * t.ti, *pt = s.part + t.ti * NP + TS * t.tj;
#pragma unroll
    for (int a = 0; a < TS; ++a) pd[a] = ar[a];     // (un-negated: inv_combine subtracts the sum -- exact, and twelve v_xor fewer per tile)
    if (!t.dia) {
#pragma unroll
      for (int b = 0; b < TS; ++b) pt[b] = ac[b];
    }
  }
  template <bool MAX>
  static MPC_HD double fold_parts(const Sh &s, int row) {   // fixed pairwise order (short dependency chains)
    double v[G];
#pragma unroll
    for (int k = 0; k < G; ++k) v[k] = MPC_LDS_LOAD64(s.part + k * NP + row);   // (one base register + immediates: see the macro)
#pragma unroll
    for (int w = 1; w < G; w *= 2)
#pragma unroll
      for (int k = 0; k + w < G; k += 2 * w) v[k] = MAX ? raw_max(v[k], v[k + w]) : v[k] + v[k + w];
    return v[0];
  }
  static MPC_HD double sum_parts(const Sh &s, int row) { return fold_parts<false>(s, row); }
  // combine the partial products of (-Minv) v for a swept row: see sweep_all()
  static MPC_HD double inv_combine(const Sh &s, int row, const double *v) { return 2.0 * v[row] - sum_parts(s, row); }
  // (entries are norms: >= 0, never NaN since fmax drops NaNs)
  static MPC_HD double max_parts(const Sh &s, int row) { return fold_parts<true>(s, row); }
  // part <- D_i max_j (|m_ij| D_j) over the tile, for its rows and (transposed) for its columns; D = 1 if null
  MPC_HD void tile_rownorms(const Tv &t, const double *D) {
    double dc[TS], dr[TS], mr[TS], mc[TS];
#pragma unroll
    for (int b = 0; b < TS; ++b) { dc[b] = D ? D[TS * t.tj + b] : 1.0; dr[b] = D ? D[TS * t.ti + b] : 1.0; mr[b] = 0; mc[b] = 0; }
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
      for (int b = 0; b < TS; ++b) {
        const double m = fabs(t.Mx[a * TS + b]);
        mr[a] = fmax(mr[a], m * dc[b]);
        mc[b] = fmax(mc[b], m * dr[a]);
      }
    double *pd = s.part + t.tj * NP + TS * t.ti, *pt = s.part + t.ti * NP + TS * t.tj;
#pragma unroll
    for (int a = 0; a < TS; ++a) pd[a] = mr[a] * dr[a];
    if (!t.dia) {
#pragma unroll
      for (int b = 0; b < TS; ++b) pt[b] = mc[b] * dc[b];
    }
  }
  // P_s in HBM is tile-major: tile `index` is the 36 doubles at G[36 index]
  MPC_HD void load_tile(Tv &t, const double *Gm) {
    const double *g = Gm + (size_t)t.index * TE;
#pragma unroll
    for (int e = 0; e < TE; ++e) t.Mx[e] = g[e];
  }
  MPC_HD void store_tile(const Tv &t, double *Gm) {
    double *g = Gm + (size_t)t.index * TE;
#pragma unroll
    for (int e = 0; e < TE; ++e) g[e] = t.Mx[e];
  }

  // ================================ 1. load: the QP record of the assembly kernel + the warm-start state =====
  MPC_HD void load() {
    ex.par([&](Th &t) {
      for (int i = t.tid; i < N; i += T) { if constexpr (C::kQInLds) s.q[i] = qp[C::QP_Q + i]; s.x[i] = state[i]; s.xt[i] = state[N + 2 * M + i]; /* q_old */ }
      for (int i = t.tid; i < M; i += T) {
        s.l[i] = qp[C::QP_L + i]; s.u[i] = qp[C::QP_U + i];
        s.zz[0][i] = state[N + i]; s.yy[0][i] = state[N + M + i];   // scaled iterates of the previous call; zeros on the first call
      }
      if (t.tid < 15) s.cone[t.tid] = qp[C::QP_CONE + t.tid];
      if (t.tid == 0) {
        const bool first = state[2 * N + 2 * M + 1] == 0.0;
        s.first = first;
        s.rho = first ? kRho0 : state[2 * N + 2 * M];
        s.status = kStUnsolved; s.status_polish = 0; s.rho_updates = 0; s.nfact = 0; s.iter = 0; s.done = 0; s.bad = 0;
      }
    });
    lap(0);
  }
  // ================================ 2. scaling (scaling.c:44-156) ===============================
  // One Ruiz pass is three phases.  P itself stays UNSCALED in the tile registers for all passes: a pass only
  // needs the row norms of c D P D, which are c D_i max_j(|P_ij| D_j) with the cumulative D and c (tile_rownorms); D, c, q, A, E are updated incrementally as in
  // scaling.c, and c D P D is formed once after the last pass.  The cost scale c_temp of pass k is folded in
  // lazily at pass k + 1.
  template <bool MAX>
  MPC_HD double fold_half(const double *p) const {   // tree-reduce NF / 2 values whose loads are issued as one batch
    constexpr int L = NF / 2;
    double v[L];
#pragma unroll
    for (int f = 0; f < L; ++f) v[f] = p[f];
    MPC_SCHED_FENCE();
#pragma unroll
    for (int w = 1; w < L; w *= 2)
#pragma unroll
      for (int k = 0; k + w < L; k += 2 * w) v[k] = MAX ? raw_max(v[k], v[k + w]) : v[k] + v[k + w];
    MPC_SCHED_FENCE();
    return v[0];
  }
  MPC_HD double pending_cost_scale() const {   // scaling.c:108-139, from the per-foot partials in cn_
    const double mean = (fold_half<false>(s.cn_) + fold_half<false>(s.cn_ + NF / 2)) * (1.0 / N);
    const double nq = limit_scaling(fmax(fold_half<true>(s.cn_ + NF), fold_half<true>(s.cn_ + NF + NF / 2)));
    return fast_recip(limit_scaling(fmax(mean, nq)));
  }
  static MPC_HD double row_scale3(double a0, double a1, double a2) {   // 1 / sqrt(|row|_inf) of a 3-entry row of A
    return fast_rsqrt(limit_scaling(fmax(fmax(fabs(a0), fabs(a1)), fabs(a2))));
  }
  MPC_HD void scale() {
    lap(2);
    ex.par([&](Th &t) {
      for_tiles(t, [&](Tv &v, int) { load_tile(v, Pg); tile_rownorms(v, nullptr); });
      if (t.tid < N) {
        s.qs[t.tid] = s.first ? q_at(t.tid) : s.xt[t.tid];   // osqp_update_P_A equilibrates with the PREVIOUS q
        s.D[t.tid] = 1.0;
      }
      for_rows(t, [&](int i) {
        const double *a = s.cone + 3 * (i % 5);
        s.E[i] = 1.0;
        s.et_[i] = row_scale3(a[0], a[1], a[2]);
      });
      for (int k = t.tid; k < NF * 15; k += T) s.As[k] = s.cone[k % 15];
      if (t.tid == 0) s.c = 1.0;
    });
    lap(3);
    pin_tiles(5);
    for (int it = 0; it < kScalingIters; ++it) {
      pin_tiles(7);
      ex.par([&](Th &t) {   // column scales from |column|_inf of [c P ; A]; D <- D_temp D
        if (t.tid < N) {
          const int j = t.tid, f = j / 3, c = j - 3 * f;
          const double ct = it > 0 ? pending_cost_scale() : 1.0;
          double av[5];
#pragma unroll
          for (int r = 0; r < 5; ++r) av[r] = s.As[15 * f + 3 * r + c];
          double mx = (s.c * ct) * max_parts(s, j);
#pragma unroll
          for (int r = 0; r < 5; ++r) mx = fmax(mx, fabs(av[r]));
          const double d = fast_rsqrt(limit_scaling(mx));
          s.dt_[j] = d;
          s.D[j] *= d;
          if (j == 0) s.ctmp = ct;
        }
      });
      MPC_SUBLAP(2, 9);
      ex.par([&](Th &t) {   // A <- E A D, q <- D (c_temp q), c <- c_temp c; new row norms of D P D and A
        const double ct = s.ctmp;
        for_tiles(t, [&](Tv &v, int) { tile_rownorms(v, s.D); });
        for_rows(t, [&](int i) {
          const int f = i / 5;
          double *a = s.As + 3 * i;
          const double e = s.et_[i], a0 = a[0], a1 = a[1], a2 = a[2];
          const double d0 = s.dt_[3 * f], d1 = s.dt_[3 * f + 1], d2 = s.dt_[3 * f + 2], eo = s.E[i];
          const double n0 = (a0 * e) * d0, n1 = (a1 * e) * d1, n2 = (a2 * e) * d2;
          a[0] = n0; a[1] = n1; a[2] = n2;
          s.E[i] = eo * e;
          s.et_[i] = row_scale3(n0, n1, n2);   // row scale of the next pass
        });
        if (t.tid < N) s.qs[t.tid] = (s.qs[t.tid] * ct) * s.dt_[t.tid];
        if (t.tid == T - 1) s.c *= ct;
      });
      MPC_SUBLAP(2, 10);
      ex.par([&](Th &t) {   // cost scaling (scaling.c:108-139): per-foot partial sums of the column norms, |q|_inf
        if (t.tid < NF) {
          const int f = t.tid;
          const double q0 = s.qs[3 * f], q1 = s.qs[3 * f + 1], q2 = s.qs[3 * f + 2];
          s.cn_[f] = s.c * ((max_parts(s, 3 * f) + max_parts(s, 3 * f + 1)) + max_parts(s, 3 * f + 2));
          s.cn_[NF + f] = dmax(dmax(fabs(q0), fabs(q1)), fabs(q2));
        }
      });
      MPC_SUBLAP(2, 11);
    }
    ex.par([&](Th &t) {   // the last pass's cost scale; P_s = c D P D
      const double ct = pending_cost_scale(), cf = s.c * ct;
      for_tiles(t, [&](Tv &v, int) {
        double dc[TS], ra[TS];
#pragma unroll
        for (int b = 0; b < TS; ++b) { dc[b] = s.D[TS * v.tj + b]; ra[b] = s.D[TS * v.ti + b] * cf; }
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
          for (int b = 0; b < TS; ++b) v.Mx[a * TS + b] = (v.Mx[a * TS + b] * dc[b]) * ra[a];
      });
      if (t.tid < N) s.qs[t.tid] *= ct;
      if (t.tid == T - 1) s.ctmp = cf;   // (s.c is still being read in this phase)
    });
    pin_tiles(6);
    lap(4);
    ex.par([&](Th &t) {
      const double cf = s.ctmp;
      if (t.tid == 0) { s.c = cf; s.cinv = 1.0 / cf; }
      if (t.tid < N) {
        if (!s.first) s.qs[t.tid] = (s.D[t.tid] * q_at(t.tid)) * cf;      // osqp_update_lin_cost (osqp.c:765-770)
      }
      for_rows(t, [&](int i) {
        s.ls[i] = s.E[i] * s.l[i];
        s.us[i] = s.E[i] * s.u[i];
        // set_rho_vec / update_rho_vec (auxil.c:79-141): rho_vec is a function of (type, rho) in both
        const int ty = (s.ls[i] < -kInfty * kMinScaling && s.us[i] > kInfty * kMinScaling) ? -1 : (s.us[i] - s.ls[i] < kRhoTol ? 1 : 0);
        s.ctype[i] = ty;
      });
      for_tiles(t, [&](Tv &v, int) { store_tile(v, Pg); });   // keep P_s for residuals, re-factorisations and polish
    });
    lap(5);
  }

  MPC_HD void set_rho_vec() {   // phase: rho_vec from (ctype, rho)  (auxil.c:79-96, osqp.c:1267-1310)
    ex.par([&](Th &t) {
      if constexpr (C::kRhoPerType) {
        if (t.tid < 3) {
          const double rv = t.tid == 0 ? kRhoMin : (t.tid == 2 ? kRhoEqOverIneq * s.rho : s.rho);
          s.rho3[t.tid] = rv;
          s.rinv3[t.tid] = 1.0 / rv;
        }
      } else {
        for_rows(t, [&](int i) {
          const int ty = s.ctype[i];
          const double rv = ty == -1 ? kRhoMin : (ty == 1 ? kRhoEqOverIneq * s.rho : s.rho);
          s.rho_vec[i] = rv;
          s.rho_inv[i] = 1.0 / rv;
        });
      }
    });
  }

  // ================================ 3. K = P_s + sigma I + A^T R A ; Mx <- -K^{-1} ==============
  // A^T R A is block diagonal (3 x 3 per foot): only the diagonal tiles (feet 2 ti, 2 ti + 1) change.
  MPC_HD void factor(bool reload) {
    ex.par([&](Th &t) {
      for_tiles(t, [&](Tv &v, int) {
        if (reload) load_tile(v, Pg);
        if (v.dia) {
#pragma unroll
          for (int fr = 0; fr < 2; ++fr) {
            const int f = 2 * v.ti + fr;
            const double *a = s.As + 15 * f;
            double rv[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) rv[r] = rho_at(5 * f + r);
#pragma unroll
            for (int c1 = 0; c1 < 3; ++c1)
#pragma unroll
              for (int c2 = 0; c2 < 3; ++c2) {
                double g = 0;
#pragma unroll
                for (int r = 0; r < 5; ++r) g += a[3 * r + c1] * rv[r] * a[3 * r + c2];
                if (c1 == c2) g += kSigma;
                v.Mx[(3 * fr + c1) * TS + 3 * fr + c2] += g;
              }
          }
        }
      });
    });
    lap(6);
    sweep_all(false);
    ex.par([&](Th &t) { if (t.tid == 0) s.nfact++; });
    lap(7);
  }

  // Symmetric sweep of every pivot (masked: the update is skipped for pivots with !isnull[k]).  After
  // all pivots the matrix equals -inverse.  Per step k:  p = a_kk;  a_ij -= a_ik a_kj / p (i,j != k);
  // a_ik -> a_ik / p;  a_kk -> -1/p.  The matrix stays symmetric, so both a_ik and a_kj are read from the
  // published pivot row, and only the lower-triangle tiles are updated.  That row carries (p - 1) in
  // slot k, which makes the generic update
  //   a_ij -= (row_k[i] / p) * row_k[j]
  // produce a_ik / p on column k and a_kj / p on row k with no per-element select; the diagonal element
  // of a swept row takes the generic update too and ends up as (true value + 2) -- the matrix-vector
  // products add 2 v[row] back (inv_combine).  Un-swept diagonal elements are exact, so the pivot is
  // read straight from the diagonal tile.
  // The pivot loop is unrolled by TS = 6 so that the pivot's position inside its tile is static.
  MPC_HD void sweep_all(bool masked) {
    int buf = 0;
    pin_tiles(0);
    ex.par([&](Th &t) { for_tiles(t, [&](Tv &v, int) { publish<0>(v, 0, 0); }); });
    for (int kt = 0; kt < G; ++kt) {
      pin_tiles(4);
      // bit A: pivot 6 kt + A is swept; bit 6: so is the first pivot of the next tile row (one LDS read per six steps)
      int bits = kt + 1 < G ? 0x7f : 0x3f;
      if (masked) bits = MPC_UNIFORM_INT(s.rowmask[kt] | (kt + 1 < G ? (s.rowmask[kt + 1] & 1) << TS : 0));
      sweep_steps<0>(bits, kt, buf);
    }
    pin_tiles(1);
  }
  template <int A>
  MPC_HD void sweep_steps(int bits, int kt, int &buf) {
    if constexpr (A < TS) {
      sweep_step<A>(bits, kt, buf);
      buf ^= 1;
      sweep_steps<A + 1>(bits, kt, buf);
    }
  }
  // Pivot step k = 6 kt + A.  Order inside the phase: first the cross through the next pivot (row AN and
  // column AN of every tile, 11 FMAs), then the tiles of tile row / tile column of the next pivot publish
  // row k + 1 -- so that the LDS stores and the reciprocal are in flight while the other 25 entries take
  // their update.
  template <int A>
  MPC_HD void sweep_step(int bits, int kt, int buf) {
    constexpr int AN = (A + 1) % TS;                 // next pivot's position inside its tile
    const int ktn = (A + 1 < TS) ? kt : kt + 1;      // tile row (= column) of the next pivot
    const bool active = (bits >> A) & 1;             // uniform
    const bool pub = (bits >> (A + 1)) & 1;          // the next pivot row is only needed if that pivot is used
    if (!active && !pub) return;
    ex.par([&](Th &t) {
      for_tiles(t, [&](Tv &v, int) {   // (a thread with two tiles finishes one before it starts the other: 24 live VGPRs of g / pc, not 48)
        double g[TS], pc[TS];
        if (active) {
          const double p = s.piv[buf][0], pinv = s.piv[buf][1];
          const double *pr = s.prow(buf);
#pragma unroll
          for (int a = 0; a < TS; ++a) { g[a] = pr[TS * v.ti + a] * pinv; pc[a] = pr[TS * v.tj + a]; }
#pragma unroll
          for (int b = 0; b < TS; ++b) v.Mx[AN * TS + b] -= g[AN] * pc[b];
#pragma unroll
          for (int a = 0; a < TS; ++a)
            if (a != AN) v.Mx[a * TS + AN] -= g[a] * pc[AN];
          if (v.index == 0 && !(p > 0)) s.bad = 1;     // not positive definite
        }
        if (pub) publish<AN>(v, buf ^ 1, ktn);
        MPC_SCHED_FENCE();
        if (active) {
#pragma unroll
          for (int a = 0; a < TS; ++a)
#pragma unroll
            for (int b = 0; b < TS; ++b)
              if (a != AN && b != AN) v.Mx[a * TS + b] -= g[a] * pc[b];
        }
      });
    });
  }
  // Row k = 6 kt + A of the matrix -> prow[b]: the tiles of tile row kt hold its part left of (and on) the
  // diagonal as their row A, the tiles of tile column kt hold the rest as their column A.  Slot k itself
  // gets (pivot - 1), and piv[b] = {pivot, 1 / pivot}.
  template <int A>
  MPC_HD void publish(const Tv &t, int b, int kt) {
    if (t.ti == kt) {
      double *pn = s.prow(b) + TS * t.tj;
#pragma unroll
      for (int bb = 0; bb < TS; ++bb)
        if (bb != A) pn[bb] = t.Mx[A * TS + bb];
      const double pivot = t.Mx[A * TS + A];
      pn[A] = t.dia ? pivot - 1.0 : pivot;
      if (t.dia) {
        s.piv[b][0] = pivot;
        s.piv[b][1] = fast_recip(pivot);
      }
    } else if (t.tj == kt) {
      // (a column of the tile: its elements are not register neighbours, and a 128-bit store would first copy each pair
      //  into an aligned register quad -- four v_mov per store on the VALU, which is the busy unit.  Volatile keeps the
      //  six 64-bit stores apart (MPC_LDS_STORE64); they go to the LDS queue, which has slack.)
      double *pn = s.prow(b) + TS * t.ti;
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        if constexpr (C::kColumnStore64) MPC_LDS_STORE64(pn + a, t.Mx[a * TS + A]);
        else pn[a] = t.Mx[a * TS + A];
      }
    }
  }
  static MPC_HD double fast_rsqrt(double d) {   // 1 / sqrt(d), d > 0 finite: v_rsq_f64 + two Newton steps (to the last ulp or two)
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = r * (1.5 - h * r * r);
    r = r * (1.5 - h * r * r);
    return r;
#else
    return 1.0 / sqrt(d);
#endif
  }
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

  // ================================ 4. ADMM (auxil.c:164-228) ===================================
  // Two phases per iteration:
  //   M: part <- (-Mx) rhs                       (tile threads; Mx = -K^{-1} + 2 I on the diagonal slots)
  //   V: one thread per foot finishes the iteration for its 3 variables and 5 constraint rows
  //      (x~ from the partials, x, z~ = A x~, z, y) and forms the next right-hand side
  //      rhs = sigma x - q + A^T (R z - y).
  MPC_HD void admm_prepare() {   // rhs for the first iteration / after a rho update
    ex.par([&](Th &t) {
      if (t.tid < NF) {
        const int f = t.tid;
        const double *a = s.As + 15 * f;
        double tm[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) tm[r] = rho_at(5 * f + r) * cz()[5 * f + r] - cy()[5 * f + r];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          double acc = 0;
#pragma unroll
          for (int r = 0; r < 5; ++r) acc += a[3 * r + c] * tm[r];
          crhs()[3 * f + c] = kSigma * s.x[3 * f + c] - s.qs[3 * f + c] + acc;
        }
      }
    });
  }
  // One ADMM iteration = four short phases (each a single LDS round trip; the tile leaves few free
  // registers, so long per-thread programs serialise into many round trips):
  //   M: part <- (-Mx) rhs                       (tile threads; Mx = -K^{-1} + 2 I on the diagonal slots)
  //   C: x~_j = sum of the partials + 2 rhs_j    (one thread per variable)
  //   R: z~_i = (A x~)_i ; z, y update ; tm_i = rho_i z_i - y_i      (one thread per constraint row)
  //   X: x update, P_s x recursion, next rhs_j = sigma x_j - q_j + (A^T tm)_j   (one thread per variable)
  MPC_HD void admm_iter() {
    ex.par([&](Th &t) {
      for_tiles(t, [&](Tv &v, int) { tile_matvec(v, crhs()); });
    });
    MPC_SUBLAP(5, 9);
    ex.par([&](Th &t) {
      if (t.tid < N) s.xt[t.tid] = inv_combine(s, t.tid, crhs());
    });
    MPC_SUBLAP(5, 10);
    ex.par([&](Th &t) {
      for_rows(t, [&](int i) {   // (all LDS loads first, as one batch: a single round trip per phase)
        const int f = i / 5, r = i - 5 * f;
        const double *a = s.As + 15 * f + 3 * r, *xt = s.xt + 3 * f;
        const double a0 = a[0], a1 = a[1], a2 = a[2], x0 = xt[0], x1 = xt[1], x2 = xt[2];
        const double zp = cz()[i], yv = cy()[i], rv = rho_at(i), ri = rinv_at(i), lo = s.ls[i], hi = s.us[i];
        MPC_SCHED_FENCE();
        const double zt = a0 * x0 + a1 * x1 + a2 * x2;
        const double zr = kAlphaRelax * zt + (1.0 - kAlphaRelax) * zp;
        const double zn = clampd(zr + ri * yv, lo, hi);
        const double yn = yv + rv * (zr - zn);
        cz()[i] = zn;
        cy()[i] = yn;
        s.tm[i] = rv * zn - yn;
        s.rzt[i] = rv * zt;
      });
    });
    MPC_SUBLAP(5, 11);
    ex.par([&](Th &t) {
      if (t.tid < N) {
        const int j = t.tid, f = j / 3, c = j - 3 * f;
        const double *a = s.As + 15 * f + c, *tm = s.tm + 5 * f, *rz = s.rzt + 5 * f;
        double av[5], tv[5], rzv[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) { av[r] = a[3 * r]; tv[r] = tm[r]; rzv[r] = rz[r]; }
        const double xt = s.xt[j], xp = s.x[j], rh = crhs()[j], px = s.Px[j], qv = s.qs[j];
        MPC_SCHED_FENCE();
        double acc = 0, arz = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) { acc += av[r] * tv[r]; arz += av[r] * rzv[r]; }
        // P_s x without a matrix product: K x~ = rhs gives P_s x~ = rhs - sigma x~ - A^T R z~, and x is affine in x~
        s.Px[j] = kAlphaRelax * (rh - kSigma * xt - arz) + (1.0 - kAlphaRelax) * px;
        const double xn = kAlphaRelax * xt + (1.0 - kAlphaRelax) * xp;
        s.x[j] = xn;
        crhs()[j] = kSigma * xn - qv + acc;
      }
    });
    MPC_SUBLAP(5, 12);
  }

  // P_s v -> out (P_s tiles read from HBM scratch, each used in both orientations).  Two phases.
  MPC_HD void mul_P(const double *v, double *out) {
    ex.par([&](Th &t) {
      for_tiles(t, [&](Tv &tv, int) {
        const double *g = Pg + (size_t)tv.index * TE;
        double vc[TS], vr[TS], ar[TS], ac[TS];
#pragma unroll
        for (int b = 0; b < TS; ++b) { vc[b] = v[TS * tv.tj + b]; vr[b] = v[TS * tv.ti + b]; ar[b] = 0; ac[b] = 0; }
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
          for (int b = 0; b < TS; ++b) {
            const double m = g[a * TS + b];
            ar[a] += m * vc[b];
            ac[b] += m * vr[a];
          }
        double *pd = s.part + tv.tj * NP + TS * tv.ti, *pt = s.part + tv.ti * NP + TS * tv.tj;
#pragma unroll
        for (int a = 0; a < TS; ++a) pd[a] = ar[a];
        if (!tv.dia) {
#pragma unroll
          for (int b = 0; b < TS; ++b) pt[b] = ac[b];
        }
      });
    });
    ex.par([&](Th &t) { if (t.tid < N) out[t.tid] = sum_parts(s, t.tid); });
  }

  // residuals of (x, z, y) (auxil.c:243-306, 563-629) + the norms termination and rho need.
  // red[] slots: 0 pri_res 1 ||Einv z|| 2 ||Einv Ax|| 3 ||rp|| 4 ||z|| 5 ||Ax||
  //              6 ||Dinv rd|| 7 ||Dinv q|| 8 ||Dinv Aty|| 9 ||Dinv Px|| 10 ||rd|| 11 ||q|| 12 ||Aty|| 13 ||Px||
  // 64 threads stride over the rows and keep running maxima in registers; 14 threads finish.
  static constexpr int kRedW = 64;
  MPC_HD void residuals(const double *x, const double *z, const double *y, const double *Px) {
    ex.par([&](Th &t) {
      if (t.tid < kRedW) {
        double mx[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) mx[k] = 0;
        for (int i = t.tid; i < M; i += kRedW) {
          const double ax = a_row_dot(s, i, x), r = ax - z[i], ei = 1.0 / s.E[i];
          mx[0] = dmax(mx[0], fabs(ei * r)); mx[1] = dmax(mx[1], fabs(ei * z[i])); mx[2] = dmax(mx[2], fabs(ei * ax));
          mx[3] = dmax(mx[3], fabs(r)); mx[4] = dmax(mx[4], fabs(z[i])); mx[5] = dmax(mx[5], fabs(ax));
        }
        for (int j = t.tid; j < N; j += kRedW) {
          const double aty = at_col_dot(s, j, y), px = Px[j], qv = s.qs[j], r = qv + px + aty, di = 1.0 / s.D[j];
          mx[6] = dmax(mx[6], fabs(di * r)); mx[7] = dmax(mx[7], fabs(di * qv)); mx[8] = dmax(mx[8], fabs(di * aty));
          mx[9] = dmax(mx[9], fabs(di * px)); mx[10] = dmax(mx[10], fabs(r)); mx[11] = dmax(mx[11], fabs(qv));
          mx[12] = dmax(mx[12], fabs(aty)); mx[13] = dmax(mx[13], fabs(px));
        }
#pragma unroll
        for (int k = 0; k < 14; ++k) s.part[k * kRedW + t.tid] = mx[k];
      }
    });
    ex.par([&](Th &t) {
      if (t.tid < 14) {
        const double *p = s.part + t.tid * kRedW;
        double m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        for (int k = 0; k < kRedW; k += 4) { m0 = dmax(m0, p[k]); m1 = dmax(m1, p[k + 1]); m2 = dmax(m2, p[k + 2]); m3 = dmax(m3, p[k + 3]); }
        s.red[t.tid] = dbits(dmax(dmax(m0, m1), dmax(m2, m3)));
      }
    });
  }

  // check_termination (auxil.c:684-793; infeasibility certificates not evaluated: the QP is always
  // feasible and strictly convex) + adapt_rho decision (auxil.c:13-77).  One thread decides.
  MPC_HD void check_and_adapt(int iter) {
    ex.par([&](Th &t) {
      if (t.tid == 0) {
        const double pri = bitsd(s.red[0]), dua = s.cinv * bitsd(s.red[6]);
        s.pri_res = pri; s.dua_res = dua; s.iter = iter; s.rho_new = 0;
        if (pri > kInfty || dua > kInfty) { s.status = kStNonCvx; s.done = 1; }
        else {
          const double eps_prim = kEpsAbs + kEpsRel * dmax(bitsd(s.red[1]), bitsd(s.red[2]));
          const double eps_dual = kEpsAbs + kEpsRel * s.cinv * dmax(dmax(bitsd(s.red[7]), bitsd(s.red[8])), bitsd(s.red[9]));
          if (pri < eps_prim && dua < eps_dual) { s.status = kStSolved; s.done = 1; }
          else {
            double pr = bitsd(s.red[3]) / (dmax(bitsd(s.red[4]), bitsd(s.red[5])) + 1e-10);
            double dr = bitsd(s.red[10]) / (dmax(dmax(bitsd(s.red[11]), bitsd(s.red[12])), bitsd(s.red[13])) + 1e-10);
            double rn = s.rho * sqrt(pr / (dr + 1e-10));
            rn = clampd(rn, kRhoMin, kRhoMax);
            if (rn > s.rho * kAdaptTol || rn < s.rho / kAdaptTol) s.rho_new = rn;
          }
        }
      }
    });
  }

  // ================================ 5. polish (polish.c) ========================================
  MPC_HD void polish() {
    // active set (polish.c:36-52) and per-foot bases
    ex.par([&](Th &t) {
      for_rows(t, [&](int i) { s.act[i] = (cz()[i] - s.ls[i] < -cy()[i]) ? -1 : ((s.us[i] - cz()[i] < cy()[i]) ? 1 : 0); });
    });
    ex.par([&](Th &t) {
      if (t.tid < NF) {
        const int f = t.tid;
        const double *a = s.As + 15 * f;
        // (all register arrays are indexed statically: a runtime row index would push them to scratch memory.
        //  Rows of Q beyond the current rank are zero, so projecting on all three rows equals projecting on the
        //  first r of them.)
        double Q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Nn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int r = 0;
#pragma unroll
        for (int row = 0; row < 5; ++row) {
          const bool cand = r < 3 && s.act[5 * f + row];
          double v[3] = {a[3 * row], a[3 * row + 1], a[3 * row + 2]};
          const double n0 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#pragma unroll
          for (int pass = 0; pass < 2; ++pass)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const double d = v[0] * Q[3 * k] + v[1] * Q[3 * k + 1] + v[2] * Q[3 * k + 2];
              v[0] -= d * Q[3 * k]; v[1] -= d * Q[3 * k + 1]; v[2] -= d * Q[3 * k + 2];
            }
          const double nr = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
          const bool take = cand && nr > 1e-6 * n0;
          const double q0 = v[0] / nr, q1 = v[1] / nr, q2 = v[2] / nr;
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
          const double nr = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
          Nn[0] = v[0] / nr; Nn[1] = v[1] / nr; Nn[2] = v[2] / nr;
          Nn[3] = Q[1] * Nn[2] - Q[2] * Nn[1]; Nn[4] = Q[2] * Nn[0] - Q[0] * Nn[2]; Nn[5] = Q[0] * Nn[1] - Q[1] * Nn[0];
        } else if (r == 2) {
          Nn[0] = Q[1] * Q[5] - Q[2] * Q[4]; Nn[1] = Q[2] * Q[3] - Q[0] * Q[5]; Nn[2] = Q[0] * Q[4] - Q[1] * Q[3];
          const double nr = sqrt(Nn[0] * Nn[0] + Nn[1] * Nn[1] + Nn[2] * Nn[2]);
          Nn[0] /= nr; Nn[1] /= nr; Nn[2] /= nr;
        }
        const int nn = 3 - r;
        s.nnull[f] = nn;
        for (int k = 0; k < 9; ++k) s.Nb[9 * f + k] = Nn[k];          // row k (< nn) = null vector k
        for (int k = 0; k < 3; ++k) s.isnull[3 * f + k] = k < nn;     // null coordinate (f, k) sits at index 3f+k
        // Gamma = Q (Q^T B Q)^{-1} Q^T with B = A_act^T A_act
        double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Gi[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int row = 0; row < 5; ++row)
          if (s.act[5 * f + row])
            for (int c1 = 0; c1 < 3; ++c1) for (int c2 = 0; c2 < 3; ++c2) B[3 * c1 + c2] += a[3 * row + c1] * a[3 * row + c2];
        for (int k1 = 0; k1 < 3; ++k1) for (int k2 = 0; k2 < 3; ++k2) {
          if (k1 >= r || k2 >= r) continue;
          double tt = 0;
          for (int c1 = 0; c1 < 3; ++c1) for (int c2 = 0; c2 < 3; ++c2) tt += Q[3 * k1 + c1] * B[3 * c1 + c2] * Q[3 * k2 + c2];
          G[3 * k1 + k2] = tt;
        }
        // Gauss-Jordan on the (identity padded) 3x3
        for (int p = 0; p < 3; ++p) {
          const double d = 1.0 / G[3 * p + p];
          for (int j = 0; j < 3; ++j) { G[3 * p + j] *= d; Gi[3 * p + j] *= d; }
          for (int i = 0; i < 3; ++i) {
            if (i == p) continue;
            const double fc = G[3 * i + p];
            for (int j = 0; j < 3; ++j) { G[3 * i + j] -= fc * G[3 * p + j]; Gi[3 * i + j] -= fc * Gi[3 * p + j]; }
          }
        }
        for (int c1 = 0; c1 < 3; ++c1) for (int c2 = 0; c2 < 3; ++c2) {
          double tt = 0;
          for (int k1 = 0; k1 < 3; ++k1) for (int k2 = 0; k2 < 3; ++k2) {
            if (k1 >= r || k2 >= r) continue;
            tt += Q[3 * k1 + c1] * Gi[3 * k1 + k2] * Q[3 * k2 + c2];
          }
          s.Gm[9 * f + 3 * c1 + c2] = tt;
        }
      }
    });
    MPC_SUBLAP(3, 9);
    // u = Gamma A_act^T b  (the point satisfying the active rows), g = -q - P u
    ex.par([&](Th &t) {
      if (t.tid < N) {
        const int j = t.tid, f = j / 3;
        double v[3];
        for (int c = 0; c < 3; ++c) {
          double tt = 0;
          for (int r = 0; r < 5; ++r) {
            const int i = 5 * f + r;
            if (s.act[i]) tt += s.As[15 * f + 3 * r + c] * (s.act[i] < 0 ? s.ls[i] : s.us[i]);
          }
          v[c] = tt;
        }
        const double *G = s.Gm + 9 * f + 3 * (j - 3 * f);
        s.u0[j] = G[0] * v[0] + G[1] * v[1] + G[2] * v[2];
        s.xN[j] = 0; s.PxN[j] = 0; wv()[j] = 0;
        if (j % TS == 0) {
          int m = 0;
          for (int b = 0; b < TS; ++b) m |= (s.isnull[j + b] ? 1 : 0) << b;
          s.rowmask[j / TS] = m;
        }
      }
    });
    MPC_SUBLAP(3, 10);
    mul_P(s.u0, s.Pu);
    MPC_SUBLAP(3, 11);
    lap(11);
    // H = N~^T P N~ + delta I on the null coordinates, identity elsewhere; N~ = blockdiag([N_f | 0]).
    // Tile-local: 2 row feet x 2 column feet of 3 x 3 blocks, all register indices static.
    ex.par([&](Th &t) {
      if (t.tid < N) s.g[t.tid] = -s.qs[t.tid] - s.Pu[t.tid];
      for_tiles(t, [&](Tv &tv, int) {
        load_tile(tv, Pg);
#pragma unroll
        for (int fr = 0; fr < 2; ++fr)
#pragma unroll
          for (int fc = 0; fc < 2; ++fc) {
            const int rf = 2 * tv.ti + fr, cf = 2 * tv.tj + fc;
            const double *nr = s.Nb + 9 * rf, *nc = s.Nb + 9 * cf;   // row k = null vector k (zero rows beyond nnull)
            const int nnr = s.nnull[rf], nnc = s.nnull[cf];
            double T1[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int k2 = 0; k2 < 3; ++k2)
                T1[3 * r + k2] = tv.Mx[(3 * fr + r) * TS + 3 * fc] * nc[3 * k2] + tv.Mx[(3 * fr + r) * TS + 3 * fc + 1] * nc[3 * k2 + 1] +
                                 tv.Mx[(3 * fr + r) * TS + 3 * fc + 2] * nc[3 * k2 + 2];
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1)
#pragma unroll
              for (int k2 = 0; k2 < 3; ++k2) {
                double v = (k1 < nnr && k2 < nnc) ? nr[3 * k1] * T1[k2] + nr[3 * k1 + 1] * T1[3 + k2] + nr[3 * k1 + 2] * T1[6 + k2] : 0.0;
                if (k1 == k2 && rf == cf) v = (k1 < nnr) ? v + kDelta : 1.0;
                tv.Mx[(3 * fr + k1) * TS + 3 * fc + k2] = v;
              }
          }
      });
    });
    MPC_SUBLAP(3, 12);
    sweep_all(true);   // Mx <- -(H + delta I)^{-1} on the null coordinates
    ex.par([&](Th &t) { if (t.tid == 0) s.nfact++; });
    lap(12);
    // delta-regularised solve + 3 refinement steps (polish.c:102-160) in the null space: with Hd = H + delta I,
    // w_{k+1} = w_k + Hd^{-1} rho_k and rho_k = b - H w_k obey rho_{k+1} = delta Hd^{-1} rho_k (H Hd^{-1} = I - delta Hd^{-1}),
    // so the refinement needs no further products with P.
    ex.par([&](Th &t) {   // rho_0 = N~^T g
      if (t.tid < N) {
        const int j = t.tid, f = j / 3, k = j - 3 * f;
        const double *nv = s.Nb + 9 * f + 3 * k;
        rw()[j] = (k < s.nnull[f]) ? nv[0] * s.g[3 * f] + nv[1] * s.g[3 * f + 1] + nv[2] * s.g[3 * f + 2] : 0.0;
      }
    });
    for (int it = 0; it <= kPolishRefine; ++it) {
      ex.par([&](Th &t) { for_tiles(t, [&](Tv &v, int) { tile_matvec(v, rw()); }); });
      ex.par([&](Th &t) {
        if (t.tid < N && s.isnull[t.tid]) {
          const double dw = inv_combine(s, t.tid, rw());
          wv()[t.tid] += dw;
          rw()[t.tid] = kDelta * dw;
        }
      });
    }
    ex.par([&](Th &t) {   // xN = N~ w
      if (t.tid < N) {
        const int j = t.tid, f = j / 3, c = j - 3 * f;
        double v = 0;
        for (int k = 0; k < 3; ++k) if (k < s.nnull[f]) v += s.Nb[9 * f + 3 * k + c] * wv()[3 * f + k];
        s.xN[j] = v;
      }
    });
    mul_P(s.xN, s.PxN);
    lap(13);
    // x = u + xN ; y = A Gamma (g - P xN) on active rows ; z = A x ; normal-cone projection (proj.c:17-31)
    ex.par([&](Th &t) {
      if (t.tid < N) {
        const int j = t.tid, f = j / 3;
        s.xt[j] = s.u0[j] + s.xN[j];                                   // polished x (scaled)
        const double *G = s.Gm + 9 * f + 3 * (j - 3 * f);
        rw()[j] = G[0] * (s.g[3 * f] - s.PxN[3 * f]) + G[1] * (s.g[3 * f + 1] - s.PxN[3 * f + 1]) + G[2] * (s.g[3 * f + 2] - s.PxN[3 * f + 2]);
      }
    });
    ex.par([&](Th &t) {
      for_rows(t, [&](int i) {
        const double yv = s.act[i] ? a_row_dot(s, i, rw()) : 0.0;
        const double tt = a_row_dot(s, i, s.xt) + yv;
        const double zc = clampd(tt, s.ls[i], s.us[i]);
        zpol()[i] = zc;
        ypol()[i] = tt - zc;
      });
    });
    // residuals at the polished point, acceptance (polish.c:306-345)
    const double pri0 = s.pri_res, dua0 = s.dua_res;
    ex.par([&](Th &t) { if (t.tid < N) s.Pu[t.tid] += s.PxN[t.tid]; });   // P_s x_pol
    residuals(s.xt, zpol(), ypol(), s.Pu);
    ex.par([&](Th &t) {
      if (t.tid == 0) {
        const double pri = bitsd(s.red[0]), dua = s.cinv * bitsd(s.red[6]);
        const bool ok = !s.bad && ((pri < pri0 && dua < dua0) || (pri < pri0 && dua0 < 1e-10) || (dua < dua0 && pri0 < 1e-10));
        s.status_polish = ok ? 1 : -1;
        if (ok) { s.pri_res = pri; s.dua_res = dua; }
      }
    });
    ex.par([&](Th &t) {
      if (s.status_polish == 1) {
        if (t.tid < N) s.x[t.tid] = s.xt[t.tid];
        for_rows(t, [&](int i) { cz()[i] = zpol()[i]; cy()[i] = ypol()[i]; });
      }
    });
  }

  // ================================ driver ======================================================
  MPC_HD void run() {
    const long long t0 = MPC_CLOCK();
    tlast = t0;
    load();
    scale();
    set_rho_vec();
    factor(false);
    if (!s.first) mul_P(s.x, s.Px);                       // warm start: P_s x_0 once, then carried by recursion
    else ex.par([&](Th &t) { if (t.tid < N) s.Px[t.tid] = 0.0; });
    lap(9);
    admm_prepare();
    lap(8);
    // kCheck iterations between termination checks (osqp.c:417-517 checks when iter % 25 == 0): the iterations
    // run in their own inner loop so that the register allocator keeps the tile resident across them and places
    // its live-range splits around the check / refactor code, which runs 25x less often.
    static_assert(kMaxIter % kCheck == 0, "the check falls on the last iteration");
    int iter = 0;
    while (!s.done && !s.bad && iter < kMaxIter) {
      pin_tiles(2);
      for (int k = 0; k < kCheck; ++k) admm_iter();
      pin_tiles(3);
      iter += kCheck;
      lap(8);
      residuals(s.x, cz(), cy(), s.Px);
      check_and_adapt(iter);
      lap(10);
      if (!s.done && s.rho_new > 0) {          // osqp_update_rho: new rho_vec, refactor
        ex.par([&](Th &t) { if (t.tid == 0) { s.rho = s.rho_new; s.rho_updates++; } });
        set_rho_vec();
        factor(true);
        admm_prepare();
        lap(8);
      }
    }
    if (C::kLoopExitFence) lap(8);   // (the clock read of an instrumented build is a fence too)
    // Register allocation of the whole kernel hinges on whether the scheduler may move code across the loop exit
    // (measured, hipcc 7.2, and re-measured whenever the code around it changes): a fence here is worth 3.5 % at h = 10 and,
    // with the current code, takes h = 16 from 0.38 to 0.54 M steps/s; h = 20 is faster without it.
    if (C::kLoopExitFence) MPC_SCHED_FENCE();
    if (!s.done && !s.bad) {   // max_iter reached (osqp.c:564-568): only SOLVED counts for the reference
      ex.par([&](Th &t) { if (t.tid == 0) s.status = kStMaxIter; });
    }
    if (s.status == kStSolved && !s.bad) polish();
    lap(14);
    tc[15] = MPC_CLOCK() - t0;
    // outputs + persistent state (store_solution, auxil.c:528-561; mpc_osqp.cc:788-790: forces = -x)
    ex.par([&](Th &t) {
      const bool solved = s.status == kStSolved && !s.bad;
      if (t.tid < N) {
        if (solved) forces[t.tid] = -(s.D[t.tid] * s.x[t.tid]);
        state[t.tid] = s.x[t.tid];
        state[N + 2 * M + t.tid] = q_at(t.tid);
      }
      for_rows(t, [&](int i) { state[N + i] = cz()[i]; state[N + M + i] = cy()[i]; });
      if (t.tid == 0) {
        state[2 * N + 2 * M] = s.rho;
        state[2 * N + 2 * M + 1] = 1.0;
        info[0] = s.iter; info[1] = s.bad ? kStNonCvx : s.status; info[2] = s.status_polish; info[3] = s.rho_updates;
        info[4] = s.nfact; info[5] = s.first; info[6] = 0; info[7] = 0;
        if (prof) for (int k = 0; k < kProfLen; ++k) if (k != 1 && k != 2) prof[k] = tc[k];   // (1, 2: the assembly kernel's)
      }
    });
  }
};

}  // namespace mpc
