// mpc_core.h -- the prep kernel of one robot's convex-MPC contact-force solve, written as sequences of barrier-separated
// phases over the threads of one workgroup: Assembler (QP assembly) and Scaler (OSQP's Ruiz equilibration on the dense P), plus
// what the solve kernel shares with them (constants, Cfg, the record layouts).  The OSQP iteration itself -- re-expressed in
// the 6 h-dimensional space of the net body wrenches -- is Solver in mpc_wrench.h.
//
//   Device build (mpc_batch.hip): Exec::par(f) = { f(thread); __syncthreads(); } -- the per-thread
//   state lives in VGPRs, the shared structs in LDS.
//   Host emulation (tests/emu): Exec::par(f) runs f for every emulated thread (in forward or reverse
//   order, to expose intra-phase races); used only by the CPU tests.
//
// What is computed (reference boundary: MPC_Controller/convex_MPC/mpc_osqp.cc:578-796,
// ConvexMpc::ComputeContactForces, OSQP branch):
//   1. single-rigid-body QP assembly (mpc_osqp.cc:606-688): x0, x_ref, A dt / B dt, the exact exponential, q, bounds, cone
//      block, and P in its wrench form  P = alpha I + BB^T Theta BB  (B6, th1, th2; the reference's block recursion :387-434
//      summed in closed form) -> QP record; the two 12 x 12 tables U1, U2 from which any entry of the dense P follows;
//   2. Ruiz scaling + cost scaling of the OSQP 0.6.0 algorithm the reference calls on it (extern/osqp/src/scaling.c:44-156)
//      -> scale record.  ADMM, residuals / termination, rho adaptation and polish: mpc_wrench.h.
//   All arithmetic is fp64: an fp32 ADMM does not reproduce OSQP's iterates (oracle/README).
//
// Thread layout of the dense P (n = 12 H; only the Ruiz norms need it): P is SYMMETRIC and is held as the lower triangle of a
// G x G grid (G = 2 H) of 6 x 6 register tiles (2 feet x 2 feet), one tile per thread (four at h = 20), built in registers from
// U1 / U2 and never stored: tile index ti (ti + 1) / 2 + tj holds tile (ti, tj), tj <= ti; thread tid owns tiles tid,
// tid + MTH, ...; a diagonal tile is held in full.  An off-diagonal tile stands for itself and for its transpose.
// Vector phases use tid < n and the constraint rows tid, tid + T, ... < m (Scaler::for_rows).
//
// Per-horizon tuning knobs (Cfg: NT, kPinMask, kQInLds) do not change any arithmetic, only how the compiler allocates
// registers, and are set from measurements (tools/isa_census.py).
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
// Two adjacent doubles (16-byte aligned) as one ds_read_b128 that is neither hoisted out of a loop nor split
typedef double mpc_double2 __attribute__((ext_vector_type(2)));
#define MPC_LDS_LOAD128(p, lo, hi) do { const mpc_double2 v2_ = *(const volatile __attribute__((address_space(3))) mpc_double2 *)(p); (lo) = v2_.x; (hi) = v2_.y; } while (0)
// Loads / stores of what one job of a solve hands to the other (mpc_wrench.h admm_job -> polish_job: the state record, forces, info): device-scope
// coherent accesses (sc1: stores write through, loads do not trust a possibly stale line of this XCD's L2), so that the hand-over needs no
// L2 write-back / invalidate -- the jobs of one solve may run on different XCDs, each with its own L2.
#define MPC_GST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MPC_GLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define MPC_GST(p, v) (*(p) = (v))
#define MPC_GLD(p) (*(p))
#define MPC_LDS_LOAD128(p, lo, hi) do { (lo) = (p)[0]; (hi) = (p)[1]; } while (0)
#define MPC_LDS_STORE64(p, v) (*(p) = (v))
#define MPC_LDS_LOAD64(p) (*(p))
#define MPC_LAUNDER(x) ((void)0)
#define MPC_SCHED_FENCE() ((void)0)
#endif
// Tuning hooks for tools/build_variant.sh experiments (the product build uses the defaults; see Cfg for what they mean):
//   MPC_NT_H16 / MPC_NT_H20   dense-P tiles per thread of the prep kernel at h = 16 / 20
//   MPC_PIN_MASK              live-range split points of the prep kernel's tile registers
//   MPC_PROW_SKEW             1: pivot rows of the solve kernel 8 bytes off the 16-byte grid
//   MPC_PART_ROWMAJOR_MAXT    the largest workgroup that keeps the partial products of the solve kernel's tile mat-vec as [row][slot] (else [slot][row]);
//   MPC_PART_PAD              extra doubles per row of the [row][slot] layout
//   MPC_QUAD_SCATTER          1: the per-step wrench sums as a quad reduce-scatter (0: all-sum of all six, then a select)
#ifndef MPC_SHARE_ROLE_REGS       // tile and foot state of a solve-kernel thread in the same registers where the roles are different threads
#define MPC_SHARE_ROLE_REGS 1
#endif
#ifndef MPC_FOOT0                 // first foot lane of the solve kernel's workgroup for horizon H with MTW tile lanes
#define MPC_FOOT0(H, MTW) (((H) == 12 || (H) == 16) ? (((MTW) + 3) / 4) * 4 : 0)
#endif
#ifndef MPC_PART_ROWMAJOR_MAXT
#define MPC_PART_ROWMAJOR_MAXT 64
#endif
#define MPC_PART_ROWMAJOR(T) ((T) <= MPC_PART_ROWMAJOR_MAXT)
#ifndef MPC_PART_PAD
#define MPC_PART_PAD 0
#endif
static_assert(MPC_PART_PAD % 2 == 0, "MPC_PART_PAD must be even: a row of the [row][slot] partial products starts on a 16-byte boundary (MPC_LDS_LOAD128 in sum_parts)");
#ifndef MPC_QUAD_SCATTER
#define MPC_QUAD_SCATTER 1
#endif
#ifndef MPC_PROW_SKEW
#define MPC_PROW_SKEW 0
#endif
#ifndef MPC_NT_H20
#define MPC_NT_H20 2
#endif
#ifndef MPC_NT_H16
#define MPC_NT_H16 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MPC_CLOCK() ((long long)__builtin_readcyclecounter())
#else
#define MPC_CLOCK() (0LL)
#endif

namespace mpc {

// IEEE binary16 input records (mpc_batch_solve_f16: BASELINE configs[4] stores the state in fp16).  The device converts with the hardware
// instruction; the host emulation (g++ has no _Float16 on x86 before GCC 12) decodes the bits.
#if defined(__HIPCC__)
typedef _Float16 mpc_half;
MPC_HD double half_to_double(mpc_half v) { return (double)v; }
#else
typedef unsigned short mpc_half;
inline double half_to_double(mpc_half b) {
  const int sgn = b >> 15, ex = (b >> 10) & 31, man = b & 1023;
  double v;
  if (ex == 0) v = ldexp((double)man, -24);
  else if (ex == 31) v = man ? NAN : INFINITY;
  else v = ldexp((double)(man | 1024), ex - 25);
  return sgn ? -v : v;
}
#endif

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
constexpr int kStSolved = 1, kStSolvedInaccurate = 2, kStPrimInfInaccurate = 3, kStDualInfInaccurate = 4, kStMaxIter = -2, kStPrimInf = -3, kStDualInf = -4,
              kStNonCvx = -7, kStUnsolved = -10;

// positions of the structural nonzeros of the 5 x 3 cone block {-1 0 mu; 1 0 mu; 0 -1 mu; 0 1 mu; 0 0 1} (mpc_osqp.cc:437-447), row-major
constexpr int kAsPos[9] = {0, 2, 3, 5, 7, 8, 10, 11, 14};

template <int H>
struct Cfg {
  static constexpr int N = 12 * H, M = 20 * H, NF = 4 * H;
  static constexpr int TS = 6;                           // register tile side (2 feet)
  static constexpr int G = N / TS;                       // tile grid G x G, lower triangle stored
  static constexpr int MT = G * (G + 1) / 2;             // lower-triangle tiles
  // Dense-P tiles per thread of the prep kernel (the only user of the dense 2H x 2H tile grid: the Ruiz norms).  Up to h = 16
  // every thread holds one tile.  The longest horizon has 820 tiles: two per thread make it a 448-thread workgroup, seven waves at the
  // 256-register cap (34 registers spilled, outside the pass loop): measured 3 % faster than four per thread (256 threads, one wave per
  // SIMD with the full 512-register budget: 1.163 -> 1.127 ms per 4096 robots, profiles/r04_ab_prep_h20.txt); three per thread spill in the
  // pass loop (3.1 ms).
  static constexpr int NT = H > 16 ? MPC_NT_H20 : (H > 12 ? MPC_NT_H16 : 1);
  static constexpr int MTH = (MT + NT - 1) / NT;         // threads that hold tiles
  static constexpr int TE = TS * TS;                     // tile elements per thread
  static constexpr int T0 = (((MTH > N ? MTH : N) + 63) / 64) * 64;
  static constexpr int T = T0 < 128 ? 128 : T0;          // prep-kernel workgroup: a thread per tile slot and per variable (the assembly's set-up phases use threads up to 104)
  static constexpr int MR = (M + T - 1) / T;             // constraint rows per thread (Scaler::for_rows): 1, or 2 at h = 20
  static constexpr int IN_LEN = 56 + 4 * H;
  static_assert(T <= 1024, "workgroup too large");
  static constexpr int NP = N + 2;                       // row stride of the prep kernel's part[] (doubles)
  // LDS diet of the short horizon (q is re-read from the QP record instead of a copy in LDS); the long horizons run one or two
  // robots per CU whatever their LDS size and keep the copy.
  static constexpr bool kQInLds = H > 12;
  // Live-range split points of the tile registers (Scaler::pin_tiles): bits 5 / 6 before / after the Ruiz passes, 7 inside them.
#ifdef MPC_PIN_MASK
  static constexpr int kPinMask = MPC_PIN_MASK;
#else
  static constexpr int kPinMask = H > 16 ? 3 : 17;
#endif
  // The QP record the assembly kernel hands to the scaling and solve kernels (doubles per robot):
  //   q[N] bnd[3 NF] cone[15] pad | B6[6 x 12] th1[6 x 6] th2[6] pad2   (the wrench-space description of P, mpc_wrench.h)
  // bnd: per foot l of row 4 and u of rows 0-3 / of row 4 -- l of rows 0-3 is 0 and the four u are equal (mpc_osqp.cc:449-477), so
  // three numbers stand for the foot's ten bounds, and E times them is bit for bit what scaling.c:152-153 computes.
  static constexpr int QP_Q = 0, QP_BND = N, QP_CONE = N + 3 * NF, QP_B6 = QP_CONE + 16, QP_TH1 = QP_B6 + 72,
                       QP_TH2 = QP_TH1 + 36, QP_LEN = QP_TH2 + 8;
  // The scale record the scaling kernel hands to the solve kernel: D[N] E[M] q_s[N] c 1/c | job[2]
  // (the scaled cone block is not in it: every foot's block is the QP record's 5 x 3 cone block times E of its rows and D of its columns, nine
  // structural nonzeros (kAsPos) that the solve kernel forms as (a E) D -- scaled_cone_entry below; the ten passes of scaling.c round (a e) d ten times,
  // a difference of a few ulp.  The scaled bounds are E times the QP record's;
  // job: primal / dual residual of the ADMM part's result, from the ADMM job of a solve to its polish job, mpc_wrench.h)
  static constexpr int SC_D = 0, SC_E = N, SC_QS = N + M, SC_C = 2 * N + M, SC_JOB = SC_C + 2, SC_LEN = SC_C + 4;
  // structural nonzero k (0 .. 8) of foot f's scaled cone block
  static MPC_HD double scaled_cone_entry(const double *qp, const double *sc, int f, int k) {
    const int pos = kAsPos[k];
    return (qp[QP_CONE + pos] * sc[SC_E + 5 * f + pos / 3]) * sc[SC_D + 3 * f + pos % 3];
  }
  // The same two records as mpc_batch_get_qp / mpc_batch_get_scale hand them out (include/mpc_batch.h: every bound, the dense cone
  // block): q[N] l[M] u[M] cone[15] pad | B6 th1 th2 pad2   and   D[N] E[M] q_s[N] A_s[15 NF] l_s[M] u_s[M] c 1/c | job[2]
  static constexpr int XQP_L = N, XQP_U = N + M, XQP_CONE = N + 2 * M, XQP_LEN = XQP_CONE + 16 + 72 + 36 + 8;
  static constexpr int XSC_AS = 2 * N + M, XSC_LS = XSC_AS + 15 * NF, XSC_US = XSC_LS + M, XSC_C = XSC_US + M, XSC_LEN = XSC_C + 4;
  // ---- wrench grid of the solve kernel (mpc_wrench.h): the 6 H x 6 H core matrix as H x H tiles of 6 x 6, one per thread
  static constexpr int NW = 6 * H, GW = H, MTW = H * (H + 1) / 2;
  // Foot lanes of the solve kernel: threads FOOT0 .. FOOT0 + NF - 1 (a multiple of four: the feet of a step are a hardware quad).
  static constexpr int FOOT0 = MPC_FOOT0(H, MTW);
  static constexpr int TWMIN = FOOT0 + NF > MTW ? FOOT0 + NF : MTW;
  static constexpr int TW = (((TWMIN > NW ? TWMIN : NW) + 63) / 64) * 64;   // solve-kernel workgroup: 64 (h <= 10), 128 (h = 12), 256 (h = 16, 20)
  static constexpr int NPW = NW + 2;                     // row stride of the solve kernel's part[]
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
// LDS of the scaling kernel
template <int H>
struct ScaleShared {
  using C = Cfg<H>;
  MPC_V q[C::kQInLds ? C::N : 2];                       // unscaled q, unless it is re-read from the HBM record
  MPC_V qs[C::N]; MPC_V As[C::NF * 15];   // scaled problem (the bounds are scaled once, at the end: E times the QP record's)
  MPC_V D[C::N]; MPC_V E[C::M];
  double c, cinv, ctmp;
  int first;
  MPC_V xt[C::N];                                       // q of the previous call (osqp_update_P_A scales with it)
  MPC_V cone[16];
  MPC_V dt_[C::N]; MPC_V et_[C::M];                     // Ruiz pass temporaries
  MPC_V part[C::NP * C::G];                             // [slot][row] partial maxima of the tile row norms
};
// LDS of the assembly kernel (one workgroup per robot, its own launch: see Assembler)
template <int H>
struct AsmShared {
  using C = Cfg<H>;
  MPC_V in[C::IN_LEN];
  static constexpr int XK = 13 * H > 74 ? 13 * H : 74;    // (xk / sdiff double as the set-up scratch of Assembler::run: 73 / 62 slots)
  MPC_V x0[13]; MPC_V xref[13 * H]; MPC_V sdiff[XK]; MPC_V xk[XK];
  MPC_V a_dt[169]; MPC_V b_dt[156]; MPC_V a_exp[169]; MPC_V b_exp[156];
  MPC_V wanb[H * 156];                                  // diag(w) A^k B
  MPC_V B6[72]; MPC_V th1[36]; MPC_V th2[8];            // wrench description of P (mpc_wrench.h), also written to the QP record
};
// LDS of the prep kernel (assembly, then Ruiz scaling, of one robot): the assembly arrays are dead when the scaling starts
template <int H>
struct PrepShared {
  union {
    AsmShared<H> as;
    ScaleShared<H> sc;
  };
  MPC_V u12[288];                                       // U1, U2: what the scaling part needs of the assembly (see Assembler::run)
};
#undef MPC_V

template <int H>
struct Thread {
  using C = Cfg<H>;
  int tid;                        // thread id
  int ti[C::NT], tj[C::NT];       // my tiles: tile row / tile column (tj <= ti); tile u has index tid + u * MTH
  bool mact[C::NT], dia[C::NT];   // tile u exists; it sits on the diagonal
  double Mx[C::NT * C::TE];       // my tiles of the dense P, row-major 6 x 6 each
  static constexpr int VPL = (C::N + 63) / 64;   // variables per lane of the first wavefront (Scaler::fold_phase): j = tid + 64 v
  double rm[VPL];                 // their column maxima
  double red[2];                  // wavefront reduction operands: a sum and a maximum
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

// index of the lowest set bit (v != 0)
MPC_HD int mpc_ffs64(unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ffsll((long long)v) - 1;
#else
  return __builtin_ctzll(v);
#endif
}

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
// ------------------------------------------------------------------------------------------------
// Diagnostic builds (-DMPC_PROFILE_SUB=<section>) split one section into slots 9..13 of the profile record:
// 1 = dynamics, 2 = one scaling pass, 3 = polish set-up, 4 = A dt / B dt set-up, 5 = the four phases of an ADMM iteration.
#ifndef MPC_PROFILE_SUB
#define MPC_PROFILE_SUB 0
#endif
#define MPC_SUBLAP(sec, k) do { if (MPC_PROFILE_SUB == (sec)) lap(k); } while (0)

// ============================================================================================================
// 1. Assembly (mpc_osqp.cc:606-688), first part of the prep kernel: one workgroup per robot builds q, the bounds, the cone
// block and the wrench description of P (B6, th1, th2) -> QP record in HBM, and the two 12 x 12 tables U1, U2 -> LDS.
// ============================================================================================================
template <int H, class Exec>
struct Assembler {
  using C = Cfg<H>;
  using Th = Thread<H>;
  static constexpr int N = C::N, M = C::M, NF = C::NF, T = C::T, TS = C::TS, TE = C::TE;

  Exec &ex;
  AsmShared<H> &s;
  const RobotModel &mdl;
  const float *in;     // [IN_LEN]   the input record as float32 ...
  const double *in64;  // [IN_LEN]   ... or as float64 (the reference's std::vector<double> arguments, mpc_osqp.cc:578-591) ...
  const mpc_half *in16;  // [IN_LEN] ... or as float16 (BASELINE configs[4]: the state stored in fp16); exactly one of the three is set.  Every
                       //            value is widened to double on load: the arithmetic is the same fp64 whatever the storage type
  double *u12;         // [288] LDS  out: U1 = B6^T th1 B6, U2 = B6^T diag(th2) B6 (12 x 12 each): P = Sigma2 (x) U1 + N (x) U2 + alpha I
  double *qp;          // [QP_LEN]   out: q, l, u, cone
  long long *prof;     // [kProfLen] slots 1 (dynamics) and 2 (q + P) are written here (may be null)
  long long tc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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

  // ================================ 1. assembly =================================================
  MPC_HD double in_at(int i) const { return in64 ? in64[i] : (in16 ? half_to_double(in16[i]) : (double)in[i]); }
  MPC_HD void run() {
    tlast = MPC_CLOCK();
    // ---- A dt, B dt (mpc_osqp.cc:299-336, 606-617, 661-673).  The chain of 3 x 3 products behind them (rotations in both of the
    // reference's conventions, the world-frame feet, the world inertia) is ~250 dependent-free FMAs: ONE thread carries it through
    // registers (static indices) while the others set up x0, x_ref and the bounds -- four barrier phases of one-entry-per-thread
    // products before (a phase costs ~1.5 k cycles of barrier and LDS latency here, whatever it computes).
    //   xk: [0..7) cos / sin of roll, pitch, yaw and tan(pitch); [43..55) fw; [64..73) iw        sdiff: [53..62) I^-1 (body)
    double *const trig = s.xk, *const fw = s.xk + 43, *const iw = s.xk + 64;
    double *const iib = s.sdiff + 53;
    static_assert(AsmShared<H>::XK >= 73, "xk / sdiff too small for the set-up scratch");
    ex.par([&](Th &t) {
      for (int i = t.tid; i < C::IN_LEN; i += T) s.in[i] = in_at(i);
      for (int i = t.tid; i < 169; i += T) s.a_dt[i] = 0;
      for (int i = t.tid; i < 156; i += T) s.b_dt[i] = 0;
      for (int i = t.tid; i < 72; i += T) { qp[C::QP_B6 + i] = 0; s.B6[i] = 0; }
      if (t.tid >= 64 && t.tid < 67) {          // (one wave-front's worth of trigonometry, off the first wavefront)
        const int k = t.tid - 64;
        const double ang = in_at(IN_RPY + k);
        trig[2 * k] = cos(ang); trig[2 * k + 1] = sin(ang);
        if (k == 1) trig[6] = tan(ang);
      } else if (t.tid >= 96 && t.tid < 105) {
        iib[t.tid - 96] = mdl.inv_inertia[t.tid - 96];
      }
    });
    MPC_SUBLAP(4, 9);
    ex.par([&](Th &t) {
      if (t.tid == 0) {
        const double cr = trig[0], sr = trig[1], cp = trig[2], sp = trig[3], cy = trig[4], sy = trig[5];
        const double rx[9] = {1.0, 0.0, 0.0, 0.0, cr, -sr, 0.0, sr, cr};      // (entry (v, u) = +sin, (u, v) = -sin about axis x / y / z)
        const double ry[9] = {cp, 0.0, sp, 0.0, 1.0, 0.0, -sp, 0.0, cp};
        const double rz[9] = {cy, -sy, 0.0, sy, cy, 0.0, 0.0, 0.0, 1.0};
        double m1[9], m2[9], rxyz[9], rzyx[9], t2[9], ib[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) { m1[e] = mat3e(rx, ry, e); m2[e] = mat3e(rz, ry, e); ib[e] = iib[e]; }   // Rx Ry (feet, :606-609), Rz Ry (inertia, :283-291)
#pragma unroll
        for (int e = 0; e < 9; ++e) { rxyz[e] = mat3e(m1, rz, e); rzyx[e] = mat3e(m2, rx, e); }
        const double *fb = s.in + in_foot<H>();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 3; ++r) fw[3 * i + r] = rxyz[3 * r] * fb[3 * i] + rxyz[3 * r + 1] * fb[3 * i + 1] + rxyz[3 * r + 2] * fb[3 * i + 2];
#pragma unroll
        for (int e = 0; e < 9; ++e) t2[e] = mat3e(rzyx, ib, e);                 // rzyx I^-1 (:670)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) iw[3 * i + j] = t2[3 * i] * rzyx[3 * j] + t2[3 * i + 1] * rzyx[3 * j + 1] + t2[3 * i + 2] * rzyx[3 * j + 2];   // ... rzyx^T (:671)
      }
      // x0 (:630-633)
      if (t.tid >= 32 && t.tid < 45) {
        const int i = t.tid - 32;
        s.x0[i] = i < 3 ? s.in[IN_RPY + i] : i < 6 ? s.in[IN_POS + i - 3] : i < 9 ? s.in[IN_ANG + i - 6] : i < 12 ? s.in[IN_VEL + i - 9] : -kGravity;
      }
      // bounds (:449-477, 685-688, 720-721)
      for (int f = t.tid; f < NF; f += T) {
        const double cst = s.in[IN_CONTACT + f];
        const double fzmax = mdl.mass * kGravity * kMaxScale, fzmin = mdl.mass * kGravity * kMinScale;
        const double mu0 = s.in[in_fric<H>()];
        qp[C::QP_BND + 3 * f] = dmax(fzmin * cst, -kInfty);                  // l of row 4 (rows 0-3: max(0, -inf) = 0)
        qp[C::QP_BND + 3 * f + 1] = dmin((mu0 + 1) * fzmax * cst, kInfty);   // u of rows 0-3
        qp[C::QP_BND + 3 * f + 2] = dmin(fzmax * cst, kInfty);               // u of row 4
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
    MPC_SUBLAP(4, 12);
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
        qp[C::QP_B6 + r * 12 + 3 * i + c] = acc;          // wrench map B6 = [I_w^-1 [r_i]x ; I / m]  (mpc_wrench.h)
        s.B6[r * 12 + 3 * i + c] = acc;
      } else if (t.tid < 48) {   // B rows 9-11: I / m
        const int k = t.tid - 36, i = k / 3, r = k - 3 * i;
        s.b_dt[(9 + r) * 12 + 3 * i + r] = mdl.inv_mass * dt;
        qp[C::QP_B6 + (3 + r) * 12 + 3 * i + r] = mdl.inv_mass;
        s.B6[(3 + r) * 12 + 3 * i + r] = mdl.inv_mass;
      } else if (t.tid < 57) {   // A rows 0-2: omega -> rpy rates (:311-312): {cy/cp, sy/cp, 0; -sy, cy, 0; cy tp, sy tp, 1}
        const int e = t.tid - 48, r = e / 3, c = e - 3 * r;
        const double cp = trig[2], cy = trig[4], sy = trig[5], tp = trig[6];
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
    // reference add exact zeros elsewhere, so only the nonzero terms are formed (same order, same values).  (A_exp itself is never
    // needed: A_exp^k acts through the closed forms below.)
    // A^k B (:368-373) and the free response A^{i+1} x0, i < H-1 (:360-364; the last A_qp block stays 0).
    // A dt is nilpotent, so A_exp^k = exp(k A dt) = I + k A dt + k^2 (A dt)^2 / 2 exactly, and (A dt)^2 B_exp = 0
    // (its only column, 12, meets the zero row 12 of B_exp):  A_exp^k B_exp = B_exp + k U,  U = (A dt) B_exp,
    // which is nonzero in rows 0-5 only and meets B_exp in its rows 6-11 only -- where B_exp = B dt (rows 6-12 of A dt are zero up to
    // column 12): U needs no phase of its own.  All k are formed at once (the reference multiplies k times; the two
    // agree to rounding).  U -> the (unused) a_exp area, (A dt) x0 and (A dt)^2 x0 -> the first 26 slots of sdiff.
    double *const U = s.a_exp;
    ex.par([&](Th &t) {
      for (int kk = t.tid; kk < 156; kk += T) {
        const int r = kk / 12, c = kk - 12 * r;
        double acc = 0;
        if (r < 3) { for (int j = 6; j < 9; ++j) acc += s.a_dt[r * 13 + j] * s.b_dt[j * 12 + c]; }
        else if (r < 6) acc += s.a_dt[r * 13 + r + 6] * s.b_dt[(r + 6) * 12 + c];
        s.b_exp[kk] = s.b_dt[kk] + acc / 2;
      }
      for (int e2 = T - 1 - t.tid; e2 < 72 + 13; e2 += T) {     // (from the last thread down: the first ones carry b_exp and th1 / th2)
        if (e2 < 72) {
          const int e = e2, r = e / 12, c = e - 12 * r;
          double acc = 0;
          if (r < 3) { for (int j = 6; j < 9; ++j) acc += s.a_dt[r * 13 + j] * s.b_dt[j * 12 + c]; }
          else acc = s.a_dt[r * 13 + r + 6] * s.b_dt[(r + 6) * 12 + c];
          U[e] = acc;
        } else {
          const int r = e2 - 72;
          double a1 = 0, a2 = 0;
          if (r < 3) { for (int j = 6; j < 9; ++j) a1 += s.a_dt[r * 13 + j] * s.x0[j]; }
          else if (r < 6) { a1 = s.a_dt[r * 13 + r + 6] * s.x0[r + 6]; a2 = (s.a_dt[r * 13 + r + 6] * s.a_dt[(r + 6) * 13 + 12]) * s.x0[12]; }
          else if (r >= 9 && r < 12) a1 = s.a_dt[r * 13 + 12] * s.x0[12];
          s.sdiff[r] = a1; s.sdiff[13 + r] = a2;
        }
      }
      // The wrench-space description of P (mpc_wrench.h): A_exp^k B_exp = Gamma_k B6 with Gamma_k = [dt^2 (k + 1/2) That ; dt I6],
      // That = blockdiag(T_rpy, I3) (A dt rows 0-2 hold dt T_rpy), so that 2 Gamma_k^T Q Gamma_k' = (k + 1/2)(k' + 1/2) th1 + diag(th2):
      //   th1 = 2 dt^4 blockdiag(T^T diag(w0..2) T, diag(w3..5)),  th2 = 2 dt^2 (w6 .. w11)
      if (t.tid < 36) {
        const int a = t.tid / 6, b = t.tid - 6 * a;
        const double dt = mdl.dt, d2 = 2.0 * dt * dt;
        double v = 0;
        if (a < 3 && b < 3) {
          for (int r = 0; r < 3; ++r) v += s.in[IN_W + r] * (s.a_dt[r * 13 + 6 + a] * s.a_dt[r * 13 + 6 + b]);
          v *= d2;
        } else if (a == b) v = (d2 * dt * dt) * s.in[IN_W + a];
        qp[C::QP_TH1 + t.tid] = v;
        s.th1[t.tid] = v;
      } else if (t.tid < 42) {
        const double v = (2.0 * mdl.dt * mdl.dt) * s.in[IN_W + 6 + t.tid - 36];
        qp[C::QP_TH2 + t.tid - 36] = v;
        s.th2[t.tid - 36] = v;
      }
    });
    MPC_SUBLAP(1, 10);
    ex.par([&](Th &t) {
      for (int e = t.tid; e < H * 156; e += T) {
        const int k = e / 156, rc = e - 156 * k, r = rc / 12;
        const double v = r < 6 ? s.b_exp[rc] + (double)k * U[rc] : s.b_exp[rc];
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
    // q (:683); and the two 12 x 12 tables from which every thread of the scaling part builds its own tile of P
    // (P = 2 B_qp^T Q B_qp + alpha I = Sigma2 (x) U1 + N (x) U2 + alpha I in the wrench form, mpc_wrench.h; the reference's block
    // recursion :387-434 gives the same numbers to rounding, and only the Ruiz norms read them)
    ex.par([&](Th &t) {
      for (int jc = t.tid; jc < N; jc += T) {
        const int j = jc / 12, c = jc - 12 * j;
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
        qp[C::QP_Q + jc] = 2 * acc;
      }
      for (int e = t.tid; e < 288; e += T) {
        const int which = e / 144, ab = e - 144 * which, a = ab / 12, b = ab - 12 * a;
        double acc = 0;
        if (which == 0) {
          for (int p = 0; p < 6; ++p) {
            double row = 0;
            for (int q = 0; q < 6; ++q) row += s.th1[6 * p + q] * s.B6[12 * q + b];
            acc += s.B6[12 * p + a] * row;
          }
        } else {
          for (int p = 0; p < 6; ++p) acc += s.B6[12 * p + a] * (s.th2[p] * s.B6[12 * p + b]);
        }
        u12[e] = acc;
      }
    });
    lap(2);
    if (prof) {
      ex.par([&](Th &t) {
        if (t.tid == 0) {
          prof[1] = tc[1]; prof[2] = tc[2];
          if (MPC_PROFILE_SUB == 1 || MPC_PROFILE_SUB == 4) for (int k = 9; k <= 13; ++k) prof[k] = tc[k];
        }
      });
    }
  }
};


// ============================================================================================================
// 2. Ruiz equilibration + cost scaling (scaling.c:44-156), second part of the prep kernel: the workgroup holds the dense
// P as 6 x 6 register tiles built from U1, U2 (the only consumer of the dense matrix: the solve kernel works in the wrench
// space, mpc_wrench.h; P never exists in memory), runs the ten passes and leaves the scaled problem vectors in the scale record.
// ============================================================================================================
template <int H, class Exec>
struct Scaler {
  using C = Cfg<H>;
  using Th = Thread<H>;
  using Sh = ScaleShared<H>;
  static constexpr int N = C::N, M = C::M, NF = C::NF, T = C::T, TS = C::TS, G = C::G, TE = C::TE, NP = C::NP;

  Exec &ex;
  Sh &s;
  const double *state; // [state_len<H>()]  warm-start record (read: q of the previous call, cold flag)
  const double *u12;   // [288] LDS  U1, U2 of the assembly part
  double alpha;
  const double *qp;    // [QP_LEN]  q, l, u, cone from the assembly kernel
  double *sc;          // [SC_LEN]  out: D, E, q_s, c, 1/c
  long long *prof = nullptr;   // [kProfLen] slots 3 (tile build + first norms), 4 (the ten passes), 5 (record store) under MPC_SECTION_PROFILE
  long long tc[6] = {0, 0, 0, 0, 0, 0};
  long long tlast = 0;
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
#ifndef MPC_SECTION_PROFILE
  MPC_HD void lap(int) {}
#else
  MPC_HD void lap(int k) { const long long now = MPC_CLOCK(); tc[k] += now - tlast; tlast = now; }
#endif
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
  // my tile of P from the two 12 x 12 tables: tile (ti, tj) = feet pair ti & 1 of step ti / 2 against feet pair tj & 1 of step tj / 2,
  //   P[(s, a), (s', b)] = s2(s, s') U1[a][b] + (H - s) U2[a][b] + [same entry] alpha,   s >= s',
  //   s2 = sum_{i < m} (i + 1/2)(i + d + 1/2) = m (4 m^2 - 1) / 12 + d m^2 / 2,  m = H - s,  d = s - s'
  MPC_HD void build_tile(Tv &t) {
    const int sr = t.ti >> 1, sc_ = t.tj >> 1, rh = t.ti & 1, ch = t.tj & 1;
    const double m = (double)(H - sr), d = (double)(sr - sc_);
    const double s2 = m * (4.0 * m * m - 1.0) / 12.0 + d * (m * m) * 0.5;
    const double *u1 = u12 + (6 * rh) * 12 + 6 * ch, *u2 = u1 + 144;
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
      for (int b = 0; b < TS; ++b) {
        double v = s2 * u1[12 * a + b] + m * u2[12 * a + b];
        if (t.dia && a == b) v += alpha;
        t.Mx[a * TS + b] = v;
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
  // ================================ 2. scaling (scaling.c:44-156) ===============================
  // One Ruiz pass is three phases.  P itself stays UNSCALED in the tile registers for all passes: a pass only
  // needs the row norms of c D P D, which are c D_i max_j(|P_ij| D_j) with the cumulative D and c (tile_rownorms); D, c, q, A, E are updated incrementally as in
  // scaling.c, and c D P D is formed once after the last pass.  The cost scale c_temp of pass k is folded in
  // lazily at pass k + 1.
  // The reduction half of a Ruiz pass, on the first wavefront alone (lane l owns the variables l, l + 64, ...): the column maxima
  // of D P D are folded from the tiles' partials once and serve both uses --
  //   COST: the pass's cost scaling (scaling.c:108-139): c_temp = 1 / max(mean_j |c D P D column j|_inf, |q|_inf), a sum and a
  //         maximum over all variables, reduced inside the wavefront (DPP row operations + four readlanes: no LDS, no barrier);
  //   DUPD: the next pass's column scales from |column|_inf of [c c_temp D P D ; A] (scaling.c:65-84), D <- D_temp D;
  //         without DUPD (after the last pass) q and c take the pending cost scale here.
  // One barrier phase; the other wavefronts wait.  (It replaces two phases in which every thread of the first two wavefronts
  // re-folded 80 per-foot partials for the same scalar.)
  template <bool COST, bool DUPD>
  MPC_HD void fold_phase() {
    ex.seq([&](Th &t) {
      if (t.tid < 64) {
        double sum = 0, qm = 0;
#pragma unroll
        for (int v = 0; v < Th::VPL; ++v) {
          const int j = t.tid + 64 * v;
          if (j < N) {
            const double m = max_parts(s, j);
            t.rm[v] = m;
            if constexpr (COST) { sum += m; qm = raw_max(qm, fabs(s.qs[j])); }
          }
        }
        t.red[0] = sum; t.red[1] = qm;
      }
    });
    if constexpr (COST) ex.wave_sum_max([](Th &t) { return t.red; });
    ex.par([&](Th &t) {
      if (t.tid < 64) {
        double ct = 1.0;
        if constexpr (COST) {
          const double mean = (s.c * t.red[0]) * (1.0 / N);
          ct = fast_recip(limit_scaling(fmax(mean, limit_scaling(t.red[1]))));
        }
#pragma unroll
        for (int v = 0; v < Th::VPL; ++v) {
          const int j = t.tid + 64 * v;
          if (j < N) {
            if constexpr (DUPD) {
              const int f = j / 3, c = j - 3 * f;
              double mx = (s.c * ct) * t.rm[v];
#pragma unroll
              for (int r = 0; r < 5; ++r) mx = fmax(mx, fabs(s.As[15 * f + 3 * r + c]));
              const double d = fast_rsqrt(limit_scaling(mx));
              s.dt_[j] = d;
              s.D[j] *= d;
            } else {
              s.qs[j] *= ct;
            }
          }
        }
        if (t.tid == 0) s.ctmp = DUPD ? ct : s.c * ct;   // the pending cost scale; after the last pass the final c
      }
    });
  }
  static MPC_HD double row_scale3(double a0, double a1, double a2) {   // 1 / sqrt(|row|_inf) of a 3-entry row of A
    return fast_rsqrt(limit_scaling(fmax(fmax(fabs(a0), fabs(a1)), fabs(a2))));
  }
  MPC_HD void scale() {
    tlast = MPC_CLOCK();
    ex.par([&](Th &t) {
      for_tiles(t, [&](Tv &v, int) { build_tile(v); tile_rownorms(v, nullptr); });
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
    fold_phase<false, true>();
    for (int it = 0; it < kScalingIters; ++it) {
      pin_tiles(7);
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
      if (it + 1 < kScalingIters) fold_phase<true, true>();
      else fold_phase<true, false>();
    }
    lap(4);
    pin_tiles(6);
    ex.par([&](Th &t) {   // the scale record
      const double cf = s.ctmp;
      if (t.tid == 0) { sc[C::SC_C] = cf; sc[C::SC_C + 1] = 1.0 / cf; }
      if (t.tid < N) {
        const double qv = s.first ? s.qs[t.tid] : (s.D[t.tid] * q_at(t.tid)) * cf;      // osqp_update_lin_cost (osqp.c:765-770)
        sc[C::SC_QS + t.tid] = qv;
        sc[C::SC_D + t.tid] = s.D[t.tid];
      }
      for_rows(t, [&](int i) { sc[C::SC_E + i] = s.E[i]; });
    });
    lap(5);
    if (prof) {
      ex.par([&](Th &t) { if (t.tid == 0) { prof[3] = tc[3]; prof[4] = tc[4]; prof[5] = tc[5]; } });
    }
  }

  // the QP record of the assembly kernel + what the warm-start record says about the previous call
  MPC_HD void load() {
    ex.par([&](Th &t) {
      for (int i = t.tid; i < N; i += T) { if constexpr (C::kQInLds) s.q[i] = qp[C::QP_Q + i]; s.xt[i] = state[N + 2 * M + i]; /* q_old */ }
      if (t.tid < 15) s.cone[t.tid] = qp[C::QP_CONE + t.tid];
      if (t.tid == 0) s.first = state[2 * N + 2 * M + 1] == 0.0;
    });
  }
  MPC_HD void run() {
    load();
    scale();
  }
};

}  // namespace mpc
