// tests/emu/emu.cpp -- HOST EMULATION of the device solver (TEST ONLY; never used by the product).
// Compiles rl-mpc-locomotion_amd/csrc/mpc_core.h with g++ and runs each barrier-separated phase for
// all emulated threads sequentially (forward or reverse order: identical results are required, which
// exposes intra-phase races).  Lets the CPU test-suite check the kernel's algorithm against the
// oracle without a GPU.
#define MPC_EMU_DEBUG 1
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "controller.h"
#include "mpc_core.h"
#include "mpc_model.h"
#include "mpc_wrench.h"

using namespace mpc;

// every planning horizon the library ships (rl-mpc-locomotion_amd/csrc/mpc_horizon.h)
#define EMU_HORIZONS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)

// What a freshly launched workgroup finds in its registers and its LDS is undefined.  emu_set_poison(1) fills the emulated thread
// state and shared memory with 0xFF bytes (NaN doubles, -1 integers) instead of zeros before a run -- only the tile registers are
// zeroed, as the kernels do -- so that a read of anything the kernel has not written shows up as a changed (NaN) result.
static int g_poison = 0;
static inline int fill_byte() { return g_poison ? 0xFF : 0; }
static int g_exact_route = 0;   // emu_set_exact_route: 0 = active set, then ADMM if it fails (the product); 1 = ADMM route only; 2 = active set only
static int g_split = 0;      // emu_set_split: the OSQP-mode solve as ADMM job + polish job (the persistent kernel's path)
static int g_max_iter = 0;   // emu_set_max_iter: OSQP's max_iter setting (0 = the default, kMaxIter)
static int *g_seed = nullptr;   // emu_set_seed_buffer: [n][4 h] working-set seeds of the exact mode (mpc_batch's d_seed), or null
static int g_seed_stride = 0;

template <class TH, int NTHREADS>
struct HostExec {
  std::vector<TH> th;
  bool reverse;
  long phases = 0;
  explicit HostExec(bool rev) : th(NTHREADS), reverse(rev) {
    for (int i = 0; i < NTHREADS; ++i) {
      std::memset((void *)&th[i], fill_byte(), sizeof(TH));
      th[i].init(i);
      for (size_t e = 0; e < sizeof(th[i].Mx) / sizeof(double); ++e) th[i].Mx[e] = 0;   // (the kernels zero their tile registers)
    }
  }
  TH &first() { return th[0]; }   // a representative thread (after a workgroup-wide reduction every thread holds the same value)
  template <class F> void par(F &&f) {
    ++phases;
    if (!reverse) for (int i = 0; i < NTHREADS; ++i) f(th[i]);
    else for (int i = NTHREADS - 1; i >= 0; --i) f(th[i]);
  }
  template <class F> void seq(F &&f) { par(f); --phases; }   // (not an LDS hand-over on the device: not counted)
  // the device's DPP quad operations (mpc_batch.hip DeviceExec), lane by lane
  template <int N, class A> void quad_allsum(A &&acc) {
    for (int q = 0; q + 3 < NTHREADS; q += 4)
      for (int i = 0; i < N; ++i) {
        const double v = (acc(th[q])[i] + acc(th[q + 1])[i]) + (acc(th[q + 2])[i] + acc(th[q + 3])[i]);
        for (int j = 0; j < 4; ++j) acc(th[q + j])[i] = v;
      }
  }
  template <class S, class D> void quad_scatter6(S &&src, D &&dst) {   // (the device's association: own pair first, then the other pair)
    for (int q = 0; q + 3 < NTHREADS; q += 4) {
      double v[6][4];
      for (int r = 0; r < 6; ++r)
        for (int j = 0; j < 4; ++j) v[r][j] = (src(th[q + j])[r] + src(th[q + (j ^ 1)])[r]) + (src(th[q + (j ^ 2)])[r] + src(th[q + (j ^ 3)])[r]);
      for (int j = 0; j < 4; ++j) { dst(th[q + j])[0] = v[j][j]; dst(th[q + j])[1] = v[4 + (j & 1)][j]; }
    }
  }
  // the device's wavefront reduction (same association order: quad, half row, row, then the four rows)
  template <class A> void wave_sum_max(A &&acc) {
    for (int w = 0; w + 63 < NTHREADS; w += 64) {
      double rs[4], rmx[4];
      for (int r = 0; r < 4; ++r) {
        double hs[2], hm[2];
        for (int hh = 0; hh < 2; ++hh) {
          double qs[2], qm[2];
          for (int q = 0; q < 2; ++q) {
            const int b = w + 16 * r + 8 * hh + 4 * q;
            qs[q] = (acc(th[b])[0] + acc(th[b + 1])[0]) + (acc(th[b + 2])[0] + acc(th[b + 3])[0]);
            qm[q] = std::fmax(std::fmax(acc(th[b])[1], acc(th[b + 1])[1]), std::fmax(acc(th[b + 2])[1], acc(th[b + 3])[1]));
          }
          hs[hh] = qs[0] + qs[1]; hm[hh] = std::fmax(qm[0], qm[1]);
        }
        rs[r] = hs[0] + hs[1]; rmx[r] = std::fmax(hm[0], hm[1]);
      }
      const double sum = (rs[0] + rs[1]) + (rs[2] + rs[3]), mx = std::fmax(std::fmax(rmx[0], rmx[1]), std::fmax(rmx[2], rmx[3]));
      for (int i = 0; i < 64; ++i) { acc(th[w + i])[0] = sum; acc(th[w + i])[1] = mx; }
    }
  }
  // workgroup-wide argmax / sum (the device: DPP + readlane inside one wavefront, LDS scratch + barriers across several)
  template <class V, class I> void wg_argmax(V &&val, I &&idx, double *) {
    double m = val(th[0])[0];
    int ml = 0;
    for (int i = 1; i < NTHREADS; ++i) if (val(th[i])[0] > m) { m = val(th[i])[0]; ml = i; }
    for (int i = 0; i < NTHREADS; ++i) { val(th[i])[0] = m; idx(th[i]) = ml; }
  }
  template <class V> void wg_sum(V &&val, double *) {
    // (the device's association order: within a wavefront quad, half row, row, four rows; then the wavefronts in order)
    double tot = 0;
    for (int w = 0; w < NTHREADS; w += 64) {
      double rs[4];
      for (int r = 0; r < 4; ++r) {
        double hs[2];
        for (int hh = 0; hh < 2; ++hh) {
          double qs[2];
          for (int q = 0; q < 2; ++q) {
            const int b = w + 16 * r + 8 * hh + 4 * q;
            qs[q] = (val(th[b])[0] + val(th[b + 1])[0]) + (val(th[b + 2])[0] + val(th[b + 3])[0]);
          }
          hs[hh] = qs[0] + qs[1];
        }
        rs[r] = hs[0] + hs[1];
      }
      const double ws = (rs[0] + rs[1]) + (rs[2] + rs[3]);
      tot = w == 0 ? ws : tot + ws;
    }
    for (int i = 0; i < NTHREADS; ++i) val(th[i])[0] = tot;
  }
  // the device's v_mfma_f64_16x16x4_f64 (mpc_device.h mfma16), wavefront by wavefront: lane l holds A[l & 15][l >> 4], B[l >> 4][l & 15] and the
  // result elements (row (l >> 4) + 4 r, column l & 15), r = 0 .. 3; the sum over k in the instruction's order (k = 0 first, fused multiply-adds)
  template <class FA, class FB, class FC> void mfma16(FA &&fa, FB &&fb, FC &&fc) {
    for (int w = 0; w + 63 < NTHREADS; w += 64) {
      double A[16][4], B[4][16];
      for (int l = 0; l < 64; ++l) { A[l & 15][l >> 4] = fa(th[w + l]); B[l >> 4][l & 15] = fb(th[w + l]); }
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          double acc = fc(th[w + l])[r];
          for (int k = 0; k < 4; ++k) acc = std::fma(A[(l >> 4) + 4 * r][k], B[k][l & 15], acc);
          fc(th[w + l])[r] = acc;
        }
    }
  }
  template <class S, class D> void quad_gather6(S &&src, D &&dst) {
    for (int q = 0; q + 3 < NTHREADS; q += 4) {
      double v[6];
      for (int r = 0; r < 6; ++r) v[r] = src(th[q + (r & 3)])[r >> 2];
      for (int j = 0; j < 4; ++j) for (int r = 0; r < 6; ++r) dst(th[q + j])[r] = v[r];
    }
  }
};

// prep kernel (assembly + Ruiz scaling) of one robot: QP record and scale record
template <int H>
static long prep_one(const RobotModel &mdl, const float *in, const double *state, double *qp, double *sc, bool reverse) {
  using C = Cfg<H>;
  PrepShared<H> *ps = new PrepShared<H>();
  std::memset((void *)ps, fill_byte(), sizeof(PrepShared<H>));
  using Ex = HostExec<Thread<H>, C::T>;
  Ex ex(reverse);
  Assembler<H, Ex> am{ex, ps->as, mdl, in, nullptr, nullptr, ps->u12, qp, nullptr};
  am.run();
  Scaler<H, Ex> sk{ex, ps->sc, state, ps->u12, mdl.alpha, qp, sc};
  sk.run();
  const long ph = ex.phases;
  delete ps;
  return ph;
}
// prep kernel -> solve kernel of one robot
template <int H>
static void solve_one(const RobotModel &mdl, const float *in, double *state, double *forces, int *info, bool reverse, long *phases, double *dbg = nullptr, bool exact = false,
                      int *seed = nullptr) {
  using C = Cfg<H>;
  std::vector<double> qp(C::QP_LEN, 0.0), sc(C::SC_LEN, 0.0);
  long ph = prep_one<H>(mdl, in, state, qp.data(), sc.data(), reverse);
  {
    Shared<H> *sh = new Shared<H>();
    std::memset((void *)sh, fill_byte(), sizeof(Shared<H>));
    using Ex = HostExec<WThread<H>, C::TW>;
    Ex ex(reverse);
    Solver<H, Ex> sv{ex, *sh, mdl, state, qp.data(), sc.data(), forces, info, nullptr};
    sv.dbg = dbg;
    if (g_max_iter > 0) sv.max_iter = g_max_iter;
    if (exact) {   // the exact-optimum mode (mpc_batch_set_solver); the caller clears the state record
      GiShared<H> *gs = new GiShared<H>();
      std::memset((void *)gs, fill_byte(), sizeof(GiShared<H>));
      sv.exact();
      sv.gi = gs;
      sv.seedrec = seed;          // the working set of this robot's previous call (emu_set_seed_buffer), or null: start empty
      const bool ok = g_exact_route == 1 ? false : sv.run_active_set();          // first launch: the active-set method
      if (getenv("EMU_GI_TRACE")) fprintf(stderr, "active_set ok=%d passes=%d adds=%d drops=%d conv=%d fail=%d hi=%d seed_k=%d seeded=%d\n", (int)ok, gs->passes, gs->adds, gs->drops, gs->converged, gs->fail, gs->hi, seed ? gs->seed_k : -1, seed ? gs->seeded : -1);
      ph += ex.phases;
      delete gs;
      if (!ok && g_exact_route != 2) {   // second launch: the ADMM route on a fresh workgroup
        Shared<H> *sh2 = new Shared<H>();
        std::memset((void *)sh2, fill_byte(), sizeof(Shared<H>));
        Ex ex2(reverse);
        Solver<H, Ex> sv2{ex2, *sh2, mdl, state, qp.data(), sc.data(), forces, info, nullptr};
        sv2.exact();
        sv2.template run<true>();
        ph += ex2.phases;
        delete sh2;
      } else if (!ok) info[1] = -99;
      delete sh;
      if (phases) *phases = ph;
      return;
    }
    else if (g_split) {                                   // the two jobs of mpc_solve_jobs_kernel: the polish on a fresh workgroup
      sv.jobrec = sc.data() + C::SC_JOB;
      const bool pol = sv.admm_job();
      ph += ex.phases;
      if (pol) {
        Shared<H> *sh2 = new Shared<H>();
        std::memset((void *)sh2, fill_byte(), sizeof(Shared<H>));
        Ex ex2(reverse);
        Solver<H, Ex> sv2{ex2, *sh2, mdl, state, qp.data(), sc.data(), forces, info, nullptr};
        sv2.jobrec = sc.data() + C::SC_JOB;
        sv2.polish_job();
        ph += ex2.phases;
        delete sh2;
      }
      delete sh;
      if (phases) *phases = ph;
      return;
    }
    else sv.run();
    ph += ex.phases;
    delete sh;
  }
  if (phases) *phases = ph;
}

// ---- K-solve probe: x~ = K^{-1} b of the cold problem for a given rho, through the solver's own phases (factor + one product) ----
template <int H>
static void ksolve_one(const RobotModel &mdl, const float *in, double rho, const double *b, double *xt, double *sc_out) {
  using C = Cfg<H>;
  std::vector<double> qp(C::QP_LEN, 0.0), sc(C::SC_LEN, 0.0), state(state_len<H>(), 0.0), forces(C::N, 0.0);
  int info[kInfoLen];
  prep_one<H>(mdl, in, state.data(), qp.data(), sc.data(), false);
  Shared<H> *sh = new Shared<H>();
  std::memset((void *)sh, fill_byte(), sizeof(Shared<H>));
  using Ex = HostExec<WThread<H>, C::TW>;
  Ex ex(false);
  Solver<H, Ex> sv{ex, *sh, mdl, state.data(), qp.data(), sc.data(), forces.data(), info, nullptr};
  sv.load();
  sh->rho = rho;
  sv.set_rho_vec();
  sv.factor();
  ex.seq([&](WThread<H> &t) {
    if (t.foot) {
      double v[3];
      for (int c = 0; c < 3; ++c) t.b[c] = b[3 * t.fid + c];
      Solver<H, Ex>::sym3_mul(t.Si, t.b, v);
      for (int c = 0; c < 3; ++c) t.xt[c] = v[c];      // S^-1 b
      sv.put_g(t, t.b);                                // Gf holds G S^-1
    }
  });
  sv.template product<Solver<H, Ex>::kHeld>();
  ex.seq([&](WThread<H> &t) {
    if (t.foot) {
      double wy[3], tt[3], o[3];
      sv.get_g(t, wy);
      for (int c = 0; c < 3; ++c) o[c] = t.xt[c] - wy[c];      // x~ = S^-1 b - (G S^-1)^T y
      for (int c = 0; c < 3; ++c) xt[3 * t.fid + c] = o[c];
    }
  });
  for (int i = 0; i < C::SC_LEN; ++i) sc_out[i] = sc[i];
  delete sh;
}
extern "C" {
int emu_ksolve(int h, const double *model, double dt, double alpha, const float *in, double rho, const double *b, double *xt, double *sc_out) {
  RobotModel mdl = make_model(model[0], model + 1, dt, alpha);
  switch (h) {
#define EMU_CASE(HH) case HH: ksolve_one<HH>(mdl, in, rho, b, xt, sc_out); return 0;
    EMU_HORIZONS(EMU_CASE)
#undef EMU_CASE
  }
  return -1;
}
int emu_solve_debug(int h, const double *model, double dt, double alpha, const float *in, double *state, double *forces, int *info, double *dbg) {
  RobotModel mdl = make_model(model[0], model + 1, dt, alpha);
  switch (h) {
#define EMU_CASE(HH) case HH: solve_one<HH>(mdl, in, state, forces, info, false, nullptr, dbg); return 0;
    EMU_HORIZONS(EMU_CASE)
#undef EMU_CASE
  }
  return -1;
}
int emu_sc_len(int h) {
#define EMU_CASE(HH) if (h == HH) return Cfg<HH>::SC_LEN;
  EMU_HORIZONS(EMU_CASE)
#undef EMU_CASE
  return -1;
}
}

// ---- the controller as a stepping object: mpc_ctrl_create / mpc_ctrl_run / mpc_ctrl_reset of the library, robot by robot on the host ----
// (closed-loop tests drive it tick by tick: the next inputs depend on the torques it returned; bench.py --emulate runs its control flow on it)
struct EmuCtrl {
  int n, h;
  GaitTable gt;
  CtrlParams cp;
  std::vector<CtrlState> st;
  std::vector<RobotConst> rc;
  std::vector<RobotModel> mdl;
  std::vector<double> state, forces;
  std::vector<int> info;
  std::vector<float> rec;
};
template <int H>
static void emu_ctrl_tick(EmuCtrl &c, int r0, int r1, const float *dof, const float *body, const float *cmd, float *torques, int exact) {
  const int inlen = 56 + 4 * H, sl = 64 * H + 2;
  for (int r = r0; r < r1; ++r) {
    float est[kEstLen];
    estimator_update(body + (size_t)r * 13, c.st[r].normal, est);
    float *rec = c.rec.data() + (size_t)r * inlen;
    ctrl_pre(c.st[r], c.rc[r], c.gt, c.cp, dof + (size_t)r * 24, est, cmd + (size_t)r * 16, rec);
    int *info = c.info.data() + (size_t)r * kInfoLen;
    if (c.st[r].do_solve) {
      if (exact) std::fill(c.state.begin() + (size_t)r * sl, c.state.begin() + (size_t)(r + 1) * sl, 0.0);
      solve_one<H>(c.mdl[r], rec, c.state.data() + (size_t)r * sl, c.forces.data() + (size_t)r * 12 * H, info, false, nullptr, nullptr, exact != 0);
    }
    const int st_ = info[1];
    const int adopt = exact ? (st_ == kStSolved || st_ == kStSolvedInaccurate || st_ == kStMaxIter) : st_ == kStSolved;
    ctrl_post(c.st[r], c.rc[r], c.forces.data() + (size_t)r * 12 * H, adopt, torques + (size_t)r * 12);
  }
}
// ---- controller replay (ctrl_pre -> emulated solve -> ctrl_post) -----------------------------------------
// robot_table: [ntypes][25] doubles (rl_mpc_locomotion_amd/quadruped.py ROBOT_TABLE64 layout);
// gait_off / gait_dur: [8][4] ints for h segments; dof [T][n][24], est [T][n][18], cmd [T][n][16];
// out: torques [T][n][12], rec_out [T][n][56+4h] (solver records, zero rows when no solve), f_ff [T][n][12].
template <int H>
static int ctrl_replay_h(int n, int ticks, const double *robot_table, const int *robot_type, const int *gait_id,
                         const int *gait_off, const int *gait_dur, int flat_ground, double dt, int iters_between_mpc, double alpha,
                         const float *dof, const float *est, const float *cmd, float *torques, float *rec_out, float *fff_out) {
  GaitTable gt;
  gt.n_seg = H;
  for (int g = 0; g < kNumGaitIds; ++g) for (int j = 0; j < 4; ++j) { gt.offsets[g][j] = (float)gait_off[4 * g + j]; gt.durations[g][j] = (float)gait_dur[4 * g + j]; }
  CtrlParams cp{dt, iters_between_mpc, dt * iters_between_mpc, H, flat_ground};
  std::vector<CtrlState> st(n);
  std::vector<RobotConst> rc(n);
  std::vector<RobotModel> mdl(n);
  std::vector<double> state((size_t)n * (64 * H + 2), 0.0), forces((size_t)n * 12 * H, 0.0);
  for (int r = 0; r < n; ++r) {
    const double *row = robot_table + 25 * robot_type[r];
    rc[r].abad = row[0]; rc[r].hip = row[1]; rc[r].knee = row[2];
    for (int k = 0; k < 3; ++k) rc[r].hiploc[k] = (float)row[3 + k];
    rc[r].body_height = row[10]; rc[r].mu = (float)row[11];
    for (int k = 0; k < 13; ++k) rc[r].weights[k] = (float)row[12 + k];
    const double inertia9[9] = {row[7], 0, 0, 0, row[8], 0, 0, 0, row[9]};
    mdl[r] = make_model(row[6], inertia9, cp.dt_mpc, alpha);
    ctrl_init(st[r], rc[r], robot_type[r], gait_id[r]);
  }
  const int inlen = 56 + 4 * H;
  for (int t = 0; t < ticks; ++t)
    for (int r = 0; r < n; ++r) {
      const size_t idx = (size_t)t * n + r;
      float *rec = rec_out + idx * inlen;
      ctrl_pre(st[r], rc[r], gt, cp, dof + idx * 24, est + idx * kEstLen, cmd + idx * 16, rec);
      int info[kInfoLen] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (st[r].do_solve) solve_one<H>(mdl[r], rec, state.data() + (size_t)r * (64 * H + 2), forces.data() + (size_t)r * 12 * H, info, false, nullptr);
      ctrl_post(st[r], rc[r], forces.data() + (size_t)r * 12 * H, info[1] == kStSolved, torques + idx * 12);
      for (int k = 0; k < 12; ++k) fff_out[idx * 12 + k] = st[r].f_ff[k];
    }
  return 0;
}

extern "C" {
int emu_ctrl_replay(int h, int n, int ticks, const double *robot_table, const int *robot_type, const int *gait_id,
                    const int *gait_off, const int *gait_dur, int flat_ground, double dt, int iters_between_mpc, double alpha,
                    const float *dof, const float *est, const float *cmd, float *torques, float *rec_out, float *fff_out) {
  switch (h) {
#define EMU_CASE(HH) case HH: return ctrl_replay_h<HH>(n, ticks, robot_table, robot_type, gait_id, gait_off, gait_dur, flat_ground, dt, iters_between_mpc, alpha, dof, est, cmd, torques, rec_out, fff_out);
    EMU_HORIZONS(EMU_CASE)
#undef EMU_CASE
    default: return -1;
  }
}

void *emu_ctrl_open(int n, int h, const double *robot_table, const int *robot_type, const int *gait_id, const int *gait_off, const int *gait_dur,
                    int flat_ground, double dt, int iters_between_mpc, double alpha) {
  EmuCtrl *c = new EmuCtrl();
  c->n = n; c->h = h;
  c->gt.n_seg = h;
  for (int g = 0; g < kNumGaitIds; ++g) for (int j = 0; j < 4; ++j) { c->gt.offsets[g][j] = (float)gait_off[4 * g + j]; c->gt.durations[g][j] = (float)gait_dur[4 * g + j]; }
  c->cp = CtrlParams{dt, iters_between_mpc, dt * iters_between_mpc, h, flat_ground};
  c->st.resize(n); c->rc.resize(n); c->mdl.resize(n);
  c->state.assign((size_t)n * (64 * h + 2), 0.0); c->forces.assign((size_t)n * 12 * h, 0.0); c->info.assign((size_t)n * kInfoLen, 0);
  c->rec.assign((size_t)n * (56 + 4 * h), 0.f);
  for (int r = 0; r < n; ++r) {
    const double *row = robot_table + 25 * robot_type[r];
    RobotConst &k = c->rc[r];
    k.abad = row[0]; k.hip = row[1]; k.knee = row[2];
    for (int i = 0; i < 3; ++i) k.hiploc[i] = (float)row[3 + i];
    k.body_height = row[10]; k.mu = (float)row[11];
    for (int i = 0; i < 13; ++i) k.weights[i] = (float)row[12 + i];
    const double inertia9[9] = {row[7], 0, 0, 0, row[8], 0, 0, 0, row[9]};
    c->mdl[r] = make_model(row[6], inertia9, c->cp.dt_mpc, alpha);
    ctrl_init(c->st[r], k, robot_type[r], gait_id[r]);
  }
  return c;
}
int emu_ctrl_run(void *hnd, const float *dof, const float *body, const float *cmd, float *torques, int exact, int nthreads) {
  EmuCtrl &c = *static_cast<EmuCtrl *>(hnd);
  auto work = [&](int r0, int r1) {
    switch (c.h) {
#define EMU_CASE(HH) case HH: emu_ctrl_tick<HH>(c, r0, r1, dof, body, cmd, torques, exact); break;
      EMU_HORIZONS(EMU_CASE)
#undef EMU_CASE
      default: break;
    }
  };
  if (nthreads <= 1 || c.n < 2) { work(0, c.n); return 0; }
  std::vector<std::thread> pool;
  const int per = (c.n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) {
    const int r0 = t * per, r1 = std::min(c.n, r0 + per);
    if (r0 < r1) pool.emplace_back(work, r0, r1);
  }
  for (auto &t : pool) t.join();
  return 0;
}
void emu_ctrl_reset(void *hnd, const int *ids, int k) {     // mpc_ctrl_reset: RobotRunnerMin.reset + a new ConvexMpc object
  EmuCtrl &c = *static_cast<EmuCtrl *>(hnd);
  const int sl = 64 * c.h + 2;
  for (int i = 0; i < (ids ? k : c.n); ++i) {
    const int r = ids ? ids[i] : i;
    if (r < 0 || r >= c.n) continue;
    ctrl_reset(c.st[r], c.rc[r]);
    std::fill(c.state.begin() + (size_t)r * sl, c.state.begin() + (size_t)(r + 1) * sl, 0.0);
  }
}
void emu_ctrl_set_gait(void *hnd, const int *gait) {     // mpc_ctrl_set_gait: Parameters.cmpc_gait, re-read by run() on every tick (ConvexMPCLocomotion.py:224)
  EmuCtrl &c = *static_cast<EmuCtrl *>(hnd);
  for (int r = 0; r < c.n; ++r) if (gait[r] >= 0 && gait[r] < kNumGaitIds) c.st[r].gait_id = gait[r];
}
void emu_ctrl_set_iteration(void *hnd, const int *it) { EmuCtrl &c = *static_cast<EmuCtrl *>(hnd); for (int r = 0; r < c.n; ++r) c.st[r].iter = it[r]; }
void emu_ctrl_get(void *hnd, int *info, double *forces, float *rec) {
  EmuCtrl &c = *static_cast<EmuCtrl *>(hnd);
  if (info) std::memcpy(info, c.info.data(), sizeof(int) * c.info.size());
  if (forces) std::memcpy(forces, c.forces.data(), sizeof(double) * c.forces.size());
  if (rec) std::memcpy(rec, c.rec.data(), sizeof(float) * c.rec.size());
}
// estimator sub-state the controller keeps (StateEstimator.py:99-143): ground_normal_yaw [n][3], contact history [n][12], CoM height [n]
void emu_ctrl_get_estimate(void *hnd, float *normal, float *hist, float *pos_z) {
  EmuCtrl &c = *static_cast<EmuCtrl *>(hnd);
  for (int r = 0; r < c.n; ++r) {
    if (normal) for (int k = 0; k < 3; ++k) normal[3 * r + k] = c.st[r].normal[k];
    if (hist) for (int k = 0; k < 12; ++k) hist[12 * r + k] = c.st[r].hist[k];
    if (pos_z) pos_z[r] = c.st[r].pos_z;
  }
}
void emu_ctrl_close(void *hnd) { delete static_cast<EmuCtrl *>(hnd); }

// ---- RobotRunnerFSM.run replay (estimator -> fsm_tick -> [ctrl_pre -> emulated solve -> ctrl_post] | joint PD), horizon 10 ----
// init_mode [n], request [T][n]: FSM_StateName values; fsm_out [T][n][3] = (state, operating mode, recovery flag) after the tick.
int emu_fsm_replay(int n, int ticks, const double *robot_table, const int *robot_type, const int *gait_id, const int *gait_off,
                   const int *gait_dur, int flat_ground, double dt, int iters_between_mpc, double alpha, int check_safety, int op_mode,
                   const int *init_mode, const float *dof, const float *body, const float *cmd, const int *request, float *torques,
                   int *fsm_out) {
  constexpr int H = 10;
  GaitTable gt;
  gt.n_seg = H;
  for (int g = 0; g < kNumGaitIds; ++g) for (int j = 0; j < 4; ++j) { gt.offsets[g][j] = (float)gait_off[4 * g + j]; gt.durations[g][j] = (float)gait_dur[4 * g + j]; }
  CtrlParams cp{dt, iters_between_mpc, dt * iters_between_mpc, H, flat_ground};
  const FsmParams P = fsm_params(dt, check_safety);
  const int inlen = 56 + 4 * H, sl = 64 * H + 2;
  for (int r = 0; r < n; ++r) {
    const double *row = robot_table + 25 * robot_type[r];
    RobotConst rc;
    rc.abad = row[0]; rc.hip = row[1]; rc.knee = row[2];
    for (int k = 0; k < 3; ++k) rc.hiploc[k] = (float)row[3 + k];
    rc.body_height = row[10]; rc.mu = (float)row[11];
    for (int k = 0; k < 13; ++k) rc.weights[k] = (float)row[12 + k];
    const double inertia9[9] = {row[7], 0, 0, 0, row[8], 0, 0, 0, row[9]};
    const RobotModel mdl = make_model(row[6], inertia9, cp.dt_mpc, alpha);
    CtrlState st;
    FsmState f;
    ctrl_init(st, rc, robot_type[r], gait_id[r]);
    fsm_init(f, init_mode[r], op_mode, st, rc, 0.f);          // fresh StateEstimate: rBody = 0
    std::vector<double> state(sl, 0.0), forces(12 * H, 0.0);
    std::vector<float> rec(inlen);
    for (int t = 0; t < ticks; ++t) {
      const size_t idx = (size_t)t * n + r;
      float est[kEstLen];
      estimator_update(body + idx * 13, st.normal, est);
      fsm_tick(f, st, rc, P, dof + idx * 24, body + idx * 13, request[idx]);
      if (f.entered_loco) std::fill(state.begin(), state.end(), 0.0);     // cMPC.initialize builds a new ConvexMpc
      if (f.run_loco) {
        ctrl_pre(st, rc, gt, cp, dof + idx * 24, est, cmd + idx * 16, rec.data());
        int info[kInfoLen] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (st.do_solve) solve_one<H>(mdl, rec.data(), state.data(), forces.data(), info, false, nullptr);
        ctrl_post(st, rc, forces.data(), info[1] == kStSolved, torques + idx * 12);
      } else {
        fsm_joint_torques(f, st, torques + idx * 12);
      }
      fsm_out[idx * 3] = f.cur; fsm_out[idx * 3 + 1] = f.op_mode; fsm_out[idx * 3 + 2] = f.rs_flag;
    }
  }
  return 0;
}

// StateEstimator.update restatement: body [n][13], normal [n][3] -> est [n][18]
int emu_estimator_update(int n, const float *body, const float *normal, float *est) {
  for (int r = 0; r < n; ++r) estimator_update(body + 13 * r, normal + 3 * r, est + kEstLen * r);
  return 0;
}

// gelsd43::solve_ones on a batch: A [n][4][3] row-major float32 -> x [n][3], rank [n]
void emu_gelsd43(int n, const float *A, float *x, int *rank) {
  for (int i = 0; i < n; ++i) rank[i] = gelsd43::solve_ones(A + 12 * i, x + 3 * i);
}
void emu_set_poison(int on) { g_poison = on; }
void emu_set_max_iter(int it) { g_max_iter = it; }
void emu_set_seed_buffer(int *buf, int stride) { g_seed = buf; g_seed_stride = stride; }
void emu_set_split(int on) { g_split = on; }
void emu_set_exact_route(int r) { g_exact_route = r; }
void emu_check_counts(long *out) { out[0] = g_checks; out[1] = g_dual_cands; }
int emu_state_len(int h) { return 24 * h + 40 * h + 2; }
int emu_shared_bytes(int h) {
#define EMU_CASE(HH) if (h == HH) return (int)sizeof(Shared<HH>);
  EMU_HORIZONS(EMU_CASE)
#undef EMU_CASE
  return -1;
}

// model: per robot {mass, inertia9[9]} (10 doubles); in: [n][56+4h] floats; state: [n][state_len];
// forces: [n][12h]; info: [n][8].  Returns -1 for an unsupported horizon.
int emu_batch_solve(int h, int n, const double *model, double dt, double alpha, const float *in, double *state,
                    double *forces, int *info, int reverse_and_mode, int nthreads, long *phases_out) {
  if (emu_sc_len(h) < 0) return -1;
  const bool reverse = reverse_and_mode & 1, exact = reverse_and_mode & 2;   // bit 1: exact-optimum mode
  const int N = 12 * h, inlen = 56 + 4 * h, sl = emu_state_len(h);
  if (nthreads < 1) nthreads = 1;
  std::vector<std::thread> pool;
  auto work = [&](int lo, int hi) {
    for (int r = lo; r < hi; ++r) {
      RobotModel mdl = make_model(model[10 * r], model + 10 * r + 1, dt, alpha);
      long ph = 0;
      const float *ri = in + (size_t)r * inlen;
      double *rs = state + (size_t)r * sl, *rf = forces + (size_t)r * N;
      int *rinfo = info + (size_t)r * kInfoLen;
      switch (h) {
#define EMU_CASE(HH) case HH: solve_one<HH>(mdl, ri, rs, rf, rinfo, reverse, &ph, nullptr, exact, exact && g_seed ? g_seed + (size_t)r * g_seed_stride : nullptr); break;
        EMU_HORIZONS(EMU_CASE)
#undef EMU_CASE
      }
      if (phases_out) phases_out[r] = ph;
    }
  };
  const int per = (n + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) {
    const int lo = t * per, hi = std::min(n, (t + 1) * per);
    if (lo >= hi) break;
    if (nthreads == 1) work(lo, hi); else pool.emplace_back(work, lo, hi);
  }
  for (auto &t : pool) t.join();
  return 0;
}
}
