// mpc_horizon.h -- the seam between the C ABI (mpc_batch.hip) and the per-horizon kernel sets (mpc_horizon.hip).
//
// ConvexMpc accepts any planning_horizon (mpc_osqp.cc:186-190, 508-574; the shipped Python uses 10, ConvexMPCLocomotion.py:27;
// BASELINE's configurations 10, 16, 20).  The kernels keep their matrices in statically indexed register tiles, so the horizon is a
// template parameter; every horizon of MPC_HORIZON_LIST is its own translation unit (mpc_horizon.hip compiled with -DMPC_H=h: prep,
// job, one-workgroup-per-robot, exact and fall-back kernels of that horizon), built in parallel and linked into the one library.
// The library looks a horizon up in the table of HorizonOps the units export; 2 <= h <= 20 ships (a 20-step workgroup is 256
// threads with 60 KB of LDS: the largest that keeps two robots on a CU).
#pragma once
#include <hip/hip_runtime.h>

#include "mpc_core.h"

#ifndef MPC_HORIZON_LIST
#define MPC_HORIZON_LIST(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)
#endif

namespace mpc {

constexpr int kSchedNext = 0, kSchedHead = 1, kSchedTail = 2, kSchedJobs = 3, kSchedLen = 4;   // job bookkeeping of a launch (mpc_solve_jobs_kernel)
constexpr int kOrderBuckets = 256, kOrderHistory = 10;                                          // dispatch order of a launch (order_block)

struct LaunchArgs {        // one solver launch on n robots: the prep kernel, then the solve kernel(s) of the selected mode
  int n;
  const RobotModel *models;
  const float *in;         // the input records [n, 56 + 4 h] as float32, float64 or float16: exactly one of the three is set
  const double *in64;
  const _Float16 *in16;
  double *state, *qp, *sc, *forces;
  int *info;
  long long *prof;
  const int *active;       // [n] or null (all): robots whose controller is due for an MPC update
  int *order;
  unsigned char *hist;
  int hist_slot;
  hipEvent_t *ev;          // [3] or null: recorded before the prep kernel, between the two, after the solve kernel(s)
  hipStream_t stream;
  int exact, max_iter;
  int *sched, *ready;
  int job_slots;
  int *seed;               // [n, 4 h] exact mode: the working set each robot's previous call ended on (mpc_wrench.h seed_working_set), or null: every call starts empty
};

struct HorizonOps {
  int h;
  int qp_len, sc_len;      // doubles per robot of the two inter-kernel records on the device ...
  int xqp_len, xsc_len;    // ... and as mpc_batch_get_qp / mpc_batch_get_scale hand them out
  int (*launch)(const LaunchArgs &);      // returns a hipError_t
  void (*expand)(const double *qp, const double *sc, double *xqp, double *xsc);   // one robot's device records -> the documented layout
};

}  // namespace mpc
