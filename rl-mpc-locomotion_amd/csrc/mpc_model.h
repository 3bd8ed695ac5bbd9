// mpc_model.h -- host-side construction of the per-robot model constants
// (ConvexMpc::ConvexMpc, mpc_osqp.cc:508-527: inv_mass_, inv_inertia_ = inertia_.inverse()).
#pragma once
#include "mpc_core.h"

namespace mpc {

// inertia9: the 9 numbers the reference passes (row-major 3x3; Eigen reads them column-major, i.e.
// transposed -- identical for the symmetric inertia tensors in use).  General cofactor inverse.
inline RobotModel make_model(double mass, const double *inertia9, double dt, double alpha) {
  double a[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) a[r * 3 + c] = inertia9[c * 3 + r];
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  RobotModel m;
  m.mass = mass;
  m.inv_mass = 1.0 / mass;
  const double inv[9] = {c00 / det, (a[2] * a[7] - a[1] * a[8]) / det, (a[1] * a[5] - a[2] * a[4]) / det,
                         c01 / det, (a[0] * a[8] - a[2] * a[6]) / det, (a[2] * a[3] - a[0] * a[5]) / det,
                         c02 / det, (a[1] * a[6] - a[0] * a[7]) / det, (a[0] * a[4] - a[1] * a[3]) / det};
  for (int i = 0; i < 9; ++i) m.inv_inertia[i] = inv[i];
  m.dt = dt;
  m.alpha = alpha;
  return m;
}

}  // namespace mpc
