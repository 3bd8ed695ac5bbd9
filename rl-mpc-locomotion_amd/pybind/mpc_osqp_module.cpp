// mpc_osqp_module.cpp -- the pybind11 extension module `mpc_osqp` of the reference (MPC_Controller/convex_MPC/mpc_osqp.cc:952-983),
// served by the MI355X library through its C ABI (include/mpc_batch.h).
//
//   import sys; sys.path.insert(0, "<repo>/rl-mpc-locomotion_amd/pybind"); import mpc_osqp as mpc      # what ConvexMPCLocomotion.py:17-22 imports
//   cpp_mpc = mpc.ConvexMpc(mass, inertia9, 4, horizon, dt_mpc, alpha, mpc.QPOASES)                        # ConvexMPCLocomotion.py:102-108
//   forces  = cpp_mpc.compute_contact_forces(w, pos, vel, rpy, normal, omega, table, feet, mu, dpos, dvel, drpy, domega)   # :171-185
//
// Same class, constructor (7 positional arguments), method (13 std::vector<double> by value, as pybind11's STL casters give the
// reference, mpc_osqp.cc:578-591), enum with exported values, __version__ and TEST as the reference's module.  One ConvexMpc
// object = one single-robot batch handle; every call is one mpc_batch_solve_host_f64 (host pointers, synchronous) -- 1/1024 of the
// GPU by construction: the seam for running the unmodified reference Python, not for throughput (that is mpc_batch_solve /
// BatchedConvexMpc on N robots).  Built by __graft_entry__.build() with g++ against pybind11's headers, linked to
// csrc/libmpc_batch.so (dlopen at import, after torch); the pure-Python mirror rl_mpc_locomotion_amd.mpc_osqp (ctypes) stays as the fallback binding.
#include <dlfcn.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mpc_batch.h"

namespace py = pybind11;

enum QPSolverName { OSQP, QPOASES };   // mpc_osqp.cc:52

// The C ABI, bound at module import (dlopen AFTER `import torch`: the library must sit on the HIP runtime torch brings along when
// torch is in the process -- a copy of libamdhip64 pulled in earlier by DT_NEEDED would be a second runtime).
namespace cabi {
decltype(&::mpc_batch_create) create;
decltype(&::mpc_batch_destroy) destroy;
decltype(&::mpc_batch_set_solver) set_solver;
decltype(&::mpc_batch_solve_host_f64) solve_host_f64;
decltype(&::mpc_input_len) input_len;
decltype(&::mpc_last_error) last_error;
void load(const std::string &module_file) {
  const std::string dir = module_file.substr(0, module_file.find_last_of('/'));
  const std::string path = dir + "/../csrc/libmpc_batch.so";
  void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) throw std::runtime_error("mpc_osqp: cannot load " + path + " (" + dlerror() + "); build it with __graft_entry__.build() -- there is no CPU fallback");
  auto sym = [&](const char *n) { void *p = dlsym(h, n); if (!p) throw std::runtime_error(std::string("mpc_osqp: missing symbol ") + n); return p; };
  create = reinterpret_cast<decltype(create)>(sym("mpc_batch_create"));
  destroy = reinterpret_cast<decltype(destroy)>(sym("mpc_batch_destroy"));
  set_solver = reinterpret_cast<decltype(set_solver)>(sym("mpc_batch_set_solver"));
  solve_host_f64 = reinterpret_cast<decltype(solve_host_f64)>(sym("mpc_batch_solve_host_f64"));
  input_len = reinterpret_cast<decltype(input_len)>(sym("mpc_input_len"));
  last_error = reinterpret_cast<decltype(last_error)>(sym("mpc_last_error"));
}
}  // namespace cabi

class ConvexMpc {
 public:
  ConvexMpc(double mass, const std::vector<double> &inertia, int num_legs, int planning_horizon, double timestep, double alpha,
            QPSolverName qp_solver_name)
      : h_(planning_horizon), exact_(qp_solver_name == QPOASES) {
    if (num_legs != 4) throw std::invalid_argument("only quadrupeds (num_legs == 4) are supported");
    if (inertia.size() != 9) throw std::invalid_argument("inertia must have 9 elements");   // assert at mpc_osqp.cc:556
    check(cabi::create(&b_, 1, planning_horizon, timestep, alpha, &mass, inertia.data()), "mpc_batch_create");
    try {      // (a constructor that throws never runs the destructor: the handle is released here)
      check(cabi::set_solver(b_, exact_ ? MPC_SOLVER_EXACT : MPC_SOLVER_OSQP), "mpc_batch_set_solver");
      rec_.assign(cabi::input_len(planning_horizon), 0.0);
      out_.assign(12 * (size_t)planning_horizon, 0.0);
    } catch (...) {
      cabi::destroy(b_);
      b_ = nullptr;
      throw;
    }
  }
  ~ConvexMpc() { cabi::destroy(b_); }
  ConvexMpc(const ConvexMpc &) = delete;
  ConvexMpc &operator=(const ConvexMpc &) = delete;

  // mpc_osqp.cc:578-591: thirteen vectors by value; the 12 h forces ([step][leg][xyz], negated), or an empty vector when the OSQP branch
  // does not report OSQP_SOLVED (:781-794); the qpOASES branch returns its vector whatever the status (:906-947)
  std::vector<double> ComputeContactForces(std::vector<double> qp_weights, std::vector<double> com_position, std::vector<double> com_velocity,
                                           std::vector<double> com_roll_pitch_yaw, std::vector<double> ground_normal_vec,
                                           std::vector<double> com_angular_velocity, std::vector<double> foot_contact_states,
                                           std::vector<double> foot_positions_body_frame, std::vector<double> foot_friction_coeffs,
                                           std::vector<double> desired_com_position, std::vector<double> desired_com_velocity,
                                           std::vector<double> desired_com_roll_pitch_yaw, std::vector<double> desired_com_angular_velocity) {
    size_t o = 0;
    auto put = [&](const std::vector<double> &v, size_t k, const char *name) {
      if (v.size() != k) throw std::invalid_argument(std::string(name) + ": wrong length");
      for (size_t i = 0; i < k; ++i) rec_[o + i] = v[i];
      o += k;
    };
    put(qp_weights, 13, "qp_weights"); put(com_position, 3, "com_position"); put(com_velocity, 3, "com_velocity");
    put(com_roll_pitch_yaw, 3, "com_roll_pitch_yaw"); put(ground_normal_vec, 3, "ground_normal_vec");
    put(com_angular_velocity, 3, "com_angular_velocity"); put(foot_contact_states, 4 * (size_t)h_, "foot_contact_states");
    put(foot_positions_body_frame, 12, "foot_positions_body_frame"); put(foot_friction_coeffs, 4, "foot_friction_coeffs");
    put(desired_com_position, 3, "desired_com_position"); put(desired_com_velocity, 3, "desired_com_velocity");
    put(desired_com_roll_pitch_yaw, 3, "desired_com_roll_pitch_yaw"); put(desired_com_angular_velocity, 3, "desired_com_angular_velocity");
    int info[MPC_INFO_LEN] = {0};
    // The GIL is held for the whole call, as in the reference (mpc_osqp.cc has no gil_scoped_release): two Python threads calling the same
    // object are serialised -- the call works on the object's record / result buffers and the handle's staging buffers.
    check(cabi::solve_host_f64(b_, rec_.data(), out_.data(), info), "mpc_batch_solve_host_f64");
    status_ = info[1];
    iterations_ = info[0];
    if (exact_ ? info[1] == MPC_STATUS_NON_CVX : info[1] != MPC_STATUS_SOLVED) return {};
    return out_;
  }
  void ResetSolver() {}   // mpc_osqp.cc:576: flips a flag that nothing reads
  int status() const { return status_; }
  int iterations() const { return iterations_; }

 private:
  static void check(int rc, const char *what) {
    if (rc != MPC_OK) throw std::runtime_error(std::string(what) + ": " + cabi::last_error());
  }
  mpc_batch *b_ = nullptr;
  int h_;
  bool exact_;
  int status_ = 0, iterations_ = 0;
  std::vector<double> rec_, out_;
};

PYBIND11_MODULE(mpc_osqp, m) {
  m.doc() = "mpc_osqp: the reference's convex-MPC plugin module (mpc_osqp.cc:952-983) on the MI355X library (include/mpc_batch.h)";
  // The library shares torch's HIP runtime when torch is in the process (device tensors are handed over elsewhere): load it first.
  try { py::module_::import("torch"); } catch (py::error_already_set &) { PyErr_Clear(); }
  cabi::load(py::cast<std::string>(m.attr("__file__")));
  py::enum_<QPSolverName>(m, "QPSolverName").value("OSQP", OSQP, "OSQP").value("QPOASES", QPOASES, "QPOASES").export_values();
  py::class_<ConvexMpc>(m, "ConvexMpc")
      .def(py::init<double, const std::vector<double> &, int, int, double, double, QPSolverName>())
      .def("compute_contact_forces", &ConvexMpc::ComputeContactForces)
      .def("reset_solver", &ConvexMpc::ResetSolver)
      .def_property_readonly("status", &ConvexMpc::status, "OSQP status value of the last call (extension of this module)")
      .def_property_readonly("iterations", &ConvexMpc::iterations);
  m.attr("__version__") = "dev";
  m.attr("TEST") = py::int_(int(42));
}
