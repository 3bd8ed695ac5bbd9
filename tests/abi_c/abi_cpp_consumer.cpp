// tests/abi_c/abi_cpp_consumer.cpp -- a C++ caller of the float64 host entry point with the reference's own argument types: thirteen
// std::vector<double>, as ConvexMpc::ComputeContactForces takes them (mpc_osqp.cc:578-591), concatenated in call order into the record
// of include/mpc_batch.h and handed to mpc_batch_solve_host_f64.  Batched over the robots of the input file (the per-call seam costs
// ~0.2 ms whatever the batch size: C++ callers should batch).
//   abi_cpp_consumer <libmpc_batch.so> <in.bin> <out.bin>
//   in:  {int n, int h, double dt, double alpha, double mass[n], double inertia9[n*9], double rec[n*(56+4h)]}   out: {int info[n*8], double forces[n*12h]}
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mpc_batch.h"

using Vec = std::vector<double>;
struct Call {   // the 13 positional arguments (mpc_osqp.cc:578-591)
  Vec qp_weights, com_position, com_velocity, com_roll_pitch_yaw, ground_normal_vec, com_angular_velocity, foot_contact_states,
      foot_positions_body_frame, foot_friction_coeffs, desired_com_position, desired_com_velocity, desired_com_roll_pitch_yaw,
      desired_com_angular_velocity;
};

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  void *lib = dlopen(argv[1], RTLD_NOW);
  if (!lib) { std::fprintf(stderr, "%s\n", dlerror()); return 3; }
  auto create = reinterpret_cast<decltype(&mpc_batch_create)>(dlsym(lib, "mpc_batch_create"));
  auto solve = reinterpret_cast<decltype(&mpc_batch_solve_host_f64)>(dlsym(lib, "mpc_batch_solve_host_f64"));
  auto destroy = reinterpret_cast<decltype(&mpc_batch_destroy)>(dlsym(lib, "mpc_batch_destroy"));
  auto err = reinterpret_cast<decltype(&mpc_last_error)>(dlsym(lib, "mpc_last_error"));
  FILE *f = std::fopen(argv[2], "rb");
  int n, h; double dt, alpha;
  if (!f || std::fread(&n, 4, 1, f) != 1 || std::fread(&h, 4, 1, f) != 1 || std::fread(&dt, 8, 1, f) != 1 || std::fread(&alpha, 8, 1, f) != 1) return 4;
  Vec mass(n), inertia(9 * (size_t)n), flat((size_t)n * (56 + 4 * h));
  if (std::fread(mass.data(), 8, n, f) != (size_t)n || std::fread(inertia.data(), 8, 9 * n, f) != (size_t)9 * n || std::fread(flat.data(), 8, flat.size(), f) != flat.size()) return 4;
  std::fclose(f);
  // the caller's data as the reference's argument lists ...
  std::vector<Call> calls(n);
  const size_t sizes[13] = {13, 3, 3, 3, 3, 3, (size_t)4 * h, 12, 4, 3, 3, 3, 3};
  for (int r = 0; r < n; ++r) {
    const double *p = flat.data() + (size_t)r * (56 + 4 * h);
    Vec *fields[13] = {&calls[r].qp_weights, &calls[r].com_position, &calls[r].com_velocity, &calls[r].com_roll_pitch_yaw, &calls[r].ground_normal_vec,
                       &calls[r].com_angular_velocity, &calls[r].foot_contact_states, &calls[r].foot_positions_body_frame, &calls[r].foot_friction_coeffs,
                       &calls[r].desired_com_position, &calls[r].desired_com_velocity, &calls[r].desired_com_roll_pitch_yaw, &calls[r].desired_com_angular_velocity};
    for (int k = 0; k < 13; ++k) { fields[k]->assign(p, p + sizes[k]); p += sizes[k]; }
  }
  // ... and back into one batch record (call order), one library call for all robots
  Vec rec;
  rec.reserve(flat.size());
  for (const Call &c : calls)
    for (const Vec *v : {&c.qp_weights, &c.com_position, &c.com_velocity, &c.com_roll_pitch_yaw, &c.ground_normal_vec, &c.com_angular_velocity, &c.foot_contact_states,
                         &c.foot_positions_body_frame, &c.foot_friction_coeffs, &c.desired_com_position, &c.desired_com_velocity, &c.desired_com_roll_pitch_yaw,
                         &c.desired_com_angular_velocity})
      rec.insert(rec.end(), v->begin(), v->end());
  mpc_batch *b = nullptr;
  if (create(&b, n, h, dt, alpha, mass.data(), inertia.data()) != MPC_OK) { std::fprintf(stderr, "create: %s\n", err()); return 5; }
  Vec forces((size_t)n * 12 * h, 0.0);
  std::vector<int> info((size_t)n * MPC_INFO_LEN, 0);
  if (solve(b, rec.data(), forces.data(), info.data()) != MPC_OK) { std::fprintf(stderr, "solve: %s\n", err()); return 6; }
  destroy(b);
  FILE *o = std::fopen(argv[3], "wb");
  std::fwrite(info.data(), 4, info.size(), o);
  std::fwrite(forces.data(), 8, forces.size(), o);
  std::fclose(o);
  return 0;
}
