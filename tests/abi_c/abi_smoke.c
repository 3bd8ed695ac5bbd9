/* tests/abi_c/abi_smoke.c -- a plain C consumer of include/mpc_batch.h (no Python, no torch, no HIP headers):
 * what a C/C++ maintainer of the reference would link against.
 *   abi_smoke <libmpc_batch.so> check              : dlopen + resolve every entry point (no GPU needed)
 *   abi_smoke <libmpc_batch.so> solve <in> <out>   : read {int n, int h, double dt, double alpha, double mass[n],
 *                                                    double inertia9[9n], float rec[n][56+4h]} from <in>, solve through
 *                                                    mpc_batch_create / mpc_batch_solve_host, write {int info[n][8],
 *                                                    double forces[n][12h]} to <out>.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpc_batch.h"

static const char *kSymbols[] = {
    "mpc_input_len", "mpc_supported_horizons", "mpc_batch_create", "mpc_batch_destroy", "mpc_batch_solve", "mpc_batch_set_solver", "mpc_batch_set_max_iter", "mpc_batch_solve_f64", "mpc_batch_solve_f16", "mpc_batch_reset", "mpc_batch_reset_device", "mpc_batch_solve_host_f64",
    "mpc_batch_solve_host", "mpc_batch_size", "mpc_batch_horizon", "mpc_batch_device_bytes", "mpc_batch_state_len",
    "mpc_batch_get_state", "mpc_batch_set_state", "mpc_batch_qp_len", "mpc_batch_scale_len", "mpc_batch_get_qp", "mpc_batch_get_scale", "mpc_batch_get_profile", "mpc_batch_enable_timing", "mpc_batch_kernel_times",
    "mpc_last_error", "mpc_ctrl_create", "mpc_ctrl_destroy",
    "mpc_ctrl_step", "mpc_ctrl_run", "mpc_ctrl_reset", "mpc_ctrl_reset_device", "mpc_ctrl_set_gait", "mpc_ctrl_set_solver", "mpc_ctrl_solver_info", "mpc_ctrl_solver_record", "mpc_ctrl_solver_forces", "mpc_ctrl_solver", "mpc_ctrl_set_iteration", "mpc_device_clock", "mpc_ctrl_fsm_init",
    "mpc_ctrl_run_fsm", "mpc_ctrl_fsm_reset", "mpc_ctrl_fsm_reset_device", "mpc_ctrl_fsm_state", "mpc_policy_create",
    "mpc_policy_destroy", "mpc_policy_step", "mpc_policy_observations", "mpc_ctrl_estimate", "mpc_ctrl_update_estimate",
    "mpc_pack_commands", "mpc_ctrl_set_gait_device", "mpc_pack_commands_scaled", "mpc_ctrl_policy_observations", "mpc_ctrl_run_fsm_estimated",
    "mpc_peer_create", "mpc_peer_handle", "mpc_peer_connect", "mpc_peer_put", "mpc_peer_wait", "mpc_peer_timeouts", "mpc_peer_destroy", "mpc_peer_last_error"};

typedef int (*create_fn)(mpc_batch **, int, int, double, double, const double *, const double *);
typedef int (*solve_host_fn)(mpc_batch *, const float *, double *, int *);
typedef void (*destroy_fn)(mpc_batch *);
typedef const char *(*err_fn)(void);
typedef int (*len_fn)(int);

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: abi_smoke lib check | solve in out\n"); return 2; }
  void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  for (size_t i = 0; i < sizeof kSymbols / sizeof *kSymbols; ++i)
    if (!dlsym(lib, kSymbols[i])) { fprintf(stderr, "missing symbol %s\n", kSymbols[i]); return 4; }
  if (!strcmp(argv[2], "check")) {
    len_fn input_len = (len_fn)dlsym(lib, "mpc_input_len");
    if (input_len(10) != 96) { fprintf(stderr, "mpc_input_len(10) != 96\n"); return 5; }
    printf("ok %zu symbols\n", sizeof kSymbols / sizeof *kSymbols);
    return 0;
  }
  if (strcmp(argv[2], "solve") || argc < 5) return 2;
  FILE *f = fopen(argv[3], "rb");
  if (!f) return 6;
  int n, h; double dt, alpha;
  if (fread(&n, sizeof n, 1, f) != 1 || fread(&h, sizeof h, 1, f) != 1 || fread(&dt, sizeof dt, 1, f) != 1 || fread(&alpha, sizeof alpha, 1, f) != 1) return 6;
  const int len = 56 + 4 * h;
  double *mass = malloc(sizeof(double) * n), *inertia = malloc(sizeof(double) * 9 * n), *forces = calloc((size_t)n * 12 * h, sizeof(double));
  float *rec = malloc(sizeof(float) * (size_t)n * len);
  int *info = calloc((size_t)n * MPC_INFO_LEN, sizeof(int));
  if (fread(mass, sizeof(double), n, f) != (size_t)n || fread(inertia, sizeof(double), 9 * n, f) != (size_t)(9 * n) ||
      fread(rec, sizeof(float), (size_t)n * len, f) != (size_t)n * len) return 6;
  fclose(f);
  create_fn create = (create_fn)dlsym(lib, "mpc_batch_create");
  solve_host_fn solve = (solve_host_fn)dlsym(lib, "mpc_batch_solve_host");
  destroy_fn destroy = (destroy_fn)dlsym(lib, "mpc_batch_destroy");
  err_fn last_error = (err_fn)dlsym(lib, "mpc_last_error");
  mpc_batch *b = NULL;
  int rc = create(&b, n, h, dt, alpha, mass, inertia);
  if (rc != MPC_OK) { fprintf(stderr, "mpc_batch_create: %d %s\n", rc, last_error()); return 7; }
  rc = solve(b, rec, forces, info);
  if (rc != MPC_OK) { fprintf(stderr, "mpc_batch_solve_host: %d %s\n", rc, last_error()); return 8; }
  destroy(b);
  f = fopen(argv[4], "wb");
  if (!f) return 9;
  fwrite(info, sizeof(int), (size_t)n * MPC_INFO_LEN, f);
  fwrite(forces, sizeof(double), (size_t)n * 12 * h, f);
  fclose(f);
  printf("solved %d robots, status[0] = %d\n", n, info[1]);
  return 0;
}
