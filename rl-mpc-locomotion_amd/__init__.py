"""MI355X-native batched convex-MPC locomotion stepper (hot path of silvery107/rl-mpc-locomotion).

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C-ABI library), the host-side
mirror of the reference's plugin interface (``mpc_osqp`` shim, batched stepper), the constant tables
and the synthetic workload generator.  See DESIGN.md.
"""
__all__ = ["layout", "quadruped", "gait", "synthetic"]
