/*
 * include/mpc_batch.h -- C ABI of the MI355X batched convex-MPC contact-force solver.
 *
 * Drop-in boundary.  Each entry point names the reference interface it replaces
 * (paths relative to the reference repository, silvery107/rl-mpc-locomotion):
 *
 *   mpc_batch_create   <- mpc_osqp.ConvexMpc(mass, inertia[9], num_legs=4, planning_horizon, timestep,
 *                         alpha, qp_solver_name)            MPC_Controller/convex_MPC/mpc_osqp.cc:508-574,
 *                         call site ConvexMPCLocomotion.py:102-108 -- one object per robot there, one
 *                         batch handle for N robots here.
 *   mpc_batch_solve    <- ConvexMpc.compute_contact_forces(13 args) -> list[12 h]
 *                                                           mpc_osqp.cc:578-796, call site
 *                         ConvexMPCLocomotion.py:171-185, looped over robots at
 *                         RL_Environment/tasks/aliengo.py:252-256.
 *   mpc_batch_reset    <- the re-construction of ConvexMpc in ConvexMPCLocomotion.initialize
 *                         (ConvexMPCLocomotion.py:89-108) reached from controllers[idx].reset(),
 *                         RL_Environment/tasks/aliengo.py:333-334: the next solve of that robot is a cold
 *                         "osqp_setup" solve (x = y = z = 0, rho = 0.1).
 *   mpc_batch_destroy  <- ~ConvexMpc (osqp_cleanup), mpc_osqp.cc:192.
 *
 * Input record (float32, length 56 + 4 h per robot): the 13 positional arguments of
 * compute_contact_forces concatenated in call order --
 *   [0,13)  qp_weights             [13,16) com_position        [16,19) com_velocity
 *   [19,22) com_roll_pitch_yaw     [22,25) ground_normal_vec   [25,28) com_angular_velocity
 *   [28,28+4h) foot_contact_states, row-major [step][leg]
 *   then foot_positions_body_frame[12] ([leg][xyz]), foot_friction_coeffs[4], desired_com_position[3],
 *   desired_com_velocity[3], desired_com_roll_pitch_yaw[3], desired_com_angular_velocity[3].
 * (The reference's Python passes float32 / float16 values; pybind11 widens them to double, which the
 * kernel does as well.  All arithmetic is fp64.)
 *
 * Output: forces, fp64, 12 h per robot, [step][leg][xyz], already negated like mpc_osqp.cc:789-790.
 * Error behaviour: the reference returns an EMPTY list unless OSQP reports OSQP_SOLVED
 * (mpc_osqp.cc:781-794).  Here the robot's force row is left untouched and info[1] (status) != 1.
 *
 * info record (int32, 8 per robot): {iterations, osqp status_val, status_polish, rho_updates,
 * factorisations, first_run, 0, 0}.  status_val takes OSQP's values (extern/osqp/include/constants.h:17-31): SOLVED (1),
 * MAX_ITER_REACHED (-2), PRIMAL_INFEASIBLE (-3) / DUAL_INFEASIBLE (-4) when the certificates of auxil.c:364-515 hold at a
 * termination check (evaluated exactly where check_termination does, auxil.c:732,744 -- a caller-made infeasible problem, e.g.
 * negative friction coefficients, ends after 25-50 iterations like in OSQP, not after max_iter), their *_INACCURATE forms
 * (2, 3, 4) from the second look at max_iter (osqp.c:563-568), and NON_CVX (-7; also NaN / inf input -- the robot's
 * warm-start record is then cleared, its next call starts cold).  Only SOLVED returns forces, as in the reference.
 *
 * All pointers named d_* are DEVICE pointers (HBM); `stream` is a hipStream_t (0 = default stream).
 * Functions return 0 on success, a negative MPC_E_* code otherwise; mpc_last_error() gives the text.
 */
#ifndef MPC_BATCH_H
#define MPC_BATCH_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpc_batch mpc_batch;

enum {
  MPC_OK = 0,
  MPC_E_ARG = -1,        /* bad argument (null pointer, n <= 0, ...) */
  MPC_E_HORIZON = -2,    /* planning horizon outside the built range (mpc_supported_horizons: 2 .. 20 in the shipped library) */
  MPC_E_HIP = -3,        /* HIP runtime error */
  MPC_E_NODEVICE = -4    /* no usable GPU */
};

#define MPC_INFO_LEN 8
#define MPC_STATUS_SOLVED 1           /* OSQP_SOLVED */
#define MPC_STATUS_SOLVED_INACCURATE 2            /* at max_iter, the last check passes at 10 x the tolerances (osqp.c:563-568) */
#define MPC_STATUS_PRIMAL_INFEASIBLE_INACCURATE 3
#define MPC_STATUS_DUAL_INFEASIBLE_INACCURATE 4
#define MPC_STATUS_MAX_ITER (-2)      /* OSQP_MAX_ITER_REACHED */
#define MPC_STATUS_PRIMAL_INFEASIBLE (-3)         /* OSQP's certificate of auxil.c:364-424 holds */
#define MPC_STATUS_DUAL_INFEASIBLE (-4)           /* ... of auxil.c:426-505 */
#define MPC_STATUS_NON_CVX (-7)       /* OSQP_NON_CVX (also: KKT matrix not positive definite) */

int mpc_input_len(int horizon);                       /* 56 + 4 h */
int mpc_supported_horizons(int *out, int cap);        /* writes up to cap horizons, returns the count */

/* mass[n], inertia9[n*9] (row-major 3x3 per robot) are HOST arrays. */
int mpc_batch_create(mpc_batch **out, int n_robots, int horizon, double timestep, double alpha,
                     const double *mass, const double *inertia9);
void mpc_batch_destroy(mpc_batch *b);

/* Which result a solve returns -- the reference's QPSolverName (mpc_osqp.cc:952-967, constructor argument of ConvexMpc):
 *   MPC_SOLVER_OSQP  (default) the OSQP branch (mpc_osqp.cc:690-796): OSQP 0.6.0's iterates at eps 1e-3 with polish, warm-started
 *                    from the previous call -- BASELINE.json's comparator;
 *   MPC_SOLVER_EXACT the qpOASES branch (:797-947, what the shipped Python selects, ConvexMPCLocomotion.py:108): the QP's optimum
 *                    (unique: the Hessian is 2 B^T Q B + alpha I), cold on every call like that branch (the RESULT: the active-set method behind it
 *                    starts from the working set of the robot's previous call, which changes how fast it gets there, not where).  qpOASES itself is an empty
 *                    submodule in the reference; its RESULT is reproduced, by an active-set method of this library's own
 *                    (csrc/mpc_wrench.h active_set: Goldfarb-Idnani's dual method on the problem with the swing feet eliminated,
 *                    like :838-856, then the polish on its set, accepted only if it passes the optimality conditions at 1e-10;
 *                    info[0] = working-set changes).  Eliminated feet return -0.0, sign bit included: that branch sets them to 0.0f and
 *                    negates the whole vector on the way out (:924-927, 940-942).  NOT reproduced, because qpOASES' source is an empty
 *                    submodule of the reference: its own iteration -- the nWSR = 100 working-set cap under setToMPC() tolerances
 *                    (:906-917), after which qpOASES hands back an unconverged iterate; here every robot ends at the optimum (or on
 *                    the fall-back route below).  That branch returns its
 *                    vector whatever its solver's status (:906-947): so does this mode -- a robot that the fall-back route (ADMM
 *                    towards 1e-9) leaves unfinished reports SOLVED_INACCURATE / MAX_ITER_REACHED WITH its iterate written; only
 *                    a non-finite / non-convex problem (NON_CVX) writes nothing. */
enum { MPC_SOLVER_OSQP = 0, MPC_SOLVER_EXACT = 1 };
int mpc_batch_set_solver(mpc_batch *b, int solver);

/* OSQP's max_iter setting (osqp_update_max_iter; the reference keeps the default 4000, extern/osqp/include/constants.h:60) for
 * the OSQP mode: a positive multiple of 25, OSQP's check_termination interval.  A robot that reaches it reports what
 * osqp.c:563-568 would: MAX_ITER_REACHED, or one of the *_INACCURATE statuses when the last check passes at ten times the
 * tolerances. */
int mpc_batch_set_max_iter(mpc_batch *b, int max_iter);

/* d_in: [n, 56+4h] float32; d_forces: [n, 12h] float64; d_info: [n, 8] int32 (may be NULL). */
int mpc_batch_solve(mpc_batch *b, const float *d_in, double *d_forces, int *d_info, void *stream);

/* The same with a float64 input record (the reference's pybind11 signature takes std::vector<double>, mpc_osqp.cc:578-591:
 * nothing is narrowed on this entry). */
int mpc_batch_solve_f64(mpc_batch *b, const double *d_in, double *d_forces, int *d_info, void *stream);

/* The same with a float16 input record (IEEE binary16: torch.float16 / numpy.float16 bits) -- BASELINE.json configs[4], "fp16 state":
 * the 13 arguments stored in half precision.  Every value is widened to double on load and the arithmetic is the same fp64 as on the
 * other entries, so the result equals the float64 entry fed the same (fp16-representable) values bit for bit.  (The reference's
 * own state arguments are partly fp16 already: com_roll_pitch_yaw arrives as numpy.float16, MPC_Controller/common/StateEstimator.py /
 * math_utils/orientation_tools.py:120-133.  "fp32 accumulate" is NOT offered: fp32 arithmetic does not hold BASELINE's 1e-3 bar against
 * OSQP on 42-52 % of the robots, profiles/r02_fp32_port_failure_rate.json.) */
int mpc_batch_solve_f16(mpc_batch *b, const unsigned short *d_in, double *d_forces, int *d_info, void *stream);

/* Cold-start the listed robots (HOST array of indices); ids == NULL resets all. */
int mpc_batch_reset(mpc_batch *b, const int *ids, int k, void *stream);

/* The same with a DEVICE array of indices (an Isaac-style env_ids tensor): no host round trip, stream-ordered. */
int mpc_batch_reset_device(mpc_batch *b, const int *d_ids, int k, void *stream);

/* Convenience for the per-robot plugin seam: host buffers, synchronous. */
int mpc_batch_solve_host(mpc_batch *b, const float *h_in, double *h_forces, int *h_info);

int mpc_batch_solve_host_f64(mpc_batch *b, const double *h_in, double *h_forces, int *h_info);   /* float64 record */

int mpc_batch_size(const mpc_batch *b);
int mpc_batch_horizon(const mpc_batch *b);
/* Bytes of HBM the handle owns (state + scratch + models). */
long long mpc_batch_device_bytes(const mpc_batch *b);
/* Warm-start state access (determinism tests / checkpointing): [n, 64 h + 2] float64. */
int mpc_batch_state_len(const mpc_batch *b);
int mpc_batch_get_state(mpc_batch *b, double *h_state);
int mpc_batch_set_state(mpc_batch *b, const double *h_state);
/* What the prep kernel handed to the solve kernel in the last launch (parity tests of the assembly and of the Ruiz scaling):
 *   QP record    [n, mpc_batch_qp_len]    q[12h] l[20h] u[20h] cone[15] pad | B6[6x12] th1[6x6] th2[6] pad -- the QP of
 *                mpc_osqp.cc:606-688 with its Hessian in the wrench form P = BB^T Theta BB + alpha I (csrc/mpc_wrench.h);
 *   scale record [n, mpc_batch_scale_len] D[12h] E[20h] q_s[12h] A_s[15*4h] l_s[20h] u_s[20h] c 1/c -- OSQP's scaling.c output -- and two
 *                doubles of job hand-over (the ADMM part's residuals, read by the solve's polish job).
 * (On the device the records are shorter -- three bound values per foot, no scaled cone block, csrc/mpc_core.h -- and the accessors
 * expand them with the solve kernel's own arithmetic: l_s = E l, u_s = E u, a foot's scaled cone block = (cone block x E of its rows) x D
 * of its columns.) */
int mpc_batch_qp_len(const mpc_batch *b);
int mpc_batch_scale_len(const mpc_batch *b);
int mpc_batch_get_qp(mpc_batch *b, double *h_qp);
int mpc_batch_get_scale(mpc_batch *b, double *h_sc);
/* Kernel timing (benchmarks): after mpc_batch_enable_timing every launch records HIP events on its stream around the
 * prep kernel (QP assembly + Ruiz scaling) and the solve kernel; mpc_batch_kernel_times synchronises and returns the durations (ms) of the last
 * `last_k` launches (oldest first, at most 64). */
int mpc_batch_enable_timing(mpc_batch *b);
int mpc_batch_kernel_times(mpc_batch *b, int last_k, float *ms_prep, float *ms_solve);
/* Shader-clock cycles of the last solve, [n, 16] int64: slot 15 = the ADMM part of the solve (always; NEGATED when the solve was a
 * cold one; the longest of a robot's last ten warm values orders the next launch's jobs longest-first), slot 0 = its polish job;
 * slots 1-14 = per section (csrc/mpc_core.h), filled only by a library built with -DMPC_SECTION_PROFILE (tools/section_profile.py)
 * and zero otherwise -- the counters cost registers. */
int mpc_batch_get_profile(mpc_batch *b, long long *h_prof);

/* ---- per-tick controller (the rest of the hot path around the solve) -------------------------------
 *
 *   mpc_ctrl_create  <- RobotRunnerMin.init (robot_runner/RobotRunnerMin.py:14-47) for N robots: Quadruped,
 *                       LegController, ConvexMPCLocomotion(dt, 27/(1000 dt)) and its ConvexMpc object.
 *   mpc_ctrl_step    <- the part of RobotRunnerMin.run (RobotRunnerMin.py:54-75) after StateEstimator.update:
 *                       LegController.updateData (LegController.py:89-106), ConvexMPCLocomotion.run
 *                       (ConvexMPCLocomotion.py:222-378: gait, estimator sub-steps, foot placement, the MPC solve
 *                       every iterationsBetweenMPC-th tick, swing Bezier, leg commands) and
 *                       LegController.updateCommand (LegController.py:108-132) -> 12 joint torques per robot.
 *   mpc_ctrl_reset   <- RobotRunnerMin.reset (RobotRunnerMin.py:49-52), RL_Environment/tasks/aliengo.py:333-334.
 *   mpc_ctrl_set_gait<- the process-global Parameters.cmpc_gait (Parameters.py:17), per robot here.
 *
 * d_dof:  [n, 12, 2] float32 (pos, vel; legs FL FR RL RR)         = dof_states of controller.run
 * d_est:  [n, 18]    float32 StateEstimator.update outputs: vBody[3], omegaBody[3], rpyBody[3] (float16
 *                    valued), ground_R_body_frame[9] (float16 valued, row-major)
 * d_cmd:  [n, 16]    float32 vx, vy, yaw_rate, 13 MPC weights      = commands of controller.run
 * d_torques: [n, 12] float32, FL FR RL RR x (hip, thigh, calf)
 * robot_table: [n_types, 25] float64 rows {abad, hip, knee link lengths, abad location xyz, mass, inertia
 *   diag xyz, body height, mu, 13 default weights} (MPC_Controller/common/Quadruped.py:16-92);
 * gait_off / gait_dur: [8, 4] int32 offsets / durations in MPC segments per gait id
 *   (ConvexMPCLocomotion.py:30-56), horizon segments per cycle.
 */
typedef struct mpc_ctrl mpc_ctrl;
int mpc_ctrl_create(mpc_ctrl **out, int n_robots, int horizon, double controller_dt, int iterations_between_mpc,
                    double alpha, int flat_ground, const int *robot_type, const int *gait_id, int n_types,
                    const double *robot_table, const int *gait_off, const int *gait_dur);
void mpc_ctrl_destroy(mpc_ctrl *c);
int mpc_ctrl_step(mpc_ctrl *c, const float *d_dof, const float *d_est, const float *d_cmd, float *d_torques, void *stream);
/* The whole controller.run(dof_states, body_states, commands) -> torques of RobotRunnerMin.run
 * (robot_runner/RobotRunnerMin.py:54-75, looped at RL_Environment/tasks/aliengo.py:252-256):
 * StateEstimator.update (common/StateEstimator.py:57-97, with its float16 / float32 arithmetic) then mpc_ctrl_step.
 * d_body: [n, 13] float32 = pos3, quat xyzw, linear velocity (world), angular velocity (world). */
int mpc_ctrl_run(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, float *d_torques, void *stream);
int mpc_ctrl_reset(mpc_ctrl *c, const int *ids, int k, void *stream);      /* HOST ids; NULL = all */
int mpc_ctrl_reset_device(mpc_ctrl *c, const int *d_ids, int k, void *stream);   /* DEVICE ids (env_ids tensor), stream-ordered */
int mpc_ctrl_set_gait(mpc_ctrl *c, const int *gait_id, void *stream);       /* HOST [n] */
/* ... the same from a DEVICE array [n] int32, stream-ordered, no host round trip: ConvexMPCLocomotion.run re-reads Parameters.cmpc_gait on EVERY
 * tick (ConvexMPCLocomotion.py:224-244), so a gait switch may fall between any two controller.run calls (BASELINE configs[2]: Trot / Walk / Bound
 * cycling every 50 steps); iterationCounter, firstSwing, swingTimeRemaining and the swing trajectories carry over.  An id outside the reference's
 * dispatch (0, 1, 2, 3, 5, 6, 7) leaves that robot's gait unchanged. */
int mpc_ctrl_set_gait_device(mpc_ctrl *c, const int *d_gait_id, void *stream);
/* The QPSolverName argument of the controller's ConvexMpc objects (ConvexMPCLocomotion.py:102-108; the shipped Python passes QPOASES):
 * MPC_SOLVER_OSQP (default here: BASELINE's comparator) or MPC_SOLVER_EXACT, see mpc_batch_set_solver. */
int mpc_ctrl_set_solver(mpc_ctrl *c, int solver);
int mpc_ctrl_solver_info(mpc_ctrl *c, int *h_info);                          /* [n, 8] of the last solves */
/* What solveDenseMPC marshalled (ConvexMPCLocomotion.py:128-185): the 13 arguments of each robot's LAST compute_contact_forces call as the
 * controller built them, HOST [n, 56 + 4 h] float32 in the layout of mpc_batch_solve (parity tests of the controller against the
 * reference's recorded calls). */
int mpc_ctrl_solver_record(mpc_ctrl *c, float *h_rec);
/* ... and what came back: each robot's force vector of its LAST solve, HOST [n, 12 h] float64 (rows of robots that never solved: zeros). */
int mpc_ctrl_solver_forces(mpc_ctrl *c, double *h_forces);
/* The controllers' ConvexMpc objects (ConvexMPCLocomotion.solver, ConvexMPCLocomotion.py:102-108) as the batch handle that mpc_ctrl_create
 * built: BORROWED (mpc_ctrl_destroy frees it) -- for the accessors of the first section on a controller's solver (mpc_batch_enable_timing /
 * _kernel_times, _get_state, _get_profile, _set_max_iter ...). */
mpc_batch *mpc_ctrl_solver(mpc_ctrl *c);
/* ConvexMPCLocomotion.iterationCounter (ConvexMPCLocomotion.py:62, a plain attribute there) of every robot, HOST [n]: the gait phase and
 * which tick is the next MPC update follow from it (benchmarks: SURVEY.md 8(d) samples the counter per robot). */
int mpc_ctrl_set_iteration(mpc_ctrl *c, const int *iteration, void *stream);

/* ---- control FSM around the controller (RobotRunnerFSM) -------------------------------------------
 *
 *   mpc_ctrl_fsm_init   <- RobotRunnerFSM.init (robot_runner/RobotRunnerFSM.py:13-39) -> ControlFSM.__init__ / initialize
 *                          (FSM_states/ControlFSM.py:28-78): fresh controller objects, then onEnter of the state named by
 *                          control_mode[r] (HOST [n]; 0 PASSIVE, 4 LOCOMOTION, 6 RECOVERY_STAND = FSM_StateName,
 *                          MPC_Controller/utils.py:26-30).  The reference's process-global Parameters.control_mode /
 *                          operatingMode / FSM_check_safety (Parameters.py:35-42) become per-robot / per-handle here;
 *                          operating_mode: 0 TEST, 1 NORMAL (utils.py:32-35).
 *   mpc_ctrl_run_fsm    <- RobotRunnerFSM.run(dof_states, body_states, commands) (:44-71): updateData, zeroCommand,
 *                          StateEstimator.update, ControlFSM.runFSM (:80-124 -- Passive / RecoveryStand joint-PD states,
 *                          Locomotion = the mpc_ctrl_run path incl. locomotionSafe, FSM_State_Locomotion.py:104-136),
 *                          LegController.updateCommand.  d_request [n] int32 = the control mode requested for each robot
 *                          this tick (what Parameters.control_mode holds when the reference's run() is called).
 *   mpc_ctrl_fsm_reset  <- RobotRunnerFSM.reset (:41-42) = ControlFSM.initialize for the robots in ids (HOST, NULL = all),
 *                          control_mode HOST [n] or NULL (keep the modes of the last initialisation).
 *   mpc_ctrl_fsm_state  <- [n, 4] int32 HOST: current FSM_StateName, operating mode, RecoveryStand flag (0 StandUp, 1 FoldLegs,
 *                          2 RollOver), and the per-robot "locomotion was unsafe" flag (Parameters.locomotionUnsafe).
 */
int mpc_ctrl_fsm_init(mpc_ctrl *c, const int *control_mode, int operating_mode, int check_safety, void *stream);
int mpc_ctrl_run_fsm(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, const int *d_request, float *d_torques, void *stream);
int mpc_ctrl_fsm_reset(mpc_ctrl *c, const int *ids, int k, const int *control_mode, void *stream);
/* mpc_ctrl_fsm_reset with a DEVICE array of indices (an env_ids tensor): stream-ordered, no host round trip; the control modes of the last
 * (re)initialisation are kept. */
int mpc_ctrl_fsm_reset_device(mpc_ctrl *c, const int *d_ids, int k, void *stream);
int mpc_ctrl_fsm_state(mpc_ctrl *c, int *h_out);

/* ---- weight policy: observations -> MPC weights (the deployment path of the learned policy) ----------
 *
 *   mpc_policy_create        <- WeightPolicy.__init__ (RL_Environment/WeightPolicy.py:33-92): the actor of rsl_rl's
 *                               ActorCritic (act_inference = actor(obs), a Linear/ELU stack, hidden sizes
 *                               LeggedCfgPPO.policy.actor_hidden_dims = [512, 256, 128],
 *                               RL_Environment/tasks/legged_config_ppo.py:5-9) with the parameters of a loaded
 *                               state_dict; weights[l] is layer l's torch Linear.weight, [dims[l+1]][dims[l]] row-major.
 *                               act_scale / act_const = Parameters.MPC_param_scale / MPC_param_const
 *                               (MPC_Controller/Parameters.py:25-33).
 *   mpc_policy_step          <- WeightPolicy.step (:94-118): actions = actor(obs); weights = clamp(actions, -1, 1) *
 *                               scale + const.  d_obs [n, dims[0]], d_actions [n, dims[L]] (raw actor output, may be
 *                               NULL), d_weights [n, dims[L]]; float32, fp32 arithmetic (MFMA f32).
 *   mpc_policy_observations  <- WeightPolicy.compute_observations (:120-139): 48 floats per robot =
 *                               vBody * lin, omegaBody * ang, -ground_normal_yaw, commands * (lin, lin, ang),
 *                               dof_pos * dof_pos_scale, dof_vel * dof_vel_scale, previous actions.
 *                               d_est [n, 18] = vBody3, omegaBody3, rpyBody3, ground_R_body_frame9 (the record
 *                               mpc_ctrl_step takes); scales4 = {lin, ang, dof_pos, dof_vel} (HOST).
 *   mpc_ctrl_estimate        <- the StateEstimate the reference hands to compute_observations: copies the result of
 *                               the last mpc_ctrl_run's StateEstimator.update ([n, 18]) and the controller's
 *                               ground_normal_yaw ([n, 3], StateEstimator.py:99-143) to caller buffers (either may be NULL).
 *   mpc_ctrl_update_estimate <- StateEstimator.update(body_states) alone (StateEstimator.py:57-97), for callers that need
 *                               the estimate before the controller runs (RobotRunnerPolicy.run, robot_runner/
 *                               RobotRunnerPolicy.py:62-92: update, observations, policy, then the FSM); mpc_ctrl_run /
 *                               mpc_ctrl_run_fsm recompute the same estimate from the same body_states.
 *   mpc_pack_commands        <- np.concatenate((commands, actions_rescale, [0.0])) (RL_Environment/tasks/aliengo.py:251):
 *                               [n, 3] + [n, 12] -> the [n, 16] command record of mpc_ctrl_step / mpc_ctrl_run.
 */
typedef struct mpc_policy mpc_policy;
int mpc_policy_create(mpc_policy **out, int n_layers, const int *dims, const float *const *weights, const float *const *biases,
                      const float *act_scale, const float *act_const);
void mpc_policy_destroy(mpc_policy *p);
int mpc_policy_step(mpc_policy *p, int n, const float *d_obs, float *d_actions, float *d_weights, void *stream);
int mpc_policy_observations(int n, const float *d_dof, const float *d_est, const float *d_ground_normal, const float *d_cmd3,
                            const float *d_prev_actions, const float *scales4, float *d_obs, void *stream);
int mpc_ctrl_update_estimate(mpc_ctrl *c, const float *d_body, void *stream);
int mpc_ctrl_estimate(mpc_ctrl *c, float *d_est, float *d_ground_normal, void *stream);
int mpc_pack_commands(int n, const float *d_cmd3, const float *d_weights12, float *d_cmd16, void *stream);
/* VecTask.pre_physics_step's glue in ONE launch (RL_Environment/tasks/aliengo.py:237-251): actions_rescale = torch.mul(actions, MPC_param_scale).add(MPC_param_const)
 * -- a float32 product then a float32 sum, bit for bit what torch's two kernels give -- packed with the commands into the [n, 16] record.  scale12 / const12: HOST [12]
 * (MPC_Controller/Parameters.py:25-33). */
int mpc_pack_commands_scaled(int n, const float *d_cmd3, const float *d_actions12, const float *scale12, const float *const12, float *d_cmd16, void *stream);
/* RobotRunnerPolicy.run (robot_runner/RobotRunnerPolicy.py:62-92) without copies: the observations straight from the controller's own estimate -- what the last
 * mpc_ctrl_update_estimate / mpc_ctrl_run wrote -- and its ground_normal_yaw (StateEstimator.py:99-143, the value of the previous controller tick, as there);
 * then mpc_ctrl_run_fsm_estimated = mpc_ctrl_run_fsm minus its StateEstimator.update: the caller has run mpc_ctrl_update_estimate on the SAME d_body this tick. */
int mpc_ctrl_policy_observations(mpc_ctrl *c, const float *d_dof, const float *d_cmd3, const float *d_prev_actions, const float *scales4, float *d_obs, void *stream);
int mpc_ctrl_run_fsm_estimated(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, const int *d_request, float *d_torques, void *stream);

/* ---- the optional torque exchange of an env batch that spans the GPUs of a node, as one-shot direct peer writes ------------------------
 *
 * robots shard over the GPUs with NO data-path collective; only a consumer that wants every robot's torques on every device needs an
 * exchange of [n_local, 12] float32 per rank (24.6 KB per GPU at 4096 robots: latency-bound on xGMI).  sharding.ShardedLocomotion issues it
 * either as an RCCL all-gather (torch.distributed) or through these entry points: every rank owns a receive region for the whole batch (fine-grained
 * device memory, two parities), exported as a hipIpc handle; mpc_peer_put is ONE kernel that stores the rank's rows into every rank's region and then
 * raises its epoch flag there (system-scope release); mpc_peer_wait ONE kernel that watches the local flags of all ranks (system-scope acquire; bounded:
 * a rank that never arrives costs a timeout count, not a hang).  Both are stream-ordered, nothing synchronises the host.
 *   Protocol per tick: put(k) ... wait(k) copies the whole batch of epoch k into the caller's buffer.  A rank must wait(k) before it puts k + 1, which
 *   bounds every rank's lead to one epoch -- what makes two parities enough.
 *   The handles (MPC_PEER_HANDLE_BYTES each, ranks in order) travel by whatever the host has (torch.distributed.all_gather_object).
 * There is no reference counterpart (the reference is one process); UNMEASURED across GPUs -- the tests run two processes on one GPU and a one-rank group. */
#define MPC_PEER_HANDLE_BYTES 64
typedef struct mpc_peer mpc_peer;
int mpc_peer_create(mpc_peer **out, int rank, int world, int n_rows_total, int row_bytes);      /* on the calling thread's current HIP device */
int mpc_peer_handle(mpc_peer *p, void *handle64);
int mpc_peer_connect(mpc_peer *p, const void *handles /* [world][MPC_PEER_HANDLE_BYTES], own entry ignored; NULL when world == 1 */);
int mpc_peer_put(mpc_peer *p, const void *d_local, int row_lo, int n_rows, void *stream);
int mpc_peer_wait(mpc_peer *p, void *d_out /* [n_rows_total, row_bytes] of the last put's epoch, or NULL: wait only */, void *stream);
int mpc_peer_timeouts(mpc_peer *p, int *count);      /* synchronises; wait kernels that gave up (2 s) so far */
void mpc_peer_destroy(mpc_peer *p);
const char *mpc_peer_last_error(void);

/* Shader clock of `device` under the solve kernel's own regime (one wave of dependent fp64 FMAs per SIMD on every CU) for about busy_ms
 * milliseconds: *ghz = shader cycles of one workgroup / HIP-event time of the launch, *ms (may be NULL) = that time.  Benchmarks record it next
 * to their numbers: boxes of one pool differ by 10 % in the clock they sustain. */
int mpc_device_clock(int device, int busy_ms, double *ghz, double *ms);

const char *mpc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MPC_BATCH_H */
