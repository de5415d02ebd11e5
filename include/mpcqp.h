/*
 * mpcqp.h -- C-ABI of the batched convex-MPC QP engine (libmpcqp.so, MI355X / gfx950).
 *
 * This is the drop-in boundary for ONE path of the reference
 * (Emilianogith/MPC-for-dynamic-locomotion-in-the-MIT-cheetah-3): the QP that src/mpc.py builds once
 * (MPC.__init__, src/mpc.py:25-173) and fills + solves every control tick (MPC.solve, src/mpc.py:176-303).
 * The reference has no native boundary of its own: its solver call is CasADi's Opti('conic') -> "osqp"
 * plugin (src/mpc.py:49-55, 258).  What crosses into that solver is the tuple of seven `opt.set_value`
 * parameters (src/mpc.py:242-255) and what comes back is `sol.value(X)`, `sol.value(U)` (src/mpc.py:265-268).
 * The entry points below are that tuple, batched and in compact form.
 *
 * The same symbols are exported by two libraries:
 *   - libmpcqp.so         (product)  pointers are DEVICE memory on the handle's GPU, work is enqueued on `stream`;
 *   - libmpcqp_oracle.so  (checker, oracle/) pointers are HOST memory, `stream` is ignored, fp64 only.
 *
 * All functions return 0 on success and a negative MPCQP_E* code on API misuse / HIP failure; the message is
 * available from mpcqp_last_error().  Numerical outcomes are per-QP `status` values, never error returns
 * (the reference would raise RuntimeError out of opt.solve(), src/mpc.py:258, with error_on_fail commented
 * out at :53).
 */
#ifndef MPCQP_H_
#define MPCQP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCQP_VERSION 0x00010300 /* 1.3.0: MpcQpConfig grew by accel / accel_restart (Anderson-accelerated ADMM blocks); 1.2.0: round-1 kernels retired
                                      (two engines: wrench-space, stage-wise); 1.1.0: horizons up to 64, tuning fields */

/* return codes */
#define MPCQP_OK 0
#define MPCQP_EINVAL (-1)   /* bad argument / unsupported configuration */
#define MPCQP_EHIP (-2)     /* a HIP runtime call failed */
#define MPCQP_ENOMEM (-3)   /* workspace allocation failed */
#define MPCQP_ENODEV (-4)   /* no usable gfx950 device (the product library never falls back to the CPU) */

/* per-QP status written to status[B] */
#define MPCQP_STATUS_UNSOLVED 0
#define MPCQP_STATUS_SOLVED_POLISHED 1 /* active-set polish passed its KKT check */
#define MPCQP_STATUS_SOLVED_ADMM 2     /* ADMM residuals under eps_abs/eps_rel */
#define MPCQP_STATUS_MAX_ITER 3        /* iteration cap reached; u is the last iterate */
#define MPCQP_STATUS_NONFINITE (-1)    /* non-finite input or iterate; outputs zeroed */

/* discretisation of the continuous single-rigid-body model (src/mpc.py:86-107) */
#define MPCQP_DISC_EULER 0 /* X+ = X + delta (A X + B U): the reference, src/mpc.py:117 */
#define MPCQP_DISC_ZOH 1   /* exact zero-order hold; closed form because A^3 = 0, A^2 B = 0 */

/* element type of x0, r, xdes, mu, u_out, X_out */
#define MPCQP_DTYPE_F32 0
#define MPCQP_DTYPE_F64 1

/* arithmetic of the engine (product library; the oracle is always all-f64) */
#define MPCQP_PREC_F32 0   /* (the all-f32 arithmetic of the round-1 kernels, GRFs within 2e-2: retired -- the product library serves the
                              request with MIXED, which is no slower and meets the 1e-4 band) */
#define MPCQP_PREC_MIXED 1 /* matrix tiles f32; structured residuals, refinement and duals in f64 */
#define MPCQP_PREC_F64 2   /* everything f64 */

/* flags */
#define MPCQP_FLAG_POLISH 1u     /* active-set polish after ADMM (OSQP's `polish`; the reference leaves it off) */
#define MPCQP_FLAG_WARM_START 2u /* primal warm start, the reference's `opt.set_initial(U, sol.value(U))` (src/mpc.py:270-271):
                                    u_out is READ as the initial guess [B,N,12] before it is overwritten with the solution
                                    (so a buffer that is reused from tick to tick seeds every solve with the previous one;
                                    all zeros = no guess).  The engine first tries active-set polish steps on the guess's own
                                    active set and falls back to an ADMM block started from the guess.  The engine also keeps,
                                    per batch slot, the multipliers of its previous solve and starts from those when it has
                                    them (slot b of consecutive calls = the same robot).  Same optimum, same
                                    status / tolerance contract as a cold solve.  Every horizon of the product library (polish on);
                                    the CPU checker accepts the flag and starts cold. */
#define MPCQP_FLAG_WARM_SHIFT 16u    /* with WARM_START: the guess is the PREVIOUS control tick's solution, left in u_out unshifted as the
                                        reference leaves it; the engine uses its stage k + 1 for stage k (last stage repeated) */
#define MPCQP_FLAG_GENERAL_KERNEL 4u /* accepted and ignored: selected the round-1 general kernel, retired in 1.2 (its figures: profiles/r01_*) */
#define MPCQP_FLAG_TILE_KERNEL 32u   /* accepted and ignored: selected the round-1 120 x 120 register-tile kernel, retired in 1.2 */
#define MPCQP_FLAG_NATURAL_ORDER 8u  /* product library: one workgroup per QP in batch order.  By default a batch that
                                        oversubscribes the device (more than 2 QPs per CU) is solved by resident workgroups that pull
                                        QPs dearest-expected-first from a queue (a pre-pass ranks the support patterns by
                                        friction demand): same per-QP results, shorter launch */

#define MPCQP_FLAG_STAGE_KERNEL 128u  /* product library: use the stage-wise (Riccati) engine (mpcqp_stage.h) at horizons 10 and 20 as well, where the
                                        dense wrench-space engine is the default; other horizons (up to 64), and cost weights whose two
                                        horizontal angular-velocity entries differ (w[6] != w[7]), always use it */
#define MPCQP_FLAG_NO_TIMING 64u     /* product library: do not record the HIP event pair around each solve that mpcqp_last_kernel_ms()
                                        reads (two extra packets on the stream per call, ~15 us of a 0.46 ms solve of 4096 QPs);
                                        mpcqp_last_kernel_ms() then returns MPCQP_EINVAL */

/*
 * Problem + solver configuration.  POD, versioned by its leading `size` field (set to sizeof(MpcQpConfig)).
 * mpcqp_default_config() fills the Lite3 constants hard-coded in the reference and the engine defaults.
 */
typedef struct MpcQpConfig {
  uint32_t size;        /* sizeof(MpcQpConfig) */
  int32_t N;            /* horizon, params['N'] (src/mpc.py:30): 1..64.  10 and 20 run on the dense wrench-space engine, every other
                           horizon -- the reference's committed 60 (src/main.py:37) -- on the stage-wise engine */
  double delta;         /* params['world_time_step'] (src/mpc.py:31) */
  double m;             /* 8.885 (src/mpc.py:71) */
  double Ibody_inv[3];  /* diag(1/0.24, 1, 1) (src/mpc.py:73-76) */
  double w[13];         /* state weights (src/mpc.py:122-134) */
  double alpha;         /* force weight; 0.0 in the reference (src/mpc.py:121).  Below 1e-2 the engine solves at 1e-2 and walks the
                           weight down by continuation; 0.0 ends at 3e-6 (2e-5 on the stage-wise engine: objective / states / net
                           wrench of the alpha = 0 optimum to 1e-6 / 1e-4 / 1e-4; the forces themselves are not unique at 0) */
  double f_min, f_max;  /* 3, 100 (src/mpc.py:45-46) */
  int32_t disc;         /* MPCQP_DISC_* */
  int32_t dtype;        /* MPCQP_DTYPE_* of the caller's buffers */
  int32_t precision;    /* MPCQP_PREC_* */
  uint32_t flags;       /* MPCQP_FLAG_* */
  double rho;           /* ADMM penalty on the constraint rows */
  double sigma;         /* ADMM proximal weight on u */
  double relax;         /* over-relaxation in (0,2) */
  int32_t max_iter;     /* ADMM iteration cap K */
  int32_t check_every;  /* ADMM block length between polish attempts (iterations); default 100, tuned for N = 10 --
                           scale both with N / 10 for other horizons (200 / 800 at N = 20), as the Python host layer does.
                           With MPCQP_FLAG_POLISH a cold solve's first block is 0.6 check_every long (0.7 without `accel`; most QPs have
                           their active set by then); ADMM-only runs test termination every check_every iterations, as OSQP does */
  double eps_abs, eps_rel;
  int32_t polish_max;   /* active-set refinement steps per polish attempt */
  int32_t device;       /* HIP device ordinal (product library) */
  /* Engine tuning (product library; the checker ignores them).  0 = the engine's own default, which is what the benchmark runs
     with; none of them changes a result beyond rounding except alpha_floor.  They exist for the measurement scripts under tools/
     and for the tests that compare a mechanism with its fallback -- the library reads no environment variables. */
  int32_t first_block;  /* iterations of a cold solve's first ADMM block; 0: 0.6 check_every (0.7 without `accel`; ADMM-only runs: check_every); -1: check_every */
  int32_t incr_legs;    /* changed leg-stages up to which a polish step updates S^-1 instead of rebuilding it; 0: 8; -1: always rebuild */
  int32_t listed_max;   /* device-fills up to which an ordered launch is one workgroup per QP (beyond: resident workgroups on a queue);
                           0: 4; -1: always queued */
  float adapt_thr;      /* residual ratio at the early rho check (iteration 25 of a cold solve's first block) beyond which a QP gets a larger
                           penalty and a longer block; 0: per precision -- and NO early check where `accel` runs (horizon 10 MIXED, stage-wise engine) */
  double alpha_floor;   /* where the regulariser continuation of an alpha = 0 request ends; 0: 3e-6 (stage-wise engine: 2e-5) */
  int32_t polish_patience; /* polish steps of a round that may fail to halve the KKT violation before the round gives up; 0: default */
  int32_t polish_cheap_steps; /* ... and the further steps a round may take beyond that as long as each only UPDATES S^-1 on at most
                                 polish_cheap_legs changed leg-stages; 0: default, -1: none */
  int32_t polish_cheap_legs;  /* 0: default */
  int32_t hard_block_x10;     /* a QP that the early rho check flags gets a first block this many TENTHS of first_block long; 0: default
                                 (10 with `accel` at N = 10: the larger penalty, the normal block; 30 without; 20 at N = 20) */
  int32_t polish_last_patience; /* the polish patience of a round that nothing follows (iteration cap reached); 0: default (unlimited: the
                                   round uses its whole budget at N = 20, 4 at N = 10), n > 0: gives up after n steps that fail to halve the KKT violation,
                                   -1: unlimited */
  int32_t accel;        /* Anderson acceleration of the ADMM blocks (with MPCQP_FLAG_POLISH only; an ADMM-only run is OSQP's algorithm 1
                           unchanged): an extrapolation of the ADMM iterate every `accel` iterations from the last four; 0: default (5),
                           -1: off.  Periods that do not divide the early rho check's iteration (25) distort that check: use 5.  Changes
                           the path to the optimum (fewer iterations on slowly converging QPs), not the optimum.  MIXED arithmetic in
                           the dense engine (all-fp64 iterations run without it); a cold solve's first block in the stage-wise engine */
  int32_t accel_restart; /* ... whose history starts afresh every `accel_restart` iterations; 0: never inside a block (the default: periods of
                            15 / 25 / 35 / 50 iterations measured, none better than the sliding window) */
  int32_t reserved0;
} MpcQpConfig;

typedef struct mpcqp_engine* mpcqp_handle;

uint32_t mpcqp_version(void);

/* Fill *cfg with the reference's constants (N=10, delta=0.03 of the benchmark configs) and engine defaults. */
int mpcqp_default_config(MpcQpConfig* cfg);

/*
 * Handles and devices.  A handle belongs to the device cfg->device; every entry point switches to that device for the
 * duration of the call and restores the caller's current device before it returns.  A handle serves ONE stream at a time:
 * its dispatch queue, timing events and warm-start record are per handle, so concurrent solves on two streams need two
 * handles (any number of handles may share a device).  Not thread-safe per handle.
 */
int mpcqp_create(const MpcQpConfig* cfg, mpcqp_handle* out);
int mpcqp_destroy(mpcqp_handle h);

/*
 * Pre-size the batch-dependent workspace (dispatch-order buffer; multiplier record of warm-started engines) for batches
 * of up to B QPs.  Synchronous.  Optional: a solve at a larger B than any reserved or seen before grows the workspace
 * itself, which costs one device-wide synchronisation + allocation inside that call; after mpcqp_reserve(h, B) no solve
 * of at most B QPs allocates or synchronises.
 */
int mpcqp_reserve(mpcqp_handle h, int64_t B);

/*
 * Solve B independent QPs.  Layouts (row-major, batch outermost; T = cfg.dtype):
 *   x0      T  [B,13]        x0_param  (src/mpc.py:61,189-198,242): [rpy(rotvec), com, omega, v, g]
 *   r       T  [B,N,4,3]     foot - com lever arms, legs FL,FR,HL,HR; replaces r1..r4_skew (src/mpc.py:80-83,
 *                            218-246): the skew matrices (src/utils.py:43-56) are expanded inside the engine
 *   contact u8 [B,N,4]       1 = stance; swing_param = 1 - contact (src/mpc.py:248-254)
 *   xdes    T  [B,N+1,13]    x_des (src/mpc.py:119,202-214,255)
 *   mu      T  [B]           friction coefficient, params['mu'] (src/mpc.py:33)
 *   u_out   T  [B,N,12]      sol.value(U) (src/mpc.py:267-268), stage-major, legs FL,FR,HL,HR x (fx,fy,fz)
 *   X_out   T  [B,N+1,13]    sol.value(X) (src/mpc.py:265-266); may be NULL
 *   status  i32[B]           MPCQP_STATUS_*
 *   iters   i32[B]           ADMM iterations used (+ 1000 * polish refinement steps)
 *   res     f32[B,2]         final primal / dual residual (inf-norm); may be NULL
 * `stream` is a hipStream_t (product) or ignored (oracle).  Asynchronous on the stream; the caller owns all
 * buffers; no allocation happens after mpcqp_reserve(h, B) / the first call at a given B.
 */
int mpcqp_solve_batch(mpcqp_handle h, int64_t B, const void* x0, const void* r, const uint8_t* contact,
                      const void* xdes, const void* mu, void* u_out, void* X_out, int32_t* status,
                      int32_t* iters, float* res, void* stream);

/*
 * Same solve, but from compact per-robot gait descriptors: the engine expands them ON THE DEVICE into contact / r / xdes
 * exactly as MPC.solve does on the host every tick (src/mpc.py:178-254) with the planner queries of
 * src/footstep_planner.py:226-246 -- about 100 B per QP cross the boundary instead of 1.1 KB, and no host loop runs.
 *   x0        T  [B,13]     as above
 *   ref       T  [B,10]     roll0, pitch0 (initial['roll'/'pitch']), yaw_start, com_pos_start[3], v_com_ref[3], theta_dot
 *                           (src/mpc.py:36-38,178-183,202-214)
 *   feet0     T  [B,4,3]    measured foot positions, stage 0 (src/mpc.py:223-226)
 *   footholds T  [B,2,4,3]  plan[step]['pos'] of the current and of the next step (src/mpc.py:306-318)
 *   gait      i32[B,4]      ticks elapsed in the current step, ss_duration, ds_duration, reserved (0)
 *   feet_id   u8 [B,2,4]    plan[step]['feet_id'] of the current and of the next step (1 = stance during single support)
 * Stage k lies in step min((t_in_step + k) / (ss + ds), S - 1) of the descriptor, S = 2 here: a horizon that runs past the last
 * described step stays in it with its time running on, i.e. all feet in stance (the planner's own clamp at the end of a plan,
 * src/footstep_planner.py:226-237) -- describe as many steps as the horizon spans (mpcqp_solve_batch_gait_steps) to avoid that.
 * The descriptors are expanded by an element-wise pre-pass into an engine-owned tuple workspace, then solved exactly as
 * mpcqp_solve_batch would.  Outputs as mpcqp_solve_batch.  Any horizon.
 */
int mpcqp_solve_batch_gait(mpcqp_handle h, int64_t B, const void* x0, const void* ref, const void* feet0,
                           const void* footholds, const int32_t* gait, const uint8_t* feet_id, const void* mu,
                           void* u_out, void* X_out, int32_t* status, int32_t* iters, float* res, void* stream);

/*
 * The same with S >= 1 plan steps per robot: footholds T [B,S,4,3], feet_id u8 [B,S,4] (plan[step + s]['pos'] / ['feet_id'] for
 * s = 0..S-1).  A horizon of N stages starting t_in_step ticks into a step of ss + ds ticks spans
 * ceil((t_in_step + N) / (ss + ds)) steps: 5 for the reference's own N = 60 with 15-tick steps (src/main.py:35-37).
 * Negative durations / ticks are clamped to 0 and ss + ds to at least 1.
 */
int mpcqp_solve_batch_gait_steps(mpcqp_handle h, int64_t B, int32_t S, const void* x0, const void* ref, const void* feet0,
                                 const void* footholds, const int32_t* gait, const uint8_t* feet_id, const void* mu,
                                 void* u_out, void* X_out, int32_t* status, int32_t* iters, float* res, void* stream);

/*
 * Closed-loop roll-out: B robots advance T control ticks on the device -- per tick the parameter fill of MPC.solve
 * (src/mpc.py:176-254: x_des from the rolled-forward reference, contact masks / planned footholds from the robot's plan table,
 * src/footstep_planner.py:226-246), the batched solve (warm-started from the previous tick when the engine has the warm-start
 * flags), the "world step" x <- X[:,1] of the kinematic single-rigid-body stand-in (the model's own predicted next state; the
 * reference steps a DART world here, src/main.py:130-188), and the reference roll-forward com_start += v d, yaw_start += w d
 * (src/mpc.py:261-262).  3 launches per tick on `stream`, no host synchronisation and no host copies.
 *   x        T  [B,13]     in: state at the first tick, out: state after T ticks
 *   ref      T  [B,10]     in/out: roll0, pitch0, yaw_start, com_pos_start[3], v_com_ref[3], theta_dot (as mpcqp_solve_batch_gait)
 *   plan_pos T  [B,S,4,3]  plan[step]['pos'] of all S steps (stance feet stand on the plan; swing feet carry no force)
 *   plan_feet_id u8 [B,S,4]  plan[step]['feet_id']
 *   plan_meta i32[B,4]     S_b (steps of this robot's plan, 1 <= S_b <= S), ss_duration (>= 0), ds_duration (>= 0, ss + ds >= 1),
 *                          reserved (0).  The table is read on the device, where the host cannot validate it: values outside
 *                          these ranges (and negative ticks) are clamped into them, never used as indices or divisors.
 *   tick     i32[B]        in/out: control tick of each robot (advanced by T)
 *   mu       T  [B]
 *   actual / desired / forces  T [B,T,12], each may be NULL: the log's TRACKING PERFORMANCE actual / desired rows and the
 *                          stage-0 forces of every tick (src/mpc.py:295, src/main.py:216-218, src/logger.py:22-46)
 *   solved   i32[B]        may be NULL: number of ticks whose QP was reported solved
 * The oracle library exports the symbol and runs the same loop on host memory.
 */
int mpcqp_rollout(mpcqp_handle h, int64_t B, int32_t T, int32_t S, void* x, void* ref, const void* plan_pos,
                  const uint8_t* plan_feet_id, const int32_t* plan_meta, int32_t* tick, const void* mu, void* actual,
                  void* desired, void* forces, int32_t* solved, void* stream);

/*
 * The step right after the solve in the reference's caller (ground_controller, src/main.py:205-214): joint torques of the
 * four legs from the stage-0 forces, tau_leg = J_leg^T (-f_leg).
 *   u    T [B,N,12]    as written by mpcqp_solve_batch (only stage 0 is read)
 *   jac  T [B,4,3,3]   world-frame 3x3 linear Jacobian block of each leg, row-major (the `getLinearJacobian(...)[:, 6:9]`
 *                      etc. slices of src/main.py:205-210; this engine does not compute kinematics)
 *   tau  T [B,4,3]     HipX, HipY, Knee torque per leg
 */
int mpcqp_torque_map(mpcqp_handle h, int64_t B, const void* u, const void* jac, void* tau, void* stream);

/*
 * Leg geometry of the robot: three revolute joints per leg (HipX, HipY, Knee), legs FL, FR, HL, HR.  Data, not code:
 * mpcqp_default_leg_geometry() fills the joint origins and axes of lite3_urdf/urdf/Lite3.urdf:44-124.
 */
typedef struct MpcQpLegGeometry {
  uint32_t size;        /* sizeof(MpcQpLegGeometry) */
  uint32_t reserved;
  double hip_x[4][3];   /* HipX joint origin in the torso frame, per leg */
  double hip_y[4][3];   /* HipY joint origin in the HipX link frame, per leg */
  double knee[3];       /* Knee joint origin in the thigh frame */
  double foot[3];       /* foot (sole) in the shank frame */
  double axis_x[3];     /* HipX joint axis */
  double axis_y[3];     /* HipY and Knee joint axis */
} MpcQpLegGeometry;

int mpcqp_default_leg_geometry(MpcQpLegGeometry* geo);

/*
 * What the reference's caller asks its physics engine for before the torque map (src/main.py:205-210:
 * `lite3.getLinearJacobian(sole, inCoordinatesOf=World)[:, 6:9]` etc.): the world-frame 3x3 linear Jacobian block of each
 * foot with respect to its own leg's joints, for B robots, from the joint angles -- the `jac` operand of mpcqp_torque_map.
 *   q     T [B,4,3]     joint angles (HipX, HipY, Knee) of FL, FR, HL, HR in rad
 *   rot   T [B,3,3]     torso orientation (world <- torso), row-major; NULL = identity (Jacobians in the torso frame)
 *   geo                 host pointer, NULL = the Lite3 (mpcqp_default_leg_geometry)
 *   jac   T [B,4,3,3]   out: d foot / d q of each leg, world orientation, row-major (row = axis, column = joint)
 *   foot  T [B,4,3]     out, may be NULL: foot position relative to the torso origin, world orientation
 * Element-wise; asynchronous on `stream`.  The oracle library exports the symbol and computes the same on host memory.
 */
int mpcqp_leg_jacobians(mpcqp_handle h, int64_t B, const void* q, const void* rot, const MpcQpLegGeometry* geo, void* jac,
                        void* foot, void* stream);

/* Duration in milliseconds of the most recent solve_batch's kernel(s), measured with HIP events recorded on
 * `stream` around the launch (after mpcqp_rollout: around all of its ticks); blocks until that work has finished.
 * MPCQP_EINVAL when the handle was created with MPCQP_FLAG_NO_TIMING.  Oracle: wall time of the call. */
int mpcqp_last_kernel_ms(mpcqp_handle h, float* ms);

const char* mpcqp_last_error(mpcqp_handle h);

#ifdef __cplusplus
}
#endif
#endif /* MPCQP_H_ */
