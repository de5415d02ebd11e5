// mpcqp_kernels.hip -- batched convex-MPC QP engine for MI355X (gfx950, wave64).  Hand-written HIP; no CPU path.
//
// Replaces, for a batch of B independent robots, the solve the reference performs once per control tick:
// CasADi Opti('conic') -> OSQP on the QP of src/mpc.py:58-173, filled at src/mpc.py:242-255, solved at :258.
//
// This translation unit is the C-ABI of include/mpcqp.h and the dispatch between two device implementations of one algorithm:
//   mpcqp_wrench.h   the engine: wrench-space (Woodbury) form, H = 2 alpha I + T'KT with a 6N x 6N system, one QP per wave
//                    (horizon 10) or per four waves (horizon 20), fp32 or fp64 ADMM, fp64 active-set polish, ADMM-only mode
//   mpcqp_stage.h    stage-wise (Riccati) form of the same engine for any other horizon up to 64 -- the reference's own N = 60
//   mpcqp_common.h   what they share: operator-tuple descriptor, policy constants, the dispatch-order pre-pass;  mpcqp_device.h: DPP helpers
// plus the element-wise kernels around the solve: gait-descriptor expansion, closed-loop roll-out (expand / advance), torque map.
// DESIGN.md has the derivations.

#include "mpcqp_wrench.h"
#include "mpcqp_stage.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace {

// tau[b][l] = J[b][l]^T (-f[b][l]) for the four legs of stage 0 (src/main.py:212-214).  Element-wise, HBM-bound:
// one thread per (robot, leg), 9 + 3 loads and 3 stores; consecutive threads touch consecutive 48 / 12-byte records.
template <typename TIO>
__global__ void __launch_bounds__(256)
mpcqp_torque_kernel(const TIO* __restrict__ u, const TIO* __restrict__ jac, TIO* __restrict__ tau, const int64_t B, const int N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // i = 4 b + leg
  if (i >= 4 * B) return;
  const int64_t b = i / 4;
  const int l = (int)(i % 4);
  const TIO* f = u + b * N * 12 + 3 * l;       // stage-0 force of this leg
  const TIO* J = jac + i * 9;                  // 3x3, row-major, world-frame linear Jacobian block of the leg
  const TIO fx = -f[0], fy = -f[1], fz = -f[2];
#pragma unroll
  for (int q = 0; q < 3; ++q) tau[i * 3 + q] = J[0 * 3 + q] * fx + J[1 * 3 + q] * fy + J[2 * 3 + q] * fz;
}

// Leg kinematics (src/main.py:205-210 asks DART for these): foot position and d foot / d q of one leg from its three joint angles,
// by composing the joint rotations (Rodrigues' formula about the geometry's axes) along the chain torso -> HipX -> HipY -> Knee -> foot.
// One thread per (robot, leg): 3 (+9) loads, 9 (+3) stores, three sincos; fp64 arithmetic for either buffer type (the kernel is
// launch- and HBM-latency sized: 48 B in, 108 B out per thread).
struct LegGeoDev { double hx[4][3], hy[4][3], kn[3], ft[3], ax[3], ay[3]; };

__device__ __forceinline__ void rodrigues(const double (&a)[3], const double ang, double (&R)[9]) {
  double s, c;
  sincos(ang, &s, &c);
  const double t = 1.0 - c;
  R[0] = c + t * a[0] * a[0];        R[1] = t * a[0] * a[1] - s * a[2]; R[2] = t * a[0] * a[2] + s * a[1];
  R[3] = t * a[1] * a[0] + s * a[2]; R[4] = c + t * a[1] * a[1];        R[5] = t * a[1] * a[2] - s * a[0];
  R[6] = t * a[2] * a[0] - s * a[1]; R[7] = t * a[2] * a[1] + s * a[0]; R[8] = c + t * a[2] * a[2];
}
__device__ __forceinline__ void mat3_mul(const double (&A)[9], const double (&Bm)[9], double (&C)[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * Bm[j] + A[3 * i + 1] * Bm[3 + j] + A[3 * i + 2] * Bm[6 + j];
}
__device__ __forceinline__ void mat3_vec(const double (&A)[9], const double (&v)[3], double (&o)[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
__device__ __forceinline__ void cross3(const double (&a)[3], const double (&b)[3], double (&o)[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

template <typename TIO>
__global__ void __launch_bounds__(256)
mpcqp_leg_jacobian_kernel(const TIO* __restrict__ q, const TIO* __restrict__ rot, const LegGeoDev geo, TIO* __restrict__ jac,
                          TIO* __restrict__ foot, const int64_t B) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // i = 4 b + leg
  if (i >= 4 * B) return;
  const int64_t b = i / 4;
  const int l = (int)(i % 4);
  double R1[9], Ry[9], R2[9], R3[9];
  rodrigues(geo.ax, (double)q[3 * i], R1);
  rodrigues(geo.ay, (double)q[3 * i + 1], Ry);
  mat3_mul(R1, Ry, R2);
  rodrigues(geo.ay, (double)q[3 * i + 2], Ry);
  mat3_mul(R2, Ry, R3);
  double hx[3], hy[3], p2[3], p3[3], pf[3], t[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { hx[a] = geo.hx[l][a]; hy[a] = geo.hy[l][a]; }   // (leg-indexed: a scalar-indexed copy per lane)
  mat3_vec(R1, hy, t);
#pragma unroll
  for (int a = 0; a < 3; ++a) p2[a] = hx[a] + t[a];
  mat3_vec(R2, geo.kn, t);
#pragma unroll
  for (int a = 0; a < 3; ++a) p3[a] = p2[a] + t[a];
  mat3_vec(R3, geo.ft, t);
#pragma unroll
  for (int a = 0; a < 3; ++a) pf[a] = p3[a] + t[a];
  double J[9], w[3], dlt[3], col[3];   // column j = (joint axis in the torso frame) x (foot - joint origin)
  mat3_vec(R1, geo.ax, w);
#pragma unroll
  for (int a = 0; a < 3; ++a) dlt[a] = pf[a] - hx[a];
  cross3(w, dlt, col);
#pragma unroll
  for (int a = 0; a < 3; ++a) J[3 * a] = col[a];
  mat3_vec(R2, geo.ay, w);
#pragma unroll
  for (int a = 0; a < 3; ++a) dlt[a] = pf[a] - p2[a];
  cross3(w, dlt, col);
#pragma unroll
  for (int a = 0; a < 3; ++a) J[3 * a + 1] = col[a];
  mat3_vec(R3, geo.ay, w);
#pragma unroll
  for (int a = 0; a < 3; ++a) dlt[a] = pf[a] - p3[a];
  cross3(w, dlt, col);
#pragma unroll
  for (int a = 0; a < 3; ++a) J[3 * a + 2] = col[a];
  if (rot) {   // world <- torso
    double Rb[9], Jw[9], pw[3];
#pragma unroll
    for (int a = 0; a < 9; ++a) Rb[a] = (double)rot[9 * b + a];
    mat3_mul(Rb, J, Jw);
    mat3_vec(Rb, pf, pw);
#pragma unroll
    for (int a = 0; a < 9; ++a) J[a] = Jw[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) pf[a] = pw[a];
  }
#pragma unroll
  for (int a = 0; a < 9; ++a) jac[9 * i + a] = (TIO)J[a];
  if (foot) {
#pragma unroll
    for (int a = 0; a < 3; ++a) foot[3 * i + a] = (TIO)pf[a];
  }
}

// Gait entry point (mpcqp_solve_batch_gait): what MPC.solve computes on the host every tick (src/mpc.py:178-254) from the planner
// queries (src/footstep_planner.py:226-246), for B robots at once, into the engine's own tuple workspace:
//   x_des[k]   = [roll0, pitch0, yaw_start + k d w, com_start + k d v, 0, 0, w, v, g]             (src/mpc.py:202-214)
//   contact[k] = feet_id[step(k)] during that step's first ss ticks, else all stance          (footstep_planner.py:239-246)
//   r[0]       = measured foot - measured com;  r[k>=1] = planned foothold of step(k) - x_des com(k)   (src/mpc.py:218-239)
// with step(k) = min((t_in_step + k) / (ss + ds), S - 1) over the S described steps (past the last one: that step with its time running
// on, all feet in stance -- the planner's clamp, src/footstep_planner.py:226-237).  Element-wise and HBM-bound (about 100 B in, 1.1 KB
// out per QP at N = 10, which the solve kernel then reads from L2): one thread per output element, consecutive threads write
// consecutive addresses.
template <typename TIO>
__global__ void __launch_bounds__(256)
mpcqp_gait_expand_kernel(const FastIn<TIO> in, const double d, const int N, const int S, const int64_t B, TIO* __restrict__ r,
                         uint8_t* __restrict__ contact, TIO* __restrict__ xdes) {
  const int nx = (N + 1) * 13, nr = N * 12, per = nx + nr;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * per) return;
  const int64_t b = t / per;
  const int e = (int)(t - b * per);
  const TIO* ref = in.ref + b * 10;
  if (e < nx) {
    const int k = e / 13, c = e % 13;
    double v;
    if (c < 2) v = (double)ref[c];
    else if (c == 2) v = (double)ref[2] + (double)k * d * (double)ref[9];
    else if (c < 6) v = (double)ref[c] + (double)k * d * (double)ref[6 + (c - 3)];
    else if (c < 8) v = 0.0;
    else if (c == 8) v = (double)ref[9];
    else if (c < 12) v = (double)ref[6 + (c - 9)];
    else v = (double)in.x0[b * 13 + 12];
    xdes[b * nx + e] = (TIO)v;
  } else {
    const int i = e - nx, k = i / 12, l = (i % 12) / 3, a = i % 3;
    const int tis = max(in.gait[b * 4 + 0], 0), ss = max(in.gait[b * 4 + 1], 0), period = max(ss + max(in.gait[b * 4 + 2], 0), 1);
    const int tau = tis + k, st = min(tau / period, S - 1), tin = tau - st * period;
    double v;
    if (k == 0) v = (double)in.feet0[b * 12 + l * 3 + a] - (double)in.x0[b * 13 + 3 + a];
    else v = (double)in.footholds[((b * S + st) * 4 + l) * 3 + a] - ((double)ref[3 + a] + (double)k * d * (double)ref[6 + a]);
    r[b * nr + i] = (TIO)v;
    if (a == 0) contact[b * (N * 4) + k * 4 + l] = (tin < ss) ? (in.feet_id[(b * S + st) * 4 + l] ? 1 : 0) : 1;
  }
}

// ---------------------------------------------------------------------------------------------------- closed-loop roll-out
// mpcqp_rollout: B robots advance T control ticks on the device.  Per tick, per robot -- what
// Lite3Controller.customPreStep / MPC.solve do on the host (src/main.py:130-188, src/mpc.py:176-271), with the DART world replaced
// by the model's own predicted next state (the kinematic stand-in of the plumbing tests):
//   expand   x_des from the rolled-forward reference (src/mpc.py:202-214, velocities zeroed on the last plan step, :178-183),
//            contact masks and planned footholds from the robot's plan table (src/footstep_planner.py:226-246), lever arms
//            (src/mpc.py:218-239; stance feet stand on the plan, swing feet carry no force)
//   solve    the batched QP, warm-started from the previous tick when the engine was created with the warm-start flags
//   advance  x <- X[:,1] (apply the first predicted step), com_start += v d, yaw_start += w d (src/mpc.py:261-262), tick += 1,
//            log the tick's actual / desired state and stage-0 forces (the log's TRACKING PERFORMANCE / FORCES, src/logger.py:22-46)
struct RolloutPlan { const void* pos; const uint8_t* feet_id; const int32_t* meta; };   // pos [B,S,4,3], feet_id [B,S,4], meta [B,4] = S, ss, ds, reserved

template <typename TIO>
__global__ void __launch_bounds__(256)
mpcqp_rollout_expand_kernel(const TIO* __restrict__ x, const TIO* __restrict__ ref, const RolloutPlan plan, const int32_t* __restrict__ tick,
                            const double d, const int N, const int Smax, const int64_t B, TIO* __restrict__ r, uint8_t* __restrict__ contact,
                            TIO* __restrict__ xdes) {
  const int nx = (N + 1) * 13, nr = N * 12, per = nx + nr;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * per) return;
  const int64_t b = t / per;
  const int e = (int)(t - b * per);
  const TIO* rf = ref + b * 10;
  // (the plan table lives in device memory the host cannot inspect: malformed rows are clamped, never indexed with --
  //  1 <= S_b <= S, ss >= 0, ss + ds >= 1, tick >= 0; include/mpcqp.h)
  const int S = min(max(plan.meta[b * 4 + 0], 1), Smax), ss = max(plan.meta[b * 4 + 1], 0), period = max(ss + max(plan.meta[b * 4 + 2], 0), 1);
  const int t0 = max(tick[b], 0);
  const int step0 = min(t0 / period, S - 1);
  const double gate = step0 == S - 1 ? 0.0 : 1.0;              // src/mpc.py:181-183: references zeroed on the last plan step
  if (e < nx) {
    const int k = e / 13, c = e % 13;
    double v;
    if (c < 2) v = (double)rf[c];
    else if (c == 2) v = (double)rf[2] + (double)k * d * gate * (double)rf[9];
    else if (c < 6) v = (double)rf[c] + (double)k * d * gate * (double)rf[6 + (c - 3)];
    else if (c < 8) v = 0.0;
    else if (c == 8) v = gate * (double)rf[9];
    else if (c < 12) v = gate * (double)rf[6 + (c - 9)];
    else v = (double)x[b * 13 + 12];
    xdes[b * nx + e] = (TIO)v;
  } else {
    const int i = e - nx, k = i / 12, l = (i % 12) / 3, a = i % 3;
    const int tau = t0 + k, si = min(tau / period, S - 1), tin = tau - si * period;   // past the plan: the last step, all stance
    const TIO* pos = (const TIO*)plan.pos + ((b * Smax + si) * 4 + l) * 3;
    const double com = k == 0 ? (double)x[b * 13 + 3 + a] : (double)rf[3 + a] + (double)k * d * gate * (double)rf[6 + a];
    r[b * nr + i] = (TIO)((double)pos[a] - com);
    if (a == 0) contact[b * (N * 4) + k * 4 + l] = (tin < ss) ? (plan.feet_id[(b * Smax + si) * 4 + l] ? 1 : 0) : 1;
  }
}

template <typename TIO>
__global__ void __launch_bounds__(256)
mpcqp_rollout_advance_kernel(TIO* __restrict__ x, TIO* __restrict__ ref, const RolloutPlan plan, int32_t* __restrict__ tick, const TIO* __restrict__ X,
                             const TIO* __restrict__ u, const int32_t* __restrict__ status, const double d, const int N, const int64_t B,
                             const int T, const int it, const int Smax, TIO* __restrict__ actual, TIO* __restrict__ desired, TIO* __restrict__ forces,
                             int32_t* __restrict__ solved) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  TIO* rf = ref + b * 10;
  const int S = min(max(plan.meta[b * 4 + 0], 1), Smax), ss = max(plan.meta[b * 4 + 1], 0), period = max(ss + max(plan.meta[b * 4 + 2], 0), 1);
  const int t0 = max(tick[b], 0);
  const double gate = min(t0 / period, S - 1) == S - 1 ? 0.0 : 1.0;
  const size_t row = ((size_t)b * T + it) * 12;
  if (actual) for (int c = 0; c < 12; ++c) actual[row + c] = x[b * 13 + c];                      // logger.log_tracking_data (src/mpc.py:295)
  if (desired) {
    const TIO des[12] = {rf[0], rf[1], rf[2], rf[3], rf[4], rf[5], (TIO)0, (TIO)0, (TIO)(gate * (double)rf[9]),
                         (TIO)(gate * (double)rf[6]), (TIO)(gate * (double)rf[7]), (TIO)(gate * (double)rf[8])};
    for (int c = 0; c < 12; ++c) desired[row + c] = des[c];
  }
  if (forces) for (int c = 0; c < 12; ++c) forces[row + c] = u[(size_t)b * N * 12 + c];          // src/main.py:216-218
  const int st = status[b];
  if (solved) solved[b] = (it == 0 ? 0 : solved[b]) + ((st == MPCQP_STATUS_SOLVED_POLISHED || st == MPCQP_STATUS_SOLVED_ADMM) ? 1 : 0);
  for (int c = 0; c < 12; ++c) x[b * 13 + c] = X[((size_t)b * (N + 1) + 1) * 13 + c];          // the world step: the model's own prediction
  for (int a = 0; a < 3; ++a) rf[3 + a] = (TIO)((double)rf[3 + a] + gate * (double)rf[6 + a] * d);   // src/mpc.py:261
  rf[2] = (TIO)((double)rf[2] + gate * (double)rf[9] * d);                                      // src/mpc.py:262
  tick[b] = tick[b] + 1;
}

}  // namespace

// ======================================================================================================
// C-ABI (include/mpcqp.h)
// ======================================================================================================
struct mpcqp_engine {
  MpcQpConfig cfg;
  DevCfg dev;
  DevCfg* dcfg = nullptr;   // device copy of `dev`
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int* order_mem = nullptr;   // dispatch order: [2 x 32 header ints (class counters, queue head), alternating between calls | ORDER_BUCKETS x order_cap indices]
  int order_phase = 0;        // which header set the next ordered launch counts into
  int order_cap = 0;
  int slots = 0;              // workgroups the device holds at once (2 per CU)
  int listed_max = 4;         // device-fills up to which an ordered launch is one workgroup per QP (MpcQpConfig.listed_max)
  double* wr_K = nullptr;     // wrench-space engine (mpcqp_wrench.h): K_q [6][N][N], K^-1 in tile layout (fp32 / fp64)
  float* wr_kinv32 = nullptr;
  double* wr_kinv64 = nullptr;
  void* roll_mem = nullptr;   // roll-out: u [B,N,12], X [B,N+1,13], status / iters [B] of the current tick
  int64_t roll_cap = 0;
  void* gait_mem = nullptr;   // gait entry point: the expanded operator tuple [r | xdes | contact] of the current batch
  int64_t gait_cap = 0;
  bool wrench_ok = false;     // the configuration admits the wrench-space form (isotropic omega weight, positive velocity weights)
  bool form_ok = false;       // ... the same condition without the horizon-specific tables (stage-wise engine)
  double* stage_ws = nullptr; // stage-wise engine: factor workspace of the resident workgroups
  int stage_slots = 0;        // ... and how many of them the device holds
  float* dual_mem = nullptr;  // warm-started engines: multipliers of the previous solve per batch slot [dual_cap][4 N][5]
  int64_t dual_cap = 0;
  bool timed = false;
  bool quiet = false;         // inside mpcqp_rollout: the per-tick solves record no events (the roll-out times itself as a whole)
  bool ev0_set = false;       // the gait entry point has already recorded the start event (in front of its expansion kernel)
  char err[512];
};

namespace {

bool stage_path_applies(const mpcqp_engine* h);

int fail(mpcqp_engine* e, int code, const char* what, hipError_t he = hipSuccess) {
  if (e) {
    if (he != hipSuccess) snprintf(e->err, sizeof(e->err), "%s: %s", what, hipGetErrorString(he));
    else snprintf(e->err, sizeof(e->err), "%s", what);
  }
  return code;
}

#ifndef MPCQP_DEBUG_DYN_LDS
#define MPCQP_DEBUG_DYN_LDS 0   // occupancy experiments only: extra dynamic LDS per workgroup
#endif
// The order buffer of one ordered launch: header set `order_phase` (zeroed by the previous ordered launch's pre-pass, or at
// allocation), the other set handed to this launch's pre-pass for clearing.
static OrderBuf next_order_buf(mpcqp_engine* e) {
  OrderBuf ob;
  int* hdr = e->order_mem + 32 * e->order_phase;
  ob.cnt = hdr; ob.head = hdr + ORDER_BUCKETS; ob.zero = e->order_mem + 32 * (1 - e->order_phase);
  ob.list = e->order_mem + 64; ob.cap = e->order_cap;
  e->order_phase ^= 1;
  return ob;
}

// Wrench-space engine (mpcqp_wrench.h).  Horizon 10: one QP per wave, 2 waves per SIMD = 8 resident workgroups per CU, queued
// launch form for batches that oversubscribe them.  Horizon 20: one QP per 4-wave workgroup, plain launch form.
template <typename TIO, int N>
hipError_t launch_wrench(mpcqp_engine* e, int64_t B, const FastIn<TIO>& in, void* u, void* X, int32_t* st, int32_t* it,
                         float* res, hipStream_t s) {
  dim3 grid((unsigned)B);
  OrderBuf ob = {nullptr, nullptr, 0, nullptr, nullptr};
  {
    const int64_t slots = (N == 10 ? 4 : 1) * (int64_t)e->slots;   // e->slots = 2 per CU: resident workgroups of the horizon-20 kernel
    if (!(e->cfg.flags & MPCQP_FLAG_NATURAL_ORDER) && slots > 0 && B > slots && e->order_cap >= B) {
      ob = next_order_buf(e);
      hipLaunchKernelGGL((mpcqp_order_kernel<TIO, N>), dim3((unsigned)((B + 63) / 64)), dim3(1024), 0, s, in, (int)B, ob);
      // Up to a few device-fills the hardware's own dispatcher does better with the ordered list than resident workgroups on an
      // atomic queue (B = 4096: 0.53-0.56 ms against 0.60-0.63, tools/order_study.py): a resident wave stays on the SIMD it
      // started on, next to whatever partner it was given, while a fresh workgroup goes where there is room.
      if (B <= (int64_t)e->listed_max * slots) ob.head = nullptr;
      else grid = dim3((unsigned)slots);
    }
  }
  const WrTabs tabs = {e->wr_K, e->wr_kinv32, e->wr_kinv64};
#ifndef MPCQP_DIAG_LDSPAD   // (diagnostic builds only: dynamic LDS that limits the resident workgroups per CU, profiles/r03f_occupancy_study.txt)
#define MPCQP_DIAG_LDSPAD 0
#endif
  if (e->cfg.precision == MPCQP_PREC_MIXED)
    hipLaunchKernelGGL((mpcqp_wrench_solve<double, float, double, TIO, N>), grid, dim3(WG<N>::NT), MPCQP_DIAG_LDSPAD, s, e->dcfg, tabs, in, (TIO*)u,
                       (TIO*)X, st, it, res, ob, (int)B);
  else if (e->dev.refine_admm)   // tight-tolerance ADMM-only runs: the instantiation with a refinement step per linear solve
    hipLaunchKernelGGL((mpcqp_wrench_solve<double, double, double, TIO, N, true>), grid, dim3(WG<N>::NT), 0, s, e->dcfg, tabs, in, (TIO*)u,
                       (TIO*)X, st, it, res, ob, (int)B);
  else
    hipLaunchKernelGGL((mpcqp_wrench_solve<double, double, double, TIO, N>), grid, dim3(WG<N>::NT), 0, s, e->dcfg, tabs, in, (TIO*)u,
                       (TIO*)X, st, it, res, ob, (int)B);
  return hipGetLastError();
}

template <typename TIO>
hipError_t launch_wrench_n(mpcqp_engine* e, int64_t B, const FastIn<TIO>& in, void* u, void* X, int32_t* st, int32_t* it,
                           float* res, hipStream_t s) {
  if (e->cfg.N == 10) return launch_wrench<TIO, 10>(e, B, in, u, X, st, it, res, s);
  return launch_wrench<TIO, 20>(e, B, in, u, X, st, it, res, s);
}

// Stage-wise engine (mpcqp_stage.h): persistent workgroups, one QP at a time each, with a factor workspace per workgroup.
template <typename TIO>
hipError_t launch_stage(mpcqp_engine* e, int64_t B, const FastIn<TIO>& in, void* u, void* X, int32_t* st, int32_t* it, float* res, hipStream_t s) {
  const int64_t slots = e->stage_slots > 0 ? e->stage_slots : 256;
  const dim3 grid((unsigned)(B < slots ? B : slots));
  if (e->cfg.precision == MPCQP_PREC_F64)
    hipLaunchKernelGGL((mpcqp_stage_solve<double, TIO>), grid, dim3(SG_NT), 0, s, e->dcfg, in, (TIO*)u, (TIO*)X, st, it, res, e->stage_ws, e->cfg.N, (int)B);
  else
    hipLaunchKernelGGL((mpcqp_stage_solve<float, TIO>), grid, dim3(SG_NT), 0, s, e->dcfg, in, (TIO*)u, (TIO*)X, st, it, res, e->stage_ws, e->cfg.N, (int)B);
  return hipGetLastError();
}

// Workspace that depends on the batch size: the dispatch-order buffer of the queued launch forms and, for warm-started
// engines, the per-slot multiplier record.  Sized by mpcqp_reserve(); a solve at a larger B than reserved grows them on the
// spot (a device-wide synchronisation + allocation -- the only ones a solve can make, and only the first time).
int reserve_workspace(mpcqp_engine* e, int64_t B) {
  if (B <= 0) return MPCQP_OK;
  const bool stage = stage_path_applies(e);
  if (stage && !e->stage_ws) {   // (fixed size: one factor workspace per resident workgroup)
    const int64_t slots = e->stage_slots > 0 ? e->stage_slots : 256;
    if (hipMalloc((void**)&e->stage_ws, (size_t)slots * SG_WS_DOUBLES * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); e->stage_ws = nullptr; return MPCQP_ENOMEM; }
  }
  if (!stage && e->order_cap < B) {
    int* mem = nullptr;
    const int64_t cap = ((B + 1023) / 1024) * 1024;
    if (hipMalloc(&mem, (size_t)(64 + ORDER_BUCKETS * cap) * sizeof(int)) == hipSuccess) {
      (void)hipDeviceSynchronize();                                                      // queued work may still read the old one
      if (e->order_mem) (void)hipFree(e->order_mem);
      (void)hipMemset(mem, 0, 64 * sizeof(int));                                         // both header sets start cleared
      (void)hipDeviceSynchronize();   // (the fill runs on the null stream: a solve on a non-blocking stream must not overtake it and count into garbage)
      e->order_mem = mem; e->order_cap = (int)cap; e->order_phase = 0;
    } else {
      (void)hipGetLastError();   // tolerated: the batch runs in natural order; do not leave the error for the launch check
    }
  }
  if ((e->cfg.flags & MPCQP_FLAG_WARM_START) && e->dual_cap < B) {
    float* mem = nullptr;
    const size_t per = (size_t)20 * e->cfg.N;   // 4 N leg-stages x 5 rows
    if (hipMalloc(&mem, (size_t)B * per * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return MPCQP_ENOMEM; }
    (void)hipDeviceSynchronize();
    (void)hipMemset(mem, 0, (size_t)B * per * sizeof(float));   // new slots start at zero = no record
    if (e->dual_mem) { (void)hipMemcpy(mem, e->dual_mem, (size_t)e->dual_cap * per * sizeof(float), hipMemcpyDeviceToDevice); (void)hipFree(e->dual_mem); }
    (void)hipDeviceSynchronize();
    e->dual_mem = mem; e->dual_cap = B;
  }
  return MPCQP_OK;
}

// Tuple workspace of the gait entry point, grown like the rest of the batch-dependent workspace.
int reserve_gait(mpcqp_engine* e, int64_t B) {
  if (B <= e->gait_cap) return MPCQP_OK;
  const size_t el = e->cfg.dtype == MPCQP_DTYPE_F64 ? 8 : 4, N = (size_t)e->cfg.N;
  void* mem = nullptr;
  if (hipMalloc(&mem, (size_t)B * ((N * 12 + (N + 1) * 13) * el + N * 4)) != hipSuccess) { (void)hipGetLastError(); return MPCQP_ENOMEM; }
  if (e->gait_mem) { (void)hipDeviceSynchronize(); (void)hipFree(e->gait_mem); }
  e->gait_mem = mem; e->gait_cap = B;
  return MPCQP_OK;
}

// Every entry point runs on the handle's device whatever the caller's current device is, and puts that one back.
struct DeviceGuard {
  int prev = -1; bool switched = false; hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// Which of the two engines serves a configuration.  The dense wrench-space engine needs its per-component tables: horizon 10 or 20 and an
// omega weight that is isotropic in x, y (K block diagonal in the wrench components); everything else -- other horizons, anisotropic
// omega weights (the recursion carries the 2 x 2 coupled weight), MPCQP_FLAG_STAGE_KERNEL -- runs on the stage-wise engine.
bool wrench_path_applies(const mpcqp_engine* h) {
  // (any alpha >= 0: a request below 1e-2 -- the reference's own 0.0 included -- is served by continuation, mpcqp_wrench.h)
  return h->wrench_ok && !(h->cfg.flags & MPCQP_FLAG_STAGE_KERNEL);
}

bool stage_path_applies(const mpcqp_engine* h) { return h->form_ok && !wrench_path_applies(h); }

}  // namespace

// Inverse of a small SPD matrix by Gauss-Jordan with partial pivoting (host, fp64); returns false when singular.
static bool invert_small(int n, const double* A, double* Ai) {
  double* M = new (std::nothrow) double[2 * n * n];
  if (!M) return false;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { M[i * 2 * n + j] = A[i * n + j]; M[i * 2 * n + n + j] = i == j ? 1.0 : 0.0; }
  bool ok = true;
  for (int k = 0; k < n && ok; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i) if (fabs(M[i * 2 * n + k]) > fabs(M[piv * 2 * n + k])) piv = i;
    if (!(fabs(M[piv * 2 * n + k]) > 0)) { ok = false; break; }
    if (piv != k) for (int j = 0; j < 2 * n; ++j) { const double t = M[k * 2 * n + j]; M[k * 2 * n + j] = M[piv * 2 * n + j]; M[piv * 2 * n + j] = t; }
    const double p = 1.0 / M[k * 2 * n + k];
    for (int j = 0; j < 2 * n; ++j) M[k * 2 * n + j] *= p;
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const double f = M[i * 2 * n + k];
      if (f != 0) for (int j = 0; j < 2 * n; ++j) M[i * 2 * n + j] -= f * M[k * 2 * n + j];
    }
  }
  if (ok) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ai[i * n + j] = M[i * 2 * n + n + j];
  delete[] M;
  return ok;
}

// Tables of the wrench-space engine (mpcqp_wrench.h): K_q = 2 (wP_q c1 + wQ_q c0) and its inverse per wrench component, the
// latter in fp64 and rounded to fp32.  Returns false when the configuration does not admit the form.
template <int N>
static bool build_wrench_tables(mpcqp_engine* e, const double* tab /* c0 | c1 */) {
  const MpcQpConfig& c = e->cfg;
  if (c.w[6] != c.w[7]) return false;                       // omega weight must be isotropic in x, y (K block diagonal in q)
  for (int i = 6; i < 12; ++i) if (!(c.w[i] > 0)) return false;   // K_q positive definite
  double* K = new (std::nothrow) double[6 * N * N];
  double* Ki = new (std::nothrow) double[6 * N * N];
  float* Ki32 = new (std::nothrow) float[6 * N * N];
  bool ok = K && Ki && Ki32;
  for (int q = 0; q < 6 && ok; ++q) {
    for (int i = 0; i < N * N; ++i) K[q * N * N + i] = 2.0 * (c.w[q] * tab[N * N + i] + c.w[6 + q] * tab[i]);
    ok = invert_small(N, K + q * N * N, Ki + q * N * N);
  }
  if (ok) {
    for (int i = 0; i < 6 * N * N; ++i) Ki32[i] = (float)Ki[i];
    hipError_t he = hipMalloc((void**)&e->wr_K, sizeof(double) * 6 * N * N);
    if (he == hipSuccess) he = hipMemcpy(e->wr_K, K, sizeof(double) * 6 * N * N, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMalloc((void**)&e->wr_kinv32, sizeof(float) * 6 * N * N);
    if (he == hipSuccess) he = hipMemcpy(e->wr_kinv32, Ki32, sizeof(float) * 6 * N * N, hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMalloc((void**)&e->wr_kinv64, sizeof(double) * 6 * N * N);
    if (he == hipSuccess) he = hipMemcpy(e->wr_kinv64, Ki, sizeof(double) * 6 * N * N, hipMemcpyHostToDevice);
    if (he != hipSuccess) { (void)hipGetLastError(); ok = false; }
  }
  delete[] K; delete[] Ki; delete[] Ki32;
  return ok;
}

static void free_engine(mpcqp_engine* h) {
  if (h->dcfg) (void)hipFree(h->dcfg);
  if (h->order_mem) (void)hipFree(h->order_mem);
  if (h->dual_mem) (void)hipFree(h->dual_mem);
  if (h->gait_mem) (void)hipFree(h->gait_mem);
  if (h->roll_mem) (void)hipFree(h->roll_mem);
  if (h->wr_K) (void)hipFree(h->wr_K);
  if (h->wr_kinv32) (void)hipFree(h->wr_kinv32);
  if (h->wr_kinv64) (void)hipFree(h->wr_kinv64);
  if (h->stage_ws) (void)hipFree(h->stage_ws);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
}

extern "C" {

uint32_t mpcqp_version(void) { return MPCQP_VERSION; }

int mpcqp_default_config(MpcQpConfig* c) {
  if (!c) return MPCQP_EINVAL;
  memset(c, 0, sizeof(*c));
  c->size = (uint32_t)sizeof(*c);
  c->N = 10;
  c->delta = 0.03;
  c->m = 8.885;                                                               // src/mpc.py:71
  c->Ibody_inv[0] = 1.0 / 0.24; c->Ibody_inv[1] = 1.0; c->Ibody_inv[2] = 1.0; // src/mpc.py:73-76
  const double w[13] = {1e4, 2.7e4, 1e4, 2.7e5, 2.7e5, 2.7e5, 1e4, 1e4, 1e4, 1.6e4, 1.6e4, 1.6e4, 0.0};
  memcpy(c->w, w, sizeof(w));                                                 // src/mpc.py:122-134
  c->alpha = 1e-2;           // benchmark default; the reference's 0.0 (src/mpc.py:121) leaves the GRFs non-unique
  c->f_min = 3.0; c->f_max = 100.0;                                           // src/mpc.py:45-46
  c->disc = MPCQP_DISC_EULER;                                                 // src/mpc.py:117
  c->dtype = MPCQP_DTYPE_F32;
  c->precision = MPCQP_PREC_MIXED;
  c->flags = MPCQP_FLAG_POLISH;
  c->rho = 1.0; c->sigma = 1e-6; c->relax = 1.6;
  c->max_iter = 400; c->check_every = 100;   // tuned on MI355X: longer ADMM blocks save polish sweeps (profiles/r01_knob_sweep.txt)
  c->eps_abs = 1e-5; c->eps_rel = 1e-6;
  c->polish_max = 4;
  c->device = 0;
  return MPCQP_OK;
}

int mpcqp_create(const MpcQpConfig* cfg, mpcqp_handle* out) {
  if (!cfg || !out || cfg->size != sizeof(MpcQpConfig)) return MPCQP_EINVAL;
  *out = nullptr;
  mpcqp_engine* e = new (std::nothrow) mpcqp_engine();
  if (!e) return MPCQP_ENOMEM;
  e->err[0] = 0;
  e->cfg = *cfg;
  const int N = cfg->N;
  auto reject = [&](int code) { free_engine(e); return code; };
  if (N < 1 || N > SG_NS) return reject(MPCQP_EINVAL);
  if (cfg->precision < MPCQP_PREC_F32 || cfg->precision > MPCQP_PREC_F64) return reject(MPCQP_EINVAL);
  if (cfg->dtype != MPCQP_DTYPE_F32 && cfg->dtype != MPCQP_DTYPE_F64) return reject(MPCQP_EINVAL);
  if (cfg->disc != MPCQP_DISC_EULER && cfg->disc != MPCQP_DISC_ZOH) return reject(MPCQP_EINVAL);
  if (!(cfg->delta > 0) || !(cfg->m > 0) || !(cfg->rho > 0) || !(cfg->sigma >= 0) || !(cfg->relax > 0 && cfg->relax < 2) ||
      cfg->max_iter < 1 || cfg->check_every < 1 || cfg->polish_max < 0 || !(cfg->alpha >= 0) || !(cfg->f_max >= cfg->f_min))
    return reject(MPCQP_EINVAL);
  for (int i = 0; i < 13; ++i)
    if (!(cfg->w[i] >= 0)) return reject(MPCQP_EINVAL);

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return reject(MPCQP_ENODEV);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return reject(MPCQP_ENODEV);
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return reject(MPCQP_ENODEV);  // gfx950 code objects only
  e->slots = 2 * prop.multiProcessorCount;   // the horizon-20 kernel's 256-VGPR, 4-wave workgroups: two per CU (horizon 10: four times that)
  DeviceGuard guard(cfg->device);            // the caller's current device is restored on return
  if (guard.err != hipSuccess) return reject(MPCQP_EHIP);

  DevCfg& d = e->dev;
  d.delta = cfg->delta; d.inv_m = 1.0 / cfg->m;
  for (int i = 0; i < 3; ++i) d.Ib[i] = cfg->Ibody_inv[i];
  for (int i = 0; i < 12; ++i) { d.w[i] = cfg->w[i]; d.sw[i] = sqrt(cfg->w[i]); }
  d.alpha = cfg->alpha; d.fmin = cfg->f_min; d.fmax = cfg->f_max;
  d.rho = cfg->rho; d.sigma = cfg->sigma; d.relax = cfg->relax;
  d.eps_abs = cfg->eps_abs; d.eps_rel = cfg->eps_rel;
  d.theta = cfg->disc == MPCQP_DISC_ZOH ? 0.5 : 0.0;
  d.max_iter = cfg->max_iter; d.check_every = cfg->check_every; d.polish_max = cfg->polish_max;
  d.flags = cfg->flags;
  // Engine tuning fields of the configuration (0 = default; include/mpcqp.h).  The library reads no environment variables.
  d.alpha_floor = cfg->alpha_floor > 0 ? cfg->alpha_floor : ALPHA_FLOOR;
  // Early rho check (wrench engine): the ratio beyond which a QP is given a larger penalty and a longer block.  The all-fp64 ADMM
  // sees a clean dual residual and larger ratios than the fp32-tile one, whose dual residual carries the solve's rounding noise;
  // chosen on batches of other seeds than the bench's (tools/adapt_sweep.py).
  d.adapt_thr = cfg->adapt_thr > 0 ? cfg->adapt_thr : (cfg->precision == MPCQP_PREC_F64 ? 15.f : (cfg->N > 10 ? 10.f : 6.f));
  // Anderson acceleration of the ADMM blocks (mpcqp_wrench.h): with the polish only -- an ADMM-only run stays OSQP's algorithm 1.
  // The dense engine runs it in its fp32 iterations (MIXED; next to an fp64 iteration tile the history does not fit).
  d.accel_p = !(cfg->flags & MPCQP_FLAG_POLISH) ? 0 : (cfg->accel > 0 ? cfg->accel : (cfg->accel < 0 ? 0 : 5));
  const bool accel_dense = d.accel_p > 0 && cfg->precision != MPCQP_PREC_F64 && (N == 10 || N == 20) && !(cfg->flags & MPCQP_FLAG_STAGE_KERNEL);
  const bool accel_n10 = accel_dense && N == 10;   // (the block lengths below were swept with the acceleration on the dense engine only)
  // The early rho check (iteration 25) was built for the plain iteration: it buys a slowly converging QP a larger penalty and a
  // longer block up front.  With the acceleration it costs more than it brings -- the residual ratio it reads is taken a few iterations
  // after an extrapolation, the rebuild restarts the history, and the extrapolation does for those QPs what the penalty did.  Never
  // flagging: horizon 10 held-out mean 9.9 -> 10.7 M QP/s, B = 65 536 15.6 -> 16.4 M; stage-wise engine at N = 60: the 1000 logged
  // ticks 143 -> 221 k QP/s (MIXED: every tick solved in its first block), synthetic 62 -> 93 k (profiles/r03f_early_check.txt).
  // Horizon 20 (config 5): 1.32 -> 1.65 M.  So where the acceleration runs the check is off unless the caller asks for it with an explicit adapt_thr.
  d.early_check = (cfg->adapt_thr > 0 || !accel_dense) ? 1 : 0;   // (stage-wise engine: decided below, once the engine is known)
  d.accel_restart = (d.accel_p > 0 && cfg->accel_restart > 0) ? cfg->accel_restart : 0;
  // A cold solve's first ADMM block is 0.7 check_every long: most QPs have their active set by then (mean iterations 114 -> 82 at
  // N = 10, B = 65 536: 14.6 -> 16.2 M QP/s, N = 20: +10 %), the others go on in full blocks; at B = 4096, where the launch is as
  // long as its hardest QPs, neutral (eight batches of other seeds, tools/adapt_sweep.py).  With the polish only: an ADMM-only
  // run keeps OSQP's uniform check interval.
  // With the acceleration the active set is there sooner: 0.6 check_every (tools/accel_sweep.py, profiles/r03_accel_sweep.txt).
  d.first_block = cfg->first_block > 0 ? cfg->first_block : (cfg->first_block < 0 ? 0 : ((cfg->flags & MPCQP_FLAG_POLISH) ? ((accel_dense ? 6 : 7) * cfg->check_every) / 10 : 0));
  d.incr_legs = cfg->incr_legs > 0 ? (cfg->incr_legs < MPCQP_W_INCR_LEGS ? cfg->incr_legs : MPCQP_W_INCR_LEGS) : (cfg->incr_legs < 0 ? 0 : MPCQP_W_INCR_LEGS);
  e->listed_max = cfg->listed_max > 0 ? cfg->listed_max : (cfg->listed_max < 0 ? 0 : 4);
  d.patience = cfg->polish_patience > 0 ? cfg->polish_patience : POLISH_PATIENCE;
  // (horizon 10 only by default: at N = 20 an update of the 120 x 120 inverse costs four times as much, and config 5 lost 7 % with the rule on)
  d.cheap_steps = cfg->polish_cheap_steps > 0 ? cfg->polish_cheap_steps : (cfg->polish_cheap_steps < 0 || N > 10 ? 0 : POLISH_CHEAP_STEPS);
  d.cheap_legs = cfg->polish_cheap_legs > 0 ? cfg->polish_cheap_legs : POLISH_CHEAP_LEGS;
  // A QP that the early rho check flags gets a first block three times the normal one at horizon 10 (seven batches of other seeds than
  // the bench's, tools/patience_sweep.py, profiles/r03_hard_block_sweep.txt: x2 / x2.5 / x3 / x3.5 -> 8.62 / 8.63 / 9.15 / 9.13 M QP/s on
  // their mean at B = 4096: fewer of them need a third round, and the launch is as long as its longest QP; B = 65 536 pays 3 % for the
  // extra iterations).  Horizon 20 keeps x2 (not re-swept).
  // With the acceleration the longer block buys nothing (x1 / x1.5 / x2 / x3: same held-out mean within run-to-run spread, the large
  // batch fastest at x1): a flagged QP gets the larger penalty and the normal block.
  d.hard_x10 = cfg->hard_block_x10 > 0 ? cfg->hard_block_x10 : (N > 10 ? 10 * HARD_ITER_FACTOR : (accel_n10 ? 10 : 30));
  // The round that nothing follows (iteration cap reached) used to run its whole polish budget; it now gives up after four steps that
  // fail to halve the KKT violation (horizon 10: 9.04 -> 9.35 M on the held-out mean, the same QPs solved, profiles/r03_last_patience_sweep.txt)
  d.last_patience = cfg->polish_last_patience > 0 ? cfg->polish_last_patience : (cfg->polish_last_patience < 0 || N > 10 ? 0 : 4);
  d.refine_admm = (!(cfg->flags & MPCQP_FLAG_POLISH) && cfg->precision == MPCQP_PREC_F64 && cfg->eps_abs < 1e-6) ? 1 : 0;

  // coefficient tables: c0[j][j'] = delta^2 (N - max(j,j')),
  // c1[j][j'] = delta^4 sum_{k > max(j,j')}^{N} (k-1-j+theta)(k-1-j'+theta)
  double* tab = new (std::nothrow) double[2 * N * N];
  if (!tab) return reject(MPCQP_ENOMEM);
  const double dl = cfg->delta, th = d.theta;
  for (int a = 0; a < N; ++a)
    for (int bq = 0; bq < N; ++bq) {
      const int mx = a > bq ? a : bq;
      tab[a * N + bq] = dl * dl * (double)(N - mx);
      double sacc = 0;
      for (int k = mx + 1; k <= N; ++k) sacc += ((double)(k - 1 - a) + th) * ((double)(k - 1 - bq) + th);
      tab[N * N + a * N + bq] = dl * dl * dl * dl * sacc;
    }
  hipError_t he = hipSuccess;
  e->form_ok = true;   // both engines need positive velocity weights (K positive definite / Pi_N > 0)
  for (int i = 6; i < 12; ++i) e->form_ok = e->form_ok && cfg->w[i] > 0;
  if (he == hipSuccess && (N == 10 || N == 20)) e->wrench_ok = N == 10 ? build_wrench_tables<10>(e, tab) : build_wrench_tables<20>(e, tab);
  // Other horizons (the reference's committed N = 60, src/main.py:37) and MPCQP_FLAG_STAGE_KERNEL: the stage-wise engine.
  if (he == hipSuccess && !e->form_ok) { delete[] tab; return reject(MPCQP_EINVAL); }   // (a zero velocity weight: no engine serves it)
  if (he == hipSuccess && stage_path_applies(e)) {
    // The recursion's solve carries ~10 x the error of the dense fp64 sweep (tools/stage_proto.py), and the Woodbury form amplifies it
    // by 1 / (2 alpha): the alpha = 0 continuation of this engine ends at 2e-5 (objective within 2e-7, states within 9e-5 of the
    // alpha = 0 optimum on the golden log ticks at N = 10 / 20 / 60; at 1e-5 the polish refinement stops contracting at N = 60,
    // tools/stage_floor.py)
    if (!(cfg->alpha_floor > 0)) e->dev.alpha_floor = SG_ALPHA_FLOOR;
    if (!(cfg->adapt_thr > 0) && e->dev.accel_p > 0) e->dev.early_check = 0;   // (accelerated first block: no early rho check, see above)
    else e->dev.early_check = 1;
    int per_cu = 0;
    const bool f64 = e->cfg.precision == MPCQP_PREC_F64;
    hipError_t oe;
    if (cfg->dtype == MPCQP_DTYPE_F64)
      oe = f64 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mpcqp_stage_solve<double, double>, SG_NT, 0)
               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mpcqp_stage_solve<float, double>, SG_NT, 0);
    else
      oe = f64 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mpcqp_stage_solve<double, float>, SG_NT, 0)
               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mpcqp_stage_solve<float, float>, SG_NT, 0);
    if (oe != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
    e->stage_slots = per_cu * prop.multiProcessorCount;
  }
  // MPCQP_PREC_F32 (everything fp32) was the arithmetic of the round-1 kernels; it held a 2e-2 band at horizon 10 and none at 20.  The
  // request is served with MIXED (fp32 tiles / chains, fp64 residuals and polish), which is no slower and meets the 1e-4 band.
  if (e->cfg.precision == MPCQP_PREC_F32) e->cfg.precision = MPCQP_PREC_MIXED;
  delete[] tab;
  if (he == hipSuccess) he = hipMalloc((void**)&e->dcfg, sizeof(DevCfg));
  // the regulariser a solve ends with and the one it starts with (continuation, mpcqp_wrench.h): decided here, not per QP on the device
  e->dev.alpha_target = e->dev.alpha > 0.0 ? e->dev.alpha : ((e->dev.flags & MPCQP_FLAG_POLISH) ? e->dev.alpha_floor : 0.0);
  e->dev.alpha_start = ((e->dev.flags & MPCQP_FLAG_POLISH) && e->dev.alpha < ALPHA_EASY) ? ALPHA_EASY : e->dev.alpha;
  if (he == hipSuccess) he = hipMemcpy(e->dcfg, &e->dev, sizeof(DevCfg), hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he != hipSuccess) return reject(MPCQP_EHIP);
  *out = e;
  return MPCQP_OK;
}

int mpcqp_destroy(mpcqp_handle h) {
  if (!h) return MPCQP_OK;
  DeviceGuard guard(h->cfg.device);
  free_engine(h);
  return MPCQP_OK;
}

const char* mpcqp_last_error(mpcqp_handle h) { return h ? h->err : "null handle"; }

int mpcqp_reserve(mpcqp_handle h, int64_t B) {
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || B > 0x7fffffff) return fail(h, MPCQP_EINVAL, "mpcqp_reserve: batch size out of range");
  DeviceGuard guard(h->cfg.device);
  if (guard.err != hipSuccess) return fail(h, MPCQP_EHIP, "hipSetDevice", guard.err);
  const int rc = reserve_workspace(h, B);
  return rc == MPCQP_OK ? rc : fail(h, rc, "mpcqp_reserve: workspace allocation failed");
}

int mpcqp_solve_batch(mpcqp_handle h, int64_t B, const void* x0, const void* r, const uint8_t* contact, const void* xdes,
                      const void* mu, void* u_out, void* X_out, int32_t* status, int32_t* iters, float* res, void* stream) {
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || B > 0x7fffffff) return fail(h, MPCQP_EINVAL, "mpcqp_solve_batch: batch size out of range");
  if (B > 0 && (!x0 || !r || !contact || !xdes || !mu || !u_out || !status || !iters))
    return fail(h, MPCQP_EINVAL, "mpcqp_solve_batch: null buffer");
  DeviceGuard guard(h->cfg.device);
  if (guard.err != hipSuccess) return fail(h, MPCQP_EHIP, "hipSetDevice", guard.err);
  hipStream_t st = (hipStream_t)stream;
  hipError_t he = hipSuccess;
  const bool stage = stage_path_applies(h), wrench = !stage;
  const bool warm = (h->cfg.flags & MPCQP_FLAG_WARM_START) != 0;   // u_out is read as the initial guess first
  if (reserve_workspace(h, B) != MPCQP_OK) return fail(h, MPCQP_ENOMEM, "mpcqp_solve_batch: workspace allocation failed");
  float* ys = (warm && B > 0) ? h->dual_mem : nullptr;
  const int shift = (h->cfg.flags & MPCQP_FLAG_WARM_SHIFT) ? 1 : 0;
  const bool timing = !(h->cfg.flags & MPCQP_FLAG_NO_TIMING) && !h->quiet;
  if (!h->ev0_set && timing) he = hipEventRecord(h->ev0, st);
  h->ev0_set = false;
  if (he != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventRecord", he);
  if (B > 0 && stage) {
    if (h->cfg.dtype == MPCQP_DTYPE_F64) {
      const FastIn<double> in = {(const double*)x0, (const double*)r, contact, (const double*)xdes, (const double*)mu,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, warm ? (const double*)u_out : nullptr, ys, shift};
      he = launch_stage<double>(h, B, in, u_out, X_out, status, iters, res, st);
    } else {
      const FastIn<float> in = {(const float*)x0, (const float*)r, contact, (const float*)xdes, (const float*)mu,
                                nullptr, nullptr, nullptr, nullptr, nullptr, warm ? (const float*)u_out : nullptr, ys, shift};
      he = launch_stage<float>(h, B, in, u_out, X_out, status, iters, res, st);
    }
    if (he != hipSuccess) return fail(h, MPCQP_EHIP, "kernel launch", he);
  } else if (B > 0 && wrench) {
    if (h->cfg.dtype == MPCQP_DTYPE_F64) {
      const FastIn<double> in = {(const double*)x0, (const double*)r, contact, (const double*)xdes, (const double*)mu,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, warm ? (const double*)u_out : nullptr, ys, shift};
      he = launch_wrench_n<double>(h, B, in, u_out, X_out, status, iters, res, st);
    } else {
      const FastIn<float> in = {(const float*)x0, (const float*)r, contact, (const float*)xdes, (const float*)mu,
                                nullptr, nullptr, nullptr, nullptr, nullptr, warm ? (const float*)u_out : nullptr, ys, shift};
      he = launch_wrench_n<float>(h, B, in, u_out, X_out, status, iters, res, st);
    }
    if (he != hipSuccess) return fail(h, MPCQP_EHIP, "kernel launch", he);
  }
  if (timing) {
    he = hipEventRecord(h->ev1, st);
    if (he != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventRecord", he);
    h->timed = true;
  }
  return MPCQP_OK;
}

int mpcqp_solve_batch_gait_steps(mpcqp_handle h, int64_t B, int32_t S, const void* x0, const void* ref, const void* feet0, const void* footholds,
                                 const int32_t* gait, const uint8_t* feet_id, const void* mu, void* u_out, void* X_out,
                                 int32_t* status, int32_t* iters, float* res, void* stream) {
  if (!h) return MPCQP_EINVAL;
  if (S < 1) return fail(h, MPCQP_EINVAL, "mpcqp_solve_batch_gait_steps: at least one plan step");
  if (B < 0 || B > 0x7fffffff) return fail(h, MPCQP_EINVAL, "mpcqp_solve_batch_gait: batch size out of range");
  if (B > 0 && (!x0 || !ref || !feet0 || !footholds || !gait || !feet_id || !mu || !u_out || !status || !iters))
    return fail(h, MPCQP_EINVAL, "mpcqp_solve_batch_gait: null buffer");
  if (B == 0) return mpcqp_solve_batch(h, 0, x0, nullptr, nullptr, nullptr, mu, u_out, X_out, status, iters, res, stream);
  DeviceGuard guard(h->cfg.device);
  if (guard.err != hipSuccess) return fail(h, MPCQP_EHIP, "hipSetDevice", guard.err);
  if (reserve_gait(h, B) != MPCQP_OK) return fail(h, MPCQP_ENOMEM, "mpcqp_solve_batch_gait: workspace allocation failed");
  // expand the descriptors on the device into the engine's tuple workspace, then the normal solve on that tuple
  const size_t el = h->cfg.dtype == MPCQP_DTYPE_F64 ? 8 : 4, N = (size_t)h->cfg.N;
  char* base = (char*)h->gait_mem;
  void* r = base;
  void* xdes = base + (size_t)B * N * 12 * el;
  uint8_t* contact = (uint8_t*)(base + (size_t)B * (N * 12 + (N + 1) * 13) * el);
  const int64_t total = B * (int64_t)(N * 12 + (N + 1) * 13);
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (!(h->cfg.flags & MPCQP_FLAG_NO_TIMING) && hipEventRecord(h->ev0, st) != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventRecord");
  if (h->cfg.dtype == MPCQP_DTYPE_F64) {
    const FastIn<double> in = {(const double*)x0, nullptr, nullptr, nullptr, (const double*)mu, (const double*)ref, (const double*)feet0,
                               (const double*)footholds, gait, feet_id, nullptr, nullptr, 0};
    hipLaunchKernelGGL((mpcqp_gait_expand_kernel<double>), grid, dim3(256), 0, st, in, h->cfg.delta, (int)N, (int)S, B, (double*)r, contact, (double*)xdes);
  } else {
    const FastIn<float> in = {(const float*)x0, nullptr, nullptr, nullptr, (const float*)mu, (const float*)ref, (const float*)feet0,
                              (const float*)footholds, gait, feet_id, nullptr, nullptr, 0};
    hipLaunchKernelGGL((mpcqp_gait_expand_kernel<float>), grid, dim3(256), 0, st, in, h->cfg.delta, (int)N, (int)S, B, (float*)r, contact, (float*)xdes);
  }
  const hipError_t he = hipGetLastError();
  if (he != hipSuccess) return fail(h, MPCQP_EHIP, "gait expansion kernel launch", he);
  h->ev0_set = true;   // the solve's timing starts in front of the expansion
  return mpcqp_solve_batch(h, B, x0, r, contact, xdes, mu, u_out, X_out, status, iters, res, stream);
}

int mpcqp_solve_batch_gait(mpcqp_handle h, int64_t B, const void* x0, const void* ref, const void* feet0, const void* footholds,
                           const int32_t* gait, const uint8_t* feet_id, const void* mu, void* u_out, void* X_out,
                           int32_t* status, int32_t* iters, float* res, void* stream) {
  return mpcqp_solve_batch_gait_steps(h, B, 2, x0, ref, feet0, footholds, gait, feet_id, mu, u_out, X_out, status, iters, res, stream);
}

int mpcqp_rollout(mpcqp_handle h, int64_t B, int32_t T, int32_t S, void* x, void* ref, const void* plan_pos, const uint8_t* plan_feet_id,
                  const int32_t* plan_meta, int32_t* tick, const void* mu, void* actual, void* desired, void* forces, int32_t* solved,
                  void* stream) {
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || B > 0x7fffffff || T < 0 || S < 1) return fail(h, MPCQP_EINVAL, "mpcqp_rollout: size out of range");
  if (B > 0 && (!x || !ref || !plan_pos || !plan_feet_id || !plan_meta || !tick || !mu)) return fail(h, MPCQP_EINVAL, "mpcqp_rollout: null buffer");
  if (B == 0 || T == 0) return MPCQP_OK;
  DeviceGuard guard(h->cfg.device);
  if (guard.err != hipSuccess) return fail(h, MPCQP_EHIP, "hipSetDevice", guard.err);
  const size_t el = h->cfg.dtype == MPCQP_DTYPE_F64 ? 8 : 4, N = (size_t)h->cfg.N;
  if (reserve_gait(h, B) != MPCQP_OK) return fail(h, MPCQP_ENOMEM, "mpcqp_rollout: workspace allocation failed");
  if (h->roll_cap < B) {
    void* mem = nullptr;
    if (hipMalloc(&mem, (size_t)B * ((N * 12 + (N + 1) * 13) * el + 8)) != hipSuccess) { (void)hipGetLastError(); return fail(h, MPCQP_ENOMEM, "mpcqp_rollout: workspace allocation failed"); }
    if (h->roll_mem) { (void)hipDeviceSynchronize(); (void)hipFree(h->roll_mem); }
    (void)hipMemset(mem, 0, (size_t)B * ((N * 12 + (N + 1) * 13) * el + 8));   // zeros = "no guess" for a warm-started engine
    (void)hipDeviceSynchronize();
    h->roll_mem = mem; h->roll_cap = B;
  }
  char* gb = (char*)h->gait_mem;
  void* r = gb;
  void* xdes = gb + (size_t)B * N * 12 * el;
  uint8_t* contact = (uint8_t*)(gb + (size_t)B * (N * 12 + (N + 1) * 13) * el);
  char* rb = (char*)h->roll_mem;
  void* u = rb;
  void* X = rb + (size_t)B * N * 12 * el;
  int32_t* status = (int32_t*)(rb + (size_t)B * (N * 12 + (N + 1) * 13) * el);
  int32_t* iters = status + B;
  const RolloutPlan plan = {plan_pos, plan_feet_id, plan_meta};
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = B * (int64_t)(N * 12 + (N + 1) * 13);
  const dim3 ge((unsigned)((total + 255) / 256)), ga((unsigned)((B + 255) / 256));
  const bool timing = !(h->cfg.flags & MPCQP_FLAG_NO_TIMING);   // (mpcqp_last_kernel_ms after a roll-out: all T ticks)
  if (timing && hipEventRecord(h->ev0, st) != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventRecord");
  struct Quiet { mpcqp_engine* e; explicit Quiet(mpcqp_engine* e_) : e(e_) { e->quiet = true; } ~Quiet() { e->quiet = false; } } quiet(h);
  for (int it = 0; it < T; ++it) {   // 3 launches per tick on the caller's stream, no host synchronisation and no copies in between
    if (el == 8)
      hipLaunchKernelGGL((mpcqp_rollout_expand_kernel<double>), ge, dim3(256), 0, st, (const double*)x, (const double*)ref, plan, tick, h->cfg.delta,
                         (int)N, (int)S, B, (double*)r, contact, (double*)xdes);
    else
      hipLaunchKernelGGL((mpcqp_rollout_expand_kernel<float>), ge, dim3(256), 0, st, (const float*)x, (const float*)ref, plan, tick, h->cfg.delta,
                         (int)N, (int)S, B, (float*)r, contact, (float*)xdes);
    const int rc = mpcqp_solve_batch(h, B, x, r, contact, xdes, mu, u, X, status, iters, nullptr, stream);
    if (rc != MPCQP_OK) return rc;
    if (el == 8)
      hipLaunchKernelGGL((mpcqp_rollout_advance_kernel<double>), ga, dim3(256), 0, st, (double*)x, (double*)ref, plan, tick, (const double*)X,
                         (const double*)u, status, h->cfg.delta, (int)N, B, (int)T, it, (int)S, (double*)actual, (double*)desired, (double*)forces, solved);
    else
      hipLaunchKernelGGL((mpcqp_rollout_advance_kernel<float>), ga, dim3(256), 0, st, (float*)x, (float*)ref, plan, tick, (const float*)X,
                         (const float*)u, status, h->cfg.delta, (int)N, B, (int)T, it, (int)S, (float*)actual, (float*)desired, (float*)forces, solved);
    const hipError_t he = hipGetLastError();
    if (he != hipSuccess) return fail(h, MPCQP_EHIP, "roll-out kernel launch", he);
  }
  if (timing) {
    if (hipEventRecord(h->ev1, st) != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventRecord");
    h->timed = true;
  }
  return MPCQP_OK;
}

int mpcqp_torque_map(mpcqp_handle h, int64_t B, const void* u, const void* jac, void* tau, void* stream) {
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || B > 0x1fffffff) return fail(h, MPCQP_EINVAL, "mpcqp_torque_map: batch size out of range");
  if (B > 0 && (!u || !jac || !tau)) return fail(h, MPCQP_EINVAL, "mpcqp_torque_map: null buffer");
  if (B == 0) return MPCQP_OK;
  DeviceGuard guard(h->cfg.device);
  if (guard.err != hipSuccess) return fail(h, MPCQP_EHIP, "hipSetDevice", guard.err);
  const dim3 grid((unsigned)((4 * B + 255) / 256));
  if (h->cfg.dtype == MPCQP_DTYPE_F64)
    hipLaunchKernelGGL((mpcqp_torque_kernel<double>), grid, dim3(256), 0, (hipStream_t)stream, (const double*)u, (const double*)jac,
                       (double*)tau, B, h->cfg.N);
  else
    hipLaunchKernelGGL((mpcqp_torque_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)u, (const float*)jac,
                       (float*)tau, B, h->cfg.N);
  const hipError_t he = hipGetLastError();
  if (he != hipSuccess) return fail(h, MPCQP_EHIP, "torque kernel launch", he);
  return MPCQP_OK;
}

int mpcqp_default_leg_geometry(MpcQpLegGeometry* g) {   // lite3_urdf/urdf/Lite3.urdf:44-124 (joint origins and axes; data)
  if (!g) return MPCQP_EINVAL;
  memset(g, 0, sizeof(*g));
  g->size = (uint32_t)sizeof(*g);
  for (int l = 0; l < 4; ++l) {
    g->hip_x[l][0] = l < 2 ? 0.1745 : -0.1745;
    g->hip_x[l][1] = (l % 2 == 0) ? 0.062 : -0.062;
    g->hip_y[l][1] = (l % 2 == 0) ? 0.0985 : -0.0985;
  }
  g->knee[2] = -0.20;
  g->foot[2] = -0.21;
  g->axis_x[0] = -1.0;
  g->axis_y[1] = -1.0;
  return MPCQP_OK;
}

int mpcqp_leg_jacobians(mpcqp_handle h, int64_t B, const void* q, const void* rot, const MpcQpLegGeometry* geo, void* jac, void* foot,
                        void* stream) {
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || B > 0x1fffffff) return fail(h, MPCQP_EINVAL, "mpcqp_leg_jacobians: batch size out of range");
  if (B > 0 && (!q || !jac)) return fail(h, MPCQP_EINVAL, "mpcqp_leg_jacobians: null buffer");
  MpcQpLegGeometry lite3;
  if (!geo) { (void)mpcqp_default_leg_geometry(&lite3); geo = &lite3; }
  if (geo->size != sizeof(MpcQpLegGeometry)) return fail(h, MPCQP_EINVAL, "mpcqp_leg_jacobians: geometry struct size mismatch");
  LegGeoDev g;
  memcpy(g.hx, geo->hip_x, sizeof(g.hx)); memcpy(g.hy, geo->hip_y, sizeof(g.hy));
  memcpy(g.kn, geo->knee, sizeof(g.kn)); memcpy(g.ft, geo->foot, sizeof(g.ft));
  for (int k = 0; k < 2; ++k) {   // unit axes (Rodrigues' formula assumes them)
    const double* a = k ? geo->axis_y : geo->axis_x;
    const double nrm = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (!(nrm > 0) || !std::isfinite(nrm)) return fail(h, MPCQP_EINVAL, "mpcqp_leg_jacobians: zero joint axis");
    for (int c = 0; c < 3; ++c) (k ? g.ay : g.ax)[c] = a[c] / nrm;
  }
  if (B == 0) return MPCQP_OK;
  DeviceGuard guard(h->cfg.device);
  if (guard.err != hipSuccess) return fail(h, MPCQP_EHIP, "hipSetDevice", guard.err);
  const dim3 grid((unsigned)((4 * B + 255) / 256));
  if (h->cfg.dtype == MPCQP_DTYPE_F64)
    hipLaunchKernelGGL((mpcqp_leg_jacobian_kernel<double>), grid, dim3(256), 0, (hipStream_t)stream, (const double*)q, (const double*)rot, g,
                       (double*)jac, (double*)foot, B);
  else
    hipLaunchKernelGGL((mpcqp_leg_jacobian_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)q, (const float*)rot, g,
                       (float*)jac, (float*)foot, B);
  const hipError_t he = hipGetLastError();
  if (he != hipSuccess) return fail(h, MPCQP_EHIP, "leg Jacobian kernel launch", he);
  return MPCQP_OK;
}

int mpcqp_last_kernel_ms(mpcqp_handle h, float* ms) {
  if (!h || !ms) return MPCQP_EINVAL;
  if (h->cfg.flags & MPCQP_FLAG_NO_TIMING) return fail(h, MPCQP_EINVAL, "mpcqp_last_kernel_ms: the handle was created with MPCQP_FLAG_NO_TIMING");
  if (!h->timed) return fail(h, MPCQP_EINVAL, "mpcqp_last_kernel_ms: no solve recorded");
  DeviceGuard guard(h->cfg.device);
  hipError_t he = hipEventSynchronize(h->ev1);
  if (he == hipSuccess) he = hipEventElapsedTime(ms, h->ev0, h->ev1);
  if (he != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventElapsedTime", he);
  return MPCQP_OK;
}

#if defined(MPCQP_STAMPS) || defined(MPCQP_TIMELINE)
extern "C" int mpcqp_debug_read_timeline(unsigned long long* out, int64_t B) {
  if (B > 65536) return MPCQP_EINVAL;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), (size_t)B * 3 * sizeof(unsigned long long)) == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}
#endif
#if defined(MPCQP_STAMPS) || defined(MPCQP_WDBG)
int mpcqp_debug_read_wdbg(double* out2048) {
  if (hipDeviceSynchronize() != hipSuccess) return MPCQP_EHIP;
  return hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_wdbg), 2048 * sizeof(double)) == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}
#endif
#ifdef MPCQP_STAMPS
// diagnostic build only: read and reset the phase counters
int mpcqp_debug_read_stamps(unsigned long long* out32) {
  unsigned long long z[32] = {0};
  if (hipDeviceSynchronize() != hipSuccess) return MPCQP_EHIP;
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamps), sizeof(z)) != hipSuccess) return MPCQP_EHIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return MPCQP_EHIP;
  return MPCQP_OK;
}
#endif

}  // extern "C"
