// mpcqp_device.h -- device-side helpers shared by the general and the fast-path kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/mpcqp.h"

// Occupancy target for the N = 10 kernels (waves per SIMD -> VGPR cap 128 / 168 / 256); tuned on hardware.
#ifndef MPCQP_WPE
#define MPCQP_WPE 4
#endif

namespace {

#if defined(MPCQP_STAMPS) || defined(MPCQP_WDBG)
__device__ double g_wdbg[2048];   // diagnostic builds: wrench-space engine, setup products of QP 0
#endif
// Diagnostic build only (-DMPCQP_STAMPS -> libmpcqp_stamps.so): per-phase shader-cycle sums over all workgroups.
// Never compiled into libmpcqp.so; the values leave through their own buffer and feed no output.
#ifdef MPCQP_STAMPS
__device__ unsigned long long g_stamps[32];
__device__ unsigned long long g_timeline[65536 * 3];
   // per QP: start tick, end tick, (xcc << 32 | hw_id): who ran it and when
#define STAMP_INIT unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_t1;
#define STAMP(i)                                                       \
  do {                                                                 \
    st_t1 = __builtin_amdgcn_s_memtime();                              \
    if (threadIdx.x == 0) {                                            \
      atomicAdd(&g_stamps[i], st_t1 - st_t0);                          \
      atomicAdd(&g_stamps[16 + i], 1ull);                              \
    }                                                                  \
    st_t0 = st_t1;                                                     \
  } while (0)
#else
#define STAMP_INIT
#define STAMP(i)
#endif

struct DevCfg {
  double delta, inv_m, Ib[3], w[12], sw[12], alpha, fmin, fmax, rho, sigma, relax, eps_abs, eps_rel, theta;
  int max_iter, check_every, polish_max;
  unsigned flags;
  double alpha_floor;   // wrench engine: where the regulariser continuation of an alpha = 0 request ends
  int first_block;      // wrench engine: iterations of a cold solve's first ADMM block (0: check_every)
  int incr_legs;        // wrench engine: changed leg-stages up to which a polish step updates the inverse (0: always rebuild)
  float adapt_thr;      // wrench engine: residual ratio at the early rho check beyond which a QP gets a larger penalty and a longer block
  int patience;         // wrench engine: polish steps of a round that may fail to halve the KKT violation before the round gives up
  int cheap_steps;      // wrench engine: ... and the steps a round may go on beyond that while they only update the inverse
  int cheap_legs;       // wrench engine: ... on at most this many changed leg-stages
  int hard_x10;         // wrench engine: first-block length of a QP the early rho check flags, in tenths of the normal first block
  int last_patience;    // wrench engine: patience of a round that nothing follows (0: unlimited)
  int refine_admm;      // all-fp64 ADMM without polish at tolerances below 1e-6: one refinement step per linear solve
};

// Problem constants staged in LDS in the vector precision (keeps ~100 scalar registers free).
template <typename TV>
struct CfgS {
  TV delta, theta, alpha, inv_m, fmin, fmax;
  TV Ib[3], w[12], sw[12];
};

// Sum over the 8 lanes of a leg group with DPP lane moves (no LDS crossbar): quad butterfly, then half-row mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ T group8_sum(T v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror: lane i <-> 7 - i inside each group of 8
  return v;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }

// Max over the 64 lanes of a full wave: four DPP steps inside each row of 16, then the four row results through
// v_readlane (a __shfl_xor butterfly is six LDS-latency ds_bpermute round trips).
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
  v = fmaxf(v, dpp_mov<0x140>(v));   // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// Workgroup-wide max of Q floats; every thread gets the result.  NaN-propagating via the isnan flag in slot Q-1
// is the caller's business.  Two barriers.
template <int Q, int NW>
__device__ __forceinline__ void block_max(float (&v)[Q], float* red, int tid) {
#pragma unroll
  for (int q = 0; q < Q; ++q) v[q] = wave_max(v[q]);
  if ((tid & 63) == 0) {
#pragma unroll
    for (int q = 0; q < Q; ++q) red[(tid >> 6) * 4 + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    float m = red[q];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w * 4 + q]);
    v[q] = m;
  }
  __syncthreads();
}

// Gradient of the reference cost (src/mpc.py:121-134, + alpha |u|^2) at s.uv, by rollout + adjoint.
// Needs s.uv visible (barrier before the call).  Leaves s.gv, s.Xs (states), s.es (weighted errors); ends
// with a barrier.  Closed forms: for Euler (theta = 0) / ZOH (theta = 1/2)
//   omega_k = omega_0 + d sum_{j<k} tau_j                      v_k = v_0 + d sum_{j<k} a_j + k d g e_z
//   Theta_k = Theta_0 + k d Rz omega_0 + d^2 sum_{j<k} (k-1-j+theta) Rz tau_j
//   p_k     = p_0 + k d v_0 + d^2 sum_{j<k} (k-1-j+theta) a_j + d^2 g (k(k-1)/2 + theta k) e_z
// with tau_j = sum_i tt_i u_i, a_j = sum_i cm_i u_i e_axis(i) over the variables of stage j
// (src/mpc.py:86-117 restated; tt_i = I_hat_inv (r x e_axis), src/mpc.py:78,98-107).
// Every phase is ONE branch-free instruction stream for all lanes: the lane's role only selects array bases, offsets
// and coefficients, and the (j < k) / (k > j) limits of the prefix sums are zero coefficients, not predicated loads
// (role branches + predicated iterations serialised one LDS latency per term: 7.9 k cycles per call, 13 % of a solve).
template <typename SM, typename TV, int N>
__device__ __forceinline__ void struct_grad(SM& s, int tid) {
  static_assert(N % 5 == 0, "prefix sums are chunked by five stages");
  constexpr int n = 12 * N;
  const TV d = s.cf.delta, th = s.cf.theta;
  if (tid < N * 9) {   // per-stage wrench: tau_j (q < 3), Rz tau_j (3..5), mass-scaled force sum a_j (6..8)
    const int j = tid / 9, q = tid % 9, dd = q % 3;
    // (one LDS base + a per-role element offset: selecting between the member arrays by pointer would lose the LDS
    //  address space and turn the reads into flat loads)
    const int roff = q < 3 ? 0 : (q < 6 ? (int)(s.ttr - s.tt) : (int)(s.cm - s.tt));
    const int stride = q < 6 ? 3 : 1, off = roff + (q < 6 ? dd : 0);
    TV acc = 0;
#pragma unroll
    for (int i0 = 0; i0 < 12; i0 += 4) {   // four terms' loads in flight at a time (the caller's register tile is live)
      asm volatile("" ::: "memory");
      TV a[4], u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = s.tt[(12 * j + i0 + i) * stride + off]; u[i] = s.uv[12 * j + i0 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const TV m = (q < 6 || (i0 + i) % 3 == dd) ? (TV)1 : (TV)0;   // a_j only collects its own axis
        acc += (a[i] * m) * u[i];
      }
    }
    s.wr[tid] = acc;
  }
  __syncthreads();
  if (tid < N * 12) {   // states X_k, k = 1..N: prefix sums of the wrench, weighted by (k-1-j+theta) for angles / position
    const int k = tid / 12 + 1, c = tid % 12, dd = c % 3;
    const TV g = s.x0[12];
    const bool weighted = c < 6;
    const int off = c < 3 ? 3 + dd : (c < 6 ? 6 + dd : (c < 9 ? dd : 6 + dd));
    TV acc = 0;
#pragma unroll
    for (int j0 = 0; j0 < N; j0 += 5) {
      asm volatile("" ::: "memory");
      TV w[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) w[j] = s.wr[(j0 + j) * 9 + off];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const TV cf = (j0 + j < k) ? (weighted ? (TV)(k - 1 - j0 - j) + th : (TV)1) : (TV)0;
        acc += cf * w[j];
      }
    }
    TV val;
    if (c < 3) val = s.x0[dd] + (TV)k * d * s.rzw0[dd] + d * d * acc;
    else if (c < 6) {
      val = s.x0[3 + dd] + (TV)k * d * s.x0[9 + dd] + d * d * acc;
      if (dd == 2) val += d * d * g * ((TV)(k * (k - 1)) * (TV)0.5 + th * (TV)k);
    } else if (c < 9) val = s.x0[6 + dd] + d * acc;
    else {
      val = s.x0[9 + dd] + d * acc;
      if (dd == 2) val += (TV)k * d * g;
    }
    s.Xs[k * 12 + c] = val;
    s.es[k * 12 + c] = s.cf.w[c] * (val - s.xd[k * 13 + c]);
  }
  __syncthreads();
  if (tid < N * 9) {   // adjoint of the prefix sums: out = c1 S1 + c2 S2, S1 plain / S2 weighted suffix sums of the errors
    const int j = tid / 9, q = tid % 9, dd = q % 3;
    const int off1 = q < 3 ? 6 + dd : 9 + dd, off2 = q < 6 ? dd : 3 + dd;
    const TV c1 = (q >= 3 && q < 6) ? (TV)0 : (TV)2 * d, c2 = q < 3 ? (TV)0 : (TV)2 * d * d;
    TV a1 = 0, a2 = 0;
#pragma unroll
    for (int k0 = 1; k0 <= N; k0 += 5) {
      asm volatile("" ::: "memory");
      TV e1[5], e2[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) { e1[k] = s.es[(k0 + k) * 12 + off1]; e2[k] = s.es[(k0 + k) * 12 + off2]; }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const TV on = (k0 + k > j) ? (TV)1 : (TV)0;
        a1 += on * e1[k];
        a2 += (on * ((TV)(k0 + k - 1 - j) + th)) * e2[k];
      }
    }
    s.adj[tid] = c1 * a1 + c2 * a2;
  }
  __syncthreads();
  if (tid < n) {
    const int j = tid / 12, a = tid % 3;
    TV gsum = (TV)2 * s.cf.alpha * s.uv[tid];
#pragma unroll
    for (int q = 0; q < 3; ++q) gsum += s.tt[tid * 3 + q] * s.adj[j * 9 + q] + s.ttr[tid * 3 + q] * s.adj[j * 9 + 3 + q];
    gsum += s.cm[tid] * s.adj[j * 9 + 6 + a];
    s.gv[tid] = gsum;
  }
  __syncthreads();
}

// Weighted 12-vector [P(6) | Q(6)] of force variable i (axis a): H_ii' = 2 (c1 P.P' + c0 Q.Q').
template <typename SM, typename TV>
__device__ __forceinline__ void var_pq(const SM& s, int i, int a, TV (&o)[12]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    o[q] = s.cf.sw[q] * s.ttr[i * 3 + q];
    o[3 + q] = (q == a) ? s.cf.sw[3 + q] * s.cm[i] : (TV)0;
    o[6 + q] = s.cf.sw[6 + q] * s.tt[i * 3 + q];
    o[9 + q] = (q == a) ? s.cf.sw[9 + q] * s.cm[i] : (TV)0;
  }
}

}  // namespace
