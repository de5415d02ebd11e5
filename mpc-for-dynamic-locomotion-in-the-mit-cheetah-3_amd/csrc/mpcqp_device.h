// mpcqp_device.h -- device-side helpers shared by the wrench-space and the stage-wise engine (gfx950, wave64).
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
#if defined(MPCQP_STAMPS) || defined(MPCQP_TIMELINE)   // (-DMPCQP_TIMELINE alone: the per-QP timeline without the phase stamps' atomics)
__device__ unsigned long long g_timeline[65536 * 3];
   // per QP: start tick, end tick, (xcc << 32 | hw_id): who ran it and when
#endif
#ifdef MPCQP_STAMPS
__device__ unsigned long long g_stamps[32];
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
  double alpha_target, alpha_start;   // ... the regulariser a solve ends with (alpha, or alpha_floor for a request of 0) and the one it starts with (>= 1e-2 with the polish)
  int first_block;      // wrench engine: iterations of a cold solve's first ADMM block (0: check_every)
  int incr_legs;        // wrench engine: changed leg-stages up to which a polish step updates the inverse (0: always rebuild)
  float adapt_thr;      // wrench engine: residual ratio at the early rho check beyond which a QP gets a larger penalty and a longer block
  int patience;         // wrench engine: polish steps of a round that may fail to halve the KKT violation before the round gives up
  int cheap_steps;      // wrench engine: ... and the steps a round may go on beyond that while they only update the inverse
  int cheap_legs;       // wrench engine: ... on at most this many changed leg-stages
  int hard_x10;         // wrench engine: first-block length of a QP the early rho check flags, in tenths of the normal first block
  int last_patience;    // wrench engine: patience of a round that nothing follows (0: unlimited)
  int refine_admm;      // all-fp64 ADMM without polish at tolerances below 1e-6: one refinement step per linear solve
  int accel_p;          // Anderson acceleration of the ADMM blocks: an extrapolation every accel_p iterations (0: off)
  int early_check;      // the single early rho check of a cold solve's first block (off by default where the acceleration runs)
  int accel_restart;    // iterations after which the acceleration's history starts afresh (0: only with a new matrix)
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

// Lane move with a row mask (rows outside the mask and invalid sources read zero).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov_rows(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
// Sum over the 64 lanes of a full wave, as a wave-uniform value (scalar register): butterfly inside each row of 16, the four row sums
// folded with row_bcast15 / row_bcast31 into row 3, one v_readlane.  Seven vector instructions per value.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror: every lane of a row holds the row's sum
  v += dpp_mov_rows<0x142, 0xA>(v);   // row_bcast15 into rows 1, 3: rows 0+1, 2+3
  v += dpp_mov_rows<0x143, 0xC>(v);   // row_bcast31 into rows 2, 3: row 3 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
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

// Workgroup-wide sum of Q floats (Q <= 12; `red` holds NW * 12 floats); every thread gets the result.  Two barriers.
template <int Q, int NW>
__device__ __forceinline__ void block_sum(float (&v)[Q], float* red, int tid) {
#pragma unroll
  for (int q = 0; q < Q; ++q) v[q] = wave_sum(v[q]);
  if ((tid & 63) == 0) {
#pragma unroll
    for (int q = 0; q < Q; ++q) red[(tid >> 6) * 12 + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    float m = red[q];
    for (int w = 1; w < NW; ++w) m += red[w * 12 + q];
    v[q] = m;
  }
  __syncthreads();
}

}  // namespace
