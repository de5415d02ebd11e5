// mpcqp_common.h -- what the engines share besides mpcqp_device.h: the operator-tuple descriptor, packed-fp32 helpers, the solver
// policy constants, and the dispatch-order pre-pass (dearest-expected-first order of a batch that oversubscribes the device).
#pragma once
#include "mpcqp_device.h"

namespace {

// Packed pairs: a register tile declared in f2 keeps each pair in an aligned register pair, and the rank-1 updates / mat-vecs map
// 1:1 onto v_pk_fma_f32.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 t = {a, b}; return t; }   // NB: (f2)(a, b) would be a cast of a comma expression
__device__ __forceinline__ f2 splat2(float v) { return mk2(v, v); }

// Inputs of one solve: the operator tuple (x0, r, contact, xdes, mu).  The descriptor fields belong to the gait entry point,
// whose expansion kernel (mpcqp_kernels.hip) turns them into a tuple in the engine's workspace before the solve.
template <typename TIO>
struct FastIn {
  const TIO* x0; const TIO* r; const uint8_t* contact; const TIO* xdes; const TIO* mu;             // tuple form
  const TIO* ref; const TIO* feet0; const TIO* footholds; const int32_t* gait; const uint8_t* feet_id;   // gait form
  const TIO* u_init;   // MPCQP_FLAG_WARM_START: primal initial guess [B,N,12] (aliases the output buffer), else null
  float* y_state;      // ... and the engine's per-slot record of the previous solve's multipliers [cap][40][5] (read, then rewritten)
  int shift;           // MPCQP_FLAG_WARM_SHIFT: the guess and the record are one control tick old: use stage k + 1 for stage k
};



#ifndef MPCQP_ADAPT_AT
#define MPCQP_ADAPT_AT 25
#endif
#ifndef MPCQP_ADAPT_THR
#define MPCQP_ADAPT_THR 10.f
#endif
#ifndef MPCQP_HARD_ITER_FACTOR
#define MPCQP_HARD_ITER_FACTOR 2
#endif
constexpr int ADAPT_AT = MPCQP_ADAPT_AT;             // iteration of the single early rho check
constexpr float ADAPT_THR = MPCQP_ADAPT_THR, ADAPT_RHO_MAX = 30.f;
constexpr int HARD_ITER_FACTOR = MPCQP_HARD_ITER_FACTOR;      // ADMM block length of the QPs that trigger it (x check_every)
constexpr int HARD_POLISH_FACTOR = 2;    // ... and their polish-step budget (x polish_max)
#ifndef MPCQP_WARM_POLISH
#define MPCQP_WARM_POLISH 2
#endif
constexpr int WARM_POLISH = MPCQP_WARM_POLISH;   // polish steps tried on a warm-start guess before the first ADMM block
#ifndef MPCQP_WARM_K
#define MPCQP_WARM_K 60
#endif
constexpr int WARM_K = MPCQP_WARM_K;             // length of the first ADMM block when it starts from remembered (u, y)
#ifndef MPCQP_WARM_FRAC10
#define MPCQP_WARM_FRAC10 4
#endif
constexpr int WARM_FRAC10 = MPCQP_WARM_FRAC10;   // ... in tenths of a cold solve's first block (at most WARM_K): 24 iterations at horizon 10
                                                 // (roll-out of 4096 robots x 100 ticks: 2 / 3 / 4 / 6 / 8 / 10 tenths -> 12.9 / 13.5 / 15.0 / 14.3 / 14.0 / 13.4 M
                                                 //  robot-ticks/s, cold solves 13.4 M; profiles/r03f_rollout_warm.txt)
constexpr float WARM_KKT_TOL = 1e-3f;            // (u0, y0) counts as a KKT point when its stationarity residual is below this x |g|

// ------------------------------------------------------------------------------------------------------ dispatch order
// Workgroups are handed out in blockIdx order, and the solve time of a QP varies by 7x (one early rho check, up to four
// ADMM blocks, 1-8 polish steps): with 4096 QPs on 512 workgroup slots the stragglers that start late set the time of
// the launch (measured 1.49 ms against 1.0 ms of evenly spread work).  A pre-pass therefore sorts the QPs into
// ORDER_BUCKETS classes of expected cost and the solve kernel takes the dearest class first.  The predictor is the
// FRICTION DEMAND of the support pattern: for a stage carried by two feet, the horizontal distance d of the com from
// the line through the feet over the com height h is the friction coefficient a static stance would need, so
// (d / h) / mu > 1 means saturated cones, a large active set and slow ADMM convergence (two-legged "amble" support at
// mu = 0.3: 95 % of those QPs trigger the rho adaptation; diagonal "trot" support: 1 %).  Order only: results are per QP.
constexpr int ORDER_BUCKETS = 16;
struct OrderBuf { int* cnt; int* list; int cap; int* head; int* zero; };   // cnt[ORDER_BUCKETS], list[ORDER_BUCKETS][cap], queue head;
                                                                           // zero: the OTHER call's 32 header ints, cleared by this call's pre-pass

__device__ __forceinline__ float support_demand(int nst, const float (&fx)[4], const float (&fy)[4], const float (&fz)[4],
                                                const bool (&st)[4]) {
  if (nst == 2) {
    int a = -1, bidx = -1;
#pragma unroll
    for (int l = 0; l < 4; ++l) { if (st[l]) { if (a < 0) a = l; else bidx = l; } }
    const float dx = fx[a] - fx[bidx], dy = fy[a] - fy[bidx];
    const float d = fabsf(fx[a] * fy[bidx] - fy[a] * fx[bidx]) / fmaxf(sqrtf(dx * dx + dy * dy), 1e-6f);
    const float h = fmaxf(-0.5f * (fz[a] + fz[bidx]), 1e-3f);
    return d / h;
  }
  if (nst == 1) {
#pragma unroll
    for (int l = 0; l < 4; ++l) if (st[l]) return sqrtf(fx[l] * fx[l] + fy[l] * fy[l]) / fmaxf(-fz[l], 1e-3f);
  }
  return 0.f;   // three or four feet (or flight): no friction-limited moment balance
}

// Expected cost of a QP in microseconds above the cheapest one: every stance leg-stage is three pivots in each of the
// ~2.5 sweeps of a solve (2.2 us), and a unit of friction demand (capped at 2) costs 34 us of extra ADMM blocks and polish
// steps -- least squares on measured per-QP times of the bench workload (tools/timeline.py; correlation 0.48, enough for
// the order: the makespan model drops from 1.36 to 1.24 ms, a clairvoyant order reaches 1.19 ms).
__device__ __forceinline__ int cost_class(float nst, float demand_over_mu) {
  const float us = 2.2f * nst + 34.f * fminf(demand_over_mu, 2.f);
  return isfinite(us) ? (int)fminf(us * 0.1f, (float)(ORDER_BUCKETS - 1)) : 0;
}

// 16 lanes per QP, one stage per lane: the loads of a QP go out together, DPP row reductions take the maximum demand and
// the stance count, lane 0 files the QP.  The class counters are bumped once per workgroup of 64 QPs (a global atomic
// per QP on a handful of addresses serialises).
template <typename TIO, int N>
__global__ void __launch_bounds__(1024)
mpcqp_order_kernel(const FastIn<TIO> in, const int B, const OrderBuf ob) {
  __shared__ int lcnt[ORDER_BUCKETS], lbase[ORDER_BUCKETS];
  if (threadIdx.x < ORDER_BUCKETS) lcnt[threadIdx.x] = 0;
  // Two sets of class counters + queue head alternate between calls: this pre-pass counts into one (cleared by the previous call's
  // pre-pass, whose solve has finished with it: one stream per handle) and clears the other for the next call -- no memset launch.
  if (blockIdx.x == 0 && threadIdx.x < 32 && ob.zero) ob.zero[threadIdx.x] = 0;
  __syncthreads();
  const int b = blockIdx.x * 64 + (threadIdx.x >> 4), k0 = threadIdx.x & 15;
  float score = 0.f, cnt = 0.f;
  for (int k = k0; k < N && b < B; k += 16) {   // (horizons beyond 16: a lane takes stages k0, k0 + 16, ..)
    float fx[4], fy[4], fz[4]; bool st[4]; int nst = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      const TIO* rp = in.r + ((size_t)b * N + k) * 12 + 3 * l;
      fx[l] = (float)rp[0]; fy[l] = (float)rp[1]; fz[l] = (float)rp[2];
      st[l] = in.contact[((size_t)b * N + k) * 4 + l] != 0;
      nst += st[l] ? 1 : 0;
    }
    float sc = support_demand(nst, fx, fy, fz, st);
    if (!isfinite(sc)) sc = 0.f;
    score = fmaxf(score, sc);
    cnt += (float)nst;
  }
  score = fmaxf(score, dpp_mov<0xB1>(score));    // max / sum over the row of 16 lanes (all lanes of the wave are active here)
  score = fmaxf(score, dpp_mov<0x4E>(score));
  score = fmaxf(score, dpp_mov<0x141>(score));
  score = fmaxf(score, dpp_mov<0x140>(score));
  cnt += dpp_mov<0xB1>(cnt);
  cnt += dpp_mov<0x4E>(cnt);
  cnt += dpp_mov<0x141>(cnt);
  cnt += dpp_mov<0x140>(cnt);
  int bucket = 0, pos = 0;
  const bool filer = b < B && k0 == 0;
  if (filer) {
    bucket = cost_class(cnt * (10.f / N), score / fmaxf(fabsf((float)in.mu[b]), 1e-3f));
    pos = atomicAdd(&lcnt[bucket], 1);
  }
  __syncthreads();
  if (threadIdx.x < ORDER_BUCKETS) lbase[threadIdx.x] = lcnt[threadIdx.x] ? atomicAdd(&ob.cnt[threadIdx.x], lcnt[threadIdx.x]) : 0;
  __syncthreads();
  if (filer) ob.list[(size_t)bucket * ob.cap + lbase[bucket] + pos] = b;
}

}  // namespace
