// mpcqp_kernels.hip -- batched convex-MPC QP engine for MI355X (gfx950, wave64).  Hand-written HIP; no CPU path.
//
// Replaces, for a batch of B independent robots, the solve the reference performs once per control tick:
// CasADi Opti('conic') -> OSQP on the QP of src/mpc.py:58-173, filled at src/mpc.py:242-255, solved at :258.
//
// Mapping (DESIGN.md has the derivations):
//   * one QP per workgroup; the n x n system matrix (n = 12 N force variables) never touches LDS or HBM: it
//     lives in registers as 3 x CW tiles, thread (leg-stage l, column chunk c) owning rows 3l..3l+2 and
//     columns c*CW..c*CW+CW-1 (CT = 8 lanes per leg-stage, so one wave = 8 leg-stages = 2 horizon stages);
//   * the matrix is built from closed forms: because the single-rigid-body A is nilpotent (A^3 = 0, A^2 B = 0)
//     the condensed Hessian is H = 2 alpha I + 2 sum over stage pairs of c1[j][j'] P_i.P_i' + c0[j][j'] Q_i.Q_i'
//     with 6-vectors P_i, Q_i per force variable (weighted angular/linear response at position and velocity
//     level) and two N x N coefficient tables that depend only on (N, delta, discretisation);
//   * M = H + diag is inverted in place, in registers, by the symmetric sweep operator (Gauss-Jordan without
//     pivoting, valid for SPD): per pivot one LDS broadcast of the pivot row + one workgroup barrier;
//   * an ADMM iteration is a register-tile x LDS-vector product, an 8-lane butterfly sum, the per-leg
//     projection onto the friction-pyramid rows in registers, and one barrier;
//   * gradients / residuals / predicted states come from an O(N) rollout + adjoint in the vector precision
//     (f64 in MIXED and F64 modes) -- the dense Hessian is never needed in high precision;
//   * optional active-set polish (OSQP's "polish" idea): the equality-constrained QP on the active rows is
//     solved with the same build+sweep code on transformed per-variable vectors, refined against the f64
//     structured gradient, and accepted only if it passes a KKT check.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/mpcqp.h"

namespace {

// Diagnostic build only (-DMPCQP_STAMPS -> libmpcqp_stamps.so): per-phase shader-cycle sums over all workgroups.
// Never compiled into libmpcqp.so; the values leave through their own buffer and feed no output.
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
};

template <int N_>
struct Geo {
  static constexpr int N = N_;
  static constexpr int n = 12 * N;          // force variables
  static constexpr int NL = 4 * N;          // leg-stages
  static constexpr int CT = 8;              // column chunks = lanes per leg-stage
  static constexpr int CW = n / CT;         // columns per chunk: 15 (N=10), 30 (N=20); multiple of 3
  static constexpr int CWP = (CW + 3) / 4 * 4;  // padded chunk stride in LDS vectors (16 / 32)
  static constexpr int VP = CT * CWP;       // padded vector length
  static constexpr int NT = NL * CT;        // threads per workgroup: 320 / 640
  static constexpr int NW = NT / 64;        // waves
  static_assert(CW % 3 == 0, "a leg's three rows must not straddle column chunks");
  static_assert(NT % 64 == 0, "whole waves");
};

// Problem constants staged in LDS in the vector precision (keeps ~100 scalar registers free).
template <typename TV>
struct CfgS {
  TV delta, theta, alpha, inv_m, fmin, fmax;
  TV Ib[3], w[12], sw[12];
};

template <typename T, typename TV, int N>
struct Smem {
  using G = Geo<N>;
  CfgS<TV> cf;
  TV x0[13];
  TV mu, cy, sy;                    // friction, cos/sin(yaw)
  TV rzw0[3];                       // Rz * omega_0
  TV xd[(N + 1) * 13];              // x_des
  TV rr[N * 12];                    // lever arms
  TV tt[G::n * 3], ttr[G::n * 3];   // angular response per unit force (unrotated / rotated by Rz)
  TV cm[G::n];                      // contact / m
  TV wr[N * 9], Xs[(N + 1) * 12], es[(N + 1) * 12], adj[N * 9];
  TV uv[G::n], gv[G::n], gl[G::n];  // point, gradient at point, linear term g
  T c0[N * N], c1[N * N];
  T pq[G::n * 12];
  T dg[G::n];
  T vbuf[2 * G::VP];
  T rhs[2 * G::VP];
  float red[G::NW * 4];
  uint8_t ct[N * 4];
  uint8_t en[G::n];
};

template <typename T>
__device__ __forceinline__ T group8_sum(T v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
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
template <typename T, typename TV, int N>
__device__ __forceinline__ void struct_grad(Smem<T, TV, N>& s, int tid) {
  constexpr int n = 12 * N;
  const TV d = s.cf.delta, th = s.cf.theta;
  if (tid < N * 9) {
    const int j = tid / 9, q = tid % 9;
    TV acc = 0;
    if (q < 3) {
#pragma unroll
      for (int i = 0; i < 12; ++i) acc += s.tt[(12 * j + i) * 3 + q] * s.uv[12 * j + i];
    } else if (q < 6) {
#pragma unroll
      for (int i = 0; i < 12; ++i) acc += s.ttr[(12 * j + i) * 3 + (q - 3)] * s.uv[12 * j + i];
    } else {
#pragma unroll
      for (int l = 0; l < 4; ++l) acc += s.cm[12 * j + 3 * l + (q - 6)] * s.uv[12 * j + 3 * l + (q - 6)];
    }
    s.wr[tid] = acc;
  }
  __syncthreads();
  if (tid < N * 12) {
    const int k = tid / 12 + 1, c = tid % 12, dd = c % 3;
    const TV g = s.x0[12];
    TV val;
    if (c < 3) {
      TV acc = 0;
      for (int j = 0; j < k; ++j) acc += ((TV)(k - 1 - j) + th) * s.wr[j * 9 + 3 + dd];
      val = s.x0[dd] + (TV)k * d * s.rzw0[dd] + d * d * acc;
    } else if (c < 6) {
      TV acc = 0;
      for (int j = 0; j < k; ++j) acc += ((TV)(k - 1 - j) + th) * s.wr[j * 9 + 6 + dd];
      val = s.x0[3 + dd] + (TV)k * d * s.x0[9 + dd] + d * d * acc;
      if (dd == 2) val += d * d * g * ((TV)(k * (k - 1)) * (TV)0.5 + th * (TV)k);
    } else if (c < 9) {
      TV acc = 0;
      for (int j = 0; j < k; ++j) acc += s.wr[j * 9 + dd];
      val = s.x0[6 + dd] + d * acc;
    } else {
      TV acc = 0;
      for (int j = 0; j < k; ++j) acc += s.wr[j * 9 + 6 + dd];
      val = s.x0[9 + dd] + d * acc;
      if (dd == 2) val += (TV)k * d * g;
    }
    s.Xs[k * 12 + c] = val;
    s.es[k * 12 + c] = s.cf.w[c] * (val - s.xd[k * 13 + c]);
  }
  __syncthreads();
  if (tid < N * 9) {
    const int j = tid / 9, q = tid % 9, dd = q % 3;
    TV out;
    if (q < 3) {
      TV acc = 0;
      for (int k = j + 1; k <= N; ++k) acc += s.es[k * 12 + 6 + dd];
      out = (TV)2 * d * acc;
    } else if (q < 6) {
      TV acc = 0;
      for (int k = j + 1; k <= N; ++k) acc += ((TV)(k - 1 - j) + th) * s.es[k * 12 + dd];
      out = (TV)2 * d * d * acc;
    } else {
      TV a1 = 0, a2 = 0;
      for (int k = j + 1; k <= N; ++k) {
        a1 += s.es[k * 12 + 9 + dd];
        a2 += ((TV)(k - 1 - j) + th) * s.es[k * 12 + 3 + dd];
      }
      out = (TV)2 * d * a1 + (TV)2 * d * d * a2;
    }
    s.adj[tid] = out;
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
template <typename T, typename TV, int N>
__device__ __forceinline__ void var_pq(const Smem<T, TV, N>& s, int i, int a, TV (&o)[12]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    o[q] = s.cf.sw[q] * s.ttr[i * 3 + q];
    o[3 + q] = (q == a) ? s.cf.sw[3 + q] * s.cm[i] : (TV)0;
    o[6 + q] = s.cf.sw[6 + q] * s.tt[i * 3 + q];
    o[9 + q] = (q == a) ? s.cf.sw[9 + q] * s.cm[i] : (TV)0;
  }
}

template <typename T, typename TV, typename TIO, int N>
__global__ void __launch_bounds__(Geo<N>::NT)
mpcqp_solve_kernel(const DevCfg* __restrict__ cfgp, const double* __restrict__ ctab, const TIO* __restrict__ x0g,
                   const TIO* __restrict__ rg, const uint8_t* __restrict__ cg, const TIO* __restrict__ xdg,
                   const TIO* __restrict__ mug, TIO* __restrict__ ug, TIO* __restrict__ Xg,
                   int* __restrict__ statusg, int* __restrict__ itersg, float* __restrict__ resg) {
  using G = Geo<N>;
  constexpr int n = G::n, CT = G::CT, CW = G::CW, CWP = G::CWP, VP = G::VP, NT = G::NT, NW = G::NW;
  __shared__ Smem<T, TV, N> s;
  const DevCfg& cfg = *cfgp;
  const int tid = threadIdx.x;
  const size_t b = blockIdx.x;
  const int leg = tid / CT;          // leg-stage handled by this 8-lane group
  const int cc = tid % CT;           // column chunk
  const int stage = leg / 4;
  const int row0 = 3 * leg;
  const int col0 = cc * CW;
  const int rbase = (row0 / CW) * CWP + row0 % CW;  // padded index of row0 (rows never straddle chunks)

  STAMP_INIT
  // ------------------------------------------------------------------ load the operator tuple (src/mpc.py:242-255)
  int bad = 0;
  for (int i = tid; i < 13; i += NT) { const TV v = (TV)x0g[b * 13 + i]; s.x0[i] = v; bad |= !isfinite(v); }
  for (int i = tid; i < (N + 1) * 13; i += NT) { const TV v = (TV)xdg[b * (N + 1) * 13 + i]; s.xd[i] = v; bad |= !isfinite(v); }
  for (int i = tid; i < N * 12; i += NT) { const TV v = (TV)rg[b * N * 12 + i]; s.rr[i] = v; bad |= !isfinite(v); }
  for (int i = tid; i < N * 4; i += NT) s.ct[i] = cg[b * N * 4 + i] ? 1 : 0;
  for (int i = tid; i < N * N; i += NT) { s.c0[i] = (T)ctab[i]; s.c1[i] = (T)ctab[N * N + i]; }
  if (tid == 0) {
    const TV m = (TV)mug[b];
    s.mu = m;
    bad |= !isfinite(m);
    s.cf.delta = (TV)cfg.delta; s.cf.theta = (TV)cfg.theta; s.cf.alpha = (TV)cfg.alpha; s.cf.inv_m = (TV)cfg.inv_m;
    s.cf.fmin = (TV)cfg.fmin; s.cf.fmax = (TV)cfg.fmax;
  }
  if (tid >= 64 && tid < 76) { s.cf.w[tid - 64] = (TV)cfg.w[tid - 64]; s.cf.sw[tid - 64] = (TV)cfg.sw[tid - 64]; }
  if (tid >= 128 && tid < 131) s.cf.Ib[tid - 128] = (TV)cfg.Ib[tid - 128];
  bad = __syncthreads_or(bad);
  if (bad) {  // uniform: non-finite input -> zero outputs, status -1 (include/mpcqp.h)
    for (int i = tid; i < n; i += NT) ug[b * n + i] = (TIO)0;
    if (Xg) for (int i = tid; i < (N + 1) * 13; i += NT) Xg[b * (N + 1) * 13 + i] = (TIO)0;
    if (tid == 0) {
      statusg[b] = MPCQP_STATUS_NONFINITE;
      itersg[b] = 0;
      if (resg) { resg[2 * b] = 0.f; resg[2 * b + 1] = 0.f; }
    }
    return;
  }
  if (tid == 0) {
    const TV yaw = s.x0[2];  // src/mpc.py:64: linearised at the measured yaw for the whole horizon
    const TV c = cos(yaw), sn = sin(yaw);
    s.cy = c; s.sy = sn;
    s.rzw0[0] = c * s.x0[6] - sn * s.x0[7];
    s.rzw0[1] = sn * s.x0[6] + c * s.x0[7];
    s.rzw0[2] = s.x0[8];
  }
  __syncthreads();
  // ------------------------------------------------------------------ per-variable response vectors (src/mpc.py:71-78, 98-107)
  if (tid < n) {
    const int i = tid, j = i / 12, l = (i % 12) / 3, a = i % 3;
    const bool st = s.ct[j * 4 + l] != 0;
    const TV rx = s.rr[(j * 4 + l) * 3 + 0], ry = s.rr[(j * 4 + l) * 3 + 1], rz = s.rr[(j * 4 + l) * 3 + 2];
    TV cx, cyv, cz;  // r x e_a  (column a of compute_skew(r), src/utils.py:43-56)
    if (a == 0) { cx = 0; cyv = rz; cz = -ry; }
    else if (a == 1) { cx = -rz; cyv = 0; cz = rx; }
    else { cx = ry; cyv = -rx; cz = 0; }
    const TV c = s.cy, sn = s.sy;
    TV bx = (c * cx + sn * cyv) * s.cf.Ib[0], by = (-sn * cx + c * cyv) * s.cf.Ib[1], bz = cz * s.cf.Ib[2];
    TV tx = c * bx - sn * by, ty = sn * bx + c * by, tz = bz;  // I_hat_inv (r x e_a) = Rz diag(Ib) Rz' (.)
    if (!st) { tx = ty = tz = 0; }                              // swing: force pinned to 0 (src/mpc.py:139-144)
    s.tt[i * 3 + 0] = tx; s.tt[i * 3 + 1] = ty; s.tt[i * 3 + 2] = tz;
    s.ttr[i * 3 + 0] = c * tx - sn * ty; s.ttr[i * 3 + 1] = sn * tx + c * ty; s.ttr[i * 3 + 2] = tz;
    s.cm[i] = st ? s.cf.inv_m : (TV)0;
    s.uv[i] = 0;
  }
  __syncthreads();
  struct_grad<T, TV, N>(s, tid);  // gradient at u = 0 is the linear term g
  if (tid < n) s.gl[tid] = s.gv[tid];
  __syncthreads();

  // ------------------------------------------------------------------ per-leg constants and ADMM state (registers, replicated on the 8 lanes)
  const bool stance = s.ct[leg] != 0;
  const TV muv = s.mu;
  const TV fminv = s.cf.fmin, fmaxv = s.cf.fmax, alpha2 = (TV)2 * s.cf.alpha;
  const int max_iter = cfg.max_iter, check_every = cfg.check_every, polish_max = cfg.polish_max;
  const float eps_abs = (float)cfg.eps_abs, eps_rel = (float)cfg.eps_rel;
  const T mu = (T)muv;
  const T BIG = (T)1e30;
  const T lo0 = stance ? (T)fminv : (T)0, hi0 = stance ? (T)fmaxv : (T)0;  // src/mpc.py:151-157
  const T loP = (T)0, hiP = stance ? BIG : (T)0;     // rows f + mu fz >= 0 (src/mpc.py:159-173)
  const T loM = stance ? -BIG : (T)0, hiM = (T)0;    // rows f - mu fz <= 0
  T g3[3];
  TV g3v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { g3v[c] = s.gl[row0 + c]; g3[c] = (T)g3v[c]; }
  float gmaxf;
  {
    float q[1] = {fmaxf(fmaxf(fabsf((float)g3v[0]), fabsf((float)g3v[1])), fabsf((float)g3v[2]))};
    block_max<1, NW>(q, s.red, tid);
    gmaxf = q[0];
  }
  const bool do_polish = (cfg.flags & MPCQP_FLAG_POLISH) && cfg.alpha > 0.0;
  T rho = (T)cfg.rho;
  const T sigma = (T)cfg.sigma, relax = (T)cfg.relax;
  T u3[3] = {0, 0, 0}, z5[5] = {0, 0, 0, 0, 0}, y5[5] = {0, 0, 0, 0, 0};
  TV uu[3] = {0, 0, 0}, yy[5] = {0, 0, 0, 0, 0};       // polish iterate
  TV uf[3] = {0, 0, 0};                                 // final answer
  int zs = 0, xs = 0, ys = 0;                           // active set of this leg (polish)
  TV up3[3] = {0, 0, 0};
  int mode = 0, it = 0, ps = 0, psteps = 0, status = MPCQP_STATUS_MAX_ITER;
  float res_p = 0.f, res_d = 0.f, rho_ratio = 1.f;
  T tile[3][CW];
  STAMP(0);

  for (;;) {
    // ---------------------------------------------------------------- matrix description -> LDS
    if (mode == 1) {
      // primal-dual active-set rule on (uu, yy), rows: 0 fz | 1 fx - mu fz <= 0 | 2 fx + mu fz >= 0 | 3,4 same for fy
      zs = xs = ys = 0;
      if (stance) {
        const TV g0 = uu[2], g1 = uu[0] - muv * uu[2], g2 = uu[0] + muv * uu[2], g3_ = uu[1] - muv * uu[2],
                 g4 = uu[1] + muv * uu[2];
        if (yy[0] + (g0 - fmaxv) > 0) zs = 1;
        else if (yy[0] + (g0 - fminv) < 0) zs = -1;
        const bool hx = yy[1] + g1 > 0, lx = yy[2] + g2 < 0;
        if (hx && lx) xs = (g1 > -g2) ? 1 : -1; else if (hx) xs = 1; else if (lx) xs = -1;
        const bool hy = yy[3] + g3_ > 0, ly = yy[4] + g4 < 0;
        if (hy && ly) ys = (g3_ > -g4) ? 1 : -1; else if (hy) ys = 1; else if (ly) ys = -1;
      }
      up3[0] = up3[1] = up3[2] = 0;
      if (stance && zs != 0) {
        const TV F = zs > 0 ? fmaxv : fminv;
        up3[2] = F;
        if (xs) up3[0] = (TV)xs * muv * F;
        if (ys) up3[1] = (TV)ys * muv * F;
      }
    }
    if (cc == 0) {
      TV px[12], py[12], pz[12];
      var_pq<T, TV, N>(s, row0 + 0, 0, px);
      var_pq<T, TV, N>(s, row0 + 1, 1, py);
      var_pq<T, TV, N>(s, row0 + 2, 2, pz);
      const TV a2 = alpha2;
      TV dx, dy, dz;
      bool ex, ey, ez;
      if (mode == 0) {
        ex = ey = ez = stance;
        dx = dy = stance ? a2 + (TV)sigma + (TV)rho * (TV)2 : (TV)1;
        dz = stance ? a2 + (TV)sigma + (TV)rho * ((TV)1 + (TV)4 * muv * muv) : (TV)1;
      } else {
        ez = stance && zs == 0;
        ex = stance && xs == 0;
        ey = stance && ys == 0;
        if (ez) {  // tied tangential forces ride on the fz slot
#pragma unroll
          for (int q = 0; q < 12; ++q) pz[q] += (TV)xs * muv * px[q] + (TV)ys * muv * py[q];
        }
        dz = ez ? a2 * ((TV)1 + muv * muv * (TV)((xs != 0) + (ys != 0))) : (TV)1;
        dx = ex ? a2 : (TV)1;
        dy = ey ? a2 : (TV)1;
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        s.pq[(row0 + 0) * 12 + q] = ex ? (T)px[q] : (T)0;
        s.pq[(row0 + 1) * 12 + q] = ey ? (T)py[q] : (T)0;
        s.pq[(row0 + 2) * 12 + q] = ez ? (T)pz[q] : (T)0;
      }
      s.dg[row0 + 0] = (T)dx; s.dg[row0 + 1] = (T)dy; s.dg[row0 + 2] = (T)dz;
      s.en[row0 + 0] = ex; s.en[row0 + 1] = ey; s.en[row0 + 2] = ez;
    }
    __syncthreads();
    STAMP(1);

    // ---------------------------------------------------------------- build the register tile
    {
      T Pr[3][12];
#pragma unroll
      for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
        for (int q = 0; q < 12; ++q) Pr[r3][q] = s.pq[(row0 + r3) * 12 + q];
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        asm volatile("" ::: "memory");  // keep the 12-float column loads of different columns from piling up
        const int ic = col0 + c, jc = ic / 12;
        const T k1 = (T)2 * s.c1[stage * N + jc], k0 = (T)2 * s.c0[stage * N + jc];
        T pc[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) pc[q] = s.pq[ic * 12 + q];
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) {
          T dp = 0, dq = 0;
#pragma unroll
          for (int q = 0; q < 6; ++q) { dp += Pr[r3][q] * pc[q]; dq += Pr[r3][6 + q] * pc[6 + q]; }
          T v = k1 * dp + k0 * dq;
          if (ic == row0 + r3) v += s.dg[ic];
          tile[r3][c] = v;
        }
      }
    }
    STAMP(2);
    // ---------------------------------------------------------------- in-register symmetric sweep: tile <- -M^{-1} on enabled vars
    {
      int step = 0;
      for (int kc = 0; kc < CT; ++kc) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          const int k = kc * CW + c;
          if (!s.en[k]) continue;               // uniform: swing / eliminated variables are identity rows
          const int rr = c % 3;                 // compile-time after unrolling
          const int og = k / 3;                 // owner leg-stage of pivot row k
          T* vb = s.vbuf + (step & 1) * VP;
          if (leg == og) {
#pragma unroll
            for (int c2 = 0; c2 < CW; ++c2) vb[cc * CWP + c2] = tile[rr][c2];
          }
          __syncthreads();
          const T p = (T)1 / vb[kc * CWP + c];
          T vr[3], vc[CW];
#pragma unroll
          for (int r3 = 0; r3 < 3; ++r3) vr[r3] = vb[rbase + r3] * p;
#pragma unroll
          for (int c2 = 0; c2 < CW; ++c2) vc[c2] = vb[cc * CWP + c2];
#pragma unroll
          for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
            for (int c2 = 0; c2 < CW; ++c2) tile[r3][c2] -= vr[r3] * vc[c2];
          if (leg == og) {
#pragma unroll
            for (int c2 = 0; c2 < CW; ++c2) tile[rr][c2] = vc[c2] * p;
          }
          if (cc == kc) {
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) tile[r3][c] = vr[r3];
            if (leg == og) tile[rr][c] = -p;
          }
          ++step;
        }
      }
    }

    STAMP(3);
    bool finished = false;
    if (mode == 0) {
      // -------------------------------------------------------------- ADMM (OSQP algorithm 1 on the 5 rows per leg-stage)
      int buf = 0;
      const T inv_rho = (T)1 / rho;
      auto write_rhs = [&](int bsel) {
        T v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) v[i] = rho * z5[i] - y5[i];
        if (cc == 0) {
          T* rb = s.rhs + bsel * VP;
          rb[rbase + 0] = sigma * u3[0] - g3[0] + v[1] + v[2];
          rb[rbase + 1] = sigma * u3[1] - g3[1] + v[3] + v[4];
          rb[rbase + 2] = sigma * u3[2] - g3[2] + v[0] + mu * (-v[1] + v[2] - v[3] + v[4]);
        }
      };
      write_rhs(0);
      __syncthreads();
      bool go_polish = false;
      while (it < max_iter) {
        T acc[3] = {0, 0, 0};
        {
          const T* rb = s.rhs + buf * VP + cc * CWP;
#pragma unroll
          for (int c2 = 0; c2 < CW; ++c2) {
            const T xv = rb[c2];
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) acc[r3] += tile[r3][c2] * xv;
          }
        }
        T ut[3];
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) ut[r3] = -group8_sum(acc[r3]);
        const T zt[5] = {ut[2], ut[0] - mu * ut[2], ut[0] + mu * ut[2], ut[1] - mu * ut[2], ut[1] + mu * ut[2]};
        const T lo[5] = {lo0, loM, loP, loM, loP}, hi[5] = {hi0, hiM, hiP, hiM, hiP};
#pragma unroll
        for (int c = 0; c < 3; ++c) u3[c] = relax * ut[c] + ((T)1 - relax) * u3[c];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const T zr = relax * zt[i] + ((T)1 - relax) * z5[i];
          T zn = zr + y5[i] * inv_rho;
          zn = zn < lo[i] ? lo[i] : (zn > hi[i] ? hi[i] : zn);
          y5[i] += rho * (zr - zn);
          z5[i] = zn;
        }
        buf ^= 1;
        write_rhs(buf);
        ++it;
        __syncthreads();
        if (it % check_every == 0 || it == max_iter) {
          STAMP(4);
          // residuals of the QP at (u, z, y): |Gu - z|_inf, |grad f(u) + G'y|_inf  (OSQP termination test)
          if (cc == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s.uv[row0 + c] = (TV)u3[c];
          }
          __syncthreads();
          struct_grad<T, TV, N>(s, tid);
          const TV gr[3] = {s.gv[row0], s.gv[row0 + 1], s.gv[row0 + 2]};
          const TV U0 = (TV)u3[0], U1 = (TV)u3[1], U2 = (TV)u3[2];
          const TV gu[5] = {U2, U0 - muv * U2, U0 + muv * U2, U1 - muv * U2, U1 + muv * U2};
          const TV Gy[3] = {(TV)y5[1] + (TV)y5[2], (TV)y5[3] + (TV)y5[4],
                            (TV)y5[0] + muv * (-(TV)y5[1] + (TV)y5[2] - (TV)y5[3] + (TV)y5[4])};
          float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            q[0] = fmaxf(q[0], fabsf((float)(gu[i] - (TV)z5[i])));
            q[2] = fmaxf(q[2], fmaxf(fabsf((float)gu[i]), fabsf((float)z5[i])));
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            q[1] = fmaxf(q[1], fabsf((float)(gr[c] + Gy[c])));
            q[3] = fmaxf(q[3], fmaxf(fabsf((float)(gr[c] - g3v[c])), fabsf((float)Gy[c])));
          }
          if (!(isfinite(q[0]) && isfinite(q[1]))) q[0] = q[1] = INFINITY;
          block_max<4, NW>(q, s.red, tid);
          res_p = q[0]; res_d = q[1];
          const float sp = q[2], sd = fmaxf(q[3], gmaxf);
          if (!(isfinite(res_p) && isfinite(res_d))) { status = MPCQP_STATUS_NONFINITE; finished = true; break; }
          // with polish enabled the KKT-checked polish is the only acceptance test: OSQP's residual test is too
          // loose in the weakly-curved (alpha-only) directions of this QP to guarantee 1e-4 on the forces
          if (!do_polish && res_p <= eps_abs + eps_rel * sp && res_d <= eps_abs + eps_rel * sd) {
            status = MPCQP_STATUS_SOLVED_ADMM;
#pragma unroll
            for (int c = 0; c < 3; ++c) uf[c] = (TV)u3[c];
            finished = true;
            break;
          }
          STAMP(5);
          rho_ratio = sqrtf((res_p / fmaxf(sp, 1e-12f)) / fmaxf(res_d / fmaxf(sd, 1e-12f), 1e-30f));
          if (do_polish) { go_polish = true; break; }
        }
      }
      if (!finished) {
        if (go_polish) {
          mode = 1; ps = 0;
#pragma unroll
          for (int c = 0; c < 3; ++c) uu[c] = (TV)u3[c];
#pragma unroll
          for (int i = 0; i < 5; ++i) yy[i] = (TV)y5[i];
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) uf[c] = (TV)u3[c];
          finished = true;  // iteration cap without polish
        }
      }
    } else {
      // -------------------------------------------------------------- polish: equality-constrained QP on the active rows,
      // solved with the swept reduced matrix as preconditioner and refined against the structured gradient
      const bool ez = stance && zs == 0, ex = stance && xs == 0, ey = stance && ys == 0;
      TV v3[3] = {ex ? uu[0] : (TV)0, ey ? uu[1] : (TV)0, ez ? uu[2] : (TV)0};
      TV uc[3];
      auto expand = [&]() {
        uc[0] = up3[0]; uc[1] = up3[1]; uc[2] = up3[2];
        if (ez) {
          uc[2] = v3[2];
          if (xs) uc[0] = (TV)xs * muv * v3[2];
          if (ys) uc[1] = (TV)ys * muv * v3[2];
        }
        if (ex) uc[0] = v3[0];
        if (ey) uc[1] = v3[1];
      };
      expand();
      const float tol_stat = (sizeof(TV) == 8) ? (1e-6f + 1e-9f * gmaxf) : (3e-7f * fmaxf(gmaxf, 1.f));
      float stat = INFINITY, prev_stat = INFINITY;
      TV gr[3] = {0, 0, 0};
      for (int rf = 0;; ++rf) {
        if (cc == 0) {
#pragma unroll
          for (int c = 0; c < 3; ++c) s.uv[row0 + c] = uc[c];
        }
        __syncthreads();
        struct_grad<T, TV, N>(s, tid);
#pragma unroll
        for (int c = 0; c < 3; ++c) gr[c] = s.gv[row0 + c];
        TV rg[3] = {ex ? gr[0] : (TV)0, ey ? gr[1] : (TV)0,
                    ez ? gr[2] + (TV)xs * muv * gr[0] + (TV)ys * muv * gr[1] : (TV)0};
        float q[1] = {fmaxf(fmaxf(fabsf((float)rg[0]), fabsf((float)rg[1])), fabsf((float)rg[2]))};
        if (!isfinite(q[0])) q[0] = INFINITY;
        block_max<1, NW>(q, s.red, tid);
        prev_stat = stat;
        stat = q[0];
        if (stat <= tol_stat || rf >= 10 || !(stat < 0.5f * prev_stat)) break;  // converged / stagnated (uniform)
        if (cc == 0) {
#pragma unroll
          for (int c = 0; c < 3; ++c) s.rhs[rbase + c] = (T)(-rg[c]);
        }
        __syncthreads();
        T acc[3] = {0, 0, 0};
        {
          const T* rb = s.rhs + cc * CWP;
#pragma unroll
          for (int c2 = 0; c2 < CW; ++c2) {
            const T xv = rb[c2];
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) acc[r3] += tile[r3][c2] * xv;
          }
        }
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) v3[r3] += (TV)(-group8_sum(acc[r3]));
        if (!ex) v3[0] = 0;
        if (!ey) v3[1] = 0;
        if (!ez) v3[2] = 0;
        expand();
      }
      STAMP(6);
      // duals from stationarity grad_leg + G_A' y_A = 0, then the KKT check (primal feasibility + dual sign)
      TV yn[5] = {0, 0, 0, 0, 0};
      float viol[3] = {0.f, 0.f, 0.f};  // primal violation, dual-sign violation, |u|
      if (stance) {
        TV zacc = gr[2];
        if (xs > 0) { yn[1] = -gr[0]; zacc += muv * (-yn[1]); }
        else if (xs < 0) { yn[2] = -gr[0]; zacc += muv * yn[2]; }
        if (ys > 0) { yn[3] = -gr[1]; zacc += muv * (-yn[3]); }
        else if (ys < 0) { yn[4] = -gr[1]; zacc += muv * yn[4]; }
        if (zs != 0) yn[0] = -zacc;
        const TV g0 = uc[2], g1 = uc[0] - muv * uc[2], g2 = uc[0] + muv * uc[2], g3_ = uc[1] - muv * uc[2],
                 g4 = uc[1] + muv * uc[2];
        TV pv = fmax(fminv - g0, g0 - fmaxv);
        pv = fmax(pv, fmax(g1, -g2));
        pv = fmax(pv, fmax(g3_, -g4));
        TV dv = fmax(fmax(-yn[1], yn[2]), fmax(-yn[3], yn[4]));
        if (zs > 0) dv = fmax(dv, -yn[0]);
        if (zs < 0) dv = fmax(dv, yn[0]);
        viol[0] = (float)fmax(pv, (TV)0);
        viol[1] = (float)fmax(dv, (TV)0);
        viol[2] = fmaxf(fmaxf(fabsf((float)uc[0]), fabsf((float)uc[1])), fabsf((float)uc[2]));
        if (!(isfinite(viol[0]) && isfinite(viol[1]))) viol[0] = viol[1] = INFINITY;
      }
      block_max<3, NW>(viol, s.red, tid);
      STAMP(7);
      ++psteps;
      const float ftol = (sizeof(TV) == 8) ? 1e-7f : 2e-5f;
      const float acc_stat = (sizeof(TV) == 8) ? (1e-5f + 1e-8f * gmaxf) : (1e-5f * fmaxf(gmaxf, 1.f));
      // dual-sign slack must stay well below alpha-curvature * force tolerance: a wrongly "active" row with multiplier -e
      // moves the forces by ~e / (2 alpha)
      const float dtol = (sizeof(TV) == 8) ? (1e-5f + 1e-9f * gmaxf) : (2e-5f * fmaxf(1.f, gmaxf));
      const bool ok = viol[0] <= ftol * fmaxf(1.f, viol[2]) && viol[1] <= dtol && stat <= acc_stat;
      if (ok) {
        status = MPCQP_STATUS_SOLVED_POLISHED;
#pragma unroll
        for (int c = 0; c < 3; ++c) uf[c] = uc[c];
        res_p = viol[0];
        res_d = fmaxf(viol[1], stat);
        finished = true;
      } else {
        ++ps;
        if (ps < polish_max) {
#pragma unroll
          for (int c = 0; c < 3; ++c) uu[c] = uc[c];
#pragma unroll
          for (int i = 0; i < 5; ++i) yy[i] = yn[i];
        } else if (it >= max_iter) {
#pragma unroll
          for (int c = 0; c < 3; ++c) uf[c] = (TV)u3[c];
          finished = true;  // status stays MAX_ITER; u is the last ADMM iterate
        } else {
          mode = 0;  // back to ADMM; OSQP's rho adaptation comes for free because the matrix is rebuilt anyway
          if (rho_ratio > 2.f || rho_ratio < 0.5f) {
            const float nr = fminf(fmaxf((float)rho * rho_ratio, 1e-4f), 1e4f);
            rho = (T)nr;
          }
        }
      }
    }
    if (finished) break;
  }

  // ------------------------------------------------------------------ outputs (src/mpc.py:265-268)
  if (!stance || status == MPCQP_STATUS_NONFINITE) uf[0] = uf[1] = uf[2] = 0;
  if (cc == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) s.uv[row0 + c] = uf[c];
  }
  __syncthreads();
  for (int i = tid; i < n; i += NT) ug[b * n + i] = (TIO)s.uv[i];
  if (Xg) {
    struct_grad<T, TV, N>(s, tid);
    for (int i = tid; i < (N + 1) * 13; i += NT) {
      const int k = i / 13, c = i % 13;
      const TV v = (c == 12 || k == 0) ? s.x0[c] : s.Xs[k * 12 + c];
      Xg[b * (N + 1) * 13 + i] = (TIO)v;
    }
  }
  STAMP(8);
  if (tid == 0) {
    statusg[b] = status;
    itersg[b] = it + 1000 * psteps;
    if (resg) { resg[2 * b] = res_p; resg[2 * b + 1] = res_d; }
  }
}

}  // namespace

// ======================================================================================================
// C-ABI (include/mpcqp.h)
// ======================================================================================================
struct mpcqp_engine {
  MpcQpConfig cfg;
  DevCfg dev;
  double* ctab = nullptr;   // [2][N][N] coefficient tables on the device
  DevCfg* dcfg = nullptr;   // device copy of `dev`
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  char err[512];
};

namespace {

int fail(mpcqp_engine* e, int code, const char* what, hipError_t he = hipSuccess) {
  if (e) {
    if (he != hipSuccess) snprintf(e->err, sizeof(e->err), "%s: %s", what, hipGetErrorString(he));
    else snprintf(e->err, sizeof(e->err), "%s", what);
  }
  return code;
}

template <typename T, typename TV, typename TIO, int N>
hipError_t launch(const mpcqp_engine* e, int64_t B, const void* x0, const void* r, const uint8_t* contact,
                  const void* xdes, const void* mu, void* u, void* X, int32_t* status, int32_t* iters, float* res,
                  hipStream_t st) {
  hipLaunchKernelGGL((mpcqp_solve_kernel<T, TV, TIO, N>), dim3((unsigned)B), dim3(Geo<N>::NT), 0, st, e->dcfg, e->ctab,
                     (const TIO*)x0, (const TIO*)r, contact, (const TIO*)xdes, (const TIO*)mu, (TIO*)u, (TIO*)X,
                     status, iters, res);
  return hipGetLastError();
}

template <typename TIO, int N>
hipError_t launch_prec(const mpcqp_engine* e, int64_t B, const void* x0, const void* r, const uint8_t* c,
                       const void* xd, const void* mu, void* u, void* X, int32_t* st, int32_t* it, float* res,
                       hipStream_t s) {
  switch (e->cfg.precision) {
    case MPCQP_PREC_F32: return launch<float, float, TIO, N>(e, B, x0, r, c, xd, mu, u, X, st, it, res, s);
    case MPCQP_PREC_MIXED: return launch<float, double, TIO, N>(e, B, x0, r, c, xd, mu, u, X, st, it, res, s);
    default:
      if constexpr (N == 10) return launch<double, double, TIO, N>(e, B, x0, r, c, xd, mu, u, X, st, it, res, s);
      else return hipErrorInvalidValue;
  }
}

}  // namespace

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
  c->max_iter = 400; c->check_every = 50;
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
  auto reject = [&](int code) { delete e; return code; };
  if (!(N == 10 || N == 20)) return reject(MPCQP_EINVAL);
  if (cfg->precision < MPCQP_PREC_F32 || cfg->precision > MPCQP_PREC_F64) return reject(MPCQP_EINVAL);
  if (cfg->precision == MPCQP_PREC_F64 && N != 10) return reject(MPCQP_EINVAL);
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
  if (hipSetDevice(cfg->device) != hipSuccess) return reject(MPCQP_EHIP);

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
  hipError_t he = hipMalloc((void**)&e->ctab, sizeof(double) * 2 * N * N);
  if (he == hipSuccess) he = hipMemcpy(e->ctab, tab, sizeof(double) * 2 * N * N, hipMemcpyHostToDevice);
  delete[] tab;
  if (he == hipSuccess) he = hipMalloc((void**)&e->dcfg, sizeof(DevCfg));
  if (he == hipSuccess) he = hipMemcpy(e->dcfg, &e->dev, sizeof(DevCfg), hipMemcpyHostToDevice);
  if (he == hipSuccess) he = hipEventCreate(&e->ev0);
  if (he == hipSuccess) he = hipEventCreate(&e->ev1);
  if (he != hipSuccess) {
    if (e->ctab) (void)hipFree(e->ctab);
    if (e->dcfg) (void)hipFree(e->dcfg);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    return reject(MPCQP_EHIP);
  }
  *out = e;
  return MPCQP_OK;
}

int mpcqp_destroy(mpcqp_handle h) {
  if (!h) return MPCQP_OK;
  if (h->ctab) (void)hipFree(h->ctab);
  if (h->dcfg) (void)hipFree(h->dcfg);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return MPCQP_OK;
}

const char* mpcqp_last_error(mpcqp_handle h) { return h ? h->err : "null handle"; }

int mpcqp_solve_batch(mpcqp_handle h, int64_t B, const void* x0, const void* r, const uint8_t* contact, const void* xdes,
                      const void* mu, void* u_out, void* X_out, int32_t* status, int32_t* iters, float* res, void* stream) {
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || B > 0x7fffffff) return fail(h, MPCQP_EINVAL, "mpcqp_solve_batch: batch size out of range");
  if (B > 0 && (!x0 || !r || !contact || !xdes || !mu || !u_out || !status || !iters))
    return fail(h, MPCQP_EINVAL, "mpcqp_solve_batch: null buffer");
  hipStream_t st = (hipStream_t)stream;
  hipError_t he = hipEventRecord(h->ev0, st);
  if (he != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventRecord", he);
  if (B > 0) {
    const bool f64io = h->cfg.dtype == MPCQP_DTYPE_F64;
    if (h->cfg.N == 10)
      he = f64io ? launch_prec<double, 10>(h, B, x0, r, contact, xdes, mu, u_out, X_out, status, iters, res, st)
                 : launch_prec<float, 10>(h, B, x0, r, contact, xdes, mu, u_out, X_out, status, iters, res, st);
    else
      he = f64io ? launch_prec<double, 20>(h, B, x0, r, contact, xdes, mu, u_out, X_out, status, iters, res, st)
                 : launch_prec<float, 20>(h, B, x0, r, contact, xdes, mu, u_out, X_out, status, iters, res, st);
    if (he != hipSuccess) return fail(h, MPCQP_EHIP, "kernel launch", he);
  }
  he = hipEventRecord(h->ev1, st);
  if (he != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventRecord", he);
  h->timed = true;
  return MPCQP_OK;
}

int mpcqp_last_kernel_ms(mpcqp_handle h, float* ms) {
  if (!h || !ms) return MPCQP_EINVAL;
  if (!h->timed) return fail(h, MPCQP_EINVAL, "mpcqp_last_kernel_ms: no solve recorded");
  hipError_t he = hipEventSynchronize(h->ev1);
  if (he == hipSuccess) he = hipEventElapsedTime(ms, h->ev0, h->ev1);
  if (he != hipSuccess) return fail(h, MPCQP_EHIP, "hipEventElapsedTime", he);
  return MPCQP_OK;
}

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
