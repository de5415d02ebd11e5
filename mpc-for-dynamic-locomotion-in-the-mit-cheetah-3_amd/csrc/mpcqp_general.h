// mpcqp_general.h -- the general (single-launch) solve kernel: every precision (F32 / MIXED / F64), N = 10 and 20,
// ADMM-only or ADMM + polish, alpha = 0.  One QP per workgroup, 3 x CW register tiles, 8 lanes per leg-stage.
// The benchmarked configuration (MIXED or F32 with polish, N = 10) takes the fast path in mpcqp_fast.h instead.
#pragma once
#include "mpcqp_device.h"

namespace {

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
  TV pu[G::n], py[G::NL * 5];       // polish iterate (u, y)
  T au[G::n], az[G::NL * 5], ay[G::NL * 5];  // ADMM state, parked here between phases
  T c0[N * N], c1[N * N];
  T pq[G::n * 12];
  T dg[G::n];
  T vbuf[2 * G::VP];
  T rhs[2 * G::VP];
  float red[G::NW * 4];
  uint8_t ct[N * 4];
  uint8_t em[G::NL];                // per-leg enable mask of the current matrix (bit a = variable 3 leg + a)
};

template <typename T, typename TV, typename TIO, int N>
__global__ void __launch_bounds__(Geo<N>::NT, (N == 10 ? MPCQP_WPE : 2))
mpcqp_solve_kernel(const DevCfg* __restrict__ cfgp, const double* __restrict__ ctab, const TIO* __restrict__ x0g,
                   const TIO* __restrict__ rg, const uint8_t* __restrict__ cg, const TIO* __restrict__ xdg,
                   const TIO* __restrict__ mug, TIO* __restrict__ ug, TIO* __restrict__ Xg,
                   int* __restrict__ statusg, int* __restrict__ itersg, float* __restrict__ resg) {
  using G = Geo<N>;
  constexpr int n = G::n, NL = G::NL, CT = G::CT, CW = G::CW, CWP = G::CWP, VP = G::VP, NT = G::NT, NW = G::NW;
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
  for (int i = tid; i < 2 * VP; i += NT) { s.vbuf[i] = (T)0; s.rhs[i] = (T)0; }     // pad slots must stay finite
  for (int i = tid; i < n; i += NT) { s.au[i] = (T)0; s.pu[i] = (TV)0; }
  for (int i = tid; i < NL * 5; i += NT) { s.az[i] = (T)0; s.ay[i] = (T)0; s.py[i] = (TV)0; }
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
  struct_grad<Smem<T, TV, N>, TV, N>(s, tid);  // gradient at u = 0 is the linear term g
  if (tid < n) s.gl[tid] = s.gv[tid];
  __syncthreads();

  // ------------------------------------------------------------------ per-leg constants (registers, replicated on the 8 lanes)
  const bool stance = s.ct[leg] != 0;
  const TV muv = s.mu;
  const int max_iter = cfg.max_iter, check_every = cfg.check_every, polish_max = cfg.polish_max;
  const T mu = (T)muv;
  float gmaxf;
  {
    float q[1] = {fmaxf(fmaxf(fabsf((float)s.gl[row0]), fabsf((float)s.gl[row0 + 1])), fabsf((float)s.gl[row0 + 2]))};
    block_max<1, NW>(q, s.red, tid);
    gmaxf = q[0];
  }
  const bool do_polish = (cfg.flags & MPCQP_FLAG_POLISH) && cfg.alpha > 0.0;
  T rho = (T)cfg.rho;
  int zs = 0, xs = 0, ys = 0;                           // active set of this leg (polish)
  int mode = 0, it = 0, ps = 0, psteps = 0, status = MPCQP_STATUS_MAX_ITER;
  float res_p = 0.f, res_d = 0.f, rho_ratio = 1.f;
  T tile[3][CWP];                                        // columns >= CW are padding and stay zero
  STAMP(0);

  for (;;) {
    // ---------------------------------------------------------------- matrix description -> LDS (lanes 0..2 of the leg: one variable each)
    if (mode == 1) {
      // primal-dual active-set rule on the polish iterate (pu, py), rows: 0 fz | 1 fx - mu fz <= 0 | 2 fx + mu fz >= 0 | 3,4 same for fy
      zs = xs = ys = 0;
      if (stance) {
        const TV fminv = s.cf.fmin, fmaxv = s.cf.fmax;
        const TV u0 = s.pu[row0], u1 = s.pu[row0 + 1], u2 = s.pu[row0 + 2];
        const TV y0 = s.py[leg * 5], y1 = s.py[leg * 5 + 1], y2 = s.py[leg * 5 + 2], y3 = s.py[leg * 5 + 3], y4 = s.py[leg * 5 + 4];
        const TV g1 = u0 - muv * u2, g2 = u0 + muv * u2, g3_ = u1 - muv * u2, g4 = u1 + muv * u2;
        if (y0 + (u2 - fmaxv) > 0) zs = 1;
        else if (y0 + (u2 - fminv) < 0) zs = -1;
        const bool hx = y1 + g1 > 0, lx = y2 + g2 < 0;
        if (hx && lx) xs = (g1 > -g2) ? 1 : -1; else if (hx) xs = 1; else if (lx) xs = -1;
        const bool hy = y3 + g3_ > 0, ly = y4 + g4 < 0;
        if (hy && ly) ys = (g3_ > -g4) ? 1 : -1; else if (hy) ys = 1; else if (ly) ys = -1;
      }
    }
    {
      const bool ez = stance && (mode == 0 || zs == 0), ex = stance && (mode == 0 || xs == 0), ey = stance && (mode == 0 || ys == 0);
      if (cc < 3) {
        const int a = cc;                                 // lane a of the leg describes variable (leg, axis a)
        const bool en = a == 0 ? ex : (a == 1 ? ey : ez);
        TV pv[12];
        var_pq<Smem<T, TV, N>, TV>(s, row0 + a, a, pv);
        const TV a2 = (TV)2 * s.cf.alpha;
        TV dgv;
        if (mode == 0) {
          dgv = stance ? a2 + (TV)cfg.sigma + (TV)rho * (a == 2 ? (TV)1 + (TV)4 * muv * muv : (TV)2) : (TV)1;
        } else {
          if (a == 2 && ez) {                             // tied tangential forces ride on the fz slot
            TV px[12], py_[12];
            var_pq<Smem<T, TV, N>, TV>(s, row0 + 0, 0, px);
            var_pq<Smem<T, TV, N>, TV>(s, row0 + 1, 1, py_);
#pragma unroll
            for (int q = 0; q < 12; ++q) pv[q] += (TV)xs * muv * px[q] + (TV)ys * muv * py_[q];
          }
          dgv = !en ? (TV)1 : (a == 2 ? a2 * ((TV)1 + muv * muv * (TV)((xs != 0) + (ys != 0))) : a2);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) s.pq[(row0 + a) * 12 + q] = en ? (T)pv[q] : (T)0;
        s.dg[row0 + a] = (T)dgv;
      }
      if (cc == 3) s.em[leg] = (uint8_t)((ex ? 1 : 0) | (ey ? 2 : 0) | (ez ? 4 : 0));
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
#pragma unroll
      for (int c = CW; c < CWP; ++c) tile[0][c] = tile[1][c] = tile[2][c] = (T)0;
    }
    STAMP(2);
    // ---------------------------------------------------------------- in-register symmetric sweep: tile <- -M^{-1} on enabled vars
    {
      int step = 0;
      for (int kc = 0; kc < CT; ++kc) {
        int emv[CW / 3];                          // the chunk's enable masks, fetched together (no dependent LDS read per leg-stage)
#pragma unroll
        for (int cb = 0; cb < CW / 3; ++cb) emv[cb] = s.em[kc * (CW / 3) + cb];
#pragma unroll
        for (int cb = 0; cb < CW / 3; ++cb) emv[cb] = __builtin_amdgcn_readfirstlane(emv[cb]);
#pragma unroll
        for (int cb = 0; cb < CW / 3; ++cb) {
          const int og = kc * (CW / 3) + cb;      // owner leg-stage of this block of three pivots
          const int em = emv[cb];                 // uniform: swing / eliminated variables are identity rows, skipped
          if (em == 0) continue;
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) {
            if (!((em >> rr) & 1)) continue;
            const int c = 3 * cb + rr;            // pivot column inside chunk kc (compile-time after unrolling)
            T* vb = s.vbuf + (step & 1) * VP;
            if (leg == og) {
#pragma unroll
              for (int c2 = 0; c2 < CWP; ++c2) vb[cc * CWP + c2] = tile[rr][c2];
            }
            __syncthreads();
            const T p = fast_rcp(vb[kc * CWP + c]);
            T vr[3], vc[CWP];
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) vr[r3] = vb[rbase + r3] * p;
#pragma unroll
            for (int c2 = 0; c2 < CWP; ++c2) vc[c2] = vb[cc * CWP + c2];
            // no divergent region in a pivot (mpcqp_fast.h): the owner's pivot row IS the published row, so the same
            // update with the multiplier 1 - p turns it into p * row; the pivot column is written with selects
            const T vrr = (leg == og) ? (T)1 - p : vr[rr];
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) {
              const T m = r3 == rr ? vrr : vr[r3];
#pragma unroll
              for (int c2 = 0; c2 < CWP; ++c2) tile[r3][c2] -= m * vc[c2];
            }
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) {
              const T v = (r3 == rr && leg == og) ? -p : vr[r3];
              tile[r3][c] = (cc == kc) ? v : tile[r3][c];
            }
            ++step;
          }
        }
      }
    }
    STAMP(3);

    bool finished = false;
    if (mode == 0) {
      // -------------------------------------------------------------- ADMM (OSQP algorithm 1 on the 5 rows per leg-stage)
      const T sigma = (T)cfg.sigma, relax = (T)cfg.relax;
      const T BIG = (T)1e30;
      const T lo0 = stance ? (T)s.cf.fmin : (T)0, hi0 = stance ? (T)s.cf.fmax : (T)0;  // src/mpc.py:151-157
      const T hiP = stance ? BIG : (T)0;                 // rows f + mu fz >= 0 (src/mpc.py:159-173): [0, inf)
      const T loM = stance ? -BIG : (T)0;                // rows f - mu fz <= 0: (-inf, 0]
      // The leg-stage's state is spread over its lanes as in the fast path (mpcqp_fast.h, LegLane): lane q = cc & 3 holds
      // component min(q, 2) and its constraint rows (0: fx and fx -+ mu fz, 1: fy and fy -+ mu fz, 2,3: fz and its box row);
      // lanes 4..7 mirror 0..3.  Full copies (u3, z5, y5) exist only at the checkpoints, reloaded from LDS.
      const int q = cc & 3, comp = q < 2 ? q : 2;
      const int rowA = q == 0 ? 1 : (q == 1 ? 3 : 0), rowB = q == 0 ? 2 : (q == 1 ? 4 : -1);
      const bool tang = q < 2;
      const T mA = tang ? -mu : (T)0, mB = tang ? mu : (T)0, aB = tang ? (T)1 : (T)0, kz = tang ? (T)0 : mu;
      const T loA = !stance ? (T)0 : (tang ? loM : lo0), hiA = !stance ? (T)0 : (tang ? (T)0 : hi0);
      const T loB = (T)0, hiB = (stance && tang) ? hiP : (T)0;
      const T gc = (T)s.gl[row0 + comp];
      T uc_ = s.au[row0 + comp];
      T zA = s.az[leg * 5 + rowA], yA = s.ay[leg * 5 + rowA];
      T zB = rowB >= 0 ? s.az[leg * 5 + rowB] : (T)0, yB = rowB >= 0 ? s.ay[leg * 5 + rowB] : (T)0;
      T wc = 0;
      T g3[3], u3[3], z5[5], y5[5];
      int buf = 0;
      const T inv_rho = (T)1 / rho, om = (T)1 - relax;
      auto update_w = [&]() {   // G'(rho z - y) of my component (+ the mu-coupled part of the tangential lanes on the fz lanes)
        const T vA = rho * zA - yA, vB = rho * zB - yB;
        const T d = vB - vA;
        const T dx = dpp_mov<0x00>(d), dy = dpp_mov<0x55>(d);
        wc = kz * (dx + dy) + vA + vB;
      };
      update_w();
      s.rhs[rbase + comp] = sigma * uc_ - gc + wc;
      __syncthreads();
      bool go_polish = false;
      while (it < max_iter) {
        T acc[3] = {0, 0, 0};
        {
          const T* rb = s.rhs + buf * VP + cc * CWP;
#pragma unroll
          for (int c2 = 0; c2 < CWP; ++c2) {
            const T xv = rb[c2];
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) acc[r3] += tile[r3][c2] * xv;
          }
        }
        T ut[3];
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) ut[r3] = -group8_sum(acc[r3]);
        const T utz = ut[2], utc = q == 0 ? ut[0] : (q == 1 ? ut[1] : ut[2]);
        uc_ = relax * utc + om * uc_;
        {
          const T gA = mA * utz + utc;
          const T zr = relax * gA + om * zA;
          T zn = zr + yA * inv_rho;
          zn = zn < loA ? loA : (zn > hiA ? hiA : zn);
          yA += rho * (zr - zn);
          zA = zn;
        }
        {
          const T gB = mB * utz + aB * utc;
          const T zr = relax * gB + om * zB;
          T zn = zr + yB * inv_rho;
          zn = zn < loB ? loB : (zn > hiB ? hiB : zn);
          yB += rho * (zr - zn);
          zB = zn;
        }
        update_w();
        buf ^= 1;
        s.rhs[buf * VP + rbase + comp] = sigma * uc_ - gc + wc;
        ++it;
        __syncthreads();
        if (it % check_every == 0 || it == max_iter) {
          STAMP(4);
          // park the ADMM state in LDS (frees its registers for the checkpoint / polish) and publish u, y
          if (cc < 3) {
            s.au[row0 + comp] = uc_; s.uv[row0 + comp] = (TV)uc_; s.pu[row0 + comp] = (TV)uc_;
            s.az[leg * 5 + rowA] = zA; s.ay[leg * 5 + rowA] = yA; s.py[leg * 5 + rowA] = (TV)yA;
            if (rowB >= 0) { s.az[leg * 5 + rowB] = zB; s.ay[leg * 5 + rowB] = yB; s.py[leg * 5 + rowB] = (TV)yB; }
          }
          __syncthreads();
#pragma unroll
          for (int c = 0; c < 3; ++c) { g3[c] = (T)s.gl[row0 + c]; u3[c] = s.au[row0 + c]; }
#pragma unroll
          for (int i = 0; i < 5; ++i) { z5[i] = s.az[leg * 5 + i]; y5[i] = s.ay[leg * 5 + i]; }
          // residuals of the QP at (u, z, y): |Gu - z|_inf, |grad f(u) + G'y|_inf  (OSQP termination test)
          struct_grad<Smem<T, TV, N>, TV, N>(s, tid);
          float q[4] = {0.f, 0.f, 0.f, 0.f};
          {
            const TV U0 = (TV)u3[0], U1 = (TV)u3[1], U2 = (TV)u3[2];
            const TV gu[5] = {U2, U0 - muv * U2, U0 + muv * U2, U1 - muv * U2, U1 + muv * U2};
            const TV Gy[3] = {(TV)y5[1] + (TV)y5[2], (TV)y5[3] + (TV)y5[4],
                              (TV)y5[0] + muv * (-(TV)y5[1] + (TV)y5[2] - (TV)y5[3] + (TV)y5[4])};
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              q[0] = fmaxf(q[0], fabsf((float)(gu[i] - (TV)z5[i])));
              q[2] = fmaxf(q[2], fmaxf(fabsf((float)gu[i]), fabsf((float)z5[i])));
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const TV gr = s.gv[row0 + c];
              q[1] = fmaxf(q[1], fabsf((float)(gr + Gy[c])));
              q[3] = fmaxf(q[3], fmaxf(fabsf((float)(gr - s.gl[row0 + c])), fabsf((float)Gy[c])));
            }
          }
          if (!(isfinite(q[0]) && isfinite(q[1]))) q[0] = q[1] = INFINITY;
          block_max<4, NW>(q, s.red, tid);
          res_p = q[0]; res_d = q[1];
          const float sp = q[2], sd = fmaxf(q[3], gmaxf);
          if (!(isfinite(res_p) && isfinite(res_d))) { status = MPCQP_STATUS_NONFINITE; finished = true; break; }
          // with polish enabled the KKT-checked polish is the only acceptance test: OSQP's residual test is too
          // loose in the weakly-curved (alpha-only) directions of this QP to guarantee 1e-4 on the forces
          const float eps_abs = (float)cfg.eps_abs, eps_rel = (float)cfg.eps_rel;
          if (!do_polish && res_p <= eps_abs + eps_rel * sp && res_d <= eps_abs + eps_rel * sd) {
            status = MPCQP_STATUS_SOLVED_ADMM;
            finished = true;
            break;
          }
          STAMP(5);
          rho_ratio = sqrtf((res_p / fmaxf(sp, 1e-12f)) / fmaxf(res_d / fmaxf(sd, 1e-12f), 1e-30f));
          if (do_polish) { go_polish = true; break; }
        }
      }
      if (!finished) {
        if (go_polish) { mode = 1; ps = 0; }
        else finished = true;  // iteration cap without polish: s.uv holds the last iterate
      }
    } else {
      // -------------------------------------------------------------- polish: equality-constrained QP on the active rows,
      // solved with the swept reduced matrix as preconditioner and refined against the structured gradient
      const TV fminv = s.cf.fmin, fmaxv = s.cf.fmax;
      const bool ez = stance && zs == 0, ex = stance && xs == 0, ey = stance && ys == 0;
      TV up3[3] = {0, 0, 0};
      if (stance && zs != 0) {
        const TV F = zs > 0 ? fmaxv : fminv;
        up3[2] = F;
        if (xs) up3[0] = (TV)xs * muv * F;
        if (ys) up3[1] = (TV)ys * muv * F;
      }
      TV v3[3] = {ex ? s.pu[row0] : (TV)0, ey ? s.pu[row0 + 1] : (TV)0, ez ? s.pu[row0 + 2] : (TV)0};
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
      const float acc_stat = (sizeof(TV) == 8) ? (1e-5f + 1e-8f * gmaxf) : (1e-5f * fmaxf(gmaxf, 1.f));
      const float ftol = (sizeof(TV) == 8) ? 1e-7f : 2e-5f;
      // dual-sign slack must stay well below alpha-curvature * force tolerance: a wrongly "active" row with multiplier -e
      // moves the forces by ~e / (2 alpha)
      const float dtol = (sizeof(TV) == 8) ? (1e-5f + 1e-9f * gmaxf) : (2e-5f * fmaxf(1.f, gmaxf));
      float stat = INFINITY, prev_stat = INFINITY;
      float viol[3];
      TV yn[5];
      bool ok = false;
      // stage 0: a couple of refinement rounds, then a loose KKT screen; only a candidate that passes is refined
      // to the tight tolerance (stage 1) and checked again.  Wrong active sets are dropped early.
      for (int stg = 0; stg < 2; ++stg) {
        const float tol = stg == 0 ? fmaxf(tol_stat, 1e-3f * fmaxf(gmaxf, 1.f)) : tol_stat;
        const int max_rf = stg == 0 ? 2 : 10;
        TV gr[3] = {0, 0, 0};
        for (int rf = 0;; ++rf) {
          if (cc == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s.uv[row0 + c] = uc[c];
          }
          __syncthreads();
          struct_grad<Smem<T, TV, N>, TV, N>(s, tid);
#pragma unroll
          for (int c = 0; c < 3; ++c) gr[c] = s.gv[row0 + c];
          const TV rg[3] = {ex ? gr[0] : (TV)0, ey ? gr[1] : (TV)0,
                            ez ? gr[2] + (TV)xs * muv * gr[0] + (TV)ys * muv * gr[1] : (TV)0};
          float q[1] = {fmaxf(fmaxf(fabsf((float)rg[0]), fabsf((float)rg[1])), fabsf((float)rg[2]))};
          if (!isfinite(q[0])) q[0] = INFINITY;
          block_max<1, NW>(q, s.red, tid);
          prev_stat = stat;
          stat = q[0];
          if (stat <= tol || rf >= max_rf || (rf > 0 && !(stat < 0.5f * prev_stat))) break;  // converged / stagnated (uniform)
          if (cc == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s.rhs[rbase + c] = (T)(-rg[c]);
          }
          __syncthreads();
          T acc[3] = {0, 0, 0};
          {
            const T* rb = s.rhs + cc * CWP;
#pragma unroll
            for (int c2 = 0; c2 < CWP; ++c2) {
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
        // duals from stationarity grad_leg + G_A' y_A = 0, then the KKT check (primal feasibility + dual sign)
#pragma unroll
        for (int i = 0; i < 5; ++i) yn[i] = 0;
        viol[0] = viol[1] = viol[2] = 0.f;  // primal violation, dual-sign violation, |u|
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
        if (stg == 0) {
          // loose screen: violations an order of magnitude above what refinement could still remove mean a wrong active set
          const bool plausible = viol[0] <= 1e-2f * fmaxf(1.f, viol[2]) && viol[1] <= 1e-2f * fmaxf(1.f, gmaxf);
          if (!plausible) break;
        } else {
          // a stationarity / dual-sign slack e moves the forces by ~e / (2 alpha): scale the acceptance with the curvature so
        // that `solved` implies the 1e-4 band for any alpha (binding only below alpha ~ 1e-2)
        // (fp64-residual modes only: the all-fp32 mode cannot resolve such slacks and keeps its documented 2e-2 band)
        const float a2f = (sizeof(TV) == 8) ? 2.f * (float)s.cf.alpha : 1e30f, uscale = fmaxf(1.f, viol[2]);
        ok = viol[0] <= ftol * uscale && viol[1] <= fminf(dtol, a2f * 1e-5f * uscale) && stat <= fminf(acc_stat, a2f * 2e-5f * uscale);
        }
      }
      STAMP(6);
      ++psteps;
      // publish the candidate (u, y) as the next polish iterate / the answer
      if (cc == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { s.pu[row0 + c] = uc[c]; s.uv[row0 + c] = uc[c]; }
#pragma unroll
        for (int i = 0; i < 5; ++i) s.py[leg * 5 + i] = yn[i];
      }
      __syncthreads();
      STAMP(7);
      if (ok) {
        status = MPCQP_STATUS_SOLVED_POLISHED;
        res_p = viol[0];
        res_d = fmaxf(viol[1], stat);
        finished = true;
      } else {
        ++ps;
        if (ps >= polish_max) {
          // give the forces back to the last ADMM iterate
          if (cc == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) s.uv[row0 + c] = (TV)s.au[row0 + c];
          }
          __syncthreads();
          if (it >= max_iter) {
            finished = true;  // status stays MAX_ITER; u is the last ADMM iterate
          } else {
            mode = 0;  // back to ADMM; OSQP's rho adaptation comes for free because the matrix is rebuilt anyway
            if (rho_ratio > 2.f || rho_ratio < 0.5f) rho = (T)fminf(fmaxf((float)rho * rho_ratio, 1e-4f), 1e4f);
          }
        }
      }
    }
    if (finished) break;
  }

  // ------------------------------------------------------------------ outputs (src/mpc.py:265-268); s.uv holds the answer
  if (cc == 0 && (!stance || status == MPCQP_STATUS_NONFINITE)) s.uv[row0] = s.uv[row0 + 1] = s.uv[row0 + 2] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += NT) ug[b * n + i] = (TIO)s.uv[i];
  if (Xg) {
    struct_grad<Smem<T, TV, N>, TV, N>(s, tid);
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
