// mpcqp_stage.h -- stage-wise engine: any horizon up to 64 stages, the reference's committed N = 60 included (src/main.py:37,41).
//
// Same problem, same algorithm and same wrench-space form as mpcqp_wrench.h (OSQP ADMM to find the active set, primal-dual
// active-set polish with a KKT acceptance test, regulariser continuation for small alpha; H = 2 alpha I + T'KT and every linear
// system "diagonal + T'KT" reduced by Woodbury to  S y = b,  S = K^-1 + E,  E = T D^-1 T' block diagonal, 6 x 6 per stage) --
// but S (6N x 6N: 360 x 360 at N = 60) is never formed.  K = C'(2W)C, where C maps a wrench sequence v to the deviation states
// z_k = (P_k, Q_k) of six decoupled double integrators (the closed forms of DESIGN.md section 2, as a recursion):
//     z_{k+1} = Phi z_k + Gam v_k,   Phi = [[I, d I], [0, I]],   Gam = [theta d^2 I ; d I],   z_0 = 0
// so S y = b is the two-point boundary problem  v_k + E_k y_k = b_k,  y_k = Gam' lam_{k+1},  lam_k = Phi' lam_{k+1} + 2W z_k,
// solved by a Riccati recursion (lam_k = Pi_k z_k + pi_k, Pi 12 x 12) -- O(N) per factorisation and per solve:
//     Psi = Gam' Pi+ Gam,  Z = (Psi^-1 + E_k)^-1,  Ehat = Psi^-1 Z E_k  (= (I + E Psi)^-1 E, product form: E - E Z E cancels when
//     the force weight is small),  L = Pi+ Gam Ehat,   Pi_k = 2W + Phi' (Pi+ - L (Pi+ Gam)') Phi
//     backward  pi_k    = Phi' (I - L Gam') (Pi+ Gam b_k + pi_{k+1})
//     forward   z_{k+1} = (I - Gam L') (Phi z_k + Gam (b_k - E_k Gam' pi_{k+1})),      y_k = Gam' (Pi_{k+1} z_{k+1} + pi_{k+1})
// (numpy prototype against the dense system: tools/stage_proto.py, 1e-13 relative at alpha = 1e-2, 1e-9 at 1e-4).
//
// Mapping: one QP per workgroup of 256 lanes, ONE LANE PER LEG-STAGE (4 N <= 256): the leg's data and its ADMM / polish iterates
// live in that lane's registers for the whole solve (no register tile exists here), stage sums are two DPP quad steps, and LDS
// holds only the wrench-space vectors (6 N) and the recursion's states (12 N).  The two serial recursions of a solve run on ONE
// group of 8 lanes -- lane c holds (P_c, Q_c) of wrench component c, the 6-vectors they need from each other are gathered with
// DPP butterflies (lane ^ 1, ^ 2, ^ 3 by quad_perm, ^ 7 by row_half_mirror) -- from per-stage factor matrices stored in that
// butterfly order and read from LDS one stage ahead (fp32: resident for the whole ADMM block; fp64: 64 KB per direction, copied
// from a global-memory workspace in front of each chain), so that nothing on the recursion's critical path waits for memory;
// everything that does not depend on the recursion's carry (Pi+ Gam b_k, E_k Gam' pi_{k+1}, y_k) is done by all lanes in parallel
// before / between / after the two chains.
// The factorisation itself always runs in fp64 (wave 0, 6 x 6 inverses in LDS); the ADMM chains use its fp32 rounding in the
// MIXED precision and fp64 in F64; the polish is all fp64.
#pragma once
#include "mpcqp_wrench.h"

namespace {

constexpr int SG_NS = 64;                 // stages a workgroup can hold
constexpr int SG_NT = 256, SG_NW = 4;     // one lane per leg-stage
constexpr int SG_NQ = 6 * SG_NS;
constexpr int SG_TL_MIN = 24;             // horizons from which the two recursions of a solve run two-level (eight chunks in parallel)
#ifndef MPCQP_SG_WARM_FRAC10
#define MPCQP_SG_WARM_FRAC10 4
#endif
constexpr int SG_WARM_FRAC10 = MPCQP_SG_WARM_FRAC10;   // first block of a solve warm-started from a neighbour, in tenths of a cold solve's
                                                       // (one robot on consecutive logged ticks: 3 / 4 / 6 / 8 tenths -> 1134 / 1122 / 1039 / 1017 solves/s, cold 855;
                                                       //  profiles/r03f_stage_warm.txt)
constexpr double SG_ALPHA_FLOOR = 2e-5;   // where this engine's continuation of an alpha = 0 request ends (mpcqp_kernels.hip)
// Workspace of one resident workgroup, in doubles: the fp64 chains' factor matrices, per stage  Lrow (128) | Lcol (128)  (the fp32
// chains keep theirs in LDS for the whole ADMM block).
constexpr int SG_WS_STAGE = 128 + 128;
constexpr size_t SG_WS_DOUBLES = (size_t)SG_NS * SG_WS_STAGE;

struct SmemS {
  double x0[13];
  double mu, cy, sy, rzw0[3], wP[6], wQ[6], delta, theta, alpha, inv_m, fmin, fmax, alpha_target, alpha_ok;
  double wQ01;   // weight coupling the rotated angular-rate components 0 and 1: Rz diag(w_wx, w_wy) Rz' (zero when the omega weight is isotropic in x, y)
  double gam[SG_NQ];                   // gradient of the cost in wrench space at u = 0
  double ww[SG_NQ], kap[SG_NQ];        // wrench of the structured gradient's point, K ww + gam
  double bq[SG_NQ];                    // S y = b: the right-hand side (wrench space)
  // The recursion's vectors, 12 per stage in chain order: component c at [12 k + 2 c] (P) and [12 k + 2 c + 1] (Q), so that chain lane c
  // moves its pair with one LDS access (+ 4: lanes 6, 7 of the chain group read past the last stage and mask the value).
  alignas(16) double s0[SG_NS * 12 + 4];        // Pi+ Gam b_k (backward half), then d_k (forward half, P slots); setup: e0 (own layout)
  alignas(16) double pist[(SG_NS + 1) * 12];    // pi_k, k = 1..N (index k); setup: x_des staging (with zst)
  alignas(16) double zst[(SG_NS + 1) * 12];     // z_k, k = 0..N; structured gradient: the deviation states of its point (own layout)
  double PGs[SG_NS * 72];              // Pi+ Gam per stage (12 x 6, rows P_0..P_5, Q_0..Q_5), written by the factorisation
  double Es[SG_NS * 36];               // E_k = T D^-1 T' per stage
  // factorisation (wave 0): operand areas of its four products
  double Pi[144], PG[72], Ps[36];
  // two-level chains (horizons >= SG_TL_MIN): transition matrices of the inner chunks, T_g = A_lo .. A_(hi-1) with A_k = Phi' (I - L_k Gam'),
  // 12 x 12 row-major over (P_0..P_5, Q_0..Q_5), in the chains' element type (fp32: T and T' per chunk, fp64: T only), and the chunks'
  // local results
  alignas(16) double Tc[6 * 144];
  alignas(16) double cmb[8 * 12];
  float red[SG_NW * 4];
  float aared[SG_NW * 12];             // Anderson acceleration: the waves' partial inner products
  float kkt[4], resid[4];
  float gmax, rho, ratio;
  int iters, psteps, hard, bad;
  unsigned ahash, ahist[32];
  uint8_t aset[SG_NT];
  // The chains' factor matrices, staged in LDS (a chain step that waits for an L2 round trip costs ~0.6 us, measured): fp32 -- Lrow
  // and Lcol of all stages, written here by the factorisation and resident for the whole ADMM block; fp64 -- the matrices of ONE
  // direction, copied from the workspace by all lanes in front of each chain.
  alignas(16) unsigned char fbuf[SG_NS * 128 * 8];
};

// One lane's leg-stage: model data and iterates (registers, for the whole solve).
struct SLeg {
  double B[9];                   // Rz Ihat^-1 [r]x masked by contact: B[3 i + a] (row i = angular component, column a = force axis)
  double cm;                     // contact / m
  double g[3];                   // linear term g = T' gam
  double ua[3], za[5], ya[5];    // last ADMM iterate (multipliers unscaled)
  double pu[3], py[5];           // polish iterate / last candidate
  double uv[3];                  // point of the structured gradient; the accepted answer
  bool stance, leg;
};

// g[s] = x of lane (lane ^ s) inside its group of 8.
template <typename T>
__device__ __forceinline__ void xor_gather8(const T x, T (&g)[8]) {
  g[0] = x;
  g[1] = dpp_mov<0xB1>(x);            // quad_perm [1,0,3,2]
  g[2] = dpp_mov<0x4E>(x);            // quad_perm [2,3,0,1]
  g[3] = dpp_mov<0x1B>(x);            // quad_perm [3,2,1,0]
  const T x7 = dpp_mov<0x141>(x);     // row_half_mirror: lane i <-> 7 - i = i ^ 7
  g[7] = x7;
  g[6] = dpp_mov<0xB1>(x7);
  g[5] = dpp_mov<0x4E>(x7);
  g[4] = dpp_mov<0x1B>(x7);
}

// v of lane (byte address / 4) of the wave, for a double (two crossbar moves; no LDS memory is touched).
__device__ __forceinline__ double sg_bperm(const double v, const int byte_addr) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int LANE>
__device__ __forceinline__ double sg_readlane(const double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, LANE), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), LANE);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Inverse of a symmetric positive definite 6 x 6 matrix held one entry per lane on the wave's 8 x 8 grid (lane = 8 i + j; i, j < 6
// carry the matrix): six symmetric sweeps, M <- -M^-1, the pivot row fetched through the crossbar (by symmetry it is the pivot
// column too) and the pivot itself through a scalar register -- no LDS round trip in a sweep.  Lanes off the 6 x 6 grid carry zeros
// that nothing reads.
__device__ __forceinline__ double sg_inv6_reg(double m, const int i, const int j) {
#define SG_SWEEP(P)                                                                   \
  {                                                                                   \
    const double rpj = sg_bperm(m, 4 * (8 * P + j)), rpi = sg_bperm(m, 4 * (8 * P + i)); \
    const double r = w_rcp(sg_readlane<9 * P>(m));                                    \
    const double upd = fma(-rpi * r, rpj, m);                                         \
    m = (i == P) ? ((j == P) ? -r : rpj * r) : ((j == P) ? rpi * r : upd);            \
  }
  SG_SWEEP(0) SG_SWEEP(1) SG_SWEEP(2) SG_SWEEP(3) SG_SWEEP(4) SG_SWEEP(5)
#undef SG_SWEEP
  return -m;
}

// Riccati factorisation for the E_k blocks in s.Es (fp64), by wave 0; the other waves wait at the closing barrier.
// Writes per stage: Lrow / Lcol (the rows / columns of L = Pi+ Gam Ehat in the butterfly order of the two chains; fp32 -> LDS,
// fp64 -> workspace) and PG = Pi+ Gam (LDS).
// The N stages are a dependent sequence and each is a handful of 6 x 6 operations, so what a stage costs is its count of
// DEPENDENT memory round trips, not its flops.  Pi (12 x 12) therefore lives in registers as its four 6 x 6 blocks, lane (i, j) of
// the wave's 8 x 8 grid holding entry (i, j) of each: Pi Gam, Psi = Gam' Pi Gam and Phi' M Phi are then lane-local, transposes and
// the two 6 x 6 inverses go through the crossbar (sg_inv6_reg), and only the four products (Z E, Psi^-1 (Z E), PG Ehat, L PG') read
// operand rows from LDS -- four round trips per stage (the first form of this routine kept every matrix in LDS: ~40 round trips,
// 10 k cycles per stage, a third of a solve at N = 60; tools/stage_stamps.py).
template <typename TM>
__device__ __forceinline__ void sg_factor(SmemS& s, double* __restrict__ ws, const int N, const int tid, const bool chunks = false) {
  if (tid < 64) {
    const int i = tid >> 3, j = tid & 7;
    const bool on = i < 6 && j < 6;
    const int ii = min(i, 5), jj = min(j, 5);                 // (addresses of the lanes off the grid stay inside the arrays)
    const int tp = 4 * (8 * j + i);                           // the transposed entry's lane
    const double d = s.delta, th = s.theta, gp = th * d * d, gq = d;   // Gam = [gp I ; gq I]
    const double w2P = (on && i == j) ? 2.0 * s.wP[ii] : 0.0;
    const double w2Q = !on ? 0.0 : (i == j ? 2.0 * s.wQ[ii] : (i + j == 1 ? 2.0 * s.wQ01 : 0.0));   // (the rotated omega weight couples 0 and 1)
    double PP = w2P, PQ = 0.0, QP = 0.0, QQ = w2Q;            // Pi_N = 2W
    // LDS operand areas of the four products (the former scratch matrices of this routine)
    double* const A6 = s.Pi;          // 36: left operand of a 6 x 6 product
    double* const B6 = s.Pi + 36;     // 36: right operand
    double* const GPm = s.Pi + 72;    // 36: PG rows P
    double* const GQm = s.Pi + 108;   // 36: PG rows Q
    double* const LPm = s.PG;         // 36: L rows P
    double* const LQm = s.PG + 36;    // 36: L rows Q
    double* const EHm = s.Ps;         // 36: Ehat
    double En = on ? s.Es[36 * (N - 1) + 6 * ii + jj] : 0.0;
#pragma unroll 1
    for (int k = N - 1; k >= 0; --k) {
      const double E = En;
      En = on ? s.Es[36 * max(k - 1, 0) + 6 * ii + jj] : 0.0;   // next stage's block, early
      const double gP = on ? gp * PP + gq * PQ : 0.0, gQ = on ? gp * QP + gq * QQ : 0.0;   // PG = Pi Gam: rows P_i / Q_i, column j
      double psi = gp * gP + gq * gQ;                                                       // Psi = Gam' PG
      psi = 0.5 * (psi + sg_bperm(psi, tp));
      const double pinv = sg_inv6_reg(on ? psi : 0.0, i, j);
      const double zm = sg_inv6_reg(on ? pinv + E : 0.0, i, j);                             // Z = (Psi^-1 + E)^-1
      if (on) { A6[6 * ii + jj] = zm; B6[6 * ii + jj] = E; GPm[6 * ii + jj] = gP; GQm[6 * ii + jj] = gQ; }
      wsync<1>();
      double t1 = 0.0;                                                                      // Z E
#pragma unroll
      for (int q = 0; q < 6; ++q) t1 = fma(A6[6 * ii + q], B6[6 * q + jj], t1);
      wsync<1>();
      if (on) { A6[6 * ii + jj] = pinv; B6[6 * ii + jj] = t1; }
      wsync<1>();
      double eh = 0.0;                                                                      // Psi^-1 Z E  (= (I + E Psi)^-1 E)
#pragma unroll
      for (int q = 0; q < 6; ++q) eh = fma(A6[6 * ii + q], B6[6 * q + jj], eh);
      eh = on ? 0.5 * (eh + sg_bperm(eh, tp)) : 0.0;
      if (on) EHm[6 * ii + jj] = eh;
      wsync<1>();
      double lP = 0.0, lQ = 0.0;                                                            // L = PG Ehat
#pragma unroll
      for (int q = 0; q < 6; ++q) { const double e = EHm[6 * q + jj]; lP = fma(GPm[6 * ii + q], e, lP); lQ = fma(GQm[6 * ii + q], e, lQ); }
      if (!on) { lP = 0.0; lQ = 0.0; }
      if (on) { LPm[6 * ii + jj] = lP; LQm[6 * ii + jj] = lQ; }
      // factors out (TM): butterfly layouts of the chains (group lane c, slot sl: partner c ^ sl; Lrow: c = row, Lcol: c = column), PG row-major
      if constexpr (sizeof(TM) == 4) {
        float* fr = reinterpret_cast<float*>(s.fbuf) + 128 * k, *fc = fr + SG_NS * 128;
        fr[16 * i + (i ^ j)] = (float)lP; fr[16 * i + 8 + (i ^ j)] = (float)lQ;
        fc[16 * j + (i ^ j)] = (float)lP; fc[16 * j + 8 + (i ^ j)] = (float)lQ;
      } else {
        double* fac = ws + (size_t)k * SG_WS_STAGE;
        fac[16 * i + (i ^ j)] = lP; fac[16 * i + 8 + (i ^ j)] = lQ;
        fac[128 + 16 * j + (i ^ j)] = lP; fac[128 + 16 * j + 8 + (i ^ j)] = lQ;
      }
      if (on) { s.PGs[72 * k + 6 * ii + jj] = gP; s.PGs[72 * k + 36 + 6 * ii + jj] = gQ; }
      wsync<1>();
      // M = Pi - L PG', symmetrised;  Pi <- 2W + Phi' M Phi = [[Mpp, d Mpp + Mpq], [d Mpp + Mqp, d^2 Mpp + d (Mpq + Mqp) + Mqq]]
      double mPP = 0.5 * (PP + sg_bperm(PP, tp)), mQQ = 0.5 * (QQ + sg_bperm(QQ, tp));
      double mPQ = 0.5 * (PQ + sg_bperm(QP, tp)), mQP = 0.5 * (QP + sg_bperm(PQ, tp));
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double lPi = LPm[6 * ii + q], lQi = LQm[6 * ii + q], lPj = LPm[6 * jj + q], lQj = LQm[6 * jj + q];
        const double gPi = GPm[6 * ii + q], gQi = GQm[6 * ii + q], gPj = GPm[6 * jj + q], gQj = GQm[6 * jj + q];
        mPP -= 0.5 * (lPi * gPj + lPj * gPi);
        mPQ -= 0.5 * (lPi * gQj + lQj * gPi);
        mQP -= 0.5 * (lQi * gPj + lPj * gQi);
        mQQ -= 0.5 * (lQi * gQj + lQj * gQi);
      }
      wsync<1>();
      PP = on ? mPP + w2P : 0.0;
      PQ = on ? fma(d, mPP, mPQ) : 0.0;
      QP = on ? fma(d, mPP, mQP) : 0.0;
      QQ = on ? fma(d * d, mPP, fma(d, mPQ + mQP, mQQ)) + w2Q : 0.0;
    }
  }
  __syncthreads();
  if (chunks && N >= SG_TL_MIN) {   // (uniform) transition matrices of the inner chunks: lane = (chunk, column), the column pushed through its stages
    const int CH = (N + 7) >> 3, G = (N + CH - 1) / CH;
    if (tid < 72) {
      const int m = tid / 12, col = tid - 12 * m, g = m + 1;
      if (g <= G - 2) {
        // in the chains' own arithmetic and from the factors in the rounding the chains use (Lrow layout: row c, slot c ^ column)
        const TM d = (TM)s.delta, gp = (TM)(s.theta * s.delta * s.delta), gq = d;
        TM v[12];
#pragma unroll
        for (int r = 0; r < 12; ++r) v[r] = r == col ? (TM)1 : (TM)0;
#pragma unroll 1
        for (int k = min(N, (g + 1) * CH) - 1; k >= g * CH; --k) {
          TM w[6];
#pragma unroll
          for (int q = 0; q < 6; ++q) w[q] = gp * v[q] + gq * v[6 + q];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            TM lP[8], lQ[8];
            if constexpr (sizeof(TM) == 4) {
              const float* fr = reinterpret_cast<const float*>(s.fbuf) + 128 * k + 16 * i;
              ld8<float>(fr, lP); ld8<float>(fr + 8, lQ);
            } else {
              const double* fr = ws + (size_t)k * SG_WS_STAGE + 16 * i;
              ld8<double>(fr, lP); ld8<double>(fr + 8, lQ);
            }
            TM aP = v[i], aQ = v[6 + i];
#pragma unroll
            for (int q = 0; q < 6; ++q) { aP = fma(-lP[i ^ q], w[q], aP); aQ = fma(-lQ[i ^ q], w[q], aQ); }
            v[i] = aP; v[6 + i] = fma(d, aP, aQ);
          }
        }
        // fp32: T and T' (each chain reads contiguous rows); fp64: T only (room for one)
        TM* T = reinterpret_cast<TM*>(s.Tc) + (sizeof(TM) == 4 ? 288 : 144) * m;
#pragma unroll
        for (int r = 0; r < 12; ++r) T[12 * r + col] = v[r];
        if constexpr (sizeof(TM) == 4) {
#pragma unroll
          for (int r = 0; r < 12; ++r) T[144 + 12 * col + r] = v[r];
        }
      }
    }
    __syncthreads();
  }
}

// E_k = sum_legs A diag(dinv) A' (fp64) -> LDS.  Leg lanes (a quad = a stage).  Ends with a barrier.
__device__ __forceinline__ void sg_build_E(SmemS& s, const LegSys<double>& L, const bool leg, const int tid) {
  double e[21];
  int k = 0;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
#pragma unroll
    for (int p = q; p < 6; ++p) {
      double a = L.dinv[0] * L.A[0][q] * L.A[0][p];
      a = fma(L.dinv[1] * L.A[1][q], L.A[1][p], a);
      a = fma(L.dinv[2] * L.A[2][q], L.A[2][p], a);
      e[k++] = quad_sum(a);
    }
  }
  if (leg && (tid & 3) == 0) {
    double* Ej = s.Es + 36 * (tid >> 2);
    k = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int p = q; p < 6; ++p) { Ej[6 * q + p] = e[k]; Ej[6 * p + q] = e[k]; ++k; }
    }
  }
  __syncthreads();
}

// One combine step of the two-level chains: out = R' x with lane c of each group holding x[P_c], x[Q_c] and rows P_c, Q_c of R
// (12 entries each over P_0..P_5, Q_0..Q_5): twelve partial outputs per lane, reduce-scattered over the group -- lane c gets out[P_c],
// out[Q_c].  R = T for the forward recursion (out = T' z) and T' for the backward one (out = T pi).  Lanes 6, 7 of a group hold zeros.
template <typename TM>
__device__ __forceinline__ void sg_combine(const TM (&R)[24], const TM xp, const TM xq, const int c, TM& op, TM& oq) {
  TM vP[8], vQ[8];
#pragma unroll
  for (int r = 0; r < 6; ++r) { vP[r] = fma(R[12 + r], xq, R[r] * xp); vQ[r] = fma(R[18 + r], xq, R[6 + r] * xp); }
  vP[6] = vP[7] = vQ[6] = vQ[7] = (TM)0;
  op = rs8<TM>(vP, c);
  oq = rs8<TM>(vQ, c);
}
// rows P_c, Q_c of a 12 x 12 row-major matrix (contiguous)
__device__ __forceinline__ void sg_ld_rows(const float* __restrict__ M, const int c, float (&R)[24]) {
  const float4* a = reinterpret_cast<const float4*>(M + 12 * c);
  const float4* b = reinterpret_cast<const float4*>(M + 72 + 12 * c);
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    const float4 u = a[h], w = b[h];
    R[4 * h] = u.x; R[4 * h + 1] = u.y; R[4 * h + 2] = u.z; R[4 * h + 3] = u.w;
    R[12 + 4 * h] = w.x; R[12 + 4 * h + 1] = w.y; R[12 + 4 * h + 2] = w.z; R[12 + 4 * h + 3] = w.w;
  }
}
__device__ __forceinline__ void sg_ld_rows(const double* __restrict__ M, const int c, double (&R)[24]) {
  const double2* a = reinterpret_cast<const double2*>(M + 12 * c);
  const double2* b = reinterpret_cast<const double2*>(M + 72 + 12 * c);
#pragma unroll
  for (int h = 0; h < 6; ++h) { const double2 u = a[h], w = b[h]; R[2 * h] = u.x; R[2 * h + 1] = u.y; R[12 + 2 * h] = w.x; R[12 + 2 * h + 1] = w.y; }
}
// columns P_c, Q_c (the rows of the transpose; fp64 keeps one copy of each matrix)
__device__ __forceinline__ void sg_ld_cols(const double* __restrict__ M, const int c, double (&R)[24]) {
#pragma unroll
  for (int r = 0; r < 12; ++r) { R[r] = M[12 * r + c]; R[12 + r] = M[12 * r + 6 + c]; }
}

// x = M^-1 rhs,  M = D + A-stack' K A-stack:  x = dinv (rhs - A' y),  S y = A-stack dinv rhs  by the two recursions.  All lanes call.
//   leg lanes   a = dinv rhs,  b_k = sum_legs A a (quad sum) -> bq;  the quad's four lanes share the 12 entries of Pi+ Gam b_k -> s0
//   chain       backward recursion (wave 0): pi_k -> pist
//   all lanes   d_k = b_k - E_k Gam' pi_{k+1} -> s0 (P slots)
//   chain       forward recursion: z_{k+1} -> zst
//   leg lanes   y_k = PG_k' z_{k+1} + Gam' pi_{k+1} (two components per lane, shared over the quad by DPP),  x = a - dinv A' y_k
// Four barriers.  The chain lanes read their pair of the NEXT step and its factor rows one step ahead; nothing in a step waits for LDS.
template <typename TM>
__device__ __forceinline__ void sg_leg_solve(SmemS& s, const double* __restrict__ ws, const LegSys<double>& L, const double (&rhs)[3], double (&x)[3],
                                             const bool leg, const int N, const int tid, const bool chunks = false) {
  using T2 = std::conditional_t<sizeof(TM) == 4, float2, double2>;
  const bool two_level = chunks && N >= SG_TL_MIN;   // (uniform; the factorisation was asked for the chunks' transition matrices)
  const int CH = (N + 7) >> 3, G = (N + CH - 1) / CH;   // two-level: chunk g = stages [g CH, min(N, (g + 1) CH)), one per lane group of wave 0
  const double d = s.delta, th = s.theta, gp = th * d * d, gq = d;
  const int kq = min(tid, 4 * N - 1) >> 2, lq = tid & 3;
  // The recursion's vectors in the chains' element type (fp32 chains: the first half of the same bytes)
  TM* const s0 = reinterpret_cast<TM*>(s.s0);
  TM* const pist = reinterpret_cast<TM*>(s.pist);
  TM* const zst = reinterpret_cast<TM*>(s.zst);
  double a[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) a[c] = L.dinv[c] * rhs[c];
  {
    double b[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) b[q] = quad_sum(fma(L.A[2][q], a[2], fma(L.A[1][q], a[1], L.A[0][q] * a[0])));
    if (leg) {
      if (lq == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) s.bq[6 * kq + q] = b[q];
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {   // rows 3 lq .. 3 lq + 2 of PG_k b_k  (row i = 6 h + c  ->  chain slot 2 c + h)
        const int i = 3 * lq + t, h = i >= 6 ? 1 : 0, c = i - 6 * h;
        const double* pg = s.PGs + 72 * kq + 6 * i;
        double v = pg[0] * b[0];
#pragma unroll
        for (int q = 1; q < 6; ++q) v = fma(pg[q], b[q], v);
        s0[12 * kq + 2 * c + h] = (TM)v;
      }
    }
    if (tid < 12) { pist[12 * N + tid] = (TM)0; zst[tid] = (TM)0; }
    if constexpr (sizeof(TM) == 8) {   // fp64 chains: Lrow of all stages -> LDS
      double* fb = reinterpret_cast<double*>(s.fbuf);
      for (int e = tid; e < 128 * N; e += SG_NT) fb[e] = ws[(size_t)(e >> 7) * SG_WS_STAGE + (e & 127)];
    }
  }
  __syncthreads();
  const TM* const frow = reinterpret_cast<const TM*>(s.fbuf);
  const TM* const fcol = reinterpret_cast<const TM*>(s.fbuf) + (sizeof(TM) == 4 ? SG_NS * 128 : 0);
  if (tid < 64 && two_level) {
    // Backward recursion, two-level: (1) every lane group runs its chunk from a zero carry, (2) the chunk results are combined serially
    // through the transition matrices (pi_lo = rho + T pi_hi: G - 2 dense 12 x 12 steps, every group computing the same sequence and
    // keeping the carry that enters ITS chunk), (3) every group runs its chunk again from that carry and stores.  2 CH + G - 2 dependent
    // steps instead of N.
    const int c = tid & 7, grp = tid >> 3;
    const bool on6 = c < 6;
    const TM tgp = (TM)gp, tgq = (TM)gq, td = (TM)d;
    const int lo = grp * CH, hi = min(N, lo + CH), kmin = max(lo, 1);
    TM pp = 0, pq = 0;
    auto run = [&](const bool store) {
      TM LA[16], LB[16];
      int k = hi - 1;
      {
        const TM* f = frow + 128 * max(k, 0) + 16 * c;
#pragma unroll
        for (int t = 0; t < 16; ++t) LA[t] = f[t];
      }
      T2 sA = *reinterpret_cast<const T2*>(s0 + 12 * max(k, 0) + 2 * c), sB;
      auto step = [&](const TM (&Lr)[16], const T2& sn, TM (&Ln)[16], T2& sn2) {
        {
          const TM* f = frow + 128 * max(k - 1, 0) + 16 * c;
#pragma unroll
          for (int t = 0; t < 16; ++t) Ln[t] = f[t];
        }
        sn2 = *reinterpret_cast<const T2*>(s0 + 12 * max(k - 1, 0) + 2 * c);
        const bool valid = k >= kmin;   // (k < hi by construction; an empty chunk has hi <= lo)
        TM sp = (on6 ? sn.x : (TM)0) + pp, sq = (on6 ? sn.y : (TM)0) + pq;
        const TM cc = tgp * sp + tgq * sq;
        TM g[8];
        xor_gather8(cc, g);
        TM a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) { a0 = fma(Lr[t], g[t], a0); a1 = fma(Lr[4 + t], g[4 + t], a1); b0 = fma(Lr[8 + t], g[t], b0); b1 = fma(Lr[12 + t], g[4 + t], b1); }
        sp -= a0 + a1; sq -= b0 + b1;
        pp = valid ? sp : pp; pq = valid ? fma(td, sp, sq) : pq;
        if (store && valid && on6) { T2 w; w.x = pp; w.y = pq; *reinterpret_cast<T2*>(pist + 12 * k + 2 * c) = w; }
        --k;
      };
      int t = 0;
#pragma unroll 1
      for (; t + 1 < CH; t += 2) { step(LA, sA, LB, sB); step(LB, sB, LA, sA); }
      if (t < CH) step(LA, sA, LB, sB);
    };
    run(false);
    TM* const cmb = reinterpret_cast<TM*>(s.cmb);
    const TM* const Tc = reinterpret_cast<const TM*>(s.Tc);
    if (on6) { T2 w; w.x = pp; w.y = pq; *reinterpret_cast<T2*>(cmb + 12 * grp + 2 * c) = w; }
    wsync<1>();
    // pi entering chunk g (from above): x_(G-1) = pi_N = 0,  x_(g-1) = rho_g + T_g x_g
    TM xp = 0, xq = 0, myp = 0, myq = 0;
    {   // outer-product form on rows of T' (fp32: stored; fp64: the columns of T), operands of the next step fetched during this one
      const int cr = min(c, 5);
      TM RA[24], RB[24];
      T2 rA, rB;
      auto ldT = [&](const int g, TM (&R)[24]) {   // T' of chunk g
        if constexpr (sizeof(TM) == 4) sg_ld_rows(Tc + 288 * (g - 1) + 144, cr, R);
        else sg_ld_cols(Tc + 144 * (g - 1), cr, R);
      };
      auto cstep = [&](const int g, const TM (&R)[24], const T2& rho, TM (&Rn)[24], T2& rhon) {
        if (g >= 2) ldT(g - 1, Rn);
        rhon = *reinterpret_cast<const T2*>(cmb + 12 * max(g - 1, 0) + 2 * cr);
        myp = grp == g ? xp : myp; myq = grp == g ? xq : myq;
        TM ap, aq;
        sg_combine<TM>(R, xp, xq, c, ap, aq);
        xp = on6 ? rho.x + ap : (TM)0; xq = on6 ? rho.y + aq : (TM)0;
      };
      ldT(G - 2, RB);   // for the second step (G >= 7 at these horizons)
      rA = *reinterpret_cast<const T2*>(cmb + 12 * (G - 1) + 2 * cr);
      rB = *reinterpret_cast<const T2*>(cmb + 12 * (G - 2) + 2 * cr);
      // g = G - 1: the carry from above is zero, no product
      xp = on6 ? rA.x : (TM)0; xq = on6 ? rA.y : (TM)0;
      int g = G - 2;
#pragma unroll 1
      for (; g >= 2; g -= 2) { cstep(g, RB, rB, RA, rA); cstep(g - 1, RA, rA, RB, rB); }
      if (g == 1) cstep(1, RB, rB, RA, rA);
    }
    pp = grp == 0 ? xp : myp; pq = grp == 0 ? xq : myq;   // (group G - 1 kept its zero)
    run(true);
  } else if (tid < 64) {   // backward chain: group 0 of wave 0 (the other groups of the wave run along on the same data)
    const int c = tid & 7;
    const bool on6 = c < 6;
    const TM tgp = (TM)gp, tgq = (TM)gq, td = (TM)d;
    TM pp = 0, pq = 0;
    TM LA[16], LB[16];
    {
      const TM* f = frow + 128 * (N - 1) + 16 * c;
#pragma unroll
      for (int t = 0; t < 16; ++t) LA[t] = f[t];
    }
    T2 sA = *reinterpret_cast<const T2*>(s0 + 12 * (N - 1) + 2 * c), sB;
    // one step: stage k with (Lr, sn), while the rows and the pair of stage k - 1 are fetched into (Ln, sn2) -- two register sets
    // that swap roles from step to step (no copies)
    auto step = [&](const int k, const TM (&Lr)[16], const T2& sn, TM (&Ln)[16], T2& sn2) {
      {
        const TM* f = frow + 128 * (k - 1) + 16 * c;   // (stage 0's are loaded and not used)
#pragma unroll
        for (int t = 0; t < 16; ++t) Ln[t] = f[t];
      }
      sn2 = *reinterpret_cast<const T2*>(s0 + 12 * (k - 1) + 2 * c);
      // (lanes 6, 7 of the group hold no component: what they read past the stage's twelve entries is SELECTED away, never multiplied
      //  by zero -- the bytes behind the last stage are whatever the previous kernel left in LDS, and 0 x NaN would reach lanes 0..5
      //  through the butterfly: 8 of the 1000 logged ticks came back NaN from one launch before this was a select)
      TM sp = (on6 ? sn.x : (TM)0) + pp, sq = (on6 ? sn.y : (TM)0) + pq;
      const TM cc = tgp * sp + tgq * sq;
      TM g[8];
      xor_gather8(cc, g);
      TM a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) { a0 = fma(Lr[t], g[t], a0); a1 = fma(Lr[4 + t], g[4 + t], a1); b0 = fma(Lr[8 + t], g[t], b0); b1 = fma(Lr[12 + t], g[4 + t], b1); }
      sp -= a0 + a1; sq -= b0 + b1;
      pp = sp; pq = fma(td, sp, sq);
      if (tid < 6) { T2 w; w.x = pp; w.y = pq; *reinterpret_cast<T2*>(pist + 12 * k + 2 * c) = w; }
    };
    int k = N - 1;
#pragma unroll 1
    for (; k >= 2; k -= 2) { step(k, LA, sA, LB, sB); step(k - 1, LB, sB, LA, sA); }
    if (k == 1) step(1, LA, sA, LB, sB);
  }
  __syncthreads();
  // d_k = b_k - E_k Gam' pi_{k+1} -> P slots of s0  (fp64 chains: Lcol of all stages -> LDS; the backward chain is done with Lrow)
  if constexpr (sizeof(TM) == 8) {
    double* fb = reinterpret_cast<double*>(s.fbuf);
    for (int e = tid; e < 128 * N; e += SG_NT) fb[e] = ws[(size_t)(e >> 7) * SG_WS_STAGE + 128 + (e & 127)];
  }
  for (int e = tid; e < 6 * N; e += SG_NT) {
    const int k = e / 6, c = e - 6 * k;
    const double* Ek = s.Es + 36 * k + 6 * c;
    const TM* pk = pist + 12 * (k + 1);
    double acc = s.bq[e];
#pragma unroll
    for (int q = 0; q < 6; ++q) acc -= Ek[q] * (gp * (double)pk[2 * q] + gq * (double)pk[2 * q + 1]);
    s0[12 * k + 2 * c] = (TM)acc;
  }
  __syncthreads();
  if (tid < 64 && two_level) {   // forward recursion, two-level (as above; the chunks' transition matrices are the transposes T')
    const int c = tid & 7, grp = tid >> 3;
    const bool on6 = c < 6;
    const TM tgp = (TM)gp, tgq = (TM)gq, td = (TM)d;
    const int lo = grp * CH, hi = min(N, lo + CH);
    TM zp = 0, zq = 0;
    auto run = [&](const bool store) {
      TM LA[16], LB[16];
      int k = lo;
      {
        const TM* f = fcol + 128 * min(k, N - 1) + 16 * c;
#pragma unroll
        for (int t = 0; t < 16; ++t) LA[t] = f[t];
      }
      TM dA = s0[12 * min(k, N - 1) + 2 * c], dB;
      auto step = [&](const TM (&Lc)[16], const TM dn, TM (&Ln)[16], TM& dn2) {
        {
          const TM* f = fcol + 128 * min(k + 1, N - 1) + 16 * c;
#pragma unroll
          for (int t = 0; t < 16; ++t) Ln[t] = f[t];
        }
        dn2 = s0[12 * min(k + 1, N - 1) + 2 * c];
        const bool valid = k < hi;
        const TM dk = on6 ? dn : (TM)0;
        const TM tp = zp + td * zq + tgp * dk, tq = zq + tgq * dk;
        TM g0[8], g1[8];
        xor_gather8(tp, g0);
        xor_gather8(tq, g1);
        TM a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) { a0 = fma(Lc[t], g0[t], a0); a1 = fma(Lc[4 + t], g0[4 + t], a1); b0 = fma(Lc[8 + t], g1[t], b0); b1 = fma(Lc[12 + t], g1[4 + t], b1); }
        const TM o = (a0 + a1) + (b0 + b1);
        zp = valid ? tp - tgp * o : zp; zq = valid ? tq - tgq * o : zq;
        if (store && valid && on6) { T2 w; w.x = zp; w.y = zq; *reinterpret_cast<T2*>(zst + 12 * (k + 1) + 2 * c) = w; }
        ++k;
      };
      int t = 0;
#pragma unroll 1
      for (; t + 1 < CH; t += 2) { step(LA, dA, LB, dB); step(LB, dB, LA, dA); }
      if (t < CH) step(LA, dA, LB, dB);
    };
    run(false);
    TM* const cmb = reinterpret_cast<TM*>(s.cmb);
    const TM* const Tc = reinterpret_cast<const TM*>(s.Tc);
    if (on6) { T2 w; w.x = zp; w.y = zq; *reinterpret_cast<T2*>(cmb + 12 * grp + 2 * c) = w; }
    wsync<1>();
    // z entering chunk g (from below): x_0 = z_0 = 0,  x_(g+1) = zeta_g + T_g' x_g
    TM xp = 0, xq = 0, myp = 0, myq = 0;
    {
      const int cr = min(c, 5);
      TM RA[24], RB[24];
      T2 rA, rB;
      constexpr int TS = sizeof(TM) == 4 ? 288 : 144;
      auto cstep = [&](const int g, const TM (&R)[24], const T2& zeta, TM (&Rn)[24], T2& zetan) {
        if (g + 1 <= G - 2) sg_ld_rows(Tc + TS * g, cr, Rn);   // T of chunk g + 1, for the next step
        zetan = *reinterpret_cast<const T2*>(cmb + 12 * min(g + 1, G - 1) + 2 * cr);
        myp = grp == g ? xp : myp; myq = grp == g ? xq : myq;
        TM ap, aq;
        sg_combine<TM>(R, xp, xq, c, ap, aq);
        xp = on6 ? zeta.x + ap : (TM)0; xq = on6 ? zeta.y + aq : (TM)0;
      };
      sg_ld_rows(Tc, cr, RB);   // chunk 1, for the second step
      rA = *reinterpret_cast<const T2*>(cmb + 2 * cr);
      rB = *reinterpret_cast<const T2*>(cmb + 12 + 2 * cr);
      // g = 0: z_0 = 0, no product (group 0 keeps its zero)
      xp = on6 ? rA.x : (TM)0; xq = on6 ? rA.y : (TM)0;
      int g = 1;
#pragma unroll 1
      for (; g + 1 <= G - 2; g += 2) { cstep(g, RB, rB, RA, rA); cstep(g + 1, RA, rA, RB, rB); }
      if (g <= G - 2) cstep(g, RB, rB, RA, rA);
    }
    zp = grp == G - 1 ? xp : myp; zq = grp == G - 1 ? xq : myq;
    run(true);
  } else if (tid < 64) {   // forward chain
    const int c = tid & 7;
    const bool on6 = c < 6;
    const TM tgp = (TM)gp, tgq = (TM)gq, td = (TM)d;
    TM zp = 0, zq = 0;
    TM LA[16], LB[16];
    {
      const TM* f = fcol + 16 * c;
#pragma unroll
      for (int t = 0; t < 16; ++t) LA[t] = f[t];
    }
    TM dA = s0[2 * c], dB;
    auto step = [&](const int k, const TM (&Lc)[16], const TM dn, TM (&Ln)[16], TM& dn2) {
      {
        const TM* f = fcol + 128 * min(k + 1, N - 1) + 16 * c;
#pragma unroll
        for (int t = 0; t < 16; ++t) Ln[t] = f[t];
      }
      dn2 = s0[12 * min(k + 1, N - 1) + 2 * c];
      const TM dk = on6 ? dn : (TM)0;
      const TM tp = zp + td * zq + tgp * dk, tq = zq + tgq * dk;
      TM g0[8], g1[8];
      xor_gather8(tp, g0);
      xor_gather8(tq, g1);
      TM a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) { a0 = fma(Lc[t], g0[t], a0); a1 = fma(Lc[4 + t], g0[4 + t], a1); b0 = fma(Lc[8 + t], g1[t], b0); b1 = fma(Lc[12 + t], g1[4 + t], b1); }
      const TM o = (a0 + a1) + (b0 + b1);
      zp = tp - tgp * o; zq = tq - tgq * o;
      if (tid < 6) { T2 w; w.x = zp; w.y = zq; *reinterpret_cast<T2*>(zst + 12 * (k + 1) + 2 * c) = w; }
    };
    int k = 0;
#pragma unroll 1
    for (; k + 1 < N; k += 2) { step(k, LA, dA, LB, dB); step(k + 1, LB, dB, LA, dA); }
    if (k < N) step(k, LA, dA, LB, dB);
  }
  __syncthreads();
  {   // y_k = PG_k' z_{k+1} + Gam' pi_{k+1}: lanes 0..2 of the quad take two components each, DPP hands them round
    const TM* zk = zst + 12 * (kq + 1);
    const TM* pk = pist + 12 * (kq + 1);
    const int c0 = 2 * min(lq, 2);
    double y0 = gp * (double)pk[2 * c0] + gq * (double)pk[2 * c0 + 1], y1 = gp * (double)pk[2 * c0 + 2] + gq * (double)pk[2 * c0 + 3];
#pragma unroll
    for (int i = 0; i < 12; ++i) {   // PG row i = 6 h + c multiplies z slot 2 c + h
      const int h = i >= 6 ? 1 : 0, c = i - 6 * h;
      const double zv = (double)zk[2 * c + h];
      y0 = fma(s.PGs[72 * kq + 6 * i + c0], zv, y0);
      y1 = fma(s.PGs[72 * kq + 6 * i + c0 + 1], zv, y1);
    }
    const double yj[6] = {dpp_mov<0x00>(y0), dpp_mov<0x00>(y1), dpp_mov<0x55>(y0), dpp_mov<0x55>(y1), dpp_mov<0xAA>(y0), dpp_mov<0xAA>(y1)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double sum = L.A[c][0] * yj[0];
#pragma unroll
      for (int q = 1; q < 6; ++q) sum = fma(L.A[c][q], yj[q], sum);
      x[c] = a[c] - L.dinv[c] * sum;
    }
  }
}

// Adjoint recursion of the double integrators: out[6 j + q] = base[6 j + q] + Gam' lam_{j+1},  lam_k = Phi' lam_{k+1} + (muP_k, muQ_k)
// with muP_k = 2 wP src[12 k + q], muQ_k = 2 wQ src[12 k + 6 + q], k = 1..N.  Six lanes (one per component).  No barrier inside.
__device__ __forceinline__ void sg_adjoint(const SmemS& s, const double* __restrict__ src, const double* __restrict__ base, double* __restrict__ out,
                                           const int N, const int tid) {
  if (tid < 6) {
    const int q = tid;
    const double d = s.delta, gp = s.theta * d * d, w2p = 2.0 * s.wP[q], w2q = 2.0 * s.wQ[q];
    const double w2c = q < 2 ? 2.0 * s.wQ01 : 0.0;   // components 0 and 1 of the rotated angular rate share a 2 x 2 weight
    const int qc = q < 2 ? 1 - q : q;
    double lP = 0, lQ = 0;
#pragma unroll 4
    for (int j = N - 1; j >= 0; --j) {
      const double mP = w2p * src[12 * (j + 1) + q], mQ = fma(w2c, src[12 * (j + 1) + 6 + qc], w2q * src[12 * (j + 1) + 6 + q]);
      const double nP = lP + mP, nQ = fma(d, lP, lQ) + mQ;
      lP = nP; lQ = nQ;
      out[6 * j + q] = (base ? base[6 * j + q] : 0.0) + gp * lP + d * lQ;
    }
  }
}

// Structured gradient at the lane's force f: returns 2 alpha f + T'(K T f + gam) for the lane's leg; leaves the stage wrenches in
// s.ww and the deviation states of f in s.zst.  All lanes call.
__device__ __forceinline__ void sg_grad(SmemS& s, const SLeg& Lg, const double (&f)[3], double (&gr)[3], const int N, const int tid) {
  {
    double w6[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) w6[i] = quad_sum(fma(Lg.B[3 * i + 2], f[2], fma(Lg.B[3 * i + 1], f[1], Lg.B[3 * i] * f[0])));
#pragma unroll
    for (int a = 0; a < 3; ++a) w6[3 + a] = quad_sum(Lg.cm * f[a]);
    if (Lg.leg && (tid & 3) == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) s.ww[6 * (tid >> 2) + q] = w6[q];
    }
  }
  __syncthreads();
  if (tid < 6) {   // deviation states of the wrench sequence, k = 1..N
    const int q = tid;
    const double d = s.delta, gp = s.theta * d * d;
    double P = 0, Q = 0;
#pragma unroll 4
    for (int k = 0; k < N; ++k) {
      const double w = s.ww[6 * k + q];
      const double Pn = P + d * Q + gp * w, Qn = Q + d * w;
      P = Pn; Q = Qn;
      s.zst[12 * (k + 1) + q] = P; s.zst[12 * (k + 1) + 6 + q] = Q;
    }
  }
  if (tid < 64) wsync<1>();
  sg_adjoint(s, s.zst, s.gam, s.kap, N, tid);
  __syncthreads();
  {
    const double* kj = s.kap + 6 * (min(tid, 4 * N - 1) >> 2);
    const double a2 = 2.0 * s.alpha;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double g = fma(a2, f[a], Lg.cm * kj[3 + a]);
#pragma unroll
      for (int i = 0; i < 3; ++i) g = fma(Lg.B[3 * i + a], kj[i], g);
      gr[a] = g;
    }
  }
  __syncthreads();
}

// The leg's 6 x 3 wrench map and inverse diagonal: ADMM (D = 2 alpha + sigma + rho G'G) / polish (reduced variables, D = 2 alpha Z'Z).
__device__ __forceinline__ void sg_admm_sys(const SmemS& s, const DevCfg& cfg, const SLeg& Lg, const double rho, LegSys<double>& Ls) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) Ls.A[c][i] = Lg.B[3 * i + c];
#pragma unroll
    for (int a = 0; a < 3; ++a) Ls.A[c][3 + a] = a == c ? Lg.cm : 0.0;
  }
  const double a2 = 2.0 * s.alpha, m = s.mu;
  Ls.dinv[0] = Ls.dinv[1] = Lg.stance ? 1.0 / (a2 + cfg.sigma + 2.0 * rho) : 0.0;
  Ls.dinv[2] = Lg.stance ? 1.0 / (a2 + cfg.sigma + rho * (1.0 + 4.0 * m * m)) : 0.0;
}

__device__ __forceinline__ void sg_polish_sys(const SmemS& s, const SLeg& Lg, const ActSet& a, LegSys<double>& Ls) {
  const double muv = s.mu, txs = (double)a.xs * muv, tys = (double)a.ys * muv, cm = Lg.cm;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Ls.A[0][i] = a.ex ? Lg.B[3 * i] : 0.0;
    Ls.A[1][i] = a.ey ? Lg.B[3 * i + 1] : 0.0;
    Ls.A[2][i] = a.ez ? Lg.B[3 * i + 2] + txs * Lg.B[3 * i] + tys * Lg.B[3 * i + 1] : 0.0;
  }
  Ls.A[0][3] = a.ex ? cm : 0.0; Ls.A[0][4] = 0; Ls.A[0][5] = 0;
  Ls.A[1][3] = 0; Ls.A[1][4] = a.ey ? cm : 0.0; Ls.A[1][5] = 0;
  Ls.A[2][3] = a.ez ? txs * cm : 0.0; Ls.A[2][4] = a.ez ? tys * cm : 0.0; Ls.A[2][5] = a.ez ? cm : 0.0;
  const double a2 = 2.0 * s.alpha;
  Ls.dinv[0] = a.ex ? 1.0 / a2 : 0.0;
  Ls.dinv[1] = a.ey ? 1.0 / a2 : 0.0;
  Ls.dinv[2] = a.ez ? 1.0 / (a2 * (1.0 + muv * muv * (double)((a.xs != 0) + (a.ys != 0)))) : 0.0;
}

// The active-set rule of the polish on the lane's (pu, py) -> ActSet code (as w_polish_rule).
__device__ __forceinline__ int sg_polish_rule(const SmemS& s, const SLeg& Lg) {
  const double muv = s.mu, fminv = s.fmin, fmaxv = s.fmax;
  int zs = 0, xs = 0, ys = 0;
  if (Lg.stance) {
    const double u0 = Lg.pu[0], u1 = Lg.pu[1], u2 = Lg.pu[2];
    const double g1 = u0 - muv * u2, g2 = u0 + muv * u2, g3_ = u1 - muv * u2, g4 = u1 + muv * u2;
    if (Lg.py[0] + (u2 - fmaxv) > 0) zs = 1;
    else if (Lg.py[0] + (u2 - fminv) < 0) zs = -1;
    const bool hx = Lg.py[1] + g1 > 0, lx = Lg.py[2] + g2 < 0;
    if (hx && lx) xs = (g1 > -g2) ? 1 : -1; else if (hx) xs = 1; else if (lx) xs = -1;
    const bool hy = Lg.py[3] + g3_ > 0, ly = Lg.py[4] + g4 < 0;
    if (hy && ly) ys = (g3_ > -g4) ? 1 : -1; else if (hy) ys = 1; else if (ly) ys = -1;
  }
  return (zs + 1) | ((xs + 1) << 2) | ((ys + 1) << 4);
}

// OSQP's residuals and rho-adaptation ratio of the ADMM iterate (u, z, y) held by the leg lanes.  Uniform result; s.resid set.
__device__ __forceinline__ float sg_ratio(SmemS& s, const SLeg& Lg, const double (&u)[3], const double (&z)[5], const double (&y)[5], const int N,
                                          const int tid) {
  double hv[3];
  sg_grad(s, Lg, u, hv, N, tid);
  float q[4] = {0.f, 0.f, 0.f, 0.f};
  if (Lg.leg) {
    const double mu = s.mu, m = mu * u[2];
    const double gu[5] = {u[2], u[0] - m, u[0] + m, u[1] - m, u[1] + m};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      q[0] = fmaxf(q[0], fabsf((float)(gu[i] - z[i])));
      q[2] = fmaxf(q[2], fmaxf(fabsf((float)gu[i]), fabsf((float)z[i])));
    }
    const double Gy[3] = {y[1] + y[2], y[3] + y[4], y[0] + mu * (-y[1] + y[2] - y[3] + y[4])};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      q[1] = fmaxf(q[1], fabsf((float)(hv[a] + Gy[a])));
      q[3] = fmaxf(q[3], fmaxf(fabsf((float)(hv[a] - Lg.g[a])), fabsf((float)Gy[a])));
    }
  }
  block_max<4, SG_NW>(q, s.red, tid);
  const float sp = q[2], sd = fmaxf(q[3], s.gmax);
  if (tid == 0) { s.resid[0] = q[0]; s.resid[1] = q[1]; s.resid[2] = sp; s.resid[3] = sd; }
  __syncthreads();
  return sqrtf((q[0] / fmaxf(sp, 1e-12f)) / fmaxf(q[1] / fmaxf(sd, 1e-12f), 1e-30f));
}

// Warm start (MPCQP_FLAG_WARM_START [+ WARM_SHIFT]; the reference seeds every solve with its previous solution, src/mpc.py:270-271):
// same contract as w_warm_start of mpcqp_wrench.h.  The lane's leg-stage reads its guess u0 (stage k + shift, last stage repeated) and the
// engine's record y0 of the previous solve's multipliers; returns 0 no guess | 1 primal guess only | 2 (u0, y0) nearly a KKT point | 3
// only a neighbour, with the lane's (pu, py) / (ua, za, ya) set to the start of the active-set iteration / of the ADMM block.
template <typename TIO>
__device__ __forceinline__ int sg_warm_start(SmemS& s, SLeg& Lg, const TIO* __restrict__ u0, const float* __restrict__ y0, const int shift,
                                             const int N, const int tid) {
  const int k = min(tid, 4 * N - 1) >> 2, l = tid & 3, ks = min(k + shift, N - 1);
  double f[3] = {0, 0, 0}, y[5] = {0, 0, 0, 0, 0};
  float amax[2] = {0.f, 0.f};
  if (Lg.leg && Lg.stance) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { const double v = (double)u0[ks * 12 + 3 * l + a]; f[a] = isfinite(v) ? v : 0.0; amax[0] = fmaxf(amax[0], fabsf((float)f[a])); }
    if (y0) {
#pragma unroll
      for (int i = 0; i < 5; ++i) { const float v = y0[(ks * 4 + l) * 5 + i]; y[i] = isfinite(v) ? (double)v : 0.0; amax[1] = fmaxf(amax[1], fabsf((float)y[i])); }
    }
  }
  block_max<2, SG_NW>(amax, s.red, tid);
  if (!(amax[0] > 0.f)) return 0;                        // uniform: no guess
  const bool duals = amax[1] > 0.f;
  double g3[3];
  sg_grad(s, Lg, f, g3, N, tid);                          // H u0 + g
  float rs[1] = {0.f};
  const double mu = s.mu, flo = s.fmin, fhi = s.fmax;
  const double g[5] = {f[2], f[0] - mu * f[2], f[0] + mu * f[2], f[1] - mu * f[2], f[1] + mu * f[2]};
  double z[5] = {0, 0, 0, 0, 0};
  if (Lg.leg && Lg.stance) {
    const double tb = 1e-3 * fmax(fabs(f[2]), 1.0), tf = 1e-3 * fmax(mu * fabs(f[2]), 1.0);
    if (!duals) {   // rows that u0 holds with equality get a unit multiplier of the right sign: the first polish step works on u0's own active set
      y[0] = f[2] >= fhi - tb ? 1.0 : (f[2] <= flo + tb ? -1.0 : 0.0);
      y[1] = g[1] >= -tf ? 1.0 : 0.0;  y[2] = g[2] <= tf ? -1.0 : 0.0;
      y[3] = g[3] >= -tf ? 1.0 : 0.0;  y[4] = g[4] <= tf ? -1.0 : 0.0;
    }
    z[0] = f[2] < flo ? flo : (f[2] > fhi ? fhi : f[2]);
    z[1] = g[1] > 0 ? 0.0 : g[1];  z[2] = g[2] < 0 ? 0.0 : g[2];
    z[3] = g[3] > 0 ? 0.0 : g[3];  z[4] = g[4] < 0 ? 0.0 : g[4];
    if (duals) {
      const double rx = g3[0] + (y[1] + y[2]), ry = g3[1] + (y[3] + y[4]), rz = g3[2] + y[0] + mu * (-y[1] + y[2] - y[3] + y[4]);
      rs[0] = (float)fmax(fmax(fabs(rx), fabs(ry)), fabs(rz));
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { Lg.pu[a] = f[a]; Lg.ua[a] = f[a]; }
#pragma unroll
  for (int i = 0; i < 5; ++i) { Lg.py[i] = y[i]; Lg.za[i] = z[i]; Lg.ya[i] = duals ? y[i] : 0.0; }
  block_max<1, SG_NW>(rs, s.red, tid);
  return !duals ? 1 : (rs[0] <= WARM_KKT_TOL * fmaxf(s.gmax, 1.f) ? 2 : 3);
}

// One ADMM block (OSQP algorithm 1, scaled duals) from the lane's (ua, za, ya) with penalty s.rho; `adapt`: the single early rho
// check of a cold solve's first block.  Updates s.rho / s.iters / s.hard and the lane's iterate (ua, za, ya) and polish start (pu, py).
template <typename TM>
__device__ __forceinline__ void sg_admm(SmemS& s, const DevCfg& cfg, SLeg& Lg, double* __restrict__ ws, const int adapt, const int kfirst,
                                        const int N, const int tid, const bool aa_on) {
  float rho = s.rho;
  int K = kfirst > 0 ? min(kfirst, cfg.check_every) : cfg.check_every;
  K = max(1, min(K, cfg.max_iter - s.iters));
  const int seg_len = (aa_on && cfg.accel_p > 0 && cfg.accel_restart > 0) ? cfg.accel_restart : (1 << 30);   // (segments: see w_admm)
  const bool check = adapt && cfg.early_check;
  // (a cold solve's first block is split at ADAPT_AT with or without the check: the history's one fresh start there is worth 2-3 %)
  int it = 0, seg_end = min(K, (adapt && MPCQP_W_ADAPT_AT < K) ? MPCQP_W_ADAPT_AT : seg_len);
  int hard = 0;
  float ratio = 0.f;
  STAMP_INIT
  __syncthreads();   // (everyone has read s.rho / s.iters)
  for (;;) {
    LegSys<double> Ls;
    sg_admm_sys(s, cfg, Lg, (double)rho, Ls);
    sg_build_E(s, Ls, Lg.leg, tid);
    sg_factor<TM>(s, ws, N, tid, true);
    STAMP(1);
    const double sigma = cfg.sigma, relax = cfg.relax, om = 1.0 - relax, BIG = 1e30, r = (double)rho, mu = s.mu;
    double u[3], z[5], yh[5];
#pragma unroll
    for (int a = 0; a < 3; ++a) u[a] = Lg.ua[a];
#pragma unroll
    for (int i = 0; i < 5; ++i) { z[i] = Lg.za[i]; yh[i] = Lg.ya[i] / r; }
    const double lo0 = Lg.stance ? s.fmin : 0.0, hi0 = Lg.stance ? s.fmax : 0.0, loA = Lg.stance ? -BIG : 0.0, hiB = Lg.stance ? BIG : 0.0;
    bool rebuild = false;
    // Anderson acceleration of the block (mpcqp_wrench.h: w_aa_step; with the polish only): an iteration costs two recursions of N
    // steps here, an extrapolation nine workgroup-wide sums -- a few per cent of the block for the iterations it saves.  In a cold
    // solve's FIRST block only: at N = 60 the later rounds of the few QPs that need them ended in polish rounds of up to 98 steps with
    // it (a refactorisation each; profiles/r03_stage_accel.txt), while the first block is where the logged ticks of the reference's
    // run are all solved.  The period must divide the early rho check's iteration (25): a check that comes one or two iterations
    // after an extrapolation sees a distorted residual ratio and flags most QPs (periods 4 and 6: 40 % slower)
    const int aa_p = aa_on ? cfg.accel_p : 0;
    LegAA aa;
    double aa_xb[5], aa_fp[5];
    for (;;) {
      bool aa_have = false;
      int aa_left = aa_p;
      if (aa_p > 0) {
        w_aa_reset(aa);
#pragma unroll
        for (int k = 0; k < 5; ++k) { aa_xb[k] = z[k] + yh[k]; aa_fp[k] = aa_xb[k]; }
      }
      for (; it < seg_end; ++it) {
        double v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = z[k] - yh[k];
        const double w0 = v[1] + v[2], w1 = v[3] + v[4], w2 = fma(mu, (v[2] - v[1]) + (v[4] - v[3]), v[0]);
        const double rhs[3] = {fma(r, w0, fma(sigma, u[0], -Lg.g[0])), fma(r, w1, fma(sigma, u[1], -Lg.g[1])), fma(r, w2, fma(sigma, u[2], -Lg.g[2]))};
        double ut[3];
        sg_leg_solve<TM>(s, ws, Ls, rhs, ut, Lg.leg, N, tid, true);
        if constexpr (sizeof(TM) == 8) {
          if (cfg.refine_admm) {   // tight-tolerance ADMM-only runs: one refinement step on M u~ = rhs (mpcqp_wrench.h)
            double hv[3], rr[3], du[3];
            sg_grad(s, Lg, ut, hv, N, tid);   // H u~ + g
            const double dg[3] = {2.0, 2.0, 1.0 + 4.0 * mu * mu};
#pragma unroll
            for (int a = 0; a < 3; ++a) rr[a] = rhs[a] - ((hv[a] - Lg.g[a]) + (sigma + r * dg[a]) * ut[a]);
            sg_leg_solve<TM>(s, ws, Ls, rr, du, Lg.leg, N, tid, true);
#pragma unroll
            for (int a = 0; a < 3; ++a) ut[a] += du[a];
          }
        }
        const double mz = mu * ut[2];
        const double gt[5] = {ut[2], ut[0] - mz, ut[0] + mz, ut[1] - mz, ut[1] + mz};
#pragma unroll
        for (int a = 0; a < 3; ++a) u[a] = fma(relax, ut[a], om * u[a]);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const double lo = k == 0 ? lo0 : ((k & 1) ? loA : 0.0), hi = k == 0 ? hi0 : ((k & 1) ? 0.0 : hiB);
          const double t = fma(relax, gt[k], om * z[k]) + yh[k];
          const double zn = fmin(fmax(t, lo), hi);
          yh[k] = t - zn;
          z[k] = zn;
        }
        if (aa_p > 0 && --aa_left == 0) {   // uniform
          aa_left = aa_p;
          if (it + 1 + aa_p <= seg_end) {   // (the segment ends with at least a period of genuine ADMM iterations, see w_admm)
            double fx[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) fx[k] = z[k] + yh[k];
            w_aa_step<double, SG_NW>(aa, aa_xb, aa_fp, fx, aa_have, Lg.leg, s.aared, tid);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              const double lo = k == 0 ? lo0 : ((k & 1) ? loA : 0.0), hi = k == 0 ? hi0 : ((k & 1) ? 0.0 : hiB);
              const double zn = fmin(fmax(aa_xb[k], lo), hi);
              yh[k] = aa_xb[k] - zn;
              z[k] = zn;
            }
          }
        }
      }
      STAMP(2);
      if (it >= K) break;
      if (check && !hard && it == MPCQP_W_ADAPT_AT) {
        double y5[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) y5[k] = r * yh[k];
        ratio = sg_ratio(s, Lg, u, z, y5, N, tid);
        STAMP(3);
        if (ratio > cfg.adapt_thr) { rebuild = true; break; }   // uniform
      }
      seg_end = min(K, it + seg_len);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { Lg.ua[a] = u[a]; Lg.pu[a] = u[a]; }
#pragma unroll
    for (int k = 0; k < 5; ++k) { const double y = r * yh[k]; Lg.za[k] = z[k]; Lg.ya[k] = y; Lg.py[k] = y; }
    if (!rebuild) break;
    rho = fminf(rho * ratio, ADAPT_RHO_MAX);
    hard = 1;
    K = max(K, min((cfg.hard_x10 * K) / 10, cfg.max_iter - s.iters));
    seg_end = min(K, it + seg_len);
  }
  __syncthreads();
  if (tid == 0) { s.rho = rho; s.iters += K; s.hard |= hard; }
  __syncthreads();
}

// The polish steps of one round (as w_polish_round; every step refactors -- there is no dense inverse to update).  Returns 1 when a
// step was accepted (answer in the lanes' uv), else 0 with (pu, py) the last candidate.
__device__ __forceinline__ int sg_polish_round(SmemS& s, const DevCfg& cfg, SLeg& Lg, double* __restrict__ ws, const int budget, const bool last,
                                               const int N, const int tid) {
  int ok = 0, ps = 0, nstall = 0;
  float vprev = INFINITY, vprev2 = INFINITY;
  STAMP_INIT
  for (;;) {
    const int code = sg_polish_rule(s, Lg);
    if (Lg.leg) s.aset[tid] = (uint8_t)code;
    if (tid == 0) s.ahash = 0u;
    __syncthreads();
    if (Lg.leg) atomicAdd(&s.ahash, ((unsigned)code + 1u) * (2654435761u * (unsigned)(2 * tid + 1)));
    __syncthreads();
    {   // an active set this round has already tried: the iteration would repeat itself
      const unsigned h = s.ahash;
      bool seen = false;
      for (int k = 0; k < min(ps, 32); ++k) seen = seen || s.ahist[k] == h;
      __syncthreads();
      if (tid == 0 && ps < 32) s.ahist[ps] = h;
      if (seen && !last) break;   // uniform (a round that nothing follows goes on: mpcqp_wrench.h)
    }
    const ActSet as(code, Lg.stance);
    LegSys<double> Ls;
    sg_polish_sys(s, Lg, as, Ls);
    STAMP(7);
    sg_build_E(s, Ls, Lg.leg, tid);
    sg_factor<double>(s, ws, N, tid);
    STAMP(4);
    const int zs = as.zs, xs = as.xs, ys = as.ys;
    const bool ez = as.ez, ex = as.ex, ey = as.ey;
    const double muv = s.mu, txs = (double)xs * muv, tys = (double)ys * muv;
    double v3[3];
    {
      const double F = zs > 0 ? s.fmax : s.fmin;
      v3[0] = ex ? Lg.pu[0] : 0.0; v3[1] = ey ? Lg.pu[1] : 0.0;
      v3[2] = ez ? Lg.pu[2] : ((Lg.stance && zs != 0) ? F : 0.0);
    }
    double uc[3], gr3[3] = {0, 0, 0};
    float stat = INFINITY, prev = INFINITY;
    for (int rf = 0;; ++rf) {
      uc[2] = v3[2]; uc[0] = ex ? v3[0] : txs * v3[2]; uc[1] = ey ? v3[1] : tys * v3[2];
      sg_grad(s, Lg, uc, gr3, N, tid);
      STAMP(5);
      const double rg[3] = {ex ? gr3[0] : 0.0, ey ? gr3[1] : 0.0, ez ? gr3[2] + txs * gr3[0] + tys * gr3[1] : 0.0};
      float q[2] = {Lg.leg ? fmaxf(fmaxf(fabsf((float)rg[0]), fabsf((float)rg[1])), fabsf((float)rg[2])) : 0.f,
                    Lg.leg ? fmaxf(fmaxf(fabsf((float)uc[0]), fabsf((float)uc[1])), fabsf((float)uc[2])) : 0.f};
      if (!isfinite(q[0])) q[0] = INFINITY;
      block_max<2, SG_NW>(q, s.red, tid);
      prev = stat; stat = q[0];
      const float gmaxl = s.gmax;
      const float tol_stat = 1e-6f + 1e-9f * gmaxl;
      const float tol = fminf(tol_stat, 0.25f * 2.f * (float)s.alpha * 2e-5f * fmaxf(1.f, q[1]));
      // (the recursion's solve is as exact as cond(S) eps -- 1e-13 at alpha = 1e-2, 1e-6 at 3e-6, tools/stage_proto.py -- and the Woodbury
      //  form amplifies that by the cancellation in dinv (rhs - A'y): small regularisers get more refinement steps)
      const int rf_max = s.alpha < 1e-3 ? 12 : 4;
      if (stat <= tol || rf >= rf_max || (rf > 0 && !(stat < 0.5f * prev))) break;   // uniform
      const double rhs[3] = {-rg[0], -rg[1], -rg[2]};
      double dx[3];
      sg_leg_solve<double>(s, ws, Ls, rhs, dx, Lg.leg, N, tid);
      v3[0] += ex ? dx[0] : 0.0; v3[1] += ey ? dx[1] : 0.0; v3[2] += ez ? dx[2] : 0.0;
      STAMP(6);
    }
    // duals from stationarity, primal feasibility + dual sign (thresholds: the fp64-buffer set of mpcqp_wrench.h)
    const double fminv = s.fmin, fmaxv = s.fmax;
    const float gmaxf = s.gmax;
    const float acc_stat = 1e-5f + 1e-8f * gmaxf, ftol = 1e-7f, dtol = 1e-5f + 1e-9f * gmaxf;
    double yn[5] = {0, 0, 0, 0, 0};
    float viol[3] = {0.f, 0.f, 0.f};
    if (Lg.leg && Lg.stance) {
      double zacc = gr3[2];
      if (xs > 0) { yn[1] = -gr3[0]; zacc += muv * (-yn[1]); }
      else if (xs < 0) { yn[2] = -gr3[0]; zacc += muv * yn[2]; }
      if (ys > 0) { yn[3] = -gr3[1]; zacc += muv * (-yn[3]); }
      else if (ys < 0) { yn[4] = -gr3[1]; zacc += muv * yn[4]; }
      if (zs != 0) yn[0] = -zacc;
      const double g0 = uc[2], g1 = uc[0] - muv * uc[2], g2 = uc[0] + muv * uc[2], g3_ = uc[1] - muv * uc[2], g4 = uc[1] + muv * uc[2];
      double pv = fmax(fminv - g0, g0 - fmaxv);
      pv = fmax(pv, fmax(g1, -g2));
      pv = fmax(pv, fmax(g3_, -g4));
      double dv = fmax(fmax(-yn[1], yn[2]), fmax(-yn[3], yn[4]));
      if (zs > 0) dv = fmax(dv, -yn[0]);
      if (zs < 0) dv = fmax(dv, yn[0]);
      viol[0] = (float)fmax(pv, 0.0);
      viol[1] = (float)fmax(dv, 0.0);
      viol[2] = fmaxf(fmaxf(fabsf((float)uc[0]), fabsf((float)uc[1])), fabsf((float)uc[2]));
      if (!(isfinite(viol[0]) && isfinite(viol[1]))) viol[0] = viol[1] = INFINITY;
    }
    block_max<3, SG_NW>(viol, s.red, tid);
    const float a2f = 2.f * (float)s.alpha, uscale = fmaxf(1.f, viol[2]);
    const bool step_ok = viol[0] <= ftol * uscale && viol[1] <= fminf(dtol, a2f * 1e-5f * uscale) && stat <= fminf(acc_stat, a2f * 2e-5f * uscale);
#pragma unroll
    for (int c = 0; c < 3; ++c) { Lg.pu[c] = uc[c]; Lg.uv[c] = uc[c]; }
#pragma unroll
    for (int i = 0; i < 5; ++i) Lg.py[i] = yn[i];
    if (tid == 0) { s.kkt[0] = stat; s.kkt[1] = viol[0]; s.kkt[2] = viol[1]; s.psteps += 1; }
    ok = step_ok ? 1 : 0;
    const float v = viol[0] + viol[1] / fmaxf(s.gmax, 1.f) * 100.f;
    ++ps;
    __syncthreads();
    if (ok || ps >= budget) break;
    {
      const int psd = ps - 1;
      const bool one_sided = fminf(viol[0], viol[1]) <= 1e-9f;
      const bool stalled = !(v < 0.5f * vprev);
      const bool alternating = one_sided && psd >= 2 && psd < 4 && v < 0.5f * vprev2;
      if (psd >= 1 && stalled && !alternating) ++nstall;
      if (nstall >= cfg.patience && !last) break;   // uniform
    }
    vprev2 = vprev; vprev = v;
  }
  return ok;
}

// ----------------------------------------------------------------------------------------------------- the kernel
// Persistent workgroups: workgroup w solves QPs w, w + gridDim.x, ...; its factor workspace is ws_all + w * SG_WS_DOUBLES.
template <typename TM, typename TIO>
__global__ void __launch_bounds__(SG_NT)
mpcqp_stage_solve(const DevCfg* __restrict__ cfgp, const FastIn<TIO> in, TIO* ug, TIO* __restrict__ Xg, int* __restrict__ statusg,
                  int* __restrict__ itersg, float* __restrict__ resg, double* __restrict__ ws_all, const int N, const int Btot) {
  __shared__ SmemS s;
  const DevCfg& cfg = *cfgp;
  const int tid = threadIdx.x, NL = 4 * N, NQ = 6 * N, n = 12 * N, NX = (N + 1) * 13;
  double* const ws = ws_all + (size_t)blockIdx.x * SG_WS_DOUBLES;
  for (int b = blockIdx.x; b < Btot; b += gridDim.x) {
    SLeg Lg;
    Lg.leg = tid < NL;
    STAMP_INIT
    // ---- constants and inputs (src/mpc.py:242-255)
    if (tid < 6) { s.wP[tid] = cfg.w[tid]; if (tid >= 2) s.wQ[tid] = cfg.w[6 + tid]; }   // (wQ[0], wQ[1], wQ01: below, they turn with the yaw)
    if (tid == 0) {
      s.delta = cfg.delta; s.theta = cfg.theta; s.inv_m = cfg.inv_m; s.fmin = cfg.fmin; s.fmax = cfg.fmax;
      s.alpha_target = cfg.alpha > 0.0 ? cfg.alpha : ((cfg.flags & MPCQP_FLAG_POLISH) ? cfg.alpha_floor : 0.0);
      s.alpha = ((cfg.flags & MPCQP_FLAG_POLISH) && cfg.alpha < ALPHA_EASY) ? ALPHA_EASY : cfg.alpha;
      s.rho = (float)cfg.rho; s.iters = 0; s.psteps = 0; s.hard = 0;
      s.kkt[0] = s.kkt[1] = s.kkt[2] = 0.f;
    }
    int bad = 0;
    double* const xd = s.pist;   // x_des staging [N+1][13] over pist | zst (contiguous members)
    static_assert(offsetof(SmemS, zst) == offsetof(SmemS, pist) + sizeof(double) * (SG_NS + 1) * 12, "x_des staging spans pist | zst");
  static_assert(sizeof(SmemS) <= 160 * 1024, "LDS of a CU");
    for (int i = tid; i < NX; i += SG_NT) { const double v = (double)in.xdes[(size_t)b * NX + i]; xd[i] = v; bad |= !isfinite(v); }
    if (tid < 13) { const double v = (double)in.x0[(size_t)b * 13 + tid]; s.x0[tid] = v; bad |= !isfinite(v); }
    if (tid == 0) { const double v = (double)in.mu[b]; s.mu = v; bad |= !isfinite(v); }
    double rr[3] = {0, 0, 0};
    Lg.stance = false;
    if (Lg.leg) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { rr[a] = (double)in.r[(size_t)b * n + 3 * tid + a]; bad |= !isfinite(rr[a]); }
      Lg.stance = in.contact[(size_t)b * NL + tid] != 0;
    }
    bad = __syncthreads_or(bad);
    if (bad) {   // non-finite input -> zero outputs, status -1
      for (int i = tid; i < n; i += SG_NT) ug[(size_t)b * n + i] = (TIO)0;
      if (Xg) for (int i = tid; i < NX; i += SG_NT) Xg[(size_t)b * NX + i] = (TIO)0;
      if (tid == 0) { statusg[b] = MPCQP_STATUS_NONFINITE; itersg[b] = 0; if (resg) { resg[2 * b] = 0.f; resg[2 * b + 1] = 0.f; } }
      __syncthreads();
      continue;
    }
    if (tid == 0) {
      const double yaw = s.x0[2];  // src/mpc.py:64
      const double c = cos(yaw), sn = sin(yaw);
      s.cy = c; s.sy = sn;
      s.rzw0[0] = c * s.x0[6] - sn * s.x0[7];
      s.rzw0[1] = sn * s.x0[6] + c * s.x0[7];
      s.rzw0[2] = s.x0[8];
      // the omega weight in the rotated coordinates Q = Rz omega:  Rz diag(w6, w7) Rz'  (src/mpc.py:128-130 weights omega per world axis)
      const double wx = cfg.w[6], wy = cfg.w[7];
      s.wQ[0] = c * c * wx + sn * sn * wy; s.wQ[1] = sn * sn * wx + c * c * wy; s.wQ01 = c * sn * (wx - wy);
    }
    __syncthreads();
    {   // src/mpc.py:71-78, 98-107; compute_skew column a = r x e_a (src/utils.py:43-56)
      const double c = s.cy, sn = s.sy, Ib0 = cfg.Ib[0], Ib1 = cfg.Ib[1], Ib2 = cfg.Ib[2], m = Lg.stance ? 1.0 : 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double cx, cyv, cz;
        if (a == 0) { cx = 0; cyv = rr[2]; cz = -rr[1]; }
        else if (a == 1) { cx = -rr[2]; cyv = 0; cz = rr[0]; }
        else { cx = rr[1]; cyv = -rr[0]; cz = 0; }
        const double bx = (c * cx + sn * cyv) * Ib0, by = (-sn * cx + c * cyv) * Ib1, bz = cz * Ib2;
        const double tx = c * bx - sn * by, ty = sn * bx + c * by;
        Lg.B[a] = m * (c * tx - sn * ty);
        Lg.B[3 + a] = m * (sn * tx + c * ty);
        Lg.B[6 + a] = m * bz;
      }
      Lg.cm = Lg.stance ? s.inv_m : 0.0;
    }
    // free response minus target, stages k = 1..N, at s0[12 k + q] (P) / s0[12 k + 6 + q] (Q) -- index k - 1 is used so that it fits
    for (int e = tid; e < NQ; e += SG_NT) {
      const int j = e / 6, q = e - 6 * j, k = j + 1;
      const double d = s.delta, th = s.theta, g = s.x0[12], kd = (double)k * d;
      const double* xk = xd + 13 * k;
      double eP, eQ;
      if (q < 3) {
        eP = s.x0[q] + kd * s.rzw0[q] - xk[q];
        const double wx = xk[6], wy = xk[7], wz = xk[8];
        const double rd = q == 0 ? s.cy * wx - s.sy * wy : (q == 1 ? s.sy * wx + s.cy * wy : wz);
        eQ = s.rzw0[q] - rd;
      } else {
        const int a = q - 3;
        eP = s.x0[3 + a] + kd * s.x0[9 + a] - xk[3 + a];
        eQ = s.x0[9 + a] - xk[9 + a];
        if (a == 2) { eP += d * d * g * ((double)(k * (k - 1)) * 0.5 + th * (double)k); eQ += kd * g; }
      }
      s.s0[12 * j + q] = eP; s.s0[12 * j + 6 + q] = eQ;
    }
    __syncthreads();
    sg_adjoint(s, s.s0 - 12, nullptr, s.gam, N, tid);   // gam = C' 2W e  (src index k = j + 1 -> s0[12 j ..])
    __syncthreads();
    float q0[1] = {0.f};
    {
      const double* gj = s.gam + 6 * (min(tid, NL - 1) >> 2);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double g = Lg.cm * gj[3 + a];
#pragma unroll
        for (int i = 0; i < 3; ++i) g = fma(Lg.B[3 * i + a], gj[i], g);
        Lg.g[a] = g;
        if (Lg.leg) q0[0] = fmaxf(q0[0], fabsf((float)g));
      }
    }
    block_max<1, SG_NW>(q0, s.red, tid);
    if (tid == 0) s.gmax = q0[0];
#pragma unroll
    for (int a = 0; a < 3; ++a) { Lg.ua[a] = 0; Lg.pu[a] = 0; Lg.uv[a] = 0; }
#pragma unroll
    for (int i = 0; i < 5; ++i) { Lg.za[i] = 0; Lg.ya[i] = 0; Lg.py[i] = 0; }
    __syncthreads();
    STAMP(0);
    // ---- rounds: ADMM block, polish steps; on failure OSQP's rho adaptation and another round (mpcqp_wrench.h, same policy)
    const int max_iter = cfg.max_iter, polish_max = cfg.polish_max;
    const bool admm_only = !(cfg.flags & MPCQP_FLAG_POLISH);
    int ok = 0;
    int warm = 0;
    if (in.u_init) warm = sg_warm_start<TIO>(s, Lg, in.u_init + (size_t)b * n, in.y_state ? in.y_state + (size_t)b * (NL * 5) : nullptr, in.shift, N, tid);
    enum { R_WARM, R_ADMM, R_CONT };
    const int warm_tries = admm_only ? 0 : (warm == 1 ? WARM_POLISH : (warm == 2 ? 1 : 0));
    int kind = warm_tries > 0 ? R_WARM : R_ADMM, round = 0, cont_retry = 0;
    double keep_u[3] = {0, 0, 0}, keep_y[5] = {0, 0, 0, 0, 0};   // the last accepted continuation level
    for (;;) {
      int budget = kind == R_WARM ? min(warm_tries, polish_max) : 2 * polish_max;
      if (kind == R_ADMM) {
        // (a warm start from remembered (u, y): a first block 0.6 of the cold one, as in the dense engine)
        sg_admm<TM>(s, cfg, Lg, ws, round == 0 ? 1 : 0, round == 0 ? (warm >= 2 ? max(1, (SG_WARM_FRAC10 * (cfg.first_block > 0 ? cfg.first_block : cfg.check_every)) / 10) : cfg.first_block) : 0, N, tid,
                    round == 0);   // (the acceleration: in the first block only, see sg_admm)
        budget = admm_only ? 0 : (s.hard ? HARD_POLISH_FACTOR : 1) * polish_max;
      }
      const bool last = kind != R_ADMM || s.iters >= max_iter;
      __syncthreads();
      if (budget > 0) ok = sg_polish_round(s, cfg, Lg, ws, budget, last, N, tid);
      if (ok == 1 && s.alpha > s.alpha_target) {   // next continuation level, from this optimum and its multipliers
#pragma unroll
        for (int a = 0; a < 3; ++a) keep_u[a] = Lg.uv[a];
#pragma unroll
        for (int i = 0; i < 5; ++i) keep_y[i] = Lg.py[i];
        __syncthreads();
        if (tid == 0) { s.alpha_ok = s.alpha; s.alpha = fmax(s.alpha * 0.1, s.alpha_target); }
        __syncthreads();
        ok = 0; cont_retry = 0;
        kind = R_CONT;
        continue;
      }
      if (ok) break;
      if (kind == R_CONT) {   // level not reached: back to the last accepted point and a smaller step, a few times
        if (++cont_retry > 3) { ok = 3; break; }
#pragma unroll
        for (int a = 0; a < 3; ++a) Lg.pu[a] = keep_u[a];
#pragma unroll
        for (int i = 0; i < 5; ++i) Lg.py[i] = keep_y[i];
        __syncthreads();
        if (tid == 0) s.alpha = sqrt(s.alpha_ok * s.alpha);
        __syncthreads();
        continue;
      }
      if (kind == R_WARM) { kind = R_ADMM; continue; }
      if (!admm_only && s.iters >= max_iter) break;
      {
        const float ratio = sg_ratio(s, Lg, Lg.ua, Lg.za, Lg.ya, N, tid);
        if (admm_only) {
          const float tp = (float)cfg.eps_abs + (float)cfg.eps_rel * s.resid[2], td = (float)cfg.eps_abs + (float)cfg.eps_rel * s.resid[3];
          if (s.resid[0] <= tp && s.resid[1] <= td) ok = 2;
          __syncthreads();
          if (tid == 0) { s.kkt[0] = s.resid[1]; s.kkt[1] = s.resid[0]; s.kkt[2] = 0.f; }
          if (ok || s.iters >= max_iter) break;
        }
        const float rtol = admm_only ? 5.f : 2.f;
        __syncthreads();
        if (tid == 0 && isfinite(ratio) && (ratio > rtol || ratio < 1.f / rtol)) s.rho = fminf(fmaxf(s.rho * ratio, 1e-4f), 1e4f);
        __syncthreads();
      }
      ++round;
    }
    if (ok == 3) {   // a continuation level was not reached: the previous level's answer
#pragma unroll
      for (int a = 0; a < 3; ++a) Lg.uv[a] = keep_u[a];
    }
    STAMP(9);   // (all rounds: the phases inside stamp themselves, 1..7)
    // ---- outputs (src/mpc.py:265-268)
    double f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { f[c] = (ok == 1 || ok == 3) ? Lg.uv[c] : Lg.ua[c]; if (!Lg.stance) f[c] = 0; }
    if (Lg.leg) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ug[(size_t)b * n + 3 * tid + c] = (TIO)f[c];
      if (in.y_state) {
#pragma unroll
        for (int i = 0; i < 5; ++i) in.y_state[(size_t)b * (NL * 5) + 5 * tid + i] = (float)(ok == 1 ? Lg.py[i] : Lg.ya[i]);
      }
    }
    if (Xg) {
      double g3[3];
      sg_grad(s, Lg, f, g3, N, tid);   // leaves the deviation states of f in s.zst
      const double d = s.delta, th = s.theta, g = s.x0[12];
      for (int i = tid; i < NX; i += SG_NT) {
        const int k = i / 13, c = i - 13 * k;
        double v;
        if (c == 12 || k == 0) v = s.x0[c];
        else if (c < 3) v = s.x0[c] + (double)k * d * s.rzw0[c] + s.zst[12 * k + c];
        else if (c < 6) {
          v = s.x0[c] + (double)k * d * s.x0[6 + c] + s.zst[12 * k + c];
          if (c == 5) v += d * d * g * ((double)(k * (k - 1)) * 0.5 + th * (double)k);
        } else if (c < 9) {
          const double ax = s.zst[12 * k + 6], ay = s.zst[12 * k + 7], az = s.zst[12 * k + 8];
          v = s.x0[c] + (c == 6 ? s.cy * ax + s.sy * ay : (c == 7 ? -s.sy * ax + s.cy * ay : az));   // omega = Rz'(Rz omega)
        } else {
          v = s.x0[c] + s.zst[12 * k + 6 + (c - 6)];
          if (c == 11) v += (double)k * d * g;
        }
        Xg[(size_t)b * NX + i] = (TIO)v;
      }
    }
    if (tid == 0) {
      statusg[b] = ok == 1 ? MPCQP_STATUS_SOLVED_POLISHED : (ok == 2 ? MPCQP_STATUS_SOLVED_ADMM : MPCQP_STATUS_MAX_ITER);
      itersg[b] = s.iters + 1000 * s.psteps;
      if (resg) { resg[2 * b] = s.kkt[1]; resg[2 * b + 1] = fmaxf(s.kkt[2], s.kkt[0]); }
    }
    __syncthreads();
    STAMP(8);
  }
}

}  // namespace
