// mpcqp_wrench.h -- wrench-space engine: one QP per WAVE (horizon 10) or per four waves (horizon 20).
//
// The condensed Hessian of the reference QP (src/mpc.py:110-134) factors through the per-stage net wrench:
//     H = 2 alpha I + T' K T,      T = blockdiag(T_j),  T_j (6 x 12): stage forces -> [Rz tau_j ; a_j]
//     tau_j = sum_l Ihat^-1 (r_l x f_l)   (src/mpc.py:78,98-107),   a_j = sum_l c_l f_l / m
//     K = (+)_{q<6} K_q,   K_q[j][j'] = 2 (wP_q c1[j][j'] + wQ_q c0[j][j'])      (N x N per wrench component)
// with c0 = d^2 (N - max(j,j')), c1 = d^4 sum_{k>max(j,j')} (k-1-j+th)(k-1-j'+th) the stage-pair tables mpcqp_create builds, wP = w[0..5] (Theta, p), wQ = w[6..11] (omega, v).  In the Rz-rotated
// angular coordinates K is block diagonal in q as long as the omega weight is isotropic in x,y (the reference's is,
// src/mpc.py:128-130) -- and then K, K^-1 are CONSTANTS of the configuration, inverted once on the host in fp64.
// Both linear systems of the solve are "diagonal + T'KT" (ADMM: D = 2 alpha + sigma + rho G'G; polish: the same on the
// free variables), so by Woodbury
//     M^-1 = D^-1 - D^-1 T' S^-1 T D^-1,      S = K^-1 + T D^-1 T'      (6N x 6N;  T D^-1 T' is block diagonal)
// a 60 x 60 sweep instead of a 120 x 120 one (8 x fewer flops), a 60 x 60 mat-vec per ADMM iteration instead of a
// 120 x 120 one, and an 8 x 8 register tile per lane on an 8 x 8 lane grid: one QP per wave, no workgroup barrier
// anywhere in the horizon-10 solve (LDS traffic of a wave is in order), 2 waves per SIMD.
// Precision: the explicit fp32 inverse of S is accurate enough for ADMM (whose job is the active set) but not for the
// polish -- cond(S) reaches 1e6 there (K^-1 ~ 1e-3 next to T D^-1 T' ~ 1e3 with D = 2 alpha) and the error of S^-1 is
// amplified by D^-1 T'T D^-1 (numpy study, tools/wrench_proto.py: the fp32 refinement diverges on half of the QPs).
// The polish therefore builds and sweeps its S in fp64 (v_fma_f64 issues at the unpacked fp32 rate on gfx950): the
// solve is then exact to ~1e-9 and no iterative refinement is needed (one solve + two gradients per step).
#pragma once
#include "mpcqp_common.h"   // FastIn, OrderBuf, the dispatch-order pre-pass, policy constants

namespace {

template <int N_>
struct WG {
  static constexpr int N = N_, NL = 4 * N, n = 12 * N, NQ = 6 * N;
  static constexpr int G = N <= 10 ? 8 : 16;      // lane grid G x G of 8 x 8 tiles
  static constexpr int NT = G * G, NW = NT / 64;
  static constexpr int DP = 8 * G;                // padded wrench dimension
  static constexpr int NB = (NQ + 7) / 8;         // pivot blocks
  static_assert(NQ <= DP, "horizon too long for the lane grid");
};

// Constant tables of a configuration (device memory, built by mpcqp_create).
struct WrTabs {
  const double* K;        // [6][N][N]   K_q
  const float* kinv32;    // [6][N][N]   K_q^-1 (w_tile_init gathers the tile entries from it)
  const double* kinv64;   // [6][N][N]
};

template <typename TV, int N>
struct SmemW {
  using Geo = WG<N>;
  TV x0[13];
  TV mu, cy, sy;
  TV rzw0[3];
  TV wP[6], wQ[6];
  TV delta, theta, alpha, inv_m, fmin, fmax;
  TV alpha_target, alpha_ok;     // continuation: where it ends; the last regulariser level whose optimum was accepted
  TV Bl[Geo::NL * 9];            // per leg-stage: Rz Ihat^-1 [r]x, masked by contact (row i = angular component, col a = force axis)
  TV cm[Geo::NL];                // contact / m
  TV gam[Geo::NQ];               // gradient of the cost in wrench space at u = 0
  TV gl[Geo::n];                 // linear term g = T' gam
  TV uv[Geo::n];                 // point of the structured gradient (the gradient itself comes back in registers)
  TV ww[Geo::NQ], kap[Geo::NQ];  // wrench of uv, K ww + gam
  union {   // the staged inputs are dead once setup has built B_l and the free response: they share the bytes of the polish iterate
    struct { TV pu[Geo::n], py[Geo::NL * 5]; };
    struct { TV rr[Geo::n]; TV xd[(N + 1) * 13]; };   // lever arms, x_des
  };
  union {   // ... and the free response (minus target) those of the ADMM iterate
    struct { TV ua[Geo::n], za[Geo::NL * 5], ya[Geo::NL * 5]; };   // last ADMM iterate (ya = multipliers, unscaled); TV so that an
                                                                      // all-fp64 ADMM keeps its digits from block to block
    struct { TV e0P[Geo::NQ], e0Q[Geo::NQ]; };   // stage k = j + 1 at [6 j + q]: P = (Theta, p), Q = (Rz omega, v)
  };
  alignas(16) double E[N * 36];                          // T D^-1 T' blocks (TM-typed view)
  alignas(16) double piv[2 * Geo::DP];                   // pivot-row broadcast (TM-typed view)
  alignas(16) double bv[Geo::DP], cv[Geo::DP];           // mat-vec in / out (TM-typed view)
  float red[Geo::NW * 4];
  float aared[Geo::NW * 12];     // Anderson acceleration: partial inner products of the waves (four waves per QP)
  float aah[N > 10 ? 45 * Geo::NL : 1];   // ... and, at horizon 20, the leg-stages' history between extrapolations ([slot][leg-stage]; at horizon 10 it stays in registers)
  float kkt[4];
  float resid[4];                // residuals of the last ADMM iterate that w_ratio looked at
  float gmax, rho, ratio;
  int iters, psteps, hard, warm, bad;
  unsigned chgmask[(Geo::NL + 31) / 32];   // four waves per QP: leg-stages whose active set changed (polish)
  unsigned ahash;                          // hash of the active set being filed (polish: cycle detection)
  unsigned ahist[32];                      // ... and of the active sets this round has already tried
  uint8_t ct[Geo::NL];
  uint8_t aset[Geo::NL];         // active set of the current polish step (ActSet code)
};

__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// A zero the optimiser cannot see through.  A 64-bit constant needs a register pair, LLVM hoists such pairs out of the persistent QP loop,
// and with the register file full it then SPILLS the constant at kernel entry and reloads it per QP (scratch stores are written through:
// 10 bytes per lane and wave of HBM writes for three zeros and a one, profiles/r03f_hbm_traffic.json).  Materialised where it is used instead.
__device__ __forceinline__ double opaque_zero_f64() {
  unsigned lo, hi;
  asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(lo), "=v"(hi));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename T> __device__ __forceinline__ T opaque_zero() {
  if constexpr (sizeof(T) == 8) return (T)opaque_zero_f64();
  else { float z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return (T)z; }
}
// A wave-uniform float, moved to a scalar register (loop-carried uniform values otherwise occupy a vector register each).
__device__ __forceinline__ float ufloat(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
// The thread index of a register phase.  One wave per QP: recomputed from the lane counter where it is needed (two VALU
// instructions, not hoistable, nothing kept alive between phases -- a copy of threadIdx.x held across the fp64 sweep is spilled,
// and so is every LDS address LLVM derives from it ahead of the round loop).  Four waves: an opaque copy of threadIdx.x.
template <int NW>
__device__ __forceinline__ int fresh_tid(int tid0) {
  if constexpr (NW == 1) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  } else {
    return opaque(tid0);
  }
}

template <int NW>
__device__ __forceinline__ void wsync() {
  if constexpr (NW == 1) {   // one wave: its LDS operations execute in order; only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#ifdef MPCQP_W_WAIT
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
#endif
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

template <int Q, int NW>
__device__ __forceinline__ void wmax(float (&v)[Q], float* red, int tid) {
  if constexpr (NW == 1) {
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = wave_max(v[q]);
  } else {
    block_max<Q, NW>(v, red, tid);
  }
}

template <int Q, int NW>
__device__ __forceinline__ void wsum(float (&v)[Q], float* red, int tid) {
  if constexpr (NW == 1) {
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = wave_sum(v[q]);
  } else {
    block_sum<Q, NW>(v, red, tid);
  }
}

template <typename T> __device__ __forceinline__ T quad_sum(T v) { v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); return v; }

// Reduce-scatter of 8 values over the 8 lanes of a group: lane gc ends with the group total of element gc.
template <typename T>
__device__ __forceinline__ T rs8(const T (&v)[8], int gc) {
  const bool hi = (gc & 4) != 0, b1 = (gc & 2) != 0, b0 = (gc & 1) != 0;
  T t[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) { const T keep = hi ? v[4 + m] : v[m], send = hi ? v[m] : v[4 + m]; t[m] = keep + dpp_mov<0x141>(send); }
  T s2[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) { const T keep = b1 ? t[2 + m] : t[m], send = b1 ? t[m] : t[2 + m]; s2[m] = keep + dpp_mov<0x4E>(send); }
  const T keep = b0 ? s2[1] : s2[0], send = b0 ? s2[0] : s2[1];
  return keep + dpp_mov<0xB1>(send);
}

__device__ __forceinline__ float w_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double w_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// ----------------------------------------------------------------------------------------------------- register tile
template <typename TM> struct WTile;
template <> struct WTile<float> { f2 v[8][4]; };
template <> struct WTile<double> { double v[8][8]; };

__device__ __forceinline__ float tget(const WTile<float>& t, int i, int j) { return (j & 1) ? t.v[i][j >> 1].y : t.v[i][j >> 1].x; }
__device__ __forceinline__ double tget(const WTile<double>& t, int i, int j) { return t.v[i][j]; }
__device__ __forceinline__ void tset(WTile<float>& t, int i, int j, float x) { if (j & 1) t.v[i][j >> 1].y = x; else t.v[i][j >> 1].x = x; }
__device__ __forceinline__ void tset(WTile<double>& t, int i, int j, double x) { t.v[i][j] = x; }

template <typename TM> __device__ __forceinline__ void ld8(const TM* p, TM (&o)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&o)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<double>(const double* p, double (&o)[8]) {
#pragma unroll
  for (int h = 0; h < 4; ++h) { const double2 a = reinterpret_cast<const double2*>(p)[h]; o[2 * h] = a.x; o[2 * h + 1] = a.y; }
}
template <typename TM> __device__ __forceinline__ void st8(TM* p, const TM (&o)[8]);
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&o)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(o[0], o[1], o[2], o[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(o[4], o[5], o[6], o[7]);
}
template <> __device__ __forceinline__ void st8<double>(double* p, const double (&o)[8]) {
#pragma unroll
  for (int h = 0; h < 4; ++h) reinterpret_cast<double2*>(p)[h] = make_double2(o[2 * h], o[2 * h + 1]);
}

// Opaque redefinition of the whole tile (no instructions): arithmetic on the tile cannot move across this point.
__device__ __forceinline__ void tpin(WTile<float>& t) {
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(t.v[i][0]), "+v"(t.v[i][1]), "+v"(t.v[i][2]), "+v"(t.v[i][3]));
}
__device__ __forceinline__ void tpin(WTile<double>& t) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
    asm volatile("" : "+v"(t.v[i][0]), "+v"(t.v[i][1]), "+v"(t.v[i][2]), "+v"(t.v[i][3]), "+v"(t.v[i][4]), "+v"(t.v[i][5]), "+v"(t.v[i][6]), "+v"(t.v[i][7]));
}

// tile[i][j] -= m[i] * vc[j]
__device__ __forceinline__ void rank1(WTile<float>& t, const float (&m)[8], const float (&vc)[8]) {
  f2 c2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c2[j] = mk2(vc[2 * j], vc[2 * j + 1]);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f2 mi = splat2(-m[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) t.v[i][j] = __builtin_elementwise_fma(mi, c2[j], t.v[i][j]);
  }
}
__device__ __forceinline__ void rank1(WTile<double>& t, const double (&m)[8], const double (&vc)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double mi = -m[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) t.v[i][j] = fma(mi, vc[j], t.v[i][j]);
  }
}
// acc[i] = sum_j tile[i][j] x[j]
__device__ __forceinline__ void tilemv(const WTile<float>& t, const float (&x)[8], float (&acc)[8]) {
  f2 x2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) x2[j] = mk2(x[2 * j], x[2 * j + 1]);
  f2 a[8];   // eight independent chains, column-major order (back-to-back dependent packed FMAs cost a wait state each)
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = t.v[i][0] * x2[0];
#pragma unroll
  for (int j = 1; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(t.v[i][j], x2[j], a[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = a[i].x + a[i].y;
}
__device__ __forceinline__ void tilemv(const WTile<double>& t, const double (&x)[8], double (&acc)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = t.v[i][0] * x[0];
#pragma unroll
  for (int j = 1; j < 8; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fma(t.v[i][j], x[j], acc[i]);
  }
}

// tile <- K^-1 (constant of the configuration) + E (block diagonal, 6 x 6 per stage, in LDS).
// K^-1 = (+)_q K_q^-1 couples only equal wrench components: entry (R, C) of the stage-major ordering (index 6 j + q) is
// kq[q][j][j'] when R = C (mod 6) and zero otherwise, so a lane's 8 x 8 tile holds one such entry per row, or two when the
// first falls in tile column 0 or 1 (the second is six columns on).  Pass 1 gathers these <= 16 values from the compact
// table kq (6 N^2 entries: 2.4 / 4.8 KB at horizon 10, resident in the CU's vector L1) and places them by column selects --
// a full tile-layout table (16 / 32 KB per load, every lane 16 x 16 B) misses L1 for every wave of the CU and its loads
// queued for ~30 k cycles at full occupancy (the miss queue of the L1, not bandwidth: phase stamps).
// Pass 2 adds E: entry (R, C) of stage j = R / 6 sits at E[6 R + C - 6 j] and exists iff 0 <= C - 6 j < 6, i.e. for tile
// column c iff (c - lo) <u 6 with lo = 6 j - 8 gc per tile row; outside a row's run the index is clamped to 0 and the value masked.
template <typename TM, int N>
__device__ __forceinline__ void w_tile_init(WTile<TM>& t, const TM* __restrict__ kq, const TM* __restrict__ E, int gr, int gc, int tid) {
  constexpr int NT = WG<N>::NT, NQ = WG<N>::NQ;
  asm volatile("" : "+v"(gr), "+v"(gc), "+v"(tid));   // (opaque: keeps the per-lane index arithmetic out of the enclosing loops' preheaders)
  TM v0[8], v1[8];
  int c0[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int R = 8 * gr + r, jR = (R * 43) >> 8, qR = R - 6 * jR;   // R / 6 for R < 128
    const int d = R - 8 * gc;                                          // tile column of the diagonal (any sign)
    const int dm = d + 126, m = dm - 6 * ((dm * 171) >> 10);           // d mod 6 in [0, 6): x / 6 = (171 x) >> 10 for x < 500
    const int C0 = 8 * gc + m, j0 = (C0 * 43) >> 8;
    const bool rowok = R < NQ, ok0 = rowok && C0 < NQ, ok1 = rowok && m < 2 && C0 + 6 < NQ;
    const TM* row = kq + (qR * N + min(jR, N - 1)) * N;
    v0[r] = row[ok0 ? j0 : 0];
    v1[r] = row[ok1 ? j0 + 1 : 0];
    if (!ok0) v0[r] = (TM)0;
    if (!ok1) v1[r] = (TM)0;
    c0[r] = m;
    if (!rowok) { v0[r] = (TM)1; c0[r] = (d >= 0 && d < 8) ? d : 8; }   // identity on the padding: the sweep pivots on all 8 G rows
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      TM v = c == c0[r] ? v0[r] : (TM)0;
      if (c >= 6) v = c == c0[r] + 6 ? v1[r] : v;
      tset(t, r, c, v);
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int R = 8 * gr + r, jR = (R * 43) >> 8;            // R / 6 for R < 128
    const int lo = 6 * jR - 8 * gc;
    const int base = R < NQ ? 6 * R - lo : 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const bool in = (unsigned)(c - lo) < 6u && R < NQ;
      const TM e = E[in ? base + c : 0];                       // (index clamped, value masked: no address outside E is ever formed)
      tset(t, r, c, tget(t, r, c) + (in ? e : (TM)0));
    }
  }
}

// In-register symmetric sweep of all NQ pivots: tile <- -S^-1.  Lanes of grid row og publish pivot row k = 8 og + rr; by
// symmetry the same vector serves as the pivot column.  One LDS broadcast per pivot; no barrier when the QP is one wave.
template <typename TM, int N>
__device__ __forceinline__ void w_sweep(WTile<TM>& t, TM* __restrict__ piv, int gr, int gc) {
  constexpr int NW = WG<N>::NW, DP = WG<N>::DP, G = WG<N>::G;
  int step = 0;
  for (int og = 0; og < G; ++og) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      if (8 * og + rr >= WG<N>::NQ) break;   // uniform; the padding rows carry an identity block and need no pivot
      TM* vb = piv + (NW > 1 ? (step & 1) * DP : 0);
      if (gr == og) {
        TM row[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] = tget(t, rr, j);
        st8<TM>(vb + 8 * gc, row);
      }
      wsync<NW>();
      const TM p = w_rcp(vb[8 * og + rr]);
      TM vc[8], vr[8];
      ld8<TM>(vb + 8 * gc, vc);
      ld8<TM>(vb + 8 * gr, vr);
      const bool prow = gr == og, pcol = gc == og;
      TM m[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { vr[i] *= p; m[i] = vr[i]; }
      // the owner's pivot row becomes p * row: its tile row IS the published row, so the multiplier 1 - p does it
      // (the same trick on the pivot column -- multiplier 1 / p - 1 on the published pivot element -- costs eps / p relative on
      //  the column entries, which the alpha = 1e-4 polish systems do not survive: measured, tests/test_gpu_parity.py)
      m[rr] = prow ? (TM)1 - p : vr[rr];
      rank1(t, m, vc);
#pragma unroll
      for (int i = 0; i < 8; ++i) {   // pivot column (selects: a divergent region that writes the tile makes the compiler copy it)
        const TM v = (i == rr && prow) ? -p : vr[i];
        tset(t, i, rr, pcol ? v : tget(t, i, rr));
      }
      if constexpr (NW == 1) wsync<1>();   // the next pivot's publication must not overtake this pivot's reads (compiler order only)
      // Keep the pivots apart: left alone, the optimiser pipelines several pivots (publishes the next row early, defers the
      // rest of the rank-1 updates), which keeps two or three pivots' row / column vectors alive next to the tile and spills
      // inside the loop.
      tpin(t);
      ++step;
    }
  }
}

#ifdef MPCQP_SYM_SWEEP
// ----------------------------------------------------------------------------------------------------- symmetric sweep (polish, horizon 10)
// EXPERIMENT, compiled only with -DMPCQP_SYM_SWEEP (round 3: correct -- all GPU tests green -- and no faster: profiles/r03f_symmetric_sweep.txt).
// S = K^-1 + E is symmetric, and so is every intermediate of the symmetric sweep: sweep only the lower block triangle.  In the
// stage-major ordering the 60 x 60 matrix is a 10 x 10 grid of 6 x 6 stage blocks; lane l = bi (bi + 1) / 2 + bj holds block (bi, bj),
// bi >= bj -- 55 lanes, 36 values each instead of 64: 36 instead of 64 fp64 FMAs per pivot and lane.  K^-1 = (+)_q K_q^-1 puts ONE entry
// on each diagonal position of a block and E_j is the diagonal block (j, j) itself, so the initialisation is six table loads per lane.
// A pivot row k = 6 p + r is published by the lanes of block row p (their local row r) and, for the columns right of the diagonal, by
// the lanes of block column p (their local COLUMN r: S[k][c] = S[c][k]).  Afterwards the inverse is handed to the 8 x 8 lane grid of
// 8 x 8 tiles that the solves and the rank-one updates work on, block row by block row through the 360 doubles of s.E -- and that
// hand-over (~4 k cycles) takes back what the sweep saves (95 instructions per pivot with 38 fp64 FMAs against 101 with 64).  What
// would make it pay: solves and updates on the symmetric layout too.
struct WSym { double v[6][6]; };

__device__ __forceinline__ void sympin(WSym& t) {
#pragma unroll
  for (int a = 0; a < 6; ++a) asm volatile("" : "+v"(t.v[a][0]), "+v"(t.v[a][1]), "+v"(t.v[a][2]), "+v"(t.v[a][3]), "+v"(t.v[a][4]), "+v"(t.v[a][5]));
}
__device__ __forceinline__ void ld6(const double* p, double (&o)[6]) {
#pragma unroll
  for (int h = 0; h < 3; ++h) { const double2 a = reinterpret_cast<const double2*>(p)[h]; o[2 * h] = a.x; o[2 * h + 1] = a.y; }
}
__device__ __forceinline__ void st6(double* p, const double (&o)[6]) {
#pragma unroll
  for (int h = 0; h < 3; ++h) reinterpret_cast<double2*>(p)[h] = make_double2(o[2 * h], o[2 * h + 1]);
}

// tile <- -(K^-1 + E)^-1 in the 8 x 8 grid layout, by way of the symmetric half.  kq: [6][10][10] K_q^-1; E: the ten 6 x 6 blocks of
// T D^-1 T' (consumed by the initialisation, then reused as the hand-over buffer); piv: pivot-row broadcast.
__device__ __forceinline__ void w_sym_build(WTile<double>& tile, const double* __restrict__ kq, double* __restrict__ E, double* __restrict__ piv,
                                            int tid) {
  constexpr int N = 10;
  asm volatile("" : "+v"(tid));
  const int l = min(tid, 54);
  const int bi = (l >= 1) + (l >= 3) + (l >= 6) + (l >= 10) + (l >= 15) + (l >= 21) + (l >= 28) + (l >= 36) + (l >= 45);
  const int bj = l - ((bi * (bi + 1)) >> 1);
  const bool on = tid < 55;
  WSym t;
  {   // initialisation
    double kd[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) kd[a] = kq[(a * N + bi) * N + bj];
    const bool dg = bi == bj;
    const double* Eb = E + 36 * bi;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const double e = Eb[6 * a + b];
        t.v[a][b] = (a == b ? kd[a] : 0.0) + (dg ? e : 0.0);
      }
    }
  }
  wsync<1>();   // (E has been read: its bytes serve the hand-over below; nothing else writes them during the sweep)
  for (int p = 0; p < N; ++p) {
    const bool prow = bi == p, pcol = bj == p;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      {   // publish pivot row k = 6 p + r
        double o[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) o[c] = prow ? t.v[r][c] : t.v[c][r];   // block row p: local row r;  block column p below the diagonal: local column r
        if (on && (prow || pcol)) st6(piv + 6 * (prow ? bj : bi), o);
      }
      wsync<1>();
      const double pinv = w_rcp(piv[6 * p + r]);
      double vr[6], vc[6], m[6];
      ld6(piv + 6 * bi, vr);
      ld6(piv + 6 * bj, vc);
#pragma unroll
      for (int a = 0; a < 6; ++a) { vr[a] *= pinv; m[a] = vr[a]; }
      m[r] = prow ? 1.0 - pinv : vr[r];     // (the owner's pivot row comes out of the same FMAs: its tile row IS the published row)
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double ma = -m[a];
#pragma unroll
        for (int b = 0; b < 6; ++b) t.v[a][b] = fma(ma, vc[b], t.v[a][b]);
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {         // pivot column (selects, as in w_sweep)
        const double v = (a == r && prow) ? -pinv : vr[a];
        t.v[a][r] = pcol ? v : t.v[a][r];
      }
      wsync<1>();
      sympin(t);
    }
  }
  // hand-over to the 8 x 8 grid: block row p of the full matrix (6 rows x 60 columns) through E
  const int gr = tid >> 3, gc = tid & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) tile.v[i][j] = (8 * gr + i >= 60 && i == j && gr == gc) ? 1.0 : 0.0;   // identity on the padding
  }
  for (int p = 0; p < N; ++p) {
    if (on && bi == p) {                    // rows 6 p .. 6 p + 5, columns of block bj
#pragma unroll
      for (int a = 0; a < 6; ++a) st6(E + 60 * a + 6 * bj, t.v[a]);
    } else if (on && bj == p) {             // the same rows, columns of block bi > p: the transposed block
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        double o[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) o[a] = t.v[a][b];
        st6(E + 60 * b + 6 * bi, o);
      }
    }
    wsync<1>();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int a = 8 * gr + i - 6 * p;     // local row of this block row, if any
      if ((unsigned)a < 6u) {
        double o[8];
        ld8<double>(E + 60 * a + 8 * gc, o);     // (gc = 7 reads four doubles into the next row / the bytes after E: masked below)
#pragma unroll
        for (int j = 0; j < 8; ++j) tile.v[i][j] = (8 * gc + j < 60) ? o[j] : 0.0;
      }
    }
    wsync<1>();
  }
}
#endif

// y = S^-1 x for x in LDS (bv, padded layout): returns element 8 gr + gc (valid on lanes gc < 8) and writes it to cv.
template <typename TM, int N>
__device__ __forceinline__ void w_matvec(const WTile<TM>& t, const TM* __restrict__ bv, TM* __restrict__ cv, int gr, int gc) {
  constexpr int G = WG<N>::G;
  TM x[8], acc[8];
  ld8<TM>(bv + 8 * gc, x);
  tilemv(t, x, acc);
  if constexpr (G == 16) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += dpp_mov<0x140>(acc[i]);   // row_mirror: both halves of the 16 now hold the pair sums
  }
  const TM tot = -rs8<TM>(acc, gc & 7);   // tile = -S^-1
  if (gc < 8) cv[8 * gr + gc] = tot;
}

// Per-leg data of a linear solve with M = D + A-stack' K A-stack: columns of the 6 x 3 wrench map and the inverse diagonal.
template <typename TM>
struct LegSys {
  TM A[3][6];    // A[c][q]: wrench component q of reduced variable c
  TM dinv[3];
};

// E_j = sum_legs A diag(dinv) A' -> LDS (full 6 x 6 per stage).  Leg lanes; ends with a sync.
template <typename TM, int N>
__device__ __forceinline__ void w_build_E(const LegSys<TM>& L, TM* __restrict__ E, int tid) {
  constexpr int NL = WG<N>::NL, NW = WG<N>::NW;
  TM e[21];
  int k = 0;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
#pragma unroll
    for (int p = q; p < 6; ++p) {
      TM a = L.dinv[0] * L.A[0][q] * L.A[0][p];
      a = fma(L.dinv[1] * L.A[1][q], L.A[1][p], a);
      a = fma(L.dinv[2] * L.A[2][q], L.A[2][p], a);
      e[k++] = quad_sum(a);
    }
  }
  if (tid < NL && (tid & 3) == 0) {
    TM* Ej = E + 36 * (tid >> 2);
    k = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int p = q; p < 6; ++p) { Ej[6 * q + p] = e[k]; Ej[6 * p + q] = e[k]; ++k; }
    }
  }
  wsync<NW>();
}

// x = M^-1 rhs through the swept tile: x = dinv (rhs - A' S^-1 A-stack dinv rhs).  All lanes call; leg lanes hold data.
template <typename TM, int N>
__device__ __forceinline__ void w_solve(const WTile<TM>& t, const LegSys<TM>& L, const TM (&rhs)[3], TM (&x)[3], TM* __restrict__ bv,
                                        TM* __restrict__ cv, int tid, int gr, int gc) {
  constexpr int NL = WG<N>::NL, NW = WG<N>::NW;
  TM a[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) a[c] = L.dinv[c] * rhs[c];
  TM b[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) b[q] = quad_sum(fma(L.A[2][q], a[2], fma(L.A[1][q], a[1], L.A[0][q] * a[0])));
  {   // lanes 0..2 of the quad write two components each (lane 3 repeats lane 2)
    const int l = min(tid & 3, 2);
    const TM v0 = l == 0 ? b[0] : (l == 1 ? b[2] : b[4]), v1 = l == 0 ? b[1] : (l == 1 ? b[3] : b[5]);
    if (tid < NL) { TM* d = bv + 6 * (tid >> 2) + 2 * l; d[0] = v0; d[1] = v1; }
  }
  wsync<NW>();
  w_matvec<TM, N>(t, bv, cv, gr, gc);
  wsync<NW>();
  const TM* cj = cv + 6 * (min(tid, NL - 1) >> 2);
  TM c6[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) c6[q] = cj[q];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    TM s = L.A[c][0] * c6[0];
#pragma unroll
    for (int q = 1; q < 6; ++q) s = fma(L.A[c][q], c6[q], s);
    x[c] = a[c] - L.dinv[c] * s;
  }
}

// ----------------------------------------------------------------------------------------------------- inputs
// Operator tuple (src/mpc.py:242-255) -> LDS (x0, lever arms, contact, x_des, mu): every global load of the QP is issued before
// the first one is consumed; returns the per-thread non-finite flag.
template <typename TV, typename TIO, int N>
__device__ __forceinline__ int w_load(SmemW<TV, N>& s, const FastIn<TIO>& in, size_t b, int tid) {
  constexpr int NT = WG<N>::NT, n = WG<N>::n, NL = WG<N>::NL, NX = (N + 1) * 13;
  constexpr int RX = (NX + NT - 1) / NT, RR = (n + NT - 1) / NT, RC = (NL + NT - 1) / NT;
  TIO vx[RX], vr[RR], v0 = 0, vm = 0;
  uint8_t vc[RC];
  const TIO* xp = in.xdes + b * (size_t)NX;
  const TIO* rp = in.r + b * (size_t)n;
  const uint8_t* cp = in.contact + b * (size_t)NL;
#pragma unroll
  for (int q = 0; q < RX; ++q) { const int i = tid + q * NT; vx[q] = i < NX ? xp[i] : (TIO)0; }
#pragma unroll
  for (int q = 0; q < RR; ++q) { const int i = tid + q * NT; vr[q] = i < n ? rp[i] : (TIO)0; }
#pragma unroll
  for (int q = 0; q < RC; ++q) { const int i = tid + q * NT; vc[q] = i < NL ? cp[i] : (uint8_t)0; }
  if (tid < 13) v0 = in.x0[b * 13 + tid];
  if (tid == 0) vm = in.mu[b];
  int bad = 0;
#pragma unroll
  for (int q = 0; q < RX; ++q) { const int i = tid + q * NT; if (i < NX) { s.xd[i] = (TV)vx[q]; bad |= !isfinite(vx[q]); } }
#pragma unroll
  for (int q = 0; q < RR; ++q) { const int i = tid + q * NT; if (i < n) { s.rr[i] = (TV)vr[q]; bad |= !isfinite(vr[q]); } }
#pragma unroll
  for (int q = 0; q < RC; ++q) { const int i = tid + q * NT; if (i < NL) s.ct[i] = vc[q] ? 1 : 0; }
  if (tid < 13) { s.x0[tid] = (TV)v0; bad |= !isfinite(v0); }
  if (tid == 0) { s.mu = (TV)vm; bad |= !isfinite(vm); }
  return bad;
}

// Structured gradient at s.uv: s.gv = 2 alpha u + T'(K T u + gam).  Leaves the wrench in s.ww.  Needs s.uv visible; the
// caller's leg gradient is also returned in registers.  Ends with a sync.
template <typename TV, int N>
__device__ __forceinline__ void w_grad(SmemW<TV, N>& s, const double* __restrict__ Ktab, int tid, TV (&gr)[3]) {
  constexpr int NL = WG<N>::NL, NQ = WG<N>::NQ, NW = WG<N>::NW, NT = WG<N>::NT;
  static_assert(NQ <= NT, "one wrench component per lane");
  const int L = min(tid, NL - 1);
  // this lane's row of K_q, requested first: its L2 latency hides behind the wrench phase
  const int e = min(tid, NQ - 1), ej = e / 6, eq = e - 6 * ej;
  constexpr bool EARLY = N <= 10;   // (twenty doubles more next to the fp64 tile would spill)
  const double2* K2 = reinterpret_cast<const double2*>(Ktab + ((size_t)eq * N + ej) * N);
  double Kr[N];
  if constexpr (EARLY) {
#pragma unroll
    for (int h = 0; h < N / 2; ++h) { const double2 v = K2[h]; Kr[2 * h] = v.x; Kr[2 * h + 1] = v.y; }
  }
  TV f[3], B[9];
#pragma unroll
  for (int a = 0; a < 3; ++a) f[a] = s.uv[3 * L + a];
#pragma unroll
  for (int i = 0; i < 9; ++i) B[i] = s.Bl[9 * L + i];
  const TV cm = s.cm[L];
  {
    TV w6[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) w6[i] = quad_sum(fma(B[3 * i + 2], f[2], fma(B[3 * i + 1], f[1], B[3 * i] * f[0])));
#pragma unroll
    for (int a = 0; a < 3; ++a) w6[3 + a] = quad_sum(cm * f[a]);
    if (tid < NL && (tid & 3) == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) s.ww[6 * (tid >> 2) + q] = w6[q];
    }
  }
  wsync<NW>();
  {
    if constexpr (!EARLY) {
#pragma unroll
      for (int h = 0; h < N / 2; ++h) { const double2 v = K2[h]; Kr[2 * h] = v.x; Kr[2 * h + 1] = v.y; }
    }
    TV acc = s.gam[e];
#pragma unroll
    for (int jp = 0; jp < N; ++jp) acc = fma((TV)Kr[jp], s.ww[6 * jp + eq], acc);
    if (tid < NQ) s.kap[e] = acc;
  }
  wsync<NW>();
  {
    const TV* kj = s.kap + 6 * (L >> 2);
    const TV a2 = (TV)2 * s.alpha;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      TV g = fma(a2, f[a], cm * kj[3 + a]);
#pragma unroll
      for (int i = 0; i < 3; ++i) g = fma(B[3 * i + a], kj[i], g);
      gr[a] = g;
    }
  }
  wsync<NW>();
}

// Per-QP setup: constants (once per workgroup), inputs, B_l, free response, gam, g, |g|_inf, cold ADMM state.
template <typename TV, typename TIO, int N>
__device__ __forceinline__ int w_setup(SmemW<TV, N>& s, const DevCfg& cfg, const WrTabs& tabs, const FastIn<TIO>& in, size_t b, int tid,
                                       bool first) {
  constexpr int NT = WG<N>::NT, NL = WG<N>::NL, NQ = WG<N>::NQ, n = WG<N>::n, NW = WG<N>::NW, DP = WG<N>::DP;
  if (first) {
    if (tid < 6) { s.wP[tid] = (TV)cfg.w[tid]; s.wQ[tid] = (TV)cfg.w[6 + tid]; }
    if (tid == 0) {
      s.delta = (TV)cfg.delta; s.theta = (TV)cfg.theta; s.inv_m = (TV)cfg.inv_m;
      s.fmin = (TV)cfg.fmin; s.fmax = (TV)cfg.fmax;
    }
    { const double z = opaque_zero_f64(); for (int i = tid; i < DP; i += NT) { s.bv[i] = z; s.cv[i] = z; s.piv[i] = z; s.piv[DP + i] = z; } }   // pad slots stay finite
  }
  int bad = w_load<TV, TIO, N>(s, in, b, tid);
  if constexpr (NW == 1) {
    bad = __any(bad) ? 1 : 0;
  } else {
    bad = __syncthreads_or(bad);
  }
  if (bad) return 1;
  wsync<NW>();
  if (tid == 0) {
    const TV yaw = s.x0[2];  // src/mpc.py:64
    const TV c = cos(yaw), sn = sin(yaw);
    s.cy = c; s.sy = sn;
    s.rzw0[0] = c * s.x0[6] - sn * s.x0[7];
    s.rzw0[1] = sn * s.x0[6] + c * s.x0[7];
    s.rzw0[2] = s.x0[8];
  }
  wsync<NW>();
  for (int L = tid; L < NL; L += NT) {   // src/mpc.py:71-78, 98-107; compute_skew column a = r x e_a (src/utils.py:43-56)
    const bool st = s.ct[L] != 0;
    const TV rx = s.rr[3 * L], ry = s.rr[3 * L + 1], rz = s.rr[3 * L + 2];
    const TV c = s.cy, sn = s.sy;
    const TV Ib0 = (TV)cfg.Ib[0], Ib1 = (TV)cfg.Ib[1], Ib2 = (TV)cfg.Ib[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      TV cx, cyv, cz;
      if (a == 0) { cx = 0; cyv = rz; cz = -ry; }
      else if (a == 1) { cx = -rz; cyv = 0; cz = rx; }
      else { cx = ry; cyv = -rx; cz = 0; }
      const TV bx = (c * cx + sn * cyv) * Ib0, by = (-sn * cx + c * cyv) * Ib1, bz = cz * Ib2;   // diag(Ib) Rz' col
      const TV tx = c * bx - sn * by, ty = sn * bx + c * by;                                   // Ihat^-1 col
      const TV m = st ? (TV)1 : (TV)0;
      s.Bl[9 * L + 0 + a] = m * (c * tx - sn * ty);                                            // Rz Ihat^-1 col
      s.Bl[9 * L + 3 + a] = m * (sn * tx + c * ty);
      s.Bl[9 * L + 6 + a] = m * bz;
    }
    s.cm[L] = st ? s.inv_m : (TV)0;
  }
  // free response minus target, stages k = 1..N (closed forms: DESIGN.md section 2)
  TV eP = 0, eQ = 0;
  if (tid < NQ) {
    const int e = tid, j = e / 6, q = e - 6 * j, k = j + 1;
    const TV d = s.delta, th = s.theta, g = s.x0[12], kd = (TV)k * d;
    const TV* xk = s.xd + 13 * k;
    if (q < 3) {
      eP = s.x0[q] + kd * s.rzw0[q] - xk[q];
      const TV wx = xk[6], wy = xk[7], wz = xk[8];
      const TV rd = q == 0 ? s.cy * wx - s.sy * wy : (q == 1 ? s.sy * wx + s.cy * wy : wz);
      eQ = s.rzw0[q] - rd;
    } else {
      const int a = q - 3;
      eP = s.x0[3 + a] + kd * s.x0[9 + a] - xk[3 + a];
      eQ = s.x0[9 + a] - xk[9 + a];
      if (a == 2) { eP += d * d * g * ((TV)(k * (k - 1)) * (TV)0.5 + th * (TV)k); eQ += kd * g; }
    }
  }
  wsync<NW>();   // everyone has read the staged inputs (rr, xd): their bytes may be reused
  if (tid < NQ) { s.e0P[tid] = eP; s.e0Q[tid] = eQ; }
  wsync<NW>();
  for (int e = tid; e < NQ; e += NT) {   // gam_jq = 2 sum_{k>j} [wP d^2 (k-1-j+th) eP_kq + wQ d eQ_kq]
    const int j = e / 6, q = e - 6 * j;
    const TV d = s.delta, th = s.theta;
    TV aP = 0, aQ = 0;
#pragma unroll
    for (int k = 1; k <= N; ++k) {
      const TV on = k > j ? (TV)1 : (TV)0;
      aP = fma(on * ((TV)(k - 1 - j) + th), s.e0P[6 * (k - 1) + q], aP);
      aQ = fma(on, s.e0Q[6 * (k - 1) + q], aQ);
    }
    s.gam[e] = (TV)2 * (s.wP[q] * d * d * aP + s.wQ[q] * d * aQ);
  }
  wsync<NW>();   // the free response has been consumed: its bytes become the ADMM iterate
  {
    const TV z = opaque_zero<TV>();
    for (int i = tid; i < n; i += NT) { s.uv[i] = z; s.ua[i] = z; s.pu[i] = z; }
    for (int i = tid; i < NL * 5; i += NT) { s.za[i] = z; s.ya[i] = z; s.py[i] = z; }
  }
  wsync<NW>();
  float q[1] = {0.f};
  if (tid < NL) {   // linear term g = T' gam (the gradient at u = 0)
    const TV* gj = s.gam + 6 * (tid >> 2);
    const TV cm = s.cm[tid];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      TV g = cm * gj[3 + a];
#pragma unroll
      for (int i = 0; i < 3; ++i) g = fma(s.Bl[9 * tid + 3 * i + a], gj[i], g);
      s.gl[3 * tid + a] = g;
      q[0] = fmaxf(q[0], fabsf((float)g));
    }
  }
  wmax<1, NW>(q, s.red, tid);
  if (tid == 0) { const int z = opaque(0); s.gmax = q[0]; s.rho = (float)cfg.rho; s.iters = z; s.psteps = z; s.hard = z; s.warm = z; }   // (an opaque zero: see opaque_zero_f64)
  // (s.alpha: set by the kernel before the setup -- the regulariser the solve STARTS with, see the continuation in the kernel)
  wsync<NW>();
#if defined(MPCQP_STAMPS) || defined(MPCQP_WDBG)
  if (b == 0) {   // diagnostic build only: the first QP's setup products
    for (int i = tid; i < n; i += NT) g_wdbg[i] = (double)s.gl[i];
    for (int i = tid; i < NQ; i += NT) g_wdbg[200 + i] = (double)s.gam[i];
    for (int i = tid; i < NL * 9; i += NT) g_wdbg[500 + i] = (double)s.Bl[i];
    for (int i = tid; i < NL; i += NT) { g_wdbg[900 + i] = (double)s.cm[i]; g_wdbg[950 + i] = (double)s.ct[i]; }
    if (tid == 0) { g_wdbg[1200] = (double)s.mu; g_wdbg[1201] = (double)s.gmax; g_wdbg[1202] = (double)s.cy; g_wdbg[1203] = (double)s.sy; }
  }
#endif
  return 0;
}

// Warm start (MPCQP_FLAG_WARM_START; the reference seeds every solve with its previous solution, src/mpc.py:270-271):
// the guess u0 (and the engine's record y0 of the previous solve's multipliers,
// both moved up one stage with MPCQP_FLAG_WARM_SHIFT) becomes the start of the active-set iteration and of the ADMM block.
// s.warm: 0 no guess (cold) | 1 primal guess only (unit multipliers on the rows it holds with equality) | 2 (u0, y0) is nearly
// a KKT point | 3 (u0, y0) is only a neighbour.
template <typename TV, typename TIO, int N>
__device__ __forceinline__ void w_warm_start(SmemW<TV, N>& s, const WrTabs& tabs, const TIO* __restrict__ u0, const float* __restrict__ y0,
                                             const int shift, const int tid) {
  constexpr int n = WG<N>::n, NT = WG<N>::NT, NL = WG<N>::NL, NW = WG<N>::NW;
  float amax[2] = {0.f, 0.f};
  for (int i = tid; i < n; i += NT) {
    const int k = min(i / 12 + shift, N - 1);
    TV v = (TV)u0[k * 12 + i % 12];
    if (!isfinite(v) || s.ct[i / 3] == 0) v = 0;          // swing feet carry no force (src/mpc.py:138-149)
    s.uv[i] = v;
    amax[0] = fmaxf(amax[0], fabsf((float)v));
  }
  for (int i = tid; i < NL * 5; i += NT) {                // the previous solve's multipliers, if the engine has them
    float y = 0.f;
    if (y0) {
      const int L = i / 5, k = min(L / 4 + shift, N - 1);
      y = y0[(k * 4 + L % 4) * 5 + i % 5];
      if (!isfinite(y) || s.ct[L] == 0) y = 0.f;
    }
    s.ya[i] = y;
    amax[1] = fmaxf(amax[1], fabsf(y));
  }
  wmax<2, NW>(amax, s.red, tid);
  wsync<NW>();
  if (!(amax[0] > 0.f)) {                                 // uniform: no guess
    const TV z = opaque_zero<TV>();
    for (int i = tid; i < NL * 5; i += NT) s.ya[i] = z;
    for (int i = tid; i < n; i += NT) s.uv[i] = z;
    wsync<NW>();
    return;
  }
  const bool duals = amax[1] > 0.f;
  TV g3[3];
  w_grad<TV, N>(s, tabs.K, tid, g3);                      // H u0 + g
  float rs[1] = {0.f};
  if (tid < NL) {
    const int L = tid;
    const bool stance = s.ct[L] != 0;
    const TV mu = s.mu, flo = s.fmin, fhi = s.fmax;
    const TV fx = s.uv[3 * L], fy = s.uv[3 * L + 1], fz = s.uv[3 * L + 2];
    s.pu[3 * L] = fx; s.pu[3 * L + 1] = fy; s.pu[3 * L + 2] = fz;
    s.ua[3 * L] = fx; s.ua[3 * L + 1] = fy; s.ua[3 * L + 2] = fz;
    const TV g[5] = {fz, fx - mu * fz, fx + mu * fz, fy - mu * fz, fy + mu * fz};
    const TV tb = (TV)1e-3 * fmax(fabs(fz), (TV)1), tf = (TV)1e-3 * fmax(mu * fabs(fz), (TV)1);
    TV y[5] = {0, 0, 0, 0, 0};
    float z[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (stance) {
      if (duals) {
#pragma unroll
        for (int i = 0; i < 5; ++i) y[i] = (TV)s.ya[5 * L + i];
      } else {   // rows that u0 holds with equality get a unit multiplier of the right sign: the first polish step works on u0's own active set
        y[0] = fz >= fhi - tb ? (TV)1 : (fz <= flo + tb ? (TV)-1 : (TV)0);
        y[1] = g[1] >= -tf ? (TV)1 : (TV)0;  y[2] = g[2] <= tf ? (TV)-1 : (TV)0;
        y[3] = g[3] >= -tf ? (TV)1 : (TV)0;  y[4] = g[4] <= tf ? (TV)-1 : (TV)0;
      }
      z[0] = (float)(fz < flo ? flo : (fz > fhi ? fhi : fz));
      z[1] = (float)(g[1] > 0 ? (TV)0 : g[1]);  z[2] = (float)(g[2] < 0 ? (TV)0 : g[2]);
      z[3] = (float)(g[3] > 0 ? (TV)0 : g[3]);  z[4] = (float)(g[4] < 0 ? (TV)0 : g[4]);
      if (duals) {   // how good is (u0, y0)?  stationarity residual |H u0 + g + G'y0|
        const float m = (float)mu;
        const float rx = (float)g3[0] + (float)(y[1] + y[2]), ry = (float)g3[1] + (float)(y[3] + y[4]);
        const float rz = (float)g3[2] + (float)y[0] + m * (float)(-y[1] + y[2] - y[3] + y[4]);
        rs[0] = fmaxf(fmaxf(fabsf(rx), fabsf(ry)), fabsf(rz));
      }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) { s.py[5 * L + i] = y[i]; s.za[5 * L + i] = (TV)z[i]; if (!duals) s.ya[5 * L + i] = opaque_zero<TV>(); }
  }
  wmax<1, NW>(rs, s.red, tid);
  if (tid == 0) s.warm = !duals ? 1 : (rs[0] <= WARM_KKT_TOL * fmaxf(s.gmax, 1.f) ? 2 : 3);
  wsync<NW>();
}

// ----------------------------------------------------------------------------------------------------- ADMM block
// OSQP algorithm 1 on the rows  fz | fx - mu fz | fx + mu fz | fy - mu fz | fy + mu fz  of a leg-stage
// (src/mpc.py:138-173), one lane per leg-stage, scaled duals yh = y / rho.
template <typename TM>
struct LegAdmm {
  TM u[3], z[5], yh[5], g[3];
  TM lo0, hi0, loA, hiB;     // fz box; friction rows: A rows in [loA, 0], B rows in [0, hiB]
  TM mu;
};

// OSQP's rho-adaptation ratio sqrt((|r_prim| / norm_prim) / (|r_dual| / norm_dual)) of the ADMM iterate (u, z, y): one
// structured gradient for H u + g, the rest per leg.  Uniform result.
template <typename TV, int N>
__device__ __forceinline__ float w_ratio(SmemW<TV, N>& s, const WrTabs& tabs, const TV (&u)[3], const TV (&z)[5], const TV (&y)[5],
                                         const TV (&g)[3], const TV mu, const bool leg, const int tid) {
  constexpr int NL = WG<N>::NL, NW = WG<N>::NW;
  if (tid < NL) {
#pragma unroll
    for (int a = 0; a < 3; ++a) s.uv[3 * tid + a] = u[a];
  }
  wsync<NW>();
  TV hv[3];
  w_grad<TV, N>(s, tabs.K, tid, hv);
  float q[4] = {0.f, 0.f, 0.f, 0.f};
  if (leg) {   // (differences in TV: the ADMM-only termination test of an fp64 run looks below fp32 resolution)
    const TV m = mu * u[2];
    const TV gu[5] = {u[2], u[0] - m, u[0] + m, u[1] - m, u[1] + m};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      q[0] = fmaxf(q[0], fabsf((float)(gu[i] - z[i])));
      q[2] = fmaxf(q[2], fmaxf(fabsf((float)gu[i]), fabsf((float)z[i])));
    }
    const TV Gy[3] = {y[1] + y[2], y[3] + y[4], y[0] + mu * (-y[1] + y[2] - y[3] + y[4])};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      q[1] = fmaxf(q[1], fabsf((float)(hv[a] + Gy[a])));
      q[3] = fmaxf(q[3], fmaxf(fabsf((float)(hv[a] - g[a])), fabsf((float)Gy[a])));
    }
  }
  wmax<4, NW>(q, s.red, tid);
  const float sp = q[2], sd = fmaxf(q[3], s.gmax);
  if (tid == 0) { s.resid[0] = q[0]; s.resid[1] = q[1]; s.resid[2] = sp; s.resid[3] = sd; }   // |r_prim|, |r_dual| and their norms (ADMM-only termination)
  return sqrtf((q[0] / fmaxf(sp, 1e-12f)) / fmaxf(q[1] / fmaxf(sd, 1e-12f), 1e-30f));
}

// The same for the iterate the last ADMM block left in LDS (only needed when its polish steps failed).
template <typename TV, int N>
__device__ __forceinline__ float w_ratio_lds(SmemW<TV, N>& s, const WrTabs& tabs, const int tid0) {
  constexpr int NL = WG<N>::NL, NW = WG<N>::NW;
  const int tid = fresh_tid<NW>(tid0), L = min(tid, NL - 1);
  TV u[3], z[5], y[5], g[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { u[a] = s.ua[3 * L + a]; g[a] = s.gl[3 * L + a]; }
#pragma unroll
  for (int i = 0; i < 5; ++i) { z[i] = s.za[5 * L + i]; y[i] = s.ya[5 * L + i]; }
  return w_ratio<TV, N>(s, tabs, u, z, y, g, s.mu, tid < NL, tid);
}

constexpr double ALPHA_EASY = 1e-2;     // regulariser at which the active-set search is done (continuation start)
constexpr double ALPHA_FLOOR = 3e-6;    // where a request for alpha = 0 ends (tools/alpha0_floor.py: 1e-5 leaves the net wrench 2e-4 off, 1e-6 stalls)
#ifndef MPCQP_W_POLISH_PATIENCE
#define MPCQP_W_POLISH_PATIENCE 1
#endif
constexpr int POLISH_CHEAP_LEGS = 3;    // ... on at most this many changed leg-stages (a changed leg-stage costs up to six rank-one updates, ~3.7 us)
constexpr int POLISH_CHEAP_STEPS = 3;   // further steps of a round beyond the patience rule while they only update the inverse on few leg-stages
                                        // (seven batches of other seeds than the bench's, tools/patience_sweep.py, profiles/r03_cheap_sweep.txt:
                                        //  3 steps on <= 3 leg-stages: +7.9 % on their mean at B = 4096; <= 5 leg-stages: -3 %; without the
                                        //  leg limit -- any step that updates, up to 8 leg-stages -- no gain: a changed leg-stage costs up to six
                                        //  rank-one updates, and a futile step on five of them costs as much as the rebuild it avoids)
constexpr int POLISH_PATIENCE = MPCQP_W_POLISH_PATIENCE;   // polish steps that may fail to halve the KKT violation before the round gives up
#ifndef MPCQP_W_ADAPT_AT
#define MPCQP_W_ADAPT_AT 25
#endif

// ----------------------------------------------------------------------------------------------------- Anderson acceleration
// The ADMM block exists to find the active set, and on the QPs that end a launch (two-legged support at low friction) plain
// ADMM needs 300 - 400 iterations for it: the iteration is a contraction with a factor close to 1 along a few directions.
// Anderson acceleration (type II, memory AA_M) of the map  v -> f^p(v),  v = z + y / rho  the pre-projection variable of
// OSQP's iteration (z = clip(v), y / rho = v - z: the five rows of a leg-stage, five numbers per lane):  every p-th iterate is
// replaced by the combination of the last AA_M + 1 of them that minimises the fixed-point residual in the least-squares sense,
//     gam = argmin | r - dF gam |,   v+ = f(v) - dX gam,     dF / dX: differences of consecutive residuals / images
// -- nine inner products over the wave (seven DPP steps each), a regularised 3 x 3 solve in uniform registers, fifteen FMAs per
// lane, once per p iterations.  numpy study on the condensed QP (tools/accel_study.py): the hardest QPs of five batches reach a
// polishable iterate in half the iterations (worst case of a batch 375 -> 250 us of solve), the easy ones are unchanged.
// Only with the polish (MPCQP_FLAG_POLISH): an ADMM-only run is OSQP's algorithm 1 unchanged.  History in fp32 (it steers an
// extrapolation, it is not part of the answer); base point and images in the iteration's element type.
#ifndef MPCQP_AA_M
#define MPCQP_AA_M 3
#endif
constexpr int AA_M = MPCQP_AA_M;
struct LegAA {
  float rp[5];                       // previous residual f(v) - v
  float dX[AA_M][5], dF[AA_M][5];    // column AA_M - 1 is the newest
};

__device__ __forceinline__ void w_aa_reset(LegAA& h) {
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    h.rp[k] = 0.f;
#pragma unroll
    for (int j = 0; j < AA_M; ++j) { h.dX[j][k] = 0.f; h.dF[j][k] = 0.f; }
  }
}

// One extrapolation: fx = f^p(xb) has just been computed.  Files (fx, fx - xb) in the history and returns the next base point in
// xb (the extrapolated iterate, or fx itself while the history is empty / when the least-squares problem is degenerate -- then the
// history restarts).  `have_prev`: an earlier image exists (uniform).  Uniform control flow; ends with the caller's state untouched
// except xb / fp / h.
template <typename TM, int NW>
__device__ __forceinline__ void w_aa_step(LegAA& h, TM (&xb)[5], TM (&fp)[5], const TM (&fx)[5], bool& have_prev, const bool leg,
                                          float* __restrict__ red, const int tid) {
  static_assert(AA_M == 3 || AA_M == 2, "the solve below is written for two or three columns");
  constexpr int M = AA_M, NQ_ = M * (M + 1) / 2 + M;
  float r[5];
  if (!have_prev) {   // (uniform) the first image of a history: nothing to combine yet -- file it and go on from it
#pragma unroll
    for (int k = 0; k < 5; ++k) { h.rp[k] = leg ? (float)(fx[k] - xb[k]) : 0.f; fp[k] = fx[k]; xb[k] = fx[k]; }
    have_prev = true;
    return;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    r[k] = leg ? (float)(fx[k] - xb[k]) : 0.f;
    const float dx = have_prev ? (float)(fx[k] - fp[k]) : 0.f, df = have_prev ? r[k] - h.rp[k] : 0.f;
#pragma unroll
    for (int j = 0; j + 1 < M; ++j) { h.dX[j][k] = h.dX[j + 1][k]; h.dF[j][k] = h.dF[j + 1][k]; }
    h.dX[M - 1][k] = dx; h.dF[M - 1][k] = df;
    fp[k] = fx[k]; h.rp[k] = r[k];
  }
  have_prev = true;
  float q[NQ_];   // M = 3: 00 01 02 11 12 22 | 0r 1r 2r;  M = 2: 00 01 11 | 0r 1r
#pragma unroll
  for (int i = 0; i < NQ_; ++i) q[i] = 0.f;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int at = 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
#pragma unroll
      for (int j = i; j < M; ++j) { q[at] = fmaf(h.dF[i][k], h.dF[j][k], q[at]); ++at; }
    }
#pragma unroll
    for (int i = 0; i < M; ++i) { q[at] = fmaf(h.dF[i][k], r[k], q[at]); ++at; }
  }
  wsum<NQ_, NW>(q, red, tid);
  float g[M];
  bool ok;
  if constexpr (M == 3) {
    const float tr = q[0] + q[3] + q[5];
    const bool have = tr > 0.f;
    const float reg = 1e-6f * tr + 1e-30f;
    // regularised normal equations by L D L' (uniform values)
    const float a00 = q[0] + reg, a11 = q[3] + reg, a22 = q[5] + reg, a01 = q[1], a02 = q[2], a12 = q[4];
    const float i0 = w_rcp(a00), l10 = a01 * i0, l20 = a02 * i0;
    const float d1 = fmaf(-l10, a01, a11), i1 = w_rcp(d1), t21 = fmaf(-l20, a01, a12), l21 = t21 * i1;
    const float d2 = fmaf(-l21, t21, fmaf(-l20, a02, a22)), i2 = w_rcp(d2);
    const float y0 = q[6], y1 = fmaf(-l10, y0, q[7]), y2 = fmaf(-l21, y1, fmaf(-l20, y0, q[8]));
    g[2] = y2 * i2; g[1] = fmaf(-l21, g[2], y1 * i1); g[0] = fmaf(-l20, g[2], fmaf(-l10, g[1], y0 * i0));
    ok = have && d1 > 0.f && d2 > 0.f && fabsf(g[0]) + fabsf(g[1]) + fabsf(g[2]) <= 1e4f;   // (a NaN fails the comparison)
    if (!have) have_prev = true; else if (!ok) { w_aa_reset(h); have_prev = false; }   // degenerate history: start again from this iterate
  } else {
    const float tr = q[0] + q[2];
    const bool have = tr > 0.f;
    const float reg = 1e-6f * tr + 1e-30f;
    const float a00 = q[0] + reg, a11 = q[2] + reg, a01 = q[1];
    const float i0 = w_rcp(a00), l10 = a01 * i0, d1 = fmaf(-l10, a01, a11), i1 = w_rcp(d1);
    const float y0 = q[3], y1 = fmaf(-l10, y0, q[4]);
    g[1] = y1 * i1; g[0] = fmaf(-l10, g[1], y0 * i0);
    ok = have && d1 > 0.f && fabsf(g[0]) + fabsf(g[1]) <= 1e4f;
    if (have && !ok) { w_aa_reset(h); have_prev = false; }
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float c = 0.f;
#pragma unroll
    for (int j = 0; j < M; ++j) c = fmaf(g[j], h.dX[j][k], c);
    xb[k] = ok ? fx[k] - (TM)c : fx[k];
  }
}

// The leg's 6 x 3 wrench map [B_l ; contact / m I] and the inverse diagonal of D = 2 alpha + sigma + rho G'G.
template <typename TV, typename TM, int N>
__device__ __forceinline__ void w_admm_sys(const SmemW<TV, N>& s, const DevCfg& cfg, int L, float rho, LegSys<TM>& Ls) {
  const bool stance = s.ct[L] != 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) Ls.A[c][i] = (TM)s.Bl[9 * L + 3 * i + c];
#pragma unroll
    for (int a = 0; a < 3; ++a) Ls.A[c][3 + a] = a == c ? (TM)s.cm[L] : (TM)0;
  }
  TM sigma = (TM)cfg.sigma;
  if constexpr (sizeof(TM) == 4) sigma = ufloat(sigma);   // (a converted configuration constant is hoisted out of the QP loop: keep it scalar)
  const TM r = (TM)rho, m = (TM)s.mu, a2 = (TM)((TV)2 * s.alpha);
  Ls.dinv[0] = Ls.dinv[1] = stance ? (TM)1 / (a2 + sigma + (TM)2 * r) : (TM)0;
  Ls.dinv[2] = stance ? (TM)1 / (a2 + sigma + r * fma((TM)4 * m, m, (TM)1)) : (TM)0;   // (an fma: the 1 is an inline operand, not half of a hoisted register pair)
}

// One ADMM block from the state in s.ua / s.za / s.ya with penalty s.rho.  `adapt`: run the single early rho check (round 0
// only); a QP that triggers it gets rho <- rho * ratio, a rebuilt matrix and a longer block.  Updates s.rho / s.iters /
// s.hard / s.ratio and leaves the new iterate in s.ua/za/ya and s.pu/py.
// Register phases that share nothing but LDS (see w_polish): E | tile + sweep | iterations.
template <typename TV, typename TM, int N, bool REFINE = false>
__device__ __forceinline__ void w_admm(SmemW<TV, N>& s, const DevCfg& cfg, const WrTabs& tabs, const TM* __restrict__ kinvT, const int adapt,
                                       const int kfirst, const int tid0) {
  constexpr int NL = WG<N>::NL, NW = WG<N>::NW, G = WG<N>::G;
  // Element type of the sweep that inverts S.  Horizon 20: fp64 even when the iterations run on an fp32 tile -- the fp32 sweep of
  // the 120 x 120 system leaves the ADMM iterate ~5e-4 off (10 x the horizon-10 figure) and the active set of 0.2 - 0.8 % of the
  // low-friction QPs never settles; rounding the fp64 inverse to fp32 costs 13 % and leaves 1 - 3 of 4096 (tools/adapt_sweep.py).
  using TS = std::conditional_t<(N > 10), double, TM>;
  TM* const E = reinterpret_cast<TM*>(s.E);
  TM* const piv = reinterpret_cast<TM*>(s.piv);
  TM* const bv = reinterpret_cast<TM*>(s.bv);
  TM* const cv = reinterpret_cast<TM*>(s.cv);
  float rho = s.rho;
  int K = kfirst > 0 ? min(kfirst, cfg.check_every) : cfg.check_every;
  K = max(1, min(K, cfg.max_iter - __builtin_amdgcn_readfirstlane(s.iters)));   // (the iteration cap is exact)
  // Segments of iterations: the early rho check (if any) sits after the first ADAPT_AT iterations, and the acceleration's history starts
  // afresh with every segment -- every cfg.accel_restart iterations where it runs
  const int seg_len = (!REFINE && sizeof(TM) == 4 && cfg.accel_p > 0 && cfg.accel_restart > 0) ? cfg.accel_restart : (1 << 30);
  const bool check = adapt && cfg.early_check;
  // (a cold solve's first block is split at ADAPT_AT with or without the check: the history's one fresh start there is worth 2-3 %)
  int it = 0, seg_end = min(K, (adapt && MPCQP_W_ADAPT_AT < K) ? MPCQP_W_ADAPT_AT : seg_len);
  int hard = 0;
  float ratio = 0.f;
  STAMP_INIT
  for (;;) {
    {   // ---- phase A: E = sum_legs A diag(dinv) A'
      const int tid = fresh_tid<NW>(tid0), L = min(tid, NL - 1);
      LegSys<TS> Ls;
      w_admm_sys<TV, TS, N>(s, cfg, L, rho, Ls);
      w_build_E<TS, N>(Ls, reinterpret_cast<TS*>(s.E), tid);
    }
    STAMP(1);
    WTile<TM> tile;
    {   // ---- phase B: S = K^-1 + E, swept in place
      const int tid = fresh_tid<NW>(tid0), gr = tid / G, gc = tid % G;
      if constexpr (sizeof(TS) == sizeof(TM)) {
        w_tile_init<TM, N>(tile, kinvT, E, gr, gc, tid);
        STAMP(2);
        w_sweep<TM, N>(tile, piv, gr, gc);
      } else {   // swept in fp64, rounded to the fp32 tile the iterations use
        WTile<TS> t64;
        w_tile_init<TS, N>(t64, tabs.kinv64, reinterpret_cast<const TS*>(s.E), gr, gc, tid);
        STAMP(2);
        w_sweep<TS, N>(t64, reinterpret_cast<TS*>(s.piv), gr, gc);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) tile.v[i][j] = mk2((float)t64.v[i][2 * j], (float)t64.v[i][2 * j + 1]);
        }
      }
    }
    STAMP(3);
    {   // ---- phase C: iterations it .. seg_end from the state in LDS, state back to LDS
      const int tid = fresh_tid<NW>(tid0), L = min(tid, NL - 1), gr = tid / G, gc = tid % G;
      const bool leg = tid < NL, stance = s.ct[L] != 0;
      const TM sigma = (TM)cfg.sigma, relax = (TM)cfg.relax, om = (TM)1 - relax, BIG = (TM)1e30, r = (TM)rho;
      if (tid >= WG<N>::NQ && tid < WG<N>::DP) bv[tid] = (TM)0;   // pad slots of the mat-vec input, in THIS phase's element type
      LegSys<TM> Ls;                                               // (the fp64 polish overlays the same bytes)
      w_admm_sys<TV, TM, N>(s, cfg, L, rho, Ls);
      LegAdmm<TM> A;
      A.mu = (TM)s.mu;
#pragma unroll
      for (int a = 0; a < 3; ++a) { A.u[a] = (TM)s.ua[3 * L + a]; A.g[a] = (TM)s.gl[3 * L + a]; }
#pragma unroll
      for (int i = 0; i < 5; ++i) { A.z[i] = (TM)s.za[5 * L + i]; A.yh[i] = (TM)(s.ya[5 * L + i] / (TV)rho); }
      A.lo0 = stance ? (TM)s.fmin : (TM)0; A.hi0 = stance ? (TM)s.fmax : (TM)0;
      A.loA = stance ? -BIG : (TM)0; A.hiB = stance ? BIG : (TM)0;
      bool rebuild = false;
      // Anderson acceleration (with the polish only): history per segment of iterations with the same matrix
      // (fp32 iterations only: next to an fp64 iteration tile the history spills.  Horizon 10: the history lives in registers;
      //  horizon 20: in LDS between extrapolations -- next to that tile 45 more live registers are 90 more spilled ones)
      constexpr bool AA_ON = !REFINE && sizeof(TM) == 4, AA_LDS = N > 10;
      const int aa_p = AA_ON ? __builtin_amdgcn_readfirstlane(cfg.accel_p) : 0;
      LegAA aa;
      TM aa_xb[5], aa_fp[5];
      auto aa_park = [&]() {      // registers -> LDS (leg lanes)
        if constexpr (AA_LDS) {
          if (leg) {
            float* h = s.aah + L;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
              h[NL * k] = (float)aa_xb[k]; h[NL * (5 + k)] = (float)aa_fp[k]; h[NL * (10 + k)] = aa.rp[k];
#pragma unroll
              for (int j = 0; j < AA_M; ++j) { h[NL * (15 + 5 * j + k)] = aa.dX[j][k]; h[NL * (15 + 5 * AA_M + 5 * j + k)] = aa.dF[j][k]; }
            }
          }
        }
      };
      auto aa_fetch = [&]() {     // LDS -> registers (every lane reads its clamped leg-stage's slots)
        if constexpr (AA_LDS) {
          const float* h = s.aah + L;
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            aa_xb[k] = (TM)h[NL * k]; aa_fp[k] = (TM)h[NL * (5 + k)]; aa.rp[k] = h[NL * (10 + k)];
#pragma unroll
            for (int j = 0; j < AA_M; ++j) { aa.dX[j][k] = h[NL * (15 + 5 * j + k)]; aa.dF[j][k] = h[NL * (15 + 5 * AA_M + 5 * j + k)]; }
          }
        }
      };
      static_assert(15 + 10 * AA_M <= 45, "slots of the parked history");
      for (;;) {   // segments of iterations with the same matrix; the early rho check sits between the first two
        const int n_it = __builtin_amdgcn_readfirstlane(seg_end - it);
        bool aa_have = false;
        int aa_left = aa_p;
        if (aa_p > 0) {
          w_aa_reset(aa);
#pragma unroll
          for (int k = 0; k < 5; ++k) { aa_xb[k] = A.z[k] + A.yh[k]; aa_fp[k] = aa_xb[k]; }
          aa_park();
        }
        for (int i = 0; i < n_it; ++i) {
          // rhs = sigma u - g + rho G'(z - yh)
          TM v[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) v[k] = A.z[k] - A.yh[k];
          const TM w0 = v[1] + v[2], w1 = v[3] + v[4], w2 = fma(A.mu, (v[2] - v[1]) + (v[4] - v[3]), v[0]);
          const TM rhs[3] = {fma(r, w0, fma(sigma, A.u[0], -A.g[0])), fma(r, w1, fma(sigma, A.u[1], -A.g[1])),
                             fma(r, w2, fma(sigma, A.u[2], -A.g[2]))};
          TM ut[3];
          w_solve<TM, N>(tile, Ls, rhs, ut, bv, cv, tid, gr, gc);
          if constexpr (REFINE) {
            // (own instantiation of the kernel, REFINE = true: the extra live values cost the all-fp64 kernel 85 more spilled registers)
            // ADMM that has to converge by itself to tight tolerances (polish off, eps below ~1e-6): one step of iterative refinement
            // on  M u~ = rhs,  M = (2 alpha + sigma) I + rho G'G + T'KT,  with the structured product -- the explicit swept inverse is
            // exact to ~1e-10, which left the dual residual of one QP in 32 stalled a decade above eps = 1e-9 (round-2 advisor)
            {
              if (leg) {
#pragma unroll
                for (int a = 0; a < 3; ++a) s.uv[3 * L + a] = (TV)ut[a];
              }
              wsync<NW>();
              TV hv[3];
              w_grad<TV, N>(s, tabs.K, tid, hv);   // H u~ + g
              const TM dg[3] = {(TM)2, (TM)2, (TM)1 + (TM)4 * A.mu * A.mu};
              TM rr[3], du[3];
#pragma unroll
              for (int a = 0; a < 3; ++a) rr[a] = rhs[a] - (((TM)hv[a] - A.g[a]) + (sigma + r * dg[a]) * ut[a]);
              w_solve<TM, N>(tile, Ls, rr, du, bv, cv, tid, gr, gc);
#pragma unroll
              for (int a = 0; a < 3; ++a) ut[a] += du[a];
            }
          }
          const TM mz = A.mu * ut[2];
          const TM gt[5] = {ut[2], ut[0] - mz, ut[0] + mz, ut[1] - mz, ut[1] + mz};
#pragma unroll
          for (int a = 0; a < 3; ++a) A.u[a] = fma(relax, ut[a], om * A.u[a]);
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const TM lo = k == 0 ? A.lo0 : ((k & 1) ? A.loA : (TM)0), hi = k == 0 ? A.hi0 : ((k & 1) ? (TM)0 : A.hiB);
            const TM t = fma(relax, gt[k], om * A.z[k]) + A.yh[k];
            const TM zn = fmin(fmax(t, lo), hi);
            A.yh[k] = t - zn;
            A.z[k] = zn;
          }
          if (aa_p > 0 && --aa_left == 0) {   // uniform
            aa_left = aa_p;
            if (i + 1 + aa_p <= n_it) {       // (the segment ends with at least a period of genuine ADMM iterations: the polish -- or the
                                              //  next history -- starts from a settled iterate; one plain iteration after an extrapolation is not one:
                                              //  warm-started blocks of 36 = 25 + 11 iterations were slower than blocks of 24, profiles/r03f_rollout_warm.txt)
              TM fx[5];
#pragma unroll
              for (int k = 0; k < 5; ++k) fx[k] = A.z[k] + A.yh[k];
              aa_fetch();
              w_aa_step<TM, NW>(aa, aa_xb, aa_fp, fx, aa_have, leg, s.aared, tid);
              aa_park();
#pragma unroll
              for (int k = 0; k < 5; ++k) {
                const TM lo = k == 0 ? A.lo0 : ((k & 1) ? A.loA : (TM)0), hi = k == 0 ? A.hi0 : ((k & 1) ? (TM)0 : A.hiB);
                const TM zn = fmin(fmax(aa_xb[k], lo), hi);
                A.yh[k] = aa_xb[k] - zn;
                A.z[k] = zn;
              }
            }
          }
        }
        it = seg_end;
        STAMP(4);
        if (it >= K) break;
        if (check && !hard && it == MPCQP_W_ADAPT_AT) {   // the single early rho check: OSQP's residual ratio after the first ADAPT_AT iterations
          const TV u3[3] = {(TV)A.u[0], (TV)A.u[1], (TV)A.u[2]}, g3[3] = {(TV)A.g[0], (TV)A.g[1], (TV)A.g[2]};
          TV z5[5], y5[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) { z5[k] = (TV)A.z[k]; y5[k] = (TV)rho * (TV)A.yh[k]; }
          ratio = w_ratio<TV, N>(s, tabs, u3, z5, y5, g3, (TV)A.mu, leg, tid);
          STAMP(5);
          if (ratio > cfg.adapt_thr) { rebuild = true; break; }      // uniform: slowly converging QP -> larger penalty, rebuilt matrix
        }
        seg_end = min(K, it + seg_len);
      }
      if (leg) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { s.ua[3 * L + a] = (TV)A.u[a]; s.pu[3 * L + a] = (TV)A.u[a]; }
#pragma unroll
        for (int k = 0; k < 5; ++k) { const TV y = (TV)rho * (TV)A.yh[k]; s.za[5 * L + k] = (TV)A.z[k]; s.ya[5 * L + k] = y; s.py[5 * L + k] = y; }
      }
      wsync<NW>();
      if (!rebuild) break;
    }
    rho = fminf(rho * ratio, ADAPT_RHO_MAX);   // ... and a longer block
    hard = 1;
    K = max(K, min((cfg.hard_x10 * K) / 10, cfg.max_iter - __builtin_amdgcn_readfirstlane(s.iters)));
    seg_end = min(K, it + seg_len);
  }
  if (fresh_tid<NW>(tid0) == 0) { s.rho = rho; s.iters += K; s.hard |= hard; }
  wsync<NW>();
}

// ----------------------------------------------------------------------------------------------------- polish step
// One primal-dual active-set step from (s.pu, s.py) (OSQP's `polish`, specialised to the 5 rows of a leg-stage): fz at a
// bound and/or fx, fy tied to +-mu fz per leg; the equality-constrained QP is solved in the free variables through the
// wrench-space system in TP precision; duals from stationarity; accepted only on a KKT check.  Returns 1 when accepted
// (answer in s.uv), else 0 with (s.pu, s.py) replaced by the candidate.
// The active set of a leg-stage as the polish uses it: zs / xs / ys in {-1, 0, +1} (fz at fmin / free / at fmax; fx, fy tied
// to -mu fz / free / tied to +mu fz), packed as (zs + 1) | (xs + 1) << 2 | (ys + 1) << 4.
struct ActSet {
  int zs, xs, ys;
  bool ez, ex, ey;
  __device__ __forceinline__ ActSet(int code, bool stance) {
    zs = (code & 3) - 1; xs = ((code >> 2) & 3) - 1; ys = ((code >> 4) & 3) - 1;
    ez = stance && zs == 0; ex = stance && xs == 0; ey = stance && ys == 0;
  }
};

// The leg's 6 x 3 reduced wrench map (tied tangential components ride on fz) and inverse diagonal 1 / (2 alpha Z'Z).
template <typename TV, typename TP, int N>
__device__ __forceinline__ void w_polish_sys(const SmemW<TV, N>& s, int L, const ActSet& a, LegSys<TP>& Ls) {
  const TV muv = s.mu, txs = (TV)a.xs * muv, tys = (TV)a.ys * muv;
  TV B[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) B[i] = s.Bl[9 * L + i];
  const TV cm = s.cm[L];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Ls.A[0][i] = a.ex ? (TP)B[3 * i] : (TP)0;
    Ls.A[1][i] = a.ey ? (TP)B[3 * i + 1] : (TP)0;
    Ls.A[2][i] = a.ez ? (TP)(B[3 * i + 2] + txs * B[3 * i] + tys * B[3 * i + 1]) : (TP)0;
  }
  Ls.A[0][3] = a.ex ? (TP)cm : (TP)0; Ls.A[0][4] = 0; Ls.A[0][5] = 0;
  Ls.A[1][3] = 0; Ls.A[1][4] = a.ey ? (TP)cm : (TP)0; Ls.A[1][5] = 0;
  Ls.A[2][3] = a.ez ? (TP)(txs * cm) : (TP)0; Ls.A[2][4] = a.ez ? (TP)(tys * cm) : (TP)0; Ls.A[2][5] = a.ez ? (TP)cm : (TP)0;
  const TV a2 = (TV)2 * s.alpha;
  Ls.dinv[0] = a.ex ? (TP)((TV)1 / a2) : (TP)0;
  Ls.dinv[1] = a.ey ? (TP)((TV)1 / a2) : (TP)0;
  Ls.dinv[2] = a.ez ? (TP)((TV)1 / (a2 * ((TV)1 + muv * muv * (TV)((a.xs != 0) + (a.ys != 0))))) : (TP)0;
}

// The active-set rule of the polish on (s.pu, s.py) for one leg-stage: rows 0 fz | 1 fx - mu fz <= 0 | 2 fx + mu fz >= 0 | 3,4 the
// same for fy -> ActSet code.
template <typename TV, int N>
__device__ __forceinline__ int w_polish_rule(const SmemW<TV, N>& s, const int L, const bool stance) {
  const TV muv = s.mu, fminv = s.fmin, fmaxv = s.fmax;
  int zs = 0, xs = 0, ys = 0;
  if (stance) {
    const TV u0 = s.pu[3 * L], u1 = s.pu[3 * L + 1], u2 = s.pu[3 * L + 2];
    const TV y0 = s.py[5 * L], y1 = s.py[5 * L + 1], y2 = s.py[5 * L + 2], y3 = s.py[5 * L + 3], y4 = s.py[5 * L + 4];
    const TV g1 = u0 - muv * u2, g2 = u0 + muv * u2, g3_ = u1 - muv * u2, g4 = u1 + muv * u2;
    if (y0 + (u2 - fmaxv) > 0) zs = 1;
    else if (y0 + (u2 - fminv) < 0) zs = -1;
    const bool hx = y1 + g1 > 0, lx = y2 + g2 < 0;
    if (hx && lx) xs = (g1 > -g2) ? 1 : -1; else if (hx) xs = 1; else if (lx) xs = -1;
    const bool hy = y3 + g3_ > 0, ly = y4 + g4 < 0;
    if (hy && ly) ys = (g3_ > -g4) ? 1 : -1; else if (hy) ys = 1; else if (ly) ys = -1;
  }
  return (zs + 1) | ((xs + 1) << 2) | ((ys + 1) << 4);
}

// Hash of the active set in s.aset (order-independent sum of per-leg-stage terms, so the result does not depend on arrival order).
// A primal-dual active-set iteration that returns to a set it has already tried repeats itself from there on: the round ends.
// Needs s.aset visible; uniform result; ends with a sync.
template <typename TV, int N>
__device__ __forceinline__ unsigned w_aset_hash(SmemW<TV, N>& s, const int tid) {
  constexpr int NL = WG<N>::NL, NW = WG<N>::NW;
  if (tid == 0) s.ahash = 0u;
  wsync<NW>();
  if (tid < NL) atomicAdd(&s.ahash, ((unsigned)s.aset[tid] + 1u) * (2654435761u * (unsigned)(2 * tid + 1)));
  wsync<NW>();
  const unsigned h = (unsigned)__builtin_amdgcn_readfirstlane((int)s.ahash);
  wsync<NW>();
  return h;
}

#ifndef MPCQP_W_INCR_LEGS
#define MPCQP_W_INCR_LEGS 8     // leg-stages whose active set may change for the inverse to be updated instead of rebuilt
#endif
#ifndef MPCQP_W_INCR_STEPS
#define MPCQP_W_INCR_STEPS 12   // updates in a row before a rebuild (caps chosen on batches of other seeds: 2/4, 4/4, 6/8, 8/12
                                //  give 0.54 / 0.51 / 0.51 / 0.50 ms at B = 4096, identical step counts and solved sets)
#endif

// The polish steps of one round: primal-dual active-set steps from (s.pu, s.py) (OSQP's `polish`, specialised to the 5 rows of a
// leg-stage): fz at a bound and/or fx, fy tied to +-mu fz per leg; the equality-constrained QP is solved in the free variables
// through the wrench-space system in TP precision; duals from stationarity; a candidate is accepted only on a KKT check.
// Returns 1 when a step was accepted (answer in s.uv), else 0 with (s.pu, s.py) the last candidate.  Up to `budget` steps, while
// they make progress (patience rule below) unless `last`.
// Register phases that share nothing but LDS and the tile: rule | E | tile + sweep | solve + KKT.  The fp64 tile is half of the
// register file: it is defined at the top of the outer loop (a full build for the active set in s.aset) and only modified in the
// inner loop, where a step that FOLLOWS another updates -S^-1 instead of rebuilding it: between consecutive steps the active set
// changes on 1.8 leg-stages on average (<= 2 in 82 % of the steps, <= 5 in 97 %), and a changed leg-stage changes S = K^-1 + E by
// at most three rank-one terms removed and three added -- Sherman-Morrison, one mat-vec and one rank-one tile update per term
// (about two pivots' work) against the 6 N pivots of a rebuild.  At most MPCQP_W_INCR_LEGS changed leg-stages
// and MPCQP_W_INCR_STEPS updates in a row; a candidate from a drifted inverse would simply fail the KKT test.
template <typename TV, typename TP, int N>
__device__ __forceinline__ int w_polish_round(SmemW<TV, N>& s, const WrTabs& tabs, const TP* __restrict__ kinvT, const int tid0,
                                              const int budget, const bool last, const int trace_tag, const int incr_legs,
                                              const int patience, const int cheap_steps, const int cheap_legs, const int last_patience) {
  constexpr int NL = WG<N>::NL, NW = WG<N>::NW, G = WG<N>::G;
  constexpr int STG = 2 * 21 + 1;   // staging record of a changed leg-stage in s.E: removed | added {A[3][6], weight[3]}, stage index
  static_assert(STG * MPCQP_W_INCR_LEGS <= N * 36, "the staging records share the bytes of E");
  TP* const E = reinterpret_cast<TP*>(s.E);
  TP* const piv = reinterpret_cast<TP*>(s.piv);
  TP* const bv = reinterpret_cast<TP*>(s.bv);
  TP* const cv = reinterpret_cast<TP*>(s.cv);
  STAMP_INIT
  {   // ---- the first step's active set
    const int tid = fresh_tid<NW>(tid0), L = min(tid, NL - 1);
    const int code = w_polish_rule<TV, N>(s, L, s.ct[L] != 0);
    if (tid < NL) s.aset[L] = (uint8_t)code;
    wsync<NW>();
    const unsigned h0 = w_aset_hash<TV, N>(s, tid);
    if (tid == 0) s.ahist[0] = h0;
  }
  int ok = 0, ps = 0, nstall = 0, cheap_used = 0;
  float vprev = INFINITY, vprev2 = INFINITY;
  bool done = false;
  while (!done) {
    {   // ---- E = T D^-1 T' for the active set in s.aset
      const int tid = fresh_tid<NW>(tid0), L = min(tid, NL - 1);
      LegSys<TP> Ls;
      w_polish_sys<TV, TP, N>(s, L, ActSet(s.aset[L], s.ct[L] != 0), Ls);
      w_build_E<TP, N>(Ls, E, tid);
    }
    STAMP(9);
    WTile<TP> tile;
    {   // ---- S = K^-1 + E, swept in place
      const int tid = fresh_tid<NW>(tid0), gr = tid / G, gc = tid % G;
#ifdef MPCQP_SYM_SWEEP   // (experiment: the lower block triangle only, see w_sym_build)
      if constexpr (N == 10 && NW == 1 && sizeof(TP) == 8) {
        w_sym_build(tile, kinvT, E, piv, tid);
        STAMP(10);
      } else
#endif
      {
        w_tile_init<TP, N>(tile, kinvT, E, gr, gc, tid);
        STAMP(10);
        w_sweep<TP, N>(tile, piv, gr, gc);
      }
    }
    STAMP(11);
    int in_row = 0;
    for (;;) {
      int step_ok_i;
      {
      // ---- phase C: solve in the free variables from the projection of pu, duals, KKT
      const int tid = fresh_tid<NW>(tid0), L = min(tid, NL - 1), gr = tid / G, gc = tid % G;
      const bool leg = tid < NL, stance = s.ct[L] != 0;
      if (tid >= WG<N>::NQ && tid < WG<N>::DP) bv[tid] = opaque_zero<TP>();   // pad slots of the mat-vec input, in this phase's element type
      const ActSet as(s.aset[L], stance);
      const int zs = as.zs, xs = as.xs, ys = as.ys;
      const bool ez = as.ez, ex = as.ex, ey = as.ey;
      // (only what the refinement loop needs is read here: the loop carries the fp64 tile, and every extra live value is a spill)
      TV txs, tys;
      TV v3[3];   // the reduced variables; where a component is fixed (bound / tied / swing) v3 holds its fixed value instead
      {
        const TV muv = s.mu;
        txs = (TV)xs * muv; tys = (TV)ys * muv;
        const TV F = zs > 0 ? (TV)s.fmax : (TV)s.fmin;
        v3[0] = ex ? s.pu[3 * L] : (TV)0; v3[1] = ey ? s.pu[3 * L + 1] : (TV)0;
        v3[2] = ez ? s.pu[3 * L + 2] : ((stance && zs != 0) ? F : (TV)0);
      }
      TV uc[3], gr3[3] = {0, 0, 0};
      auto expand = [&]() {
        uc[2] = v3[2];
        uc[0] = ex ? v3[0] : txs * v3[2];
        uc[1] = ey ? v3[1] : tys * v3[2];
      };
      float stat = INFINITY, prev = INFINITY;
      for (int rf = 0;; ++rf) {
        expand();
        if (leg) {
    #pragma unroll
          for (int c = 0; c < 3; ++c) s.uv[3 * L + c] = uc[c];
        }
        wsync<NW>();
        w_grad<TV, N>(s, tabs.K, tid, gr3);
        TV rg[3] = {ex ? gr3[0] : (TV)0, ey ? gr3[1] : (TV)0, ez ? gr3[2] + txs * gr3[0] + tys * gr3[1] : (TV)0};
        float q[2] = {leg ? fmaxf(fmaxf(fabsf((float)rg[0]), fabsf((float)rg[1])), fabsf((float)rg[2])) : 0.f,
                      leg ? fmaxf(fmaxf(fabsf((float)uc[0]), fabsf((float)uc[1])), fabsf((float)uc[2])) : 0.f};
        if (!isfinite(q[0])) q[0] = INFINITY;
        wmax<2, NW>(q, s.red, tid);
        prev = stat; stat = ufloat(q[0]);
        // refine until the stationarity residual is safely below what the acceptance test will ask for (it scales with 2 alpha:
        // binding for alpha < 1e-2, where one fp64 solve -- residual ~1e-9 |g| -- is not enough)
        const float gmaxl = s.gmax;
        const float tol_stat = (sizeof(TV) == 8) ? (1e-6f + 1e-9f * gmaxl) : (3e-7f * fmaxf(gmaxl, 1.f));
        const float tol = fminf(tol_stat, 0.25f * ((sizeof(TV) == 8) ? 2.f * (float)s.alpha : 1e30f) * 2e-5f * fmaxf(1.f, q[1]));
        if (stat <= tol || rf >= 4 || (rf > 0 && !(stat < 0.5f * prev))) break;   // converged / stagnated (uniform)
        const TP rhs[3] = {(TP)(-rg[0]), (TP)(-rg[1]), (TP)(-rg[2])};
        TP dx[3];
        {
          LegSys<TP> Ls;
          w_polish_sys<TV, TP, N>(s, L, as, Ls);
          w_solve<TP, N>(tile, Ls, rhs, dx, bv, cv, tid, gr, gc);
        }
        v3[0] += ex ? (TV)dx[0] : (TV)0; v3[1] += ey ? (TV)dx[1] : (TV)0; v3[2] += ez ? (TV)dx[2] : (TV)0;
      }
      // duals from stationarity grad_leg + G_A' y_A = 0, then primal feasibility + dual sign
      const TV muv = s.mu, fminv = s.fmin, fmaxv = s.fmax;
      const float gmaxf = s.gmax;
      const float acc_stat = (sizeof(TV) == 8) ? (1e-5f + 1e-8f * gmaxf) : (1e-5f * fmaxf(gmaxf, 1.f));
      const float ftol = (sizeof(TV) == 8) ? 1e-7f : 2e-5f;
      const float dtol = (sizeof(TV) == 8) ? (1e-5f + 1e-9f * gmaxf) : (2e-5f * fmaxf(1.f, gmaxf));
      TV yn[5] = {0, 0, 0, 0, 0};
      float viol[3] = {0.f, 0.f, 0.f};
      if (leg && stance) {
        TV zacc = gr3[2];
        if (xs > 0) { yn[1] = -gr3[0]; zacc += muv * (-yn[1]); }
        else if (xs < 0) { yn[2] = -gr3[0]; zacc += muv * yn[2]; }
        if (ys > 0) { yn[3] = -gr3[1]; zacc += muv * (-yn[3]); }
        else if (ys < 0) { yn[4] = -gr3[1]; zacc += muv * yn[4]; }
        if (zs != 0) yn[0] = -zacc;
        const TV g0 = uc[2], g1 = uc[0] - muv * uc[2], g2 = uc[0] + muv * uc[2], g3_ = uc[1] - muv * uc[2], g4 = uc[1] + muv * uc[2];
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
      wmax<3, NW>(viol, s.red, tid);
      // a stationarity / dual-sign slack e moves the forces by ~e / (2 alpha): scale the acceptance with the curvature so that
      // `solved` implies the 1e-4 band for any alpha
      const float a2f = (sizeof(TV) == 8) ? 2.f * (float)s.alpha : 1e30f, uscale = fmaxf(1.f, viol[2]);
      const bool step_ok = viol[0] <= ftol * uscale && viol[1] <= fminf(dtol, a2f * 1e-5f * uscale) && stat <= fminf(acc_stat, a2f * 2e-5f * uscale);
      STAMP(6);
      if (leg) {   // publish the candidate as the next polish iterate / the answer (s.uv already holds it)
    #pragma unroll
        for (int c = 0; c < 3; ++c) s.pu[3 * L + c] = uc[c];
    #pragma unroll
        for (int i = 0; i < 5; ++i) s.py[5 * L + i] = yn[i];
      }
      if (tid == 0) { s.kkt[0] = stat; s.kkt[1] = viol[0]; s.kkt[2] = viol[1]; s.psteps += 1; }
      wsync<NW>();
        step_ok_i = step_ok ? 1 : 0;
      }
      STAMP(7);
      ok = __builtin_amdgcn_readfirstlane(step_ok_i);
      const float v = ufloat(s.kkt[1] + s.kkt[2] / fmaxf(s.gmax, 1.f) * 100.f);
#if defined(MPCQP_STAMPS)   // (diagnostic build, single-QP launches: a trace of the polish steps, tools/hardest.py)
      if (trace_tag >= 0 && fresh_tid<NW>(tid0) == 0 && s.psteps <= 80) {
        double* rec = g_wdbg + 1300 + 8 * (s.psteps - 1);
        rec[0] = trace_tag; rec[1] = ps; rec[2] = s.kkt[0]; rec[3] = s.kkt[1]; rec[4] = s.kkt[2]; rec[5] = s.rho; rec[6] = s.iters; rec[7] = ok + 10 * in_row;
      }
#endif
      ++ps;
      if (ok || ps >= budget) { done = true; break; }
      {
        // Active-set steps while they make progress: a step that does not at least halve the KKT violation of the previous one
        // (primal + dual-sign, each relative to its scale) means ADMM has not settled the active set yet -- back to ADMM rather
        // than through the rest of the budget (each rebuilt step costs an fp64 sweep, about 50 ADMM iterations) -- except that the
        // candidates of a converging sequence often ALTERNATE between a primal-feasible one with a wrong multiplier sign and a
        // dual-feasible one with a small constraint violation (the hardest QP of the bench batch lost two ADMM rounds to being cut
        // one step short, tools/hardest.py): such a one-sided candidate is compared with the one two steps before it, like with
        // like, for up to two extra steps.
        const int psd = ps - 1;
        const bool one_sided = fminf(s.kkt[1], s.kkt[2]) <= 1e-9f;
        const bool stalled = !(v < 0.5f * vprev);
        const bool alternating = one_sided && psd >= 2 && psd < 4 && v < 0.5f * vprev2;
        if (psd >= 1 && stalled && !alternating) ++nstall;
      }
      // (the decision is taken below, once it is known whether the next step would be a cheap one)
      vprev2 = vprev; vprev = v;
      // ---- the next step's active set; what changed is staged for an update of the inverse
      int nupd = 0;
      bool incr = false;
      {
        const int tid = fresh_tid<NW>(tid0), L = min(tid, NL - 1);
        const bool stance = s.ct[L] != 0;
        const int code = w_polish_rule<TV, N>(s, L, stance);
        {
          const bool chg = tid < NL && stance && s.aset[L] != (uint8_t)code;
          int slot = 0;
          if constexpr (NW == 1) {
            const unsigned long long mask = __ballot(chg);
            nupd = __builtin_popcountll(mask);
            slot = __builtin_popcountll(mask & ((1ull << tid) - 1ull));
          } else {   // four waves: a bit mask in LDS; slots in leg order, so that the result does not depend on arrival order
            constexpr int NM = (NL + 31) / 32;
            if (tid < NM) s.chgmask[tid] = 0u;
            wsync<NW>();
            if (chg) atomicOr(&s.chgmask[L >> 5], 1u << (L & 31));
            wsync<NW>();
#pragma unroll
            for (int w = 0; w < NM; ++w) {
              const unsigned m = s.chgmask[w];
              nupd += __builtin_popcount(m);
              slot += w < (L >> 5) ? __builtin_popcount(m) : (w == (L >> 5) ? __builtin_popcount(m & ((1u << (L & 31)) - 1u)) : 0);
            }
          }
          incr = in_row < MPCQP_W_INCR_STEPS && nupd <= incr_legs;
          if (incr && chg) {
            TP* rec = E + STG * slot;
            LegSys<TP> Lo;
            w_polish_sys<TV, TP, N>(s, L, ActSet(s.aset[L], stance), Lo);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
              for (int q = 0; q < 6; ++q) rec[6 * c + q] = Lo.A[c][q];
              rec[18 + c] = -Lo.dinv[c];
            }
            w_polish_sys<TV, TP, N>(s, L, ActSet(code, stance), Lo);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
              for (int q = 0; q < 6; ++q) rec[21 + 6 * c + q] = Lo.A[c][q];
              rec[21 + 18 + c] = Lo.dinv[c];
            }
            rec[42] = (TP)(L >> 2);
          }
        }
        if (tid < NL) s.aset[L] = (uint8_t)code;
        wsync<NW>();
        {   // an active set this round has already tried: the iteration would repeat itself from here on
          const unsigned h = w_aset_hash<TV, N>(s, tid);
          bool seen = false;
          for (int k = 0; k < min(ps, 32); ++k) seen = seen || s.ahist[k] == h;
          if (tid == 0 && ps < 32) s.ahist[ps] = h;
          wsync<NW>();
          // (not in a round that nothing follows: at small regularisers the solve is only as exact as its refinement, and a second
          //  visit of an active set starts that refinement from a better point -- the alpha = 0 continuation lost 1 % of its QPs to this test)
          if (seen && !last) { done = true; break; }   // uniform
        }
        // ... and a round whose steps have stopped making progress ends -- unless the next step is a cheap one (its active set
        // differs on few leg-stages, so -S^-1 is updated, not rebuilt: ~4 us against the ~85 us of another ADMM round; the hardest
        // QP of the bench batch repeated the same two candidates in three rounds, one step short of its optimum each time,
        // tools/hardest.py), for up to `cheap_steps` such steps per round.
        if (nstall >= patience && !last) {
          if (incr && nupd <= cheap_legs && cheap_used < cheap_steps) ++cheap_used;
          else { done = true; break; }   // uniform
        }
        if (last && last_patience > 0 && nstall >= last_patience) { done = true; break; }   // uniform
      }
      if (done) break;
      if (!incr) break;   // rebuild for the new active set (outer loop)
      {   // ---- -S^-1 updated term by term:  S' = S + w a a'  =>  -S'^-1 = -S^-1 + w / (1 + w a'y) y y',  y = S^-1 a
        const int tid = fresh_tid<NW>(tid0), gr = tid / G, gc = tid % G;
        for (int k = 0; k < nupd; ++k) {
          const TP* rec = E + STG * k;
          const int j = (int)rec[42];
          for (int h = 0; h < 6; ++h) {
            const TP* a = rec + 21 * (h / 3) + 6 * (h % 3);
            const TP wgt = rec[21 * (h / 3) + 18 + h % 3];
            if (wgt == (TP)0) continue;          // uniform
            if (tid < WG<N>::DP) bv[tid] = (tid >= 6 * j && tid < 6 * j + 6) ? a[tid - 6 * j] : (TP)0;
            wsync<NW>();
            w_matvec<TP, N>(tile, bv, cv, gr, gc);
            wsync<NW>();
            TP gam = (TP)0;
#pragma unroll
            for (int q = 0; q < 6; ++q) gam = fma(a[q], cv[6 * j + q], gam);
            const TP beta = wgt / ((TP)1 + wgt * gam);
            TP yr[8], yc[8];
            ld8<TP>(cv + 8 * gr, yr);
            ld8<TP>(cv + 8 * gc, yc);
#pragma unroll
            for (int i = 0; i < 8; ++i) yr[i] *= -beta;
            rank1(tile, yr, yc);
            tpin(tile);
            wsync<NW>();
          }
        }
        ++in_row;
      }
      STAMP(12);
    }
  }
  return ok;
}

// ----------------------------------------------------------------------------------------------------- output
template <typename TV, typename TIO, int N>
__device__ __forceinline__ void w_output(SmemW<TV, N>& s, const WrTabs& tabs, TIO* __restrict__ ug, TIO* __restrict__ Xg, int* __restrict__ statusg,
                                         int* __restrict__ itersg, float* __restrict__ resg, float* __restrict__ y_state, const size_t b,
                                         const int ok, const int tid) {
  constexpr int NT = WG<N>::NT, NL = WG<N>::NL, n = WG<N>::n, NW = WG<N>::NW;
  STAMP_INIT
  for (int L = tid; L < NL; L += NT) {
    const bool stance = s.ct[L] != 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      TV v = ok == 1 ? s.uv[3 * L + c] : (TV)s.ua[3 * L + c];   // ADMM termination / iteration cap: the last ADMM iterate
      if (!stance) v = 0;
      s.uv[3 * L + c] = v;
    }
  }
  wsync<NW>();
  for (int i = tid; i < n; i += NT) ug[b * n + i] = (TIO)s.uv[i];   // src/mpc.py:267-268
  if (y_state) {
    for (int i = tid; i < NL * 5; i += NT) y_state[b * (NL * 5) + i] = (float)(ok == 1 ? s.py[i] : s.ya[i]);
  }
  if (Xg) {                                                          // src/mpc.py:265-266: roll the model forward (closed forms)
    TV g3[3];
    w_grad<TV, N>(s, tabs.K, tid, g3);                               // leaves the stage wrenches in s.ww
    const TV d = s.delta, th = s.theta, g = s.x0[12];
    for (int i = tid; i < (N + 1) * 13; i += NT) {
      const int k = i / 13, c = i % 13;
      TV v;
      if (c == 12 || k == 0) v = s.x0[c];
      else if (c < 6) {
        const int q = c;
        TV acc = 0;
        for (int j = 0; j < k; ++j) acc = fma((TV)(k - 1 - j) + th, s.ww[6 * j + q], acc);
        v = c < 3 ? s.x0[c] + (TV)k * d * s.rzw0[c] + d * d * acc : s.x0[c] + (TV)k * d * s.x0[6 + c] + d * d * acc;
        if (c == 5) v += d * d * g * ((TV)(k * (k - 1)) * (TV)0.5 + th * (TV)k);
      } else if (c < 9) {
        TV ax = 0, ay = 0, az = 0;
        for (int j = 0; j < k; ++j) { ax += s.ww[6 * j]; ay += s.ww[6 * j + 1]; az += s.ww[6 * j + 2]; }
        const TV rot = c == 6 ? s.cy * ax + s.sy * ay : (c == 7 ? -s.sy * ax + s.cy * ay : az);   // omega = Rz'(Rz omega)
        v = s.x0[c] + d * rot;
      } else {
        TV acc = 0;
        for (int j = 0; j < k; ++j) acc += s.ww[6 * j + (c - 6)];
        v = s.x0[c] + d * acc;
        if (c == 11) v += (TV)k * d * g;
      }
      Xg[b * (size_t)(N + 1) * 13 + i] = (TIO)v;
    }
  }
  if (tid == 0) {
    statusg[b] = ok == 1 ? MPCQP_STATUS_SOLVED_POLISHED : (ok == 2 ? MPCQP_STATUS_SOLVED_ADMM : MPCQP_STATUS_MAX_ITER);
    itersg[b] = s.iters + 1000 * s.psteps;
    if (resg) { resg[2 * b] = s.kkt[1]; resg[2 * b + 1] = fmaxf(s.kkt[2], s.kkt[0]); }
  }
  STAMP(8);
}

// ----------------------------------------------------------------------------------------------------- the kernel
// Plain form: blockIdx = QP.  Listed form (ob.list != null, ob.head == null): one workgroup per QP, workgroup k takes the k-th QP
// of the dearest-expected-first order that the pre-pass (mpcqp_common.h) files; the hardware's dispatcher places the workgroups.
// Queued form (ob.head != null): only as many workgroups as the device holds, each pulling QPs from the head of that order until
// it is empty (no workgroup turnover: the form for batches many times the device).
template <typename TV, typename TM, typename TP, typename TIO, int N, bool REFINE = false>
__global__ void __launch_bounds__(WG<N>::NT, 2)
mpcqp_wrench_solve(const DevCfg* __restrict__ cfgp, const WrTabs tabs, const FastIn<TIO> in, TIO* ug, TIO* __restrict__ Xg,
                   int* __restrict__ statusg, int* __restrict__ itersg, float* __restrict__ resg, const OrderBuf ob, const int Btot) {
  constexpr int NT = WG<N>::NT, n = WG<N>::n, G = WG<N>::G, NW = WG<N>::NW;
  __shared__ SmemW<TV, N> s;
  __shared__ int s_next;
  const DevCfg& cfg = *cfgp;
  const TM* kinvM; const TP* kinvP;
  if constexpr (sizeof(TM) == 4) kinvM = tabs.kinv32; else kinvM = tabs.kinv64;
  if constexpr (sizeof(TP) == 4) kinvP = tabs.kinv32; else kinvP = tabs.kinv64;
  for (int guard = 0; guard <= Btot; ++guard) {
    // The lane index is made opaque once per QP: everything derived from it (addresses into the constant tables, role
    // masks, loop coefficients) is invariant across the QPs of a resident wave, and LLVM would hoist all of it out of
    // this loop -- several hundred registers' worth, spilled for the whole kernel.
    const int tid0 = threadIdx.x;
    int tid = fresh_tid<NW>(tid0);
    size_t b = blockIdx.x;
    if (ob.list) {
      int i = blockIdx.x;                    // listed form: workgroup k takes the k-th QP of the dearest-first order
      if (ob.head) {                         // queued form: resident workgroups pull from the head of that order
        if (tid == 0) s_next = atomicAdd(ob.head, 1);
        __syncthreads();
        i = __builtin_amdgcn_readfirstlane(s_next);
        __syncthreads();
      }
      if (i >= Btot) break;                  // uniform
      int cnt[ORDER_BUCKETS];                // all class counts in one scalar load, then the search in registers
#pragma unroll
      for (int k = 0; k < ORDER_BUCKETS; ++k) cnt[k] = ob.cnt[k];
      int cls = 0, at = 0;
      bool found = false;
#pragma unroll
      for (int k = ORDER_BUCKETS - 1; k >= 0; --k) {
        const bool hit = !found && i < cnt[k];
        if (hit) { cls = k; at = i; }
        found = found || hit;
        i -= found ? 0 : cnt[k];
      }
      b = (size_t)__builtin_amdgcn_readfirstlane(ob.list[(size_t)cls * ob.cap + at]);
    }
#if defined(MPCQP_STAMPS) || defined(MPCQP_TIMELINE)
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();   // the 100 MHz constant clock: comparable across CUs and XCDs
#endif
#ifdef MPCQP_LDS_POISON   // diagnostic build: every QP starts from an LDS block full of NaN patterns (finds reads of stale LDS)
    {
      unsigned* raw = reinterpret_cast<unsigned*>(&s);
      for (int i = tid; i < (int)(sizeof(s) / 4); i += NT) raw[i] = 0xFFFFFFFFu;
      wsync<NW>();
    }
#endif
    STAMP_INIT
#ifdef MPCQP_LDS_POISON
    const bool first_qp = true;
#else
    const bool first_qp = guard == 0;
#endif
    // Regulariser continuation: a request below ALPHA_EASY (the reference's own cost has alpha = 0, src/mpc.py:121) is solved at
    // ALPHA_EASY first -- the well-conditioned problem every QP of the bench workload solves -- and then walked down by factors of
    // ten with polish steps from the previous level's optimum and multipliers (the active set barely moves between levels,
    // while a cold active-set search at alpha <= 1e-4 cycles on the nearly flat force-distribution directions).  alpha = 0 ends
    // at ALPHA_FLOOR: objective within 1e-7 relative, states and net wrench within 1e-4 of the alpha = 0 optimum (tools/alpha0_floor.py),
    // forces = (nearly) the minimum-norm member of the non-unique optimal set.
    // (both kept in LDS, not in registers: whatever lives across the fp64 sweep is spilled)
    if (tid == 0) {
      s.alpha_target = (TV)cfg.alpha_target;   // (both decided on the host: a double constant compared here lives in a hoisted, spilled register pair)
      s.alpha = (TV)cfg.alpha_start;
    }
    if (w_setup<TV, TIO, N>(s, cfg, tabs, in, b, tid, first_qp)) {   // non-finite input -> zero outputs, status -1
      for (int i = tid; i < n; i += NT) ug[b * n + i] = (TIO)0;
      if (Xg) for (int i = tid; i < (N + 1) * 13; i += NT) Xg[b * (size_t)(N + 1) * 13 + i] = (TIO)0;
      if (tid == 0) {
        statusg[b] = MPCQP_STATUS_NONFINITE;
        itersg[b] = 0;
        if (resg) { const float z = opaque_zero<float>(); resg[2 * b] = z; resg[2 * b + 1] = z; }
      }
      if (!ob.list || !ob.head) break;
      continue;
    }
    if (in.u_init) w_warm_start<TV, TIO, N>(s, tabs, in.u_init + b * n, in.y_state ? in.y_state + b * (WG<N>::NL * 5) : nullptr, in.shift, tid);
    STAMP(0);
    const int max_iter = cfg.max_iter, polish_max = cfg.polish_max;
    int ok = 0;
    const int warm = __builtin_amdgcn_readfirstlane(s.warm);
    // One loop, three kinds of round, so that the polish and the ADMM block are each inlined exactly once:
    //   WARM   (warm start only) polish steps on the guess's own active set before any ADMM block
    //   ADMM   an ADMM block, then polish steps; on failure OSQP's rho adaptation and another round until max_iter is spent
    //   CONT   (continuation) the regulariser has just been lowered: polish steps from the previous level's optimum
    enum { R_WARM, R_ADMM, R_CONT };
    const int warm_tries = !(cfg.flags & MPCQP_FLAG_POLISH) ? 0 : (warm == 1 ? WARM_POLISH : (warm == 2 ? 1 : 0));
    int kind = warm_tries > 0 ? R_WARM : R_ADMM, round = 0, cont_retry = 0;
    for (;;) {
      int budget = kind == R_WARM ? min(warm_tries, polish_max) : 2 * polish_max;
      const bool admm_only = !(cfg.flags & MPCQP_FLAG_POLISH);
      if (kind == R_ADMM) {
        w_admm<TV, TM, N, REFINE>(s, cfg, tabs, kinvM, round == 0 ? 1 : 0, round == 0 ? (warm >= 2 ? (cfg.first_block > 0 ? min(WARM_K, (WARM_FRAC10 * cfg.first_block) / 10) : WARM_K) : cfg.first_block) : 0, tid0);
        budget = admm_only ? 0 : (__builtin_amdgcn_readfirstlane(s.hard) ? HARD_POLISH_FACTOR : 1) * polish_max;
      }
      // Active-set steps while they make progress: a step that does not at least halve the KKT violation of the previous one
      // (primal + dual-sign, each relative to its scale) means ADMM has not settled the active set yet -- back to ADMM rather
      // than through the rest of the budget (each step costs an fp64 sweep, about 50 ADMM iterations).
      const bool last = kind != R_ADMM || __builtin_amdgcn_readfirstlane(s.iters) >= max_iter;   // (a round that nothing follows keeps its full budget)
#ifdef MPCQP_STAMPS
      const int trace_tag = Btot == 1 ? kind * 100 + round : -1;
#else
      const int trace_tag = -1;
#endif
      if (budget > 0) ok = __builtin_amdgcn_readfirstlane(w_polish_round<TV, TP, N>(s, tabs, kinvP, tid0, budget, last, trace_tag, cfg.incr_legs, cfg.patience, cfg.cheap_steps, cfg.cheap_legs, (kind == R_ADMM) ? cfg.last_patience : 0));
      if (ok == 1 && s.alpha > s.alpha_target) {   // next continuation level, from this optimum and its multipliers
        const int tid = fresh_tid<NW>(tid0);
        for (int i = tid; i < n; i += NT) s.ua[i] = s.uv[i];            // the last accepted answer and its multipliers (the ADMM
        for (int i = tid; i < WG<N>::NL * 5; i += NT) s.ya[i] = s.py[i];   // iterate is not needed any more)
        if (tid == 0) { s.alpha_ok = s.alpha; s.alpha = fmax(s.alpha * (TV)0.1, s.alpha_target); }
        wsync<NW>();
        ok = 0; cont_retry = 0;
        kind = R_CONT;
        continue;
      }
      if (ok) break;
      if (kind == R_CONT) {   // level not reached: back to the last accepted point and a smaller step (geometric bisection), a few times
        if (++cont_retry > 3) { ok = 3; break; }                         // ... then the previous level's answer, status MAX_ITER
        const int tid = fresh_tid<NW>(tid0);
        for (int i = tid; i < n; i += NT) s.pu[i] = s.ua[i];
        for (int i = tid; i < WG<N>::NL * 5; i += NT) s.py[i] = s.ya[i];
        if (tid == 0) s.alpha = sqrt(s.alpha_ok * s.alpha);
        wsync<NW>();
        continue;
      }
      if (kind == R_WARM) { kind = R_ADMM; continue; }
      if (!admm_only && s.iters >= max_iter) break;
      {   // residuals of the last ADMM iterate: OSQP's termination test (ADMM only -- what the reference runs, polish off,
          // src/mpc.py:51-55) and its rho adaptation for the next block
        const float ratio = w_ratio_lds<TV, N>(s, tabs, tid0);
        const int tid = fresh_tid<NW>(tid0);
        wsync<NW>();
        if (admm_only) {
          const float tp = (float)cfg.eps_abs + (float)cfg.eps_rel * s.resid[2], td = (float)cfg.eps_abs + (float)cfg.eps_rel * s.resid[3];
          if (s.resid[0] <= tp && s.resid[1] <= td) ok = 2;
          wsync<NW>();
          if (tid == 0) { s.kkt[0] = s.resid[1]; s.kkt[1] = s.resid[0]; s.kkt[2] = 0.f; }
          if (ok || s.iters >= max_iter) break;
        }
        // (tolerance 2 between ADMM + polish rounds; OSQP's own 5 when ADMM has to converge by itself: frequent changes of the
        //  penalty stall the tail of a long ADMM run)
        const float rtol = admm_only ? 5.f : 2.f;
        if (tid == 0 && isfinite(ratio) && (ratio > rtol || ratio < 1.f / rtol)) s.rho = fminf(fmaxf(s.rho * ratio, 1e-4f), 1e4f);
        wsync<NW>();
      }
      ++round;
    }
    w_output<TV, TIO, N>(s, tabs, ug, Xg, statusg, itersg, resg, in.y_state, b, ok, fresh_tid<NW>(tid0));
#if defined(MPCQP_STAMPS) || defined(MPCQP_TIMELINE)
    if (tid == 0 && b < 65536) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_timeline[3 * b] = tl_t0; g_timeline[3 * b + 1] = __builtin_amdgcn_s_memrealtime();
      g_timeline[3 * b + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    if (!ob.list || !ob.head) break;
  }
}

}  // namespace
