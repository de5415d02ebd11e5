// mpcqp_fast.h -- fast path for the benchmarked configuration: horizon 10, ADMM + active-set polish, fp32 matrix
// tiles with fp64 (MIXED) or fp32 (F32) structured residuals.
//
// One kernel, one QP per workgroup at a time (small batches: blockIdx = QP; batches that oversubscribe the device:
// resident workgroups pulling QPs dearest-expected-first from a queue, see "dispatch order" below), but the phases of
// the solve are separate NOINLINE device functions that talk to each other only through the workgroup's LDS block (a
// file-scope __shared__ object, so the callees use plain ds_* addressing):
//   ph_setup        operator tuple -> per-variable response vectors, linear term g
//   ph_admm         M = H + sigma I + rho G'G -> register tiles -> in-register sweep -> K ADMM iterations (fp32),
//                   with one early OSQP rho check (residuals tracked locally by the iteration) that rebuilds the matrix
//                   for slowly converging QPs
//   ph_polish_step  active set -> reduced matrix tiles -> sweep -> solve refined against the TV gradient -> KKT
//   ph_output       forces / predicted states / status
// Every phase that needs the matrix rebuilds it, so the 6 x 16 register tile is local to a phase and each phase gets
// its own register allocation: the fp64 polish no longer pushes the fp32 ADMM loop into scratch (and vice versa), and a
// QP whose polish fails just runs its next ADMM block + polish in place while the rest of the grid moves on.
//
// Geometry (n = 120 force variables, 40 leg-stages): thread (g, c) of a 160-thread workgroup (192 launched: the
// last 32 mirror group 19 so every wave is full) owns rows 6g..6g+5 (two leg-stages) and columns 15c..15c+14 as a
// 6 x 16 register tile (column 15 is padding and stays zero, so packed fp32 FMAs pair up).  LDS vectors are stored in
// 8 chunks of 15 with a chunk stride of 20 floats: the 128-bit reads of the 8 lanes of a group then fall on disjoint
// banks (a stride of 16 puts them on two: 4-way conflicts, 60 % of all LDS cycles in the first version).
#pragma once
#include "mpcqp_device.h"

// Occupancy target (waves per SIMD) of the fast-path kernel and its phase functions.
#ifndef MPCQP_FAST_WPE
#define MPCQP_FAST_WPE 2
#endif
#define MPCQP_PHASE __device__ __attribute__((noinline))

namespace {

struct FG {
  static constexpr int N = 10, n = 120, NL = 40;
  static constexpr int RT = 6, CT = 8, CW = 15, CWP = 16;
  static constexpr int S = 20;             // chunk stride of LDS vectors (floats)
  static constexpr int VP = CT * S;
  static constexpr int NG = n / RT;        // 20 row groups
  static constexpr int NT = 192;           // threads launched (160 + 32 mirrors)
  static constexpr int NW = 3;
};

template <typename TV>
struct SmemF {
  CfgS<TV> cf;
  TV x0[13];
  TV mu, cy, sy;
  TV rzw0[3];
  TV xd[11 * 13];
  TV rr[120];
  TV tt[360], ttr[360];
  TV cm[120];
  TV wr[90], Xs[132], es[132], adj[90];
  TV uv[120], gv[120], gl[120];
  TV pu[120], py[200];
  float ua[120], za[200], ya[200];          // last ADMM iterate (fallback answer, start of the next block)
  float hva[120];                           // H u + g at that iterate, tracked by the iteration
  float c0[100], c1[100];                   // 2 x the stage-pair tables (H = 2 (c1 P.P' + c0 Q.Q'))
  alignas(16) float pq[120 * 12];
  float dg[120];
  alignas(16) float vbuf[2 * FG::VP];
  alignas(16) float rhs[2 * FG::VP];
  float red[FG::NW * 4];
  float kkt[4];                             // stat, primal violation, dual violation of the last polish step
  float gmax;                               // |g|_inf
  float rho;                                // current ADMM penalty
  float ratio;                              // OSQP residual ratio at the end of the last ADMM block
  int iters, psteps, hard, warm;            // bookkeeping shared by the phases
  uint8_t ct[40];
  uint8_t em[40];
};

// The workgroup's LDS block.  File scope so that the noinline phase functions address it directly (ds_*); one raw
// buffer shared by both instantiations.  The library is built with -amdgpu-lower-module-lds-strategy=module, which
// gives the block the same fixed address in every kernel: with the default (a per-kernel offset table, because the
// phase functions are reachable from several kernels) the callees re-loaded the block's base from memory inside the
// sweep -- one scalar load + wait on the critical path of every pivot.
__shared__ __attribute__((aligned(16))) unsigned char g_lds_raw[sizeof(SmemF<double>)];
static_assert(sizeof(SmemF<double>) >= sizeof(SmemF<float>), "raw LDS block must hold either instantiation");
template <typename TV> __device__ __forceinline__ SmemF<TV>& lds() { return *reinterpret_cast<SmemF<TV>*>(g_lds_raw); }

struct Lane {   // who am I inside the workgroup
  int tid, grp, cc, myleg, row0, rbA, rbB, rbM;
  bool second;
  __device__ __forceinline__ Lane() {
    tid = threadIdx.x;
    grp = min(tid / 8, FG::NG - 1);   // threads 160..191 mirror group 19 (full waves, identical writes)
    cc = tid % 8;
    second = cc >= 4;                 // lanes 0-3 look after leg-stage 2g, lanes 4-7 after 2g+1
    myleg = 2 * grp + (second ? 1 : 0);
    row0 = 3 * myleg;
    rbA = (6 * grp / FG::CW) * FG::S + (6 * grp) % FG::CW;
    rbB = ((6 * grp + 3) / FG::CW) * FG::S + (6 * grp + 3) % FG::CW;
    rbM = second ? rbB : rbA;
  }
};

__device__ __forceinline__ int fpidx(int i) { return (i / FG::CW) * FG::S + i % FG::CW; }

// The register tile is declared as 6 x 8 packed pairs so that the compiler keeps each pair in an aligned register pair
// and the rank-1 updates / mat-vecs map 1:1 onto v_pk_fma_f32 (with a plain float[6][16] it shuffled ~45 moves per pivot).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef f2 Tile[6][8];
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

// Tuple form (src/mpc.py:242-255).  Returns the per-thread "non-finite input" flag.
template <typename TV, typename TIO>
__device__ __forceinline__ int fast_load_tuple(SmemF<TV>& s, const FastIn<TIO>& in, size_t b, int tid) {
  constexpr int N = FG::N, NT = FG::NT;
  int bad = 0;
  for (int i = tid; i < 13; i += NT) { const TV v = (TV)in.x0[b * 13 + i]; s.x0[i] = v; bad |= !isfinite(v); }
  for (int i = tid; i < (N + 1) * 13; i += NT) { const TV v = (TV)in.xdes[b * (N + 1) * 13 + i]; s.xd[i] = v; bad |= !isfinite(v); }
  for (int i = tid; i < N * 12; i += NT) { const TV v = (TV)in.r[b * N * 12 + i]; s.rr[i] = v; bad |= !isfinite(v); }
  for (int i = tid; i < N * 4; i += NT) s.ct[i] = in.contact[b * N * 4 + i] ? 1 : 0;
  if (tid == 0) { const TV m = (TV)in.mu[b]; s.mu = m; bad |= !isfinite(m); }
  return bad;
}

// Stage the constants, build the per-variable response vectors and the linear term g.  Returns the uniform
// "non-finite input" flag.  Ends with a barrier.
template <typename TV>
__device__ __forceinline__ int fast_setup(SmemF<TV>& s, const DevCfg& cfg, const double* __restrict__ ctab, int bad, int tid,
                                          bool first) {
  constexpr int N = FG::N, n = FG::n, NT = FG::NT;
  if (first) {   // configuration constants: staged once per workgroup (the queued form solves many QPs with one)
    for (int i = tid; i < N * N; i += NT) { s.c0[i] = (float)(2.0 * ctab[i]); s.c1[i] = (float)(2.0 * ctab[N * N + i]); }   // the factor 2 of H = 2 Su'WSu rides on the tables
    if (tid == 0) {
      s.cf.delta = (TV)cfg.delta; s.cf.theta = (TV)cfg.theta; s.cf.alpha = (TV)cfg.alpha; s.cf.inv_m = (TV)cfg.inv_m;
      s.cf.fmin = (TV)cfg.fmin; s.cf.fmax = (TV)cfg.fmax;
    }
    if (tid >= 64 && tid < 76) { s.cf.w[tid - 64] = (TV)cfg.w[tid - 64]; s.cf.sw[tid - 64] = (TV)cfg.sw[tid - 64]; }
    if (tid >= 128 && tid < 131) s.cf.Ib[tid - 128] = (TV)cfg.Ib[tid - 128];
  }
  for (int i = tid; i < 2 * FG::VP; i += NT) { s.vbuf[i] = 0.f; s.rhs[i] = 0.f; }   // pad slots must stay finite
  bad = __syncthreads_or(bad);
  if (bad) return 1;
  if (tid == 0) {
    const TV yaw = s.x0[2];  // src/mpc.py:64
    const TV c = cos(yaw), sn = sin(yaw);
    s.cy = c; s.sy = sn;
    s.rzw0[0] = c * s.x0[6] - sn * s.x0[7];
    s.rzw0[1] = sn * s.x0[6] + c * s.x0[7];
    s.rzw0[2] = s.x0[8];
  }
  __syncthreads();
  if (tid < n) {  // src/mpc.py:71-78, 98-107; compute_skew column a = r x e_a (src/utils.py:43-56)
    const int i = tid, j = i / 12, l = (i % 12) / 3, a = i % 3;
    const bool st = s.ct[j * 4 + l] != 0;
    const TV rx = s.rr[(j * 4 + l) * 3 + 0], ry = s.rr[(j * 4 + l) * 3 + 1], rz = s.rr[(j * 4 + l) * 3 + 2];
    TV cx, cyv, cz;
    if (a == 0) { cx = 0; cyv = rz; cz = -ry; }
    else if (a == 1) { cx = -rz; cyv = 0; cz = rx; }
    else { cx = ry; cyv = -rx; cz = 0; }
    const TV c = s.cy, sn = s.sy;
    TV bx = (c * cx + sn * cyv) * s.cf.Ib[0], by = (-sn * cx + c * cyv) * s.cf.Ib[1], bz = cz * s.cf.Ib[2];
    TV tx = c * bx - sn * by, ty = sn * bx + c * by, tz = bz;
    if (!st) { tx = ty = tz = 0; }
    s.tt[i * 3 + 0] = tx; s.tt[i * 3 + 1] = ty; s.tt[i * 3 + 2] = tz;
    s.ttr[i * 3 + 0] = c * tx - sn * ty; s.ttr[i * 3 + 1] = sn * tx + c * ty; s.ttr[i * 3 + 2] = tz;
    s.cm[i] = st ? s.cf.inv_m : (TV)0;
    s.uv[i] = 0;
  }
  __syncthreads();
  struct_grad<SmemF<TV>, TV, N>(s, tid);   // gradient at u = 0 = linear term g
  if (tid < n) s.gl[tid] = s.gv[tid];
  __syncthreads();
  return 0;
}

// Register tile M[6g..6g+5][15c..15c+14] = 2 (c1 P.P' + c0 Q.Q') + diag, from s.pq / s.dg.
// s.pq holds, per variable, the six pairs (P_q, Q_q): one v_pk_fma_f32 then advances both dot products of an entry, the
// (c1, c0) weights are folded into the column's pairs once per column (the six rows of a thread share one stage), and
// the diagonal is added after the loop (its position inside the tile depends only on 6g - 15c).
template <typename TV>
__device__ __forceinline__ void fast_build(Tile& tile, const SmemF<TV>& s, int grp, int cc) {
  constexpr int N = FG::N;
  const int col0 = cc * FG::CW, row0 = 6 * grp, stage = row0 / 12;
#pragma unroll
  for (int h = 0; h < 2; ++h) {   // two passes of three rows: bounds the live registers
    f2 Pr[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float4* pp = reinterpret_cast<const float4*>(s.pq + (row0 + 3 * h + r) * 12);
#pragma unroll
      for (int q4 = 0; q4 < 3; ++q4) { const float4 v = pp[q4]; Pr[r][2 * q4] = mk2(v.x, v.y); Pr[r][2 * q4 + 1] = mk2(v.z, v.w); }
    }
#pragma unroll
    for (int c = 0; c < FG::CW; ++c) {
      if (c % 3 == 0) asm volatile("" ::: "memory");  // at most three columns' loads in flight
      const int ic = col0 + c, jc = ic / 12;
      const float k1 = s.c1[stage * N + jc], k0 = s.c0[stage * N + jc];
      f2 pc[6];
      const float4* pp = reinterpret_cast<const float4*>(s.pq + ic * 12);
#pragma unroll
      for (int q4 = 0; q4 < 3; ++q4) { const float4 v = pp[q4]; pc[2 * q4] = mk2(v.x, v.y); pc[2 * q4 + 1] = mk2(v.z, v.w); }
      f2 acc[3];   // three independent chains, interleaved (back-to-back dependent packed FMAs cost a wait state each)
#pragma unroll
      for (int r = 0; r < 3; ++r) acc[r] = Pr[r][0] * pc[0];
#pragma unroll
      for (int q = 1; q < 6; ++q) {
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[r] = __builtin_elementwise_fma(Pr[r][q], pc[q], acc[r]);
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float v;   // k1 P.P' + k0 Q.Q' (asm: keeps the SLP vectoriser from re-pairing the two halves across entries)
        asm("v_mul_f32 %0, %1, %2\n\tv_fmac_f32 %0, %3, %4" : "=&v"(v) : "v"(k0), "v"(acc[r].y), "v"(k1), "v"(acc[r].x));
        if (c % 2 == 0) tile[3 * h + r][c / 2].x = v; else tile[3 * h + r][c / 2].y = v;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) tile[r][7].y = 0.f;
  // diagonal entries (row r, column d + r) with d = 6g - 15c in {0, 6, 12, -3, 3, 9} when the tile meets the diagonal
  float dgr[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) dgr[r] = s.dg[row0 + r];
  const int d = row0 - col0;
#pragma unroll
  for (int dd = -3; dd <= 12; dd += 3) {
    if (d == dd) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const int c = dd + r;     // compile-time
        if (c >= 0 && c < FG::CW) { if (c % 2 == 0) tile[r][c / 2].x += dgr[r]; else tile[r][c / 2].y += dgr[r]; }
      }
    }
  }
}

// In-register symmetric sweep over the enabled variables: tile <- -M^{-1} (Gauss-Jordan without pivoting, SPD).
// One LDS broadcast of the pivot row and one barrier per pivot; disabled variables (identity rows) are skipped.
template <typename TV>
__device__ __forceinline__ void fast_sweep(Tile& tile, SmemF<TV>& s, int grp, int cc, int rbA, int rbB) {
  constexpr int S = FG::S, VP = FG::VP;
  int step = 0;
  for (int kc2 = 0; kc2 < 4; ++kc2) {
    // enable masks of the ten leg-stages of this pass, fetched together (a dependent LDS read in front of every
    // leg-stage sat on the critical path: no load can be hoisted above the pivots' barriers)
    int emv[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) emv[i] = s.em[10 * kc2 + i];
#pragma unroll
    for (int i = 0; i < 10; ++i) emv[i] = __builtin_amdgcn_readfirstlane(emv[i]);
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int kc = 2 * kc2 + par;
#pragma unroll
      for (int c = 0; c < FG::CW; ++c) {
        const int rr = (c + 3 * par) % 6;        // row inside the owner's tile (compile-time after unrolling)
        const int k = FG::CW * kc + c;           // pivot index
        const int em = emv[5 * par + c / 3];     // uniform: enable mask of the pivot's leg-stage
        if (!((em >> (c % 3)) & 1)) continue;
        const int og = k / 6;
        float* vb = s.vbuf + (step & 1) * VP;
        if (grp == og) {
          float4* w4 = reinterpret_cast<float4*>(vb + cc * S);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) w4[q4] = make_float4(tile[rr][2 * q4].x, tile[rr][2 * q4].y, tile[rr][2 * q4 + 1].x, tile[rr][2 * q4 + 1].y);
        }
        __syncthreads();
        const float p = __builtin_amdgcn_rcpf(vb[kc * S + c]);
        float vr[6];
        f2 vc[8];
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) { vr[r3] = vb[rbA + r3] * p; vr[3 + r3] = vb[rbB + r3] * p; }
        {
          const float4* r4 = reinterpret_cast<const float4*>(vb + cc * S);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) { const float4 v = r4[q4]; vc[2 * q4] = mk2(v.x, v.y); vc[2 * q4 + 1] = mk2(v.z, v.w); }
        }
        // The owner's pivot row becomes p * row.  Its tile row IS the published row (bit for bit), so the same FMA
        // does it with the multiplier 1 - p:  row - (1 - p) row = p row  (one select instead of a masked 8-multiply
        // region; the rounding of 1 - p costs eps |1 - p| / p relative, ~1e-5 only for the rho = 30 ADMM matrices).
        const float vrr = (grp == og) ? 1.f - p : vr[rr];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const f2 m = splat2(r == rr ? -vrr : -vr[r]);
#pragma unroll
          for (int j = 0; j < 8; ++j) tile[r][j] = __builtin_elementwise_fma(m, vc[j], tile[r][j]);
        }
        {   // pivot column (selects, not a branch: a divergent region makes the compiler copy tile registers at the join)
          const bool col = cc == kc;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            const float v = (r == rr && grp == og) ? -p : vr[r];
            if (c % 2 == 0) tile[r][c / 2].x = col ? v : tile[r][c / 2].x; else tile[r][c / 2].y = col ? v : tile[r][c / 2].y;
          }
        }
        ++step;
      }
    }
  }
}

// out[r] = -(row r of MY leg-stage) . x for r = 0..2, summed over the 8 lanes of the group (x in the padded LDS layout).
// Reduce-scatter: a lane only needs its own half's three rows, so the first step trades the other half's partial sums
// with the mirror lane (lane i <-> 7 - i, which sits in the other half and wants exactly those), then a quad butterfly
// finishes the three values: 15 cross-lane ops instead of the 18 + 3 selects of an all-reduce of six.
__device__ __forceinline__ void fast_matvec(const Tile& tile, const float* __restrict__ x, int cc, bool second, float (&out)[3]) {
  f2 acc[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) acc[r] = mk2(0.f, 0.f);
  const float4* r4 = reinterpret_cast<const float4*>(x + cc * FG::S);
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const float4 v = r4[q4];
    const f2 xa = mk2(v.x, v.y), xb = mk2(v.z, v.w);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      acc[r] = __builtin_elementwise_fma(tile[r][2 * q4], xa, acc[r]);
      acc[r] = __builtin_elementwise_fma(tile[r][2 * q4 + 1], xb, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float lo = acc[r].x + acc[r].y, hi = acc[3 + r].x + acc[3 + r].y;
    float t = (second ? hi : lo) + dpp_mov<0x141>(second ? lo : hi);
    t += dpp_mov<0xB1>(t);
    t += dpp_mov<0x4E>(t);
    out[r] = -t;
  }
}

// Writes the 12-vectors / diagonal / enable mask of one leg-stage's three variables (lanes a = 0..2 of the leg's
// four lanes; lane 3 writes the mask).  mode 0: ADMM matrix H + sigma I + rho G'G; mode 1: reduced polish matrix.
template <typename TV>
__device__ __forceinline__ void fast_describe(SmemF<TV>& s, int myleg, int a, bool stance, TV muv, int mode, float rho,
                                              float sigma, int zs, int xs, int ys) {
  const int row0 = 3 * myleg;
  const bool ez = stance && (mode == 0 || zs == 0), ex = stance && (mode == 0 || xs == 0), ey = stance && (mode == 0 || ys == 0);
  if (a < 3) {
    const bool en = a == 0 ? ex : (a == 1 ? ey : ez);
    TV pv[12];
    var_pq<SmemF<TV>, TV>(s, row0 + a, a, pv);
    const TV a2 = (TV)2 * s.cf.alpha;
    TV dgv;
    if (mode == 0) {
      dgv = stance ? a2 + (TV)sigma + (TV)rho * (a == 2 ? (TV)1 + (TV)4 * muv * muv : (TV)2) : (TV)1;
    } else {
      if (a == 2 && ez) {  // tied tangential forces ride on the fz slot
        TV px[12], py_[12];
        var_pq<SmemF<TV>, TV>(s, row0 + 0, 0, px);
        var_pq<SmemF<TV>, TV>(s, row0 + 1, 1, py_);
#pragma unroll
        for (int q = 0; q < 12; ++q) pv[q] += (TV)xs * muv * px[q] + (TV)ys * muv * py_[q];
      }
      dgv = !en ? (TV)1 : (a == 2 ? a2 * ((TV)1 + muv * muv * (TV)((xs != 0) + (ys != 0))) : a2);
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) s.pq[(row0 + a) * 12 + (q < 6 ? 2 * q : 2 * (q - 6) + 1)] = en ? (float)pv[q] : 0.f;   // pairs (P_q, Q_q)
    s.dg[row0 + a] = (float)dgv;
  } else {
    s.em[myleg] = (uint8_t)((ex ? 1 : 0) | (ey ? 2 : 0) | (ez ? 4 : 0));
  }
}

// ------------------------------------------------------------------------------------------------------ ADMM block
// OSQP algorithm 1 on the rows  fz | fx - mu fz | fx + mu fz | fy - mu fz | fy + mu fz  of a leg-stage
// (src/mpc.py:138-173).  The leg-stage's state is spread over its four lanes so that the per-iteration update is one
// short instruction stream instead of the whole leg replicated on every lane (the update, not the mat-vec, was the
// longer half of an iteration):
//   lane q = 0: component fx, rows A = fx - mu fz <= 0 and B = fx + mu fz >= 0
//   lane q = 1: component fy, rows A = fy - mu fz <= 0 and B = fy + mu fz >= 0
//   lane q = 2: component fz, row  A = fmin <= fz <= fmax (row B is an inert dummy);  lane 3 mirrors lane 2, never writes
// The only cross-lane traffic is the mu-coupling of the fz component (two quad broadcasts).
struct LegLane {
  int q, comp, rowA, rowB;              // role, own component (0..2), own rows in the 5-row layout (rowB < 0: none)
  float u, hv, g, w;                    // own component: iterate, H u + g, linear term, G'(rho z - y)
  float zA, yA, zB, yB;                 // own constraint rows
  float mA, mB, aB, kz, loA, hiA, loB, hiB;

  __device__ __forceinline__ LegLane(int cc, bool stance, float mu, float fmin, float fmax) {
    const float BIG = 1e30f;
    q = cc & 3;
    comp = q < 2 ? q : 2;
    rowA = q == 0 ? 1 : (q == 1 ? 3 : 0);
    rowB = q == 0 ? 2 : (q == 1 ? 4 : -1);
    const bool tang = q < 2;
    mA = tang ? -mu : 0.f;  mB = tang ? mu : 0.f;  aB = tang ? 1.f : 0.f;  kz = tang ? 0.f : mu;
    loA = !stance ? 0.f : (tang ? -BIG : fmin);   hiA = !stance ? 0.f : (tang ? 0.f : fmax);   // src/mpc.py:151-173
    loB = 0.f;                                     hiB = (stance && tang) ? BIG : 0.f;
  }
  template <typename SM>
  __device__ __forceinline__ void load(const SM& s, int myleg) {
    const int i = 3 * myleg + comp;
    u = s.ua[i]; hv = s.hva[i]; g = (float)s.gl[i];
    zA = s.za[5 * myleg + rowA]; yA = s.ya[5 * myleg + rowA];
    zB = rowB >= 0 ? s.za[5 * myleg + rowB] : 0.f; yB = rowB >= 0 ? s.ya[5 * myleg + rowB] : 0.f;
    w = 0.f;
  }
  template <typename SM, typename TV>
  __device__ __forceinline__ void publish(SM& s, int myleg) const {
    if (q < 3) {
      const int i = 3 * myleg + comp;
      s.ua[i] = u; s.pu[i] = (TV)u; s.hva[i] = hv;
      s.za[5 * myleg + rowA] = zA; s.ya[5 * myleg + rowA] = yA; s.py[5 * myleg + rowA] = (TV)yA;
      if (rowB >= 0) { s.za[5 * myleg + rowB] = zB; s.ya[5 * myleg + rowB] = yB; s.py[5 * myleg + rowB] = (TV)yB; }
    }
  }
  // G'(rho z - y) of my component from my rows (+ the mu-coupled part of the tangential lanes on the fz lanes)
  __device__ __forceinline__ void update_w(float rho) {
    const float vA = rho * zA - yA, vB = rho * zB - yB;
    const float d = vB - vA;
    const float dx = dpp_mov<0x00>(d), dy = dpp_mov<0x55>(d);   // quad broadcasts of lanes 0 and 1
    w = fmaf(kz, dx + dy, vA + vB);
  }
};

// `iters` iterations; state in registers, right-hand sides double-buffered in LDS, one barrier per iteration.
// tile = -M^{-1}.  hv follows H u + g without applying H: M ut = sigma u - g + w  =>  H ut + g = sigma (u - ut) + w - rho G'G ut.
template <typename TV>
__device__ __forceinline__ void fast_admm_iters(const Tile& tile, SmemF<TV>& s, const int iters, const float rho,
                                                const float sigma, const float relax, const float mu, const bool stance,
                                                LegLane& L, const int cc, const bool second, const int rbM) {
  constexpr int VP = FG::VP;
  const float inv_rho = 1.f / rho, om = 1.f - relax;
  const float dd = L.comp == 2 ? sigma + rho * (1.f + 4.f * mu * mu) : sigma + 2.f * rho;   // diag(sigma I + rho G'G), my component
  const float sten = stance ? 1.f : 0.f;
  L.update_w(rho);
  s.rhs[rbM + L.comp] = fmaf(sigma, L.u, L.w - L.g);   // (lane 3 repeats lane 2's store: no divergent region in the loop)
  __syncthreads();
  int buf = 0;
  const int n_it = __builtin_amdgcn_readfirstlane(iters);   // scalar trip count: plain s_cmp / s_cbranch loop control
  for (int it = 0; it < n_it; ++it) {
    float sum[3];
    fast_matvec(tile, s.rhs + buf * VP, cc, second, sum);
    const float t0 = sum[0], t1 = sum[1], utz = sum[2];
    const float utc = L.q == 0 ? t0 : (L.q == 1 ? t1 : utz);
    const float hq = sten * fmaf(-dd, utc, fmaf(sigma, L.u, L.w));
    L.hv = fmaf(relax, hq, om * L.hv);
    L.u = fmaf(relax, utc, om * L.u);
    {
      const float gA = fmaf(L.mA, utz, utc);
      const float zr = fmaf(relax, gA, om * L.zA);
      const float zn = fminf(fmaxf(fmaf(L.yA, inv_rho, zr), L.loA), L.hiA);
      L.yA = fmaf(rho, zr - zn, L.yA);
      L.zA = zn;
    }
    {
      const float gB = fmaf(L.mB, utz, L.aB * utc);
      const float zr = fmaf(relax, gB, om * L.zB);
      const float zn = fminf(fmaxf(fmaf(L.yB, inv_rho, zr), L.loB), L.hiB);
      L.yB = fmaf(rho, zr - zn, L.yB);
      L.zB = zn;
    }
    L.update_w(rho);
    buf ^= 1;
    s.rhs[buf * VP + rbM + L.comp] = fmaf(sigma, L.u, L.w - L.g);
    __syncthreads();
  }
}

// OSQP's rho-adaptation ratio sqrt((|r_prim| / norm_prim) / (|r_dual| / norm_dual)) from the lane-distributed state:
// hv = H u + g is tracked by the iteration itself, so no gradient evaluation is needed.  Uniform result.
template <typename TV>
__device__ __forceinline__ float fast_local_ratio(SmemF<TV>& s, const LegLane& L, const int tid) {
  const float uz = dpp_mov<0xAA>(L.u);                            // quad broadcast of the fz lane
  const float guA = fmaf(L.mA, uz, L.u), guB = fmaf(L.mB, uz, L.aB * L.u);
  const float e = L.yB - L.yA;
  const float ex = dpp_mov<0x00>(e), ey = dpp_mov<0x55>(e);
  const float Gy = fmaf(L.kz, ex + ey, L.yA + L.yB);              // (G'y) of my component
  float q[4];
  q[0] = fmaxf(fabsf(guA - L.zA), fabsf(guB - L.zB));
  q[2] = fmaxf(fmaxf(fabsf(guA), fabsf(L.zA)), fmaxf(fabsf(guB), fabsf(L.zB)));
  q[1] = fabsf(L.hv + Gy);
  q[3] = fmaxf(fabsf(L.hv - L.g), fabsf(Gy));
  block_max<4, FG::NW>(q, s.red, tid);
  const float sp = q[2], sd = fmaxf(q[3], s.gmax);
  return sqrtf((q[0] / fmaxf(sp, 1e-12f)) / fmaxf(q[1] / fmaxf(sd, 1e-12f), 1e-30f));
}

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
constexpr float WARM_KKT_TOL = 1e-3f;            // (u0, y0) counts as a KKT point when its stationarity residual is below this x |g|

// Warm start (MPCQP_FLAG_WARM_START; the reference seeds every solve with its previous solution, src/mpc.py:270-271,
// primal only and unshifted).  The guess u0 becomes (a) the start of the primal-dual active-set iteration: constraints
// that u0 satisfies with equality (to 1e-3) get a unit multiplier of the right sign, so the first polish step works on
// u0's own active set, and the kernel tries up to WARM_POLISH polish steps BEFORE any ADMM block; (b) the start
// (u, z = clamp(G u), y = 0) of the ADMM block that follows if those steps fail.  An all-zero guess means "none" (first tick): cold start.
template <typename TV, typename TIO>
__device__ __forceinline__ void fast_warm_start(SmemF<TV>& s, const TIO* __restrict__ u0, const float* __restrict__ y0,
                                                const int shift, int tid) {
  constexpr int n = FG::n, NT = FG::NT, N = FG::N;
  float amax[2] = {0.f, 0.f};
  for (int i = tid; i < n; i += NT) {
    const int k = min(i / 12 + shift, N - 1);
    TV v = (TV)u0[k * 12 + i % 12];
    if (!isfinite(v) || s.ct[i / 3] == 0) v = 0;          // swing feet carry no force (src/mpc.py:138-149)
    s.uv[i] = v;
    amax[0] = fmaxf(amax[0], fabsf((float)v));
  }
  for (int i = tid; i < FG::NL * 5; i += NT) {            // the previous solve's multipliers, if the engine has them
    float y = 0.f;
    if (y0) {
      const int L = i / 5, k = min(L / 4 + shift, N - 1);
      y = y0[(k * 4 + L % 4) * 5 + i % 5];
      if (!isfinite(y) || s.ct[L] == 0) y = 0.f;
    }
    s.ya[i] = y;
    amax[1] = fmaxf(amax[1], fabsf(y));
  }
  block_max<2, FG::NW>(amax, s.red, tid);                 // (two barriers: s.uv / s.ya are visible afterwards)
  if (!(amax[0] > 0.f)) {                                 // uniform: no guess
    for (int i = tid; i < FG::NL * 5; i += NT) s.ya[i] = 0.f;
    __syncthreads();
    return;
  }
  const bool duals = amax[1] > 0.f;
  struct_grad<SmemF<TV>, TV, FG::N>(s, tid);              // s.gv = H u0 + g
  for (int i = tid; i < n; i += NT) { s.pu[i] = s.uv[i]; s.ua[i] = (float)s.uv[i]; s.hva[i] = (float)s.gv[i]; }
  if (tid < FG::NL) {
    const int L = tid;
    const bool stance = s.ct[L] != 0;
    const TV mu = s.mu, flo = s.cf.fmin, fhi = s.cf.fmax;
    const TV fx = s.uv[3 * L], fy = s.uv[3 * L + 1], fz = s.uv[3 * L + 2];
    const TV g[5] = {fz, fx - mu * fz, fx + mu * fz, fy - mu * fz, fy + mu * fz};
    const TV tb = (TV)1e-3 * fmax(fabs(fz), (TV)1), tf = (TV)1e-3 * fmax(mu * fabs(fz), (TV)1);
    TV y[5] = {0, 0, 0, 0, 0};
    float z[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (stance) {
      if (duals) {
#pragma unroll
        for (int i = 0; i < 5; ++i) y[i] = (TV)s.ya[5 * L + i];
      } else {
        // rows that u0 holds with equality get a unit multiplier of the right sign: the first polish step then works on
        // u0's own active set.  (Multipliers from stationarity at u0 were tried: away from the optimum they are often
        // wrong-signed and made the active-set iteration longer, tools/warm_study.py.)
        y[0] = fz >= fhi - tb ? (TV)1 : (fz <= flo + tb ? (TV)-1 : (TV)0);
        y[1] = g[1] >= -tf ? (TV)1 : (TV)0;  y[2] = g[2] <= tf ? (TV)-1 : (TV)0;     // fx - mu fz <= 0 <= fx + mu fz
        y[3] = g[3] >= -tf ? (TV)1 : (TV)0;  y[4] = g[4] <= tf ? (TV)-1 : (TV)0;
      }
      z[0] = (float)(fz < flo ? flo : (fz > fhi ? fhi : fz));
      z[1] = (float)(g[1] > 0 ? (TV)0 : g[1]);  z[2] = (float)(g[2] < 0 ? (TV)0 : g[2]);
      z[3] = (float)(g[3] > 0 ? (TV)0 : g[3]);  z[4] = (float)(g[4] < 0 ? (TV)0 : g[4]);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) { s.py[5 * L + i] = y[i]; s.za[5 * L + i] = z[i]; if (!duals) s.ya[5 * L + i] = 0.f; }
  }
  // How good is (u0, y0)?  Its stationarity residual |H u0 + g + G'y0| says whether it is (nearly) a KKT point -- then an
  // active-set step on it is worth trying first -- or only a neighbour, in which case a short ADMM block from it comes first.
  float rs[1] = {0.f};
  if (duals && tid < FG::NL && s.ct[tid] != 0) {
    const int L = tid;
    const float mu = (float)s.mu;
    const float y0_ = s.ya[5 * L], y1 = s.ya[5 * L + 1], y2 = s.ya[5 * L + 2], y3 = s.ya[5 * L + 3], y4 = s.ya[5 * L + 4];
    const float rx = (float)s.gv[3 * L] + y1 + y2, ry = (float)s.gv[3 * L + 1] + y3 + y4;
    const float rz = (float)s.gv[3 * L + 2] + y0_ + mu * (-y1 + y2 - y3 + y4);
    rs[0] = fmaxf(fmaxf(fabsf(rx), fabsf(ry)), fabsf(rz));
  }
  block_max<1, FG::NW>(rs, s.red, tid);
  if (tid == 0) s.warm = !duals ? 1 : (rs[0] <= WARM_KKT_TOL * fmaxf(s.gmax, 1.f) ? 2 : 3);
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------ phases
template <typename TV, typename TIO>
MPCQP_PHASE int ph_setup(const DevCfg* __restrict__ cfgp, const double* __restrict__ ctab, const FastIn<TIO> in, size_t b,
                         const int first) {
  SmemF<TV>& s = lds<TV>();
  const int tid = threadIdx.x;
  STAMP_INIT
  const int bad = fast_load_tuple<TV, TIO>(s, in, b, tid);
  STAMP(13);
  if (fast_setup<TV>(s, *cfgp, ctab, bad, tid, first != 0)) return 1;
  STAMP(14);
  for (int i = tid; i < FG::n; i += FG::NT) { s.ua[i] = 0.f; s.pu[i] = (TV)0; s.hva[i] = (float)s.gl[i]; }   // H 0 + g = g
  for (int i = tid; i < FG::NL * 5; i += FG::NT) { s.za[i] = 0.f; s.ya[i] = 0.f; s.py[i] = (TV)0; }
  float q[1] = {tid < FG::n ? fabsf((float)s.gl[tid]) : 0.f};
  block_max<1, FG::NW>(q, s.red, tid);
  if (tid == 0) { s.gmax = q[0]; s.rho = (float)cfgp->rho; s.iters = 0; s.psteps = 0; s.hard = 0; s.warm = 0; }
  __syncthreads();
  if (in.u_init) fast_warm_start<TV, TIO>(s, in.u_init + b * FG::n, in.y_state ? in.y_state + b * (FG::NL * 5) : nullptr, in.shift, tid);
  STAMP(0);
  return 0;
}

// One ADMM block from the state in s.ua / s.za / s.ya with penalty s.rho: matrix tiles -> sweep -> iterations.
// `adapt`: run the single early rho check (round 0 only); a QP that triggers it gets rho <- rho * ratio, a rebuilt
// matrix and a longer block.  Updates s.rho / s.iters / s.hard and leaves the new iterate in s.ua/za/ya and s.pu/py.
template <typename TV>
MPCQP_PHASE void ph_admm(const DevCfg* __restrict__ cfgp, const int adapt, const int kfirst) {
  SmemF<TV>& s = lds<TV>();
  const DevCfg& cfg = *cfgp;
  const Lane L;
  const bool stance = s.ct[L.myleg] != 0;
  const float mu = (float)s.mu, sigma = (float)cfg.sigma, relax = (float)cfg.relax;
  const float fmin = (float)s.cf.fmin, fmax = (float)s.cf.fmax;
  float rho = s.rho;
  LegLane A(L.cc, stance, mu, fmin, fmax);
  A.load(s, L.myleg);
  int K = kfirst > 0 ? min(kfirst, cfg.check_every) : cfg.check_every;
  int it = 0, seg_end = (adapt && ADAPT_AT < K) ? ADAPT_AT : K;
  bool need_build = true;
  int hard = 0;
  Tile tile;
  STAMP_INIT
  for (;;) {
    if (need_build) {
      __syncthreads();
      fast_describe<TV>(s, L.myleg, L.cc & 3, stance, s.mu, 0, rho, sigma, 0, 0, 0);
      __syncthreads();
      STAMP(1);
      fast_build<TV>(tile, s, L.grp, L.cc);
      STAMP(2);
      fast_sweep<TV>(tile, s, L.grp, L.cc, L.rbA, L.rbB);
      STAMP(3);
      need_build = false;
    }
    fast_admm_iters<TV>(tile, s, seg_end - it, rho, sigma, relax, mu, stance, A, L.cc, L.second, L.rbM);
    it = seg_end;
    STAMP(4);
    if (it >= K) break;
    const float ratio = fast_local_ratio<TV>(s, A, L.tid);   // early rho check, no gradient call
    if (ratio > ADAPT_THR) {              // uniform
      rho = fminf(rho * ratio, ADAPT_RHO_MAX);
      need_build = true;
      hard = 1;
      K = min(HARD_ITER_FACTOR * K, cfg.max_iter);
    }
    seg_end = K;
    STAMP(5);
  }
  const float end_ratio = fast_local_ratio<TV>(s, A, L.tid);   // for the next block's rho, should one be needed
  A.template publish<SmemF<TV>, TV>(s, L.myleg);
  if (L.tid == 0) { s.rho = rho; s.iters += K; s.hard |= hard; s.ratio = end_ratio; }
  __syncthreads();
}

// One active-set polish step from the iterate (s.pu, s.py): returns 1 when the refined equality-constrained solve passes
// the KKT check (answer in s.uv), else 0 with (s.pu, s.py) replaced by the candidate for the next step.
template <typename TV>
MPCQP_PHASE int ph_polish_step() {
  constexpr int N = FG::N, NW = FG::NW;
  SmemF<TV>& s = lds<TV>();
  const Lane L;
  const int tid = L.tid, grp = L.grp, cc = L.cc, myleg = L.myleg, row0 = L.row0, rbA = L.rbA, rbB = L.rbB, rbM = L.rbM;
  const bool second = L.second;
  const bool stance = s.ct[myleg] != 0;
  const TV muv = s.mu;
  const TV fminv = s.cf.fmin, fmaxv = s.cf.fmax;
  const float gmaxf = s.gmax;
  const float tol_stat = (sizeof(TV) == 8) ? (1e-6f + 1e-9f * gmaxf) : (3e-7f * fmaxf(gmaxf, 1.f));
  const float acc_stat = (sizeof(TV) == 8) ? (1e-5f + 1e-8f * gmaxf) : (1e-5f * fmaxf(gmaxf, 1.f));
  const float ftol = (sizeof(TV) == 8) ? 1e-7f : 2e-5f;
  // dual-sign slack must stay well below alpha-curvature * force tolerance (a wrongly active row with multiplier -e
  // moves the forces by ~e / (2 alpha))
  const float dtol = (sizeof(TV) == 8) ? (1e-5f + 1e-9f * gmaxf) : (2e-5f * fmaxf(1.f, gmaxf));
  bool ok = false;
  float stat = INFINITY, viol[3] = {0.f, 0.f, 0.f};
  Tile tile;
  STAMP_INIT
  {
  // primal-dual active-set rule on (pu, py): rows 0 fz | 1 fx - mu fz <= 0 | 2 fx + mu fz >= 0 | 3,4 same for fy
  int zs = 0, xs = 0, ys = 0;
  if (stance) {
    const TV u0 = s.pu[row0], u1 = s.pu[row0 + 1], u2 = s.pu[row0 + 2];
    const TV y0 = s.py[myleg * 5], y1 = s.py[myleg * 5 + 1], y2 = s.py[myleg * 5 + 2], y3 = s.py[myleg * 5 + 3], y4 = s.py[myleg * 5 + 4];
    const TV g1 = u0 - muv * u2, g2 = u0 + muv * u2, g3_ = u1 - muv * u2, g4 = u1 + muv * u2;
    if (y0 + (u2 - fmaxv) > 0) zs = 1;
    else if (y0 + (u2 - fminv) < 0) zs = -1;
    const bool hx = y1 + g1 > 0, lx = y2 + g2 < 0;
    if (hx && lx) xs = (g1 > -g2) ? 1 : -1; else if (hx) xs = 1; else if (lx) xs = -1;
    const bool hy = y3 + g3_ > 0, ly = y4 + g4 < 0;
    if (hy && ly) ys = (g3_ > -g4) ? 1 : -1; else if (hy) ys = 1; else if (ly) ys = -1;
  }
  const bool ez = stance && zs == 0, ex = stance && xs == 0, ey = stance && ys == 0;
  TV up3[3] = {0, 0, 0};
  if (stance && zs != 0) {
    const TV F = zs > 0 ? fmaxv : fminv;
    up3[2] = F;
    if (xs) up3[0] = (TV)xs * muv * F;
    if (ys) up3[1] = (TV)ys * muv * F;
  }
  TV v3[3] = {ex ? s.pu[row0] : (TV)0, ey ? s.pu[row0 + 1] : (TV)0, ez ? s.pu[row0 + 2] : (TV)0};
  __syncthreads();   // everyone has read pu / py of this round before pq / em are rewritten
  fast_describe<TV>(s, myleg, cc & 3, stance, muv, 1, 0.f, 0.f, zs, xs, ys);
  __syncthreads();
  STAMP(1);
  fast_build<TV>(tile, s, grp, cc);
  STAMP(2);
  fast_sweep<TV>(tile, s, grp, cc, rbA, rbB);
  STAMP(3);

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
  float prev_stat = INFINITY;
  stat = INFINITY;
  TV yn[5];
  ok = false;
  // stage 0: two refinement rounds, then a loose KKT screen; only a plausible candidate is refined to the tight
  // tolerance (stage 1) and checked for real.  Wrong active sets are dropped early.
  TV gr[3] = {0, 0, 0}, rgv[3] = {0, 0, 0};
  for (int stg = 0; stg < 2; ++stg) {
    const float tol = stg == 0 ? fmaxf(tol_stat, 1e-3f * fmaxf(gmaxf, 1.f)) : tol_stat;
    const int max_rf = stg == 0 ? 2 : 10;
    for (int rf = 0;; ++rf) {
      if (!(stg == 1 && rf == 0)) {   // (stage 1 starts from the gradient stage 0 ended with: the candidate has not moved)
        if ((cc & 3) == 0) {
#pragma unroll
          for (int c = 0; c < 3; ++c) s.uv[row0 + c] = uc[c];
        }
        __syncthreads();
        struct_grad<SmemF<TV>, TV, N>(s, tid);
#pragma unroll
        for (int c = 0; c < 3; ++c) gr[c] = s.gv[row0 + c];
        rgv[0] = ex ? gr[0] : (TV)0; rgv[1] = ey ? gr[1] : (TV)0;
        rgv[2] = ez ? gr[2] + (TV)xs * muv * gr[0] + (TV)ys * muv * gr[1] : (TV)0;
        float q[1] = {fmaxf(fmaxf(fabsf((float)rgv[0]), fabsf((float)rgv[1])), fabsf((float)rgv[2]))};
        if (!isfinite(q[0])) q[0] = INFINITY;
        block_max<1, NW>(q, s.red, tid);
        prev_stat = stat;
        stat = q[0];
      }
      if (stat <= tol || rf >= max_rf || (rf > 0 && !(stat < 0.5f * prev_stat))) break;  // converged / stagnated (uniform)
      if ((cc & 3) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) s.rhs[rbM + c] = (float)(-rgv[c]);
      }
      __syncthreads();
      float sum[3];
      fast_matvec(tile, s.rhs, cc, second, sum);
      v3[0] += (TV)sum[0];
      v3[1] += (TV)sum[1];
      v3[2] += (TV)sum[2];
      if (!ex) v3[0] = 0;
      if (!ey) v3[1] = 0;
      if (!ez) v3[2] = 0;
      expand();
    }
    // duals from stationarity grad_leg + G_A' y_A = 0, then primal feasibility + dual sign
#pragma unroll
    for (int i = 0; i < 5; ++i) yn[i] = 0;
    viol[0] = viol[1] = viol[2] = 0.f;
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
  if ((cc & 3) == 0) {   // publish the candidate as the next polish iterate / the answer
#pragma unroll
    for (int c = 0; c < 3; ++c) { s.pu[row0 + c] = uc[c]; s.uv[row0 + c] = uc[c]; }
#pragma unroll
    for (int i = 0; i < 5; ++i) s.py[myleg * 5 + i] = yn[i];
  }
  __syncthreads();
    STAMP(7);
  }
  if (tid == 0) { s.kkt[0] = stat; s.kkt[1] = viol[0]; s.kkt[2] = viol[1]; s.psteps += 1; }
  __syncthreads();
  return ok ? 1 : 0;
}

template <typename TV, typename TIO>
MPCQP_PHASE void ph_output(TIO* __restrict__ ug, TIO* __restrict__ Xg, int* __restrict__ statusg, int* __restrict__ itersg,
                           float* __restrict__ resg, float* __restrict__ y_state, const size_t b, const int ok) {
  constexpr int N = FG::N, n = FG::n, NT = FG::NT;
  SmemF<TV>& s = lds<TV>();
  const Lane L;
  const int tid = L.tid, row0 = L.row0;
  STAMP_INIT
  const bool stance = s.ct[L.myleg] != 0;
  if (!ok && (L.cc & 3) == 0) {   // iteration cap: hand back the last ADMM iterate
#pragma unroll
    for (int c = 0; c < 3; ++c) s.uv[row0 + c] = (TV)s.ua[row0 + c];
  }
  if ((L.cc & 3) == 0 && !stance) s.uv[row0] = s.uv[row0 + 1] = s.uv[row0 + 2] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += NT) ug[b * n + i] = (TIO)s.uv[i];   // src/mpc.py:267-268
  if (y_state) {   // warm-started engines remember the multipliers of this slot for the next call
    for (int i = tid; i < FG::NL * 5; i += NT) y_state[b * (FG::NL * 5) + i] = ok ? (float)s.py[i] : s.ya[i];
  }
  if (Xg) {                                                          // src/mpc.py:265-266
    struct_grad<SmemF<TV>, TV, N>(s, tid);
    for (int i = tid; i < (N + 1) * 13; i += NT) {
      const int k = i / 13, c = i % 13;
      const TV v = (c == 12 || k == 0) ? s.x0[c] : s.Xs[k * 12 + c];
      Xg[b * (N + 1) * 13 + i] = (TIO)v;
    }
  }
  if (tid == 0) {
    statusg[b] = ok ? MPCQP_STATUS_SOLVED_POLISHED : MPCQP_STATUS_MAX_ITER;
    itersg[b] = s.iters + 1000 * s.psteps;
    if (resg) { resg[2 * b] = s.kkt[1]; resg[2 * b + 1] = fmaxf(s.kkt[2], s.kkt[0]); }
  }
  STAMP(8);
}

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
template <typename TIO, int N = FG::N>
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

// ------------------------------------------------------------------------------------------------------ the kernel
// Two launch forms.  Plain: one workgroup per QP, blockIdx = QP.  Queued (ob.list != null, batches that oversubscribe
// the device): the grid is only as large as the device holds at once (2 workgroups per CU) and every workgroup pulls
// QPs from the dearest-first order until the queue is empty -- no workgroup turnover between QPs, and the long solves
// start first.  Every wave leaves the loop when the queue index passes B (bounded by the `guard` count as well).
template <typename TV, typename TIO>
__global__ void __launch_bounds__(FG::NT, MPCQP_FAST_WPE)
mpcqp_fast_solve(const DevCfg* __restrict__ cfgp, const double* __restrict__ ctab, const FastIn<TIO> in,
                 TIO* ug, TIO* __restrict__ Xg, int* __restrict__ statusg, int* __restrict__ itersg,
                 float* __restrict__ resg, const OrderBuf ob, const int Btot) {   // (ug is not restrict: the warm-start guess is read from the same buffer)
  constexpr int N = FG::N, n = FG::n, NT = FG::NT;
  __shared__ int s_next;
  const int tid = threadIdx.x;
  for (int guard = 0; guard <= Btot; ++guard) {
    size_t b = blockIdx.x;
    if (ob.list) {
      if (tid == 0) s_next = atomicAdd(ob.head, 1);
      __syncthreads();
      int i = s_next;
      __syncthreads();                       // (also fences the previous QP's last LDS reads from the next one's setup)
      if (i >= Btot) break;                  // uniform
      for (int k = ORDER_BUCKETS - 1; k >= 0; --k) {   // dearest class first (uniform scalar walk over the class counts)
        const int c = ob.cnt[k];
        if (i < c) { b = (size_t)ob.list[(size_t)k * ob.cap + i]; break; }
        i -= c;
      }
    }
#ifdef MPCQP_STAMPS
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (ph_setup<TV, TIO>(cfgp, ctab, in, b, guard == 0 ? 1 : 0)) {   // non-finite input -> zero outputs, status -1
      for (int i = tid; i < n; i += NT) ug[b * n + i] = (TIO)0;
      if (Xg) for (int i = tid; i < (N + 1) * 13; i += NT) Xg[b * (N + 1) * 13 + i] = (TIO)0;
      if (tid == 0) {
        statusg[b] = MPCQP_STATUS_NONFINITE;
        itersg[b] = 0;
        if (resg) { resg[2 * b] = 0.f; resg[2 * b + 1] = 0.f; }
      }
      if (!ob.list) break;
      continue;
    }
    const int max_iter = cfgp->max_iter, polish_max = cfgp->polish_max;
    int ok = 0;
    const int warm = lds<TV>().warm;
    if (warm) {   // warm start: the guess's own active set first, no ADMM unless that fails
      const int tries = warm == 1 ? WARM_POLISH : (warm == 2 ? 1 : 0);
      for (int ps = 0; ps < polish_max && ps < tries && !ok; ++ps) ok = ph_polish_step<TV>();
    }
    for (int round = 0; !ok; ++round) {
      ph_admm<TV>(cfgp, round == 0 ? 1 : 0, (round == 0 && warm >= 2) ? WARM_K : 0);
      SmemF<TV>& s = lds<TV>();
      const int budget = (s.hard ? HARD_POLISH_FACTOR : 1) * polish_max;
      for (int ps = 0; ps < budget && !ok; ++ps) ok = ph_polish_step<TV>();
      if (ok || s.iters >= max_iter) break;
      // not solved: OSQP's rho adaptation from the residuals of the last ADMM iterate, then another block
      const float ratio = s.ratio;
      if (tid == 0 && isfinite(ratio) && (ratio > 2.f || ratio < 0.5f)) s.rho = fminf(fmaxf(s.rho * ratio, 1e-4f), 1e4f);
      __syncthreads();
    }
    ph_output<TV, TIO>(ug, Xg, statusg, itersg, resg, in.y_state, b, ok);
#ifdef MPCQP_STAMPS
    if (tid == 0 && b < 65536) {
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_timeline[3 * b] = tl_t0; g_timeline[3 * b + 1] = __builtin_amdgcn_s_memtime();
      g_timeline[3 * b + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    if (!ob.list) break;
  }
}

}  // namespace
