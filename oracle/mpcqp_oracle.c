/*
 * mpcqp_oracle.c -- CPU fp64 restatement of the reference's convex-MPC QP path, exporting the C-ABI of
 * include/mpcqp.h with HOST pointers.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / reported CPU baseline -- never on the product path.
 *
 * PARITY STATUS.  Problem construction follows the reference line by line (citations below) and is pinned
 * against the inputs recoverable from the reference's committed run log (tests/test_oracle_pinning.py).
 * The solver the reference calls -- CasADi Opti('conic') -> OSQP (src/mpc.py:49-55,258), both un-vendored
 * and un-pinned (no requirements file; README.md:138 names osqp without a version) -- is absent from this
 * image, so solver OUTPUTS are "parity unpinned" against OSQP.  They are certified instead by the KKT
 * conditions of the QP that src/mpc.py:58-173 defines (checked independently in numpy by oracle/qp_spec.py).
 * The algorithm is the published OSQP ADMM (Stellato et al., "OSQP: an operator splitting solver for
 * quadratic programs", Math. Prog. Comp. 2020, Algorithm 1, with rho-adaptation and polish) applied to the
 * condensed form of the same optimal-control problem.
 *
 * Deliberately generic: dense 13x13 / 13x12 matrices, matrix-product condensing, dense Cholesky.  It shares
 * no closed forms with the HIP engine, so agreement between the two is evidence, not tautology.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/mpcqp.h"

#define NX 13
#define NU 12

struct mpcqp_engine {
  MpcQpConfig cfg;
  char err[256];
  float last_ms;
};

uint32_t mpcqp_version(void) { return MPCQP_VERSION; }

int mpcqp_default_config(MpcQpConfig* c) {
  if (!c) return MPCQP_EINVAL;
  memset(c, 0, sizeof(*c));
  c->size = (uint32_t)sizeof(*c);
  c->N = 10;
  c->delta = 0.03;
  c->m = 8.885;                                      /* src/mpc.py:71 */
  c->Ibody_inv[0] = 1.0 / 0.24; c->Ibody_inv[1] = 1.0; c->Ibody_inv[2] = 1.0; /* src/mpc.py:73-76 */
  const double w[13] = {1e4, 2.7e4, 1e4, 2.7e5, 2.7e5, 2.7e5, 1e4, 1e4, 1e4, 1.6e4, 1.6e4, 1.6e4, 0.0};
  memcpy(c->w, w, sizeof(w));                        /* src/mpc.py:122-134 */
  c->alpha = 1e-2;                                   /* benchmark default (reference: 0.0, src/mpc.py:121) */
  c->f_min = 3.0; c->f_max = 100.0;                  /* src/mpc.py:45-46 */
  c->disc = MPCQP_DISC_EULER;
  c->dtype = MPCQP_DTYPE_F64;
  c->precision = MPCQP_PREC_F64;
  c->flags = MPCQP_FLAG_POLISH;
  c->rho = 1.0; c->sigma = 1e-6; c->relax = 1.6;
  c->max_iter = 4000; c->check_every = 25;
  c->eps_abs = 1e-9; c->eps_rel = 1e-9;
  c->polish_max = 10;
  c->device = 0;
  return MPCQP_OK;
}

int mpcqp_create(const MpcQpConfig* cfg, mpcqp_handle* out) {
  if (!cfg || !out || cfg->size != sizeof(MpcQpConfig)) return MPCQP_EINVAL;
  if (cfg->N < 1 || cfg->N > 256 || cfg->dtype != MPCQP_DTYPE_F64) return MPCQP_EINVAL;
  struct mpcqp_engine* e = (struct mpcqp_engine*)calloc(1, sizeof(*e));
  if (!e) return MPCQP_ENOMEM;
  e->cfg = *cfg;
  *out = e;
  return MPCQP_OK;
}

int mpcqp_destroy(mpcqp_handle h) { free(h); return MPCQP_OK; }
const char* mpcqp_last_error(mpcqp_handle h) { return h ? h->err : "null handle"; }
int mpcqp_reserve(mpcqp_handle h, int64_t B) { (void)B; return h ? MPCQP_OK : MPCQP_EINVAL; }   /* host memory: nothing to pre-size */
int mpcqp_last_kernel_ms(mpcqp_handle h, float* ms) { if (!h || !ms) return MPCQP_EINVAL; *ms = h->last_ms; return MPCQP_OK; }

/* ---------------------------------------------------------------- model, literal to src/mpc.py ------- */
static void skew(const double* v, double S[9]) { /* src/utils.py:43-56 */
  S[0] = 0; S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2]; S[4] = 0; S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static void matmul(const double* A, const double* B, double* C, int m, int k, int n) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
      C[i * n + j] = s;
    }
}
static void build_A(double yaw, double A[NX * NX]) { /* src/mpc.py:64-69, 86-96 */
  memset(A, 0, sizeof(double) * NX * NX);
  double c = cos(yaw), s = sin(yaw);
  double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i * NX + 6 + j] = Rz[i * 3 + j]; /* Theta_dot = Rz omega */
  for (int i = 0; i < 3; i++) A[(3 + i) * NX + 9 + i] = 1.0;       /* p_dot = v */
  A[11 * NX + 12] = 1.0;                                            /* v_z_dot += g state */
}
static void build_B(double yaw, const double* r4, const MpcQpConfig* cfg, double B[NX * NU]) { /* src/mpc.py:71-78, 98-107 */
  memset(B, 0, sizeof(double) * NX * NU);
  double c = cos(yaw), s = sin(yaw);
  double Rz[9] = {c, -s, 0, s, c, 0, 0, 0, 1}, RzT[9] = {c, s, 0, -s, c, 0, 0, 0, 1};
  double Ib[9] = {cfg->Ibody_inv[0], 0, 0, 0, cfg->Ibody_inv[1], 0, 0, 0, cfg->Ibody_inv[2]};
  double T1[9], Ihat[9];
  matmul(Rz, Ib, T1, 3, 3, 3);
  matmul(T1, RzT, Ihat, 3, 3, 3);
  for (int j = 0; j < 4; j++) {
    double S[9], IS[9];
    skew(r4 + 3 * j, S);
    matmul(Ihat, S, IS, 3, 3, 3);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) B[(6 + a) * NU + 3 * j + b] = IS[a * 3 + b];
    for (int a = 0; a < 3; a++) B[(9 + a) * NU + 3 * j + a] = 1.0 / cfg->m;
  }
}
static void discretise(const double* A, const double* B, double delta, int disc, double* Ad, double* Bd) {
  if (disc == MPCQP_DISC_EULER) { /* src/mpc.py:117 */
    for (int i = 0; i < NX * NX; i++) Ad[i] = delta * A[i];
    for (int i = 0; i < NX; i++) Ad[i * NX + i] += 1.0;
    for (int i = 0; i < NX * NU; i++) Bd[i] = delta * B[i];
  } else { /* exact ZOH: expm series terminates (A^3 = 0, A^2 B = 0) */
    double A2[NX * NX], AB[NX * NU];
    matmul(A, A, A2, NX, NX, NX);
    matmul(A, B, AB, NX, NX, NU);
    for (int i = 0; i < NX * NX; i++) Ad[i] = delta * A[i] + 0.5 * delta * delta * A2[i];
    for (int i = 0; i < NX; i++) Ad[i * NX + i] += 1.0;
    for (int i = 0; i < NX * NU; i++) Bd[i] = delta * B[i] + 0.5 * delta * delta * AB[i];
  }
}

/* ---------------------------------------------------------------- dense helpers ------------------------ */
static int cholesky(double* M, int n) { /* in place, lower; returns -1 if not PD */
  for (int j = 0; j < n; j++) {
    double d = M[j * n + j];
    for (int k = 0; k < j; k++) d -= M[j * n + k] * M[j * n + k];
    if (!(d > 0)) return -1;
    d = sqrt(d);
    M[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = M[i * n + j];
      for (int k = 0; k < j; k++) s -= M[i * n + k] * M[j * n + k];
      M[i * n + j] = s / d;
    }
  }
  return 0;
}
static void chol_solve(const double* L, int n, double* x) {
  for (int i = 0; i < n; i++) {
    double s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}
static void symv(const double* H, int n, const double* x, double* y) {
  for (int i = 0; i < n; i++) {
    double s = 0;
    const double* row = H + (size_t)i * n;
    for (int j = 0; j < n; j++) s += row[j] * x[j];
    y[i] = s;
  }
}

/* per leg-stage constraint rows (unique rows of src/mpc.py:151-173):
 *   0: fz            1: fx - mu fz    2: fx + mu fz    3: fy - mu fz    4: fy + mu fz            */
static inline void G_leg(const double* u3, double mu, double* z5) {
  z5[0] = u3[2];
  z5[1] = u3[0] - mu * u3[2]; z5[2] = u3[0] + mu * u3[2];
  z5[3] = u3[1] - mu * u3[2]; z5[4] = u3[1] + mu * u3[2];
}
static inline void GT_leg(const double* v5, double mu, double* o3) {
  o3[0] = v5[1] + v5[2];
  o3[1] = v5[3] + v5[4];
  o3[2] = v5[0] + mu * (-v5[1] + v5[2] - v5[3] + v5[4]);
}

typedef struct {
  int N, n, nleg;
  double *Sx, *Su, *H, *g, *M, *AdAll, *BdAll;
  double *u, *ut, *z, *y, *rhs, *tmp, *tmp2, *lo, *hi, *rho;
  double *Hr, *vr, *gr;
  int* map;
} Work;

static Work* work_alloc(int N) {
  Work* w = (Work*)calloc(1, sizeof(Work));
  int n = NU * N, s = NX * (N + 1), nl = 4 * N;
  w->N = N; w->n = n; w->nleg = nl;
  w->Sx = (double*)malloc(sizeof(double) * s * NX);
  w->Su = (double*)malloc(sizeof(double) * (size_t)s * n);
  w->H = (double*)malloc(sizeof(double) * (size_t)n * n);
  w->M = (double*)malloc(sizeof(double) * (size_t)n * n);
  w->Hr = (double*)malloc(sizeof(double) * (size_t)n * n);
  w->AdAll = (double*)malloc(sizeof(double) * NX * NX);
  w->BdAll = (double*)malloc(sizeof(double) * N * NX * NU);
  w->g = (double*)malloc(sizeof(double) * n); w->u = (double*)malloc(sizeof(double) * n);
  w->ut = (double*)malloc(sizeof(double) * n); w->rhs = (double*)malloc(sizeof(double) * n);
  w->tmp = (double*)malloc(sizeof(double) * (n > s ? n : s)); w->tmp2 = (double*)malloc(sizeof(double) * (n > s ? n : s));
  w->vr = (double*)malloc(sizeof(double) * n); w->gr = (double*)malloc(sizeof(double) * n);
  w->z = (double*)malloc(sizeof(double) * 5 * nl); w->y = (double*)malloc(sizeof(double) * 5 * nl);
  w->lo = (double*)malloc(sizeof(double) * 5 * nl); w->hi = (double*)malloc(sizeof(double) * 5 * nl);
  w->rho = (double*)malloc(sizeof(double) * nl);
  w->map = (int*)malloc(sizeof(int) * n);
  return w;
}
static void work_free(Work* w) {
  free(w->Sx); free(w->Su); free(w->H); free(w->M); free(w->Hr); free(w->AdAll); free(w->BdAll); free(w->g);
  free(w->u); free(w->ut); free(w->rhs); free(w->tmp); free(w->tmp2); free(w->vr); free(w->gr); free(w->z);
  free(w->y); free(w->lo); free(w->hi); free(w->rho); free(w->map); free(w);
}

/* Build condensed QP: X = Sx x0 + Su U by recursion on (Ad, Bd_k) (src/mpc.py:110-117); swing-leg columns of
 * Bd_k are zeroed (their forces are pinned to 0 by src/mpc.py:139-144).  H = 2 Su' W Su + 2 alpha I,
 * g = 2 Su' W (Sx x0 - xdes) (cost: src/mpc.py:121-134). */
static void build_qp(Work* w, const MpcQpConfig* cfg, const double* x0, const double* r, const uint8_t* contact,
                     const double* xdes) {
  const int N = w->N, n = w->n;
  double A[NX * NX], B[NX * NU];
  build_A(x0[2], A); /* yaw = x0_param[2], src/mpc.py:64 */
  double* Ad = w->AdAll;
  memset(w->Sx, 0, sizeof(double) * NX * (N + 1) * NX);
  memset(w->Su, 0, sizeof(double) * (size_t)NX * (N + 1) * n);
  for (int i = 0; i < NX; i++) w->Sx[i * NX + i] = 1.0;
  for (int k = 0; k < N; k++) {
    double* Bd = w->BdAll + (size_t)k * NX * NU;
    build_B(x0[2], r + (size_t)k * 12, cfg, B);
    discretise(A, B, cfg->delta, cfg->disc, Ad, Bd);
    for (int j = 0; j < 4; j++)
      if (!contact[k * 4 + j])
        for (int i = 0; i < NX; i++)
          for (int a = 0; a < 3; a++) Bd[i * NU + 3 * j + a] = 0.0;
    matmul(Ad, w->Sx + (size_t)k * NX * NX, w->Sx + (size_t)(k + 1) * NX * NX, NX, NX, NX);
    const double* Sk = w->Su + (size_t)k * NX * n;
    double* Sk1 = w->Su + (size_t)(k + 1) * NX * n;
    for (int i = 0; i < NX; i++)
      for (int c = 0; c < NU * k; c++) {
        double s = 0;
        for (int l = 0; l < NX; l++) s += Ad[i * NX + l] * Sk[(size_t)l * n + c];
        Sk1[(size_t)i * n + c] = s;
      }
    for (int i = 0; i < NX; i++)
      for (int c = 0; c < NU; c++) Sk1[(size_t)i * n + NU * k + c] += Bd[i * NU + c];
  }
  const int s = NX * (N + 1);
  double* e0 = w->tmp; /* W (Sx x0 - xdes) */
  for (int i = 0; i < s; i++) {
    double v = 0;
    for (int l = 0; l < NX; l++) v += w->Sx[(size_t)i * NX + l] * x0[l];
    e0[i] = cfg->w[i % NX] * (v - xdes[i]);
  }
  for (int a = 0; a < n; a++) {
    double gs = 0;
    for (int i = 0; i < s; i++) gs += w->Su[(size_t)i * n + a] * e0[i];
    w->g[a] = 2.0 * gs;
  }
  memset(w->H, 0, sizeof(double) * (size_t)n * n);
  for (int i = NX; i < s; i++) { /* rows of stage 0 are zero in Su */
    const double wi = 2.0 * cfg->w[i % NX];
    if (wi == 0.0) continue;
    const double* row = w->Su + (size_t)i * n;
    const int kmax = NU * (i / NX); /* causal: stage k depends on U_0..U_{k-1} */
    for (int a = 0; a < kmax; a++) {
      const double ra = wi * row[a];
      if (ra == 0.0) continue;
      double* Ha = w->H + (size_t)a * n;
      for (int b = 0; b <= a; b++) Ha[b] += ra * row[b];
    }
  }
  for (int a = 0; a < n; a++) {
    for (int b = 0; b < a; b++) w->H[(size_t)b * n + a] = w->H[(size_t)a * n + b];
    w->H[(size_t)a * n + a] += 2.0 * cfg->alpha;
  }
}

static void rollout(const Work* w, const double* x0, const double* u, double* X) { /* src/mpc.py:113-117 */
  const int N = w->N;
  memcpy(X, x0, sizeof(double) * NX);
  for (int k = 0; k < N; k++) {
    const double* Bd = w->BdAll + (size_t)k * NX * NU;
    for (int i = 0; i < NX; i++) {
      double s = 0;
      for (int l = 0; l < NX; l++) s += w->AdAll[i * NX + l] * X[k * NX + l];
      for (int l = 0; l < NU; l++) s += Bd[i * NU + l] * u[k * NU + l];
      X[(k + 1) * NX + i] = s;
    }
  }
}

static int factor_M(Work* w, const MpcQpConfig* cfg, const double* mu_leg, double sigma) {
  const int n = w->n;
  memcpy(w->M, w->H, sizeof(double) * (size_t)n * n);
  for (int l = 0; l < w->nleg; l++) {
    const double mu = mu_leg[l], rh = w->rho[l];
    w->M[(size_t)(3 * l + 0) * n + 3 * l + 0] += sigma + 2.0 * rh;
    w->M[(size_t)(3 * l + 1) * n + 3 * l + 1] += sigma + 2.0 * rh;
    w->M[(size_t)(3 * l + 2) * n + 3 * l + 2] += sigma + rh * (1.0 + 4.0 * mu * mu);
  }
  (void)cfg;
  return cholesky(w->M, n);
}

/* residuals of the QP: primal |Gu - z|_inf, dual |Hu + g + G'y|_inf, and the OSQP normalisers */
static void residuals(Work* w, const double* mu_leg, double* rp, double* rd, double* sp, double* sd) {
  const int n = w->n;
  double a = 0, b = 0, np_ = 0, nz = 0, nHu = 0, nGy = 0, ng = 0;
  symv(w->H, n, w->u, w->tmp);
  for (int l = 0; l < w->nleg; l++) {
    double gu[5], gy[3];
    G_leg(w->u + 3 * l, mu_leg[l], gu);
    GT_leg(w->y + 5 * l, mu_leg[l], gy);
    for (int i = 0; i < 5; i++) {
      a = fmax(a, fabs(gu[i] - w->z[5 * l + i]));
      np_ = fmax(np_, fabs(gu[i])); nz = fmax(nz, fabs(w->z[5 * l + i]));
    }
    for (int c = 0; c < 3; c++) {
      b = fmax(b, fabs(w->tmp[3 * l + c] + w->g[3 * l + c] + gy[c]));
      nHu = fmax(nHu, fabs(w->tmp[3 * l + c])); nGy = fmax(nGy, fabs(gy[c])); ng = fmax(ng, fabs(w->g[3 * l + c]));
    }
  }
  *rp = a; *rd = b; *sp = fmax(np_, nz); *sd = fmax(fmax(nHu, nGy), ng);
}

/* Active-set polish (OSQP's polish idea, specialised to the per-leg rows).  For each leg the active rows fix
 * fz and/or tie fx, fy to +-mu fz; the equality-constrained QP is solved in the reduced variables, duals are
 * read from the gradient, and the active set is updated by the primal-dual rule until it stops changing.
 * Returns 1 when the result passes the KKT check. */
typedef struct { int zs, xs, ys; } LegAS; /* zs: 0 free, -1 at f_min, +1 at f_max; xs/ys: 0 free, -1: f = -mu fz, +1: f = +mu fz */

static int polish(Work* w, const MpcQpConfig* cfg, const uint8_t* contact, const double* mu_leg, int* steps_out) {
  const int n = w->n, nl = w->nleg;
  LegAS* as = (LegAS*)malloc(sizeof(LegAS) * nl);
  LegAS* prev = (LegAS*)malloc(sizeof(LegAS) * nl);
  double* u = w->ut;      /* candidate */
  double* yfull = (double*)calloc(5 * (size_t)nl, sizeof(double));
  memcpy(u, w->u, sizeof(double) * n);
  memcpy(yfull, w->y, sizeof(double) * 5 * nl);
  int ok = 0, step;
  const double c = 1.0;
  for (step = 0; step < cfg->polish_max; step++) {
    /* primal-dual active-set rule on (u, y) */
    for (int l = 0; l < nl; l++) {
      as[l].zs = as[l].xs = as[l].ys = 0;
      if (!contact[l]) continue;
      double gu[5];
      const double mu = mu_leg[l];
      G_leg(u + 3 * l, mu, gu);
      const double* yl = yfull + 5 * l;
      if (yl[0] + c * (gu[0] - cfg->f_max) > 0) as[l].zs = 1;
      else if (yl[0] + c * (gu[0] - cfg->f_min) < 0) as[l].zs = -1;
      double hi_x = yl[1] + c * gu[1], lo_x = yl[2] + c * gu[2]; /* row1: fx - mu fz <= 0 ; row2: fx + mu fz >= 0 */
      if (hi_x > 0 && lo_x < 0) as[l].xs = (gu[1] > -gu[2]) ? 1 : -1;
      else if (hi_x > 0) as[l].xs = 1;
      else if (lo_x < 0) as[l].xs = -1;
      double hi_y = yl[3] + c * gu[3], lo_y = yl[4] + c * gu[4];
      if (hi_y > 0 && lo_y < 0) as[l].ys = (gu[3] > -gu[4]) ? 1 : -1;
      else if (hi_y > 0) as[l].ys = 1;
      else if (lo_y < 0) as[l].ys = -1;
    }
    if (step > 0 && memcmp(as, prev, sizeof(LegAS) * nl) == 0) { ok = 1; break; }
    memcpy(prev, as, sizeof(LegAS) * nl);
    /* reduced variables: per leg u = Z v + up */
    int nr = 0;
    double* up = w->rhs;
    memset(up, 0, sizeof(double) * n);
    for (int i = 0; i < n; i++) w->map[i] = -1;
    for (int l = 0; l < nl; l++) {
      if (!contact[l]) continue;
      const double mu = mu_leg[l];
      const LegAS a = as[l];
      if (a.zs != 0) {
        const double F = a.zs > 0 ? cfg->f_max : cfg->f_min;
        up[3 * l + 2] = F;
        if (a.xs) up[3 * l + 0] = a.xs * mu * F; else w->map[3 * l + 0] = nr++;
        if (a.ys) up[3 * l + 1] = a.ys * mu * F; else w->map[3 * l + 1] = nr++;
      } else {
        w->map[3 * l + 2] = nr++;
        if (!a.xs) w->map[3 * l + 0] = nr++;
        if (!a.ys) w->map[3 * l + 1] = nr++;
      }
    }
    /* Z as a sparse column list: column j of Z has entries (row, coef) */
    /* reduced Hessian Hr = Z' H Z, reduced gradient gr = Z'(g + H up) */
    symv(w->H, n, up, w->tmp);
    for (int i = 0; i < n; i++) w->tmp[i] += w->g[i];
    /* build HZ in place into M (n x nr), then Z'(HZ) */
    double* HZ = w->M;
    for (int i = 0; i < n; i++) {
      const double* Hi = w->H + (size_t)i * n;
      double* o = HZ + (size_t)i * nr;
      for (int l = 0; l < nl; l++) {
        if (!contact[l]) continue;
        const double mu = mu_leg[l];
        const LegAS a = as[l];
        if (a.zs == 0) {
          double v = Hi[3 * l + 2];
          if (a.xs) v += a.xs * mu * Hi[3 * l + 0];
          if (a.ys) v += a.ys * mu * Hi[3 * l + 1];
          o[w->map[3 * l + 2]] = v;
        }
        if (w->map[3 * l + 0] >= 0) o[w->map[3 * l + 0]] = Hi[3 * l + 0];
        if (w->map[3 * l + 1] >= 0) o[w->map[3 * l + 1]] = Hi[3 * l + 1];
      }
    }
    for (int j = 0; j < nr; j++) w->gr[j] = 0;
    memset(w->Hr, 0, sizeof(double) * (size_t)nr * nr);
    for (int l = 0; l < nl; l++) {
      if (!contact[l]) continue;
      const double mu = mu_leg[l];
      const LegAS a = as[l];
      for (int cidx = 0; cidx < 3; cidx++) {
        const int row = 3 * l + cidx;
        int j; double coef = 1.0;
        if (w->map[row] >= 0) j = w->map[row];
        else if (a.zs == 0 && cidx == 0 && a.xs) { j = w->map[3 * l + 2]; coef = a.xs * mu; }
        else if (a.zs == 0 && cidx == 1 && a.ys) { j = w->map[3 * l + 2]; coef = a.ys * mu; }
        else continue;
        w->gr[j] += coef * w->tmp[row];
        const double* hz = HZ + (size_t)row * nr;
        double* Hj = w->Hr + (size_t)j * nr;
        for (int q = 0; q < nr; q++) Hj[q] += coef * hz[q];
      }
    }
    if (nr > 0) {
      memcpy(w->M, w->Hr, sizeof(double) * (size_t)nr * nr);
      if (cholesky(w->M, nr) != 0) break;
      for (int j = 0; j < nr; j++) w->vr[j] = -w->gr[j];
      chol_solve(w->M, nr, w->vr);
      for (int it = 0; it < 3; it++) { /* iterative refinement */
        for (int j = 0; j < nr; j++) {
          double s2 = w->gr[j];
          for (int q = 0; q < nr; q++) s2 += w->Hr[(size_t)j * nr + q] * w->vr[q];
          w->tmp2[j] = -s2;
        }
        double* dv = (double*)malloc(sizeof(double) * nr);
        memcpy(dv, w->tmp2, sizeof(double) * nr);
        chol_solve(w->M, nr, dv);
        for (int j = 0; j < nr; j++) w->vr[j] += dv[j];
        free(dv);
      }
    }
    /* expand u = Z v + up */
    memcpy(u, up, sizeof(double) * n);
    for (int l = 0; l < nl; l++) {
      if (!contact[l]) continue;
      const double mu = mu_leg[l];
      const LegAS a = as[l];
      if (a.zs == 0) {
        const double fz = w->vr[w->map[3 * l + 2]];
        u[3 * l + 2] = fz;
        if (a.xs) u[3 * l + 0] = a.xs * mu * fz;
        if (a.ys) u[3 * l + 1] = a.ys * mu * fz;
      }
      if (w->map[3 * l + 0] >= 0) u[3 * l + 0] = w->vr[w->map[3 * l + 0]];
      if (w->map[3 * l + 1] >= 0) u[3 * l + 1] = w->vr[w->map[3 * l + 1]];
    }
    /* duals from the gradient: grad_leg + G_A' y_A = 0 */
    symv(w->H, n, u, w->tmp);
    memset(yfull, 0, sizeof(double) * 5 * nl);
    for (int l = 0; l < nl; l++) {
      if (!contact[l]) continue;
      const double mu = mu_leg[l];
      const LegAS a = as[l];
      const double gx = w->tmp[3 * l + 0] + w->g[3 * l + 0], gy = w->tmp[3 * l + 1] + w->g[3 * l + 1],
                   gz = w->tmp[3 * l + 2] + w->g[3 * l + 2];
      double* yl = yfull + 5 * l;
      double zacc = gz; /* gz + y0 + mu*(-y1 + y2 - y3 + y4) = 0 */
      if (a.xs > 0) { yl[1] = -gx; zacc += mu * (-yl[1]); }
      else if (a.xs < 0) { yl[2] = -gx; zacc += mu * (yl[2]); }
      if (a.ys > 0) { yl[3] = -gy; zacc += mu * (-yl[3]); }
      else if (a.ys < 0) { yl[4] = -gy; zacc += mu * (yl[4]); }
      if (a.zs != 0) yl[0] = -zacc;
    }
  }
  *steps_out = step;
  if (ok) {
    /* KKT check of the candidate */
    double stat = 0, prim = 0, dsgn = 0, umax = 1.0;
    symv(w->H, n, u, w->tmp);
    for (int i = 0; i < n; i++) umax = fmax(umax, fabs(u[i]));
    for (int l = 0; l < nl; l++) {
      double gu[5], gy[3];
      const double mu = mu_leg[l];
      if (!contact[l]) { for (int cidx = 0; cidx < 3; cidx++) prim = fmax(prim, fabs(u[3 * l + cidx])); continue; }
      G_leg(u + 3 * l, mu, gu);
      GT_leg(yfull + 5 * l, mu, gy);
      for (int cidx = 0; cidx < 3; cidx++) stat = fmax(stat, fabs(w->tmp[3 * l + cidx] + w->g[3 * l + cidx] + gy[cidx]));
      prim = fmax(prim, fmax(cfg->f_min - gu[0], gu[0] - cfg->f_max));
      prim = fmax(prim, fmax(gu[1], -gu[2])); prim = fmax(prim, fmax(gu[3], -gu[4]));
      const double* yl = yfull + 5 * l;
      /* sign conventions: upper-bounded rows (fz at f_max, rows 1,3) need y >= 0; lower-bounded (fz at f_min, rows 2,4) y <= 0 */
      if (as[l].zs > 0) dsgn = fmax(dsgn, -yl[0]);
      if (as[l].zs < 0) dsgn = fmax(dsgn, yl[0]);
      dsgn = fmax(dsgn, -yl[1]); dsgn = fmax(dsgn, yl[2]); dsgn = fmax(dsgn, -yl[3]); dsgn = fmax(dsgn, yl[4]);
    }
    double gmax = 1.0;
    for (int i = 0; i < n; i++) gmax = fmax(gmax, fabs(w->g[i]));
    if (!(stat <= 1e-9 * gmax && prim <= 1e-9 * umax && dsgn <= 1e-9 * gmax)) ok = 0;
  }
  if (ok) {
    memcpy(w->u, u, sizeof(double) * n);
    memcpy(w->y, yfull, sizeof(double) * 5 * nl);
    for (int l = 0; l < nl; l++) G_leg(w->u + 3 * l, mu_leg[l], w->z + 5 * l);
  }
  free(as); free(prev); free(yfull);
  return ok;
}

static void solve_one(Work* w, const MpcQpConfig* cfg, const double* x0, const double* r, const uint8_t* contact,
                      const double* xdes, double mu, double* u_out, double* X_out, int32_t* status, int32_t* iters,
                      float* res) {
  const int N = w->N, n = w->n, nl = w->nleg;
  int finite = isfinite(mu);
  for (int i = 0; i < NX && finite; i++) finite = isfinite(x0[i]);
  for (int i = 0; i < N * 12 && finite; i++) finite = isfinite(r[i]);
  for (int i = 0; i < (N + 1) * NX && finite; i++) finite = isfinite(xdes[i]);
  if (!finite) {
    memset(u_out, 0, sizeof(double) * n);
    if (X_out) memset(X_out, 0, sizeof(double) * NX * (N + 1));
    *status = MPCQP_STATUS_NONFINITE; *iters = 0;
    if (res) { res[0] = res[1] = 0; }
    return;
  }
  build_qp(w, cfg, x0, r, contact, xdes);
  double* mu_leg = (double*)malloc(sizeof(double) * nl);
  double rho = cfg->rho;
  const double sigma = cfg->sigma, relax = cfg->relax, INF = 1e300;
  for (int l = 0; l < nl; l++) {
    mu_leg[l] = mu;
    double* lo = w->lo + 5 * l; double* hi = w->hi + 5 * l;
    if (contact[l]) { /* src/mpc.py:151-173 */
      lo[0] = cfg->f_min; hi[0] = cfg->f_max;
      lo[1] = -INF; hi[1] = 0; lo[2] = 0; hi[2] = INF; lo[3] = -INF; hi[3] = 0; lo[4] = 0; hi[4] = INF;
      w->rho[l] = rho;
    } else { /* src/mpc.py:139-144: swing * U == 0 */
      for (int i = 0; i < 5; i++) lo[i] = hi[i] = 0;
      w->rho[l] = 1e3 * rho;
    }
  }
  memset(w->u, 0, sizeof(double) * n);
  memset(w->z, 0, sizeof(double) * 5 * nl);
  memset(w->y, 0, sizeof(double) * 5 * nl);
  int st = MPCQP_STATUS_MAX_ITER, it = 0, psteps = 0;
  double rp = 0, rd = 0, sp = 1, sd = 1;
  if (factor_M(w, cfg, mu_leg, sigma) != 0) { st = MPCQP_STATUS_NONFINITE; goto done; }
  for (it = 1; it <= cfg->max_iter; it++) {
    for (int l = 0; l < nl; l++) {
      double v[5], gt[3];
      for (int i = 0; i < 5; i++) v[i] = w->rho[l] * w->z[5 * l + i] - w->y[5 * l + i];
      GT_leg(v, mu_leg[l], gt);
      for (int cidx = 0; cidx < 3; cidx++) w->rhs[3 * l + cidx] = sigma * w->u[3 * l + cidx] - w->g[3 * l + cidx] + gt[cidx];
    }
    memcpy(w->ut, w->rhs, sizeof(double) * n);
    chol_solve(w->M, n, w->ut);
    for (int l = 0; l < nl; l++) {
      double zt[5];
      G_leg(w->ut + 3 * l, mu_leg[l], zt);
      for (int cidx = 0; cidx < 3; cidx++) w->u[3 * l + cidx] = relax * w->ut[3 * l + cidx] + (1 - relax) * w->u[3 * l + cidx];
      for (int i = 0; i < 5; i++) {
        const double zr = relax * zt[i] + (1 - relax) * w->z[5 * l + i];
        double zn = zr + w->y[5 * l + i] / w->rho[l];
        zn = fmin(fmax(zn, w->lo[5 * l + i]), w->hi[5 * l + i]);
        w->y[5 * l + i] += w->rho[l] * (zr - zn);
        w->z[5 * l + i] = zn;
      }
    }
    if (it % cfg->check_every == 0 || it == cfg->max_iter) {
      residuals(w, mu_leg, &rp, &rd, &sp, &sd);
      if (!isfinite(rp) || !isfinite(rd)) { st = MPCQP_STATUS_NONFINITE; break; }
      if (rp <= cfg->eps_abs + cfg->eps_rel * sp && rd <= cfg->eps_abs + cfg->eps_rel * sd) { st = MPCQP_STATUS_SOLVED_ADMM; break; }
      if (it % 100 == 0) { /* OSQP rho adaptation */
        const double ratio = sqrt((rp / fmax(sp, 1e-12)) / fmax(rd / fmax(sd, 1e-12), 1e-30));
        if (ratio > 5.0 || ratio < 0.2) {
          rho = fmin(fmax(rho * ratio, 1e-6), 1e6);
          for (int l = 0; l < nl; l++) w->rho[l] = contact[l] ? rho : 1e3 * rho;
          if (factor_M(w, cfg, mu_leg, sigma) != 0) { st = MPCQP_STATUS_NONFINITE; break; }
        }
      }
    }
  }
  if (it > cfg->max_iter) it = cfg->max_iter;
  if ((cfg->flags & MPCQP_FLAG_POLISH) && st != MPCQP_STATUS_NONFINITE) {
    if (polish(w, cfg, contact, mu_leg, &psteps)) {
      st = MPCQP_STATUS_SOLVED_POLISHED;
      residuals(w, mu_leg, &rp, &rd, &sp, &sd);
    }
  }
done:
  for (int l = 0; l < nl; l++)
    if (!contact[l]) w->u[3 * l] = w->u[3 * l + 1] = w->u[3 * l + 2] = 0.0;
  if (st == MPCQP_STATUS_NONFINITE) memset(w->u, 0, sizeof(double) * n);
  memcpy(u_out, w->u, sizeof(double) * n);
  if (X_out) rollout(w, x0, w->u, X_out);
  *status = st; *iters = it + 1000 * psteps;
  if (res) { res[0] = (float)rp; res[1] = (float)rd; }
  free(mu_leg);
}

int mpcqp_solve_batch(mpcqp_handle h, int64_t B, const void* x0v, const void* rv, const uint8_t* contact,
                      const void* xdesv, const void* muv, void* uv, void* Xv, int32_t* status, int32_t* iters,
                      float* res, void* stream) {
  (void)stream;
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || (B > 0 && (!x0v || !rv || !contact || !xdesv || !muv || !uv || !status || !iters))) {
    snprintf(h->err, sizeof(h->err), "mpcqp_solve_batch: null buffer or negative batch");
    return MPCQP_EINVAL;
  }
  const MpcQpConfig* cfg = &h->cfg;
  const int N = cfg->N;
  const double *x0 = (const double*)x0v, *r = (const double*)rv, *xdes = (const double*)xdesv, *mu = (const double*)muv;
  double *u = (double*)uv, *X = (double*)Xv;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel
  {
    Work* w = work_alloc(N);
#pragma omp for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; b++)
      solve_one(w, cfg, x0 + b * NX, r + b * N * 12, contact + b * N * 4, xdes + b * (N + 1) * NX, mu[b],
                u + b * N * NU, X ? X + b * (N + 1) * NX : NULL, status + b, iters + b, res ? res + 2 * b : NULL);
    work_free(w);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  h->last_ms = (float)((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
  return MPCQP_OK;
}

/* Gait entry: expand the compact descriptors on the host exactly as MPC.solve does (src/mpc.py:178-254, planner queries
 * src/footstep_planner.py:226-246), then solve as above.  Literal loops; shares nothing with the device expansion. */
int mpcqp_solve_batch_gait_steps(mpcqp_handle h, int64_t B, int32_t S, const void* x0v, const void* refv, const void* feet0v,
                                 const void* footholdsv, const int32_t* gait, const uint8_t* feet_id, const void* muv, void* uv,
                                 void* Xv, int32_t* status, int32_t* iters, float* res, void* stream) {
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || S < 1 || (B > 0 && (!x0v || !refv || !feet0v || !footholdsv || !gait || !feet_id || !muv || !uv || !status || !iters))) {
    snprintf(h->err, sizeof(h->err), "mpcqp_solve_batch_gait: null buffer, negative batch or no plan step");
    return MPCQP_EINVAL;
  }
  const int N = h->cfg.N;
  const double d = h->cfg.delta;
  const double *x0 = (const double*)x0v, *ref = (const double*)refv, *feet0 = (const double*)feet0v, *fh = (const double*)footholdsv;
  double* r = (double*)malloc(sizeof(double) * (size_t)(B ? B : 1) * N * 12);
  double* xd = (double*)malloc(sizeof(double) * (size_t)(B ? B : 1) * (N + 1) * NX);
  uint8_t* ct = (uint8_t*)malloc((size_t)(B ? B : 1) * N * 4);
  if (!r || !xd || !ct) { free(r); free(xd); free(ct); return MPCQP_ENOMEM; }
  for (int64_t b = 0; b < B; b++) {
    const double* rf = ref + b * 10;
    const int tis = gait[b * 4 + 0] < 0 ? 0 : gait[b * 4 + 0], ss = gait[b * 4 + 1] < 0 ? 0 : gait[b * 4 + 1];
    const int per = (ss + (gait[b * 4 + 2] < 0 ? 0 : gait[b * 4 + 2])) < 1 ? 1 : ss + (gait[b * 4 + 2] < 0 ? 0 : gait[b * 4 + 2]);
    for (int k = 0; k <= N; k++) { /* src/mpc.py:202-214 */
      double* xk = xd + (b * (N + 1) + k) * NX;
      memset(xk, 0, sizeof(double) * NX);
      xk[0] = rf[0]; xk[1] = rf[1];
      xk[2] = rf[2] + k * d * rf[9];
      for (int a = 0; a < 3; a++) { xk[3 + a] = rf[3 + a] + k * d * rf[6 + a]; xk[9 + a] = rf[6 + a]; }
      xk[8] = rf[9];
      xk[12] = x0[b * NX + 12];
    }
    for (int k = 0; k < N; k++) {
      const int tau = tis + k;
      int st = tau / per; if (st > S - 1) st = S - 1;      /* past the descriptor's last step: that step, its time keeps running */
      const int tin = tau - st * per;                      /* (footstep_planner.py:226-237 clamps the same way) */
      for (int l = 0; l < 4; l++) {
        ct[(b * N + k) * 4 + l] = (tin < ss) ? (feet_id[((size_t)b * S + st) * 4 + l] ? 1 : 0) : 1;   /* footstep_planner.py:239-246 */
        for (int a = 0; a < 3; a++)                                                           /* src/mpc.py:218-239 */
          r[((b * N + k) * 4 + l) * 3 + a] = k == 0 ? feet0[b * 12 + l * 3 + a] - x0[b * NX + 3 + a]
                                                    : fh[(((size_t)b * S + st) * 4 + l) * 3 + a] - xd[(b * (N + 1) + k) * NX + 3 + a];
      }
    }
  }
  const int rc = mpcqp_solve_batch(h, B, x0v, r, ct, xd, muv, uv, Xv, status, iters, res, stream);
  free(r); free(xd); free(ct);
  return rc;
}

int mpcqp_solve_batch_gait(mpcqp_handle h, int64_t B, const void* x0v, const void* refv, const void* feet0v,
                           const void* footholdsv, const int32_t* gait, const uint8_t* feet_id, const void* muv, void* uv,
                           void* Xv, int32_t* status, int32_t* iters, float* res, void* stream) {
  return mpcqp_solve_batch_gait_steps(h, B, 2, x0v, refv, feet0v, footholdsv, gait, feet_id, muv, uv, Xv, status, iters, res, stream);
}

/* Closed-loop roll-out on host memory, fp64: the literal per-tick loop of Lite3Controller.customPreStep / MPC.solve
 * (src/main.py:130-188, src/mpc.py:176-271) with the world step replaced by the model's predicted next state X[:,1].
 * Checker for the product library's device roll-out (same argument layout, include/mpcqp.h). */
int mpcqp_rollout(mpcqp_handle h, int64_t B, int32_t T, int32_t S, void* xv, void* refv, const void* plan_posv, const uint8_t* plan_feet_id,
                  const int32_t* plan_meta, int32_t* tick, const void* muv, void* actualv, void* desiredv, void* forcesv, int32_t* solved,
                  void* stream) {
  if (!h) return MPCQP_EINVAL;
  if (B < 0 || T < 0 || S < 1 || (B > 0 && (!xv || !refv || !plan_posv || !plan_feet_id || !plan_meta || !tick || !muv))) return MPCQP_EINVAL;
  if (B == 0 || T == 0) return MPCQP_OK;
  const int N = h->cfg.N;
  const double d = h->cfg.delta;
  double *x = (double*)xv, *ref = (double*)refv, *actual = (double*)actualv, *desired = (double*)desiredv, *forces = (double*)forcesv;
  const double* pos = (const double*)plan_posv;
  double* r = (double*)malloc(sizeof(double) * (size_t)B * N * 12);
  double* xd = (double*)malloc(sizeof(double) * (size_t)B * (N + 1) * NX);
  double* u = (double*)malloc(sizeof(double) * (size_t)B * N * 12);
  double* X = (double*)malloc(sizeof(double) * (size_t)B * (N + 1) * NX);
  uint8_t* ct = (uint8_t*)malloc((size_t)B * N * 4);
  int32_t* st = (int32_t*)malloc(sizeof(int32_t) * (size_t)B * 2);
  int rc = (r && xd && u && X && ct && st) ? MPCQP_OK : MPCQP_ENOMEM;
  for (int it = 0; it < T && rc == MPCQP_OK; it++) {
    for (int64_t b = 0; b < B; b++) {
      const double* rf = ref + b * 10;
      /* malformed plan rows are clamped, never indexed with: 1 <= S_b <= S, ss >= 0, ss + ds >= 1, tick >= 0 (include/mpcqp.h) */
      const int Sb = plan_meta[b * 4] < 1 ? 1 : (plan_meta[b * 4] > S ? S : plan_meta[b * 4]);
      const int ss = plan_meta[b * 4 + 1] < 0 ? 0 : plan_meta[b * 4 + 1], per = (ss + (plan_meta[b * 4 + 2] < 0 ? 0 : plan_meta[b * 4 + 2])) < 1 ? 1 : ss + (plan_meta[b * 4 + 2] < 0 ? 0 : plan_meta[b * 4 + 2]);
      const int t0 = tick[b] < 0 ? 0 : tick[b];
      int step0 = t0 / per; if (step0 > Sb - 1) step0 = Sb - 1;
      const double gate = step0 == Sb - 1 ? 0.0 : 1.0;                                        /* src/mpc.py:181-183 */
      for (int k = 0; k <= N; k++) {                                                           /* src/mpc.py:202-214 */
        double* xk = xd + (b * (N + 1) + k) * NX;
        memset(xk, 0, sizeof(double) * NX);
        xk[0] = rf[0]; xk[1] = rf[1];
        xk[2] = rf[2] + k * d * gate * rf[9];
        for (int a = 0; a < 3; a++) { xk[3 + a] = rf[3 + a] + k * d * gate * rf[6 + a]; xk[9 + a] = gate * rf[6 + a]; }
        xk[8] = gate * rf[9];
        xk[12] = x[b * NX + 12];
      }
      for (int k = 0; k < N; k++) {
        const int tau = t0 + k;
        int si = tau / per; if (si > Sb - 1) si = Sb - 1;
        const int tin = tau - si * per;
        for (int l = 0; l < 4; l++) {
          ct[(b * N + k) * 4 + l] = (tin < ss) ? (plan_feet_id[(b * S + si) * 4 + l] ? 1 : 0) : 1;   /* footstep_planner.py:239-246 */
          for (int a = 0; a < 3; a++)                                                          /* src/mpc.py:218-239 */
            r[((b * N + k) * 4 + l) * 3 + a] = pos[((b * S + si) * 4 + l) * 3 + a] - (k == 0 ? x[b * NX + 3 + a] : xd[(b * (N + 1) + k) * NX + 3 + a]);
        }
      }
    }
    rc = mpcqp_solve_batch(h, B, x, r, ct, xd, muv, u, X, st, st + B, NULL, stream);
    if (rc != MPCQP_OK) break;
    for (int64_t b = 0; b < B; b++) {
      double* rf = ref + b * 10;
      const int Sb = plan_meta[b * 4] < 1 ? 1 : (plan_meta[b * 4] > S ? S : plan_meta[b * 4]);
      const int ss = plan_meta[b * 4 + 1] < 0 ? 0 : plan_meta[b * 4 + 1], per = (ss + (plan_meta[b * 4 + 2] < 0 ? 0 : plan_meta[b * 4 + 2])) < 1 ? 1 : ss + (plan_meta[b * 4 + 2] < 0 ? 0 : plan_meta[b * 4 + 2]);
      const int t0 = tick[b] < 0 ? 0 : tick[b];
      int step0 = t0 / per; if (step0 > Sb - 1) step0 = Sb - 1;
      const double gate = step0 == Sb - 1 ? 0.0 : 1.0;
      const size_t row = ((size_t)b * T + it) * 12;
      if (actual) for (int c = 0; c < 12; c++) actual[row + c] = x[b * NX + c];               /* src/mpc.py:295 */
      if (desired) {
        const double des[12] = {rf[0], rf[1], rf[2], rf[3], rf[4], rf[5], 0, 0, gate * rf[9], gate * rf[6], gate * rf[7], gate * rf[8]};
        for (int c = 0; c < 12; c++) desired[row + c] = des[c];
      }
      if (forces) for (int c = 0; c < 12; c++) forces[row + c] = u[(size_t)b * N * 12 + c];   /* src/main.py:216-218 */
      if (solved) solved[b] = (it == 0 ? 0 : solved[b]) + ((st[b] == MPCQP_STATUS_SOLVED_POLISHED || st[b] == MPCQP_STATUS_SOLVED_ADMM) ? 1 : 0);
      for (int c = 0; c < 12; c++) x[b * NX + c] = X[(b * (N + 1) + 1) * NX + c];
      for (int a = 0; a < 3; a++) rf[3 + a] += gate * rf[6 + a] * d;                          /* src/mpc.py:261 */
      rf[2] += gate * rf[9] * d;                                                              /* src/mpc.py:262 */
      tick[b] = tick[b] + 1;
    }
  }
  free(r); free(xd); free(u); free(X); free(ct); free(st);
  return rc;
}

/* lite3_urdf/urdf/Lite3.urdf:44-124: joint origins and axes (data). */
int mpcqp_default_leg_geometry(MpcQpLegGeometry* g) {
  if (!g) return MPCQP_EINVAL;
  memset(g, 0, sizeof(*g));
  g->size = (uint32_t)sizeof(*g);
  const double sx[4] = {1, 1, -1, -1}, sy[4] = {1, -1, 1, -1};
  for (int l = 0; l < 4; l++) { g->hip_x[l][0] = 0.1745 * sx[l]; g->hip_x[l][1] = 0.062 * sy[l]; g->hip_y[l][1] = 0.0985 * sy[l]; }
  g->knee[2] = -0.20; g->foot[2] = -0.21; g->axis_x[0] = -1.0; g->axis_y[1] = -1.0;
  return MPCQP_OK;
}

/* Foot position of one leg in the torso frame: the chain of homogeneous transforms torso -> HipX -> HipY -> Knee -> foot, each joint a
 * rotation exp(angle [axis]x) summed as its power series (checker: no closed form shared with the device kernel). */
static void rot_series(const double* axis, double ang, double R[9]) {
  const double n = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  const double a[3] = {axis[0] / n * ang, axis[1] / n * ang, axis[2] / n * ang};
  const double K[9] = {0, -a[2], a[1], a[2], 0, -a[0], -a[1], a[0], 0};
  double term[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, nxt[9];
  for (int i = 0; i < 9; i++) R[i] = term[i];
  for (int k = 1; k < 40; k++) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) nxt[3 * i + j] = (term[3 * i] * K[j] + term[3 * i + 1] * K[3 + j] + term[3 * i + 2] * K[6 + j]) / k;
    for (int i = 0; i < 9; i++) { term[i] = nxt[i]; R[i] += term[i]; }
  }
}
static void leg_fk(const MpcQpLegGeometry* g, int l, const double q[3], double pf[3]) {
  double R[3][9];
  rot_series(g->axis_x, q[0], R[0]); rot_series(g->axis_y, q[1], R[1]); rot_series(g->axis_y, q[2], R[2]);
  const double* off[3] = {g->hip_y[l], g->knee, g->foot};
  double v[3] = {0, 0, 0};
  for (int j = 2; j >= 0; j--) {   /* innermost link first: v <- R_j (off_j + v) */
    double t[3] = {off[j][0] + v[0], off[j][1] + v[1], off[j][2] + v[2]};
    for (int i = 0; i < 3; i++) v[i] = R[j][3 * i] * t[0] + R[j][3 * i + 1] * t[1] + R[j][3 * i + 2] * t[2];
  }
  for (int i = 0; i < 3; i++) pf[i] = g->hip_x[l][i] + v[i];
}

/* src/main.py:205-210: the world-frame linear Jacobian block of each foot w.r.t. its leg's joints.  Checker: central differences of
 * the forward kinematics with Richardson extrapolation (error ~ h^4), then rotated by the torso orientation. */
int mpcqp_leg_jacobians(mpcqp_handle h, int64_t B, const void* qv, const void* rotv, const MpcQpLegGeometry* geo, void* jacv, void* footv,
                        void* stream) {
  (void)stream;
  if (!h || B < 0 || (B > 0 && (!qv || !jacv))) return MPCQP_EINVAL;
  MpcQpLegGeometry lite3;
  if (!geo) { mpcqp_default_leg_geometry(&lite3); geo = &lite3; }
  if (geo->size != sizeof(MpcQpLegGeometry)) return MPCQP_EINVAL;
  const double *q = (const double*)qv, *rot = (const double*)rotv;
  double *jac = (double*)jacv, *foot = (double*)footv;
  for (int64_t i = 0; i < 4 * B; i++) {
    const int l = (int)(i % 4);
    double J[9], pf[3];
    leg_fk(geo, l, q + 3 * i, pf);
    for (int j = 0; j < 3; j++) {
      double d1[3], d2[3];
      for (int pass = 0; pass < 2; pass++) {
        const double hh = pass ? 2e-3 : 1e-3;
        double qp[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]}, qm[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]}, fp[3], fm[3];
        qp[j] += hh; qm[j] -= hh;
        leg_fk(geo, l, qp, fp); leg_fk(geo, l, qm, fm);
        for (int a = 0; a < 3; a++) (pass ? d2 : d1)[a] = (fp[a] - fm[a]) / (2 * hh);
      }
      for (int a = 0; a < 3; a++) J[3 * a + j] = (4 * d1[a] - d2[a]) / 3;
    }
    if (rot) {
      const double* Rb = rot + 9 * (i / 4);
      double Jw[9], pw[3];
      for (int a = 0; a < 3; a++) {
        for (int j = 0; j < 3; j++) Jw[3 * a + j] = Rb[3 * a] * J[j] + Rb[3 * a + 1] * J[3 + j] + Rb[3 * a + 2] * J[6 + j];
        pw[a] = Rb[3 * a] * pf[0] + Rb[3 * a + 1] * pf[1] + Rb[3 * a + 2] * pf[2];
      }
      memcpy(J, Jw, sizeof(J)); memcpy(pf, pw, sizeof(pf));
    }
    memcpy(jac + 9 * i, J, sizeof(J));
    if (foot) memcpy(foot + 3 * i, pf, sizeof(pf));
  }
  return MPCQP_OK;
}

/* src/main.py:212-214: tau[leg] = J[leg].T @ -forces[leg]; stage-0 forces only. */
int mpcqp_torque_map(mpcqp_handle h, int64_t B, const void* uv, const void* jacv, void* tauv, void* stream) {
  (void)stream;
  if (!h || B < 0 || (B > 0 && (!uv || !jacv || !tauv))) return MPCQP_EINVAL;
  const double *u = (const double*)uv, *jac = (const double*)jacv;
  double* tau = (double*)tauv;
  const int N = h->cfg.N;
  for (int64_t b = 0; b < B; b++)
    for (int l = 0; l < 4; l++) {
      const double* f = u + b * N * NU + 3 * l;
      const double* J = jac + (b * 4 + l) * 9;
      for (int q = 0; q < 3; q++) {
        double s = 0;
        for (int a = 0; a < 3; a++) s += J[a * 3 + q] * -f[a];
        tau[(b * 4 + l) * 3 + q] = s;
      }
    }
  return MPCQP_OK;
}

