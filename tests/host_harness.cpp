// Test-only C wrapper around miosqp_amd/csrc/factor.cpp so that the host-side setup logic
// (equilibration, block LDL^T, pre-inverted tail, padded row layout) can be checked on a CPU
// against numpy.  apply_kinv() performs, in plain loops, the same four row-wise stages the HIP
// kernels perform on the device (engine.hip: k_panel_fwd, k_tail_fwd, k_tail_bwd, k_panel_bwd).
#include <cstring>
#include <string>

#include "../miosqp_amd/csrc/factor.hpp"

using namespace miosqp;

struct Harness {
  Scaled sc;
  Factor fa;
};

static double prow(const PCsr &m, const std::vector<double> &val, int row, const double *v) {
  double s = 0;
  for (int k = m.ptr[row]; k < m.ptr[row + 1]; k++) s += val[k] * v[m.idx[k]];
  return s;
}

extern "C" {

void *hh_build(int n, int M, const int *Pp, const int *Pi, const double *Px, const int *Ap, const int *Ai,
               const double *Ax, const double *q, int passes, double rho, double sigma) {
  Harness *h = new Harness();
  scale_problem(n, M, Pp, Pi, Px, Ap, Ai, Ax, q, passes, h->sc);
  std::string err;
  if (!build_factor(h->sc, Pp, Pi, Px, rho, sigma, h->fa, err)) {
    delete h;
    return nullptr;
  }
  return h;
}

void hh_free(void *p) { delete (Harness *)p; }

void hh_scaling(void *p, double *D, double *E, double *c, double *qs) {
  Harness *h = (Harness *)p;
  memcpy(D, h->sc.D.data(), sizeof(double) * h->sc.n);
  memcpy(E, h->sc.E.data(), sizeof(double) * h->sc.M);
  memcpy(qs, h->sc.q.data(), sizeof(double) * h->sc.n);
  *c = h->sc.c;
}

// dense copies for inspection: Linv (n x n, strict lower), d2inv (n)
void hh_tail(void *p, double *Linv, double *LinvT, double *d2inv) {
  Harness *h = (Harness *)p;
  int n = h->fa.n, ld = h->fa.ld;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      Linv[(size_t)i * n + j] = h->fa.Linv[(size_t)i * ld + j];
      LinvT[(size_t)i * n + j] = h->fa.LinvT[(size_t)i * ld + j];
    }
  memcpy(d2inv, h->fa.d2inv.data(), sizeof(double) * n);
}

// rows padded to even length, ptr even, padding value zero: returns 1 if the layout holds
int hh_layout_ok(void *p) {
  Harness *h = (Harness *)p;
  const PCsr *ms[4] = {&h->fa.panel_by_var, &h->fa.panel_by_con, &h->fa.Pbar, &h->fa.Praw};
  for (const PCsr *m : ms) {
    for (int r = 0; r <= m->rows; r++)
      if (m->ptr[r] & 1) return 0;
    for (int r = 0; r < m->rows; r++)
      for (int k = m->ptr[r]; k < m->ptr[r + 1]; k++)
        if (m->idx[k] < 0 || m->idx[k] >= m->cols) return 0;
  }
  return 1;
}

long hh_nnz_panel(void *p) { return (long)((Harness *)p)->fa.nnz_panel; }

// [xt; nu] = K^-1 [rx; rz] with K = [Pbar + sigma I, Abar'; Abar, -1/rho I]
void hh_apply_kinv(void *p, const double *rx, const double *rz, double *xt, double *nu) {
  Harness *h = (Harness *)p;
  const Factor &f = h->fa;
  int n = f.n, M = f.M, ld = f.ld;
  std::vector<double> c(n), u(n);
  for (int i = 0; i < n; i++) c[i] = rx[i] - prow(f.panel_by_var, f.panel_by_var.val, i, rz);
  for (int i = 0; i < n; i++) {
    double s = c[i];
    for (int j = 0; j < i; j++) s += f.Linv[(size_t)i * ld + j] * c[j];
    u[i] = f.d2inv[i] * s;
  }
  for (int i = 0; i < n; i++) {
    double s = u[i];
    for (int j = i + 1; j < n; j++) s += f.LinvT[(size_t)i * ld + j] * u[j];
    xt[i] = s;
  }
  for (int j = 0; j < M; j++) nu[j] = -f.rho * rz[j] - prow(f.panel_by_con, f.panel_by_con.val, j, xt);
}

// the same K^-1 through the product-form factor (engine.hip: k_fold_fwd, k_fold_bwd)
void hh_apply_kinv_folded(void *p, const double *rx, const double *rz, double *xt, double *nu) {
  Harness *h = (Harness *)p;
  const Factor &f = h->fa;
  Folded fo;
  build_folded(f, fo);
  int n = f.n, M = f.M;
  std::vector<double> b(M + n), u(n);
  for (int j = 0; j < M; j++) b[j] = rz[j];
  for (int i = 0; i < n; i++) b[M + i] = rx[i];
  for (int i = 0; i < n; i++) {
    double s = rx[i];
    for (int k = 0; k < M + i; k++) s += fo.rows[(size_t)i * fo.ldf + k] * b[k];
    u[i] = f.d2inv[i] * s;
  }
  for (int i = 0; i < n; i++) {
    double s = u[i];
    for (int j = i + 1; j < n; j++) s += f.LinvT[(size_t)i * f.ld + j] * u[j];
    xt[i] = s;
  }
  for (int j = 0; j < M; j++) {
    double s = -f.rho * rz[j];
    for (int i = 0; i < n; i++) s += fo.GmT[(size_t)j * fo.ldn + i] * u[i];
    nu[j] = s;
  }
}

// y = Abar x via the by-constraint rows, w = Abar' v via the by-variable rows, t = Pbar x
void hh_products(void *p, const double *x, const double *v, double *Ax, double *Atv, double *Px, double *Praw_x) {
  Harness *h = (Harness *)p;
  const Factor &f = h->fa;
  for (int j = 0; j < f.M; j++) Ax[j] = prow(f.panel_by_con, f.A_val, j, x);
  for (int i = 0; i < f.n; i++) Atv[i] = prow(f.panel_by_var, f.At_val, i, v);
  for (int i = 0; i < f.n; i++) Px[i] = prow(f.Pbar, f.Pbar.val, i, x);
  for (int i = 0; i < f.n; i++) Praw_x[i] = prow(f.Praw, f.Praw.val, i, x);
}
}
