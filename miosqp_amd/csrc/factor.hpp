// Host-side (CPU, one-time) preparation of the relaxation engine: Ruiz equilibration and the
// block LDL^T factor of the quasi-definite KKT matrix.  See DESIGN.md "Factor layout".
//
// Stands in for what osqp.OSQP().setup() does once per MIQP
// (/root/reference/miosqp/workspace.py:63-68).  With the elimination order "constraint rows
// first, variables last" (what a minimum-degree ordering yields on these problems, SURVEY.md
// sec. 0.5) the factor  K = L D L^T  of
//
//        K = [ -1/rho I    A  ]        L = [  I     0  ]     D = [ -1/rho I   0  ]
//            [   A^T   P+sigma I ]         [ L21   L22 ]         [    0      D22 ]
//
// is   L21 = -rho A^T  (sparse, pattern of A^T: the "panel"),
//      L22 D22 L22^T = P + sigma I + rho A^T A  (dense trailing triangle: the "tail").
// The tail is kept as its pre-inverted triangular factor  Linv = L22^-1  (unit lower, same
// size and sparsity as L22) so that both triangular sweeps over it are row-parallel.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace miosqp {

// Row-major compressed rows, every row padded to an even number of entries (value 0, a valid
// column) so that a lane can fetch two entries with one 16-byte + one 8-byte load.
struct PCsr {
  int rows = 0, cols = 0;
  std::vector<int> ptr;   // rows + 1, all even
  std::vector<int> idx;
  std::vector<double> val;
  int64_t nnz = 0;  // true (unpadded) entry count
};

struct Scaled {
  int n = 0, M = 0;
  std::vector<double> D, E, Dinv, Einv;
  double c = 1.0, cinv = 1.0;
  // scaled data
  std::vector<int> Pp, Pi;  // upper triangle, CSC
  std::vector<double> Px;
  std::vector<int> Ap, Ai;  // CSC
  std::vector<double> Ax;
  std::vector<double> q;  // scaled linear cost
};

struct Factor {
  int n = 0, M = 0, ld = 0;
  double rho = 0, sigma = 0;
  // panel L21, stored twice: by tail row (variable i: entries over constraints j) for the
  // forward sweep, by head row (constraint j: entries over variables i) for the backward sweep.
  PCsr panel_by_var;  // n rows x M cols, values L21[i][j] = -rho * Abar[j][i]
  PCsr panel_by_con;  // M rows x n cols, values L21[i][j] seen from row j
  // same patterns with the plain scaled matrix values (residual checks, warm start)
  std::vector<double> At_val;  // aligned with panel_by_var.idx   (Abar^T rows)
  std::vector<double> A_val;   // aligned with panel_by_con.idx   (Abar rows)
  // tail
  std::vector<double> Linv;   // n x ld row-major, strict lower part of L22^-1 (diag = 1 implied)
  std::vector<double> LinvT;  // n x ld row-major, strict upper part = Linv^T
  std::vector<double> d2inv;  // 1 / D22
  // symmetric matrices by row
  PCsr Pbar;  // scaled P, full symmetric
  PCsr Praw;  // unscaled P, full symmetric (node objective, data.py:99-103)
  int64_t nnz_panel = 0, nnz_tail = 0;
};

// Product form of the factor ("folded"): L^-1 = [ I 0 ; -G  Linv ] with G = L22^-1 L21 dense.
// Chosen when the panel is dense enough that G costs no more bytes than the two sparse copies
// of L21 (12 B/entry at ~70 % density vs 8 B/entry dense): both triangular sweeps then need ONE
// row-parallel kernel each instead of two (panel + tail), halving the launches per iteration.
struct Folded {
  int n = 0, M = 0, ldf = 0, ldn = 0, ldm = 0;
  // dense copies of the scaled matrices for the batched termination test (same tile kernels)
  std::vector<double> Ad;   // M x ldn : Abar
  std::vector<double> Atd;  // n x ldm : Abar^T
  std::vector<double> Pd;   // n x ldn : Pbar (full symmetric)
  std::vector<double> rows;  // n x ldf row-major: row i = [ -G[i][0..M) | Linv[i][0..i) | 0.. ]
  std::vector<double> GmT;   // M x ldn row-major: GmT[j][i] = -G[i][j]
};
void build_folded(const Factor &f, Folded &out);

// Optional accelerator for the equilibration: the three array-sized operations of a pass
// on device-resident copies of the matrices.  They are maxima and element-wise products only, so the result is
// bitwise the host's; everything scalar (square roots, clamps, the mean of the norms -- a serial sum --, q, D, E, c)
// stays in scale_problem.  Every function returns 0 on success.
struct RuizOps {
  void *ctx = nullptr;
  // upload (upper triangle of P and A, CSC)
  int (*begin)(void *ctx, int n, int M, const int *Pp, const int *Pi, const double *Px, const int *Ap, const int *Ai,
               const double *Ax) = nullptr;
  // dt[j] = max(|P(:, j)| as a symmetric matrix, with_A ? |A(:, j)| : 0); et[i] = |A(i, :)| (with_A only)
  int (*norms)(void *ctx, double *dt, double *et, int with_A) = nullptr;
  // Px[p] *= dt[col] * dt[row];  Ax[p] *= dt[col] * et[row]
  int (*scale)(void *ctx, const double *dt, const double *et) = nullptr;
  // Px[p] *= ct
  int (*scale_cost)(void *ctx, double ct) = nullptr;
  // download the scaled values; releases the device copies (also to be called after an error)
  int (*end)(void *ctx, double *Px, double *Ax) = nullptr;
};

// Ruiz equilibration + cost normalisation (OSQP paper sec. 5.1).  P: CSC, only row<=col read.  Returns false when
// `ops` failed (the caller repeats the call without it); always true without `ops`.
bool scale_problem(int n, int M, const int32_t *Pp, const int32_t *Pi, const double *Px,
                   const int32_t *Ap, const int32_t *Ai, const double *Ax, const double *q,
                   int passes, Scaled &out, const RuizOps *ops = nullptr);

// Optional accelerator for the dense part of the setup: given the
// reduced Hessian S (n x ld row-major, lower triangle valid) it must produce d (D22), Linv
// (strict lower of L22^-1, n x ld) and LinvT (its transpose); returns 0 on success, 1 when a
// pivot is not positive, <0 on a device error.  The engine passes its HIP implementation here.
typedef int (*DenseLdlInv)(int n, int ld, const double *S, double *d, double *Linv, double *LinvT, void *ctx);

// What `accel_ctx` points to.  keep_on_device: the accelerator leaves Linv / LinvT where it computed them (device
// pointers returned in dLinv / dLinvT, each n * ld + 64 doubles, owned by the caller) and the host copies of the
// Factor stay empty -- for an engine that only ever reads them on the device (no product form built on the host):
// at n = 5000 the round trip is 400 MB down, 400 MB of zero fill and 400 MB up again.
struct DenseAccelCtx {
  void *stream = nullptr;
  int keep_on_device = 0;
  double *dLinv = nullptr, *dLinvT = nullptr;
  // schur_on_device: the accelerator is called with S == nullptr and assembles S = Pbar + sigma I + rho Abar^T Abar
  // itself from the arrays below (host pointers, filled in by build_factor), with the host loop's order of
  // additions per entry (constraint rows ascending, fused multiply-add): bitwise the same S.  Saves the 200 MB host
  // array, its assembly and its upload at n = 5000.
  int schur_on_device = 0;
  double rho = 0, sigma = 0;
  const int *Pp = nullptr, *Pi = nullptr;  // scaled P, upper triangle, CSC (row indices strictly ascending per column)
  const double *Px = nullptr;
  const int *Ap = nullptr, *Ai = nullptr;  // scaled A, CSC
  const double *Ax = nullptr;
  const int *Rptr = nullptr, *Ridx = nullptr;  // rows of Abar (padded rows, see PCsr) ...
  const double *Rval = nullptr;                // ... with the values of Abar
  int64_t nnzP = 0, nnzA = 0, nnzR = 0;        // array lengths
  int M = 0;
};

// Builds the factor; returns false with `err` set when D22 loses positivity.
// reuse_rows: `f` holds the factor of the SAME matrices at another rho (rho chosen at set-up): the packed rows -- patterns,
// the plain values of Abar / Abar^T, the symmetric matrices -- are kept, the panel's values are formed from the plain
// ones with the new rho (-rho a: the same product a fresh build forms) and only the dense part is built again.
bool build_factor(const Scaled &s, const int32_t *Pp_raw, const int32_t *Pi_raw,
                  const double *Px_raw, double rho, double sigma, Factor &f, std::string &err,
                  DenseLdlInv accel = nullptr, void *accel_ctx = nullptr, bool reuse_rows = false);

}  // namespace miosqp
