// MI355X (gfx950) relaxation engine: device kernels + host driver + C ABI (include/miosqp_amd.h).
//
// One ADMM iteration of the OSQP algorithm (paper Algorithm 1) = one application of K^-1
// through the block factor of factor.hpp plus the fused vector updates, as FOUR row-parallel
// kernels (each row = one CSR / triangular row handled by a wavefront group: lanes stride the
// row with coalesced loads, partial sums meet in a DPP butterfly and, for multi-wave rows, in
// LDS).  Every reduction has a fixed order, so reruns are bit-identical.
//
//   k_panel_fwd   c  = sigma x - q - L21 wh                    (n rows, pattern of A^T)
//   k_tail_fwd    ut = D22^-1 (c + strict_lower(Linv) c)       (n rows, triangular)
//   k_tail_bwd    xt = ut + strict_upper(Linv^T) ut ;  x, dx   (n rows, triangular)
//   k_panel_bwd   nu = -rho wh - L21^T xt ; z~, z, y, dy, wh   (M rows, pattern of A)
//
// with wh = z - y/rho (the permuted right-hand side of the constraint block).  The termination
// test (every check_termination iterations) is three more kernels and one device-side decision;
// the host only reads a 64-byte control block per chunk.  The per-chunk kernel sequence is
// captured once in a hipGraph.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/miosqp_amd.h"
#include "factor.hpp"

#define QP_INFTY 1e30
#define QP_MIN_SCALING 1e-4
#define QP_DIVISION_TOL (1.0 / QP_INFTY)

namespace {

thread_local std::string g_err;

void set_err(const char *what, hipError_t e, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  g_err = buf;
}

#define HIPCHK(call)                                  \
  do {                                                \
    hipError_t e__ = (call);                          \
    if (e__ != hipSuccess) {                          \
      set_err(#call, e__, __FILE__, __LINE__);        \
      return MIOSQP_EHIP;                             \
    }                                                 \
  } while (0)

struct Ctrl {
  int done, status, iter, pad;
  double pri_res, dua_res, obj_val, lower;
  double nrm_dy, nrm_dx;  // certificate normalisers
  double pad2[2];
};

// everything a kernel needs, passed by value
struct Dev {
  int n, M, ld, n_int, m_orig;
  double rho, sigma, alpha, eps_abs, eps_rel, eps_pinf, eps_dinf, c, cinv;
  // panel by variable (n rows) / by constraint (M rows)
  const int *pv_ptr, *pv_idx;
  const double *pv_L, *pv_At;
  const int *pc_ptr, *pc_idx;
  const double *pc_L, *pc_A;
  // tail
  const double *Linv, *LinvT, *d2inv;
  // symmetric matrices by row
  const int *pb_ptr, *pb_idx;
  const double *pb_val;
  const int *pr_ptr, *pr_idx;
  const double *pr_val;
  const double *D, *Dinv, *E, *Einv;
  const int *i_idx;
  // scaled vectors
  double *q, *l, *u, *x, *z, *y, *wh, *cv, *ut, *xt, *dx, *dy;
  double *qraw;
  // scratch for the termination test: sm = 8 x M, sn = 10 x n
  double *sm, *sn;
  // staging: raw (unscaled) inputs and outputs
  double *raw_l, *raw_u, *raw_x, *raw_y, *out_x, *out_y;
  Ctrl *ctrl;
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
template <int TPR, int NV>
__device__ __forceinline__ void row_reduce(double (&v)[NV], double *lds) {
  constexpr int W = TPR < 64 ? TPR : 64;
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] += __shfl_xor(v[k], off, 64);
  }
  if constexpr (TPR > 64) {
    constexpr int WPR = TPR / 64;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < NV; k++) lds[k * 4 + wave] = v[k];
    }
    __syncthreads();
    const int base = (wave / WPR) * WPR;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double s = lds[k * 4 + base];
#pragma unroll
      for (int w = 1; w < WPR; w++) s += lds[k * 4 + base + w];
      v[k] = s;
    }
  }
}

// sum_k val[k] * v[idx[k]] over one padded row, this thread's share (two entries per step)
template <int TPR>
__device__ __forceinline__ double prow_dot(const int *__restrict__ idx, const double *__restrict__ val,
                                           int s, int e, int t, const double *__restrict__ v) {
  double a0 = 0.0, a1 = 0.0;
  for (int k = s + 2 * t; k < e; k += 2 * TPR) {
    const double2 a = *reinterpret_cast<const double2 *>(val + k);
    const int2 j = *reinterpret_cast<const int2 *>(idx + k);
    a0 = fma(a.x, v[j.x], a0);
    a1 = fma(a.y, v[j.y], a1);
  }
  return a0 + a1;
}

// two right-hand vectors at once (matrix read once)
template <int TPR>
__device__ __forceinline__ void prow_dot2(const int *__restrict__ idx, const double *__restrict__ val,
                                          int s, int e, int t, const double *__restrict__ v,
                                          const double *__restrict__ w, double &rv, double &rw) {
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  for (int k = s + 2 * t; k < e; k += 2 * TPR) {
    const double2 a = *reinterpret_cast<const double2 *>(val + k);
    const int2 j = *reinterpret_cast<const int2 *>(idx + k);
    a0 = fma(a.x, v[j.x], a0);
    a1 = fma(a.y, v[j.y], a1);
    b0 = fma(a.x, w[j.x], b0);
    b1 = fma(a.y, w[j.y], b1);
  }
  rv = a0 + a1;
  rw = b0 + b1;
}

#define ROW_SETUP(TPR)                                          \
  __shared__ double lds[16];                                    \
  constexpr int RPB = 256 / TPR;                                \
  const int row_raw = blockIdx.x * RPB + threadIdx.x / TPR;     \
  const int t = threadIdx.x % TPR;

// ------------------------------------------------------------------------------------------
// the four kernels of one ADMM iteration
// ------------------------------------------------------------------------------------------
template <int TPR>
__global__ __launch_bounds__(256) void k_panel_fwd(Dev d) {
  if (d.ctrl->done) return;
  ROW_SETUP(TPR)
  const bool live = row_raw < d.n;
  const int row = live ? row_raw : d.n - 1;
  double acc[1];
  acc[0] = prow_dot<TPR>(d.pv_idx, d.pv_L, d.pv_ptr[row], d.pv_ptr[row + 1], t, d.wh);
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) d.cv[row] = d.sigma * d.x[row] - d.q[row] - acc[0];
}

template <int TPR>
__global__ __launch_bounds__(256) void k_tail_fwd(Dev d) {
  if (d.ctrl->done) return;
  ROW_SETUP(TPR)
  const bool live = row_raw < d.n;
  const int row = live ? row_raw : d.n - 1;
  const double *__restrict__ Lr = d.Linv + (size_t)row * d.ld;
  const double *__restrict__ c = d.cv;
  double a0 = 0.0, a1 = 0.0;
  int j = t;
  for (; j + TPR < row; j += 2 * TPR) {
    a0 = fma(Lr[j], c[j], a0);
    a1 = fma(Lr[j + TPR], c[j + TPR], a1);
  }
  if (j < row) a0 = fma(Lr[j], c[j], a0);
  double acc[1] = {a0 + a1};
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) d.ut[row] = d.d2inv[row] * (c[row] + acc[0]);
}

template <int TPR>
__global__ __launch_bounds__(256) void k_tail_bwd(Dev d) {
  if (d.ctrl->done) return;
  ROW_SETUP(TPR)
  const bool live = row_raw < d.n;
  const int row = live ? row_raw : d.n - 1;
  const double *__restrict__ Ur = d.LinvT + (size_t)row * d.ld;
  const double *__restrict__ u = d.ut;
  const int n = d.n;
  double a0 = 0.0, a1 = 0.0;
  int j = row + 1 + t;
  for (; j + TPR < n; j += 2 * TPR) {
    a0 = fma(Ur[j], u[j], a0);
    a1 = fma(Ur[j + TPR], u[j + TPR], a1);
  }
  if (j < n) a0 = fma(Ur[j], u[j], a0);
  double acc[1] = {a0 + a1};
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) {
    const double xt = u[row] + acc[0];
    const double xp = d.x[row];
    const double xn = d.alpha * xt + (1.0 - d.alpha) * xp;
    d.xt[row] = xt;
    d.x[row] = xn;
    d.dx[row] = xn - xp;
  }
}

template <int TPR>
__global__ __launch_bounds__(256) void k_panel_bwd(Dev d) {
  if (d.ctrl->done) return;
  ROW_SETUP(TPR)
  const bool live = row_raw < d.M;
  const int row = live ? row_raw : d.M - 1;
  double acc[1];
  acc[0] = prow_dot<TPR>(d.pc_idx, d.pc_L, d.pc_ptr[row], d.pc_ptr[row + 1], t, d.xt);
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) {
    const double rho = d.rho, alpha = d.alpha;
    const double zp = d.z[row], yp = d.y[row];
    const double nu = -rho * d.wh[row] - acc[0];
    const double zt = zp + (nu - yp) / rho;
    const double zr = alpha * zt + (1.0 - alpha) * zp;
    const double v = zr + yp / rho;
    const double zn = fmin(fmax(v, d.l[row]), d.u[row]);
    const double dy = rho * (zr - zn);
    const double yn = yp + dy;
    d.z[row] = zn;
    d.y[row] = yn;
    d.dy[row] = dy;
    d.wh[row] = zn - yn / rho;
  }
}

// ------------------------------------------------------------------------------------------
// termination test (OSQP paper sec. 3.4), every check_termination iterations
// ------------------------------------------------------------------------------------------
// rows of Abar: residual pieces and certificate pieces per constraint
template <int TPR>
__global__ __launch_bounds__(256) void k_check_con(Dev d) {
  if (d.ctrl->done) return;
  ROW_SETUP(TPR)
  const bool live = row_raw < d.M;
  const int row = live ? row_raw : d.M - 1;
  double acc[2];
  prow_dot2<TPR>(d.pc_idx, d.pc_A, d.pc_ptr[row], d.pc_ptr[row + 1], t, d.x, d.dx, acc[0], acc[1]);
  row_reduce<TPR, 2>(acc, lds);
  if (live && t == 0) {
    const int M = d.M;
    const double ei = d.Einv[row], z = d.z[row], l = d.l[row], u = d.u[row];
    const bool uinf = u > QP_INFTY * QP_MIN_SCALING, linf = l < -QP_INFTY * QP_MIN_SCALING;
    double v = d.dy[row];
    if (uinf && linf) v = 0.0;
    else if (uinf) v = fmin(v, 0.0);
    else if (linf) v = fmax(v, 0.0);
    const double adx = ei * acc[1];
    d.sm[0 * M + row] = ei * (acc[0] - z);  // primal residual
    d.sm[1 * M + row] = ei * acc[0];        // |A x|
    d.sm[2 * M + row] = ei * z;             // |z|
    d.sm[3 * M + row] = v;                  // projected delta_y
    d.sm[4 * M + row] = d.E[row] * v;       // its unscaled size
    d.sm[5 * M + row] = u * fmax(v, 0.0) + l * fmin(v, 0.0);
    d.sm[6 * M + row] = uinf ? -1.7e308 : adx;  // A dx on rows with a finite upper bound
    d.sm[7 * M + row] = linf ? 1.7e308 : adx;   // ... finite lower bound
  }
}

// rows of Pbar and of Abar^T: dual residual pieces and certificate pieces per variable
template <int TPR_P, int TPR_A>
__global__ __launch_bounds__(256) void k_check_var(Dev d) {
  if (d.ctrl->done) return;
  __shared__ double lds[16];
  const int n = d.n;
  // pass 1: P rows
  {
    constexpr int RPB = 256 / TPR_P;
    const int nblk = (n + RPB - 1) / RPB;
    if ((int)blockIdx.x < nblk) {
      const int row_raw = blockIdx.x * RPB + threadIdx.x / TPR_P;
      const int t = threadIdx.x % TPR_P;
      const bool live = row_raw < n;
      const int row = live ? row_raw : n - 1;
      double acc[2];
      prow_dot2<TPR_P>(d.pb_idx, d.pb_val, d.pb_ptr[row], d.pb_ptr[row + 1], t, d.x, d.dx, acc[0], acc[1]);
      row_reduce<TPR_P, 2>(acc, lds);
      if (live && t == 0) {
        d.sn[0 * n + row] = acc[0];                // P x
        d.sn[1 * n + row] = d.Dinv[row] * acc[1];  // P dx, unscaled
      }
      return;
    }
  }
  // pass 2: A^T rows (blocks after the P blocks)
  {
    constexpr int RPB = 256 / TPR_A;
    const int nblkP = (n + (256 / TPR_P) - 1) / (256 / TPR_P);
    const int b = blockIdx.x - nblkP;
    const int row_raw = b * RPB + threadIdx.x / TPR_A;
    const int t = threadIdx.x % TPR_A;
    const bool live = row_raw < n;
    const int row = live ? row_raw : n - 1;
    double acc[2];
    prow_dot2<TPR_A>(d.pv_idx, d.pv_At, d.pv_ptr[row], d.pv_ptr[row + 1], t, d.y, d.sm + 3 * (size_t)d.M,
                     acc[0], acc[1]);
    row_reduce<TPR_A, 2>(acc, lds);
    if (live && t == 0) {
      d.sn[2 * n + row] = acc[0];                // A' y
      d.sn[3 * n + row] = d.Dinv[row] * acc[1];  // A' v, unscaled
    }
  }
}

// fixed-order block reductions (1024 threads)
__device__ __forceinline__ double block_max(double v, double *lds) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = lds[0];
  for (int w = 1; w < 16; w++) r = fmax(r, lds[w]);
  return r;
}
__device__ __forceinline__ double block_sum(double v, double *lds) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = lds[0];
  for (int w = 1; w < 16; w++) r += lds[w];
  return r;
}

// one workgroup: all norms, all sums, the decision
__global__ __launch_bounds__(1024) void k_check_decide(Dev d, int iters_in_chunk) {
  if (d.ctrl->done) return;
  __shared__ double lds[16];
  const int n = d.n, M = d.M, tid = threadIdx.x;
  double pri = 0, nAx = 0, nz = 0, nEv = 0, lhs = 0, amax_u = -1.7e308, amin_l = 1.7e308;
  for (int j = tid; j < M; j += 1024) {
    pri = fmax(pri, fabs(d.sm[0 * M + j]));
    nAx = fmax(nAx, fabs(d.sm[1 * M + j]));
    nz = fmax(nz, fabs(d.sm[2 * M + j]));
    nEv = fmax(nEv, fabs(d.sm[4 * M + j]));
    lhs += d.sm[5 * M + j];
    amax_u = fmax(amax_u, d.sm[6 * M + j]);
    amin_l = fmin(amin_l, d.sm[7 * M + j]);
  }
  double dua = 0, nPx = 0, nAty = 0, nq = 0, nPdx = 0, nAtv = 0, ndx = 0, qdx = 0, xPx = 0, qx = 0;
  for (int i = tid; i < n; i += 1024) {
    const double di = d.Dinv[i], px = d.sn[0 * n + i], aty = d.sn[2 * n + i], q = d.q[i], x = d.x[i],
                 dx = d.dx[i];
    dua = fmax(dua, fabs(di * (px + q + aty)));
    nPx = fmax(nPx, fabs(di * px));
    nAty = fmax(nAty, fabs(di * aty));
    nq = fmax(nq, fabs(di * q));
    nPdx = fmax(nPdx, fabs(d.sn[1 * n + i]));
    nAtv = fmax(nAtv, fabs(d.sn[3 * n + i]));
    ndx = fmax(ndx, fabs(d.D[i] * dx));
    qdx += q * dx;
    xPx += x * px;
    qx += q * x;
  }
  pri = block_max(pri, lds);
  nAx = block_max(nAx, lds);
  nz = block_max(nz, lds);
  nEv = block_max(nEv, lds);
  lhs = block_sum(lhs, lds);
  amax_u = block_max(amax_u, lds);
  amin_l = -block_max(-amin_l, lds);
  dua = block_max(dua, lds) * d.cinv;
  nPx = block_max(nPx, lds);
  nAty = block_max(nAty, lds);
  nq = block_max(nq, lds);
  nPdx = block_max(nPdx, lds);
  nAtv = block_max(nAtv, lds);
  ndx = block_max(ndx, lds);
  qdx = block_sum(qdx, lds);
  xPx = block_sum(xPx, lds);
  qx = block_sum(qx, lds);
  if (tid != 0) return;
  Ctrl *c = d.ctrl;
  c->iter += iters_in_chunk;
  c->pri_res = pri;
  c->dua_res = dua;
  c->obj_val = d.cinv * (0.5 * xPx + qx);
  const double eps_pri = d.eps_abs + d.eps_rel * fmax(nAx, nz);
  const double eps_dua = d.eps_abs + d.eps_rel * d.cinv * fmax(fmax(nPx, nAty), nq);
  const bool pri_ok = (M == 0) || (pri < eps_pri);
  const bool dua_ok = dua < eps_dua;
  bool pinf = false, dinf = false;
  if (!pri_ok && nEv > QP_DIVISION_TOL && lhs < -d.eps_pinf * nEv) pinf = nAtv < d.eps_pinf * nEv;
  if (!dua_ok && ndx > QP_DIVISION_TOL && qdx < -d.c * d.eps_dinf * ndx && nPdx < d.c * d.eps_dinf * ndx)
    dinf = !(amax_u > d.eps_dinf * ndx) && !(amin_l < -d.eps_dinf * ndx);
  int st = 0;
  if (pri_ok && dua_ok) st = MIOSQP_QP_SOLVED;
  else if (pinf) { st = MIOSQP_QP_PRIMAL_INFEASIBLE; c->obj_val = QP_INFTY; }
  else if (dinf) { st = MIOSQP_QP_DUAL_INFEASIBLE; c->obj_val = -QP_INFTY; }
  if (st) {
    c->status = st;
    c->done = 1;
  }
}

// ------------------------------------------------------------------------------------------
// per-solve prologue / epilogue
// ------------------------------------------------------------------------------------------
__global__ void k_reset_ctrl(Dev d) {
  Ctrl *c = d.ctrl;
  c->done = 0;
  c->status = MIOSQP_QP_UNSOLVED;
  c->iter = 0;
  c->pri_res = c->dua_res = c->obj_val = 0.0;
  c->lower = __builtin_nan("");
}

__global__ void k_scale_bounds(Dev d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d.M) return;
  d.l[j] = d.E[j] * fmax(d.raw_l[j], -QP_INFTY);
  d.u[j] = d.E[j] * fmin(d.raw_u[j], QP_INFTY);
}

__global__ void k_scale_warm(Dev d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d.n) d.x[j] = d.Dinv[j] * d.raw_x[j];
  if (j < d.M) d.y[j] = d.c * d.Einv[j] * d.raw_y[j];
}

__global__ void k_scale_q(Dev d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d.n) {
    d.qraw[j] = d.raw_x[j];
    d.q[j] = d.c * d.D[j] * d.raw_x[j];
  }
}

__global__ void k_zero_iterates(Dev d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d.n) d.x[j] = 0.0;
  if (j < d.M) d.z[j] = d.y[j] = 0.0;
}

// z = Abar x (node.py:105: warm_start derives z from x)
template <int TPR>
__global__ __launch_bounds__(256) void k_warm_z(Dev d) {
  ROW_SETUP(TPR)
  const bool live = row_raw < d.M;
  const int row = live ? row_raw : d.M - 1;
  double acc[1];
  acc[0] = prow_dot<TPR>(d.pc_idx, d.pc_A, d.pc_ptr[row], d.pc_ptr[row + 1], t, d.x);
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) d.z[row] = acc[0];
}

__global__ void k_init_wh(Dev d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d.M) d.wh[j] = d.z[j] - d.y[j] / d.rho;
}

// one workgroup: unscale the answer (or build the certificate), then the integer clamp of
// node.py:131-136 when `node` is set
__global__ __launch_bounds__(1024) void k_finish(Dev d, int node, int max_iter) {
  __shared__ double lds[16];
  const int n = d.n, M = d.M, tid = threadIdx.x;
  Ctrl *c = d.ctrl;
  int st = c->status;
  if (st == MIOSQP_QP_UNSOLVED) st = MIOSQP_QP_MAX_ITER_REACHED;
  const double nan = __builtin_nan("");
  if (st == MIOSQP_QP_PRIMAL_INFEASIBLE) {
    double nd = 0;
    for (int j = tid; j < M; j += 1024) nd = fmax(nd, fabs(d.E[j] * d.dy[j]));
    nd = block_max(nd, lds);
    for (int i = tid; i < n; i += 1024) d.out_x[i] = nan;
    for (int j = tid; j < M; j += 1024) d.out_y[j] = d.E[j] * d.dy[j] / nd;
  } else if (st == MIOSQP_QP_DUAL_INFEASIBLE) {
    double nd = 0;
    for (int i = tid; i < n; i += 1024) nd = fmax(nd, fabs(d.D[i] * d.dx[i]));
    nd = block_max(nd, lds);
    for (int i = tid; i < n; i += 1024) d.out_x[i] = d.D[i] * d.dx[i] / nd;
    for (int j = tid; j < M; j += 1024) d.out_y[j] = nan;
  } else {
    for (int i = tid; i < n; i += 1024) d.out_x[i] = d.D[i] * d.x[i];
    for (int j = tid; j < M; j += 1024) d.out_y[j] = d.cinv * d.E[j] * d.y[j];
    if (node) {
      __syncthreads();
      for (int k = tid; k < d.n_int; k += 1024) {
        const int i = d.i_idx[k];
        d.out_x[i] = fmin(fmax(d.out_x[i], d.raw_l[d.m_orig + k]), d.raw_u[d.m_orig + k]);
      }
    }
  }
  if (tid == 0) {
    c->status = st;
    c->done = 1;
    (void)max_iter;
  }
}

// rows of the unscaled P: t_i = x_i (0.5 (P x)_i + q_i)   (data.py:99-103)
template <int TPR>
__global__ __launch_bounds__(256) void k_obj_rows(Dev d) {
  ROW_SETUP(TPR)
  const bool live = row_raw < d.n;
  const int row = live ? row_raw : d.n - 1;
  double acc[1];
  acc[0] = prow_dot<TPR>(d.pr_idx, d.pr_val, d.pr_ptr[row], d.pr_ptr[row + 1], t, d.out_x);
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) d.sn[row] = d.out_x[row] * (0.5 * acc[0] + d.qraw[row]);
}

__global__ __launch_bounds__(1024) void k_obj_sum(Dev d) {
  __shared__ double lds[16];
  double s = 0;
  for (int i = threadIdx.x; i < d.n; i += 1024) s += d.sn[i];
  s = block_sum(s, lds);
  if (threadIdx.x == 0) {
    const int st = d.ctrl->status;
    d.ctrl->lower = (st == MIOSQP_QP_SOLVED || st == MIOSQP_QP_MAX_ITER_REACHED) ? s : __builtin_nan("");
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
inline int pick_tpr(double avg_row) {
  if (avg_row <= 48) return 16;
  if (avg_row <= 192) return 32;
  if (avg_row <= 640) return 64;
  if (avg_row <= 1536) return 128;
  return 256;
}

#define DISPATCH_TPR(tpr, KERNEL, rows, stream, ...)                                              \
  do {                                                                                            \
    switch (tpr) {                                                                                \
      case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(((rows) + 15) / 16), dim3(256), 0, stream, __VA_ARGS__); break;   \
      case 32: hipLaunchKernelGGL(KERNEL<32>, dim3(((rows) + 7) / 8), dim3(256), 0, stream, __VA_ARGS__); break;     \
      case 64: hipLaunchKernelGGL(KERNEL<64>, dim3(((rows) + 3) / 4), dim3(256), 0, stream, __VA_ARGS__); break;     \
      case 128: hipLaunchKernelGGL(KERNEL<128>, dim3(((rows) + 1) / 2), dim3(256), 0, stream, __VA_ARGS__); break;   \
      default: hipLaunchKernelGGL(KERNEL<256>, dim3(rows), dim3(256), 0, stream, __VA_ARGS__); break;                \
    }                                                                                             \
  } while (0)

template <typename T>
int upload(const std::vector<T> &h, T **dptr) {
  size_t bytes = (h.size() ? h.size() : 1) * sizeof(T);
  HIPCHK(hipMalloc((void **)dptr, bytes));
  if (h.size()) HIPCHK(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

}  // namespace

struct miosqp_qp_engine {
  int n = 0, M = 0;
  miosqp_qp_settings st{};
  miosqp::Scaled sc;
  miosqp::Factor fa;
  Dev d{};
  std::vector<void *> allocs;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evc0 = nullptr, evc1 = nullptr;
  double loop_ms = 0.0;
  int64_t loop_iters = 0;
  hipGraph_t g_full = nullptr, g_tail = nullptr;
  hipGraphExec_t x_full = nullptr, x_tail = nullptr;
  int chunk = 25, tail_iters = 0;
  int tpr_pv = 64, tpr_pc = 64, tpr_tail = 64, tpr_pb = 64, tpr_pr = 64;
  // pinned staging: [l | u | x0 | y0] in, [x | y] out, ctrl
  double *h_in = nullptr, *h_out = nullptr;
  Ctrl *h_ctrl = nullptr;
  double *d_in = nullptr;
  bool have_int = false;
  int64_t nnzA = 0, nnzPtriu = 0;
};

namespace {

template <typename T>
int dalloc(miosqp_qp_engine *e, T **p, size_t count) {
  HIPCHK(hipMalloc((void **)p, (count ? count : 1) * sizeof(T)));
  HIPCHK(hipMemset(*p, 0, (count ? count : 1) * sizeof(T)));
  e->allocs.push_back(*p);
  return 0;
}
template <typename T>
int dupload(miosqp_qp_engine *e, const std::vector<T> &h, const T **p) {
  T *q = nullptr;
  int rc = upload(h, &q);
  if (rc) return rc;
  e->allocs.push_back(q);
  *p = q;
  return 0;
}

void launch_iteration(miosqp_qp_engine *e) {
  const Dev &d = e->d;
  DISPATCH_TPR(e->tpr_pv, k_panel_fwd, d.n, e->stream, d);
  DISPATCH_TPR(e->tpr_tail, k_tail_fwd, d.n, e->stream, d);
  DISPATCH_TPR(e->tpr_tail, k_tail_bwd, d.n, e->stream, d);
  DISPATCH_TPR(e->tpr_pc, k_panel_bwd, d.M, e->stream, d);
}

void launch_check(miosqp_qp_engine *e, int iters_in_chunk) {
  const Dev &d = e->d;
  DISPATCH_TPR(e->tpr_pc, k_check_con, d.M, e->stream, d);
  // P rows and A^T rows in one launch; both use 64 threads per row
  const int nblk = 2 * ((d.n + 3) / 4);
  hipLaunchKernelGGL((k_check_var<64, 64>), dim3(nblk), dim3(256), 0, e->stream, d);
  hipLaunchKernelGGL(k_check_decide, dim3(1), dim3(1024), 0, e->stream, d, iters_in_chunk);
}

int capture_chunk(miosqp_qp_engine *e, int iters, hipGraph_t *g, hipGraphExec_t *x) {
  HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < iters; i++) launch_iteration(e);
  launch_check(e, iters);
  HIPCHK(hipStreamEndCapture(e->stream, g));
  HIPCHK(hipGraphInstantiate(x, *g, nullptr, nullptr, 0));
  return 0;
}

double wall() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// runs the ADMM loop on the device until a status is decided or max_iter is reached
int run_loop(miosqp_qp_engine *e) {
  const int nfull = e->st.max_iter / e->chunk;
  for (int k = 0; k < nfull; k++) {
    HIPCHK(hipEventRecord(e->evc0, e->stream));
    HIPCHK(hipGraphLaunch(e->x_full, e->stream));
    HIPCHK(hipEventRecord(e->evc1, e->stream));
    HIPCHK(hipMemcpyAsync(e->h_ctrl, e->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->evc0, e->evc1));
    e->loop_ms += ms;
    e->loop_iters += e->chunk;
    if (e->h_ctrl->done) return 0;
  }
  if (e->tail_iters > 0) HIPCHK(hipGraphLaunch(e->x_tail, e->stream));
  return 0;
}

int finish_and_fetch(miosqp_qp_engine *e, int node, double *x_out, double *y_out, miosqp_qp_info *info,
                     double t0) {
  const Dev &d = e->d;
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, e->stream, d, node, e->st.max_iter);
  if (node) {
    DISPATCH_TPR(e->tpr_pr, k_obj_rows, d.n, e->stream, d);
    hipLaunchKernelGGL(k_obj_sum, dim3(1), dim3(1024), 0, e->stream, d);
  }
  HIPCHK(hipMemcpyAsync(e->h_out, d.out_x, sizeof(double) * (e->n + e->M), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_ctrl, d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  memcpy(x_out, e->h_out, sizeof(double) * e->n);
  memcpy(y_out, e->h_out + e->n, sizeof(double) * e->M);
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  info->status_val = e->h_ctrl->status;
  info->iter = e->h_ctrl->iter;
  info->obj_val = e->h_ctrl->obj_val;
  info->pri_res = e->h_ctrl->pri_res;
  info->dua_res = e->h_ctrl->dua_res;
  info->lower = e->h_ctrl->lower;
  info->device_time = 1e-3 * ms;
  info->run_time = wall() - t0;
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char *miosqp_qp_last_error(void) { return g_err.c_str(); }

int miosqp_qp_default_settings(miosqp_qp_settings *s) {
  if (!s) return MIOSQP_EARG;
  memset(s, 0, sizeof *s);
  s->rho = 0.1;
  s->sigma = 1e-6;
  s->alpha = 1.6;
  s->eps_abs = 1e-3;
  s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4;
  s->eps_dual_inf = 1e-4;
  s->max_iter = 4000;
  s->scaling = 10;
  s->check_termination = 25;
  s->warm_start = 1;
  s->device = -1;
  s->max_batch = 1;
  return 0;
}

int miosqp_qp_constant(const char *name) {
  if (!name) return 0;
  if (!strcmp(name, "OSQP_SOLVED")) return MIOSQP_QP_SOLVED;
  if (!strcmp(name, "OSQP_MAX_ITER_REACHED")) return MIOSQP_QP_MAX_ITER_REACHED;
  if (!strcmp(name, "OSQP_PRIMAL_INFEASIBLE")) return MIOSQP_QP_PRIMAL_INFEASIBLE;
  if (!strcmp(name, "OSQP_DUAL_INFEASIBLE")) return MIOSQP_QP_DUAL_INFEASIBLE;
  if (!strcmp(name, "OSQP_UNSOLVED")) return MIOSQP_QP_UNSOLVED;
  return 0;
}

int miosqp_qp_cleanup(miosqp_qp_engine *e) {
  if (!e) return 0;
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->x_full) hipGraphExecDestroy(e->x_full);
  if (e->x_tail) hipGraphExecDestroy(e->x_tail);
  if (e->g_full) hipGraphDestroy(e->g_full);
  if (e->g_tail) hipGraphDestroy(e->g_tail);
  for (void *p : e->allocs) hipFree(p);
  if (e->h_in) hipHostFree(e->h_in);
  if (e->h_out) hipHostFree(e->h_out);
  if (e->h_ctrl) hipHostFree(e->h_ctrl);
  if (e->ev0) hipEventDestroy(e->ev0);
  if (e->ev1) hipEventDestroy(e->ev1);
  if (e->evc0) hipEventDestroy(e->evc0);
  if (e->evc1) hipEventDestroy(e->evc1);
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
  return 0;
}

int miosqp_qp_setup(miosqp_qp_engine **out, int32_t n, int32_t M, const int32_t *Pp, const int32_t *Pi,
                    const double *Px, const int32_t *Ap, const int32_t *Ai, const double *Ax,
                    const double *q, const double *l, const double *u, const miosqp_qp_settings *s) {
  if (!out || n <= 0 || M < 0 || !Pp || !Ap || !q || !s || (M > 0 && (!l || !u))) {
    g_err = "setup: bad argument";
    return MIOSQP_EARG;
  }
  for (int i = 0; i < M; i++)
    if (l[i] > u[i]) {
      g_err = "setup: lower bound above upper bound";
      return MIOSQP_EARG;
    }
  if (!(s->rho > 0) || !(s->sigma > 0) || !(s->alpha > 0 && s->alpha < 2) || s->max_iter <= 0) {
    g_err = "setup: settings out of range";
    return MIOSQP_EARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device visible: the relaxation engine has no CPU fallback";
    return MIOSQP_ENODEV;
  }
  if (s->device >= 0) HIPCHK(hipSetDevice(s->device));
  miosqp_qp_engine *e = new miosqp_qp_engine();
  e->n = n;
  e->M = M;
  e->st = *s;
  if (e->st.check_termination <= 0 || e->st.check_termination > e->st.max_iter)
    e->st.check_termination = e->st.max_iter;
  miosqp::scale_problem(n, M, Pp, Pi, Px, Ap, Ai, Ax, q, s->scaling, e->sc);
  std::string err;
  if (!miosqp::build_factor(e->sc, Pp, Pi, Px, s->rho, s->sigma, e->fa, err)) {
    g_err = err;
    delete e;
    return MIOSQP_EFACTOR;
  }
  e->nnzA = Ap[n];
  e->nnzPtriu = (int64_t)e->sc.Pi.size();
  const miosqp::Factor &f = e->fa;
  Dev &d = e->d;
  d.n = n; d.M = M; d.ld = f.ld; d.n_int = 0; d.m_orig = M;
  d.rho = s->rho; d.sigma = s->sigma; d.alpha = s->alpha; d.eps_abs = s->eps_abs; d.eps_rel = s->eps_rel;
  d.eps_pinf = s->eps_prim_inf; d.eps_dinf = s->eps_dual_inf; d.c = e->sc.c; d.cinv = e->sc.cinv;
#define UP(vec, field)                                   \
  do {                                                   \
    int rc__ = dupload(e, vec, &d.field);                \
    if (rc__) { miosqp_qp_cleanup(e); return rc__; }     \
  } while (0)
  UP(f.panel_by_var.ptr, pv_ptr); UP(f.panel_by_var.idx, pv_idx); UP(f.panel_by_var.val, pv_L); UP(f.At_val, pv_At);
  UP(f.panel_by_con.ptr, pc_ptr); UP(f.panel_by_con.idx, pc_idx); UP(f.panel_by_con.val, pc_L); UP(f.A_val, pc_A);
  UP(f.Linv, Linv); UP(f.LinvT, LinvT); UP(f.d2inv, d2inv);
  UP(f.Pbar.ptr, pb_ptr); UP(f.Pbar.idx, pb_idx); UP(f.Pbar.val, pb_val);
  UP(f.Praw.ptr, pr_ptr); UP(f.Praw.idx, pr_idx); UP(f.Praw.val, pr_val);
  UP(e->sc.D, D); UP(e->sc.Dinv, Dinv); UP(e->sc.E, E); UP(e->sc.Einv, Einv);
#undef UP
#define AL(field, count)                                 \
  do {                                                   \
    int rc__ = dalloc(e, &d.field, (size_t)(count));     \
    if (rc__) { miosqp_qp_cleanup(e); return rc__; }     \
  } while (0)
  AL(q, n); AL(qraw, n); AL(l, M); AL(u, M); AL(x, n); AL(z, M); AL(y, M); AL(wh, M); AL(cv, n); AL(ut, n);
  AL(xt, n); AL(dx, n); AL(dy, M); AL(sm, 8 * (size_t)M); AL(sn, 10 * (size_t)n);
  AL(ctrl, 1);
  // staging block: raw_l | raw_u | raw_x | raw_y contiguous, out_x | out_y contiguous
  AL(raw_l, 2 * (size_t)M + n + M);
  d.raw_u = d.raw_l + M; d.raw_x = d.raw_u + M; d.raw_y = d.raw_x + n;
  AL(out_x, (size_t)n + M);
  d.out_y = d.out_x + n;
  {
    int *ii = nullptr;
    int rc = dalloc(e, &ii, (size_t)n);
    if (rc) { miosqp_qp_cleanup(e); return rc; }
    d.i_idx = ii;
  }
#undef AL
  HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&e->ev0));
  HIPCHK(hipEventCreate(&e->ev1));
  HIPCHK(hipEventCreate(&e->evc0));
  HIPCHK(hipEventCreate(&e->evc1));
  HIPCHK(hipHostMalloc((void **)&e->h_in, sizeof(double) * (2 * (size_t)M + n + M + 1), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&e->h_out, sizeof(double) * ((size_t)n + M + 1), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&e->h_ctrl, sizeof(Ctrl), hipHostMallocDefault));
  // scaled q, raw q, scaled bounds
  HIPCHK(hipMemcpy(d.q, e->sc.q.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.qraw, q, sizeof(double) * n, hipMemcpyHostToDevice));
  if (M > 0) {
    HIPCHK(hipMemcpy(d.raw_l, l, sizeof(double) * M, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d.raw_u, u, sizeof(double) * M, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_scale_bounds, dim3((M + 255) / 256), dim3(256), 0, e->stream, d);
  }
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, d);
  e->tpr_pv = pick_tpr((double)f.nnz_panel / n);
  e->tpr_pc = pick_tpr(M > 0 ? (double)f.nnz_panel / M : 1.0);
  e->tpr_tail = pick_tpr(0.5 * n);
  e->tpr_pb = pick_tpr((double)f.Pbar.nnz / n);
  e->tpr_pr = pick_tpr((double)f.Praw.nnz / n);
  e->chunk = e->st.check_termination;
  e->tail_iters = e->st.max_iter % e->chunk;
  HIPCHK(hipStreamSynchronize(e->stream));
  int rc = capture_chunk(e, e->chunk, &e->g_full, &e->x_full);
  if (!rc && e->tail_iters > 0) rc = capture_chunk(e, e->tail_iters, &e->g_tail, &e->x_tail);
  if (rc) { miosqp_qp_cleanup(e); return rc; }
  *out = e;
  return 0;
}

int miosqp_qp_update_bounds(miosqp_qp_engine *e, const double *l, const double *u) {
  if (!e || !l || !u) return MIOSQP_EARG;
  for (int i = 0; i < e->M; i++)
    if (l[i] > u[i]) return MIOSQP_EBOUNDS;
  memcpy(e->h_in, l, sizeof(double) * e->M);
  memcpy(e->h_in + e->M, u, sizeof(double) * e->M);
  HIPCHK(hipMemcpyAsync(e->d.raw_l, e->h_in, sizeof(double) * 2 * e->M, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_scale_bounds, dim3((e->M + 255) / 256), dim3(256), 0, e->stream, e->d);
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

int miosqp_qp_update_lin_cost(miosqp_qp_engine *e, const double *q) {
  if (!e || !q) return MIOSQP_EARG;
  double *hx = e->h_in + 2 * (size_t)e->M;
  memcpy(hx, q, sizeof(double) * e->n);
  HIPCHK(hipMemcpyAsync(e->d.raw_x, hx, sizeof(double) * e->n, hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_scale_q, dim3((e->n + 255) / 256), dim3(256), 0, e->stream, e->d);
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

static int enqueue_warm(miosqp_qp_engine *e) {
  const int big = e->n > e->M ? e->n : e->M;
  hipLaunchKernelGGL(k_scale_warm, dim3((big + 255) / 256), dim3(256), 0, e->stream, e->d);
  DISPATCH_TPR(e->tpr_pc, k_warm_z, e->M, e->stream, e->d);
  return 0;
}

int miosqp_qp_warm_start(miosqp_qp_engine *e, const double *x, const double *y) {
  if (!e || !x || !y) return MIOSQP_EARG;
  double *hx = e->h_in + 2 * (size_t)e->M;
  memcpy(hx, x, sizeof(double) * e->n);
  memcpy(hx + e->n, y, sizeof(double) * e->M);
  HIPCHK(hipMemcpyAsync(e->d.raw_x, hx, sizeof(double) * (e->n + e->M), hipMemcpyHostToDevice, e->stream));
  enqueue_warm(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

static int begin_solve(miosqp_qp_engine *e) {
  const int big = e->n > e->M ? e->n : e->M;
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  if (!e->st.warm_start) hipLaunchKernelGGL(k_zero_iterates, dim3((big + 255) / 256), dim3(256), 0, e->stream, e->d);
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, e->d);
  hipLaunchKernelGGL(k_init_wh, dim3((e->M + 255) / 256), dim3(256), 0, e->stream, e->d);
  return 0;
}

int miosqp_qp_solve(miosqp_qp_engine *e, double *x_out, double *y_out, miosqp_qp_info *info) {
  if (!e || !x_out || !y_out || !info) return MIOSQP_EARG;
  const double t0 = wall();
  int rc = begin_solve(e);
  if (!rc) rc = run_loop(e);
  if (!rc) rc = finish_and_fetch(e, 0, x_out, y_out, info, t0);
  return rc;
}

int miosqp_qp_set_integer_rows(miosqp_qp_engine *e, int32_t n_int, const int32_t *i_idx, int32_t m_orig) {
  if (!e || n_int < 0 || n_int > e->n || m_orig < 0 || m_orig + n_int != e->M || (n_int && !i_idx)) {
    g_err = "set_integer_rows: need m_orig + n_int == M";
    return MIOSQP_EARG;
  }
  for (int k = 0; k < n_int; k++)
    if (i_idx[k] < 0 || i_idx[k] >= e->n) return MIOSQP_EARG;
  if (n_int) HIPCHK(hipMemcpy((void *)e->d.i_idx, i_idx, sizeof(int) * n_int, hipMemcpyHostToDevice));
  e->d.n_int = n_int;
  e->d.m_orig = m_orig;
  e->have_int = true;
  // the captured graphs hold Dev by value but never read n_int / m_orig / i_idx contents
  return 0;
}

int miosqp_qp_solve_node(miosqp_qp_engine *e, const double *l, const double *u, const double *x0,
                         const double *y0, double *x_out, double *y_out, miosqp_qp_info *info) {
  if (!e || !l || !u || !x0 || !y0 || !x_out || !y_out || !info) return MIOSQP_EARG;
  if (!e->have_int) {
    g_err = "solve_node: call miosqp_qp_set_integer_rows first";
    return MIOSQP_EARG;
  }
  const double t0 = wall();
  const int n = e->n, M = e->M;
  for (int i = 0; i < M; i++)
    if (l[i] > u[i]) return MIOSQP_EBOUNDS;
  memcpy(e->h_in, l, sizeof(double) * M);
  memcpy(e->h_in + M, u, sizeof(double) * M);
  memcpy(e->h_in + 2 * (size_t)M, x0, sizeof(double) * n);
  memcpy(e->h_in + 2 * (size_t)M + n, y0, sizeof(double) * M);
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  HIPCHK(hipMemcpyAsync(e->d.raw_l, e->h_in, sizeof(double) * (3 * (size_t)M + n), hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(k_scale_bounds, dim3((M + 255) / 256), dim3(256), 0, e->stream, e->d);
  enqueue_warm(e);
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, e->d);
  hipLaunchKernelGGL(k_init_wh, dim3((M + 255) / 256), dim3(256), 0, e->stream, e->d);
  int rc = run_loop(e);
  if (!rc) rc = finish_and_fetch(e, 1, x_out, y_out, info, t0);
  return rc;
}

int miosqp_qp_solve_batch(miosqp_qp_engine *e, int32_t B, const double *l, const double *u, const double *x0,
                          const double *y0, double *x_out, double *y_out, miosqp_qp_info *info) {
  if (!e || B < 0) return MIOSQP_EARG;
  const size_t n = e->n, M = e->M;
  for (int b = 0; b < B; b++) {
    int rc = miosqp_qp_solve_node(e, l + b * M, u + b * M, x0 + b * n, y0 + b * M, x_out + b * n, y_out + b * M,
                                  info + b);
    if (rc) return rc;
  }
  return 0;
}

int miosqp_qp_debug_iterate(miosqp_qp_engine *e, int32_t k, double *x, double *z, double *y) {
  if (!e || k < 0) return MIOSQP_EARG;
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, e->d);
  hipLaunchKernelGGL(k_init_wh, dim3((e->M + 255) / 256), dim3(256), 0, e->stream, e->d);
  for (int i = 0; i < k; i++) launch_iteration(e);
  HIPCHK(hipStreamSynchronize(e->stream));
  if (x) HIPCHK(hipMemcpy(x, e->d.x, sizeof(double) * e->n, hipMemcpyDeviceToHost));
  if (z) HIPCHK(hipMemcpy(z, e->d.z, sizeof(double) * e->M, hipMemcpyDeviceToHost));
  if (y) HIPCHK(hipMemcpy(y, e->d.y, sizeof(double) * e->M, hipMemcpyDeviceToHost));
  return 0;
}

int miosqp_qp_get_scaling(miosqp_qp_engine *e, double *D, double *E, double *c) {
  if (!e) return MIOSQP_EARG;
  if (D) memcpy(D, e->sc.D.data(), sizeof(double) * e->n);
  if (E) memcpy(E, e->sc.E.data(), sizeof(double) * e->M);
  if (c) *c = e->sc.c;
  return 0;
}

static void kernel_bytes(const miosqp_qp_engine *e, double b[5]) {
  const double n = e->n, M = e->M, np = (double)e->fa.nnz_panel, nt = (double)e->fa.nnz_tail;
  // 12 B per factor entry (value + index), 4 B row pointers, 8 B vectors
  b[0] = np * 12 + (n + 1) * 4 + (M + 3 * n) * 8;          // panel forward: wh in; x, q in; c out
  b[1] = nt * 12 + (n + 1) * 4 + (3 * n) * 8;              // tail forward: c in, d2inv in, ut out
  b[2] = nt * 12 + (n + 1) * 4 + (5 * n) * 8;              // tail backward: ut in, x in; xt, x, dx out
  b[3] = np * 12 + (M + 1) * 4 + (n + 10 * M) * 8;         // panel backward + z/y update
  const double k = e->st.check_termination;
  const double NK = n + M;
  b[4] = 2 * (np + nt) * 12 + 2 * (NK + 1) * 4 + NK * 8 + NK * 20 + (6 * n + 16 * M) * 8 +
         (2.0 * e->nnzA + e->nnzPtriu) * 12 / k;
}

int miosqp_qp_get_factor_stats(miosqp_qp_engine *e, int64_t *out) {
  if (!e || !out) return MIOSQP_EARG;
  double b[5];
  kernel_bytes(e, b);
  out[0] = e->fa.nnz_panel + e->fa.nnz_tail;
  out[1] = e->fa.nnz_panel;
  out[2] = e->n;
  out[3] = (int64_t)b[4];
  out[4] = e->tpr_pv; out[5] = e->tpr_pc; out[6] = e->tpr_tail; out[7] = 0;
  return 0;
}

int miosqp_qp_get_loop_stats(miosqp_qp_engine *e, double *ms, int64_t *iters, int32_t reset) {
  if (!e) return MIOSQP_EARG;
  if (ms) *ms = e->loop_ms;
  if (iters) *iters = e->loop_iters;
  if (reset) {
    e->loop_ms = 0.0;
    e->loop_iters = 0;
  }
  return 0;
}

int miosqp_qp_time_kernel(miosqp_qp_engine *e, int32_t which, int32_t reps, double *usec, double *bytes) {
  if (!e || which < 0 || which > 4 || reps <= 0 || !usec) return MIOSQP_EARG;
  const Dev &d = e->d;
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, d);
  auto one = [&]() {
    switch (which) {
      case 0: DISPATCH_TPR(e->tpr_pv, k_panel_fwd, d.n, e->stream, d); break;
      case 1: DISPATCH_TPR(e->tpr_tail, k_tail_fwd, d.n, e->stream, d); break;
      case 2: DISPATCH_TPR(e->tpr_tail, k_tail_bwd, d.n, e->stream, d); break;
      case 3: DISPATCH_TPR(e->tpr_pc, k_panel_bwd, d.M, e->stream, d); break;
      default: launch_iteration(e); break;
    }
  };
  for (int i = 0; i < 5; i++) one();
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  for (int i = 0; i < reps; i++) one();
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  *usec = 1e3 * ms / reps;
  if (bytes) {
    double b[5];
    kernel_bytes(e, b);
    *bytes = b[which];
  }
  return 0;
}

}  // extern "C"
