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

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/miosqp_amd.h"
#include "factor.hpp"

#define QP_INFTY 1e30
#define QP_MIN_SCALING 1e-4
#define QP_DIVISION_TOL (1.0 / QP_INFTY)

// dense_setup.hip: dense LDL^T + triangular inverse on the device (miosqp::DenseLdlInv)
int miosqp_device_ldl_inverse(int n, int ld, const double *S, double *d, double *Linv, double *LinvT, void *ctx);
// dense_setup.hip: explicit KKT inverse W = F^T D22^-1 F from the product-form rows (device pointers)
int miosqp_device_kkt_inverse(const double *F, int ldf, const double *dinv, int n, int M, double *W, int ldw,
                              hipStream_t stream);

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
  int B, ndone, pad3[2];  // batched mode: active columns, columns already decided
  double pri_res, dua_res, obj_val, lower;
  double nrm_dy, nrm_dx;  // certificate normalisers
  double pad2[2];
  // node digest (branching epilogue, workspace.py:245-272 on the device)
  int int_inf, nextvar, pad4[2];
  double heur_viol, heur_obj;
};

// everything a kernel needs, passed by value
struct Dev {
  int n, M, ld, n_int, m_orig;
  double rho, rho_inv, sigma, alpha, eps_abs, eps_rel, eps_pinf, eps_dinf, c, cinv;
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
  // ---- folded (product-form) factor: rows of L^-1, see factor.hpp ----
  const double *f_rows, *f_GmT, *f_Ad, *f_Atd, *f_Pd;
  int ldf, ldn, ldm;
  double *rx;  // sigma x - q, kept right behind wh so that [wh | rx] is one contiguous vector
  // node digest: rounded candidate (unscaled / scaled), root bounds, tolerances
  double *xi, *xis, *root_l, *root_u;
  double eps_int, eps_lin;
  int digest;
  int bm_ablate;  // debug: 1 = no operand loads, 2 = no matrix-core instructions
  unsigned long long *prof;  // debug timeline (per-block start/end, 100 MHz wall clock) or nullptr
  // ---- cooperative register-resident solver (k_coop): explicit KKT inverse, exchange buffers ----
  const double *W;               // N x ldw, N = M + n, ordering [constraints ; variables]
  int ldw;
  unsigned *coop_tag;            // tag of the last exchange round that completed
  unsigned long long *coop_buf;  // 2 parities x N x {lo32|tag, hi32|tag}
  unsigned long long *coop_chk;  // 2 x coop_half: the test's operands [y ; x] and [proj(dy) ; dx]
  unsigned long long *coop_q;    // (256 + 16) x COOP_QS: per-workgroup, then per-group norms / sums of the test
  unsigned *coop_reg;            // start-up registration counter (zeroed before every launch)
  const double *Kc;              // [ 0 Abar ; Abar^T Pbar ], N x ldw
  int coop_dbg;                  // debug ablation: 1 = no gather
  int coop_stride;               // 8-byte words between the blocks of consecutive workgroups (>= 2 RW)
  size_t coop_half;              // words per parity
  // ---- batched mode: B nodes share the factor; vectors are [len][Bs], batch index fastest ----
  int Bs;  // column stride, multiple of 64
  double *b_l, *b_u, *b_x, *b_z, *b_y, *b_wh, *b_rx, *b_cv, *b_ut, *b_xt, *b_dx, *b_dy;
  double *b_sm, *b_sn;      // 8 x M x Bs, 4 x n x Bs
  double *b_xfin, *b_yfin;  // unscaled answers, batch-fastest
  double *b_xi, *b_xis;     // rounded candidates (node digest), unscaled / scaled
  double *b_part;           // partial reductions of the batched termination test
  int *c_intinf, *c_nextvar;
  int *c_node;   // column position -> node of the wave (columns are swapped when the wave is compacted)
  int *c_pairs;  // swap list of the current compaction
  double *c_hviol, *c_hobj;
  double *b_raw;            // node-major staging in:  l[B][M] | u[B][M] | x0[B][n] | y0[B][M]
  double *b_out;            // node-major staging out: x[B][n] | y[B][M]
  int *c_done, *c_status, *c_iter;
  double *c_pri, *c_dua, *c_obj, *c_lower;
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
// lane exchange inside a 16-lane row through DPP (no LDS crossbar): quad_perm xor 1 / xor 2, then
// row_half_mirror and row_mirror, which act as xor 4 / xor 8 once the smaller groups are uniform
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// v + (v of lane ^ 16) and v + (v of lane ^ 32) without the LDS crossbar: gfx950's permlane swaps exchange
// the odd 16-lane rows (upper 32 lanes) of one register with the even rows (lower lanes) of another; with
// both holding v, the two results are the two halves of every pair
__device__ __forceinline__ double add_xor16(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double add_xor32(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// butterfly sum over aligned groups of W lanes (W a power of two), small strides first; fixed order
template <int NV>
__device__ __forceinline__ void group_sum(double (&v)[NV], int W) {
#define GS_DPP(MIN, CTRL)                                                            \
  if (W > MIN) {                                                                     \
    _Pragma("unroll") for (int k = 0; k < NV; k++) v[k] += dpp_get<CTRL>(v[k]);      \
  }
#define GS_SWAP(MIN, FN)                                                             \
  if (W > MIN) {                                                                     \
    _Pragma("unroll") for (int k = 0; k < NV; k++) v[k] = FN(v[k]);                   \
  }
  GS_DPP(1, 0xB1) GS_DPP(2, 0x4E) GS_DPP(4, 0x141) GS_DPP(8, 0x140) GS_SWAP(16, add_xor16) GS_SWAP(32, add_xor32)
#undef GS_DPP
#undef GS_SWAP
}

template <int TPR, int NV>
__device__ __forceinline__ void row_reduce(double (&v)[NV], double *lds) {
  constexpr int W = TPR < 64 ? TPR : 64;
  group_sum<NV>(v, W);
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
  if (d.ctrl->done) return;  // chunk queued ahead of a decided test
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
  if (d.ctrl->done) return;  // chunk queued ahead of a decided test
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
  if (d.ctrl->done) return;  // chunk queued ahead of a decided test
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
  if (d.ctrl->done) return;  // chunk queued ahead of a decided test
  ROW_SETUP(TPR)
  const bool live = row_raw < d.M;
  const int row = live ? row_raw : d.M - 1;
  double acc[1];
  acc[0] = prow_dot<TPR>(d.pc_idx, d.pc_L, d.pc_ptr[row], d.pc_ptr[row + 1], t, d.xt);
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) {
    const double rho = d.rho, rinv = d.rho_inv, alpha = d.alpha;
    const double zp = d.z[row], yp = d.y[row];
    const double nu = -rho * d.wh[row] - acc[0];
    const double zt = zp + rinv * (nu - yp);
    const double zr = alpha * zt + (1.0 - alpha) * zp;
    const double v = zr + rinv * yp;
    const double zn = fmin(fmax(v, d.l[row]), d.u[row]);
    const double dy = rho * (zr - zn);
    const double yn = yp + dy;
    d.z[row] = zn;
    d.y[row] = yn;
    d.dy[row] = dy;
    d.wh[row] = zn - rinv * yn;
  }
}

// ------------------------------------------------------------------------------------------
// folded factor: the same iteration as TWO kernels (rows of L^-1, rows of L^-T)
//   k_fold_fwd   ut = D22^-1 ( rx + [ -G | strict_lower(Linv) ] [wh ; rx] )          (n rows)
//   k_fold_bwd   rows 0..n:   x~ = ut + strict_upper(Linv^T) ut ; x, dx, rx
//                rows n..n+M: nu = -rho wh + (-G)^T ut ; z~, z, y, dy, wh
// ------------------------------------------------------------------------------------------
template <int TPR>
__device__ __forceinline__ double drow_dot(const double *__restrict__ r, int len, const double *__restrict__ v,
                                           int t, int skip = 0) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int len2 = len & ~1;
  int j = 2 * t + skip;
  for (; j + 2 * TPR < len2; j += 4 * TPR) {
    const double2 a = *reinterpret_cast<const double2 *>(r + j);
    const double2 b = *reinterpret_cast<const double2 *>(v + j);
    const double2 c = *reinterpret_cast<const double2 *>(r + j + 2 * TPR);
    const double2 e = *reinterpret_cast<const double2 *>(v + j + 2 * TPR);
    a0 = fma(a.x, b.x, a0);
    a1 = fma(a.y, b.y, a1);
    a2 = fma(c.x, e.x, a2);
    a3 = fma(c.y, e.y, a3);
  }
  if (j < len2) {
    const double2 a = *reinterpret_cast<const double2 *>(r + j);
    const double2 b = *reinterpret_cast<const double2 *>(v + j);
    a0 = fma(a.x, b.x, a0);
    a1 = fma(a.y, b.y, a1);
  }
  if (t == 0 && (len & 1)) a2 = fma(r[len - 1], v[len - 1], a2);
  return (a0 + a1) + (a2 + a3);
}

template <int TPR>
__global__ __launch_bounds__(256) void k_fold_fwd(Dev d) {
  const int done = d.ctrl->done;  // tested below, after the first loads are in flight
  const unsigned long long t_in = d.prof ? wall_clock64() : 0;
  ROW_SETUP(TPR)
  const bool live = row_raw < d.n;
  const int row = live ? row_raw : d.n - 1;
  const double rxi = d.rx[row], di = d.d2inv[row];
  const double *__restrict__ rr = d.f_rows + (size_t)row * d.ldf;
  const int len = d.M + row;
  double2 pa = make_double2(0.0, 0.0), pb = make_double2(0.0, 0.0);
  if (2 * t < (len & ~1)) {
    pa = *reinterpret_cast<const double2 *>(rr + 2 * t);
    pb = *reinterpret_cast<const double2 *>(d.wh + 2 * t);
  }
  if (done) return;  // chunk queued ahead of a decided test: nothing to do
  double acc[1];
  acc[0] = fma(pa.x, pb.x, pa.y * pb.y) + drow_dot<TPR>(rr, len, d.wh, t, 2 * TPR);
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) d.ut[row] = di * (rxi + acc[0]);
  if (d.prof && threadIdx.x == 0) {
    d.prof[2 * blockIdx.x] = t_in;
    d.prof[2 * blockIdx.x + 1] = wall_clock64();
  }
}

template <int TPR_X, int TPR_C>
__global__ __launch_bounds__(256) void k_fold_bwd(Dev d) {
  const int done = d.ctrl->done;  // tested below, after the first loads are in flight
  __shared__ double lds[16];
  const unsigned long long t_in = d.prof ? wall_clock64() : 0;
  const int n = d.n;
  constexpr int RPX = 256 / TPR_X, RPC = 256 / TPR_C;
  const int nbx = (n + RPX - 1) / RPX;
  if ((int)blockIdx.x < nbx) {
    const int row_raw = blockIdx.x * RPX + threadIdx.x / TPR_X;
    const int t = threadIdx.x % TPR_X;
    const bool live = row_raw < n;
    const int row = live ? row_raw : n - 1;
    const double *__restrict__ Ur = d.LinvT + (size_t)row * d.ld;
    const double *__restrict__ u = d.ut;
    const double ui = u[row], xp = d.x[row], qi = d.q[row];
    if (done) return;
    // entries on and below the diagonal are stored zeros, so start at the even column <= row + 1
    const int j0 = (row + 1) & ~1;
    double acc[1] = {drow_dot<TPR_X>(Ur + j0, n - j0, u + j0, t)};
    row_reduce<TPR_X, 1>(acc, lds);
    if (live && t == 0) {
      const double xt = ui + acc[0];
      const double xn = d.alpha * xt + (1.0 - d.alpha) * xp;
      d.xt[row] = xt;
      d.x[row] = xn;
      d.dx[row] = xn - xp;
      d.rx[row] = d.sigma * xn - qi;
    }
    if (d.prof && threadIdx.x == 0) {
      d.prof[2 * blockIdx.x] = t_in;
      d.prof[2 * blockIdx.x + 1] = wall_clock64();
    }
    return;
  }
  const int row_raw = (blockIdx.x - nbx) * RPC + threadIdx.x / TPR_C;
  const int t = threadIdx.x % TPR_C;
  const bool live = row_raw < d.M;
  const int row = live ? row_raw : d.M - 1;
  const double whj = d.wh[row], zp = d.z[row], yp = d.y[row], lj = d.l[row], uj = d.u[row];
  const double *__restrict__ gr = d.f_GmT + (size_t)row * d.ldn;
  double2 pa = make_double2(0.0, 0.0), pb = make_double2(0.0, 0.0);
  if (2 * t < (n & ~1)) {
    pa = *reinterpret_cast<const double2 *>(gr + 2 * t);
    pb = *reinterpret_cast<const double2 *>(d.ut + 2 * t);
  }
  if (done) return;
  double acc[1];
  acc[0] = fma(pa.x, pb.x, pa.y * pb.y) + drow_dot<TPR_C>(gr, n, d.ut, t, 2 * TPR_C);
  row_reduce<TPR_C, 1>(acc, lds);
  if (live && t == 0) {
    const double rho = d.rho, rinv = d.rho_inv, alpha = d.alpha;
    const double nu = -rho * whj + acc[0];
    const double zt = zp + rinv * (nu - yp);
    const double zr = alpha * zt + (1.0 - alpha) * zp;
    const double v = zr + rinv * yp;
    const double zn = fmin(fmax(v, lj), uj);
    const double dy = rho * (zr - zn);
    const double yn = yp + dy;
    d.z[row] = zn;
    d.y[row] = yn;
    d.dy[row] = dy;
    d.wh[row] = zn - rinv * yn;
  }
  if (d.prof && threadIdx.x == 0) {
    d.prof[2 * blockIdx.x] = t_in;
    d.prof[2 * blockIdx.x + 1] = wall_clock64();
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

// The termination decision from the reduced quantities (OSQP paper sec. 3.4); shared by the
// single-node and the batched path.  Returns 0 (keep iterating) or a final status.
struct Norms {
  double pri, nAx, nz, nEv, amax_u, amin_l, dua, nPx, nAty, nq, nPdx, nAtv, ndx, lhs, qdx, xPx, qx;
};
template <class P>
__device__ __forceinline__ int decide_status(const P &d, const Norms &v, double &obj) {
  obj = d.cinv * (0.5 * v.xPx + v.qx);
  const double eps_pri = d.eps_abs + d.eps_rel * fmax(v.nAx, v.nz);
  const double eps_dua = d.eps_abs + d.eps_rel * d.cinv * fmax(fmax(v.nPx, v.nAty), v.nq);
  const bool pri_ok = (d.M == 0) || (v.pri < eps_pri);
  const bool dua_ok = v.dua < eps_dua;
  bool pinf = false, dinf = false;
  if (!pri_ok && v.nEv > QP_DIVISION_TOL && v.lhs < -d.eps_pinf * v.nEv) pinf = v.nAtv < d.eps_pinf * v.nEv;
  if (!dua_ok && v.ndx > QP_DIVISION_TOL && v.qdx < -d.c * d.eps_dinf * v.ndx &&
      v.nPdx < d.c * d.eps_dinf * v.ndx)
    dinf = !(v.amax_u > d.eps_dinf * v.ndx) && !(v.amin_l < -d.eps_dinf * v.ndx);
  if (pri_ok && dua_ok) return MIOSQP_QP_SOLVED;
  if (pinf) { obj = QP_INFTY; return MIOSQP_QP_PRIMAL_INFEASIBLE; }
  if (dinf) { obj = -QP_INFTY; return MIOSQP_QP_DUAL_INFEASIBLE; }
  return 0;
}

constexpr int NQ = 17, NQ_MAX = 13;  // quantities 0..12 reduce with max, 13..16 with +

// one workgroup (4 waves): all norms and sums in ONE pass (two barriers), then the decision
__global__ __launch_bounds__(256) void k_check_decide(Dev d, int iters_in_chunk) {
  if (d.ctrl->done) return;
  __shared__ double part[NQ][4];
  __shared__ double res[NQ];
  const int n = d.n, M = d.M, tid = threadIdx.x;
  double v[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = 0.0;
  v[4] = -1.7e308;
  v[5] = -1.7e308;
  for (int j = tid; j < M; j += 256) {
    v[0] = fmax(v[0], fabs(d.sm[0 * M + j]));
    v[1] = fmax(v[1], fabs(d.sm[1 * M + j]));
    v[2] = fmax(v[2], fabs(d.sm[2 * M + j]));
    v[3] = fmax(v[3], fabs(d.sm[4 * M + j]));
    v[4] = fmax(v[4], d.sm[6 * M + j]);
    v[5] = fmax(v[5], -d.sm[7 * M + j]);
    v[13] += d.sm[5 * M + j];
  }
  for (int i = tid; i < n; i += 256) {
    const double di = d.Dinv[i], px = d.sn[0 * n + i], aty = d.sn[2 * n + i], q = d.q[i], x = d.x[i],
                 dx = d.dx[i];
    v[6] = fmax(v[6], fabs(di * (px + q + aty)));
    v[7] = fmax(v[7], fabs(di * px));
    v[8] = fmax(v[8], fabs(di * aty));
    v[9] = fmax(v[9], fabs(di * q));
    v[10] = fmax(v[10], fabs(d.sn[1 * n + i]));
    v[11] = fmax(v[11], fabs(d.sn[3 * n + i]));
    v[12] = fmax(v[12], fabs(d.D[i] * dx));
    v[14] += q * dx;
    v[15] += x * px;
    v[16] += q * x;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const double o = __shfl_xor(v[q], off, 64);
      v[q] = q < NQ_MAX ? fmax(v[q], o) : v[q] + o;
    }
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int q = 0; q < NQ; q++) part[q][tid >> 6] = v[q];
  }
  __syncthreads();
  if (tid < NQ) {
    double r = part[tid][0];
    for (int w = 1; w < 4; w++) r = tid < NQ_MAX ? fmax(r, part[tid][w]) : r + part[tid][w];
    res[tid] = r;
  }
  __syncthreads();
  if (tid != 0) return;
  Norms nm{res[0], res[1], res[2], res[3], res[4], -res[5], res[6] * d.cinv, res[7], res[8], res[9], res[10],
           res[11], res[12], res[13], res[14], res[15], res[16]};
  Ctrl *c = d.ctrl;
  double obj;
  const int st = decide_status(d, nm, obj);
  c->iter += iters_in_chunk;
  c->pri_res = nm.pri;
  c->dua_res = nm.dua;
  c->obj_val = obj;
  if (st) {
    c->status = st;
    c->done = 1;
  }
}

// ------------------------------------------------------------------------------------------
// cooperative register-resident solver: the same iteration as ONE exchange per step.
//   [nu + rho wh ; x~] = W [wh ; rx],  W = K^-1 restricted as in dense_setup.hip (ks_kkt_inverse).
// Workgroup b keeps rows [b RW, (b+1) RW) of W in registers for the whole launch (thread t holds
// columns t + k*COOP_B, COOP_B = 512 threads per workgroup), owns the iterates of those rows, and after every step publishes its RW new
// entries of [wh ; rx]; every workgroup then gathers the whole vector.  The exchange needs no
// barrier and no flag: an entry travels as two 8-byte words {low half | tag}, {high half | tag}
// (8-byte stores are single-copy atomic), written at agent scope into the buffer of the round's
// parity, and a reader polls its own CPT entries until both tags match.  A workgroup can only
// reach round k+1 after it has seen every entry of round k, so two buffers suffice.
// The termination test runs inside the same launch (every `check_every` iterations): the owners
// also publish [y ; x] and [proj(dy) ; dx], every workgroup applies ITS rows of
//   Kc = [ 0  Abar ; Abar^T  Pbar ]   (read from HBM, only at a test)
// to both, reduces the 17 norms / sums of sec. 3.4 over its rows, publishes them, gathers those of
// all workgroups and takes the decision itself -- identical arithmetic in identical order
// everywhere, so all workgroups agree without a further exchange.
// All workgroups must be co-resident: the grid never exceeds the CU count and one workgroup fits
// per CU next to anything else this engine launches.  Polling is bounded; on expiry ctrl->pad is
// set and the host reports MIOSQP_EHIP.
// ------------------------------------------------------------------------------------------
#define COOP_SPIN_LIMIT (1u << 19)

typedef unsigned ll_u4 __attribute__((ext_vector_type(4)));
// one 16-byte agent-scope store / load per entry ({lo, tag, hi, tag}); the tags in BOTH 8-byte halves
// keep the hand-off correct even if the 16 bytes were ever observed torn
__device__ __forceinline__ void ll_publish(unsigned long long *slot, double v, unsigned tag) {
  ll_u4 w;
  w.x = (unsigned)__double2loint(v); w.y = tag; w.z = (unsigned)__double2hiint(v); w.w = tag;
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot), "v"(w) : "memory");
}
__device__ __forceinline__ ll_u4 ll_peek(const unsigned long long *slot) {
  ll_u4 w;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w) : "v"(slot) : "memory");
  return w;
}

// transposed butterfly: RW values per lane in, the sum over the wave of ONE row per lane out
// (row coop_row<RW>(lane)); halves the live values at every exchange instead of reducing each
// row separately.  Partners: xor 1, xor 2 (quad_perm), 7 - i (row_half_mirror), 15 - i (row_mirror);
// the keep/send choice of a stage must differ between partners of that stage and agree between
// partners of every later stage, hence the mixed bits.
template <int RW>
__device__ __forceinline__ int coop_row(int lane) {
  const int cA = (lane ^ (lane >> 2)) & 1, cB = ((lane >> 1) ^ (lane >> 2)) & 1, cC = ((lane >> 2) ^ (lane >> 3)) & 1,
            cD = (lane >> 3) & 1;
  return RW == 16 ? 8 * cA + 4 * cB + 2 * cC + cD : 4 * cA + 2 * cB + cC;
}
template <int RW>
__device__ __forceinline__ double wave_tsum(double (&v)[RW], int lane) {
  static_assert(RW == 8 || RW == 16, "rows per workgroup");
  const bool cA = (lane ^ (lane >> 2)) & 1, cB = ((lane >> 1) ^ (lane >> 2)) & 1, cC = ((lane >> 2) ^ (lane >> 3)) & 1,
             cD = (lane >> 3) & 1;
#define HALVE(H, C, CTRL)                                        \
  _Pragma("unroll") for (int k = 0; k < (H) / 2; k++) {          \
    const double keep = (C) ? v[k + (H) / 2] : v[k];             \
    const double send = (C) ? v[k] : v[k + (H) / 2];             \
    v[k] = keep + dpp_get<CTRL>(send);                           \
  }
  if constexpr (RW == 16) {
    HALVE(16, cA, 0xB1) HALVE(8, cB, 0x4E) HALVE(4, cC, 0x141) HALVE(2, cD, 0x140)
  } else {
    (void)cD;
    HALVE(8, cA, 0xB1) HALVE(4, cB, 0x4E) HALVE(2, cC, 0x141)
    v[0] += dpp_get<0x140>(v[0]);
  }
#undef HALVE
  return add_xor32(add_xor16(v[0]));
}

constexpr int COOP_QS = 48;  // 8-byte words per workgroup in the norm exchange (NQ entries of 2 words, padded)

// The termination test of the cooperative solver (once per `check_every` iterations).  Returns the
// status (0 = keep iterating); identical in every workgroup.  Must stay inlined: as a real call it
// broke at 512 threads per workgroup (ROCm 7.2), and it takes what it reads of Dev by value.
struct CoopTest {  // what the test reads of Dev, by value (a reference would pin Dev to the stack)
  unsigned long long *coop_chk, *coop_q;
  const double *Kc, *Einv, *E, *Dinv, *D;
  Ctrl *ctrl;
  int n, M, ldw, coop_stride;
  double c, cinv, eps_abs, eps_rel, eps_pinf, eps_dinf;
};

template <int COOP_B, int RW, int CPT>
__device__ __forceinline__ int coop_test(const CoopTest d, unsigned tag, int it, bool own, bool con, double sa,
                                      double sb, double lo, double up, double delta, int *s_fail) {
  constexpr int NW = COOP_B / 64;
  static_assert(NQ <= 32, "norm gather layout");
  // scratch with two uses that never overlap in time: the operands lv[2][2048] (N <= 2048), then the
  // gathered partial results qall[member][NQ]
  __shared__ __attribute__((aligned(16))) double tsc[256 * NQ];
  static_assert(256 * NQ >= 4096, "operand staging");
  double *const qall = tsc;
  double(*const lv)[2048] = reinterpret_cast<double(*)[2048]>(tsc);
  __shared__ double cpart[4][1][RW];
  __shared__ double qrow[RW][NQ];
  __shared__ double qres[NQ];
  __shared__ int s_status;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int M = d.M, N = d.n + d.M, T = gridDim.x;
  const int r0 = blockIdx.x * RW, r = r0 + t;
  int slot[CPT];
#pragma unroll
  for (int k = 0; k < CPT; k++) {
    const int c = t + k * COOP_B;
    slot[k] = (c / RW) * d.coop_stride + 2 * (c % RW);
  }
  double vproj = 0.0;
  if (own && con) {
    const bool uinf = up > QP_INFTY * QP_MIN_SCALING, linf = lo < -QP_INFTY * QP_MIN_SCALING;
    vproj = delta;
    if (uinf && linf) vproj = 0.0;
    else if (uinf) vproj = fmin(vproj, 0.0);
    else if (linf) vproj = fmax(vproj, 0.0);
  }
  // both operands go to LDS; wave w then applies rows 2w, 2w+1 of this workgroup's block of Kc to
  // them (rows are read once, 16 bytes per lane; a constraint row only has its Abar part).
  // Columns < M of a variable row are Abar^T, the others Pbar.
  {
    // both operands of a column sit next to each other: {y | x, proj(dy) | dx}, polled together
    bool have[CPT];
#pragma unroll
    for (int k = 0; k < CPT; k++) have[k] = t + k * COOP_B >= N;
    unsigned spins = 0;
    for (;;) {
      ll_u4 w1[CPT], w2[CPT];
#pragma unroll
      for (int k = 0; k < CPT; k++)
        if (!have[k]) {
          w1[k] = ll_peek(d.coop_chk + 2 * slot[k]);
          w2[k] = ll_peek(d.coop_chk + 2 * slot[k] + 2);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bool all = true;
#pragma unroll
      for (int k = 0; k < CPT; k++)
        if (!have[k]) {
          if (w1[k].y == tag && w1[k].w == tag && w2[k].y == tag && w2[k].w == tag) {
            lv[0][t + k * COOP_B] = __hiloint2double((int)w1[k].z, (int)w1[k].x);
            lv[1][t + k * COOP_B] = __hiloint2double((int)w2[k].z, (int)w2[k].x);
            have[k] = true;
          } else {
            all = false;
          }
        }
      if (all) break;
      if (++spins > COOP_SPIN_LIMIT) {
        *s_fail = 1;
        d.ctrl->pad = 3;
        break;
      }
    }
  }
  __syncthreads();
  if (*s_fail) return 0;
  {
    constexpr int RPW = RW / NW;  // rows per wave
    static_assert(RPW * NW == RW && (RPW == 1 || RPW == 2), "rows per wave");
    double a8[8];  // per row: A^T y, A^T proj(dy), P x (A x), P dx (A dx)
#pragma unroll
    for (int k = 0; k < 8; k++) a8[k] = 0.0;
    const int Ne = N & ~1;
#pragma unroll
    for (int h = 0; h < RPW; h++) {
      const int rr = r0 + RPW * wave + h;
      if (rr >= N) continue;
      const double *row = d.Kc + (size_t)rr * d.ldw;
      double sA1 = 0.0, sA2 = 0.0, sP1 = 0.0, sP2 = 0.0;
      // all 16-byte pieces of the row are requested before the first is used (N <= 2048: at most 16)
      constexpr int NCH = 16;
      const int cbeg = (rr < M ? (M & ~1) : 0) + 2 * lane;
      double2 a[NCH];
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int c = cbeg + 128 * j;
        a[j] = c < Ne ? *reinterpret_cast<const double2 *>(row + c) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int c = cbeg + 128 * j;
        if (c >= Ne) continue;
        const double2 p = *reinterpret_cast<const double2 *>(&lv[0][c]);
        const double2 s2 = *reinterpret_cast<const double2 *>(&lv[1][c]);
        if (c < M) { sA1 = fma(a[j].x, p.x, sA1); sA2 = fma(a[j].x, s2.x, sA2); }
        else { sP1 = fma(a[j].x, p.x, sP1); sP2 = fma(a[j].x, s2.x, sP2); }
        if (c + 1 < M) { sA1 = fma(a[j].y, p.y, sA1); sA2 = fma(a[j].y, s2.y, sA2); }
        else { sP1 = fma(a[j].y, p.y, sP1); sP2 = fma(a[j].y, s2.y, sP2); }
      }
      if ((N & 1) && lane == 0) {  // last column of an odd width (always a variable column)
        const double a = row[N - 1];
        sP1 = fma(a, lv[0][N - 1], sP1);
        sP2 = fma(a, lv[1][N - 1], sP2);
      }
      a8[4 * h] = sA1; a8[4 * h + 1] = sA2; a8[4 * h + 2] = sP1; a8[4 * h + 3] = sP2;
    }
    const double ws8 = wave_tsum<8>(a8, lane);
    if (lane < 8) {
      const int e = coop_row<8>(lane);  // = 4 h + quantity
      if ((e >> 2) < RPW) cpart[e & 3][0][RPW * wave + (e >> 2)] = ws8;
    }
  }
  __syncthreads();
  if (*s_fail) return 0;
  if (t < RW) {
#pragma unroll
    for (int k = 0; k < NQ; k++) qrow[t][k] = (k == 4 || k == 5) ? -1.7e308 : 0.0;
    if (own) {
      const double sA1 = cpart[0][0][t], sA2 = cpart[1][0][t], sP1 = cpart[2][0][t], sP2 = cpart[3][0][t];
      if (con) {
        const double ei = d.Einv[r];
        const bool uinf = up > QP_INFTY * QP_MIN_SCALING, linf = lo < -QP_INFTY * QP_MIN_SCALING;
        const double adx = ei * sP2;
        qrow[t][0] = fabs(ei * (sP1 - sa));
        qrow[t][1] = fabs(ei * sP1);
        qrow[t][2] = fabs(ei * sa);
        qrow[t][3] = fabs(d.E[r] * vproj);
        qrow[t][4] = uinf ? -1.7e308 : adx;
        qrow[t][5] = -(linf ? 1.7e308 : adx);
        qrow[t][13] = up * fmax(vproj, 0.0) + lo * fmin(vproj, 0.0);
      } else {
        const int i = r - M;
        const double di = d.Dinv[i], px = sP1, aty = sA1;
        qrow[t][6] = fabs(di * (px + sb + aty));
        qrow[t][7] = fabs(di * px);
        qrow[t][8] = fabs(di * aty);
        qrow[t][9] = fabs(di * sb);
        qrow[t][10] = fabs(di * sP2);
        qrow[t][11] = fabs(di * sA2);
        qrow[t][12] = fabs(d.D[i] * delta);
        qrow[t][14] = sb * delta;
        qrow[t][15] = sa * px;
        qrow[t][16] = sb * sa;
      }
    }
  }
  __syncthreads();
  if (t < NQ) {
    double rq = qrow[0][t];
    for (int w = 1; w < RW; w++) rq = t < NQ_MAX ? fmax(rq, qrow[w][t]) : rq + qrow[w][t];
    ll_publish(d.coop_q + (size_t)blockIdx.x * COOP_QS + 2 * t, rq, tag);
  }
  // Two hops instead of one wide all-gather (an exchange costs by the cache line): workgroup g < NL
  // reduces the partial results of workgroups g, g + NL, g + 2 NL, ... and publishes the group's;
  // then everybody gathers the NL group results.  Fixed order, same arithmetic in every workgroup.
  constexpr int NL = 16, NMEM = 256 / NL;  // at most 256 workgroups
  auto collect = [&](const unsigned long long *src, int first, int step, int count) {
    // entries (member m, quantity q), m < count: source workgroup first + m * step; result in qall[m * NQ + q]
    constexpr int KC = (NMEM * NQ + COOP_B - 1) / COOP_B;
    ll_u4 w[KC];
    bool have[KC];
    const unsigned long long *ptr[KC];
#pragma unroll
    for (int k = 0; k < KC; k++) {
      const int e = t + k * COOP_B;
      have[k] = e >= count * NQ;
      ptr[k] = src + (size_t)(first + (e / NQ) * step) * COOP_QS + 2 * (e % NQ);
    }
    unsigned spins = 0;
    for (;;) {
#pragma unroll
      for (int k = 0; k < KC; k++)
        if (!have[k]) w[k] = ll_peek(ptr[k]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bool all = true;
#pragma unroll
      for (int k = 0; k < KC; k++)
        if (!have[k]) {
          if (w[k].y == tag && w[k].w == tag) {
            qall[t + k * COOP_B] = __hiloint2double((int)w[k].z, (int)w[k].x);
            have[k] = true;
          } else {
            all = false;
          }
        }
      if (all) break;
      if (++spins > COOP_SPIN_LIMIT) {
        *s_fail = 1;
        d.ctrl->pad = 4 + (src != d.coop_q);
        break;
      }
    }
  };
  const int nl = T < NL ? T : NL;
  if ((int)blockIdx.x < nl) {
    const int count = (T - (int)blockIdx.x + nl - 1) / nl;
    collect(d.coop_q, blockIdx.x, nl, count);
    __syncthreads();
    if (t < NQ) {
      double rq = qall[t];
      for (int m = 1; m < count; m++) rq = t < NQ_MAX ? fmax(rq, qall[m * NQ + t]) : rq + qall[m * NQ + t];
      ll_publish(d.coop_q + (size_t)(256 + blockIdx.x) * COOP_QS + 2 * t, rq, tag);
    }
    __syncthreads();  // qall is reused below
  }
  collect(d.coop_q + (size_t)256 * COOP_QS, 0, 1, nl);
  __syncthreads();
  if (*s_fail) return 0;
  if (t < NQ) {
    double rq = qall[t];
    for (int g = 1; g < nl; g++) rq = t < NQ_MAX ? fmax(rq, qall[g * NQ + t]) : rq + qall[g * NQ + t];
    qres[t] = rq;
  }
  __syncthreads();
  if (t == 0) {
    Norms nm{qres[0], qres[1], qres[2], qres[3], qres[4], -qres[5], qres[6] * d.cinv, qres[7], qres[8], qres[9],
             qres[10], qres[11], qres[12], qres[13], qres[14], qres[15], qres[16]};
    double obj;
    const int st = decide_status(d, nm, obj);
    if (blockIdx.x == 0) {
      Ctrl *c = d.ctrl;
      c->iter = it;
      c->pri_res = nm.pri;
      c->dua_res = nm.dua;
      c->obj_val = obj;
      if (st) {
        c->status = st;
        c->done = 1;
      }
    }
    s_status = st;
  }
  __syncthreads();
  return s_status;
}

template <int COOP_B, int RW, int CPT>
__global__ __launch_bounds__(COOP_B) void k_coop(Dev d, int max_iter, int check_every, int final_check) {
  if (d.ctrl->done) return;
  constexpr int NW = COOP_B / 64;
  __shared__ __attribute__((aligned(16))) double part[2][RW][NW];
  __shared__ int s_fail;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int M = d.M, N = d.n + d.M, T = gridDim.x;
  const int r0 = blockIdx.x * RW;
  const unsigned base = *d.coop_tag;
  const bool prof = d.prof != nullptr;
  double Kr[RW][CPT], v[CPT];
  int slot[CPT];  // word offset of this thread's columns inside an exchange buffer
#pragma unroll
  for (int k = 0; k < CPT; k++) {
    const int c = t + k * COOP_B;
    v[k] = c < N ? d.wh[c] : 0.0;
    slot[k] = (c / RW) * d.coop_stride + 2 * (c % RW);
#pragma unroll
    for (int rw = 0; rw < RW; rw++) Kr[rw][k] = (c < N && r0 + rw < N) ? d.W[(size_t)(r0 + rw) * d.ldw + c] : 0.0;
  }
  // iterates of the row this thread owns (threads 0..RW-1)
  const int r = r0 + t;
  const bool own = t < RW && r < N, con = r < M;
  double sa = 0.0, sb = 0.0, lo = 0.0, up = 0.0, sw = 0.0, delta = 0.0;  // (z, y, l, u, wh, dy) or (x, q, -, -, -, dx)
  if (own) {
    if (con) { sa = d.z[r]; sb = d.y[r]; lo = d.l[r]; up = d.u[r]; sw = d.wh[r]; }
    else { sa = d.x[r - M]; sb = d.q[r - M]; }
  }
  // start-up: every workgroup registers and waits for all the others -- the proof that the whole
  // grid is resident before anybody starts to depend on it
  if (t == 0) {
    s_fail = 0;
    __hip_atomic_fetch_add(d.coop_reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(d.coop_reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)T) {
      if (++spins > COOP_SPIN_LIMIT) {
        s_fail = 1;
        d.ctrl->pad = 1;
        break;
      }
    }
  }
  __syncthreads();
  const double rho = d.rho, rinv = d.rho_inv, alpha = d.alpha, sigma = d.sigma;
  const size_t my_slot = (size_t)blockIdx.x * d.coop_stride + 2 * t;

  // all CPT entries of this thread's columns from one exchange buffer; waits for the LAST column
  // alone first (fewer requests in flight while nothing has arrived yet)
  auto gather = [&](const unsigned long long *buf, unsigned tag, double (&out)[CPT]) {
    bool have[CPT];
#pragma unroll
    for (int k = 0; k < CPT; k++) have[k] = t + k * COOP_B >= N;
    unsigned spins = 0;
    if (!have[CPT - 1] && !(d.coop_dbg & 32)) {
      for (;;) {
        const ll_u4 w = ll_peek(buf + slot[CPT - 1]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (w.y == tag && w.w == tag) {
          out[CPT - 1] = __hiloint2double((int)w.z, (int)w.x);
          have[CPT - 1] = true;
          break;
        }
        if (++spins > COOP_SPIN_LIMIT) break;
      }
    }
    for (;;) {
      ll_u4 w[CPT];
#pragma unroll
      for (int k = 0; k < CPT; k++)
        if (!have[k]) w[k] = ll_peek(buf + slot[k]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bool all = true;
#pragma unroll
      for (int k = 0; k < CPT; k++)
        if (!have[k]) {
          if (w[k].y == tag && w[k].w == tag) {
            out[k] = __hiloint2double((int)w[k].z, (int)w[k].x);
            have[k] = true;
          } else {
            all = false;
          }
        }
      if (all) break;
      if (++spins > COOP_SPIN_LIMIT) {
        s_fail = 1;
        d.ctrl->pad = 2;
        break;
      }
    }
  };

  long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ph4 = 0, ph5 = 0, nchk = 0;  // debug phase clocks (thread 0, d.prof set)
  int it = 0, status = 0, to_test = check_every > 0 ? check_every : -1;
  for (it = 1; it <= max_iter; it++) {
    const unsigned tag = base + (unsigned)it;
    bool chk = final_check && it == max_iter;
    if (--to_test == 0) {
      chk = true;
      to_test = check_every;
    }
    const long long c0 = prof ? clock64() : 0;
    double acc[RW];
#pragma unroll
    for (int rw = 0; rw < RW; rw++) {
      double a = Kr[rw][0] * v[0];
#pragma unroll
      for (int k = 1; k < CPT; k++) a = fma(Kr[rw][k], v[k], a);
      acc[rw] = a;
    }
    const double ws = wave_tsum<RW>(acc, lane);
    if (lane < RW) part[it & 1][coop_row<RW>(lane)][wave] = ws;
    __syncthreads();
    if (s_fail) break;
    const long long c1 = prof ? clock64() : 0;
    unsigned long long *buf = d.coop_buf + (size_t)(tag & 1u) * d.coop_half;
    double vproj = 0.0;
    if (own) {
      double s = part[it & 1][t][0];
#pragma unroll
      for (int w = 1; w < NW; w++) s += part[it & 1][t][w];
      double pub;
      if (con) {
        // z~ = z + (nu - y) / rho with nu = -rho wh + s, relaxed and projected; the entry to publish,
        // wh+ = z+ - y+ / rho = 2 z+ - (z_r + y / rho), is formed before y+ (shortest path to the store)
        const double zt = sa + rinv * ((s - rho * sw) - sb);
        const double zr = alpha * zt + (1.0 - alpha) * sa;
        const double vv = zr + rinv * sb;
        const double zn = fmin(fmax(vv, lo), up);
        pub = 2.0 * zn - vv;
        ll_publish(buf + my_slot, pub, tag);
        delta = rho * (zr - zn);
        sa = zn;
        sb += delta;
        sw = pub;
      } else {
        const double xn = alpha * s + (1.0 - alpha) * sa;
        pub = sigma * xn - sb;
        ll_publish(buf + my_slot, pub, tag);
        delta = xn - sa;
        sa = xn;
      }
      if (chk) {  // the test's operands travel with the same round: [y ; x] and [proj(dy) ; dx]
        if (con) {
          const bool uinf = up > QP_INFTY * QP_MIN_SCALING, linf = lo < -QP_INFTY * QP_MIN_SCALING;
          vproj = delta;
          if (uinf && linf) vproj = 0.0;
          else if (uinf) vproj = fmin(vproj, 0.0);
          else if (linf) vproj = fmax(vproj, 0.0);
        }
        ll_publish(d.coop_chk + 2 * my_slot, con ? sb : sa, tag);
        ll_publish(d.coop_chk + 2 * my_slot + 2, con ? vproj : delta, tag);
      }
    }
    const long long c2 = prof ? clock64() : 0;
    if (!(d.coop_dbg & 1)) gather(buf, tag, v);
    if (prof) {
      ph0 += c1 - c0; ph1 += c2 - c1; ph2 += clock64() - c2;
    }
    if (!chk) continue;

    // ---- termination test ----
    const long long k0 = prof ? clock64() : 0;
    {
      const CoopTest ta{d.coop_chk, d.coop_q, d.Kc, d.Einv, d.E, d.Dinv, d.D, d.ctrl, d.n, d.M, d.ldw, d.coop_stride,
                        d.c, d.cinv, d.eps_abs, d.eps_rel, d.eps_pinf, d.eps_dinf};
      status = coop_test<COOP_B, RW, CPT>(ta, tag, it, own, con, sa, sb, lo, up, delta, &s_fail);
    }
    if (prof) {
      ph3 += clock64() - k0;
      nchk++;
    }
    if (s_fail) break;
    if (status) break;
  }
  if (it > max_iter) it = max_iter;
  if (prof && t == 0) {
    unsigned long long *o = d.prof + 8 * blockIdx.x;
    o[0] = ph0; o[1] = ph1; o[2] = ph2; o[3] = it; o[4] = ph3; o[5] = ph4; o[6] = nchk; o[7] = ph5;
  }
  if (own) {
    if (con) { d.z[r] = sa; d.y[r] = sb; d.dy[r] = delta; d.wh[r] = sw; }
    else { d.x[r - M] = sa; d.dx[r - M] = delta; d.rx[r - M] = sigma * sa - sb; }
  }
  if (blockIdx.x == 0 && t == 0) {
    *d.coop_tag = base + (unsigned)it;
    if (!final_check) d.ctrl->iter = it;
  }
}

// Kc = [ 0  Abar ; Abar^T  Pbar ] dense, N x ldw, from the dense copies of the scaled matrices
__global__ void k_build_kc(Dev d, double *Kc) {
  const int N = d.n + d.M, M = d.M;
  const int rr = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  double a;
  if (rr < M) a = c < M ? 0.0 : d.f_Ad[(size_t)rr * d.ldn + (c - M)];
  else a = c < M ? d.f_Atd[(size_t)(rr - M) * d.ldm + c] : d.f_Pd[(size_t)(rr - M) * d.ldn + (c - M)];
  Kc[(size_t)rr * d.ldw + c] = a;
}

// ------------------------------------------------------------------------------------------
// LDS-resident solver for small problems (BASELINE configs 1 and 4): ONE workgroup keeps the
// product-form factor -- the n x (M+n) matrix [ -G | strict_lower(Linv) ] -- and every iterate
// in LDS and runs the WHOLE ADMM loop, termination tests included, in a single launch; the two
// all-to-all exchanges of an iteration become two __syncthreads.  The forward sweep reads the
// matrix by rows, the backward sweep reads the SAME matrix by columns (x rows: Linv^T; constraint
// rows: (-G)^T); the row stride is odd, so both directions are bank-conflict free.
// ------------------------------------------------------------------------------------------
constexpr int RES_THREADS = 512, RES_WAVES = RES_THREADS / 64;

__host__ __device__ inline size_t resident_lds_doubles(int n, int M) {
  const size_t ldr = (size_t)((M + n) | 1);
  return (size_t)n * ldr + (size_t)(M + n) + 5 * (size_t)n + 6 * (size_t)M + 8 * (size_t)M + 4 * (size_t)n +
         (size_t)NQ * RES_WAVES + NQ + 8;
}

template <int NV>
__device__ __forceinline__ void group_reduce(double (&v)[NV], int tg) {
  group_sum<NV>(v, tg);
}

// strided dot product over LDS operands with four independent chains (LDS latency, not
// bandwidth, bounds a single-workgroup solver): sum_{k = k0, k0+st, ... < k1} a[k*sa] * b[k]
__device__ __forceinline__ double lds_dot(const double *a, size_t sa, const double *b, int k0, int k1, int st) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int k = k0;
  for (; k + 3 * st < k1; k += 4 * st) {
    s0 = fma(a[(size_t)k * sa], b[k], s0);
    s1 = fma(a[(size_t)(k + st) * sa], b[k + st], s1);
    s2 = fma(a[(size_t)(k + 2 * st) * sa], b[k + 2 * st], s2);
    s3 = fma(a[(size_t)(k + 3 * st) * sa], b[k + 3 * st], s3);
  }
  for (; k < k1; k += st) s0 = fma(a[(size_t)k * sa], b[k], s0);
  return (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(RES_THREADS) void k_resident(Dev d, int max_iter, int check_every, int final_check,
                                                          int TG1, int TG2) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = d.n, M = d.M, tid = threadIdx.x;
  const int ldr = (M + n) | 1;
  double *Rm = sm;
  double *wr = Rm + (size_t)n * ldr;  // [wh (M) | rx (n)]
  double *x = wr + M + n, *q = x + n, *d2 = q + n, *ut = d2 + n, *dx = ut + n;
  double *z = dx + n, *y = z + M, *l = y + M, *u = l + M, *dy = u + M, *vp = dy + M;
  double *smc = vp + M;       // 8 x M
  double *snc = smc + 8 * M;  // 4 x n
  double *part = snc + 4 * n; // NQ x waves
  double *res = part + NQ * RES_WAVES;
  const double rho = d.rho, rinv = d.rho_inv, sigma = d.sigma, alpha = d.alpha;
  const unsigned long long clk0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
  // ---- load factor and state ----
  for (int row = tid >> 6; row < n; row += RES_WAVES) {
    const double *src = d.f_rows + (size_t)row * d.ldf;
    for (int k = tid & 63; k < ldr; k += 64) Rm[(size_t)row * ldr + k] = k < M + row ? src[k] : 0.0;
  }
  for (int i = tid; i < n; i += RES_THREADS) {
    const double xi = d.x[i], qi = d.q[i];
    x[i] = xi;
    q[i] = qi;
    d2[i] = d.d2inv[i];
    dx[i] = 0.0;
    wr[M + i] = sigma * xi - qi;
  }
  for (int j = tid; j < M; j += RES_THREADS) {
    const double zj = d.z[j], yj = d.y[j];
    z[j] = zj;
    y[j] = yj;
    l[j] = d.l[j];
    u[j] = d.u[j];
    dy[j] = 0.0;
    wr[j] = zj - rinv * yj;
  }
  __syncthreads();
  const int NG1 = RES_THREADS / TG1, NG2 = RES_THREADS / TG2;
  const int g1 = tid / TG1, t1 = tid % TG1, g2 = tid / TG2, t2 = tid % TG2;
  int status = 0, it = 0;
  // termination test on the LDS-resident iterates (sparse Abar / Pbar rows come from L2)
  auto run_check = [&](int iter_now) -> int {
    for (int j = g1; j < M; j += NG1) {
      double acc[2] = {0.0, 0.0};
      for (int k = d.pc_ptr[j] + t1; k < d.pc_ptr[j + 1]; k += TG1) {
        const double a = d.pc_A[k];
        const int c = d.pc_idx[k];
        acc[0] = fma(a, x[c], acc[0]);
        acc[1] = fma(a, dx[c], acc[1]);
      }
      group_reduce<2>(acc, TG1);
      if (t1 == 0) {
        const double ei = d.Einv[j], zz = z[j], lo = l[j], hi = u[j];
        const bool uinf = hi > QP_INFTY * QP_MIN_SCALING, linf = lo < -QP_INFTY * QP_MIN_SCALING;
        double v = dy[j];
        if (uinf && linf) v = 0.0;
        else if (uinf) v = fmin(v, 0.0);
        else if (linf) v = fmax(v, 0.0);
        const double adx = ei * acc[1];
        smc[0 * M + j] = ei * (acc[0] - zz);
        smc[1 * M + j] = ei * acc[0];
        smc[2 * M + j] = ei * zz;
        vp[j] = v;
        smc[4 * M + j] = d.E[j] * v;
        smc[5 * M + j] = hi * fmax(v, 0.0) + lo * fmin(v, 0.0);
        smc[6 * M + j] = uinf ? -1.7e308 : adx;
        smc[7 * M + j] = linf ? 1.7e308 : adx;
      }
    }
    __syncthreads();
    for (int i = g1; i < n; i += NG1) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = d.pb_ptr[i] + t1; k < d.pb_ptr[i + 1]; k += TG1) {
        const double a = d.pb_val[k];
        const int c = d.pb_idx[k];
        acc[0] = fma(a, x[c], acc[0]);
        acc[1] = fma(a, dx[c], acc[1]);
      }
      for (int k = d.pv_ptr[i] + t1; k < d.pv_ptr[i + 1]; k += TG1) {
        const double a = d.pv_At[k];
        const int c = d.pv_idx[k];
        acc[2] = fma(a, y[c], acc[2]);
        acc[3] = fma(a, vp[c], acc[3]);
      }
      group_reduce<4>(acc, TG1);
      if (t1 == 0) {
        snc[0 * n + i] = acc[0];
        snc[1 * n + i] = d.Dinv[i] * acc[1];
        snc[2 * n + i] = acc[2];
        snc[3 * n + i] = d.Dinv[i] * acc[3];
      }
    }
    __syncthreads();
    double v[NQ];
#pragma unroll
    for (int k = 0; k < NQ; k++) v[k] = 0.0;
    v[4] = -1.7e308;
    v[5] = -1.7e308;
    for (int j = tid; j < M; j += RES_THREADS) {
      v[0] = fmax(v[0], fabs(smc[0 * M + j]));
      v[1] = fmax(v[1], fabs(smc[1 * M + j]));
      v[2] = fmax(v[2], fabs(smc[2 * M + j]));
      v[3] = fmax(v[3], fabs(smc[4 * M + j]));
      v[4] = fmax(v[4], smc[6 * M + j]);
      v[5] = fmax(v[5], -smc[7 * M + j]);
      v[13] += smc[5 * M + j];
    }
    for (int i = tid; i < n; i += RES_THREADS) {
      const double di = d.Dinv[i], px = snc[0 * n + i], aty = snc[2 * n + i], qi = q[i], xi = x[i], dxi = dx[i];
      v[6] = fmax(v[6], fabs(di * (px + qi + aty)));
      v[7] = fmax(v[7], fabs(di * px));
      v[8] = fmax(v[8], fabs(di * aty));
      v[9] = fmax(v[9], fabs(di * qi));
      v[10] = fmax(v[10], fabs(snc[1 * n + i]));
      v[11] = fmax(v[11], fabs(snc[3 * n + i]));
      v[12] = fmax(v[12], fabs(d.D[i] * dxi));
      v[14] += qi * dxi;
      v[15] += xi * px;
      v[16] += qi * xi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const double o = __shfl_xor(v[k], off, 64);
        v[k] = k < NQ_MAX ? fmax(v[k], o) : v[k] + o;
      }
    }
    if ((tid & 63) == 0) {
#pragma unroll
      for (int k = 0; k < NQ; k++) part[k * RES_WAVES + (tid >> 6)] = v[k];
    }
    __syncthreads();
    if (tid < NQ) {
      double r = part[tid * RES_WAVES];
      for (int w = 1; w < RES_WAVES; w++)
        r = tid < NQ_MAX ? fmax(r, part[tid * RES_WAVES + w]) : r + part[tid * RES_WAVES + w];
      res[tid] = r;
    }
    __syncthreads();
    if (tid == 0) {
      Norms nm{res[0], res[1], res[2], res[3], res[4], -res[5], res[6] * d.cinv, res[7], res[8], res[9], res[10],
               res[11], res[12], res[13], res[14], res[15], res[16]};
      double obj;
      const int st = decide_status(d, nm, obj);
      Ctrl *c = d.ctrl;
      c->iter = iter_now;
      c->pri_res = nm.pri;
      c->dua_res = nm.dua;
      c->obj_val = obj;
      if (st) {
        c->status = st;
        c->done = 1;
      }
      res[NQ] = (double)st;
    }
    __syncthreads();
    return (int)res[NQ];
  };

  bool checked = false;
  for (it = 1; it <= max_iter; it++) {
    // forward sweep: rows of L^-1
    for (int row = g1; row < n; row += NG1) {
      const double *r = Rm + (size_t)row * ldr;
      double acc[1] = {lds_dot(r, 1, wr, t1, M + row, TG1)};
      group_reduce<1>(acc, TG1);
      if (t1 == 0) ut[row] = d2[row] * (wr[M + row] + acc[0]);
    }
    __syncthreads();
    // backward sweep: columns of the same matrix, fused x / z / y update
    for (int r = g2; r < n + M; r += NG2) {
      double acc[1] = {0.0};
      if (r < n) {
        acc[0] = lds_dot(Rm + M + r, ldr, ut, r + 1 + t2, n, TG2);
        group_reduce<1>(acc, TG2);
        if (t2 == 0) {
          const double xt = ut[r] + acc[0], xp = x[r];
          const double xn = alpha * xt + (1.0 - alpha) * xp;
          x[r] = xn;
          dx[r] = xn - xp;
          wr[M + r] = sigma * xn - q[r];
        }
      } else {
        const int j = r - n;
        acc[0] = lds_dot(Rm + j, ldr, ut, t2, n, TG2);
        group_reduce<1>(acc, TG2);
        if (t2 == 0) {
          const double zp = z[j], yp = y[j];
          const double nu = -rho * wr[j] + acc[0];
          const double zt = zp + rinv * (nu - yp);
          const double zr = alpha * zt + (1.0 - alpha) * zp;
          const double v = zr + rinv * yp;
          const double zn = fmin(fmax(v, l[j]), u[j]);
          const double dyj = rho * (zr - zn);
          const double yn = yp + dyj;
          z[j] = zn;
          y[j] = yn;
          dy[j] = dyj;
          wr[j] = zn - rinv * yn;
        }
      }
    }
    __syncthreads();
    checked = false;
    if (check_every > 0 && it % check_every == 0) {
      checked = true;
      status = run_check(it);
      if (status) break;
    }
  }
  if (it > max_iter) it = max_iter;
  if (!status && !checked && final_check) status = run_check(it);
  if (tid == 0 && !final_check) d.ctrl->iter = it;
  if (tid == 0) {  // shader cycles and 100 MHz wall ticks of this launch (clock diagnosis)
    d.ctrl->nrm_dy = (double)(__builtin_readcyclecounter() - clk0);
    d.ctrl->nrm_dx = (double)(wall_clock64() - wall0);
  }
  // ---- write the iterates back ----
  for (int i = tid; i < n; i += RES_THREADS) {
    d.x[i] = x[i];
    d.dx[i] = dx[i];
    d.rx[i] = wr[M + i];
  }
  for (int j = tid; j < M; j += RES_THREADS) {
    d.z[j] = z[j];
    d.y[j] = y[j];
    d.dy[j] = dy[j];
    d.wh[j] = wr[j];
  }
}

// ------------------------------------------------------------------------------------------
// per-solve prologue / epilogue
// ------------------------------------------------------------------------------------------
__global__ void k_reset_ctrl(Dev d) {
  Ctrl *c = d.ctrl;
  c->done = 0;
  c->pad = 0;  // cooperative solver: exchange timed out
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
  if (j < d.M) d.wh[j] = d.z[j] - d.rho_inv * d.y[j];
  if (j < d.n) d.rx[j] = d.sigma * d.x[j] - d.q[j];
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
      if (d.digest) {
        // is_int_feas + pick_nextvar (workspace.py:245-264, 205-230) and the rounded candidate of
        // get_integer_solution (workspace.py:266-272); np.round == rint (half to even)
        __syncthreads();
        for (int i = tid; i < n; i += 1024) d.xi[i] = d.out_x[i];
        __syncthreads();
        int cnt = 0, bestk = 0x7fffffff;
        double best = -1.0;
        for (int k = tid; k < d.n_int; k += 1024) {
          const int i = d.i_idx[k];
          const double v = d.out_x[i], r = rint(v), f = fabs(v - r);
          d.xi[i] = r;
          cnt += f > d.eps_int;
          if (f > best) { best = f; bestk = k; }
        }
        for (int off = 32; off > 0; off >>= 1) {
          const double ob = __shfl_xor(best, off, 64);
          const int ok = __shfl_xor(bestk, off, 64);
          cnt += __shfl_xor(cnt, off, 64);
          if (ob > best || (ob == best && ok < bestk)) { best = ob; bestk = ok; }
        }
        __shared__ double sb[16];
        __shared__ int sk[16], sc[16];
        if ((tid & 63) == 0) { sb[tid >> 6] = best; sk[tid >> 6] = bestk; sc[tid >> 6] = cnt; }
        __syncthreads();
        if (tid == 0) {
          for (int w = 1; w < 16; w++) {
            cnt += sc[w];
            if (sb[w] > best || (sb[w] == best && sk[w] < bestk)) { best = sb[w]; bestk = sk[w]; }
          }
          c->int_inf = cnt;
          c->nextvar = bestk == 0x7fffffff ? -1 : bestk;
        }
        for (int i = tid; i < n; i += 1024) d.xis[i] = d.Dinv[i] * d.xi[i];
      }
    }
  }
  if (tid == 0) {
    c->status = st;
    c->done = 1;
    if (!(node && d.digest && (st == MIOSQP_QP_SOLVED || st == MIOSQP_QP_MAX_ITER_REACHED))) {
      c->int_inf = -1;
      c->nextvar = -1;
    }
    (void)max_iter;
  }
}

// rounding heuristic: rows of A against the rounded candidate, worst violation of the ROOT bounds
// with the eps_abs slack of satisfies_lin_constraints (workspace.py:232-243)
template <int TPR>
__global__ __launch_bounds__(256) void k_heur_rows(Dev d) {
  ROW_SETUP(TPR)
  const bool live = row_raw < d.M;
  const int row = live ? row_raw : d.M - 1;
  double acc[1];
  acc[0] = prow_dot<TPR>(d.pc_idx, d.pc_A, d.pc_ptr[row], d.pc_ptr[row + 1], t, d.xis);
  row_reduce<TPR, 1>(acc, lds);
  if (live && t == 0) {
    const double z = d.Einv[row] * acc[0];
    d.sm[row] = fmax(d.root_l[row] - d.eps_lin - z, z - d.root_u[row] - d.eps_lin);
  }
}

// rows of the unscaled P: t_i = x_i (0.5 (P x)_i + q_i)   (data.py:99-103)
template <int TPR>
__global__ __launch_bounds__(256) void k_obj_rows(Dev d) {
  ROW_SETUP(TPR)
  const bool live = row_raw < d.n;
  const int row = live ? row_raw : d.n - 1;
  double acc[2];
  prow_dot2<TPR>(d.pr_idx, d.pr_val, d.pr_ptr[row], d.pr_ptr[row + 1], t, d.out_x, d.digest ? d.xi : d.out_x, acc[0],
                 acc[1]);
  row_reduce<TPR, 2>(acc, lds);
  if (live && t == 0) {
    d.sn[row] = d.out_x[row] * (0.5 * acc[0] + d.qraw[row]);
    if (d.digest) d.sn[d.n + row] = d.xi[row] * (0.5 * acc[1] + d.qraw[row]);
  }
}

__global__ __launch_bounds__(1024) void k_obj_sum(Dev d) {
  __shared__ double lds[16];
  double s = 0, s2 = 0, vmax = -1.7e308;
  for (int i = threadIdx.x; i < d.n; i += 1024) s += d.sn[i];
  s = block_sum(s, lds);
  if (d.digest) {
    for (int i = threadIdx.x; i < d.n; i += 1024) s2 += d.sn[d.n + i];
    for (int j = threadIdx.x; j < d.M; j += 1024) vmax = fmax(vmax, d.sm[j]);
    s2 = block_sum(s2, lds);
    vmax = block_max(vmax, lds);
  }
  if (threadIdx.x == 0) {
    const int st = d.ctrl->status;
    const bool ok = st == MIOSQP_QP_SOLVED || st == MIOSQP_QP_MAX_ITER_REACHED;
    d.ctrl->lower = ok ? s : __builtin_nan("");
    d.ctrl->heur_obj = ok && d.digest ? s2 : __builtin_nan("");
    d.ctrl->heur_viol = ok && d.digest ? vmax : __builtin_nan("");
  }
}

// ------------------------------------------------------------------------------------------
// batched mode: B independent nodes share the factor (leaves of one wave).  Vectors are
// [len][Bs] with the batch index fastest, so one wavefront = one matrix row x 64 nodes: the
// row's entries are wave-uniform (scalar loads, read once for 64 nodes) and every vector access
// is one coalesced 512-byte line.  No cross-lane reduction is needed at all.
// ------------------------------------------------------------------------------------------
#define BSETUP                                                                \
  const int lane = threadIdx.x & 63;                                          \
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     \
  const int b = blockIdx.y * 64 + lane;                                       \
  const size_t Bs = (size_t)d.Bs;

// sum_k val[k] * V[idx[k]][b] over one padded row (wave-uniform row)
__device__ __forceinline__ double brow_dot(const int *__restrict__ idx, const double *__restrict__ val, int s,
                                           int e, const double *__restrict__ V, size_t Bs) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = s;
  for (; k + 4 <= e; k += 4) {
    a0 = fma(val[k], V[(size_t)idx[k] * Bs], a0);
    a1 = fma(val[k + 1], V[(size_t)idx[k + 1] * Bs], a1);
    a2 = fma(val[k + 2], V[(size_t)idx[k + 2] * Bs], a2);
    a3 = fma(val[k + 3], V[(size_t)idx[k + 3] * Bs], a3);
  }
  for (; k < e; k += 2) {
    a0 = fma(val[k], V[(size_t)idx[k] * Bs], a0);
    a1 = fma(val[k + 1], V[(size_t)idx[k + 1] * Bs], a1);
  }
  return (a0 + a1) + (a2 + a3);
}
__device__ __forceinline__ void brow_dot2(const int *__restrict__ idx, const double *__restrict__ val, int s, int e,
                                          const double *__restrict__ V, const double *__restrict__ W, size_t Bs,
                                          double &rv, double &rw) {
  double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
  for (int k = s; k < e; k += 2) {
    const size_t o0 = (size_t)idx[k] * Bs, o1 = (size_t)idx[k + 1] * Bs;
    a0 = fma(val[k], V[o0], a0);
    a1 = fma(val[k + 1], V[o1], a1);
    c0 = fma(val[k], W[o0], c0);
    c1 = fma(val[k + 1], W[o1], c1);
  }
  rv = a0 + a1;
  rw = c0 + c1;
}
// dense contiguous row segment [j0, j1) against V[j][b]
__device__ __forceinline__ double bdense_dot(const double *__restrict__ row, int j0, int j1,
                                             const double *__restrict__ V, size_t Bs) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int j = j0;
  for (; j + 4 <= j1; j += 4) {
    a0 = fma(row[j], V[(size_t)j * Bs], a0);
    a1 = fma(row[j + 1], V[(size_t)(j + 1) * Bs], a1);
    a2 = fma(row[j + 2], V[(size_t)(j + 2) * Bs], a2);
    a3 = fma(row[j + 3], V[(size_t)(j + 3) * Bs], a3);
  }
  for (; j < j1; j++) a0 = fma(row[j], V[(size_t)j * Bs], a0);
  return (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void kb_panel_fwd(Dev d) {
  if (d.ctrl->done) return;
  BSETUP
  const int row = blockIdx.x * 4 + wv;
  if (row >= d.n) return;
  const double acc = brow_dot(d.pv_idx, d.pv_L, d.pv_ptr[row], d.pv_ptr[row + 1], d.b_wh + b, Bs);
  d.b_cv[row * Bs + b] = d.sigma * d.b_x[row * Bs + b] - d.q[row] - acc;
}

__global__ __launch_bounds__(256) void kb_tail_fwd(Dev d) {
  if (d.ctrl->done) return;
  BSETUP
  const int row = blockIdx.x * 4 + wv;
  if (row >= d.n) return;
  const double acc = bdense_dot(d.Linv + (size_t)row * d.ld, 0, row, d.b_cv + b, Bs);
  d.b_ut[row * Bs + b] = d.d2inv[row] * (d.b_cv[row * Bs + b] + acc);
}

__global__ __launch_bounds__(256) void kb_tail_bwd(Dev d) {
  if (d.ctrl->done) return;
  BSETUP
  const int row = blockIdx.x * 4 + wv;
  if (row >= d.n) return;
  const double acc = bdense_dot(d.LinvT + (size_t)row * d.ld, row + 1, d.n, d.b_ut + b, Bs);
  const double xt = d.b_ut[row * Bs + b] + acc;
  d.b_xt[row * Bs + b] = xt;
  if (!d.c_done[b]) {
    const double xp = d.b_x[row * Bs + b];
    const double xn = d.alpha * xt + (1.0 - d.alpha) * xp;
    d.b_x[row * Bs + b] = xn;
    d.b_dx[row * Bs + b] = xn - xp;
  }
}

__global__ __launch_bounds__(256) void kb_panel_bwd(Dev d) {
  if (d.ctrl->done) return;
  BSETUP
  const int row = blockIdx.x * 4 + wv;
  if (row >= d.M) return;
  const double acc = brow_dot(d.pc_idx, d.pc_L, d.pc_ptr[row], d.pc_ptr[row + 1], d.b_xt + b, Bs);
  if (d.c_done[b]) return;
  const size_t o = row * Bs + b;
  const double rho = d.rho, rinv = d.rho_inv, alpha = d.alpha;
  const double zp = d.b_z[o], yp = d.b_y[o];
  const double nu = -rho * d.b_wh[o] - acc;
  const double zt = zp + rinv * (nu - yp);
  const double zr = alpha * zt + (1.0 - alpha) * zp;
  const double v = zr + rinv * yp;
  const double zn = fmin(fmax(v, d.b_l[o]), d.b_u[o]);
  const double dy = rho * (zr - zn);
  const double yn = yp + dy;
  d.b_z[o] = zn;
  d.b_y[o] = yn;
  d.b_dy[o] = dy;
  d.b_wh[o] = zn - rinv * yn;
}

__global__ __launch_bounds__(256) void kb_check_con(Dev d) {
  if (d.ctrl->done) return;
  BSETUP
  const int row = blockIdx.x * 4 + wv;
  if (row >= d.M) return;
  double ax, adx0;
  brow_dot2(d.pc_idx, d.pc_A, d.pc_ptr[row], d.pc_ptr[row + 1], d.b_x + b, d.b_dx + b, Bs, ax, adx0);
  const size_t o = row * Bs + b, MB = (size_t)d.M * Bs;
  const double ei = d.Einv[row], z = d.b_z[o], l = d.b_l[o], u = d.b_u[o];
  const bool uinf = u > QP_INFTY * QP_MIN_SCALING, linf = l < -QP_INFTY * QP_MIN_SCALING;
  double v = d.b_dy[o];
  if (uinf && linf) v = 0.0;
  else if (uinf) v = fmin(v, 0.0);
  else if (linf) v = fmax(v, 0.0);
  const double adx = ei * adx0;
  d.b_sm[0 * MB + o] = ei * (ax - z);
  d.b_sm[1 * MB + o] = ei * ax;
  d.b_sm[2 * MB + o] = ei * z;
  d.b_sm[3 * MB + o] = v;
  d.b_sm[4 * MB + o] = d.E[row] * v;
  d.b_sm[5 * MB + o] = u * fmax(v, 0.0) + l * fmin(v, 0.0);
  d.b_sm[6 * MB + o] = uinf ? -1.7e308 : adx;
  d.b_sm[7 * MB + o] = linf ? 1.7e308 : adx;
}

// blockIdx.x in [0, nrb): P rows; [nrb, 2 nrb): A^T rows
__global__ __launch_bounds__(256) void kb_check_var(Dev d) {
  if (d.ctrl->done) return;
  BSETUP
  const int nrb = (d.n + 3) / 4;
  const bool second = (int)blockIdx.x >= nrb;
  const int row = ((int)blockIdx.x - (second ? nrb : 0)) * 4 + wv;
  if (row >= d.n) return;
  const size_t o = row * Bs + b, NB = (size_t)d.n * Bs;
  double r0, r1;
  if (!second) {
    brow_dot2(d.pb_idx, d.pb_val, d.pb_ptr[row], d.pb_ptr[row + 1], d.b_x + b, d.b_dx + b, Bs, r0, r1);
    d.b_sn[0 * NB + o] = r0;
    d.b_sn[1 * NB + o] = d.Dinv[row] * r1;
  } else {
    brow_dot2(d.pv_idx, d.pv_At, d.pv_ptr[row], d.pv_ptr[row + 1], d.b_y + b, d.b_sm + 3 * (size_t)d.M * Bs + b, Bs,
              r0, r1);
    d.b_sn[2 * NB + o] = r0;
    d.b_sn[3 * NB + o] = d.Dinv[row] * r1;
  }
}

// ------------------------------------------------------------------------------------------
// batched mode on the product-form factor: dense row blocks.  One workgroup = 8 waves =
// 4 row groups x 2 halves of the k range; a wave owns 2 rows x 64 nodes.  The right-hand tile
// V[k0..k0+16)[64 nodes] is staged once per workgroup in LDS (double-buffered, next tile's global
// loads in flight during the FMAs), matrix entries are wave-uniform scalar loads, each LDS value
// feeds both rows.  The two k halves are added in LDS in a fixed order.
// ------------------------------------------------------------------------------------------
constexpr int BD_KT = 16, BD_R = 2;

// RG row groups x KS slices of the k range = RG*KS waves per workgroup, 2*RG rows per workgroup
template <int RG, int KS>
__device__ __forceinline__ void bd_dot(const double *__restrict__ A, int ld, int nrows, int row0, int kbeg,
                                       int kend, const double *__restrict__ V, size_t Bs, double *lds,
                                       double (&acc)[BD_R]) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rg = w % RG, kh = w / RG;
  constexpr int LPT = BD_KT / RG;  // tile rows staged per wave
  const int k0 = kbeg & ~(BD_KT - 1);
  const int nt_slice = ((kend - k0 + BD_KT - 1) / BD_KT + KS - 1) / KS;
  const int kstart = k0 + kh * nt_slice * BD_KT;
  double *buf = lds + kh * (2 * BD_KT * 64);
  int ra = row0 + rg * BD_R, rb = ra + 1;
  ra = ra < nrows ? ra : nrows - 1;
  rb = rb < nrows ? rb : nrows - 1;
  const double *__restrict__ a0 = A + (size_t)ra * ld;
  const double *__restrict__ a1 = A + (size_t)rb * ld;
  acc[0] = acc[1] = 0.0;
  double p[LPT];
#pragma unroll
  for (int i = 0; i < LPT; i++) {
    const int k = kstart + rg + RG * i;
    p[i] = k < kend ? V[(size_t)k * Bs] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < LPT; i++) buf[(rg + RG * i) * 64 + lane] = p[i];
  __syncthreads();
  for (int t = 0; t < nt_slice; t++) {
    const int kt = kstart + t * BD_KT;
    const bool more = t + 1 < nt_slice;
    if (more) {
#pragma unroll
      for (int i = 0; i < LPT; i++) {
        const int k = kt + BD_KT + rg + RG * i;
        p[i] = k < kend ? V[(size_t)k * Bs] : 0.0;
      }
    }
    const double *cur = buf + (t & 1) * (BD_KT * 64);
#pragma unroll
    for (int kk = 0; kk < BD_KT; kk++) {
      const double v = cur[kk * 64 + lane];
      acc[0] = fma(a0[kt + kk], v, acc[0]);
      acc[1] = fma(a1[kt + kk], v, acc[1]);
    }
    if (more) {
      double *nxt = buf + ((t + 1) & 1) * (BD_KT * 64);
#pragma unroll
      for (int i = 0; i < LPT; i++) nxt[(rg + RG * i) * 64 + lane] = p[i];
    }
    __syncthreads();
  }
  // add the k slices in a fixed order: slice 0 + slice 1 + ...
  double *red = lds + KS * 2 * BD_KT * 64;
  if (kh > 0) {
    red[((kh - 1) * RG * BD_R + rg * BD_R + 0) * 64 + lane] = acc[0];
    red[((kh - 1) * RG * BD_R + rg * BD_R + 1) * 64 + lane] = acc[1];
  }
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int s = 1; s < KS; s++) {
      acc[0] += red[((s - 1) * RG * BD_R + rg * BD_R + 0) * 64 + lane];
      acc[1] += red[((s - 1) * RG * BD_R + rg * BD_R + 1) * 64 + lane];
    }
  }
}

template <int RG, int KS>
struct BdCfg {
  static constexpr int ROWS = RG * BD_R, THREADS = RG * KS * 64;
  static constexpr int LDS = KS * 2 * BD_KT * 64 + (KS - 1) * RG * BD_R * 64;
};

// ut = D22^-1 ( rx + [ -G | strict_lower(Linv) ] [wh ; rx] ), rows of L^-1
template <int RG, int KS>
__global__ __launch_bounds__(RG *KS * 64) void kbd_fwd(Dev d) {
  if (d.ctrl->done) return;
  using C = BdCfg<RG, KS>;
  __shared__ double lds[C::LDS];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t Bs = (size_t)d.Bs;
  const int b = blockIdx.y * 64 + lane;
  const int row0 = blockIdx.x * C::ROWS;
  int kend = d.M + row0 + C::ROWS;
  if (kend > d.M + d.n) kend = d.M + d.n;
  double acc[BD_R];
  bd_dot<RG, KS>(d.f_rows, d.ldf, d.n, row0, 0, kend, d.b_wh + b, Bs, lds, acc);
  if (w >= RG) return;
#pragma unroll
  for (int r = 0; r < BD_R; r++) {
    const int row = row0 + w * BD_R + r;
    if (row < d.n) d.b_ut[row * Bs + b] = d.d2inv[row] * (d.b_rx[row * Bs + b] + acc[r]);
  }
}

// rows of L^-T: blocks [0, nbx) the x rows (strict upper Linv^T), the rest the constraint rows (-G)^T
template <int RG, int KS>
__global__ __launch_bounds__(RG *KS * 64) void kbd_bwd(Dev d) {
  if (d.ctrl->done) return;
  using C = BdCfg<RG, KS>;
  __shared__ double lds[C::LDS];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t Bs = (size_t)d.Bs;
  const int b = blockIdx.y * 64 + lane;
  const int nbx = (d.n + C::ROWS - 1) / C::ROWS;
  double acc[BD_R];
  if ((int)blockIdx.x < nbx) {
    const int row0 = blockIdx.x * C::ROWS;
    bd_dot<RG, KS>(d.LinvT, d.ld, d.n, row0, row0 + 1, d.n, d.b_ut + b, Bs, lds, acc);
    if (w >= RG) return;
    const bool frozen = d.c_done[b] != 0;
#pragma unroll
    for (int r = 0; r < BD_R; r++) {
      const int row = row0 + w * BD_R + r;
      if (row >= d.n) continue;
      const size_t o = row * Bs + b;
      const double xt = d.b_ut[o] + acc[r];
      d.b_xt[o] = xt;
      if (!frozen) {
        const double xp = d.b_x[o];
        const double xn = d.alpha * xt + (1.0 - d.alpha) * xp;
        d.b_x[o] = xn;
        d.b_dx[o] = xn - xp;
        d.b_rx[o] = d.sigma * xn - d.q[row];
      }
    }
    return;
  }
  const int row0 = (blockIdx.x - nbx) * C::ROWS;
  bd_dot<RG, KS>(d.f_GmT, d.ldn, d.M, row0, 0, d.n, d.b_ut + b, Bs, lds, acc);
  if (w >= RG) return;
  if (d.c_done[b]) return;
  const double rho = d.rho, rinv = d.rho_inv, alpha = d.alpha;
#pragma unroll
  for (int r = 0; r < BD_R; r++) {
    const int row = row0 + w * BD_R + r;
    if (row >= d.M) continue;
    const size_t o = row * Bs + b;
    const double zp = d.b_z[o], yp = d.b_y[o];
    const double nu = -rho * d.b_wh[o] + acc[r];
    const double zt = zp + rinv * (nu - yp);
    const double zr = alpha * zt + (1.0 - alpha) * zp;
    const double v = zr + rinv * yp;
    const double zn = fmin(fmax(v, d.b_l[o]), d.b_u[o]);
    const double dy = rho * (zr - zn);
    const double yn = yp + dy;
    d.b_z[o] = zn;
    d.b_y[o] = yn;
    d.b_dy[o] = dy;
    d.b_wh[o] = zn - rinv * yn;
  }
}

// ------------------------------------------------------------------------------------------
// batched mode, product-form factor, fp64 matrix cores.  With 256 right-hand sides the sweeps are
// dense fp64 GEMMs (500 x 1750 x 256 for config 2); v_mfma_f64_16x16x4_f64 has the vector-FMA peak
// but needs 1/8 of the operand traffic and no wave-uniform operand at all (the scalar cache is what
// limits the kbd_* kernels).  One workgroup = KS waves sharing ONE 16 x 32 output tile: wave w takes
// the 16-deep k chunks w, w+KS, ...; partial tiles are added in LDS in wave order (fixed order).
// Fragment layout (MI355X_MICROARCH / cdna_hip_programming sec. 3): A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], C/D row = (l>>4) + 4*reg, col = l&15.  Within a 16-deep chunk MFMA step s
// covers k = k0 + 4*(l>>4) + s, so each lane reads its four A values as one 32-byte segment.
// ------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));
#ifndef MIOSQP_BM_NT
#define MIOSQP_BM_NT 2
#endif
#ifndef MIOSQP_BM_KS
#define MIOSQP_BM_KS 8
#endif
constexpr int BM_KS = MIOSQP_BM_KS, BM_NT = MIOSQP_BM_NT, BM_COLS = 16 * BM_NT, BM_OUT = BM_NT * 256;

__device__ __forceinline__ void bm_tile(const double *__restrict__ A, int ld, int nrows, int row0, int kbeg,
                                        int kend, const double *__restrict__ V, size_t Bs, double4_t (&acc)[BM_NT],
                                        int abl = 0) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r = lane & 15, j = lane >> 4;
  int row = row0 + r;
  row = row < nrows ? row : nrows - 1;
  const double *__restrict__ ar = A + (size_t)row * ld + 4 * j;
  const double *__restrict__ vr = V + r;
#pragma unroll
  for (int t = 0; t < BM_NT; t++) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double4_t acc2[BM_NT];
#pragma unroll
  for (int t = 0; t < BM_NT; t++) acc2[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int k0 = kbeg & ~15;
  // software pipeline: operands of chunk c+1 are in flight while chunk c feeds the matrix core
  double a[4], b[BM_NT][4], an[4], bn[BM_NT][4];
  auto fetch = [&](int kc, double (&pa)[4], double (&pb)[BM_NT][4]) {
    if (abl == 1) {  // debug ablation: no operand traffic
#pragma unroll
      for (int s = 0; s < 4; s++) {
        pa[s] = 1.0 + kc;
#pragma unroll
        for (int t = 0; t < BM_NT; t++) pb[t][s] = 2.0 + s;
      }
      return;
    }
    const double2 a01 = *reinterpret_cast<const double2 *>(ar + kc);
    const double2 a23 = *reinterpret_cast<const double2 *>(ar + kc + 2);
    pa[0] = a01.x; pa[1] = a01.y; pa[2] = a23.x; pa[3] = a23.y;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int k = kc + 4 * j + s;
      const bool in = k < kend;
#pragma unroll
      for (int t = 0; t < BM_NT; t++) pb[t][s] = in ? vr[(size_t)k * Bs + 16 * t] : 0.0;
    }
  };
  int kc = k0 + 16 * w;
  if (kc < kend) fetch(kc, a, b);
  for (; kc < kend; kc += 16 * BM_KS) {
    const int kn = kc + 16 * BM_KS;
    if (kn < kend) fetch(kn, an, bn);
    if (abl == 2) {  // debug ablation: no matrix-core work
#pragma unroll
      for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int t = 0; t < BM_NT; t++) acc[t][0] += a[s] + b[t][s];
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; s += 2) {  // two accumulator sets: four independent matrix-core chains
#pragma unroll
        for (int t = 0; t < BM_NT; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[t][s], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < BM_NT; t++)
          acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s + 1], b[t][s + 1], acc2[t], 0, 0, 0);
      }
    }
    if (kn < kend) {
#pragma unroll
      for (int s = 0; s < 4; s++) {
        a[s] = an[s];
#pragma unroll
        for (int t = 0; t < BM_NT; t++) b[t][s] = bn[t][s];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < BM_NT; t++) acc[t] += acc2[t];
}

// adds the KS partial tiles in wave order; thread e < 512 ends up with output element e:
// tile t = e / 256, reg = (e % 256) / 64, lane' = e % 64 -> row (lane'>>4) + 4*reg, col 16 t + (lane'&15)
__device__ __forceinline__ double bm_reduce(const double4_t (&acc)[BM_NT], double *lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < BM_NT; t++) {
#pragma unroll
    for (int g = 0; g < 4; g++) lds[((w * BM_NT + t) * 4 + g) * 64 + lane] = acc[t][g];
  }
  __syncthreads();
  const int e = threadIdx.x;  // threads e < BM_OUT own one output element each
  if (e >= BM_OUT) return 0.0;
  double s = lds[e];
#pragma unroll
  for (int ww = 1; ww < BM_KS; ww++) s += lds[ww * BM_OUT + e];
  return s;
}

__global__ __launch_bounds__(BM_KS * 64) void kbm_fwd(Dev d) {
  if (d.ctrl->done) return;
  __shared__ double lds[BM_KS * BM_NT * 256];
  const size_t Bs = (size_t)d.Bs;
  const int row0 = blockIdx.x * 16, col0 = blockIdx.y * BM_COLS;
  int kend = d.M + row0 + 16;
  if (kend > d.M + d.n) kend = d.M + d.n;
  double4_t acc[BM_NT];
  bm_tile(d.f_rows, d.ldf, d.n, row0, 0, kend, d.b_wh + col0, Bs, acc, d.bm_ablate);
  const double s = bm_reduce(acc, lds);
  const int e = threadIdx.x, t = e >> 8, g = (e & 255) >> 6, ll = e & 63;
  const int row = row0 + (ll >> 4) + 4 * g, b = col0 + 16 * t + (ll & 15);
  if (e < BM_OUT && row < d.n) d.b_ut[row * Bs + b] = d.d2inv[row] * (d.b_rx[row * Bs + b] + s);
}

__global__ __launch_bounds__(BM_KS * 64) void kbm_bwd(Dev d) {
  if (d.ctrl->done) return;
  __shared__ double lds[BM_KS * BM_NT * 256];
  const size_t Bs = (size_t)d.Bs;
  const int col0 = blockIdx.y * BM_COLS;
  const int nbx = (d.n + 15) / 16;
  const int e = threadIdx.x, t = e >> 8, g = (e & 255) >> 6, ll = e & 63;
  const int b = col0 + 16 * t + (ll & 15);
  double4_t acc[BM_NT];
  if ((int)blockIdx.x < nbx) {
    const int row0 = blockIdx.x * 16;
    bm_tile(d.LinvT, d.ld, d.n, row0, row0 + 1, d.n, d.b_ut + col0, Bs, acc);
    const double s = bm_reduce(acc, lds);
    const int row = row0 + (ll >> 4) + 4 * g;
    if (e >= BM_OUT || row >= d.n) return;
    const size_t o = row * Bs + b;
    const double xt = d.b_ut[o] + s;
    d.b_xt[o] = xt;
    if (!d.c_done[b]) {
      const double xp = d.b_x[o];
      const double xn = d.alpha * xt + (1.0 - d.alpha) * xp;
      d.b_x[o] = xn;
      d.b_dx[o] = xn - xp;
      d.b_rx[o] = d.sigma * xn - d.q[row];
    }
    return;
  }
  const int row0 = (blockIdx.x - nbx) * 16;
  bm_tile(d.f_GmT, d.ldn, d.M, row0, 0, d.n, d.b_ut + col0, Bs, acc);
  const double s = bm_reduce(acc, lds);
  const int row = row0 + (ll >> 4) + 4 * g;
  if (e >= BM_OUT || row >= d.M || d.c_done[b]) return;
  const size_t o = row * Bs + b;
  const double rho = d.rho, rinv = d.rho_inv, alpha = d.alpha;
  const double zp = d.b_z[o], yp = d.b_y[o];
  const double nu = -rho * d.b_wh[o] + s;
  const double zt = zp + rinv * (nu - yp);
  const double zr = alpha * zt + (1.0 - alpha) * zp;
  const double v = zr + rinv * yp;
  const double zn = fmin(fmax(v, d.b_l[o]), d.b_u[o]);
  const double dy = rho * (zr - zn);
  const double yn = yp + dy;
  d.b_z[o] = zn;
  d.b_y[o] = yn;
  d.b_dy[o] = dy;
  d.b_wh[o] = zn - rinv * yn;
}

// column-wise reductions: 1024 threads = 64 columns x 16 row groups; fixed order
#define COLRED(name, OP, init)                                                          \
  __device__ __forceinline__ double name(double v, double *lds, int bl, int rg) {       \
    __syncthreads();                                                                    \
    lds[rg * 64 + bl] = v;                                                              \
    __syncthreads();                                                                    \
    double r = lds[bl];                                                                 \
    for (int w = 1; w < 16; w++) r = OP(r, lds[w * 64 + bl]);                           \
    return r;                                                                           \
  }
__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
COLRED(colred_max, fmax, 0)
COLRED(colred_sum, op_add, 0)

// batched termination test on the dense copies, same matrix-core tiles as the sweeps
__global__ __launch_bounds__(BM_KS * 64) void kbm_check_con(Dev d) {
  if (d.ctrl->done) return;
  __shared__ double lds[BM_KS * BM_NT * 256];
  const size_t Bs = (size_t)d.Bs, MB = (size_t)d.M * Bs;
  const int row0 = blockIdx.x * 16, col0 = blockIdx.y * BM_COLS;
  const int e = threadIdx.x, t = e >> 8, g = (e & 255) >> 6, ll = e & 63;
  const int row = row0 + (ll >> 4) + 4 * g, b = col0 + 16 * t + (ll & 15);
  double4_t acc[BM_NT];
  bm_tile(d.f_Ad, d.ldn, d.M, row0, 0, d.n, d.b_x + col0, Bs, acc);
  const double ax = bm_reduce(acc, lds);
  __syncthreads();
  bm_tile(d.f_Ad, d.ldn, d.M, row0, 0, d.n, d.b_dx + col0, Bs, acc);
  const double adx0 = bm_reduce(acc, lds);
  if (e >= BM_OUT || row >= d.M) return;
  const size_t o = row * Bs + b;
  const double ei = d.Einv[row], z = d.b_z[o], l = d.b_l[o], u = d.b_u[o];
  const bool uinf = u > QP_INFTY * QP_MIN_SCALING, linf = l < -QP_INFTY * QP_MIN_SCALING;
  double v = d.b_dy[o];
  if (uinf && linf) v = 0.0;
  else if (uinf) v = fmin(v, 0.0);
  else if (linf) v = fmax(v, 0.0);
  const double adx = ei * adx0;
  d.b_sm[0 * MB + o] = ei * (ax - z);
  d.b_sm[1 * MB + o] = ei * ax;
  d.b_sm[2 * MB + o] = ei * z;
  d.b_sm[3 * MB + o] = v;
  d.b_sm[4 * MB + o] = d.E[row] * v;
  d.b_sm[5 * MB + o] = u * fmax(v, 0.0) + l * fmin(v, 0.0);
  d.b_sm[6 * MB + o] = uinf ? -1.7e308 : adx;
  d.b_sm[7 * MB + o] = linf ? 1.7e308 : adx;
}

// blocks [0, nbx): rows of Pbar against x, dx; blocks [nbx, 2 nbx): rows of Abar^T against y, projected dy
__global__ __launch_bounds__(BM_KS * 64) void kbm_check_var(Dev d) {
  if (d.ctrl->done) return;
  __shared__ double lds[BM_KS * BM_NT * 256];
  const size_t Bs = (size_t)d.Bs, NB = (size_t)d.n * Bs;
  const int nbx = (d.n + 15) / 16;
  const bool second = (int)blockIdx.x >= nbx;
  const int row0 = ((int)blockIdx.x - (second ? nbx : 0)) * 16, col0 = blockIdx.y * BM_COLS;
  const int e = threadIdx.x, t = e >> 8, g = (e & 255) >> 6, ll = e & 63;
  const int row = row0 + (ll >> 4) + 4 * g, b = col0 + 16 * t + (ll & 15);
  double4_t acc[BM_NT];
  const double *A = second ? d.f_Atd : d.f_Pd;
  const int ld = second ? d.ldm : d.ldn, K = second ? d.M : d.n;
  const double *V0 = (second ? d.b_y : d.b_x) + col0;
  const double *V1 = (second ? d.b_sm + 3 * (size_t)d.M * Bs : d.b_dx) + col0;
  bm_tile(A, ld, d.n, row0, 0, K, V0, Bs, acc);
  const double r0 = bm_reduce(acc, lds);
  __syncthreads();
  bm_tile(A, ld, d.n, row0, 0, K, V1, Bs, acc);
  const double r1 = bm_reduce(acc, lds);
  if (e >= BM_OUT || row >= d.n) return;
  const size_t o = row * Bs + b;
  d.b_sn[(second ? 2 : 0) * NB + o] = r0;
  d.b_sn[(second ? 3 : 1) * NB + o] = d.Dinv[row] * r1;
}

// Stage 1 of the batched decision: grid (column tiles, KR row slices); every workgroup folds its
// slice of rows for 64 columns and 17 quantities into b_part[tile][slice][q][64].
constexpr int KR = 32;
__global__ __launch_bounds__(256) void kb_check_reduce(Dev d) {
  if (d.ctrl->done) return;
  __shared__ double part[NQ][4][64];
  const int n = d.n, M = d.M, tid = threadIdx.x, bl = tid & 63, rg = tid >> 6;
  const int b = blockIdx.x * 64 + bl, sl = blockIdx.y;
  const size_t Bs = (size_t)d.Bs, MB = (size_t)M * Bs, NB = (size_t)n * Bs;
  double v[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = 0.0;
  v[4] = -1.7e308;
  v[5] = -1.7e308;
  for (int j = sl * 4 + rg; j < M; j += 4 * KR) {
    const size_t o = j * Bs + b;
    v[0] = fmax(v[0], fabs(d.b_sm[0 * MB + o]));
    v[1] = fmax(v[1], fabs(d.b_sm[1 * MB + o]));
    v[2] = fmax(v[2], fabs(d.b_sm[2 * MB + o]));
    v[3] = fmax(v[3], fabs(d.b_sm[4 * MB + o]));
    v[4] = fmax(v[4], d.b_sm[6 * MB + o]);
    v[5] = fmax(v[5], -d.b_sm[7 * MB + o]);
    v[13] += d.b_sm[5 * MB + o];
  }
  for (int i = sl * 4 + rg; i < n; i += 4 * KR) {
    const size_t o = i * Bs + b;
    const double di = d.Dinv[i], px = d.b_sn[0 * NB + o], aty = d.b_sn[2 * NB + o], q = d.q[i], x = d.b_x[o],
                 dx = d.b_dx[o];
    v[6] = fmax(v[6], fabs(di * (px + q + aty)));
    v[7] = fmax(v[7], fabs(di * px));
    v[8] = fmax(v[8], fabs(di * aty));
    v[9] = fmax(v[9], fabs(di * q));
    v[10] = fmax(v[10], fabs(d.b_sn[1 * NB + o]));
    v[11] = fmax(v[11], fabs(d.b_sn[3 * NB + o]));
    v[12] = fmax(v[12], fabs(d.D[i] * dx));
    v[14] += q * dx;
    v[15] += x * px;
    v[16] += q * x;
  }
#pragma unroll
  for (int q = 0; q < NQ; q++) part[q][rg][bl] = v[q];
  __syncthreads();
  if (rg != 0) return;
  double *out = d.b_part + ((size_t)(blockIdx.x * KR + sl) * NQ) * 64;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    double r = part[q][0][bl];
#pragma unroll
    for (int w = 1; w < 4; w++) r = q < NQ_MAX ? fmax(r, part[q][w][bl]) : r + part[q][w][bl];
    out[q * 64 + bl] = r;
  }
}

// Stage 2: one wave per column tile combines the KR slices in slice order and decides
__global__ __launch_bounds__(64) void kb_check_decide(Dev d) {
  if (d.ctrl->done) return;
  const int bl = threadIdx.x, b = blockIdx.x * 64 + bl;
  double v[NQ];
  const double *in = d.b_part + ((size_t)blockIdx.x * KR * NQ) * 64;
#pragma unroll
  for (int q = 0; q < NQ; q++) v[q] = in[q * 64 + bl];
  for (int sl = 1; sl < KR; sl++) {
    const double *p = in + (size_t)sl * NQ * 64;
#pragma unroll
    for (int q = 0; q < NQ; q++) v[q] = q < NQ_MAX ? fmax(v[q], p[q * 64 + bl]) : v[q] + p[q * 64 + bl];
  }
  Norms nm{v[0], v[1], v[2], v[3], v[4], -v[5], v[6] * d.cinv, v[7], v[8], v[9], v[10], v[11], v[12], v[13], v[14],
           v[15], v[16]};
  Ctrl *c = d.ctrl;
  bool newly = false;
  if (!d.c_done[b]) {
    double obj;
    const int st = decide_status(d, nm, obj);
    d.c_pri[b] = nm.pri;
    d.c_dua[b] = nm.dua;
    d.c_obj[b] = obj;
    if (st) {
      d.c_status[b] = st;
      d.c_iter[b] = c->iter;  // kb_tick already counted this chunk
      d.c_done[b] = 1;
      newly = true;
    }
  }
  const int cnt = __popcll(__ballot(newly));
  if (bl == 0 && cnt > 0) {
    const int before = atomicAdd(&c->ndone, cnt);
    if (before + cnt >= c->B) c->done = 1;
  }
}

// first kernel of a batched chunk: counts the chunk's iterations unless everything is decided
__global__ void kb_tick(Dev d, int iters_in_chunk) {
  if (!d.ctrl->done) d.ctrl->iter += iters_in_chunk;
}

__global__ void kb_reset(Dev d, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < d.Bs) {
    d.c_done[b] = b >= B;
    d.c_node[b] = b;
    d.c_status[b] = MIOSQP_QP_UNSOLVED;
    d.c_iter[b] = 0;
    d.c_pri[b] = d.c_dua[b] = d.c_obj[b] = 0.0;
    d.c_lower[b] = __builtin_nan("");
  }
  if (b == 0) {
    Ctrl *c = d.ctrl;
    c->done = 0;
    c->status = MIOSQP_QP_UNSOLVED;
    c->iter = 0;
    c->pad = 0;
    c->B = B;
    c->ndone = 0;
  }
}

// node-major staging -> scaled, batch-fastest working vectors (block = 64 columns x 4 rows)
__global__ __launch_bounds__(256) void kb_prepare(Dev d, int B) {
  const int lane = threadIdx.x & 63, b = blockIdx.y * 64 + lane;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t Bs = (size_t)d.Bs, n = d.n, M = d.M;
  const bool act = b < B;
  const double *rl = d.b_raw, *ru = rl + (size_t)B * M, *rx = ru + (size_t)B * M, *ry = rx + (size_t)B * n;
  if (j < d.M) {
    const size_t o = j * Bs + b;
    d.b_l[o] = act ? d.E[j] * fmax(rl[(size_t)b * M + j], -QP_INFTY) : 0.0;
    d.b_u[o] = act ? d.E[j] * fmin(ru[(size_t)b * M + j], QP_INFTY) : 0.0;
    d.b_y[o] = act ? d.c * d.Einv[j] * ry[(size_t)b * M + j] : 0.0;
    d.b_dy[o] = 0.0;
  }
  if (j < d.n) {
    const size_t o = j * Bs + b;
    const double xs = act ? d.Dinv[j] * rx[(size_t)b * n + j] : 0.0;
    d.b_x[o] = xs;
    d.b_dx[o] = 0.0;
    d.b_rx[o] = d.sigma * xs - d.q[j];
  }
}

__global__ __launch_bounds__(256) void kb_warm_z(Dev d) {
  BSETUP
  const int row = blockIdx.x * 4 + wv;
  if (row >= d.M) return;
  const double z = brow_dot(d.pc_idx, d.pc_A, d.pc_ptr[row], d.pc_ptr[row + 1], d.b_x + b, Bs);
  const size_t o = row * Bs + b;
  d.b_z[o] = z;
  d.b_wh[o] = z - d.rho_inv * d.b_y[o];
}

// Compaction of a wave: decided columns are swapped towards the tail so that the columns still
// iterating fill the first tiles and later chunks launch fewer tiles.  grid (row chunks, pairs).
__global__ __launch_bounds__(256) void kb_swap_cols(Dev d, int npairs) {
  const int p = blockIdx.y;
  if (p >= npairs) return;
  const int a = d.c_pairs[2 * p], b = d.c_pairs[2 * p + 1];
  const size_t Bs = (size_t)d.Bs;
  const int r = blockIdx.x * 256 + threadIdx.x;
#define SWAPROW(arr, rows)                                  \
  if (r < (rows)) {                                         \
    const double t_ = arr[r * Bs + a];                      \
    arr[r * Bs + a] = arr[r * Bs + b];                      \
    arr[r * Bs + b] = t_;                                   \
  }
  SWAPROW(d.b_l, d.M) SWAPROW(d.b_u, d.M) SWAPROW(d.b_z, d.M) SWAPROW(d.b_y, d.M) SWAPROW(d.b_dy, d.M)
  SWAPROW(d.b_wh, d.M + d.n)  // wh | rx
  SWAPROW(d.b_x, d.n) SWAPROW(d.b_dx, d.n)
#undef SWAPROW
  if (r == 0) {
#define SWAP1(T, arr) { const T t_ = arr[a]; arr[a] = arr[b]; arr[b] = t_; }
    SWAP1(int, d.c_done) SWAP1(int, d.c_status) SWAP1(int, d.c_iter) SWAP1(int, d.c_node)
    SWAP1(double, d.c_pri) SWAP1(double, d.c_dua) SWAP1(double, d.c_obj)
#undef SWAP1
  }
}

// unscale (or build the certificate) per column, then the integer clamp of node.py:131-136
__global__ __launch_bounds__(1024) void kb_finish(Dev d, int B) {
  __shared__ double lds[16 * 64];
  const int n = d.n, M = d.M, tid = threadIdx.x, bl = tid & 63, rg = tid >> 6;
  const int b = blockIdx.x * 64 + bl;
  const size_t Bs = (size_t)d.Bs;
  int st = d.c_status[b];
  if (st == MIOSQP_QP_UNSOLVED) st = MIOSQP_QP_MAX_ITER_REACHED;
  double ndy = 0, ndx = 0;
  for (int j = rg; j < M; j += 16) ndy = fmax(ndy, fabs(d.E[j] * d.b_dy[j * Bs + b]));
  for (int i = rg; i < n; i += 16) ndx = fmax(ndx, fabs(d.D[i] * d.b_dx[i * Bs + b]));
  ndy = colred_max(ndy, lds, bl, rg);
  ndx = colred_max(ndx, lds, bl, rg);
  const double nan = __builtin_nan("");
  for (int i = rg; i < n; i += 16) {
    const size_t o = i * Bs + b;
    double v;
    if (st == MIOSQP_QP_PRIMAL_INFEASIBLE) v = nan;
    else if (st == MIOSQP_QP_DUAL_INFEASIBLE) v = d.D[i] * d.b_dx[o] / ndx;
    else v = d.D[i] * d.b_x[o];
    d.b_xfin[o] = v;
  }
  for (int j = rg; j < M; j += 16) {
    const size_t o = j * Bs + b;
    double v;
    if (st == MIOSQP_QP_PRIMAL_INFEASIBLE) v = d.E[j] * d.b_dy[o] / ndy;
    else if (st == MIOSQP_QP_DUAL_INFEASIBLE) v = nan;
    else v = d.cinv * d.E[j] * d.b_y[o];
    d.b_yfin[o] = v;
  }
  __syncthreads();
  if (b < B && (st == MIOSQP_QP_SOLVED || st == MIOSQP_QP_MAX_ITER_REACHED)) {
    const double *rl = d.b_raw, *ru = rl + (size_t)B * M;
    for (int k = rg; k < d.n_int; k += 16) {
      const size_t o = (size_t)d.i_idx[k] * Bs + b;
      const size_t nb_ = (size_t)d.c_node[b];  // the node this column holds after compaction swaps
      const double lo = rl[nb_ * M + d.m_orig + k], hi = ru[nb_ * M + d.m_orig + k];
      d.b_xfin[o] = fmin(fmax(d.b_xfin[o], lo), hi);
    }
  }
  if (d.digest) {
    // per column: is_int_feas + pick_nextvar + rounded candidate (workspace.py:245-272)
    const bool okc = b < B && (st == MIOSQP_QP_SOLVED || st == MIOSQP_QP_MAX_ITER_REACHED);
    __syncthreads();
    for (int i = rg; i < n; i += 16) d.b_xi[i * Bs + b] = d.b_xfin[i * Bs + b];
    __syncthreads();
    int cnt = 0, bestk = 0x7fffffff;
    double best = -1.0;
    for (int k = rg; k < d.n_int; k += 16) {
      const size_t o = (size_t)d.i_idx[k] * Bs + b;
      const double v = d.b_xfin[o], r = rint(v), f = fabs(v - r);
      d.b_xi[o] = r;
      cnt += f > d.eps_int;
      if (f > best) { best = f; bestk = k; }
    }
    __syncthreads();
    lds[rg * 64 + bl] = best;
    __shared__ int lk[16 * 64], lc[16 * 64];
    lk[rg * 64 + bl] = bestk;
    lc[rg * 64 + bl] = cnt;
    __syncthreads();
    if (rg == 0) {
      for (int w = 1; w < 16; w++) {
        const double ob = lds[w * 64 + bl];
        const int ok = lk[w * 64 + bl];
        cnt += lc[w * 64 + bl];
        if (ob > best || (ob == best && ok < bestk)) { best = ob; bestk = ok; }
      }
      d.c_intinf[b] = okc ? cnt : -1;
      d.c_nextvar[b] = okc && bestk != 0x7fffffff ? bestk : -1;
    }
    for (int i = rg; i < n; i += 16) d.b_xis[i * Bs + b] = d.Dinv[i] * d.b_xi[i * Bs + b];
  }
  if (rg == 0 && d.c_status[b] == MIOSQP_QP_UNSOLVED) {
    d.c_status[b] = MIOSQP_QP_MAX_ITER_REACHED;
    d.c_iter[b] = d.ctrl->iter;
  }
}

// rounding heuristic per column: rows of A against the rounded candidates (workspace.py:232-243)
__global__ __launch_bounds__(256) void kb_heur_rows(Dev d) {
  BSETUP
  const int row = blockIdx.x * 4 + wv;
  if (row >= d.M) return;
  const double acc = brow_dot(d.pc_idx, d.pc_A, d.pc_ptr[row], d.pc_ptr[row + 1], d.b_xis + b, Bs);
  const double z = d.Einv[row] * acc;
  d.b_sm[row * Bs + b] = fmax(d.root_l[row] - d.eps_lin - z, z - d.root_u[row] - d.eps_lin);
}

__global__ __launch_bounds__(256) void kb_obj_rows(Dev d) {
  BSETUP
  const int row = blockIdx.x * 4 + wv;
  if (row >= d.n) return;
  double acc, acc2;
  brow_dot2(d.pr_idx, d.pr_val, d.pr_ptr[row], d.pr_ptr[row + 1], d.b_xfin + b, (d.digest ? d.b_xi : d.b_xfin) + b, Bs,
            acc, acc2);
  d.b_sn[row * Bs + b] = d.b_xfin[row * Bs + b] * (0.5 * acc + d.qraw[row]);
  if (d.digest) d.b_sn[(size_t)d.n * Bs + row * Bs + b] = d.b_xi[row * Bs + b] * (0.5 * acc2 + d.qraw[row]);
}

__global__ __launch_bounds__(1024) void kb_obj_sum(Dev d) {
  __shared__ double lds[16 * 64];
  const int tid = threadIdx.x, bl = tid & 63, rg = tid >> 6, b = blockIdx.x * 64 + bl;
  const size_t Bs = (size_t)d.Bs;
  double s = 0, s2 = 0, vmax = -1.7e308;
  for (int i = rg; i < d.n; i += 16) s += d.b_sn[i * Bs + b];
  s = colred_sum(s, lds, bl, rg);
  if (d.digest) {
    for (int i = rg; i < d.n; i += 16) s2 += d.b_sn[(size_t)d.n * Bs + i * Bs + b];
    for (int j = rg; j < d.M; j += 16) vmax = fmax(vmax, d.b_sm[j * Bs + b]);
    s2 = colred_sum(s2, lds, bl, rg);
    vmax = colred_max(vmax, lds, bl, rg);
  }
  if (rg == 0) {
    const int st = d.c_status[b];
    const bool ok = st == MIOSQP_QP_SOLVED || st == MIOSQP_QP_MAX_ITER_REACHED;
    d.c_lower[b] = ok ? s : __builtin_nan("");
    d.c_hobj[b] = ok && d.digest ? s2 : __builtin_nan("");
    d.c_hviol[b] = ok && d.digest ? vmax : __builtin_nan("");
  }
}

// batch-fastest answers -> node-major staging out
__global__ __launch_bounds__(256) void kb_export(Dev d, int B) {
  const int lane = threadIdx.x & 63, b = blockIdx.y * 64 + lane;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const size_t Bs = (size_t)d.Bs, n = d.n, M = d.M;
  double *ox = d.b_out, *oy = ox + (size_t)B * n;
  if (j < d.n) ox[(size_t)b * n + j] = d.b_xfin[j * Bs + b];
  if (j < d.M) oy[(size_t)b * M + j] = d.b_yfin[j * Bs + b];
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

}  // namespace

struct miosqp_qp_engine {
  int n = 0, M = 0;
  miosqp_qp_settings st{};
  miosqp::Scaled sc;
  miosqp::Factor fa;
  Dev d{};
  std::vector<void *> allocs;  // pool chunks
  char *pool_base = nullptr;
  size_t pool_cap = 0, pool_used = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evc0 = nullptr, evc1 = nullptr;
  double loop_ms = 0.0;
  int64_t loop_iters = 0;
  hipGraph_t g_full = nullptr, g_tail = nullptr;
  hipGraphExec_t x_full = nullptr, x_tail = nullptr;
  int chunk = 25, tail_iters = 0;
  int tpr_pv = 64, tpr_pc = 64, tpr_tail = 64, tpr_pb = 64, tpr_pr = 64;
  int bd_cfg = 0;  // batched dense kernels: 0 = fp64 matrix-core tiles, else 10*row_groups + k_slices (vector FMA)
  int tpr_ff = 256, tpr_fx = 64, tpr_fc = 64;  // product-form kernels: forward rows, x rows, constraint rows
  // pinned staging: [l | u | x0 | y0] in, [x | y] out, ctrl
  double *h_in = nullptr, *h_out = nullptr;
  Ctrl *h_ctrl = nullptr;
  double *d_in = nullptr;
  bool have_int = false;
  bool fold = false;
  bool setup_on_device = false;
  bool res_pending = false, loop_pending = false;
  Ctrl *h_ctrl2 = nullptr;  // two pinned slots for the pipelined chunk loop
  hipEvent_t ev_chunk[2] = {nullptr, nullptr};
  bool resident = false;  // whole solve in one LDS-resident workgroup (small problems)
  bool coop = false;      // register-resident cooperative solver (k_coop), one exchange per iteration
  int coop_rw = 8, coop_cpt = 4, coop_T = 0;
  int res_tg1 = 64, res_tg2 = 64;
  size_t res_lds = 0;
  miosqp::Folded fo;
  int64_t nnzA = 0, nnzPtriu = 0;
  // batched mode
  int Bcap = 0;  // capacity (columns), multiple of 64; 0 = batched mode not allocated
  double *hb_in = nullptr, *hb_out = nullptr;
  int *hb_int = nullptr;       // status | iter
  double *hb_dbl = nullptr;    // pri | dua | obj | lower
  hipGraphExec_t xb_full[16] = {}, xb_tail[16] = {};
  hipGraph_t gb_full[16] = {}, gb_tail[16] = {};
  bool compact = true;   // compaction of converged columns in solve_batch (MIOSQP_COMPACT=0 disables)
  int64_t compactions = 0;
  double bloop_ms = 0.0;
  int64_t bloop_iters = 0, bloop_node_iters = 0;
};

namespace {

// Device memory comes from a few large zero-filled chunks (one hipMalloc per chunk instead of one
// per array: setup of a small problem is dominated by allocation calls otherwise).
int pool_reserve(miosqp_qp_engine *e, size_t bytes) {
  void *p = nullptr;
  bytes = (bytes + 4095) & ~(size_t)4095;
  HIPCHK(hipMalloc(&p, bytes));
  HIPCHK(hipMemset(p, 0, bytes));
  e->allocs.push_back(p);
  e->pool_base = (char *)p;
  e->pool_cap = bytes;
  e->pool_used = 0;
  return 0;
}
int pool_alloc(miosqp_qp_engine *e, void **out, size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  if (e->pool_used + bytes > e->pool_cap) {
    int rc = pool_reserve(e, bytes > ((size_t)8 << 20) ? bytes : ((size_t)8 << 20));
    if (rc) return rc;
  }
  *out = e->pool_base + e->pool_used;
  e->pool_used += bytes;
  return 0;
}
template <typename T>
int dalloc(miosqp_qp_engine *e, T **p, size_t count) {
  return pool_alloc(e, (void **)p, (count ? count : 1) * sizeof(T));
}
template <typename T>
int dupload(miosqp_qp_engine *e, const std::vector<T> &h, const T **p) {
  T *q = nullptr;
  int rc = pool_alloc(e, (void **)&q, (h.size() + 64) * sizeof(T));  // slack: tile loads may overshoot a row
  if (rc) return rc;
  if (h.size()) HIPCHK(hipMemcpy(q, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return 0;
}

void launch_fold_fwd(miosqp_qp_engine *e) {
  const Dev &d = e->d;
  DISPATCH_TPR(e->tpr_ff, k_fold_fwd, d.n, e->stream, d);
}

template <int TX>
void launch_fold_bwd_c(miosqp_qp_engine *e) {
  const Dev &d = e->d;
  const int nbx = (d.n + 256 / TX - 1) / (256 / TX);
#define FB(TC)                                                                                         \
  hipLaunchKernelGGL((k_fold_bwd<TX, TC>), dim3(nbx + (d.M + 256 / TC - 1) / (256 / TC)), dim3(256), 0, \
                     e->stream, d)
  switch (e->tpr_fc) {
    case 32: FB(32); break;
    case 128: FB(128); break;
    case 256: FB(256); break;
    default: FB(64); break;
  }
#undef FB
}

void launch_fold_bwd(miosqp_qp_engine *e) {
  switch (e->tpr_fx) {
    case 16: launch_fold_bwd_c<16>(e); break;
    case 32: launch_fold_bwd_c<32>(e); break;
    case 128: launch_fold_bwd_c<128>(e); break;
    default: launch_fold_bwd_c<64>(e); break;
  }
}

int launch_resident(miosqp_qp_engine *e, int max_iter, int check_every, int final_check) {
  hipLaunchKernelGGL(k_resident, dim3(1), dim3(RES_THREADS), e->res_lds, e->stream, e->d, max_iter, check_every,
                     final_check, e->res_tg1, e->res_tg2);
  return 0;
}

void launch_coop(miosqp_qp_engine *e, int max_iter, int check_every, int final_check) {
  const Dev &d = e->d;
  (void)hipMemsetAsync(d.coop_reg, 0, 64, e->stream);
#define CO(B, RW, CPT)                                                                                   \
  hipLaunchKernelGGL((k_coop<B, RW, CPT>), dim3(e->coop_T), dim3(B), 0, e->stream, d, max_iter, check_every, \
                     final_check)
  if (e->coop_cpt == 2) CO(512, 8, 2);
  else CO(512, 8, 4);
#undef CO
}

void launch_iteration(miosqp_qp_engine *e) {
  const Dev &d = e->d;
  if (e->coop) {
    launch_coop(e, 1, 0, 0);
    return;
  }
  if (e->fold) {
    launch_fold_fwd(e);
    launch_fold_bwd(e);
    return;
  }
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
  hipLaunchKernelGGL(k_check_decide, dim3(1), dim3(256), 0, e->stream, d, iters_in_chunk);
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
  if (e->resident) {
    HIPCHK(hipEventRecord(e->evc0, e->stream));
    int rc = launch_resident(e, e->st.max_iter, e->st.check_termination, 1);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e->evc1, e->stream));
    e->res_pending = true;
    return 0;
  }
  if (e->coop) {  // the cooperative solver runs the whole loop, tests included, in one launch
    HIPCHK(hipEventRecord(e->evc0, e->stream));
    launch_coop(e, e->st.max_iter, e->st.check_termination, 1);
    HIPCHK(hipEventRecord(e->evc1, e->stream));
    e->res_pending = true;
    return 0;
  }
  // One chunk is always queued AHEAD of the one whose verdict the host is waiting for, so the GPU
  // never idles across the host round trip; every kernel of a chunk exits at once when the
  // previous test already decided (ctrl->done).
  const int nfull = e->st.max_iter / e->chunk;
  const int total = nfull + (e->tail_iters > 0 ? 1 : 0);
  HIPCHK(hipEventRecord(e->evc0, e->stream));
  auto enqueue = [&](int k) -> int {
    HIPCHK(hipGraphLaunch(k < nfull ? e->x_full : e->x_tail, e->stream));
    HIPCHK(hipMemcpyAsync(&e->h_ctrl2[k & 1], e->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipEventRecord(e->ev_chunk[k & 1], e->stream));
    return 0;
  };
  int rc = enqueue(0);
  if (rc) return rc;
  for (int k = 0; k < total; k++) {
    if (k + 1 < total) {
      rc = enqueue(k + 1);
      if (rc) return rc;
    }
    HIPCHK(hipEventSynchronize(e->ev_chunk[k & 1]));
    if (e->h_ctrl2[k & 1].done) break;
  }
  HIPCHK(hipEventRecord(e->evc1, e->stream));
  e->loop_pending = true;
  return 0;
}

int finish_and_fetch(miosqp_qp_engine *e, int node, double *x_out, double *y_out, miosqp_qp_info *info,
                     double t0) {
  const Dev &d = e->d;
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, e->stream, d, node, e->st.max_iter);
  if (node) {
    if (d.digest) DISPATCH_TPR(e->tpr_pc, k_heur_rows, d.M, e->stream, d);
    DISPATCH_TPR(e->tpr_pr, k_obj_rows, d.n, e->stream, d);
    hipLaunchKernelGGL(k_obj_sum, dim3(1), dim3(1024), 0, e->stream, d);
  }
  HIPCHK(hipMemcpyAsync(e->h_out, d.out_x, sizeof(double) * (e->n + e->M), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->h_ctrl, d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->h_ctrl->pad) {
    g_err = "cooperative solver: exchange timed out (workgroups not co-resident?), stage " + std::to_string(e->h_ctrl->pad);
    return MIOSQP_EHIP;
  }
  memcpy(x_out, e->h_out, sizeof(double) * e->n);
  memcpy(y_out, e->h_out + e->n, sizeof(double) * e->M);
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  if (e->res_pending || e->loop_pending) {
    float lms = 0;
    HIPCHK(hipEventElapsedTime(&lms, e->evc0, e->evc1));
    e->loop_ms += lms;
    e->loop_iters += e->h_ctrl->iter;
    e->res_pending = e->loop_pending = false;
  }
  info->status_val = e->h_ctrl->status;
  info->iter = e->h_ctrl->iter;
  info->obj_val = e->h_ctrl->obj_val;
  info->pri_res = e->h_ctrl->pri_res;
  info->dua_res = e->h_ctrl->dua_res;
  info->lower = e->h_ctrl->lower;
  info->int_inf = node ? e->h_ctrl->int_inf : -1;
  info->nextvar = node ? e->h_ctrl->nextvar : -1;
  info->heur_viol = e->h_ctrl->heur_viol;
  info->heur_obj = e->h_ctrl->heur_obj;
  info->device_time = 1e-3 * ms;
  info->run_time = wall() - t0;
  return 0;
}


// ---- batched mode (host) -------------------------------------------------------------------
// which: 0 forward sweep, 1 backward sweep; workgroup shape from e->bd_cfg (row groups x k slices)
void launch_bd(miosqp_qp_engine *e, int ntiles, int which) {
  const Dev &d = e->d;
  if (e->bd_cfg == 0) {  // fp64 matrix-core tiles: 16 rows x 32 columns per workgroup
    const int nbx = (d.n + 15) / 16, nbc = (d.M + 15) / 16, ncol = ntiles * (64 / BM_COLS);
    if (which == 0) hipLaunchKernelGGL(kbm_fwd, dim3(nbx, ncol), dim3(BM_KS * 64), 0, e->stream, d);
    else hipLaunchKernelGGL(kbm_bwd, dim3(nbx + nbc, ncol), dim3(BM_KS * 64), 0, e->stream, d);
    return;
  }
#define BD(RG, KS)                                                                                         \
  do {                                                                                                     \
    constexpr int R = BdCfg<RG, KS>::ROWS, T = BdCfg<RG, KS>::THREADS;                                      \
    const int nbx = (d.n + R - 1) / R, nbc = (d.M + R - 1) / R;                                            \
    if (which == 0) hipLaunchKernelGGL((kbd_fwd<RG, KS>), dim3(nbx, ntiles), dim3(T), 0, e->stream, d);      \
    else hipLaunchKernelGGL((kbd_bwd<RG, KS>), dim3(nbx + nbc, ntiles), dim3(T), 0, e->stream, d);           \
  } while (0)
  switch (e->bd_cfg) {
    case 24: BD(2, 4); break;
    case 44: BD(4, 4); break;
    case 22: BD(2, 2); break;
    case 28: BD(2, 8); break;
    default: BD(4, 2); break;
  }
#undef BD
}

void launch_iteration_b(miosqp_qp_engine *e, int ntiles) {
  const Dev &d = e->d;
  if (e->fold) {
    launch_bd(e, ntiles, 0);
    launch_bd(e, ntiles, 1);
    return;
  }
  hipLaunchKernelGGL(kb_panel_fwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d);
  hipLaunchKernelGGL(kb_tail_fwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d);
  hipLaunchKernelGGL(kb_tail_bwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d);
  hipLaunchKernelGGL(kb_panel_bwd, dim3((d.M + 3) / 4, ntiles), dim3(256), 0, e->stream, d);
}

int capture_chunk_b(miosqp_qp_engine *e, int iters, int ntiles, hipGraph_t *g, hipGraphExec_t *x) {
  const Dev &d = e->d;
  HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(kb_tick, dim3(1), dim3(1), 0, e->stream, d, iters);
  for (int i = 0; i < iters; i++) launch_iteration_b(e, ntiles);
  if (e->fold && e->bd_cfg == 0) {
    const int ncol = ntiles * (64 / BM_COLS);
    hipLaunchKernelGGL(kbm_check_con, dim3((d.M + 15) / 16, ncol), dim3(BM_KS * 64), 0, e->stream, d);
    hipLaunchKernelGGL(kbm_check_var, dim3(2 * ((d.n + 15) / 16), ncol), dim3(BM_KS * 64), 0, e->stream, d);
  } else {
    hipLaunchKernelGGL(kb_check_con, dim3((d.M + 3) / 4, ntiles), dim3(256), 0, e->stream, d);
    hipLaunchKernelGGL(kb_check_var, dim3(2 * ((d.n + 3) / 4), ntiles), dim3(256), 0, e->stream, d);
  }
  hipLaunchKernelGGL(kb_check_reduce, dim3(ntiles, KR), dim3(256), 0, e->stream, d);
  hipLaunchKernelGGL(kb_check_decide, dim3(ntiles), dim3(64), 0, e->stream, d);
  HIPCHK(hipStreamEndCapture(e->stream, g));
  HIPCHK(hipGraphInstantiate(x, *g, nullptr, nullptr, 0));
  return 0;
}

int alloc_batch(miosqp_qp_engine *e, int cap) {
  Dev &d = e->d;
  const size_t n = e->n, M = e->M;
  const size_t Bs = ((size_t)cap + 63) & ~(size_t)63;
  d.Bs = (int)Bs;
  {
    int rc0 = pool_reserve(e, Bs * ((size_t)24 * M + (size_t)16 * n + 64) * 8 + ((size_t)1 << 16));
    if (rc0) return rc0;
  }
#define ALB(field, count)                            \
  do {                                               \
    int rc__ = dalloc(e, &d.field, (size_t)(count)); \
    if (rc__) return rc__;                           \
  } while (0)
  ALB(b_l, M * Bs); ALB(b_u, M * Bs); ALB(b_x, n * Bs); ALB(b_z, M * Bs); ALB(b_y, M * Bs); ALB(b_wh, (M + n) * Bs);
  ALB(b_cv, n * Bs); ALB(b_ut, n * Bs); ALB(b_xt, n * Bs); ALB(b_dx, n * Bs); ALB(b_dy, M * Bs);
  ALB(b_sm, 8 * M * Bs); ALB(b_sn, 4 * n * Bs); ALB(b_xfin, n * Bs); ALB(b_yfin, M * Bs); ALB(b_xi, n * Bs); ALB(b_xis, n * Bs);
  ALB(c_intinf, Bs); ALB(c_nextvar, Bs); ALB(c_hviol, Bs); ALB(c_hobj, Bs); ALB(c_node, Bs); ALB(c_pairs, 2 * Bs);
  ALB(b_part, (Bs / 64) * 32 * 17 * 64);
  ALB(b_raw, Bs * (3 * M + n)); ALB(b_out, Bs * (n + M));
  ALB(c_done, Bs); ALB(c_status, Bs); ALB(c_iter, Bs); ALB(c_pri, Bs); ALB(c_dua, Bs); ALB(c_obj, Bs);
  ALB(c_lower, Bs);
#undef ALB
  d.b_rx = d.b_wh + M * Bs;
  HIPCHK(hipHostMalloc((void **)&e->hb_in, sizeof(double) * Bs * (3 * M + n), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&e->hb_out, sizeof(double) * Bs * (n + M), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&e->hb_int, sizeof(int) * 6 * Bs, hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void **)&e->hb_dbl, sizeof(double) * 6 * Bs, hipHostMallocDefault));
  e->Bcap = (int)Bs;
  return 0;
}

// one slice of at most Bcap nodes
int solve_slice(miosqp_qp_engine *e, int B, const double *l, const double *u, const double *x0, const double *y0,
                double *x_out, double *y_out, miosqp_qp_info *info) {
  const Dev &d = e->d;
  const size_t n = e->n, M = e->M;
  const double t0 = wall();
  const int ntiles = (B + 63) / 64;
  if (ntiles > 16) { g_err = "solve_batch: more than 1024 columns per slice"; return MIOSQP_EARG; }
  if (!e->xb_full[ntiles - 1]) {
    int rc = capture_chunk_b(e, e->chunk, ntiles, &e->gb_full[ntiles - 1], &e->xb_full[ntiles - 1]);
    if (!rc && e->tail_iters > 0)
      rc = capture_chunk_b(e, e->tail_iters, ntiles, &e->gb_tail[ntiles - 1], &e->xb_tail[ntiles - 1]);
    if (rc) return rc;
  }
  double *h = e->hb_in;
  memcpy(h, l, sizeof(double) * B * M);
  memcpy(h + B * M, u, sizeof(double) * B * M);
  memcpy(h + 2 * B * M, x0, sizeof(double) * B * n);
  memcpy(h + 2 * B * M + B * n, y0, sizeof(double) * B * M);
  HIPCHK(hipEventRecord(e->ev0, e->stream));
  HIPCHK(hipMemcpyAsync(d.b_raw, h, sizeof(double) * B * (3 * M + n), hipMemcpyHostToDevice, e->stream));
  hipLaunchKernelGGL(kb_reset, dim3((d.Bs + 255) / 256), dim3(256), 0, e->stream, d, B);
  const int big = (int)(n > M ? n : M);
  hipLaunchKernelGGL(kb_prepare, dim3((big + 3) / 4, ntiles), dim3(256), 0, e->stream, d, B);
  hipLaunchKernelGGL(kb_warm_z, dim3((d.M + 3) / 4, ntiles), dim3(256), 0, e->stream, d);
  const int nfull = e->st.max_iter / e->chunk;
  bool done = false;
  int decided = 0;
  int cur = ntiles;  // tiles still launched; shrinks as the wave is compacted
  for (int k = 0; k < nfull && !done; k++) {
    HIPCHK(hipEventRecord(e->evc0, e->stream));
    HIPCHK(hipGraphLaunch(e->xb_full[cur - 1], e->stream));
    HIPCHK(hipEventRecord(e->evc1, e->stream));
    HIPCHK(hipMemcpyAsync(e->h_ctrl, d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e->evc0, e->evc1));
    e->bloop_ms += ms;
    e->bloop_iters += e->chunk;
    e->bloop_node_iters += (int64_t)e->chunk * (B - decided);  // columns still iterating in this chunk
    decided = e->h_ctrl->ndone;
    done = e->h_ctrl->done != 0;
    const int active = B - decided, want = (active + 63) / 64;
    if (!done && e->compact && want < cur) {
      // move the still-iterating columns into the first `want` tiles
      const int cols = cur * 64;
      int *flags = e->hb_int;  // scratch: the info arrays are only filled at the end
      HIPCHK(hipMemcpyAsync(flags, d.c_done, sizeof(int) * cols, hipMemcpyDeviceToHost, e->stream));
      HIPCHK(hipStreamSynchronize(e->stream));
      int *pairs = e->hb_int + cols;
      int np = 0, lo = 0, hi = cols - 1;
      for (;;) {
        while (lo < active && !flags[lo]) lo++;      // a decided column inside the front region
        while (hi >= active && flags[hi]) hi--;      // a live column behind it
        if (lo >= active || hi < active) break;
        pairs[2 * np] = lo;
        pairs[2 * np + 1] = hi;
        np++;
        flags[lo] = 0;
        flags[hi] = 1;
      }
      if (np > 0) {
        HIPCHK(hipMemcpyAsync(d.c_pairs, pairs, sizeof(int) * 2 * np, hipMemcpyHostToDevice, e->stream));
        const int rows = (int)(M + n);
        hipLaunchKernelGGL(kb_swap_cols, dim3((rows + 255) / 256, np), dim3(256), 0, e->stream, d, np);
      }
      cur = want;
      if (!e->xb_full[cur - 1]) {
        int rc = capture_chunk_b(e, e->chunk, cur, &e->gb_full[cur - 1], &e->xb_full[cur - 1]);
        if (!rc && e->tail_iters > 0)
          rc = capture_chunk_b(e, e->tail_iters, cur, &e->gb_tail[cur - 1], &e->xb_tail[cur - 1]);
        if (rc) return rc;
      }
      e->compactions++;
    }
  }
  if (!done && e->tail_iters > 0) HIPCHK(hipGraphLaunch(e->xb_tail[cur - 1], e->stream));
  hipLaunchKernelGGL(kb_finish, dim3(ntiles), dim3(1024), 0, e->stream, d, B);
  if (d.digest) hipLaunchKernelGGL(kb_heur_rows, dim3((d.M + 3) / 4, ntiles), dim3(256), 0, e->stream, d);
  hipLaunchKernelGGL(kb_obj_rows, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d);
  hipLaunchKernelGGL(kb_obj_sum, dim3(ntiles), dim3(1024), 0, e->stream, d);
  hipLaunchKernelGGL(kb_export, dim3((big + 3) / 4, ntiles), dim3(256), 0, e->stream, d, B);
  HIPCHK(hipMemcpyAsync(e->hb_out, d.b_out, sizeof(double) * B * (n + M), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_int, d.c_status, sizeof(int) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_int + B, d.c_iter, sizeof(int) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_dbl, d.c_pri, sizeof(double) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_dbl + B, d.c_dua, sizeof(double) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_dbl + 2 * B, d.c_obj, sizeof(double) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_dbl + 3 * B, d.c_lower, sizeof(double) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_dbl + 4 * B, d.c_hviol, sizeof(double) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_dbl + 5 * B, d.c_hobj, sizeof(double) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_int + 2 * B, d.c_intinf, sizeof(int) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_int + 3 * B, d.c_nextvar, sizeof(int) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipMemcpyAsync(e->hb_int + 4 * B, d.c_node, sizeof(int) * B, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipEventRecord(e->ev1, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  const int *node_of = e->hb_int + 4 * B;  // column position -> node
  for (int c = 0; c < B; c++) {
    memcpy(x_out + (size_t)node_of[c] * n, e->hb_out + (size_t)c * n, sizeof(double) * n);
    memcpy(y_out + (size_t)node_of[c] * M, e->hb_out + B * n + (size_t)c * M, sizeof(double) * M);
  }
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  const double wall_s = wall() - t0;
  for (int c = 0; c < B; c++) {
    const int b = node_of[c];
    info[b].status_val = e->hb_int[c];
    info[b].iter = e->hb_int[B + c];
    info[b].pri_res = e->hb_dbl[c];
    info[b].dua_res = e->hb_dbl[B + c];
    info[b].obj_val = e->hb_dbl[2 * B + c];
    info[b].lower = e->hb_dbl[3 * B + c];
    info[b].int_inf = d.digest ? e->hb_int[2 * B + c] : -1;
    info[b].nextvar = d.digest ? e->hb_int[3 * B + c] : -1;
    info[b].heur_viol = e->hb_dbl[4 * B + c];
    info[b].heur_obj = e->hb_dbl[5 * B + c];
    info[b].run_time = wall_s / B;  // the wave's wall time, shared equally
    info[b].device_time = 1e-3 * ms / B;
  }
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
  s->fold = -1;
  s->resident = -1;
  s->setup_on_device = -1;
  s->coop = -1;
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
  if (e->h_ctrl2) hipHostFree(e->h_ctrl2);
  if (e->ev_chunk[0]) hipEventDestroy(e->ev_chunk[0]);
  if (e->ev_chunk[1]) hipEventDestroy(e->ev_chunk[1]);
  if (e->hb_in) hipHostFree(e->hb_in);
  if (e->hb_out) hipHostFree(e->hb_out);
  if (e->hb_int) hipHostFree(e->hb_int);
  if (e->hb_dbl) hipHostFree(e->hb_dbl);
  for (int k = 0; k < 16; k++) {
    if (e->xb_full[k]) hipGraphExecDestroy(e->xb_full[k]);
    if (e->xb_tail[k]) hipGraphExecDestroy(e->xb_tail[k]);
    if (e->gb_full[k]) hipGraphDestroy(e->gb_full[k]);
    if (e->gb_tail[k]) hipGraphDestroy(e->gb_tail[k]);
  }
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
  if (M == 0) {
    // miOSQP always has rows: data.py:5-33 appends one identity row per integer variable
    g_err = "setup: the engine needs at least one constraint row (M >= 1)";
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
  // dense part of the factorisation on the device for large problems
  int on_dev = s->setup_on_device;
  if (on_dev < 0) on_dev = n >= 1024 ? 1 : 0;
  e->setup_on_device = on_dev != 0;
  if (!miosqp::build_factor(e->sc, Pp, Pi, Px, s->rho, s->sigma, e->fa, err,
                            on_dev ? miosqp_device_ldl_inverse : nullptr, nullptr)) {
    g_err = err;
    delete e;
    return MIOSQP_EFACTOR;
  }
  e->nnzA = Ap[n];
  e->nnzPtriu = (int64_t)e->sc.Pi.size();
  const miosqp::Factor &f = e->fa;
  Dev &d = e->d;
  d.n = n; d.M = M; d.ld = f.ld; d.n_int = 0; d.m_orig = M;
  d.rho = s->rho; d.rho_inv = 1.0 / s->rho; d.sigma = s->sigma; d.alpha = s->alpha; d.eps_abs = s->eps_abs; d.eps_rel = s->eps_rel;
  d.eps_pinf = s->eps_prim_inf; d.eps_dinf = s->eps_dual_inf; d.c = e->sc.c; d.cinv = e->sc.cinv;
  {
    // one chunk for everything this setup will place on the device
    const size_t np2 = f.panel_by_var.idx.size() + f.panel_by_con.idx.size();
    size_t est = np2 * (4 + 16) + 2 * f.Linv.size() * 8 + (f.Pbar.idx.size() + f.Praw.idx.size()) * 12 +
                 ((size_t)60 * n + (size_t)60 * M) * 8 + ((size_t)n * ((M + n + 8)) + (size_t)M * (n + 8)) * 8 +
                 200 * 512 + ((size_t)1 << 16);
    int rc0 = pool_reserve(e, est);
    if (rc0) { miosqp_qp_cleanup(e); return rc0; }
  }
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
  AL(q, n); AL(qraw, n); AL(l, M); AL(u, M); AL(x, n); AL(z, M); AL(y, M); AL(wh, (size_t)M + n); AL(cv, n); AL(ut, n);
  AL(xt, n); AL(dx, n); AL(dy, M); AL(sm, 8 * (size_t)M); AL(sn, 10 * (size_t)n);
  d.rx = d.wh + M;
  AL(xi, n); AL(xis, n); AL(root_l, M); AL(root_u, M);
  d.digest = 0;
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
  HIPCHK(hipHostMalloc((void **)&e->h_ctrl2, 2 * sizeof(Ctrl), hipHostMallocDefault));
  HIPCHK(hipEventCreateWithFlags(&e->ev_chunk[0], hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&e->ev_chunk[1], hipEventDisableTiming));
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
  {
    // product-form factor: worth it when the panel is dense (bytes no worse, half the launches)
    const double dens = M > 0 ? (double)f.nnz_panel / ((double)n * M) : 0.0;
    const double fold_bytes = 8.0 * ((double)n * (M + n) + (double)M * n);
    int want = s->fold;
    const bool fits_lds = resident_lds_doubles(n, M) * sizeof(double) <= 150 * 1024;
    // the cooperative solver only needs the product form to build its explicit inverse from, whatever
    // the sparsity; it is preferred from n + M = 64 on (measured: equal to the LDS-resident workgroup
    // below 100, 1.8x at 160, 2.2x at 240)
    int coop_req = s->coop;
    if (const char *ev = getenv("MIOSQP_COOP")) coop_req = atoi(ev);
    const bool coop_fits = M > 0 && n + M <= 2048;
    const bool coop_pref = coop_fits && (coop_req == 1 || (coop_req < 0 && n + M >= 64)) && s->resident != 1;
    if (want < 0)
      want = (M > 0 && ((dens >= 0.30 && fold_bytes <= 4.0e9) || (fits_lds && s->resident != 0) || coop_pref)) ? 1 : 0;
    if (want && M > 0) {
      miosqp::build_folded(f, e->fo);
      int rc = dupload(e, e->fo.rows, &d.f_rows);
      if (!rc) rc = dupload(e, e->fo.GmT, &d.f_GmT);
      if (!rc) rc = dupload(e, e->fo.Ad, &d.f_Ad);
      if (!rc) rc = dupload(e, e->fo.Atd, &d.f_Atd);
      if (!rc) rc = dupload(e, e->fo.Pd, &d.f_Pd);
      d.ldm = e->fo.ldm;
      if (rc) { miosqp_qp_cleanup(e); return rc; }
      d.ldf = e->fo.ldf;
      d.ldn = e->fo.ldn;
      e->fold = true;
      e->tpr_ff = pick_tpr(M + 0.5 * n);
      e->tpr_fx = pick_tpr(0.5 * n);
      e->tpr_fc = pick_tpr((double)n);
      {
        const size_t need = resident_lds_doubles(n, M) * sizeof(double);
        int wantr = s->resident;
        if (wantr < 0) wantr = (need <= 150 * 1024 && !coop_pref) ? 1 : 0;
        if (wantr && need <= 160 * 1024) {
          e->resident = true;
          e->res_lds = need;
          HIPCHK(hipFuncSetAttribute((const void *)k_resident, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)need));
          // lanes per row: as many as keep every row of a sweep in flight at once
          auto pow2_floor = [](int v) { int p = 1; while (2 * p <= v) p *= 2; return p; };
          e->res_tg1 = std::min(64, std::max(1, pow2_floor(RES_THREADS / n)));
          e->res_tg2 = std::min(64, std::max(1, pow2_floor(RES_THREADS / (n + M))));
          if (const char *ev = getenv("MIOSQP_RES_TG")) {
            int a1 = 0, b1 = 0;
            if (sscanf(ev, "%d,%d", &a1, &b1) == 2) { e->res_tg1 = a1; e->res_tg2 = b1; }
          }
        }
      }
      {
        // cooperative register-resident solver: the explicit KKT inverse spread over the register
        // files of up to one workgroup per CU
        const int N = n + M;
        int wantc = coop_pref ? 1 : 0;
        int dev_now = 0;
        HIPCHK(hipGetDevice(&dev_now));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, dev_now));
        const int rw = 8;
        if (const char *ev = getenv("MIOSQP_COOP_DBG")) d.coop_dbg = atoi(ev);
        const int T = (N + rw - 1) / rw;
        const bool can = !e->resident && N <= 2048 && T <= prop.multiProcessorCount;
        if (wantc && can) {
          e->coop = true;
          e->coop_rw = rw;
          e->coop_cpt = N <= 1024 ? 2 : 4;
          e->coop_T = T;
          d.ldw = (N + 7) & ~7;
          double *Wd = nullptr;
          rc = dalloc(e, &Wd, (size_t)N * d.ldw + 64);
          if (!rc) rc = dalloc(e, &d.coop_tag, 64);
          d.coop_stride = 2 * rw;
          if (const char *ev = getenv("MIOSQP_COOP_STRIDE")) d.coop_stride = std::max(2 * rw, atoi(ev) / 8);
          d.coop_half = (size_t)T * d.coop_stride;
          int memkind = 0;
          if (const char *ev = getenv("MIOSQP_COOP_MEM")) memkind = atoi(ev);
          if (!rc && memkind) {
            void *pb = nullptr;
            const size_t bytes = (2 * d.coop_half + 64) * 8;
            HIPCHK(hipExtMallocWithFlags(&pb, bytes, memkind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
            HIPCHK(hipMemset(pb, 0, bytes));
            e->allocs.push_back(pb);
            d.coop_buf = (unsigned long long *)pb;
          } else if (!rc) rc = dalloc(e, &d.coop_buf, 2 * d.coop_half + 64);
          if (!rc) rc = dalloc(e, &d.coop_chk, 2 * d.coop_half + 64);
          if (!rc) rc = dalloc(e, &d.coop_q, (size_t)(256 + 16) * COOP_QS + 64);
          if (!rc) rc = dalloc(e, &d.coop_reg, 64);
          double *Kc = nullptr;
          if (!rc) rc = dalloc(e, &Kc, (size_t)N * d.ldw + 64);
          if (!rc) rc = miosqp_device_kkt_inverse(d.f_rows, d.ldf, d.d2inv, n, M, Wd, d.ldw, e->stream);
          if (rc) { miosqp_qp_cleanup(e); return rc; }
          d.W = Wd;
          d.Kc = Kc;
          hipLaunchKernelGGL(k_build_kc, dim3((N + 255) / 256, N), dim3(256), 0, e->stream, d, Kc);
        }
      }
      if (const char *ev = getenv("MIOSQP_BD_CFG")) e->bd_cfg = atoi(ev);
      if (const char *ev = getenv("MIOSQP_COMPACT")) e->compact = atoi(ev) != 0;
      if (const char *ev = getenv("MIOSQP_BM_ABLATE")) d.bm_ablate = atoi(ev);
      if (const char *ev = getenv("MIOSQP_FOLD_TPR")) {  // tuning hook: "fwd,x,c"
        int a = 0, b = 0, c = 0;
        if (sscanf(ev, "%d,%d,%d", &a, &b, &c) == 3) {
          e->tpr_ff = a;
          e->tpr_fx = b;
          e->tpr_fc = c;
        }
      }
      std::vector<double>().swap(e->fo.rows);
      std::vector<double>().swap(e->fo.GmT);
      std::vector<double>().swap(e->fo.Ad);
      std::vector<double>().swap(e->fo.Atd);
      std::vector<double>().swap(e->fo.Pd);
    }
  }
  e->chunk = e->st.check_termination;
  e->tail_iters = e->st.max_iter % e->chunk;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (!e->resident && !e->coop) {  // the single-launch solvers need no captured chunk
    int rc = capture_chunk(e, e->chunk, &e->g_full, &e->x_full);
    if (!rc && e->tail_iters > 0) rc = capture_chunk(e, e->tail_iters, &e->g_tail, &e->x_tail);
    if (rc) { miosqp_qp_cleanup(e); return rc; }
  }
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
  hipLaunchKernelGGL(k_init_wh, dim3(((e->M > e->n ? e->M : e->n) + 255) / 256), dim3(256), 0, e->stream, e->d);
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

int miosqp_qp_set_root(miosqp_qp_engine *e, const double *l_root, const double *u_root, double eps_int_feas,
                       double eps_lin) {
  if (!e || !l_root || !u_root || !e->have_int) {
    g_err = "set_root: call miosqp_qp_set_integer_rows first";
    return MIOSQP_EARG;
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  if (e->M > 0) {
    HIPCHK(hipMemcpy(e->d.root_l, l_root, sizeof(double) * e->M, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d.root_u, u_root, sizeof(double) * e->M, hipMemcpyHostToDevice));
  }
  e->d.eps_int = eps_int_feas;
  e->d.eps_lin = eps_lin;
  e->d.digest = 1;
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
  hipLaunchKernelGGL(k_init_wh, dim3(((M > n ? M : n) + 255) / 256), dim3(256), 0, e->stream, e->d);
  int rc = run_loop(e);
  if (!rc) rc = finish_and_fetch(e, 1, x_out, y_out, info, t0);
  return rc;
}

int miosqp_qp_solve_batch(miosqp_qp_engine *e, int32_t B, const double *l, const double *u, const double *x0,
                          const double *y0, double *x_out, double *y_out, miosqp_qp_info *info) {
  if (!e || B < 0 || (B > 0 && (!l || !u || !x0 || !y0 || !x_out || !y_out || !info))) return MIOSQP_EARG;
  if (!e->have_int) {
    g_err = "solve_batch: call miosqp_qp_set_integer_rows first";
    return MIOSQP_EARG;
  }
  const size_t n = e->n, M = e->M;
  for (size_t k = 0; k < (size_t)B * M; k++)
    if (l[k] > u[k]) return MIOSQP_EBOUNDS;
  if (e->Bcap == 0) {
    int cap = e->st.max_batch > 1 ? e->st.max_batch : 64;
    if (cap > 1024) cap = 1024;
    int rc = alloc_batch(e, cap);
    if (rc) return rc;
  }
  for (int s0 = 0; s0 < B; s0 += e->Bcap) {
    const int nb = B - s0 < e->Bcap ? B - s0 : e->Bcap;
    int rc = solve_slice(e, nb, l + s0 * M, u + s0 * M, x0 + s0 * n, y0 + s0 * M, x_out + s0 * n, y_out + s0 * M,
                         info + s0);
    if (rc) return rc;
  }
  return 0;
}

int miosqp_qp_debug_iterate(miosqp_qp_engine *e, int32_t k, double *x, double *z, double *y) {
  if (!e || k < 0) return MIOSQP_EARG;
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, e->d);
  hipLaunchKernelGGL(k_init_wh, dim3(((e->M > e->n ? e->M : e->n) + 255) / 256), dim3(256), 0, e->stream, e->d);
  if (e->resident) {
    if (k > 0) {
      int rc = launch_resident(e, k, 0, 0);
      if (rc) return rc;
    }
  } else if (e->coop) {
    if (k > 0) launch_coop(e, k, 0, 0);
  } else {
    for (int i = 0; i < k; i++) launch_iteration(e);
  }
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
  out[4] = e->tpr_pv; out[5] = e->tpr_pc; out[6] = e->tpr_tail; out[7] = (e->fold ? 1 : 0) | (e->resident ? 2 : 0) | (e->setup_on_device ? 4 : 0) | (e->coop ? 8 : 0);
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

// debug: per-block (start, end) wall-clock stamps (100 MHz) of ONE launch of a product-form kernel
int miosqp_qp_debug_timeline(miosqp_qp_engine *e, int32_t which, uint64_t *out, int32_t max_blocks,
                             int32_t *nblocks) {
  if (!e || !out || !e->fold || which < 0 || which > 2 || (which == 2 && !e->coop)) return MIOSQP_EARG;
  unsigned long long *buf = nullptr;
  HIPCHK(hipMalloc((void **)&buf, sizeof(unsigned long long) * 2 * 8192));
  HIPCHK(hipMemset(buf, 0, sizeof(unsigned long long) * 2 * 8192));
  for (int i = 0; i < 20; i++) launch_iteration(e);
  Dev saved = e->d;
  e->d.prof = buf;
  // which == 2: 1000 iterations of the cooperative solver; per workgroup 8 words: shader clocks of
  // thread 0 in {reduce, update+publish, gather}, iterations, {test operands+rows, test norms}, tests, -
  if (which == 2) launch_coop(e, 1000, 25, 0);
  else if (which == 0) launch_fold_fwd(e); else launch_fold_bwd(e);
  e->d = saved;
  HIPCHK(hipStreamSynchronize(e->stream));
  const int nb = max_blocks < 8192 ? max_blocks : 8192;
  HIPCHK(hipMemcpy(out, buf, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost));
  if (nblocks) *nblocks = nb;
  hipFree(buf);
  return 0;
}

// debug: shader cycles and 100 MHz wall ticks recorded by the last LDS-resident launch
int miosqp_qp_debug_clock(miosqp_qp_engine *e, double *cycles, double *ticks) {
  if (!e) return MIOSQP_EARG;
  HIPCHK(hipStreamSynchronize(e->stream));
  Ctrl c;
  HIPCHK(hipMemcpy(&c, e->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost));
  *cycles = c.nrm_dy;
  *ticks = c.nrm_dx;
  return 0;
}

// debug counters: 0 = wave compactions performed by solve_batch so far
int64_t miosqp_qp_debug_counter(miosqp_qp_engine *e, int32_t which) {
  if (!e) return -1;
  return which == 0 ? e->compactions : -1;
}

int miosqp_qp_get_batch_stats(miosqp_qp_engine *e, double *ms, int64_t *batch_iters, int64_t *node_iters,
                              int32_t reset) {
  if (!e) return MIOSQP_EARG;
  if (ms) *ms = e->bloop_ms;
  if (batch_iters) *batch_iters = e->bloop_iters;
  if (node_iters) *node_iters = e->bloop_node_iters;
  if (reset) {
    e->bloop_ms = 0.0;
    e->bloop_iters = e->bloop_node_iters = 0;
  }
  return 0;
}

int miosqp_qp_time_kernel(miosqp_qp_engine *e, int32_t which, int32_t reps, double *usec, double *bytes) {
  if (!e || which < 0 || (which > 4 && which < 10) || which > 14 || reps <= 0 || !usec) return MIOSQP_EARG;
  if (which >= 10 && e->Bcap == 0) {
    g_err = "time_kernel: batched kernels need a prior solve_batch";
    return MIOSQP_EARG;
  }
  const Dev &d = e->d;
  const int ntiles = e->Bcap / 64;
  if (which >= 10) hipLaunchKernelGGL(kb_reset, dim3((d.Bs + 255) / 256), dim3(256), 0, e->stream, d, e->Bcap);
  hipLaunchKernelGGL(k_reset_ctrl, dim3(1), dim3(1), 0, e->stream, d);
  auto one = [&]() {
    if (e->fold && which < 4) {
      if (which == 0) launch_fold_fwd(e);
      if (which == 1) launch_fold_bwd(e);
      return;
    }
    if (e->fold && which >= 10 && which < 14) {
      if (which == 10) launch_bd(e, ntiles, 0);
      if (which == 11) launch_bd(e, ntiles, 1);
      return;
    }
    switch (which) {
      case 10: hipLaunchKernelGGL(kb_panel_fwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d); break;
      case 11: hipLaunchKernelGGL(kb_tail_fwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d); break;
      case 12: hipLaunchKernelGGL(kb_tail_bwd, dim3((d.n + 3) / 4, ntiles), dim3(256), 0, e->stream, d); break;
      case 13: hipLaunchKernelGGL(kb_panel_bwd, dim3((d.M + 3) / 4, ntiles), dim3(256), 0, e->stream, d); break;
      case 14: launch_iteration_b(e, ntiles); break;
      case 0: DISPATCH_TPR(e->tpr_pv, k_panel_fwd, d.n, e->stream, d); break;
      case 1: DISPATCH_TPR(e->tpr_tail, k_tail_fwd, d.n, e->stream, d); break;
      case 2: DISPATCH_TPR(e->tpr_tail, k_tail_bwd, d.n, e->stream, d); break;
      case 3: DISPATCH_TPR(e->tpr_pc, k_panel_bwd, d.M, e->stream, d); break;
      default: launch_iteration(e); break;
    }
  };
  if (e->coop && which == 4) {  // `reps` iterations of the cooperative solver in ONE launch, no tests
    launch_coop(e, 5, 0, 0);
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    launch_coop(e, reps, 0, 0);
    HIPCHK(hipEventRecord(e->ev1, e->stream));
  } else {
    for (int i = 0; i < 5; i++) one();
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (int i = 0; i < reps; i++) one();
    HIPCHK(hipEventRecord(e->ev1, e->stream));
  }
  HIPCHK(hipStreamSynchronize(e->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
  *usec = 1e3 * ms / reps;
  if (bytes) {
    double b[5];
    kernel_bytes(e, b);
    if (which < 10) {
      *bytes = b[which];
      if (e->fold && which < 4)  // forward sweep = panel + tail forward, backward likewise
        *bytes = which == 0 ? b[0] + b[1] : which == 1 ? b[2] + b[3] : 0.0;
    } else {
      // batched: matrix terms once per launch, per-node vector terms times the columns
      const double n = e->n, M = e->M, np = (double)e->fa.nnz_panel, nt = (double)e->fa.nnz_tail, B = e->Bcap;
      const double mat[5] = {np * 12 + (n + 1) * 4, nt * 12 + (n + 1) * 4, nt * 12 + (n + 1) * 4,
                             np * 12 + (M + 1) * 4, 2 * (np + nt) * 12 + 2 * (n + M + 1) * 4 + (n + M) * 12};
      const double vec[5] = {(M + 3 * n) * 8, 3 * n * 8, 5 * n * 8, (n + 10 * M) * 8, (6 * n + 16 * M) * 8};
      *bytes = mat[which - 10] + B * vec[which - 10];
      if (e->fold && which < 14)
        *bytes = which == 10 ? mat[0] + mat[1] + B * (vec[0] + vec[1])
                 : which == 11 ? mat[2] + mat[3] + B * (vec[2] + vec[3]) : 0.0;
    }
  }
  return 0;
}

}  // extern "C"
