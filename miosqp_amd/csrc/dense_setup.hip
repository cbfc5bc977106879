// Device side of the one-time setup: the dense part of the KKT
// factorisation -- blocked LDL^T of the reduced Hessian S = Pbar + sigma I + rho Abar^T Abar, the
// inverse of its unit-lower factor, and the transpose -- on the GPU.  This is the one dense
// contraction of the whole path (n^3/3 flops each; 4e10 at config 5); everything is 64 x 64 tiles
// staged in LDS, fp64 vector FMA, fixed summation order.
//
//   ks_diag    factor one 64 x 64 diagonal block in LDS (right-looking, unblocked)
//   ks_panel   rows below it: L[i, block] and W = L * diag(d)
//   ks_update  trailing update  S[i,j] -= W[i,:] . L[j,:]   over lower-triangular tiles
//   ks_inv     block row I of X = L^-1:  X[I,J] = L_II^-1 ( [I == J] - sum_{K=J}^{I-1} L[I,K] X[K,J] )
//   ks_out     strict lower part of X -> Linv, its transpose -> LinvT
#include <hip/hip_runtime.h>

#include <cstdio>

#include "factor.hpp"  // DenseAccelCtx

namespace {

constexpr int NB = 64, LP = NB + 1;  // tile size, padded LDS row

#define SCK(call)                                                                           \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) {                                                                \
      fprintf(stderr, "[miosqp setup] %s: %s\n", #call, hipGetErrorString(e__));            \
      rc = -2;                                                                              \
      goto done;                                                                            \
    }                                                                                       \
  } while (0)

__global__ __launch_bounds__(256) void ks_diag(double *S, int ld, int jb, int w, double *d, int *flag) {
  __shared__ double A[NB][LP];
  __shared__ double col[NB];
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, j = e % NB;
    A[i][j] = (i < w && j <= i) ? S[(size_t)(jb + i) * ld + jb + j] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < w; j++) {
    const double dj = A[j][j];
    if (!(dj > 0.0)) {
      if (tid == 0) *flag = 1;
      return;  // uniform: every thread reads the same A[j][j]
    }
    if (tid > j && tid < w) {
      const double l = A[tid][j] / dj;
      col[tid] = l;
      A[tid][j] = l;
    }
    __syncthreads();
    // A[i][k] -= l_i d_j l_k for j < k <= i < w
    for (int e = tid; e < NB * NB; e += 256) {
      const int i = e / NB, k = e % NB;
      if (k > j && k <= i && i < w) A[i][k] -= col[i] * dj * col[k];
    }
    __syncthreads();
  }
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, j = e % NB;
    if (i < w && j < i) S[(size_t)(jb + i) * ld + jb + j] = A[i][j];
  }
  if (tid < w) d[jb + tid] = A[tid][tid];
}

// one wave per 64 rows: thread r owns row je + 64 b + r of the panel
__global__ __launch_bounds__(64) void ks_panel(double *S, int ld, int n, int jb, int w, const double *d, double *W) {
  __shared__ double P[NB][LP];   // panel rows
  __shared__ double DL[NB][LP];  // DL[j][k] = d_k * L_jj[j][k]
  __shared__ double dd[NB];
  const int r = threadIdx.x, je = jb + w;
  const int i0 = je + blockIdx.x * NB, i = i0 + r;
  for (int e = r; e < NB * NB; e += 64) {
    const int j = e / NB, k = e % NB;
    DL[j][k] = (j < w && k < j) ? S[(size_t)(jb + j) * ld + jb + k] * d[jb + k] : 0.0;
    const int ii = i0 + j;
    P[j][k] = (ii < n && k < w) ? S[(size_t)ii * ld + jb + k] : 0.0;
  }
  if (r < w) dd[r] = d[jb + r];
  __syncthreads();
  for (int j = 0; j < w; j++) {
    double v = P[r][j];
    for (int k = 0; k < j; k++) v -= P[r][k] * DL[j][k];
    P[r][j] = v / dd[j];
  }
  __syncthreads();
  for (int e = r; e < NB * NB; e += 64) {
    const int j = e / NB, k = e % NB;
    const int ii = i0 + j;
    if (ii < n && k < w) {
      S[(size_t)ii * ld + jb + k] = P[j][k];
      W[(size_t)ii * NB + k] = P[j][k] * dd[k];
    }
  }
  (void)i;
}

// 64 x 64 tile (ti, tj), tj <= ti, of the trailing matrix: S -= W L^T over the current 64 columns
__global__ __launch_bounds__(256) void ks_update(double *S, int ld, int n, int jb, int w, const double *W) {
  const int je = jb + w;
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (tj > ti) return;
  __shared__ double Wt[NB][LP], Lt[NB][LP];
  const int i0 = je + ti * NB, j0 = je + tj * NB;
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  for (int e = tid; e < NB * NB; e += 256) {
    const int a = e / NB, k = e % NB;
    Wt[a][k] = (i0 + a < n && k < w) ? W[(size_t)(i0 + a) * NB + k] : 0.0;
    Lt[a][k] = (j0 + a < n && k < w) ? S[(size_t)(j0 + a) * ld + jb + k] : 0.0;
  }
  __syncthreads();
  double acc[4][4] = {};
  for (int k = 0; k < NB; k++) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      a[u] = Wt[ty + 16 * u][k];
      b[u] = Lt[tx + 16 * u][k];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int v = 0; v < 4; v++) acc[u][v] = fma(a[u], b[v], acc[u][v]);
  }
#pragma unroll
  for (int u = 0; u < 4; u++)
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int i = i0 + ty + 16 * u, j = j0 + tx + 16 * v;
      if (i < n && j <= i) S[(size_t)i * ld + j] -= acc[u][v];
    }
}

// block row I of X = L^-1 (X holds explicit ones on its diagonal while it is being built)
__global__ __launch_bounds__(256) void ks_inv(const double *L, double *X, int ld, int n, int I) {
  const int J = blockIdx.x;  // column block, J <= I
  __shared__ double At[NB][LP], Bt[NB][LP];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int i0 = I * NB, j0 = J * NB;
  double acc[4][4] = {};
  for (int K = J; K < I; K++) {
    const int k0 = K * NB;
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) {
      const int a = e / NB, k = e % NB;
      At[a][k] = (i0 + a < n) ? L[(size_t)(i0 + a) * ld + k0 + k] : 0.0;                  // L[I,K]
      Bt[a][k] = (k0 + a < n && j0 + k < n) ? X[(size_t)(k0 + a) * ld + j0 + k] : 0.0;   // X[K,J]
    }
    __syncthreads();
    for (int k = 0; k < NB; k++) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        a[u] = At[ty + 16 * u][k];
        b[u] = Bt[k][tx + 16 * u];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int v = 0; v < 4; v++) acc[u][v] = fma(a[u], b[v], acc[u][v]);
    }
  }
  __syncthreads();
  // R = [I == J] - acc into Bt, L_II (unit lower) into At, then one thread per column substitutes
#pragma unroll
  for (int u = 0; u < 4; u++)
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int r = ty + 16 * u, c = tx + 16 * v;
      Bt[r][c] = ((I == J && r == c) ? 1.0 : 0.0) - acc[u][v];
    }
  for (int e = tid; e < NB * NB; e += 256) {
    const int a = e / NB, k = e % NB;
    At[a][k] = (i0 + a < n && k < a) ? L[(size_t)(i0 + a) * ld + i0 + k] : 0.0;
  }
  __syncthreads();
  if (tid < NB) {
    const int c = tid;
    for (int r = 1; r < NB; r++) {
      double v = Bt[r][c];
      for (int k = 0; k < r; k++) v -= At[r][k] * Bt[k][c];
      Bt[r][c] = v;
    }
  }
  __syncthreads();
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e / NB, c = e % NB;
    if (i0 + r < n && j0 + c < n) X[(size_t)(i0 + r) * ld + j0 + c] = (I == J && c > r) ? 0.0 : Bt[r][c];
  }
}

// Linv = strict lower part of X, LinvT = its transpose (64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void ks_out(const double *X, double *Linv, double *LinvT, int ld, int n) {
  __shared__ double T[NB][LP];
  const int i0 = blockIdx.x * NB, j0 = blockIdx.y * NB, tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e / NB, c = e % NB;
    const int i = i0 + r, j = j0 + c;
    const double v = (i < n && j < i) ? X[(size_t)i * ld + j] : 0.0;
    T[r][c] = v;
    if (i < n && j < n) Linv[(size_t)i * ld + j] = v;
  }
  __syncthreads();
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e / NB, c = e % NB;  // LinvT[j0 + r][i0 + c] = X[i0 + c][j0 + r]
    if (j0 + r < n && i0 + c < n) LinvT[(size_t)(j0 + r) * ld + i0 + c] = T[c][r];
  }
}

// Row i1 of S = Pbar + sigma I + rho Abar^T Abar (lower triangle; the rest of the row is written as zeros) by ONE
// workgroup, accumulated in LDS in exactly the host's order (factor.cpp): the row starts as Pbar's entries, then sigma
// on the diagonal, then for every constraint row r that holds variable i1 (ascending) w = rho Abar[r][i1] times row r
// of Abar up to column i1, one fused multiply-add per entry.  Entries of one constraint row are distinct columns
// (the zero-valued pad that repeats the last column is skipped: it adds nothing), so the lanes never collide; a
// barrier separates consecutive constraint rows.
__global__ __launch_bounds__(256) void ks_schur_row(double *__restrict__ S, int ld, int n, double rho, double sigma,
                                                    const int *__restrict__ Pp, const int *__restrict__ Pi,
                                                    const double *__restrict__ Px, const int *__restrict__ Ap,
                                                    const int *__restrict__ Ai, const double *__restrict__ Ax,
                                                    const int *__restrict__ Rptr, const int *__restrict__ Ridx,
                                                    const double *__restrict__ Rval) {
  extern __shared__ double acc[];
  const int i1 = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < ld; c += 256) acc[c] = 0.0;
  __syncthreads();
  for (int p = Pp[i1] + tid; p < Pp[i1 + 1]; p += 256) acc[Pi[p]] += Px[p];  // (i <= i1, distinct)
  __syncthreads();
  if (tid == 0) acc[i1] += sigma;
  __syncthreads();
  for (int p = Ap[i1]; p < Ap[i1 + 1]; p++) {
    const int r = Ai[p];
    const double w = rho * Ax[p];
    const int k0 = Rptr[r], k1 = Rptr[r + 1];
    for (int k = k0 + tid; k < k1; k += 256) {
      const int i2 = Ridx[k];
      if (i2 > i1) continue;
      if (k > k0 && Ridx[k - 1] == i2) continue;  // the pad
      acc[i2] = fma(w, Rval[k], acc[i2]);
    }
    __syncthreads();
  }
  for (int c = tid; c < ld; c += 256) S[(size_t)i1 * ld + c] = acc[c];
}

}  // namespace

// matches miosqp::DenseLdlInv (factor.hpp)
int miosqp_device_ldl_inverse(int n, int ld, const double *S, double *d, double *Linv, double *LinvT, void *ctx) {
  miosqp::DenseAccelCtx *actx = static_cast<miosqp::DenseAccelCtx *>(ctx);
  hipStream_t st = actx ? (hipStream_t)actx->stream : nullptr;
  const bool keep = actx && actx->keep_on_device;
  double *dS = nullptr, *dX = nullptr, *dT = nullptr, *dW = nullptr, *dd = nullptr;
  char *dIn = nullptr;
  int *dflag = nullptr, hflag = 0, rc = 0;
  const size_t mat = (size_t)n * ld * sizeof(double);
  const int nt = (n + NB - 1) / NB;
  SCK(hipMalloc((void **)&dS, mat + 64 * sizeof(double)));  // (+64: the engine's tile loads may overshoot a row)
  SCK(hipMalloc((void **)&dX, mat));
  SCK(hipMalloc((void **)&dT, mat + 64 * sizeof(double)));
  SCK(hipMalloc((void **)&dW, (size_t)n * NB * sizeof(double)));
  SCK(hipMalloc((void **)&dd, (size_t)n * sizeof(double)));
  SCK(hipMalloc((void **)&dflag, sizeof(int)));
  if (S) {
    SCK(hipMemcpyAsync(dS, S, mat, hipMemcpyHostToDevice, st));
  } else {
    // S assembled here from the scaled matrices (factor.hpp: schur_on_device)
    if (!actx || !actx->schur_on_device) { rc = -2; goto done; }
    const size_t bi = sizeof(int), bd = sizeof(double);
    const size_t sizes[9] = {(size_t)(n + 1) * bi, (size_t)actx->nnzP * bi, (size_t)actx->nnzP * bd, (size_t)(n + 1) * bi,
                             (size_t)actx->nnzA * bi, (size_t)actx->nnzA * bd, (size_t)(actx->M + 1) * bi,
                             (size_t)actx->nnzR * bi, (size_t)actx->nnzR * bd};
    const void *src[9] = {actx->Pp, actx->Pi, actx->Px, actx->Ap, actx->Ai, actx->Ax, actx->Rptr, actx->Ridx, actx->Rval};
    size_t off[10];
    off[0] = 0;
    for (int k = 0; k < 9; k++) off[k + 1] = off[k] + ((sizes[k] + 255) & ~(size_t)255);
    SCK(hipMalloc((void **)&dIn, off[9] + 256));
    for (int k = 0; k < 9; k++)
      if (sizes[k]) SCK(hipMemcpyAsync(dIn + off[k], src[k], sizes[k], hipMemcpyHostToDevice, st));
    SCK(hipFuncSetAttribute((const void *)ks_schur_row, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(ld * bd)));
    hipLaunchKernelGGL(ks_schur_row, dim3(n), dim3(256), ld * bd, st, dS, ld, n, actx->rho, actx->sigma,
                       (const int *)(dIn + off[0]), (const int *)(dIn + off[1]), (const double *)(dIn + off[2]),
                       (const int *)(dIn + off[3]), (const int *)(dIn + off[4]), (const double *)(dIn + off[5]),
                       (const int *)(dIn + off[6]), (const int *)(dIn + off[7]), (const double *)(dIn + off[8]));
  }
  SCK(hipMemsetAsync(dX, 0, mat, st));
  SCK(hipMemsetAsync(dT, 0, mat, st));
  SCK(hipMemsetAsync(dflag, 0, sizeof(int), st));
  for (int jb = 0; jb < n; jb += NB) {
    const int w = n - jb < NB ? n - jb : NB, je = jb + w;
    hipLaunchKernelGGL(ks_diag, dim3(1), dim3(256), 0, st, dS, ld, jb, w, dd, dflag);
    if (je >= n) break;
    const int rt = (n - je + NB - 1) / NB;
    hipLaunchKernelGGL(ks_panel, dim3(rt), dim3(64), 0, st, dS, ld, n, jb, w, dd, dW);
    hipLaunchKernelGGL(ks_update, dim3(rt, rt), dim3(256), 0, st, dS, ld, n, jb, w, dW);
  }
  SCK(hipMemcpyAsync(&hflag, dflag, sizeof(int), hipMemcpyDeviceToHost, st));
  SCK(hipStreamSynchronize(st));
  if (hflag) {
    rc = 1;
    goto done;
  }
  for (int I = 0; I < nt; I++) hipLaunchKernelGGL(ks_inv, dim3(I + 1), dim3(256), 0, st, dS, dX, ld, n, I);
  // dS is free now: reuse it for Linv
  hipLaunchKernelGGL(ks_out, dim3(nt, nt), dim3(256), 0, st, dX, dS, dT, ld, n);
  SCK(hipMemcpyAsync(d, dd, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
  if (!keep) {
    SCK(hipMemcpyAsync(Linv, dS, mat, hipMemcpyDeviceToHost, st));
    SCK(hipMemcpyAsync(LinvT, dT, mat, hipMemcpyDeviceToHost, st));
  }
  SCK(hipStreamSynchronize(st));
  if (keep) {  // the caller owns them from here on
    actx->dLinv = dS;
    actx->dLinvT = dT;
    dS = dT = nullptr;
  }
done:
  if (dIn) hipFree(dIn);
  if (dS) hipFree(dS);
  hipFree(dX);
  if (dT) hipFree(dT);
  hipFree(dW);
  hipFree(dd);
  hipFree(dflag);
  return rc;
}

// ------------------------------------------------------------------------------------------
// Equilibration on the device (miosqp::RuizOps, factor.hpp): column / row maxima and element-wise products over
// device-resident copies of P (upper triangle, CSC) and A (CSC).  Maxima of non-negative doubles are taken on their
// bit patterns with integer atomics (order-independent); the products are the host's two roundings
// (t = dt[col] * dt[row], then value * t).  One wavefront per column.
// ------------------------------------------------------------------------------------------
namespace {

struct DevRuiz {
  hipStream_t st = nullptr;
  int n = 0, M = 0;
  int64_t nnzP = 0, nnzA = 0;
  char *buf = nullptr;  // one allocation: Pp Pi Px Ap Ai Ax dt et dn en
  int *Pp = nullptr, *Pi = nullptr, *Ap = nullptr, *Ai = nullptr;
  double *Px = nullptr, *Ax = nullptr, *dt = nullptr, *et = nullptr;
  unsigned long long *dn = nullptr, *en = nullptr;
};

__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ __launch_bounds__(256) void kr_norms(int n, const int *__restrict__ Pp, const int *__restrict__ Pi,
                                                const double *__restrict__ Px, const int *__restrict__ Ap,
                                                const int *__restrict__ Ai, const double *__restrict__ Ax,
                                                unsigned long long *dn, unsigned long long *en, int with_A) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= n) return;
  double m = 0.0;
  for (int p = Pp[j] + lane; p < Pp[j + 1]; p += 64) {
    const double a = fabs(Px[p]);
    m = fmax(m, a);
    const int i = Pi[p];
    if (i != j) atomicMax(dn + i, (unsigned long long)__double_as_longlong(a));
  }
  if (with_A)
    for (int p = Ap[j] + lane; p < Ap[j + 1]; p += 64) {
      const double a = fabs(Ax[p]);
      m = fmax(m, a);
      atomicMax(en + Ai[p], (unsigned long long)__double_as_longlong(a));
    }
  m = wave_max_d(m);
  if (lane == 0) atomicMax(dn + j, (unsigned long long)__double_as_longlong(m));
}

__global__ __launch_bounds__(256) void kr_scale(int n, const int *__restrict__ Pp, const int *__restrict__ Pi, double *Px,
                                                const int *__restrict__ Ap, const int *__restrict__ Ai, double *Ax,
                                                const double *__restrict__ dt, const double *__restrict__ et) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= n) return;
  const double dj = dt[j];
  for (int p = Pp[j] + lane; p < Pp[j + 1]; p += 64) {
    const double t = dj * dt[Pi[p]];
    Px[p] = Px[p] * t;
  }
  for (int p = Ap[j] + lane; p < Ap[j + 1]; p += 64) {
    const double t = dj * et[Ai[p]];
    Ax[p] = Ax[p] * t;
  }
}

__global__ void kr_scale_cost(int64_t nnz, double *Px, double ct) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < nnz) Px[p] = Px[p] * ct;
}

#define RCK(call)                                                                      \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) {                                                           \
      fprintf(stderr, "[miosqp setup] %s: %s\n", #call, hipGetErrorString(e__));       \
      return -2;                                                                       \
    }                                                                                  \
  } while (0)

int ruiz_begin(void *ctx, int n, int M, const int *Pp, const int *Pi, const double *Px, const int *Ap, const int *Ai,
               const double *Ax) {
  DevRuiz &r = *static_cast<DevRuiz *>(ctx);
  r.n = n; r.M = M; r.nnzP = Pp[n]; r.nnzA = Ap[n];
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t sz[10] = {al((size_t)(n + 1) * 4), al((size_t)r.nnzP * 4), al((size_t)r.nnzP * 8), al((size_t)(n + 1) * 4),
                         al((size_t)r.nnzA * 4), al((size_t)r.nnzA * 8), al((size_t)n * 8), al((size_t)(M + 1) * 8),
                         al((size_t)n * 8), al((size_t)(M + 1) * 8)};
  size_t off[11];
  off[0] = 0;
  for (int k = 0; k < 10; k++) off[k + 1] = off[k] + sz[k];
  RCK(hipMalloc((void **)&r.buf, off[10] + 256));
  r.Pp = (int *)(r.buf + off[0]); r.Pi = (int *)(r.buf + off[1]); r.Px = (double *)(r.buf + off[2]);
  r.Ap = (int *)(r.buf + off[3]); r.Ai = (int *)(r.buf + off[4]); r.Ax = (double *)(r.buf + off[5]);
  r.dt = (double *)(r.buf + off[6]); r.et = (double *)(r.buf + off[7]);
  r.dn = (unsigned long long *)(r.buf + off[8]); r.en = (unsigned long long *)(r.buf + off[9]);
  RCK(hipMemcpyAsync(r.Pp, Pp, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, r.st));
  RCK(hipMemcpyAsync(r.Pi, Pi, (size_t)r.nnzP * 4, hipMemcpyHostToDevice, r.st));
  RCK(hipMemcpyAsync(r.Px, Px, (size_t)r.nnzP * 8, hipMemcpyHostToDevice, r.st));
  RCK(hipMemcpyAsync(r.Ap, Ap, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, r.st));
  if (r.nnzA) {
    RCK(hipMemcpyAsync(r.Ai, Ai, (size_t)r.nnzA * 4, hipMemcpyHostToDevice, r.st));
    RCK(hipMemcpyAsync(r.Ax, Ax, (size_t)r.nnzA * 8, hipMemcpyHostToDevice, r.st));
  }
  return 0;
}

int ruiz_norms(void *ctx, double *dt, double *et, int with_A) {
  DevRuiz &r = *static_cast<DevRuiz *>(ctx);
  RCK(hipMemsetAsync(r.dn, 0, (size_t)r.n * 8, r.st));
  if (with_A) RCK(hipMemsetAsync(r.en, 0, (size_t)(r.M + 1) * 8, r.st));
  hipLaunchKernelGGL(kr_norms, dim3((r.n + 3) / 4), dim3(256), 0, r.st, r.n, r.Pp, r.Pi, r.Px, r.Ap, r.Ai, r.Ax, r.dn, r.en,
                     with_A);
  RCK(hipMemcpyAsync(dt, r.dn, (size_t)r.n * 8, hipMemcpyDeviceToHost, r.st));
  if (with_A && r.M) RCK(hipMemcpyAsync(et, r.en, (size_t)r.M * 8, hipMemcpyDeviceToHost, r.st));
  RCK(hipStreamSynchronize(r.st));
  return 0;
}

int ruiz_scale(void *ctx, const double *dt, const double *et) {
  DevRuiz &r = *static_cast<DevRuiz *>(ctx);
  RCK(hipMemcpyAsync(r.dt, dt, (size_t)r.n * 8, hipMemcpyHostToDevice, r.st));
  if (r.M) RCK(hipMemcpyAsync(r.et, et, (size_t)r.M * 8, hipMemcpyHostToDevice, r.st));
  RCK(hipStreamSynchronize(r.st));  // the sources are the caller's vectors
  hipLaunchKernelGGL(kr_scale, dim3((r.n + 3) / 4), dim3(256), 0, r.st, r.n, r.Pp, r.Pi, r.Px, r.Ap, r.Ai, r.Ax, r.dt, r.et);
  return 0;
}

int ruiz_scale_cost(void *ctx, double ct) {
  DevRuiz &r = *static_cast<DevRuiz *>(ctx);
  if (r.nnzP)
    hipLaunchKernelGGL(kr_scale_cost, dim3((unsigned)((r.nnzP + 255) / 256)), dim3(256), 0, r.st, r.nnzP, r.Px, ct);
  return 0;
}

int ruiz_end(void *ctx, double *Px, double *Ax) {
  DevRuiz &r = *static_cast<DevRuiz *>(ctx);
  int rc = 0;
  if (r.buf) {
    if (hipMemcpyAsync(Px, r.Px, (size_t)r.nnzP * 8, hipMemcpyDeviceToHost, r.st) != hipSuccess) rc = -2;
    if (r.nnzA && hipMemcpyAsync(Ax, r.Ax, (size_t)r.nnzA * 8, hipMemcpyDeviceToHost, r.st) != hipSuccess) rc = -2;
    if (hipStreamSynchronize(r.st) != hipSuccess) rc = -2;
    hipFree(r.buf);
    r.buf = nullptr;
  }
  return rc;
}

}  // namespace

// fills `ops` with the device implementation; `storage` must stay alive until ops->end has been called
struct MiosqpDevRuizStorage {
  DevRuiz r;
};
void miosqp_device_ruiz_ops(miosqp::RuizOps *ops, void **storage, void *stream) {
  MiosqpDevRuizStorage *s = new MiosqpDevRuizStorage();
  s->r.st = (hipStream_t)stream;
  *storage = s;
  ops->ctx = &s->r;
  ops->begin = ruiz_begin;
  ops->norms = ruiz_norms;
  ops->scale = ruiz_scale;
  ops->scale_cost = ruiz_scale_cost;
  ops->end = ruiz_end;
}
void miosqp_device_ruiz_free(void *storage) {
  MiosqpDevRuizStorage *s = static_cast<MiosqpDevRuizStorage *>(storage);
  if (s && s->r.buf) hipFree(s->r.buf);
  delete s;
}

// ------------------------------------------------------------------------------------------
// Explicit KKT inverse for the register-resident cooperative solver (engine.hip, k_coop):
//   W = F^T D22^-1 F ,   F = [ -G | L22^-1 ]  (n x N, N = M + n; the rows of the product-form
//   factor with their unit diagonal restored),  so that  W [wh ; rx] = [ rho A x~ + .. ; x~ ].
// One 64 x 64 tile of W per workgroup, 4 x 4 per thread, 16 factor rows per step through LDS.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ks_kkt_inverse(const double *__restrict__ F, int ldf, const double *__restrict__ dinv,
                                                      int n, int M, double *__restrict__ W, int ldw) {
  __shared__ double As[16][64 + 1], Bs[16][64 + 1];
  const int N = n + M;
  const int a0 = blockIdx.y * 64, b0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[4][4] = {};
  // column c >= M of F is zero above row c - M
  const int lo = a0 > b0 ? a0 : b0;
  int i0 = lo > M ? ((lo - M) & ~15) : 0;
  for (; i0 < n; i0 += 16) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int k = e >> 6, c = e & 63, i = i0 + k;
      double va = 0.0, vb = 0.0;
      if (i < n) {
        const int ca = a0 + c, cb = b0 + c;
        const double di = dinv[i];
        if (ca < N) va = (ca == M + i ? 1.0 : (ca > M + i ? 0.0 : F[(size_t)i * ldf + ca])) * di;
        if (cb < N) vb = cb == M + i ? 1.0 : (cb > M + i ? 0.0 : F[(size_t)i * ldf + cb]);
      }
      As[k][c] = va;
      Bs[k][c] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) {
      double a[4], b[4];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        a[p] = As[k][ty + 16 * p];
        b[p] = Bs[k][tx + 16 * p];
      }
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[p][q] = fma(a[p], b[q], acc[p][q]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int a = a0 + ty + 16 * p, b = b0 + tx + 16 * q;
      if (a < N && b < N) W[(size_t)a * ldw + b] = acc[p][q];
    }
}

// F, dinv, W are device pointers; runs on `stream`
int miosqp_device_kkt_inverse(const double *F, int ldf, const double *dinv, int n, int M, double *W, int ldw,
                              hipStream_t stream) {
  const int N = n + M, nt = (N + 63) / 64;
  hipLaunchKernelGGL(ks_kkt_inverse, dim3(nt, nt), dim3(256), 0, stream, F, ldf, dinv, n, M, W, ldw);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
