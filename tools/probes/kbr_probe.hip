// Feasibility probe for a persistent, register-resident form of the batched sweeps:
// 256 workgroups = 8 column groups (group g = block % 8 -> XCD g) x 32 members; a member keeps ITS rows of the three
// factor blocks in registers for the whole launch (forward: 16 rows x 1536; backward: 3 x 16 rows x 512), the
// right-hand sides (32 columns per group) travel through L2 with agent-scope loads/stores, two group barriers per
// iteration.  Synthetic data, config-3 shape.  Prints us per iteration for: full, no barriers, no loads.
//   hipcc --offload-arch=gfx950 -O3 kbr_probe.hip -o _bin/kbr_probe && _bin/kbr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KS = 8, FC = 12, BC = 4, BT = 3;  // waves, forward chunks per wave, backward chunks per wave, backward tiles
constexpr int NF = 512, KF = KS * FC * 16 /*1536*/, KB = KS * BC * 16 /*512*/, NCON = 1024, BS = 256;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ double ldv(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stv(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void group_barrier(unsigned long long *ctr, unsigned long long target, int mode) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (mode & 1) {
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
}

// one sweep of NCH chunks with the A fragments in registers; V: right-hand rows (stride BS), this group's 32 columns
template <int NCH, int mode, class AF>
__device__ __forceinline__ void sweep(AF a, const double *__restrict__ V, int w, int lane, d4 (&acc)[2]) {
  const int r = lane & 15, j = lane >> 4;
  acc[0] = (d4){0, 0, 0, 0};
  acc[1] = (d4){0, 0, 0, 0};
  d4 acc2[2] = {(d4){0, 0, 0, 0}, (d4){0, 0, 0, 0}};
  double b[2][4][2];
  // the base is made opaque HERE, inside the iteration: otherwise the 8 x NCH addresses are hoisted out of the loop
  // of iterations and kept in registers
  const double *vb = V + (size_t)(16 * w + 4 * j) * BS + 2 * r;
  asm volatile("" : "+v"(vb));
  auto fetch = [&](int c, int slot) {
    const double *p = vb + (size_t)(16 * KS * c) * BS;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (mode & 2) { b[slot][q][0] = ldv(p + (size_t)q * BS); b[slot][q][1] = ldv(p + (size_t)q * BS + 1); }
      else { b[slot][q][0] = 1e-3 * (q + c); b[slot][q][1] = 1e-3 * (q - c); }
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int s = c & 1;
    if (c + 1 < NCH) fetch(c + 1, s ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    const double a0 = a(c, 0), a1 = a(c, 1), a2 = a(c, 2), a3 = a(c, 3);
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[s][0][0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[s][0][1], acc[1], 0, 0, 0);
    acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[s][1][0], acc2[0], 0, 0, 0);
    acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[s][1][1], acc2[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b[s][2][0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b[s][2][1], acc[1], 0, 0, 0);
    acc2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b[s][3][0], acc2[0], 0, 0, 0);
    acc2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b[s][3][1], acc2[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  acc[0] += acc2[0];
  acc[1] += acc2[1];
}

// adds the KS partial tiles in wave order; thread e gets element (row e / 32, column e % 32)
__device__ __forceinline__ double reduce(const d4 (&acc)[2], double *lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, e = threadIdx.x;
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int g = 0; g < 4; g++) lds[((w * 2 + t) * 4 + g) * 64 + lane] = acc[t][g];
  __syncthreads();
  const int row = (e >> 5) & 15, col = e & 31, t = col & 1, g = row >> 2, ll = (col >> 1) + 16 * (row & 3);
  const int idx = (t * 4 + g) * 64 + ll;
  double s = lds[idx];
#pragma unroll
  for (int ww = 1; ww < KS; ww++) s += lds[ww * 512 + idx];
  __syncthreads();
  return s;
}

template <int mode>
__global__ __launch_bounds__(512) void kbr(const double *__restrict__ Ff, const double *__restrict__ Bx, const double *__restrict__ Bc,
                                           double *V, double *U, unsigned long long *bar, int iters) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *lds = smem;                       // KS * 512 doubles: the reduction
  double *la = smem + KS * 512;             // 2 tiles x BC chunks x 4 x 512 threads: the constraint tiles' A fragments
  const int g = blockIdx.x & 7, m = blockIdx.x >> 3;  // column group (-> XCD), member
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, j = lane >> 4;
  const int col0 = 32 * g, e = threadIdx.x, orow = (e >> 5) & 15, ocol = e & 31;
  double af[FC][4], ab[BC][4];
#pragma unroll
  for (int c = 0; c < FC; c++)
#pragma unroll
    for (int q = 0; q < 4; q++) af[c][q] = Ff[(size_t)(16 * m + r) * KF + 16 * (w + KS * c) + 4 * j + q];
#pragma unroll
  for (int t = 0; t < BT; t++)
#pragma unroll
    for (int c = 0; c < BC; c++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double *A = t == 0 ? Bx + (size_t)(16 * m + r) * KB : Bc + (size_t)(16 * (m + 32 * (t - 1)) + r) * KB;
        const double v = A[16 * (w + KS * c) + 4 * j + q];
        if (t == 0) ab[c][q] = v;
        else la[(((t - 1) * BC + c) * 4 + q) * 512 + threadIdx.x] = v;  // (conflict-free: consecutive threads, consecutive words)
      }
  unsigned long long *ctr = bar + 16 * g;
  unsigned long long target = 0;
  d4 acc[2];
  for (int it = 0; it < iters; it++) {
    sweep<FC, mode>([&](int c, int q) { return af[c][q]; }, V + col0, w, lane, acc);
    double s = reduce(acc, lds);
    stv(U + (size_t)(16 * m + orow) * BS + col0 + ocol, 1e-3 * s);
    target += 32;
    group_barrier(ctr, target, mode);
#pragma unroll
    for (int t = 0; t < BT; t++) {
      if (t == 0) sweep<BC, mode>([&](int c, int q) { return ab[c][q]; }, U + col0, w, lane, acc);
      else sweep<BC, mode>([&](int c, int q) { return la[(((t - 1) * BC + c) * 4 + q) * 512 + threadIdx.x]; }, U + col0, w, lane, acc);
      s = reduce(acc, lds);
      const int row = t == 0 ? NCON + 16 * m + orow : 16 * (m + 32 * (t - 1)) + orow;
      stv(V + (size_t)row * BS + col0 + ocol, 1e-3 * s + 1e-6);
    }
    target += 32;
    group_barrier(ctr, target, mode);
  }
}

int main() {
  std::vector<double> hF((size_t)NF * KF), hX((size_t)NF * KB), hC((size_t)NCON * KB), hV((size_t)(KF + 16) * BS, 0.01);
  for (size_t i = 0; i < hF.size(); i++) hF[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
  for (size_t i = 0; i < hX.size(); i++) hX[i] = ((i * 40503u) % 1000) * 1e-3 - 0.5;
  for (size_t i = 0; i < hC.size(); i++) hC[i] = ((i * 69069u) % 1000) * 1e-3 - 0.5;
  double *F, *X, *C, *V, *U;
  unsigned long long *bar;
  CK(hipMalloc(&F, hF.size() * 8)); CK(hipMalloc(&X, hX.size() * 8)); CK(hipMalloc(&C, hC.size() * 8));
  CK(hipMalloc(&V, hV.size() * 8)); CK(hipMalloc(&U, (size_t)(NF + 16) * BS * 8)); CK(hipMalloc(&bar, 8 * 16 * 8));
  CK(hipMemcpy(F, hF.data(), hF.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(X, hX.data(), hX.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(C, hC.data(), hC.size() * 8, hipMemcpyHostToDevice));
  const int LDS_BYTES = (KS * 512 + 2 * BC * 4 * 512) * 8;
  CK(hipFuncSetAttribute((const void *)kbr<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CK(hipFuncSetAttribute((const void *)kbr<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CK(hipFuncSetAttribute((const void *)kbr<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  CK(hipFuncSetAttribute((const void *)kbr<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[4] = {"no loads, no barriers", "barriers only", "loads only", "full"};
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 4; mode++) {
      CK(hipMemcpy(V, hV.data(), hV.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemset(U, 0, (size_t)(NF + 16) * BS * 8));
      CK(hipMemset(bar, 0, 8 * 16 * 8));
      const int iters = 200;
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(kbr<0>, dim3(256), dim3(512), LDS_BYTES, 0, F, X, C, V, U, bar, iters); break;
        case 1: hipLaunchKernelGGL(kbr<1>, dim3(256), dim3(512), LDS_BYTES, 0, F, X, C, V, U, bar, iters); break;
        case 2: hipLaunchKernelGGL(kbr<2>, dim3(256), dim3(512), LDS_BYTES, 0, F, X, C, V, U, bar, iters); break;
        default: hipLaunchKernelGGL(kbr<3>, dim3(256), dim3(512), LDS_BYTES, 0, F, X, C, V, U, bar, iters); break;
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      double chk = 0;
      CK(hipMemcpy(hV.data(), V, 8 * 64, hipMemcpyDeviceToHost));
      for (int i = 0; i < 8; i++) chk += hV[i];
      if (rep) printf("%-24s %7.2f us per iteration  (%d iterations, %.2f ms; checksum %.6g)\n", names[mode], 1e3 * ms / iters, iters, ms, chk);
      std::fill(hV.begin(), hV.end(), 0.01);
    }
  return 0;
}
