// kbr_probe.hip with ONE wave per SIMD: 256 threads per workgroup, every wave plays two of the eight k-slices ("virtual waves"
// w and w + 4, their own accumulators, so the sums are the ones of the 8-wave kernel), 512 registers per lane: fragments in
// registers as before, but room for a deeper operand pipeline.  Synthetic data, config-3 shape; prints us per iteration.
//   hipcc --offload-arch=gfx950 -O3 kbr_probe2.hip -o _bin/kbr_probe2 && _bin/kbr_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int KS = 8, RW = 4, FC = 12, BC = 4, BT = 3, DEPTH = 3;
constexpr int NF = 512, KF = KS * FC * 16 /*1536*/, KB = KS * BC * 16 /*512*/, NCON = 1024, BS = 256;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void stv(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void group_barrier(unsigned long long *ctr, unsigned long long target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// one sweep of NCH chunks per virtual wave (two virtual waves per wave), A fragments at hand; operands DEPTH chunks ahead
template <int NCH, class AF>
__device__ __forceinline__ void sweep(AF a, __amdgpu_buffer_rsrc_t V, unsigned voff, unsigned rowbytes, int w, d4 (&acc)[2][2]) {
  d4 acc2[2][2];
#pragma unroll
  for (int v = 0; v < 2; v++)
#pragma unroll
    for (int t = 0; t < 2; t++) { acc[v][t] = (d4){0, 0, 0, 0}; acc2[v][t] = (d4){0, 0, 0, 0}; }
  double2 b[DEPTH][2][4];
  asm volatile("" : "+v"(voff));
  auto fetch = [&](int c, int slot) {
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const unsigned o = voff + (unsigned)(16 * (w + RW * v + KS * c)) * rowbytes;
#pragma unroll
      for (int q = 0; q < 4; q++) b[slot][v][q] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(V, o + (unsigned)q * rowbytes, 0, 16));
    }
  };
#pragma unroll
  for (int c = 0; c < DEPTH - 1 && c < NCH; c++) fetch(c, c);
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int s = c % DEPTH;
    if (c + DEPTH - 1 < NCH) fetch(c + DEPTH - 1, (c + DEPTH - 1) % DEPTH);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const double a0 = a(v, c, 0), a1 = a(v, c, 1), a2 = a(v, c, 2), a3 = a(v, c, 3);
      acc[v][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[s][v][0].x, acc[v][0], 0, 0, 0);
      acc[v][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[s][v][0].y, acc[v][1], 0, 0, 0);
      acc2[v][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[s][v][1].x, acc2[v][0], 0, 0, 0);
      acc2[v][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[s][v][1].y, acc2[v][1], 0, 0, 0);
      acc[v][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b[s][v][2].x, acc[v][0], 0, 0, 0);
      acc[v][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b[s][v][2].y, acc[v][1], 0, 0, 0);
      acc2[v][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b[s][v][3].x, acc2[v][0], 0, 0, 0);
      acc2[v][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b[s][v][3].y, acc2[v][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int v = 0; v < 2; v++) { acc[v][0] += acc2[v][0]; acc[v][1] += acc2[v][1]; }
}

// adds the KS partial tiles in virtual-wave order; thread e gets elements e and e + 256 (row (e' / 32) % 16, column e' % 32)
__device__ __forceinline__ void reduce(const d4 (&acc)[2][2], double *lds, double (&out)[2]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < 2; v++)
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int g = 0; g < 4; g++) lds[(((w + RW * v) * 2 + t) * 4 + g) * 64 + lane] = acc[v][t][g];
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int e = threadIdx.x + 256 * h;
    const int row = (e >> 5) & 15, col = e & 31, t = col & 1, g = row >> 2, ll = (col >> 1) + 16 * (row & 3);
    const int idx = (t * 4 + g) * 64 + ll;
    double s = lds[idx];
#pragma unroll
    for (int ww = 1; ww < KS; ww++) s += lds[ww * 512 + idx];
    out[h] = s;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void kbr(const double *__restrict__ Ff, const double *__restrict__ Bx, const double *__restrict__ Bc,
                                           double *V, double *U, unsigned long long *bar, int iters) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double *lds = smem;                       // KS * 512 doubles: the reduction
  double *la = smem + KS * 512;             // 2 tiles x 2 virtual waves x BC chunks x 4 x 256 threads
  const int g = blockIdx.x & 7, m = blockIdx.x >> 3;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, j = lane >> 4;
  const int col0 = 32 * g;
  double af[2][FC][4], ab[2][BC][4];
#pragma unroll
  for (int v = 0; v < 2; v++)
#pragma unroll
    for (int c = 0; c < FC; c++)
#pragma unroll
      for (int q = 0; q < 4; q++) af[v][c][q] = Ff[(size_t)(16 * m + r) * KF + 16 * (w + RW * v + KS * c) + 4 * j + q];
#pragma unroll
  for (int t = 0; t < BT; t++)
#pragma unroll
    for (int v = 0; v < 2; v++)
#pragma unroll
      for (int c = 0; c < BC; c++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const double *A = t == 0 ? Bx + (size_t)(16 * m + r) * KB : Bc + (size_t)(16 * (m + 32 * (t - 1)) + r) * KB;
          const double val = A[16 * (w + RW * v + KS * c) + 4 * j + q];
          if (t == 0) ab[v][c][q] = val;
          else la[((((t - 1) * 2 + v) * BC + c) * 4 + q) * 256 + threadIdx.x] = val;
        }
  const unsigned rowbytes = BS * 8;
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void *)V, 0, (KF + 16) * BS * 8, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void *)U, 0, (NF + 16) * BS * 8, 0x00020000);
  const unsigned lane_off = (unsigned)(4 * j) * rowbytes + (unsigned)(2 * r + col0) * 8u;
  unsigned long long *ctr = bar + 16 * g;
  unsigned long long target = 0;
  d4 acc[2][2];
  double out[2];
  for (int it = 0; it < iters; it++) {
    sweep<FC>([&](int v, int c, int q) { return af[v][c][q]; }, rsV, lane_off, rowbytes, w, acc);
    reduce(acc, lds, out);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int e = threadIdx.x + 256 * h;
      stv(U + (size_t)(16 * m + ((e >> 5) & 15)) * BS + col0 + (e & 31), 1e-3 * out[h]);
    }
    target += 32;
    group_barrier(ctr, target);
#pragma unroll
    for (int t = 0; t < BT; t++) {
      if (t == 0) sweep<BC>([&](int v, int c, int q) { return ab[v][c][q]; }, rsU, lane_off, rowbytes, w, acc);
      else sweep<BC>([&](int v, int c, int q) { return la[((((t - 1) * 2 + v) * BC + c) * 4 + q) * 256 + threadIdx.x]; }, rsU, lane_off, rowbytes, w, acc);
      reduce(acc, lds, out);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int e = threadIdx.x + 256 * h;
        const int row = (t == 0 ? NCON + 16 * m : 16 * (m + 32 * (t - 1))) + ((e >> 5) & 15);
        stv(V + (size_t)row * BS + col0 + (e & 31), 1e-3 * out[h] + 1e-6);
      }
    }
    target += 32;
    group_barrier(ctr, target);
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
  const int LDS_BYTES = (KS * 512 + 2 * 2 * BC * 4 * 256) * 8;
  CK(hipFuncSetAttribute((const void *)kbr, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipMemcpy(V, hV.data(), hV.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(U, 0, (size_t)(NF + 16) * BS * 8));
    CK(hipMemset(bar, 0, 8 * 16 * 8));
    const int iters = 200;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kbr, dim3(256), dim3(256), LDS_BYTES, 0, F, X, C, V, U, bar, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double chk = 0;
    CK(hipMemcpy(hV.data(), V, 8 * 64, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) chk += hV[i];
    if (rep) printf("one wave per SIMD, depth %d: %7.2f us per iteration  (%d iterations, %.2f ms; checksum %.6g)\n", DEPTH, 1e3 * ms / iters, iters, ms, chk);
    std::fill(hV.begin(), hV.end(), 0.01);
  }
  return 0;
}
