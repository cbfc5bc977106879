// fp64 issue rates on gfx950: v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64 and v_fma_f64, cycles per instruction per
// SIMD (s_memtime shader clock) and wall time (-> effective clock, chip TFLOP/s).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void k_mfma16(double *out, long long *cyc, int iters, double a, double b) {
  d4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) acc[c] = (d4){0, 0, 0, 0};
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
__global__ void k_mfma4(double *out, long long *cyc, int iters, double a, double b) {
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) acc[c] = 0;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CH>
__global__ void k_fma(double *out, long long *cyc, int iters, double a, double b) {
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) acc[c] = threadIdx.x * 1e-9 + c;
  const long long t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CH; c++) acc[c] = fma(acc[c], a, b);
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
void run(const char *name, F launch, int blocks, int threads, int iters, int ch, double flop_per_inst) {
  double *out; long long *cyc;
  hipMalloc(&out, sizeof(double) * blocks * threads);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[1024]; hipMemcpy(h, cyc, sizeof(long long) * (blocks < 1024 ? blocks : 1024), hipMemcpyDeviceToHost);
  const double waves_per_simd = (double)threads / 64 / 4;  // one block per CU
  const double inst = (double)iters * ch;
  printf("%-28s blocks %4d threads %4d: %7.1f clk/inst/wave, %6.1f clk/inst/SIMD, wall %8.1f us, eff clock %.2f GHz, %7.1f TFLOP/s\n",
         name, blocks, threads, h[0] / inst, h[0] / inst / (waves_per_simd < 1 ? 1 : waves_per_simd), ms * 1e3,
         h[0] / (ms * 1e-3) * 1e-9, inst * flop_per_inst * (threads / 64) * blocks / (ms * 1e-3) * 1e-12);
  hipFree(out); hipFree(cyc);
}

int main() {
  const int it = 20000;
  for (int threads : {256, 512}) {
    run("mfma_f64_16x16x4 x4 chains", [&](double *o, long long *c, int n) { hipLaunchKernelGGL(k_mfma16<4>, dim3(256), dim3(threads), 0, 0, o, c, n, 1.0, 1.0); }, 256, threads, it, 4, 2048.0);
    run("mfma_f64_16x16x4 x8 chains", [&](double *o, long long *c, int n) { hipLaunchKernelGGL(k_mfma16<8>, dim3(256), dim3(threads), 0, 0, o, c, n, 1.0, 1.0); }, 256, threads, it, 8, 2048.0);
    run("mfma_f64_4x4x4_4b x8 chains", [&](double *o, long long *c, int n) { hipLaunchKernelGGL(k_mfma4<8>, dim3(256), dim3(threads), 0, 0, o, c, n, 1.0, 1.0); }, 256, threads, it, 8, 512.0);
    run("v_fma_f64 x16 chains", [&](double *o, long long *c, int n) { hipLaunchKernelGGL(k_fma<16>, dim3(256), dim3(threads), 0, 0, o, c, n, 1.0000001, 1e-9); }, 256, threads, it, 16, 128.0);
  }
  run("mfma_f64_16x16x4 x4, 1 wave", [&](double *o, long long *c, int n) { hipLaunchKernelGGL(k_mfma16<4>, dim3(1), dim3(64), 0, 0, o, c, n, 1.0, 1.0); }, 1, 64, it, 4, 2048.0);
  run("mfma_f64_16x16x4 x1, 1 wave", [&](double *o, long long *c, int n) { hipLaunchKernelGGL(k_mfma16<1>, dim3(1), dim3(64), 0, 0, o, c, n, 1.0, 1.0); }, 1, 64, it, 1, 2048.0);
  run("v_fma_f64 x16, 1 wave", [&](double *o, long long *c, int n) { hipLaunchKernelGGL(k_fma<16>, dim3(1), dim3(64), 0, 0, o, c, n, 1.0000001, 1e-9); }, 1, 64, it, 16, 128.0);
  return 0;
}
