// Probe: block -> XCD placement, CU-masked streams, intra-XCD barrier latency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s at %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

__global__ void k_where(unsigned *out) {
  extern __shared__ double big[];
  if (threadIdx.x == 0) {
    big[0] = 1.0;
    out[2 * blockIdx.x] = xcc_id();
    out[2 * blockIdx.x + 1] = hw_id();
  }
}

// T workgroups barrier-synchronise `iters` times through one counter; every round each WG
// publishes a value (sc1 store) and reads all T values (sc1 loads).
__global__ void k_barrier(int T, int iters, unsigned *cnt, double *slots, unsigned long long *tout, unsigned *xcc,
                          double *sink) {
  const int me = blockIdx.x;
  if (threadIdx.x == 0) xcc[me] = xcc_id();
  unsigned long long t0 = wall_clock64();
  double acc = 0;
  for (int it = 1; it <= iters; it++) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(&slots[me], (double)(it + me), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it * T);
      { unsigned sp = 0; while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++sp < 2000000u) {} }
    }
    __syncthreads();
    if (threadIdx.x < T) acc += __hip_atomic_load(&slots[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  if (threadIdx.x == 0) tout[me] = wall_clock64() - t0;
  if (threadIdx.x < T) sink[me * 64 + threadIdx.x] = acc;
}

// same-XCD team: 256 workgroups launched (one per CU through a 150 KB LDS request); those on XCD
// `want` form a team (rank by arrival), the others exit.  Same publish + barrier + gather round.
__global__ void k_team(int want, int iters, unsigned *reg, unsigned *cnt, double *slots, unsigned long long *tout,
                       unsigned *tsize, double *sink, int payload) {
  extern __shared__ double big[];
  __shared__ int s_rank, s_T;
  const unsigned x = xcc_id();
  if (x != (unsigned)want) return;
  if (threadIdx.x == 0) {
    big[0] = 1.0;
    s_rank = (int)__hip_atomic_fetch_add(reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the team size is known once every workgroup of the launch has started: wait for 32 (expected)
    { unsigned sp = 0; while (__hip_atomic_load(reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u && ++sp < 2000000u) {} }
    s_T = 32;
  }
  __syncthreads();
  const int me = s_rank, T = s_T;
  unsigned long long t0 = wall_clock64();
  double acc = 0;
  for (int it = 1; it <= iters; it++) {
    // publish `payload` doubles per workgroup
    for (int k = threadIdx.x; k < payload; k += blockDim.x)
      __hip_atomic_store(&slots[me * payload + k], (double)(it + me + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(it * T);
      { unsigned sp = 0; while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++sp < 2000000u) {} }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < T * payload; k += blockDim.x)
      acc += __hip_atomic_load(&slots[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  if (threadIdx.x == 0) { tout[me] = wall_clock64() - t0; *tsize = T; }
  sink[me * 1024 + threadIdx.x] = acc;
}

// same-XCD team, L2-local synchronisation: counter RMWs at WORKGROUP scope (no sc bits: performed in
// this XCD's L2), payload by plain stores (write-through to the L2) and non-temporal loads (bypass L1,
// L2-served).  Only meaningful because every member verified the same XCC id.
__global__ void k_team_l2(int want, int iters, unsigned *reg, unsigned *cnt, double *slots, unsigned long long *tout,
                          double *sink, int payload, unsigned *bad) {
  extern __shared__ double big[];
  __shared__ int s_rank, s_fail;
  const unsigned x = xcc_id();
  if (x != (unsigned)want) return;
  if (threadIdx.x == 0) {
    big[0] = 1.0;
    s_fail = 0;
    s_rank = (int)__hip_atomic_fetch_add(reg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 32u && ++spins < 2000000u) {}
    if (spins >= 2000000u) s_fail = 1;
  }
  __syncthreads();
  const int me = s_rank, T = 32;
  unsigned long long t0 = wall_clock64();
  double acc = 0;
  unsigned nbad = 0;
  for (int it = 1; it <= iters; it++) {
    for (int k = threadIdx.x; k < payload; k += blockDim.x) slots[me * payload + k] = (double)(it * 1000 + me);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned target = (unsigned)(it * T);
      unsigned spins = 0;
      while (__builtin_nontemporal_load(cnt) < target && ++spins < 200000u) {}
      if (spins >= 200000u) s_fail = 1;
    }
    __syncthreads();
    if (s_fail) break;
    for (int k = threadIdx.x; k < T * payload; k += blockDim.x) {
      const double v = __builtin_nontemporal_load(&slots[k]);
      acc += v;
      if (v != (double)(it * 1000 + k / payload)) nbad++;
    }
    __syncthreads();
    // second barrier so that nobody overwrites slots while others still read (ping-pong would avoid it)
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned target = (unsigned)(it * T);
      unsigned spins = 0;
      while (__builtin_nontemporal_load(cnt + 1) < target && ++spins < 200000u) {}
      if (spins >= 200000u) s_fail = 1;
    }
    __syncthreads();
    if (s_fail) break;
  }
  if (threadIdx.x == 0) tout[me] = wall_clock64() - t0;
  if (s_fail && threadIdx.x == 0) atomicAdd(bad, 1000000u);
  if (nbad) atomicAdd(bad, nbad);
  sink[me * 1024 + threadIdx.x] = acc;
}

int main() {
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d\n", p.name, p.multiProcessorCount);
  unsigned *d_out;
  CK(hipMalloc(&d_out, 4096 * sizeof(unsigned)));
  std::vector<unsigned> h(4096);
  CK(hipFuncSetAttribute((const void *)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  // 1) plain launch, 256 blocks, 150 KB LDS each (one per CU)
  hipLaunchKernelGGL(k_where, dim3(256), dim3(64), 150 * 1024, 0, d_out);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), d_out, 512 * sizeof(unsigned), hipMemcpyDeviceToHost));
  printf("plain: block->xcc first 24:");
  for (int b = 0; b < 24; b++) printf(" %u", h[2 * b]);
  int per[8] = {0};
  for (int b = 0; b < 256; b++) per[h[2 * b] & 7]++;
  printf("\n  per-xcc counts:");
  for (int x = 0; x < 8; x++) printf(" %d", per[x]);
  printf("\n");
  // 2) CU-masked streams: try two mask layouts
  for (int layout = 0; layout < 2; layout++) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (layout == 0) mask[0] = 0xffffffffu;  // bits 0..31
    else for (int i = 0; i < 256; i += 8) mask[i / 32] |= 1u << (i % 32);  // every 8th bit
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
    if (e != hipSuccess) { printf("cumask layout %d: create failed: %s\n", layout, hipGetErrorString(e)); continue; }
    hipLaunchKernelGGL(k_where, dim3(64), dim3(64), 150 * 1024, st, d_out);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d_out, 128 * sizeof(unsigned), hipMemcpyDeviceToHost));
    int c[8] = {0};
    for (int b = 0; b < 64; b++) c[h[2 * b] & 7]++;
    printf("cumask layout %d: per-xcc counts of 64 blocks:", layout);
    for (int x = 0; x < 8; x++) printf(" %d", c[x]);
    printf("\n");
    // 3) barrier bench on this masked stream with T = 32 (only if all on one XCC)
    int onexcc = 0;
    for (int x = 0; x < 8; x++) onexcc += c[x] == 64;
    unsigned *cnt; double *slots, *sink; unsigned long long *tout; unsigned *xcc;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&slots, 64 * 8)); CK(hipMalloc(&sink, 64 * 64 * 8)); CK(hipMalloc(&tout, 64 * 8)); CK(hipMalloc(&xcc, 64 * 4));
    for (int T : {8, 16, 32}) {
      CK(hipMemset(cnt, 0, 4));
      const int iters = 2000;
      hipLaunchKernelGGL(k_barrier, dim3(T), dim3(64), 0, st, T, iters, cnt, slots, tout, xcc, sink);
      hipError_t se = hipStreamSynchronize(st);
      if (se != hipSuccess) { printf("barrier sync failed %s\n", hipGetErrorString(se)); break; }
      std::vector<unsigned long long> tt(64);
      std::vector<unsigned> xx(64);
      CK(hipMemcpy(tt.data(), tout, T * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(xx.data(), xcc, T * 4, hipMemcpyDeviceToHost));
      unsigned long long mx = *std::max_element(tt.begin(), tt.begin() + T);
      int cc[8] = {0};
      for (int b = 0; b < T; b++) cc[xx[b] & 7]++;
      printf("  layout %d T=%d: %.3f us per round (publish + barrier + gather); xcc spread:", layout, T, mx * 10.0 / 1000.0 / iters);
      for (int x = 0; x < 8; x++) printf(" %d", cc[x]);
      printf("\n");
    }
    CK(hipStreamDestroy(st));
  }
  // 4) barrier bench unmasked, T = 32 and 256 (cross-XCD)
  {
    unsigned *cnt; double *slots, *sink; unsigned long long *tout; unsigned *xcc;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&slots, 256 * 8)); CK(hipMalloc(&sink, 256 * 64 * 8)); CK(hipMalloc(&tout, 256 * 8)); CK(hipMalloc(&xcc, 256 * 4));
    for (int T : {8, 32, 64}) {
      CK(hipMemset(cnt, 0, 4));
      const int iters = 1000;
      hipLaunchKernelGGL(k_barrier, dim3(T), dim3(64), 0, 0, T, iters, cnt, slots, tout, xcc, sink);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> tt(256);
      CK(hipMemcpy(tt.data(), tout, T * 8, hipMemcpyDeviceToHost));
      unsigned long long mx = *std::max_element(tt.begin(), tt.begin() + T);
      printf("  unmasked T=%d (spread over XCDs): %.3f us per round\n", T, mx * 10.0 / 1000.0 / iters);
    }
  }
  // 5) same-XCD team of 32 workgroups
  {
    unsigned *reg, *cnt, *tsize; double *slots, *sink; unsigned long long *tout;
    CK(hipMalloc(&reg, 4)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&tsize, 4)); CK(hipMalloc(&slots, 32 * 64 * 8));
    CK(hipMalloc(&sink, 64 * 1024 * 8)); CK(hipMalloc(&tout, 64 * 8));
    CK(hipFuncSetAttribute((const void *)k_team, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int payload : {1, 16, 56}) {
      CK(hipMemset(reg, 0, 4)); CK(hipMemset(cnt, 0, 4));
      const int iters = 2000;
      hipLaunchKernelGGL(k_team, dim3(256), dim3(256), 150 * 1024, 0, 3, iters, reg, cnt, slots, tout, tsize, sink, payload);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> tt(64);
      CK(hipMemcpy(tt.data(), tout, 32 * 8, hipMemcpyDeviceToHost));
      unsigned long long mx = *std::max_element(tt.begin(), tt.begin() + 32);
      printf("  same-XCD team of 32, payload %d doubles/WG (gather %d): %.3f us per round\n", payload, 32 * payload,
             mx * 10.0 / 1000.0 / iters);
    }
  }
  // 6) same-XCD team with L2-local synchronisation (two barriers per round, payload verified)
  {
    unsigned *reg, *cnt, *bad; double *slots, *sink; unsigned long long *tout;
    CK(hipMalloc(&reg, 4)); CK(hipMalloc(&cnt, 8)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&slots, 32 * 64 * 8));
    CK(hipMalloc(&sink, 64 * 1024 * 8)); CK(hipMalloc(&tout, 64 * 8));
    CK(hipFuncSetAttribute((const void *)k_team_l2, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int payload : {1, 16, 56}) {
      CK(hipMemset(reg, 0, 4)); CK(hipMemset(cnt, 0, 8)); CK(hipMemset(bad, 0, 4));
      const int iters = 2000;
      hipLaunchKernelGGL(k_team_l2, dim3(256), dim3(256), 150 * 1024, 0, 5, iters, reg, cnt, slots, tout, sink, payload, bad);
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("k_team_l2 failed: %s\n", hipGetErrorString(e)); break; }
      std::vector<unsigned long long> tt(64);
      unsigned hb = 0;
      CK(hipMemcpy(tt.data(), tout, 32 * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      unsigned long long mx = *std::max_element(tt.begin(), tt.begin() + 32);
      printf("  same-XCD L2-local team, payload %d doubles/WG: %.3f us per round (2 barriers), stale reads %u\n", payload,
             mx * 10.0 / 1000.0 / iters, hb);
    }
  }
  return 0;
}
