// Probe: one-way latency of a tagged-word hand-off between two workgroups (ping-pong / 2),
// across XCDs and inside one XCD, for the memory-scope idioms the cooperative solver can use.
//   mode 0: agent-scope relaxed atomic store + agent-scope relaxed atomic load  (sc1)
//   mode 1: plain store + non-temporal load  (served by the XCD's L2; only valid inside one XCD)
//   mode 2: agent-scope store + non-temporal load
//   mode 3: plain store (write-through to the XCD's L2) + sc0 load (past the CU's L1, from the XCD's L2): inside one XCD only
//   mode 4: L2 atomic add without sc1 (executed in the XCD's L2) + sc0 load: a barrier counter inside one XCD
// hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int MODE>
__device__ __forceinline__ void put(unsigned long long *p, unsigned long long v) {
  if (MODE == 3) { asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory"); return; }
  if (MODE == 4) { unsigned long long one = 1; asm volatile("global_atomic_add_x2 %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(one) : "memory"); return; }
  if (MODE == 1) *(volatile unsigned long long *)p = v;
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__device__ __forceinline__ unsigned long long get(unsigned long long *p) {
  if (MODE == 3 || MODE == 4) { unsigned long long v; asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_nontemporal_load(p);
}
// blocks A and B play; everybody else exits.  slots: [0] A->B, [16] B->A (separate lines)
template <int MODE>
__global__ void k_pp(int A, int B, int rounds, unsigned long long *slots, unsigned long long *out, unsigned *xcc) {
  const int me = blockIdx.x;
  if (threadIdx.x != 0 || (me != A && me != B)) return;
  xcc[me == A ? 0 : 1] = xcc_id();
  unsigned long long *mine = slots + (me == A ? 0 : 16), *theirs = slots + (me == A ? 16 : 0);
  unsigned long long t0 = wall_clock64();
  unsigned bad = 0;
  for (int r = 1; r <= rounds; r++) {
    if (me == A) put<MODE>(mine, (unsigned long long)r);
    unsigned sp = 0;
    while (get<MODE>(theirs) < (unsigned long long)r && ++sp < 4000000u) {}
    if (sp >= 4000000u) { bad = 1; break; }
    if (me == B) put<MODE>(mine, (unsigned long long)r);
  }
  out[me == A ? 0 : 1] = wall_clock64() - t0;
  out[2 + (me == A ? 0 : 1)] = bad;
}
template <int MODE>
int run(const char *name, int A, int B, unsigned long long *slots, unsigned long long *out, unsigned *xcc) {
  const int rounds = 20000;
  CK(hipMemset(slots, 0, 64 * 8));
  CK(hipMemset(out, 0, 4 * 8));
  hipLaunchKernelGGL(k_pp<MODE>, dim3(64), dim3(64), 0, 0, A, B, rounds, slots, out, xcc);
  CK(hipDeviceSynchronize());
  unsigned long long h[4]; unsigned x[2];
  CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
  CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
  printf("%-34s blocks %2d,%2d xcc %u,%u : one-way %.0f ns %s\n", name, A, B, x[0], x[1],
         (double)h[0] * 10.0 / rounds / 2.0, (h[2] | h[3]) ? "(TIMED OUT)" : "");
  return 0;
}
// load latency: dependent chain of loads over a small buffer (pointer chase), one lane
template <int MODE>
__global__ void k_chase(unsigned long long *buf, int steps, unsigned long long *out) {
  if (threadIdx.x != 0) return;
  unsigned long long idx = 0;
  unsigned long long t0 = wall_clock64();
  for (int s = 0; s < steps; s++) idx = get<MODE>(buf + idx);
  out[0] = wall_clock64() - t0;
  out[1] = idx;
}
int main() {
  unsigned long long *slots, *out, *buf; unsigned *xcc;
  CK(hipMalloc(&slots, 64 * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&xcc, 8));
  run<0>("agent store / agent load", 0, 1, slots, out, xcc);
  run<0>("agent store / agent load", 0, 8, slots, out, xcc);
  run<2>("agent store / nt load", 0, 1, slots, out, xcc);
  run<2>("agent store / nt load", 0, 8, slots, out, xcc);
  run<3>("plain store / sc0 load", 0, 8, slots, out, xcc);
  run<3>("plain store / sc0 load", 0, 16, slots, out, xcc);
  run<3>("plain store / sc0 load (cross!)", 0, 1, slots, out, xcc);
  run<4>("L2 atomic add / sc0 load", 0, 8, slots, out, xcc);
  run<4>("L2 atomic add / sc0 load (cross!)", 0, 1, slots, out, xcc);
  run<1>("plain store / nt load", 0, 8, slots, out, xcc);
  run<1>("plain store / nt load (cross!)", 0, 1, slots, out, xcc);
  // pointer chase over 64 lines (stride 128 B): agent loads vs nt loads
  const int L = 64;
  unsigned long long h[L * 16] = {0};
  for (int i = 0; i < L; i++) h[i * 16] = (unsigned long long)(((i + 17) % L) * 16);
  CK(hipMalloc(&buf, sizeof(h)));
  CK(hipMemcpy(buf, h, sizeof(h), hipMemcpyHostToDevice));
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      if (mode == 0) hipLaunchKernelGGL(k_chase<0>, dim3(1), dim3(64), 0, 0, buf, 20000, out);
      else hipLaunchKernelGGL(k_chase<1>, dim3(1), dim3(64), 0, 0, buf, 20000, out);
      CK(hipDeviceSynchronize());
    }
    unsigned long long r[2];
    CK(hipMemcpy(r, out, 16, hipMemcpyDeviceToHost));
    printf("dependent %s load: %.0f ns\n", mode == 0 ? "agent-scope (sc1)" : "non-temporal", (double)r[0] * 10.0 / 20000);
  }
  return 0;
}
