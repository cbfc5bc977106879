// Probe: what an all-gather round of the cooperative solver costs by the FORMAT of the tagged entries (VERDICT r4 item 4:
// "16-byte entries carry 8 bytes of payload; pack ... so a round's gather is 12-15 KB per CU instead of 24").
// T workgroups (one per CU, all resident), each owns R rows of a vector of N = T R doubles; a round: the owners publish
// their R values, every workgroup polls the whole vector into LDS, barrier.  Two buffers by round parity.
//   format 0: one value per 16 bytes {lo, tag, hi, tag}                      (k_coop today: 16 N bytes polled per CU)
//   format 1: three values per 32 bytes, two 16-byte units of 12 bytes payload + a 4-byte tag each
//             {d0.lo, d0.hi, d1.lo, tag} {d1.hi, d2.lo, d2.hi, tag}           (10.7 N bytes; needs 16-byte store atomicity)
//   format 2: values untagged (8 N bytes) behind ONE tagged word per workgroup, written after the values have been
//             acknowledged (s_waitcnt vmcnt(0)): the reader polls T flags, then reads the values -- two dependent trips
//   format 3: one value per 12 bytes {lo, hi, tag}, five per 64-byte line (12.8 N bytes): every owner stores its own entry
//             (no packing across lanes); needs a 12-byte store / load inside one line to be one transaction
// usage: allgather [T] [R] [rounds] [nap]      hipcc --offload-arch=gfx950 -O3 allgather.hip -o allgather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(void *p, u4 w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory"); }
__device__ __forceinline__ void st8(void *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u4 ld16(const void *p) { u4 w; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory"); return w; }
__device__ __forceinline__ void ld16x4(const void *p0, const void *p1, const void *p2, const void *p3, u4 &a, u4 &b, u4 &c, u4 &d) {
  asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
               "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
typedef unsigned u3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void st12(void *p, u3 w) { asm volatile("global_store_dwordx3 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory"); }
__device__ __forceinline__ void ld12x4(const void *p0, const void *p1, const void *p2, const void *p3, u3 &a, u3 &b, u3 &c, u3 &d) {
  asm volatile("global_load_dwordx3 %0, %4, off sc1\n\tglobal_load_dwordx3 %1, %5, off sc1\n\tglobal_load_dwordx3 %2, %6, off sc1\n\t"
               "global_load_dwordx3 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}
__device__ __forceinline__ size_t off12(int i) { return 64 * (size_t)(i / 5) + 12 * (size_t)(i % 5); }
__device__ __forceinline__ double mk(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
constexpr int B = 512, NMAX = 4096;

template <int FMT>
__global__ __launch_bounds__(B) void k_ag(int T, int R, int rounds, int nap, unsigned char *buf0, unsigned char *buf1, unsigned *flag0,
                                          unsigned *flag1, unsigned long long *reg, unsigned long long *out, double *sums) {
  __shared__ double vec[NMAX];
  __shared__ int bad;
  const int t = threadIdx.x, b = blockIdx.x, N = T * R;
  if (t == 0) {
    bad = 0;
    atomicAdd(reg, 1ull);
    while (__hip_atomic_load(reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)T) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  double acc = 0.0;
  for (int r = 1; r <= rounds && !bad; r++) {
    const unsigned tag = (unsigned)r;
    unsigned char *buf = (r & 1) ? buf1 : buf0;
    unsigned *flag = (r & 1) ? flag1 : flag0;
    // ---- publish the R owned values: value (row i, round r) = i + r / 1024
    if (FMT == 0) {
      if (t < R) {
        const int i = b * R + t;
        const double v = (double)i + (double)r / 1024.0;
        st16(buf + 16 * (size_t)i, u4{(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag});
      }
    } else if (FMT == 1) {
      const int G = (R + 2) / 3;  // groups of three per workgroup (the last one padded)
      if (t < G) {
        double v[3];
        for (int k = 0; k < 3; k++) { const int i = b * R + 3 * t + k; v[k] = (3 * t + k < R) ? (double)i + (double)r / 1024.0 : 0.0; }
        unsigned char *p = buf + 32 * (size_t)(b * G + t);
        st16(p, u4{(unsigned)__double2loint(v[0]), (unsigned)__double2hiint(v[0]), (unsigned)__double2loint(v[1]), tag});
        st16(p + 16, u4{(unsigned)__double2hiint(v[1]), (unsigned)__double2loint(v[2]), (unsigned)__double2hiint(v[2]), tag});
      }
    } else if (FMT == 3) {
      if (t < R) {
        const int i = b * R + t;
        const double v = (double)i + (double)r / 1024.0;
        st12(buf + off12(i), u3{(unsigned)__double2loint(v), (unsigned)__double2hiint(v), tag});
      }
    } else {
      if (t < R) {
        const int i = b * R + t;
        st8(buf + 8 * (size_t)i, (double)i + (double)r / 1024.0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) { asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(flag + 16 * b), "v"(tag) : "memory"); }
    }
    for (int k = 0; k < nap; k++) __builtin_amdgcn_s_sleep(1);
    // ---- gather
    unsigned spins = 0;
    if (FMT == 0) {
      for (int base = 0; base < N; base += 4 * B) {
        int c[4]; bool have[4]; const void *p[4];
        for (int k = 0; k < 4; k++) { c[k] = base + t + k * B; have[k] = c[k] >= N; p[k] = buf + 16 * (size_t)(have[k] ? 0 : c[k]); }
        for (;;) {
          u4 w[4];
          ld16x4(p[0], p[1], p[2], p[3], w[0], w[1], w[2], w[3]);
          bool all = true;
          for (int k = 0; k < 4; k++)
            if (!have[k]) { if (w[k].y == tag && w[k].w == tag) { vec[c[k]] = mk(w[k].x, w[k].z); have[k] = true; } else all = false; }
          if (all) break;
          if (++spins > 2000000u) { bad = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    } else if (FMT == 1) {
      const int G = (R + 2) / 3, NG = T * G;
      for (int base = 0; base < NG; base += 2 * B) {
        int c[2]; bool have[2]; const unsigned char *p[2];
        for (int k = 0; k < 2; k++) { c[k] = base + t + k * B; have[k] = c[k] >= NG; p[k] = buf + 32 * (size_t)(have[k] ? 0 : c[k]); }
        for (;;) {
          u4 w[4];
          ld16x4(p[0], p[0] + 16, p[1], p[1] + 16, w[0], w[1], w[2], w[3]);
          bool all = true;
          for (int k = 0; k < 2; k++)
            if (!have[k]) {
              const u4 a = w[2 * k], bb = w[2 * k + 1];
              if (a.w == tag && bb.w == tag) {
                const int wg = c[k] / G, g = c[k] % G, i = wg * R + 3 * g;
                vec[i] = mk(a.x, a.y);
                if (3 * g + 1 < R) vec[i + 1] = mk(a.z, bb.x);
                if (3 * g + 2 < R) vec[i + 2] = mk(bb.y, bb.z);
                have[k] = true;
              } else all = false;
            }
          if (all) break;
          if (++spins > 2000000u) { bad = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    } else if (FMT == 3) {
      for (int base = 0; base < N; base += 4 * B) {
        int c[4]; bool have[4]; const void *p[4];
        for (int k = 0; k < 4; k++) { c[k] = base + t + k * B; have[k] = c[k] >= N; p[k] = buf + off12(have[k] ? 0 : c[k]); }
        for (;;) {
          u3 w[4];
          ld12x4(p[0], p[1], p[2], p[3], w[0], w[1], w[2], w[3]);
          bool all = true;
          for (int k = 0; k < 4; k++)
            if (!have[k]) { if (w[k].z == tag) { vec[c[k]] = mk(w[k].x, w[k].y); have[k] = true; } else all = false; }
          if (all) break;
          if (++spins > 2000000u) { bad = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    } else {
      if (t < T) {
        for (;;) {
          unsigned f;
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(f) : "v"(flag + 16 * t) : "memory");
          if (f == tag) break;
          if (++spins > 2000000u) { bad = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      for (int c = 2 * t; c < N; c += 2 * B) {
        const u4 w = ld16(buf + 8 * (size_t)c);
        vec[c] = mk(w.x, w.y);
        if (c + 1 < N) vec[c + 1] = mk(w.z, w.w);
      }
    }
    __syncthreads();
    // (something that reads the vector: a checksum over this thread's share)
    for (int c = t; c < N; c += B) acc += vec[c] - ((double)c + (double)r / 1024.0);
    __syncthreads();
  }
  if (t == 0) { out[2 * b] = wall_clock64() - t0; out[2 * b + 1] = bad; }
  atomicAdd(sums + b, acc);  // 0 when every value arrived as published
}

template <int FMT>
int run(const char *name, int T, int R, int rounds, int nap) {
  unsigned char *b0, *b1; unsigned *f0, *f1; unsigned long long *reg, *out; double *sums;
  const size_t bytes = (size_t)NMAX * 32;
  CK(hipMalloc(&b0, bytes)); CK(hipMalloc(&b1, bytes)); CK(hipMalloc(&f0, 256 * 64)); CK(hipMalloc(&f1, 256 * 64));
  CK(hipMalloc(&reg, 64)); CK(hipMalloc(&out, 256 * 16)); CK(hipMalloc(&sums, 256 * 8));
  CK(hipMemset(b0, 0, bytes)); CK(hipMemset(b1, 0, bytes)); CK(hipMemset(f0, 0, 256 * 64)); CK(hipMemset(f1, 0, 256 * 64));
  double best = 1e30;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemset(b0, 0, bytes)); CK(hipMemset(b1, 0, bytes)); CK(hipMemset(f0, 0, 256 * 64)); CK(hipMemset(f1, 0, 256 * 64));
    CK(hipMemset(reg, 0, 64)); CK(hipMemset(out, 0, 256 * 16)); CK(hipMemset(sums, 0, 256 * 8));
    hipLaunchKernelGGL(k_ag<FMT>, dim3(T), dim3(B), 0, 0, T, R, rounds, nap, b0, b1, f0, f1, reg, out, sums);
    CK(hipDeviceSynchronize());
    unsigned long long h[512]; double s[256];
    CK(hipMemcpy(h, out, sizeof(unsigned long long) * 2 * T, hipMemcpyDeviceToHost));
    CK(hipMemcpy(s, sums, sizeof(double) * T, hipMemcpyDeviceToHost));
    unsigned long long mx = 0; int bad = 0; double err = 0;
    for (int b = 0; b < T; b++) { if (h[2 * b] > mx) mx = h[2 * b]; bad |= (int)h[2 * b + 1]; err += s[b] < 0 ? -s[b] : s[b]; }
    const double ns = (double)mx * 10.0 / rounds;
    if (bad || err != 0.0) { printf("%-44s T %d R %d: %s (checksum %.3g)\n", name, T, R, bad ? "TIMED OUT" : "WRONG VALUES", err); return 0; }
    if (ns < best) best = ns;
  }
  printf("%-44s T %3d R %2d nap %2d: %7.0f ns per round\n", name, T, R, nap, best);
  hipFree(b0); hipFree(b1); hipFree(f0); hipFree(f1); hipFree(reg); hipFree(out); hipFree(sums);
  return 0;
}

int main(int argc, char **argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 219, R = argc > 2 ? atoi(argv[2]) : 8, rounds = argc > 3 ? atoi(argv[3]) : 4000;
  const int nap = argc > 4 ? atoi(argv[4]) : 0;
  if (T > 256 || T * R > NMAX) { printf("T <= 256, T R <= %d\n", NMAX); return 1; }
  run<0>("16 B per value {lo, tag, hi, tag}", T, R, rounds, nap);
  run<1>("32 B per three values (12 B + tag per unit)", T, R, rounds, nap);
  run<2>("8 B per value behind one flag per workgroup", T, R, rounds, nap);
  run<3>("12 B per value {lo, hi, tag}, 5 per line", T, R, rounds, nap);
  return 0;
}
