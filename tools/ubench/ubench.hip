// Micro-benchmarks that calibrate the cost model of the decode kernel's phases on gfx950 (one workgroup per CU):
// barrier cost vs waves, dependent LDS chains, LDS atomics, DPP scans.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define DPP(old, v, ctrl, rm) __builtin_amdgcn_update_dpp((int)(old), (int)(v), (ctrl), (rm), 0xf, false)

__device__ inline void light_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE>
__global__ void k(long long *out, int iters, int *sink) {
  __shared__ int lds[4096];
  __shared__ int bins[1024];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 7 + 1) & 4095;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) bins[i] = 0;
  __syncthreads();
  int v = threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) light_barrier();
    if (MODE == 1) __syncthreads();
    if (MODE == 2) { v = lds[v & 4095]; }                       // dependent LDS chain, all waves
    if (MODE == 3) { if (threadIdx.x < 64) v = lds[v & 4095]; } // dependent LDS chain, one wave
    if (MODE == 4) { atomicAdd(&bins[(v * 2654435761u >> 22) & 1023], 1); v += 17; }  // spread LDS atomics
    if (MODE == 5) { atomicAdd(&bins[v & 3], 1); v += 1; }      // contended LDS atomics
    if (MODE == 6) {                                            // DPP inclusive scan
      v += DPP(0, v, 0x111, 0xf); v += DPP(0, v, 0x112, 0xf); v += DPP(0, v, 0x114, 0xf); v += DPP(0, v, 0x118, 0xf);
      v += DPP(0, v, 0x142, 0xa); v += DPP(0, v, 0x143, 0xc);
    }
    if (MODE == 7) { v = __shfl_xor(v, 1 + (it & 31), 64); }     // bpermute shuffle chain
    if (MODE == 8) { v = lds[v & 4095]; light_barrier(); }      // one dependent read + barrier per "phase"
    if (MODE == 9) { if (threadIdx.x < 128) { v = lds[v & 4095]; v = lds[v & 4095]; v = lds[v & 4095]; } light_barrier(); }  // K-thread phase
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (v == 0x7fffffff) sink[0] = v + bins[3];
}

int main() {
  long long *out; int *sink;
  CK(hipMalloc(&out, 256 * 8)); CK(hipMalloc(&sink, 64));
  const char *names[] = {"light barrier", "__syncthreads", "dependent LDS read (all waves)", "dependent LDS read (1 wave active)", "LDS atomic spread",
                         "LDS atomic 4 addresses", "DPP scan (6 steps)", "bpermute shuffle", "read+barrier phase", "3 reads (128 thr) + barrier phase"};
  const int iters = 2000;
  for (int threads : {256, 512, 1024}) {
    for (int mode = 0; mode < 10; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (mode) {
#define L(M) case M: hipLaunchKernelGGL(k<M>, dim3(256), dim3(threads), 0, 0, out, iters, sink); break;
          L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9)
        }
        CK(hipDeviceSynchronize());
      }
      std::vector<long long> h(256);
      CK(hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost));
      double s = 0; for (auto x : h) s += x;
      printf("threads %4d  %-36s %8.1f clock64-ticks/iter\n", threads, names[mode], s / 256 / iters);
    }
  }
  // clock64 rate
  return 0;
}
