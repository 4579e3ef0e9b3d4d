// What one CU's f64 pipe does with v_mfma_f64_16x16x4 (2048 FLOP) and v_mfma_f64_4x4x4 (4 blocks, 512 FLOP): ONE workgroup of 1 .. 16 waves,
// every wave the same chain of MFMAs; per wave the shader cycles per instruction (wave 0 and the slowest wave), and the kernel's wall
// time by HIP events.  Reading (profiles/r06_mfma64_issue_bench.txt): 64 cycles per 16x16x4 on a SIMD whether the accumulators depend on
// each other or not (1 or 4 chains: the same), = 32 FLOP per cycle and SIMD, the part's nominal f64 matrix rate and the same as the
// f64 FMA; waves that share a SIMD are served strictly oldest first (wave 0 keeps 64 cycles, the youngest of three sees 192).
// The "113 cycles" of rounds 3-5 came from a chip-wide run (tools/mfma64_bench.hip, 256+ workgroups) whose clock is power-limited:
// it does not apply to a kernel on one or two CUs.
// hipcc --offload-arch=gfx950 -O3 -w -std=c++17 tools/mfma64_issue_bench.hip -o /tmp/mfma64_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int WHICH>
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, int iters, double a0, double b0) {
  double a = a0 + threadIdx.x, b = b0;
  d4 acc[NACC];
  double s1[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = d4{0, 0, 0, 0}; s1[i] = 0.0; }
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if constexpr (WHICH == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      else if constexpr (WHICH == 1) s1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1[i], 0, 0, 0);
      else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(s1[i]) : "v"(a), "v"(b));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + s1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 12 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NACC, int WHICH>
void run(int threads, int iters, const char* name, double flop) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * 1024);
  hipMalloc(&cyc, sizeof(long long) * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, WHICH>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, WHICH>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const int waves = threads / 64;
  const double n_inst = (double)iters * NACC;
  long long mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  printf("%-18s acc %2d waves %2d: %7.1f cycles per instruction (wave 0), slowest wave %7.1f; kernel %.1f us = %.1f ns per instruction\n", name, NACC, waves,
         h[0] / n_inst, mx / n_inst, ms * 1e3, ms * 1e6 / n_inst);
}
int main() {
  for (int threads : {64, 256, 512, 768, 1024}) {
    run<1, 0>(threads, 8000, "mfma_f64_16x16x4", 2048);
    run<4, 0>(threads, 2000, "mfma_f64_16x16x4", 2048);
    run<2, 1>(threads, 4000, "mfma_f64_4x4x4", 512);
  }
  return 0;
}
