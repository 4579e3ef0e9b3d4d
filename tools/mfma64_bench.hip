#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters) {
  double* out;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double mf = (double)blocks * 4 * iters * NACC;
  double tf = mf * 2048 / (ms * 1e-3) / 1e12;
  // cycles per MFMA per SIMD assuming waves spread evenly over 1024 SIMDs at 2.4 GHz
  double waves_per_simd = blocks * 4 / 1024.0;
  if (waves_per_simd < 1) waves_per_simd = 1;
  double cyc = ms * 1e-3 * 2.4e9 / (iters * NACC * waves_per_simd);
  printf("NACC=%d blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s  ~%.1f cycles/MFMA/SIMD\n", NACC, blocks, iters, ms, tf, cyc);
  hipFree(out);
}
int main() {
  run<1>(256, 4000);
  run<4>(256, 2000);
  run<16>(256, 1000);
  run<4>(1024, 2000);
  run<16>(1024, 500);
  run<4>(64, 2000);
  return 0;
}
