#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  double a = a0, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters) {
  double* out;
  (void)hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999, 1e-3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999, 1e-3);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 256 * iters * NACC * 2;
  printf("VALU fma f64 NACC=%d blocks=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, ms, fl / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}
int main() {
  run<8>(1024, 4000);
  run<16>(1024, 2000);
  run<16>(2048, 2000);
  run<16>(4096, 2000);
  return 0;
}
