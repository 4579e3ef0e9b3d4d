// Where do the workgroups of a launch and of a one-workgroup launch on another stream (forked in front of it) land?
// Prints, per repetition: XCC of the side workgroup, XCC of main block 0..9, workgroups of the main launch per XCC.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ int xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return (int)(x & 15);
}
__global__ __launch_bounds__(512) void k_main(long long spin, int* where) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) where[blockIdx.x] = xcc_id();
  const long long t0 = wall_clock64();
  lds[threadIdx.x] = (double)t0;
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
__global__ __launch_bounds__(768) void k_side(long long spin, int* where) {
  extern __shared__ double lds[];
  if (threadIdx.x == 0) where[0] = xcc_id();
  const long long t0 = wall_clock64();
  lds[threadIdx.x] = (double)t0;
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
int main() {
  hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t ef, ej;
  (void)hipEventCreateWithFlags(&ef, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&ej, hipEventDisableTiming);
  int* d;
  (void)hipMalloc(&d, 4 * 300);
  int rate = 0;
  (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  const long long spin = (long long)rate * 30 / 1000;
  (void)hipFuncSetAttribute((const void*)k_main, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute((const void*)k_side, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int N : {248, 250, 256})
    for (int r = 0; r < 6; ++r) {
      (void)hipMemsetAsync(d, 0xff, 4 * 300, s1);
      (void)hipEventRecord(ef, s1);
      (void)hipStreamWaitEvent(s2, ef, 0);
      hipLaunchKernelGGL(k_side, dim3(1), dim3(768), 100 * 1024, s2, spin, d + 299);
      (void)hipEventRecord(ej, s2);
      hipLaunchKernelGGL(k_main, dim3(N), dim3(512), 160 * 1024 - 64, s1, spin, d);
      (void)hipStreamWaitEvent(s1, ej, 0);
      (void)hipStreamSynchronize(s1);
      int h[300];
      (void)hipMemcpy(h, d, 4 * 300, hipMemcpyDeviceToHost);
      int cnt[8] = {0};
      for (int b = 0; b < N; ++b) cnt[h[b] & 7]++;
      printf("N=%3d: side on XCC %d | main blocks 0..9 on", N, h[299]);
      for (int b = 0; b < 10; ++b) printf(" %d", h[b]);
      printf(" | per XCC:");
      for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
      printf("\n");
    }
  return 0;
}
