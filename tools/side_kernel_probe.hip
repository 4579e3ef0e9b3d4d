// How many 160 KB-LDS workgroups may a launch have before a one-workgroup kernel on ANOTHER stream no longer runs beside it?
// (ovp_build_gate_gram_tail, mode 4: chol(P) as k_chol2 on the side stream beside the feature kernel.)  Main kernel: N workgroups x 512
// threads, 160 KB LDS each (one per CU), spinning ~60 us; side kernel: 1 workgroup x 768 threads, 100 KB LDS, forked by an event in
// front of the main launch exactly as the library does, records its start / end on the device wall clock, and so does the main kernel.
// "beside" = the side kernel ended before the main kernel did AND the main launch took no longer than its spin + 15 us (a workgroup
// that had to wait for the side kernel's CU makes it 100 us).
// hipcc --offload-arch=gfx950 -O2 tools/side_kernel_probe.hip -o tools/ab/side_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512) void k_main(long long spin, long long* tmax, int early_exit0) {
  extern __shared__ double lds[];
  if (early_exit0 && blockIdx.x == 0) return;
  const long long t0 = wall_clock64();
  lds[threadIdx.x] = (double)t0;
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    atomicMax((unsigned long long*)tmax, (unsigned long long)wall_clock64());
    atomicMin((unsigned long long*)(tmax + 3), (unsigned long long)t0);
  }
}
__global__ __launch_bounds__(768) void k_side(long long spin, long long* tse) {
  extern __shared__ double lds[];
  const long long t0 = wall_clock64();
  lds[threadIdx.x] = (double)t0;
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { tse[0] = t0; tse[1] = wall_clock64(); }
}

int main() {
  hipStream_t s1, s2;
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t ef, ej;
  hipEventCreateWithFlags(&ef, hipEventDisableTiming);
  hipEventCreateWithFlags(&ej, hipEventDisableTiming);
  long long* d;
  hipMalloc(&d, 64);
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
  const long long spin_main = (long long)rate * 60 / 1000, spin_side = (long long)rate * 40 / 1000;  // 60 us, 40 us
  hipFuncSetAttribute((const void*)k_main, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_side, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int ee = 0; ee < 2; ++ee)
    for (int N : {247, 248, 249, 250, 251, 252, 253, 254, 255, 256}) {
      int beside = 0;
      double span_sum = 0.0;
      const int reps = 40;
      for (int r = 0; r < reps + 3; ++r) {
        hipMemsetAsync(d, 0, 24, s1);
        hipMemsetAsync(d + 3, 0xff, 8, s1);
        hipEventRecord(ef, s1);
        hipStreamWaitEvent(s2, ef, 0);
        hipLaunchKernelGGL(k_side, dim3(1), dim3(768), 100 * 1024, s2, spin_side, d + 1);
        hipEventRecord(ej, s2);
        hipLaunchKernelGGL(k_main, dim3(N), dim3(512), 160 * 1024 - 64, s1, spin_main, d, ee);
        hipStreamWaitEvent(s1, ej, 0);
        hipStreamSynchronize(s1);
        long long h[4];
        hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
        const double span_us = 1e3 * (double)(h[0] - h[3]) / (double)rate;
        if (r >= 3) span_sum += span_us;
        if (r >= 3 && h[2] < h[0] && span_us < 75.0) ++beside;
      }
      printf("main launch of %3d workgroups%s: side kernel beside it in %2d of %d runs, main launch %.1f us on average\n", N,
             ee ? " (workgroup 0 returns at once)" : "", beside, reps, span_sum / reps);
    }
  return 0;
}
