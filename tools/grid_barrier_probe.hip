// What does a device-wide barrier inside ONE kernel cost against a kernel boundary, for phases that read what other workgroups
// (on other XCDs) wrote in the phase before?  P phases; in each, workgroup b sums `chunk` doubles of workgroup (37 b + phase) % B's
// region of the previous phase and writes its own region.
//   (a) P launches of one phase each          (b) one launch, P - 1 barriers (atomic counter, agent-scope release / acquire)
// hipcc --offload-arch=gfx950 -O2 tools/grid_barrier_probe.hip -o scratch/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void phase_body(const double* src, double* dst, int chunk, int B, int b, int phase) {
  const int from = (37 * b + phase) % B;
  double s = 0.0;
  for (int i = threadIdx.x; i < chunk; i += blockDim.x) s += src[(size_t)from * chunk + i];
  for (int i = threadIdx.x; i < chunk; i += blockDim.x) dst[(size_t)b * chunk + i] = s + 1.0 + i;
}
__global__ void k_phase(const double* src, double* dst, int chunk, int phase) { phase_body(src, dst, chunk, gridDim.x, blockIdx.x, phase); }

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);  // agent scope: this workgroup's stores are written back first
    while (__atomic_load_n(ctr, __ATOMIC_RELAXED) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // every wave: stale lines of L1 / L2 dropped
}
__global__ void k_fused(double* buf0, double* buf1, int chunk, int P, unsigned* ctr) {
  double* src = buf0;
  double* dst = buf1;
  for (int ph = 0; ph < P; ++ph) {
    phase_body(src, dst, chunk, gridDim.x, blockIdx.x, ph);
    if (ph + 1 < P) grid_barrier(ctr, (unsigned)(gridDim.x * (ph + 1)));
    double* t = src;
    src = dst;
    dst = t;
  }
}

int main() {
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int P = 5, reps = 200;
  for (int B : {128, 256, 512}) {
    for (int chunk : {128, 512, 2048}) {
      double *b0, *b1;
      unsigned* ctr;
      hipMalloc(&b0, sizeof(double) * B * chunk);
      hipMalloc(&b1, sizeof(double) * B * chunk);
      hipMalloc(&ctr, 4 * (reps + 8));
      hipMemset(b0, 0, sizeof(double) * B * chunk);
      hipMemset(ctr, 0, 4 * (reps + 8));
      std::vector<double> ha(B * chunk), hb(B * chunk);
      // (a)
      float ms_a = 0, ms_b = 0;
      for (int pass = 0; pass < 2; ++pass) {
        hipMemsetAsync(b0, 0, sizeof(double) * B * chunk, s);
        hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) {
          double *src = b0, *dst = b1;
          for (int ph = 0; ph < P; ++ph) {
            hipLaunchKernelGGL(k_phase, dim3(B), dim3(256), 0, s, src, dst, chunk, ph);
            std::swap(src, dst);
          }
        }
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        hipEventElapsedTime(&ms_a, e0, e1);
      }
      hipMemcpy(ha.data(), b1, sizeof(double) * B * chunk, hipMemcpyDeviceToHost);
      for (int pass = 0; pass < 2; ++pass) {
        hipMemsetAsync(b0, 0, sizeof(double) * B * chunk, s);
        hipMemsetAsync(ctr, 0, 4 * (reps + 8), s);
        hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_fused, dim3(B), dim3(256), 0, s, b0, b1, chunk, P, ctr + r);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        hipEventElapsedTime(&ms_b, e0, e1);
      }
      hipMemcpy(hb.data(), b1, sizeof(double) * B * chunk, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < B * chunk; ++i) bad += ha[i] != hb[i];
      printf("B %3d  chunk %4d (%5.0f KB per phase): %d launches %.2f us per phase | fused %.2f us per phase | mismatches %d\n", B, chunk,
             B * chunk * 8 / 1024.0, P, 1e3 * ms_a / (reps * P), 1e3 * ms_b / (reps * P), bad);
      hipFree(b0);
      hipFree(b1);
      hipFree(ctr);
    }
  }
  return 0;
}
