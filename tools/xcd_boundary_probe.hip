// Does data written by a kernel on XCD x arrive faster in the NEXT kernel when its reader also sits on XCD x?
// Producer: 32 blocks, all on XCD `px` (blocks b with b % 8 == px of an 8x larger grid; the others exit), write 256 KB.
// Consumer (next kernel, same stream): one 1024-thread block on XCD `cx` reads the 256 KB (16-byte loads, all in flight) and reports
// wall-clock nanoseconds.  hipcc --offload-arch=gfx950 -O2 tools/xcd_boundary_probe.hip -o scratch/xcd_boundary
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ void k_prod(double2* buf, int n2, int px, double v) {
  if ((int)(blockIdx.x & 7) != px) return;
  const int b = blockIdx.x >> 3, nb = gridDim.x >> 3;
  for (int i = b * blockDim.x + threadIdx.x; i < n2; i += nb * blockDim.x) buf[i] = double2{v + i, v - i};
}
__global__ void k_cons(const double2* buf, int n2, int cx, long long* out, double* sink, int* where) {
  if ((int)(blockIdx.x & 7) != cx) return;
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  __syncthreads();
  const long long t0 = wall_clock64();
  double s = 0.0;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) {
    const double2 v = buf[i];
    s += v.x + v.y;
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  sink[threadIdx.x] = s;
  if (threadIdx.x == 0) out[0] = t1 - t0, where[0] = (int)(x & 15);
}

int main() {
  const int n2 = 16384;  // 256 KB
  double2* buf;
  long long* dout;
  double* sink;
  int* dw;
  hipMalloc(&buf, n2 * sizeof(double2));
  hipMalloc(&dout, 8);
  hipMalloc(&sink, 1024 * 8);
  hipMalloc(&dw, 4);
  hipStream_t s;
  hipStreamCreate(&s);
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", rate);
  for (int cx = 0; cx < 8; ++cx) {
    std::vector<double> ts;
    int wh = -1;
    for (int rep = 0; rep < 30; ++rep) {
      hipLaunchKernelGGL(k_prod, dim3(32 * 8), dim3(256), 0, s, buf, n2, 0, (double)rep);
      hipLaunchKernelGGL(k_cons, dim3(8), dim3(1024), 0, s, buf, n2, cx, dout, sink, dw);
      hipStreamSynchronize(s);
      long long t;
      hipMemcpy(&t, dout, 8, hipMemcpyDeviceToHost);
      hipMemcpy(&wh, dw, 4, hipMemcpyDeviceToHost);
      if (rep >= 5) ts.push_back(1e6 * (double)t / (double)rate);
    }
    std::sort(ts.begin(), ts.end());
    printf("producer on XCC 0 -> consumer block %d (XCC %d): read of 256 KB median %.0f ns (min %.0f)\n", cx, wh, ts[ts.size() / 2], ts[0]);
  }
  return 0;
}
