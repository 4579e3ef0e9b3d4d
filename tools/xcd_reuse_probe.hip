// What does ONE workgroup get out of its XCD's L2 across kernel boundaries?  (The plane loop's k_chol2 reads 240 KB of T_try and
// 230 KB of L0 per plane at ~16 B/clk; tools/xcd_boundary_probe.hip showed 58 B/clk for lines the same XCD wrote one kernel earlier.)
// Consumer: one 768-thread block ON XCC cx (eight candidates are launched, the one whose HW_REG_XCC_ID matches reads, the others exit)
// reads 240 KB with 16-byte loads and reports nanoseconds.  Producer: 256 blocks, the ones on XCC px write the buffer (work by ticket).
//   A  written on XCC 0, read on XCC 0 in the next kernel          B  written on XCC 0, read on XCC 3
//   C  as A with N filler kernels (8 MB streamed by all XCDs) in between: does the written copy stay?
//   D  written on XCC 3, read on XCC 0 twice in consecutive kernels: does the second read find the (clean) lines in XCC 0's L2?
//   E  as D with N filler kernels between the two reads
// hipcc --offload-arch=gfx950 -O2 tools/xcd_reuse_probe.hip -o /tmp/xcd_reuse && /tmp/xcd_reuse
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ int xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return (int)(x & 15);
}
__global__ void k_prod(double2* buf, int n2, int px, double v, int* ticket) {
  if (xcc_id() != px) return;
  __shared__ int t;
  if (threadIdx.x == 0) t = atomicAdd(ticket, 1);
  __syncthreads();
  const int chunk = 1024;  // double2 per ticket
  for (int base = t * chunk; base < n2; base += 32 * chunk)
    for (int i = base + threadIdx.x; i < base + chunk && i < n2; i += blockDim.x) buf[i] = double2{v + i, v - i};
}
__global__ void k_cons(const double2* buf, int n2, int cx, long long* out, double* sink, int* claim) {
  if (xcc_id() != cx) return;
  __shared__ int first;
  if (threadIdx.x == 0) first = atomicAdd(claim, 1);
  __syncthreads();
  if (first != 0) return;
  const long long t0 = wall_clock64();
  double s = 0.0;
  for (int i = threadIdx.x; i < n2; i += blockDim.x) {
    const double2 v = buf[i];
    s += v.x + v.y;
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  sink[threadIdx.x] = s;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_fill(const double2* src, double2* dst, int n2) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
  const int n2 = 15360;  // 240 KB
  const int nf2 = 256 * 1024;  // 4 MB read + 4 MB written per filler kernel
  double2 *buf, *fa, *fb;
  long long* dout;
  double* sink;
  int* words;
  hipMalloc(&buf, n2 * sizeof(double2));
  hipMalloc(&fa, nf2 * sizeof(double2));
  hipMalloc(&fb, nf2 * sizeof(double2));
  hipMemset(fa, 0, nf2 * sizeof(double2));
  hipMalloc(&dout, 8);
  hipMalloc(&sink, 1024 * 8);
  hipMalloc(&words, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  auto prod = [&](int px, double v) {
    hipMemsetAsync(words, 0, 64, s);
    hipLaunchKernelGGL(k_prod, dim3(256), dim3(256), 0, s, buf, n2, px, v, words);
  };
  auto cons = [&](int cx) {
    hipMemsetAsync(words + 4, 0, 4, s);
    hipLaunchKernelGGL(k_cons, dim3(64), dim3(768), 0, s, buf, n2, cx, dout, sink, words + 4);
  };
  auto fill = [&](int n) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_fill, dim3(512), dim3(256), 0, s, fa, fb, nf2);
  };
  auto read_ns = [&]() {
    hipStreamSynchronize(s);
    long long t;
    hipMemcpy(&t, dout, 8, hipMemcpyDeviceToHost);
    return 1e6 * (double)t / (double)rate;
  };
  auto report = [&](const char* name, std::vector<double>& ts) {
    std::sort(ts.begin(), ts.end());
    printf("%-64s median %6.0f ns  (min %6.0f)  = %5.1f B/clk at 2.1 GHz\n", name, ts[ts.size() / 2], ts[0],
           240.0 * 1024.0 / (ts[ts.size() / 2] * 2.1));
  };
  const int reps = 25;
  {
    std::vector<double> ts;
    for (int r = 0; r < reps; ++r) { prod(0, r); cons(0); double t = read_ns(); if (r >= 5) ts.push_back(t); }
    report("A  written on XCC 0 -> read on XCC 0, next kernel", ts);
  }
  {
    std::vector<double> ts;
    for (int r = 0; r < reps; ++r) { prod(0, r); cons(3); double t = read_ns(); if (r >= 5) ts.push_back(t); }
    report("B  written on XCC 0 -> read on XCC 3, next kernel", ts);
  }
  for (int nfill : {1, 4, 16}) {
    std::vector<double> ts;
    for (int r = 0; r < reps; ++r) { prod(0, r); fill(nfill); cons(0); double t = read_ns(); if (r >= 5) ts.push_back(t); }
    char nm[96];
    snprintf(nm, sizeof nm, "C  written on XCC 0, %2d filler kernels (8 MB each), read on XCC 0", nfill);
    report(nm, ts);
  }
  {
    std::vector<double> t1, t2;
    for (int r = 0; r < reps; ++r) {
      prod(3, r); cons(0); double a = read_ns(); cons(0); double b = read_ns();
      if (r >= 5) { t1.push_back(a); t2.push_back(b); }
    }
    report("D1 written on XCC 3 -> first read on XCC 0", t1);
    report("D2 ... second read on XCC 0, next kernel (clean lines)", t2);
  }
  for (int nfill : {1, 4, 16}) {
    std::vector<double> ts;
    for (int r = 0; r < reps; ++r) { prod(3, r); cons(0); read_ns(); fill(nfill); cons(0); double t = read_ns(); if (r >= 5) ts.push_back(t); }
    char nm[96];
    snprintf(nm, sizeof nm, "E  second read on XCC 0 behind %2d filler kernels", nfill);
    report(nm, ts);
  }
  {
    // host-uploaded data (H2D copy: what ovp_cov_upload leaves) read by one block
    std::vector<double> ts;
    std::vector<double2> h(n2);
    for (int r = 0; r < reps; ++r) {
      for (int i = 0; i < n2; ++i) h[i] = double2{(double)i + r, 1.0};
      hipMemcpyAsync(buf, h.data(), n2 * sizeof(double2), hipMemcpyHostToDevice, s);
      cons(0); double t = read_ns(); if (r >= 5) ts.push_back(t);
    }
    report("F  uploaded by the host (H2D copy) -> read on XCC 0", ts);
  }
  return 0;
}
