// Which XCDs does a stream created with hipExtStreamCreateWithCUMask run on?  (tools: hipcc --offload-arch=gfx950 -O2 tools/xcd_mask_probe.hip -o /tmp/xcd_probe)
// A kernel of 512 one-wave blocks records HW_REG_XCC_ID and the CU id of every block; the histogram per mask pattern is printed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__global__ void k_where(int* xcc, int* cu, long long spin) {
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = (int)(x & 0xf);
    cu[blockIdx.x] = (int)((h >> 8) & 0xf) | (int)(((h >> 13) & 0x7) << 4);  // CU_ID | SE_ID << 4 (gfx9 HW_ID layout)
  }
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) {
    printf("%-28s hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e));
    return;
  }
  const int nb = 512;
  int *dx, *dc;
  hipMalloc(&dx, nb * sizeof(int));
  hipMalloc(&dc, nb * sizeof(int));
  hipLaunchKernelGGL(k_where, dim3(nb), dim3(64), 0, s, dx, dc, 200000LL);
  hipStreamSynchronize(s);
  std::vector<int> hx(nb), hc(nb);
  hipMemcpy(hx.data(), dx, nb * sizeof(int), hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), dc, nb * sizeof(int), hipMemcpyDeviceToHost);
  int hist[16] = {0};
  for (int i = 0; i < nb; ++i) hist[hx[i] & 15]++;
  printf("%-28s blocks per XCC:", name);
  for (int i = 0; i < 8; ++i) printf(" %4d", hist[i]);
  // distinct (xcc, cu) pairs
  std::vector<char> seen(16 * 256, 0);
  int distinct = 0;
  for (int i = 0; i < nb; ++i) {
    const int key = (hx[i] & 15) * 256 + (hc[i] & 255);
    if (!seen[key]) seen[key] = 1, ++distinct;
  }
  printf("   distinct CUs used: %d\n", distinct);
  hipFree(dx);
  hipFree(dc);
  hipStreamDestroy(s);
}

int main() {
  std::vector<uint32_t> all(8, 0xffffffffu);
  run("all 256", all);
  for (int w = 0; w < 8; ++w) {
    std::vector<uint32_t> m(8, 0u);
    m[w] = 0xffffffffu;
    char nm[64];
    snprintf(nm, sizeof nm, "word %d (bits %d..%d)", w, 32 * w, 32 * w + 31);
    run(nm, m);
  }
  {
    std::vector<uint32_t> m(8, 0u);
    for (int b = 0; b < 256; b += 8) m[b / 32] |= 1u << (b % 32);
    run("bits = 0 mod 8", m);
  }
  {
    std::vector<uint32_t> m(8, 0u);
    for (int b = 0; b < 256; ++b)
      if ((b / 4) % 8 == 0) m[b / 32] |= 1u << (b % 32);
    run("bits b/4 = 0 mod 8", m);
  }
  return 0;
}
