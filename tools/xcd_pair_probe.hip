// Two workgroups of ONE launch that must talk to each other (k_chol2's split factorization: panel tiles, gate, y blocks).
// Workgroups go to the eight XCDs in turn, so blocks b and b + 8 of a grid share an XCD - and its L2.  Questions:
//   1. how often do blocks 0 and `stride` of a (stride + 1)-block launch (130 KB LDS each, the others return at once) really sit on
//      the same XCC, with other grids dispatched in between;
//   2. what a hand-over costs (a 2 KB tile + a flag one way, a flag back) through memory (`sc0 sc1` stores and loads, agent-scope
//      acquire on the flag: what k_chol2 does today) against through the shared L2 (plain stores, `sc1` loads: L1 bypassed, L2 hit);
//   3. whether the L2 form is ever stale on the same XCC (it must be on different ones - that is the control).
// Output: one line per (stride, mode): same-XCC launches, cycles per round trip (s_memtime, 100 MHz) -> ns, stale tiles seen.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double dbl2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return (int)(x & 15);
}
__device__ __forceinline__ void st_sys(double* p, dbl2_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_l2(double* p, dbl2_t v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ dbl2_t ld_sys(const double* p) {
  dbl2_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ dbl2_t ld_l2(const double* p) {
  dbl2_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned flag_ld_l2(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void flag_st_l2(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// out: [0] xcc of A, [1] xcc of B, [2] wall-clock ticks of A for the rounds, [3] stale tiles seen by B, [4] timed out
__global__ __launch_bounds__(768) void k_pair(int stride, int mode, int rounds, unsigned seq0, double* tile, unsigned* flags, long long* out) {
  extern __shared__ double lds[];
  const int b = blockIdx.x;
  if (b != 0 && b != stride) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  lds[tid] = 0.0;
  if (tid == 0) out[b == 0 ? 0 : 1] = xcc_id();
  if (wave != 0) return;
  unsigned* f_go = flags;       // A -> B
  unsigned* f_ack = flags + 64;  // B -> A (another line)
  long long stale = 0, to = 0;
  const long long t0 = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
    const unsigned seq = seq0 + (unsigned)r;
    double* tp = tile + (size_t)(r & 7) * 256 + 4 * lane;  // eight tile slots in turn, 32 bytes per lane
    if (b == 0) {
      const dbl2_t v = {(double)seq, (double)lane};
      if (mode == 0) {
        st_sys(tp, v);
        st_sys(tp + 2, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(f_go, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(f_ack, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq)
          if (++spins > (1 << 16)) { to = 1; break; }
      } else {
        st_l2(tp, v);
        st_l2(tp + 2, v);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) flag_st_l2(f_go, seq);
        int spins = 0;
        while (flag_ld_l2(f_ack) != seq)
          if (++spins > (1 << 16)) { to = 1; break; }
      }
    } else {
      dbl2_t v0, v1;
      if (mode == 0) {
        int spins = 0;
        while (__hip_atomic_load(f_go, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq)
          if (++spins > (1 << 16)) { to = 1; break; }
        v0 = ld_sys(tp);
        v1 = ld_sys(tp + 2);
        if (v0[0] != (double)seq || v1[0] != (double)seq || v0[1] != (double)lane) ++stale;
        if (lane == 0) __hip_atomic_store(f_ack, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        int spins = 0;
        while (flag_ld_l2(f_go) != seq)
          if (++spins > (1 << 16)) { to = 1; break; }
        v0 = ld_l2(tp);
        v1 = ld_l2(tp + 2);
        if (v0[0] != (double)seq || v1[0] != (double)seq || v0[1] != (double)lane) ++stale;
        if (lane == 0) flag_st_l2(f_ack, seq);
      }
    }
    if (to) break;
  }
  const long long t1 = wall_clock64();
  for (int m = 32; m >= 1; m >>= 1) stale += __shfl_xor(stale, m);
  if (lane == 0) {
    if (b == 0) out[2] = t1 - t0;
    else out[3] = stale;
    if (to) out[4] = 1;
  }
}
__global__ void k_filler(int* x) {
  if (threadIdx.x == 0 && blockIdx.x == 0) x[0]++;
}
int main() {
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  double* tile;
  unsigned* flags;
  long long* out;
  int* fx;
  (void)hipMalloc(&tile, 8 * 256 * 8);
  (void)hipMalloc(&flags, 4 * 128);
  (void)hipMalloc(&out, 8 * 8);
  (void)hipMalloc(&fx, 64);
  (void)hipMemset(flags, 0, 4 * 128);
  (void)hipMemset(fx, 0, 64);
  int rate = 0;
  (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
  (void)hipFuncSetAttribute((const void*)k_pair, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024);
  const int rounds = 200, reps = 40;
  unsigned seq = 1;
  for (int stride : {1, 2, 8, 16})
    for (int mode : {0, 1}) {
      int same = 0, timed_out = 0;
      long long stale_same = 0, stale_diff = 0;
      double ticks_same = 0, ticks_diff = 0;
      int n_same = 0, n_diff = 0;
      for (int r = 0; r < reps; ++r) {
        // other grids in between, as in the plane loop (the round over the XCDs starts somewhere else every time)
        hipLaunchKernelGGL(k_filler, dim3(1 + (r * 7) % 23), dim3(64), 0, s, fx);
        hipLaunchKernelGGL(k_filler, dim3(17 + (r * 5) % 11), dim3(256), 0, s, fx);
        (void)hipMemsetAsync(out, 0, 64, s);
        hipLaunchKernelGGL(k_pair, dim3(stride + 1), dim3(768), 130 * 1024, s, stride, mode, rounds, seq, tile, flags, out);
        seq += rounds + 8;
        (void)hipStreamSynchronize(s);
        long long h[8];
        (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        const bool sm = h[0] == h[1];
        same += sm;
        timed_out += (int)h[4];
        if (sm) {
          stale_same += h[3];
          ticks_same += (double)h[2];
          ++n_same;
        } else {
          stale_diff += h[3];
          ticks_diff += (double)h[2];
          ++n_diff;
        }
      }
      const double ns = 1e6 / rate;  // ns per tick
      printf("stride %2d  %-22s same XCC in %2d of %d launches | round trip: same XCC %7.0f ns (%d), other XCC %7.0f ns (%d) | stale tiles: same %lld, other %lld | timeouts %d\n",
             stride, mode == 0 ? "through memory (sys)" : "through the L2 (sc1)", same, reps, n_same ? ticks_same * ns / n_same / rounds : 0.0, n_same,
             n_diff ? ticks_diff * ns / n_diff / rounds : 0.0, n_diff, stale_same, stale_diff, timed_out);
    }
  return 0;
}
