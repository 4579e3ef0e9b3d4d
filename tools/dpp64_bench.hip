// What does the 16-column elimination of k_chol2 cost, instruction by instruction?  Cycle counts (s_memtime) of one wave per SIMD:
//   * issue rate of v_fmac_f64, v_fmac_f64_dpp row_newbcast, v_mov_b64_dpp, v_readlane_b32 x 2 + v_fma_f64 with an SGPR operand
//   * the elimination as k_chol2 has it (a copy of the diagonal block in every DPP row, broadcast inside the FMA)
//   * the same elimination with the diagonal block in lanes 0..15 only, three panel tiles in the other DPP rows and the column of L
//     broadcast through SGPRs (v_readlane): one FMA per column serves the diagonal block AND the panel tiles
// hipcc --offload-arch=gfx950 -O3 -w -std=c++20 tools/dpp64_bench.hip -o /tmp/dpp64_bench   (output of the final tree: profiles/r03_e_dpp64_bench.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <utility>

template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  [&]<int... I>(std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

#define FMAC_DPP(acc, src_dpp, mul, J) \
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src_dpp), "v"(mul))
#define FMAC(acc, a, b) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b))
#define MOV_DPP(dst, src, J) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// ---- issue rates -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rates(long long* out, double* sink, double a0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 1e-3 * lane + i;
  double a = a0, b = 1e-3 * lane;
  long long t[8];
  __syncthreads();
  t[0] = __builtin_readcyclecounter();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) FMAC(acc[i], a, b);
  }
  t[1] = __builtin_readcyclecounter();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) FMAC_DPP(acc[i], b, a, 3);
  }
  t[2] = __builtin_readcyclecounter();
  double m[16];
#pragma unroll
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) MOV_DPP(m[i], acc[i], 5);
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(m[i]));
  }
  t[3] = __builtin_readcyclecounter();
  // readlane pair -> SGPR operand of a plain FMA (256 broadcasts)
#pragma unroll
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double s = readlane_f64(m[i], i);
      acc[i] = fma(s, a, acc[i]);
      asm volatile("" : "+v"(acc[i]));
    }
  }
  t[4] = __builtin_readcyclecounter();
  // dependent chain of plain FMAs
  double c = b;
#pragma unroll
  for (int it = 0; it < 256; ++it) {
    c = fma(c, a, a);
    asm volatile("" : "+v"(c));
  }
  t[5] = __builtin_readcyclecounter();
  // dependent chain of rsq
#pragma unroll
  for (int it = 0; it < 64; ++it) {
    c = __builtin_amdgcn_rsq(c);
    asm volatile("" : "+v"(c));
  }
  t[6] = __builtin_readcyclecounter();
  // dependent chain through readlane: VGPR -> SGPR -> VGPR
#pragma unroll
  for (int it = 0; it < 64; ++it) {
    const double s = readlane_f64(c, it & 15);
    c = fma(s, a, b);
    asm volatile("" : "+v"(c));
  }
  t[7] = __builtin_readcyclecounter();
  double s = c;
  for (int i = 0; i < 16; ++i) s += acc[i] + m[i];
  sink[threadIdx.x] = s;
  if (lane == 0)
    for (int i = 0; i < 8; ++i) out[wave * 8 + i] = t[i];
}

// ---- the elimination, as k_chol2 has it ------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ void fmac_bcast(double& acc, const double& s, const double& mul, bool nop) {
#define CASE(K)                                                                                                                   \
  if constexpr (J == K) {                                                                                                         \
    if (nop) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(mul)); \
    else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(mul));           \
  }
  CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
}
template <int J>
__device__ __forceinline__ double bcast_row(const double& src) {
  double dst;
#define CASE(K) \
  if constexpr (J == K) asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src));
  CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
  return dst;
}

__device__ __forceinline__ void elim_dpp(double (&d)[16], double (&p)[16]) {
  double piv = bcast_row<0>(d[0]);
  sfor<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const double y0 = __builtin_amdgcn_rsq(piv);
    const double hy = (0.5 * piv) * y0;
    const double ly = d[c] * y0, py = p[c] * y0;
    const double e = fma(-hy, y0, 0.5);
    const double l = fma(ly, e, ly);
    const double q = fma(py, e, py);
    double nl = fma(-ly, e, -ly);
    d[c] = l;
    p[c] = q;
    if constexpr (c + 1 < 16) {
      fmac_bcast<c + 1>(d[c + 1], nl, l, true);
      piv = bcast_row<c + 1>(d[c + 1]);
      fmac_bcast<c + 1>(p[c + 1], nl, q, false);
      sfor<14 - c>([&](auto jc) {
        constexpr int j = c + 2 + decltype(jc)::value;
        fmac_bcast<j>(d[j], nl, l, false);
        fmac_bcast<j>(p[j], nl, q, false);
      });
    }
  });
}

// ---- the elimination with the column of L in SGPRs --------------------------------------------------------------------------
// lanes 0..15: row r of the diagonal block, lanes 16..63: row r of three panel tiles; x = that row.  Column c: every lane scales
// its entry by 1 / sqrt(pivot) (lane c of the first DPP row holds the pivot), and x[j] -= x[c] * L_jc with L_jc = lane j's scaled
// entry, read into an SGPR pair.
__device__ __forceinline__ void elim_sgpr(double (&x)[16]) {
  double piv = readlane_f64(x[0], 0);
  sfor<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const double y0 = __builtin_amdgcn_rsq(piv);
    const double hy = (0.5 * piv) * y0;
    const double e = fma(-hy, y0, 0.5);
    const double ly = x[c] * y0;
    const double l = fma(ly, e, ly);
    const double nl = fma(-ly, e, -ly);
    x[c] = l;
    if constexpr (c + 1 < 16) {
      {
        const double s = readlane_f64(nl, c + 1);
        x[c + 1] = fma(s, l, x[c + 1]);
        piv = readlane_f64(x[c + 1], c + 1);
      }
      sfor<14 - c>([&](auto jc) {
        constexpr int j = c + 2 + decltype(jc)::value;
        const double s = readlane_f64(nl, j);
        x[j] = fma(s, l, x[j]);
      });
    }
  });
}


typedef double dbl2_t __attribute__((ext_vector_type(2)));
// the elimination exactly as k_chol2 runs it: pivots written out, pivot test, pivot floor, panel rows stored pair by pair
template <int WP, int PW, int FL>
__device__ __forceinline__ bool elim_dpp_full(double (&d)[16], double (&p)[16], double* __restrict__ piv_out, const bool write_piv,
                                              const double floor, dbl2_t* __restrict__ pw) {
  bool bad = false;
  double piv = bcast_row<0>(d[0]);
  double mypiv = 0.0;
  const int r = threadIdx.x & 15;
  sfor<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (WP == 1) {
      if (write_piv) piv_out[c] = piv;
    }
    if constexpr (WP == 2) mypiv = (r == c) ? piv : mypiv;
    if constexpr (FL == 1) bad = bad || !(piv > 0.0);
    const double y0 = (FL == 1 && floor > 0.0 && !(piv >= floor)) ? 0.0 : __builtin_amdgcn_rsq(piv);
    const double hy = (0.5 * piv) * y0;
    const double ly = d[c] * y0, py = p[c] * y0;
    const double e = fma(-hy, y0, 0.5);
    const double l = fma(ly, e, ly);
    const double q = fma(py, e, py);
    double nl = fma(-ly, e, -ly);
    d[c] = l;
    p[c] = q;
    if constexpr (c + 1 < 16) {
      fmac_bcast<c + 1>(d[c + 1], nl, l, true);
      piv = bcast_row<c + 1>(d[c + 1]);
      fmac_bcast<c + 1>(p[c + 1], nl, q, false);
      sfor<14 - c>([&](auto jc) {
        constexpr int j = c + 2 + decltype(jc)::value;
        fmac_bcast<j>(d[j], nl, l, false);
        fmac_bcast<j>(p[j], nl, q, false);
      });
    }
    if constexpr ((c & 1) == 1) {
      if constexpr (PW == 1) {
        if (pw) pw[c >> 1] = dbl2_t{p[c - 1], p[c]};
      }
      if constexpr (PW == 2) pw[c >> 1] = dbl2_t{p[c - 1], p[c]};
    }
  });
  if constexpr (WP == 2) {
    if (write_piv) piv_out[r] = mypiv;
    if constexpr (FL == 2) bad = !(mypiv > 0.0);
  }
  if constexpr (PW == 3) {
#pragma unroll
    for (int q = 0; q < 8; ++q) pw[q] = dbl2_t{p[2 * q], p[2 * q + 1]};
  }
  return bad;
}
__device__ __forceinline__ bool elim_sgpr_full(double (&x)[16], double* __restrict__ piv_out, const bool write_piv, const double floor,
                                               dbl2_t* __restrict__ pw) {
  bool bad = false;
  double piv = readlane_f64(x[0], 0);
  sfor<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (write_piv) piv_out[c] = piv;
    bad = bad || !(piv > 0.0);
    const double y0 = (floor > 0.0 && !(piv >= floor)) ? 0.0 : __builtin_amdgcn_rsq(piv);
    const double hy = (0.5 * piv) * y0;
    const double e = fma(-hy, y0, 0.5);
    const double ly = x[c] * y0;
    const double l = fma(ly, e, ly);
    const double nl = fma(-ly, e, -ly);
    x[c] = l;
    if constexpr (c + 1 < 16) {
      {
        const double s = readlane_f64(nl, c + 1);
        x[c + 1] = fma(s, l, x[c + 1]);
        piv = readlane_f64(x[c + 1], c + 1);
      }
      sfor<14 - c>([&](auto jc) {
        constexpr int j = c + 2 + decltype(jc)::value;
        const double s = readlane_f64(nl, j);
        x[j] = fma(s, l, x[j]);
      });
    }
    if constexpr ((c & 1) == 1) {
      if (pw) pw[c >> 1] = dbl2_t{x[c - 1], x[c]};
    }
  });
  return bad;
}

__device__ __forceinline__ long long now_fenced() {
  long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// A: 16 x 16 s.p.d. block, P: 4 panel tiles (16 x 16 each), row-major.  variant 0: DPP stripped, 1: SGPR stripped, 2: DPP as in the
// kernel, 3: SGPR with the same extras.  Waves 0..3 eliminate (one per SIMD), waves 4.. poll an LDS counter like idle tile waves.
__global__ __launch_bounds__(768) void k_elim(const double* A, const double* P, double* Lout, double* Pout, long long* cyc, int variant,
                                              int reps, double floor) {
  __shared__ double tiles[5 * 16 * 18];
  __shared__ double pivs[16];
  __shared__ int flag[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
  if (threadIdx.x < 2) flag[threadIdx.x] = 0;
  for (int i = threadIdx.x; i < 5 * 256; i += blockDim.x) {
    const int t = i >> 8, rr = (i >> 4) & 15, c = i & 15;
    tiles[t * 288 + rr * 18 + c] = t == 0 ? A[rr * 16 + c] : P[(t - 1) * 256 + rr * 16 + c];
  }
  __syncthreads();
  if (wave >= 4) {  // idle tile waves
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {
      asm volatile("s_nop 7");
      if (++spins > (1 << 22)) break;
    }
    return;
  }
  __builtin_amdgcn_s_setprio(3);
  long long t0 = 0, t1 = 0, t2 = 0;
  bool bad = false;
  double d[16], p[16];
  for (int rep = 0; rep < reps; ++rep) {
    t0 = now_fenced();
    if (variant != 1 && variant != 3) {
      const dbl2_t* dr = reinterpret_cast<const dbl2_t*>(tiles + r * 18);
      const dbl2_t* pr = reinterpret_cast<const dbl2_t*>(tiles + (1 + g) * 288 + r * 18);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const dbl2_t dv = dr[q], pq = pr[q];
        d[2 * q] = dv[0], d[2 * q + 1] = dv[1], p[2 * q] = pq[0], p[2 * q + 1] = pq[1];
      }
    } else {
      const dbl2_t* pr = reinterpret_cast<const dbl2_t*>(tiles + g * 288 + r * 18);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const dbl2_t pq = pr[q];
        p[2 * q] = pq[0], p[2 * q + 1] = pq[1];
      }
    }
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(d[c]), "+v"(p[c]));
    t1 = now_fenced();
    dbl2_t* pw = reinterpret_cast<dbl2_t*>(tiles + ((variant == 1 || variant == 3) ? g : 1 + g) * 288 + r * 18);
    if (variant == 0) elim_dpp(d, p);
    else if (variant == 1) elim_sgpr(p);
    else if (variant == 2) bad = elim_dpp_full<1, 1, 1>(d, p, pivs, wave == 0 && lane == 0, floor, g < 3 ? pw : nullptr) || bad;
    else if (variant == 4) bad = elim_dpp_full<0, 1, 1>(d, p, pivs, wave == 0 && lane == 0, floor, g < 3 ? pw : nullptr) || bad;
    else if (variant == 5) bad = elim_dpp_full<1, 0, 1>(d, p, pivs, wave == 0 && lane == 0, floor, pw) || bad;
    else if (variant == 6) bad = elim_dpp_full<1, 1, 0>(d, p, pivs, wave == 0 && lane == 0, floor, g < 3 ? pw : nullptr) || bad;
    else if (variant == 7) bad = elim_dpp_full<2, 2, 1>(d, p, pivs, wave == 0 && lane < 16, floor, pw) || bad;
    else if (variant == 8) bad = elim_dpp_full<2, 2, 2>(d, p, pivs, wave == 0 && lane < 16, floor, pw) || bad;
    else if (variant == 9) bad = elim_dpp_full<2, 3, 2>(d, p, pivs, wave == 0 && lane < 16, floor, pw) || bad;
    else if (variant == 10) bad = elim_dpp_full<0, 2, 0>(d, p, pivs, wave == 0 && lane < 16, floor, pw) || bad;
    else if (variant == 11) bad = elim_dpp_full<2, 0, 0>(d, p, pivs, wave == 0 && lane < 16, floor, pw) || bad;
    else bad = elim_sgpr_full(p, pivs, wave == 0 && lane == 0, floor, g ? pw : nullptr) || bad;
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(d[c]), "+v"(p[c]));
    t2 = now_fenced();
    // (restore the input for the next repetition)
    __builtin_amdgcn_wave_barrier();
    if (rep + 1 < reps && wave == 0)
      for (int i = lane; i < 5 * 256; i += 64) {
        const int t = i >> 8, rr = (i >> 4) & 15, c = i & 15;
        tiles[t * 288 + rr * 18 + c] = t == 0 ? A[rr * 16 + c] : P[(t - 1) * 256 + rr * 16 + c];
      }
  }
  if (lane == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (wave == 0) {
    for (int c = 0; c < 16; ++c) {
      if (variant == 1 || variant == 3) {
        if (g == 0) Lout[r * 16 + c] = p[c];
        else Pout[((g - 1) * 16 + r) * 16 + c] = p[c];
      } else {
        if (g == 0) Lout[r * 16 + c] = d[c];
        Pout[(g * 16 + r) * 16 + c] = p[c];
      }
    }
  }
  if (lane == 0) {
    cyc[wave * 2] = t1 - t0;
    cyc[wave * 2 + 1] = t2 - t1 + (bad ? 1000000 : 0);
  }
}

int main() {
  long long* out;
  double* sink;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&sink, 4096 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_rates, dim3(1), dim3(256), 0, 0, out, sink, 0.999);
    (void)hipDeviceSynchronize();
  }
  long long h[32];
  (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[7] = {"v_fmac_f64 x256", "v_fmac_f64_dpp x256", "v_mov_b64_dpp x256", "readlane x2 + v_fma(sgpr) x256", "dependent v_fma_f64 x256",
                          "dependent v_rsq_f64 x64", "dependent readlane x2 + v_fma x64"};
  const int cnt[7] = {256, 256, 256, 256, 256, 64, 64};
  for (int w = 0; w < 4; w += 3)
    for (int i = 0; i < 7; ++i) printf("wave %d  %-36s %6lld cycles  %.1f per instruction (group)\n", w, names[i], h[w * 8 + i + 1] - h[w * 8 + i], double(h[w * 8 + i + 1] - h[w * 8 + i]) / cnt[i]);

  // elimination variants
  std::vector<double> A(256), P(4 * 256), M(16 * 20);
  srand(1);
  for (auto& v : M) v = rand() / double(RAND_MAX) - 0.5;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = 0; k < 20; ++k) s += M[i * 20 + k] * M[j * 20 + k];
      A[i * 16 + j] = s;
    }
  for (auto& v : P) v = rand() / double(RAND_MAX) - 0.5;
  double *dA, *dP, *dL, *dPo;
  long long* dc;
  (void)hipMalloc(&dA, 256 * 8);
  (void)hipMalloc(&dP, 1024 * 8);
  (void)hipMalloc(&dL, 256 * 8);
  (void)hipMalloc(&dPo, 1024 * 8);
  (void)hipMalloc(&dc, 256);
  (void)hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dP, P.data(), 1024 * 8, hipMemcpyHostToDevice);
  std::vector<double> L0(256), P0(1024), L1(256), P1(1024);
  const char* vn[12] = {"DPP, stripped", "SGPR, stripped", "DPP as in k_chol2", "SGPR + the kernel's extras", "k_chol2 - pivot store", "k_chol2 - panel stores", "k_chol2 - floor/bad",
                        "pivot by select, stores unmasked", "... + bad at the end", "... + panel stored at the end", "only unmasked panel stores", "only pivot by select"};
  for (int variant = 0; variant < 12; ++variant) {
    (void)hipMemset(dPo, 0, 1024 * 8);
    for (int threads : {256}) {
      for (double floor : {0.0, 1e-9}) {
        if (variant < 2 && floor > 0) continue;
        hipLaunchKernelGGL(k_elim, dim3(1), dim3(threads), 0, 0, dA, dP, dL, dPo, dc, variant, 3, floor);
        (void)hipDeviceSynchronize();
        long long c[8];
        (void)hipMemcpy(c, dc, 64, hipMemcpyDeviceToHost);
        printf("elimination %-34s %2d waves, floor %g: LDS loads %lld, elimination %lld cycles (wave 0)\n", vn[variant], threads / 64, floor, c[0], c[1]);
      }
    }
    if (variant < 2) {
      (void)hipMemcpy(variant ? L1.data() : L0.data(), dL, 256 * 8, hipMemcpyDeviceToHost);
      (void)hipMemcpy(variant ? P1.data() : P0.data(), dPo, 1024 * 8, hipMemcpyDeviceToHost);
    }
  }
  double dl = 0, dp = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j <= i; ++j) dl = fmax(dl, fabs(L0[i * 16 + j] - L1[i * 16 + j]));
  for (int i = 0; i < 3 * 256; ++i) dp = fmax(dp, fabs(P0[i] - P1[i]));
  // against a host elimination
  std::vector<double> Lh(A);
  for (int c = 0; c < 16; ++c) {
    Lh[c * 16 + c] = sqrt(Lh[c * 16 + c]);
    for (int i = c + 1; i < 16; ++i) Lh[i * 16 + c] /= Lh[c * 16 + c];
    for (int j = c + 1; j < 16; ++j)
      for (int i = j; i < 16; ++i) Lh[i * 16 + j] -= Lh[i * 16 + c] * Lh[j * 16 + c];
  }
  double dh = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j <= i; ++j) dh = fmax(dh, fabs(Lh[i * 16 + j] - L1[i * 16 + j]));
  printf("max |L_dpp - L_sgpr| %.3g, panel tiles 0..2 %.3g, |L_host - L_sgpr| %.3g\n", dl, dp, dh);
  return 0;
}
