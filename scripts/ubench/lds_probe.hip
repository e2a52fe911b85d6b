// Micro-benchmark (diagnostic): what the one-tile sweep's shared resources cost on gfx950.
//   A. LDS pipe occupancy per CU of ds_read_b64 / b128 with all lanes, 14 lanes, and same-address (broadcast) reads,
//      12 waves per CU (the sweep's residency), every wave issuing back-to-back independent reads.
//   B. does a wave's fp64 MFMA stream slow another wave's fp64 FMA stream on the same SIMD?
//   C. semantics of v_permlane32_swap / v_permlane16_swap / DPP row_newbcast (printed lane maps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ long long clk() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 0: b64 all lanes distinct (stride 1)   1: b64 lanes < 14   2: b128 same address   3: b128 distinct
// 4: b64 same address   5: b64 lanes < 16 (one row)   6: b64 lanes with j==14 (4 lanes)
template <int MODE>
__global__ void __launch_bounds__(768) lds_k(long long* out, double* sink, int iters) {
  __shared__ __align__(16) double lds[4096];
  const int t = threadIdx.x, l = t & 63;
  for (int e = t; e < 4096; e += blockDim.x) lds[e] = e;
  __syncthreads();
  double acc = 0, acc2 = 0;
  const int base = (t >> 6) * 64;
  const long long t0 = clk();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int o = (u * 128 + it) & 2047;
      if (MODE == 0) acc += lds[o + base + l];
      if (MODE == 1) { if (l < 14) acc += lds[o + base + l]; }
      if (MODE == 5) { if (l < 16) acc += lds[o + base + l]; }
      if (MODE == 6) { if ((l & 15) == 14) acc += lds[o + base + l]; }
      if (MODE == 4) acc += lds[o + base];
      if (MODE == 2) { const double2 v = *reinterpret_cast<const double2*>(&lds[(o + base) & ~1]); acc += v.x; acc2 += v.y; }
      if (MODE == 3) { const double2 v = *reinterpret_cast<const double2*>(&lds[((o + base) & ~1) + 2 * l]); acc += v.x; acc2 += v.y; }
    }
  }
  const long long t1 = clk();
  if (l == 0) out[blockIdx.x * (blockDim.x >> 6) + (t >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + t] = acc + acc2;
}

// waves 0-3 of a block: MFMA stream (if mf), waves 4-7: independent fp64 FMA streams (if fm)
__global__ void __launch_bounds__(512) share_k(long long* out, double* sink, int iters, int mf, int fm) {
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  double a = 1.0 + l * 1e-3, b = 0.5;
  v4d c = {0, 0, 0, 0};
  double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
  __syncthreads();
  const long long t0 = clk();
  if (w < 4) {
    if (mf)
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
      }
  } else if (fm) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        x0 = __builtin_fma(x0, 1.0000001, 0.5); x1 = __builtin_fma(x1, 1.0000001, 0.5);
        x2 = __builtin_fma(x2, 1.0000001, 0.5); x3 = __builtin_fma(x3, 1.0000001, 0.5);
        x4 = __builtin_fma(x4, 1.0000001, 0.5); x5 = __builtin_fma(x5, 1.0000001, 0.5);
        x6 = __builtin_fma(x6, 1.0000001, 0.5); x7 = __builtin_fma(x7, 1.0000001, 0.5);
      }
    }
  }
  const long long t1 = clk();
  if (l == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  sink[blockIdx.x * blockDim.x + t] = c[0] + c[1] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void perm_k(int* out) {
  const int l = threadIdx.x;
  int a = l, b = 100 + l;
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
  }
  {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + l] = r[0]; out[192 + l] = r[1];
  }
#endif
  // DPP row_newbcast:5 (0x150 + lane): lane 5 of each row of 16 to the whole row
  out[256 + l] = __builtin_amdgcn_update_dpp(0, a, 0x155, 0xf, 0xf, false);
  // row_shr:1 (0x111), row_ror:4 (0x124)
  out[320 + l] = __builtin_amdgcn_update_dpp(-1, a, 0x111, 0xf, 0xf, false);
  out[384 + l] = __builtin_amdgcn_update_dpp(-1, a, 0x124, 0xf, 0xf, false);
}

template <int MODE>
static void run_lds(const char* name, int waves_per_block, int blocks, long long* dout, double* dsink) {
  const int iters = 200;
  hipLaunchKernelGGL(lds_k<MODE>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, dout, dsink, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(lds_k<MODE>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, dout, dsink, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * waves_per_block);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += double(v);
  mean /= h.size();
  // every wave issued iters*16 reads; the CU's LDS served waves_per_block * that in `mean` cycles
  printf("%-34s %2d waves/CU: %.1f cycles per read per wave, %.2f LDS cycles per read (CU aggregate)\n", name, waves_per_block,
         mean / (iters * 16.0), mean / (iters * 16.0 * waves_per_block));
}

int main() {
  long long* dout; double* dsink;
  hipMalloc(&dout, 1 << 20); hipMalloc(&dsink, 64 << 20);
  for (int wpb : {1, 4, 12}) {
    run_lds<0>("b64 64 lanes distinct", wpb, 256, dout, dsink);
    run_lds<1>("b64 lanes < 14", wpb, 256, dout, dsink);
    run_lds<5>("b64 lanes < 16", wpb, 256, dout, dsink);
    run_lds<6>("b64 4 lanes (j == 14)", wpb, 256, dout, dsink);
    run_lds<4>("b64 same address", wpb, 256, dout, dsink);
    run_lds<2>("b128 same address", wpb, 256, dout, dsink);
    run_lds<3>("b128 64 lanes distinct", wpb, 256, dout, dsink);
  }
  for (int cfg = 0; cfg < 3; cfg++) {
    const int mf = cfg != 1, fm = cfg != 0, iters = 2000;
    hipLaunchKernelGGL(share_k, dim3(256), dim3(512), 0, 0, dout, dsink, iters, mf, fm);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    printf("mfma %d fma %d: cycles per 8 MFMA (waves 0-3) %lld %lld %lld %lld | per 64 FMA (waves 4-7) %lld %lld %lld %lld\n", mf, fm,
           h[0] / iters, h[1] / iters, h[2] / iters, h[3] / iters, h[4] / iters, h[5] / iters, h[6] / iters, h[7] / iters);
  }
  int* dperm; hipMalloc(&dperm, 448 * 4); hipMemset(dperm, 0xff, 448 * 4);
  hipLaunchKernelGGL(perm_k, dim3(1), dim3(64), 0, 0, dperm);
  hipDeviceSynchronize();
  int hp[448]; hipMemcpy(hp, dperm, sizeof(hp), hipMemcpyDeviceToHost);
  const char* names[7] = {"permlane32_swap r0 (a=l, b=100+l)", "permlane32_swap r1", "permlane16_swap r0", "permlane16_swap r1",
                          "dpp row_newbcast:5", "dpp row_shr:1 (old -1)", "dpp row_ror:4"};
  for (int q = 0; q < 7; q++) {
    printf("%s:\n ", names[q]);
    for (int l = 0; l < 64; l++) printf(" %3d%s", hp[q * 64 + l], (l & 15) == 15 ? "\n " : "");
    printf("\n");
  }
  return 0;
}
