// Micro-benchmark (diagnostic): where do the wavefronts of the solver's workgroups land?
// Launches a grid with the trial kernel's geometry (2 waves, 40 KB LDS: four workgroups per CU) and one with the sweep
// kernel's (3 waves, 30 KB), every wave records HW_ID / XCC_ID and stays resident for ~40 us so that the whole grid is
// co-resident, then the host prints, per CU, which (workgroup, wave) sits on which SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  extern __shared__ int lds[];
  const int wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    out[(blockIdx.x * W + wave) * 2 + 0] = hw;
    out[(blockIdx.x * W + wave) * 2 + 1] = xcc;
  }
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(16);
  __syncthreads();
}

static void run(int grid, int waves, int lds_bytes, int working = 0) {  // working > 0: only waves < working count
  unsigned* d;
  const size_t n = size_t(grid) * waves * 2;
  hipMalloc(&d, n * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(probe, dim3(grid), dim3(64 * waves), lds_bytes, 0, d, 100000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(n);
  hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  // key = (xcc, se, sh, cu)
  std::map<unsigned, std::vector<std::pair<int, int>>> cu;  // -> (block*W+wave, simd)
  for (int i = 0; i < grid * waves; i++) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
    const unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back({i, int(simd)});
  }
  printf("grid %d x %d waves, %d B LDS: %zu distinct CUs\n", grid, waves, lds_bytes, cu.size());
  int shown = 0;
  std::map<std::string, int> patterns;
  for (auto& kv : cu) {
    std::string pat;
    int per_simd[4] = {0, 0, 0, 0}, w0_simd[4] = {0, 0, 0, 0};
    for (auto& e : kv.second) {
      if (working > 0 && e.first % waves >= working) continue;
      per_simd[e.second]++;
      if (e.first % waves == 0) w0_simd[e.second]++;
    }
    char buf[128];
    snprintf(buf, sizeof buf, "waves/simd %d %d %d %d | wave0/simd %d %d %d %d", per_simd[0], per_simd[1], per_simd[2],
             per_simd[3], w0_simd[0], w0_simd[1], w0_simd[2], w0_simd[3]);
    patterns[buf]++;
    if (shown < 6) {
      printf("  cu %05x:", kv.first);
      for (auto& e : kv.second) printf(" b%d.w%d@s%d", e.first / waves, e.first % waves, e.second);
      printf("\n");
      shown++;
    }
  }
  for (auto& p : patterns) printf("  %4d CUs: %s\n", p.second, p.first.c_str());
  hipFree(d);
}

int main() {
  run(1024, 2, 40176);
  run(1024, 3, 30272);
  run(512, 4, 80352);
  run(256, 8, 160000);
  printf("-- the sweep's geometry with an idle fourth wave (only waves 0 .. 2 counted)\n");
  run(1024, 4, 35072, 3);
  run(1024, 4, 30272, 3);
  return 0;
}
