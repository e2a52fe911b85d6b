// Diagnostic: sub_integrate_stages (a lane per RK4 stage, exchange through LDS) against sub_integrate_stages_seq (the
// eight stages in one lane) on random states and controls, bit for bit.   hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -I include -mllvm -amdgpu-mfma-vgpr-form=1 scripts/ubench/integ_seq_check.hip -o /tmp/isc && /tmp/isc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../../ilqgames_amd/csrc/ilqg_models.hpp"
using namespace ilqg;

template <typename T>
__global__ void k(const T* in, T* out_par, T* out_seq, int kind, double L, double dt) {
  __shared__ T gth[192];
  const int t = threadIdx.x, grp = t >> 3, q = t & 7;
  const size_t item = size_t(blockIdx.x) * 8 + grp;  // 8 items per block of 64 lanes
  T x[6], y[6];
  for (int e = 0; e < 6; e++) x[e] = y[e] = in[item * 8 + e];
  const T u0 = in[item * 8 + 6], u1 = in[item * 8 + 7];
  sub_integrate_stages<T, false, false>(kind, T(L), dt, x, u0, u1, q, t, gth, true);
  sub_integrate_stages_seq<T, false>(kind, T(L), dt, y, u0, u1);
  if (q == 0)
    for (int e = 0; e < 6; e++) {
      out_par[item * 6 + e] = x[e];
      out_seq[item * 6 + e] = y[e];
    }
}

template <typename T>
int run(const char* name) {
  const int items = 1 << 20;
  std::mt19937_64 rng(5);
  std::uniform_real_distribution<double> u(-1, 1);
  std::vector<T> in(size_t(items) * 8);
  for (int i = 0; i < items; i++) {
    const double scale = (i % 7 == 0) ? 1e3 : ((i % 11 == 0) ? 1e7 : 1.0);
    in[i * 8 + 0] = T(30 * u(rng)); in[i * 8 + 1] = T(30 * u(rng));
    in[i * 8 + 2] = T(3.2 * u(rng) * scale); in[i * 8 + 3] = T(0.6 * u(rng) * scale);
    in[i * 8 + 4] = T(10 * u(rng)); in[i * 8 + 5] = T(2 * u(rng));
    in[i * 8 + 6] = T(1.0 * u(rng) * scale); in[i * 8 + 7] = T(3 * u(rng));
  }
  T *din, *dp, *ds;
  hipMalloc(&din, in.size() * sizeof(T)); hipMalloc(&dp, size_t(items) * 6 * sizeof(T)); hipMalloc(&ds, size_t(items) * 6 * sizeof(T));
  hipMemcpy(din, in.data(), in.size() * sizeof(T), hipMemcpyHostToDevice);
  int bad_total = 0;
  for (int kind : {ILQG_DYN_UNICYCLE_4D, ILQG_DYN_CAR_5D, ILQG_DYN_CAR_6D}) {
    hipLaunchKernelGGL(k<T>, dim3(items / 8), dim3(64), 0, 0, din, dp, ds, kind, 4.0, 0.1);
    std::vector<T> a(size_t(items) * 6), b(size_t(items) * 6);
    hipMemcpy(a.data(), dp, a.size() * sizeof(T), hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), ds, b.size() * sizeof(T), hipMemcpyDeviceToHost);
    int bad[6] = {0, 0, 0, 0, 0, 0}, shown = 0;
    for (int i = 0; i < items; i++)
      for (int e = 0; e < 6; e++)
        if (std::memcmp(&a[i * 6 + e], &b[i * 6 + e], sizeof(T)) != 0 && !(a[i * 6 + e] != a[i * 6 + e] && b[i * 6 + e] != b[i * 6 + e])) {
          bad[e]++;
          if (shown++ < 3) printf("  item %d comp %d: par %.17g seq %.17g (x2 %.17g phi %.17g u0 %.17g)\n", i, e, double(a[i * 6 + e]), double(b[i * 6 + e]), double(in[i * 8 + 2]), double(in[i * 8 + 3]), double(in[i * 8 + 6]));
        }
    printf("%s kind %d: mismatches per component %d %d %d %d %d %d of %d\n", name, kind, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], items);
    for (int e = 0; e < 6; e++) bad_total += bad[e];
  }
  return bad_total;
}

int main() {
  int bad = run<double>("double") + run<float>("float");
  printf("%s\n", bad ? "MISMATCH" : "IDENTICAL");
  return bad ? 1 : 0;
}
