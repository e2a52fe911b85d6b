// Accuracy of v_rcp_f64 on gfx950 (diagnostic): max relative error of the raw instruction and after one / two
// Newton steps against the IEEE quotient, over 1M doubles spread over many binades.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* e0, double* e1, double* e2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  const double ex = 1.0 / v;
  e0[i] = fabs(r - ex) / fabs(ex);
  r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
  e1[i] = fabs(r - ex) / fabs(ex);
  r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
  e2[i] = fabs(r - ex) / fabs(ex);
}
int main() {
  const int n = 1 << 20;
  double *x, *e0, *e1, *e2;
  hipMallocManaged(&x, n * 8); hipMallocManaged(&e0, n * 8); hipMallocManaged(&e1, n * 8); hipMallocManaged(&e2, n * 8);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double m = 1.0 + double(s >> 11) / 9007199254740992.0;
    x[i] = ldexp(m, int(s % 61) - 30) * ((s >> 5) & 1 ? 1.0 : -1.0);
  }
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, e0, e1, e2, n);
  hipDeviceSynchronize();
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; i++) { m0 = fmax(m0, e0[i]); m1 = fmax(m1, e1[i]); m2 = fmax(m2, e2[i]); }
  printf("v_rcp_f64 max rel err: raw %.3e  after one Newton step %.3e  after two %.3e  (ulp = 1.1e-16)\n", m0, m1, m2);
  return 0;
}
