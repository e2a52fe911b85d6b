// Micro-benchmark (diagnostic): dependent-chain latency (shader cycles, one wave) of the fp64 / fp32 libm calls on
// the rollout's critical path, next to the range-limited versions of ilqg_trig.hpp, and their worst disagreement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../ilqgames_amd/csrc/ilqg_trig.hpp"
__device__ __forceinline__ long long clk() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define USE(x) asm volatile("" :: "v"(x))
#define TOUCH(x) asm volatile("" : "+v"(x))
constexpr int R = 32;
__global__ void k(long long* out, double* sink, double* errs) {
  const int l = threadIdx.x;
  double x = 0.3 + l * 0.01; TOUCH(x);
  long long t0 = clk();
#pragma unroll 1
  for (int i = 0; i < R; i++) { double s, c; sincos(x, &s, &c); x = s + c + 0.1; }
  USE(x); long long t1 = clk();
  double y = 0.3 + l * 0.01; TOUCH(y);
#pragma unroll 1
  for (int i = 0; i < R; i++) { y = tan(y) * 0.25 + 0.1; }
  USE(y); long long t2 = clk();
  double z = 0.3 + l * 0.01; TOUCH(z);
#pragma unroll 1
  for (int i = 0; i < R; i++) { double s, c; ilqg::fast_sincos(z, &s, &c); z = s + c + 0.1; }
  USE(z); long long t3 = clk();
  double w = 0.3 + l * 0.01; TOUCH(w);
#pragma unroll 1
  for (int i = 0; i < R; i++) { w = ilqg::fast_tan(w) * 0.25 + 0.1; }
  USE(w); long long t4 = clk();
  float xf = 0.3f + l * 0.01f; TOUCH(xf);
#pragma unroll 1
  for (int i = 0; i < R; i++) { float s, c; sincosf(xf, &s, &c); xf = s + c + 0.1f; }
  USE(xf); long long t5 = clk();
  float yf = 0.3f + l * 0.01f; TOUCH(yf);
#pragma unroll 1
  for (int i = 0; i < R; i++) { yf = tanf(yf) * 0.25f + 0.1f; }
  USE(yf); long long t6 = clk();
  float zf = 0.3f + l * 0.01f; TOUCH(zf);
#pragma unroll 1
  for (int i = 0; i < R; i++) { float s, c; ilqg::fast_sincos(zf, &s, &c); zf = s + c + 0.1f; }
  USE(zf); long long t7 = clk();
  float wf = 0.3f + l * 0.01f; TOUCH(wf);
#pragma unroll 1
  for (int i = 0; i < R; i++) { wf = ilqg::fast_tan(wf) * 0.25f + 0.1f; }
  USE(wf); long long t8 = clk();
  double h = 1.0 + l; TOUCH(h);
#pragma unroll 1
  for (int i = 0; i < R; i++) { h = hypot(h, 0.5) + 0.25; }
  USE(h); long long t9 = clk();
  if (l == 0) {
    out[0] = (t1 - t0) / R; out[1] = (t2 - t1) / R; out[2] = (t3 - t2) / R; out[3] = (t4 - t3) / R;
    out[4] = (t5 - t4) / R; out[5] = (t6 - t5) / R; out[6] = (t7 - t6) / R; out[7] = (t8 - t7) / R; out[8] = (t9 - t8) / R;
  }
  sink[l] = x + y + z + w + xf + yf + zf + wf + h;
  // accuracy sweep: lane l covers arguments in [-40, 40] (sincos) and [-1.4, 1.4] (tan)
  double es = 0, et = 0, esf = 0, etf = 0;
  for (int i = 0; i < 4000; i++) {
    const double a = -40.0 + 80.0 * ((l * 4000 + i) + 0.37) / (64.0 * 4000.0);
    double s0, c0, s1, c1;
    sincos(a, &s0, &c0); ilqg::fast_sincos(a, &s1, &c1);
    es = fmax(es, fmax(fabs(s0 - s1), fabs(c0 - c1)));
    const double b = a * (1.4 / 40.0);
    const double t0v = tan(b), t1v = ilqg::fast_tan(b);
    et = fmax(et, fabs(t0v - t1v) / fmax(1.0, fabs(t0v)));
    float sf0, cf0, sf1, cf1;
    sincosf((float)a, &sf0, &cf0); ilqg::fast_sincos((float)a, &sf1, &cf1);
    esf = fmax(esf, (double)fmaxf(fabsf(sf0 - sf1), fabsf(cf0 - cf1)));
    const float tf0 = tanf((float)b), tf1 = ilqg::fast_tan((float)b);
    etf = fmax(etf, (double)(fabsf(tf0 - tf1) / fmaxf(1.0f, fabsf(tf0))));
  }
  errs[4 * l + 0] = es; errs[4 * l + 1] = et; errs[4 * l + 2] = esf; errs[4 * l + 3] = etf;
}
int main() {
  long long* o; double *s, *e;
  hipMalloc(&o, 128); hipMalloc(&s, 64 * 8); hipMalloc(&e, 256 * 8);
  for (int rep = 0; rep < 2; rep++) k<<<1, 64>>>(o, s, e);
  long long h[16]; double he[256];
  hipMemcpy(h, o, 128, hipMemcpyDeviceToHost);
  hipMemcpy(he, e, 256 * 8, hipMemcpyDeviceToHost);
  double m[4] = {0, 0, 0, 0};
  for (int i = 0; i < 256; i++) m[i & 3] = fmax(m[i & 3], he[i]);
  printf("cycles/call (dependent chain, 1 wave): sincos_f64 libm %lld fast %lld | tan_f64 libm %lld fast %lld | sincosf libm %lld fast %lld | tanf libm %lld fast %lld | hypot_f64 %lld\n",
         h[0], h[2], h[1], h[3], h[4], h[6], h[5], h[7], h[8]);
  printf("max |libm - fast|: sincos_f64 %.3e  tan_f64 (rel) %.3e  sincosf %.3e  tanf (rel) %.3e\n", m[0], m[1], m[2], m[3]);
  return 0;
}
