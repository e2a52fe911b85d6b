// Micro-benchmark (diagnostic): cycles per v_mfma_{f64,f32}_16x16x4 for dependent and independent
// chains, plus fp64 FMA / LDS read / readlane dependent latencies, one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long clk() { long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define USE(x) asm volatile("" :: "v"(x))
#define TOUCH(x) asm volatile("" : "+v"(x))
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(long long* out, double* sink) {
  __shared__ double lds[256];
  const int l = threadIdx.x;
  lds[l] = l; lds[l + 64] = 1.0 + l;
  __syncthreads();
  double a = 1.0 + l * 1e-3, b = 0.5;
  v4d c = {0, 0, 0, 0}, c2 = {1, 1, 1, 1}, c3 = {2, 2, 2, 2}, c4 = {3, 3, 3, 3};
  long long t0 = clk(); TOUCH(a); TOUCH(b);
#pragma unroll
  for (int i = 0; i < 64; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  USE(c[0]); long long t1 = clk(); TOUCH(a);
#pragma unroll
  for (int i = 0; i < 16; i++) {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
  }
  USE(c[0]); USE(c2[0]); USE(c3[0]); USE(c4[0]); long long t2 = clk();
  float af = 1.0f + l * 1e-3f, bf = 0.5f; TOUCH(af); TOUCH(bf);
  v4f d = {0, 0, 0, 0}, d2 = {1, 1, 1, 1}, d3 = {2, 2, 2, 2}, d4 = {3, 3, 3, 3};
#pragma unroll
  for (int i = 0; i < 64; i++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, d, 0, 0, 0);
  USE(d[0]); long long t3 = clk(); TOUCH(af);
#pragma unroll
  for (int i = 0; i < 16; i++) {
    d = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, d, 0, 0, 0);
    d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, d2, 0, 0, 0);
    d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, d3, 0, 0, 0);
    d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, d4, 0, 0, 0);
  }
  USE(d[0]); USE(d2[0]); USE(d3[0]); USE(d4[0]); long long t4 = clk();
  double x = a; TOUCH(x);
#pragma unroll
  for (int i = 0; i < 64; i++) x = __builtin_fma(x, 1.0000001, 0.5);
  USE(x); long long t5 = clk();
  double y = 0; int idx = l; TOUCH(idx);
#pragma unroll
  for (int i = 0; i < 64; i++) { y += lds[idx & 127]; idx = (int)y & 63; }
  USE(y); long long t6 = clk();
  double z = a; TOUCH(z);
#pragma unroll
  for (int i = 0; i < 64; i++) {
    int lo = __builtin_amdgcn_readlane(__double2loint(z), 3), hi = __builtin_amdgcn_readlane(__double2hiint(z), 3);
    z = z * 0.999 + __hiloint2double(hi, lo);
  }
  USE(z); long long t7 = clk();
  double w = a; TOUCH(w);
#pragma unroll
  for (int i = 0; i < 64; i++) w = w * 0.999 + __shfl(w, 3, 64);
  USE(w); long long t8 = clk();
  double q = a + 2; TOUCH(q);
#pragma unroll
  for (int i = 0; i < 32; i++) q = 1.0 / (q + 1.5);
  USE(q); long long t9 = clk();
  double sq = a + 2; TOUCH(sq);
#pragma unroll
  for (int i = 0; i < 32; i++) sq = sqrt(sq + 1.5);
  USE(sq); long long t10 = clk();
  float xf = af; TOUCH(xf);
#pragma unroll
  for (int i = 0; i < 64; i++) xf = __builtin_fmaf(xf, 1.0000001f, 0.5f);
  USE(xf); long long t11 = clk();
  if (l == 0) {
    out[0] = (t1 - t0) / 64; out[1] = (t2 - t1) / 64; out[2] = (t3 - t2) / 64; out[3] = (t4 - t3) / 64;
    out[4] = (t5 - t4) / 64; out[5] = (t6 - t5) / 64; out[6] = (t7 - t6) / 64; out[7] = (t8 - t7) / 64;
    out[8] = (t9 - t8) / 32; out[9] = (t10 - t9) / 32; out[10] = (t11 - t10) / 64;
  }
  sink[l] = c[0] + c2[1] + c3[2] + c4[3] + d[0] + d2[1] + d3[2] + d4[3] + x + y + z + w + q + sq + xf;
}
int main() {
  long long* o; double* s;
  hipMalloc(&o, 128); hipMalloc(&s, 64 * 8);
  for (int rep = 0; rep < 2; rep++) k<<<1, 64>>>(o, s);
  long long h[16];
  hipMemcpy(h, o, 128, hipMemcpyDeviceToHost);
  printf("cycles/op (1 wave): mfma_f64 dep %lld  mfma_f64 4-indep %lld | mfma_f32 dep %lld  4-indep %lld | fma_f64 dep %lld | lds dep-read %lld | readlane+fma %lld | shfl+fma %lld | div_f64 dep %lld | sqrt_f64 dep %lld | fma_f32 dep %lld\n",
         h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
  return 0;
}
