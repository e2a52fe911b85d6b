// Where does an LDS-DMA land when only the upper half of the wave is active?  (diagnostic for rollout_pair)
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/ldsdma_half.hip -o /tmp/ldsdma_half && /tmp/ldsdma_half
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const __attribute__((address_space(1))) void* glb_vptr;
typedef __attribute__((address_space(3))) void* lds_vptr;
__global__ void k(const int* src, int* out, int mode) {
  __shared__ int sm[256];
  const int t = threadIdx.x;
  for (int i = t; i < 256; i += 64) sm[i] = -1;
  __syncthreads();
  if (mode == 0) {
    __builtin_amdgcn_global_load_lds((glb_vptr)(src + t), (lds_vptr)(sm + 64), 4, 0, 0);
  } else if (t >= 32) {  // upper half only: lane L's dword expected at base + 4 L
    __builtin_amdgcn_global_load_lds((glb_vptr)(src + (t - 32)), (lds_vptr)(sm + 64), 4, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = t; i < 256; i += 64) out[i] = sm[i];
}
int main() {
  int h[64], *d, *o, r[256];
  for (int i = 0; i < 64; i++) h[i] = 1000 + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, mode);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("mode %d:", mode);
    for (int i = 0; i < 256; i++) if (r[i] != -1) printf(" [%d]=%d", i, r[i]);
    printf("\n");
  }
  return 0;
}
