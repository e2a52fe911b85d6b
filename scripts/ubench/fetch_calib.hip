// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (diagnostic): kernels that move a KNOWN number of bytes
// with the access shapes the solver uses — 16-byte and 8-byte per-lane global loads, LDS-DMA (global_load_lds) in
// 16-byte and 4-byte pieces, 8-byte and 16-byte stores.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
// (separate passes); scripts/calibrate_counters.py divides the counters by the byte counts printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

__global__ void read16(const float4* p, size_t n, float* sink) {
  float s = 0;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) *sink = s;
}
__global__ void read8(const double* p, size_t n, double* sink) {
  double s = 0;
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) s += p[i];
  if (s == 123.456) *sink = s;
}
template <int PIECE>
__global__ void dma(const char* p, size_t bytes, float* sink) {
  __shared__ __align__(16) char buf[256 * 16];
  const size_t per_block = size_t(blockDim.x) * PIECE;
  for (size_t off = blockIdx.x * per_block; off + per_block <= bytes; off += size_t(gridDim.x) * per_block) {
    const char* src = p + off + size_t(threadIdx.x) * PIECE;
    char* dst = buf + (threadIdx.x & ~63) * PIECE;  // wave-uniform; the hardware adds lane * PIECE
    if constexpr (PIECE == 16)
      __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)dst, 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)dst, 4, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (buf[threadIdx.x] == 77 && buf[threadIdx.x + 1] == 78 && threadIdx.x == 999) *sink = 1.0f;
}
__global__ void write16(float4* p, size_t n) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    p[i] = make_float4(1.f, 2.f, 3.f, float(i));
}
__global__ void write8(double* p, size_t n) {
  for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) p[i] = double(i);
}

int main() {
  const size_t bytes = size_t(1) << 30;  // 1 GiB: far beyond L2 (32 MiB) and the Infinity Cache (256 MiB)
  char* a;
  float* sink;
  hipMalloc(&a, bytes);
  hipMalloc(&sink, 64);
  hipMemset(a, 1, bytes);
  hipDeviceSynchronize();
  const int grid = 256 * 8, block = 256;
  hipLaunchKernelGGL(read16, dim3(grid), dim3(block), 0, 0, (const float4*)a, bytes / 16, sink);
  hipLaunchKernelGGL(read8, dim3(grid), dim3(block), 0, 0, (const double*)a, bytes / 8, (double*)sink);
  hipLaunchKernelGGL(dma<16>, dim3(grid), dim3(block), 0, 0, a, bytes, sink);
  hipLaunchKernelGGL(dma<4>, dim3(grid), dim3(block), 0, 0, a, bytes, sink);
  hipLaunchKernelGGL(write16, dim3(grid), dim3(block), 0, 0, (float4*)a, bytes / 16);
  hipLaunchKernelGGL(write8, dim3(grid), dim3(block), 0, 0, (double*)a, bytes / 8);
  hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", bytes);
  return 0;
}
