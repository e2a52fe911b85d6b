// Streaming-copy bandwidth of the box with a few kernel shapes (diagnostic; picks the shape of ilqg_copy_bandwidth).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/copy_bw.hip -o scripts/ubench/_bin/copy_bw && scripts/ubench/_bin/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int UN, bool NT>
__global__ void __launch_bounds__(256) copy_k(v4f* __restrict__ dst, const v4f* __restrict__ src, size_t n16) {
  const size_t stride = size_t(gridDim.x) * 256;
  size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  for (; i + (UN - 1) * stride < n16; i += UN * stride) {
    v4f v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
// contiguous chunk per workgroup
template <int UN>
__global__ void __launch_bounds__(256) copy_chunk(v4f* __restrict__ dst, const v4f* __restrict__ src, size_t n16) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t b0 = size_t(blockIdx.x) * per, b1 = b0 + per < n16 ? b0 + per : n16;
  size_t i = b0 + threadIdx.x;
  for (; i + (UN - 1) * 256 < b1; i += UN * 256) {
    v4f v[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) v[u] = src[i + u * 256];
#pragma unroll
    for (int u = 0; u < UN; u++) dst[i + u * 256] = v[u];
  }
  for (; i < b1; i += 256) dst[i] = src[i];
}

template <class F>
static double time_it(F launch, size_t bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 10; r++) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return 2.0 * bytes / (best * 1e-3) / 1e9;
}

int main() {
  const size_t bytes = size_t(1) << 30, n16 = bytes / 16;
  v4f *src, *dst;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
  hipMemset(src, 1, bytes);
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
    printf("blocks %6d: gs4 %.0f  gs4nt %.0f  gs8 %.0f  gs2 %.0f  chunk4 %.0f  chunk8 %.0f GB/s\n", blocks,
           time_it([&] { hipLaunchKernelGGL((copy_k<4, false>), dim3(blocks), dim3(256), 0, 0, dst, src, n16); }, bytes),
           time_it([&] { hipLaunchKernelGGL((copy_k<4, true>), dim3(blocks), dim3(256), 0, 0, dst, src, n16); }, bytes),
           time_it([&] { hipLaunchKernelGGL((copy_k<8, false>), dim3(blocks), dim3(256), 0, 0, dst, src, n16); }, bytes),
           time_it([&] { hipLaunchKernelGGL((copy_k<2, false>), dim3(blocks), dim3(256), 0, 0, dst, src, n16); }, bytes),
           time_it([&] { hipLaunchKernelGGL((copy_chunk<4>), dim3(blocks), dim3(256), 0, 0, dst, src, n16); }, bytes),
           time_it([&] { hipLaunchKernelGGL((copy_chunk<8>), dim3(blocks), dim3(256), 0, 0, dst, src, n16); }, bytes));
  }
  printf("hipMemcpyDtoD: %.0f GB/s\n", time_it([&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); }, bytes));
  return 0;
}
