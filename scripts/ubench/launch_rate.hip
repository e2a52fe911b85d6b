// Micro-benchmark (diagnostic): how fast does the chip start workgroups?  Grids of G workgroups of W waves whose
// kernel does (almost) nothing, with and without LDS and with a kernel-argument block of the size the solver's
// kernels carry (DevProblem + SolveArgs by value), timed with HIP events over 20 launches.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/launch_rate.hip -o scripts/ubench/_bin/launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big { int w[480]; };  // ~1.9 KB of kernel arguments

__global__ void tiny(int* out) {
  extern __shared__ int lds[];
  if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = lds[0];
}
__global__ void tiny_big(Big b, int* out) {
  extern __shared__ int lds[];
  if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = lds[0] + b.w[threadIdx.x];
}
// a workgroup that stays ~`spin` clocks (one global round trip chain stands in for the decision kernel)
__global__ void chain(const int* in, int* out, int hops) {
  extern __shared__ int lds[];
  int v = blockIdx.x & 1023;
  for (int h = 0; h < hops; h++) v = in[v];
  if (threadIdx.x == 0) out[blockIdx.x] = v + lds[0] * 0;
}

// every workgroup ends with one returning atomic on the same word (the round counters of the solver's trial kernels)
__global__ void atom(int* ctr, int* out, int spread) {
  if (threadIdx.x == 0) out[blockIdx.x] = atomicAdd(ctr + (spread ? (blockIdx.x & 63) * 32 : 0), 1);
}
__global__ void atom_noret(int* ctr, int* out) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename F>
static float timed(F&& launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; i++) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 20; i++) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / 20.f;
}

int main() {
  int *d, *in;
  hipMalloc(&d, 1 << 20);
  hipMalloc(&in, 4096);
  int h[1024];
  for (int i = 0; i < 1024; i++) h[i] = (i * 37 + 11) & 1023;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  Big big{};
  for (int grid : {1024, 4096, 8192, 16384, 32768})
    for (int waves : {1, 2, 4})
      for (int lds : {0, 8192}) {
        const float t0 = timed([&] { hipLaunchKernelGGL(tiny, dim3(grid), dim3(64 * waves), lds, 0, d); });
        const float t1 = timed([&] { hipLaunchKernelGGL(tiny_big, dim3(grid), dim3(64 * waves), lds, 0, big, d); });
        const float t2 = timed([&] { hipLaunchKernelGGL(chain, dim3(grid), dim3(64 * waves), lds, 0, in, d, 12); });
        printf("grid %6d x %d waves, LDS %5d B: empty %7.1f us (%.1f ns / wave), 1.9 KB of arguments %7.1f us, 12-hop chain %7.1f us\n",
               grid, waves, lds, t0, 1000.f * t0 / (grid * waves), t1, t2);
      }
  int* ctr;
  hipMalloc(&ctr, 1 << 16);
  hipMemset(ctr, 0, 1 << 16);
  for (int grid : {1024, 8192, 16384}) {
    const float a0 = timed([&] { hipLaunchKernelGGL(atom, dim3(grid), dim3(64), 0, 0, ctr, d, 0); });
    const float a1 = timed([&] { hipLaunchKernelGGL(atom, dim3(grid), dim3(64), 0, 0, ctr, d, 1); });
    const float a2 = timed([&] { hipLaunchKernelGGL(atom_noret, dim3(grid), dim3(64), 0, 0, ctr, d); });
    printf("grid %6d: one returning atomic per workgroup on one word %7.1f us (%.1f ns each), on 64 words %7.1f us, not returning %7.1f us\n",
           grid, a0, 1000.f * a0 / grid, a1, a2);
  }
  return 0;
}
