// ilqg_mfma.hpp — 16x16 tile algebra on the gfx950 matrix cores for the LQ sweep.
//
// One wavefront holds a 16x16 matrix X as 4 scalars per lane in the accumulator ("D") layout of
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32:
//     lane l = 16*g + j  (g = l>>4, j = l&15),  register r  <->  X[row(g,r)][j]
//     row(g,r) = g + 4r  (f64)        row(g,r) = 4g + r  (f32)
// The identities the sweep is built on (K is traversed in 4 blocks kb = 0..3, lane group g
// supplying k = row(g,kb) — for f32 that is a permutation of the usual k order, which a sum over
// k does not care about):
//     * register kb of the D layout of X   is the B operand of block kb of  (.) * X
//     * register kb of the D layout of X   is the A operand of block kb of  X^T * (.)
// so with Yd = D-layout(Z^T), Fd = D-layout(F):
//     Wd  = sum_kb mfma(Yd[kb], Fd[kb])        = D-layout(Z F)
//     Yd' = sum_kb mfma(Wd[kb], Fd[kb]) + Cd   = D-layout((Z F)^T F + C) = D-layout((F^T Z F)^T + C)
// i.e. the whole F^T Z F recursion runs register-to-register, no LDS traffic and no re-layout.
// f32-input MFMA is an exact fp32 FMA chain (no reduced precision), f64 likewise.
#pragma once

#include "ilqg_common.hpp"

namespace ilqg {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <typename T> struct Tile;
template <> struct Tile<double> {
  using vec = v4d;
  static constexpr __host__ __device__ __forceinline__ int row(int g, int r) { return g + 4 * r; }
  static __device__ __forceinline__ vec mfma(double a, double b, vec c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
};
template <> struct Tile<float> {
  using vec = v4f;
  static constexpr __host__ __device__ __forceinline__ int row(int g, int r) { return 4 * g + r; }
  static __device__ __forceinline__ vec mfma(float a, float b, vec c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// D-layout(X^T * Y) from D-layout(X), D-layout(Y), plus accumulator C (D layout).
template <typename T>
__device__ __forceinline__ typename Tile<T>::vec tile_xty(const typename Tile<T>::vec& xd,
                                                          const typename Tile<T>::vec& yd,
                                                          typename Tile<T>::vec c) {
#pragma unroll
  for (int kb = 0; kb < 4; kb++) c = Tile<T>::mfma(xd[kb], yd[kb], c);
  return c;
}

// Which of the four k blocks of a tile_xty product can be non-zero when the LEFT operand X has non-zero rows only
// in [row_begin, row_end): register kb of the D layout holds rows {g + 4 kb} (f64) / {4 g + kb} (f32), g = 0..3.
template <typename T>
constexpr int kblock_mask(int row_begin, int row_end) {
  int mask = 0;
  for (int kb = 0; kb < 4; kb++)
    for (int g = 0; g < 4; g++) {
      const int row = Tile<T>::row(g, kb);
      if (row >= row_begin && row < row_end) mask |= 1 << kb;
    }
  return mask;
}

// tile_xty restricted to the k blocks in MASK (the others multiply rows of X that are identically zero).  fp64
// MFMA issues at its latency (64 cycles per 16x16x4, no overlap between independent ones — scripts/ubench), so every
// skipped block is 64 cycles of a SIMD's matrix pipe that the other resident waves get back.
template <typename T, int MASK>
__device__ __forceinline__ typename Tile<T>::vec tile_xty_blocks(const typename Tile<T>::vec& xd,
                                                                 const typename Tile<T>::vec& yd,
                                                                 typename Tile<T>::vec c) {
#pragma unroll
  for (int kb = 0; kb < 4; kb++)
    if (MASK & (1 << kb)) c = Tile<T>::mfma(xd[kb], yd[kb], c);
  return c;
}

// Self-test kernel body: given 16x16 column-major X, Y, C in global memory computes
// out = X^T * Y + C through the D-layout path (one wavefront).
template <typename T>
__device__ void mfma_selftest(const T* X, const T* Y, const T* C, T* out) {
  const int l = threadIdx.x & 63, g = l >> 4, j = l & 15;
  typename Tile<T>::vec xd, yd, cd;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = Tile<T>::row(g, r);
    xd[r] = X[row + 16 * j];
    yd[r] = Y[row + 16 * j];
    cd[r] = C[row + 16 * j];
  }
  const typename Tile<T>::vec d = tile_xty<T>(xd, yd, cd);
#pragma unroll
  for (int r = 0; r < 4; r++) out[Tile<T>::row(g, r) + 16 * j] = d[r];
}

}  // namespace ilqg
