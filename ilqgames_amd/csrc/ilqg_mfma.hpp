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

// The same with a run-time (wave-uniform) mask on top of the compile-time one: block structure known per problem, not per
// instantiation (a scalar branch around a 64-cycle instruction).
template <typename T, int MASK>
__device__ __forceinline__ typename Tile<T>::vec tile_xty_blocks_rt(const typename Tile<T>::vec& xd,
                                                                    const typename Tile<T>::vec& yd,
                                                                    typename Tile<T>::vec c, int rt_mask) {
#pragma unroll
  for (int kb = 0; kb < 4; kb++)
    if ((MASK & (1 << kb)) && (rt_mask & (1 << kb))) c = Tile<T>::mfma(xd[kb], yd[kb], c);
  return c;
}
// k blocks of a 16-row k tile (rows 16 c ..) that hold a row in [lo, hi)
template <typename T>
__host__ __device__ inline int kblock_mask_rt(int c, int lo, int hi) {
  int mask = 0;
  for (int kb = 0; kb < 4; kb++)
    for (int g = 0; g < 4; g++) {
      const int row = 16 * c + Tile<T>::row(g, kb);
      if (row >= lo && row < hi) mask |= 1 << kb;
    }
  return mask;
}

// ---- cross-row exchanges of a wavefront's four 16-lane rows on the vector unit (gfx950: v_permlane32_swap /
// v_permlane16_swap; no LDS) ----
// permlane32_swap(a, b): a' = [a.row0 a.row1 b.row0 b.row1], b' = [a.row2 a.row3 b.row2 b.row3]
// permlane16_swap(a, b): a' = [a.row0 b.row0 a.row2 b.row2], b' = [a.row1 b.row1 a.row3 b.row3]
// (lane maps printed by scripts/ubench/lds_probe.hip)
__device__ __forceinline__ void permlane32_swap(float a, float b, float& ra, float& rb) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  ra = __uint_as_float(r[0]);
  rb = __uint_as_float(r[1]);
}
__device__ __forceinline__ void permlane16_swap(float a, float b, float& ra, float& rb) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  ra = __uint_as_float(r[0]);
  rb = __uint_as_float(r[1]);
}
__device__ __forceinline__ void permlane32_swap(double a, double b, double& ra, double& rb) {
  const auto lo = __builtin_amdgcn_permlane32_swap(unsigned(__double2loint(a)), unsigned(__double2loint(b)), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(unsigned(__double2hiint(a)), unsigned(__double2hiint(b)), false, false);
  ra = __hiloint2double(int(hi[0]), int(lo[0]));
  rb = __hiloint2double(int(hi[1]), int(lo[1]));
}
__device__ __forceinline__ void permlane16_swap(double a, double b, double& ra, double& rb) {
  const auto lo = __builtin_amdgcn_permlane16_swap(unsigned(__double2loint(a)), unsigned(__double2loint(b)), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(unsigned(__double2hiint(a)), unsigned(__double2hiint(b)), false, false);
  ra = __hiloint2double(int(hi[0]), int(lo[0]));
  rb = __hiloint2double(int(hi[1]), int(lo[1]));
}
// Sum over the four rows of every lane column: all rows return (x.row0 + x.row2) + (x.row1 + x.row3).
template <typename T>
__device__ __forceinline__ T rows_allreduce(T x) {
  T a, b;
  permlane32_swap(x, x, a, b);
  const T s = a + b;  // [x0 + x2, x1 + x3, x0 + x2, x1 + x3]
  permlane16_swap(s, s, a, b);
  return a + b;
}
// Four per-lane partial sums p0 .. p3 (each to be summed over the four rows) -> one register whose row q holds the
// total of p_q', q' = (q >> 1) + 2 * (q & 1)... see the body: row 0 = sum p0, row 1 = sum p2, row 2 = sum p1, row 3 = sum p3.
template <typename T>
__device__ __forceinline__ T rows_reduce4(T p0, T p1, T p2, T p3) {
  T a, b;
  permlane32_swap(p0, p1, a, b);
  const T t = a + b;  // rows: p0(0+2), p0(1+3), p1(0+2), p1(1+3)
  permlane32_swap(p2, p3, a, b);
  const T u = a + b;  // rows: p2(0+2), p2(1+3), p3(0+2), p3(1+3)
  permlane16_swap(t, u, a, b);
  return a + b;  // rows: p0, p2, p1, p3
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
