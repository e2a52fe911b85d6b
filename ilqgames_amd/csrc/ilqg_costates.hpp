// ilqg_costates.hpp — costates of the feedback LQ solution (LQFeedbackSolver::Solve, src/lq_feedback_solver.cpp:216-228):
//     costate_i,k = -Z_i[k+1] delta_x_k - zeta_i[k+1]   (k < T-1),   0 at k = T-1.
// The reference keeps Z_i, zeta_i of every step for this and nothing else (lq_feedback_solver.h:100-110); the device
// sweeps keep them in registers one step at a time.  ILQSolver ignores the costates (ilq_solver.cpp:382-385), so they
// are produced off the hot path: a second kernel re-runs the value recursion (:186-212) from the strategies the sweep
// has already written,
//     F = A - sum_j B_j P_j,  beta = -sum_j B_j alpha_j
//     zeta_i <- F^T (zeta_i + Z_i beta) + l_i + sum_j P_j^T (R_ij alpha_j - r_ij)
//     Z_i    <- F^T Z_i F + Q_i + sum_j P_j^T R_ij P_j
// (no linear system: P, alpha are inputs here), leaves Z_i, zeta_i of every step in a scratch buffer and then forms the
// costates from the sweep's delta_xs.  Run-time dimensions; one workgroup per instance.
#pragma once

#include "ilqg_common.hpp"

namespace ilqg {

struct CostateDims {
  int n, N, m, T;
  int uoff[kMaxPlayers + 1];
};

__host__ __device__ inline size_t costates_scratch_elems(int n, int N, int T) { return size_t(T) * N * (n * n + n); }
__host__ __device__ inline size_t costates_lds_elems(int n, int N) { return size_t(2 * N + 1) * n * n + size_t(2 * N + 1) * n; }

template <typename T>
__global__ void __launch_bounds__(256)
lq_feedback_costates_kernel(CostateDims d, PairTable pt, const T* A, const T* Bm, const T* Q, const T* l, const T* R,
                            const T* r, const T* P, const T* alpha, const T* dx, T* zs, T* costates) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const int n = d.n, N = d.N, m = d.m, Tn = d.T, nn = n * n;
  const int t = threadIdx.x, NT = blockDim.x;
  const size_t b = blockIdx.x;
  A += b * Tn * nn; Bm += b * Tn * n * m; Q += b * Tn * N * nn; l += b * Tn * N * n;
  R += b * Tn * pt.Rsz; r += b * Tn * pt.rsz; P += b * Tn * m * n; alpha += b * Tn * m; dx += b * Tn * n;
  zs += b * costates_scratch_elems(n, N, Tn);
  costates += b * Tn * N * n;
  T* sF = sm;                 // n x n
  T* sZ = sF + nn;            // N x (n x n)
  T* sU = sZ + N * nn;        // N x (n x n):  Z_i F
  T* sBeta = sU + N * nn;     // n
  T* sZeta = sBeta + n;       // N x n
  T* sZt = sZeta + N * n;     // N x n:  zeta_i + Z_i beta
  auto zrow = [&](int k) { return zs + size_t(k) * N * (nn + n); };
  // terminal step (:102-105)
  for (int e = t; e < N * nn; e += NT) sZ[e] = Q[size_t(Tn - 1) * N * nn + e];
  for (int e = t; e < N * n; e += NT) sZeta[e] = l[size_t(Tn - 1) * N * n + e];
  __syncthreads();
  for (int e = t; e < N * nn; e += NT) zrow(Tn - 1)[e] = sZ[e];
  for (int e = t; e < N * n; e += NT) zrow(Tn - 1)[N * nn + e] = sZeta[e];
  for (int k = Tn - 2; k >= 0; k--) {
    const T* Ak = A + size_t(k) * nn;
    const T* Bk = Bm + size_t(k) * n * m;
    const T* Pk = P + size_t(k) * m * n;
    const T* ak = alpha + size_t(k) * m;
    const T* Rk = R + size_t(k) * pt.Rsz;
    const T* rk = r + size_t(k) * pt.rsz;
    for (int e = t; e < nn; e += NT) {
      const int rr = e % n, c = e / n;
      T s = Ak[e];
      for (int q = 0; q < m; q++) s -= Bk[rr + n * q] * Pk[q + m * c];
      sF[e] = s;
    }
    for (int rr = t; rr < n; rr += NT) {
      T s = T(0);
      for (int q = 0; q < m; q++) s -= Bk[rr + n * q] * ak[q];
      sBeta[rr] = s;
    }
    __syncthreads();
    for (int e = t; e < N * nn; e += NT) {
      const int i = e / nn, rr = (e % nn) % n, c = (e % nn) / n;
      T s = T(0);
      for (int cc = 0; cc < n; cc++) s += sZ[i * nn + rr + n * cc] * sF[cc + n * c];
      sU[e] = s;
    }
    for (int e = t; e < N * n; e += NT) {
      const int i = e / n, rr = e % n;
      T s = sZeta[e];
      for (int cc = 0; cc < n; cc++) s += sZ[i * nn + rr + n * cc] * sBeta[cc];
      sZt[e] = s;
    }
    __syncthreads();
    for (int e = t; e < N * nn; e += NT) {
      const int i = e / nn, rr = (e % nn) % n, c = (e % nn) / n;
      T s = Q[(size_t(k) * N + i) * nn + rr + n * c];
      for (int kk = 0; kk < n; kk++) s += sF[kk + n * rr] * sU[i * nn + kk + n * c];
      for (int pe = 0; pe < pt.npairs; pe++) {
        if (pt.pi[pe] != i) continue;
        const int jj = pt.pj[pe], mu = d.uoff[jj + 1] - d.uoff[jj], u0 = d.uoff[jj];
        const T* Rij = Rk + pt.roff[pe];
        for (int aa = 0; aa < mu; aa++) {
          T h = T(0);  // (R_ij P_j)[aa][c]
          for (int bb = 0; bb < mu; bb++) h += Rij[aa + mu * bb] * Pk[(u0 + bb) + m * c];
          s += Pk[(u0 + aa) + m * rr] * h;
        }
      }
      sZ[e] = s;  // Z_i is not read in this phase (U_i = Z_i F is)
    }
    for (int e = t; e < N * n; e += NT) {
      const int i = e / n, rr = e % n;
      T s = l[(size_t(k) * N + i) * n + rr];
      for (int kk = 0; kk < n; kk++) s += sF[kk + n * rr] * sZt[i * n + kk];
      for (int pe = 0; pe < pt.npairs; pe++) {
        if (pt.pi[pe] != i) continue;
        const int jj = pt.pj[pe], mu = d.uoff[jj + 1] - d.uoff[jj], u0 = d.uoff[jj];
        const T* Rij = Rk + pt.roff[pe];
        const T* rij = rk + pt.rgoff[pe];
        for (int aa = 0; aa < mu; aa++) {
          T ww = -rij[aa];
          for (int bb = 0; bb < mu; bb++) ww += Rij[aa + mu * bb] * ak[u0 + bb];
          s += Pk[(u0 + aa) + m * rr] * ww;
        }
      }
      sZeta[e] = s;
    }
    __syncthreads();
    for (int e = t; e < N * nn; e += NT) zrow(k)[e] = sZ[e];
    for (int e = t; e < N * n; e += NT) zrow(k)[N * nn + e] = sZeta[e];
  }
  __syncthreads();  // the rows above were written by other threads of this workgroup
  for (int e = t; e < Tn * N * n; e += NT) {
    const int k = e / (N * n), i = (e / n) % N, rr = e % n;
    T s = T(0);
    if (k < Tn - 1) {
      const T* zr = zrow(k + 1);
      s = -zr[N * nn + i * n + rr];
      for (int c = 0; c < n; c++) s -= zr[i * nn + rr + n * c] * dx[size_t(k) * n + c];
    }
    costates[e] = s;
  }
}

}  // namespace ilqg
