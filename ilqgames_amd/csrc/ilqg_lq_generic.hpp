// ilqg_lq_generic.hpp — the two LQ Nash sweeps with RUN-TIME dimensions (any n <= 32, N <= 8, any m_i, sum m_i <= 16).
//
// The specialised sweeps (ilqg_lq.hpp, ilqg_lq_feedback2.hpp, ilqg_lq_openloop.hpp) are compiled per (n, N, m_i) with one
// control dimension for all players; a game outside that list — players with different control dimensions, a state
// dimension nobody instantiated — runs here instead of being refused.  Same recursions, restated from
//   LQFeedbackSolver::Solve  src/lq_feedback_solver.cpp:71-244  (Gershgorin step :163-176, Householder QR :180)
//   LQOpenLoopSolver::Solve  src/lq_open_loop_solver.cpp:73-195
//   ILQSolver::ExpectedDecrease  src/ilq_solver.cpp:364-398
// with every matrix in LDS and plain loops: one workgroup per game instance, every phase a parallel-for over output
// entries followed by a workgroup barrier.  This is the correctness path, not the fast one.
//
// The file has no HIP dependency beyond the __host__ __device__ markers: the phases are written against an executor
// (`Par`) whose device form strides the entries over the workgroup's threads and ends each phase with __syncthreads(),
// and whose host form (tests/host/generic_lq_check.cpp) runs them one after the other — so the arithmetic and the
// indexing of these kernels are checked on the CPU, without a GPU, by the `-m "not gpu"` tests.
#pragma once

#include <math.h>
#include <stddef.h>

#include "ilqg_pairs.hpp"

#if defined(__HIPCC__)
#define ILQG_HD __host__ __device__ __forceinline__
#else
#define ILQG_HD inline
#endif

namespace ilqg {

struct GenDims {
  int n, N, m, T;
  int udim[kMaxPlayers], uoff[kMaxPlayers + 1];
};

template <typename T>
struct GenLQArgs {
  const T *A, *Bm, *Q, *l, *R, *r;  // instance bases: [T][n*n], [T][n*m], [T][N][n*n], [T][N][n], [T][Rsz], [T][rsz]
  const T* x0;                      // [n] or nullptr (zero)
  T *P, *alpha;                     // [T][m*n], [T][m]
  T* dx;                            // [T][n] or nullptr
  T* costates;                      // open loop only: [T][N][n] or nullptr (needs dx and the fat scratch rows)
  T* scratch;                       // open loop: [T][gen_ol_row_elems(.., costates != nullptr)]; feedback: unused
  T* ed_out;                        // ILQSolver::ExpectedDecrease (one scalar) or nullptr
  int adaptive;                     // feedback: the Gershgorin step
};

// ---- executors -------------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
// Device: entry e of a phase goes to thread e mod blockDim; a phase ends with a workgroup barrier.
struct ParDevice {
  template <typename F>
  __device__ __forceinline__ void operator()(int count, F&& f) const {
    for (int e = threadIdx.x; e < count; e += blockDim.x) f(e);
    __syncthreads();
  }
};
#endif
// Host (logic test): the entries of a phase one after the other.  A phase whose entries depended on each other would
// pass here and race on the device, so phases only ever read what EARLIER phases wrote (or their own outputs).
struct ParHost {
  template <typename F>
  void operator()(int count, F&& f) const {
    for (int e = 0; e < count; e++) f(e);
  }
};

ILQG_HD int gen_player_of(const GenDims& d, int row) {  // the player whose control rows hold `row`
  int i = 0;
  while (i + 1 < d.N && row >= d.uoff[i + 1]) i++;
  return i;
}
template <typename T>
ILQG_HD T gen_abs(T v) { return v < T(0) ? -v : v; }
ILQG_HD float gen_sqrt(float v) { return sqrtf(v); }
ILQG_HD double gen_sqrt(double v) { return sqrt(v); }
template <typename T>
ILQG_HD T gen_tiny() { return sizeof(T) == 4 ? T(1.17549435e-38f) : T(2.2250738585072014e-308); }

// Solves S X = Y for the M x M matrix S and NR right-hand sides stored behind it, [S | Y] column-major with leading
// dimension M at `sy` (LDS): Householder QR as `S.householderQr().solve(Y)` does it (src/lq_feedback_solver.cpp:180,
// src/lq_open_loop_solver.cpp:131,144,148 — Eigen's unblocked householder_qr_inplace: makeHouseholder,
// applyHouseholderOnTheLeft, then the triangular solve).  X replaces Y.  `hv`: 2 scratch elements.
template <typename T, typename Par>
ILQG_HD void gen_qr_solve(T* sy, int M, int NR, T* hv, const Par& par) {
  const int ncols = M + NR;
  for (int k = 0; k < M; k++) {
    par(1, [&](int) {  // the reflector of column k, in place: S[k][k] = beta, the essential part below it
      T* col = sy + size_t(M) * k;
      T tailsq = T(0);
      for (int i = k + 1; i < M; i++) tailsq += col[i] * col[i];
      const T c0 = col[k];
      T tau, beta;
      if (M - k == 1 || tailsq <= gen_tiny<T>()) {
        tau = T(0);
        beta = c0;
        for (int i = k + 1; i < M; i++) col[i] = T(0);
      } else {
        beta = gen_sqrt(c0 * c0 + tailsq);
        if (c0 >= T(0)) beta = -beta;
        for (int i = k + 1; i < M; i++) col[i] = col[i] / (c0 - beta);
        tau = (beta - c0) / beta;
      }
      col[k] = beta;
      hv[0] = tau;
    });
    par(ncols - k - 1, [&](int e) {  // H_k applied to every later column (each entry owns one column)
      const T tau = hv[0];
      if (tau == T(0)) return;
      const T* v = sy + size_t(M) * k;
      T* col = sy + size_t(M) * (k + 1 + e);
      T tmp = T(0);
      for (int i = k + 1; i < M; i++) tmp += v[i] * col[i];
      tmp += col[k];
      col[k] -= tau * tmp;
      for (int i = k + 1; i < M; i++) col[i] -= tau * v[i] * tmp;
    });
  }
  par(NR, [&](int e) {  // back substitution, one right-hand side per entry
    T* x = sy + size_t(M) * (M + e);
    for (int i = M - 1; i >= 0; i--) {
      T s = x[i];
      for (int k2 = i + 1; k2 < M; k2++) s -= sy[i + size_t(M) * k2] * x[k2];
      x[i] = s / sy[i + size_t(M) * i];
    }
  });
}

// ExpectedDecrease terms of step k (src/ilq_solver.cpp:376-395) into `terms` [2 N]: alpha_i^T R_ii r_ii evaluated as
// (alpha^T R) r like Eigen, and dx^T (Q_i l_i).  Entry i of the phase belongs to player i.
template <typename T>
ILQG_HD void gen_ed_terms(const GenDims& d, const GenLQArgs<T>& a, const PairTable& pt, int k, const T* alpha_k,
                          const T* x, int i, T* terms) {
  const int n = d.n, mi = d.udim[i], q = pt.pii[i];
  const T* Rii = a.R + size_t(k) * pt.Rsz + pt.roff[q];
  const T* rii = a.r + size_t(k) * pt.rsz + pt.rgoff[q];
  T ct = T(0);
  for (int c = 0; c < mi; c++) {
    T aR = T(0);
    for (int b = 0; b < mi; b++) aR += alpha_k[d.uoff[i] + b] * Rii[b + mi * c];
    ct += aR * rii[c];
  }
  const T* Qi = a.Q + (size_t(k) * d.N + i) * n * n;
  const T* li = a.l + (size_t(k) * d.N + i) * n;
  T st = T(0);
  for (int c = 0; c < n; c++) {  // x^T (Q_i l_i)
    T ql = T(0);
    for (int e = 0; e < n; e++) ql += Qi[c + n * e] * li[e];
    st += x[c] * ql;
  }
  terms[2 * i] = ct;
  terms[2 * i + 1] = st;
}

// The same with Q_i l_i of the step already formed (`ql` [n], the sums over e in the order above) — the forward passes
// form them in a phase of their own, one entry per (player, row), instead of n^2 dependent global reads on one thread.
template <typename T>
ILQG_HD void gen_ed_terms_ql(const GenDims& d, const GenLQArgs<T>& a, const PairTable& pt, int k, const T* alpha_k,
                             const T* x, int i, const T* ql, T* terms) {
  const int n = d.n, mi = d.udim[i], q = pt.pii[i];
  const T* Rii = a.R + size_t(k) * pt.Rsz + pt.roff[q];
  const T* rii = a.r + size_t(k) * pt.rsz + pt.rgoff[q];
  T ct = T(0);
  for (int c = 0; c < mi; c++) {
    T aR = T(0);
    for (int b = 0; b < mi; b++) aR += alpha_k[d.uoff[i] + b] * Rii[b + mi * c];
    ct += aR * rii[c];
  }
  T st = T(0);
  for (int c = 0; c < n; c++) st += x[c] * ql[c];
  terms[2 * i] = ct;
  terms[2 * i + 1] = st;
}
template <typename T>
ILQG_HD T gen_ql_entry(const GenDims& d, const GenLQArgs<T>& a, int k, int i, int c) {  // (Q_i l_i)[c] of step k
  const int n = d.n;
  const T* Qi = a.Q + (size_t(k) * d.N + i) * n * n;
  const T* li = a.l + (size_t(k) * d.N + i) * n;
  T ql = T(0);
  for (int e = 0; e < n; e++) ql += Qi[c + n * e] * li[e];
  return ql;
}

// LDS elements of the feedback sweep: [Z (N n^2) | zeta (N n) | BZ (m n) | SY (m (m + n + 1)) | P (m n) | alpha (m) |
// F (n^2) | beta (n) | U (n^2) | tz (n) | yz (m) | x (2 n) | terms (2 N) | hv (4)]
// ... | sA (n^2) | sB (n m) | sR (<= N m^2) | sr (<= N m)]: the step's A, B, R, r, staged once per step (round 6: the
// phases read them n or m_j^2 times per entry, and read from global memory every one of those was a dependent
// global-memory round trip inside a dot product)
ILQG_HD size_t gen_feedback_lds_elems(int n, int N, int m) {
  return size_t(N) * n * n + size_t(N) * n + size_t(m) * n + size_t(m) * (m + n + 1) + size_t(m) * n + m + size_t(n) * n +
         n + size_t(n) * n + n + m + 2 * size_t(n) + 2 * size_t(N) + 4 +
         size_t(n) * n + size_t(n) * m + size_t(N) * m * m + size_t(N) * m;
}

template <typename T, typename Par>
ILQG_HD void lq_feedback_generic(const GenDims& d, const GenLQArgs<T>& a, const PairTable& pt, T* sm, const Par& par) {
  const int n = d.n, N = d.N, m = d.m, Tn = d.T;
  T* Z = sm;
  T* zeta = Z + size_t(N) * n * n;
  T* BZ = zeta + size_t(N) * n;
  T* SY = BZ + size_t(m) * n;
  T* sP = SY + size_t(m) * (m + n + 1);
  T* sAl = sP + size_t(m) * n;
  T* F = sAl + m;
  T* beta = F + size_t(n) * n;
  T* U = beta + n;
  T* tz = U + size_t(n) * n;
  T* yz = tz + n;
  T* xb = yz + m;  // two buffers of n for the forward pass
  T* terms = xb + 2 * n;
  T* hv = terms + 2 * N;
  T* sA = hv + 4;
  T* sB = sA + size_t(n) * n;
  T* sR = sB + size_t(n) * m;
  T* sr = sR + size_t(N) * m * m;

  // ---- terminal step: Z_i = Q_i[T-1], zeta_i = l_i[T-1]; strategies at T-1 stay zero (:102-105, strategy.h:64-70) ----
  par(N * n * n, [&](int e) { Z[e] = a.Q[size_t(Tn - 1) * N * n * n + e]; });
  par(N * n, [&](int e) { zeta[e] = a.l[size_t(Tn - 1) * N * n + e]; });
  par(m * n + m, [&](int e) {
    if (e < m * n)
      a.P[size_t(Tn - 1) * m * n + e] = T(0);
    else
      a.alpha[size_t(Tn - 1) * m + (e - m * n)] = T(0);
  });

  for (int k = Tn - 2; k >= 0; k--) {
    const T* Q = a.Q + size_t(k) * N * n * n;
    const T* l = a.l + size_t(k) * N * n;
    {  // the step's A, B, R, r into LDS: one coalesced pass, one exposed global round trip
      const T* gA = a.A + size_t(k) * n * n;
      const T* gB = a.Bm + size_t(k) * n * m;
      const T* gR = a.R + size_t(k) * pt.Rsz;
      const T* gr = a.r + size_t(k) * pt.rsz;
      const int cA = n * n, cB = n * m, cR = pt.Rsz, cr = pt.rsz;
      par(cA + cB + cR + cr, [&](int e) {
        if (e < cA) sA[e] = gA[e];
        else if (e < cA + cB) sB[e - cA] = gB[e - cA];
        else if (e < cA + cB + cR) sR[e - cA - cB] = gR[e - cA - cB];
        else sr[e - cA - cB - cR] = gr[e - cA - cB - cR];
      });
    }
    const T *A = sA, *B = sB, *R = sR, *r = sr;
    // ---- B_i^T Z_i (the rows of the stacked system) and B_i^T zeta_i + r_ii (:128, :154-157) ----
    par(m * n + m, [&](int e) {
      if (e < m * n) {
        const int row = e % m, c = e / m, i = gen_player_of(d, row);
        const T* Zi = Z + size_t(i) * n * n;
        T s = T(0);
        for (int q = 0; q < n; q++) s += B[q + n * row] * Zi[q + n * c];
        BZ[row + m * c] = s;
      } else {
        const int row = e - m * n, i = gen_player_of(d, row);
        T s = T(0);
        for (int q = 0; q < n; q++) s += B[q + n * row] * zeta[i * n + q];
        yz[row] = s + r[pt.rgoff[pt.pii[i]] + (row - d.uoff[i])];
      }
    });
    // ---- [S | Y]: S = BZ B (+ R_ii on the diagonal blocks), Y = [BZ A | y_zeta]  (:131-157) ----
    par(m * (m + n + 1), [&](int e) {
      const int row = e % m, t = e / m;
      if (t == m + n) {
        SY[e] = yz[row];
        return;
      }
      const T* colp = t < m ? B + size_t(n) * t : A + size_t(n) * (t - m);
      T s = T(0);
      for (int c = 0; c < n; c++) s += BZ[row + m * c] * colp[c];
      if (t < m) {
        const int i = gen_player_of(d, row);
        if (gen_player_of(d, t) == i) s += R[pt.roff[pt.pii[i]] + (row - d.uoff[i]) + d.udim[i] * (t - d.uoff[i])];
      }
      SY[e] = s;
    });
    // ---- Gershgorin regularisation, column by column (:163-176; a column's test reads that column only) ----
    if (a.adaptive)
      par(m, [&](int c) {
        T* col = SY + size_t(m) * c;
        T l1 = T(0);
        for (int row = 0; row < m; row++) l1 += gen_abs(col[row]);
        const T radius = l1 - gen_abs(col[c]);
        if (col[c] - radius < T(1e-3f)) col[c] += radius + T(1e-3f);
      });
    gen_qr_solve<T>(SY, m, n + 1, hv, par);
    // ---- rows of X are [P_i | alpha_i] (:183-186) ----
    par(m * n + m, [&](int e) {
      if (e < m * n) {
        const T v = SY[size_t(m) * m + e];
        sP[e] = v;
        a.P[size_t(k) * m * n + e] = v;
      } else {
        const T v = SY[size_t(m) * (m + n) + (e - m * n)];
        sAl[e - m * n] = v;
        a.alpha[size_t(k) * m + (e - m * n)] = v;
      }
    });
    // ---- F = A - sum B_i P_i, beta = -sum B_i alpha_i (:189-194) ----
    par(n * n + n, [&](int e) {
      if (e < n * n) {
        const int row = e % n, c = e / n;
        T s = A[e];
        for (int q = 0; q < m; q++) s -= B[row + n * q] * sP[q + m * c];
        F[e] = s;
      } else {
        const int row = e - n * n;
        T s = T(0);
        for (int q = 0; q < m; q++) s -= B[row + n * q] * sAl[q];
        beta[row] = s;
      }
    });
    // ---- zeta_i, Z_i updates, player after player (:198-212) ----
    for (int i = 0; i < N; i++) {
      T* Zi = Z + size_t(i) * n * n;
      T* zi = zeta + size_t(i) * n;
      par(n * n + n, [&](int e) {
        if (e < n * n) {  // U = F^T Z_i
          const int row = e % n, c = e / n;
          T s = T(0);
          for (int q = 0; q < n; q++) s += F[q + n * row] * Zi[q + n * c];
          U[e] = s;
        } else {  // zeta_i + Z_i beta
          const int row = e - n * n;
          T s = zi[row];
          for (int q = 0; q < n; q++) s += Zi[row + n * q] * beta[q];
          tz[row] = s;
        }
      });
      par(n * n + n, [&](int e) {
        if (e < n * n) {  // Z_i = U F + Q_i + sum_j P_j^T R_ij P_j
          const int row = e % n, c = e / n;
          T s = T(0);
          for (int q = 0; q < n; q++) s += U[row + n * q] * F[q + n * c];
          s += Q[size_t(i) * n * n + e];
          for (int p = 0; p < pt.npairs; p++) {
            if (pt.pi[p] != i) continue;
            const int j = pt.pj[p], mj = d.udim[j], uo = d.uoff[j];
            const T* Rij = R + pt.roff[p];
            T add = T(0);
            for (int aa = 0; aa < mj; aa++) {
              T rp = T(0);  // (R_ij P_j)[aa, c]
              for (int b = 0; b < mj; b++) rp += Rij[aa + mj * b] * sP[(uo + b) + m * c];
              add += sP[(uo + aa) + m * row] * rp;
            }
            s += add;
          }
          Zi[e] = s;
        } else {  // zeta_i = F^T (zeta_i + Z_i beta) + l_i + sum_j P_j^T (R_ij alpha_j - r_ij)
          const int row = e - n * n;
          T s = T(0);
          for (int q = 0; q < n; q++) s += F[q + n * row] * tz[q];
          s += l[size_t(i) * n + row];
          for (int p = 0; p < pt.npairs; p++) {
            if (pt.pi[p] != i) continue;
            const int j = pt.pj[p], mj = d.udim[j], uo = d.uoff[j];
            const T* Rij = R + pt.roff[p];
            const T* rij = r + pt.rgoff[p];
            T add = T(0);
            for (int aa = 0; aa < mj; aa++) {
              T ww = T(0);
              for (int b = 0; b < mj; b++) ww += Rij[aa + mj * b] * sAl[uo + b];
              add += sP[(uo + aa) + m * row] * (ww - rij[aa]);
            }
            s += add;
          }
          zi[row] = s;
        }
      });
    }
  }

  // ---- forward pass: delta_xs without the feedback term (:217-241) and ILQSolver::ExpectedDecrease (:364-398) ----
  if (a.dx == nullptr && a.ed_out == nullptr) return;
  par(n, [&](int e) { xb[e] = a.x0 ? a.x0[e] : T(0); });
  par(1, [&](int) { hv[1] = T(0); });
  int cur = 0;
  for (int k = 0; k < Tn; k++) {
    const T* x = xb + cur * n;
    T* xn = xb + (1 - cur) * n;
    // the step's A, B, alpha (and Q_i l_i for ExpectedDecrease) staged / formed in one parallel phase: the recursion
    // below then reads LDS only (the Z tiles of the backward sweep are free by now: Q_i l_i goes there)
    T* const ql = Z;
    {
      const T* gA = a.A + size_t(k) * n * n;
      const T* gB = a.Bm + size_t(k) * n * m;
      const T* gal = a.alpha + size_t(k) * m;
      const int cA = n * n, cB = n * m;
      par(cA + cB + m + (a.ed_out ? N * n : 0), [&](int e) {
        if (e < cA) sA[e] = gA[e];
        else if (e < cA + cB) sB[e - cA] = gB[e - cA];
        else if (e < cA + cB + m) sAl[e - cA - cB] = gal[e - cA - cB];
        else {
          const int w = e - cA - cB - m;
          ql[w] = gen_ql_entry<T>(d, a, k, w / n, w % n);
        }
      });
    }
    const T* al = sAl;
    par(n + N, [&](int e) {
      if (e < n) {
        if (a.dx) a.dx[size_t(k) * n + e] = x[e];
        T s = T(0);
        for (int c = 0; c < n; c++) s += sA[e + n * c] * x[c];
        T bsum = T(0);
        for (int q = 0; q < m; q++) bsum -= sB[e + n * q] * al[q];
        xn[e] = s + bsum;
      } else if (a.ed_out) {
        gen_ed_terms_ql<T>(d, a, pt, k, al, x, e - n, ql + size_t(e - n) * n, terms);
      }
    });
    if (a.ed_out)
      par(1, [&](int) {
        T ed = hv[1];
        for (int i = 0; i < N; i++) {
          ed -= terms[2 * i];
          if (k > 0) ed -= terms[2 * i + 1];
        }
        hv[1] = ed;
      });
    cur = 1 - cur;
  }
  if (a.ed_out) par(1, [&](int) { *a.ed_out = hv[1]; });
}

// ---------------------------------------------------------------------------------------------------------------------
// Open loop (src/lq_open_loop_solver.cpp:73-195).  Per backward step: W_i = R_ii^{-1} B_i^T, w_i = R_ii^{-1} r_ii (:119-126;
// the reference factors R_ii by LDL^T, here the m_i x m_i system is eliminated with partial pivoting), Lambda = I + sum_i
// B_i W_i M_i (:127), c = -sum_i B_i (W_i m_i + w_i) (:134-139), X = Lambda^{-1} A, y = Lambda^{-1} c by Householder QR
// (:131,144,148), M_i <- Q_i + A^T M_i X, m_i <- l_i + A^T (m_i + M_i y) (:141-150).  The forward pass (:156-185) needs,
// per step, X, y, V_i = W_i M_i[k+1] and g_i = W_i m_i[k+1] + w_i: x_{k+1} = X x_k + y, alpha_i,k = V_i x_{k+1} + g_i —
// one scratch row [X | y | V | g] per step; with costates also M_i[k+1], m_i[k+1] (the "fat" row).
// ---------------------------------------------------------------------------------------------------------------------
ILQG_HD int gen_ol_row_elems(int n, int m, int N, bool fat) {
  return n * n + n + m * n + m + (fat ? N * (n * n + n) : 0);
}
// LDS: [M (N n^2) | mv (N n) | W (m n) | w (m) | V (m n) | g (m) | SY (n (2 n + 1)) | MX (n^2) | tv (n) | Rw (m (m + n + 1)) |
//       x (2 n) | terms (2 N) | hv (4) | cs (N n)]
//       ... | sA (n^2) | sB (n m)]: the step's A and B, staged once per step (round 6, as in the feedback sweep)
ILQG_HD size_t gen_openloop_lds_elems(int n, int N, int m) {
  return size_t(N) * n * n + size_t(N) * n + 2 * (size_t(m) * n + m) + size_t(n) * (2 * n + 1) + size_t(n) * n + n +
         size_t(m) * (m + n + 1) + 2 * size_t(n) + 2 * size_t(N) + 4 + size_t(N) * n + size_t(n) * n + size_t(n) * m;
}

template <typename T, typename Par>
ILQG_HD void lq_openloop_generic(const GenDims& d, const GenLQArgs<T>& a, const PairTable& pt, T* sm, const Par& par) {
  const int n = d.n, N = d.N, m = d.m, Tn = d.T;
  const bool fat = a.costates != nullptr;
  const int ROW = gen_ol_row_elems(n, m, N, fat);
  T* M = sm;
  T* mv = M + size_t(N) * n * n;
  T* W = mv + size_t(N) * n;
  T* w = W + size_t(m) * n;
  T* V = w + m;
  T* g = V + size_t(m) * n;
  T* SY = g + m;
  T* MX = SY + size_t(n) * (2 * n + 1);
  T* tv = MX + size_t(n) * n;
  T* Rw = tv + n;  // per player: [R_ii | B_i^T | r_ii], m_i rows, at row offset uoff_i of an m x (m + n + 1) block
  T* xb = Rw + size_t(m) * (m + n + 1);
  T* terms = xb + 2 * n;
  T* hv = terms + 2 * N;
  T* cs = hv + 4;  // M_i[k+1] x_{k+1} + m_i[k+1] of every player (costates)
  T* sA = cs + size_t(N) * n;
  T* sB = sA + size_t(n) * n;

  par(N * n * n, [&](int e) { M[e] = a.Q[size_t(Tn - 1) * N * n * n + e]; });  // :105-108
  par(N * n, [&](int e) { mv[e] = a.l[size_t(Tn - 1) * N * n + e]; });
  par(Tn * m * n, [&](int e) { a.P[e] = T(0); });  // P stays zero (:96-102)
  par(m, [&](int e) { a.alpha[size_t(Tn - 1) * m + e] = T(0); });

  for (int k = Tn - 2; k >= 0; k--) {
    {  // the step's A and B into LDS: one coalesced pass instead of a global read per term of every dot product
      const T* gA = a.A + size_t(k) * n * n;
      const T* gB = a.Bm + size_t(k) * n * m;
      const int cA = n * n;
      par(cA + n * m, [&](int e) {
        if (e < cA) sA[e] = gA[e];
        else sB[e - cA] = gB[e - cA];
      });
    }
    const T *A = sA, *B = sB;
    const T* Q = a.Q + size_t(k) * N * n * n;
    const T* l = a.l + size_t(k) * N * n;
    const T* R = a.R + size_t(k) * pt.Rsz;
    const T* r = a.r + size_t(k) * pt.rsz;
    T* row_out = a.scratch + size_t(k) * ROW;
    // ---- R_ii [W_i | w_i] = [B_i^T | r_ii] ----
    const int ldw = m;  // leading dimension of the Rw block (rows = stacked controls)
    par(m * (m + n + 1), [&](int e) {
      const int row = e % m, c = e / m, i = gen_player_of(d, row), mi = d.udim[i], uo = d.uoff[i];
      T v = T(0);
      if (c < m) {
        if (c >= uo && c < uo + mi) v = R[pt.roff[pt.pii[i]] + (row - uo) + mi * (c - uo)];
      } else if (c < m + n) {
        v = B[(c - m) + n * row];  // B_i^T
      } else {
        v = r[pt.rgoff[pt.pii[i]] + (row - uo)];
      }
      Rw[row + ldw * c] = v;
    });
    par(N, [&](int i) {  // one small elimination with partial pivoting per player
      const int mi = d.udim[i], uo = d.uoff[i];
      auto at = [&](int rr, int cc) -> T& { return Rw[(uo + rr) + ldw * cc]; };  // cc: uo.. = R_ii columns, m.. = rhs
      for (int kk = 0; kk < mi; kk++) {
        int piv = kk;
        T best = gen_abs(at(kk, uo + kk));
        for (int rr = kk + 1; rr < mi; rr++)
          if (gen_abs(at(rr, uo + kk)) > best) {
            best = gen_abs(at(rr, uo + kk));
            piv = rr;
          }
        if (piv != kk) {
          for (int cc = uo + kk; cc < uo + mi; cc++) { const T t0 = at(kk, cc); at(kk, cc) = at(piv, cc); at(piv, cc) = t0; }
          for (int cc = m; cc < m + n + 1; cc++) { const T t0 = at(kk, cc); at(kk, cc) = at(piv, cc); at(piv, cc) = t0; }
        }
        const T pinv = T(1) / at(kk, uo + kk);
        for (int rr = kk + 1; rr < mi; rr++) {
          const T f = at(rr, uo + kk) * pinv;
          for (int cc = uo + kk + 1; cc < uo + mi; cc++) at(rr, cc) -= f * at(kk, cc);
          for (int cc = m; cc < m + n + 1; cc++) at(rr, cc) -= f * at(kk, cc);
        }
      }
      for (int cc = m; cc < m + n + 1; cc++)
        for (int rr = mi - 1; rr >= 0; rr--) {
          T s = at(rr, cc);
          for (int k2 = rr + 1; k2 < mi; k2++) s -= at(rr, uo + k2) * at(k2, cc);
          at(rr, cc) = s / at(rr, uo + rr);
        }
    });
    par(m * n + m, [&](int e) {
      if (e < m * n)
        W[e] = Rw[(e % m) + ldw * (m + e / m)];
      else
        w[e - m * n] = Rw[(e - m * n) + ldw * (m + n)];
    });
    // ---- V_i = W_i M_i[k+1], g_i = W_i m_i[k+1] + w_i ----
    par(m * n + m, [&](int e) {
      if (e < m * n) {
        const int row = e % m, c = e / m, i = gen_player_of(d, row);
        const T* Mi = M + size_t(i) * n * n;
        T s = T(0);
        for (int q = 0; q < n; q++) s += W[row + m * q] * Mi[q + n * c];
        V[e] = s;
      } else {
        const int row = e - m * n, i = gen_player_of(d, row);
        T s = T(0);
        for (int q = 0; q < n; q++) s += W[row + m * q] * mv[i * n + q];
        g[row] = s + w[row];
      }
    });
    // ---- [Lambda | A | c] ----
    par(n * (2 * n + 1), [&](int e) {
      const int row = e % n, c = e / n;
      if (c < n) {
        T s = row == c ? T(1) : T(0);
        for (int q = 0; q < m; q++) s += B[row + n * q] * V[q + m * c];
        SY[e] = s;
      } else if (c < 2 * n) {
        SY[e] = A[row + n * (c - n)];
      } else {
        T s = T(0);
        for (int q = 0; q < m; q++) s -= B[row + n * q] * g[q];
        SY[e] = s;
      }
    });
    gen_qr_solve<T>(SY, n, n + 1, hv, par);
    const T* X = SY + size_t(n) * n;
    const T* y = SY + size_t(n) * 2 * n;
    // the forward pass's row of this step (M_i[k+1], m_i[k+1] too when costates are asked for)
    par(ROW, [&](int e) {
      T v;
      if (e < n * n + n) v = X[e];  // X then y: contiguous behind Lambda
      else if (e < n * n + n + m * n) v = V[e - (n * n + n)];
      else if (e < n * n + n + m * n + m) v = g[e - (n * n + n + m * n)];
      else if (e < n * n + n + m * n + m + N * n * n) v = M[e - (n * n + n + m * n + m)];
      else v = mv[e - (n * n + n + m * n + m + N * n * n)];
      row_out[e] = v;
    });
    // ---- M_i = Q_i + A^T (M_i X), m_i = l_i + A^T (m_i + M_i y), player after player ----
    for (int i = 0; i < N; i++) {
      T* Mi = M + size_t(i) * n * n;
      T* mi_ = mv + size_t(i) * n;
      par(n * n + n, [&](int e) {
        if (e < n * n) {
          const int row = e % n, c = e / n;
          T s = T(0);
          for (int q = 0; q < n; q++) s += Mi[row + n * q] * X[q + n * c];
          MX[e] = s;
        } else {
          const int row = e - n * n;
          T s = T(0);
          for (int q = 0; q < n; q++) s += Mi[row + n * q] * y[q];
          tv[row] = s + mi_[row];
        }
      });
      par(n * n + n, [&](int e) {
        if (e < n * n) {
          const int row = e % n, c = e / n;
          T s = T(0);
          for (int q = 0; q < n; q++) s += A[q + n * row] * MX[q + n * c];
          Mi[e] = s + Q[size_t(i) * n * n + e];
        } else {
          const int row = e - n * n;
          T s = T(0);
          for (int q = 0; q < n; q++) s += A[q + n * row] * tv[q];
          mi_[row] = s + l[size_t(i) * n + row];
        }
      });
    }
  }

  // ---- forward pass (:156-192) ----
  par(n, [&](int e) { xb[e] = a.x0 ? a.x0[e] : T(0); });
  par(1, [&](int) { hv[1] = T(0); });
  int cur = 0;
  for (int k = 0; k < Tn; k++) {
    const T* x = xb + cur * n;
    T* xn = xb + (1 - cur) * n;
    if (a.dx) par(n, [&](int e) { a.dx[size_t(k) * n + e] = x[e]; });
    if (k == Tn - 1) {
      if (a.costates) par(N * n, [&](int e) { a.costates[size_t(k) * N * n + e] = T(0); });  // :191
      if (a.ed_out) {
        par(N, [&](int i) { gen_ed_terms<T>(d, a, pt, k, a.alpha + size_t(k) * m, x, i, terms); });
        par(1, [&](int) {
          T ed = hv[1];
          for (int i = 0; i < N; i++) {
            ed -= terms[2 * i];
            if (k > 0) ed -= terms[2 * i + 1];
          }
          hv[1] = ed;
        });
      }
      break;
    }
    const T* row = a.scratch + size_t(k) * ROW;
    // the row's [X | y | V | g] staged into LDS (SY, V, g are free after the backward sweep), and Q_i l_i of the step
    // formed in parallel for ExpectedDecrease (into M: the forward pass reads M_i[k+1] from the row)
    T* const ql = M;
    {
      const int cXy = n * n + n, cV = m * n + m;
      par(cXy + cV + (a.ed_out ? N * n : 0), [&](int e) {
        if (e < cXy) SY[e] = row[e];
        else if (e < cXy + m * n) V[e - cXy] = row[e];
        else if (e < cXy + cV) g[e - cXy - m * n] = row[e];
        else {
          const int w = e - cXy - cV;
          ql[w] = gen_ql_entry<T>(d, a, k, w / n, w % n);
        }
      });
    }
    par(n, [&](int e) {  // x_{k+1} = X x_k + y
      T s = T(0);
      for (int c = 0; c < n; c++) s += SY[e + n * c] * x[c];
      xn[e] = s + SY[n * n + e];
    });
    par(m + (fat ? N * n : 0), [&](int e) {
      if (e < m) {  // alpha_i,k = V_i x_{k+1} + g_i
        T s = T(0);
        for (int c = 0; c < n; c++) s += V[e + m * c] * xn[c];
        a.alpha[size_t(k) * m + e] = s + g[e];
      } else {  // M_i[k+1] x_{k+1} + m_i[k+1], for the costate
        const int i = (e - m) / n, rr = (e - m) % n;
        const T* Mi = row + n * n + n + m * n + m + size_t(i) * n * n;
        const T* mi_ = row + n * n + n + m * n + m + size_t(N) * n * n + size_t(i) * n;
        T s = T(0);
        for (int c = 0; c < n; c++) s += Mi[rr + n * c] * xn[c];
        cs[size_t(i) * n + rr] = s + mi_[rr];
      }
    });
    if (fat)
      par(N * n, [&](int e) {  // costate_i,k = A_k^T (M_i[k+1] x_{k+1} + m_i[k+1])  (:171-176)
        const int i = e / n, rr = e % n;
        const T* A = a.A + size_t(k) * n * n;
        T s = T(0);
        for (int q = 0; q < n; q++) s += A[q + n * rr] * cs[size_t(i) * n + q];
        a.costates[(size_t(k) * N + i) * n + rr] = s;
      });
    if (a.ed_out) {
      par(N, [&](int i) { gen_ed_terms_ql<T>(d, a, pt, k, a.alpha + size_t(k) * m, x, i, ql + size_t(i) * n, terms); });
      par(1, [&](int) {
        T ed = hv[1];
        for (int i = 0; i < N; i++) {
          ed -= terms[2 * i];
          if (k > 0) ed -= terms[2 * i + 1];
        }
        hv[1] = ed;
      });
    }
    cur = 1 - cur;
  }
  if (a.ed_out) par(1, [&](int) { *a.ed_out = hv[1]; });
}

}  // namespace ilqg
