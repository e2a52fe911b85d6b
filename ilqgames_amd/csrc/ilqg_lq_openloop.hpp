// ilqg_lq_openloop.hpp — open-loop LQ Nash sweep for ONE game instance per workgroup (gfx950).
//
// Computes what LQOpenLoopSolver::Solve computes (src/lq_open_loop_solver.cpp:73-195):
//   backward, k = T-2 .. 0:
//     W_i = R_ii^{-1} B_i^T, w_i = R_ii^{-1} r_ii                       (LDLT of R_ii, :119-126)
//     Lambda = I + sum_i B_i W_i M_i[k+1]                               (:127-128)
//     c = -sum_i B_i (W_i m_i[k+1] + w_i)                               (:134-139)
//     X = Lambda^{-1} A, y = Lambda^{-1} c                              (:131,144,148; here through the matrix
//                                                                        inversion lemma: an m x m system)
//     M_i[k] = Q_i + A^T M_i[k+1] X,   m_i[k] = l_i + A^T (m_i[k+1] + M_i[k+1] y)   (:141-150)
//   forward, k = 0 .. T-2:
//     x_{k+1} = Lambda_k^{-1} (A x_k + c_k) = X_k x_k + y_k             (:165)
//     alpha_i,k = W_i,k (M_i[k+1] x_{k+1} + m_i[k+1]) + w_i,k           (:169-172);  P == 0
// Mapping: lane (i,c) owns column c of player i's M_i update; the m x m system K [Z | z] = [V A | V c] behind
// Lambda^{-1} sits one column per lane in wave 0 (m + n + 1 columns) and reuses the feedback sweep's shuffle-free
// Householder QR; the per-step blocks are staged by LDS-DMA exactly like the feedback sweep.  The
// backward pass leaves one scratch row per step ([X|y|W|w|M|m|Q l]) for the forward pass.
#pragma once

#include "ilqg_lq.hpp"

namespace ilqg {

template <typename T, int NX, int NP, int MU>
struct OLCfg {
  using C = LQCfg<T, NX, NP, MU>;
  static constexpr int M = NP * MU;
  static constexpr int NT = C::NT;
  static_assert(NP * MU + NX + 1 <= 64, "K [Z | z] = [V A | V c] must fit one wavefront");
  // scratch row (global), one per time step
  static constexpr int rX = 0;
  static constexpr int ry = rX + NX * NX;
  static constexpr int rW = ry + NX;
  static constexpr int rw = rW + M * NX;
  static constexpr int rM = rw + M;
  static constexpr int rm = rM + NP * NX * NX;
  static constexpr int rql = rm + NP * NX;
  static constexpr int ROW = (rql + NP * NX + 3) & ~3;
  // LDS (elements): DMA image(s), then working set.  Two images (DMA of step k-1 under step k) when that keeps
  // the instance under ~48 KB; the n = 24 problems would take 92 KB and leave a CU with a single resident
  // instance (two waves on four SIMDs), so they run single-buffered: a step there is ~40 us, the exposed DMA
  // round trip ~1 us, and three instances per CU are worth far more than the overlap.
  static constexpr int FS_ = ROW + NP * NX * NX + NP * NX + 4;
  static constexpr bool DB = (2 * (C::IMG > FS_ ? C::IMG : FS_) + NP * NX * NX + 2 * NX * NX) * int(sizeof(T)) <= 48 * 1024;
  static constexpr int NB = DB ? 2 : 1;
  static constexpr int oM = NB * C::IMG;
  static constexpr int om = oM + NP * NX * NX;
  static constexpr int oW = om + NP * NX;
  static constexpr int ow = oW + M * NX;
  static constexpr int oV = ow + M;
  static constexpr int og = oV + M * NX;
  static constexpr int oX = og + M;
  static constexpr int oy = oX + NX * NX;
  static constexpr int oT = oy + NX;
  static constexpr int LDS_BWD = oT + NP * NX;
  // forward pass: overlays the backward working set — two staged scratch rows, then x, x+, it, alpha
  static constexpr int FS = FS_;
  static constexpr int fxs = NB * FS;
  static constexpr int fT = fxs + 2 * NX;
  static constexpr int fg = fT + NP * NX;
  static constexpr int LDS_FWD = fg + M;
  static constexpr int LDS_ELEMS = LDS_FWD > LDS_BWD ? LDS_FWD : LDS_BWD;
};

// Solve R y = b for a small SPD block by LDL^T without pivoting (Eigen::LDLT at
// src/lq_open_loop_solver.cpp:124-126; R_ii is diagonally dominant in every config).
template <typename T, int MU>
__device__ __forceinline__ void ldlt_solve(const T* R /* MU x MU col-major */, T (&b)[MU]) {
  T Lm[MU][MU], D[MU];
#pragma unroll
  for (int jx = 0; jx < MU; jx++) {
    T dj = R[jx + MU * jx];
#pragma unroll
    for (int k = 0; k < jx; k++) dj -= Lm[jx][k] * Lm[jx][k] * D[k];
    D[jx] = dj;
#pragma unroll
    for (int i = jx + 1; i < MU; i++) {
      T s = R[i + MU * jx];
#pragma unroll
      for (int k = 0; k < jx; k++) s -= Lm[i][k] * Lm[jx][k] * D[k];
      Lm[i][jx] = s / dj;
    }
  }
#pragma unroll
  for (int i = 0; i < MU; i++)
#pragma unroll
    for (int k = 0; k < i; k++) b[i] -= Lm[i][k] * b[k];
#pragma unroll
  for (int i = 0; i < MU; i++) b[i] /= D[i];
#pragma unroll
  for (int i = MU - 1; i >= 0; i--)
#pragma unroll
    for (int k = i + 1; k < MU; k++) b[i] -= Lm[k][i] * b[k];
}

// a.scratch must hold T_steps rows of OLCfg::ROW elements.  a.P is written as zero (:96-102).
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void lq_openloop_instance(const LQArgs<T>& a, const PairTable& pt, T* sm) {
  using C = LQCfg<T, NX, NP, MU>;
  using O = OLCfg<T, NX, NP, MU>;
  constexpr int M = C::M, L = C::L, NT = C::NT, ROW = O::ROW;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const bool zl = t < L;
  const int pi = zl ? t / NX : 0;
  const int pc = zl ? t % NX : 0;
  const int Tn = a.T_steps;
  const PairRegs<NP> pr(pt);
  T *sB, *sA, *sQ, *sl, *sR, *sr;
  auto set_img = [&](int which) {
    T* img = sm + which * C::IMG;
    sB = img + C::oB; sA = img + C::oA; sQ = img + C::oQ; sl = img + C::ol; sR = img + C::oR; sr = img + C::or_;
  };
  T* sM = sm + O::oM;
  T* smv = sm + O::om;
  T* sW = sm + O::oW;
  T* sw = sm + O::ow;
  T* sV = sm + O::oV;
  T* sg = sm + O::og;
  T* sX = sm + O::oX;
  T* sy = sm + O::oy;
  T* sT = sm + O::oT;
  auto row_of = [&](int k) { return a.scratch + size_t(k) * ROW; };
  auto ro_ii = [&](int i) {
    int v = 0;
#pragma unroll
    for (int e = 0; e < NP; e++) v = (i == e) ? pr.ro[e][e] : v;
    return v;
  };
  auto rg_ii = [&](int i) {
    int v = 0;
#pragma unroll
    for (int e = 0; e < NP; e++) v = (i == e) ? pr.rg[e][e] : v;
    return v;
  };
  // store M_i, m_i (value functions AT step k) and Q_i l_i into scratch row k
  auto store_value_row = [&](int k) {
    T* row = row_of(k);
    for (int e = t; e < NP * NX * NX; e += NT) row[O::rM + e] = sM[e];
    for (int e = t; e < NP * NX; e += NT) row[O::rm + e] = smv[e];
    if (zl) {
      T s = T(0);
#pragma unroll
      for (int c = 0; c < NX; c++) s += sQ[pi * NX * NX + pc + NX * c] * sl[pi * NX + c];
      row[O::rql + t] = s;
    }
  };

  // diagnostics (ILQG_PROFILE build): shader-clock cycles of thread 0 per phase, summed over the steps
  long long ph_c = (kProfile && a.ph) ? clock64() : 0;
  auto PH = [&](int slot) {
    if (kProfile && a.ph && t == 0) {
      const long long c = clock64();
      a.ph[slot] += c - ph_c;
      ph_c = c;
    }
  };
  // ---- terminal step (:105-108) ----
  int cur = 0;
  lq_stage_issue<T, NX, NP, MU>(a, pt, Tn - 1, sm, t);
  dma_wait();
  __syncthreads();
  set_img(0);
  for (int e = t; e < NP * NX * NX; e += NT) sM[e] = sQ[e];
  for (int e = t; e < NP * NX; e += NT) smv[e] = sl[e];
  lds_sync(NT <= 64);
  store_value_row(Tn - 1);
  if (O::DB) {
    if (Tn >= 2) lq_stage_issue<T, NX, NP, MU>(a, pt, Tn - 2, sm + C::IMG, t);
    dma_wait();
    lds_sync(NT <= 64);
    cur = 1;
  } else {
    lds_sync(NT <= 64);  // every read of the terminal image is done
    if (Tn >= 2) lq_stage_issue<T, NX, NP, MU>(a, pt, Tn - 2, sm, t);
    dma_wait();
    lds_sync(NT <= 64);
    cur = 0;
  }
  set_img(cur);

#pragma unroll 1
  for (int k = Tn - 2; k >= 0; k--) {
    if (O::DB && k > 0) lq_stage_issue<T, NX, NP, MU>(a, pt, k - 1, sm + (1 - cur) * C::IMG, t);
    // ---- W_i = R_ii^{-1} B_i^T (column c by lane (i,c)), w_i = R_ii^{-1} r_ii ----
    if (zl) {
      T b[MU];
#pragma unroll
      for (int aa = 0; aa < MU; aa++) b[aa] = sB[pc + NX * (pi * MU + aa)];
      ldlt_solve<T, MU>(sR + ro_ii(pi), b);
#pragma unroll
      for (int aa = 0; aa < MU; aa++) sW[(pi * MU + aa) + M * pc] = b[aa];
    }
    if (t < NP) {
      T b[MU];
#pragma unroll
      for (int aa = 0; aa < MU; aa++) b[aa] = sr[rg_ii(t) + aa];
      ldlt_solve<T, MU>(sR + ro_ii(t), b);
#pragma unroll
      for (int aa = 0; aa < MU; aa++) sw[t * MU + aa] = b[aa];
    }
    lds_sync(NT <= 64);
    PH(0);
    // ---- V_i = W_i M_i (column c), g = W_i m_i + w_i ----
    if (zl) {
#pragma unroll
      for (int aa = 0; aa < MU; aa++) {
        T s = T(0);
#pragma unroll
        for (int r = 0; r < NX; r++) s += sW[(pi * MU + aa) + M * r] * sM[pi * NX * NX + r + NX * pc];
        sV[(pi * MU + aa) + M * pc] = s;
      }
    }
    if (t < M) {
      const int i = t / MU;
      T s = T(0);
#pragma unroll
      for (int r = 0; r < NX; r++) s += sW[t + M * r] * smv[i * NX + r];
      sg[t] = s + sw[t];
    }
    lds_sync(NT <= 64);
    PH(1);
    // ---- Lambda [X | y] = [A | c] through the matrix inversion lemma (wave 0) ----
    // Lambda = I + B V with B = [B_0 .. B_{N-1}] (n x m) and V = [W_0 M_0; ..; W_{N-1} M_{N-1}] (m x n): a rank-m
    // update of the identity, so  Lambda^{-1} = I - B K^{-1} V,  K = I_m + V B  (m x m), and
    //     X = A - B Z,  y = c - B z,   K [Z | z] = [V A | V c].
    // The reference factors the n x n Lambda by Householder QR (:131) and solves for n + 1 right-hand sides — 24
    // dependent reflections per step at n = 24, which was half of this sweep; the m x m system (8 x 8) takes the
    // same column-per-lane QR (M columns of K, then the NX + 1 right-hand sides).  Both are the solution of the same
    // linear system; the parity tests compare with the QR oracle.
    if (t < 64) {
      static_assert(M + NX + 1 <= 64, "K [Z | z] = [V A | V c] must fit one wavefront");
      T col[M], x[M];
#pragma unroll
      for (int q = 0; q < M; q++) {
        col[q] = T(0);
        x[q] = T(0);
      }
      const bool isK = t < M, isA = t >= M && t < M + NX, isc = t == M + NX;
      const int cidx = isA ? t - M : 0;
      // this lane's vector v (a column of B, a column of A, or c = -B g), then V v
      T cvec[NX];
#pragma unroll
      for (int r = 0; r < NX; r++) {
        T v = T(0);
        if (isK) {
          v = sB[r + NX * t];
        } else if (isA) {
          v = sA[r + NX * cidx];
        } else if (isc) {
#pragma unroll
          for (int q = 0; q < M; q++) v -= sB[r + NX * q] * sg[q];
        }
        cvec[r] = v;
      }
#pragma unroll
      for (int r = 0; r < NX; r++) {
#pragma unroll
        for (int q = 0; q < M; q++) col[q] += sV[q + M * r] * cvec[r];
      }
      if (isK) {
#pragma unroll
        for (int q = 0; q < M; q++) col[q] += (q == t) ? T(1) : T(0);
      }
      qr_solve_columns<T, M>(col, lane, x);
      if (isA || isc) {
#pragma unroll
        for (int r = 0; r < NX; r++) {
          T v = cvec[r];
#pragma unroll
          for (int q = 0; q < M; q++) v -= sB[r + NX * q] * x[q];
          if (isA)
            sX[r + NX * cidx] = v;
          else
            sy[r] = v;
        }
      }
    }
    {  // W, w of this step -> scratch row k (forward pass)
      T* row = row_of(k);
      for (int e = t; e < M * NX; e += NT) row[O::rW + e] = sW[e];
      if (t < M) row[O::rw + t] = sw[t];
    }
    lds_sync(NT <= 64);
    {  // X, y -> scratch row k, from LDS so that the stores are contiguous (a lane owns a COLUMN of X: stored
       // from its registers, every store instruction would touch one cache line per lane)
      T* row = row_of(k);
      for (int e = t; e < NX * NX; e += NT) row[O::rX + e] = sX[e];
      if (t < NX) row[O::ry + t] = sy[t];
    }
    PH(2);
    // ---- M_i[:,c] = Q_i[:,c] + A^T (M_i X[:,c]);  t_i = m_i + M_i y ----
    T mn[NX];
    T tv = T(0);
    if (zl) {
      T xc[NX], u[NX];
#pragma unroll
      for (int kk = 0; kk < NX; kk++) xc[kk] = sX[kk + NX * pc];
#pragma unroll
      for (int r = 0; r < NX; r++) {
        T s = T(0);
#pragma unroll
        for (int kk = 0; kk < NX; kk++) s += sM[pi * NX * NX + r + NX * kk] * xc[kk];
        u[r] = s;
      }
#pragma unroll
      for (int r = 0; r < NX; r++) {
        T s = T(0);
#pragma unroll
        for (int kk = 0; kk < NX; kk++) s += sA[kk + NX * r] * u[kk];
        mn[r] = sQ[pi * NX * NX + r + NX * pc] + s;
      }
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < NX; kk++) s += sM[pi * NX * NX + pc + NX * kk] * sy[kk];
      tv = smv[pi * NX + pc] + s;
    }
    lds_sync(NT <= 64);
    if (zl) {
#pragma unroll
      for (int r = 0; r < NX; r++) sM[pi * NX * NX + r + NX * pc] = mn[r];
      sT[t] = tv;
    }
    lds_sync(NT <= 64);
    if (zl) {
      T s = T(0);
#pragma unroll
      for (int kk = 0; kk < NX; kk++) s += sA[kk + NX * pc] * sT[pi * NX + kk];
      smv[t] = sl[t] + s;
    }
    lds_sync(NT <= 64);
    PH(3);
    store_value_row(k);
    PH(4);
    if (O::DB) {
      dma_wait();
      lds_sync(NT <= 64);
      cur = 1 - cur;
      set_img(cur);
    } else {
      lds_sync(NT <= 64);  // the image is free: refill it in place
      if (k > 0) lq_stage_issue<T, NX, NP, MU>(a, pt, k - 1, sm, t);
      dma_wait();
      lds_sync(NT <= 64);
    }
    PH(5);
  }

  // ---- forward pass (:156-192) ----
  __syncthreads();  // scratch rows were written by other lanes
  T* f0 = sm;               // staged [row k | M,m of row k+1]  (overlays the backward working set)
  T* sx = sm + O::fxs;      // x_k
  T* sxn = sx + NX;         // x_{k+1}
  sT = sm + O::fT;
  sg = sm + O::fg;
  constexpr int FS = O::FS;
  constexpr int S = int(sizeof(T));
  auto stage = [&](int k, int which) {
    T* dst = f0 + which * FS;
    dma_g2l<NT, false>(row_of(k), dst, (O::rM) * S, t);                                  // X, y, W, w
    dma_g2l<NT, false>(row_of(k) + O::rql, dst + O::rql, NP * NX * S, t);                // Q_i l_i of step k
    dma_g2l<NT, false>(row_of(k + 1) + O::rM, dst + ROW, (NP * NX * NX + NP * NX) * S, t);  // M, m at k+1
  };
  if (t < NX) sx[t] = a.x0 ? a.x0[t] : T(0);
  for (int e = t; e < M * NX; e += NT)
    for (int k = 0; k < Tn; k++) a.P[size_t(k) * M * NX + e] = T(0);  // open loop: P stays zero
  T ed = T(0);
  cur = 0;
  if (Tn >= 2) stage(0, 0);
  dma_wait();
  lds_sync(NT <= 64);
#pragma unroll 1
  for (int k = 0; k < Tn - 1; k++) {
    if (O::DB && k + 2 < Tn) stage(k + 1, 1 - cur);
    const T* fr = f0 + cur * FS;
    if (a.dx && t < NX) a.dx[size_t(k) * NX + t] = sx[t];
    if (t < NX) {
      T s = fr[O::ry + t];
#pragma unroll
      for (int c = 0; c < NX; c++) s += fr[O::rX + t + NX * c] * sx[c];
      sxn[t] = s;
    }
    lds_sync(NT <= 64);
    if (zl) {  // it_i = M_i[k+1] x_{k+1} + m_i[k+1]
      T s = fr[ROW + NP * NX * NX + t];
#pragma unroll
      for (int c = 0; c < NX; c++) s += fr[ROW + pi * NX * NX + pc + NX * c] * sxn[c];
      sT[t] = s;
    }
    lds_sync(NT <= 64);
    T al = T(0);
    if (t < M) {
      const int i = t / MU;
      T s = T(0);
#pragma unroll
      for (int r = 0; r < NX; r++) s += fr[O::rW + t + M * r] * sT[i * NX + r];
      al = s + fr[O::rw + t];
      a.alpha[size_t(k) * M + t] = al;
      sg[t] = al;
    }
    if (a.ed_out) {  // ILQSolver::ExpectedDecrease (ilq_solver.cpp:364-398) for this step
      lds_sync(NT <= 64);
      if (t < 64) {
        T st = T(0), ct = T(0);
        if (t < NP) {
          const T* Rg = a.R + size_t(k) * pt.Rsz + ro_ii(t);
          const T* rg = a.r + size_t(k) * pt.rsz + rg_ii(t);
#pragma unroll
          for (int c = 0; c < MU; c++) {
            T aR = T(0);
#pragma unroll
            for (int b = 0; b < MU; b++) aR += sg[t * MU + b] * Rg[b + MU * c];
            ct += aR * rg[c];
          }
          if (k > 0) {
#pragma unroll
            for (int c = 0; c < NX; c++) st += sx[c] * fr[O::rql + t * NX + c];
          }
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
          ed -= shfl(ct, i);
          if (k > 0) ed -= shfl(st, i);
        }
      }
    }
    lds_sync(NT <= 64);
    if (t < NX) sx[t] = sxn[t];
    if (O::DB) {
      dma_wait();
      lds_sync(NT <= 64);
      cur = 1 - cur;
    } else {
      lds_sync(NT <= 64);  // staged row consumed: refill in place
      if (k + 2 < Tn) stage(k + 1, 0);
      dma_wait();
      lds_sync(NT <= 64);
    }
  }
  PH(6);
  if (a.dx && t < NX) a.dx[size_t(Tn - 1) * NX + t] = sx[t];  // :188-192
  if (t < M) a.alpha[size_t(Tn - 1) * M + t] = T(0);
  if (a.ed_out) {
    // step T-1: alpha = 0; state term delta_x^T Q l
    __syncthreads();
    if (t < 64) {
      T st = T(0);
      if (t < NP && Tn > 1) {
        const T* ql = row_of(Tn - 1) + O::rql;
#pragma unroll
        for (int c = 0; c < NX; c++) st += sx[c] * ql[t * NX + c];
      }
#pragma unroll
      for (int i = 0; i < NP; i++) ed -= shfl(st, i);
    }
    if (t == 0) *a.ed_out = ed;
  }
}

}  // namespace ilqg
