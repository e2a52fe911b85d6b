// ilqg_lq_feedback1w.hpp — the one-tile feedback sweep (n < 16) with ONE wavefront per game instance: the throughput form.
//
// The player-parallel sweep (lq_feedback_instance_mfma_pw, ilqg_lq.hpp) gives an instance one wave per player: the
// shortest chain per step, which is what a batch of a few instances per CU needs — but every wave repeats what the
// players share (F = A - B P, the operand loads), two waves idle through the m x m solve, each step costs two workgroup
// barriers, and three waves of 168 registers with a double-buffered 36 KB image hold a CU to four instances whatever the
// batch (B = 8192 ran at the per-instance rate of B = 1024: profiles/r03_headline_f64_b8192.md).  A batch of many
// instances per CU does not need the short chain, it needs the fewest instructions per instance and as many instances
// in flight as the LDS holds.  Here one wave walks the players one after the other:
//   * the shared work is done once per step (about half the vector instructions of the three-wave form per instance);
//   * the players' matrix-instruction chains are independent of each other and sit next to each other in the
//     instruction stream, so the matrix pipe is fed from one wave;
//   * no workgroup barrier at all — a wave's LDS operations execute in order;
//   * one step image instead of two (the compact row of step k - 1 is scattered over it at the end of step k, out of a
//     staging row the DMA engine filled during the step): 20 KB per instance in fp64, 10 KB in fp32 — eight / sixteen
//     instances per CU, each with the register budget of two / four waves per SIMD.
// Same recursion, same operand layouts, same order of operations per player as the player-parallel form (so the same
// results to rounding; the forced-step parity tests run both).  Only what the solver's LQ part needs is built: compact
// rows in (LQArgs::compact), symmetric costs, zeta riding in the spare tile column (n < 16), the forward pass deferred
// to the next trial pass (scratch rows out).
// ILQSolver::ExpectedDecrease (src/ilq_solver.cpp:364-398) without a forward pass.  The reference sums
// -sum_k sum_i [alpha_i,k^T R_ii r_ii + [k > 0] dx_k^T Q_i l_i] over the delta_x of LQFeedbackSolver's forward pass
// (dx_0 = 0, dx_{k+1} = A_k dx_k + beta_k, beta_k = -sum B_i alpha_i: lq_feedback_solver.cpp:217-241) — a recursion that
// runs forwards in time, after a sweep that runs backwards.  With q_k = sum_i Q_i,k l_i,k and dx_k = sum_{j<k} A_{k-1} ..
// A_{j+1} beta_j, the state part is  sum_k q_k^T dx_k = sum_j beta_j^T mu_{j+1},  mu_k = q_k + A_k^T mu_{k+1}, mu_T = 0:
// an ADJOINT recursion that runs backwards, i.e. inside the sweep — one n-term product per lane and step more, no second
// pass over the horizon, no scratch rows, no delta_x (which only ever feeds this sum).  Same value up to the order of
// summation (the tests hold it to 1e-9 of the reference's order in fp64).  LQArgs::ed_out selects it; with
// LQArgs::defer_forward the scratch rows of the trial kernel's forward pass are written instead.
// Reference: LQFeedbackSolver::Solve, src/lq_feedback_solver.cpp:110-213.
#pragma once

#include "ilqg_lq.hpp"

namespace ilqg {

template <typename T, int NX, int NP, int MU>
struct W1Cfg {
  using C = LQCfg<T, NX, NP, MU>;
  static constexpr int M = NP * MU;
  static constexpr int LD = sizeof(T) == 8 ? 18 : 17;  // padded tile columns (bank-conflict-free accumulator-layout reads)
  static constexpr int TILE = (16 * LD + 3) & ~3;
  // the step image [tA | tB | tQ_0.. | l | R | r]
  static constexpr int oTA = 0;
  static constexpr int oTB = oTA + TILE;
  static constexpr int oTQ = oTB + TILE;
  static constexpr int oVl = oTQ + NP * TILE;
  static constexpr int oVR = (oVl + NP * NX + 3) & ~3;
  static constexpr int oVr = (oVR + C::RMAX + 3) & ~3;
  static constexpr int IMG = (oVr + C::rMAX + 3) & ~3;
  // intermediates
  static constexpr int oPt = IMG;               // [P | alpha] as a padded tile
  static constexpr int oAl = oPt + TILE;        // alpha (M, padded to 16)
  static constexpr int oYz = oAl + 16;          // y_zeta (M, padded to 16)
  // [S | Y] (M x 32, column-major) shares the [P | alpha] tile's memory where it fits: the solve has its columns in
  // registers before it writes P, and the next step's rows are written after the last read of P (one wave: its LDS
  // operations execute in order)
  static constexpr bool SY_IN_PT = M * 32 <= TILE;
  static constexpr int oSY = SY_IN_PT ? oPt : oYz + 16;
  static constexpr int oG = oYz + 16 + (SY_IN_PT ? 0 : M * 32);  // per player: its MU columns of G = Z^T B, interleaved [row][aa]
  // compact rows of at most kWords words (three per lane of the scattering wave; the games with a spare tile column have
  // 136 - 170): the staging row the DMA lands in, and where each of its words goes in the image (ints)
  static constexpr int kWords = 192;
  static constexpr int oSB = oG + NP * MU * 16;
  static constexpr int oCD = oSB + kWords;
  static constexpr int CD_ELEMS = (kWords * 4 + int(sizeof(T)) - 1) / int(sizeof(T));
  // ExpectedDecrease by the adjoint recursion (below): Q_i l_i of every player (NP x 16), mu (two buffers of 16)
  static constexpr int oQL = (oCD + CD_ELEMS + 3) & ~3;
  static constexpr int oMu = oQL + NP * 16;
  static constexpr int ELEMS = (oMu + 32 + 3) & ~3;
  static constexpr bool SUPPORTED = NX < 16 && M <= 16 && C::NSOLVE <= 32 && NP <= 4 && MU == 2;  // (MU == 2: rows_reduce4)
};

template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void lq_feedback_instance_mfma_1w(const LQArgs<T>& a, const PairTable& pt, T* sm) {
  using C = LQCfg<T, NX, NP, MU>;
  using W = W1Cfg<T, NX, NP, MU>;
  using TL = Tile<T>;
  using vec = typename TL::vec;
  constexpr int M = C::M, S = int(sizeof(T)), LD = W::LD, SCR = C::SCR, JB = NX;
  static_assert(W::SUPPORTED, "the single-wave sweep needs a spare tile column (n < 16) and a Nash system of <= 32 columns");
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int Tn = a.T_steps;
  const PairRegs<NP> pr(pt);
  const vec zero4 = {T(0), T(0), T(0), T(0)};
  constexpr int RS = TL::row(0, 1) - TL::row(0, 0);
  const int row0 = TL::row(g, 0);
  const int offD = row0 + LD * j;  // [row][col] of a padded tile
  const int offT = j + LD * row0;  // its transpose
  auto ldD = [&](const T* tile) {
    vec v;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = tile[offD + RS * r];
    return v;
  };
  auto ldDT = [&](const T* tile) {
    vec v;
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = tile[offT + LD * RS * r];
    return v;
  };
  const T mCols = (j < NX) ? T(1) : T(0);    // a proper state column
  const T mVecCol = (j == JB) ? T(1) : T(0);  // the column zeta / beta / alpha ride in
  (void)mCols;
  constexpr int gJ = sizeof(T) == 8 ? JB % 4 : JB / 4, rJ = sizeof(T) == 8 ? JB / 4 : JB % 4;
  static_assert(TL::row(gJ, rJ) == JB, "accumulator-layout position of row JB");

  T* const tA = sm + W::oTA;
  T* const tB = sm + W::oTB;
  T* const tQ0 = sm + W::oTQ;
  T* const sl = sm + W::oVl;
  T* const sR = sm + W::oVR;
  T* const sr = sm + W::oVr;
  T* const sPt = sm + W::oPt;
  const T* const sAl = sm + W::oPt + LD * JB;  // alpha: column JB of the [P | alpha] tile
  T* const sYz = sm + W::oYz;
  T* const sSY = sm + W::oSY;
  T* const sG0 = sm + W::oG;
  T* const sSB = sm + W::oSB;
  int* const sCD = reinterpret_cast<int*>(sm + W::oCD);
  T* const sQL = sm + W::oQL;
  T* const sMu = sm + W::oMu;
  const bool adj = a.ed_out != nullptr;  // ExpectedDecrease by the adjoint recursion, in this sweep
  T ed = T(0);
  int mub = 0;  // which half of sMu holds mu_{k+1}

  // compact rows: array << 24 | offset in the array's row  ->  offset inside the image
  auto cdecode = [&](int code) -> int {
    const int arr = code >> 24, off = code & 0xffffff;
    if (arr == RA_Q) {
      const int i = off / (NX * NX), wd = off - i * NX * NX;
      const int col = wd / NX, row = wd - col * NX;
      return W::oTQ + i * W::TILE + row + LD * col;
    }
    if (arr == RA_A || arr == RA_B) {
      const int col = off / NX, row = off - col * NX;
      return (arr == RA_A ? W::oTA : W::oTB) + row + LD * col;
    }
    return (arr == RA_L ? W::oVl : (arr == RA_R ? W::oVR : W::oVr)) + off;
  };
  const int CWD = a.compact_tab[RC_W];
  const int* const ctab = a.compact_tab + RC_BASE + NP + 1;  // destination of each word, then the constants
  const T* const gC = uniform_ptr(a.compact);
  // row k straight from global memory (before the loop), row by DMA into the staging row, staged row over the image
  auto scatter_sync = [&](int k) {
    for (int c = lane; c < CWD; c += 64) sm[cdecode(ctab[c])] = gC[size_t(k) * CWD + c];
  };
  auto request_row = [&](int k) { dma_g2l<64, false>(gC + size_t(k) * CWD, sSB, CWD * S, lane); };
  auto scatter_staged = [&]() {
    constexpr int WPL = W::kWords / 64;
    T v[WPL];
    int cd[WPL];
#pragma unroll
    for (int q = 0; q < WPL; q++) {
      cd[q] = sCD[lane + 64 * q];
      v[q] = sSB[lane + 64 * q];
    }
#pragma unroll
    for (int q = 0; q < WPL; q++)
      if (cd[q] >= 0) sm[cd[q]] = v[q];
  };
  // (Q_i l_i) of the image's step for ExpectedDecrease: lane group i takes player i, lane j of the group row j
  auto stash_ql = [&](int k) {
    if (g < NP && j < NX) {
      const T* tQi = tQ0 + g * W::TILE;
      T s = T(0);
#pragma unroll
      for (int c = 0; c < NX; c++) s += tQi[j + LD * c] * sl[g * NX + c];
      if (adj)
        sQL[g * 16 + j] = s;
      else
        a.scratch[size_t(k) * SCR + g * NX + j] = s;
    }
  };
  // mu <- q + A^T mu (q = sum_i Q_i l_i from sQL; A of the image; `first`: mu = q): lanes < NX, into the other half
  auto adjoint_step = [&](bool first) {
    if (lane < NX) {
      T q = T(0);
#pragma unroll
      for (int i = 0; i < NP; i++) q += sQL[i * 16 + lane];
      if (!first) {
        const T* mu = sMu + mub * 16;
        T s = T(0);
#pragma unroll
        for (int c = 0; c < NX; c++) s += tA[c + LD * lane] * mu[c];
        q += s;
      }
      sMu[(1 - mub) * 16 + lane] = q;
    }
  };

  // ---- zero the tile padding (and everything else the scatter does not write), once; the constants of the image ----
  for (int e = lane; e < W::ELEMS; e += 64) sm[e] = T(0);
  lds_sync(true);
  for (int c = lane; c < W::kWords; c += 64) sCD[c] = c < CWD ? cdecode(ctab[c]) : -1;
  {
    const int nbg = a.compact_tab[RC_NBG];
    const int* bg = ctab + CWD;
    for (int e = lane; e < nbg; e += 64) {
      const int off = cdecode(bg[RC_BG_WORDS * e]);
      const int kind = bg[RC_BG_WORDS * e + 1];
      sm[off] = kind == RC_DT ? T(a.dt) : (kind == RC_NEG_DT ? T(-a.dt) : T(__int_as_float(bg[RC_BG_WORDS * e + 2])));
    }
  }
  lds_sync(true);

  // ---- terminal step: Z_i = Q_i[T-1], zeta_i = l_i[T-1] (:102-105); zeta_i rides in column JB of the Z_i tile ----
  scatter_sync(Tn - 1);
  lds_sync(true);
  vec Zd[NP];
#pragma unroll
  for (int w = 0; w < NP; w++) {
    Zd[w] = ldD(tQ0 + w * W::TILE);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = row0 + RS * r;
      Zd[w][r] += (row < NX ? sl[w * NX + (row < NX ? row : 0)] : T(0)) * mVecCol;
    }
  }
  stash_ql(Tn - 1);
  for (int e = lane; e < M * NX; e += 64) a.P[size_t(Tn - 1) * M * NX + e] = T(0);  // strategy.h:64-70
  if (lane < M) a.alpha[size_t(Tn - 1) * M + lane] = T(0);
  if (adj) {
    lds_sync(true);
    adjoint_step(true);  // mu_{T-1} = q_{T-1}
    mub = 1 - mub;
  } else {
    if (lane < NP) a.scratch[size_t(Tn - 1) * SCR + NP * NX + lane] = T(0);
    if (lane < NX) a.scratch[size_t(Tn - 1) * SCR + NP * (NX + 1) + lane] = T(0);
  }
  lds_sync(true);
  if (Tn >= 2) {
    scatter_sync(Tn - 2);
    if (Tn >= 3) request_row(Tn - 3);
  }
  lds_sync(true);

#pragma unroll 1
  for (int k = Tn - 2; k >= 0; k--) {
    // ---- every player's MU rows of the stacked Nash system: (B_w^T Z_w) [B | A] (:128-157) ----
    // As in the player-parallel form (ilqg_lq.hpp): only MU columns of G_w = Z_w^T B are used, so they are formed as
    // matrix-vector products on the vector unit — lane (g, j) multiplies its four rows of column j of the Z_w tile with
    // B[row][w MU + aa] and the lane rows are summed with v_permlane swaps — and the MU x (M + NX) block G_w^T [B | A]
    // likewise from the accumulator-layout registers of the B and A tiles (rows_reduce4).  No matrix instruction here.
    static_assert(MU == 2, "rows_reduce4 packs (S, Y) x two controls");
    const vec Bd = ldD(tB);
    const vec Ad = ldD(tA);
#pragma unroll
    for (int w = 0; w < NP; w++) {
      T p0 = T(0), p1 = T(0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const T* bp = tB + (row0 + RS * r) + LD * (w * MU);
        p0 += Zd[w][r] * bp[0];
        p1 += Zd[w][r] * bp[LD];
      }
      T sa, sb;
      permlane32_swap(p0, p1, sa, sb);
      const T tt = sa + sb;
      permlane16_swap(tt, tt, sa, sb);
      const T gcol = sa + sb;  // rows 0, 1: G_w[j][w MU]; rows 2, 3: G_w[j][w MU + 1]
      if ((g & 1) == 0) {
        const int aa = g >> 1;
        sG0[w * MU * 16 + j * MU + aa] = gcol;
        // y_zeta = B_w^T zeta_w + r_ww (:154-157): zeta_w rides in column JB of the Z_w tile
        if (j == JB) sYz[w * MU + aa] = gcol + sr[pr.rg[w][w] + aa];
      }
    }
    lds_sync(true);
    {
      const int aa = g >> 1;
      const bool isY = (g & 1) != 0;
      const int c = isY ? M + j : j;  // column of [S | Y]
#pragma unroll
      for (int w = 0; w < NP; w++) {
        const T* const sGw = sG0 + w * MU * 16;
        T pS[MU], pY[MU];
#pragma unroll
        for (int q = 0; q < MU; q++) pS[q] = pY[q] = T(0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const T* gp = sGw + (row0 + RS * r) * MU;
#pragma unroll
          for (int q = 0; q < MU; q++) {
            const T gval = gp[q];
            pS[q] += gval * Bd[r];
            pY[q] += gval * Ad[r];
          }
        }
        // rows after the reduction: 0 = S(aa 0), 1 = Y(aa 0), 2 = S(aa 1), 3 = Y(aa 1)
        const T tot = rows_reduce4<T>(pS[0], pS[1], pY[0], pY[1]);
        const bool diag = !isY && j / MU == w;  // + R_ww on this player's diagonal block of S (:148-150)
        const T rd = sR[pr.ro[w][w] + aa + MU * (diag ? j - w * MU : 0)];
        const T val = tot + (diag ? rd : T(0));
        if (isY ? j < NX : j < M) sSY[(w * MU + aa) + M * c] = val;
      }
    }
    lds_sync(true);

    // ---- column `lane` of [S | Y]: Gershgorin (:163-176), then the M x M solve (:180) ----
    // (Measured and dropped: keeping the column in the registers of the lane that forms it, player after player — no LDS
    // trip for the Nash system — holds M more values across the row products: 186 -> 256 VGPRs and 320 B of scratch.)
    {
      T col[M], x[M];
      const bool isS = lane < M;
      const T* src = (lane < M + NX) ? sSY + M * lane : sYz;
#pragma unroll
      for (int r = 0; r < M; r++) {
        col[r] = src[r];
        x[r] = T(0);
      }
      {
        T l1 = T(0);
#pragma unroll
        for (int r = 0; r < M; r++) l1 += lq_abs(col[r]);
        const T diag = src[isS ? lane : 0];
        const T radius = l1 - lq_abs(diag);
        const T eval_lo = diag - radius;
        const T bump = (isS && a.adaptive && eval_lo < T(1e-3f)) ? radius + T(1e-3f) : T(0);
#pragma unroll
        for (int r = 0; r < M; r++) col[r] = col[r] + ((r == lane) ? bump : T(0));
      }
      if (a.adaptive)
        lu_solve_columns<T, M>(col, lane, x);
      else
        qr_solve_columns<T, M>(col, lane, x);
      // [P | alpha] in one piece: column lane - M of the tile, alpha in column NX = JB (F = A - B [P | alpha] then carries
      // beta = -B alpha); alpha is read from there too (sAl below)
      if (lane >= M && lane <= M + NX) {
#pragma unroll
        for (int r = 0; r < M; r++) sPt[r + LD * (lane - M)] = x[r];
      }
    }
    lds_sync(true);
    {
      // explicit global address space: through a generic pointer these are FLAT stores, which count on the LDS counter too
      typedef __attribute__((address_space(1))) T gT;
      if (lane < NX) {
        T pv[M];
#pragma unroll
        for (int r = 0; r < M; r++) pv[r] = sPt[r + LD * lane];
        gT* dst = (gT*)(uniform_ptr(a.P + size_t(k) * M * NX)) + unsigned(M * lane);
#pragma unroll
        for (int r = 0; r < M; r++) dst[r] = pv[r];
      } else if (lane < NX + M) {
        ((gT*)(uniform_ptr(a.alpha + size_t(k) * M)))[unsigned(lane - NX)] = sAl[lane - NX];
      }
    }

    // ---- [F | beta] = A - B [P | alpha] (:189-194): once for all players.  Used unmasked on both sides of the products:
    // as a left operand its column JB only produces row JB of the result, and row JB of a Z_w tile never reaches anything
    // (every right operand it meets has a zero row JB: tile padding); column 15 of every operand is zero. ----
    const vec Pd = ldD(sPt);
    vec nBT = ldDT(tB);
#pragma unroll
    for (int r = 0; r < 4; r++) nBT[r] = -nBT[r];
    const vec Fd = tile_xty_blocks<T, kblock_mask<T>(0, M)>(nBT, Pd, Ad);
    // mu_{k+1} by rows (adjoint mode): this lane's rows of it, for beta_k^T mu_{k+1} and A_k^T mu_{k+1}
    T muv[4] = {T(0), T(0), T(0), T(0)};
    if (adj) {
#pragma unroll
      for (int r = 0; r < 4; r++) muv[r] = sMu[mub * 16 + row0 + RS * r];  // entries >= NX are zero
    } else if (j == JB) {
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (row0 + RS * r < NX) a.scratch[size_t(k) * SCR + NP * (NX + 1) + row0 + RS * r] = Fd[r];
    }
    T ctrl = T(0);
    if (lane < NP) {  // alpha_i^T R_ii r_ii, evaluated (alpha^T R) r like Eigen (ilq_solver.cpp:384-386)
      int ro_ii = 0, rg_ii = 0;
#pragma unroll
      for (int e = 0; e < NP; e++) {
        ro_ii = (lane == e) ? pr.ro[e][e] : ro_ii;
        rg_ii = (lane == e) ? pr.rg[e][e] : rg_ii;
      }
      T acc = T(0);
#pragma unroll
      for (int c = 0; c < MU; c++) {
        T aR = T(0);
#pragma unroll
        for (int b = 0; b < MU; b++) aR += sAl[lane * MU + b] * sR[ro_ii + b + MU * c];
        acc += aR * sr[rg_ii + c];
      }
      if (adj)
        ctrl = acc;
      else
        a.scratch[size_t(k) * SCR + NP * NX + lane] = acc;
    }
    if (adj) {
      // ExpectedDecrease: the players' control terms of this step, then beta_k^T mu_{k+1} (beta = column JB of [F | beta])
#pragma unroll
      for (int i = 0; i < NP; i++) ed -= bcast(ctrl, i);
      T bm = T(0);
#pragma unroll
      for (int r = 0; r < 4; r++) bm += Fd[r] * muv[r];
      ed -= bcast(rows_allreduce<T>(bm), JB);
    }

    // ---- Z_w <- F^T Z_w F + Q_w + sum_jj P_jj^T R_w,jj P_jj, zeta_w in column JB (:198-212), player after player ----
    T qsum = T(0);  // adjoint mode: sum_w (Q_w l_w)[j], every lane row
    static_for<NP>([&](auto WW) {
      constexpr int w = decltype(WW)::value;
      vec Cd = ldD(tQ0 + w * W::TILE);
      {
        // Q_w l_w for ExpectedDecrease from the tile just loaded: Q_w is symmetric, so entry j is sum_row Q_w[row][j] l_w[row]
        T part = T(0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = row0 + RS * r;
          part += Cd[r] * (row < NX ? sl[w * NX + (row < NX ? row : 0)] : T(0));
        }
        part = rows_allreduce<T>(part);
        if (adj)
          qsum += part;
        else if (g == 0 && j < NX)
          a.scratch[size_t(k) * SCR + w * NX + j] = part;
      }
      // column JB of C_w: l_w + sum_jj P_jj^T (R_w,jj alpha_jj - r_w,jj), and the state columns' sum_jj P_jj^T (R_w,jj P_jj):
      // ONE product [P]^T [H | q] over the M rows of P, H = blockdiag(R_w,jj) P in the state columns, q in column JB
      vec Qy;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = row0 + RS * r;  // a row of P: player jj = row / MU, control aa = row % MU
        const bool in = row < M;
        const int rowc = in ? row : 0;
        const int jj = rowc / MU, aa = rowc % MU;
        int qw = -1, ro_wj = 0, rg_wj = 0;
#pragma unroll
        for (int f = 0; f < NP; f++)
          if (jj == f) {
            qw = pr.q[w][f];
            ro_wj = pr.ro[w][f];
            rg_wj = pr.rg[w][f];
          }
        T ww = -sr[rg_wj + aa], h = T(0);
#pragma unroll
        for (int b = 0; b < MU; b++) {
          const T rv = sR[ro_wj + aa + MU * b];
          ww += rv * sAl[jj * MU + b];
          h += rv * sPt[(jj * MU + b) + LD * j];
        }
        Qy[r] = (in && qw >= 0) ? ww * mVecCol + h * mCols : T(0);
        const int srow = row < NX ? row : 0;
        Cd[r] += (row < NX ? sl[w * NX + srow] : T(0)) * mVecCol;
      }
      Cd = tile_xty_blocks<T, kblock_mask<T>(0, M)>(Pd, Qy, Cd);
      const vec Wd = tile_xty<T>(Zd[w], Fd, zero4);  // Z_w [F | beta] (Z_w symmetric: its own transpose)
      vec Wz;
#pragma unroll
      for (int r = 0; r < 4; r++) Wz[r] = Wd[r] + Zd[w][r] * mVecCol;  // column JB: zeta_w + Z_w beta
      Zd[w] = tile_xty<T>(Fd, Wz, Cd);  // [Z_w' | F^T (zeta_w + Z_w beta)] + C_w
    });
    if (adj) {
      // mu_k = q_k + A_k^T mu_{k+1} (A of this image, still in place): entry j = sum_row A[row][j] mu_{k+1}[row]
      T am = T(0);
#pragma unroll
      for (int r = 0; r < 4; r++) am += Ad[r] * muv[r];
      const T munew = qsum + rows_allreduce<T>(am);
      if (g == 0 && j < NX) sMu[(1 - mub) * 16 + j] = munew;
      mub = 1 - mub;
    }

    // ---- the image of the next step: the staged compact row over this one, then the row after it requested ----
    if (k >= 1) {
      dma_wait();
      lds_sync(true);
      scatter_staged();
      lds_sync(true);
      if (k >= 2) request_row(k - 2);
    }
    lds_sync(true);
  }
  if (adj && lane == 0) *a.ed_out = ed;
}

}  // namespace ilqg
