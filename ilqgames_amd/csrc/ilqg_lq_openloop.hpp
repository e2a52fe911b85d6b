// ilqg_lq_openloop.hpp — open-loop LQ Nash sweep for ONE game instance per workgroup (gfx950), on the matrix cores.
//
// Computes what LQOpenLoopSolver::Solve computes (src/lq_open_loop_solver.cpp:73-195):
//   backward, k = T-2 .. 0:
//     W_i = R_ii^{-1} B_i^T, w_i = R_ii^{-1} r_ii                       (LDLT of R_ii, :119-126)
//     Lambda = I + sum_i B_i W_i M_i[k+1]                               (:127-128)
//     c = -sum_i B_i (W_i m_i[k+1] + w_i)                               (:134-139)
//     X = Lambda^{-1} A, y = Lambda^{-1} c                              (:131,144,148)
//     M_i[k] = Q_i + A^T M_i[k+1] X,   m_i[k] = l_i + A^T (m_i[k+1] + M_i[k+1] y)   (:141-150)
//   forward, k = 0 .. T-2:
//     x_{k+1} = Lambda_k^{-1} (A x_k + c_k) = X_k x_k + y_k             (:165)
//     alpha_i,k = W_i,k (M_i[k+1] x_{k+1} + m_i[k+1]) + w_i,k           (:169-172);  P == 0
//     costate_i,k = A_k^T (M_i[k+1] x_{k+1} + m_i[k+1])                 (:176; zero at k = T-1, :191)
//
// How it is laid out here.
//   * Homogeneous coordinates.  With n' = n + 1 and
//         Ma_i = [M_i m_i; 0 0],  Xa = [X y; 0 1],  Aa = [A 0; 0 1],  Qa_i = [Q_i l_i; 0 0]
//     the two recursions are ONE matrix recursion  Ma_i[k] = Qa_i + Aa^T (Ma_i[k+1] Xa):  the vector parts ride in
//     column n of products that are computed anyway.
//   * One wavefront per player.  Wave i holds Ma_i (and its transpose) as 16 x 16 accumulator-layout tiles
//     (ilqg_mfma.hpp; n' <= 32: a 2 x 2 block of tiles) and runs its products as chains of v_mfma_*_16x16x4; k blocks
//     whose rows are identically zero are skipped at compile time.  The transpose for the next step goes through the
//     wave's own LDS tile (write the result, read it back transposed) — the same tile the DMA engine fills with
//     Q_i | l_i of the next step once it has been read.
//   * Lambda^{-1} through the matrix inversion lemma.  Lambda = I + B V with B = [B_0 .. B_{N-1}] (n x m) and
//     V = [W_0 M_0; ..] (m x n) is a rank-m update of the identity:
//         Lambda^{-1} = I - B K^{-1} V,  K = I_m + V B          =>   X = A - B Z,  Z = K^{-1} (V A)
//                                                                     y = -B K^{-1} g,  g_i = W_i m_i + w_i
//     (y: c = -B g and V c = -(K - I) g, so K^{-1} V c = -g + K^{-1} g.)  The reference factors the n x n Lambda by
//     Householder QR (:131); here the m x m system K [Z | z] = [V A | g] is solved by Gaussian elimination with partial
//     pivoting, one column per lane (m + n + 1 columns), in wave 0.  Both forms solve the same linear system; the
//     parity tests compare with the reference arithmetic (QR).
//   * A step has two workgroup barriers:  [V_i | g_i] = R_ii^{-1} (B_i^T Ma_i + [0 | r_ii]) (MFMA + LDL^T) and this
//     player's rows of [K | V A]; barrier; elimination (wave 0); barrier; Xa = Aa - [B; 0] [Z | z] (every wave needs
//     it, so every wave computes it), W = Ma Xa, Ma' = Qa + Aa^T W, transpose.  The elimination is a dependent chain of
//     ~5k cycles during which three waves of the instance wait: the kernel is sized (registers, LDS) for three
//     instances per CU so that other instances' products fill the fp64 pipes meanwhile.  (Hoisting the Zt-independent
//     part of the products in front of the solve — Ma' = G - H Zt — was measured and dropped: fp64 MFMA runs at the
//     vector fp64 rate on gfx950, the extra products cost what the overlap saves.)
//   * The forward pass needs, per step, X, y, V = [W_i M_i[k+1]], g and (for the expected decrease) R_ii r_ii and
//     Q_i l_i:  alpha_i,k = V_i x_{k+1} + g_i.  That is the scratch row the backward pass leaves (n^2 + n + m n + 2 m
//     + N n elements; M_i, m_i are appended only when costates are asked for).
#pragma once

#include "ilqg_lq.hpp"

namespace ilqg {

// Leading dimension of an accumulator-layout operand in LDS: lane (g, j) reads element [row(g, r)][j], so the 16
// columns of a tile must fall on different banks.  fp64 (ds_read_b64, 64 banks, lanes 0-31 = g in {0, 1} per LDS
// cycle): LD = 2 * odd puts the 16 columns on the even double-words and g on the odd ones.  fp32: LD odd.
template <typename T>
constexpr int pad_ld(int n) {
  if (sizeof(T) == 8) {
    int l = n;
    while (l % 4 != 2) l++;
    return l;
  }
  return n | 1;
}

template <typename T, int NX, int NP, int MU>
struct OLCfg {
  using C = LQCfg<T, NX, NP, MU>;
  static constexpr int M = NP * MU;
  static constexpr int NT = 64 * NP;  // one wavefront per player
  static constexpr int NH = NX + 1;   // homogeneous dimension
  static constexpr int NTL = (NH + 15) / 16;
  static_assert(NH <= 32, "the homogeneous matrices are held as 2 x 2 tiles at most");
  static_assert(M <= 16, "the rows of V must fit one tile");
  static_assert(M + NX + 1 <= 64, "K [Z | z] = [V A | g] must fit one wavefront");
  static constexpr int LD = pad_ld<T>(NH);
  static constexpr int MAT = (NH * LD + 3) & ~3;  // one padded n' x n' matrix
  static constexpr int LDZ = pad_ld<T>(M);
  // scratch row (global), one per time step: [X | y | V | g | R_ii r_ii | Q_i l_i]  (+ [M_i | m_i] for costates)
  static constexpr int rX = 0;
  static constexpr int ry = rX + NX * NX;
  static constexpr int rV = ry + NX;
  static constexpr int rg = rV + M * NX;
  static constexpr int rRr = rg + M;
  static constexpr int rql = rRr + M;
  static constexpr int ROW = (rql + NP * NX + 3) & ~3;
  static constexpr int rM = ROW;
  static constexpr int rm = rM + NP * NX * NX;
  static constexpr int ROW_FAT = (rm + NP * NX + 3) & ~3;
  // LDS (elements), backward pass.  The accumulator-layout reads of edge tiles touch (and discard) elements up to
  // 31 rows / columns from a matrix base, so the padded matrices come last and SLACK elements follow them.
  static constexpr int BIMG = ((NX * M + 3) & ~3) + C::RMAX + C::rMAX;  // [B | R | r] of one step (double-buffered)
  static constexpr int oV = 0;                    // V (m x n, column-major)
  static constexpr int og = oV + M * NX;          // g (m)
  static constexpr int oRr = og + M;              // R_ii r_ii (m)
  static constexpr int oKA = (oRr + M + 3) & ~3;  // [K | V A] (m x (m + n))
  static constexpr int oBt = oKA + M * (M + NX);  // per player: B_i^T Ma_i bounce (mu x n')
  static constexpr int oZs = (oBt + NP * MU * NH + 3) & ~3;  // [Z | z] (m x n', leading dimension LDZ)
  static constexpr int oB = (oZs + LDZ * NH + 3) & ~3;       // two [B | R | r] images
  static constexpr int oA = oB + 2 * BIMG;        // two Aa images
  static constexpr int oZ = oA + 2 * MAT;         // per player: Qa_i image / transposition tile
  static constexpr int SLACK = (32 + 32 * LD - MAT + 3) & ~3;
  static constexpr int LDS_BWD0 = oZ + NP * MAT + (SLACK > 0 ? SLACK : 0);
  // compact rows (ilqg_common.hpp): two staging rows, the destination of each word (ints), the non-zero constants of
  // the players' tiles (destination ints + values)
  static constexpr int oSB = LDS_BWD0;
  static constexpr int oCD = oSB + 2 * kCompactMaxWords;
  static constexpr int oBGc = oCD + kCompactMaxWords;
  static constexpr int oBGv = oBGc + kCompactMaxBg;
  static constexpr int LDS_BWD = oBGv + kCompactMaxBg;
  static_assert(oZs + 16 + 32 * LDZ <= LDS_BWD && oB + 32 + NX * 32 <= LDS_BWD, "edge-tile reads stay inside the LDS");
  // forward pass (one wave; overlays the backward working set): two staged rows, x_k, x_{k+1}, alpha, it
  static constexpr int fx = 2 * ROW;
  static constexpr int fa = fx + 2 * NX;
  static constexpr int fit = (fa + M + 3) & ~3;
  static constexpr int LDS_FWD = fit + NX;
  static constexpr int LDS_ELEMS = LDS_FWD > LDS_BWD ? LDS_FWD : LDS_BWD;
};

// Elements of one open-loop scratch row from run-time dimensions (OLCfg::ROW / ROW_FAT).
__host__ __device__ constexpr int ol_row_elems(int n, int m, int N, bool fat = false) {
  const int slim = (n * n + n + m * n + 2 * m + N * n + 3) & ~3;
  return fat ? ((slim + N * n * n + N * n + 3) & ~3) : slim;
}

// Solve R y = b for a small SPD block by LDL^T without pivoting (Eigen::LDLT at
// src/lq_open_loop_solver.cpp:124-126; R_ii is diagonally dominant in every config).
template <typename T, int MU>
__device__ __forceinline__ void ldlt_solve(const T* R /* MU x MU col-major */, T (&b)[MU]) {
  T Lm[MU][MU], D[MU], Dinv[MU];  // one reciprocal per pivot (fast_recip: within an ulp of the quotient)
#pragma unroll
  for (int jx = 0; jx < MU; jx++) {
    T dj = R[jx + MU * jx];
#pragma unroll
    for (int k = 0; k < jx; k++) dj -= Lm[jx][k] * Lm[jx][k] * D[k];
    D[jx] = dj;
    Dinv[jx] = fast_recip(dj);
#pragma unroll
    for (int i = jx + 1; i < MU; i++) {
      T s = R[i + MU * jx];
#pragma unroll
      for (int k = 0; k < jx; k++) s -= Lm[i][k] * Lm[jx][k] * D[k];
      Lm[i][jx] = s * Dinv[jx];
    }
  }
#pragma unroll
  for (int i = 0; i < MU; i++)
#pragma unroll
    for (int k = 0; k < i; k++) b[i] -= Lm[i][k] * b[k];
#pragma unroll
  for (int i = 0; i < MU; i++) b[i] *= Dinv[i];
#pragma unroll
  for (int i = MU - 1; i >= 0; i--)
#pragma unroll
    for (int k = i + 1; k < MU; k++) b[i] -= Lm[k][i] * b[k];
}

// Accumulator layout of tile (a, b) of the matrix X with X[R][Cc] = mat[R + ld * Cc] (TR: mat[Cc + ld * R]) for
// R < RL, Cc < CL and zero elsewhere.  `p` is mat plus this lane's offset (tile_lane_offset); every element is then
// p[compile-time constant] — one address register per (matrix, orientation) instead of one per element, which is what
// the step loop would otherwise keep live.  Elements outside [0, RL) x [0, CL) are READ (the address stays inside the
// workgroup's LDS: OLCfg orders its regions for that) and replaced by zero; the range tests fold away for interior
// tiles (a, b, RL, CL are compile-time constants at every call site).
template <typename T, bool TR>
__device__ __forceinline__ int tile_lane_offset(int ld, int g, int j) {
  return TR ? j + ld * Tile<T>::row(g, 0) : Tile<T>::row(g, 0) + ld * j;
}
template <typename T, bool TR>
__device__ __forceinline__ typename Tile<T>::vec ld_tile(const T* p, int ld, int a, int b, int RL, int CL, int g, int j) {
  constexpr int RS = Tile<T>::row(0, 1) - Tile<T>::row(0, 0);
  typename Tile<T>::vec v;
  const bool cok = 16 * b + j < CL;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const bool rok = 16 * a + Tile<T>::row(g, r) < RL;
    const T x = TR ? p[16 * b + ld * (16 * a + RS * r)] : p[16 * a + RS * r + ld * (16 * b)];
    v[r] = (cok && rok) ? x : T(0);
  }
  return v;
}

// k blocks (of the two 16-row k tiles) that a contraction over KD rows has to visit
template <typename T>
constexpr int kd_mask(int KD, int c) {
  const int rows = KD - 16 * c;
  return rows <= 0 ? 0 : kblock_mask<T>(0, rows < 16 ? rows : 16);
}

// a.scratch must hold T_steps rows of OLCfg::ROW elements (ROW_FAT when a.costates).  a.P is written as zero (:96-102).
// Executed by a workgroup of OLCfg::NT threads (one wave per player).
// CMP: the step's [A | B | Q_i | l_i | R | r] come from compact rows (LQArgs::compact, ilqg_common.hpp) instead of the
// dense arrays: the row of step k - 2 is DMA'd into a staging row behind barrier 2 of step k; behind barrier 1 of step
// k - 1 the waves that wait for the elimination scatter its shared words (A, B, R, r) into the images of that parity —
// whose constants were written once — and behind barrier 1 of step k - 2 they clear the players' tiles and scatter the
// Q_i | l_i words into them.
template <typename T, int NX, int NP, int MU, bool CMP = false>
__device__ __forceinline__ void lq_openloop_instance(const LQArgs<T>& a, const PairTable& pt, T* sm) {
  using C = LQCfg<T, NX, NP, MU>;
  using O = OLCfg<T, NX, NP, MU>;
  using TL = Tile<T>;
  using vec = typename TL::vec;
  constexpr int M = O::M, NT = O::NT, NH = O::NH, NTL = O::NTL, LD = O::LD, LDZ = O::LDZ, MAT = O::MAT;
  constexpr int S = int(sizeof(T));
  constexpr int BOFF_R = (NX * M + 3) & ~3, BOFF_r = BOFF_R + C::RMAX;
  const int t = threadIdx.x;
  const int w = t >> 6;  // wave = player
  const int wp = w;
  const int lane = t & 63, g = lane >> 4, j = lane & 15;
  const int Tn = a.T_steps;
  const int ROWS = a.costates ? O::ROW_FAT : O::ROW;
  const PairRegs<NP> pr(pt);
  const vec zero4 = {T(0), T(0), T(0), T(0)};
  int ro_ww = 0, rg_ww = 0;  // this player's offsets in the R / r rows
#pragma unroll
  for (int e = 0; e < NP; e++) {
    ro_ww = (wp == e) ? pr.ro[e][e] : ro_ww;
    rg_ww = (wp == e) ? pr.rg[e][e] : rg_ww;
  }
  T* const sZ = sm + O::oZ + wp * MAT;  // this player's tile
  T* const sV = sm + O::oV;
  T* const sg = sm + O::og;
  T* const sRr = sm + O::oRr;
  T* const sKA = sm + O::oKA;
  T* const sZs = sm + O::oZs;
  T* const sBt = sm + O::oBt + wp * (MU * NH);
  auto bimg = [&](int which) { return sm + O::oB + which * O::BIMG; };
  auto aimg = [&](int which) { return sm + O::oA + which * MAT; };
  auto row_of = [&](int k) { return a.scratch + size_t(k) * ROWS; };

  // ---- DMA plumbing: columns of NX elements (contiguous in global memory) into padded columns of LD elements ----
  // Piece p of the padded image is column p / PPC, offset p % PPC; one wave moves a whole matrix.
  constexpr int PS = ((NX * S) % 16 == 0 && (LD * S) % 16 == 0) ? 16 : 4;  // DMA piece (bytes)
  constexpr int PPC = LD * S / PS;                                          // pieces of a padded column
  constexpr int VPC = NX * S / PS;                                          // of which carry data
  // The per-lane source offsets are the same every step: worked out once (the division stays out of the step loop).
  constexpr int WI = (NX * PPC + 63) / 64;  // DMA instructions of an NX-column image
  int plan[WI];                              // byte offset of this lane's piece of instruction h in the source, or -1
#pragma unroll
  for (int h = 0; h < WI; h++) {
    const int p = h * 64 + lane;
    const int c = p / PPC, inb = p % PPC;
    plan[h] = (p < NX * PPC && inb < VPC) ? c * NX * S + inb * PS : -1;
  }
  auto dma_piece = [&](const char* s, char* d) {  // d is wave-uniform; the hardware adds lane * PS
    if constexpr (PS == 16)
      __builtin_amdgcn_global_load_lds((glb_vptr)s, (lds_vptr)d, 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((glb_vptr)s, (lds_vptr)d, 4, 0, 0);
  };
  auto dma_matrix = [&](const T* src, T* dst) {  // NX columns
#pragma unroll
    for (int h = 0; h < WI; h++)
      if (plan[h] >= 0) dma_piece(reinterpret_cast<const char*>(src) + plan[h], reinterpret_cast<char*>(dst) + h * 64 * PS);
  };
  auto dma_column = [&](const T* src, T* dst) {  // one column
    static_assert(VPC <= 64, "a column is one DMA instruction");
    if (lane < VPC) dma_piece(reinterpret_cast<const char*>(src) + lane * PS, reinterpret_cast<char*>(dst));
  };
  // player wave: Q_i | l_i of step k into its tile (columns 0..NX-1 and column NX)
  auto issue_Q = [&](int k) {
    dma_matrix(a.Q + (size_t(k) * NP + wp) * NX * NX, sZ);
    dma_column(a.l + (size_t(k) * NP + wp) * NX, sZ + LD * NX);
  };
  // A and [B | R | r] of step k into the images of its parity; the waves share the instructions.
  auto dma_matrix_shared = [&](const T* src, T* dst) {
#pragma unroll
    for (int h = 0; h < WI; h++)
      if (h % NP == wp && plan[h] >= 0) dma_piece(reinterpret_cast<const char*>(src) + plan[h], reinterpret_cast<char*>(dst) + h * 64 * PS);
  };
  auto issue_shared = [&](int k) {
    dma_matrix_shared(a.A + size_t(k) * NX * NX, aimg(k & 1));
    T* img = bimg(k & 1);
    const int tp = wp * 64 + lane;
    dma_g2l<64 * NP, false>(a.Bm + size_t(k) * NX * M, img, NX * M * S, tp);
    dma_g2l<64 * NP, false>(a.R + size_t(k) * pt.Rsz, img + BOFF_R, pt.Rsz * S, tp);
    dma_g2l<64 * NP, false>(a.r + size_t(k) * pt.rsz, img + BOFF_r, pt.rsz * S, tp);
  };
  auto lds_drain = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  // ---- compact rows (CMP) ----
  // A word's destination: space << 16 | offset; space 0 = the A image, 1 = the [B | R | r] image, 2 + i = player i's tile.
  T* const sSB = sm + O::oSB;
  int* const sCD = reinterpret_cast<int*>(sm + O::oCD);
  int* const sBGc = reinterpret_cast<int*>(sm + O::oBGc);
  T* const sBGv = sm + O::oBGv;
  const int CWD = CMP ? a.compact_tab[RC_W] : 0;
  int nbg_tile = 0;  // entries of (sBGc, sBGv): the tiles' non-zero constants
  auto cdecode = [&](int code) -> int {
    const int arr = code >> 24, off = code & 0xffffff;
    if (arr == RA_A) return (0 << 16) | ((off % NX) + LD * (off / NX));
    if (arr == RA_B) return (1 << 16) | off;
    if (arr == RA_R) return (1 << 16) | (BOFF_R + off);
    if (arr == RA_r) return (1 << 16) | (BOFF_r + off);
    if (arr == RA_Q) {
      const int i = off / (NX * NX), wd = off - i * NX * NX;
      return ((2 + i) << 16) | ((wd % NX) + LD * (wd / NX));
    }
    const int i = off / NX;  // RA_L
    return ((2 + i) << 16) | ((off - i * NX) + LD * NX);
  };
  auto crow_dma = [&](int k) {  // compact row k -> its staging row, all waves share the pieces
    dma_g2l<64 * NP, false>(a.compact + size_t(k) * CWD, sSB + (k & 1) * kCompactMaxWords, CWD * S, wp * 64 + lane);
  };
  // the shared words (A, B, R, r) of staged row k into the images of its parity: every thread its words
  auto scatter_shared = [&](int k) {
    const T* row = sSB + (k & 1) * kCompactMaxWords;
    for (int c = t; c < CWD; c += NT) {
      const int code = sCD[c], sp = code >> 16, off = code & 0xffff;
      const T v = row[c];
      if (sp == 0) aimg(k & 1)[off] = v;
      if (sp == 1) bimg(k & 1)[off] = v;
    }
  };
  // tiles lo .. hi <- Qa_i of staged row k: cleared, their constants, their words (one wave)
  auto fill_tiles = [&](int k, int lo, int hi) {
    T* const z0 = sm + O::oZ + lo * MAT;
    for (int e = lane; e < (hi - lo + 1) * MAT; e += 64) z0[e] = T(0);
    lds_sync(true);
    for (int e = lane; e < nbg_tile; e += 64) {
      const int code = sBGc[e], sp = (code >> 16) - 2;
      if (sp >= lo && sp <= hi) sm[O::oZ + sp * MAT + (code & 0xffff)] = sBGv[e];
    }
    const T* row = sSB + (k & 1) * kCompactMaxWords;
    for (int c = lane; c < CWD; c += 64) {
      const int code = sCD[c], sp = (code >> 16) - 2;
      if (sp >= lo && sp <= hi) sm[O::oZ + sp * MAT + (code & 0xffff)] = row[c];
    }
  };

  // this lane's offsets into accumulator-layout operands (plain / transposed) for the three leading dimensions
  const int oD = tile_lane_offset<T, false>(LD, g, j), oT = tile_lane_offset<T, true>(LD, g, j);
  const int oDn = tile_lane_offset<T, false>(NX, g, j), oTn = tile_lane_offset<T, true>(NX, g, j);
  const int oDz = tile_lane_offset<T, false>(LDZ, g, j);
  constexpr int RS = TL::row(0, 1) - TL::row(0, 0);

  // ---- block algebra on NTL x NTL tiles ----
  struct Blk {
    vec v[NTL][NTL];
  };
  auto load_blk = [&](const T* mat, bool transposed) {  // D(mat) / D(mat^T) of a padded n' x n' matrix
    Blk o;
#pragma unroll
    for (int aa = 0; aa < NTL; aa++)
#pragma unroll
      for (int bb = 0; bb < NTL; bb++)
        o.v[aa][bb] = transposed ? ld_tile<T, true>(mat + oT, LD, aa, bb, NH, NH, g, j) : ld_tile<T, false>(mat + oD, LD, aa, bb, NH, NH, g, j);
    return o;
  };
  constexpr int KH0 = kd_mask<T>(NH, 0), KH1 = kd_mask<T>(NH, 1);  // contraction over n' rows
  constexpr int KN0 = kd_mask<T>(NX, 0), KN1 = kd_mask<T>(NX, 1);  // over n rows
  constexpr int KM0 = kd_mask<T>(M, 0);                            // over m rows

  // ---- once per sweep: zero the tiles' padding, the homogeneous column of both Aa images ----
  static_assert(O::oZ == O::oA + 2 * MAT, "the Aa images and the tiles are zeroed in one piece");
  for (int e = t; e < (NP + 2) * MAT; e += NT) sm[O::oA + e] = T(0);
  if constexpr (CMP) {
    for (int e = t; e < 2 * O::BIMG; e += NT) sm[O::oB + e] = T(0);
    for (int c = t; c < kCompactMaxWords; c += NT) sCD[c] = c < CWD ? cdecode(a.compact_tab[RC_BASE + NP + 1 + c]) : -1;
  }
  lds_sync(false);
  if (t < 2) aimg(t)[NX + LD * NX] = T(1);
  if constexpr (CMP) {
    // the constants: those of A and B go into both images once; those of the tiles into the LDS list fill_tile reads
    const int nbg = a.compact_tab[RC_NBG];
    const int* bg = a.compact_tab + RC_BASE + NP + 1 + CWD;
    int nt = 0;
    for (int e = 0; e < nbg; e++) {  // every thread walks the (short) list: the tile entries keep their list order
      const int code = cdecode(bg[RC_BG_WORDS * e]), kind = bg[RC_BG_WORDS * e + 1];
      const T v = kind == RC_DT ? T(a.dt) : (kind == RC_NEG_DT ? T(-a.dt) : T(__int_as_float(bg[RC_BG_WORDS * e + 2])));
      const int sp = code >> 16, off = code & 0xffff;
      if (sp == 0 && t < 2) aimg(t)[off] = v;
      if (sp == 1 && t < 2) bimg(t)[off] = v;
      if (sp >= 2) {
        if (t == 0 && nt < kCompactMaxBg) {
          sBGc[nt] = code;
          sBGv[nt] = v;
        }
        nt++;
      }
    }
    nbg_tile = nt < kCompactMaxBg ? nt : kCompactMaxBg;
  }
  lds_sync(false);

  // ---- terminal step (:105-108): Ma_i = Qa_i[T-1] ----
  if constexpr (CMP) {
    crow_dma(Tn - 1);
    if (Tn >= 2) crow_dma(Tn - 2);
    dma_wait();
    lds_sync(false);
    fill_tiles(Tn - 1, wp, wp);
    if (Tn >= 2) scatter_shared(Tn - 2);
    lds_sync(false);
    if (Tn >= 3) crow_dma(Tn - 3);  // into row T-1's staging row, whose words have all been placed
    dma_wait();
  } else {
    issue_Q(Tn - 1);
    if (Tn >= 2) issue_shared(Tn - 2);
    dma_wait();
  }
  lds_sync(false);
  // Q_i l_i of a step (expected decrease, ilq_solver.cpp:392) -> scratch row, and M_i, m_i when costates are wanted;
  // both read this wave's tile
  auto store_row_from_tile = [&](int k, bool tile_holds_Q) {
    T* row = row_of(k);
    if (tile_holds_Q && lane < NX) {
      T s = T(0);
#pragma unroll
      for (int c = 0; c < NX; c++) s += sZ[lane + LD * c] * sZ[c + LD * NX];
      row[O::rql + wp * NX + lane] = s;
    }
    if (!tile_holds_Q && a.costates) {
      for (int e = lane; e < NX * NX; e += 64) row[O::rM + wp * NX * NX + e] = sZ[(e % NX) + LD * (e / NX)];
      if (lane < NX) row[O::rm + wp * NX + lane] = sZ[lane + LD * NX];
    }
  };

  // Structural zeros of block-diagonal dynamics (LQArgs::nsub): which k blocks of the two k tiles can be non-zero in
  //   Aa^T W, output row tile aa: the rows of the subsystems that reach into rows [16 aa, 16 aa + 16)
  // (wave-uniform; everything allowed when the caller gave no structure).
  int mA[2][2] = {{15, 15}, {15, 15}};  // [k tile][aa]
  if (a.nsub > 0) {
#pragma unroll
    for (int aa = 0; aa < 2; aa++) {
      int lo = NX, hi = 0;
#pragma unroll
      for (int i = 0; i < NP; i++) {  // (compile-time indices: a run-time-indexed member would put the arguments in scratch)
        const int x0 = a.xoff[i], x1 = a.xoff[i + 1];
        if (i < a.nsub && x1 > 16 * aa && x0 < 16 * aa + 16) {
          lo = x0 < lo ? x0 : lo;
          hi = x1 > hi ? x1 : hi;
        }
      }
#pragma unroll
      for (int c = 0; c < 2; c++) mA[c][aa] = __builtin_amdgcn_readfirstlane(kblock_mask_rt<T>(c, lo, hi));
    }
  }

  long long ph_c = (kProfile && a.ph) ? clock64() : 0;
  auto PH = [&](int slot) {
    if (kProfile && a.ph && t == 0) {
      const long long c = clock64();
      a.ph[slot] += c - ph_c;
      ph_c = c;
    }
  };

  // (M_i of an open-loop Nash game is NOT symmetric — M_i = Q_i + A^T M_i Lambda^{-1} A with the other players' terms in
  // Lambda — so both D(Ma_i) and D(Ma_i^T) are kept; replacing one by the other was tried in round 5 and is wrong.)
  Blk Md = load_blk(sZ, false);  // D(Ma_i)
  Blk MT = load_blk(sZ, true);   // D(Ma_i^T)
  store_row_from_tile(Tn - 1, true);
  if (a.costates) store_row_from_tile(Tn - 1, false);  // M[T-1] = Q[T-1]: the tile holds both
  lds_sync(true);
  lds_drain();  // the tile has been read: the DMA engine may refill it
  if constexpr (!CMP) {
    if (Tn >= 2) issue_Q(Tn - 2);
  }
#pragma unroll 1
  for (int k = Tn - 2; k >= 0; k--) {
    const T* sB = bimg(k & 1);
    const T* sR = sB + BOFF_R;
    const T* sr = sB + BOFF_r;
    const T* sAa = aimg(k & 1);
    // ---- [V_i | g_i] = R_ii^{-1} (B_i^T Ma_i + [0 | r_ii]),  R_ii r_ii ----
    {
      if constexpr (MU == 2 && NTL <= 2) {
        // B_i^T Ma_i has MU = 2 rows: as a tile product fourteen of sixteen output rows are zeros.  Row sums instead
        // (ilqg_mfma.hpp): lane (g, j) multiplies its rows of column j of every column tile of Ma_i with
        // B_i[row][aa] (from the image: the address depends on g only), and the (aa, bb) partial sums are reduced over the
        // four lane rows with v_permlane swaps — twelve matrix instructions less per step at n = 24.
        T pp[2][2] = {{T(0), T(0)}, {T(0), T(0)}};  // [aa][bb]
#pragma unroll
        for (int c = 0; c < NTL; c++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 16 * c + TL::row(g, r);
            const bool rok = row < NX;
            const T* bp = sB + NX * (w * MU) + (rok ? row : 0);
            const T b0 = rok ? bp[0] : T(0), b1 = rok ? bp[NX] : T(0);
#pragma unroll
            for (int bb = 0; bb < NTL; bb++) {
              pp[0][bb] += Md.v[c][bb][r] * b0;
              pp[1][bb] += Md.v[c][bb][r] * b1;
            }
          }
        // rows after the reduction: 0 = (aa 0, bb 0), 1 = (aa 0, bb 1), 2 = (aa 1, bb 0), 3 = (aa 1, bb 1)
        const T tot = rows_reduce4<T>(pp[0][0], pp[1][0], pp[0][1], pp[1][1]);
        const int col = 16 * (g & 1) + j;
        if (((g & 1) < NTL) && col < NH) sBt[(g >> 1) + MU * col] = tot;
      } else {
      vec Bd[NTL];  // D(B_i): n x mu
#pragma unroll
      for (int c = 0; c < NTL; c++) Bd[c] = ld_tile<T, false>(sB + NX * (w * MU) + oDn, NX, c, 0, NX, MU, g, j);
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        vec acc = tile_xty_blocks<T, KN0>(Bd[0], Md.v[0][bb], zero4);
        if constexpr (NTL == 2) acc = tile_xty_blocks<T, KN1>(Bd[1], Md.v[1][bb], acc);
        const int col = 16 * bb + j;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = TL::row(g, r);
          if (row < MU && col < NH) sBt[row + MU * col] = acc[r];
        }
      }
      }
      lds_sync(true);
      if (lane < NH) {
        T b[MU];
#pragma unroll
        for (int aa = 0; aa < MU; aa++) b[aa] = sBt[aa + MU * lane] + (lane == NX ? sr[rg_ww + aa] : T(0));
        ldlt_solve<T, MU>(sR + ro_ww, b);
#pragma unroll
        for (int aa = 0; aa < MU; aa++) {
          if (lane < NX)
            sV[(w * MU + aa) + M * lane] = b[aa];
          else
            sg[w * MU + aa] = b[aa];
        }
      } else if (lane < NH + MU) {
        const int aa = lane - NH;
        T s = T(0);
#pragma unroll
        for (int c = 0; c < MU; c++) s += sR[ro_ww + aa + MU * c] * sr[rg_ww + c];
        sRr[w * MU + aa] = s;
      }
      lds_sync(true);
      // this player's rows of [K | V A] = [I + V B | V A]
      for (int e = lane; e < MU * (M + NX); e += 64) {
        const int q = w * MU + e % MU, c = e / MU;
        const T* colp = c < M ? sB + NX * c : sAa + LD * (c - M);
        T s = (c == q) ? T(1) : T(0);
#pragma unroll
        for (int r = 0; r < NX; r++) s += sV[q + M * r] * colp[r];
        sKA[q + M * c] = s;
      }
    }
    if constexpr (CMP) dma_wait();  // this wave's pieces of compact row k - 1 (requested behind barrier 2 of step k + 1)
    lds_sync(NT <= 64);  // barrier 1: [K | V A], g complete
    PH(0);
    // (Rotating the solving wave over the players, so that the elimination's vector instructions load the four SIMDs
    // evenly, was measured in round 5: no difference — the sweep is not bound by the solving wave's SIMD.)
    const int rel = wp;  // 0: solves this step; 1 .. NP-1: helpers
    // Every wave has left step k + 1 behind: the images of the other parity (last read there) are free for the next
    // step's A, [B | R | r], and every tile has been read back transposed: free for Qa_i of this step.
    if constexpr (CMP) {
      // By the waves that wait through the elimination (the solving wave goes straight to its columns): the shared words
      // of row k - 1 (staged during step k + 1, waited for in front of barrier 1) into the images, and the tiles' words of
      // row k — wave 1 fills the solving wave's tile too.  (Until round 5 every wave filled its own tile at the end of
      // the previous step, on the way to barrier 1; here it costs the instance nothing.)  Row k - 2 goes into row k's
      // staging row behind barrier 2, when the tiles' words have been taken out of it.
      if constexpr (NP > 1) {
        if (rel != 0) {
          const int ht = (rel - 1) * 64 + lane;  // thread index among the helpers
          if (k > 0) {
            const T* row = sSB + ((k - 1) & 1) * kCompactMaxWords;
            for (int c = ht; c < CWD; c += NT - 64) {
              const int code = sCD[c], sp = code >> 16, off = code & 0xffff;
              const T v = row[c];
              if (sp == 0) aimg((k - 1) & 1)[off] = v;
              if (sp == 1) bimg((k - 1) & 1)[off] = v;
            }
          }
          fill_tiles(k, rel == 1 ? 0 : wp, wp);
        }
      } else {
        if (k > 0) scatter_shared(k - 1);
        fill_tiles(k, 0, 0);
      }
    } else {
      if (k > 0) issue_shared(k - 1);
    }
    // [V | g | R_ii r_ii] -> scratch row k (forward pass): by the last wave, which waits through the elimination anyway
    // (it was the solving wave's job until round 5: four LDS-read / store trips at the head of the step's longest chain)
    if (rel == NP - 1) {
      static_assert(O::og == O::oV + M * NX && O::oRr == O::og + M && O::rg == O::rV + M * NX && O::rRr == O::rg + M,
                    "[V | g | R r] is copied to the scratch row in one piece");
      T* row = row_of(k);
      for (int e = lane; e < M * NX + 2 * M; e += 64) row[O::rV + e] = sV[e];
    }
    // ---- K [Z | z] = [V A | g] (the step's solving wave, column per lane) ----
    if (rel == 0) {
      T col[M], x[M];
#pragma unroll
      for (int q = 0; q < M; q++) {
        col[q] = lane < M + NX ? sKA[q + M * (lane < M + NX ? lane : 0)] : (lane == M + NX ? sg[q] : T(0));
        x[q] = T(0);
      }
      {
        // Partial pivoting never exchanges rows of a matrix whose columns are strictly diagonally dominant (elimination
        // keeps the property for every Schur complement), so for such a K the pivot searches and row exchanges — more
        // than a third of the chain's instructions — are skipped and the result is the same to the last bit.
        T l1 = T(0);
#pragma unroll
        for (int q = 0; q < M; q++) l1 += lq_abs(col[q]);
        const T dg = lq_abs(sKA[(lane < M ? lane : 0) * (M + 1)]);
        const bool dominant = __all(lane >= M || dg > l1 - dg);
        if (dominant)
          lu_solve_columns<T, M>(col, lane, x);
        else
          lu_pp_solve_columns<T, M>(col, lane, x);
      }
      if (lane >= M && lane <= M + NX) {
#pragma unroll
        for (int q = 0; q < M; q++) sZs[q + LDZ * (lane - M)] = x[q];
      }
    }
    PH(1);
    dma_wait();  // dense arrays: this wave's share of the next step's images, and Q_i | l_i of this step (issued at the
                 // end of the previous one) in its tile
    lds_sync(NT <= 64);  // barrier 2: [Z | z], the next step's images and this step's tiles published
    PH(2);
    if constexpr (CMP) {
      if (k > 1) crow_dma(k - 2);  // into row k's staging row
    }
    // ---- Xa = Aa - Bt Zt  (every wave) ----
    Blk Xd;
    {
      vec nBT[NTL], Zd[NTL];  // D(-B^T) (m x n), D([Z | z]) (m x n')
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        nBT[bb] = ld_tile<T, true>(sB + oTn, NX, 0, bb, M, NX, g, j);
#pragma unroll
        for (int r = 0; r < 4; r++) nBT[bb][r] = -nBT[bb][r];
        Zd[bb] = ld_tile<T, false>(sZs + oDz, LDZ, 0, bb, M, NH, g, j);
      }
#pragma unroll
      for (int aa = 0; aa < NTL; aa++)
#pragma unroll
        for (int bb = 0; bb < NTL; bb++)
          Xd.v[aa][bb] = tile_xty_blocks<T, KM0>(nBT[aa], Zd[bb], ld_tile<T, false>(sAa + oD, LD, aa, bb, NH, NH, g, j));
    }
    // ---- W = Ma Xa ----
    Blk Wd;
#pragma unroll
    for (int aa = 0; aa < NTL; aa++)
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        vec acc = tile_xty_blocks<T, KH0>(MT.v[0][aa], Xd.v[0][bb], zero4);
        if constexpr (NTL == 2) acc = tile_xty_blocks<T, KH1>(MT.v[1][aa], Xd.v[1][bb], acc);
        Wd.v[aa][bb] = acc;
      }
    // X, y -> scratch row k (the waves share the tiles), Q_i l_i.  Stored after this step's wait for the DMA, so that the
    // stores have a whole step to drain before the next one.
    {
      T* row = row_of(k);
#pragma unroll
      for (int aa = 0; aa < NTL; aa++)
#pragma unroll
        for (int bb = 0; bb < NTL; bb++)
          if ((aa * NTL + bb) % NP == w) {
            const int col = 16 * bb + j;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int rw = 16 * aa + TL::row(g, r);
              if (rw < NX && col <= NX) row[(col < NX ? O::rX + NX * col : O::ry) + rw] = Xd.v[aa][bb][r];
            }
          }
    }
    // Q_i l_i (expected decrease): with symmetric costs entry j is sum_row Q_i[row][j] l_i[row] — row sums of the D(Qa_i)
    // tiles this update loads anyway (l_i is their column NX), reduced over the lane rows below; otherwise the row-by-
    // row products from the tile
    const bool ql_rows = a.symmetric != 0 && NTL <= 2;
    if (!ql_rows) store_row_from_tile(k, true);
    T qlp[2] = {T(0), T(0)};
    // ---- Ma' = Qa + Aa^T W ----
#pragma unroll
    for (int aa = 0; aa < NTL; aa++)
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        vec acc = ld_tile<T, false>(sZ + oD, LD, aa, bb, NH, NH, g, j);  // D(Qa_i)
        if (ql_rows) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 16 * aa + TL::row(g, r);
            qlp[bb] += acc[r] * (row < NX ? sZ[(row < NX ? row : 0) + LD * NX] : T(0));
          }
        }
        acc = tile_xty_blocks_rt<T, KN0>(ld_tile<T, false>(sAa + oD, LD, 0, aa, NH, NH, g, j), Wd.v[0][bb], acc, mA[0][aa]);
        if constexpr (NTL == 2)
          acc = tile_xty_blocks_rt<T, KN1>(ld_tile<T, false>(sAa + oD, LD, 1, aa, NH, NH, g, j), Wd.v[1][bb], acc, mA[1][aa]);
        Md.v[aa][bb] = acc;
      }
    if (ql_rows) {
      const T tot = rows_reduce4<T>(qlp[0], T(0), qlp[1], T(0));  // row 0: columns 0 .. 15, row 1: columns 16 ..
      const int col = 16 * g + j;
      if (g < NTL && col < NX) row_of(k)[O::rql + wp * NX + col] = tot;
    }
    PH(3);
    // transpose through this wave's tile: write D(Ma'), read D(Ma'^T)
    lds_sync(true);  // every read of Qa_i is done
#pragma unroll
    for (int aa = 0; aa < NTL; aa++)
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        const int col = 16 * bb + j;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int rw = 16 * aa + TL::row(g, r);
          if (rw < NH && col < NH) sZ[oD + 16 * aa + RS * r + LD * (16 * bb)] = Md.v[aa][bb][r];
        }
      }
    lds_sync(true);
    MT = load_blk(sZ, true);
    if (a.costates) store_row_from_tile(k, false);
    lds_sync(true);
    lds_drain();  // the tile has been read: the DMA engine may refill it
    if constexpr (!CMP) {
      if (k > 0) issue_Q(k - 1);
    }
    PH(4);
  }

  // ---- forward pass (:156-192), wave 0 ----
  for (int e = t; e < M * NX; e += NT)
    for (int k = 0; k < Tn; k++) a.P[size_t(k) * M * NX + e] = T(0);  // open loop: P stays zero
  __syncthreads();  // scratch rows were written by other waves
  // ---- forward pass, two stages (round 5; the one-wave step-by-step pass below stays for costates and for horizons whose
  // state history does not fit the LDS).  The only sequential part is x_{k+1} = X_k x_k + y_k: stage 1 runs it alone (wave
  // 0: the 2 x 32 lanes split each row's terms in two, X_k | y_k prefetched kFwdDepth steps ahead from the scratch rows
  // into registers, x_k handed on through the LDS history) — about 300 cycles per step instead of the 6000 of the pass
  // that also formed alpha, the expected decrease and the stores inside the chain.  Stage 2 is parallel over the steps
  // and runs on all waves: alpha_k = V_k x_{k+1} + g_k, the expected-decrease terms, the stores.
  constexpr int kFwdDepth = 4;
  constexpr int FCH = (NX + 1) / 2;
  T* const xh = sm;  // x_k, k = 0 .. T-1 (overlays the backward working set)
  const int red_off = (Tn * NX + 3) & ~3;
  if (!a.costates && red_off + NP <= O::LDS_ELEMS && NX <= 32) {
    if (w == 0) {
      const int h = lane >> 5, i = lane & 31, ii = i < NX ? i : 0, c0 = h * FCH;
      if (lane < NX) xh[lane] = a.x0 ? a.x0[lane] : T(0);
      T xb[kFwdDepth][FCH], yb[kFwdDepth];
      auto fetch = [&](int k, T (&xs_)[FCH], T& y_) {
        const T* row = row_of(k);
#pragma unroll
        for (int c = 0; c < FCH; c++) {
          const int cc = c0 + c;
          const T v = row[O::rX + ii + NX * (cc < NX ? cc : 0)];
          xs_[c] = cc < NX ? v : T(0);
        }
        const T yv = row[O::ry + ii];
        y_ = h == 0 ? yv : T(0);
      };
#pragma unroll
      for (int d = 0; d < kFwdDepth; d++)
        if (d < Tn - 1) fetch(d, xb[d], yb[d]);
      lds_sync(true);
#pragma unroll 1
      for (int kb = 0; kb < Tn - 1; kb += kFwdDepth) {
#pragma unroll
        for (int d = 0; d < kFwdDepth; d++) {
          const int k = kb + d;
          if (k < Tn - 1) {
            const T* xk = xh + k * NX;
            T s0 = yb[d], s1 = T(0);
#pragma unroll
            for (int c = 0; c < FCH; c++) {
              const int cc = c0 + c;
              const T xv = xk[cc < NX ? cc : 0];
              if (c & 1)
                s1 += xb[d][c] * xv;
              else
                s0 += xb[d][c] * xv;
            }
            const T sh = s0 + s1;
            T lo_, hi_;
            permlane32_swap(sh, sh, lo_, hi_);  // [rows 0 1 | rows 0 1], [rows 2 3 | rows 2 3]
            if (lane < NX) xh[(k + 1) * NX + lane] = lo_ + hi_;
            if (k + kFwdDepth < Tn - 1) fetch(k + kFwdDepth, xb[d], yb[d]);
            lds_sync(true);
          }
        }
      }
    }
    __syncthreads();  // the state history is complete
    T edp = T(0);
    constexpr int SL = NT / M;  // steps per pass of the workgroup
    const int ks = t / M, q = t % M;
    if (ks < SL) {
#pragma unroll 1
      for (int k = ks; k < Tn - 1; k += SL) {
        const T* row = row_of(k);
        const T* xn = xh + (k + 1) * NX;
        const T* xk = xh + k * NX;
        T s0 = row[O::rg + q], s1 = T(0);
#pragma unroll
        for (int c = 0; c < NX; c++) {
          if (c & 1)
            s1 += row[O::rV + q + M * c] * xn[c];
          else
            s0 += row[O::rV + q + M * c] * xn[c];
        }
        const T al = s0 + s1;
        a.alpha[size_t(k) * M + q] = al;
        if (a.ed_out) {  // ILQSolver::ExpectedDecrease (ilq_solver.cpp:364-398): this lane's share of the step's terms
          T e = al * row[O::rRr + q];  // alpha_i^T (R_ii r_ii), (:384-386)
          if (k > 0) {
#pragma unroll
            for (int it = 0; it < (NP * NX + M - 1) / M; it++) {  // delta_x^T Q_i l_i (:392)
              const int c = q + M * it;
              if (c < NP * NX) e += xk[c % NX] * row[O::rql + c];
            }
          }
          edp -= e;
        }
        if (a.dx) {
          for (int c = q; c < NX; c += M) a.dx[size_t(k) * NX + c] = xk[c];
        }
      }
    }
    // step T-1 (:188-192): alpha = 0, the state term delta_x^T Q l
    const T* xl = xh + (Tn - 1) * NX;
    if (t < M) a.alpha[size_t(Tn - 1) * M + t] = T(0);
    if (a.dx && t < NX) a.dx[size_t(Tn - 1) * NX + t] = xl[t];
    if (a.ed_out) {
      if (Tn > 1 && t < NP * NX) edp -= xl[t % NX] * row_of(Tn - 1)[O::rql + t];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) edp += __shfl_xor(edp, off, 64);
      T* const red = sm + red_off;
      if (lane == 0) red[w] = edp;
      __syncthreads();
      if (t == 0) {
        T tot = T(0);
#pragma unroll
        for (int e = 0; e < NP; e++) tot += red[e];
        *a.ed_out = tot;
      }
    }
    return;
  }
  if (w != 0) return;
  constexpr int ROW = O::ROW;
  T* const sx = sm + O::fx;    // x_k
  T* const sxn = sx + NX;      // x_{k+1}
  T* const sal = sm + O::fa;   // alpha_k
  T* const sit = sm + O::fit;  // M_i[k+1] x_{k+1} + m_i[k+1] (costates)
  auto stage = [&](int k, int which) { dma_g2l<64, false>(row_of(k), sm + which * ROW, ROW * S, lane); };
  if (lane < NX) sx[lane] = a.x0 ? a.x0[lane] : T(0);
  T ed = T(0);
  int cur = 0;
  if (Tn >= 2) stage(0, 0);
  dma_wait();
  lds_sync(true);
#pragma unroll 1
  for (int k = 0; k < Tn - 1; k++) {
    if (k + 2 < Tn) stage(k + 1, 1 - cur);
    const T* fr = sm + cur * ROW;
    if (a.dx && lane < NX) a.dx[size_t(k) * NX + lane] = sx[lane];
    if (lane < NX) {
      T s = fr[O::ry + lane];
#pragma unroll
      for (int c = 0; c < NX; c++) s += fr[O::rX + lane + NX * c] * sx[c];
      sxn[lane] = s;
    }
    lds_sync(true);
    if (lane < M) {
      T s = fr[O::rg + lane];
#pragma unroll
      for (int c = 0; c < NX; c++) s += fr[O::rV + lane + M * c] * sxn[c];
      a.alpha[size_t(k) * M + lane] = s;
      sal[lane] = s;
    }
    if (a.costates) {  // A_k^T (M_i[k+1] x_{k+1} + m_i[k+1]) (:176)
      const T* nrow = row_of(k + 1);
      const T* Ak = a.A + size_t(k) * NX * NX;
#pragma unroll 1
      for (int i = 0; i < NP; i++) {
        if (lane < NX) {
          T s = nrow[O::rm + i * NX + lane];
          for (int c = 0; c < NX; c++) s += nrow[O::rM + i * NX * NX + lane + NX * c] * sxn[c];
          sit[lane] = s;
        }
        lds_sync(true);
        if (lane < NX) {
          T s = T(0);
          for (int r = 0; r < NX; r++) s += Ak[r + NX * lane] * sit[r];
          a.costates[(size_t(k) * NP + i) * NX + lane] = s;
        }
        lds_sync(true);
      }
    }
    if (a.ed_out) {  // ILQSolver::ExpectedDecrease (ilq_solver.cpp:364-398) for this step
      lds_sync(true);
      T st = T(0), ct = T(0);
      if (lane < NP) {
        // alpha_i^T R_ii r_ii with R_ii r_ii from the backward pass (the reference forms (alpha^T R) r, :384-386)
#pragma unroll
        for (int c = 0; c < MU; c++) ct += sal[lane * MU + c] * fr[O::rRr + lane * MU + c];
        if (k > 0) {
#pragma unroll
          for (int c = 0; c < NX; c++) st += sx[c] * fr[O::rql + lane * NX + c];
        }
      }
#pragma unroll
      for (int i = 0; i < NP; i++) {
        ed -= shfl(ct, i);
        if (k > 0) ed -= shfl(st, i);
      }
    }
    lds_sync(true);
    if (lane < NX) sx[lane] = sxn[lane];
    dma_wait();
    lds_sync(true);
    cur = 1 - cur;
  }
  if (a.dx && lane < NX) a.dx[size_t(Tn - 1) * NX + lane] = sx[lane];  // :188-192
  if (lane < M) a.alpha[size_t(Tn - 1) * M + lane] = T(0);
  if (a.costates)
    for (int e = lane; e < NP * NX; e += 64) a.costates[size_t(Tn - 1) * NP * NX + e] = T(0);
  if (a.ed_out) {
    // step T-1: alpha = 0; state term delta_x^T Q l
    T st = T(0);
    if (lane < NP && Tn > 1) {
      const T* ql = row_of(Tn - 1) + O::rql;
#pragma unroll
      for (int c = 0; c < NX; c++) st += sx[c] * ql[lane * NX + c];
    }
#pragma unroll
    for (int i = 0; i < NP; i++) ed -= shfl(st, i);
    if (lane == 0) *a.ed_out = ed;
  }
}

}  // namespace ilqg
