// ilqg_lq_feedback2.hpp — the coupled feedback sweep for 16 < n <= 31 on the matrix cores (gfx950): one wavefront per
// player, value functions as 2 x 2 blocks of 16 x 16 accumulator-layout tiles.
//
// Computes what LQFeedbackSolver::Solve computes (src/lq_feedback_solver.cpp:71-244), like
// lq_feedback_instance_mfma_pw (n <= 16, ilqg_lq.hpp) and with the machinery of the open-loop sweep
// (ilqg_lq_openloop.hpp: padded tiles, ld_tile, DMA plan, transposition through the wave's own tile):
//   per backward step   S [P | alpha] = [Y | y_zeta]        (:148-180; S_ij = B_i^T Z_i B_j (+ R_ii), Y_i = B_i^T Z_i A)
//                       F = A - B P,  beta = -B alpha       (:189-194)
//                       zeta_i <- F^T (zeta_i + Z_i beta) + l_i + sum_j P_j^T (R_ij alpha_j - r_ij)   (:198-201, 206-212)
//                       Z_i    <- F^T Z_i F + Q_i + sum_j P_j^T R_ij P_j                              (:202-205)
// In homogeneous coordinates, n' = n + 1:  Za_i = [Z_i zeta_i; 0 0],  Fa = [F beta; 0 1],  Pa_j = [P_j | alpha_j],
//     Za_i <- Ca_i + Fa^T (Za_i Fa),   Ca_i = [Q_i l_i; 0 0] + sum_j P_j^T (R_ij Pa_j - [0 | r_ij]),
// so the vector recursion rides in column n of products that run anyway, and row n of G_i = Za_i^T [B; 0] is
// zeta_i^T B (y_zeta).  Wave i keeps Za_i and Za_i^T in registers; wave 0 solves the m x m system (one column per lane,
// Gershgorin step + elimination as in the n <= 16 sweep); two workgroup barriers per step.  The forward pass is
// lq_forward_pass_body (same scratch rows as the other feedback sweeps).
#pragma once

#include "ilqg_lq_openloop.hpp"

namespace ilqg {

template <typename T, int NX, int NP, int MU>
struct FB2Cfg {
  using C = LQCfg<T, NX, NP, MU>;
  static constexpr int M = NP * MU;
  static constexpr int NT = 64 * NP;
  static constexpr int NH = NX + 1;
  static constexpr int NTL = (NH + 15) / 16;
  static constexpr bool OK = NX > 16 && NH <= 32 && M <= 16 && M + NX + 1 <= 64;
  static constexpr int LD = pad_ld<T>(NH);
  static constexpr int MAT = (NH * LD + 3) & ~3;
  static constexpr int LDZ = pad_ld<T>(M);
  static constexpr int BIMG = ((NX * M + 3) & ~3) + C::RMAX + C::rMAX;  // [B | R | r] of one step (double-buffered)
  // LDS (elements): small regions first, padded matrices last + slack (edge-tile reads, see OLCfg)
  static constexpr int oSY = 0;                            // [S | Y | y_zeta]: m x (m + n + 1), column-major
  static constexpr int oAl = oSY + M * (M + NX + 1);       // alpha (m)
  static constexpr int oPa = (oAl + M + 3) & ~3;           // [P | alpha]: m x n', leading dimension LDZ
  static constexpr int oB = (oPa + LDZ * NH + 3) & ~3;     // two [B | R | r] images
  static constexpr int oA = oB + 2 * BIMG;                 // two Aa images
  static constexpr int oZ = oA + 2 * MAT;                  // per player: Qa_i image / transposition tile
  static constexpr int SLACK = (32 + 32 * LD - MAT + 3) & ~3;
  static constexpr int LDS_SWEEP = oZ + NP * MAT + (SLACK > 0 ? SLACK : 0);
  static_assert(!OK || (oPa + 16 + 32 * LDZ <= LDS_SWEEP && oB + 32 + NX * 32 <= LDS_SWEEP), "edge-tile reads stay inside the LDS");
  // forward pass (lq_forward_pass_body on wave 0): G staged steps of [A | scratch row], twice, plus x
  static constexpr int FSLOT = (NX * NX + C::SCR + 3) & ~3;
  static constexpr int LDS_FWD = 2 * 2 * FSLOT + NX + 8;
  static constexpr int LDS_ELEMS = LDS_SWEEP > LDS_FWD ? LDS_SWEEP : LDS_FWD;
};

// LDS elements of the matrix-core feedback sweep of a shape (either tile count), and the slot behind it where a sweep that
// runs its own forward pass leaves the expected decrease.
template <typename T, int NX, int NP, int MU>
struct MfmaSweepLds {
  static constexpr int ELEMS = LQCfg<T, NX, NP, MU>::MFMA_ONE_TILE ? PWCfg<T, NX, NP, MU>::LDS_ELEMS : FB2Cfg<T, NX, NP, MU>::LDS_ELEMS;
};

template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void lq_feedback_instance_mfma_pw2(const LQArgs<T>& a, const PairTable& pt, T* sm) {
  using C = LQCfg<T, NX, NP, MU>;
  using W = FB2Cfg<T, NX, NP, MU>;
  using TL = Tile<T>;
  using vec = typename TL::vec;
  constexpr int M = W::M, NT = W::NT, NH = W::NH, NTL = W::NTL, LD = W::LD, LDZ = W::LDZ, MAT = W::MAT;
  constexpr int S = int(sizeof(T)), SCR = C::SCR;
  constexpr int BOFF_R = (NX * M + 3) & ~3, BOFF_r = BOFF_R + C::RMAX;
  static_assert(NTL == 2, "the n <= 16 sweep is lq_feedback_instance_mfma_pw");
  const int t = threadIdx.x;
  const int w = t >> 6;  // wave = player
  const int lane = t & 63, g = lane >> 4, j = lane & 15;
  const int Tn = a.T_steps;
  const PairRegs<NP> pr(pt);
  const bool want_fwd = a.dx != nullptr || a.ed_out != nullptr || a.defer_forward != 0;
  const vec zero4 = {T(0), T(0), T(0), T(0)};
  int ro_ww = 0, rg_ww = 0;
#pragma unroll
  for (int e = 0; e < NP; e++) {
    ro_ww = (w == e) ? pr.ro[e][e] : ro_ww;
    rg_ww = (w == e) ? pr.rg[e][e] : rg_ww;
  }
  T* const sZ = sm + W::oZ + w * MAT;
  T* const sSY = sm + W::oSY;
  T* const sYz = sSY + M * (M + NX);
  T* const sAl = sm + W::oAl;
  T* const sPa = sm + W::oPa;
  auto bimg = [&](int which) { return sm + W::oB + which * W::BIMG; };
  auto aimg = [&](int which) { return sm + W::oA + which * MAT; };

  // ---- DMA plumbing (as in the open-loop sweep) ----
  constexpr int PS = ((NX * S) % 16 == 0 && (LD * S) % 16 == 0) ? 16 : 4;
  constexpr int PPC = LD * S / PS, VPC = NX * S / PS;
  constexpr int WI = (NX * PPC + 63) / 64;
  int plan[WI];
#pragma unroll
  for (int h = 0; h < WI; h++) {
    const int p = h * 64 + lane;
    const int c = p / PPC, inb = p % PPC;
    plan[h] = (p < NX * PPC && inb < VPC) ? c * NX * S + inb * PS : -1;
  }
  auto dma_piece = [&](const char* s, char* d) {
    if constexpr (PS == 16)
      __builtin_amdgcn_global_load_lds((glb_vptr)s, (lds_vptr)d, 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((glb_vptr)s, (lds_vptr)d, 4, 0, 0);
  };
  auto dma_matrix = [&](const T* src_, T* dst, int first, int step) {
    const char* src = reinterpret_cast<const char*>(uniform_ptr(src_));
#pragma unroll
    for (int h = 0; h < WI; h++)
      if (h % step == first && plan[h] >= 0) dma_piece(src + unsigned(plan[h]), reinterpret_cast<char*>(dst) + h * 64 * PS);
  };
  auto issue_Q = [&](int k) {  // Q_i | l_i of step k into this wave's tile
    dma_matrix(a.Q + (size_t(k) * NP + w) * NX * NX, sZ, 0, 1);
    static_assert(VPC <= 64, "a column is one DMA instruction");
    const char* src = reinterpret_cast<const char*>(uniform_ptr(a.l + (size_t(k) * NP + w) * NX));
    if (lane < VPC) dma_piece(src + unsigned(lane * PS), reinterpret_cast<char*>(sZ + LD * NX));
  };
  auto issue_shared = [&](int k) {  // A and [B | R | r] of step k into the images of its parity (the waves share)
    dma_matrix(a.A + size_t(k) * NX * NX, aimg(k & 1), w, NP);
    T* img = bimg(k & 1);
    dma_g2l<NT, false>(a.Bm + size_t(k) * NX * M, img, NX * M * S, t);
    dma_g2l<NT, false>(a.R + size_t(k) * pt.Rsz, img + BOFF_R, pt.Rsz * S, t);
    dma_g2l<NT, false>(a.r + size_t(k) * pt.rsz, img + BOFF_r, pt.rsz * S, t);
  };
  auto lds_drain = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  const int oD = tile_lane_offset<T, false>(LD, g, j), oT = tile_lane_offset<T, true>(LD, g, j);
  const int oDn = tile_lane_offset<T, false>(NX, g, j), oTn = tile_lane_offset<T, true>(NX, g, j);
  const int oDz = tile_lane_offset<T, false>(LDZ, g, j);
  constexpr int RS = TL::row(0, 1) - TL::row(0, 0);
  const int row0 = TL::row(g, 0);
  struct Blk {
    vec v[NTL][NTL];
  };
  auto load_blk = [&](const T* mat, bool transposed) {
    Blk o;
#pragma unroll
    for (int aa = 0; aa < NTL; aa++)
#pragma unroll
      for (int bb = 0; bb < NTL; bb++)
        o.v[aa][bb] = transposed ? ld_tile<T, true>(mat + oT, LD, aa, bb, NH, NH, g, j) : ld_tile<T, false>(mat + oD, LD, aa, bb, NH, NH, g, j);
    return o;
  };
  constexpr int KH0 = kd_mask<T>(NH, 0), KH1 = kd_mask<T>(NH, 1);
  constexpr int KN0 = kd_mask<T>(NX, 0), KN1 = kd_mask<T>(NX, 1);
  constexpr int KM0 = kd_mask<T>(M, 0);

  // Q_i l_i of a step (ExpectedDecrease) -> scratch row; reads this wave's tile while it holds Qa_i
  auto stash_ql = [&](int k) {
    if (want_fwd && lane < NX) {
      T s = T(0);
#pragma unroll
      for (int c = 0; c < NX; c++) s += sZ[lane + LD * c] * sZ[c + LD * NX];
      a.scratch[size_t(k) * SCR + w * NX + lane] = s;
    }
  };

  // ---- once per sweep: zero the padding, the homogeneous column of both Aa images ----
  static_assert(W::oZ == W::oA + 2 * MAT, "the Aa images and the tiles are zeroed in one piece");
  for (int e = t; e < (NP + 2) * MAT; e += NT) sm[W::oA + e] = T(0);
  lds_sync(false);
  if (t < 2) aimg(t)[NX + LD * NX] = T(1);
  lds_sync(false);

  // ---- terminal step: Z_w = Q_w[T-1], zeta_w = l_w[T-1]  (:102-105) ----
  issue_Q(Tn - 1);
  if (Tn >= 2) issue_shared(Tn - 2);
  dma_wait();
  lds_sync(false);
  Blk Zd = load_blk(sZ, false);  // D(Za_w)
  Blk ZT = load_blk(sZ, true);   // D(Za_w^T)
  stash_ql(Tn - 1);
  for (int e = t; e < M * NX; e += NT) a.P[size_t(Tn - 1) * M * NX + e] = T(0);
  if (t < M) a.alpha[size_t(Tn - 1) * M + t] = T(0);
  if (want_fwd) {
    if (t < NP) a.scratch[size_t(Tn - 1) * SCR + NP * NX + t] = T(0);
    if (t < NX) a.scratch[size_t(Tn - 1) * SCR + NP * (NX + 1) + t] = T(0);
  }
  lds_sync(true);
  lds_drain();  // the tile has been read: the DMA engine may refill it
  if (Tn >= 2) issue_Q(Tn - 2);

#pragma unroll 1
  for (int k = Tn - 2; k >= 0; k--) {
    const T* sB = bimg(k & 1);
    const T* sR = sB + BOFF_R;
    const T* sr = sB + BOFF_r;
    const T* sAa = aimg(k & 1);
    // ---- this player's MU rows of [S | Y | y_zeta]: G = Za_w^T [B; 0], then G_w^T [B | A] ----
    {
      vec Btd[NTL];  // D([B; 0]): n' x m
#pragma unroll
      for (int c = 0; c < NTL; c++) Btd[c] = ld_tile<T, false>(sB + oDn, NX, c, 0, NX, M, g, j);
      vec G[NTL];  // Za_w^T [B; 0]: n' x m (row n: zeta_w^T B)
#pragma unroll
      for (int aa = 0; aa < NTL; aa++) {
        vec acc = tile_xty_blocks<T, KN0>(Zd.v[0][aa], Btd[0], zero4);
        acc = tile_xty_blocks<T, KN1>(Zd.v[1][aa], Btd[1], acc);
        G[aa] = acc;
      }
      // y_zeta = B_w^T zeta_w + r_ww (:154-157): row n of G, this player's columns
      {
        constexpr int rr = NX - 16;  // row n sits in tile row 1
        constexpr int gJ = sizeof(T) == 8 ? rr % 4 : rr / 4, rJ = sizeof(T) == 8 ? rr / 4 : rr % 4;
        static_assert(TL::row(gJ, rJ) == rr, "accumulator-layout position of row n");
        if (g == gJ && j / MU == w) sYz[j] = G[1][rJ] + sr[rg_ww + (j - w * MU)];
      }
      // (G^T [B | A]) rows w*MU .. : G as the left operand; columns of B (one tile) and of A (two tiles)
#pragma unroll
      for (int bb = 0; bb < NTL + 1; bb++) {
        vec acc = zero4;
#pragma unroll
        for (int c = 0; c < NTL; c++) {
          const vec rhs = bb == 0 ? Btd[c] : ld_tile<T, false>(sAa + oD, LD, c, bb - 1, NX, NX, g, j);
          acc = c == 0 ? tile_xty_blocks<T, KN0>(G[0], rhs, acc) : tile_xty_blocks<T, KN1>(G[1], rhs, acc);
        }
        const int col = bb == 0 ? j : M + 16 * (bb - 1) + j;         // column of [S | Y]
        const bool cok = bb == 0 ? j < M : 16 * (bb - 1) + j < NX;
        const bool diag = bb == 0 && j / MU == w;                     // + R_ww on this player's diagonal block (:148-150)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int q = row0 + RS * r;  // row of the stacked system
          if (cok && q / MU == w && q < M)
            sSY[q + M * col] = acc[r] + (diag ? sR[ro_ww + (q - w * MU) + MU * (diag ? j - w * MU : 0)] : T(0));
        }
      }
    }
    lds_sync(NT <= 64);  // barrier 1: [S | Y | y_zeta] complete
    if (k > 0) issue_shared(k - 1);  // every wave has left step k + 1: the other images are free
    // ---- wave 0: Gershgorin (:163-176), the m x m solve (:180) ----
    if (w == 0) {
      T col[M], x[M];
      const bool isS = lane < M;
      const T* src = (lane < M + NX) ? sSY + M * lane : sYz;
#pragma unroll
      for (int r = 0; r < M; r++) {
        col[r] = src[r];
        x[r] = T(0);
      }
      {
        T l1 = T(0), diag = T(0);
#pragma unroll
        for (int r = 0; r < M; r++) {
          l1 += (col[r] < T(0) ? -col[r] : col[r]);
          diag = (r == lane) ? col[r] : diag;
        }
        const T radius = l1 - (diag < T(0) ? -diag : diag);
        const T eval_lo = diag - radius;
        const T bump = (isS && a.adaptive && eval_lo < T(1e-3f)) ? radius + T(1e-3f) : T(0);
#pragma unroll
        for (int r = 0; r < M; r++) col[r] = col[r] + ((r == lane) ? bump : T(0));
      }
      if (a.adaptive)
        lu_solve_columns<T, M>(col, lane, x);
      else
        qr_solve_columns<T, M>(col, lane, x);
      if (lane >= M && lane <= M + NX) {
#pragma unroll
        for (int r = 0; r < M; r++) sPa[r + LDZ * (lane - M)] = x[r];  // [P | alpha], column n = alpha
        if (lane == M + NX) {
#pragma unroll
          for (int r = 0; r < M; r++) sAl[r] = x[r];
        }
      }
    }
    dma_wait();  // this wave's share of the next step's images; Q_w | l_w of this step in its tile
    lds_sync(NT <= 64);  // barrier 2: (P, alpha) and the next step's images published
    if (w == NP - 1) {  // the strategies go to global memory from the last wave
      if (lane < NX) {
#pragma unroll
        for (int r = 0; r < M; r++) uniform_ptr(a.P + size_t(k) * M * NX)[unsigned(r + M * lane)] = sPa[r + LDZ * lane];
      } else if (lane < NX + M) {
        uniform_ptr(a.alpha + size_t(k) * M)[unsigned(lane - NX)] = sAl[lane - NX];
      }
    }
    // ---- Fa = Aa - [B; 0] [P | alpha]  (every wave) ----
    vec Pad[NTL];  // D([P | alpha]): m x n'
    Blk Fd;
    {
      vec nBT[NTL];
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        nBT[bb] = ld_tile<T, true>(sB + oTn, NX, 0, bb, M, NX, g, j);
#pragma unroll
        for (int r = 0; r < 4; r++) nBT[bb][r] = -nBT[bb][r];
        Pad[bb] = ld_tile<T, false>(sPa + oDz, LDZ, 0, bb, M, NH, g, j);
      }
#pragma unroll
      for (int aa = 0; aa < NTL; aa++)
#pragma unroll
        for (int bb = 0; bb < NTL; bb++)
          Fd.v[aa][bb] = tile_xty_blocks<T, KM0>(nBT[aa], Pad[bb], ld_tile<T, false>(sAa + oD, LD, aa, bb, NH, NH, g, j));
    }
    if (want_fwd && w == 0) {
      // beta_k = -B alpha_k: column n of Fa (rows < n) -> scratch row; alpha_i^T R_ii r_ii (ilq_solver.cpp:384-386)
#pragma unroll
      for (int aa = 0; aa < NTL; aa++) {
        if (16 + j == NX) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int rw = 16 * aa + row0 + RS * r;
            if (rw < NX) a.scratch[size_t(k) * SCR + NP * (NX + 1) + rw] = Fd.v[aa][1][r];
          }
        }
      }
      if (lane < NP) {
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
        a.scratch[size_t(k) * SCR + NP * NX + lane] = acc;
      }
    }
    // ---- W = Za_w Fa ----
    Blk Wd;
#pragma unroll
    for (int aa = 0; aa < NTL; aa++)
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        vec acc = tile_xty_blocks<T, KH0>(ZT.v[0][aa], Fd.v[0][bb], zero4);
        acc = tile_xty_blocks<T, KH1>(ZT.v[1][aa], Fd.v[1][bb], acc);
        Wd.v[aa][bb] = acc;
      }
    stash_ql(k);
    // ---- Ca_w = [Q_w l_w; 0 0] + sum_jj P_jj^T (R_w,jj [P_jj | alpha_jj] - [0 | r_w,jj]);  Za_w' = Ca_w + Fa^T W ----
    Blk Cd = load_blk(sZ, false);
    static_for<NP>([&](auto JJ) {
      constexpr int jj = decltype(JJ)::value;
      int qw = -1, ro_wj = 0, rg_wj = 0;
#pragma unroll
      for (int e = 0; e < NP; e++) {
        qw = (w == e) ? pr.q[e][jj] : qw;
        ro_wj = (w == e) ? pr.ro[e][jj] : ro_wj;
        rg_wj = (w == e) ? pr.rg[e][jj] : rg_wj;
      }
      if (qw < 0) return;  // wave-uniform
      const T* Rij = sR + ro_wj;
      const T* rij = sr + rg_wj;
      vec Pj[NTL], Hd[NTL];  // rows jj*MU .. of [P | 0] and of R [P | alpha] - [0 | r]
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        const int col = 16 * bb + j;
        const int colc = col < NH ? col : 0;
        T pb[MU];
#pragma unroll
        for (int b = 0; b < MU; b++) pb[b] = sPa[(jj * MU + b) + LDZ * colc];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int aa = row0 + RS * r - jj * MU;
          const bool in = aa >= 0 && aa < MU && col < NH;
          const int ac = (aa >= 0 && aa < MU) ? aa : 0;
          T h = (col == NX) ? -rij[ac] : T(0);
#pragma unroll
          for (int b = 0; b < MU; b++) h += Rij[ac + MU * b] * pb[b];
          Hd[bb][r] = in ? h : T(0);
          Pj[bb][r] = (in && col < NX) ? Pad[bb][r] : T(0);
        }
      }
#pragma unroll
      for (int aa = 0; aa < NTL; aa++)
#pragma unroll
        for (int bb = 0; bb < NTL; bb++)
          Cd.v[aa][bb] = tile_xty_blocks<T, kblock_mask<T>(jj * MU, jj * MU + MU)>(Pj[aa], Hd[bb], Cd.v[aa][bb]);
    });
#pragma unroll
    for (int aa = 0; aa < NTL; aa++)
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        vec acc = tile_xty_blocks<T, KN0>(Fd.v[0][aa], Wd.v[0][bb], Cd.v[aa][bb]);
        acc = tile_xty_blocks<T, KN1>(Fd.v[1][aa], Wd.v[1][bb], acc);
        Zd.v[aa][bb] = acc;
      }
    // transpose through this wave's tile: write D(Za'), read D(Za'^T)
    lds_sync(true);  // every read of Qa_w is done
#pragma unroll
    for (int aa = 0; aa < NTL; aa++)
#pragma unroll
      for (int bb = 0; bb < NTL; bb++) {
        const int col = 16 * bb + j;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int rw = 16 * aa + TL::row(g, r);
          if (rw < NH && col < NH) sZ[oD + 16 * aa + RS * r + LD * (16 * bb)] = Zd.v[aa][bb][r];
        }
      }
    lds_sync(true);
    ZT = load_blk(sZ, true);
    lds_sync(true);
    lds_drain();  // the tile has been read: the DMA engine may refill it
    if (k > 0) issue_Q(k - 1);
  }

  if (want_fwd && !a.defer_forward) {
    __syncthreads();  // scratch rows written by all waves
    if (w == 0) lq_forward_pass_body<T, NX, NP, MU, 64, W::LDS_ELEMS>(a, sm, lane);
  }
}

}  // namespace ilqg
