// ilqg_lq.hpp — coupled backward LQ Nash sweep for ONE game instance per workgroup (gfx950).
//
// Computes what LQFeedbackSolver::Solve computes (src/lq_feedback_solver.cpp:71-244):
//   k = T-2 .. 0:  S X = Y  (stacked Nash system, Gershgorin-regularised, Householder QR)
//                  F = A - B P, beta = -B alpha,
//                  Z_i <- F^T Z_i F + Q_i + sum_j P_j^T R_ij P_j
//                  zeta_i <- F^T (zeta_i + Z_i beta) + l_i + sum_j P_j^T (R_ij alpha_j - r_ij)
//   forward pass:  dx_{k+1} = A dx_k - B alpha_k   (no feedback term, :237-239)
// and, fused into the same pass over the data, ILQSolver::ExpectedDecrease
// (src/ilq_solver.cpp:364-398).
//
// Two formulations (this is not the reference's loop nest):
//   * n <= 16: lq_feedback_instance_mfma_pw — one wavefront per player, Z_i in MFMA accumulator-layout
//     registers, every product a chain of v_mfma_*_16x16x4, operands DMA'd into zero-padded
//     conflict-free LDS tiles (see the comment on that function and DESIGN.md §3.2);
//   * larger n: lq_feedback_instance — lane t = i*NX + c owns COLUMN c of Z_i in registers;
//     F^T Z_i F is two passes of "uniform matrix x private column" with broadcast LDS operands.
// In both, the (m x m) Nash system with its n+1 right-hand sides lives one COLUMN PER LANE in wave 0
// (v_readlane broadcasts, no LDS and no barriers inside the factorisation): Householder QR as the
// reference, or elimination without pivoting once the Gershgorin step has made S diagonally dominant.
// The per-step [B|A|Q|l|R|r] block is staged global -> LDS by DMA one step ahead (double-buffered).
#pragma once

#include "ilqg_common.hpp"
#include "ilqg_mfma.hpp"

namespace ilqg {

template <typename T>
struct LQArgs {
  const T *A, *Bm, *Q, *l, *R, *r;  // instance bases: [T][n*n], [T][n*m], [T][N][n*n], [T][N][n], [T][Rsz], [T][rsz]
  const T* x0;                      // [n] or nullptr (zero)
  T *P, *alpha, *dx;                // [T][m*n], [T][m], [T][n] (dx may be nullptr)
  T* scratch;                       // [T][N*(n+1) + n] when dx/ed requested, else nullptr
  T* ed_out;                        // expected decrease (one scalar) or nullptr
  T* costates = nullptr;            // [T][N][n] or nullptr (lq_solver.h:63-69)
  int T_steps;
  int adaptive;
  int symmetric = 0;                // 1: every Q_i and R_ij is exactly symmetric (what the quadraticisation stage writes)
  int defer_forward = 0;            // 1: leave the scratch rows for a forward pass that runs elsewhere (the solve's
                                    //    trial kernel runs it beside the next rollout): no dx, no expected decrease here
  long long* ph = nullptr;          // optional: 8 shader-clock accumulators per instance (phase profile)
  const T* compact = nullptr;       // [T][compact_tab[RC_W]] compact rows of [Q | l | R | r] (ilqg_common.hpp) instead of the
  const int* compact_tab = nullptr; // dense arrays: the one-tile player-parallel sweep only; compact_tab = the row program's block
  double dt = 0.0;                  // the time step (a constant of the compact rows' background)
  int prio_div = 0;                 // > 0: rotate the wave priority every step, phase = blockIdx.x / prio_div (see the sweep)
  long long* tl = nullptr;          // optional: timeline stamps (ilqg_common.hpp, -DILQG_TIMELINE=1)
  int tl_b = 0;
  // > 0: A is block diagonal over `nsub` subsystems with state offsets xoff[0 .. nsub] and B_i is confined to the rows of
  // subsystem i (what ConcatenatedDynamicalSystem::Linearize produces, src/concatenated_dynamical_system.cpp:86-107):
  // the open-loop sweep then skips the matrix-instruction blocks that only meet structural zeros.  0: dense.
  int nsub = 0;
  int xoff[ILQG_MAX_PLAYERS + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
};

template <typename T, int NX, int NP, int MU>
struct LQCfg {
  static constexpr int M = NP * MU;
  static constexpr int L = NP * NX;
  static constexpr int NSOLVE = M + NX + 1;  // columns of [S | Y]
  static constexpr int NT = ((L > NSOLVE ? L : NSOLVE) + 63) / 64 * 64;
  static constexpr int RMAX = NP * NP * MU * MU;
  static constexpr int rMAX = NP * NP * MU;
  // staged image [B | A | Q | l | R | r] — same element order as the global arrays.  Two images:
  // the step being processed and the one the LDS-DMA engine is filling for the next step.
  static constexpr int oB = 0;
  static constexpr int oA = oB + NX * M;
  static constexpr int oQ = oA + NX * NX;
  static constexpr int ol = oQ + NP * NX * NX;
  static constexpr int oR = ol + NP * NX;
  static constexpr int or_ = oR + RMAX;
  static constexpr int IMG = (or_ + rMAX + 3) & ~3;  // keeps the second image 16-byte aligned
  // intermediates
  static constexpr int NXS = NX | 1;  // odd leading dimension: conflict-free column writes
  static constexpr int oF = 2 * IMG;
  static constexpr int oUt = oF + NX * NXS;
  static constexpr int oBZ = oUt + NP * NX * NXS;
  static constexpr int oP = oBZ + M * NX;
  static constexpr int oAl = oP + M * NX;
  static constexpr int oBeta = oAl + M;
  static constexpr int oZeta = oBeta + NX;
  static constexpr int oYz = oZeta + NP * NX;
  static constexpr int oX = oYz + M;
  // MFMA variant (NX <= 16): transpose scratch (16 x 17) and a 16-vector
  // a player-parallel sweep on the matrix cores exists: one 16 x 16 tile per value function (n <= 16,
  // lq_feedback_instance_mfma_pw below) or a 2 x 2 block of them (n <= 31, ilqg_lq_feedback2.hpp)
  static constexpr bool USE_MFMA = (NX <= 16 && NP * NX <= 64) || (NX > 16 && NX <= 31 && M <= 16 && M + NX + 1 <= 64);
  static constexpr bool MFMA_ONE_TILE = NX <= 16;
  static constexpr int oTr = oX + NX;
  static constexpr int oTv = oTr + 16 * 17;
  static constexpr int oSY = oTv + 16;  // [S | Y] bounce buffer of the MFMA variant: M x 32, column-major
  static constexpr int LDS_ELEMS = oSY + M * 32;
  static constexpr int SCR = NP * (NX + 1) + NX;  // scratch row: [Q_i l_i (N*n) | alpha_i^T R_ii r_ii (N) | beta (n)]
  static_assert(RMAX <= 64, "R blocks are copied by one DMA instruction");
  static_assert(NSOLVE <= 64, "the stacked Nash system must fit one wavefront");
  static_assert(SCR <= NP * NX * NX + NP * NX, "the forward pass parks the scratch row in the [Q|l] slots");
};

// The (i,j) block table pulled into registers once per sweep.  Indexing the kernel-argument copy
// with a run-time q costs a dependent scalar load (~100+ cycles) per access — dozens per step.
template <int NP>
struct PairRegs {
  int q[NP][NP];   // block index of (i,j) or -1
  int ro[NP][NP];  // offset inside the R row
  int rg[NP][NP];  // offset inside the r row
  __device__ __forceinline__ explicit PairRegs(const PairTable& pt) {
#pragma unroll
    for (int i = 0; i < NP; i++)
#pragma unroll
      for (int j = 0; j < NP; j++) {
        q[i][j] = -1;
        ro[i][j] = 0;
        rg[i][j] = 0;
      }
    for (int e = 0; e < pt.npairs; e++) {
      const int pi = pt.pi[e], pj = pt.pj[e], r0 = pt.roff[e], g0 = pt.rgoff[e];
#pragma unroll
      for (int i = 0; i < NP; i++)
#pragma unroll
        for (int j = 0; j < NP; j++)
          if (pi == i && pj == j) {
            q[i][j] = e;
            ro[i][j] = r0;
            rg[i][j] = g0;
          }
    }
  }
};

// Starts the DMA of step k's [B|A|Q|l|R|r] block into the image at `img`.
template <typename T, int NX, int NP, int MU, int NT = LQCfg<T, NX, NP, MU>::NT>
__device__ __forceinline__ void lq_stage_issue(const LQArgs<T>& a, const PairTable& pt, int k, T* img, int t) {
  using C = LQCfg<T, NX, NP, MU>;
  constexpr int M = C::M, S = int(sizeof(T));
  constexpr bool wB = (NX * M * S) % 16 == 0 && (C::oB * S) % 16 == 0;
  constexpr bool wA = (NX * NX * S) % 16 == 0 && (C::oA * S) % 16 == 0;
  constexpr bool wQ = (NP * NX * NX * S) % 16 == 0 && (C::oQ * S) % 16 == 0;
  constexpr bool wl = (NP * NX * S) % 16 == 0 && (C::ol * S) % 16 == 0;
  dma_g2l<NT, wB>(a.Bm + size_t(k) * NX * M, img + C::oB, NX * M * S, t);
  dma_g2l<NT, wA>(a.A + size_t(k) * NX * NX, img + C::oA, NX * NX * S, t);
  dma_g2l<NT, wQ>(a.Q + size_t(k) * NP * NX * NX, img + C::oQ, NP * NX * NX * S, t);
  dma_g2l<NT, wl>(a.l + size_t(k) * NP * NX, img + C::ol, NP * NX * S, t);
  dma_g2l<NT, false>(a.R + size_t(k) * pt.Rsz, img + C::oR, pt.Rsz * S, t);
  dma_g2l<NT, false>(a.r + size_t(k) * pt.rsz, img + C::or_, pt.rsz * S, t);
}

// Broadcast from a compile-time-known lane through v_readlane (scalar path): no LDS crossbar
// round trip, unlike __shfl -> ds_bpermute.  `l` must be wave-uniform.
__device__ __forceinline__ float bcast(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double bcast(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lq_abs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double lq_abs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float lq_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double lq_sqrt(double x) { return sqrt(x); }

// Householder QR solve of S X = Y with one column of [S | Y] per lane (wave 0).
// Restates Eigen's householder_qr_inplace_unblocked + makeHouseholder +
// applyHouseholderOnTheLeft + triangular solve — the arithmetic behind
// `S_.householderQr().solve(Y_)` (src/lq_feedback_solver.cpp:180).
// On return lanes >= M hold the solution column in x[].
template <typename T, int M>
__device__ __forceinline__ void qr_solve_columns(T (&col)[M], int lane, T (&x)[M]) {
#pragma unroll
  for (int k = 0; k < M; k++) {
    T tailsq = T(0);
#pragma unroll
    for (int i = k + 1; i < M; i++) tailsq += col[i] * col[i];
    const T c0 = col[k];
    T beta, tau;
    T ess[M];
    if (M - k == 1 || tailsq <= (sizeof(T) == 4 ? T(1.17549435e-38f) : T(2.2250738585072014e-308))) {
      tau = T(0);
      beta = c0;
#pragma unroll
      for (int i = k + 1; i < M; i++) ess[i] = T(0);
    } else {
      beta = lq_sqrt(c0 * c0 + tailsq);
      if (c0 >= T(0)) beta = -beta;
      const T inv = T(1) / (c0 - beta);  // one division; Eigen divides each entry (last-ulp difference)
#pragma unroll
      for (int i = k + 1; i < M; i++) ess[i] = col[i] * inv;
      tau = (beta - c0) / beta;
    }
    const T tau_k = bcast(tau, k);
    T v[M];
#pragma unroll
    for (int i = k + 1; i < M; i++) v[i] = bcast(ess[i], k);
    if (lane == k) {
      col[k] = beta;
#pragma unroll
      for (int i = k + 1; i < M; i++) col[i] = ess[i];
    } else if (lane > k) {
      if (M - k == 1) {
        col[k] *= (T(1) - tau_k);
      } else if (tau_k != T(0)) {
        T tmp = T(0);
#pragma unroll
        for (int i = k + 1; i < M; i++) tmp += v[i] * col[i];
        tmp += col[k];
        col[k] -= tau_k * tmp;
#pragma unroll
        for (int i = k + 1; i < M; i++) col[i] -= tau_k * v[i] * tmp;
      }
    }
  }
  // reciprocal of R's diagonal, one division per lane off the back-substitution chain
  T diag = T(1);
#pragma unroll
  for (int i = 0; i < M; i++) diag = (lane == i) ? col[i] : diag;
  const T dinv = T(1) / diag;
#pragma unroll
  for (int i = M - 1; i >= 0; i--) {
    T s = col[i];
#pragma unroll
    for (int k2 = i + 1; k2 < M; k2++) s -= bcast(col[i], k2) * x[k2];
    x[i] = s * bcast(dinv, i);
  }
}

// Gaussian elimination solve of S X = Y in the same column-per-lane layout, no pivoting.
// Used when the Gershgorin step (:163-176) has run: it leaves every column of S strictly diagonally
// dominant, for which elimination without pivoting is backward stable (growth factor <= 2).  The
// reference factors the same S with Householder QR (:180); both return the solution of the same
// well-conditioned system, so they agree to a few ulps times cond(S) — far inside the 1e-6 bar —
// while the elimination has no square roots and no dependent dot-product chains: each step is one
// reciprocal, M-k-1 independent broadcasts and M-k-1 independent FMAs per lane.
// 1 / x from the hardware reciprocal refined by Newton steps (to within an ulp of the correctly rounded
// quotient): a fraction of the latency of the IEEE division sequence, and the pivots' reciprocals
// sit on the solve's critical path.
__device__ __forceinline__ double fast_recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fast_recip(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
  return r;
}

template <typename T, int M>
__device__ __forceinline__ void lu_solve_columns(T (&col)[M], int lane, T (&x)[M]) {
  T dinv[M];  // 1 / U[k][k], wave-uniform: lane k forms the pivot's reciprocal for the multipliers anyway
#pragma unroll
  for (int k = 0; k + 1 < M; k++) {
    const T rinv = fast_recip(col[k]);  // the pivot's reciprocal where it matters: on lane k
    dinv[k] = bcast(rinv, k);
    T f[M];
#pragma unroll
    for (int i = k + 1; i < M; i++) f[i] = bcast(col[i] * rinv, k);  // multipliers S[i][k] / S[k][k]
#pragma unroll
    for (int i = k + 1; i < M; i++) col[i] -= f[i] * col[k];        // row_i -= f_i row_k, every column at once
  }
  dinv[M - 1] = bcast(fast_recip(col[M - 1]), M - 1);
#pragma unroll
  for (int i = M - 1; i >= 0; i--) {
    T s = col[i];
#pragma unroll
    for (int k2 = i + 1; k2 < M; k2++) s -= bcast(col[i], k2) * x[k2];
    x[i] = s * dinv[i];
  }
}

// Gaussian elimination with PARTIAL PIVOTING in the same column-per-lane layout, for systems that carry no
// diagonal-dominance guarantee (the m x m system of the open-loop sweep).  Lane k holds column k of S, i.e. the whole
// pivot column of step k: it picks the row, the choice is broadcast (wave-uniform), and every lane swaps the two rows of
// its own column with selects.  Backward stable in practice like the Householder QR the reference uses for these
// systems; per step one reciprocal instead of a square root and two divisions.
template <typename T, int M>
__device__ __forceinline__ void lu_pp_solve_columns(T (&col)[M], int lane, T (&x)[M]) {
#pragma unroll
  for (int k = 0; k + 1 < M; k++) {
    int p = k;
    T best = col[k] < T(0) ? -col[k] : col[k];
#pragma unroll
    for (int i = k + 1; i < M; i++) {
      const T v = col[i] < T(0) ? -col[i] : col[i];
      const bool gt = v > best;
      best = gt ? v : best;
      p = gt ? i : p;
    }
    p = __builtin_amdgcn_readlane(p, k);
    const T ck = col[k];
    T cp = ck;
#pragma unroll
    for (int i = k + 1; i < M; i++) {
      const bool sel = (i == p);
      cp = sel ? col[i] : cp;
      col[i] = sel ? ck : col[i];
    }
    col[k] = cp;
    const T rinv = fast_recip(col[k]);
    T f[M];
#pragma unroll
    for (int i = k + 1; i < M; i++) f[i] = bcast(col[i] * rinv, k);
#pragma unroll
    for (int i = k + 1; i < M; i++) col[i] -= f[i] * col[k];
  }
  T diag = T(1);
#pragma unroll
  for (int i = 0; i < M; i++) diag = (lane == i) ? col[i] : diag;
  const T dinv = fast_recip(diag);
#pragma unroll
  for (int i = M - 1; i >= 0; i--) {
    T s = col[i];
#pragma unroll
    for (int k2 = i + 1; k2 < M; k2++) s -= bcast(col[i], k2) * x[k2];
    x[i] = s * bcast(dinv, i);
  }
}

// Forward pass of the sweep: delta_xs (src/lq_feedback_solver.cpp:217-241 — no feedback term) and
// ILQSolver::ExpectedDecrease (src/ilq_solver.cpp:364-398) from the per-step scratch rows.
// A_{k+1} and scratch row k+1 are DMA'd into the idle image while step k is computed.
// NT = number of threads that execute the pass (the caller has already made the scratch rows
// visible: __syncthreads after the sweep).  LDSE = LDS elements available at `sm`.
// One step is a 14-FMA chain (~300 cycles), far shorter than the global-load latency of its operands, so
// A_k and scratch row k are staged G steps at a time into two LDS groups: the DMA of group g+1 runs
// while group g is computed and only one wait per group is exposed.
// Single-wave form of the forward pass (the one the MFMA sweeps and the solve's trial kernel run).  What the step costs
// is one LDS exchange of delta_x and one NX-term FMA chain, so everything else is taken off that chain:
//   * the [A_k | scratch row k] images of the next G steps travel global -> registers -> LDS (plain loads issued a whole
//     group ahead; an LDS-DMA here made the compiler wait for the queue before every step's LDS reads, it cannot tell
//     the DMA's destination from the image being read);
//   * a step's operands (row t of A_k, delta_x_k, Q_i l_i) are requested together and waited for once;
//   * delta_x is double-buffered in LDS: one wave-level sync per step;
//   * ExpectedDecrease takes its terms from lanes 0 .. NP-1 through v_readlane, in the reference's order, beside the chain.
// CMP: A_k comes from compact rows (LQArgs::compact) — compile-time, so that each form holds only its own prefetch
// registers (the trial kernel that hosts this pass is at its register limit in both precisions).
template <typename T, int NX, int NP, int MU, int LDSE, bool CMP>
__device__ __forceinline__ void lq_forward_pass_wave(const LQArgs<T>& a, T* sm, int t) {
  using C = LQCfg<T, NX, NP, MU>;
  constexpr int SCR = C::SCR;
  constexpr int FSLOT = (NX * NX + SCR + 3) & ~3;
  constexpr int GFIT = (LDSE - 2 * NX - 4) / (2 * FSLOT);
  constexpr int G = GFIT < 1 ? 1 : (GFIT > 4 ? 4 : GFIT);
  static_assert(2 * G * FSLOT + 2 * NX <= LDSE, "forward-pass staging does not fit the LDS it is given");
  constexpr int ROW = NX * NX + SCR;           // elements of one step's image
  constexpr int PER = (ROW + 63) / 64;         // loads per lane and step
  constexpr int SQ = (SCR + 63) / 64;          // ... of which the scratch row's, when A comes from compact rows
  const int Tn = a.T_steps;
  T* sX = sm + 2 * G * FSLOT;                  // delta_x, two buffers of NX
  // explicit global address space: through generic pointers the prefetch loads below are FLAT loads, which count on the
  // LDS counter too — every wait for a step's LDS operands then also waited for the group being prefetched
  typedef const __attribute__((address_space(1))) T gcT;
  const gcT* gA = (const gcT*)uniform_ptr(a.A);
  const gcT* gS = (const gcT*)uniform_ptr(a.scratch);
  // Compact rows (LQArgs::compact): A_k is a constant background — written into every staging slot once — plus the
  // Jacobian pass's words of the step's compact row (lane = word; the words that belong to B are skipped).
  constexpr bool cmp = CMP;
  const gcT* gC = (const gcT*)uniform_ptr(a.compact);
  const int CWD = cmp ? a.compact_tab[RC_W] : 0;
  int nJ = 0, cbase = 0, cdstA = -1;
  if (cmp) {
    constexpr int NPB = NP + 1;  // bases: the Jacobian pass, then the players
    cbase = a.compact_tab[RC_BASE];
    nJ = a.compact_tab[RC_BASE + 1] - cbase;
    if (nJ > 64) nJ = 64;  // (the Jacobian pass of these games has 2 .. 6 words per subsystem)
    if (t < nJ) {
      const int code = a.compact_tab[RC_BASE + NPB + cbase + t];
      cdstA = (code >> 24) == RA_A ? (code & 0xffffff) : -1;
    }
    for (int e = t; e < 2 * G * FSLOT; e += 64) sm[e] = T(0);
    lds_sync(true);
    const int nbg = a.compact_tab[RC_NBG];
    const int* bg = a.compact_tab + RC_BASE + NPB + CWD;
    for (int e = t; e < nbg; e += 64) {
      const int code = bg[RC_BG_WORDS * e], kind = bg[RC_BG_WORDS * e + 1];
      if ((code >> 24) != RA_A) continue;
      const T v = kind == RC_DT ? T(a.dt) : (kind == RC_NEG_DT ? T(-a.dt) : T(__int_as_float(bg[RC_BG_WORDS * e + 2])));
      for (int sl = 0; sl < 2 * G; sl++) sm[sl * FSLOT + (code & 0xffffff)] = v;
    }
    lds_sync(true);
  }
  T pre[G][CMP ? 1 + SQ : PER];
  auto fetch_group = [&](int grp) {
#pragma unroll
    for (int s = 0; s < G; s++) {
      const int k = grp * G + s;
      if constexpr (cmp) {
        // [0]: this lane's word of the Jacobian pass, [1]: its word of the scratch row
        pre[s][0] = (k < Tn && t < nJ) ? gC[size_t(k) * CWD + cbase + t] : T(0);
#pragma unroll
        for (int q = 0; q < SQ; q++) pre[s][1 + q] = (k < Tn && q * 64 + t < SCR) ? gS[size_t(k) * SCR + q * 64 + t] : T(0);
      } else {
#pragma unroll
        for (int q = 0; q < PER; q++) {
          const int e = q * 64 + t;
          const bool inA = e < NX * NX;
          const gcT* src = inA ? gA + (size_t(k) * NX * NX + e) : gS + (size_t(k) * SCR + (e - NX * NX));
          pre[s][q] = (k < Tn && e < ROW) ? *src : T(0);
        }
      }
    }
  };
  auto commit_group = [&](int which) {
#pragma unroll
    for (int s = 0; s < G; s++) {
      if constexpr (cmp) {
        T* slot = sm + (which * G + s) * FSLOT;
        if (cdstA >= 0) slot[cdstA] = pre[s][0];
#pragma unroll
        for (int q = 0; q < SQ; q++)
          if (q * 64 + t < SCR) slot[NX * NX + q * 64 + t] = pre[s][1 + q];
      } else {
#pragma unroll
        for (int q = 0; q < PER; q++) {
          const int e = q * 64 + t;
          if (e < ROW) sm[(which * G + s) * FSLOT + e] = pre[s][q];
        }
      }
    }
  };
  fetch_group(0);
  if (t < NX) sX[t] = a.x0 ? a.x0[t] : T(0);
  commit_group(0);
  const int ngroups = (Tn + G - 1) / G;
  if (ngroups > 1) fetch_group(1);
  lds_sync(true);
  T ed = T(0);
  int cur = 0, xb = 0;
#pragma unroll 1
  for (int grp = 0; grp < ngroups; grp++) {
#pragma unroll 1
    for (int s = 0; s < G; s++) {
      const int k = grp * G + s;
      if (k >= Tn) break;
      const T* fA = sm + (cur * G + s) * FSLOT;
      const T* fS = fA + NX * NX;  // [ql (N*n) | ctrl (N) | beta (n)]
      const T* x = sX + xb * NX;
      const int tr = t < NX ? t : 0, tp = t < NP ? t : 0;
      T xv[NX], av[NX], qv[NX];
#pragma unroll
      for (int c = 0; c < NX; c++) xv[c] = x[c];
#pragma unroll
      for (int c = 0; c < NX; c++) av[c] = fA[tr + NX * c];
#pragma unroll
      for (int c = 0; c < NX; c++) qv[c] = fS[tp * NX + c];
      const T ct = fS[NP * NX + tp];
      const T beta = fS[NP * (NX + 1) + tr];  // beta_k = -B alpha_k
      if (a.dx && t < NX) a.dx[size_t(k) * NX + t] = xv[tr];
      T xn = T(0), st = T(0);
#pragma unroll
      for (int c = 0; c < NX; c++) {
        xn += av[c] * xv[c];
        st += xv[c] * qv[c];
      }
      xn += beta;
      if (t < NX) sX[(1 - xb) * NX + t] = xn;
      if (a.ed_out) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
          ed -= bcast(ct, i);
          if (k > 0) ed -= bcast(st, i);
        }
      }
      xb = 1 - xb;
      lds_sync(true);
    }
    if (grp + 1 < ngroups) {
      commit_group(1 - cur);  // waits for the loads issued a group ago
      if (grp + 2 < ngroups) fetch_group(grp + 2);
      lds_sync(true);
    }
    cur = 1 - cur;
  }
  if (a.ed_out && t == 0) *a.ed_out = ed;
}

template <typename T, int NX, int NP, int MU, int NT = LQCfg<T, NX, NP, MU>::NT, int LDSE = LQCfg<T, NX, NP, MU>::LDS_ELEMS>
__device__ __forceinline__ void lq_forward_pass_body(const LQArgs<T>& a, T* sm, int t) {
  using C = LQCfg<T, NX, NP, MU>;
  if constexpr (NT == 64) {
    if (a.compact)
      lq_forward_pass_wave<T, NX, NP, MU, LDSE, true>(a, sm, t);
    else
      lq_forward_pass_wave<T, NX, NP, MU, LDSE, false>(a, sm, t);
    return;
  }
  constexpr int SCR = C::SCR, S = int(sizeof(T));
  constexpr int FSLOT = (NX * NX + SCR + 3) & ~3;
  constexpr int GFIT = (LDSE - NX - 4) / (2 * FSLOT);
  constexpr int G = GFIT < 1 ? 1 : (GFIT > 8 ? 8 : GFIT);
  static_assert(2 * G * FSLOT + NX <= LDSE, "forward-pass staging does not fit the LDS of the sweep");
  constexpr bool wA = (NX * NX * S) % 16 == 0 && (FSLOT * S) % 16 == 0;
  const int Tn = a.T_steps;
  T* sX = sm + 2 * G * FSLOT;
  auto stage_group = [&](int grp, int which) {
#pragma unroll
    for (int s = 0; s < G; s++) {
      const int k = grp * G + s;
      if (k < Tn) {
        T* slot = sm + (which * G + s) * FSLOT;
        dma_g2l<NT, wA>(a.A + size_t(k) * NX * NX, slot, NX * NX * S, t);
        dma_g2l<NT, false>(a.scratch + size_t(k) * SCR, slot + NX * NX, SCR * S, t);
      }
    }
  };
  int cur = 0;
  stage_group(0, 0);
  if (t < NX) sX[t] = a.x0 ? a.x0[t] : T(0);
  T ed = T(0);
  dma_wait();
  lds_sync(NT <= 64);
  const int ngroups = (Tn + G - 1) / G;
#pragma unroll 1
  for (int grp = 0; grp < ngroups; grp++) {
    if (grp + 1 < ngroups) stage_group(grp + 1, 1 - cur);
#pragma unroll 1
    for (int s = 0; s < G; s++) {
      const int k = grp * G + s;
      if (k >= Tn) break;
      const T* fA = sm + (cur * G + s) * FSLOT;
      const T* fS = fA + NX * NX;  // [ql (N*n) | ctrl (N) | beta (n)]
      if (a.dx && t < NX) a.dx[size_t(k) * NX + t] = sX[t];
      if (a.ed_out && t < 64) {
        T st = T(0), ct = T(0);
        if (t < NP) {
          ct = fS[NP * NX + t];
          if (k > 0) {
#pragma unroll
            for (int c = 0; c < NX; c++) st += sX[c] * fS[t * NX + c];
          }
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {
          ed -= shfl(ct, i);
          if (k > 0) ed -= shfl(st, i);
        }
      }
      T xn = T(0);
      if (t < NX) {
#pragma unroll
        for (int c = 0; c < NX; c++) xn += fA[t + NX * c] * sX[c];
        xn += fS[NP * (NX + 1) + t];  // beta_k = -B alpha_k
      }
      lds_sync(NT <= 64);
      if (t < NX) sX[t] = xn;
      lds_sync(NT <= 64);
    }
    dma_wait();
    lds_sync(NT <= 64);
    cur = 1 - cur;
  }
  if (a.ed_out && t == 0) *a.ed_out = ed;
}

template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void lq_forward_pass(const LQArgs<T>& a, T* sm, int t) {
  if ((a.dx == nullptr && a.ed_out == nullptr) || a.defer_forward) return;
  __syncthreads();  // scratch rows were written to global memory by other lanes during the sweep
  lq_forward_pass_body<T, NX, NP, MU>(a, sm, t);
}

// One instance, executed by a workgroup of LQCfg::NT threads.  `sm` is LDS scratch
// of LQCfg::LDS_ELEMS elements.  All threads of the workgroup must call.
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void lq_feedback_instance(const LQArgs<T>& a, const PairTable& pt, T* sm) {
  using C = LQCfg<T, NX, NP, MU>;
  constexpr int M = C::M, L = C::L, NT = C::NT, NXS = C::NXS;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const bool zl = t < L;  // this lane owns a Z column
  const int pi = zl ? t / NX : 0;
  const int pc = zl ? t % NX : 0;
  const int Tn = a.T_steps;
  const bool want_fwd = a.dx != nullptr || a.ed_out != nullptr || a.defer_forward != 0;
  constexpr int SCR = C::SCR;

  T *sB, *sA, *sQ, *sl, *sR, *sr;  // views into the image of the step being processed
  auto set_img = [&](int which) {
    T* img = sm + which * C::IMG;
    sB = img + C::oB;
    sA = img + C::oA;
    sQ = img + C::oQ;
    sl = img + C::ol;
    sR = img + C::oR;
    sr = img + C::or_;
  };
  int cur = 0;
  T* sF = sm + C::oF;
  T* sUt = sm + C::oUt;
  T* sBZ = sm + C::oBZ;
  T* sP = sm + C::oP;
  T* sAl = sm + C::oAl;
  T* sBeta = sm + C::oBeta;
  T* sZeta = sm + C::oZeta;
  T* sYz = sm + C::oYz;

  // (Q_i l_i) of the step currently staged -> scratch, for ExpectedDecrease
  auto stash_ql = [&](int k) {
    if (want_fwd && zl) {
      T s = T(0);
#pragma unroll
      for (int c = 0; c < NX; c++) s += sQ[pi * NX * NX + pc + NX * c] * sl[pi * NX + c];
      a.scratch[size_t(k) * SCR + pi * NX + pc] = s;
    }
  };

  // ---- terminal step: Z_i = Q_i[T-1], zeta_i = l_i[T-1]  (:102-105) ----
  lq_stage_issue<T, NX, NP, MU>(a, pt, Tn - 1, sm, t);
  dma_wait();
  __syncthreads();
  set_img(0);
  T z[NX];
  T zeta = T(0);
  if (zl) {
#pragma unroll
    for (int r = 0; r < NX; r++) z[r] = sQ[pi * NX * NX + r + NX * pc];
    zeta = sl[pi * NX + pc];
  } else {
#pragma unroll
    for (int r = 0; r < NX; r++) z[r] = T(0);
  }
  stash_ql(Tn - 1);
  // strategies at T-1 stay zero (strategy.h:64-70)
  for (int e = t; e < M * NX; e += NT) a.P[size_t(Tn - 1) * M * NX + e] = T(0);
  if (t < M) a.alpha[size_t(Tn - 1) * M + t] = T(0);
  if (want_fwd) {
    if (t < NP) a.scratch[size_t(Tn - 1) * SCR + NP * NX + t] = T(0);
    if (t < NX) a.scratch[size_t(Tn - 1) * SCR + NP * (NX + 1) + t] = T(0);
  }
  if (Tn >= 2) lq_stage_issue<T, NX, NP, MU>(a, pt, Tn - 2, sm + C::IMG, t);
  if (zl) sZeta[t] = zeta;
  dma_wait();
  lds_sync(NT <= 64);
  cur = 1;
  set_img(1);

#pragma unroll 1
  for (int k = Tn - 2; k >= 0; k--) {
    if (k > 0) lq_stage_issue<T, NX, NP, MU>(a, pt, k - 1, sm + (1 - cur) * C::IMG, t);
    stash_ql(k);

    // ---- P1: BZ = B_i^T Z_i (rows of the stacked system), y_zeta = B_i^T zeta_i + r_ii ----
    if (zl) {
#pragma unroll
      for (int aa = 0; aa < MU; aa++) {
        T s = T(0);
#pragma unroll
        for (int r = 0; r < NX; r++) s += sB[r + NX * (pi * MU + aa)] * z[r];
        sBZ[(pi * MU + aa) + M * pc] = s;
      }
      if (pc < MU) {
        T s = T(0);
#pragma unroll
        for (int r = 0; r < NX; r++) s += sB[r + NX * (pi * MU + pc)] * sZeta[pi * NX + r];
        sYz[pi * MU + pc] = s + sr[pt.rgoff[pt.pii[pi]] + pc];
      }
    }
    lds_sync(NT <= 64);

    // ---- P2: column `t` of [S | Y], Gershgorin, QR solve (wave 0) ----
    if (t < 64) {
      T col[M], x[M];
#pragma unroll
      for (int r = 0; r < M; r++) { col[r] = T(0); x[r] = T(0); }
      if (t < M + NX) {
        // column t of [B | A] is contiguous in the staged image
        T mc[NX];
#pragma unroll
        for (int c = 0; c < NX; c++) mc[c] = sB[c + NX * t];  // column t of [B | A], contiguous in the image
#pragma unroll
        for (int r = 0; r < M; r++) {
          T s = T(0);
#pragma unroll
          for (int c = 0; c < NX; c++) s += sBZ[r + M * c] * mc[c];
          col[r] = s;
        }
        if (t < M) {
          // + R_ii on the diagonal block (:131-149)
          const int pj = t / MU, b = t % MU;
          const T* Rii = sR + pt.roff[pt.pii[pj]];
#pragma unroll
          for (int r = 0; r < M; r++)
            if (r / MU == pj) col[r] = col[r] + Rii[(r % MU) + MU * b];
          if (a.adaptive) {  // :163-176 — columns are independent, so lane-parallel is exact
            T l1 = T(0), diag = T(0);
#pragma unroll
            for (int r = 0; r < M; r++) {
              l1 += (col[r] < T(0) ? -col[r] : col[r]);
              if (r == t) diag = col[r];
            }
            const T radius = l1 - (diag < T(0) ? -diag : diag);
            const T eval_lo = diag - radius;
            if (eval_lo < T(1e-3f)) {
#pragma unroll
              for (int r = 0; r < M; r++)
                if (r == t) col[r] += radius + T(1e-3f);
            }
          }
        }
      } else if (t == M + NX) {
#pragma unroll
        for (int r = 0; r < M; r++) col[r] = sYz[r];
      }
      qr_solve_columns<T, M>(col, lane, x);
      if (t >= M && t < M + NX) {
#pragma unroll
        for (int r = 0; r < M; r++) {
          sP[r + M * (t - M)] = x[r];
          a.P[size_t(k) * M * NX + r + M * (t - M)] = x[r];
        }
      } else if (t == M + NX) {
#pragma unroll
        for (int r = 0; r < M; r++) {
          sAl[r] = x[r];
          a.alpha[size_t(k) * M + r] = x[r];
        }
      }
    }
    lds_sync(NT <= 64);

    // ---- P3: F[:,c] = A[:,c] - B P[:,c]; beta = -B alpha ----
    T f[NX], pcol[M];
#pragma unroll
    for (int r = 0; r < M; r++) pcol[r] = zl ? sP[r + M * pc] : T(0);
#pragma unroll
    for (int r = 0; r < NX; r++) {
      T s = zl ? sA[r + NX * pc] : T(0);
#pragma unroll
      for (int q = 0; q < M; q++) s -= sB[r + NX * q] * pcol[q];
      f[r] = s;
    }
    if (t < NX) {
#pragma unroll
      for (int r = 0; r < NX; r++) sF[r + NXS * t] = f[r];
      T s = T(0);
#pragma unroll
      for (int q = 0; q < M; q++) s -= sB[t + NX * q] * sAl[q];
      sBeta[t] = s;
      if (want_fwd) a.scratch[size_t(k) * SCR + NP * (NX + 1) + t] = s;
    }
    if (want_fwd && t < NP) {
      // alpha_i^T R_ii r_ii, evaluated (alpha^T R) r like Eigen (ilq_solver.cpp:384-386)
      const int q = pt.pii[t];
      T acc = T(0);
#pragma unroll
      for (int c = 0; c < MU; c++) {
        T aR = T(0);
#pragma unroll
        for (int b = 0; b < MU; b++) aR += sAl[t * MU + b] * sR[pt.roff[q] + b + MU * c];
        acc += aR * sr[pt.rgoff[q] + c];
      }
      a.scratch[size_t(k) * SCR + NP * NX + t] = acc;
    }
    lds_sync(NT <= 64);

    // ---- P4: U_i[:,c] = F^T Z_i[:,c], stored transposed for row access ----
    if (zl) {
#pragma unroll
      for (int r = 0; r < NX; r++) {
        T s = T(0);
#pragma unroll
        for (int kk = 0; kk < NX; kk++) s += sF[kk + NXS * r] * z[kk];
        sUt[(pi * NX + r) * NXS + pc] = s;
      }
    }
    lds_sync(NT <= 64);

    // ---- P5: Z_i'[:,c] = U_i F[:,c] + Q_i[:,c] + sum_j P_j^T R_ij P_j[:,c]; zeta update ----
    T zeta_new = T(0);
    if (zl) {
#pragma unroll
      for (int r = 0; r < NX; r++) {
        T s = T(0);
#pragma unroll
        for (int cc = 0; cc < NX; cc++) s += sUt[(pi * NX + r) * NXS + cc] * f[cc];
        z[r] = s + sQ[pi * NX * NX + r + NX * pc];
      }
      // zeta_i'[c] = F[:,c].zeta_i + U_i[c,:].beta + l_i[c] + ...
      T s1 = T(0), s2 = T(0);
#pragma unroll
      for (int kk = 0; kk < NX; kk++) {
        s1 += f[kk] * sZeta[pi * NX + kk];
        s2 += sUt[(pi * NX + pc) * NXS + kk] * sBeta[kk];
      }
      zeta_new = (s1 + s2) + sl[pi * NX + pc];
      for (int q = 0; q < pt.npairs; q++) {
        if (pt.pi[q] != pi) continue;
        const int j = pt.pj[q];
        const T* Rij = sR + pt.roff[q];
        const T* rij = sr + pt.rgoff[q];
        T v[MU], w[MU];
#pragma unroll
        for (int aa = 0; aa < MU; aa++) {
          T sv = T(0), sw = T(0);
#pragma unroll
          for (int b = 0; b < MU; b++) {
            sv += Rij[aa + MU * b] * sP[(j * MU + b) + M * pc];
            sw += Rij[aa + MU * b] * sAl[j * MU + b];
          }
          v[aa] = sv;
          w[aa] = sw - rij[aa];
        }
        T add = T(0);
#pragma unroll
        for (int aa = 0; aa < MU; aa++) {
          add += sP[(j * MU + aa) + M * pc] * w[aa];
        }
        zeta_new += add;
#pragma unroll
        for (int r = 0; r < NX; r++) {
          T s = T(0);
#pragma unroll
          for (int aa = 0; aa < MU; aa++) s += sP[(j * MU + aa) + M * r] * v[aa];
          z[r] += s;
        }
      }
    }
    lds_sync(NT <= 64);
    if (zl) sZeta[t] = zeta_new;
    dma_wait();
    lds_sync(NT <= 64);
    cur = 1 - cur;
    set_img(cur);
  }

  lq_forward_pass<T, NX, NP, MU>(a, sm, t);
}

// ---------------------------------------------------------------------------
// Player-parallel MFMA sweep: a workgroup of NP wavefronts per instance, wave i owns player i.
//
// The step's dependency chain is  [S|Y] rows  ->  M x M Nash solve  ->  F  ->  Z_i update; the first
// and the last link are per player and independent of each other, so each wave keeps ITS Z_i / Z_i^T
// in registers and produces its MU rows of [S | Y] and its own Z_i', zeta_i'.  Only the small solve is
// serial (wave 0, column per lane).  Three workgroup barriers per step: [S|Y] complete,
// (P, alpha) published, image swap (two with compact rows).
//
// Round 5: an fp64 matrix instruction holds the SIMD's vector pipe for 64 cycles and the LDS executes
// one instruction per ~3-4 cycles per CU, so with three waves per SIMD the sweep was bound by the instructions it
// issues.  Everything that is a matrix-VECTOR product or has two useful rows — the player's columns of G = Z_w^T B,
// its rows of [S | Y], Q_w l_w, for n = 16 also Z_w beta and F^T t — is a row sum over the accumulator-layout registers
// (v_permlane32_swap / v_permlane16_swap reductions, ilqg_mfma.hpp) instead of a tile product or an LDS gather; the
// matrix cores keep F (2), the C terms (2), Z_w F (4) and F^T W (4).
//
// Staging: A, B and every Q_i are DMA'd (global_load_lds with a per-lane gather address) straight into
// zero-padded 16 x 16 LDS tiles, so every accumulator-layout operand is read with one per-lane base
// offset plus immediates — no predication, no selects, no masks kept live across the step loop; the
// padding is zeroed once per sweep and never written again (the DMA only touches the valid pieces).
//
// When the state leaves a spare tile column (NX < 16) two matrix-vector products ride along in
// column NX of tile products that are needed anyway:  W_i = Z_i [F | beta]  carries Z_i beta, and
// Z_i' = F^T [W_i | zeta_i + Z_i beta] + C_i  carries F^T (zeta_i + Z_i beta).
// ---------------------------------------------------------------------------
template <typename T, int NX, int NP, int MU>
struct PWCfg {
  using C = LQCfg<T, NX, NP, MU>;
  static constexpr int M = NP * MU;
  // leading dimension of a padded tile: 16 would put every other column on the same LDS banks
  // (8-way conflicts on accumulator-layout reads); an odd number of elements spreads them.  fp64 ran on 18 until round 4
  // (16-byte DMA pieces for the dense tiles); with compact rows nothing is DMA'd into the tiles and 17 / 19 / 21 all
  // measure 2.2 % faster on the headline batch (1.444 -> 1.476 M it/s; SQ_LDS_BANK_CONFLICT was 17 % of the LDS-active
  // cycles, the LDS busy 59 % of the kernel) — the smallest it is.  The dense path pays with 4-byte DMA pieces.
  static constexpr int LD = 17;
  static constexpr int TILE = (16 * LD + 3) & ~3;
  // one staged step: [tA | tB | tQ_0.. | l | R | r]
  static constexpr int oTA = 0;
  static constexpr int oTB = oTA + TILE;
  static constexpr int oTQ = oTB + TILE;
  static constexpr int oVl = oTQ + NP * TILE;
  static constexpr int oVR = (oVl + NP * NX + 3) & ~3;
  static constexpr int oVr = (oVR + C::RMAX + 3) & ~3;
  static constexpr int IMG = (oVr + C::rMAX + 3) & ~3;
  // intermediates
  static constexpr int oPt = 2 * IMG;           // P as a padded tile: [row][col] at row + 16*col
  static constexpr int oAl = oPt + TILE;        // alpha (M, padded to 16)
  static constexpr int oYz = oAl + 16;          // y_zeta (M, padded to 16)
  static constexpr int oSY = oYz + 16;          // [S | Y] bounce: M x 32, column-major
  static constexpr int oVec = oSY + M * 32;     // per player: beta strip (16) and zeta strip (16)
  static constexpr int oG = oVec + 2 * NP * 16;  // per player: its MU columns of G = Z^T B, 16 entries each
  static constexpr int oSB = oG + NP * MU * 16;  // compact rows: two staging rows (the DMA's landing place, one step ahead)
  static constexpr int oCD = oSB + 2 * kCompactMaxWords;  // ... and where each word of a row goes in an image (ints)
  static constexpr int ELEMS_PW = oCD + kCompactMaxWords;
  // the forward pass reuses the LDS with the single-wave layout
  static constexpr int LDS_ELEMS = ELEMS_PW > C::LDS_ELEMS ? ELEMS_PW : C::LDS_ELEMS;
  // DMA piece: 16 bytes when every column of the source and of the padded tile starts 16-byte aligned, else 4
  static constexpr int PS = ((NX * int(sizeof(T))) % 16 == 0 && (LD * int(sizeof(T))) % 16 == 0) ? 16 : 4;
  static constexpr int WAVE_INSTRS = (16 * LD * int(sizeof(T)) / PS + 63) / 64;  // DMA instructions of one wave per tile
};

// DMA of an (nrows x ncols) column-major matrix with leading dimension NX into a zero-padded 16 x 16
// tile, executed by ONE wave.  Lane `lane` of instruction h moves piece 64*h + lane of the tile.
template <typename T, int NX, int LD, int PS, int WI>
__device__ __forceinline__ void dma_tile(const T* g_, T* tile, int nrows, int ncols, int lane) {
  const T* g = uniform_ptr(g_);
  constexpr int COLB = LD * int(sizeof(T));  // bytes of a padded column
#pragma unroll
  for (int h = 0; h < WI; h++) {
    const int pos = (64 * h + lane) * PS;  // byte position inside the padded tile
    const int c = pos / COLB, inb = pos % COLB;
    if (c < ncols && inb + PS <= nrows * int(sizeof(T))) {
      // unsigned: a wave-uniform base plus a zero-extended 32-bit lane offset (the scalar-base addressing mode; a signed
      // offset makes the compiler keep a 64-bit per-lane address per instruction live across the step loop)
      const char* src = reinterpret_cast<const char*>(g) + unsigned(c * NX * int(sizeof(T)) + inb);
      char* dst = reinterpret_cast<char*>(tile) + 64 * h * PS;  // wave-uniform; the hardware adds lane * PS
      if constexpr (PS == 16)
        __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)dst, 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((glb_vptr)src, (lds_vptr)dst, 4, 0, 0);
    }
  }
}

// SOLVER: the instantiation the solve's sweep kernel runs (compact rows in, symmetric costs, regularised system, forward
// pass deferred to the trial kernel) with those choices made at compile time: the dense staging, the QR solve and the
// forward pass are not in its loop or its register allocation.
template <typename T, int NX, int NP, int MU, bool SOLVER = false>
__device__ __forceinline__ void lq_feedback_instance_mfma_pw(const LQArgs<T>& a, const PairTable& pt, T* sm) {
  using C = LQCfg<T, NX, NP, MU>;
  using W = PWCfg<T, NX, NP, MU>;
  using TL = Tile<T>;
  using vec = typename TL::vec;
  constexpr int M = C::M, NT = 64 * NP, NS = C::NSOLVE, S = int(sizeof(T)), LD = W::LD;
  static_assert(M <= 16 && NS <= 32, "the Nash system must fit two 16-column tiles");
  constexpr bool SPARE = NX < 16;     // tile column NX is free
  constexpr int JB = SPARE ? NX : 0;  // the column that carries vector operands when SPARE
  const int t = threadIdx.x;
  const int w = t >> 6;  // wave = player
  const int lane = t & 63, g = lane >> 4, j = lane & 15;
  const int Tn = a.T_steps;
  const PairRegs<NP> pr(pt);
  const bool want_fwd = SOLVER || a.dx != nullptr || a.ed_out != nullptr || a.defer_forward != 0;
  constexpr int SCR = C::SCR;
  const vec zero4 = {T(0), T(0), T(0), T(0)};
  constexpr int RS = TL::row(0, 1) - TL::row(0, 0);  // row step between accumulator registers
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
  // 0/1 multipliers instead of selects (all masked quantities are finite)
  const T mCols = (j < NX) ? T(1) : T(0);      // a proper state column
  const T mVecCol = (j == (SPARE ? JB : w)) ? T(1) : T(0);
  // SPARE: zeta_w rides in column JB of the Z_w tile (rows < NX), so every product with Z_w carries the matching
  // product with zeta_w: row JB of G = Z_w^T B is zeta_w^T B (y_zeta), column JB of Z_w' = F^T [..] + C_w is the new
  // zeta_w once column JB of C_w holds its additive terms.  Register rJ of lane group gJ holds row JB.
  constexpr int gJ = sizeof(T) == 8 ? JB % 4 : JB / 4, rJ = sizeof(T) == 8 ? JB / 4 : JB % 4;
  static_assert(!SPARE || TL::row(gJ, rJ) == JB, "accumulator-layout position of row JB");
  const T mZcols = SPARE ? mCols + mVecCol : mCols;  // the columns of the Z_w tile that are kept

  // this player's offsets in the R / r rows (wave-uniform selects on a register table)
  int ro_ww = 0, rg_ww = 0;
#pragma unroll
  for (int e = 0; e < NP; e++) {
    ro_ww = (w == e) ? pr.ro[e][e] : ro_ww;
    rg_ww = (w == e) ? pr.rg[e][e] : rg_ww;
  }

  T* img = sm;  // image of the step being processed
  T *tA, *tB, *tQ, *sl, *sR, *sr;
  auto set_img = [&](int which) {
    img = sm + which * W::IMG;
    tA = img + W::oTA;
    tB = img + W::oTB;
    tQ = img + W::oTQ + w * W::TILE;
    sl = img + W::oVl;
    sR = img + W::oVR;
    sr = img + W::oVr;
  };
  T* sPt = sm + W::oPt;
  T* sAl = sm + W::oAl;
  const T* const sAlr = SPARE ? sm + W::oPt + LD * JB : sAl;  // where alpha is read from (SPARE: column JB of the [P | alpha] tile)
  T* sYz = sm + W::oYz;
  T* sSY = sm + W::oSY;
  T* sBw = sm + W::oVec + w * 16;         // this player's beta, entries NX..15 zero
  T* sZw = sm + W::oVec + (NP + w) * 16;  // this player's zeta, entries NX..15 zero
  T* sGw = sm + W::oG + w * MU * 16;       // this player's columns of Z_w^T B

  // Staging jobs of one step: Q_0 .. Q_{NP-1}, A, B, the vectors (l, R, r) — NP + 3 of them, each executed
  // by one wave.  `slot` of `nslots` takes every nslots-th job.  Before the loop all NP waves share the
  // work; inside it the waves that would otherwise idle through wave 0's solve do all of it, so staging
  // (and the Q_i l_i products for ExpectedDecrease) leaves the step's critical path.
  auto stage = [&](int k, int which, int slot, int nslots) {
    T* dst = sm + which * W::IMG;
#pragma unroll
    for (int job = 0; job < NP + 3; job++) {
      if (job % nslots != slot) continue;
      if (job < NP) {
        dma_tile<T, NX, W::LD, W::PS, W::WAVE_INSTRS>(a.Q + (size_t(k) * NP + job) * NX * NX, dst + W::oTQ + job * W::TILE,
                                                        NX, NX, lane);
      } else if (job == NP) {
        dma_tile<T, NX, W::LD, W::PS, W::WAVE_INSTRS>(a.A + size_t(k) * NX * NX, dst + W::oTA, NX, NX, lane);
      } else if (job == NP + 1) {
        dma_tile<T, NX, W::LD, W::PS, W::WAVE_INSTRS>(a.Bm + size_t(k) * NX * M, dst + W::oTB, NX, M, lane);
      } else {
        dma_g2l<64, false>(a.l + size_t(k) * NP * NX, dst + W::oVl, NP * NX * S, lane);
        dma_g2l<64, false>(a.R + size_t(k) * pt.Rsz, dst + W::oVR, pt.Rsz * S, lane);
        dma_g2l<64, false>(a.r + size_t(k) * pt.rsz, dst + W::oVr, pt.rsz * S, lane);
      }
    }
  };

  // Compact rows (LQArgs::compact): nothing dense is read; the touched words of [A | B | Q | l | R | r] arrive as one
  // short row that a wave scatters over the image's constant background (written once, below).  The row of step
  // k - 2 is DMA'd into a staging row while step k runs and scattered into the image of step k - 1 ... one step later.
  const bool cmp = SOLVER || a.compact != nullptr;
  T* const sSB = sm + W::oSB;
  auto cdecode = [&](int code) -> int {  // array << 24 | offset in the array's row  ->  offset inside an image
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
  const int CWD = cmp ? a.compact_tab[RC_W] : 0;
  int* const sCD = reinterpret_cast<int*>(sm + W::oCD);  // where each word of a compact row goes in an image (-1: none);
                                                         // kept in LDS: the sweep has no registers to spare
  auto stage_c = [&](int k, int which, int slot, int nslots) {
    T* dst = sm + which * W::IMG;
#pragma unroll
    for (int job = 2; job < 4; job++) {
      if (job % nslots != slot) continue;
      if (job == 2) {
        const T* row = sSB + (k & 1) * kCompactMaxWords;
        constexpr int WPL = kCompactMaxWords / 64;  // words per lane
        T v[WPL];
        int cd[WPL];
#pragma unroll
        for (int q = 0; q < WPL; q++) {
          cd[q] = sCD[lane + 64 * q];
          v[q] = row[lane + 64 * q];
        }
#pragma unroll
        for (int q = 0; q < WPL; q++)
          if (cd[q] >= 0) dst[cd[q]] = v[q];
      } else if (k >= 1) {
        if ((CWD * S) % 16 == 0)  // rows start 16-byte aligned: 16-byte pieces (a row is one or two instructions)
          dma_g2l<64, true>(a.compact + size_t(k - 1) * CWD, sSB + ((k - 1) & 1) * kCompactMaxWords, CWD * S, lane);
        else
        dma_g2l<64, false>(a.compact + size_t(k - 1) * CWD, sSB + ((k - 1) & 1) * kCompactMaxWords, CWD * S, lane);
      }
    }
  };
  // the same for a step whose compact row is not staged yet (before the loop): straight from global memory
  auto stage_c_sync = [&](int k, int which) {
    T* dst = sm + which * W::IMG;
    for (int c = t; c < CWD; c += NT) dst[cdecode(a.compact_tab[RC_BASE + NP + 1 + c])] = a.compact[size_t(k) * CWD + c];
  };

  // (Q_i l_i) of the staged step -> scratch, for ExpectedDecrease
  auto stash_ql_of = [&](int k, int i) {
    if (want_fwd && lane < NX) {
      const T* tQi = img + W::oTQ + i * W::TILE;
      T s = T(0);
#pragma unroll
      for (int c = 0; c < NX; c++) s += tQi[lane + LD * c] * sl[i * NX + c];
      a.scratch[size_t(k) * SCR + i * NX + lane] = s;
    }
  };
  auto stash_ql = [&](int k) { stash_ql_of(k, w); };
  constexpr int HELPERS = NP > 1 ? NP - 1 : 1;  // waves that stage inside the loop (wave 0 itself when NP == 1)

  // ---- zero the tile padding (and everything else the DMA does not write), once ----
  for (int e = t; e < W::ELEMS_PW; e += NT) sm[e] = T(0);
  __syncthreads();

  if (cmp) {
    for (int c = t; c < kCompactMaxWords; c += NT) sCD[c] = c < CWD ? cdecode(a.compact_tab[RC_BASE + NP + 1 + c]) : -1;
  }
  if (cmp) {  // the images' constants that are not zero (the identity and the dt entries of A, B; untouched diagonal entries of Q_i)
    const int nbg = a.compact_tab[RC_NBG];
    const int* bg = a.compact_tab + RC_BASE + NP + 1 + CWD;
    for (int e = t; e < nbg; e += NT) {
      const int off = cdecode(bg[RC_BG_WORDS * e]);
      const int kind = bg[RC_BG_WORDS * e + 1];
      const T v = kind == RC_DT ? T(a.dt) : (kind == RC_NEG_DT ? T(-a.dt) : T(__int_as_float(bg[RC_BG_WORDS * e + 2])));
      sm[off] = v;
      sm[W::IMG + off] = v;
    }
  }
  // ---- terminal step: Z_w = Q_w[T-1], zeta_w = l_w[T-1]  (:102-105) ----
  if (cmp)
    stage_c_sync(Tn - 1, 0);
  else
    stage(Tn - 1, 0, w, NP);
  dma_wait();
  __syncthreads();
  set_img(0);
  // With symmetric costs Z_w is symmetric up to rounding, so Z_w^T (only ever used as the left factor of
  // W = Z_w F) is replaced by Z_w itself: two of the nine tile products of a step and the transposed loads
  // disappear.  The C ABI's general LQ entry (arbitrary Q, R) keeps both layouts.
  const bool sym = SOLVER || a.symmetric != 0;
  const bool adaptive = SOLVER || a.adaptive != 0;
  vec Zd = ldD(tQ);
  vec Yd = sym ? Zd : ldDT(tQ);
  if constexpr (SPARE) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = row0 + RS * r;
      Zd[r] += (row < NX ? sl[w * NX + (row < NX ? row : 0)] : T(0)) * mVecCol;  // column JB of the padded Q tile is zero
    }
    if (sym) Yd = Zd;
  } else {
    if (lane < NX) sZw[lane] = sl[w * NX + lane];
  }
  stash_ql(Tn - 1);
  for (int e = t; e < M * NX; e += NT) a.P[size_t(Tn - 1) * M * NX + e] = T(0);
  if (t < M) a.alpha[size_t(Tn - 1) * M + t] = T(0);
  if (want_fwd) {
    if (t < NP) a.scratch[size_t(Tn - 1) * SCR + NP * NX + t] = T(0);
    if (t < NX) a.scratch[size_t(Tn - 1) * SCR + NP * (NX + 1) + t] = T(0);
  }
  if (Tn >= 2) {
    if (cmp) {
      stage_c_sync(Tn - 2, 1);
      if (Tn >= 3 && w == 0)  // the first step's helpers scatter row T-3: it has to be staged by then
        dma_g2l<64, false>(a.compact + size_t(Tn - 3) * CWD, sSB + ((Tn - 3) & 1) * kCompactMaxWords, CWD * S, lane);
    } else {
      stage(Tn - 2, 1, w, NP);
    }
  }
  dma_wait();
  __syncthreads();
  int cur = 1;
  set_img(1);

  long long phacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // phase profile, kept in registers until the sweep ends
  tl_stamp(a.tl, a.tl_b, 17, t == 0);
  if (kTimeline && a.tl && lane == 0 && w < 4)  // where this wave sits: HW_ID | XCC_ID << 32
    a.tl[size_t(a.tl_b) * 96 + 32 + 24 + w] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                              ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#pragma unroll 1
  for (int k = Tn - 2; k >= 0; k--) {
    if (a.prio_div > 0) {
      // Issue arbitration among the instances that share a CU is by priority, then AGE: left alone the workgroup that
      // arrived first runs at the speed of a lone instance and the last one pays for it — and the launch ends when the
      // last one does (measured, B = 1024 fp64: finishing times 356 / 386 / 428 / 465 us in arrival order).  Rotating
      // the user priority with the step index shares the delay out: every instance ends within 10 us of 430 us.
      // Only when the whole batch is resident at once (prio_div = CUs: blocks b, b + CUs, ... share a CU).
      set_wave_prio(k + int(blockIdx.x) / a.prio_div);
    }
    long long pc0 = (kProfile && a.ph) ? clock64() : 0, pc1;
#define ILQG_PH(i) do { if (kProfile && a.ph) { __builtin_amdgcn_sched_barrier(0); pc1 = clock64(); __builtin_amdgcn_sched_barrier(0); phacc[i] += pc1 - pc0; pc0 = pc1; } } while (0)
    if (NP == 1) {
      if (k > 0) {
        if (cmp)
          stage_c(k - 1, 1 - cur, 0, 1);
        else
          stage(k - 1, 1 - cur, 0, 1);
      }
      if constexpr (!SOLVER)
      stash_ql(k);
    }
    ILQG_PH(0);

    // ---- this player's MU rows of the stacked Nash system: (B_w^T Z_w) [B | A] ----
    // Only this player's MU columns of G = Z_w^T B are needed, and of the MU x (M + NX) block G_w^T [B | A] that follows
    // two rows of a tile would be useful: neither is a job for the matrix pipe, which the SIMD's resident waves share
    // (an fp64 matrix instruction holds it for 64 cycles).
    const vec Bd = ldD(tB);
    constexpr bool kRowSums = MU == 2;  // the vector-unit forms below (rows_reduce4 packs (S, Y) x two controls)
    T ba[kRowSums ? 1 : NX];
    if constexpr (!kRowSums) {
      // column `lane` of [B | A], requested while the matrix pipe forms G (lanes past the last column re-read column 0)
      const T* colp = lane < M ? tB + LD * lane : tA + LD * (lane < M + NX ? lane - M : 0);
#pragma unroll
      for (int kk = 0; kk < NX; kk++) ba[kk] = colp[kk];
    }
    if constexpr (kRowSums) {
      // G[:, w MU + aa] as two matrix-vector products: lane (g, j) multiplies its four rows of column j of the Z_w tile
      // with B[row][w MU + aa] (read from the B tile: the address depends on g only) and the four lane rows are summed on
      // the vector unit (v_permlane32_swap / v_permlane16_swap).  4 LDS reads + 8 FMA + 6 permlane / add instead of four
      // matrix instructions.  SPARE: zeta_w rides in column JB of the Z_w tile, so entry JB of the result is zeta_w^T B_w;
      // otherwise (n = 16) zeta_w^T B_w is formed beside it from the same B entries and this lane's rows of zeta_w.
      T p0 = T(0), p1 = T(0), z0 = T(0), z1 = T(0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const T* bp = tB + (row0 + RS * r) + LD * (w * MU);
        const T b0 = bp[0], b1 = bp[LD];
        p0 += Zd[r] * b0;
        p1 += Zd[r] * b1;
        if constexpr (!SPARE) {
          const T zr = sZw[row0 + RS * r];
          z0 += zr * b0;
          z1 += zr * b1;
        }
      }
      T sa, sb;
      permlane32_swap(p0, p1, sa, sb);
      const T tt = sa + sb;  // rows: p0(0+2), p0(1+3), p1(0+2), p1(1+3)
      permlane16_swap(tt, tt, sa, sb);
      const T gcol = sa + sb;  // rows 0, 1: G[j][w MU]; rows 2, 3: G[j][w MU + 1]
      T ycol = T(0);
      if constexpr (!SPARE) {
        permlane32_swap(z0, z1, sa, sb);
        const T tz = sa + sb;
        permlane16_swap(tz, tz, sa, sb);
        ycol = sa + sb;  // rows 0, 1: zeta_w^T B[:, w MU]; rows 2, 3: zeta_w^T B[:, w MU + 1] (every lane of the row)
      }
      if ((g & 1) == 0) {
        const int aa = g >> 1;
        sGw[j * MU + aa] = gcol;  // interleaved [row][aa]: a lane reads the MU entries of a row with one instruction
        // y_zeta = B_w^T zeta_w + r_ww (:154-157)
        if constexpr (SPARE) {
          if (j == JB) sYz[w * MU + aa] = gcol + sr[rg_ww + aa];
        } else {
          if (j == 0) sYz[w * MU + aa] = ycol + sr[rg_ww + aa];
        }
      }
    } else {
      const vec G = tile_xty<T>(Zd, Bd, zero4);  // Z_w^T B
      if (j / MU == w) {
#pragma unroll
        for (int r = 0; r < 4; r++) sGw[(row0 + RS * r) * MU + (j - w * MU)] = G[r];
        if constexpr (SPARE) {
          // y_zeta = B_w^T zeta_w + r_ww (:154-157): row JB of G (zeta_w rides in column JB of the Z_w tile)
          if (g == gJ) sYz[j] = G[rJ] + sr[rg_ww + (j - w * MU)];
        }
      }
    }
    lds_sync(true);
    ILQG_PH(12);
    if constexpr (kRowSums) {
      // Lane (g, c) takes the rows {row(g, r)} of the dot products G_w[:, aa]^T [B | A][:, c] for column c of the B tile
      // AND of the A tile — the operands are the accumulator-layout registers Bd / Ad it holds anyway — and the four
      // partial sums per lane (S / Y part x two controls) are reduced over the lane rows on the vector unit
      // (rows_reduce4).  Four 16-byte LDS reads per lane instead of NX + NX / 2, 4-term FMA chains instead of NX-term.
      const vec Ad2 = ldD(tA);
      T pS[MU], pY[MU];
#pragma unroll
      for (int aa = 0; aa < MU; aa++) pS[aa] = pY[aa] = T(0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const T* gp = sGw + (row0 + RS * r) * MU;
#pragma unroll
        for (int aa = 0; aa < MU; aa++) {
          const T gval = gp[aa];
          pS[aa] += gval * Bd[r];
          pY[aa] += gval * Ad2[r];
        }
      }
      // rows after the reduction: 0 = S(aa 0), 1 = Y(aa 0), 2 = S(aa 1), 3 = Y(aa 1)
      const T tot = rows_reduce4<T>(pS[0], pS[MU - 1], pY[0], pY[MU - 1]);
      const int aa = g >> 1;
      const bool isY = (g & 1) != 0;
      const int c = isY ? M + j : j;  // column of [S | Y]
      // + R_ww on this player's diagonal block of S (:148-150); every lane reads (a clamped address): no divergent region
      const bool diag = !isY && j / MU == w;
      const T rd = sR[ro_ww + aa + MU * (diag ? j - w * MU : 0)];
      const T val = tot + (diag ? rd : T(0));
      if (isY ? j < NX : j < M) sSY[(w * MU + aa) + M * c] = val;
    } else {
      // lane c < M + NX takes column c of [B | A] and the MU dot products over the state (sequential sums)
      T gv[NX][MU];
#pragma unroll
      for (int kk = 0; kk < NX; kk++)
#pragma unroll
        for (int aa = 0; aa < MU; aa++) gv[kk][aa] = sGw[kk * MU + aa];
      T acc[MU];
#pragma unroll
      for (int aa = 0; aa < MU; aa++) acc[aa] = T(0);
#pragma unroll
      for (int kk = 0; kk < NX; kk++)
#pragma unroll
        for (int aa = 0; aa < MU; aa++) acc[aa] += gv[kk][aa] * ba[kk];
      if (lane < M + NX) {
        // + R_ww on this player's diagonal block of S (:148-150)
        const bool diag = lane / MU == w;
        const int b = lane - w * MU;
#pragma unroll
        for (int aa = 0; aa < MU; aa++)
          sSY[(w * MU + aa) + M * lane] = acc[aa] + (diag ? sR[ro_ww + aa + MU * (diag ? b : 0)] : T(0));
      }
    }
    ILQG_PH(13);
    if constexpr (SPARE || kRowSums) {
    } else if (lane < MU) {
      const int tt = w * MU + lane;
      T s = T(0);
#pragma unroll
      for (int r = 0; r < NX; r++) s += tB[r + LD * tt] * sZw[r];
      sYz[tt] = s + sr[rg_ww + lane];
    }
    ILQG_PH(1);
    if (cmp) dma_wait();  // the compact row requested a step ago has landed (see the end of the step)
    lds_sync(false);  // [S | Y] and y_zeta complete (LDS only: no wait on the DMA or on global stores)
    ILQG_PH(7);
    if (NP > 1 && w != 0) {  // while wave 0 solves: next step's image and this step's Q_i l_i
      if (k > 0) {
        if (cmp)
          stage_c(k - 1, 1 - cur, w - 1, HELPERS);
        else
          stage(k - 1, 1 - cur, w - 1, HELPERS);
      }
      ILQG_PH(14);
      if constexpr (!SOLVER)
      {
        stash_ql(k);
        if (w == 1) stash_ql_of(k, 0);
      }
      ILQG_PH(15);
    }

    // ---- wave 0: column `lane` of [S | Y]: + R_ii, Gershgorin (:163-176), then the M x M solve (:180) ----
    if (w == 0) {
      // Written without divergent regions: every lane loads a column (lanes past the last right-hand side
      // re-read y_zeta and are never stored), the R_ii and Gershgorin terms are added as value-or-zero.
      T col[M], x[M];
      const bool isS = lane < M;
      const T* src = (lane < M + NX) ? sSY + M * lane : sYz;
#pragma unroll
      for (int r = 0; r < M; r++) {
        col[r] = src[r];
        x[r] = T(0);
      }
      {
        // Gershgorin (columns are independent, so lane-parallel reproduces the sequential loop)
        // (|x| as a source modifier of the add; the diagonal entry re-read from LDS instead of a select chain)
        T l1 = T(0);
#pragma unroll
        for (int r = 0; r < M; r++) l1 += lq_abs(col[r]);
        const T diag = src[isS ? lane : 0];
        const T radius = l1 - lq_abs(diag);
        const T eval_lo = diag - radius;
        const T bump = (isS && adaptive && eval_lo < T(1e-3f)) ? radius + T(1e-3f) : T(0);
#pragma unroll
        for (int r = 0; r < M; r++) col[r] = col[r] + ((r == lane) ? bump : T(0));
      }
      ILQG_PH(10);
      // (Measured and dropped, round 5: the Gauss-Jordan form — every pivot clears its column in all other rows, no
      // back-substitution chain; same instruction count, headline -0.7 %, single-instance latency unchanged.)
      if (adaptive)
        lu_solve_columns<T, M>(col, lane, x);
      else
        qr_solve_columns<T, M>(col, lane, x);
      ILQG_PH(11);
      if constexpr (SPARE) {
        // [P | alpha] in one piece: column lane - M of the tile, alpha in column NX = JB (F = A - B [P | alpha] then
        // carries beta = -B alpha); alpha is read from there too (sAlr) — one divergent region and M stores less on the
        // solving wave's way to the barrier
        if (lane >= M && lane <= M + NX) {
#pragma unroll
          for (int r = 0; r < M; r++) sPt[r + LD * (lane - M)] = x[r];
        }
      } else {
        if (lane >= M && lane < M + NX) {
#pragma unroll
          for (int r = 0; r < M; r++) sPt[r + LD * (lane - M)] = x[r];
        } else if (lane == M + NX) {
#pragma unroll
          for (int r = 0; r < M; r++) sAl[r] = x[r];
        }
      }
    }
    ILQG_PH(2);
    lds_sync(false);  // (P, alpha) published
    ILQG_PH(8);
    // the strategies go to global memory from the last wave (wave 0 is the one the others wait for)
    if (w == NP - 1) {
      // explicit global address space: through a generic pointer these are FLAT stores, which count on the LDS counter
      // too, and the compiler then waits for every LDS read before the store that follows it (six round trips in a row)
      typedef __attribute__((address_space(1))) T gT;
      if (lane < NX) {
        T pv[M];
#pragma unroll
        for (int r = 0; r < M; r++) pv[r] = sPt[r + LD * lane];
        gT* dst = (gT*)(uniform_ptr(a.P + size_t(k) * M * NX)) + unsigned(M * lane);
#pragma unroll
        for (int r = 0; r < M; r++) dst[r] = pv[r];
      } else if (lane < NX + M) {
        ((gT*)(uniform_ptr(a.alpha + size_t(k) * M)))[unsigned(lane - NX)] = sAlr[lane - NX];
      }
    }

    // ---- F = A - B P (:189-194), beta = -B alpha; every wave needs them, so every wave computes them ----
    const vec Pd = ldD(sPt);
    vec nBT = ldDT(tB);  // B^T
#pragma unroll
    for (int r = 0; r < 4; r++) nBT[r] = -nBT[r];
    const vec Fraw = tile_xty_blocks<T, kblock_mask<T>(0, M)>(nBT, Pd, ldD(tA));  // rows >= M of -B^T are zero
    vec Fd, BetaD, zetaD;  // F proper; beta / zeta_w down the vector column, zero elsewhere
    if constexpr (SPARE) {
      // column JB of the padded A is zero and column JB of the P tile holds alpha: Fraw = [F | beta]
      constexpr bool kNoMasks = SOLVER;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        // kNoMasks: [F | beta] is used as it is on both sides of the products.  As a LEFT operand its column JB only
        // produces ROW JB of the result, and row JB of a Z_w tile never reaches anything: every right operand it meets
        // (B, [F | beta]) has a zero row JB (tile padding), and the recursion does not feed it back (the row is rebuilt
        // from this step's operands).  Column 15 of every operand is zero by construction, so it needs no mask either.
        Fd[r] = kNoMasks ? Fraw[r] : Fraw[r] * mCols;
        BetaD[r] = kNoMasks ? T(0) : Fraw[r] * mVecCol;
        zetaD[r] = Zd[r] * mVecCol;
      }
      if (want_fwd && w == 0 && j == JB) {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (row0 + RS * r < NX) a.scratch[size_t(k) * SCR + NP * (NX + 1) + row0 + RS * r] = Fraw[r];
      }
    } else {
      Fd = Fraw;
      T s = T(0);  // one entry of beta per lane (rows >= NX of the padded B are zero)
#pragma unroll
      for (int q = 0; q < M; q++) s -= tB[(lane & 15) + LD * q] * sAl[q];
      if (lane < 16) sBw[lane] = s;
      lds_sync(true);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        // kRowSums: this lane's rows of beta, unmasked, for the row-sum products below (BetaD stands in for them)
        BetaD[r] = kRowSums ? sBw[row0 + RS * r] : sBw[row0 + RS * r] * mVecCol;
        zetaD[r] = kRowSums ? T(0) : sZw[row0 + RS * r] * mVecCol;
      }
      if (want_fwd && w == 0 && lane < NX) a.scratch[size_t(k) * SCR + NP * (NX + 1) + lane] = sBw[lane];
    }
    if (want_fwd && w == 0) {
      if (lane < NP) {
        // alpha_i^T R_ii r_ii, evaluated (alpha^T R) r like Eigen (ilq_solver.cpp:384-386)
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
          for (int b = 0; b < MU; b++) aR += sAlr[lane * MU + b] * sR[ro_ii + b + MU * c];
          acc += aR * sr[rg_ii + c];
        }
        a.scratch[size_t(k) * SCR + NP * NX + lane] = acc;
      }
    }
    ILQG_PH(3);

    // ---- Z_w <- F^T Z_w F + Q_w + sum_jj P_jj^T R_w,jj P_jj  (:198-212), both layouts ----
    vec Cd = ldD(tQ);
    vec CTd = sym ? Cd : ldDT(tQ);
    if constexpr (SOLVER) {
      // Q_w l_w for ExpectedDecrease from the tile this wave has just loaded: Q_w is symmetric, so entry j is
      // sum_row Q_w[row][j] l_w[row] — every lane multiplies its four rows and the lane rows are summed on the vector
      // unit (rows_allreduce).  Six LDS reads per player and step instead of 2 NX, and no helper wave's time.
      T part = T(0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = row0 + RS * r;
        part += Cd[r] * (row < NX ? sl[w * NX + (row < NX ? row : 0)] : T(0));
      }
      part = rows_allreduce<T>(part);
      if (g == 0 && j < NX) a.scratch[size_t(k) * SCR + w * NX + j] = part;
    }
    if constexpr (SPARE) {
      // column JB of C_w: l_w + sum_jj P_jj^T (R_w,jj alpha_jj - r_w,jj)   (:198-201, 206-212), so that column JB of
      // Z_w' = F^T [..] + C_w is the new zeta_w.  The sum is [P | alpha]^T q with q = (R_w,jj alpha_jj - r_w,jj) stacked:
      // one product over the M rows of P with q in column JB of the right operand.  Every lane forms q for its own rows.
      vec Qy;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = row0 + RS * r;          // a row of P: player jj = row / MU, control aa = row % MU
        const bool in = row < M;
        const int rowc = in ? row : 0;
        const int jj = rowc / MU, aa = rowc % MU;
        int qw = -1, ro_wj = 0, rg_wj = 0;
#pragma unroll
        for (int e = 0; e < NP; e++)
#pragma unroll
          for (int f = 0; f < NP; f++)
            if (w == e && jj == f) {
              qw = pr.q[e][f];
              ro_wj = pr.ro[e][f];
              rg_wj = pr.rg[e][f];
            }
        T ww = -sr[rg_wj + aa];
#pragma unroll
        for (int b = 0; b < MU; b++) ww += sR[ro_wj + aa + MU * b] * sAlr[jj * MU + b];
        Qy[r] = (in && qw >= 0) ? ww * mVecCol : T(0);
        const int srow = row < NX ? row : 0;
        Cd[r] += (row < NX ? sl[w * NX + srow] : T(0)) * mVecCol;
      }
      vec Pm;  // P proper: column JB of the tile holds alpha
      if constexpr (SOLVER) {
        Pm = Pd;  // a left operand: its column JB only produces row JB of the result (see [F | beta] above)
      } else
#pragma unroll
      for (int r = 0; r < 4; r++) Pm[r] = Pd[r] * mCols;
      if constexpr (SOLVER) {
        // + sum_jj P_jj^T (R_w,jj P_jj) in the SAME product: H = blockdiag(R_w,jj) P has the rows of P and lives in the
        // state columns, q in column JB — one right operand [H | q], one chain over the M rows of P instead of one more
        // matrix instruction per (w, jj) block.  (Symmetric costs: no transposed copy to keep.)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int row = row0 + RS * r;
          const bool in = row < M;
          const int rowc = in ? row : 0;
          const int jj = rowc / MU, aa = rowc % MU;
          int qw = -1, ro_wj = 0;
#pragma unroll
          for (int e = 0; e < NP; e++)
#pragma unroll
            for (int f = 0; f < NP; f++)
              if (w == e && jj == f) {
                qw = pr.q[e][f];
                ro_wj = pr.ro[e][f];
              }
          T h = T(0);
#pragma unroll
          for (int b = 0; b < MU; b++) h += sR[ro_wj + aa + MU * b] * sPt[(jj * MU + b) + LD * j];
          Qy[r] += (in && qw >= 0) ? h * mCols : T(0);
        }
      }
      Cd = tile_xty_blocks<T, kblock_mask<T>(0, M)>(Pm, Qy, Cd);
    }
    if constexpr (!(SOLVER && SPARE))
    static_for<NP>([&](auto JJ) {  // jj is a compile-time constant: it selects the k blocks of the product
      constexpr int jj = decltype(JJ)::value;
      // + P_jj^T R_w,jj P_jj (and its transpose): H = R P_jj and H' = R^T P_jj sit in rows
      // jj*MU.. of a tile, P_jj likewise, so both products are one MFMA chain each.
      int qw = -1, ro_wj = 0;
#pragma unroll
      for (int e = 0; e < NP; e++) {
        qw = (w == e) ? pr.q[e][jj] : qw;
        ro_wj = (w == e) ? pr.ro[e][jj] : ro_wj;
      }
      if (qw < 0) return;  // wave-uniform
      const T* Rij = sR + ro_wj;
      T pb[MU];
#pragma unroll
      for (int b = 0; b < MU; b++) pb[b] = sPt[(jj * MU + b) + LD * j] * mCols;  // column NX of the tile holds alpha
      vec Pj, Hd, Htd;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int aa = row0 + RS * r - jj * MU;
        const bool in = aa >= 0 && aa < MU;
        const int ac = in ? aa : 0;
        const T mk = in ? T(1) : T(0);
        T h = T(0), ht = T(0);
#pragma unroll
        for (int b = 0; b < MU; b++) {
          h += Rij[ac + MU * b] * pb[b];
          ht += Rij[b + MU * ac] * pb[b];
        }
        Pj[r] = Pd[r] * (mk * mCols);
        Hd[r] = h * mk;
        Htd[r] = ht * mk;
      }
      // only rows jj*MU .. of P_jj are non-zero: one or two of the four k blocks
      Cd = tile_xty_blocks<T, kblock_mask<T>(jj * MU, jj * MU + MU)>(Pj, Hd, Cd);  // P_jj^T (R P_jj)
      if (!sym) CTd = tile_xty_blocks<T, kblock_mask<T>(jj * MU, jj * MU + MU)>(Pj, Htd, CTd);  // (P_jj^T R P_jj)^T
    });
    vec FT = zero4;
    if constexpr (SPARE) {
      if constexpr (SOLVER) {
        const vec Wd = tile_xty<T>(Yd, Fd, zero4);  // Z_w [F | beta]
        vec Wz;
#pragma unroll
        for (int r = 0; r < 4; r++) Wz[r] = Wd[r] + zetaD[r];  // column JB: zeta_w + Z_w beta
        Zd = tile_xty<T>(Fd, Wz, Cd);  // [Z_w' | F^T (zeta_w + Z_w beta)] + C_w; column 15 stays zero, row JB is never read
        Yd = Zd;
      } else
      {
      vec Fx;  // [F | beta]: F is zero in column JB, beta is zero outside it
#pragma unroll
      for (int r = 0; r < 4; r++) Fx[r] = Fd[r] + BetaD[r];
      const vec Wd = tile_xty<T>(Yd, Fx, zero4);  // Z_w [F | beta]
      vec Wm, Wz;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        Wm[r] = Wd[r] * mCols;
        Wz[r] = Wm[r] + (Wd[r] * mVecCol + zetaD[r]);  // column JB: zeta_w + Z_w beta (Wm is zero there)
      }
      if (!sym) Yd = tile_xty<T>(Wm, Fd, CTd);  // (Z_w F)^T F + C_w^T
      const vec Zx = tile_xty<T>(Fd, Wz, Cd);   // F^T [Z_w F | zeta_w + Z_w beta] + C_w
#pragma unroll
      for (int r = 0; r < 4; r++) Zd[r] = Zx[r] * mZcols;  // [Z_w' | F^T (zeta_w + Z_w beta)]
      if (sym) Yd = Zd;
      }
    } else if constexpr (kRowSums) {
      // n = 16: no spare tile column for the vector recursion, and a tile product per matrix-vector product (Z_w beta,
      // F^T t) is sixteen times the work: both are row sums over the accumulator-layout registers instead.
      //   (Z_w beta)[j] = sum_row Z_w^T[row][j] beta[row]  (Yd holds Z_w^T; Z_w itself when the costs are symmetric)
      //   (F^T t)[j]    = sum_row F[row][j] t[row],  t = zeta_w + Z_w beta
      // each 4 FMA per lane + rows_allreduce; t changes from "entry j on lane j" to "this lane's rows" through the
      // player's beta strip in LDS (beta's rows are in registers by now).
      const vec Wd = tile_xty<T>(Yd, Fd, zero4);  // Z_w F
      T zb = T(0);
#pragma unroll
      for (int r = 0; r < 4; r++) zb += Yd[r] * BetaD[r];
      zb = rows_allreduce<T>(zb);
      const T tj = sZw[j] + zb;  // zeta_w + Z_w beta, entry j (every lane row)
      if (!sym) Yd = tile_xty<T>(Wd, Fd, CTd);
      Zd = tile_xty<T>(Fd, Wd, Cd);
      if (sym) Yd = Zd;
      if (g == 0) sBw[j] = tj;
      lds_sync(true);
      T ft = T(0);
#pragma unroll
      for (int r = 0; r < 4; r++) ft += Fd[r] * sBw[row0 + RS * r];
      FT[0] = rows_allreduce<T>(ft);  // entry j of F^T (zeta_w + Z_w beta), every lane row
    } else {
      const vec Wd = tile_xty<T>(Yd, Fd, zero4);     // Z_w F
      const vec ZB = tile_xty<T>(Yd, BetaD, zero4);  // column w = Z_w beta
      vec TD;
#pragma unroll
      for (int r = 0; r < 4; r++) TD[r] = ZB[r] * mVecCol + zetaD[r];
      if (!sym) Yd = tile_xty<T>(Wd, Fd, CTd);
      Zd = tile_xty<T>(Fd, Wd, Cd);
      if (sym) Yd = Zd;
      FT = tile_xty<T>(Fd, TD, zero4);  // column w = F^T (zeta_w + Z_w beta)
    }
    ILQG_PH(4);
    if constexpr (!SPARE) {
    if constexpr (!kRowSums) {
    if (j == w) {  // the reads of the old zeta (zetaD) precede this by data dependence
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (row0 + RS * r < NX) sZw[row0 + RS * r] = FT[r];  // F^T (zeta_w + Z_w beta)
    }
    lds_sync(true);
    }
    if (lane < NX) {  // + l_w + sum_jj P_jj^T (R_w,jj alpha_jj - r_w,jj)   (:198-201, 206-212)
      T zn = (kRowSums ? FT[0] : sZw[lane]) + sl[w * NX + lane];
#pragma unroll
      for (int jj = 0; jj < NP; jj++) {
        int qw = -1, ro_wj = 0, rg_wj = 0;
#pragma unroll
        for (int e = 0; e < NP; e++) {
          qw = (w == e) ? pr.q[e][jj] : qw;
          ro_wj = (w == e) ? pr.ro[e][jj] : ro_wj;
          rg_wj = (w == e) ? pr.rg[e][jj] : rg_wj;
        }
        if (qw < 0) continue;
        const T* Rij = sR + ro_wj;
        const T* rij = sr + rg_wj;
        T add = T(0);
#pragma unroll
        for (int aa = 0; aa < MU; aa++) {
          T ww = T(0);
#pragma unroll
          for (int b = 0; b < MU; b++) ww += Rij[aa + MU * b] * sAlr[jj * MU + b];
          add += sPt[(jj * MU + aa) + LD * lane] * (ww - rij[aa]);
        }
        zn += add;
      }
      sZw[lane] = zn;
    }
    }
    ILQG_PH(5);
    if (!cmp) {
      dma_wait();
      ILQG_PH(9);
      lds_sync(false);  // next image complete (the DMA fills it directly); everyone is done with P / alpha / this image
    }
    // Compact rows: no barrier here.  The next image was completed by LDS stores (the scatter) in front of this step's
    // second barrier; P / alpha, the [S | Y] bounce and this image are next written behind the NEXT step's first
    // barrier, which every wave reaches only after its last read of them; the staged compact row is waited for in front
    // of that barrier by the wave that requested it.
    cur = 1 - cur;
    set_img(cur);
    ILQG_PH(6);
  }
#undef ILQG_PH
  if (a.prio_div > 0) __builtin_amdgcn_s_setprio(0);
  tl_stamp(a.tl, a.tl_b, 18, t == 0);
  if (kProfile && a.ph && lane == 0) {  // wave w's row of the profile: a.ph[16 * w + i]
#pragma unroll
    for (int i = 0; i < 16; i++) a.ph[16 * w + i] += phacc[i];
  }

  if constexpr (!SOLVER) {
    if (want_fwd && !a.defer_forward) {
      __syncthreads();  // scratch rows written by all waves
      if (w == 0) lq_forward_pass_body<T, NX, NP, MU, 64, W::LDS_ELEMS>(a, sm, lane);
    }
  }
}

// Threads per instance of the feedback sweep.
template <typename T, int NX, int NP, int MU, bool FORCE_VALU = false>
struct LQFeedbackThreads {
  static constexpr bool PLAYER_WAVES = LQCfg<T, NX, NP, MU>::USE_MFMA && !FORCE_VALU;
  static constexpr int NT = PLAYER_WAVES ? 64 * NP : LQCfg<T, NX, NP, MU>::NT;
};

template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void lq_feedback_instance_mfma_pw2(const LQArgs<T>& a, const PairTable& pt, T* sm);  // ilqg_lq_feedback2.hpp

// The matrix-core sweep of this shape (LQCfg::USE_MFMA).
template <typename T, int NX, int NP, int MU>
__device__ __forceinline__ void lq_feedback_instance_mfma(const LQArgs<T>& a, const PairTable& pt, T* sm) {
  if constexpr (LQCfg<T, NX, NP, MU>::MFMA_ONE_TILE)
    lq_feedback_instance_mfma_pw<T, NX, NP, MU>(a, pt, sm);
  else
    lq_feedback_instance_mfma_pw2<T, NX, NP, MU>(a, pt, sm);
}

template <typename T, int NX, int NP, int MU, bool FORCE_VALU = false>
__device__ __forceinline__ void lq_feedback_dispatch(const LQArgs<T>& a, const PairTable& pt, T* sm) {
  if constexpr (LQCfg<T, NX, NP, MU>::USE_MFMA && !FORCE_VALU) {
    lq_feedback_instance_mfma<T, NX, NP, MU>(a, pt, sm);
  } else {
    lq_feedback_instance<T, NX, NP, MU>(a, pt, sm);
  }
}

}  // namespace ilqg
