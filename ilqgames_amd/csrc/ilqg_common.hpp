// ilqg_common.hpp — device-side descriptors shared by every gfx950 kernel of libilqg_hip.so.
//
// Execution model used throughout: ONE WORKGROUP OWNS ONE GAME INSTANCE.  The
// workgroup is ceil(N*n/64) wavefronts (one for every config up to N*n = 64);
// lane t = i*n + c of the LQ sweep owns column c of player i's value matrix
// Z_i in registers, the per-step blocks (A, B, Q_i, l_i, R_ij, r_ij) are staged
// through LDS, and the instance's trajectory-major arrays stream through HBM
// exactly once per stage.  Every stage is a __device__ function so the same
// code runs as a standalone kernel (the C-ABI stage entry points) and inside
// the persistent per-instance iLQ kernel.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ilqg.h"

namespace ilqg {

constexpr int kMaxPlayers = ILQG_MAX_PLAYERS;
constexpr int kMaxPairs = 16;  // device kernels: at most NP*NP <= 16 control blocks
constexpr int kMaxT = 256;

// (i,j) control-block table of one problem (QuadraticCostApproximation::control keys).
struct PairTable {
  int npairs;
  int pi[kMaxPairs], pj[kMaxPairs];
  int roff[kMaxPairs];   // offset of block p inside an R row (elements)
  int rgoff[kMaxPairs];  // offset inside an r row
  int from_cost[kMaxPairs];
  int pii[kMaxPlayers];  // index of the (i,i) block
  int Rsz, rsz;
};

struct DevTerm {
  int kind, role, player, arg;
  int idx[4];
  float weight, value;
  int flags, polyline, child_begin, child_count, slot;
  // filled by ilqg_problem_create for the term-parallel quadraticisation stage:
  int round;     // scatter round: terms of one round touch disjoint tile entries, and a term's
                 // round is later than that of every earlier term sharing an entry, so each
                 // Hessian/gradient entry is accumulated in the reference's order
  int tile_h;    // offset of the Hessian tile inside the [Q|l|R|r] image
  int tile_g;    // offset of the gradient vector inside the same image
  int ld;        // leading dimension of the Hessian tile
  int arg_off;   // offset of the argument vector inside the [x|u] image
  int arg_dim;   // its length
};

// Flattened Problem (dynamics + PlayerCosts) living in kernel-argument space;
// `terms`, `poly_off`, `poly_pts` point to small device tables.
struct DevProblem {
  int N, n, m, T;
  double dt;
  int sub_kind[kMaxPlayers], xoff[kMaxPlayers + 1], uoff[kMaxPlayers + 1], udim[kMaxPlayers];
  float sub_param[kMaxPlayers];
  float state_reg[kMaxPlayers], control_reg[kMaxPlayers];
  int structure[kMaxPlayers];
  int num_terms;
  const DevTerm* terms;
  int num_polylines;
  const int* poly_off;
  const float* poly_pts;
  int num_constraints;
  int num_rounds;
  PairTable pairs;
};

template <typename T>
__device__ __forceinline__ T sgn(T x) {
  return T((T(0) < x) - (x < T(0)));
}

template <typename T>
__device__ __forceinline__ T shfl(T v, int lane) {
  return __shfl(v, lane, 64);
}

template <typename T>
__device__ __forceinline__ T dinf() {
  return T(__builtin_huge_val());
}

}  // namespace ilqg
