// ilqg_pairs.hpp — the (i, j) control-block table of one problem, free of HIP dependencies so that the
// run-time-dimensioned sweeps (ilqg_lq_generic.hpp) can also be compiled by a host compiler for their logic test.
#pragma once

#include "../../include/ilqg.h"

namespace ilqg {

constexpr int kMaxPlayers = ILQG_MAX_PLAYERS;
constexpr int kMaxPairs = 16;  // device kernels: at most 16 control blocks (QuadraticCostApproximation::control keys)

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

}  // namespace ilqg
