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
#include "ilqg_pairs.hpp"  // kMaxPlayers, kMaxPairs, PairTable

#include <type_traits>
#include <utility>

namespace ilqg {

constexpr int kMaxT = 256;
// Phase-profile instrumentation (scripts/stage_bench.py) is compiled in only with -DILQG_PROFILE=1: the
// accumulators are live across the hot loops and the kernels are register-bound.
#ifndef ILQG_PROFILE
#define ILQG_PROFILE 0
#endif
constexpr bool kProfile = ILQG_PROFILE != 0;
// Timeline stamps (-DILQG_TIMELINE=1, scripts/timeline.py): a handful of wall-clock stamps per instance and launch, each
// stored at once by one lane (nothing is kept live across the hot loops, unlike the phase profile): slot i of instance
// b at prof[b * 96 + 32 + i], in 10 ns units of the constant-rate counter.
#ifndef ILQG_TIMELINE
#define ILQG_TIMELINE 0
#endif
constexpr bool kTimeline = ILQG_TIMELINE != 0;
__device__ __forceinline__ void tl_stamp(long long* prof, int b, int slot, bool who) {
  if constexpr (kTimeline) {
    if (prof && who) prof[size_t(b) * 96 + 32 + slot] = wall_clock64();
  }
}

struct DevTerm {
  int kind, role, player, arg;
  int idx[4];
  float weight, value;
  int flags, polyline, child_begin, child_count, slot;
  // filled by ilqg_problem_create:
  int arg_off;   // offset of the argument vector inside a row's [x | u]
  int arg_dim;   // its length
  int k_start;   // FinalTimeCost: first time step the term is active at (0 = always)
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
  const int* poly_off;      // [num_polylines+1] offsets in points (segment s of polyline q is poly_off[q]-q+s)
  const float* poly_pts;
  // LineSegment2 objects precomputed on the host in both precisions, 21 scalars per segment:
  // [p1x p1y p2x p2y len ux uy | shortcut(prev.p1 -> p2) | shortcut(p1 -> next.p2)]
  // (include/ilqgames/geometry/line_segment2.h:55-62; shortcuts: src/polyline2.cpp:126-133)
  const float* segs_f;
  const double* segs_d;
  int total_segs;
  // Per-step nominals of the time-dependent costs (NominalPathLengthCost: t_k * speed; RouteProgressCost: the route
  // point at pos0 + t_k * speed), [table][T][2] doubles, one copy per geometry precision; DevTerm::polyline of such a
  // term is its table.  Tabulated by ilqg_problem_create.
  const double* time_nominal_f;
  const double* time_nominal_d;
  // Coefficient blocks of the affine constraints, one copy per precision; DevTerm::polyline of such a term is its block's
  // offset.  AffineScalarConstraint: [a (d) | b]; AffineVectorConstraint: [A (d x d, column-major) | b (d) | A^T A | A A^T]
  // (the two products as its constructor forms them, affine_vector_constraint.h:60-61).  Built by ilqg_problem_create.
  const float* dense_f;
  const double* dense_d;
  // TotalCosts summation order: per player [count, term indices...] (state costs, then control costs)
  const int* cost_order;
  int cost_order_stride;
  int num_constraints;
  // MultiPlayerIntegrableSystem::DistanceBetween as Problem::SyncToExistingProblem uses it (src/problem.cpp:105-110):
  // squared distance over the first sync_dist_dims entries of the state — the position of the first subsystem for
  // the car / unicycle / point-mass models and TwoPlayerUnicycle4D (their overrides: two_player_unicycle_4d.h:141-147)
  // and Air3D, the first subsystem's whole state where the model inherits the default (SinglePlayerDubinsCar only)
  int sync_dist_dims;
  PairTable pairs;
  // Row program of the lane-per-time-step quadraticisation stage (ilqg_rows.hpp; built by build_row_program)
  const int* row_prog;
  int row_prog_words;
  int rp_pslots, rp_lslots;  // persistent / most pass-local slots: sizes the stage's LDS
  int rp_gslots;             // most gradient slots of a pass (numbered first: a merit-only evaluation keeps only these)
  int rp_maps_off, rp_maps_words;  // the program's word -> slot maps (copied into LDS by every workgroup)
  int rp_compact_off, rp_compact_w;  // compact rows (ilqg_rows.hpp): the block's offset in row_prog, words per row (0: none)
};

// arrays of a time step's image, as the row program's regions and the compact rows name them
enum { RA_A = 0, RA_B = 1, RA_Q = 2, RA_L = 3, RA_R = 4, RA_r = 5 };
// Compact rows (the solve's own interchange between the row stage, ilqg_rows.hpp, and the one-tile sweep + forward pass,
// ilqg_lq.hpp): of a time step's [A | B | Q_i | l_i | R_ij | r_ij] only the words a Jacobian or a cost term can touch vary
// — the pass-local slots of the row program's passes — so the stage writes those, pass after pass ("compact row" of
// RC_W words), and the consumers scatter them over a constant background in their LDS images.  Block at RP_OFF_COMPACT:
// [RC_W, RC_NBG, base of the Jacobian pass, of player 0 .. N-1 | destination of each word | (destination, kind, value)
// of each non-zero constant].  A destination is array << 24 | offset inside the array's row; a constant's kind is
// RC_LITERAL (the float `value`), RC_DT or RC_NEG_DT (the time step, in the problem's precision).
enum { RC_W = 0, RC_NBG = 1, RC_BASE = 2 };
enum { RC_LITERAL = 0, RC_DT = 1, RC_NEG_DT = 2, RC_BG_WORDS = 3 };
constexpr int kCompactMaxWords = 256;  // four words per lane of a scattering wave (roundabout, n = 24, N = 4: 192)
constexpr int kCompactMaxBg = 128;     // non-zero constants an LDS copy of the background list holds (open-loop sweep)

constexpr int kSegStride = 21;

template <typename T> __device__ __forceinline__ const T* problem_segs(const DevProblem& p);
template <> __device__ __forceinline__ const float* problem_segs<float>(const DevProblem& p) { return p.segs_f; }
template <> __device__ __forceinline__ const double* problem_segs<double>(const DevProblem& p) { return p.segs_d; }
template <typename T> __device__ __forceinline__ const T* problem_dense(const DevProblem& p);
template <> __device__ __forceinline__ const float* problem_dense<float>(const DevProblem& p) { return p.dense_f; }
template <> __device__ __forceinline__ const double* problem_dense<double>(const DevProblem& p) { return p.dense_d; }
template <typename T> __device__ __forceinline__ const double* problem_time_nominal(const DevProblem& p);
template <> __device__ __forceinline__ const double* problem_time_nominal<float>(const DevProblem& p) { return p.time_nominal_f; }
template <> __device__ __forceinline__ const double* problem_time_nominal<double>(const DevProblem& p) { return p.time_nominal_d; }

// user priority 0..3 of the calling wave (s_setprio takes an immediate)
__device__ __forceinline__ void set_wave_prio(int pr) {
  pr &= 3;
  if (pr == 0) __builtin_amdgcn_s_setprio(0);
  else if (pr == 1) __builtin_amdgcn_s_setprio(1);
  else if (pr == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}

template <typename T>
__device__ __forceinline__ T sgn(T x) {
  return T((T(0) < x) - (x < T(0)));
}

// fma in the operands' type (__builtin_fma on floats is the DOUBLE operation behind two conversions)
__host__ __device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__host__ __device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T>
__device__ __forceinline__ T shfl(T v, int lane) {
  return __shfl(v, lane, 64);
}

// LDS-only synchronisation inside a stage.  A __syncthreads() also drains the vector-memory
// queue (s_waitcnt vmcnt(0)), which stalls every step on the prefetch loads and the output stores
// still in flight (~1-2k cycles each, measured).  Within one wavefront the LDS executes a wave's
// DS operations in issue order, so a single-wave workgroup only needs the compiler not to reorder
// them; multi-wave workgroups use a raw s_barrier behind an lgkmcnt-only wait.  Global-memory
// hand-offs between lanes (stage boundaries) still use __syncthreads().
__device__ __forceinline__ void lds_sync(bool single_wave) {
  if (single_wave) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
}

// ---- global -> LDS DMA (global_load_lds_*): no VGPR round trip, completes in the background ----
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

// Copies `nbytes` (a multiple of 4) from global `g` to LDS `l` with the workgroup's NT threads.
// The LDS destination of one instruction is wave-uniform base + lane * width, i.e. the LDS image is
// the global image.  WIDE = 16-byte pieces (both addresses 16-byte aligned), else 4-byte pieces.
// A wave-uniform pointer, pinned to scalar registers.  The DMA sources below are "uniform base + 32-bit lane offset";
// left to itself the compiler folds the lane offset into a loop-invariant 64-bit per-lane address per stream and keeps
// it in VGPRs across the step loops (two registers each — in the sweeps they were what spilled).
template <class P>
__device__ __forceinline__ P* uniform_ptr(P* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return reinterpret_cast<P*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

template <int NT, bool WIDE>
__device__ __forceinline__ void dma_g2l(const void* g_, void* l, int nbytes, int t) {
  const void* g = uniform_ptr(g_);
  constexpr int BPL = WIDE ? 16 : 4;
  const int wbase = (t & ~63) * BPL;  // this wave's slice of each NT*BPL chunk
  const int lane = t & 63;
  for (int off = 0; off < nbytes; off += NT * BPL) {
    const int my = off + wbase + lane * BPL;
    if (my < nbytes) {
      if constexpr (WIDE)
        __builtin_amdgcn_global_load_lds((glb_vptr)((const char*)g + unsigned(my)), (lds_vptr)((char*)l + off + wbase), 16, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((glb_vptr)((const char*)g + unsigned(my)), (lds_vptr)((char*)l + off + wbase), 4, 0, 0);
    }
  }
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}).
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <typename T>
__device__ __forceinline__ T dinf() {
  return T(__builtin_huge_val());
}

}  // namespace ilqg
