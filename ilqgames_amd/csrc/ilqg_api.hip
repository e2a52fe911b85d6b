// ilqg_api.hip — kernels and the C ABI of libilqg_hip.so (gfx950 only).
// Entry points and the reference methods they replace: include/ilqg.h.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "ilqg_common.hpp"
#include "ilqg_lq.hpp"
#include "ilqg_lq_openloop.hpp"
#include "ilqg_lq_feedback2.hpp"
#include "ilqg_lq_generic.hpp"
#include "ilqg_costates.hpp"
#include "ilqg_models.hpp"
#include "ilqg_nash.hpp"
#include "ilqg_receding.hpp"
#include "ilqg_rowprog.hpp"
#include "ilqg_rowprog_static.hpp"
#include "ilqg_solve.hpp"
#include "ilqg_stages.hpp"

using namespace ilqg;

// The library is built from several translation units of this one file (__graft_entry__.build): one "main" unit
// with the C ABI and every kernel that does not depend on the (n, N, m_i) instantiation, and one unit per entry of
// ILQG_FOR_DIMS (compiled with -DILQG_PART_NX/NP/MU) that holds DimsLaunch<T, n, N, m_i> and the kernels it
// launches — they compile in parallel.  Without those macros the file is a single self-contained unit.
// This is the state the units share.
namespace ilqg_shared __attribute__((visibility("hidden"))) {
// g_prof: the diagnostic builds' buffer (-DILQG_PROFILE=1 / -DILQG_TIMELINE=1: ilqg_debug_set_profile_buffer exists only
// there); the product library has neither the symbol nor the global.
#define ILQG_DIAGNOSTIC_BUILD (ILQG_PROFILE || ILQG_TIMELINE)
#if defined(ILQG_PART_NX)
extern thread_local std::string g_err;
#if ILQG_DIAGNOSTIC_BUILD
extern long long* g_prof;
#endif
#else
thread_local std::string g_err;
#if ILQG_DIAGNOSTIC_BUILD
long long* g_prof = nullptr;  // set through ilqg_debug_set_profile_buffer
#endif
#endif
}  // namespace ilqg_shared
using ilqg_shared::g_err;
#if ILQG_DIAGNOSTIC_BUILD
using ilqg_shared::g_prof;
#endif

namespace {

ilqg_status fail(ilqg_status s, const std::string& msg) {
  g_err = msg;
  return s;
}
#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(ILQG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

}  // namespace

// Device scratch of the stand-alone entry points that take no workspace argument (ilqg_lq_*_batch with delta_x /
// costates, ilqg_total_costs_batch, the equilibrium checks).  The caller may hand the library a buffer of its own
// (ilqg_set_scratch): then nothing is allocated here and a call that needs more fails with the size it needs.  Without
// one the library keeps a grow-only allocation per calling thread.  (The solves never use it: ilqg_workspace_bytes.)
namespace ilqg_shared __attribute__((visibility("hidden"))) {
struct ScratchState {
  void* ptr = nullptr;
  size_t bytes = 0;
  bool caller_owned = false;
};
// One per calling thread, shared by the library's translation units — through an accessor: an `extern thread_local` of a
// constant-initialised class type makes the other units call a TLS init function that the defining unit never emits.
ScratchState& scratch_state();
#if !defined(ILQG_PART_NX)
ScratchState& scratch_state() {
  static thread_local ScratchState st;
  return st;
}
#endif
}  // namespace ilqg_shared

namespace {
struct Scratch {  // this unit's view of the shared state
  ilqg_status reserve(size_t need) {
    ilqg_shared::ScratchState& st = ilqg_shared::scratch_state();
    if (need <= st.bytes) return ILQG_OK;
    if (st.caller_owned)
      return fail(ILQG_ERR_INVALID, "the scratch buffer given to ilqg_set_scratch is too small: this call needs " +
                                        std::to_string(need) + " bytes");
    if (st.ptr) (void)hipFree(st.ptr);
    st.ptr = nullptr;
    st.bytes = 0;
    HIP_TRY(hipMalloc(&st.ptr, need));
    st.bytes = need;
    return ILQG_OK;
  }
};
}  // namespace

namespace {


// ------------------------------------------------------------------------------------
// Kernels — one workgroup per game instance
// ------------------------------------------------------------------------------------
template <typename T>
struct LQBatchArgs {
  const T *A, *Bm, *Q, *l, *R, *r, *x0;
  T *P, *alpha, *dx, *scratch;
  int T_steps, adaptive, batch, force_valu;
  T* costates = nullptr;
};

template <typename T, int NX, int NP, int MU, bool FORCE_VALU>
__global__ void __launch_bounds__((LQFeedbackThreads<T, NX, NP, MU, FORCE_VALU>::NT),
                                  (LQFeedbackThreads<T, NX, NP, MU, FORCE_VALU>::PLAYER_WAVES ? (NX <= 16 ? NP : 2) : 1))
lq_feedback_kernel(LQBatchArgs<T> g, PairTable pt) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const int b = blockIdx.x;
  constexpr int M = NP * MU;
  const size_t Tn = g.T_steps;
  LQArgs<T> a;
  a.A = g.A + b * Tn * NX * NX;
  a.Bm = g.Bm + b * Tn * NX * M;
  a.Q = g.Q + b * Tn * NP * NX * NX;
  a.l = g.l + b * Tn * NP * NX;
  a.R = g.R + b * Tn * pt.Rsz;
  a.r = g.r + b * Tn * pt.rsz;
  a.x0 = g.x0 ? g.x0 + size_t(b) * NX : nullptr;
  a.P = g.P + b * Tn * M * NX;
  a.alpha = g.alpha + b * Tn * M;
  a.dx = g.dx ? g.dx + b * Tn * NX : nullptr;
  a.scratch = g.scratch ? g.scratch + b * Tn * (NP * (NX + 1) + NX) : nullptr;
  a.ed_out = nullptr;
  a.T_steps = g.T_steps;
  a.adaptive = g.adaptive;
  lq_feedback_dispatch<T, NX, NP, MU, FORCE_VALU>(a, pt, sm);
}

template <typename T, int NX, int NP, int MU>
__global__ void __launch_bounds__((OLCfg<T, NX, NP, MU>::NT), 3)  // three instances per CU (see ilqg_lq_openloop.hpp)
lq_openloop_kernel(LQBatchArgs<T> g, PairTable pt) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const int b = blockIdx.x;
  constexpr int M = NP * MU;
  const size_t Tn = g.T_steps;
  LQArgs<T> a;
  a.A = g.A + b * Tn * NX * NX;
  a.Bm = g.Bm + b * Tn * NX * M;
  a.Q = g.Q + b * Tn * NP * NX * NX;
  a.l = g.l + b * Tn * NP * NX;
  a.R = g.R + b * Tn * pt.Rsz;
  a.r = g.r + b * Tn * pt.rsz;
  a.x0 = g.x0 ? g.x0 + size_t(b) * NX : nullptr;
  a.P = g.P + b * Tn * M * NX;
  a.alpha = g.alpha + b * Tn * M;
  a.dx = g.dx ? g.dx + b * Tn * NX : nullptr;
  a.scratch = g.scratch + b * Tn * (g.costates ? OLCfg<T, NX, NP, MU>::ROW_FAT : OLCfg<T, NX, NP, MU>::ROW);
  a.costates = g.costates ? g.costates + b * Tn * NP * NX : nullptr;
  a.ed_out = nullptr;
  a.T_steps = g.T_steps;
  a.adaptive = 0;
  lq_openloop_instance<T, NX, NP, MU>(a, pt, sm);
}

// The sweeps with run-time dimensions (ilqg_lq_generic.hpp): whatever has no specialised instantiation above.
template <typename T>
__global__ void __launch_bounds__(256) lq_generic_kernel(LQBatchArgs<T> g, GenDims d, PairTable pt, int open_loop,
                                                         int scratch_row) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const size_t b = blockIdx.x, Tn = d.T, n = d.n, m = d.m, N = d.N;
  GenLQArgs<T> a;
  a.A = g.A + b * Tn * n * n;
  a.Bm = g.Bm + b * Tn * n * m;
  a.Q = g.Q + b * Tn * N * n * n;
  a.l = g.l + b * Tn * N * n;
  a.R = g.R + b * Tn * pt.Rsz;
  a.r = g.r + b * Tn * pt.rsz;
  a.x0 = g.x0 ? g.x0 + b * n : nullptr;
  a.P = g.P + b * Tn * m * n;
  a.alpha = g.alpha + b * Tn * m;
  a.dx = g.dx ? g.dx + b * Tn * n : nullptr;
  a.costates = (open_loop && g.costates) ? g.costates + b * Tn * N * n : nullptr;
  a.scratch = g.scratch ? g.scratch + b * Tn * size_t(scratch_row) : nullptr;
  a.ed_out = nullptr;
  a.adaptive = g.adaptive;
  const ParDevice par;
  if (open_loop)
    lq_openloop_generic<T>(d, a, pt, sm, par);
  else
    lq_feedback_generic<T>(d, a, pt, sm, par);
}

template <typename T>
struct RolloutBatchArgs {
  const T *x0, *xs_ref, *us_ref, *P, *alpha, *alpha_scale;
  T *xs, *us;
  const int* active;
};

template <typename T>
__global__ void rollout_kernel(DevProblem p, RolloutBatchArgs<T> g) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const size_t b = blockIdx.x;
  if (g.active && !g.active[b]) return;
  const size_t Tn = p.T, n = p.n, m = p.m;
  RolloutArgs<T> a{g.x0 + b * n,          g.xs_ref + b * Tn * n, g.us_ref + b * Tn * m, g.P + b * Tn * m * n,
                   g.alpha + b * Tn * m,  g.alpha_scale ? g.alpha_scale[b] : T(1),
                   g.xs + b * Tn * n,     g.us + b * Tn * m};
  rollout_instance_rt<T>(p, a, sm, threadIdx.x);
}

template <typename T>
struct QuadBatchArgs {
  const T *xs, *us, *lambdas, *mu;
  const int* t_extreme;
  T *A, *Bm, *Q, *l, *R, *r, *merit_part, *cost_part;
  const int* active;
};

template <typename T>
__device__ __forceinline__ QuadArgs<T> quad_args_of(const DevProblem& p, const QuadBatchArgs<T>& g, size_t b) {
  const size_t Tn = p.T, n = p.n, m = p.m, N = p.N;
  QuadArgs<T> a;
  a.xs = g.xs + b * Tn * n;
  a.us = g.us + b * Tn * m;
  a.lambdas = g.lambdas ? g.lambdas + b * p.num_constraints * Tn : nullptr;
  a.mu = g.mu ? g.mu[b] : T(10);
  a.t_extreme = g.t_extreme ? g.t_extreme + b * N : nullptr;
  a.t_init = 0.0;
  a.A = g.A ? g.A + b * Tn * n * n : nullptr;
  a.Bm = g.Bm ? g.Bm + b * Tn * n * m : nullptr;
  a.Q = g.Q ? g.Q + b * Tn * N * n * n : nullptr;
  a.l = g.l ? g.l + b * Tn * N * n : nullptr;
  a.R = g.R ? g.R + b * Tn * p.pairs.Rsz : nullptr;
  a.r = g.r ? g.r + b * Tn * p.pairs.rsz : nullptr;
  a.merit_part = g.merit_part ? g.merit_part + b * Tn * N * 2 : nullptr;
  a.cost_part = g.cost_part ? g.cost_part + b * Tn * N : nullptr;
  return a;
}

// Linearise / quadraticise / cost stage on its own (ilqg_rows.hpp): grid = (ceil(T / cw), B), one wavefront takes cw
// consecutive rows of one instance.
template <typename T, int NX, int NP, int MU>
__global__ void __launch_bounds__(64) rows_kernel(DevProblem p, QuadBatchArgs<T> g, int cw) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const size_t b = blockIdx.y;
  if (g.active && !g.active[b]) return;
  const short* maps = rows_maps_load(p, smem_raw);
  T* sm = reinterpret_cast<T*>(smem_raw + rows_maps_bytes(p));
  const QuadArgs<T> a = quad_args_of<T>(p, g, b);
  const int k0 = int(blockIdx.x) * cw;
  const int nrows = p.T - k0 < cw ? p.T - k0 : cw;
  rows_chunk<T, NX, NP * MU, NP>(p, maps, a, k0, nrows, cw, sm, int(threadIdx.x));
}

// ilqg_solve_state_batch: the loop state of every instance out of the workspace
template <typename T>
__global__ void solve_state_kernel(const T* ws, size_t ws_stride, size_t state_off, int batch, T* last_merit,
                                   T* expected_decrease, T* step, int* backtracks) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const SolveState<T>* s = reinterpret_cast<const SolveState<T>*>(ws + size_t(b) * ws_stride + state_off);
  if (last_merit) last_merit[b] = s->last_merit;
  if (expected_decrease) expected_decrease[b] = s->expected_decrease;
  if (step) step[b] = s->acc_scale;
  if (backtracks) backtracks[b] = s->rejected;
}

// What follows the per-instance blocks in a solve's workspace: two lists of instance ids (this round's back-tracking
// instances, the next round's) and the pool of the speculative line search.
struct WsTail {
  size_t ids_off, pool_off, total;
  int pool_entries;
};
inline WsTail ws_tail(const DevProblem& d, int batch, size_t elem, int ol_row) {
  auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
  const WsLayout L(d.n, d.m, d.N, d.T, d.pairs.Rsz, d.pairs.rsz, ol_row, d.num_constraints, 1);
  WsTail t;
  t.ids_off = up(size_t(L.total) * elem * size_t(batch));
  t.pool_off = up(t.ids_off + size_t(2) * batch * sizeof(int));
  // sixteen candidates per instance up to kProbeEntries; a small batch gets every candidate of every instance up to
  // one round's budget (a few dozen instances in failing line searches walk through all their step sizes, 128 a round
  // each: config 5's 47 surviving plans as a dense batch, 752 entries: 75 ms per replan; 6016: see DESIGN.md 3.13),
  // and never fewer than two full rounds' worth for a single instance
  const long long small = std::min<long long>((long long)batch * kProbeCandidates, kProbeRoundBudget);
  const long long want = std::max<long long>((long long)batch * 16, small);
  t.pool_entries = int(std::min<long long>(want, kProbeEntries));
  if (t.pool_entries < 2 * kProbeCandidates) t.pool_entries = 2 * kProbeCandidates;
  t.total = up(t.pool_off + size_t(t.pool_entries) * ProbeEntry(d.n, d.m, d.N, d.T).total * elem);
  return t;
}

template <typename T>
__global__ void costs_reduce_kernel(DevProblem p, const T* cost_part, T* costs, int* t_extreme, const int* active) {
  const size_t b = blockIdx.x;
  if (active && !active[b]) return;
  costs_reduce<T>(p, cost_part + b * p.T * p.N, costs + b * p.N, t_extreme ? t_extreme + b * p.N : nullptr);
}

// Wavefronts per instance in the trial kernel: wave 0 integrates, the others take the chunks of rows it has
// produced (ilqg_solve.hpp).  One row wave keeps up with the integration; each more costs a row scratch of LDS.
#ifndef ILQG_TRIAL_WAVES_F32
#define ILQG_TRIAL_WAVES_F32 2
#endif
#ifndef ILQG_TRIAL_WAVES_F64
#define ILQG_TRIAL_WAVES_F64 2
#endif
template <typename T>
struct TrialWaves {
  static constexpr int W = sizeof(T) == 4 ? ILQG_TRIAL_WAVES_F32 : ILQG_TRIAL_WAVES_F64;
};

// Trial kernel: rollout + linearise/quadraticise + line-search decision (ilqg_solve.hpp).  The second
// launch-bound argument is the number of waves per SIMD the register allocation must allow.  fp32 was compiled for
// three (168 registers: six instances per CU) until round 5; since the large batches of the one-tile shapes run the
// split pass, the fused kernel serves four instances per CU at most there, and the full register file is worth 1.6 %
// at the headline batch in fp32 (2.105 -> 2.140 M it/s).  -DILQG_TRIAL_OCCUPANCY_F32=3 restores the old build.
#ifndef ILQG_TRIAL_OCCUPANCY_F32
#define ILQG_TRIAL_OCCUPANCY_F32 2
#endif
template <typename T, int NX, int NP, int MU, int W, int PROGID = 0>
__global__ void __launch_bounds__(64 * W, (sizeof(T) == 4 && W == 2) ? ILQG_TRIAL_OCCUPANCY_F32 : W) ilq_trial_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x;
  if (!sa.first) {  // instances that are done (or waiting for the LQ kernel) leave without touching LDS
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    const int stage = reinterpret_cast<const SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage;
    if (stage != ST_ROLLOUT && stage != ST_QUAD) return;
  } else if (sa.active && !sa.active[b]) {  // skipped instance: done before it starts, outputs untouched
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    if (threadIdx.x == 0)
      reinterpret_cast<SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage = ST_DONE;
    return;
  }
  const short* maps = rows_maps_load(p, smem_raw);
  T* sm = reinterpret_cast<T*>(smem_raw + rows_maps_bytes(p));
  trial_part_instance<T, NX, NP, MU, W, TRIAL_FUSED, PROGID>(p, maps, sa, b, sm);
}

// The same pass cut into three launches (ilqg_solve.hpp, TRIAL_ROLL / rows_part_instance / TRIAL_DECIDE), for
// problems whose fused trial kernel fits fewer than three instances on a CU.
#ifndef ILQG_SPLIT_ROW_EXTRA_CHUNKS
#define ILQG_SPLIT_ROW_EXTRA_CHUNKS 0
#endif
#ifndef ILQG_ROLL_WAVES
#define ILQG_ROLL_WAVES 4
#endif
template <typename T, int NX, int NP, int MU>
__global__ void __launch_bounds__(64, sizeof(T) == 8 ? ILQG_ROLL_WAVES : 1) ilq_roll_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  if constexpr (rollout_pairs(NX, NP, MU)) {  // two instances per wavefront: entries 2 x and 2 x + 1 of the round
    const int i0 = 2 * int(blockIdx.x), i1 = i0 + 1;
    const int b0 = sa.ids ? sa.ids[i0] : i0;
    const int b1 = i1 < sa.round_count ? (sa.ids ? sa.ids[i1] : i1) : -1;
    roll_pair_instances<T, NX, NP, MU>(p, sa, b0, b1, reinterpret_cast<T*>(smem_raw));
    return;
  }
  const int b = sa.ids ? sa.ids[blockIdx.x] : int(blockIdx.x);
  if (!sa.first) {
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    const int stage = reinterpret_cast<const SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage;
    if (stage != ST_ROLLOUT) return;
  } else if (sa.active && !sa.active[b]) {
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    if (threadIdx.x == 0)
      reinterpret_cast<SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage = ST_DONE;
    return;
  }
  trial_part_instance<T, NX, NP, MU, 1, TRIAL_ROLL>(p, nullptr, sa, b, reinterpret_cast<T*>(smem_raw));
}

template <typename T, int NX, int NP, int MU, int PROGID = 0>
__global__ void __launch_bounds__(64) ilq_rows_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = sa.ids ? sa.ids[blockIdx.y] : int(blockIdx.y);
  {
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    const int stage = reinterpret_cast<const SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage;
    if (stage != ST_ROLLOUT && stage != ST_QUAD) return;
  }
  // compact rows: nothing dense is written, so the word maps of the dense images are not needed (nor their LDS)
  const short* maps = sa.compact ? nullptr : rows_maps_load(p, smem_raw);
  T* sm = reinterpret_cast<T*>(smem_raw + (sa.compact ? 0 : rows_maps_bytes(p)));
  rows_part_instance<T, NX, NP, MU, PROGID>(p, maps, sa, b, int(blockIdx.x), sm);
}

template <typename T, int NX, int NP, int MU>
__global__ void __launch_bounds__(64) ilq_decide_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = sa.ids ? sa.ids[blockIdx.x] : int(blockIdx.x);
  {
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    const int stage = reinterpret_cast<const SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage;
    if (stage != ST_ROLLOUT && stage != ST_QUAD) return;
  }
  trial_part_instance<T, NX, NP, MU, 1, TRIAL_DECIDE>(p, nullptr, sa, b, reinterpret_cast<T*>(smem_raw));
}

// Speculative line search of the listed instances (ilqg_solve.hpp): candidate j of list entry `slot`.
// FAT (fp64): compiled for two waves per SIMD instead of ILQG_ROLL_WAVES — 256 registers, none spilled.  At four waves per
// SIMD the paired fp64 rollout keeps 42 registers in scratch memory, reloaded inside the time-step chain: a round of a few
// candidates, which costs the latency of one rollout however empty the chip is, runs a third faster without them (n = 16,
// 6 instances x 128 candidates: 248 -> 166 us; the n = 16 constrained workload at B = 1024: 517 -> 533 k it/s); a round
// that fills the chip wants the occupancy back (config 5's scene: 255 k it/s lean, 244 k fat).  The launcher picks.
// SINGLE: one candidate per wavefront (rollout_instance) also where the shape has the paired form — for a round whose
// candidates all find a SIMD of their own, where a launch costs one rollout's chain and the paired chain is the longer
// one (n = 15, a few instances x 128 candidates: 176 us paired).
template <typename T, int NX, int NP, int MU, bool FAT = false, bool SINGLE = false>
__global__ void __launch_bounds__(64, sizeof(T) == 8 ? (FAT ? 2 : ILQG_ROLL_WAVES) : 1) ilq_probe_roll_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  if constexpr (rollout_pairs(NX, NP, MU) && !SINGLE)  // candidates 2 y and 2 y + 1 in the two halves of the wave
    probe_roll_pair<T, NX, NP, MU>(p, sa, sa.ids[blockIdx.x], blockIdx.x, 2 * int(blockIdx.y), reinterpret_cast<T*>(smem_raw));
  else
    probe_roll_instance<T, NX, NP, MU>(p, sa, sa.ids[blockIdx.x], blockIdx.x, blockIdx.y, reinterpret_cast<T*>(smem_raw));
}

// The probing rollouts of a round that fills the chip: 64 / NP candidates of one instance per wavefront, a lane per
// (candidate, subsystem), the RK4 stages in sequence (rollout_lanes, ilqg_stages.hpp) — an eighth of the instructions
// per trajectory of the paired form above, a longer chain per step: the launcher takes it where the round has more
// rollouts than the chip holds at once.
template <typename T, int NX, int NP, int MU>
__global__ void __launch_bounds__(64, 4) ilq_probe_roll_lanes_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  if constexpr (rollout_pairs(NX, NP, MU))
    probe_roll_lanes<T, NX, NP, MU>(p, sa, sa.ids[blockIdx.x], blockIdx.x, rollout_lanes_per_wave(NP) * int(blockIdx.y),
                                    reinterpret_cast<T*>(smem_raw));
}

template <typename T, int NX, int NP, int MU, int PROGID = 0>
__global__ void __launch_bounds__(64, sizeof(T) == 8 ? 4 : 1) ilq_probe_rows_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int slot = blockIdx.y / sa.probe_k, j = blockIdx.y % sa.probe_k;
  const int b = sa.ids[slot];
  {  // leave before the table load if this candidate is not wanted
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    const SolveState<T> s = state_load<T>(sa.ws + size_t(b) * sa.ws_stride, L);
    if (!probe_wanted(sa, s, j)) return;
  }
  const short* maps = sa.compact ? nullptr : rows_maps_load(p, smem_raw);  // merit only: no dense image either way
  T* sm = reinterpret_cast<T*>(smem_raw + (sa.compact ? 0 : rows_maps_bytes(p)));
  probe_rows_instance<T, NX, NP, MU, PROGID>(p, maps, sa, b, slot, j, int(blockIdx.x), sm);
}

template <typename T>
__global__ void __launch_bounds__(kProbeCandidates) ilq_probe_pick_kernel(DevProblem p, SolveArgs<T> sa) {
  __shared__ T merits[kProbeCandidates];  // the candidates' merit values, reduced here one lane per candidate
  probe_pick_instance<T>(p, sa, sa.ids[blockIdx.x], blockIdx.x, merits);
}

// Iterate log and anytime exit (ilqg_solve.hpp): only launched by solves that ask for them.
template <typename T>
__global__ void __launch_bounds__(256) ilq_log_kernel(DevProblem p, SolveArgs<T> sa, IterLog<T> lg) {
  log_part_instance<T>(p, sa, lg, int(blockIdx.x));
}
template <typename T>
__global__ void __launch_bounds__(64) ilq_deadline_kernel(DevProblem p, SolveArgs<T> sa) {
  deadline_part_instance<T>(p, sa, int(blockIdx.x));
}

// Exit kernel: return path of ILQSolver::Solve / AugmentedLagrangianSolver bookkeeping for the instances
// whose inner solve has ended (converged, out of iterations, or line search exhausted).
template <typename T, int NX, int NP, int MU>
__global__ void __launch_bounds__(256) ilq_exit_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x;
  {
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    const int stage = reinterpret_cast<const SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage;
    if (stage != ST_INNER_DONE) return;
  }
  const QuadTables<T> tb = quad_tables_load<T>(p, smem_raw);
  T* sm = reinterpret_cast<T*>(smem_raw + quad_tables_bytes(p, sizeof(T)));
  exit_part_instance<T, NX, NP, MU>(p, tb, sa, b, sm);
}

// LQ kernel: the Riccati sweep at the accepted operating point of every instance that asked for one.
// KIND LQ_PLAYER_WAVES_PACKED is LQ_PLAYER_WAVES compiled for four waves per SIMD (128 registers): the fp32 one-tile
// sweep is three registers over that by itself, and at batches of many instances per CU a fifth resident instance is
// worth more than the two spilled registers cost (n = 14 fp32, B = 8192: 2.15 M vs 2.04 M it/s; B = 1024, where only
// four instances per CU exist: 1.81 M vs 1.85 M — so the launcher picks it for large batches only).
#ifndef ILQG_1W_WAVES_F64
#define ILQG_1W_WAVES_F64 2
#endif
// KIND LQ_SINGLE_WAVE: one wave per instance (ilqg_lq_feedback1w.hpp), compiled for as many waves per SIMD as its LDS
// lets a CU hold instances (fp64: 20 KB -> eight per CU, two per SIMD; fp32: 10 KB -> sixteen, four per SIMD).
template <typename T, int NX, int NP, int MU, int KIND>
__global__ void __launch_bounds__((KIND == LQ_SINGLE_WAVE ? 64 : KIND == LQ_VALU_FEEDBACK ? LQCfg<T, NX, NP, MU>::NT : ((KIND == LQ_OPEN_LOOP || KIND == LQ_OPEN_LOOP_COMPACT) ? OLCfg<T, NX, NP, MU>::NT : 64 * NP)),
                                  (KIND == LQ_SINGLE_WAVE ? (sizeof(T) == 4 ? 4 : ILQG_1W_WAVES_F64) : KIND == LQ_PLAYER_WAVES_PACKED ? 4 : KIND == LQ_PLAYER_WAVES ? (NX <= 16 ? NP : 2) : ((KIND == LQ_OPEN_LOOP || KIND == LQ_OPEN_LOOP_COMPACT) ? 3 : 1)))
ilq_lq_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x;
  if (sa.clear_counters && b == 0 && threadIdx.x < 4) sa.unfinished[threadIdx.x] = 0;  // (SolveArgs::clear_counters)
  {
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    const int stage = reinterpret_cast<const SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage;
    if (stage != ST_LQ) return;
  }
  lq_part_instance<T, NX, NP, MU, (KIND == LQ_PLAYER_WAVES_PACKED ? LQ_PLAYER_WAVES : KIND)>(p, sa, b, reinterpret_cast<T*>(smem_raw));
}

// ---------------------------------------------------------------------------------------------------------------
// The sweep of a run-time-dimensioned solve on a SPECIALISED sweep: a game whose (n, N, m_i) has no instantiation is
// embedded in the smallest instantiated shape (NX >= n, NP == N, MU >= max m_i) —
//   states n .. NX - 1:   A = 1 on the diagonal, no column of any B, no row of any Q_i or l_i;
//   controls m_i .. MU - 1 of player i:  a zero column of B_i, a unit diagonal entry of R_ii, zeros in every other block —
// which solves to exact zeros in the added rows of [P | alpha] and leaves every other entry the recursion of
// src/lq_feedback_solver.cpp:110-213 (src/lq_open_loop_solver.cpp:73-195) produces for the game itself: the added
// column of S = R + B^T Z B is the unit vector (the Gershgorin step of :163-176 reads a column at a time and leaves it
// alone: radius 0, diagonal 1), the added rows and columns of every Z_i stay zero.  One workgroup per instance whose
// stage is LQ: (1) the instance's dense rows, written by the run-time-dimensioned row stage, are copied into the padded
// layout (a region of the library's scratch buffer), (2) the specialised sweep of the padded shape runs on them —
// strategies, delta_x and ILQSolver::ExpectedDecrease as in lq_part_instance —, (3) the rows of [P | alpha] that belong
// to the game are copied to where the solve keeps its strategies.  The copies are ~2 x the sweep's own traffic; the sweep
// is the register-tiled one instead of the all-LDS form with a barrier per phase (ilqg_lq_generic.hpp): n = 8,
// m_i = (1, 2), B = 1024: 1.7 ms -> see DESIGN.md 3.8.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int NX, int NP, int MU, bool OL>
struct PadLayout {
  static constexpr int M = NP * MU;
  size_t A, B, Q, l, R, r, P, al, dx, scr, x0, total;
  __host__ __device__ PadLayout(int T_steps, int Rsz, int rsz) {
    size_t o = 0;
    auto take = [&](size_t cnt) {
      const size_t at = o;
      o += (cnt + 3) & ~size_t(3);
      return at;
    };
    const size_t Tn = size_t(T_steps);
    A = take(Tn * NX * NX);
    B = take(Tn * NX * M);
    Q = take(Tn * NP * NX * NX);
    l = take(Tn * NP * NX);
    R = take(Tn * Rsz);
    r = take(Tn * rsz);
    P = take(Tn * M * NX);
    al = take(Tn * M);
    dx = take(Tn * NX);
    scr = take(Tn * size_t(OL ? OLCfg<T, NX, NP, MU>::ROW : NP * (NX + 1) + NX));
    x0 = take(NX);
    total = o;
  }
};

template <typename T>
struct PadArgs {
  T* pad;             // [batch][PadLayout::total]
  size_t pad_stride;  // elements
  PairTable ptp;      // the game's control blocks at MU x MU each
};

template <typename T, int NX, int NP, int MU, bool OL>
struct PadSweep {
  using C = LQCfg<T, NX, NP, MU>;
  static constexpr bool PW = !OL && C::USE_MFMA;
  static constexpr int NT = OL ? OLCfg<T, NX, NP, MU>::NT : LQFeedbackThreads<T, NX, NP, MU, false>::NT;
  static constexpr int WG_PER_CU = OL ? 3 : (PW ? (NX <= 16 ? NP : 2) : 1);
  // the slot the sweep leaves ILQSolver::ExpectedDecrease in (lq_part_instance), and the LDS the launch asks for
  static constexpr int ED_SLOT = OL ? OLCfg<T, NX, NP, MU>::LDS_ELEMS
                                 : (PW && !C::MFMA_ONE_TILE) ? FB2Cfg<T, NX, NP, MU>::LDS_ELEMS : C::oX;
  static constexpr size_t LDS_ELEMS = OL ? OLCfg<T, NX, NP, MU>::LDS_ELEMS + 4
                                      : PW ? MfmaSweepLds<T, NX, NP, MU>::ELEMS + (C::MFMA_ONE_TILE ? 0 : 4) : C::LDS_ELEMS;
};

// The game as the copies see it: its own dimensions and control blocks, and where its rows are.
template <typename T>
struct PadGame {
  int n, m, N, T_steps;
  const int *udim, *uoff;            // [N], [N + 1]
  const PairTable* pt;               // the game's blocks at their own sizes
  const T *A, *Bm, *Q, *l, *R, *r;   // instance bases, the game's dense rows
  const T* x0;                       // [n] or nullptr
  T *P, *alpha, *dx;                 // instance bases of the results; dx may be nullptr
  int want_ed;                       // ILQSolver::ExpectedDecrease into sm[PadSweep::ED_SLOT]
  int adaptive, symmetric;
  const int* xoff;                   // open loop: [N + 1] state offsets of a block-diagonal A, or nullptr (dense)
};

// Steps (1) - (3) above for one instance, by the whole workgroup; `q`: the instance's region of the padded buffer.
template <typename T, int NX, int NP, int MU, bool OL>
__device__ __forceinline__ void padded_sweep_instance(const PadGame<T>& g, const PairTable& ptp, T* q, T* sm) {
  using PS = PadSweep<T, NX, NP, MU, OL>;
  constexpr int M = NP * MU, NT = PS::NT;
  const int tid = threadIdx.x;
  const int n = g.n, m = g.m, Tn = g.T_steps;
  const PadLayout<T, NX, NP, MU, OL> PL(Tn, ptp.Rsz, ptp.rsz);
  // (eight loads in flight per lane: the workgroup is two to four waves, and a load-store pair per trip leaves the
  // copy bound by one memory latency per element)
  auto copy = [&](T* dst, int count, auto src_of) {
    constexpr int U = 8;
    for (int e0 = tid; e0 < count; e0 += NT * U) {
      T v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * NT;
        v[u] = e < count ? src_of(e) : T(0);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int e = e0 + u * NT;
        if (e < count) dst[e] = v[u];
      }
    }
  };
  // ---- (1) dense rows of the game -> rows of the padded shape ----
  {
    const T *A = g.A, *Bm = g.Bm, *Q = g.Q, *l = g.l, *R = g.R, *r = g.r;
    const PairTable& pt = *g.pt;
    copy(q + PL.A, Tn * NX * NX, [&](int e) {
      const int k = e / (NX * NX), f = e - k * (NX * NX), c = f / NX, row = f - c * NX;
      return (row < n && c < n) ? A[size_t(k) * n * n + row + n * c] : (row == c ? T(1) : T(0));
    });
    copy(q + PL.B, Tn * NX * M, [&](int e) {
      const int k = e / (NX * M), f = e - k * (NX * M), c = f / NX, row = f - c * NX, i = c / MU, ce = c - i * MU;
      return (row < n && ce < g.udim[i]) ? Bm[size_t(k) * n * m + row + n * (g.uoff[i] + ce)] : T(0);
    });
    copy(q + PL.Q, Tn * NP * NX * NX, [&](int e) {
      const int ki = e / (NX * NX), f = e - ki * (NX * NX), c = f / NX, row = f - c * NX;  // ki = k * NP + i
      return (row < n && c < n) ? Q[size_t(ki) * n * n + row + n * c] : T(0);
    });
    copy(q + PL.l, Tn * NP * NX, [&](int e) {
      const int ki = e / NX, row = e - ki * NX;
      return row < n ? l[size_t(ki) * n + row] : T(0);
    });
    const int Rsz_p = ptp.Rsz, rsz_p = ptp.rsz;  // npairs * MU * MU, npairs * MU
    copy(q + PL.R, Tn * Rsz_p, [&](int e) {
      const int k = e / Rsz_p, f = e - k * Rsz_p, pq = f / (MU * MU), h = f - pq * (MU * MU), cb = h / MU, ca = h - cb * MU;
      const int mj = g.udim[pt.pj[pq]];
      return (ca < mj && cb < mj) ? R[size_t(k) * pt.Rsz + pt.roff[pq] + ca + mj * cb]
                                  : ((pt.pi[pq] == pt.pj[pq] && ca == cb) ? T(1) : T(0));
    });
    copy(q + PL.r, Tn * rsz_p, [&](int e) {
      const int k = e / rsz_p, f = e - k * rsz_p, pq = f / MU, ca = f - pq * MU;
      const int mj = g.udim[pt.pj[pq]];
      return ca < mj ? r[size_t(k) * pt.rsz + pt.rgoff[pq] + ca] : T(0);
    });
    if (g.x0 && tid < NX) q[PL.x0 + tid] = tid < n ? g.x0[tid] : T(0);
  }
  __threadfence_block();
  __syncthreads();
  // ---- (2) the specialised sweep (dense rows, its own forward pass) ----
  LQArgs<T> la;
  la.A = q + PL.A; la.Bm = q + PL.B; la.Q = q + PL.Q; la.l = q + PL.l; la.R = q + PL.R; la.r = q + PL.r;
  la.x0 = g.x0 ? q + PL.x0 : nullptr;
  la.P = q + PL.P; la.alpha = q + PL.al;
  const bool forward = g.dx != nullptr || g.want_ed;
  la.dx = forward ? q + PL.dx : nullptr;
  la.scratch = (forward || OL) ? q + PL.scr : nullptr;
  la.ed_out = g.want_ed ? sm + PS::ED_SLOT : nullptr;
  la.T_steps = Tn;
  la.adaptive = g.adaptive;
  la.symmetric = g.symmetric;
  if constexpr (OL) {
    if (g.xoff) {  // the added states extend the last player's block (their A entries are its diagonal's)
      la.nsub = NP;
#pragma unroll
      for (int i = 0; i < NP; i++) la.xoff[i] = g.xoff[i];
      la.xoff[NP] = NX;
    }
    lq_openloop_instance<T, NX, NP, MU>(la, ptp, sm);
  } else {
    lq_feedback_dispatch<T, NX, NP, MU, false>(la, ptp, sm);
  }
  __threadfence_block();
  __syncthreads();
  // ---- (3) the game's rows of [P | alpha] (and of delta_x) ----
  copy(g.P, Tn * m * n, [&](int e) {
    const int k = e / (m * n), f = e - k * (m * n), c = f / m, row = f - c * m;
    int i = 0;
    while (i + 1 < g.N && row >= g.uoff[i + 1]) i++;
    return q[PL.P + size_t(k) * M * NX + (i * MU + row - g.uoff[i]) + M * c];
  });
  copy(g.alpha, Tn * m, [&](int e) {
    const int k = e / m, row = e - k * m;
    int i = 0;
    while (i + 1 < g.N && row >= g.uoff[i + 1]) i++;
    return q[PL.al + size_t(k) * M + i * MU + row - g.uoff[i]];
  });
  if (g.dx)
    copy(g.dx, Tn * n, [&](int e) {
      const int k = e / n, row = e - k * n;
      return q[PL.dx + size_t(k) * NX + row];
    });
}

template <typename T, int NX, int NP, int MU, bool OL>
__global__ void __launch_bounds__((PadSweep<T, NX, NP, MU, OL>::NT), (PadSweep<T, NX, NP, MU, OL>::WG_PER_CU))
padded_lq_kernel(DevProblem p, SolveArgs<T> sa, PadArgs<T> pa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  using PS = PadSweep<T, NX, NP, MU, OL>;
  const int b = blockIdx.x;
  const int Tn = p.T;
  const WsLayout L(p.n, p.m, p.N, Tn, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
  T* const w = sa.ws + size_t(b) * sa.ws_stride;
  SolveState<T>* const st = reinterpret_cast<SolveState<T>*>(w + L.state);
  if (st->stage != ST_LQ) return;
  const int sacc = __builtin_amdgcn_readfirstlane(st->sacc);
  PadGame<T> g;
  g.n = p.n; g.m = p.m; g.N = p.N; g.T_steps = Tn;
  g.udim = p.udim; g.uoff = p.uoff; g.pt = &p.pairs;
  g.A = w + L.A; g.Bm = w + L.B; g.Q = w + L.Q; g.l = w + L.l; g.R = w + L.R; g.r = w + L.r;
  g.x0 = nullptr;
  g.P = sacc ? sa.P + size_t(b) * Tn * p.m * p.n : w + L.P1;  // strategy buffer 1 - sacc
  g.alpha = sacc ? sa.alpha + size_t(b) * Tn * p.m : w + L.al1;
  g.dx = nullptr;
  g.want_ed = 1;
  g.adaptive = 1;
  g.symmetric = 1;  // what the quadraticisation stage writes
  bool blocks = OL;
  for (int i = 0; i < p.N; i++) blocks = blocks && p.xoff[i + 1] > p.xoff[i];
  g.xoff = blocks ? p.xoff : nullptr;
  padded_sweep_instance<T, NX, NP, MU, OL>(g, pa.ptp, pa.pad + size_t(b) * pa.pad_stride, sm);
  if (threadIdx.x == 0) {
    st->expected_decrease = sm[PS::ED_SLOT];
    st->num_iterations += 1;
    st->step = sa.forced_steps ? sa.forced_steps[size_t(b) * sa.fixed_iters + (st->num_iterations - 1)]
                               : T(sa.prm.initial_alpha_scaling);
    st->bt = 0;
    st->stage = ST_ROLLOUT;
  }
}

// ilqg_lq_feedback_batch / ilqg_lq_openloop_batch for a shape without an instantiation of its own, on the padded sweep.
template <typename T, int NX, int NP, int MU, bool OL>
__global__ void __launch_bounds__((PadSweep<T, NX, NP, MU, OL>::NT), (PadSweep<T, NX, NP, MU, OL>::WG_PER_CU))
padded_lq_batch_kernel(LQBatchArgs<T> a, GenDims d, PairTable pt, PadArgs<T> pa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* sm = reinterpret_cast<T*>(smem_raw);
  const size_t b = blockIdx.x, Tn = d.T, n = d.n, m = d.m, N = d.N;
  PadGame<T> g;
  g.n = d.n; g.m = d.m; g.N = d.N; g.T_steps = d.T;
  g.udim = d.udim; g.uoff = d.uoff; g.pt = &pt;
  g.A = a.A + b * Tn * n * n;
  g.Bm = a.Bm + b * Tn * n * m;
  g.Q = a.Q + b * Tn * N * n * n;
  g.l = a.l + b * Tn * N * n;
  g.R = a.R + b * Tn * pt.Rsz;
  g.r = a.r + b * Tn * pt.rsz;
  g.x0 = a.x0 ? a.x0 + b * n : nullptr;
  g.P = a.P + b * Tn * m * n;
  g.alpha = a.alpha + b * Tn * m;
  g.dx = a.dx ? a.dx + b * Tn * n : nullptr;
  g.want_ed = 0;
  g.adaptive = OL ? 0 : a.adaptive;
  g.symmetric = 0;
  g.xoff = nullptr;
  padded_sweep_instance<T, NX, NP, MU, OL>(g, pa.ptp, pa.pad + b * pa.pad_stride, sm);
}

// The sweep of the run-time-dimensioned solve path (lq_part_generic, ilqg_solve.hpp).
template <typename T>
__global__ void __launch_bounds__(256) gen_lq_kernel(DevProblem p, SolveArgs<T> sa) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int b = blockIdx.x;
  {
    const WsLayout L(p.n, p.m, p.N, p.T, p.pairs.Rsz, p.pairs.rsz, sa.ol_row, p.num_constraints, sa.al_mode);
    const int stage = reinterpret_cast<const SolveState<T>*>(sa.ws + size_t(b) * sa.ws_stride + L.state)->stage;
    if (stage != ST_LQ) return;
  }
  lq_part_generic<T>(p, sa, b, reinterpret_cast<T*>(smem_raw));
}

template <typename T>
__global__ void mfma_selftest_kernel(const T* X, const T* Y, const T* C, T* out) {
  mfma_selftest<T>(X, Y, C, out);
}

// The measured-bandwidth denominator of bench.py's roofline ("measure achievable BW on the box with a copy
// kernel"): every workgroup copies one contiguous 16 KB piece, four 16-byte loads per lane in flight before the first
// store.  scripts/ubench/copy_bw.hip compares shapes on the box: this one 5.65-5.98 TB/s, a grid-stride loop over the
// same pieces 4.3-5.6, hipMemcpyDtoD 5.5.
typedef float copy_v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) copy_bandwidth_kernel(copy_v4f* __restrict__ dst, const copy_v4f* __restrict__ src, size_t n16) {
  const size_t b0 = size_t(blockIdx.x) * 1024, b1 = b0 + 1024 < n16 ? b0 + 1024 : n16;
  const size_t i = b0 + threadIdx.x;
  if (b1 - b0 == 1024) {
    const copy_v4f a = src[i], b = src[i + 256], c = src[i + 512], e = src[i + 768];
    dst[i] = a;
    dst[i + 256] = b;
    dst[i + 512] = c;
    dst[i + 768] = e;
  } else {
    for (size_t k = i; k < b1; k += 256) dst[k] = src[k];
  }
}

// ------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------

// Large dynamic-LDS launches: ask for the opt-in limit; a refusal is not fatal by itself (the launch
// reports the real error if the size is unusable), so it must not poison the sticky error state.
void raise_lds_limit(const void* kern, size_t lds) {
  if (lds <= 48 * 1024) return;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) (void)hipGetLastError();
}

bool build_pairs(const ilqg_pair* pairs, int npairs, const int* udim, int N, PairTable* pt, std::string* err) {
  if (npairs > kMaxPairs) {
    *err = "too many control blocks";
    return false;
  }
  std::memset(pt, 0, sizeof(*pt));
  pt->npairs = npairs;
  for (int i = 0; i < kMaxPlayers; i++) pt->pii[i] = -1;
  int Rsz = 0, rsz = 0;
  for (int q = 0; q < npairs; q++) {
    const int i = pairs[q].i, j = pairs[q].j;
    if (i < 0 || i >= N || j < 0 || j >= N) {
      *err = "control block index out of range";
      return false;
    }
    pt->pi[q] = i;
    pt->pj[q] = j;
    pt->roff[q] = Rsz;
    pt->rgoff[q] = rsz;
    pt->from_cost[q] = 1;
    Rsz += udim[j] * udim[j];
    rsz += udim[j];
    if (i == j) pt->pii[i] = q;
  }
  pt->Rsz = Rsz;
  pt->rsz = rsz;
  for (int i = 0; i < N; i++)
    if (pt->pii[i] < 0) {
      *err = "player " + std::to_string(i) + " is missing a control Hessian";  // lq_feedback_solver.cpp:139-140
      return false;
    }
  return true;
}

// Supported (n, N, m_i) instantiations.  n=14/16/15/24: BASELINE configs 2-5;
// (4,2,2): config 1 (TwoPlayerUnicycle4D); (2,2,1): test_lq_solver's point mass;
// (6,3,2): synthetic parity cases.
#if defined(ILQG_DIMS_HEADER)
#include ILQG_DIMS_HEADER  // experiment builds: a generated subset of the list below (__graft_entry__.build_hip_library)
#else
#define ILQG_FOR_DIMS(X) X(14, 3, 2) X(16, 3, 2) X(15, 3, 2) X(24, 4, 2) X(18, 3, 2) X(12, 2, 2) X(10, 2, 2) X(4, 2, 2) X(6, 2, 1) X(3, 2, 1) X(3, 1, 1) X(2, 2, 1) X(6, 3, 2) X(2, 1, 2) X(8, 2, 2) X(17, 3, 2) X(8, 2, 1)
#endif

}  // namespace

// Launchers of the kernels that are instantiated per (n, N, m_i).  Members are defined out of class (not inline),
// so `extern template struct DimsLaunch<...>` in the main unit of a split build leaves their code — and the
// kernels behind them — to the unit that instantiates them explicitly.
template <typename T, int NX, int NP, int MU>
struct __attribute__((visibility("hidden"))) DimsLaunch {
  static ilqg_status lq(const ilqg_dims* d, const PairTable& pt, const void* A, const void* Bm, const void* Q,
                        const void* l, const void* R, const void* r, const void* x0, void* P, void* alpha, void* dx,
                        hipStream_t stream);
  static ilqg_status lq_openloop(const ilqg_dims* d, const PairTable& pt, const void* A, const void* Bm,
                                 const void* Q, const void* l, const void* R, const void* r, const void* x0, void* P,
                                 void* alpha, void* dx, void* costates, hipStream_t stream);
  static ilqg_status solve(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us, void* P, void* alpha,
                           void* total_costs, int32_t* iters, int32_t* status, int32_t* converged, void* workspace,
                           const ilqg_solve_options& opt, hipStream_t stream);
  // pointers in QuadBatchArgs' order: xs us lambdas mu t_extreme A Bm Q l R r merit_part cost_part active
  static ilqg_status rows(const DevProblem& d, int32_t batch, const void* const* ptrs, hipStream_t stream);
  // the sweep of a run-time-dimensioned solve whose game is embedded in this shape (padded_lq_kernel);
  // pad_elems_out != nullptr: only report the scratch elements one instance needs
  static ilqg_status lq_padded(const DevProblem& d, const void* solve_args, const PairTable& ptp, void* pad,
                               size_t* pad_elems_out, bool open_loop, hipStream_t stream);
  // ilqg_lq_feedback_batch / ilqg_lq_openloop_batch of a game embedded in this shape (padded_lq_batch_kernel): `d`, `pt`
  // are the game's own dimensions and control blocks, `ptp` its blocks at MU x MU
  static ilqg_status lq_padded_batch(const ilqg_dims* d, const PairTable& pt, const PairTable& ptp, bool open_loop,
                                     const void* A, const void* Bm, const void* Q, const void* l, const void* R,
                                     const void* r, const void* x0, void* P, void* alpha, void* dx, hipStream_t stream);
};

template <typename T, int NX, int NP, int MU>
ilqg_status DimsLaunch<T, NX, NP, MU>::rows(const DevProblem& d, int32_t batch, const void* const* q,
                                            hipStream_t stream) {
  const QuadBatchArgs<T> g{(const T*)q[0], (const T*)q[1], (const T*)q[2], (const T*)q[3], (const int*)q[4], (T*)q[5],
                           (T*)q[6], (T*)q[7], (T*)q[8], (T*)q[9], (T*)q[10], (T*)q[11], (T*)q[12], (const int*)q[13]};
  // the widest chunk that leaves a CU three workgroups of this kernel
  const int cw = rows_chunk_width(d.n, d.m, d.rp_pslots, d.rp_lslots, sizeof(T), size_t(48) * 1024);
  const size_t lds = rows_maps_bytes(d) + rows_lds_elems(d.n, d.m, d.rp_pslots, d.rp_lslots, cw) * sizeof(T);
  auto kern = rows_kernel<T, NX, NP, MU>;
  raise_lds_limit((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3((d.T + cw - 1) / cw, batch), dim3(64), lds, stream, d, g, cw);
  HIP_TRY(hipGetLastError());
  return ILQG_OK;
}

template <typename T, int NX, int NP, int MU>
ilqg_status DimsLaunch<T, NX, NP, MU>::lq(const ilqg_dims* d, const PairTable& pt, const void* A, const void* Bm,
                                          const void* Q, const void* l, const void* R, const void* r, const void* x0,
                                          void* P, void* alpha, void* dx, hipStream_t stream) {
  using C = LQCfg<T, NX, NP, MU>;
  LQBatchArgs<T> g;
  g.A = (const T*)A; g.Bm = (const T*)Bm; g.Q = (const T*)Q; g.l = (const T*)l;
  g.R = (const T*)R; g.r = (const T*)r; g.x0 = (const T*)x0;
  g.P = (T*)P; g.alpha = (T*)alpha; g.dx = (T*)dx;
  g.scratch = nullptr;
  if (dx) {
    const size_t need = size_t(d->batch) * d->T * (NP * (NX + 1) + NX) * sizeof(T);
    ilqg_status s = Scratch().reserve(need);
    if (s != ILQG_OK) return s;
    g.scratch = (T*)ilqg_shared::scratch_state().ptr;
  }
  g.T_steps = d->T;
  g.adaptive = d->adaptive_regularization;
  g.batch = d->batch;
  g.force_valu = 0;
  // ilqg_dims::sweep_formulation = ILQG_CHOICE_OFF selects the VALU/LDS formulation where the MFMA one is the default
  const bool valu = C::USE_MFMA && d->sweep_formulation == ILQG_CHOICE_OFF;
  const bool use_pw = C::USE_MFMA && !valu;
  const size_t lds = size_t(use_pw ? MfmaSweepLds<T, NX, NP, MU>::ELEMS : C::LDS_ELEMS) * sizeof(T);
  auto kern = valu ? lq_feedback_kernel<T, NX, NP, MU, true> : lq_feedback_kernel<T, NX, NP, MU, false>;
  const int nt = valu ? LQFeedbackThreads<T, NX, NP, MU, true>::NT : LQFeedbackThreads<T, NX, NP, MU, false>::NT;
  raise_lds_limit((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(d->batch), dim3(nt), lds, stream, g, pt);
  HIP_TRY(hipGetLastError());
  return ILQG_OK;
}

template <typename T, int NX, int NP, int MU>
ilqg_status DimsLaunch<T, NX, NP, MU>::lq_openloop(const ilqg_dims* d, const PairTable& pt, const void* A,
                                                   const void* Bm, const void* Q, const void* l, const void* R,
                                                   const void* r, const void* x0, void* P, void* alpha, void* dx,
                                                   void* costates, hipStream_t stream) {
  using O = OLCfg<T, NX, NP, MU>;
  LQBatchArgs<T> g;
  g.A = (const T*)A; g.Bm = (const T*)Bm; g.Q = (const T*)Q; g.l = (const T*)l;
  g.R = (const T*)R; g.r = (const T*)r; g.x0 = (const T*)x0;
  g.P = (T*)P; g.alpha = (T*)alpha; g.dx = (T*)dx;
  g.costates = (T*)costates;
  const size_t need = size_t(d->batch) * d->T * (costates ? O::ROW_FAT : O::ROW) * sizeof(T);
  ilqg_status s = Scratch().reserve(need);
  if (s != ILQG_OK) return s;
  g.scratch = (T*)ilqg_shared::scratch_state().ptr;
  g.T_steps = d->T;
  g.adaptive = 0;
  g.batch = d->batch;
  g.force_valu = 0;
  const size_t lds = size_t(O::LDS_ELEMS) * sizeof(T);
  auto kern = lq_openloop_kernel<T, NX, NP, MU>;
  raise_lds_limit((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(d->batch), dim3(O::NT), lds, stream, g, pt);
  HIP_TRY(hipGetLastError());
  return ILQG_OK;
}

template <typename T, int NX, int NP, int MU>
ilqg_status DimsLaunch<T, NX, NP, MU>::lq_padded(const DevProblem& d, const void* solve_args, const PairTable& ptp,
                                                 void* pad, size_t* pad_elems_out, bool open_loop, hipStream_t stream) {
  if constexpr (NX == 0) {
    return fail(ILQG_ERR_UNSUPPORTED, "no shape to embed the game in");
  } else {
    auto go = [&](auto ol) -> ilqg_status {
      constexpr bool OL = decltype(ol)::value;
      const PadLayout<T, NX, NP, MU, OL> PL(d.T, ptp.Rsz, ptp.rsz);
      if (pad_elems_out) {
        *pad_elems_out = PL.total;
        return ILQG_OK;
      }
      const SolveArgs<T>& sa = *static_cast<const SolveArgs<T>*>(solve_args);
      PadArgs<T> pa{(T*)pad, PL.total, ptp};
      using PS = PadSweep<T, NX, NP, MU, OL>;
      auto kern = padded_lq_kernel<T, NX, NP, MU, OL>;
      const size_t lds = PS::LDS_ELEMS * sizeof(T);
      raise_lds_limit((const void*)kern, lds);
      hipLaunchKernelGGL(kern, dim3(sa.batch), dim3(PS::NT), lds, stream, d, sa, pa);
      HIP_TRY(hipGetLastError());
      return ILQG_OK;
    };
    return open_loop ? go(std::true_type{}) : go(std::false_type{});
  }
}

template <typename T, int NX, int NP, int MU>
ilqg_status DimsLaunch<T, NX, NP, MU>::lq_padded_batch(const ilqg_dims* d, const PairTable& pt, const PairTable& ptp,
                                                       bool open_loop, const void* A, const void* Bm, const void* Q,
                                                       const void* l, const void* R, const void* r, const void* x0,
                                                       void* P, void* alpha, void* dx, hipStream_t stream) {
  if constexpr (NX == 0) {
    return fail(ILQG_ERR_UNSUPPORTED, "no shape to embed the game in");
  } else {
    auto go = [&](auto ol) -> ilqg_status {
      constexpr bool OL = decltype(ol)::value;
      const PadLayout<T, NX, NP, MU, OL> PL(d->T, ptp.Rsz, ptp.rsz);
      const ilqg_status s = Scratch().reserve(size_t(d->batch) * PL.total * sizeof(T));
      if (s != ILQG_OK) return s;
      GenDims gd{};
      gd.n = d->n; gd.N = d->num_players; gd.T = d->T;
      gd.uoff[0] = 0;
      for (int i = 0; i < gd.N; i++) {
        gd.udim[i] = d->udim[i];
        gd.uoff[i + 1] = gd.uoff[i] + d->udim[i];
      }
      gd.m = gd.uoff[gd.N];
      LQBatchArgs<T> g;
      g.A = (const T*)A; g.Bm = (const T*)Bm; g.Q = (const T*)Q; g.l = (const T*)l;
      g.R = (const T*)R; g.r = (const T*)r; g.x0 = (const T*)x0;
      g.P = (T*)P; g.alpha = (T*)alpha; g.dx = (T*)dx;
      g.scratch = nullptr;
      g.T_steps = d->T;
      g.adaptive = d->adaptive_regularization;
      g.batch = d->batch;
      g.force_valu = 0;
      PadArgs<T> pa{(T*)ilqg_shared::scratch_state().ptr, PL.total, ptp};
      using PS = PadSweep<T, NX, NP, MU, OL>;
      auto kern = padded_lq_batch_kernel<T, NX, NP, MU, OL>;
      const size_t lds = PS::LDS_ELEMS * sizeof(T);
      raise_lds_limit((const void*)kern, lds);
      hipLaunchKernelGGL(kern, dim3(d->batch), dim3(PS::NT), lds, stream, g, gd, pt, pa);
      HIP_TRY(hipGetLastError());
      return ILQG_OK;
    };
    return open_loop ? go(std::true_type{}) : go(std::false_type{});
  }
}

namespace {

bool uniform_udim(const int32_t* udim, int N, int* mu) {
  for (int i = 1; i < N; i++)
    if (udim[i] != udim[0]) return false;
  *mu = udim[0];
  return true;
}

ilqg_status check_device() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail(ILQG_ERR_NO_DEVICE, "no HIP device visible: libilqg_hip.so has no CPU fallback");
  return ILQG_OK;
}

}  // namespace

struct ilqg_problem {
  DevProblem dev;
  ilqg_problem_desc desc;
  std::vector<ilqg_cost_term> terms_host;
  DevTerm* d_terms = nullptr;
  int* d_poly_off = nullptr;
  float* d_poly_pts = nullptr;
  float* d_segs_f = nullptr;
  double* d_segs_d = nullptr;
  float* d_dense_f = nullptr;
  double* d_dense_d = nullptr;
  double* d_tnom_f = nullptr;
  double* d_tnom_d = nullptr;
  int* d_cost_order = nullptr;
  int* d_row_prog = nullptr;
  std::vector<int> row_prog_host;  // the program as built (ilqg_problem_row_program)
  int static_prog = 0;             // id of the registered structure it matches (ilqg_rowprog_static.hpp), 0: none
  int* d_unfinished = nullptr;  // instances still running after an LQ-kernel launch
  int* h_unfinished = nullptr;  // pinned host mirror: [0..3] the counters, [8] the sequence number of read_round_counters
  int* h_unfinished_dev = nullptr;  // ... as the device addresses it
  int publish_seq = 0;
  bool counters_clean = false;  // d_unfinished was cleared by the last thing that touched it (read_round_counters)
  int mu_uniform = 0;
  bool has_route_progress = false;  // a RouteProgressCost term: its tables are a first solve's (initial time 0)
  int last_schedule = 0;  // ILQG_SCHEDULE_* of the last solve (ilqg_problem_last_schedule)
  bool generic = false;  // no specialised instantiation holds this problem: every entry point runs the run-time-dimensioned kernels
  // LoopTimer of the solver object (include/ilqgames/utils/loop_timer.h:60-98, src/loop_timer.cpp:55-92): the last ten
  // iteration times, kept across solves as the reference's member is; only solves with a max_runtime feed and read it
  double loop_times[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int loop_count = 0, loop_next = 0;
  void loop_add(double seconds) {
    loop_times[loop_next] = seconds;
    loop_next = (loop_next + 1) % 10;
    if (loop_count < 10) loop_count++;
  }
  double loop_upper_bound() const {  // mean + 3 sigma (unbiased), 0.02 s until two samples exist
    if (loop_count < 2) return 0.02;
    double mean = 0.0, var = 0.0;
    for (int i = 0; i < loop_count; i++) mean += loop_times[i];
    mean /= loop_count;
    for (int i = 0; i < loop_count; i++) var += (loop_times[i] - mean) * (loop_times[i] - mean);
    return mean + 3.0 * std::sqrt(var / (loop_count - 1));
  }
  // AugmentedLagrangianSolver's own LoopTimer (solver/augmented_lagrangian_solver.h: `timer_`), over its outer iterations
  double al_loop_times[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int al_loop_count = 0, al_loop_next = 0;
  void al_loop_add(double seconds) {
    al_loop_times[al_loop_next] = seconds;
    al_loop_next = (al_loop_next + 1) % 10;
    if (al_loop_count < 10) al_loop_count++;
  }
  double al_loop_upper_bound() const {
    if (al_loop_count < 2) return 0.02;
    double mean = 0.0, var = 0.0;
    for (int i = 0; i < al_loop_count; i++) mean += al_loop_times[i];
    mean /= al_loop_count;
    for (int i = 0; i < al_loop_count; i++) var += (al_loop_times[i] - mean) * (al_loop_times[i] - mean);
    return mean + 3.0 * std::sqrt(var / (al_loop_count - 1));
  }
};

// The round counters' way back to the host.  A counted solve reads them once per round (twice with the augmented
// Lagrangian's restarts), and while it does the device idles: what a read-back costs is the gap between two rounds.
// A copy command plus a stream synchronisation is ~20 us of that; instead a one-wave kernel behind the round's last
// launch stores the four counters and then a sequence number into host memory (pinned, coherent, mapped), and the
// host spins on the sequence number.  ILQG_READBACK=copy keeps the copy + synchronise form (A/B measurements);
// a device fault shows up in the periodic hipStreamQuery.
namespace {
__global__ void ilq_publish_kernel(int* counts, int* host, int seq) {
  const int t = threadIdx.x;
  if (t < 4) {
    __hip_atomic_store(host + t, counts[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    counts[t] = 0;  // ... and the next round finds them cleared (ilqg_problem::counters_clean): one fill command less
  }
  __threadfence_system();
  if (t == 0) __hip_atomic_store(host + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// How many instances of a masked batch take part (ilqg_solve_options::active): into the fourth round counter.
__global__ void ilq_count_active_kernel(const int* active, int batch, int* counts) {
  int mine = 0;
  for (int b = threadIdx.x; b < batch; b += blockDim.x) mine += active[b] != 0;
  for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(counts + 3, mine);
}

inline ilqg_status read_round_counters(ilqg_problem* p, hipStream_t stream) {
  static const bool copy_form = [] {
    const char* e = getenv("ILQG_READBACK");
    return e && std::string(e) == "copy";
  }();
  p->counters_clean = false;
  if (copy_form || !p->h_unfinished_dev) {
    HIP_TRY(hipMemcpyAsync(p->h_unfinished, p->d_unfinished, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return ILQG_OK;
  }
  const int seq = ++p->publish_seq;
  hipLaunchKernelGGL(ilq_publish_kernel, dim3(1), dim3(64), 0, stream, p->d_unfinished, p->h_unfinished_dev, seq);
  HIP_TRY(hipGetLastError());
  p->counters_clean = true;
  volatile int* const flag = p->h_unfinished + 8;
  for (long long spins = 0;; spins++) {
    if (__atomic_load_n(const_cast<int*>(flag), __ATOMIC_ACQUIRE) == seq) return ILQG_OK;
    __builtin_ia32_pause();
    if ((spins & 0xffff) == 0xffff) {  // every ~65 k polls: has the stream died, or drained without our store?
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(const_cast<int*>(flag), __ATOMIC_ACQUIRE) == seq) return ILQG_OK;
        return fail(ILQG_ERR_HIP, "the round counters never reached the host");
      }
      if (q != hipErrorNotReady) return fail(ILQG_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
    }
  }
}
}  // namespace

// The outer loop's clock of AugmentedLagrangianSolver::Solve under a max_runtime (src/augmented_lagrangian_solver.cpp:
// 104-110,193): `elapsed` starts at the first inner solve's ALLOWANCE (max_runtime / max_solver_iters, not the time it
// took), every outer iteration adds its wall time, and another one starts only while
// elapsed < max_runtime - timer_.RuntimeUpperBound().  A batch shares the clock.  It runs without a gap from the first
// exit launch that restarted an instance until the solve returns — whoever is restarted or not at the launches in
// between (round 5 re-armed it only when a launch restarted somebody: with staggered instances whole inner solves went
// uncounted).  The LoopTimer's samples are outer-iteration durations: every restart-bearing exit launch opens an
// interval, and each later exit launch (one exists only when some inner solve has ended) closes the oldest open one —
// exact for a batch in lockstep, the inner-solve length of the earliest group when instances are staggered; never the
// gap between two neighbouring rounds.  before_exit() is called in front of every exit launch and says whether instances
// whose inner solve has ended may still restart.
struct AlOuterClock {
  ilqg_problem* p;
  bool on;
  double max_runtime, elapsed, last = 0.0;
  bool running = false;
  static constexpr int kOpen = 32;
  double open_since[kOpen];
  int open_head = 0, open_count = 0;
  AlOuterClock(ilqg_problem* p_, bool on_, double max_runtime_, double first_allowance)
      : p(p_), on(on_), max_runtime(max_runtime_), elapsed(first_allowance) {}
  static double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  bool before_exit() {  // true: the outer loop is closed
    if (!on) return false;
    const double now = wall();
    if (running) {
      elapsed += now - last;
      last = now;
      if (open_count > 0) {
        p->al_loop_add(now - open_since[open_head]);
        open_head = (open_head + 1) % kOpen;
        open_count--;
      }
    }
    return !(elapsed < max_runtime - p->al_loop_upper_bound());
  }
  void after_exit(int restarted) {
    if (!on || !restarted) return;
    const double now = wall();
    if (!running) {
      running = true;
      last = now;
    }
    if (open_count < kOpen) {
      open_since[(open_head + open_count) % kOpen] = now;
      open_count++;
    }
  }
};

#define DT_DISPATCH(p, CALL) ((p)->desc.dtype == ILQG_F32 ? CALL(float) : CALL(double))

static ilqg_status launch_linquad(const ilqg_problem* p, int32_t batch, const void* xs, const void* us,
                                  const void* lambdas, const void* mu, const int32_t* t_extreme, void* A, void* Bm,
                                  void* Q, void* l, void* R, void* r, void* merit_part, void* cost_part,
                                  const int32_t* active, void* stream);

template <typename T, int NX, int NP, int MU>
ilqg_status DimsLaunch<T, NX, NP, MU>::solve(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us,
                                             void* P, void* alpha, void* total_costs, int32_t* iters, int32_t* status,
                                             int32_t* converged, void* workspace, const ilqg_solve_options& opt,
                                             hipStream_t stream) {
  using C = LQCfg<T, NX, NP, MU>;
  const DevProblem& d = p->dev;
  const int32_t fixed_iters = opt.fixed_iters;
  const int al_mode = opt.augmented_lagrangian ? 1 : 0, resume = opt.resume ? 1 : 0;
  const int32_t* const active = opt.active;
  auto choice = [](int32_t c, bool automatic) { return c == ILQG_CHOICE_ON ? true : (c == ILQG_CHOICE_OFF ? false : automatic); };
  static_assert(OLCfg<T, NX, NP, MU>::ROW == ol_row_elems(NX, NP * MU, NP) && OLCfg<T, NX, NP, MU>::ROW_FAT == ol_row_elems(NX, NP * MU, NP, true), "ol_row_elems");
  const int ol_row = p->desc.params.open_loop ? ol_row_elems(d.n, d.m, d.N) : 0;
  const WsLayout L(d.n, d.m, d.N, d.T, d.pairs.Rsz, d.pairs.rsz, ol_row, d.num_constraints, al_mode);
  SolveArgs<T> sa{};
  sa.ol_row = ol_row;
  sa.al_mode = al_mode;
  sa.x0 = (const T*)x0; sa.xs = (T*)xs; sa.us = (T*)us; sa.P = (T*)P; sa.alpha = (T*)alpha;
  sa.total_costs = (T*)total_costs; sa.iters = iters; sa.status = status; sa.converged = converged;
  sa.ws = (T*)workspace; sa.ws_stride = L.total; sa.fixed_iters = fixed_iters; sa.batch = batch;
  sa.prm = p->desc.params;
  sa.active = active;
#if ILQG_DIAGNOSTIC_BUILD
  sa.prof = g_prof;
#else
  sa.prof = nullptr;
#endif
  sa.forced_steps = (const T*)opt.forced_steps;
  sa.unfinished = p->d_unfinished;
  // the tail of the workspace: the two lists of back-tracking instances and the line-search probe pool
  const WsTail tail = ws_tail(d, batch, sizeof(T), ol_row);
  // ws_tail places the lists behind the augmented-Lagrangian layout (the larger one): whatever al_mode this solve runs in
  if (WsLayout(d.n, d.m, d.N, d.T, d.pairs.Rsz, d.pairs.rsz, ol_row, d.num_constraints, 1).total < L.total)
    return fail(ILQG_ERR_INVALID, "workspace layout: the tail offset does not cover this solve's per-instance blocks");
  int* const pass_ids = reinterpret_cast<int*>(static_cast<char*>(workspace) + tail.ids_off);
  T* const probe_pool = reinterpret_cast<T*>(static_cast<char*>(workspace) + tail.pool_off);
  sa.ids = nullptr;
  sa.ids_next = nullptr;
  sa.probe_pool = nullptr;
  sa.probe_k = 0;
  constexpr int W = TrialWaves<T>::W;
  // LDS of the sweep kernel that will run: the open-loop sweep's own working set plus the slot the expected
  // decrease is handed over in (n = 24: 54 KB, three instances per CU; the feedback layout would take 85 KB)
  size_t lq_elems = (C::USE_MFMA && !p->desc.params.open_loop) ? MfmaSweepLds<T, NX, NP, MU>::ELEMS + (C::MFMA_ONE_TILE ? 0 : 4) : C::LDS_ELEMS;
  if (p->desc.params.open_loop) lq_elems = OLCfg<T, NX, NP, MU>::LDS_ELEMS + 4;
  const size_t lds_lq_multi = lq_elems * sizeof(T);
  // Rows per chunk of the row stage: the widest whose scratch lets a CU hold four instances of the fused trial kernel
  // (the headline batch is four instances per CU); the split row kernels get the same width.
  {
    const size_t fixed = rows_maps_bytes(d) + ((rollout_lds_elems(d.n, d.m) + 3) & ~size_t(3)) * sizeof(T) + 16;
    const size_t per_instance = size_t(160) * 1024 / 4;
    const size_t budget = per_instance > fixed ? (per_instance - fixed) / trial_row_waves(W) : 0;
    sa.rows_cw = rows_chunk_width(d.n, d.m, d.rp_pslots, d.rp_lslots, sizeof(T), budget);
  }
  auto k_trial = ilq_trial_kernel<T, NX, NP, MU, W>;
  const bool pw = C::USE_MFMA && !p->desc.params.open_loop;  // one wave per player (MFMA feedback sweep)
  // Compact rows (ilqg_common.hpp) between the row stage and the sweep: the one-tile player-parallel sweep and the
  // open-loop sweep read them; the other sweeps take the dense arrays.
  // ... and only when the [T][rp_compact_w] rows fit the space they are kept in: the dense Q, l, R, r arrays up to the
  // sweep's scratch rows (a small or densely coupled problem's row carries the A and B words too)
  const bool compact_on = d.rp_compact_w > 0 && size_t(d.T) * size_t(d.rp_compact_w) <= L.lqscr - L.Q &&
                          choice(opt.compact_rows, true);
  const bool ol_compact = p->desc.params.open_loop && compact_on;
  // A registered row-program structure (ilqg_rowprog_static.hpp): the row stage as straight-line code for it — in the
  // fused kernel (state rows in registers, i.e. without the chunk's (x, u) image in LDS, so it takes the 64-row chunk
  // where the interpreter's scratch would not fit four instances on a CU: n = 16 in fp64), in the split row kernel and
  // in the merit-only row kernel of the speculative line search.  Same results, bit for bit.
  int static_id = 0;    // the structure the split / probing row kernels are compiled for (0: they interpret)
  int static_prog = 0;  // ... and the fused kernel
  auto k_rows = ilq_rows_kernel<T, NX, NP, MU>;
  auto k_prows = ilq_probe_rows_kernel<T, NX, NP, MU>;
  // (the static kernels exist for solves on compact rows: they write nothing dense)
  const bool will_compact = (pw && C::MFMA_ONE_TILE && compact_on) || ol_compact;
  bool static_in_regs = false;  // ... with the slots of a chunk in registers (ilqg_rows.hpp: no row scratch in LDS)
  if (opt.static_rows != ILQG_CHOICE_OFF && will_compact) {
#define X(ID_, NX_, NP_, MU_)                                                                                  \
    if constexpr (NX_ == NX && NP_ == NP && MU_ == MU)                                                         \
      if (p->static_prog == ID_) {                                                                             \
        static_id = ID_;                                                                                       \
        static_in_regs = static_prog_in_registers<StaticRowProg<ID_>>();                                       \
        k_rows = ilq_rows_kernel<T, NX, NP, MU, ID_>;                                                          \
        k_prows = ilq_probe_rows_kernel<T, NX, NP, MU, ID_>;                                                   \
        if (rows_state_in_registers(NX, NP * MU) && trial_lds_bytes<T>(d, W, 64, true) <= size_t(160) * 1024 / 4) { \
          k_trial = ilq_trial_kernel<T, NX, NP, MU, W, ID_>;                                                   \
          static_prog = ID_;                                                                                   \
        }                                                                                                      \
      }
    ILQG_STATIC_PROGS(X)
#undef X
  }
  const int rows_cw_interpreted = sa.rows_cw;
  if (static_prog) sa.rows_cw = 64;
  const size_t lds_trial = trial_lds_bytes<T>(d, W, sa.rows_cw, static_prog != 0);
  // The schedules below are chosen by how many instances the chip will hold at once.  Under a mask that is the number
  // of instances taking part, not the length of the buffers: a receding-horizon loop replans the few dozen plans
  // still running of a batch of 2048 (src/receding_horizon_simulator.cpp:77), and the throughput forms chosen for 2048
  // — single-wave sweep, adjoint expected decrease, split trial pass — are latency forms three times slower for them
  // (config 5 as written: 400 us per sweep launch instead of 190).  A free-running solve waits for the device every
  // round anyway, so it counts its mask first (one more read-back per call); a fixed-iteration solve stays
  // asynchronous and keeps the buffer length.
  int sched_batch = batch;
  if (active && !opt.forced_steps && !opt.deterministic && !(fixed_iters > 0 && !al_mode)) {
    HIP_TRY(hipMemsetAsync(p->d_unfinished, 0, 4 * sizeof(int), stream));  // (whatever an earlier solve left there)
    hipLaunchKernelGGL(ilq_count_active_kernel, dim3(1), dim3(256), 0, stream, active, int(batch), p->d_unfinished);
    HIP_TRY(hipGetLastError());
    if (read_round_counters(p, stream) != ILQG_OK) return ILQG_ERR_HIP;
    sched_batch = p->h_unfinished[3] > 0 ? p->h_unfinished[3] : 1;
  }
  // fp32, one-tile sweep of three player waves, many instances per CU: the 128-register build (see ilq_lq_kernel)
  constexpr bool has_packed = sizeof(T) == 4 && C::USE_MFMA && C::MFMA_ONE_TILE && NP == 3;
  const bool packed = has_packed && pw && size_t(sched_batch) >= size_t(5) * 256;
  auto k_lq_multi = packed ? ilq_lq_kernel<T, NX, NP, MU, (has_packed ? LQ_PLAYER_WAVES_PACKED : LQ_VALU_FEEDBACK)>
            : pw ? ilq_lq_kernel<T, NX, NP, MU, (C::USE_MFMA ? LQ_PLAYER_WAVES : LQ_VALU_FEEDBACK)>
                 : (p->desc.params.open_loop ? (ol_compact ? ilq_lq_kernel<T, NX, NP, MU, LQ_OPEN_LOOP_COMPACT> : ilq_lq_kernel<T, NX, NP, MU, LQ_OPEN_LOOP>)
                                             : ilq_lq_kernel<T, NX, NP, MU, LQ_VALU_FEEDBACK>);
  const int nt_lq_multi = p->desc.params.open_loop ? OLCfg<T, NX, NP, MU>::NT : (pw ? 64 * NP : C::NT);
  raise_lds_limit((const void*)k_trial, lds_trial);

  // One round = trial kernel, then (for the instances that asked) the exit kernel and the LQ kernel.
  // The trial kernel counts what its instances wait for; with fixed_iters = K (no AL) the sequence is
  // known — trial, K x (LQ, trial), exit — otherwise the host reads the counts back each round, which
  // makes a free-running solve synchronous with respect to `stream`.
  auto k_exit = ilq_exit_kernel<T, NX, NP, MU>;
  const size_t lds_exit = quad_tables_bytes(d, sizeof(T)) + 64 * sizeof(T);
  // Split passes where the fused trial kernel's LDS leaves a CU with fewer than three instances (n = 24); the
  // host then counts every round, because an instance may ask for another pass (back-tracking) before its sweep.
  // ILQG_SPLIT_TRIAL=0/1 overrides the choice (A/B measurements).
  // Split passes: where the fused kernel cannot keep four instances on a CU, and for batches that are several times
  // what it keeps resident when the split integration kernel (a quarter of the registers, 4 KB of LDS) gains from
  // the co-residency — measured: n = 24, B = 4096: 65 k vs 39 k it/s.  For n <= 16 the fused kernel stays
  // (round 3, with compact rows: n = 14 fp32, B = 8192: 1.97 M fused vs 1.79 M split; fp64: 1.34 M vs 1.27 M).
  int num_cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cus = prop.multiProcessorCount;
  }
  // (the exit kernel copies an instance's final iterate, T m n words of strategies among them: four waves where the
  // instances are few and the kernel is a link of the round's chain, one where they share the chip's bandwidth anyway)
  const int nt_exit = sched_batch <= 2 * num_cus ? 256 : 64;
  const bool big_batch = size_t(sched_batch) >= size_t(8) * num_cus;
  // the whole batch resident at once (four instances per CU): fair issue arbitration among the co-resident instances
#ifndef ILQG_PRIO_ROTATION
#define ILQG_PRIO_ROTATION 1
#endif
  sa.prio_div = (ILQG_PRIO_ROTATION && sched_batch > num_cus && sched_batch <= 4 * num_cus) ? num_cus : 0;
  // ... and wherever the single-wave sweep (below) will run: it takes its expected decrease from its own adjoint pass, so
  // nothing is left for the fused kernel's row wave to overlap with the rollout, and the three split kernels each keep
  // more instances on a CU than the fused one (measured, n = 14, B = 8192, LQ single-wave + adjoint: fp64 1.58 M it/s
  // fused vs 1.65 M split, fp32 2.61 M vs 2.77 M; B = 2048 fp64 1.40 M vs 1.50 M).
  constexpr bool has_1w = W1Cfg<T, NX, NP, MU>::SUPPORTED && C::USE_MFMA && C::MFMA_ONE_TILE;
  const bool want_1w = has_1w && pw && compact_on && !kProfile && d.rp_compact_w <= W1Cfg<T, NX, NP, MU>::kWords &&
                       choice(opt.single_wave_sweep, !opt.deterministic && size_t(sched_batch) >= size_t(5) * num_cus) &&
                       opt.adjoint_expected_decrease != ILQG_CHOICE_OFF;
  bool split = choice(opt.split_trial, 4 * lds_trial > size_t(160) * 1024 || (big_batch && NX > 16) || want_1w);
  if (kProfile || opt.forced_steps) split = false;  // the phase profile reads the fused kernel's counters
  if (split) {  // the split row kernels interpret the program: their chunk width is the interpreter's
    static_prog = 0;
    sa.rows_cw = rows_cw_interpreted;
  }
  sa.compact = ((pw && C::MFMA_ONE_TILE && compact_on) || ol_compact) ? 1 : 0;
  const size_t split_maps_bytes = sa.compact ? 0 : rows_maps_bytes(d);  // the split row kernels' copy of the word maps
  if (split) {
    // The split row kernel is one wave per chunk with the chunk's scratch to itself, and its instances per CU are what
    // the scratch leaves room for: chunks of equal width (T = 100: 2 x 50 rows, 28 KB, five per CU — not 64 + 36 at 36 KB
    // and four), or one chunk more where that buys a second wave per SIMD.
    const int base = (d.T + sa.rows_cw - 1) / sa.rows_cw;
    int best_cw = sa.rows_cw;
    double best = 0.0;
    for (int chunks = base; chunks <= base + ILQG_SPLIT_ROW_EXTRA_CHUNKS && chunks <= d.T; chunks++) {
      const int cw = (d.T + chunks - 1) / chunks;
      // (a static row kernel's scratch has the fixed strides of a 64-row chunk whatever its width: ilqg_rows.hpp)
      const size_t lds = (static_id && static_in_regs) ? size_t(16) : split_maps_bytes + split_rows_elems(d, NX, static_id ? 64 : cw) * sizeof(T);
      size_t per_cu = size_t(160) * 1024 / (lds + 256);
      if (per_cu > 8) per_cu = 8;
      const double score = double(per_cu) / double(chunks);
      if (score > best * 1.05) {
        best = score;
        best_cw = cw;
      }
    }
    sa.rows_cw = best_cw;
  }
  const bool counted = !opt.forced_steps && (split || !(fixed_iters > 0 && !al_mode) || choice(opt.counted, false));
  // Hand-off: whenever the host counts rounds anyway, the fused kernel keeps an instance only until its line
  // search rejects a step; the back-tracking instances then go through split passes with the speculative line
  // search (ILQG_HANDOFF=0 keeps every pass in the fused kernel).
  const bool handoff = counted && !split && !kProfile && sa.prm.linesearch && choice(opt.handoff, true);
  const bool lists = split || handoff;
  // The sweep's forward pass runs in the fused trial kernel that follows it, beside the rollout, whenever that is the
  // kernel that follows (split passes and the open-loop sweep keep it in the sweep's kernel).
  // Compact rows between the row stage and the one-tile player-parallel sweep (ilqg_common.hpp): what the row stage
  // writes and the sweep reads per time step shrinks from N (n^2 + n) + ... words to the ones a cost term can touch.
  sa.defer_forward = (!split && !p->desc.params.open_loop) ? 1 : 0;
  {
    constexpr size_t fwd_elems = 4 * 2 * ((NX * NX + C::SCR + 3) & ~3) + 2 * NX + 8;
    if (trial_rows_elems(d, sa.rows_cw, static_prog != 0) < fwd_elems + 8) sa.defer_forward = 0;
  }
  // The throughput form of the one-tile feedback sweep — one wave per instance, twice the instances per CU
  // (ilqg_lq_feedback1w.hpp) — for batches of five or more instances per CU (measured, n = 14 fp64: B = 1024 1.47 M it/s
  // player-parallel vs 1.11 M single-wave; 1280: 1.09 vs 1.10; 1536: 1.17 vs 1.25; 2048: 1.20 vs 1.45; 8192: 1.45 vs
  // 1.68); it reads compact rows and leaves the forward pass to the trial kernel.
  // ilqg_solve_options::single_wave_sweep overrides the choice (same results to rounding).
  // Where the expected decrease of a single-wave sweep comes from: its own adjoint recursion, or — only possible with the
  // fused trial kernel, whose row wave runs it — the deferred forward pass over the sweep's scratch rows.
  const bool adjoint = choice(opt.adjoint_expected_decrease, sa.defer_forward == 0);
  const bool single_wave = has_1w && pw && sa.compact && !kProfile && (adjoint || sa.defer_forward) &&
                           d.rp_compact_w <= W1Cfg<T, NX, NP, MU>::kWords &&
                           choice(opt.single_wave_sweep, !opt.deterministic && size_t(sched_batch) >= size_t(5) * num_cus);
  auto k_lq = single_wave ? ilq_lq_kernel<T, NX, NP, MU, (has_1w ? LQ_SINGLE_WAVE : LQ_VALU_FEEDBACK)> : k_lq_multi;
  const int nt_lq = single_wave ? 64 : nt_lq_multi;
  const size_t lds_lq = single_wave ? size_t(W1Cfg<T, NX, NP, MU>::ELEMS + 4) * sizeof(T) : lds_lq_multi;
  if (single_wave) {
    sa.prio_div = 0;
    if (adjoint) sa.defer_forward = 0;  // the sweep forms the expected decrease itself: no forward pass anywhere
  }
  raise_lds_limit((const void*)k_lq, lds_lq);
  p->last_schedule = (single_wave ? ILQG_SCHEDULE_SINGLE_WAVE_SWEEP : 0) | ((single_wave && adjoint) ? ILQG_SCHEDULE_ADJOINT_DECREASE : 0) |
                     (split ? ILQG_SCHEDULE_SPLIT_TRIAL : 0) | (sa.compact ? ILQG_SCHEDULE_COMPACT_ROWS : 0) |
                     (counted ? ILQG_SCHEDULE_COUNTED : 0) | (p->desc.params.open_loop ? ILQG_SCHEDULE_OPEN_LOOP : 0) |
                     ((static_prog || (static_id && split)) ? ILQG_SCHEDULE_STATIC_ROWS : 0);
  long long cap = al_mode ? (long long)(sa.prm.max_solver_iters + 1) * (sa.prm.unconstrained_solver_max_iters + 2)
                          : (long long)sa.prm.max_solver_iters + 2;
  if (split || counted) cap = (cap + 2) * ((long long)sa.prm.max_backtracking_steps + 3);
  auto k_roll = ilq_roll_kernel<T, NX, NP, MU>;
  auto k_decide = ilq_decide_kernel<T, NX, NP, MU>;
  const size_t lds_roll = trial_phase_lds_bytes<T>(d, TRIAL_ROLL, sa.rows_cw),
               lds_decide = trial_phase_lds_bytes<T>(d, TRIAL_DECIDE, sa.rows_cw);
  // (a static row kernel whose slots are registers needs no scratch beyond the (x, u) image of shapes past 16 states)
  const size_t static_regs_lds = (rows_state_in_registers(NX, NP * MU) ? 0 : size_t(d.n + d.m) * 64) * sizeof(T) + 16;
  const size_t lds_rows = (static_id && static_in_regs) ? static_regs_lds
                                                        : split_maps_bytes + split_rows_elems(d, NX, static_id ? 64 : sa.rows_cw) * sizeof(T);
  const size_t lds_prows = (static_id && static_in_regs) ? static_regs_lds
                                                         : split_maps_bytes + probe_rows_elems(d, NX, static_id ? 64 : sa.rows_cw) * sizeof(T);  // merit only
  constexpr bool pairs = rollout_pairs(NX, NP, MU);  // two rollouts per wavefront (ilqg_stages.hpp)
  const size_t lds_proll = pairs ? size_t(rollout_pair_lds_elems(d.n, d.m)) * sizeof(T) + 16 : lds_roll;
  const bool probe = lists && sa.prm.linesearch && choice(opt.probe, true);
  auto k_proll = ilq_probe_roll_kernel<T, NX, NP, MU>;
  auto k_proll_fat = ilq_probe_roll_kernel<T, NX, NP, MU, sizeof(T) == 8>;  // (fp32: the same kernel)
  auto k_proll_single = ilq_probe_roll_kernel<T, NX, NP, MU, sizeof(T) == 8, true>;
  auto k_proll_lanes = ilq_probe_roll_lanes_kernel<T, NX, NP, MU>;
  const size_t lds_proll_lanes = pairs ? size_t(rollout_lanes_lds_elems(d.n, d.m, d.N)) * sizeof(T) + 16 : 0;
  const int probe_lanes_min = opt.probe_lanes == ILQG_CHOICE_OFF ? (1 << 30) : (opt.probe_lanes == ILQG_CHOICE_ON ? 2 : 8);
  const int decide_elems = int(trial_phase_quad_elems<T>(d, TRIAL_DECIDE, sa.rows_cw));
  const int row_chunks = (d.T + sa.rows_cw - 1) / sa.rows_cw;  // workgroups per instance of the row kernels
  int round_instances = batch, list = 0;  // split passes: how many instances this round covers, which list is free
  int tail_rounds = 0;                    // rounds since the whole batch was last in one
  bool probed_in_tail = false;            // the current tail has launched a probing pass
  bool deep_tails = false;                // a tail of this solve kept half of its list through three rounds (see the ramp below)
  int tail_first_instances = 0;
  if (lists) {
    raise_lds_limit((const void*)k_roll, lds_proll);
    raise_lds_limit((const void*)k_rows, lds_rows);
    raise_lds_limit((const void*)k_decide, lds_decide);
    raise_lds_limit((const void*)k_proll, lds_proll);
    raise_lds_limit((const void*)k_proll_fat, lds_proll);
    raise_lds_limit((const void*)k_proll_single, lds_roll);
    raise_lds_limit((const void*)k_proll_lanes, lds_proll_lanes);
    raise_lds_limit((const void*)k_prows, lds_prows);
  }
  sa.clear_counters = 1;
  sa.first = resume ? 2 : 1;
  p->counters_clean = false;  // whatever an earlier solve left in the round counters
  int waiting_lq = 0, waiting_exit = 0;  // split passes: instances already through this iteration's line search
  bool exit_pending = false;             // a burst round went without its exit launch (see the burst loop)
  // Free-running solves: the kernels select their instances by the stage each one is in, so a round launched for
  // nobody is harmless — the host therefore enqueues BURSTS of whole rounds (trial, exit, sweep) and reads the
  // counters back once per burst instead of once per round (a read-back is a stream synchronisation: ~20-30 us against
  // a ~0.45 ms round of a single instance).  The burst doubles up to eight rounds while no instance is back-tracking
  // and falls back to one as soon as one is (those go through the probing passes, which need the lists every round).
  // the anytime exit (ilqg_solve_options::max_runtime): the host's clock is read where the batch is about to start an
  // iteration, so rounds are not batched into bursts; the augmented-Lagrangian solver gives each inner solve of a
  // constrained problem max_runtime / max_solver_iters (src/augmented_lagrangian_solver.cpp:85-88)
  const bool timed = opt.max_runtime > 0.0 && counted;
  const double budget = !timed ? 0.0 : ((al_mode && d.num_constraints > 0) ? opt.max_runtime / double(sa.prm.max_solver_iters) : opt.max_runtime);
  auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double inner_elapsed = 0.0, tic = 0.0;
  bool iteration_open = false;
  AlOuterClock outer(p, timed && al_mode && d.num_constraints > 0, opt.max_runtime, budget);
  const bool bursts = counted && !kProfile && !timed && choice(opt.round_bursts, true);
  int burst = 1;
  // the iterate log (ilqg_solve_options::iterate_log): copied in front of every exit / sweep launch
  IterLog<T> lg{};
  const bool logging = opt.iterate_log != nullptr;
  if (logging) {
    const ilqg_iterate_log& il = *opt.iterate_log;
    if (!il.xs || !il.us || !il.costs || !il.count || il.capacity < 1)
      return fail(ILQG_ERR_INVALID, "iterate log: xs, us, costs, count and a capacity of at least one are required");
    lg = IterLog<T>{(T*)il.xs, (T*)il.us, (T*)il.costs, (T*)il.P, (T*)il.alpha, il.count, il.capacity};
    HIP_TRY(hipMemsetAsync(il.count, 0, sizeof(int) * size_t(batch), stream));
  }
  auto log_iterates = [&]() -> ilqg_status {
    if (!logging) return ILQG_OK;
    hipLaunchKernelGGL(ilq_log_kernel<T>, dim3(batch), dim3(256), 0, stream, d, sa, lg);
    HIP_TRY(hipGetLastError());
    return ILQG_OK;
  };
  for (long long round = 0;; round++) {
    if (bursts && !sa.ids) {
      // a fixed-iteration solve knows its last round: no burst runs past it
      const long long left = (fixed_iters > 0 && !al_mode) ? (long long)fixed_iters - round : (long long)burst;
      for (int q = 1; q < burst && q <= left; q++) {  // rounds without a read-back
        if (!p->counters_clean) HIP_TRY(hipMemsetAsync(p->d_unfinished, 0, 4 * sizeof(int), stream));
        p->counters_clean = false;
        if (split) {  // the three-kernel form of the pass over the whole batch (no probing: nobody is listed)
          sa.ids_next = pass_ids + size_t(list) * batch;
          sa.round_count = batch;
          hipLaunchKernelGGL(k_roll, dim3(pairs ? (batch + 1) / 2 : batch), dim3(64), lds_proll, stream, d, sa);
          HIP_TRY(hipGetLastError());
          sa.first = 0;
          hipLaunchKernelGGL(k_rows, dim3(row_chunks, batch), dim3(64), lds_rows, stream, d, sa);
          HIP_TRY(hipGetLastError());
          hipLaunchKernelGGL(k_decide, dim3(batch), dim3(64), lds_decide, stream, d, sa);
        } else {
          sa.ids_next = handoff ? pass_ids + size_t(list) * batch : nullptr;
          hipLaunchKernelGGL(k_trial, dim3(batch), dim3(64 * W), lds_trial, stream, d, sa);
        }
        HIP_TRY(hipGetLastError());
        sa.first = 0;
        if (log_iterates() != ILQG_OK) return ILQG_ERR_HIP;
        // The exit path inside a burst only where it starts something (the augmented Lagrangian's next inner solve);
        // an ILQSolver::Solve that ends here waits for the burst's last round, whose exit launch is then unconditional
        // (a launch for nobody is ~5 us of every round of a lone instance).
        if (al_mode) {
          hipLaunchKernelGGL(k_exit, dim3(batch), dim3(nt_exit), lds_exit, stream, d, sa);
          HIP_TRY(hipGetLastError());
        } else {
          exit_pending = true;
        }
        hipLaunchKernelGGL(k_lq, dim3(batch), dim3(nt_lq), lds_lq, stream, d, sa);
        HIP_TRY(hipGetLastError());
        p->counters_clean = true;  // (SolveArgs::clear_counters)
        round++;
      }
    }
    if (counted && !p->counters_clean) HIP_TRY(hipMemsetAsync(p->d_unfinished, 0, 4 * sizeof(int), stream));
    p->counters_clean = false;  // (the round's kernels count into them)
    if (split || sa.ids) {  // a split pass: the whole batch (split mode) or the listed back-tracking instances
      sa.ids_next = pass_ids + size_t(list) * batch;
      // Step sizes probed per listed instance: as many as the pool holds for a list this long, doubling from two
      // over the first rounds of a tail (most line searches that back-track at all end within a step or two; the
      // ones that do not are mostly on their way through all max_backtracking_steps of a failing search, and a round
      // for the few instances left costs the latency of its launches whatever it probes: 2, 4, 8, 16, 32, ...).
      int probe_k = sa.ids ? tail.pool_entries / round_instances : 0;
      if (probe_k > kProbeCandidates) probe_k = kProbeCandidates;
      {
        // the ramp: `first` candidates in a tail's first round, doubling per round (ilqg_solve_options::probe_first)
        // The library's choice: as many candidates per instance as keep the round's rollouts within one filling of
        // the chip (kProbeRoundBudget of them), between 2 and kProbeCandidates.  A short list is a few deep searches — the instances
        // that back-track at all mostly go on for tens of steps (the n = 16 intersection: ~10 % of the batch, mean
        // depth ~30) — and 32 at once ends them in a round or two (measured, B = 1024: 280 k -> 335 k it/s); a long list
        // (config 4: ~40 % of 4096 instances, most done within a step or two) pays for every candidate it does not
        // need (226 k it/s at 2, 193 k at 32).  Round 4, with two rollouts per wavefront and the gradient-only row pass: a
        // budget of 8192 rollouts (config 5 / n = 16 constrained / config 4: 249 k / 483 k / 248 k it/s; 4096: 240 / 469 /
        // 252; 16384: 261 / 483 / 236; 4096 growing fourfold per round: 245 / 457 / 251).
        int first = opt.probe_first;
        if (first <= 0) {
          first = kProbeRoundBudget / (round_instances > 0 ? round_instances : 1);
          first = first < 2 ? 2 : (first > kProbeCandidates ? kProbeCandidates : first);
        }
        long long ramp = (long long)first << (tail_rounds < 8 ? tail_rounds : 8);
        if (ramp < 2) ramp = 2;
        // Deep searches: once a tail of this solve has kept half of its list through three rounds (2 + 4 + 8 or more
        // rejected candidates each: they are mostly on their way through all max_backtracking_steps), later tails skip the
        // ramp.  A probing round costs one wave's chain per time step until its waves fill the chip, so every candidate up
        // to that point is free: with the lane form (rollout_lanes: 64 / N candidates per wavefront) that is
        // C floor(2400 / instances) candidates per instance — whole wavefronts —, as far as the pool holds them.
        // (config 5's scene: rounds of 4, 10, 20, 21, 42 ... candidates -> 16 - 63 from a tail's first round on;
        // config 4's ~1600 back-tracking instances are mostly done after a step or two: they keep the ramp.)
        constexpr int C_lane = rollout_lanes_per_wave(NP > 0 ? NP : 1);
        if (sa.ids && tail_rounds == 0) tail_first_instances = round_instances;
        if (sa.ids && tail_rounds == 3 && 2 * round_instances >= tail_first_instances) deep_tails = true;
        if (sa.ids && deep_tails && opt.probe_first <= 0) {
          int k_deep = probe_k;  // what the pool holds
          if (pairs && opt.probe_lanes != ILQG_CHOICE_OFF) {
            const int free_waves = 2400 / round_instances;
            const int k_lane = C_lane * (free_waves > 1 ? free_waves : 1);
            if (k_deep > k_lane) k_deep = k_lane;
            if (k_deep >= C_lane) k_deep = k_deep / C_lane * C_lane;
          }
          if (ramp < k_deep) ramp = k_deep;
        }
        if (probe_k > ramp) probe_k = int(ramp);
      }
#ifdef ILQG_DEBUG_PROBE
      if (sa.ids) fprintf(stderr, "tail %d instances %d probe_k %d deep %d\n", tail_rounds, round_instances, probe_k, int(deep_tails));
#endif
      if (sa.ids) tail_rounds++;
      // see generic_solve: instances in ST_PROBE are only picked up again by a probing launch
      if (probe && sa.ids && probed_in_tail && probe_k < 2) return fail(ILQG_ERR_HIP, "a probing line-search tail lost its probe launch");
      if (probe && probe_k >= 2) {
        probed_in_tail = true;
        // the listed instances' next step sizes side by side; their states move to the first acceptable one
        sa.probe_pool = probe_pool;
        sa.probe_k = probe_k;
        const int proll_y = pairs ? (probe_k + 1) / 2 : probe_k;
        // the register-rich build while every rollout of the round is resident at once at two waves per SIMD
        const bool fat = (long long)round_instances * proll_y <= 8ll * num_cus;
        // A lane per (candidate, subsystem) where that is the shorter round.  A round of W waves takes about
        // max(one wave's chain, W / SIMDs x a wave's issue time) per time step; measured on the n = 15 scene (cycles per
        // step): the paired form ~1500 / ~550, the lane form — thirteen trigonometric evaluations in sequence,
        // ~1180 instructions — ~3500 / ~1530, for 64 / N candidates instead of two.  So the lane form wins once its
        // own waves fill the chip (config 5's scene from 16 candidates per instance on), and loses a round of a few
        // deep searches (n = 16: ~100 instances x 128 candidates are 700 lane waves, one chain long).
        constexpr int C = rollout_lanes_per_wave(NP > 0 ? NP : 1);
        const double simds = 4.0 * num_cus;
        const double w_pair = double(round_instances) * proll_y, w_lane = double(round_instances) * ((probe_k + C - 1) / C);
        const double t_pair = std::max(1500.0, w_pair / simds * 550.0), t_lane = std::max(3500.0, w_lane / simds * 1530.0);
        // (n > 16: the lane form's step carries a 2 n-term control product per lane and, for the six-state cars, spills
        // inside the time loop — n = 24 on the feedback sweep: 2.2 ms per launch against 0.4-0.6 for the paired form,
        // 99 k -> 83 k it/s; the model's constants are the n = 15 scene's, so AUTO leaves those shapes on pairs.)
        const bool lanes = pairs && probe_k >= probe_lanes_min &&
                           (opt.probe_lanes == ILQG_CHOICE_ON || (NX <= 16 && t_lane < 0.9 * t_pair));
        // a candidate per wavefront while every one of them finds a SIMD of its own
        const bool single = pairs && !lanes && (long long)round_instances * probe_k <= 4ll * num_cus &&
                            opt.probe_lanes != ILQG_CHOICE_ON;
        if (lanes) {
          hipLaunchKernelGGL(k_proll_lanes, dim3(round_instances, (probe_k + C - 1) / C), dim3(64), lds_proll_lanes, stream, d, sa);
        } else if (single) {
          hipLaunchKernelGGL(k_proll_single, dim3(round_instances, probe_k), dim3(64), lds_roll, stream, d, sa);
        } else {
          hipLaunchKernelGGL(fat ? k_proll_fat : k_proll, dim3(round_instances, proll_y), dim3(64), lds_proll, stream, d, sa);
        }
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_prows, dim3(row_chunks, round_instances * probe_k),
                           dim3(64), lds_prows, stream, d, sa);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(ilq_probe_pick_kernel<T>, dim3(round_instances), dim3(kProbeCandidates), 0, stream, d, sa);
        HIP_TRY(hipGetLastError());
      }
      sa.round_count = round_instances;
      hipLaunchKernelGGL(k_roll, dim3(pairs ? (round_instances + 1) / 2 : round_instances), dim3(64), lds_proll, stream, d, sa);
      HIP_TRY(hipGetLastError());
      sa.first = 0;
      hipLaunchKernelGGL(k_rows, dim3(row_chunks, round_instances), dim3(64),
                         lds_rows, stream, d, sa);
      HIP_TRY(hipGetLastError());
      hipLaunchKernelGGL(k_decide, dim3(round_instances), dim3(64), lds_decide, stream, d, sa);
    } else {
      sa.ids_next = handoff ? pass_ids + size_t(list) * batch : nullptr;
      hipLaunchKernelGGL(k_trial, dim3(batch), dim3(64 * W), lds_trial, stream, d, sa);
    }
    HIP_TRY(hipGetLastError());
    sa.first = 0;
    int want_lq = 1, want_exit = 0, restarted = 0, again = 0;
    if (counted) {
      if (read_round_counters(p, stream) != ILQG_OK) return ILQG_ERR_HIP;
      want_lq = p->h_unfinished[0];
      want_exit = p->h_unfinished[1];
      again = p->h_unfinished[3];
      if (lists) {
        // keep the batch in step: the sweep is launched once per iteration, when the last back-tracking instance
        // has made up its mind (a sweep launch with a handful of instances costs a full sweep's latency)
        waiting_lq += want_lq;
        waiting_exit += want_exit;
        if (again) {  // the next round covers the listed instances only
          burst = 1;
          if (round > cap) return fail(ILQG_ERR_HIP, "solve did not terminate within its iteration bound");
          sa.ids = sa.ids_next;
          round_instances = again;
          list ^= 1;
          continue;
        }
        sa.ids = nullptr;
        round_instances = batch;
        tail_rounds = 0;
        probed_in_tail = false;
        want_lq = waiting_lq;
        want_exit = waiting_exit;
        waiting_lq = waiting_exit = 0;
      }
      if (bursts) burst = again ? 1 : (burst < 8 ? burst * 2 : 8);
    } else if (round == fixed_iters) {
      want_lq = 0;
      want_exit = 1;
    }
    if (exit_pending) {  // instances that ended in a burst round without an exit launch
      want_exit = 1;
      exit_pending = false;
    }
    if (timed && want_lq) {
      // the loop condition of src/ilq_solver.cpp:123-124 for the iteration the batch is about to start
      const double now = wall();
      if (iteration_open) {
        p->loop_add(now - tic);
        inner_elapsed += now - tic;
        iteration_open = false;
      }
      if (!(inner_elapsed < budget - p->loop_upper_bound())) {
        hipLaunchKernelGGL(ilq_deadline_kernel<T>, dim3(batch), dim3(64), 0, stream, d, sa);
        HIP_TRY(hipGetLastError());
        want_exit = 1;
        want_lq = 0;
      }
    }
    if ((want_exit || want_lq) && log_iterates() != ILQG_OK) return ILQG_ERR_HIP;
    if (want_exit) {
      if (outer.before_exit()) sa.outer_closed = 1;  // out of time: inner solves that end now are the last ones
      hipLaunchKernelGGL(k_exit, dim3(batch), dim3(nt_exit), lds_exit, stream, d, sa);
      HIP_TRY(hipGetLastError());
      p->counters_clean = false;
      if (counted && al_mode) {
        if (read_round_counters(p, stream) != ILQG_OK) return ILQG_ERR_HIP;
        restarted = p->h_unfinished[2];
        if (restarted) inner_elapsed = 0.0;  // the next inner solve's own budget
        outer.after_exit(restarted);
      }
    }
    if (want_lq) {
      if (timed) {
        tic = wall();
        iteration_open = true;
      }
      hipLaunchKernelGGL(k_lq, dim3(batch), dim3(nt_lq), lds_lq, stream, d, sa);
      HIP_TRY(hipGetLastError());
      p->counters_clean = true;  // (SolveArgs::clear_counters)
    }
    if (!want_lq && !restarted) break;
    if (round > cap) return fail(ILQG_ERR_HIP, "solve did not terminate within its iteration bound");
  }
  return ILQG_OK;
}

#if defined(ILQG_PART_NX)
// instantiation unit of a split build: this (n, N, m_i) in both precisions, nothing else
template struct DimsLaunch<float, ILQG_PART_NX, ILQG_PART_NP, ILQG_PART_MU>;
template struct DimsLaunch<double, ILQG_PART_NX, ILQG_PART_NP, ILQG_PART_MU>;
#else
#if defined(ILQG_SPLIT_BUILD)
#define X(NX_, NP_, MU_)                                   \
  extern template struct DimsLaunch<float, NX_, NP_, MU_>; \
  extern template struct DimsLaunch<double, NX_, NP_, MU_>;
ILQG_FOR_DIMS(X)
#undef X
#endif

static ilqg_status launch_linquad(const ilqg_problem* p, int32_t batch, const void* xs, const void* us,
                                  const void* lambdas, const void* mu, const int32_t* t_extreme, void* A, void* Bm,
                                  void* Q, void* l, void* R, void* r, void* merit_part, void* cost_part,
                                  const int32_t* active, void* stream) {
  const DevProblem& d = p->dev;
  const void* const ptrs[14] = {xs, us, lambdas, mu, t_extreme, A, Bm, Q, l, R, r, merit_part, cost_part, active};
  if (p->generic)  // the row stage with run-time dimensions (rows_chunk<T, 0, 0, 0>)
    return p->desc.dtype == ILQG_F32 ? DimsLaunch<float, 0, 0, 0>::rows(d, batch, ptrs, (hipStream_t)stream)
                                     : DimsLaunch<double, 0, 0, 0>::rows(d, batch, ptrs, (hipStream_t)stream);
#define X(NX_, NP_, MU_)                                                                          \
  if (d.n == NX_ && d.N == NP_ && p->mu_uniform == MU_)                                           \
    return p->desc.dtype == ILQG_F32                                                              \
               ? DimsLaunch<float, NX_, NP_, MU_>::rows(d, batch, ptrs, (hipStream_t)stream)      \
               : DimsLaunch<double, NX_, NP_, MU_>::rows(d, batch, ptrs, (hipStream_t)stream);
  ILQG_FOR_DIMS(X)
#undef X
  return fail(ILQG_ERR_UNSUPPORTED, "no device kernel instantiated for this problem's dimensions");
}

// The smallest instantiated shape a game embeds in (same player count, at least its states and its widest control), for
// the padded sweeps (padded_lq_kernel / padded_lq_batch_kernel); false: none.
static bool pick_padded_shape(int n, int N, const int32_t* udim, int* nx, int* mu) {
  int mumax = 0, best_nx = 0, best_mu = 0;
  for (int i = 0; i < N; i++) mumax = udim[i] > mumax ? udim[i] : mumax;
#define X(NX_, NP_, MU_)                                                                                                 \
  if (NP_ == N && NX_ >= n && MU_ >= mumax && (best_nx == 0 || NX_ < best_nx || (NX_ == best_nx && MU_ < best_mu))) { \
    best_nx = NX_;                                                                                                       \
    best_mu = MU_;                                                                                                       \
  }
  ILQG_FOR_DIMS(X)
#undef X
  *nx = best_nx;
  *mu = best_mu;
  return best_nx != 0;
}

// The game's control blocks at mu x mu each (the padded shape's PairTable)
static bool padded_pairs(const PairTable& pt, int N, int mu, PairTable* ptp, std::string* err) {
  std::vector<ilqg_pair> pairs(pt.npairs);
  std::vector<int> udim_p(kMaxPlayers, mu);
  for (int q = 0; q < pt.npairs; q++) pairs[q] = {pt.pi[q], pt.pj[q]};
  if (!build_pairs(pairs.data(), pt.npairs, udim_p.data(), N, ptp, err)) return false;
  for (int q = 0; q < pt.npairs; q++) ptp->from_cost[q] = pt.from_cost[q];
  return true;
}

// Threads of a workgroup of the run-time-dimensioned sweeps (ilqg_lq_generic.hpp: every phase strides its entries over
// the workgroup and ends with a barrier).  -DILQG_GEN_THREADS_SMALL=n: the size for games whose phases have at most 160
// entries (A/B measurements).
#ifndef ILQG_GEN_THREADS_SMALL
#define ILQG_GEN_THREADS_SMALL 256
#endif
static inline int generic_sweep_threads(int n, int m) { return n * (n + m) <= 160 ? ILQG_GEN_THREADS_SMALL : 256; }

// LDS a CU can give one workgroup (gfx950: 160 KB)
static constexpr size_t kLdsPerWorkgroup = size_t(160) * 1024;

// The whole solve on the run-time-dimensioned kernels: the split trial pass (integrate / rows / decide, ilqg_solve.hpp)
// over the whole batch in a round without back-tracking instances, the host counting rounds; the sweeps of
// ilqg_lq_generic.hpp on dense rows.  Instances whose step is rejected are listed and their next step sizes probed side
// by side (round 5: the speculative line search of DimsLaunch::solve, bit-identical with probe = OFF).  No fused trial
// kernel, no compact rows: the path of every shape without a specialised instantiation.
template <typename T>
static ilqg_status generic_solve(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us, void* P, void* alpha,
                                 void* total_costs, int32_t* iters, int32_t* status, int32_t* converged, void* workspace,
                                 const ilqg_solve_options& opt, hipStream_t stream) {
  const DevProblem& d = p->dev;
  const int al_mode = opt.augmented_lagrangian ? 1 : 0, resume = opt.resume ? 1 : 0;
  const int ol_row = p->desc.params.open_loop ? ol_row_elems(d.n, d.m, d.N) : 0;
  const WsLayout L(d.n, d.m, d.N, d.T, d.pairs.Rsz, d.pairs.rsz, ol_row, d.num_constraints, al_mode);
  SolveArgs<T> sa{};
  sa.ol_row = ol_row;
  sa.al_mode = al_mode;
  sa.x0 = (const T*)x0; sa.xs = (T*)xs; sa.us = (T*)us; sa.P = (T*)P; sa.alpha = (T*)alpha;
  sa.total_costs = (T*)total_costs; sa.iters = iters; sa.status = status; sa.converged = converged;
  sa.ws = (T*)workspace; sa.ws_stride = L.total; sa.fixed_iters = opt.fixed_iters; sa.batch = batch;
  sa.prm = p->desc.params;
  sa.active = opt.active;
  sa.prof = nullptr;
  sa.forced_steps = (const T*)opt.forced_steps;
  sa.unfinished = p->d_unfinished;
  p->last_schedule = ILQG_SCHEDULE_GENERIC | ILQG_SCHEDULE_SPLIT_TRIAL | ILQG_SCHEDULE_COUNTED |
                     (p->desc.params.open_loop ? ILQG_SCHEDULE_OPEN_LOOP : 0);
  sa.rows_cw = rows_chunk_width(d.n, d.m, d.rp_pslots, d.rp_lslots, sizeof(T), size_t(48) * 1024);
  const size_t lds_roll = trial_phase_lds_bytes<T>(d, TRIAL_ROLL, sa.rows_cw),
               lds_decide = trial_phase_lds_bytes<T>(d, TRIAL_DECIDE, sa.rows_cw);
  const size_t lds_rows = rows_maps_bytes(d) + trial_rows_elems(d, sa.rows_cw) * sizeof(T);
  const size_t lds_exit = quad_tables_bytes(d, sizeof(T)) + 64 * sizeof(T);
  const size_t lds_lq = ((p->desc.params.open_loop ? gen_openloop_lds_elems(d.n, d.N, d.m) : gen_feedback_lds_elems(d.n, d.N, d.m)) + 4) * sizeof(T);
  if (lds_lq > kLdsPerWorkgroup || lds_rows > kLdsPerWorkgroup)
    return fail(ILQG_ERR_UNSUPPORTED, "the game does not fit a CU's LDS");
  auto k_roll = ilq_roll_kernel<T, 0, 0, 0>;
  auto k_rows = ilq_rows_kernel<T, 0, 0, 0>;
  auto k_decide = ilq_decide_kernel<T, 0, 0, 0>;
  auto k_exit = ilq_exit_kernel<T, 0, 0, 0>;
  auto k_lq = gen_lq_kernel<T>;
  // Back-tracking instances are listed and their next step sizes probed side by side, as in the specialised solve
  // (DimsLaunch::solve): without it a single failing line search of 100 steps costs the whole batch 100 serial passes
  // (round 5: mixed_dubins_car_scene, B = 1024: 76 passes per iteration, 42 k it/s).
  auto k_proll = ilq_probe_roll_kernel<T, 0, 0, 0>;
  auto k_prows = ilq_probe_rows_kernel<T, 0, 0, 0>;
  const size_t lds_prows = rows_maps_bytes(d) + probe_rows_elems(d, 0, sa.rows_cw) * sizeof(T);
  const WsTail tail = ws_tail(d, batch, sizeof(T), ol_row);
  int* const pass_ids = reinterpret_cast<int*>(static_cast<char*>(workspace) + tail.ids_off);
  T* const probe_pool = reinterpret_cast<T*>(static_cast<char*>(workspace) + tail.pool_off);
  const bool probe = !opt.forced_steps && sa.prm.linesearch && opt.probe != ILQG_CHOICE_OFF;
  raise_lds_limit((const void*)k_proll, lds_roll);
  raise_lds_limit((const void*)k_prows, lds_prows);
  raise_lds_limit((const void*)k_roll, lds_roll);
  raise_lds_limit((const void*)k_rows, lds_rows);
  raise_lds_limit((const void*)k_decide, lds_decide);
  raise_lds_limit((const void*)k_exit, lds_exit);
  raise_lds_limit((const void*)k_lq, lds_lq);
  // The sweep on a specialised kernel: the smallest instantiated shape the game embeds in (padded_lq_kernel).  AUTO: for
  // problems that have no instantiation of their own (ilqg_problem::generic); a problem sent here by
  // ilqg_solve_options::generic_kernels keeps the run-time-dimensioned sweeps unless padded_sweep = ON.
  int pad_nx = 0, pad_mu = 0;
  PairTable ptp;
  void* pad_buf = nullptr;
  auto padded_launch = [&](size_t* elems_out) -> ilqg_status {
#define X(NX_, NP_, MU_)                                                                                                      \
    if (pad_nx == NX_ && d.N == NP_ && pad_mu == MU_)                                                                          \
      return DimsLaunch<T, NX_, NP_, MU_>::lq_padded(d, &sa, ptp, pad_buf, elems_out, p->desc.params.open_loop != 0, stream);
    ILQG_FOR_DIMS(X)
#undef X
    return fail(ILQG_ERR_UNSUPPORTED, "no shape to embed the game in");
  };
  if (opt.padded_sweep == ILQG_CHOICE_ON || (opt.padded_sweep == ILQG_CHOICE_AUTO && p->generic)) {
    pick_padded_shape(d.n, d.N, d.udim, &pad_nx, &pad_mu);
    if (pad_nx == 0 && opt.padded_sweep == ILQG_CHOICE_ON)
      return fail(ILQG_ERR_UNSUPPORTED, "padded_sweep = ON: no instantiated shape holds this game (same player count, at "
                                        "least its states and its widest control)");
    if (pad_nx) {
      std::string err;
      if (!padded_pairs(d.pairs, d.N, pad_mu, &ptp, &err)) return fail(ILQG_ERR_INVALID, err);
      size_t elems = 0;
      ilqg_status s = padded_launch(&elems);
      if (s != ILQG_OK) return s;
      s = Scratch().reserve(size_t(batch) * elems * sizeof(T));
      if (s != ILQG_OK) return s;
      pad_buf = ilqg_shared::scratch_state().ptr;
      p->last_schedule |= ILQG_SCHEDULE_PADDED_SWEEP;
    }
  }
  const int row_chunks = (d.T + sa.rows_cw - 1) / sa.rows_cw;
  long long cap = al_mode ? (long long)(sa.prm.max_solver_iters + 1) * (sa.prm.unconstrained_solver_max_iters + 2)
                          : (long long)(opt.fixed_iters > 0 ? opt.fixed_iters : sa.prm.max_solver_iters) + 2;
  cap = (cap + 2) * ((long long)sa.prm.max_backtracking_steps + 3);
  const bool timed = opt.max_runtime > 0.0;
  const double budget = !timed ? 0.0 : ((al_mode && d.num_constraints > 0) ? opt.max_runtime / double(sa.prm.max_solver_iters) : opt.max_runtime);
  auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double inner_elapsed = 0.0, tic = 0.0;
  bool iteration_open = false;
  AlOuterClock outer(p, timed && al_mode && d.num_constraints > 0, opt.max_runtime, budget);
  IterLog<T> lg{};
  const bool logging = opt.iterate_log != nullptr;
  if (logging) {
    const ilqg_iterate_log& il = *opt.iterate_log;
    if (!il.xs || !il.us || !il.costs || !il.count || il.capacity < 1)
      return fail(ILQG_ERR_INVALID, "iterate log: xs, us, costs, count and a capacity of at least one are required");
    lg = IterLog<T>{(T*)il.xs, (T*)il.us, (T*)il.costs, (T*)il.P, (T*)il.alpha, il.count, il.capacity};
    HIP_TRY(hipMemsetAsync(il.count, 0, sizeof(int) * size_t(batch), stream));
  }
  sa.first = resume ? 2 : 1;
  p->counters_clean = false;  // whatever an earlier solve left in the round counters
  sa.ids = nullptr;
  sa.ids_next = nullptr;
  int waiting_lq = 0, waiting_exit = 0;
  int round_instances = batch, list = 0, tail_rounds = 0;
  bool probed_in_tail = false;
  for (long long round = 0;; round++) {
    if (!p->counters_clean) HIP_TRY(hipMemsetAsync(p->d_unfinished, 0, 4 * sizeof(int), stream));
    p->counters_clean = false;
    sa.ids_next = pass_ids + size_t(list) * batch;
    if (sa.ids && probe) {
      // step sizes probed per listed instance this round: what the pool holds for a list this long, ramping up over the
      // rounds of a tail from a budget of kProbeRoundBudget rollouts (the rule of DimsLaunch::solve)
      int probe_k = tail.pool_entries / round_instances;
      if (probe_k > kProbeCandidates) probe_k = kProbeCandidates;
      int first = opt.probe_first;
      if (first <= 0) {
        first = kProbeRoundBudget / round_instances;
        first = first < 2 ? 2 : (first > kProbeCandidates ? kProbeCandidates : first);
      }
      long long ramp = (long long)first << (tail_rounds < 8 ? tail_rounds : 8);
      if (ramp < 2) ramp = 2;
      if (probe_k > ramp) probe_k = int(ramp);
      tail_rounds++;
      // an instance every candidate of which was rejected waits in ST_PROBE and is skipped by the regular pass: it is
      // only picked up again by the next probing launch, so a tail that has probed must keep probing (the list only
      // shrinks and the ramp only grows, so this cannot trigger; if it ever does, fail instead of dropping instances)
      if (probed_in_tail && probe_k < 2) return fail(ILQG_ERR_HIP, "a probing line-search tail lost its probe launch");
      if (probe_k >= 2) {
        probed_in_tail = true;
        sa.probe_pool = probe_pool;
        sa.probe_k = probe_k;
        hipLaunchKernelGGL(k_proll, dim3(round_instances, probe_k), dim3(64), lds_roll, stream, d, sa);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_prows, dim3(row_chunks, round_instances * probe_k), dim3(64), lds_prows, stream, d, sa);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(ilq_probe_pick_kernel<T>, dim3(round_instances), dim3(kProbeCandidates), 0, stream, d, sa);
        HIP_TRY(hipGetLastError());
      }
    }
    sa.round_count = round_instances;
    hipLaunchKernelGGL(k_roll, dim3(round_instances), dim3(64), lds_roll, stream, d, sa);
    HIP_TRY(hipGetLastError());
    sa.first = 0;
    hipLaunchKernelGGL(k_rows, dim3(row_chunks, round_instances), dim3(64), lds_rows, stream, d, sa);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_decide, dim3(round_instances), dim3(64), lds_decide, stream, d, sa);
    HIP_TRY(hipGetLastError());
    if (read_round_counters(p, stream) != ILQG_OK) return ILQG_ERR_HIP;
    waiting_lq += p->h_unfinished[0];
    waiting_exit += p->h_unfinished[1];
    if (round > cap) return fail(ILQG_ERR_HIP, "solve did not terminate within its iteration bound");
    if (p->h_unfinished[3]) {  // some instances want another pass (initial quadraticisation, back-tracking): they are listed
      sa.ids = sa.ids_next;
      round_instances = p->h_unfinished[3];
      list ^= 1;
      continue;
    }
    sa.ids = nullptr;
    round_instances = batch;
    tail_rounds = 0;
    probed_in_tail = false;
    int want_lq = waiting_lq, want_exit = waiting_exit, restarted = 0;
    waiting_lq = waiting_exit = 0;
    if (timed && want_lq) {  // the loop condition of src/ilq_solver.cpp:123-124 (see DimsLaunch::solve)
      const double now = wall();
      if (iteration_open) {
        p->loop_add(now - tic);
        inner_elapsed += now - tic;
        iteration_open = false;
      }
      if (!(inner_elapsed < budget - p->loop_upper_bound())) {
        hipLaunchKernelGGL(ilq_deadline_kernel<T>, dim3(batch), dim3(64), 0, stream, d, sa);
        HIP_TRY(hipGetLastError());
        want_exit = 1;
        want_lq = 0;
      }
    }
    if (logging && (want_exit || want_lq)) {
      hipLaunchKernelGGL(ilq_log_kernel<T>, dim3(batch), dim3(256), 0, stream, d, sa, lg);
      HIP_TRY(hipGetLastError());
    }
    if (want_exit) {
      if (!p->counters_clean) HIP_TRY(hipMemsetAsync(p->d_unfinished, 0, 4 * sizeof(int), stream));
      if (outer.before_exit()) sa.outer_closed = 1;
      hipLaunchKernelGGL(k_exit, dim3(batch), dim3(64), lds_exit, stream, d, sa);
      HIP_TRY(hipGetLastError());
      p->counters_clean = false;
      if (al_mode) {
        if (read_round_counters(p, stream) != ILQG_OK) return ILQG_ERR_HIP;
        restarted = p->h_unfinished[2];
        if (restarted) inner_elapsed = 0.0;
        outer.after_exit(restarted);
      }
    }
    if (want_lq) {
      if (timed) {
        tic = wall();
        iteration_open = true;
      }
      if (pad_nx) {
        const ilqg_status s = padded_launch(nullptr);
        if (s != ILQG_OK) return s;
      } else {
        hipLaunchKernelGGL(k_lq, dim3(batch), dim3(generic_sweep_threads(d.n, d.m)), lds_lq, stream, d, sa);
        HIP_TRY(hipGetLastError());
      }
    }
    if (!want_lq && !restarted) break;
  }
  return ILQG_OK;
}

static GenDims gen_dims_of(int n, int N, const int32_t* udim, int T) {
  GenDims g{};
  g.n = n; g.N = N; g.T = T;
  g.uoff[0] = 0;
  for (int i = 0; i < N; i++) {
    g.udim[i] = udim[i];
    g.uoff[i + 1] = g.uoff[i] + udim[i];
  }
  g.m = g.uoff[N];
  return g;
}

// ilqg_lq_feedback_batch / ilqg_lq_openloop_batch of a shape without an instantiation on the padded sweep of (nx, N, mu)
template <typename T>
static ilqg_status launch_lq_padded(const ilqg_dims* d, const PairTable& pt, int nx, int mu, bool open_loop, const void* A,
                                    const void* Bm, const void* Q, const void* l, const void* R, const void* r,
                                    const void* x0, void* P, void* alpha, void* dx, hipStream_t stream) {
  PairTable ptp;
  std::string err;
  if (!padded_pairs(pt, d->num_players, mu, &ptp, &err)) return fail(ILQG_ERR_INVALID, err);
#define X(NX_, NP_, MU_)                                                                                                  \
  if (nx == NX_ && d->num_players == NP_ && mu == MU_)                                                                     \
    return DimsLaunch<T, NX_, NP_, MU_>::lq_padded_batch(d, pt, ptp, open_loop, A, Bm, Q, l, R, r, x0, P, alpha, dx, stream);
  ILQG_FOR_DIMS(X)
#undef X
  return fail(ILQG_ERR_UNSUPPORTED, "no shape to embed the game in");
}

// ilqg_lq_feedback_batch / ilqg_lq_openloop_batch for a shape without a specialised instantiation (or when the caller
// asks for these kernels: ilqg_dims::sweep_formulation = ILQG_SWEEP_GENERIC).
template <typename T>
static ilqg_status launch_lq_generic(const ilqg_dims* d, const PairTable& pt, bool open_loop, const void* A, const void* Bm,
                                     const void* Q, const void* l, const void* R, const void* r, const void* x0, void* P,
                                     void* alpha, void* dx, void* costates, hipStream_t stream) {
  const GenDims gd = gen_dims_of(d->n, d->num_players, d->udim, d->T);
  if (gd.m > ILQG_MAX_UDIM_TOTAL) return fail(ILQG_ERR_UNSUPPORTED, "more than ILQG_MAX_UDIM_TOTAL controls in total");
  const size_t lds = (open_loop ? gen_openloop_lds_elems(gd.n, gd.N, gd.m) : gen_feedback_lds_elems(gd.n, gd.N, gd.m)) * sizeof(T);
  if (lds > kLdsPerWorkgroup) return fail(ILQG_ERR_UNSUPPORTED, "the game's value functions do not fit a CU's LDS");
  LQBatchArgs<T> g;
  g.A = (const T*)A; g.Bm = (const T*)Bm; g.Q = (const T*)Q; g.l = (const T*)l;
  g.R = (const T*)R; g.r = (const T*)r; g.x0 = (const T*)x0;
  g.P = (T*)P; g.alpha = (T*)alpha; g.dx = (T*)dx;
  g.costates = open_loop ? (T*)costates : nullptr;
  g.scratch = nullptr;
  const int row = open_loop ? gen_ol_row_elems(gd.n, gd.m, gd.N, costates != nullptr) : 0;
  if (open_loop) {
    const ilqg_status s = Scratch().reserve(size_t(d->batch) * d->T * size_t(row) * sizeof(T));
    if (s != ILQG_OK) return s;
    g.scratch = (T*)ilqg_shared::scratch_state().ptr;
  }
  g.T_steps = d->T;
  g.adaptive = open_loop ? 0 : d->adaptive_regularization;
  g.batch = d->batch;
  g.force_valu = 0;
  auto kern = lq_generic_kernel<T>;
  raise_lds_limit((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(d->batch), dim3(generic_sweep_threads(gd.n, gd.m)), lds, stream, g, gd, pt, open_loop ? 1 : 0, row);
  HIP_TRY(hipGetLastError());
  return ILQG_OK;
}

// RouteProgressCost subtracts RelativeTimeTracker::initial_time_ from the time it is handed (src/route_progress_cost.cpp:
// 58), which Problem::SetUpNextRecedingHorizon resets to the window's start (src/problem.cpp:120); the device tables
// are tabulated once with initial time 0 (ilqg_problem_create).  Re-anchoring such a problem would silently diverge
// from the reference after the first re-plan, so the receding-horizon entry points refuse it.
static const char* const kRouteProgressReceding =
    "receding horizon: the problem holds a RouteProgressCost, whose per-step nominals are tabulated for a first solve "
    "(initial time 0) only";

extern "C" {

const char* ilqg_last_error(void) { return g_err.c_str(); }

ilqg_status ilqg_set_scratch(void* device_buffer, size_t bytes) {
  ilqg_shared::ScratchState& st = ilqg_shared::scratch_state();
  if (st.ptr && !st.caller_owned) (void)hipFree(st.ptr);
  st.ptr = device_buffer;
  st.bytes = device_buffer ? bytes : 0;
  st.caller_owned = device_buffer != nullptr;
  return ILQG_OK;
}

// Diagnostics: device buffer [B][8] of int64 receiving per-stage shader-clock cycles of the next solves.
#if ILQG_DIAGNOSTIC_BUILD
void ilqg_debug_set_profile_buffer(void* buf) { g_prof = (long long*)buf; }
#endif

// out = X^T Y + C for 16x16 column-major device matrices, through the MFMA accumulator-layout
// path the LQ sweep uses (tests pin the register layouts with asymmetric inputs).
ilqg_status ilqg_selftest_mfma(int32_t dtype, const void* X, const void* Y, const void* C, void* out, void* stream) {
  ilqg_status s = check_device();
  if (s != ILQG_OK) return s;
  if (dtype == ILQG_F32)
    hipLaunchKernelGGL(mfma_selftest_kernel<float>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)X,
                       (const float*)Y, (const float*)C, (float*)out);
  else
    hipLaunchKernelGGL(mfma_selftest_kernel<double>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)X,
                       (const double*)Y, (const double*)C, (double*)out);
  HIP_TRY(hipGetLastError());
  return ILQG_OK;
}
int32_t ilqg_abi_version(void) { return ILQG_ABI_VERSION; }

ilqg_status ilqg_copy_bandwidth(void* dst, const void* src, size_t bytes, void* stream) {
  ilqg_status s = check_device();
  if (s != ILQG_OK) return s;
  if (!dst || !src || bytes < 16 || (bytes & 15) || (reinterpret_cast<uintptr_t>(dst) & 15) || (reinterpret_cast<uintptr_t>(src) & 15))
    return fail(ILQG_ERR_INVALID, "ilqg_copy_bandwidth: 16-byte aligned buffers and a multiple of 16 bytes");
  const size_t n16 = bytes / 16;
  const size_t blocks = (n16 + 1023) / 1024;  // 16 KB per workgroup
  hipLaunchKernelGGL(copy_bandwidth_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<copy_v4f*>(dst), reinterpret_cast<const copy_v4f*>(src), n16);
  HIP_TRY(hipGetLastError());
  return ILQG_OK;
}

ilqg_status ilqg_problem_row_program(const ilqg_problem* p, int32_t* words_out, int32_t capacity, int32_t* num_words,
                                     int32_t* static_id) {
  if (!p || !num_words) return fail(ILQG_ERR_INVALID, "null argument");
  *num_words = int32_t(p->row_prog_host.size());
  if (static_id) *static_id = p->static_prog;
  if (words_out) {
    if (capacity < *num_words) return fail(ILQG_ERR_INVALID, "ilqg_problem_row_program: buffer too small");
    std::memcpy(words_out, p->row_prog_host.data(), sizeof(int32_t) * p->row_prog_host.size());
  }
  return ILQG_OK;
}

ilqg_status ilqg_problem_last_schedule(const ilqg_problem* p, int32_t* schedule_out) {
  if (!p || !schedule_out) return fail(ILQG_ERR_INVALID, "null argument");
  *schedule_out = p->last_schedule;
  return ILQG_OK;
}

ilqg_status ilqg_device_info(char* name_out, int32_t name_len, int32_t* num_cus) {
  ilqg_status s = check_device();
  if (s != ILQG_OK) return s;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (name_out && name_len > 0) {
    std::snprintf(name_out, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (num_cus) *num_cus = prop.multiProcessorCount;
  return ILQG_OK;
}

void ilqg_default_solver_params(ilqg_solver_params* p) {  // solver_params.h:50-84
  p->convergence_tolerance = 1e-1f;
  p->max_solver_iters = 1000;
  p->linesearch = 1;
  p->initial_alpha_scaling = 0.5f;
  p->geometric_alpha_scaling = 0.5f;
  p->max_backtracking_steps = 10;
  p->expected_decrease_fraction = 0.1f;
  p->open_loop = 0;
  p->unconstrained_solver_max_iters = 10;
  p->geometric_mu_scaling = 1.1f;
  p->geometric_mu_downscaling = 0.5f;
  p->geometric_lambda_downscaling = 0.5f;
  p->constraint_error_tolerance = 1e-1f;
}

ilqg_status ilqg_lq_feedback_batch(const ilqg_dims* d, const void* A, const void* Bm, const void* Q, const void* l,
                                   const void* R, const void* r, const ilqg_pair* pairs_host, int32_t npairs,
                                   const void* x0, void* P, void* alpha, void* dx, void* costates, void* stream) {
  if (!d || !A || !Bm || !Q || !l || !R || !r || !pairs_host || !P || !alpha)
    return fail(ILQG_ERR_INVALID, "null argument");
  if (d->num_players < 1 || d->n < 1 || d->T < 1 || d->T > kMaxT || d->batch < 0)
    return fail(ILQG_ERR_INVALID, "bad dimensions");
  if (d->num_players > ILQG_MAX_PLAYERS || d->n > ILQG_MAX_XDIM)
    return fail(ILQG_ERR_UNSUPPORTED, "more than ILQG_MAX_XDIM states or ILQG_MAX_PLAYERS players");
  if (costates && !dx) return fail(ILQG_ERR_INVALID, "costates come with delta_xs (lq_feedback_solver.cpp:77-78)");
  for (const void* ptr : {A, Bm, Q, l, R, r})
    if (reinterpret_cast<uintptr_t>(ptr) % 16 != 0) return fail(ILQG_ERR_INVALID, "array bases must be 16-byte aligned");
  PairTable pt;
  std::string err;
  if (!build_pairs(pairs_host, npairs, d->udim, d->num_players, &pt, &err)) return fail(ILQG_ERR_INVALID, err);
  int mu = 0;
  const bool uniform = uniform_udim(d->udim, d->num_players, &mu);  // the specialised sweeps: one m_i for all players
  ilqg_status s = check_device();
  if (s != ILQG_OK) return s;
  if (d->batch == 0) return ILQG_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t elem = d->dtype == ILQG_F32 ? 4 : 8;
  const int N = d->num_players;
  int m = 0;
  for (int i = 0; i < N; i++) m += d->udim[i];
  // costates: Z_i, zeta_i of every step (this translation unit's scratch; the sweep's launcher has its own)
  if (costates) {
    s = Scratch().reserve(size_t(d->batch) * costates_scratch_elems(d->n, N, d->T) * elem);
    if (s != ILQG_OK) return s;
  }
  auto finish = [&](ilqg_status launched) -> ilqg_status {
    if (launched != ILQG_OK || !costates) return launched;
    CostateDims cd;
    cd.n = d->n; cd.N = N; cd.m = m; cd.T = d->T;
    cd.uoff[0] = 0;
    for (int i = 0; i < N; i++) cd.uoff[i + 1] = cd.uoff[i] + d->udim[i];
    const size_t lds = costates_lds_elems(d->n, N) * elem;
    void* zs = ilqg_shared::scratch_state().ptr;
    if (d->dtype == ILQG_F32) {
      auto kern = lq_feedback_costates_kernel<float>;
      raise_lds_limit((const void*)kern, lds);
      hipLaunchKernelGGL(kern, dim3(d->batch), dim3(256), lds, st, cd, pt, (const float*)A, (const float*)Bm,
                         (const float*)Q, (const float*)l, (const float*)R, (const float*)r, (const float*)P,
                         (const float*)alpha, (const float*)dx, (float*)zs, (float*)costates);
    } else {
      auto kern = lq_feedback_costates_kernel<double>;
      raise_lds_limit((const void*)kern, lds);
      hipLaunchKernelGGL(kern, dim3(d->batch), dim3(256), lds, st, cd, pt, (const double*)A, (const double*)Bm,
                         (const double*)Q, (const double*)l, (const double*)R, (const double*)r, (const double*)P,
                         (const double*)alpha, (const double*)dx, (double*)zs, (double*)costates);
    }
    HIP_TRY(hipGetLastError());
    return ILQG_OK;
  };
#define X(NX_, NP_, MU_)                                                                              \
  if (uniform && d->sweep_formulation != ILQG_SWEEP_GENERIC && d->n == NX_ && d->num_players == NP_ && mu == MU_) { \
    return finish(d->dtype == ILQG_F32                                                                \
               ? DimsLaunch<float, NX_, NP_, MU_>::lq(d, pt, A, Bm, Q, l, R, r, x0, P, alpha, dx, st)      \
               : DimsLaunch<double, NX_, NP_, MU_>::lq(d, pt, A, Bm, Q, l, R, r, x0, P, alpha, dx, st));   \
  }
  ILQG_FOR_DIMS(X)
#undef X
  // no specialised instantiation (or players with different control dimensions): the game embedded in the smallest
  // instantiated shape that holds it, on that shape's sweep (padded_lq_batch_kernel) ...
  {
    int nx = 0, pmu = 0;
    if (d->sweep_formulation != ILQG_SWEEP_GENERIC && pick_padded_shape(d->n, N, d->udim, &nx, &pmu))
      return finish(d->dtype == ILQG_F32 ? launch_lq_padded<float>(d, pt, nx, pmu, false, A, Bm, Q, l, R, r, x0, P, alpha, dx, st)
                                         : launch_lq_padded<double>(d, pt, nx, pmu, false, A, Bm, Q, l, R, r, x0, P, alpha, dx, st));
  }
  // ... or, where there is none (or the caller asks for it), the run-time-dimensioned sweep
  return finish(d->dtype == ILQG_F32
                    ? launch_lq_generic<float>(d, pt, false, A, Bm, Q, l, R, r, x0, P, alpha, dx, nullptr, st)
                    : launch_lq_generic<double>(d, pt, false, A, Bm, Q, l, R, r, x0, P, alpha, dx, nullptr, st));
}

ilqg_status ilqg_lq_openloop_batch(const ilqg_dims* d, const void* A, const void* Bm, const void* Q, const void* l,
                                   const void* R, const void* r, const ilqg_pair* pairs_host, int32_t npairs,
                                   const void* x0, void* P, void* alpha, void* dx, void* costates, void* stream) {
  if (!d || !A || !Bm || !Q || !l || !R || !r || !pairs_host || !P || !alpha)
    return fail(ILQG_ERR_INVALID, "null argument");
  if (d->num_players < 1 || d->n < 1 || d->T < 2 || d->T > kMaxT || d->batch < 0)
    return fail(ILQG_ERR_INVALID, "bad dimensions");
  if (d->num_players > ILQG_MAX_PLAYERS || d->n > ILQG_MAX_XDIM)
    return fail(ILQG_ERR_UNSUPPORTED, "more than ILQG_MAX_XDIM states or ILQG_MAX_PLAYERS players");
  if (costates && !dx) return fail(ILQG_ERR_INVALID, "costates come with delta_xs (lq_open_loop_solver.cpp:83-84)");
  for (const void* ptr : {A, Bm, Q, l, R, r})
    if (reinterpret_cast<uintptr_t>(ptr) % 16 != 0) return fail(ILQG_ERR_INVALID, "array bases must be 16-byte aligned");
  PairTable pt;
  std::string err;
  if (!build_pairs(pairs_host, npairs, d->udim, d->num_players, &pt, &err)) return fail(ILQG_ERR_INVALID, err);
  int mu = 0;
  const bool uniform = uniform_udim(d->udim, d->num_players, &mu);
  ilqg_status s = check_device();
  if (s != ILQG_OK) return s;
  if (d->batch == 0) return ILQG_OK;
  hipStream_t st = (hipStream_t)stream;
#define X(NX_, NP_, MU_)                                                                                      \
  if (uniform && d->sweep_formulation != ILQG_SWEEP_GENERIC && d->n == NX_ && d->num_players == NP_ && mu == MU_) { \
    return d->dtype == ILQG_F32                                                                               \
               ? DimsLaunch<float, NX_, NP_, MU_>::lq_openloop(d, pt, A, Bm, Q, l, R, r, x0, P, alpha, dx, costates, st)     \
               : DimsLaunch<double, NX_, NP_, MU_>::lq_openloop(d, pt, A, Bm, Q, l, R, r, x0, P, alpha, dx, costates, st);   \
  }
  ILQG_FOR_DIMS(X)
#undef X
  {
    int nx = 0, pmu = 0;  // (the padded sweep does not carry costates out: with them, the run-time-dimensioned sweep)
    if (!costates && d->sweep_formulation != ILQG_SWEEP_GENERIC && pick_padded_shape(d->n, d->num_players, d->udim, &nx, &pmu))
      return d->dtype == ILQG_F32 ? launch_lq_padded<float>(d, pt, nx, pmu, true, A, Bm, Q, l, R, r, x0, P, alpha, dx, st)
                                  : launch_lq_padded<double>(d, pt, nx, pmu, true, A, Bm, Q, l, R, r, x0, P, alpha, dx, st);
  }
  return d->dtype == ILQG_F32 ? launch_lq_generic<float>(d, pt, true, A, Bm, Q, l, R, r, x0, P, alpha, dx, costates, st)
                              : launch_lq_generic<double>(d, pt, true, A, Bm, Q, l, R, r, x0, P, alpha, dx, costates, st);
}

// `host_only`: everything ilqg_problem_create does on the host — validation, flattening, the row program and its match
// against the registered structures — without touching a device; the handle then only serves ilqg_problem_row_program
// and ilqg_problem_destroy (ilqg_row_program_build).
static ilqg_status problem_create_impl(const ilqg_problem_desc* desc, ilqg_problem** out, bool host_only) {
  if (!desc || !out) return fail(ILQG_ERR_INVALID, "null argument");
  if (desc->num_players < 1 || desc->num_players > ILQG_MAX_PLAYERS) return fail(ILQG_ERR_INVALID, "bad player count");
  if (desc->T < 2 || desc->T > kMaxT) return fail(ILQG_ERR_INVALID, "bad horizon");
  if (!host_only) {
    ilqg_status s = check_device();
    if (s != ILQG_OK) return s;
  }
  auto* p = new ilqg_problem;
  p->desc = *desc;
  DevProblem& d = p->dev;
  std::memset(&d, 0, sizeof(d));
  d.N = desc->num_players;
  d.T = desc->T;
  d.dt = desc->dt;
  d.xoff[0] = 0;
  d.uoff[0] = 0;
  for (int i = 0; i < d.N; i++) {
    const ilqg_subsystem& sub = desc->subsystems[i];
    const int want_x = (sub.kind == ILQG_DYN_UNICYCLE_4D || sub.kind == ILQG_DYN_UNICYCLE_4D_DISTURBED ||
                        sub.kind == ILQG_DYN_POINT_MASS_2D || sub.kind == ILQG_DYN_DELAYED_DUBINS_CAR) ? 4
                       : sub.kind == ILQG_DYN_UNICYCLE_5D ? 5 : sub.kind == ILQG_DYN_CAR_7D ? 7
                       : sub.kind == ILQG_DYN_CAR_5D ? 5 : sub.kind == ILQG_DYN_CAR_6D ? 6
                       : sub.kind == ILQG_DYN_PLANAR_DISTURBANCE ? 0 : sub.kind == ILQG_DYN_DUBINS_CAR ? 3
                       : sub.kind == ILQG_DYN_AIR_3D_EVADER ? 3 : sub.kind == ILQG_DYN_AIR_3D_PURSUER ? 0 : -1;
    const bool one_control = sub.kind == ILQG_DYN_DUBINS_CAR || sub.kind == ILQG_DYN_AIR_3D_EVADER ||
                             sub.kind == ILQG_DYN_AIR_3D_PURSUER || sub.kind == ILQG_DYN_DELAYED_DUBINS_CAR;
    const int want_u = one_control ? 1 : 2;
    // TwoPlayerUnicycle4D is exactly the pair (disturbed unicycle, disturbance) and nothing else
    const bool paired = sub.kind == ILQG_DYN_UNICYCLE_4D_DISTURBED || sub.kind == ILQG_DYN_PLANAR_DISTURBANCE ||
                        sub.kind == ILQG_DYN_AIR_3D_EVADER || sub.kind == ILQG_DYN_AIR_3D_PURSUER;
    const bool pair_ok = !paired ||
                         (desc->num_players == 2 && desc->subsystems[0].kind == ILQG_DYN_UNICYCLE_4D_DISTURBED &&
                          desc->subsystems[1].kind == ILQG_DYN_PLANAR_DISTURBANCE) ||
                         (desc->num_players == 2 && desc->subsystems[0].kind == ILQG_DYN_AIR_3D_EVADER &&
                          desc->subsystems[1].kind == ILQG_DYN_AIR_3D_PURSUER);
    if (!pair_ok) {
      delete p;
      return fail(ILQG_ERR_UNSUPPORTED, "the shared-state kinds only occur as the pairs (4, 5) and (7, 8)");
    }
    if ((sub.kind == ILQG_DYN_POINT_MASS_2D) != (desc->subsystems[0].kind == ILQG_DYN_POINT_MASS_2D)) {
      delete p;
      return fail(ILQG_ERR_UNSUPPORTED, "point masses (kind 9) only occur in games made of point masses");
    }
    if (want_x < 0 || sub.xdim != want_x || sub.udim != want_u) {
      delete p;
      return fail(ILQG_ERR_UNSUPPORTED, "unknown subsystem kind / dimension");
    }
    d.sub_kind[i] = sub.kind;
    d.sub_param[i] = sub.param0;
    d.udim[i] = sub.udim;
    d.xoff[i + 1] = d.xoff[i] + sub.xdim;
    d.uoff[i + 1] = d.uoff[i] + sub.udim;
    d.state_reg[i] = desc->player_costs[i].state_regularization;
    d.control_reg[i] = desc->player_costs[i].control_regularization;
    d.structure[i] = desc->player_costs[i].structure;
  }
  d.n = d.xoff[d.N];
  d.m = d.uoff[d.N];
  if (d.n > ILQG_MAX_XDIM || d.m > ILQG_MAX_UDIM_TOTAL) {  // before any table is sized by them
    delete p;
    return fail(ILQG_ERR_UNSUPPORTED, "more than ILQG_MAX_XDIM states or ILQG_MAX_UDIM_TOTAL controls");
  }
  {
    bool plain = false;
    for (int i = 0; i < d.N; i++) plain = plain || is_plain_rk4_kind(d.sub_kind[i]);
    // Unicycle5D / Car7D / DelayedDubinsCar rows need an instantiation that carries the plain RK4 (dims_use_plain_rk4,
    // csrc/ilqg_stages.hpp); in any other shape they run on the run-time-dimensioned path, which picks its integrator
    // from the models
    if (plain && !dims_use_plain_rk4(d.n, d.N, d.udim[0])) p->generic = true;
  }
  // DistanceBetween of the first subsystem: (px, py) where the model overrides it (two_player_unicycle_4d.h:141-147
  // too), the whole block where it does not (the two Dubins cars: single_player_dynamical_system.h:69-71)
  d.sync_dist_dims = d.sub_kind[0] == ILQG_DYN_DUBINS_CAR ? 3 : (d.sub_kind[0] == ILQG_DYN_DELAYED_DUBINS_CAR ? 4 : 2);
  // pair table in PlayerCost first-touch order: control costs, then control constraints
  std::vector<ilqg_pair> pairs;
  std::vector<int> from_cost;
  for (int i = 0; i < d.N; i++)
    for (int pass = 0; pass < 2; pass++)
      for (int ti = 0; ti < desc->num_terms; ti++) {
        const ilqg_cost_term& t = desc->terms[ti];
        if (t.player != i) continue;
        if (pass == 0 && t.role != ILQG_ROLE_CONTROL_COST) continue;
        if (pass == 1 && t.role != ILQG_ROLE_CONTROL_CONSTRAINT) continue;
        bool found = false;
        for (auto& pr : pairs) found = found || (pr.i == i && pr.j == t.arg);
        if (!found) {
          pairs.push_back({i, t.arg});
          from_cost.push_back(pass == 0 ? 1 : 0);
        }
      }
  std::string err;
  if (!build_pairs(pairs.data(), (int)pairs.size(), d.udim, d.N, &d.pairs, &err)) {
    delete p;
    return fail(ILQG_ERR_INVALID, err);
  }
  for (size_t q = 0; q < pairs.size(); q++) d.pairs.from_cost[q] = from_cost[q];
  d.num_terms = desc->num_terms;
  d.num_polylines = desc->num_polylines;
  std::vector<DevTerm> dt(desc->num_terms > 0 ? desc->num_terms : 1);
  int nc = 0;
  for (int ti = 0; ti < desc->num_terms; ti++) {
    const ilqg_cost_term& t = desc->terms[ti];
    DevTerm& o = dt[ti];
    o.kind = t.kind; o.role = t.role; o.player = t.player; o.arg = t.arg;
    for (int q = 0; q < 4; q++) o.idx[q] = t.idx[q];
    o.weight = t.weight; o.value = t.value; o.flags = t.flags; o.polyline = t.polyline;
    o.child_begin = t.child_begin; o.child_count = t.child_count; o.slot = t.constraint_slot;
    o.k_start = t.first_step;
    if (t.kind == ILQG_COST_WEIGHTED_CONVEX_PROXIMITY) {  // its two speed indices ride in `polyline` (wcp_indices)
      if (t.role != ILQG_ROLE_STATE_COST || t.idx_extra[0] < 0 || t.idx_extra[0] >= d.n || t.idx_extra[1] < 0 ||
          t.idx_extra[1] >= d.n) {
        delete p;
        return fail(ILQG_ERR_INVALID, "WeightedConvexProximityCost must be a top-level state cost with speed indices "
                                      "inside the state");
      }
      o.polyline = t.idx_extra[0] | (t.idx_extra[1] << 16);
    }
    // Constraint::is_equality_ is only carried for the affine constraints (ilqg.h): on any other kind the multiplier
    // update would drop its clip at zero while the mu gate stayed an inequality's
    if ((t.flags & ILQG_FLAG_EQUALITY) && t.kind != ILQG_CONSTRAINT_AFFINE_SCALAR && t.kind != ILQG_CONSTRAINT_AFFINE_VECTOR) {
      delete p;
      return fail(ILQG_ERR_INVALID, "ILQG_FLAG_EQUALITY is only defined for the affine constraints");
    }
    if (t.constraint_slot >= 0 && t.constraint_slot + 1 > nc) nc = t.constraint_slot + 1;
  }
  d.num_constraints = nc;
  // ---- where each term's argument vector sits inside a row's [x | u] ----
  for (int ti = 0; ti < desc->num_terms; ti++) {
    DevTerm& o = dt[ti];
    const bool on_state = o.role == ILQG_ROLE_STATE_COST || o.role == ILQG_ROLE_STATE_CONSTRAINT ||
                          o.role == ILQG_ROLE_CHILD;
    o.arg_off = on_state ? 0 : d.n + d.uoff[o.arg];
    o.arg_dim = on_state ? d.n : d.udim[o.arg];
  }
  // ---- coefficient blocks of the affine constraints, in both precisions (DevProblem::dense_f / dense_d) ----
  std::vector<float> dense_f;
  std::vector<double> dense_d;
  for (int ti = 0; ti < desc->num_terms; ti++) {
    DevTerm& o = dt[ti];
    if (!term_is_affine(o.kind)) continue;
    const int dim = o.arg_dim;
    const bool vec = o.kind == ILQG_CONSTRAINT_AFFINE_VECTOR;
    const long long count = vec ? (long long)dim * dim + dim : dim + 1;
    const bool constraint_role = o.role == ILQG_ROLE_STATE_CONSTRAINT || o.role == ILQG_ROLE_CONTROL_CONSTRAINT;
    if (!constraint_role || o.slot < 0 || desc->dense_params == nullptr || desc->terms[ti].polyline < 0 ||
        (long long)desc->terms[ti].polyline + count > desc->num_dense_params) {
      delete p;
      return fail(ILQG_ERR_INVALID, "an affine constraint must be a state / control constraint with a multiplier slot "
                                    "and a coefficient block inside ilqg_problem_desc::dense_params");
    }
    const float* src = desc->dense_params + desc->terms[ti].polyline;
    o.polyline = int(dense_f.size());  // from here on: the offset of its block in the device tables
    auto emit = [&](auto& out) {
      using S = typename std::decay<decltype(out)>::type::value_type;
      for (long long e = 0; e < count; e++) out.push_back(S(src[e]));
      if (vec)  // ATA_ = A^T A, AAT_ = A A^T as the constructor forms them (affine_vector_constraint.h:60-61)
        for (int which = 0; which < 2; which++)
          for (int j = 0; j < dim; j++)
            for (int i = 0; i < dim; i++) {
              S acc = S(0);
              for (int q = 0; q < dim; q++)
                acc += which == 0 ? S(src[q + dim * i]) * S(src[q + dim * j]) : S(src[i + dim * q]) * S(src[j + dim * q]);
              out.push_back(acc);
            }
    };
    const size_t before = dense_f.size();
    emit(dense_f);
    emit(dense_d);
    (void)before;
  }
  p->terms_host.assign(desc->terms, desc->terms + desc->num_terms);
  const int npts = desc->num_polylines ? desc->polyline_offsets[desc->num_polylines] : 0;
  // LineSegment2 objects of every polyline, in both precisions (line_segment2.h:55-62)
  std::vector<float> segs_f;
  std::vector<double> segs_d;
  {
    auto emit = [&](auto& out, auto ax, auto ay, auto bx, auto by) {
      using S = decltype(ax);
      const S dx = ax - bx, dy = ay - by;
      const S len = std::sqrt(dx * dx + dy * dy);
      out.push_back(ax); out.push_back(ay); out.push_back(bx); out.push_back(by);
      out.push_back(len); out.push_back((bx - ax) / len); out.push_back((by - ay) / len);
    };
    for (int q = 0; q < desc->num_polylines; q++) {
      const int b0 = desc->polyline_offsets[q], e0 = desc->polyline_offsets[q + 1];
      const float* pts = desc->polyline_points + 2 * b0;
      const int nseg = e0 - b0 - 1;
      for (int c = 0; c < nseg; c++) {
        auto P = [&](int idx, int xy) { return pts[2 * idx + xy]; };
        const int pm = c > 0 ? c - 1 : c, pn = c + 2 <= nseg ? c + 2 : c + 1;
        emit(segs_f, P(c, 0), P(c, 1), P(c + 1, 0), P(c + 1, 1));
        emit(segs_f, P(pm, 0), P(pm, 1), P(c + 1, 0), P(c + 1, 1));
        emit(segs_f, P(c, 0), P(c, 1), P(pn, 0), P(pn, 1));
        emit(segs_d, double(P(c, 0)), double(P(c, 1)), double(P(c + 1, 0)), double(P(c + 1, 1)));
        emit(segs_d, double(P(pm, 0)), double(P(pm, 1)), double(P(c + 1, 0)), double(P(c + 1, 1)));
        emit(segs_d, double(P(c, 0)), double(P(c, 1)), double(P(pn, 0)), double(P(pn, 1)));
      }
    }
  }
  d.total_segs = int(segs_f.size() / kSegStride);
  // Per-step nominals of the time-dependent costs, one table per such term and geometry precision (doubles: the
  // path-length nominal is a double product in the reference, nominal_path_length_cost.cpp:53; the route point is a
  // pair of the geometry's scalars, exact in double).  t = RelativeTime(k) = double(k) * dt (relative_time_tracker.h:
  // 63-65); the route position is a float (a scalar of the geometry) made from a double expression
  // (route_progress_cost.cpp:57-59) and Polyline2::PointAt walks the cumulative lengths (src/polyline2.cpp:68-103).
  std::vector<double> tnom_f, tnom_d;
  {
    int ntab = 0;
    for (int ti = 0; ti < desc->num_terms; ti++) {
      DevTerm& o = dt[ti];
      if (!term_is_time_dependent(o.kind)) continue;
      const bool route = o.kind == ILQG_COST_ROUTE_PROGRESS;
      if (o.role != ILQG_ROLE_STATE_COST || (route && (o.polyline < 0 || o.polyline >= desc->num_polylines))) {
        ilqg_problem_destroy(p);
        return fail(ILQG_ERR_INVALID, "a time-dependent cost must be a top-level state cost (with a polyline, for "
                                      "RouteProgressCost)");
      }
      const int src_poly = o.polyline;
      if (route) {
        // Polyline2::PointAt CHECKs its argument (src/polyline2.cpp:68-103): a route position that is negative at any
        // step, or a polyline without a segment, is a programmer error there and ILQG_ERR_INVALID here
        const int nseg_r = desc->polyline_offsets[src_poly + 1] - desc->polyline_offsets[src_poly] - 1;
        const double pos_first = double(desc->terms[ti].value2);
        const double pos_last = pos_first + double(d.T - 1) * d.dt * double(o.value);
        if (nseg_r < 1 || !(pos_first >= 0.0) || !(pos_last >= 0.0)) {
          ilqg_problem_destroy(p);
          return fail(ILQG_ERR_INVALID, "RouteProgressCost: the route needs a segment and a route position that stays "
                                        "non-negative over the horizon (initial_route_pos, nominal_speed)");
        }
        p->has_route_progress = true;
      }
      auto point_at = [&](const auto& segs, auto route_pos, double* px, double* py) {
        using S = decltype(route_pos);
        const int first = desc->polyline_offsets[src_poly] - src_poly;
        const int nseg = desc->polyline_offsets[src_poly + 1] - desc->polyline_offsets[src_poly] - 1;
        std::vector<S> cumulative(1, S(0));
        for (int c = 0; c < nseg; c++) cumulative.push_back(cumulative.back() + segs[size_t(first + c) * kSegStride + 4]);
        auto upper = std::upper_bound(cumulative.begin(), cumulative.end(), route_pos);
        if (upper == cumulative.end()) upper--;
        upper--;
        const size_t idx = size_t(upper - cumulative.begin());
        const S remaining = route_pos - cumulative[idx];
        const S* sg = &segs[size_t(first + idx) * kSegStride];
        *px = double(S(sg[0] + remaining * sg[5]));
        *py = double(S(sg[1] + remaining * sg[6]));
      };
      for (int k = 0; k < d.T; k++) {
        const double t = double(k) * d.dt;
        double f0 = t * double(o.value), f1 = 0.0, d0 = f0, d1 = 0.0;
        if (route) {
          const double pos = double(desc->terms[ti].value2) + (t - 0.0) * double(o.value);
          point_at(segs_f, float(pos), &f0, &f1);
          point_at(segs_d, double(pos), &d0, &d1);
        }
        tnom_f.push_back(f0); tnom_f.push_back(f1);
        tnom_d.push_back(d0); tnom_d.push_back(d1);
      }
      o.polyline = ntab++;  // from here on: the term's table
    }
  }
  // TotalCosts summation order per player: state costs then control costs, table order
  int maxc = 0;
  for (int i = 0; i < d.N; i++) {
    int cnt = 0;
    for (int ti = 0; ti < desc->num_terms; ti++)
      if (dt[ti].player == i && (dt[ti].role == ILQG_ROLE_STATE_COST || dt[ti].role == ILQG_ROLE_CONTROL_COST)) cnt++;
    if (cnt > maxc) maxc = cnt;
  }
  d.cost_order_stride = maxc + 1;
  std::vector<int> order(size_t(d.N) * d.cost_order_stride, 0);
  for (int i = 0; i < d.N; i++) {
    int* o = order.data() + size_t(i) * d.cost_order_stride;
    for (int role = 0; role < 2; role++)
      for (int ti = 0; ti < desc->num_terms; ti++)
        if (dt[ti].player == i && dt[ti].role == role) o[1 + o[0]++] = ti;
  }
  RowProgramHost rph;
  {
    std::string rerr;
    if (!build_row_program(d, dt, desc->polyline_offsets, &rph, &rerr)) {
      ilqg_problem_destroy(p);
      return fail(ILQG_ERR_UNSUPPORTED, rerr);
    }
  }
  d.row_prog_words = int(rph.words.size());
  p->row_prog_host = rph.words;
  {
    // a registered structure (ilqg_rowprog_static.hpp)?  Word for word, parameters masked.
    std::vector<int> masked = rph.words;
    row_program_mask_parameters(&masked);
#define X(ID_, NX_, NP_, MU_)                                                                                         \
    if (p->static_prog == 0 && d.n == NX_ && d.N == NP_ && int(masked.size()) == StaticRowProg<ID_>::kWords &&         \
        std::memcmp(masked.data(), StaticRowProg<ID_>::w, sizeof(int) * masked.size()) == 0)                          \
      p->static_prog = ID_;
    ILQG_STATIC_PROGS(X)
#undef X
  }
  d.rp_pslots = rph.num_pslots;
  d.rp_lslots = rph.max_lslots;
  d.rp_gslots = rph.max_gslots;
  d.rp_maps_off = rph.maps_off;
  d.rp_maps_words = rph.maps_words;
  d.rp_compact_off = rph.compact_off;
  d.rp_compact_w = rph.compact_w;
  if (host_only) {
    *out = p;
    return ILQG_OK;
  }
  // ---- device tables ----
  hipError_t e = hipMalloc(&p->d_terms, sizeof(DevTerm) * dt.size());
  if (e == hipSuccess) e = hipMalloc(&p->d_poly_off, sizeof(int) * (desc->num_polylines + 1));
  if (e == hipSuccess && desc->num_polylines)
    e = hipMemcpy(p->d_poly_off, desc->polyline_offsets, sizeof(int) * (desc->num_polylines + 1), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&p->d_poly_pts, sizeof(float) * 2 * (npts > 0 ? npts : 1));
  if (e == hipSuccess && npts)
    e = hipMemcpy(p->d_poly_pts, desc->polyline_points, sizeof(float) * 2 * npts, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&p->d_segs_f, sizeof(float) * (segs_f.size() + 1));
  if (e == hipSuccess && !segs_f.empty())
    e = hipMemcpy(p->d_segs_f, segs_f.data(), sizeof(float) * segs_f.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&p->d_segs_d, sizeof(double) * (segs_d.size() + 1));
  if (e == hipSuccess && !segs_d.empty())
    e = hipMemcpy(p->d_segs_d, segs_d.data(), sizeof(double) * segs_d.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&p->d_tnom_f, sizeof(double) * (tnom_f.size() + 2));
  if (e == hipSuccess && !tnom_f.empty())
    e = hipMemcpy(p->d_tnom_f, tnom_f.data(), sizeof(double) * tnom_f.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&p->d_tnom_d, sizeof(double) * (tnom_d.size() + 2));
  if (e == hipSuccess && !tnom_d.empty())
    e = hipMemcpy(p->d_tnom_d, tnom_d.data(), sizeof(double) * tnom_d.size(), hipMemcpyHostToDevice);
  d.time_nominal_f = p->d_tnom_f;
  d.time_nominal_d = p->d_tnom_d;
  if (e == hipSuccess) e = hipMalloc(&p->d_dense_f, sizeof(float) * (dense_f.size() + 1));
  if (e == hipSuccess && !dense_f.empty())
    e = hipMemcpy(p->d_dense_f, dense_f.data(), sizeof(float) * dense_f.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&p->d_dense_d, sizeof(double) * (dense_d.size() + 1));
  if (e == hipSuccess && !dense_d.empty())
    e = hipMemcpy(p->d_dense_d, dense_d.data(), sizeof(double) * dense_d.size(), hipMemcpyHostToDevice);
  d.dense_f = p->d_dense_f;
  d.dense_d = p->d_dense_d;
  if (e == hipSuccess) e = hipMalloc(&p->d_cost_order, sizeof(int) * order.size());
  if (e == hipSuccess) e = hipMemcpy(p->d_cost_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice);
  // the term table is uploaded last: it carries the argument offsets computed above
  if (e == hipSuccess) e = hipMemcpy(p->d_terms, dt.data(), sizeof(DevTerm) * dt.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&p->d_unfinished, 4 * sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc(&p->h_unfinished, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) {
    for (int i = 0; i < 16; i++) p->h_unfinished[i] = 0;
    if (hipHostGetDevicePointer((void**)&p->h_unfinished_dev, p->h_unfinished, 0) != hipSuccess) p->h_unfinished_dev = nullptr;
  }
  if (e == hipSuccess) e = hipMalloc(&p->d_row_prog, sizeof(int) * rph.words.size());
  if (e == hipSuccess) e = hipMemcpy(p->d_row_prog, rph.words.data(), sizeof(int) * rph.words.size(), hipMemcpyHostToDevice);
  d.row_prog = p->d_row_prog;
  if (e != hipSuccess) {
    ilqg_problem_destroy(p);
    return fail(ILQG_ERR_HIP, std::string("problem tables: ") + hipGetErrorString(e));
  }
  d.segs_f = p->d_segs_f;
  d.segs_d = p->d_segs_d;
  d.cost_order = p->d_cost_order;
  d.terms = p->d_terms;
  d.poly_off = p->d_poly_off;
  d.poly_pts = p->d_poly_pts;
  p->desc.terms = nullptr;
  p->desc.polyline_offsets = nullptr;
  p->desc.polyline_points = nullptr;
  p->desc.dense_params = nullptr;
  if (!uniform_udim(d.udim, d.N, &p->mu_uniform)) p->mu_uniform = 0;
  {
    bool instantiated = false;
#define X(NX_, NP_, MU_) instantiated = instantiated || (d.n == NX_ && d.N == NP_ && p->mu_uniform == MU_);
    ILQG_FOR_DIMS(X)
#undef X
    if (!instantiated) p->generic = true;
  }
  *out = p;
  return ILQG_OK;
}

ilqg_status ilqg_problem_create(const ilqg_problem_desc* desc, ilqg_problem** out) { return problem_create_impl(desc, out, false); }

ilqg_status ilqg_row_program_build(const ilqg_problem_desc* desc, int32_t* words_out, int32_t capacity, int32_t* num_words,
                                   int32_t* static_id) {
  if (!num_words) return fail(ILQG_ERR_INVALID, "null argument");
  ilqg_problem* p = nullptr;
  ilqg_status s = problem_create_impl(desc, &p, true);
  if (s != ILQG_OK) return s;
  s = ilqg_problem_row_program(p, words_out, capacity, num_words, static_id);
  ilqg_problem_destroy(p);
  return s;
}

void ilqg_problem_destroy(ilqg_problem* p) {
  if (!p) return;
  if (p->d_terms) (void)hipFree(p->d_terms);
  if (p->d_poly_off) (void)hipFree(p->d_poly_off);
  if (p->d_poly_pts) (void)hipFree(p->d_poly_pts);
  if (p->d_segs_f) (void)hipFree(p->d_segs_f);
  if (p->d_segs_d) (void)hipFree(p->d_segs_d);
  if (p->d_dense_f) (void)hipFree(p->d_dense_f);
  if (p->d_dense_d) (void)hipFree(p->d_dense_d);
  if (p->d_tnom_f) (void)hipFree(p->d_tnom_f);
  if (p->d_tnom_d) (void)hipFree(p->d_tnom_d);
  if (p->d_cost_order) (void)hipFree(p->d_cost_order);
  if (p->d_row_prog) (void)hipFree(p->d_row_prog);
  if (p->d_unfinished) (void)hipFree(p->d_unfinished);
  if (p->h_unfinished) (void)hipHostFree(p->h_unfinished);
  delete p;
}

ilqg_status ilqg_problem_pairs(const ilqg_problem* p, ilqg_pair* pairs_host, int32_t* npairs) {
  if (!p || !pairs_host || !npairs) return fail(ILQG_ERR_INVALID, "null argument");
  *npairs = p->dev.pairs.npairs;
  for (int q = 0; q < *npairs; q++) pairs_host[q] = {p->dev.pairs.pi[q], p->dev.pairs.pj[q]};
  return ILQG_OK;
}

ilqg_status ilqg_workspace_bytes(const ilqg_problem* p, int32_t batch, uint64_t* bytes) {
  if (!p || !bytes) return fail(ILQG_ERR_INVALID, "null argument");
  const DevProblem& d = p->dev;
  const size_t elem = p->desc.dtype == ILQG_F32 ? 4 : 8;
  *bytes = ws_tail(d, batch > 0 ? batch : 0, elem, p->desc.params.open_loop ? ol_row_elems(d.n, d.m, d.N) : 0).total;
  return ILQG_OK;
}


ilqg_status ilqg_rollout_batch(const ilqg_problem* p, int32_t batch, const void* x0, const void* xs_ref,
                               const void* us_ref, const void* P, const void* alpha, const void* alpha_scale,
                               void* xs, void* us, const int32_t* active, void* stream) {
  if (!p || !x0 || !xs_ref || !us_ref || !P || !alpha || !xs || !us) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  const DevProblem& d = p->dev;
#define CALL(TY_)                                                                                              \
  [&]() -> ilqg_status {                                                                                     \
    RolloutBatchArgs<TY_> g{(const TY_*)x0, (const TY_*)xs_ref, (const TY_*)us_ref, (const TY_*)P, (const TY_*)alpha,    \
                          (const TY_*)alpha_scale, (TY_*)xs, (TY_*)us, active};                                    \
    hipLaunchKernelGGL(rollout_kernel<TY_>, dim3(batch), dim3(64), rollout_lds_elems(d.n, d.m) * sizeof(TY_),    \
                       (hipStream_t)stream, d, g);                                                           \
    HIP_TRY(hipGetLastError());                                                                              \
    return ILQG_OK;                                                                                          \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

ilqg_status ilqg_linearize_batch(const ilqg_problem* p, int32_t batch, const void* xs, const void* us, void* A,
                                 void* Bm, const int32_t* active, void* stream) {
  if (!p || !xs || !us || !A || !Bm) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  return launch_linquad(p, batch, xs, us, nullptr, nullptr, nullptr, A, Bm, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, active, stream);
}

ilqg_status ilqg_quadraticize_batch(const ilqg_problem* p, int32_t batch, const void* xs, const void* us,
                                    const void* lambdas, const void* mu, const int32_t* t_extreme, void* Q, void* l,
                                    void* R, void* r, const int32_t* active, void* stream) {
  if (!p || !xs || !us || !Q || !l || !R || !r) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  return launch_linquad(p, batch, xs, us, lambdas, mu, t_extreme, nullptr, nullptr, Q, l, R, r, nullptr, nullptr,
                        active, stream);
}

ilqg_status ilqg_total_costs_batch(const ilqg_problem* p, int32_t batch, const void* xs, const void* us, void* costs,
                                   int32_t* t_extreme, const int32_t* active, void* stream) {
  if (!p || !xs || !us || !costs) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  const DevProblem& d = p->dev;
  const size_t esz = p->desc.dtype == ILQG_F32 ? 4 : 8;
  ilqg_status s = Scratch().reserve(size_t(batch) * d.T * d.N * esz);
  if (s != ILQG_OK) return s;
  s = launch_linquad(p, batch, xs, us, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                     nullptr, nullptr, ilqg_shared::scratch_state().ptr, active, stream);
  if (s != ILQG_OK) return s;
#define CALL(TY_)                                                                                             \
  [&]() -> ilqg_status {                                                                                    \
    hipLaunchKernelGGL(costs_reduce_kernel<TY_>, dim3(batch), dim3(64), 0, (hipStream_t)stream, d,            \
                       (const TY_*)ilqg_shared::scratch_state().ptr, (TY_*)costs, t_extreme, active);                              \
    HIP_TRY(hipGetLastError());                                                                             \
    return ILQG_OK;                                                                                         \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

void ilqg_default_solve_options(ilqg_solve_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));  // reference semantics, every scheduling choice ILQG_CHOICE_AUTO
}

ilqg_status ilqg_solve_batch_ex(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us, void* P,
                                void* alpha, void* total_costs, int32_t* iters, int32_t* status, int32_t* converged,
                                void* workspace, const ilqg_solve_options* options, void* stream) {
  if (!p || !x0 || !xs || !us || !P || !alpha || !total_costs || !iters || !status || !converged || !workspace || !options)
    return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  const ilqg_solve_options& o = *options;
  if (o.fixed_iters < 0) return fail(ILQG_ERR_INVALID, "fixed_iters must not be negative");
  if (o.forced_steps && (o.fixed_iters <= 0 || o.augmented_lagrangian))
    return fail(ILQG_ERR_INVALID, "forced_steps needs fixed_iters > 0 and no augmented-Lagrangian loop");
  if (o.max_runtime > 0.0 && (o.fixed_iters > 0 || o.forced_steps))
    return fail(ILQG_ERR_INVALID, "max_runtime needs a free-running solve (fixed_iters = 0, no forced steps)");
  if (o.single_wave_sweep < ILQG_CHOICE_AUTO || o.single_wave_sweep > ILQG_CHOICE_ON ||
      o.adjoint_expected_decrease < ILQG_CHOICE_AUTO || o.adjoint_expected_decrease > ILQG_CHOICE_ON)
    return fail(ILQG_ERR_INVALID, "scheduling choices are ilqg_choice values");
  if (o.probe_first < 0 || o.probe_first > 1024) return fail(ILQG_ERR_INVALID, "probe_first: 0 (the library's choice) or a count");
  for (int32_t c : {o.split_trial, o.handoff, o.probe, o.counted})
    if (c < ILQG_CHOICE_AUTO || c > ILQG_CHOICE_ON) return fail(ILQG_ERR_INVALID, "scheduling choices are ilqg_choice values");
  const DevProblem& d = p->dev;
  hipStream_t st = (hipStream_t)stream;
  if (o.generic_kernels < ILQG_CHOICE_AUTO || o.generic_kernels > ILQG_CHOICE_ON || o.padded_sweep < ILQG_CHOICE_AUTO ||
      o.padded_sweep > ILQG_CHOICE_ON || o.probe_lanes < ILQG_CHOICE_AUTO || o.probe_lanes > ILQG_CHOICE_ON)
    return fail(ILQG_ERR_INVALID, "scheduling choices are ilqg_choice values");
  if (p->generic || o.generic_kernels == ILQG_CHOICE_ON)
    return p->desc.dtype == ILQG_F32
               ? generic_solve<float>(p, batch, x0, xs, us, P, alpha, total_costs, iters, status, converged, workspace, o, st)
               : generic_solve<double>(p, batch, x0, xs, us, P, alpha, total_costs, iters, status, converged, workspace, o, st);
#define X(NX_, NP_, MU_)                                                                                        \
  if (d.n == NX_ && d.N == NP_ && p->mu_uniform == MU_) {                                                       \
    return p->desc.dtype == ILQG_F32                                                                            \
               ? DimsLaunch<float, NX_, NP_, MU_>::solve(p, batch, x0, xs, us, P, alpha, total_costs, iters, status,  \
                                                         converged, workspace, o, st)                            \
               : DimsLaunch<double, NX_, NP_, MU_>::solve(p, batch, x0, xs, us, P, alpha, total_costs, iters, status, \
                                                          converged, workspace, o, st);                          \
  }
  ILQG_FOR_DIMS(X)
#undef X
  return fail(ILQG_ERR_UNSUPPORTED, "no device kernel instantiated for this problem's dimensions");
}

ilqg_status ilqg_ilq_solve_batch(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us, void* P,
                                 void* alpha, void* total_costs, int32_t* iters, int32_t* status, int32_t* converged,
                                 void* workspace, int32_t fixed_iters, void* stream) {
  ilqg_solve_options o;
  ilqg_default_solve_options(&o);
  o.fixed_iters = fixed_iters;
  return ilqg_solve_batch_ex(p, batch, x0, xs, us, P, alpha, total_costs, iters, status, converged, workspace, &o, stream);
}

ilqg_status ilqg_al_solve_batch(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us, void* P,
                                void* alpha, void* total_costs, int32_t* iters, int32_t* status, int32_t* converged,
                                void* workspace, void* stream) {
  ilqg_solve_options o;
  ilqg_default_solve_options(&o);
  o.augmented_lagrangian = 1;
  return ilqg_solve_batch_ex(p, batch, x0, xs, us, P, alpha, total_costs, iters, status, converged, workspace, &o, stream);
}

ilqg_status ilqg_solve_again_batch(ilqg_problem* p, int32_t batch, const void* x0, void* xs, void* us, void* P,
                                   void* alpha, void* total_costs, int32_t* iters, int32_t* status,
                                   int32_t* converged, void* workspace, int32_t augmented_lagrangian,
                                   const int32_t* active, void* stream) {
  ilqg_solve_options o;
  ilqg_default_solve_options(&o);
  o.augmented_lagrangian = augmented_lagrangian ? 1 : 0;
  o.resume = 1;
  o.active = active;
  return ilqg_solve_batch_ex(p, batch, x0, xs, us, P, alpha, total_costs, iters, status, converged, workspace, &o, stream);
}

ilqg_status ilqg_solve_state_batch(const ilqg_problem* p, int32_t batch, const void* workspace,
                                   int32_t augmented_lagrangian, void* last_merit, void* expected_decrease,
                                   void* step, int32_t* backtracks, void* stream) {
  if (!p || !workspace) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  const DevProblem& d = p->dev;
  const int ol_row = p->desc.params.open_loop ? ol_row_elems(d.n, d.m, d.N) : 0;
  const WsLayout L(d.n, d.m, d.N, d.T, d.pairs.Rsz, d.pairs.rsz, ol_row, d.num_constraints, augmented_lagrangian ? 1 : 0);
#define CALL(TY_)                                                                                                  \
  [&]() -> ilqg_status {                                                                                         \
    hipLaunchKernelGGL(solve_state_kernel<TY_>, dim3((batch + 63) / 64), dim3(64), 0, (hipStream_t)stream,        \
                       (const TY_*)workspace, L.total, L.state, batch, (TY_*)last_merit, (TY_*)expected_decrease,   \
                       (TY_*)step, backtracks);                                                                    \
    HIP_TRY(hipGetLastError());                                                                                  \
    return ILQG_OK;                                                                                              \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

ilqg_status ilqg_receding_horizon_shift_batch(const ilqg_problem* p, int32_t batch, const void* x0, double t0,
                                              double planner_runtime, double plan_t0, void* xs, void* us, void* P,
                                              void* alpha, void* x0_next, int32_t* first_step,
                                              double* new_plan_t0_host, void* stream) {
  if (!p || !x0 || !xs || !us || !P || !alpha || !x0_next || !first_step) return fail(ILQG_ERR_INVALID, "null argument");
  if (p->has_route_progress) return fail(ILQG_ERR_UNSUPPORTED, kRouteProgressReceding);
  const DevProblem& d = p->dev;
  const double dt = d.dt, horizon = dt * d.T;
  // the reference's CHECKs (src/problem.cpp:68-70)
  if (planner_runtime < 0.0 || planner_runtime + t0 > plan_t0 + horizon || t0 < plan_t0)
    return fail(ILQG_ERR_INVALID, "receding horizon: t0 / planner_runtime outside the stored plan");
  // SyncToExistingProblem's time bookkeeping (:75-102) is identical for every instance of this batch; the
  // kernel repeats it per instance, here it validates the call and reports the new plan's start time
  const float kRoundingError = 0.9f;
  const double relative_t0 = t0 - plan_t0;
  size_t current_timestep = static_cast<size_t>(relative_t0 / dt);
  double remaining = (current_timestep + 1) * dt - relative_t0;
  if (remaining < kRoundingError * dt) {
    current_timestep += 1;
    remaining = dt - remaining;
  }
  const size_t itn_step = static_cast<size_t>((relative_t0 + kSmallNumberF) / dt);  // IntegrateToNextTimeStep's own (:104-113)
  if (itn_step >= size_t(d.T)) return fail(ILQG_ERR_INVALID, "receding horizon: t0 past the last plan step");
  double new_t0 = t0 + remaining;
  int int_begin = int(current_timestep) + 1, int_end = int_begin;
  if (remaining <= planner_runtime) {
    const size_t num_steps = static_cast<size_t>(kSmallNumberF + (planner_runtime - remaining) / dt);
    int_end = int(current_timestep + num_steps);
    if (int_end < int_begin) int_end = int_begin;
    new_t0 += dt * double(num_steps);
  }
  if (int_end > d.T) return fail(ILQG_ERR_INVALID, "receding horizon: integration runs past the plan");
  if (new_plan_t0_host) *new_plan_t0_host = new_t0;
  if (batch <= 0) return ILQG_OK;
#define CALL(TY_)                                                                                                  \
  [&]() -> ilqg_status {                                                                                         \
    RecedingArgs<TY_> g{};                                                                                         \
    g.plan = PlanBuffers<TY_>{(TY_*)xs, (TY_*)us, (TY_*)P, (TY_*)alpha, nullptr, nullptr, plan_t0, d.T};          \
    g.x = (const TY_*)x0; g.t = t0; g.planner_runtime = planner_runtime;                                           \
    g.xs = (TY_*)xs; g.us = (TY_*)us; g.P = (TY_*)P; g.alpha = (TY_*)alpha; g.x0_next = (TY_*)x0_next;             \
    g.first_step = first_step;                                                                                     \
    hipLaunchKernelGGL(receding_sync_kernel<TY_>, dim3(batch), dim3(64), (d.n + d.m + 2) * sizeof(TY_),                \
                       (hipStream_t)stream, d, g);                                                                 \
    HIP_TRY(hipGetLastError());                                                                                  \
    return ILQG_OK;                                                                                              \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

static ilqg_status launch_strategy_costs(const ilqg_problem* p, int32_t batch, const void* x0, const void* xs,
                                         const void* us, const void* P, const void* alpha, double eps, int open_loop,
                                         int euler, int moves, void* costs, void* stream) {
  const DevProblem& d = p->dev;
#define CALL(TY_)                                                                                                  \
  [&]() -> ilqg_status {                                                                                         \
    StrategyCostArgs<TY_> g{(const TY_*)x0, (const TY_*)xs, (const TY_*)us, (const TY_*)P, (const TY_*)alpha,      \
                            (TY_*)costs, TY_(eps), open_loop, euler, batch};                                       \
    const size_t lds = quad_tables_bytes(d, sizeof(TY_)) + strategy_cost_lds_elems(d.n, d.m, d.num_terms) * sizeof(TY_); \
    hipLaunchKernelGGL(strategy_costs_kernel<TY_>, dim3(batch, moves), dim3(64), lds, (hipStream_t)stream, d, g);  \
    HIP_TRY(hipGetLastError());                                                                                  \
    return ILQG_OK;                                                                                              \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

ilqg_status ilqg_strategy_costs_batch(const ilqg_problem* p, int32_t batch, const void* x0, const void* xs,
                                      const void* us, const void* P, const void* alpha, int32_t open_loop,
                                      int32_t euler, void* costs, void* stream) {
  if (!p || !x0 || !xs || !us || !P || !alpha || !costs) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  return launch_strategy_costs(p, batch, x0, xs, us, P, alpha, 0.0, open_loop ? 1 : 0, euler ? 1 : 0, 1, costs, stream);
}

ilqg_status ilqg_check_local_nash_batch(const ilqg_problem* p, int32_t batch, const void* x0, const void* xs,
                                        const void* us, const void* P, const void* alpha, double max_perturbation,
                                        int32_t open_loop, int32_t* is_nash, void* margin, void* stream) {
  if (!p || !x0 || !xs || !us || !P || !alpha || !is_nash) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  const DevProblem& d = p->dev;
  const int moves = 1 + 2 * d.m * (d.T - 1);
  if (moves > 65535) return fail(ILQG_ERR_UNSUPPORTED, "too many perturbations for one launch");
  const size_t esz = p->desc.dtype == ILQG_F32 ? 4 : 8;
  ilqg_status s = Scratch().reserve(size_t(moves) * batch * d.N * esz);
  if (s != ILQG_OK) return s;
  // the reference switches the integrator to one-step Euler for this check (check_local_nash_equilibrium.cpp:75-79)
  s = launch_strategy_costs(p, batch, x0, xs, us, P, alpha, max_perturbation, open_loop ? 1 : 0, 1, moves, ilqg_shared::scratch_state().ptr,
                            stream);
  if (s != ILQG_OK) return s;
#define CALL(TY_)                                                                                                  \
  [&]() -> ilqg_status {                                                                                         \
    hipLaunchKernelGGL(nash_verdict_kernel<TY_>, dim3(batch), dim3(64), 0, (hipStream_t)stream, d,                 \
                       (const TY_*)ilqg_shared::scratch_state().ptr, moves, batch, is_nash, (TY_*)margin);                            \
    HIP_TRY(hipGetLastError());                                                                                  \
    return ILQG_OK;                                                                                              \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

ilqg_status ilqg_check_sufficient_nash_batch(const ilqg_problem* p, int32_t batch, const void* xs, const void* us,
                                             int32_t* is_psd, void* stream) {
  if (!p || !xs || !us || !is_psd) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  // PlayerCost::Quadraticize of every player at every step (:168-172), whatever the player's time structure: the
  // quadraticisation kernel is launched on a copy of the problem whose players are all time-additive
  ilqg_problem full = *p;
  for (int i = 0; i < full.dev.N; i++) full.dev.structure[i] = ILQG_SUM;
  const DevProblem& d = full.dev;
  const size_t esz = p->desc.dtype == ILQG_F32 ? 4 : 8;
  const size_t per_inst = size_t(d.T) * (size_t(d.N) * d.n * d.n + size_t(d.N) * d.n + d.pairs.Rsz + d.pairs.rsz) * esz;
  int chunk = int((size_t(256) << 20) / per_inst);
  if (chunk < 1) chunk = 1;
  if (chunk > batch) chunk = batch;
  ilqg_status s = Scratch().reserve(per_inst * chunk);
  if (s != ILQG_OK) return s;
  hipLaunchKernelGGL(fill_int_kernel<int>, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, is_psd, 1, batch);
  HIP_TRY(hipGetLastError());
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int nb = (batch - b0 < chunk) ? batch - b0 : chunk;
    char* base = (char*)ilqg_shared::scratch_state().ptr;
    char* Q = base;
    char* l = Q + size_t(nb) * d.T * d.N * d.n * d.n * esz;
    char* R = l + size_t(nb) * d.T * d.N * d.n * esz;
    char* r = R + size_t(nb) * d.T * d.pairs.Rsz * esz;
    const char* xs_c = (const char*)xs + size_t(b0) * d.T * d.n * esz;
    const char* us_c = (const char*)us + size_t(b0) * d.T * d.m * esz;
    s = launch_linquad(&full, nb, xs_c, us_c, nullptr, nullptr, nullptr, nullptr, nullptr, Q, l, R, r, nullptr, nullptr,
                       nullptr, stream);
    if (s != ILQG_OK) break;
#define CALL(TY_)                                                                                                  \
  [&]() -> ilqg_status {                                                                                         \
    hipLaunchKernelGGL(psd_check_kernel<TY_>, dim3(d.T, nb), dim3(64), 0, (hipStream_t)stream, d, (TY_*)Q, (TY_*)R, \
                       is_psd + b0);                                                                               \
    HIP_TRY(hipGetLastError());                                                                                  \
    return ILQG_OK;                                                                                              \
  }()
    s = DT_DISPATCH(p, CALL);
#undef CALL
    if (s != ILQG_OK) break;
  }
  // `full` is a shallow copy: it must not run the destructor logic of the handle it was copied from
  return s;
}

static ilqg_status check_plan(const ilqg_problem* p, int32_t plan_rows, const void* a, const void* b, const void* c,
                              const void* d, const void* e, const void* f) {
  if (!p || !a || !b || !c || !d || !e || !f) return fail(ILQG_ERR_INVALID, "null argument");
  if (plan_rows < p->dev.T) return fail(ILQG_ERR_INVALID, "plan_rows must be at least the problem's T");
  return ILQG_OK;
}

ilqg_status ilqg_plan_integrate_batch(const ilqg_problem* p, int32_t batch, int32_t plan_rows, const void* plan_xs,
                                      const void* plan_us, const void* plan_P, const void* plan_alpha,
                                      const int32_t* plan_len, const double* plan_t0, double t_from, double t_to,
                                      double must_contain, void* x, int32_t* active, void* stream) {
  ilqg_status s = check_plan(p, plan_rows, plan_xs, plan_us, plan_P, plan_alpha, plan_len, plan_t0);
  if (s != ILQG_OK) return s;
  if (!x || !active) return fail(ILQG_ERR_INVALID, "null argument");
  if (batch <= 0) return ILQG_OK;
  const DevProblem& d = p->dev;
#define CALL(TY_)                                                                                                  \
  [&]() -> ilqg_status {                                                                                         \
    PlanIntegrateArgs<TY_> g{};                                                                                    \
    g.plan = PlanBuffers<TY_>{(TY_*)plan_xs, (TY_*)plan_us, (TY_*)plan_P, (TY_*)plan_alpha, (int*)plan_len,       \
                              (double*)plan_t0, 0.0, plan_rows};                                                   \
    g.t_from = t_from; g.t_to = t_to; g.must_contain = must_contain; g.x = (TY_*)x; g.active = active;             \
    hipLaunchKernelGGL(plan_integrate_kernel<TY_>, dim3(batch), dim3(64), (d.n + d.m + 2) * sizeof(TY_),               \
                       (hipStream_t)stream, d, g);                                                                 \
    HIP_TRY(hipGetLastError());                                                                                  \
    return ILQG_OK;                                                                                              \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

ilqg_status ilqg_receding_horizon_sync_batch(const ilqg_problem* p, int32_t batch, int32_t plan_rows,
                                             const void* plan_xs, const void* plan_us, const void* plan_P,
                                             const void* plan_alpha, const int32_t* plan_len, const double* plan_t0,
                                             const void* x, double t, double planner_runtime, void* xs, void* us,
                                             void* P, void* alpha, void* x0_next, double* solve_t0,
                                             int32_t* first_step, int32_t* active, void* stream) {
  ilqg_status s = check_plan(p, plan_rows, plan_xs, plan_us, plan_P, plan_alpha, plan_len, plan_t0);
  if (s != ILQG_OK) return s;
  if (!x || !xs || !us || !P || !alpha || !x0_next || !solve_t0 || !first_step || !active)
    return fail(ILQG_ERR_INVALID, "null argument");
  if (xs == plan_xs || us == plan_us || P == plan_P || alpha == plan_alpha)
    return fail(ILQG_ERR_INVALID, "the next solve's buffers must not alias the stored plan");
  if (p->has_route_progress) return fail(ILQG_ERR_UNSUPPORTED, kRouteProgressReceding);
  if (batch <= 0) return ILQG_OK;
  const DevProblem& d = p->dev;
#define CALL(TY_)                                                                                                  \
  [&]() -> ilqg_status {                                                                                         \
    RecedingArgs<TY_> g{};                                                                                         \
    g.plan = PlanBuffers<TY_>{(TY_*)plan_xs, (TY_*)plan_us, (TY_*)plan_P, (TY_*)plan_alpha, (int*)plan_len,       \
                              (double*)plan_t0, 0.0, plan_rows};                                                   \
    g.x = (const TY_*)x; g.t = t; g.planner_runtime = planner_runtime;                                             \
    g.xs = (TY_*)xs; g.us = (TY_*)us; g.P = (TY_*)P; g.alpha = (TY_*)alpha; g.x0_next = (TY_*)x0_next;             \
    g.solve_t0 = solve_t0; g.first_step = first_step; g.active = active;                                           \
    hipLaunchKernelGGL(receding_sync_kernel<TY_>, dim3(batch), dim3(64), (d.n + d.m + 2) * sizeof(TY_),                \
                       (hipStream_t)stream, d, g);                                                                 \
    HIP_TRY(hipGetLastError());                                                                                  \
    return ILQG_OK;                                                                                              \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

ilqg_status ilqg_solution_splice_batch(const ilqg_problem* p, int32_t batch, int32_t plan_rows, void* plan_xs,
                                       void* plan_us, void* plan_P, void* plan_alpha, int32_t* plan_len,
                                       double* plan_t0, const void* xs, const void* us, const void* P,
                                       const void* alpha, const double* solve_t0, const int32_t* converged,
                                       const int32_t* active, void* stream) {
  ilqg_status s = check_plan(p, plan_rows, plan_xs, plan_us, plan_P, plan_alpha, plan_len, plan_t0);
  if (s != ILQG_OK) return s;
  if (!xs || !us || !P || !alpha || !solve_t0) return fail(ILQG_ERR_INVALID, "null argument");
  if (plan_rows < p->dev.T + 5)
    return fail(ILQG_ERR_INVALID, "plan_rows must leave room for five saved rows (T + 5)");
  if (batch <= 0) return ILQG_OK;
  const DevProblem& d = p->dev;
#define CALL(TY_)                                                                                                  \
  [&]() -> ilqg_status {                                                                                         \
    SpliceArgs<TY_> g{};                                                                                           \
    g.plan = PlanBuffers<TY_>{(TY_*)plan_xs, (TY_*)plan_us, (TY_*)plan_P, (TY_*)plan_alpha, plan_len, plan_t0,    \
                              0.0, plan_rows};                                                                     \
    g.xs = (const TY_*)xs; g.us = (const TY_*)us; g.P = (const TY_*)P; g.alpha = (const TY_*)alpha;               \
    g.solve_t0 = solve_t0; g.converged = converged; g.active = active;                                             \
    hipLaunchKernelGGL(splice_kernel<TY_>, dim3(batch), dim3(64), 0, (hipStream_t)stream, d, g);                   \
    HIP_TRY(hipGetLastError());                                                                                  \
    return ILQG_OK;                                                                                              \
  }()
  return DT_DISPATCH(p, CALL);
#undef CALL
}

}  // extern "C"
#endif  // !ILQG_PART_NX
