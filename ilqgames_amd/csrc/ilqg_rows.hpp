// ilqg_rows.hpp — linearise + quadraticise with ONE LANE PER TIME STEP (gfx950).
//
// Computes, for a chunk of consecutive time steps ("rows") of one game instance, what
//   ILQSolver::ComputeLinearization (src/ilq_solver.cpp:437-455),
//   ILQSolver::ComputeCostQuadraticization (:471-490) -> PlayerCost::Quadraticize (src/player_cost.cpp:194-225),
//   the per-step pieces of ILQSolver::MeritFunction (:400-435) and ILQSolver::TotalCosts (:220-257)
// compute, one row at a time, in the reference.
//
// Mapping.  The rows of a trajectory are independent of each other and run the SAME cost list, so lane r of the
// wavefront takes row k0 + r and the wave walks the cost list once for all of them: the term being evaluated, its
// kind, indices, weights and the polyline segments are wave-uniform — they are read through the scalar data cache
// (s_load from the row program in constant memory) and live in scalar registers, branches on them are scalar
// branches — and no lane diverges from its neighbours except where a cost is inactive at its row.  A lane meets the
// terms of a player in PlayerCost::Quadraticize's order, so every Hessian / gradient entry is accumulated in the
// reference's order by construction.  (The first design of this stage gave a lane to each cost TERM of one row:
// ~1000 instructions and ~16 k cycles per row, every cost kind's branch executed once per row whatever the number of
// lanes in it.)
//
// The images [A | B | Q_i | l_i | R_ij | r_ij] of 64 rows do not fit the LDS, but cost Hessians are sparse: the host
// lists the image words any term (or a Jacobian) can touch — "slots" — and a lane accumulates its row's slots in
// LDS, acc[slot][lane].  Writing a row out is then a gather: every lane owns fixed words of the row's image, looks up
// "which slot (or constant) is my word" once per chunk, and the wave streams the dense rows with full-width stores.
// The work is cut into passes (the Jacobians, then one per player) so that the slots of one player's
// Q_i, l_i, R_i*, r_i* are all that is live at a time.  The "row program" (passes, ops, slot lists, word maps) is built
// by ilqg_problem_create (ilqg_rowprog.hpp).
#pragma once

#include <type_traits>
#include <utility>

#include "ilqg_common.hpp"
#include "ilqg_models.hpp"

#ifndef ILQG_ROWS_XREG
#define ILQG_ROWS_XREG 1  // the split row kernels keep a row's state in registers (0: the LDS image, for A/B measurements)
#endif
#ifndef ILQG_PROFILE2
#define ILQG_PROFILE2 0  // diagnostic: per-op phase stamps instead of the per-pass ones
#endif

namespace ilqg {

// no floating-point contraction in the row stage (see ilqg_models.hpp: every form of it must round alike)
#pragma clang fp contract(off)

// Loads through this pointer type are scalar loads whenever the address is wave-uniform (the constant address
// space: the program is written by the host before any kernel runs).
typedef const __attribute__((address_space(4))) int* rp_cptr;
template <typename T> struct ConstPtr;
template <> struct ConstPtr<float> { typedef const __attribute__((address_space(4))) float* type; };
template <> struct ConstPtr<double> { typedef const __attribute__((address_space(4))) double* type; };

// ---- row program: a flat int32 buffer; header words give the offsets of its tables ----
enum {
  RP_NUM_PASSES = 0, RP_NUM_PSLOTS, RP_MAX_LSLOTS, RP_OFF_PASS, RP_OFF_OPS, RP_OFF_SIDS, RP_OFF_PINIT, RP_OFF_LINIT,
  RP_OFF_REGIONS, RP_OFF_MAPS, RP_MAPS_WORDS, RP_OFF_MERIT, RP_WORDS, RP_OFF_COMPACT, RP_HEADER = 16
};
enum { RPASS_WORDS = 9, ROP_FIELDS = 20, ROP_INLINE_SIDS = 20, ROP_WORDS = 40, RREG_WORDS = 4, RMERIT_WORDS = 4, RINIT_WORDS = 2 };
// An op is self-contained: [mode, first slot id, slot ids, aux | the fields of the term it evaluates | its first
// ROP_INLINE_SIDS slot ids] — one batch of scalar loads per op (a PAT_ALL term on a wide vector reads the rest of its
// slot ids from the list at RO_SID).
//   TERM       one top-level cost / constraint
//   EXT_EVAL   child `aux` of an ExtremeValueCost: its value competes for the extreme (no derivatives yet)
//   EXT_APPLY  the same child again: the lanes whose extreme it is take its derivatives
//   JACOBIAN   one subsystem's Linearize (kind = dynamics kind, idx0 / idx1 = its state / control offsets,
//              weight / value = its parameter and the next subsystem's)
//   CLOSEST    Polyline2::ClosestPoint of (x[idx0], x[idx1]) to the polyline whose segments start at `poly_first`
//              (`poly_nseg` of them); the polyline terms that follow read the result
// An ExtremeValueCost becomes EXT_EVAL x children, EXT_APPLY x children; role / player / constraint slot / first
// active step of those ops are the parent's.
//   AFFINE     an AffineScalarConstraint / AffineVectorConstraint: dense over its whole argument vector, evaluated and
//              scattered by rows_affine from its coefficient block (`poly_first` = the block's offset); its slot ids are
//              [G(0..d-1) | H(r, c) column-major (d^2) | 2 d scratch slots (the vector constraint's delta and A^T delta)]
enum { ROP_TERM = 0, ROP_EXT_EVAL = 1, ROP_EXT_APPLY = 2, ROP_JACOBIAN = 3, ROP_CLOSEST = 4, ROP_AFFINE = 5 };
enum {
  RO_MODE = 0, RO_SID, RO_NSID, RO_AUX, RO_KIND, RO_ROLE, RO_PLAYER, RO_FLAGS, RO_IDX0, RO_IDX1, RO_IDX2, RO_IDX3,
  RO_WEIGHT, RO_VALUE, RO_POLY_FIRST, RO_SLOT, RO_ARG_OFF, RO_ARG_DIM, RO_K_START, RO_PATTERN_NSEG
};
// how a slot starts a chunk (persistent slots) or a pass (pass-local slots): [kind | player << 8 | from_cost << 16,
// value (float bits)]; RI_CREG is the value only where the reference would have created the control block
enum { RI_VALUE = 0, RI_DT = 2, RI_NEG_DT = 3, RI_CREG = 5 };
enum { RPASS_JACOBIANS = 0, RPASS_PLAYER = 1 };
constexpr int kRowSlotZero = 0;  // persistent slot 0 is always the constant zero

// The transposed argument image of a chunk: entry e of row r at arg[e * cw + r].
template <typename T>
struct RowArg {
  const T* base;  // &arg[first entry * cw + this lane's row]
  int stride;
  __device__ __forceinline__ T operator[](int i) const { return base[i * stride]; }
};

// The same with the row's (x, u) held in registers (rows_chunk<..., XREG>: n, m <= 16, compile-time dimensions) and no
// LDS image at all.  A term's argument is the state or one player's control vector; the (up to four) entries its
// descriptor names — the first four, for a cost over the whole vector — are picked out of this lane's register copy
// when the op is decoded (wave-uniform register indices) and the accessor hands them out by index; any other entry (a
// cost over a long vector, an affine constraint) is read from the trajectory in memory.
template <typename T>
struct MixArg {
  T a0, a1, a2, a3;   // entries id0 .. id3 of the argument vector at this lane's row
  int id0, id1, id2, id3;
  const T* g;         // the argument vector of this lane's row in memory
  __device__ __forceinline__ T operator[](int i) const {
    if (i == id0) return a0;
    if (i == id1) return a1;
    if (i == id2) return a2;
    if (i == id3) return a3;
    return g[i];
  }
};
template <typename T, int W> struct RowVecN;
template <typename T> struct RowVecN<T, 8> { typedef T type __attribute__((ext_vector_type(8))); };
template <typename T> struct RowVecN<T, 16> { typedef T type __attribute__((ext_vector_type(16))); };
constexpr int rows_vec_regs(int dim) { return dim <= 8 ? 8 : 16; }  // elements of the register copy of a dim-vector

// The accessor of a term's argument vector (entries off .. of this lane's row; id0 .. id3: the entries the op names).
// The register copy itself never goes into a struct: a vector inside a stack object is read back through scratch.
template <typename T, bool XREG, int XW, int UW>
__device__ __forceinline__ typename std::conditional<XREG, MixArg<T>, RowArg<T>>::type rows_make_arg(
    typename RowVecN<T, XW>::type xrow, typename RowVecN<T, UW>::type urow, const T* xg, const T* ug, const T* arg, int cw,
    int cn, int off, int rl, int id0, int id1, int id2, int id3) {
  if constexpr (XREG) {
    const bool in_u = off >= cn;
    const int o = in_u ? off - cn : off;
    if (id0 < 0) {  // a term over its whole argument vector walks it from the front
      id0 = 0; id1 = 1; id2 = 2; id3 = 3;
    }
    MixArg<T> m;
    m.id0 = id0; m.id1 = id1; m.id2 = id2; m.id3 = id3;
    const int e0 = o + id0, e1 = o + (id1 < 0 ? 0 : id1), e2 = o + (id2 < 0 ? 0 : id2), e3 = o + (id3 < 0 ? 0 : id3);
    if (in_u) {
      m.a0 = urow[e0 & (UW - 1)]; m.a1 = urow[e1 & (UW - 1)]; m.a2 = urow[e2 & (UW - 1)]; m.a3 = urow[e3 & (UW - 1)];
    } else {
      m.a0 = xrow[e0 & (XW - 1)]; m.a1 = xrow[e1 & (XW - 1)]; m.a2 = xrow[e2 & (XW - 1)]; m.a3 = xrow[e3 & (XW - 1)];
    }
    m.g = (in_u ? ug : xg) + o;
    return m;
  } else {
    return RowArg<T>{arg + off * cw + rl, cw};
  }
}

// LDS of one wave of this stage: [arg (n + m) x cw | acc (persistent + pass-local slots) x (cw + 1)], and the word
// maps, shared by the waves of a workgroup
__host__ __device__ inline size_t rows_lds_elems(int n, int m, int num_pslots, int max_lslots, int cw) {
  return size_t(n + m) * cw + size_t(num_pslots + max_lslots) * (cw + 1);
}
// ... of the variant that keeps the (x, u) rows in registers (the split row kernels of the shapes with n, m <= 16)
__host__ __device__ constexpr bool rows_state_in_registers(int cn, int cm) {
  return cn > 0 && cn <= 16 && cm <= 16 && ILQG_ROWS_XREG;
}
__host__ __device__ inline size_t rows_lds_elems_xreg(int num_pslots, int max_lslots, int cw) {
  return size_t(num_pslots + max_lslots) * (cw + 1);
}
// ... and of a merit-only evaluation on top of that (rows_chunk<..., XREG, GRAD_ONLY>): gradient slots only
__host__ __device__ inline size_t rows_lds_elems_grad(int num_pslots, int max_gslots, int cw) {
  return size_t(num_pslots + max_gslots) * (cw + 1);
}
// Largest chunk width (64, 32 or 16 rows) whose scratch fits `budget` bytes.
__host__ __device__ inline int rows_chunk_width(int n, int m, int num_pslots, int max_lslots, size_t elem, size_t budget) {
  for (int cw = 64; cw > 16; cw >>= 1)
    if (rows_lds_elems(n, m, num_pslots, max_lslots, cw) * elem <= budget) return cw;
  return 16;
}

// Largest vector width (elements) that keeps every row's block of `unit`-element granularity 16 / 8 / 4-byte aligned.
template <typename T>
constexpr int rows_vec_width(int unit) {
  constexpr int full = 16 / int(sizeof(T));
  return (unit % full == 0) ? full : ((unit % 2 == 0 && full >= 2) ? 2 : 1);
}

template <typename T, int VW> struct RowVec;
template <typename T> struct RowVec<T, 1> { typedef T type; };
template <typename T> struct RowVec<T, 2> { typedef T type __attribute__((ext_vector_type(2))); };
template <typename T> struct RowVec<T, 4> { typedef T type __attribute__((ext_vector_type(4))); };

// Streams rows [0, nrows) of one output block: W words per row at g0 + r * stride, word w of a row from
// acc[map[w]][r].  Every lane fetches the slots of its words once; the LDS reads of RB rows are issued together, then
// their stores (one exposed LDS round trip per RB rows instead of one per store).
template <typename T, int W, int VW>
__device__ __forceinline__ void rows_writeout(T* g0, size_t stride, const short* map, const T* acc, int cws, int nrows,
                                              int lane) {
  constexpr int NST = (W + 64 * VW - 1) / (64 * VW);
  constexpr int RB = (NST * VW <= 4) ? 4 : 2;
  typedef typename RowVec<T, VW>::type vec;
  int off[NST][VW];
#pragma unroll
  for (int j = 0; j < NST; j++)
#pragma unroll
    for (int q = 0; q < VW; q++) {
      const int w = (j * 64 + lane) * VW + q;
      off[j][q] = (w < W ? int(map[w]) : kRowSlotZero) * cws;
    }
#pragma unroll 1
  for (int r0 = 0; r0 < nrows; r0 += RB) {
    T val[RB][NST][VW];
#pragma unroll
    for (int rr = 0; rr < RB; rr++) {
      const int r = r0 + rr < nrows ? r0 + rr : nrows - 1;
#pragma unroll
      for (int j = 0; j < NST; j++)
#pragma unroll
        for (int q = 0; q < VW; q++) val[rr][j][q] = acc[off[j][q] + r];
    }
#pragma unroll
    for (int rr = 0; rr < RB; rr++) {
      if (r0 + rr < nrows) {
        T* g = g0 + size_t(r0 + rr) * stride;
#pragma unroll
        for (int j = 0; j < NST; j++) {
          const int w0 = (j * 64 + lane) * VW;
          if (w0 < W) {
            if constexpr (VW == 1) {
              g[w0] = val[rr][j][0];
            } else {
              vec v;
#pragma unroll
              for (int q = 0; q < VW; q++) v[q] = val[rr][j][q];
              *reinterpret_cast<vec*>(g + w0) = v;
            }
          }
        }
      }
    }
  }
}
// A block of any size known at run time only (the run-time-dimensioned path, CN = 0 in rows_chunk): 64 words at a time.
template <typename T>
__device__ __forceinline__ void rows_writeout_small(T* g0, size_t stride, int W, const short* map, const T* acc, int cws,
                                                    int nrows, int lane);
template <typename T>
__device__ __forceinline__ void rows_writeout_rt(T* g0, size_t stride, int W, const short* map, const T* acc, int cws,
                                                 int nrows, int lane) {
  for (int w0 = 0; w0 < W; w0 += 64) rows_writeout_small<T>(g0 + w0, stride, W - w0 < 64 ? W - w0 : 64, map + w0, acc, cws, nrows, lane);
}
// The same for a block of at most 64 words whose size is only known at run time (l_i, R_ij, r_ij).
template <typename T>
__device__ __forceinline__ void rows_writeout_small(T* g0, size_t stride, int W, const short* map, const T* acc, int cws,
                                                    int nrows, int lane) {
  constexpr int RB = 8;
  const int off = (lane < W ? int(map[lane]) : kRowSlotZero) * cws;
#pragma unroll 1
  for (int r0 = 0; r0 < nrows; r0 += RB) {
    T val[RB];
#pragma unroll
    for (int rr = 0; rr < RB; rr++) val[rr] = acc[off + (r0 + rr < nrows ? r0 + rr : nrows - 1)];
#pragma unroll
    for (int rr = 0; rr < RB; rr++)
      if (r0 + rr < nrows && lane < W) g0[size_t(r0 + rr) * stride + lane] = val[rr];
  }
}

// Adds one term's contribution to this lane's slots.  `sid` (scalar memory) lists the slots in the order of the
// term's pattern:
//   SINGLE  G(d), H(d,d)
//   PAIR2   G(x), G(y), H(x,x), H(y,y), H(x,y), H(y,x)
//   PAIR4   G(x1), G(x2), G(y1), G(y2), then H(x1,x1) (x1,x2) (x2,x1) (x2,x2), the same four for y, then
//           H(x1,y1) (y1,x1) (x1,y2) (y2,x1) (x2,y1) (y1,x2) (x2,y2) (y2,x2)  — with s = (+,+,-,-) over (x1,y1,x2,y2):
//           G[p] += s_p g_type(p), H(p,q) += s_p s_q h_type(p),type(q)
//   ALL     G(0..dim-1), H(0,0)..H(dim-1,dim-1)
// The slots of one leaf are distinct (build_row_program checks), so the read-modify-writes of a term do not depend
// on each other: all reads, then all adds, then all writes — one LDS round trip per term instead of one per entry.
// Where an op's slot ids are read from: the op's own descriptor words, held one word per lane in a vector register and
// read through v_readlane (the index is wave-uniform), or — ops with more ids than fit inline — the program's id table
// in scalar memory.
struct SidsInline {
  int words, base;
  __device__ __forceinline__ int operator[](int e) const { return __builtin_amdgcn_readlane(words, base + e); }
};
struct SidsTable {
  rp_cptr ptr;
  __device__ __forceinline__ int operator[](int e) const { return ptr[e]; }
};

template <typename T, typename V, typename SID>
__device__ __forceinline__ void rows_scatter(int pattern, const TermOut<T>& o, const SID& sid, int nsid, T* col, int cws,
                                             const V& v, bool want_h) {
  if (pattern == PAT_SINGLE) {
    T* const p0 = col + sid[0] * cws;
    T* const p1 = col + sid[1] * cws;
    const T a0 = *p0, a1 = *p1;
    *p0 = a0 + o.gx;
    if (want_h) *p1 = a1 + o.hxx;
  } else if (pattern == PAT_PAIR2) {
    T* pp[6];
    T av[6];
#pragma unroll
    for (int e = 0; e < 6; e++) pp[e] = col + sid[e] * cws;
#pragma unroll
    for (int e = 0; e < 6; e++) av[e] = *pp[e];
    *pp[0] = av[0] + o.gx;
    *pp[1] = av[1] + o.gy;
    if (want_h) {
      *pp[2] = av[2] + o.hxx;
      *pp[3] = av[3] + o.hyy;
      *pp[4] = av[4] + o.hxy;
      *pp[5] = av[5] + o.hxy;
    }
  } else if (pattern == PAT_PAIR4) {
    {
      T* pp[4];
      T av[4];
#pragma unroll
      for (int e = 0; e < 4; e++) pp[e] = col + sid[e] * cws;
#pragma unroll
      for (int e = 0; e < 4; e++) av[e] = *pp[e];
      *pp[0] = av[0] + o.gx;
      *pp[1] = av[1] - o.gx;
      *pp[2] = av[2] + o.gy;
      *pp[3] = av[3] - o.gy;
    }
    if (want_h) {
      T* pp[16];
      T av[16];
#pragma unroll
      for (int e = 0; e < 16; e++) pp[e] = col + sid[4 + e] * cws;
#pragma unroll
      for (int e = 0; e < 16; e++) av[e] = *pp[e];
      const T add[16] = {o.hxx, -o.hxx, -o.hxx, o.hxx, o.hyy, -o.hyy, -o.hyy, o.hyy,
                         o.hxy, o.hxy, -o.hxy, -o.hxy, -o.hxy, -o.hxy, o.hxy, o.hxy};
#pragma unroll
      for (int e = 0; e < 16; e++) *pp[e] = av[e] + add[e];
    }
  } else if (pattern == PAT_ALL) {
    const int dim = nsid >> 1;
    for (int i = 0; i < dim; i++) {
      col[sid[i] * cws] += o.gx * (v[i] - o.gy);
      if (want_h) col[sid[dim + i] * cws] += o.gx;
    }
  } else if (pattern == PAT_CROSS4 && want_h) {
    //   H(p1,v1) (p1,v2) (p2,v1) (p2,v2) (v1,p1) (v1,p2) (v2,p1) (v2,p2)
    T* pp[8];
    T av[8];
#pragma unroll
    for (int e = 0; e < 8; e++) pp[e] = col + sid[e] * cws;
#pragma unroll
    for (int e = 0; e < 8; e++) av[e] = *pp[e];
    const T add[8] = {o.hxx, o.hyy, -o.hxx, -o.hyy, o.hxx, -o.hxx, o.hyy, -o.hyy};
#pragma unroll
    for (int e = 0; e < 8; e++) *pp[e] = av[e] + add[e];
  }
}

// The two affine constraints (constraint/affine_scalar_constraint.h:70-88, affine_vector_constraint.h:77-101): the
// augmented-Lagrangian term's gradient and DENSE Hessian over the whole argument vector of this lane's row, added to the
// op's slots.  `blk`: the constraint's coefficient block (uniform address); `v`: the row's argument vector; the vector
// constraint keeps delta = A v - b and A^T delta in its 2 d scratch slots (LDS, this lane's column) instead of registers.
template <typename T, typename V, typename SID>
__device__ __forceinline__ void rows_affine(int kind, bool is_equality, const T* blk, int d, const SID& sid, T* col, int cws,
                                            const V& v, T lambda, T mu, bool want_h) {
  if (kind == ILQG_CONSTRAINT_AFFINE_SCALAR) {
    const T b = blk[d];
    T s = T(0);
    for (int i = 0; i < d; i++) s += blk[i] * v[i];
    const T mu_eff = constraint_mu<T>(lambda, s - b, mu, is_equality);
    for (int i = 0; i < d; i++) {
      // hess_of_sq_ = a a^T is a matrix of the reference's scalars: the product a_i a_j is rounded before it meets x
      const T ai = blk[i];
      T hx = T(0);
      for (int j = 0; j < d; j++) {
        const T aij = ai * blk[j];
        hx += aij * v[j];
        if (want_h) col[sid[d + i + d * j] * cws] += mu_eff * aij;
      }
      col[sid[i] * cws] += lambda * ai + mu_eff * (hx - b * ai);
    }
    return;
  }
  const T* const bb = blk + d * d;
  const T* const ata = bb + d;
  const T* const aat = ata + d * d;
  T* const t_delta = col;  // scratch slot q of this lane: col[sid[d + d * d + q] * cws]
  const int s0 = d + d * d;
  T sq = T(0);
  for (int i = 0; i < d; i++) {
    T dlt = T(0);
    for (int j = 0; j < d; j++) dlt += blk[i + d * j] * v[j];
    dlt -= bb[i];
    t_delta[sid[s0 + i] * cws] = dlt;
    sq += dlt * dlt;
  }
  const T value = t_sqrt(sq);
  for (int i = 0; i < d; i++) {  // A^T delta
    T atd = T(0);
    for (int j = 0; j < d; j++) atd += blk[j + d * i] * t_delta[sid[s0 + j] * cws];
    t_delta[sid[s0 + d + i] * cws] = atd;
  }
  const T mu_eff = constraint_mu<T>(lambda, value, mu, is_equality);
  const T lv = lambda / value;
  for (int i = 0; i < d; i++) {
    const T atd_i = t_delta[sid[s0 + d + i] * cws];
    col[sid[i] * cws] += (mu_eff + lv) * atd_i;
    if (want_h)
      for (int j = 0; j < d; j++) {
        const T atd_j = t_delta[sid[s0 + d + j] * cws];
        col[sid[d + i + d * j] * cws] += lv * (aat[i + d * j] - atd_i * atd_j / (value * value)) + mu_eff * ata[i + d * j];
      }
  }
}

// The scatter pattern a cost kind produces when it is active (term_compute_leaf), known without evaluating it.
__host__ __device__ inline int term_pattern_of(int kind, int idx0) {
  switch (kind) {
    case ILQG_COST_QUADRATIC: return idx0 >= 0 ? PAT_SINGLE : PAT_ALL;
    case ILQG_COST_SEMIQUADRATIC:
    case ILQG_COST_ORIENTATION:
    case ILQG_COST_NOMINAL_PATH_LENGTH:
    case ILQG_CONSTRAINT_SINGLE_DIMENSION: return PAT_SINGLE;
    case ILQG_COST_QUADRATIC_POLYLINE2:
    case ILQG_COST_SEMIQUADRATIC_POLYLINE2:
    case ILQG_COST_POLYLINE2_SIGNED_DISTANCE:
    case ILQG_COST_QUADRATIC_NORM:
    case ILQG_COST_SEMIQUADRATIC_NORM:
    case ILQG_COST_CURVATURE:
    case ILQG_COST_ROUTE_PROGRESS:
    case ILQG_CONSTRAINT_POLYLINE2_SIGNED_DISTANCE: return PAT_PAIR2;
    case ILQG_COST_PROXIMITY:
    case ILQG_COST_SIGNED_DISTANCE:
    case ILQG_COST_QUADRATIC_DIFFERENCE:
    case ILQG_COST_RELATIVE_DISTANCE:
    case ILQG_COST_LOCALLY_CONVEX_PROXIMITY:
    case ILQG_COST_WEIGHTED_CONVEX_PROXIMITY:
    case ILQG_CONSTRAINT_PROXIMITY: return PAT_PAIR4;
    case ILQG_INTERNAL_WCP_SPEED: return PAT_PAIR2;
    case ILQG_INTERNAL_WCP_CROSS_X:
    case ILQG_INTERNAL_WCP_CROSS_Y: return PAT_CROSS4;
  }
  return PAT_NONE;
}
__host__ __device__ inline bool term_is_polyline(int kind) {
  return kind == ILQG_COST_QUADRATIC_POLYLINE2 || kind == ILQG_COST_SEMIQUADRATIC_POLYLINE2 ||
         kind == ILQG_COST_POLYLINE2_SIGNED_DISTANCE || kind == ILQG_CONSTRAINT_POLYLINE2_SIGNED_DISTANCE;
}

// Polyline2::ClosestPoint (src/polyline2.cpp:105-174) with the segment table in scalar memory: the same scan as
// polyline_closest (ilqg_models.hpp), every lane its own query point against the same segments.
template <typename T>
__device__ __forceinline__ Seg<T> load_seg_const(typename ConstPtr<T>::type s) {
  Seg<T> o;
  o.p1x = s[0]; o.p1y = s[1]; o.p2x = s[2]; o.p2y = s[3]; o.len = s[4]; o.ux = s[5]; o.uy = s[6];
  return o;
}
template <typename T>
__device__ __forceinline__ Closest<T> polyline_closest_rows(typename ConstPtr<T>::type segs, int first, int nseg, T qx,
                                                            T qy) {
  typename ConstPtr<T>::type base = segs + first * kSegStride;
  Closest<T> out;
  T best = dinf<T>();
  out.cx = T(0);
  out.cy = T(0);
  out.is_vertex = false;
  out.seg = load_seg_const<T>(base);
#pragma unroll 1
  for (int c = 0; c < nseg; c++) {
    const Seg<T> s = load_seg_const<T>(base + c * kSegStride);
    T px, py, cur;
    bool se;
    seg_closest(s, qx, qy, &px, &py, &se, &cur);
    if (t_abs(cur) < t_abs(best)) {
      const bool at2 = (px == s.p2x && py == s.p2y);
      const bool at1 = (px == s.p1x && py == s.p1y);
      if (se && (c > 0 || at2) && (c < nseg - 1 || at1)) {
        // the "shortcut" segment of the vertex (both candidates are scalar data; the lane picks)
        const Seg<T> s1 = load_seg_const<T>(base + c * kSegStride + 7), s2 = load_seg_const<T>(base + c * kSegStride + 14);
        Seg<T> sc;
        sc.p1x = at1 ? s1.p1x : s2.p1x; sc.p1y = at1 ? s1.p1y : s2.p1y;
        sc.ux = at1 ? s1.ux : s2.ux; sc.uy = at1 ? s1.uy : s2.uy;
        cur *= seg_side(sc, qx, qy) ? sgn(cur) : -sgn(cur);
      }
      best = cur;
      out.cx = px;
      out.cy = py;
      out.is_vertex = se;
      out.seg = s;
    }
  }
  out.ssd = best;
  const Seg<T> s0 = load_seg_const<T>(base), sl = load_seg_const<T>(base + (nseg - 1) * kSegStride);
  const T ax = out.cx - s0.p1x, ay = out.cy - s0.p1y;
  const T bx = out.cx - sl.p2x, by = out.cy - sl.p2y;
  out.is_endpoint = (ax * ax + ay * ay < T(1e-4f)) || (bx * bx + by * by < T(1e-4f));
  return out;
}

// ---- where a chunk reads its program from ----
// ProgDynamic: the row program built by ilqg_problem_create, in scalar memory — the interpreter: one loop over the ops,
// every field of an op read out of its descriptor at run time.
// ProgStatic<ID>: the STRUCTURE of the program (passes, op modes, kinds, indices, flags, slot ids, merit lists, compact
// bases: everything but floating-point parameters and polyline ranges) is a compile-time table (StaticRowProg<ID>,
// ilqg_rowprog_static.hpp, generated from build_row_program's output for a known problem structure): the op loop is
// unrolled into straight-line code with every kind switch folded, every slot an LDS immediate offset and every state
// entry a direct register read.  The PARAMETERS (weights, nominal values, thresholds, regularisation, polyline segment
// ranges) are still read from the run-time program, so any problem with the same structure — other weights, other lane
// geometry — runs the same code.  ilqg_problem_create matches the program it built against the registered structures
// word for word (parameters masked, row_program_mask_parameters); anything else runs the interpreter.  Both produce the
// same bits: the same expressions in the same order on the same data (tests/test_gpu_parity.py).
// The scratch of a static chunk has FIXED strides, so that every slot and every entry of the (x, u) image is an LDS
// immediate offset: kStaticRowStride elements between slots (the 64-row chunk's width + 1), 64 between image entries —
// whatever the number of rows the chunk really holds (any width up to 64).
constexpr int kStaticRowStride = 65;
struct ProgDynamic { static constexpr bool STATIC = false; };
template <int ID> struct StaticRowProg;  // { static constexpr int kWords, w[kWords]; } per registered structure
template <int ID> struct ProgStatic {
  static constexpr bool STATIC = true;
  typedef StaticRowProg<ID> S;
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// An op's descriptor.  field<F>(): a structural field (compile-time under ProgStatic); param<F>(): a parameter field
// (always run time); sid(e): the e-th inline slot id.
struct OpDynamic {
  int words;  // the descriptor, word w in lane w
  template <int F> __device__ __forceinline__ int field() const { return __builtin_amdgcn_readlane(words, F); }
  template <int F> __device__ __forceinline__ int param() const { return __builtin_amdgcn_readlane(words, F); }
  __device__ __forceinline__ SidsInline sids() const { return SidsInline{words, ROP_FIELDS}; }
};
template <class S, int OP>
struct SidsStatic {
  __device__ __forceinline__ constexpr int operator[](int e) const { return S::w[S::w[RP_OFF_OPS] + OP * ROP_WORDS + ROP_FIELDS + e]; }
};
template <class S, int OP>
struct OpStatic {
  rp_cptr rt;  // this op's descriptor in the run-time program (parameters)
  template <int F> __device__ __forceinline__ constexpr int field() const {
    constexpr int v = S::w[S::w[RP_OFF_OPS] + OP * ROP_WORDS + F];
    return v;
  }
  template <int F> __device__ __forceinline__ int param() const { return rt[F]; }
  __device__ __forceinline__ SidsStatic<S, OP> sids() const { return SidsStatic<S, OP>{}; }
};
struct PassDynamic {
  int op_begin, op_end, reg_begin, reg_end, li_begin, li_count, pkind, player, li_grad;
};
template <class S, int PS>
struct PassStatic {
  static constexpr int b = S::w[RP_OFF_PASS] + PS * RPASS_WORDS;
  static constexpr int op_begin = S::w[b + 0], op_end = S::w[b + 1], reg_begin = S::w[b + 2], reg_end = S::w[b + 3],
                       li_begin = S::w[b + 4], li_count = S::w[b + 5], pkind = S::w[b + 6], player = S::w[b + 7],
                       li_grad = S::w[b + 8];
};

// Can a static chunk keep its accumulators in REGISTERS?  Every slot id of every op must be a compile-time constant (an
// op whose ids spill into the program's id table, or an affine constraint, reads them at run time) and the slots of a
// pass must fit the register file beside the row's (x, u).  Then the slots are a per-lane register array — a lane is a
// row, so nothing is shared — the ops update registers instead of read-modify-writing LDS, and a pass's compact row
// is stored by its own lane straight from them (contiguous slots: vector stores), with no transposed LDS read-out.
// Such a kernel only exists for solves on compact rows (nothing dense is written by it): the launcher checks.
template <class S>
constexpr bool static_prog_in_registers() {
  const int nops = (S::w[RP_OFF_SIDS] - S::w[RP_OFF_OPS]) / ROP_WORDS;
  for (int op = 0; op < nops; op++) {
    const int b = S::w[RP_OFF_OPS] + op * ROP_WORDS;
    if (S::w[b + RO_MODE] == ROP_AFFINE || S::w[b + RO_NSID] > ROP_INLINE_SIDS) return false;
  }
  return S::w[RP_NUM_PSLOTS] + S::w[RP_MAX_LSLOTS] <= 64 && S::w[S::w[RP_OFF_COMPACT] + RC_W] > 0;
}
// The passes of a static program cut into two sets of about equal work (ops per pass, greedily, largest first): the
// fused trial kernel gives the chunk that can only start when the rollout has ended to BOTH its waves.  Returns the
// first set as a bit mask over the passes.
template <class S>
constexpr unsigned static_prog_pass_split() {
  const int np = S::w[RP_NUM_PASSES];
  int wgt[32] = {0};
  for (int ps = 0; ps < np && ps < 32; ps++) {
    const int b = S::w[RP_OFF_PASS] + ps * RPASS_WORDS;
    wgt[ps] = S::w[b + 1] - S::w[b + 0];
  }
  unsigned first = 0, taken = 0;
  int wa = 0, wb = 0;
  for (int k = 0; k < np && k < 32; k++) {
    int best = -1;
    for (int ps = 0; ps < np && ps < 32; ps++)
      if (!((taken >> ps) & 1u) && (best < 0 || wgt[ps] > wgt[best])) best = ps;
    taken |= 1u << best;
    if (wa <= wb) { first |= 1u << best; wa += wgt[best]; } else { wb += wgt[best]; }
  }
  return first;
}
template <class PROG, bool ST = PROG::STATIC> struct ProgInRegisters { static constexpr bool value = false; static constexpr int slots = 1; };
template <class PROG> struct ProgInRegisters<PROG, true> {
  static constexpr bool value = static_prog_in_registers<typename PROG::S>();
  static constexpr int slots = value ? PROG::S::w[RP_NUM_PSLOTS] + PROG::S::w[RP_MAX_LSLOTS] : 1;
};

// One chunk: rows [k0, k0 + nrows), nrows <= cw, executed by ONE wavefront (`lane` of 64) with its own LDS `sm`
// (rows_lds_elems for this cw: 64, 32 or 16).  `maps` is the workgroup's LDS copy of the program's word maps
// (rows_maps_load).  What is produced follows QuadArgs: A / Bm (null: skip the Jacobians), Q / l / R / r (null: not
// written; derivatives are still accumulated when merit_part is set), merit_part, cost_part.
// CN, CM, CNP: the problem's state / total control dimension / player count at compile time, or all 0 — the
// run-time-dimensioned path (shapes without an instantiation): the same program, dimensions read from `p`.
// XREG: the (x, u) row of this lane lives in registers (MixArg) and there is no LDS image (rows_lds_elems_xreg) — a
// third less scratch per row, i.e. that many more chunks resident on a CU; n, m <= 16.
// GRAD_ONLY: a merit-only evaluation (the probing passes of the line search: merit_part, nothing else): no Hessian
// entry is formed or stored, and the scratch holds the persistent slots and each pass's gradient slots only — the row
// program numbers those first (rows_lds_elems_grad).
// PROG: ProgDynamic (the interpreter) or ProgStatic<ID> (straight-line code for a registered structure; the scratch is
// laid out with the fixed strides of kStaticRowStride whatever `cw`: size it for cw = 64).
// pass_mask: bit ps = run pass ps of the program (the passes are independent of each other: the Jacobians, then one per
// player, each with its own slots and outputs).
template <typename T, int CN_, int CM_, int CNP_, bool XREG = false, bool GRAD_ONLY = false, class PROG = ProgDynamic>
__device__ __forceinline__ void rows_chunk(const DevProblem& p, const short* maps, const QuadArgs<T>& a, int k0,
                                           int nrows, int cw, T* sm, int lane, unsigned pass_mask = ~0u) {
  constexpr bool RT = CN_ == 0;
  constexpr bool ST = PROG::STATIC;
  static_assert(!XREG || (CN_ > 0 && CN_ <= 16 && CM_ <= 16), "register-held rows: compile-time n, m <= 16");
  const int CN = RT ? p.n : CN_, CM = RT ? p.m : CM_, CNP = RT ? p.N : CNP_;
  const int NA = CN + CM;
  constexpr bool REGACC = ProgInRegisters<PROG>::value;  // the slots are this lane's registers, not LDS columns
  const int cws = REGACC ? 1 : (ST ? kStaticRowStride : cw + 1);  // elements between the slots of the accumulator scratch
  const int ast = ST ? 64 : cw;                     // ... between the entries of the (x, u) image
  T* const arg = sm;
  T* const acc = sm + (XREG ? 0 : NA) * ast;
  typedef typename std::conditional<XREG, MixArg<T>, RowArg<T>>::type Arg;
  const rp_cptr rp = (rp_cptr)p.row_prog;
  const typename ConstPtr<T>::type segs = (typename ConstPtr<T>::type)problem_segs<T>(p);
  // table offsets: compile-time under ProgStatic (the run-time program has the same layout: it matched word for word)
  auto hdr = [&](auto f) -> int {
    if constexpr (ST) { constexpr int v = PROG::S::w[decltype(f)::value]; return v; } else return rp[decltype(f)::value];
  };
#define ILQG_RP_HDR(F) hdr(std::integral_constant<int, F>{})
  const int num_passes = ILQG_RP_HDR(RP_NUM_PASSES), NPS = ILQG_RP_HDR(RP_NUM_PSLOTS);
  const rp_cptr passes = rp + ILQG_RP_HDR(RP_OFF_PASS);
  const rp_cptr ops = rp + ILQG_RP_HDR(RP_OFF_OPS);
  const rp_cptr sids = rp + ILQG_RP_HDR(RP_OFF_SIDS);
  const rp_cptr pinit = rp + ILQG_RP_HDR(RP_OFF_PINIT);
  const rp_cptr linit = rp + ILQG_RP_HDR(RP_OFF_LINIT);
  const rp_cptr regions = rp + ILQG_RP_HDR(RP_OFF_REGIONS);
  const rp_cptr merit = rp + ILQG_RP_HDR(RP_OFF_MERIT);
  const bool quad_out = !GRAD_ONLY && (a.Q != nullptr || (a.compact != nullptr && a.compact_quad));
  const bool do_quad = quad_out || a.merit_part != nullptr;
  const bool want_cost = !GRAD_ONLY && a.cost_part != nullptr;
  const PairTable& pt = p.pairs;
  // phase profile (scripts/stage_bench.py): [0] staging + slot init, [1] Jacobian ops, [2] cost ops, [3] write-out
  long long qc0 = (kProfile && a.phacc) ? clock64() : 0, qc1;
#define ILQG_QPH(i) do { if (kProfile && a.phacc) { __builtin_amdgcn_sched_barrier(0); qc1 = clock64(); __builtin_amdgcn_sched_barrier(0); a.phacc[i] += qc1 - qc0; qc0 = qc1; } } while (0)

  // ---- the chunk's (x, u) rows, transposed: global reads are contiguous, a lane later reads down its column ----
  // (loads in batches of eight ahead of their LDS stores: one exposed global latency per batch instead of one per element)
  {
    auto stage_rows = [&](const T* g, int dim, int first) {
      constexpr int UN = 8;
      const int total = cw * dim;
      for (int i0 = 0; i0 < total; i0 += 64 * UN) {
        T v[UN];
        int dst[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
          const int i = i0 + u * 64 + lane;
          const bool in = i < total;
          const int ic = in ? i : 0;
          const int r = ic / dim, e = ic - r * dim;
          const int rs = r < nrows ? r : nrows - 1;  // lanes past the chunk's end repeat its last row (results unused)
          dst[u] = in ? (first + e) * ast + r : -1;
          v[u] = g[rs * dim + e];
        }
#pragma unroll
        for (int u = 0; u < UN; u++)
          if (dst[u] >= 0) arg[dst[u]] = v[u];
      }
    };
    if constexpr (!XREG) {
      stage_rows(a.xs + size_t(k0) * CN, CN, 0);
      stage_rows(a.us + size_t(k0) * CM, CM, CN);
    }
  }
  constexpr int XW = rows_vec_regs(CN_), UW = rows_vec_regs(CM_);
  typename RowVecN<T, XW>::type xrow = {};
  typename RowVecN<T, UW>::type urow = {};
  const T *xg = a.xs, *ug = a.us;
  if constexpr (XREG) {
    // this lane's state row (lanes past the chunk's end: its last row; lanes past the chunk width: lane 0's)
    const int r = lane < cw ? (lane < nrows ? lane : nrows - 1) : 0;
    xg = a.xs + size_t(k0 + r) * CN;
#pragma unroll
    for (int e = 0; e < CN_; e++) xrow[e] = xg[e];
    ug = a.us + size_t(k0 + r) * CM;
#pragma unroll
    for (int e = 0; e < CM_; e++) urow[e] = ug[e];
  }
  lds_sync(true);
  tl_stamp(a.tl, a.tl_b, 40, lane == 0);

  // Every lane runs the op loop (an op's descriptor is read one word per lane, so all 64 must be live): the lanes past
  // the chunk width (cw < 64) repeat lane 0's arguments and accumulate into the padding column nobody reads.
  const bool rowlane = lane < cw;
  const bool valid = lane < nrows;
  const int row = k0 + (valid ? lane : nrows - 1);
  const int rl = rowlane ? lane : 0;
  T racc[ProgInRegisters<PROG>::slots];  // (REGACC; every index is a compile-time constant: registers)
  T* const col = REGACC ? racc : acc + (rowlane ? lane : cw);
  // PlayerCost::Quadraticize vs QuadraticizeControlCosts (src/ilq_solver.cpp:483-487): bit i = player i is
  // quadraticised in full at this lane's row
  unsigned full = 0;
#pragma unroll
  for (int i = 0; i < CNP; i++) {
    const bool f = p.structure[i] == ILQG_SUM || (a.t_extreme ? a.t_extreme[i] == row : row == 0);
    full |= (f ? 1u : 0u) << i;
  }
  const double tt = double(row) * p.dt;
  const int tidx = int(static_cast<size_t>((tt - a.t_init) / p.dt));  // relative_time_tracker.h:69-72
  auto init_value = [&](int code, int valbits) -> T {
    const T val = T(__int_as_float(valbits));
    const int kind = code & 255, pl = (code >> 8) & 255;
    if (kind == RI_DT) return T(p.dt);
    if (kind == RI_NEG_DT) return T(-p.dt);
    if (kind == RI_CREG)  // sigma_u I on a control block the reference would have created (player_cost.cpp:70-74)
      return (((full >> pl) & 1u) || ((code >> 16) & 1)) ? val : T(0);
    return val;  // 0, 1, sigma_x (player_cost.cpp:196)
  };
  if constexpr (ST) {
    static_for<0, PROG::S::w[RP_NUM_PSLOTS]>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int code = PROG::S::w[PROG::S::w[RP_OFF_PINIT] + s * RINIT_WORDS];
      col[s * cws] = init_value(code, pinit[s * RINIT_WORDS + 1]);
    });
  } else {
    for (int s = 0; s < NPS; s++) col[s * cws] = init_value(pinit[s * RINIT_WORDS], pinit[s * RINIT_WORDS + 1]);
  }
  tl_stamp(a.tl, a.tl_b, 41, lane == 0);
  Closest<T> cc;  // result of the pass's last CLOSEST op
  cc.cx = cc.cy = cc.ssd = T(0);
  cc.is_vertex = cc.is_endpoint = false;
  cc.seg = Seg<T>{T(0), T(0), T(0), T(0), T(1), T(1), T(0)};
  ILQG_QPH(0);

  // ---- one pass: `pd` is a PassDynamic (run-time fields) or a PassStatic<S, PS> (compile-time fields) ----
  auto run_pass = [&](auto pd, int ps) {
    typedef decltype(pd) PD;
    const int op_begin = pd.op_begin, op_end = pd.op_end, reg_begin = pd.reg_begin, reg_end = pd.reg_end,
              li_begin = pd.li_begin, li_count = pd.li_count;
    const int pkind = pd.pkind, player = pd.player;
    if (!((pass_mask >> ps) & 1u)) return;  // another wave takes this pass of the chunk (the fused kernel's last chunk)
    if (GRAD_ONLY && pkind == RPASS_JACOBIANS) return;
    if (pkind == RPASS_JACOBIANS && a.A == nullptr && !(a.compact && a.compact_lin)) return;
    if (pkind == RPASS_PLAYER && !do_quad && !want_cost) return;
    T ctot = T(0);  // PlayerCost::Evaluate of this pass's player at this lane's row
    T ext_value = T(0);  // ExtremeValueCost in flight: its value and active child at this lane's row
    int ext_best = 0;
    {
      const int li_live = GRAD_ONLY ? pd.li_grad : li_count;  // GRAD_ONLY: the pass's gradient slots, numbered first
      if constexpr (ST) {
        static_for<0, (GRAD_ONLY ? PD::li_grad : PD::li_count)>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          constexpr int code = PROG::S::w[PROG::S::w[RP_OFF_LINIT] + (PD::li_begin + s) * RINIT_WORDS];
          col[(PROG::S::w[RP_NUM_PSLOTS] + s) * cws] = init_value(code, linit[(PD::li_begin + s) * RINIT_WORDS + 1]);
        });
      } else {
        for (int s = 0; s < li_live; s++)
          col[(NPS + s) * cws] = init_value(linit[(li_begin + s) * RINIT_WORDS], linit[(li_begin + s) * RINIT_WORDS + 1]);
      }
      // ---- one op: `od` is an OpDynamic or an OpStatic<S, OP> ----
      auto run_op = [&](auto od) {
        const int mode = od.template field<RO_MODE>();
        const int nsid = od.template field<RO_NSID>(), aux = od.template field<RO_AUX>();
        const bool sid_inline = nsid <= ROP_INLINE_SIDS;
        const auto sid = od.sids();
        DevTerm c;
        c.kind = od.template field<RO_KIND>(); c.role = od.template field<RO_ROLE>(); c.player = od.template field<RO_PLAYER>();
        c.flags = od.template field<RO_FLAGS>();
        c.idx[0] = od.template field<RO_IDX0>(); c.idx[1] = od.template field<RO_IDX1>();
        c.idx[2] = od.template field<RO_IDX2>(); c.idx[3] = od.template field<RO_IDX3>();
        c.weight = __int_as_float(od.template param<RO_WEIGHT>()); c.value = __int_as_float(od.template param<RO_VALUE>());
        c.slot = od.template field<RO_SLOT>(); c.arg_off = od.template field<RO_ARG_OFF>();
        c.arg_dim = od.template field<RO_ARG_DIM>();
        c.k_start = od.template field<RO_K_START>();
        c.polyline = od.template param<RO_POLY_FIRST>();  // segment range / table offset / WeightedConvexProximity's packed indices
        c.arg = 0; c.child_begin = 0; c.child_count = 0;
        if (mode == ROP_JACOBIAN) {
          // ---- ConcatenatedDynamicalSystem::Linearize, one subsystem (src/concatenated_dynamical_system.cpp:86-107):
          // the entries each model's Linearize adds to (I, 0) — single_player_unicycle_4d.h:102-116,
          // single_player_car_5d.h:113-133, single_player_car_6d.h:116-138 (mixed float * double products kept),
          // single_player_dubins_car.h:105-117, air_3d.h:127-148 — go to the slots the program lists in this order;
          // the constant entries (dt, -dt, the identity) are in the word maps.
          const int kind = c.kind, xo = c.idx[0], uo = c.idx[1];
          const T L = T(c.weight);
          const Arg x = rows_make_arg<T, XREG, XW, UW>(xrow, urow, xg, ug, arg, ast, CN, xo, rl, 2, kind == ILQG_DYN_AIR_3D_EVADER ? 0 : 3,
                                                   kind == ILQG_DYN_AIR_3D_EVADER ? 1 : 4, -1);
          auto put = [&](int e, T val) { if (e < nsid) col[sid[e] * cws] = val; };
          if (kind == ILQG_DYN_POINT_MASS_2D || kind == ILQG_DYN_PLANAR_DISTURBANCE || kind == ILQG_DYN_AIR_3D_PURSUER)
            return;  // constants only
          T sth, cth;
          t_sincos(x[2], &sth, &cth);
          const T ct = T(double(cth) * p.dt), st = T(double(sth) * p.dt);
          if (kind == ILQG_DYN_AIR_3D_EVADER) {  // air_3d.h:127-146
            const T own = XREG ? T(urow[uo & (UW - 1)]) : arg[(CN + uo) * ast + rl];  // its own turn rate; c.value = the pursuer's speed
            put(0, T(double(own) * p.dt));             // A(0,1)
            put(1, T(0) - T(c.value) * st);            // A(0,2)
            put(2, T(0) - T(double(own) * p.dt));      // A(1,0)
            put(3, T(c.value) * ct);                   // A(1,2)
            put(4, T(double(x[1]) * p.dt));            // B(0,0)
            put(5, T(double(-x[0]) * p.dt));           // B(1,0)
          } else if (kind == ILQG_DYN_DUBINS_CAR || kind == ILQG_DYN_DELAYED_DUBINS_CAR) {
            // single_player_dubins_car.h:105-117, single_player_delayed_dubins_car.h:115-127
            put(0, T(0) + -L * st);  // A(0,2)
            put(1, T(0) + L * ct);   // A(1,2)
          } else {
            const bool uni = is_unicycle(kind) || kind == ILQG_DYN_UNICYCLE_5D;  // single_player_unicycle_5d.h:105-122
            const T v = x[uni ? 3 : 4];
            put(0, T(0) + -v * st);  // A(0,2)
            put(1, ct);              // A(0,v)
            put(2, v * ct);          // A(1,2)
            put(3, st);              // A(1,v)
            if (!uni) {
              T sphi, cphi;
              t_sincos(x[3], &sphi, &cphi);
              const T tphi = sphi / cphi;
              put(4, T(double(x[4]) * p.dt / double(L * cphi * cphi)));  // A(2,3)
              put(5, T(double(tphi) * p.dt / double(L)));               // A(2,4)
              if (kind == ILQG_DYN_CAR_7D) {  // single_player_car_7d.h:141-151: the curvature row, all-double products
                const T own = XREG ? T(urow[uo & (UW - 1)]) : arg[(CN + uo) * ast + rl];  // omega
                const T den = cphi * cphi * L;
                put(6, T(2.0 * p.dt * double(own) * double(tphi) / double(den)));  // A(5,3)
                put(7, T(p.dt / double(den)));                                     // B(5,0)
              }
            }
          }
          return;
        }
        const Arg v = rows_make_arg<T, XREG, XW, UW>(xrow, urow, xg, ug, arg, ast, CN, c.arg_off, rl, c.idx[0], c.idx[1], c.idx[2], c.idx[3]);
        if (mode == ROP_AFFINE) {
          // constraints are quadraticised with the player's full PlayerCost::Quadraticize only (:483-487), from their
          // first active step on (FinalTimeConstraint)
          const bool deriv = do_quad && row >= c.k_start && ((full >> c.player) & 1u);
          if (!__any(deriv)) return;
          const T lambda = (c.slot >= 0 && a.lambdas) ? a.lambdas[c.slot * p.T + tidx] : T(0);
          if (deriv)
            rows_affine<T>(c.kind, (c.flags & ILQG_FLAG_EQUALITY) != 0, problem_dense<T>(p) + od.template param<RO_POLY_FIRST>(), c.arg_dim,
                           SidsTable{sids + od.template field<RO_SID>()}, col, cws, v, lambda, a.mu, quad_out);
          return;
        }
        if (mode == ROP_CLOSEST) {
          cc = polyline_closest_rows<T>(segs, od.template param<RO_POLY_FIRST>(), od.template param<RO_PATTERN_NSEG>(), v[c.idx[0]], v[c.idx[1]]);
          return;
        }
#if ILQG_PROFILE2
        ILQG_QPH(1);
#endif
        const bool is_cost = c.role == ILQG_ROLE_STATE_COST || c.role == ILQG_ROLE_CONTROL_COST;
        const bool live = row >= c.k_start;  // FinalTimeCost: nothing before its threshold
        const bool deriv = do_quad && live && (((full >> c.player) & 1u) || c.role == ILQG_ROLE_CONTROL_COST);
        const bool need = deriv || (want_cost && is_cost && live);
        if (!__any(need)) return;
        const T lambda = (c.slot >= 0 && a.lambdas) ? a.lambdas[c.slot * p.T + tidx] : T(0);
#if ILQG_PROFILE2
        ILQG_QPH(2);
#endif
        TermOut<T> o;
        double tnom[2] = {0.0, 0.0};
        if (term_is_time_dependent(c.kind)) {  // this row's nominal; the op's RO_POLY_FIRST field is the term's table
          const double* tn = problem_time_nominal<T>(p) + (size_t(od.template param<RO_POLY_FIRST>()) * p.T + row) * 2;
          tnom[0] = tn[0];
          tnom[1] = tn[1];
        }
        term_compute_leaf<T, Arg, true>(QuadTables<T>{}, c, v, lambda, a.mu, &o, &cc, tnom);
#if ILQG_PROFILE2
        ILQG_QPH(3);
#endif
        bool act = deriv && o.pattern != PAT_NONE;
        if (mode == ROP_EXT_EVAL) {
          // ExtremeValueCost::ExtremeCost (src/extreme_value_cost.cpp:66-85): the first strict improvement wins
          const bool is_min = c.flags & ILQG_FLAG_IS_MIN;
          if (aux == 0 || (is_min && o.value < ext_value) || (!is_min && o.value > ext_value)) {
            ext_value = o.value;
            ext_best = aux;
          }
          return;
        }
        if (mode == ROP_EXT_APPLY) {
          if (aux == 0 && is_cost && live) ctot += ext_value;
          act = act && ext_best == aux;
        } else if (is_cost && live) {
          ctot += o.value;
        }
        if (act) {
          if (sid_inline)
            rows_scatter<T>(od.template field<RO_PATTERN_NSEG>(), o, sid, nsid, col, cws, v, quad_out);
          else
            rows_scatter<T>(od.template field<RO_PATTERN_NSEG>(), o, SidsTable{sids + od.template field<RO_SID>()}, nsid, col, cws, v, quad_out);
        }
#if ILQG_PROFILE2
        ILQG_QPH(4);
#endif
      };
      if constexpr (ST) {
        static_for<PD::op_begin, PD::op_end>([&](auto opc) {
          constexpr int OP = decltype(opc)::value;
          run_op(OpStatic<typename PROG::S, OP>{ops + OP * ROP_WORDS});
        });
      } else {
        // An op's descriptor (ROP_WORDS words) is fetched as ONE vector load, word w by lane w, one op ahead, and its
        // fields are read out through v_readlane: scalar loads of the descriptor count on lgkmcnt together with the LDS
        // traffic of the op before and return out of order, so every decode drained the accumulators' read-modify-writes
        // and then waited a scalar-cache round trip.
        static_assert(ROP_WORDS <= 64, "an op descriptor is one word per lane");
        const int* const opsv = reinterpret_cast<const int*>(p.row_prog) + rp[RP_OFF_OPS];
        int next_words = (lane < ROP_WORDS && op_begin < op_end) ? opsv[op_begin * ROP_WORDS + lane] : 0;
#pragma unroll 1
        for (int op = op_begin; op < op_end; op++) {
          const int words = next_words;
          if (op + 1 < op_end) next_words = lane < ROP_WORDS ? opsv[(op + 1) * ROP_WORDS + lane] : 0;
          run_op(OpDynamic{words});
        }
      }
      if (pkind == RPASS_PLAYER && valid) {
        if (want_cost) a.cost_part[size_t(row) * CNP + player] = ctot;
        if (a.merit_part) {
          // pieces of ILQSolver::MeritFunction (:419-430): |r_ii|^2 and |l_i|^2 of this lane's row.  Entries no term
          // touches are exact zeros, so the sums over the touched slots (in index order) are the reference's sums.
          T s1 = T(0), s2 = T(0);
          if constexpr (ST) {
            typedef typename PROG::S S;
            constexpr int mb = S::w[RP_OFF_MERIT] + PD::player * RMERIT_WORDS, sb = S::w[RP_OFF_SIDS];
            static_for<0, S::w[mb + 3]>([&](auto dc) {
              const T rv = col[S::w[sb + S::w[mb + 2] + decltype(dc)::value] * cws];
              s1 += rv * rv;
            });
            static_for<0, S::w[mb + 1]>([&](auto dc) {
              const T lv = col[S::w[sb + S::w[mb + 0] + decltype(dc)::value] * cws];
              s2 += lv * lv;
            });
          } else {
            const rp_cptr md = merit + player * RMERIT_WORDS;
            const rp_cptr ls = sids + md[0];
            const rp_cptr rs = sids + md[2];
            const int lcnt = md[1], rcnt = md[3];
            for (int d = 0; d < rcnt; d++) {
              const T rv = col[rs[d] * cws];
              s1 += rv * rv;
            }
            for (int d = 0; d < lcnt; d++) {
              const T lv = col[ls[d] * cws];
              s2 += lv * lv;
            }
          }
          a.merit_part[(size_t(row) * CNP + player) * 2 + 0] = s1;
          a.merit_part[(size_t(row) * CNP + player) * 2 + 1] = s2;
        }
      }
    }
    lds_sync(true);  // the slots of this pass are complete: rows are written out by all lanes
    tl_stamp(a.tl, a.tl_b, 42 + 2 * (ps < 4 ? ps : 4), lane == 0);
#if !ILQG_PROFILE2
    if (pkind == RPASS_JACOBIANS) ILQG_QPH(1); else ILQG_QPH(2);
#else
    ILQG_QPH(0);
#endif
    if constexpr (!REGACC)
    for (int rg = reg_begin; rg < (GRAD_ONLY ? reg_begin : reg_end); rg++) {
      const rp_cptr rd = regions + rg * RREG_WORDS;
      const int arr = rd[0], words = rd[1], offs = rd[2];
      const short* const map = maps + rd[3];
      if (arr == RA_A) {
        if (a.A) {
          if constexpr (RT) rows_writeout_rt<T>(a.A + size_t(k0) * CN * CN, size_t(CN) * CN, CN * CN, map, acc, cws, nrows, lane);
          else rows_writeout<T, CN_ * CN_, rows_vec_width<T>(CN_ * CN_)>(a.A + size_t(k0) * CN * CN, CN * CN, map, acc, cws, nrows, lane);
        }
      } else if (arr == RA_B) {
        if (a.Bm) {
          if constexpr (RT) rows_writeout_rt<T>(a.Bm + size_t(k0) * CN * CM, size_t(CN) * CM, CN * CM, map, acc, cws, nrows, lane);
          else rows_writeout<T, CN_ * CM_, rows_vec_width<T>(CN_ * CM_)>(a.Bm + size_t(k0) * CN * CM, CN * CM, map, acc, cws, nrows, lane);
        }
      } else if (arr == RA_Q) {
        if (a.Q) {
          if constexpr (RT) rows_writeout_rt<T>(a.Q + size_t(k0) * CNP * CN * CN + offs, size_t(CNP) * CN * CN, CN * CN, map, acc, cws, nrows, lane);
          else rows_writeout<T, CN_ * CN_, rows_vec_width<T>(CN_ * CN_)>(a.Q + size_t(k0) * CNP * CN * CN + offs, size_t(CNP) * CN * CN, map, acc, cws, nrows, lane);
        }
      } else if (arr == RA_L) {
        if (a.l) rows_writeout_small<T>(a.l + size_t(k0) * CNP * CN + offs, CNP * CN, words, map, acc, cws, nrows, lane);
      } else if (arr == RA_R) {
        if (a.R) rows_writeout_small<T>(a.R + size_t(k0) * pt.Rsz + offs, pt.Rsz, words, map, acc, cws, nrows, lane);
      } else {
        if (a.r) rows_writeout_small<T>(a.r + size_t(k0) * pt.rsz + offs, pt.rsz, words, map, acc, cws, nrows, lane);
      }
    }
    if (!GRAD_ONLY && a.compact && (pkind == RPASS_PLAYER ? a.compact_quad : a.compact_lin)) {
      // compact row: this pass's local slots, in slot order, at the pass's base (lane = slot, loop = row)
      int CWD, base;
      if constexpr (ST) {
        typedef typename PROG::S S;
        constexpr int cbo = S::w[RP_OFF_COMPACT];
        CWD = S::w[cbo + RC_W];
        base = S::w[cbo + RC_BASE + (PD::pkind == RPASS_PLAYER ? 1 + PD::player : 0)];
      } else {
        const rp_cptr cb = rp + rp[RP_OFF_COMPACT];
        CWD = cb[RC_W];
        base = cb[RC_BASE + (pkind == RPASS_PLAYER ? 1 + player : 0)];
      }
      T* const g0 = a.compact + size_t(k0) * CWD + base;
      constexpr int RB = 8;
      if constexpr (REGACC) {
        // this lane's row, from its registers: the pass's slots are contiguous in the compact row
        typedef typename PROG::S S;
        constexpr int cbo = S::w[RP_OFF_COMPACT], P0 = S::w[RP_NUM_PSLOTS], NL = PD::li_count;
        constexpr int base_c = S::w[cbo + RC_BASE + (PD::pkind == RPASS_PLAYER ? 1 + PD::player : 0)], cwd_c = S::w[cbo + RC_W];
        constexpr int full = 16 / int(sizeof(T));
        constexpr int VW = (base_c % full == 0 && cwd_c % full == 0) ? full : ((base_c % 2 == 0 && cwd_c % 2 == 0 && full >= 2) ? 2 : 1);
        typedef typename RowVec<T, VW>::type vec;
        if (valid) {
          T* const g = g0 + size_t(lane) * cwd_c;
          static_for<0, NL / VW>([&](auto sc) {
            constexpr int s0 = decltype(sc)::value * VW;
            if constexpr (VW == 1) {
              g[s0] = racc[P0 + s0];
            } else {
              vec v;
#pragma unroll
              for (int q = 0; q < VW; q++) v[q] = racc[P0 + s0 + q];
              *reinterpret_cast<vec*>(g + s0) = v;
            }
          });
          static_for<(NL / VW) * VW, NL>([&](auto sc) { g[decltype(sc)::value] = racc[P0 + decltype(sc)::value]; });
        }
      } else
      for (int s0 = 0; s0 < li_count; s0 += 64) {
        const bool in = s0 + lane < li_count;
        const int off = (NPS + (in ? s0 + lane : 0)) * cws;
#pragma unroll 1
        for (int r0 = 0; r0 < nrows; r0 += RB) {
          T val[RB];
#pragma unroll
          for (int rr = 0; rr < RB; rr++) val[rr] = acc[off + (r0 + rr < nrows ? r0 + rr : nrows - 1)];
#pragma unroll
          for (int rr = 0; rr < RB; rr++)
            if (r0 + rr < nrows && in) g0[size_t(r0 + rr) * CWD + s0 + lane] = val[rr];
        }
      }
    }
    lds_sync(true);  // pass-local slots are re-initialised by the next pass
    tl_stamp(a.tl, a.tl_b, 43 + 2 * (ps < 4 ? ps : 4), lane == 0);
#if !ILQG_PROFILE2
    ILQG_QPH(3);
#else
    ILQG_QPH(5);
#endif
  };
  if constexpr (ST) {
    static_for<0, PROG::S::w[RP_NUM_PASSES]>([&](auto psc) {
      run_pass(PassStatic<typename PROG::S, decltype(psc)::value>{}, decltype(psc)::value);
    });
  } else {
#pragma unroll 1
    for (int ps = 0; ps < num_passes; ps++) {
      const rp_cptr pr = passes + ps * RPASS_WORDS;
      run_pass(PassDynamic{pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7], pr[8]}, ps);
    }
  }
#undef ILQG_QPH
#undef ILQG_RP_HDR
}

// The word maps of the row program, copied into LDS once per workgroup (every thread calls, then syncs): the
// write-out reads them with a per-lane index.
__host__ __device__ inline size_t rows_maps_bytes(const DevProblem& p) { return (size_t(p.rp_maps_words) * 4 + 15) & ~size_t(15); }
__device__ __forceinline__ const short* rows_maps_load(const DevProblem& p, void* region) {
  int* dst = reinterpret_cast<int*>(region);
  const int* src = p.row_prog + p.rp_maps_off;
  for (int e = threadIdx.x; e < p.rp_maps_words; e += blockDim.x) dst[e] = src[e];
  __syncthreads();
  return reinterpret_cast<const short*>(dst);
}

#pragma clang fp contract(fast)  // back to the translation unit's default

}  // namespace ilqg
