// ilqg_rowprog.hpp — host side of the lane-per-time-step quadraticisation stage: compiles a problem's dynamics and
// cost list into the "row program" ilqg_rows.hpp executes (passes, ops, slot lists, word -> slot maps).
//
// A slot is one word of the per-row image [A | B | Q_i | l_i | R_ij | r_ij] that some term (or a Jacobian) can
// write; every other word of the image is a constant (0, 1, +-dt, a regularisation weight).  The constants are the
// persistent slots; everything a pass accumulates (the Jacobian entries; the touched entries of one player's Q_i and
// l_i, the diagonals and touched entries of its R_ij, its touched r_ij entries) is pass-local and its LDS is re-used
// by the next pass.  Accumulation order is PlayerCost::Quadraticize's (src/player_cost.cpp:194-225): per player
// state costs, control costs, state constraints, control constraints, each in table order.
#pragma once

#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "ilqg_rows.hpp"

namespace ilqg {

struct RowProgramHost {
  std::vector<int> words;  // the device image
  int num_pslots = 0, max_lslots = 0, max_gslots = 0, maps_off = 0, maps_words = 0;
  int compact_off = 0, compact_w = 0;  // the compact-row block (ilqg_rows.hpp: RP_OFF_COMPACT) and its row length
};

// `poly_off`: the problem's polyline offsets (points), host copy.
inline bool build_row_program(const DevProblem& d, const std::vector<DevTerm>& dt, const int* poly_off,
                              RowProgramHost* out, std::string* err) {
  const int n = d.n, m = d.m, N = d.N;
  const PairTable& pt = d.pairs;
  auto fbits = [](float v) { int b; std::memcpy(&b, &v, sizeof(b)); return b; };
  std::vector<int> pinit;  // init records of the persistent slots
  auto new_pslot = [&](int code, float value) {
    pinit.push_back(code);
    pinit.push_back(fbits(value));
    return int(pinit.size()) / RINIT_WORDS - 1;
  };
  const int S_ZERO = new_pslot(RI_VALUE, 0.0f), S_ONE = new_pslot(RI_VALUE, 1.0f), S_DT = new_pslot(RI_DT, 0.0f),
            S_NDT = new_pslot(RI_NEG_DT, 0.0f);
  if (S_ZERO != kRowSlotZero) { *err = "row program: slot 0 must be the zero constant"; return false; }
  std::vector<int> S_SREG(N);
  for (int i = 0; i < N; i++) S_SREG[i] = new_pslot(RI_VALUE, d.state_reg[i]);
  const int NPS = int(pinit.size()) / RINIT_WORDS;

  auto pair_of = [&](int i, int j) {
    for (int q = 0; q < pt.npairs; q++)
      if (pt.pi[q] == i && pt.pj[q] == j) return q;
    return -1;
  };
  // ---- the entries a leaf term adds to: (is_hessian, a, b) in rows_scatter's order ----
  struct Entry { bool h; int a, b; };
  auto leaf_entries = [&](const DevTerm& c, int dim, std::vector<Entry>* e) {
    if (term_is_affine(c.kind)) {  // dense: every gradient entry, every Hessian entry column by column (rows_affine)
      for (int i = 0; i < dim; i++) e->push_back({false, i, 0});
      for (int j = 0; j < dim; j++)
        for (int i = 0; i < dim; i++) e->push_back({true, i, j});
      return;
    }
    const int pat = term_pattern_of(c.kind, c.idx[0]);
    if (pat == PAT_SINGLE) {
      e->push_back({false, c.idx[0], 0});
      e->push_back({true, c.idx[0], c.idx[0]});
    } else if (pat == PAT_PAIR2) {
      const int x = c.idx[0], y = c.idx[1];
      e->push_back({false, x, 0}); e->push_back({false, y, 0});
      e->push_back({true, x, x}); e->push_back({true, y, y}); e->push_back({true, x, y}); e->push_back({true, y, x});
    } else if (pat == PAT_PAIR4) {
      const int x1 = c.idx[0], y1 = c.idx[1], x2 = c.idx[2], y2 = c.idx[3];
      e->push_back({false, x1, 0}); e->push_back({false, x2, 0}); e->push_back({false, y1, 0}); e->push_back({false, y2, 0});
      const int hh[16][2] = {{x1, x1}, {x1, x2}, {x2, x1}, {x2, x2}, {y1, y1}, {y1, y2}, {y2, y1}, {y2, y2},
                             {x1, y1}, {y1, x1}, {x1, y2}, {y2, x1}, {x2, y1}, {y1, x2}, {x2, y2}, {y2, x2}};
      for (auto& q : hh) e->push_back({true, q[0], q[1]});
    } else if (pat == PAT_ALL) {
      for (int i = 0; i < dim; i++) e->push_back({false, i, 0});
      for (int i = 0; i < dim; i++) e->push_back({true, i, i});
    } else if (pat == PAT_CROSS4) {
      const int p1 = c.idx[0], p2 = c.idx[1], v1 = c.idx[2], v2 = c.idx[3];
      const int hh[8][2] = {{p1, v1}, {p1, v2}, {p2, v1}, {p2, v2}, {v1, p1}, {v1, p2}, {v2, p1}, {v2, p2}};
      for (auto& q : hh) e->push_back({true, q[0], q[1]});
    }
  };
  auto on_state = [&](const DevTerm& c) {
    return c.role == ILQG_ROLE_STATE_COST || c.role == ILQG_ROLE_STATE_CONSTRAINT || c.role == ILQG_ROLE_CHILD;
  };

  // ---- tables ----
  std::vector<int> passes, ops, sids, linit, regions, merit;
  std::vector<short> maps;
  int max_lslots = 0, max_gslots = 0;
  auto emit_op = [&](int mode, int sid_begin, int nsid, int aux, const DevTerm& c, const DevTerm& owner, int poly_first,
                     int pattern_or_nseg) {
    // `owner`: the top-level term whose role / player / constraint slot / first step apply (c itself, or the
    // ExtremeValueCost c is a child of)
    int o[ROP_WORDS] = {0};
    o[RO_MODE] = mode; o[RO_SID] = sid_begin; o[RO_NSID] = nsid; o[RO_AUX] = aux;
    o[RO_KIND] = c.kind; o[RO_ROLE] = owner.role; o[RO_PLAYER] = owner.player;
    o[RO_FLAGS] = c.flags | (owner.flags & ILQG_FLAG_IS_MIN);
    for (int q = 0; q < 4; q++) o[RO_IDX0 + q] = c.idx[q];
    o[RO_WEIGHT] = fbits(c.weight); o[RO_VALUE] = fbits(c.value); o[RO_POLY_FIRST] = poly_first;
    o[RO_SLOT] = owner.slot; o[RO_ARG_OFF] = owner.arg_off; o[RO_ARG_DIM] = owner.arg_dim; o[RO_K_START] = owner.k_start;
    o[RO_PATTERN_NSEG] = pattern_or_nseg;
    for (int e = 0; e < ROP_INLINE_SIDS; e++)  // the op's first slot ids ride in its record (they are already listed)
      o[ROP_FIELDS + e] = (e < nsid && size_t(sid_begin + e) < sids.size()) ? sids[sid_begin + e] : 0;
    ops.insert(ops.end(), o, o + ROP_WORDS);
  };
  auto add_region = [&](int arr, int words, int offs, const std::vector<short>& map) {
    regions.push_back(arr); regions.push_back(words); regions.push_back(offs); regions.push_back(int(maps.size()));
    maps.insert(maps.end(), map.begin(), map.end());
    if (maps.size() & 1) maps.push_back(0);
  };
  auto add_linit = [&](int code, float value) { linit.push_back(code); linit.push_back(fbits(value)); };

  // Compact rows (ilqg_common.hpp): a pass's local slots in slot order, where each lands in the dense arrays, and the
  // constants around them.
  std::vector<int> cbase, cdst, cbg;
  bool compact_ok = true;
  auto collect_compact = [&](int reg_begin, int nl) {
    const int base = int(cdst.size());
    cbase.push_back(base);
    cdst.resize(size_t(base) + nl, -1);
    for (int rg = reg_begin; rg < int(regions.size()) / RREG_WORDS; rg++) {
      const int arr = regions[rg * RREG_WORDS + 0], words = regions[rg * RREG_WORDS + 1], offs = regions[rg * RREG_WORDS + 2];
      const int mo = regions[rg * RREG_WORDS + 3];
      for (int wd = 0; wd < words; wd++) {
        const int slot = maps[size_t(mo) + wd];
        if (slot >= NPS) {
          if (cdst[size_t(base) + slot - NPS] >= 0) compact_ok = false;  // a slot feeds one word
          cdst[size_t(base) + slot - NPS] = (arr << 24) | (offs + wd);
        } else if (slot != S_ZERO) {
          const int kind = pinit[size_t(slot) * RINIT_WORDS] & 255;
          float v = 0.0f;
          std::memcpy(&v, &pinit[size_t(slot) * RINIT_WORDS + 1], sizeof(v));
          const int ck = kind == RI_VALUE ? RC_LITERAL : (kind == RI_DT ? RC_DT : (kind == RI_NEG_DT ? RC_NEG_DT : -1));
          if (ck < 0) compact_ok = false;
          if (ck != RC_LITERAL || v != 0.0f) {
            cbg.push_back((arr << 24) | (offs + wd));
            cbg.push_back(ck);
            cbg.push_back(fbits(v));
          }
        }
      }
    }
    for (int e = base; e < int(cdst.size()); e++)
      if (cdst[e] < 0) compact_ok = false;
  };

  // pass 0: the Jacobians
  {
    std::vector<short> mapA(size_t(n) * n, short(S_ZERO)), mapB(size_t(n) * m, short(S_ZERO));
    for (int i = 0; i < n; i++) mapA[i + n * i] = short(S_ONE);
    const int op_begin = int(ops.size()) / ROP_WORDS, li_begin = int(linit.size()) / RINIT_WORDS;
    int nl = 0;
    for (int s = 0; s < N; s++) {
      const int kind = d.sub_kind[s], xo = d.xoff[s], uo = d.uoff[s];
      std::vector<std::pair<int, int>> av, bv;  // computed entries, in the order rows_chunk's Jacobian op writes them
      auto A = [&](int r, int c) -> short& { return mapA[(xo + r) + n * (xo + c)]; };
      auto B = [&](int r, int c) -> short& { return mapB[(xo + r) + n * (uo + c)]; };
      if (kind == ILQG_DYN_UNICYCLE_4D || kind == ILQG_DYN_UNICYCLE_4D_DISTURBED || kind == ILQG_DYN_UNICYCLE_5D) {
        av = {{0, 2}, {0, 3}, {1, 2}, {1, 3}};
        B(2, 0) = short(S_DT); B(3, 1) = short(S_DT);
        if (kind == ILQG_DYN_UNICYCLE_5D) A(4, 3) = short(S_DT);
      } else if (kind == ILQG_DYN_CAR_5D || kind == ILQG_DYN_CAR_6D || kind == ILQG_DYN_CAR_7D) {
        av = {{0, 2}, {0, 4}, {1, 2}, {1, 4}, {2, 3}, {2, 4}};
        if (kind == ILQG_DYN_CAR_5D) { B(3, 0) = short(S_DT); B(4, 1) = short(S_DT); }
        else if (kind == ILQG_DYN_CAR_6D) { A(4, 5) = short(S_DT); B(3, 0) = short(S_DT); B(5, 1) = short(S_DT); }
        else {  // the curvature row: A(kappa, phi) and B(kappa, omega) are computed
          av.push_back({5, 3});
          bv = {{5, 0}};
          A(6, 4) = short(S_DT); B(3, 0) = short(S_DT); B(4, 1) = short(S_DT);
        }
      } else if (kind == ILQG_DYN_DUBINS_CAR || kind == ILQG_DYN_DELAYED_DUBINS_CAR) {
        av = {{0, 2}, {1, 2}};
        if (kind == ILQG_DYN_DUBINS_CAR) B(2, 0) = short(S_DT);
        else { A(2, 3) = short(S_DT); B(3, 0) = short(S_DT); }
      } else if (kind == ILQG_DYN_POINT_MASS_2D) {
        A(0, 2) = short(S_DT); A(1, 3) = short(S_DT); B(2, 0) = short(S_DT); B(3, 1) = short(S_DT);
      } else if (kind == ILQG_DYN_AIR_3D_EVADER) {
        av = {{0, 1}, {0, 2}, {1, 0}, {1, 2}};
        bv = {{0, 0}, {1, 0}};
        B(2, 0) = short(S_NDT);
      } else if (kind == ILQG_DYN_AIR_3D_PURSUER) {
        B(-1, 0) = short(S_DT);
      } else if (kind == ILQG_DYN_PLANAR_DISTURBANCE) {
        B(-4, 0) = short(S_DT); B(-3, 1) = short(S_DT);
      } else {
        *err = "row program: unknown subsystem kind";
        return false;
      }
      DevTerm j{};
      j.kind = kind;
      j.idx[0] = xo; j.idx[1] = uo;
      j.weight = d.sub_param[s];
      j.value = s + 1 < N ? d.sub_param[s + 1] : 0.0f;  // Air3D: the pursuer's speed enters the evader's rows
      const int sid_begin = int(sids.size());
      for (auto& rc : av) {
        A(rc.first, rc.second) = short(NPS + nl);
        sids.push_back(NPS + nl);
        add_linit(RI_VALUE, 0.0f);
        nl++;
      }
      for (auto& rc : bv) {
        B(rc.first, rc.second) = short(NPS + nl);
        sids.push_back(NPS + nl);
        add_linit(RI_VALUE, 0.0f);
        nl++;
      }
      emit_op(ROP_JACOBIAN, sid_begin, int(av.size() + bv.size()), s, j, j, 0, 0);
    }
    const int reg_begin = int(regions.size()) / RREG_WORDS;
    add_region(RA_A, n * n, 0, mapA);
    add_region(RA_B, n * m, 0, mapB);
    passes.insert(passes.end(), {op_begin, int(ops.size()) / ROP_WORDS, reg_begin, int(regions.size()) / RREG_WORDS,
                                 li_begin, nl, RPASS_JACOBIANS, 0, 0});
    if (nl > max_lslots) max_lslots = nl;
    collect_compact(reg_begin, nl);
  }

  // passes 1..N: the players
  merit.assign(size_t(N) * RMERIT_WORDS, 0);
  for (int i = 0; i < N; i++) {
    std::map<int, int> q_slot;                  // word of Q_i (a + n * b) -> pass-local slot
    std::map<int, int> l_slot;                  // d
    std::map<std::pair<int, int>, int> R_slot;  // (pair, a + mj * b)
    std::map<std::pair<int, int>, int> r_slot;  // (pair, d)
    const int op_begin = int(ops.size()) / ROP_WORDS, li_begin = int(linit.size()) / RINIT_WORDS;
    const size_t sids_begin = sids.size();
    std::vector<int> scratch_slots;  // an affine vector constraint's temporaries
    int nl = 0;
    // sigma_u on the diagonal of every control block of this player (player_cost.cpp:70-74)
    for (int q = 0; q < pt.npairs; q++) {
      if (pt.pi[q] != i) continue;
      const int mj = d.udim[pt.pj[q]];
      for (int a = 0; a < mj; a++) {
        R_slot[{q, a + mj * a}] = nl++;
        add_linit(RI_CREG | (i << 8) | ((pt.from_cost[q] ? 1 : 0) << 16), d.control_reg[i]);
      }
    }
    bool ok = true;
    // slot ids of one leaf's entries, in rows_scatter's order
    auto leaf_sids = [&](const DevTerm& leaf, const DevTerm& owner) {
      const int begin = int(sids.size());
      std::vector<Entry> le;
      leaf_entries(leaf, owner.arg_dim, &le);
      const bool st = on_state(owner);
      const int q = st ? -1 : pair_of(owner.player, owner.arg);
      const int mj = st ? 0 : d.udim[owner.arg];
      if (!st && q < 0) { ok = false; return begin; }
      for (auto& e : le) {
        const int lim = st ? n : mj;
        if (e.a < 0 || e.a >= lim || (e.h && (e.b < 0 || e.b >= lim))) { ok = false; return begin; }
        int* slot;
        float init = 0.0f;
        if (st && e.h) {
          const int w = e.a + n * e.b;
          if (!q_slot.count(w)) q_slot[w] = -1;
          slot = &q_slot[w];
          init = e.a == e.b ? d.state_reg[i] : 0.0f;
        } else if (st) {
          if (!l_slot.count(e.a)) l_slot[e.a] = -1;
          slot = &l_slot[e.a];
        } else if (e.h) {
          if (!R_slot.count({q, e.a + mj * e.b})) R_slot[{q, e.a + mj * e.b}] = -1;
          slot = &R_slot[{q, e.a + mj * e.b}];
        } else {
          if (!r_slot.count({q, e.a})) r_slot[{q, e.a}] = -1;
          slot = &r_slot[{q, e.a}];
        }
        if (*slot < 0) {
          *slot = nl++;
          add_linit(RI_VALUE, init);
        }
        sids.push_back(NPS + *slot);
      }
      // rows_scatter reads all of a leaf's slots before it writes any of them
      for (size_t x = begin; x < sids.size(); x++)
        for (size_t y = x + 1; y < sids.size(); y++)
          if (sids[x] == sids[y]) { ok = false; return begin; }
      return begin;
    };
    int closest_key[4] = {-1, -1, -1, -1};  // (polyline, x index, y index, argument offset) of the pass's last CLOSEST op
    auto want_closest = [&](const DevTerm& leaf, const DevTerm& owner) {
      if (!term_is_polyline(leaf.kind)) return;
      const int key[4] = {leaf.polyline, leaf.idx[0], leaf.idx[1], owner.arg_off};
      if (std::memcmp(key, closest_key, sizeof(key)) == 0) return;
      std::memcpy(closest_key, key, sizeof(key));
      if (leaf.polyline < 0 || leaf.polyline >= d.num_polylines) { ok = false; return; }
      const int first = poly_off[leaf.polyline] - leaf.polyline;  // segments before this polyline
      const int nseg = poly_off[leaf.polyline + 1] - poly_off[leaf.polyline] - 1;
      if (nseg < 1) { ok = false; return; }
      emit_op(ROP_CLOSEST, 0, 0, 0, leaf, owner, first, nseg);
    };
    for (int role = 0; role < 4 && ok; role++)
      for (int ti = 0; ti < d.num_terms && ok; ti++) {
        const DevTerm& c = dt[ti];
        if (c.player != i || c.role != role) continue;
        if (c.kind == ILQG_COST_EXTREME_VALUE) {
          if (c.child_count < 1 || c.child_count > 255) { *err = "row program: bad ExtremeValueCost"; return false; }
          for (int q = 0; q < c.child_count; q++) {
            const DevTerm& ch = dt[c.child_begin + q];
            want_closest(ch, c);
            emit_op(ROP_EXT_EVAL, 0, 0, q, ch, c, 0, term_pattern_of(ch.kind, ch.idx[0]));
          }
          for (int q = 0; q < c.child_count; q++) {
            const DevTerm& ch = dt[c.child_begin + q];
            want_closest(ch, c);
            const int b0 = leaf_sids(ch, c);
            emit_op(ROP_EXT_APPLY, b0, int(sids.size()) - b0, q, ch, c, 0, term_pattern_of(ch.kind, ch.idx[0]));
          }
        } else if (c.kind == ILQG_COST_WEIGHTED_CONVEX_PROXIMITY) {
          // four ops (ilqg_models.hpp): the position block carries the value, then the speed block and the two
          // position x speed blocks
          const WcpIdx ix = wcp_indices(c);
          for (int e : {ix.x1, ix.y1, ix.x2, ix.y2, ix.v1, ix.v2})
            if (e < 0 || e >= n) ok = false;
          const int which[4] = {0, ILQG_INTERNAL_WCP_SPEED, ILQG_INTERNAL_WCP_CROSS_X, ILQG_INTERNAL_WCP_CROSS_Y};
          for (int q = 0; q < 4 && ok; q++) {
            const DevTerm sub = wcp_sub_term(c, ix, which[q]);
            const int b0 = leaf_sids(sub, c);
            emit_op(ROP_TERM, b0, int(sids.size()) - b0, 0, sub, c, sub.polyline, term_pattern_of(sub.kind, sub.idx[0]));
          }
        } else if (term_is_affine(c.kind)) {
          // G and H slots as any leaf's, then 2 d scratch slots that no output word reads (the vector constraint's
          // temporaries; unmapped slots also keep such a problem off the compact rows, whose every slot feeds a word)
          const int b0 = leaf_sids(c, c);
          for (int q = 0; q < 2 * c.arg_dim; q++) {
            scratch_slots.push_back(nl);
            sids.push_back(NPS + nl++);
            add_linit(RI_VALUE, 0.0f);
          }
          emit_op(ROP_AFFINE, b0, int(sids.size()) - b0, 0, c, c, c.polyline, PAT_NONE);
        } else {
          want_closest(c, c);
          const int b0 = leaf_sids(c, c);
          emit_op(ROP_TERM, b0, int(sids.size()) - b0, 0, c, c, term_is_time_dependent(c.kind) ? c.polyline : 0,
                  term_pattern_of(c.kind, c.idx[0]));
        }
      }
    if (!ok) { *err = "row program: a term's indices are out of range or not distinct"; return false; }
    // Gradient slots first.  A merit-only evaluation (the speculative line search: rows_chunk<..., GRAD_ONLY>) touches the
    // slots of l_i and r_ij — and an affine constraint's temporaries — only, so with those numbered [0, ng) its scratch is
    // ng slots wide instead of nl.  The order in which anything is accumulated does not change: only the names do.
    int ng = 0;
    {
      std::vector<int> perm((size_t)nl, -1);
      std::vector<char> grad((size_t)nl, 0);
      for (auto& kv : l_slot) grad[kv.second] = 1;
      for (auto& kv : r_slot) grad[kv.second] = 1;
      for (int sl : scratch_slots) grad[sl] = 1;
      for (int o = 0; o < nl; o++)
        if (grad[o]) perm[o] = ng++;
      int nh = ng;
      for (int o = 0; o < nl; o++)
        if (!grad[o]) perm[o] = nh++;
      auto re = [&](int sid) { return sid >= NPS ? NPS + perm[sid - NPS] : sid; };
      for (auto& kv : q_slot) kv.second = perm[kv.second];
      for (auto& kv : l_slot) kv.second = perm[kv.second];
      for (auto& kv : R_slot) kv.second = perm[kv.second];
      for (auto& kv : r_slot) kv.second = perm[kv.second];
      for (size_t e = sids_begin; e < sids.size(); e++) sids[e] = re(sids[e]);
      for (int op = op_begin; op < int(ops.size()) / ROP_WORDS; op++) {
        int* o = ops.data() + size_t(op) * ROP_WORDS;
        for (int e = 0; e < ROP_INLINE_SIDS && e < o[RO_NSID]; e++) o[ROP_FIELDS + e] = re(o[ROP_FIELDS + e]);
      }
      std::vector<int> li(linit.begin() + size_t(li_begin) * RINIT_WORDS, linit.end());
      for (int o = 0; o < nl; o++)
        for (int q = 0; q < RINIT_WORDS; q++) linit[(size_t(li_begin) + perm[o]) * RINIT_WORDS + q] = li[size_t(o) * RINIT_WORDS + q];
    }
    if (ng > max_gslots) max_gslots = ng;
    // what this pass writes: Q_i, l_i and the R / r blocks of this player's control pairs
    const int reg_begin = int(regions.size()) / RREG_WORDS;
    {
      std::vector<short> mapQ(size_t(n) * n, short(S_ZERO));
      for (int e = 0; e < n; e++) mapQ[e + n * e] = short(S_SREG[i]);
      for (auto& kv : q_slot) mapQ[kv.first] = short(NPS + kv.second);
      add_region(RA_Q, n * n, i * n * n, mapQ);
      std::vector<short> mapl((size_t)n, (short)S_ZERO);
      for (auto& kv : l_slot) mapl[kv.first] = short(NPS + kv.second);
      add_region(RA_L, n, i * n, mapl);
      for (int q = 0; q < pt.npairs; q++) {
        if (pt.pi[q] != i) continue;
        const int mj = d.udim[pt.pj[q]];
        std::vector<short> mapR((size_t)mj * mj, (short)S_ZERO), mapr((size_t)mj, (short)S_ZERO);
        for (auto& kv : R_slot)
          if (kv.first.first == q) mapR[kv.first.second] = short(NPS + kv.second);
        for (auto& kv : r_slot)
          if (kv.first.first == q) mapr[kv.first.second] = short(NPS + kv.second);
        add_region(RA_R, mj * mj, pt.roff[q], mapR);
        add_region(RA_r, mj, pt.rgoff[q], mapr);
      }
    }
    // merit lists: |l_i|^2 over the touched entries (ascending d), |r_ii|^2 over all of the player's own controls
    {
      const int lb = int(sids.size());
      for (auto& kv : l_slot) sids.push_back(NPS + kv.second);  // std::map: ascending d
      const int rb = int(sids.size());
      const int q = pt.pii[i], mi = d.udim[i];
      for (int a = 0; a < mi; a++) sids.push_back(r_slot.count({q, a}) ? NPS + r_slot[{q, a}] : kRowSlotZero);
      int* md = merit.data() + size_t(i) * RMERIT_WORDS;
      md[0] = lb; md[1] = int(l_slot.size()); md[2] = rb; md[3] = mi;
    }
    passes.insert(passes.end(), {op_begin, int(ops.size()) / ROP_WORDS, reg_begin, int(regions.size()) / RREG_WORDS,
                                 li_begin, nl, RPASS_PLAYER, i, ng});
    if (nl > max_lslots) max_lslots = nl;
    collect_compact(reg_begin, nl);
  }
  if (NPS + max_lslots > 32000) { *err = "row program: too many slots"; return false; }
  // rows_writeout_small streams l_i, R_ij, r_ij with one word per lane
  for (int q = 0; q < pt.npairs; q++)
    if (d.udim[pt.pj[q]] * d.udim[pt.pj[q]] > 64) { *err = "row program: a control block of more than 64 words"; return false; }
  if (n > 64) { *err = "row program: more than 64 states"; return false; }

  // ---- flatten ----
  std::vector<int>& w = out->words;
  w.assign(RP_HEADER, 0);
  auto put = [&](const std::vector<int>& v) {
    const int at = int(w.size());
    w.insert(w.end(), v.begin(), v.end());
    if (v.empty()) w.push_back(0);
    return at;
  };
  w[RP_NUM_PASSES] = int(passes.size()) / RPASS_WORDS;
  w[RP_NUM_PSLOTS] = NPS;
  w[RP_MAX_LSLOTS] = max_lslots;
  w[RP_OFF_PASS] = put(passes);
  w[RP_OFF_OPS] = put(ops);
  w[RP_OFF_SIDS] = put(sids);
  w[RP_OFF_PINIT] = put(pinit);
  w[RP_OFF_LINIT] = put(linit);
  w[RP_OFF_REGIONS] = put(regions);
  w[RP_OFF_MERIT] = put(merit);
  {
    if (cdst.size() > size_t(kCompactMaxWords) || cdst.size() >= (1u << 24)) compact_ok = false;
    // what the compact rows' consumers hold: the open-loop sweep keeps the non-zero tile constants (those of Q_i, l_i)
    // in an LDS list of kCompactMaxBg entries (ilqg_lq_openloop.hpp), the forward pass scatters the Jacobian pass with
    // one word per lane (ilqg_lq.hpp) — a problem past either takes the dense rows instead of losing entries
    {
      int tile_constants = 0;
      for (size_t e = 0; e + RC_BG_WORDS <= cbg.size(); e += RC_BG_WORDS) {
        const int arr = cbg[e] >> 24;
        if (arr == RA_Q || arr == RA_L) tile_constants++;
      }
      if (tile_constants > kCompactMaxBg) compact_ok = false;
      const int jac_words = cbase.size() >= 2 ? cbase[1] - cbase[0] : int(cdst.size());
      if (jac_words > 64) compact_ok = false;
    }
    std::vector<int> blk;
    blk.push_back(compact_ok ? int(cdst.size()) : 0);
    blk.push_back(compact_ok ? int(cbg.size()) / RC_BG_WORDS : 0);
    for (int i = 0; i < N + 1; i++) blk.push_back(i < int(cbase.size()) ? cbase[i] : 0);  // Jacobian pass, players
    if (compact_ok) {
      blk.insert(blk.end(), cdst.begin(), cdst.end());
      blk.insert(blk.end(), cbg.begin(), cbg.end());
    }
    w[RP_OFF_COMPACT] = put(blk);
    out->compact_off = w[RP_OFF_COMPACT];
    out->compact_w = blk[0];
  }
  while (w.size() & 3) w.push_back(0);
  w[RP_OFF_MAPS] = int(w.size());
  for (size_t e = 0; e < maps.size(); e += 2) {
    const unsigned lo = (unsigned short)maps[e], hi = (unsigned short)(e + 1 < maps.size() ? maps[e + 1] : 0);
    w.push_back(int(lo | (hi << 16)));
  }
  w[RP_MAPS_WORDS] = int(w.size()) - w[RP_OFF_MAPS];
  while (w.size() & 3) w.push_back(0);
  w[RP_WORDS] = int(w.size());
  out->num_pslots = NPS;
  out->max_lslots = max_lslots;
  out->max_gslots = max_gslots;
  out->maps_off = w[RP_OFF_MAPS];
  out->maps_words = w[RP_MAPS_WORDS];
  return true;
}

// What a statically specialised row stage (ProgStatic<ID>, ilqg_rows.hpp) still reads at run time: the floating-point
// parameters of the ops (weight, value), their polyline segment range / table offset, the initial values of the slots
// and the literal constants of the compact rows' background.  Zeroing them leaves the STRUCTURE of the program: two
// problems whose masked programs are equal word for word run the same straight-line code.
inline void row_program_mask_parameters(std::vector<int>* words) {
  std::vector<int>& w = *words;
  if (w.size() < size_t(RP_HEADER)) return;
  const int nops = (w[RP_OFF_SIDS] - w[RP_OFF_OPS]) / ROP_WORDS;
  for (int op = 0; op < nops; op++) {
    int* o = w.data() + w[RP_OFF_OPS] + size_t(op) * ROP_WORDS;
    o[RO_WEIGHT] = 0; o[RO_VALUE] = 0; o[RO_POLY_FIRST] = 0;
    if (o[RO_MODE] == ROP_CLOSEST) o[RO_PATTERN_NSEG] = 0;
  }
  for (int at = w[RP_OFF_PINIT] + 1; at < w[RP_OFF_LINIT]; at += RINIT_WORDS) w[at] = 0;
  for (int at = w[RP_OFF_LINIT] + 1; at < w[RP_OFF_REGIONS]; at += RINIT_WORDS) w[at] = 0;
  const int cb = w[RP_OFF_COMPACT], nplayers = w[RP_NUM_PASSES] - 1;
  const int cw = w[cb + RC_W], nbg = w[cb + RC_NBG];
  const int bg0 = cb + RC_BASE + (nplayers + 1) + cw;
  for (int e = 0; e < nbg; e++) w[bg0 + e * RC_BG_WORDS + 2] = 0;
}

}  // namespace ilqg
