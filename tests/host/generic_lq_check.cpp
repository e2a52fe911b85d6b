// The run-time-dimensioned LQ sweeps (ilqgames_amd/csrc/ilqg_lq_generic.hpp) executed on the HOST: the kernels are
// written against an executor, and the host executor runs a phase's entries one after the other.  This checks their
// arithmetic and indexing against the oracle without a GPU (tests/test_generic_lq_host.py); the device executes the
// same phases with a workgroup barrier between them (`-m gpu` tests).  Test infrastructure: nothing links this.
//
//   generic_lq_check <in.bin> <out.bin>
// in:  int32 header [dtype(0 f32 / 1 f64), open_loop, n, N, T, adaptive, want_costates, npairs, udim[8], pi[16], pj[16]]
//      then A, Bm, Q, l, R, r, x0 (one instance, layouts of include/ilqg.h) in the dtype
// out: P, alpha, dx, costates (N n per step, open loop with want_costates only), expected_decrease — in the dtype
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../ilqgames_amd/csrc/ilqg_lq_generic.hpp"

using namespace ilqg;

template <typename T>
int run(FILE* fi, FILE* fo, const int32_t* h) {
  GenDims d{};
  d.n = h[2]; d.N = h[3]; d.T = h[4];
  const int open_loop = h[1], adaptive = h[5], want_co = h[6], npairs = h[7];
  d.uoff[0] = 0;
  for (int i = 0; i < d.N; i++) {
    d.udim[i] = h[8 + i];
    d.uoff[i + 1] = d.uoff[i] + d.udim[i];
  }
  d.m = d.uoff[d.N];
  PairTable pt{};
  pt.npairs = npairs;
  for (int i = 0; i < kMaxPlayers; i++) pt.pii[i] = -1;
  for (int q = 0; q < npairs; q++) {
    pt.pi[q] = h[16 + q];
    pt.pj[q] = h[32 + q];
    pt.roff[q] = pt.Rsz;
    pt.rgoff[q] = pt.rsz;
    pt.Rsz += d.udim[pt.pj[q]] * d.udim[pt.pj[q]];
    pt.rsz += d.udim[pt.pj[q]];
    if (pt.pi[q] == pt.pj[q]) pt.pii[pt.pi[q]] = q;
  }
  const size_t Tn = d.T, n = d.n, m = d.m, N = d.N;
  auto rd = [&](size_t count) {
    std::vector<T> v(count);
    if (fread(v.data(), sizeof(T), count, fi) != count) { fprintf(stderr, "short input\n"); exit(3); }
    return v;
  };
  const std::vector<T> A = rd(Tn * n * n), Bm = rd(Tn * n * m), Q = rd(Tn * N * n * n), l = rd(Tn * N * n),
                       R = rd(Tn * pt.Rsz), r = rd(Tn * pt.rsz), x0 = rd(n);
  std::vector<T> P(Tn * m * n, T(-7)), alpha(Tn * m, T(-7)), dx(Tn * n, T(-7)), co(Tn * N * n, T(-7)), ed(1, T(0));
  std::vector<T> scratch(Tn * size_t(gen_ol_row_elems(d.n, d.m, d.N, want_co != 0)));
  GenLQArgs<T> a{};
  a.A = A.data(); a.Bm = Bm.data(); a.Q = Q.data(); a.l = l.data(); a.R = R.data(); a.r = r.data(); a.x0 = x0.data();
  a.P = P.data(); a.alpha = alpha.data(); a.dx = dx.data();
  a.costates = (open_loop && want_co) ? co.data() : nullptr;
  a.scratch = scratch.data();
  a.ed_out = ed.data();
  a.adaptive = adaptive;
  const ParHost par;
  if (open_loop) {
    std::vector<T> sm(gen_openloop_lds_elems(d.n, d.N, d.m));
    lq_openloop_generic<T>(d, a, pt, sm.data(), par);
  } else {
    std::vector<T> sm(gen_feedback_lds_elems(d.n, d.N, d.m));
    lq_feedback_generic<T>(d, a, pt, sm.data(), par);
  }
  fwrite(P.data(), sizeof(T), P.size(), fo);
  fwrite(alpha.data(), sizeof(T), alpha.size(), fo);
  fwrite(dx.data(), sizeof(T), dx.size(), fo);
  if (open_loop && want_co) fwrite(co.data(), sizeof(T), co.size(), fo);
  fwrite(ed.data(), sizeof(T), 1, fo);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* fi = fopen(argv[1], "rb");
  FILE* fo = fopen(argv[2], "wb");
  if (!fi || !fo) return 2;
  int32_t h[48];
  if (fread(h, sizeof(int32_t), 48, fi) != 48) return 3;
  const int rc = h[0] == 0 ? run<float>(fi, fo, h) : run<double>(fi, fo, h);
  fclose(fi);
  fclose(fo);
  return rc;
}
