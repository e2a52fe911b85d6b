// Host-side unit check of csrc/ilqg_trig.hpp (its functions are __host__ __device__): the fast sine / cosine / tangent
// against the C library evaluated in long double, over the range the rollout can hand them (|x| <= kTrigFastLimit for
// double, kTrigFastLimitF for float), dense around the quadrant boundaries where the reduction matters.
// Prints the worst error in units of the last place and exits non-zero beyond the documented 2 ulp.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../ilqgames_amd/csrc/ilqg_trig.hpp"

namespace {

template <typename T>
double ulps(T got, long double want) {
  if (want == 0.0L) return got == T(0) ? 0.0 : 1e9;
  const T w = static_cast<T>(want);
  const long double ulp = std::fabs(static_cast<long double>(std::nextafter(w, T(INFINITY))) - static_cast<long double>(w));
  return static_cast<double>(std::fabs(static_cast<long double>(got) - want) / ulp);
}

template <typename T>
int check(const char* name, T limit, double bound) {
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> wide(-double(limit), double(limit)), small(-10.0, 10.0), tiny(-1e-3, 1e-3);
  double worst_s = 0, worst_c = 0, worst_t = 0;
  T at_s = 0, at_c = 0, at_t = 0;
  const long double half_pi = 1.57079632679489661923132169163975144L;
  auto one = [&](T x) {
    if (!(std::fabs(x) <= limit)) return;
    T s, c;
    ilqg::fast_sincos_core<T>(x, &s, &c);
    const long double xs = static_cast<long double>(x);
    const double es = ulps(s, sinl(xs)), ec = ulps(c, cosl(xs));
    if (es > worst_s) { worst_s = es; at_s = x; }
    if (ec > worst_c) { worst_c = ec; at_c = x; }
    // the tangent's relative accuracy is only meaningful away from its poles and zeros of cos within rounding of x
    const long double cw = cosl(xs);
    if (std::fabs(static_cast<double>(cw)) > 1e-3) {
      const double et = ulps(ilqg::fast_tan_core<T>(x), tanl(xs));
      if (et > worst_t) { worst_t = et; at_t = x; }
    }
  };
  for (int i = 0; i < 2000000; i++) {
    one(static_cast<T>(wide(rng)));
    one(static_cast<T>(small(rng)));
    one(static_cast<T>(tiny(rng)));
    // next to a multiple of pi/2
    const int k = int(rng() % 2001) - 1000;
    one(static_cast<T>(k * half_pi + tiny(rng)));
  }
  std::printf("%s: worst sin %.3f ulp at %.17g, cos %.3f ulp at %.17g, tan %.3f ulp at %.17g\n", name, worst_s,
              double(at_s), worst_c, double(at_c), worst_t, double(at_t));
  return (worst_s <= bound && worst_c <= bound && worst_t <= bound + 1.0) ? 0 : 1;
}

}  // namespace

int main() {
  int bad = 0;
  bad += check<double>("double", ilqg::kTrigFastLimit, 2.0);
  bad += check<float>("float", ilqg::kTrigFastLimitF, 2.0);
  return bad ? EXIT_FAILURE : EXIT_SUCCESS;
}
