// Host-side unit check of csrc/ilqg_trig.hpp (its functions are __host__ __device__): the fast sine / cosine / tangent
// against the C library evaluated in long double, over the range the rollout can hand them (|x| <= kTrigFastLimit for
// double, kTrigFastLimitF for float), dense around the quadrant boundaries where the reduction matters.
// Prints the worst error in units of the last place and exits non-zero beyond the documented 2 ulp.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
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
    // ... and next to one anywhere in the range (the value nearest the multiple, and a neighbour)
    const long double kk = static_cast<long double>(rng() % static_cast<unsigned long long>(double(limit) / 1.5707963267948966)) + 1.0L;
    one(static_cast<T>(kk * half_pi));
    one(static_cast<T>(-kk * half_pi + tiny(rng)));
  }
  std::printf("%s: worst sin %.3f ulp at %.17g, cos %.3f ulp at %.17g, tan %.3f ulp at %.17g\n", name, worst_s,
              double(at_s), worst_c, double(at_c), worst_t, double(at_t));
  return (worst_s <= bound && worst_c <= bound && worst_t <= bound + 1.0) ? 0 : 1;
}

}  // namespace

// Beyond the fast range: the total functions (fast_sincos / fast_tan take the large reduction there, trig_reduce_large)
// over every binade up to the largest finite argument, next to multiples of pi/2 (where the reduction cancels), at the
// range limit itself, and on the non-finite arguments.
template <typename T>
int check_large(const char* name, T limit, double bound) {
  std::mt19937_64 rng(777);
  std::uniform_real_distribution<double> mant(1.0, 2.0), tiny(-1e-3, 1e-3);
  const long double half_pi = 1.57079632679489661923132169163975144L;
  double worst_s = 0, worst_c = 0, worst_t = 0;
  T at_s = 0, at_c = 0, at_t = 0;
  long count = 0;
  auto one = [&](T x) {
    if (!std::isfinite(x) || !(std::fabs(x) > limit)) return;
    count++;
    T s, c;
    ilqg::fast_sincos(x, &s, &c);
    const long double xs = static_cast<long double>(x);
    const double es = ulps(s, sinl(xs)), ec = ulps(c, cosl(xs));
    if (es > worst_s) { worst_s = es; at_s = x; }
    if (ec > worst_c) { worst_c = ec; at_c = x; }
    const long double cw = cosl(xs), sw = sinl(xs);
    if (std::fabs(static_cast<double>(cw)) > 1e-3 && std::fabs(static_cast<double>(sw)) > 1e-3) {
      const double et = ulps(ilqg::fast_tan(x), tanl(xs));
      if (et > worst_t) { worst_t = et; at_t = x; }
    }
  };
  const int max_exp = std::numeric_limits<T>::max_exponent - 1;
  for (int e = 10; e <= max_exp; e++)
    for (int i = 0; i < 4000; i++) {
      const T x = static_cast<T>(std::ldexp(mant(rng), e));
      one(x);
      one(-x);
    }
  for (int i = 0; i < 1000000; i++) {
    const long double k = static_cast<long double>(rng() % (1ull << (sizeof(T) == 8 ? 50 : 22))) + 1.0L;
    one(static_cast<T>(k * half_pi));                  // the double / float nearest a multiple of pi/2
    one(static_cast<T>(k * half_pi + tiny(rng)));
  }
  one(std::nextafter(limit, std::numeric_limits<T>::infinity()));
  one(std::numeric_limits<T>::max());
  one(-std::numeric_limits<T>::max());
  if (sizeof(T) == 8) {
    one(static_cast<T>(0x1.6ac5b262ca1ffp+849));  // the classical worst case of a double's reduction (~2^-62 from a multiple of pi/2)
    one(static_cast<T>(6381956970095103.0 * 64.0));
  }
  int bad = 0;
  for (T x : {std::numeric_limits<T>::infinity(), -std::numeric_limits<T>::infinity(), std::numeric_limits<T>::quiet_NaN()}) {
    T s = 0, c = 0;
    ilqg::fast_sincos(x, &s, &c);
    if (!std::isnan(s) || !std::isnan(c) || !std::isnan(ilqg::fast_tan(x))) bad++;
  }
  // continuity with the fast path at the limit: the large reduction also holds below it
  for (int i = 0; i < (sizeof(T) == 8 ? 200000 : 0); i++) {
    const double x = std::ldexp(mant(rng), 10 + int(rng() % 7));
    double s0, c0, s1, c1;
    ilqg::fast_sincos_core<double>(x, &s0, &c0);
    ilqg::large_sincos(x, &s1, &c1);
    if (ulps(s1, sinl((long double)x)) > bound || ulps(c1, cosl((long double)x)) > bound) bad++;
    (void)s0; (void)c0;
  }
  std::printf("%s beyond %g (%ld arguments): worst sin %.3f ulp at %.17g, cos %.3f ulp at %.17g, tan %.3f ulp at %.17g; %d bad\n",
              name, double(limit), count, worst_s, double(at_s), worst_c, double(at_c), worst_t, double(at_t), bad);
  // (the tangent beyond the range is the plain quotient of the two: their errors add)
  return (worst_s <= bound && worst_c <= bound && worst_t <= 2.0 * bound && bad == 0) ? 0 : 1;
}

int main() {
  int bad = 0;
  bad += check<double>("double", ilqg::kTrigFastLimit, 2.0);
  bad += check<float>("float", ilqg::kTrigFastLimitF, 2.0);
  bad += check_large<double>("double", ilqg::kTrigFastLimit, 2.0);
  bad += check_large<float>("float", ilqg::kTrigFastLimitF, 1.0);
  return bad ? EXIT_FAILURE : EXIT_SUCCESS;
}
