// Host-side check of the closed-form quartic solver against polynomials built
// from known roots.  Build: hipcc -O2 -std=c++17 tools/quartic_check.hip -o /tmp/quartic_check
#include <cstdio>
#include <random>
#include <algorithm>
#include <vector>
#include "../slam_amd/csrc/quartic.h"

using namespace slamhip_dev;

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  long n_cases = 0, n_count_bad = 0;
  double worst = 0.0;
  for (int kind = 0; kind < 5; ++kind) {
    for (int t = 0; t < 200000; ++t) {
      const double sc = std::pow(10.0, 2.0 * U(rng));
      long double c[5] = {1, 0, 0, 0, 0};  // ascending coefficients of the product so far
      int deg = 0;
      std::vector<double> real_roots;
      auto mul_lin = [&](long double r) {      // (x - r)
        for (int i = deg + 1; i >= 1; --i) c[i] = c[i - 1] - r * c[i];
        c[0] = -r * c[0];
        ++deg;
      };
      auto mul_quad = [&](long double re, long double im) {   // (x-re)^2 + im^2
        long double b1 = -2 * re, b0 = re * re + im * im, n[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i <= deg; ++i) { n[i] += c[i] * b0; n[i + 1] += c[i] * b1; n[i + 2] += c[i]; }
        for (int i = 0; i < 5; ++i) c[i] = n[i];
        deg += 2;
      };
      if (kind == 0) { for (int k = 0; k < 4; ++k) { double r = sc * U(rng); real_roots.push_back(r); mul_lin(r); } }
      if (kind == 1) { for (int k = 0; k < 2; ++k) { double r = sc * U(rng); real_roots.push_back(r); mul_lin(r); }
                       mul_quad(sc * U(rng), sc * (0.05 + std::fabs(U(rng)))); }
      if (kind == 2) { mul_quad(sc * U(rng), sc * (0.05 + std::fabs(U(rng)))); mul_quad(sc * U(rng), sc * (0.05 + std::fabs(U(rng)))); }
      if (kind == 3) { for (int k = 0; k < 4; ++k) { double r = sc * (0.1 + 5.0 * std::fabs(U(rng))); real_roots.push_back(r); mul_lin(r); } }  // positive, P3P-like
      if (kind == 4) { double r = sc * U(rng), s = sc * U(rng); real_roots = {r, s, -r, -s}; for (double z : real_roots) mul_lin(z); }  // biquadratic
      const double lead = 0.5 + std::fabs(U(rng)) * 100.0;
      double q[5], out[4];
      for (int i = 0; i < 5; ++i) q[i] = (double)(c[i] * lead);
      const int n = quartic_real_roots_closed(q, out);
      std::sort(real_roots.begin(), real_roots.end());
      // separation of the true roots decides how well conditioned the case is
      double sep = 1e300;
      for (size_t i = 1; i < real_roots.size(); ++i) sep = std::min(sep, real_roots[i] - real_roots[i - 1]);
      const bool well = real_roots.size() < 2 || sep > 1e-3 * sc;
      if (!well) continue;
      ++n_cases;
      if (n != (int)real_roots.size()) { ++n_count_bad; continue; }
      for (int i = 0; i < n; ++i) {
        // error in units of the root's own rounding sensitivity: eps * sum|c_i x^i| / |x f'(x)|
        const long double x = real_roots[i];
        long double sum = 0, df = 0, xp = 1;
        for (int k = 0; k <= 4; ++k) { sum += fabsl(c[k]) * fabsl(xp); if (k < 4) df += (k + 1) * c[k + 1] * xp; xp *= x; }
        const double cond = (double)(sum / std::max((long double)1e-300, fabsl(df)));
        const double e = std::fabs(out[i] - real_roots[i]) / std::max(1e-300, 2.2e-16 * cond);
        worst = std::max(worst, e);
      }
    }
    std::printf("kind %d: cases %ld, wrong root count %ld, worst error / (eps * condition) %.3e\n", kind, n_cases, n_count_bad, worst);
  }
  return (n_count_bad == 0 && worst < 64.0) ? 0 : 1;
}
