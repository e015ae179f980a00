// Host check of fg_dlogpdf / fg_digamma (fugue_amd/csrc/fg_math.h: the opt-in analytic gradients of all 17 distributions) against
// central differences of fg_logpdf in long-double-free plain doubles with a Richardson step, on random parameters and values inside
// every family's support; and of fg_digamma against the difference quotient of lgamma.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../fugue_amd/csrc/fg_math.h"

static double lp(uint32_t kind, double x, long long xi, double p0, double p1, double p2) {
    const double hh[5] = {0, 0, 0, 0, 0};
    return fg_logpdf(kind, false, false, x, xi, p0, p1, p2, hh);
}
// d/dt of lp along direction (dx, d0, d1, d2): Richardson-extrapolated central difference
static double fd(uint32_t kind, double x, long long xi, double p0, double p1, double p2, double dx, double d0, double d1, double d2) {
    auto f = [&](double t) { return lp(kind, x + t * dx, xi, p0 + t * d0, p1 + t * d1, p2 + t * d2); };
    const double h = 1e-4;
    const double a = (f(h) - f(-h)) / (2 * h), b = (f(h / 2) - f(-h / 2)) / h;
    return (4 * b - a) / 3;
}

int main() {
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    double worst = 0; int n_checked = 0;
    for (double x : {0.05, 0.3, 0.9, 1.0, 1.7, 2.5, 6.0, 13.2, 57.0, 400.0}) {
        const double h = 1e-5 * x, q = (std::lgamma(x + h) - std::lgamma(x - h)) / (2 * h);
        const double err = std::fabs(fg_digamma(x) - q) / (1 + std::fabs(q));
        if (err > 1e-8) { std::printf("digamma(%g) = %.15g, lgamma quotient %.15g\n", x, fg_digamma(x), q); return 1; }
    }
    if (std::fabs(fg_digamma(1.0) + 0.5772156649015329) > 1e-12 || std::fabs(fg_digamma(0.5) + 1.9635100260214235) > 1e-12) { std::printf("digamma constants\n"); return 1; }
    for (int it = 0; it < 40000; ++it) {
        const uint32_t kind = (uint32_t)(g() % 17);
        if (kind == 3) continue;                                      // Categorical: a table lookup, differentiated in the generated code
        double x = 0, p0 = 0, p1 = 0, p2 = 0; long long xi = 0;
        const double a = 0.3 + 4 * U(g), b = 0.3 + 4 * U(g), c = 0.3 + 3 * U(g), loc = 4 * U(g) - 2;
        switch (kind) {
        case 12: p0 = loc; p1 = a; x = loc + (6 * U(g) - 3) * a; break;
        case 15: p0 = loc; p1 = loc + a; x = p0 + a * (0.05 + 0.9 * U(g)); break;
        case 11: p0 = loc * 0.5; p1 = 0.2 + a * 0.3; x = std::exp(p0 + (4 * U(g) - 2) * p1); break;
        case 7: p0 = a; x = 3 * U(g) / a + 1e-3; break;
        case 0: p0 = 0.05 + 0.9 * U(g); xi = (long long)(g() & 1); break;
        case 1: p0 = a; p1 = b; x = 0.02 + 0.96 * U(g); break;
        case 8: p0 = a; p1 = b; x = 0.05 + 5 * U(g); break;
        case 2: p0 = (double)(1 + g() % 30); p1 = 0.05 + 0.9 * U(g); xi = (long long)(g() % ((unsigned long long)p0 + 1)); break;
        case 13: p0 = 0.2 + 8 * U(g); xi = (long long)(g() % 15); break;
        case 14: p0 = 0.5 + 8 * U(g); p1 = loc; p2 = c; x = loc + (8 * U(g) - 4) * c; break;
        case 4: p0 = loc; p1 = a; x = loc + (10 * U(g) - 5) * a; break;
        case 10: p0 = loc; p1 = a; x = loc + (6 * U(g) - 3) * a; if (std::fabs(x - loc) < 1e-2 * a) x = loc + 0.1 * a; break;
        case 16: p0 = a; p1 = b; x = 0.05 + 4 * U(g); break;
        case 5: p0 = 0.5 + 8 * U(g); x = 0.05 + 10 * U(g); break;
        case 9: p0 = a; p1 = b; x = 0.1 + 5 * U(g); break;
        case 6: p0 = -3; p1 = 9; xi = (long long)(g() % 13) - 3; break;
        default: continue;
        }
        const bool discrete_x = kind == 0 || kind == 2 || kind == 13 || kind == 6;
        double dx = discrete_x ? 0.0 : U(g) - 0.5, d0 = U(g) - 0.5, d1 = U(g) - 0.5, d2 = U(g) - 0.5;
        if (kind == 2 || kind == 6) d0 = 0.0;                         // an integer parameter (n; the bounds) is not differentiated
        if (kind == 6) d1 = 0.0;
        if (kind == 15) dx = 0.0;                                     // the density is flat in x; the support moves with the parameters
        const double want = fd(kind, x, xi, p0, p1, p2, dx, d0, d1, d2);
        const double got = fg_dlogpdf(kind, x, xi, p0, p1, p2, dx, d0, d1, d2);
        if (!std::isfinite(want)) continue;
        const double err = std::fabs(got - want) / (1.0 + std::fabs(want));
        if (!(err < 2e-6)) { std::printf("kind %u x %.6g xi %lld p (%.6g %.6g %.6g) d (%.3g %.3g %.3g %.3g): got %.12g want %.12g\n", kind, x, xi, p0, p1, p2, dx, d0, d1, d2, got, want); return 1; }
        worst = std::fmax(worst, err); ++n_checked;
    }
    // outside the support / invalid parameters: NaN (the force is non-finite, the transition diverges)
    if (!std::isnan(fg_dlogpdf(8, -1.0, 0, 2.0, 1.0, 0, 1, 0, 0, 0)) || !std::isnan(fg_dlogpdf(12, 0.3, 0, 0.0, -1.0, 0, 1, 0, 0, 0))) { std::printf("support\n"); return 1; }
    std::printf("fg_dlogpdf: %d directional derivatives, worst relative error %.3g\n", n_checked, worst);
    return 0;
}
