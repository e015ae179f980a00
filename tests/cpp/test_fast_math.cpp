// Host check of fg_fast_log / fg_fast_sincos (fugue_amd/csrc/fg_math.h) against libm, on the inputs the normal generators
// feed them: u in (0, 1] on the 2^-53 grid (and down to 1e-10 for gaussian_z), angles 2 pi u in [0, 2 pi).
// Errors are measured in ulps of the libm result (long double reference).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../../fugue_amd/csrc/fg_math.h"

static double ulps(double got, long double want) {
    if (want == 0.0L) return got == 0.0 ? 0.0 : 1e9;
    int e; std::frexp((double)want, &e);
    const long double u = std::ldexp(1.0L, e - 53);
    return (double)(fabsl((long double)got - want) / u);
}

int main() {
    std::mt19937_64 g(12345);
    double worst_log = 0, worst_sin = 0, worst_cos = 0;
    double abs_sin = 0, abs_cos = 0;
    const int N = 4000000;
    for (int i = 0; i < N; i++) {
        const uint64_t a = g(), b = g();
        double u1 = ((double)(a >> 11) + 1.0) * 0x1.0p-53;
        if (i % 7 == 0) u1 = std::ldexp(u1, -(int)(a % 40));          // small arguments too
        if (i % 1000 == 1) u1 = 1.0 - std::ldexp((double)(a % 4096), -53);   // next to 1
        if (i == 5) u1 = 1.0;
        if (i == 6) u1 = 1e-10;
        const double l = fg_fast_log(u1);
        worst_log = std::fmax(worst_log, ulps(l, logl((long double)u1)));
        double u2 = (double)(b >> 11) * 0x1.0p-53;
        if (i % 1000 == 2) u2 = 0.25 * (double)(b % 4) + std::ldexp((double)((b >> 8) % 4096), -53);   // next to multiples of pi / 2
        const double th = 2.0 * M_PI * u2;
        double sn, cs;
        fg_fast_sincos(th, sn, cs);
        const long double rs = sinl((long double)th), rc = cosl((long double)th);
        abs_sin = std::fmax(abs_sin, (double)fabsl(sn - rs)); abs_cos = std::fmax(abs_cos, (double)fabsl(cs - rc));
        if (fabsl(rs) > 1e-8L) worst_sin = std::fmax(worst_sin, ulps(sn, rs));
        if (fabsl(rc) > 1e-8L) worst_cos = std::fmax(worst_cos, ulps(cs, rc));
    }
    std::printf("log max %.3f ulp; sin max %.3f ulp (abs %.3g); cos max %.3f ulp (abs %.3g)\n", worst_log, worst_sin, abs_sin, worst_cos, abs_cos);
    const bool ok = worst_log < 1.0 && worst_sin < 1.0 && worst_cos < 1.0 && abs_sin < 2.3e-16 && abs_cos < 2.3e-16;
    std::printf(ok ? "fast math ok\n" : "fast math FAILED\n");
    return ok ? 0 : 1;
}
