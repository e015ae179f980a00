// fg_div_const (fugue_amd/csrc/fg_math.h) must give the bits of IEEE division for every divisor that
// fg_div_const_ok admits: random operands over 80 binades plus adversarial numerators whose quotient sits next to
// a rounding boundary (k + 1/2 ulp patterns), for random divisors and for the divisors the engine actually uses
// (2h with h = 1e-5, sigmas such as 0.8, 2.5, 3.0, 0.7).  Compiled with -ffp-contract=off like the library.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../fugue_amd/csrc/fg_math.h"

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t nxt() { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static double rnd(int emin, int emax) {
    uint64_t bits = ((uint64_t)(emin + (int)(nxt() % (uint64_t)(emax - emin + 1)) + 1023) << 52) | (nxt() & 0xFFFFFFFFFFFFFull);
    if (nxt() & 1) bits |= 1ull << 63;
    double d; memcpy(&d, &bits, 8); return d;
}
int main() {
    const double fixed[] = {2e-5, 0.8, 2.5, 3.0, 0.7, 0.3, 1.0 / 3.0, 0.1, 1e-3, 7.0, 1e5};
    long bad = 0, n = 0, rejected = 0;
    for (long i = 0; i < 20000000L; i++) {
        double b = (i % 4 == 0) ? fixed[(i / 4) % 11] : rnd(-40, 40);
        if (!fg_div_const_ok(b)) { rejected++; continue; }
        double a = rnd(-60, 60);
        if (i % 3 == 0) { const double k = (double)(nxt() & 0xFFFFFFFFFFFFull) + 0.5; a = b * k; }          // quotient near a midpoint
        if (i % 7 == 0) { uint64_t u; double q = rnd(-30, 30); memcpy(&u, &q, 8); u |= 0xFFFFFull; memcpy(&q, &u, 8); a = q * b; }
        const double y = 1.0 / b;
        if (fg_div_const(a, b, y) != a / b) bad++;
        n++;
    }
    // divisors with an all-ones significand or an extreme exponent are refused, zero numerators are exact
    uint64_t ones = 0x3FEFFFFFFFFFFFFFull; double bo; memcpy(&bo, &ones, 8);
    if (fg_div_const_ok(bo) || fg_div_const_ok(1e200) || fg_div_const_ok(1e-200) || fg_div_const_ok(0.0) || !fg_div_const_ok(2e-5)) bad++;
    if (fg_div_const(0.0, 0.8, 1.0 / 0.8) != 0.0) bad++;
    std::printf("checked %ld quotients (%ld divisors refused), mismatches %ld\n", n, rejected, bad);
    return bad ? 1 : 0;
}
