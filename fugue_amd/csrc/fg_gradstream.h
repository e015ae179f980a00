// fg_gradstream.h -- fused finite-difference gradient for programs whose force terms are all
// fast Normals (FG_OP_NORMAL_FAST): ONE software-pipelined pass over a pre-built stream of
// 64-byte records computes, for every f64 coordinate i, the reference's central difference
//     g_i = (lp(q + h e_i) - lp(q - h e_i)) / (2h)                       (hmc.rs:315-327)
// over exactly the statements that read q_i, evaluating both signs of a record together
// (the same additions/multiplications, in the same order, as two sparse evaluations), and applies
// the leapfrog half-kick(s) to p_i as soon as g_i is known (hmc.rs:389,400).
//
// Pipeline (per record k):   wait -> [ fetch record k+2 | LDS-read operands of k+1 ] -> math(k)
// so the scalar-load and LDS latencies of the next records hide behind the ~30 f64 VALU
// instructions of the current one.  s_waitcnt is placed by hand: SMEM returns out of order, so
// the compiler would otherwise wait for the just-issued prefetch before the first LDS use.
#pragma once
#include "fg_interp.h"

__device__ __forceinline__ fg_u32x16 fg_fetch_grec(const FgGradRec *g, int k) {
    return *(const FG_AS4 fg_u32x16 *)(uintptr_t)(g + k);
}

struct FgGradAcc { double sp, sm, prip, prim; bool bad; };

// one record: r = {xi, mi, flags, coord, ximm, mimm, sigma, inv, lns, 0.5 ln 2pi}; xs, ms = LDS values of
// its operands; pv = p[coord] (pre-read).  Within a coordinate the prior records come first, then
// the observe records (FG_G_SWITCH on the first of them): log_prior and log_likelihood are summed
// separately and added at the end, exactly like total_log_weight (trace.rs:198-200).
// A non-finite x or mu gives z = NaN or +-inf -> lp NaN or -inf -> a non-finite g -> divergent,
// the same verdict as the reference's -inf log-density (hmc.rs:323-325); no guard is needed here.
__device__ __forceinline__ void fg_grec_math(const fg_u32x16 &r, double xs, double ms, double pv, FgGradAcc &A, double h, double hk,
                                             bool two_kicks, double *pl, int tw, double *gout, long long gstride, bool live) {
    const uint32_t fl = r[2];
    const double lns = fg_dbl(r[12], r[13]), c2 = fg_dbl(r[14], r[15]);
    // x - mu at q_i + h and q_i - h.  A slot operand holds orig +- h when it is the perturbed coordinate
    // (hmc.rs:317-319) and orig otherwise; a constant operand is its immediate.  (The generic fast Normal
    // computes `imm + slot` with imm = 0 for slots and slot = 0 for constants: the same values.)
    double dlp, dlm;
    if (fl & FG_G_M_CONST) {                              // prior-like: x = site, mu constant
        const double mimm = fg_dbl(r[6], r[7]);
        const double hx = (fl & FG_G_PERT_X) ? h : 0.0;
        dlp = (xs + hx) - mimm; dlm = (xs - hx) - mimm;
    } else if (fl & FG_G_X_CONST) {                       // likelihood-like: x observed constant, mu = site
        const double ximm = fg_dbl(r[4], r[5]);
        const double hm = (fl & FG_G_PERT_M) ? h : 0.0;
        dlp = ximm - (ms + hm); dlm = ximm - (ms - hm);
    } else {                                              // both are slots (e.g. x#i ~ N(mu, 1))
        const double hx = (fl & FG_G_PERT_X) ? h : 0.0, hm = (fl & FG_G_PERT_M) ? h : 0.0;
        dlp = (xs + hx) - (ms + hm); dlm = (xs - hx) - (ms - hm);
    }
    double zp, zm;
    if (fl & FG_G_POW2) { const double inv = fg_dbl(r[10], r[11]); zp = dlp * inv; zm = dlm * inv; }
    else { const double sg = fg_dbl(r[8], r[9]); zp = dlp / sg; zm = dlm / sg; }
    const double lpp = -0.5 * zp * zp - lns - c2;       // distribution.rs:207
    const double lpm = -0.5 * zm * zm - lns - c2;
    if (fl & FG_G_SWITCH) {                               // a real (scalar) branch, not eight v_cndmask
        A.prip = A.sp; A.prim = A.sm; A.sp = 0.0; A.sm = 0.0;
        asm volatile("" ::: "memory");
    }
    A.sp += lpp; A.sm += lpm;
    if (fl & FG_G_END) {
        // no observe record: the running sums are the prior sums and prip = prim = 0 (= log_likelihood)
        const double tp = A.prip + A.sp + 0.0, tm = A.prim + A.sm + 0.0;      // total_log_weight, log_factors = 0
        const double g = (tp - tm) / (2.0 * h);
        A.bad = A.bad || !fg_finite(g);
        double p = pv + hk * g;
        if (two_kicks) p += hk * g;
        pl[r[3] * tw] = p;
        if (gout && live) gout[(long long)r[3] * gstride] = g;
        A.sp = A.sm = A.prip = A.prim = 0.0;
    }
}

#define FG_GSTAGE(RA, XA, MA, PA, RB, XB, MB, PB, RD)                                              \
    __builtin_amdgcn_s_waitcnt(0xc07f);            /* lgkmcnt(0): RB, RC (issued a stage ago), XA, MA, PA have landed */ \
    RD = fg_fetch_grec(g, k + 3);                                                                   \
    XB = slots[RB[0] * tw]; MB = slots[RB[1] * tw]; PB = pl[RB[3] * tw];                           \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    fg_grec_math(RA, XA, MA, PA, A, h, hk, two_kicks, pl, tw, gout, gstride, live);                 \
    if (++k >= n) break;

// One gradient: every coordinate's g_i and the half-kick(s) on p_i.  Returns "some force
// component was non-finite" for this lane.  Records are fetched THREE ahead (four rotating
// 16-SGPR buffers), operands one ahead.
__device__ __forceinline__ bool fg_grad_stream(const FgProgramDev &P, double *slots, double *pl, int tw, double h, double hk,
                                               bool two_kicks, double *gout, long long gstride, bool live) {
    const FgGradRec *g = P.gstream;
    const int n = P.n_gstream;
    FgGradAcc A = {0.0, 0.0, 0.0, 0.0, false};   // running sums, stashed prior sums
    fg_u32x16 r0 = fg_fetch_grec(g, 0), r1 = fg_fetch_grec(g, 1), r2 = fg_fetch_grec(g, 2), r3;
    double x0 = slots[r0[0] * tw], m0 = slots[r0[1] * tw], p0 = pl[r0[3] * tw];
    double x1, m1, p1, x2, m2, p2, x3, m3, p3;
    int k = 0;
    for (;;) {
        FG_GSTAGE(r0, x0, m0, p0, r1, x1, m1, p1, r3)
        FG_GSTAGE(r1, x1, m1, p1, r2, x2, m2, p2, r0)
        FG_GSTAGE(r2, x2, m2, p2, r3, x3, m3, p3, r1)
        FG_GSTAGE(r3, x3, m3, p3, r0, x0, m0, p0, r2)
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    return A.bad;
}
