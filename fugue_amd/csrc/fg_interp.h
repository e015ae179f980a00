// fg_interp.h -- the site-program evaluator on gfx950: one wavefront lane = one chain.
//
// Restates `run(handler, model)` (src/runtime/handler.rs:124-209) for the two handlers the
// hot path needs -- ScoreGivenTrace (src/runtime/interpreters.rs:138-163) and PriorHandler
// (:88-104) -- over a flattened program.  The program counter is wave-uniform:
//   * each 96-byte instruction is fetched with two scalar loads (s_load_dwordx16 + x8)
//     from the constant address space, one instruction AHEAD of the one being executed,
//     so the fetch latency hides behind the previous instruction's f64 math (there is
//     only one wave per SIMD at 65 536 chains -- nothing else would hide it);
//   * every branch on opcode / operand kind / flags is a scalar branch: 64 chains run in
//     lockstep, no divergence except inside the PRIOR-mode rejection samplers.
// Per-lane state lives in LDS as a [slots][64] tile of 8-byte cells: lane l reads slot k
// at lds[k*64 + l] -> 64 consecutive 8-byte words per wave access, conflict-free for
// ds_read_b64 / ds_write_b64 (MI355X_MICROARCH.md, LDS table).
//
// fg_exec is force-inlined and every kernel is written so that it has exactly ONE call
// site of it (state machines around a single evaluation loop): no function-call ABI in
// the hot path and one copy of the interpreter per kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "fg_math.h"
#include "../../include/fugue_amd.h"

#define FG_WAVE 64            /* hardware wavefront width */
#define FG_MW_MAX 16          /* waves per 64-chain tile in the multi-wave HMC kernel */
#ifndef FG_MIN_WAVES
#define FG_MIN_WAVES 2       /* __launch_bounds__ 2nd arg: waves per SIMD the register budget must allow */
#endif
#define FG_AS4 __attribute__((address_space(4)))
typedef uint32_t fg_u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t fg_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t fg_u32x4 __attribute__((ext_vector_type(4)));

struct FgAcc3 { double prior, lik, fac; };   // Trace accumulators, src/runtime/trace.rs:168-177

enum { FG_MODE_SCORE = 0, FG_MODE_PRIOR = 1, FG_MODE_MH = 2 };

// One instruction = 24 dwords, fetched by ONE vector load: lane l (< 24) holds dword l, and a field is moved to an
// SGPR with v_readlane when (and only if) it is used.  Why not scalar loads: SMEM and LDS share one counter and SMEM
// returns out of order, so any LDS access forces s_waitcnt lgkmcnt(0) -- a scalar prefetch issued at the top of an
// instruction is waited for at that instruction's first operand read and every interpreted instruction exposes a
// scalar-cache round trip (PMC: 45-63 % of the wave's cycles in s_waitcnt).  Vector loads return in order on their
// own counter (vmcnt), so instructions are fetched two ahead and the wait is usually free.
// PL: the program was staged in LDS by the kernel (fg_hmc_interp.hip): the fetch is one ds_read_b32 -- ~100 cycles and in order on
// lgkmcnt like the operand reads around it -- instead of a vector-memory round trip (measured ~1 000 cycles per interpreted
// instruction with the two-ahead global fetch, whatever the instruction did: tools/mb_interp_costs.py).
// HAZARD: the VGPR is an ordinary per-lane value to the compiler.  If it is spilled and reloaded inside DIVERGENT control flow, the
// lanes that were inactive there hold garbage, and a field read (v_readlane of a fixed lane) returns it -- seen at the 128 / 168 VGPR
// budgets as proposal kinds decided from a garbage density.  Every field is therefore read where all lanes are active, before any
// per-lane branch that uses it.
struct FgInsRegs { uint32_t w; };
template <bool PL = false>
__device__ __forceinline__ FgInsRegs fg_fetch_ins(const FgIns *prog, int pc) {
    const int l = (int)(threadIdx.x & (FG_WAVE - 1));
    FgInsRegs r;
    if (PL) r.w = ((const __attribute__((address_space(3))) uint32_t *)(prog + pc))[l < 24 ? l : 23];
    else r.w = ((const uint32_t *)(prog + pc))[l < 24 ? l : 23];
    return r;
}
__device__ __forceinline__ double fg_dbl(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
// field accessors (layout of FgIns, fg_ir.h): dw0 op, dw1..4 opnd, dw5 aux, dw6.. imm[4], dw14.. h[5]
#define FG_I_DW(r, k) ((uint32_t)__builtin_amdgcn_readlane((int)(r).w, (k)))
#define FG_I_OP(r) FG_I_DW(r, 0)
#define FG_I_OPND(r, k) FG_I_DW(r, 1 + (k))
#define FG_I_AUX(r) FG_I_DW(r, 5)
#define FG_I_IMM(r, k) fg_dbl(FG_I_DW(r, 6 + 2 * (k)), FG_I_DW(r, 7 + 2 * (k)))
__device__ __forceinline__ double fg_ins_h(const FgInsRegs &r, int k) { return fg_dbl(FG_I_DW(r, 14 + 2 * k), FG_I_DW(r, 15 + 2 * k)); }

// Row remapping of the multi-wave interpreter kernel (fg_hmc_interp.hip): the tile's SITE rows [0, n_shared) are shared by the
// waves of a workgroup, every row above (expression temporaries, Categorical tables, select options, the zero slot) is private to
// the wave -- row k of the program is row k + woff of the tile -- and reads of the coordinate under perturbation (`pi`) go to the
// wave's private row `pert` instead of the shared one.  All scalar arithmetic (the program counter is wave-uniform).
struct FgRemap { uint32_t pi, n_shared, woff, pert; };
template <bool RM>
__device__ __forceinline__ uint32_t fg_row(const FgRemap *rm, uint32_t idx) {
    if (!RM) return idx;
    return idx == rm->pi ? rm->pert : (idx < rm->n_shared ? idx : idx + rm->woff);
}
template <bool RM = false>
__device__ __forceinline__ double fg_operand(uint32_t w, double imm, const double *slots, const double *pool, int tw, const FgRemap *rm = nullptr) {
    const uint32_t kind = FG_OPND_KIND(w), idx = FG_OPND_IDX(w);
    if (kind == FG_OPND_IMM) return imm;
    if (kind == FG_OPND_SLOT_F) return slots[fg_row<RM>(rm, idx) * tw];
    if (kind == FG_OPND_SLOT_I) return (double)fg_as_i64(slots[fg_row<RM>(rm, idx) * tw]);
    return pool[idx];
}

// Integer value of an observed expression (`Value::as_bool` / `as_index`,
// crates/fugue-wasm/src/dsl.rs:68-81,1002-1021)
__device__ __forceinline__ long long fg_int_of(double v, uint32_t vtype) {
    if (vtype == 1u /*bool*/) return v != 0.0;
    return fg_finite(v) ? (long long)v : 0;
}

// ---- single-site MH proposal context (SingleSiteProposalHandler, src/inference/mh.rs:324-617) ----
// proposal kinds FG_PROP_*: include/fugue_amd.h
struct FgMhCtx {
    int target;                 // this lane's target site (sorted index)
    double scale;               // adaptation.get_scale(target)
    double z;                   // gaussian_z drawn from block 1 of the step's stream (mh.rs:128-132)
    FgStream rng;               // the step's stream positioned at block 1 (for sampler-based proposals)
    int next_block;             // block holding the accept uniform (1 if the proposal drew nothing, else >= 2)
    double lqf, lqr;            // log q(x'|x), log q(x|x')  (mh.rs:408-409)
    int kind;                   // cached f64 proposal kind of the target (0 = undecided), updated when decided
    double old_cell;            // the target's current value (to roll back on rejection)
    const int *ov_kind;         // [S] per-site overrides (SiteProposal, mh.rs:145-161) or null
    const double *ov_lo, *ov_hi;
};
// the transcendental opcodes out of line: inlined, ocml's polynomial coefficients are hoisted out of the interpreter loop into
// registers and, at the 128-VGPR budget of the multi-wave kernels, spilled -- every log then reloads them from scratch one dependent
// round trip at a time (six per log in k_hmc_interp_mw_steps).  Behind a call they are literals of the callee.
__device__ __noinline__ double fg_op_exp(double x) { return exp(x); }
__device__ __noinline__ double fg_op_log(double x) { return log(x); }
__device__ __noinline__ double fg_op_sin(double x) { return sin(x); }
__device__ __noinline__ double fg_op_cos(double x) { return cos(x); }
__device__ __noinline__ double fg_op_tanh(double x) { return tanh(x); }
__device__ __noinline__ double fg_op_pow(double x, double y) { return pow(x, y); }
// rare paths kept out of line so the interpreter stays small
__device__ __noinline__ double fg_logpdf_cold(uint32_t kind, bool hoisted, bool pow2, double xf, long long xi, double p0, double p1,
                                              double p2, double h0, double h1, double h2, double h3, double h4, bool sh, bool xh = false) {
    const double hh[5] = { h0, h1, h2, h3, h4 };
    return fg_logpdf(kind, hoisted, pow2, xf, xi, p0, p1, p2, hh, sh, xh);
}
__device__ __noinline__ long long fg_sample_cold(uint32_t kind, bool hoisted, double p0, double p1, double p2, FgStream *s) {
    return fg_sample_dist(kind, hoisted, p0, p1, p2, *s);
}
// LogSpaceWalkProposal::log_proposal_prob (mh.rs:217-223) with normal_logpdf (mh.rs:135-138)
__device__ __forceinline__ double fg_logspace_lq(double from, double to, double scale) {
    if (from <= 0.0 || to <= 0.0) return 0.0;
    const double zz = (log(to) - log(from)) / scale;
    return (-0.5 * zz * zz - log(scale) - 0.5 * log(2.0 * M_PI)) - log(to);
}

// The model-independent proposals of single-site MH for the lane's target (LDS slot `aux`, value type `vtype`):
// Gaussian / log-space / reflected random walks on f64 sites (mh.rs:183-257), the bool flip (:263-269), the u64 and
// i64 discrete walks (:285-294, :557-567).  Shared by the propose-and-score interpreter mode and by the pre-run
// proposal of k_mh_steps so that both produce the same values and consume the same RNG blocks.
__device__ __forceinline__ void fg_mh_walk_proposal(FgMhCtx &mh, uint32_t vtype, int kind, int aux, double *slots, int tw) {
    const double curd = slots[aux * tw];
    const long long curi = fg_as_i64(curd);
    if (vtype == 0u) {
        double prop = curd, f = 0.0, r = 0.0;
        if (kind == FG_PROP_GAUSSIAN) prop = curd + mh.scale * mh.z;                          // mh.rs:183-187
        else if (kind == FG_PROP_LOGSPACE) {                                                  // mh.rs:201-224
            if (curd <= 0.0) { prop = FG_MIN_POSITIVE; mh.next_block = 1; }
            else { const double pr = exp(log(curd) + mh.scale * mh.z);
                   prop = fg_finite(pr) ? fmax(pr, FG_MIN_POSITIVE) : FG_F64_MAX; }
            f = fg_logspace_lq(curd, prop, mh.scale); r = fg_logspace_lq(prop, curd, mh.scale);
        } else {                                                                              // Reflect: mh.rs:237-257
            const double lo = mh.ov_lo[aux], hi = mh.ov_hi[aux];
            double pr = curd + mh.scale * mh.z;
            if (hi - lo <= 0.0) prop = curd;
            else { for (int it = 0; it < 100000 && (pr < lo || pr > hi); ++it) {
                       if (pr < lo) pr = 2.0 * lo - pr;
                       if (pr > hi) pr = 2.0 * hi - pr; }
                   prop = pr < lo ? lo : (pr > hi ? hi : pr); }
        }
        mh.lqf += f; mh.lqr += r;
        slots[aux * tw] = prop;
    } else if (vtype == 1u) {                // FlipProposal: mh.rs:263-269 (draws nothing)
        slots[aux * tw] = fg_as_double(curi ? 0LL : 1LL);
        mh.next_block = 1;
    } else if (vtype == 2u) {                // DiscreteWalkProposal: mh.rs:285-294
        const long long k = curi + fg_f2i_sat(round(mh.scale * mh.z));
        slots[aux * tw] = fg_as_double(k >= 0 ? k : -k - 1);
    } else {                                 // i64 walk: mh.rs:557-567
        slots[aux * tw] = fg_as_double(curi + fg_f2i_sat(round(mh.scale * mh.z)));
    }
}

// The same proposals as a pure function of the target's current cell (the speculative proposer of the pipelined multi-wave MH
// kernel, fg_mh_mw2_body.h, forms a step's proposal for BOTH outcomes of the step before it): the operations of
// fg_mh_walk_proposal, in its order.  lqf / lqr start from 0.0 like a fresh FgMhCtx; nb = the block of the accept uniform.
struct FgMhCand { double prop, lqf, lqr; int nb; };
__device__ __forceinline__ FgMhCand fg_mh_walk_pure(uint32_t vtype, int kind, double curd, double scale, double z, double lo, double hi) {
    FgMhCand c; c.lqf = 0.0; c.lqr = 0.0; c.nb = 2;
    const long long curi = fg_as_i64(curd);
    if (vtype == 0u) {
        double prop = curd, f = 0.0, r = 0.0;
        if (kind == FG_PROP_GAUSSIAN) prop = curd + scale * z;                                // mh.rs:183-187
        else if (kind == FG_PROP_LOGSPACE) {                                                  // mh.rs:201-224
            if (curd <= 0.0) { prop = FG_MIN_POSITIVE; c.nb = 1; }
            else { const double pr = exp(log(curd) + scale * z);
                   prop = fg_finite(pr) ? fmax(pr, FG_MIN_POSITIVE) : FG_F64_MAX; }
            f = fg_logspace_lq(curd, prop, scale); r = fg_logspace_lq(prop, curd, scale);
        } else {                                                                              // Reflect: mh.rs:237-257
            double pr = curd + scale * z;
            if (hi - lo <= 0.0) prop = curd;
            else { for (int it = 0; it < 100000 && (pr < lo || pr > hi); ++it) {
                       if (pr < lo) pr = 2.0 * lo - pr;
                       if (pr > hi) pr = 2.0 * hi - pr; }
                   prop = pr < lo ? lo : (pr > hi ? hi : pr); }
        }
        c.lqf += f; c.lqr += r;
        c.prop = prop;
    } else if (vtype == 1u) {                // FlipProposal: mh.rs:263-269 (draws nothing)
        c.prop = fg_as_double(curi ? 0LL : 1LL);
        c.nb = 1;
    } else if (vtype == 2u) {                // DiscreteWalkProposal: mh.rs:285-294
        const long long k = curi + fg_f2i_sat(round(scale * z));
        c.prop = fg_as_double(k >= 0 ? k : -k - 1);
    } else {                                 // i64 walk: mh.rs:557-567
        c.prop = fg_as_double(curi + fg_f2i_sat(round(scale * z)));
    }
    return c;
}

// Executes instructions [0, n) of `prog` for this lane.  `slots` = &lds_tile[lane]; `tw` = tile width
// (lanes of the wave that own a chain = blockDim.x): slot k of this lane is slots[k * tw].
// `prog` must have two readable instructions past `n` (the host pads the arrays).
// logp_out: optional global column pointer (stride logp_stride) for per-site log-densities.
// TM: instead of adding to the accumulators, statement k of the run leaves its term in row k of `terms` (the caller adds the rows
// in program order: fg_mh_interp.hip splits a scoring run between waves and still forms the reference's in-order sums).
template <int MODE, bool WITH_LOGP, bool RM = false, bool PL = false, bool TM = false>
__device__ __forceinline__ void fg_exec(const FgIns *prog, int n, const double *pool, double *slots, int tw, FgAcc3 &A,
                                        FgStream *rng, double *logp_out, long long logp_stride, bool live, FgMhCtx *mh = nullptr,
                                        const FgRemap *rm = nullptr, double *terms = nullptr) {
    static_assert(!RM || MODE == FG_MODE_SCORE, "row remapping: scoring runs only");
    static_assert(!TM || MODE == FG_MODE_SCORE, "term rows: scoring runs only");
    int tk = 0;
    double acc = 0.0;
    FgInsRegs I = fg_fetch_ins<PL>(prog, 0), Inext = fg_fetch_ins<PL>(prog, 1);
    for (int pc = 0; pc < n; ++pc) {
        const FgInsRegs Inext2 = fg_fetch_ins<PL>(prog, pc + 2);     // two ahead, in order on vmcnt (PL: lgkmcnt)
        const uint32_t op = FG_I_OP(I);
        const uint32_t code = FG_INS_OPCODE(op);
        if (MODE == FG_MODE_SCORE && code == FG_OP_NORMAL_FAST) {
            // Normal, constant sigma (distribution.rs:189-208): operands are `imm + slot` (the zero slot for
            // constants), ln(sigma) hoisted, (x-mu)/sigma an exact multiply when sigma = 2^k.  A non-finite x or
            // mu makes z NaN or +-inf: NaN -> -inf by the guard, +-inf -> -inf by the formula itself.
            const double xv = FG_I_IMM(I, 0) + slots[fg_row<RM>(rm, FG_I_OPND(I, 0)) * tw];
            const double mv = FG_I_IMM(I, 1) + slots[fg_row<RM>(rm, FG_I_OPND(I, 1)) * tw];
            const double dl = xv - mv;
            double z = dl * fg_ins_h(I, 4);                                        // exact quotient when sigma = 2^k
            if (!(op & FG_F_POW2SCALE)) z = (op & FG_F_RCPSCALE) ? fg_div_const(dl, FG_I_IMM(I, 2), fg_ins_h(I, 4)) : dl / FG_I_IMM(I, 2);
            double lp = -0.5 * z * z - fg_ins_h(I, 0) - 0.5 * FG_LN_2PI;
            lp = (z != z) ? FG_NEG_INF : lp;
            if (TM) { terms[tk * tw] = lp; ++tk; }
            else if (op & FG_F_OBSERVE) A.lik += lp;
            else {
                A.prior += lp;
                if (WITH_LOGP) { const uint32_t aux_f = FG_I_AUX(I); if (live && logp_out) logp_out[(long long)aux_f * logp_stride] = lp; }   // the field is read outside the divergent branch (see FG_I_DW)
            }
        } else if (code < 17u) {
            // ---------------- sample / observe site: dist.log_prob(x) ----------------
            const bool hoisted = (op & FG_F_HOISTED) != 0u;
            const bool observe = (op & FG_F_OBSERVE) != 0u;
            const uint32_t vtype = FG_INS_VTYPE(op);
            const uint32_t xw = FG_I_OPND(I, 0);
            const uint32_t aux = FG_I_AUX(I);
            double lp;
            if (code == 3u) {                            // Categorical: distribution.rs:771-791
                const uint32_t bw = FG_I_OPND(I, 1);
                const int K = (int)FG_I_OPND(I, 2);
                const bool in_pool = FG_OPND_KIND(bw) == FG_OPND_POOL;
                const uint32_t base = in_pool ? FG_OPND_IDX(bw) : fg_row<RM>(rm, FG_OPND_IDX(bw));   // a table in slots is a run of temporaries
                long long xi;
                if (MODE == FG_MODE_PRIOR && !observe) {
                    // first i with cumulative[i] >= u, clamped to K-1 (partition_point(c < u))
                    const double u = fg_rng_u01(*rng);
                    double cum = 0.0; int idx = K;
                    for (int i = 0; i < K; ++i) {
                        const double pi = in_pool ? pool[base + i] : slots[(base + i) * tw];
                        cum += pi;
                        if (idx == K && !(cum < u)) idx = i;
                    }
                    xi = idx < K - 1 ? idx : K - 1;
                    slots[aux * tw] = fg_as_double(xi);
                } else if (FG_OPND_KIND(xw) == FG_OPND_SLOT_I) {
                    xi = fg_as_i64(slots[fg_row<RM>(rm, FG_OPND_IDX(xw)) * tw]);
                    if (MODE == FG_MODE_MH && !observe) {
                        // usize target: resample from the site's prior; lqf/lqr = prior log-probs (mh.rs:516-530)
                        const bool is_t = ((int)aux == mh->target);
                        if (__any(is_t)) {
                            if (is_t) {
                                FgStream s1 = mh->rng;
                                const double u = fg_rng_u01(s1);
                                double cum = 0.0; int idx = K;
                                for (int i = 0; i < K; ++i) {
                                    const double pi = in_pool ? pool[base + i] : slots[(base + i) * tw];
                                    cum += pi;
                                    if (idx == K && !(cum < u)) idx = i;
                                }
                                const long long prop = idx < K - 1 ? idx : K - 1;
                                const bool inv = (op & FG_F_INVALID) != 0u;
                                const double pp = in_pool ? pool[base + (int)prop] : slots[(base + (int)prop) * tw];
                                const bool cur_ok = !(xi < 0 || xi >= (long long)K);
                                const double pc = !cur_ok ? 0.0 : (in_pool ? pool[base + (int)xi] : slots[(base + (int)xi) * tw]);
                                mh->lqf += (inv || !(pp > 0.0)) ? FG_NEG_INF : log(pp);
                                mh->lqr += (inv || !(pc > 0.0)) ? FG_NEG_INF : log(pc);
                                mh->next_block = (int)s1.c1;
                                xi = prop;
                                slots[aux * tw] = fg_as_double(xi);
                            }
                        }
                    }
                } else {
                    xi = fg_int_of(fg_operand<RM>(xw, FG_I_IMM(I, 0), slots, pool, tw, rm), vtype);
                }
                if ((op & FG_F_INVALID) != 0u || xi < 0 || xi >= (long long)K) lp = FG_NEG_INF;
                else if (in_pool) lp = pool[base + K + (int)xi];          // precomputed ln p (or -inf)
                else { const double p = slots[(base + (int)xi) * tw]; lp = p > 0.0 ? log(p) : FG_NEG_INF; }
            } else {
                const double p0 = fg_operand<RM>(FG_I_OPND(I, 1), FG_I_IMM(I, 1), slots, pool, tw, rm);
                const double p1 = fg_operand<RM>(FG_I_OPND(I, 2), FG_I_IMM(I, 2), slots, pool, tw, rm);
                const double p2 = fg_operand<RM>(FG_I_OPND(I, 3), FG_I_IMM(I, 3), slots, pool, tw, rm);
                if (MODE == FG_MODE_PRIOR && !observe) {
                    const long long cell = fg_sample_cold(code, hoisted, p0, p1, p2, rng);
                    slots[aux * tw] = fg_as_double(cell);
                }
                if (MODE == FG_MODE_MH && !observe) {
                    const bool is_t = ((int)aux == mh->target);
                    if (__any(is_t)) {
                        // instruction fields are read HERE, where every lane is active (FG_I_DW)
                        const double hq0 = fg_ins_h(I, 0), hq1 = fg_ins_h(I, 1), hq2 = fg_ins_h(I, 2), hq3 = fg_ins_h(I, 3), hq4 = fg_ins_h(I, 4);
                        if (is_t) {
                            const double curd = slots[aux * tw];
                            const bool p2s = (op & FG_F_POW2SCALE) != 0u;
                            if (vtype == 0u) {                       // on_sample_f64: mh.rs:362-420
                                int kind = mh->ov_kind ? mh->ov_kind[aux] : FG_PROP_AUTO;
                                if (kind == FG_PROP_AUTO) {              // f64_kind: mh.rs:339-358
                                    kind = mh->kind;
                                    if (kind == FG_PROP_AUTO) {
                                        const double probe = ((op & FG_F_INVALID) != 0u) ? FG_NEG_INF
                                            : fg_logpdf_cold(code, hoisted, p2s, -1.0, 0, p0, p1, p2, hq0, hq1, hq2, hq3, hq4, (op & FG_F_SCALEHOIST) != 0u);
                                        kind = (curd > 0.0 && !fg_finite(probe)) ? FG_PROP_LOGSPACE : FG_PROP_GAUSSIAN;
                                        mh->kind = kind;
                                    }
                                }
                                if (kind == FG_PROP_GAUSSIAN || kind == FG_PROP_LOGSPACE || kind == FG_PROP_REFLECT) {
                                    fg_mh_walk_proposal(*mh, 0u, kind, (int)aux, slots, tw);
                                } else {                                                              // PriorResample: mh.rs:400-403
                                    FgStream s1 = mh->rng;
                                    const double prop = fg_as_double(fg_sample_cold(code, hoisted, p0, p1, p2, &s1));
                                    mh->next_block = (int)s1.c1;
                                    const bool inv = (op & FG_F_INVALID) != 0u;
                                    const bool shf = (op & FG_F_SCALEHOIST) != 0u;
                                    const double f = inv ? FG_NEG_INF : fg_logpdf_cold(code, hoisted, p2s, prop, 0, p0, p1, p2, hq0, hq1, hq2, hq3, hq4, shf);
                                    const double r = inv ? FG_NEG_INF : fg_logpdf_cold(code, hoisted, p2s, curd, 0, p0, p1, p2, hq0, hq1, hq2, hq3, hq4, shf);
                                    mh->lqf += f; mh->lqr += r;
                                    slots[aux * tw] = prop;
                                }
                            } else fg_mh_walk_proposal(*mh, vtype, 0, (int)aux, slots, tw);   // bool flip / u64 walk / i64 walk
                        }
                    }
                }
                double xf = 0.0; long long xi = 0;
                if (vtype == 0u) xf = fg_operand<RM>(xw, FG_I_IMM(I, 0), slots, pool, tw, rm);
                else if (FG_OPND_KIND(xw) == FG_OPND_SLOT_I) xi = fg_as_i64(slots[fg_row<RM>(rm, FG_OPND_IDX(xw)) * tw]);
                else xi = fg_int_of(fg_operand<RM>(xw, FG_I_IMM(I, 0), slots, pool, tw, rm), vtype);
                if ((op & FG_F_INVALID) != 0u) lp = FG_NEG_INF;
                else if (code == 12u && hoisted) {
                    // Normal with constant parameters -- the hot case (distribution.rs:189-208):
                    // ln(sigma) hoisted; (x-mu)/sigma as an exact multiply when sigma = 2^k.
                    const double hn0 = fg_ins_h(I, 0), hn4 = fg_ins_h(I, 4);     // read where every lane is active (FG_I_DW)
                    if (!fg_finite(xf)) lp = FG_NEG_INF;
                    else {
                        const double z = (op & FG_F_POW2SCALE) ? (xf - p0) * hn4 : (xf - p0) / p1;
                        lp = -0.5 * z * z - hn0 - 0.5 * FG_LN_2PI;
                    }
                } else {
                    // out of line: the other sixteen densities (lgamma, log1p, pow ...) stay out of the interpreter loop's registers
                    lp = fg_logpdf_cold(code, hoisted, (op & FG_F_POW2SCALE) != 0u, xf, xi, p0, p1, p2, fg_ins_h(I, 0), fg_ins_h(I, 1), fg_ins_h(I, 2),
                                        fg_ins_h(I, 3), fg_ins_h(I, 4), (op & FG_F_SCALEHOIST) != 0u, (op & FG_F_XHOIST) != 0u);
                }
            }
            if (TM) { terms[tk * tw] = lp; ++tk; }
            else if (observe) A.lik += lp;               // interpreters.rs:76-83
            else {
                A.prior += lp;                           // interpreters.rs:150-158
                if (WITH_LOGP) { if (live && logp_out) logp_out[(long long)aux * logp_stride] = lp; }
            }
        } else {
            const double x0 = fg_operand<RM>(FG_I_OPND(I, 0), FG_I_IMM(I, 0), slots, pool, tw, rm);
            switch (code) {
            case FG_OP_FACTOR: if (TM) { terms[tk * tw] = x0; ++tk; } else A.fac += x0; break;       // Handler::on_factor
            case FG_OP_LOAD: acc = x0; break;
            case FG_OP_ADD: acc = acc + x0; break;
            case FG_OP_SUB: acc = acc - x0; break;
            case FG_OP_MUL: acc = acc * x0; break;
            case FG_OP_DIV: acc = acc / x0; break;
            case FG_OP_RSUB: acc = x0 - acc; break;
            case FG_OP_RDIV: acc = x0 / acc; break;
            case FG_OP_NEG: acc = -acc; break;
            case FG_OP_EXP: acc = fg_op_exp(acc); break;
            case FG_OP_LN: acc = fg_op_log(acc); break;
            case FG_OP_SQRT: acc = sqrt(acc); break;
            case FG_OP_ABS: acc = fabs(acc); break;
            case FG_OP_FLOOR: acc = floor(acc); break;
            case FG_OP_SIN: acc = fg_op_sin(acc); break;
            case FG_OP_COS: acc = fg_op_cos(acc); break;
            case FG_OP_TANH: acc = fg_op_tanh(acc); break;
            case FG_OP_POW: acc = fg_op_pow(acc, x0); break;
            case FG_OP_RPOW: acc = fg_op_pow(x0, acc); break;
            case FG_OP_MIN: acc = fmin(acc, x0); break;
            case FG_OP_MAX: acc = fmax(acc, x0); break;
            case FG_OP_CLAMP: acc = fg_clamp(acc, x0, fg_operand<RM>(FG_I_OPND(I, 1), FG_I_IMM(I, 1), slots, pool, tw, rm)); break;
            case FG_OP_MAC: { const double t = x0 * fg_operand<RM>(FG_I_OPND(I, 1), FG_I_IMM(I, 1), slots, pool, tw, rm);
                              acc = acc + t; break; }
            case FG_OP_STORE: slots[fg_row<RM>(rm, FG_I_AUX(I)) * tw] = acc; break;
            case FG_OP_GATHER: { const int k = (int)FG_I_OPND(I, 1);
                                 const bool ok = (acc >= 0.0) && (acc < (double)k) && (acc == floor(acc));
                                 const int j = ok ? (int)acc : 0;
                                 const double v = slots[(fg_row<RM>(rm, FG_I_AUX(I)) + j) * tw];     // the options are a run of temporaries
                                 acc = ok ? v : NAN; break; }
            case FG_OP_CONSTLIK: if (TM) { terms[tk * tw] = FG_I_IMM(I, 0); ++tk; } else A.lik += FG_I_IMM(I, 0); break;
            case FG_OP_DOT: {                             // n MACs (slot x constant), terms fetched 4 at a time by scalar loads
                const int n = (int)FG_I_OPND(I, 1);
                const FG_AS4 char *tb = (const FG_AS4 char *)(uintptr_t)(pool + FG_I_AUX(I));
                int t = 0;
                for (; t + 4 <= n; t += 4) {
                    const fg_u32x16 q = *(const FG_AS4 fg_u32x16 *)(tb + 16 * t);
                    const double v0 = slots[fg_row<RM>(rm, q[0]) * tw], v1 = slots[fg_row<RM>(rm, q[4]) * tw], v2 = slots[fg_row<RM>(rm, q[8]) * tw], v3 = slots[fg_row<RM>(rm, q[12]) * tw];
                    acc = acc + v0 * fg_dbl(q[2], q[3]);
                    acc = acc + v1 * fg_dbl(q[6], q[7]);
                    acc = acc + v2 * fg_dbl(q[10], q[11]);
                    acc = acc + v3 * fg_dbl(q[14], q[15]);
                }
                for (; t < n; ++t) {
                    const fg_u32x4 q = *(const FG_AS4 fg_u32x4 *)(tb + 16 * t);
                    acc = acc + slots[fg_row<RM>(rm, q[0]) * tw] * fg_dbl(q[2], q[3]);
                }
                break; }
            default: break;
            }
        }
        I = Inext; Inext = Inext2;
    }
}

// total_log_weight (src/runtime/trace.rs:198-200)
__device__ __forceinline__ double fg_total(const FgAcc3 &A) { return A.prior + A.lik + A.fac; }
