// fg_dev_types.h -- kernel-argument structs and small device helpers shared by the engine's translation units AND the run-time
// compiled model kernels (fg_jit.cpp embeds this header's text: keep it free of host-only includes).
#pragma once
#include "fg_ir.h"

struct FgChainCtx {
    long long C;          // chains in this engine
    uint32_t chain0;      // global id of chain 0 (RNG stream key)
    unsigned long long seed;
    long long *values;    // [S][C]
    double *gtile;        // global-memory tiles [tiles][gtile_rows][64] of the one-wave kernels, or null (tiles in LDS)
    int gtile_rows;
};


struct FgHmcDev {
    double *lj, *eps, *frozen, *da_mu, *da_leb, *da_hbar;
    unsigned long long *da_m;
    double *m_inv, *mass_sqrt, *w_mean, *w_m2;     // [d][C] or null
    unsigned long long *w_n;
    double *alpha_sum; unsigned long long *n_div;
    double *p0_scratch;                             // [d][C] (eps search / injected momentum)
    int L; double h, target; int grad_mode; int use_mass;
};


// in-order kinetic energy and momentum draw shared by the HMC kernels
__device__ __forceinline__ double fg_kinetic(const FgProgramDev &P, const double *pl, int tw, const double *m_inv, long long C) {
    double s = 0.0;
    if (m_inv) {
        for (int i = 0; i < P.d; ++i) { const double p = pl[i * tw]; s += p * p * m_inv[(long long)i * C]; }
    } else {                                              // identity mass: p*p*1.0 == p*p exactly
#pragma unroll 8
        for (int i = 0; i < P.d; ++i) { const double p = pl[i * tw]; s += p * p; }
    }
    return 0.5 * s;
}


// DiminishingAdaptation's state of one (site, chain) (mcmc_utils.rs:40-62) + the decided proposal kind, as two 16-byte groups:
// what a proposal reads {scale, kind} and what an update reads and writes {log_scale, total, accepted} are one 16-byte access each
// (lanes hold different sites: every access of a wave touches 64 different lines, so the NUMBER of accesses is the cost).
struct alignas(16) FgMhAdapt { double scale; int32_t kind, pad; double log_scale; uint32_t tot, acc; };
static_assert(sizeof(FgMhAdapt) == 32, "FgMhAdapt is two 16-byte groups");

struct FgMhDev {
    double *lw;
    FgMhAdapt *ad;                                       // [S][C]
    const int *ov_kind; const double *ov_lo, *ov_hi;    // [S] overrides or null
    unsigned long long *n_acc;                           // [C] accepted proposals
    const int *rec;                                      // [n_rec] recorded sites
    int n_rec;
    int rec_all;                                         // record during adaptation too (fg_mh_set_recording: incremental sessions)
    const double *step_tab; uint32_t step_n;             // 1 / n^0.7 for n < step_n (DiminishingAdaptation's step, mcmc_utils.rs:118)
};

// the tile's site values <-> its LDS slot rows (slot row r of lane l: slots[r * tw]; `slots` already points at the lane)
__device__ __forceinline__ void fg_load_values(const FgProgramDev &P, const FgChainCtx &X, long long c, double *slots, int tw) {
    for (int j = 0; j < P.S; ++j) slots[P.site_slot[j] * tw] = fg_as_double(X.values[(long long)j * X.C + c]);
    slots[(P.n_slots - 1) * tw] = 0.0;                   // the always-zero slot (constant operands of fast opcodes)
}
__device__ __forceinline__ void fg_store_values(const FgProgramDev &P, const FgChainCtx &X, long long c, const double *slots, int tw) {
    for (int j = 0; j < P.S; ++j) X.values[(long long)j * X.C + c] = fg_as_i64(slots[P.site_slot[j] * tw]);
}

// ---- adaptive_smc (fg_smc.hip; the rejuvenation kernel of a model compiled at run time, fg_hmc_jit_body.h)
struct FgSmcScalars {      // device-resident scalars of one SMC run
    double beta, bnew, lo, hi, mid, one, target_ess;
    double log_evidence, log_norm, lse1, lse2, ess;
    int done, force_one;
    // lookahead bisection (k_smc_ess_pass): candidates of the current pass, bisection steps taken, arrival ticket
    double cand[8];
    int n_cand, iters, first;
    unsigned int ticket;
    double dbeta;          // bnew - beta of the reweight in flight (k_smc_finish phase 3 advances beta itself; k_smc_apply uses this)
    double fin_max;        // max of the log-weights the last reweight left (k_smc_ess2_apply -> k_smc_final_norm)
    int need_sum, pad2;    // k_smc_ess2_apply: beta' is no candidate of the passes -- the separate reduction kernels take the step
    double beta2[2];       // beta as the passes and k_smc_ess2_apply read / write it: two slots that swap from step to step
};
// rejuvenation: tempered_single_site_mh (smc.rs:631-688), one move per particle
struct FgSmcDev {
    double *ll, *lprior;              // [N]
    double *scale, *log_scale;        // [S] shared DiminishingAdaptation (smc.rs:482)
    long long *acc, *tot;             // [S]
    unsigned int *sw_n, *sw_a;        // [S] per-sweep proposal / accept counts
    unsigned int *blk;                // [n_blocks][2][S] per-block proposal / accept counts of the sweep in flight
    int S;
};
#define FG_SMC_HIST 320               /* sites of a program (LDS bounds a tile to 320 cells) */
