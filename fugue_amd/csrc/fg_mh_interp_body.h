// fg_mh_interp_body.h -- the step loop of the multi-wave MH kernel for programs without a score stream (see fg_mh_interp.hip for the
// design).  Compiled twice: into the library around the interpreter (fg_mh_interp.hip), and by hiprtc around a model compiled at run
// time (fg_jit.cpp), where FG_MHI_SCORE runs the generated statement segments and only wave 0 keeps a block of temporary rows (for the
// propose-and-score pass of a model-dependent proposal; the generated code keeps its temporaries in registers).
#ifndef FG_MHI_SCORE      /* the library: this wave's run of statements, interpreted, each term into its LDS row */
#define FG_MHI_SCORE() fg_exec<FG_MODE_SCORE, false, true, false, true>(P.ins_fast + i0, i1 - i0, P.pool, slots, tw, A, nullptr, nullptr, 0, false, nullptr, &rm, terms + (long long)s0 * tw)
#define FG_MHI_PRIV_BLOCKS(W) (W)
#endif

#define FG_MHI_MAX 8

struct FgMhi {
    int ins_off[FG_MHI_MAX + 1];     // wave w interprets instructions [ins_off[w], ins_off[w + 1]) of ins_fast ...
    int stmt_off[FG_MHI_MAX + 1];    // ... which hold statements [stmt_off[w], stmt_off[w + 1])
    const unsigned char *stmt_acc;   // [n_stmt] accumulator of each statement: 0 log_prior, 1 log_likelihood, 2 log_factors
    const int *site_ins;             // [S][2] {first instruction, count} of each site's own sample statement in P.ins (generic opcodes)
    int n_stmt;
    int direct;                      // compiled kernels only: 1 = the program has more statements than LDS has term rows -- it is scored with the
                                     // in-order accumulators themselves on wave 0, the statements of its plates shared by the waves through a ring
                                     // of 2 x 32 LDS rows (fg_jit.cpp: fg_jit_score_coop)
};

// propose_and_score (SingleSiteProposalHandler, mh.rs:298-570) behind a call, as in fg_engine.hip
static __device__ __noinline__ FgAcc3 fg_mhi_cold_mh_exec(const FgIns *ins, int n_ins, const double *pool, double *slots, int tw, bool live, FgMhCtx *mh) {
    FgAcc3 A = {0.0, 0.0, 0.0};
    fg_exec<FG_MODE_MH, false>(ins, n_ins, pool, slots, tw, A, nullptr, nullptr, 0, live, mh);
    return A;
}

// A step in which some lane's proposal needs the model (an undecided kind, PriorResample, a Categorical site with a computed table),
// out of line (the step loop keeps its registers).  Such a lane gets its proposal from its target's OWN statement, interpreted in
// the propose-and-score mode (SingleSiteProposalHandler, mh.rs:298-570) ahead of the scoring run: the statement's parameters read
// only other sites, which hold the chain's current values, so the proposed value, log q(x'|x), log q(x|x'), the decided kind and the
// accept uniform's block are those of a whole propose-and-score run.  One pass per distinct such target in the wave (the other
// lanes see no target there); the remaining lanes make their model-independent proposals as in the usual step.
struct FgMhiPre { double lqf, lqr; int kind, next_block; };
static __device__ __noinline__ FgMhiPre fg_mhi_mixed_proposals(const FgIns *ins, const int *site_ins, const double *pool, double *slots, bool live, bool walk, int target,
                                                               int tv, int kind_eff, int cat_base, int cat_K, FgMhCtx mh) {
    constexpr int tw = FG_WAVE;
    const int tslot = mh.target;
    FgMhCtx pre = mh;
    pre.target = walk ? -1 : tslot;
    unsigned long long todo = __ballot(!walk);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int tl = __builtin_amdgcn_readlane(target, leader);
        const unsigned long long same = __ballot(!walk && target == tl);
        (void)fg_mhi_cold_mh_exec(ins + site_ins[2 * tl], site_ins[2 * tl + 1], pool, slots, tw, live, &pre);
        todo &= ~same;
    }
    if (walk) {
        if (tv == 3) {                                        // usize target: resample from the constant prior table (mh.rs:516-530)
            FgStream s1 = mh.rng;
            const double uu = fg_rng_u01(s1);
            double cum = 0.0; int idx = cat_K;
            for (int i = 0; i < cat_K; ++i) { cum += pool[cat_base + i]; if (idx == cat_K && !(cum < uu)) idx = i; }
            const long long prop = idx < cat_K - 1 ? idx : cat_K - 1;
            const long long cur = fg_as_i64(mh.old_cell);
            mh.lqf += pool[cat_base + cat_K + (int)prop];
            mh.lqr += (cur < 0 || cur >= (long long)cat_K) ? FG_NEG_INF : pool[cat_base + cat_K + (int)cur];
            mh.next_block = (int)s1.c1;
            slots[tslot * tw] = fg_as_double(prop);
        } else fg_mh_walk_proposal(mh, (uint32_t)tv, kind_eff, tslot, slots, tw);
    }
    FgMhiPre r;
    r.lqf = walk ? mh.lqf : pre.lqf; r.lqr = walk ? mh.lqr : pre.lqr; r.kind = walk ? mh.kind : pre.kind; r.next_block = walk ? mh.next_block : pre.next_block;
    return r;
}

template <int OCC_UNUSED>
__device__ __forceinline__ void fg_mh_interp_mw_body(const FgProgramDev &P, const FgChainCtx &X, const FgMhDev &M, const FgMhi &seg, int iter0, int n_steps,
                                                     int n_warmup, long long *draws, int first_sample_t) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    const int np = P.n_slots - P.S + 1;                                              // private rows of a wave (temporaries, zero slot, one spare: FgRemap's layout)
    double *slots = lds + lane;                                                      // site rows [0, S) shared; wave 0's private block follows, so
                                                                                     // wave 0 may also run the program WITHOUT the remap (the general path)
    double *terms = lds + (long long)(P.S + FG_MHI_PRIV_BLOCKS(W) * np) * tw + lane;                     // one row per statement (none in the direct mode)                     // one row per statement
    FgRemap rm;
    rm.pi = 0xffffffffu; rm.n_shared = (uint32_t)P.S; rm.woff = (uint32_t)(wv * np); rm.pert = (uint32_t)(P.n_slots + wv * np);
    for (int j = wv; j < P.S; j += W) slots[P.site_slot[j] * tw] = fg_as_double(X.values[(long long)j * X.C + c]);
    if (FG_MHI_PRIV_BLOCKS(W) == W || wv == 0) slots[(P.n_slots - 1 + rm.woff) * tw] = 0.0;     // the wave's always-zero slot
    const int i0 = seg.ins_off[wv], i1 = seg.ins_off[wv + 1], s0 = seg.stmt_off[wv];
    double lw = 0.0;
    unsigned long long nacc = 0;
    if (wv == 0) lw = M.lw[c];
    __syncthreads();
    for (int t = 0; t < n_steps; ++t) {
        const int iter = iter0 + t;
        const bool adapt = iter < n_warmup;
        FgMhCtx mh;
        long long g = 0; int tslot = 0, kind0 = 0;
        mh.lqf = 0.0; mh.lqr = 0.0; mh.scale = 0.0; mh.kind = 0; mh.next_block = 2; mh.old_cell = 0.0; mh.target = 0; mh.z = 0.0;
        if (wv == 0) {                                       // the proposal: k_mh_steps' code (fg_engine.hip)
            FgStream rng = fg_stream(X.seed, X.chain0 + (uint32_t)c, (uint32_t)iter, FG_RNG_MH);
            unsigned long long ra, rb;
            fg_rng_block(rng, ra, rb);
            const int target = (int)fg_pick(ra, (uint32_t)P.S);               // sites[rng.gen_range(0..len)]  mh.rs:716
            g = (long long)target * X.C + c;
            tslot = P.site_slot[target];
            mh.target = tslot;
            { const fg_u32x4 a0 = *(const fg_u32x4 *)(M.ad + g); mh.scale = fg_dbl(a0[0], a0[1]); mh.kind = (int)a0[2]; }   // get_scale  mcmc_utils.rs:70-77
            kind0 = mh.kind;
            mh.rng = rng;                                                      // at block 1
            fg_rng_block(rng, ra, rb);
            mh.z = fg_cold_gaussian_z(ra, rb);
            mh.next_block = 2;
            mh.ov_kind = M.ov_kind; mh.ov_lo = M.ov_lo; mh.ov_hi = M.ov_hi;
            mh.old_cell = slots[tslot * tw];
            const uint32_t tv = (uint32_t)P.site_vtype[target];
            int kind_eff = FG_PROP_AUTO;
            if (tv == 0u) { kind_eff = mh.ov_kind ? mh.ov_kind[tslot] : FG_PROP_AUTO; if (kind_eff == FG_PROP_AUTO) kind_eff = mh.kind; }
            const int cat_base = P.site_cat[2 * target], cat_K = P.site_cat[2 * target + 1];
            const bool walk = tv == 0u ? (kind_eff == FG_PROP_GAUSSIAN || kind_eff == FG_PROP_LOGSPACE || kind_eff == FG_PROP_REFLECT)
                                       : (tv == 1u || tv == 2u || tv == 4u || (tv == 3u && cat_K > 0));
            // A lane whose proposal needs the model (an undecided kind, PriorResample, a Categorical site with a computed table) gets it from
            // its target's OWN statement, interpreted in the propose-and-score mode (SingleSiteProposalHandler, mh.rs:298-570) ahead of the
            // scoring run: the statement's parameters read only other sites, which hold the chain's current values, so the proposal,
            // log q(x'|x), log q(x|x'), the decided kind and the accept uniform's block are those of a whole propose-and-score run.  One
            // pass per distinct such target in the wave; the other lanes see no target there (target = -1).
            if (__all(walk)) {                                   // the usual step: every lane's proposal is model-independent
                if (tv == 3u) {                                   // usize target: resample from the constant prior table (mh.rs:516-530)
                    FgStream s1 = mh.rng;
                    const double uu = fg_rng_u01(s1);
                    double cum = 0.0; int idx = cat_K;
                    for (int i = 0; i < cat_K; ++i) { cum += P.pool[cat_base + i]; if (idx == cat_K && !(cum < uu)) idx = i; }
                    const long long prop = idx < cat_K - 1 ? idx : cat_K - 1;
                    const long long cur = fg_as_i64(mh.old_cell);
                    mh.lqf += P.pool[cat_base + cat_K + (int)prop];
                    mh.lqr += (cur < 0 || cur >= (long long)cat_K) ? FG_NEG_INF : P.pool[cat_base + cat_K + (int)cur];
                    mh.next_block = (int)s1.c1;
                    slots[tslot * tw] = fg_as_double(prop);
                } else fg_mh_walk_proposal(mh, tv, kind_eff, tslot, slots, tw);
            } else {
                const FgMhiPre r = fg_mhi_mixed_proposals(P.ins, seg.site_ins, P.pool, slots, live, walk, target, (int)tv, kind_eff, cat_base, cat_K, mh);
                mh.lqf = r.lqf; mh.lqr = r.lqr; mh.kind = r.kind; mh.next_block = r.next_block;
            }
        }
        __syncthreads();                                     // the proposed values are in the site rows
        FgAcc3 A = {0.0, 0.0, 0.0};
#ifdef FG_MHI_DIRECT_SCORE
        if (seg.direct) { FG_MHI_DIRECT_SCORE(); }             // every wave: plates are shared through a ring of LDS rows, the rest is wave 0's
        else
#endif
        { FG_MHI_SCORE(); }
        __syncthreads();                                     // every statement's term is in its row
        if (wv == 0 && !seg.direct)
            for (int k = 0; k < seg.n_stmt; ++k) {           // the three accumulators, each in program order (trace.rs:168-177)
                const double v = terms[k * tw];
                const int a = (int)seg.stmt_acc[k];
                if (a == 0) A.prior += v; else if (a == 1) A.lik += v; else A.fac += v;
            }
        if (wv == 0) {
            const double prop_lw = fg_total(A);
            const double log_alpha = prop_lw - lw + (mh.lqr - mh.lqf);         // + dim_term == 0 (fixed structure)  mh.rs:731-732
            const double u = fg_cold_u01_pair((uint32_t)X.seed, (uint32_t)(X.seed >> 32), X.chain0 + (uint32_t)c, (uint32_t)mh.next_block, (uint32_t)iter, FG_RNG_MH).a;
            const bool accept = (log_alpha >= 0.0) || (u < fg_cold_exp(log_alpha));    // mh.rs:733
            if (adapt) {                                                       // DiminishingAdaptation::update  mcmc_utils.rs:88-150
                const fg_u32x4 a1 = *(const fg_u32x4 *)((const char *)(M.ad + g) + 16);
                const uint32_t tot = a1[2] + 1u;
                const uint32_t acn = a1[3] + (accept ? 1u : 0u);
                double sc = mh.scale, ls = fg_dbl(a1[0], a1[1]);
                if (tot >= 10u) { const FgD2 r = fg_cold_mh_adapt(ls, acn, tot, M.step_tab, M.step_n); sc = r.a; ls = r.b; }
                if (live) {
                    const unsigned long long lb = (unsigned long long)__double_as_longlong(ls);
                    const fg_u32x4 w1 = { (uint32_t)lb, (uint32_t)(lb >> 32), tot, acn };
                    *(fg_u32x4 *)((char *)(M.ad + g) + 16) = w1;
                    M.ad[g].scale = sc;
                }
            }
            if (live && mh.kind != kind0) M.ad[g].kind = mh.kind;
            if (accept) { lw = prop_lw; nacc += 1ull; if (live) X.values[g] = fg_as_i64(slots[tslot * tw]); }
            else slots[tslot * tw] = mh.old_cell;
            if ((!adapt || M.rec_all) && draws && live) {
                long long *row = draws + (long long)(t - first_sample_t) * M.n_rec * X.C + c;
                for (int r = 0; r < M.n_rec; ++r) row[(long long)r * X.C] = fg_as_i64(slots[M.rec[r] * tw]);
            }
        }
        // no barrier here: the other waves read the site rows only between the two barriers above, and wave 0 writes them only outside
    }
    if (wv == 0 && live) { M.lw[c] = lw; M.n_acc[c] += nacc; }
}

