// fg_mh_mw2_body.h -- the step loop of the multi-wave MH kernel with the step's serial recipe SPLIT OVER WAVES (round 4).
// Same kernel arguments, same tile layout, same phase B (fg_mh_mw_body.h: fg_mh_terms / fg_mh_group4 over the kind-sorted score
// stream, or generated statements), same results bit for bit -- what changes is who does what in phase A:
//
//   decider  (wave 0)   adds the terms of step t-1 in program order, decides (mh.rs:731-733), commits or rolls back, then SELECTS the
//                       proposal of step t from two candidates that are already in LDS;
//   proposer (wave 1)   while the decider adds: forms the proposal of step t (mh.rs:183-294, 516-530, 557-567) for BOTH outcomes of
//                       step t-1.  Only lanes whose target is the site step t-1 moved see two different candidates (cell after an
//                       accepted / a rejected step t-1; adaptation scale after DiminishingAdaptation::update with / without the
//                       acceptance, mcmc_utils.rs:88-150); every other lane's candidates are one number.  The proposer owns ALL
//                       traffic of the adaptation state (so its loads and stores are one wave's program order): in phase B it learns the
//                       decision, stores the chosen outcome of step t-1's update, forms both outcomes of step t's update and
//                       prefetches {scale, kind} of step t+1's target;
//   random-number waves (W-1, W-2) as before, one step ahead; block 1 is generated once (programs with a Categorical site: twice);
//   the accept test `log_alpha >= 0 || u < exp(log_alpha)` is decided from ln u, which the random-number wave forms off the path:
//                       ln u < log_alpha - m accepts and ln u > log_alpha + m rejects for m = 1e-9 (1 + |log_alpha|), three orders
//                       above the error of the two transcendentals; inside the margin (probability ~1e-9 per lane and step), for u = 0, a NaN and
//                       when the proposal consumed another block than expected, the decider evaluates the reference's own expression.
//                       The decisions are the reference's, not an approximation of them.
//
// Waves meet at the two workgroup barriers of a step; inside phase A the decider waits for the candidates (and, SPLIT, for the
// log_likelihood sum of a second adding wave) on a tag word in LDS: a writer stores its rows, waits for the LDS counter, stores the
// tag; a reader loads the tag FIRST and the rows behind it in one batch (a wave's LDS instructions execute in order) and repeats the
// batch while the tag is stale.  An undecided proposal kind (f64_kind, mh.rs:339-358: decided once per (site, chain) from the
// chain's state) makes the proposer wait for the decision instead of speculating.
//
// Exchange rows behind the term rows (NR = 3 + has_cat + has_bool per random-number buffer, two buffers by step parity):
//   R(p) + 0  {target 16 | LDS slot 16 | value type 3 | K 7 | resampled index 7} of the step    R + 1  gaussian_z
//   R + 2  ln u of block 2    [R + 3  Categorical target: prior log-probability of the resampled index]    [R + last  ln u of block 1]
//   CB + 0 / 1  proposed value after an accepted / a rejected step t-1    CB + 2 / 3  log q(x'|x)    CB + 4 / 5  log q(x|x')
//   CB + 6  the target's cell as the proposer read it (the cell the proposal replaces, unless step t-1 moved the same site: then the
//           decider has both possible values in registers)    CB + 7  {tag | blocks of the two accept uniforms | slot | target == step t-1's}
//   DR  {tag | accepted}    [LK, LK + 1  log_likelihood sum and its tag]
#define FG_MH2_TAG(it) ((uint32_t)(it) + 1u)
struct FgMh2Spec { static constexpr bool after = false; };        // candidates(): speculative form / behind the decision (compile-time: two bodies)
struct FgMh2After { static constexpr bool after = true; };

// out of line, like everything transcendental of a step (fg_cold.h): inlined, the polynomial constants of ln / exp are hoisted out of
// the step loop into registers the kernel does not have (57 spilled VGPRs at the 128 of four waves per SIMD)
static __device__ __noinline__ FgMhCand fg_cold_walk_pure(uint32_t vtype, int kind, double curd, double scale, double z, double lo, double hi) {
    return fg_mh_walk_pure(vtype, kind, curd, scale, z, lo, hi);
}

template <int RK, bool SPLIT /* the two in-order sums on two waves */>
__device__ __forceinline__ void fg_mh_mw2_body(const FgProgramDev &P, const FgChainCtx &X, const FgMhDev &M, const FgGradRec *srt /* the kind-sorted score stream */, const FgMhSeg &seg, int iter0, int n_steps, int n_warmup,
                                               long long *draws, int first_sample_t, int exp_mask /* 32: no wave priorities (A/B); 64: phase-B priority; 128 / 256: staggered start; 512: the program has a Categorical site with a constant table; 1024: ... a bool site */,
                                               int pool_n /* > 0: the constant pool (pool_n doubles) is staged into LDS behind the exchange rows */) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    const int n_cu = seg.n_cu;                                      // (terms without a row: FgMhSeg)
    const int n_s = P.n_sstream - n_cu, n_pri = P.n_prior_terms - n_cu, n_lik = n_s - n_pri;
    const bool has_cat = (exp_mask & 512) != 0, has_bool = (exp_mask & 1024) != 0;
    const int NR = 3 + (has_cat ? 1 : 0) + (has_bool ? 1 : 0);
    double *slots = lds + lane;
    double *terms = lds + (long long)P.n_slots * tw + lane;
    double *xch = terms + (long long)n_s * tw;
    double *CB = xch + (long long)(2 * NR) * tw;                    // candidates of the step being proposed
    double *DR = CB + 8 * tw;                                       // the decision about the step being finished
    double *LK = DR + tw;                                           // SPLIT: log_likelihood sum, its tag
    const int xrows = 2 * NR + 9 + (SPLIT ? 2 : 0);
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    const bool b_prio = wv != 0 && (exp_mask & 64) != 0;
    const int rng_wave = W - 1;
    const int rng_wave1 = W >= 3 ? W - 2 : W - 1;                  // the wave of part 1
    const int w_sum = SPLIT ? 1 : -1;
    const int w_pro = (SPLIT && W > 2) ? 2 : 1;
    const int rng_wave2 = W >= 6 ? W - 3 : w_pro;                  // the wave of part 2 (the proposer has nothing else to do in phase A)

    double *pool_l = lds + (long long)(P.n_slots + n_s + xrows) * tw;
    auto pool_rd = [&](int idx) __attribute__((always_inline)) { return pool_n > 0 ? pool_l[idx] : P.pool[idx]; };
    // rows that cross between waves inside a phase: volatile LDS accesses (the cast keeps them ds_read / ds_write: a volatile access through
    // a generic pointer is a FLAT one, hundreds of cycles and a vmcnt wait each)
    typedef __attribute__((address_space(3))) double fg_lds_double;
    auto vread = [](const double *p) __attribute__((always_inline)) { return *(const volatile fg_lds_double *)p; };
    auto vwrite = [](double *p, double v) __attribute__((always_inline)) { *(volatile fg_lds_double *)p = v; };

    // Everything of step `it` that does not depend on the chain's state -> buffer (it & 1).
    //   part 0: gen_range target (mh.rs:716) with its site-table entries and -- for a Categorical target -- the index resampled from the
    //           constant prior table (block 1's uniform) with its prior log-probability (mh.rs:516-530);
    //   part 1: gaussian_z (mh.rs:128-132) from block 1;
    //   part 2: ln of the accept uniform of block 2 (and of block 1 where a bool site can be the target: its flip draws nothing and the
    //           accept uniform is block 1's).
    // Three waves where the tile has them (a part is ~150 instructions of a lone wave: together they were longer than the decider's path).
    auto publish_rng = [&](int it, int part) __attribute__((always_inline)) {
        double *b = xch + (long long)(NR * (it & 1)) * tw;
        FgStream rng; rng.k0 = sk0; rng.k1 = sk1; rng.c0 = gchain; rng.c2 = (uint32_t)it; rng.c3 = FG_RNG_MH;
        unsigned long long ra, rb;
        if (part == 0) {
            rng.c1 = 0; fg_rng_block(rng, ra, rb);
            const int tg = (int)fg_pick(ra, (uint32_t)P.S);
            const int ts = P.site_slot[tg], tvv = P.site_vtype[tg];               // per-lane gathers of small tables
            int cK = 0, prop = 0;
            if (has_cat) {
                const int cb = P.site_cat[2 * tg];
                cK = P.site_cat[2 * tg + 1];
                rng.c1 = 1; fg_rng_block(rng, ra, rb);
                const double u1 = fg_u01_of(ra);
                if (tvv == 3 && cK > 0) {                              // first index whose cumulative probability reaches u, clamped (distribution.rs:771-784)
                    double cum = 0.0; int idx = cK;
                    for (int i0 = 0; i0 < cK; i0 += 4) {              // four table entries in flight
                        double pv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) pv[q] = (i0 + q < cK) ? pool_rd(cb + i0 + q) : 0.0;
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (i0 + q < cK) { cum += pv[q]; if (idx == cK && !(cum < u1)) idx = i0 + q; }
                    }
                    prop = idx < cK - 1 ? idx : cK - 1;
                    b[3 * tw] = pool_rd(cb + cK + prop);              // the table's precomputed ln p (-inf for p <= 0)
                } else cK = 0;
            }
            b[0] = fg_as_double((long long)(uint32_t)tg | ((long long)(uint32_t)ts << 16) | ((long long)tvv << 32) | ((long long)cK << 35) | ((long long)prop << 42));
        } else if (part == 1) {
            rng.c1 = 1; fg_rng_block(rng, ra, rb);
            b[tw] = fg_cold_gaussian_z(ra, rb);
        } else {
            if (has_bool) { rng.c1 = 1; fg_rng_block(rng, ra, rb); b[(NR - 1) * tw] = fg_cold_lnu(ra); }
            rng.c1 = 2; fg_rng_block(rng, ra, rb);
            b[2 * tw] = fg_cold_lnu(ra);                              // (NaN for u = 0: the decider evaluates the reference's expression)
        }
    };
#define FG_MH2_TARGET(m) ((int)((m) & 0xffff))
#define FG_MH2_TSLOT(m) ((int)(((m) >> 16) & 0xffff))
#define FG_MH2_TV(m) ((uint32_t)(((m) >> 32) & 7))
#define FG_MH2_CATK(m) ((int)(((m) >> 35) & 127))
#define FG_MH2_CATPROP(m) ((long long)(((m) >> 42) & 127))

    // ---- per-wave state.  A wave is the decider OR the proposer, never both, so the two roles' variables share registers (the kernel
    // runs four waves per SIMD at 128 VGPRs: two sets side by side spilled).  st[] / va..vd hold
    //   decider:  lw, the replaced cell, log q(x'|x), log q(x|x'), ln u of the accept uniform, the proposed value | slot, block of the accept uniform, nbad (this
    //             chain's row-less Categorical sites whose index is out of range), cur_bad (the current proposal replaces one) | g, nacc
    //   proposer: two steps are in flight -- the OLDER one (o_*: proposed, selected, both outcomes of its adaptation update formed -- u_* --
    //             and waiting for the decision that says which to store) and the NEWER one (q_*, locals of a step: both candidates formed,
    //             waiting for the decision about the older step that selects one): o_prop, o_cur (the older step's proposed value and the
    //             cell it replaces), u_sc_a, u_sc_r, u_ls_a, u_ls_r ({scale, log_scale} after an accepted / a rejected step) | va, vb =
    //             {scale, kind} and {log_scale, total, accepted} of the next target, on their way | o_kind0, o_kind_new, u_tot, u_acn_r |
    //             o_g, flags
    double st[12] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // (6 .. 11: the decider's copy of the candidate rows while it adds)
    fg_u32x4 va = {0u, 0u, 0u, 0u}, vb = {0u, 0u, 0u, 0u}, vc = {0u, 0u, 0u, 0u}, vd = {0u, 0u, 0u, 0u};
    double &lw = st[0], &old_cell = st[1], &lqf = st[2], &lqr = st[3], &lnu1 = st[4] /* ln u of the pending step's accept uniform */, &lnu2 = st[5] /* its proposed value */;
#define tslot_u va[0]
#define nb_u va[1]
#define nbad_u va[2]
#define cur_bad_u va[3]
#define FG_MH2_G() ((long long)(((unsigned long long)vb[1] << 32) | (unsigned long long)vb[0]))
#define FG_MH2_SET_G(x) { const unsigned long long g_ = (unsigned long long)(x); vb[0] = (uint32_t)g_; vb[1] = (uint32_t)(g_ >> 32); }
#define FG_MH2_NACC_INC() { const unsigned long long n_ = (((unsigned long long)vb[3] << 32) | (unsigned long long)vb[2]) + 1ull; vb[2] = (uint32_t)n_; vb[3] = (uint32_t)(n_ >> 32); }
    double &o_prop = st[0], &o_cur = st[1], &u_sc_a = st[2], &u_sc_r = st[3], &u_ls_a = st[4], &u_ls_r = st[5];
    fg_u32x4 &a0n = va, &a1n = vb;
#define o_kind0_u vc[0]
#define o_kind_new_u vc[1]
#define u_tot vc[2]
#define u_acn_r vc[3]
#define o_flags vd[2]                                               /* bit 0 o_have, bit 1 o_adapt */
#define q_kind_new_u vd[3]                                          /* the newer step's kind after its proposal (everything else of it is re-read from the rows in phase B) */
#define o_target_u vd[0]                                            /* the older step's target site */
    if (wv == w_pro) { u_sc_a = 1.0; u_sc_r = 1.0; }
    bool cb_zero = false;                                          // the proposer: the log q rows hold 0.0 (the common Gaussian step leaves them alone)

    if (wv == 0) {
        fg_load_values(P, X, c, slots, tw);
        lw = M.lw[c];
        for (int j = 0; j < n_cu; ++j) {
            const FgMhCatU cu = seg.catu[j];
            const long long zi = fg_as_i64(slots[cu.slot * tw]);
            nbad_u += (zi < 0 || zi >= (long long)cu.K) ? 1u : 0u;
        }
        CB[7 * tw] = 0.0; DR[0] = 0.0;                             // no tag yet
        if (SPLIT) LK[tw] = 0.0;
        // the decider's instruction stream is the path of its tile: it is served before the other waves of the tiles it shares a SIMD
        // with (exp_mask bit 32 switches this off: A/B)
        if (!(exp_mask & 32)) __builtin_amdgcn_s_setprio(2);
    }
    for (int k = (int)threadIdx.x; k < pool_n; k += (int)blockDim.x) pool_l[k] = P.pool[k];
    if (pool_n > 0) __syncthreads();                               // the random-number waves read the staged tables below
    if (wv == rng_wave) { publish_rng(iter0, 0); if (n_steps > 1) publish_rng(iter0 + 1, 0); }
    if (wv == rng_wave1) { publish_rng(iter0, 1); if (n_steps > 1) publish_rng(iter0 + 1, 1); }
    if (wv == rng_wave2) { publish_rng(iter0, 2); if (n_steps > 1) publish_rng(iter0 + 1, 2); }
    int sa_[FG_MH_NCLS], sb_[FG_MH_NCLS];
#pragma unroll
    for (int q = 0; q < FG_MH_NCLS; ++q) { sa_[q] = RK == 0 ? 0 : seg.r[q][wv]; sb_[q] = RK == 0 ? 0 : seg.r[q][wv + 1]; }
#define sa(q) (RK == 0 ? seg.r[q][wv] : sa_[q])
#define sb(q) (RK == 0 ? seg.r[q][wv + 1] : sb_[q])
    __syncthreads();
    // the adaptation state of a step's target, requested a phase ahead (the proposer's loads and stores of it are one wave's program order)
    auto prefetch_ad = [&](int it) __attribute__((always_inline)) {
        const long long m0 = fg_as_i64(xch[(long long)(NR * (it & 1)) * tw]);
        const long long g0 = (long long)FG_MH2_TARGET(m0) * X.C + c;
        a0n = *(const fg_u32x4 *)(M.ad + g0);
        if (it < n_warmup) a1n = *(const fg_u32x4 *)((const char *)(M.ad + g0) + 16);
    };
#ifdef FG_MH_PROF
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = 0;
#endif
    // Both candidates of step `it` (the proposer).  Speculative form (after_decision = false; phase B of the step before, or the
    // launch's first step): everything it reads is settled -- the random numbers of step `it`, the older step's selected proposal and
    // both outcomes of its adaptation update, the cells the older step did not move.  Returns false, writing nothing, when a lane's kind
    // is undecided: f64_kind (mh.rs:339-358) asks about the chain's state AFTER the older step, so that step's candidates are formed in
    // phase A behind the decision (after_decision = true; transient: a kind is decided once per (site, chain)).
    auto candidates = [&](int it, auto mode_) __attribute__((always_inline)) -> bool {
        constexpr bool after_decision = decltype(mode_)::after;
        const double *b = xch + (long long)(NR * (it & 1)) * tw;
        const uint32_t tag = FG_MH2_TAG(it);
        const long long m_t = fg_as_i64(b[0]);
        const double z = b[tw];
        const int target = FG_MH2_TARGET(m_t), ts = FG_MH2_TSLOT(m_t);
        const bool same = (o_flags & 1u) != 0u && (uint32_t)target == o_target_u;
        const fg_u32x4 a0 = a0n;
        int kind_mem = same ? (int)o_kind_new_u : (int)a0[2];
        const double sc_mem = fg_dbl(a0[0], a0[1]);
        if (!after_decision && (exp_mask & 2048) && __all(kind_mem == FG_PROP_GAUSSIAN)) {
            // the common step of a program of f64 sites without overrides, every lane's kind decided Gaussian (mh.rs:183-187):
            // x' = x + scale z for both outcomes of the older step, log q = 0 both ways, the accept uniform in block 2
            const double v = slots[ts * tw];                               // a cell the older step's decision does not touch -- except the one that
            const double cur_a = same ? o_prop : v, cur_r = same ? o_cur : v;   // step moved, which this wave knows itself
            const double sc_A = same ? u_sc_a : sc_mem, sc_R = same ? u_sc_r : sc_mem;
            CB[0] = cur_a + sc_A * z; CB[tw] = cur_r + sc_R * z; CB[6 * tw] = v;
            if (!cb_zero) { CB[2 * tw] = 0.0; CB[3 * tw] = 0.0; CB[4 * tw] = 0.0; CB[5 * tw] = 0.0; cb_zero = true; }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            vwrite(CB + 7 * tw, fg_as_double((long long)tag | (0x22LL << 32) | ((long long)ts << 40) | ((long long)(same ? 3 : 0) << 56)));
            q_kind_new_u = (uint32_t)kind_mem;
#ifdef FG_MH_PROF
            prof_[7] += 1000000ull;
#endif
            return true;
        }
#ifdef FG_MH_PROF
        prof_[7] += (exp_mask & 2048) ? 1ull : 1000ull;
#endif
        const uint32_t tv = FG_MH2_TV(m_t);
        int kind_eff = FG_PROP_AUTO;
        if (tv == 0u) { kind_eff = M.ov_kind ? M.ov_kind[ts] : FG_PROP_AUTO; if (kind_eff == FG_PROP_AUTO) kind_eff = kind_mem; }
        const bool undecided = tv == 0u && kind_eff == FG_PROP_AUTO;
        const bool slow = __any(undecided);
        if (slow && !after_decision) return false;
        cb_zero = false;
        double cur_a, cur_r, sc_A, sc_R, v_cell;
        if constexpr (after_decision) {
            // the older step is decided and committed: one candidate from the chain's state as it is.  One probe per distinct undecided site in the wave.
            bool acc_o = false;
            for (;;) {
                const long long d_ = fg_as_i64(vread(DR));
                if (__all((uint32_t)d_ == tag)) { acc_o = ((d_ >> 32) & 1) != 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            cur_a = cur_r = v_cell = vread(slots + ts * tw);
            sc_A = sc_R = same ? (acc_o ? u_sc_a : u_sc_r) : sc_mem;
            unsigned long long todo = __ballot(undecided);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int tl = __builtin_amdgcn_readlane(target, leader);
                const unsigned long long sm = __ballot(undecided && target == tl);
                const fg_u32x16 r = fg_fetch_grec(P.sstream, P.site_rec[tl]);
                FgAcc3 dummy = {0.0, 0.0, 0.0};
                const double probe = fg_score_one<RK>(r, -1.0, vread(slots + r[1] * tw), P.pool, slots, tw, dummy);
                if (undecided && target == tl) { kind_eff = (cur_a > 0.0 && !fg_finite(probe)) ? FG_PROP_LOGSPACE : FG_PROP_GAUSSIAN; kind_mem = kind_eff; }
                todo &= ~sm;
            }
        } else {
            const double v = slots[ts * tw];
            cur_a = same ? o_prop : v; cur_r = same ? o_cur : v;
            sc_A = same ? u_sc_a : sc_mem; sc_R = same ? u_sc_r : sc_mem;
            v_cell = v;
        }
        double lo = 0.0, hi = 0.0;
        if (tv == 0u && kind_eff != FG_PROP_GAUSSIAN && kind_eff != FG_PROP_LOGSPACE && M.ov_lo) { lo = M.ov_lo[ts]; hi = M.ov_hi[ts]; }
        FgMhCand ka, kr;
        if (tv == 3u) {                                                    // usize target: the index resampled from the constant prior table (mh.rs:516-530)
            const int cat_K = FG_MH2_CATK(m_t), cat_base = P.site_cat[2 * target];
            const double lp_new = has_cat ? b[3 * tw] : 0.0;
            const long long ia = fg_as_i64(cur_a), ir = fg_as_i64(cur_r);
            ka.prop = kr.prop = fg_as_double(FG_MH2_CATPROP(m_t));
            ka.lqf = kr.lqf = 0.0 + lp_new;                                // prior log-probabilities of the proposed and the current index
            ka.lqr = 0.0 + ((ia < 0 || ia >= (long long)cat_K) ? FG_NEG_INF : pool_rd(cat_base + cat_K + (int)ia));
            kr.lqr = 0.0 + ((ir < 0 || ir >= (long long)cat_K) ? FG_NEG_INF : pool_rd(cat_base + cat_K + (int)ir));
            ka.nb = kr.nb = 2;
        } else if (__all(tv == 0u && kind_eff == FG_PROP_GAUSSIAN)) {      // GaussianWalkProposal (mh.rs:183-187), both candidates; nothing to call
            ka.prop = cur_a + sc_A * z; ka.lqf = 0.0 + 0.0; ka.lqr = 0.0 + 0.0; ka.nb = 2;
            kr.prop = cur_r + sc_R * z; kr.lqf = ka.lqf; kr.lqr = ka.lqr; kr.nb = 2;
        } else {
            ka = fg_cold_walk_pure(tv, kind_eff, cur_a, sc_A, z, lo, hi);
            kr = ka;
            if (__any(same && !after_decision)) kr = fg_cold_walk_pure(tv, kind_eff, cur_r, sc_R, z, lo, hi);
        }
        CB[0] = ka.prop; CB[tw] = kr.prop; CB[2 * tw] = ka.lqf; CB[3 * tw] = kr.lqf; CB[4 * tw] = ka.lqr; CB[5 * tw] = kr.lqr; CB[6 * tw] = v_cell;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        vwrite(CB + 7 * tw, fg_as_double((long long)tag | ((long long)ka.nb << 32) | ((long long)kr.nb << 36) | ((long long)ts << 40) | ((long long)((same && !after_decision) ? 1 : 0) << 56) |
                                          ((long long)(same ? 1 : 0) << 57) | ((long long)(after_decision ? 1 : 0) << 58)));
        q_kind_new_u = (uint32_t)kind_mem;
        return true;
    };
    bool deferred = false;                                         // the proposer: the next step's candidates wait for the decision
    if (wv == w_pro) { prefetch_ad(iter0); deferred = true; }      // (the launch's first step: formed in phase A, behind the "decision" about no step)
#ifdef FG_MH_PROF
    tprev_ = __builtin_readcyclecounter();
#endif
    if (exp_mask & 384) {                                          // tiles that start together on a CU run their serial phases at the same time: started a part of a step apart they fill each other's gaps
        const unsigned ph = (exp_mask & 128) ? (blockIdx.x & 3u) : ((blockIdx.x * 4u / gridDim.x) & 3u);
        for (unsigned q = 0; q < ph; ++q) __builtin_amdgcn_s_sleep(27);
    }
    for (int t = 0; t <= n_steps; ++t) {
        const int iter = iter0 + t;
        const uint32_t tag = FG_MH2_TAG(iter);
        const double *b = xch + (long long)(NR * (iter & 1)) * tw;
        const bool do_prop = t < n_steps;
        // ================================================================ phase A
        // ---- SPLIT: the log_likelihood terms of step t - 1 on a second adding wave
        if (SPLIT && wv == w_sum && t > 0) {
            if (!(exp_mask & 32)) __builtin_amdgcn_s_setprio(2);
            vwrite(LK, fg_inorder_sum1(terms + (long long)n_pri * tw, n_lik, tw));
            __builtin_amdgcn_s_waitcnt(0xc07f);
            vwrite(LK + tw, fg_as_double((long long)tag));
            if (!(exp_mask & 32)) __builtin_amdgcn_s_setprio(0);
        }
        // ---- the decider.  Its instruction count is its tile's path (a lone wave issues one instruction per ~8 cycles): everything it
        // needs for the selection is requested before / between the sums, and what follows the decision is selects and two LDS writes.
        if (wv == 0) {
            long long m_t = 0;
            double n_lnu1 = NAN, n_lnu2 = NAN;
            if (do_prop) {
                m_t = fg_as_i64(b[0]);
                n_lnu2 = b[2 * tw];
                if (has_bool) n_lnu1 = b[(NR - 1) * tw];
            }
            bool accept = false;
            double pri = 0.0, lik = 0.0;
            // the candidates of step t: tag first, rows behind it (requested between the chunks and the tails of the sums: the proposer is
            // normally done by then; checked after the sums, re-read while the tag is stale)
            double cm_ = 0.0, c_cur = 0.0;
            double &c_pa = st[6], &c_pr = st[7], &c_fa = st[8], &c_fr = st[9], &c_ra = st[10], &c_rr = st[11];
#define FG_MH2_CAND_BATCH { cm_ = vread(CB + 7 * tw); c_pa = vread(CB); c_pr = vread(CB + tw); c_fa = vread(CB + 2 * tw); c_fr = vread(CB + 3 * tw); \
                            c_ra = vread(CB + 4 * tw); c_rr = vread(CB + 5 * tw); c_cur = vread(CB + 6 * tw); }
            if (do_prop) FG_MH2_CAND_BATCH
            if (t > 0) {                                                   // finish step t - 1
                // the row-less tail of log_prior (FgMhSeg): the constants, eight per scalar load; from the cells while a chain holds a bad index
#define FG_MH2_CATU_TAIL                                                                                                    \
                if (n_cu > 0) {                                                                                             \
                    if (__builtin_expect(__any(nbad_u != 0u), 0)) {                                                         \
                        for (int j = 0; j < n_cu; ++j) {                                                                    \
                            const FgMhCatU cu = seg.catu[j];                                                                \
                            const long long zi = fg_as_i64(slots[cu.slot * tw]);                                            \
                            pri += (zi < 0 || zi >= (long long)cu.K) ? FG_NEG_INF : fg_uniform(seg.catu_c[j]);              \
                        }                                                                                                   \
                    } else if (seg.catu_same) {                     /* one constant for all of them (equal tables): no loads */    \
                        const double c0_ = fg_uniform(seg.catu_c0);                                                         \
                        int j = 0;                                                                                          \
                        for (; j + 8 <= n_cu; j += 8) { _Pragma("unroll") for (int q = 0; q < 8; ++q) pri += c0_; }          \
                        for (; j < n_cu; ++j) pri += c0_;                                                                   \
                    } else {                                        /* eight per scalar load, the next eight on their way */  \
                        fg_u32x16 cb_ = fg_fetch_grec((const FgGradRec *)seg.catu_c, 0);                                    \
                        for (int j = 0; j < n_cu; j += 8) {                                                                 \
                            const fg_u32x16 cn_ = fg_fetch_grec((const FgGradRec *)seg.catu_c, (j >> 3) + 1);               \
                            _Pragma("unroll") for (int q = 0; q < 8; ++q) if (j + q < n_cu) pri += fg_dbl(cb_[2 * q], cb_[2 * q + 1]); \
                            cb_ = cn_;                                                                                      \
                        }                                                                                                   \
                    }                                                                                                       \
                }
                if (SPLIT) {
                    pri = fg_inorder_sum1(terms, n_pri, tw);
                    FG_MH2_CATU_TAIL
                    for (;;) {                                             // the second adding wave's sum: tag first, value behind it
                        const double tg_ = vread(LK + tw), v_ = vread(LK);
                        if (__all((uint32_t)fg_as_i64(tg_) == tag)) { lik = v_; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                } else {
                    fg_inorder_sums2(terms, n_pri, terms + (long long)n_pri * tw, n_lik, tw, pri, lik);
                    FG_MH2_CATU_TAIL
                }
#undef FG_MH2_CATU_TAIL
                FG_PROF_T(0)
                const double prop_lw = pri + lik + 0.0;                    // total_log_weight (no factor statement has a record)  trace.rs:168-177
                const double log_alpha = prop_lw - lw + (lqr - lqf);       // + dim_term == 0 (fixed structure)  mh.rs:731-732
                // mh.rs:733  log_alpha >= 0 || u < exp(log_alpha), decided from ln u outside a margin far above the transcendentals' errors
                // (lnu1: this step's ln u -- the block was chosen when the step was selected)
                const double mrg = 1e-9 * (1.0 + fabs(log_alpha));
                const bool sure_acc = log_alpha >= 0.0 || lnu1 < log_alpha - mrg;
                const bool sure_rej = lnu1 > log_alpha + mrg;
                accept = sure_acc;
                if (__builtin_expect(__any(!sure_acc && !sure_rej), 0)) {  // inside the margin, u = 0, NaN, an unexpected block: the reference's expression itself
                    const double u_ = fg_cold_u01_pair(sk0, sk1, gchain, nb_u, (uint32_t)(iter - 1), FG_RNG_MH).a;
                    const bool exact = (log_alpha >= 0.0) || (u_ < fg_cold_exp(log_alpha));
                    if (!sure_acc && !sure_rej) accept = exact;
                }
                if (accept) { lw = prop_lw; FG_MH2_NACC_INC() if (live) X.values[FG_MH2_G()] = fg_as_i64(lnu2 /* the proposed value */); if (cur_bad_u) nbad_u -= 1u; }
                else slots[tslot_u * tw] = old_cell;
                if (__builtin_expect(((iter - 1) >= n_warmup || M.rec_all) && draws != nullptr, 0)) {   // recorded cells of the CURRENT state (mh.rs:1010)
                    if (live) {
                        long long *row = draws + (long long)(t - 1 - first_sample_t) * M.n_rec * X.C + c;
                        for (int r = 0; r < M.n_rec; ++r) row[(long long)r * X.C] = fg_as_i64(slots[M.rec[r] * tw]);
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                            // the commit / roll-back is in LDS before the decision is announced
            vwrite(DR, fg_as_double((long long)tag | ((long long)(accept ? 1 : 0) << 32)));
            FG_PROF_T(1)
            if (do_prop) {                                                 // select the proposal of step t
                while (!__all((uint32_t)fg_as_i64(cm_) == tag)) { __builtin_amdgcn_s_sleep(1); FG_MH2_CAND_BATCH }
#undef FG_MH2_CAND_BATCH
                const long long cm = fg_as_i64(cm_);
                const bool same_ = ((cm >> 56) & 1) != 0;                  // step t's target is the site step t - 1 moved: its cell is the proposed value or the restored one
                const double prev_prop = lnu2, prev_old = old_cell;
                const double cur = same_ ? (accept ? prev_prop : prev_old) : c_cur;
                const uint32_t nb_ = (uint32_t)((cm >> (accept ? 32 : 36)) & 15);
                tslot_u = (uint32_t)((cm >> 40) & 0xffff);
                nb_u = nb_;
                FG_MH2_SET_G((long long)FG_MH2_TARGET(m_t) * X.C + c)
                old_cell = cur;
                lqf = accept ? c_fa : c_fr; lqr = accept ? c_ra : c_rr;
                lnu1 = nb_ == 2u ? n_lnu2 : (nb_ == 1u ? n_lnu1 : NAN);    // ln of the accept uniform of the block the proposal left
                lnu2 = accept ? c_pa : c_pr;                               // (st[5] from here on: the proposed value, for the commit)
                slots[tslot_u * tw] = lnu2;
                if (n_cu > 0) { const long long ci = fg_as_i64(cur); const int cK = FG_MH2_CATK(m_t); cur_bad_u = (FG_MH2_TV(m_t) == 3u && cK > 0 && (ci < 0 || ci >= (long long)cK)) ? 1u : 0u; }
            }
            FG_PROF_T(2)
        }
        // ---- the proposer in phase A: only a step whose candidates had to wait for the decision (an undecided kind)
        if (wv == w_pro && do_prop && deferred) {
            if (!(exp_mask & 32)) __builtin_amdgcn_s_setprio(1);
            (void)candidates(iter, FgMh2After());
            deferred = false;
            if (!(exp_mask & 32)) __builtin_amdgcn_s_setprio(0);
            FG_PROF_T(2)
        }
        // ---- random numbers of step t + 1 (buffer (iter + 1) & 1 was last read in phase A of step t - 1)
        if (t > 0 && t + 1 < n_steps) {
            if (wv == rng_wave) publish_rng(iter + 1, 0);
            if (wv == rng_wave1) publish_rng(iter + 1, 1);
            if (wv == rng_wave2) publish_rng(iter + 1, 2);
        }
        // ---- the proposer, once the decision about step t - 1 is out (phase B; in the last pass: here, behind the tag): store the chosen
        // outcome of that step's update, select step t's candidate, form both outcomes of step t's update, request step t + 1's state
        // (the older step's stores come first: the request for step t + 1's state follows them in this wave's program order)
#define FG_MH2_BOOK_STORE(ACC_O)                                                                                            \
            const bool acc_o_ = (ACC_O);                                                                                    \
            const bool o_adapt_ = (o_flags & 2u) != 0u;                                                                     \
            double f_ls = 0.0; uint32_t f_tot = 0u, f_acn = 0u;                                                             \
            if (o_flags & 1u) {                                                                                             \
                const long long o_g_ = (long long)o_target_u * X.C + c;                                                     \
                if (o_adapt_) {                                    /* DiminishingAdaptation::update  mcmc_utils.rs:88-150 */  \
                    f_ls = acc_o_ ? u_ls_a : u_ls_r; f_tot = u_tot; f_acn = u_acn_r + (acc_o_ ? 1u : 0u);                    \
                    if (live) {                                                                                             \
                        const unsigned long long lb = (unsigned long long)__double_as_longlong(f_ls);                       \
                        const fg_u32x4 w1 = { (uint32_t)lb, (uint32_t)(lb >> 32), f_tot, f_acn };                           \
                        *(fg_u32x4 *)((char *)(M.ad + o_g_) + 16) = w1;                                                     \
                        M.ad[o_g_].scale = acc_o_ ? u_sc_a : u_sc_r;                                                        \
                    }                                                                                                       \
                }                                                                                                           \
                if (live && o_kind_new_u != o_kind0_u) M.ad[o_g_].kind = (int)o_kind_new_u;                                 \
            }
#define FG_MH2_BOOK_NEWER()                                                                                                 \
            {                                                                                                               \
                /* the newer step, from its rows (intact until the next phase A) and the state requested for it */           \
                const long long qm_ = fg_as_i64(b[0]), cmq_ = fg_as_i64(CB[7 * tw]);                                        \
                const double q_pa_ = CB[0], q_pr_ = CB[tw], q_v_ = CB[6 * tw];                                              \
                const bool q_same_ = ((cmq_ >> 57) & 1) != 0, q_slow_ = ((cmq_ >> 58) & 1) != 0;                            \
                const bool q_adapt_ = iter < n_warmup;                                                                      \
                const double sc_mem_ = fg_dbl(a0n[0], a0n[1]);                                                              \
                /* (a step proposed after the decision -- slow -- used the committed cell and the chosen scale: q_v_, and the same selects) */ \
                const double n_scale = q_same_ ? (acc_o_ ? u_sc_a : u_sc_r) : sc_mem_;                                      \
                const double n_cur = (q_same_ && !q_slow_) ? (acc_o_ ? o_prop : o_cur) : q_v_;                              \
                const uint32_t n_kind0 = q_same_ ? o_kind_new_u : a0n[2];                                                   \
                const fg_u32x4 q_a1_ = a1n;                                                                                 \
                u_sc_a = u_sc_r = n_scale;                                                                                  \
                if (q_adapt_) {                                                                                             \
                    const bool fwd = q_same_ && o_adapt_;          /* the record this wave has just written */                \
                    const double ls0 = fwd ? f_ls : fg_dbl(q_a1_[0], q_a1_[1]);                                             \
                    u_tot = (fwd ? f_tot : q_a1_[2]) + 1u; u_acn_r = fwd ? f_acn : q_a1_[3];                                 \
                    u_ls_a = u_ls_r = ls0;                                                                                  \
                    if (u_tot >= 10u) {                                                                                     \
                        const FgD2 ra_ = fg_cold_mh_adapt(ls0, u_acn_r + 1u, u_tot, M.step_tab, M.step_n);                   \
                        const FgD2 rr_ = fg_cold_mh_adapt(ls0, u_acn_r, u_tot, M.step_tab, M.step_n);                        \
                        u_sc_a = ra_.a; u_ls_a = ra_.b; u_sc_r = rr_.a; u_ls_r = rr_.b;                                      \
                    }                                                                                                       \
                }                                                                                                           \
                o_flags = 1u | (q_adapt_ ? 2u : 0u); o_target_u = (uint32_t)FG_MH2_TARGET(qm_);                             \
                o_kind0_u = n_kind0; o_kind_new_u = q_kind_new_u;                                                           \
                o_prop = acc_o_ ? q_pa_ : q_pr_; o_cur = n_cur;                                                             \
            }
        if (t == n_steps) {
            if (wv == w_pro) {
                bool acc_o = false;
                for (;;) {
                    const long long d_ = fg_as_i64(vread(DR));
                    if (__all((uint32_t)d_ == tag)) { acc_o = ((d_ >> 32) & 1) != 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                FG_MH2_BOOK_STORE(acc_o)
                (void)f_ls; (void)f_tot; (void)f_acn;
            }
            break;
        }
        FG_PROF_T(3)
        __builtin_amdgcn_s_waitcnt(0xc07f);                                // LDS only crosses this barrier: the proposer's requests of the adaptation state stay in flight
        __builtin_amdgcn_s_barrier();                                      // the proposal is in the tile; random numbers of step t + 1 published
        FG_PROF_T(4)
        // ================================================================ phase B: every wave scores its share of the statements
        if (wv == w_pro) {
            // the decision about step t - 1: store the chosen outcome of its update; request the adaptation state of step t + 1's target;
            // select step t's candidate and form both outcomes of ITS update; then both candidates of step t + 1
            FG_MH2_BOOK_STORE((fg_as_i64(vread(DR)) >> 32) & 1)
            fg_u32x4 a0x = {0u, 0u, 0u, 0u}, a1x = {0u, 0u, 0u, 0u};
            if (t + 1 < n_steps) {
                const long long m1 = fg_as_i64(xch[(long long)(NR * ((iter + 1) & 1)) * tw]);
                const long long g1 = (long long)FG_MH2_TARGET(m1) * X.C + c;
                a0x = *(const fg_u32x4 *)(M.ad + g1);
                if (iter + 1 < n_warmup) a1x = *(const fg_u32x4 *)((const char *)(M.ad + g1) + 16);
            }
            FG_MH2_BOOK_NEWER()
            if (t + 1 < n_steps) { a0n = a0x; a1n = a1x; deferred = !candidates(iter + 1, FgMh2Spec()); }
        }
#undef FG_MH2_BOOK_STORE
#undef FG_MH2_BOOK_NEWER
        if (b_prio) __builtin_amdgcn_s_setprio(1);
#ifndef FG_MHMW_ALL          /* (a unit compiled at run time that generates every statement has no record runs) */
        if (RK >= 2) {
            if (pool_n > 0) {
                for (int k = sa(0); k < sb(0); k += 4) fg_mh_group4<0>(srt, k, sb(0), pool_l, slots, tw, terms);
                for (int k = sa(1); k < sb(1); k += 4) fg_mh_group4<1>(srt, k, sb(1), pool_l, slots, tw, terms);
            } else {
                for (int k = sa(0); k < sb(0); k += 4) fg_mh_group4<0>(srt, k, sb(0), P.pool, slots, tw, terms);
                for (int k = sa(1); k < sb(1); k += 4) fg_mh_group4<1>(srt, k, sb(1), P.pool, slots, tw, terms);
            }
        }
        fg_mh_terms<RK, 1>(srt, sa(2), sb(2), P.pool, nullptr, slots, tw, terms);
        fg_mh_terms<RK, 2>(srt, sa(3), sb(3), P.pool, nullptr, slots, tw, terms);
        fg_mh_terms<RK, 3>(srt, sa(4), sb(4), P.pool, nullptr, slots, tw, terms);
#endif
#ifdef FG_MHMW_PHASE_B5     /* a unit compiled at run time (fg_jit.cpp): the general records as generated straight-line code, a share of the segments per wave */
        FG_MHMW_PHASE_B5();
#else
        if (RK != 0 && pool_n > 0) fg_mh_terms<RK>(srt, sa(5), sb(5), P.pool, pool_l, slots, tw, terms);
        else fg_mh_terms<RK>(srt, sa(5), sb(5), P.pool, nullptr, slots, tw, terms);
#endif
        FG_PROF_T(5)
        if (b_prio) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        FG_PROF_T(6)
    }
#ifdef FG_MH_PROF
    if (blockIdx.x == 0 && lane == 0) for (int q = 0; q < 8; ++q) fg_mh_prof[wv][q] = prof_[q];
#endif
    if (wv == 0 && live) { M.lw[c] = lw; M.n_acc[c] += ((unsigned long long)vb[3] << 32) | (unsigned long long)vb[2]; }
#undef sa
#undef sb
#undef FG_MH2_TARGET
#undef FG_MH2_TSLOT
#undef FG_MH2_TV
#undef FG_MH2_CATK
#undef FG_MH2_CATPROP
#undef FG_MH2_G
#undef FG_MH2_SET_G
#undef FG_MH2_NACC_INC
#undef o_target_u
#undef tslot_u
#undef nb_u
#undef nbad_u
#undef cur_bad_u
#undef o_kind0_u
#undef o_kind_new_u
#undef u_tot
#undef u_acn_r
#undef o_flags
#undef q_kind_new_u
}
