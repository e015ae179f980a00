// fg_mh_mw_body.h -- the step loop of the multi-wave MH kernel for score-stream programs (design: fg_mh.hip).  Compiled into the library
// (k_mh_mw_steps: phase B interprets the kind-sorted record stream) and, as text, by the run-time compiler (fg_jit.cpp:
// k_mh_mw_jit_steps -- FG_MHMW_PHASE_B5 runs this wave's share of the general records as generated code; everything else is this file).
#define FG_MH_WMAX 16

// FG_MH_PROF (experiment builds, tools/prof_mh_phases.py): cycles tile 0's waves spend in each part of a step
#ifdef FG_MH_PROF
__device__ unsigned long long fg_mh_prof[FG_MH_WMAX][8];
#define FG_PROF_T(i) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_[i] += now_ - tprev_; tprev_ = now_; }
#else
#define FG_PROF_T(i)
#endif

// records [r0, r1) of the score stream at the current tile state -> term rows (row = the record's `coord` field).  Records
// are fetched two ahead into three rotating 16-SGPR buffers (the loop is unrolled by three so that no buffer is ever
// copied), operands one ahead; s_waitcnt by hand (SMEM returns out of order).
#ifdef FG_EXP_MH_NOFETCH      /* timing experiments only (tools/exp_mh_terms.sh): results are wrong */
#define FG_MH_FETCH(RC) RC = fg_fetch_grec(g, r0 + ((k + 2) & 1));
#else
#define FG_MH_FETCH(RC) RC = fg_fetch_grec(g, k + 2);
#endif
// PAT = operand pattern of a run of plain Normal records with sigma = 2^k (the host sorts them together): 1 x and mu are sites,
// 2 x is a constant (an observation), 3 mu is a constant; 0 = any record (fg_score_one).  A pattern run reads only the
// operands that are sites and forms -0.5 z z - ln sigma as one fma with an exact product (the rounding of the reference's two
// operations, fg_hmc_sep.hip).  It leaves out the "z != z -> -inf" select of the scoring kernels: a NaN term (only inf - inf
// operands produce one) makes log_alpha NaN, a -inf term makes it -inf, and both reject (mh.rs:733: `log_alpha >= 0 || u <
// exp(log_alpha)` is false either way) -- the term rows are rewritten by the next step and never stored.  The same holds
// for z z in [2^1024, 2^1025), where the fused form is -inf and the reference's finite but below -2^1023.
#ifdef FG_EXP_MH_NOLDS
#define FG_MH_OPND(RB, XB, MB) XB = xa; MB = ma;
#else
#define FG_MH_OPND(RB, XB, MB) if (PAT != 2) XB = slots[RB[0] * tw]; if (PAT != 3) MB = slots[RB[1] * tw];
#endif
#define FG_MH_TSTAGE(RA, XA, MA, RB, XB, MB, RC)                                                  \
    __builtin_amdgcn_s_waitcnt(0xc07f);                                                           \
    FG_MH_FETCH(RC)                                                                               \
    FG_MH_OPND(RB, XB, MB)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (PAT == 0) { FgAcc3 dummy = {0.0, 0.0, 0.0}; terms[RA[3] * tw] = fg_score_one<RK>(RA, XA, MA, pool, slots, tw, dummy, lane_pool); } \
    else {                                                                                        \
        const double x_ = PAT == 2 ? fg_dbl(RA[4], RA[5]) : XA, m_ = PAT == 3 ? fg_dbl(RA[6], RA[7]) : MA; \
        const double z_ = (x_ - m_) * fg_dbl(RA[10], RA[11]);                                     \
        terms[RA[3] * tw] = __builtin_fma(-0.5, z_ * z_, -fg_dbl(RA[12], RA[13])) - 0.5 * FG_LN_2PI; \
    }                                                                                             \
    if (++k >= r1) break;
template <int RK, int PAT = 0>
__device__ __forceinline__ void fg_mh_terms(const FgGradRec *g, int r0, int r1, const double *pool, const double *lane_pool, const double *slots, int tw, double *terms) {
    if (r0 >= r1) return;
    fg_u32x16 ra = fg_fetch_grec(g, r0), rb = fg_fetch_grec(g, r0 + 1), rc;
    double xa = 0.0, ma = 0.0, xb = 0.0, mb = 0.0, xc = 0.0, mc = 0.0;
    if (PAT != 2) xa = slots[ra[0] * tw];
    if (PAT != 3) ma = slots[ra[1] * tw];
    int k = r0;
    for (;;) {
        FG_MH_TSTAGE(ra, xa, ma, rb, xb, mb, rc)
        FG_MH_TSTAGE(rb, xb, mb, rc, xc, mc, ra)
        FG_MH_TSTAGE(rc, xc, mc, ra, xa, ma, rb)
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// Phase B works on a KIND-SORTED copy of the score stream: every record writes its own term row and the in-order sums are
// the control wave's, so the order in which the terms are evaluated is free.  Sorted by kind, four records of one kind are
// evaluated together in straight-line code -- their LDS round trips (operands; for an option select the option entry, then
// the slot it names; for a Categorical table the entry) overlap instead of following each other record by record, which is
// what a step of a mixture model waited for (C5: +15 %).
//   class 0  Normal(x; options[z], sigma = 2^k)     class 1  Categorical site with a constant table
//   classes 2, 3, 4  plain Normals with sigma = 2^k by operand pattern (site / site, constant x, constant mu) and
//   class 5  everything else: the pipelined one-at-a-time loop of fg_mh_terms (plain Normals measured faster there than four
//            at a time: the loop fetches records two ahead, a group of four waits for its own)
#define FG_MH_NCLS 6
// Categorical sites whose constant table is uniform (a mixture's assignments with equal weights): ln p[z] is the same number c for
// every index in range, so such a site's log_prior term needs neither a row nor a lookup.  When these terms are the LAST n_cu rows
// of log_prior, the kernel keeps no rows for them (the mixture of BASELINE's C5: 64 of 222 rows -- and with them the second tile a
// CU's LDS did not have room for) and the control wave adds the constants c[0 .. n_cu) after the rows, in the same order.  A site
// whose index is out of range (only ever an injected value: proposals come from the table) makes its term -inf: the control wave
// counts such sites per chain and, while any chain of the tile has one, forms the tail from the cells themselves.
struct FgMhCatU { int slot, K; };
struct FgMhSeg {
    int r[FG_MH_NCLS][FG_MH_WMAX + 1];   // records [r[c][w], r[c][w + 1]) of the sorted stream are wave w's share of class c
    int n_cu;                            // uniform-table Categorical terms at the end of log_prior that have no row (0: none)
    int catu_same; double catu_c0;       // every constant is catu_c0 (the sites share one table)
    const double *catu_c;                // [n_cu, padded with 0.0 to a multiple of 8 plus 8] their constants
    const FgMhCatU *catu;                // [n_cu] their sites' cells and table sizes
};

// Four records of class 0 (CLS = 0) or class 1 at the tile's state -> their term rows.  `tab`: the constant pool -- the LDS copy
// when the kernel staged it (the call sites pass the shared-memory pointer itself, so the lookups are ds_reads), else global.
//   class 0: x is the observation (the record's immediate), mu = the site named by option z of a list of sites; an index
//            outside the list makes mu NaN (fg_nsel_mu_lane) and the term NaN where the scoring kernels write -inf: both
//            reject (see FG_MH_TSTAGE) and the rows are rewritten by the next step;
//   class 1: ln p[z] from the table's precomputed logarithms, -inf outside it (distribution.rs:785-791).
template <int CLS>
__device__ __forceinline__ void fg_mh_group4(const FgGradRec *g, int k, int kend, const double *tab, const double *slots, int tw, double *terms) {
    fg_u32x16 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fg_fetch_grec(g, k + i < kend ? k + i : kend - 1);   // a short group repeats its last record (same term, same row)
    __builtin_amdgcn_sched_barrier(0);                         // the four scalar fetches are in flight together
    double zs[4], lp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) zs[i] = slots[r[i][CLS == 0 ? 1 : 0] * tw];                 // the index site
    __builtin_amdgcn_sched_barrier(0);                         // ... and so are the operand reads
    bool ok[4];
    if (CLS == 0) {
        uint32_t sl[4]; double v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long zi = (unsigned long long)fg_as_i64(zs[i]);
            ok[i] = zi < (unsigned long long)r[i][7];                                        // 0 <= z < K
            sl[i] = ((const uint32_t *)(tab + r[i][6]))[4 * (ok[i] ? (uint32_t)zi : 0u)];   // option entry {u32 slot, u32 is_const = 0, f64 unused}
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = slots[sl[i] * tw];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double m = ok[i] ? v[i] : NAN;
            const double z = (fg_dbl(r[i][4], r[i][5]) - m) * fg_dbl(r[i][10], r[i][11]);
            lp[i] = __builtin_fma(-0.5, z * z, -fg_dbl(r[i][12], r[i][13])) - 0.5 * FG_LN_2PI;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long zi = (unsigned long long)fg_as_i64(zs[i]);
            const uint32_t base = r[i][6], K = r[i][7];
            ok[i] = zi < (unsigned long long)K;
            lp[i] = tab[base + K + (ok[i] ? (uint32_t)zi : 0u)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) lp[i] = ok[i] ? lp[i] : FG_NEG_INF;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) terms[r[i][3] * tw] = lp[i];
}

#ifdef FG_MHMW_NS
// Programs without a score stream (a unit compiled at run time; `srt` is then the [S][2] table of each site's own statement in the
// generic program: first instruction, count).  The lanes whose proposal needs the model get it from that statement, interpreted in the
// propose-and-score mode (SingleSiteProposalHandler, mh.rs:298-570) ahead of the scoring run: the statement's parameters read only
// other sites, which hold the chain's current values, so the proposed value, log q(x'|x), log q(x|x'), the decided kind and the accept
// uniform's block are those of a whole propose-and-score run (fg_mh_interp_body.h does the same).  One pass per distinct such target
// in the wave; the other lanes see no target there.
struct FgMhmwPre { double lqf, lqr; int kind, next_block; };
static __device__ __noinline__ FgMhmwPre fg_mhmw_model_proposals(const FgIns *ins, const int *site_ins, const double *pool, double *slots_generic, bool live, bool walk, int target,
                                                                 FgMhCtx pre, unsigned long long seed, uint32_t gchain, uint32_t iter) {
    // the tile is LDS: through the parameter's generic pointer every access of this out-of-line function would be a FLAT one
    double *slots = (double *)(__attribute__((address_space(3))) double *)slots_generic;
    FgStream rng = fg_stream(seed, gchain, iter, FG_RNG_MH);
    unsigned long long ra, rb;
    fg_rng_block(rng, ra, rb);
    pre.rng = rng;                                            // at block 1
    if (walk) pre.target = -1;
    unsigned long long todo = __ballot(!walk);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int tl = __builtin_amdgcn_readlane(target, leader);
        const unsigned long long same = __ballot(!walk && target == tl);
        FgAcc3 A = {0.0, 0.0, 0.0};
#ifdef FG_MHMW_JIT_PROP      /* Categorical sites with a computed table: the statement as generated code (fg_jit.cpp) */
        if (!FG_MHMW_JIT_PROP(tl, slots, pre))
#endif
        fg_exec<FG_MODE_MH, false>(ins + site_ins[2 * tl], site_ins[2 * tl + 1], pool, slots, FG_WAVE, A, nullptr, nullptr, 0, live, &pre);
        todo &= ~same;
    }
    FgMhmwPre r = { pre.lqf, pre.lqr, pre.kind, pre.next_block };
    return r;
}
#endif

template <int RK, bool SPLIT /* the two in-order sums on two waves */>
__device__ __forceinline__ void fg_mh_mw_body(const FgProgramDev &P, const FgChainCtx &X, const FgMhDev &M, const FgGradRec *srt /* the kind-sorted score stream */, const FgMhSeg &seg, int iter0, int n_steps, int n_warmup,
                                              long long *draws, int first_sample_t, int exp_mask /* bits 1, 2, 4, 8: timing experiments only (FG_MH_EXP; results are wrong); 32: no wave priorities (A/B); 64: phase-B priority; 128 / 256: staggered start */,
                                              int pool_n /* > 0: the constant pool (pool_n doubles) is staged into LDS behind the exchange rows */) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef FG_MHMW_K_W        /* a unit compiled at run time for ONE launch shape (fg_jit.cpp): literals */
    constexpr int W = FG_MHMW_K_W, n_slots_ = FG_MHMW_K_NSLOTS;
    exp_mask = FG_MHMW_K_EXP; pool_n = FG_MHMW_K_POOLN;
#else
    const int W = (int)(blockDim.x >> 6), n_slots_ = P.n_slots;
#endif
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
#ifdef FG_MHMW_NS          /* a compiled unit of a program without a score stream: its statements, log_prior rows first */
    constexpr int n_s = FG_MHMW_NS, n_pri = FG_MHMW_NPRI, n_fac = FG_MHMW_NFAC /* `factor` statements: the last rows */, n_lik = n_s - n_pri - n_fac, n_cu = 0;
#elif defined(FG_MHMW_K_NS)
    constexpr int n_cu = FG_MHMW_K_NCU, n_s = FG_MHMW_K_NS, n_pri = FG_MHMW_K_NPRI, n_lik = n_s - n_pri;
#else
    const int n_cu = seg.n_cu;                                      // (terms without a row: FgMhSeg)
    const int n_s = P.n_sstream - n_cu, n_pri = P.n_prior_terms - n_cu, n_lik = n_s - n_pri;
#endif
    double *slots = lds + lane;
    double *terms = lds + (long long)n_slots_ * tw + lane;
    // exchange rows, double-buffered by step parity (8 rows each; row 16: the log_likelihood sum on its way to the control wave):
    // 0 target site, 1 gaussian_z, 2 u(block 1), 3 u(block 2),
    // 4 {LDS slot, value type} of the target, 5 Categorical targets: {pool base, K} of the constant table, 6 their proposed
    // index, 7 its prior log-probability
    double *xch = terms + (long long)n_s * tw;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    const bool b_prio = wv != 0 && (exp_mask & 64) != 0;
    const int rng_wave = W - 1;
    const int rng_wave1 = W >= 3 ? W - 2 : W - 1;                  // the wave of part 1
    const bool skip_u1 = (exp_mask & 4096) != 0;

    // small constant pools (Categorical tables, option lists) are read per lane: from the LDS copy a lookup costs an LDS round
    // trip instead of a vector-memory one
    double *pool_l = lds + (long long)(n_slots_ + n_s + 17) * tw;
    auto pool_rd = [&](int idx) __attribute__((always_inline)) { return pool_n > 0 ? pool_l[idx] : P.pool[idx]; };

    // Everything of step `it` that does not depend on the chain's state -> buffer (it & 1).
    //   part 0: gen_range target (mh.rs:716) with its site-table entries, the uniform of block 1 and -- for a Categorical
    //           target -- the index resampled from the constant prior table with its prior log-probability (mh.rs:516-530);
    //   part 1: gaussian_z (mh.rs:128-132) from block 1 and ln u of the accept uniform of block 2.
    // With W >= 3 the two parts run on two waves (block 1 is then generated twice: the parts stay independent) -- unless nobody
    // needs block 1's uniform itself (exp_mask bit 4096, the host's: a stream program without Categorical or bool sites draws it only
    // for the accept test of a log-space walk at a non-positive value, which then regenerates it).
    // The accept test `log_alpha >= 0 || u < exp(log_alpha)` (mh.rs:733) is decided from ln u where that decides it: ln u < log_alpha - m
    // accepts and ln u > log_alpha + m rejects for m = 1e-9 (1 + |log_alpha|), three orders above the error of the two transcendentals
    // (exp underflows only where ln u > log_alpha + m anyway: u >= 2^-53); inside the margin (probability ~1e-9 per lane and step), for
    // u = 0, a NaN and when the accept uniform is another block's, the control wave evaluates the reference's own expression.  The
    // decisions are the reference's -- and exp leaves the control wave's path.
    auto publish_rng = [&](int it, int part) __attribute__((always_inline)) {
        double *b = xch + (long long)(8 * (it & 1)) * tw;
        FgStream rng; rng.k0 = sk0; rng.k1 = sk1; rng.c0 = gchain; rng.c2 = (uint32_t)it; rng.c3 = FG_RNG_MH;
        unsigned long long ra, rb;
        if (part == 0) {
            rng.c1 = 0; fg_rng_block(rng, ra, rb);
            const int tg = (int)fg_pick(ra, (uint32_t)P.S);
            const int ts = P.site_slot[tg], tvv = P.site_vtype[tg];               // per-lane gathers of small tables
            const int cb = P.site_cat[2 * tg], cK = P.site_cat[2 * tg + 1];
            double u1 = 0.0;
            if (!skip_u1) { rng.c1 = 1; fg_rng_block(rng, ra, rb); u1 = fg_u01_of(ra); b[2 * tw] = u1; }
            b[0] = fg_as_double((long long)tg);
            b[4 * tw] = fg_as_double((long long)(uint32_t)ts | ((long long)tvv << 32));
            b[5 * tw] = fg_as_double((long long)(uint32_t)cb | ((long long)cK << 32));
            if (tvv == 3 && cK > 0) {                                  // first index whose cumulative probability reaches u, clamped (distribution.rs:771-784)
                double cum = 0.0; int idx = cK;
                for (int i0 = 0; i0 < cK; i0 += 4) {                  // four table entries in flight
                    double pv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) pv[q] = (i0 + q < cK) ? pool_rd(cb + i0 + q) : 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (i0 + q < cK) { cum += pv[q]; if (idx == cK && !(cum < u1)) idx = i0 + q; }
                }
                const int prop = idx < cK - 1 ? idx : cK - 1;
                b[6 * tw] = fg_as_double((long long)prop);
                b[7 * tw] = pool_rd(cb + cK + prop);                  // the table's precomputed ln p (-inf for p <= 0)
            }
        } else {
            rng.c1 = 1; fg_rng_block(rng, ra, rb);
            b[tw] = fg_cold_gaussian_z(ra, rb);
            rng.c1 = 2; fg_rng_block(rng, ra, rb);
            b[3 * tw] = fg_cold_lnu(ra);                              // ln u of block 2 (NaN for u = 0), off the control wave's path
        }
    };

    // ---- control-wave state
    double lw = 0.0, old_cell = 0.0, lqf = 0.0, lqr = 0.0, scale = 1.0, u_acc = 0.0 /* the accept uniform, or NaN: regenerate it */, lnu_acc = 0.0 /* its ln, or NaN */;
    int tslot = 0, kind0 = 0, kind_new = 0, nb_acc = 2;
    long long g = 0;
    unsigned long long nacc = 0;
    int nbad = 0;                                                  // this chain's row-less Categorical sites whose index is out of range
    bool cur_bad = false;                                          // ... and whether the current proposal replaces one

    if (wv == 0) {
        fg_load_values(P, X, c, slots, tw);
        lw = M.lw[c];
        for (int j = 0; j < n_cu; ++j) {
            const FgMhCatU cu = seg.catu[j];
            const long long zi = fg_as_i64(slots[cu.slot * tw]);
            nbad += (zi < 0 || zi >= (long long)cu.K) ? 1 : 0;
        }
        // the control wave's instruction stream is the path of its tile: it is served before the term and random-number waves
        // of the tiles it shares a SIMD with (exp_mask bit 32 switches this off: A/B)
        if (!(exp_mask & 32)) __builtin_amdgcn_s_setprio(2);
    }
    for (int k = (int)threadIdx.x; k < pool_n; k += (int)blockDim.x) pool_l[k] = P.pool[k];
    if (pool_n > 0) __syncthreads();                               // the random-number waves read the staged tables below
    if (wv == rng_wave) { publish_rng(iter0, 0); if (n_steps > 1) publish_rng(iter0 + 1, 0); }
    if (wv == rng_wave1) { publish_rng(iter0, 1); if (n_steps > 1) publish_rng(iter0 + 1, 1); }
    // this wave's share of every record class, read once (indexing the kernel argument by the wave number inside the step loop
    // is a scalar-memory round trip per class per step)
    // -- for the instantiations with lookup classes, whose phase B is a chain of round trips; the fast-Normal one measured faster
    // re-reading its three bounds than keeping them (it is short of scalar registers)
    int sa_[FG_MH_NCLS], sb_[FG_MH_NCLS];
#pragma unroll
    for (int q = 0; q < FG_MH_NCLS; ++q) { sa_[q] = RK == 0 ? 0 : seg.r[q][wv]; sb_[q] = RK == 0 ? 0 : seg.r[q][wv + 1]; }
#define sa(q) (RK == 0 ? seg.r[q][wv] : sa_[q])
#define sb(q) (RK == 0 ? seg.r[q][wv + 1] : sb_[q])
    __syncthreads();
#ifdef FG_MH_PROF
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = __builtin_readcyclecounter();
#endif
    if (exp_mask & 384) {                                          // tiles that start together on a CU run their control-wave phases at the same time: started a part of a step apart
                                                                   // (bit 128: by tile number, the host's choice with >= 3 tiles per CU; 256: by quarter of the grid) they fill each other's gaps
        const unsigned ph = (exp_mask & 128) ? (blockIdx.x & 3u) : ((blockIdx.x * 4u / gridDim.x) & 3u);
        for (unsigned q = 0; q < ph; ++q) __builtin_amdgcn_s_sleep(27);
    }
    for (int t = 0; t <= n_steps; ++t) {
        const int iter = iter0 + t;
        // ---- phase A
        // the next proposal's inputs first: its adaptation state comes from HBM / L2 while step t - 1 is being finished
        const double *b = xch + (long long)(8 * (iter & 1)) * tw;
        int n_target = 0, n_tslot = 0; uint32_t n_tv = 0u; long long n_g = 0;
        double n_scale = 1.0; int n_kind = 0;
        const bool do_prop = t < n_steps && !(exp_mask & 4);
#define FG_MH_NEXT_INPUTS                                                                                                   \
        if (do_prop) {                                                                                                      \
            n_target = (int)fg_as_i64(b[0]);                                                                                \
            const long long st = fg_as_i64(b[4 * tw]);                                                                      \
            n_tslot = (int)(uint32_t)st; n_tv = (uint32_t)(st >> 32);                                                       \
            n_g = (long long)n_target * X.C + c;                                                                            \
            const fg_u32x4 a0 = *(const fg_u32x4 *)(M.ad + n_g);           /* {scale, kind}: get_scale  mcmc_utils.rs:70-77 */ \
            n_scale = fg_dbl(a0[0], a0[1]);                                                                                 \
            n_kind = (int)a0[2];                                                                                            \
        }
        // the row-less tail of log_prior (FgMhSeg): the constants, eight per scalar load; from the cells while a chain holds a bad index
#define FG_MH_CATU_TAIL                                                                                                     \
        if (n_cu > 0) {                                                                                                     \
            if (__builtin_expect(__any(nbad != 0), 0)) {                                                                    \
                for (int j = 0; j < n_cu; ++j) {                                                                            \
                    const FgMhCatU cu = seg.catu[j];                                                                        \
                    const long long zi = fg_as_i64(slots[cu.slot * tw]);                                                    \
                    pri += (zi < 0 || zi >= (long long)cu.K) ? FG_NEG_INF : fg_uniform(seg.catu_c[j]);                      \
                }                                                                                                           \
            } else if (seg.catu_same) {                             /* one constant for all of them (equal tables): no loads */    \
                const double c0_ = fg_uniform(seg.catu_c0);                                                                 \
                int j = 0;                                                                                                  \
                for (; j + 8 <= n_cu; j += 8) { _Pragma("unroll") for (int q = 0; q < 8; ++q) pri += c0_; }                  \
                for (; j < n_cu; ++j) pri += c0_;                                                                           \
            } else {                                                /* eight per scalar load, the next eight on their way */  \
                fg_u32x16 cb_ = fg_fetch_grec((const FgGradRec *)seg.catu_c, 0);                                            \
                for (int j = 0; j < n_cu; j += 8) {                                                                         \
                    const fg_u32x16 cn_ = fg_fetch_grec((const FgGradRec *)seg.catu_c, (j >> 3) + 1);                       \
                    _Pragma("unroll") for (int q = 0; q < 8; ++q) if (j + q < n_cu) pri += fg_dbl(cb_[2 * q], cb_[2 * q + 1]); \
                    cb_ = cn_;                                                                                              \
                }                                                                                                           \
            }                                                                                                               \
        }
        // log_prior and log_likelihood of step t - 1's proposal: the terms in program order.  SPLIT (long programs): one chain per
        // wave -- a wave alone on its tile's path issues an instruction every 6 to 9 cycles
        // (profiles/round1_f64_issue_microbench.txt), so the two independent sums run side by side on two waves and meet at a barrier
        double pri = 0.0, lik = 0.0;
        if (SPLIT) {
            if (wv == 0) { FG_MH_NEXT_INPUTS }
            if (t > 0 && !(exp_mask & 2)) {
#ifdef FG_MHMW_SUM_PRI       /* a unit compiled at run time: the sums as straight-line code (fg_jit.cpp) */
                if (wv == 0) { pri = FG_MHMW_SUM_PRI(); FG_MH_CATU_TAIL }
                else if (wv == 1) xch[16 * tw] = FG_MHMW_SUM_LIK();
#else
                if (wv == 0) { pri = fg_inorder_sum1(terms, n_pri, tw); FG_MH_CATU_TAIL }
                else if (wv == 1) {
                    // the second adder is on its tile's path like the control wave: served first while it adds (exp_mask bit 16384 switches this off: A/B)
                    if (!(exp_mask & (32 | 16384))) __builtin_amdgcn_s_setprio(2);
                    xch[16 * tw] = fg_inorder_sum1(terms + (long long)n_pri * tw, n_lik, tw);
                    if (!(exp_mask & (32 | 16384))) __builtin_amdgcn_s_setprio(0);
                }
#endif
                // LDS only crosses this barrier: wait for the LDS counter and leave the control wave's adaptation-state gather (64
                // lines from L2, issued above) in flight -- __syncthreads() would drain it here
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();
                if (wv == 0) lik = xch[16 * tw];
            }
        }
        if (wv == 0) {
            if (!SPLIT) {                                                  // short programs: both chains on the control wave, no third barrier
                FG_MH_NEXT_INPUTS
#ifdef FG_MHMW_SUMS2
                if (t > 0 && !(exp_mask & 2)) { FG_MHMW_SUMS2(pri, lik); FG_MH_CATU_TAIL }
#else
                if (t > 0 && !(exp_mask & 2)) { fg_inorder_sums2(terms, n_pri, terms + (long long)n_pri * tw, n_lik, tw, pri, lik); FG_MH_CATU_TAIL }
#endif
            }
#undef FG_MH_NEXT_INPUTS
#undef FG_MH_CATU_TAIL
            if (t > 0 && !(exp_mask & 2)) {                                // finish step t - 1
                const int itp = iter - 1;
                const bool adapt = itp < n_warmup;
                FG_PROF_T(0)
#ifdef FG_MHMW_NS
                double fac = 0.0;
                if (n_fac > 0) fac = fg_inorder_sum1(terms + (long long)(n_pri + n_lik) * tw, n_fac, tw);
                const double prop_lw = pri + lik + fac;                    // total_log_weight  trace.rs:168-177
#else
                const double prop_lw = pri + lik + 0.0;                    // total_log_weight (no factor statement has a record)
#endif
                const double log_alpha = prop_lw - lw + (lqr - lqf);       // + dim_term == 0 (fixed structure)  mh.rs:731-732
                // mh.rs:733  log_alpha >= 0 || u < exp(log_alpha), from ln u outside the margin (above)
                const double mrg = 1e-9 * (1.0 + fabs(log_alpha));
                const bool sure_acc = log_alpha >= 0.0 || lnu_acc < log_alpha - mrg;
                const bool sure_rej = lnu_acc > log_alpha + mrg;
                bool accept = sure_acc;
                if (__builtin_expect(__any(!sure_acc && !sure_rej), 0)) {
                    double u_ = u_acc;
                    if (__any(!sure_acc && !sure_rej && u_ != u_)) { const double ur = fg_cold_u01_pair(sk0, sk1, gchain, (uint32_t)nb_acc, (uint32_t)itp, FG_RNG_MH).a; if (u_ != u_) u_ = ur; }
                    const bool exact = (log_alpha >= 0.0) || (u_ < fg_cold_exp(log_alpha));
                    if (!sure_acc && !sure_rej) accept = exact;
                }
                double sc = scale;
                if (adapt) {                                               // DiminishingAdaptation::update  mcmc_utils.rs:88-150
                    const fg_u32x4 a1 = *(const fg_u32x4 *)((const char *)(M.ad + g) + 16);   // {log_scale, total, accepted}
                    const uint32_t tot = a1[2] + 1u;
                    const uint32_t acn = a1[3] + (accept ? 1u : 0u);
                    double ls = fg_dbl(a1[0], a1[1]);
                    if (tot >= 10u) { const FgD2 r = fg_cold_mh_adapt(ls, acn, tot, M.step_tab, M.step_n); sc = r.a; ls = r.b; }
                    if (live) {
                        const unsigned long long lb = (unsigned long long)__double_as_longlong(ls);
                        const fg_u32x4 w1 = { (uint32_t)lb, (uint32_t)(lb >> 32), tot, acn };
                        *(fg_u32x4 *)((char *)(M.ad + g) + 16) = w1;
                        M.ad[g].scale = sc;
                    }
                }
                if (live && kind_new != kind0) M.ad[g].kind = kind_new;
                if (accept) { lw = prop_lw; nacc += 1ull; if (live) X.values[g] = fg_as_i64(slots[tslot * tw]); if (cur_bad) nbad -= 1; }
                else slots[tslot * tw] = old_cell;
                if ((!adapt || M.rec_all) && draws && live) {              // recorded cells of the CURRENT state (mh.rs:1010)
                    long long *row = draws + (long long)(t - 1 - first_sample_t) * M.n_rec * X.C + c;
                    for (int r = 0; r < M.n_rec; ++r) row[(long long)r * X.C] = fg_as_i64(slots[M.rec[r] * tw]);
                }
                // the state fetched above is stale where the next proposal hits the site just updated
                if (do_prop && n_g == g) { n_scale = sc; n_kind = kind_new; }
            }
            FG_PROF_T(1)
            if (do_prop) {                                                 // proposal of step t (mh.rs:183-294, 516-530, 557-567)
                FgMhCtx mh;
                mh.z = b[tw];
                const double lnu2 = b[3 * tw];
                g = n_g; tslot = n_tslot;
                mh.target = tslot; mh.scale = n_scale; mh.kind = n_kind;
                kind0 = n_kind;
                mh.next_block = 2;
                mh.lqf = 0.0; mh.lqr = 0.0;
                mh.ov_kind = M.ov_kind; mh.ov_lo = M.ov_lo; mh.ov_hi = M.ov_hi;
                mh.old_cell = slots[tslot * tw];
                const uint32_t tv = n_tv;
                int kind_eff = FG_PROP_AUTO;
                if (tv == 0u) { kind_eff = mh.ov_kind ? mh.ov_kind[tslot] : FG_PROP_AUTO; if (kind_eff == FG_PROP_AUTO) kind_eff = mh.kind; }
#ifdef FG_MHMW_NS            /* a program without a score stream: a lane whose proposal needs the model (an undecided kind, PriorResample, a Categorical site
                                with a computed table) gets it from its target's OWN statement, interpreted (fg_mhmw_model_proposals) */
                const int cat_K0 = (int)(fg_as_i64(b[5 * tw]) >> 32);
                const bool walk = tv == 0u ? (kind_eff == FG_PROP_GAUSSIAN || kind_eff == FG_PROP_LOGSPACE || kind_eff == FG_PROP_REFLECT)
                                           : (tv == 1u || tv == 2u || tv == 4u || (tv == 3u && cat_K0 > 0));
                const bool mixed = !__all(walk);
                FgMhmwPre mp = {0.0, 0.0, 0, 2};
                if (mixed) mp = fg_mhmw_model_proposals(P.ins, (const int *)srt, P.pool, slots, live, walk, n_target, mh, X.seed, gchain, (uint32_t)iter);
                if (walk) {
#else
                // f64_kind (mh.rs:339-358): an undecided site is LogSpace iff its current value is positive and its prior density
                // at -1.0 is -inf.  Lanes hold different sites: one pass per distinct undecided site in the wave (transient -- a
                // kind is decided once per (site, chain)).
                const bool undecided = tv == 0u && kind_eff == FG_PROP_AUTO;
                unsigned long long todo = __ballot(undecided);
                while (todo) {
                    const int leader = __ffsll((long long)todo) - 1;
                    const int tl = __builtin_amdgcn_readlane(n_target, leader);
                    const unsigned long long same = __ballot(undecided && n_target == tl);
                    const fg_u32x16 r = fg_fetch_grec(P.sstream, P.site_rec[tl]);
                    FgAcc3 dummy = {0.0, 0.0, 0.0};
                    const double probe = fg_score_one<RK>(r, -1.0, slots[r[1] * tw], P.pool, slots, tw, dummy);
                    if (undecided && n_target == tl) { kind_eff = (mh.old_cell > 0.0 && !fg_finite(probe)) ? FG_PROP_LOGSPACE : FG_PROP_GAUSSIAN; mh.kind = kind_eff; }
                    todo &= ~same;
                }
#endif
                if (tv == 3u) {                               // usize target: the index resampled from the constant prior table (mh.rs:516-530)
                    const long long ct = fg_as_i64(b[5 * tw]);
                    const int cat_base = (int)(uint32_t)ct, cat_K = (int)(ct >> 32);
                    const long long prop = fg_as_i64(b[6 * tw]);
                    const long long cur = fg_as_i64(mh.old_cell);
                    mh.lqf += b[7 * tw];                                   // prior log-probabilities of the proposed and the current index
                    mh.lqr += (cur < 0 || cur >= (long long)cat_K) ? FG_NEG_INF : pool_rd(cat_base + cat_K + (int)cur);
                    mh.next_block = 2;
                    slots[tslot * tw] = fg_as_double(prop);
                } else fg_mh_walk_proposal(mh, tv, kind_eff, tslot, slots, tw);
                if (n_cu > 0) { const long long cur = fg_as_i64(mh.old_cell); const int cK = (int)(fg_as_i64(b[5 * tw]) >> 32); cur_bad = tv == 3u && cK > 0 && (cur < 0 || cur >= (long long)cK); }
#ifdef FG_MHMW_NS
                }
                if (!walk) { mh.lqf = mp.lqf; mh.lqr = mp.lqr; mh.kind = mp.kind; mh.next_block = mp.next_block; }
#endif
                old_cell = mh.old_cell; lqf = mh.lqf; lqr = mh.lqr; scale = mh.scale; kind_new = mh.kind;
                nb_acc = mh.next_block;                                    // the accept uniform's block (mh.rs:733)
                lnu_acc = nb_acc == 2 ? lnu2 : NAN;
                u_acc = NAN;
                if (!skip_u1 && __any(nb_acc == 1)) { const double u1 = b[2 * tw]; if (nb_acc == 1) u_acc = u1; }
            }
        } else if (t > 0 && t + 1 < n_steps && !(exp_mask & 8)) {          // buffer (iter + 1) & 1 was last read in phase A of step t - 1
            if (wv == rng_wave) publish_rng(iter + 1, 0);
            if (wv == rng_wave1) publish_rng(iter + 1, 1);
        }
        if (t == n_steps) break;
        FG_PROF_T(2)
        __syncthreads();                                                   // the proposal is in the tile; random numbers of step t + 1 published
        FG_PROF_T(3)
        // ---- phase B: every wave scores its share of the statements (on the step's path: ahead of other tiles' random-number waves)
        if (b_prio) __builtin_amdgcn_s_setprio(1);
        if (!(exp_mask & 1)) {
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
        }
        FG_PROF_T(4)
        if (b_prio) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        FG_PROF_T(5)
    }
#ifdef FG_MH_PROF
    if (blockIdx.x == 0 && lane == 0) for (int q = 0; q < 8; ++q) fg_mh_prof[wv][q] = prof_[q];
#endif
    if (wv == 0 && live) { M.lw[c] = lw; M.n_acc[c] += nacc; }
#undef sa
#undef sb
}
