// fg_mh_interp.hip -- adaptive_mcmc_chain's step loop (src/inference/mh.rs:698-744, 938-1014) for programs that need the interpreter
// (no score stream: a parameter that is an expression, a select ...), with a 64-chain tile shared by W waves.
//
// k_mh_steps gives such a program one wave per tile, and a step is one proposal plus ONE scoring run of the whole program
// (mh.rs:1186-1202) -- S + O statements interpreted one after the other by a lone wave (~600-1 400 cycles each,
// tools/mb_interp_costs.py).  The statements of a scoring run are independent of each other; only the three accumulators are
// sequential.  Here wave 0 makes the proposal exactly as k_mh_steps does (the same code), then every wave interprets ITS contiguous
// run of statements (host split by instruction cost) and leaves each statement's term in an LDS row (fg_exec's TM mode); wave 0
// adds the rows in program order into log_prior / log_likelihood / log_factors -- the reference's in-order sums, bit for bit --
// and finishes the step (accept, DiminishingAdaptation, recording).  Site rows are shared; expression temporaries, Categorical
// tables and select options are private to the wave (FgRemap, fg_interp.h).  A lane whose proposal needs the model (an undecided
// kind, PriorResample, a Categorical site with a computed table) gets it from its target's own statement, run in the
// propose-and-score mode on wave 0 before the scoring run -- not from a whole propose-and-score run of the program, which is what
// k_mh_steps falls back to as soon as ONE lane of the wave needs it.  Identical to k_mh_steps for every W (tests/test_gpu_mh.py::test_mh_interp_multiwave_is_bit_identical).
#include "fg_engine_internal.h"
#include "fg_cold.h"

#define FG_MHI_MAX 8

struct FgMhi {
    int ins_off[FG_MHI_MAX + 1];     // wave w interprets instructions [ins_off[w], ins_off[w + 1]) of ins_fast ...
    int stmt_off[FG_MHI_MAX + 1];    // ... which hold statements [stmt_off[w], stmt_off[w + 1])
    const unsigned char *stmt_acc;   // [n_stmt] accumulator of each statement: 0 log_prior, 1 log_likelihood, 2 log_factors
    const int *site_ins;             // [S][2] {first instruction, count} of each site's own sample statement in P.ins (generic opcodes)
    int n_stmt;
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
    double *terms = lds + (long long)(P.S + W * np) * tw + lane;                     // one row per statement
    FgRemap rm;
    rm.pi = 0xffffffffu; rm.n_shared = (uint32_t)P.S; rm.woff = (uint32_t)(wv * np); rm.pert = (uint32_t)(P.n_slots + wv * np);
    for (int j = wv; j < P.S; j += W) slots[P.site_slot[j] * tw] = fg_as_double(X.values[(long long)j * X.C + c]);
    slots[(P.n_slots - 1 + rm.woff) * tw] = 0.0;                                     // the wave's always-zero slot
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
        fg_exec<FG_MODE_SCORE, false, true, false, true>(P.ins_fast + i0, i1 - i0, P.pool, slots, tw, A, nullptr, nullptr, 0, false, nullptr, &rm, terms + (long long)s0 * tw);
        __syncthreads();                                     // every statement's term is in its row
        if (wv == 0)
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

// OCC = waves per SIMD the register budget allows: 2 (256 VGPRs) when LDS holds a CU to eight waves anyway, 4 (128 VGPRs, more of the
// cold propose-and-score path spilled) when more tiles fit -- poisson_glm 2.7e9 -> 3.8e9 chain-steps/s at 65 536 chains
#define FG_MHI_KERNEL(OCC) \
__global__ __attribute__((amdgpu_waves_per_eu(OCC, OCC))) __launch_bounds__(FG_WAVE * FG_MHI_MAX) \
void k_mh_interp_mw_steps_occ##OCC(FgProgramDev P, FgChainCtx X, FgMhDev M, FgMhi seg, int iter0, int n_steps, int n_warmup, long long *draws, int first_sample_t) { \
    fg_mh_interp_mw_body<OCC>(P, X, M, seg, iter0, n_steps, n_warmup, draws, first_sample_t); }
FG_MHI_KERNEL(2)
FG_MHI_KERNEL(4)

static long long mhi_ins_cost(const FgIns &in) {       // the weights of fg_hmc_interp.hip's split
    const uint32_t code = FG_INS_OPCODE(in.op);
    if (code == FG_OP_NORMAL_FAST) return 3;
    if (code < 17u) return (in.op & FG_F_HOISTED) ? 10 : 16;
    switch (code) {
    case FG_OP_EXP: case FG_OP_LN: case FG_OP_SIN: case FG_OP_COS: case FG_OP_TANH: return 6;
    case FG_OP_POW: case FG_OP_RPOW: return 14;
    case FG_OP_DIV: case FG_OP_RDIV: case FG_OP_SQRT: return 3;
    case FG_OP_DOT: return 1 + (long long)in.opnd[1] / 2;
    default: return 1;
    }
}

int fg_mh_interp_launch(fg_engine *e, int iter0, int n_steps, long long *draws, int first_sample_t) {
    if (e->interp_mw_disabled || e->gt || e->tw != FG_WAVE || e->P.sstream != nullptr) return FG_E_UNSUPPORTED;
    for (int j = 0; j < e->S; ++j) if (e->prog->site_slot[j] >= e->S) return FG_E_UNSUPPORTED;
    const std::vector<FgIns> &ins = e->prog->ins_fast;
    const int n_ins = e->prog->n_ins;
    if (e->mhi_W <= 0) {
        // statements: an instruction that ends one adds a term (a distribution, FACTOR, CONSTLIK)
        std::vector<int> stmt_end;                           // index past the last instruction of each statement
        std::vector<unsigned char> acc;
        for (int k = 0; k < n_ins; ++k) {
            const uint32_t code = FG_INS_OPCODE(ins[k].op);
            if (code == FG_OP_NORMAL_FAST || code < 17u) { stmt_end.push_back(k + 1); acc.push_back((ins[k].op & FG_F_OBSERVE) ? 1 : 0); }
            else if (code == FG_OP_FACTOR) { stmt_end.push_back(k + 1); acc.push_back(2); }
            else if (code == FG_OP_CONSTLIK) { stmt_end.push_back(k + 1); acc.push_back(1); }
        }
        const int n_stmt = (int)stmt_end.size();
        if (n_stmt < 4 || stmt_end.back() != n_ins) return FG_E_UNSUPPORTED;
        auto lds_for = [&](int W) { return (size_t)((long long)e->S + (long long)W * (e->n_slots - e->S + 1) + n_stmt) * FG_WAVE * sizeof(double); };
        int W = 2;
        int forced = 0;
        if (const char *sp = std::getenv("FG_MH_INTERP_WAVES")) forced = std::atoi(sp);
        const int wcap = std::min(FG_MHI_MAX, n_stmt / 2);
        if (forced > 0) W = std::max(2, std::min(forced, wcap));
        else {
            // W = (a CU's sixteen wave slots) / (tiles it gets), as far as LDS keeps all of its tiles resident: 65 536 chains -> 4 for
            // short programs, 2 for alldists (its tile is 46 KB at W = 4); 16 384 and fewer -> 8   [tools/bench_mh_interp.py]
            const long long n_cu = std::max(1, e->n_simd / 4), tiles = (e->C + FG_WAVE - 1) / FG_WAVE, per_cu = (tiles + n_cu - 1) / n_cu;
            while (2 * W <= wcap && 2 * W * per_cu <= 16 && lds_for(2 * W) * (size_t)per_cu <= 160 * 1024) W *= 2;
        }
        while (W > 1 && lds_for(W) > 160 * 1024) --W;
        if (W < 2) return FG_E_UNSUPPORTED;
        std::vector<long long> cum(n_stmt + 1, 0);           // work before statement k
        for (int k = 0, i = 0; k < n_stmt; ++k) { long long cs = 0; for (; i < stmt_end[k]; ++i) cs += mhi_ins_cost(ins[i]); cum[k + 1] = cum[k] + cs; }
        e->mhi_ins_off.assign(FG_MHI_MAX + 1, n_ins); e->mhi_stmt_off.assign(FG_MHI_MAX + 1, n_stmt);
        e->mhi_ins_off[0] = 0; e->mhi_stmt_off[0] = 0;
        for (int w = 1, k = 0; w < W; ++w) {                 // contiguous runs of statements, cut nearest to w / W of the work
            const long long target = cum[n_stmt] * w / W;
            while (k < n_stmt && cum[k] < target) ++k;
            k = std::min(std::max(k, e->mhi_stmt_off[w - 1] + 1), n_stmt - (W - w));      // every wave gets at least one statement
            e->mhi_stmt_off[w] = k; e->mhi_ins_off[w] = stmt_end[k - 1];
        }
        HIPCHK(hipMalloc((void **)&e->d_mhi_acc, (size_t)n_stmt));
        HIPCHK(hipMemcpy(e->d_mhi_acc, acc.data(), (size_t)n_stmt, hipMemcpyHostToDevice));
        // every site's own sample statement in the generic program: from the instruction after the previous statement's last to its distribution
        std::vector<int> site_ins((size_t)2 * e->S, -1);
        const std::vector<FgIns> &gen = e->prog->ins;
        for (int k = 0, begin = 0; k < n_ins; ++k) {
            const uint32_t code = FG_INS_OPCODE(gen[k].op);
            const bool ends = code < 17u || code == FG_OP_FACTOR || code == FG_OP_CONSTLIK;
            if (!ends) continue;
            if (code < 17u && !(gen[k].op & FG_F_OBSERVE))
                for (int j = 0; j < e->S; ++j) if (e->prog->site_slot[j] == (int)gen[k].aux) { site_ins[2 * j] = begin; site_ins[2 * j + 1] = k + 1 - begin; }
            begin = k + 1;
        }
        for (int j = 0; j < e->S; ++j) if (site_ins[2 * j] < 0) { fg_set_error("fg_mh_interp: a site without a sample statement"); return FG_E_STATE; }
        if (dev_upload(&e->d_mhi_site_ins, site_ins)) return FG_E_HIP;
        e->mhi_W = W; e->mhi_n_stmt = n_stmt; e->mhi_lds = lds_for(W);
        {   // more than eight waves on a CU need the 128-VGPR build
            const long long n_cu = std::max(1, e->n_simd / 4), tiles = (e->C + FG_WAVE - 1) / FG_WAVE, per_cu = (tiles + n_cu - 1) / n_cu;
            const long long resident = std::min<long long>(per_cu, (160 * 1024) / (long long)e->mhi_lds);
            e->mhi_occ = resident * W > 8 ? 4 : 2;
            if (const char *sp = std::getenv("FG_MH_INTERP_OCC")) e->mhi_occ = std::atoi(sp) <= 2 ? 2 : 4;
        }
    }
    FgMhi seg;
    for (int w = 0; w <= FG_MHI_MAX; ++w) { seg.ins_off[w] = e->mhi_ins_off[w]; seg.stmt_off[w] = e->mhi_stmt_off[w]; }
    seg.stmt_acc = e->d_mhi_acc; seg.n_stmt = e->mhi_n_stmt; seg.site_ins = e->d_mhi_site_ins;
    const unsigned tiles = (unsigned)((e->C + e->tw - 1) / e->tw);
#define FG_MHI_LAUNCH(K) do { if (int rc = set_lds(K, e->mhi_lds)) return rc; \
    hipLaunchKernelGGL(K, dim3(tiles), dim3(FG_WAVE * e->mhi_W), e->mhi_lds, e->stream, e->P, e->X, e->M, seg, iter0, n_steps, e->mh_warmup, draws, first_sample_t); } while (0)
    if (e->mhi_occ == 4) FG_MHI_LAUNCH(k_mh_interp_mw_steps_occ4); else FG_MHI_LAUNCH(k_mh_interp_mw_steps_occ2);
#undef FG_MHI_LAUNCH
    HIPCHK(hipGetLastError());
    return FG_OK;
}
