// fg_mh.hip -- adaptive single-site Metropolis-Hastings (single_site_mh_step x n, src/inference/mh.rs:698-744, 938-1014) with a
// 64-chain tile shared by W waves, for programs whose statements all have a score-stream record.
//
// One MH step of a chain is a short serial recipe -- pick a site, propose, re-run the model (S + O log-densities), accept,
// adapt -- and at 65 536 chains a one-wave-per-tile kernel puts ONE wave on every SIMD: in-order issue, nothing hides the
// random-number generation, the transcendental functions or the gathers of the adaptation state (round 1: 0.03 of the f64
// peak, 390 spilled VGPRs).  Here the step is software-pipelined over the waves of a workgroup:
//
//   phase A   wave 0 (control): finishes step t-1 -- adds the statements' log-density TERMS in program order (log_prior
//             terms, then log_likelihood terms: the same sums in the same order as one scoring run, interpreters.rs:76-163),
//             accept (mh.rs:731-733), DiminishingAdaptation::update (mcmc_utils.rs:88-150), commit / roll back -- and
//             makes the proposal of step t (mh.rs:183-294, 516-530, 557-567) from the random numbers published two
//             barriers earlier;
//             wave W-1 (random numbers), at the same time: Philox blocks 0..2 of step t+1 -> target site, gaussian_z
//             (mh.rs:128-132) and both candidate accept uniforms, into the other half of a double buffer;
//   barrier
//   phase B   every wave: its share of the statements' log-densities at the proposed state -> term rows in LDS;
//   barrier
//
// Two barriers per step; the scalar record fetches, LDS reads and arithmetic of a term are independent of the other
// terms'.  All proposals here are model-independent (random walks on f64 / u64 / i64 sites, the bool flip, prior resampling
// of Categorical sites with a constant table): the engine keeps programs with other needs on k_mh_steps (fg_engine.hip).
// Values and decisions are those of k_mh_steps: the same random numbers, the same operations in the same order.
#include "fg_engine_internal.h"
#include "fg_gradstream.h"
#include "fg_cold.h"
#include "fg_jit.h"

#include "fg_mh_mw_body.h"
#include "fg_mh_mw2_body.h"
#ifdef FG_MH_PROF
extern hipModule_t fg_mh_prof_module;
static inline void fg_mh_prof_set_module(hipModule_t m) { fg_mh_prof_module = m; }
#endif

// the same step with its serial recipe split over waves (round 4, fg_mh_mw2_body.h: decider / speculative proposer); FG_MH_PIPE=0 keeps
// the one-control-wave loop above (A/B, identity tests)
template <int RK, bool SPLIT>
__global__ __launch_bounds__(FG_WAVE * FG_MH_WMAX, 4) void k_mh_mw2_steps(FgProgramDev P, FgChainCtx X, FgMhDev M, const FgGradRec *srt, FgMhSeg seg, int iter0, int n_steps, int n_warmup,
                                                                            long long *draws, int first_sample_t, int exp_mask, int pool_n) {
    fg_mh_mw2_body<RK, SPLIT>(P, X, M, srt, seg, iter0, n_steps, n_warmup, draws, first_sample_t, exp_mask, pool_n);
}

template <int RK, bool SPLIT>
__global__ __launch_bounds__(FG_WAVE * FG_MH_WMAX, 4) void k_mh_mw_steps(FgProgramDev P, FgChainCtx X, FgMhDev M, const FgGradRec *srt, FgMhSeg seg, int iter0, int n_steps, int n_warmup,
                                                                           long long *draws, int first_sample_t, int exp_mask, int pool_n) {
    fg_mh_mw_body<RK, SPLIT>(P, X, M, srt, seg, iter0, n_steps, n_warmup, draws, first_sample_t, exp_mask, pool_n);
}

// Launch shape of the multi-wave kernel for a program of n_s statements: LDS bytes, waves per tile, the experiment / priority mask,
// whether the two in-order sums run on two waves.
struct FgMhMwShape { size_t lds; int W, exp_mask, split_sums, pool_n, pipe, resident; unsigned tiles; };
static int mh_mw_shape(const fg_engine *e, int n_s, bool stage_pool, FgMhMwShape &sh, bool pipe_ok = false) {
    const fg_program *p = e->prog;
    // the pipelined step loop (fg_mh_mw2_body.h): stream programs
    // -- opt-in (FG_MH_PIPE=1): identical results, but measured 5-20 % slower than the one-control-wave loop at every chain count
    // (profiles/round4_mh_pipeline_experiment.txt): what it takes off the decider's path comes back as the proposer's phase B
    sh.pipe = (pipe_ok && std::getenv("FG_MH_PIPE") && std::atoi(std::getenv("FG_MH_PIPE")) == 1 && !std::getenv("FG_MH_EXP")) ? 1 : 0;
    sh.split_sums = std::getenv("FG_MH_SPLIT") ? (std::atoi(std::getenv("FG_MH_SPLIT")) != 0 ? 1 : 0) : (n_s >= 64 ? 1 : 0);
    int pipe_bits = 0, xrows = 17;                                          // one-control-wave loop: 2 x 8 exchange rows + the log_likelihood sum
    bool all_f64 = true;
    for (int j = 0; j < e->S; ++j) { if (p->site_vtype[j] == FG_USIZE) pipe_bits |= 512; if (p->site_vtype[j] == FG_BOOL) pipe_bits |= 1024; all_f64 = all_f64 && p->site_vtype[j] == FG_F64; }
    if (sh.pipe) {
        if (all_f64 && !e->M.ov_kind) pipe_bits |= 2048;                    // every proposal is a walk on an f64 site with the support-based kind: the proposer's short path
        const int nr = 3 + ((pipe_bits & 512) ? 1 : 0) + ((pipe_bits & 1024) ? 1 : 0);
        xrows = 2 * nr + 9 + (sh.split_sums ? 2 : 0);                       // fg_mh_mw2_body.h: two random-number buffers, 8 candidate rows, the decision, (sum + tag)
    } else if (pipe_ok && !(pipe_bits & (512 | 1024))) pipe_bits |= 4096;   // a stream program without Categorical / bool sites: nobody reads block 1's uniform (fg_mh_mw_body.h)
    sh.lds = (size_t)(e->n_slots + n_s + xrows) * FG_WAVE * sizeof(double); // site values, term rows, exchange rows
    if (sh.lds > 160 * 1024) return FG_E_UNSUPPORTED;
    sh.pool_n = 0;                                                          // stage the constant pool into LDS when it is small and the tile leaves room
    if (stage_pool && p->pool.size() * 8 <= 24 * 1024 && sh.lds + p->pool.size() * 8 <= 160 * 1024 &&
        (160 * 1024) / sh.lds == (160 * 1024) / (sh.lds + p->pool.size() * 8)) { sh.pool_n = (int)p->pool.size(); sh.lds += p->pool.size() * 8; }
    sh.tiles = (unsigned)((e->C + FG_WAVE - 1) / FG_WAVE);
    const long long n_cu = std::max(1, e->n_simd / 4);
    const long long resident = std::max(1LL, std::min<long long>((160 * 1024) / (long long)sh.lds, ((long long)sh.tiles + n_cu - 1) / n_cu));
    sh.resident = (int)resident;
    int W = e->mw_override > 0 ? e->mw_override : 2;
    // (sixteen waves only pay with >= 6 rows per wave where the rows are a score stream's records: reference_model(20), 39 rows, one tile per CU at 8 192 chains:
    // W = 8 4.30 / 6.24e9 adapting / sampling, W = 16 4.18 / 5.93e9; the statements of a program without a stream are whole expression programs -- rule unchanged)
    if (e->mw_override <= 0) while (W < FG_MH_WMAX && resident * W < 16 && n_s >= ((pipe_ok && W >= 8) ? 12 : 4) * W) W *= 2;
    sh.W = std::max(W, 2);                                                  // control wave + random-number wave
    sh.exp_mask = (std::getenv("FG_MH_EXP") ? std::atoi(std::getenv("FG_MH_EXP")) : 0) | pipe_bits;
    if (std::getenv("FG_MH_PRIO") && std::atoi(std::getenv("FG_MH_PRIO")) == 0) sh.exp_mask |= 32;
    else if (resident >= 2) sh.exp_mask |= 64;
    if (resident >= 3 && !(std::getenv("FG_MH_STAGGER") && std::atoi(std::getenv("FG_MH_STAGGER")) == 0)) sh.exp_mask |= 128;   // bit 128: the tiles of a CU start a quarter of a step apart (reference_model(20), four tiles per CU: +4.7 %; two tiles: nothing)   // bit 64: phase-B waves ahead of the random-number waves of the OTHER tiles on the CU (reference_model(20) +3 %; a lone tile loses 2 %)
    if (std::getenv("FG_MH_PRIO2") && std::atoi(std::getenv("FG_MH_PRIO2")) == 0) sh.exp_mask |= 16384;
    // long programs: log_prior and log_likelihood are added by two waves (C5: +11 %); a short one pays more for the extra barrier than
    // the second wave returns (reference_model(20), 4 tiles per CU: -3 %) -- split_sums, above
    return FG_OK;
}
static bool mh_mw_sites_ok(const fg_engine *e) {     // every site must take a model-independent proposal: Categorical sites need a constant table, no PriorResample override
    if (e->gt || e->S < 1 || e->mh_mw_disabled || e->mh_has_prior_resample) return false;     // (tiles in global memory: the one-wave-per-tile kernels, fg_engine.hip)
    for (int j = 0; j < e->S; j++) if (e->prog->site_vtype[j] == FG_USIZE && e->prog->site_cat[2 * j + 1] <= 0) return false;
    return true;
}

// Programs WITHOUT a score stream (expression parameters, ...) on the same kernel: every statement generated (fg_jit_mhmw_source),
// rows in accumulator order (log_prior terms, log_likelihood terms, then the terms of `factor` statements), proposals that need the model (undecided kinds, PriorResample, computed Categorical tables) through the interpreter on
// the target's own statement (`site_ins`, device: [S][2] first instruction and count in the generic program).  Called by
// fg_mh_interp_launch with its statement table.
int fg_mh_mw_nostream_launch(fg_engine *e, int iter0, int n_steps, long long *draws, int first_sample_t, const std::vector<int> &stmt_end,
                             const std::vector<unsigned char> &acc, const int *d_site_ins) {
    if (e->jit_mhns_state < 0 || e->gt || e->S < 1 || e->mh_mw_disabled || n_steps < 1 || e->tw != FG_WAVE) return FG_E_UNSUPPORTED;
    const fg_program *p = e->prog;
    const int n_s = (int)stmt_end.size();
    FgMhMwShape sh;
    if (mh_mw_shape(e, n_s, false, sh) != FG_OK) return FG_E_UNSUPPORTED;
    if (e->jit_mhns_state == 0) {
        e->jit_mhns_state = -1;
        const char *sp = std::getenv("FG_JIT");
        if ((sp && std::atoi(sp) == 0) || std::getenv("FG_MH_EXP") || p->ins_fast.size() > 200000 || acc.size() != (size_t)n_s) return FG_E_UNSUPPORTED;
        int n_acc[3] = {0, 0, 0};
        for (int k = 0; k < n_s; ++k) { if (acc[(size_t)k] > 2) return FG_E_UNSUPPORTED; n_acc[acc[(size_t)k]] += 1; }
        const int n_pri = n_acc[0], n_fac = n_acc[2];
        std::vector<int> rows((size_t)n_s);
        for (int k = 0, a = 0, b = n_pri, c = n_pri + n_acc[1]; k < n_s; ++k) rows[(size_t)k] = acc[(size_t)k] == 0 ? a++ : acc[(size_t)k] == 1 ? b++ : c++;
        std::vector<long long> cost((size_t)p->n_ins);
        for (int k = 0; k < p->n_ins; ++k) cost[(size_t)k] = fg_mhi_ins_cost(p->ins_fast[(size_t)k]);
        std::vector<double> ctab;
        const int nseg_ns = (std::getenv("FG_MH_NSEG_NS") && std::atoi(std::getenv("FG_MH_NSEG_NS")) == 0) ? 0 : sh.W;       // one statement segment per wave (logistic +12 %, poisson_glm +16 %, hier_logsigma +8 %, alldists level)
        // (the launch shape as literals, as for stream programs below)
        const int baked[7] = { 0, n_s, n_pri, e->n_slots, sh.W, sh.exp_mask, sh.pool_n };
        const bool bake = nseg_ns == sh.W && !(std::getenv("FG_MH_BAKE") && std::atoi(std::getenv("FG_MH_BAKE")) == 0);
        if (bake) { std::memcpy(e->jit_mhmw_baked, baked, sizeof baked); e->jit_mhmw_has_baked = true; }
        const std::string src = fg_jit_mhmw_source(p, cost, std::vector<char>((size_t)n_s, 1), 3, sh.split_sums, &ctab, &rows, n_pri, n_fac, true, false, nseg_ns, 16, -1, -1, bake ? baked : nullptr);
        std::vector<char> code;
        if (!src.empty() && src.size() <= (6u << 20) && fg_jit_get_code(src, code, e->jit_log) == FG_OK &&
            hipModuleLoadData(&e->jit_mhns_mod, code.data()) == hipSuccess &&
            hipModuleGetFunction(&e->jit_mhns_fn, e->jit_mhns_mod, "k_mh_mw_jit_steps") == hipSuccess &&
            hipFuncSetAttribute((const void *)e->jit_mhns_fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
            fg_jit_bind_tables(e->jit_mhns_mod, ctab, &e->d_jit_mhns_tab, e->stream) == FG_OK) { e->jit_mhns_state = 1; e->jit_mhns_split = sh.split_sums; }
        else {
            (void)hipGetLastError();
            if (std::getenv("FG_JIT_VERBOSE")) fprintf(stderr, "fugue_amd: multi-wave MH kernel not compiled at run time (%s)\n", e->jit_log.c_str());
            return FG_E_UNSUPPORTED;
        }
    }
    if (e->jit_mhmw_has_baked && (e->jit_mhmw_baked[3] != e->n_slots || e->jit_mhmw_baked[4] != sh.W || e->jit_mhmw_baked[5] != sh.exp_mask || e->jit_mhmw_baked[6] != sh.pool_n)) {
        fg_set_error("the compiled multi-wave MH kernel was generated for another launch shape"); return FG_E_STATE;
    }
    FgMhSeg seg;
    std::memset(&seg, 0, sizeof(seg));
    int n_warmup = e->mh_warmup;
    const FgGradRec *site_ins_as_srt = (const FgGradRec *)d_site_ins;        // the unit reads its `srt` argument as the site_ins table (FG_MHMW_PROBE)
    void *args[] = { &e->P, &e->X, &e->M, &site_ins_as_srt, &seg, &iter0, &n_steps, &n_warmup, &draws, &first_sample_t, &sh.exp_mask, &sh.pool_n };
    HIPCHK(hipModuleLaunchKernel(e->jit_mhns_fn, sh.tiles, 1, 1, FG_WAVE * sh.W, 1, 1, (unsigned)sh.lds, e->stream, args, nullptr));
    e->last_mh_kernel = "k_mh_mw_jit_steps W=" + std::to_string(sh.W) + " (a program without a record stream; statements compiled at run time)";
    return FG_OK;
}

int fg_mh_mw_launch(fg_engine *e, int iter0, int n_steps, long long *draws, int first_sample_t) {
    if (!e->P.sstream || n_steps < 1 || !mh_mw_sites_ok(e)) return FG_E_UNSUPPORTED;
    const fg_program *p = e->prog;
    const int n_s = e->P.n_sstream, n_pri = e->P.n_prior_terms;
    // Categorical sites with a uniform constant table whose terms are the last rows of log_prior: no rows (FgMhSeg)
    if (e->mh_ncu < 0) {
        e->mh_ncu = 0;
        std::vector<int> ks;
        for (int k = 0; k < n_s; ++k) if (p->sstream[k].flags & FG_G_CATC) ks.push_back(k);
        int n_tab_sites = 0;
        for (int j = 0; j < e->S; ++j) n_tab_sites += (p->site_vtype[j] == FG_USIZE && p->site_cat[2 * j + 1] > 0) ? 1 : 0;
        const int n_c = (int)ks.size();
        bool ok = n_c >= 4 && n_c == n_tab_sites && n_c <= n_pri && !(std::getenv("FG_MH_CATU") && std::atoi(std::getenv("FG_MH_CATU")) == 0);
        std::vector<double> cs; std::vector<FgMhCatU> info;
        for (int q = 0; q < n_c && ok; ++q) {
            const FgGradRec &r = p->sstream[ks[(size_t)q]];
            uint32_t w[2]; std::memcpy(w, &r.mimm, 8);                       // {pool base, K}: p[0 .. K), then ln p[0 .. K)
            ok = (int)r.coord == n_pri - n_c + q && w[1] >= 1;               // the last rows of log_prior, in program order
            for (uint32_t i = 1; i < w[1] && ok; ++i) ok = fg_as_i64(p->pool[w[0] + w[1] + i]) == fg_as_i64(p->pool[w[0] + w[1]]) && p->pool[w[0] + i] > 0.0;
            if (ok) ok = p->pool[w[0]] > 0.0;
            if (ok) { cs.push_back(p->pool[w[0] + w[1]]); FgMhCatU cu; cu.slot = (int)r.xi; cu.K = (int)w[1]; info.push_back(cu); }
        }
        if (ok) {
            e->mh_catu_same = 1; e->mh_catu_c0 = cs[0];
            for (double v : cs) if (fg_as_i64(v) != fg_as_i64(cs[0])) e->mh_catu_same = 0;
            while (cs.size() % 8 || cs.size() < (size_t)n_c + 16) cs.push_back(0.0);      // read eight at a time, eight ahead
            FgMhCatU *d_info = nullptr;
            if (dev_upload(&e->d_mh_catu_c, cs) || dev_upload(&d_info, info)) return FG_E_HIP;
            e->d_mh_catu = d_info;
            e->mh_ncu = n_c;
        }
    }
    const int n_cu = e->mh_ncu, n_rows = n_s - n_cu;                       // term rows of the tile
    FgMhMwShape sh;
    if (mh_mw_shape(e, n_rows, e->P.sstream_kinds != 0, sh, true) != FG_OK) return FG_E_UNSUPPORTED;
    const size_t lds = sh.lds;
    const int W = sh.W, split_sums = sh.split_sums;
    int pool_n = sh.pool_n, exp_mask = sh.exp_mask;
    const unsigned tiles = sh.tiles;
    // the kind-sorted copy of the score stream (once per engine); within a class the records keep their program order
    const uint32_t zero_slot = (uint32_t)(e->n_slots - 1);
    auto cls_of = [zero_slot, p](const FgGradRec &r) {
        if (r.flags & FG_G_CATC) return 1;
        if ((r.flags & (FG_G_GEN | FG_G_LIN)) || !(r.flags & FG_G_POW2)) return 5;
        const bool xc = r.xi == zero_slot, mc = r.mi == zero_slot;                // a constant operand reads the always-zero slot and carries its value as the immediate
        if (r.flags & FG_G_NSEL) {                                                 // class 0: an observation against options that are all sites
            uint32_t w[2]; std::memcpy(w, &r.mimm, 8);
            bool sites_only = xc;
            for (uint32_t q = 0; q < w[1] && sites_only; ++q) sites_only = (uint32_t)(fg_as_i64(p->pool[w[0] + 2 * q]) >> 32) == 0u;
            return sites_only ? 0 : 5;
        }
        if (!xc && !mc && r.ximm == 0.0 && r.mimm == 0.0) return 2;
        if (xc && !mc && r.mimm == 0.0) return 3;
        if (!xc && mc && r.ximm == 0.0) return 4;
        return 5;
    };
    if (!e->d_mh_srt) {
        std::vector<FgGradRec> srt;
        e->mh_cls_off[0] = 0;
        for (int c = 0; c < FG_MH_NCLS; ++c) {
            for (int k = 0; k < n_s; ++k) if (cls_of(p->sstream[k]) == c && !(n_cu > 0 && c == 1)) {
                srt.push_back(p->sstream[k]);
                if (n_cu > 0 && (int)srt.back().coord >= n_pri) srt.back().coord -= (uint32_t)n_cu;      // log_likelihood rows follow the shortened log_prior
            }
            e->mh_cls_off[c + 1] = (int)srt.size();
        }
        for (int q = 0; q < 4; ++q) srt.push_back(p->sstream[(size_t)n_s + (size_t)(q & 1)]);    // readable records past the end (fetched ahead, never evaluated)
        if (dev_upload(&e->d_mh_srt, srt)) return FG_E_HIP;
    }
    FgMhSeg seg;
    // in phase B all waves share the records of every class evenly; the remainders of successive classes go to different waves
    int shift = 0;
    // the pipelined loop's proposer spends phase B on the adaptation state and the next step's candidates: no records where the tile has
    // waves to spare, half a share otherwise
    const int w_pro = (sh.pipe && W >= 3) ? ((split_sums && W > 2) ? 2 : 1) : -1;
    for (int c = 0; c < FG_MH_NCLS; ++c) {
        const int a = e->mh_cls_off[c], n = e->mh_cls_off[c + 1] - a;
        int cnt[FG_MH_WMAX] = {0};
        if (w_pro < 0) {
            for (int w = 0; w < W; ++w) cnt[(w + shift) % W] = (int)((long long)n * (w + 1) / W - (long long)n * w / W);
            shift += n % W;
        } else {                                                            // 2 (W - 1) half shares for the others, one (W < 8) or none for the proposer
            const int units = 2 * (W - 1) + (W < 8 ? 1 : 0);
            int at_u = 0, given = 0;
            for (int q = 0; q < W; ++q) {
                const int w = (q + shift) % W;
                const int u = w == w_pro ? (W < 8 ? 1 : 0) : 2;
                const int upto = (int)((long long)n * (at_u + u) / units);
                cnt[w] = upto - given; given = upto; at_u += u;
            }
            shift += n % W;
        }
        int at = a;
        for (int w = 0; w <= FG_MH_WMAX; ++w) { seg.r[c][w] = at; if (w < W) at += cnt[w]; }
    }
    seg.n_cu = n_cu; seg.catu_c = e->d_mh_catu_c; seg.catu = (const FgMhCatU *)e->d_mh_catu; seg.catu_same = e->mh_catu_same; seg.catu_c0 = e->mh_catu_c0;
    const int rk = e->P.sstream_kinds == 0 ? 0 : (e->P.sstream_gen ? 2 : 3);       // record kinds the instantiation understands (fg_score_one)
    // the program compiled at run time (fg_jit.cpp): the same kernel with the general records (class 5: fg_score_one over the record)
    // as sixteen generated statement segments; where they are the minority the operand-pattern classes stay the hand-written
    // runs, which are shorter than what the generator writes for them (reference_model(20), all pattern records: 2.08e10
    // hand-written, 1.57e10 generated; C5, 8 general records of 136: 4.33e9 with the runs, 3.89e9 all generated)
    if (e->jit_mhmw_state == 0) {
        e->jit_mhmw_state = -1;
        const char *sp = std::getenv("FG_JIT");
        if ((!sp || std::atoi(sp) != 0) && !std::getenv("FG_MH_EXP") && p->ins_fast.size() <= 200000) {
            std::vector<long long> cost((size_t)p->n_ins);
            for (int k = 0; k < p->n_ins; ++k) cost[(size_t)k] = fg_mhi_ins_cost(p->ins_fast[(size_t)k]);
            std::vector<char> generated((size_t)n_s);
            int n_gen = 0;
            std::vector<int> rows((size_t)n_s);
            for (int k = 0; k < n_s; ++k) {
                n_gen += (generated[(size_t)k] = cls_of(p->sstream[k]) == 5 ? 1 : 0);
                rows[(size_t)k] = (int)p->sstream[k].coord - ((n_cu > 0 && (int)p->sstream[k].coord >= n_pri) ? n_cu : 0);
            }
            // mostly general records: the few pattern records too (their runs' set-up costs more than the generated statements:
            // linreg, 2 pattern records of 22: 2.49e10 all generated, 2.08e10 with the two runs, 1.66e10 hand-written)
            // Programs of pattern records only (plain Normals with sigma = 2^k): round 3 kept the hand-written record runs -- the generated
            // functions read their tile through generic pointers then (FLAT accesses) and lost.  With LDS-qualified pointers the generated
            // statements, one segment per wave, win (reference_model(20): sampling 2.24e10 -> 2.70e10 at 65 536 chains, 3.9e9 -> 4.6e9 at 8 192;
            // reference_model(8) 3.0e10 -> 3.4e10 / 4.4e9 -> 5.6e9; normal32, reference_model(50) +13 %) -- not where phase B is table lookups
            // (C5: -19 %): profiles/round4_mh_generated_statements.txt.  FG_MH_GEN_ALL = 0 / 1 forces either.
            // ... and whatever the mix of pattern and general records, down to two statements (a survey of the test zoo at 65 536 chains,
            // profiles/round4_zoo_mh.txt: a program of 5 pattern + 5 general records 2.3e10 -> 4.0e10, the README model 3.4e10 -> 4.0e10, none slower).
            bool gen_all = n_s >= (std::getenv("FG_MH_GEN_MIN") ? std::atoi(std::getenv("FG_MH_GEN_MIN")) : 1) && e->mh_cls_off[2] == 0;              // (no class-0 / class-1 lookup records)
            if (const char *gv = std::getenv("FG_MH_GEN_ALL")) gen_all = std::atoi(gv) != 0;
            // one segment per wave of this launch shape (all of a wave's statements in one straight-line function) where every statement
            // is generated; the control wave takes `ctl16` sixteenths of a share
            int nseg = W, ctl16 = 16;                                                            // (reference_model(20), 65 536 chains: sampling 2.21e10 -> 2.61e10; linreg +19 %, hier_scale +9 %)
            if (const char *nv = std::getenv("FG_MH_NSEG")) { if (std::atoi(nv) == 0) nseg = 0; }
            if (const char *cv = std::getenv("FG_MH_CTL16")) ctl16 = std::max(0, std::min(16, std::atoi(cv)));
            if (gen_all) n_gen = n_s;
            if (2 * n_gen >= n_s) for (int k = 0; k < n_s; ++k) generated[(size_t)k] = (n_cu > 0 && (p->sstream[k].flags & FG_G_CATC)) ? 0 : 1;   // (row-less terms have no statement to run)
            std::vector<double> ctab;
            // the launch shape as literals in the unit (one segment per wave only: the unit is then this W's anyway); FG_MH_BAKE=0: kernel arguments as before
            // [7]: the form of the control wave's in-order sums -- the two chains pinned side by side with the rows requested four pairs ahead where a CU holds ONE
            // tile (nothing else fills the control wave's waits: 8 192 chains +8 %); with two tiles per CU the plain statements measured 4 % faster in the sampling phase
            const int baked[8] = { n_cu, n_s - n_cu, n_pri - n_cu, e->n_slots, W, exp_mask, pool_n, sh.resident <= 1 ? 4 : 0 };
            const bool bake = nseg == W && !sh.pipe && !(std::getenv("FG_MH_BAKE") && std::atoi(std::getenv("FG_MH_BAKE")) == 0);
            if (bake) { std::memcpy(e->jit_mhmw_baked, baked, 7 * sizeof(int)); e->jit_mhmw_has_baked = true; }
            // (a handful of general records among many pattern records: the runs alone -- C5 with two tiles on a CU: 7.0e9 against 6.7e9)
            const bool jit_any = std::getenv("FG_MH_JIT_ANY") && std::atoi(std::getenv("FG_MH_JIT_ANY")) != 0;
            const std::string src = (8 * n_gen >= n_s || jit_any) ? fg_jit_mhmw_source(p, cost, generated, rk, split_sums, &ctab, &rows, -1, 0, false, sh.pipe != 0, (2 * n_gen >= n_s) ? nseg : 0, ctl16,
                                                                       (std::getenv("FG_MH_JIT_SUMS") && std::atoi(std::getenv("FG_MH_JIT_SUMS")) == 0) ? -1 : n_pri - n_cu, n_s - n_pri, bake ? baked : nullptr, baked[7]) : std::string();   // (the control wave's in-order sums as inlined straight-line code with the row counts as literals: reference_model(20) sampling 2.71e10 -> 2.89e10; behind a CALL they lost -- a call drains the adaptation-state gather that is in flight across the sums)
            std::vector<char> code;
            if (!src.empty() && src.size() <= (6u << 20) && fg_jit_get_code(src, code, e->jit_log) == FG_OK &&
                hipModuleLoadData(&e->jit_mhmw_mod, code.data()) == hipSuccess &&
                hipModuleGetFunction(&e->jit_mhmw_fn, e->jit_mhmw_mod, "k_mh_mw_jit_steps") == hipSuccess &&
                hipFuncSetAttribute((const void *)e->jit_mhmw_fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess &&
                fg_jit_bind_tables(e->jit_mhmw_mod, ctab, &e->d_jit_mhmw_tab, e->stream) == FG_OK) e->jit_mhmw_state = 1;
            else { (void)hipGetLastError(); if (std::getenv("FG_JIT_VERBOSE")) fprintf(stderr, "fugue_amd: multi-wave MH kernel not compiled at run time (%s)\n", e->jit_log.c_str()); }
        }
    }
    if (e->jit_mhmw_state == 1 && e->jit_mhmw_has_baked) {          // a launch the unit was not generated for (the shape is a function of the engine: never)
        const int now[7] = { n_cu, n_s - n_cu, n_pri - n_cu, e->n_slots, W, exp_mask, pool_n };
        if (std::memcmp(now, e->jit_mhmw_baked, sizeof now) != 0) { fg_set_error("the compiled multi-wave MH kernel was generated for another launch shape"); return FG_E_STATE; }
    }
    if (e->jit_mhmw_state == 1) {
        int n_warmup = e->mh_warmup;
        void *args[] = { &e->P, &e->X, &e->M, &e->d_mh_srt, &seg, &iter0, &n_steps, &n_warmup, &draws, &first_sample_t, &exp_mask, &pool_n };
        HIPCHK(hipModuleLaunchKernel(e->jit_mhmw_fn, tiles, 1, 1, FG_WAVE * W, 1, 1, (unsigned)lds, e->stream, args, nullptr));
#ifdef FG_MH_PROF
        fg_mh_prof_set_module(e->jit_mhmw_mod);
#endif
        e->last_mh_kernel = std::string(sh.pipe ? "k_mh_mw2_jit_steps W=" : "k_mh_mw_jit_steps W=") + std::to_string(W) + " (statements compiled at run time)";
        return FG_OK;
    }
    static bool attr_set_dev[64][16];
    const int variant = 2 * (rk == 0 ? 0 : (rk == 2 ? 1 : 2)) + split_sums + (sh.pipe ? 6 : 0);
    const void *fns[12] = { (const void *)k_mh_mw_steps<0, false>, (const void *)k_mh_mw_steps<0, true>, (const void *)k_mh_mw_steps<2, false>,
                            (const void *)k_mh_mw_steps<2, true>, (const void *)k_mh_mw_steps<3, false>, (const void *)k_mh_mw_steps<3, true>,
                            (const void *)k_mh_mw2_steps<0, false>, (const void *)k_mh_mw2_steps<0, true>, (const void *)k_mh_mw2_steps<2, false>,
                            (const void *)k_mh_mw2_steps<2, true>, (const void *)k_mh_mw2_steps<3, false>, (const void *)k_mh_mw2_steps<3, true> };
    bool &attr_set = attr_set_dev[e->device & 63][variant];
    if (!attr_set) {
        const hipError_t he = hipFuncSetAttribute(fns[variant], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (he != hipSuccess) { fg_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)); return FG_E_HIP; }
        attr_set = true;
    }
#define FG_MH_LAUNCH(K, R, SP) hipLaunchKernelGGL((K<R, SP>), dim3(tiles), dim3(FG_WAVE * W), lds, e->stream, e->P, e->X, e->M, e->d_mh_srt, seg, iter0, n_steps, \
                                                  e->mh_warmup, draws, first_sample_t, exp_mask, pool_n)
    switch (variant) {
        case 0: FG_MH_LAUNCH(k_mh_mw_steps, 0, false); break; case 1: FG_MH_LAUNCH(k_mh_mw_steps, 0, true); break; case 2: FG_MH_LAUNCH(k_mh_mw_steps, 2, false); break;
        case 3: FG_MH_LAUNCH(k_mh_mw_steps, 2, true); break;  case 4: FG_MH_LAUNCH(k_mh_mw_steps, 3, false); break; case 5: FG_MH_LAUNCH(k_mh_mw_steps, 3, true); break;
        case 6: FG_MH_LAUNCH(k_mh_mw2_steps, 0, false); break; case 7: FG_MH_LAUNCH(k_mh_mw2_steps, 0, true); break; case 8: FG_MH_LAUNCH(k_mh_mw2_steps, 2, false); break;
        case 9: FG_MH_LAUNCH(k_mh_mw2_steps, 2, true); break;  case 10: FG_MH_LAUNCH(k_mh_mw2_steps, 3, false); break; default: FG_MH_LAUNCH(k_mh_mw2_steps, 3, true); break;
    }
#undef FG_MH_LAUNCH
    HIPCHK(hipGetLastError());
#ifdef FG_MH_PROF
    fg_mh_prof_set_module(nullptr);
#endif
    e->last_mh_kernel = std::string(sh.pipe ? "k_mh_mw2_steps W=" : "k_mh_mw_steps W=") + std::to_string(W);
    return FG_OK;
}

#ifdef FG_MH_PROF
hipModule_t fg_mh_prof_module = nullptr;      // the module of the last launch when that was the kernel compiled at run time (its own counters)
extern "C" int fg_debug_mh_prof(unsigned long long *out) {
    if (fg_mh_prof_module) {
        hipDeviceptr_t dp = nullptr; size_t bytes = 0;
        if (hipModuleGetGlobal(&dp, &bytes, fg_mh_prof_module, "fg_mh_prof") != hipSuccess || bytes < sizeof(unsigned long long) * FG_MH_WMAX * 8) return -1;
        return hipMemcpy(out, dp, sizeof(unsigned long long) * FG_MH_WMAX * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fg_mh_prof), sizeof(unsigned long long) * FG_MH_WMAX * 8) == hipSuccess ? 0 : -1;
}
#endif
