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
#include "fg_jit.h"

#include "fg_mh_interp_body.h"

// OCC = waves per SIMD the register budget allows: 2 (256 VGPRs) when LDS holds a CU to eight waves anyway, 4 (128 VGPRs, more of the
// cold propose-and-score path spilled) when more tiles fit -- poisson_glm 2.7e9 -> 3.8e9 chain-steps/s at 65 536 chains
#define FG_MHI_KERNEL(OCC) \
__global__ __attribute__((amdgpu_waves_per_eu(OCC, OCC))) __launch_bounds__(FG_WAVE * FG_MHI_MAX) \
void k_mh_interp_mw_steps_occ##OCC(FgProgramDev P, FgChainCtx X, FgMhDev M, FgMhi seg, int iter0, int n_steps, int n_warmup, long long *draws, int first_sample_t) { \
    fg_mh_interp_mw_body<OCC>(P, X, M, seg, iter0, n_steps, n_warmup, draws, first_sample_t); }
FG_MHI_KERNEL(2)
FG_MHI_KERNEL(4)

long long fg_mhi_ins_cost(const FgIns &in) {         // relative cost of an instruction, in units of about ten VALU instructions
    const uint32_t code = FG_INS_OPCODE(in.op);
    if (code == FG_OP_NORMAL_FAST) return 3;
    if (code < 17u) {
        // by distribution: the size of the compiled log-density (the code objects of `alldists`; the lgamma-based ones at 0.6 of
        // their static size -- one branch of lgamma runs), general / hoisted parameters
        static const int general[17] = { 24, 25, 120, 10, 25, 60, 3, 12, 12, 64, 14, 21, 13, 62, 130, 2, 40 };
        static const int hoisted[17] = { 3, 25, 110, 5, 16, 11, 1, 2, 12, 12, 2, 12, 13, 56, 16, 2, 33 };
        if ((in.op & FG_F_XHOIST) && (in.op & FG_F_HOISTED)) return 13;
        return (in.op & FG_F_HOISTED) ? hoisted[code] : general[code];
    }
    switch (code) {
    case FG_OP_EXP: case FG_OP_LN: case FG_OP_SIN: case FG_OP_COS: case FG_OP_TANH: return 6;
    case FG_OP_POW: case FG_OP_RPOW: return 26;
    case FG_OP_DIV: case FG_OP_RDIV: case FG_OP_SQRT: return 4;
    case FG_OP_DOT: return 1 + (long long)in.opnd[1] / 2;
    default: return 1;
    }
}

int fg_mh_interp_launch(fg_engine *e, int iter0, int n_steps, long long *draws, int first_sample_t) {
    const bool force_jit = std::getenv("FG_JIT") && std::atoi(std::getenv("FG_JIT")) == 2;      // experiments: the compiled form even where a stream kernel exists
    if (e->interp_mw_disabled || e->gt || e->tw != FG_WAVE || (e->P.sstream != nullptr && !force_jit)) return FG_E_UNSUPPORTED;
    for (int j = 0; j < e->S; ++j) if (e->prog->site_slot[j] >= e->S) return FG_E_UNSUPPORTED;
    const std::vector<FgIns> &ins = e->prog->ins_fast;
    const int n_ins = e->prog->n_ins;
    if (!e->mhi_setup_done) {
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
            auto waves_on_cu = [&](int w) { return (long long)w * std::max<long long>(1, std::min<long long>(per_cu, (160 * 1024) / (long long)lds_for(w))); };
            if (lds_for(2) * (size_t)per_cu <= 160 * 1024)        // every tile of the CU stays resident: as many waves each as the sixteen slots allow
                while (2 * W <= wcap && 2 * W * per_cu <= 16 && lds_for(2 * W) * (size_t)per_cu <= 160 * 1024) W *= 2;
            else                                                   // a long program (its term rows): the tiles that fit, as many waves as they take
                while (2 * W <= wcap && lds_for(2 * W) <= 160 * 1024 && waves_on_cu(2 * W) <= 16 && waves_on_cu(2 * W) >= waves_on_cu(W)) W *= 2;
        }
        while (W > 1 && lds_for(W) > 160 * 1024) --W;
        if (W < 2) W = 0;                                    // more statements than LDS has term rows: only the compiled kernel's direct mode below can take it
        std::vector<long long> cum(n_stmt + 1, 0);           // work before statement k
        for (int k = 0, i = 0; k < n_stmt; ++k) { long long cs = 0; for (; i < stmt_end[k]; ++i) cs += fg_mhi_ins_cost(ins[i]); cum[k + 1] = cum[k] + cs; }
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
        e->mhi_stmt_end = stmt_end; e->mhi_acc_host = acc;
        e->mhi_W = W; e->mhi_n_stmt = n_stmt; e->mhi_lds = lds_for(std::max(W, 1)); e->mhi_setup_done = true;
        {   // more than eight waves on a CU need the 128-VGPR build
            const long long n_cu = std::max(1, e->n_simd / 4), tiles = (e->C + FG_WAVE - 1) / FG_WAVE, per_cu = (tiles + n_cu - 1) / n_cu;
            const long long resident = std::min<long long>(per_cu, std::max<long long>(1, (160 * 1024) / (long long)e->mhi_lds));
            e->mhi_occ = resident * W > 8 ? 4 : 2;
            if (const char *sp = std::getenv("FG_MH_INTERP_OCC")) e->mhi_occ = std::atoi(sp) <= 2 ? 2 : 4;
        }
    }
    // first choice: the pipelined multi-wave kernel (fg_mh.hip) around this program's generated statements (FG_JIT=2 keeps the
    // statement-segment kernel below, for comparison)
    if (!e->P.sstream && !force_jit && !(std::getenv("FG_MH_NOSTREAM_MW") && std::atoi(std::getenv("FG_MH_NOSTREAM_MW")) == 0)) {
        const int rc = fg_mh_mw_nostream_launch(e, iter0, n_steps, draws, first_sample_t, e->mhi_stmt_end, e->mhi_acc_host, e->d_mhi_site_ins);
        if (rc != FG_E_UNSUPPORTED) return rc;
    }
    FgMhi seg;
    for (int w = 0; w <= FG_MHI_MAX; ++w) { seg.ins_off[w] = e->mhi_ins_off[w]; seg.stmt_off[w] = e->mhi_stmt_off[w]; }
    seg.stmt_acc = e->d_mhi_acc; seg.n_stmt = e->mhi_n_stmt; seg.site_ins = e->d_mhi_site_ins;
    const unsigned tiles = (unsigned)((e->C + e->tw - 1) / e->tw);
    // the model compiled at run time (fg_jit.cpp): the scoring run as eight generated statement segments shared by W = 1, 2, 4 or 8 waves
    if (e->jit_mh_state == 0) {
        e->jit_mh_state = -1;
        const char *sp = std::getenv("FG_JIT");
        // launch shape first (the unit is compiled for ONE register budget): LDS, direct mode, waves per tile, waves per SIMD
        size_t lds = (size_t)((long long)e->S + (e->n_slots - e->S + 1) + e->mhi_n_stmt) * FG_WAVE * sizeof(double);    // site rows, wave 0's temporaries, term rows
        int direct = 0;
        if (lds > 64 * 1024) {                               // too many statements for term rows: the accumulators themselves, plates through a ring of 2 x 32 rows
            lds = (size_t)((long long)e->S + (e->n_slots - e->S + 1) + 64) * FG_WAVE * sizeof(double);
            direct = 1;
        }
        const long long n_cu = std::max(1, e->n_simd / 4), per_cu = ((long long)tiles + n_cu - 1) / n_cu;
        const long long resident = std::max<long long>(1, std::min<long long>(per_cu, (160 * 1024) / (long long)lds));    // tiles a CU holds at once
        int W = 1, forced = 0;
        if (const char *fw = std::getenv("FG_MH_INTERP_WAVES")) forced = std::atoi(fw);
        if (forced > 0) { while (2 * W <= std::min(forced, 8)) W *= 2; }
        else while (2 * W <= 8 && 2 * W * resident <= 16) W *= 2;
        int occ = resident * W > 8 ? 4 : 2;
        if (const char *fo = std::getenv("FG_MH_INTERP_OCC")) occ = std::atoi(fo) <= 2 ? 2 : 4;
        if ((!sp || std::atoi(sp) != 0) && e->prog->ins_fast.size() <= 64000000 && lds <= 64 * 1024) {
            std::vector<long long> cost((size_t)e->prog->n_ins);
            for (int k = 0; k < e->prog->n_ins; ++k) cost[(size_t)k] = fg_mhi_ins_cost(e->prog->ins_fast[(size_t)k]);
            std::vector<double> ctab;
            const std::string src = fg_jit_mh_source(e->prog, cost, occ, &ctab);
            std::vector<char> code;
            if (!src.empty() && src.size() <= (6u << 20) && fg_jit_get_code(src, code, e->jit_log) == FG_OK &&
                hipModuleLoadData(&e->jit_mh_mod, code.data()) == hipSuccess &&
                hipModuleGetFunction(&e->jit_mh_fn[0], e->jit_mh_mod, occ == 4 ? "k_mh_jit_steps_occ4" : "k_mh_jit_steps_occ2") == hipSuccess &&
                fg_jit_bind_tables(e->jit_mh_mod, ctab, &e->d_jit_mh_tab, e->stream) == FG_OK) {
                e->jit_mh_state = 1; e->jit_mh_W = W; e->jit_mh_lds = lds; e->jit_mh_direct = direct;
            } else { (void)hipGetLastError(); if (std::getenv("FG_JIT_VERBOSE")) fprintf(stderr, "fugue_amd: MH kernel not compiled at run time (%s)\n", e->jit_log.c_str()); }
        }
    }
    seg.direct = 0;
    if (e->jit_mh_state == 1) {
        seg.direct = e->jit_mh_direct;
        int n_warmup = e->mh_warmup;
        void *args[] = { &e->P, &e->X, &e->M, &seg, &iter0, &n_steps, &n_warmup, &draws, &first_sample_t };
        HIPCHK(hipModuleLaunchKernel(e->jit_mh_fn[0], tiles, 1, 1, FG_WAVE * e->jit_mh_W, 1, 1, (unsigned)e->jit_mh_lds, e->stream, args, nullptr));
        e->last_mh_kernel = "k_mh_jit_steps W=" + std::to_string(e->jit_mh_W) + (seg.direct ? " (compiled at run time; in-order accumulators, plates shared through an LDS ring)" : " (compiled at run time)");
        return FG_OK;
    }
    if (e->mhi_W < 2) return FG_E_UNSUPPORTED;
    e->last_mh_kernel = "k_mh_interp_mw_steps W=" + std::to_string(e->mhi_W);
#define FG_MHI_LAUNCH(K) do { if (int rc = set_lds(K, e->mhi_lds)) return rc; \
    hipLaunchKernelGGL(K, dim3(tiles), dim3(FG_WAVE * e->mhi_W), e->mhi_lds, e->stream, e->P, e->X, e->M, seg, iter0, n_steps, e->mh_warmup, draws, first_sample_t); } while (0)
    if (e->mhi_occ == 4) FG_MHI_LAUNCH(k_mh_interp_mw_steps_occ4); else FG_MHI_LAUNCH(k_mh_interp_mw_steps_occ2);
#undef FG_MHI_LAUNCH
    HIPCHK(hipGetLastError());
    return FG_OK;
}
