// fg_hmc_interp.hip -- HmcSession::step (hmc.rs:819-919) for programs that need the interpreter (a parameter that is an expression,
// a guard, a select ...: no gradient stream), with a 64-chain tile shared by W waves.
//
// k_hmc_steps gives such a program ONE wave per tile: at 65 536 chains that is one wave per SIMD, so every instruction fetch, LDS
// round trip, scalar branch and out-of-line density call of the interpreter sits exposed between its f64 instructions (a lone
// wave issues a dependent f64 op every ~6 cycles, two or more one per ~4.4: profiles/round1_f64_issue_microbench.txt), and at 8 192
// chains seven SIMDs in eight are idle.  The finite-difference gradient (hmc.rs:304-329) is 2 d independent model runs: here
// wave w evaluates ITS (coordinate, sign) tasks -- a host-side longest-processing-time split by sub-program cost.  The site rows
// of the tile are shared and read-only inside a gradient; what an evaluation writes -- the perturbed coordinate, expression
// temporaries, Categorical tables, select options -- lives in a small block of rows private to the wave (FgRemap, fg_interp.h:
// the interpreter redirects reads of slot i to the wave's `pert` row and offsets every row above the sites), so a tile costs
// S + W (temporaries + 2) rows of LDS, not W copies.  Every evaluation leaves its log-joint in an LDS row; after a workgroup
// barrier wave k mod W forms g_k, kicks p_k and drifts q_k, and a second barrier publishes it.  The sequential parts
// (Hamiltonians, the endpoint score in program order, accept, dual averaging) run on wave 0 exactly as in k_hmc_stream_steps.
// Per coordinate the operations and their order are those of fg_trajectory (fg_engine.hip), so the kernel is bit-identical to
// k_hmc_steps for every W (tests/test_gpu_parity.py::test_hmc_interp_multiwave_is_bit_identical).
#include "fg_engine_internal.h"
#include "fg_cold.h"
#include "fg_jit.h"

#define FG_MWI_MAX 16         /* waves per tile */

struct FgMwi { int off[FG_MWI_MAX + 1]; const int *order; long long *prof; int n_sub_ins, n_fast_ins; };   // wave w owns the tasks order[off[w] .. off[w + 1]): 2 i + sign = evaluate at q_i + h / q_i - h;
                                                                                  // prof: [2 d] cycles of each task (tile 0, first gradient of the launch) or null

template <bool PL>
__device__ __forceinline__ void fg_hmc_interp_mw_body(const FgProgramDev &P, const FgChainCtx &X, const FgHmcDev &H, const FgMwi &seg, int iter0, int n_steps,
                                                      int n_warmup, int welford_on, double *draws, int first_sample_t,
                                                      double *pos_all /*[n][d][C] or null*/, double *info /*[n][4][C] or null*/) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    const int d = P.d, L = H.L;
    const int np = P.n_slots - P.S + 1;                                              // private rows of a wave: temporaries, zero slot, perturbed coordinate
    double *slots = lds + lane;                                                      // site rows [0, S), shared
    double *pl = lds + (long long)(P.S + W * np) * tw + lane;                        // momentum rows, shared
    double *ev_lp = lds + ((long long)(P.S + W * np) + d) * tw + lane;               // log-joint of evaluation (coordinate i, sign): row 2 i + sign
    double *xch = lds + ((long long)(P.S + W * np) + 3 * d) * tw + lane;             // rows: 0 step size, 1 accepted, 2.. per-wave divergence flags
    // PL: the sub-programs and the whole program, staged in LDS behind the rows (instruction fetch = ds_read_b32)
    const FgIns *l_sub = P.sub, *l_fast = P.ins_fast;
    if (PL) {
        uint32_t *dst = (uint32_t *)(lds + ((long long)(P.S + W * np) + 3 * d + 2 + W) * tw);
        const int n_sub_dw = seg.n_sub_ins * 24, n_fast_dw = seg.n_fast_ins * 24;
        for (int k = (int)threadIdx.x; k < n_sub_dw; k += (int)blockDim.x) dst[k] = ((const uint32_t *)P.sub)[k];
        for (int k = (int)threadIdx.x; k < n_fast_dw; k += (int)blockDim.x) dst[n_sub_dw + k] = ((const uint32_t *)P.ins_fast)[k];
        l_sub = (const FgIns *)dst; l_fast = (const FgIns *)(dst + n_sub_dw);
    }
    FgRemap rm;
    rm.pi = 0xffffffffu; rm.n_shared = (uint32_t)P.S; rm.woff = (uint32_t)(wv * np); rm.pert = (uint32_t)(P.n_slots + wv * np);
    const int j0 = seg.off[wv], j1 = seg.off[wv + 1];
    const bool sparse = H.grad_mode != FG_GRAD_FD_DENSE;
    const double *mi = H.use_mass ? H.m_inv + c : nullptr;
    const double *ms = H.use_mass ? H.mass_sqrt + c : nullptr;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    const double h = H.h;
    for (int j = wv; j < P.S; j += W) slots[P.site_slot[j] * tw] = fg_as_double(X.values[(long long)j * X.C + c]);
    slots[(P.n_slots - 1 + rm.woff) * tw] = 0.0;                                     // the wave's always-zero slot
    // wave 0 owns the per-chain sampler state
    double lj = 0.0, eps = 0.0, frozen = 0.0, da_mu = 0.0, da_leb = 0.0, da_hbar = 0.0, asum = 0.0;
    unsigned long long da_m = 0, ndiv = 0;
    if (wv == 0) {
        lj = H.lj[c]; eps = H.eps[c]; frozen = H.frozen[c];
        da_mu = H.da_mu[c]; da_leb = H.da_leb[c]; da_hbar = H.da_hbar[c]; da_m = H.da_m[c];
    }
    for (int t = 0; t < n_steps; ++t) {
        const int iter = iter0 + t;
        const bool warming = iter < n_warmup;
        double h0 = 0.0, u = 0.0;
        // p0 ~ N(0, M) (hmc.rs:436-441): Box-Muller pair j of the chain's (iteration) stream is Philox block j
        const int n_pairs = (d + 1) >> 1;
        for (int j = wv; j < n_pairs; j += W) {
            const FgD2 zz = fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)j, (uint32_t)iter, FG_RNG_HMC);
            const int i = 2 * j;
            pl[i * tw] = zz.a * (ms ? ms[(long long)i * X.C] : 1.0);
            if (i + 1 < d) pl[(i + 1) * tw] = zz.b * (ms ? ms[(long long)(i + 1) * X.C] : 1.0);
        }
        if (wv == 0) {
            double e;
            if (warming) e = eps;
            else {                                             // frozen_or_current: hmc.rs:789-798
                if (frozen == frozen) e = frozen;
                else if (n_warmup > 0) e = fg_cold_exp(da_leb);
                else e = eps;
                frozen = e;
            }
            u = fg_cold_u01_pair(sk0, sk1, gchain, (uint32_t)n_pairs, (uint32_t)iter, FG_RNG_HMC).a;
            xch[0] = e;
        }
        __syncthreads();
        if (wv == 0) h0 = -lj + fg_kinetic(P, pl, tw, mi, X.C);  // hmc.rs:442-443 (all of p0, before any kick)
        __syncthreads();
        const double e = xch[0], hk = 0.5 * e;
        // leapfrog (hmc.rs:353-407) + endpoint score (hmc.rs:283-299) as one flat loop of model evaluations around the interpreter's
        // single call site: this wave's (coordinate, sign) tasks for gradients 0 .. L, then (wave 0) the whole program.  An evaluation
        // leaves its log-joint in row 2 i + sign of `ev_lp`; behind the barrier wave k mod W forms g_k, kicks p_k and drifts q_k.
        const int n_evals = (L + 1) * (j1 - j0) + (wv == 0 ? 1 : 0);
        bool bad = false;
        int s = 0, jj = j0;
        double lj_new = FG_NEG_INF;
        for (int ev = 0; ev < n_evals; ++ev) {
            const bool is_final = (wv == 0) && (ev == n_evals - 1);
            const FgIns *prog = l_fast;
            int n = P.n_ins, task = 0;
            rm.pi = 0xffffffffu;
            if (!is_final) {
                // the perturbed value goes to the wave's private row; the shared q_i is only read (hmc.rs:317-319 restores it: here it never changes)
                task = seg.order[jj];
                const FgCoord cd = P.coord[task >> 1];
                const double orig = slots[cd.slot * tw];
                slots[rm.pert * tw] = (task & 1) ? orig - h : orig + h;
                rm.pi = (uint32_t)cd.slot;
                if (sparse) { prog = l_sub + cd.sub_off; n = cd.sub_n; }
            }
            const bool clocked = seg.prof != nullptr && blockIdx.x == 0 && t == 0 && s == 0 && !is_final;
            long long t0 = 0;
            if (clocked) t0 = (long long)clock64();
            FgAcc3 A = {0.0, 0.0, 0.0};
            fg_exec<FG_MODE_SCORE, false, true, PL>(prog, n, P.pool, slots, tw, A, nullptr, nullptr, 0, false, nullptr, &rm);
            const double tot = fg_total(A);
            if (is_final) { lj_new = tot; break; }
            ev_lp[task * tw] = tot;
            if (clocked && lane == 0) seg.prof[task] = (long long)clock64() - t0;
            if (++jj == j1) {
                jj = j0;
                __syncthreads();                                      // every evaluation of this gradient done, every read of q done
                for (int k = wv; k < d; k += W) {
                    const double g = (ev_lp[2 * k * tw] - ev_lp[(2 * k + 1) * tw]) / (2.0 * h);    // hmc.rs:322
                    bad = bad || !fg_finite(g);
                    double p = pl[k * tw];
                    p += hk * g;                                      // hmc.rs:389 / :400
                    if (s > 0 && s < L) p += hk * g;                  // trailing kick of step s + leading kick of s + 1
                    pl[k * tw] = p;
                    if (s < L) {                                      // q += eps * M^-1 p   (hmc.rs:391-393)
                        const double mk = mi ? mi[(long long)k * X.C] : 1.0;
                        slots[k * tw] += e * mk * p;
                    }
                }
                __syncthreads();
                ++s;
            }
        }
        xch[(2 + wv) * tw] = bad ? 1.0 : 0.0;
        __syncthreads();
        if (wv == 0) {
            bool div = false;
            for (int w = 0; w < W; ++w) div = div || xch[(2 + w) * tw] != 0.0;
            div = div || !fg_finite(lj_new);
            double ap = 0.0; bool acc = false;
            if (!div) {
                const double h_new = -lj_new + fg_kinetic(P, pl, tw, mi, X.C);
                ap = fg_cold_accept_prob(h0, h_new);             // hmc.rs:460
                acc = u < ap;                                    // hmc.rs:461
            }
            if (acc) lj = lj_new;
            xch[tw] = acc ? 1.0 : 0.0;
            asum += ap; ndiv += div ? 1ull : 0ull;
            if (live && info) {                                  // HmcStepInfo: hmc.rs:587-602
                double *r = info + (long long)t * 4 * X.C + c;
                r[0] = acc ? 1.0 : 0.0; r[X.C] = div ? 1.0 : 0.0; r[2 * X.C] = ap; r[3 * X.C] = e;
            }
            if (warming) {                                       // DualAveraging::update: hmc.rs:168-178
                da_m += 1ull;
                const FgD3 r = fg_cold_da_update(da_hbar, da_leb, (double)da_m, da_mu, H.target, ap);
                eps = r.a; da_hbar = r.b; da_leb = r.c;
            }
        }
        __syncthreads();
        const bool acc = xch[tw] != 0.0;
        unsigned long long wn = 0;
        if (warming && welford_on) wn = H.w_n[c] + 1ull;          // every wave reads the old count before wave 0 bumps it below
        for (int k = wv; k < d; k += W) {                         // commit or roll back: coordinate k by wave k mod W
            const long long g = (long long)P.f64_site[k] * X.C + c;
            if (acc) { if (live) X.values[g] = fg_as_i64(slots[k * tw]); }
            else slots[k * tw] = fg_as_double(X.values[g]);
            const double x = slots[k * tw];
            if (live && pos_all) pos_all[((long long)t * d + k) * X.C + c] = x;
            if (warming) {
                if (welford_on) {                                 // Welford::push: hmc.rs:202-211
                    const long long gi = (long long)k * X.C + c;
                    const double n = (double)wn;
                    double mean = H.w_mean[gi];
                    const double delta = x - mean;
                    mean += delta / n;
                    const double delta2 = x - mean;
                    if (live) { H.w_mean[gi] = mean; H.w_m2[gi] += delta * delta2; }
                }
            } else if (draws && live) draws[((long long)(t - first_sample_t) * d + k) * X.C + c] = x;   // hmc.rs:577-582
        }
        if (warming && welford_on) {
            __syncthreads();                                      // all waves hold the old count
            if (wv == 0 && live) H.w_n[c] = wn;
        }
    }
    if (wv == 0 && live) {
        H.lj[c] = lj; H.eps[c] = eps; H.frozen[c] = frozen;
        H.da_mu[c] = da_mu; H.da_leb[c] = da_leb; H.da_hbar[c] = da_hbar; H.da_m[c] = da_m;
        H.alpha_sum[c] += asum; H.n_div[c] += ndiv;
    }
}

// OCC = waves per SIMD the register budget allows (2: 256 VGPRs, no spills; 4: 128 with the interpreter's cold paths spilling -- the
// faster one wherever enough waves exist: tools/bench_interp_mw.py); PL: program staged in LDS
#define FG_MWI_KERNEL(OCC, PL, NAME) \
__global__ __attribute__((amdgpu_waves_per_eu(OCC, OCC))) __launch_bounds__(FG_WAVE * 4 * OCC) \
void NAME(FgProgramDev P, FgChainCtx X, FgHmcDev H, FgMwi seg, int iter0, int n_steps, int n_warmup, int welford_on, double *draws, \
          int first_sample_t, double *pos_all, double *info) { \
    fg_hmc_interp_mw_body<PL>(P, X, H, seg, iter0, n_steps, n_warmup, welford_on, draws, first_sample_t, pos_all, info); }
FG_MWI_KERNEL(2, false, k_hmc_interp_mw_steps_occ2)
FG_MWI_KERNEL(4, false, k_hmc_interp_mw_steps_occ4)
FG_MWI_KERNEL(2, true, k_hmc_interp_mw_steps_lds_occ2)
FG_MWI_KERNEL(4, true, k_hmc_interp_mw_steps_lds_occ4)

// cost of one interpreted instruction in the split (relative: an out-of-line density with its logs / lgammas against an add)
static long long mwi_ins_cost(const FgIns &in) {
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

// longest-processing-time split of the 2 d tasks (task 2 k + sign costs cost[k]) over W waves; returns the makespan.
// plus_only: the "-" tasks cost nothing and are left out of the bins (FG_GRAD_ANALYTIC in the compiled kernel: one task per coordinate)
static long long mwi_split(const std::vector<long long> &cost, int W, std::vector<std::vector<int>> *bins_out, bool plus_only = false) {
    const int n_tasks = 2 * (int)cost.size();
    std::vector<int> by;
    for (int k = 0; k < n_tasks; ++k) if (!plus_only || !(k & 1)) by.push_back(k);
    std::stable_sort(by.begin(), by.end(), [&](int a, int b) { return cost[a >> 1] > cost[b >> 1]; });
    std::vector<std::vector<int>> bins(W);
    std::vector<long long> load(W, 0);
    for (int k : by) {
        int best = 0;
        for (int w = 1; w < W; ++w) if (load[w] < load[best]) best = w;
        bins[best].push_back(k); load[best] += cost[k >> 1];
    }
    int lightest = 0;                                           // wave 0 also runs the endpoint score: it gets the lightest bin
    for (int w = 1; w < W; ++w) if (load[w] < load[lightest]) lightest = w;
    std::swap(bins[0], bins[lightest]);
    if (bins_out) *bins_out = bins;
    return *std::max_element(load.begin(), load.end());
}

int fg_hmc_interp_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info) {
    if (e->interp_mw_disabled || e->gt || e->tw != FG_WAVE || e->d < 2) return FG_E_UNSUPPORTED;
    const unsigned tiles = (unsigned)((e->C + e->tw - 1) / e->tw);
    const bool sparse = e->cfg.grad_mode != FG_GRAD_FD_DENSE;
    for (int j = 0; j < e->S; ++j) if (e->prog->site_slot[j] >= e->S) return FG_E_UNSUPPORTED;    // site rows first, the private rows above them
    for (int k = 0; k < e->d; ++k) if (e->prog->coord[k].slot != k) return FG_E_UNSUPPORTED;
    // the program in LDS (instruction fetch by ds_read_b32: -5 ... -10 % time) when that does not cost a resident tile: with two or more
    // tiles per CU the workgroup must stay under half the LDS, a CU's only tile may take all of it
    const size_t prog_bytes = (e->prog->sub.size() + e->prog->ins_fast.size()) * sizeof(FgIns);
    const long long n_cu = std::max(1, e->n_simd / 4), per_cu = ((long long)tiles + n_cu - 1) / n_cu;
    auto rows_for = [&](int W) { return (size_t)((long long)e->S + (long long)W * (e->n_slots - e->S + 1) + 3LL * e->d + 2 + W) * FG_WAVE * sizeof(double); };
    auto lds_for = rows_for;
    int occ = 4;                                               // measured (tools/bench_interp_mw.py): 128 VGPRs with the cold paths spilling beats 168 and 198 -- the waves hide more than the spills cost
    if (const char *sp = std::getenv("FG_HMC_INTERP_OCC")) occ = std::atoi(sp) <= 2 ? 2 : 4;
    const int wmax = 4 * occ;                                  // a workgroup's waves must fit one CU at that occupancy
    const int wcap = std::min(wmax, 2 * e->d);
    const int n_tasks = 2 * e->d;
    int forced = e->mw_override;
    if (const char *sp = std::getenv("FG_HMC_INTERP_WAVES")) forced = std::atoi(sp);
    if (!e->d_mwi_order) {
        HIPCHK(hipMalloc((void **)&e->d_mwi_order, (size_t)2 * n_tasks * sizeof(int)));
        HIPCHK(hipMalloc((void **)&e->d_mwi_prof, (size_t)n_tasks * sizeof(long long)));
        HIPCHK(hipMemsetAsync(e->d_mwi_prof, 0, (size_t)n_tasks * sizeof(long long), e->stream));
    }
    const bool debug = std::getenv("FG_HMC_INTERP_DEBUG") != nullptr;
    if (debug && e->mwi_calibrated == 1) {                      // FG_HMC_INTERP_DEBUG: the cycles the previous launch clocked per task, beside the static costs
        std::vector<long long> prof(n_tasks, 0);
        HIPCHK(hipMemcpyAsync(prof.data(), e->d_mwi_prof, (size_t)n_tasks * sizeof(long long), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        for (int k = 0; k < e->d; ++k) fprintf(stderr, "coord %d static %lld measured %lld %lld sub_n %d\n", k, e->mwi_cost[k], prof[2 * k], prof[2 * k + 1], e->prog->coord[k].sub_n);
        e->mwi_calibrated = 2;
    }
    if (e->mwi_sparse != (int)sparse || e->mwi_cost.empty() || e->mwi_W <= 0) {
        // task costs: static weights per interpreted instruction (they track the measured cycles within ~30 %: FG_HMC_INTERP_DEBUG)
        e->mwi_cost.assign(e->d, 1);
        if (sparse)
            for (int k = 0; k < e->d; ++k) {
                long long cs = 0;
                for (int q = 0; q < e->prog->coord[k].sub_n; ++q) cs += mwi_ins_cost(e->prog->sub[e->prog->coord[k].sub_off + q]);
                e->mwi_cost[k] = std::max(1LL, cs);
            }
        // waves per tile: eight (two tiles fill a CU's sixteen wave slots, one tile still gives every SIMD two waves), a power of two
        // (measured: 6 and 12 lose to 4 and 8 on every model), never more than the 2 d tasks
        int W = 2;
        if (forced > 0) W = std::max(2, std::min(forced, wcap));
        else while (2 * W <= std::min(8, wcap) && lds_for(2 * W) <= 160 * 1024) W *= 2;
        while (W > 1 && lds_for(W) > 160 * 1024) --W;
        if (W < 2) return FG_E_UNSUPPORTED;
        std::vector<std::vector<int>> bins;
        mwi_split(e->mwi_cost, W, &bins);
        std::vector<int> order;
        e->mwi_off.assign(FG_MWI_MAX + 1, n_tasks);
        for (int w = 0; w < W; ++w) {
            e->mwi_off[w] = (int)order.size();
            std::sort(bins[w].begin(), bins[w].end());
            order.insert(order.end(), bins[w].begin(), bins[w].end());
        }
        HIPCHK(hipMemcpyAsync(e->d_mwi_order, order.data(), (size_t)n_tasks * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));                // `order` is a local
        e->mwi_W = W; e->mwi_sparse = (int)sparse; e->mwi_calibrated = 0;
    }
    const int W = e->mwi_W;
    FgMwi seg;
    for (int w = 0; w <= FG_MWI_MAX; ++w) seg.off[w] = e->mwi_off[w];
    seg.order = e->d_mwi_order;
    seg.prof = (debug && e->mwi_calibrated == 0) ? e->d_mwi_prof : nullptr;
    if (debug && e->mwi_calibrated == 0) e->mwi_calibrated = 1;
    seg.n_sub_ins = (int)e->prog->sub.size(); seg.n_fast_ins = (int)e->prog->ins_fast.size();
    bool pl = rows_for(W) + prog_bytes <= (size_t)(per_cu >= 2 ? 80 : 160) * 1024;
    if (const char *sp = std::getenv("FG_HMC_INTERP_LDSPROG")) pl = std::atoi(sp) != 0 && rows_for(W) + prog_bytes <= 160 * 1024;
    const size_t lds = rows_for(W) + (pl ? prog_bytes : 0);
#define FG_MWI_LAUNCH(K) do { if (int rc = set_lds(K, lds)) return rc; \
    hipLaunchKernelGGL(K, dim3(tiles), dim3(FG_WAVE * W), lds, e->stream, e->P, e->X, e->H, seg, iter0, n, e->n_warmup, welford_on, \
                       draws, first_sample_t, pos_all, info); } while (0)
    if (pl) { if (occ == 4) FG_MWI_LAUNCH(k_hmc_interp_mw_steps_lds_occ4); else FG_MWI_LAUNCH(k_hmc_interp_mw_steps_lds_occ2); }
    else { if (occ == 4) FG_MWI_LAUNCH(k_hmc_interp_mw_steps_occ4); else FG_MWI_LAUNCH(k_hmc_interp_mw_steps_occ2); }
#undef FG_MWI_LAUNCH
    HIPCHK(hipGetLastError());
    e->last_hmc_kernel = "k_hmc_interp_mw_steps W=" + std::to_string(W) + (occ != 4 ? " occ=" + std::to_string(occ) : std::string()) + (pl ? std::string() : std::string(" (program in global memory)"));
    return FG_OK;
}

// ---- the same kernel around a model compiled at run time (fg_jit.cpp, fg_hmc_jit_body.h) --------------------------------------
struct FgJitSeg { int off[FG_MWI_MAX + 1]; const int *order; int baked; };   // (fg_hmc_jit_body.h's)

// waves per tile of the compiled HMC kernels and the split of the sparse finite difference's 2 d tasks over them -- a function of the engine alone
// (program, chain count, FG_HMC_WAVES / FG_HMC_INTERP_WAVES / FG_HMC_JIT_OCC at the time): the unit is generated BEHIND it (fg_jit_wave_tasks)
static int jit_sparse_split(fg_engine *e, unsigned tiles, std::vector<long long> &cost, std::vector<std::vector<int>> &bins, std::vector<std::vector<int>> *cbins = nullptr) {
    const int n_tasks = 2 * e->d;
    cost.assign(e->d, 1);
    for (int k = 0; k < e->d; ++k) {
        long long cs = 0;
        for (int q = 0; q < e->prog->coord[k].sub_n; ++q) cs += mwi_ins_cost(e->prog->sub[e->prog->coord[k].sub_off + q]);
        cost[k] = std::max(1LL, cs);
    }
    int forced = e->mw_override;
    if (const char *sp = std::getenv("FG_HMC_INTERP_WAVES")) forced = std::atoi(sp);
    int jocc = 4;
    if (const char *oc = std::getenv("FG_HMC_JIT_OCC")) { const int o = std::atoi(oc); if (o >= 2 && o <= 4) jocc = o; }
    const int wcap = std::min(std::min(FG_MWI_MAX, 4 * jocc), n_tasks);
    int W = 1;
    const long long n_cu = std::max(1, e->n_simd / 4);
    if (forced > 0) W = std::max(1, std::min(forced, wcap));
    else if ((long long)tiles <= n_cu) W = wcap;             // a CU has at most one tile: a wave per task (logistic regression, 8 192 chains: W = 6 beats 4 by 45 %)
    else {
        // several tiles per CU: sixteen waves per CU is all that is ever resident (128 VGPRs), so four tiles of four waves where the LDS holds four tiles and a tile has at most sixteen tasks --
        // fewer, longer task lists per wave and half the waves at every barrier (reference_model(8) at 65 536 chains 2.32e10 -> 2.81e10 leapfrog-steps/s,
        // hier 1.75e10 -> 2.05e10, mixture +6 %) -- and eight waves where it holds two or three (reference_model(20): 1.02e10 with four, 1.12e10 with eight;
        // reference_model(32) 6.1e9 / 7.4e9): profiles/round4_hmc_jit_waves.txt
        const long long lds8 = ((long long)e->S + 3LL * e->d + 2 + 8) * FG_WAVE * (long long)sizeof(double);
        const long long resident = std::min<long long>((160 * 1024) / std::max<long long>(1, lds8), ((long long)tiles + n_cu - 1) / n_cu);
        const int target = (resident >= 4 && n_tasks <= 16) ? 4 : 8;     // (alldists, 24 heavy tasks, four tiles per CU: 9.3e8 with eight waves, 8.7e8 with four)
        while (2 * W <= std::min(target, wcap)) W *= 2;
        if (std::getenv("FG_JIT_VERBOSE")) { long long tot = 0; for (long long c : cost) tot += 2 * c; fprintf(stderr, "fugue_amd: compiled HMC unit: d %d, task cost %lld, resident %lld, W %d\n", e->d, tot, resident, W); }
    }
    const long long span_tasks = mwi_split(cost, W, &bins);
    for (int w = 0; w < W; ++w) std::sort(bins[w].begin(), bins[w].end());
    if (cbins) {
        // whole coordinates per wave (the one-barrier gradient of fg_jit_wave_grad): both evaluations of a coordinate on one wave.  Taken where the coarser
        // split stretches the longest wave by less than a barrier costs; FG_JIT_FUSED=0 / 1 forces.
        std::vector<std::vector<int>> pb;
        mwi_split(cost, W, &pb, true);
        // the stretch in the units the rule below was measured in: the split's costs are the interpreter's, where a general density is 10 - 16 against a fast
        // Normal's 3; compiled, the ratio is about twice that
        std::vector<long long> tcost((size_t)e->d, 1);
        for (int k = 0; k < e->d; ++k) {
            long long cs = 0;
            for (int q = 0; q < e->prog->coord[k].sub_n; ++q) { const FgIns &in = e->prog->sub[e->prog->coord[k].sub_off + q]; cs += mwi_ins_cost(in) * ((FG_INS_OPCODE(in.op) < 17u) ? 2 : 1); }
            tcost[(size_t)k] = std::max(1LL, cs);
        }
        long long span_tasks_t = 0, span_coords_t = 0;
        for (int w = 0; w < W; ++w) {
            long long a = 0, b = 0;
            for (int t : bins[(size_t)w]) a += tcost[(size_t)(t >> 1)];
            for (int t : pb[(size_t)w]) b += 2 * tcost[(size_t)(t >> 1)];
            span_tasks_t = std::max(span_tasks_t, a); span_coords_t = std::max(span_coords_t, b);
        }
        const long long span_coords = span_coords_t; (void)span_tasks;
        cbins->assign((size_t)W, std::vector<int>());
        for (int w = 0; w < W; ++w) { for (int t : pb[w]) (*cbins)[(size_t)w].push_back(t >> 1); std::sort((*cbins)[(size_t)w].begin(), (*cbins)[(size_t)w].end()); }
        const long long lds1 = ((long long)e->S + 3LL * e->d + 2 + W) * FG_WAVE * (long long)sizeof(double), lds2 = lds1 + (long long)e->S * FG_WAVE * (long long)sizeof(double);
        const long long per_cu = ((long long)tiles + n_cu - 1) / n_cu;
        // (a barrier is worth about 96 such units of the longest wave: reference_model(8) 36 -> 48 units +8 %, reference_model(20) 60 -> 120 +7 % / +15 % at 8 192
        // chains, reference_model(32) 96 -> 192 +8 %, hier_scale 74 -> 148 of the split's units (general densities) -8 %, logistic regression 465 -> 930 -29 %; the second copy may cost a resident tile but not the last but one:
        // reference_model(32), two tiles -> one, -14 % -- profiles/round4_hmc_jit_one_barrier.txt)
        const long long t1 = std::min<long long>((160 * 1024) / lds1, per_cu), t2 = lds2 <= 160 * 1024 ? std::min<long long>((160 * 1024) / lds2, per_cu) : 0;
        bool ok = t2 >= std::min<long long>(2, t1) && span_coords_t - span_tasks_t <= 96;
        if (std::getenv("FG_JIT_VERBOSE")) fprintf(stderr, "fugue_amd: compiled HMC unit: W %d, longest wave %lld (tasks) / %lld (whole coordinates), tiles per CU %lld / %lld\n", W, span_tasks_t, span_coords,
                                                   std::min<long long>((160 * 1024) / lds1, per_cu), std::min<long long>((160 * 1024) / lds2, per_cu));
        if (const char *fv = std::getenv("FG_JIT_FUSED")) ok = std::atoi(fv) != 0 && lds2 <= 160 * 1024;
        if (!ok) cbins->clear();
    }
    return W;
}

// the dense mode's one-barrier gradient: every (coordinate, sign) task is the whole program, so whole coordinates per wave cost nothing exactly when
// 2 ceil(d / W) = ceil(2 d / W); the second copy of the site rows under the same LDS rule as the sparse form's
static void jit_dense_coord_split(fg_engine *e, unsigned tiles, int W, std::vector<std::vector<int>> &cbins) {
    cbins.clear();
    const int d = e->d;
    if (W < 1 || 2 * ((d + W - 1) / W) != (2 * d + W - 1) / W) return;
    const long long n_cu = std::max(1, e->n_simd / 4), per_cu = ((long long)tiles + n_cu - 1) / n_cu;
    const long long lds1 = ((long long)e->S + 3LL * d + 2 + W) * FG_WAVE * (long long)sizeof(double), lds2 = lds1 + (long long)e->S * FG_WAVE * (long long)sizeof(double);
    const long long t1 = std::min<long long>((160 * 1024) / lds1, per_cu), t2 = lds2 <= 160 * 1024 ? std::min<long long>((160 * 1024) / lds2, per_cu) : 0;
    bool ok = t2 >= std::min<long long>(2, t1);
    if (const char *fv = std::getenv("FG_JIT_FUSED")) ok = std::atoi(fv) != 0 && lds2 <= 160 * 1024;
    if (!ok) return;
    cbins.assign((size_t)W, std::vector<int>());
    for (int k = 0; k < d; ++k) cbins[(size_t)(k % W)].push_back(k);
}

// the program's compiled module (once per engine): HMC transitions, the step-size search, adaptive_smc's rejuvenation move
static int jit_hmc_module(fg_engine *e) {
    if (e->jit_state < 0 || e->gt || e->tw != FG_WAVE || e->d < 1) return FG_E_UNSUPPORTED;
    if (e->jit_state == 0) {
        e->jit_state = -1;
        if (const char *sp = std::getenv("FG_JIT")) if (std::atoi(sp) == 0) return FG_E_UNSUPPORTED;
        for (int j = 0; j < e->S; ++j) if (e->prog->site_slot[j] >= e->S) return FG_E_UNSUPPORTED;
        for (int k = 0; k < e->d; ++k) if (e->prog->coord[k].slot != k) return FG_E_UNSUPPORTED;
        if (e->prog->sub.size() + e->prog->ins_fast.size() > 64000000) return FG_E_UNSUPPORTED;
        std::vector<double> ctab;
        bool has_ad = false, has_dense = false;
        std::vector<long long> cost0;
        e->jit_baked_bins.clear();
        e->jit_baked_cbins.clear();
        e->jit_baked_cbins_dense.clear();
        if (!(std::getenv("FG_JIT_TASKS") && std::atoi(std::getenv("FG_JIT_TASKS")) == 0)) {
            const unsigned tiles0 = (unsigned)((e->C + e->tw - 1) / e->tw);
            const int W0 = jit_sparse_split(e, tiles0, cost0, e->jit_baked_bins, &e->jit_baked_cbins);
            jit_dense_coord_split(e, tiles0, W0, e->jit_baked_cbins_dense);
        }
        const std::string src = fg_jit_hmc_source(e->prog, &ctab, &has_ad, &has_dense, e->jit_baked_bins.empty() ? nullptr : &e->jit_baked_bins, e->jit_baked_cbins.empty() ? nullptr : &e->jit_baked_cbins,
                                                  e->jit_baked_cbins_dense.empty() ? nullptr : &e->jit_baked_cbins_dense);
        if (src.empty() || src.size() > (6u << 20)) return FG_E_UNSUPPORTED;                            // plates roll into loops; what stays straight-line must stay compilable in seconds
        std::vector<char> code;
        const int rc = fg_jit_get_code(src, code, e->jit_log);
        if (rc != FG_OK) {
            if (std::getenv("FG_JIT_VERBOSE")) fprintf(stderr, "fugue_amd: run-time compilation unavailable (%s): the interpreter kernels take this program\n", e->jit_log.c_str());
            return FG_E_UNSUPPORTED;
        }
        if (hipModuleLoadData(&e->jit_mod, code.data()) != hipSuccess || hipModuleGetFunction(&e->jit_fn, e->jit_mod, "k_hmc_jit_steps") != hipSuccess ||
            hipModuleGetFunction(&e->jit_fn_eps, e->jit_mod, "k_hmc_jit_find_eps") != hipSuccess ||
            hipModuleGetFunction(&e->jit_fn_rejuv, e->jit_mod, "k_smc_jit_rejuv") != hipSuccess ||
            fg_jit_bind_tables(e->jit_mod, ctab, &e->d_jit_tab, e->stream) != FG_OK) {
            e->jit_log = "hipModuleLoadData / hipModuleGetFunction / table upload failed"; (void)hipGetLastError();
            return FG_E_UNSUPPORTED;
        }
        // (optional entry points: a unit whose generic program the generator does not cover has no k_prior_jit)
        if (hipModuleGetFunction(&e->jit_fn_prior, e->jit_mod, "k_prior_jit") != hipSuccess) { e->jit_fn_prior = nullptr; (void)hipGetLastError(); }
        if (hipModuleGetFunction(&e->jit_fn_lj, e->jit_mod, "k_log_joint_jit") != hipSuccess) { e->jit_fn_lj = nullptr; (void)hipGetLastError(); }
        e->jit_state = 1; e->jit_has_ad = has_ad; e->jit_has_dense = has_dense;
    }
    return FG_OK;
}

// run(PriorHandler) / run(ScoreGivenTrace) of every chain through the compiled model (k_prior_jit / k_log_joint_jit): FG_E_UNSUPPORTED when
// there is none (the caller takes the interpreter kernels).  `compile`: build the unit now if it is not there yet (callers for which the
// draw is a visible part of the work: adaptive_smc, sessions of programs that will step through the unit anyway).
static int jit_tile_launch(fg_engine *e, hipFunction_t fn, void **args) {
    const size_t tile = (size_t)e->S * FG_WAVE * sizeof(double);
    if (!fn || tile == 0 || tile > 64 * 1024) return FG_E_UNSUPPORTED;
    const int wpb = (int)std::max<size_t>(1, std::min<size_t>(4, (64 * 1024) / tile));
    const unsigned nblk = (unsigned)((e->C + (long long)FG_WAVE * wpb - 1) / ((long long)FG_WAVE * wpb));
    HIPCHK(hipModuleLaunchKernel(fn, nblk, 1, 1, FG_WAVE * wpb, 1, 1, (unsigned)(tile * wpb), e->stream, args, nullptr));
    return FG_OK;
}
int fg_jit_prior_launch(fg_engine *e, uint32_t iteration, uint32_t purpose, double *d_acc, double *d_lj, bool compile) {
    if (e->jit_state != 1 && !(compile && e->jit_state == 0)) return FG_E_UNSUPPORTED;
    if (int rc = jit_hmc_module(e)) {
        if (std::getenv("FG_JIT_VERBOSE")) fprintf(stderr, "fugue_amd: no compiled prior draw (state %d, d %d, S %d, tw %d, gt %d: %s)\n", e->jit_state, e->d, e->S, e->tw, (int)(e->gt != 0), e->jit_log.c_str());
        return rc;
    }
    if (!e->jit_fn_prior && std::getenv("FG_JIT_VERBOSE")) fprintf(stderr, "fugue_amd: the compiled unit has no k_prior_jit\n");
    void *args[] = { &e->P, &e->X, &iteration, &purpose, &d_acc, &d_lj };
    return jit_tile_launch(e, e->jit_fn_prior, args);
}
int fg_jit_log_joint_launch(fg_engine *e, double *d_acc, double *d_lj, bool compile) {
    if (e->jit_state != 1 && !(compile && e->jit_state == 0)) return FG_E_UNSUPPORTED;
    if (int rc = jit_hmc_module(e)) return rc;
    void *args[] = { &e->P, &e->X, &d_acc, &d_lj };
    return jit_tile_launch(e, e->jit_fn_lj, args);
}

// does the compiled module hold the analytic gradient of this program (FG_GRAD_ANALYTIC beyond Normal force terms)?  Compiles it if need be.
bool fg_hmc_jit_has_ad(fg_engine *e) { return jit_hmc_module(e) == FG_OK && e->jit_has_ad; }

// adaptive_smc's rejuvenation move of a program without a score stream through the compiled model (k_smc_jit_rejuv, fg_hmc_jit_body.h);
// FG_E_UNSUPPORTED: the interpreter kernel k_smc_rejuv<-1> takes it.  n_blk_out: blocks launched (rows of M.blk that k_smc_adapt adds).
int fg_smc_jit_rejuv_launch(fg_engine *e, const FgSmcDev &M, const FgSmcScalars *st, uint32_t move_id, unsigned *n_blk_out, const long long *vsrc, double *pmax) {
    if (e->S > FG_SMC_HIST) return FG_E_UNSUPPORTED;
    if (int rc = jit_hmc_module(e)) return rc;
    const size_t tile = (size_t)e->S * FG_WAVE * sizeof(double);
    if (tile > 150 * 1024) return FG_E_UNSUPPORTED;
    // tiles (waves) per block: four; sixteen for a program of a few sites (fewer blocks, fewer rows of counts and block maxima: 0.39 -> 0.375 ms
    // per 1 048 576-particle run of the one-site model)
    const int wpb = (int)std::max<size_t>(1, std::min<size_t>(16 * tile <= 16 * 1024 ? 16 : 4, (150 * 1024) / tile));
    const size_t lds = tile * wpb;
    if (lds > 64 * 1024 && !e->jit_rejuv_attr) {
        if (hipFuncSetAttribute((const void *)e->jit_fn_rejuv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); return FG_E_UNSUPPORTED; }
        e->jit_rejuv_attr = true;
    }
    const unsigned nblk = (unsigned)((e->C + (long long)FG_WAVE * wpb - 1) / ((long long)FG_WAVE * wpb));
    FgSmcDev Mv = M;
    void *args[] = { &e->P, &e->X, &Mv, &st, &move_id, &vsrc, &pmax };
    HIPCHK(hipModuleLaunchKernel(e->jit_fn_rejuv, nblk, 1, 1, FG_WAVE * wpb, 1, 1, (unsigned)lds, e->stream, args, nullptr));
    if (n_blk_out) *n_blk_out = nblk;
    return FG_OK;
}

// ... and the task split of its HMC kernels; FG_E_UNSUPPORTED when there is none
static int jit_hmc_prepare(fg_engine *e, unsigned tiles) {
    if (int rc = jit_hmc_module(e)) return rc;
    const bool dense = e->cfg.grad_mode == FG_GRAD_FD_DENSE;
    if (dense && !e->jit_has_dense) return FG_E_UNSUPPORTED;      // (d copies of the program were too much to compile: the interpreter kernels)
    const int n_tasks = 2 * e->d;
    {   // LDS: S site rows + d momentum rows + 2 d evaluation rows + exchange rows; beyond 64 KB the module's functions need the attribute
        const size_t lds_max = (size_t)((long long)e->S + 3LL * e->d + 2 + FG_MWI_MAX) * FG_WAVE * sizeof(double);
        if (lds_max > 160 * 1024) return FG_E_UNSUPPORTED;
        if (lds_max + (size_t)e->S * FG_WAVE * sizeof(double) > 64 * 1024 && !e->jit_lds_attr) {      // (+ the second copy of the site rows of the one-barrier gradient)
            if (hipFuncSetAttribute((const void *)e->jit_fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void *)e->jit_fn_eps, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { (void)hipGetLastError(); e->jit_state = -1; return FG_E_UNSUPPORTED; }
            e->jit_lds_attr = true;
        }
    }
    if (!e->d_mwi_order) {
        HIPCHK(hipMalloc((void **)&e->d_mwi_order, (size_t)2 * n_tasks * sizeof(int)));      // (the second half: the split of the analytic mode)
        HIPCHK(hipMalloc((void **)&e->d_mwi_prof, (size_t)n_tasks * sizeof(long long)));
    }
    const int split_key = dense ? 4 : 2;                   // (2: the split of the compiled kernel; 4: its dense mode -- every task is the whole program)
    if (e->mwi_sparse != split_key || e->mwi_W <= 0) {
        std::vector<std::vector<int>> bins, cbins;
        int W = jit_sparse_split(e, tiles, e->mwi_cost, bins, &cbins);
        e->mwi_fused = !dense && !cbins.empty() && cbins == e->jit_baked_cbins;
        if (dense) { std::vector<std::vector<int>> dc; jit_dense_coord_split(e, tiles, W, dc); e->mwi_fused = !dc.empty() && dc == e->jit_baked_cbins_dense; }
        if (dense) { e->mwi_cost.assign(e->d, 1); mwi_split(e->mwi_cost, W, &bins); }      // (every task is the whole program)
        e->mwi_baked = !dense && !e->jit_baked_bins.empty() && (int)e->jit_baked_bins.size() == W;
        for (int w = 0; w < W && e->mwi_baked; ++w) { std::vector<int> b = bins[w]; std::sort(b.begin(), b.end()); e->mwi_baked = b == e->jit_baked_bins[w]; }
        std::vector<int> order;
        e->mwi_off.assign(FG_MWI_MAX + 1, n_tasks);
        for (int w = 0; w < W; ++w) {
            e->mwi_off[w] = (int)order.size();
            std::sort(bins[w].begin(), bins[w].end());
            order.insert(order.end(), bins[w].begin(), bins[w].end());
        }
        // FG_GRAD_ANALYTIC: one derivative task per coordinate, dealt over the same W waves (the step-size search keeps the split above)
        mwi_split(e->mwi_cost, W, &bins, true);
        e->mwi_off_an.assign(FG_MWI_MAX + 1, n_tasks + e->d);
        order.resize(n_tasks);
        for (int w = 0; w < W; ++w) {
            e->mwi_off_an[w] = (int)order.size();
            std::sort(bins[w].begin(), bins[w].end());
            order.insert(order.end(), bins[w].begin(), bins[w].end());
        }
        HIPCHK(hipMemcpyAsync(e->d_mwi_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        e->mwi_W = W; e->mwi_sparse = split_key;
    }
    return FG_OK;
}

// find_reasonable_epsilon through the compiled kernel (k_hmc_jit_find_eps)
int fg_hmc_jit_find_eps(fg_engine *e, uint32_t instance, int injected, double *d_eps_out) {
    const unsigned tiles = (unsigned)((e->C + e->tw - 1) / e->tw);
    // the choice of kernel must be the one fg_hmc_step will make: only programs the compiled form takes by default or by force
    if (int rc = jit_hmc_prepare(e, tiles)) return rc;
    const int W = e->mwi_W;
    FgJitSeg seg;
    for (int w = 0; w <= FG_MWI_MAX; ++w) seg.off[w] = e->mwi_off[w];
    seg.order = e->d_mwi_order; seg.baked = 0;
    const size_t lds = (size_t)((long long)e->S + 3LL * e->d + 2 + W) * FG_WAVE * sizeof(double);
    void *args[] = { &e->P, &e->X, &e->H, &seg, &instance, &injected, &d_eps_out };
    HIPCHK(hipModuleLaunchKernel(e->jit_fn_eps, tiles, 1, 1, FG_WAVE * W, 1, 1, (unsigned)lds, e->stream, args, nullptr));
    return FG_OK;
}

int fg_hmc_jit_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info) {
    const unsigned tiles = (unsigned)((e->C + e->tw - 1) / e->tw);
    if (int rc = jit_hmc_prepare(e, tiles)) return rc;
    auto lds_for = [&](int W) { return (size_t)((long long)e->S + 3LL * e->d + 2 + W) * FG_WAVE * sizeof(double); };
    const int W = e->mwi_W;
    FgJitSeg seg;
    for (int w = 0; w <= FG_MWI_MAX; ++w) seg.off[w] = e->mwi_off[w];
    seg.order = e->d_mwi_order;
    seg.baked = (e->mwi_baked && e->cfg.grad_mode == FG_GRAD_FD_SPARSE) ? 1 : 0;      // the unit holds this very split as straight-line code (fg_jit_wave_tasks)
    const bool fused = e->mwi_fused && (e->cfg.grad_mode == FG_GRAD_FD_SPARSE || e->cfg.grad_mode == FG_GRAD_FD_DENSE);   // ... or whole coordinates per wave (fg_jit_wave_grad / _dense): a second copy of the site rows
    if (fused) seg.baked = e->cfg.grad_mode == FG_GRAD_FD_DENSE ? 3 : 2;
    if (e->cfg.grad_mode == FG_GRAD_ANALYTIC && e->jit_has_ad) for (int w = 0; w <= FG_MWI_MAX; ++w) seg.off[w] = e->mwi_off_an[w];
    int n_warmup = e->n_warmup;
    void *args[] = { &e->P, &e->X, &e->H, &seg, &iter0, &n, &n_warmup, &welford_on, &draws, &first_sample_t, &pos_all, &info };
    const size_t lds = lds_for(W) + (fused ? (size_t)e->S * FG_WAVE * sizeof(double) : 0);
    HIPCHK(hipModuleLaunchKernel(e->jit_fn, tiles, 1, 1, FG_WAVE * W, 1, 1, (unsigned)lds, e->stream, args, nullptr));
    e->last_hmc_kernel = "k_hmc_jit_steps W=" + std::to_string(W) + (e->cfg.grad_mode == FG_GRAD_FD_DENSE ? (fused ? " (dense; compiled at run time, one barrier per gradient)" : " (dense; compiled at run time)") : fused ? " (compiled at run time, one barrier per gradient)" : " (compiled at run time)");
    return FG_OK;
}
