// fg_hmc_interp.hip -- HmcSession::step (hmc.rs:819-919) for programs that need the interpreter (a parameter that is an expression,
// a guard, a select ...: no gradient stream), with a 64-chain tile shared by W waves.
//
// k_hmc_steps gives such a program ONE wave per tile: at 65 536 chains that is one wave per SIMD, so every instruction fetch, LDS
// round trip, scalar branch and out-of-line density call of the interpreter sits exposed between its f64 instructions (a lone
// wave issues a dependent f64 op every ~6 cycles, two or more one per ~4.4: profiles/round1_f64_issue_microbench.txt), and at 8 192
// chains seven SIMDs in eight are idle.  The finite-difference gradient (hmc.rs:304-329) is d independent pairs of model runs:
// here wave w evaluates the pairs of ITS coordinates (a host-side longest-processing-time split by sub-program cost).  The site
// rows of the tile are shared and read-only inside a gradient; what an evaluation writes -- the perturbed coordinate, expression
// temporaries, Categorical tables, select options -- lives in a small block of rows private to the wave (FgRemap, fg_interp.h:
// the interpreter redirects reads of slot i to the wave's `pert` row and offsets every row above the sites), so a tile costs
// S + W (temporaries + 2) rows of LDS, not W copies.  Each wave kicks its p_i in the shared momentum rows; after a workgroup
// barrier the drift of coordinate k is applied by wave k mod W, and a second barrier publishes it.  The sequential parts (Hamiltonians, the endpoint score
// in program order, accept, dual averaging) run on wave 0 exactly as in k_hmc_stream_steps.  Per coordinate the operations and
// their order are those of fg_trajectory (fg_engine.hip), so the kernel is bit-identical to k_hmc_steps for every W
// (tests/test_gpu_parity.py::test_hmc_interp_multiwave_is_bit_identical).
#include "fg_engine_internal.h"
#include "fg_cold.h"

#define FG_MWI_MAX 16         /* waves per tile */

struct FgMwi { int off[FG_MWI_MAX + 1]; const int *order; };   // wave w owns coordinates order[off[w] .. off[w + 1])

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
    double *xch = lds + ((long long)(P.S + W * np) + d) * tw + lane;                 // rows: 0 step size, 1 accepted, 2.. per-wave divergence flags
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
        // single call site: this wave's coordinates at +h then -h for gradients 0 .. L, then (wave 0) the whole program
        const int n_evals = (L + 1) * 2 * (j1 - j0) + (wv == 0 ? 1 : 0);
        bool bad = false;
        int s = 0, jj = j0, i = 0, slot = 0;
        FgCoord cd = {0, 0, 0, 0};
        double orig = 0.0, lp_plus = 0.0, lj_new = FG_NEG_INF;
        for (int ev = 0; ev < n_evals; ++ev) {
            const bool is_final = (wv == 0) && (ev == n_evals - 1);
            const bool minus = (ev & 1) != 0;
            const FgIns *prog = P.ins_fast;
            int n = P.n_ins;
            if (!is_final) {
                // the perturbed value goes to the wave's private row; the shared q_i is only read (hmc.rs:317-319 restores it: here it never changes)
                if (!minus) { i = seg.order[jj]; cd = P.coord[i]; slot = cd.slot; orig = slots[slot * tw]; slots[rm.pert * tw] = orig + h; }
                else slots[rm.pert * tw] = orig - h;
                if (sparse) { prog = P.sub + cd.sub_off; n = cd.sub_n; }
            }
            rm.pi = is_final ? 0xffffffffu : (uint32_t)slot;
            FgAcc3 A = {0.0, 0.0, 0.0};
            fg_exec<FG_MODE_SCORE, false, true>(prog, n, P.pool, slots, tw, A, nullptr, nullptr, 0, false, nullptr, &rm);
            const double tot = fg_total(A);
            if (is_final) { lj_new = tot; break; }
            if (!minus) { lp_plus = tot; continue; }
            const double g = (lp_plus - tot) / (2.0 * h);            // hmc.rs:322
            bad = bad || !fg_finite(g);
            double p = pl[i * tw];
            p += hk * g;                                              // hmc.rs:389 / :400
            if (s > 0 && s < L) p += hk * g;                          // trailing kick of step s + leading kick of s + 1
            pl[i * tw] = p;
            if (++jj == j1) {
                jj = j0;
                __syncthreads();                                      // every p kicked, every read of q done
                if (s < L) {                                          // q += eps * M^-1 p   (hmc.rs:391-393): coordinate k by wave k mod W
                    for (int k = wv; k < d; k += W) {
                        const double mk = mi ? mi[(long long)k * X.C] : 1.0;
                        slots[k * tw] += e * mk * pl[k * tw];
                    }
                    __syncthreads();
                }
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

// OCC = waves per SIMD the register budget allows (2: 256 VGPRs, no spills; 3: 168; 4: 128 with the interpreter's cold paths spilling)
#define FG_MWI_KERNEL(OCC) \
__global__ __attribute__((amdgpu_waves_per_eu(OCC, OCC))) __launch_bounds__(FG_WAVE * 4 * OCC) \
void k_hmc_interp_mw_steps_occ##OCC(FgProgramDev P, FgChainCtx X, FgHmcDev H, FgMwi seg, int iter0, int n_steps, int n_warmup, int welford_on, double *draws, \
                                    int first_sample_t, double *pos_all, double *info) { \
    fg_hmc_interp_mw_body(P, X, H, seg, iter0, n_steps, n_warmup, welford_on, draws, first_sample_t, pos_all, info); }
FG_MWI_KERNEL(2)
FG_MWI_KERNEL(3)
FG_MWI_KERNEL(4)

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

int fg_hmc_interp_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info) {
    if (e->interp_mw_disabled || e->gt || e->tw != FG_WAVE || e->d < 2) return FG_E_UNSUPPORTED;
    const unsigned tiles = (unsigned)((e->C + e->tw - 1) / e->tw);
    const bool sparse = e->cfg.grad_mode != FG_GRAD_FD_DENSE;
    for (int j = 0; j < e->S; ++j) if (e->prog->site_slot[j] >= e->S) return FG_E_UNSUPPORTED;    // site rows first, the private rows above them
    for (int k = 0; k < e->d; ++k) if (e->prog->coord[k].slot != k) return FG_E_UNSUPPORTED;
    auto lds_for = [&](int W) { return (size_t)((long long)e->S + (long long)W * (e->n_slots - e->S + 1) + e->d + 2 + W) * FG_WAVE * sizeof(double); };
    int occ = 4;                                               // measured (tools/bench_interp_mw.py): 128 VGPRs with the cold paths spilling beats 168 and 198 -- the waves hide more than the spills cost
    if (const char *sp = std::getenv("FG_HMC_INTERP_OCC")) occ = std::min(4, std::max(2, std::atoi(sp)));
    const int wmax = 4 * occ;                                  // a workgroup's waves must fit one CU at that occupancy
    const int wcap = std::min(wmax, e->d);
    int W;
    int forced = e->mw_override;
    if (const char *sp = std::getenv("FG_HMC_INTERP_WAVES")) forced = std::atoi(sp);
    if (forced > 0) W = std::max(2, std::min(forced, wcap));
    else {
        // the fewest waves per tile that fill the CU's 4 x occ wave slots with the tiles it gets (each tile's LDS = W copies of the slots):
        // many tiles -> few waves each, more coordinates per wave and a better balance; few tiles -> the tile is all its CU has
        const long long n_cu = std::max(1, e->n_simd / 4), per_cu = ((long long)tiles + n_cu - 1) / n_cu;
        for (W = 2; W < wcap; ++W) {
            const long long resident = std::min<long long>(per_cu, (160 * 1024) / (long long)lds_for(W));
            if (resident * W >= 4 * occ) break;
        }
    }
    while (W > 1 && lds_for(W) > 160 * 1024) --W;
    if (W < 2) return FG_E_UNSUPPORTED;
    if (e->mwi_W != W || e->mwi_sparse != (int)sparse || !e->d_mwi_order) {     // the split: longest processing time first
        std::vector<long long> cost(e->d, 1);
        if (sparse)
            for (int k = 0; k < e->d; ++k) {
                long long cs = 0;
                for (int q = 0; q < e->prog->coord[k].sub_n; ++q) cs += mwi_ins_cost(e->prog->sub[e->prog->coord[k].sub_off + q]);
                cost[k] = std::max(1LL, cs);
            }
        std::vector<int> by(e->d);
        for (int k = 0; k < e->d; ++k) by[k] = k;
        std::stable_sort(by.begin(), by.end(), [&](int a, int b) { return cost[a] > cost[b]; });
        std::vector<std::vector<int>> bins(W);
        std::vector<long long> load(W, 0);
        for (int k : by) {
            int best = 0;
            for (int w = 1; w < W; ++w) if (load[w] < load[best]) best = w;
            bins[best].push_back(k); load[best] += cost[k];
        }
        // wave 0 also runs the endpoint score: give it the lightest bin
        int lightest = 0;
        for (int w = 1; w < W; ++w) if (load[w] < load[lightest]) lightest = w;
        std::swap(bins[0], bins[lightest]);
        std::vector<int> order;
        e->mwi_off.assign(FG_MWI_MAX + 1, e->d);
        for (int w = 0; w < W; ++w) {
            e->mwi_off[w] = (int)order.size();
            std::sort(bins[w].begin(), bins[w].end());
            order.insert(order.end(), bins[w].begin(), bins[w].end());
        }
        for (int w = W; w <= FG_MWI_MAX; ++w) e->mwi_off[w] = e->d;
        if (!e->d_mwi_order) HIPCHK(hipMalloc((void **)&e->d_mwi_order, (size_t)e->d * sizeof(int)));
        HIPCHK(hipMemcpyAsync(e->d_mwi_order, order.data(), (size_t)e->d * sizeof(int), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));                // `order` is a local
        e->mwi_W = W; e->mwi_sparse = (int)sparse;
    }
    FgMwi seg;
    for (int w = 0; w <= FG_MWI_MAX; ++w) seg.off[w] = e->mwi_off[w];
    seg.order = e->d_mwi_order;
    const size_t lds = lds_for(W);
#define FG_MWI_LAUNCH(OCC) do { if (int rc = set_lds(k_hmc_interp_mw_steps_occ##OCC, lds)) return rc; \
    hipLaunchKernelGGL(k_hmc_interp_mw_steps_occ##OCC, dim3(tiles), dim3(FG_WAVE * W), lds, e->stream, e->P, e->X, e->H, seg, iter0, n, e->n_warmup, welford_on, \
                       draws, first_sample_t, pos_all, info); } while (0)
    if (occ == 4) FG_MWI_LAUNCH(4); else if (occ == 3) FG_MWI_LAUNCH(3); else FG_MWI_LAUNCH(2);
#undef FG_MWI_LAUNCH
    HIPCHK(hipGetLastError());
    e->last_hmc_kernel = "k_hmc_interp_mw_steps W=" + std::to_string(W) + (occ != 2 ? " occ=" + std::to_string(occ) : std::string());
    return FG_OK;
}
