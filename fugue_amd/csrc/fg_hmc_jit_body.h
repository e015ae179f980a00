// fg_hmc_jit_body.h -- HmcSession::step (hmc.rs:819-919) around a model compiled at run time (fg_jit.cpp).
//
// This text is compiled by hiprtc, behind the generated functions of ONE model:
//     double fg_jit_task(int k, double pert, const double *slots)   log-joint terms that read coordinate k, with q_k replaced by `pert`
//                                                                  -- the straight-line form of the sparse finite difference's sub-program k
//     void   fg_jit_score(const double *slots, double &pr, double &lk, double &fc)   the whole program (score_full, hmc.rs:283-299)
// Both perform the interpreter's operations in the interpreter's order (fg_interp.h: one C++ statement per FgIns), with expression
// temporaries in registers instead of LDS rows and every instruction field a literal.  The kernel is fg_hmc_interp.hip's: a tile of 64
// chains shared by W waves, (coordinate, sign) tasks split over the waves, kick + drift by wave k mod W behind a barrier, the
// sequential parts on wave 0 -- so its results are bit-identical to k_hmc_interp_mw_steps and k_hmc_steps
// (tests/test_gpu_jit.py).  LDS: S site rows + d momentum rows + 2 d evaluation rows + 2 + W exchange rows.
#define FG_JIT_WMAX 16
#ifndef FG_JIT_OCC          /* waves per SIMD the register budget allows: 4 = 128 VGPRs (fg_jit.cpp may define 2 or 3 for register-hungry programs) */
#define FG_JIT_OCC 4
#endif
struct FgJitSeg { int off[FG_JIT_WMAX + 1]; const int *order; int baked; };   // wave w owns tasks order[off[w] .. off[w + 1]): 2 k + sign; baked: 1 = this IS the split fg_jit_wave_tasks was generated for,
                                                                           // 2 = whole coordinates per wave as fg_jit_wave_grad holds them (one barrier per gradient; S more rows of LDS)

extern "C" __global__ __attribute__((amdgpu_waves_per_eu(FG_JIT_OCC, FG_JIT_OCC))) __launch_bounds__(FG_WAVE * 4 * FG_JIT_OCC)
void k_hmc_jit_steps(FgProgramDev P, FgChainCtx X, FgHmcDev H, FgJitSeg seg, int iter0, int n_steps, int n_warmup, int welford_on, double *draws,
                     int first_sample_t, double *pos_all /*[n][d][C] or null*/, double *info /*[n][4][C] or null*/) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
#ifdef FG_JIT_K_D          /* the program's own constants as literals (fg_jit.cpp): LDS rows at instruction offsets, loop bounds known */
    constexpr int d = FG_JIT_K_D, S_ = FG_JIT_K_S;
    const int L = H.L;
#else
    const int d = P.d, L = H.L, S_ = P.S;
#endif
    double *slots = lds + lane;                                                      // site rows [0, S)
    double *pl = lds + (long long)S_ * tw + lane;                                   // momentum rows
    double *ev_lp = lds + (long long)(S_ + d) * tw + lane;                          // log-joint of evaluation (coordinate k, sign): row 2 k + sign
    double *xch = lds + (long long)(S_ + 3 * d) * tw + lane;                        // rows: 0 step size, 1 accepted, 2.. per-wave divergence flags
    const int j0 = seg.off[wv], j1 = seg.off[wv + 1];
    const double *mi = H.use_mass ? H.m_inv + c : nullptr;
    const double *ms = H.use_mass ? H.mass_sqrt + c : nullptr;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    const double h = H.h;
    const bool analytic = H.grad_mode == 2 /* FG_GRAD_ANALYTIC */;      // (the step-size search keeps the finite difference, like the stream kernels')
    const bool dense = H.grad_mode == 0 /* FG_GRAD_FD_DENSE */; (void)dense;
#ifdef FG_JIT_BAKED_W
    const bool baked = seg.baked == 1 && W == FG_JIT_BAKED_W && !analytic && !dense;
#endif
#if defined(FG_JIT_FUSED_DENSE_W) && !defined(FG_JIT_FUSED_W)
#define FG_JIT_FUSED_W 0   /* (a unit with the dense form only: the sparse one never matches) */
#define FG_JIT_NO_SPARSE_FUSED 1
#endif
#ifdef FG_JIT_FUSED_W
    // one barrier per gradient: a wave runs whole coordinates (both evaluations, force, kick, drift) and writes the new position into the second copy of
    // the site rows; the two copies change roles at the barrier.  Sites that are not coordinates never change during a transition: equal in both copies.
#ifdef FG_JIT_FUSED_DENSE_W
    const bool fused_dense = seg.baked == 3 && W == FG_JIT_FUSED_DENSE_W && dense;      // the same for the dense mode (whole-program evaluations)
#else
    const bool fused_dense = false;
#endif
    const bool fused = (seg.baked == 2 && W == FG_JIT_FUSED_W && !analytic && !dense) || fused_dense;
    double *slots2 = lds + (long long)(S_ + 3 * d + 2 + W) * tw + lane;
#else
    const bool fused = false;
#endif
    for (int j = wv; j < S_; j += W) slots[P.site_slot[j] * tw] = fg_as_double(X.values[(long long)j * X.C + c]);
#ifdef FG_JIT_FUSED_W
    if (fused) for (int j = wv; j < S_; j += W) slots2[P.site_slot[j] * tw] = fg_as_double(X.values[(long long)j * X.C + c]);
#endif
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
        const int n_pairs = (d + 1) >> 1;
        for (int j = wv; j < n_pairs; j += W) {              // p0 ~ N(0, M) (hmc.rs:436-441): Box-Muller pair j = Philox block j
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
        bool bad = false;
        double *cur = slots;                                  // the copy of the site rows that holds the current position
#ifdef FG_JIT_FUSED_W
        double *alt = slots2;
#endif
        for (int s = 0; s <= L; ++s) {                        // gradients 0 .. L of the leapfrog (hmc.rs:353-407)
#ifdef FG_JIT_FUSED_W
            if (fused) {
                bool b_ = false;
#ifdef FG_JIT_FUSED_DENSE_W
                if (fused_dense) b_ = fg_jit_wave_grad_dense(wv, h, hk, e, s > 0 && s < L, s < L, FG_JIT_LDS(cur), FG_JIT_LDS(alt), FG_JIT_LDS(pl), mi, X.C);
                else
#endif
#ifndef FG_JIT_NO_SPARSE_FUSED
                b_ = fg_jit_wave_grad(wv, h, hk, e, s > 0 && s < L, s < L, FG_JIT_LDS(cur), FG_JIT_LDS(alt), FG_JIT_LDS(pl), mi, X.C);
#else
                { }
#endif
                bad = bad || b_;
                if (s < L) { __syncthreads(); double *t_ = cur; cur = alt; alt = t_; }      // every new position written, every read of the old one done
                continue;
            }
#endif
#ifdef FG_JIT_BAKED_W
            if (baked) fg_jit_wave_tasks(wv, h, FG_JIT_LDS(slots), FG_JIT_LDS(ev_lp));      // hmc.rs:317-321, this wave's tasks as straight-line code
            else
#endif
            for (int jj = j0; jj < j1; ++jj) {
                const int task = seg.order[jj];
                const int k = task >> 1;
#ifdef FG_JIT_HAS_AD       /* FG_GRAD_ANALYTIC (opt-in): the "+" task of a coordinate forms d log pi / d q_k itself, the "-" task nothing */
                if (analytic) { if (!(task & 1)) ev_lp[task * tw] = fg_jit_dtask(k, FG_JIT_LDS(slots)); continue; }
#endif
                const double orig = slots[k * tw];
#ifdef FG_JIT_HAS_DENSE    /* FG_GRAD_FD_DENSE: the whole log-joint at q +- h e_k (grad_log_joint verbatim, hmc.rs:304-329) */
                if (dense) { ev_lp[task * tw] = fg_jit_dense_task(k, (task & 1) ? orig - h : orig + h, FG_JIT_LDS(slots)); continue; }
#endif
                ev_lp[task * tw] = fg_jit_task(k, (task & 1) ? orig - h : orig + h, FG_JIT_LDS(slots));      // hmc.rs:317-321
            }
            __syncthreads();                                  // every evaluation of this gradient done, every read of q done
            for (int k = wv; k < d; k += W) {
                const double g = analytic ? ev_lp[2 * k * tw] : (ev_lp[2 * k * tw] - ev_lp[(2 * k + 1) * tw]) / (2.0 * h);    // hmc.rs:322
                bad = bad || !fg_finite(g);
                double p = pl[k * tw];
                p += hk * g;                                  // hmc.rs:389 / :400
                if (s > 0 && s < L) p += hk * g;              // trailing kick of step s + leading kick of s + 1
                pl[k * tw] = p;
                if (s < L) {                                  // q += eps * M^-1 p   (hmc.rs:391-393)
                    const double mk = mi ? mi[(long long)k * X.C] : 1.0;
                    slots[k * tw] += e * mk * p;
                }
            }
            __syncthreads();
        }
        double lj_new = FG_NEG_INF;
        if (wv == 0) {
            double pr = 0.0, lk = 0.0, fc = 0.0;
            fg_jit_score(FG_JIT_LDS(cur), pr, lk, fc);
            lj_new = pr + lk + fc;                            // total_log_weight (trace.rs:198-200)
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
            if (acc) { const double v = cur[k * tw]; if (live) X.values[g] = fg_as_i64(v); slots[k * tw] = v; }      // (the next transition starts from the first copy)
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

// find_reasonable_epsilon (hmc.rs:479-535) around the same generated functions: the doubling / halving search of k_hmc_find_eps
// (fg_engine.hip), one leapfrog step per trial, the two gradients of a trial split over the waves like a transition's.  Every wave
// carries the per-lane search state (it is a function of the trial's log-ratio, which wave 0 publishes), so the loop condition is
// the same on all of them.
extern "C" __global__ __attribute__((amdgpu_waves_per_eu(FG_JIT_OCC, FG_JIT_OCC))) __launch_bounds__(FG_WAVE * 4 * FG_JIT_OCC)
void k_hmc_jit_find_eps(FgProgramDev P, FgChainCtx X, FgHmcDev H, FgJitSeg seg, uint32_t instance, int injected, double *eps_out) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = (int)(blockDim.x >> 6);
    const long long chain = (long long)blockIdx.x * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
#ifdef FG_JIT_K_D
    constexpr int d = FG_JIT_K_D, S_ = FG_JIT_K_S;
#else
    const int d = P.d, S_ = P.S;
#endif
    double *slots = lds + lane;
    double *pl = lds + (long long)S_ * tw + lane;
    double *ev_lp = lds + (long long)(S_ + d) * tw + lane;
    double *xch = lds + (long long)(S_ + 3 * d) * tw + lane;                        // rows: 0 the trial's log-ratio, 2.. per-wave divergence flags
    const int j0 = seg.off[wv], j1 = seg.off[wv + 1];
    const double *mi = H.use_mass ? H.m_inv + c : nullptr;
    const double *ms = H.use_mass ? H.mass_sqrt + c : nullptr;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    const double h = H.h;
    for (int j = wv; j < S_; j += W) slots[P.site_slot[j] * tw] = fg_as_double(X.values[(long long)j * X.C + c]);
    const int n_pairs = (d + 1) >> 1;
    if (!injected) {                                          // p0 from the chain's (instance) EPS stream: pair j = Philox block j
        for (int j = wv; j < n_pairs; j += W) {
            const FgD2 zz = fg_cold_normal_pair(sk0, sk1, gchain, (uint32_t)j, instance, FG_RNG_EPS);
            const int i = 2 * j;
            const double a0 = zz.a * (ms ? ms[(long long)i * X.C] : 1.0);
            pl[i * tw] = a0;
            if (live) H.p0_scratch[(long long)i * X.C + c] = a0;
            if (i + 1 < d) { const double a1 = zz.b * (ms ? ms[(long long)(i + 1) * X.C] : 1.0); pl[(i + 1) * tw] = a1; if (live) H.p0_scratch[(long long)(i + 1) * X.C + c] = a1; }
        }
    } else for (int j = wv; j < n_pairs; j += W) for (int i = 2 * j; i < 2 * j + 2 && i < d; ++i) pl[i * tw] = H.p0_scratch[(long long)i * X.C + c];
    __syncthreads();
    double h0 = 0.0;
    if (wv == 0) h0 = -H.lj[c] + fg_kinetic(P, pl, tw, mi, X.C);
    // p0 of a dead lane: the pair its mirror chain drew (the same counter), so the restore below reads what the lane itself would have written
    const double ln_half = log(0.5), ln2 = log(2.0);            // folded at compile time, as in k_hmc_find_eps
    double eps = 1.0, lr = 0.0, a = 1.0;
    bool active = true, first = true;
    unsigned iters = 0;
    __syncthreads();
    while (__any(active)) {
        const double eps_try = first ? 1.0 : eps * ((a > 0.0) ? 2.0 : 0.5);   // eps * 2^a
        const double hk = 0.5 * eps_try;
        bool bad = false;
        for (int s = 0; s <= 1; ++s) {                        // one leapfrog step (hmc.rs:500-511): two gradients
            for (int jj = j0; jj < j1; ++jj) {
                const int task = seg.order[jj];
                const int k = task >> 1;
                const double orig = slots[k * tw];
                ev_lp[task * tw] = fg_jit_task(k, (task & 1) ? orig - h : orig + h, FG_JIT_LDS(slots));
            }
            __syncthreads();
            for (int k = wv; k < d; k += W) {
                const double g = (ev_lp[2 * k * tw] - ev_lp[(2 * k + 1) * tw]) / (2.0 * h);
                bad = bad || !fg_finite(g);
                double p = pl[k * tw];
                p += hk * g;
                pl[k * tw] = p;
                if (s < 1) { const double mk = mi ? mi[(long long)k * X.C] : 1.0; slots[k * tw] += eps_try * mk * p; }
            }
            __syncthreads();
        }
        xch[(2 + wv) * tw] = bad ? 1.0 : 0.0;
        __syncthreads();
        if (wv == 0) {
            double pr = 0.0, lk = 0.0, fc = 0.0;
            fg_jit_score(FG_JIT_LDS(slots), pr, lk, fc);
            const double lj1 = pr + lk + fc;
            bool div = !fg_finite(lj1);
            for (int w = 0; w < W; ++w) div = div || xch[(2 + w) * tw] != 0.0;
            double lr_try = FG_NEG_INF;
            if (!div) lr_try = h0 - (-lj1 + fg_kinetic(P, pl, tw, mi, X.C));
            xch[0] = lr_try;
        }
        __syncthreads();
        const double lr_try = xch[0];
        for (int j = wv; j < n_pairs; j += W)                 // restore (q, p0) for the next trial: the pairs this wave drew
            for (int i = 2 * j; i < 2 * j + 2 && i < d; ++i) {
                slots[i * tw] = fg_as_double(X.values[(long long)P.f64_site[i] * X.C + c]);
                pl[i * tw] = H.p0_scratch[(long long)i * X.C + c];
            }
        __syncthreads();
        if (first) {
            lr = lr_try;
            a = (lr > ln_half) ? 1.0 : -1.0;
            active = a * lr > -a * ln2;
            first = false;
        } else if (active) {
            eps = eps_try; lr = lr_try; iters += 1;
            if (iters > 100 || !(eps >= 1e-12 && eps <= 1e12)) active = false;
            else if (a > 0.0 && lr == FG_NEG_INF) { eps /= 2.0; active = false; }
            else active = a * lr > -a * ln2;
        }
    }
    if (wv == 0 && live) eps_out[c] = fmin(fmax(eps, 1e-6), 1e3);
}


// ---- adaptive_smc's rejuvenation move (tempered_single_site_mh, smc.rs:631-688) around the same compiled model: k_smc_rejuv<-1>
// (fg_smc.hip) with the two scoring runs of a move as fg_jit_score instead of the interpreter -- the same operations in the same order,
// so the same particles, decisions and counts (tests/test_gpu_jit.py).  One tile (S site rows) per wave, `blockDim.x / 64` tiles per block.
extern "C" __global__ __launch_bounds__(FG_WAVE * 16)
void k_smc_jit_rejuv(FgProgramDev P, FgChainCtx X, FgSmcDev M, const FgSmcScalars *st, uint32_t move_id, const long long *vsrc /* the resampled population, or null: X.values */,
                     double *pmax /* the block maxima of the new log-likelihoods, or null (fg_smc.hip: k_smc_rejuv's two extras) */) {
    extern __shared__ double lds[];
    __shared__ unsigned int hist[2][FG_SMC_HIST];                   // the block's proposal / accept counts per site
    __shared__ double shmax[16];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1), wv = (int)(threadIdx.x >> 6);
    for (int j = (int)threadIdx.x; j < 2 * FG_SMC_HIST; j += (int)blockDim.x) (&hist[0][0])[j] = 0u;
    __syncthreads();
    const long long chain = ((long long)blockIdx.x * (blockDim.x >> 6) + wv) * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + (long long)wv * P.S * tw + lane;                // one tile per wave: the site rows (expression temporaries are registers here)
    const long long *vin = vsrc ? vsrc : X.values;
    for (int j = 0; j < P.S; ++j) slots[P.site_slot[j] * tw] = fg_as_double(vin[(long long)j * X.C + c]);
    const double beta = st->beta;
    const uint32_t sk0 = (uint32_t)X.seed, sk1 = (uint32_t)(X.seed >> 32), gchain = X.chain0 + (uint32_t)c;
    FgStream rng = fg_stream(X.seed, gchain, move_id, FG_RNG_SMC_REJUV);
    unsigned long long ra, rb;
    fg_rng_block(rng, ra, rb);
    const int k = (int)fg_pick(ra, (uint32_t)P.d);            // f64_sites[rng.gen_range(0..len)]  smc.rs:650
    const int site = P.f64_site[k];                           // sorted site index (adaptation / values row)
    const double scale = M.scale[site];                       // get_scale  smc.rs:651
    const double z = fg_cold_normal_pair(sk0, sk1, gchain, 1u, move_id, FG_RNG_SMC_REJUV).a;      // Normal(0,1).sample  smc.rs:655 (block 1)
    const double cur = slots[k * tw];                         // LDS slot of coordinate k is k
    const double prop = cur + scale * z;
    double pri[2], lik[2];
    for (int pass = 0; pass < 2; ++pass) {                    // score current, then proposed: two model runs  smc.rs:662-675
        slots[k * tw] = pass ? prop : cur;
        double pr, lk, fc;
        fg_jit_score(FG_JIT_LDS(slots), pr, lk, fc);
        pri[pass] = pr; lik[pass] = lk + fc;
    }
    const double log_alpha = (pri[1] - pri[0]) + beta * (lik[1] - lik[0]);                   // smc.rs:678-679
    const double u = fg_cold_u01_pair(sk0, sk1, gchain, 2u, move_id, FG_RNG_SMC_REJUV).a;     // block 2
    const bool accept = (log_alpha >= 0.0) || (u < fg_cold_exp(log_alpha));                  // smc.rs:680
    const double ll_new = accept ? lik[1] : lik[0];
    if (live) {
        if (vsrc) for (int j = 0; j < P.S; ++j) X.values[(long long)j * X.C + c] = vsrc[(long long)j * X.C + c];
        if (accept) X.values[(long long)site * X.C + c] = fg_as_i64(prop);
        M.lprior[c] = accept ? pri[1] : pri[0];               // the freshly scored trace is returned either way
        M.ll[c] = ll_new;
    }
    if (pmax) {                                               // (fmax from -inf, like k_smc_red_max)
        double m = live ? ll_new : -INFINITY;
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
        if (lane == 0) shmax[wv] = m;
    }
    unsigned long long todo = __ballot(live);
    const unsigned long long acc_mask = __ballot(live && accept);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int s_lead = __builtin_amdgcn_readlane(site, leader);
        const unsigned long long same = __ballot(live && site == s_lead);
        if (lane == leader) {
            atomicAdd(&hist[0][s_lead], (unsigned int)__popcll(same));
            const unsigned int na = (unsigned int)__popcll(same & acc_mask);
            if (na) atomicAdd(&hist[1][s_lead], na);
        }
        todo &= ~same;
    }
    __syncthreads();
    unsigned int *row = M.blk + (long long)blockIdx.x * 2 * M.S;
    for (int j = (int)threadIdx.x; j < M.S; j += (int)blockDim.x) { row[j] = hist[0][j]; row[M.S + j] = hist[1][j]; }
    if (pmax && threadIdx.x == 0) { double m = shmax[0]; for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = fmax(m, shmax[k]); pmax[blockIdx.x] = m; }
}


// ---- run(PriorHandler, model) and run(ScoreGivenTrace, model) per chain (interpreters.rs:88-104, 138-163) as generated code: k_prior_init /
// k_log_joint (fg_engine.hip) with fg_jit_prior / fg_jit_score instead of the interpreter -- the same draws from the same stream in the
// same order, the same accumulators.  One tile (S site rows) per wave, blockDim.x / 64 tiles per block.
#ifdef FG_JIT_HAS_PRIOR
extern "C" __global__ __launch_bounds__(FG_WAVE * 4)
void k_prior_jit(FgProgramDev P, FgChainCtx X, uint32_t iteration, uint32_t purpose, double *acc_out /*[3][C] or null*/, double *lj_out /*[C] or null*/) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1), wv = (int)(threadIdx.x >> 6);
    const long long chain = ((long long)blockIdx.x * (blockDim.x >> 6) + wv) * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + (long long)wv * P.S * tw + lane;
    for (int j = 0; j < P.S; ++j) slots[j * tw] = 0.0;
    const FgStream rng = fg_stream(X.seed, X.chain0 + (uint32_t)c, iteration, purpose);
    double pr, lk, fc;
    fg_jit_prior(FG_JIT_LDS(slots), rng, pr, lk, fc);
    if (live) {
        for (int j = 0; j < P.S; ++j) X.values[(long long)j * X.C + c] = fg_as_i64(slots[P.site_slot[j] * tw]);
        if (acc_out) { acc_out[c] = pr; acc_out[X.C + c] = lk; acc_out[2 * X.C + c] = fc; }
        if (lj_out) lj_out[c] = pr + lk + fc;
    }
}
#endif
extern "C" __global__ __launch_bounds__(FG_WAVE * 4)
void k_log_joint_jit(FgProgramDev P, FgChainCtx X, double *acc_out /*[3][C] or null*/, double *lj_out /*[C] or null*/) {
    extern __shared__ double lds[];
    constexpr int tw = FG_WAVE;
    const int lane = threadIdx.x & (FG_WAVE - 1), wv = (int)(threadIdx.x >> 6);
    const long long chain = ((long long)blockIdx.x * (blockDim.x >> 6) + wv) * tw + lane;
    const bool live = chain < X.C;
    const long long c = live ? chain : X.C - 1;
    double *slots = lds + (long long)wv * P.S * tw + lane;
    for (int j = 0; j < P.S; ++j) slots[P.site_slot[j] * tw] = fg_as_double(X.values[(long long)j * X.C + c]);
    double pr, lk, fc;
    fg_jit_score(FG_JIT_LDS(slots), pr, lk, fc);
    if (live) {
        if (acc_out) { acc_out[c] = pr; acc_out[X.C + c] = lk; acc_out[2 * X.C + c] = fc; }
        if (lj_out) lj_out[c] = pr + lk + fc;
    }
}
