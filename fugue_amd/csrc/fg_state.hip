// fg_state.hip -- the incremental-driver surface of HmcSession / the wasm samplers:
//   * fg_state_export / fg_state_import: the whole per-chain sampler state as one flat blob -- exactly the fields of
//     HmcSession (hmc.rs:643-661: q (= the trace values), lj_cur, eps, frozen_eps, DualAveraging, m_inv, mass_sqrt, Welford,
//     mass_adapt_at, n_warmup, iter, l, h, target_accept) for every chain, the MH chain state (current trace, its log-weight,
//     DiminishingAdaptation per site: mcmc_utils.rs:30-42, the decided proposal kinds) and the iteration counters that
//     position the counter-based random streams.  run(a); export; new engine; import; run(b) == run(a + b), bit for bit.
//   * fg_hmc_step_recorded: HmcSession::step_recorded (hmc.rs:811-817): one transition of every chain plus, for a few chosen
//     chains, the leapfrog trajectory with the Hamiltonian at each integration point (hmc.rs:371-381).  Recording is
//     RNG-neutral (hmc.rs:1058-1087): the trajectory is REPLAYED by a read-only kernel from the chain's current state and
//     its (chain, iteration) Philox stream before the real transition runs, with the gradient arithmetic of the production
//     kernels (the gradient stream where the program has one, the interpreter otherwise).
#include "fg_engine_internal.h"
#include "fg_gradstream.h"

// ======================================================================================
// recorded trajectories
// ======================================================================================
// One gradient at the tile's current q into the LDS rows `grow` (p is left alone: the recorded Hamiltonian needs p between
// the trailing half-kick of a step and the leading half-kick of the next, which the production kernels apply together).
// Returns "some force component was non-finite".  The arithmetic is the production kernels': the gradient stream / dense
// stream where the program has one, the interpreter otherwise.
template <bool AN>
__device__ __forceinline__ bool fg_rec_gradient(const FgProgramDev &P, double *slots, double *pl, double *grow, int tw, double h, int grad_mode) {
    const bool sparse = grad_mode != FG_GRAD_FD_DENSE;
    if (sparse && P.gstream) {
        if (AN) return fg_grad_stream<1, true>(P.gstream, P.n_gstream, P.pool, slots, pl, tw, h, 0.0, false, grow, tw, true);
        return fg_grad_stream<2>(P.gstream, P.n_gstream, P.pool, slots, pl, tw, h, 0.0, false, grow, tw, true);
    }
    if (!sparse && P.sstream && P.sstream_kinds == 0)
        return fg_grad_dense_stream(P.sstream, P.n_sstream, 0, P.d, slots, pl, tw, h, 0.0, false, grow, tw, true);
    bool bad = false;
    for (int i = 0; i < P.d; ++i) {                            // grad_log_joint, hmc.rs:304-329
        const FgCoord cd = P.coord[i];
        const double orig = slots[cd.slot * tw];
        double lp[2];
        for (int sgn = 0; sgn < 2; ++sgn) {
            slots[cd.slot * tw] = sgn ? orig - h : orig + h;
            FgAcc3 A = {0.0, 0.0, 0.0};
            if (sparse) fg_exec<FG_MODE_SCORE, false>(P.sub + cd.sub_off, cd.sub_n, P.pool, slots, tw, A, nullptr, nullptr, 0, false);
            else fg_exec<FG_MODE_SCORE, false>(P.ins_fast, P.n_ins, P.pool, slots, tw, A, nullptr, nullptr, 0, false);
            lp[sgn] = fg_total(A);
        }
        slots[cd.slot * tw] = orig;
        const double g = (lp[0] - lp[1]) / (2.0 * h);          // hmc.rs:322
        bad = bad || !fg_finite(g);
        grow[i * tw] = g;
    }
    return bad;
}

// Replays the next transition's leapfrog trajectory of the chains ids[0..K): point 0 = (q0, p0), then one point per
// completed step; traj [K][L+1][d], ham [K][L+1] = -log pi(q) + kinetic energy (hmc.rs:371-381), n_points [K] (a trajectory
// that leaves the support stops at its last finite point, hmc.rs:384-398).  Nothing of the engine's state is written.
// gt: a program whose tile (sites + momentum + gradient rows) exceeds a CU's LDS records from a scratch in global memory
// ([blocks][n_slots + 2 d][64]): step_recorded has no size limit in the reference (hmc.rs:811-817).
template <bool AN, bool GT = false>
__global__ __launch_bounds__(FG_WAVE, FG_MIN_WAVES) void k_hmc_record(FgProgramDev P, FgChainCtx X, FgHmcDev H, const long long *ids, int K, int iter, int n_warmup,
                                                                      double *traj, double *ham, int *n_points, double *gt) {
    extern __shared__ double lds_[];
    double *lds = GT ? gt + (size_t)blockIdx.x * (P.n_slots + 2 * P.d) * FG_WAVE : lds_;
    constexpr int tw = FG_WAVE;
    const int l = blockIdx.x * tw + threadIdx.x;
    const bool live = l < K;
    const long long c = ids[live ? l : K - 1];
    double *slots = lds + threadIdx.x;
    double *pl = lds + (long long)P.n_slots * tw + threadIdx.x;
    double *grow = pl + (long long)P.d * tw;
    fg_load_values(P, X, c, slots, tw);
    const int d = P.d, L = H.L;
    const double *mi = H.use_mass ? H.m_inv + c : nullptr;
    const double *ms = H.use_mass ? H.mass_sqrt + c : nullptr;
    double e;                                                  // the step size the transition will use (hmc.rs:771-777)
    if (iter < n_warmup) e = H.eps[c];
    else { const double fr = H.frozen[c]; e = (fr == fr) ? fr : (n_warmup > 0 ? exp(H.da_leb[c]) : H.eps[c]); }
    FgStream rng = fg_stream(X.seed, X.chain0 + (uint32_t)c, (uint32_t)iter, FG_RNG_HMC);
    fg_draw_momentum(P, rng, pl, tw, ms, X.C);
    const double hk = 0.5 * e;
    int np = 0;
    bool bad = false;
    auto record = [&]() __attribute__((always_inline)) {       // record_point, hmc.rs:371-381 (inlined: np / bad stay in registers)
        FgAcc3 A = {0.0, 0.0, 0.0};
        fg_exec<FG_MODE_SCORE, false>(P.ins_fast, P.n_ins, P.pool, slots, tw, A, nullptr, nullptr, 0, false);
        const double hval = -fg_total(A) + fg_kinetic(P, pl, tw, mi, X.C);
        if (live && !bad) {
            double *row = traj + ((long long)l * (L + 1) + np) * d;
            for (int i = 0; i < d; ++i) row[i] = slots[i * tw];
            ham[(long long)l * (L + 1) + np] = hval;
            np += 1;
        }
    };
    record();
    for (int gs = 0; gs <= L; ++gs) {                          // leapfrog, hmc.rs:383-405
        bad = fg_rec_gradient<AN>(P, slots, pl, grow, tw, H.h, H.grad_mode) || bad;
        if (gs > 0) {                                          // trailing half-kick of step gs: the point is complete (hmc.rs:399-404)
            for (int k = 0; k < d; ++k) pl[k * tw] = pl[k * tw] + hk * grow[k * tw];
            record();
        }
        if (gs < L) {                                          // leading half-kick and drift of step gs + 1 (hmc.rs:387-393)
            for (int k = 0; k < d; ++k) pl[k * tw] = pl[k * tw] + hk * grow[k * tw];
            for (int k = 0; k < d; ++k) slots[k * tw] += (mi ? e * mi[(long long)k * X.C] : e) * pl[k * tw];
        }
    }
    if (live) n_points[l] = np;
}

extern "C" {

int fg_hmc_step_recorded(fg_engine *e, int n_recorded, const int64_t *h_chain_ids, double *h_traj, double *h_ham, int32_t *h_n_points, double *d_info) {
    NEED_ENGINE(e);
    if (!e->hmc_ready) { fg_set_error("fg_hmc_step_recorded before fg_hmc_init"); return FG_E_STATE; }
    if (n_recorded < 0 || (n_recorded > 0 && (!h_chain_ids || !h_traj || !h_ham || !h_n_points))) return FG_E_BAD_ARG;
    if (e->d == 0 && n_recorded > 0) { fg_set_error("fg_hmc_step_recorded: model has no continuous sites"); return FG_E_UNSUPPORTED; }
    for (int k = 0; k < n_recorded; ++k) if (h_chain_ids[k] < 0 || h_chain_ids[k] >= e->C) { fg_set_error("fg_hmc_step_recorded: chain id out of range"); return FG_E_BAD_ARG; }
    // the mass-matrix reset runs between transitions: a recorded transition never straddles it
    if (n_recorded > 0) {
        const int L = e->H.L, d = e->d;
        long long *d_ids = nullptr; double *d_traj = nullptr, *d_ham = nullptr, *d_gt = nullptr; int *d_np = nullptr;
        const size_t nt = (size_t)n_recorded * (L + 1) * d, nh = (size_t)n_recorded * (L + 1);
        if (dev_alloc(&d_ids, (size_t)n_recorded) || dev_alloc(&d_traj, nt) || dev_alloc(&d_ham, nh) || dev_alloc(&d_np, (size_t)n_recorded)) return FG_E_HIP;
        hipError_t he = hipMemcpyAsync(d_ids, h_chain_ids, (size_t)n_recorded * 8, hipMemcpyHostToDevice, e->stream);
        const unsigned nb = (unsigned)((n_recorded + FG_WAVE - 1) / FG_WAVE);
        if (he == hipSuccess) {
            const size_t lds = (size_t)(e->n_slots + 2 * e->d) * FG_WAVE * sizeof(double);
            if (lds > 160 * 1024) {                                        // the tile in global memory (zeroed: the always-zero slot is written by fg_load_values anyway)
                if (dev_alloc(&d_gt, (size_t)nb * (e->n_slots + 2 * e->d) * FG_WAVE)) he = hipErrorOutOfMemory;
                if (he == hipSuccess) {
                    if (e->cfg.grad_mode == FG_GRAD_ANALYTIC) hipLaunchKernelGGL((k_hmc_record<true, true>), dim3(nb), dim3(FG_WAVE), 0, e->stream, e->P, e->X, e->H, (const long long *)d_ids, n_recorded, e->iter, e->n_warmup, d_traj, d_ham, d_np, d_gt);
                    else hipLaunchKernelGGL((k_hmc_record<false, true>), dim3(nb), dim3(FG_WAVE), 0, e->stream, e->P, e->X, e->H, (const long long *)d_ids, n_recorded, e->iter, e->n_warmup, d_traj, d_ham, d_np, d_gt);
                }
            } else if (e->cfg.grad_mode == FG_GRAD_ANALYTIC) {
                he = hipFuncSetAttribute((const void *)k_hmc_record<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (he == hipSuccess) hipLaunchKernelGGL(k_hmc_record<true>, dim3(nb), dim3(FG_WAVE), lds, e->stream, e->P, e->X, e->H, (const long long *)d_ids, n_recorded, e->iter, e->n_warmup, d_traj, d_ham, d_np, (double *)nullptr);
            } else {
                he = hipFuncSetAttribute((const void *)k_hmc_record<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (he == hipSuccess) hipLaunchKernelGGL(k_hmc_record<false>, dim3(nb), dim3(FG_WAVE), lds, e->stream, e->P, e->X, e->H, (const long long *)d_ids, n_recorded, e->iter, e->n_warmup, d_traj, d_ham, d_np, (double *)nullptr);
            }
        }
        if (he == hipSuccess) he = hipGetLastError();
        if (he == hipSuccess) he = hipMemcpyAsync(h_traj, d_traj, nt * 8, hipMemcpyDeviceToHost, e->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(h_ham, d_ham, nh * 8, hipMemcpyDeviceToHost, e->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(h_n_points, d_np, (size_t)n_recorded * 4, hipMemcpyDeviceToHost, e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        (void)hipFree(d_ids); (void)hipFree(d_traj); (void)hipFree(d_ham); (void)hipFree(d_np); if (d_gt) (void)hipFree(d_gt);
        if (he != hipSuccess) { fg_set_error(hipGetErrorString(he)); return FG_E_HIP; }
    }
    return fg_internal_hmc_step(e, 1, nullptr, nullptr, d_info);       // the transition itself: the ordinary path, same random numbers
}

// ======================================================================================
// checkpoint of the per-chain state
// ======================================================================================
struct FgStateHeader {
    uint64_t magic; uint32_t version, flags;         // flags: 1 HMC state, 2 mass adaptation arrays, 4 MH state, 8 MH overrides
    int64_t C; int32_t S, d, n_slots, reserved0;
    uint64_t seed; uint32_t chain0, reserved1;
    int32_t iter, n_warmup, mass_adapt_at, use_mass;
    fg_hmc_config cfg;
    int32_t mh_iter, mh_warmup;
    uint64_t bytes;                                  // whole blob
};
static const uint64_t FG_STATE_MAGIC = 0x3145544154534746ull;   // "FGSTATE1"

static size_t state_bytes_of(const fg_engine *e, uint32_t flags) {
    const size_t C = (size_t)e->C, S = (size_t)std::max(1, e->S), d = (size_t)std::max(1, e->d);
    size_t n = sizeof(FgStateHeader) + S * C * 8;
    if (flags & 1u) n += 9 * C * 8;                                      // lj eps frozen da_mu da_leb da_hbar da_m alpha_sum n_div
    if (flags & 2u) n += 4 * d * C * 8 + C * 8;                          // m_inv mass_sqrt w_mean w_m2, w_n
    if (flags & 4u) n += C * 8 + S * C * sizeof(FgMhAdapt) + C * 8;        // lw, {scale, kind, log_scale, tot, acc} per (site, chain), n_acc
    if (flags & 8u) n += S * sizeof(fg_site_proposal);
    return n;
}
static size_t state_bytes(const fg_engine *e, uint32_t &flags) {
    flags = (e->hmc_ready ? 1u : 0u) | ((e->hmc_ready && e->H.m_inv) ? 2u : 0u) | (e->mh_ready ? 4u : 0u) | ((e->mh_ready && !e->mh_overrides.empty()) ? 8u : 0u);
    return state_bytes_of(e, flags);
}

int64_t fg_state_size(fg_engine *e) {
    if (!e) return FG_E_BAD_ARG;
    uint32_t fl;
    return (int64_t)state_bytes(e, fl);
}

int fg_state_export(fg_engine *e, void *h_buf, size_t capacity) {
    NEED_ENGINE(e);
    uint32_t fl;
    const size_t need = state_bytes(e, fl);
    if (!h_buf || capacity < need) { fg_set_error("fg_state_export: buffer too small (fg_state_size)"); return FG_E_BAD_ARG; }
    HIPCHK(hipStreamSynchronize(e->stream));
    FgStateHeader hd; std::memset(&hd, 0, sizeof(hd));
    hd.magic = FG_STATE_MAGIC; hd.version = 2; hd.flags = fl; hd.C = e->C; hd.S = e->S; hd.d = e->d; hd.n_slots = e->n_slots;
    hd.seed = e->seed; hd.chain0 = e->chain0; hd.reserved1 = e->M.rec_all ? 1u : 0u;   // MhSession-style recording while adapting (fg_mh_set_recording)
    hd.iter = e->iter; hd.n_warmup = e->n_warmup; hd.mass_adapt_at = e->mass_adapt_at; hd.use_mass = e->H.use_mass;
    hd.cfg = e->cfg; hd.mh_iter = e->mh_iter; hd.mh_warmup = e->mh_warmup; hd.bytes = need;
    char *o = (char *)h_buf;
    std::memcpy(o, &hd, sizeof(hd)); o += sizeof(hd);
    const size_t C = (size_t)e->C, S = (size_t)std::max(1, e->S), d = (size_t)std::max(1, e->d);
    auto dl = [&](const void *dev, size_t bytes) { const hipError_t he = hipMemcpy(o, dev, bytes, hipMemcpyDeviceToHost); o += bytes; return he; };
    HIPCHK(dl(e->d_values, S * C * 8));
    if (fl & 1u) {
        const void *a[9] = { e->H.lj, e->H.eps, e->H.frozen, e->H.da_mu, e->H.da_leb, e->H.da_hbar, e->H.da_m, e->H.alpha_sum, e->H.n_div };
        for (const void *p : a) HIPCHK(dl(p, C * 8));
    }
    if (fl & 2u) {
        const void *a[4] = { e->H.m_inv, e->H.mass_sqrt, e->H.w_mean, e->H.w_m2 };
        for (const void *p : a) HIPCHK(dl(p, d * C * 8));
        HIPCHK(dl(e->H.w_n, C * 8));
    }
    if (fl & 4u) {
        HIPCHK(dl(e->M.lw, C * 8)); HIPCHK(dl(e->M.ad, S * C * sizeof(FgMhAdapt))); HIPCHK(dl(e->M.n_acc, C * 8));
    }
    if (fl & 8u) { std::vector<fg_site_proposal> ov(S); std::copy(e->mh_overrides.begin(), e->mh_overrides.end(), ov.begin()); std::memcpy(o, ov.data(), S * sizeof(fg_site_proposal)); o += S * sizeof(fg_site_proposal); }
    return FG_OK;
}

int fg_state_import(fg_engine *e, const void *h_buf, size_t size) {
    NEED_ENGINE(e);
    if (!h_buf || size < sizeof(FgStateHeader)) { fg_set_error("fg_state_import: truncated blob"); return FG_E_BAD_ARG; }
    FgStateHeader hd; std::memcpy(&hd, h_buf, sizeof(hd));
    if (hd.magic != FG_STATE_MAGIC || hd.version != 2) { fg_set_error("fg_state_import: not a fugue_amd state blob (magic / version)"); return FG_E_BAD_ARG; }
    if (hd.C != e->C || hd.S != e->S || hd.d != e->d || hd.n_slots != e->n_slots) {
        fg_set_error("fg_state_import: the blob was exported from a different program or chain count"); return FG_ERR_UNEXPECTED_STRUCTURE; }
    // Everything the uploads below will read is validated BEFORE the engine is touched: the flags name a layout, the layout
    // names a size, and header and caller must both agree with it (a blob with flipped flag bits or a short `bytes` would
    // otherwise make the copies run past the caller's buffer).
    const uint32_t fl = hd.flags;
    if ((fl & ~15u) || ((fl & 2u) && !(fl & 1u)) || ((fl & 8u) && !(fl & 4u))) { fg_set_error("fg_state_import: unknown or inconsistent section flags"); return FG_E_BAD_ARG; }
    const size_t expected = state_bytes_of(e, fl);
    if (hd.bytes != expected || size < expected) { fg_set_error("fg_state_import: truncated or corrupted blob (size does not match its sections)"); return FG_E_BAD_ARG; }
    if (fl & 1u) {
        const fg_hmc_config &c = hd.cfg;
        const bool ok = (c.grad_mode == FG_GRAD_FD_DENSE || c.grad_mode == FG_GRAD_FD_SPARSE || c.grad_mode == FG_GRAD_ANALYTIC) && c.n_leapfrog >= 1 &&
                        c.n_leapfrog <= 100000 && c.finite_diff_eps > 0.0 && std::isfinite(c.finite_diff_eps) && hd.iter >= 0 && hd.n_warmup >= 0 &&
                        (hd.use_mass == 0 || hd.use_mass == 1) && (hd.use_mass == 0 || (fl & 2u)) && hd.mass_adapt_at >= -1;
        if (!ok) { fg_set_error("fg_state_import: corrupted HMC section header"); return FG_E_BAD_ARG; }
    }
    if ((fl & 4u) && (hd.mh_iter < 0 || hd.mh_warmup < 0)) { fg_set_error("fg_state_import: corrupted MH section header"); return FG_E_BAD_ARG; }
    const size_t C = (size_t)e->C, S = (size_t)std::max(1, e->S), d = (size_t)std::max(1, e->d);
    std::vector<fg_site_proposal> ov;
    if (fl & 8u) {
        ov.resize(S);
        std::memcpy(ov.data(), (const char *)h_buf + expected - S * sizeof(fg_site_proposal), S * sizeof(fg_site_proposal));
        for (int j = 0; j < e->S; j++) if (ov[j].kind < 0 || ov[j].kind > 4) { fg_set_error("fg_state_import: corrupted proposal overrides"); return FG_E_BAD_ARG; }
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (fl & 1u) { if (int rc = fg_internal_hmc_alloc(e, (fl & 2u) != 0u)) return rc; }
    if (fl & 4u) { if (int rc = fg_internal_mh_alloc(e)) return rc; }
    // From here on only a HIP failure can interrupt: the sessions are marked not ready while their arrays are being replaced
    // and committed (seed, counters, ready flags) only after every upload has succeeded.
    const bool had_hmc = e->hmc_ready, had_mh = e->mh_ready;
    e->hmc_ready = false; e->mh_ready = false;
    const char *o = (const char *)h_buf + sizeof(hd);
    auto ul = [&](void *dev, size_t bytes) { const hipError_t he = hipMemcpy(dev, o, bytes, hipMemcpyHostToDevice); o += bytes; return he; };
    HIPCHK(ul(e->d_values, S * C * 8));
    if (fl & 1u) {
        void *a[9] = { e->H.lj, e->H.eps, e->H.frozen, e->H.da_mu, e->H.da_leb, e->H.da_hbar, e->H.da_m, e->H.alpha_sum, e->H.n_div };
        for (void *p : a) HIPCHK(ul(p, C * 8));
    }
    if (fl & 2u) {
        void *a[4] = { e->H.m_inv, e->H.mass_sqrt, e->H.w_mean, e->H.w_m2 };
        for (void *p : a) HIPCHK(ul(p, d * C * 8));
        HIPCHK(ul(e->H.w_n, C * 8));
    }
    if (fl & 4u) {
        HIPCHK(ul(e->M.lw, C * 8)); HIPCHK(ul(e->M.ad, S * C * sizeof(FgMhAdapt))); HIPCHK(ul(e->M.n_acc, C * 8));
        if (int rc = fg_internal_mh_set_overrides(e, (fl & 8u) ? ov.data() : nullptr)) return rc;
    }
    // commit.  The random streams are keyed by (seed, global chain id, iteration): the importing engine takes the exporter's key
    e->seed = hd.seed; e->chain0 = hd.chain0; e->X.seed = hd.seed; e->X.chain0 = hd.chain0;
    if (fl & 1u) {
        fg_internal_hmc_set_cfg(e, &hd.cfg);
        e->H.use_mass = hd.use_mass; e->n_warmup = hd.n_warmup; e->iter = hd.iter; e->mass_adapt_at = hd.mass_adapt_at; e->hmc_ready = true;
    } else e->hmc_ready = had_hmc;
    if (fl & 4u) { e->mh_iter = hd.mh_iter; e->mh_warmup = hd.mh_warmup; e->M.rec_all = hd.reserved1 ? 1 : 0; e->mh_ready = true; }
    else e->mh_ready = had_mh;
    return FG_OK;
}

}  // extern "C"
