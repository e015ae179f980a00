// fg_diag_host.cpp -- the cross-chain combination of split R-hat and multi-chain ESS in C++ (host only: usable and tested
// without a GPU).  Restates r_hat_from_f64_chains / split_f64_chains (src/inference/diagnostics.rs:240-304) and
// ess_from_chains (src/inference/mcmc_utils.rs:253-339) over per-chain MOMENTS instead of raw draws:
//   moments [d][6][m]: per (coordinate, chain) mean and sum of squared deviations of the full chain, of its first half and
//   of its second half (fg_diag_chain_moments);
//   pooled lag sums: sum over ALL chains of the biased lag-t autocovariance, fetched on demand in chunks through a callback
//   (on the GPU path: k_diag_autocov + an RCCL all-reduce; in the CPU tests: numpy + gloo).
// Sums over chains run in chain order, as the reference's iterators do.
#include <cmath>
#include <map>
#include <vector>

#include "fg_program.h"

extern "C" {
// h_sums [d][n_lags] <- sum over all chains of acov_t for t in [lag0, lag0 + n_lags); returns 0 on success
typedef int (*fg_acov_fn)(void *user, int lag0, int n_lags, double *h_sums);
typedef int (*fg_reduce_fn)(void *user, int stage, const double *h_in, double *h_out);
}

static double rhat_of(const double *means, const double *ssds, long long stride, long long m, double n) {   // diagnostics.rs:262-304
    if (m < 2) return 1.0;
    if (n == 0.0) return NAN;
    const double mf = (double)m;
    double s = 0.0;
    for (long long j = 0; j < m; j++) s += means[j * stride];
    const double overall = s / mf;
    double bsum = 0.0, wsum = 0.0;
    for (long long j = 0; j < m; j++) { const double dv = means[j * stride] - overall; bsum += dv * dv; }
    const double b = n / (mf - 1.0) * bsum;
    for (long long j = 0; j < m; j++) wsum += ssds[j * stride] / (n - 1.0);
    const double w = wsum / mf;
    const double var_plus = ((n - 1.0) / n) * w + (1.0 / n) * b;
    return std::sqrt(var_plus / w);
}

// split R-hat from aggregate sums (diagnostics.rs:262-304): `mm` chains of `nn` draws, between = sum_j (mean_j - overall)^2,
// wsum_n1 = sum_j ssd_j
static double rhat_from_sums(double between, double ssd_sum, long long mm, double nn) {
    if (mm < 2) return 1.0;
    if (nn == 0.0) return NAN;
    const double mf = (double)mm;
    const double b = nn / (mf - 1.0) * between;
    const double w = (ssd_sum / (nn - 1.0)) / mf;
    const double var_plus = ((nn - 1.0) / nn) * w + (1.0 / nn) * b;
    return std::sqrt(var_plus / w);
}

// multi-chain ESS of one coordinate (mcmc_utils.rs:253-339) from mean_var = mean over chains of acov0 n / (n - 1), between =
// sum_j (mean_j - overall)^2 and the pooled lag autocovariances (rho asks for lags in increasing order)
template <typename AcovMean>
static int ess_from_pooled(int64_t m, int n, double mean_var, double between, AcovMean &&acov_mean, double &out) {
    const int max_lag = std::min(n - 1, 2048);
    const double nf = (double)n, mf = (double)m;
    if (mean_var <= 0.0) { out = (double)((long long)m * n); return FG_OK; }
    double var_plus = mean_var * (nf - 1.0) / nf;
    if (m > 1) var_plus += between / (mf - 1.0);
    int rc = FG_OK;
    auto rho = [&](int t) { double a = 0.0; const int r = acov_mean(t, a); if (r) rc = r; return 1.0 - (mean_var - a) / var_plus; };
    std::vector<double> rho_hat((size_t)max_lag + 1, 0.0);
    rho_hat[0] = 1.0;
    if (max_lag >= 1) rho_hat[1] = rho(1);
    int t = 1, max_t = std::min(1, max_lag);
    while (t + 2 <= max_lag && !rc) {                       // Geyer initial positive sequence
        const double re = rho(t + 1), ro = rho(t + 2);
        if (re + ro < 0.0) break;
        rho_hat[t + 1] = re; rho_hat[t + 2] = ro;
        max_t = t + 2; t += 2;
    }
    if (rc) return rc;
    for (int k = 1; k + 2 <= max_t; k += 2) {               // monotone pair sums
        const double prev = rho_hat[k - 1] + rho_hat[k], cur = rho_hat[k + 1] + rho_hat[k + 2];
        if (cur > prev) { rho_hat[k + 1] = prev / 2.0; rho_hat[k + 2] = prev / 2.0; }
    }
    double sum_rho = 0.0;
    for (int k = 0; k <= max_t; k++) sum_rho += rho_hat[k];
    const double tau = std::max(-1.0 + 2.0 * sum_rho, 1.0);
    out = (double)((long long)m * n) / tau;
    return FG_OK;
}

// pooled lag autocovariances, fetched 32 lags at a time through the callback and cached
struct AcovCache {
    std::map<int, std::vector<double>> cache;                   // lag chunk -> [d][32] means over chains
    fg_acov_fn acov; void *user; int n, d; int64_t m;
    int get(int t, int i, double &out) {
        const int chunk = 32, k = t / chunk;
        auto it = cache.find(k);
        if (it == cache.end()) {
            const int lag0 = k * chunk, nl = std::min(chunk, n - lag0);
            std::vector<double> sums((size_t)d * chunk, 0.0), tmp((size_t)d * nl, 0.0);
            if (!acov) { fg_set_error("fg_diag_combine: ESS needs the autocovariance callback"); return FG_E_BAD_ARG; }
            const int rc = acov(user, lag0, nl, tmp.data());
            if (rc) return rc;
            for (int q = 0; q < d; q++) for (int l = 0; l < nl; l++) sums[(size_t)q * chunk + l] = tmp[(size_t)q * nl + l] / (double)m;
            it = cache.emplace(k, std::move(sums)).first;
        }
        out = it->second[(size_t)i * chunk + (t - k * chunk)];
        return FG_OK;
    }
};

extern "C" {

int fg_diag_combine(const double *h_moments /*[d][6][m]*/, int64_t m, int n, int d, fg_acov_fn acov, void *user,
                    double *h_rhat, double *h_ess, double *h_mean, double *h_std) {
    if (!h_moments || m < 0 || n < 0 || d <= 0) { fg_set_error("fg_diag_combine: bad argument"); return FG_E_BAD_ARG; }
    const int half = n / 2;
    AcovCache AC{ {}, acov, user, n, d, m };
    for (int i = 0; i < d; i++) {
        const double *mom = h_moments + (size_t)i * 6 * m;
        // ---- split R-hat over the 2m half-chains c0h0, c0h1, c1h0, ... (diagnostics.rs:218-224, 240-260); n < 2: classic
        if (h_rhat) {
            if (half == 0) h_rhat[i] = rhat_of(mom, mom + m, 1, m, (double)n);
            else {
                std::vector<double> means((size_t)2 * m), ssds((size_t)2 * m);
                for (int64_t j = 0; j < m; j++) { means[2 * j] = mom[2 * m + j]; means[2 * j + 1] = mom[4 * m + j]; ssds[2 * j] = mom[3 * m + j]; ssds[2 * j + 1] = mom[5 * m + j]; }
                h_rhat[i] = rhat_of(means.data(), ssds.data(), 1, 2 * m, (double)half);
            }
        }
        // ---- pooled mean / sample std of all m n values (summarize_f64_parameter, diagnostics.rs:331-352)
        double gm = 0.0;
        for (int64_t j = 0; j < m; j++) gm += mom[j];
        gm = m > 0 ? gm / (double)m : NAN;
        if (h_mean) h_mean[i] = gm;
        if (h_std) {
            double ss = 0.0, bs = 0.0;
            for (int64_t j = 0; j < m; j++) { ss += mom[m + j]; const double dv = mom[j] - gm; bs += dv * dv; }
            h_std[i] = std::sqrt((ss + (double)n * bs) / ((double)m * (double)n - 1.0));
        }
        // ---- multi-chain ESS (mcmc_utils.rs:253-339)
        if (!h_ess) continue;
        if (m == 0) { h_ess[i] = 0.0; continue; }
        if (n < 4) { h_ess[i] = (double)std::max<long long>((long long)m * n, 1); continue; }
        const double nf = (double)n, mf = (double)m;
        double vs = 0.0;
        for (int64_t j = 0; j < m; j++) vs += (mom[m + j] / nf) * nf / (nf - 1.0);      // acov0 * n / (n - 1)
        const double mean_var = vs / mf;
        double between = 0.0;
        if (m > 1) {
            double s = 0.0;
            for (int64_t j = 0; j < m; j++) s += mom[j];
            const double overall = s / mf;
            for (int64_t j = 0; j < m; j++) { const double dv = mom[j] - overall; between += dv * dv; }
        }
        const int rc = ess_from_pooled(m, n, mean_var, between, [&](int t, double &a) { return AC.get(t, i, a); }, h_ess[i]);
        if (rc) return rc;
    }
    return FG_OK;
}

// The same statistics from SUMS over chains only -- what ranks that shard the chains exchange with all-reduces of O(d) doubles
// instead of gathering every chain's moments (diagnostics.rs:262-304 and mcmc_utils.rs:253-339 need the
// chains only through these sums).  `reduce` is called twice per combination and returns sums over ALL chains of all ranks:
//   stage 1: out [d][6] = sum over chains of the six moment rows (mean, ssd of the full chain, of the first and of the second half);
//   stage 2: in [d][2] = the overall means {sum mean / m, (sum mean_h1 + sum mean_h2) / 2m};
//            out [d][2] = {sum_j (mean_j - in0)^2, sum_j (mean_h1_j - in1)^2 + (mean_h2_j - in1)^2}
// -- the reference's two-pass between-chain sums of squares, with the pass over chains distributed.  Chain sums are formed per
// rank and then added, so results agree with fg_diag_combine to rounding (~1e-15 relative), not bit for bit.
int fg_diag_combine_reduced(int64_t m, int n, int d, fg_reduce_fn reduce, fg_acov_fn acov, void *user,
                            double *h_rhat, double *h_ess, double *h_mean, double *h_std) {
    if (!reduce || m < 0 || n < 0 || d <= 0) { fg_set_error("fg_diag_combine_reduced: bad argument"); return FG_E_BAD_ARG; }
    const int half = n / 2;
    std::vector<double> s1((size_t)d * 6, 0.0), ov((size_t)d * 2, 0.0), s2((size_t)d * 2, 0.0);
    int rc = reduce(user, 1, nullptr, s1.data());
    if (rc) return rc;
    const double mf = (double)m;
    for (int i = 0; i < d; i++) { ov[2 * i] = m > 0 ? s1[6 * i] / mf : NAN; ov[2 * i + 1] = m > 0 ? (s1[6 * i + 2] + s1[6 * i + 4]) / (2.0 * mf) : NAN; }
    rc = reduce(user, 2, ov.data(), s2.data());
    if (rc) return rc;
    AcovCache AC{ {}, acov, user, n, d, m };
    for (int i = 0; i < d; i++) {
        const double *a = &s1[(size_t)6 * i];
        const double between_full = s2[2 * i], between_split = s2[2 * i + 1];
        if (h_rhat) h_rhat[i] = half == 0 ? rhat_from_sums(between_full, a[1], m, (double)n) : rhat_from_sums(between_split, a[3] + a[5], 2 * m, (double)half);
        if (h_mean) h_mean[i] = ov[2 * i];
        if (h_std) h_std[i] = std::sqrt((a[1] + (double)n * between_full) / (mf * (double)n - 1.0));
        if (!h_ess) continue;
        if (m == 0) { h_ess[i] = 0.0; continue; }
        if (n < 4) { h_ess[i] = (double)std::max<long long>((long long)m * n, 1); continue; }
        const double nf = (double)n;
        const double mean_var = (a[1] / (nf - 1.0)) / mf;                      // mean over chains of acov0 n / (n - 1) = ssd / (n - 1)
        rc = ess_from_pooled(m, n, mean_var, between_full, [&](int t, double &v) { return AC.get(t, i, v); }, h_ess[i]);
        if (rc) return rc;
    }
    return FG_OK;
}

}  // extern "C"
