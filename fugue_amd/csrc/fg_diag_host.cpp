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

int fg_diag_combine(const double *h_moments /*[d][6][m]*/, int64_t m, int n, int d, fg_acov_fn acov, void *user,
                    double *h_rhat, double *h_ess, double *h_mean, double *h_std) {
    if (!h_moments || m < 0 || n < 0 || d <= 0) { fg_set_error("fg_diag_combine: bad argument"); return FG_E_BAD_ARG; }
    const int half = n / 2;
    std::map<int, std::vector<double>> cache;                   // lag chunk -> [d][32] means over chains
    auto acov_mean = [&](int t, int i, double &out) -> int {
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
    };
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
        const int max_lag = std::min(n - 1, 2048);
        const double nf = (double)n, mf = (double)m;
        double vs = 0.0;
        for (int64_t j = 0; j < m; j++) vs += (mom[m + j] / nf) * nf / (nf - 1.0);      // acov0 * n / (n - 1)
        const double mean_var = vs / mf;
        if (mean_var <= 0.0) { h_ess[i] = (double)((long long)m * n); continue; }
        double var_plus = mean_var * (nf - 1.0) / nf;
        if (m > 1) {
            double s = 0.0;
            for (int64_t j = 0; j < m; j++) s += mom[j];
            const double overall = s / mf;
            double between = 0.0;
            for (int64_t j = 0; j < m; j++) { const double dv = mom[j] - overall; between += dv * dv; }
            var_plus += between / (mf - 1.0);
        }
        int rc = FG_OK;
        auto rho = [&](int t) { double a = 0.0; const int r = acov_mean(t, i, a); if (r) rc = r; return 1.0 - (mean_var - a) / var_plus; };
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
        h_ess[i] = (double)((long long)m * n) / tau;
    }
    return FG_OK;
}

}  // extern "C"
