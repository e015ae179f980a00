"""Cross-chain diagnostics over many chains on one or several GPUs.

Mirrors `r_hat_f64` / `classic_r_hat_f64` / `summarize_f64_parameter`
(/root/reference/src/inference/diagnostics.rs:218-391) and `effective_sample_size_multichain`
(/root/reference/src/inference/mcmc_utils.rs:214-339) for draws laid out [n][d][C] in HBM.

This is the ONLY place chains interact, hence the only collective of the engine:
every rank reduces its own draws to per-chain moments (`fg_diag_chain_moments`, [d][6][C_local])
and pooled per-lag autocovariance sums (`fg_diag_autocov_sums`, [d][lags]) on its GPU; ranks then
exchange only those -- `all_gather` of the chain means / sums of squares (R-hat needs every chain
mean) and `all_reduce` of the lag sums -- over RCCL/xGMI (`backend="nccl"`), or gloo in the CPU
tests.  The final formulas run on the host in float64 and follow the reference line by line.
"""
from __future__ import annotations

import math
from typing import Optional, Protocol

import numpy as np


class MomentProvider(Protocol):
    n: int          # draws per chain
    d: int          # coordinates
    def moments(self) -> np.ndarray: ...                       # [d][6][C_local]
    def autocov_sums(self, lag0: int, n_lags: int) -> np.ndarray: ...   # [d][n_lags]


class EngineMoments:
    """MomentProvider backed by the HIP kernels of an `engine.Engine` and a device draw buffer."""

    def __init__(self, engine, d_draws: int, n: int, d: int):
        self.engine, self.d_draws, self.n, self.d = engine, d_draws, int(n), int(d)
        self._d_mom = engine.device_alloc(max(1, d * 6 * engine.C) * 8)
        engine.diag_chain_moments(d_draws, n, d, self._d_mom)
        self._mom: Optional[np.ndarray] = None

    def moments(self) -> np.ndarray:
        if self._mom is None:
            self._mom = self.engine.download(self._d_mom, (self.d, 6, self.engine.C))
        return self._mom

    def autocov_sums(self, lag0: int, n_lags: int) -> np.ndarray:
        return self.engine.diag_autocov_sums(self.d_draws, self.n, self.d, self._d_mom, lag0, n_lags)

    def close(self):
        if self._d_mom:
            self.engine.device_free(self._d_mom)
            self._d_mom = 0


class HostMoments:
    """MomentProvider over host draws [n][d][C] (numpy): the same statistics the GPU kernels produce, for callers that
    already hold the draws on the host (fugue_amd.validation) and for CPU tests of the combination logic."""

    def __init__(self, draws: np.ndarray):
        self.x = np.asarray(draws, dtype=np.float64)
        self.n, self.d, self.C = self.x.shape

    def moments(self) -> np.ndarray:
        n, half = self.n, self.n // 2
        out = np.zeros((self.d, 6, self.C))
        for k, (a, b) in enumerate(((0, n), (0, half), (half, 2 * half))):
            seg = self.x[a:b]
            mean = seg.sum(axis=0) / max(1, b - a)
            out[:, 2 * k] = mean
            out[:, 2 * k + 1] = ((seg - mean) ** 2).sum(axis=0)
        return out

    def autocov_sums(self, lag0: int, n_lags: int) -> np.ndarray:
        c = self.x - self.x.mean(axis=0, keepdims=True)
        out = np.zeros((self.d, n_lags))
        for k in range(n_lags):
            lag = lag0 + k
            if lag < self.n:
                out[:, k] = ((c[:self.n - lag] * c[lag:]).sum(axis=0) / self.n).sum(axis=1)
        return out


# ---- collectives (identity when not distributed) ------------------------------------------
def _dist(group):
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def _all_gather_concat(x: np.ndarray, group=None, device=None) -> np.ndarray:
    """Concatenate `x` ([..., C_local]) of every rank along the last axis, in rank order."""
    dist = _dist(group)
    if dist is None or dist.get_world_size(group) == 1:
        return x
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x))
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, t, group=group)          # equal C_local on every rank (chains are sharded evenly)
    return np.concatenate([o.cpu().numpy() for o in outs], axis=-1)


def _all_reduce_sum(x: np.ndarray, group=None, device=None) -> np.ndarray:
    dist = _dist(group)
    if dist is None or dist.get_world_size(group) == 1:
        return x
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


# ---- formulas ------------------------------------------------------------------------------
def _rhat(means: np.ndarray, ssds: np.ndarray, n: int) -> float:
    """r_hat_from_f64_chains (diagnostics.rs:262-304) from per-chain means and sums of squared deviations."""
    m = means.shape[0]
    if m < 2:
        return 1.0
    if n == 0:
        return float("nan")
    overall = means.sum() / m
    b = n / (m - 1.0) * ((means - overall) ** 2).sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        w = (ssds / (n - 1.0)).sum() / m
        var_plus = ((n - 1.0) / n) * w + (1.0 / n) * b
        return float(np.sqrt(var_plus / w))


class ChainDiagnostics:
    """Diagnostics of `C_total` chains x `n` draws x `d` coordinates, sharded over the ranks of `group`."""

    def __init__(self, provider: MomentProvider, group=None, device=None):
        self.p, self.group, self.device = provider, group, device
        self.n, self.d = provider.n, provider.d
        self._mom = _all_gather_concat(provider.moments(), group, device)        # [d][6][C_total]
        self.m = self._mom.shape[2]
        self._acov_cache = {}

    # r_hat_f64: split-R-hat over 2m half-chains (diagnostics.rs:218-224, 240-260)
    def split_rhat(self) -> np.ndarray:
        half = self.n // 2
        if half == 0:
            return self.classic_rhat()
        out = np.empty(self.d)
        for i in range(self.d):
            means = np.stack([self._mom[i, 2], self._mom[i, 4]], axis=1).ravel()   # c0h0, c0h1, c1h0, ...
            ssds = np.stack([self._mom[i, 3], self._mom[i, 5]], axis=1).ravel()
            out[i] = _rhat(means, ssds, half)
        return out

    # classic_r_hat_f64 (diagnostics.rs:226-238)
    def classic_rhat(self) -> np.ndarray:
        return np.array([_rhat(self._mom[i, 0], self._mom[i, 1], self.n) for i in range(self.d)])

    def pooled_mean(self) -> np.ndarray:
        return self._mom[:, 0].mean(axis=1)

    def pooled_std(self) -> np.ndarray:
        """sample std of all m*n values (summarize_f64_parameter, diagnostics.rs:348-352)."""
        gm = self.pooled_mean()
        ss = self._mom[:, 1].sum(axis=1) + self.n * ((self._mom[:, 0] - gm[:, None]) ** 2).sum(axis=1)
        return np.sqrt(ss / (self.m * self.n - 1.0))

    def _acov_mean(self, t: int) -> np.ndarray:
        """mean over ALL chains of the lag-t autocovariance, fetched in chunks of 32 lags."""
        chunk = 32
        k = t // chunk
        if k not in self._acov_cache:
            lag0 = k * chunk
            n_l = min(chunk, self.n - lag0)
            sums = self.p.autocov_sums(lag0, n_l)
            self._acov_cache[k] = _all_reduce_sum(sums, self.group, self.device) / self.m
        return self._acov_cache[k][:, t - k * chunk]

    # effective_sample_size_multichain / ess_from_chains (mcmc_utils.rs:214-224, 253-339)
    def ess(self) -> np.ndarray:
        n, m, d = self.n, self.m, self.d
        if m == 0:
            return np.zeros(d)
        if n < 4:
            return np.full(d, float(max(m * n, 1)))
        max_lag = min(n - 1, 2048)
        nf, mf = float(n), float(m)
        out = np.empty(d)
        chain_means = self._mom[:, 0]                          # [d][m]
        chain_vars = (self._mom[:, 1] / nf) * nf / (nf - 1.0)  # acov0 * n/(n-1)
        for i in range(d):
            mean_var = chain_vars[i].sum() / mf
            if mean_var <= 0.0:
                out[i] = float(m * n)
                continue
            var_plus = mean_var * (nf - 1.0) / nf
            if m > 1:
                overall = chain_means[i].sum() / mf
                var_plus += ((chain_means[i] - overall) ** 2).sum() / (mf - 1.0)

            def rho(t, i=i, mean_var=mean_var, var_plus=var_plus):
                return 1.0 - (mean_var - self._acov_mean(t)[i]) / var_plus

            rho_hat = np.zeros(max_lag + 1)
            rho_hat[0] = 1.0
            if max_lag >= 1:
                rho_hat[1] = rho(1)
            t, max_t = 1, min(1, max_lag)
            while t + 2 <= max_lag:                             # Geyer initial positive sequence
                re, ro = rho(t + 1), rho(t + 2)
                if re + ro < 0.0:
                    break
                rho_hat[t + 1], rho_hat[t + 2] = re, ro
                max_t = t + 2
                t += 2
            k = 1
            while k + 2 <= max_t:                               # monotone pair sums
                prev = rho_hat[k - 1] + rho_hat[k]
                cur = rho_hat[k + 1] + rho_hat[k + 2]
                if cur > prev:
                    rho_hat[k + 1] = rho_hat[k + 2] = prev / 2.0
                k += 2
            tau = max(-1.0 + 2.0 * rho_hat[:max_t + 1].sum(), 1.0)
            out[i] = m * n / tau
        return out

    def summary(self) -> dict:
        return dict(mean=self.pooled_mean(), std=self.pooled_std(), r_hat=self.split_rhat(), ess=self.ess())


def quantiles_f64(values: np.ndarray, ps=(0.025, 0.25, 0.5, 0.75, 0.975)) -> dict:
    """summarize_f64_parameter's quantile rule: sorted[round((len-1) * p)] (diagnostics.rs:355-371)."""
    v = np.sort(np.asarray(values, dtype=np.float64).ravel())
    names = {0.025: "2.5%", 0.25: "25%", 0.5: "50%", 0.75: "75%", 0.975: "97.5%"}
    out = {}
    for p in ps:
        idx = int(math.floor((len(v) - 1) * p + 0.5))          # f64::round: half away from zero
        out[names.get(p, f"{100 * p:g}%")] = float(v[idx])
    return out


def geweke_diagnostic(chain: np.ndarray) -> float:
    """geweke_diagnostic (mcmc_utils.rs:354-421): single chain, host-side."""
    x = np.asarray(chain, dtype=np.float64)
    n = len(x)
    if n < 20:
        return float("nan")
    a, b = x[:n // 10], x[n // 2:]
    if len(a) < 2 or len(b) < 2:
        return float("nan")

    def spec_var_mean(seg):
        k = len(seg)
        mean = seg.sum() / k
        s2 = ((seg - mean) ** 2).sum() / (k - 1.0)
        if s2 == 0.0:
            return 0.0
        c = seg - mean
        max_lag = min(k - 1, 1024)
        var0 = (c * c).sum() / k
        if var0 <= 0.0:
            return 0.0
        tau = 1.0
        for lag in range(1, max_lag + 1):
            r = (c[:k - lag] * c[lag:]).sum() / k / var0
            if r <= 0.0:
                break
            tau += 2.0 * r
        return s2 * tau / k

    se = math.sqrt(spec_var_mean(a) + spec_var_mean(b))
    if se == 0.0:
        return 0.0
    return float((a.sum() / len(a) - b.sum() / len(b)) / se)


def format_diagnostics(names, cd: "ChainDiagnostics", draws: Optional[np.ndarray] = None) -> str:
    """`print_diagnostics` (diagnostics.rs:394-456) for many chains: the same table and convergence verdict.  `draws`
    [n][d][C] (host) supplies the 2.5 % / 50 % / 97.5 % quantiles (`quantiles_f64` rule); without it they print as NaN."""
    s = cd.summary()
    lines = ["MCMC Diagnostics:",
             "{:<15} {:>8} {:>8} {:>8} {:>8} {:>8} {:>8} {:>8}".format("Parameter", "Mean", "Std", "2.5%", "50%", "97.5%", "R-hat", "ESS"),
             "-" * 80]
    for i, name in enumerate(names):
        q = quantiles_f64(draws[:, i, :], (0.025, 0.5, 0.975)) if draws is not None else {}
        vals = [q.get(k, float("nan")) for k in ("2.5%", "50%", "97.5%")]
        lines.append("{:<15} {:>8.3f} {:>8.3f} {:>8.3f} {:>8.3f} {:>8.3f} {:>8.3f} {:>8.0f}".format(
            str(name), s["mean"][i], s["std"][i], vals[0], vals[1], vals[2], s["r_hat"][i], s["ess"][i]))
    fin = s["r_hat"][np.isfinite(s["r_hat"])]
    if fin.size:
        mx, avg = float(fin.max()), float(fin.mean())
        lines.append("")
        lines.append("Convergence Assessment:")
        if mx < 1.01:
            lines.append("✓ Excellent convergence (max R-hat = {:.3f})".format(mx))
        elif mx < 1.1:
            lines.append("⚠ Good convergence (max R-hat = {:.3f})".format(mx))
        else:
            lines.append("✗ Poor convergence (max R-hat = {:.3f}) - consider more samples".format(mx))
        lines.append("  Average R-hat: {:.3f}".format(avg))
    return "\n".join(lines)


def print_diagnostics(names, cd: "ChainDiagnostics", draws: Optional[np.ndarray] = None) -> None:
    print(format_diagnostics(names, cd, draws))
