"""Cross-chain diagnostics over many chains on one or several GPUs.

Mirrors `r_hat_f64` / `classic_r_hat_f64` / `summarize_f64_parameter`
(/root/reference/src/inference/diagnostics.rs:218-391) and `effective_sample_size_multichain`
(/root/reference/src/inference/mcmc_utils.rs:214-339) for draws laid out [n][d][C] in HBM.

This is the ONLY place chains interact, hence the only collective of the engine:
every rank reduces its own draws to per-chain moments (`fg_diag_chain_moments`, [d][6][C_local])
and pooled per-lag autocovariance sums (`fg_diag_autocov_sums`, [d][lags]) on its GPU.  Chains
enter split R-hat, the pooled moments and the ESS only through SUMS over chains, so the ranks
exchange `all_reduce`s of 6 d + 2 d doubles (+ 32 d per chunk of lags) -- over RCCL/xGMI
(`backend="nccl"`), or gloo in the CPU tests; `exchange="gather"` keeps the older `all_gather` of
every chain's moments (a single process's summation order).  The final formulas run in the
library in float64 and follow the reference line by line.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
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


# ---- the combination: C++ (fg_diag_combine, fugue_amd/csrc/fg_diag_host.cpp) ------------------------------------------
class ChainDiagnostics:
    """Diagnostics of `C_total` chains x `n` draws x `d` coordinates, sharded over the ranks of `group`: every rank's
    moments are all-gathered, pooled lag sums all-reduced (torch.distributed: RCCL on GPUs, gloo in the CPU tests), and the
    R-hat / Geyer-ESS formulas run in the library.  (A process that owns an RCCL communicator can skip this class and call
    `Engine.diag_rhat_ess(..., comm)`: the same computation with the collectives inside the library.)"""

    def __init__(self, provider: MomentProvider, group=None, device=None, exchange: str = "reduce"):
        from . import engine as E
        assert exchange in ("reduce", "gather")
        self.p, self.group, self.device, self.exchange = provider, group, device, exchange
        self.n, self.d = provider.n, provider.d
        self.exchange_bytes = 0                          # bytes this rank contributed to collectives
        if exchange == "gather":
            local = provider.moments()
            self._mom = _all_gather_concat(local, group, device)        # [d][6][C_total]
            self.m = self._mom.shape[2]
            if self.m != local.shape[2]:
                self.exchange_bytes += local.nbytes
        else:
            self._local = np.ascontiguousarray(provider.moments())      # [d][6][C_local]
            dist = _dist(group)
            world = dist.get_world_size(group) if dist is not None else 1
            self.m = self._local.shape[2] * world                       # chains are sharded evenly
            self._world = world
            self._sums = None
        self._res = None
        self._E = E

    def _acov(self, lag0: int, n_lags: int) -> np.ndarray:
        a = self.p.autocov_sums(lag0, n_lags)
        if _dist(self.group) is not None and _dist(self.group).get_world_size(self.group) > 1:
            self.exchange_bytes += a.nbytes
        return _all_reduce_sum(a, self.group, self.device)

    def _reduce(self, stage: int, overall):
        """Sums over this rank's chains, all-reduced: the chain sums of fg_diag_combine_reduced."""
        mom = self._local
        if stage == 1:
            a = mom.sum(axis=2)                                          # [d][6]
        else:
            a = np.stack([((mom[:, 0] - overall[:, 0:1]) ** 2).sum(axis=1),
                          ((mom[:, 2] - overall[:, 1:2]) ** 2 + (mom[:, 4] - overall[:, 1:2]) ** 2).sum(axis=1)], axis=1)    # [d][2]
        if self._world > 1:
            self.exchange_bytes += a.nbytes
        out = _all_reduce_sum(np.ascontiguousarray(a), self.group, self.device)
        if stage == 1:
            self._sums = [out, None]
        else:
            self._sums[1] = out
        return out

    def _combine(self):
        if self._res is None:
            if self.exchange == "gather":
                self._res = self._E.diag_combine(self._mom, self.n, self._acov)
            else:
                self._res = self._E.diag_combine_reduced(self.m, self.n, self.d, self._reduce, self._acov)
        return self._res

    def split_rhat(self) -> np.ndarray:                 # r_hat_f64 (diagnostics.rs:218-224, 240-260)
        return self._combine()["r_hat"]

    def classic_rhat(self) -> np.ndarray:               # classic_r_hat_f64 (diagnostics.rs:226-238): whole chains, no split
        m, n = self.m, float(self.n)
        out = np.empty(self.d)
        if self.exchange == "reduce":                   # from the chain sums the combination already exchanged
            self._combine()
            s1, s2 = self._sums
        for i in range(self.d):
            if m < 2:
                out[i] = 1.0
                continue
            if self.exchange == "gather":
                means, ssds = self._mom[i, 0], self._mom[i, 1]
                overall = means.sum() / m
                between, ssd = ((means - overall) ** 2).sum(), (ssds / (n - 1.0)).sum()
            else:
                between, ssd = s2[i, 0], s1[i, 1] / (n - 1.0)
            b = n / (m - 1.0) * between
            w = ssd / m
            out[i] = math.sqrt((((n - 1.0) / n) * w + b / n) / w) if w > 0 else float("nan")
        return out

    def pooled_mean(self) -> np.ndarray:
        return self._combine()["mean"]

    def pooled_std(self) -> np.ndarray:
        return self._combine()["std"]

    def ess(self) -> np.ndarray:                        # effective_sample_size_multichain (mcmc_utils.rs:214-339)
        return self._combine()["ess"]

    def summary(self) -> dict:
        r = self._combine()
        return dict(mean=r["mean"], std=r["std"], r_hat=r["r_hat"], ess=r["ess"])


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


# ---- the reference's per-parameter entry points over a many-chain result (fugue_amd.inference.ChainBatch) -----------------
@dataclass
class ParameterSummary:                 # diagnostics.rs:306-318
    mean: float
    std: float
    quantiles: dict
    r_hat: float
    ess: float


def _site_draws(chains, address: str) -> np.ndarray:
    """[n_samples][1][n_chains] f64 draws of one site (extract_f64_values, diagnostics.rs:206-216)."""
    return np.ascontiguousarray(chains.get_f64(address))[:, None, :]


def r_hat_f64(chains, address: str) -> float:
    """Split R-hat of one f64 site over the chains of a ChainBatch (r_hat_f64, diagnostics.rs:218-224, 240-260)."""
    return float(ChainDiagnostics(HostMoments(_site_draws(chains, address))).split_rhat()[0])


def classic_r_hat_f64(chains, address: str) -> float:
    """classic_r_hat_f64 (diagnostics.rs:226-238): whole chains, no split."""
    return float(ChainDiagnostics(HostMoments(_site_draws(chains, address))).classic_rhat()[0])


def effective_sample_size_multichain(values: np.ndarray) -> float:
    """effective_sample_size_multichain (mcmc_utils.rs:214-229) of draws [n_samples][n_chains]."""
    x = np.asarray(values, dtype=np.float64)
    return float(ChainDiagnostics(HostMoments(x[:, None, :])).ess()[0])


def summarize_f64_parameter(chains, address: str) -> ParameterSummary:
    """summarize_f64_parameter (diagnostics.rs:320-392): pooled mean / std (n - 1) / quantiles, split R-hat, multi-chain ESS."""
    x = _site_draws(chains, address)
    allv = x.ravel()
    if allv.size == 0:
        return ParameterSummary(float("nan"), float("nan"), {}, float("nan"), 0.0)
    cd = ChainDiagnostics(HostMoments(x))
    mean = float(allv.sum() / allv.size)
    std = float(np.sqrt(((allv - mean) ** 2).sum() / (allv.size - 1))) if allv.size > 1 else float("nan")
    return ParameterSummary(mean, std, quantiles_f64(allv), float(cd.split_rhat()[0]), float(cd.ess()[0]))


def effective_sample_size(particles) -> float:
    """effective_sample_size (smc.rs:230-233) of a weighted population (SMCResult): 1 / sum w_i^2."""
    w = np.asarray(particles.weights, dtype=np.float64)
    return float(1.0 / (w * w).sum()) if w.size else 0.0
