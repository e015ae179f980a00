"""Statistical validation harness -- host-side mirror of /root/reference/src/inference/validation.rs
for many-chain output: the two-sample Kolmogorov-Smirnov check of a sampler (:17-70), the conjugate
Normal-Normal and Beta-Bernoulli posterior checkers (:73-162) and their shared scoring rule (:169-230:
sample mean / variance within 2 Monte-Carlo standard errors computed from the effective sample size, and
at least 10 % sampling efficiency), applied chain by chain exactly as the reference applies it to its one
chain, plus a pooled verdict over all chains (multichain ESS, mcmc_utils.rs:214-339)."""
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from .diagnostics import ChainDiagnostics, HostMoments


# ---- Kolmogorov-Smirnov (validation.rs:17-70) ------------------------------------------------
def ks_statistic(sample1: np.ndarray, sample2: np.ndarray) -> float:
    """`ks_statistic` (validation.rs:47-70) on two SORTED samples: the merge walk with the (i+1)/n convention."""
    s1, s2 = np.asarray(sample1, dtype=np.float64), np.asarray(sample2, dtype=np.float64)
    n1, n2 = float(len(s1)), float(len(s2))
    i1 = i2 = 0
    max_diff = 0.0
    while i1 < len(s1) and i2 < len(s2):
        max_diff = max(max_diff, abs((i1 + 1) / n1 - (i2 + 1) / n2))
        if s1[i1] <= s2[i2]:
            i1 += 1
        else:
            i2 += 1
    return max_diff


def ks_test_distribution(our_samples: Sequence[float], reference_samples: Sequence[float], alpha: float) -> bool:
    """`ks_test_distribution` (validation.rs:17-44) with the draws already made (the engine's prior sampler fills
    `our_samples`): statistic below the two-sample critical value sqrt(-ln(alpha)/2) * sqrt((n1+n2)/(n1 n2))."""
    ours, ref = np.sort(np.asarray(our_samples, dtype=np.float64)), np.sort(np.asarray(reference_samples, dtype=np.float64))
    n1, n2 = float(len(ours)), float(len(ref))
    critical = np.sqrt(-0.5 * np.log(alpha)) * np.sqrt((n1 + n2) / (n1 * n2))
    return ks_statistic(ours, ref) < critical


# ---- configs (validation.rs:73-86, 122-133) --------------------------------------------------
@dataclass
class ConjugateNormalConfig:
    prior_mu: float
    prior_sigma: float
    likelihood_sigma: float
    observation: float
    n_samples: int
    n_warmup: int

    def posterior(self):
        pp, lp = 1.0 / (self.prior_sigma * self.prior_sigma), 1.0 / (self.likelihood_sigma * self.likelihood_sigma)
        var = 1.0 / (pp + lp)
        return var * (pp * self.prior_mu + lp * self.observation), var


@dataclass
class ConjugateBetaBernoulliConfig:
    prior_alpha: float
    prior_beta: float
    observations: List[bool] = field(default_factory=list)
    n_samples: int = 0
    n_warmup: int = 0

    def posterior(self):
        s, n = float(sum(bool(b) for b in self.observations)), float(len(self.observations))
        a, b = self.prior_alpha + s, self.prior_beta + (n - s)
        t = a + b
        return a / t, (a * b) / (t * t * (t + 1.0))


# ---- result (validation.rs:232-310) ----------------------------------------------------------
@dataclass
class ValidationResult:
    failed: Optional[str] = None
    mean_error: float = 0.0
    var_error: float = 0.0
    effective_sample_size: float = 0.0
    mean_within_bounds: bool = False
    var_within_bounds: bool = False
    ess_adequate: bool = False
    posterior_mu: float = 0.0
    posterior_sigma: float = 0.0
    sample_mean: float = 0.0
    sample_sigma: float = 0.0

    def is_valid(self) -> bool:
        return self.failed is None and self.mean_within_bounds and self.var_within_bounds and self.ess_adequate

    def summary(self) -> str:
        if self.failed is not None:
            return f"Validation FAILED: {self.failed}"
        pf = lambda ok: "PASS" if ok else "FAIL"
        return "\n".join([
            "Validation Results:",
            f"  True posterior: N({self.posterior_mu:.4f}, {self.posterior_sigma:.4f})",
            f"  Sample estimates: N({self.sample_mean:.4f}, {self.sample_sigma:.4f})",
            f"  Mean error: {self.mean_error:.6f} ({pf(self.mean_within_bounds)})",
            f"  Var error: {self.var_error:.6f} ({pf(self.var_within_bounds)})",
            f"  ESS: {self.effective_sample_size:.1f} ({pf(self.ess_adequate)})",
            f"  Overall: {pf(self.is_valid())}"])

    def print_summary(self) -> None:
        print(self.summary())


def effective_sample_size_mcmc(chain: np.ndarray) -> float:
    """mcmc_utils.rs:190-201: n for n < 4, else ess_from_chains of the single chain."""
    x = np.asarray(chain, dtype=np.float64).ravel()
    if x.size < 4:
        return float(x.size)
    return float(ChainDiagnostics(HostMoments(x[:, None, None])).ess()[0])


def _score(sample_mean, sample_var, ess, posterior_mu, posterior_variance, n_samples) -> ValidationResult:
    sigma = np.sqrt(posterior_variance)
    se_mean, se_var = sigma / np.sqrt(ess), posterior_variance * np.sqrt(2.0 / ess)
    me, ve = abs(sample_mean - posterior_mu), abs(sample_var - posterior_variance)
    return ValidationResult(None, me, ve, ess, bool(me < 2.0 * se_mean), bool(ve < 2.0 * se_var), bool(ess > n_samples * 0.1),
                            posterior_mu, float(sigma), float(sample_mean), float(np.sqrt(sample_var)))


def validate_against_analytical_posterior(param_samples: np.ndarray, posterior_mu: float, posterior_variance: float,
                                          n_samples: int) -> ValidationResult:
    """validation.rs:169-230 for ONE chain of draws of the parameter."""
    x = np.asarray(param_samples, dtype=np.float64).ravel()
    if x.size == 0:
        return ValidationResult(failed="No samples extracted")
    mean = x.sum() / x.size
    var = ((x - mean) ** 2).sum() / (x.size - 1)
    return _score(mean, var, effective_sample_size_mcmc(x), posterior_mu, posterior_variance, n_samples)


@dataclass
class ChainBatchValidation:
    per_chain: List[ValidationResult]
    pooled: ValidationResult

    @property
    def fraction_valid(self) -> float:
        return sum(r.is_valid() for r in self.per_chain) / max(1, len(self.per_chain))


def validate_chains(draws: np.ndarray, posterior_mu: float, posterior_variance: float, n_samples: int,
                    max_chains: int = 256) -> ChainBatchValidation:
    """`draws` [n][C]: the reference's rule chain by chain (first `max_chains` chains) and once for the pooled draws
    with the multichain ESS -- the verdict a many-chain run is judged by."""
    x = np.asarray(draws, dtype=np.float64)
    n, C = x.shape
    per = [validate_against_analytical_posterior(x[:, c], posterior_mu, posterior_variance, n_samples) for c in range(min(C, max_chains))]
    cd = ChainDiagnostics(HostMoments(x[:, None, :]))
    mean = x.mean()
    var = ((x - mean) ** 2).sum() / (x.size - 1)
    pooled = _score(mean, var, float(cd.ess()[0]), posterior_mu, posterior_variance, n_samples * C)
    return ChainBatchValidation(per, pooled)


def test_conjugate_normal_model(mcmc_fn: Callable[[int, int], np.ndarray], config: ConjugateNormalConfig) -> ChainBatchValidation:
    """validation.rs:92-118: `mcmc_fn(n_samples, n_warmup)` returns the draws [n][C] of the site "mu"."""
    mu, var = config.posterior()
    return validate_chains(mcmc_fn(config.n_samples, config.n_warmup), mu, var, config.n_samples)


def test_conjugate_beta_bernoulli_model(mcmc_fn: Callable[[int, int], np.ndarray], config: ConjugateBetaBernoulliConfig) -> ChainBatchValidation:
    """validation.rs:144-162: `mcmc_fn` returns the draws [n][C] of the site "theta"."""
    mu, var = config.posterior()
    return validate_chains(mcmc_fn(config.n_samples, config.n_warmup), mu, var, config.n_samples)


test_conjugate_normal_model.__test__ = False          # not pytest tests: the reference's public names
test_conjugate_beta_bernoulli_model.__test__ = False
