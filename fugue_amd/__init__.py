"""fugue_amd: the MI355X-native many-chain engine for Fugue's `src/inference` hot path.

    from fugue_amd import sample, observe, factor, pure, addr, Normal, hmc_chain, HMCConfig
    model = lambda: sample(addr("mu"), Normal(0.0, 1.0)).bind(lambda mu: observe(addr("y"), Normal(mu, 0.5), 1.2).map(lambda _: mu))
    chains = hmc_chain(42, model, 1000, 500, HMCConfig(), n_chains=65536)
    chains.get_f64(addr("mu")).mean()          # 0.96
"""
from .model import (Bernoulli, Beta, Binomial, Categorical, Cauchy, ChiSquared, DiscreteUniform, Exponential, FugueError, Gamma,  # noqa: F401
                    InverseGamma, Laplace, LogNormal, Model, Normal, Poisson, Program, StudentT, Uniform, Weibull, addr, factor, guard,
                    observe, plate, pure, sample, sequence_vec, traverse_vec, zip_models)
from .inference import (ChainBatch, HMCConfig, ResamplingMethod, SMCConfig, SMCResult, SiteProposal, adaptive_mcmc_chain,  # noqa: F401
                        adaptive_mcmc_chain_with_overrides, adaptive_smc, hmc_chain)
from .diagnostics import (ParameterSummary, classic_r_hat_f64, effective_sample_size, effective_sample_size_multichain,  # noqa: F401
                          geweke_diagnostic, r_hat_f64, summarize_f64_parameter)
from .validation import effective_sample_size_mcmc  # noqa: F401
