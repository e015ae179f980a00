"""The three drivers of the reference's `src/inference` under their own names, over many chains:

    hmc_chain(seed, model_fn, n_samples, n_warmup, config, n_chains)          hmc.rs:566-583
    adaptive_mcmc_chain(seed, model_fn, n_samples, n_warmup, n_chains)        mh.rs:921-944
    adaptive_mcmc_chain_with_overrides(..., overrides)                        mh.rs:946-1014
    adaptive_smc(seed, num_particles, model_fn, config)                       smc.rs:455-581

`model_fn` is what the reference passes (`Fn() -> Model<A>`): here a zero-argument callable returning a
`fugue_amd.model.Model`, or an already traced `Program`.  The reference threads `&mut R`; the engine's RNG is
counter-based, so a `seed` replaces it (results do not depend on how chains are sharded).  Results come back as
`ChainBatch` / `SMCResult`, the many-chain form of `Vec<(A, Trace)>` / `Vec<Particle>`: `get_f64(addr)` etc. return
every chain's values of a site."""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import engine as E
from . import model as M


@dataclass
class HMCConfig:                      # hmc.rs:106-135, same defaults
    n_leapfrog: int = 16
    target_accept: float = 0.8
    init_step_size: Optional[float] = None
    finite_diff_eps: float = 1e-5
    adapt_mass: bool = False
    grad_mode: int = E.GRAD_FD_SPARSE  # engine extension: FD_DENSE is the reference's arithmetic verbatim

    def raw(self):
        return E.hmc_config(self.n_leapfrog, self.target_accept, self.init_step_size, self.finite_diff_eps, self.adapt_mass, self.grad_mode)


class ResamplingMethod:               # smc.rs:236-241
    Multinomial, Systematic, Stratified = E.RESAMPLE_MULTINOMIAL, E.RESAMPLE_SYSTEMATIC, E.RESAMPLE_STRATIFIED


@dataclass
class SMCConfig:                      # smc.rs:316-337
    resampling_method: int = ResamplingMethod.Systematic
    ess_threshold: float = 0.5
    rejuvenation_steps: int = 0


@dataclass
class SiteProposal:                   # mh.rs:145-161
    kind: int = E.PROP_AUTO
    lower: float = 0.0
    upper: float = 0.0

    @staticmethod
    def Gaussian(): return SiteProposal(E.PROP_GAUSSIAN)
    @staticmethod
    def LogSpace(): return SiteProposal(E.PROP_LOGSPACE)
    @staticmethod
    def Reflect(lower, upper): return SiteProposal(E.PROP_REFLECT, float(lower), float(upper))
    @staticmethod
    def PriorResample(): return SiteProposal(E.PROP_PRIOR_RESAMPLE)


@dataclass
class ChainBatch:
    """`n_samples` post-warmup states of `n_chains` chains: the many-chain `Vec<(A, Trace)>`."""
    sites: List[str]
    vtypes: List[int]
    cells: np.ndarray                 # [n_samples][n_sites][n_chains] int64 cells (f64 bits or integers)
    accept_rate: float = 0.0
    mean_step_size: float = float("nan")
    n_divergent: int = 0

    def _row(self, address: str) -> int:
        if address not in self.sites:
            raise M.FugueError(f"address not found: {address}", M.ErrorCode.TraceAddressNotFound)
        return self.sites.index(address)

    def get_f64(self, address: str) -> np.ndarray:            # Trace::get_f64 for every draw and chain: [n_samples][n_chains]
        j = self._row(address)
        if self.vtypes[j] != 0:
            raise M.FugueError(f"site {address} is not f64", M.ErrorCode.TypeMismatch)
        return np.ascontiguousarray(self.cells[:, j, :]).view(np.float64)

    def get_int(self, address: str) -> np.ndarray:            # get_bool / get_u64 / get_usize / get_i64
        j = self._row(address)
        if self.vtypes[j] == 0:
            raise M.FugueError(f"site {address} is f64", M.ErrorCode.TypeMismatch)
        return self.cells[:, j, :]


@dataclass
class SMCResult:
    sites: List[str]
    vtypes: List[int]
    cells: np.ndarray                 # [n_sites][n_particles]
    weights: np.ndarray
    log_weights: np.ndarray
    log_evidence: float
    betas: np.ndarray = field(default_factory=lambda: np.zeros(0))

    def get_f64(self, address: str) -> np.ndarray:
        return np.ascontiguousarray(self.cells[self.sites.index(address)]).view(np.float64)


def _compile(model_fn) -> E.CompiledProgram:
    return model_fn if isinstance(model_fn, E.CompiledProgram) else E.compile_model(model_fn)


def hmc_chain(seed: int, model_fn, n_samples: int, n_warmup: int, config: Optional[HMCConfig] = None, n_chains: int = 1,
              device: int = 0) -> ChainBatch:
    cp = _compile(model_fn)
    cfg = config or HMCConfig()
    eng = E.Engine(cp, n_chains, seed=seed, device=device)
    cells = np.zeros((n_samples, cp.S, n_chains), dtype=np.int64)
    if cp.d == 0:                                               # no continuous site: every step is a fresh prior draw (hmc.rs:826-845)
        eng.hmc_init(cfg.raw(), n_warmup)
        eng.hmc_step(n_warmup)
        for t in range(n_samples):
            eng.hmc_step(1)
            cells[t] = eng.get_values()
        st = eng.hmc_stats()
    else:
        buf = eng.device_alloc(max(1, n_samples * cp.d * n_chains) * 8)
        st = eng.hmc_run(cfg.raw(), n_samples, n_warmup, buf)
        draws = eng.download(buf, (n_samples, cp.d, n_chains), dtype=np.int64)
        eng.device_free(buf)
        cells[:] = eng.get_values()[None]                       # HMC moves the f64 sites; discrete sites keep their prior draw (hmc.rs:238-260)
        cells[:, cp.f64_sites, :] = draws
    out = ChainBatch(list(cp.site_names), list(cp.site_vtypes), cells, st.accept_rate, st.mean_step_size, int(st.n_divergent))
    eng.close()
    return out


def adaptive_mcmc_chain_with_overrides(seed: int, model_fn, n_samples: int, n_warmup: int, overrides: Sequence[Tuple[str, SiteProposal]],
                                       n_chains: int = 1, device: int = 0) -> ChainBatch:
    cp = _compile(model_fn)
    ov = None
    if overrides:
        ov = [None] * cp.S
        for a, p in overrides:
            if a not in cp.site_names:
                raise M.FugueError(f"address not found: {a}", M.ErrorCode.TraceAddressNotFound)
            ov[cp.site_names.index(a)] = (p.kind, p.lower, p.upper)
    eng = E.Engine(cp, n_chains, seed=seed, device=device)
    rec = list(range(cp.S))
    buf = eng.device_alloc(max(1, n_samples * cp.S * n_chains) * 8)
    st = eng.mh_run(n_samples, n_warmup, ov, rec, buf)
    cells = eng.download(buf, (n_samples, cp.S, n_chains), dtype=np.int64)
    eng.device_free(buf)
    out = ChainBatch(list(cp.site_names), list(cp.site_vtypes), cells, st.accept_rate)
    eng.close()
    return out


def adaptive_mcmc_chain(seed: int, model_fn, n_samples: int, n_warmup: int, n_chains: int = 1, device: int = 0) -> ChainBatch:
    return adaptive_mcmc_chain_with_overrides(seed, model_fn, n_samples, n_warmup, (), n_chains, device)


def adaptive_smc(seed: int, num_particles: int, model_fn, config: Optional[SMCConfig] = None, device: int = 0) -> SMCResult:
    cp = _compile(model_fn)
    cfg = config or SMCConfig()
    if num_particles == 0:                                         # smc.rs:462-467
        return SMCResult(list(cp.site_names), list(cp.site_vtypes), np.zeros((cp.S, 0), dtype=np.int64), np.zeros(0), np.zeros(0), 0.0)   # empty population: log_evidence 0.0
    eng = E.Engine(cp, num_particles, seed=seed, device=device)
    r = eng.smc_run(cfg.resampling_method, cfg.ess_threshold, cfg.rejuvenation_steps)
    out = SMCResult(list(cp.site_names), list(cp.site_vtypes), r["values"], r["weights"], r["log_w"], r["log_evidence"], r["betas"])
    eng.close()
    return out
