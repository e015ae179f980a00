"""Incremental many-chain sessions: the `new / step(n) / values_since` protocol of the reference's interactive front-end
(crates/fugue-wasm/src/mh.rs:44-283, hmc.rs:46-156) over the engine's `fg_mh_init / fg_mh_step` and `fg_hmc_init / fg_hmc_step`.
A session owns one engine; `step(n)` advances every chain by n transitions in one launch and appends the n new states of every
site to a bounded host-side history (the reference keeps `max_history` traces per chain and drops the oldest half when full).
Diagnostics come from the library's combination (`fugue_amd.diagnostics`)."""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np

from . import diagnostics as D
from . import engine as E
from . import model as M


def _compile(model, data_json: Optional[str]):
    if isinstance(model, str):                                  # the `prob!`-subset DSL (crates/fugue-wasm/src/dsl.rs:10-35)
        return E.CompiledProgram.from_dsl(model, data_json)
    return E.compile_model(model)


class _History:
    """[steps retained][sites][chains] 8-byte cells; `dropped` = states discarded from the front (mh.rs:120-127)."""

    def __init__(self, n_sites: int, n_chains: int, max_history: int):
        self.buf = np.zeros((0, n_sites, n_chains), dtype=np.int64)
        self.max_history, self.dropped = int(max_history), 0

    def append(self, cells: np.ndarray):
        self.buf = np.concatenate([self.buf, cells], axis=0)
        if self.buf.shape[0] > self.max_history:
            drop = self.max_history // 2
            self.buf = self.buf[drop:]
            self.dropped += drop

    def since(self, start: int) -> np.ndarray:
        return self.buf[max(0, start - self.dropped):]


class MhSession:
    """WasmMh (crates/fugue-wasm/src/mh.rs:44-283) for many chains: chains start from prior draws, every step is
    `adaptive_single_site_mh` (the adaptation never stops), every state is kept."""

    def __init__(self, model: Union[str, object], data_json: Optional[str] = None, n_chains: int = 4, seed: int = 0, device: int = 0,
                 max_history: int = 20000):
        self.cp = _compile(model, data_json)
        self.eng = E.Engine(self.cp, int(n_chains), seed=int(seed), device=device)
        self.eng.mh_init(2 ** 31 - 1)                            # DiminishingAdaptation for ever (mh.rs:66-70, 104-113)
        self.eng.mh_set_recording(True)
        self.C, self.S = int(n_chains), self.cp.S
        self.steps = 0
        self.hist = _History(self.S, self.C, max_history)
        self._names: List[str] = list(self.cp.site_names)

    def close(self):
        self.eng.close()

    def step(self, n: int) -> int:                               # mh.rs:94-128
        n = int(n)
        if n > 0 and self.S > 0:
            buf = self.eng.device_alloc(n * self.S * self.C * 8)
            self.eng.mh_step(n, list(range(self.S)), buf)
            self.hist.append(self.eng.download(buf, (n, self.S, self.C), dtype=np.int64))
            self.eng.device_free(buf)
        self.steps += n
        return self.steps

    def n_chains(self) -> int:
        return self.C

    def total_steps(self) -> int:
        return self.steps

    def site_names(self) -> List[str]:                           # f64 sites in trace order (mh.rs:142-150)
        return [n for n, vt in zip(self._names, self.cp.site_vtypes) if vt == 0]

    def _row(self, site: str) -> Optional[int]:
        return self._names.index(site) if site in self._names and self.cp.site_vtypes[self._names.index(site)] == 0 else None

    def values_since(self, site: str, chain: int, start: int) -> np.ndarray:   # mh.rs:155-168
        j = self._row(site)
        if j is None or not (0 <= chain < self.C):
            return np.zeros(0)
        return np.ascontiguousarray(self.hist.since(int(start))[:, j, chain]).view(np.float64)

    def current_values(self, site: str) -> np.ndarray:          # mh.rs:171-177
        j = self._row(site)
        if j is None:
            return np.full(self.C, np.nan)
        return np.ascontiguousarray(self.eng.get_values()[j]).view(np.float64)

    def log_weights(self) -> np.ndarray:                         # mh.rs:180-185
        return self.eng.mh_log_weight()

    def _draws(self, site: str, window: int = 0) -> Optional[np.ndarray]:
        j = self._row(site)
        if j is None or self.hist.buf.shape[0] == 0:
            return None
        x = np.ascontiguousarray(self.hist.buf[:, j, :]).view(np.float64)
        return x[-window:] if window > 0 and x.shape[0] > window else x

    def r_hat(self, site: str, window: int = 0) -> float:       # mh.rs:190-206: split R-hat over the last `window` retained draws
        x = self._draws(site, window)
        return float("nan") if x is None else float(D.ChainDiagnostics(D.HostMoments(x[:, None, :])).split_rhat()[0])

    def ess(self, site: str) -> float:                           # mh.rs:210-218
        x = self._draws(site)
        return 0.0 if x is None else D.effective_sample_size_multichain(x)

    def acceptance_rate(self) -> float:                          # mh.rs:222-235
        return float(self.eng.mh_stats().accept_rate) if self.steps > 0 else float("nan")

    def set_value(self, chain: int, site: str, value: float):   # mh.rs:239-255 (the engine re-scores the trace at its next step)
        j = self._row(site)
        if j is None or not (0 <= chain < self.C):
            return
        v = self.eng.get_values()
        v[j, chain] = np.float64(value).view(np.int64)
        self.eng.set_values(v)

    def summary(self, site: str) -> List[float]:                # mh.rs:276-281: [mean, std, r_hat, ess]
        x = self._draws(site)
        if x is None:
            return [float("nan"), float("nan"), float("nan"), 0.0]
        cd = D.ChainDiagnostics(D.HostMoments(x[:, None, :]))
        allv = x.ravel()
        mean = float(allv.mean())
        std = float(allv.std(ddof=1)) if allv.size > 1 else float("nan")
        return [mean, std, float(cd.split_rhat()[0]), float(cd.ess()[0])]

    def warnings(self) -> List[str]:                             # mh.rs:284-286
        return list(getattr(self.cp, "warnings", []))


class HmcSession:
    """WasmHmc (crates/fugue-wasm/src/hmc.rs:46-156) = HmcSession (src/inference/hmc.rs:643-920) for many chains."""

    def __init__(self, model: Union[str, object], data_json: Optional[str] = None, n_chains: int = 1, seed: int = 0, n_warmup: int = 200,
                 n_leapfrog: int = 16, adapt_mass: bool = False, device: int = 0, max_history: int = 20000):
        self.cp = _compile(model, data_json)
        self.eng = E.Engine(self.cp, int(n_chains), seed=int(seed), device=device)
        self.cfg = E.hmc_config(n_leapfrog=int(n_leapfrog), adapt_mass=bool(adapt_mass))
        self.eng.hmc_init(self.cfg, int(n_warmup))
        self.n_warmup = int(n_warmup)
        self.C, self.d = int(n_chains), self.cp.d
        self.hist = _History(self.d, self.C, max_history)
        self._names = [self.cp.site_names[j] for j in self.cp.f64_sites]

    def close(self):
        self.eng.close()

    def site_names(self) -> List[str]:
        return list(self._names)

    def step(self, n: int) -> int:                               # hmc.rs:104-115: the state after EVERY transition is kept, warmup included (push_history)
        n = int(n)
        if n > 0:
            pos, _ = self.eng.hmc_step_info(n)                   # [n][d][C]: each transition's state (fg_hmc_step_info)
            if self.d > 0:
                self.hist.append(np.ascontiguousarray(pos).view(np.int64))
        return self.eng.hmc_iterations()

    def step_recorded(self, chain: int = 0):                     # hmc.rs:83-101: the next transition's leapfrog path of one chain; its state joins the history
        L = int(self.cfg.n_leapfrog)
        traj, ham, npts = self.eng.hmc_step_recorded([int(chain)], L)
        if self.d > 0:
            cells = self.eng.get_values()
            self.hist.append(np.ascontiguousarray(cells[self.cp.f64_sites])[None, :, :])
        return dict(positions=traj[0, :npts[0]], hamiltonians=ham[0, :npts[0]], n_points=int(npts[0]))

    def set_step_size(self, eps: float):                         # hmc.rs:118-121: pins the step size and ends the warmup (hmc.rs:741-747)
        self.eng.hmc_set_step_size(float(eps))
        self.n_warmup = min(self.n_warmup, self.eng.hmc_iterations())

    def set_n_leapfrog(self, l: int):                            # hmc.rs:123-126
        self.eng.hmc_set_n_leapfrog(int(l))
        self.cfg.n_leapfrog = int(l)

    def is_warming_up(self) -> bool:                             # hmc.rs:128-131
        return self.eng.hmc_is_warming_up()

    def step_size(self) -> np.ndarray:                           # hmc.rs:133-136 (one per chain)
        return self.eng.hmc_step_sizes()

    def values(self, site: str) -> np.ndarray:                   # hmc.rs:138-146: [retained states][chains], every transition since the session began
        if site not in self._names:
            return np.zeros((0, self.C))
        return np.ascontiguousarray(self.hist.buf[:, self._names.index(site), :]).view(np.float64)

    def ess(self, site: str) -> float:                           # hmc.rs:148-151
        x = self.values(site)
        return 0.0 if x.shape[0] == 0 else D.effective_sample_size_multichain(x)
