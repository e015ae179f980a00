//! The reference's drivers over many chains on one MI355X: `hmc_chain` (`src/inference/hmc.rs:566-583`),
//! `adaptive_mcmc_chain[_with_overrides]` (`src/inference/mh.rs:921-1014`) and `adaptive_smc`
//! (`src/inference/smc.rs:455-581`).  The reference threads `&mut R`; the engine's generator is counter-based
//! (Philox keyed by (seed, global chain id, iteration)), so a `seed` takes its place and results do not depend on how chains
//! are sharded over GPUs.  Results are materialised as the reference's `(A, Trace)` pairs only on request: on the device a
//! trace is a column of the `[sites x chains]` array.
//! UNVERIFIED SOURCE -- never compiled.
use std::collections::HashMap;

use fugue::inference::smc::{Particle, SMCConfig, SMCResult, ResamplingMethod};
use fugue::inference::mh::SiteProposal;
use fugue::runtime::handler::run;
use fugue::runtime::interpreters::ScoreGivenTrace;
use fugue::*;

use crate::ffi::*;
use crate::flatten::{flatten, FlatProgram};

fn check(rc: i32) -> FugueResult<()> {
    if rc == 0 { return Ok(()); }
    Err(FugueError::ModelError { address: None, reason: format!("fugue_amd error {rc}: {}", last_error()),
                                 code: if rc == 301 { ErrorCode::AddressConflict } else { ErrorCode::UnexpectedModelStructure }, context: Default::default() })
}

struct Engine(*mut fg_engine);
impl Drop for Engine { fn drop(&mut self) { unsafe { fg_engine_free(self.0) } } }

pub struct GpuBackend { pub device: i32, pub n_chains: i64, pub chain_offset: u32 }

/// `n_samples` post-warmup states of `n_chains` chains: the many-chain `Vec<(A, Trace)>`.  `cells[draw][site row][chain]`
/// are raw 8-byte trace cells (f64 bits or integers) in the engine's site order (= `BTreeMap<Address, _>` order).
pub struct ChainDraws {
    pub sites: Vec<Address>,        // engine row order
    pub vtypes: Vec<i32>,
    pub n_samples: usize,
    pub n_chains: usize,
    pub cells: Vec<i64>,
    pub accept_rate: f64,
}
impl ChainDraws {
    fn cell(&self, draw: usize, row: usize, chain: usize) -> i64 { self.cells[(draw * self.sites.len() + row) * self.n_chains + chain] }
    /// `Trace::get_f64(addr)` for every draw of one chain.
    pub fn get_f64(&self, addr: &Address, chain: usize) -> Option<Vec<f64>> {
        let row = self.sites.iter().position(|a| a == addr)?;
        if self.vtypes[row] != FG_F64 { return None; }
        Some((0..self.n_samples).map(|t| f64::from_bits(self.cell(t, row, chain) as u64)).collect())
    }
    /// The unscored trace (values only) of one state.
    pub fn base_trace(&self, draw: usize, chain: usize) -> Trace {
        let mut t = Trace::default();
        for (row, addr) in self.sites.iter().enumerate() {
            let c = self.cell(draw, row, chain);
            let v = match self.vtypes[row] {
                FG_F64 => ChoiceValue::F64(f64::from_bits(c as u64)), FG_BOOL => ChoiceValue::Bool(c != 0), FG_U64 => ChoiceValue::U64(c as u64),
                FG_USIZE => ChoiceValue::Usize(c as usize), _ => ChoiceValue::I64(c),
            };
            t.insert_choice(addr.clone(), v, 0.0);
        }
        t
    }
    /// `(A, Trace)` of one state, exactly what `hmc_chain` / `adaptive_mcmc_chain` push: the model re-scored at the stored
    /// values (`score_full`, hmc.rs:283-299) -- fresh log-probabilities, `total_log_weight` valid.
    pub fn state<A>(&self, model_fn: &impl Fn() -> Model<A>, draw: usize, chain: usize) -> (A, Trace) {
        run(ScoreGivenTrace { base: self.base_trace(draw, chain), trace: Trace::default() }, model_fn())
    }
    /// The reference's return shape for one chain: `Vec<(A, Trace)>` of its `n_samples` states.
    pub fn chain<A>(&self, model_fn: &impl Fn() -> Model<A>, chain: usize) -> Vec<(A, Trace)> {
        (0..self.n_samples).map(|t| self.state(model_fn, t, chain)).collect()
    }
}

impl GpuBackend {
    fn engine(&self, prog: &FlatProgram, seed: u64) -> FugueResult<Engine> {
        let e = unsafe { fg_engine_new(prog.raw, self.n_chains, seed, self.chain_offset, self.device) };
        if e.is_null() { check(FG_E_NO_DEVICE)?; }
        Ok(Engine(e))
    }
    fn sorted_sites(prog: &FlatProgram) -> (Vec<Address>, Vec<i32>) {
        let s = prog.sites.len();
        let mut sites = vec![Address::new(""); s];
        let mut vt = vec![0i32; s];
        for h in 0..s { sites[prog.row_of_handle[h]] = prog.sites[h].clone(); vt[prog.row_of_handle[h]] = prog.vtypes[h]; }
        (sites, vt)
    }

    /// `hmc_chain(rng, model_fn, n_samples, n_warmup, config)` (hmc.rs:566-572) for every chain.
    pub fn hmc_chain<A>(&self, seed: u64, model_fn: impl Fn() -> Model<A>, n_samples: usize, n_warmup: usize, config: HMCConfig) -> FugueResult<ChainDraws> {
        let prog = flatten(&model_fn, seed)?;
        let eng = self.engine(&prog, seed)?;
        let (sites, vtypes) = Self::sorted_sites(&prog);
        let c = self.n_chains as usize;
        let d = unsafe { fg_program_n_f64(prog.raw) } as usize;
        let mut cfg = fg_hmc_config { n_leapfrog: config.n_leapfrog as i32, target_accept: config.target_accept,
                                      init_step_size: config.init_step_size.unwrap_or(f64::NAN), finite_diff_eps: config.finite_diff_eps,
                                      adapt_mass: config.adapt_mass as i32, grad_mode: FG_GRAD_FD_SPARSE };
        let _ = &mut cfg;
        // hmc_chain returns every site of the trace; HMC moves only the f64 sites, the others keep their prior draw (hmc.rs:238-260)
        let bytes = n_samples.max(1) * d.max(1) * c * 8;
        let d_draws = unsafe { fg_device_alloc(eng.0, bytes) } as *mut f64;
        let mut st = fg_hmc_stats::default();
        check(unsafe { fg_hmc_run(eng.0, &cfg, n_samples as i32, n_warmup as i32, d_draws, &mut st) })?;
        let mut draws = vec![0f64; n_samples * d * c];
        check(unsafe { fg_device_download(eng.0, draws.as_mut_ptr() as *mut _, d_draws as *const _, n_samples * d * c * 8) })?;
        check(unsafe { fg_device_free(eng.0, d_draws as *mut _) })?;
        // discrete sites: constant over the run -> the engine's final values
        let s = sites.len();
        let mut last = vec![0i64; s.max(1) * c];
        check(unsafe { fg_engine_get_values(eng.0, last.as_mut_ptr() as *mut _) })?;
        let f64_rows: Vec<usize> = (0..d).map(|k| unsafe { fg_program_f64_site(prog.raw, k as i32) } as usize).collect();
        let mut cells = vec![0i64; n_samples * s * c];
        for t in 0..n_samples {
            for row in 0..s { for ch in 0..c { cells[(t * s + row) * c + ch] = last[row * c + ch]; } }
            for (k, &row) in f64_rows.iter().enumerate() { for ch in 0..c { cells[(t * s + row) * c + ch] = draws[(t * d + k) * c + ch].to_bits() as i64; } }
        }
        Ok(ChainDraws { sites, vtypes, n_samples, n_chains: c, cells, accept_rate: st.accept_rate })
    }

    /// `adaptive_mcmc_chain_with_overrides(rng, model_fn, n_samples, n_warmup, overrides)` (mh.rs:921-1014) for every chain.
    pub fn adaptive_mcmc_chain<A>(&self, seed: u64, model_fn: impl Fn() -> Model<A>, n_samples: usize, n_warmup: usize,
                                  overrides: &HashMap<Address, SiteProposal>) -> FugueResult<ChainDraws> {
        let prog = flatten(&model_fn, seed)?;
        let eng = self.engine(&prog, seed)?;
        let (sites, vtypes) = Self::sorted_sites(&prog);
        let (s, c) = (sites.len(), self.n_chains as usize);
        let ov: Vec<fg_site_proposal> = sites.iter().map(|a| match overrides.get(a) {
            None => fg_site_proposal { kind: 0, lower: 0.0, upper: 0.0 },
            Some(SiteProposal::Gaussian) => fg_site_proposal { kind: 1, lower: 0.0, upper: 0.0 },
            Some(SiteProposal::LogSpace) => fg_site_proposal { kind: 2, lower: 0.0, upper: 0.0 },
            Some(SiteProposal::Reflect { lower, upper }) => fg_site_proposal { kind: 3, lower: *lower, upper: *upper },
            Some(SiteProposal::PriorResample) => fg_site_proposal { kind: 4, lower: 0.0, upper: 0.0 },
        }).collect();
        let rec: Vec<i32> = (0..s as i32).collect();
        let bytes = n_samples.max(1) * s.max(1) * c * 8;
        let d_draws = unsafe { fg_device_alloc(eng.0, bytes) };
        let mut st = fg_mh_stats::default();
        check(unsafe { fg_mh_run(eng.0, n_samples as i32, n_warmup as i32, if overrides.is_empty() { std::ptr::null() } else { ov.as_ptr() },
                                 rec.as_ptr(), s as i32, d_draws, &mut st) })?;
        let mut cells = vec![0i64; n_samples * s * c];
        check(unsafe { fg_device_download(eng.0, cells.as_mut_ptr() as *mut _, d_draws, n_samples * s * c * 8) })?;
        check(unsafe { fg_device_free(eng.0, d_draws) })?;
        Ok(ChainDraws { sites, vtypes, n_samples, n_chains: c, cells, accept_rate: st.accept_rate })
    }

    /// `adaptive_smc(rng, num_particles, model_fn, config)` (smc.rs:455-460): particles = the engine's chains.
    pub fn adaptive_smc<A>(&self, seed: u64, num_particles: usize, model_fn: impl Fn() -> Model<A>, config: SMCConfig) -> FugueResult<SMCResult> {
        let prog = flatten(&model_fn, seed)?;
        let me = GpuBackend { device: self.device, n_chains: num_particles as i64, chain_offset: 0 };
        let eng = me.engine(&prog, seed)?;
        let (sites, vtypes) = Self::sorted_sites(&prog);
        let (s, n) = (sites.len(), num_particles);
        let cfg = fg_smc_config { resampling_method: match config.resampling_method { ResamplingMethod::Multinomial => 0, ResamplingMethod::Systematic => 1, ResamplingMethod::Stratified => 2 },
                                  ess_threshold: config.ess_threshold, rejuvenation_steps: config.rejuvenation_steps as i32,
                                  sequential_adaptation: 0 /* 1 = the reference's particle-by-particle adaptation, sequential by construction */ };
        let (mut lw, mut w) = (vec![0f64; n], vec![0f64; n]);
        let mut res = fg_smc_result::default();
        check(unsafe { fg_smc_run(eng.0, &cfg, lw.as_mut_ptr(), w.as_mut_ptr(), &mut res, std::ptr::null_mut(), 0) })?;
        let mut cells = vec![0i64; s.max(1) * n];
        check(unsafe { fg_engine_get_values(eng.0, cells.as_mut_ptr() as *mut _) })?;
        let draws = ChainDraws { sites, vtypes, n_samples: 1, n_chains: n, cells, accept_rate: 0.0 };
        let particles = (0..n).map(|i| {
            let (_a, trace) = draws.state(&model_fn, 0, i);
            Particle { trace, weight: w[i], log_weight: lw[i] }
        }).collect();
        Ok(SMCResult { particles, log_evidence: res.log_evidence })
    }
}
