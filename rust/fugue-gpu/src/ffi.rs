//! `extern "C"` view of `include/fugue_amd.h` (ABI version 1).  Every `c_int` result is 0, a reference `ErrorCode` value
//! (`src/error.rs:40-59`) or a negative `FG_E_*`; `fg_last_error()` carries the message.
//! UNVERIFIED SOURCE -- never compiled.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct fg_program { _p: [u8; 0] }
#[repr(C)] pub struct fg_engine { _p: [u8; 0] }

pub const FG_E_NO_DEVICE: c_int = -1;
pub const FG_E_HIP: c_int = -2;
pub const FG_E_BAD_ARG: c_int = -3;
pub const FG_E_STATE: c_int = -5;
pub const FG_E_UNSUPPORTED: c_int = -6;
pub const FG_E_LIMIT: c_int = -7;

// expression tokens (postfix), include/fugue_amd.h FG_T_*
pub const FG_T_CONST: i32 = 0;
pub const FG_T_SITE: i32 = 1;
pub const FG_T_ADD: i32 = 12;
pub const FG_T_MUL: i32 = 14;
pub const FG_T_SELECT: i32 = 20;

// value types (ChoiceValue tags) and gradient modes
pub const FG_F64: c_int = 0;
pub const FG_BOOL: c_int = 1;
pub const FG_U64: c_int = 2;
pub const FG_USIZE: c_int = 3;
pub const FG_I64: c_int = 4;
pub const FG_GRAD_FD_DENSE: i32 = 0;
pub const FG_GRAD_FD_SPARSE: i32 = 1;

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct fg_tok { pub op: i32, pub a: i32, pub b: i32, pub reserved: i32, pub imm: f64 }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct fg_hmc_config {
    pub n_leapfrog: i32, pub target_accept: f64, pub init_step_size: f64 /* NaN = None */,
    pub finite_diff_eps: f64, pub adapt_mass: i32, pub grad_mode: i32,
}
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct fg_hmc_stats { pub accept_rate: f64, pub mean_step_size: f64, pub n_divergent: i64, pub n_transitions: i64 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct fg_site_proposal { pub kind: i32, pub lower: f64, pub upper: f64 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct fg_mh_stats { pub accept_rate: f64, pub n_steps: i64 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct fg_smc_config { pub resampling_method: i32, pub ess_threshold: f64, pub rejuvenation_steps: i32, pub sequential_adaptation: i32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct fg_smc_result { pub log_evidence: f64, pub n_steps: i32, pub n_model_runs: i64 }

pub type fg_acov_fn = Option<unsafe extern "C" fn(user: *mut c_void, lag0: c_int, n_lags: c_int, h_sums: *mut f64) -> c_int>;
pub type fg_reduce_fn = Option<unsafe extern "C" fn(user: *mut c_void, stage: c_int, h_in: *const f64, h_out: *mut f64) -> c_int>;
pub const FG_DIAG_REDUCE: c_int = 0;
pub const FG_DIAG_GATHER: c_int = 1;

#[link(name = "fugue_amd")]
extern "C" {
    pub fn fg_last_error() -> *const c_char;
    pub fn fg_abi_version() -> c_int;
    // ---- site programs (replaces Model<A> + Handler + run: model.rs:20-131, handler.rs:29-209)
    pub fn fg_program_new() -> *mut fg_program;
    pub fn fg_program_free(p: *mut fg_program);
    pub fn fg_program_data(p: *mut fg_program, name: *const c_char, v: *const f64, n: i64) -> c_int;
    pub fn fg_program_sample(p: *mut fg_program, addr: *const c_char, dist: c_int, toks: *const fg_tok, param_len: *const i32, n_params: c_int) -> c_int;
    pub fn fg_program_sample_discrete_uniform(p: *mut fg_program, addr: *const c_char, lo: i64, hi: i64) -> c_int;
    pub fn fg_program_observe(p: *mut fg_program, addr: *const c_char, dist: c_int, toks: *const fg_tok, param_len: *const i32, n_params: c_int,
                              value: *const fg_tok, n_value: c_int) -> c_int;
    pub fn fg_program_factor(p: *mut fg_program, toks: *const fg_tok, n: c_int) -> c_int;
    pub fn fg_program_finalize(p: *mut fg_program) -> c_int;
    pub fn fg_program_n_sites(p: *const fg_program) -> c_int;
    pub fn fg_program_n_f64(p: *const fg_program) -> c_int;
    pub fn fg_program_n_observe(p: *const fg_program) -> c_int;
    pub fn fg_program_site_name(p: *const fg_program, j: c_int, buf: *mut c_char, len: c_int) -> c_int;
    pub fn fg_program_site_vtype(p: *const fg_program, j: c_int) -> c_int;
    pub fn fg_program_site_of_handle(p: *const fg_program, h: c_int) -> c_int;
    pub fn fg_program_f64_site(p: *const fg_program, k: c_int) -> c_int;
    pub fn fg_program_stream_records(p: *const fg_program, which: c_int) -> c_int;
    pub fn fg_dsl_compile(source_utf8: *const c_char, data_json_utf8: *const c_char) -> *mut fg_program;
    // ---- engine
    pub fn fg_engine_new(p: *const fg_program, n_chains: i64, seed: u64, chain_offset: u32, device: c_int) -> *mut fg_engine;
    pub fn fg_engine_free(e: *mut fg_engine);
    pub fn fg_engine_synchronize(e: *mut fg_engine) -> c_int;
    pub fn fg_engine_set_values(e: *mut fg_engine, h_cells: *const c_void) -> c_int;
    pub fn fg_engine_get_values(e: *mut fg_engine, h_cells: *mut c_void) -> c_int;
    pub fn fg_prior_init(e: *mut fg_engine, iteration: u32, h_acc: *mut f64) -> c_int;
    pub fn fg_log_joint(e: *mut fg_engine, h_acc: *mut f64, h_logp: *mut f64) -> c_int;
    // ---- HMC (hmc.rs:566-583, 643-920)
    pub fn fg_hmc_config_default(cfg: *mut fg_hmc_config);
    pub fn fg_hmc_init(e: *mut fg_engine, cfg: *const fg_hmc_config, n_warmup: c_int) -> c_int;
    pub fn fg_hmc_step(e: *mut fg_engine, n: c_int, d_draws: *mut f64) -> c_int;
    pub fn fg_hmc_step_info(e: *mut fg_engine, n: c_int, d_positions: *mut f64, d_info: *mut f64) -> c_int;
    pub fn fg_hmc_step_recorded(e: *mut fg_engine, n_recorded: c_int, h_chain_ids: *const i64, h_traj: *mut f64, h_ham: *mut f64,
                                h_n_points: *mut i32, d_info: *mut f64) -> c_int;
    pub fn fg_hmc_run(e: *mut fg_engine, cfg: *const fg_hmc_config, n_samples: c_int, n_warmup: c_int, d_draws: *mut f64, stats: *mut fg_hmc_stats) -> c_int;
    pub fn fg_hmc_get_stats(e: *mut fg_engine, stats: *mut fg_hmc_stats) -> c_int;
    pub fn fg_hmc_get_step_sizes(e: *mut fg_engine, h_eps: *mut f64) -> c_int;
    pub fn fg_hmc_set_step_size(e: *mut fg_engine, eps: f64) -> c_int;
    pub fn fg_hmc_set_n_leapfrog(e: *mut fg_engine, n_leapfrog: c_int) -> c_int;
    pub fn fg_hmc_is_warming_up(e: *const fg_engine) -> c_int;
    pub fn fg_hmc_iterations(e: *const fg_engine) -> i64;
    pub fn fg_hmc_last_kernel(e: *const fg_engine) -> *const c_char;
    pub fn fg_mh_last_kernel(e: *const fg_engine) -> *const c_char;
    // ---- single-site MH (mh.rs:921-1014)
    pub fn fg_mh_init(e: *mut fg_engine, n_warmup: c_int, overrides: *const fg_site_proposal) -> c_int;
    pub fn fg_mh_step(e: *mut fg_engine, n: c_int, rec_sites: *const i32, n_rec: c_int, d_draws: *mut c_void) -> c_int;
    pub fn fg_mh_set_recording(e: *mut fg_engine, during_adaptation: c_int) -> c_int;
    pub fn fg_mh_run(e: *mut fg_engine, n_samples: c_int, n_warmup: c_int, overrides: *const fg_site_proposal, rec_sites: *const i32, n_rec: c_int,
                     d_draws: *mut c_void, stats: *mut fg_mh_stats) -> c_int;
    // ---- SMC (smc.rs:230-349, 455-790)
    pub fn fg_smc_config_default(cfg: *mut fg_smc_config);
    pub fn fg_smc_run(e: *mut fg_engine, cfg: *const fg_smc_config, h_log_w: *mut f64, h_weights: *mut f64, res: *mut fg_smc_result,
                      h_betas: *mut f64, max_betas: c_int) -> c_int;
    pub fn fg_smc_prior_particles(e: *mut fg_engine, iteration: u32) -> c_int;
    pub fn fg_smc_normalize(e: *mut fg_engine) -> c_int;
    pub fn fg_smc_ess(e: *mut fg_engine, out: *mut f64) -> c_int;
    pub fn fg_smc_resample(e: *mut fg_engine, method: c_int, step: u32, h_indices: *mut i64) -> c_int;
    pub fn fg_smc_rejuvenate(e: *mut fg_engine, beta: f64, steps: c_int, first_move_id: u32, h_accept_rate: *mut f64) -> c_int;
    pub fn fg_smc_get_weights(e: *mut fg_engine, h_log_w: *mut f64, h_w: *mut f64) -> c_int;
    pub fn fg_smc_set_log_weights(e: *mut fg_engine, h_log_w: *const f64) -> c_int;
    // ---- cross-chain diagnostics (diagnostics.rs:218-304, mcmc_utils.rs:214-421)
    pub fn fg_diag_rhat_ess(e: *mut fg_engine, d_draws: *const f64, n: c_int, d: c_int, rccl_comm: *mut c_void, h_rhat: *mut f64, h_ess: *mut f64,
                            h_mean: *mut f64, h_std: *mut f64, total_chains: *mut i64) -> c_int;
    pub fn fg_diag_geweke(e: *mut fg_engine, d_draws: *const f64, n: c_int, d: c_int, d_z: *mut f64) -> c_int;
    pub fn fg_diag_combine(h_moments: *const f64, m: i64, n: c_int, d: c_int, acov: fg_acov_fn, user: *mut c_void, h_rhat: *mut f64, h_ess: *mut f64,
                           h_mean: *mut f64, h_std: *mut f64) -> c_int;
    pub fn fg_diag_combine_reduced(m: i64, n: c_int, d: c_int, reduce: fg_reduce_fn, acov: fg_acov_fn, user: *mut c_void, h_rhat: *mut f64, h_ess: *mut f64,
                                   h_mean: *mut f64, h_std: *mut f64) -> c_int;
    pub fn fg_diag_quantiles(e: *mut fg_engine, d_draws: *const f64, n: c_int, d: c_int, rccl_comm: *mut c_void, h_probs: *const f64, n_probs: c_int,
                             h_out: *mut f64) -> c_int;
    pub fn fg_diag_set_exchange(e: *mut fg_engine, mode: c_int) -> c_int;
    pub fn fg_diag_exchange_bytes(e: *const fg_engine) -> i64;
    pub fn fg_comm_unique_id(out_128_bytes: *mut c_void) -> c_int;
    pub fn fg_comm_init(e: *mut fg_engine, world: c_int, rank: c_int, id_128_bytes: *const c_void, out_comm: *mut *mut c_void) -> c_int;
    pub fn fg_comm_destroy(comm: *mut c_void) -> c_int;
    // ---- checkpoint / resume (HmcSession fields hmc.rs:643-661)
    pub fn fg_state_size(e: *mut fg_engine) -> i64;
    pub fn fg_state_export(e: *mut fg_engine, h_buf: *mut c_void, capacity: usize) -> c_int;
    pub fn fg_state_import(e: *mut fg_engine, h_buf: *const c_void, size: usize) -> c_int;
    // ---- raw device memory
    pub fn fg_device_alloc(e: *mut fg_engine, bytes: usize) -> *mut c_void;
    pub fn fg_device_free(e: *mut fg_engine, p: *mut c_void) -> c_int;
    pub fn fg_device_download(e: *mut fg_engine, h: *mut c_void, d: *const c_void, bytes: usize) -> c_int;
    pub fn fg_device_upload(e: *mut fg_engine, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    // ---- introspection of programs and engines
    pub fn fg_program_n_instructions(p: *const fg_program) -> c_int;
    pub fn fg_program_n_slots(p: *const fg_program) -> c_int;
    pub fn fg_program_dep_count(p: *const fg_program, k: c_int) -> c_int;
    pub fn fg_dsl_warning_count(p: *const fg_program) -> c_int;
    pub fn fg_dsl_warning(p: *const fg_program, i: c_int) -> *const c_char;
    pub fn fg_engine_stream(e: *mut fg_engine) -> *mut c_void;                       // hipStream_t
    pub fn fg_engine_set_stream(e: *mut fg_engine, hip_stream: *mut c_void) -> c_int;
    pub fn fg_engine_n_chains(e: *const fg_engine) -> i64;
    pub fn fg_engine_values_device(e: *mut fg_engine) -> *mut c_void;                // d_cells [S][C]
    pub fn fg_log_joint_stream(e: *mut fg_engine, h_acc: *mut f64, h_rec_lp: *mut f64) -> c_int;
    // ---- pieces of the samplers with their randomness injected (what the reference's unit tests drive: hmc.rs:304-329, 419-535)
    pub fn fg_hmc_get_log_joint(e: *mut fg_engine, h_lj: *mut f64) -> c_int;
    pub fn fg_hmc_get_mass(e: *mut fg_engine, h_m_inv: *mut f64) -> c_int;
    pub fn fg_hmc_grad(e: *mut fg_engine, h: f64, grad_mode: c_int, h_grad: *mut f64, h_ok: *mut i32) -> c_int;
    pub fn fg_hmc_transition_injected(e: *mut fg_engine, cfg: *const fg_hmc_config, eps: f64, h_p0: *const f64, h_u: *const f64,
                                      h_accepted: *mut i32, h_alpha: *mut f64, h_divergent: *mut i32, h_lj: *mut f64) -> c_int;
    pub fn fg_hmc_find_eps_injected(e: *mut fg_engine, cfg: *const fg_hmc_config, h_p0: *const f64, h_eps: *mut f64) -> c_int;
    pub fn fg_mh_get_stats(e: *mut fg_engine, h_stats: *mut fg_mh_stats) -> c_int;
    pub fn fg_mh_get_scales(e: *mut fg_engine, h_scales: *mut f64) -> c_int;
    pub fn fg_mh_get_log_weight(e: *mut fg_engine, h_lw: *mut f64) -> c_int;
    // ---- population primitives on host arrays (numerical.rs:15-38, smc.rs:255-314, 588-622)
    pub fn fg_device_log_sum_exp(device_ordinal: c_int, h_x: *const f64, n: i64, out: *mut f64) -> c_int;
    pub fn fg_device_next_beta(device_ordinal: c_int, beta: f64, h_log_w: *const f64, h_loglik: *const f64, n: i64, target_ess: f64, out_beta: *mut f64) -> c_int;
    pub fn fg_device_resample_indices(device_ordinal: c_int, method: c_int, h_weights: *const f64, n: i64, h_u: *const f64, h_idx: *mut i64) -> c_int;
    pub fn fg_diag_chain_moments(e: *mut fg_engine, d_draws: *const f64, n: c_int, d: c_int, d_moments: *mut f64) -> c_int;
    pub fn fg_diag_autocov_sums(e: *mut fg_engine, d_draws: *const f64, n: c_int, d: c_int, d_moments: *const f64, lag0: c_int, n_lags: c_int, h_sums: *mut f64) -> c_int;
}

/// The message of the last failed call on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = fg_last_error();
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
