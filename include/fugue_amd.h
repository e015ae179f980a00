/*
 * fugue_amd.h -- C ABI of the MI355X-native many-chain / many-particle engine for
 * Fugue's `src/inference` hot path (hmc_chain, adaptive_mcmc_chain, adaptive_smc).
 *
 * The reference (alexnodeland/fugue, crate fugue-ppl 0.2.0) has no FFI for this path:
 * the boundary today is Rust generics.  Each entry point below names the reference
 * interface it replaces (file:line under /root/reference); INTEGRATION.md shows the
 * `extern "C"` block and the `GpuBackend` shim a maintainer would add on the Rust side.
 *
 * Conventions
 *   - plain C types only; handles are opaque; every `int` return is 0 on success, a
 *     reference `ErrorCode` value (src/error.rs:40-59: 100-106, 301, 302, 500, 600) for
 *     model errors, or a negative FG_E_* for engine / HIP failures.  fg_last_error()
 *     returns a thread-local message.
 *   - `h_` pointers are host memory, `d_` pointers are device (HBM) memory of the
 *     engine's device.  All [a][b] arrays are row-major with the LAST index = chain
 *     (struct-of-arrays: [sites x chains], site-major so a wavefront's 64 lanes read 64
 *     consecutive chains).
 *   - trace cells are 8 bytes: f64 sites hold the double, bool/u64/usize/i64 sites hold
 *     an int64 (bool 0/1) -- the flattened `ChoiceValue` (src/runtime/trace.rs:32-43).
 *   - site order everywhere = lexicographic order of the address strings = the
 *     reference's `BTreeMap<Address, Choice>` order (src/core/address.rs:150-157);
 *     f64 coordinate order (HMC `q`) is that order restricted to f64 sites
 *     (src/inference/hmc.rs:238-248).
 *   - there is NO CPU fallback: every engine call fails with FG_E_NO_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef FUGUE_AMD_H
#define FUGUE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FG_ABI_VERSION 1

/* engine errors (negative); model errors reuse the reference ErrorCode numbers */
enum {
    FG_OK = 0,
    FG_E_NO_DEVICE = -1,      /* no usable HIP device / wrong arch */
    FG_E_HIP = -2,            /* a HIP runtime call failed */
    FG_E_BAD_ARG = -3,
    FG_E_NOT_FINALIZED = -4,
    FG_E_STATE = -5,          /* call order (e.g. hmc_step before hmc_init) */
    FG_E_UNSUPPORTED = -6,
    FG_E_LIMIT = -7           /* model exceeds an engine limit (LDS budget, K > 64, ...) */
};
enum {
    FG_ERR_INVALID_PROBABILITY = 102, FG_ERR_INVALID_COUNT = 106,
    FG_ERR_ADDRESS_CONFLICT = 301, FG_ERR_UNEXPECTED_STRUCTURE = 302,
    FG_ERR_ADDRESS_NOT_FOUND = 500, FG_ERR_TYPE_MISMATCH = 600
};

/* the 17 distributions, in the order of the crate-root re-export (src/lib.rs:18-22) */
enum {
    FG_BERNOULLI = 0, FG_BETA, FG_BINOMIAL, FG_CATEGORICAL, FG_CAUCHY, FG_CHISQUARED,
    FG_DISCRETEUNIFORM, FG_EXPONENTIAL, FG_GAMMA, FG_INVERSEGAMMA, FG_LAPLACE, FG_LOGNORMAL,
    FG_NORMAL, FG_POISSON, FG_STUDENTT, FG_UNIFORM, FG_WEIBULL, FG_N_DISTS
};
/* ChoiceValue tags (src/runtime/trace.rs:32-43) */
enum { FG_F64 = 0, FG_BOOL = 1, FG_U64 = 2, FG_USIZE = 3, FG_I64 = 4 };

/* ------------------------------------------------------------------ site programs
 * Replaces: `Model<A>` + `Handler` + `run` (src/core/model.rs:20-131,
 * src/runtime/handler.rs:29-209).  A model is flattened ONCE into a fixed-structure
 * site program; parameters that depend on earlier sites are postfix expressions over
 * the DSL's operator set (crates/fugue-wasm/src/dsl.rs:92-102,569-582).            */
enum {
    FG_T_CONST = 0,  /* push imm */
    FG_T_SITE,       /* push value of sample site `a` (handle from fg_program_sample), as f64 */
    FG_T_DATA,       /* push data[a][b] */
    FG_T_NEG, FG_T_EXP, FG_T_LN, FG_T_SQRT, FG_T_ABS, FG_T_FLOOR, FG_T_SIN, FG_T_COS, FG_T_TANH,
    FG_T_ADD, FG_T_SUB, FG_T_MUL, FG_T_DIV, FG_T_POW, FG_T_MIN, FG_T_MAX,
    FG_T_CLAMP,      /* x lo hi -> f64::clamp */
    FG_T_SELECT      /* idx opt0 .. opt{a-1} -> opt[idx]  (a = number of options) */
};
typedef struct fg_tok { int32_t op; int32_t a; int32_t b; int32_t reserved; double imm; } fg_tok;

typedef struct fg_program fg_program;

fg_program *fg_program_new(void);
void        fg_program_free(fg_program *p);
/* named data arrays (DSL `{"y":[...]}`, dsl.rs:1066-1106); returns the array id >= 0 */
int fg_program_data(fg_program *p, const char *name, const double *h_values, int64_t n);
/* `sample(addr, dist)` (src/core/model.rs:144-266).  `toks` holds the parameter
 * expressions back to back, `param_len[i]` tokens each (Categorical: one per
 * probability, 1..64).  Returns the site handle >= 0 (program order), or an error. */
int fg_program_sample(fg_program *p, const char *addr_utf8, int dist, const fg_tok *toks,
                      const int32_t *param_len, int n_params);
/* `observe(addr, dist, value)` (model.rs:381-424) */
int fg_program_observe(fg_program *p, const char *addr_utf8, int dist, const fg_tok *toks,
                       const int32_t *param_len, int n_params, const fg_tok *value, int n_value);
/* `sample(addr, DiscreteUniform::new(lo, hi))` with exact i64 bounds (src/core/distribution.rs:1842-1853 takes i64;
 * the token form above carries them as f64, exact only up to 2^53). */
int fg_program_sample_discrete_uniform(fg_program *p, const char *addr_utf8, int64_t lo, int64_t hi);
/* `factor(logw)` (model.rs:426-431) */
int fg_program_factor(fg_program *p, const fg_tok *toks, int n);
/* sorts sites by address, rejects duplicate addresses (AddressConflict = 301, the panic of
 * src/runtime/interpreters.rs:23-33), compiles the device program. */
int fg_program_finalize(fg_program *p);
int fg_program_n_sites(const fg_program *p);      /* S */
int fg_program_n_f64(const fg_program *p);        /* d */
int fg_program_n_observe(const fg_program *p);    /* O */
int fg_program_n_instructions(const fg_program *p);
int fg_program_n_slots(const fg_program *p);      /* S + expression temporaries */
/* coordinate order == reference BTreeMap order; returns bytes needed incl. NUL */
int fg_program_site_name(const fg_program *p, int sorted_idx, char *buf, int buf_len);
int fg_program_site_vtype(const fg_program *p, int sorted_idx);
int fg_program_site_of_handle(const fg_program *p, int handle);
int fg_program_f64_site(const fg_program *p, int k);
/* number of instructions re-evaluated when f64 coordinate k is perturbed (sparse FD) */
int fg_program_dep_count(const fg_program *p, int k);
/* record streams the compiler could build for the stream kernels (0 = the interpreter kernels are used):
 * which = 0: records of the fused finite-difference gradient stream; 1: records of the score stream;
 * 2: record kinds present (0 fast Normals only, 1 + linear predictors, 2 + general distribution records);
 * 3: records of the register-resident trajectory kernel (> 0 only for independent-sites programs: every force term reads
 *    one coordinate and constants) */
int fg_program_stream_records(const fg_program *p, int which);

const char *fg_last_error(void);
int         fg_abi_version(void);

/* Model-language front-end: the `prob!` subset of the reference's playground
 * (crates/fugue-wasm/src/dsl.rs:10-35 grammar, :1062-1120 CompiledModel::compile).  `data_json` is a JSON
 * object of number/boolean arrays, a bare array (bound to `data`), "null" or NULL.  Returns a FINALIZED
 * program, or NULL with the reference-worded message ("line N: expected .., found ..") in
 * fg_last_error().  Warnings (out-of-bounds data index -> NaN, dsl.rs:715-722) are kept on the program. */
fg_program *fg_dsl_compile(const char *source_utf8, const char *data_json_utf8);
int         fg_dsl_warning_count(const fg_program *p);
const char *fg_dsl_warning(const fg_program *p, int i);

/* ------------------------------------------------------------------ engine
 * One engine = one batch of `n_chains` independent chains (or particles) of one program
 * on one GPU, with its own HIP stream.  Chain c uses the counter-based RNG stream
 * (seed, chain_offset + c): results do not depend on how chains are sharded over GPUs. */
typedef struct fg_engine fg_engine;

fg_engine *fg_engine_new(const fg_program *p, int64_t n_chains, uint64_t seed,
                         uint32_t chain_offset, int device_ordinal);
void  fg_engine_free(fg_engine *e);
int   fg_engine_synchronize(fg_engine *e);
void *fg_engine_stream(fg_engine *e);                       /* hipStream_t */
/* run on a caller-owned hipStream_t (e.g. PyTorch's current stream) instead of the engine's own */
int   fg_engine_set_stream(fg_engine *e, void *hip_stream);
int64_t fg_engine_n_chains(const fg_engine *e);
/* current trace values, cells [S][C]; set_values re-scores the cached log-joint of a live HMC / MH session at the new values */
int fg_engine_set_values(fg_engine *e, const void *h_cells);
int fg_engine_get_values(fg_engine *e, void *h_cells);
void *fg_engine_values_device(fg_engine *e);                 /* d_cells [S][C] */

/* `run(PriorHandler, model)` per chain (src/runtime/interpreters.rs:88-104): draws every
 * site from its prior, scores it; h_acc (optional) gets [3][C] = log_prior,
 * log_likelihood, log_factors.  `iteration` selects the RNG sub-stream. */
int fg_prior_init(fg_engine *e, uint32_t iteration, double *h_acc);
/* `run(ScoreGivenTrace, model)` per chain (interpreters.rs:138-163) on the engine's current
 * values; h_acc [3][C]; h_logp (optional) [S][C] fresh per-site log-densities. */
int fg_log_joint(fg_engine *e, double *h_acc, double *h_logp);
/* The same scoring run evaluated over the SCORE STREAM (one 64-byte record per statement in program order) -- the
 * evaluator of the HMC endpoint score_full (src/inference/hmc.rs:283-299), of single_site_mh_step's model run
 * (src/inference/mh.rs:698-744) and of SMC rejuvenation (src/inference/smc.rs:662-675) for programs whose statements
 * all have a record form.  h_acc [3][C]; h_rec_lp (optional) [n_records][C], n_records =
 * fg_program_stream_records(p, 1): the log-density of every statement.  FG_E_UNSUPPORTED when the program has no
 * score stream. */
int fg_log_joint_stream(fg_engine *e, double *h_acc, double *h_rec_lp);

/* ------------------------------------------------------------------ HMC
 * Replaces hmc_chain / HmcSession (src/inference/hmc.rs:566-583, 643-920). */
enum { FG_GRAD_FD_DENSE = 0,   /* hmc.rs:304-329 verbatim: 2d full model runs per gradient */
       FG_GRAD_FD_SPARSE = 1,  /* same central difference, re-evaluating only the terms that
                                  depend on the perturbed coordinate */
       FG_GRAD_ANALYTIC = 2    /* the derivative itself.  Programs whose force terms are all Normals with constant
                                  sigma whose mean is a site, a constant or a linear predictor: the closed form
                                  d/dq_i sum of -(x - mu)^2 / (2 sigma^2).  Every other program: the forward-mode
                                  derivative of each coordinate's sub-program in the unit compiled at run time
                                  (all 17 log-densities in value and parameters).  NOT the reference's arithmetic
                                  (it has no analytic mode): agrees with the finite difference to its O(h^2) +
                                  rounding error; the step-size search still uses FG_GRAD_FD_SPARSE.  fg_hmc_init
                                  returns FG_E_UNSUPPORTED where neither applies (no run-time compiler, FG_JIT=0). */ };
typedef struct fg_hmc_config {      /* HMCConfig, hmc.rs:106-135 (same defaults) */
    int32_t n_leapfrog;             /* 16 */
    double  target_accept;          /* 0.8 */
    double  init_step_size;         /* NaN = None: Hoffman-Gelman Alg. 4 (hmc.rs:479-535) */
    double  finite_diff_eps;        /* 1e-5 */
    int32_t adapt_mass;             /* 0 */
    int32_t grad_mode;              /* FG_GRAD_* (engine extension): default FG_GRAD_FD_SPARSE; FG_GRAD_FD_DENSE = the reference verbatim */
} fg_hmc_config;
typedef struct fg_hmc_stats {
    double  accept_rate;            /* mean acceptance probability over chains x transitions */
    double  mean_step_size;         /* mean over chains of the step size in use */
    int64_t n_divergent;
    int64_t n_transitions;          /* chains x transitions executed so far */
} fg_hmc_stats;
void fg_hmc_config_default(fg_hmc_config *cfg);
/* HmcSession::new (hmc.rs:667-729): prior draw, positions, initial step size */
int fg_hmc_init(fg_engine *e, const fg_hmc_config *cfg, int n_warmup);
/* HmcSession::step x n (hmc.rs:819-919).  Post-warmup positions are appended to
 * d_draws [n][d][C] when non-NULL (rows of warmup transitions are left untouched). */
int fg_hmc_step(fg_engine *e, int n_transitions, double *d_draws);
/* HmcSession::step x n returning every transition's HmcStepInfo (hmc.rs:587-602,803-805):
 * d_positions [n][d][C] = position after each transition (warmup included), d_info [n][4][C] =
 * accepted (0/1), divergent (0/1), accept_prob, step_size.  Either may be NULL. */
int fg_hmc_step_info(fg_engine *e, int n_transitions, double *d_positions, double *d_info);
/* hmc_chain (hmc.rs:566-583): init + n_warmup + n_samples; d_draws [n_samples][d][C] */
int fg_hmc_run(fg_engine *e, const fg_hmc_config *cfg, int n_samples, int n_warmup,
               double *d_draws, fg_hmc_stats *h_stats);
int fg_hmc_get_stats(fg_engine *e, fg_hmc_stats *h_stats);
int fg_hmc_get_step_sizes(fg_engine *e, double *h_eps /*[C]*/);
int fg_hmc_get_log_joint(fg_engine *e, double *h_lj /*[C]*/);
/* current diagonal inverse mass (HmcSession::m_inv, hmc.rs:652): h_m_inv [d][C] (all 1 without adapt_mass) */
int fg_hmc_get_mass(fg_engine *e, double *h_m_inv);
/* set_step_size (hmc.rs:741-747) for every chain */
int fg_hmc_set_step_size(fg_engine *e, double eps);
/* HmcSession::set_n_leapfrog / is_warming_up / iterations (hmc.rs:751-753, 780-782, 785-787) */
int     fg_hmc_set_n_leapfrog(fg_engine *e, int n_leapfrog);
int     fg_hmc_is_warming_up(const fg_engine *e);
int64_t fg_hmc_iterations(const fg_engine *e);
/* Which kernel the engine's last fg_hmc_step / fg_hmc_run launch ran, with its waves per 64-chain tile, e.g.
 * "k_hmc_sep_steps W=8" (independent sites: whole trajectories in registers), "k_hmc_lin_steps W=8" (dense regressions:
 * observation-major gradient), "k_hmc_stream_steps W=4" (gradient / score streams), "k_hmc_steps W=1" (interpreter);
 * "" before the first launch.  The pointer is valid until the engine's next launch. */
const char *fg_hmc_last_kernel(const fg_engine *e);
/* The same for fg_mh_step / fg_mh_run: "k_mh_mw_steps W=4" (score-stream programs of plain Normal records), "k_mh_mw_jit_steps W=4 ..."
 * (the same pipelined kernel around statements compiled at run time: score-stream programs with general records, programs without a
 * record stream), "k_mh_jit_steps W=4 (compiled at run time)" (programs whose term rows do not fit LDS), "k_mh_interp_mw_steps W=2",
 * "k_mh_steps W=1".  Compilation at run time uses hiprtc or hipcc when present (FG_JIT=0 keeps every program on the hand-written and
 * interpreter kernels). */
const char *fg_mh_last_kernel(const fg_engine *e);
/* HmcSession::step_recorded (hmc.rs:811-817) for every chain: ONE transition, and for the n_recorded chains h_chain_ids
 * the leapfrog trajectory with the Hamiltonian at each integration point (LeapfrogPoint, hmc.rs:338-343):
 * h_traj [n_recorded][L+1][d] positions, h_ham [n_recorded][L+1], h_n_points [n_recorded] (L + 1, fewer when the
 * trajectory left the support: recording stops at the last finite point).  d_info (optional) [4][C] as fg_hmc_step_info.
 * Recording consumes no randomness: the chains advance exactly as under fg_hmc_step (hmc.rs:1058-1087). */
int fg_hmc_step_recorded(fg_engine *e, int n_recorded, const int64_t *h_chain_ids, double *h_traj, double *h_ham,
                         int32_t *h_n_points, double *d_info);
/* test / diagnostics hooks under injected randomness -------------------------------- */
/* grad_log_joint (hmc.rs:304-329) at the engine's current values; h_grad [d][C], h_ok [C] */
int fg_hmc_grad(fg_engine *e, double h, int grad_mode, double *h_grad, int32_t *h_ok);
/* hmc_transition (hmc.rs:419-473) with momentum h_p0 [d][C] and uniform h_u [C] injected
 * and one step size for all chains; outputs are host arrays [C] (any may be NULL).
 * The engine's values / log-joint advance exactly as a real transition would. */
int fg_hmc_transition_injected(fg_engine *e, const fg_hmc_config *cfg, double eps,
                               const double *h_p0, const double *h_u, int32_t *h_accepted,
                               double *h_alpha, int32_t *h_divergent, double *h_lj);
/* find_reasonable_epsilon (hmc.rs:479-535) with momentum injected; h_eps [C] */
int fg_hmc_find_eps_injected(fg_engine *e, const fg_hmc_config *cfg, const double *h_p0,
                             double *h_eps);

/* ------------------------------------------------------------------ adaptive single-site MH
 * Replaces adaptive_mcmc_chain[_with_overrides] (src/inference/mh.rs:921-1014). */
enum { FG_PROP_AUTO = 0,      /* support-based choice: Gaussian, or LogSpace for positive support (mh.rs:339-358) */
       FG_PROP_GAUSSIAN = 1, FG_PROP_LOGSPACE = 2, FG_PROP_REFLECT = 3, FG_PROP_PRIOR_RESAMPLE = 4 };
typedef struct fg_site_proposal {   /* SiteProposal, mh.rs:145-161 */
    int32_t kind; double lower, upper;
} fg_site_proposal;
typedef struct fg_mh_stats { double accept_rate; int64_t n_steps; } fg_mh_stats;
/* prior init + DiminishingAdaptation::new(0.44, 0.7) per chain (mh.rs:945-965).
 * overrides: [S] in site order (HashMap<Address, SiteProposal> flattened) or NULL. */
int fg_mh_init(fg_engine *e, int n_warmup, const fg_site_proposal *h_overrides);
/* n x single_site_mh_step (mh.rs:698-744); adapts while iteration < n_warmup.  After every
 * sampling-phase step the current values of h_rec_sites[0..n_rec) are appended to
 * d_draws [n_sampling_steps][n_rec][C] (8-byte cells).  n_rec = 0 records nothing. */
int fg_mh_step(fg_engine *e, int n_steps, const int32_t *h_rec_sites, int n_rec, void *d_draws);
/* Incremental drivers (the step(n) / values_since protocol of crates/fugue-wasm/src/mh.rs:92-168, whose chains adapt for ever
 * and keep every state): with during_adaptation != 0 fg_mh_step records after EVERY step, d_draws [n_steps][n_rec][C]. */
int fg_mh_set_recording(fg_engine *e, int during_adaptation);
int fg_mh_run(fg_engine *e, int n_samples, int n_warmup, const fg_site_proposal *h_overrides,
              const int32_t *h_rec_sites, int n_rec, void *d_draws, fg_mh_stats *h_stats);
int fg_mh_get_stats(fg_engine *e, fg_mh_stats *h_stats);
int fg_mh_get_scales(fg_engine *e, double *h_scales /*[S][C]*/);     /* DiminishingAdaptation::scales */
int fg_mh_get_log_weight(fg_engine *e, double *h_lw /*[C]*/);        /* total_log_weight of the current trace */

/* ------------------------------------------------------------------ likelihood-tempered SMC
 * Replaces adaptive_smc (src/inference/smc.rs:455-581).  Particles = the engine's n_chains. */
enum { FG_RESAMPLE_MULTINOMIAL = 0, FG_RESAMPLE_SYSTEMATIC = 1, FG_RESAMPLE_STRATIFIED = 2 };   /* ResamplingMethod, smc.rs:133-140 */
typedef struct fg_smc_config {      /* SMCConfig, smc.rs:172-189 (same defaults) */
    int32_t resampling_method;      /* systematic */
    double  ess_threshold;          /* 0.5 */
    int32_t rejuvenation_steps;     /* 0 */
    int32_t sequential_adaptation;  /* 0: every particle of a rejuvenation sweep uses the scales from the sweep's start and the shared
                                     * DiminishingAdaptation is updated once per sweep from per-site counts (the many-particle form);
                                     * 1: the reference's own order (smc.rs:482,544-553,698-713) -- particle-major, the one adaptation
                                     * updated after EVERY move -- walked by one wave, sequential by construction (a few us per move:
                                     * for checking parity with the reference's semantics, not for speed).  (Sits in what was padding:
                                     * the struct's size and the other fields' offsets are unchanged.) */
} fg_smc_config;
typedef struct fg_smc_result {      /* SMCResult minus the particles (smc.rs:361-366) */
    double  log_evidence;
    int32_t n_steps;                /* tempering steps taken */
    int64_t n_model_runs;
} fg_smc_result;
void fg_smc_config_default(fg_smc_config *cfg);
/* Runs the whole ladder.  On return the engine's values [S][N] hold the final particles;
 * h_log_w / h_weights (optional, [N]) get the normalised log-weights / weights;
 * h_betas (optional) the inverse-temperature ladder. */
int fg_smc_run(fg_engine *e, const fg_smc_config *cfg, double *h_log_w, double *h_weights,
               fg_smc_result *h_result, double *h_betas, int max_betas);
/* The reference's standalone SMC building blocks over the engine's particles (values [S][N] in the engine; log-weights,
 * weights and log-likelihoods in HBM beside them):
 *   fg_smc_prior_particles  smc_prior_particles (smc.rs:764-790): prior draws, log_weight = log_likelihood + log_factors, normalised
 *   fg_smc_normalize        normalize_particles (smc.rs:719-755): weight = exp(log_weight - lse) / sum; uniform if all -inf
 *   fg_smc_ess              effective_sample_size (smc.rs:230-233): 1 / sum w^2
 *   fg_smc_resample         resample_particles (smc.rs:326-349): ancestors (h_indices, optional), clones, weight = 1/N
 *   fg_smc_rejuvenate       rejuvenate_particles (smc.rs:698-713): pi_beta-invariant MH moves, weights untouched
 *   fg_smc_get_weights / fg_smc_set_log_weights: the population's log-weights and weights (a caller's own reweighting) */
int fg_smc_prior_particles(fg_engine *e, uint32_t iteration);
int fg_smc_normalize(fg_engine *e);
int fg_smc_ess(fg_engine *e, double *out_ess);
int fg_smc_resample(fg_engine *e, int method, uint32_t step, int64_t *h_indices /*[N] or NULL*/);
int fg_smc_rejuvenate(fg_engine *e, double beta, int steps, uint32_t first_move_id, double *h_accept_rate /*or NULL*/);
int fg_smc_get_weights(fg_engine *e, double *h_log_w, double *h_weights);
int fg_smc_set_log_weights(fg_engine *e, const double *h_log_w);
/* population-wide primitives on device `device_ordinal`, usable without a program:
 * log_sum_exp (src/core/numerical.rs:15-38), next_beta (smc.rs:588-622) and
 * {multinomial,systematic,stratified}_indices (smc.rs:255-314) with the uniforms injected
 * (h_u: 1 value for systematic, n otherwise). */
int fg_device_log_sum_exp(int device_ordinal, const double *h_x, int64_t n, double *out);
int fg_device_next_beta(int device_ordinal, double beta, const double *h_log_w, const double *h_loglik,
                        int64_t n, double target_ess, double *out_beta);
int fg_device_resample_indices(int device_ordinal, int method, const double *h_weights, int64_t n,
                               const double *h_u, int64_t *h_idx);

/* ------------------------------------------------------------------ cross-chain diagnostics
 * Per-chain statistics behind r_hat_f64 / effective_sample_size_multichain
 * (src/inference/diagnostics.rs:218-304, src/inference/mcmc_utils.rs:214-339).  Draws stay on the
 * GPU; only these small summaries are exchanged between GPUs (RCCL) and combined on the host. */
/* d_draws [n][d][C] -> d_moments [d][6][C] = mean, sum of squared deviations of the full chain,
 * of its first half and of its second half (half = n/2, middle draw dropped when n is odd). */
int fg_diag_chain_moments(fg_engine *e, const double *d_draws, int n, int d, double *d_moments);
/* h_sums [d][n_lags] = sum over this engine's chains of the biased lag-t autocovariances
 * (mcmc_utils.rs:231-244) for t in [lag0, lag0 + n_lags). */
int fg_diag_autocov_sums(fg_engine *e, const double *d_draws, int n, int d, const double *d_moments,
                         int lag0, int n_lags, double *h_sums);

/* ------------------------------------------------------------------ checkpoint / resume
 * The per-chain sampler state as one flat host blob: the fields of HmcSession (hmc.rs:643-661) for every chain, the MH
 * chain state (current trace, log-weight, DiminishingAdaptation per site, decided proposal kinds, overrides) and the
 * iteration counters that position the counter-based random streams (cf. the wasm samplers' step(n) protocol,
 * crates/fugue-wasm/src/mh.rs:92-168).  run(a); export; [new engine of the same program and chain count] import;
 * run(b) reproduces run(a + b) bit for bit. */
int64_t fg_state_size(fg_engine *e);
int     fg_state_export(fg_engine *e, void *h_buf, size_t capacity);
int     fg_state_import(fg_engine *e, const void *h_buf, size_t size);

/* geweke_diagnostic (src/inference/mcmc_utils.rs:354-421) of every (coordinate, chain): d_draws [n][d][C] -> d_z [d][C] */
int fg_diag_geweke(fg_engine *e, const double *d_draws, int n, int d, double *d_z);
/* r_hat_f64 (split R-hat, diagnostics.rs:218-224,240-304), effective_sample_size_multichain (mcmc_utils.rs:214-339) and the
 * pooled mean / sample std of summarize_f64_parameter (diagnostics.rs:331-352) for every coordinate of d_draws [n][d][C],
 * over the chains of EVERY rank of `rccl_comm` (an ncclComm_t; NULL = this engine's chains only).  With a communicator the
 * chain sums and the pooled lag sums are all-reduced over RCCL / xGMI inside the call (fg_diag_set_exchange); all ranks call
 * it with equal n, d and chain count and receive the same numbers.  h_* are [d]; any may be NULL. */
int fg_diag_rhat_ess(fg_engine *e, const double *d_draws, int n, int d, void *rccl_comm, double *h_rhat, double *h_ess,
                     double *h_mean, double *h_std, int64_t *out_total_chains);
/* How the ranks of `rccl_comm` exchange chain statistics inside fg_diag_rhat_ess.  FG_DIAG_REDUCE (default): chains enter split
 * R-hat (diagnostics.rs:262-304), the pooled mean / std and the multi-chain ESS (mcmc_utils.rs:253-339) only through sums over
 * chains, so every rank reduces its own chains on the device and the ranks all-reduce 6 d + 2 d doubles plus 32 d per chunk of
 * lags -- nothing proportional to the chain count leaves a GPU.  FG_DIAG_GATHER: all-gather of every chain's moments
 * ([d][6][C] per rank) and the combination in global chain order on every rank, as a single process would sum them.  The two
 * agree to rounding of the sums over chains (~1e-15 relative).  fg_diag_exchange_bytes: bytes this rank contributed to
 * collectives during the last fg_diag_rhat_ess (0 without a communicator). */
#define FG_DIAG_REDUCE 0
#define FG_DIAG_GATHER 1
int fg_diag_set_exchange(fg_engine *e, int mode);
int64_t fg_diag_exchange_bytes(const fg_engine *e);
/* The quantiles of summarize_f64_parameter (diagnostics.rs:355-371): for every coordinate of d_draws [n][d][C] and every
 * probability p, sorted[round((len - 1) p)] of ALL len = ranks x C x n draws of the coordinate (the reference's "2.5%", "25%",
 * "50%", "75%", "97.5%" are p = 0.025, 0.25, 0.5, 0.75, 0.975).  Radix select on the device (eight histogram passes over the
 * draws; the 256-bin counters are all-reduced over `rccl_comm` when the chains are sharded): exactly the element a sort would put
 * at that index.  n_probs <= 8; h_out [d][n_probs]. */
int fg_diag_quantiles(fg_engine *e, const double *d_draws, int n, int d, void *rccl_comm, const double *h_probs, int n_probs,
                      double *h_out);
/* The combination alone, on host buffers (no GPU needed): h_moments [d][6][m] of ALL chains in global chain order;
 * `acov` returns h_sums [d][n_lags] = sum over all chains of the biased lag-t autocovariances for t in
 * [lag0, lag0 + n_lags) (it is asked for 32 lags at a time, only as far as Geyer's sequence runs). */
typedef int (*fg_acov_fn)(void *user, int lag0, int n_lags, double *h_sums);
int fg_diag_combine(const double *h_moments, int64_t m, int n, int d, fg_acov_fn acov, void *user, double *h_rhat,
                    double *h_ess, double *h_mean, double *h_std);
/* The same combination from sums over chains (what FG_DIAG_REDUCE exchanges).  `reduce` returns sums over ALL chains of all ranks:
 * stage 1: h_out [d][6] = sums of the six moment rows; stage 2: h_in [d][2] = overall means {full chains, half chains},
 * h_out [d][2] = {sum_j (mean_j - in0)^2, sum_j (mean_h1_j - in1)^2 + (mean_h2_j - in1)^2} -- the reference's two-pass
 * between-chain sums of squares (diagnostics.rs:275-289, mcmc_utils.rs:296-304) with the pass over chains distributed. */
typedef int (*fg_reduce_fn)(void *user, int stage, const double *h_in, double *h_out);
int fg_diag_combine_reduced(int64_t m, int n, int d, fg_reduce_fn reduce, fg_acov_fn acov, void *user, double *h_rhat,
                            double *h_ess, double *h_mean, double *h_std);
/* RCCL communicator of the ranks of one run (one process per GPU): rank 0 obtains a 128-byte id (ncclGetUniqueId), the
 * host distributes it by any means, every rank calls fg_comm_init.  RCCL is bound at run time (librccl.so). */
int fg_comm_unique_id(void *out_128_bytes);
int fg_comm_init(fg_engine *e, int world_size, int rank, const void *unique_id_128_bytes, void **out_comm);
int fg_comm_destroy(void *comm);

/* raw device memory helpers so a host without a HIP binding can own draw buffers */
void *fg_device_alloc(fg_engine *e, size_t bytes);
int   fg_device_free(fg_engine *e, void *d_ptr);
int   fg_device_download(fg_engine *e, void *h_dst, const void *d_src, size_t bytes);
int   fg_device_upload(fg_engine *e, void *d_dst, const void *h_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* FUGUE_AMD_H */
