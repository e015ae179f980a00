/*
 * fugue_oracle.h -- CPU ORACLE for the fugue `src/inference` hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  It is a plain-C, per-chain
 * sequential restatement of the reference algorithm (alexnodeland/fugue,
 * crate fugue-ppl 0.2.0), written from a text reading of the reference
 * sources; every function cites the reference file:line it follows.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  The product library (fugue_amd/lib/libfugue_amd.so) never links,
 * imports or calls anything in this directory.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - log-pdfs / log-sum-exp / diagnostics: PINNED by the reference's own
 *     known-answer tests (tests/golden/ JSON files, harvested from
 *     src/core/distribution.rs:2073-2353,2525-2592,
 *     tests/f_dist_distributions.rs:32-332, tests/f_dist_numerical.rs:22-85)
 *     and by tests/gen_refs.py (the reference's own Python helper) run here.
 *   - samplers / RNG streams: PARITY UNPINNED.  The reference draws from
 *     rand 0.8.5 / rand_chacha 0.3.1 / rand_distr 0.4.3 (Cargo.lock), whose
 *     sources are not under /root/reference and no reference test pins a
 *     seeded draw.  Oracle and GPU share a counter-based Philox4x32-10
 *     stream instead, so oracle<->GPU comparisons are draw-for-draw.
 *   - HMC / MH / SMC control flow: restated line by line; pinned only by
 *     the reference's deterministic unit tests restated in tests/ and by the
 *     closed-form posterior targets of its statistical tests.
 */
#ifndef FUGUE_ORACLE_H
#define FUGUE_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Distribution kinds, in the order of the reference re-export list
 * (src/lib.rs:18-22). */
enum {
    ORC_BERNOULLI = 0, ORC_BETA, ORC_BINOMIAL, ORC_CATEGORICAL, ORC_CAUCHY,
    ORC_CHISQUARED, ORC_DISCRETEUNIFORM, ORC_EXPONENTIAL, ORC_GAMMA,
    ORC_INVERSEGAMMA, ORC_LAPLACE, ORC_LOGNORMAL, ORC_NORMAL, ORC_POISSON,
    ORC_STUDENTT, ORC_UNIFORM, ORC_WEIBULL, ORC_N_DISTS
};

/* Value types (ChoiceValue tags, src/runtime/trace.rs:32-43). */
enum { ORC_F64 = 0, ORC_BOOL, ORC_U64, ORC_USIZE, ORC_I64 };

/* Expression node ops (superset of the DSL Expr set,
 * crates/fugue-wasm/src/dsl.rs:92-102,569-582). */
enum {
    ORC_X_CONST = 0, ORC_X_SITE, ORC_X_DATA, ORC_X_NEG, ORC_X_ADD, ORC_X_SUB,
    ORC_X_MUL, ORC_X_DIV, ORC_X_EXP, ORC_X_LN, ORC_X_SQRT, ORC_X_ABS,
    ORC_X_FLOOR, ORC_X_SIN, ORC_X_COS, ORC_X_TANH, ORC_X_POW, ORC_X_MIN,
    ORC_X_MAX, ORC_X_CLAMP, ORC_X_SELECT
};

enum { ORC_STMT_SAMPLE = 0, ORC_STMT_OBSERVE = 1, ORC_STMT_FACTOR = 2 };

/* One 8-byte trace cell: f64 sites hold .f, all discrete sites hold .i
 * (bool as 0/1). */
typedef union { double f; int64_t i; } orc_cell;

typedef struct orc_model orc_model;

/* ---- scalar numerics (src/core/distribution.rs, src/core/numerical.rs) ---- */
double orc_logpdf(int dist, int is_int, double xf, int64_t xi,
                  const double *params, int nparams);
double orc_log_sum_exp(const double *x, size_t n);
void   orc_normalize_log_probs(const double *x, size_t n, double *out);
double orc_log1p_exp(double x);
double orc_safe_ln(double x);

/* ---- counter-based RNG (shared spec with the GPU engine) ---- */
typedef struct { uint32_t key0, key1, c0, c1, c2, c3; } orc_stream;
enum {
    ORC_RNG_PRIOR = 1, ORC_RNG_HMC = 2, ORC_RNG_EPS = 3, ORC_RNG_MH = 4,
    ORC_RNG_SMC_RESAMPLE = 5, ORC_RNG_SMC_REJUV = 6, ORC_RNG_SMC_PRIOR = 7
};
void   orc_stream_init(orc_stream *s, uint64_t seed, uint32_t chain,
                       uint32_t iter, uint32_t purpose);
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                         uint32_t out[4]);
void   orc_stream_block(orc_stream *s, uint64_t *a, uint64_t *b);
double orc_stream_u01(orc_stream *s);
double orc_stream_normal(orc_stream *s);
void   orc_stream_normal_pair(orc_stream *s, double *z0, double *z1);
double orc_stream_gaussian_z(orc_stream *s);
/* sample one value from a distribution (prior init / prior-resample) */
orc_cell orc_sample_dist(int dist, const double *params, int nparams,
                         orc_stream *s);

/* ---- model building ---- */
orc_model *orc_model_new(void);
void  orc_model_free(orc_model *m);
int   orc_model_add_data(orc_model *m, const double *v, int n);
int   orc_model_add_node(orc_model *m, int op, int a, int b, int c, double v);
int   orc_model_add_args(orc_model *m, const int *ids, int n);
/* params: node ids; for ORC_CATEGORICAL pass the K probability nodes.
 * value: node id for observe/factor (ignored for sample).
 * returns the site handle (program-order sample index) for samples,
 * statement index otherwise, or <0 on error. */
int   orc_model_add_stmt(orc_model *m, int kind, int dist, const char *addr,
                         const int *params, int nparams, int value);
int   orc_model_finalize(orc_model *m);   /* 0 ok, 301 duplicate address */
int   orc_model_n_sites(const orc_model *m);
int   orc_model_n_f64(const orc_model *m);
int   orc_model_n_observe(const orc_model *m);
const char *orc_model_site_name(const orc_model *m, int sorted_idx);
int   orc_model_site_vtype(const orc_model *m, int sorted_idx);
int   orc_model_site_of_handle(const orc_model *m, int handle);
int   orc_model_f64_site(const orc_model *m, int k); /* k-th f64 site -> sorted idx */

/* ---- model runs (src/runtime/handler.rs:124-209, interpreters.rs:76-163) ---- */
/* acc[0..2] = log_prior, log_likelihood, log_factors */
void orc_run_score(const orc_model *m, const orc_cell *values, double acc[3],
                   double *logp /* [S] or NULL */);
void orc_run_prior(const orc_model *m, orc_stream *s, orc_cell *values,
                   double acc[3], double *logp);

/* ---- HMC (src/inference/hmc.rs) ---- */
typedef struct {
    int32_t n_leapfrog;       /* default 16 */
    double  target_accept;    /* 0.8 */
    double  init_step_size;   /* NaN = None */
    double  finite_diff_eps;  /* 1e-5 */
    int32_t adapt_mass;       /* 0 */
} orc_hmc_config;

double orc_log_joint_at(const orc_model *m, const orc_cell *base,
                        const double *q);
int    orc_grad_log_joint(const orc_model *m, const orc_cell *base,
                          const double *q, double h, double *g);
int    orc_leapfrog(const orc_model *m, const orc_cell *base, const double *q0,
                    const double *p0, double eps, int l, double h,
                    const double *m_inv, double *q, double *p);
/* transition with injected momentum p0[d] and uniform u */
void   orc_hmc_transition(const orc_model *m, const orc_cell *base,
                          const double *q_cur, double lj_cur, double eps,
                          int l, double h, const double *m_inv,
                          const double *p0, double u, double *q_out,
                          double *lj_out, int *accepted, double *alpha,
                          int *divergent);
double orc_find_reasonable_epsilon(const orc_model *m, const orc_cell *base,
                                   const double *q, double lj_q, double h,
                                   const double *m_inv, const double *p0);
void   orc_hmc_momentum(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose,
                        const double *mass_sqrt, int d, double *p0, double *u);
double orc_dual_averaging_run(double eps0, double target, const double *alphas,
                              int n, double *eps_trace, double *frozen);

typedef struct {
    double  accept_rate;   /* mean over chains & all transitions */
    double  mean_step_size;/* mean over chains of the final (frozen) eps */
    int64_t n_divergent;
    int64_t n_model_evals;
} orc_hmc_stats;

/* Runs chains [chain0, chain0+n_chains) of `hmc_chain`.
 * draws: [n_samples][d][n_chains] (may be NULL); final_values [S][n_chains];
 * final_eps [n_chains] (may be NULL). */
void orc_hmc_run(const orc_model *m, const orc_hmc_config *cfg, uint64_t seed,
                 uint32_t chain0, int n_chains, int n_warmup, int n_samples,
                 double *draws, orc_cell *final_values, double *final_eps,
                 orc_hmc_stats *stats, int n_threads);

/* ---- MH (src/inference/mh.rs, src/inference/mcmc_utils.rs:30-175) ---- */
enum { ORC_PROP_AUTO = 0, ORC_PROP_GAUSSIAN = 1, ORC_PROP_LOGSPACE = 2,
       ORC_PROP_REFLECT = 3, ORC_PROP_PRIOR = 4 };
typedef struct { int32_t kind; double lower, upper; } orc_site_proposal;

double orc_adapt_update(double *scale, double *log_scale, int64_t *acc,
                        int64_t *tot, int accepted, double target, double gamma);

typedef struct {
    double  accept_rate;
    int64_t n_model_evals;
} orc_mh_stats;

/* overrides: [S] in sorted site order or NULL.  rec_sites: sorted site idx to
 * record; draws: [n_samples][n_rec][n_chains] cells.  scales_out [S][n_chains]. */
void orc_mh_run(const orc_model *m, uint64_t seed, uint32_t chain0,
                int n_chains, int n_warmup, int n_samples,
                const orc_site_proposal *overrides, const int *rec_sites,
                int n_rec, orc_cell *draws, orc_cell *final_values,
                double *scales_out, orc_mh_stats *stats, int n_threads);

/* ---- SMC (src/inference/smc.rs) ---- */
enum { ORC_RESAMPLE_MULTINOMIAL = 0, ORC_RESAMPLE_SYSTEMATIC = 1,
       ORC_RESAMPLE_STRATIFIED = 2 };
typedef struct {
    int32_t resampling_method; /* default systematic */
    double  ess_threshold;     /* 0.5 */
    int32_t rejuvenation_steps;/* 0 */
    int32_t batched_adaptation;/* 0 = reference sequential shared adaptation;
                                  1 = per-sweep batched update (GPU semantics) */
} orc_smc_config;

void   orc_systematic_indices(const double *w, int64_t n, double u, int64_t *idx);
void   orc_stratified_indices(const double *w, int64_t n, const double *u, int64_t *idx);
void   orc_multinomial_indices(const double *w, int64_t n, const double *u, int64_t *idx);
double orc_ess_particles(const double *w, int64_t n);
double orc_next_beta(double beta, const double *log_w, const double *ll,
                     int64_t n, double target_ess);

/* values [S][N], log_w [N], weights [N]; betas: up to max_betas entries. */
int orc_smc_run(const orc_model *m, int64_t n, const orc_smc_config *cfg,
                uint64_t seed, orc_cell *values, double *log_w, double *weights,
                double *log_evidence, double *betas, int max_betas,
                int64_t *n_model_evals);

/* ---- diagnostics (src/inference/diagnostics.rs, mcmc_utils.rs:195-421) ---- */
/* chains: row-major [m][n] */
double orc_split_rhat(const double *chains, int m, int n);
double orc_classic_rhat(const double *chains, int m, int n);
double orc_ess_multichain(const double *chains, int m, int n);
double orc_ess_single(const double *x, int n);
double orc_geweke(const double *x, int n);
/* out: mean, std, q2.5, q25, q50, q75, q97.5, rhat, ess */
void   orc_summarize(const double *chains, int m, int n, double out[9]);

#ifdef __cplusplus
}
#endif
#endif
