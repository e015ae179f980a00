/*
 * orc_inference.c -- CPU ORACLE (test infrastructure, not the product):
 * per-chain sequential restatement of hmc_chain (src/inference/hmc.rs),
 * adaptive_mcmc_chain (src/inference/mh.rs + mcmc_utils.rs:30-175),
 * adaptive_smc (src/inference/smc.rs) and the R-hat / ESS diagnostics
 * (src/inference/diagnostics.rs, mcmc_utils.rs:195-421).
 *
 * RNG: the reference threads one `&mut R` (ChaCha12) through everything; its
 * streams are unpinned (fugue_oracle.h), so each (chain, iteration, purpose)
 * gets its own Philox stream and draws are taken in the reference's
 * documented order inside that stream.
 */
#define _GNU_SOURCE
#include "orc_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NEG_INF (-INFINITY)

/* ===================================================================== */
/* HMC                                                                    */
/* ===================================================================== */

static void set_positions(const orc_model *m, orc_cell *vals, const double *q) {
    for (int k = 0; k < m->n_f64; k++) vals[m->f64_sites[k]].f = q[k];
}

/* log_joint_at: hmc.rs:264-279 (trace_with_positions :252-260 + one
 * ScoreGivenTrace run; discrete sites keep the base trace's values). */
double orc_log_joint_at(const orc_model *m, const orc_cell *base, const double *q) {
    int S = m->n_samples;
    orc_cell *v = (orc_cell *)alloca((size_t)(S + 1) * sizeof(orc_cell));
    memcpy(v, base, (size_t)S * sizeof(orc_cell));
    set_positions(m, v, q);
    double acc[3];
    orc_run_score(m, v, acc, NULL);
    return acc[0] + acc[1] + acc[2];
}

/* grad_log_joint: hmc.rs:304-329 -- central FD, one coordinate at a time. */
int orc_grad_log_joint(const orc_model *m, const orc_cell *base, const double *q, double h, double *g) {
    int d = m->n_f64, ok = 1;
    double *qq = (double *)alloca((size_t)(d + 1) * sizeof(double));
    memcpy(qq, q, (size_t)d * sizeof(double));
    for (int i = 0; i < d; i++) {
        double orig = qq[i];
        qq[i] = orig + h;
        double lp = orc_log_joint_at(m, base, qq);
        qq[i] = orig - h;
        double lm = orc_log_joint_at(m, base, qq);
        qq[i] = orig;
        double gi = (lp - lm) / (2.0 * h);
        if (!isfinite(gi)) ok = 0;
        g[i] = gi;
    }
    return ok;
}

/* leapfrog: hmc.rs:353-407.  Returns 1 when divergent. */
int orc_leapfrog(const orc_model *m, const orc_cell *base, const double *q0, const double *p0,
                 double eps, int l, double h, const double *m_inv, double *q, double *p) {
    int d = m->n_f64;
    double *grad = (double *)alloca((size_t)(d + 1) * sizeof(double));
    memcpy(q, q0, (size_t)d * sizeof(double));
    memcpy(p, p0, (size_t)d * sizeof(double));
    if (!orc_grad_log_joint(m, base, q, h, grad)) return 1;
    for (int s = 0; s < l; s++) {
        for (int i = 0; i < d; i++) p[i] += 0.5 * eps * grad[i];
        for (int i = 0; i < d; i++) q[i] += eps * m_inv[i] * p[i];
        if (!orc_grad_log_joint(m, base, q, h, grad)) return 1;
        for (int i = 0; i < d; i++) p[i] += 0.5 * eps * grad[i];
    }
    return 0;
}

static double kinetic(const double *p, const double *m_inv, int d) {
    double s = 0.0;
    for (int i = 0; i < d; i++) s += p[i] * p[i] * m_inv[i];
    return 0.5 * s;
}

/* hmc_transition: hmc.rs:419-473 with the momentum and the accept uniform
 * injected (u is consumed only on the non-divergent path). */
void orc_hmc_transition(const orc_model *m, const orc_cell *base, const double *q_cur, double lj_cur,
                        double eps, int l, double h, const double *m_inv, const double *p0, double u,
                        double *q_out, double *lj_out, int *accepted, double *alpha, int *divergent) {
    int d = m->n_f64;
    double *qn = (double *)alloca((size_t)(d + 1) * sizeof(double));
    double *pn = (double *)alloca((size_t)(d + 1) * sizeof(double));
    double h0 = -lj_cur + kinetic(p0, m_inv, d);
    memcpy(q_out, q_cur, (size_t)d * sizeof(double));
    *lj_out = lj_cur; *accepted = 0; *alpha = 0.0; *divergent = 0;
    if (orc_leapfrog(m, base, q_cur, p0, eps, l, h, m_inv, qn, pn)) { *divergent = 1; return; }
    double lj_new = orc_log_joint_at(m, base, qn);   /* score_full: hmc.rs:283-299 */
    if (!isfinite(lj_new)) { *divergent = 1; return; }
    double h_new = -lj_new + kinetic(pn, m_inv, d);
    double ap = fmin(exp(h0 - h_new), 1.0);
    *alpha = ap;
    if (u < ap) { *accepted = 1; memcpy(q_out, qn, (size_t)d * sizeof(double)); *lj_out = lj_new; }
}

/* find_reasonable_epsilon: hmc.rs:479-535, momentum p0 injected. */
double orc_find_reasonable_epsilon(const orc_model *m, const orc_cell *base, const double *q, double lj_q,
                                   double h, const double *m_inv, const double *p0) {
    int d = m->n_f64;
    double *q1 = (double *)alloca((size_t)(d + 1) * sizeof(double));
    double *p1 = (double *)alloca((size_t)(d + 1) * sizeof(double));
    double h0 = -lj_q + kinetic(p0, m_inv, d);
#define LOG_RATIO_AT(EPS, OUT) do { \
        if (orc_leapfrog(m, base, q, p0, (EPS), 1, h, m_inv, q1, p1)) { (OUT) = NEG_INF; } \
        else { double lj1 = orc_log_joint_at(m, base, q1); \
               if (!isfinite(lj1)) (OUT) = NEG_INF; \
               else (OUT) = h0 - (-lj1 + kinetic(p1, m_inv, d)); } } while (0)
    double eps = 1.0, lr;
    LOG_RATIO_AT(eps, lr);
    double ln_half = log(0.5), ln2 = log(2.0);
    double a = (lr > ln_half) ? 1.0 : -1.0;
    unsigned iters = 0;
    while (a * lr > -a * ln2) {
        eps *= pow(2.0, a);
        LOG_RATIO_AT(eps, lr);
        iters += 1;
        if (iters > 100 || !(eps >= 1e-12 && eps <= 1e12)) break;
        if (a > 0.0 && lr == NEG_INF) { eps /= 2.0; break; }
    }
#undef LOG_RATIO_AT
    return fmin(fmax(eps, 1e-6), 1e3);
}

/* DualAveraging: hmc.rs:141-184 */
typedef struct { double mu, log_eps_bar, h_bar; uint64_t m; double gamma, t0, kappa, target; } da_t;
static void da_new(da_t *da, double eps0, double target) {
    da->mu = log(10.0 * eps0); da->log_eps_bar = 0.0; da->h_bar = 0.0; da->m = 0;
    da->gamma = 0.05; da->t0 = 10.0; da->kappa = 0.75; da->target = target;
}
static double da_update(da_t *da, double alpha) {
    da->m += 1;
    double mm = (double)da->m;
    double a = alpha < 0.0 ? 0.0 : (alpha > 1.0 ? 1.0 : alpha);   /* f64::clamp */
    double frac = 1.0 / (mm + da->t0);
    da->h_bar = (1.0 - frac) * da->h_bar + frac * (da->target - a);
    double log_eps = da->mu - (sqrt(mm) / da->gamma) * da->h_bar;
    double w = pow(mm, -da->kappa);
    da->log_eps_bar = w * log_eps + (1.0 - w) * da->log_eps_bar;
    return exp(log_eps);
}
double orc_dual_averaging_run(double eps0, double target, const double *alphas, int n,
                              double *eps_trace, double *frozen) {
    da_t da; da_new(&da, eps0, target);
    double e = eps0;
    for (int i = 0; i < n; i++) { e = da_update(&da, alphas[i]); if (eps_trace) eps_trace[i] = e; }
    if (frozen) *frozen = exp(da.log_eps_bar);
    return e;
}

static void draw_momentum(orc_stream *st, const double *mass_sqrt, int d, double *p0) {
    for (int i = 0; i < d; i += 2) {
        double z0, z1;
        orc_stream_normal_pair(st, &z0, &z1);
        p0[i] = z0 * mass_sqrt[i];
        if (i + 1 < d) p0[i + 1] = z1 * mass_sqrt[i + 1];
    }
}

/* The momentum (p0 = z * mass_sqrt, hmc.rs:436-441) and accept uniform of transition `iter`
 * of chain `chain` (purpose = ORC_RNG_HMC), or of eps-search instance `iter` (ORC_RNG_EPS). */
void orc_hmc_momentum(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose, const double *mass_sqrt, int d,
                      double *p0, double *u) {
    orc_stream st;
    double *ones = NULL;
    if (!mass_sqrt) { ones = (double *)malloc(((size_t)d + 1) * sizeof(double)); for (int i = 0; i < d; i++) ones[i] = 1.0; mass_sqrt = ones; }
    orc_stream_init(&st, seed, chain, iter, purpose);
    draw_momentum(&st, mass_sqrt, d, p0);
    if (u) *u = orc_stream_u01(&st);
    free(ones);
}

/* one chain of hmc_chain (hmc.rs:566-583) via the HmcSession state machine
 * (hmc.rs:667-729, 819-919). */
static void hmc_one_chain(const orc_model *m, const orc_hmc_config *cfg, uint64_t seed, uint32_t chain,
                          int n_warmup, int n_samples, double *draws, int col, int n_cols,
                          orc_cell *final_values, double *final_eps, double *acc_sum, int64_t *n_div,
                          int64_t *n_evals) {
    int S = m->n_samples, d = m->n_f64;
    orc_cell *vals = (orc_cell *)calloc((size_t)S + 1, sizeof(orc_cell));
    double *q = (double *)calloc((size_t)d + 1, sizeof(double));
    double *qn = (double *)calloc((size_t)d + 1, sizeof(double));
    double *p0 = (double *)calloc((size_t)d + 1, sizeof(double));
    double *m_inv = (double *)malloc(((size_t)d + 1) * sizeof(double));
    double *mass_sqrt = (double *)malloc(((size_t)d + 1) * sizeof(double));
    double *wmean = (double *)calloc((size_t)d + 1, sizeof(double));
    double *wm2 = (double *)calloc((size_t)d + 1, sizeof(double));
    uint64_t wn = 0;
    double acc[3];
    orc_stream st;

    orc_stream_init(&st, seed, chain, 0, ORC_RNG_PRIOR);
    orc_run_prior(m, &st, vals, acc, NULL);                 /* hmc.rs:673-679 */
    for (int k = 0; k < d; k++) q[k] = vals[m->f64_sites[k]].f;   /* :238-248 */
    double h = cfg->finite_diff_eps;
    int l = cfg->n_leapfrog > 1 ? cfg->n_leapfrog : 1;      /* :684 */
    for (int i = 0; i < d; i++) { m_inv[i] = 1.0; mass_sqrt[i] = 1.0; }
    double lj_cur = acc[0] + acc[1] + acc[2];
    double eps0;
    if (d == 0) eps0 = 1.0;
    else if (!isnan(cfg->init_step_size)) eps0 = cfg->init_step_size;
    else {
        orc_stream_init(&st, seed, chain, 0, ORC_RNG_EPS);
        draw_momentum(&st, mass_sqrt, d, p0);
        eps0 = orc_find_reasonable_epsilon(m, vals, q, lj_cur, h, m_inv, p0);
    }
    da_t da; da_new(&da, eps0, cfg->target_accept);
    int mass_adapt_at = (cfg->adapt_mass && n_warmup >= 4) ? n_warmup / 2 : -1;   /* :704-708 */
    double eps = eps0, frozen_eps = NAN;
    int total = n_warmup + n_samples;

    for (int iter = 0; iter < total; iter++) {
        if (d == 0) {                                       /* :826-845 */
            orc_stream_init(&st, seed, chain, (uint32_t)(iter + 1), ORC_RNG_PRIOR);
            orc_run_prior(m, &st, vals, acc, NULL);
            lj_cur = acc[0] + acc[1] + acc[2];
            *acc_sum += 1.0;
        } else {
            int warming = iter < n_warmup;
            double e;
            if (warming) e = eps;
            else {                                          /* :789-798, 848-854 */
                if (!isnan(frozen_eps)) e = frozen_eps;
                else if (n_warmup > 0) e = exp(da.log_eps_bar);
                else e = eps;
                frozen_eps = e;
            }
            orc_stream_init(&st, seed, chain, (uint32_t)iter, ORC_RNG_HMC);
            draw_momentum(&st, mass_sqrt, d, p0);           /* :436-441 */
            /* the accept uniform is the next draw of the same stream; it is
             * only *used* on the non-divergent path (:447-461) */
            double u = orc_stream_u01(&st);
            int accepted, divergent; double alpha, lj_new;
            orc_hmc_transition(m, vals, q, lj_cur, e, l, h, m_inv, p0, u, qn, &lj_new, &accepted, &alpha, &divergent);
            *n_evals += (int64_t)(2 * d) * (l + 1) + 1;
            if (accepted) { memcpy(q, qn, (size_t)d * sizeof(double)); lj_cur = lj_new; set_positions(m, vals, q); }
            if (divergent) *n_div += 1;
            *acc_sum += alpha;
            if (warming) {                                  /* :880-909 */
                eps = da_update(&da, alpha);
                if (mass_adapt_at >= 0) {                   /* Welford push :202-211 */
                    wn += 1; double n = (double)wn;
                    for (int i = 0; i < d; i++) {
                        double delta = q[i] - wmean[i];
                        wmean[i] += delta / n;
                        double delta2 = q[i] - wmean[i];
                        wm2[i] += delta * delta2;
                    }
                }
                if (iter + 1 == mass_adapt_at) {
                    for (int i = 0; i < d; i++) {           /* variances :216-232 */
                        double v = 1.0;
                        if (wn >= 2) { double vv = wm2[i] / (double)(wn - 1); v = (isfinite(vv) && vv > 1e-8) ? vv : 1.0; }
                        m_inv[i] = v; mass_sqrt[i] = sqrt(1.0 / v);
                    }
                    orc_stream_init(&st, seed, chain, 1, ORC_RNG_EPS);
                    draw_momentum(&st, mass_sqrt, d, p0);
                    double er = orc_find_reasonable_epsilon(m, vals, q, lj_cur, h, m_inv, p0);
                    da_new(&da, er, cfg->target_accept);
                    eps = er;
                }
            }
        }
        if (iter >= n_warmup && draws) {                    /* :577-582 */
            int t = iter - n_warmup;
            for (int k = 0; k < d; k++) draws[((size_t)t * d + k) * n_cols + col] = q[k];
        }
    }
    if (final_values) for (int j = 0; j < S; j++) final_values[(size_t)j * n_cols + col] = vals[j];
    if (final_eps) {
        double e = !isnan(frozen_eps) ? frozen_eps : (n_warmup > 0 ? exp(da.log_eps_bar) : eps);
        if (total <= n_warmup) e = eps;
        final_eps[col] = e;
    }
    free(vals); free(q); free(qn); free(p0); free(m_inv); free(mass_sqrt); free(wmean); free(wm2);
}

void orc_hmc_run(const orc_model *m, const orc_hmc_config *cfg, uint64_t seed, uint32_t chain0,
                 int n_chains, int n_warmup, int n_samples, double *draws, orc_cell *final_values,
                 double *final_eps, orc_hmc_stats *stats, int n_threads) {
    double acc_sum = 0.0; int64_t n_div = 0, n_evals = 0;
    double eps_sum = 0.0;
    double *eps_buf = final_eps ? final_eps : (double *)malloc((size_t)n_chains * sizeof(double));
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1) reduction(+ : acc_sum, n_div, n_evals)
#endif
    for (int c = 0; c < n_chains; c++) {
        double a = 0.0; int64_t dv = 0, ev = 0;
        hmc_one_chain(m, cfg, seed, chain0 + (uint32_t)c, n_warmup, n_samples, draws, c, n_chains,
                      final_values, eps_buf, &a, &dv, &ev);
        acc_sum += a; n_div += dv; n_evals += ev;
    }
    for (int c = 0; c < n_chains; c++) eps_sum += eps_buf[c];
    if (stats) {
        int total = n_warmup + n_samples;
        stats->accept_rate = total > 0 ? acc_sum / ((double)n_chains * total) : 0.0;
        stats->mean_step_size = eps_sum / n_chains;
        stats->n_divergent = n_div; stats->n_model_evals = n_evals;
    }
    if (!final_eps) free(eps_buf);
}

/* ===================================================================== */
/* MH                                                                     */
/* ===================================================================== */

/* DiminishingAdaptation::update: mcmc_utils.rs:88-150 (one site). */
double orc_adapt_update(double *scale, double *log_scale, int64_t *acc, int64_t *tot, int accepted,
                        double target, double gamma) {
    *tot += 1;
    if (accepted) *acc += 1;
    if (*tot < 10) return *scale;
    double rate = (double)*acc / (double)*tot;
    double step = 1.0 / pow((double)*tot, gamma);
    *log_scale += step * (rate - target);
    double ns = exp(*log_scale);
    *scale = (isfinite(ns) && ns > 0.0) ? fmin(fmax(ns, 0.001), 100.0) : 1.0;
    if (*scale == 1.0) *log_scale = 0.0; else *log_scale = log(*scale);
    return *scale;
}

/* normal_logpdf: mh.rs:135-138 */
static double normal_logpdf(double x, double mean, double sd) {
    double z = (x - mean) / sd;
    return -0.5 * z * z - log(sd) - 0.5 * log(2.0 * M_PI);
}
/* LogSpaceWalkProposal::log_proposal_prob: mh.rs:217-223 */
static double logspace_lq(double from, double to, double scale) {
    if (from <= 0.0 || to <= 0.0) return 0.0;
    return normal_logpdf(log(to), log(from), scale) - log(to);
}
static int64_t f2i_sat(double v) {
    if (isnan(v)) return 0;
    if (v >= 9223372036854775807.0) return INT64_MAX;
    if (v <= -9223372036854775808.0) return INT64_MIN;
    return (int64_t)v;
}
static double stmt_lp(const orc_stmt *s, const double *p, int np, orc_cell x) {
    return s->vtype == ORC_F64 ? orc_logpdf(s->dist, 0, x.f, 0, p, np) : orc_logpdf(s->dist, 1, 0.0, x.i, p, np);
}
static orc_cell obs_value(const orc_model *m, const orc_stmt *s, const orc_cell *vals) {
    double v = orc_eval(m, s->value, vals);
    orc_cell c;
    if (s->vtype == ORC_F64) c.f = v;
    else if (s->vtype == ORC_BOOL) c.i = (v != 0.0);
    else c.i = isfinite(v) ? (int64_t)v : 0;
    return c;
}

/* One run of SingleSiteProposalHandler (mh.rs:361-617) for a fixed-structure
 * model: propose at `target`, replay and re-score everything else.
 * pv: in = current values, out = proposed values. */
static void propose_and_score(const orc_model *m, orc_stream *st, orc_cell *pv, int target, double scale,
                              const orc_site_proposal *ov, int8_t *kind_cache, double acc[3],
                              double *lqf, double *lqr) {
    double p[ORC_MAX_PARAMS];
    acc[0] = acc[1] = acc[2] = 0.0; *lqf = 0.0; *lqr = 0.0;
    for (int i = 0; i < m->n_stmts; i++) {
        const orc_stmt *s = &m->stmts[i];
        if (s->kind == ORC_STMT_FACTOR) { acc[2] += orc_eval(m, s->value, pv); continue; }
        int np = orc_stmt_params(m, s, pv, p);
        if (s->kind == ORC_STMT_OBSERVE) { acc[1] += stmt_lp(s, p, np, obs_value(m, s, pv)); continue; }
        int j = s->sorted;
        if (j == target) {
            orc_cell cur = pv[j], prop = cur;
            if (s->vtype == ORC_F64) {
                int kind;                                   /* f64_kind: mh.rs:339-358 */
                if (ov && ov[j].kind != ORC_PROP_AUTO) kind = ov[j].kind;
                else if (kind_cache[j]) kind = kind_cache[j];
                else {
                    orc_cell probe; probe.f = -1.0;         /* NEG_SUPPORT_PROBE */
                    kind = (cur.f > 0.0 && !isfinite(stmt_lp(s, p, np, probe))) ? ORC_PROP_LOGSPACE : ORC_PROP_GAUSSIAN;
                    kind_cache[j] = (int8_t)kind;
                }
                double f = 0.0, r = 0.0;
                switch (kind) {
                case ORC_PROP_GAUSSIAN:                     /* mh.rs:183-187 */
                    prop.f = cur.f + scale * orc_stream_gaussian_z(st); break;
                case ORC_PROP_LOGSPACE:                     /* mh.rs:201-224 */
                    if (cur.f <= 0.0) prop.f = 2.2250738585072014e-308;
                    else {
                        double z = orc_stream_gaussian_z(st);
                        double pr = exp(log(cur.f) + scale * z);
                        prop.f = isfinite(pr) ? fmax(pr, 2.2250738585072014e-308) : 1.7976931348623157e308;
                    }
                    f = logspace_lq(cur.f, prop.f, scale); r = logspace_lq(prop.f, cur.f, scale); break;
                case ORC_PROP_REFLECT: {                    /* mh.rs:237-257 */
                    double lo = ov[j].lower, hi = ov[j].upper;
                    double pr = cur.f + scale * orc_stream_gaussian_z(st);
                    if (hi - lo <= 0.0) prop.f = cur.f;
                    else {
                        while (pr < lo || pr > hi) { if (pr < lo) pr = 2.0 * lo - pr; if (pr > hi) pr = 2.0 * hi - pr; }
                        prop.f = pr < lo ? lo : (pr > hi ? hi : pr);
                    }
                    break; }
                case ORC_PROP_PRIOR:                        /* mh.rs:400-403 */
                    prop = orc_sample_dist(s->dist, p, np, st);
                    f = stmt_lp(s, p, np, prop); r = stmt_lp(s, p, np, cur); break;
                }
                *lqf += f; *lqr += r;
            } else if (s->vtype == ORC_BOOL) {              /* FlipProposal mh.rs:263-269 */
                prop.i = !cur.i;
            } else if (s->vtype == ORC_U64) {               /* DiscreteWalk mh.rs:285-294 */
                int64_t delta = f2i_sat(round(scale * orc_stream_gaussian_z(st)));
                int64_t k = cur.i + delta;
                prop.i = k >= 0 ? k : -k - 1;
            } else if (s->vtype == ORC_USIZE) {             /* prior resample mh.rs:516-530 */
                prop = orc_sample_dist(s->dist, p, np, st);
                *lqf += stmt_lp(s, p, np, prop);
                *lqr += stmt_lp(s, p, np, cur);
            } else {                                        /* i64 walk mh.rs:557-567 */
                int64_t delta = f2i_sat(round(scale * orc_stream_gaussian_z(st)));
                prop.i = cur.i + delta;
            }
            pv[j] = prop;
        }
        acc[0] += stmt_lp(s, p, np, pv[j]);
    }
}

static void mh_one_chain(const orc_model *m, uint64_t seed, uint32_t chain, int n_warmup, int n_samples,
                         const orc_site_proposal *ov, const int *rec, int n_rec, orc_cell *draws, int col,
                         int n_cols, orc_cell *final_values, double *scales_out, int64_t *n_acc, int64_t *n_evals) {
    int S = m->n_samples;
    orc_cell *cur = (orc_cell *)calloc((size_t)S + 1, sizeof(orc_cell));
    orc_cell *pv = (orc_cell *)calloc((size_t)S + 1, sizeof(orc_cell));
    double *scale = (double *)malloc(((size_t)S + 1) * sizeof(double));
    double *lscale = (double *)calloc((size_t)S + 1, sizeof(double));
    int64_t *acc_n = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    int64_t *tot_n = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    int8_t *kind_cache = (int8_t *)calloc((size_t)S + 1, 1);
    for (int j = 0; j < S; j++) scale[j] = 1.0;            /* get_scale: mcmc_utils.rs:70-77 */
    double acc[3];
    orc_stream st;
    orc_stream_init(&st, seed, chain, 0, ORC_RNG_PRIOR);
    orc_run_prior(m, &st, cur, acc, NULL);                  /* mh.rs:950-957 */
    double cur_lw = acc[0] + acc[1] + acc[2];
    int total = n_warmup + n_samples;
    for (int iter = 0; iter < total; iter++) {
        int adapt = iter < n_warmup;
        if (S > 0) {                                        /* single_site_mh_step: mh.rs:698-744 */
            orc_stream_init(&st, seed, chain, (uint32_t)iter, ORC_RNG_MH);
            uint64_t ra, rb; orc_stream_block(&st, &ra, &rb);
            int target = (int)(((unsigned __int128)ra * (uint64_t)S) >> 64);   /* gen_range(0..S) */
            double sc = scale[target];
            memcpy(pv, cur, (size_t)S * sizeof(orc_cell));
            double lqf, lqr;
            propose_and_score(m, &st, pv, target, sc, ov, kind_cache, acc, &lqf, &lqr);
            *n_evals += 1;
            double prop_lw = acc[0] + acc[1] + acc[2];
            double dim_term = log((double)S) - log((double)S);
            double log_alpha = prop_lw - cur_lw + (lqr - lqf) + dim_term;
            int accept = (log_alpha >= 0.0) || (orc_stream_u01(&st) < exp(log_alpha));
            if (adapt) orc_adapt_update(&scale[target], &lscale[target], &acc_n[target], &tot_n[target], accept, 0.44, 0.7);
            if (accept) { memcpy(cur, pv, (size_t)S * sizeof(orc_cell)); cur_lw = prop_lw; *n_acc += 1; }
        }
        if (!adapt && draws) {
            int t = iter - n_warmup;
            for (int r = 0; r < n_rec; r++) draws[((size_t)t * n_rec + r) * n_cols + col] = cur[rec[r]];
        }
    }
    if (final_values) for (int j = 0; j < S; j++) final_values[(size_t)j * n_cols + col] = cur[j];
    if (scales_out) for (int j = 0; j < S; j++) scales_out[(size_t)j * n_cols + col] = scale[j];
    free(cur); free(pv); free(scale); free(lscale); free(acc_n); free(tot_n); free(kind_cache);
}

void orc_mh_run(const orc_model *m, uint64_t seed, uint32_t chain0, int n_chains, int n_warmup, int n_samples,
                const orc_site_proposal *overrides, const int *rec_sites, int n_rec, orc_cell *draws,
                orc_cell *final_values, double *scales_out, orc_mh_stats *stats, int n_threads) {
    int64_t n_acc = 0, n_evals = 0;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1) reduction(+ : n_acc, n_evals)
#endif
    for (int c = 0; c < n_chains; c++) {
        int64_t a = 0, e = 0;
        mh_one_chain(m, seed, chain0 + (uint32_t)c, n_warmup, n_samples, overrides, rec_sites, n_rec, draws, c,
                     n_chains, final_values, scales_out, &a, &e);
        n_acc += a; n_evals += e;
    }
    if (stats) {
        int total = n_warmup + n_samples;
        stats->accept_rate = total > 0 ? (double)n_acc / ((double)n_chains * total) : 0.0;
        stats->n_model_evals = n_evals;
    }
}

/* ===================================================================== */
/* SMC                                                                    */
/* ===================================================================== */

/* effective_sample_size: smc.rs:230-233 */
double orc_ess_particles(const double *w, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += w[i] * w[i];
    return 1.0 / s;
}
/* systematic_indices: smc.rs:255-272.  u = U/n is formed here from U.
 * Deviation: U == 0.0 makes `i - 1` underflow in the reference (debug panic /
 * release wrap to the last particle); idx_0 = 0 is used instead. */
void orc_systematic_indices(const double *w, int64_t n, double U, int64_t *idx) {
    double u = U / (double)n, cum = 0.0; int64_t i = 0;
    for (int64_t j = 0; j < n; j++) {
        double thr = u + (double)j / (double)n;
        while (cum < thr && i < n) { cum += w[i]; i++; }
        int64_t k = i - 1; if (k < 0) k = 0;
        idx[j] = k < n - 1 ? k : n - 1;
    }
}
/* stratified_indices: smc.rs:275-292 */
void orc_stratified_indices(const double *w, int64_t n, const double *U, int64_t *idx) {
    double cum = 0.0; int64_t i = 0;
    for (int64_t j = 0; j < n; j++) {
        double thr = ((double)j + U[j]) / (double)n;
        while (cum < thr && i < n) { cum += w[i]; i++; }
        int64_t k = i - 1; if (k < 0) k = 0;
        idx[j] = k < n - 1 ? k : n - 1;
    }
}
/* multinomial_indices: smc.rs:295-314 */
void orc_multinomial_indices(const double *w, int64_t n, const double *U, int64_t *idx) {
    for (int64_t j = 0; j < n; j++) {
        double u = U[j], cum = 0.0; int64_t sel = n - 1;
        for (int64_t i = 0; i < n; i++) { cum += w[i]; if (u <= cum) { sel = i; break; } }
        idx[j] = sel;
    }
}
/* next_beta: smc.rs:588-622 */
static double ess_at(double b, double beta, const double *lw, const double *ll, int64_t n, double *tmp) {
    for (int64_t i = 0; i < n; i++) tmp[i] = lw[i] + (b - beta) * ll[i];
    double lse1 = orc_log_sum_exp(tmp, (size_t)n);
    for (int64_t i = 0; i < n; i++) tmp[i] = 2.0 * tmp[i];
    double lse2 = orc_log_sum_exp(tmp, (size_t)n);
    if (!isfinite(lse1) || !isfinite(lse2)) return (double)n;
    return exp(2.0 * lse1 - lse2);
}
double orc_next_beta(double beta, const double *lw, const double *ll, int64_t n, double target) {
    double *tmp = (double *)malloc((size_t)(n + 1) * sizeof(double));
    double out;
    if (ess_at(1.0, beta, lw, ll, n, tmp) >= target) out = 1.0;
    else {
        double lo = beta, hi = 1.0;
        for (int it = 0; it < 64; it++) {
            double mid = 0.5 * (lo + hi);
            if (ess_at(mid, beta, lw, ll, n, tmp) < target) hi = mid; else lo = mid;
        }
        out = fmin(fmax(hi, beta + 1e-9), 1.0);
    }
    free(tmp);
    return out;
}

/* tempered_single_site_mh: smc.rs:631-688.  Returns accept flag; updates the
 * particle's values and its (log_prior, loglik). */
static int tempered_move(const orc_model *m, orc_stream *st, orc_cell *vals, double *lprior, double *ll,
                         double beta, double scale, int *site_out, int64_t *n_evals) {
    int d = m->n_f64, S = m->n_samples;
    if (d == 0) { *site_out = -1; return 0; }
    uint64_t ra, rb; orc_stream_block(st, &ra, &rb);
    int k = (int)(((unsigned __int128)ra * (uint64_t)d) >> 64);
    int site = m->f64_sites[k];
    *site_out = site;
    double z = orc_stream_normal(st);                       /* Normal(0,1).sample: smc.rs:655 */
    orc_cell *pv = (orc_cell *)alloca((size_t)(S + 1) * sizeof(orc_cell));
    memcpy(pv, vals, (size_t)S * sizeof(orc_cell));
    pv[site].f = vals[site].f + scale * z;
    double ca[3], pa[3];
    orc_run_score(m, vals, ca, NULL);                       /* two model runs :662-675 */
    orc_run_score(m, pv, pa, NULL);
    *n_evals += 2;
    double log_alpha = (pa[0] - ca[0]) + beta * ((pa[1] + pa[2]) - (ca[1] + ca[2]));
    int accept = (log_alpha >= 0.0) || (orc_stream_u01(st) < exp(log_alpha));
    if (accept) { memcpy(vals, pv, (size_t)S * sizeof(orc_cell)); *lprior = pa[0]; *ll = pa[1] + pa[2]; }
    else { *lprior = ca[0]; *ll = ca[1] + ca[2]; }
    return accept;
}

/* Batched form of DiminishingAdaptation::update for a sweep in which a site
 * received `n` proposals of which `a` were accepted, all made at the same
 * scale: the n per-move steps t^-gamma (t = T+1..T+n) are summed and applied
 * once with the post-sweep cumulative acceptance rate.  (GPU semantics; the
 * reference's strictly sequential shared update, smc.rs:482,544-553, cannot
 * be parallelised -- documented deviation.) */
static void adapt_update_batched(double *scale, double *log_scale, int64_t *acc, int64_t *tot, int64_t n,
                                 int64_t a, double target, double gamma) {
    if (n <= 0) return;
    int64_t T0 = *tot;
    *tot += n; *acc += a;
    if (*tot < 10) return;
    double rate = (double)*acc / (double)*tot;
    int64_t lo = T0 + 1 < 10 ? 10 : T0 + 1;               /* steps with total < 10 are skipped */
    double step = 0.0;
    if (*tot - lo + 1 <= 64) { for (int64_t t = lo; t <= *tot; t++) step += 1.0 / pow((double)t, gamma); }
    else {
        /* Euler-Maclaurin: sum_{t=lo}^{hi} t^-g ~ integral + endpoint average */
        double hi = (double)*tot, l = (double)lo, e = 1.0 - gamma;
        step = (pow(hi, e) - pow(l, e)) / e + 0.5 * (pow(l, -gamma) + pow(hi, -gamma));
    }
    *log_scale += step * (rate - target);
    double ns = exp(*log_scale);
    *scale = (isfinite(ns) && ns > 0.0) ? fmin(fmax(ns, 0.001), 100.0) : 1.0;
    if (*scale == 1.0) *log_scale = 0.0; else *log_scale = log(*scale);
}

/* adaptive_smc: smc.rs:455-581 */
int orc_smc_run(const orc_model *m, int64_t n, const orc_smc_config *cfg, uint64_t seed, orc_cell *values,
                double *log_w, double *weights, double *log_evidence, double *betas, int max_betas,
                int64_t *n_model_evals) {
    int S = m->n_samples;
    int64_t evals = 0;
    int n_steps = 0;
    *log_evidence = 0.0;
    if (n == 0) return 0;
    double nf = (double)n;
    orc_cell *pv = (orc_cell *)calloc((size_t)n * (S > 0 ? S : 1), sizeof(orc_cell));   /* AoS [particle][site] */
    orc_cell *pv2 = (orc_cell *)calloc((size_t)n * (S > 0 ? S : 1), sizeof(orc_cell));
    double *ll = (double *)malloc((size_t)n * sizeof(double));
    double *lprior = (double *)malloc((size_t)n * sizeof(double));
    double *comb = (double *)malloc((size_t)n * sizeof(double));
    double *w = (double *)malloc((size_t)n * sizeof(double));
    int64_t *idx = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    double *scale = (double *)malloc(((size_t)S + 1) * sizeof(double));
    double *lscale = (double *)calloc((size_t)S + 1, sizeof(double));
    int64_t *acc_n = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    int64_t *tot_n = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    int64_t *sw_n = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    int64_t *sw_a = (int64_t *)calloc((size_t)S + 1, sizeof(int64_t));
    for (int j = 0; j < S; j++) scale[j] = 1.0;

    /* smc_prior_particles: smc.rs:764-790 */
    for (int64_t i = 0; i < n; i++) {
        orc_stream st; double acc[3];
        orc_stream_init(&st, seed, (uint32_t)i, 0, ORC_RNG_SMC_PRIOR);
        orc_run_prior(m, &st, pv + (size_t)i * S, acc, NULL);
        lprior[i] = acc[0]; ll[i] = acc[1] + acc[2];       /* particle_log_likelihood :381-383 */
        evals++;
    }
    for (int64_t i = 0; i < n; i++) log_w[i] = -log(nf);
    double beta = 0.0;
    double target_ess = fmin(fmax(cfg->ess_threshold * nf, 1.0), nf);

    if (cfg->rejuvenation_steps == 0) {                     /* :484-493 */
        for (int64_t i = 0; i < n; i++) comb[i] = -log(nf) + ll[i];
        *log_evidence = orc_log_sum_exp(comb, (size_t)n);
        beta = 1.0;
        memcpy(log_w, comb, (size_t)n * sizeof(double));
        if (betas && max_betas > 0) betas[0] = 1.0;
        n_steps = 1;
    } else {
        int steps = 0;
        while (beta < 1.0) {                                /* :501-560 */
            steps += 1;
            double beta_new = orc_next_beta(beta, log_w, ll, n, target_ess);
            if (steps >= 10000) beta_new = 1.0;
            double d_beta = beta_new - beta;
            for (int64_t i = 0; i < n; i++) comb[i] = log_w[i] + d_beta * ll[i];
            double log_norm = orc_log_sum_exp(comb, (size_t)n);
            *log_evidence += log_norm;
            if (isfinite(log_norm)) for (int64_t i = 0; i < n; i++) log_w[i] = comb[i] - log_norm;
            else for (int64_t i = 0; i < n; i++) log_w[i] = -log(nf);
            beta = beta_new;
            if (betas && n_steps < max_betas) betas[n_steps] = beta;
            n_steps++;
            if (beta < 1.0) {
                for (int64_t i = 0; i < n; i++) w[i] = exp(log_w[i]);
                orc_stream st;
                if (cfg->resampling_method == ORC_RESAMPLE_SYSTEMATIC) {
                    orc_stream_init(&st, seed, 0, (uint32_t)steps, ORC_RNG_SMC_RESAMPLE);
                    orc_systematic_indices(w, n, orc_stream_u01(&st), idx);
                } else {
                    double *U = (double *)malloc((size_t)n * sizeof(double));
                    for (int64_t j = 0; j < n; j++) {
                        orc_stream_init(&st, seed, (uint32_t)j, (uint32_t)steps, ORC_RNG_SMC_RESAMPLE);
                        U[j] = orc_stream_u01(&st);
                    }
                    if (cfg->resampling_method == ORC_RESAMPLE_STRATIFIED) orc_stratified_indices(w, n, U, idx);
                    else orc_multinomial_indices(w, n, U, idx);
                    free(U);
                }
                for (int64_t i = 0; i < n; i++) memcpy(pv2 + (size_t)i * S, pv + (size_t)idx[i] * S, (size_t)S * sizeof(orc_cell));
                { orc_cell *t = pv; pv = pv2; pv2 = t; }
                for (int64_t i = 0; i < n; i++) { comb[i] = ll[idx[i]]; }
                memcpy(ll, comb, (size_t)n * sizeof(double));
                for (int64_t i = 0; i < n; i++) log_w[i] = -log(nf);
                int R = cfg->rejuvenation_steps;
                if (!cfg->batched_adaptation) {
                    /* reference order: particle-major, ONE shared adaptation
                     * updated after every move (smc.rs:482,544-553) */
                    for (int64_t i = 0; i < n; i++)
                        for (int r = 0; r < R; r++) {
                            int site;
                            orc_stream_init(&st, seed, (uint32_t)i, (uint32_t)((steps - 1) * R + r), ORC_RNG_SMC_REJUV);
                            /* scale must be read after the site is picked: peek the pick */
                            orc_stream pk = st; uint64_t ra, rb; orc_stream_block(&pk, &ra, &rb);
                            int kk = m->n_f64 ? (int)(((unsigned __int128)ra * (uint64_t)m->n_f64) >> 64) : 0;
                            double sc = m->n_f64 ? scale[m->f64_sites[kk]] : 1.0;
                            int acc = tempered_move(m, &st, pv + (size_t)i * S, &lprior[i], &ll[i], beta, sc, &site, &evals);
                            if (site >= 0) orc_adapt_update(&scale[site], &lscale[site], &acc_n[site], &tot_n[site], acc, 0.44, 0.7);
                        }
                } else {
                    /* sweep-major with a per-sweep batched update */
                    for (int r = 0; r < R; r++) {
                        for (int j = 0; j < S; j++) { sw_n[j] = 0; sw_a[j] = 0; }
                        for (int64_t i = 0; i < n; i++) {
                            int site;
                            orc_stream_init(&st, seed, (uint32_t)i, (uint32_t)((steps - 1) * R + r), ORC_RNG_SMC_REJUV);
                            orc_stream pk = st; uint64_t ra, rb; orc_stream_block(&pk, &ra, &rb);
                            int kk = m->n_f64 ? (int)(((unsigned __int128)ra * (uint64_t)m->n_f64) >> 64) : 0;
                            double sc = m->n_f64 ? scale[m->f64_sites[kk]] : 1.0;
                            int acc = tempered_move(m, &st, pv + (size_t)i * S, &lprior[i], &ll[i], beta, sc, &site, &evals);
                            if (site >= 0) { sw_n[site]++; sw_a[site] += acc; }
                        }
                        for (int j = 0; j < S; j++)
                            adapt_update_batched(&scale[j], &lscale[j], &acc_n[j], &tot_n[j], sw_n[j], sw_a[j], 0.44, 0.7);
                    }
                }
            }
        }
    }
    /* final normalisation: smc.rs:565-575 */
    double log_norm = orc_log_sum_exp(log_w, (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        if (isfinite(log_norm)) { double nz = log_w[i] - log_norm; log_w[i] = nz; if (weights) weights[i] = exp(nz); }
        else { log_w[i] = -log(nf); if (weights) weights[i] = 1.0 / nf; }
    }
    if (values) for (int64_t i = 0; i < n; i++) for (int j = 0; j < S; j++) values[(size_t)j * n + i] = pv[(size_t)i * S + j];
    if (n_model_evals) *n_model_evals = evals;
    free(pv); free(pv2); free(ll); free(lprior); free(comb); free(w); free(idx);
    free(scale); free(lscale); free(acc_n); free(tot_n); free(sw_n); free(sw_a);
    return n_steps;
}

/* ===================================================================== */
/* Diagnostics                                                            */
/* ===================================================================== */

/* r_hat_from_f64_chains: diagnostics.rs:262-304; chains[m][n] row-major with
 * row stride `stride`, using columns [off, off+n). */
static double rhat_core(const double *ch, int m, int n, int stride, const int *offs) {
    if (m < 2) return 1.0;
    if (n == 0) return NAN;
    double mf = (double)m, nf = (double)n;
    double *means = (double *)alloca((size_t)m * sizeof(double));
    for (int j = 0; j < m; j++) {
        const double *x = ch + (size_t)(j / (offs ? 2 : 1)) * stride + (offs ? offs[j % 2] : 0);
        double s = 0.0; for (int i = 0; i < n; i++) s += x[i];
        means[j] = s / nf;
    }
    double overall = 0.0; for (int j = 0; j < m; j++) overall += means[j];
    overall /= mf;
    double bs = 0.0; for (int j = 0; j < m; j++) bs += (means[j] - overall) * (means[j] - overall);
    double b = nf / (mf - 1.0) * bs;
    double wsum = 0.0;
    for (int j = 0; j < m; j++) {
        const double *x = ch + (size_t)(j / (offs ? 2 : 1)) * stride + (offs ? offs[j % 2] : 0);
        double s = 0.0; for (int i = 0; i < n; i++) s += (x[i] - means[j]) * (x[i] - means[j]);
        wsum += s / (nf - 1.0);
    }
    double w = wsum / mf;
    double var_plus = ((nf - 1.0) / nf) * w + (1.0 / nf) * b;
    return sqrt(var_plus / w);
}
double orc_classic_rhat(const double *chains, int m, int n) { return rhat_core(chains, m, n, n, NULL); }
/* split_f64_chains: diagnostics.rs:240-253 (halve, drop the middle draw if odd) */
double orc_split_rhat(const double *chains, int m, int n) {
    int half = n / 2;
    if (half == 0) return rhat_core(chains, m, n, n, NULL);
    int offs[2] = { 0, half };
    return rhat_core(chains, 2 * m, half, n, offs);
}

/* autocovariances: mcmc_utils.rs:231-244 */
static void autocov(const double *x, int n, int max_lag, double *acov) {
    double mean = 0.0; for (int i = 0; i < n; i++) mean += x[i];
    mean /= (double)n;
    double *c = (double *)malloc((size_t)n * sizeof(double));
    for (int i = 0; i < n; i++) c[i] = x[i] - mean;
    for (int lag = 0; lag <= max_lag; lag++) {
        double s = 0.0;
        for (int i = 0; i < n - lag; i++) s += c[i] * c[i + lag];
        acov[lag] = s / (double)n;
    }
    free(c);
}
/* ess_from_chains: mcmc_utils.rs:253-339 */
double orc_ess_multichain(const double *chains, int m, int n) {
    if (m == 0) return 0.0;
    if (n < 4) { int64_t t = (int64_t)m * n; return (double)(t > 1 ? t : 1); }
    int max_lag = (n - 1) < 2048 ? (n - 1) : 2048;
    double *acovs = (double *)malloc((size_t)m * (max_lag + 1) * sizeof(double));
    double nf = (double)n, mf = (double)m;
    double *means = (double *)malloc((size_t)m * sizeof(double));
    double mean_var = 0.0;
    for (int j = 0; j < m; j++) {
        const double *x = chains + (size_t)j * n;
        autocov(x, n, max_lag, acovs + (size_t)j * (max_lag + 1));
        double s = 0.0; for (int i = 0; i < n; i++) s += x[i];
        means[j] = s / nf;
        mean_var += acovs[(size_t)j * (max_lag + 1)] * nf / (nf - 1.0);
    }
    mean_var /= mf;
    double out;
    if (mean_var <= 0.0) { out = (double)((int64_t)m * n); goto done; }
    {
        double var_plus = mean_var * (nf - 1.0) / nf;
        if (m > 1) {
            double overall = 0.0; for (int j = 0; j < m; j++) overall += means[j];
            overall /= mf;
            double bt = 0.0; for (int j = 0; j < m; j++) bt += (means[j] - overall) * (means[j] - overall);
            var_plus += bt / (mf - 1.0);
        }
#define RHO(T, OUT) do { double a_ = 0.0; for (int j_ = 0; j_ < m; j_++) a_ += acovs[(size_t)j_ * (max_lag + 1) + (T)]; \
                         a_ /= mf; (OUT) = 1.0 - (mean_var - a_) / var_plus; } while (0)
        double *rho_hat = (double *)calloc((size_t)max_lag + 2, sizeof(double));
        rho_hat[0] = 1.0;
        if (max_lag >= 1) RHO(1, rho_hat[1]);
        int t = 1, max_t = 1 < max_lag ? 1 : max_lag;
        while (t + 2 <= max_lag) {
            double re, ro; RHO(t + 1, re); RHO(t + 2, ro);
            if (re + ro < 0.0) break;
            rho_hat[t + 1] = re; rho_hat[t + 2] = ro; max_t = t + 2; t += 2;
        }
        int k = 1;
        while (k + 2 <= max_t) {
            double prev = rho_hat[k - 1] + rho_hat[k], cur = rho_hat[k + 1] + rho_hat[k + 2];
            if (cur > prev) { double avg = prev / 2.0; rho_hat[k + 1] = avg; rho_hat[k + 2] = avg; }
            k += 2;
        }
        double sum_rho = 0.0; for (int i = 0; i <= max_t; i++) sum_rho += rho_hat[i];
        double tau = fmax(-1.0 + 2.0 * sum_rho, 1.0);
        out = (double)((int64_t)m * n) / tau;
        free(rho_hat);
#undef RHO
    }
done:
    free(acovs); free(means);
    return out;
}
/* effective_sample_size_mcmc: mcmc_utils.rs:195-201 */
double orc_ess_single(const double *x, int n) {
    if (n < 4) return (double)n;
    return orc_ess_multichain(x, 1, n);
}
/* spectral_variance_of_mean: mcmc_utils.rs:393-421 */
static double spectral_var_mean(const double *seg, int n) {
    if (n < 2) return 0.0;
    double mean = 0.0; for (int i = 0; i < n; i++) mean += seg[i];
    mean /= (double)n;
    double s2 = 0.0; for (int i = 0; i < n; i++) s2 += (seg[i] - mean) * (seg[i] - mean);
    s2 /= ((double)n - 1.0);
    if (s2 == 0.0) return 0.0;
    int max_lag = (n - 1) < 1024 ? (n - 1) : 1024;
    double *acov = (double *)malloc(((size_t)max_lag + 1) * sizeof(double));
    autocov(seg, n, max_lag, acov);
    double var0 = acov[0], tau = 1.0;
    if (var0 <= 0.0) { free(acov); return 0.0; }
    for (int k = 1; k <= max_lag; k++) { double r = acov[k] / var0; if (r <= 0.0) break; tau += 2.0 * r; }
    free(acov);
    return s2 * tau / (double)n;
}
/* geweke_diagnostic: mcmc_utils.rs:354-384 */
double orc_geweke(const double *x, int n) {
    if (n < 20) return NAN;
    int first_end = n / 10, last_start = n / 2;
    int n1 = first_end, n2 = n - last_start;
    if (n1 < 2 || n2 < 2) return NAN;
    double m1 = 0.0, m2 = 0.0;
    for (int i = 0; i < n1; i++) m1 += x[i];
    m1 /= (double)n1;
    for (int i = last_start; i < n; i++) m2 += x[i];
    m2 /= (double)n2;
    double se = sqrt(spectral_var_mean(x, n1) + spectral_var_mean(x + last_start, n2));
    if (se == 0.0) return 0.0;
    return (m1 - m2) / se;
}
static int cmp_d(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }
/* summarize_f64_parameter: diagnostics.rs:331-391 */
void orc_summarize(const double *chains, int m, int n, double out[9]) {
    size_t len = (size_t)m * n;
    if (len == 0) { for (int i = 0; i < 8; i++) out[i] = NAN; out[8] = 0.0; return; }
    double mean = 0.0; for (size_t i = 0; i < len; i++) mean += chains[i];
    mean /= (double)len;
    double var = 0.0; for (size_t i = 0; i < len; i++) var += (chains[i] - mean) * (chains[i] - mean);
    var /= (double)(len - 1);
    out[0] = mean; out[1] = sqrt(var);
    double *s = (double *)malloc(len * sizeof(double));
    memcpy(s, chains, len * sizeof(double));
    qsort(s, len, sizeof(double), cmp_d);
    const double ps[5] = { 0.025, 0.25, 0.5, 0.75, 0.975 };
    for (int k = 0; k < 5; k++) out[2 + k] = s[(size_t)round((double)(len - 1) * ps[k])];
    free(s);
    out[7] = orc_split_rhat(chains, m, n);
    out[8] = orc_ess_multichain(chains, m, n);
}
