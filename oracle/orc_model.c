/*
 * orc_model.c -- CPU ORACLE (test infrastructure, not the product):
 * model description (expression trees + statements) and the two model runs
 * the hot path needs: PriorHandler and ScoreGivenTrace semantics
 * (src/runtime/interpreters.rs:76-163, src/runtime/handler.rs:124-209).
 *
 * Deliberately a different shape from the product's flattened site program:
 * a tree-walking evaluator over per-chain AoS cells, one chain at a time.
 */
#define _GNU_SOURCE
#include "orc_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

orc_model *orc_model_new(void) { return (orc_model *)calloc(1, sizeof(orc_model)); }

void orc_model_free(orc_model *m) {
    if (!m) return;
    for (int i = 0; i < m->n_data; i++) free(m->data[i]);
    for (int i = 0; i < m->n_stmts; i++) { free(m->stmts[i].addr); free(m->stmts[i].params); }
    free(m->data); free(m->data_len); free(m->nodes); free(m->args); free(m->stmts);
    free(m->handle_to_sorted); free(m->sorted_stmt); free(m->f64_sites);
    free(m);
}

#define GROW(ptr, n, cap, T) do { if ((n) >= (cap)) { (cap) = (cap) ? 2 * (cap) : 16; \
        (ptr) = (T *)realloc((ptr), (size_t)(cap) * sizeof(T)); } } while (0)

int orc_model_add_data(orc_model *m, const double *v, int n) {
    GROW(m->data, m->n_data, m->cap_data, double *);
    m->data_len = (int *)realloc(m->data_len, (size_t)m->cap_data * sizeof(int));
    m->data[m->n_data] = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    memcpy(m->data[m->n_data], v, (size_t)n * sizeof(double));
    m->data_len[m->n_data] = n;
    return m->n_data++;
}
int orc_model_add_node(orc_model *m, int op, int a, int b, int c, double v) {
    GROW(m->nodes, m->n_nodes, m->cap_nodes, orc_node);
    orc_node nd = { op, a, b, c, v };
    m->nodes[m->n_nodes] = nd;
    return m->n_nodes++;
}
int orc_model_add_args(orc_model *m, const int *ids, int n) {
    int start = m->n_args;
    for (int i = 0; i < n; i++) { GROW(m->args, m->n_args, m->cap_args, int); m->args[m->n_args++] = ids[i]; }
    return start;
}
static int vtype_of_dist(int dist) {
    switch (dist) {
    case ORC_BERNOULLI: return ORC_BOOL;
    case ORC_CATEGORICAL: return ORC_USIZE;
    case ORC_BINOMIAL: case ORC_POISSON: return ORC_U64;
    case ORC_DISCRETEUNIFORM: return ORC_I64;
    default: return ORC_F64;
    }
}
int orc_model_add_stmt(orc_model *m, int kind, int dist, const char *addr,
                       const int *params, int nparams, int value) {
    GROW(m->stmts, m->n_stmts, m->cap_stmts, orc_stmt);
    orc_stmt *s = &m->stmts[m->n_stmts];
    memset(s, 0, sizeof(*s));
    s->kind = kind; s->dist = dist; s->value = value; s->nparams = nparams;
    s->addr = addr ? strdup(addr) : NULL;
    s->params = (int *)malloc((size_t)(nparams > 0 ? nparams : 1) * sizeof(int));
    for (int i = 0; i < nparams; i++) s->params[i] = params[i];
    s->vtype = (kind == ORC_STMT_FACTOR) ? ORC_F64 : vtype_of_dist(dist);
    s->handle = -1;
    int ret = m->n_stmts;
    if (kind == ORC_STMT_SAMPLE) { s->handle = m->n_samples; ret = m->n_samples++; }
    if (kind == ORC_STMT_OBSERVE) m->n_observes++;
    m->n_stmts++;
    m->finalized = 0;
    return ret;
}

static const orc_model *g_sort_model;
static int cmp_addr(const void *a, const void *b) {
    /* Address Ord = byte-wise lexicographic on the backing string
     * (src/core/address.rs:150-157); strcmp is the same order. */
    int ia = *(const int *)a, ib = *(const int *)b;
    return strcmp(g_sort_model->stmts[ia].addr, g_sort_model->stmts[ib].addr);
}
int orc_model_finalize(orc_model *m) {
    int S = m->n_samples;
    free(m->handle_to_sorted); free(m->sorted_stmt); free(m->f64_sites);
    m->handle_to_sorted = (int *)malloc((size_t)(S + 1) * sizeof(int));
    m->sorted_stmt = (int *)malloc((size_t)(S + 1) * sizeof(int));
    m->f64_sites = (int *)malloc((size_t)(S + 1) * sizeof(int));
    int k = 0;
    for (int i = 0; i < m->n_stmts; i++) if (m->stmts[i].kind == ORC_STMT_SAMPLE) m->sorted_stmt[k++] = i;
    g_sort_model = m;
    qsort(m->sorted_stmt, (size_t)S, sizeof(int), cmp_addr);
    for (int j = 0; j + 1 < S; j++)
        if (strcmp(m->stmts[m->sorted_stmt[j]].addr, m->stmts[m->sorted_stmt[j + 1]].addr) == 0)
            return 301;   /* ErrorCode::AddressConflict, src/error.rs:51 */
    m->n_f64 = 0;
    for (int j = 0; j < S; j++) {
        orc_stmt *s = &m->stmts[m->sorted_stmt[j]];
        s->sorted = j;
        m->handle_to_sorted[s->handle] = j;
        if (s->vtype == ORC_F64) m->f64_sites[m->n_f64++] = j;
    }
    m->finalized = 1;
    return 0;
}
int orc_model_n_sites(const orc_model *m) { return m->n_samples; }
int orc_model_n_f64(const orc_model *m) { return m->n_f64; }
int orc_model_n_observe(const orc_model *m) { return m->n_observes; }
const char *orc_model_site_name(const orc_model *m, int j) { return m->stmts[m->sorted_stmt[j]].addr; }
int orc_model_site_vtype(const orc_model *m, int j) { return m->stmts[m->sorted_stmt[j]].vtype; }
int orc_model_site_of_handle(const orc_model *m, int h) { return m->handle_to_sorted[h]; }
int orc_model_f64_site(const orc_model *m, int k) { return m->f64_sites[k]; }

/* Rust f64::clamp: NaN stays NaN */
static double clampd(double x, double lo, double hi) { if (x < lo) return lo; if (x > hi) return hi; return x; }

/* Tree-walking expression evaluator.  A site reference yields the site's
 * value as f64 (`Value::as_f64`, crates/fugue-wasm/src/dsl.rs:53-66). */
double orc_eval(const orc_model *m, int id, const orc_cell *vals) {
    const orc_node *n = &m->nodes[id];
    switch (n->op) {
    case ORC_X_CONST: return n->v;
    case ORC_X_SITE: {
        int j = m->handle_to_sorted[n->a];
        int vt = m->stmts[m->sorted_stmt[j]].vtype;
        return vt == ORC_F64 ? vals[j].f : (double)vals[j].i;
    }
    case ORC_X_DATA:
        if (n->b < 0 || n->b >= m->data_len[n->a]) return NAN;
        return m->data[n->a][n->b];
    case ORC_X_NEG:  return -orc_eval(m, n->a, vals);
    case ORC_X_ADD:  return orc_eval(m, n->a, vals) + orc_eval(m, n->b, vals);
    case ORC_X_SUB:  return orc_eval(m, n->a, vals) - orc_eval(m, n->b, vals);
    case ORC_X_MUL:  return orc_eval(m, n->a, vals) * orc_eval(m, n->b, vals);
    case ORC_X_DIV:  return orc_eval(m, n->a, vals) / orc_eval(m, n->b, vals);
    case ORC_X_EXP:  return exp(orc_eval(m, n->a, vals));
    case ORC_X_LN:   return log(orc_eval(m, n->a, vals));
    case ORC_X_SQRT: return sqrt(orc_eval(m, n->a, vals));
    case ORC_X_ABS:  return fabs(orc_eval(m, n->a, vals));
    case ORC_X_FLOOR:return floor(orc_eval(m, n->a, vals));
    case ORC_X_SIN:  return sin(orc_eval(m, n->a, vals));
    case ORC_X_COS:  return cos(orc_eval(m, n->a, vals));
    case ORC_X_TANH: return tanh(orc_eval(m, n->a, vals));
    case ORC_X_POW:  return pow(orc_eval(m, n->a, vals), orc_eval(m, n->b, vals));
    case ORC_X_MIN:  return fmin(orc_eval(m, n->a, vals), orc_eval(m, n->b, vals));
    case ORC_X_MAX:  return fmax(orc_eval(m, n->a, vals), orc_eval(m, n->b, vals));
    case ORC_X_CLAMP:return clampd(orc_eval(m, n->a, vals), orc_eval(m, n->b, vals), orc_eval(m, n->c, vals));
    case ORC_X_SELECT: {
        double di = orc_eval(m, n->a, vals);
        if (!(di >= 0.0) || di >= (double)n->c || di != floor(di)) return NAN;
        return orc_eval(m, m->args[n->b + (int)di], vals);
    }
    default: return NAN;
    }
}

int orc_stmt_params(const orc_model *m, const orc_stmt *s, const orc_cell *vals, double *p) {
    for (int i = 0; i < s->nparams; i++) p[i] = orc_eval(m, s->params[i], vals);
    return s->nparams;
}

static double stmt_logpdf(const orc_stmt *s, const double *p, int np, orc_cell x) {
    if (s->vtype == ORC_F64) return orc_logpdf(s->dist, 0, x.f, 0, p, np);
    return orc_logpdf(s->dist, 1, 0.0, x.i, p, np);
}

static orc_cell observe_value(const orc_model *m, const orc_stmt *s, const orc_cell *vals) {
    double v = orc_eval(m, s->value, vals);
    orc_cell c;
    switch (s->vtype) {
    case ORC_F64:  c.f = v; break;
    case ORC_BOOL: c.i = (v != 0.0); break;           /* Value::as_bool, dsl.rs:76-81 */
    default: c.i = isfinite(v) ? (int64_t)v : 0; break;
    }
    return c;
}

/* ScoreGivenTrace: interpreters.rs:138-163; observe :76-83; factor adds to
 * log_factors; total = prior + lik + factors (trace.rs:198-200). */
void orc_run_score(const orc_model *m, const orc_cell *vals, double acc[3], double *logp) {
    double pbuf[ORC_MAX_PARAMS];
    acc[0] = acc[1] = acc[2] = 0.0;
    for (int i = 0; i < m->n_stmts; i++) {
        const orc_stmt *s = &m->stmts[i];
        if (s->kind == ORC_STMT_FACTOR) { acc[2] += orc_eval(m, s->value, vals); continue; }
        int np = orc_stmt_params(m, s, vals, pbuf);
        if (s->kind == ORC_STMT_SAMPLE) {
            double lp = stmt_logpdf(s, pbuf, np, vals[s->sorted]);
            acc[0] += lp;
            if (logp) logp[s->sorted] = lp;
        } else {
            acc[1] += stmt_logpdf(s, pbuf, np, observe_value(m, s, vals));
        }
    }
}

/* PriorHandler: interpreters.rs:88-104 -- sample, score, record. */
void orc_run_prior(const orc_model *m, orc_stream *st, orc_cell *vals, double acc[3], double *logp) {
    double pbuf[ORC_MAX_PARAMS];
    acc[0] = acc[1] = acc[2] = 0.0;
    for (int i = 0; i < m->n_stmts; i++) {
        const orc_stmt *s = &m->stmts[i];
        if (s->kind == ORC_STMT_FACTOR) { acc[2] += orc_eval(m, s->value, vals); continue; }
        int np = orc_stmt_params(m, s, vals, pbuf);
        if (s->kind == ORC_STMT_SAMPLE) {
            orc_cell x = orc_sample_dist(s->dist, pbuf, np, st);
            vals[s->sorted] = x;
            double lp = stmt_logpdf(s, pbuf, np, x);
            acc[0] += lp;
            if (logp) logp[s->sorted] = lp;
        } else {
            acc[1] += stmt_logpdf(s, pbuf, np, observe_value(m, s, vals));
        }
    }
}
