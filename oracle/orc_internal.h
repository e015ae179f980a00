/* orc_internal.h -- CPU ORACLE internals (test infrastructure, not the product). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "fugue_oracle.h"

#define ORC_MAX_PARAMS 64   /* Categorical takes up to 64 probabilities (dsl.rs:817) */

typedef struct { int op, a, b, c; double v; } orc_node;

typedef struct {
    int kind, dist, vtype;
    char *addr;
    int *params; int nparams;
    int value;      /* node id (observe value / factor log-weight) */
    int handle;     /* program-order sample index, -1 otherwise */
    int sorted;     /* address-sorted site index (samples only) */
} orc_stmt;

struct orc_model {
    double **data; int *data_len; int n_data, cap_data;
    orc_node *nodes; int n_nodes, cap_nodes;
    int *args; int n_args, cap_args;
    orc_stmt *stmts; int n_stmts, cap_stmts;
    int n_samples, n_observes, n_f64;
    int *handle_to_sorted;  /* [n_samples] */
    int *sorted_stmt;       /* [n_samples] sorted idx -> stmt idx */
    int *f64_sites;         /* [n_f64] sorted idx of each f64 site, ascending */
    int finalized;
};

double orc_eval(const orc_model *m, int id, const orc_cell *vals);
int    orc_stmt_params(const orc_model *m, const orc_stmt *s, const orc_cell *vals, double *p);
double orc_logpdf_du(int64_t x, int64_t lo, int64_t hi);

#endif
