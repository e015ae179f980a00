// fg_program.h -- internal definition of the opaque fg_program handle.
#pragma once
#include <string>
#include <vector>

#include "../../include/fugue_amd.h"
#include "fg_math.h"

struct FgNode {
    int op = 0, a = 0, b = 0;
    bool is_const = false;
    double cval = 0.0;
    std::vector<int> kids;
};
struct FgStmt {
    int kind = 0;            // 0 sample, 1 observe, 2 factor
    int dist = -1, vtype = 0;
    std::string addr;
    std::vector<int> params; // expression roots
    int value = -1;          // observe value / factor log-weight root
    int handle = -1, sorted = -1;
    bool exact_bounds = false; long long lo = 0, hi = 0;   // DiscreteUniform with exact i64 bounds
};
struct fg_program {
    std::vector<std::vector<double>> data;
    std::vector<std::string> data_names;
    std::vector<FgNode> nodes;
    std::vector<FgStmt> stmts;
    int n_samples = 0, n_observes = 0;
    // compiled
    bool finalized = false;
    std::vector<int> sorted_stmt, handle_to_sorted, site_vtype, f64_slot /* sorted site index of coordinate k */, site_slot /* LDS slot of site j */, sub_off;
    std::vector<FgIns> ins, ins_fast, sub;
    std::vector<FgCoord> coord;
    std::vector<FgGradRec> gstream;   // empty unless every sub-program is all-fast
    int n_gstream = 0;
    std::vector<FgGradRec> sstream;   // empty unless the whole program is fast Normals
    int n_sstream = 0;
    bool sstream_has_lin = false;       // some record is a linear-predictor Normal (FG_G_LIN)
    bool sstream_has_genrec = false;    // ... a FG_G_GEN record proper (sstream_has_gen also counts option selects / Categorical tables)
    bool sstream_has_gen = false;       // some record is a general distribution record (FG_G_GEN)
    std::vector<FgSepRec> sep;        // empty unless the program is an independent-sites model (fg_ir.h)
    std::vector<FgSepCoord> sep_coord;
    std::vector<FgSepFree> sep_free; int n_prior_terms = 0;
    std::vector<uint32_t> sobs;       // observe bits of the score stream
    std::vector<int> site_rec;        // [S] score-stream record of each site's sample statement
    // dense regressions (fg_hmc_lin.hip): every coordinate's force terms are its own prior record(s) plus the SAME lin_n linear-predictor
    // observe statements, each reading all d coordinates once in one common term order
    std::vector<double> lin_tab;      // [lin_n + 1] rows of FgLinRow doubles (fg_ir.h); empty when the program is not of that shape
    std::vector<int> lin_meta;        // [dp] coordinate at term position t (dp = d padded to 8 / 16 / 32: the always-zero slot beyond d), then [d][2] {first prior record in gstream, count}
    int lin_n = 0, lin_p2 = 0;        // observations; every observe sigma is a power of two
    std::vector<double> pool;
    int n_slots = 0, n_ins = 0;
    std::vector<int> site_cat;               // [S][2] {pool base, K} of Categorical sites with a valid constant table, else -1
    std::vector<std::string> dsl_warnings;   // fg_dsl.cpp

    int  parse(const fg_tok *toks, int n);
    void collect_sites(int node, std::vector<int> &out) const;
    void compile_stmt(const FgStmt &s, std::vector<FgIns> &out, int &temp_max);
    int  finalize();
};
void fg_set_error(const std::string &s);
bool fg_categorical_const_valid(const std::vector<double> &p);
