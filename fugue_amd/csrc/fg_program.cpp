// fg_program.cpp -- host side of the boundary: builds a fixed-structure site program from
// sample/observe/factor calls (include/fugue_amd.h) and compiles it to the device IR
// (fg_ir.h).  Replaces, for the hot path, the reference's per-evaluation model rebuild +
// trampoline (src/core/model.rs:20-131, src/runtime/handler.rs:124-209): the program is
// flattened ONCE; every later "model run" is a pass of the interpreter kernels over it.
//
// Host-only code (no device calls): usable and tested without a GPU.
#include "fg_program.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_err;
void fg_set_error(const std::string &s) { g_err = s; }
extern "C" const char *fg_last_error(void) { return g_err.c_str(); }
extern "C" int fg_abi_version(void) { return FG_ABI_VERSION; }

namespace {

int vtype_of_dist(int dist) {
    switch (dist) {
    case FG_BERNOULLI: return FG_BOOL;
    case FG_CATEGORICAL: return FG_USIZE;
    case FG_BINOMIAL: case FG_POISSON: return FG_U64;
    case FG_DISCRETEUNIFORM: return FG_I64;
    default: return FG_F64;
    }
}
int n_params_of_dist(int dist) {
    switch (dist) {
    case FG_BERNOULLI: case FG_CHISQUARED: case FG_EXPONENTIAL: case FG_POISSON: return 1;
    case FG_STUDENTT: return 3;
    case FG_CATEGORICAL: return -1;
    default: return 2;
    }
}

double fold_unary(int op, double x) {
    switch (op) {
    case FG_T_NEG: return -x; case FG_T_EXP: return std::exp(x); case FG_T_LN: return std::log(x);
    case FG_T_SQRT: return std::sqrt(x); case FG_T_ABS: return std::fabs(x); case FG_T_FLOOR: return std::floor(x);
    case FG_T_SIN: return std::sin(x); case FG_T_COS: return std::cos(x); case FG_T_TANH: return std::tanh(x);
    }
    return NAN;
}
double fold_binary(int op, double x, double y) {
    switch (op) {
    case FG_T_ADD: return x + y; case FG_T_SUB: return x - y; case FG_T_MUL: return x * y; case FG_T_DIV: return x / y;
    case FG_T_POW: return std::pow(x, y); case FG_T_MIN: return std::fmin(x, y); case FG_T_MAX: return std::fmax(x, y);
    }
    return NAN;
}

}  // namespace

// ---- postfix tokens -> expression tree (constant subtrees folded with the host libm) ----
int fg_program::parse(const fg_tok *toks, int n) {
    std::vector<int> st;
    for (int i = 0; i < n; i++) {
        const fg_tok &t = toks[i];
        FgNode nd{};
        nd.op = t.op;
        switch (t.op) {
        case FG_T_CONST: nd.is_const = true; nd.cval = t.imm; break;
        case FG_T_SITE:
            if (t.a < 0 || t.a >= n_samples) { fg_set_error("expression references an unknown site handle"); return -1; }
            nd.a = t.a; break;
        case FG_T_DATA:
            if (t.a < 0 || t.a >= (int)data.size() || t.b < 0 || t.b >= (int)data[t.a].size()) {
                fg_set_error("data reference out of range"); return -1; }
            nd.op = FG_T_CONST; nd.is_const = true; nd.cval = data[t.a][t.b]; break;
        case FG_T_NEG: case FG_T_EXP: case FG_T_LN: case FG_T_SQRT: case FG_T_ABS: case FG_T_FLOOR:
        case FG_T_SIN: case FG_T_COS: case FG_T_TANH: {
            if (st.empty()) { fg_set_error("malformed expression (unary)"); return -1; }
            int a = st.back(); st.pop_back();
            nd.kids = {a};
            if (nodes[a].is_const) { nd.op = FG_T_CONST; nd.is_const = true; nd.cval = fold_unary(t.op, nodes[a].cval); nd.kids.clear(); }
            break; }
        case FG_T_ADD: case FG_T_SUB: case FG_T_MUL: case FG_T_DIV: case FG_T_POW: case FG_T_MIN: case FG_T_MAX: {
            if (st.size() < 2) { fg_set_error("malformed expression (binary)"); return -1; }
            int b = st.back(); st.pop_back(); int a = st.back(); st.pop_back();
            nd.kids = {a, b};
            if (nodes[a].is_const && nodes[b].is_const) {
                nd.op = FG_T_CONST; nd.is_const = true; nd.cval = fold_binary(t.op, nodes[a].cval, nodes[b].cval); nd.kids.clear(); }
            break; }
        case FG_T_CLAMP: {
            if (st.size() < 3) { fg_set_error("malformed expression (clamp)"); return -1; }
            int hi = st.back(); st.pop_back(); int lo = st.back(); st.pop_back(); int x = st.back(); st.pop_back();
            nd.kids = {x, lo, hi};
            if (nodes[x].is_const && nodes[lo].is_const && nodes[hi].is_const) {
                double v = nodes[x].cval, l = nodes[lo].cval, h = nodes[hi].cval;
                nd.op = FG_T_CONST; nd.is_const = true; nd.cval = v < l ? l : (v > h ? h : v); nd.kids.clear(); }
            break; }
        case FG_T_SELECT: {
            int k = t.a;
            if (k < 1 || (int)st.size() < k + 1) { fg_set_error("malformed expression (select)"); return -1; }
            std::vector<int> opts(st.end() - k, st.end());
            st.resize(st.size() - k);
            int idx = st.back(); st.pop_back();
            nd.kids.push_back(idx);
            nd.kids.insert(nd.kids.end(), opts.begin(), opts.end());
            if (nodes[idx].is_const) {
                double di = nodes[idx].cval;
                if (!(di >= 0.0) || di >= (double)k || di != std::floor(di)) { nd.op = FG_T_CONST; nd.is_const = true; nd.cval = NAN; nd.kids.clear(); }
                else { st.push_back(opts[(int)di]); continue; }
            }
            break; }
        default: fg_set_error("unknown expression token"); return -1;
        }
        nodes.push_back(nd);
        st.push_back((int)nodes.size() - 1);
    }
    if (st.size() != 1) { fg_set_error("malformed expression (stack)"); return -1; }
    return st[0];
}

void fg_program::collect_sites(int node, std::vector<int> &out) const {
    const FgNode &n = nodes[node];
    if (n.op == FG_T_SITE) out.push_back(n.a);
    for (int k : n.kids) collect_sites(k, out);
}

// ---- code generation: accumulator machine over the LDS slot file ----
struct FgGen {
    const fg_program &P;
    std::vector<FgIns> &out;
    int temp_next, temp_max;
    FgGen(const fg_program &p, std::vector<FgIns> &o) : P(p), out(o), temp_next(p.n_samples), temp_max(p.n_samples) {}

    int new_temp(int n = 1) { int t = temp_next; temp_next += n; temp_max = std::max(temp_max, temp_next); return t; }
    static FgIns blank(uint32_t op) { FgIns I; std::memset(&I, 0, sizeof(I)); I.op = op; return I; }
    bool is_leaf(int node) const { int op = P.nodes[node].op; return op == FG_T_CONST || op == FG_T_SITE; }
    // operand word (+ immediate) of a leaf
    void leaf_operand(int node, uint32_t &word, double &imm) const {
        const FgNode &n = P.nodes[node];
        if (n.op == FG_T_CONST) { word = FG_OPND(FG_OPND_IMM, 0); imm = n.cval; return; }
        const int site = P.handle_to_sorted[n.a];
        word = FG_OPND(P.site_vtype[site] == FG_F64 ? FG_OPND_SLOT_F : FG_OPND_SLOT_I, P.site_slot[site]);
        imm = 0.0;
    }
    void emit1(uint32_t op, int operand_node) {          // op with one leaf operand
        FgIns I = blank(op);
        leaf_operand(operand_node, I.opnd[0], I.imm[0]);
        out.push_back(I);
    }
    void emit_slot(uint32_t op, int slot) {               // op with a temp-slot operand
        FgIns I = blank(op);
        I.opnd[0] = FG_OPND(FG_OPND_SLOT_F, slot);
        out.push_back(I);
    }
    void store(int slot) { FgIns I = blank(FG_OP_STORE); I.aux = (uint32_t)slot; out.push_back(I); }

    // leaves the value of `node` in the accumulator
    void gen(int node) {
        const FgNode &n = P.nodes[node];
        switch (n.op) {
        case FG_T_CONST: case FG_T_SITE: emit1(FG_OP_LOAD, node); return;
        case FG_T_NEG: gen(n.kids[0]); out.push_back(blank(FG_OP_NEG)); return;
        case FG_T_EXP: gen(n.kids[0]); out.push_back(blank(FG_OP_EXP)); return;
        case FG_T_LN: gen(n.kids[0]); out.push_back(blank(FG_OP_LN)); return;
        case FG_T_SQRT: gen(n.kids[0]); out.push_back(blank(FG_OP_SQRT)); return;
        case FG_T_ABS: gen(n.kids[0]); out.push_back(blank(FG_OP_ABS)); return;
        case FG_T_FLOOR: gen(n.kids[0]); out.push_back(blank(FG_OP_FLOOR)); return;
        case FG_T_SIN: gen(n.kids[0]); out.push_back(blank(FG_OP_SIN)); return;
        case FG_T_COS: gen(n.kids[0]); out.push_back(blank(FG_OP_COS)); return;
        case FG_T_TANH: gen(n.kids[0]); out.push_back(blank(FG_OP_TANH)); return;
        case FG_T_ADD: case FG_T_SUB: case FG_T_MUL: case FG_T_DIV: case FG_T_POW: case FG_T_MIN: case FG_T_MAX: {
            int L = n.kids[0], R = n.kids[1];
            uint32_t fwd, rev; bool comm = false;
            switch (n.op) {
            case FG_T_ADD: fwd = rev = FG_OP_ADD; comm = true; break;
            case FG_T_MUL: fwd = rev = FG_OP_MUL; comm = true; break;
            case FG_T_MIN: fwd = rev = FG_OP_MIN; comm = true; break;
            case FG_T_MAX: fwd = rev = FG_OP_MAX; comm = true; break;
            case FG_T_SUB: fwd = FG_OP_SUB; rev = FG_OP_RSUB; break;
            case FG_T_DIV: fwd = FG_OP_DIV; rev = FG_OP_RDIV; break;
            default: fwd = FG_OP_POW; rev = FG_OP_RPOW; break;
            }
            (void)comm;
            // acc + a*b with leaf a, b: one MAC (two roundings, identical to the tree)
            if (n.op == FG_T_ADD && P.nodes[R].op == FG_T_MUL && is_leaf(P.nodes[R].kids[0]) && is_leaf(P.nodes[R].kids[1])) {
                gen(L);
                FgIns I = blank(FG_OP_MAC);
                leaf_operand(P.nodes[R].kids[0], I.opnd[0], I.imm[0]);
                leaf_operand(P.nodes[R].kids[1], I.opnd[1], I.imm[1]);
                out.push_back(I);
                return;
            }
            if (is_leaf(R)) { gen(L); emit1(fwd, R); return; }
            if (is_leaf(L)) { gen(R); emit1(rev, L); return; }
            gen(R);
            int t = new_temp();
            store(t);
            gen(L);
            emit_slot(fwd, t);
            return; }
        case FG_T_CLAMP: {
            int lo = n.kids[1], hi = n.kids[2];
            FgIns I = blank(FG_OP_CLAMP);
            if (is_leaf(lo)) leaf_operand(lo, I.opnd[0], I.imm[0]);
            else { gen(lo); int t = new_temp(); store(t); I.opnd[0] = FG_OPND(FG_OPND_SLOT_F, t); }
            if (is_leaf(hi)) leaf_operand(hi, I.opnd[1], I.imm[1]);
            else { gen(hi); int t = new_temp(); store(t); I.opnd[1] = FG_OPND(FG_OPND_SLOT_F, t); }
            gen(n.kids[0]);
            out.push_back(I);
            return; }
        case FG_T_SELECT: {
            int k = (int)n.kids.size() - 1;
            int base = new_temp(k);
            for (int i = 0; i < k; i++) { gen(n.kids[1 + i]); store(base + i); }
            gen(n.kids[0]);
            FgIns I = blank(FG_OP_GATHER);
            I.aux = (uint32_t)base; I.opnd[1] = (uint32_t)k;
            out.push_back(I);
            return; }
        }
    }
    // operand for a distribution parameter / observed value
    void operand(int node, uint32_t &word, double &imm) {
        if (is_leaf(node)) { leaf_operand(node, word, imm); return; }
        gen(node);
        int t = new_temp();
        store(t);
        word = FG_OPND(FG_OPND_SLOT_F, t); imm = 0.0;
    }
};

bool fg_categorical_const_valid(const std::vector<double> &p) {   // distribution.rs:679-715
    if (p.empty()) return false;
    double sum = 0.0;
    for (double v : p) sum += v;
    if (std::fabs(sum - 1.0) > 1e-6) return false;
    for (double v : p) if (!std::isfinite(v) || v < 0.0) return false;
    return true;
}

// A run of >= 4 consecutive MACs of (f64 slot) x (constant) -- the linear predictor of a regression written as
// `mean = mean + beta[j] * x[i][j]` -- becomes one FG_OP_DOT whose terms sit in the constant pool: the same mul and add
// per term, in the same order, without a 96-byte instruction fetch and an operand decode per term.
static void fuse_dots(std::vector<FgIns> &out, size_t start, std::vector<double> &pool) {
    auto term = [](const FgIns &I, uint32_t &slot, double &c) {
        if (FG_INS_OPCODE(I.op) != FG_OP_MAC) return false;
        const uint32_t k0 = FG_OPND_KIND(I.opnd[0]), k1 = FG_OPND_KIND(I.opnd[1]);
        if (k0 == FG_OPND_SLOT_F && k1 == FG_OPND_IMM) { slot = FG_OPND_IDX(I.opnd[0]); c = I.imm[1]; return true; }
        if (k0 == FG_OPND_IMM && k1 == FG_OPND_SLOT_F) { slot = FG_OPND_IDX(I.opnd[1]); c = I.imm[0]; return true; }
        return false;
    };
    std::vector<FgIns> res;                                            // the statement's instructions only (a copy of the whole prefix per statement made compilation quadratic)
    for (size_t i = start; i < out.size();) {
        uint32_t slot; double c;
        size_t j = i;
        while (j < out.size() && term(out[j], slot, c)) ++j;
        if (j - i < 4) { for (size_t k = i; k < std::max(j, i + 1); ++k) res.push_back(out[k]); i = std::max(j, i + 1); continue; }
        if (pool.size() & 1) pool.push_back(0.0);                  // 16-byte aligned terms
        FgIns D = FgGen::blank(FG_OP_DOT);
        D.opnd[0] = FG_OPND(FG_OPND_IMM, 0);
        D.opnd[1] = (uint32_t)(j - i);
        D.aux = (uint32_t)pool.size();
        for (size_t k = i; k < j; ++k) { term(out[k], slot, c); pool.push_back(fg_as_double((long long)slot)); pool.push_back(c); }
        res.push_back(D);
        i = j;
    }
    out.resize(start);
    out.insert(out.end(), res.begin(), res.end());
}

void fg_program::compile_stmt(const FgStmt &s, std::vector<FgIns> &out, int &temp_max) {
    FgGen G(*this, out);
    const size_t start = out.size();
    if (s.kind == 2) {                                   // factor
        FgIns I = FgGen::blank(FG_OP_FACTOR);
        G.operand(s.value, I.opnd[0], I.imm[0]);
        out.push_back(I);
        temp_max = std::max(temp_max, G.temp_max);
        fuse_dots(out, start, pool);
        return;
    }
    uint32_t op = (uint32_t)s.dist | ((uint32_t)s.vtype << FG_F_VTYPE_SHIFT);
    if (s.kind == 1) op |= FG_F_OBSERVE;
    FgIns I = FgGen::blank(op);
    bool params_const = true;
    for (int p : s.params) params_const = params_const && nodes[p].is_const;

    if (s.dist == FG_CATEGORICAL) {
        int K = (int)s.params.size();
        I.opnd[2] = (uint32_t)K;
        if (params_const) {
            std::vector<double> pr;
            for (int p : s.params) pr.push_back(nodes[p].cval);
            int base = (int)pool.size();
            for (double v : pr) pool.push_back(v);
            for (double v : pr) pool.push_back(v > 0.0 ? std::log(v) : -INFINITY);   // distribution.rs:785-791
            I.opnd[1] = FG_OPND(FG_OPND_POOL, base);
            I.op |= FG_F_HOISTED;
            if (!fg_categorical_const_valid(pr)) I.op |= FG_F_INVALID;
        } else {
            int base = G.new_temp(K);
            for (int k = 0; k < K; k++) { G.gen(s.params[k]); G.store(base + k); }
            I.opnd[1] = FG_OPND(FG_OPND_SLOT_F, base);
        }
    } else {
        for (size_t k = 0; k < s.params.size(); k++) G.operand(s.params[k], I.opnd[1 + k], I.imm[1 + k]);
        if (params_const) {
            double p0 = s.params.size() > 0 ? nodes[s.params[0]].cval : 0.0;
            double p1 = s.params.size() > 1 ? nodes[s.params[1]].cval : 0.0;
            double p2 = s.params.size() > 2 ? nodes[s.params[2]].cval : 0.0;
            I.op |= FG_F_HOISTED;
            if (s.dist == FG_DISCRETEUNIFORM) {
                long long lo = fg_f2i_sat(p0), hi = fg_f2i_sat(p1);
                if (s.exact_bounds) { lo = s.lo; hi = s.hi; }
                I.imm[1] = fg_as_double(lo); I.imm[2] = fg_as_double(hi);
                I.h[0] = (hi < lo) ? -INFINITY : fg_du_logp(lo, hi);
            } else if (!fg_hoist((uint32_t)s.dist, p0, p1, p2, I.h)) {
                I.op |= FG_F_INVALID;
            } else {
                const int sp = fg_scale_param((uint32_t)s.dist);
                const double sc = sp == 1 ? p1 : p2;
                if (sp >= 0 && fg_pow2_scale(sc)) { I.op |= FG_F_POW2SCALE; I.h[4] = 1.0 / sc; }
            }
        }
    }
    // scale-only hoisting: Normal / LogNormal / Cauchy / Laplace with a constant valid scale and a varying location
    if (!params_const && (s.dist == FG_NORMAL || s.dist == FG_LOGNORMAL || s.dist == FG_CAUCHY || s.dist == FG_LAPLACE) &&
        nodes[s.params[1]].is_const) {
        const double sc = nodes[s.params[1]].cval;
        if (sc > 0.0 && std::isfinite(sc)) {
            I.op |= FG_F_SCALEHOIST;
            I.h[0] = (s.dist == FG_CAUCHY) ? (-FG_LN_PI - std::log(sc)) : (s.dist == FG_LAPLACE) ? -std::log(2.0 * sc) : std::log(sc);
            if (fg_pow2_scale(sc)) { I.op |= FG_F_POW2SCALE; I.h[4] = 1.0 / sc; }
        }
    }
    if (s.kind == 0) {                                   // sample: x is the site's own slot
        I.aux = (uint32_t)site_slot[s.sorted];
        I.opnd[0] = FG_OPND(s.vtype == FG_F64 ? FG_OPND_SLOT_F : FG_OPND_SLOT_I, site_slot[s.sorted]);
    } else {
        G.operand(s.value, I.opnd[0], I.imm[0]);
        // a constant observed count under varying parameters (a count regression): its own term -- ln k!, ln C(n, k) -- once, here
        if (!params_const && nodes[s.value].is_const && s.vtype != FG_F64 && s.vtype != FG_BOOL && (s.dist == FG_POISSON || s.dist == FG_BINOMIAL)) {
            const double v = nodes[s.value].cval;
            const long long xi = std::isfinite(v) ? (long long)v : 0LL;                 // fg_int_of (fg_interp.h)
            if (s.dist == FG_POISSON && xi >= 0) { I.op |= FG_F_XHOIST; I.h[3] = fg_lgamma((double)xi + 1.0); }
            if (s.dist == FG_BINOMIAL && nodes[s.params[0]].is_const && xi >= 0) {
                const unsigned long long n = (unsigned long long)nodes[s.params[0]].cval, k = (unsigned long long)xi;
                if (k <= n) { I.op |= FG_F_XHOIST; I.h[3] = fg_lgamma((double)n + 1.0) - fg_lgamma((double)k + 1.0) - fg_lgamma((double)(n - k) + 1.0); }
            }
        }
    }
    out.push_back(I);
    temp_max = std::max(temp_max, G.temp_max);
    fuse_dots(out, start, pool);
}

int fg_program::finalize() {
    // site order = lexicographic order of the address bytes (src/core/address.rs:150-157)
    std::vector<int> order;
    for (int i = 0; i < (int)stmts.size(); i++) if (stmts[i].kind == 0) order.push_back(i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return stmts[a].addr < stmts[b].addr; });
    for (size_t j = 0; j + 1 < order.size(); j++)
        if (stmts[order[j]].addr == stmts[order[j + 1]].addr) {
            fg_set_error("address `" + stmts[order[j]].addr + "` was sampled twice (AddressConflict)");
            return FG_ERR_ADDRESS_CONFLICT;   // the panic of interpreters.rs:23-33
        }
    int S = (int)order.size();
    sorted_stmt = order;
    handle_to_sorted.assign(S, 0);
    site_vtype.assign(S, 0);
    f64_slot.clear();
    for (int j = 0; j < S; j++) {
        FgStmt &s = stmts[order[j]];
        s.sorted = j;
        handle_to_sorted[s.handle] = j;
        site_vtype[j] = s.vtype;
        if (s.vtype == FG_F64) f64_slot.push_back(j);
    }
    // LDS slot numbering: the f64 coordinates first (slot k = coordinate k, so the leapfrog loops need no
    // index table), then the discrete sites; expression temporaries follow from slot S
    site_slot.assign(S, 0);
    {
        int next = (int)f64_slot.size();
        for (int k = 0; k < (int)f64_slot.size(); k++) site_slot[f64_slot[k]] = k;
        for (int j = 0; j < S; j++) if (site_vtype[j] != FG_F64) site_slot[j] = next++;
    }
    // compile every statement; remember its instruction range and the sites it reads
    ins.clear(); sub.clear(); sub_off.clear(); pool.clear();
    site_cat.assign(2 * (size_t)std::max(1, S), -1);
    int temp_max = S;
    std::vector<std::pair<int, int>> range(stmts.size());
    std::vector<std::vector<int>> reads(stmts.size());
    for (size_t i = 0; i < stmts.size(); i++) {
        int b = (int)ins.size();
        compile_stmt(stmts[i], ins, temp_max);
        range[i] = {b, (int)ins.size()};
        if (stmts[i].kind == 0) {              // Categorical site with a valid constant table: prior-resample needs only the table
            const FgIns &D = ins.back();
            if (FG_INS_OPCODE(D.op) == (uint32_t)FG_CATEGORICAL && FG_OPND_KIND(D.opnd[1]) == FG_OPND_POOL && !(D.op & FG_F_INVALID)) {
                site_cat[2 * stmts[i].sorted] = (int)FG_OPND_IDX(D.opnd[1]);
                site_cat[2 * stmts[i].sorted + 1] = (int)D.opnd[2];
            }
        }
        std::vector<int> hs;
        for (int p : stmts[i].params) collect_sites(p, hs);
        if (stmts[i].kind != 0 && stmts[i].value >= 0) collect_sites(stmts[i].value, hs);
        if (stmts[i].kind == 0) hs.push_back(stmts[i].handle);
        for (int h : hs) reads[i].push_back(handle_to_sorted[h]);
    }
    // slot `temp_max` is the always-zero slot read by the constant operands of the fast opcodes
    const int zero_slot = temp_max;
    n_slots = temp_max + 1;
    // score-only variant: Normal sites with a constant valid sigma and (constant | f64 slot) x, mu
    ins_fast = ins;
    for (FgIns &I : ins_fast) {
        if (FG_INS_OPCODE(I.op) != (uint32_t)FG_NORMAL || (I.op & FG_F_INVALID)) continue;
        const uint32_t kx = FG_OPND_KIND(I.opnd[0]), km = FG_OPND_KIND(I.opnd[1]), ks = FG_OPND_KIND(I.opnd[2]);
        const double sg = I.imm[2];
        if (ks != FG_OPND_IMM || !(sg > 0.0) || !std::isfinite(sg)) continue;
        if ((kx != FG_OPND_IMM && kx != FG_OPND_SLOT_F) || (km != FG_OPND_IMM && km != FG_OPND_SLOT_F)) continue;
        if (km == FG_OPND_IMM && !std::isfinite(I.imm[1])) continue;
        FgIns F = FgGen::blank(FG_OP_NORMAL_FAST | (I.op & FG_F_OBSERVE));
        F.opnd[0] = kx == FG_OPND_IMM ? (uint32_t)zero_slot : FG_OPND_IDX(I.opnd[0]);
        F.opnd[1] = km == FG_OPND_IMM ? (uint32_t)zero_slot : FG_OPND_IDX(I.opnd[1]);
        F.imm[0] = kx == FG_OPND_IMM ? I.imm[0] : 0.0;
        F.imm[1] = km == FG_OPND_IMM ? I.imm[1] : 0.0;
        F.imm[2] = sg;
        F.aux = I.aux;
        F.h[0] = std::log(sg);                                   // ln(sigma), distribution.rs:207
        if (fg_pow2_scale(sg)) { F.op |= FG_F_POW2SCALE; F.h[4] = 1.0 / sg; }
        else if (fg_div_const_ok(sg)) { F.op |= FG_F_RCPSCALE; F.h[4] = 1.0 / sg; }
        I = F;
    }
    // per-coordinate sub-programs for the sparse finite difference (built from the score-only variant)
    sub_off.push_back(0);
    coord.clear();
    for (int slot : f64_slot) {
        for (size_t i = 0; i < stmts.size(); i++)
            if (std::find(reads[i].begin(), reads[i].end(), slot) != reads[i].end())
                sub.insert(sub.end(), ins_fast.begin() + range[i].first, ins_fast.begin() + range[i].second);
        FgCoord cd; cd.slot = (int)coord.size(); cd.sub_off = sub_off.back(); cd.sub_n = (int)sub.size() - sub_off.back(); cd.flags = 0;
        coord.push_back(cd);
        sub_off.push_back((int)sub.size());
    }
    // ---- statement shapes the record streams understand --------------------------------------------------------
    //   FAST: one FG_OP_NORMAL_FAST instruction (x, mu each an f64 slot or a constant, constant sigma);
    //   LIN : a Normal with constant sigma whose mean is a linear predictor,  LOAD c0; {MAC | DOT}...; STORE t;
    //         NORMAL_FAST(x, mu = slot t)  --  mu = (..((c0 + s_0 c_0) + s_1 c_1)..): its terms go to the pool once.
    struct Shape { int kind = 0; FgIns F; double init = 0.0; uint32_t toff = 0, tn = 0; std::vector<uint32_t> tslot; uint32_t zslot = 0; };
    std::vector<Shape> shape(stmts.size());
    for (size_t i = 0; i < stmts.size(); i++) {
        const int b = range[i].first, e = range[i].second;
        Shape &sh = shape[i];
        if (e - b == 1 && FG_INS_OPCODE(ins_fast[b].op) == FG_OP_NORMAL_FAST) { sh.kind = 1; sh.F = ins_fast[b]; continue; }
        if (e - b == 1 && stmts[i].kind != 2 && FG_INS_OPCODE(ins_fast[b].op) == (uint32_t)FG_CATEGORICAL) {   // CATC: constant-table Categorical site
            const FgIns &D = ins_fast[b];
            if (FG_OPND_KIND(D.opnd[1]) == FG_OPND_POOL && !(D.op & FG_F_INVALID) && FG_OPND_KIND(D.opnd[0]) == FG_OPND_SLOT_I) {
                sh.kind = 5; sh.F = D; sh.toff = FG_OPND_IDX(D.opnd[1]); sh.tn = D.opnd[2];
            }
            continue;
        }
        {   // NSEL: K x [LOAD option; STORE base+k]; LOAD z; GATHER base, K; STORE t; NORMAL_FAST(x, mu = t)
            const int n = e - b;
            if (n >= 6 && FG_INS_OPCODE(ins_fast[e - 1].op) == FG_OP_NORMAL_FAST && FG_INS_OPCODE(ins_fast[e - 2].op) == FG_OP_STORE &&
                FG_INS_OPCODE(ins_fast[e - 3].op) == FG_OP_GATHER && FG_INS_OPCODE(ins_fast[e - 4].op) == FG_OP_LOAD) {
                const FgIns &F = ins_fast[e - 1], &St = ins_fast[e - 2], &G = ins_fast[e - 3], &Lz = ins_fast[e - 4];
                const uint32_t K = G.opnd[1], base = G.aux;
                bool ok = (int)K >= 1 && (int)K <= 64 && n == 2 * (int)K + 4 && F.opnd[1] == St.aux && (int)St.aux >= S && F.opnd[0] != St.aux && F.imm[1] == 0.0 &&
                          FG_OPND_KIND(Lz.opnd[0]) == FG_OPND_SLOT_I;
                std::vector<double> opts;
                for (uint32_t k = 0; k < K && ok; k++) {
                    const FgIns &Lo = ins_fast[b + 2 * k], &So = ins_fast[b + 2 * k + 1];
                    ok = FG_INS_OPCODE(Lo.op) == FG_OP_LOAD && FG_INS_OPCODE(So.op) == FG_OP_STORE && So.aux == base + k;
                    if (!ok) break;
                    if (FG_OPND_KIND(Lo.opnd[0]) == FG_OPND_SLOT_F && (int)FG_OPND_IDX(Lo.opnd[0]) < (int)f64_slot.size()) { opts.push_back(fg_as_double((long long)FG_OPND_IDX(Lo.opnd[0]))); opts.push_back(0.0); }
                    else if (FG_OPND_KIND(Lo.opnd[0]) == FG_OPND_IMM) { opts.push_back(fg_as_double((long long)zero_slot | (1LL << 32))); opts.push_back(Lo.imm[0]); }
                    else ok = false;
                }
                if (ok) {
                    if (pool.size() & 1) pool.push_back(0.0);
                    // an option list already in the pool is shared (a mixture's observations all select among the same means): the
                    // pool stays small enough to be staged into LDS by the kernels that look options up per lane
                    size_t at = pool.size();
                    for (size_t o = 0; o + opts.size() <= pool.size(); o += 2)
                        if (std::memcmp(&pool[o], opts.data(), opts.size() * sizeof(double)) == 0) { at = o; break; }
                    sh.kind = 4; sh.F = F; sh.toff = (uint32_t)at; sh.tn = K; sh.zslot = FG_OPND_IDX(Lz.opnd[0]);
                    if (at == pool.size()) pool.insert(pool.end(), opts.begin(), opts.end());
                    continue;
                }
            }
        }
        if (e - b == 1 && stmts[i].kind != 2) {             // GEN: one distribution instruction whose operands are all leaves
            const FgIns &D = ins_fast[b];
            const uint32_t code = FG_INS_OPCODE(D.op);
            bool ok = code < 17u && code != (uint32_t)FG_CATEGORICAL && code != (uint32_t)FG_DISCRETEUNIFORM;
            const uint32_t kx = FG_OPND_KIND(D.opnd[0]);
            const uint32_t vt0 = FG_INS_VTYPE(D.op);
            ok = ok && (kx == FG_OPND_IMM || (kx == FG_OPND_SLOT_F && vt0 == (uint32_t)FG_F64) || (kx == FG_OPND_SLOT_I && vt0 != (uint32_t)FG_F64));
            for (int q = 1; q <= 3 && ok; q++) { const uint32_t kq = FG_OPND_KIND(D.opnd[q]); ok = kq == FG_OPND_IMM || kq == FG_OPND_SLOT_F; }
            if (ok && (D.op & FG_F_HOISTED) && (D.h[2] != 0.0 || D.h[3] != 0.0)) ok = false;   // only h0, h1 fit the record (Binomial with constant parameters stays out)
            if (ok) { sh.kind = 3; sh.F = D; continue; }
        }
        if (e - b < 3 || FG_INS_OPCODE(ins_fast[e - 1].op) != FG_OP_NORMAL_FAST) continue;
        const FgIns &L = ins_fast[b], &St = ins_fast[e - 2], &F = ins_fast[e - 1];
        if (FG_INS_OPCODE(L.op) != FG_OP_LOAD) continue;
        if (FG_INS_OPCODE(St.op) != FG_OP_STORE || F.opnd[1] != St.aux || (int)St.aux < S || F.opnd[0] == St.aux || F.imm[1] != 0.0) continue;
        // every form becomes  mu = c0 + sum_t slot_t * c_t  with the same value: LOAD slot is 0 + slot*1, LOAD slot; MUL c is
        // 0 + slot*c, `+ slot` is `+ slot*1` (a product with 1 and a sum with +0 are exact; only the sign of a zero mean can
        // differ, which x - mu cannot see).  `+ constant` in the middle of a predictor is left to the interpreter.
        std::vector<double> terms;
        auto push = [&](uint32_t slot, double c) { terms.push_back(fg_as_double((long long)slot)); terms.push_back(c); };
        bool ok = true;
        int k = b + 1;
        double init = 0.0;
        if (FG_OPND_KIND(L.opnd[0]) == FG_OPND_IMM) init = L.imm[0];
        else if (FG_OPND_KIND(L.opnd[0]) == FG_OPND_SLOT_F) {
            if (k < e - 2 && FG_INS_OPCODE(ins_fast[k].op) == FG_OP_MUL && FG_OPND_KIND(ins_fast[k].opnd[0]) == FG_OPND_IMM) { push(FG_OPND_IDX(L.opnd[0]), ins_fast[k].imm[0]); ++k; }
            else push(FG_OPND_IDX(L.opnd[0]), 1.0);
        } else continue;
        for (; k < e - 2 && ok; k++) {
            const FgIns &I = ins_fast[k];
            const uint32_t oc = FG_INS_OPCODE(I.op), k0 = FG_OPND_KIND(I.opnd[0]), k1 = FG_OPND_KIND(I.opnd[1]);
            if (oc == FG_OP_DOT) {
                for (uint32_t t = 0; t < I.opnd[1]; t++) push((uint32_t)fg_as_i64(pool[I.aux + 2 * t]), pool[I.aux + 2 * t + 1]);
            } else if (oc == FG_OP_MAC) {
                if (k0 == FG_OPND_SLOT_F && k1 == FG_OPND_IMM) push(FG_OPND_IDX(I.opnd[0]), I.imm[1]);
                else if (k0 == FG_OPND_IMM && k1 == FG_OPND_SLOT_F) push(FG_OPND_IDX(I.opnd[1]), I.imm[0]);
                else ok = false;
            } else if (oc == FG_OP_ADD && k0 == FG_OPND_SLOT_F) push(FG_OPND_IDX(I.opnd[0]), 1.0);
            else ok = false;
        }
        if (!ok || terms.empty() || terms.size() / 2 > 0xffffu) continue;
        for (size_t t = 0; t < terms.size(); t += 2) {
            const uint32_t sl = (uint32_t)fg_as_i64(terms[t]);
            if ((int)sl >= (int)f64_slot.size()) ok = false;                 // every term must be an f64 coordinate (slot k = coordinate k)
            sh.tslot.push_back(sl);
        }
        if (!ok) { sh.tslot.clear(); continue; }
        if (pool.size() & 1) pool.push_back(0.0);
        sh.kind = 2; sh.F = F; sh.init = init; sh.toff = (uint32_t)pool.size(); sh.tn = (uint32_t)(terms.size() / 2);
        pool.insert(pool.end(), terms.begin(), terms.end());
    }
    for (int q = 0; q < 8; q++) pool.push_back(0.0);                   // term groups are read 4 at a time
    auto make_rec = [&](const Shape &sh, int coord_k /* -1: score record */) {
        const FgIns &F = sh.F;
        FgGradRec r; std::memset(&r, 0, sizeof(r));
        if (sh.kind == 5) {                                          // Categorical site, constant table
            r.xi = FG_OPND_IDX(F.opnd[0]); r.mi = (uint32_t)zero_slot; r.flags = FG_G_CATC;
            const uint32_t w[2] = { sh.toff, sh.tn }; std::memcpy(&r.mimm, w, 8);
            r.sigma = 1.0; r.inv = 1.0;
            return r;
        }
        if (sh.kind == 3) {
            FgGenRec g; std::memset(&g, 0, sizeof(g));
            const uint32_t vt = FG_INS_VTYPE(F.op), kx = FG_OPND_KIND(F.opnd[0]);
            g.xi = kx == FG_OPND_IMM ? (uint32_t)zero_slot : FG_OPND_IDX(F.opnd[0]);
            g.mi = (uint32_t)zero_slot;
            g.flags = FG_G_GEN | (FG_INS_OPCODE(F.op) << 16) | (kx == FG_OPND_IMM ? FG_G_X_CONST : 0u) | (vt != (uint32_t)FG_F64 ? FG_G_GEN_XINT : 0u) |
                      ((F.op & FG_F_HOISTED) ? FG_G_GEN_HOISTED : 0u) | ((F.op & FG_F_SCALEHOIST) ? FG_G_GEN_SH : 0u) | ((F.op & FG_F_INVALID) ? FG_G_GEN_INVALID : 0u) |
                      ((F.op & FG_F_XHOIST) ? FG_G_GEN_XH : 0u);
            // an observed constant of a discrete distribution is its integer value (fg_int_of, fg_interp.h)
            if (kx == FG_OPND_IMM) g.x = vt == (uint32_t)FG_F64 ? F.imm[0]
                                        : fg_as_double(vt == (uint32_t)FG_BOOL ? (long long)(F.imm[0] != 0.0) : (std::isfinite(F.imm[0]) ? (long long)F.imm[0] : 0LL));
            for (int q = 0; q < 3; q++) {
                if (FG_OPND_KIND(F.opnd[1 + q]) == FG_OPND_SLOT_F) { g.flags |= FG_G_GEN_P0SLOT << q; g.p[q] = fg_as_double((long long)FG_OPND_IDX(F.opnd[1 + q])); }
                else g.p[q] = F.imm[1 + q];
            }
            g.h[0] = F.h[0]; g.h[1] = (F.op & FG_F_XHOIST) ? F.h[3] : F.h[1];
            if (coord_k >= 0) g.coord = (uint32_t)coord_k;
            std::memcpy(&r, &g, sizeof(r));
            return r;
        }
        r.xi = F.opnd[0]; r.mi = sh.kind == 2 ? (uint32_t)zero_slot : F.opnd[1];
        r.flags = ((F.op & FG_F_POW2SCALE) ? FG_G_POW2 : 0u) | ((F.op & (FG_F_POW2SCALE | FG_F_RCPSCALE)) ? 0u : FG_G_DIV) |
                  (r.xi == (uint32_t)zero_slot ? FG_G_X_CONST : 0u) | ((sh.kind == 1 && r.mi == (uint32_t)zero_slot) ? FG_G_M_CONST : 0u);
        r.ximm = F.imm[0]; r.mimm = sh.kind == 2 ? sh.init : F.imm[1]; r.sigma = F.imm[2]; r.inv = 1.0 / F.imm[2]; r.lns = F.h[0];
        if (sh.kind == 4) {                                          // mu = options[z]
            r.mi = sh.zslot; r.flags |= FG_G_NSEL; r.flags &= ~(uint32_t)FG_G_M_CONST;
            const uint32_t w[2] = { sh.toff, sh.tn }; std::memcpy(&r.mimm, w, 8);
        }
        if (sh.kind == 2) {
            uint32_t pos = sh.tn;                                      // first term that reads the coordinate (tn: none)
            uint32_t hits = 0;
            if (coord_k >= 0) for (uint32_t t = 0; t < sh.tn; t++) if (sh.tslot[t] == (uint32_t)coord_k) { if (!hits) pos = t; ++hits; }
            r.flags |= FG_G_LIN | (pos << 16) | (hits == 1u ? FG_G_LIN1 : 0u);
            r.maskx = sh.toff; r.maskm = sh.tn;                        // LIN: pool offset and number of terms
        }
        if (coord_k >= 0) {
            r.coord = (uint32_t)coord_k;
            if (r.xi == (uint32_t)coord_k) r.flags |= FG_G_PERT_X;
            if (sh.kind == 1 && r.mi == (uint32_t)coord_k) r.flags |= FG_G_PERT_M;
        }
        return r;
    };
    FgGradRec pad; std::memset(&pad, 0, sizeof(pad));
    pad.xi = pad.mi = (uint32_t)zero_slot; pad.sigma = 1.0; pad.inv = 1.0; pad.flags = FG_G_POW2;
    // fused FD gradient stream: possible when every statement that reads an f64 coordinate is FAST or LIN
    gstream.clear(); n_gstream = 0;
    {
        bool ok = !f64_slot.empty();
        std::vector<std::vector<int>> dep(f64_slot.size());          // statements reading coordinate k, program order
        for (size_t k = 0; k < f64_slot.size() && ok; k++) {
            for (size_t i = 0; i < stmts.size(); i++)
                if (std::find(reads[i].begin(), reads[i].end(), f64_slot[k]) != reads[i].end()) { dep[k].push_back((int)i); ok = ok && shape[i].kind != 0; }
            ok = ok && !dep[k].empty();
        }
        if (ok) {
            // log_prior and log_likelihood are separate accumulators (trace.rs:168-177), so within one coordinate
            // the prior records may be emitted before the observe records without changing either sum
            for (size_t k = 0; k < f64_slot.size(); k++) {
                std::vector<int> order;
                for (int pass = 0; pass < 2; pass++)
                    for (int i : dep[k]) if (((shape[i].F.op & FG_F_OBSERVE) != 0u) == (pass == 1)) order.push_back(i);
                bool seen_obs = false;
                for (size_t q = 0; q < order.size(); q++) {
                    const bool obs = (shape[order[q]].F.op & FG_F_OBSERVE) != 0u;
                    FgGradRec r = make_rec(shape[order[q]], (int)k);
                    if (obs && !seen_obs) r.flags |= FG_G_SWITCH;
                    if (q + 1 == order.size()) r.flags |= FG_G_END;
                    if (shape[order[q]].kind == 1) r.maskx = (uint32_t)order[q];    // fast Normal: its statement index (= score-stream index)
                    seen_obs = seen_obs || obs;
                    gstream.push_back(r);
                }
            }
            n_gstream = (int)gstream.size();
            for (int q = 0; q < 4; q++) gstream.push_back(pad);   // the stream is read 3 records ahead
        }
    }
    // score stream: when the WHOLE program is FAST / LIN statements its endpoint score (score_full, hmc.rs:283-299)
    // is a lean pass over one 64-byte record per statement, in program order (the accumulation order of
    // PriorHandler / ScoreGivenTrace), instead of a pass of the general interpreter
    sstream.clear(); n_sstream = 0; sstream_has_lin = false; sstream_has_gen = false; sstream_has_genrec = false;
    {
        bool ok = !stmts.empty();
        for (const Shape &sh : shape) ok = ok && sh.kind != 0;
        if (ok) {
            for (const Shape &sh : shape) {
                FgGradRec r = make_rec(sh, -1);
                if (sh.F.op & FG_F_OBSERVE) r.flags |= FG_S_OBS;
                sstream_has_lin = sstream_has_lin || sh.kind == 2;
                sstream_has_gen = sstream_has_gen || sh.kind >= 3;
                sstream_has_genrec = sstream_has_genrec || sh.kind == 3;
                sstream.push_back(r);
            }
            n_sstream = (int)sstream.size();
            for (int q = 0; q < 4; q++) sstream.push_back(pad);
        }
    }
    // observe bits of the score stream (the parallel-terms endpoint score sums prior and likelihood terms apart)
    sobs.assign((size_t)(n_sstream + 31) / 32 + 1, 0u);
    for (int k = 0; k < n_sstream; k++) if (sstream[k].flags & FG_S_OBS) sobs[k >> 5] |= 1u << (k & 31);
    // term rows of the score stream (multi-wave kernels: every wave evaluates a share of the statements' log-densities into
    // LDS rows, one wave adds them in program order): log_prior terms first, then log_likelihood terms, each in program
    // order; the row sits in the record's (otherwise unused) `coord` field.  site_rec: record of each site's sample statement.
    n_prior_terms = 0;
    site_rec.assign((size_t)std::max(1, S), -1);
    {
        int n_pri = 0, n_lik = 0;
        for (int k = 0; k < n_sstream; k++) if (!(sstream[k].flags & FG_S_OBS)) sstream[k].coord = (uint32_t)n_pri++;
        for (int k = 0; k < n_sstream; k++) if (sstream[k].flags & FG_S_OBS) sstream[k].coord = (uint32_t)(n_pri + n_lik++);
        n_prior_terms = n_pri;
        if (n_sstream > 0) for (size_t i = 0; i < stmts.size(); i++) if (stmts[i].kind == 0) site_rec[stmts[i].sorted] = (int)i;
    }
    // independent-sites programs: compact per-coordinate records for the register-resident trajectories
    sep.clear(); sep_coord.clear(); sep_free.clear();
    if (n_gstream > 0 && n_sstream > 0 && !sstream_has_lin && !sstream_has_gen) {
        bool ok = true;
        // LDS row of every statement's score term: prior terms first, then likelihood terms, each in program order
        std::vector<uint32_t> trow((size_t)n_sstream, 0u);
        for (int k = 0; k < n_sstream; k++) trow[k] = sstream[k].coord;
        std::vector<char> covered((size_t)n_sstream, 0);
        std::vector<FgSepRec> recs; std::vector<FgSepCoord> cds(f64_slot.size(), FgSepCoord{0, 0});
        for (int k = 0; k < n_gstream && ok; k++) {
            const FgGradRec &r = gstream[k];
            const uint32_t ck = r.coord;
            if (r.flags & (FG_G_LIN | FG_G_GEN | FG_G_NSEL | FG_G_CATC)) { ok = false; break; }
            const bool x_own = (r.flags & FG_G_PERT_X) != 0u, m_own = (r.flags & FG_G_PERT_M) != 0u;
            if (x_own == m_own) ok = false;                              // exactly one operand is the coordinate ...
            if (!x_own && !(r.flags & FG_G_X_CONST)) ok = false;        // ... and the other one a constant
            if (!m_own && !(r.flags & FG_G_M_CONST)) ok = false;
            const int si = (int)r.maskx;                                 // statement index (set when the stream was built)
            if (si < 0 || si >= n_sstream || covered[si]) ok = false;
            if (!ok) break;
            covered[si] = 1;
            const bool first = cds[ck].n == 0, obs = (sstream[si].flags & FG_S_OBS) != 0u;
            if (first == obs) ok = false;                                // record 0 = the coordinate's own sample statement, the rest observes
            if (first && !x_own) ok = false;
            FgSepRec q; std::memset(&q, 0, sizeof(q));
            q.flags = r.flags & (FG_G_POW2 | FG_G_DIV);
            q.trow = trow[si];
            q.c = x_own ? r.mimm : r.ximm;
            q.inv = r.inv; q.lns = r.lns; q.sigma = r.sigma;
            if (first) cds[ck].off = (int)recs.size();
            cds[ck].n += 1;
            if (cds[ck].n > FG_SEP_MAXREC) ok = false;
            recs.push_back(q);
        }
        for (size_t k = 0; k < cds.size() && ok; k++) {
            ok = cds[k].n > 0;
            bool p2 = true;
            for (int q = 0; q < cds[k].n && ok; q++) p2 = p2 && (recs[cds[k].off + q].flags & FG_G_POW2);
            if (ok && p2) cds[k].n |= 256;
            if (ok && p2) {                                              // the own sample statement is Normal(0, 1): c = 0, 1 / sigma = 1, ln sigma = 0
                const FgSepRec &r0 = recs[cds[k].off];
                if (r0.c == 0.0 && r0.inv == 1.0 && r0.lns == 0.0 && r0.sigma == 1.0) cds[k].n |= 512;
            }
        }
        if ((size_t)n_slots + 2 * f64_slot.size() + (size_t)n_sstream + 3 > 320) ok = false;     // q, kinetic and term rows of a tile in 160 KB of LDS
        if (ok) {
            for (int k = 0; k < n_sstream; k++) if (!covered[k]) sep_free.push_back(FgSepFree{(uint32_t)k, trow[k]});
            sep = recs; sep_coord = cds;
            for (int q = 0; q < FG_SEP_MAXREC; q++) { FgSepRec z; std::memset(&z, 0, sizeof(z)); sep.push_back(z); }
        }
    }
    // dense regressions (fg_hmc_lin.hip): every coordinate's records are >= 1 prior records that read only the coordinate itself,
    // followed by the SAME N linear-predictor observe records, and every predictor reads all d coordinates once, in one common
    // term order.  The table holds each statement's constants and coefficients once, in term order.
    lin_tab.clear(); lin_meta.clear(); lin_n = 0; lin_p2 = 0;
    if (n_gstream > 0 && n_sstream > 0 && sstream_has_lin && !sstream_has_gen && f64_slot.size() >= 2) {
        const int d = (int)f64_slot.size();
        bool ok = true;
        std::vector<int> cstart((size_t)d + 1, n_gstream);
        for (int k = n_gstream - 1; k >= 0; --k) cstart[gstream[k].coord] = k;
        std::vector<int> npri((size_t)d, 0);
        std::vector<uint32_t> stm;                                   // pool offsets of the statements' terms = their identity
        for (int k = 0; k < d && ok; k++) {
            const int b = cstart[k], e = cstart[k + 1];
            int q = b;
            for (; q < e && !(gstream[q].flags & FG_G_LIN); ++q) {
                const FgGradRec &r = gstream[q];
                const bool x_own = (r.flags & FG_G_PERT_X) != 0u, m_own = (r.flags & FG_G_PERT_M) != 0u;
                if ((r.flags & (FG_G_GEN | FG_G_NSEL | FG_G_CATC | FG_G_SWITCH)) || x_own == m_own || (!x_own && !(r.flags & FG_G_X_CONST)) ||
                    (!m_own && !(r.flags & FG_G_M_CONST))) ok = false;
            }
            npri[k] = q - b;
            if (npri[k] < 1 || q == e || !(gstream[q].flags & FG_G_SWITCH)) ok = false;
            std::vector<uint32_t> mine;
            for (; q < e && ok; ++q) {
                const FgGradRec &r = gstream[q];
                if ((r.flags & (FG_G_LIN | FG_G_LIN1 | FG_G_X_CONST | FG_G_PERT_X)) != (FG_G_LIN | FG_G_LIN1 | FG_G_X_CONST) || (int)r.maskm != d) ok = false;
                mine.push_back(r.maskx);
            }
            if (k == 0) stm = mine; else if (mine != stm) ok = false;
        }
        std::vector<int> coord_at((size_t)d, -1);
        for (size_t s = 0; s < stm.size() && ok; s++)
            for (int t = 0; t < d && ok; t++) {
                const int slot = (int)fg_as_i64(pool[stm[s] + 2 * (size_t)t]);
                if (s == 0) { coord_at[t] = slot; for (int u = 0; u < t; u++) if (coord_at[u] == slot) ok = false; }
                else if (coord_at[t] != slot) ok = false;
                if (slot < 0 || slot >= d) ok = false;
            }
        if (ok && !stm.empty() && d <= 64) {
            // the kernel is built for 8, 16, 32 and 64 term positions: any other d <= 64 is padded with terms that read the always-zero slot
            // with coefficient +0.0 -- behind the last real term every sum only gains "+ (+0.0)", which changes no bit of a sum that is
            // not -0.0 and no log-density at all (the predictor enters through (y - mu)^2)
            const int dp = d <= 8 ? 8 : (d <= 16 ? 16 : (d <= 32 ? 32 : 64));
            const int rowd = FG_LIN_ROW_DOUBLES(dp);
            lin_n = (int)stm.size(); lin_p2 = 1;
            lin_tab.assign(((size_t)lin_n + 1) * rowd, 0.0);
            for (int s = 0; s < lin_n; s++) {
                const FgGradRec &r = gstream[cstart[0] + npri[0] + s];
                double *row = &lin_tab[(size_t)s * rowd];
                row[0] = r.mimm;
                for (int t = 0; t < d; t++) row[2 + t] = pool[r.maskx + 2 * (size_t)t + 1];
                double *tail = row + 2 + dp;
                tail[0] = r.ximm + 0.0;                              // (x + hx) with hx = 0: what fg_grec_math forms for a constant x
                tail[1] = r.inv; tail[2] = r.lns; tail[3] = r.sigma;
                tail[4] = fg_as_double((long long)(r.flags & (FG_G_POW2 | FG_G_DIV)));
                if (!(r.flags & FG_G_POW2)) lin_p2 = 0;
            }
            lin_meta.assign(coord_at.begin(), coord_at.end());
            for (int t = d; t < dp; t++) lin_meta.push_back(n_slots - 1);            // padded term positions read the always-zero slot
            for (int k = 0; k < d; k++) { lin_meta.push_back(cstart[k]); lin_meta.push_back(npri[k]); }
            lin_meta.push_back(0); lin_meta.push_back(0);                            // (readable past the end)
        }
    }
    if (pool.empty()) pool.push_back(0.0);
    // the kernels prefetch two instructions ahead: keep two readable no-ops past each array
    n_ins = (int)ins.size();
    for (int q = 0; q < 2; ++q) { ins.push_back(FgGen::blank(0xffu)); ins_fast.push_back(FgGen::blank(0xffu)); sub.push_back(FgGen::blank(0xffu)); }
    finalized = true;
    return FG_OK;
}

// =================================== C ABI ===================================
extern "C" {

fg_program *fg_program_new(void) { return new fg_program(); }
void fg_program_free(fg_program *p) { delete p; }

int fg_program_data(fg_program *p, const char *name, const double *v, int64_t n) {
    if (!p || (n > 0 && !v) || n < 0) { fg_set_error("fg_program_data: bad argument"); return FG_E_BAD_ARG; }
    p->data.emplace_back(v, v + n);
    p->data_names.emplace_back(name ? name : "");
    return (int)p->data.size() - 1;
}

static int add_dist_stmt(fg_program *p, int kind, const char *addr, int dist, const fg_tok *toks, const int32_t *plen,
                         int n_params, const fg_tok *value, int n_value) {
    if (!p || !addr || dist < 0 || dist >= FG_N_DISTS || n_params < 0) { fg_set_error("bad argument"); return FG_E_BAD_ARG; }
    int want = n_params_of_dist(dist);
    if (want >= 0 && n_params != want) { fg_set_error("wrong number of distribution parameters"); return FG_E_BAD_ARG; }
    if (dist == FG_CATEGORICAL && (n_params < 1 || n_params > 64)) {
        fg_set_error("Categorical takes 1..64 probabilities"); return n_params < 1 ? FG_ERR_INVALID_PROBABILITY : FG_E_LIMIT; }
    FgStmt s;
    s.kind = kind; s.dist = dist; s.addr = addr; s.vtype = vtype_of_dist(dist);
    int off = 0;
    for (int k = 0; k < n_params; k++) {
        int r = p->parse(toks + off, plen[k]);
        if (r < 0) return FG_E_BAD_ARG;
        s.params.push_back(r);
        off += plen[k];
    }
    if (kind == 1) {
        int r = p->parse(value, n_value);
        if (r < 0) return FG_E_BAD_ARG;
        s.value = r;
    }
    if (dist == FG_CATEGORICAL) {       // constructor validation of constant probabilities
        bool allc = true; std::vector<double> pr;
        for (int q : s.params) { allc = allc && p->nodes[q].is_const; pr.push_back(p->nodes[q].cval); }
        if (allc && !fg_categorical_const_valid(pr)) { fg_set_error("Categorical: invalid probability vector"); return FG_ERR_INVALID_PROBABILITY; }
    }
    p->finalized = false;
    if (kind == 0) { s.handle = p->n_samples++; p->stmts.push_back(s); return s.handle; }
    p->n_observes++;
    p->stmts.push_back(s);
    return FG_OK;
}

int fg_program_sample(fg_program *p, const char *addr, int dist, const fg_tok *toks, const int32_t *plen, int n_params) {
    return add_dist_stmt(p, 0, addr, dist, toks, plen, n_params, nullptr, 0);
}
int fg_program_sample_discrete_uniform(fg_program *p, const char *addr, int64_t lo, int64_t hi) {
    if (!p || !addr) { fg_set_error("bad argument"); return FG_E_BAD_ARG; }
    fg_tok t[2] = { { FG_T_CONST, 0, 0, 0, (double)lo }, { FG_T_CONST, 0, 0, 0, (double)hi } };
    const int32_t plen[2] = { 1, 1 };
    const int h = add_dist_stmt(p, 0, addr, FG_DISCRETEUNIFORM, t, plen, 2, nullptr, 0);
    if (h < 0) return h;
    FgStmt &s = p->stmts.back();
    s.exact_bounds = true; s.lo = (long long)lo; s.hi = (long long)hi;
    return h;
}
int fg_program_observe(fg_program *p, const char *addr, int dist, const fg_tok *toks, const int32_t *plen, int n_params,
                       const fg_tok *value, int n_value) {
    if (!value || n_value <= 0) { fg_set_error("observe needs a value expression"); return FG_E_BAD_ARG; }
    return add_dist_stmt(p, 1, addr, dist, toks, plen, n_params, value, n_value);
}
int fg_program_factor(fg_program *p, const fg_tok *toks, int n) {
    if (!p || !toks || n <= 0) { fg_set_error("bad argument"); return FG_E_BAD_ARG; }
    FgStmt s; s.kind = 2; s.dist = -1; s.vtype = FG_F64;
    int r = p->parse(toks, n);
    if (r < 0) return FG_E_BAD_ARG;
    s.value = r;
    p->stmts.push_back(s);
    p->finalized = false;
    return FG_OK;
}
int fg_program_finalize(fg_program *p) { if (!p) return FG_E_BAD_ARG; return p->finalize(); }

#define NEED_FINAL(p) do { if (!(p) || !(p)->finalized) { fg_set_error("program is not finalized"); return FG_E_NOT_FINALIZED; } } while (0)
int fg_program_n_sites(const fg_program *p) { NEED_FINAL(p); return (int)p->sorted_stmt.size(); }
int fg_program_n_f64(const fg_program *p) { NEED_FINAL(p); return (int)p->f64_slot.size(); }
int fg_program_n_observe(const fg_program *p) { NEED_FINAL(p); return p->n_observes; }
int fg_program_n_instructions(const fg_program *p) { NEED_FINAL(p); return p->n_ins; }
int fg_program_n_slots(const fg_program *p) { NEED_FINAL(p); return p->n_slots; }
int fg_program_site_name(const fg_program *p, int j, char *buf, int len) {
    NEED_FINAL(p);
    if (j < 0 || j >= (int)p->sorted_stmt.size()) return FG_ERR_ADDRESS_NOT_FOUND;
    const std::string &a = p->stmts[p->sorted_stmt[j]].addr;
    if (buf && len > 0) { int n = std::min<int>(len - 1, (int)a.size()); std::memcpy(buf, a.data(), n); buf[n] = 0; }
    return (int)a.size() + 1;
}
int fg_program_stream_records(const fg_program *p, int which) {
    if (!p || !p->finalized) return FG_E_STATE;
    if (which == 0) return p->n_gstream;
    if (which == 1) return p->n_sstream;
    if (which == 3) return (int)p->sep_coord.size() > 0 ? (int)p->sep.size() - FG_SEP_MAXREC : 0;   // records of the register-resident trajectories
    if (which == 4) return p->lin_n;                            // rows of the dense-regression table (fg_hmc_lin.hip)
    int k = p->sstream_has_gen ? 2 : (p->sstream_has_lin ? 1 : 0);
    for (int i = 0; i < p->n_gstream; i++) k = std::max(k, (p->gstream[i].flags & FG_G_GEN) ? 2 : ((p->gstream[i].flags & FG_G_LIN) ? 1 : 0));
    return k;
}
int fg_program_site_vtype(const fg_program *p, int j) {
    NEED_FINAL(p);
    if (j < 0 || j >= (int)p->site_vtype.size()) return FG_ERR_ADDRESS_NOT_FOUND;
    return p->site_vtype[j];
}
int fg_program_site_of_handle(const fg_program *p, int h) {
    NEED_FINAL(p);
    if (h < 0 || h >= (int)p->handle_to_sorted.size()) return FG_ERR_ADDRESS_NOT_FOUND;
    return p->handle_to_sorted[h];
}
int fg_program_f64_site(const fg_program *p, int k) {
    NEED_FINAL(p);
    if (k < 0 || k >= (int)p->f64_slot.size()) return FG_ERR_ADDRESS_NOT_FOUND;
    return p->f64_slot[k];
}
int fg_program_dep_count(const fg_program *p, int k) {
    NEED_FINAL(p);
    if (k < 0 || k >= (int)p->f64_slot.size()) return FG_ERR_ADDRESS_NOT_FOUND;
    return p->sub_off[k + 1] - p->sub_off[k];
}

}  // extern "C"
