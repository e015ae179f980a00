// fg_dsl.cpp -- front-end for the `prob!`-subset model language of the reference's playground
// (grammar: crates/fugue-wasm/src/dsl.rs:10-35; parser :338-636; evaluation :697-800; data binding
// :1066-1106), emitting a site program through the C ABI instead of interpreting into `Model<f64>`.
//
//     let p <- sample(addr!("p"), Beta(2.0, 2.0));
//     let mu = 2.0 * p - 1.0;
//     for i in 0..y.len() { observe(addr!("y", i), Normal(mu, 0.8), y[i]); }
//     factor(-0.5 * mu * mu);
//     pure(p)
//
// `Dist::new(..)` and a trailing `.unwrap()` are accepted sugar; data comes from a JSON object of
// number arrays or a bare array (bound to `data`); addresses are built with the `addr!` encoding
// (src/core/address.rs:189-223), so they are byte-identical to compiled Rust.
//
// The reference evaluates the program at run time, once per model execution.  Here it is evaluated
// ONCE, symbolically: numbers, integers, booleans and data arrays are folded at build time (integer
// `+ - *` stay integers, dsl.rs:752-759), sampled values become expressions, `for` loops are unrolled
// (their bounds must be build-time integers).  Static errors (syntax, unknown names, arities) are
// reported with a line number like the reference's; an out-of-bounds data index becomes NaN plus a
// warning (dsl.rs:715-722); invalid distribution parameters need no special casing because every
// log-density already returns -inf for them (the reference maps them to factor(-inf), :961-977).
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "fg_program.h"

namespace {

struct DslError { std::string msg; };

// ---------------------------------------------------------------- lexer (dsl.rs:150-262)
enum class T { Ident, Num, Str, Sym, Eof };
struct Tok { T t; std::string s; double v = 0.0; bool dotted = false; size_t byte = 0; };

struct Lexer {
    std::vector<Tok> toks; size_t pos = 0; std::vector<size_t> line_starts;
    explicit Lexer(const std::string &src) {
        static const char *SYMS[] = { "<-", "::", "..", "(", ")", "{", "}", "[", "]", ",", ";", "+", "-", "*", "/", "=", ".", "!" };
        line_starts.push_back(0);
        for (size_t i = 0; i < src.size(); i++) if (src[i] == '\n') line_starts.push_back(i + 1);
        size_t i = 0;
        while (i < src.size()) {
            const char c = src[i];
            if (isspace((unsigned char)c)) { i++; continue; }
            if (c == '/' && i + 1 < src.size() && src[i + 1] == '/') { while (i < src.size() && src[i] != '\n') i++; continue; }
            if (c == '"') {
                size_t j = i + 1;
                while (j < src.size() && src[j] != '"') j++;
                if (j >= src.size()) throw DslError{"unterminated string at byte " + std::to_string(i)};
                toks.push_back({T::Str, src.substr(i + 1, j - i - 1), 0, false, i});
                i = j + 1; continue;
            }
            if (isdigit((unsigned char)c)) {
                const size_t start = i; bool dot = false;
                while (i < src.size()) {
                    const char d = src[i];
                    if (isdigit((unsigned char)d)) i++;
                    else if (d == '.' && !dot && i + 1 < src.size() && isdigit((unsigned char)src[i + 1])) { dot = true; i++; }   // `0..n` lexes as 0 `..` n
                    else if ((d == 'e' || d == 'E') && i + 1 < src.size() && (isdigit((unsigned char)src[i + 1]) || src[i + 1] == '-' || src[i + 1] == '+')) {
                        dot = true; i += 2; while (i < src.size() && isdigit((unsigned char)src[i])) i++; break; }
                    else break;
                }
                toks.push_back({T::Num, src.substr(start, i - start), std::strtod(src.substr(start, i - start).c_str(), nullptr), dot, start});
                continue;
            }
            if (isalpha((unsigned char)c) || c == '_') {
                const size_t start = i;
                while (i < src.size() && (isalnum((unsigned char)src[i]) || src[i] == '_')) i++;
                toks.push_back({T::Ident, src.substr(start, i - start), 0, false, start});
                continue;
            }
            bool matched = false;
            for (const char *s : SYMS) if (src.compare(i, std::strlen(s), s) == 0) { toks.push_back({T::Sym, s, 0, false, i}); i += std::strlen(s); matched = true; break; }
            if (!matched) throw DslError{std::string("unexpected character `") + c + "` at byte " + std::to_string(i)};
        }
        toks.push_back({T::Eof, "", 0, false, src.size()});
    }
    const Tok &peek() const { return toks[pos]; }
    const Tok &peek2() const { return toks[std::min(pos + 1, toks.size() - 1)]; }
    Tok next() { Tok t = toks[pos]; if (pos + 1 < toks.size()) pos++; return t; }
    bool is_sym(const char *s) const { return peek().t == T::Sym && peek().s == s; }
    size_t line_of(size_t byte) const { size_t l = 0; while (l + 1 < line_starts.size() && line_starts[l + 1] <= byte) l++; return l + 1; }
    [[noreturn]] void fail(const std::string &what) const {
        const Tok &t = peek();
        std::string found = t.t == T::Eof ? "end of input" : (t.t == T::Str ? "\"" + t.s + "\"" : "`" + t.s + "`");
        throw DslError{"line " + std::to_string(line_of(t.byte)) + ": expected " + what + ", found " + found};
    }
    void expect_sym(const char *s) { if (is_sym(s)) next(); else fail(std::string("`") + s + "`"); }
    std::string expect_ident() { if (peek().t != T::Ident) fail("an identifier"); return next().s; }
    bool eat_kw(const char *kw) { if (peek().t == T::Ident && peek().s == kw) { next(); return true; } return false; }
};

// ---------------------------------------------------------------- AST (dsl.rs:91-145)
struct Expr;
using EP = std::shared_ptr<Expr>;
struct Expr { enum K { Num, Int, Bool, Var, Index, Len, Neg, Bin, Call } k; double num = 0; long long i = 0; bool b = false; std::string name; char op = 0;
              std::vector<EP> args; };
struct Stmt;
using SP = std::shared_ptr<Stmt>;
struct Stmt { enum K { SampleLet, LetExpr, Observe, Factor, For } k; std::string name, addr_name, dist; EP addr_index, expr, lo, hi; std::vector<EP> dist_args;
              std::vector<SP> body; size_t line = 0; };

const struct { const char *name; int arity; int op; } MATH[] = {
    {"exp", 1, FG_T_EXP}, {"ln", 1, FG_T_LN}, {"log", 1, FG_T_LN}, {"sqrt", 1, FG_T_SQRT}, {"abs", 1, FG_T_ABS}, {"floor", 1, FG_T_FLOOR},
    {"sin", 1, FG_T_SIN}, {"cos", 1, FG_T_COS}, {"tanh", 1, FG_T_TANH}, {"pow", 2, FG_T_POW}, {"min", 2, FG_T_MIN}, {"max", 2, FG_T_MAX} };
const struct { const char *name; int kind, lo, hi; } DISTS[] = {    // dist_arity, dsl.rs:807-820
    {"Normal", FG_NORMAL, 2, 2}, {"Uniform", FG_UNIFORM, 2, 2}, {"LogNormal", FG_LOGNORMAL, 2, 2}, {"Beta", FG_BETA, 2, 2}, {"Gamma", FG_GAMMA, 2, 2},
    {"InverseGamma", FG_INVERSEGAMMA, 2, 2}, {"Cauchy", FG_CAUCHY, 2, 2}, {"Laplace", FG_LAPLACE, 2, 2}, {"Weibull", FG_WEIBULL, 2, 2},
    {"Binomial", FG_BINOMIAL, 2, 2}, {"DiscreteUniform", FG_DISCRETEUNIFORM, 2, 2}, {"Exponential", FG_EXPONENTIAL, 1, 1}, {"Poisson", FG_POISSON, 1, 1},
    {"Bernoulli", FG_BERNOULLI, 1, 1}, {"ChiSquared", FG_CHISQUARED, 1, 1}, {"StudentT", FG_STUDENTT, 3, 3}, {"Categorical", FG_CATEGORICAL, 1, 64} };

EP parse_expr(Lexer &lx);
EP mk(Expr::K k) { auto e = std::make_shared<Expr>(); e->k = k; return e; }
struct Depth { int &d; explicit Depth(int &x) : d(x) { if (++d > 200) throw DslError{"expression or block nesting deeper than 200"}; } ~Depth() { --d; } };
thread_local int g_depth = 0;      // parser recursion guard: per thread, so concurrent fg_dsl_compile calls do not share it
EP parse_primary(Lexer &lx) {
    Depth guard(g_depth);
    const Tok t = lx.peek();
    if (t.t == T::Num) { lx.next(); EP e = mk(t.dotted ? Expr::Num : Expr::Int); e->num = t.v; e->i = (long long)t.v; return e; }
    if (t.t == T::Sym && t.s == "(") { lx.next(); EP e = parse_expr(lx); lx.expect_sym(")"); return e; }
    if (t.t == T::Ident) {
        lx.next();
        if (t.s == "true" || t.s == "false") { EP e = mk(Expr::Bool); e->b = t.s == "true"; return e; }
        if (lx.is_sym("(")) {
            int arity = -1;
            for (auto &m : MATH) if (t.s == m.name) arity = m.arity;
            if (arity < 0) throw DslError{"unknown function `" + t.s + "`"};
            lx.next();
            EP e = mk(Expr::Call); e->name = t.s;
            if (!lx.is_sym(")")) for (;;) { e->args.push_back(parse_expr(lx)); if (lx.is_sym(",")) lx.next(); else break; }
            lx.expect_sym(")");
            if ((int)e->args.size() != arity) throw DslError{"`" + t.s + "` takes " + std::to_string(arity) + " argument(s)"};
            return e;
        }
        EP e = mk(Expr::Var); e->name = t.s; return e;
    }
    lx.fail("an expression");
}
EP parse_postfix(Lexer &lx) {
    EP e = parse_primary(lx);
    for (;;) {
        if (lx.is_sym("[")) { lx.next(); EP idx = parse_expr(lx); lx.expect_sym("]"); EP n = mk(Expr::Index); n->args = {e, idx}; e = n; }
        else if (lx.is_sym(".") && lx.peek2().t == T::Ident && lx.peek2().s == "len") {
            lx.next(); lx.next(); lx.expect_sym("("); lx.expect_sym(")"); EP n = mk(Expr::Len); n->args = {e}; e = n; }
        else return e;
    }
}
EP parse_unary(Lexer &lx) { if (lx.is_sym("-")) { lx.next(); EP n = mk(Expr::Neg); n->args = {parse_unary(lx)}; return n; } return parse_postfix(lx); }
EP parse_mul(Lexer &lx) { EP l = parse_unary(lx); while (lx.is_sym("*") || lx.is_sym("/")) { char op = lx.next().s[0]; EP r = parse_unary(lx); EP n = mk(Expr::Bin); n->op = op; n->args = {l, r}; l = n; } return l; }
EP parse_expr(Lexer &lx) { EP l = parse_mul(lx); while (lx.is_sym("+") || lx.is_sym("-")) { char op = lx.next().s[0]; EP r = parse_mul(lx); EP n = mk(Expr::Bin); n->op = op; n->args = {l, r}; l = n; } return l; }

void parse_addr(Lexer &lx, Stmt &s) {
    if (!lx.eat_kw("addr")) lx.fail("`addr!(..)`");
    lx.expect_sym("!"); lx.expect_sym("(");
    if (lx.peek().t != T::Str) lx.fail("a string literal address name");
    s.addr_name = lx.next().s;
    if (lx.is_sym(",")) { lx.next(); s.addr_index = parse_expr(lx); }
    lx.expect_sym(")");
}
void parse_dist(Lexer &lx, Stmt &s) {
    s.dist = lx.expect_ident();
    if (lx.is_sym("::")) { lx.next(); const std::string m = lx.expect_ident(); if (m != "new") throw DslError{"unknown distribution constructor `" + s.dist + "::" + m + "`"}; }
    lx.expect_sym("(");
    if (!lx.is_sym(")")) for (;;) { s.dist_args.push_back(parse_expr(lx)); if (lx.is_sym(",")) lx.next(); else break; }
    lx.expect_sym(")");
    if (lx.is_sym(".") && lx.peek2().t == T::Ident && lx.peek2().s == "unwrap") { lx.next(); lx.next(); lx.expect_sym("("); lx.expect_sym(")"); }
    for (auto &d : DISTS) if (s.dist == d.name) {
        const int n = (int)s.dist_args.size();
        if (n < d.lo || n > d.hi) throw DslError{"`" + s.dist + "` takes " + (d.lo == d.hi ? std::to_string(d.lo) : std::to_string(d.lo) + "..=" + std::to_string(d.hi)) +
                                                 " argument(s), got " + std::to_string(n)};
        return;
    }
    throw DslError{"unknown distribution `" + s.dist + "`"};
}
SP parse_stmt(Lexer &lx) {
    Depth guard(g_depth);
    auto s = std::make_shared<Stmt>();
    s->line = lx.line_of(lx.peek().byte);
    if (lx.eat_kw("let")) {
        s->name = lx.expect_ident();
        if (lx.is_sym("<-")) {
            lx.next();
            if (!lx.eat_kw("sample")) lx.fail("`sample`");
            lx.expect_sym("("); parse_addr(lx, *s); lx.expect_sym(","); parse_dist(lx, *s); lx.expect_sym(")"); lx.expect_sym(";");
            s->k = Stmt::SampleLet; return s;
        }
        lx.expect_sym("="); s->expr = parse_expr(lx); lx.expect_sym(";"); s->k = Stmt::LetExpr; return s;
    }
    if (lx.eat_kw("observe")) { lx.expect_sym("("); parse_addr(lx, *s); lx.expect_sym(","); parse_dist(lx, *s); lx.expect_sym(","); s->expr = parse_expr(lx);
                                lx.expect_sym(")"); lx.expect_sym(";"); s->k = Stmt::Observe; return s; }
    if (lx.eat_kw("factor")) { lx.expect_sym("("); s->expr = parse_expr(lx); lx.expect_sym(")"); lx.expect_sym(";"); s->k = Stmt::Factor; return s; }
    if (lx.eat_kw("for")) {
        s->name = lx.expect_ident();
        if (!lx.eat_kw("in")) lx.fail("`in`");
        s->lo = parse_expr(lx); lx.expect_sym(".."); s->hi = parse_expr(lx); lx.expect_sym("{");
        while (!lx.is_sym("}")) { if (lx.peek().t == T::Eof) lx.fail("`}`"); s->body.push_back(parse_stmt(lx)); }
        lx.next(); s->k = Stmt::For; return s;
    }
    lx.fail("a statement (`let`, `observe`, `factor`, `for`, or `pure`)");
}

// ---------------------------------------------------------------- values (dsl.rs:46-88) + symbolic
struct Val { enum K { F64, INT, BOOL, ARR, SYM } k = F64; double f = 0; long long i = 0; bool b = false; std::shared_ptr<std::vector<double>> arr; std::vector<fg_tok> sym;
    bool numeric() const { return k == F64 || k == INT || k == BOOL; }
    double as_f64() const { return k == F64 ? f : k == INT ? (double)i : k == BOOL ? (b ? 1.0 : 0.0) : NAN; }
    bool as_index(long long &out) const { if (k == INT) { out = i; return true; } if (k == F64 && std::isfinite(f) && f == std::floor(f)) { out = (long long)f; return true; } return false; }
    std::vector<fg_tok> toks() const { if (k == SYM) return sym; return { fg_tok{FG_T_CONST, 0, 0, 0, as_f64()} }; } };
Val vf(double x) { Val v; v.k = Val::F64; v.f = x; return v; }
Val vi(long long x) { Val v; v.k = Val::INT; v.i = x; return v; }
Val vsym(std::vector<fg_tok> t) { Val v; v.k = Val::SYM; v.sym = std::move(t); return v; }

struct Env { std::map<std::string, Val> vars; std::vector<std::string> *warnings; void warn(const std::string &m) { if (warnings->size() < 64) warnings->push_back(m); } };

Val eval(const EP &e, Env &env) {
    switch (e->k) {
    case Expr::Num: return vf(e->num);
    case Expr::Int: return vi(e->i);
    case Expr::Bool: { Val v; v.k = Val::BOOL; v.b = e->b; return v; }
    case Expr::Var: { auto it = env.vars.find(e->name); if (it == env.vars.end()) throw DslError{"unknown variable `" + e->name + "`"}; return it->second; }
    case Expr::Len: { Val a = eval(e->args[0], env); if (a.k == Val::ARR) return vi((long long)a.arr->size()); env.warn("`.len()` on a non-array value"); return vi(0); }
    case Expr::Index: {
        Val a = eval(e->args[0], env), ix = eval(e->args[1], env);
        if (a.k != Val::ARR) { env.warn("indexing a non-array value"); return vf(NAN); }
        if (ix.k == Val::SYM) {       // data[z] with a sampled index: a first-order select over the array
            if (a.arr->size() > 64) throw DslError{"indexing an array of more than 64 elements with a sampled value is not supported"};
            std::vector<fg_tok> t = ix.sym;
            for (double v : *a.arr) t.push_back(fg_tok{FG_T_CONST, 0, 0, 0, v});
            t.push_back(fg_tok{FG_T_SELECT, (int)a.arr->size(), 0, 0, 0.0});
            return vsym(t);
        }
        long long i;
        if (!ix.as_index(i)) { env.warn("indexing a non-array value"); return vf(NAN); }
        if (i >= 0 && (size_t)i < a.arr->size()) return vf((*a.arr)[(size_t)i]);
        env.warn("index " + std::to_string(i) + " out of bounds (len " + std::to_string(a.arr->size()) + ")");
        return vf(NAN);
    }
    case Expr::Neg: { Val a = eval(e->args[0], env); if (a.k == Val::INT) return vi(-a.i); if (a.k == Val::SYM) { auto t = a.sym; t.push_back(fg_tok{FG_T_NEG, 0, 0, 0, 0}); return vsym(t); } return vf(-a.as_f64()); }
    case Expr::Bin: {
        Val a = eval(e->args[0], env), b = eval(e->args[1], env);
        if (a.k == Val::INT && b.k == Val::INT && e->op != '/') return vi(e->op == '+' ? a.i + b.i : e->op == '-' ? a.i - b.i : a.i * b.i);   // dsl.rs:752-759
        if (a.k == Val::SYM || b.k == Val::SYM) {
            std::vector<fg_tok> t = a.toks(), tb = b.toks();
            t.insert(t.end(), tb.begin(), tb.end());
            t.push_back(fg_tok{e->op == '+' ? FG_T_ADD : e->op == '-' ? FG_T_SUB : e->op == '*' ? FG_T_MUL : FG_T_DIV, 0, 0, 0, 0});
            return vsym(t);
        }
        const double x = a.as_f64(), y = b.as_f64();
        return vf(e->op == '+' ? x + y : e->op == '-' ? x - y : e->op == '*' ? x * y : x / y);
    }
    case Expr::Call: {
        std::vector<Val> a; bool sym = false;
        for (auto &x : e->args) { a.push_back(eval(x, env)); sym = sym || a.back().k == Val::SYM; }
        int op = 0; for (auto &m : MATH) if (e->name == m.name) op = m.op;
        if (sym) { std::vector<fg_tok> t; for (auto &v : a) { auto tv = v.toks(); t.insert(t.end(), tv.begin(), tv.end()); } t.push_back(fg_tok{op, 0, 0, 0, 0}); return vsym(t); }
        const double x = a[0].as_f64(), y = a.size() > 1 ? a[1].as_f64() : 0.0;
        switch (op) { case FG_T_EXP: return vf(std::exp(x)); case FG_T_LN: return vf(std::log(x)); case FG_T_SQRT: return vf(std::sqrt(x)); case FG_T_ABS: return vf(std::fabs(x));
                      case FG_T_FLOOR: return vf(std::floor(x)); case FG_T_SIN: return vf(std::sin(x)); case FG_T_COS: return vf(std::cos(x)); case FG_T_TANH: return vf(std::tanh(x));
                      case FG_T_POW: return vf(std::pow(x, y)); case FG_T_MIN: return vf(std::fmin(x, y)); default: return vf(std::fmax(x, y)); }
    }
    }
    return vf(NAN);
}

// addr!(name) / addr!(name, i): src/core/address.rs:189-223
std::string esc(const std::string &s) { if (s.find('\\') == std::string::npos && s.find('#') == std::string::npos) return s;
    std::string o; for (char c : s) { if (c == '\\') o += "\\\\"; else if (c == '#') o += "\\#"; else o += c; } return o; }
std::string make_addr(const Stmt &s, Env &env) {
    if (!s.addr_index) return esc(s.addr_name);
    Val v = eval(s.addr_index, env);
    long long i;
    if (v.numeric() && v.as_index(i)) return esc(s.addr_name) + "#" + esc(std::to_string(i));
    throw DslError{"line " + std::to_string(s.line) + ": address index of `" + s.addr_name + "` must be a build-time integer"};
}

struct Builder {
    fg_program *p; std::vector<std::string> warnings;
    long long n_emitted = 0;      // statements emitted so far: the model is unrolled at build time, so bound it
    void count() { if (++n_emitted > (1LL << 20)) throw DslError{"model unrolls to more than 1048576 statements"}; }
    // Parameters that are build-time numbers and fail the distribution's constructor: the reference keeps the
    // model alive and kills the weight (dsl.rs:961-977 sample, :1002-1006 observe).  Returns the message or "".
    std::string dist_toks(const Stmt &s, Env &env, int &kind, std::vector<fg_tok> &toks, std::vector<int32_t> &lens) {
        for (auto &d : DISTS) if (s.dist == d.name) kind = d.kind;
        bool all_num = true; std::vector<double> a;
        for (auto &e : s.dist_args) { Val v = eval(e, env); if (v.k == Val::ARR) throw DslError{"line " + std::to_string(s.line) + ": array used as a distribution parameter"};
            all_num = all_num && v.numeric(); a.push_back(v.as_f64());
            auto t = v.toks(); lens.push_back((int32_t)t.size()); toks.insert(toks.end(), t.begin(), t.end()); }
        if (!all_num) return "";
        if (kind == FG_BINOMIAL && (a[0] < 0.0 || a[0] != std::floor(a[0]))) { char b[96]; snprintf(b, sizeof b, "Binomial n must be a non-negative integer, got %g", a[0]); return b; }   // dsl.rs:838-847
        if (kind == FG_CATEGORICAL) return fg_categorical_const_valid(a) ? "" : "Categorical: invalid probability vector";
        if (kind == FG_DISCRETEUNIFORM) return fg_f2i_sat(a[0]) <= fg_f2i_sat(a[1]) ? "" : "DiscreteUniform: low must not exceed high";
        double h[8];
        return fg_hoist((uint32_t)kind, a[0], a.size() > 1 ? a[1] : 0.0, a.size() > 2 ? a[2] : 0.0, h) ? "" : "invalid parameters for " + s.dist;
    }
    void neg_inf_factor() { const fg_tok t{FG_T_CONST, 0, 0, 0, -INFINITY}; if (fg_program_factor(p, &t, 1)) throw DslError{fg_last_error()}; }
    void run(const std::vector<SP> &stmts, Env &env) {
        for (auto &sp : stmts) {
            const Stmt &s = *sp;
            switch (s.k) {
            case Stmt::LetExpr: count(); env.vars[s.name] = eval(s.expr, env); break;
            case Stmt::Factor: { count(); Val v = eval(s.expr, env); auto t = v.toks();
                if (v.numeric() && std::isnan(v.as_f64())) t = { fg_tok{FG_T_CONST, 0, 0, 0, -INFINITY} };      // NaN -> -inf, dsl.rs:911-913
                if (fg_program_factor(p, t.data(), (int)t.size())) throw DslError{fg_last_error()}; break; }
            case Stmt::SampleLet: case Stmt::Observe: {
                count();
                int kind = 0; std::vector<fg_tok> toks; std::vector<int32_t> lens;
                const std::string a = make_addr(s, env);
                const std::string bad = dist_toks(s, env, kind, toks, lens);
                if (!bad.empty()) {
                    if (s.k == Stmt::SampleLet) {       // placeholder Normal(0,1) site + factor(-inf)
                        env.warn("sample `" + s.name + "`: " + bad);
                        const fg_tok pt[2] = { {FG_T_CONST, 0, 0, 0, 0.0}, {FG_T_CONST, 0, 0, 0, 1.0} }; const int32_t pl[2] = {1, 1};
                        const int h = fg_program_sample(p, a.c_str(), FG_NORMAL, pt, pl, 2);
                        if (h < 0) throw DslError{fg_last_error()};
                        env.vars[s.name] = vsym({ fg_tok{FG_T_SITE, h, 0, 0, 0.0} });
                    } else env.warn("observe at `" + a + "`: " + bad);
                    neg_inf_factor();
                    break;
                }
                if (s.k == Stmt::SampleLet) {
                    const int h = fg_program_sample(p, a.c_str(), kind, toks.data(), lens.data(), (int)lens.size());
                    if (h < 0) throw DslError{"line " + std::to_string(s.line) + ": sample `" + s.name + "`: " + fg_last_error()};
                    env.vars[s.name] = vsym({ fg_tok{FG_T_SITE, h, 0, 0, 0.0} });
                } else {
                    Val v = eval(s.expr, env);
                    if (v.k == Val::ARR) throw DslError{"line " + std::to_string(s.line) + ": array used as an observed value"};
                    auto vt = v.toks();
                    if (fg_program_observe(p, a.c_str(), kind, toks.data(), lens.data(), (int)lens.size(), vt.data(), (int)vt.size()))
                        throw DslError{"line " + std::to_string(s.line) + ": observe at `" + a + "`: " + fg_last_error()};
                }
                break; }
            case Stmt::For: {
                Val lo = eval(s.lo, env), hi = eval(s.hi, env);
                long long l = 0, h = 0;
                if (lo.k == Val::SYM || hi.k == Val::SYM) throw DslError{"line " + std::to_string(s.line) + ": `for` bounds must not depend on sampled values"};
                if (!lo.as_index(l)) { env.warn("`for` lower bound is not an integer"); l = 0; }      // dsl.rs:1010-1017
                if (!hi.as_index(h)) { env.warn("`for` upper bound is not an integer"); h = 0; }
                const bool had = env.vars.count(s.name) != 0; const Val saved = had ? env.vars[s.name] : Val();
                for (long long i = l; i < h; i++) { count(); env.vars[s.name] = vi(i); run(s.body, env); }
                if (had) env.vars[s.name] = saved; else env.vars.erase(s.name);
                break; }
            }
        }
    }
};

// ---------------------------------------------------------------- data JSON (dsl.rs:1066-1106)
struct Json { const std::string &s; size_t i = 0;
    void ws() { while (i < s.size() && isspace((unsigned char)s[i])) i++; }
    [[noreturn]] void bad(const std::string &m) { throw DslError{"data is not valid JSON: " + m + " at byte " + std::to_string(i)}; }
    std::string str() { if (s[i] != '"') bad("expected string"); size_t j = ++i; while (j < s.size() && s[j] != '"') { if (s[j] == '\\') j++; j++; } if (j >= s.size()) bad("unterminated string");
        std::string o = s.substr(i, j - i); i = j + 1; return o; }
    std::vector<double> arr() { if (s[i] != '[') throw DslError{"data arrays must be JSON arrays"}; i++; std::vector<double> v; ws();
        if (i < s.size() && s[i] == ']') { i++; return v; }
        for (;;) { ws();
            if (s.compare(i, 4, "true") == 0) { v.push_back(1.0); i += 4; } else if (s.compare(i, 5, "false") == 0) { v.push_back(0.0); i += 5; }
            else { char *end = nullptr; const double x = std::strtod(s.c_str() + i, &end); if (end == s.c_str() + i) throw DslError{"data arrays must hold numbers or booleans"}; v.push_back(x); i = (size_t)(end - s.c_str()); }
            ws(); if (i < s.size() && s[i] == ',') { i++; continue; } if (i < s.size() && s[i] == ']') { i++; return v; } bad("expected `,` or `]`"); } } };
void bind_data(const std::string &json, std::map<std::string, Val> &vars) {
    Json j{json}; j.ws();
    if (j.i >= json.size() || json.compare(j.i, 4, "null") == 0) return;
    auto mk = [](std::vector<double> v) { Val a; a.k = Val::ARR; a.arr = std::make_shared<std::vector<double>>(std::move(v)); return a; };
    if (json[j.i] == '[') { vars["data"] = mk(j.arr()); return; }
    if (json[j.i] != '{') throw DslError{"data must be a JSON array or object of arrays"};
    j.i++; j.ws();
    if (j.i < json.size() && json[j.i] == '}') return;
    for (;;) { j.ws(); const std::string k = j.str(); j.ws(); if (j.i >= json.size() || json[j.i] != ':') j.bad("expected `:`"); j.i++; j.ws(); vars[k] = mk(j.arr()); j.ws();
        if (j.i < json.size() && json[j.i] == ',') { j.i++; continue; } if (j.i < json.size() && json[j.i] == '}') return; j.bad("expected `,` or `}`"); }
}

}  // namespace

extern "C" {

// CompiledModel::compile (dsl.rs:1062-1120): parse + bind data + build the site program.  Returns a FINALIZED
// program, or NULL with the message (carrying a line number for syntax errors) in fg_last_error().
fg_program *fg_dsl_compile(const char *source_utf8, const char *data_json_utf8) {
    if (!source_utf8) { fg_set_error("fg_dsl_compile: null source"); return nullptr; }
    fg_program *p = fg_program_new();
    g_depth = 0;
    try {
        Lexer lx(source_utf8);
        std::vector<SP> stmts; EP ret;
        for (;;) {
            if (lx.eat_kw("pure")) { lx.expect_sym("("); ret = parse_expr(lx); lx.expect_sym(")"); if (lx.is_sym(";")) lx.next();
                if (lx.peek().t != T::Eof) lx.fail("end of input after `pure(..)`"); break; }
            if (lx.peek().t == T::Eof) throw DslError{"model must end with `pure(<expr>)`"};
            stmts.push_back(parse_stmt(lx));
        }
        Builder b{p, {}};
        Env env; env.warnings = &b.warnings;
        bind_data(data_json_utf8 ? data_json_utf8 : "", env.vars);
        b.run(stmts, env);
        (void)eval(ret, env);                       // validates the names used by pure(..)
        p->dsl_warnings = b.warnings;
        const int rc = fg_program_finalize(p);
        if (rc) { fg_program_free(p); return nullptr; }
        return p;
    } catch (const DslError &e) {
        fg_set_error(e.msg);
        fg_program_free(p);
        return nullptr;
    }
}
int fg_dsl_warning_count(const fg_program *p) { return p ? (int)p->dsl_warnings.size() : 0; }
const char *fg_dsl_warning(const fg_program *p, int i) { return (p && i >= 0 && i < (int)p->dsl_warnings.size()) ? p->dsl_warnings[i].c_str() : ""; }

}  // extern "C"
