// fugue_amd.hpp -- header-only C++17 mirror of Fugue's model + inference surface for the hot path,
// written above the C ABI (fugue_amd.h).  The reference is Rust; this image has no Rust toolchain, so
// the host side that a Fugue user would touch is mirrored here with the same names, argument
// meaning and error behaviour:
//
//   addr!("x") / addr!("x", i)                    -> fugue::addr("x") / fugue::addr("x", i)      (src/core/address.rs:189-257)
//   Normal::new(mu, sigma).unwrap() ...           -> fugue::Normal(mu, sigma) ... (throws FugueError with the
//                                                    reference ErrorCode on invalid constant parameters)
//   sample / observe / factor / pure / guard      -> same names, returning Model<A>               (src/core/model.rs:144-716)
//   ModelExt::{bind, map, and_then}, zip, sequence_vec, traverse_vec, plate!                    (model.rs:447-716, macros/mod.rs:72-90)
//   hmc_chain / adaptive_mcmc_chain / adaptive_smc + HMCConfig / SMCConfig / SiteProposal     (src/inference/{hmc,mh,smc}.rs)
//
// A Model<A> here is run ONCE with symbolic values (Expr) to record a fixed-structure site program;
// the many-chain engine then evaluates that program on the GPU.  Values flowing through `bind`
// continuations are Expr, so `Normal(mu, 1.0)` inside a continuation describes itself symbolically.
// Models whose structure depends on a sampled value cannot be expressed (Expr has no conversion to
// bool) -- the engine refuses structure-varying models.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "fugue_amd.h"

namespace fugue {

// ---- errors: ErrorCode numbers of src/error.rs:40-59 ---------------------------------------
enum class ErrorCode : int { InvalidMean = 100, InvalidVariance = 101, InvalidProbability = 102, InvalidRange = 103,
                             InvalidShape = 104, InvalidRate = 105, InvalidCount = 106, AddressConflict = 301,
                             UnexpectedModelStructure = 302, TraceAddressNotFound = 500, TypeMismatch = 600, Engine = -1 };
struct FugueError : std::runtime_error {
    int code;
    FugueError(const std::string &m, int c) : std::runtime_error(m + " (ErrorCode " + std::to_string(c) + ")"), code(c) {}
    FugueError(const std::string &m, ErrorCode c) : FugueError(m, (int)c) {}
};
inline void check(int rc, const char *what) {
    if (rc != 0) throw FugueError(std::string(what) + ": " + fg_last_error(), rc);
}

// ---- addresses ---------------------------------------------------------------------------------
using Address = std::string;
inline std::string escape_addr_segment(const std::string &s) {
    if (s.find('\\') == std::string::npos && s.find('#') == std::string::npos) return s;
    std::string o;
    for (char c : s) { if (c == '\\') o += "\\\\"; else if (c == '#') o += "\\#"; else o += c; }
    return o;
}
inline Address addr(const std::string &name) { return escape_addr_segment(name); }
template <class I> Address addr(const std::string &name, I index) {
    std::string idx;
    if constexpr (std::is_arithmetic_v<I>) idx = std::to_string(index); else idx = std::string(index);
    return escape_addr_segment(name) + "#" + escape_addr_segment(idx);
}

// ---- symbolic f64 ---------------------------------------------------------------------------------
struct Node { int op; int a = 0, b = 0; double imm = 0.0; std::vector<std::shared_ptr<Node>> kids; };
class Expr {
  public:
    std::shared_ptr<Node> n;
    Expr() : Expr(0.0) {}
    Expr(double v) : n(std::make_shared<Node>(Node{FG_T_CONST, 0, 0, v, {}})) {}
    Expr(int v) : Expr((double)v) {}
    Expr(bool v) : Expr(v ? 1.0 : 0.0) {}
    explicit Expr(std::shared_ptr<Node> p) : n(std::move(p)) {}
    static Expr site(int handle) { return Expr(std::make_shared<Node>(Node{FG_T_SITE, handle, 0, 0.0, {}})); }
    static Expr un(int op, const Expr &x) { return Expr(std::make_shared<Node>(Node{op, 0, 0, 0.0, {x.n}})); }
    static Expr bin(int op, const Expr &x, const Expr &y) { return Expr(std::make_shared<Node>(Node{op, 0, 0, 0.0, {x.n, y.n}})); }
    bool is_const() const { return n->op == FG_T_CONST; }
    double value() const { return n->imm; }
    Expr clamp(const Expr &lo, const Expr &hi) const { return Expr(std::make_shared<Node>(Node{FG_T_CLAMP, 0, 0, 0.0, {n, lo.n, hi.n}})); }
    Expr exp() const { return un(FG_T_EXP, *this); }
    Expr ln() const { return un(FG_T_LN, *this); }
    Expr sqrt() const { return un(FG_T_SQRT, *this); }
    Expr abs() const { return un(FG_T_ABS, *this); }
    Expr powf(const Expr &y) const { return bin(FG_T_POW, *this, y); }
    Expr min(const Expr &y) const { return bin(FG_T_MIN, *this, y); }
    Expr max(const Expr &y) const { return bin(FG_T_MAX, *this, y); }
    void postfix(std::vector<fg_tok> &out) const { emit(n.get(), out); }
  private:
    static void emit(const Node *p, std::vector<fg_tok> &out) {
        for (auto &k : p->kids) emit(k.get(), out);
        out.push_back(fg_tok{p->op, p->a, p->b, 0, p->imm});
    }
};
inline Expr operator+(const Expr &x, const Expr &y) { return Expr::bin(FG_T_ADD, x, y); }
inline Expr operator-(const Expr &x, const Expr &y) { return Expr::bin(FG_T_SUB, x, y); }
inline Expr operator*(const Expr &x, const Expr &y) { return Expr::bin(FG_T_MUL, x, y); }
inline Expr operator/(const Expr &x, const Expr &y) { return Expr::bin(FG_T_DIV, x, y); }
inline Expr operator-(const Expr &x) { return Expr::un(FG_T_NEG, x); }
/// options[index] for a sampled integer index: the first-order form of `if z == 0 {..} else {..}`
inline Expr select(const Expr &index, const std::vector<Expr> &options) {
    auto p = std::make_shared<Node>(Node{FG_T_SELECT, (int)options.size(), 0, 0.0, {index.n}});
    for (auto &o : options) p->kids.push_back(o.n);
    return Expr(p);
}
struct Unit {};

// ---- distributions (constructor validation: distribution.rs:133-152 ... 1842-1853) ------------------
struct Dist { int kind; std::vector<Expr> params; };
namespace detail {
inline void need(bool bad, const char *msg, ErrorCode c) { if (bad) throw FugueError(msg, c); }
inline bool cbad_nonfinite(const Expr &e) { return e.is_const() && !std::isfinite(e.value()); }
inline bool cbad_nonpos(const Expr &e) { return e.is_const() && !(e.value() > 0.0 && std::isfinite(e.value())); }
inline bool cbad_prob(const Expr &e) { return e.is_const() && !(std::isfinite(e.value()) && e.value() >= 0.0 && e.value() <= 1.0); }
}  // namespace detail
inline Dist Normal(Expr mu, Expr sigma) { detail::need(detail::cbad_nonfinite(mu), "Normal: invalid mean", ErrorCode::InvalidMean);
    detail::need(detail::cbad_nonpos(sigma), "Normal: invalid sigma", ErrorCode::InvalidVariance); return {FG_NORMAL, {mu, sigma}}; }
inline Dist Uniform(Expr lo, Expr hi) { detail::need(detail::cbad_nonfinite(lo) || detail::cbad_nonfinite(hi) || (lo.is_const() && hi.is_const() && lo.value() >= hi.value()),
    "Uniform: invalid range", ErrorCode::InvalidRange); return {FG_UNIFORM, {lo, hi}}; }
inline Dist LogNormal(Expr mu, Expr sigma) { detail::need(detail::cbad_nonfinite(mu), "LogNormal: invalid mean", ErrorCode::InvalidMean);
    detail::need(detail::cbad_nonpos(sigma), "LogNormal: invalid sigma", ErrorCode::InvalidVariance); return {FG_LOGNORMAL, {mu, sigma}}; }
inline Dist Exponential(Expr rate) { detail::need(detail::cbad_nonpos(rate), "Exponential: invalid rate", ErrorCode::InvalidRate); return {FG_EXPONENTIAL, {rate}}; }
inline Dist Bernoulli(Expr p) { detail::need(detail::cbad_prob(p), "Bernoulli: invalid probability", ErrorCode::InvalidProbability); return {FG_BERNOULLI, {p}}; }
inline Dist Categorical(std::vector<Expr> probs) {
    detail::need(probs.empty(), "Categorical: probability vector cannot be empty", ErrorCode::InvalidProbability);
    detail::need(probs.size() > 64, "Categorical: at most 64 categories", ErrorCode::InvalidCount);
    bool allc = true; double sum = 0.0;
    for (auto &p : probs) { allc = allc && p.is_const(); if (p.is_const()) sum += p.value(); }
    if (allc) { detail::need(std::fabs(sum - 1.0) > 1e-6, "Categorical: probabilities must sum to 1.0", ErrorCode::InvalidProbability);
        for (auto &p : probs) detail::need(!std::isfinite(p.value()) || p.value() < 0.0, "Categorical: invalid probability", ErrorCode::InvalidProbability); }
    return {FG_CATEGORICAL, std::move(probs)}; }
inline Dist Beta(Expr a, Expr b) { detail::need(detail::cbad_nonpos(a) || detail::cbad_nonpos(b), "Beta: invalid shape", ErrorCode::InvalidShape); return {FG_BETA, {a, b}}; }
inline Dist Gamma(Expr shape, Expr rate) { detail::need(detail::cbad_nonpos(shape), "Gamma: invalid shape", ErrorCode::InvalidShape);
    detail::need(detail::cbad_nonpos(rate), "Gamma: invalid rate", ErrorCode::InvalidRate); return {FG_GAMMA, {shape, rate}}; }
inline Dist Binomial(Expr n, Expr p) { detail::need(detail::cbad_prob(p), "Binomial: invalid probability", ErrorCode::InvalidProbability); return {FG_BINOMIAL, {n, p}}; }
inline Dist Poisson(Expr lambda) { detail::need(detail::cbad_nonpos(lambda), "Poisson: invalid rate", ErrorCode::InvalidRate); return {FG_POISSON, {lambda}}; }
inline Dist StudentT(Expr df, Expr loc, Expr scale) { detail::need(detail::cbad_nonpos(df), "StudentT: invalid df", ErrorCode::InvalidShape);
    detail::need(detail::cbad_nonfinite(loc), "StudentT: invalid loc", ErrorCode::InvalidMean);
    detail::need(detail::cbad_nonpos(scale), "StudentT: invalid scale", ErrorCode::InvalidVariance); return {FG_STUDENTT, {df, loc, scale}}; }
inline Dist Cauchy(Expr loc, Expr scale) { detail::need(detail::cbad_nonfinite(loc), "Cauchy: invalid loc", ErrorCode::InvalidMean);
    detail::need(detail::cbad_nonpos(scale), "Cauchy: invalid scale", ErrorCode::InvalidVariance); return {FG_CAUCHY, {loc, scale}}; }
inline Dist Laplace(Expr loc, Expr scale) { detail::need(detail::cbad_nonfinite(loc), "Laplace: invalid loc", ErrorCode::InvalidMean);
    detail::need(detail::cbad_nonpos(scale), "Laplace: invalid scale", ErrorCode::InvalidVariance); return {FG_LAPLACE, {loc, scale}}; }
inline Dist Weibull(Expr shape, Expr scale) { detail::need(detail::cbad_nonpos(shape), "Weibull: invalid shape", ErrorCode::InvalidShape);
    detail::need(detail::cbad_nonpos(scale), "Weibull: invalid scale", ErrorCode::InvalidVariance); return {FG_WEIBULL, {shape, scale}}; }
inline Dist ChiSquared(Expr k) { detail::need(detail::cbad_nonpos(k), "ChiSquared: invalid k", ErrorCode::InvalidShape); return {FG_CHISQUARED, {k}}; }
inline Dist InverseGamma(Expr shape, Expr rate) { detail::need(detail::cbad_nonpos(shape), "InverseGamma: invalid shape", ErrorCode::InvalidShape);
    detail::need(detail::cbad_nonpos(rate), "InverseGamma: invalid rate", ErrorCode::InvalidRate); return {FG_INVERSEGAMMA, {shape, rate}}; }
inline Dist DiscreteUniform(Expr lo, Expr hi) { detail::need(lo.is_const() && hi.is_const() && hi.value() < lo.value(), "DiscreteUniform: high < low", ErrorCode::InvalidRange);
    return {FG_DISCRETEUNIFORM, {lo, hi}}; }

// ---- the recorded site program -----------------------------------------------------------------------
class Program {
  public:
    Program() : p_(fg_program_new()) {}
    // CompiledModel::compile (crates/fugue-wasm/src/dsl.rs:1062-1120): a finalized program from `prob!`-subset source
    static std::unique_ptr<Program> from_dsl(const std::string &source, const std::string &data_json = "") {
        fg_program *h = fg_dsl_compile(source.c_str(), data_json.c_str());
        if (!h) throw FugueError(fg_last_error(), ErrorCode::Engine);
        return std::unique_ptr<Program>(new Program(h));
    }
    std::vector<std::string> warnings() const { std::vector<std::string> w; for (int i = 0; i < fg_dsl_warning_count(p_); ++i) w.push_back(fg_dsl_warning(p_, i)); return w; }
    ~Program() { if (p_) fg_program_free(p_); }
    Program(const Program &) = delete;
    Program &operator=(const Program &) = delete;
    Expr sample(const Address &a, const Dist &d) {
        std::vector<fg_tok> toks; std::vector<int32_t> lens;
        pack(d, toks, lens);
        int h = fg_program_sample(p_, a.c_str(), d.kind, toks.data(), lens.data(), (int)lens.size());
        if (h < 0) throw FugueError(std::string("sample(") + a + "): " + fg_last_error(), h);
        return Expr::site(h);
    }
    void observe(const Address &a, const Dist &d, const Expr &value) {
        std::vector<fg_tok> toks, vt; std::vector<int32_t> lens;
        pack(d, toks, lens);
        value.postfix(vt);
        check(fg_program_observe(p_, a.c_str(), d.kind, toks.data(), lens.data(), (int)lens.size(), vt.data(), (int)vt.size()), "observe");
    }
    void factor(const Expr &logw) { std::vector<fg_tok> t; logw.postfix(t); check(fg_program_factor(p_, t.data(), (int)t.size()), "factor"); }
    void finalize() { int rc = fg_program_finalize(p_); if (rc) throw FugueError(fg_last_error(), rc); }
    fg_program *raw() const { return p_; }
    int n_sites() const { return fg_program_n_sites(p_); }
    int n_f64() const { return fg_program_n_f64(p_); }
    std::string site_name(int j) const { char buf[1024]; fg_program_site_name(p_, j, buf, sizeof buf); return buf; }
    int f64_site(int k) const { return fg_program_f64_site(p_, k); }
    int site_index(const Address &a) const { for (int j = 0; j < n_sites(); ++j) if (site_name(j) == a) return j;
        throw FugueError("address not found: " + a, ErrorCode::TraceAddressNotFound); }
  private:
    explicit Program(fg_program *h) : p_(h) {}
    static void pack(const Dist &d, std::vector<fg_tok> &toks, std::vector<int32_t> &lens) {
        for (auto &e : d.params) { size_t n0 = toks.size(); e.postfix(toks); lens.push_back((int32_t)(toks.size() - n0)); }
    }
    fg_program *p_;
};

// ---- Model<A>: the monadic surface, interpreted once into a Program -----------------------------------
template <class A> class Model {
  public:
    using Fn = std::function<A(Program &)>;
    explicit Model(Fn f) : run_(std::move(f)) {}
    A run(Program &p) const { return run_(p); }
    template <class F> auto bind(F f) const -> decltype(f(std::declval<A>())) {          // ModelExt::bind, model.rs:491-527
        using MB = decltype(f(std::declval<A>()));
        Fn r = run_;
        return MB([r, f](Program &p) { A a = r(p); return f(a).run(p); });
    }
    template <class F> auto and_then(F f) const { return bind(f); }
    template <class F> auto map(F f) const -> Model<decltype(f(std::declval<A>()))> {    // ModelExt::map, model.rs:529-561
        using B = decltype(f(std::declval<A>()));
        Fn r = run_;
        return Model<B>([r, f](Program &p) { return f(r(p)); });
    }
  private:
    Fn run_;
};
template <class A> Model<A> pure(A a) { return Model<A>([a](Program &) { return a; }); }
inline Model<Expr> sample(Address a, Dist d) { return Model<Expr>([a, d](Program &p) { return p.sample(a, d); }); }
inline Model<Unit> observe(Address a, Dist d, Expr v) { return Model<Unit>([a, d, v](Program &p) { p.observe(a, d, v); return Unit{}; }); }
inline Model<Unit> factor(Expr logw) { return Model<Unit>([logw](Program &p) { p.factor(logw); return Unit{}; }); }
inline Model<Unit> guard(bool pred) { return pred ? pure(Unit{}) : factor(Expr(-INFINITY)); }    // model.rs:710-716
template <class A, class B> Model<std::pair<A, B>> zip(Model<A> ma, Model<B> mb) {
    return ma.bind([mb](A a) { return mb.map([a](B b) { return std::make_pair(a, b); }); });
}
template <class A> Model<std::vector<A>> sequence_vec(std::vector<Model<A>> ms) {                 // model.rs:623-658
    return Model<std::vector<A>>([ms](Program &p) { std::vector<A> out; for (auto &m : ms) out.push_back(m.run(p)); return out; });
}
template <class T, class F> auto traverse_vec(const std::vector<T> &items, F f) {
    using MA = decltype(f(items[0]));
    std::vector<MA> ms; for (auto &x : items) ms.push_back(f(x));
    return sequence_vec(ms);
}
/// plate!(i in 0..n => body)
template <class F> auto plate(int n, F f) { std::vector<int> idx(n); for (int i = 0; i < n; ++i) idx[i] = i; return traverse_vec(idx, f); }

// ---- engine handle + many-chain drivers ---------------------------------------------------------------
enum class GradMode { FdDense = FG_GRAD_FD_DENSE, FdSparse = FG_GRAD_FD_SPARSE, Analytic = FG_GRAD_ANALYTIC };
struct HMCConfig {                       // hmc.rs:106-135, same defaults
    size_t n_leapfrog = 16; double target_accept = 0.8; std::optional<double> init_step_size; double finite_diff_eps = 1e-5;
    bool adapt_mass = false; GradMode grad_mode = GradMode::FdSparse;
};
enum class ResamplingMethod { Multinomial = 0, Systematic = 1, Stratified = 2 };
struct SMCConfig { ResamplingMethod resampling_method = ResamplingMethod::Systematic; double ess_threshold = 0.5; size_t rejuvenation_steps = 0; };
struct SiteProposal { int kind = FG_PROP_AUTO; double lower = 0, upper = 0;
    static SiteProposal Gaussian() { return {FG_PROP_GAUSSIAN, 0, 0}; }
    static SiteProposal LogSpace() { return {FG_PROP_LOGSPACE, 0, 0}; }
    static SiteProposal Reflect(double lo, double hi) { return {FG_PROP_REFLECT, lo, hi}; }
    static SiteProposal PriorResample() { return {FG_PROP_PRIOR_RESAMPLE, 0, 0}; } };

/// Draws of many chains: the many-chain analogue of Vec<(A, Trace)>.  value(t, site, chain) is the
/// choice at address `sites[site]` of draw t of chain `chain`.
struct ChainBatch {
    std::vector<Address> sites; size_t n_samples = 0, n_sites = 0, n_chains = 0;
    std::vector<double> draws;            // [n_samples][n_sites][n_chains]; integer sites are converted to f64
    double accept_rate = 0.0, mean_step_size = 0.0; long long n_divergent = 0;
    double value(size_t t, size_t site, size_t chain) const { return draws[(t * n_sites + site) * n_chains + chain]; }
    double mean(size_t site) const { double s = 0; for (size_t t = 0; t < n_samples; ++t) for (size_t c = 0; c < n_chains; ++c) s += value(t, site, c);
        return s / (double)(n_samples * n_chains); }
};
struct SMCResult { std::vector<Address> sites; size_t n_particles = 0; std::vector<double> values /*[n_sites][N]*/, weights, log_weights;
    double log_evidence = 0.0; std::vector<double> betas; };

class Engine {
  public:
    Engine(const Program &p, int64_t n_chains, uint64_t seed, int device = 0, uint32_t chain_offset = 0)
        : e_(fg_engine_new(p.raw(), n_chains, seed, chain_offset, device)), n_(n_chains) {
        if (!e_) throw FugueError(std::string("engine: ") + fg_last_error(), ErrorCode::Engine);
    }
    ~Engine() { if (e_) fg_engine_free(e_); }
    Engine(const Engine &) = delete;
    fg_engine *raw() const { return e_; }
    int64_t n_chains() const { return n_; }
  private:
    fg_engine *e_; int64_t n_;
};

template <class A, class F> std::unique_ptr<Program> flatten(F model_fn) {
    auto p = std::make_unique<Program>();
    Model<A> m = model_fn();
    (void)m.run(*p);
    p->finalize();
    return p;
}

/// hmc_chain (hmc.rs:566-583) for `n_chains` chains at once.
template <class A, class F>
ChainBatch hmc_chain(uint64_t seed, F model_fn, size_t n_samples, size_t n_warmup, HMCConfig config, int64_t n_chains, int device = 0) {
    auto prog = flatten<A>(model_fn);
    Engine eng(*prog, n_chains, seed, device);
    const size_t d = (size_t)prog->n_f64();
    fg_hmc_config c{(int32_t)config.n_leapfrog, config.target_accept, config.init_step_size ? *config.init_step_size : NAN,
                    config.finite_diff_eps, config.adapt_mass ? 1 : 0, (int32_t)config.grad_mode};
    ChainBatch out; out.n_samples = n_samples; out.n_sites = d; out.n_chains = (size_t)n_chains;
    for (size_t k = 0; k < d; ++k) out.sites.push_back(prog->site_name(prog->f64_site((int)k)));
    const size_t bytes = std::max<size_t>(1, n_samples * d * (size_t)n_chains) * sizeof(double);
    double *d_draws = (double *)fg_device_alloc(eng.raw(), bytes);
    if (!d_draws) throw FugueError(fg_last_error(), ErrorCode::Engine);
    fg_hmc_stats st{};
    int rc = fg_hmc_run(eng.raw(), &c, (int)n_samples, (int)n_warmup, d_draws, &st);
    out.draws.resize(n_samples * d * (size_t)n_chains);
    if (!rc && !out.draws.empty()) rc = fg_device_download(eng.raw(), out.draws.data(), d_draws, out.draws.size() * sizeof(double));
    fg_device_free(eng.raw(), d_draws);
    check(rc, "hmc_chain");
    out.accept_rate = st.accept_rate; out.mean_step_size = st.mean_step_size; out.n_divergent = st.n_divergent;
    return out;
}

/// adaptive_mcmc_chain[_with_overrides] (mh.rs:921-1014) for `n_chains` chains; records every site.
template <class A, class F>
ChainBatch adaptive_mcmc_chain(uint64_t seed, F model_fn, size_t n_samples, size_t n_warmup, int64_t n_chains,
                               const std::vector<std::pair<Address, SiteProposal>> &overrides = {}, int device = 0) {
    auto prog = flatten<A>(model_fn);
    Engine eng(*prog, n_chains, seed, device);
    const size_t S = (size_t)prog->n_sites();
    std::vector<fg_site_proposal> ov(S, fg_site_proposal{FG_PROP_AUTO, 0, 0});
    for (auto &o : overrides) { int j = prog->site_index(o.first); ov[j] = fg_site_proposal{o.second.kind, o.second.lower, o.second.upper}; }
    std::vector<int32_t> rec(S); for (size_t j = 0; j < S; ++j) rec[j] = (int32_t)j;
    ChainBatch out; out.n_samples = n_samples; out.n_sites = S; out.n_chains = (size_t)n_chains;
    for (size_t j = 0; j < S; ++j) out.sites.push_back(prog->site_name((int)j));
    const size_t cells = std::max<size_t>(1, n_samples * S * (size_t)n_chains);
    void *d_draws = fg_device_alloc(eng.raw(), cells * 8);
    if (!d_draws) throw FugueError(fg_last_error(), ErrorCode::Engine);
    fg_mh_stats st{};
    int rc = fg_mh_run(eng.raw(), (int)n_samples, (int)n_warmup, overrides.empty() ? nullptr : ov.data(), rec.data(), (int)S, d_draws, &st);
    std::vector<int64_t> raw(n_samples * S * (size_t)n_chains);
    if (!rc && !raw.empty()) rc = fg_device_download(eng.raw(), raw.data(), d_draws, raw.size() * 8);
    fg_device_free(eng.raw(), d_draws);
    check(rc, "adaptive_mcmc_chain");
    out.draws.resize(raw.size());
    for (size_t t = 0; t < n_samples; ++t) for (size_t j = 0; j < S; ++j) {
        const bool is_f64 = fg_program_site_vtype(prog->raw(), (int)j) == FG_F64;
        for (size_t c = 0; c < (size_t)n_chains; ++c) { const size_t k = (t * S + j) * (size_t)n_chains + c; double v;
            if (is_f64) std::memcpy(&v, &raw[k], 8); else v = (double)raw[k];
            out.draws[k] = v; } }
    out.accept_rate = st.accept_rate;
    return out;
}

/// adaptive_smc (smc.rs:455-581) with `num_particles` particles.
template <class A, class F>
SMCResult adaptive_smc(uint64_t seed, size_t num_particles, F model_fn, SMCConfig config, int device = 0) {
    SMCResult out;
    if (num_particles == 0) return out;                          // smc.rs:462-467
    auto prog = flatten<A>(model_fn);
    Engine eng(*prog, (int64_t)num_particles, seed, device);
    const size_t S = (size_t)prog->n_sites();
    fg_smc_config c{(int32_t)config.resampling_method, config.ess_threshold, (int32_t)config.rejuvenation_steps, 0};
    fg_smc_result r{};
    out.n_particles = num_particles; out.weights.resize(num_particles); out.log_weights.resize(num_particles); out.betas.resize(10000);
    check(fg_smc_run(eng.raw(), &c, out.log_weights.data(), out.weights.data(), &r, out.betas.data(), (int)out.betas.size()), "adaptive_smc");
    out.betas.resize((size_t)r.n_steps); out.log_evidence = r.log_evidence;
    std::vector<int64_t> raw(std::max<size_t>(1, S * num_particles));
    check(fg_engine_get_values(eng.raw(), raw.data()), "get_values");
    out.values.resize(S * num_particles);
    for (size_t j = 0; j < S; ++j) { out.sites.push_back(prog->site_name((int)j));
        const bool is_f64 = fg_program_site_vtype(prog->raw(), (int)j) == FG_F64;
        for (size_t i = 0; i < num_particles; ++i) { double v; if (is_f64) std::memcpy(&v, &raw[j * num_particles + i], 8); else v = (double)raw[j * num_particles + i];
            out.values[j * num_particles + i] = v; } }
    return out;
}

}  // namespace fugue
