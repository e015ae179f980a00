// Exercises the C++ mirror (include/fugue_amd.hpp) the way the reference's own tests exercise Fugue:
//   CPU part  : model building, address encoding/order, constructor validation, duplicate addresses.
//   --gpu part: hmc_chain on the conjugate Normal of tests/f_hmc_posterior.rs / README.md:74-80, adaptive_mcmc_chain on the
//               same model, adaptive_smc on examples/smc_inference.rs:36-39 -- closed-form targets.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "fugue_amd.hpp"
using namespace fugue;

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static Model<Expr> readme_model() {      // README.md:74-80
    return sample(addr("mu"), Normal(0.0, 1.0)).bind([](Expr mu) {
        return observe(addr("y"), Normal(mu, 0.5), 1.2).map([mu](Unit) { return mu; });
    });
}
static Model<Expr> smc_model() {         // examples/smc_inference.rs:36-39
    return sample(addr("mu"), Normal(0.0, 1.0)).bind([](Expr mu) {
        return observe(addr("y"), Normal(mu, 0.5), 1.5).map([mu](Unit) { return mu; });
    });
}
static Model<std::vector<Unit>> plate_model() {
    return sample(addr("p"), Beta(2.0, 2.0)).bind([](Expr p) {
        return plate(12, [p](int i) { return observe(addr("flip", i), Bernoulli(p.clamp(1e-10, 1.0 - 1e-10)), i % 3 != 0); });
    });
}

int main(int argc, char **argv) {
    // address encoding (address.rs:189-223)
    REQUIRE(addr("x", 10) == "x#10");
    REQUIRE(addr("a#b", "c\\d") == "a\\#b#c\\\\d");
    // constructor validation throws with the reference ErrorCode
    try { Normal(0.0, -1.0); REQUIRE(false); } catch (const FugueError &e) { REQUIRE(e.code == 101); }
    try { Bernoulli(1.5); REQUIRE(false); } catch (const FugueError &e) { REQUIRE(e.code == 102); }
    try { Categorical({0.5, 0.6}); REQUIRE(false); } catch (const FugueError &e) { REQUIRE(e.code == 102); }
    // site order is the BTreeMap (lexicographic) order
    {
        Program p;
        for (int i = 0; i < 12; ++i) { Expr x = p.sample(addr("x", i), Normal(0.0, 1.0)); p.observe(addr("y", i), Normal(x, 0.5), 0.2 * i - 1.0); }
        p.finalize();
        REQUIRE(p.n_sites() == 12 && p.n_f64() == 12);
        REQUIRE(p.site_name(0) == "x#0" && p.site_name(1) == "x#1" && p.site_name(2) == "x#10" && p.site_name(3) == "x#11" && p.site_name(4) == "x#2");
    }
    {   // monadic surface
        auto prog = flatten<Expr>(readme_model);
        REQUIRE(prog->n_sites() == 1 && prog->site_name(0) == "mu");
        auto prog2 = flatten<std::vector<Unit>>(plate_model);
        REQUIRE(prog2->n_sites() == 1 && fg_program_n_observe(prog2->raw()) == 12);
    }
    {   // duplicate address = AddressConflict (interpreters.rs:23-33)
        Program p; p.sample(addr("x"), Normal(0.0, 1.0)); p.sample(addr("x"), Normal(0.0, 1.0));
        try { p.finalize(); REQUIRE(false); } catch (const FugueError &e) { REQUIRE(e.code == 301); }
    }
    {   // model-language front-end (crates/fugue-wasm/src/dsl.rs:1149-1187 coin model; :1233-1256 static errors)
        auto coin = Program::from_dsl("let p <- sample(addr!(\"p\"), Beta(2.0, 2.0));\n"
                                      "for i in 0..data.len() { observe(addr!(\"flip\", i), Bernoulli(p), data[i]); }\npure(p)",
                                      "[1,0,1,1,0,1,1,0,1,1]");
        REQUIRE(coin->n_sites() == 1 && coin->site_name(0) == addr("p") && fg_program_n_observe(coin->raw()) == 10 && coin->warnings().empty());
        try { Program::from_dsl("pure(nope)"); REQUIRE(false); } catch (const FugueError &e) { REQUIRE(std::string(e.what()).find("unknown variable") != std::string::npos); }
        try { Program::from_dsl("let x <- sample(addr!(\"x\") Normal(0,1)); pure(x)"); REQUIRE(false); }
        catch (const FugueError &e) { REQUIRE(std::string(e.what()).find("line 1") != std::string::npos); }
    }
    if (argc > 1 && std::strcmp(argv[1], "--gpu") == 0) {
        HMCConfig cfg;                                            // HMCConfig::default()
        ChainBatch h = hmc_chain<Expr>(42, readme_model, 200, 200, cfg, 4096);
        std::printf("hmc  mean(mu) = %.5f  accept = %.3f  eps = %.3f\n", h.mean(0), h.accept_rate, h.mean_step_size);
        REQUIRE(h.sites.size() == 1 && h.sites[0] == "mu");
        REQUIRE(std::fabs(h.mean(0) - 0.96) < 5e-3 && h.n_divergent == 0);
        ChainBatch m = adaptive_mcmc_chain<Expr>(42, readme_model, 400, 500, 4096);
        std::printf("mh   mean(mu) = %.5f  accept = %.3f\n", m.mean(0), m.accept_rate);
        REQUIRE(std::fabs(m.mean(0) - 0.96) < 5e-3);
        SMCConfig sc; sc.rejuvenation_steps = 3;
        SMCResult s = adaptive_smc<Expr>(42, 1 << 18, smc_model, sc);
        double mean = 0; for (size_t i = 0; i < s.n_particles; ++i) mean += s.weights[i] * s.values[i];
        std::printf("smc  logZ = %.5f  mean(mu) = %.5f  steps = %zu\n", s.log_evidence, mean, s.betas.size());
        REQUIRE(std::fabs(s.log_evidence - (-1.9305103088617774)) < 1e-2 && std::fabs(mean - 1.2) < 1e-2);
    } else {
        // without a device the engine refuses to run: no CPU fallback
        bool refused = false;
        try { auto prog = flatten<Expr>(readme_model); Engine e(*prog, 64, 1); } catch (const FugueError &) { refused = true; }
        if (argc > 1 && std::strcmp(argv[1], "--expect-no-device") == 0) REQUIRE(refused);
    }
    std::printf("C++ mirror OK\n");
    return 0;
}
