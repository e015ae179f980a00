// Host-side sanitizer driver (AddressSanitizer + UndefinedBehaviorSanitizer, CPU only): the parser over caller-supplied text
// (fg_dsl.cpp), the program builder / compiler (fg_program.cpp) and the diagnostics combination (fg_diag_host.cpp) built with
// -fsanitize=address,undefined and driven through their C ABI.  tests/test_sanitizers_cpu.py feeds it the model-language sources
// of tests/dsl_models.py, malformed and truncated variants of them, and random diagnostics inputs.
//   san_driver dsl <file>      file = records "SRC\n<source>\nDATA\n<json>\nEND\n"...: compiles each, prints ok / error per record
//   san_driver diag <seed>     random moments through fg_diag_combine / fg_diag_combine_reduced
//   san_driver program <seed>  random token streams through the builder (malformed expressions included)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/fugue_amd.h"
#include "../../fugue_amd/csrc/fg_jit.h"
#include "../../fugue_amd/csrc/fg_program.h"

// the run-time code generator (fg_jit.cpp, host-only build: FG_JIT_NO_HIP) over a finalized program: both translation units
static size_t jit_sources(const fg_program *p) {
    std::vector<double> ta, tb;
    const std::string a = fg_jit_hmc_source(p, &ta);
    const std::string b = fg_jit_mh_source(p, std::vector<long long>((size_t)p->n_ins, 1), 4, &tb);
    return a.size() + b.size() + ta.size() + tb.size();
}

static int run_dsl(const char *path) {
    std::ifstream f(path);
    std::string line, src, data;
    int state = 0, n = 0, ok = 0;
    while (std::getline(f, line)) {
        if (line == "SRC") { state = 1; src.clear(); data.clear(); continue; }
        if (line == "DATA") { state = 2; continue; }
        if (line == "END") {
            fg_program *p = fg_dsl_compile(src.c_str(), data == "<null>" ? nullptr : data.c_str());
            ++n;
            if (p) {
                ++ok;
                char buf[64];
                for (int j = 0; j < fg_program_n_sites(p); ++j) fg_program_site_name(p, j, buf, sizeof buf);
                for (int w = 0; w < fg_dsl_warning_count(p); ++w) (void)std::strlen(fg_dsl_warning(p, w));
                for (int k = 0; k < 5; ++k) (void)fg_program_stream_records(p, k);
                (void)jit_sources(p);
                std::printf("ok %d sites %d observes %d instructions\n", fg_program_n_sites(p), fg_program_n_observe(p), fg_program_n_instructions(p));
                fg_program_free(p);
            } else std::printf("error %s\n", fg_last_error());
            state = 0;
            continue;
        }
        if (state == 1) { src += line; src += '\n'; }
        else if (state == 2) { data += line; }
    }
    std::printf("compiled %d of %d\n", ok, n);
    return 0;
}

struct Mom { std::vector<double> mom, acov; int d, n; long long m; };
static int acov_cb(void *user, int lag0, int n_lags, double *out) {
    Mom *M = (Mom *)user;
    for (int i = 0; i < M->d; ++i) for (int k = 0; k < n_lags; ++k) out[(size_t)i * n_lags + k] = M->acov[(size_t)i * M->n + std::min(M->n - 1, lag0 + k)] * (double)M->m;
    return 0;
}
static int reduce_cb(void *user, int stage, const double *in, double *out) {
    Mom *M = (Mom *)user;
    for (int i = 0; i < M->d; ++i) {
        const double *mo = &M->mom[(size_t)i * 6 * M->m];
        if (stage == 1) for (int k = 0; k < 6; ++k) { double s = 0; for (long long j = 0; j < M->m; ++j) s += mo[k * M->m + j]; out[6 * i + k] = s; }
        else {
            double a = 0, b = 0;
            for (long long j = 0; j < M->m; ++j) { const double x = mo[j] - in[2 * i], y = mo[2 * M->m + j] - in[2 * i + 1], z = mo[4 * M->m + j] - in[2 * i + 1]; a += x * x; b += y * y + z * z; }
            out[2 * i] = a; out[2 * i + 1] = b;
        }
    }
    return 0;
}
static int run_diag(unsigned seed) {
    std::mt19937_64 g(seed);
    std::normal_distribution<double> N(0.0, 1.0);
    for (int rep = 0; rep < 40; ++rep) {
        Mom M; M.d = 1 + (int)(g() % 4); M.n = (int)(g() % 70); M.m = (long long)(g() % 9);
        M.mom.resize((size_t)M.d * 6 * std::max<long long>(1, M.m));
        for (double &v : M.mom) v = std::fabs(N(g));
        if (rep % 7 == 3) for (double &v : M.mom) v = 0.0;                                  // constant chains
        M.acov.resize((size_t)M.d * std::max(1, M.n));
        for (size_t k = 0; k < M.acov.size(); ++k) M.acov[k] = std::pow(0.8, (double)(k % std::max(1, M.n))) * (rep % 5 == 1 ? -1.0 : 1.0);
        std::vector<double> r(M.d), e(M.d), mu(M.d), sd(M.d);
        int rc = fg_diag_combine(M.mom.data(), M.m, M.n, M.d, acov_cb, &M, r.data(), e.data(), mu.data(), sd.data());
        int rc2 = fg_diag_combine_reduced(M.m, M.n, M.d, reduce_cb, acov_cb, &M, r.data(), e.data(), mu.data(), sd.data());
        std::printf("diag d=%d n=%d m=%lld rc=%d rc2=%d ess0=%g\n", M.d, M.n, M.m, rc, rc2, e[0]);
    }
    return 0;
}
static int run_program(unsigned seed) {
    std::mt19937_64 g(seed);
    int built = 0, refused = 0;
    for (int rep = 0; rep < 300; ++rep) {
        fg_program *p = fg_program_new();
        const double dat[4] = {0.5, -1.0, 2.0, 1e300};
        fg_program_data(p, "y", dat, 4);
        int n_sites = 0;
        bool ok = true;
        const int n_stmt = 1 + (int)(g() % 6);
        for (int sidx = 0; sidx < n_stmt && ok; ++sidx) {
            // a well-formed postfix expression over constants, data and the sites sampled so far -- with a chance of corruption per
            // token (an invalid opcode, an out-of-range handle or data index, a NaN constant, a dropped operand)
            std::function<void(std::vector<fg_tok> &, int)> rnd_expr = [&](std::vector<fg_tok> &t, int depth) {
                fg_tok q; std::memset(&q, 0, sizeof q);
                const bool corrupt = g() % 12 == 0;
                const int kind = (depth <= 0) ? (int)(g() % 3) : (int)(g() % 6);
                if (kind == 0) { q.op = 0; q.imm = corrupt ? NAN : (double)((long long)(g() % 400) + 1) / 100.0; t.push_back(q); }
                else if (kind == 1 && n_sites > 0) { q.op = 1; q.a = corrupt ? n_sites + 3 : (int)(g() % n_sites); t.push_back(q); }
                else if (kind == 1 || kind == 2) { q.op = 2; q.a = corrupt ? 7 : 0; q.b = corrupt ? -1 : (int)(g() % 4); t.push_back(q); }
                else if (kind == 3) { rnd_expr(t, depth - 1); q.op = corrupt ? 99 : 3 + (int)(g() % 9); t.push_back(q); }                 // unary
                else if (kind == 4) { rnd_expr(t, depth - 1); if (!corrupt) rnd_expr(t, depth - 1); q.op = 12 + (int)(g() % 7); t.push_back(q); }   // binary
                else { rnd_expr(t, depth - 1); rnd_expr(t, depth - 1); rnd_expr(t, depth - 1); q.op = 19; t.push_back(q); }                // clamp
            };
            std::vector<fg_tok> toks; std::vector<int32_t> plen;
            const int dist = (g() % 15 == 0) ? (int)(g() % 21) - 2 : (int)(g() % 17);
            static const int want[17] = {1, 2, 2, -1, 2, 1, -1, 1, 2, 2, 2, 2, 2, 1, 3, 2, 2};   // parameters of the 17 families (include/fugue_amd.h order), -1: special
            const int np = (dist >= 0 && dist < 17 && want[dist] > 0 && g() % 10) ? want[dist] : (int)(g() % 4);
            for (int k = 0; k < np; ++k) { const size_t b = toks.size(); rnd_expr(toks, 2); plen.push_back((int32_t)(toks.size() - b)); }
            char addr[32]; std::snprintf(addr, sizeof addr, "s#%d", (int)(g() % 5));       // collisions happen
            int rc;
            if (g() % 3) { rc = fg_program_sample(p, addr, dist, toks.data(), plen.data(), np); if (rc >= 0) ++n_sites; }
            else { std::vector<fg_tok> v; rnd_expr(v, 1); rc = fg_program_observe(p, addr, dist, toks.data(), plen.data(), np, v.data(), (int)v.size()); }
            if (rc < 0) ok = false;
        }
        if (ok && fg_program_finalize(p) == 0) { ++built; for (int k = 0; k < 5; ++k) (void)fg_program_stream_records(p, k); (void)fg_program_n_slots(p); (void)jit_sources(p); }
        else ++refused;
        fg_program_free(p);
    }
    std::printf("programs built %d refused %d\n", built, refused);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 3 && !std::strcmp(argv[1], "dsl")) return run_dsl(argv[2]);
    if (argc >= 3 && !std::strcmp(argv[1], "diag")) return run_diag((unsigned)std::atoi(argv[2]));
    if (argc >= 3 && !std::strcmp(argv[1], "program")) return run_program((unsigned)std::atoi(argv[2]));
    std::fprintf(stderr, "usage: san_driver dsl <file> | diag <seed> | program <seed>\n");
    return 2;
}
