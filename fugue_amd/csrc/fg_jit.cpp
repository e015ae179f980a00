// fg_jit.cpp -- models compiled at run time: straight-line HIP for the programs the record streams do not cover.
//
// A program with an expression parameter (a GLM's link function, exp(log_sigma), a select ...) has no gradient / score stream and runs
// on the interpreter (fg_interp.h), which spends ~100 issue slots decoding each FgIns -- most of a step for models whose arithmetic
// is light (logistic regression: a leapfrog step costs 7x its f64 instructions; tools/mb_interp_costs.py).  The interpreter's input
// is already a flat list of FgIns per sub-program, so each becomes ONE C++ statement here -- the same operation on the same
// operands in the same order, fields as literals, expression temporaries as locals instead of LDS rows -- and hiprtc compiles the
// result for gfx950 behind fg_hmc_jit_body.h (the multi-wave HMC kernel of fg_hmc_interp.hip around two generated functions).
// Results are bit-identical to the interpreter kernels (tests/test_gpu_jit.py); when hiprtc is missing, the program is too long or
// the compilation fails the engine stays on them (FG_JIT=0 forces that).  Nothing here runs on the CPU at sampling time.
#include <dlfcn.h>
#include <fcntl.h>
#include <cerrno>
#include <functional>

#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#ifndef FG_JIT_NO_HIP             /* host-only builds (sanitizer tests of the generator): no hiprtc binding */
#include <hip/hip_runtime.h>
#endif

#include "fg_program.h"
#include "fg_jit.h"
#include "_fg_jit_embed.inc"      // FG_JIT_EMBED: fg_ir.h, fg_math.h, fg_cold.h, fg_dev_types.h and fg_hmc_jit_body.h as text (fugue_amd/build.py)

namespace {

std::string lit(double v) {                      // a double literal with exactly these bits
    char b[64];
    if (std::isfinite(v)) { std::snprintf(b, sizeof b, "%a", v); return std::string("(") + b + ")"; }
    long long bits; std::memcpy(&bits, &v, 8);
    std::snprintf(b, sizeof b, "fg_as_double((long long)0x%016llxULL)", (unsigned long long)bits);
    return b;
}

// The constants of the rolled runs of one module.  They live in ONE device buffer whose address the engine writes into the module's
// global `fg_jit_ctab_ptr` after loading it -- not in the source text: the generated unit then depends on the STRUCTURE of a program,
// not on its data (a regression on another data set of the same shape finds its code object in the cache), and its size does not
// grow with the number of observations.  The code reads them through the constant address space (uniform addresses: scalar loads).
struct FgJitTabs {
    std::vector<double> data;
    std::map<std::string, size_t> index;         // table bytes -> offset (in doubles)
    size_t add(const std::vector<double> &t) {
        const std::string key((const char *)t.data(), t.size() * sizeof(double));
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        const size_t off = data.size();
        data.insert(data.end(), t.begin(), t.end());
        index[key] = off;
        return off;
    }
};

struct Gen {
    const fg_program &p;
    int pert_slot = -1;                          // slot whose reads become `pert` (a finite-difference task) or -1
    std::string body;
    std::set<int> temps;
    std::map<std::string, std::string> *lp_fns;  // signature -> definition of the per-signature density wrappers
    std::vector<std::string> *tables;            // file-scope constant tables
    bool ok = true;
    int term = -1;                               // >= 0: a scoring run for MH -- statement k leaves its term in row k of `terms` (fg_exec's TM mode)
    // Rolled runs: >= 4 consecutive statements that differ only in their constants (a plate of observations) become ONE loop over a
    // constant table -- the same statements in the same order, the constants read from memory instead of the instruction stream.
    std::vector<double> *cvec = nullptr;         // non-null: constants go here and the code says c[j]
    bool rolled_term = false;                    // ... and a term row is (term + r)
    FgJitTabs *ctabs = nullptr;                  // the constants of rolled runs: ONE device buffer per module, identical tables (the d sub-programs of a
                                                 // regression hold the same observations) stored once
    // Cooperative scoring (the MH kernel's direct mode): the function is entered by ALL waves of the tile.  The statements of a plate are
    // shared by the waves -- statement r0 + rr of a chunk by wave rr mod W, its term into row rr of one half of a ring of LDS rows --
    // and wave 0 adds each chunk's rows to the accumulator IN ORDER behind a barrier (the other half of the ring is being filled
    // meanwhile); everything that is not a plate runs on wave 0 alone.  The accumulators are wave 0's.
    bool coop = false;
    bool ring_term = false;                      // the statement being emitted writes its term to ring row (buf * CH + rr)
    bool mh_terms = false;                       // the multi-wave MH kernel's phase B: a term row may hold NaN for -inf (see FG_OP_NORMAL_FAST)
    bool prior = false;                          // run(PriorHandler): every sample statement first DRAWS its value from `rng` (interpreters.rs:88-104), then scores it
    const std::vector<int> *rows = nullptr;      // TM mode: term row of statement k when it is not k itself (the multi-wave stream kernel's rows: log_prior
                                                 // terms first, then log_likelihood terms)

    std::string lit(double v) {
        if (!cvec) return ::lit(v);
        cvec->push_back(v);
        return "c[" + std::to_string(cvec->size() - 1) + "]";
    }

    std::string slot(uint32_t idx) {
        if ((int)idx == p.n_slots - 1) return "0.0";                        // the always-zero slot
        if ((int)idx < (int)p.site_vtype.size()) return (int)idx == pert_slot ? std::string("pert") : "slots[" + std::to_string(idx) + " * FG_WAVE]";
        temps.insert((int)idx);
        return "t" + std::to_string(idx);
    }
    std::string opnd(uint32_t w, double imm) {
        const uint32_t kind = FG_OPND_KIND(w), idx = FG_OPND_IDX(w);
        if (kind == FG_OPND_IMM) return lit(imm);
        if (kind == FG_OPND_SLOT_F) return slot(idx);
        if (kind == FG_OPND_SLOT_I) return "(double)fg_as_i64(" + slot(idx) + ")";
        return lit(p.pool[idx]);
    }
    // per-lane choice among consecutive temporaries t[base .. base + K): v = t[base + j]
    std::string pick(int base, int K, const std::string &j, const std::string &out) {
        std::string s = "double " + out + " = " + slot((uint32_t)base) + "; ";
        for (int q = 1; q < K; ++q) s += out + " = (" + j + " == " + std::to_string(q) + ") ? " + slot((uint32_t)(base + q)) + " : " + out + "; ";
        return s;
    }
    void add(const std::string &s) { body += "    " + s + "\n"; }
    std::string term_row() {                     // the LDS row of the statement being emitted (TM mode); a rolled run emits ONE statement for R rows
        if (ring_term) return "ring[(buf * FG_JIT_CH + rr) * FG_WAVE]";
        if (rolled_term) return "terms[(" + std::to_string(rows ? (*rows)[(size_t)term] : term) + " + r) * FG_WAVE]";
        const int k = term++;
        return "terms[" + std::to_string(rows ? (*rows)[(size_t)k] : k) + " * FG_WAVE]";
    }
    static bool ends_statement(const FgIns &I) {
        const uint32_t code = FG_INS_OPCODE(I.op);
        return code == FG_OP_NORMAL_FAST || code < 17u || code == FG_OP_FACTOR || code == FG_OP_CONSTLIK;
    }
    // everything of an instruction that shapes the generated code, i.e. all but the values of its constants
    std::string shape(const FgIns &I) const {
        const uint32_t code = FG_INS_OPCODE(I.op);
        if (code == 3u) return "";                                            // Categorical: a table of its own -- never rolled
        char b[128];
        std::snprintf(b, sizeof b, "%x:%x:%x:%x:%x:%x;", I.op, I.opnd[0], I.opnd[1], I.opnd[2], I.opnd[3], I.aux);
        std::string k = b;
        if (code == FG_OP_DOT) {                                              // its terms' slots shape the code, its coefficients are constants
            k = "dot" + std::to_string(I.opnd[1]) + ":";
            for (int t = 0; t < (int)I.opnd[1]; ++t) { long long sb; std::memcpy(&sb, &p.pool[(size_t)I.aux + 2 * t], 8); k += std::to_string(sb) + ","; }
            k += ";";
        }
        // an operand that reads the constant pool is a constant too
        for (int q = 0; q < 4; ++q) if (FG_OPND_KIND(I.opnd[q]) == FG_OPND_POOL && code != 3u) { k += "P"; }
        return k;
    }
    // instructions [b, e) of v: statement by statement, rolling runs of isomorphic statements
    void emit(const std::vector<FgIns> &v, size_t b, size_t e) {
        struct St { size_t b, e; std::string key; };
        std::vector<St> st;
        for (size_t q = b, s0 = b; q < e; ++q)
            if (ends_statement(v[q]) || q + 1 == e) {
                St x{s0, q + 1, ""};
                bool rollable = true;
                for (size_t i = s0; i <= q; ++i) { const std::string k = shape(v[i]); if (k.empty()) rollable = false; x.key += k; }
                if (!rollable) x.key = "#" + std::to_string(st.size());       // unique: never equal to a neighbour's
                st.push_back(x); s0 = q + 1;
            }
        for (size_t i = 0; i < st.size();) {
            size_t j = i + 1;
            while (j < st.size() && st[j].key == st[i].key && st[i].key[0] != '#') ++j;
            const size_t R = j - i;
            if (R < 4 || !ctabs) {
                if (coop) add("if (wv == 0) {");
                for (size_t q = st[i].b; q < st[i].e; ++q) ins(v[q]);
                if (coop) add("}");
                i += 1; continue;
            }
            if (term >= 0 && rows) {                                          // a rolled run writes rows ROW0 + r: they must be consecutive
                bool consecutive = true;
                for (size_t q = 1; q < R; ++q) consecutive = consecutive && (*rows)[(size_t)term + q] == (*rows)[(size_t)term] + (int)q;
                if (!consecutive) { for (size_t s2 = i; s2 < j; ++s2) for (size_t q = st[s2].b; q < st[s2].e; ++q) ins(v[q]); i = j; continue; }
            }
            const bool share = coop && R >= 16;                               // a plate long enough to share between the waves
            // one statement's code with its constants as c[0 .. K); every statement's constants into the table, in the same order
            std::vector<double> first; std::string code;
            {
                const std::string keep = body; body.clear();
                cvec = &first; rolled_term = term >= 0; ring_term = share;
                for (size_t q = st[i].b; q < st[i].e; ++q) ins(v[q]);
                code = body; body = keep;
            }
            const size_t K = first.size();
            std::vector<double> tab = first;
            bool same_k = true;
            for (size_t s = i + 1; s < j && same_k; ++s) {
                std::vector<double> row; std::string scratch_body = body; std::set<int> scratch_temps = temps;
                body.clear(); cvec = &row;
                for (size_t q = st[s].b; q < st[s].e; ++q) ins(v[q]);
                body = scratch_body; temps = scratch_temps;
                if (row.size() != K) same_k = false;
                tab.insert(tab.end(), row.begin(), row.end());
            }
            cvec = nullptr; rolled_term = false; ring_term = false;
            if (!same_k || !ok || K == 0) {                                   // (cannot happen for equal shapes; stay safe: emit them one by one)
                for (size_t s = i; s < j; ++s) for (size_t q = st[s].b; q < st[s].e; ++q) ins(v[q]);
                i = j; continue;
            }
            const std::string name = "(fg_jit_ctab() + " + std::to_string(ctabs->add(tab)) + ")";
            if (share) {
                const FgIns &last = v[st[i].e - 1];
                const uint32_t lc = FG_INS_OPCODE(last.op);
                const char *accn = lc == FG_OP_FACTOR ? "fc" : ((lc == FG_OP_CONSTLIK || (last.op & FG_F_OBSERVE)) ? "lk" : "pr");
                add("{ int buf = 0;");
                add("for (int r0 = 0; r0 < " + std::to_string(R) + "; r0 += FG_JIT_CH, buf ^= 1) { const int rn = (" + std::to_string(R) + " - r0) < FG_JIT_CH ? (" + std::to_string(R) + " - r0) : FG_JIT_CH;");
                add("for (int rr = wv; rr < rn; rr += W) { const FG_JIT_AS4 double *c = " + name + " + (size_t)(r0 + rr) * " + std::to_string(K) + ";");
                body += code;
                add("}");
                add("__syncthreads();");
                add(std::string("if (wv == 0) for (int rr = 0; rr < rn; ++rr) ") + accn + " += ring[(buf * FG_JIT_CH + rr) * FG_WAVE];");
                add("}");
                add("__syncthreads(); }");
            } else {
                if (coop) add("if (wv == 0) {");
                add("#pragma unroll 2");
                add("for (int r = 0; r < " + std::to_string(R) + "; ++r) { const FG_JIT_AS4 double *c = " + name + " + (size_t)r * " + std::to_string(K) + ";");
                body += code;
                add("}");
                if (coop) add("}");
            }
            if (term >= 0) term += (int)R;
            i = j;
        }
    }

    // ---- FG_GRAD_ANALYTIC for any program (opt-in; the reference has only the finite difference): forward-mode derivative of a
    // coordinate's sub-program with respect to that coordinate.  Every instruction carries (value, d value / d q_k): acc / dacc, tN / dtN;
    // the coordinate's own slot has derivative 1, every other cell and constant 0; a statement adds d lp = fg_dlogpdf(...) to dtot.
    // Values are the interpreter's operations in its order; the derivatives are the textbook rules.
    int ad_slot = -1;                            // >= 0: emit derivative code (ins -> ins_ad)
    std::string dslot(uint32_t idx) {
        if ((int)idx == p.n_slots - 1) return "0.0";
        if ((int)idx < (int)p.site_vtype.size()) return (int)idx == ad_slot ? "1.0" : "0.0";
        temps.insert((int)idx);
        return "dt" + std::to_string(idx);
    }
    std::string dopnd(uint32_t w) { return FG_OPND_KIND(w) == FG_OPND_SLOT_F ? dslot(FG_OPND_IDX(w)) : std::string("0.0"); }
    void ins_ad(const FgIns &I) {
        const uint32_t op = I.op, code = FG_INS_OPCODE(op);
        if (code == FG_OP_NORMAL_FAST) {
            const std::string dxv = dslot(I.opnd[0]), dmv = dslot(I.opnd[1]);
            // (the constants are named whether or not the statement moves with the coordinate: rolled statements share one text and one table layout)
            const std::string i0 = lit(I.imm[0]), i1 = lit(I.imm[1]), sg = lit(I.imm[2]), h4 = lit(I.h[4]);
            if (dxv == "0.0" && dmv == "0.0") return;
            const std::string z = (op & FG_F_POW2SCALE) ? "dl * " + h4 : "dl / " + sg;
            const std::string dz = (op & FG_F_POW2SCALE) ? "(" + dxv + " - " + dmv + ") * " + h4 : "(" + dxv + " - " + dmv + ") / " + sg;
            add("{ const double xv = " + i0 + " + " + slot(I.opnd[0]) + "; const double mv = " + i1 + " + " + slot(I.opnd[1]) + "; const double dl = xv - mv; const double z = " + z +
                "; dtot += -z * (" + dz + "); }");
            return;
        }
        if (code < 17u) {
            const bool invalid = (op & FG_F_INVALID) != 0u;
            const uint32_t vtype = FG_INS_VTYPE(op), xw = I.opnd[0];
            if (code == 3u) {                                                 // Categorical: lp = ln p[x]; a constant table has no derivative
                const uint32_t bw = I.opnd[1]; const int K = (int)I.opnd[2];
                if (FG_OPND_KIND(bw) == FG_OPND_POOL) return;
                const int base = (int)FG_OPND_IDX(bw);
                std::string xi = FG_OPND_KIND(xw) == FG_OPND_SLOT_I ? "fg_as_i64(" + slot(FG_OPND_IDX(xw)) + ")"
                                                                      : "fg_jit_int_of(" + opnd(xw, I.imm[0]) + ", " + std::to_string(vtype) + "u)";
                if (invalid) { add("dtot = NAN;"); return; }
                std::string dpick = "double dpv = " + dslot((uint32_t)base) + "; ";
                for (int q = 1; q < K; ++q) dpick += "dpv = (j == " + std::to_string(q) + ") ? " + dslot((uint32_t)(base + q)) + " : dpv; ";
                add("{ const long long xi = " + xi + "; const bool oob = xi < 0 || xi >= " + std::to_string(K) + "LL; const int j = oob ? 0 : (int)xi; " + pick(base, K, "j", "pv") + dpick +
                    "dtot += (oob || !(pv > 0.0)) ? NAN : dpv / pv; }");
                return;
            }
            const std::string p0 = opnd(I.opnd[1], I.imm[1]), p1 = opnd(I.opnd[2], I.imm[2]), p2 = opnd(I.opnd[3], I.imm[3]);
            (void)lit(I.h[0]); (void)lit(I.h[1]); (void)lit(I.h[2]); (void)lit(I.h[3]); (void)lit(I.h[4]);     // (the value code's table layout)
            const std::string dx = vtype == 0u ? dopnd(xw) : std::string("0.0"), d0 = dopnd(I.opnd[1]), d1 = dopnd(I.opnd[2]), d2 = dopnd(I.opnd[3]);
            std::string xs;
            if (vtype == 0u) xs = "const double xf = " + opnd(xw, I.imm[0]) + "; const long long xi = 0;";
            else if (FG_OPND_KIND(xw) == FG_OPND_SLOT_I) xs = "const double xf = 0.0; const long long xi = fg_as_i64(" + slot(FG_OPND_IDX(xw)) + ");";
            else xs = "const double xf = 0.0; const long long xi = fg_jit_int_of(" + opnd(xw, I.imm[0]) + ", " + std::to_string(vtype) + "u);";
            if (dx == "0.0" && d0 == "0.0" && d1 == "0.0" && d2 == "0.0") return;
            if (invalid) { add("dtot = NAN;"); return; }
            char sig[64];
            std::snprintf(sig, sizeof sig, "fg_jit_dlp_%u", code);
            if (!lp_fns->count(sig)) {
                char def[512];
                std::snprintf(def, sizeof def,
                              "static __device__ FG_JIT_CALL double %s(double xf, long long xi, double p0, double p1, double p2, double dx, double d0, double d1, double d2) {\n"
                              "    return fg_dlogpdf(%uu, xf, xi, p0, p1, p2, dx, d0, d1, d2);\n}\n", sig, code);
                (*lp_fns)[sig] = def;
            }
            add("{ " + xs + " dtot += " + sig + "(xf, xi, " + p0 + ", " + p1 + ", " + p2 + ", " + dx + ", " + d0 + ", " + d1 + ", " + d2 + "); }");
            return;
        }
        const std::string x0 = opnd(I.opnd[0], I.imm[0]), dx0 = dopnd(I.opnd[0]);
        switch (code) {
        case FG_OP_FACTOR: if (dx0 != "0.0") add("dtot += " + dx0 + ";"); break;
        case FG_OP_LOAD: add("acc = " + x0 + "; dacc = " + dx0 + ";"); break;
        case FG_OP_ADD: add("acc = acc + " + x0 + "; dacc = dacc + " + dx0 + ";"); break;
        case FG_OP_SUB: add("acc = acc - " + x0 + "; dacc = dacc - " + dx0 + ";"); break;
        case FG_OP_MUL: add("{ const double a_ = acc, b_ = " + x0 + "; acc = a_ * b_; dacc = dacc * b_ + a_ * " + dx0 + "; }"); break;
        case FG_OP_DIV: add("{ const double b_ = " + x0 + "; acc = acc / b_; dacc = (dacc - acc * " + dx0 + ") / b_; }"); break;
        case FG_OP_RSUB: add("acc = " + x0 + " - acc; dacc = " + dx0 + " - dacc;"); break;
        case FG_OP_RDIV: add("{ const double a_ = acc; acc = " + x0 + " / a_; dacc = (" + dx0 + " - acc * dacc) / a_; }"); break;
        case FG_OP_NEG: add("acc = -acc; dacc = -dacc;"); break;
        case FG_OP_EXP: add("acc = fg_jit_exp(acc); dacc = acc * dacc;"); break;
        case FG_OP_LN: add("dacc = dacc / acc; acc = fg_jit_log(acc);"); break;
        case FG_OP_SQRT: add("acc = sqrt(acc); dacc = dacc / (2.0 * acc);"); break;
        case FG_OP_ABS: add("dacc = acc < 0.0 ? -dacc : dacc; acc = fabs(acc);"); break;
        case FG_OP_FLOOR: add("acc = floor(acc); dacc = 0.0;"); break;
        case FG_OP_SIN: add("{ const double a_ = acc; acc = fg_jit_sin(a_); dacc = fg_jit_cos(a_) * dacc; }"); break;
        case FG_OP_COS: add("{ const double a_ = acc; acc = fg_jit_cos(a_); dacc = -fg_jit_sin(a_) * dacc; }"); break;
        case FG_OP_TANH: add("acc = fg_jit_tanh(acc); dacc = (1.0 - acc * acc) * dacc;"); break;
        case FG_OP_POW:                                                   // acc ^ x0
            if (dx0 == "0.0") add("{ const double a_ = acc, b_ = " + x0 + "; acc = fg_jit_pow(a_, b_); dacc = b_ * fg_jit_pow(a_, b_ - 1.0) * dacc; }");
            else add("{ const double a_ = acc, b_ = " + x0 + "; acc = fg_jit_pow(a_, b_); dacc = acc * (" + dx0 + " * fg_jit_log(a_) + b_ * dacc / a_); }");
            break;
        case FG_OP_RPOW:                                                  // x0 ^ acc
            add("{ const double a_ = acc, b_ = " + x0 + "; acc = fg_jit_pow(b_, a_); dacc = acc * (dacc * fg_jit_log(b_) + a_ * " + dx0 + " / b_); }");
            break;
        case FG_OP_MIN: add("{ const double b_ = " + x0 + "; dacc = (acc <= b_) ? dacc : " + dx0 + "; acc = fmin(acc, b_); }"); break;
        case FG_OP_MAX: add("{ const double b_ = " + x0 + "; dacc = (acc >= b_) ? dacc : " + dx0 + "; acc = fmax(acc, b_); }"); break;
        case FG_OP_CLAMP: add("{ const double lo_ = " + x0 + ", hi_ = " + opnd(I.opnd[1], I.imm[1]) + "; dacc = acc < lo_ ? " + dx0 + " : (acc > hi_ ? " + dopnd(I.opnd[1]) + " : dacc); acc = fg_clamp(acc, lo_, hi_); }"); break;
        case FG_OP_MAC: add("{ const double a_ = " + x0 + ", b_ = " + opnd(I.opnd[1], I.imm[1]) + "; const double t_ = a_ * b_; acc = acc + t_; dacc = dacc + (" + dx0 + " * b_ + a_ * " + dopnd(I.opnd[1]) + "); }"); break;
        case FG_OP_STORE: { const std::string t = slot(I.aux); if (t == "0.0" || t == "pert" || t[0] == 's') { ok = false; break; } add(t + " = acc; d" + t + " = dacc;"); break; }
        case FG_OP_GATHER: {
            const int K = (int)I.opnd[1];
            std::string dpick = "double dgv = " + dslot((uint32_t)I.aux) + "; ";
            for (int q = 1; q < K; ++q) dpick += "dgv = (j == " + std::to_string(q) + ") ? " + dslot((uint32_t)((int)I.aux + q)) + " : dgv; ";
            add("{ const bool ok_ = (acc >= 0.0) && (acc < " + lit((double)K) + ") && (acc == floor(acc)); const int j = ok_ ? (int)acc : 0; " + pick((int)I.aux, K, "j", "gv") + dpick +
                "acc = ok_ ? gv : NAN; dacc = ok_ ? dgv : NAN; }");
            break; }
        case FG_OP_CONSTLIK: (void)lit(I.imm[0]); break;
        case FG_OP_DOT: {
            const int n = (int)I.opnd[1];
            for (int t = 0; t < n; ++t) {
                long long sb; std::memcpy(&sb, &p.pool[(size_t)I.aux + 2 * t], 8);
                const std::string cf = lit(p.pool[(size_t)I.aux + 2 * t + 1]);
                add("acc = acc + " + slot((uint32_t)sb) + " * " + cf + "; dacc = dacc + " + dslot((uint32_t)sb) + " * " + cf + ";");
            }
            break; }
        default: ok = false; break;
        }
    }

    std::string acc_int;                         // acc == (double) of this integer cell (set by a LOAD of an integer slot, valid for the next instruction only)
    void ins(const FgIns &I) {
        if (ad_slot >= 0) { ins_ad(I); return; }
        const std::string acc_int_was = acc_int;
        acc_int.clear();
        const uint32_t op = I.op, code = FG_INS_OPCODE(op);
        const bool observe = (op & FG_F_OBSERVE) != 0u;
        const bool ends = code == FG_OP_NORMAL_FAST || code < 17u;
        const std::string accum = ((term >= 0 || ring_term) && ends) ? term_row() + " = lp;" : (observe ? "lk += lp;" : "pr += lp;");
        if (code == FG_OP_NORMAL_FAST) {                                     // fg_interp.h: the fast Normal of score-only programs
            std::string z;
            if (op & FG_F_POW2SCALE) z = "double z = dl * " + lit(I.h[4]) + ";";
            else if (op & FG_F_RCPSCALE) z = "double z = fg_div_const(dl, " + lit(I.imm[2]) + ", " + lit(I.h[4]) + ");";
            else z = "double z = dl / " + lit(I.imm[2]) + ";";
            // operand = imm + slot (fg_interp.h).  Outside rolled runs (whose statements share one text): `0.0 + x` is x up to the sign of a
            // zero, which the density cannot see -- x and mu enter through (x - mu)^2 -- so a zero immediate is not added, and a constant
            // operand (the always-zero slot) is one literal.  mh_terms (the multi-wave MH kernel's term rows): no "NaN -> -inf" select -- a NaN
            // term makes log_alpha NaN where -inf makes it -inf, both reject, and the rows are never stored (fg_mh_mw_body.h).
            auto operand = [&](double imm, uint32_t sl) -> std::string {
                if (cvec) return lit(imm) + " + " + slot(sl);
                if ((int)sl == p.n_slots - 1) return ::lit(imm + 0.0);
                if (imm == 0.0) return slot(sl);
                return lit(imm) + " + " + slot(sl);
            };
            add("{ const double xv = " + operand(I.imm[0], I.opnd[0]) + "; const double mv = " + operand(I.imm[1], I.opnd[1]) +
                "; const double dl = xv - mv; " + z + " double lp = -0.5 * z * z - " + lit(I.h[0]) + " - 0.5 * FG_LN_2PI; " +
                ((mh_terms && !cvec) ? std::string() : std::string("lp = (z != z) ? FG_NEG_INF : lp; ")) + accum + " }");
            return;
        }
        if (code < 17u) {
            const bool hoisted = (op & FG_F_HOISTED) != 0u, invalid = (op & FG_F_INVALID) != 0u;
            const uint32_t vtype = FG_INS_VTYPE(op), xw = I.opnd[0];
            if (prior && !observe) {                                          // the draw (fg_exec's FG_MODE_PRIOR): the value cell, then the statement scores it
                const std::string cell = slot(I.aux);
                if (cell.rfind("slots[", 0) != 0) { ok = false; return; }
                if (code == 3u) {                                             // first i with cumulative[i] >= u, clamped to K - 1
                    const uint32_t bw = I.opnd[1]; const int K = (int)I.opnd[2];
                    const bool in_pool = FG_OPND_KIND(bw) == FG_OPND_POOL; const int base = (int)FG_OPND_IDX(bw);
                    std::string c = "{ const double u = fg_rng_u01(rng); double cum = 0.0; int idx = " + std::to_string(K) + "; ";
                    for (int i = 0; i < K; ++i)
                        c += "cum += " + (in_pool ? lit(p.pool[(size_t)base + i]) : slot((uint32_t)(base + i))) + "; if (idx == " + std::to_string(K) + " && !(cum < u)) idx = " + std::to_string(i) + "; ";
                    add(c + cell + " = fg_as_double((long long)(idx < " + std::to_string(K - 1) + " ? idx : " + std::to_string(K - 1) + ")); }");
                } else
                    add("{ const long long cell_ = fg_jit_sample(" + std::to_string(code) + "u, " + (hoisted ? "true" : "false") + ", " + opnd(I.opnd[1], I.imm[1]) + ", " + opnd(I.opnd[2], I.imm[2]) + ", " +
                        opnd(I.opnd[3], I.imm[3]) + ", rng); " + cell + " = fg_as_double(cell_); }");
            }
            if (code == 3u) {                                                 // Categorical: distribution.rs:771-791
                const uint32_t bw = I.opnd[1]; const int K = (int)I.opnd[2];
                const bool in_pool = FG_OPND_KIND(bw) == FG_OPND_POOL; const int base = (int)FG_OPND_IDX(bw);
                std::string xi = FG_OPND_KIND(xw) == FG_OPND_SLOT_I ? "fg_as_i64(" + slot(FG_OPND_IDX(xw)) + ")"
                                                                      : "fg_jit_int_of(" + opnd(xw, I.imm[0]) + ", " + std::to_string(vtype) + "u)";
                if (invalid) { add("{ double lp = FG_NEG_INF; " + accum + " }"); return; }
                if (in_pool) {
                    std::string tn = "fg_jit_tab" + std::to_string(tables->size());
                    std::string t = "static __device__ const double " + tn + "[" + std::to_string(K) + "] = {";
                    for (int q = 0; q < K; ++q) t += (q ? ", " : "") + lit(p.pool[(size_t)base + K + q]);
                    tables->push_back(t + "};");
                    add("{ const long long xi = " + xi + "; double lp = (xi < 0 || xi >= " + std::to_string(K) + "LL) ? FG_NEG_INF : " + tn + "[(xi < 0 || xi >= " +
                        std::to_string(K) + "LL) ? 0 : (int)xi]; " + accum + " }");
                } else {
                    add("{ const long long xi = " + xi + "; const bool oob = xi < 0 || xi >= " + std::to_string(K) + "LL; const int j = oob ? 0 : (int)xi; " +
                        pick(base, K, "j", "pv") + "double lp = oob ? FG_NEG_INF : (pv > 0.0 ? log(pv) : FG_NEG_INF); " + accum + " }");
                }
                return;
            }
            const std::string p0 = opnd(I.opnd[1], I.imm[1]), p1 = opnd(I.opnd[2], I.imm[2]), p2 = opnd(I.opnd[3], I.imm[3]);
            std::string xs;
            if (vtype == 0u) xs = "const double xf = " + opnd(xw, I.imm[0]) + "; const long long xi = 0;";
            else if (FG_OPND_KIND(xw) == FG_OPND_SLOT_I) xs = "const double xf = 0.0; const long long xi = fg_as_i64(" + slot(FG_OPND_IDX(xw)) + ");";
            else xs = "const double xf = 0.0; const long long xi = fg_jit_int_of(" + opnd(xw, I.imm[0]) + ", " + std::to_string(vtype) + "u);";
            if (invalid) { add("{ double lp = FG_NEG_INF; " + accum + " }"); return; }
            if (code == 12u && hoisted) {                                     // Normal with constant parameters, inline as in fg_interp.h
                const std::string z = (op & FG_F_POW2SCALE) ? "(xf - p0) * " + lit(I.h[4]) : std::string("(xf - p0) / p1");
                add("{ " + xs + " const double p0 = " + p0 + ", p1 = " + p1 + "; (void)p1; (void)xi; double lp; if (!fg_finite(xf)) lp = FG_NEG_INF; else { const double z = " + z +
                    "; lp = -0.5 * z * z - " + lit(I.h[0]) + " - 0.5 * FG_LN_2PI; } " + accum + " }");
                return;
            }
            const bool pow2 = (op & FG_F_POW2SCALE) != 0u, sh = (op & FG_F_SCALEHOIST) != 0u, xh = (op & FG_F_XHOIST) != 0u;
            char sig[96];
            std::snprintf(sig, sizeof sig, "fg_jit_lp_%u_%d%d%d%d", code, (int)hoisted, (int)pow2, (int)sh, (int)xh);
            if (!lp_fns->count(sig)) {
                char def[768];
                std::snprintf(def, sizeof def,
                              "static __device__ FG_JIT_CALL double %s(double xf, long long xi, double p0, double p1, double p2, double h0, double h1, double h2, double h3, double h4) {\n"
                              "    const double hh[5] = { h0, h1, h2, h3, h4 };\n    return fg_logpdf(%uu, %s, %s, FG_JIT_OPQ(xf), xi, FG_JIT_OPQ(p0), FG_JIT_OPQ(p1), FG_JIT_OPQ(p2), hh, %s, %s);\n}\n",
                              sig, code, hoisted ? "true" : "false", pow2 ? "true" : "false", sh ? "true" : "false", xh ? "true" : "false");
                (*lp_fns)[sig] = def;
            }
            add("{ " + xs + " double lp = " + sig + "(xf, xi, " + p0 + ", " + p1 + ", " + p2 + ", " + lit(I.h[0]) + ", " + lit(I.h[1]) + ", " + lit(I.h[2]) + ", " + lit(I.h[3]) + ", " +
                lit(I.h[4]) + "); " + accum + " }");
            return;
        }
        const std::string x0 = opnd(I.opnd[0], I.imm[0]);
        switch (code) {
        case FG_OP_FACTOR: add((term >= 0 || ring_term) ? term_row() + " = " + x0 + ";" : "fc += " + x0 + ";"); break;
        case FG_OP_LOAD:
            add("acc = " + x0 + ";");
            if (FG_OPND_KIND(I.opnd[0]) == FG_OPND_SLOT_I) { acc_int = slot(FG_OPND_IDX(I.opnd[0])); return; }     // (a GATHER that follows indexes with the integer itself)
            break;
        case FG_OP_ADD: add("acc = acc + " + x0 + ";"); break;
        case FG_OP_SUB: add("acc = acc - " + x0 + ";"); break;
        case FG_OP_MUL: add("acc = acc * " + x0 + ";"); break;
        case FG_OP_DIV: add("acc = acc / " + x0 + ";"); break;
        case FG_OP_RSUB: add("acc = " + x0 + " - acc;"); break;
        case FG_OP_RDIV: add("acc = " + x0 + " / acc;"); break;
        case FG_OP_NEG: add("acc = -acc;"); break;
        case FG_OP_EXP: add("acc = fg_jit_exp(acc);"); break;
        case FG_OP_LN: add("acc = fg_jit_log(acc);"); break;
        case FG_OP_SQRT: add("acc = sqrt(acc);"); break;
        case FG_OP_ABS: add("acc = fabs(acc);"); break;
        case FG_OP_FLOOR: add("acc = floor(acc);"); break;
        case FG_OP_SIN: add("acc = fg_jit_sin(acc);"); break;
        case FG_OP_COS: add("acc = fg_jit_cos(acc);"); break;
        case FG_OP_TANH: add("acc = fg_jit_tanh(acc);"); break;
        case FG_OP_POW: add("acc = fg_jit_pow(acc, " + x0 + ");"); break;
        case FG_OP_RPOW: add("acc = fg_jit_pow(" + x0 + ", acc);"); break;
        case FG_OP_MIN: add("acc = fmin(acc, " + x0 + ");"); break;
        case FG_OP_MAX: add("acc = fmax(acc, " + x0 + ");"); break;
        case FG_OP_CLAMP: add("acc = fg_clamp(acc, " + x0 + ", " + opnd(I.opnd[1], I.imm[1]) + ");"); break;
        case FG_OP_MAC: add("{ const double t_ = " + x0 + " * " + opnd(I.opnd[1], I.imm[1]) + "; acc = acc + t_; }"); break;
        case FG_OP_STORE: { const std::string t = slot(I.aux); if (t == "0.0" || t == "pert" || t[0] == 's') { ok = false; break; } add(t + " = acc;"); break; }
        case FG_OP_GATHER: {
            const int K = (int)I.opnd[1];
            // the index: acc >= 0, < K and integral (fg_interp.h).  When acc is an integer site's cell converted to double (the LOAD just
            // before), the same three tests on the integer: no conversion there and back (exact for every 64-bit value: K <= 64)
            if (!acc_int_was.empty() && !cvec)
                add("{ const long long ji_ = fg_as_i64(" + acc_int_was + "); const bool ok_ = (unsigned long long)ji_ < " + std::to_string(K) + "ull; const int j = ok_ ? (int)ji_ : 0; " +
                    pick((int)I.aux, K, "j", "gv") + "acc = ok_ ? gv : NAN; }");
            else
            add("{ const bool ok_ = (acc >= 0.0) && (acc < " + lit((double)K) + ") && (acc == floor(acc)); const int j = ok_ ? (int)acc : 0; " + pick((int)I.aux, K, "j", "gv") +
                "acc = ok_ ? gv : NAN; }");
            break; }
        case FG_OP_CONSTLIK: { const std::string v = lit(I.imm[0]); add((term >= 0 || ring_term) ? term_row() + " = " + v + ";" : "lk += " + v + ";"); break; }
        case FG_OP_DOT: {                                                    // acc = (..((acc + s_0 c_0) + s_1 c_1)..): one product, one sum per term
            const int n = (int)I.opnd[1];
            for (int t = 0; t < n; ++t) {
                long long sb; std::memcpy(&sb, &p.pool[(size_t)I.aux + 2 * t], 8);
                add("acc = acc + " + slot((uint32_t)sb) + " * " + lit(p.pool[(size_t)I.aux + 2 * t + 1]) + ";");
            }
            break; }
        default: ok = false; break;
        }
    }
    std::string decls() const {
        std::string s = "    double acc = 0.0, pr = 0.0, lk = 0.0, fc = 0.0;\n";
        if (ad_slot >= 0) s += "    double dacc = 0.0, dtot = 0.0;\n";
        for (int t : temps) s += "    double t" + std::to_string(t) + " = 0.0;\n" + (ad_slot >= 0 ? "    double dt" + std::to_string(t) + " = 0.0;\n" : std::string());
        return s;
    }
};

// ---- hiprtc, bound at run time ---------------------------------------------------------------------------------------------
struct Rtc {
    void *h = nullptr;
    int (*create)(void **, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(void *, int, const char **) = nullptr;
    int (*log_size)(void *, size_t *) = nullptr;
    int (*log)(void *, char *) = nullptr;
    int (*code_size)(void *, size_t *) = nullptr;
    int (*code)(void *, char *) = nullptr;
    int (*destroy)(void **) = nullptr;
    bool ok = false;
};
Rtc &rtc() {
    static Rtc R;
#ifndef FG_JIT_NO_HIP
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names;
        Dl_info di;
        if (dladdr((void *)hipGetDeviceCount, &di) && di.dli_fname) {       // the copy beside the HIP runtime this library runs on
            std::string d(di.dli_fname); const size_t s = d.rfind('/');
            if (s != std::string::npos) { names.push_back(d.substr(0, s + 1) + "libhiprtc.so"); names.push_back(d.substr(0, s + 1) + "libhiprtc.so.7"); }
        }
        names.push_back("libhiprtc.so"); names.push_back("/opt/rocm/lib/libhiprtc.so");
        for (const std::string &nm : names) { R.h = dlopen(nm.c_str(), RTLD_NOW | RTLD_LOCAL); if (R.h) break; }
        if (!R.h) return;
        R.create = (decltype(R.create))dlsym(R.h, "hiprtcCreateProgram");
        R.compile = (decltype(R.compile))dlsym(R.h, "hiprtcCompileProgram");
        R.log_size = (decltype(R.log_size))dlsym(R.h, "hiprtcGetProgramLogSize");
        R.log = (decltype(R.log))dlsym(R.h, "hiprtcGetProgramLog");
        R.code_size = (decltype(R.code_size))dlsym(R.h, "hiprtcGetCodeSize");
        R.code = (decltype(R.code))dlsym(R.h, "hiprtcGetCode");
        R.destroy = (decltype(R.destroy))dlsym(R.h, "hiprtcDestroyProgram");
        R.ok = R.create && R.compile && R.log_size && R.log && R.code_size && R.code && R.destroy;
    });
#endif
    return R;
}

const char *PROLOGUE = R"FGJ(
// the tile rows a generated function reads and writes are LDS: through a generic pointer every access of a noinline function is a FLAT
// one (hundreds of cycles, a vmcnt wait each, no batching) -- the qualified pointer makes them ds_read / ds_write with immediate offsets
#define FG_LDSQ __attribute__((address_space(3)))
#define FG_JIT_LDS(p) ((FG_LDSQ double *)(p))
#define FG_BUILD 1
#define FG_JIT_RTC 1
#define FG_WAVE 64
#ifndef INFINITY
#define INFINITY __builtin_huge_val()
#endif
#ifndef NAN
#define NAN __builtin_nan("")
#endif
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define FG_HD __device__ __forceinline__
#ifdef __HIPCC_RTC__              /* hiprtc: no standard headers; under hipcc the HIP wrapper headers bring these */
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned long uintptr_t;
typedef unsigned long size_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
)FGJ";

const char *HELPERS = R"FGJ(
// the transcendental opcodes behind calls, as in fg_interp.h (same ocml functions: same bits)
// FG_JIT_INLINED (experiment, see fg_jit_inlined): densities and transcendentals are inlined into the statements that use them (no call,
// no arguments moved into the callee's registers).  An inlined libm call must still see RUN-TIME arguments: with a literal
// the compiler would evaluate log(0.7) itself, or turn pow(x, 2.0) into x * x -- other bits than ocml's code gives the interpreter.
// fg_jit_opq hands its argument through an empty (not volatile: it may move and merge) asm statement: the value is in a register and opaque to constant propagation.
#ifdef FG_JIT_INLINED
#define FG_JIT_CALL __forceinline__
static __device__ __forceinline__ double fg_jit_opq(double x) { asm("" : "+v"(x)); return x; }
#define FG_JIT_OPQ(x) fg_jit_opq(x)
#else
#define FG_JIT_CALL __noinline__
#define FG_JIT_OPQ(x) (x)
#endif
static __device__ FG_JIT_CALL double fg_jit_exp(double x) { return exp(FG_JIT_OPQ(x)); }
static __device__ FG_JIT_CALL double fg_jit_log(double x) { return log(FG_JIT_OPQ(x)); }
static __device__ FG_JIT_CALL double fg_jit_sin(double x) { return sin(FG_JIT_OPQ(x)); }
static __device__ FG_JIT_CALL double fg_jit_cos(double x) { return cos(FG_JIT_OPQ(x)); }
static __device__ FG_JIT_CALL double fg_jit_tanh(double x) { return tanh(FG_JIT_OPQ(x)); }
static __device__ FG_JIT_CALL double fg_jit_pow(double x, double y) { return pow(FG_JIT_OPQ(x), FG_JIT_OPQ(y)); }
// the module's constant tables (FgJitTabs): one device buffer, its address stored here by the engine after hipModuleLoadData
#define FG_JIT_AS4 __attribute__((address_space(4)))
__device__ const double *fg_jit_ctab_ptr = nullptr;
static __device__ __forceinline__ const FG_JIT_AS4 double *fg_jit_ctab() { return (const FG_JIT_AS4 double *)(unsigned long)fg_jit_ctab_ptr; }
// the 17 samplers behind one call, like the interpreter's fg_sample_cold (the same fg_sample_dist: the same draws)
static __device__ __noinline__ long long fg_jit_sample(uint32_t kind, bool hoisted, double p0, double p1, double p2, FgStream &s) { return fg_sample_dist(kind, hoisted, p0, p1, p2, s); }
// fg_int_of (fg_interp.h): the integer value of an observed expression
static __device__ __forceinline__ long long fg_jit_int_of(double v, unsigned vtype) { if (vtype == 1u) return v != 0.0; return fg_finite(v) ? (long long)v : 0; }
)FGJ";

}  // namespace

// densities and transcendentals inlined into the statements (HELPERS: FG_JIT_INLINED)?  An experiment (FG_JIT_INLINE=1), not the default:
// with arguments the compiler can see through, inlining gained 2 ... 20 % -- by evaluating lgamma(2.0) or log(0.5) at compile time with
// the host's libm, i.e. possibly other bits than the interpreter's; with the arguments made opaque (as here) the same work is done at run
// time in many copies and the kernels are slower than with calls (hier_scale HMC 3.5e9 -> 2.1e9, MH 7.7e9 -> 6.0e9).
static bool fg_jit_inlined() { const char *v = std::getenv("FG_JIT_INLINE"); return v && std::atoi(v) != 0; }

// The generated translation unit of one program's HMC kernel, or "" when the program holds something the generator does not cover.
std::string fg_jit_hmc_source(const fg_program *p, std::vector<double> *ctab_out, bool *has_ad_out, bool *has_dense_out, const std::vector<std::vector<int>> *wave_tasks,
                              const std::vector<std::vector<int>> *wave_coords, const std::vector<std::vector<int>> *wave_coords_dense) {
    std::map<std::string, std::string> lp_fns;
    FgJitTabs ctabs;
    std::vector<std::string> tables;
    std::string fns;
    const int d = (int)p->coord.size();
    for (int k = 0; k < d; ++k) {
        Gen g{*p}; g.pert_slot = p->coord[k].slot; g.lp_fns = &lp_fns; g.tables = &tables;
        g.ctabs = &ctabs;
        g.emit(p->sub, (size_t)p->coord[k].sub_off, (size_t)p->coord[k].sub_off + (size_t)p->coord[k].sub_n);
        if (!g.ok) return "";
        // (the body once, inlinable: the per-wave task lists below inline short sub-programs; everything else calls the out-of-line copy)
        fns += "static __device__ __forceinline__ double fg_jit_subi_" + std::to_string(k) + "(double pert, const FG_LDSQ double *slots) {\n" + g.decls() + g.body +
               "    (void)acc;\n    return pr + lk + fc;\n}\n"
               "static __device__ __noinline__ double fg_jit_sub_" + std::to_string(k) + "(double pert, const FG_LDSQ double *slots) { return fg_jit_subi_" + std::to_string(k) + "(pert, slots); }\n";
    }
    // FG_GRAD_ANALYTIC: d/dq_k of the same sub-program (forward mode; Gen::ins_ad).  A program the derivative emission does not cover keeps
    // the finite difference only (the engine then refuses the analytic mode for it, as before).
    bool has_ad = true;
    std::string dfns;
    for (int k = 0; k < d && has_ad; ++k) {
        Gen g{*p}; g.ad_slot = p->coord[k].slot; g.lp_fns = &lp_fns; g.tables = &tables;
        g.ctabs = &ctabs;
        g.emit(p->sub, (size_t)p->coord[k].sub_off, (size_t)p->coord[k].sub_off + (size_t)p->coord[k].sub_n);
        if (!g.ok) { has_ad = false; break; }
        dfns += "static __device__ __noinline__ double fg_jit_dsub_" + std::to_string(k) + "(const FG_LDSQ double *slots) {\n" + g.decls() + g.body +
                "    (void)acc; (void)pr; (void)lk; (void)fc; (void)dacc;\n    return dtot;\n}\n";
    }
    if (has_ad_out) *has_ad_out = has_ad;
    if (has_ad) {
        fns += dfns;
        fns += "#define FG_JIT_HAS_AD 1\nstatic __device__ __forceinline__ double fg_jit_dtask(int k, const FG_LDSQ double *slots) {\n    switch (k) {\n";
        for (int k = 0; k < d; ++k) fns += "    case " + std::to_string(k) + ": return fg_jit_dsub_" + std::to_string(k) + "(slots);\n";
        fns += "    default: return 0.0;\n    }\n}\n";
    }
    // The tasks of every wave as straight-line code (the engine's launches all use ONE split: fg_hmc_interp.hip generates the unit behind it): no task
    // list in memory, no dispatch on the coordinate, and a short sub-program (<= 64 instructions) inlined -- its LDS reads overlap the previous task's
    // arithmetic.  Per task the operations are fg_jit_sub_k's: identical results.
    fns += "#define FG_JIT_K_D " + std::to_string(d) + "\n#define FG_JIT_K_S " + std::to_string(p->sorted_stmt.size()) + "\n";      // (the kernels' row offsets and loop bounds as literals)
    if (wave_tasks && !wave_tasks->empty() && wave_tasks->size() <= 16) {
        const char *im = std::getenv("FG_JIT_TASK_INLINE");
        const int inline_max = im ? std::atoi(im) : 64;                  // (16 and 64 measured alike but for linreg at 8 192 chains: +8 % with 64)
        fns += "#define FG_JIT_BAKED_W " + std::to_string(wave_tasks->size()) + "\nstatic __device__ __forceinline__ void fg_jit_wave_tasks(int wv, double h, const FG_LDSQ double *slots, FG_LDSQ double *ev) {\n    switch (wv) {\n";
        for (size_t w = 0; w < wave_tasks->size(); ++w) {
            fns += "    case " + std::to_string(w) + ": {\n";
            for (int task : (*wave_tasks)[w]) {
                const int k = task >> 1;
                if (k < 0 || k >= d) return "";
                fns += "        ev[" + std::to_string(task) + " * FG_WAVE] = fg_jit_sub" + (p->coord[k].sub_n <= inline_max ? "i_" : "_") + std::to_string(k) + "(slots[" + std::to_string(k) + " * FG_WAVE] " +
                       ((task & 1) ? "-" : "+") + " h, slots);\n";
            }
            fns += "    } break;\n";
        }
        fns += "    default: break;\n    }\n}\n";
    }
    // Whole coordinates per wave (fg_hmc_jit_body.h's one-barrier gradient): both evaluations of a coordinate, its force component, the kick and the
    // drift in one piece -- log-joints stay in registers, the new position goes to the OTHER copy of the site rows (the positions every wave still reads
    // stay untouched), so a gradient ends in ONE barrier.  Per coordinate the operations are the two-barrier loop's, in its order.
    if (wave_coords && !wave_coords->empty() && wave_coords->size() <= 16) {
        const char *im = std::getenv("FG_JIT_TASK_INLINE");
        const int inline_max = im ? std::atoi(im) : 64;
        fns += "#define FG_JIT_FUSED_W " + std::to_string(wave_coords->size()) + "\n"
               "#define FG_JIT_GRAD_COORD(K, SUB) { const double q_ = cur[(K) * FG_WAVE]; const double tp_ = SUB(q_ + h, cur); const double tm_ = SUB(q_ - h, cur); \\\n"
               "        const double g_ = (tp_ - tm_) / (2.0 * h); bad = bad || !fg_finite(g_); double p_ = pl[(K) * FG_WAVE]; p_ += hk * g_; if (two_kicks) p_ += hk * g_; pl[(K) * FG_WAVE] = p_; \\\n"
               "        if (drift) { const double mk_ = mi ? mi[(long long)(K) * XC] : 1.0; alt[(K) * FG_WAVE] = q_ + e * mk_ * p_; } }\n"
               "static __device__ __forceinline__ bool fg_jit_wave_grad(int wv, double h, double hk, double e, bool two_kicks, bool drift, const FG_LDSQ double *cur, FG_LDSQ double *alt,\n"
               "                                                        FG_LDSQ double *pl, const double *mi, long long XC) {\n    bool bad = false;\n    switch (wv) {\n";
        std::vector<char> seen((size_t)d, 0);
        for (size_t w = 0; w < wave_coords->size(); ++w) {
            fns += "    case " + std::to_string(w) + ": {\n";
            for (int k : (*wave_coords)[w]) {
                if (k < 0 || k >= d || seen[(size_t)k]) return "";
                seen[(size_t)k] = 1;
                fns += "        FG_JIT_GRAD_COORD(" + std::to_string(k) + ", fg_jit_sub" + (p->coord[k].sub_n <= inline_max ? "i_" : "_") + std::to_string(k) + ")\n";
            }
            fns += "    } break;\n";
        }
        for (int k = 0; k < d; ++k) if (!seen[(size_t)k]) return "";
        fns += "    default: break;\n    }\n    return bad;\n}\n";
    }
    fns += "static __device__ __forceinline__ double fg_jit_task(int k, double pert, const FG_LDSQ double *slots) {\n    switch (k) {\n";
    for (int k = 0; k < d; ++k) fns += "    case " + std::to_string(k) + ": return fg_jit_sub_" + std::to_string(k) + "(pert, slots);\n";
    fns += "    default: return 0.0;\n    }\n}\n";
    {
        Gen g{*p}; g.lp_fns = &lp_fns; g.tables = &tables;
        g.ctabs = &ctabs;
        g.emit(p->ins_fast, 0, (size_t)p->n_ins);
        if (!g.ok) return "";
        fns += "static __device__ __noinline__ void fg_jit_score(const FG_LDSQ double *slots, double &pr_out, double &lk_out, double &fc_out) {\n    const double pert = 0.0; (void)pert;\n" + g.decls() + g.body +
               "    (void)acc;\n    pr_out = pr; lk_out = lk; fc_out = fc;\n}\n";
    }
    // FG_GRAD_FD_DENSE (grad_log_joint verbatim, hmc.rs:304-329): the WHOLE program at q +- h e_k, one function per coordinate with the
    // reads of its slot replaced (plates stay rolled: their constant tables are shared between the d copies).  Only while d copies of the
    // program stay compilable in seconds; beyond that the dense mode keeps the interpreter kernels.
    bool has_dense = false;
    {
        size_t n_stmt = 0;
        for (int k = 0; k < p->n_ins; ++k) if (Gen::ends_statement(p->ins_fast[(size_t)k])) ++n_stmt;
        if (d >= 1 && (size_t)d * (size_t)p->n_ins <= 24000 && !(std::getenv("FG_JIT_DENSE") && std::atoi(std::getenv("FG_JIT_DENSE")) == 0)) {
            std::string ffns;
            has_dense = true;
            for (int k = 0; k < d && has_dense; ++k) {
                Gen g{*p}; g.pert_slot = p->coord[k].slot; g.lp_fns = &lp_fns; g.tables = &tables; g.ctabs = &ctabs;
                g.emit(p->ins_fast, 0, (size_t)p->n_ins);
                if (!g.ok) { has_dense = false; break; }
                ffns += "static __device__ __noinline__ double fg_jit_full_" + std::to_string(k) + "(double pert, const FG_LDSQ double *slots) {\n" + g.decls() + g.body +
                        "    (void)acc;\n    return pr + lk + fc;\n}\n";
            }
            if (has_dense && fns.size() + ffns.size() > (2u << 20)) has_dense = false;     // (the unit as a whole must stay well inside what compiles in seconds: the sparse mode is the default's)
            if (has_dense) {
                fns += ffns;
                fns += "#define FG_JIT_HAS_DENSE 1\nstatic __device__ __forceinline__ double fg_jit_dense_task(int k, double pert, const FG_LDSQ double *slots) {\n    switch (k) {\n";
                for (int k = 0; k < d; ++k) fns += "    case " + std::to_string(k) + ": return fg_jit_full_" + std::to_string(k) + "(pert, slots);\n";
                fns += "    default: return 0.0;\n    }\n}\n";
                // ... and the one-barrier gradient of the dense mode: whole coordinates per wave, the two WHOLE log-joints of a coordinate in registers
                if (wave_coords_dense && !wave_coords_dense->empty() && wave_coords_dense->size() <= 16) {
                    std::string gf = "#define FG_JIT_FUSED_DENSE_W " + std::to_string(wave_coords_dense->size()) + "\n"
                                     "#ifndef FG_JIT_GRAD_COORD\n"
                                     "#define FG_JIT_GRAD_COORD(K, SUB) { const double q_ = cur[(K) * FG_WAVE]; const double tp_ = SUB(q_ + h, cur); const double tm_ = SUB(q_ - h, cur); \\\n"
                                     "        const double g_ = (tp_ - tm_) / (2.0 * h); bad = bad || !fg_finite(g_); double p_ = pl[(K) * FG_WAVE]; p_ += hk * g_; if (two_kicks) p_ += hk * g_; pl[(K) * FG_WAVE] = p_; \\\n"
                                     "        if (drift) { const double mk_ = mi ? mi[(long long)(K) * XC] : 1.0; alt[(K) * FG_WAVE] = q_ + e * mk_ * p_; } }\n"
                                     "#endif\n"
                                     "static __device__ __forceinline__ bool fg_jit_wave_grad_dense(int wv, double h, double hk, double e, bool two_kicks, bool drift, const FG_LDSQ double *cur, FG_LDSQ double *alt,\n"
                                     "                                                              FG_LDSQ double *pl, const double *mi, long long XC) {\n    bool bad = false;\n    switch (wv) {\n";
                    std::vector<char> seen((size_t)d, 0);
                    bool okd = true;
                    for (size_t w = 0; w < wave_coords_dense->size() && okd; ++w) {
                        gf += "    case " + std::to_string(w) + ": {\n";
                        for (int k : (*wave_coords_dense)[w]) {
                            if (k < 0 || k >= d || seen[(size_t)k]) { okd = false; break; }
                            seen[(size_t)k] = 1;
                            gf += "        FG_JIT_GRAD_COORD(" + std::to_string(k) + ", fg_jit_full_" + std::to_string(k) + ")\n";
                        }
                        gf += "    } break;\n";
                    }
                    for (int k = 0; k < d; ++k) okd = okd && seen[(size_t)k];
                    if (!okd) return "";
                    fns += gf + "    default: break;\n    }\n    return bad;\n}\n";
                }
            }
        }
    }
    if (has_dense_out) *has_dense_out = has_dense;
    {   // run(PriorHandler) (interpreters.rs:88-104) from the generic program: every sample statement draws, then scores (k_prior_jit)
        Gen g{*p}; g.lp_fns = &lp_fns; g.tables = &tables; g.prior = true;
        for (size_t q = 0; q < (size_t)p->n_ins && q < p->ins.size() && g.ok; ++q) {     // (the program ends with a stop instruction behind n_ins)
            g.ins(p->ins[q]);
            if (!g.ok && std::getenv("FG_JIT_VERBOSE")) fprintf(stderr, "fugue_amd: no compiled prior draw: instruction %zu (opcode %u, flags %#x) is not covered\n", q, FG_INS_OPCODE(p->ins[q].op), p->ins[q].op);
        }
        if (g.ok)
            fns += "#define FG_JIT_HAS_PRIOR 1\nstatic __device__ __noinline__ void fg_jit_prior(FG_LDSQ double *slots, FgStream rng, double &pr_out, double &lk_out, double &fc_out) {\n    const double pert = 0.0; (void)pert;\n" +
                   g.decls() + g.body + "    (void)acc;\n    pr_out = pr; lk_out = lk; fc_out = fc;\n}\n";
    }
    std::string src = PROLOGUE;
    if (fg_jit_inlined()) src += "#define FG_JIT_INLINED 1\n";
    if (const char *oc = std::getenv("FG_HMC_JIT_OCC")) { const int o = std::atoi(oc); if (o >= 2 && o <= 4) src += "#define FG_JIT_OCC " + std::to_string(o) + "\n"; }   // experiments: register budget
    src += FG_JIT_EMBED_HEAD;                    // fg_ir.h, fg_math.h, fg_cold.h, fg_dev_types.h
    src += HELPERS;
    for (const std::string &t : tables) src += t + "\n";
    for (const auto &kv : lp_fns) src += kv.second;
    src += fns;
    src += FG_JIT_EMBED_HMC_BODY;                // fg_hmc_jit_body.h
    if (std::getenv("FG_JIT_BREAK")) src += "\n#error FG_JIT_BREAK: a compilation that fails (tests of the fallback to the interpreter kernels)\n";
    if (ctab_out) *ctab_out = ctabs.data;
    return src;
}

// The generated translation unit of one program's MH kernel: the scoring run as FG_JIT_NSEG statement segments (contiguous, balanced by
// instruction cost at generation time) that the waves of a tile share, behind fg_mh_interp_body.h; the interpreter itself is part
// of the unit for the propose-and-score pass of model-dependent proposals.
std::string fg_jit_mh_source(const fg_program *p, const std::vector<long long> &ins_cost, int occ, std::vector<double> *ctab_out) {
    constexpr int NSEG = 8;
    std::map<std::string, std::string> lp_fns;
    FgJitTabs ctabs;
    std::vector<std::string> tables;
    std::vector<int> stmt_end;
    for (int k = 0; k < p->n_ins; ++k) {
        const uint32_t code = FG_INS_OPCODE(p->ins_fast[(size_t)k].op);
        if (code == FG_OP_NORMAL_FAST || code < 17u || code == FG_OP_FACTOR || code == FG_OP_CONSTLIK) stmt_end.push_back(k + 1);
    }
    const int n_stmt = (int)stmt_end.size();
    if (n_stmt < 1 || stmt_end.back() != p->n_ins) return "";
    std::vector<long long> cum((size_t)n_stmt + 1, 0);
    for (int k = 0, i = 0; k < n_stmt; ++k) { long long cs = 0; for (; i < stmt_end[k]; ++i) cs += ins_cost[(size_t)i]; cum[(size_t)k + 1] = cum[(size_t)k] + cs; }
    std::string fns;
    int s_at = 0;
    for (int sg = 0; sg < NSEG; ++sg) {
        int s_to = n_stmt;
        if (sg + 1 < NSEG) { const long long target = cum[(size_t)n_stmt] * (sg + 1) / NSEG; s_to = s_at; while (s_to < n_stmt && cum[(size_t)s_to] < target) ++s_to; }
        Gen g{*p}; g.lp_fns = &lp_fns; g.tables = &tables; g.term = s_at;
        g.ctabs = &ctabs;
        g.emit(p->ins_fast, (size_t)(s_at > 0 ? stmt_end[(size_t)s_at - 1] : 0), (size_t)(s_to > 0 ? stmt_end[(size_t)s_to - 1] : 0));
        if (!g.ok) return "";
        fns += "static __device__ __noinline__ void fg_jit_seg_" + std::to_string(sg) + "(const FG_LDSQ double *__restrict__ slots, FG_LDSQ double *__restrict__ terms) {\n    const double pert = 0.0; (void)pert;\n" + g.decls() + g.body +
               "    (void)acc; (void)pr; (void)lk; (void)fc;\n}\n";
        s_at = s_to;
    }
    {   // the whole program with the accumulators themselves, entered by every wave: the direct mode of programs with more statements than
        // LDS has term rows (plates shared between the waves through a ring of 2 x FG_JIT_CH rows, everything else on wave 0)
        Gen g{*p}; g.lp_fns = &lp_fns; g.tables = &tables; g.ctabs = &ctabs; g.coop = true;
        g.emit(p->ins_fast, 0, (size_t)p->n_ins);
        if (!g.ok) return "";
        fns += "#define FG_JIT_CH 32\nstatic __device__ __noinline__ void fg_jit_score_coop(const FG_LDSQ double *slots, FG_LDSQ double *ring, int wv, int W, double &pr_out, double &lk_out, double &fc_out) {\n"
               "    const double pert = 0.0; (void)pert;\n" + g.decls() + g.body + "    (void)acc;\n    pr_out = pr; lk_out = lk; fc_out = fc;\n}\n";
    }
    fns += "static __device__ __forceinline__ void fg_jit_terms(int sg, const FG_LDSQ double *slots, FG_LDSQ double *terms) {\n    switch (sg) {\n";
    for (int sg = 0; sg < NSEG; ++sg) fns += "    case " + std::to_string(sg) + ": fg_jit_seg_" + std::to_string(sg) + "(slots, terms); break;\n";
    fns += "    default: break;\n    }\n}\n";
    std::string src = PROLOGUE;
    if (fg_jit_inlined()) src += "#define FG_JIT_INLINED 1\n";
    src += FG_JIT_EMBED_API;                     // include/fugue_amd.h (proposal kinds, error codes)
    src += FG_JIT_EMBED_HEAD;                    // fg_ir.h, fg_math.h, fg_cold.h, fg_dev_types.h
    src += FG_JIT_EMBED_INTERP;                  // fg_interp.h: the propose-and-score mode for model-dependent proposals
    src += HELPERS;
    for (const std::string &t : tables) src += t + "\n";
    for (const auto &kv : lp_fns) src += kv.second;
    src += fns;
    src += "#define FG_JIT_NSEG " + std::to_string(NSEG) + "\n"
           "#define FG_MHI_SCORE() do { for (int sg_ = wv; sg_ < FG_JIT_NSEG; sg_ += W) fg_jit_terms(sg_, FG_JIT_LDS(slots), FG_JIT_LDS(terms)); (void)i0; (void)i1; (void)s0; } while (0)\n"
           "#define FG_MHI_DIRECT_SCORE() fg_jit_score_coop(FG_JIT_LDS(slots), FG_JIT_LDS(terms), wv, W, A.prior, A.lik, A.fac)\n"
           "#define FG_MHI_PRIV_BLOCKS(W) 1\n";
    src += FG_JIT_EMBED_MH_BODY;                 // fg_mh_interp_body.h
    src += R"FGJ(
#define FG_MH_JIT_KERNEL(OCC) \
extern "C" __global__ __attribute__((amdgpu_waves_per_eu(OCC, OCC))) __launch_bounds__(FG_WAVE * FG_MHI_MAX) \
void k_mh_jit_steps_occ##OCC(FgProgramDev P, FgChainCtx X, FgMhDev M, FgMhi seg, int iter0, int n_steps, int n_warmup, long long *draws, int first_sample_t) { \
    fg_mh_interp_mw_body<OCC>(P, X, M, seg, iter0, n_steps, n_warmup, draws, first_sample_t); }
)FGJ";
    src += occ == 4 ? "FG_MH_JIT_KERNEL(4)\n" : "FG_MH_JIT_KERNEL(2)\n";     // one register budget per unit (the launch site knows which: halves the compilation)
    if (std::getenv("FG_JIT_BREAK")) src += "\n#error FG_JIT_BREAK: a compilation that fails (tests of the fallback to the interpreter kernels)\n";
    if (ctab_out) *ctab_out = ctabs.data;
    return src;
}

// The generated translation unit of the multi-wave MH kernel of a score-stream program (fg_mh_mw_body.h: pipelined random numbers, control
// wave, in-order sums, the operand-pattern runs of plain Normal records -- all of it the hand-written kernel's) with the GENERAL
// records of phase B (generated[k] != 0: statement k's log-density term into its LDS row) as FG_JIT_NSEG generated statement
// segments instead of fg_score_one over the record stream.
std::string fg_jit_mhmw_source(const fg_program *p, const std::vector<long long> &ins_cost, const std::vector<char> &generated, int rk, int split, std::vector<double> *ctab_out,
                               const std::vector<int> *rows_in, int n_pri, int n_fac, bool no_stream, bool pipe, int nseg, int ctl_share16, int sum_pri, int sum_lik, const int *baked, int sums_form) {
    // statement segments: sixteen dealt to the waves (sg = wave, wave + W, ...), or -- nseg = the launch's waves per tile -- ONE per wave:
    // a wave's statements are then one straight-line function whose LDS reads are all in flight together
    const int NSEG = (nseg >= 2 && nseg <= 16) ? nseg : 16;
    std::map<std::string, std::string> lp_fns;
    FgJitTabs ctabs;
    std::vector<std::string> tables;
    std::vector<int> stmt_end;
    for (int k = 0; k < p->n_ins; ++k) if (Gen::ends_statement(p->ins_fast[(size_t)k])) stmt_end.push_back(k + 1);
    const int n_stmt = (int)stmt_end.size();
    if (n_stmt < 1 || stmt_end.back() != p->n_ins || (int)generated.size() != n_stmt) return "";
    if (rows_in ? (int)rows_in->size() != n_stmt : n_stmt != p->n_sstream) return "";         // one record per statement, in program order
    std::vector<int> rows((size_t)n_stmt);
    for (int k = 0; k < n_stmt; ++k) rows[(size_t)k] = rows_in ? (*rows_in)[(size_t)k] : (int)p->sstream[(size_t)k].coord;
    std::vector<long long> cum((size_t)n_stmt + 1, 0);
    for (int k = 0, i = 0; k < n_stmt; ++k) {             // work before statement k (the generated statements only)
        long long cs = 0; for (; i < stmt_end[(size_t)k]; ++i) cs += ins_cost[(size_t)i];
        cum[(size_t)k + 1] = cum[(size_t)k] + (generated[(size_t)k] ? cs : 0);
    }
    if (cum[(size_t)n_stmt] == 0) return "";
    auto ins_at = [&](int k) { return (size_t)(k > 0 ? stmt_end[(size_t)k - 1] : 0); };
    // statement -> segment.  The terms go to rows, so the order of evaluation is free: a short program deals its statements to the
    // segments heaviest first, each to the lightest segment so far (one lgamma-based density weighs ten Normals: contiguous cuts
    // leave a wave with two of them); a long one -- plates that roll into loops -- is cut into contiguous runs of equal cost.
    std::vector<int> seg_of((size_t)n_stmt, -1);
    long long heaviest = 0;
    for (int k = 0; k < n_stmt; ++k) heaviest = std::max(heaviest, cum[(size_t)k + 1] - cum[(size_t)k]);
    if (n_stmt <= 256 && heaviest >= 40) {                // (programs of light statements only measure better cut contiguously: linreg 2.56e10 / 2.43e10)
        std::vector<int> order;
        for (int k = 0; k < n_stmt; ++k) if (generated[(size_t)k]) order.push_back(k);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cum[(size_t)a + 1] - cum[(size_t)a] > cum[(size_t)b + 1] - cum[(size_t)b]; });
        // the control wave (segments 0 and 8 at W <= 8) also proposes, adds the terms and decides, and scores at the lower
        // priority: it starts with half a segment on its account (alldists, 65 536 chains: 1.29e9 without, 1.60e9 with; 1.46e9 cut contiguously)
        std::vector<long long> load((size_t)NSEG, 0);
        load[0] = std::min<long long>((45 + n_stmt) / 2, cum[(size_t)n_stmt] / (2 * NSEG));
        if (NSEG > 8) load[8] = load[0];
        for (int k : order) {
            int best = 0;
            for (int sg = 1; sg < NSEG; ++sg) if (load[sg] < load[best]) best = sg;
            seg_of[(size_t)k] = best; load[best] += cum[(size_t)k + 1] - cum[(size_t)k];
        }
    } else {
        // (one segment per wave: the control wave's -- segment 0 -- is `ctl_share` sixteenths of the others', it has the step's path besides)
        const int ctl16 = (nseg >= 2 && nseg <= 16) ? ctl_share16 : 16;
        const long long units = 16LL * (NSEG - 1) + ctl16;
        int s_at = 0;
        long long at_u = 0;
        for (int sg = 0; sg < NSEG; ++sg) {
            at_u += sg == 0 ? ctl16 : 16;
            int s_to = n_stmt;
            if (sg + 1 < NSEG) { const long long target = cum[(size_t)n_stmt] * at_u / units; s_to = s_at; while (s_to < n_stmt && cum[(size_t)s_to] < target) ++s_to; }
            for (int k = s_at; k < s_to; ++k) if (generated[(size_t)k]) seg_of[(size_t)k] = sg;
            s_at = s_to;
        }
    }
    std::string fns;
    for (int sg = 0; sg < NSEG; ++sg) {
        Gen g{*p}; g.lp_fns = &lp_fns; g.tables = &tables; g.rows = &rows; g.ctabs = &ctabs; g.mh_terms = !no_stream;
        for (int k = 0; k < n_stmt;) {                    // maximal runs of consecutive statements of this segment (plates roll within a run)
            if (seg_of[(size_t)k] != sg) { ++k; continue; }
            int k2 = k; while (k2 < n_stmt && seg_of[(size_t)k2] == sg) ++k2;
            g.term = k;
            g.emit(p->ins_fast, ins_at(k), ins_at(k2));
            k = k2;
        }
        if (!g.ok) return "";
        fns += "static __device__ __noinline__ void fg_jit_mhb_" + std::to_string(sg) + "(const FG_LDSQ double *__restrict__ slots, FG_LDSQ double *__restrict__ terms) {\n    const double pert = 0.0; (void)pert;\n" + g.decls() + g.body +
               "    (void)acc; (void)pr; (void)lk; (void)fc;\n}\n";
    }
    // A Categorical site whose table is computed (expressions of other sites): its prior-resample proposal (mh.rs:516-530) needs the model.
    // The multi-wave kernel interprets the target's own statement for such lanes (fg_mhmw_model_proposals) -- on the control wave's path,
    // some lane of nearly every wave, every step.  Here the statement's expressions are generated like any scoring statement (temporaries
    // as locals) and the distribution instruction becomes the interpreter's resample code over those locals: the same operations on the
    // same numbers.  fg_jit_prop(site, ...) returns false for every other site (undecided kinds, PriorResample overrides: interpreted).
    std::string prop_cases;
    if (no_stream)
        for (int k = 0; k < n_stmt; ++k) {
            const size_t a = ins_at(k), b = ins_at(k + 1);
            const FgIns &L = p->ins_fast[b - 1];
            if (FG_INS_OPCODE(L.op) != 3u || (L.op & (FG_F_OBSERVE | FG_F_INVALID)) != 0u) continue;
            if (FG_OPND_KIND(L.opnd[1]) == FG_OPND_POOL || FG_OPND_KIND(L.opnd[0]) != FG_OPND_SLOT_I) continue;        // constant tables are resampled by the random-number wave
            int site = -1;
            for (int j = 0; j < (int)p->site_slot.size(); ++j) if (p->site_slot[(size_t)j] == (int)L.aux) site = j;
            const int K = (int)L.opnd[2], base = (int)FG_OPND_IDX(L.opnd[1]);
            if (site < 0 || K < 1 || K > 64 || (int)FG_OPND_IDX(L.opnd[0]) != (int)L.aux) continue;
            Gen g{*p}; g.lp_fns = &lp_fns; g.tables = &tables; g.ctabs = nullptr;
            for (size_t q = a; q + 1 < b; ++q) g.ins(p->ins_fast[q]);
            if (!g.ok) continue;
            const std::string cell = "slots[" + std::to_string((int)L.aux) + " * FG_WAVE]";
            std::string c = "    FgJitProp out = {0.0, 0.0, 2};\n    if (is_t) {\n        FgStream s1 = rng;\n        const double u = fg_rng_u01(s1);\n        double cum = 0.0; int idx = " + std::to_string(K) + ";\n";
            for (int i = 0; i < K; ++i) c += "        cum += " + g.slot((uint32_t)(base + i)) + "; if (idx == " + std::to_string(K) + " && !(cum < u)) idx = " + std::to_string(i) + ";\n";
            c += "        const long long prop = idx < " + std::to_string(K - 1) + " ? idx : " + std::to_string(K - 1) + ";\n"
                 "        const long long xi = fg_as_i64(" + cell + ");\n"
                 "        const bool cur_ok = !(xi < 0 || xi >= " + std::to_string(K) + "LL);\n"
                 "        const int jp = (int)prop, jc = cur_ok ? (int)xi : 0;\n        " + g.pick(base, K, "jp", "pp") + "\n        " + g.pick(base, K, "jc", "pc0") + "\n"
                 "        const double pc = cur_ok ? pc0 : 0.0;\n"
                 "        out.f = !(pp > 0.0) ? FG_NEG_INF : log(pp);\n        out.r = !(pc > 0.0) ? FG_NEG_INF : log(pc);\n"
                 "        out.nb = (int)s1.c1;\n        " + cell + " = fg_as_double(prop);\n    }\n";
            fns += "static __device__ __noinline__ FgJitProp fg_jit_prop_" + std::to_string(site) + "(FG_LDSQ double *slots, FgStream rng, bool is_t) {\n    const double pert = 0.0; (void)pert;\n" +
                   g.decls() + g.body + c + "    (void)acc; (void)pr; (void)lk; (void)fc;\n    return out;\n}\n";
            prop_cases += "    case " + std::to_string(site) + ": { const bool is_t = mh.target == " + std::to_string((int)L.aux) + "; const FgJitProp r = fg_jit_prop_" + std::to_string(site) +
                          "(slots, mh.rng, is_t); if (is_t) { mh.lqf += r.f; mh.lqr += r.r; mh.next_block = r.nb; } return true; }\n";
        }
    if (!prop_cases.empty())
        fns = "struct FgJitProp { double f, r; int nb; };\n" + fns;
    if (!prop_cases.empty())
        fns += "static __device__ __forceinline__ bool fg_jit_prop(int site, FG_LDSQ double *slots, FgMhCtx &mh) {\n    switch (site) {\n" + prop_cases + "    default: return false;\n    }\n}\n"
               "#define FG_MHMW_JIT_PROP(site, slots, mh) fg_jit_prop(site, FG_JIT_LDS(slots), mh)\n";
    fns += "static __device__ __forceinline__ void fg_jit_mhb(int sg, const FG_LDSQ double *slots, FG_LDSQ double *terms) {\n    switch (sg) {\n";
    for (int sg = 0; sg < NSEG; ++sg) fns += "    case " + std::to_string(sg) + ": fg_jit_mhb_" + std::to_string(sg) + "(slots, terms); break;\n";
    fns += "    default: break;\n    }\n}\n";
    std::string src = PROLOGUE;
    if (fg_jit_inlined()) src += "#define FG_JIT_INLINED 1\n";
    src += FG_JIT_EMBED_API;
    src += FG_JIT_EMBED_HEAD;
    src += FG_JIT_EMBED_INTERP;                  // fg_interp.h (fg_gradstream.h builds on it)
    src += FG_JIT_EMBED_GRADSTREAM;              // fg_gradstream.h: fg_score_one for the control wave's undecided-kind probes, the in-order sums
    src += HELPERS;
    for (const std::string &t : tables) src += t + "\n";
    for (const auto &kv : lp_fns) src += kv.second;
    src += fns;
    bool all = true;
    for (char gch : generated) all = all && gch != 0;
    if (all) src += "#define FG_MHMW_ALL 1\n";
    if (no_stream) {
        // a program without a score stream: the statement count and the log_prior rows are the unit's own; a proposal that needs the
        // model comes from the target's OWN statement of the generic program, interpreted (fg_mh_mw_body.h: the kernel's `srt` argument
        // is then the [S][2] table of first instruction and count)
        if (!all) return "";
        src += "#define FG_MHMW_NS " + std::to_string(n_stmt) + "\n#define FG_MHMW_NPRI " + std::to_string(n_pri) + "\n#define FG_MHMW_NFAC " + std::to_string(n_fac) + "\n";
    }
    // the control wave's in-order sums with the row counts as literals: straight-line loads and additions (every row of a chain is
    // added in program order from 0.0 -- fg_inorder_sums2's additions -- without chunk loops, tails of selected zeros or address arithmetic)
    if (!no_stream && sum_pri >= 0 && sum_lik >= 0 && sum_pri + sum_lik <= 48) {     // (short programs: reference_model(50) and normal32 -- 64 and 99 rows -- measured 8-12 % slower with straight-line sums than with the chunked loops)
        std::string f = "static __device__ __forceinline__ void fg_jit_sums2(const FG_LDSQ double *terms, double &pri_out, double &lik_out) {\n    double a = 0.0, b = 0.0;\n";   // (inline: a call would drain the control wave's adaptation-state gather, which is in flight across the sums)
        const int form = std::getenv("FG_MH_SUMS_FORM") ? std::atoi(std::getenv("FG_MH_SUMS_FORM")) : sums_form;   // (0: plain statements; 3: pinned, no prefetch; n >= 4: pinned, rows n pairs ahead -- profiles/round4_mh_sums_form.txt)
        if (form >= 3) {         // (every row requested first instead -- 78 live VGPRs -- spilled 36 registers and lost 7 %: profiles/round4_mh_sums_form.txt)
            // the two chains pinned side by side (b is only used behind the branches that follow: the sink pass moves its whole chain there, and the scheduler
            // runs a to its end first -- 39 dependent additions where 20 pairs do); loads do not cross the pins, so the rows are requested `form` pairs ahead
            const int ahead = form == 3 ? 0 : form, n = std::max(sum_pri, sum_lik);
            auto ld = [&](int k) {
                std::string o;
                if (k < sum_pri) o += "    const double y" + std::to_string(k) + " = terms[" + std::to_string(k) + " * FG_WAVE];";
                if (k < sum_lik) o += "    const double y" + std::to_string(sum_pri + k) + " = terms[" + std::to_string(sum_pri + k) + " * FG_WAVE];";
                return o + "\n";
            };
            for (int k = 0; k < std::min(ahead, n); ++k) f += ld(k);
            for (int k = 0; k < n; ++k) {
                if (ahead == 0) f += ld(k);
                if (k < sum_pri) f += "    a += y" + std::to_string(k) + ";";
                if (k < sum_lik) f += "    b += y" + std::to_string(sum_pri + k) + ";";
                f += "\n";
                if (ahead > 0 && k + ahead < n) f += ld(k + ahead);
                f += "    asm volatile(\"\" : \"+v\"(a), \"+v\"(b));\n";
            }
        } else
        for (int k = 0; k < std::max(sum_pri, sum_lik); ++k) {
            if (k < sum_pri) f += "    a += terms[" + std::to_string(k) + " * FG_WAVE];";
            if (k < sum_lik) f += "    b += terms[" + std::to_string(sum_pri + k) + " * FG_WAVE];";
            f += "\n";
        }
        f += "    pri_out = a; lik_out = b;\n}\n";
        f += "static __device__ __forceinline__ double fg_jit_sum_pri(const FG_LDSQ double *terms) {\n    double a = 0.0;\n";
        for (int k = 0; k < sum_pri; ++k) f += "    a += terms[" + std::to_string(k) + " * FG_WAVE];\n";
        f += "    return a;\n}\nstatic __device__ __forceinline__ double fg_jit_sum_lik(const FG_LDSQ double *terms) {\n    double b = 0.0;\n";
        for (int k = 0; k < sum_lik; ++k) f += "    b += terms[" + std::to_string(sum_pri + k) + " * FG_WAVE];\n";
        f += "    return b;\n}\n";
        src += f;
        src += "#define FG_MHMW_SUMS2(PRI, LIK) fg_jit_sums2(FG_JIT_LDS(terms), PRI, LIK)\n#define FG_MHMW_SUM_PRI() fg_jit_sum_pri(FG_JIT_LDS(terms))\n#define FG_MHMW_SUM_LIK() fg_jit_sum_lik(FG_JIT_LDS(terms))\n";
    }
    // what every launch of this unit passes anyway (the engine checks it does): row counts, tile layout, waves per tile and the mode bits as literals --
    // the step loop loses their scalar tests and branches, LDS addresses become instruction offsets
    if (baked && !pipe) {
        if (!no_stream && !(std::getenv("FG_MH_BAKE") && std::atoi(std::getenv("FG_MH_BAKE")) == 2)) src += "#define FG_MHMW_K_NCU " + std::to_string(baked[0]) + "\n#define FG_MHMW_K_NS " + std::to_string(baked[1]) + "\n#define FG_MHMW_K_NPRI " + std::to_string(baked[2]) + "\n";
        src += "#define FG_MHMW_K_NSLOTS " + std::to_string(baked[3]) + "\n#define FG_MHMW_K_W " + std::to_string(baked[4]) + "\n#define FG_MHMW_K_EXP " + std::to_string(baked[5]) +
               "\n#define FG_MHMW_K_POOLN " + std::to_string(baked[6]) + "\n";
    }
    src += "#define FG_JIT_NSEG " + std::to_string(NSEG) + "\n"
           "#define FG_MHMW_PHASE_B5() do { for (int sg_ = wv; sg_ < FG_JIT_NSEG; sg_ += W) fg_jit_mhb(sg_, FG_JIT_LDS(slots), FG_JIT_LDS(terms)); } while (0)\n";
#ifdef FG_MH_PROF
    src += "#define FG_MH_PROF 1\n";              // (experiment builds: the compiled kernel carries the phase counters too)
#endif
    src += FG_JIT_EMBED_MHMW_BODY;               // fg_mh_mw_body.h
    const bool pipe2 = pipe && !no_stream;       // the step loop with the serial recipe split over waves (stream programs)
    if (pipe2) src += FG_JIT_EMBED_MHMW2_BODY;   // fg_mh_mw2_body.h
    // (128 VGPRs, four waves per SIMD: a 256-register budget for launches of <= 8 waves per tile -- 135 - 165 used -- lost 40 % at 65 536 chains)
    const std::string lb = "FG_WAVE * FG_MH_WMAX, 4";
    src += "extern \"C\" __global__ __launch_bounds__(" + lb + ") void k_mh_mw_jit_steps(FgProgramDev P, FgChainCtx X, FgMhDev M, const FgGradRec *srt, FgMhSeg seg, int iter0, int n_steps,\n"
           "        int n_warmup, long long *draws, int first_sample_t, int exp_mask, int pool_n) {\n"
           "    " + std::string(pipe2 ? "fg_mh_mw2_body<" : "fg_mh_mw_body<") + std::to_string(rk) + ", " + (split ? "true" : "false") + ">(P, X, M, srt, seg, iter0, n_steps, n_warmup, draws, first_sample_t, exp_mask, pool_n);\n}\n";
    if (std::getenv("FG_JIT_BREAK")) src += "\n#error FG_JIT_BREAK: a compilation that fails (tests of the fallback to the interpreter kernels)\n";
    if (ctab_out) *ctab_out = ctabs.data;
    return src;
}

// ---- the compiler -----------------------------------------------------------------------------------------------------------
// Two ways to the same code object.  In-process: hiprtc, bound with dlopen beside the HIP runtime in use.  Out-of-process: `hipcc --genco`
// of the system ROCm as a child process with a clean environment.  The second exists because a host process may run on a HIP runtime
// that is NOT a ROCm installation's -- PyTorch wheels bundle their own libamdhip64 / libhiprtc / libamd_comgr, and the code objects that
// bundle's hiprtc produced for these units did not run ("invalid kernel file" / HSA_STATUS_ERROR_INVALID_ISA on MI355X, ROCm 7.0 bundle
// inside a ROCm 7.2 image), while everything the system toolchain compiles -- this library included -- runs on either runtime.  So:
// hiprtc only when the runtime in use sits in a ROCm installation (a bin/hipcc beside its lib directory), else hipcc, else nothing
// (the interpreter kernels).  FG_JIT_COMPILER=hiprtc / hipcc forces one.
#ifndef FG_JIT_NO_HIP
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
extern char **environ;

static std::string runtime_dir() {                           // directory of the libamdhip64 this process runs on
    Dl_info di;
    if (dladdr((void *)hipGetDeviceCount, &di) && di.dli_fname) { std::string d(di.dli_fname); const size_t s = d.rfind('/'); if (s != std::string::npos) return d.substr(0, s); }
    return "";
}
static bool is_file(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
static std::string find_hipcc() {
    std::vector<std::string> c;
    if (const char *h = std::getenv("HIPCC")) c.push_back(h);
    if (const char *r = std::getenv("ROCM_PATH")) c.push_back(std::string(r) + "/bin/hipcc");
    c.push_back("/opt/rocm/bin/hipcc");
    for (const std::string &p : c) if (is_file(p) && ::access(p.c_str(), X_OK) == 0) return p;
    return "";
}
static int compile_with_hipcc(const std::string &hipcc, const std::string &src, std::vector<char> &code, std::string &log) {
    std::string dir = std::getenv("FG_JIT_CACHE") ? std::getenv("FG_JIT_CACHE") : "/tmp/fugue_amd_jit_" + std::to_string((long long)getuid());
    (void)mkdir(dir.c_str(), 0700);
    const std::string base = dir + "/build_" + std::to_string((long long)getpid()) + "_" + std::to_string((long long)std::hash<std::string>{}(src) & 0xffffff);
    const std::string in = base + ".hip", out = base + ".hsaco", err = base + ".log";
    { FILE *f = std::fopen(in.c_str(), "wb"); if (!f) { log = "cannot write " + in; return FG_E_HIP; }
      const bool okw = std::fwrite(src.data(), 1, src.size(), f) == src.size(); if (std::fclose(f) != 0 || !okw) { log = "cannot write " + in; return FG_E_HIP; } }
    std::vector<std::string> av = { hipcc, "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-w", "-x", "hip", in, "-o", out };
    std::vector<char *> argv; for (std::string &a : av) argv.push_back(&a[0]); argv.push_back(nullptr);
    // the child's environment: this process's, without what a profiler or a Python wheel put there for THIS process
    std::vector<std::string> ev; std::vector<char *> envp;
    for (char **e = environ; e && *e; ++e) {
        const std::string kv(*e);
        if (!kv.compare(0, 11, "LD_PRELOAD=") || !kv.compare(0, 16, "LD_LIBRARY_PATH=") || !kv.compare(0, 5, "ROCP_") || !kv.compare(0, 12, "ROCPROFILER_") || !kv.compare(0, 10, "HSA_TOOLS_")) continue;
        ev.push_back(kv);
    }
    for (std::string &e : ev) envp.push_back(&e[0]); envp.push_back(nullptr);
    posix_spawn_file_actions_t fa; posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, err.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    const int sp = posix_spawn(&pid, hipcc.c_str(), &fa, nullptr, argv.data(), envp.data());
    posix_spawn_file_actions_destroy(&fa);
    int status = -1;
    if (sp == 0) { while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {} }
    { std::FILE *f = std::fopen(err.c_str(), "rb"); if (f) { char b[4096]; size_t n; while ((n = std::fread(b, 1, sizeof b, f)) > 0) log.append(b, n); std::fclose(f); } }
    int rc = FG_E_HIP;
    if (sp == 0 && WIFEXITED(status) && WEXITSTATUS(status) == 0) {
        if (FILE *f = std::fopen(out.c_str(), "rb")) {
            std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
            if (n > 0) { code.resize((size_t)n); if (std::fread(code.data(), 1, (size_t)n, f) == (size_t)n) rc = FG_OK; }
            std::fclose(f);
        }
    } else if (sp != 0) log += " (posix_spawn of " + hipcc + " failed)";
    (void)std::remove(in.c_str()); (void)std::remove(out.c_str()); (void)std::remove(err.c_str());
    return rc;
}
#endif

// Compiles `src` for gfx950; on success `code` holds the code object.  `log` receives the compiler's messages.
int fg_jit_compile(const std::string &src, std::vector<char> &code, std::string &log) {
#ifndef FG_JIT_NO_HIP
    const char *force = std::getenv("FG_JIT_COMPILER");
    const std::string rt = runtime_dir(), hipcc = find_hipcc();
    const bool rocm_install = !rt.empty() && is_file(rt + "/../bin/hipcc") && is_file(rt + "/libhiprtc.so");
    const bool use_rtc = force ? !std::strcmp(force, "hiprtc") : rocm_install;
    if (!use_rtc) {
        if (hipcc.empty()) { log = "no compiler: the HIP runtime in use is not part of a ROCm installation and no hipcc was found"; return FG_E_UNSUPPORTED; }
        return compile_with_hipcc(hipcc, src, code, log);
    }
#endif
    Rtc &R = rtc();
    if (!R.ok) { log = "hiprtc not available"; return FG_E_UNSUPPORTED; }
    void *prog = nullptr;
    if (R.create(&prog, src.c_str(), "fg_model.hip", 0, nullptr, nullptr) != 0) { log = "hiprtcCreateProgram failed"; return FG_E_HIP; }
    const char *opts[] = { "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-variable", "-Wno-unused-but-set-variable" };
    const int rc = R.compile(prog, (int)(sizeof opts / sizeof opts[0]), opts);
    size_t n = 0;
    if (R.log_size(prog, &n) == 0 && n > 1) { log.resize(n); R.log(prog, &log[0]); }
    if (rc != 0) { R.destroy(&prog); return FG_E_HIP; }
    if (R.code_size(prog, &n) != 0 || n == 0) { R.destroy(&prog); log += " (no code)"; return FG_E_HIP; }
    code.resize(n);
    R.code(prog, code.data());
    R.destroy(&prog);
    return FG_OK;
}

// ---- compiled code objects: per process by source text, and on disk (FG_JIT_CACHE, default /tmp/fugue_amd_jit_<uid>) by its hash -----------
#include <sys/stat.h>
#include <unistd.h>

int fg_jit_get_code(const std::string &src, std::vector<char> &code, std::string &log) {
    static std::mutex mu;
    static std::map<std::string, std::vector<char>> mem;
    std::lock_guard<std::mutex> lk(mu);
    auto it = mem.find(src);
    if (it != mem.end()) { code = it->second; return FG_OK; }
    unsigned long long hsh = 1469598103934665603ULL;                     // FNV-1a of the source text
    for (unsigned char ch : src) { hsh ^= ch; hsh *= 1099511628211ULL; }
    std::string dir = std::getenv("FG_JIT_CACHE") ? std::getenv("FG_JIT_CACHE") : "/tmp/fugue_amd_jit_" + std::to_string((long long)getuid());
    char name[64]; std::snprintf(name, sizeof name, "/%016llx_%zu.hsaco", hsh, src.size());
    const std::string path = dir + name;
    // a cache file is {magic, source length, source text, code object}: the text is compared in full, so a hash collision or a truncated
    // file can never hand back another program's code
    static const char MAGIC[8] = { 'F', 'G', 'J', 'I', 'T', '0', '1', '\n' };
    if (FILE *f = std::fopen(path.c_str(), "rb")) {
        std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        std::vector<char> buf;
        if (n > (long)(sizeof MAGIC + 8)) { buf.resize((size_t)n); if (std::fread(buf.data(), 1, (size_t)n, f) != (size_t)n) buf.clear(); }
        std::fclose(f);
        unsigned long long sl = 0;
        if (!buf.empty() && !std::memcmp(buf.data(), MAGIC, sizeof MAGIC)) std::memcpy(&sl, buf.data() + sizeof MAGIC, 8);
        const size_t head = sizeof MAGIC + 8;
        if (!buf.empty() && sl == src.size() && buf.size() > head + sl && !std::memcmp(buf.data() + head, src.data(), sl)) {
            code.assign(buf.begin() + (long)(head + sl), buf.end());
            mem[src] = code;
            return FG_OK;
        }
    }
    const int rc = fg_jit_compile(src, code, log);
    if (rc != FG_OK) return rc;
    mem[src] = code;
    (void)mkdir(dir.c_str(), 0700);
    const std::string tmp = path + ".tmp." + std::to_string((long long)getpid());
    if (FILE *f = std::fopen(tmp.c_str(), "wb")) {
        const unsigned long long sl = src.size();
        const bool okw = std::fwrite(MAGIC, 1, sizeof MAGIC, f) == sizeof MAGIC && std::fwrite(&sl, 1, 8, f) == 8 && std::fwrite(src.data(), 1, src.size(), f) == src.size() &&
                         std::fwrite(code.data(), 1, code.size(), f) == code.size();
        std::fclose(f);
        if (okw) (void)std::rename(tmp.c_str(), path.c_str()); else (void)std::remove(tmp.c_str());
    }
    return FG_OK;
}

// test hook (no GPU needed: hiprtc cross-compiles): generated source and compiler log of a program's HMC kernel
extern "C" int fg_debug_jit_compile(const fg_program *p, char *src_out, long long src_cap, char *log_out, long long log_cap, long long *code_bytes) {
    if (!p) return FG_E_BAD_ARG;
    const bool mh = std::getenv("FG_DEBUG_JIT_MH") != nullptr;            // the MH unit instead of the HMC one
    if (std::getenv("FG_DEBUG_JIT_MHMW")) {                                // the multi-wave stream MH unit
        std::string s2;
        // FG_DEBUG_JIT_BAKE="exp_mask": the launch shape as literals (FG_DEBUG_JIT_NSEG waves per tile, no row-less terms, no staged pool)
        const int dbg_baked[7] = { 0, p->n_sstream, p->n_prior_terms, p->n_slots, std::getenv("FG_DEBUG_JIT_NSEG") ? std::atoi(std::getenv("FG_DEBUG_JIT_NSEG")) : 0, std::getenv("FG_DEBUG_JIT_BAKE") ? std::atoi(std::getenv("FG_DEBUG_JIT_BAKE")) : 0, 0 };
        if (p->n_sstream > 0) s2 = fg_jit_mhmw_source(p, std::vector<long long>((size_t)p->n_ins, 1), std::vector<char>((size_t)p->n_sstream, 1), p->sstream_has_gen ? (p->sstream_has_genrec ? 2 : 3) : 0, p->n_sstream >= 64, nullptr, nullptr, -1, 0, false, std::getenv("FG_MH_PIPE") && std::atoi(std::getenv("FG_MH_PIPE")) == 1, std::getenv("FG_DEBUG_JIT_NSEG") ? std::atoi(std::getenv("FG_DEBUG_JIT_NSEG")) : 0, 16, p->n_prior_terms, p->n_sstream - p->n_prior_terms, std::getenv("FG_DEBUG_JIT_BAKE") ? dbg_baked : nullptr, std::getenv("FG_MH_SUMS_FORM") ? std::atoi(std::getenv("FG_MH_SUMS_FORM")) : 0);
        else {                                                             // a program without a score stream: rows in accumulator order (fg_mh_mw_nostream_launch)
            std::vector<int> rows; int n_pri = 0, n_lik = 0, n_fac = 0;
            for (int k = 0; k < p->n_ins; ++k) if (Gen::ends_statement(p->ins_fast[(size_t)k])) {
                const uint32_t code = FG_INS_OPCODE(p->ins_fast[(size_t)k].op);
                const int a = code == FG_OP_FACTOR ? 2 : ((p->ins_fast[(size_t)k].op & FG_F_OBSERVE) != 0 || code == FG_OP_CONSTLIK) ? 1 : 0;
                rows.push_back(a); (a == 0 ? n_pri : a == 1 ? n_lik : n_fac) += 1;
            }
            for (int k = 0, a = 0, b = n_pri, c = n_pri + n_lik; k < (int)rows.size(); ++k) rows[(size_t)k] = rows[(size_t)k] == 0 ? a++ : rows[(size_t)k] == 1 ? b++ : c++;
            s2 = fg_jit_mhmw_source(p, std::vector<long long>((size_t)p->n_ins, 1), std::vector<char>(rows.size(), 1), 3, rows.size() >= 64, nullptr, &rows, n_pri, n_fac, true);
        }
        if (src_out && src_cap > 0) std::snprintf(src_out, (size_t)src_cap, "%s", s2.c_str());
        if (code_bytes) *code_bytes = 0;
        if (s2.empty()) return FG_E_UNSUPPORTED;
        std::vector<char> code2; std::string log2;
        const int rc2 = fg_jit_compile(s2, code2, log2);
        if (log_out && log_cap > 0) std::snprintf(log_out, (size_t)log_cap, "%s", log2.c_str());
        if (code_bytes) *code_bytes = (long long)code2.size();
        if (const char *out = std::getenv("FG_DEBUG_JIT_OUT")) if (rc2 == FG_OK) if (FILE *f = std::fopen(out, "wb")) { std::fwrite(code2.data(), 1, code2.size(), f); std::fclose(f); }
        return rc2;
    }
    std::vector<std::vector<int>> dbg_bins;                               // FG_DEBUG_JIT_TASKS=W: the 2 d tasks dealt round-robin over W waves as straight-line code (fg_jit_wave_tasks)
    if (const char *tw_ = std::getenv("FG_DEBUG_JIT_TASKS")) { const int W_ = std::max(1, std::min(16, std::atoi(tw_))); dbg_bins.resize((size_t)W_); for (int t = 0; t < 2 * (int)p->coord.size(); ++t) dbg_bins[(size_t)(t % W_)].push_back(t); }
    std::vector<std::vector<int>> dbg_cbins;                              // FG_DEBUG_JIT_COORDS=W: whole coordinates dealt round-robin over W waves (fg_jit_wave_grad)
    if (const char *cw_ = std::getenv("FG_DEBUG_JIT_COORDS")) { const int W_ = std::max(1, std::min(16, std::atoi(cw_))); dbg_cbins.resize((size_t)W_); for (int k = 0; k < (int)p->coord.size(); ++k) dbg_cbins[(size_t)(k % W_)].push_back(k); }
    const std::string src = mh ? fg_jit_mh_source(p, std::vector<long long>((size_t)p->n_ins, 1), 4, nullptr) : fg_jit_hmc_source(p, nullptr, nullptr, nullptr, dbg_bins.empty() ? nullptr : &dbg_bins, dbg_cbins.empty() ? nullptr : &dbg_cbins, dbg_cbins.empty() ? nullptr : &dbg_cbins);
    if (src_out && src_cap > 0) { std::snprintf(src_out, (size_t)src_cap, "%s", src.c_str()); }
    if (code_bytes) *code_bytes = 0;
    if (src.empty()) return FG_E_UNSUPPORTED;
    std::vector<char> code; std::string log;
    const int rc = fg_jit_compile(src, code, log);
    if (log_out && log_cap > 0) std::snprintf(log_out, (size_t)log_cap, "%s", log.c_str());
    if (code_bytes) *code_bytes = (long long)code.size();
    if (const char *out = std::getenv("FG_DEBUG_JIT_OUT")) if (rc == FG_OK) if (FILE *f = std::fopen(out, "wb")) { std::fwrite(code.data(), 1, code.size(), f); std::fclose(f); }   // for llvm-objdump / readelf
    return rc;
}

#ifndef FG_JIT_NO_HIP
// Uploads a module's constant tables and writes their device address into its global `fg_jit_ctab_ptr`.  *d_tab receives the
// allocation (the caller frees it with the engine).
int fg_jit_bind_tables(hipModule_t mod, const std::vector<double> &tab, double **d_tab, hipStream_t stream) {
    *d_tab = nullptr;
    if (tab.empty()) return FG_OK;
    if (hipMalloc((void **)d_tab, tab.size() * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return FG_E_HIP; }
    if (hipMemcpyAsync(*d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return FG_E_HIP;
    hipDeviceptr_t gp = nullptr; size_t gb = 0;
    if (hipModuleGetGlobal(&gp, &gb, mod, "fg_jit_ctab_ptr") != hipSuccess || gb != sizeof(double *)) { (void)hipGetLastError(); return FG_E_HIP; }
    if (hipMemcpyHtoD(gp, d_tab, sizeof(double *)) != hipSuccess) return FG_E_HIP;
    return FG_OK;
}
#endif
