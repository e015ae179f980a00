// fg_ir.h -- device IR of a flattened site program (shared by the host compiler and the
// gfx950 kernels).
//
// One instruction is 96 bytes = 24 dwords, fetched by a wave with scalar loads (the
// program counter is wave-uniform: every lane = one chain runs the SAME program on its
// own column of the [slots x 64] LDS tile).
#pragma once
#include <stdint.h>

// operand word: kind in bits 31:30, index in bits 29:0
enum : uint32_t { FG_OPND_IMM = 0u, FG_OPND_SLOT_F = 1u, FG_OPND_SLOT_I = 2u, FG_OPND_POOL = 3u };
#define FG_OPND(kind, idx) ((uint32_t)(((uint32_t)(kind) << 30) | ((uint32_t)(idx) & 0x3fffffffu)))
#define FG_OPND_KIND(o) ((o) >> 30)
#define FG_OPND_IDX(o) ((o) & 0x3fffffffu)

// opcodes.  0..16 = distribution kinds (include/fugue_amd.h order)
enum : uint32_t {
    FG_OP_NORMAL_FAST = 20,  // Normal with constant valid sigma, x and mu each `imm + slot` (score-only programs):
                         //   opnd[0], opnd[1] = plain slot indices of x, mu (the always-zero slot for constants),
                         //   imm[0], imm[1] = their constants (0 for slots), imm[2] = sigma, h[0] = ln sigma, h[4] = 1/sigma
    FG_OP_FACTOR = 32,   // log_factors += x
    FG_OP_LOAD = 40,     // acc = x
    FG_OP_ADD, FG_OP_SUB, FG_OP_MUL, FG_OP_DIV,   // acc = acc (op) x
    FG_OP_RSUB, FG_OP_RDIV,                       // acc = x (op) acc
    FG_OP_NEG, FG_OP_EXP, FG_OP_LN, FG_OP_SQRT, FG_OP_ABS, FG_OP_FLOOR, FG_OP_SIN, FG_OP_COS, FG_OP_TANH,
    FG_OP_POW, FG_OP_RPOW,                        // acc = pow(acc, x) / pow(x, acc)
    FG_OP_MIN, FG_OP_MAX,
    FG_OP_CLAMP,         // acc = clamp(acc, x, p0)
    FG_OP_MAC,           // acc = acc + x * p0   (two roundings, as the expression tree)
    FG_OP_STORE,         // slot[aux] = acc
    FG_OP_GATHER,        // acc = slot[aux + (int)acc] for 0 <= acc < n (n = opnd[1] raw), else NaN
    FG_OP_CONSTLIK,      // log_likelihood += imm[0]  (observe with constant params and value)
    FG_OP_DOT            // acc = (..((acc + s_0 c_0) + s_1 c_1)..) + s_{n-1} c_{n-1}: a run of n = opnd[1] MACs of an f64 slot and a
                         //   constant (a linear predictor), terms {u32 slot, u32 0, f64 c} at pool[aux ..]; same two roundings per term
};
// flags in op bits 8..
enum : uint32_t {
    FG_F_OBSERVE = 1u << 8,        // else SAMPLE
    FG_F_HOISTED = 1u << 9,        // all parameters constant: guards checked, h[] valid
    FG_F_INVALID = 1u << 10,       // constant parameters are invalid: log-density is -inf
    FG_F_POW2SCALE = 1u << 11,     // hoisted scale is 2^k: h[4] = 1/scale, (x-loc)/scale == (x-loc)*h[4] exactly
    FG_F_VTYPE_SHIFT = 12,         // 3 bits: FG_F64..FG_I64
    FG_F_RCPSCALE = 1u << 16,      // h[4] = RN(1/scale) and fg_div_const_ok(scale): (x - loc) / scale via fg_div_const (same bits)
    FG_F_SCALEHOIST = 1u << 15,    // location-scale family with a constant valid scale but a varying location:
                                   // h[0] = the scale-only term (ln sigma, ...), h[4] = 1/scale when FG_F_POW2SCALE
    FG_F_XHOIST = 1u << 17         // observed COUNT is a constant while the parameters vary: h[3] = the term of the value alone --
                                   // Poisson ln k!, Binomial (constant n) ln C(n, k) -- one lgamma chain per evaluation less (count regressions)
};
#define FG_INS_OPCODE(op) ((op) & 0xffu)
#define FG_INS_VTYPE(op) (((op) >> FG_F_VTYPE_SHIFT) & 7u)

struct FgIns {
    uint32_t op;        // opcode | flags
    uint32_t opnd[4];   // x, p0, p1, p2   (Categorical: p0 = probability base, p1 = K raw)
    uint32_t aux;       // sample: site slot; STORE/GATHER: slot
    double   imm[4];    // immediates of x, p0, p1, p2 (IMM kind)
    double   h[5];      // hoisted constants (FG_F_HOISTED), per distribution (fg_math.h)
};
static_assert(sizeof(FgIns) == 96, "FgIns must be 96 bytes");

// A compiled program as the kernels see it.
// per f64 coordinate: what the finite-difference force needs, in one 16-byte scalar load
struct FgCoord { int slot, sub_off, sub_n, flags; };

// One record of the fused finite-difference gradient stream (programs whose sub-programs are all
// FG_OP_NORMAL_FAST): the instruction plus which of its operands is the perturbed coordinate.
enum : uint32_t { FG_S_OBS = 1u,      // score stream: the record is an observe statement (adds to log_likelihood)
                  FG_G_SWITCH = 1u,   // first observe record of a coordinate: stash the prior sums, restart the running sums
                  FG_G_POW2 = 2u, FG_G_PERT_X = 4u, FG_G_PERT_M = 8u, FG_G_END = 16u,
                  FG_G_X_CONST = 32u,  // x is a constant (ximm), no slot read
                  FG_G_M_CONST = 64u,  // mu is a constant (mimm), no slot read
                  FG_G_DIV = 128u,     // sigma outside the range where fg_div_const is proven exact: IEEE division
                  FG_G_NSEL = 1u << 27,   // Normal(x; mu = options[z], constant sigma): mi = slot of the index site z, mimm dwords = {pool offset of the
                                          //   K option entries {u32 slot, u32 is_const, f64 constant}, K}
                  FG_G_CATC = 1u << 28,   // Categorical site with a constant table (score stream only): xi = its slot, mimm dwords = {pool base, K}
                  FG_G_GEN = 1024u,    // any of the 17 distributions with leaf operands (fg_logpdf): layout below, kind in flags >> 16
                  FG_G_GEN_HOISTED = 2048u, FG_G_GEN_SH = 4096u, FG_G_GEN_INVALID = 8192u, FG_G_GEN_XINT = 16384u,
                  FG_G_GEN_XH = 32768u,   // FG_F_XHOIST of the statement: the value-only term sits in h1 (a record with varying parameters has no other use for it)
                  FG_G_GEN_P0SLOT = 1u << 24, FG_G_GEN_P1SLOT = 1u << 25, FG_G_GEN_P2SLOT = 1u << 26,
                  FG_G_LIN = 256u,     // mu = mimm + sum_t slot[s_t] c_t: maskx = pool offset of the terms {u32 s, u32 0, f64 c},
                                       //   maskm = their number, flags >> 16 = first term that reads the record's coordinate
                  FG_G_LIN1 = 512u     // ... and no other term reads it
};
struct FgGradRec {
    uint32_t xi, mi;        // slot indices of x and mu (the zero slot for constants)
    uint32_t flags, coord;  // FG_G_*; f64 coordinate this record belongs to
    double ximm, mimm;      // constants of x, mu (0 for slots): operand value = slot + imm, like FG_OP_NORMAL_FAST
    double sigma, inv;      // sigma and RN(1/sigma) (exact when FG_G_POW2; seed of the exact-division sequence otherwise)
    double lns;             // ln sigma
    uint32_t maskx, maskm;  // FG_G_LIN: pool offset / number of the linear predictor's terms (else unused)
};
// FG_G_GEN records reuse the 48 payload bytes as  x (imm double / int64 bits), p0, p1, p2 (imm double, or the LDS slot index in
// the low dword when FG_G_GEN_PkSLOT), h0, h1 (hoisted constants of fg_hoist): see FgGenRec.
struct FgGenRec { uint32_t xi, mi, flags, coord; double x; double p[3]; double h[2]; };
static_assert(sizeof(FgGenRec) == 64, "FgGenRec must be 64 bytes");
static_assert(sizeof(FgGradRec) == 64, "FgGradRec must be 64 bytes");

// Register-resident trajectories (fg_hmc_sep.hip): when every record of the gradient stream reads only its OWN coordinate
// and constants (an independent-sites model: no coordinate ever sees another one inside a trajectory), the whole L-step
// trajectory of a coordinate runs with q_i, p_i in registers and its <= FG_SEP_MAXREC records in SGPRs.  One compact
// record per (coordinate, dependent statement), in gradient-stream order: record 0 is the coordinate's own sample
// statement (the only prior term such a coordinate can have), the others are observe statements.
//   c = the statement's constant operand (x - mu is +-(q - c): the sign changes no bit of the log-density);
//   flags: FG_G_POW2 / FG_G_DIV as in the gradient stream;  trow = LDS row of the statement's endpoint-score term
//   (prior terms first, then likelihood terms, each in program order).
#define FG_SEP_MAXREC 4
struct FgSepRec { uint32_t flags, trow; double c, inv, lns, sigma; double pad1[3]; };   // dwords 0..7 = one s_load_dwordx8, sigma = dwords 8..9
static_assert(sizeof(FgSepRec) == 64, "FgSepRec must be 64 bytes");
struct FgSepCoord { int off, n; };     // records of coordinate k: sep[off .. off + (n & 7)); bit 8 of n: every sigma is a power of two; bit 9: and record 0 is Normal(0, 1)
struct FgSepFree { uint32_t sidx, trow; };   // score-stream statements that read no coordinate: evaluated once per launch

// Dense regressions (fg_hmc_lin.hip).  One table row per linear-predictor observe statement, read with scalar loads, all
// doubles:  [c0, 0 | c_0 .. c_{D/2-1} | c_{D/2} .. c_{D-1} | y, 1/sigma, ln sigma, sigma | flags (FG_G_POW2 / FG_G_DIV as an integer), 0]
// = D + 8 doubles, the coefficients in TERM order (term t reads coordinate lin_meta[t]); one zero row follows the last one (look-ahead).
#define FG_LIN_ROW_DOUBLES(D) ((D) + 8)

struct FgProgramDev {
    const FgIns  *ins;       // full program (generic opcodes only: PRIOR / MH / SCORE), n_ins
    const FgIns  *ins_fast;  // the same program with score-only fast opcodes substituted (HMC / SMC / log-joint)
    const FgCoord *coord;    // [d]
    const FgIns  *sub;       // concatenated per-coordinate sub-programs (sparse FD)
    const int    *sub_off;   // [d+1] offsets into sub
    const double *pool;      // data arrays + constant tables
    const int    *f64_site;  // [d] sorted site index of each f64 coordinate (its LDS slot is the coordinate index itself)
    const int    *site_slot; // [S] LDS slot of each site: f64 sites first (coordinate order), then the discrete sites
    const int    *site_vtype;// [S]
    const int    *site_cat;  // [S][2] {pool base, K} of Categorical sites with a valid constant probability table, else -1
    const FgGradRec *gstream;  // fused gradient stream or null
    const FgGradRec *sstream;  // score stream (the whole program as records, program order) or null: every statement is a fast Normal
    const FgSepRec *sep;       // compact per-coordinate records or null (see FgSepRec)
    const FgSepCoord *sep_coord;   // [d]
    const int *site_rec;           // [S] score-stream record of each site's own sample statement (or null without a score stream)
    const FgSepFree *sep_free;     // [n_sep_free]
    int n_sep_free, n_prior_terms; // term rows [0, n_prior_terms) are log_prior terms, [n_prior_terms, n_sstream) log_likelihood terms
    const uint32_t *sobs;      // bit k = record k of the score stream is an observe statement ((n_sstream + 31) / 32 words)
    int n_gstream, n_sstream;
    int sstream_gen;           // the score stream holds general distribution records (FG_G_GEN)
    int sstream_kinds;         // record kinds in the score stream: 0 fast Normals, 1 + linear predictors, 2 + general records
    int n_ins, n_slots, S, d;
    const double *lin_tab;     // dense-regression table (above) or null
    const int *lin_meta;       // [d] coordinate at term position t, then [d][2] {first prior record of coordinate k in gstream, count}
    int lin_n, lin_p2;         // observe statements in the table; every sigma is a power of two
};

// RNG stream purposes (counter word 3); shared spec with the test oracle
enum : uint32_t {
    FG_RNG_PRIOR = 1, FG_RNG_HMC = 2, FG_RNG_EPS = 3, FG_RNG_MH = 4,
    FG_RNG_SMC_RESAMPLE = 5, FG_RNG_SMC_REJUV = 6, FG_RNG_SMC_PRIOR = 7
};
