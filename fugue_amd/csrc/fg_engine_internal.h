// fg_engine_internal.h -- definitions shared by the engine translation units
// (fg_engine.hip: prior/score/HMC/MH; fg_smc.hip: SMC + reductions; fg_diag.hip: diagnostics).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fg_interp.h"
#include "fg_program.h"
#include "fg_dev_types.h"

#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    fg_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return FG_E_HIP; } } while (0)

// ======================================================================================
// kernels
// ======================================================================================
static __global__ void k_fill(double *p, long long n, double v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct fg_engine {
    const fg_program *prog = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    long long C = 0;
    unsigned long long seed = 0;
    uint32_t chain0 = 0;
    int S = 0, d = 0, n_slots = 0;
    // device copies of the program
    FgIns *d_ins = nullptr, *d_ins_fast = nullptr, *d_sub = nullptr;
    FgCoord *d_coord = nullptr;
    FgGradRec *d_gstream = nullptr, *d_sstream = nullptr;
    FgSepRec *d_sep = nullptr; FgSepCoord *d_sep_coord = nullptr; FgSepFree *d_sep_free = nullptr; uint32_t *d_sobs = nullptr; int *d_site_rec = nullptr;
    FgGradRec *d_mh_srt = nullptr; int mh_cls_off[7] = {0, 0, 0, 0, 0, 0, 0};   // kind-sorted score stream of the multi-wave MH kernel (fg_mh.hip), class boundaries
    bool mh_mw_disabled = false;  // FG_MH_MW=0: keep every program on the one-wave-per-tile MH kernel (A/B tests)
    bool mh_has_prior_resample = false;   // an override asks for PriorResample on some site (needs the model-driven proposal path)
    bool sep_disabled = false;   // FG_HMC_SEP=0: keep independent-sites programs on the gradient-stream kernel (A/B tests)
    bool gt = false;             // the program's tile exceeds a CU's LDS: tiles in global memory, one-wave-per-tile kernels only
    double *d_gtile = nullptr;
    bool lin_disabled = false;   // FG_HMC_LIN=0: keep dense regressions on the gradient-stream kernel (A/B tests)
    double *d_lin_tab = nullptr; int *d_lin_meta = nullptr;   // dense-regression table (fg_hmc_lin.hip)
    int *d_sub_off = nullptr, *d_f64_slot = nullptr, *d_site_slot = nullptr, *d_vtype = nullptr, *d_site_cat = nullptr;
    double *d_pool = nullptr;
    FgProgramDev P{};
    FgChainCtx X{};
    long long *d_values = nullptr;
    double *d_acc = nullptr, *d_logp = nullptr;
    // HMC
    bool hmc_ready = false;
    fg_hmc_config cfg{};
    FgHmcDev H{};
    std::vector<void *> hmc_allocs;
    int n_warmup = 0, iter = 0, mass_adapt_at = -1;
    // MH
    bool mh_ready = false;
    FgMhDev M{};
    std::vector<void *> mh_allocs;
    int mh_warmup = 0, mh_iter = 0;
    std::vector<fg_site_proposal> mh_overrides;   // as given to fg_mh_init (site order), for fg_state_export
    int *d_rec = nullptr; int rec_cap = 0;
    bool smc_pop_ready = false;   // the arena holds a particle population (log-weights, weights, log-likelihoods) for the standalone SMC calls
    void *smc_arena = nullptr; size_t smc_arena_bytes = 0;   // scratch of fg_smc_run, allocated once per engine (fg_smc.hip)
    int smc_epoch = 0;            // tags the pass flags of one next_beta search in smc_host
    void *smc_host = nullptr;     // pinned host scalars the SMC kernels write (beta, log-evidence, flags): the host's look at them needs no copy
    double *d_tmp = nullptr;     // [C] scratch
    int *d_itmp = nullptr;       // [3][C] scratch
    size_t lds_bytes = 0;        // LDS tile of the HMC kernels (slots + momentum + exchange rows)
    size_t lds_score = 0;        // LDS tile of the score / prior / MH / SMC kernels (slots only)
    int tw = 64;               // tile width (threads per block)
    int n_simd = 1024;         // SIMDs on the device (4 per CU)
    int diag_mode = 0;         // FG_DIAG_REDUCE / FG_DIAG_GATHER: how fg_diag_rhat_ess exchanges chain statistics between ranks
    long long diag_bytes = 0;  // bytes this rank put into collectives during the last fg_diag_rhat_ess
    std::string last_hmc_kernel;   // kernel (and waves per tile) the last fg_hmc_step launch ran (fg_hmc_last_kernel)
    bool interp_mw_disabled = false;   // FG_HMC_INTERP_MW=0: keep interpreter programs on the one-wave-per-tile HMC kernel (A/B tests)
    std::vector<std::vector<int>> jit_baked_bins, jit_baked_cbins, jit_baked_cbins_dense; bool mwi_baked = false, mwi_fused = false;   // the task split the compiled HMC unit was generated behind; does the current split equal it
    int *d_mwi_order = nullptr; long long *d_mwi_prof = nullptr; std::vector<int> mwi_off, mwi_off_an; std::vector<long long> mwi_cost; int mwi_W = 0, mwi_sparse = -1, mwi_calibrated = 0;   // coordinate split of k_hmc_interp_mw_steps (fg_hmc_interp.hip)
    int mhi_W = 0, mhi_n_stmt = 0, mhi_occ = 2; bool mhi_setup_done = false; size_t mhi_lds = 0; std::vector<int> mhi_ins_off, mhi_stmt_off; unsigned char *d_mhi_acc = nullptr; int *d_mhi_site_ins = nullptr; std::vector<int> mhi_stmt_end; std::vector<unsigned char> mhi_acc_host;   // statement split of k_mh_interp_mw_steps (fg_mh_interp.hip)
    int jit_state = 0;           // run-time compiled HMC kernel of this program: 0 not tried, 1 loaded, -1 unavailable (fg_jit.cpp; FG_JIT=0 switches it off)
    hipModule_t jit_mod = nullptr; hipFunction_t jit_fn = nullptr, jit_fn_eps = nullptr, jit_fn_rejuv = nullptr, jit_fn_prior = nullptr, jit_fn_lj = nullptr; std::string jit_log; bool jit_lds_attr = false, jit_rejuv_attr = false, jit_has_ad = false, jit_has_dense = false, an_jit = false /* FG_GRAD_ANALYTIC runs on the compiled unit's derivative code */; double *d_jit_tab = nullptr, *d_jit_mh_tab = nullptr;   // the modules' constant tables (fg_jit_bind_tables)
    int mh_ncu = -1, mh_catu_same = 0; double mh_catu_c0 = 0.0; double *d_mh_catu_c = nullptr; void *d_mh_catu = nullptr;   // row-less uniform Categorical terms of the multi-wave MH kernel (FgMhSeg; -1: not decided)
    int jit_mhns_state = 0, jit_mhns_split = 0; hipModule_t jit_mhns_mod = nullptr; hipFunction_t jit_mhns_fn = nullptr; double *d_jit_mhns_tab = nullptr;   // ... the same kernel for a program without a score stream
    int jit_mhmw_baked[7] = {0, 0, 0, 0, 0, 0, 0}; bool jit_mhmw_has_baked = false;   // the launch shape the unit below was generated for (fg_jit_mhmw_source)
    int jit_mhmw_state = 0; hipModule_t jit_mhmw_mod = nullptr; hipFunction_t jit_mhmw_fn = nullptr; double *d_jit_mhmw_tab = nullptr;   // ... the multi-wave stream MH kernel with phase B generated (fg_mh.hip)
    int jit_mh_state = 0, jit_mh_W = 1, jit_mh_direct = 0; size_t jit_mh_lds = 0; hipModule_t jit_mh_mod = nullptr; hipFunction_t jit_mh_fn[2] = {nullptr, nullptr};   // ... and its MH kernel (128- and 256-VGPR builds)
    std::string last_mh_kernel;  // kernel the last fg_mh_step launch ran (fg_mh_last_kernel)
    int mw_override = 0;       // FG_HMC_WAVES env: force waves per tile of the multi-wave HMC kernel (tests)
};

// p0 ~ N(0, M): hmc.rs:436-441.  Box-Muller pairs from the chain's (iteration) stream.
__device__ __forceinline__ void fg_draw_momentum(const FgProgramDev &P, FgStream &rng, double *pl, int tw, const double *mass_sqrt,
                                                 long long C) {
    for (int i = 0; i < P.d; i += 2) {
        double z0, z1;
        fg_rng_normal_pair(rng, z0, z1);
        pl[i * tw] = z0 * (mass_sqrt ? mass_sqrt[(long long)i * C] : 1.0);
        if (i + 1 < P.d) pl[(i + 1) * tw] = z1 * (mass_sqrt ? mass_sqrt[(long long)(i + 1) * C] : 1.0);
    }
}


namespace {

// Tile width = lanes per wave that own a chain = 64.  Spreading 65 536 chains over narrower, half-empty waves was
// measured NOT to help (profiles/round1_occupancy_sweep.txt): an instruction costs the same whatever the number of
// active lanes, and the LDS tile, not the wave count, limits residency.  More waves per SIMD come from SHARING a tile
// between waves instead (k_hmc_stream_steps); the tile stride stays a compile-time constant.
int tile_width_for(long long C) { (void)C; return FG_WAVE; }

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return FG_OK;
}

template <typename T>
int dev_alloc(T **p, size_t n) {
    HIPCHK(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
    HIPCHK(hipMemset(*p, 0, (n ? n : 1) * sizeof(T)));
    return FG_OK;
}
template <typename T>
int dev_upload(T **p, const std::vector<T> &v) {
    int rc = dev_alloc(p, v.size());
    if (rc) return rc;
    if (!v.empty()) HIPCHK(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return FG_OK;
}

}  // namespace


#define NEED_ENGINE(e) do { if (!(e)) { fg_set_error("null engine"); return FG_E_BAD_ARG; } \
    if (hipSetDevice((e)->device) != hipSuccess) { fg_set_error("hipSetDevice failed"); return FG_E_HIP; } } while (0)

// fg_hmc_interp.hip: adaptive_smc's rejuvenation move through the model compiled at run time (FG_E_UNSUPPORTED: the interpreter kernel)
struct FgSmcDev; struct FgSmcScalars;
int fg_smc_jit_rejuv_launch(fg_engine *e, const FgSmcDev &M, const FgSmcScalars *st, uint32_t move_id, unsigned *n_blk_out, const long long *vsrc = nullptr, double *pmax = nullptr);
int fg_jit_prior_launch(fg_engine *e, uint32_t iteration, uint32_t purpose, double *d_acc, double *d_lj, bool compile);   // k_prior_jit, or FG_E_UNSUPPORTED
int fg_jit_log_joint_launch(fg_engine *e, double *d_acc, double *d_lj, bool compile);                                     // k_log_joint_jit, or FG_E_UNSUPPORTED
bool fg_hmc_jit_has_ad(fg_engine *e);      // the compiled module holds the forward-mode derivative of every sub-program (FG_GRAD_ANALYTIC for any program)

// fg_hmc_sep.hip: register-resident trajectories for independent-sites programs (FG_E_UNSUPPORTED: not applicable)
int fg_hmc_sep_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info);

// fg_hmc_lin.hip: observation-major finite-difference gradient for dense regressions (FG_E_UNSUPPORTED: not applicable)
int fg_hmc_lin_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info);

// fg_hmc_interp.hip: the program compiled at run time (fg_jit.cpp) behind the same multi-wave kernel (FG_E_UNSUPPORTED: not applicable)
int fg_hmc_jit_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info);
int fg_hmc_jit_find_eps(fg_engine *e, uint32_t instance, int injected, double *d_eps_out);
// fg_hmc_interp.hip: interpreter programs with a tile shared by W waves (FG_E_UNSUPPORTED: not applicable)
int fg_hmc_interp_launch(fg_engine *e, int iter0, int n, int welford_on, double *draws, int first_sample_t, double *pos_all, double *info);

// fg_engine.hip internals used by fg_state.hip
extern "C" {
int fg_internal_hmc_alloc(fg_engine *e, bool mass);
void fg_internal_hmc_set_cfg(fg_engine *e, const fg_hmc_config *cfg);
int fg_internal_hmc_step(fg_engine *e, int n_transitions, double *d_draws, double *d_pos_all, double *d_info);
int fg_internal_mh_alloc(fg_engine *e);
int fg_internal_mh_set_overrides(fg_engine *e, const fg_site_proposal *overrides);
}

// fg_mh_interp.hip: multi-wave single-site MH for interpreter programs (FG_E_UNSUPPORTED: not applicable)
long long fg_mhi_ins_cost(const FgIns &in);   // relative cost of an instruction (statement splits)
int fg_mh_interp_launch(fg_engine *e, int iter0, int n_steps, long long *draws, int first_sample_t);

// fg_mh.hip: multi-wave single-site MH for programs with a score stream (FG_E_UNSUPPORTED: not applicable)
int fg_mh_mw_launch(fg_engine *e, int iter0, int n_steps, long long *draws, int first_sample_t);
int fg_mh_mw_nostream_launch(fg_engine *e, int iter0, int n_steps, long long *draws, int first_sample_t, const std::vector<int> &stmt_end,
                             const std::vector<unsigned char> &acc, const int *d_site_ins);

extern "C" int fg_launch_prior(fg_engine *e, uint32_t iteration, uint32_t purpose, double *d_acc, double *d_lj);
