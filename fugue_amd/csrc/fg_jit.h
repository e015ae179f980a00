// fg_jit.h -- run-time compiled model kernels (fg_jit.cpp)
#pragma once
#include <string>
#include <vector>

struct fg_program;
// ctab_out (optional) receives the constants of the rolled runs: the engine uploads them and writes their device address into the
// loaded module's global `fg_jit_ctab_ptr` (fg_jit_bind_tables)
std::string fg_jit_hmc_source(const fg_program *p, std::vector<double> *ctab_out, bool *has_ad_out = nullptr, bool *has_dense_out = nullptr,
                              const std::vector<std::vector<int>> *wave_tasks = nullptr /* the sparse finite difference's tasks (2 k + sign) of every wave of the engine's launches: straight-line code per wave */,
                              const std::vector<std::vector<int>> *wave_coords = nullptr /* ... or whole coordinates per wave: both evaluations, the kick and the drift of a coordinate in one piece (fg_jit_wave_grad) */,
                              const std::vector<std::vector<int>> *wave_coords_dense = nullptr /* the same for the dense mode (whole-program evaluations: fg_jit_wave_grad_dense) */);       // "" = not covered by the generator; has_ad: the unit also holds the forward-mode derivative of every sub-program (FG_GRAD_ANALYTIC)
std::string fg_jit_mh_source(const fg_program *p, const std::vector<long long> &ins_cost, int occ, std::vector<double> *ctab_out);   // ins_cost[k]: relative cost of instruction k of ins_fast; occ: 2 / 4 waves per SIMD (256 / 128 VGPRs)
std::string fg_jit_mhmw_source(const fg_program *p, const std::vector<long long> &ins_cost, const std::vector<char> &generated, int rk, int split,
                               std::vector<double> *ctab_out, const std::vector<int> *rows_in = nullptr, int n_pri = -1, int n_fac = 0, bool no_stream = false, bool pipe = false, int nseg = 0 /* 2 .. 16: one statement segment per wave of a launch with that many waves per tile; else sixteen */, int ctl_share16 = 16 /* ... the control wave's share of a wave's statements, in sixteenths */,
                               int sum_pri = -1, int sum_lik = -1 /* >= 0: the tile's log_prior / log_likelihood term rows -- the control wave's in-order sums as straight-line code */,
                               const int *baked = nullptr /* {row-less terms, term rows, log_prior rows, site slots, waves per tile, exp_mask, pool_n} of every launch of this unit: literals in the kernel */,
                               int sums_form = 0 /* fg_jit_sums2: 0 plain statements; n >= 4: the two chains pinned side by side, rows n pairs ahead (FG_MH_SUMS_FORM overrides) */);   // the multi-wave stream MH kernel, the general records of phase B generated; pipe: around fg_mh_mw2_body.h's step loop
                               // (rows_in: a program without a score stream -- statement k's term row: log_prior rows [0, n_pri) first, the n_fac `factor` rows last)
int fg_jit_compile(const std::string &src, std::vector<char> &code, std::string &log);  // FG_OK / FG_E_UNSUPPORTED (no hiprtc) / FG_E_HIP
int fg_jit_get_code(const std::string &src, std::vector<char> &code, std::string &log);   // fg_jit_compile behind a per-process and an on-disk cache
#ifndef FG_JIT_NO_HIP
#include <hip/hip_runtime.h>
int fg_jit_bind_tables(hipModule_t mod, const std::vector<double> &tab, double **d_tab, hipStream_t stream);
#endif
