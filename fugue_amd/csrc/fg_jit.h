// fg_jit.h -- run-time compiled model kernels (fg_jit.cpp)
#pragma once
#include <string>
#include <vector>

struct fg_program;
std::string fg_jit_hmc_source(const fg_program *p);                                     // "" = not covered by the generator
std::string fg_jit_mh_source(const fg_program *p, const std::vector<long long> &ins_cost, int occ);   // ins_cost[k]: relative cost of instruction k of ins_fast; occ: 2 / 4 waves per SIMD (256 / 128 VGPRs)
int fg_jit_compile(const std::string &src, std::vector<char> &code, std::string &log);  // FG_OK / FG_E_UNSUPPORTED (no hiprtc) / FG_E_HIP
int fg_jit_get_code(const std::string &src, std::vector<char> &code, std::string &log);   // fg_jit_compile behind a per-process and an on-disk cache
