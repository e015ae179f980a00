"""Where a transition of k_hmc_sep_steps goes, by wave of tile 0: build with FG_EXTRA_DEFS=FG_HMC_PROF,FG_MH_PROF to FG_LIB_PATH first
(tools/prof_mh_phases.sh).  Columns: cycles per transition in  momenta + trajectories + endpoint terms | barrier | wave 0's in-order sums,
accept, dual averaging | barrier | commit / draws."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
C = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = E.Engine(E.compile_model(W.normal_sites(32)), C, seed=1)
n = 100
for label, nw in (("adapting", 100000), ("sampling", 0)):
    eng.hmc_init(E.hmc_config(), nw); eng.hmc_step(25); eng.synchronize()
    eng.hmc_step(n); eng.synchronize()
    out = (ctypes.c_ulonglong * (16 * 8))()
    assert E.lib().fg_debug_hmc_prof(out) == 0
    a = np.array(out, dtype=np.float64).reshape(16, 8) / n
    print("north-star model,", C, "chains,", label, ": cycles per transition (s_memtime ticks), waves of tile 0:")
    for w in range(16):
        if a[w].sum() > 0:
            print("  wave %2d: " % w + " ".join("%7.0f" % x for x in a[w, :5]) + "   total %7.0f" % a[w, :5].sum())
