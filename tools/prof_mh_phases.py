"""Where a step of k_mh_mw_steps goes, by wave of tile 0: build with FG_EXTRA_DEFS=FG_MH_PROF to FG_LIB_PATH first
(tools/prof_mh_phases.sh).  Columns: cycles per step in  sums | accept+adapt+commit | proposal (other waves: random numbers) |
barrier 1 | phase B terms | barrier 2.
The pipelined loop (fg_mh_mw2_body.h; FG_MH_PIPE unset or 1): sums | decision + commit | select (proposer: candidates) | rest of phase A
(random numbers, waiting) | barrier 1 | phase B (proposer: bookkeeping first) | barrier 2."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
which = sys.argv[1] if len(sys.argv) > 1 else "ref"
if which == "ref":
    cp, C = E.compile_model(W.reference_model(20)), 65536
else:
    cp, C = E.compile_model(W.mixture(W.mixture_data(64)[0])), 262144
if len(sys.argv) > 2:
    C = int(sys.argv[2])
eng = E.Engine(cp, C, seed=1)
n = 200
for label, nw in (("adapting", 100000), ("sampling", 0)):
    eng.mh_init(nw); eng.mh_step(400); eng.synchronize()
    eng.mh_step(n); eng.synchronize()
    out = (ctypes.c_ulonglong * (16 * 8))()
    assert E.lib().fg_debug_mh_prof(out) == 0
    a = np.array(out, dtype=np.float64).reshape(16, 8) / n
    print(which, C, "chains", eng.mh_last_kernel(), label, "cycles per step (s_memtime ticks), waves 0..15:")
    for w in range(16):
        if a[w].sum() > 0:
            print("  wave %2d: " % w + " ".join("%7.0f" % x for x in a[w, :7]) + "   total %7.0f" % a[w, :7].sum() + ("   c7 %.0f" % (a[w, 7] * n) if a[w, 7] else ""))
