"""Phase stamps of k_smc_ess2_pass (experiment build with FG_SMC_PROF): per pass, block 0 and the last block, microseconds from
the pass's first stamp, and the distance between consecutive passes' first stamps (= the pass as the stream sees it)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fugue_amd import engine as E, workloads as W
eng = E.Engine(E.compile_model(W.smc_normal()), 1 << 20, seed=42)
for _ in range(3):
    r = eng.smc_run(rejuvenation_steps=3, download=False)
buf = np.zeros((64, 2, 8), dtype=np.int64)
assert E.lib().fg_debug_smc_prof(buf.ctypes.data_as(C.c_void_p)) == 0
names = ["start", "state", "collect", "decide", "sums", "reduce", "store"]
print("the LAST tempering step's passes overwrite the first two rows; rows 2.. are the first step's passes (stamps in us since the pass's first stamp)")
prev = None
for p in range(24):
    t = buf[p]
    if t[0, 0] == 0: continue
    row = " ".join(f"{names[i]} {((t[0, i] - t[0, 0]) / 100.0 if t[0, i] else float('nan')):5.2f}|{((t[1, i] - t[0, 0]) / 100.0 if t[1, i] else float('nan')):5.2f}" for i in range(7))
    gap = (t[0, 0] - prev) / 100.0 if prev else float("nan")
    print(f"pass {p:2d}: since previous pass start {gap:6.2f} us | block0|last: {row}")
    prev = t[0, 0]
