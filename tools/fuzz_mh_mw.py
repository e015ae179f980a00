"""Randomised cross-check of the multi-wave MH kernel against the one-wave kernel: the random programs of tests/random_models.py,
random chain counts, warmup / sampling lengths and waves per tile -- recorded draws,
final state, adapted scales, log-weights and accept counts must be identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fugue_amd as F
from fugue_amd import model as M
from fugue_amd import engine as E

from tests.random_models import random_program

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_models = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed0)

bad = 0
for it in range(n_models):
    try:
        cp = E.compile_model(random_program(seed0 * 100003 + it))
    except Exception as ex:
        print("model", it, "not compilable:", repr(ex)[:100]); continue
    C, nw, ns = int(rng.integers(1, 300)), int(rng.integers(0, 60)), int(rng.integers(1, 40))
    W = int(rng.choice([0, 2, 4, 8, 16]))
    out = []
    for mw in (1, 0):
        os.environ["FG_MH_MW"] = str(mw)
        os.environ["FG_HMC_WAVES"] = str(W if mw else 0)
        eng = E.Engine(cp, C, seed=500 + it, chain_offset=it)
        buf = eng.device_alloc(max(1, ns * cp.S * C) * 8)
        st = eng.mh_run(ns, nw, None, list(range(cp.S)), buf)
        out.append((eng.download(buf, (ns, cp.S, C), dtype=np.int64), eng.get_values(), eng.mh_scales(), eng.mh_log_weight(), st.accept_rate))
        eng.device_free(buf); eng.close()
    ok = all(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True) for a, b in zip(out[0], out[1]))
    bad += 0 if ok else 1
    print(f"model {it:3d}: S={cp.S:2d} O={cp.O:2d} records={cp.stream_records} C={C:3d} warm={nw:2d} n={ns:2d} W={W:2d} accept={out[0][4]:.3f} -> {'identical' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
