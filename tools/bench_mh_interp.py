"""adaptive_mcmc_chain on models that need the interpreter kernel k_mh_steps (FG_MH_MW=0 forces it for any model)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E
from tests.models import ZOO
for name, C, mw in [(n, c, m) for n in (sys.argv[1:] or ["alldists", "poisson_glm", "hier_logsigma", "logistic100"]) for c in (65536, 16384, 8192) for m in (0, 1, -1)]:
    os.environ["FG_JIT"] = "1" if mw < 0 else "0"            # -1: the model compiled at run time (fg_jit.cpp)
    mw = abs(mw)
    os.environ["FG_HMC_INTERP_MW"] = str(min(mw, 1))           # 0: k_mh_steps (one wave per tile); else k_mh_interp_mw_steps: 1 = the engine's W, else W
    if mw > 1: os.environ["FG_MH_INTERP_WAVES"] = str(mw)
    else: os.environ.pop("FG_MH_INTERP_WAVES", None)
    from fugue_amd import workloads as W
    cp = E.compile_model(W.logistic_regression(*W.classification_data(100)[:2]) if name == "logistic100" else ZOO[name]())
    eng = E.Engine(cp, C, seed=2)
    eng.mh_init(100)
    eng.mh_step(100); eng.synchronize()
    t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:12s} C={C:6d} multi-wave={mw} S={cp.S:3d} statements={cp.S + cp.O:4d} records={cp.stream_records}  {C * 200 / dt:.3e} chain-steps/s  accept {eng.mh_stats().accept_rate:.3f}  [{eng.mh_last_kernel()}]", flush=True)
    eng.close()
