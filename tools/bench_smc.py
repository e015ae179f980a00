"""C4: adaptive_smc, 1 048 576 particles (examples/smc_inference.rs model), Systematic / 0.5 / 3 rejuvenation moves."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
N = int(os.environ.get("FG_PARTICLES", 1 << 20))
eng = E.Engine(E.compile_model(W.smc_normal()), N, seed=42)
eng.smc_run(rejuvenation_steps=3, download=False)
for _ in range(3):
    t0 = time.perf_counter(); r = eng.smc_run(rejuvenation_steps=3, download=False); dt = time.perf_counter() - t0
    print(f"smc {N} particles: {dt * 1e3:.2f} ms, steps {len(r['betas'])}, model runs {r['n_model_runs']}, logZ {r['log_evidence']:.6f}", flush=True)
