"""adaptive_smc on programs without a score stream: interpreter rejuvenation (FG_JIT=0) against the compiled model (default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E
from tests.models import ZOO
N = int(os.environ.get("FG_PARTICLES", 262144))
for name in ("alldists", "logistic", "poisson_glm", "hier_logsigma"):
    cp = E.compile_model(ZOO[name]())
    for jit in ("0", "1"):
        os.environ["FG_JIT"] = jit
        eng = E.Engine(cp, N, seed=5)
        eng.smc_run(rejuvenation_steps=3, download=False)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); r = eng.smc_run(rejuvenation_steps=3, download=False); best = min(best, time.perf_counter() - t0)
        moves = (r["n_model_runs"] - N) / 2
        print(f"{name:14s} N={N} FG_JIT={jit}: {best * 1e3:8.2f} ms per run, {len(r['betas'])} tempering steps, {moves / best:.3e} particle-moves/s", flush=True)
        eng.close()
