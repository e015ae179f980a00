"""hmc_chain throughput of the parity-test models at 65 536 chains: stream kernel vs (FG_NO_STREAM=1) interpreter kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E
from tests.models import ZOO
C, L = 65536, 16
for name in sys.argv[1:] or ["hier_scale", "hier", "linreg", "refmodel8", "alldists"]:
    cp = E.compile_model(ZOO[name]())
    eng = E.Engine(cp, C, seed=2)
    eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=L), 20)
    eng.hmc_step(20); eng.synchronize()
    n = 20
    t0 = time.perf_counter(); eng.hmc_step(n); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:12s} d={cp.d:3d} statements={cp.S + cp.O:4d} records={cp.stream_records}  {C * n * L / dt:.3e} leapfrog-steps/s"
          f"  ({'interpreter kernel' if os.environ.get('FG_NO_STREAM') else 'stream kernel' if cp.stream_records[0] else 'interpreter kernel'})", flush=True)
