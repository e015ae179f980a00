"""FG_GRAD_FD_DENSE (the reference's arithmetic verbatim) on programs outside the independent-sites / dense-stream kernels: the unit compiled at
run time (fg_jit_full_k) against the interpreter kernels (FG_JIT=0), leapfrog-steps/s at 65 536 chains."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from fugue_amd import engine as E
    from tests.models import ZOO
    for name in ("alldists", "logistic", "poisson_glm", "hier_logsigma", "hier_scale", "mixture"):
        cp = E.compile_model(ZOO[name]())
        C = 65536
        eng = E.Engine(cp, C, seed=1)
        eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_DENSE), 5); eng.hmc_step(5); eng.synchronize()
        t0 = time.perf_counter(); eng.hmc_step(10); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"FG_JIT={os.environ.get('FG_JIT', '1')} {name:14s} C={C} {C * 10 * 16 / dt:.3e} [{eng.hmc_last_kernel()[:44]}]", flush=True)
        eng.close()
else:
    for jit in ("0", "1"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, FG_JIT=jit))
