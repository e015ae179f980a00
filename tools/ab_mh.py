"""A/B of the MH kernels: reference_model(20) at 65 536 chains and the C5 mixture at 262 144 chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fugue_amd import engine as E, workloads as W
which = sys.argv[1] if len(sys.argv) > 1 else "ref"
if which == "ref":
    cp, C, nw, ns = E.compile_model(W.reference_model(20)), 65536, 200, 400
else:
    cp, C, nw, ns = E.compile_model(W.mixture(W.mixture_data(64)[0])), 262144, 200, 200
eng = E.Engine(cp, C, seed=1)
eng.mh_init(nw); eng.mh_step(50); eng.synchronize()
eng.mh_init(nw)
t0 = time.perf_counter(); eng.mh_step(nw); eng.synchronize(); t1 = time.perf_counter(); eng.mh_step(ns); eng.synchronize(); t2 = time.perf_counter()
print(f"{which} MW={os.environ.get('FG_MH_MW','1')} W={os.environ.get('FG_HMC_WAVES','auto')}: warmup {C*nw/(t1-t0):.3e} steps/s, sampling {C*ns/(t2-t1):.3e} steps/s, all {C*(nw+ns)/(t2-t0):.3e}, accept {eng.mh_stats().accept_rate:.4f}", flush=True)
