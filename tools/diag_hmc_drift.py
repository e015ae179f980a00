import numpy as np, sys
sys.path.insert(0, ".")
from fugue_amd import engine as E, workloads as W
from oracle import oracle as orc
for name, prog in (("readme", W.readme_normal()), ("normal32", W.normal_sites(32))):
    cp, om = E.compile_model(prog), orc.OracleModel(prog)
    C, nw, ns = 96, 0, 50
    # all draws recorded: run with n_warmup=nw adaptive by stepping manually
    for nw in (0, 25):
        eng = E.Engine(cp, C, seed=5, chain_offset=7)
        cfg = E.hmc_config(n_leapfrog=8)
        tot = 50
        d = eng.device_alloc((tot - nw) * cp.d * C * 8)
        eng.hmc_run(cfg, tot - nw, nw, d)
        draws = eng.download(d, (tot - nw, cp.d, C))
        od, _, oe, _ = om.hmc_run(5, C, nw, tot - nw, orc.HmcConfig.default(n_leapfrog=8), chain0=7, n_threads=8)
        rel = np.abs(draws - od) / (1e-3 + np.abs(od))
        print(name, "nw", nw, "max rel err per recorded transition:", np.array2string(rel.max(axis=(1, 2)), precision=1, max_line_width=250))
        print("   eps rel diff max", np.max(np.abs(eng.hmc_step_sizes() - oe) / oe))
