import os, sys
sys.path.insert(0, os.getcwd())
os.environ["FG_HMC_INTERP_DEBUG"]="1"; os.environ["FG_HMC_INTERP_WAVES"]=sys.argv[1] if len(sys.argv) > 1 else "2"
from fugue_amd import engine as E, model as M
def chain(kind, n):
    P = M.Program()
    a = P.sample(M.addr("a"), M.Normal(0.0, 1.0))
    b = P.sample(M.addr("b"), M.Normal(0.0, 1.0))
    x = a * 0.001
    for i in range(n):
        if kind == "add": x = x + 0.5
        elif kind == "mulslot": x = x * b
        elif kind == "exp": x = M.exp(x * 0.5)
        elif kind == "ln": x = M.ln(x + 2.5)
    if kind == "obs":
        for i in range(n): P.observe(M.addr("y", i), M.Normal(a * 0.5, M.exp(b * 0.1)), 0.1 * i)
    elif kind == "pois":
        for i in range(n): P.observe(M.addr("y", i), M.Poisson(M.exp(a * 0.1 + 0.01 * i)), i % 5)
    else:
        P.factor(x * 1e-3)
    return P
import itertools
for (kind, n), occ, pl in itertools.product([("add", 200), ("mulslot", 200), ("exp", 100), ("obs", 50)], (2, 4), (0, 1)):
    os.environ["FG_HMC_INTERP_OCC"] = str(occ); os.environ["FG_HMC_INTERP_LDSPROG"] = str(pl)
    print("occ", occ, "program in LDS", pl, file=sys.stderr)
    cp = E.compile_model(chain(kind, n))
    print(kind, n, file=sys.stderr)
    eng = E.Engine(cp, 64, seed=2)
    eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE, n_leapfrog=4, init_step_size=0.01), 0)
    eng.hmc_step(2); eng.hmc_step(2); eng.synchronize(); eng.close()
