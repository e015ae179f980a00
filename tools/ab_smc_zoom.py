"""next_beta with and without the zoom passes (FG_SMC_ZOOM): the ladders must agree to the last digits; run time per ladder."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
from fugue_amd import engine as E, workloads as W
sys.path.insert(0, os.path.join(%r, "tests"))
out = {}
for name, prog, N, R in (("smc_normal", W.smc_normal(), 1 << 20, 3), ("smc_normal_small", W.smc_normal(), 5000, 2), ("normal8", W.normal_sites(8), 1 << 16, 2),
                         ("refmodel8", W.reference_model(8), 1 << 16, 2), ("refmodel20", W.reference_model(20), 1 << 15, 1), ("normal32", W.normal_sites(32), 1 << 15, 1)):
    eng = E.Engine(E.compile_model(prog), N, seed=7)
    r = eng.smc_run(rejuvenation_steps=R, download=False)
    t0 = time.perf_counter(); r = eng.smc_run(rejuvenation_steps=R, download=False); dt = time.perf_counter() - t0
    out[name] = {"betas": [float(b) for b in r["betas"]], "logZ": float(r["log_evidence"]), "ms": dt * 1e3}
    eng.close()
print("RESULT" + json.dumps(out))
''' % (ROOT, ROOT)
res = {}
for z in ("0", "1"):
    env = dict(os.environ, FG_SMC_ZOOM=z)
    o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
    if not line: print(o.stdout[-2000:], o.stderr[-2000:]); sys.exit(1)
    res[z] = json.loads(line[0][6:])
for name in res["0"]:
    a, b = res["0"][name], res["1"][name]
    worst = max([abs(x - y) for x, y in zip(a["betas"], b["betas"])] + [0.0])
    print(f"{name:18s} steps {len(a['betas'])}/{len(b['betas'])}  max |d beta| {worst:.3e}  d logZ {abs(a['logZ'] - b['logZ']):.3e}  ms {a['ms']:.3f} -> {b['ms']:.3f}")
    print("   betas(plain):", " ".join(f"{x:.17g}" for x in a["betas"][:6]), "..." if len(a["betas"]) > 6 else "")
    print("   betas(zoom): ", " ".join(f"{x:.17g}" for x in b["betas"][:6]), "..." if len(b["betas"]) > 6 else "")
