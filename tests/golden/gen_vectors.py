#!/usr/bin/env python3
"""Generates tests/golden/logpdf_vectors.json.  Run in the BUILD container only.

Part A imports the reference's own Python helper (/root/reference/tests/gen_refs.py, the only
Python in the reference; its gamma/beta/studentt/chi2/invgamma pdfs "match the
src/core/distribution.rs formulas exactly") and tabulates log(pdf) over parameter grids.
Part B tabulates the remaining families with scipy.stats (the expressions the reference's KAT
comments cite, tests/f_dist_distributions.rs:4-9) -- independent of the oracle.
Only the resulting numbers are committed; nothing of the reference travels.
"""
import importlib.util, json, math, os, sys
import numpy as np
from scipy import stats

out = {"_about": "log-pdf grids: part A from the reference's tests/gen_refs.py, part B from scipy.stats",
       "gen_refs": [], "scipy": []}

ref = "/root/reference/tests/gen_refs.py"
if os.path.exists(ref):
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("gen_refs", ref)
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    xs = [0.05, 0.3, 0.5, 0.9, 1.5, 3.0, 7.5]
    for shape, rate in [(0.5, 1.0), (2.0, 1.0), (3.0, 2.0), (7.5, 0.25)]:
        f = g.gamma_pdf(shape, rate)
        out["gen_refs"] += [dict(dist="Gamma", params=[shape, rate], x=x, expected=math.log(f(x))) for x in xs]
        f = g.invgamma_pdf(shape, rate)
        out["gen_refs"] += [dict(dist="InverseGamma", params=[shape, rate], x=x, expected=math.log(f(x))) for x in xs]
    for a, b in [(0.5, 0.5), (2.0, 3.0), (2.0, 5.0), (9.0, 5.0)]:
        f = g.beta_pdf(a, b)
        out["gen_refs"] += [dict(dist="Beta", params=[a, b], x=x, expected=math.log(f(x))) for x in (0.01, 0.2, 0.5, 0.77, 0.99)]
    for df, loc, sc in [(3.0, 1.0, 2.0), (5.0, 0.0, 1.0), (10.0, 2.0, 0.5), (1.0, 0.0, 1.0)]:
        f = g.studentt_pdf(df, loc, sc)
        out["gen_refs"] += [dict(dist="StudentT", params=[df, loc, sc], x=x, expected=math.log(f(x))) for x in (-4.0, -0.5, 0.0, 1.0, 2.5, 9.0)]
    for k in (1.0, 2.5, 4.0, 11.0):
        f = g.chi2_pdf(k)
        out["gen_refs"] += [dict(dist="ChiSquared", params=[k], x=x, expected=math.log(f(x))) for x in xs]

S = out["scipy"]
xr = [-3.0, -0.7, 0.0, 0.4, 2.5, 11.0]
for mu, sg in [(0.0, 1.0), (1.0, 2.0), (-2.0, 0.3)]:
    S += [dict(dist="Normal", params=[mu, sg], x=x, expected=float(stats.norm.logpdf(x, mu, sg))) for x in xr]
    S += [dict(dist="LogNormal", params=[mu, sg], x=x, expected=float(stats.lognorm.logpdf(x, sg, scale=math.exp(mu)))) for x in (0.05, 0.4, 1.0, 2.5, 11.0)]
    S += [dict(dist="Cauchy", params=[mu, sg], x=x, expected=float(stats.cauchy.logpdf(x, mu, sg))) for x in xr]
    S += [dict(dist="Laplace", params=[mu, sg], x=x, expected=float(stats.laplace.logpdf(x, mu, sg))) for x in xr]
for lo, hi in [(-2.0, 2.0), (0.0, 1.0), (3.0, 10.5)]:
    S += [dict(dist="Uniform", params=[lo, hi], x=x, expected=float(-math.log(hi - lo))) for x in (lo, 0.5 * (lo + hi))]
for r in (0.5, 2.0, 9.0):
    S += [dict(dist="Exponential", params=[r], x=x, expected=float(stats.expon.logpdf(x, scale=1 / r))) for x in (0.0, 0.3, 1.0, 7.0)]
for k, lam in [(1.5, 2.0), (2.0, 1.5), (0.7, 3.0)]:
    S += [dict(dist="Weibull", params=[k, lam], x=x, expected=float(stats.weibull_min.logpdf(x, k, scale=lam))) for x in (0.1, 1.0, 2.0, 6.0)]
for n, p in [(10, 0.5), (20, 0.3), (100, 0.07)]:
    S += [dict(dist="Binomial", params=[n, p], x=k, expected=float(stats.binom.logpmf(k, n, p))) for k in (0, 1, 5, 7, 10)]
for lam in (0.3, 3.0, 4.0, 50.0, 800.0):
    S += [dict(dist="Poisson", params=[lam], x=k, expected=float(stats.poisson.logpmf(k, lam))) for k in (0, 1, 2, 7, 60)]
for p in (0.3, 0.5, 0.9):
    S += [dict(dist="Bernoulli", params=[p], x=1, expected=math.log(p)), dict(dist="Bernoulli", params=[p], x=0, expected=math.log(1 - p))]
for lo, hi in [(-2, 5), (0, 10), (1, 6)]:
    S += [dict(dist="DiscreteUniform", params=[lo, hi], x=lo, expected=-math.log(hi - lo + 1)), dict(dist="DiscreteUniform", params=[lo, hi], x=hi, expected=-math.log(hi - lo + 1))]

json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "logpdf_vectors.json"), "w"), indent=1)
print(len(out["gen_refs"]), len(out["scipy"]))
