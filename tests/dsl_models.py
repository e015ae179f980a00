"""Model-language sources (restating the test models of crates/fugue-wasm/src/dsl.rs:1149-1328 plus a
few that stress the evaluator) paired with the same model written against the Python mirror, so a
DSL-built site program can be checked against a hand-built one and, through it, against the oracle."""
import numpy as np

from fugue_amd import model as M

COIN = '''
    let p <- sample(addr!("p"), Beta(2.0, 2.0));
    for i in 0..data.len() {
        observe(addr!("flip", i), Bernoulli(p), data[i]);
    }
    pure(p)
'''
COIN_DATA = "[1,0,1,1,0,1,1,0,1,1]"      # 7 heads, 3 tails -> posterior Beta(9, 5)


def coin_mirror():
    P = M.Program()
    p = P.sample(M.addr("p"), M.Beta(2.0, 2.0))
    for i, y in enumerate([1, 0, 1, 1, 0, 1, 1, 0, 1, 1]):
        P.observe(M.addr("flip", i), M.Bernoulli(p), float(y))
    return P


REGRESSION = '''
    let a <- sample(addr!("a"), Normal(0.0, 2.5));
    let b <- sample(addr!("b"), Normal(0.0, 2.5));
    for i in 0..x.len() {
        observe(addr!("y", i), Normal(a * x[i] + b, 0.8), y[i]);
    }
    pure(a)
'''
REGRESSION_DATA = {"x": [-2, -1, 0, 1, 2], "y": [-2.1, -1.3, -0.4, 0.5, 1.2]}


def regression_mirror():
    P = M.Program()
    a = P.sample(M.addr("a"), M.Normal(0.0, 2.5))
    b = P.sample(M.addr("b"), M.Normal(0.0, 2.5))
    for i, (x, y) in enumerate(zip(REGRESSION_DATA["x"], REGRESSION_DATA["y"])):
        P.observe(M.addr("y", i), M.Normal(a * float(x) + b, 0.8), y)
    return P


INDEXED = '''
    for i in 0..3 {
        let z <- sample(addr!("z", i), Normal(0.0, 1.0));
    }
    pure(0.0)
'''


def indexed_mirror():
    P = M.Program()
    for i in range(3):
        P.sample(M.addr("z", i), M.Normal(0.0, 1.0))
    return P


SUGAR = '''
    let mu <- sample(addr!("mu"), Normal::new(0.0, 1.0).unwrap());
    observe(addr!("y"), Normal::new(mu, 1.0).unwrap(), 0.5);
    pure(mu)
'''


def sugar_mirror():
    P = M.Program()
    mu = P.sample(M.addr("mu"), M.Normal(0.0, 1.0))
    P.observe(M.addr("y"), M.Normal(mu, 1.0), 0.5)
    return P


FACTOR_MATH = '''
    let x <- sample(addr!("x"), Normal(0.0, 1.0));
    factor(-0.5 * pow(x - 1.0, 2.0));
    pure(exp(x) / (1.0 + exp(x)))
'''


def factor_math_mirror():
    P = M.Program()
    x = P.sample(M.addr("x"), M.Normal(0.0, 1.0))
    P.factor(-0.5 * M.powf(x - 1.0, 2.0))
    return P


DISCRETE = '''
    let k <- sample(addr!("k"), Poisson(4.0));
    let z <- sample(addr!("z"), Categorical(0.3, 0.7));
    observe(addr!("n"), Binomial(10, 0.5), 7);
    observe(addr!("flag"), Bernoulli(0.5), true);
    pure(k + z)
'''


def discrete_mirror():
    P = M.Program()
    P.sample(M.addr("k"), M.Poisson(4.0))
    P.sample(M.addr("z"), M.Categorical([0.3, 0.7]))
    P.observe(M.addr("n"), M.Binomial(10, 0.5), 7)
    P.observe(M.addr("flag"), M.Bernoulli(0.5), 1.0)
    return P


# evaluator stress: comments, let-bound derived values, integer arithmetic in addresses and bounds,
# nested loops, every math function, a data array indexed by a sampled Categorical, unary minus
HIERARCHY = '''
    // two groups with their own means, shared scale
    let tau <- sample(addr!("tau"), Gamma(2.0, 2.0));
    let scale = sqrt(1.0 / tau);
    let z <- sample(addr!("z"), Categorical(0.25, 0.75));
    let shift = offsets[z];
    for g in 0..2 {
        let m <- sample(addr!("m", g * 10 + 1), Normal(shift, 2.0));
        for j in 0..3 {
            observe(addr!("obs", g * 3 + j), Normal(m, scale), ys[g * 3 + j]);
        }
    }
    let w <- sample(addr!("w"), Uniform(-1.0, 1.0));
    factor(-abs(w) + min(tanh(w), 0.5) - max(sin(w), cos(w)) + ln(1.0 + exp(-w)) + floor(2.0 * w) * 0.0);
    pure(scale)
'''
HIERARCHY_DATA = {"offsets": [-1.5, 2.0], "ys": [0.1, -0.4, 0.9, 2.2, 1.7, 2.9]}


def hierarchy_mirror():
    P = M.Program()
    tau = P.sample(M.addr("tau"), M.Gamma(2.0, 2.0))
    scale = M.sqrt(1.0 / tau)
    z = P.sample(M.addr("z"), M.Categorical([0.25, 0.75]))
    shift = M.select(z, HIERARCHY_DATA["offsets"])
    ys = HIERARCHY_DATA["ys"]
    for g in range(2):
        m = P.sample(M.addr("m", g * 10 + 1), M.Normal(shift, 2.0))
        for j in range(3):
            P.observe(M.addr("obs", g * 3 + j), M.Normal(m, scale), ys[g * 3 + j])
    w = P.sample(M.addr("w"), M.Uniform(-1.0, 1.0))
    P.factor(-M.fabs(w) + M.fmin(M.tanh(w), 0.5) - M.fmax(M.sin(w), M.cos(w)) + M.ln(1.0 + M.exp(-w))
             + M.floor(2.0 * w) * 0.0)
    return P


PAIRS = {
    "coin": (COIN, COIN_DATA, coin_mirror),
    "regression": (REGRESSION, REGRESSION_DATA, regression_mirror),
    "indexed": (INDEXED, None, indexed_mirror),
    "sugar": (SUGAR, "", sugar_mirror),
    "factor_math": (FACTOR_MATH, None, factor_math_mirror),
    "discrete": (DISCRETE, None, discrete_mirror),
    "hierarchy": (HIERARCHY, HIERARCHY_DATA, hierarchy_mirror),
}
