"""Host-side mirror of Fugue's model surface (`Model<A>`, `sample/observe/factor/pure`,
`bind/map`, `addr!`, the 17 distributions) for the GPU engine.

The reference builds a CPS `Model<A>` whose continuations receive *runtime* values
(`/root/reference/src/core/model.rs:20-131,447-581`).  The engine needs a fixed-structure,
first-order *site program* instead, so here the same combinators are run once with
**symbolic** values (`Expr`): every `bind` continuation is called with an expression
standing for the sampled value, and the trace of `sample/observe/factor` effects becomes
a `Program` (statements in program order + expression trees).  Models whose address set
or control flow depends on a sampled value cannot be traced this way and are refused
(`StructureError`), matching SURVEY.md section 7 "hard parts".

Nothing here computes densities or draws numbers: `Program` is only a description that
`fugue_amd.engine` lowers onto the C-ABI (`include/fugue_amd.h`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Union

import numpy as np

# --------------------------------------------------------------------------------------
# Error taxonomy: numeric codes of `ErrorCode` (/root/reference/src/error.rs:40-59)
# --------------------------------------------------------------------------------------
class ErrorCode:
    InvalidMean = 100
    InvalidVariance = 101
    InvalidProbability = 102
    InvalidRange = 103
    InvalidShape = 104
    InvalidRate = 105
    InvalidCount = 106
    AddressConflict = 301
    UnexpectedModelStructure = 302
    TraceAddressNotFound = 500
    TypeMismatch = 600


class FugueError(Exception):
    def __init__(self, message: str, code: int):
        super().__init__(f"{message} (ErrorCode {code})")
        self.code = code


class StructureError(FugueError):
    def __init__(self, message: str):
        super().__init__(message, ErrorCode.UnexpectedModelStructure)


# --------------------------------------------------------------------------------------
# Addresses: `addr!` encoding (/root/reference/src/core/address.rs:189-223)
# --------------------------------------------------------------------------------------
def escape_addr_segment(segment: str) -> str:
    if "\\" in segment or "#" in segment:
        return segment.replace("\\", "\\\\").replace("#", "\\#")
    return segment


def addr(name, index=None) -> str:
    """`addr!("name")` / `addr!("name", i)` -> `"name"` / `"name#i"` with `\\`-escaping."""
    if index is None:
        return escape_addr_segment(str(name))
    return f"{escape_addr_segment(str(name))}#{escape_addr_segment(str(index))}"


# --------------------------------------------------------------------------------------
# Symbolic expressions (superset of the DSL `Expr`,
# /root/reference/crates/fugue-wasm/src/dsl.rs:92-102,569-582)
# --------------------------------------------------------------------------------------
UNARY = ("neg", "exp", "ln", "sqrt", "abs", "floor", "sin", "cos", "tanh")
BINARY = ("add", "sub", "mul", "div", "pow", "min", "max")


class Expr:
    __slots__ = ("op", "args", "value", "a", "b")

    def __init__(self, op, args=(), value=0.0, a=0, b=0):
        self.op = op          # const | site | data | unary | binary | clamp | select
        self.args = tuple(args)
        self.value = float(value)
        self.a = int(a)       # site handle / data array id
        self.b = int(b)       # data index

    # arithmetic -----------------------------------------------------------------------
    def __add__(self, o): return Expr("add", (self, as_expr(o)))
    def __radd__(self, o): return Expr("add", (as_expr(o), self))
    def __sub__(self, o): return Expr("sub", (self, as_expr(o)))
    def __rsub__(self, o): return Expr("sub", (as_expr(o), self))
    def __mul__(self, o): return Expr("mul", (self, as_expr(o)))
    def __rmul__(self, o): return Expr("mul", (as_expr(o), self))
    def __truediv__(self, o): return Expr("div", (self, as_expr(o)))
    def __rtruediv__(self, o): return Expr("div", (as_expr(o), self))
    def __neg__(self): return Expr("neg", (self,))
    def __pow__(self, o): return Expr("pow", (self, as_expr(o)))
    def clamp(self, lo, hi): return Expr("clamp", (self, as_expr(lo), as_expr(hi)))
    def exp(self): return Expr("exp", (self,))
    def ln(self): return Expr("ln", (self,))
    def sqrt(self): return Expr("sqrt", (self,))
    def abs(self): return Expr("abs", (self,))
    def powf(self, o): return Expr("pow", (self, as_expr(o)))
    def min(self, o): return Expr("min", (self, as_expr(o)))
    def max(self, o): return Expr("max", (self, as_expr(o)))

    # comparisons of sampled values: only `guard` consumes them (model.rs:710-716 guards are weights, not structure)
    def __lt__(self, o): return Cond(as_expr(o) - self, strict=True)
    def __gt__(self, o): return Cond(self - as_expr(o), strict=True)
    def __le__(self, o): return Cond(as_expr(o) - self, strict=False)
    def __ge__(self, o): return Cond(self - as_expr(o), strict=False)

    def is_const(self) -> bool:
        return self.op == "const"

    def __bool__(self):
        raise StructureError("a sampled value was used in Python control flow; "
                             "structure-varying models cannot be flattened (use select())")

    __eq__ = object.__eq__
    __hash__ = object.__hash__

    def __repr__(self):
        if self.op == "const":
            return repr(self.value)
        if self.op == "site":
            return f"site[{self.a}]"
        if self.op == "data":
            return f"data{self.a}[{self.b}]"
        return f"{self.op}({', '.join(map(repr, self.args))})"


class Cond:
    """`a < b` / `a <= b` on expressions of sampled values: true iff `margin` > 0 (strict) or >= 0 (non-strict).  Not a Python
    bool -- a model whose STRUCTURE depends on it cannot be flattened -- but `guard(cond)` is only a weight (0 or -inf), which
    the site program can express.  Strictness matters wherever equality has positive probability: predicates on discrete
    sites, or on expressions through floor / min / max / clamp (`guard(k >= 1)`, `guard(sigma.max(0.01) >= 0.01)`)."""
    __slots__ = ("margin", "strict")

    def __init__(self, margin: "Expr", strict: bool = True):
        self.margin = margin
        self.strict = strict

    def __bool__(self):
        raise StructureError("a comparison of sampled values was used in Python control flow; only guard(...) accepts it")


Number = Union[int, float, Expr]


def as_expr(x: Number) -> Expr:
    if isinstance(x, Expr):
        return x
    if isinstance(x, (bool, np.bool_)):
        return Expr("const", value=1.0 if x else 0.0)
    return Expr("const", value=float(x))


def exp(x): return as_expr(x).exp()
def ln(x): return as_expr(x).ln()
def sqrt(x): return as_expr(x).sqrt()
def fabs(x): return as_expr(x).abs()
def floor(x): return Expr("floor", (as_expr(x),))
def sin(x): return Expr("sin", (as_expr(x),))
def cos(x): return Expr("cos", (as_expr(x),))
def tanh(x): return Expr("tanh", (as_expr(x),))
def powf(x, y): return Expr("pow", (as_expr(x), as_expr(y)))
def fmin(x, y): return Expr("min", (as_expr(x), as_expr(y)))
def fmax(x, y): return Expr("max", (as_expr(x), as_expr(y)))
def clamp(x, lo, hi): return as_expr(x).clamp(lo, hi)


def select(index: Number, options: Sequence[Number]) -> Expr:
    """`options[index]` for a sampled integer index (the `if z == 0 {..} else {..}` of
    /root/reference/examples/mixture_models.rs:91-107, made first-order)."""
    return Expr("select", (as_expr(index),) + tuple(as_expr(o) for o in options))


class DataArray:
    """A named data array bound into the program (DSL `{"y": [...]}`, dsl.rs:1066-1106)."""

    def __init__(self, program: "Program", array_id: int, values: np.ndarray):
        self._program, self.array_id, self.values = program, array_id, values

    def __len__(self):
        return len(self.values)

    def __getitem__(self, i) -> Expr:
        i = int(i)
        if i < 0 or i >= len(self.values):
            raise IndexError(i)
        return Expr("data", a=self.array_id, b=i, value=float(self.values[i]))


# --------------------------------------------------------------------------------------
# Distributions (constructors validate constant parameters like `Dist::new`,
# /root/reference/src/core/distribution.rs:133-152 ... 1842-1853)
# --------------------------------------------------------------------------------------
# kinds in the order of the crate-root re-export list (/root/reference/src/lib.rs:18-22)
DIST_KINDS = ["Bernoulli", "Beta", "Binomial", "Categorical", "Cauchy", "ChiSquared",
              "DiscreteUniform", "Exponential", "Gamma", "InverseGamma", "Laplace", "LogNormal",
              "Normal", "Poisson", "StudentT", "Uniform", "Weibull"]
F64, BOOL, U64, USIZE, I64 = range(5)
VTYPE = {"Bernoulli": BOOL, "Categorical": USIZE, "Binomial": U64, "Poisson": U64, "DiscreteUniform": I64}


def _cval(e: Expr) -> Optional[float]:
    return e.value if e.op in ("const", "data") else None


@dataclass
class Dist:
    name: str
    params: List[Expr]
    i64_bounds: Optional[tuple] = None     # DiscreteUniform::new(lo: i64, hi: i64) with exact integer bounds

    @property
    def kind(self) -> int:
        return DIST_KINDS.index(self.name)

    @property
    def vtype(self) -> int:
        return VTYPE.get(self.name, F64)


def _check(cond_bad: Callable[[float], bool], e: Expr, dist: str, what: str, code: int):
    v = _cval(e)
    if v is not None and cond_bad(v):
        raise FugueError(f"{dist}: invalid {what} {v}", code)


_nonfinite = lambda v: not math.isfinite(v)
_nonpos = lambda v: v <= 0.0 or not math.isfinite(v)
_notprob = lambda v: (not math.isfinite(v)) or v < 0.0 or v > 1.0


def Normal(mu, sigma) -> Dist:
    mu, sigma = as_expr(mu), as_expr(sigma)
    _check(_nonfinite, mu, "Normal", "mean", ErrorCode.InvalidMean)
    _check(_nonpos, sigma, "Normal", "sigma", ErrorCode.InvalidVariance)
    return Dist("Normal", [mu, sigma])


def Uniform(low, high) -> Dist:
    low, high = as_expr(low), as_expr(high)
    _check(_nonfinite, low, "Uniform", "bound", ErrorCode.InvalidRange)
    _check(_nonfinite, high, "Uniform", "bound", ErrorCode.InvalidRange)
    if _cval(low) is not None and _cval(high) is not None and _cval(low) >= _cval(high):
        raise FugueError("Uniform: low >= high", ErrorCode.InvalidRange)
    return Dist("Uniform", [low, high])


def LogNormal(mu, sigma) -> Dist:
    mu, sigma = as_expr(mu), as_expr(sigma)
    _check(_nonfinite, mu, "LogNormal", "mean", ErrorCode.InvalidMean)
    _check(_nonpos, sigma, "LogNormal", "sigma", ErrorCode.InvalidVariance)
    return Dist("LogNormal", [mu, sigma])


def Exponential(rate) -> Dist:
    rate = as_expr(rate)
    _check(_nonpos, rate, "Exponential", "rate", ErrorCode.InvalidRate)
    return Dist("Exponential", [rate])


def Bernoulli(p) -> Dist:
    p = as_expr(p)
    _check(_notprob, p, "Bernoulli", "probability", ErrorCode.InvalidProbability)
    return Dist("Bernoulli", [p])


def Categorical(probs: Sequence[Number]) -> Dist:
    ps = [as_expr(p) for p in probs]
    if len(ps) == 0:
        raise FugueError("Categorical: probability vector cannot be empty", ErrorCode.InvalidProbability)
    if len(ps) > 64:
        raise FugueError("Categorical: at most 64 categories", ErrorCode.InvalidCount)
    cv = [_cval(p) for p in ps]
    if all(v is not None for v in cv):
        if abs(sum(cv) - 1.0) > 1e-6:
            raise FugueError("Categorical: probabilities must sum to 1.0", ErrorCode.InvalidProbability)
        if any((not math.isfinite(v)) or v < 0.0 for v in cv):
            raise FugueError("Categorical: probabilities must be non-negative and finite",
                             ErrorCode.InvalidProbability)
    return Dist("Categorical", ps)


def Beta(alpha, beta) -> Dist:
    alpha, beta = as_expr(alpha), as_expr(beta)
    _check(_nonpos, alpha, "Beta", "alpha", ErrorCode.InvalidShape)
    _check(_nonpos, beta, "Beta", "beta", ErrorCode.InvalidShape)
    return Dist("Beta", [alpha, beta])


def Gamma(shape, rate) -> Dist:
    shape, rate = as_expr(shape), as_expr(rate)
    _check(_nonpos, shape, "Gamma", "shape", ErrorCode.InvalidShape)
    _check(_nonpos, rate, "Gamma", "rate", ErrorCode.InvalidRate)
    return Dist("Gamma", [shape, rate])


def Binomial(n, p) -> Dist:
    n, p = as_expr(n), as_expr(p)
    _check(_notprob, p, "Binomial", "probability", ErrorCode.InvalidProbability)
    _check(lambda v: v < 0 or v != math.floor(v), n, "Binomial", "n", ErrorCode.InvalidCount)
    return Dist("Binomial", [n, p])


def Poisson(lam) -> Dist:
    lam = as_expr(lam)
    _check(_nonpos, lam, "Poisson", "rate", ErrorCode.InvalidRate)
    return Dist("Poisson", [lam])


def StudentT(df, loc, scale) -> Dist:
    df, loc, scale = as_expr(df), as_expr(loc), as_expr(scale)
    _check(_nonpos, df, "StudentT", "df", ErrorCode.InvalidShape)
    _check(_nonfinite, loc, "StudentT", "loc", ErrorCode.InvalidMean)
    _check(_nonpos, scale, "StudentT", "scale", ErrorCode.InvalidVariance)
    return Dist("StudentT", [df, loc, scale])


def Cauchy(loc, scale) -> Dist:
    loc, scale = as_expr(loc), as_expr(scale)
    _check(_nonfinite, loc, "Cauchy", "loc", ErrorCode.InvalidMean)
    _check(_nonpos, scale, "Cauchy", "scale", ErrorCode.InvalidVariance)
    return Dist("Cauchy", [loc, scale])


def Laplace(loc, scale) -> Dist:
    loc, scale = as_expr(loc), as_expr(scale)
    _check(_nonfinite, loc, "Laplace", "loc", ErrorCode.InvalidMean)
    _check(_nonpos, scale, "Laplace", "scale", ErrorCode.InvalidVariance)
    return Dist("Laplace", [loc, scale])


def Weibull(shape, scale) -> Dist:
    shape, scale = as_expr(shape), as_expr(scale)
    _check(_nonpos, shape, "Weibull", "shape", ErrorCode.InvalidShape)
    _check(_nonpos, scale, "Weibull", "scale", ErrorCode.InvalidVariance)
    return Dist("Weibull", [shape, scale])


def ChiSquared(k) -> Dist:
    k = as_expr(k)
    _check(_nonpos, k, "ChiSquared", "k", ErrorCode.InvalidShape)
    return Dist("ChiSquared", [k])


def InverseGamma(shape, rate) -> Dist:
    shape, rate = as_expr(shape), as_expr(rate)
    _check(_nonpos, shape, "InverseGamma", "shape", ErrorCode.InvalidShape)
    _check(_nonpos, rate, "InverseGamma", "rate", ErrorCode.InvalidRate)
    return Dist("InverseGamma", [shape, rate])


def DiscreteUniform(low, high) -> Dist:
    exact = (int(low), int(high)) if all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in (low, high)) else None
    low, high = as_expr(low), as_expr(high)
    if exact is not None and exact[1] < exact[0]:
        raise FugueError("DiscreteUniform: high < low", ErrorCode.InvalidRange)
    if _cval(low) is not None and _cval(high) is not None and _cval(high) < _cval(low):
        raise FugueError("DiscreteUniform: high < low", ErrorCode.InvalidRange)
    return Dist("DiscreteUniform", [low, high], exact)


# --------------------------------------------------------------------------------------
# Program = flattened description; Model = monadic surface traced into it
# --------------------------------------------------------------------------------------
SAMPLE, OBSERVE, FACTOR = 0, 1, 2


@dataclass
class Stmt:
    kind: int
    dist: Optional[Dist]
    addr: Optional[str]
    value: Optional[Expr]      # observed value / factor log-weight
    handle: int = -1           # program-order sample index


@dataclass
class Program:
    stmts: List[Stmt] = field(default_factory=list)
    data: List[np.ndarray] = field(default_factory=list)
    data_names: List[str] = field(default_factory=list)
    result: object = None
    n_samples: int = 0

    # imperative builder (the `prob!` do-notation flattened) -----------------------------
    def bind_data(self, name: str, values) -> DataArray:
        arr = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel())
        self.data.append(arr)
        self.data_names.append(name)
        return DataArray(self, len(self.data) - 1, arr)

    def sample(self, address: str, dist: Dist) -> Expr:
        h = self.n_samples
        self.n_samples += 1
        self.stmts.append(Stmt(SAMPLE, dist, address, None, h))
        return Expr("site", a=h)

    def observe(self, address: str, dist: Dist, value: Number) -> None:
        self.stmts.append(Stmt(OBSERVE, dist, address, as_expr(value)))

    def factor(self, logw: Number) -> None:
        self.stmts.append(Stmt(FACTOR, None, None, as_expr(logw)))

    # derived ---------------------------------------------------------------------------
    def sample_addresses(self) -> List[str]:
        return [s.addr for s in self.stmts if s.kind == SAMPLE]

    def sorted_sites(self) -> List[str]:
        """Site order = `BTreeMap<Address,_>` order = byte-wise lexicographic
        (/root/reference/src/core/address.rs:150-157): `"x#10" < "x#2"`."""
        return sorted(self.sample_addresses(), key=lambda s: s.encode("utf-8"))


class Model:
    """`Model<A>` (/root/reference/src/core/model.rs:20-131) with symbolic continuations."""

    def __init__(self, tag: str, payload=None, k: Optional[Callable] = None):
        self.tag, self.payload, self.k = tag, payload, k

    # ModelExt (model.rs:447-581)
    def bind(self, f: Callable[[object], "Model"]) -> "Model":
        if self.tag == "pure":
            return Model("thunk", None, lambda _: f(self.payload))
        k0 = self.k
        return Model(self.tag, self.payload, lambda v: (k0(v) if k0 else pure(v)).bind(f))

    def map(self, f: Callable[[object], object]) -> "Model":
        return self.bind(lambda v: pure(f(v)))

    and_then = bind


def pure(value) -> Model:
    return Model("pure", value)


def sample(address: str, dist: Dist) -> Model:
    return Model("sample", (address, dist), None)


def observe(address: str, dist: Dist, value: Number) -> Model:
    return Model("observe", (address, dist, value), None)


def factor(logw: Number) -> Model:
    return Model("factor", logw, None)


def guard(pred) -> Model:
    """model.rs:710-716: `pure(())` when the predicate holds, `factor(-inf)` otherwise.  A predicate on sampled values becomes a
    weight the site program can express: strict (`guard(phi.abs() < 0.95)`): ln(clamp(margin * 1e300, 0, 1)) = 0 when margin >=
    1e-300, -inf when margin <= 0; non-strict (`guard(k >= 1)`): ln(clamp(margin * 1e300 + 1, 0, 1)) = 0 when margin >= 0, -inf
    when margin <= -1e-300 -- equality is accepted exactly as the reference's bool does.  (Margins of magnitude below 1e-300
    other than 0 get a finite weight; no f64 computation of a difference of O(1) quantities produces one.)"""
    if isinstance(pred, Cond):
        m = pred.margin
        if m.is_const():
            ok = m.value > 0.0 if pred.strict else m.value >= 0.0
            return pure(None) if ok else factor(float("-inf"))
        if pred.strict:
            return factor((m * 1e300).clamp(0.0, 1.0).ln())
        return factor((m * 1e300 + 1.0).clamp(0.0, 1.0).ln())
    return pure(None) if pred else factor(float("-inf"))


def zip_models(a: Model, b: Model) -> Model:
    return a.bind(lambda x: b.map(lambda y: (x, y)))


def sequence_vec(models: Sequence[Model]) -> Model:
    """model.rs:623-658"""
    def go(i, acc):
        if i == len(models):
            return pure(list(acc))
        return models[i].bind(lambda v: go(i + 1, acc + [v]))
    return go(0, [])


def traverse_vec(items: Sequence, f: Callable[[object], Model]) -> Model:
    return sequence_vec([f(x) for x in items])


def plate(items: Sequence, f: Callable[[object], Model]) -> Model:
    """`plate!(i in range => body)` (/root/reference/src/macros/mod.rs:72-90)"""
    return traverse_vec(list(items), f)


def trace_model(model_or_fn, program: Optional[Program] = None) -> Program:
    """Run the model once with symbolic values: the analogue of
    `run(handler, model)` (/root/reference/src/runtime/handler.rs:124-209) where the
    "handler" records the effects instead of interpreting them."""
    prog = program if program is not None else Program()
    m = model_or_fn(prog) if callable(model_or_fn) and _wants_program(model_or_fn) else (
        model_or_fn() if callable(model_or_fn) else model_or_fn)
    if isinstance(m, Program):
        return m
    while True:                                   # iterative trampoline (stack safe)
        if m.tag == "pure":
            prog.result = m.payload
            return prog
        if m.tag == "thunk":
            m = m.k(None)
            continue
        if m.tag == "sample":
            address, dist = m.payload
            v = prog.sample(address, dist)
        elif m.tag == "observe":
            address, dist, value = m.payload
            prog.observe(address, dist, value)
            v = None
        else:
            prog.factor(m.payload)
            v = None
        m = m.k(v) if m.k else pure(v)


def _wants_program(fn) -> bool:
    import inspect
    try:
        return len(inspect.signature(fn).parameters) == 1
    except (TypeError, ValueError):
        return False
