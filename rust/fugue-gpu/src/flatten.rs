//! Flattening a `Fn() -> Model<A>` into a fixed-structure site program (`fg_program`).
//!
//! A `Model<A>` is an opaque chain of `FnOnce` continuations (`src/core/model.rs:20-131`): its structure can only be
//! observed by RUNNING it.  `flatten` runs it through a recording `Handler` (`src/runtime/handler.rs:29-96`) -- the
//! mechanism every interpreter of the reference uses -- several times:
//!
//!   1. a BASE run at values drawn from the prior (in support, so user code such as `Normal::new(mu, sigma).unwrap()`
//!      does not panic) records the statement list: address, value type, `Distribution::describe()` = (kind, concrete
//!      parameters), observed values, factor weights;
//!   2. for every sample site one site value is PERTURBED (twice, the second time by twice as much) and the run repeated:
//!      a parameter that changes depends on that site; two probes give the slope, the third checks that the dependence is
//!      affine.  Categorical (`usize`) sites are walked over all their categories, which recovers `options[z]` selects.
//!
//! What comes out is, per parameter, `c0 + sum_j b_j * site_j`, a bare site, a constant, or `select(z, options)` --
//! exactly the expression forms the engine compiles into stream records.  Anything else (a non-affine dependence, an
//! address set or order that changes between runs) is refused with `ErrorCode::UnexpectedModelStructure`, as the reference's
//! replay handlers refuse structure changes (`src/runtime/interpreters.rs:23-33`); such models can be given to the engine
//! through the DSL front-end (`fg_dsl_compile`) or written against the C ABI directly.
//! The recovered coefficients carry the rounding of a finite difference of an exactly affine function (relative 1e-13).
//!
//! UNVERIFIED SOURCE -- never compiled (no Rust toolchain in the build image).
use std::collections::BTreeMap;
use std::ffi::CString;

use fugue::runtime::handler::{run, Handler};
use fugue::*;
use rand::rngs::StdRng;
use rand::SeedableRng;

use crate::ffi::*;

/// Concrete value of one site in one probing run.
#[derive(Clone, Copy, Debug, PartialEq)]
pub enum Cell { F64(f64), Bool(bool), U64(u64), Usize(usize), I64(i64) }
impl Cell {
    fn as_f64(self) -> f64 {
        match self { Cell::F64(x) => x, Cell::Bool(b) => b as u8 as f64, Cell::U64(k) => k as f64, Cell::Usize(k) => k as f64, Cell::I64(k) => k as f64 }
    }
    fn vtype(self) -> i32 { match self { Cell::F64(_) => FG_F64, Cell::Bool(_) => FG_BOOL, Cell::U64(_) => FG_U64, Cell::Usize(_) => FG_USIZE, Cell::I64(_) => FG_I64 } }
}

/// One recorded effect of a run.
#[derive(Clone, Debug)]
enum Stmt {
    Sample { addr: Address, desc: DistDesc, value: Cell },
    Observe { addr: Address, desc: DistDesc, value: Cell },
    Factor { logw: f64 },
}
impl Stmt {
    fn same_shape(&self, o: &Stmt) -> bool {
        match (self, o) {
            (Stmt::Sample { addr: a, desc: d, value: v }, Stmt::Sample { addr: b, desc: e, value: w }) => a == b && d.kind == e.kind && d.params.len() == e.params.len() && v.vtype() == w.vtype(),
            (Stmt::Observe { addr: a, desc: d, value: v }, Stmt::Observe { addr: b, desc: e, value: w }) => a == b && d.kind == e.kind && d.params.len() == e.params.len() && v.vtype() == w.vtype(),
            (Stmt::Factor { .. }, Stmt::Factor { .. }) => true,
            _ => false,
        }
    }
    /// the numbers of the statement that may depend on earlier sites: parameters, then the observed value / factor weight
    fn numbers(&self) -> Vec<f64> {
        match self {
            Stmt::Sample { desc, .. } => desc.params.clone(),
            Stmt::Observe { desc, value, .. } => { let mut v = desc.params.clone(); v.push(value.as_f64()); v }
            Stmt::Factor { logw } => vec![*logw],
        }
    }
}

/// The recording handler: replays `assign` for the sites it names, draws the others from their prior.
struct ProbeHandler<'a> {
    rng: &'a mut StdRng,
    assign: &'a BTreeMap<Address, Cell>,
    out: Vec<Stmt>,
    err: Option<FugueError>,
}
fn not_flattenable(addr: &Address) -> FugueError {
    FugueError::ModelError { address: Some(addr.clone()), reason: "distribution does not describe() itself: only the built-in distributions can be flattened".into(),
                             code: ErrorCode::UnexpectedModelStructure, context: Default::default() }
}
macro_rules! probe_sample {
    ($name:ident, $t:ty, $variant:ident) => {
        fn $name(&mut self, addr: &Address, dist: &dyn Distribution<$t>) -> $t {
            let v: $t = match self.assign.get(addr) { Some(Cell::$variant(x)) => *x, _ => dist.sample(self.rng) };
            match dist.describe() {
                Some(desc) => self.out.push(Stmt::Sample { addr: addr.clone(), desc, value: Cell::$variant(v) }),
                None => if self.err.is_none() { self.err = Some(not_flattenable(addr)) },
            }
            v
        }
    };
}
macro_rules! probe_observe {
    ($name:ident, $t:ty, $variant:ident) => {
        fn $name(&mut self, addr: &Address, dist: &dyn Distribution<$t>, value: $t) {
            match dist.describe() {
                Some(desc) => self.out.push(Stmt::Observe { addr: addr.clone(), desc, value: Cell::$variant(value) }),
                None => if self.err.is_none() { self.err = Some(not_flattenable(addr)) },
            }
        }
    };
}
impl<'a> Handler for ProbeHandler<'a> {
    probe_sample!(on_sample_f64, f64, F64);
    probe_sample!(on_sample_bool, bool, Bool);
    probe_sample!(on_sample_u64, u64, U64);
    probe_sample!(on_sample_usize, usize, Usize);
    probe_sample!(on_sample_i64, i64, I64);
    probe_observe!(on_observe_f64, f64, F64);
    probe_observe!(on_observe_bool, bool, Bool);
    probe_observe!(on_observe_u64, u64, U64);
    probe_observe!(on_observe_usize, usize, Usize);
    probe_observe!(on_observe_i64, i64, I64);
    fn on_factor(&mut self, logw: f64) { self.out.push(Stmt::Factor { logw }); }
    fn finish(self) -> Trace { Trace::default() }
}

fn probe<A>(model_fn: &impl Fn() -> Model<A>, rng: &mut StdRng, assign: &BTreeMap<Address, Cell>) -> FugueResult<Vec<Stmt>> {
    let mut h = ProbeHandler { rng, assign, out: Vec::new(), err: None };
    // `run` consumes the handler by value and returns only the trace: record through a raw pointer to keep the log
    let hp: *mut ProbeHandler = &mut h;
    struct Fwd(*mut ProbeHandler<'static>);
    // SAFETY: `h` outlives the call to `run`; Fwd only forwards to it.
    impl Handler for Fwd {
        fn on_sample_f64(&mut self, a: &Address, d: &dyn Distribution<f64>) -> f64 { unsafe { (*self.0).on_sample_f64(a, d) } }
        fn on_sample_bool(&mut self, a: &Address, d: &dyn Distribution<bool>) -> bool { unsafe { (*self.0).on_sample_bool(a, d) } }
        fn on_sample_u64(&mut self, a: &Address, d: &dyn Distribution<u64>) -> u64 { unsafe { (*self.0).on_sample_u64(a, d) } }
        fn on_sample_usize(&mut self, a: &Address, d: &dyn Distribution<usize>) -> usize { unsafe { (*self.0).on_sample_usize(a, d) } }
        fn on_sample_i64(&mut self, a: &Address, d: &dyn Distribution<i64>) -> i64 { unsafe { (*self.0).on_sample_i64(a, d) } }
        fn on_observe_f64(&mut self, a: &Address, d: &dyn Distribution<f64>, v: f64) { unsafe { (*self.0).on_observe_f64(a, d, v) } }
        fn on_observe_bool(&mut self, a: &Address, d: &dyn Distribution<bool>, v: bool) { unsafe { (*self.0).on_observe_bool(a, d, v) } }
        fn on_observe_u64(&mut self, a: &Address, d: &dyn Distribution<u64>, v: u64) { unsafe { (*self.0).on_observe_u64(a, d, v) } }
        fn on_observe_usize(&mut self, a: &Address, d: &dyn Distribution<usize>, v: usize) { unsafe { (*self.0).on_observe_usize(a, d, v) } }
        fn on_observe_i64(&mut self, a: &Address, d: &dyn Distribution<i64>, v: i64) { unsafe { (*self.0).on_observe_i64(a, d, v) } }
        fn on_factor(&mut self, w: f64) { unsafe { (*self.0).on_factor(w) } }
        fn finish(self) -> Trace { Trace::default() }
    }
    let _ = run(Fwd(hp as *mut ProbeHandler<'static>), model_fn());
    match h.err.take() { Some(e) => Err(e), None => Ok(h.out) }
}

/// `c0 + sum_j coef[j] * site_j`, or `select(index site, options)` where every option is itself an `Affine`.
#[derive(Clone, Debug)]
pub enum ParamExpr {
    Affine { c0: f64, terms: Vec<(usize /* sample handle */, f64)> },
    Select { index: usize, options: Vec<ParamExpr> },
}

/// The flattened program and what is needed to rebuild `(A, Trace)` results from engine output.
pub struct FlatProgram {
    pub raw: *mut fg_program,
    /// addresses of the sample sites in PROGRAM order (handle = position)
    pub sites: Vec<Address>,
    /// value type of each site (FG_F64 ...)
    pub vtypes: Vec<i32>,
    /// sorted site index (engine row) of each handle
    pub row_of_handle: Vec<usize>,
}
impl Drop for FlatProgram { fn drop(&mut self) { unsafe { fg_program_free(self.raw) } } }

fn structure_error(what: String) -> FugueError {
    FugueError::ModelError { address: None, reason: what, code: ErrorCode::UnexpectedModelStructure, context: Default::default() }
}

fn engine_error(rc: i32) -> FugueError {
    let code = match rc { 301 => ErrorCode::AddressConflict, 302 => ErrorCode::UnexpectedModelStructure, 102 => ErrorCode::InvalidProbability,
                          106 => ErrorCode::InvalidCount, 500 => ErrorCode::TraceAddressNotFound, 600 => ErrorCode::TypeMismatch, _ => ErrorCode::UnexpectedModelStructure };
    FugueError::ModelError { address: None, reason: format!("fugue_amd error {rc}: {}", last_error()), code, context: Default::default() }
}

fn emit(expr: &ParamExpr, out: &mut Vec<fg_tok>) {
    let tok = |op: i32, a: i32, imm: f64| fg_tok { op, a, b: 0, reserved: 0, imm };
    match expr {
        ParamExpr::Affine { c0, terms } => {
            // a bare site stays a bare site (the engine's fast records want leaves); otherwise c0 + t_0 + t_1 + ... left to right
            if terms.len() == 1 && *c0 == 0.0 && terms[0].1 == 1.0 { out.push(tok(FG_T_SITE, terms[0].0 as i32, 0.0)); return; }
            out.push(tok(FG_T_CONST, 0, *c0));
            for (h, b) in terms {
                out.push(tok(FG_T_SITE, *h as i32, 0.0));
                out.push(tok(FG_T_CONST, 0, *b));
                out.push(tok(FG_T_MUL, 0, 0.0));
                out.push(tok(FG_T_ADD, 0, 0.0));
            }
        }
        ParamExpr::Select { index, options } => {
            out.push(tok(FG_T_SITE, *index as i32, 0.0));
            for o in options { emit(o, out); }
            out.push(tok(FG_T_SELECT, options.len() as i32, 0.0));
        }
    }
}

/// Flatten `model_fn` (see the module documentation).  `seed` fixes the base prior draw.
pub fn flatten<A>(model_fn: &impl Fn() -> Model<A>, seed: u64) -> FugueResult<FlatProgram> {
    let mut rng = StdRng::seed_from_u64(seed);
    let empty = BTreeMap::new();
    let base = probe(model_fn, &mut rng, &empty)?;
    // base assignment: every sample site at its base value, handles in program order
    let mut assign: BTreeMap<Address, Cell> = BTreeMap::new();
    let mut sites: Vec<Address> = Vec::new();
    let mut cells: Vec<Cell> = Vec::new();
    for s in &base {
        if let Stmt::Sample { addr, value, .. } = s {
            if assign.insert(addr.clone(), *value).is_some() {
                return Err(FugueError::ModelError { address: Some(addr.clone()), reason: "address sampled twice".into(), code: ErrorCode::AddressConflict, context: Default::default() });
            }
            sites.push(addr.clone()); cells.push(*value);
        }
    }
    // a replay of the base assignment must reproduce the base run (it consumes no randomness)
    let again = probe(model_fn, &mut rng, &assign)?;
    if again.len() != base.len() || again.iter().zip(&base).any(|(a, b)| !a.same_shape(b) || a.numbers() != b.numbers()) {
        return Err(structure_error("the model is not a deterministic function of its sampled values".into()));
    }
    let base_nums: Vec<Vec<f64>> = base.iter().map(|s| s.numbers()).collect();
    // expr[stmt][number]: starts as the constant of the base run
    let mut expr: Vec<Vec<ParamExpr>> = base_nums.iter().map(|v| v.iter().map(|&c| ParamExpr::Affine { c0: c, terms: vec![] }).collect()).collect();
    let rel = |a: f64, b: f64| (a - b).abs() <= 1e-9 * (1.0 + a.abs().max(b.abs()));

    for (h, addr) in sites.iter().enumerate() {
        match cells[h] {
            Cell::F64(x0) => {
                // two probes on the same side of x0 (stays inside (0, inf) / (0, 1)-type supports for small steps)
                let step = if x0 != 0.0 { x0.abs() * (1.0 / 1024.0) } else { 1.0 / 1024.0 };
                let mut runs = Vec::new();
                for k in [1.0, 2.0] {
                    let mut a = assign.clone();
                    a.insert(addr.clone(), Cell::F64(x0 + k * step));
                    let r = probe(model_fn, &mut rng, &a)?;
                    if r.len() != base.len() || r.iter().zip(&base).any(|(p, q)| !p.same_shape(q)) {
                        return Err(structure_error(format!("the set or order of statements depends on the value sampled at `{addr}`")));
                    }
                    runs.push(r.iter().map(|s| s.numbers()).collect::<Vec<_>>());
                }
                for si in 0..base.len() {
                    for pi in 0..base_nums[si].len() {
                        let (p0, p1, p2) = (base_nums[si][pi], runs[0][si][pi], runs[1][si][pi]);
                        if p1 == p0 && p2 == p0 { continue; }
                        let b = (p1 - p0) / step;
                        if !rel(p2, p0 + 2.0 * step * b) {
                            return Err(structure_error(format!("statement {si}: parameter {pi} is not an affine function of `{addr}`; give this model to the engine through the DSL front-end")));
                        }
                        // own value of a sample statement is not a parameter; `numbers()` never lists it
                        if let ParamExpr::Affine { c0, terms } = &mut expr[si][pi] { *c0 -= b * x0; terms.push((h, b)); }
                    }
                }
            }
            Cell::Usize(z0) => {
                // categories of the site: length of its own probability vector
                let k_cat = base.iter().find_map(|s| match s { Stmt::Sample { addr: a, desc, .. } if a == addr => Some(desc.params.len()), _ => None }).unwrap_or(0);
                let mut per_cat: Vec<Vec<Vec<f64>>> = Vec::new();
                for z in 0..k_cat {
                    let mut a = assign.clone();
                    a.insert(addr.clone(), Cell::Usize(z));
                    let r = probe(model_fn, &mut rng, &a)?;
                    if r.len() != base.len() || r.iter().zip(&base).any(|(p, q)| !p.same_shape(q)) {
                        return Err(structure_error(format!("the set or order of statements depends on the category sampled at `{addr}`")));
                    }
                    per_cat.push(r.iter().map(|s| s.numbers()).collect());
                }
                for si in 0..base.len() {
                    for pi in 0..base_nums[si].len() {
                        if per_cat.iter().all(|r| r[si][pi] == base_nums[si][pi]) { continue; }
                        // options[z]: each one a constant or the value of an f64 site (checked against the base assignment)
                        let mut options = Vec::new();
                        for z in 0..k_cat {
                            let v = per_cat[z][si][pi];
                            let site = cells.iter().position(|c| matches!(c, Cell::F64(x) if *x == v));
                            options.push(match site { Some(hs) => ParamExpr::Affine { c0: 0.0, terms: vec![(hs, 1.0)] }, None => ParamExpr::Affine { c0: v, terms: vec![] } });
                        }
                        let _ = z0;
                        expr[si][pi] = ParamExpr::Select { index: h, options };
                    }
                }
            }
            // bool / u64 / i64 sites as parameters of later statements: probe the neighbouring value, affine as for f64
            other => {
                let x0 = other.as_f64();
                let bumped = match other { Cell::Bool(b) => Cell::Bool(!b), Cell::U64(k) => Cell::U64(k + 1), Cell::I64(k) => Cell::I64(k + 1), c => c };
                let mut a = assign.clone();
                a.insert(addr.clone(), bumped);
                let r = probe(model_fn, &mut rng, &a)?;
                if r.len() != base.len() || r.iter().zip(&base).any(|(p, q)| !p.same_shape(q)) {
                    return Err(structure_error(format!("the set or order of statements depends on the value sampled at `{addr}`")));
                }
                let dx = bumped.as_f64() - x0;
                for si in 0..base.len() {
                    for pi in 0..base_nums[si].len() {
                        let (p0, p1) = (base_nums[si][pi], r[si].numbers()[pi]);
                        if p1 == p0 { continue; }
                        let b = (p1 - p0) / dx;
                        if let ParamExpr::Affine { c0, terms } = &mut expr[si][pi] { *c0 -= b * x0; terms.push((h, b)); }
                    }
                }
            }
        }
    }
    // a select whose options are sites must not ALSO have been given affine terms for those sites: the select wins (the
    // affine probe of an option site sees the parameter move only in the chains where z names it)
    // ---- build the fg_program
    let raw = unsafe { fg_program_new() };
    let mut prog = FlatProgram { raw, sites: sites.clone(), vtypes: cells.iter().map(|c| c.vtype()).collect(), row_of_handle: vec![] };
    for (si, s) in base.iter().enumerate() {
        let mut toks: Vec<fg_tok> = Vec::new();
        let mut lens: Vec<i32> = Vec::new();
        let n_params = match s { Stmt::Sample { desc, .. } | Stmt::Observe { desc, .. } => desc.params.len(), Stmt::Factor { .. } => 0 };
        for pi in 0..n_params { let n0 = toks.len(); emit(&expr[si][pi], &mut toks); lens.push((toks.len() - n0) as i32); }
        let rc = match s {
            Stmt::Sample { addr, desc, .. } => {
                let a = CString::new(addr.as_str()).unwrap();
                match desc.i64_bounds {
                    Some((lo, hi)) => unsafe { fg_program_sample_discrete_uniform(raw, a.as_ptr(), lo, hi) },
                    None => unsafe { fg_program_sample(raw, a.as_ptr(), desc.kind as i32, toks.as_ptr(), lens.as_ptr(), n_params as i32) },
                }
            }
            Stmt::Observe { addr, desc, .. } => {
                let a = CString::new(addr.as_str()).unwrap();
                let mut vt = Vec::new();
                emit(&expr[si][n_params], &mut vt);
                unsafe { fg_program_observe(raw, a.as_ptr(), desc.kind as i32, toks.as_ptr(), lens.as_ptr(), n_params as i32, vt.as_ptr(), vt.len() as i32) }
            }
            Stmt::Factor { .. } => { let mut vt = Vec::new(); emit(&expr[si][0], &mut vt); unsafe { fg_program_factor(raw, vt.as_ptr(), vt.len() as i32) } }
        };
        if rc < 0 || (rc > 0 && !matches!(s, Stmt::Sample { .. })) { return Err(engine_error(rc)); }
    }
    let rc = unsafe { fg_program_finalize(raw) };
    if rc != 0 { return Err(engine_error(rc)); }
    prog.row_of_handle = (0..sites.len()).map(|h| unsafe { fg_program_site_of_handle(raw, h as i32) } as usize).collect();
    Ok(prog)
}
