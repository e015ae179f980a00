"""The model-language front-end (fg_dsl_compile) on the CPU: it must build the same site program as
the hand-written mirror model, produce `addr!`-identical addresses, and report static errors the way
the reference's playground does (crates/fugue-wasm/src/dsl.rs:1149-1328 are the tests restated here;
the inference halves of those tests are in tests/test_gpu_dsl.py)."""
import pytest

from fugue_amd import engine as E
from fugue_amd import model as M
from tests.dsl_models import PAIRS


@pytest.mark.parametrize("name", list(PAIRS))
def test_dsl_builds_the_same_site_program_as_the_mirror(name):
    src, data, mirror = PAIRS[name]
    cp = E.CompiledProgram.from_dsl(src, data)
    ref = E.compile_model(mirror())
    assert cp.site_names == ref.site_names and cp.site_vtypes == ref.site_vtypes
    assert (cp.S, cp.d, cp.O) == (ref.S, ref.d, ref.O)
    assert (cp.n_instructions, cp.n_slots, cp.dep_counts) == (ref.n_instructions, ref.n_slots, ref.dep_counts)
    assert cp.warnings == []


def test_coin_addresses_match_addr_macro():
    cp = E.CompiledProgram.from_dsl(PAIRS["coin"][0], PAIRS["coin"][1])
    assert cp.site_names == [M.addr("p")] and cp.O == 10        # observations are scored, not recorded as choices


def test_indexed_sample_addresses_match_addr_macro():
    cp = E.CompiledProgram.from_dsl(PAIRS["indexed"][0])
    assert cp.site_names == [M.addr("z", 0), M.addr("z", 1), M.addr("z", 2)]


def test_static_errors_are_caught():
    for src in ["pure(nope)",                                                    # unknown variable
                'let x <- sample(addr!("x"), Nope(1.0)); pure(x)',               # unknown distribution
                'let x <- sample(addr!("x"), Normal(1.0)); pure(x)',             # arity
                "let x = 1.0;",                                                  # no pure
                'let x <- sample(addr!("x"), Normal(0, 1)); pure(frob(x))',      # unknown function
                'let x <- sample(addr!("x"), Normal(0, 1)); pure(pow(x))',       # function arity
                'let x <- sample(addr!("x"), Normal(0, 1)); pure(x) let y = 1;',  # trailing input after pure
                'let x <- sample(addr!("x"), Normal::make(0, 1)); pure(x)',      # only ::new is sugar
                'let s = "abc; pure(0)']:                                        # unterminated string
        with pytest.raises(E.DslError):
            E.CompiledProgram.from_dsl(src)
    with pytest.raises(E.DslError) as ei:                                        # missing comma must not parse
        E.CompiledProgram.from_dsl('let a = 1;\nlet x <- sample(addr!("x") Normal(0,1)); pure(x)')
    assert "line 2" in str(ei.value) and "expected `,`" in str(ei.value) and "found `Normal`" in str(ei.value)
    with pytest.raises(E.DslError) as ei:
        E.CompiledProgram.from_dsl("let x = 1.0;")
    assert "model must end with `pure(<expr>)`" in str(ei.value)
    with pytest.raises(E.DslError) as ei:
        E.CompiledProgram.from_dsl('let x <- sample(addr!("x"), Categorical()); pure(x)')
    assert "1..=64" in str(ei.value)


def test_data_binding_rules():
    """dsl.rs:1066-1106: object of arrays, bare array -> `data`, booleans -> 0/1, anything else rejected."""
    src = 'let m <- sample(addr!("m"), Normal(0, 1)); for i in 0..data.len() { observe(addr!("y", i), Normal(m, 1.0), data[i]); } pure(m)'
    assert E.CompiledProgram.from_dsl(src, "[true, false, 2.5]").O == 3
    assert E.CompiledProgram.from_dsl(src, {"data": [1, 2]}).O == 2
    assert E.CompiledProgram.from_dsl(src, '  {"data": [1e0, -2.5E-1, 3], "other": []}  ').O == 3
    with pytest.raises(E.DslError):
        E.CompiledProgram.from_dsl(src, '{"data": 3}')
    with pytest.raises(E.DslError):
        E.CompiledProgram.from_dsl(src, '{"data": ["a"]}')
    with pytest.raises(E.DslError):
        E.CompiledProgram.from_dsl(src, '"just a string"')
    with pytest.raises(E.DslError):                      # `data` unbound without data
        E.CompiledProgram.from_dsl(src, None)


def test_invalid_params_kill_weight_not_process():
    """sigma = -1 is impossible: the model still builds, with a Normal(0,1) placeholder site, a -inf
    factor and a warning (dsl.rs:961-977); an impossible observation likewise (:1002-1006)."""
    cp = E.CompiledProgram.from_dsl('let mu <- sample(addr!("mu"), Normal(0.0, -1.0)); pure(mu)')
    assert cp.site_names == ["mu"] and cp.site_vtypes == [0]
    assert len(cp.warnings) == 1 and cp.warnings[0].startswith("sample `mu`:")
    cp = E.CompiledProgram.from_dsl('let k <- sample(addr!("k"), Binomial(2.5, 0.5)); observe(addr!("o"), Gamma(-1.0, 1.0), 2.0); pure(k)')
    assert cp.site_vtypes == [0] and cp.O == 0           # placeholder is an f64 site; the observe became a factor
    assert "Binomial n must be a non-negative integer, got 2.5" in cp.warnings[0]
    assert cp.warnings[1].startswith("observe at `o`:")


def test_out_of_bounds_index_warns_and_yields_nan():
    cp = E.CompiledProgram.from_dsl('let m <- sample(addr!("m"), Normal(0, 1)); observe(addr!("y"), Normal(m, 1.0), data[5]); pure(m)', "[1, 2]")
    assert cp.warnings == ["index 5 out of bounds (len 2)"]


def test_integer_arithmetic_and_scoping():
    """Int (+,-,*) Int stays Int so it can index and address (dsl.rs:752-759); `/` goes to f64; a loop
    variable shadows and then restores an outer binding; `0..n` lexes as a range, `0.5` as a number."""
    src = '''
        let i = 7;
        let n = 2 * 3 - 4;            // 2
        for i in 0..n { let v <- sample(addr!("v", i * 2 + 1), Normal(0.5, 1.0)); }
        let w <- sample(addr!("w", i), Normal(6 / 4, 1.0));   // i is 7 again; 6/4 = 1.5
        pure(w)
    '''
    cp = E.CompiledProgram.from_dsl(src)
    assert cp.site_names == ["v#1", "v#3", "w#7"]
    with pytest.raises(E.DslError):                      # loop variable is gone after the loop
        E.CompiledProgram.from_dsl('for j in 0..2 { let v <- sample(addr!("v", j), Normal(0, 1)); } pure(j)')
    with pytest.raises(E.DslError):                      # a sampled value cannot drive a build-time loop
        E.CompiledProgram.from_dsl('let k <- sample(addr!("k"), Poisson(3.0)); for j in 0..k { factor(0.0); } pure(k)')


def test_duplicate_address_is_reported():
    with pytest.raises(E.DslError) as ei:
        E.CompiledProgram.from_dsl('for i in 0..2 { let v <- sample(addr!("v"), Normal(0, 1)); } pure(0)')
    assert "sampled twice" in str(ei.value)


def test_build_time_limits():
    """The model is unrolled when it is built, so runaway sizes are refused instead of exhausting memory."""
    with pytest.raises(E.DslError) as ei:
        E.CompiledProgram.from_dsl('for i in 0..2000000 { factor(0.0); } pure(0)')
    assert "1048576 statements" in str(ei.value)
    with pytest.raises(E.DslError) as ei:
        E.CompiledProgram.from_dsl("pure(" + "(" * 500 + "1" + ")" * 500 + ")")
    assert "nesting" in str(ei.value)
    with pytest.raises(E.DslError):                      # an empty loop with a huge trip count is bounded too
        E.CompiledProgram.from_dsl('for i in 0..9000000000 { } pure(0)')
