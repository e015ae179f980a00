"""The line bench.py prints must fit the driver's record: contract keys first, `roofline` + `cpu_baseline`, `legs` LAST, <= 6 KB."""
import glob
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _bench():
    spec = importlib.util.spec_from_file_location("fg_bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compact_line_of_every_committed_verbose_document():
    bench = _bench()
    docs = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_bench_full.json")))
    assert docs
    for path in docs[-1:]:
        full = json.load(open(path))
        line = bench.compact(full)
        text = json.dumps(line, separators=(",", ":"))
        assert len(text) <= 6144, (path, len(text))
        keys = list(line)
        assert keys[: len(CONTRACT)] == list(CONTRACT)
        assert keys[-1] == "legs" and "roofline" in line and "workload" in line["config"] and "model" not in line["config"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in line["roofline"]
        if "cpu_baseline" in full:
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in line["cpu_baseline"]
        for leg in ("mh", "c5", "smc", "c3_65536", "c3_8192", "hmc_fd_dense"):
            e = line["legs"][leg]
            assert e["value"] > 0 and 0 < e["frac"] < 1.5 and e["kernel"]
        assert line["legs"]["mh"]["cpu"]["value"] > 0 and line["legs"]["smc"]["cpu"]["value"] > 0
        assert json.loads(text) == json.loads(json.dumps(line))


def test_default_arguments_fit_the_contract():
    bench = _bench()
    a = bench.parse([])
    assert a.gpus == 1 and a.steps > 0 and a.warmup >= 0 and not a.full
