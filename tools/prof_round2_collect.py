"""Turns the raw rocprofv3 output of tools/prof_round2.sh into the small files committed under profiles/round2_*.
Writes them to <out>/profiles/ (gpurun merges gpurun_out/ back; copy them into profiles/ from there)."""
import collections, csv, glob, json, os, shutil, sys

O = sys.argv[1]
P = os.path.join(O, "profiles")
os.makedirs(P, exist_ok=True)


def first(pattern):
    g = sorted(glob.glob(os.path.join(O, pattern), recursive=True))
    return g[0] if g else None


def counters(d):
    """kernel base name -> counter -> values in dispatch order"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        for r in rows:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def pick(acc, key):
    out = {}
    for k, cs in acc.items():
        if key in k:
            for c, v in cs.items():
                out.setdefault(c, []).extend(v)
    return out


KIB = 1024.0
doc = {
    "_about": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (tools/prof_round2.sh), MI355X, round 2. "
              "Counter unit: KiB (x1024 = bytes). gfx950 (MI355X_MICROARCH.md, HBM section): FETCH_SIZE under-counts wide 16 B/lane "
              "streaming reads by 2x and WRITE_SIZE is exact for 16 B/lane stores; these kernels move 8 B/lane rows (uncalibrated "
              "width), so the values are reported raw next to the bytes the kernel is known to write (draw rows), which calibrates them.",
}
# HMC headline: bench.py --steps 100 --warmup 50 --launch 25 -> 2 warmup launches + 4 sampling launches of k_hmc_sep_steps
hf, hw = pick(counters("hmc_fetch"), "k_hmc_sep_steps<false, 0>"), pick(counters("hmc_write"), "k_hmc_sep_steps<false, 0>")
f, w = hf.get("FETCH_SIZE", []), hw.get("WRITE_SIZE", [])
if f and w:
    fs, ws = f[-4:], w[-4:]
    fb, wb = sum(fs) / len(fs) * KIB, sum(ws) / len(ws) * KIB
    doc["config"] = {"chains": 65536, "n_sites": 32, "grad": "fd_sparse", "transitions_per_launch": 25,
                     "launch_kind": "sampling (draw rows written)", "kernel": "k_hmc_sep_steps<0>"}
    doc["FETCH_SIZE_KiB_per_launch"], doc["WRITE_SIZE_KiB_per_launch"] = f, w
    doc["sampling_launch_bytes"] = {"fetch": fb, "write": wb, "total": fb + wb}
    doc["expected_bytes"] = {
        "draw_rows_written": 25 * 32 * 65536 * 8,
        "state_rows_loaded_and_stored_once_per_launch": "values + step size + dual-averaging rows: ~(32 + 8) x 65536 x 8 B each way",
        "algorithmic_SURVEY_8d": 25 * 16 * 65536 * 32 * 32,
        "note": "q, p and the gradient live in registers / LDS for the whole launch; HBM sees the draw rows and one state round trip",
    }
# MH: tools/bench_mh.py -> reference_model(20): launches of k_mh_mw_steps (100 warm-up + 100 + 400 steps), then C5
mf, mw_ = pick(counters("mh_fetch"), "k_mh_mw_steps"), pick(counters("mh_write"), "k_mh_mw_steps")
if mf and mw_:
    doc["mh"] = {"kernel": "k_mh_mw_steps", "command": "python3 tools/bench_mh.py (reference_model(20) at 65536 chains, then C5 mixture(32) at 262144 chains)",
                 "FETCH_SIZE_KiB_per_launch": mf.get("FETCH_SIZE", []), "WRITE_SIZE_KiB_per_launch": mw_.get("WRITE_SIZE", [])}
    f_, w_ = mf.get("FETCH_SIZE", []), mw_.get("WRITE_SIZE", [])
    if len(f_) >= 2 and len(w_) >= 2:       # launches 0, 1 = reference_model(20): 100 adapting steps, then 400 sampling steps, 65 536 chains
        doc["mh"]["reference_model20_bytes_per_chain_step"] = {
            "adapting": (f_[0] + w_[0]) * KIB / (65536 * 100), "sampling": (f_[1] + w_[1]) * KIB / (65536 * 400),
            "note": "FETCH_SIZE + WRITE_SIZE of the launch / (chains x steps); read by bench.py for mh.roofline.traffic"}
# SMC: tools/bench_smc.py -> 4 runs of fg_smc_run at 1 048 576 particles; total over every kernel / 4
for name, d in (("fetch", "smc_fetch"), ("write", "smc_write")):
    acc = counters(d)
    tot, per = 0.0, {}
    for k, cs in acc.items():
        for c, v in cs.items():
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                tot += sum(v)
                per[k[:60]] = per.get(k[:60], 0.0) + sum(v)
    doc.setdefault("smc", {"command": "python3 tools/bench_smc.py (4 x adaptive_smc, 1 048 576 particles, R=3)", "runs": 4})
    doc["smc"][name + "_bytes_per_run"] = tot * KIB / 4
    doc["smc"][name + "_KiB_by_kernel_4_runs"] = dict(sorted(per.items(), key=lambda kv: -kv[1])[:12])
json.dump(doc, open(os.path.join(P, "round2_hbm_traffic.json"), "w"), indent=1)

# instruction counts of the headline kernel per launch (bench.py --steps 50 --warmup 25 --launch 25: the last two launches sample)
mix = {}
for d in ("hmc_pmc1", "hmc_pmc2", "hmc_pmc3"):
    for c, v in pick(counters(d), "k_hmc_sep_steps<false, 0>").items():
        mix[c] = sum(v[-2:]) / max(1, len(v[-2:]))
if mix:
    json.dump({"_about": "rocprofv3 --pmc passes over k_hmc_sep_steps (tools/prof_round2.sh), wave-instruction counts per 25-transition sampling launch "
                         "at 65 536 chains, fd_sparse; read by bench.py for roofline.executed",
               "config": {"chains": 65536, "grad": "fd_sparse", "transitions_per_launch": 25}, "per_launch": mix},
              open(os.path.join(P, "round2_hmc_pmc.json"), "w"), indent=1)

# kernel stats of the default bench command + the bench line itself
ks = first("bench/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(P, "round2_bench_kernel_stats.csv"))
try:
    line = [l for l in open(os.path.join(O, "bench.json")).read().splitlines() if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(P, "round2_bench_full.json"), "w"), indent=1)
    j = json.loads(line)
    print("bench under the profiler: value %.4g, avg_launch_ms %.4f" % (j["value"], j["roofline"]["avg_launch_ms"]))
except Exception as e:                                            # noqa: BLE001
    print("no bench line:", e)
# the stats average above mixes adapting launches, the clock spin-up's scratch launches, the 40 timed launches and the validity leg's
# two 200-transition launches: list every dispatch of the headline kernel in order and average the launches of the timed region
# (order in `python3 bench.py`: 8 adapting, spin-up, 40 timed, ..., 2 validity = the last two)
kt = first("bench/**/*kernel_trace.csv")
if kt:
    rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if "k_hmc_sep_steps<false, 0>" in r["Kernel_Name"]]   # not the dense variant of the extras leg
    if len(d) >= 50:
        adapt, spin, timed, valid = d[:8], d[8:-42], d[-42:-2], d[-2:]
        with open(os.path.join(P, "round2_hmc_timed_region.txt"), "w") as fo:
            fo.write("k_hmc_sep_steps dispatches of `python3 bench.py` under rocprofv3 --kernel-trace, in order, ms:\n")
            fo.write("  8 adapting launches (warmup 200 / 25): %s\n" % " ".join("%.3f" % x for x in adapt))
            fo.write("  clock spin-up on the scratch engine (%d launches of 100 transitions, untimed): mean %.3f\n" % (len(spin), sum(spin) / max(1, len(spin))))
            fo.write("  40 timed launches (steps 1000 / 25):   %s\n" % " ".join("%.3f" % x for x in timed))
            fo.write("  validity leg (200 + 200 in two launches): %s\n" % " ".join("%.3f" % x for x in valid))
            fo.write("average of the 40 timed launches: %.4f ms  (bench.py's HIP-event avg_launch_ms in the same run: see round2_bench_full.json)\n" % (sum(timed) / 40))
ks = first("smc/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(P, "round2_smc_kernel_stats.csv"))
for leg in ("hmc", "mh"):
    src = os.path.join(O, leg + "_instruction_mix.txt")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "round2_%s_pmc_instruction_mix.txt" % leg))
for f in sorted(os.listdir(P)):
    print(f, os.path.getsize(os.path.join(P, f)))
if os.path.exists(os.path.join(P, "round2_bench_kernel_stats.csv")):
    print(open(os.path.join(P, "round2_bench_kernel_stats.csv")).read()[:1500])
print(json.dumps({k: doc[k] for k in ("sampling_launch_bytes",) if k in doc}))
