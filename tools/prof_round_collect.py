"""Condenses the raw rocprofv3 output of tools/prof_round.sh into small files: <out>/summary/pmc_part.json (entries keyed
like bench.py's pmc_entry keys: counters per unit, FETCH / WRITE bytes per unit, kernel-trace average duration) and a text
summary per configuration.  tools/prof_round_merge.py merges the parts of several runs into profiles/<round>_pmc.json (FG_PROF_ROUND, default round4)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_driver import CONFIGS

O = sys.argv[1]
ROUND = os.environ.get("FG_PROF_ROUND", "round4")
S = os.path.join(O, "summary")
os.makedirs(S, exist_ok=True)
KIB = 1024.0


def counters(d, last_run_from=None):
    """kernel base name -> counter -> values in dispatch order (last_run_from: only the dispatches from the LAST dispatch of that kernel on)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
        if last_run_from:
            starts = [int(r["Dispatch_Id"]) for r in rows if last_run_from in r["Kernel_Name"]]
            rows = [r for r in rows if starts and int(r["Dispatch_Id"]) >= max(starts)]
        for r in rows:
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def trace(d, last_run_from=None):
    """kernel name -> list of durations (ns) in dispatch order"""
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        if last_run_from:
            starts = [int(r["Start_Timestamp"]) for r in rows if last_run_from in r["Kernel_Name"]]
            rows = [r for r in rows if starts and int(r["Start_Timestamp"]) >= max(starts)]
        for r in rows:
            out[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


doc = {"_about": "rocprofv3 --kernel-trace [--stats | --pmc ...] passes of tools/prof_driver.py configurations on MI355X (tools/prof_round.sh); counters are "
                 "per UNIT of the entry (a transition / a chain step of every chain / a run), averaged over the kept dispatches.  FETCH_SIZE / WRITE_SIZE are in KiB "
                 "(x 1024 = bytes; MI355X_MICROARCH.md HBM section: FETCH_SIZE under-counts 16 B/lane streaming reads 2x, these kernels move 8 B/lane rows -- "
                 "reported raw)", "entries": {}}
for kd in sorted(glob.glob(os.path.join(O, "*", "key.txt"))):
    D = os.path.dirname(kd)
    key = open(kd).read().strip()
    match, units, keep, unit = CONFIGS[key]
    ent = {"unit": unit, "units_per_dispatch": units, "dispatches_kept": keep}
    lines = [f"== {key}  (unit = {unit}; {units} per dispatch)"]
    whole = "k_smc_init" if not match else None            # a whole run (SMC): the LAST run of the driver, which starts with k_smc_init
    tr = trace(os.path.join(D, "stats"), whole)
    if match:
        names = [k for k in tr if match in k]
        durs = [v for k in names for v in tr[k]][-keep:] if keep else []
        ent["kernel"] = names[0].split("(")[0] if names else None
        if durs:
            ent["kernel_trace_avg_ms_per_dispatch"] = sum(durs) / len(durs) / 1e6
            lines.append(f"kernel {ent['kernel']}: {len(durs)} dispatches kept, average {ent['kernel_trace_avg_ms_per_dispatch']:.4f} ms (min {min(durs) / 1e6:.4f}, max {max(durs) / 1e6:.4f})")
    else:                                              # whole run (SMC): every kernel and copy of the driver's last run
        tot = sum(sum(v) for v in tr.values())
        ent["kernel_time_ms_per_unit"] = tot / 1e6
        ent["launches_per_unit"] = sum(len(v) for v in tr.values())
        rows = sorted(((sum(v), len(v), k.split("(")[0]) for k, v in tr.items()), reverse=True)
        lines.append(f"kernel time of the run {ent['kernel_time_ms_per_unit']:.4f} ms over {ent['launches_per_unit']:.0f} launches (summed durations; the host-timed run is in bench.py's smc leg)")
        for s_, n_, k_ in rows[:16]:
            lines.append(f"   {k_[:60]:60s} {n_:4d} launches {s_ / 1e3:9.1f} us  avg {s_ / n_ / 1e3:7.2f} us")
    cpu = {}
    for p in ("pmc1", "pmc2", "pmc3", "fetch", "write"):
        acc = counters(os.path.join(D, p), whole)
        for kname, cs in acc.items():
            if match and match not in kname:
                continue
            for c, v in cs.items():
                if match:
                    vv = v[-keep:]
                    cpu[c] = cpu.get(c, 0.0) + sum(vv) / len(vv) / units
                else:
                    cpu[c] = cpu.get(c, 0.0) + sum(v)
    if "FETCH_SIZE" in cpu:
        ent["fetch_bytes_per_unit"] = cpu.pop("FETCH_SIZE") * KIB
    if "WRITE_SIZE" in cpu:
        ent["write_bytes_per_unit"] = cpu.pop("WRITE_SIZE") * KIB
    ent["counters_per_unit"] = cpu
    if cpu.get("SQ_INSTS_VALU"):
        f64 = cpu.get("SQ_INSTS_VALU_ADD_F64", 0) + cpu.get("SQ_INSTS_VALU_MUL_F64", 0) + cpu.get("SQ_INSTS_VALU_FMA_F64", 0)
        lines.append(f"per unit: VALU {cpu['SQ_INSTS_VALU']:.4g} wave-instr (f64 add {cpu.get('SQ_INSTS_VALU_ADD_F64', 0):.4g}, mul {cpu.get('SQ_INSTS_VALU_MUL_F64', 0):.4g}, "
                     f"fma {cpu.get('SQ_INSTS_VALU_FMA_F64', 0):.4g} = {100 * f64 / cpu['SQ_INSTS_VALU']:.1f} % of VALU), SALU {cpu.get('SQ_INSTS_SALU', 0):.4g}, SMEM {cpu.get('SQ_INSTS_SMEM', 0):.4g}, "
                     f"LDS {cpu.get('SQ_INSTS_LDS', 0):.4g}, branch {cpu.get('SQ_INSTS_BRANCH', 0):.4g}, waves {cpu.get('SQ_WAVES', 0):.4g}")
        if cpu.get("SQ_WAVE_CYCLES"):
            lines.append(f"          wave-cycles {cpu['SQ_WAVE_CYCLES']:.4g}, busy cycles {cpu.get('SQ_BUSY_CYCLES', 0):.4g}, ACTIVE_INST_VALU {cpu.get('SQ_ACTIVE_INST_VALU', 0):.4g}, WAIT_ANY {cpu.get('SQ_WAIT_ANY', 0):.4g} "
                         f"({100 * cpu.get('SQ_WAIT_ANY', 0) / cpu['SQ_WAVE_CYCLES']:.1f} % of wave-cycles), WAIT_INST_ANY {cpu.get('SQ_WAIT_INST_ANY', 0):.4g}")
        if match and "kernel_trace_avg_ms_per_dispatch" in ent:
            sec = ent["kernel_trace_avg_ms_per_dispatch"] * 1e-3 / units
            lines.append(f"          VALU issue = {4 * cpu['SQ_INSTS_VALU'] / (sec * 2.1e9 * 1024) * 100:.1f} % of the SIMD cycles at 2.1 GHz (4 cycles per wave64 VALU instruction, 1 024 SIMDs); "
                         f"executed f64 {64 * (cpu.get('SQ_INSTS_VALU_ADD_F64', 0) + cpu.get('SQ_INSTS_VALU_MUL_F64', 0) + 2 * cpu.get('SQ_INSTS_VALU_FMA_F64', 0)) / sec / 1e12:.2f} TFLOP/s")
    if "fetch_bytes_per_unit" in ent:
        lines.append(f"HBM per unit: FETCH {ent['fetch_bytes_per_unit'] / 1e6:.3f} MB, WRITE {ent.get('write_bytes_per_unit', 0) / 1e6:.3f} MB")
    doc["entries"][key] = ent
    open(os.path.join(S, "summary_" + os.path.basename(D) + ".txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # the --stats table of the stats pass, as rocprofv3 wrote it
    for f in glob.glob(os.path.join(D, "stats", "**", "*kernel_stats.csv"), recursive=True):
        os.system(f"cp '{f}' '{os.path.join(S, 'kernel_stats_' + os.path.basename(D) + '.csv')}'")
json.dump(doc, open(os.path.join(S, "pmc_part.json"), "w"), indent=1)
