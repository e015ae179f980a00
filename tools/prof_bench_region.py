"""The timed region of `python bench.py` in a rocprofv3 --kernel-trace: the 25-transition sampling launches of the headline kernel.
The bench issues, in order: spin-up launches of 100 transitions on a scratch engine (> 3 ms each), ceil(W / 25) adaptive warmup
launches, then R x (K / 25) timed sampling launches.  usage: prof_bench_region.py <rocprof dir> <full bench document> <out txt>"""
import csv, glob, json, sys
d, jf, out = sys.argv[1:4]
j = json.load(open(jf))          # the full document (bench.py --full-out)
K, W, R, per = j["steps"], j["warmup"], j["timed_regions"]["repeats"], j["config"]["transitions_per_launch"]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hk = [r for r in rows if "k_hmc_sep_steps<false, 0, 0>" in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in hk]
last_spin = max(i for i, t in enumerate(dur) if t > 3.0 and i < len(dur) - R * (K // per))
n_warm = (W + per - 1) // per
first = last_spin + 1 + n_warm
timed = dur[first:first + R * (K // per)]
lines = [f"rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline` (K = {K}, W = {W}, {R} timed regions, {per} transitions per launch)",
         f"kernel: {hk[0]['Kernel_Name'].split('(')[0]}; {len(hk)} dispatches in the run; spin-up ends at dispatch {last_spin}, {n_warm} warmup launches, then the timed ones",
         f"timed sampling launches: {len(timed)}, average {sum(timed) / len(timed):.4f} ms, min {min(timed):.4f}, max {max(timed):.4f}",
         f"bench.py's own HIP-event average of the same launches (unprofiled run differs by the profiler's overhead): roofline.avg_launch_ms = {j['roofline']['avg_launch_ms']:.4f} ms",
         "per timed region (average launch ms): " + ", ".join(f"{sum(timed[r * (K // per):(r + 1) * (K // per)]) / (K // per):.4f}" for r in range(R)),
         f"value of this (profiled) run: {j['value']:.4g} leapfrog-steps/s; regions: " + ", ".join(f"{v:.4g}" for v in j["timed_regions"]["value"]["all"])]
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
