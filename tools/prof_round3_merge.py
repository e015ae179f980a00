"""Merges gpurun_out/r3prof_*/summary/round3_pmc_part.json into profiles/round3_pmc.json and copies the per-configuration text
summaries / kernel-stats tables to profiles/round3_<config>.{txt,csv}.  Later parts override earlier ones per key."""
import glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
parts = sorted(glob.glob(os.path.join(R, "gpurun_out", "r3prof_*", "summary", "round3_pmc_part.json")), key=os.path.getmtime)
if len(sys.argv) > 1:
    parts = [p for p in parts if any(t in p for t in sys.argv[1:])]
doc = None
for p in parts:
    d = json.load(open(p))
    if doc is None:
        doc = {"_about": d["_about"], "entries": {}}
    doc["entries"].update(d["entries"])
    S = os.path.dirname(p)
    for f in glob.glob(os.path.join(S, "summary_*.txt")):
        shutil.copy(f, os.path.join(R, "profiles", "round3_" + os.path.basename(f)[len("summary_"):]))
    for f in glob.glob(os.path.join(S, "kernel_stats_*.csv")):
        shutil.copy(f, os.path.join(R, "profiles", "round3_" + os.path.basename(f)))
json.dump(doc, open(os.path.join(R, "profiles", "round3_pmc.json"), "w"), indent=1)
print("merged", len(parts), "parts:", sorted(doc["entries"]))
