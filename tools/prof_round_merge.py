"""Merges gpurun_out/prof_<round>_*/summary/pmc_part.json into profiles/<round>_pmc.json (FG_PROF_ROUND, default round4) and copies the
per-configuration text summaries / kernel-stats tables to profiles/<round>_<config>.{txt,csv}.  Later parts override earlier ones per key."""
import glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("FG_PROF_ROUND", "round4")
parts = sorted(glob.glob(os.path.join(R, "gpurun_out", "prof_" + ROUND + "_*", "summary", "pmc_part.json")), key=os.path.getmtime)
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
        shutil.copy(f, os.path.join(R, "profiles", ROUND + "_" + os.path.basename(f)[len("summary_"):]))
    for f in glob.glob(os.path.join(S, "kernel_stats_*.csv")):
        shutil.copy(f, os.path.join(R, "profiles", ROUND + "_" + os.path.basename(f)))
json.dump(doc, open(os.path.join(R, "profiles", ROUND + "_pmc.json"), "w"), indent=1)
print("merged", len(parts), "parts:", sorted(doc["entries"]))
