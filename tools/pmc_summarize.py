"""Averages rocprofv3 --pmc counter_collection CSVs per kernel name."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if not ("k_hmc_s" in k or "k_mh_" in k or "k_smc" in k):
            continue
        print(d, k, {c: sum(v) / len(v) for c, v in sorted(cs.items())}, "dispatches", len(next(iter(cs.values()))))
