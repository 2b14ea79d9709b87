#!/usr/bin/env python3
"""Condense a gpurun_out/prof_* directory (rocprofv3 --kernel-trace --stats and --pmc CSVs) into the text summary
committed under profiles/.  usage: summarize.py <prof_dir> <title> > profiles/<name>.txt"""
import collections
import csv
import glob
import sys

base, title = sys.argv[1], sys.argv[2]
print(f"# {title}")
print("# kernel stats: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline")
print("# PMC: one rocprofv3 --pmc run per counter group, bench.py --steps 1 --warmup 0 (1 timed + 1 counting frame);")
print("#      per_launch = value summed over the kernel's launches / number of launches")
print()
for f in glob.glob(base + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(f'{r["Name"][:64]:64s} calls={r["Calls"]:>3s} avg_ms={float(r["AverageNs"]) / 1e6:11.3f} pct={float(r["Percentage"]):7.3f}')
print()
for d in sorted(glob.glob(base + "/pmc_*")):
    agg, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            if "k_" not in k:
                continue
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, v in sorted(agg.items()):
        print(f"{k[0]:40s} {k[1]:30s} launches={n[k]:2d} per_launch={v / n[k]:.5g}")
