#!/usr/bin/env python3
"""Condense a gpurun_out/prof_* directory written by tools/profile.sh (rocprofv3 --kernel-trace --stats and --pmc
runs, CSV or rocpd/sqlite output) into the text summary committed under profiles/.
usage: summarize.py <prof_dir> <title> > profiles/<name>.txt"""
import collections
import csv
import glob
import json
import sqlite3
import sys

base, title = sys.argv[1], sys.argv[2]
print(f"# {title}")
try:
    print("# kernel-source-sha256: " + open(base + "/source_hash.txt").read().strip())
    print("# workload: " + json.loads(open(base + "/bench_line.json").read())["config"]["workload"])
except (OSError, ValueError, KeyError):
    pass  # profile directories written before these files existed
print("# kernel stats: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline")
print("# PMC: one rocprofv3 --pmc run per counter group, bench.py --steps 1 --warmup 0 (1 timed + 1 counting frame);")
print("#      per_launch = value summed over the kernel's launches / number of launches")
print()
for f in glob.glob(base + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(f'{r["Name"][:64]:64s} calls={r["Calls"]:>3s} avg_ms={float(r["AverageNs"]) / 1e6:11.3f} pct={float(r["Percentage"]):7.3f}')
for f in glob.glob(base + "/trace/*/*_results.db"):
    for name, calls, _, avg, pct in sqlite3.connect(f).execute("select * from top_kernels"):
        print(f"{name[:64]:64s} calls={calls:3d} avg_ms={avg / 1e3:11.3f} pct={pct:7.3f}")
print()
for d in sorted(glob.glob(base + "/pmc_*/")):
    agg, n = collections.defaultdict(float), collections.Counter()
    rows = []
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        rows += [(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f))]
    for f in glob.glob(d + "*/*_results.db"):
        rows += list(sqlite3.connect(f).execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"))
    per_dispatch = collections.defaultdict(float)
    for disp, kern, ctr, val in rows:
        k = kern.split("(")[0][:40]
        if "k_" in k:
            per_dispatch[(disp, k, ctr)] += val  # one row per hardware instance of the counter
    for (disp, k, ctr), v in per_dispatch.items():
        agg[(k, ctr)] += v
        n[(k, ctr)] += 1
    for k, v in sorted(agg.items()):
        print(f"{k[0]:40s} {k[1]:30s} launches={n[k]:2d} per_launch={v / n[k]:.5g}")
