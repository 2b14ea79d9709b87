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

def derive(per_launch, out=sys.stdout):
    """Ratios per traversal / shading kernel from the per-launch counter sums {(kernel, counter): value}.
    One XCD's GRBM_GUI_ACTIVE is the kernel's duration in shader clocks; the counter is summed over the 8 XCDs."""
    kernels = sorted({k for k, _ in per_launch if "k_trace" in k or "k_shade" in k or "k_resolve" in k})
    if not kernels:
        return
    print(file=out)
    print("# derived, per kernel: clocks = GRBM_GUI_ACTIVE / 8 XCDs; per-CU figures divide by 256 CUs", file=out)
    print("#   lanes/VALU inst = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; TA busy = TA_TA_BUSY_sum / (clocks * 256);", file=out)
    print("#   VALU inst/clk/CU = SQ_INSTS_VALU / (clocks * 256); L1 lookups/clk/CU = TCP_TOTAL_CACHE_ACCESSES_sum / (clocks * 256);", file=out)
    print("#   VMEM rd inst/clk/CU = SQ_INSTS_VMEM_RD / (clocks * 256); L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS)", file=out)
    for k in kernels:
        g = lambda c: per_launch.get((k, c))
        clocks = g("GRBM_GUI_ACTIVE") / 8 if g("GRBM_GUI_ACTIVE") else None
        cells = []
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            cells.append(f"lanes/VALU inst {g('SQ_THREAD_CYCLES_VALU') / g('SQ_ACTIVE_INST_VALU'):5.1f}")
        if clocks:
            cu = clocks * 256
            for label, c in (("TA busy", "TA_TA_BUSY_sum"), ("VALU inst/clk/CU", "SQ_INSTS_VALU"), ("SALU inst/clk/CU", "SQ_INSTS_SALU"),
                             ("VMEM rd inst/clk/CU", "SQ_INSTS_VMEM_RD"), ("L1 lookups/clk/CU", "TCP_TOTAL_CACHE_ACCESSES_sum")):
                if g(c) is not None:
                    cells.append(f"{label} {g(c) / cu:6.3f}")
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            cells.append(f"L2 hit {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):6.4f}")
        print(f"{k:40s} " + "; ".join(cells), file=out)


if sys.argv[1] == "--derive":  # append the derived table to a summary written earlier
    table = {}
    for line in open(sys.argv[2]):
        if " per_launch=" in line and " launches=" in line:
            left, value = line.rsplit(" per_launch=", 1)
            left = left.rsplit(" launches=", 1)[0].rstrip()
            kernel, counter = left.rsplit(None, 1)
            table[(kernel.strip(), counter)] = float(value)
    derive(table, open(sys.argv[2], "a"))
    sys.exit(0)


base, title = sys.argv[1], sys.argv[2]
everything = {}
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
        everything[k] = v / n[k]
derive(everything)
