#!/bin/bash
# A/B of kernel build variants (tools/mkvar.sh) on one GPU: each variant runs the dragon bench (and teapot with
# AB_TEAPOT=1) and prints one line; the name "main" is the in-tree build.
# usage (on the GPU box): tools/run_ab.sh variantA variantB ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" = main ]; then unset SRT_LIB_ROOT; else export SRT_LIB_ROOT=$PWD/variants/$v; fi
  timeout -k 10 120 python3 -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab/$v.dragon.log 2>&1 || { echo "$v dragon FAILED"; tail -3 gpurun_out/ab/$v.dragon.log; exit 1; }
  if [ -n "$AB_TEAPOT" ]; then
    timeout -k 10 120 python3 -u bench.py --workload teapot --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab/$v.teapot.log 2>&1 || { echo "$v teapot FAILED"; exit 1; }
  fi
  python3 - $v <<'PY'
import json, os, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab/{v}.dragon.log").read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_frame"]; w = d["roofline"]["work_per_frame"]["trace_shadow"]
line = (f'{v:10s} dragon {d["ms_per_step"]:8.2f} ms shadow {k["trace_shadow"]:8.2f} prim {k["trace_primary"]:6.2f} shade {k["shade"]:6.2f} | lanes/inner '
        f'{w["inner_visits"] / max(w["wave_inner_steps"], 1):5.1f} tri/leafstep {w["tri_tests"] / max(w["wave_leaf_steps"], 1):5.1f}')
if os.path.exists(f"gpurun_out/ab/{v}.teapot.log") and os.environ.get("AB_TEAPOT"):
    t = json.loads(open(f"gpurun_out/ab/{v}.teapot.log").read().strip().splitlines()[-1])
    line += f' | teapot {t["ms_per_step"]:6.3f} ms'
print(line, flush=True)
PY
done
