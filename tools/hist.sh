#!/bin/bash
# experiment driver: fraction of the shadow walk's inner-node visits above level k (variants hist<k>)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for k in "$@"; do
  SRT_LIB_ROOT=$PWD/variants/hist$k timeout -k 10 120 python3 -u bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/ab/hist$k.log 2>&1 || { tail -3 gpurun_out/ab/hist$k.log; exit 1; }
  python3 - $k <<'PY'
import json, sys
k = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab/hist{k}.log").read().strip().splitlines()[-1])
w = d["roofline"]["work_per_frame"]["trace_shadow"]
print(f"levels < {k}: {w['box_tests'] / w['inner_visits']:.3f} of {w['inner_visits']} inner visits", flush=True)
PY
done
