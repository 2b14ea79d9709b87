#!/bin/bash
# A/B of build variants on the two reflection workloads (monkey 1920x1080, dragon 8K 64 spp); "main" = in-tree build.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in "$@"; do
  if [ "$v" = main ]; then unset SRT_LIB_ROOT; else export SRT_LIB_ROOT=$PWD/variants/$v; fi
  timeout -k 10 200 python3 -u bench.py --workload monkey --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab/$v.monkey.log 2>&1 || { echo "$v monkey FAILED"; tail -3 gpurun_out/ab/$v.monkey.log; exit 1; }
  if [ -z "$AB_NO_8K" ]; then
    timeout -k 10 400 python3 -u bench.py --workload dragon8k --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/ab/$v.dragon8k.log 2>&1 || { echo "$v dragon8k FAILED"; tail -3 gpurun_out/ab/$v.dragon8k.log; exit 1; }
  fi
  python3 - $v <<'PY'
import json, os, sys
v = sys.argv[1]
out = f"{v:10s}"
for wl in ("monkey", "dragon8k"):
    f = f"gpurun_out/ab/{v}.{wl}.log"
    if not os.path.exists(f) or (wl == "dragon8k" and os.environ.get("AB_NO_8K")):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = d["roofline"]["kernel_ms_per_frame"]; w = d["roofline"]["work_per_frame"]["trace_secondary"]
    out += (f' | {wl} {d["ms_per_step"]:9.3f} ms secondary {k["trace_secondary"]:8.3f} primary {k["trace_primary"]:7.3f} shadow {k["trace_shadow"]:9.3f} shade {k["shade"]:7.3f} '
            f'lanes/inner {w["inner_visits"] / max(w["wave_inner_steps"], 1):5.1f}')
print(out, flush=True)
PY
done
