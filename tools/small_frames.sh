#!/bin/bash
# The three small BASELINE workloads (teapot, monkey, textured cube at 1920x1080) on one GPU: ms per frame and the sum of
# the kernels' own times, for each build variant named ("main" = in-tree).  usage (on the GPU box): tools/small_frames.sh main ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/small
for v in "$@"; do
  if [ "$v" = main ]; then unset SRT_LIB_ROOT; else export SRT_LIB_ROOT=$PWD/variants/$v; fi
  for wl in teapot monkey cube_textured; do
    timeout -k 10 120 python3 -u bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/small/$v.$wl.log 2>&1 || { echo "$v $wl FAILED"; tail -3 gpurun_out/small/$v.$wl.log; exit 1; }
  done
  python3 - $v <<'PY'
import json, sys
v = sys.argv[1]
out = f"{v:10s}"
for wl in ("teapot", "monkey", "cube_textured"):
    d = json.loads(open(f"gpurun_out/small/{v}.{wl}.log").read().strip().splitlines()[-1])
    k = d["roofline"]["kernel_ms_per_frame"]
    out += f' | {wl} {d["ms_per_step"]:6.3f} ms (kernels {sum(k.values()):6.3f}: ' + " ".join(f"{n.replace('trace_', '')} {t:.3f}" for n, t in k.items() if t > 0) + ")"
print(out, flush=True)
PY
done
