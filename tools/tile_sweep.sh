#!/bin/bash
# One GPU, the many-rank frame path (bench.py --gather-probe): frame time against tile size and number of lanes.
# usage (on the GPU box): tools/tile_sweep.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/tiles
run() { # name, args...
  n=$1; shift
  timeout -k 10 120 python3 -u bench.py --gather-probe --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/tiles/$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/tiles/$n.log; exit 1; }
  python3 - $n <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/tiles/{sys.argv[1]}.log").read().strip().splitlines()[-1])
k = d["roofline"]["kernel_ms_per_frame"]
print(f'{sys.argv[1]:14s} tiles {d["config"]["tiles"]:3d} lanes {d["roofline"]["concurrent_lanes"]} {d["ms_per_step"]:8.2f} ms  no-events {d["ms_per_step_without_launch_events"]:8.2f} ms  kernels {sum(k.values()):8.2f}', flush=True)
PY
}
run t64_l2 --tile 480 270 --lanes 2 || exit 1
run t64_l3 --tile 480 270 --lanes 3 || exit 1
run t64_l4 --tile 480 270 --lanes 4 || exit 1
run t16_l2 --tile 960 540 --lanes 2 || exit 1
run t16_l3 --tile 960 540 --lanes 3 || exit 1
