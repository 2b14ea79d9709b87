#!/bin/bash
# monkey (reflection-heavy) bench line per variant; usage on the GPU box: tools/ab_monkey.sh variantA ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = main ]; then unset SRT_LIB_ROOT; else export SRT_LIB_ROOT=$PWD/variants/$v; fi
  timeout -k 10 120 python3 -u bench.py --workload monkey --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python3 -c "
import json,sys; l=json.loads(sys.stdin.read()); print('$v monkey', round(l['ms_per_step'],4), {k: round(x,4) for k,x in l['roofline']['kernel_ms_per_frame'].items()})"
done
