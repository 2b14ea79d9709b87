#!/bin/bash
# dragon frame cut into 480x270 tiles (two lanes) per variant; usage on the GPU box: tools/ab_tiles.sh variantA ...
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = main ]; then unset SRT_LIB_ROOT; else export SRT_LIB_ROOT=$PWD/variants/$v; fi
  timeout -k 10 200 python3 -u bench.py --steps 3 --warmup 2 --no-cpu-baseline --tile 480 270 2>/dev/null | grep '^{' | python3 -c "
import json,sys; l=json.loads(sys.stdin.read()); print('$v tiled 480x270', round(l['ms_per_step'],2), 'ms')"
done
