#!/bin/bash
# End-of-round measurements on the GPU box: profile of the headline workload, then the bench lines kept under profiles/.
# usage: tools/round_end.sh <tag>      (writes gpurun_out/prof_<tag>_dragon/ and gpurun_out/<tag>_bench_*.jsonl)
T=$1
R=$GRAFT_REPO_ROOT
cd $R
[ -n "$SKIP_PROFILE" ] || bash tools/profile.sh ${T}_dragon || exit 1   # SKIP_PROFILE=1: bench lines only
# under its final name, so that the bench line below takes its traffic and cache counters from it (and records that path)
[ -n "$SKIP_PROFILE" ] || cp gpurun_out/prof_${T}_dragon/summary.txt profiles/${T}_dragon_summary.txt
line() { grep '^{"metric"' | tail -1; }
timeout -k 10 300 python3 -u bench.py 2> gpurun_out/${T}_bench_dragon.err | line > gpurun_out/${T}_bench_dragon.jsonl || exit 1
echo "dragon done"
for w in teapot monkey cube_textured; do
  timeout -k 10 200 python3 -u bench.py --workload $w --steps 20 --warmup 3 --cpu-seconds 10 2> gpurun_out/${T}_bench_$w.err | line > gpurun_out/${T}_bench_$w.jsonl || exit 1
  echo "$w done"
done
# the two-rank code path on one GPU (gloo, the ranks share the card): tiles, stealing, gather
timeout -k 10 300 python3 -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 \
  --backend gloo --share-gpu --steps 2 --warmup 1 --no-cpu-baseline 2> gpurun_out/${T}_rehearse2.err | line > gpurun_out/${T}_bench_dragon_2ranks_gloo.jsonl || { tail -5 gpurun_out/${T}_rehearse2.err; exit 1; }
echo "two ranks done"
