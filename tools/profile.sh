#!/bin/bash
# Profiles bench.py on the GPU box: one rocprofv3 --kernel-trace --stats run, then one --pmc run per counter group
# (a group holds what one pass can collect).  usage: tools/profile.sh <name> [bench.py arguments]
# Output: gpurun_out/prof_<name>/{summary.txt,bench_line.json,source_hash.txt}: the raw rocprofv3 output is condensed by
# profiles/summarize.py on the box and deleted there (it can exceed what gpurun copies back); KEEP_RAW=1 keeps it.
R=$GRAFT_REPO_ROOT
NAME=$1
O=$R/gpurun_out/prof_$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/trace.log 2>&1 || { echo "trace run failed"; tail -5 $O/trace.log; exit 1; }
grep '^{"metric"' $O/trace.log | tail -1 > $O/bench_line.json   # the profiler prints its own lines after the program's
# which kernels this profile belongs to (bench.py only takes counters from a summary whose hash matches its tree)
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_hash())" > $O/source_hash.txt
run_group() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d $O/pmc_$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "${EXTRA[@]}" > $O/pmc_$name.log 2>&1 || { echo "pmc group $name failed"; tail -5 $O/pmc_$name.log; exit 1; }
  echo "pmc group $name done"
}
EXTRA=("$@")
run_group sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
run_group sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH &&
run_group sq3 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT &&
run_group tc1 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum &&
run_group ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE &&
run_group fetch FETCH_SIZE &&
run_group write WRITE_SIZE || exit 1
python3 $R/profiles/summarize.py $O "$NAME: bench.py $*" > $O/summary.txt || exit 1
[ -n "$KEEP_RAW" ] || rm -rf $O/trace $O/pmc_*/
echo "summary written"
