#!/bin/bash
# Counter comparison of kernel build variants on the dragon bench: a few rocprofv3 --pmc passes per variant, printed
# per launch of the shadow kernel.  usage (on the GPU box): tools/pmc_ab.sh variantA variantB ...   ("main" = in-tree)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset SRT_LIB_ROOT; else export SRT_LIB_ROOT=$R/variants/$v; fi
  O=$R/gpurun_out/pmc_$v; mkdir -p $O
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $grp -d $O/g$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/g$i.log 2>&1 || { echo "$v group $i failed"; tail -5 $O/g$i.log; exit 1; }
  done
  python3 - $O $v <<'PY'
import collections, csv, glob, sqlite3, sys
base, v = sys.argv[1], sys.argv[2]
agg, n = collections.defaultdict(float), collections.Counter()
for d in sorted(glob.glob(base + "/g*/")):
    rows = []
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        rows += [(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f))]
    for f in glob.glob(d + "*/*_results.db"):
        rows += list(sqlite3.connect(f).execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"))
    per = collections.defaultdict(float)
    for disp, kern, ctr, val in rows:
        if "k_trace_shadow<true, false" in kern or "k_trace_shadowILb1ELb0" in kern:
            per[(disp, ctr)] += val
    for (disp, ctr), val in per.items():
        agg[ctr] += val
        n[ctr] += 1
print(f"== {v}: k_trace_shadow<true,false>, per launch")
for k in sorted(agg):
    print(f"{k:40s} {agg[k] / n[k]:.5g}")
PY
done
