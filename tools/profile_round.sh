#!/bin/bash
# Profile the bench on the GPU box: rocprofv3 kernel trace + stats, then PMC passes (each in its
# own run, kernel-trace only -- never combined with sys/hip/hsa tracing).  Raw output goes to /tmp;
# only the small CSV summaries are copied under gpurun_out/prof/ (then into profiles/ by hand).
#   usage: bash tools/profile_round.sh [bench args...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT /tmp/rp && mkdir -p $OUT /tmp/rp
cd $R
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-sub-results $@"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp/trace -o trace -- python bench.py $ARGS > $OUT/trace_stdout.log 2>&1
echo trace rc=$?
find /tmp/rp/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find /tmp/rp/trace -name "*kernel_trace.csv" -exec sh -c 'head -400 "$1" > '$OUT'/kernel_trace_head.csv' _ {} \;
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rp/pmc$i -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-sub-results $@ > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($grp) rc=$?"
  f=$(find /tmp/rp/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$OUT/pmc$i.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = (r.get("Kernel_Name", "?")[:90], r.get("Counter_Name"))
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += float(r.get("Counter_Value", 0) or 0)
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches_x_dims,sum,mean_per_row\n")
    for (k, c), (n, s) in agg.items():
        f.write(f"\"{k}\",{c},{n},{s:.6g},{s / n:.6g}\n")
PY
done
ls -la $OUT
du -sh $OUT
