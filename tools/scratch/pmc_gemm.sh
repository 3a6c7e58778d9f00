#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
  i=$((i+1))
  rm -rf /tmp/rp$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rp$i -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-probe --workload fb15k237-distmult > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/rp$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "rank_gemm_bf16" not in r.get("Kernel_Name", ""): continue
    a = agg.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"] or 0)
for c, (n, s) in agg.items(): print(f"{c:32s} {s/n:14.6g}  (x{n})")
PY
done
