cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_LDS SQ_WAVES_EQ_64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/rpq$i -o pmc -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-probe $@ > /tmp/q$i.log 2>&1
  f=$(find /tmp/rpq$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "rank_" not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"][:40], r["Counter_Name"])
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"] or 0)
for (k, c), (n, s) in agg.items(): print(k, c, n, "%.4g" % (s / n))
PY
done
