#!/bin/bash
# Per-kernel times of one bench workload under rocprofv3 (kernel trace + stats only):
#   bash tools/kernel_stats_one.sh <workload> [out-dir under gpurun_out/]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-fb15k237-transe}
OUT=$R/gpurun_out/${2:-ks}
mkdir -p $OUT && rm -rf /tmp/rp1 && mkdir -p /tmp/rp1
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1/trace -o trace -- python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-sub-results --no-hbm-probe > $OUT/$W.stdout.log 2>&1
find /tmp/rp1/trace -name "*kernel_stats.csv" -exec cp {} $OUT/$W.kernel_stats.csv \;
python - $OUT/$W.kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']:>6s} %")
PY
