#!/bin/bash
# rocprofv3 kernel stats of a bench workload, the top kernels by total time:  bash tools/kernel_stats.sh <workload> [n]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
w=${1:-fb15k237-transe}
rm -rf /tmp/ks_$w
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$w -o t -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-sub-results --no-hbm-probe > /dev/null 2>&1
f=$(find /tmp/ks_$w -name "*kernel_stats.csv" | head -1)
python - "$f" ${2:-10} <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    print(f"{name[:60]:60s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e3:10.1f} us  {float(r['Percentage']):5.1f} %")
PY
