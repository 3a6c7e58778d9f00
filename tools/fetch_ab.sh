#!/bin/bash
# FETCH_SIZE (fabric reads) and duration of one kernel for variant builds of the library, one workload:
#   bash tools/fetch_ab.sh <workload> <kernel substring> lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
W=$1; K=$2; shift 2
for lib in "$@"; do
  rm -rf /tmp/fab && mkdir -p /tmp/fab
  (cd $R && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fab -o p -- python tools/step_ab.py --child $W $lib > /tmp/fab/out.log 2>&1)
  f=$(find /tmp/fab -name "*counter_collection.csv" | head -1)
  python - "$f" "$K" "$lib" "$(tail -1 /tmp/fab/out.log)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE"]
n = len(rows); tot = sum(float(r["Counter_Value"]) for r in rows)
print(f"{sys.argv[3].split('/')[-1]:30s} {sys.argv[2]}: {n} launches, FETCH {tot / max(n, 1) * 2048 / 1e6:9.1f} MB per launch; step {sys.argv[4]} ms (under the profiler)")
PY
done
