#!/bin/bash
# Per-kernel times of the in-batch loss step (forward + backward launches) under rocprofv3 (kernel trace + stats only), one
# run per shape of bench.INBATCH_SHAPES:   bash tools/inbatch_profile.sh [out-dir under gpurun_out/]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-inbatch}
mkdir -p $OUT
cd $R
for S in inbatch-fb15k237 inbatch-wikidata5m-complex-fp16 inbatch-wikidata5m-complex-fp16-b1024; do
  rm -rf /tmp/rp_ib && mkdir -p /tmp/rp_ib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_ib/trace -o trace -- python tools/inbatch_kernels.py $S > $OUT/$S.stdout.log 2>&1
  find /tmp/rp_ib/trace -name "*kernel_stats.csv" -exec cp {} $OUT/$S.kernel_stats.csv \;
  echo "== $S"; tail -1 $OUT/$S.stdout.log
  python - $OUT/$S.kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
done
