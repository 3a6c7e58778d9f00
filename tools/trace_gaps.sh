#!/bin/bash
# Kernel-by-kernel timeline (durations and the idle gaps between consecutive launches) of one ranking
# call at the per-rank size of an 8-GPU query-sharded FB15k-237 evaluation (SHARDS=827: one reference
# eval batch of 64 triples).  usage: [SHARDS=n] bash tools/trace_gaps.sh [model]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-transe}
cat > /tmp/one_rank.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from blp_amd import ops
import bench
cfg = bench.WORKLOADS["fb15k237-$M"]
dev = torch.device("cuda", 0)
table, rel_w, heads, tails, rels = bench.make_data(cfg, dev)
q_fixed, q_rel, true_row = bench.build_queries(table, rel_w, heads, tails, rels)
T = heads.shape[0]; t = (T + ${SHARDS:-8} - 1) // ${SHARDS:-8}
qf = torch.cat((q_fixed[:t], q_fixed[T:T + t])); qr = torch.cat((q_rel[:t], q_rel[T:T + t])); tr = torch.cat((true_row[:t], true_row[T:T + t]))
for _ in range(6):
    c = ops.rank_all("$M", table, qf, qr, t, true_row=tr)
    s = ops.rank_metric_sums(c)
torch.cuda.synchronize()
PY
rm -rf /tmp/tg && rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o t -- python /tmp/one_rank.py > /tmp/tg.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/tg/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last call = from the last true_key launch on (the first kernel of a pre-pass call)
starts = [i for i, r in enumerate(rows) if "true_key" in r["Kernel_Name"]]
seg = rows[starts[-1]:]
prev_end = None; busy = 0; gaps = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("void blp::", "")[:44]
    print(f"{name:46s} {(e - s) / 1e3:8.1f} us   gap before {gap:6.1f} us")
    busy += e - s; gaps += max(0, s - prev_end) if prev_end else 0; prev_end = e
print(f"busy {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us, span {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3:.1f} us")
PY
