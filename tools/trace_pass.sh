#!/bin/bash
# Kernel-by-kernel timeline of ONE blp_rank_all call with Q queries (half head-, half tail-replacing) against the first
# ROWS rows of a random normalised table.  usage: bash tools/trace_pass.sh <model> <rows> <queries> [D]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-transe}; ROWS=${2:-575000}; Q=${3:-4}; D=${4:-128}
cat > /tmp/one_pass.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from blp_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
table = torch.nn.functional.normalize(torch.randn(($ROWS, $D), device=dev, generator=g), dim=-1)
rel = torch.randn((822, $D), device=dev, generator=g) * 0.1
fixed = torch.randint(0, $ROWS, ($Q,), device=dev, generator=g)
true = torch.randint(0, $ROWS, ($Q,), device=dev, generator=g)
r = torch.randint(0, 822, ($Q,), device=dev, generator=g)
qf, qr, qt = table[fixed].contiguous(), rel[r].contiguous(), table[true].contiguous()
for _ in range(200):
    c = ops.rank_all("$M", table, qf, qr, $Q // 2, q_true=qt)
torch.cuda.synchronize()
PY
rm -rf /tmp/tp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -o t -- python /tmp/one_pass.py > /tmp/tp.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/tp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "blp::" in r["Kernel_Name"]]
names = [r["Kernel_Name"] for r in rows]
first = names[-1]  # walk back to the start of the last call: the last occurrence of the call's first kernel
n_call = len(rows) // 200
seg = rows[-n_call:]
prev_end = None; busy = 0; gaps = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("void blp::", "")[:44]
    print(f"{name:46s} {(e - s) / 1e3:8.1f} us   gap before {gap:6.1f} us")
    busy += e - s; gaps += max(0, s - prev_end) if prev_end else 0; prev_end = e
print(f"busy {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us, span {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3:.1f} us")
PY
