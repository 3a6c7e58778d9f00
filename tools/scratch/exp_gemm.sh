#!/bin/bash
for e in 0 4 5 6; do
  BLP_EXTRA_HIPCC_FLAGS="-DBLP_BF_TILES=16 -DBLP_EXP=$e" python -c "from blp_amd import build; build.build(force=True)" 2>&1 | grep -i "error"
  echo "EXP=$e: $(python bench.py --workload fb15k237-distmult --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-probe 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["parity_check"])')"
done
