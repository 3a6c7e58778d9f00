#!/bin/bash
# eager launches against hipGraph replay of the same step, full test set and one 8-way query shard's share
for w in fb15k237-transe fb15k237-distmult wikidata5m-transe; do
  for g in off on; do
    python bench.py --workload $w --graph $g --steps 20 --warmup 3 --no-cpu-baseline --no-hbm-probe 2>&1 | grep '^{' | \
      python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$w', '$g', round(r['ms_per_step'],4), 'ms', r['config']['launch'], r['parity_check'])"
  done
done
