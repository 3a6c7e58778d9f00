#!/bin/bash
# All bench workloads: bench JSON (with CPU baseline for the default workload) + tools/profile_round.sh
# summaries, under gpurun_out/r/<workload>/.  Copy into profiles/rNN/ by hand.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r
for w in ${WORKLOADS:-fb15k237-transe fb15k237-distmult fb15k237-complex fb15k237-simple fb15k237-transe-d768 wikidata5m-transe wikidata5m-complex wikidata5m-transe-block}; do
  mkdir -p gpurun_out/r/$w
  extra="--no-cpu-baseline --no-sub-results"
  [ "$w" = "fb15k237-transe" ] && extra="--no-sub-results"   # (the headline keeps its CPU baseline; the sub-results: the default run)
  python bench.py --workload $w --steps 10 --warmup 2 $extra --details gpurun_out/r/$w/bench_details.json 2> gpurun_out/r/$w/bench.stderr | tail -1 > gpurun_out/r/$w/bench.json
  bash tools/profile_round.sh --workload $w --no-hbm-probe > gpurun_out/r/$w/profile.log 2>&1
  cp gpurun_out/prof/kernel_stats.csv gpurun_out/prof/pmc*.csv gpurun_out/r/$w/ 2>/dev/null
  echo "$w done: $(cut -c1-160 gpurun_out/r/$w/bench.json)"
done
rm -rf gpurun_out/prof
du -sh gpurun_out/r
