#!/bin/bash
# The per-call logs under profiles/rNN/ (small blocks, few queries, odd widths): writes gpurun_out/r/*.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r
python tools/bench_small_blocks.py 2>&1 | grep -v amdgpu > gpurun_out/r/small_blocks.log
python tools/wide_small_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r/wide_small_probe.log
python tools/wn18rr_small_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r/wn18rr_small_probe.log
python tools/few_queries_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r/few_queries_probe.log
python tools/stream_check.py 2>&1 | grep -v amdgpu > gpurun_out/r/stream_check.log
( for a in "transe 14541 128" "distmult 14541 128" "transe 14541 64" "distmult 14541 64" "transe 40943 128"; do echo "== $a (model rows queries)"; bash tools/trace_pass.sh $a; done ) > gpurun_out/r/trace_pass_eval_batch.log 2>&1
( for a in "transe 575000 4" "transe 4600000 4" "distmult 4600000 4" "complex 4600000 4" "simple 4600000 4" "distmult 575000 4"; do echo "== $a (model rows queries)"; bash tools/trace_pass.sh $a; done ) > gpurun_out/r/trace_pass_few_queries.log 2>&1
python tools/exact_small_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r/exact_small_probe.log
for f in gpurun_out/r/*.log; do echo "== $f"; tail -n 3 $f; done
