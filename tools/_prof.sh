cd $GRAFT_REPO_ROOT
bash tools/profile_all.sh 2>&1 | tail -12
python tools/bench_eval_loop.py transe 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r/eval_loop.log
python tools/bench_eval_loop.py complex 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r/eval_loop.log
python tools/bench_small_blocks.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r/small_blocks.log; tail -8 gpurun_out/r/small_blocks.log
SHARDS=827 bash tools/trace_gaps.sh transe 2>&1 | tee gpurun_out/r/trace_gaps_transe_128q.log | tail -14
SHARDS=827 bash tools/trace_gaps.sh distmult 2>&1 | tee gpurun_out/r/trace_gaps_distmult_128q.log | tail -12
