mkdir -p gpurun_out/r
WORKLOADS="wikidata5m-transe" bash tools/profile_all.sh > gpurun_out/r/profile_all.log 2>&1
python tools/wikidata_shard_model.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r/wikidata_shard_model.log
( for a in "transe 575000 4" "transe 4600000 4" "distmult 4600000 4" "complex 4600000 4" "simple 4600000 4" "distmult 575000 4"; do echo "== $a (model rows queries)"; bash tools/trace_pass.sh $a; done ) > gpurun_out/r/trace_pass_few_queries.log 2>&1
( for a in "transe 14541 128" "distmult 14541 128" "transe 14541 64" "distmult 14541 64"; do echo "== $a (model rows queries)"; bash tools/trace_pass.sh $a; done ) > gpurun_out/r/trace_pass_eval_batch.log 2>&1
python tools/stream_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r/stream_check.log
tail -3 gpurun_out/r/profile_all.log; cat gpurun_out/r/wikidata_shard_model.log | head -4
