#!/bin/bash
# Everything profiles/r04/ holds, in one go on the GPU box (writes gpurun_out/r/): the per-workload bench lines and rocprofv3
# summaries (tools/profile_all.sh), the default bench line, the micro-benchmark the TransE roof rests on, the in-batch loss
# kernels, and the small logs (reference-batched passes, shard model, loop layouts).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r
export WORKLOADS="fb15k237-transe fb15k237-distmult fb15k237-complex fb15k237-simple fb15k237-transe-d768 wikidata5m-transe wikidata5m-complex wikidata5m-transe-f16 wikidata5m-complex-f16 wikidata5m-transe-block wikidata5m-complex-block wikidata5m-protocol"
bash tools/profile_all.sh > gpurun_out/r/profile_all.log 2>&1
python bench.py --steps 20 --warmup 5 2> gpurun_out/r/bench_default.stderr | tail -1 > gpurun_out/r/bench_default.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/sad tools/sad_ubench.hip > gpurun_out/r/sad_ubench.log 2>&1 && /tmp/sad >> gpurun_out/r/sad_ubench.log 2>&1
mkdir -p gpurun_out/r/inbatch
cd /tmp && export TMPDIR=/tmp
for shape in inbatch-fb15k237 inbatch-wikidata5m-complex-fp16 inbatch-wikidata5m-complex-fp16-b1024; do
  rm -rf /tmp/ib && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ib -o t -- python $R/tools/inbatch_kernels.py $shape > /dev/null 2>&1
  find /tmp/ib -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r/inbatch/$shape.kernel_stats.csv \;
done
cd $R
python tools/queries_per_pass.py 2>&1 | grep -v amdgpu > gpurun_out/r/queries_per_pass.log
python tools/bench_batches.py 2>&1 | grep -v amdgpu > gpurun_out/r/bench_batches.log
python tools/wikidata_shard_model.py 2>&1 | grep -v amdgpu > gpurun_out/r/wikidata_shard_model.log
python tools/stream_check.py 2>&1 | grep -v amdgpu > gpurun_out/r/stream_check.log
python tools/bench_small_blocks.py 2>&1 | grep -v amdgpu > gpurun_out/r/small_blocks.log
ls gpurun_out/r; du -sh gpurun_out/r
