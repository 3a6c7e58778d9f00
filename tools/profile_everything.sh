#!/bin/bash
# Everything profiles/rNN/ holds, in one go on the GPU box (writes gpurun_out/r/): per-workload bench lines (compact line + the
# details file), rocprofv3 kernel stats and the seven PMC passes (tools/profile_all.sh), the default bench line, the in-batch
# loss kernels, and the small logs (clustered sweep, compute_loss probes, reference-batched passes, shard model, loop layouts).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r
export WORKLOADS="${WORKLOADS:-fb15k237-transe fb15k237-distmult fb15k237-complex fb15k237-simple fb15k237-transe-clustered fb15k237-distmult-clustered fb15k237-transe-d768 wikidata5m-transe wikidata5m-complex wikidata5m-transe-f16 wikidata5m-complex-f16 wikidata5m-transe-block wikidata5m-complex-block wikidata5m-transe-full wikidata5m-complex-full wikidata5m-protocol}"
bash tools/profile_all.sh > gpurun_out/r/profile_all.log 2>&1
[ -n "$ONLY_WORKLOADS" ] && { ls gpurun_out/r; exit 0; }
python bench.py --steps 20 --warmup 5 --details gpurun_out/r/bench_default_details.json 2> gpurun_out/r/bench_default.stderr > gpurun_out/r/bench_default.json
bash tools/inbatch_profile.sh r/inbatch > gpurun_out/r/inbatch_profile.log 2>&1
python tools/inbatch_probe.py 2>&1 | grep "^inbatch" > gpurun_out/r/inbatch_probe.log
python tools/clustered_sweep.py 2>&1 | grep -v amdgpu > gpurun_out/r/clustered_sweep.log
python tools/loss_step_probe.py 2>&1 | grep -v "amdgpu\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" > gpurun_out/r/loss_step_probe.log
python tools/autograd_floor_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r/autograd_floor_probe.log
python tools/queries_per_pass.py 2>&1 | grep -v amdgpu > gpurun_out/r/queries_per_pass.log
python tools/bench_batches.py 2>&1 | grep -v amdgpu > gpurun_out/r/bench_batches.log
python tools/wikidata_shard_model.py 2>&1 | grep -v amdgpu > gpurun_out/r/wikidata_shard_model.log
python tools/bench_small_blocks.py 2>&1 | grep -v amdgpu > gpurun_out/r/small_blocks.log
python -c "
import sys; sys.path.insert(0, '.')
from blp_amd import _lib
print('blp_selftest:', _lib.selftest(0)); print(_lib.device_caps(0))" 2>&1 | grep -v amdgpu > gpurun_out/r/selftest.log
BLP_BENCH_BACKEND=gloo python bench.py --gpus 8 --dry-nccl 2> gpurun_out/r/dry_nccl_8_gloo.stderr > gpurun_out/r/dry_nccl_8_gloo.json
BLP_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 2 --warmup 1 --details gpurun_out/r/bench_8_gloo_ranks_one_gpu_functional_details.json 2> gpurun_out/r/bench_8_gloo.stderr > gpurun_out/r/bench_8_gloo_ranks_one_gpu_functional.json
ls gpurun_out/r; du -sh gpurun_out/r
