mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q --timeout 900 -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/pytest_r02b.log
tail -15 gpurun_out/pytest_r02b.log
for lib in blp_amd/libblp_hip.so blp_amd/libblp_hip.noasm.so blp_amd/libblp_hip.timing.so; do timeout 300 python tools/gemm_ab.py $lib distmult complex 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/gemm_ab.log; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o t -- python $GRAFT_REPO_ROOT/bench.py --workload fb15k237-distmult --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-probe --no-sub-results > $GRAFT_REPO_ROOT/gpurun_out/prof_distmult.log 2>&1
find /tmp/rp -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/distmult_kernel_stats.csv \;
head -25 $GRAFT_REPO_ROOT/gpurun_out/distmult_kernel_stats.csv | cut -c1-160
