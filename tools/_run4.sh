mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^$" | tail -30 > gpurun_out/pytest_r02d.log
tail -8 gpurun_out/pytest_r02d.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r02d.json 2> gpurun_out/bench_r02d.err; tail -3 gpurun_out/bench_r02d.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench_r02d.json"))
print("transe ms/step", r["ms_per_step"], "raw", r["ms_per_step_raw_only"], "kernel", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"])
for k, v in r["sub_results"].items():
    print(k, "ms/step", v["ms_per_step"], "raw", v["ms_per_step_raw_only"], "kernel", v["roofline"]["kernel_ms"], "frac", v["roofline"]["frac"])
print("hbm", r["hbm_probe"]["frac"], r["hbm_probe"]["kernel_ms"])
PY
python tools/bench_small_blocks.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/small_blocks_r02d.log
