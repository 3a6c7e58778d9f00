mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q --timeout 900 -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/pytest_r02c.log
tail -12 gpurun_out/pytest_r02c.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r02c.json 2> gpurun_out/bench_r02c.err; tail -3 gpurun_out/bench_r02c.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench_r02c.json"))
print("transe ms/step", r["ms_per_step"], "raw", r["ms_per_step_raw_only"], "kernel", r["roofline"]["kernel_ms"], "frac", r["roofline"]["frac"])
for k, v in r["sub_results"].items():
    print(k, "ms/step", v["ms_per_step"], "raw", v["ms_per_step_raw_only"], "kernel", v["roofline"]["kernel_ms"], "frac", v["roofline"]["frac"])
print("hbm", r["hbm_probe"]["frac"], r["hbm_probe"]["kernel_ms"])
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp -o t -- python $GRAFT_REPO_ROOT/bench.py --workload fb15k237-transe --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-probe --no-sub-results > $GRAFT_REPO_ROOT/gpurun_out/prof_transe.log 2>&1
find /tmp/rp -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/transe_kernel_stats.csv \;
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/transe_kernel_stats.csv")))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']:>6s}%")
PY
