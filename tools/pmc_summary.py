"""profiles/pmc_traffic.json from the per-workload rocprofv3 summaries of a round:
    python tools/pmc_summary.py profiles/r01
HBM-side read bytes per launch of each workload's dominant kernel (pmc4.csv: FETCH_SIZE in KiB, x 1024 x 2 on
gfx950 -- MI355X_MICROARCH.md, HBM section) and, for the MFMA kernels, the matrix-pipe busy fraction
(pmc7.csv: SQ_VALU_MFMA_BUSY_CYCLES summed over the chip, normalised by 1024 SIMDs x kernel time x 2.4 GHz;
kernel time from kernel_stats.csv).  Every entry also records `bench_pass_ms`, the ranking-pass time of the bench line taken
with the profile (bench.json: roofline.kernel_ms) -- bench.py drops the counters when its live pass time has moved away from
it -- and the file is stamped with the round directory and the commit it describes (`_profile`)."""
import csv, json, os, sys

import subprocess

DOMINANT = {"fb15k237-transe": "rank_sad_kernel", "fb15k237-distmult": "rank_gemm_bf16", "fb15k237-complex": "rank_gemm_bf16",
            "fb15k237-simple": "rank_gemm_bf16", "fb15k237-transe-d768": "wide_rank_sad_kernel",
            "fb15k237-transe-clustered": "rank_sad_kernel", "fb15k237-distmult-clustered": "rank_gemm_bf16",
            "wikidata5m-transe": "rank_stream_kernel", "wikidata5m-complex": "rank_stream_dot_kernel",  # (the ring kernels: all passes of a step in one launch)
            "wikidata5m-transe-block": "rank_sad_kernel", "wikidata5m-complex-block": "rank_gemm_bf16",
            "wikidata5m-protocol": "rank_sad_kernel",
            "wikidata5m-transe-full": "rank_stream_kernel", "wikidata5m-complex-full": "rank_stream_dot_kernel",  # (3 447 passes in one launch)
            "wikidata5m-transe-f16": "rank_stream16_kernel", "wikidata5m-complex-f16": "rank_stream_dot16_kernel"}  # (the 16-bit copy of the table)


def rows(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def main(root):
    out_path = os.path.join(os.path.dirname(os.path.abspath(root)), "pmc_traffic.json")
    old = json.load(open(out_path)) if os.path.exists(out_path) else {}
    try:
        commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        commit = "unknown"
    out = {"_note": old.get("_note", ""), "_profile": {"round": os.path.basename(os.path.normpath(root)), "commit": commit}}
    for workload, kernel in DOMINANT.items():
        d = os.path.join(root, workload)
        entry = {"kernel": kernel}
        if "comment" in old.get(workload, {}):
            entry["comment"] = old[workload]["comment"]
        for r in rows(os.path.join(d, "pmc4.csv")):
            if kernel in r["kernel"] and r["counter"] == "FETCH_SIZE":
                entry["FETCH_SIZE_KiB_mean"] = float(r["mean_per_row"])
                entry["hbm_bytes_per_launch"] = float(r["mean_per_row"]) * 1024 * 2
        ns = None
        for r in rows(os.path.join(d, "kernel_stats.csv")):
            if kernel in r["Name"]:
                ns = float(r["AverageNs"])
                entry["kernel_ns"] = ns
                break
        for r in rows(os.path.join(d, "pmc7.csv")):
            if kernel in r["kernel"] and r["counter"] == "SQ_VALU_MFMA_BUSY_CYCLES" and float(r["mean_per_row"]) > 0 and ns:
                entry["mfma_busy_cycles"] = float(r["mean_per_row"])
                entry["mfma_busy_frac_at_2.4GHz"] = float(r["mean_per_row"]) / (1024 * ns * 2.4)
        try:
            line = json.load(open(os.path.join(d, "bench.json")))
            entry["bench_pass_ms"] = line["roofline"]["kernel_ms"]
            ppl = int(line["roofline"].get("passes_per_launch", 1))
            if ppl > 1:  # one launch walks all passes of the step: the counters of a launch / the passes = one pass's share
                entry["passes_per_launch"] = ppl
                for key in ("FETCH_SIZE_KiB_mean", "hbm_bytes_per_launch", "kernel_ns"):
                    if key in entry:
                        entry[key] = entry[key] / ppl
        except (OSError, ValueError, KeyError):
            pass
        if len(entry) > 1:
            out[workload] = entry
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "comment"} for k, v in out.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r01")
