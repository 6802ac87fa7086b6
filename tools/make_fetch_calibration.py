#!/usr/bin/env python3
"""profiles/<round>_fetch_calibration.json: FETCH_SIZE against known byte counts, per access pattern of the pipeline.

  python tools/make_fetch_calibration.py <out.json>

Inputs (GPU box, tools/profile_round.sh):
  gpurun_out/calib_truth.txt               stdout of build_variants/fetch_calib: per kernel, bytes requested by the lanes and the
                                           bytes of the unique 64-B / 128-B lines it touches (every line exactly once, buffer 2 GiB)
  gpurun_out/pmc_calib/**/*counter_collection.csv    rocprofv3 --pmc FETCH_SIZE of the same run (KiB per dispatch)
  gpurun_out/pmc_calib_raw/**              optional second pass with the raw TCC request counters

For every pattern: factor128 = unique-128-B-line bytes / FETCH_SIZE bytes and factor64 likewise.  The L2 fetches whole 128-B lines
when factor128 comes out at 2.0 for a pattern (each request tallied at 64 B): then 2 x FETCH_SIZE IS the traffic of any kernel with
that pattern, over-fetch of partially used lines included.  `fetch_factor_of_kernel` maps the pipeline kernels to the factor of
their pattern (tools/make_traffic_json.py reads it).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

PATTERN_OF_KERNEL = {"k_gray_bgr8": "c_wide16", "k_pyr3_stream": "c_wide16", "k_pyr_down_stream": "c_dword4", "k_mineig_pair": "c_rows128",
                     "k_mineig_stream": "c_rows128", "k_select_prep": "c_qword8", "k_select_pick": "c_qword8", "k_select": "c_qword8",
                     "k_select_greedy": "c_qword8", "k_lk15q": "c_lkrows<9,32>", "k_lk15": "c_lkrows<9,32>", "k_pairs_solve": "c_dword4",
                     "k_zero_detect_state": "c_dword4"}


def norm(name):
    n = name.replace("void ", "").strip().split("(")[0]
    return n.replace(" ", "")


def main():
    out_path = sys.argv[1]
    truth = {}
    for line in open("gpurun_out/calib_truth.txt"):
        m = re.match(r"(\S+) requested (\d+) unique64 (\d+) unique128 (\d+)", line)
        if m:
            truth[m.group(1)] = {"requested": int(m.group(2)), "unique64": int(m.group(3)), "unique128": int(m.group(4))}
    if not truth:
        sys.exit("gpurun_out/calib_truth.txt holds no pattern line")
    counters = collections.defaultdict(dict)
    for d in ("gpurun_out/pmc_calib", "gpurun_out/pmc_calib_raw"):
        files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            k = norm(r["Kernel_Name"])
            if k in truth:
                counters[k][r["Counter_Name"]] = counters[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    patterns = {}
    for k, t in truth.items():
        c = counters.get(k, {})
        if "FETCH_SIZE" not in c:
            sys.exit(f"no FETCH_SIZE row for {k}")
        fb = c["FETCH_SIZE"] * 1024.0
        patterns[k] = dict(t, FETCH_SIZE_bytes=int(fb), factor64=round(t["unique64"] / fb, 4), factor128=round(t["unique128"] / fb, 4),
                           requested_over_fetch=round(t["requested"] / fb, 4), raw={n: v for n, v in c.items() if n != "FETCH_SIZE"})
    # the factor of a pattern: 2.0 when FETCH_SIZE x 2 reproduces the 128-B-line bytes (within 5 %), 1.0 when FETCH_SIZE itself
    # reproduces the 64-B-line bytes, else the measured 128-B ratio
    factor = {}
    for k, p in patterns.items():
        factor[k] = 2.0 if abs(p["factor128"] - 2.0) <= 0.1 else (1.0 if abs(p["factor64"] - 1.0) <= 0.05 else p["factor128"])
    json.dump({"_note": "tools/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE: every kernel touches each cache line once in a 2 GiB buffer, so "
                        "the bytes that cross the fabric are known (unique 64-B / 128-B lines).  factor128 = unique128 / FETCH_SIZE.",
               "patterns": patterns, "factor_of_pattern": factor, "pattern_of_kernel": PATTERN_OF_KERNEL,
               "fetch_factor_of_kernel": {k: factor[p] for k, p in PATTERN_OF_KERNEL.items() if p in factor}}, open(out_path, "w"), indent=1)
    for k, p in patterns.items():
        print(f"{k:18s} FETCH {p['FETCH_SIZE_bytes'] / 1e6:9.1f} MB  requested {p['requested'] / 1e6:9.1f}  u64 {p['unique64'] / 1e6:9.1f} (x{p['factor64']})  "
              f"u128 {p['unique128'] / 1e6:9.1f} (x{p['factor128']})  -> factor {factor[k]}")


if __name__ == "__main__":
    main()
