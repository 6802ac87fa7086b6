#!/usr/bin/env python3
"""profiles/<round>_traffic_pmc.json from the two rocprofv3 PMC passes of tools/profile_round.sh.

  python tools/make_traffic_json.py <out.json> <batch> [H W]

Input : gpurun_out/pmc_FETCH_SIZE/**/_counter_collection.csv and gpurun_out/pmc_WRITE_SIZE/** (one row per dispatch
        and counter; TCC FETCH_SIZE / WRITE_SIZE in KiB).
Output: per pipeline KERNEL and per stage, HBM bytes per bench step = (fetch_factor*FETCH_SIZE + WRITE_SIZE) * 1024 summed over
        the launches of a step.  fetch_factor: on gfx950 FETCH_SIZE counts every fabric read request as 64 B although the L2 asks
        for 128-B lines (MI355X_MICROARCH.md, HBM section), so a wide coalesced read reports exactly half of its bytes; the factor
        of every other access pattern of this pipeline is taken from the calibration run (tools/fetch_calib.hip ->
        profiles/<round>_fetch_calibration.json), 2.0 when that file is absent.
Checks: kernels are matched by EXACT name (round 2 matched substrings: `k_select` also counted `k_select_prep`, which doubled the
        step count and halved every stage); steps are counted from k_pairs_solve (one launch per step); the gray conversion's
        bytes must equal its algorithmic 2 * 4 * P * batch within 1 % (the guide's calibrated case) or the script fails.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

# exact kernel base name (template arguments and parameter list stripped) -> stage
STAGE_OF = {"k_gray_bgr8": "gray", "k_pyr_down": "pyr", "k_pyr_down_stream": "pyr", "k_pyr3_stream": "pyr",
            "k_mineig": "eig", "k_mineig_stream": "eig", "k_mineig_pair": "eig", "k_zero_detect_state": "eig",
            "k_select_prep": "select", "k_select_pick": "select", "k_select": "select", "k_select_greedy": "select",
            "k_lk15q": "lk", "k_lk15": "lk", "k_lk": "lk", "k_pairs_solve": "solve", "k_pairs_solve_wg": "solve"}
TAG = "r03"
STEP_KERNEL = "k_pairs_solve"                                    # exactly one launch per bench step and slice


def base_name(kernel_name):
    n = kernel_name.replace("void ", "").strip()
    n = n.split("(")[0]
    return re.sub(r"<.*$", "", n).strip()


def collect(counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    files = glob.glob(f"gpurun_out/pmc_{TAG}_{counter}/**/*counter_collection.csv", recursive=True) or \
        glob.glob(f"gpurun_out/pmc_{counter}/**/*counter_collection.csv", recursive=True)
    if not files:
        sys.exit(f"no counter CSV for {counter}")
    files = [max(files, key=os.path.getmtime)]                  # gpurun merges every call's output into gpurun_out/: one run = the newest file
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = base_name(r["Kernel_Name"])
            if k in STAGE_OF:
                tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n


def main():
    global TAG
    out_path, batch = sys.argv[1], int(sys.argv[2])
    H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1080, 1920)
    TAG = tag = os.path.basename(out_path).split("_")[0]
    factors, calib_src = {}, None
    for p in (f"gpurun_out/{tag}_fetch_calibration.json", f"profiles/{tag}_fetch_calibration.json"):     # a fresh run's file first
        if os.path.exists(p):
            factors = json.load(open(p)).get("fetch_factor_of_kernel", {}); calib_src = p
            break
    fetch, nf = collect("FETCH_SIZE")
    write, nw = collect("WRITE_SIZE")
    steps_f, steps_w = nf[STEP_KERNEL], nw[STEP_KERNEL]
    if steps_f < 1 or steps_w < 1:
        sys.exit(f"{STEP_KERNEL} not found in the counter CSVs: cannot count steps")
    kernels, stages = {}, collections.defaultdict(lambda: {"kernels": [], "hbm_bytes_per_step": 0, "FETCH_SIZE_KiB_per_step": 0.0, "WRITE_SIZE_KiB_per_step": 0.0})
    for k in sorted(set(fetch) | set(write)):
        f_kib = fetch[k] / steps_f; w_kib = write[k] / steps_w
        ff = float(factors.get(k, 2.0))
        kernels[k] = {"stage": STAGE_OF[k], "launches_per_step": round(nf[k] / steps_f, 3), "FETCH_SIZE_KiB_per_step": round(f_kib, 1),
                      "WRITE_SIZE_KiB_per_step": round(w_kib, 1), "fetch_factor": ff, "hbm_bytes_per_step": int((ff * f_kib + w_kib) * 1024)}
        s = stages[STAGE_OF[k]]
        s["kernels"].append(k); s["hbm_bytes_per_step"] += kernels[k]["hbm_bytes_per_step"]
        s["FETCH_SIZE_KiB_per_step"] = round(s["FETCH_SIZE_KiB_per_step"] + f_kib, 1); s["WRITE_SIZE_KiB_per_step"] = round(s["WRITE_SIZE_KiB_per_step"] + w_kib, 1)
    gray_alg = 2 * 4 * H * W * batch
    gray = stages["gray"]["hbm_bytes_per_step"] if "gray" in stages else 0
    if not (0.99 * gray_alg <= gray <= 1.01 * gray_alg):
        sys.exit(f"calibration check failed: k_gray_bgr8 moves {gray} bytes per step by the counters, its algorithmic traffic is "
                 f"2*4*P*B = {gray_alg} (steps counted: {steps_f}); the step count or the FETCH factor is off")
    json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/profile_round.sh) over bench.py --streams 1; KiB "
                        "per bench step, summed over the launches of a kernel; steps counted from k_pairs_solve (one launch per step).  "
                        "hbm_bytes_per_step = (fetch_factor*FETCH_SIZE + WRITE_SIZE)*1024; fetch_factor per kernel from " + (calib_src or "the guide's 2.0 (no calibration file)") +
                        ".  Self-check passed: k_gray_bgr8 = 2*4*P*B within 1 %.",
               "batch": batch, "steps_sampled": steps_f, "gray_check": {"counters": gray, "algorithmic": gray_alg},
               "stages": dict(stages), "kernels": kernels}, open(out_path, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
