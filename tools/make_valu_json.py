#!/usr/bin/env python3
"""profiles/<round>_valu_pmc.json from the SQ counter passes of tools/profile_round.sh (gpurun_out/pmc_SQ*, one row per dispatch
and counter): mean counter values per launch of every pipeline kernel, bench.py --no-overlap --streams 1 (every kernel alone,
one launch per step covering the whole batch).  bench.py reads SQ_INSTS_VALU_per_launch for roofline.achieved."""
import collections
import csv
import glob
import json
import sys

import os
import re

# exact kernel base name -> stage (round 2 matched substrings: `k_select` swallowed `k_select_prep`).  A stage's entry describes its
# MAIN kernel (the one bench.py's roofline prices); the others are listed per kernel under "kernels".
MAIN_OF = {"k_gray_bgr8": "gray", "k_pyr3_stream": "pyr", "k_mineig_pair": "eig", "k_select_greedy": "select", "k_lk15q": "lk", "k_pairs_solve": "solve"}
OTHER = ("k_pyr_down", "k_pyr_down_stream", "k_mineig", "k_mineig_stream", "k_zero_detect_state", "k_select_prep", "k_select_pick", "k_select", "k_lk15", "k_lk")


def base_name(kernel_name):
    n = kernel_name.replace("void ", "").strip().split("(")[0]
    return re.sub(r"<.*$", "", n).strip()


def main():
    out_path, batch = sys.argv[1], int(sys.argv[2])
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    names = collections.defaultdict(set)
    tag = os.path.basename(out_path).split("_")[0]
    dirs = sorted(glob.glob(f"gpurun_out/pmc_{tag}_SQ*")) or sorted(glob.glob("gpurun_out/pmc_SQ*"))
    for d in dirs:
        files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
        if not files:
            continue
        f = max(files, key=os.path.getmtime)                      # gpurun merges every call's output: one run = the newest file of a pass
        for r in csv.DictReader(open(f)):
            k = base_name(r["Kernel_Name"])
            if k in MAIN_OF or k in OTHER:
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
                names[k].add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
    if not tot:
        sys.exit("no SQ counter CSV under gpurun_out/pmc_SQ*")
    stages, kernels = {}, {}
    for k in tot:
        d = {"kernel": ", ".join(sorted(names[k])), "dispatches_sampled": max(n[k].values())}
        for c in sorted(tot[k]):
            d[f"{c}_per_launch"] = int(tot[k][c] / n[k][c])
        kernels[k] = d
        if k in MAIN_OF:
            stages[MAIN_OF[k]] = d
    json.dump({"_note": "rocprofv3 --pmc SQ_* --kernel-trace (tools/profile_round.sh; bench.py --no-overlap --streams 1, so one launch per "
                        "stage and step covers the whole batch): mean counter value per launch.  SQ_INSTS_VALU = wave-level VALU "
                        "instructions; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).  bench.py prices "
                        "SQ_INSTS_VALU per launch / kernel duration against 1228.8 G wave-instr/s (one per 2 clocks per SIMD, 1024 SIMDs, 2.4 GHz).",
               "batch": batch, "valu_peak_ginstr_per_s": 1228.8, "stages": stages, "kernels": kernels}, open(out_path, "w"), indent=1)
    print(json.dumps(stages, indent=1))


if __name__ == "__main__":
    main()
