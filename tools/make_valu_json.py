#!/usr/bin/env python3
"""profiles/<round>_valu_pmc.json from the SQ counter passes of tools/profile_round.sh (gpurun_out/pmc_SQ*, one row per dispatch
and counter): mean counter values per launch of every pipeline kernel, bench.py --no-overlap --streams 1 (every kernel alone,
one launch per step covering the whole batch).  bench.py reads SQ_INSTS_VALU_per_launch for roofline.achieved."""
import collections
import csv
import glob
import json
import sys

STAGE_OF = [("k_gray_bgr8", "gray"), ("k_pyr_down", "pyr"), ("k_pyr3", "pyr"), ("k_mineig", "eig"), ("k_select", "select"), ("k_lk", "lk"),
            ("k_pairs_solve", "solve")]


def main():
    out_path, batch = sys.argv[1], int(sys.argv[2])
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    names = collections.defaultdict(set)
    for f in glob.glob("gpurun_out/pmc_SQ*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for sub, stage in STAGE_OF:
                if sub in r["Kernel_Name"]:
                    tot[stage][r["Counter_Name"]] += float(r["Counter_Value"]); n[stage][r["Counter_Name"]] += 1
                    names[stage].add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
                    break
    if not tot:
        sys.exit("no SQ counter CSV under gpurun_out/pmc_SQ*")
    stages = {}
    for s in tot:
        d = {"kernel": ", ".join(sorted(names[s])), "dispatches_sampled": max(n[s].values())}
        for c in sorted(tot[s]):
            d[f"{c}_per_launch"] = int(tot[s][c] / n[s][c])
        stages[s] = d
    json.dump({"_note": "rocprofv3 --pmc SQ_* --kernel-trace (tools/profile_round.sh; bench.py --no-overlap --streams 1, so one launch per "
                        "stage and step covers the whole batch): mean counter value per launch.  SQ_INSTS_VALU = wave-level VALU "
                        "instructions; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).  bench.py prices "
                        "SQ_INSTS_VALU per launch / kernel duration against 1228.8 G wave-instr/s (one per 2 clocks per SIMD, 1024 SIMDs, 2.4 GHz).",
               "batch": batch, "valu_peak_ginstr_per_s": 1228.8, "stages": stages}, open(out_path, "w"), indent=1)
    print(json.dumps(stages, indent=1))


if __name__ == "__main__":
    main()
