#!/usr/bin/env python3
"""profiles/<round>_traffic_pmc.json from the two rocprofv3 PMC passes of tools/profile_round.sh.

Input : gpurun_out/pmc_FETCH_SIZE/**/_counter_collection.csv and gpurun_out/pmc_WRITE_SIZE/** (one row per dispatch
        and counter; TCC FETCH_SIZE / WRITE_SIZE in KiB).
Output: per pipeline stage, HBM bytes per bench step = (2*FETCH_SIZE + WRITE_SIZE) * 1024 summed over the stage's launches
        of a step (on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads: MI355X_MICROARCH.md, HBM section;
        exact for k_gray_bgr8, an upper bound for narrow access patterns).  bench.py reads this file for roofline.traffic.
"""
import collections
import csv
import glob
import json
import sys

STAGE_OF = [("k_gray_bgr8", "gray"), ("k_pyr_down", "pyr"), ("k_pyr3", "pyr"), ("k_mineig", "eig"), ("k_select", "select"), ("k_lk", "lk"),
            ("k_pairs_solve", "solve")]


def collect(counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(int); names = collections.defaultdict(set)
    files = glob.glob(f"gpurun_out/pmc_{counter}/**/*counter_collection.csv", recursive=True)
    if not files:
        sys.exit(f"no counter CSV for {counter}")
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for sub, stage in STAGE_OF:
                if sub in r["Kernel_Name"]:
                    tot[stage] += float(r["Counter_Value"]); n[stage] += 1
                    names[stage].add(r["Kernel_Name"].split("(")[0].replace("void ", ""))
                    break
    return tot, n, names


def main():
    out_path, batch = sys.argv[1], int(sys.argv[2])
    fetch, nf, names = collect("FETCH_SIZE")
    write, nw, _ = collect("WRITE_SIZE")
    steps = nf["select"]                                  # one k_select launch per bench step (timed, warm-up and isolated pass alike)
    stages = {}
    for _, s in STAGE_OF:
        f_kib = fetch[s] / steps; w_kib = write[s] / max(1, nw["select"])
        stages[s] = {"kernels": sorted(names[s]), "launches_per_step": nf[s] / steps, "FETCH_SIZE_KiB_per_step": round(f_kib, 1),
                     "WRITE_SIZE_KiB_per_step": round(w_kib, 1), "hbm_bytes_per_step": int((2 * f_kib + w_kib) * 1024)}
    json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/profile_round.sh) over bench.py; KiB per "
                        "bench step, summed over the launches of the stage.  hbm_bytes_per_step = (2*FETCH_SIZE + WRITE_SIZE)*1024: on "
                        "gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section; exact "
                        "for k_gray_bgr8); for dword/byte access patterns the factor is uncalibrated (upper bound).",
               "batch": batch, "steps_sampled": steps, "stages": stages}, open(out_path, "w"), indent=1)
    print(json.dumps(stages, indent=1))


if __name__ == "__main__":
    main()
