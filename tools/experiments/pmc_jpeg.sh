#!/bin/bash
# SQ counters of the JPEG decoder kernels (GPU box):  bash tools/experiments/pmc_jpeg.sh [batch] [tag]
# Two counter passes (kernel trace only); summary: per kernel mean duration, wave-instructions by class per launch, busy share;
# gpurun_out/<tag>_jpeg_decoder.json = what bench.py reports as ingest_inclusive.decoder (copy into profiles/).
B=${1:-256}; TAG=${2:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_jpeg_a $O/pmc_jpeg_b
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_jpeg_a -- python3 $R/tools/bench_jpeg.py --batch $B --reps 1 > $O/pmc_jpeg_a.log 2>&1 || { tail -5 $O/pmc_jpeg_a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_jpeg_b -- python3 $R/tools/bench_jpeg.py --batch $B --reps 1 > $O/pmc_jpeg_b.log 2>&1 || { tail -5 $O/pmc_jpeg_b.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
summary = {}
for tag in "ab":
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
    fs = glob.glob("$O/pmc_jpeg_%s/**/*counter_collection.csv" % tag, recursive=True)
    seen = set()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if "jpeg" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); n[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k in sorted(acc):
        print(tag, k, "launches", n[k], "mean_us %.1f" % (dur[k] / n[k] / 1e3), {c: "%.3g" % (v / n[k]) for c, v in sorted(acc[k].items())})
        if tag == "a":
            # one decode of the run = $B pairs = 2 * $B frames (ofk_pairs_upload_jpeg: both frames of every pair in one decoder batch)
            calls = [x for x in (n[k],)][0]
            summary[k] = {"launches_in_run": calls, "total_ms_in_run": round(dur[k] / 1e6, 4), "mean_us": round(dur[k] / n[k] / 1e3, 1),
                          "SQ_INSTS_VALU_in_run": int(acc[k].get("SQ_INSTS_VALU", 0)), "SQ_WAVE_CYCLES_in_run": int(acc[k].get("SQ_WAVE_CYCLES", 0)),
                          "SQ_WAIT_INST_ANY_in_run": int(acc[k].get("SQ_WAIT_INST_ANY", 0)), "SQ_ACTIVE_INST_VALU_in_run": int(acc[k].get("SQ_ACTIVE_INST_VALU", 0))}
decodes = max(1, summary.get("k_jpeg_scan", {}).get("launches_in_run", 1))        # one k_jpeg_scan per decode
frames = 2 * $B
out = {"_note": "rocprofv3 --kernel-trace --pmc SQ_* over tools/bench_jpeg.py --batch $B --reps 1 (tools/experiments/pmc_jpeg.sh): every decoder kernel alone on the chip; "
                "per decode of %d 1080p 4:2:0 quality-80 frames (counter passes run at a lower clock than unprofiled runs: durations read 1-3 %% long)" % frames,
       "frames_per_decode": frames, "decodes_in_run": decodes, "kernels": {}}
for k, v in summary.items():
    ms = v["total_ms_in_run"] / decodes
    valu = v["SQ_INSTS_VALU_in_run"] / decodes
    out["kernels"][k] = {"launches_per_decode": round(v["launches_in_run"] / decodes, 2), "ms_per_decode": round(ms, 4), "ms_per_512_frames": round(ms * 512 / frames, 4),
                         "valu_instr_per_decode": int(valu), "valu_ginstr_per_s": round(valu / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                         "valu_frac_of_1228.8": round(valu / (ms * 1e-3) / 1e9 / 1228.8, 4) if ms > 0 else None,
                         "wait_share_of_wave_cycles": round(v["SQ_WAIT_INST_ANY_in_run"] / max(1, v["SQ_WAVE_CYCLES_in_run"]), 3)}
out["decoder_ms_per_512_frames"] = round(sum(x["ms_per_512_frames"] for n_, x in out["kernels"].items() if n_.startswith("k_jpeg")), 4)
json.dump(out, open("$O/${TAG}_jpeg_decoder.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)); print("decoder ms per 512 frames:", out["decoder_ms_per_512_frames"])
PY
