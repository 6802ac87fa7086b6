#!/bin/bash
# SQ counters of the JPEG decoder kernels (GPU box):  bash tools/experiments/pmc_jpeg.sh [batch]
# Two counter passes (kernel trace only); summary: per kernel mean duration, wave-instructions by class per launch, busy share.
B=${1:-256}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_jpeg_a $O/pmc_jpeg_b
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_jpeg_a -- python3 $R/tools/bench_jpeg.py --batch $B --reps 1 > $O/pmc_jpeg_a.log 2>&1 || { tail -5 $O/pmc_jpeg_a.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_jpeg_b -- python3 $R/tools/bench_jpeg.py --batch $B --reps 1 > $O/pmc_jpeg_b.log 2>&1 || { tail -5 $O/pmc_jpeg_b.log; exit 1; }
python3 - <<PY
import csv, glob, collections
for tag in "ab":
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
    fs = glob.glob("$O/pmc_jpeg_%s/**/*counter_collection.csv" % tag, recursive=True)
    seen = set()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "jpeg" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); n[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k in sorted(acc):
        print(tag, k, "launches", n[k], "mean_us %.1f" % (dur[k] / n[k] / 1e3), {c: "%.3g" % (v / n[k]) for c, v in sorted(acc[k].items())})
PY
