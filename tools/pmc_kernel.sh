#!/bin/bash
# SQ counters of one kernel of bench.py (GPU box).  Each --pmc group is its own run (the hardware has few SQ counters),
# with --kernel-trace only, as the pool requires.   tools/pmc_kernel.sh <kernel-name-substring> [bench flags]
K=${1:-k_mineig_stream}; shift
OUT=/root/repo/gpurun_out/pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU"
G3="SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES"
i=0
for G in "$G1" "$G2" "$G3"; do
    i=$((i + 1))
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$OUT/g$i" -o pmc -- python3 /root/repo/bench.py --no-overlap --steps 2 --warmup 1 --cpu-sample 0 "$@" > "$OUT/g$i.log" 2>&1 || { tail -5 "$OUT/g$i.log"; exit 1; }
done
cd /root/repo && python3 - "$K" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('gpurun_out/pmc/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for c in sorted(tot):
    print(f"{c:28s} {tot[c] / n[c]:16.0f}  (mean of {n[c]} dispatches)")
PY
