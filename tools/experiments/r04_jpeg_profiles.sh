#!/bin/bash
# decoder profiles of the round (GPU box): SQ counters per decoder kernel, in-loop timeline of one double-buffered decode, the bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
timeout -k 10 300 bash $R/tools/experiments/pmc_jpeg.sh 256 r04 > $O/r04_jpeg_pmc.txt 2>&1 || { tail -5 $O/r04_jpeg_pmc.txt; exit 1; }
tail -1 $O/r04_jpeg_pmc.txt
cp $O/r04_jpeg_decoder.json $R/profiles/r04_jpeg_decoder.json
cd /tmp && export TMPDIR=/tmp && rm -rf $O/r04_probe
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/r04_probe -- python3 $R/tools/experiments/probe_ingest.py --batch 512 --reps 8 > $O/r04_probe.txt 2>&1 || { tail -5 $O/r04_probe.txt; exit 1; }
grep -v "^[EWI]2026" $O/r04_probe.txt | tail -3
cd $R && timeout -k 10 900 bash tools/experiments/r04_benchlines.sh
