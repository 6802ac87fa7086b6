#!/bin/bash
# Profiles of one round (GPU box):  bash tools/profile_round.sh <tag> [batch]
#   1. HBM traffic: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (kernel trace only, as the pool requires)
#      -> profiles/r01_traffic_pmc.json (tools/make_traffic_json.py), read by bench.py for roofline.traffic
#   2. rocprofv3 --kernel-trace --stats of the default bench command and of --no-overlap --streams 1 (every kernel alone on the chip)
#      -> gpurun_out/prof_<tag>_{overlap,serial}/...kernel_stats.csv (copy into profiles/)
#   3. the plain bench line -> gpurun_out/bench_<tag>.json
TAG=${1:-r01}; BATCH=${2:-256}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --batch $BATCH --steps 2 --warmup 1 --cpu-sample 0 --streams 1 > $R/gpurun_out/pmc_$c.log 2>&1 || { tail -5 $R/gpurun_out/pmc_$c.log; exit 1; }
done
(cd $R && python3 tools/make_traffic_json.py profiles/r01_traffic_pmc.json $BATCH > gpurun_out/traffic_$TAG.log 2>&1 && cp profiles/r01_traffic_pmc.json gpurun_out/traffic_$TAG.json) || { tail -5 $R/gpurun_out/traffic_$TAG.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_overlap -o p -- python3 $R/bench.py --batch $BATCH --steps 10 --warmup 2 --cpu-sample 0 --no-isolated > $R/gpurun_out/prof_${TAG}_overlap.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_serial -o p -- python3 $R/bench.py --batch $BATCH --steps 10 --warmup 2 --cpu-sample 0 --no-overlap --streams 1 > $R/gpurun_out/prof_${TAG}_serial.log 2>&1 || exit 1
cd $R && python3 bench.py --batch $BATCH > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err && python3 tools/show_bench.py gpurun_out/bench_$TAG.json
