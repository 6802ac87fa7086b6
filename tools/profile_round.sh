#!/bin/bash
# Profiles of one round (GPU box):  bash tools/profile_round.sh <tag> [batch] [config]     (results under gpurun_out/; copy into profiles/)
#   config = c1 (default) | c2 | c4 (bench.py --config); for c2 / c4 use the tags <round>c2 / <round>c4: bench.py looks for profiles/<round><config>_*.json
#   0. tools/valu_rates.hip (issue cost per instruction kind)                      -> gpurun_out/<tag>_valu_rates.txt
#   1. HBM traffic: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (kernel trace only, as the pool requires)
#      -> gpurun_out/<tag>_traffic_pmc.json (tools/make_traffic_json.py); bench.py reads profiles/<tag>_traffic_pmc.json for roofline.traffic
#   2. SQ counters (SQ_INSTS_VALU ...) of every kernel, two runs  -> gpurun_out/<tag>_valu_pmc.json (tools/make_valu_json.py)
#   3. rocprofv3 --kernel-trace --stats of the default bench command and of --no-overlap --streams 1 (every kernel alone on the chip)
#      -> gpurun_out/<tag>_kernel_stats_{overlap,serial}.csv
#   4. the plain bench line -> gpurun_out/<tag>_bench.json
TAG=${1:-r04}; BATCH=${2:-512}; CONFIG=${3:-c1}
case $CONFIG in c2) HW="480 640";; c4) HW="2160 3840";; *) HW="1080 1920";; esac
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ $CONFIG = c1 ] && [ -x $R/build_variants/valu_rates ]; then $R/build_variants/valu_rates > $O/${TAG}_valu_rates.txt 2>&1 || exit 1; fi
QUIET="--cpu-sample 0 --no-ingest --config $CONFIG"
#   0b. FETCH_SIZE calibration on known byte counts (tools/fetch_calib.hip) -> gpurun_out/<tag>_fetch_calibration.json
if [ $CONFIG = c1 ] && [ -x $R/build_variants/fetch_calib ]; then
  rm -rf $O/pmc_calib $O/pmc_calib_raw
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_calib -- $R/build_variants/fetch_calib > $O/calib_truth.txt 2> $O/calib.err || { tail -5 $O/calib.err; exit 1; }
  # raw request counters (optional: names differ between ROCm releases, a refusal does not stop the round)
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $O/pmc_calib_raw -- $R/build_variants/fetch_calib > /dev/null 2> $O/calib_raw.err || echo "raw TCC counters not collected (see calib_raw.err)"
  (cd $R && python3 tools/make_fetch_calibration.py gpurun_out/${TAG}_fetch_calibration.json > gpurun_out/calib_$TAG.log 2>&1) || { tail -5 $O/calib_$TAG.log; exit 1; }
  cat $O/calib_$TAG.log
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_${TAG}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${TAG}_$c -- python3 $R/bench.py --batch $BATCH --steps 2 --warmup 1 $QUIET --streams 1 > $O/pmc_$c.log 2>&1 || { tail -5 $O/pmc_$c.log; exit 1; }
done
(cd $R && python3 tools/make_traffic_json.py gpurun_out/${TAG}_traffic_pmc.json $BATCH $HW > gpurun_out/traffic_$TAG.log 2>&1) || { tail -5 $O/traffic_$TAG.log; exit 1; }
rm -rf $O/pmc_${TAG}_SQ
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_${TAG}_SQ -- python3 $R/bench.py --batch $BATCH --steps 2 --warmup 1 $QUIET --no-overlap --streams 1 > $O/pmc_SQ.log 2>&1 || { tail -5 $O/pmc_SQ.log; exit 1; }
rm -rf $O/pmc_${TAG}_SQ2
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${TAG}_SQ2 -- python3 $R/bench.py --batch $BATCH --steps 2 --warmup 1 $QUIET --no-overlap --streams 1 > $O/pmc_SQ2.log 2>&1 || { tail -5 $O/pmc_SQ2.log; exit 1; }
(cd $R && python3 tools/make_valu_json.py gpurun_out/${TAG}_valu_pmc.json $BATCH > gpurun_out/valu_$TAG.log 2>&1) || { tail -5 $O/valu_$TAG.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_overlap -o p -- python3 $R/bench.py --batch $BATCH --steps 10 --warmup 2 $QUIET --no-isolated > $O/prof_${TAG}_overlap.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_serial -o p -- python3 $R/bench.py --batch $BATCH --steps 10 --warmup 2 $QUIET --no-overlap --streams 1 > $O/prof_${TAG}_serial.log 2>&1 || exit 1
cp $(find $O/prof_${TAG}_overlap -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_overlap.csv
cp $(find $O/prof_${TAG}_serial -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats_serial.csv
cd $R && python3 bench.py --batch $BATCH --config $CONFIG > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && python3 tools/show_bench.py gpurun_out/${TAG}_bench.json
