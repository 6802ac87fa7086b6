#!/bin/bash
# bench.py under a sweep of one ofk_set_tuning knob (GPU box):  bash tools/experiments/sweep_tune.sh KNOB "v1 v2 ..." [bench flags]
K=$1; VALS=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in $VALS; do
  timeout -k 10 200 python $R/bench.py --cpu-sample 0 --steps 20 --no-ingest --tune $K=$v "$@" > $R/gpurun_out/sweep_${K}_$v.json 2>/dev/null || { echo "$K=$v failed"; exit 1; }
  echo "== $K=$v"; python $R/tools/show_bench.py $R/gpurun_out/sweep_${K}_$v.json | grep -E "frame-pairs|gray|eig|pyr|lk "
done
