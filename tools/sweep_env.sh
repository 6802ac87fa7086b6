#!/bin/bash
# bench.py under a list of values of one environment knob (GPU box):  tools/sweep_env.sh NAME v1 v2 ... [-- bench flags]
NAME=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for v in "${VALS[@]}"; do
  env $NAME=$v timeout -k 10 200 python bench.py --cpu-sample 0 "$@" > gpurun_out/sw_${NAME}_$v.log 2>&1 || { echo "$NAME=$v failed"; tail -3 gpurun_out/sw_${NAME}_$v.log; exit 1; }
  echo "== $NAME=$v"; python tools/show_bench.py gpurun_out/sw_${NAME}_$v.log | grep -E "frame-pairs|eig |lk |gray|pyr"
done
