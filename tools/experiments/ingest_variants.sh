#!/bin/bash
# double-buffered compressed ingest (tools/experiments/probe_ingest.py, steady-state decode time and loop) for every build_variants/libofk_*.so
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; PKG=$R/drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd
cp $PKG/libofk.so /tmp/libofk_product.so
for rep in 1 2; do
for lib in $R/build_variants/libofk_*.so; do
  name=$(basename $lib .so); name=${name#libofk_}
  cp $lib $PKG/libofk.so
  (cd $R && timeout -k 10 200 python3 tools/experiments/probe_ingest.py --batch 512 --reps 12 > $O/iv_$name.txt 2>&1) || { tail -3 $O/iv_$name.txt; }
  echo "== $name"; grep "decode  slot" $O/iv_$name.txt | tail -6 | awk '{d=$6-$4; s+=d; n++} END {printf "   steady decode call %.2f ms\n", s/n}'
  grep "decode  slot" $O/iv_$name.txt | tail -7 | awk 'NR==1{t0=$4} {t1=$4} END {printf "   steady loop period %.2f ms -> %.0f pairs/s\n", (t1-t0)/6, 512000*6/(t1-t0)}'
done
done
cp /tmp/libofk_product.so $PKG/libofk.so
