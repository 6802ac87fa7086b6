#!/bin/bash
# build_variants/libofk_<name>.so with k_jpeg.hip compiled under extra flags:  bash tools/experiments/build_jpeg_variants.sh name1 "flags1" name2 "flags2" ...
R=/root/repo; CS="$R/drone-stabilisation-using-optical-flow-gps-and-inertial-sensors_amd/csrc"
make -s -C $CS -j 8 || exit 1
mkdir -p $R/build_variants; rm -f $R/build_variants/libofk_*.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-unused-value -I$R/include -I$CS"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $extra -c $CS/k_jpeg.hip -o /tmp/k_jpeg_$name.o || exit 1
  objs=$(ls $CS/build/*.o | grep -v k_jpeg.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/libofk_$name.so $objs /tmp/k_jpeg_$name.o -ldl || exit 1
  echo built $name
done
