#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. -DFLAG2=.." : builds build/variants/NAME.so (experiments; defines
# OCTL_EXPERIMENTS, without which the result-changing / instrumentation switches are a compile error: common.h)
set -e
NAME=$1; FLAGS=$2
OUT=build/variants
mkdir -p $OUT/obj_$NAME
for f in octreelib_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-function -Iinclude -DOCTL_EXPERIMENTS $FLAGS -c $f -o $OUT/obj_$NAME/$b.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj_$NAME/*.o -o $OUT/$NAME.so -ldl
rm -rf $OUT/obj_$NAME
