#!/bin/bash
# A/B builds of the two traversal sources with extra -D flags, linked against the current objects of the other sources:
#   tools/build_variant.sh NAME "-DNGACF_PK_DOT -DNGACF_BWD_UNROLL=2"   ->  ngacf_amd/lib/variants/NAME.so   (NGACF_LIB_PATH selects it)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2
mkdir -p ngacf_amd/lib/variants build/var_$name
for f in propagate propagate_bwd; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I include $flags -c ngacf_amd/csrc/$f.hip -o build/var_$name/$f.o &
done
wait
objs=$(ls build/obj/*.o | grep -v "/propagate.o\|/propagate_bwd.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ngacf_amd/lib/variants/$name.so $objs build/var_$name/propagate.o build/var_$name/propagate_bwd.o
echo built ngacf_amd/lib/variants/$name.so
