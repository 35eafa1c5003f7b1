#!/bin/bash
# GPU box: the same probes under each variant library of tools/build_variant.sh
#   tools/ab_variants.sh v0 v1 ...   ->  gpurun_out/ab_variants.log
cd "$(dirname "$0")/.."
out=gpurun_out/ab_variants.log; : > $out
for v in "$@"; do
  export NGACF_LIB_PATH=$PWD/ngacf_amd/lib/variants/$v.so
  echo "== $v" >> $out
  python tools/pair_probe.py 2>/dev/null | head -n 2 >> $out
  python tools/train_profile.py 2>/dev/null | tail -n 1 | cut -c1-40 >> $out
  NGACF_FUSE_BWD=0 python tools/train_profile.py 2>/dev/null | tail -n 1 | cut -c1-40 >> $out
done
cat $out
