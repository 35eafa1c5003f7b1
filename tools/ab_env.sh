#!/bin/bash
# GPU box: tools/train_profile.py under several environment settings, one line each
#   tools/ab_env.sh "" "NGACF_SPLIT_JOIN=0" "NGACF_HOIST_ROWS=0 NGACF_SPLIT_JOIN=0"   ->  gpurun_out/ab_env.log
cd "$(dirname "$0")/.."
out=gpurun_out/ab_env.log; : > $out
for e in "$@"; do
  echo "== [$e] $(env $e python tools/train_profile.py $AB_ARGS 2>/dev/null | tail -n 1 | cut -c1-42)" >> $out
done
cat $out
