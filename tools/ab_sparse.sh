#!/bin/bash
# A/B of the sparse last stage (NGACF_SPARSE_LAST=1/0): bench.py under rocprofv3 --kernel-trace --stats, per-kernel averages.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
for m in ${MODES:-1 0}; do
  export NGACF_SPARSE_LAST=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$m -o ab -- python3 bench.py --no-probe --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/ab_$m.log 2>&1 || exit 1
  f=$(find gpurun_out/ab_$m -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/ab_${m}_kernel_stats.csv
  find gpurun_out/ab_$m -type f ! -name '*stats.csv' -delete
done
