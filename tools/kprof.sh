#!/bin/bash
# Per-kernel GPU durations of tools/kbench.py under rocprofv3 (run on the GPU box):
#   tools/kprof.sh <tag> [kbench args...]   -> gpurun_out/kprof_<tag>.txt
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/kprof_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o k -- python tools/kbench.py "$@" > "$out.log" 2>&1
python - "$out/k_kernel_stats.csv" > "gpurun_out/kprof_$tag.txt" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ngacf' in r['Name']:
        print("%9.2f us avg  %6s calls  min %8.2f  %s" % (float(r['AverageNs'])/1e3, r['Calls'], float(r['MinNs'])/1e3, r['Name'][:110]))
PY
cat "gpurun_out/kprof_$tag.txt"
