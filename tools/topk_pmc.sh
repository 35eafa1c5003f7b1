#!/bin/bash
# SQ counters of the ranking kernel (tools/topk_bench.py, LDS-shared form) and its traffic across the L2 <-> fabric boundary
# (FETCH_SIZE / WRITE_SIZE in KB, separate passes; FETCH_SIZE is to be doubled on gfx950), one rocprofv3 --pmc pass per set.
#   gpurun -- 'bash tools/topk_pmc.sh [extra topk_bench args]'   -> gpurun_out/topk_pmc.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/topk_pmc
rm -rf $out; mkdir -p $out
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
      "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT"
      "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
i=0
for s in "${sets[@]}"; do
  rocprofv3 --kernel-trace --pmc $s --output-format csv -d $out/p$i -o p -- python3 tools/topk_bench.py --dtype bf16 --forms lds8 --reps 2 "$@" > $out/run$i.log 2>&1 || { tail -5 $out/run$i.log; }
  i=$((i+1))
done
python3 - <<'PY' > gpurun_out/topk_pmc.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/topk_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "eval_topk" not in k: continue
        k = k.split("(")[0][-60:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print("   %-28s %16.0f per launch (%d launches)" % (c, tot[k][c] / n[k][c], n[k][c]))
PY
cat gpurun_out/topk_pmc.txt
