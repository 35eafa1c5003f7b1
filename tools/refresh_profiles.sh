#!/bin/bash
# Regenerates the round's bench / profile files under gpurun_out/refresh/ on the GPU box (copy what is wanted into profiles/).
#   gpurun --timeout 1200 -- 'bash tools/refresh_profiles.sh [tag] [part]'      part: all | bench | roof | pmc | train | aux
tag=${1:-r05}
part=${2:-all}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/refresh
mkdir -p $out
say() { echo "[refresh] $(date +%H:%M:%S) $*"; }
prof() {   # prof <name> <bench args...>: rocprofv3 kernel stats of a bench run
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o k -- python3 bench.py --no-roof "$@" > $out/${tag}_bench_${name}_under_rocprof.json 2> $out/prof_$name.err || return 1
  cp "$(find $out/prof_$name -name 'k_kernel_stats.csv' | head -1)" $out/${tag}_bench_${name}_kernel_stats.csv
  rm -rf $out/prof_$name $out/prof_$name.err
}
if [ "$part" = all ] || [ "$part" = bench ]; then
say "default bench (with the Yelp and 10^9-edge probes)"; timeout -k 10 600 python bench.py > $out/${tag}_bench_gowalla_bf16.json 2> $out/err.log || exit 1
say "gowalla bf16 under rocprof";        prof gowalla_bf16 --steps 20 --warmup 5 --no-probe || exit 1
say "yelp bf16 / fp32 under rocprof";    prof yelp_bf16 --workload yelp --steps 20 --warmup 5 --no-probe --no-train --no-cpu-baseline || exit 1
prof yelp_fp32 --workload yelp --dtype fp32 --steps 20 --warmup 5 --no-probe --no-train --no-cpu-baseline || exit 1
say "gowalla bf16, transforms as their own launches (A/B)"; NGACF_FUSE_NEXT=0 timeout -k 10 300 python bench.py --no-probe --no-cpu-baseline > $out/${tag}_bench_gowalla_bf16_unfused.json 2>> $out/err.log || exit 1
for cfg in "yelp bf16" "yelp fp32" "gowalla fp32" "gowalla128 bf16" "gowalla128 fp32"; do
  set -- $cfg
  say "bench $1 $2";                     timeout -k 10 300 python bench.py --workload $1 --dtype $2 --no-probe --cpu-budget 4 > $out/${tag}_bench_$1_$2.json 2>> $out/err.log || exit 1
done
say "synthetic200m";                     timeout -k 10 500 python bench.py --workload synthetic200m --steps 10 --warmup 2 > $out/${tag}_bench_synthetic200m_bf16.json 2>> $out/err.log || exit 1
say "synthetic200m under rocprof";       prof synthetic200m_bf16 --workload synthetic200m --steps 5 --warmup 1 --no-cpu-baseline || exit 1
for w in gowalla_real yelp_real; do say "bench $w"; timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $out/${tag}_bench_${w}_bf16.json 2>> $out/err.log || exit 1; done
say "gowalla fp32 under rocprof";       prof gowalla_fp32 --dtype fp32 --steps 20 --warmup 5 --no-probe --no-train --no-cpu-baseline || exit 1
say "hub-finish soak";                   timeout -k 10 600 python tools/hub_soak.py --launches 120000 > $out/${tag}_hub_soak.json 2>> $out/err.log || exit 1
say "rehearsal (1 RCCL rank)";           NGACF_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --gpus 1 > $out/${tag}_bench_rehearsal_1rank_rccl.json 2>> $out/err.log || exit 1
fi
if [ "$part" = all ] || [ "$part" = roof ]; then
say "measured gather roof: traversal kernels against their stripped instantiations (Gowalla, Yelp; bf16, fp32; 8 heads, 1 head)"
timeout -k 10 600 python tools/gather_roof.py > $out/${tag}_gather_roof.jsonl 2>> $out/err.log || exit 1
fi
if [ "$part" = all ] || [ "$part" = pmc ]; then
say "PMC traffic + L2 hit rate: gowalla"; timeout -k 10 600 python tools/traffic.py gowalla bf16,fp32 > $out/traffic_gowalla.log 2>&1 && cp gpurun_out/${tag}_traffic_gowalla.json $out/
say "PMC traffic + L2 hit rate: yelp";    timeout -k 10 600 python tools/traffic.py yelp bf16,fp32 > $out/traffic_yelp.log 2>&1 && cp gpurun_out/${tag}_traffic_yelp.json $out/
say "PMC traffic: synthetic1b";           timeout -k 10 900 python tools/traffic.py synthetic1b bf16 > $out/traffic_1b.log 2>&1 && cp gpurun_out/${tag}_traffic_synthetic1b.json $out/
say "PMC traffic: synthetic200m";         timeout -k 10 600 python tools/traffic.py synthetic200m bf16 > $out/traffic_200m.log 2>&1 && cp gpurun_out/${tag}_traffic_synthetic200m.json $out/
say "PMC traffic + L2 hit rate: the kernels of the train step (Gowalla, Yelp; bf16, fp32)"
timeout -k 10 600 python tools/traffic.py trainstep bf16,fp32 > $out/traffic_trainstep.log 2>&1 && cp gpurun_out/${tag}_traffic_trainstep.json $out/
timeout -k 10 600 python tools/traffic.py trainstep bf16,fp32 yelp > $out/traffic_trainstep_yelp.log 2>&1 && cp gpurun_out/${tag}_traffic_trainstep_yelp.json $out/
fi
if [ "$part" = all ] || [ "$part" = train ]; then
say "train step breakdown";              timeout -k 10 300 python tools/train_profile.py --profile > $out/tp_refsem.log 2>&1 && cp gpurun_out/trainprof_refsem.txt $out/${tag}_trainstep_kernels_refsem.txt
python tools/step_timeline.py gpurun_out/trainprof/t_kernel_trace.csv > $out/${tag}_trainstep_timeline.txt
NGACF_FUSE_BWD=0 timeout -k 10 300 python tools/train_profile.py --profile > $out/tp_refsem0.log 2>&1 && cp gpurun_out/trainprof_refsem.txt $out/${tag}_trainstep_kernels_refsem_unfused.txt
python tools/step_timeline.py gpurun_out/trainprof/t_kernel_trace.csv > $out/${tag}_trainstep_timeline_unfused.txt
timeout -k 10 300 python tools/train_profile.py --profile --shared > $out/tp_shared.log 2>&1 && cp gpurun_out/trainprof_shared.txt $out/${tag}_trainstep_kernels_shared.txt
timeout -k 10 200 python tools/train_profile.py > $out/${tag}_trainstep.jsonl 2>/dev/null; timeout -k 10 200 python tools/train_profile.py --shared >> $out/${tag}_trainstep.jsonl 2>/dev/null
NGACF_SPARSE_LAST=0 timeout -k 10 200 python tools/train_profile.py >> $out/${tag}_trainstep.jsonl 2>/dev/null
NGACF_FUSE_NEXT=0 timeout -k 10 200 python tools/train_profile.py >> $out/${tag}_trainstep.jsonl 2>/dev/null
for f in 0 1; do NGACF_FUSE_BWD=$f timeout -k 10 200 python tools/train_profile.py >> $out/${tag}_trainstep.jsonl 2>/dev/null; done
NGACF_FUSE_SPARSE=1 timeout -k 10 200 python tools/train_profile.py >> $out/${tag}_trainstep.jsonl 2>/dev/null
NGACF_DEFER_DW=0 timeout -k 10 200 python tools/train_profile.py >> $out/${tag}_trainstep.jsonl 2>/dev/null
timeout -k 10 200 python tools/bwd_probe.py > $out/${tag}_bwd_probe.log 2>&1; timeout -k 10 200 python tools/pair_probe.py > $out/${tag}_pair_probe.log 2>&1
timeout -k 10 200 python tools/train_profile.py --torch-adam >> $out/${tag}_trainstep.jsonl 2>/dev/null
say "the train step launched eagerly, one chain at a time: every kernel's stand-alone duration"
timeout -k 10 300 python tools/train_profile.py --profile --eager --one-chain > $out/tp_eager.log 2>&1 && cp gpurun_out/trainprof_eager_kernel_stats.csv $out/${tag}_trainstep_eager_kernel_stats.csv && cp gpurun_out/trainprof_refsem_eager.txt $out/${tag}_trainstep_kernels_eager.txt
say "node-partitioned train step on one RCCL rank (replayed)"
timeout -k 10 200 python tools/shard_step_profile.py > $out/${tag}_shard_step_world1.json 2>> $out/err.log
say "epoch + evaluation";                timeout -k 10 300 python tools/eval_bench.py --epochs 3 > $out/${tag}_epoch_eval_gowalla.log 2>&1
say "the ranking kernel alone, both forms (Gowalla, Yelp) + its SQ counters"
timeout -k 10 200 python tools/topk_bench.py --dtype bf16,fp32 --forms wave,lds4,lds8,lds12 > $out/${tag}_topk_bench_gowalla.jsonl 2>> $out/err.log
timeout -k 10 200 python tools/topk_bench.py --workload yelp --dtype bf16 --forms wave,lds8,lds12 > $out/${tag}_topk_bench_yelp.jsonl 2>> $out/err.log
bash tools/topk_pmc.sh > /dev/null 2>&1 && cp gpurun_out/topk_pmc.txt $out/${tag}_topk_pmc_sq_counters.txt
say "ranking kernel hand-off soak (120 000 launches, ~5 min)"; timeout -k 10 900 python tools/topk_soak.py --launches 20000 > $out/${tag}_topk_soak.json 2>> $out/err.log
rm -rf gpurun_out/trainprof
fi
if [ "$part" = all ] || [ "$part" = aux ]; then
say "dense attention (N4)"; : > $out/${tag}_mha_bench_ml100k.jsonl
for fl in "" "--dim 128" "--dim 256"; do timeout -k 10 200 python tools/mha_bench.py $fl >> $out/${tag}_mha_bench_ml100k.jsonl 2>> $out/err.log || exit 1; done
say "node numbering probe"; : > $out/${tag}_renumber_probe.jsonl
for w in gowalla yelp synthetic200m; do timeout -k 10 400 python tools/renumber_probe.py --workload $w >> $out/${tag}_renumber_probe.jsonl 2>> $out/err.log || exit 1; done
say "stand-alone transform at 70 k and 15 M rows"; timeout -k 10 200 python tools/transform_probe.py > $out/${tag}_transform_probe.jsonl 2>> $out/err.log || exit 1
say "peer-store exchange, 2 and 4 ranks sharing the GPU"; timeout -k 10 300 python tools/peer_probe.py --world 2 > $out/${tag}_peer_probe.jsonl 2>> $out/err.log || exit 1
timeout -k 10 300 python tools/peer_probe.py --world 4 >> $out/${tag}_peer_probe.jsonl 2>> $out/err.log || exit 1
say "peer-store protocol soak (+ negative control)"
timeout -k 10 400 python tools/peer_soak.py --world 2 --rounds 20000 2>> $out/err.log | grep "^{" > $out/${tag}_peer_soak.jsonl || exit 1
timeout -k 10 400 python tools/peer_soak.py --world 3 --rounds 10000 2>> $out/err.log | grep "^{" >> $out/${tag}_peer_soak.jsonl || exit 1
timeout -k 10 200 python tools/peer_soak.py --world 2 --rounds 2000 --no-wait 2>> $out/err.log | grep "^{" >> $out/${tag}_peer_soak.jsonl
say "one-rank rehearsal with peer stores (overlapped, then in line)"
NGACF_BENCH_FORCE_DIST=1 NGACF_PEER_GATHER=1 timeout -k 10 400 python bench.py --gpus 1 --no-probe --no-cpu-baseline --no-train > $out/${tag}_bench_rehearsal_1rank_peer.json 2>> $out/err.log || exit 1
NGACF_BENCH_FORCE_DIST=1 NGACF_PEER_GATHER=1 timeout -k 10 400 python bench.py --gpus 1 --no-probe --no-cpu-baseline --no-train --no-overlap > $out/${tag}_bench_rehearsal_1rank_peer_inline.json 2>> $out/err.log || exit 1
fi
say done
