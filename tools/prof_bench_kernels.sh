cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/refresh; tag=r04; mkdir -p $out
prof() { local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o k -- python3 bench.py "$@" > $out/${tag}_bench_${name}_under_rocprof.json 2> $out/prof_$name.err || return 1
  cp "$(find $out/prof_$name -name 'k_kernel_stats.csv' | head -1)" $out/${tag}_bench_${name}_kernel_stats.csv
  rm -rf $out/prof_$name $out/prof_$name.err; }
prof gowalla_bf16 --steps 20 --warmup 5 --no-probe || exit 1
prof yelp_bf16 --workload yelp --steps 20 --warmup 5 --no-probe --no-train --no-cpu-baseline || exit 1
prof yelp_fp32 --workload yelp --dtype fp32 --steps 20 --warmup 5 --no-probe --no-train --no-cpu-baseline || exit 1
prof gowalla_fp32 --dtype fp32 --steps 20 --warmup 5 --no-probe --no-train --no-cpu-baseline || exit 1
for n in gowalla_bf16 yelp_bf16 yelp_fp32 gowalla_fp32; do python3 - <<PY
import csv,json
rows=list(csv.DictReader(open("$out/r04_bench_${n}_kernel_stats.csv")))
d=json.load(open("$out/r04_bench_${n}_under_rocprof.json"))
f=[float(r["AverageNs"])/1e3 for r in rows if "prop_fwd_kernel" in r["Name"] and ", true, false, false" in r["Name"].replace("ngacf::","")]
print("$n", "csv prop_fwd eval rows", [round(x,2) for x in f], "mean", round(sum(f)/max(len(f),1),2), "line avg_launch_us", d["roofline"]["avg_launch_us"])
PY
done
