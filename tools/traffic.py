"""HBM-side traffic of the dominant kernel from PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE in SEPARATE rocprofv3 --pmc passes of the same bench command (kernel-trace only), units of KB, FETCH_SIZE
doubled on gfx950; a third pass collects TCC_HIT_sum / TCC_MISS_sum (the L2 hit rate bench.py's cache-gather roof is built
from).  Writes gpurun_out/r05_traffic_<workload>.json -- copy it to profiles/ (bench.py reads roofline.traffic and
roofline.l2_hit_rate from there and labels them from_profile).
Run on the GPU box:  python tools/traffic.py gowalla   (this process never touches the GPU itself).
    python tools/traffic.py trainstep [bf16,fp32] [yelp]   -> gpurun_out/r05_traffic_trainstep[_yelp].json: the same three passes over the
    replayed reference-semantics TRAIN step (tools/train_profile.py), per kernel family of the backward (bench.py's train_roofline
    reads the dense edge backward's L2 hit rate from there).  Under --pmc every dispatch runs alone: these are the kernels' own
    figures, not those of two chains sharing the GPU."""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counters, workload, dtype, tag):
    """{counter: {kernel family: (mean value per launch, launches)}} of one rocprofv3 --pmc pass (counters: a name or a list)."""
    counters = [counters] if isinstance(counters, str) else list(counters)
    out = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s_%s" % (tag, dtype, counters[0]))
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, TMPDIR="/tmp")
    steps = ["--steps", "3", "--warmup", "1", "--passes-per-step", "4"] if workload not in ("synthetic1b", "synthetic200m") else \
            ["--steps", "2", "--warmup", "1", "--passes-per-step", "1"]
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py")] + steps + ["--no-cpu-baseline", "--no-train", "--no-probe", "--no-roof", "--workload", workload, "--dtype", dtype]
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tot = {c: collections.defaultdict(float) for c in counters}
    n = {c: collections.defaultdict(int) for c in counters}
    for root, _, files in os.walk(out):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    c = r["Counter_Name"]
                    if c not in tot:
                        continue
                    if ", 1>(" in r["Kernel_Name"] or ", 2>(" in r["Kernel_Name"]:
                        continue                            # (a measurement-only ROOF instantiation: never part of a pass)
                    for k in KERNELS:
                        if k + "<" in r["Kernel_Name"] or k + "(" in r["Kernel_Name"]:
                            tot[c][k] += float(r["Counter_Value"])
                            n[c][k] += 1
    subprocess.run(["rm", "-rf", out])
    res = {c: {k: (tot[c][k] / max(n[c][k], 1), n[c][k]) for k in KERNELS} for c in counters}
    return res[counters[0]] if len(counters) == 1 else res


KERNELS = ("prop_fwd_kernel", "prop_fwd_split_kernel", "transform_mfma_kernel", "transform_fwd_kernel")
ROUND = "r05"


def train_pass(counters, dtype, workload="gowalla"):
    """{counter: {kernel label: (mean per launch, launches)}} over the kernels of the replayed train step."""
    counters = [counters] if isinstance(counters, str) else list(counters)
    out = os.path.join(ROOT, "gpurun_out", "pmc_train_%s_%s" % (dtype, counters[0]))
    subprocess.run(["rm", "-rf", out])
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "tools", "train_profile.py"), "--steps", "6", "--dtype", dtype, "--workload", workload]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tot = {c: collections.defaultdict(float) for c in counters}
    n = {c: collections.defaultdict(int) for c in counters}

    def label(name):
        import re
        m = re.search(r"ngacf::(\w+)(<([^>]*)>)?", name)
        if not m:
            return None
        k, targs = m.group(1), (m.group(3) or "").replace(" ", "").split(",")
        if k == "prop_bwd_kernel":                         # <T, D, HEADS, DROP, SPARSE, FUSE>
            return "prop_bwd_kernel " + ("scored batch" if targs[4] == "true" else "dense")
        if k == "prop_fwd_kernel":
            return "prop_fwd_kernel " + ("scored batch" if targs[5] == "true" else "dense (train: out / agg / R written)")
        if k in ("prop_bwd_node_kernel", "transform_dw_mfma_kernel", "transform_dw_batch_kernel", "transform_bwd_reduce_kernel",
                 "transform_dw_batch_reduce_kernel", "transform_mfma_kernel", "adam_kernel", "transform_bwd_dw_kernel", "transform_bwd_dx_kernel",
                 "transform_f32_tiles_kernel", "transform_fwd_kernel"):
            return k
        return None
    for root, _, files in os.walk(out):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    c, lb = r["Counter_Name"], label(r["Kernel_Name"])
                    if c in tot and lb:
                        tot[c][lb] += float(r["Counter_Value"])
                        n[c][lb] += 1
    subprocess.run(["rm", "-rf", out])
    return {c: {k: (tot[c][k] / n[c][k], n[c][k]) for k in tot[c]} for c in counters}


def main_train(dtypes, workload="gowalla"):
    res = {"workload": workload, "what": "the kernels of one replayed reference-semantics train step (B = 2048, dropout 0.1, SPUIMultiGACF)",
           "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} -- python tools/train_profile.py --steps 6 "
                      "[--dtype fp32] [--workload yelp]  (tools/traffic.py trainstep; three separate passes; FETCH_SIZE x2 = gfx950 correction; under --pmc every "
                      "dispatch runs alone)", "by_dtype": {}}
    for dt in dtypes:
        fetch = train_pass("FETCH_SIZE", dt, workload)["FETCH_SIZE"]
        write = train_pass("WRITE_SIZE", dt, workload)["WRITE_SIZE"]
        hm = train_pass(["TCC_HIT_sum", "TCC_MISS_sum"], dt, workload)
        ks = {}
        for k in sorted(fetch):
            hit, miss = hm["TCC_HIT_sum"].get(k, (0.0, 0))[0], hm["TCC_MISS_sum"].get(k, (0.0, 0))[0]
            f0, w0 = fetch[k][0], write.get(k, (0.0, 0))[0]
            ks[k] = {"launches": fetch[k][1], "fetch_size_kb": round(f0, 1), "write_size_kb": round(w0, 1),
                     "traffic_bytes": int(round((2.0 * f0 + w0) * 1024)), "l2_hit_rate": round(hit / max(hit + miss, 1.0), 4),
                     "tcc_hit_per_launch": round(hit, 1), "tcc_miss_per_launch": round(miss, 1)}
            print(dt, k, ks[k], flush=True)
        res["by_dtype"][dt] = {"kernels": ks}
    with open(os.path.join(ROOT, "gpurun_out", "%s_traffic_trainstep%s.json" % (ROUND, "" if workload == "gowalla" else "_" + workload)), "w") as f:
        json.dump(res, f, indent=1)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
    dtypes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16", "fp32"]
    if wl == "trainstep":
        return main_train(dtypes, sys.argv[3] if len(sys.argv) > 3 else "gowalla")
    res = {"workload": wl, "stages": 3,
           "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} -- python bench.py --steps 3 --warmup 1 "
                      "--passes-per-step 4 --no-cpu-baseline --no-train --no-probe --no-roof --workload %s [--dtype fp32]  (tools/traffic.py; "
                      "three separate passes, the eval forward as benched: the fused traversals of a bf16 pass included)" % wl,
           "by_dtype": {}}
    for dt in dtypes:
        fetch = one_pass("FETCH_SIZE", wl, dt, wl)
        write = one_pass("WRITE_SIZE", wl, dt, wl)
        hm = one_pass(["TCC_HIT_sum", "TCC_MISS_sum"], wl, dt, wl)
        f0, nf = fetch["prop_fwd_kernel"]
        w0, nw = write["prop_fwd_kernel"]
        hit, miss = hm["TCC_HIT_sum"]["prop_fwd_kernel"][0], hm["TCC_MISS_sum"]["prop_fwd_kernel"][0]
        res["by_dtype"][dt] = {"kernel": "prop_fwd_kernel", "launches": [nf, nw], "fetch_size_kb": round(f0, 1),
                               "write_size_kb": round(w0, 1),
                               "l2_hit_rate": round(hit / max(hit + miss, 1.0), 4), "tcc_hit_per_launch": round(hit, 1), "tcc_miss_per_launch": round(miss, 1),
                               "traffic_bytes": int(round((2.0 * f0 + w0) * 1024)),
                               "note": "bytes crossing the L2<->fabric boundary per launch (Infinity-Cache hits included), "
                                       "avg over the launches of all stages; FETCH_SIZE x2 = gfx950 correction",
                               "other_kernels": {k: {"fetch_size_kb": round(fetch[k][0], 1), "write_size_kb": round(write[k][0], 1),
                                                     "launches": fetch[k][1]} for k in KERNELS[1:] if fetch[k][1]}}
        print(wl, dt, res["by_dtype"][dt], flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "%s_traffic_%s.json" % (ROUND, wl)), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
