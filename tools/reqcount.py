"""Cache-request counters of the two traversal kernels: how many L1 (TCP) accesses, L1->L2 read
requests and L2 hits / misses one directed edge costs in the forward (one gathered row) and in the backward (H row, Gn
row, t entry).  Separate rocprofv3 --pmc passes over tools/kbench.py (Gowalla scale), per kernel instantiation.
    python tools/reqcount.py [workload] [dtype]        (this process never touches the GPU itself)"""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TA_FLAT_READ_WAVEFRONTS_sum"],
          ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum"],
          ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_VMEM_RD"]]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
    dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    sys.path.insert(0, ROOT)
    from ngacf_amd import data as D
    U, I, E = D.SHAPES[wl]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for ctrs in PASSES:
        out = os.path.join(ROOT, "gpurun_out", "req_%s_%s_%s" % (wl, dt, ctrs[0]))
        subprocess.run(["rm", "-rf", out])
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", out, "-o", "p", "--",
               sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--workload", wl, "--dtype", dt, "--what", "fwd,bwd"]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for root, _, files in os.walk(out):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    for r in csv.DictReader(open(os.path.join(root, f))):
                        k = r["Kernel_Name"]
                        if "prop_fwd_kernel" in k or "prop_bwd_kernel" in k:
                            name = k[k.index("prop_"):k.index("(")]
                            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
                            cnt[name][r["Counter_Name"]] += 1
    res = {"workload": wl, "dtype": dt, "directed_edges_per_launch": 2 * E, "kernels": {}}
    for name in sorted(tot):
        per = {c: tot[name][c] / cnt[name][c] for c in tot[name]}
        row = {c: round(v / (2.0 * E), 3) for c, v in per.items()}
        if "TCC_HIT_sum" in per:
            row["L2_hit_rate"] = round(per["TCC_HIT_sum"] / max(per["TCC_HIT_sum"] + per["TCC_MISS_sum"], 1.0), 4)
        res["kernels"][name] = {"per_directed_edge": row, "launches": max(cnt[name].values())}
        print(name, json.dumps(res["kernels"][name]), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "r03_reqcount_%s_%s.json" % (wl, dt)), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
