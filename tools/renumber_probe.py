"""Which node numbering does the traversal want?  The same graph under the data set's ids, the ingest's degree order, two
locality orders beyond it (users by their hottest item + items by their heaviest user; reverse Cuthill-McKee) and a random
relabelling (control): eval forward of the plain single-GPU model, replayed as one hipGraph; per-stage kernel time from the
dispatch timestamps; with --hitrate also the traversal kernel's L2 hit rate (one rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum child
run per numbering).  Workloads: the generator's graphs and the REAL interaction lists the reference ships (gowalla_real,
yelp_real: data.real_graph).
  python tools/renumber_probe.py --workload gowalla|yelp|synthetic200m|gowalla_real|yelp_real [--dtype bf16|fp32] [--hitrate]"""
import argparse
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORDERS = ("dataset ids", "degree order", "cluster order", "rcm order", "random ids")


def graph_under(D, torch, workload, order, dev):
    if workload in D.REAL_WORKLOADS:
        edge, U, I = D.real_graph(workload, dev)
    else:
        U, I, _ = D.SHAPES[workload]
        edge = D.workload_graph(workload, dev)
    if order == "dataset ids":
        return edge, U, I
    if order == "random ids":
        g = torch.Generator(device=dev).manual_seed(11)
        perm = (torch.randperm(U, device=dev, generator=g), torch.randperm(I, device=dev, generator=g))
    else:
        perm = {"degree order": D.degree_order, "cluster order": D.cluster_order, "rcm order": D.rcm_order}[order](edge, U, I)
    return D.renumber(edge, *perm), U, I


def hit_rate(workload, dtype, order):
    """L2 hit rate of the prop_fwd_kernel launches of a child run under rocprofv3 --pmc (the parent never touches the GPU here)"""
    out = os.path.join(ROOT, "gpurun_out", "pmc_renumber")
    subprocess.run(["rm", "-rf", out])
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", "TCC_HIT_sum", "TCC_MISS_sum", "--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.abspath(__file__), "--workload", workload, "--dtype", dtype, "--order", order, "--reps", "3", "--child"]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tot = {"TCC_HIT_sum": 0.0, "TCC_MISS_sum": 0.0}
    for root, _, files in os.walk(out):
        for f in files:
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(root, f))):
                    if "prop_fwd_kernel<" in r["Kernel_Name"] and r["Counter_Name"] in tot:
                        tot[r["Counter_Name"]] += float(r["Counter_Value"])
    subprocess.run(["rm", "-rf", out])
    return round(tot["TCC_HIT_sum"] / max(tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"], 1.0), 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--order", default=None, help="one numbering only")
    ap.add_argument("--hitrate", action="store_true")
    ap.add_argument("--child", action="store_true", help="(internal: the run under the counters -- a few eager passes, no output)")
    a = ap.parse_args()
    if a.hitrate and not a.child:
        rates = {o: hit_rate(a.workload, a.dtype, o) for o in ([a.order] if a.order else ORDERS)}
    import torch
    import ngacf_amd
    from ngacf_amd import data as D, ops
    from ngacf_amd.graphs import GraphedPropagate
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    for name in ([a.order] if a.order else ORDERS):
        e, U, I = graph_under(D, torch, a.workload, name, dev)
        E = int(e.shape[1])
        reps = a.reps if a.workload not in D.BIG_WORKLOADS else 10
        torch.manual_seed(0)
        m = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev).eval()
        ngacf_amd.set_compute_dtype(m, torch.bfloat16 if a.dtype == "bf16" else torch.float32)
        with torch.no_grad():
            if a.child:
                for _ in range(reps):
                    m.propagate(e)
                torch.cuda.synchronize()
                continue
            run = GraphedPropagate(m, e)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(reps):
                run()
            t1.record()
            torch.cuda.synchronize()
            with ops.KernelTimer() as kt:
                for _ in range(10):
                    m.propagate(e)
                torch.cuda.synchronize()
        d = kt.durations_ms()
        L = 3
        per = [round(1e3 * sum(d[s::L]) / (len(d) // L), 2) for s in range(L)]
        ms = t0.elapsed_time(t1) / reps
        line = {"workload": a.workload, "dtype": a.dtype, "numbering": name, "ms_per_pass": round(ms, 5),
                "G_edges_per_s": round(3 * 2 * E / (ms * 1e-3) / 1e9, 2), "per_stage_us": per}
        if a.hitrate:
            line["prop_fwd_l2_hit_rate"] = rates[name]
        if name == "dataset ids":                       # the degree law of the graph, once
            plan = build_plan(e, U, I, device=dev)
            du, di = torch.bincount(e[0], minlength=U), torch.bincount(e[1], minlength=I)
            line["graph"] = {"U": U, "I": I, "E": E, "max_degree_user": int(du.max()), "max_degree_item": int(di.max()),
                             "isolated_items": int((di == 0).sum()), "rows_over_64_edges": int((du > 64).sum() + (di > 64).sum()),
                             "hub_rows_over_256_edges": int(plan.nsplit), "hub_slices": int(plan.nparts), "long_segments": int(plan.nseg),
                             "short_rows": int(plan.nshort)}
        print(json.dumps(line), flush=True)
        del m


if __name__ == "__main__":
    main()
