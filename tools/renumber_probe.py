"""Does a degree-ordered node numbering speed the traversal up?  The same graph (bench workload) under its own ids and with users
and items renumbered by degree, descending (hot rows contiguous): eval forward of the plain single-GPU model, replayed as one
hipGraph; per-stage kernel time from the dispatch timestamps.
  python tools/renumber_probe.py --workload gowalla|yelp|synthetic200m [--dtype bf16|fp32]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngacf_amd  # noqa: E402
from ngacf_amd import data as D, ops  # noqa: E402
from ngacf_amd.graphs import GraphedPropagate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    U, I, _ = D.SHAPES[a.workload]
    edge = D.workload_graph(a.workload, dev)
    E = int(edge.shape[1])
    du = torch.bincount(edge[0], minlength=U)
    di = torch.bincount(edge[1], minlength=I)
    orders = {"dataset ids": None,
              "degree order": (torch.argsort(torch.argsort(du, descending=True, stable=True)),
                               torch.argsort(torch.argsort(di, descending=True, stable=True))),
              "random ids": (torch.randperm(U, device=dev), torch.randperm(I, device=dev))}
    reps = a.reps if a.workload not in D.BIG_WORKLOADS else 10
    for name, perm in orders.items():
        e = edge if perm is None else torch.stack([perm[0][edge[0]], perm[1][edge[1]]])
        torch.manual_seed(0)
        m = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev).eval()
        ngacf_amd.set_compute_dtype(m, torch.bfloat16 if a.dtype == "bf16" else torch.float32)
        with torch.no_grad():
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
        print(json.dumps({"workload": a.workload, "dtype": a.dtype, "numbering": name, "ms_per_pass": round(ms, 5),
                          "G_edges_per_s": round(3 * 2 * E / (ms * 1e-3) / 1e9, 2), "per_stage_us": per}), flush=True)
        del m, run


if __name__ == "__main__":
    main()
