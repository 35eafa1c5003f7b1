#!/usr/bin/env python3
"""The evaluation's ranking kernel alone (ngacf_eval_topk), both forms, on a table at a named scale:
times the launch with events and checks the LDS-shared form against the wave-private one (integer table: exact, with ties).
Usage: python tools/topk_bench.py [--workload gowalla] [--dtype bf16,fp32] [--users N] [--reps 5]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngacf_amd import data as D, ops  # noqa: E402


def run(form, waves, F, users, U, I, pool, ptr, items, K, reps):
    os.environ["NGACF_TOPK_FORM"] = form
    if waves:
        os.environ["NGACF_TOPK_LDS_WAVES"] = str(waves)
    else:
        os.environ.pop("NGACF_TOPK_LDS_WAVES", None)
    top, sc = ops.eval_topk(F, users, U, I, pool, ptr, items, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        top, sc = ops.eval_topk(F, users, U, I, pool, ptr, items, K)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return top, sc, min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dtype", default="bf16,fp32")
    ap.add_argument("--users", type=int, default=0, help="rank only the first N users (0: all)")
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--forms", default="wave,lds8,lds12")
    ap.add_argument("--nopool", action="store_true", help="empty candidate pool: scoring only, no selection work")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    U, I, E = D.SHAPES[a.workload]
    K = 20
    g = torch.Generator(device=dev).manual_seed(7)
    edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
    keep = torch.rand(edge.shape[1], device=dev, generator=g) < 0.8          # "train" positives: 80 % of the edges
    tu, ti = edge[0, keep], edge[1, keep]
    ptr = torch.zeros(U + 1, dtype=torch.long, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(tu, minlength=U), 0)
    items = ti.to(torch.int32).contiguous()                                     # edge list is sorted by (user, item)
    pool = torch.ones(I, dtype=torch.uint8, device=dev)
    pool[torch.bincount(edge[1], minlength=I) == 0] = 0
    if a.nopool:
        pool.zero_()
    users = torch.arange(a.users or U, device=dev)
    for dts in a.dtype.split(","):
        dt = torch.bfloat16 if dts == "bf16" else torch.float32
        for table in ("integer", "normal"):
            if table == "integer":
                F = torch.randint(-2, 3, (U + I, a.d), device=dev, generator=g).float().to(dt)
            else:
                F = (torch.randn(U + I, a.d, device=dev, generator=g) * 0.1).to(dt)
            ref = None
            for form in a.forms.split(","):
                top, sc, ms = run("wave" if form == "wave" else "lds", 0 if form == "wave" else int(form[3:]), F, users, U, I, pool, ptr, items, K, a.reps)
                rec = {"workload": a.workload, "users": int(users.numel()), "items": I, "d": a.d, "dtype": dts, "table": table, "form": form,
                       "ms": round(ms, 4)}
                if ref is None:
                    ref = (top, sc)
                else:
                    rec["items_equal_first_form"] = bool(torch.equal(top, ref[0]))
                    rec["items_differ"] = int((top != ref[0]).sum())
                    rec["scores_max_abs_diff"] = float((sc - ref[1]).abs().max())
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
