#!/usr/bin/env python3
"""Soak of the ranking kernel's producer / consumer hand-offs (csrc/eval_topk.hip, LDS-shared form): many launches of the same
problem, every output word of every launch compared with the wave-private form's answer (integer table: exact, thousands of ties),
while a second stream keeps copying a large buffer (other traffic on the chip shifts the waves' timing against each other).
    python tools/topk_soak.py [--launches 3000] [--workload gowalla]      -> one JSON line"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngacf_amd import data as D, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--launches", type=int, default=3000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    U, I, E = D.SHAPES[a.workload]
    g = torch.Generator(device=dev).manual_seed(3)
    edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
    keep = torch.rand(edge.shape[1], device=dev, generator=g) < 0.8
    tu, ti = edge[0, keep], edge[1, keep]
    ptr = torch.zeros(U + 1, dtype=torch.long, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(tu, minlength=U), 0)
    items = ti.to(torch.int32).contiguous()
    pool = (torch.bincount(edge[1], minlength=I) > 0).to(torch.uint8)
    pool[torch.randperm(I, device=dev, generator=g)[:I // 50]] = 0
    users = torch.arange(U, device=dev)
    side = torch.cuda.Stream()
    big_a = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    big_b = torch.empty_like(big_a)
    out = {"workload": a.workload, "users": U, "items": I, "launches_per_config": a.launches, "configs": []}
    t0 = time.time()
    for dts, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        F = torch.randint(-2, 3, (U + I, 64), device=dev, generator=g).float().to(dt)
        os.environ["NGACF_TOPK_FORM"] = "wave"
        want, want_s = ops.eval_topk(F, users, U, I, pool, ptr, items, 20)
        for nw in (4, 8, 12):
            os.environ["NGACF_TOPK_FORM"] = "lds"
            os.environ["NGACF_TOPK_LDS_WAVES"] = str(nw)
            bad = 0
            for k in range(a.launches):
                if k % 8 == 0:                                   # bursts of other traffic beside every eighth launch
                    with torch.cuda.stream(side):
                        big_b.copy_(big_a)
                top, top_s = ops.eval_topk(F, users, U, I, pool, ptr, items, 20)
                bad += int(not (torch.equal(top, want) and torch.equal(top_s, want_s)))
                if k % 500 == 499:
                    print("[soak] %s lds%d: %d launches, %d mismatching, %.0f s" % (dts, nw, k + 1, bad, time.time() - t0), file=sys.stderr, flush=True)
            out["configs"].append({"dtype": dts, "consumer_waves": nw, "launches": a.launches, "mismatching_launches": bad})
    torch.cuda.synchronize()
    out["seconds"] = round(time.time() - t0, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
