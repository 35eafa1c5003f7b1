"""Time the all-item evaluation (train_eval.eval_neg_all) on a synthetic graph at a named scale.
Usage: python tools/eval_bench.py [--workload gowalla] [--dtype bf16] [--reps 3]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngacf_amd as ng  # noqa: E402
from ngacf_amd import data as D  # noqa: E402
from ngacf_amd.train_eval import Interactions, eval_neg_all  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--user-batch", type=int, default=1 << 20)
    ap.add_argument("--epochs", type=int, default=0, help="also time this many training epochs (train_bpr(graph=True), B = 2048)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    U, I, E = D.SHAPES[a.workload]
    edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    te = torch.rand(edge.shape[1], device=dev, generator=g) < 0.2
    first = torch.ones(edge.shape[1], dtype=torch.bool, device=dev)
    first[1:] = edge[0, 1:] != edge[0, :-1]
    te &= ~first                                                   # every user keeps a train interaction
    tr = ~te
    data = Interactions(edge[0, tr].cpu().numpy(), edge[1, tr].cpu().numpy(), edge[0, te].cpu().numpy(), edge[1, te].cpu().numpy(), U, I, dev,
                        renumber="degree")                      # the product's ingest (run_Gowalla.py --renumber degree)
    adj = data.adjacency()
    torch.manual_seed(0)
    m = ng.SPUIMultiGACF(U, I, 64, [64, 64], 0.1).to(dev)
    ng.set_compute_dtype(m, torch.bfloat16 if a.dtype == "bf16" else torch.float32)
    if a.epochs:
        from ngacf_amd.optim import Adam
        from ngacf_amd.train_eval import train_bpr
        opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-5)
        gen = torch.Generator(device=dev).manual_seed(3)
        loss = train_bpr(m, 2048, data, adj, opt, ng.BPRLoss(), gen, graph=True)      # capture + first epoch
        torch.cuda.synchronize()
        for _ in range(a.epochs):
            t0 = time.perf_counter()
            loss = train_bpr(m, 2048, data, adj, opt, ng.BPRLoss(), gen, graph=True)
            torch.cuda.synchronize()
            print("train epoch %s: %d pairs, %d steps of 2048: %.3f s  loss %.6f" % (a.workload, data.train_u.numel(), data.train_u.numel() // 2048,
                                                                                  time.perf_counter() - t0, float(loss)), flush=True)
    res = eval_neg_all(m, 2048, data, adj, user_batch=a.user_batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        res = eval_neg_all(m, 2048, data, adj, user_batch=a.user_batch)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print("eval_neg_all %s U=%d I=%d test=%d: %.1f ms (min of %d)  recall@20 %.4f ndcg@20 %.4f"
          % (a.workload, U, I, int(te.sum()), 1e3 * min(ts), a.reps, res["recall"][3], res["ndcg"][3]))


if __name__ == "__main__":
    main()
