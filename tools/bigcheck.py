"""Full-size parity spot check: one propagation stage on a large synthetic graph, the kernel's output rows compared
with a direct float64 evaluation of the reference formula (SPUIGACF.py:356-397) on a random sample of destination
rows (hub rows included).  python tools/bigcheck.py --workload synthetic200m [--dtype bf16] [--rows 2000]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngacf_amd import data as D, ops  # noqa: E402
from ngacf_amd.plan import build_plan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synthetic200m")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rows", type=int, default=2000)
    ap.add_argument("--heads", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    U, I, E = D.SHAPES[a.workload]
    t0 = time.time()
    edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
    E = int(edge.shape[1])
    plan = build_plan(edge, U, I)
    del edge
    torch.cuda.synchronize()
    print("graph + plan: %.1f s, U=%d I=%d E=%d nseg=%d nshort=%d nsplit=%d" % (time.time() - t0, U, I, E, plan.nseg, plan.nshort, plan.nsplit), flush=True)
    N, d, heads = U + I, 64, a.heads
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    g = torch.Generator(device=dev).manual_seed(1)
    X = (torch.randn(N, d, device=dev, generator=g) * 0.3).to(dt)
    W = torch.randn(2, d, d, device=dev, generator=g) * 0.2
    avec = torch.randn(2, d, device=dev, generator=g) * 0.3
    H, S = ops.transform_fwd(X, W, avec, U, I, heads, dt)
    out = ops.prop_fwd(plan, H, S, S, avec, heads, 0.2, True)
    torch.cuda.synchronize()
    # sample: random rows of both types + the highest-degree rows
    deg = (plan.rowptr[1:] - plan.rowptr[:-1])
    top = torch.topk(deg, 20).indices
    rows = torch.cat([torch.randint(0, N, (a.rows,), device=dev, generator=g), top]).unique()
    hw = d // heads
    H64 = None
    worst, worst_rel = 0.0, 0.0
    Sd = S.double()
    for r in rows.tolist():
        b, e = int(plan.rowptr[r]), int(plan.rowptr[r + 1])
        h_own = H[r].double()
        if e == b:
            ref = torch.where(h_own > 0, h_own, torch.expm1(h_own))
        else:
            c = plan.col[b:e].long()
            s = Sd[r][None, :] + Sd[c]                                   # [deg, heads]
            w = torch.exp(-torch.where(s > 0, s, 0.2 * s))
            hc = H[c].double().view(-1, heads, hw)
            agg = (w[:, :, None] * hc).sum(0) / w.sum(0)[:, None]
            x = h_own + agg.reshape(d)
            ref = torch.where(x > 0, x, torch.expm1(x))
        err = float((out[r].double() - ref).abs().max())
        worst = max(worst, err)
        worst_rel = max(worst_rel, err / max(float(ref.abs().max()), 1e-6))
    tol = 2e-2 if a.dtype == "bf16" else 1e-4
    print("%s %s heads=%d: %d rows checked (max degree %d), max abs err %.3e, max rel err %.3e  (tolerance %.0e: %s)"
          % (a.workload, a.dtype, heads, rows.numel(), int(deg.max()), worst, worst_rel, tol, "OK" if worst_rel < tol else "FAIL"), flush=True)
    sys.exit(0 if worst_rel < tol else 1)


if __name__ == "__main__":
    main()
