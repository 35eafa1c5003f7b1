"""Stand-alone times of the backward's launches, separate against fused (ngacf_prop_bwd_dx), on the Gowalla-scale graph:
n calls captured in one hipGraph and replayed, so the figure is GPU time per call without host gaps.
    python tools/bwd_probe.py [workload] [--drop 0.1]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.sparse_probe import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="gowalla")
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--batch", type=int, default=2048)
    a = ap.parse_args()
    from ngacf_amd import data as D, ops
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    U, I, _ = D.SHAPES[a.workload]
    edge = D.workload_graph(a.workload, dev)
    edge = D.renumber(edge, *D.degree_order(edge, U, I))
    plan = build_plan(edge, U, I, device=dev)
    N, d, bf, p = U + I, 64, torch.bfloat16, a.drop
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    res = {"workload": a.workload, "drop": p, "N": N, "E": int(edge.shape[1])}
    W = rnd(2, d, d) * 0.2
    panel = ops.pack_panels([W], kinds=[1])[0]
    p_out, p_agg = (rnd(N, d) * 0.5).to(bf), (rnd(2, N, d) * 0.5).to(bf)
    p_R, p_S = rnd(2, N, 8).abs() + 0.1, rnd(N, 8) * 0.3
    for heads in (8, 1):
        X = (rnd(N, d) * 0.5).to(bf)
        avec = rnd(2, d) * 0.2
        H, S = ops.transform_fwd(X, W, avec, U, I, heads, h_dtype=bf, feat_drop_p=p, seed=3)
        for sparse in ((False,) if heads == 8 else (False, True)):
            rows = None
            if sparse:
                ub, ib = torch.randint(0, U, (a.batch,), device=dev, generator=g), torch.randint(0, I, (a.batch,), device=dev, generator=g)
                rows = ops.mark_rows(ub, ib, U, I, grad_d=d)
                rows.g32.copy_(rnd(N, d) * rows.mask()[:, None])
                gout = rows.g32
            else:
                gout = rnd(N, d).to(bf)
            out, agg, R = ops.prop_fwd(plan, H, S, S, avec, heads, save=True, needed=rows, drop_p=p, seed=5)
            kw = dict(drop_p=p, seed=5, needed=rows, gout_f32=sparse)
            tag = "h%d%s" % (heads, "_sparse" if sparse else "")
            r = {}
            r["node+edges"] = timed(lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, **kw))
            dH, dS = ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, **kw)
            if not sparse:
                Gn, T, P1 = torch.zeros(N, d, dtype=bf, device=dev), torch.zeros(N, heads, 2, device=dev), torch.zeros(N, heads, device=dev)
                r["edges (node=)"] = timed(lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, node=(Gn, T, P1), **kw))
            r["dW+dX (bf16 X)"] = timed(lambda: ops.transform_bwd(X, W, H, dH, dS, U, I, heads, p, 3))
            r["dW only"] = timed(lambda: ops.transform_bwd(X, W, H, dH, dS, U, I, heads, p, 3, part="dw"))
            r["dW+dX (fp32 X)"] = timed(lambda: ops.transform_bwd(X.float(), W, H, dH, dS, U, I, heads, p, 3))
            dx1 = ops.BwdDx(panel, torch.float32, p, 3)
            r["edges+dX fp32 (fuse 1)"] = timed(lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, dx=dx1, **kw))
            dx3 = ops.BwdDx(panel, bf, p, 3)
            r["edges+dX bf16 (fuse 1, later stages)"] = timed(lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, dx=dx3, **kw))
            res[tag] = {k: round(v, 1) for k, v in r.items()}
            print(tag, json.dumps(res[tag]), flush=True)
    G = torch.empty(N, d, device=dev)
    Gn, T, P1 = torch.empty(N, d, dtype=bf, device=dev), torch.empty(N, 8, 2, device=dev), torch.empty(N, 8, device=dev)
    dXb = rnd(N, d).to(bf)
    res["prev node step alone"] = round(timed(lambda: ops.prop_bwd_node(p_out, p_agg, p_R, dXb, p_S, G, Gn, T, P1, U, 8, (0, U, 0, I), False)), 1)
    print("prev node step alone", res["prev node step alone"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bwd_probe_%s.json" % a.workload), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
