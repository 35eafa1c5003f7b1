"""Two chains share the GPU in a train step.  What do two copies of one backward launch cost when they run side by side (two
streams of one hipGraph), against one after the other?  Separate launches against the fused ones.
    python tools/pair_probe.py [--drop 0.1]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed_pair(fa, fb, n=10, reps=20, parallel=True):
    """us per PAIR of calls: n x (fa on stream 1 | fb on stream 2) in one graph"""
    fa(); fb()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if parallel:
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                for _ in range(n):
                    fa()
            with torch.cuda.stream(s2):
                for _ in range(n):
                    fb()
            cur.wait_stream(s1); cur.wait_stream(s2)
        else:
            for _ in range(n):
                fa(); fb()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (n * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--drop", type=float, default=0.1)
    a = ap.parse_args()
    from ngacf_amd import data as D, ops
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    U, I, _ = D.SHAPES["gowalla"]
    edge = D.workload_graph("gowalla", dev)
    edge = D.renumber(edge, *D.degree_order(edge, U, I))
    plan = build_plan(edge, U, I, device=dev)
    N, d, bf, p, heads = U + I, 64, torch.bfloat16, a.drop, 8
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    W = rnd(2, d, d) * 0.2
    panel = ops.pack_panels([W], kinds=[1])[0]

    def chain():
        X = (rnd(N, d) * 0.5).to(bf)
        avec = rnd(2, d) * 0.2
        H, S = ops.transform_fwd(X, W, avec, U, I, heads, h_dtype=bf, feat_drop_p=p, seed=3)
        gout = rnd(N, d).to(bf)
        out, agg, R = ops.prop_fwd(plan, H, S, S, avec, heads, save=True, drop_p=p, seed=5)
        p_out, p_agg, p_R, p_S = (rnd(N, d) * 0.5).to(bf), (rnd(2, N, d) * 0.5).to(bf), rnd(2, N, 8).abs() + 0.1, rnd(N, 8) * 0.3
        Gn, T, P1 = torch.zeros(N, d, dtype=bf, device=dev), torch.zeros(N, heads, 2, device=dev), torch.zeros(N, heads, device=dev)
        G = torch.empty(N, d, device=dev)
        kw = dict(drop_p=p, seed=5)
        dH, dS = ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, **kw)
        c = {}
        c["fwd (train)"] = lambda: ops.prop_fwd(plan, H, S, S, avec, heads, save=True, drop_p=p, seed=5, out=out, agg=agg, R=R)
        c["edges"] = lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, node=(Gn, T, P1), **kw)
        c["node"] = lambda: ops.prop_bwd_node(p_out, p_agg, p_R, gout, p_S, G, Gn, T, P1, U, 8, (0, U, 0, I), False)
        c["dW+dX"] = lambda: ops.transform_bwd(X, W, H, dH, dS, U, I, heads, p, 3)
        c["dW"] = lambda: ops.transform_bwd(X, W, H, dH, dS, U, I, heads, p, 3, part="dw")
        dx1, dx3 = ops.BwdDx(panel, torch.float32, p, 3), ops.BwdDx(panel, bf, p, 3)
        c["edges+dX f32"] = lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, node=(Gn, T, P1), dx=dx1, **kw)
        c["edges+dX bf16"] = lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, node=(Gn, T, P1), dx=dx3, **kw)
        c["separate: node, edges, dW+dX"] = lambda: (c["node"](), c["edges"](), c["dW+dX"]())
        c["fused 3: node, edges+dX, dW"] = lambda: (c["node"](), c["edges+dX bf16"](), c["dW"]())
        return c

    A, B = chain(), chain()
    res = {}
    for k in A:
        ser = timed_pair(A[k], B[k], parallel=False)
        par = timed_pair(A[k], B[k], parallel=True)
        res[k] = {"one_after_the_other_us": round(ser, 1), "side_by_side_us": round(par, 1)}
        print("%-32s pair: one after the other %6.1f us   side by side %6.1f us" % (k, ser, par), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pair_probe.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
