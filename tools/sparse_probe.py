"""Where does the time of the sparse last stage go?  One stage (forward, backward) on the Gowalla-scale graph, timed with
events over a loop, for batches of different size (0 rows ... every row) against the full stage.
    python tools/sparse_probe.py [workload] [dtype]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, n=20, reps=20):
    """us per call, GPU side: n calls captured in one hipGraph, replayed."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
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
    wl = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
    dt = torch.bfloat16 if (sys.argv[2] if len(sys.argv) > 2 else "bf16") == "bf16" else torch.float32
    from ngacf_amd import data as D, ops
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    U, I, E = D.SHAPES[wl]
    edge = D.workload_graph(wl, dev)
    plan = build_plan(edge, U, I, device=dev)
    N, d, heads = U + I, 64, 1
    g = torch.Generator(device=dev).manual_seed(1)
    X = (torch.randn(N, d, device=dev, generator=g) * 0.5).to(dt)
    W = torch.randn(2, d, d, device=dev, generator=g) * 0.2
    avec = torch.randn(2, d, device=dev, generator=g) * 0.2
    H, S = ops.transform_fwd(X, W, avec, U, I, heads, h_dtype=dt)
    gout_full = (torch.randn(N, d, device=dev, generator=g)).to(dt)
    res = {"workload": wl, "dtype": str(dt), "N": N, "E": int(edge.shape[1]), "nseg": plan.nseg, "nshort": plan.nshort, "nsplit": plan.nsplit}
    for B in (None, 0, 64, 512, 2048, 8192, 32768):
        needed = None
        if B is not None:
            ub = torch.randint(0, U, (max(B, 1),), device=dev, generator=g)[:B]
            ib = torch.randint(0, I, (max(B, 1),), device=dev, generator=g)[:B]
            needed = ops.mark_rows(ub, ib, U, I)
            gout = gout_full * needed.mask()[:, None]
        else:
            gout = gout_full
        for p in (0.0, 0.1):
            out, agg, R = ops.prop_fwd(plan, H, S, S, avec, heads, save=True, needed=needed, drop_p=p, seed=5)
            tf = timed(lambda: ops.prop_fwd(plan, H, S, S, avec, heads, save=True, needed=needed, drop_p=p, seed=5, out=out, agg=agg, R=R))
            tb = timed(lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, drop_p=p, seed=5, needed=needed))
            res["B=%s p=%.1f" % (B, p)] = {"fwd_us": round(tf, 1), "bwd_us": round(tb, 1)}
            print("B=%s p=%.1f fwd %.1f us  bwd (node + edges + split) %.1f us" % (B, p, tf, tb), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "sparse_probe_%s.json" % wl), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
