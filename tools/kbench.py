#!/usr/bin/env python3
"""Kernel micro-benchmarks on one GPU: times single C-ABI entry points with events,
interleaved over variants (same process, same data).  Usage: python tools/kbench.py [what] ..."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ngacf_amd import data as D, ops  # noqa: E402
from ngacf_amd.plan import build_plan  # noqa: E402


def timeit(fn, iters=30, warm=5):
    """Back-to-back launches, one event pair: (avg us per call, 0).  Host-bound when the host
    needs longer than the kernel -- read per-kernel durations from `rocprofv3 --kernel-trace
    --stats -- python tools/kbench.py ...` (tools/kprof.sh) instead."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters, 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--what", default="fwd,bwd,transform")
    ap.add_argument("--short", default="64")
    ap.add_argument("--seg", default="256")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    U, I, E = D.SHAPES[args.workload]
    edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
    N = U + I
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    g = torch.Generator(device=dev); g.manual_seed(1)
    print("graph built", flush=True)
    X = torch.randn(N, 64, device=dev, generator=g) * 0.1
    W = torch.randn(2, 64, 64, device=dev, generator=g) * 0.2
    avec = torch.randn(2, 64, device=dev, generator=g) * 0.2
    what = args.what.split(",")
    print("workload %s U=%d I=%d E=%d dtype=%s" % (args.workload, U, I, edge.shape[1], args.dtype))
    for short in [int(x) for x in args.short.split(",")]:
        for seg in [int(x) for x in args.seg.split(",")]:
            plan = build_plan(edge, U, I, seg_len=seg, short_max=short)
            tag = "short<=%d seg=%d (nshort=%d nseg=%d nsplit=%d)" % (short, seg, plan.nshort, plan.nseg, plan.nsplit)
            for heads in (8, 1):
                H, S = ops.transform_fwd(X.to(dt) if dt != torch.float32 else X, W, avec, U, I, heads, dt)
                if "transform" in what and short == int(args.short.split(",")[0]) and seg == int(args.seg.split(",")[0]):
                    Xin = X.to(dt)
                    print("  transform_fwd heads=%d x=%s: med %.1f us min %.1f" %
                          ((heads, args.dtype) + timeit(lambda: ops.transform_fwd(Xin, W, avec, U, I, heads, dt))))
                if "fwd" in what:
                    for fr in (True, False):
                        for p in (0.0, 0.1):
                            med, mn = timeit(lambda: ops.prop_fwd(plan, H, S, S, avec, heads, score_from_row=fr, drop_p=p, seed=3))
                            print("  prop_fwd  %s heads=%d from_row=%d drop=%.1f: med %.1f us min %.1f" % (tag, heads, fr, p, med, mn))
                if "bwd" in what:
                    out, agg, R = ops.prop_fwd(plan, H, S, S, avec, heads, save=True)
                    go = torch.randn(N, 64, device=dev, generator=g).to(dt)
                    for fr in (True, False):
                        for p in (0.0, 0.1):
                            med, mn = timeit(lambda: ops.prop_bwd(plan, H, S, S, avec, out, agg, R, go, heads, score_from_row=fr, drop_p=p, seed=3))
                            print("  prop_bwd  %s heads=%d from_row=%d drop=%.1f: med %.1f us min %.1f" % (tag, heads, fr, p, med, mn))
                    if "transform" in what:
                        dH = torch.randn(N, 64, device=dev, generator=g).to(dt); dS = torch.randn(N, heads, device=dev, generator=g)
                        Xin = X.to(dt)
                        print("  transform_bwd heads=%d: med %.1f us min %.1f" %
                              ((heads,) + timeit(lambda: ops.transform_bwd(Xin, W, H, dH, dS, U, I, heads))))


if __name__ == "__main__":
    main()
