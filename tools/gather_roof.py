#!/usr/bin/env python3
"""The measured ceiling of the traversal's access pattern: the traversal kernels against their own
stripped instantiations (NGACF_ROOF_ADD / NGACF_ROOF_XOR, include/ngacf.h) over the BENCHED plan -- the degree-renumbered
synthetic graph bench.py builds, the same work order, lane layout, index loads and row gathers.

    python tools/gather_roof.py [--workloads gowalla,yelp] [--dtypes bf16,fp32] [--launches 200]    -> one JSON line per case

Per (workload, dtype, heads):
  fwd:  ngacf_prop_fwd, eval mode, plain launch (the kernel of a pass's last stage; the fused launches add the next stage's
        transform tail)                                 vs. roof=1 (one add per gathered element) vs. roof=2 (one xor per dword)
  fwd_train: the train-mode forward (dropout, second plane, five stores) -- same gathers, so the same two roofs apply
  bwd:  ngacf_prop_bwd edge step alone (node=: step 1 done elsewhere), no dropout / with dropout   vs. its roof=1 / roof=2
        (Gn row + one (S, t) pair per head per edge)
Durations: dispatch-packet timestamps (ops.KernelTimer) of eager launches, mean over --launches after a warm-up.
`measure()` is what bench.py imports for roofline.measured_gather_roof_GBps."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _time(torch, ops, fn, tag, n, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    ms = kt.durations_ms(tag)
    if len(ms) != n:
        raise RuntimeError("timed %d launches, expected %d" % (len(ms), n))
    ms = sorted(ms)
    return 1e3 * sum(ms) / n, 1e3 * ms[n // 2]


def measure(torch, ops, plan, U, I, dtype, heads, n=200, what=("fwd", "bwd"), drop_p=0.1, seed=1):
    """{name: {"mean_us", "median_us"}} for the traversal kernels of one (plan, dtype, heads) and their roofs.  Inputs are random
    tables of the model's scale (N(0, 0.1) rows: exp() stays finite)."""
    dev, N, d = plan.device, U + I, 64
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    X = (rnd(N, d) * 0.1).to(dtype)
    W, avec = rnd(2, d, d) * 0.2, rnd(2, d) * 0.2
    H, S = ops.transform_fwd(X, W, avec, U, I, heads, h_dtype=dtype)
    out = torch.empty(N, d, dtype=dtype, device=dev)
    res = {}
    for _ in range(300):                                    # (clocks: the first timed kernel of a cold process reads 10-15 % slow)
        ops.prop_fwd(plan, H, S, S, avec, heads, out=out)
    torch.cuda.synchronize()

    def put(name, fn, tag):
        mean, med = _time(torch, ops, fn, tag, n)
        res[name] = {"mean_us": round(mean, 2), "median_us": round(med, 2)}
    T = ops.KernelTimer
    if "fwd" in what:
        put("fwd_eval", lambda: ops.prop_fwd(plan, H, S, S, avec, heads, out=out), T.TAG_PROP_FWD)
        put("fwd_roof_add", lambda: ops.prop_fwd(plan, H, S, S, avec, heads, out=out, roof=1), T.TAG_PROP_FWD)
        put("fwd_roof_xor", lambda: ops.prop_fwd(plan, H, S, S, avec, heads, out=out, roof=2), T.TAG_PROP_FWD)
    if "fwd_train" in what or "bwd" in what:
        agg = torch.empty(2, N, d, dtype=dtype, device=dev)
        R = torch.empty(2, N, heads, dtype=torch.float32, device=dev)
        fo, agg, R = ops.prop_fwd(plan, H, S, S, avec, heads, out=out, save=True, agg=agg, R=R, drop_p=drop_p, seed=5)
    if "fwd_train" in what:
        put("fwd_train", lambda: ops.prop_fwd(plan, H, S, S, avec, heads, out=out, save=True, agg=agg, R=R, drop_p=drop_p, seed=5), T.TAG_PROP_FWD)
    if "bwd" in what:
        gout = (rnd(N, d) * 0.1).to(dtype)
        ops.prop_bwd(plan, H, S, S, avec, fo, agg, R, gout, heads, drop_p=drop_p, seed=5)      # (fills nothing we keep: a warm-up of step 1 + 2)
        Gn = (rnd(N, d) * 0.1).to(dtype)
        ST, P1 = rnd(N, heads, 2) * 0.1, rnd(N, heads) * 0.1
        node = lambda: (Gn, ST, P1.clone())
        P1s = [P1.clone() for _ in range(4)]
        k = [0]

        def bwd(**kw):
            k[0] += 1
            return ops.prop_bwd(plan, H, S, S, avec, fo, agg, R, gout, heads, node=(Gn, ST, P1s[k[0] & 3]), **kw)
        put("bwd_edges", lambda: bwd(), T.TAG_PROP_BWD)
        put("bwd_edges_drop", lambda: bwd(drop_p=drop_p, seed=5), T.TAG_PROP_BWD)
        put("bwd_roof_add", lambda: bwd(roof=1), T.TAG_PROP_BWD)
        put("bwd_roof_xor", lambda: bwd(roof=2), T.TAG_PROP_BWD)
    return res


def bytes_per_launch(U, I, E, elem, heads, bwd):
    """forward 2E (4 + d T) + N (2 d T + 8); the split-sign edge backward gathers a row and a (S, t) pair per head:
    2E (4 + d T + 8 heads) + N (2 d T + 8)."""
    N, d = U + I, 64
    return 2 * E * (4 + d * elem + (8 * heads if bwd else 0)) + N * (2 * d * elem + 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="gowalla,yelp")
    ap.add_argument("--dtypes", default="bf16,fp32")
    ap.add_argument("--heads", default="8,1")
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--what", default="fwd,fwd_train,bwd")
    a = ap.parse_args()
    import torch
    from ngacf_amd import data as D, ops
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    for wl in a.workloads.split(","):
        U, I, _ = D.SHAPES[wl]
        edge = D.workload_graph(wl, dev)
        edge = D.renumber(edge, *D.degree_order(edge, U, I))         # bench.py's ingest (ingest_graph)
        E = int(edge.shape[1])
        plan = build_plan(edge, U, I, device=dev)
        for dt in a.dtypes.split(","):
            dtype = torch.bfloat16 if dt == "bf16" else torch.float32
            elem = 2 if dt == "bf16" else 4
            for heads in [int(h) for h in a.heads.split(",")]:
                r = measure(torch, ops, plan, U, I, dtype, heads, n=a.launches, what=tuple(a.what.split(",")))
                bf, bb = bytes_per_launch(U, I, E, elem, heads, False), bytes_per_launch(U, I, E, elem, heads, True)
                line = {"workload": wl, "U": U, "I": I, "E": E, "dtype": dt, "heads": heads, "row_bytes": 64 * elem,
                        "pair_bytes_per_edge": 8 * heads, "node_numbering": "degree", "launches": a.launches,
                        "bytes_fwd": bf, "bytes_bwd": bb, "us": r}
                for k, v in r.items():
                    b = bb if k.startswith("bwd") else bf
                    v["GBps"] = round(b / (v["mean_us"] * 1e-6) / 1e9, 1)
                if "fwd_eval" in r:
                    line["fwd_frac_of_roof_add"] = round(r["fwd_roof_add"]["mean_us"] / r["fwd_eval"]["mean_us"], 4)
                    line["fwd_frac_of_roof_xor"] = round(r["fwd_roof_xor"]["mean_us"] / r["fwd_eval"]["mean_us"], 4)
                if "fwd_train" in r and "fwd_roof_add" in r:
                    line["fwd_train_frac_of_roof_add"] = round(r["fwd_roof_add"]["mean_us"] / r["fwd_train"]["mean_us"], 4)
                if "bwd_edges" in r:
                    line["bwd_frac_of_roof_add"] = round(r["bwd_roof_add"]["mean_us"] / r["bwd_edges"]["mean_us"], 4)
                    line["bwd_frac_of_roof_xor"] = round(r["bwd_roof_xor"]["mean_us"] / r["bwd_edges"]["mean_us"], 4)
                    line["bwd_drop_frac_of_roof_add"] = round(r["bwd_roof_add"]["mean_us"] / r["bwd_edges_drop"]["mean_us"], 4)
                print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
