"""Soak test of the fence-free in-kernel hub-row finish (propagate.hip / propagate_bwd.hip: write-through `sc1` stores of the
slices' partial sums -> s_waitcnt vmcnt(0) -> relaxed agent-scope ticket -> agent-scope atomic (`sc1`) loads by the row's
last-arriving slice).  VERDICT r2 item 7: every train step depends on a protocol the guide calls "not an architectural
guarantee", so it is exercised the way MI355X_MICROARCH.md asks ("under UNEVEN load, consumer L1-warm, checking every word"):

  * Gowalla-scale graph, bf16 and fp32 tables, the forward (8-head, train mode with dropout, agg/R saved) and the backward
    traversal, >= `--launches` launches in all, the two kinds on TWO STREAMS at once (the two chains of a graphed train step);
  * K different input tables in rotation, so consecutive launches leave DIFFERENT partial sums in the same slots: a stale read
    of the previous launch's sums cannot equal the fresh ones;
  * the partial-sum slots are POISONED with NaN by a plain fill kernel in front of every launch: the fill leaves those lines
    in the XCDs' L2s; a last-arriver whose `sc1` load were served by a stale L2/L1 line would add a NaN into its row;
  * every output word of every launch is compared, on the device, with the two-kernel path's result (NGACF_INKERNEL_HUB=0:
    the partial sums cross a kernel boundary) -- BITWISE; mismatching launches are counted, nothing is averaged away;
  * every few launches a streaming kernel (a 64 MB reduction) runs beside them: uneven load, L2 contents churned.

  python tools/hub_soak.py [--launches 120000] [--seg-len 256]        -> one JSON line (mismatches must be 0)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=120000, help="prop_fwd + prop_bwd launches in all (10 000 train steps = 120 000)")
    ap.add_argument("--seg-len", type=int, default=256)
    ap.add_argument("--tables", type=int, default=3)
    ap.add_argument("--workload", default="gowalla")
    a = ap.parse_args(argv)
    import torch
    from ngacf_amd import data as D, ops
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    U, I, E = D.SHAPES[a.workload]
    edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
    N = U + I
    plan = build_plan(edge, U, I, seg_len=a.seg_len)
    assert plan.nsplit > 0 and plan.nparts <= ops.INKERNEL_MAX_PARTS and ops.INKERNEL_HUB_FINISH
    g = torch.Generator(device=dev).manual_seed(3)
    W = torch.randn(2, 64, 64, device=dev, generator=g) * 0.2
    avec = torch.randn(2, 64, device=dev, generator=g) * 0.2
    t0 = time.time()
    report = {"workload": a.workload, "E": int(edge.shape[1]), "hub_rows": plan.nsplit, "slices": plan.nparts, "seg_len": a.seg_len,
              "tables_in_rotation": a.tables}
    total_launches = total_bad = 0
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for dt in (torch.bfloat16, torch.float32):
        heads, p, seed = 8, 0.1, 12345
        cases = []
        ops.INKERNEL_HUB_FINISH = False                     # the reference results: a SECOND KERNEL adds the partial sums
        for k in range(a.tables):
            X = (torch.randn(N, 64, device=dev, generator=g) * (0.5 + 0.25 * k)).to(dt)
            H, S = ops.transform_fwd(X, W, avec, U, I, heads, dt)
            gout = torch.randn(N, 64, device=dev, generator=g).to(dt)
            fwd = ops.prop_fwd(plan, H, S, S, avec, heads, drop_p=p, seed=seed + k, save=True)
            bwd = ops.prop_bwd(plan, H, S, S, avec, *fwd, gout, heads, drop_p=p, seed=seed + k)
            cases.append((H, S, gout, fwd, bwd))
        ops.INKERNEL_HUB_FINISH = True
        torch.cuda.synchronize()
        bad = torch.zeros(2, dtype=torch.int64, device=dev)          # [forward launches that differ, backward launches that differ]
        nan = float("nan")
        # (a forward that saves keeps two planes of partial sums per slot: ops.prop_fwd)
        scr = {s_: (torch.empty(max(plan.nparts, 1), k_ * 64, device=dev), torch.empty(max(plan.nparts, 1), k_ * heads, device=dev))
               for s_, k_ in (("f", 2), ("b", 1))}
        noise = torch.randn(16_000_000, device=dev)
        per_dt = a.launches // 2
        it = 0
        n_f = n_b = 0
        while n_f + n_b < per_dt:
            H, S, gout, fwd, bwd = cases[it % a.tables]
            Hb, Sb, goutb, fwdb, bwdb = cases[(it + 1) % a.tables]
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur)
            sb.wait_stream(cur)
            with torch.cuda.stream(sa):                     # chain A: a forward launch
                scr["f"][0].fill_(nan); scr["f"][1].fill_(nan)
                out = ops.prop_fwd(plan, H, S, S, avec, heads, drop_p=p, seed=seed + it % a.tables, save=True, scratch=scr["f"])
                d = sum((x.view(torch.int32) != y.view(torch.int32)).any().long() for x, y in zip(out, fwd))
                bad[0] += (d > 0).long()
            with torch.cuda.stream(sb):                     # chain B: a backward launch of ANOTHER table, at the same time
                scr["b"][0].fill_(nan); scr["b"][1].fill_(nan)
                res = ops.prop_bwd(plan, Hb, Sb, Sb, avec, *fwdb, goutb, heads, drop_p=p, seed=seed + (it + 1) % a.tables, scratch=scr["b"])
                d = sum((x.view(torch.int32) != y.view(torch.int32)).any().long() for x, y in zip(res, bwdb))
                bad[1] += (d > 0).long()
                if it % 5 == 0:
                    noise.sum()                             # a streaming kernel beside chain A: uneven load, L2 churn
            cur.wait_stream(sa)
            cur.wait_stream(sb)
            n_f += 1
            n_b += 1
            it += 1
            if it % 5000 == 0:
                torch.cuda.synchronize()
                print("[soak] %s: %d launches, mismatching launches so far fwd %d bwd %d (%.0f s)"
                      % (str(dt).split(".")[1], n_f + n_b, int(bad[0]), int(bad[1]), time.time() - t0), file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        tick = sum(int(t.abs().sum()) for t in plan._tickets.values())
        report[str(dt).split(".")[1]] = {"fwd_launches": n_f, "bwd_launches": n_b, "fwd_mismatching_launches": int(bad[0]),
                                         "bwd_mismatching_launches": int(bad[1]), "tickets_left_nonzero": tick}
        total_launches += n_f + n_b
        total_bad += int(bad[0]) + int(bad[1]) + tick
        del cases, scr, noise
    report.update(launches=total_launches, mismatches=total_bad, seconds=round(time.time() - t0, 1),
                  compared="every output word of every launch (out, agg, R / dH, dS), bitwise, against the two-kernel path; partial-sum "
                           "slots NaN-filled before every launch; forward and backward launches overlapped on two streams")
    print(json.dumps(report), flush=True)
    return 0 if total_bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
