"""Compute side of the node-partitioned forward on ONE GPU: the kernels rank 0 of a P-way partition runs (half-table
transforms + the two half traversals per stage over its sharded plan), with the collectives skipped (comm.dry_run: the
remote rows of H hold stale data -- same access pattern, meaningless values).  Gives the MEASURED term of the scaling
model of DESIGN.md section 6, per stage and per half; the all-gather term is modelled from bytes / an assumed xGMI rate.

  python tools/shard_compute.py --workload gowalla|synthetic200m|synthetic1b --worlds 1,2,4,8 [--dtype bf16] [--stages 3]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngacf_amd import data as D, ops  # noqa: E402
from ngacf_amd.dist import ShardedSPUIGACF, sharded_graph  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--stages", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--xgmi-gbs", type=float, default=300.0, help="assumed all-gather receive rate per GPU")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    U, I, E = D.SHAPES[a.workload]
    big = a.workload in D.BIG_WORKLOADS
    elem = 2 if a.dtype == "bf16" else 4
    edge = None if big else D.workload_graph(a.workload, dev)
    t1 = None
    for P in [int(x) for x in a.worlds.split(",")]:
        torch.manual_seed(0)
        if big:
            part, plan, E, _ = sharded_graph(a.workload, P, 0, dev)
            m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world=P, rank=0, stages=a.stages, partition=part).to(dev)
            m.attach_plan(plan)
            mask = torch.zeros(2, 1, dtype=torch.long, device=dev)
        else:
            E = int(edge.shape[1])
            m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world=P, rank=0, stages=a.stages, adj=edge).to(dev)
            mask = edge
            plan = m._plan(edge)
        m.comm.dry_run = True
        m.compute_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
        m.eval()
        with torch.no_grad():
            for _ in range(2):
                m.propagate_local(mask)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                m.propagate_local(mask)
            for _ in range(2):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            with ops.KernelTimer("ngacf_prop_fwd") as kt:
                for _ in range(a.reps):
                    m.propagate_local(mask)
                torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        durs = kt.durations_ms()
        n = len(durs) // a.reps                                            # 2 half launches per stage
        per = [sum(durs[j::n]) / a.reps for j in range(n)]
        t1 = ms if P == 1 else t1
        p = m.part
        recv_stage = (P - 1) / P * p.N * 64 * elem                         # bytes a rank receives per stage (both halves)
        ag_ms = recv_stage * a.stages / (a.xgmi_gbs * 1e9) * 1e3
        out = {"workload": a.workload, "E": E, "P": P, "dtype": a.dtype, "rank0_directed_edges": plan.n_edges_directed,
               "share_of_2E": round(plan.n_edges_directed / (2.0 * E), 4),
               "forward_ms_graph_replay_no_collectives": round(ms, 4), "prop_half_launch_us": [round(x * 1e3, 1) for x in per],
               "allgather_MB_received_per_forward": round(recv_stage * a.stages / 1e6, 1),
               "allgather_ms_at_%dGBps" % a.xgmi_gbs: round(ag_ms, 4)}
        if t1:
            exposed = recv_stage / 2 / (a.xgmi_gbs * 1e9) * 1e3            # only the first half-table gather of a forward is exposed
            out["speedup_vs_P1"] = {"serial": round(t1 / (ms + ag_ms), 2), "pipelined (max of the two + the exposed first gather)":
                                    round(t1 / (max(ms, ag_ms) + exposed), 2), "compute only": round(t1 / ms, 2)}
        print(json.dumps(out), flush=True)
        del m, plan
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
