"""Compute side of the node-partitioned forward on ONE GPU: the kernels rank r of a P-way partition would run
(row-window transform + pull traversal of its rows over the full gathered table), with the collectives skipped
(remote rows of H hold stale data: same access pattern, meaningless values).  Gives the measured term of the
scaling model of DESIGN.md section 6; the all-gather term is modelled from bytes / xGMI rate.

  python tools/shard_compute.py --workload gowalla --worlds 1,2,4,8 [--dtype bf16] [--stages 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngacf_amd import data as D  # noqa: E402
from ngacf_amd.dist import ShardedSPUIGACF  # noqa: E402


class NullComm:
    def __init__(self, comm):
        self.__dict__.update(comm.__dict__)

    def all_gather_rows(self, *fulls):
        pass

    def all_reduce_sum(self, *ts):
        return ts[0] if len(ts) == 1 else ts


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
    edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
    E = int(edge.shape[1])
    elem = 2 if a.dtype == "bf16" else 4
    print("workload %s U=%d I=%d E=%d dtype=%s stages=%d" % (a.workload, U, I, E, a.dtype, a.stages), flush=True)
    t1 = None
    for P in [int(x) for x in a.worlds.split(",")]:
        torch.manual_seed(0)
        m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world=P, rank=0, stages=a.stages).to(dev)
        m.comm = NullComm(m.comm)
        m.compute_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
        m.eval()
        with torch.no_grad():
            for _ in range(2):
                m.propagate_local(edge)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                m.propagate_local(edge)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        t1 = ms if P == 1 else t1
        p = m.part
        recv = (P - 1) / P * p.N * 64 * elem * a.stages                  # bytes a rank receives per forward
        ag_ms = recv / (a.xgmi_gbs * 1e9) * 1e3
        line = "P=%d rank0: compute %.3f ms/forward (eager launches), all-gather %.1f MB -> %.3f ms at %.0f GB/s" % (
            P, ms, recv / 1e6, ag_ms, a.xgmi_gbs)
        if t1:
            line += " | speedup vs P=1: serial %.2fx, overlapped %.2fx" % (t1 / (ms + ag_ms), t1 / max(ms, ag_ms))
        print(line, flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
