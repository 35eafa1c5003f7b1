"""Time ngacf_eval_topk alone: all users of a named scale against all items, random table.
   python tools/topk_bench.py [--workload gowalla] [--dtype fp32] [--nopool]   (--nopool: empty candidate pool =
   scoring only, no selection work)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngacf_amd import data as D, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--nopool", action="store_true")
    ap.add_argument("--scale", type=float, default=0.01)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    U, I, _ = D.SHAPES[a.workload]
    g = torch.Generator(device=dev).manual_seed(0)
    F = torch.randn(U + I, a.d, device=dev, generator=g) * a.scale
    F = F.to(torch.bfloat16) if a.dtype == "bf16" else F
    pool = torch.zeros(I, dtype=torch.uint8, device=dev) if a.nopool else torch.ones(I, dtype=torch.uint8, device=dev)
    deg = torch.full((U,), 27, device=dev)
    ptr = torch.zeros(U + 1, dtype=torch.long, device=dev)
    ptr[1:] = torch.cumsum(deg, 0)
    items = torch.sort(torch.randint(0, I, (U, 27), device=dev, generator=g), 1).values.reshape(-1).to(torch.int32)
    users = torch.arange(U, device=dev)
    for _ in range(2):
        ops.eval_topk(F, users, U, I, pool, ptr, items, 20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.eval_topk(F, users, U, I, pool, ptr, items, 20)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("eval_topk %s %s d=%d nopool=%d: %.3f ms  (%.1f TFLOP/s)" % (a.workload, a.dtype, a.d, a.nopool, ms, 2.0 * U * I * a.d / ms / 1e9))


if __name__ == "__main__":
    main()
