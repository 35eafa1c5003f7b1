"""us per launch of ngacf_score_bwd (deterministic, no atomics) at B = 2048, Gowalla-sized table; and its equality with a float64
index_add statement.  python tools/score_bwd_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ngacf_amd import ops
dev = torch.device("cuda:0")
U, I, B, d = 29858, 40981, 2048, 64
g = torch.Generator(device=dev).manual_seed(1)
for dt in (torch.bfloat16, torch.float32):
    F = torch.randn(U + I, d, device=dev, generator=g).to(dt)
    u = torch.randint(0, U, (B,), device=dev, generator=g); it = torch.randint(0, I, (B,), device=dev, generator=g)
    if os.environ.get("DUPS") == "1":
        u[:64] = u[64:128]; it[100:140] = it[300:340]                # a run of repeated users and items (worst case: eight owners' walks in one wave)
    gs = torch.randn(B, device=dev, generator=g)
    out = ops.score_bwd(F, u, it, gs, U)
    want = torch.zeros(U + I, d, dtype=torch.float64, device=dev)
    want.index_add_(0, u, gs.double()[:, None] * F[it + U].double()); want.index_add_(0, it + U, gs.double()[:, None] * F[u].double())
    err = float((out.double() - want).abs().max() / want.abs().max())
    into = torch.zeros(U + I, d, device=dev)
    for _ in range(20): ops.score_bwd(F, u, it, gs, U, into=into)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): ops.score_bwd(F, u, it, gs, U, into=into)
    b.record(); torch.cuda.synchronize()
    print(dt, "us per launch (incl. host gap) %.2f" % (a.elapsed_time(b) * 1e3 / 200), "max rel err vs float64 %.2e" % err, flush=True)
