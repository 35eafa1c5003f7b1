"""Same seeds, same batch: the parallel-branch graphed step must follow the sequential graphed step (differences only
from the float-atomic order in score_bwd)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngacf_amd as ng
from ngacf_amd import data as D
from ngacf_amd.graphs import GraphedBPRStep
dev = torch.device("cuda:0")
U, I, E = D.SHAPES["gowalla"]
edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
g = torch.Generator(device=dev); g.manual_seed(7)
B = 2048
upn = torch.stack([torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I, (B,), device=dev, generator=g),
                   torch.randint(0, I, (B,), device=dev, generator=g)])
res = {}
for conc in (False, True):
    torch.manual_seed(2019)
    m = ng.SPUIMultiGACF(U, I, 64, [64, 64], 0.1).to(dev)
    ng.set_compute_dtype(m, torch.bfloat16)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True, fused=True)
    torch.manual_seed(11)
    gs = GraphedBPRStep(m, ng.BPRLoss(), opt, edge, B, concurrent_forwards=conc)
    losses = [float(gs(upn)) for _ in range(40)]
    res[conc] = (losses, m.uEmbd.weight.detach().float().clone())
a, b = res[False], res[True]
rel = max(abs(x - y) / max(abs(x), 1e-12) for x, y in zip(a[0], b[0]))
print("loss[0..3] seq", a[0][:3], "par", b[0][:3])
print("max rel loss diff over 40 steps %.3e; weight diff %.3e (max |w| %.3e)" % (rel, float((a[1] - b[1]).abs().max()), float(a[1].abs().max())))
assert rel < 1e-3
