"""cProfile of the eager sharded forward (host-side launch cost), P-way partition emulated on one GPU."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngacf_amd import data as D
from ngacf_amd.dist import ShardedSPUIGACF
from tools.shard_compute import NullComm
dev = torch.device("cuda:0")
U, I, E = D.SHAPES["gowalla"]
edge = D.synthetic_bipartite(U, I, E, seed=2019, device=dev)
m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world=8, rank=0, stages=3).to(dev)
m.comm = NullComm(m.comm)
m.compute_dtype = torch.bfloat16
m.eval()
with torch.no_grad():
    for _ in range(5):
        m.propagate_local(edge)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        m.propagate_local(edge)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr).sort_stats("cumulative")
st.print_stats(28)
