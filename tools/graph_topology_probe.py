"""Does this HIP runtime instantiate a captured graph with a 'butterfly' (two branches, each waiting for the other's event)?
    python tools/graph_topology_probe.py butterfly|join"""
import sys
import torch

dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev)
b = torch.zeros(1 << 20, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def body(kind):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        a.add_(1.0)
        ea = sa.record_event()
    with torch.cuda.stream(sb):
        b.add_(2.0)
        eb = sb.record_event()
    if kind == "butterfly":
        with torch.cuda.stream(sb):
            sb.wait_event(ea)
            b.add_(a)
        with torch.cuda.stream(sa):
            sa.wait_event(eb)
            a.mul_(1.5)
    else:
        cur.wait_stream(sa)
        cur.wait_stream(sb)
        c = a + b
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            a.add_(c)
        with torch.cuda.stream(sb):
            b.add_(c)
    cur.wait_stream(sa)
    cur.wait_stream(sb)


kind = sys.argv[1]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body(kind)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body(kind)
print("captured", kind, flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed", kind, float(a[0]), float(b[0]), flush=True)
