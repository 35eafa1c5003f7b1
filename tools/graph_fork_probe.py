"""How late does the second branch of a fork behind a join start in a replayed hipGraph?  Two branches of n spinning kernels each
(torch.cuda._sleep: one thread, no bandwidth) behind a join kernel; graph time against n * t tells how many kernels of the first
branch the second one's start waits for.
    python tools/graph_fork_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def calibrate():
    cyc = 1_000_000
    torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(cyc)
    e1.record()
    torch.cuda.synchronize()
    return cyc / (e0.elapsed_time(e1) * 1e3)          # cycles per us


def body(n, cyc, first, head=0):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    for s in (sa, sb):                                  # two branches in front of the join, like the step's forwards
        with torch.cuda.stream(s):
            torch.cuda._sleep(cyc)
    cur.wait_stream(sa)
    cur.wait_stream(sb)
    torch.cuda._sleep(cyc // 4)                         # the join kernel
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    for s in ((sa, sb) if first == "a" else (sb, sa)):
        with torch.cuda.stream(s):
            for k in range(n):
                torch.cuda._sleep(cyc // 8 if k < head else cyc)
    cur.wait_stream(sa)
    cur.wait_stream(sb)


def body_inline(n, cyc, first, head=0):
    """the same work with branch a ON the capture stream and the join kernel on branch b's stream: b waits for a's first part,
    runs the join and goes on; a waits for the join only"""
    cur = torch.cuda.current_stream()
    sb.wait_stream(cur)
    torch.cuda._sleep(cyc)                              # a, first part (on cur)
    ea = cur.record_event()
    with torch.cuda.stream(sb):
        torch.cuda._sleep(cyc)                          # b, first part
        sb.wait_event(ea)
        torch.cuda._sleep(cyc // 4)                     # the join kernel
        ej = sb.record_event()
        for k in range(n):
            torch.cuda._sleep(cyc // 8 if k < head else cyc)
    cur.wait_event(ej)
    for k in range(n):
        torch.cuda._sleep(cyc // 8 if k < head else cyc)
    cur.wait_stream(sb)


def timed(n, cyc, first, head=0):
    global body
    if first == "inline":
        keep, body = body, body_inline
        try:
            return timed(n, cyc, "a", head)
        finally:
            body = keep
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body(n, cyc, first, head)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(n, cyc, first, head)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e6


cpu = calibrate()
t = 30.0
cyc = int(t * cpu)
print("cycles per us %.1f; unit kernel %.0f us" % (cpu, t))
for n in (1, 2, 4):
    for first in ("a", "b"):
        us = timed(n, cyc, first)
        print("n=%d first=%s: graph %.1f us; ideal (fork at once) %.1f; second branch behind ALL of the first: %.1f"
              % (n, first, us, t + t / 4 + n * t, t + t / 4 + 2 * n * t), flush=True)
us = timed(4, cyc, "inline")
print("n=4 inline (branch a on the capture stream, join on b's stream): graph %.1f us; ideal %.1f" % (us, t + t / 4 + 4 * t))
for head in (1, 2):
    us = timed(4, cyc, "a", head)
    print("n=4 with %d short (t/8) kernels at the head of each branch: graph %.1f us; ideal %.1f" % (head, us, t + t / 4 + (4 - head) * t + head * t / 8))
