"""Stress of hipGraph capture next to c10d's watchdog thread (one RCCL rank): every collective form the node-partitioned path
issues inside a capture -- plain and async all-gather, the coalescing manager with async_ops, a grouped all-reduce -- captured
again and again right behind eager collectives.  python tools/capture_race.py <quiesce seconds> <iterations> [forms bitmask: 1 async all-gather, 2 async group, 4 grouped all-reduce, 8 all-reduce]"""
import os, sys, time, torch, torch.distributed as dist
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
quiesce = float(sys.argv[1])
t = torch.ones(1 << 16, device=dev); u = torch.ones(1 << 12, device=dev)
cmgr = dist.distributed_c10d._coalescing_manager


forms = int(sys.argv[3]) if len(sys.argv) > 3 else 15


def body():
    for _ in range(3):
        if forms & 1:
            w = dist.all_gather_into_tensor(t, t, async_op=True)
            torch.mul(u, 1.0001, out=u)
            w.wait()
        if forms & 2:
            with cmgr(async_ops=True) as cm:
                dist.all_gather_into_tensor(t, t)
                dist.all_gather_into_tensor(u, u)
            torch.mul(u, 1.0001, out=u)
            cm.wait()
    if forms & 4:
        with cmgr():
            dist.all_reduce(t)
            dist.all_reduce(u)
    if forms & 8:
        dist.all_reduce(u)


for it in range(int(sys.argv[2])):
    body()                                   # eager, like the warm-up in front of a capture
    dist.barrier()
    torch.cuda.synchronize()
    if quiesce > 0:
        time.sleep(quiesce)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
        time.sleep(0.02)
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    if it % 20 == 0:
        print("iter", it, "ok", flush=True)
dist.destroy_process_group()
print("DONE", flush=True)
