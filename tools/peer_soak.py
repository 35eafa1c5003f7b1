"""Soak of the peer-store exchange protocol (ngacf_amd/peer.py, csrc/peer.hip): W ranks sharing the one GPU run `--rounds` rounds of

    fill my block of table A with a pattern of (round, rank)  ->  push + wait  ->  count the words of EVERY block that differ
    from the pattern their owner must have written this round  ->  push + wait on a second, tiny table (the acknowledgement
    that lets the next round overwrite A)

with the rounds unrolled 50 at a time into one replayed hipGraph per rank (so pushes, waits and checks run back to back on the
device, no host in between) and a second stream hammering the L2 with an unrelated copy.  A flag seen before its data, a stale
line served to the checker, a lost epoch or a timed-out wait all show up as a non-zero count (--no-wait: the check without the
wait, which must FAIL).

  python tools/peer_soak.py --world 2 --rounds 20000 > gpurun_out/r03_peer_soak.json"""
import argparse
import json
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q, a):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from ngacf_amd.dist import Partition
        from ngacf_amd.peer import PeerComm
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        rows = a.block_rows * world
        part = Partition(rows, world, world, rank)
        comm = PeerComm(part, device=dev)
        lease = comm.new_lease()
        A = comm.table(("soak", 0), (part.N, 64), torch.int32, dev, lease)          # 256 B per row
        B = comm.table(("ack", 0), (part.N, 1), torch.int32, dev, lease)
        mine = A[rank * part.cu:(rank + 1) * part.cu]
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        rnd = torch.zeros(1, dtype=torch.int32, device=dev)
        owner = (torch.arange(rows, device=dev) // part.cu).to(torch.int32).view(-1, 1)
        col = torch.arange(64, device=dev, dtype=torch.int32).view(1, -1)
        noise_a = torch.empty(8 << 20, dtype=torch.int32, device=dev)
        noise_b = torch.empty_like(noise_a)
        side = torch.cuda.Stream()
        per = 50

        def pattern(owners):
            return (rnd * 1000003 + owners * 7919 + col * 31 + 12345)

        def chunk():
            for _ in range(per):
                rnd.add_(1)
                mine.copy_(pattern(owner[rank * part.cu:(rank + 1) * part.cu]))
                h = comm.start_gather("user", A)
                if not a.no_wait:                                                    # (--no-wait: the negative control)
                    h.wait()
                bad.add_((A[:rows] != pattern(owner)).sum())
                comm.start_gather("user", B).wait()                                  # everybody has checked: A may be rewritten
            comm.join()
        chunk()                                                                     # eager once (allocations, first epochs)
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                chunk()
        reps = max(a.rounds // per, 1)
        for k in range(reps):
            g.replay()
            if k % 8 == 0:
                with torch.cuda.stream(side):
                    noise_b.copy_(noise_a)                                          # unrelated traffic through the same L2s
        torch.cuda.synchronize()
        comm.check()
        out = {"world": world, "rank": rank, "rounds": (reps + 1) * per, "block_bytes": part.cu * 256,
               "words_checked_per_round": rows * 64, "mismatching_words": int(bad.item()), "timeouts": 0}
        dist.barrier()
        q.put(out)
        lease.release()
        comm.close()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put({"rank": rank, "exception": "%s\n%s" % (e, traceback.format_exc())})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=20000)
    ap.add_argument("--block-rows", type=int, default=2048)
    ap.add_argument("--no-wait", action="store_true", help="negative control: check the table WITHOUT waiting for the peers' flags")
    a = ap.parse_args()
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, a.world, port, q, a)) for r in range(a.world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [r for r in res if "exception" in r]
    if bad:
        print(bad[0]["exception"], file=sys.stderr)
        sys.exit(1)
    res = sorted(res, key=lambda r: r["rank"])
    print(json.dumps({"world": a.world, "rounds": res[0]["rounds"], "block_bytes": res[0]["block_bytes"],
                      "words_checked_per_round_and_rank": res[0]["words_checked_per_round"],
                      "mismatching_words_by_rank": [r["mismatching_words"] for r in res],
                      "waited": not a.no_wait, "ok": all(r["mismatching_words"] == 0 for r in res)}))
    sys.exit(0 if all(r["mismatching_words"] == 0 for r in res) else 2)


if __name__ == "__main__":
    main()
