"""What an exchange of the peer-store all-gather (ngacf_amd/peer.py) costs, measured with W ranks SHARING the one GPU of the build
box (process group: gloo, for the handle exchange only; there is no second GPU here, so no xGMI and no second cache hierarchy --
the numbers are the protocol's launch / flag / wait cost, not a link rate):

  1. ping   : push of a `--ping-kb` block to every peer + wait for theirs, back to back in one replayed hipGraph of 100 rounds
              -> microseconds per exchange round (a push kernel, W-1 flag words, a wait kernel on each rank)
  2. forward: the node-partitioned evaluation forward at `--workload` scale, each rank's pass replayed as ONE hipGraph,
              (a) with the peer-store gathers, (b) with the gathers skipped (comm.dry_run: same kernels, stale remote rows)
              -> (a) - (b) = what the exchanges add to a pass when W ranks time-share the GPU.

  python tools/peer_probe.py --world 2 [--workload gowalla] [--dtype bf16] > gpurun_out/r03_peer_probe.jsonl"""
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
        from ngacf_amd import data as D
        from ngacf_amd.dist import Partition, ShardedSPUIGACF
        from ngacf_amd.peer import PeerComm
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        out = {"world": world, "rank": rank}

        def sync():
            torch.cuda.synchronize()
            dist.barrier()

        def replay_ms(g, reps):
            for _ in range(3):
                g.replay()
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)

        # ---- 1. ping ---------------------------------------------------------------------------------------------------
        rows = max(world, a.ping_kb * 1024 // 128 // world * world)             # a [rows, 64] bf16 table, rows / world per rank
        part = Partition(rows, world, world, rank)                              # (user half = the table, item half: one row each)
        comm = PeerComm(part, device=dev)
        lease = comm.new_lease()
        tab = comm.table(("ping", 0), (part.N, 64), torch.bfloat16, dev, lease)
        rounds = 100

        def ping():
            for _ in range(rounds):
                comm.start_gather("user", tab).wait()
            comm.join()
        ping()
        sync()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                ping()
        out["ping"] = {"block_bytes_per_rank": part.cu * 128, "rounds_per_replay": rounds,
                       "us_per_exchange_round": round(1e3 * replay_ms(g, 20) / rounds, 3)}
        comm.check()
        lease.release()

        # ---- 2. the forward ----------------------------------------------------------------------------------------------
        U, I, _ = D.SHAPES[a.workload]
        edge = D.workload_graph(a.workload, dev)
        res = {}
        for name, peer in (("peer_store", True), ("no_exchange", False)):
            torch.manual_seed(0)
            m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world, rank, stages=a.stages, adj=edge, peer_store=peer).to(dev).eval()
            m.compute_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
            m.comm.dry_run = not peer
            with torch.no_grad():
                for _ in range(2):
                    m.propagate_local(edge)
                sync()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(s):
                    with torch.cuda.graph(g, stream=s):
                        m.propagate_local(edge)
                res[name] = round(replay_ms(g, a.reps), 5)
            if peer:
                m.comm.check()
                res["exchanges_per_pass"] = m.comm.pushes // 3               # (two eager passes + the captured one)
        out["forward"] = dict(res, workload=a.workload, dtype=a.dtype, stages=a.stages,
                              us_added_per_exchange=round(1e3 * (res["peer_store"] - res["no_exchange"]) / max(res["exchanges_per_pass"], 1), 2))
        dist.barrier()
        q.put(out)
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put({"rank": rank, "exception": "%s\n%s" % (e, traceback.format_exc())})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--stages", type=int, default=3)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--ping-kb", type=int, default=1024)
    a = ap.parse_args()
    import torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, a.world, port, q, a)) for r in range(a.world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    bad = [r for r in res if "exception" in r]
    if bad:
        print(bad[0]["exception"], file=sys.stderr)
        sys.exit(1)
    print(json.dumps(sorted(res, key=lambda r: r["rank"])[0]))


if __name__ == "__main__":
    main()
