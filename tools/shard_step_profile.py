"""The node-partitioned model's replayed train step on ONE RCCL rank, in this process (no torchrun: RANK / WORLD_SIZE are set here),
so that rocprofv3 can trace it directly:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shardprof -o t -- python3 tools/shard_step_profile.py
    python tools/step_timeline.py gpurun_out/shardprof/.../t_kernel_trace.csv
Prints ms per step (graph replay)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")


def main():
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    import ngacf_amd as ng
    from ngacf_amd import data as D
    from ngacf_amd.dist import ShardedSPUIGACF
    from ngacf_amd.graphs import GraphedBPRStep
    from ngacf_amd.optim import Adam
    torch.manual_seed(2019)
    U, I, _ = D.SHAPES["gowalla"]
    edge = D.workload_graph("gowalla", dev)
    model = ShardedSPUIGACF(U, I, 64, [64, 64], 0.1, 1, 0, stages=3, adj=edge).to(dev)
    model.compute_dtype = torch.bfloat16
    model.comm.force = True
    model.comm.defer_weight_reduce = True
    model.train()
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    B = 2048
    g = torch.Generator(device=dev).manual_seed(7)
    upn = torch.stack([torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I, (B,), device=dev, generator=g),
                       torch.randint(0, I, (B,), device=dev, generator=g)])
    gs = GraphedBPRStep(model, ng.BPRLoss(), opt, edge, B, concurrent_forwards=model.concurrent_chains, after_backward=model.reduce_grads)
    for _ in range(5):
        gs(upn)
    torch.cuda.synchronize()
    n = int(os.environ.get("STEPS", "100"))
    t0 = time.perf_counter()
    for _ in range(n):
        gs(upn)
    torch.cuda.synchronize()
    print(json.dumps({"sharded_train_ms_per_step_world1": round(1e3 * (time.perf_counter() - t0) / n, 4), "steps": n}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
