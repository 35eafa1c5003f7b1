"""Per-kernel breakdown of ONE reference-semantics train step (2 propagations forward + backward, BPR loss, Adam), replayed
as a hipGraph (bench.py's train leg on its own).
    python tools/train_profile.py [--steps 300] [--shared]                   -> ms per step
    python tools/train_profile.py --profile [--shared]                        -> runs itself under rocprofv3 --kernel-trace --stats
                                                                                 and prints us per step of every kernel"""
import argparse
import csv
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import torch
    import bench
    import ngacf_amd as ng
    from ngacf_amd import data as D
    from ngacf_amd.graphs import GraphedBPRStep
    dev = torch.device("cuda:0")
    torch.manual_seed(2019)
    wl = args.workload
    U, I, _ = D.SHAPES[wl]
    edge = D.workload_graph(wl, dev)
    if os.environ.get("NGACF_RENUMBER", "degree") == "degree":       # the product's ingest (run_Gowalla.py --renumber degree)
        edge = D.renumber(edge, *D.degree_order(edge, U, I))
    model = bench.make_model(ng, torch, U, I, 64, args.stages, 0.1, args.dtype, dev)
    model.train()
    from ngacf_amd.optim import Adam
    opt = (torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True, fused=True) if args.torch_adam
           else Adam(model.parameters(), lr=1e-3, weight_decay=1e-5))
    g = torch.Generator(device=dev).manual_seed(7)
    B = args.batch
    upn = torch.stack([torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I, (B,), device=dev, generator=g),
                       torch.randint(0, I, (B,), device=dev, generator=g)])
    step = GraphedBPRStep(model, ng.BPRLoss(), opt, edge, B, shared_propagation=args.shared)
    for _ in range(5):
        step(upn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(upn)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    print(json.dumps({"train_ms_per_step": round(ms, 4), "steps": args.steps, "shared": args.shared, "batch": B,
                      "sparse_last_stage": type(model).sparse_last_stage, "torch_adam": bool(args.torch_adam),
                      "switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("NGACF_")}}), flush=True)


def profile(args):
    out = os.path.join(ROOT, "gpurun_out", "trainprof")
    subprocess.run(["rm", "-rf", out])
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable,
           os.path.abspath(__file__), "--steps", str(args.steps), "--workload", args.workload, "--dtype", args.dtype,
           "--stages", str(args.stages), "--batch", str(args.batch)] + (["--shared"] if args.shared else [])
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "(no output, rc=%d)" % r.returncode)
    for root, _, files in os.walk(out):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                rows = list(csv.DictReader(open(os.path.join(root, f))))
                n = args.steps + 5 + 3                      # timed + warm-up replays + the capture's warm-up iterations
                tot = 0.0
                lines = []
                for r_ in rows:
                    per = float(r_["TotalDurationNs"]) / 1e3 / n
                    tot += per
                    lines.append("%8.1f us/step  %6.2f calls/step  avg %7.1f us  %s" % (per, int(r_["Calls"]) / n, float(r_["AverageNs"]) / 1e3, r_["Name"][:110]))
                print("sum of kernel time: %.1f us per step" % tot)
                print("\n".join(lines[:45]))
                tag = "shared" if args.shared else "refsem"
                with open(os.path.join(ROOT, "gpurun_out", "trainprof_%s.txt" % tag), "w") as fo:
                    fo.write("sum of kernel time: %.1f us per step\n" % tot + "\n".join(lines) + "\n")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--workload", default="gowalla")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--stages", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--shared", action="store_true")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--torch-adam", action="store_true")
    a = ap.parse_args()
    profile(a) if a.profile else run(a)
