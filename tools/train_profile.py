"""Per-kernel breakdown of ONE reference-semantics train step (2 propagations forward + backward, BPR loss, Adam), replayed
as a hipGraph (bench.py's train leg on its own).
    python tools/train_profile.py [--steps 300] [--shared]                   -> ms per step
    python tools/train_profile.py --profile [--shared]                        -> runs itself under rocprofv3 --kernel-trace --stats
                                                                                 and prints us per step of every kernel
    python tools/train_profile.py --profile --eager --one-chain               -> the same step launched EAGERLY: every kernel one after
        the other on one stream, one propagation chain at a time (no hipGraph, no second stream) -- the stand-alone duration of every
        kernel of the step; the kernel-stats CSV is kept as gpurun_out/trainprof_eager_kernel_stats.csv (profiles/r05_trainstep_eager_
        kernel_stats.csv: what bench.py's train_roofline.avg_launch_us can be recomputed from)"""
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
    if args.eager:                                  # one launch after the other on the current stream, autograd's own order
        lossfn = ng.BPRLoss()

        def step(upn_):
            opt.zero_grad(set_to_none=True)
            loss = lossfn(model(upn_[0], upn_[1], edge), model(upn_[0], upn_[2], edge))
            loss.backward()
            opt.step()
            return loss
    else:
        step = GraphedBPRStep(model, ng.BPRLoss(), opt, edge, B, shared_propagation=args.shared)
    for _ in range(5):
        step(upn)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(upn)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    print(json.dumps({"train_ms_per_step": round(ms, 4), "steps": args.steps, "shared": args.shared, "batch": B, "eager": bool(args.eager),
                      "sparse_last_stage": type(model).sparse_last_stage, "torch_adam": bool(args.torch_adam),
                      "switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("NGACF_")}}), flush=True)


def profile(args):
    out = os.path.join(ROOT, "gpurun_out", "trainprof")
    subprocess.run(["rm", "-rf", out])
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable,
           os.path.abspath(__file__), "--steps", str(args.steps), "--workload", args.workload, "--dtype", args.dtype,
           "--stages", str(args.stages), "--batch", str(args.batch)] + (["--shared"] if args.shared else []) + (["--eager", "--one-chain"] if args.eager else [])
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "(no output, rc=%d)" % r.returncode)
    for root, _, files in os.walk(out):
        for f in files:
            if f.endswith("kernel_stats.csv"):
                rows = list(csv.DictReader(open(os.path.join(root, f))))
                n = args.steps + 5 + (0 if args.eager else 3)     # timed + warm-up steps (+ the capture's warm-up iterations)
                tot = 0.0
                lines = []
                for r_ in rows:
                    per = float(r_["TotalDurationNs"]) / 1e3 / n
                    tot += per
                    lines.append("%8.1f us/step  %6.2f calls/step  avg %7.1f us  %s" % (per, int(r_["Calls"]) / n, float(r_["AverageNs"]) / 1e3, r_["Name"][:110]))
                print("sum of kernel time: %.1f us per step" % tot)
                print("\n".join(lines[:45]))
                tag = ("shared" if args.shared else "refsem") + ("_eager" if args.eager else "")
                with open(os.path.join(ROOT, "gpurun_out", "trainprof_%s.txt" % tag), "w") as fo:
                    fo.write("sum of kernel time: %.1f us per step\n" % tot + "\n".join(lines) + "\n")
                if args.eager:
                    import shutil
                    shutil.copy(os.path.join(root, f), os.path.join(ROOT, "gpurun_out", "trainprof_eager_kernel_stats.csv"))


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
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python, one after the other (no hipGraph)")
    ap.add_argument("--one-chain", action="store_true", help="with --eager (which implies it): one propagation chain at a time, one stream")
    a = ap.parse_args()
    profile(a) if a.profile else run(a)
