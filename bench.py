#!/usr/bin/env python3
"""bench.py -- propagation edges/s (+ BPR pairs/s) of the HIP path on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W          (default workload: Gowalla-scale, d=64, L=3, bf16)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, RCCL)

A *step* of the headline number is one full forward propagation of the model (L stages x 2E
directed edge traversals) in eval mode; `value` = stages * 2E * K / t over all ranks.  A second
timed region runs K reference-semantics train steps (2 propagations + scoring + BPR loss +
backward + Adam, B = 2048) and is reported as `bpr_pairs_per_s`.  Inputs are synthetic, generated
on the device, resident in HBM before the timed region starts.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="gowalla", choices=["ml100k", "gowalla", "yelp", "gowalla128"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--stages", type=int, default=3, help="propagation stages L (2 = SPUIGACF, 3 = SPUIMultiGACF)")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--droprate", type=float, default=0.1)
    ap.add_argument("--seg-len", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    return ap.parse_args()


def make_model(ng, U, I, d, stages, droprate, dtype, dev):
    if stages == 2:
        m = ng.SPUIGACF(U, I, d, [64, 64], droprate)
    elif stages == 3 and d == 64:
        m = ng.SPUIMultiGACF(U, I, d, [64, 64], droprate)
    else:
        m = ng.SPUIStackedGACF(U, I, d, [64, 64], droprate, stages=stages)
    m = m.to(dev)
    ng.set_compute_dtype(m, torch.bfloat16 if dtype == "bf16" else torch.float32)
    return m


def stage_dims(model):
    return model.gat.stage_shapes()


_RESULT_FD = None


def emit(line):
    """The ONE JSON line, on the real stdout.  Everything else that lands on fd 1 during the run (RCCL prints a
    version banner there at init) has been pointed at stderr."""
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, (line + "\n").encode())


def main():
    global _RESULT_FD
    args = parse()
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("NGACF_BENCH_FORCE_DIST") == "1"      # rehearse the N>1 code path on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import ngacf_amd as ng
    from ngacf_amd import data as D, ops

    wl = "gowalla" if args.workload == "gowalla128" else args.workload
    d = 128 if args.workload == "gowalla128" else 64
    U, I, E_target = D.SHAPES[wl]
    elem = 2 if args.dtype == "bf16" else 4

    if world > 1 or force_dist:
        from ngacf_amd import dist as ND
        return ND.bench_main(args, ng, rank, world, dev, U, I, E_target, d, elem, HBM_PEAK_GBS, emit)

    torch.manual_seed(2019)
    if wl == "ml100k":
        u, i, _ = D.load_ml100k()
        edge = torch.from_numpy(D.ui_mat_indices(u, i, I)).to(dev)
    else:
        edge = D.synthetic_bipartite(U, I, E_target, seed=2019, device=dev)
    E = int(edge.shape[1])
    N = U + I
    model = make_model(ng, U, I, d, args.stages, args.droprate, args.dtype, dev)
    for m in model.modules():
        if hasattr(m, "seg_len"):
            m.seg_len = args.seg_len
    dims = stage_dims(model)
    L = len(dims)

    def sync():
        torch.cuda.synchronize()

    # ---------------- headline: forward propagation, eval mode ----------------
    model.eval()
    with torch.no_grad():
        if args.eager:
            run = lambda: model.propagate(edge)
        else:                                   # the same launches, captured once into a hipGraph and replayed
            from ngacf_amd.graphs import GraphedPropagate
            run = GraphedPropagate(model, edge)
        for _ in range(args.warmup):
            run()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        sync()
        t_fwd = time.perf_counter() - t0
        # per-launch duration of the dominant kernel: events on the launch stream around every ngacf_prop_fwd
        # call of K eager steps (events cannot be placed inside a replayed graph; same kernels, same data)
        with ops.KernelTimer("ngacf_prop_fwd") as kt:
            for _ in range(args.steps):
                model.propagate(edge)
            sync()
        durs = kt.durations_ms()
    trav_per_step = L * 2 * E
    value = trav_per_step * args.steps / t_fwd
    ms_per_step = 1e3 * t_fwd / args.steps

    # roofline of the dominant kernel (attention propagation): algorithmic bytes per launch / avg launch time
    per_launch_bytes = [D.algorithmic_bytes_stage(U, I, E, din, dout, elem)[0] for din, dout, _ in dims]
    assert len(durs) == L * args.steps
    avg_ms = sum(durs) / len(durs)
    avg_bytes = sum(per_launch_bytes) / L
    achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
    per_stage_ms = [sum(durs[s::L]) / args.steps for s in range(L)]
    roofline = {"bound": "hbm", "kernel": "prop_fwd_kernel (ngacf_prop_fwd)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "bytes_per_launch": int(avg_bytes), "avg_launch_us": round(avg_ms * 1e3, 2),
                "per_stage_us": [round(x * 1e3, 2) for x in per_stage_ms],
                "edges_per_s_in_kernel": round(2 * E / (avg_ms * 1e-3), 1)}
    # the box's own streaming rate next to the spec peak: device-to-device copy of 1 GiB (read + write counted)
    cp_a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    cp_b = torch.empty_like(cp_a)
    for _ in range(3):
        cp_b.copy_(cp_a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        cp_b.copy_(cp_a)
    e1.record()
    torch.cuda.synchronize()
    roofline["measured_copy_GBps"] = round(10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del cp_a, cp_b
    # HBM-side bytes per launch from the committed PMC passes of this same command (profiles/): separate
    # --pmc FETCH_SIZE / WRITE_SIZE runs with the gfx950 correction of MI355X_MICROARCH.md §HBM
    tf = os.path.join(ROOT, "profiles", "r01_traffic_%s.json" % wl)
    if os.path.exists(tf) and args.stages == 3 and d == 64:
        roofline["traffic"] = json.load(open(tf))["by_dtype"].get(args.dtype, {}).get("traffic_bytes")
    fwd_bytes = sum(sum(D.algorithmic_bytes_stage(U, I, E, din, dout, elem)) for din, dout, _ in dims)

    # ---------------- BPR train steps (reference semantics: 2 propagations per step) ----------------
    train = None
    if not args.no_train:
        model.train()
        lossfn = ng.BPRLoss()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=not args.eager,
                               fused=not args.eager)    # one multi-tensor kernel instead of ~100 per step
        g = torch.Generator(device=dev)
        g.manual_seed(7)
        B = args.batch
        ub = torch.randint(0, U, (B,), device=dev, generator=g)
        pb = torch.randint(0, I, (B,), device=dev, generator=g)
        nb = torch.randint(0, I, (B,), device=dev, generator=g)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = lossfn(model(ub, pb, edge), model(ub, nb, edge))
            loss.backward()
            opt.step()
            return loss
        ksteps = max(2, min(args.steps, 20))
        if not args.eager:                      # whole step (2 fwd, loss, bwd, Adam) as one replayed hipGraph
            from ngacf_amd.graphs import GraphedBPRStep
            gstep = GraphedBPRStep(model, lossfn, opt, edge, B)
            upn = torch.stack([ub, pb, nb])              # the sampler's output, packed: one device copy per step
            step = lambda: gstep(upn)
        for _ in range(min(args.warmup, 3)):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(ksteps):
            loss = step()
        sync()
        t_tr = time.perf_counter() - t0
        train = {"bpr_pairs_per_s": round(B * ksteps / t_tr, 1), "train_ms_per_step": round(1e3 * t_tr / ksteps, 3),
                 "train_steps": ksteps, "batch": B, "droprate": args.droprate, "final_loss": round(float(loss.detach()), 6),
                 "fwd_bwd_traversals_per_s": round(4 * trav_per_step * ksteps / t_tr, 1)}
        if not args.eager:
            # reported SEPARATELY: the opt-in mode that shares one propagation between the
            # positive and the negative scoring of a step -- not reference semantics, never the headline number
            opt2 = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True, fused=True)
            gshared = GraphedBPRStep(model, lossfn, opt2, edge, B, shared_propagation=True)
            for _ in range(min(args.warmup, 3)):
                gshared(upn)
            sync()
            t0 = time.perf_counter()
            for _ in range(ksteps):
                gshared(upn)
            sync()
            t_sh = time.perf_counter() - t0
            train["shared_propagation_optin"] = {"bpr_pairs_per_s": round(B * ksteps / t_sh, 1),
                                                 "train_ms_per_step": round(1e3 * t_sh / ksteps, 3)}

    # ---------------- CPU baseline: reference operator order on the host cores (checker code) -------------
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import cpu_baseline as CB
        from oracle import ngacf_oracle as O
        # the GPU box gives one GPU's share of the host: 16 cores (more threads only oversubscribe)
        ncores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(ncores)
        names = stage_layer_names(model)
        p_cpu = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
        x_cpu = torch.cat([p_cpu["uEmbd.weight"], p_cpu["iEmbd.weight"]])
        sec, iters = CB.time_forward(p_cpu, x_cpu, edge.cpu(), U, I, names, budget_s=args.cpu_budget)
        model_name = "unknown"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    model_name = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu = {"value": round(trav_per_step / sec, 1), "unit": "edges/s", "cores": ncores, "kind": "port",
               "cpu_model": model_name, "host_cpus": os.cpu_count(),
               "sample": "%d eval forward(s) of the same %s graph and weights (L=%d, fp32), %.2f s each, torch CPU ops in "
                         "the reference's operator order (oracle/cpu_baseline.py)" % (iters, wl, L, sec)}

    out = {"metric": "propagation edges/sec (directed edge traversals, forward, all stages)", "value": round(value, 1),
           "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "%s-scale bipartite graph U=%d I=%d E=%d, d=%d, L=%d stages (%s), 8 heads x 8 + 1 x %d"
                                  % (wl, U, I, E, d, L, type(model).__name__, d),
                      "edges_directed_per_step": trav_per_step, "seg_len": args.seg_len, "parallelism": "1 GPU",
                      "launch": "eager" if args.eager else "hipGraph replay"},
           "fwd_algorithmic_GBps": round(fwd_bytes / (ms_per_step * 1e-3) / 1e9, 1),
           "roofline": roofline, "cpu_baseline": cpu}
    if train:
        out.update(train)
    emit(json.dumps(out))


def stage_layer_names(model):
    """state_dict prefixes of the layers of every stage, in execution order."""
    return [["gat." + n for n in names] for names in model.gat.stage_layer_names()]


if __name__ == "__main__":
    main()
