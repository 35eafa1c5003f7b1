#!/usr/bin/env python3
"""bench.py -- propagation edges/s (+ BPR pairs/s) of the HIP path on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gowalla|yelp|gowalla128|synthetic200m|synthetic1b|gowalla_real|yelp_real]
  (gowalla_real / yelp_real: the REAL interaction lists the reference ships -- its test halves, data/real/*_test_pairs.npz;
  one GPU; the degree law of every workload's graph is printed in config.graph)

N = 1 runs in this process.  N > 1 (or NGACF_BENCH_FORCE_DIST=1: the N > 1 code path rehearsed on one rank) without a
torchrun environment: this process -- which imports no torch and makes no GPU call -- starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...` as a
child and relays its one JSON line; under torchrun (RANK / WORLD_SIZE set) it is a rank.

A *pass* is one full forward propagation of the model (L stages x 2E directed edge traversals), eval mode, inputs
synthetic, generated on the device and resident in HBM before the timed region starts.  A *step* is R back-to-back
passes (R is chosen once, before the timed region, so that the K timed steps last >= ~0.6 s -- a 3 ms region moves by
several percent with one scheduler hiccup; R is in the line as config.passes_per_step); exactly K steps are timed
between barrier + synchronize() pairs; `value` = K * R * L * 2E / t over all ranks, `value_median_window` the same from
the median step.  A second timed region runs reference-semantics train steps (2 propagations + scoring + BPR loss +
backward + Adam, B = 2048) -> `bpr_pairs_per_s`.  One JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = ["ml100k", "gowalla", "yelp", "gowalla128", "synthetic200m", "synthetic1b", "gowalla_real", "yelp_real"]


def _benchkit():
    """ngacf_amd/benchkit.py loaded by path: the launcher parent must not import the package (torch, the HIP library)."""
    spec = importlib.util.spec_from_file_location("_ngacf_benchkit", os.path.join(ROOT, "ngacf_amd", "benchkit.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="gowalla", choices=WORKLOADS)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--stages", type=int, default=0, help="propagation stages L (2 = SPUIGACF, 3 = SPUIMultiGACF; default 3, "
                    "4 for --workload gowalla128 = BASELINE config 4's d=128 L=4)")
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--droprate", type=float, default=0.1)
    ap.add_argument("--seg-len", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-cpu-train", action="store_true", help="skip the CPU baseline's two train steps (~20-40 s)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--target-seconds", type=float, default=0.6, help="length of the timed region the pass count R is sized for")
    ap.add_argument("--passes-per-step", type=int, default=0, help="R (0 = sized from --target-seconds)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: serial all-gather -> traversal schedule")
    ap.add_argument("--renumber", default="degree", choices=["degree", "none"],
                    help="node numbering at ingest (run_Gowalla.py --renumber): by degree, descending, or the generator's ids")
    ap.add_argument("--no-roof", action="store_true", help="skip the measured gather roof (the stripped instantiations' launches): counter "
                    "passes and kernel traces of the pass itself")
    ap.add_argument("--no-probe", action="store_true", help="skip the secondary legs: the Yelp / fp32 / 10^9-edge probes of the default "
                    "(gowalla) line and the two-passes-in-flight figure")
    a = ap.parse_args(argv)
    if not a.stages:
        a.stages = 4 if a.workload == "gowalla128" else 3
    return a


_RESULT_FD = None


def emit(line):
    """The ONE JSON line, on the real stdout.  Everything else that lands on fd 1 during the run (RCCL prints a
    version banner there at init) has been pointed at stderr."""
    os.write(_RESULT_FD if _RESULT_FD is not None else 1, (line + "\n").encode())


def make_model(ng, torch, U, I, d, stages, droprate, dtype, dev):
    if stages == 2:
        m = ng.SPUIGACF(U, I, d, [64, 64], droprate)
    elif stages == 3 and d == 64:
        m = ng.SPUIMultiGACF(U, I, d, [64, 64], droprate)
    else:
        m = ng.SPUIStackedGACF(U, I, d, [64, 64], droprate, stages=stages)
    m = m.to(dev)
    ng.set_compute_dtype(m, torch.bfloat16 if dtype == "bf16" else torch.float32)
    return m


def stack_fuses_next(NF, ops, model, edge, U, I, seg_len):
    """Does the forward run every later stage's transform in the previous traversal's epilogue (ngacf_prop_fwd_next)?  The model's
    own rule (SPUIGACF._run), bf16 and fp32 storage alike."""
    if not NF.FUSE_NEXT:
        return False
    plan = model.gat._plans.get(edge, U, I, seg_len)
    shapes = model.gat.stage_shapes()
    return all(ops.can_fuse_next(plan, model.gat.compute_dtype, shapes[s - 1][1], shapes[s - 1][2]) and shapes[s][0] == 64 and shapes[s][1] in (64, 128)
               for s in range(1, len(shapes)))


def graph_law(torch, model, edge, U, I, seg_len):
    """the degree law of the benched graph and what the plan makes of it"""
    plan = model.gat._plans.get(edge, U, I, seg_len)
    du, di = torch.bincount(edge[0], minlength=U), torch.bincount(edge[1], minlength=I)
    return {"max_degree_user": int(du.max()), "max_degree_item": int(di.max()), "isolated_items": int((di == 0).sum()),
            "rows_over_64_edges": int((du > 64).sum() + (di > 64).sum()), "hub_rows_cut_into_slices": int(plan.nsplit),
            "hub_slices": int(plan.nparts), "long_segments": int(plan.nseg), "short_rows": int(plan.nshort)}


def stage_layer_names(model):
    """state_dict prefixes of the layers of every stage, in execution order."""
    return [["gat." + n for n in names] for names in model.gat.stage_layer_names()]


def cpu_baseline_leg(torch, p_cpu, names, edge_c, Us, Is, L, budget_s, sample):
    """`cpu_baseline`: oracle/cpu_baseline.py (the reference's operator order, torch CPU ops) on this box's host cores, TWICE:
    on a GPU's share of the host (16 threads: the figure in `value`) and on every core (torch.set_num_threads(os.cpu_count()),
    BASELINE.md section 3) -- both in the line, half the budget each.
    p_cpu: reference-keyed CPU state_dict of the (sub)graph's Us users / Is items; edge_c: its [2,E] CPU edge list."""
    from oracle import cpu_baseline as CB
    bk = _benchkit()
    x_cpu = torch.cat([p_cpu["uEmbd.weight"], p_cpu["iEmbd.weight"]])
    trav = L * 2 * int(edge_c.shape[1])
    host = os.cpu_count() or 1

    def run(nthreads, budget):
        torch.set_num_threads(nthreads)
        sec, iters = CB.time_forward(p_cpu, x_cpu, edge_c, Us, Is, names, budget_s=budget)
        return sec, iters
    usable = bk.usable_cpus()
    share = min(usable, 16)
    sec, iters = run(share, budget_s * 0.6 if usable > share else budget_s)
    out = {"value": round(trav / sec, 1), "unit": "edges/s", "cores": share, "kind": "port",
           "cpu_model": bk.cpu_model_name(), "host_cpus": host, "usable_cpus": usable,
           "sample": "%d eval forward(s), %s, %.2f s each (oracle/cpu_baseline.py)" % (iters, sample, sec)}
    # every core (torch.set_num_threads(os.cpu_count())).  Only where the process really HAS more cores
    # than the 16-thread share: on the MI355X pool a one-GPU job gets 16 of the host's 256 CPUs (cgroup quota), and 256 torch
    # threads on them take 65 s for ONE forward over 10^5 edges (9 k edges/s, measured once: profiles/r03_cpu_all_cores.json)
    if usable > share:
        sec_a, it_a = run(usable, budget_s * 0.4)
        torch.set_num_threads(share)
        out["all_cores"] = {"value": round(trav / sec_a, 1), "unit": "edges/s", "cores": usable,
                            "sample": "%d forward(s), %.2f s each, torch.set_num_threads(%d)" % (it_a, sec_a, usable)}
    else:
        out["all_cores"] = {"value": None, "cores": host, "skipped": "job may use %d of %d CPUs" % (usable, host)}
    return out


def cpu_train_leg(torch, model, edge, U, I, wl, names, B, droprate, target_edges=100_000):
    """`cpu_baseline.train_*` ("... and 2 timed train steps"): two reference-semantics BPR train steps of
    oracle/cpu_baseline.py (the reference's operator order, torch autograd, torch.optim.Adam) on the induced subgraph of the first
    users holding ~10^5 edges (a full Gowalla-scale step is 145 s on 8 threads); same batch size as the GPU leg."""
    from oracle import cpu_baseline as CB
    bk = _benchkit()
    p_cpu = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    E = int(edge.shape[1])
    k = min(target_edges, E - 1)
    u_hi = int(edge[0][k]) + 1
    n_e = int((edge[0] < u_hi).sum())
    sub = edge[:, :n_e]
    items, inv = torch.unique(sub[1], return_inverse=True)
    p_cpu["uEmbd.weight"] = p_cpu["uEmbd.weight"][:u_hi]
    p_cpu["iEmbd.weight"] = p_cpu["iEmbd.weight"][items.cpu()]
    share = min(bk.usable_cpus(), 16)
    torch.set_num_threads(share)
    sec, n = CB.time_train_steps(p_cpu, torch.stack([sub[0], inv]).cpu(), u_hi, int(items.numel()), names, B=B, droprate=droprate, steps=2)
    return {"train_pairs_per_s": round(B / sec, 2), "train_s_per_step": round(sec, 2), "train_cores": share,
            "train_sample": "%d steps, B=%d, subgraph of the first %d users (%d edges, %d items)" % (n, B, u_hi, n_e, int(items.numel()))}


def cpu_sample(torch, model, edge, U, I, wl):
    """The BOUNDED sample the CPU leg runs on: the whole graph at Gowalla/Yelp scale (~2 s per forward); for the big
    synthetic graphs the induced subgraph of the first users holding ~10^6 edges (the reference's [2k, E] edge-feature
    matrix alone is 100 GB at 2x10^8 edges)."""
    p_cpu = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    E = int(edge.shape[1])
    if E <= 4_000_000:
        return p_cpu, edge.cpu(), U, I, "whole %s graph" % wl
    u_hi = int(edge[0][1_000_000]) + 1
    n_e = int((edge[0] < u_hi).sum())
    sub = edge[:, :n_e]
    items, inv = torch.unique(sub[1], return_inverse=True)
    p_cpu["uEmbd.weight"] = p_cpu["uEmbd.weight"][:u_hi]
    p_cpu["iEmbd.weight"] = p_cpu["iEmbd.weight"][items.cpu()]
    return (p_cpu, torch.stack([sub[0], inv]).cpu(), u_hi, int(items.numel()),
            "subgraph of the first %d users of %s (%d edges, %d items)" % (u_hi, wl, n_e, int(items.numel())))


def measure_forward(torch, BK, ops, model, edge, steps, warmup, target_s, passes_per_step, eager, n_timed):
    """The eval-mode forward propagation of `model`: K steps of R passes (hipGraph replay unless eager) between synchronize()
    pairs, then `n_timed` eager passes whose traversal / transform launches carry dispatch-timestamp events (ops.KernelTimer).
    Returns a dict (seconds of the timed region, windows, R, per-launch kernel durations by stage)."""
    def sync():
        torch.cuda.synchronize()
    model.eval()
    with torch.no_grad():
        if eager:
            run = lambda: model.propagate(edge)
        else:                                   # the same launches, captured once into a hipGraph and replayed
            from ngacf_amd.graphs import GraphedPropagate
            run = GraphedPropagate(model, edge)
        run()
        t_pass = BK.pass_time(run, sync)
        R = passes_per_step or BK.passes_per_step(t_pass, steps, target_s)
        for _ in range(warmup):
            for _ in range(R):
                run()
        t_fwd, windows = BK.timed_steps(run, steps, R, sync, torch)
        # duration of every kernel of the pass from its dispatch packet's own timestamps (hipExtLaunchKernel start / stop
        # events): eager launches of the same kernels on the same data -- no host launch gap inside the figure, it is what a
        # rocprofv3 kernel trace reports for the replayed graph (profiles/README.md)
        for _ in range(3):
            model.propagate(edge)
        sync()
        with ops.KernelTimer() as kt:
            for _ in range(n_timed):
                model.propagate(edge)
            sync()
    return {"t": t_fwd, "windows": windows, "R": R, "prop_ms": kt.durations_ms(), "transform_ms": kt.durations_ms(kt.TAG_TRANSFORM_FWD),
            "n_timed": n_timed}


def measure_roofs(torch, BK, ops, model, edge, U, I, seg_len, dims, dtype, n, bwd=False):
    """The measured gather roof (benchkit.gather_roof_*_us: the traversal kernels' stripped instantiations over the BENCHED plan) per
    distinct head count of the model's 64-column stages; None when a stage is not 64 columns wide (no stripped instantiation)."""
    if any(dout != 64 for _, dout, _ in dims):
        return None
    plan = model.gat._plans.get(edge, U, I, seg_len)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    fn = BK.gather_roof_bwd_us if bwd else BK.gather_roof_fwd_us
    return {h: fn(torch, ops, plan, U, I, tdt, h, n) for h in sorted({int(hd) for _, _, hd in dims})}


def roofline_of(BK, D, wl, dtype, U, I, E, dims, elem, prop_ms, n_timed, fused, roofs=None):
    """`roofline` of the dominant kernel (attention propagation): algorithmic bytes per launch (2E (4 + d T) + N (2 d T
    + 8)) over its average duration, against
      * table far beyond the 256 MiB Infinity Cache: the HBM peak (8 TB/s) -- `frac` is the fraction of the HBM roofline;
      * table inside L2 + Infinity Cache (Gowalla / Yelp): the MEASURED gather roof -- the same bytes over the duration of the
        kernel's stripped instantiation on the same plan (`roofs`, measure_roofs: one xor per gathered dword), ONE yardstick for
        both storage types.  The guide-rate blends at the profiled L2 hit rate stay in the line as secondary figures
        (profiles/README.md explains every field)."""
    L = len(dims)
    N, d = U + I, dims[0][1]
    per_launch_bytes = [D.algorithmic_bytes_stage(U, I, E, din, dout, elem)[0] for din, dout, _ in dims]
    assert len(prop_ms) == L * n_timed, (len(prop_ms), L, n_timed)
    avg_ms = sum(prop_ms) / len(prop_ms)
    avg_bytes = sum(per_launch_bytes) / L
    achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
    per_stage_ms = [sum(prop_ms[s::L]) / n_timed for s in range(L)]
    regime = BK.gather_regime(N * d * elem, max(U, I) * d * elem)
    traffic, _ = BK.traffic_from_profile(ROOT, wl, dtype, L, d)
    hit, _ = BK.l2_hit_from_profile(ROOT, wl, dtype)
    roof_us = roof_add_us = roof_GBps = None
    if roofs:
        roof_us = sum(roofs[int(hd)]["xor"] for _, _, hd in dims) / L
        roof_add_us = sum(roofs[int(hd)]["add"] for _, _, hd in dims) / L
        roof_GBps = avg_bytes / (roof_us * 1e-6) / 1e9
    blend = l2blend = None
    if hit is not None:
        blend = 1.0 / (hit / BK.ROOF_L2_GATHER_GBS + (1.0 - hit) / BK.ROOF_MALL_GATHER_GBS)
        l2blend = 1.0 / (hit / BK.L2_PEAK_GBS + (1.0 - hit) / BK.ROOF_MALL_GATHER_GBS)
    if regime["regime"] == "HBM":
        bound, peak = "hbm", BK.HBM_PEAK_GBS
    elif roof_GBps:
        bound, peak = "l2+mall gather (measured roof)", roof_GBps
    else:
        bound, peak = "l2+mall gather (guide blend)", (blend if blend else regime["roof_GBps"])
    return {"bound": bound, "kernel": "prop_fwd_kernel" + (" (+ next stage's transform in its tail on all but the last stage)" if fused else ""),
            "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "GB/s", "frac": round(achieved / peak, 4),
            "traffic": traffic, "avg_launch_us": round(avg_ms * 1e3, 2), "bytes_per_launch": int(avg_bytes),
            "measured_gather_roof_GBps": round(roof_GBps, 1) if roof_GBps else None,
            "measured_gather_roof_us": round(roof_us, 2) if roof_us else None,
            "gather_roof_one_add_per_element_us": round(roof_add_us, 2) if roof_add_us else None,
            "frac_of_measured_gather_roof": round(achieved / roof_GBps, 4) if roof_GBps else None,
            "frac_hbm_algorithmic": round(achieved / BK.HBM_PEAK_GBS, 4),
            "frac_of_guide_gather_blend": round(achieved / blend, 4) if blend else None,
            "frac_of_l2_peak_blend": round(achieved / l2blend, 4) if l2blend else None,
            "l2_hit_rate": hit, "regime": regime["regime"],
            "per_stage_us": [round(x * 1e3, 2) for x in per_stage_ms], "launches_timed": len(prop_ms),
            "edges_per_s_in_kernel": round(2 * E / (avg_ms * 1e-3), 1)}


def train_roofline_of(torch, BK, ops, eager_step, model, wl, dtype, U, I, E, dims, elem, n_steps=10, roofs=None):
    """`train_roofline`: the train step's dominant kernel -- the dense stages' edge backward (ngacf_prop_bwd / ngacf_prop_bwd_dx) --
    from its algorithmic bytes, 2E (4 + d T + 8 heads) + N (2 d T + 8) (+ the fused tail's own rows), and its own duration:
    dispatch-packet timestamps of EAGER train steps (one launch at a time; inside the replayed two-chain step a launch shares the
    GPU with the other chain's).  Roof: the stripped edge backward over the same plan (`roofs`), as in `roofline`."""
    L = len(dims)
    N, d = U + I, dims[0][1]
    for _ in range(2):
        eager_step()
    torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        for _ in range(n_steps):
            eager_step()
        torch.cuda.synchronize()
    ms = kt.durations_ms(kt.TAG_PROP_BWD)
    if len(ms) != 2 * L * n_steps:
        raise RuntimeError("expected %d edge-backward launches, timed %d" % (2 * L * n_steps, len(ms)))
    sparse_last = bool(type(model).sparse_last_stage)
    # autograd runs a chain's stages last to first: launch k of a step is stage L - 1 - (k mod L)
    per_stage = [[ms[k] for k in range(len(ms)) if L - 1 - (k % L) == s_] for s_ in range(L)]
    dense = [x for s_ in range(L) for x in per_stage[s_] if not (sparse_last and s_ == L - 1)]
    avg_ms = sum(dense) / len(dense)
    heads_dense = int(dims[0][2]) if len(dims[0]) > 2 else d // 8
    bytes_edges = 2 * E * (4 + d * elem + 8 * heads_dense) + N * (2 * d * elem + 8)
    fused = bool(ops.FUSE_BWD) and dtype == "bf16"
    n_dense = L - 1 if sparse_last else L
    tail = 0
    if fused and n_dense > 0:
        later = N * d * elem                              # a later dense stage's tail writes its bf16 dX rows
        tail = (N * d * 4 + (n_dense - 1) * later) / n_dense
    bytes_launch = bytes_edges + tail
    achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
    hit, _ = BK.bwd_hit_from_profile(ROOT, wl, dtype)
    regime = BK.gather_regime(N * (d * elem + 8 * heads_dense), max(U, I) * (d * elem + 8 * heads_dense))
    roof_us = roofs[heads_dense]["xor"] if roofs and heads_dense in roofs else None
    if roof_us is not None and roofs[heads_dense]["add"] < roof_us:      # (the xor form is not always the faster of the two stripped forms)
        roof_us = roofs[heads_dense]["add"]
    roof_GBps = bytes_edges / (roof_us * 1e-6) / 1e9 if roof_us else None
    blend = 1.0 / (hit / BK.ROOF_L2_GATHER_GBS + (1.0 - hit) / BK.ROOF_MALL_GATHER_GBS) if hit is not None else None
    if regime["regime"] == "HBM":
        bound, peak = "hbm", BK.HBM_PEAK_GBS
    elif roof_GBps:
        bound, peak = "l2+mall gather (measured roof)", roof_GBps
    else:
        bound, peak = "l2 peak (upper bound)", (blend if blend else BK.L2_PEAK_GBS)
    return {"bound": bound, "kernel": "prop_bwd_kernel, dense stages, NGACF_FUSE_BWD=%d" % ops.FUSE_BWD,
            "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "GB/s", "frac": round(achieved / peak, 4),
            "avg_launch_us": round(avg_ms * 1e3, 2), "bytes_per_launch": int(bytes_launch), "bytes_edge_step_only": int(bytes_edges),
            "measured_gather_roof_us": round(roof_us, 2) if roof_us else None,
            "frac_hbm_algorithmic": round(achieved / BK.HBM_PEAK_GBS, 4),
            "frac_of_guide_gather_blend": round(achieved / blend, 4) if blend else None, "l2_hit_rate": hit,
            "per_stage_us": [round(1e3 * sum(x) / len(x), 2) for x in per_stage], "launches_timed": len(dense)}


def performed_train_traversals(torch, model, plan, ub, pb, nb_, L, E):
    """Directed-edge gathers one reference-semantics train step really performs (the last stage of a scored batch touches the
    batch's rows only): dense stages 2E each, forward and backward, for both propagations; the last stage: the edges of the
    batch's rows (forward), and in the backward those plus every other row's edges INTO the batch."""
    U = model.userNum
    N = U + model.itemNum
    if not model._sparse_batch(ub, 1):
        return 4 * L * 2 * E
    deg = (plan.rowptr[1:] - plan.rowptr[:-1]).long()
    tot = 0
    for items in (pb, nb_):
        m = torch.zeros(N, dtype=torch.bool, device=ub.device)
        m[ub] = True
        m[U + items] = True
        own = int(deg[m].sum())                      # forward: every edge of a batch row; backward: the same rows again
        into = int(m[plan.col.long()].sum()) - int((m[plan.col.long()] & torch.repeat_interleave(m, deg)).sum())
        tot += 2 * (L - 1) * 2 * E + own + (own + into)
    return tot


def worker(args):
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("NGACF_BENCH_FORCE_DIST") == "1"      # rehearse the N>1 code path on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    import ngacf_amd as ng
    from ngacf_amd import benchkit as BK, data as D, functional as NF, ops

    wl = "gowalla" if args.workload == "gowalla128" else args.workload
    d = 128 if args.workload == "gowalla128" else 64
    U, I, E_target = D.SHAPES[wl]
    elem = 2 if args.dtype == "bf16" else 4

    if (world > 1 or force_dist) and wl in D.REAL_WORKLOADS:
        raise SystemExit("--workload %s: one GPU (the real lists are a fifth of the size of the Gowalla-scale graph)" % wl)
    if world > 1 or force_dist:
        from ngacf_amd import dist as ND
        return ND.bench_main(args, ng, rank, world, dev, wl, d, elem, emit, cpu_baseline_leg)

    torch.manual_seed(2019)
    edge = ingest_graph(D, wl, dev, args)
    E = int(edge.shape[1])
    N = U + I
    model = make_model(ng, torch, U, I, d, args.stages, args.droprate, args.dtype, dev)
    for m in model.modules():
        if hasattr(m, "seg_len"):
            m.seg_len = args.seg_len
    dims = model.gat.stage_shapes()
    L = len(dims)
    big = wl in D.BIG_WORKLOADS
    fused = stack_fuses_next(NF, ops, model, edge, U, I, args.seg_len)

    def sync():
        torch.cuda.synchronize()

    # ---------------- headline: forward propagation, eval mode ----------------
    fm = measure_forward(torch, BK, ops, model, edge, args.steps, args.warmup, args.target_seconds, args.passes_per_step, args.eager,
                         max(args.steps, 20 if big else 100))
    t_fwd, windows, R = fm["t"], fm["windows"], fm["R"]
    trav_per_pass = L * 2 * E
    trav_per_step = trav_per_pass * R
    value = trav_per_step * args.steps / t_fwd
    ms_per_step = 1e3 * t_fwd / args.steps
    med_ms = BK.median(windows)
    roofs = None
    try:
        if not args.no_roof:
            roofs = measure_roofs(torch, BK, ops, model, edge, U, I, args.seg_len, dims, args.dtype, 10 if big else 100)
    except Exception as e:                        # (a secondary figure must not cost the line)
        print("[bench] gather roof failed: %r" % (e,), file=sys.stderr, flush=True)
    roofline = roofline_of(BK, D, wl, args.dtype, U, I, E, dims, elem, fm["prop_ms"], fm["n_timed"], fused, roofs)
    roofline["measured_copy_GBps"] = BK.copy_rate_GBps(torch, dev)
    if fm["transform_ms"]:
        roofline["transform_launch_us"] = round(1e3 * sum(fm["transform_ms"]) / len(fm["transform_ms"]), 2)
    fwd_bytes = sum(sum(D.algorithmic_bytes_stage(U, I, E, din, dout, elem)) for din, dout, _ in dims)

    tif = None
    if not args.eager and not big and not args.no_probe:      # (before the train leg: the freshly initialised model, like the headline;
                                                              #  --no-probe skips it: a kernel trace of that run must not average
                                                              #  overlapped launches into the traversal kernel's row)
        try:
            tif = two_in_flight(torch, model, edge, E, L)
        except Exception as e:                    # (a secondary figure must not cost the line)
            print("[bench] two_passes_in_flight failed: %r" % (e,), file=sys.stderr, flush=True)

    # ---------------- BPR train steps (reference semantics: 2 propagations per step) ----------------
    train = None
    if not args.no_train:
        model.train()
        lossfn = ng.BPRLoss()
        from ngacf_amd.optim import Adam                 # torch.optim.Adam's arithmetic and state, one ngacf_adam_step launch per step
        opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
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
        eager_step = step
        if not args.eager:                      # whole step (2 fwd, loss, bwd, Adam) as one replayed hipGraph
            from ngacf_amd.graphs import GraphedBPRStep
            gstep = GraphedBPRStep(model, lossfn, opt, edge, B)
            upn = torch.stack([ub, pb, nb])              # the sampler's output, packed: one device copy per step
            step = lambda: gstep(upn)

        def time_train(fn):
            for _ in range(min(args.warmup, 3)):
                fn()
            sync()
            t0 = time.perf_counter()
            fn()
            sync()
            t1 = time.perf_counter() - t0
            n = int(max(args.steps, min(2000, 0.4 / max(t1, 1e-6))))       # >= ~0.4 s of steps
            sync()
            t0 = time.perf_counter()
            for _ in range(n):
                out = fn()
            sync()
            return time.perf_counter() - t0, n, out
        t_tr, ksteps, loss = time_train(step)
        train_rl = None
        try:
            broofs = None if (big or args.no_roof) else measure_roofs(torch, BK, ops, model, edge, U, I, args.seg_len, dims, args.dtype, 100, bwd=True)
            train_rl = train_roofline_of(torch, BK, ops, eager_step, model, wl, args.dtype, U, I, E, dims, elem, roofs=broofs)
        except Exception as e:                        # (a secondary figure must not cost the line)
            print("[bench] train_roofline failed: %r" % (e,), file=sys.stderr, flush=True)
        plan = model.gat._plans.get(edge, U, I, args.seg_len)
        performed = performed_train_traversals(torch, model, plan, ub, pb, nb, L, E)
        train = {"bpr_pairs_per_s": round(B * ksteps / t_tr, 1), "train_ms_per_step": round(1e3 * t_tr / ksteps, 3),
                 "train_steps": ksteps, "batch": B, "droprate": args.droprate, "final_loss": round(float(loss.detach()), 6),
                 "train_traversals_per_step": performed, "train_traversals_per_s": round(performed * ksteps / t_tr, 1),
                 "train_roofline": train_rl}
        if not args.eager and not big:
            # reported SEPARATELY: the opt-in mode that shares one propagation between the
            # positive and the negative scoring of a step -- not reference semantics, never the headline number
            opt2 = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
            gshared = GraphedBPRStep(model, lossfn, opt2, edge, B, shared_propagation=True)
            t_sh, n_sh, _ = time_train(lambda: gshared(upn))
            train["shared_propagation_optin"] = {"bpr_pairs_per_s": round(B * n_sh / t_sh, 1),
                                                 "train_ms_per_step": round(1e3 * t_sh / n_sh, 3)}

    # ---------------- the evaluation's ranking: every user against every item, top 20 ----------------
    evl = None
    if not args.no_probe and U * I <= 10 ** 10:
        try:
            evl = eval_leg(torch, ops, model, edge, U, I)
        except Exception as e:                            # (a secondary figure must not cost the line)
            print("[bench] eval leg failed: %r" % (e,), file=sys.stderr, flush=True)

    # ---------------- CPU baseline: reference operator order on the host cores (checker code) -------------
    cpu = None
    if not args.no_cpu_baseline:
        p_cpu, e_c, Us, Is, desc = cpu_sample(torch, model, edge, U, I, wl)
        cpu = cpu_baseline_leg(torch, p_cpu, stage_layer_names(model), e_c, Us, Is, L, args.cpu_budget, desc)
        if train and not args.no_cpu_train:
            try:
                cpu.update(cpu_train_leg(torch, model, edge, U, I, wl, stage_layer_names(model), args.batch, args.droprate))
            except Exception as e:
                print("[bench] cpu train leg failed: %r" % (e,), file=sys.stderr, flush=True)

    # The driver's record keeps the SCALARS of `config`, `roofline` and `cpu_baseline`: the second half of the metric (BPR pairs/s)
    # and the train leg's roofline therefore also sit there, flat.  profiles/README.md explains every field of the line.
    config = {"workload": "%s U=%d I=%d E=%d, d=%d, L=%d (%s)"
                          % ("real %s test-half list" % wl if wl in D.REAL_WORKLOADS else "%s-scale synthetic bipartite graph" % wl,
                             U, I, E, d, L, type(model).__name__),
              "passes_per_step": R, "edges_directed_per_pass": trav_per_pass, "seg_len": args.seg_len, "parallelism": "1 GPU",
              "launch": "eager" if args.eager else "hipGraph replay", "launches_per_pass": (L + 1 if fused else 2 * L),
              "fused_next_stage_transform": bool(fused), "node_numbering": args.renumber,
              "graph": graph_law(torch, model, edge, U, I, args.seg_len)}
    if evl:
        config.update(evl)
    if train:
        config.update({"bpr_pairs_per_s": train["bpr_pairs_per_s"], "train_ms_per_step": train["train_ms_per_step"], "batch": train["batch"],
                       "droprate": train["droprate"]})
        trl = train.get("train_roofline")
        if trl:
            roofline.update({"train_kernel": trl["kernel"], "train_avg_launch_us": trl["avg_launch_us"], "train_frac": trl["frac"],
                             "train_achieved_GBps": trl["achieved"], "train_peak_GBps": trl["peak"],
                             "train_measured_gather_roof_us": trl["measured_gather_roof_us"]})
    out = {"metric": "propagation edges/sec (directed edge traversals, forward, all stages) + BPR pairs/sec", "value": round(value, 1),
           "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": config, "ms_per_pass": round(ms_per_step / R, 5), "timed_region_s": round(t_fwd, 4),
           "value_median_window": round(trav_per_step / (med_ms * 1e-3), 1) if med_ms > 0 else None,
           "window_ms": {"median": round(med_ms, 4), "min": round(min(windows), 4), "max": round(max(windows), 4)},
           "fwd_algorithmic_GBps": round(fwd_bytes * R / (ms_per_step * 1e-3) / 1e9, 1),
           "roofline": roofline, "cpu_baseline": cpu}
    if train:
        out.update(train)
    if tif is not None:
        out["two_passes_in_flight"] = tif
    if wl == "gowalla" and d == 64 and not args.no_probe:
        del model, edge
        torch.cuda.empty_cache()
        # north_star's roofline target is stated at Yelp scale: the same forward on the Yelp-scale graph, a short second region
        # in the reference's precision (fp32: BASELINE.md section 5, the dtype of north_star's 1e-4 bar) AND in bf16 storage; plus the
        # headline graph's forward in fp32
        for key, pwl, pdt in (("yelp_probe", "yelp", "fp32"), ("yelp_probe_bf16", "yelp", "bf16"), ("gowalla_fp32_probe", "gowalla", "fp32")):
            try:
                out[key] = forward_probe(args, torch, ng, BK, D, NF, ops, dev, pwl, pdt)
            except Exception as e:
                print("[bench] %s failed: %r" % (key, e), file=sys.stderr, flush=True)
                out[key] = {"error": repr(e)}
        # the same forward on the 10^9-edge graph (BASELINE config 5) at this GPU count: the regime where the node partition
        # can pay; lets one `bench.py --gpus N` sweep yield a second strong-scaling curve (ngacf_amd.dist.scaling_probe)
        try:
            from ngacf_amd.dist import scaling_probe
            out["scaling_probe"] = scaling_probe(args, 1, 0, dev, d)
        except Exception as e:
            print("[bench] scaling probe failed: %r" % (e,), file=sys.stderr, flush=True)
            out["scaling_probe"] = {"error": repr(e)}
    emit(json.dumps(out))


def eval_leg(torch, ops, model, edge, U, I, K=20, reps=5):
    """ngacf_eval_topk over the model's propagated table: all U users against all I items, the graph's own interactions as
    the train positives to leave out, top K -- the launch train_eval.eval_neg_all makes once per evaluation
    (train_eval_Gowalla.py:274-385).  Kernel time from events around single launches (min and mean of `reps`)."""
    dev = edge.device
    model.eval()
    with torch.no_grad():
        F = model.score_table(edge).contiguous()
    ptr = torch.zeros(U + 1, dtype=torch.long, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(edge[0], minlength=U), 0)
    order = torch.argsort(edge[0] * I + edge[1])                          # (user, item) ascending: the CSR's row order
    items = edge[1][order].to(torch.int32).contiguous()
    pool = (torch.bincount(edge[1], minlength=I) > 0).to(torch.uint8)
    users = torch.arange(U, device=dev)
    ops.eval_topk(F, users, U, I, pool, ptr, items, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.eval_topk(F, users, U, I, pool, ptr, items, K)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return {"eval_topk_ms": round(min(ts), 4), "eval_topk_K": K}          # (the line must stay under 6 KB: two scalars)


def two_in_flight(torch, model, edge, E, L, seconds=0.3):
    """Secondary figure, not the headline: the same eval forward with TWO independent passes in flight -- two captured graphs of
    the same model (each with its own intermediate tables), replayed alternately on two streams, as an inference service with two
    outstanding requests would.  The first transform of one pass (launch- and latency-bound, ~10 us on a cold L2) then runs under
    the other pass's last traversal.  Every pass is complete; none shares a result with another."""
    import time
    from ngacf_amd.graphs import GraphedPropagate
    with torch.no_grad():
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        runs = []
        for st in streams:                                   # captured on its OWN stream: the hub rows' arrival counters are
            runs.append(GraphedPropagate(model, edge, stream=st))   # per stream (plan.tickets_for); two passes in flight must not share them
        want = model.propagate(edge).clone()

        def burst(n):
            cur = torch.cuda.current_stream()
            for st in streams:
                st.wait_stream(cur)
            for k in range(n):
                with torch.cuda.stream(streams[k & 1]):
                    runs[k & 1]()
            for st in streams:
                cur.wait_stream(st)
        burst(20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        burst(100)
        torch.cuda.synchronize()
        n = max(100, int(100 * seconds / max(time.perf_counter() - t0, 1e-6)))
        n += n & 1
        t0 = time.perf_counter()
        burst(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        bits = lambda t: t.view(torch.int16 if t.element_size() == 2 else torch.int32)
        same = bool(torch.equal(bits(runs[0].out), bits(want))) and bool(torch.equal(bits(runs[1].out), bits(want)))
    if not same:
        d = [float((r.out.float() - want.float()).abs().max()) for r in runs] + [float((runs[0].out.float() - runs[1].out.float()).abs().max())]
        raise RuntimeError("two passes in flight: a pass differs from the sequential forward (max |diff| pass 0 / pass 1 / between: %r)" % (d,))
    return {"value": round(L * 2 * E * n / dt, 1), "unit": "edges/s", "passes": n, "ms_per_pass_throughput": round(1e3 * dt / n, 5),
            "outputs_equal_the_sequential_forward": same}


def ingest_graph(D, wl, dev, args):
    """The workload's graph as the product's ingest hands it to the model: the generator's [2,E] indices, node ids renumbered by
    degree (ngacf_amd.data.degree_order -- run_Gowalla.py's default, --renumber degree; a relabelling of the same graph)."""
    U, I, _ = D.SHAPES[wl]
    edge = D.workload_graph(wl, dev)
    if args.renumber == "degree":
        edge = D.renumber(edge, *D.degree_order(edge, U, I))
    return edge


def forward_probe(args, torch, ng, BK, D, NF, ops, dev, wl, dtype):
    """The headline measurement repeated on another graph / in another storage type: forward only, ~0.3 s timed region, its own
    roofline.  Yelp scale is BASELINE config 3's graph and north_star's roofline target (">= 50 % of the HBM roofline on the
    attention-propagation kernel at Yelp scale on 1 GPU"); fp32 is the reference's precision."""
    U, I, _ = D.SHAPES[wl]
    torch.manual_seed(2019)
    edge = ingest_graph(D, wl, dev, args)
    E = int(edge.shape[1])
    elem = 2 if dtype == "bf16" else 4
    model = make_model(ng, torch, U, I, 64, 3, args.droprate, dtype, dev)
    dims = model.gat.stage_shapes()
    K = max(5, min(args.steps, 10))
    fm = measure_forward(torch, BK, ops, model, edge, K, 2, 0.3, 0, args.eager, 50)
    trav = len(dims) * 2 * E
    roofs = None
    try:
        roofs = measure_roofs(torch, BK, ops, model, edge, U, I, 256, dims, dtype, 50)
    except Exception as e:
        print("[bench] gather roof (%s %s) failed: %r" % (wl, dtype, e), file=sys.stderr, flush=True)
    rl = roofline_of(BK, D, wl, dtype, U, I, E, dims, elem, fm["prop_ms"], fm["n_timed"], stack_fuses_next(NF, ops, model, edge, U, I, 256), roofs)
    keep = ("bound", "achieved", "peak", "frac", "avg_launch_us", "measured_gather_roof_us", "frac_hbm_algorithmic", "frac_of_guide_gather_blend",
            "frac_of_l2_peak_blend", "l2_hit_rate", "per_stage_us", "bytes_per_launch")
    return {"workload": "%s-scale U=%d I=%d E=%d d=64 L=3 eval forward" % (wl, U, I, E),
            "value": round(trav * fm["R"] * K / fm["t"], 1), "unit": "edges/s", "ms_per_pass": round(1e3 * fm["t"] / K / fm["R"], 5),
            "dtype": dtype, "roofline": {k: rl[k] for k in keep}}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    in_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    want_dist = args.gpus > 1 or os.environ.get("NGACF_BENCH_FORCE_DIST") == "1"
    if want_dist and not in_torchrun:
        # launcher: no torch import, no GPU call in this process -- the ranks are child processes
        sys.exit(_benchkit().self_launch(args.gpus, argv))
    worker(args)


if __name__ == "__main__":
    main()
