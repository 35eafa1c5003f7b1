#!/usr/bin/env python3
"""bench.py -- propagation edges/s (+ BPR pairs/s) of the HIP path on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gowalla|yelp|gowalla128|synthetic200m|synthetic1b|gowalla_real|yelp_real]
  (gowalla_real / yelp_real: the REAL interaction lists the reference ships -- its test halves, tests/golden/*_test_pairs.npz;
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
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--target-seconds", type=float, default=0.6, help="length of the timed region the pass count R is sized for")
    ap.add_argument("--passes-per-step", type=int, default=0, help="R (0 = sized from --target-seconds)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: serial all-gather -> traversal schedule")
    ap.add_argument("--renumber", default="degree", choices=["degree", "none"],
                    help="node numbering at ingest (run_Gowalla.py --renumber): by degree, descending, or the generator's ids")
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
           "sample": "%d eval forward(s) over %s, same weights (L=%d, fp32), %.2f s each, torch CPU ops in the reference's "
                     "operator order (oracle/cpu_baseline.py)" % (iters, sample, L, sec)}
    # every core (torch.set_num_threads(os.cpu_count())).  Only where the process really HAS more cores
    # than the 16-thread share: on the MI355X pool a one-GPU job gets 16 of the host's 256 CPUs (cgroup quota), and 256 torch
    # threads on them take 65 s for ONE forward over 10^5 edges (9 k edges/s, measured once: profiles/r03_cpu_all_cores.json)
    if usable > share:
        sec_a, it_a = run(usable, budget_s * 0.4)
        torch.set_num_threads(share)
        out["all_cores"] = {"value": round(trav / sec_a, 1), "unit": "edges/s", "cores": usable,
                            "sample": "%d forward(s), %.2f s each, torch.set_num_threads(%d)" % (it_a, sec_a, usable)}
    else:
        out["all_cores"] = {"value": None, "cores": host, "skipped": "the job may use %d of the host's %d CPUs (affinity / cgroup quota): "
                            "the 16-thread figure IS the all-usable-cores figure; %d threads on them were measured once at 9.1 k edges/s "
                            "(profiles/r03_cpu_all_cores.json)" % (usable, host, host)}
    return out


def cpu_sample(torch, model, edge, U, I, wl):
    """The BOUNDED sample the CPU leg runs on: the whole graph at Gowalla/Yelp scale (~2 s per forward); for the big
    synthetic graphs the induced subgraph of the first users holding ~10^6 edges (the reference's [2k, E] edge-feature
    matrix alone is 100 GB at 2x10^8 edges)."""
    p_cpu = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    E = int(edge.shape[1])
    if E <= 4_000_000:
        return p_cpu, edge.cpu(), U, I, "the whole %s graph" % wl
    u_hi = int(edge[0][1_000_000]) + 1
    n_e = int((edge[0] < u_hi).sum())
    sub = edge[:, :n_e]
    items, inv = torch.unique(sub[1], return_inverse=True)
    p_cpu["uEmbd.weight"] = p_cpu["uEmbd.weight"][:u_hi]
    p_cpu["iEmbd.weight"] = p_cpu["iEmbd.weight"][items.cpu()]
    return (p_cpu, torch.stack([sub[0], inv]).cpu(), u_hi, int(items.numel()),
            "the subgraph of the first %d users of the %s graph (%d edges, %d items touched)" % (u_hi, wl, n_e, int(items.numel())))


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


def roofline_of(BK, D, wl, dtype, U, I, E, dims, elem, prop_ms, n_timed, fused):
    """`roofline` of the dominant kernel (attention propagation) from its algorithmic bytes per launch (SURVEY.md 8(d):
    2E (4 + d T) + N (2 d T + 8)) and its average duration.  The roof depends on where the gathered rows come from:
      * table far beyond the 256 MiB Infinity Cache: HBM, peak 8 TB/s -- `frac` is the fraction of the HBM roofline;
      * table inside L2 + Infinity Cache (Gowalla / Yelp): the gather never reaches HBM; the roof is the blend of the measured
        L2-hit and Infinity-Cache row-gather rates (MI355X_MICROARCH.md, "Indexed rows") at the kernel's measured L2 hit
        rate, 1 / (hit / 17.8 + miss / 8.6) TB/s; the algorithmic rate over the HBM peak is reported separately as
        `frac_hbm_algorithmic` (it may exceed 1 there and is NOT an HBM statement)."""
    L = len(dims)
    N, d = U + I, dims[0][1]
    per_launch_bytes = [D.algorithmic_bytes_stage(U, I, E, din, dout, elem)[0] for din, dout, _ in dims]
    assert len(prop_ms) == L * n_timed, (len(prop_ms), L, n_timed)
    avg_ms = sum(prop_ms) / len(prop_ms)
    avg_bytes = sum(per_launch_bytes) / L
    achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
    per_stage_ms = [sum(prop_ms[s::L]) / n_timed for s in range(L)]
    regime = BK.gather_regime(N * d * elem, max(U, I) * d * elem)
    traffic, traffic_src = BK.traffic_from_profile(ROOT, wl, dtype, L, d)
    hit, hit_src = BK.l2_hit_from_profile(ROOT, wl, dtype)
    blend = l2blend = None
    if regime["regime"] == "HBM":
        bound, peak, peak_note = "hbm", BK.HBM_PEAK_GBS, "HBM3E spec peak"
    else:
        # The ceiling is fixed by regime and dtype BEFORE anything is measured.  The guide's two row-gather rates (L2-resident
        # 17.8 TB/s, Infinity-Cache 8.6 TB/s) are measured LOWER bounds of one loop over 1152-byte rows; blended by the kernel's L2
        # hit rate they are the roof of the bf16 kernel (128-byte rows: one line per gathered row).  fp32's 256-byte rows (two
        # lines per request burst) are known to exceed that blend, so their roof takes the L2's own peak (34.5 TB/s) for the hits.
        # Both fractions are in the line whatever the dtype.
        bound = "l2+mall gather"
        if hit is not None:
            blend = 1.0 / (hit / BK.ROOF_L2_GATHER_GBS + (1.0 - hit) / BK.ROOF_MALL_GATHER_GBS)
            l2blend = 1.0 / (hit / BK.L2_PEAK_GBS + (1.0 - hit) / BK.ROOF_MALL_GATHER_GBS)
            if elem == 2:
                peak = blend
                peak_note = ("bf16 rows: 1 / (hit / %.0f + miss / %.0f) GB/s (the guide's L2-resident and Infinity-Cache row-gather rates) at "
                             "the kernel's L2 hit rate %.3f (%s)" % (BK.ROOF_L2_GATHER_GBS, BK.ROOF_MALL_GATHER_GBS, hit, hit_src))
            else:
                peak = l2blend
                peak_note = ("fp32 rows: 1 / (hit / %.0f [the L2's own peak] + miss / %.0f) GB/s at the kernel's L2 hit rate %.3f (%s); "
                             "chosen by dtype before measuring -- 256-byte rows exceed the guide's gather blend, a measured lower bound"
                             % (BK.L2_PEAK_GBS, BK.ROOF_MALL_GATHER_GBS, hit, hit_src))
        else:
            peak, peak_note = regime["roof_GBps"], "no request-counter profile for this workload: the regime's row-gather rate"
    out = {"bound": bound, "kernel": "prop_fwd_kernel (ngacf_prop_fwd%s); traffic / l2_hit_rate: from_profile, not collected in this run"
                                     % (": with the next stage's transform in its epilogue on all but the last stage" if fused else ""),
           "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "GB/s", "frac": round(achieved / peak, 4),
           "peak_note": peak_note, "traffic": traffic, "traffic_source": traffic_src,
           "frac_hbm_algorithmic": round(achieved / BK.HBM_PEAK_GBS, 4),
           "frac_of_guide_gather_blend": round(achieved / blend, 4) if blend else None,
           "frac_of_l2_peak_blend": round(achieved / l2blend, 4) if l2blend else None,
           "counter_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
           "regime": regime["regime"], "regime_note": regime["note"], "l2_hit_rate": hit,
           "bytes_per_launch": int(avg_bytes), "avg_launch_us": round(avg_ms * 1e3, 2),
           "per_stage_us": [round(x * 1e3, 2) for x in per_stage_ms], "launches_timed": len(prop_ms),
           "timing": "dispatch-packet timestamps (hipExtLaunchKernel start/stop events) of eager launches of the replayed graph's kernels",
           "edges_per_s_in_kernel": round(2 * E / (avg_ms * 1e-3), 1)}
    return out


def train_roofline_of(torch, BK, ops, eager_step, model, wl, dtype, U, I, E, dims, elem, n_steps=10):
    """`train_roofline`: the train step's dominant kernel -- the dense stages' edge backward (ngacf_prop_bwd / ngacf_prop_bwd_dx,
    four launches per step, the bulk of the backward) -- from its algorithmic bytes, 2E (4 + d T + 8 heads) + N (2 d T + 8) (the
    split-sign form gathers ONE row and a (S, t) pair per head per edge: include/ngacf.h; SURVEY.md 8(d)'s form had 2 d T), and its own
    duration: dispatch-packet timestamps of eager train steps (launches one after the other; inside the replayed two-chain step
    a launch shares the GPU with the other chain's and lasts 1.3-1.7 x as long: profiles/r04_trainstep_timeline.txt).  Roof as in
    `roofline`: the cache-gather blend at the kernel's own L2 hit rate (profiles/r04_traffic_trainstep.json)."""
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
    bytes_survey = 2 * E * (4 + 2 * d * elem) + N * (2 * d * elem + 8)
    # the fused tail's own rows (bf16 plans): stage 0 writes the fp32 dX; a later dense stage writes dX and, at level 2, reads the
    # previous stage's out / agg / aggn / R / Rn / S and writes its Gn / ST / P1
    fused = bool(ops.FUSE_BWD) and dtype == "bf16"
    n_dense = L - 1 if sparse_last else L
    tail = 0
    if fused and n_dense > 0:
        later = N * d * elem + (N * (4 * d * elem + 24 * heads_dense) if ops.FUSE_BWD >= 2 else 0)
        tail = (N * d * 4 + (n_dense - 1) * later) / n_dense
    bytes_launch = bytes_edges + tail
    achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
    hit, hit_src = BK.bwd_hit_from_profile(ROOT, wl, dtype)
    regime = BK.gather_regime(N * (d * elem + 8 * heads_dense), max(U, I) * (d * elem + 8 * heads_dense))   # gathered: Gn rows and (S, t) pairs
    blend = l2blend = None
    if regime["regime"] == "HBM":
        bound, peak, note = "hbm", BK.HBM_PEAK_GBS, "HBM3E spec peak"
    elif hit is not None:
        bound = "l2+mall gather"
        blend = 1.0 / (hit / BK.ROOF_L2_GATHER_GBS + (1.0 - hit) / BK.ROOF_MALL_GATHER_GBS)
        l2blend = 1.0 / (hit / BK.L2_PEAK_GBS + (1.0 - hit) / BK.ROOF_MALL_GATHER_GBS)
        peak = blend if elem == 2 else l2blend
        note = "as `roofline` (ceiling fixed by dtype), at the edge backward's own L2 hit rate %.3f (%s)" % (hit, hit_src)
    else:
        # no counter profile of this workload / dtype: the L2's own peak is the only ceiling that holds whatever the hit rate is
        bound, peak, note = "l2 peak (upper bound)", BK.L2_PEAK_GBS, ("no request-counter profile of the edge backward for this workload and dtype "
                                                                       "(tools/traffic.py trainstep): measured against the L2 peak, an upper bound of any blend")
    from ngacf_amd import functional as NF
    return {"bound": bound, "kernel": "prop_bwd_kernel, dense stages (NGACF_FUSE_BWD=%d: %s); l2_hit_rate: from_profile"
                                      % (ops.FUSE_BWD, "with the stage's dX%s in its tail" % (" and the previous stage's node step" if ops.FUSE_BWD >= 2 else "")
                                         if ops.FUSE_BWD and dtype == "bf16" else "edge backward alone"),
            "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "GB/s", "frac": round(achieved / peak, 4), "peak_note": note,
            "frac_hbm_algorithmic": round(achieved / BK.HBM_PEAK_GBS, 4),
            "frac_of_guide_gather_blend": round(achieved / blend, 4) if blend else None,
            "frac_of_l2_peak_blend": round(achieved / l2blend, 4) if l2blend else None,
            "l2_hit_rate": hit, "bytes_per_launch": int(bytes_launch),
            "bytes_note": "2E (4 + d T + 8 heads) + N (2 d T + 8): one gathered row and a (S, t) pair per head per edge (split-sign backward)"
                          + (", + the fused tail's own rows, averaged over the dense stages: %d bytes per launch (dX; at level 2 the previous stage's out / agg / aggn "
                             "in, Gn / ST / P1 out)" % int(tail) if tail else ""),
            "bytes_edge_step_only": int(bytes_edges),
            "GBps_if_counted_in_the_two_row_form_of_SURVEY_8d": round(bytes_survey / (avg_ms * 1e-3) / 1e9, 1),
            "avg_launch_us": round(avg_ms * 1e3, 2),
            "per_stage_us": [round(1e3 * sum(x) / len(x), 2) for x in per_stage],
            "last_stage_is_the_scored_batch_only": sparse_last and NF is not None,
            "launches_timed": len(dense), "edges_per_s_in_kernel": round(2 * E / (avg_ms * 1e-3), 1),
            "timing": "dispatch-packet timestamps of EAGER train steps (one launch at a time)"}


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
    roofline = roofline_of(BK, D, wl, args.dtype, U, I, E, dims, elem, fm["prop_ms"], fm["n_timed"], fused)
    roofline["measured_copy_GBps"] = BK.copy_rate_GBps(torch, dev)
    if fm["transform_ms"]:
        roofline["transform_launch_us"] = round(1e3 * sum(fm["transform_ms"]) / len(fm["transform_ms"]), 2)
        roofline["transform_launches_per_pass"] = len(fm["transform_ms"]) // fm["n_timed"]
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
            train_rl = train_roofline_of(torch, BK, ops, eager_step, model, wl, args.dtype, U, I, E, dims, elem)
        except Exception as e:                        # (a secondary figure must not cost the line)
            print("[bench] train_roofline failed: %r" % (e,), file=sys.stderr, flush=True)
        plan = model.gat._plans.get(edge, U, I, args.seg_len)
        performed = performed_train_traversals(torch, model, plan, ub, pb, nb, L, E)
        train = {"bpr_pairs_per_s": round(B * ksteps / t_tr, 1), "train_ms_per_step": round(1e3 * t_tr / ksteps, 3),
                 "train_steps": ksteps, "batch": B, "droprate": args.droprate, "final_loss": round(float(loss.detach()), 6),
                 "train_traversals_per_step": {"performed": performed, "if_every_stage_ran_in_full": 4 * trav_per_pass,
                                               "note": "directed-edge gathers of one step (2 propagations, forward + backward); the last "
                                                       "stage of a scored batch walks the batch's rows only"},
                 "train_traversals_per_s": round(performed * ksteps / t_tr, 1), "train_roofline": train_rl}
        if not args.eager and not big:
            # reported SEPARATELY: the opt-in mode that shares one propagation between the
            # positive and the negative scoring of a step -- not reference semantics, never the headline number
            opt2 = Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
            gshared = GraphedBPRStep(model, lossfn, opt2, edge, B, shared_propagation=True)
            t_sh, n_sh, _ = time_train(lambda: gshared(upn))
            train["shared_propagation_optin"] = {"bpr_pairs_per_s": round(B * n_sh / t_sh, 1),
                                                 "train_ms_per_step": round(1e3 * t_sh / n_sh, 3)}

    # ---------------- CPU baseline: reference operator order on the host cores (checker code) -------------
    cpu = None
    if not args.no_cpu_baseline:
        p_cpu, e_c, Us, Is, desc = cpu_sample(torch, model, edge, U, I, wl)
        cpu = cpu_baseline_leg(torch, p_cpu, stage_layer_names(model), e_c, Us, Is, L, args.cpu_budget, desc)

    out = {"metric": "propagation edges/sec (directed edge traversals, forward, all stages)", "value": round(value, 1),
           "unit": "edges/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "%s U=%d I=%d E=%d, d=%d, L=%d stages (%s), 8 heads x 8 + 1 x %d"
                                  % (("the reference's data/%s_test.csv interaction list (real topology; the train half is a missing blob)"
                                      % {"gowalla_real": "Gowalla/g", "yelp_real": "Yelp/y"}[wl]) if wl in D.REAL_WORKLOADS
                                     else "%s-scale bipartite graph" % wl, U, I, E, d, L, type(model).__name__, d),
                      "graph": graph_law(torch, model, edge, U, I, args.seg_len),
                      "passes_per_step": R, "edges_directed_per_pass": trav_per_pass, "edges_directed_per_step": trav_per_step,
                      "seg_len": args.seg_len, "parallelism": "1 GPU",
                      "launch": "eager" if args.eager else "hipGraph replay",
                      "launches_per_pass": (L + 1 if fused else 2 * L),
                      "fused_next_stage_transform": bool(fused),
                      "node_numbering": "by degree, descending, at ingest (ngacf_amd.data.degree_order: run_Gowalla.py's default)"
                                        if args.renumber == "degree" else "the generator's ids"},
           "ms_per_pass": round(ms_per_step / R, 5), "timed_region_s": round(t_fwd, 4),
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
            "outputs_equal_the_sequential_forward": same,
            "note": "two complete, independent passes in flight on two streams (two captured graphs with their own tables); "
                    "NOT the headline: `value` above times passes one after the other"}


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
    rl = roofline_of(BK, D, wl, dtype, U, I, E, dims, elem, fm["prop_ms"], fm["n_timed"], stack_fuses_next(NF, ops, model, edge, U, I, 256))
    return {"workload": "%s-scale bipartite graph U=%d I=%d E=%d, d=64, L=3, eval forward" % (wl, U, I, E),
            "value": round(trav * fm["R"] * K / fm["t"], 1), "unit": "edges/s", "ms_per_pass": round(1e3 * fm["t"] / K / fm["R"], 5),
            "steps": K, "passes_per_step": fm["R"], "dtype": dtype, "roofline": rl}


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
