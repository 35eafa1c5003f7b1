"""GPU tests of the row-window entry points the node-partitioned path (ngacf_amd/dist.py) uses: every rank's
window/plan, run one after the other into shared global-row tables (what the in-place all-gather produces),
must reproduce the single-GPU result; and the sharded model with world size 1 equals the plain model."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(U, I, E, seed):
    rs = np.random.RandomState(seed)
    u = np.concatenate([np.arange(U), rs.randint(0, U, E - U)])
    i = np.concatenate([rs.randint(0, I, U), rs.randint(0, I, E - U)])
    key = np.unique(u.astype(np.int64) * I + i)
    return torch.from_numpy(np.stack([key // I, key % I]))


@pytest.mark.parametrize("P,dt", [(2, torch.float32), (3, torch.float32), (4, torch.bfloat16), (8, torch.float32), (8, torch.bfloat16)])
def test_partitioned_stage_equals_single_gpu(P, dt):
    from ngacf_amd import ops
    from ngacf_amd.dist import Partition
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    U, I, heads = 333, 251, 8
    edge = _graph(U, I, 7000, 3).to(dev)
    parts = [Partition(U, I, P, r) for r in range(P)]
    Up, Ip, N = parts[0].Up, parts[0].Ip, parts[0].N
    g = torch.Generator(device=dev); g.manual_seed(0)
    X = torch.zeros(N, 64, device=dev)
    X[:U] = torch.randn(U, 64, device=dev, generator=g)
    X[Up:Up + I] = torch.randn(I, 64, device=dev, generator=g)
    W = torch.randn(2, 64, 64, device=dev, generator=g) * 0.2
    avec = torch.randn(2, 64, device=dev, generator=g) * 0.2
    gout = torch.randn(N, 64, device=dev, generator=g).to(dt)
    # single-GPU reference on the padded graph
    full = build_plan(edge, Up, Ip, require_user_edges=False, seg_len=32, short_max=8)
    Xt = X.to(dt) if dt != torch.float32 else X
    H0, S0 = ops.transform_fwd(Xt, W, avec, Up, Ip, heads, dt, 0.1, 5)
    out0, agg0, R0 = ops.prop_fwd(full, H0, S0, S0, avec, heads, drop_p=0.1, seed=5, save=True)
    dH0, dS0 = ops.prop_bwd(full, H0, S0, S0, avec, out0, agg0, R0, gout, heads, drop_p=0.1, seed=5)
    dX0, dW0, da0 = ops.transform_bwd(Xt, W, H0, dH0, dS0, Up, Ip, heads, 0.1, 5)
    # the same, rank by rank, into shared tables
    H = torch.empty_like(H0); S = torch.empty_like(S0)
    for p in parts:
        ops.transform_fwd_rows(p.local_rows(Xt).contiguous(), W, avec, H, S, Up, Ip, heads, p.win, True, 0.1, 5)
    assert torch.equal(H, H0) and torch.equal(S, S0)
    out = torch.zeros_like(out0); agg = torch.zeros_like(agg0); R = torch.zeros_like(R0)
    plans = [build_plan(edge, Up, Ip, require_user_edges=False, seg_len=32, short_max=8, row_range=p.row_ranges) for p in parts]
    assert sum(pl.n_edges_directed for pl in plans) == full.n_edges_directed       # sharded plans: every edge once
    assert all(pl.n_edges_directed < 0.8 * full.n_edges_directed for pl in plans)
    for k, (p, pl) in enumerate(zip(parts, plans)):
        if k % 2 == 0:                                    # whole launch ...
            o, a_, r_ = ops.prop_fwd(pl, H, S, S, avec, heads, drop_p=0.1, seed=5, save=True)
        else:                                             # ... or the two half launches of the pipelined schedule
            o = torch.zeros_like(out0); a_ = torch.zeros_like(agg0); r_ = torch.zeros_like(R0)
            for half in ("item", "user"):
                ops.prop_fwd(pl, H, S, S, avec, heads, drop_p=0.1, seed=5, save=True, out=o, agg=a_, R=r_, half=half)
        for lo, hi in p.row_ranges:                       # (agg / R: two planes each, ops.prop_fwd)
            out[lo:hi], agg[:, lo:hi], R[:, lo:hi] = o[lo:hi], a_[:, lo:hi], r_[:, lo:hi]
    tol = 1e-6 if dt == torch.float32 else 1e-2
    assert float((out.float() - out0.float()).abs().max()) <= tol and float((R - R0).abs().max() / R0.abs().max()) < 1e-5
    assert float((agg.float() - agg0.float()).abs().max()) <= tol
    G = torch.empty(N, 64, device=dev); Gn = torch.empty(N, 64, device=dev, dtype=dt); ST = torch.empty(N, heads, 2, device=dev)
    P1 = torch.zeros(N, heads, device=dev)                # step 1's half of dS, per row
    for p in parts:
        ops.prop_bwd_node(out0, agg0, R0, p.local_rows(gout).contiguous(), S, G, Gn, ST, P1, Up, heads, p.win, True)

    def own_p1(p):                                        # a rank's dS table: P1 at its own rows, zero elsewhere
        t = torch.zeros_like(P1)
        for lo, hi in p.row_ranges:
            t[lo:hi] = P1[lo:hi]
        return t
    dH = torch.zeros_like(dH0); dS = torch.zeros_like(dS0)
    dX = torch.zeros_like(dX0).float(); dW = torch.zeros_like(dW0); da = torch.zeros_like(da0)
    for k, (p, pl) in enumerate(zip(parts, plans)):
        if k % 2 == 0:
            h_, s_ = ops.prop_bwd_edges(pl, H, S, avec, G, Gn, ST, heads, drop_p=0.1, seed=5, dS=own_p1(p))
            x_, w_, a_ = ops.transform_bwd_rows(p.local_rows(Xt).contiguous(), W, H, h_, s_, Up, Ip, heads, p.win, True, 0.1, 5)
            dW += w_; da += a_
            dX[p.u_lo:p.u_lo + p.cu] = x_[:p.cu].float()
            dX[Up + p.i_lo:Up + p.i_lo + p.ci] = x_[p.cu:].float()
        else:                                             # per half: edge step, then the half's transform backward
            h_ = torch.zeros_like(dH0); s_ = own_p1(p)
            for half in ("user", "item"):
                ops.prop_bwd_edges(pl, H, S, avec, G, Gn, ST, heads, drop_p=0.1, seed=5, half=half, dH=h_, dS=s_)
                x_, w_, a_ = ops.transform_bwd_rows(p.block(Xt, half).contiguous(), W, H, h_, s_, Up, Ip, heads, p.half_win(half), True, 0.1, 5)
                ty = 0 if half == "user" else 1
                dW[ty] += w_[ty]; da[ty] += a_[ty]
                p.block(dX, half).copy_(x_.float())
        dH += h_; dS += s_
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
    assert rel(dH, dH0) < 1e-5 and rel(dS, dS0) < 1e-4
    assert rel(dX, dX0.float()) < 1e-5 and rel(dW, dW0) < 1e-4 and rel(da, da0) < 1e-4


@pytest.mark.parametrize("balanced,d,stages,halves", [(False, 64, 3, False), (True, 64, 3, False), (True, 64, 3, True), (True, 128, 4, False),
                                                       (False, 128, 4, True)])
def test_sharded_model_world1_equals_plain_model(balanced, d, stages, halves, monkeypatch):
    """world size 1: the node-partitioned stack (whole-stage schedule, or the pipelined half launches; row exchange, renumbering
    when `balanced`; d = 64 L = 3 and BASELINE config 4's d = 128 L = 4) equals the plain model -- scores, every gradient, the
    state_dict round trip, the propagated table in the data set's numbering."""
    import ngacf_amd
    from ngacf_amd.dist import ShardedSPUIGACF
    monkeypatch.setenv("NGACF_DIST_WHOLE_BYTES", "0" if halves else str(32 << 20))
    dev = torch.device("cuda:0")
    U, I = 333, 251
    edge = _graph(U, I, 7000, 3).to(dev)
    torch.manual_seed(1)
    ref = (ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0) if (d, stages) == (64, 3)
           else ngacf_amd.SPUIStackedGACF(U, I, d, [64, 64], 0.0, stages=stages)).to(dev).train()
    with torch.no_grad():
        ref.uEmbd.weight.mul_(30); ref.iEmbd.weight.mul_(30)
    m = ShardedSPUIGACF(U, I, d, [64, 64], 0.0, world=1, rank=0, stages=stages, adj=edge if balanced else None).to(dev).train()
    assert m.comm.whole_stage(m.part.N * 128 * 4) == (not halves)
    m.load_full_state_dict(ref.state_dict())
    u = torch.randint(0, U, (128,), device=dev); it = torch.randint(0, I, (128,), device=dev)
    a, b = ref(u, it, edge), m(u, it, edge)
    assert float((a - b).detach().abs().max() / a.detach().abs().max()) < 1e-6
    a.sum().backward(); b.sum().backward()
    (ids_u, g_u), (ids_i, g_i), g_pad = m.local_grads_original()
    assert float((ref.uEmbd.weight.grad[ids_u] - g_u).abs().max() / ref.uEmbd.weight.grad.abs().max()) < 1e-5
    assert float((ref.iEmbd.weight.grad[ids_i] - g_i).abs().max() / ref.iEmbd.weight.grad.abs().max()) < 1e-5
    assert g_pad.numel() == 0
    with torch.no_grad():
        assert float((m.propagate(edge) - ref.propagate(edge)).abs().max()) < 1e-5
    for (k, p), (_, q) in zip(ref.gat.named_parameters(), m.gat.named_parameters()):
        assert float((p.grad - q.grad).abs().max() / p.grad.abs().max().clamp_min(1e-12)) < 1e-4, k
    sd = m.full_state_dict()
    assert all(torch.equal(sd[k], v) for k, v in ref.state_dict().items())


@pytest.mark.parametrize("bf16", [False, True])
def test_fused_score_exchange_equals_the_two_launch_form_bitwise(bf16, monkeypatch):
    """ngacf_rows_scatter_scored (the scores' backward formed inside the scatter of the exchanged rows' gradient) against
    ngacf_score_bwd over the row table + ngacf_rows_scatter: scores and every embedding gradient bit for bit; a batch with users and
    items named many times (the scatter sums them in one fixed order)."""
    import ngacf_amd
    from ngacf_amd.dist import ShardedSPUIGACF
    dev = torch.device("cuda:0")
    U, I = 333, 251
    edge = _graph(U, I, 7000, 3).to(dev)
    torch.manual_seed(3)
    m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world=1, rank=0, stages=3, adj=edge).to(dev).train()
    with torch.no_grad():
        m.uEmbd.weight.mul_(30); m.iEmbd.weight.mul_(30)
    if bf16:
        m.compute_dtype = torch.bfloat16
    g = torch.Generator(device=dev); g.manual_seed(9)
    u = torch.randint(0, 40, (512,), device=dev, generator=g); it = torch.randint(0, 50, (512,), device=dev, generator=g)
    res = []
    for fused in ("1", "0"):
        monkeypatch.setenv("NGACF_DIST_FUSED_SCORE", fused)
        m.zero_grad(set_to_none=True)
        sc = m(u, it, edge)
        (sc * torch.linspace(-1, 1, sc.numel(), device=dev)).sum().backward()
        res.append((sc.detach().clone(), m.uEmbd.weight.grad.clone(), m.iEmbd.weight.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][1].abs().max()) > 0


@pytest.mark.parametrize("bf16", [False, True])
def test_lockstep_two_chains_equal_the_two_calls_bitwise(bf16, monkeypatch):
    """forward_two (both propagations of a BPR step in lock-step, one grouped all-gather per stage for the two: _ShardedStackTwo)
    against forward(u, pos), forward(u, neg) one after the other: with dropout, the same seeds in the same order -- scores and
    every gradient bit for bit."""
    import ngacf_amd
    from ngacf_amd.dist import ShardedSPUIGACF
    dev = torch.device("cuda:0")
    U, I = 333, 251
    edge = _graph(U, I, 7000, 3).to(dev)
    torch.manual_seed(3)
    m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.1, world=1, rank=0, stages=3, adj=edge).to(dev).train()
    with torch.no_grad():
        m.uEmbd.weight.mul_(30); m.iEmbd.weight.mul_(30)
    if bf16:
        m.compute_dtype = torch.bfloat16
    g = torch.Generator(device=dev); g.manual_seed(9)
    u = torch.randint(0, U, (256,), device=dev, generator=g)
    p = torch.randint(0, I, (256,), device=dev, generator=g); n = torch.randint(0, I, (256,), device=dev, generator=g)
    res = []
    for lock in ("1", "0"):
        monkeypatch.setenv("NGACF_DIST_LOCKSTEP", lock)
        torch.manual_seed(11)                                        # (dropout seeds come from torch's CPU generator)
        m.zero_grad(set_to_none=True)
        ps, ns = m.forward_two(u, p, n, edge)
        (-torch.nn.functional.logsigmoid(ps - ns)).mean().backward()
        res.append([ps.detach().clone(), ns.detach().clone()] + [q.grad.clone() for q in m.parameters()])
    assert len(res[0]) == len(res[1]) and float(res[0][2].abs().max()) > 0
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_bench_distributed_path_rehearsal_on_one_rank():
    """bench.py's N>1 code path on a one-rank RCCL group, reached the way the driver reaches it: `python bench.py --gpus 1`
    with NGACF_BENCH_FORCE_DIST=1 and NO torchrun environment -- the parent (no GPU call) launches torch.distributed.run as
    a child.  Eager timing with the async half-table all-gathers overlapped, then the hipGraph-captured forward and train
    step with the RCCL launches inside; exactly one JSON line carrying roofline and cpu_baseline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NGACF_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
                        "--target-seconds", "0.2", "--cpu-budget", "3"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert "torch.distributed.run" in r.stderr and "graph_fallback" not in d, (r.stderr[-2000:], d.get("graph_fallback"))
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 1e9
    # (a Gowalla-scale table is small: one grouped all-gather + one launch per stage, Comm.whole_below; the 10^9-edge probe below
    # runs the pipelined half-table schedule)
    assert "3 whole-table all-gathers per forward" in d["config"]["parallelism"]
    assert d["config"]["passes_per_step"] >= 1 and d["timed_region_s"] > 0.1
    rf = d["roofline"]
    assert rf and rf["frac"] > 0 and rf["allgather"]["per_forward"] == 3 and rf["launches_per_forward"] == 3 and rf["allgather"]["avg_us_serial"] > 0
    assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    pr = d["scaling_probe"]                                              # the 10^9-edge forward through the same N>1 code path
    assert "error" not in pr and pr["n_gpus"] == 1 and pr["value"] > 1e10 and 0.5 < pr["prop_fwd_frac_of_hbm_peak"] < 1.2, pr
    assert d["ms_per_pass_eager"] > 0 and d["bpr_pairs_per_s"] > 1e5 and d["collective_launches_per_train_step"] > 0


def _gloo_gpu_worker(rank, world, port, q, bf16, overlap, halves=True):
    """One rank of a world-`world` run whose ranks all use cuda:0 (gloo moves the collectives through the host; RCCL refuses
    two ranks per GPU).  Everything else is the product path: HIP kernels, sharded plans, pipelined schedule."""
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ["NGACF_DIST_WHOLE_BYTES"] = "0" if halves else str(32 << 20)      # (the pipelined halves / what a table this small takes by default)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ngacf_amd
        from ngacf_amd.dist import ShardedSPUIGACF
        dev = torch.device("cuda:0")
        U, I = 333, 251
        edge = _graph(U, I, 7000, 3).to(dev)
        torch.manual_seed(1)
        ref = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev).train()
        with torch.no_grad():
            ref.uEmbd.weight.mul_(30); ref.iEmbd.weight.mul_(30)
        m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world, rank, stages=3, adj=edge).to(dev).train()
        m.load_full_state_dict(ref.state_dict())
        m.comm.overlap = overlap
        if bf16:
            ngacf_amd.set_compute_dtype(ref, torch.bfloat16)
            m.compute_dtype = torch.bfloat16
        g = torch.Generator(device=dev); g.manual_seed(5)
        u = torch.randint(0, U, (128,), device=dev, generator=g); it = torch.randint(0, I, (128,), device=dev, generator=g)
        a, b = ref(u, it, edge), m(u, it, edge)
        a.sum().backward(); b.sum().backward()
        (ids_u, g_u), (ids_i, g_i), g_pad = m.local_grads_original()
        rel = lambda x, y: float((x.float() - y.float()).abs().max() / y.float().abs().max().clamp_min(1e-12))
        err = {"score": rel(b.detach(), a.detach()), "gu": rel(g_u, ref.uEmbd.weight.grad[ids_u]), "gi": rel(g_i, ref.iEmbd.weight.grad[ids_i]),
               "pad": float(g_pad.abs().max()) if g_pad.numel() else 0.0}
        worst = 0.0
        for (k, p), (_, qq) in zip(ref.gat.named_parameters(), m.gat.named_parameters()):
            worst = max(worst, rel(qq.grad, p.grad))
        err["gat"] = worst
        with torch.no_grad():
            err["table"] = rel(m.propagate(edge), ref.propagate(edge))
        err["edges"] = (m._plan(edge).n_edges_directed, 2 * int(edge.shape[1]))
        q.put((rank, err))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {"exception": "%s\n%s" % (e, traceback.format_exc())}))


@pytest.mark.parametrize("world,bf16,overlap,halves", [(2, False, True, True), (3, False, True, True), (2, True, True, True), (2, False, False, True),
                                                       (2, False, True, False), (3, True, True, False)])
def test_ranks_sharing_one_gpu_under_gloo_match_the_single_gpu_model(world, bf16, overlap, halves):
    """The node-partitioned path END TO END with the real kernels and MORE THAN ONE RANK: `world` processes, all on cuda:0, gloo
    for the collectives -- degree-balanced partition, each rank's sharded plan, the pipelined half-table all-gathers in
    forward and backward (or, halves = False, the whole-stage schedule a small table takes: one grouped gather + one launch per
    stage), the 2B-row score exchange, the weight-gradient all-reduce -- against the plain single-GPU model
    (scores, every gradient, the propagated table).  fp32: 1e-5 / 2e-4; bf16 storage: both sides round alike up to summation
    order, 2e-2."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_gpu_worker, args=(r, world, port, q, bf16, overlap, halves)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    tol_s, tol_g = (2e-2, 6e-2) if bf16 else (1e-5, 2e-4)
    for rank, err in res:
        assert "exception" not in err, err.get("exception")
        assert err["score"] < tol_s and err["table"] < tol_s, (rank, err)
        assert max(err["gu"], err["gi"], err["gat"]) < tol_g and err["pad"] == 0.0, (rank, err)
    assert sum(err["edges"][0] for _, err in res) == res[0][1]["edges"][1]


def _run_runner(tmp, flags, world, extra_env=None):
    """run_Gowalla.py as a user would start it: one process (world 1) or `world` processes with torchrun's environment, all on
    cuda:0 with gloo collectives (RCCL refuses two ranks per GPU).  Returns rank 0's stdout."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base["PYTHONPATH"] = root + os.pathsep + base.get("PYTHONPATH", "")
    base.update(extra_env or {})
    cmd = [sys.executable, os.path.join(root, "run_Gowalla.py")] + flags
    if world == 1:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp, env=base)
        assert r.returncode == 0, r.stderr[-3000:]
        return r.stdout
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(world):
        env = dict(base, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NGACF_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=tmp, env=env))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    assert "train_loss" not in outs[1][0]                              # only rank 0 reports
    return outs[0][0]


def _parse_runner(out):
    import ast
    import re
    losses = [float(x) for x in re.findall(r"train_loss:([0-9.eE+-]+)", out)]
    metrics = [ast.literal_eval(m) for m in re.findall(r"metrics:(\{.*\})", out)]
    return losses, metrics


def test_run_gowalla_under_torchrun_environment_equals_the_single_process_run(tmp_path):
    """`torchrun run_Gowalla.py` -- the reference's train / eval loop on the node partition.  Two ranks (both on
    cuda:0, gloo) against the single-process runner, ml100k, README hyper-parameters without dropout (so that both are
    deterministic): same epoch losses and the same Recall/NDCG/precision/hit ratio @1,5,10,20 (each rank ranks half of the users);
    the checkpoint rank 0 wrote (reference layout) resumes in BOTH runners onto the same third epoch.  With dropout 0.2 the
    two-rank run lands in the reference's own band (fixture g12)."""
    flags = ["--dataset", "ml100k", "--model", "SPUIGACF", "--epochs", "2", "--eval_every", "1", "--save_every", "2", "--lr", "0.002",
             "--weight_decay", "0.000001", "--droprate", "0.0", "--eager"]
    d1, d2 = tmp_path / "one", tmp_path / "two"
    d1.mkdir(); d2.mkdir()
    l1, m1 = _parse_runner(_run_runner(str(d1), flags, 1))
    out2 = _run_runner(str(d2), flags, 2)
    l2, m2 = _parse_runner(out2)
    assert "node partition over 2 ranks" in out2 and len(l1) == len(l2) == 2 and len(m1) == len(m2) == 2
    assert np.allclose(l1, l2, rtol=2e-3), (l1, l2)
    for a, b in zip(m1, m2):
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            assert np.allclose(a[k], b[k], atol=5e-3), (k, a[k], b[k])
    # resume: the sharded run's checkpoint in the single-process runner and vice versa
    import shutil
    (d1 / "ckpts_own").mkdir()
    shutil.copy(str(d2 / "ckpts" / "SPUIGACF_ml100k_002.pkl"), str(d1 / "ckpts" / "SPUIGACF_ml100k_002.pkl"))
    res = flags[:4] + ["--epochs", "3", "--resume_from", "2"] + flags[6:]
    l1r, _ = _parse_runner(_run_runner(str(d1), res, 1))
    l2r, _ = _parse_runner(_run_runner(str(d2), res, 2))
    assert len(l1r) == len(l2r) == 1 and np.allclose(l1r, l2r, rtol=2e-3) and l2r[0] < l2[1] * 1.05, (l1r, l2r, l2)
    # with dropout: inside the reference's band (the distributional statement of test_ml100k_two_epochs_reach_reference_metrics)
    from conftest import load_golden
    ref = load_golden("g12_ref_ml100k_train")
    d3 = tmp_path / "drop"
    d3.mkdir()
    flags_d = [f if f != "0.0" else "0.2" for f in flags]
    _, m3 = _parse_runner(_run_runner(str(d3), flags_d, 2))
    nref = len(ref["random_seeds"])
    for key in ("recall", "ndcg"):
        r = np.array([ref["%s.%d" % (key, j)][3] for j in range(nref)])
        w = r.max() - r.min()
        assert r.min() - 0.5 * w < m3[-1][key][3] < r.max() + 0.5 * w, (key, m3[-1][key][3], r)


def _peer_worker(rank, world, port, q, bf16, graph):
    """One rank of a world-`world` run on cuda:0 whose half-table gathers go by PEER STORES (ngacf_amd/peer.py: hipIpc-mapped
    tables, push + flag, wait kernel) -- next to the same model on the process group's all-gather (gloo here)."""
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ngacf_amd
        from ngacf_amd.dist import ShardedSPUIGACF
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        U, I = 333, 251
        edge = _graph(U, I, 7000, 3).to(dev)
        torch.manual_seed(1)
        ref = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev).train()
        with torch.no_grad():
            ref.uEmbd.weight.mul_(30); ref.iEmbd.weight.mul_(30)
        models = []
        for peer in (False, True):
            m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world, rank, stages=3, adj=edge, peer_store=peer).to(dev).train()
            m.load_full_state_dict(ref.state_dict())
            if bf16:
                m.compute_dtype = torch.bfloat16
            models.append(m)
        a, b = models
        assert b.comm.peer_store and not a.comm.peer_store
        g = torch.Generator(device=dev); g.manual_seed(5)
        u = torch.randint(0, U, (128,), device=dev, generator=g); it = torch.randint(0, I, (128,), device=dev, generator=g)
        err = {}
        for step in range(3):                                  # (tables are reused from the second step on)
            for m in models:
                m.zero_grad(set_to_none=True)
            sa, sb = a(u, it, edge), b(u, it, edge)
            sa.sum().backward(); sb.sum().backward()
            b.comm.check()
            same = torch.equal(sa, sb)                         # the forward: bit for bit
            diffs = {"score": float((sa - sb).abs().max())}
            for (k, p), (_, qq) in zip(a.named_parameters(), b.named_parameters()):
                # (the score backward adds with float atomics: two runs of ONE model differ in the last bits too)
                diffs[k] = float((p.grad - qq.grad).abs().max() / p.grad.abs().max().clamp_min(1e-20))
                same = same and diffs[k] < (2e-2 if bf16 else 1e-5)
            err["step%d" % step] = bool(same)
            if not same:
                err["diffs%d" % step] = diffs
        with torch.no_grad():
            ta, tb = a.propagate(edge), b.propagate(edge)
            b.comm.check()
            err["table"] = bool(torch.equal(ta, tb))
            if graph:                                          # the evaluation forward as ONE replayed hipGraph on every rank
                b.eval()
                want = b.propagate_local(edge)
                torch.cuda.synchronize()
                dist.barrier()
                gr = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    with torch.cuda.graph(gr, stream=s):
                        got = b.propagate_local(edge)
                ok = True
                for _ in range(5):
                    got[0].zero_(); got[1].zero_()
                    gr.replay()
                    torch.cuda.synchronize()
                    ok = ok and torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
                b.comm.check()
                err["graph"] = bool(ok)
        err["pushes"] = b.comm.pushes
        dist.barrier()
        q.put((rank, err))
        b.comm.close()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {"exception": "%s\n%s" % (e, traceback.format_exc())}))


@pytest.mark.parametrize("world,bf16,graph", [(2, False, True), (3, True, False)])
def test_peer_store_gather_equals_the_collective_path_bitwise(world, bf16, graph):
    """the half-table all-gathers of forward and backward by stores into the peers' tables through hipIpc-mapped
    pointers + flag words (no collective launch), `world` processes on the ONE GPU: scores and the propagated table BITWISE
    equal to the same model on the process group's all-gather (gradients to 1e-5: the score backward's float atomics), over three steps (pool tables are reused from the
    second on), no wait ever timed out; and the evaluation forward replayed as one hipGraph per rank."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, q, bf16, graph)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert "exception" not in err, err.get("exception")
        assert all(err["step%d" % k] for k in range(3)) and err["table"], (rank, err)
        assert err.get("graph", True), (rank, err)
        assert err["pushes"] > 0


def test_run_gowalla_on_peer_stores_equals_the_collective_run(tmp_path):
    """The whole train / eval loop of `torchrun run_Gowalla.py` with the peer-store transport (NGACF_PEER_GATHER=1, two ranks on
    cuda:0): Adam moves the tables every step, the pool's tables are rewritten by the peer every step -- same epoch losses and
    metrics as the two-rank run on the process group's all-gather, with and without dropout (same seeds = same masks)."""
    for drop in ("0.0", "0.2"):
        flags = ["--dataset", "ml100k", "--model", "SPUIGACF", "--epochs", "2", "--eval_every", "1", "--lr", "0.002",
                 "--weight_decay", "0.000001", "--droprate", drop, "--eager"]
        da, db = tmp_path / ("gather" + drop), tmp_path / ("peer" + drop)
        da.mkdir(); db.mkdir()
        la, ma = _parse_runner(_run_runner(str(da), flags, 2))
        outb = _run_runner(str(db), flags, 2, {"NGACF_PEER_GATHER": "1"})
        lb, mb = _parse_runner(outb)
        assert "peer stores" in outb and len(la) == len(lb) == 2
        assert np.allclose(la, lb, rtol=1e-4), (drop, la, lb)
        for a, b in zip(ma, mb):
            for k in ("precision", "recall", "ndcg", "hit_ratio"):
                assert np.allclose(a[k], b[k], atol=2e-3), (drop, k, a[k], b[k])


def _peer_step_worker(rank, world, port, q):
    """The whole BPR train step of the node-partitioned model on peer stores -- half-table gathers, the scores' row exchange and
    the weight-gradient sum, no collective inside -- captured as ONE hipGraph, against eager steps of the same model on the
    process group (gloo)."""
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import ngacf_amd
        from ngacf_amd.dist import ShardedSPUIGACF
        from ngacf_amd.graphs import GraphedBPRStep
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        U, I, B = 333, 251, 256
        edge = _graph(U, I, 7000, 3).to(dev)
        torch.manual_seed(1)
        ref = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev)
        with torch.no_grad():
            ref.uEmbd.weight.mul_(30); ref.iEmbd.weight.mul_(30)
        ms = []
        for peer in (False, True):
            m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.0, world, rank, stages=3, adj=edge, peer_store=peer).to(dev).train()
            m.load_full_state_dict(ref.state_dict())
            m.comm.defer_weight_reduce = True
            ms.append(m)
        a, b = ms
        lossfn = ngacf_amd.BPRLoss()
        oa = ngacf_amd.optim.Adam(a.parameters(), lr=1e-2)
        ob = ngacf_amd.optim.Adam(b.parameters(), lr=1e-2)
        gs = GraphedBPRStep(b, lossfn, ob, edge, B, preserve_state=True, concurrent_forwards=False, after_backward=b.reduce_grads)
        g = torch.Generator(device=dev); g.manual_seed(5)
        worst, losses = 0.0, []
        for step in range(4):
            u = torch.randint(0, U, (B,), device=dev, generator=g)
            p = torch.randint(0, I, (B,), device=dev, generator=g)
            n = torch.randint(0, I, (B,), device=dev, generator=g)
            oa.zero_grad(set_to_none=True)
            la = lossfn(a(u, p, edge), a(u, n, edge))
            la.backward()
            a.reduce_grads()
            oa.step()
            lb = gs(torch.stack([u, p, n]))
            torch.cuda.synchronize()
            losses.append((float(la), float(lb)))
            for pa, pb in zip(a.parameters(), b.parameters()):
                worst = max(worst, float((pa - pb).abs().max() / pa.abs().max().clamp_min(1e-12)))
        b.comm.check()
        dist.barrier()
        q.put((rank, {"worst": worst, "losses": losses, "pushes": b.comm.pushes}))
        b.comm.close()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {"exception": "%s\n%s" % (e, traceback.format_exc())}))


def test_peer_store_train_step_without_a_collective_as_one_graph():
    """With every exchange of the step on peer stores (PeerComm.all_reduce_sum for the scores' row exchange and the weight-gradient
    sum) the captured step holds no collective at all.  Two ranks on cuda:0, four Adam steps replayed from ONE hipGraph per rank:
    losses and every parameter against eager steps on the process group's collectives (1e-4: float atomics in the score
    backward).  (The two forwards stay one after the other: as parallel branches the step stalls, ngacf_amd/peer.py.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_peer_step_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, out in res:
        assert "exception" not in out, out.get("exception")
        assert out["worst"] < 1e-4, (rank, out)
        for la, lb in out["losses"]:
            assert abs(la - lb) <= 1e-5 * max(1.0, abs(la)), (rank, out["losses"])
        assert out["pushes"] > 0
