"""world_size-2 (and 3) gloo runs of the node-partitioned path on CPU.  The local compute is the TEST-ONLY torch
stand-in tests/fake_backend.py (there is no GPU here); what is under test is ngacf_amd/dist.py: the partition,
the padded numbering, the in-place all-gathers in forward AND backward, the weight-gradient all-reduce and the
autograd wiring.  Expected values: the reference's own float64 run (tests/golden/g234_*.npz)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tag, multi, defer, q, d=64, stages=None, balanced=True, overlap=True, halves=True, two=False):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # halves: the pipelined schedule of big graphs (two half-table gathers behind two half-launches per stage); otherwise what a
        # table this small takes by default: one grouped gather + one launch per stage (Comm.whole_below)
        os.environ["NGACF_DIST_WHOLE_BYTES"] = "0" if halves else str(32 << 20)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import fake_backend
        from ngacf_amd.dist import ShardedSPUIGACF
        g = dict(np.load(tag if os.path.isabs(tag) else os.path.join(HERE, "golden", "g234_%s.npz" % tag)))
        U, I = int(g["U"]), int(g["I"])
        sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
        edge = torch.from_numpy(g["edge"])
        m = ShardedSPUIGACF(U, I, d, [64, 64], 0.0, world, rank, stages=stages or (3 if multi else 2), backend=fake_backend,
                            adj=edge if balanced else None)
        m.load_full_state_dict(sd)
        m.comm.defer_weight_reduce = defer
        m.comm.overlap = overlap
        m.train()
        assert m.comm.whole_stage(m.part.N * 64 * 4) == (not halves)
        u, pos, neg = (torch.from_numpy(g[k]) for k in ("u", "pos", "neg"))
        # the two propagations of a step on the model's TWO communicators (ShardedSPUIGACF.chain: what the graphed train step does
        # with its two concurrent branches); the assertions below then cover the second communicator's forward and backward too
        assert m._comms[1] is not m._comms[0] and m._comms[1].group is not m._comms[0].group and not m.concurrent_chains   # (captured concurrency: opt-in)
        if two:                                          # both propagations in lock-step (what the captured one-queue step runs)
            ps, ns = m.forward_two(u, pos, neg, edge)
        else:
            ps = m(u, pos, edge)
            with m.chain(1, inline=not overlap):         # (inline: the gathers in the caller's own order, the captured step's form)
                ns = m(u, neg, edge)
            assert m._comms[1].launches > 0 and m._comms[1].overlap == overlap and m._comms[1].defer_weight_reduce == defer
            assert m._comms[1].inline == (not overlap) and not m._comms[0].inline
        loss = -torch.log(torch.sigmoid(ps - ns)).mean()
        loss.backward()
        if defer:
            m.reduce_grads()                  # one all-reduce per step instead of one per stage
        p = m.part
        err = {"whole_gathers": m._comms[0].whole_gathers + m._comms[1].whole_gathers}
        smax = float(np.abs(g["pos_f64"]).max())
        err["scores"] = max(float((ps.detach().double() - torch.from_numpy(g["pos_f64"])).abs().max()),
                            float((ns.detach().double() - torch.from_numpy(g["neg_f64"])).abs().max())) / smax
        err["loss"] = abs(float(loss.detach()) - float(g["loss_f64"]))
        gu, gi = g["g_f64.uEmbd.weight"], g["g_f64.iEmbd.weight"]
        scale = max(np.abs(gu).max(), np.abs(gi).max())
        (ids_u, g_u), (ids_i, g_i), g_pad = m.local_grads_original()     # this rank's rows, ids in the data set's numbering
        err["emb"] = max(float(np.abs(g_u.numpy() - gu[ids_u.numpy()]).max()), float(np.abs(g_i.numpy() - gi[ids_i.numpy()]).max())) / scale
        err["pad"] = float(g_pad.abs().max()) if g_pad.numel() else 0.0
        # the plan is sharded: this rank holds only the directed edges that end in its rows
        plan = m._plan(edge)
        err["plan_edges"] = (plan.n_edges_directed, 2 * int(edge.shape[1]))
        # eval-side plumbing: the full table in the data set's numbering equals the concatenated owners' rows
        with torch.no_grad():
            Ff = m.propagate(edge)
        err["propagate_shape"] = tuple(Ff.shape) == (U + I, d)
        worst = 0.0
        for k, gr in m.gat.reference_named("grad").items():
            ref = g["g_f64.gat." + k]
            worst = max(worst, float(np.abs(gr.numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-12))
        err["gat"] = worst
        full = m.full_state_dict()
        err["sd"] = max(float((full[k] - sd[k]).abs().max()) for k in sd)
        err["keys"] = set(full) == set(sd)
        q.put((rank, err))
        dist.destroy_process_group()
    except Exception as e:                                  # surface the failure in the parent
        import traceback
        q.put((rank, {"exception": "%s\n%s" % (e, traceback.format_exc())}))


@pytest.mark.parametrize("world,tag,multi,defer,balanced,overlap,halves,two", [
    (2, "spuigacf", False, False, True, True, True, False), (2, "spuimultigacf", True, True, True, True, True, False),
    (3, "spuigacf", False, True, True, True, True, False), (3, "spuimultigacf", True, False, False, True, True, False),
    (2, "spuimultigacf", True, True, True, False, True, False),
    (2, "spuimultigacf", True, True, True, True, False, False), (3, "spuigacf", False, False, False, True, False, False),
    (2, "spuimultigacf", True, True, True, True, False, True), (3, "spuigacf", False, False, True, True, False, True),
    (2, "spuigacf", False, True, True, True, True, True)])
def test_sharded_model_matches_reference_under_gloo(world, tag, multi, defer, balanced, overlap, halves, two):
    """Forward, loss and every gradient of the node-partitioned model (degree-balanced renumbering or plain id blocks,
    pipelined or serial schedule of half-table gathers, or the whole-stage schedule a small table takes -- one grouped gather and
    one launch per stage -- for one chain at a time or, two = True, both propagations of the step in lock-step with one grouped gather
    per stage for the two (forward_two; on the half schedule it falls back to the two calls); sharded plans, 2B-row score exchange)
    against the reference's float64 run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tag, multi, defer, q, 64, None, balanced, overlap, halves, two)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert "exception" not in err, err.get("exception")
        assert err["scores"] < 1e-5 and err["loss"] < 1e-5, (rank, err)
        assert err["emb"] < 2e-4 and err["gat"] < 2e-4, (rank, err)          # fp32 stand-in vs float64 reference
        assert err["pad"] == 0.0 and err["sd"] == 0.0 and err["keys"] and err["propagate_shape"], (rank, err)
        # the schedule that ran: no whole-table gather on the pipelined one; L per forward and L per backward, two chains, on the other
        assert err["whole_gathers"] == (0 if halves else (2 if two else 4) * (3 if multi else 2)), (rank, halves, two, err["whole_gathers"])
    total = sum(err["plan_edges"][0] for _, err in res)
    assert total == res[0][1]["plan_edges"][1], "the ranks' plans together hold every directed edge exactly once"
    assert all(err["plan_edges"][0] < 0.75 * err["plan_edges"][1] for _, err in res)


def test_partition_covers_every_node_once():
    from ngacf_amd.dist import Partition
    for U, I, P in ((50, 40, 2), (50, 40, 3), (7, 5, 8), (29858, 40981, 8)):
        seen = np.zeros(0, dtype=np.int64)
        parts = [Partition(U, I, P, r) for r in range(P)]
        assert all(p.Up >= U and p.Ip >= I and p.Up % P == 0 and p.Ip % P == 0 for p in parts)
        rows = np.concatenate([np.concatenate([np.arange(lo, hi) for lo, hi in p.row_ranges]) for p in parts])
        assert np.array_equal(np.sort(rows), np.arange(parts[0].N))
        ids = [p.local_original_ids() for p in parts]
        assert np.array_equal(np.sort(np.concatenate([a[a >= 0].numpy() for a, _ in ids])), np.arange(U))
        assert np.array_equal(np.sort(np.concatenate([b[b >= 0].numpy() for _, b in ids])), np.arange(I))


def test_degree_balanced_partition_equalises_rows_and_nnz():
    """blocks balanced by nnz.  Ids sorted by popularity (the worst case for blocks of consecutive ids):
    the snake deal gives every rank the same number of rows (+-1) AND the same number of edges (within 2 %)."""
    from ngacf_amd.dist import Partition
    rs = np.random.RandomState(0)
    U, I, P = 5000, 3000, 8
    deg_i = np.sort((rs.pareto(1.2, I) * 20 + 1).astype(np.int64))[::-1].copy()        # item 0 is the hub
    deg_u = np.sort((rs.lognormal(3, 1, U) + 1).astype(np.int64))[::-1].copy()
    naive = [int(deg_i[r * 375:(r + 1) * 375].sum()) for r in range(P)]
    assert max(naive) > 4 * min(naive)                                                   # consecutive-id blocks: badly skewed
    parts = [Partition(U, I, P, r, torch.from_numpy(deg_u), torch.from_numpy(deg_i)) for r in range(P)]
    for kind, deg in ((0, deg_u), (1, deg_i)):
        nnz, rows = [], []
        for p in parts:
            ids = p.local_original_ids()[kind]
            ids = ids[ids >= 0].numpy()
            rows.append(ids.size)
            nnz.append(int(deg[ids].sum()))
        assert max(rows) - min(rows) <= 1
        assert max(nnz) - min(nnz) <= max(0.02 * np.mean(nnz), deg.max()), (kind, nnz)
    p = parts[3]
    e = torch.stack([torch.arange(U) % U, torch.arange(U) % I])
    m = p.map_edges(e)
    assert torch.equal(p._on("inv_u", "cpu")[m[0]], e[0]) and torch.equal(p._on("inv_i", "cpu")[m[1]], e[1])


def test_sharded_d128_four_stages_matches_oracle_under_gloo(tmp_path):
    """BASELINE config 3 (embedSize 128, L = 4, table sharded with one all-gather per stage) at world size 2: the
    expected values come from the float64 oracle on the unsharded model."""
    import torch
    from oracle import ngacf_oracle as O
    import ngacf_amd
    rs = np.random.RandomState(4)
    U, I, B = 70, 55, 48
    uu = np.concatenate([np.arange(U), rs.randint(0, U, 900)]); ii = np.concatenate([rs.randint(0, I - 3, U), rs.randint(0, I - 3, 900)])
    key = np.unique(uu.astype(np.int64) * I + ii)
    edge = np.stack([key // I, key % I])
    torch.manual_seed(9)
    ref = ngacf_amd.SPUIStackedGACF(U, I, 128, [64, 64], 0.0, useCuda=False, stages=4)
    sd = {k: v.detach().numpy().astype(np.float64) for k, v in ref.state_dict().items()}
    sd["uEmbd.weight"] *= 8; sd["iEmbd.weight"] *= 8      # scores O(1): the worker's plain fp32 -log(sigmoid) stays finite
    u, pos, neg = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
    loss, ps, ns, grads = O.bpr_step_grads(sd, u, pos, neg, edge, hidden=3)
    path = str(tmp_path / "d128.npz")
    np.savez(path, U=U, I=I, edge=edge, u=u, pos=pos, neg=neg, pos_f64=ps, neg_f64=ns, loss_f64=loss,
             **{"p." + k: v.astype(np.float32) for k, v in sd.items()}, **{"g_f64." + k: v for k, v in grads.items()})
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, False, True, q, 128, 4)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert "exception" not in err, err.get("exception")
        assert err["scores"] < 1e-5 and err["loss"] < 1e-5, (rank, err)
        assert err["emb"] < 5e-4 and err["gat"] < 5e-4, (rank, err)
        assert err["pad"] == 0.0 and err["sd"] == 0.0 and err["keys"], (rank, err)


def test_chunk_generated_shards_assemble_the_whole_graph():
    """ngacf_amd.dist.sharded_graph (the 10^9-edge workloads: every rank runs the chunked generator and keeps only the edges
    ending in its rows): the ranks agree on the partition, their shards together hold every directed edge exactly once, and
    each shard's plan equals the plan built by filtering the whole (renumbered) edge list."""
    from ngacf_amd import data as D
    from ngacf_amd.dist import sharded_graph
    from ngacf_amd.plan import build_plan
    shape = (300, 200, 5000)
    whole = D.synthetic_bipartite_big(*shape, seed=11, device="cpu", chunk_users=64)
    E = int(whole.shape[1])
    assert 3500 < E <= 5000 and int(torch.bincount(whole[0], minlength=300).min()) >= 1     # (duplicates are merged: dense toy graph)
    world = 3
    shards = [sharded_graph(shape, world, r, "cpu", seed=11, seg_len=16, chunk_users=64) for r in range(world)]
    assert all(e == E for _, _, e, _ in shards)
    assert all(torch.equal(shards[0][0].perm_u, p.perm_u) and torch.equal(shards[0][0].perm_i, p.perm_i) for p, _, _, _ in shards)
    assert sum(pl.n_edges_directed for _, pl, _, _ in shards) == 2 * E
    for part, plan, _, first in shards:
        ref = build_plan(part.map_edges(whole), part.Up, part.Ip, seg_len=16, require_user_edges=False, row_range=part.row_ranges)
        assert torch.equal(plan.rowptr, ref.rowptr) and torch.equal(plan.col, ref.col)
        assert torch.equal(plan.segs, ref.segs) and torch.equal(plan.srow, ref.srow) and torch.equal(plan.split, ref.split)
        assert torch.equal(first, whole[:, :first.shape[1]])                    # the CPU baseline's sample: the first user chunk
    nnz = [pl.n_edges_directed for _, pl, _, _ in shards]
    assert max(nnz) - min(nnz) < 0.05 * (2 * E / world)                        # degree-balanced shards


def test_sharded_plan_cache_never_reuses_a_plan_for_another_tensor():
    """ShardedSPUIGACF._plan keys on the mask tensor AND holds it (SPUIGACF._PlanCache), like the single-GPU
    modules: same tensor -> same plan, equal shape + new tensor -> its own plan, in-place edit -> new plan."""
    from ngacf_amd.dist import ShardedSPUIGACF
    import fake_backend
    a = torch.tensor([[0, 1, 2, 3], [0, 1, 0, 2]])
    m = ShardedSPUIGACF(4, 3, 64, [64, 64], 0.0, world=2, rank=0, backend=fake_backend)
    p1 = m._plan(a)
    assert m._plan(a) is p1
    b = torch.tensor([[0, 1, 2, 3], [1, 0, 1, 2]])                        # same shape, other edges
    p2 = m._plan(b)
    assert p2 is not p1 and not torch.equal(p2.col, p1.col)
    held = [t for _, t, _ in m._plans._d.values()]
    assert any(t is a for t in held) and any(t is b for t in held)        # the keyed storage stays allocated while cached
    before = p1.col.clone()
    a[1, 0] = 2                                                           # in-place edit bumps the version
    p3 = m._plan(a)
    assert p3 is not p1 and torch.equal(p1.col, before) and not torch.equal(p3.col, before)
    assert p1.n_edges_directed < 2 * 4                                    # a rank's plan holds its shard only


def _loop_data(seed=0, U=60, I=45, n=1400):
    from ngacf_amd.train_eval import Interactions
    rs = np.random.RandomState(seed)
    u = np.concatenate([np.arange(U), rs.randint(0, U, n)])
    i = np.concatenate([rs.randint(0, I, U), rs.randint(0, I, n)])
    key = np.unique(u.astype(np.int64) * I + i)
    u, i = key // I, key % I
    perm = rs.permutation(len(u))
    te = perm[:len(u) // 5]
    tr = perm[len(u) // 5:]
    tr = np.concatenate([tr, [np.nonzero(u == x)[0][0] for x in range(U) if x not in set(u[tr])]]).astype(np.int64)   # every user trains
    return Interactions(u[tr], i[tr], u[te], i[te], U, I, "cpu")


def _run_loop(world, rank, ckpt_dir, epochs=2):
    """Two epochs + evaluation + checkpoint of the sharded runner's loop on CPU tensors (torch stand-in kernels, host sampler)."""
    import fake_backend
    from ngacf_amd import sharded_loop as SL
    data = _loop_data()
    adj = data.adjacency()
    torch.manual_seed(2019)
    m = SL.create_model("SPUIMultiGACF", 0, data.U, data.I, 64, [64, 64], 0.0, adj, torch.device("cpu"), backend=fake_backend,
                        world=world, rank=rank)
    with torch.no_grad():
        m.uEmbd.weight.mul_(20.0); m.iEmbd.weight.mul_(20.0)
    opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=1e-5)
    lossfn = lambda p, n: -torch.log(torch.sigmoid(p - n)).mean()
    gen = torch.Generator().manual_seed(5)
    losses = [SL.train_bpr(m, 256, data, adj, opt, lossfn, gen) for _ in range(epochs)]
    metrics = SL.eval_neg_all(m, 256, data, adj, topk=fake_backend.eval_topk)
    path = os.path.join(ckpt_dir, "w%d.pkl" % world)
    SL.save_checkpoint(path, m, opt)
    return losses, metrics, m, opt, path


def _loop_worker(rank, world, port, ckpt_dir, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        from ngacf_amd import sharded_loop as SL
        r, w, dev = SL.init(device=torch.device("cpu"))
        assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
        losses, metrics, m, opt, path = _run_loop(world, rank, ckpt_dir)
        # resume on this world size from the checkpoint rank 0 wrote: same moments on the rows this rank owns
        m2 = SL.create_model("SPUIMultiGACF", 0, m.userNum, m.itemNum, 64, [64, 64], 0.0, _loop_data().adjacency(), torch.device("cpu"),
                             backend=__import__("fake_backend"), world=world, rank=rank)
        opt2 = torch.optim.Adam(m2.parameters(), lr=0.01, weight_decay=1e-5)
        SL.load_checkpoint(path, m2, opt2)
        same = all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
        for (pa, sa), (pb, sb) in zip(opt.state.items(), opt2.state.items()):
            same = same and torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]) and float(sa["step"]) == float(sb["step"])
        q.put((rank, {"losses": losses, "metrics": {k: np.asarray(v).tolist() for k, v in metrics.items() if k != "auc"}, "resume_same": same}))
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {"exception": "%s\n%s" % (e, traceback.format_exc())}))


def test_sharded_runner_loop_under_gloo_equals_one_rank(tmp_path):
    """ngacf_amd/sharded_loop.py (what `torchrun run_Gowalla.py` runs) at world size 2 under gloo against the same loop on one rank:
    same batches on every rank, same epoch losses, same Recall/NDCG/precision/hit ratio (each rank ranks half of the users, the
    sums are all-reduced), a checkpoint in the reference's layout that resumes at world size 2 AND at world size 1 with the same
    weights and Adam moments."""
    from ngacf_amd import sharded_loop as SL
    import fake_backend
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in range(world):
        assert "exception" not in res[r], res[r].get("exception")
    assert res[0]["losses"] == res[1]["losses"] and res[0]["metrics"] == res[1]["metrics"] and res[0]["resume_same"] and res[1]["resume_same"]
    losses1, metrics1, m1, opt1, path1 = _run_loop(1, 0, str(tmp_path))
    assert np.allclose(res[0]["losses"], losses1, rtol=1e-5, atol=1e-8), (res[0]["losses"], losses1)
    assert res[0]["losses"][1] < res[0]["losses"][0]
    for k in ("precision", "recall", "ndcg", "hit_ratio"):
        assert np.allclose(res[0]["metrics"][k], metrics1[k], atol=1e-9), (k, res[0]["metrics"][k], metrics1[k])
    assert metrics1["hit_ratio"][3] > 0
    # the world-2 checkpoint (reference layout: 53 optimizer entries, full tables) resumes on ONE rank with the one-rank run's state
    ck2, ck1 = torch.load(str(tmp_path / "w2.pkl")), torch.load(path1)
    assert list(ck2["model"]) == list(ck1["model"]) and len(ck2["optim"]["param_groups"][0]["params"]) == 53
    for k in ck1["model"]:
        assert float((ck2["model"][k] - ck1["model"][k]).abs().max()) < 1e-5, k
    for j in range(53):
        for k in ("exp_avg", "exp_avg_sq"):
            a, b = ck2["optim"]["state"][j][k], ck1["optim"]["state"][j][k]
            assert a.shape == b.shape and float((a - b).abs().max()) < 1e-5 * max(float(b.abs().max()), 1e-6) + 1e-9, (j, k)
    m3 = SL.create_model("SPUIMultiGACF", 0, m1.userNum, m1.itemNum, 64, [64, 64], 0.0, _loop_data().adjacency(), torch.device("cpu"),
                         backend=fake_backend, world=1, rank=0)
    opt3 = torch.optim.Adam(m3.parameters(), lr=0.01, weight_decay=1e-5)
    SL.load_checkpoint(str(tmp_path / "w2.pkl"), m3, opt3)
    assert float((m3.uEmbd.weight - m1.uEmbd.weight).abs().max()) < 1e-5


def test_sharded_model_draws_the_unsharded_models_initial_values():
    """Same torch seed -> the sharded model's full state_dict equals the unsharded model's, at any world size (the constructor
    draws in the unsharded constructor's order on full-size tables and keeps its rows)."""
    import ngacf_amd
    import fake_backend
    from ngacf_amd.dist import ShardedSPUIGACF
    U, I = 37, 29
    for stages, cls in ((2, ngacf_amd.SPUIGACF), (3, ngacf_amd.SPUIMultiGACF)):
        torch.manual_seed(7)
        ref = cls(U, I, 64, [64, 64], 0.1, useCuda=False).state_dict()
        for world in (1, 3):
            got = {}
            for rank in range(world):
                torch.manual_seed(7)
                m = ShardedSPUIGACF(U, I, 64, [64, 64], 0.1, world=world, rank=rank, stages=stages, backend=fake_backend)
                ou, oi = m.part.local_original_ids()
                for k, v in m.gat.state_dict().items():
                    assert torch.equal(v, ref["gat." + k]), (world, rank, k)
                got.setdefault("u", torch.zeros(U, 64))[ou[ou >= 0]] = m.uEmbd.weight.detach()[ou >= 0]
                got.setdefault("i", torch.zeros(I, 64))[oi[oi >= 0]] = m.iEmbd.weight.detach()[oi >= 0]
            assert torch.equal(got["u"], ref["uEmbd.weight"]) and torch.equal(got["i"], ref["iEmbd.weight"]), (stages, world)


def test_runner_parallel_flag_names_the_torchrun_command():
    import run_Gowalla as RG
    with pytest.raises(SystemExit) as e:
        RG.parse(["--dataset", "ml100k", "--parallel", "True", "--epochs", "3"])
    msg = str(e.value)
    assert "torch.distributed.run" in msg and "--nproc-per-node" in msg and "run_Gowalla.py --dataset ml100k --epochs 3" in msg
