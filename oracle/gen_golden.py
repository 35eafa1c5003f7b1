#!/usr/bin/env python3
"""Golden-vector generator: runs the *reference itself* (imported read-only from
/root/reference, CPU) and stores inputs + expected outputs as small .npz fixtures
under tests/golden/.  Runs only in the build container; the GPU box never sees the
reference.  Re-run:  python oracle/gen_golden.py

Shims (no reference source is modified or copied):
  * SPUIGACF.py:342,367,370 hard-code device='cuda' -> a TorchFunctionMode maps
    device='cuda' to 'cpu'; models are built with useCuda=False.
  * metrics.py:52,77 use np.asfarray (gone in NumPy 2) -> aliased to np.asarray(float).
  * dropout sites are pinned by replacing the layer *instance's* .dropout attribute
    with a fixed-mask module and by patching F.dropout while a fixed-mask run is on.
"""
import hashlib
import os
import sys

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

if not hasattr(np, "asfarray"):
    np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)

import graphattention.SPUIGACF as R          # noqa: E402
import graphattention.SPGA as RS             # noqa: E402
import graphattention.BPRLoss as RB          # noqa: E402
import graphattention.metrics as RM          # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class CpuDevice(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get("device", None) == "cuda":
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


class FixedMask(torch.nn.Module):
    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask


def toy_graph(U=50, I=40, E=320, seed=7):
    """3 isolated items (37,38,39), user 0 has degree 1, item 0 is a hub."""
    rs = np.random.RandomState(seed)
    pairs = {(0, 0)}
    for u in range(1, U):
        pairs.add((u, int(rs.randint(0, I - 3))))        # every user >= 1 edge
    for u in range(1, U, 2):
        pairs.add((u, 0))                                 # hub item 0
    while len(pairs) < E:
        u, i = int(rs.randint(1, U)), int(rs.randint(0, I - 3))
        pairs.add((u, i))
    arr = np.array(sorted(pairs), dtype=np.int64).T      # row-major sorted = coalesce() order
    return arr


def sd_numpy(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def g1_layers(edge, U, I):
    """G1: one SpUIGraphAttentionLayer, k=8/concat and k=64/no-concat, with e_rowsum/e_colsum."""
    for tag, k, concat in (("k8_concat", 8, True), ("k64_noconcat", 64, False)):
        torch.manual_seed(11 + k)
        layer = R.SpUIGraphAttentionLayer(64, k, dropout=0.0, alpha=0.2, concat=concat).eval()
        X = torch.randn(U + I, 64) * 0.5
        e_t = torch.from_numpy(edge)
        with CpuDevice():
            out = layer(torch.arange(U), torch.arange(I) + U, X, e_t)
        # recompute the two normalisers with the reference's own ops for the fixture
        u_h, i_h = X[:U] @ layer.W_u, X[U:] @ layer.W_i
        eh = torch.cat((u_h[e_t[0]], i_h[e_t[1]]), 1).t()
        ee = torch.exp(-layer.leakyrelu(layer.a.mm(eh).squeeze()))
        A = torch.sparse_coo_tensor(e_t, ee, (U, I))
        rs = torch.sparse.mm(A, torch.ones(I, 1))
        cs = torch.sparse.mm(A.t(), torch.ones(U, 1))
        save("g1_layer_" + tag, X=X.numpy(), W_u=layer.W_u.detach().numpy(), W_i=layer.W_i.detach().numpy(),
             a=layer.a.detach().numpy(), edge=edge, U=U, I=I, out=out.detach().numpy(),
             e_rowsum=rs.detach().numpy().ravel(), e_colsum=cs.detach().numpy().ravel())


def g2_g3_g4_model(edge, U, I, multi, tag):
    """G2/G5 stack output, G3 scores + BPR loss, G4 all parameter grads (droprate 0, train mode),
    fp32 and a float64 re-run for tolerance calibration."""
    cls = R.SPUIMultiGACF if multi else R.SPUIGACF
    rs = np.random.RandomState(3)
    B = 32
    u = rs.randint(0, U, B).astype(np.int64)
    pos = rs.randint(0, I, B).astype(np.int64)
    neg = rs.randint(0, I, B).astype(np.int64)
    out = {}
    state32 = None
    for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.set_default_dtype(dt)
        torch.manual_seed(5)
        m = cls(U, I, 64, [64, 64], 0.0, useCuda=False)
        if state32 is None:
            with torch.no_grad():              # embeddings scaled up so gradients are not ~1e-9
                m.uEmbd.weight.mul_(30.0)
                m.iEmbd.weight.mul_(30.0)
            state32 = {k_: v.clone() for k_, v in m.state_dict().items()}
        else:                                  # identical parameters, promoted to float64
            m.load_state_dict({k_: v.double() for k_, v in state32.items()})
        m.train()
        e_t = torch.from_numpy(edge)
        with CpuDevice():
            _, _, feats = m.getFeatureMat()
            Fout = m.gat(torch.arange(U), torch.arange(I) + U, feats, e_t)
            ps = m(torch.from_numpy(u), torch.from_numpy(pos), e_t)
            ns = m(torch.from_numpy(u), torch.from_numpy(neg), e_t)
            loss = RB.BPRLoss()(ps, ns)
            loss.backward()
        if dt_name == "f32":
            for k_, v in sd_numpy(m).items():
                out["p." + k_] = v
        out["F_" + dt_name] = Fout.detach().numpy()
        out["pos_" + dt_name] = ps.detach().numpy()
        out["neg_" + dt_name] = ns.detach().numpy()
        out["loss_" + dt_name] = loss.detach().numpy()
        for k_, p in m.named_parameters():
            out["g_%s.%s" % (dt_name, k_)] = p.grad.detach().numpy()
    torch.set_default_dtype(torch.float32)
    save("g234_" + tag, edge=edge, U=U, I=I, u=u, pos=pos, neg=neg, **out)


def g4b_fixed_dropout(edge, U, I):
    """Train-mode SPUIGACF with *pinned* dropout masks at all three sites; pins the
    'numerators only' semantics (SPUIGACF.py:375) and the feature dropout (:208,:213)."""
    torch.manual_seed(9)
    p = 0.25
    m = R.SPUIGACF(U, I, 64, [64, 64], p, useCuda=False)
    with torch.no_grad():
        m.uEmbd.weight.mul_(30.0)
        m.iEmbd.weight.mul_(30.0)
    m.train()
    rs = np.random.RandomState(21)
    E = edge.shape[1]
    keep = 1.0 / (1.0 - p)
    emasks = (rs.rand(9, E) >= p).astype(np.float32) * keep
    fmasks = (rs.rand(2, U + I, 64) >= p).astype(np.float32) * keep
    layers = [getattr(m.gat, "attention_%d" % h) for h in range(8)] + [m.gat.out_att]
    for l, em in zip(layers, emasks):
        l.dropout = FixedMask(torch.from_numpy(em))
    calls = {"n": 0}
    real = F.dropout

    def fake(x, p=0.5, training=True, inplace=False):
        i = calls["n"] % 2
        calls["n"] += 1
        return x * torch.from_numpy(fmasks[i])
    B = 32
    u = rs.randint(0, U, B).astype(np.int64)
    it = rs.randint(0, I, B).astype(np.int64)
    R.F.dropout = fake
    try:
        with CpuDevice():
            sc = m(torch.from_numpy(u), torch.from_numpy(it), torch.from_numpy(edge))
            sc.sum().backward()
    finally:
        R.F.dropout = real
    out = {"p." + k_: v for k_, v in sd_numpy(m).items() if "mask" not in k_}
    for k_, p_ in m.named_parameters():
        out["g." + k_] = p_.grad.detach().numpy()
    save("g4b_fixed_dropout", edge=edge, U=U, I=I, u=u, item=it, emasks=emasks, fmasks=fmasks,
         scores=sc.detach().numpy(), **out)


def g6_spga():
    torch.manual_seed(13)
    N = 60
    rs = np.random.RandomState(4)
    adj = (rs.rand(N, N) < 0.08)
    adj[np.arange(N), np.arange(N)] = True               # every row non-empty
    adj_t = torch.from_numpy(adj.astype(np.float32))
    X = torch.randn(N, 64) * 0.5
    res = {}
    for tag, k, concat in (("k8", 8, True), ("k64", 64, False)):
        l = RS.SpGraphAttentionLayer(64, k, dropout=0.0, alpha=0.2, concat=concat).eval()
        out = l(X, adj_t)
        res.update({"W_" + tag: l.W.detach().numpy(), "a_" + tag: l.a.detach().numpy(),
                    "out_" + tag: out.detach().numpy()})
    save("g6_spga_layer", X=X.numpy(), edge=np.stack(np.nonzero(adj)).astype(np.int64), N=N, **res)


def g7_static_spmm(edge, U, I):
    """GPLayer (L + I) @ X with L = mean_adj built by the reference's buildLaplacianMat."""
    import pandas as pd
    from data.loadGowalla import buildLaplacianMat, scipySP_torchSP
    rt = pd.DataFrame({"userId": edge[0], "itemId": edge[1], "rating": np.ones(edge.shape[1])})
    L = buildLaplacianMat(rt, U, I, "mean_adj")
    Lt = scipySP_torchSP(L).coalesce()
    N = U + I
    torch.manual_seed(17)
    X = torch.randn(N, 64)
    eye = torch.sparse_coo_tensor(torch.arange(N).repeat(2, 1), torch.ones(N), (N, N))
    out = torch.sparse.mm(Lt + eye, X)
    save("g7_static_spmm", edge=edge, U=U, I=I, X=X.numpy(), L_row=Lt.indices()[0].numpy(),
         L_col=Lt.indices()[1].numpy(), L_val=Lt.values().numpy(), out=out.numpy())


def g7b_norm_adj(edge, U, I):
    """adj_type 'norm_adj' straight from the reference's buildLaplacianMat ('plain_adj': g7c_plain_adj)."""
    import pandas as pd
    from data.loadGowalla import buildLaplacianMat
    rt = pd.DataFrame({"userId": edge[0], "itemId": edge[1], "rating": np.ones(edge.shape[1])})
    L = buildLaplacianMat(rt, U, I, "norm_adj").tocsr()
    L.sort_indices()
    L = L.tocoo()
    save("g7b_norm_adj", edge=edge, U=U, I=I, row=L.row.astype(np.int64), col=L.col.astype(np.int64), val=L.data.astype(np.float64))


def g7c_plain_adj(edge, U, I):
    """adj_type 'plain_adj' from the reference's buildLaplacianMat, with non-trivial ratings."""
    import pandas as pd
    from data.loadGowalla import buildLaplacianMat
    rating = (np.random.RandomState(5).randint(1, 6, edge.shape[1])).astype(np.float64)
    rt = pd.DataFrame({"userId": edge[0], "itemId": edge[1], "rating": rating})
    L = buildLaplacianMat(rt, U, I, "plain_adj").tocsr()
    L.sort_indices()
    L = L.tocoo()
    save("g7c_plain_adj", edge=edge, U=U, I=I, rating=rating, shape=np.array(L.shape), row=L.row.astype(np.int64),
         col=L.col.astype(np.int64), val=L.data.astype(np.float64))


def g8_metrics():
    import train_eval_Gowalla as TE
    rs = np.random.RandomState(8)
    rvecs = [np.array([1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1]),
             np.zeros(20, dtype=np.int64), np.ones(20, dtype=np.int64),
             np.array([0] * 19 + [1]), (rs.rand(20) < 0.3).astype(np.int64),
             np.array([0, 1, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0])]
    Ks = [1, 5, 10, 20]
    npos = [7, 3, 25, 1, 9, 3]
    res = np.zeros((len(rvecs), 4, 4))
    for j, (r, n) in enumerate(zip(rvecs, npos)):
        perf = TE.get_performance(list(range(n)), list(r), 0.0, Ks)
        for c, key in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
            res[j, c] = perf[key]
    # ranklist_by_heapq with score ties
    ratings = np.round(rs.rand(50), 1)                    # many ties
    test_items = [int(x) for x in rs.permutation(50)[:35]]
    user_pos = [int(x) for x in rs.permutation(50)[:6]]
    r, _ = TE.ranklist_by_heapq(user_pos, test_items, list(ratings), Ks)
    save("g8_metrics", rvecs=np.stack(rvecs), npos=np.array(npos), perf=res, ratings=ratings,
         test_items=np.array(test_items), user_pos=np.array(user_pos), r=np.array(r))


def g9_g10_ml100k():
    """ml100k ui_mat digest (no file is written next to the reference data: the matrix is
    built in memory with the reference's buildLaplacianMat + coalesce) and a medium-graph
    SpUIGAT eval output."""
    import pandas as pd
    from data.loadGowalla import buildLaplacianMat, scipySP_torchSP
    rt = pd.read_table(REF + "/data/1K/u.data", sep="\t", names=["userId", "itemId", "rating", "timestamp"])
    U, I = int(rt["userId"].max()), int(rt["itemId"].max())
    rt["userId"] -= 1
    rt["itemId"] -= 1
    adj = scipySP_torchSP(buildLaplacianMat(rt, U, I, "ui_mat")).coalesce()
    idx = adj.indices().numpy()
    digest = hashlib.sha256(np.ascontiguousarray(idx).tobytes()).hexdigest()
    ml100k_interactions()
    torch.manual_seed(2019)
    m = R.SPUIGACF(U, I, 64, [64, 64], 0.1, useCuda=False).eval()
    emb = (np.random.RandomState(2019).randn(U + I, 64) * 0.01).astype(np.float32)   # N(0, 0.01^2), :27-28
    with torch.no_grad():
        m.uEmbd.weight.copy_(torch.from_numpy(emb[:U]))
        m.iEmbd.weight.copy_(torch.from_numpy(emb[U:]))
    with CpuDevice():
        _, _, feats = m.getFeatureMat()
        Fout = m.gat(torch.arange(U), torch.arange(I) + U, feats, adj.indices())
    Fo = Fout.detach().numpy()
    rows = np.random.RandomState(1).permutation(U + I)[:256]
    gat = {"p." + k_: v for k_, v in sd_numpy(m).items() if k_.startswith("gat.")}
    save("g9_g10_ml100k", shape=np.array([U, I]), nnz=idx.shape[1], sha256=np.array(digest),
         first16=idx[:, :16], last16=idx[:, -16:], emb_seed=2019,
         rows=rows, F_rows=Fo[rows], F_sum=np.float64(Fo.astype(np.float64).sum()),
         F_abs_sum=np.float64(np.abs(Fo.astype(np.float64)).sum()), **gat)


def g13_ref_checkpoint(edge, U, I):
    """A checkpoint as the REFERENCE writes it (run_Gowalla.py:142-143: {'model': state_dict, 'optim': Adam state_dict})
    after two training steps, the batch of a third step, and the reference's parameters after that third step: the
    resume test loads the checkpoint into this package's model + optimizer (different parameter layout) and must land on
    the same parameters.  droprate 0 (masks cannot be matched); SPUIMultiGACF so that two fused 8-head stages are covered."""
    torch.manual_seed(23)
    m = R.SPUIMultiGACF(U, I, 64, [64, 64], 0.0, useCuda=False)
    with torch.no_grad():
        m.uEmbd.weight.mul_(30.0)
        m.iEmbd.weight.mul_(30.0)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=1e-5)             # run_Gowalla.py:114
    rs = np.random.RandomState(29)
    e_t = torch.from_numpy(edge)
    B = 48
    batches = [(rs.randint(0, U, B).astype(np.int64), rs.randint(0, I, B).astype(np.int64), rs.randint(0, I, B).astype(np.int64))
               for _ in range(3)]

    def step(b):
        u, p_, n = (torch.from_numpy(x) for x in b)
        opt.zero_grad()
        with CpuDevice():
            loss = RB.BPRLoss()(m(u, p_, e_t), m(u, n, e_t))
        loss.backward()
        opt.step()
        return float(loss)
    losses = [step(batches[0]), step(batches[1])]
    out = {"p." + k_: v for k_, v in sd_numpy(m).items()}
    osd = opt.state_dict()
    names = [k_ for k_, _ in m.named_parameters()]
    for j, pid in enumerate(osd["param_groups"][0]["params"]):
        st = osd["state"][pid]
        out["o.%d.exp_avg" % j] = st["exp_avg"].numpy().copy()
        out["o.%d.exp_avg_sq" % j] = st["exp_avg_sq"].numpy().copy()
        out["o.%d.step" % j] = np.float64(float(st["step"]))
    g = osd["param_groups"][0]
    losses.append(step(batches[2]))
    out.update({"q." + k_: v for k_, v in sd_numpy(m).items()})                    # parameters after the third step
    save("g13_ref_checkpoint", edge=edge, U=U, I=I, u=batches[2][0], pos=batches[2][1], neg=batches[2][2],
         lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
         param_names=np.array(names), losses=np.array(losses), **out)


def g14_dense_mha(edge, U, I):
    """N4: the dense multi-head attention of graphattention/modules.py -- MultiHeadAttention (:62-107) with
    and without mask (sizes that are not multiples of 32; one fully masked query row -> uniform attention), outputs and
    float64 gradients of every input and weight; ATTLayer (:7-18); and the two models built on it: GACFV1
    (GACFmodel1.py) and GACFMask (GACFMask.py) forward scores on the toy graph with the reference's mean_adj Laplacian."""
    import pandas as pd
    import graphattention.modules as RMod
    import graphattention.GACFmodel1 as G1
    import graphattention.GACFMask as GM
    from data.loadGowalla import buildLaplacianMat, scipySP_torchSP
    rs = np.random.RandomState(31)
    Nq, Nk, dim, heads = 45, 70, 64, 8
    out = {}
    q_in, k_in = (rs.randn(Nq, dim) * 0.7).astype(np.float32).astype(np.float64), (rs.randn(Nk, dim) * 0.7).astype(np.float32).astype(np.float64)
    mask = (rs.rand(Nq, Nk) < 0.3).astype(np.float32)
    mask[3] = 0.0                                           # a fully masked row: softmax over -1e9 everywhere = uniform
    mask[np.arange(Nq), rs.randint(0, Nk, Nq)] = np.where(np.arange(Nq) == 3, 0.0, 1.0)
    G = rs.randn(Nq, dim).astype(np.float32).astype(np.float64)
    torch.manual_seed(37)
    mha32 = RMod.MultiHeadAttention(heads, dim, dropout=0.0)
    with torch.no_grad():
        for lin in (mha32.q_linear, mha32.k_linear, mha32.v_linear, mha32.out):
            lin.bias.normal_(std=0.1)                       # the reference zeroes them at init; trained values are not zero
    for k_, v in mha32.state_dict().items():
        out["mha.p." + k_] = v.numpy().copy()
    for tag, mk in (("nomask", None), ("mask", mask)):
        for dt_name, dt in (("f32", torch.float32), ("f64", torch.float64)):
            m = RMod.MultiHeadAttention(heads, dim, dropout=0.0).to(dt)
            m.load_state_dict({k_: v.to(dt) for k_, v in mha32.state_dict().items()})
            q = torch.tensor(q_in, dtype=dt, requires_grad=True)
            kv = torch.tensor(k_in, dtype=dt, requires_grad=True)
            o = m(q, kv, kv, None if mk is None else torch.tensor(mk, dtype=dt))
            out["mha.%s.out_%s" % (tag, dt_name)] = o.detach().numpy()
            if dt_name == "f64":
                (o * torch.tensor(G)).sum().backward()
                out["mha.%s.g.q_in" % tag] = q.grad.numpy().astype(np.float32)      # (float64 values, stored as float32)
                out["mha.%s.g.kv_in" % tag] = kv.grad.numpy().astype(np.float32)
                for k_, p_ in m.named_parameters():
                    out["mha.%s.g.%s" % (tag, k_)] = p_.grad.numpy().astype(np.float32)
    # ATTLayer: users attend to items and items to users (:14)
    torch.manual_seed(41)
    att = RMod.ATTLayer(dim, heads, 0.0)
    for k_, v in att.state_dict().items():
        out["att.p." + k_] = v.numpy().copy()
    out["att.out"] = att(torch.tensor(q_in, dtype=torch.float32), torch.tensor(k_in, dtype=torch.float32)).detach().numpy()
    # the two model families on the toy graph
    e37 = edge[:, edge[1] < 37]
    I37 = 37
    rt = pd.DataFrame({"userId": e37[0], "itemId": e37[1], "rating": np.ones(e37.shape[1])})
    L = scipySP_torchSP(buildLaplacianMat(rt, U, I37, "mean_adj")).coalesce()
    N = U + I37
    dense_adj = torch.zeros(N, N)
    dense_adj[torch.from_numpy(e37[0]), torch.from_numpy(e37[1]) + U] = 1.0
    dense_adj = dense_adj + dense_adj.t() + torch.eye(N)     # a node attends to its neighbours and itself
    B = 40
    u = rs.randint(0, U, B).astype(np.int64)
    it = rs.randint(0, I37, B).astype(np.int64)
    real_cuda = (torch.Tensor.cuda, torch.nn.Module.cuda)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        with CpuDevice():
            torch.manual_seed(43)
            m1 = G1.GACFV1(U, I37, L, 64, [64, 64], 0.0, useCuda=False)
            m1.gacf.useCuda = False
            with torch.no_grad():
                m1.gacf.uEmbd.weight.mul_(30.0); m1.gacf.iEmbd.weight.mul_(30.0)
            m1.eval()
            s1 = m1(torch.from_numpy(u), torch.from_numpy(it))
            torch.manual_seed(47)
            m2 = GM.GACFMask(U, I37, L, 64, [64, 64], 0.0, useCuda=False)
            m2.gacf.useCuda = False
            with torch.no_grad():
                m2.gacf.uEmbd.weight.mul_(30.0); m2.gacf.iEmbd.weight.mul_(30.0)
            m2.eval()
            s2 = m2(torch.from_numpy(u), torch.from_numpy(it), dense_adj)
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = real_cuda
    for k_, v in m1.state_dict().items():
        out["v1.p." + k_] = v.numpy().copy()
    for k_, v in m2.state_dict().items():
        out["mk.p." + k_] = v.numpy().copy()
    save("g14_dense_mha", q_in=q_in.astype(np.float32), k_in=k_in.astype(np.float32), mask=mask.astype(np.uint8), G=G.astype(np.float32), heads=heads, edge=e37, U=U, I=I37,
         L_row=L.indices()[0].numpy(), L_col=L.indices()[1].numpy(), L_val=L.values().numpy(),
         dense_adj=dense_adj.numpy().astype(np.uint8), u=u, item=it, v1_scores=s1.detach().numpy(), mk_scores=s2.detach().numpy(), **out)


def g15_spga_models(edge, U, I):
    """The SPGA family's model wrappers (SPGA.py:85-316): SPGACF, MultiLayerSPGA, SPGAMGP forward scores on the toy graph
    (mask = dense bipartite adjacency + self loops over N nodes, Laplacian = the reference's mean_adj), eval mode."""
    import pandas as pd
    from data.loadGowalla import buildLaplacianMat, scipySP_torchSP
    e37 = edge[:, edge[1] < 37]
    I37 = 37
    rt = pd.DataFrame({"userId": e37[0], "itemId": e37[1], "rating": np.ones(e37.shape[1])})
    L = scipySP_torchSP(buildLaplacianMat(rt, U, I37, "mean_adj")).coalesce()
    N = U + I37
    dense = torch.zeros(N, N)
    dense[torch.from_numpy(e37[0]), torch.from_numpy(e37[1]) + U] = 1.0
    dense = dense + dense.t() + torch.eye(N)
    rs = np.random.RandomState(53)
    B = 40
    u, it = rs.randint(0, U, B).astype(np.int64), rs.randint(0, I37, B).astype(np.int64)
    out = {}
    real_cuda = (torch.Tensor.cuda, torch.nn.Module.cuda)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        with CpuDevice():
            for tag, cls, layers in (("spgacf", RS.SPGACF, [64, 64]), ("multi", RS.MultiLayerSPGA, [64]), ("mgp", RS.SPGAMGP, [64, 64])):
                torch.manual_seed(59)
                m = cls(U, I37, L, 64, list(layers), 0.0, useCuda=False)
                with torch.no_grad():
                    m.uEmbd.weight.mul_(30.0); m.iEmbd.weight.mul_(30.0)
                m.eval()
                sc = m(torch.from_numpy(u), torch.from_numpy(it), dense)
                out[tag + ".scores"] = sc.detach().numpy()
                for k_, v in m.state_dict().items():
                    out[tag + ".p." + k_] = v.numpy().copy()
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = real_cuda
    save("g15_spga_models", edge=e37, U=U, I=I37, L_row=L.indices()[0].numpy(), L_col=L.indices()[1].numpy(),
         L_val=L.values().numpy(), dense_adj=dense.numpy().astype(np.uint8), u=u, item=it, **out)


def g16_gacf_variants(edge, U, I):
    """GACFV2 .. GACFV5 (GACFmodel2-5.py) forward scores on the toy graph.  V3/V4/V5 have GACFV1's parameter set, so they
    are run on the parameters already stored in g14 (v1.p.*): only their scores are new; V2 brings its own."""
    import pandas as pd
    import graphattention.GACFmodel2 as G2
    import graphattention.GACFmodel3 as G3
    import graphattention.GACFmodel4 as G4
    import graphattention.GACFmodel5 as G5
    from data.loadGowalla import buildLaplacianMat, scipySP_torchSP
    g14 = np.load(os.path.join(OUT, "g14_dense_mha.npz"))
    e37, I37 = g14["edge"], int(g14["I"])
    rt = pd.DataFrame({"userId": e37[0], "itemId": e37[1], "rating": np.ones(e37.shape[1])})
    L = scipySP_torchSP(buildLaplacianMat(rt, U, I37, "mean_adj")).coalesce()
    u, it = torch.from_numpy(g14["u"]), torch.from_numpy(g14["item"])
    v1 = {k_[len("v1.p."):]: torch.from_numpy(g14[k_]) for k_ in g14.files if k_.startswith("v1.p.")}
    out = {}
    real_cuda = (torch.Tensor.cuda, torch.nn.Module.cuda)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        with CpuDevice():
            for tag, cls in (("v3", G3.GACFV3), ("v4", G4.GACFV4), ("v5", G5.GACFV5)):
                m = cls(U, I37, L, 64, [64, 64], 0.0)
                m.gacf.useCuda = False
                m.load_state_dict(v1)
                m.eval()
                out[tag + "_scores"] = m(u, it).detach().numpy()
            torch.manual_seed(61)
            m2 = G2.GACFV2(U, I37, L, 64, [64, 64], 0.0, useCuda=False)
            m2.gacf.useCuda = False
            with torch.no_grad():
                m2.gacf.uEmbd.weight.mul_(30.0); m2.gacf.iEmbd.weight.mul_(30.0)
            m2.eval()
            out["v2_scores"] = m2(u, it).detach().numpy()
            for k_, v in m2.state_dict().items():
                out["v2.p." + k_] = v.numpy().copy()
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = real_cuda
    save("g16_gacf_variants", **out)


def g17_gacf_mlp_heads(edge, U, I):
    """GACFV6 (GACFmodel6.py:45-110: the Laplacian built from the rating table by the model itself, :139-161; attention fed
    straight into affine2, no ReLU inside the layer) and GACFV3x (GACFmodel3.py:114-200: GACFV3's layers, ReLU around each, the
    same three-layer MLP head on cat(user row, item row))."""
    import pandas as pd
    import graphattention.GACFmodel3 as G3
    import graphattention.GACFmodel6 as G6
    from data.loadGowalla import buildLaplacianMat, scipySP_torchSP
    g14 = np.load(os.path.join(OUT, "g14_dense_mha.npz"))
    e37, I37 = g14["edge"], int(g14["I"])
    rs = np.random.RandomState(17)
    rating = rs.randint(1, 6, e37.shape[1]).astype(np.float64)                     # ml100k-style ratings 1..5
    rt = pd.DataFrame({"userId": e37[0], "itemId": e37[1], "rating": rating})
    rt1 = pd.DataFrame({"userId": e37[0], "itemId": e37[1], "rating": np.ones(e37.shape[1])})
    L = scipySP_torchSP(buildLaplacianMat(rt1, U, I37, "mean_adj")).coalesce()
    u, it = torch.from_numpy(g14["u"]), torch.from_numpy(g14["item"])
    out = {"rating": rating.astype(np.float32)}
    real_cuda = (torch.Tensor.cuda, torch.nn.Module.cuda)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        with CpuDevice():
            for tag, make in (("v6", lambda: G6.GACFV6(U, I37, rt, 64, [64, 64, 64], 0.0, useCuda=False)),
                              ("v3x", lambda: G3.GACFV3x(U, I37, L, 64, [64, 64, 64], 0.0, useCuda=False))):
                torch.manual_seed(71)
                m = make()
                emb = m
                emb.useCuda = False
                with torch.no_grad():
                    emb.uEmbd.weight.mul_(30.0); emb.iEmbd.weight.mul_(30.0)
                m.eval()
                out[tag + "_scores"] = m(u, it).detach().numpy()
                for k_, v in m.state_dict().items():
                    out[tag + ".p." + k_] = v.numpy().copy()
                if tag == "v6":                                                   # the Laplacian the model built (GACFmodel6.py:139-161)
                    Lm = m.LaplacianMat.coalesce()
                    out["v6_L_row"], out["v6_L_col"], out["v6_L_val"] = Lm.indices()[0].numpy(), Lm.indices()[1].numpy(), Lm.values().numpy()
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda = real_cuda
    save("g17_gacf_mlp_heads", **out)


def g11_factors(U, I, seed=7):
    rs = np.random.RandomState(seed)
    cont = (rs.randn(U, 64).astype(np.float32).astype(np.float64), rs.randn(I, 64).astype(np.float32).astype(np.float64))
    ties = (rs.randint(-2, 3, (U, 64)).astype(np.float64), rs.randint(-2, 3, (I, 64)).astype(np.float64))
    return {"cont": cont, "ties": ties}


def g11_eval_neg_all():
    """A8 pin (ii): the reference's OWN eval_neg_all (train_eval_Gowalla.py:274-354: user batches of 64,
    item batches, report_one_user in a process pool, heapq.nlargest, metrics.py) on a FIXED score table, for the ml100k
    split the reference's runner makes: seeds as run_Gowalla.py:191-193, then train_test_split(rt, test_size=0.2) (:80),
    train_pos_neg_exclude_test (:85), test_positives (:87).  The model is a stand-in that returns S[u, i] from a seeded
    table -- once with continuous scores, once with small-integer scores (thousands of exact ties: pins the tie order of
    heapq.nlargest over the reference's Python sets).  Shims: Tensor.cuda -> identity, torch.cuda.empty_cache -> no-op."""
    import pandas as pd
    from sklearn.model_selection import train_test_split
    import train_eval_Gowalla as TE
    from data.loadGowalla import load100KRatings, train_pos_neg_exclude_test, test_positives
    seed = 2019
    torch.manual_seed(seed)                               # run_Gowalla.py:191-193
    np.random.seed(seed)
    rt = load100KRatings(REF + "/data/1K")                # read-only; get_adj_mat (which writes an .npz there) is not called
    U, I = int(rt["userId"].max()), int(rt["itemId"].max())
    rt["userId"] = rt["userId"] - 1
    rt["itemId"] = rt["itemId"] - 1
    train_df, test_df_rows = train_test_split(rt, test_size=0.2)                       # :80
    train_pos_neg = train_pos_neg_exclude_test(rt, train_df)                           # :85
    test_df = test_positives(test_df_rows)                                             # :87
    # score tables of rank 64, S = A @ B^T, so that the device path can be fed the same scores as propagated embeddings
    # F = [A; B] (tests/test_gpu_train_eval.py); integer factors make the tie table exact in fp32 on both sides
    tables = {k: (a @ b.T).astype(np.float32) for k, (a, b) in g11_factors(U, I).items()}

    class Table(torch.nn.Module):
        def __init__(self, t):
            super().__init__()
            self.t = torch.from_numpy(t)

        def forward(self, u, i, adj):
            return self.t[u, i]
    real_cuda, real_empty = torch.Tensor.cuda, torch.cuda.empty_cache
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    out = {}
    try:
        for tag, tab in tables.items():
            res = TE.eval_neg_all(Table(tab), 2048, test_df, train_pos_neg, np.zeros((2, 1), dtype=np.int64), I, False)
            for k in ("precision", "recall", "ndcg", "hit_ratio"):
                out["%s.%s" % (tag, k)] = np.asarray(res[k], dtype=np.float64)
            out["%s.auc" % tag] = np.float64(res["auc"])
    finally:
        torch.Tensor.cuda, torch.cuda.empty_cache = real_cuda, real_empty
    save("g11_eval_neg_all", U=U, I=I, seed=seed, table_seed=7,
         train_rows=train_df.index.values.astype(np.int32), test_rows=test_df_rows.index.values.astype(np.int32),
         n_train_users=len(train_pos_neg), n_test_users=len(test_df),
         test_users_order=test_df["userId"].values.astype(np.int32), **out)


def ml100k_interactions():
    """the raw interaction table in file order (data, small ints) so the GPU box can rebuild everything; with the timestamps
    (split_loo, data/loadGowalla.py:307-313, ranks a user's rows by them)"""
    import pandas as pd
    rt = pd.read_table(REF + "/data/1K/u.data", sep="\t", names=["userId", "itemId", "rating", "timestamp"])
    save("ml100k_interactions", userId=(rt["userId"].values - 1).astype(np.int16), itemId=(rt["itemId"].values - 1).astype(np.int16),
         rating=rt["rating"].values.astype(np.int8), timestamp=rt["timestamp"].values.astype(np.int32))


def g18_neg_sampling_modes():
    """The reference's OTHER runner modes on fixed score tables (run_Gowalla.py --train_mode NegSampling --eval_mode SampledNeg):
    its own split_loo (data/loadGowalla.py:307-313), positives_negtives (:63-67), train_neg_sample (train_eval_Gowalla.py:36-88:
    4 sampled negatives per train row, BCEWithLogitsLoss, total_loss / len(train_df)) and eval_neg_sample (:193-259: 99 sampled
    negatives per test row, top-10 of the 100 scores, HR / NDCG of graphattention/evaluation.py:4-13) on the ml100k table.
    The model is a stand-in returning S[u, i] (+ 0 * w, so that the loop's backward and optimizer step have something to do);
    the negatives the reference's random.sample draws are RECORDED (wrappers around train_neg_sampling / test_neg_sampling), so
    that another implementation can be fed the same samples.  Shims: Tensor.cuda -> identity, torch.cuda.empty_cache -> no-op,
    torch.cuda.device_count -> 1, Tensor.backward(ones(1)) on the 0-dim loss -> backward()."""
    import random
    import pandas as pd
    import train_eval_Gowalla as TE
    from data.loadGowalla import load100KRatings, split_loo, positives_negtives
    seed = 2019
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)                                     # (the reference leaves `random` unseeded; the draws are recorded anyway)
    rt = load100KRatings(REF + "/data/1K")
    U, I = int(rt["userId"].max()), int(rt["itemId"].max())
    rt["userId"] = rt["userId"] - 1
    rt["itemId"] = rt["itemId"] - 1
    train_df, test_df = split_loo(rt)                     # run_Gowalla.py:77
    pos_neg = positives_negtives(rt)                      # :90
    A, B = g11_factors(U, I)["cont"]
    table = (0.25 * (A @ B.T)).astype(np.float32)         # (logits of a few units: the BCE terms are neither 0 nor saturated)

    class Table(torch.nn.Module):
        def __init__(self, t):
            super().__init__()
            self.t = torch.from_numpy(t)
            self.w = torch.nn.Parameter(torch.zeros(()))

        def forward(self, u, i, adj):
            return self.t[u, i] + 0.0 * self.w
    rec = {"train": [], "test": []}
    real_tr, real_te = TE.train_neg_sampling, TE.test_neg_sampling

    def tr_wrap(df, pn):
        out = real_tr(df, pn)
        rec["train"].append(np.stack(out["neg_samples"].values).astype(np.int32))
        assert (out["userId"].values == df["userId"].values).all() and (out["itemId"].values == df["itemId"].values).all()
        return out

    def te_wrap(df, pn):
        out, n = real_te(df, pn)
        rec["test"].append(np.stack(out["neg_samples"].values).astype(np.int32))
        assert (out["userId"].values == df["userId"].values).all() and (out["itemId"].values == df["itemId"].values).all()
        return out, n
    real = (torch.Tensor.cuda, torch.cuda.empty_cache, torch.cuda.device_count, torch.Tensor.backward)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    torch.cuda.device_count = lambda: 1
    torch.Tensor.backward = lambda self, gradient=None, *a, **k: real[3](self, None if (gradient is not None and self.dim() == 0) else gradient, *a, **k)
    TE.train_neg_sampling, TE.test_neg_sampling = tr_wrap, te_wrap
    try:
        model = Table(table)
        optim = torch.optim.Adam(model.parameters(), lr=1e-3)
        train_loss = TE.train_neg_sample(model, 2048, train_df, pos_neg, np.zeros((2, 1), dtype=np.int64), optim, torch.nn.BCEWithLogitsLoss(), False)
        HR, NDCG = TE.eval_neg_sample(model, 2048, test_df, pos_neg, np.zeros((2, 1), dtype=np.int64), 10, False)
    finally:
        torch.Tensor.cuda, torch.cuda.empty_cache, torch.cuda.device_count, torch.Tensor.backward = real
        TE.train_neg_sampling, TE.test_neg_sampling = real_tr, real_te
    save("g18_neg_sampling_modes", U=U, I=I, table_seed=7, table_scale=0.25,
         train_rows=train_df.index.values.astype(np.int32), test_rows=test_df.index.values.astype(np.int32),
         train_negs=np.concatenate(rec["train"]), test_negs=np.concatenate(rec["test"]),
         train_loss=np.float64(train_loss), HR=np.float64(HR), NDCG=np.float64(NDCG), batch_size=2048)
    print("g18: train rows %d, test rows %d, loss %.8f, HR %.6f, NDCG %.6f" % (len(train_df), len(test_df), train_loss, HR, NDCG))


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-g11" in sys.argv:
        return g11_eval_neg_all()
    if "--only-g18" in sys.argv:
        return g18_neg_sampling_modes()
    if "--only-ml100k-interactions" in sys.argv:
        return ml100k_interactions()
    edge = toy_graph()
    U, I = 50, 40
    if "--only-g7c" in sys.argv:
        return g7c_plain_adj(edge[:, edge[1] < 37], U, 37)
    if "--only-g13" in sys.argv:
        return g13_ref_checkpoint(edge, U, I)
    if "--only-g14" in sys.argv:
        return g14_dense_mha(edge, U, I)
    if "--only-g15" in sys.argv:
        return g15_spga_models(edge, U, I)
    if "--only-g16" in sys.argv:
        return g16_gacf_variants(edge, U, I)
    if "--only-g17" in sys.argv:
        return g17_gacf_mlp_heads(edge, U, I)
    if "--only-g7b" in sys.argv:                      # added after the first batch: leaves the other fixtures untouched
        return g7b_norm_adj(edge[:, edge[1] < 37], U, 37)
    g1_layers(edge, U, I)
    g2_g3_g4_model(edge, U, I, multi=False, tag="spuigacf")
    g2_g3_g4_model(edge, U, I, multi=True, tag="spuimultigacf")
    g4b_fixed_dropout(edge, U, I)
    g6_spga()
    g7_static_spmm(edge[:, edge[1] < 37], U, 37)      # buildLaplacianMat infers shapes from max ids
    g7b_norm_adj(edge[:, edge[1] < 37], U, 37)
    g7c_plain_adj(edge[:, edge[1] < 37], U, 37)
    g8_metrics()
    g9_g10_ml100k()
    g11_eval_neg_all()
    g13_ref_checkpoint(edge, U, I)
    g14_dense_mha(edge, U, I)
    g15_spga_models(edge, U, I)
    g16_gacf_variants(edge, U, I)
    g17_gacf_mlp_heads(edge, U, I)
    g18_neg_sampling_modes()


if __name__ == "__main__":
    main()
