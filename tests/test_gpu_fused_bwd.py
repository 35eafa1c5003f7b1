"""The backward's fused launch plan (ngacf_prop_bwd_dx: the stage's dX and the previous stage's step 1 in the edge backward's
tail) against the separate launches it replaces -- bit for bit at the level of one stage, and through the whole stack.
Both plans are checked against the reference's float64 gradients by the golden tests; this is their A/B.
Reference: autograd of graphattention/SPUIGACF.py:356-397 chained through :207-215."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _graph(U, I, E, seed, hub=0):
    rs = np.random.RandomState(seed)
    u = np.concatenate([np.arange(U), rs.randint(0, U, E - U)])
    i = np.concatenate([rs.randint(0, I - 3, U), rs.randint(0, I - 3, E - U)])     # last 3 items isolated
    if hub:
        u = np.concatenate([u, rs.permutation(U)[:hub]])
        i = np.concatenate([i, np.zeros(hub, dtype=np.int64)])
    key = np.unique(u.astype(np.int64) * I + i)
    return np.stack([key // I, key % I])


def _bits(t):
    return t.detach().contiguous().view(torch.uint8)


@pytest.mark.parametrize("heads", [8, 1])
@pytest.mark.parametrize("drop", [0.0, 0.2])
@pytest.mark.parametrize("sparse", [False, True])
def test_edge_backward_with_dx_bitwise(dev, heads, drop, sparse):
    """One stage: prop_bwd(dx=...) against prop_bwd + transform_bwd on the same inputs: dH, dS, dX (bf16 and fp32 rows), dW, da,
    every row class (short rows, long rows, hub rows finished in-kernel, isolated items), with and without dropout, over all rows
    and over a scored batch's rows (the batch lists duplicates).  (Round 4's form with the previous stage's node step in the same
    tail was deleted in round 5: worth nothing inside the train step.)"""
    from ngacf_amd import ops
    from ngacf_amd.plan import build_plan
    U, I = 1100, 901
    N = U + I
    edge = torch.from_numpy(_graph(U, I, 45000, 5, hub=700)).to(dev)
    plan = build_plan(edge, U, I, seg_len=128)
    assert plan.nsplit > 0 and plan.nseg > 0 and plan.nshort > 0
    rs = np.random.RandomState(11 + heads)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    bf = torch.bfloat16
    X = t(rs.randn(N, 64).astype(np.float32) * 0.5, bf)                 # the stage's input = the previous stage's output
    W = t((rs.randn(2, 64, 64) * 0.2).astype(np.float32))
    a = t((rs.randn(2, 64) * 0.3).astype(np.float32))
    H, S = ops.transform_fwd(X, W, a, U, I, heads, bf, feat_drop_p=drop, seed=5)
    rows = None
    if sparse:
        g = torch.Generator(device=dev); g.manual_seed(3)
        B = 203                                                          # odd: the row list's user / item seam falls inside a workgroup
        bu, bi = torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I, (B,), device=dev, generator=g)
        bu[:7] = bu[7:14]                                                # duplicates
        rows = ops.mark_rows(bu, bi, U, I, grad_d=64)
        rows.g32.zero_()
        gsel = torch.cat([bu, U + bi])
        rows.g32[gsel] = t(rs.randn(2 * B, 64).astype(np.float32) * 0.1)
        gout = rows.g32
    else:
        gout = t(rs.randn(N, 64).astype(np.float32) * 0.1, bf)
    out, agg, R = ops.prop_fwd(plan, H, S, S, a, heads, drop_p=drop, seed=9, save=True, needed=rows)
    kw = dict(drop_p=drop, seed=9, needed=rows, gout_f32=sparse)
    for x_dtype in (bf, torch.float32):
        Xs = X if x_dtype == bf else X.float()
        # the separate launches
        dH, dS = ops.prop_bwd(plan, H, S, S, a, out, agg, R, gout, heads, **kw)
        dX, dW, da = ops.transform_bwd(Xs, W, H, dH, dS, U, I, heads, drop, 5)
        # the fused launch
        panel = ops.pack_panels([W], kinds=[1])[0]
        dx = ops.BwdDx(panel, x_dtype, drop, 5)
        dH2, dS2 = ops.prop_bwd(plan, H, S, S, a, out, agg, R, gout, heads, dx=dx, **kw)
        _, dW2, da2 = ops.transform_bwd(Xs, W, H, dH2, dS2, U, I, heads, drop, 5, part="dw")
        torch.cuda.synchronize()
        assert torch.equal(_bits(dH), _bits(dH2)) and torch.equal(_bits(dS), _bits(dS2))
        assert torch.equal(_bits(dX), _bits(dx.dX)), (x_dtype, float((dX.float() - dx.dX.float()).abs().max()))
        assert torch.equal(_bits(dW), _bits(dW2)) and torch.equal(_bits(da), _bits(da2))


@pytest.mark.parametrize("kind", ["spuigacf", "multi"])
@pytest.mark.parametrize("sparse_last", [True, False])
@pytest.mark.parametrize("level", [1])
def test_fused_backward_stack_equals_unfused_backward_stack(dev, kind, sparse_last, level):
    """The whole train step's gradients with NGACF_FUSE_BWD = 1 (dX in the edge backward's tail) against 0 (separate launches),
    bf16 storage, dropout on (same seeds = same masks): the loss is bitwise the same; gradients to 1e-5 (the batched and the
    per-stage weight-gradient launches reduce in different orders)."""
    import ngacf_amd
    from ngacf_amd import ops
    U, I = 900, 700
    edge = torch.from_numpy(_graph(U, I, 40000, 3, hub=500)).to(dev)
    mk = {"spuigacf": lambda: ngacf_amd.SPUIGACF(U, I, 64, [64, 64], 0.1),
          "multi": lambda: ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.1)}[kind]
    torch.manual_seed(4)
    m = mk().to(dev)
    with torch.no_grad():
        m.uEmbd.weight.mul_(30.0); m.iEmbd.weight.mul_(30.0)
    ngacf_amd.set_compute_dtype(m, torch.bfloat16)
    m.sparse_last_stage = sparse_last
    g = torch.Generator(device=dev); g.manual_seed(1)
    u, pi, ni = (torch.randint(0, k, (255,), device=dev, generator=g) for k in (U, I, I))
    from ngacf_amd import functional as NF
    res = {}
    was, was_sp = ops.FUSE_BWD, NF.FUSE_SPARSE
    NF.FUSE_SPARSE = True                                     # (off by default: the stage of a scored batch; tested here)
    try:
        for fuse in (True, False):
            ops.FUSE_BWD = level if fuse else 0
            m.train()
            m.zero_grad(set_to_none=True)
            torch.manual_seed(9)
            n0 = _count_dx_calls()
            loss = ngacf_amd.BPRLoss()(m(u, pi, edge), m(u, ni, edge))
            loss.backward()
            torch.cuda.synchronize()
            res[fuse] = (float(loss.detach()), {n: p.grad.clone() for n, p in m.named_parameters()}, _count_dx_calls() - n0)
    finally:
        ops.FUSE_BWD, NF.FUSE_SPARSE = was, was_sp
    L = 2 if kind == "spuigacf" else 3
    assert res[True][2] == 2 * L and res[False][2] == 0, (res[True][2], res[False][2])     # every stage of both chains took the fused launch
    assert res[True][0] == res[False][0]
    for n in res[True][1]:
        a_, b_ = res[True][1][n].double(), res[False][1][n].double()
        assert float((a_ - b_).norm() / b_.norm().clamp_min(1e-30)) < 1e-5, n


_DX_CALLS = [0]


def _count_dx_calls():
    return _DX_CALLS[0]


@pytest.fixture(autouse=True)
def _spy_on_dx(monkeypatch):
    """count the fused launches (so the A/B test knows the fused plan really ran)"""
    from ngacf_amd import ops
    real = ops.prop_bwd

    def spy(*a, **k):
        if k.get("dx") is not None:
            _DX_CALLS[0] += 1
        return real(*a, **k)
    monkeypatch.setattr(ops, "prop_bwd", spy)
    yield


def test_batched_weight_gradients_equal_the_per_stage_launches(dev):
    """ngacf_transform_dw_batch (three stages' dW / davec in one launch + one reduction; fp32 and bf16 inputs, 8 heads and 1, with
    and without feature dropout) against ngacf_transform_bwd's weight half: the same per-workgroup arithmetic, a different number
    of partial sums -> equal to fp32 summation noise."""
    from ngacf_amd import ops
    U, I = 3100, 2501
    N = U + I
    g = torch.Generator(device=dev); g.manual_seed(5)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    bf = torch.bfloat16
    jobs, want = [], []
    for x_dtype, heads, p in ((torch.float32, 8, 0.2), (bf, 8, 0.0), (bf, 1, 0.1), (bf, 8, 0.3)):
        X = (rnd(N, 64) * 0.5).to(x_dtype)
        W = rnd(2, 64, 64) * 0.2
        H, dH = (rnd(N, 64) * 0.5).to(bf), (rnd(N, 64) * 0.1).to(bf)
        dS = rnd(N, heads) * 0.1
        _, dW, da = ops.transform_bwd(X, W, H, dH, dS, U, I, heads, p, 17, part="dw")
        want.append((dW, da))
        jobs.append((X, H, dH, dS, heads, p, 17, torch.empty(2, 64, 64, device=dev), torch.empty(2, 64, device=dev)))
    ops.transform_dw_batch(jobs, U, I)
    torch.cuda.synchronize()
    for (dW, da), j in zip(want, jobs):
        for a_, b_ in ((dW, j[7]), (da, j[8])):
            assert float((a_.double() - b_.double()).norm() / a_.double().norm()) < 2e-6


def test_batched_weight_gradients_fp32_storage(dev):
    """The fp32-storage form of ngacf_transform_dw_batch (exact fp32 matrix cores, v_mfma_f32_16x16x4_f32) against ngacf_transform_bwd's
    weight half and against float64: 8 heads and 1, with and without feature dropout, row counts that are no multiple of the tile;
    a launch that mixes storage types is refused."""
    from ngacf_amd import ops, rng
    U, I = 3100, 2501
    N = U + I
    g = torch.Generator(device=dev); g.manual_seed(6)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    jobs, want = [], []
    for heads, p in ((8, 0.2), (1, 0.0), (8, 0.0)):
        X, W = rnd(N, 64) * 0.5, rnd(2, 64, 64) * 0.2
        H, dH, dS = rnd(N, 64) * 0.5, rnd(N, 64) * 0.1, rnd(N, heads) * 0.1
        _, dW, da = ops.transform_bwd(X, W, H, dH, dS, U, I, heads, p, 23, part="dw")
        Xm = X.double()
        if p > 0:
            n_idx = torch.arange(N, device=dev, dtype=torch.int32).repeat_interleave(64)
            c_idx = torch.arange(64, device=dev, dtype=torch.int32).repeat(N)
            keep = ((ops.dropout_bits(0, 23, n_idx, c_idx).long() & 0xFFFFFFFF) >= int(rng.threshold(p))).view(N, 64)
            Xm = Xm * keep / (1 - p)
        ref_dW = torch.stack([Xm[:U].t() @ dH[:U].double(), Xm[U:].t() @ dH[U:].double()])
        ref_da = torch.stack([(dS[:U].double().repeat_interleave(64 // heads, 1) * H[:U].double()).sum(0),
                              (dS[U:].double().repeat_interleave(64 // heads, 1) * H[U:].double()).sum(0)])
        want.append((dW, da, ref_dW, ref_da))
        jobs.append((X, H, dH, dS, heads, p, 23, torch.empty(2, 64, 64, device=dev), torch.empty(2, 64, device=dev)))
    ops.transform_dw_batch(jobs, U, I)
    torch.cuda.synchronize()
    rel = lambda a_, b_: float((a_.double() - b_.double()).norm() / b_.double().norm())
    for (dW, da, ref_dW, ref_da), j in zip(want, jobs):
        assert rel(j[7], dW) < 2e-6 and rel(j[8], da) < 2e-6
        assert rel(j[7], ref_dW) < 2e-6 and rel(j[8], ref_da) < 2e-6
    bad = list(jobs[:1]) + [(jobs[1][0], jobs[1][1].bfloat16(), jobs[1][2].bfloat16()) + tuple(jobs[1][3:])]
    with pytest.raises(ValueError):
        ops.transform_dw_batch(bad, U, I)


@pytest.mark.parametrize("level,sparse_fused", [(1, False), (2, False), (2, True)])
def test_graphed_step_with_deferred_weight_gradients(dev, level, sparse_fused):
    """GraphedBPRStep (two chains, gradient sinks: the weight gradients of the fused stages are left to one batched launch per
    chain) against the same replayed steps with NGACF_FUSE_BWD = 0: same parameters after four Adam steps (no dropout, so that the
    two captures see the same masks), to the noise of fp32 sums and of the score gradient's float atomics."""
    import copy
    import ngacf_amd
    from ngacf_amd import ops
    from ngacf_amd.graphs import GraphedBPRStep
    from ngacf_amd.optim import Adam
    U, I, B = 900, 700, 64
    edge = torch.from_numpy(_graph(U, I, 30000, 2, hub=400)).to(dev)
    torch.manual_seed(1)
    m0 = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev)
    with torch.no_grad():
        m0.uEmbd.weight.mul_(30.0); m0.iEmbd.weight.mul_(30.0)
    ngacf_amd.set_compute_dtype(m0, torch.bfloat16)
    g = torch.Generator(device=dev); g.manual_seed(3)
    upn = torch.stack([torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I, (B,), device=dev, generator=g),
                       torch.randint(0, I, (B,), device=dev, generator=g)])
    from ngacf_amd import functional as NF
    res = {}
    was, was_sp = ops.FUSE_BWD, NF.FUSE_SPARSE
    NF.FUSE_SPARSE = sparse_fused
    try:
        for fuse in (level, 0):
            ops.FUSE_BWD = fuse
            m = copy.deepcopy(m0)
            opt = Adam(m.parameters(), lr=1e-2, weight_decay=1e-5)
            n0 = _count_dx_calls()
            step = GraphedBPRStep(m, ngacf_amd.BPRLoss(), opt, edge, B, preserve_state=True)
            losses = [float(step(upn)) for _ in range(4)]
            torch.cuda.synchronize()
            res[fuse] = (losses, {n: p.detach().clone() for n, p in m.named_parameters()}, _count_dx_calls() - n0)
    finally:
        ops.FUSE_BWD, NF.FUSE_SPARSE = was, was_sp
    assert res[level][2] > 0 and res[0][2] == 0
    assert np.allclose(res[level][0], res[0][0], rtol=1e-4, atol=1e-7), (res[level][0], res[0][0])
    for n in res[0][1]:
        a_, b_ = res[level][1][n].double(), res[0][1][n].double()
        assert float((a_ - b_).norm() / b_.norm().clamp_min(1e-30)) < 1e-4, n


@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_fp32_dx_with_the_previous_node_step_bitwise(dev, drop):
    """ngacf_transform_dx_node (fp32 storage: dX and the previous stage's step 1 in one launch) against ngacf_transform_bwd's dX half
    followed by ngacf_prop_bwd_node on the stored dX: dX, Gn, the (S, t) pairs and P1 bit for bit; row counts that are no multiple
    of the 64-row tile, an isolated row of the previous stage (R = 0)."""
    from ngacf_amd import ops
    U, I = 1033, 771
    N = U + I
    g = torch.Generator(device=dev); g.manual_seed(12)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    W, dH = rnd(2, 64, 64) * 0.2, rnd(N, 64) * 0.1
    X, H, dS = rnd(N, 64) * 0.5, rnd(N, 64) * 0.5, rnd(N, 8) * 0.1
    p_out, p_agg, p_R, p_S = rnd(N, 64) * 0.5, rnd(2, N, 64) * 0.5, rnd(2, N, 8).abs() + 0.1, rnd(N, 8) * 0.3
    p_R[0, 7, 2] = 0.0
    p_R[0, U + 5] = 0.0
    dX, _, _ = ops.transform_bwd(X, W, H, dH, dS, U, I, 8, drop, 31, part="dx")
    G = torch.empty(N, 64, device=dev); Gn = torch.empty(N, 64, device=dev)
    ST = torch.empty(N, 8, 2, device=dev); P1 = torch.empty(N, 8, device=dev)
    ops.prop_bwd_node(p_out, p_agg, p_R, dX, p_S, G, Gn, ST, P1, U, 8, (0, U, 0, I), False, elu=True, alpha=0.2)
    dX2, Gn2, ST2, P12 = ops.transform_dx_node(dH, W, U, I, drop, 31, None, (p_out, p_agg, p_R, p_S, True, 0.2))
    torch.cuda.synchronize()
    assert torch.equal(_bits(dX), _bits(dX2)) and torch.equal(_bits(Gn), _bits(Gn2))
    assert torch.equal(_bits(ST), _bits(ST2)) and torch.equal(_bits(P1), _bits(P12))
    assert float(Gn2[U + 5].abs().max()) == 0.0 and float(ST2[7, 2, 1]) == 0.0


def test_graphed_fp32_step_with_deferred_weight_gradients(dev):
    """fp32 storage: the graphed two-chain step with every stage's dW / davec left to the chain's batched launch (exact fp32 matrix
    cores) against the same steps with NGACF_DEFER_DW = 0 (a dW launch + a reduction per stage): same losses and the same parameters
    after four Adam steps, to the noise of fp32 sums (dropout off: both captures see the same masks)."""
    import copy
    import ngacf_amd
    from ngacf_amd import functional as NF
    from ngacf_amd.graphs import GraphedBPRStep
    from ngacf_amd.optim import Adam
    U, I, B = 900, 700, 64
    edge = torch.from_numpy(_graph(U, I, 30000, 2, hub=400)).to(dev)
    torch.manual_seed(2)
    m0 = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev)
    with torch.no_grad():
        m0.uEmbd.weight.mul_(30.0); m0.iEmbd.weight.mul_(30.0)
    g = torch.Generator(device=dev); g.manual_seed(4)
    upn = torch.stack([torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I, (B,), device=dev, generator=g),
                       torch.randint(0, I, (B,), device=dev, generator=g)])
    res, calls = {}, {}
    was, real = NF.DEFER_DW, NF.ops.transform_dw_batch
    try:
        for defer in (True, False):
            NF.DEFER_DW = defer
            n = [0]

            def spy(*a, _n=n, **k):
                _n[0] += 1
                return real(*a, **k)
            NF.ops.transform_dw_batch = spy
            m = copy.deepcopy(m0)
            opt = Adam(m.parameters(), lr=1e-2, weight_decay=1e-5)
            step = GraphedBPRStep(m, ngacf_amd.BPRLoss(), opt, edge, B, preserve_state=True)
            losses = [float(step(upn)) for _ in range(4)]
            torch.cuda.synchronize()
            res[defer], calls[defer] = (losses, {k: p.detach().clone() for k, p in m.named_parameters()}), n[0]
    finally:
        NF.DEFER_DW, NF.ops.transform_dw_batch = was, real
    assert calls[True] > 0 and calls[False] == 0
    assert np.allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-7), (res[True][0], res[False][0])
    for k in res[False][1]:
        a_, b_ = res[True][1][k].double(), res[False][1][k].double()
        assert float((a_ - b_).norm() / b_.norm().clamp_min(1e-30)) < 1e-5, k


@pytest.mark.parametrize("heads", [8, 1])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_split_sign_backward_of_one_stage_vs_float64_autograd(dev, heads, dt):
    """The regrouped backward (include/ngacf.h, ngacf_prop_bwd): the forward's second planes (the sums over the edges with a
    non-positive score), step 1's (S, t) pairs and P1, and the stage's dH / dS, each against a float64 statement of
    SPUIGACF.py:359-397 and its autograd (fixed dropout mask = the kernels' own bits), on a graph with short rows, long rows, a hub
    row cut into slices and isolated items."""
    import torch.nn.functional as F
    from ngacf_amd import ops
    from ngacf_amd.plan import build_plan
    U, I = 700, 503
    N, d, hw, alpha, p = U + I, 64, 64 // heads, 0.2, 0.25
    edge_np = _graph(U, I, 21000, 7, hub=450)
    edge = torch.from_numpy(edge_np).to(dev)
    plan = build_plan(edge, U, I, seg_len=128)
    assert plan.nsplit > 0
    rs = np.random.RandomState(3 + heads)
    X = torch.from_numpy(rs.randn(N, 64).astype(np.float32) * 0.7).to(dev).to(dt)
    W = torch.from_numpy((rs.randn(2, 64, 64) * 0.25).astype(np.float32)).to(dev)
    a = torch.from_numpy((rs.randn(2, 64) * 0.4).astype(np.float32)).to(dev)
    H, S = ops.transform_fwd(X, W, a, U, I, heads, dt)
    gout = torch.from_numpy(rs.randn(N, 64).astype(np.float32) * 0.2).to(dev).to(dt)
    out, agg, R = ops.prop_fwd(plan, H, S, S, a, heads, alpha, drop_p=p, seed=11, save=True)
    dH, dS = ops.prop_bwd(plan, H, S, S, a, out, agg, R, gout, heads, alpha, drop_p=p, seed=11)
    G = torch.empty(N, d, device=dev); Gn = torch.empty(N, d, dtype=dt, device=dev)
    ST = torch.empty(N, heads, 2, device=dev); P1 = torch.empty(N, heads, device=dev)
    ops.prop_bwd_node(out, agg, R, gout, S, G, Gn, ST, P1, U, heads, (0, U, 0, I), False, alpha=alpha)
    # ---- float64 statement: both directions of every (user, item) pair, the kernels' own mask bits
    eu, ei = edge[0], edge[1] + U
    dst, src = torch.cat([eu, ei]), torch.cat([ei, eu])
    from ngacf_amd import rng
    a32, b32 = edge[0].int().contiguous(), (edge[1] + U).int().contiguous()
    thr = int(rng.threshold(p))
    keep = torch.stack([(ops.dropout_bits(1, 11, a32, b32, torch.full_like(a32, h)).long() & 0xFFFFFFFF) >= thr for h in range(heads)], 1).double() / (1 - p)
    m = torch.cat([keep, keep])                                       # one mask per pair, both directions (SPUIGACF.py:375)
    Hd = H.double().requires_grad_(True)
    Sd = S.double().requires_grad_(True)
    s = Sd[dst] + Sd[src]
    e = torch.exp(-F.leaky_relu(s, alpha))
    Rr = torch.zeros(N, heads, dtype=torch.double, device=dev).index_add(0, dst, e)
    A = torch.zeros(N, d, dtype=torch.double, device=dev).index_add(0, dst, (m * e).repeat_interleave(hw, 1) * Hd[src])
    Rw = Rr.repeat_interleave(hw, 1)
    aggr = torch.where(Rw > 0, A / Rw.clamp_min(1e-300), torch.zeros_like(A))
    o = F.elu(Hd + aggr)
    gH, gS = torch.autograd.grad(o, [Hd, Sd], gout.double())
    with torch.no_grad():
        neg = (s <= 0).double()
        Rn = torch.zeros_like(Rr).index_add(0, dst, neg * e)
        An = torch.zeros_like(A).index_add(0, dst, (neg * m * e).repeat_interleave(hw, 1) * Hd[src])
        aggn = torch.where(Rw > 0, An / Rw.clamp_min(1e-300), torch.zeros_like(An))
        g = gout.double() * torch.where(o > 0, torch.ones_like(o), o + 1.0)
        gn = torch.where(Rw > 0, g / Rw.clamp_min(1e-300), torch.zeros_like(g))
        t = (gn * aggr).view(N, heads, hw).sum(2)
        p1 = (1 - alpha) * (Rr * (gn * aggn).view(N, heads, hw).sum(2) - Rn * t)
        a_own = torch.cat([a[0].expand(U, d), a[1].expand(I, d)]).double()
        dH_ref = gH + gS.repeat_interleave(hw, 1) * a_own                # S = H . a: the transform's share of dH, as the kernel folds it in
    tol = 2e-5 if dt == torch.float32 else 4e-2
    rel = lambda x, y: float((x.double() - y).abs().max() / y.abs().max().clamp_min(1e-30))
    # a score within rounding of zero may fall on either side of the split: exclude rows that have one (none with these seeds in fp32)
    near = torch.zeros(N, dtype=torch.bool, device=dev).index_put_((dst,), (s.detach().abs() < 1e-5).any(1), accumulate=True)
    ok = ~near
    assert int(ok.sum()) > 0.98 * N
    assert rel(R[0], Rr.detach()) < 1e-5 and rel(R[1][ok], Rn[ok]) < (1e-5 if dt == torch.float32 else 2e-2)
    assert rel(agg[0], aggr.detach()) < tol and rel(agg[1][ok], aggn[ok]) < tol
    assert rel(Gn, gn) < tol and rel(ST[:, :, 1], t) < tol and torch.equal(ST[:, :, 0], S)
    assert rel(P1[ok], p1[ok]) < (2e-4 if dt == torch.float32 else 6e-2)
    assert rel(dS, gS) < (2e-4 if dt == torch.float32 else 6e-2) and rel(dH, dH_ref) < tol


@pytest.mark.parametrize("drop", [0.0, 0.2])
def test_flagged_backward_equals_the_unflagged_one_on_a_gradient_that_is_zero_outside(dev, drop):
    """ngacf_prop_bwd with a scored batch's row flags (only edges whose source is flagged are walked, step 1 over the batch's list)
    against the plain call on the same gradient, which is zero outside the batch: dH and dS bit for bit.  The forward's saved rows
    exist for the batch's rows only in the flagged case: the plain call gets a full forward."""
    from ngacf_amd import ops
    from ngacf_amd.plan import build_plan
    U, I = 900, 640
    N = U + I
    edge = torch.from_numpy(_graph(U, I, 30000, 2, hub=600)).to(dev)
    plan = build_plan(edge, U, I, seg_len=128)
    rs = np.random.RandomState(5)
    bf = torch.bfloat16
    X = torch.from_numpy(rs.randn(N, 64).astype(np.float32) * 0.5).to(dev).to(bf)
    W = torch.from_numpy((rs.randn(2, 64, 64) * 0.2).astype(np.float32)).to(dev)
    a = torch.from_numpy((rs.randn(2, 64) * 0.3).astype(np.float32)).to(dev)
    H, S = ops.transform_fwd(X, W, a, U, I, 1, bf)
    g = torch.Generator(device=dev); g.manual_seed(8)
    B = 150
    bu, bi = torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I - 3, (B,), device=dev, generator=g)
    rows = ops.mark_rows(bu, bi, U, I, grad_d=64)
    rows.g32.zero_()
    rows.g32[torch.cat([bu, U + bi])] = torch.from_numpy(rs.randn(2 * B, 64).astype(np.float32) * 0.1).to(dev)
    gfull = torch.zeros(N, 64, device=dev)
    sel = rows.mask()
    gfull[sel] = rows.g32[sel]
    kw = dict(drop_p=drop, seed=4)
    out_s, agg_s, R_s = ops.prop_fwd(plan, H, S, S, a, 1, save=True, needed=rows, **kw)
    out_f, agg_f, R_f = ops.prop_fwd(plan, H, S, S, a, 1, save=True, **kw)
    dH_s, dS_s = ops.prop_bwd(plan, H, S, S, a, out_s, agg_s, R_s, rows.g32, 1, needed=rows, gout_f32=True, **kw)
    dH_f, dS_f = ops.prop_bwd(plan, H, S, S, a, out_f, agg_f, R_f, gfull, 1, gout_f32=True, **kw)
    torch.cuda.synchronize()
    # (the flagged forward adds a batch row's short-row terms in another order than the plain one: its saved rows agree to rounding,
    #  so the comparison of the backward runs on the plain forward's rows for both)
    dH_s2, dS_s2 = ops.prop_bwd(plan, H, S, S, a, out_f, agg_f, R_f, rows.g32, 1, needed=rows, gout_f32=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(_bits(dH_s2), _bits(dH_f)) and torch.equal(_bits(dS_s2), _bits(dS_f))
    assert float((dH_s.float() - dH_f.float()).abs().max()) <= 2e-2 * float(dH_f.float().abs().max())


def test_deferred_weight_gradients_are_written_even_when_the_lowest_stage_has_no_backward(dev):
    """inside a deferred gradient sink a stage hands out dW / davec tensors that the chain's ONE batched launch fills
    behind the backward of the stack's lowest stage.  When that backward never runs (torch.autograd.grad on the LATER stages'
    parameters only) the end-of-pass callback must launch the batch: the gradients equal the ones of a plain backward."""
    import ngacf_amd
    from ngacf_amd import functional as NF
    U, I = 700, 500
    edge = torch.from_numpy(_graph(U, I, 20000, 3, hub=300)).to(dev)
    torch.manual_seed(3)
    m = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev).train()
    with torch.no_grad():
        m.uEmbd.weight.mul_(30.0); m.iEmbd.weight.mul_(30.0)
    ngacf_amd.set_compute_dtype(m, torch.bfloat16)
    later = [p for n, p in m.named_parameters() if "stage1" in n or "stage2" in n]
    assert later
    g = torch.Generator(device=dev); g.manual_seed(2)
    u, it = torch.randint(0, U, (128,), device=dev, generator=g), torch.randint(0, I, (128,), device=dev, generator=g)
    type(m).sparse_last_stage, was = False, type(m).sparse_last_stage      # (every stage dense: the jobs of stages 1 and 2 are deferred)
    try:
        want = torch.autograd.grad(m(u, it, edge).sum(), later)
        sink = {}
        with NF.grad_sink(sink, deferred_ok=True):
            y = m(u, it, edge).sum()
        torch.autograd.grad(y, [p for p in later], allow_unused=True)       # the sunk parameters deliver to `sink`; stage 0's backward does not run
        torch.cuda.synchronize()
    finally:
        type(m).sparse_last_stage = was
    assert len(sink) >= len(later)
    for p_, w in zip(later, want):
        # (the batched launch and the per-stage one reduce their partial sums in different orders: equal to fp32 rounding)
        err = float((sink[p_].double() - w.double()).norm() / w.double().norm().clamp_min(1e-30))
        assert bool(torch.isfinite(sink[p_]).all()) and err < 1e-5, (tuple(p_.shape), err)
