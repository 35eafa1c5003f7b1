"""GPU parity tests: the HIP path (through the C ABI in ngacf_amd/_lib.py) against the
CPU oracle on the same inputs, against the reference-generated golden fixtures, and -- at
Gowalla scale -- through size-independent properties.  fp32 bar: max-abs 1e-4 (north_star);
tighter where the arithmetic allows.  Run with:  pytest -m gpu"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, params_of
from oracle import ngacf_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star: "matching the reference PyTorch CPU path within 1e-4 fp32"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _t(x, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return t if dtype is None else t.to(dtype)


def _maxabs(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.max(np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64))))


def _fused(params, names):
    """W [2,din,dout], avec [2,dout] of a stage from reference-keyed params (numpy)."""
    k = params[names[0] + ".W_u"].shape[1]
    Wu = np.concatenate([params[n + ".W_u"] for n in names], 1)
    Wi = np.concatenate([params[n + ".W_i"] for n in names], 1)
    au = np.concatenate([params[n + ".a"][0, :k] for n in names])
    ai = np.concatenate([params[n + ".a"][0, k:] for n in names])
    return np.stack([Wu, Wi]).astype(np.float32), np.stack([au, ai]).astype(np.float32)


def _random_graph(U, I, E, seed, hub=0):
    rs = np.random.RandomState(seed)
    u = np.concatenate([np.arange(U), rs.randint(0, U, E - U)])
    i = np.concatenate([rs.randint(0, I - 3, U), rs.randint(0, I - 3, E - U)])     # last 3 items isolated
    if hub:
        u = np.concatenate([u, rs.permutation(U)[:hub]])
        i = np.concatenate([i, np.zeros(hub, dtype=np.int64)])
    key = np.unique(u.astype(np.int64) * I + i)
    return np.stack([key // I, key % I])


def _random_params(seed, multi=False, d=64, scale=0.3):
    rs = np.random.RandomState(seed)
    p = {}
    for names in O.stage_names(multi):
        k = 8 if len(names) > 1 else d
        for n in names:
            p[n + ".W_u"] = (rs.randn(64, k) * scale).astype(np.float32)
            p[n + ".W_i"] = (rs.randn(64, k) * scale).astype(np.float32)
            p[n + ".a"] = (rs.randn(1, 2 * k) * scale).astype(np.float32)
    return p


# ---------------------------------------------------------------------------
# library / ABI
# ---------------------------------------------------------------------------
def test_library_loaded_and_rejects_bad_shapes(dev):
    import ngacf_amd
    from ngacf_amd import _lib
    assert "gfx950" in ngacf_amd.version()
    # unsupported row width is refused before any launch
    rc = _lib.lib.ngacf_prop_fwd(1, 1, 0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 48, 1, 0.2, 0.0, 0, 0, 3, 0)
    assert rc == -2
    # a table of 2^32 bytes or more does not fit the 32-bit row offsets: refused, not wrapped
    rc = _lib.lib.ngacf_prop_fwd(1, 1, 0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1 << 24, 64, 8, 0.2, 0.0, 0, 0, 3, 0)
    assert rc == -2
    assert _lib.lib.ngacf_score_fwd(0, 0, 0, 0, 0, 0, 5, 1, 1, 64, 0) == -1


# ---------------------------------------------------------------------------
# G1: single layer vs the reference
# ---------------------------------------------------------------------------
def test_g1_layer_k64_noconcat_and_rowsums(dev):
    from ngacf_amd import ops
    from ngacf_amd.plan import build_plan
    g = load_golden("g1_layer_k64_noconcat")
    U, I = int(g["U"]), int(g["I"])
    k = 64
    W = _t(np.stack([g["W_u"], g["W_i"]]), dev)
    avec = _t(np.stack([g["a"][0, :k], g["a"][0, k:]]), dev)
    plan = build_plan(_t(g["edge"], dev), U, I)
    H, S = ops.transform_fwd(_t(g["X"], dev), W, avec, U, I, 1)
    for from_row in (False, True):
        out, agg, R = ops.prop_fwd(plan, H, S, S, avec, 1, 0.2, elu=False, score_from_row=from_row, save=True)
        assert _maxabs(out, g["out"]) < 1e-5
        R = R[0]                                                           # (plane 0: the denominators; plane 1: the s <= 0 part)
        assert _maxabs(R[:U, 0], g["e_rowsum"]) < 1e-4 and _maxabs(R[U:, 0], g["e_colsum"]) < 1e-4
        assert float(R[U + 37:].abs().max()) == 0.0                       # isolated items
        assert torch.equal(out[U + 37:], H[U + 37:])                       # NaN->0 path: out = h


def test_g1_layer_k8_concat_standalone_module(dev):
    import ngacf_amd
    g = load_golden("g1_layer_k8_concat")
    U, I = int(g["U"]), int(g["I"])
    layer = ngacf_amd.SpUIGraphAttentionLayer(64, 8, dropout=0.0, alpha=0.2, concat=True).to(dev).eval()
    with torch.no_grad():
        layer.W_u.copy_(_t(g["W_u"], dev)); layer.W_i.copy_(_t(g["W_i"], dev)); layer.a.copy_(_t(g["a"], dev))
        out = layer(torch.arange(U, device=dev), torch.arange(I, device=dev) + U, _t(g["X"], dev), _t(g["edge"], dev))
    assert out.shape == (U + I, 8)
    assert _maxabs(out, g["out"]) < 1e-5


# ---------------------------------------------------------------------------
# G2-G5: full stacks, scores, loss, every parameter gradient
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("tag,multi", [("spuigacf", False), ("spuimultigacf", True)])
def test_g234_model_forward_loss_grads(dev, tag, multi):
    import ngacf_amd
    g = load_golden("g234_" + tag)
    U, I = int(g["U"]), int(g["I"])
    cls = ngacf_amd.SPUIMultiGACF if multi else ngacf_amd.SPUIGACF
    m = cls(U, I, 64, [64, 64], 0.0).to(dev)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in params_of(g).items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys      # checkpoint interchange
    m.train()
    edge = _t(g["edge"], dev)
    F = m.propagate(edge)
    assert _maxabs(F, g["F_f32"]) < 1e-5
    u, pos, neg = _t(g["u"], dev), _t(g["pos"], dev), _t(g["neg"], dev)
    ps, ns = m(u, pos, edge), m(u, neg, edge)
    smax = float(np.abs(g["pos_f64"]).max())                              # scores are O(10) here: relative bar
    assert _maxabs(ps, g["pos_f64"]) < 1e-6 * smax and _maxabs(ns, g["neg_f64"]) < 1e-6 * smax
    loss = ngacf_amd.BPRLoss()(ps, ns)
    assert abs(float(loss.detach()) - float(g["loss_f64"])) < 1e-5
    loss.backward()
    worst = 0.0
    from ngacf_amd.SPUIGACF import reference_named_grads
    grads = reference_named_grads(m)                                       # per-head views of the fused stage gradients
    assert set(grads) == {k[len("g_f64."):] for k in g if k.startswith("g_f64.")}
    for k, gr in grads.items():
        ref = g["g_f64." + k]
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = _maxabs(gr, ref) / scale
        ref_err = _maxabs(g["g_f32." + k], ref) / scale                   # the reference's own fp32 noise
        worst = max(worst, err)
        assert err < max(2e-4, 20 * ref_err), (k, err, ref_err)
    assert worst < 5e-3


def test_split_rows_match_unsplit(dev):
    """Rows cut into slices (+ partial-sum pass) give the same result as whole rows."""
    from ngacf_amd import ops
    from ngacf_amd.plan import build_plan
    U, I = 300, 200
    edge = _random_graph(U, I, 9000, 5, hub=250)
    p = _random_params(1)
    names = O.stage_names()[0]
    W, avec = (_t(x, dev) for x in _fused(p, names))
    X = _t(np.random.RandomState(2).randn(U + I, 64).astype(np.float32), dev)
    H, S = ops.transform_fwd(X, W, avec, U, I, 8)
    g = torch.randn(U + I, 64, device=dev)                 # fp32 H here, so an fp32 gradient
    ref = None
    # (seg_len, short_max, hub_cap, long_max): all rows as whole-wave segments / long rows as workgroup-per-row hubs (the
    # default) / cut into slices with partial sums (hub_cap 0) / hubs AND slices / every mix of the slot-per-row mode with
    # the long mode / all rows in the slot-per-row mode
    for seg, short, cap, lmax in ((1 << 20, -1, 0, None), (16, -1, None, None), (8, 4, None, 8), (64, -1, 0, None), (8, 4, 0, None),
                                  (8, 4, 60, 8), (256, 64, 0, None), (16, 30, None, None), (1 << 20, 1 << 20, 0, None)):
        plan = build_plan(_t(edge, dev), U, I, seg_len=seg, short_max=short, hub_cap=cap, long_max=lmax)
        if (seg, short, cap) == (8, 4, 0):
            assert plan.nsplit > 100 and plan.nshort > 0 and plan.nhub == 0
        if (seg, short, cap) == (8, 4, None):
            assert plan.nhub > 100 and plan.nsplit == 0 and 0 < plan.nhub_user < plan.nhub
        if cap == 60:
            assert plan.nhub > 50 and plan.nsplit > 0
        if short == 1 << 20:
            assert plan.nseg == 0 and plan.nhub == 0 and int((plan.srow >= 0).sum()) == U + I
        out, agg, R = ops.prop_fwd(plan, H, S, S, avec, 8, save=True)
        dH, dS = ops.prop_bwd(plan, H, S, S, avec, out, agg, R, g, 8)
        if ref is None:
            assert plan.nsplit == 0 and plan.nshort == 0 and plan.nhub == 0
            ref = (out, R, dH, dS)
            continue
        assert float((out - ref[0]).abs().max()) < 1e-5
        assert float(((R - ref[1]).abs() / ref[1].clamp_min(1e-9)).max()) < 1e-5
        assert float((dH - ref[2]).abs().max() / ref[2].abs().max()) < 1e-5
        assert float((dS - ref[3]).abs().max() / ref[3].abs().max()) < 1e-4


@pytest.mark.parametrize("multi", [False, True])
def test_random_graph_vs_oracle_forward_backward(dev, multi):
    import ngacf_amd
    U, I = 700, 500
    edge = _random_graph(U, I, 16000, 11 + multi, hub=600)
    rs = np.random.RandomState(3)
    cls = ngacf_amd.SPUIMultiGACF if multi else ngacf_amd.SPUIGACF
    torch.manual_seed(4)
    m = cls(U, I, 64, [64, 64], 0.0).to(dev).train()
    with torch.no_grad():
        m.uEmbd.weight.mul_(30.0); m.iEmbd.weight.mul_(30.0)
    for mod in m.modules():
        if hasattr(mod, "seg_len"):
            mod.seg_len = 64                                              # exercise the long-row path
    p64 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    B = 256
    u, pos, neg = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
    loss_o, pos_o, neg_o, grads_o = O.bpr_step_grads(p64, u, pos, neg, edge, multi=multi)
    e_t = _t(edge, dev)
    ps, ns = m(_t(u, dev), _t(pos, dev), e_t), m(_t(u, dev), _t(neg, dev), e_t)
    loss = ngacf_amd.BPRLoss()(ps, ns)
    loss.backward()
    assert _maxabs(ps, pos_o) < TOL and _maxabs(ns, neg_o) < TOL
    assert abs(float(loss.detach()) - loss_o) < 1e-5
    from ngacf_amd.SPUIGACF import reference_named_grads
    for k, gr in reference_named_grads(m).items():
        scale = max(float(np.abs(grads_o[k]).max()), 1e-12)
        assert _maxabs(gr, grads_o[k]) / scale < 1e-3, k


def test_g10_ml100k_eval_output(dev):
    import ngacf_amd
    g = load_golden("g9_g10_ml100k")
    d = load_golden("ml100k_interactions")
    U, I = int(g["shape"][0]), int(g["shape"][1])
    edge = np.stack([d["userId"].astype(np.int64), d["itemId"].astype(np.int64)])      # file order, unsorted
    m = ngacf_amd.SPUIGACF(U, I, 64, [64, 64], 0.1).to(dev).eval()
    sd = {k: torch.from_numpy(v) for k, v in params_of(g).items()}
    X = (np.random.RandomState(int(g["emb_seed"])).randn(U + I, 64) * 0.01).astype(np.float32)
    sd["uEmbd.weight"], sd["iEmbd.weight"] = torch.from_numpy(X[:U]), torch.from_numpy(X[U:])
    m.load_state_dict(sd)
    with torch.no_grad():
        F = m.propagate(_t(edge, dev))
    assert _maxabs(F[_t(g["rows"], dev)], g["F_rows"]) < 1e-6
    assert abs(float(F.double().sum()) - float(g["F_sum"])) < 1e-2
    assert abs(float(F.double().abs().sum()) - float(g["F_abs_sum"])) < 1e-2


# ---------------------------------------------------------------------------
# dropout: device RNG pinned to the host statement, then semantics vs oracle with those masks
# ---------------------------------------------------------------------------
def test_dropout_bits_match_host_hash(dev):
    from ngacf_amd import ops, rng
    rs = np.random.RandomState(0)
    n = 100000
    a, b = rs.randint(0, 2 ** 31 - 1, n).astype(np.int32), rs.randint(0, 2 ** 31 - 1, n).astype(np.int32)
    c = rs.randint(0, 16, n).astype(np.int32)
    for seed in (0, 2019, 0x1234567890ABCDEF):
        fb = ops.dropout_bits(0, seed, _t(a, dev), _t(b, dev)).cpu().numpy().view(np.uint32)
        assert np.array_equal(fb, rng.feat_bits(seed, a, b))
        eb = ops.dropout_bits(1, seed, _t(a, dev), _t(b, dev), _t(c, dev)).cpu().numpy().view(np.uint32)
        want = np.empty(n, dtype=np.uint32)
        for h in range(16):
            want[c == h] = rng.edge_bits(seed, h, a[c == h], b[c == h])
        assert np.array_equal(eb, want)
    keep = (rng.feat_bits(7, a, b) >= rng.threshold(0.1)).mean()
    assert abs(keep - 0.9) < 5e-3


def test_train_mode_dropout_semantics_vs_oracle(dev):
    """Same masks on both sides: edge dropout hits numerators only (SPUIGACF.py:375), feature
    dropout sits in front of every stage (:208,:213); forward and all gradients."""
    from ngacf_amd import functional as NF, rng
    from ngacf_amd.plan import build_plan
    U, I, p, seed = 200, 150, 0.25, 99
    edge = _random_graph(U, I, 4000, 8, hub=150)
    N = U + I
    prm = _random_params(6)
    X = np.random.RandomState(7).randn(N, 64).astype(np.float32)
    plan = build_plan(_t(edge, dev), U, I, seg_len=32)
    stages = O.stage_names()
    seeds = [NF.stage_seed(seed, s) for s in range(2)]
    nn_, cc_ = np.meshgrid(np.arange(N), np.arange(64), indexing="ij")
    fm = [rng.keep_scale(rng.feat_bits(sd, nn_, cc_), p).astype(np.float64) for sd in seeds]
    em = [[rng.keep_scale(rng.edge_bits(seeds[si], h, edge[0], edge[1] + U), p).astype(np.float64)
           for h in range(len(names))] for si, names in enumerate(stages)]
    p64 = {k: v.astype(np.float64) for k, v in prm.items()}
    F_o, tape = O.spuigat_forward(p64, X.astype(np.float64), edge, U, I, feat_masks=fm, edge_masks=em, keep=True)
    gF = np.random.RandomState(9).randn(N, 64)
    dX_o, grads_o = O.spuigat_backward(gF, tape, p64, edge, U, I, feat_masks=fm, edge_masks=em)

    x = _t(X, dev).requires_grad_(True)
    leaves, cur = [], x
    for si, names in enumerate(stages):
        W, avec = (_t(v, dev).requires_grad_(True) for v in _fused(prm, names))
        leaves.append((W, avec))
        cur = NF.stage(cur, W, avec, plan, len(names), 0.2, True, p, p, seeds[si])
    assert _maxabs(cur, F_o) < TOL
    cur.backward(_t(gF.astype(np.float32), dev))
    assert _maxabs(x.grad, dX_o) / np.abs(dX_o).max() < 1e-4
    for si, names in enumerate(stages):
        W, avec = leaves[si]
        k = prm[names[0] + ".W_u"].shape[1]
        for h, n in enumerate(names):
            sl = slice(h * k, (h + 1) * k)
            for got, want in ((W.grad[0][:, sl], grads_o[n + ".W_u"]), (W.grad[1][:, sl], grads_o[n + ".W_i"]),
                              (avec.grad[0][sl], grads_o[n + ".a"][0, :k]), (avec.grad[1][sl], grads_o[n + ".a"][0, k:])):
                assert _maxabs(got, want) / max(np.abs(want).max(), 1e-9) < 1e-4, (n,)


def test_module_dropout_is_reproducible_and_unbiased(dev):
    import ngacf_amd
    U, I = 300, 200
    edge = _t(_random_graph(U, I, 6000, 2), dev)
    torch.manual_seed(1)
    m = ngacf_amd.SPUIGACF(U, I, 64, [64, 64], 0.2).to(dev).train()
    torch.manual_seed(5); a = m.propagate(edge)
    torch.manual_seed(5); b = m.propagate(edge)
    c = m.propagate(edge)
    assert torch.equal(a, b) and not torch.equal(a, c)
    m.eval()
    e1, e2 = m.propagate(edge), m.propagate(edge)
    assert torch.equal(e1, e2)


# ---------------------------------------------------------------------------
# bf16 storage (fp32 accumulate)
# ---------------------------------------------------------------------------
def test_bf16_storage_close_to_fp32(dev):
    import ngacf_amd
    U, I = 700, 500
    edge = _t(_random_graph(U, I, 16000, 21, hub=400), dev)
    torch.manual_seed(2)
    m = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.0).to(dev).train()
    with torch.no_grad():
        m.uEmbd.weight.mul_(30.0); m.iEmbd.weight.mul_(30.0)
    u = torch.randint(0, U, (512,), device=dev); it = torch.randint(0, I, (512,), device=dev)
    s32 = m(u, it, edge); s32.sum().backward()
    g32 = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    ngacf_amd.set_compute_dtype(m, torch.bfloat16)
    F16 = m.propagate(edge)
    assert F16.dtype == torch.bfloat16
    s16 = m(u, it, edge); s16.sum().backward()
    # bf16 has 8 significant bits: 3 stages of storage rounding -> ~1e-2 relative
    assert float((s16 - s32).abs().max() / s32.abs().max()) < 3e-2
    for k, p in m.named_parameters():
        rel = float((p.grad - g32[k]).norm() / g32[k].norm().clamp_min(1e-12))    # relative L2
        assert rel < 5e-2, (k, rel)


@pytest.mark.parametrize("din,dout,heads", [(64, 64, 8), (64, 64, 1), (128, 64, 8), (64, 128, 16), (128, 128, 1)])
@pytest.mark.parametrize("xdt", ["bf16", "fp32"])
def test_bf16_transform_backward_on_matrix_cores(dev, din, dout, heads, xdt):
    """bf16 storage runs dX and dW on MFMA (transposed LDS reads for dW): compare with fp32 torch matmuls of the
    SAME bf16-rounded operands (fp32 accumulate on both sides, so only the summation order differs)."""
    from ngacf_amd import ops, rng
    g = torch.Generator(device=dev).manual_seed(din + dout + heads)
    for U, I, p in ((1, 1, 0.0), (37, 91, 0.0), (1000, 1531, 0.25)):      # ragged tiles, both types, with feature dropout
        N = U + I
        X = torch.randn(N, din, device=dev, generator=g)
        X = X.to(torch.bfloat16) if xdt == "bf16" else X
        W = torch.randn(2, din, dout, device=dev, generator=g) * 0.2
        avec = torch.randn(2, dout, device=dev, generator=g)
        H, _ = ops.transform_fwd(X, W, avec, U, I, heads, torch.bfloat16, feat_drop_p=p, seed=11)
        dH = torch.randn(N, dout, device=dev, generator=g).to(torch.bfloat16)        # in H's storage type, as ngacf_prop_bwd writes it
        dS = torch.randn(N, heads, device=dev, generator=g)
        dX, dW, da = ops.transform_bwd(X, W, H, dH, dS, U, I, heads, feat_drop_p=p, seed=11)
        Xm = X.float()
        if p > 0:
            n = torch.arange(N, device=dev, dtype=torch.int64)[:, None].expand(N, din)
            c = torch.arange(din, device=dev, dtype=torch.int64)[None].expand(N, din)
            bits = rng.feat_bits(11, n.cpu().numpy().astype(np.uint32), c.cpu().numpy().astype(np.uint32))
            keep = torch.from_numpy((bits >= rng.threshold(p)).astype(np.float32)).to(dev) / (1 - p)
        else:
            keep = torch.ones(N, din, device=dev)
        Xb = (Xm * keep).to(torch.bfloat16).float()
        gb = dH.to(torch.bfloat16).float()
        Wb = W.to(torch.bfloat16).float()
        # either half alone (NULL for the other) gives the same bits
        only_dx = ops.transform_bwd(X, W, H, dH, dS, U, I, heads, feat_drop_p=p, seed=11, part="dx")
        only_dw = ops.transform_bwd(X, W, H, dH, dS, U, I, heads, feat_drop_p=p, seed=11, part="dw")
        assert only_dx[1] is None and only_dw[0] is None
        assert torch.equal(only_dx[0], dX) and torch.equal(only_dw[1], dW) and torch.equal(only_dw[2], da)
        for t, (lo, hi) in enumerate(((0, U), (U, N))):
            ref_dw = Xb[lo:hi].T @ gb[lo:hi]
            assert float((dW[t] - ref_dw).abs().max()) <= 1e-4 * max(1.0, float(ref_dw.abs().max())), (t, U, I)
            ref_dx = (gb[lo:hi] @ Wb[t].T) * keep[lo:hi]
            tol = 2e-2 if xdt == "bf16" else 1e-4
            assert float((dX[lo:hi].float() - ref_dx).abs().max()) <= tol * max(1.0, float(ref_dx.abs().max())), (t, U, I)
            hw = dout // heads
            ref_da = (dS[lo:hi].repeat_interleave(hw, 1) * H[lo:hi].float()).sum(0)
            assert float((da[t] - ref_da).abs().max()) <= 1e-4 * max(1.0, float(ref_da.abs().max())), (t, U, I)



# ---------------------------------------------------------------------------
# scoring, BPR, static SpMM
# ---------------------------------------------------------------------------
def test_score_and_bpr_kernels(dev):
    from ngacf_amd import ops
    rs = np.random.RandomState(0)
    U, I, B = 100, 80, 1000
    F = rs.randn(U + I, 64).astype(np.float32)
    u, it = rs.randint(0, U, B), rs.randint(0, I, B)
    sc = ops.score_fwd(_t(F, dev), _t(u, dev), _t(it, dev), U)
    want = np.sum(F[u] * F[it + U], 1)
    assert _maxabs(sc, want) < 1e-4
    g = rs.randn(B).astype(np.float32)
    gF = ops.score_bwd(_t(F, dev), _t(u, dev), _t(it, dev), _t(g, dev), U)
    ref = np.zeros_like(F, dtype=np.float64)
    np.add.at(ref, u, g[:, None] * F[it + U]); np.add.at(ref, it + U, g[:, None] * F[u])
    assert _maxabs(gF, ref) < 1e-3
    pos, neg = rs.randn(B).astype(np.float32) * 3, rs.randn(B).astype(np.float32) * 3
    loss = ops.bpr_fwd(_t(pos, dev), _t(neg, dev))
    assert abs(float(loss) - O.bpr_loss(pos.astype(np.float64), neg.astype(np.float64))) < 1e-5
    gp, gn = ops.bpr_bwd(_t(pos, dev), _t(neg, dev), torch.tensor(1.0, device=dev))
    want_g = -(1.0 / (1.0 + np.exp(pos.astype(np.float64) - neg))) / B
    assert _maxabs(gp, want_g) < 1e-7 and _maxabs(gn, -want_g) < 1e-7
    l2, gp2, gn2 = ops.bpr_fwd_bwd(_t(pos, dev), _t(neg, dev))        # loss + unit-gradient in one launch
    assert abs(float(l2) - float(loss)) < 1e-7 and _maxabs(gp2, want_g) < 1e-7 and _maxabs(gn2, -want_g) < 1e-7
    # empty batch
    assert ops.score_fwd(_t(F, dev), _t(u[:0], dev), _t(it[:0], dev), U).numel() == 0


def test_out_of_range_indices_never_become_addresses(dev):
    """userIdx / itemIdx values are data on the device.  The reference's indexing raises (SPUIGACF.py:49-50);
    here a bad pair scores NaN, its gradient is dropped (no wild atomicAdd), the good pairs are untouched, and the host
    check raises IndexError."""
    from ngacf_amd import ops
    rs = np.random.RandomState(3)
    U, I, B = 100, 80, 512
    F = rs.randn(U + I, 64).astype(np.float32)
    u, it = rs.randint(0, U, B), rs.randint(0, I, B)
    bad = np.array([5, 17, 100, 333])
    u[5], it[17], u[100], it[333] = U, I, -1, 1 << 40          # item id U+I-1 < it < ... : would alias a user row / wrap
    ops.raise_if_bad_indices(dev)                               # clean slate
    Fd, g = _t(F, dev), _t(rs.randn(B).astype(np.float32), dev)
    sc = ops.score_fwd(Fd, _t(u, dev), _t(it, dev), U).cpu().numpy()
    good = np.setdiff1d(np.arange(B), bad)
    assert np.isnan(sc[bad]).all() and not np.isnan(sc[good]).any()
    assert np.abs(sc[good] - np.sum(F[u[good]] * F[it[good] + U], 1)).max() < 1e-4
    gF = ops.score_bwd(Fd, _t(u, dev), _t(it, dev), g, U).cpu().numpy()
    ref = np.zeros_like(F, dtype=np.float64)
    gg = g.cpu().numpy()
    np.add.at(ref, u[good], gg[good, None] * F[it[good] + U]); np.add.at(ref, it[good] + U, gg[good, None] * F[u[good]])
    assert np.abs(gF - ref).max() < 1e-3
    assert ops.bad_index_count(dev) == 8                        # 4 in the forward, 4 in the backward
    with pytest.raises(IndexError):
        ops.raise_if_bad_indices(dev)
    assert ops.bad_index_count(dev) == 0
    # the module surface: a drop-in forward with a bad index ends in the same IndexError at the host check
    import ngacf_amd
    torch.manual_seed(0)
    m = ngacf_amd.SPUIGACF(6, 5, 64, [64, 64], 0.0).to(dev).eval()
    edge = torch.tensor([[0, 1, 2, 3, 4, 5], [0, 1, 2, 3, 4, 0]], device=dev)
    s = m(torch.tensor([0, 6], device=dev), torch.tensor([1, 1], device=dev), edge)
    assert bool(torch.isnan(s[1])) and not bool(torch.isnan(s[0]))
    with pytest.raises(IndexError):
        ops.raise_if_bad_indices(dev)


def test_g7_static_spmm(dev):
    from ngacf_amd import ops
    from ngacf_amd.plan import build_plan
    g = load_golden("g7_static_spmm")
    U, I = int(g["U"]), int(g["I"])
    N = U + I
    # the Laplacian is N x N over the unified node ids: a plan with U = N "users" pointing at themselves
    from ngacf_amd.spmm import build_square_plan
    plan, val = build_square_plan(_t(g["L_row"], dev), _t(g["L_col"], dev), _t(g["L_val"], dev), N, seg_len=16)
    out = ops.spmm_fwd(plan, val, _t(g["X"], dev), self_loop=True)
    assert _maxabs(out, g["out"]) < 1e-5


# ---------------------------------------------------------------------------
# Gowalla scale: size-independent properties, checked against torch GPU ops
# ---------------------------------------------------------------------------
def test_gowalla_scale_properties(dev):
    from ngacf_amd import ops
    from ngacf_amd.data import synthetic_bipartite
    from ngacf_amd.plan import build_plan
    U, I, E = 29858, 40981, 1027370
    edge = synthetic_bipartite(U, I, E, seed=2019, device=dev)
    plan = build_plan(edge, U, I)
    assert plan.E == edge.shape[1] and abs(plan.E - E) < 0.01 * E
    N = U + I
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.randn(N, 64, device=dev, generator=g)
    W = torch.randn(2, 64, 64, device=dev, generator=g) * 0.2
    avec = torch.randn(2, 64, device=dev, generator=g) * 0.2
    eu, ei = edge[0], edge[1] + U
    src = torch.cat([ei, eu]); dst = torch.cat([eu, ei])
    for heads in (8, 1):
        H, S = ops.transform_fwd(X, W, avec, U, I, heads)
        Href = torch.cat([X[:U] @ W[0], X[U:] @ W[1]])
        assert float((H - Href).abs().max()) < 1e-4
        hw = 64 // heads
        Sref = (Href * torch.cat([avec[0].expand(U, 64), avec[1].expand(I, 64)])).view(N, heads, hw).sum(2)
        assert float((S - Sref).abs().max()) < 1e-4
        out, agg, R = ops.prop_fwd(plan, H, S, S, avec, heads, save=True)
        s = S[dst] + S[src]
        e = torch.exp(-torch.nn.functional.leaky_relu(s, 0.2))                      # [2E, heads]
        Rref = torch.zeros(N, heads, device=dev).index_add_(0, dst, e)
        assert float(((R[0] - Rref).abs() / Rref.clamp_min(1e-6)).max()) < 1e-4     # checksum of all 2E coefficients
        # the second planes: the same sums over the edges with a non-positive score (what the backward's row-local half reads)
        en = torch.where(s > 0, torch.zeros_like(e), e)
        Rn = torch.zeros(N, heads, device=dev).index_add_(0, dst, en)
        far = (s.abs() > 1e-4).float()                                              # (a score within rounding of 0 may fall on either side)
        Rn_sure = torch.zeros(N, heads, device=dev).index_add_(0, dst, en * far)
        slack = torch.zeros(N, heads, device=dev).index_add_(0, dst, e * (1 - far))
        assert bool(((R[1] - Rn_sure) >= -1e-4 * Rref.clamp_min(1e-6)).all()) and bool(((R[1] - Rn_sure) <= slack + 1e-4 * Rref.clamp_min(1e-6)).all())
        if float(slack.max()) == 0.0:
            numn = torch.zeros(N, 64, device=dev).index_add_(0, dst, en.repeat_interleave(hw, 1) * H[src])
            aggn = torch.where(Rref.repeat_interleave(hw, 1) > 0, numn / Rref.repeat_interleave(hw, 1), torch.zeros_like(numn))
            assert float((agg[1] - aggn).abs().max()) < TOL and float((R[1] - Rn).abs().max()) < 1e-4 * float(Rref.max())
        num = torch.zeros(N, 64, device=dev).index_add_(0, dst, e.repeat_interleave(hw, 1) * H[src])
        pre = H + torch.where(Rref.repeat_interleave(hw, 1) > 0, num / Rref.repeat_interleave(hw, 1), torch.zeros_like(num))
        ref = torch.nn.functional.elu(pre)
        assert float((out - ref).abs().max()) < TOL
        # attention-free limit: a = 0 -> e = 1 -> plain mean aggregation, R = degree
        z = torch.zeros_like(avec)
        H0, S0 = ops.transform_fwd(X, W, z, U, I, heads)
        out0, _, R0 = ops.prop_fwd(plan, H0, S0, S0, z, heads, elu=False, save=True)
        assert torch.equal(R0[0, :, 0], plan.degrees.float())
    # linearity of the aggregate in H at fixed scores: prop(2H) - 2 prop(H) == 0 without ELU
    o1 = ops.prop_fwd(plan, H, S, S, avec, 1, elu=False, score_from_row=False)
    o2 = ops.prop_fwd(plan, 2 * H, S, S, avec, 1, elu=False, score_from_row=False)
    assert float((o2 - 2 * o1).abs().max()) < 1e-4


def test_spuigagpcf_static_propagation_forward_backward(dev):
    """SPUIGAGPCF (SPUIGACF.py:103-172): GAT + GPLayer (L + I) @ X stack, against the same math in torch."""
    import ngacf_amd
    from ngacf_amd.spmm import mean_adj_coo
    U, I = 120, 90
    edge = _t(_random_graph(U, I, 1500, 4), dev)
    r, c, v = mean_adj_coo(edge, U, I)
    torch.manual_seed(0)
    m = ngacf_amd.SPUIGAGPCF(U, I, (r, c, v), 64, [64, 64], 0.0).to(dev).train()
    with torch.no_grad():
        m.uEmbd.weight.mul_(30); m.iEmbd.weight.mul_(30)
    u = torch.randint(0, U, (64,), device=dev); it = torch.randint(0, I, (64,), device=dev)
    sc = m(u, it, edge)
    sc.sum().backward()
    # torch restatement of the tail on the same propagated features
    F0 = m.propagate(edge).detach().requires_grad_(True)
    L = torch.sparse_coo_tensor(torch.stack([r, c]), v, (U + I, U + I)).coalesce()
    eye = torch.sparse_coo_tensor(torch.arange(U + I, device=dev).repeat(2, 1), torch.ones(U + I, device=dev), (U + I, U + I))
    x, outs = F0, [F0]
    for aff in m.Affinelayers:
        x = torch.relu(aff(torch.sparse.mm((L + eye).coalesce(), x)))
        outs.append(x)
    fin = torch.cat(outs, 1)
    ref = (fin[u] * fin[it + U]).sum(1)
    assert float((sc - ref).abs().max() / ref.abs().max()) < 1e-5
    g_ref = torch.autograd.grad(ref.sum(), m.Affinelayers[0].weight, retain_graph=True)[0]
    assert float((m.Affinelayers[0].weight.grad - g_ref).abs().max() / g_ref.abs().max()) < 1e-4
    assert m.uEmbd.weight.grad.abs().sum() > 0
    assert set(k for k in m.state_dict() if not k.startswith("gat.")) == \
        {"uEmbd.weight", "iEmbd.weight", "Affinelayers.0.weight", "Affinelayers.0.bias", "Affinelayers.1.weight", "Affinelayers.1.bias"}


@pytest.mark.parametrize("U,I,pairs", [(1, 1, [(0, 0)]), (3, 2, [(0, 0), (1, 0), (2, 1)]), (5, 9, [(u, (3 * u) % 9) for u in range(5)] + [(0, 8), (4, 8)])])
def test_tiny_graphs_vs_oracle(dev, U, I, pairs):
    """Graphs smaller than one wave / one tile; items without any user; every user of degree 1."""
    import ngacf_amd
    from ngacf_amd.SPUIGACF import reference_named_grads
    edge = np.array(sorted(set(pairs)), dtype=np.int64).T
    torch.manual_seed(U * 7 + I)
    m = ngacf_amd.SPUIGACF(U, I, 64, [64, 64], 0.0).to(dev).train()
    with torch.no_grad():
        m.uEmbd.weight.mul_(40); m.iEmbd.weight.mul_(40)
    p64 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    u = np.arange(U) % U; pos = np.arange(U) % I; neg = (np.arange(U) + 1) % I
    loss_o, pos_o, neg_o, grads_o = O.bpr_step_grads(p64, u, pos, neg, edge)
    e = _t(edge, dev)
    ps, ns = m(_t(u, dev), _t(pos, dev), e), m(_t(u, dev), _t(neg, dev), e)
    loss = ngacf_amd.BPRLoss()(ps, ns)
    loss.backward()
    smax = max(float(np.abs(pos_o).max()), float(np.abs(neg_o).max()), 1.0)      # scores are O(100) here
    assert _maxabs(ps, pos_o) < 1e-5 * smax and _maxabs(ns, neg_o) < 1e-5 * smax
    assert abs(float(loss.detach()) - loss_o) < 1e-5 * max(abs(loss_o), 1.0)
    for k, gr in reference_named_grads(m).items():
        assert _maxabs(gr, grads_o[k]) <= 1e-3 * max(float(np.abs(grads_o[k]).max()), 1e-9) + 1e-7, k


def test_d128_four_stage_stack_vs_oracle(dev):
    """BASELINE config 4 shape: embedSize 128, three 8-head stages + the 128-wide out layer (rows of 256/512 B)."""
    import ngacf_amd
    from ngacf_amd.SPUIGACF import reference_named_grads
    U, I = 260, 190
    edge = _random_graph(U, I, 5000, 31, hub=200)
    torch.manual_seed(3)
    m = ngacf_amd.SPUIStackedGACF(U, I, 128, [64, 64], 0.0, stages=4).to(dev).train()
    with torch.no_grad():
        m.uEmbd.weight.mul_(30); m.iEmbd.weight.mul_(30)
    assert m.gat.stage_shapes() == [(128, 64, 8), (64, 64, 8), (64, 64, 8), (64, 128, 1)]
    p64 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    rs = np.random.RandomState(0)
    B = 128
    u, pos, neg = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
    loss_o, pos_o, neg_o, grads_o = O.bpr_step_grads(p64, u, pos, neg, edge, hidden=3)
    e = _t(edge, dev)
    ps, ns = m(_t(u, dev), _t(pos, dev), e), m(_t(u, dev), _t(neg, dev), e)
    loss = ngacf_amd.BPRLoss()(ps, ns)
    loss.backward()
    smax = float(np.abs(pos_o).max())
    assert _maxabs(ps, pos_o) < 1e-5 * max(smax, 1.0) and abs(float(loss.detach()) - loss_o) < 1e-5
    for k, gr in reference_named_grads(m).items():
        assert _maxabs(gr, grads_o[k]) / max(float(np.abs(grads_o[k]).max()), 1e-12) < 1e-3, k
    # bf16 storage on the same shape runs and stays close
    ngacf_amd.set_compute_dtype(m, torch.bfloat16)
    ps16 = m(_t(u, dev), _t(pos, dev), e)
    assert float((ps16 - ps).abs().max() / ps.abs().max()) < 5e-2


def test_inkernel_hub_finish_is_bitwise_equal_to_two_kernel_path(dev):
    """Hub rows are finished by their last-arriving slice inside the launch (agent-scope release/acquire ticket).
    Partials are added in slot order on both paths, so the outputs must be BITWISE equal -- any stale read of
    another workgroup's partial sums would show.  Repeated under load, caches warmed by the previous launch."""
    from ngacf_amd import ops
    from ngacf_amd.data import synthetic_bipartite
    from ngacf_amd.plan import build_plan
    U, I, E = 29858, 40981, 1027370
    edge = synthetic_bipartite(U, I, E, seed=7, device=dev)
    N = U + I
    g = torch.Generator(device=dev); g.manual_seed(3)
    X = torch.randn(N, 64, device=dev, generator=g)
    W = torch.randn(2, 64, 64, device=dev, generator=g) * 0.2
    avec = torch.randn(2, 64, device=dev, generator=g) * 0.2
    for seg, dt in ((256, torch.float32), (32, torch.bfloat16), (64, torch.float32)):
        plan = build_plan(edge, U, I, seg_len=seg)
        assert plan.nsplit > 400
        H, S = ops.transform_fwd(X.to(dt), W, avec, U, I, 8, dt)
        gout = torch.randn(N, 64, device=dev, generator=g).to(dt)
        was = ops.INKERNEL_HUB_FINISH
        ops.INKERNEL_HUB_FINISH = False
        try:
            ref = ops.prop_fwd(plan, H, S, S, avec, 8, save=True)      # a second kernel finishes the hub rows
            ref_b = ops.prop_bwd(plan, H, S, S, avec, *ref, gout, 8, drop_p=0.1, seed=9)
        finally:
            ops.INKERNEL_HUB_FINISH = was
        assert ops.INKERNEL_HUB_FINISH                                 # the default
        side = torch.cuda.Stream()
        for it in range(25):
            out = ops.prop_fwd(plan, H, S, S, avec, 8, save=True)
            for a_, b_ in zip(out, ref):
                assert torch.equal(a_, b_), (seg, dt, it)
            if it % 3 == 0:                                            # the backward's hub rows, the same protocol
                for a_, b_ in zip(ops.prop_bwd(plan, H, S, S, avec, *ref, gout, 8, drop_p=0.1, seed=9), ref_b):
                    assert torch.equal(a_, b_), (seg, dt, it, "backward")
            if it % 5 == 0:                                            # unrelated traffic between launches
                torch.randn(4_000_000, device=dev).sum()
            if it % 7 == 0:                                            # an overlapping launch of the same plan on another stream
                side.wait_stream(torch.cuda.current_stream())         # (its own arrival counters)
                with torch.cuda.stream(side):
                    o2 = ops.prop_fwd(plan, H, S, S, avec, 8, save=True)
                o1 = ops.prop_fwd(plan, H, S, S, avec, 8, save=True)
                torch.cuda.current_stream().wait_stream(side)
                for a_, b_, c_ in zip(o1, o2, ref):
                    assert torch.equal(a_, c_) and torch.equal(b_, c_), (seg, dt, it)
        torch.cuda.synchronize()
        for t in plan._tickets.values():
            assert int(t.abs().sum()) == 0                             # counters are left clean
    # more launch streams than the plan's pool of counter sets: further sets are allocated on demand
    streams = [torch.cuda.Stream() for _ in range(20)]
    outs = []
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            outs.append(ops.prop_fwd(plan, H, S, S, avec, 8))
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    assert len(plan._tickets) >= 20 and all(torch.equal(o, ref[0]) for o in outs)


def test_inkernel_hub_finish_soak_with_poisoned_partial_sums(dev, capsys):
    """tools/hub_soak.py, short form (the 120 000-launch run is profiles/r03_hub_soak.json): rotating input tables, the
    partial-sum slots NaN-filled before every launch, forward and backward launches overlapped on two streams, every output
    word compared bitwise with the two-kernel path.  A stale read of another slice's sums -- from a previous launch or from
    the NaN fill sitting in an L2 -- would show as a mismatching launch."""
    import json
    from tools import hub_soak
    rc = hub_soak.main(["--launches", os.environ.get("NGACF_SOAK_LAUNCHES", "4000")])
    rep = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rc == 0 and rep["mismatches"] == 0 and rep["launches"] >= 4000, rep


@pytest.mark.parametrize("multi", [False, True])
def test_sparse_last_stage_equals_full_propagation(dev, multi):
    """model.forward / forward_pairs compute the LAST stage only for the rows of the batch (SPUIGACF.py:49-52 reads nothing
    else): scores and gradients equal to the full propagation's up to the order of the fp32 additions."""
    from ngacf_amd.SPUIGACF import SPUIGACF, SPUIMultiGACF
    from ngacf_amd.BPRLoss import BPRLoss
    U, I = 700, 500
    edge = _random_graph(U, I, 9000, 3, hub=600)
    adj = _t(edge, dev)
    torch.manual_seed(5)
    m = (SPUIMultiGACF if multi else SPUIGACF)(U, I, 64, [64, 64], 0.3).to(dev)
    with torch.no_grad():
        m.uEmbd.weight.mul_(30.0); m.iEmbd.weight.mul_(30.0)
    for mod in m.modules():
        if hasattr(mod, "seg_len"):
            mod.seg_len = 64                                              # long and split rows too
    gen = torch.Generator(device=dev).manual_seed(1)
    u = torch.randint(0, U, (96,), device=dev, generator=gen)
    pi = torch.randint(0, I, (96,), device=dev, generator=gen)
    ni = torch.randint(0, I, (96,), device=dev, generator=gen)
    out = {}
    for sparse in (False, True):
        m.sparse_last_stage = sparse
        m.zero_grad(set_to_none=True)
        m.eval()
        with torch.no_grad():
            s_eval = m(u, pi, adj)
        m.train(); torch.manual_seed(11)                                  # dropout seeds come from torch's CPU generator
        pos, neg = m.forward_pairs(u, pi, ni, adj)
        loss = BPRLoss()(pos, neg)
        loss.backward()
        out[sparse] = (s_eval, pos.detach(), neg.detach(), [p.grad.clone() for p in m.parameters()])
    for a, b in zip(out[False][:3], out[True][:3]):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max())
    for (n, _), a, b in zip(m.named_parameters(), out[False][3], out[True][3]):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-12, n
    # the full table stays available: propagate() ignores the batch
    m.eval()
    with torch.no_grad():
        full = m.propagate(adj)
    assert torch.isfinite(full).all() and full.shape[0] == U + I


def test_sparse_last_stage_cutoff(dev):
    """Beyond sparse_max_fraction (default 0.2 of the rows, the measured crossover is ~0.23) the modules run the full stage."""
    from ngacf_amd.SPUIGACF import SPUIGACF
    m = SPUIGACF(100, 80, 64, [64, 64], 0.1).to(dev)
    u = torch.zeros(10, dtype=torch.long, device=dev)
    m.sparse_max_fraction = 0.2
    assert m._batch_rows(u, u) is not None and m._batch_rows(u, u, u).n == 30          # 20 and 30 of 180 rows
    assert m._batch_rows(torch.zeros(19, dtype=torch.long, device=dev), torch.zeros(19, dtype=torch.long, device=dev)) is None   # 38 > 36
    m.sparse_last_stage = False
    assert m._batch_rows(u, u) is None


def test_integration_md_stub_runs(dev):
    """The ctypes binding shown in INTEGRATION.md is executed verbatim (library path redirected to the in-tree .so)
    and must reproduce the packaged stage."""
    import re
    from conftest import ROOT
    from ngacf_amd import functional as NF
    from ngacf_amd.plan import build_plan
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\nimport ctypes, torch\n(.*?)```", md, flags=re.S).group(0)[len("```python\n"):-3]
    code = code.replace('"libngacf_hip.so"', repr(os.path.join(ROOT, "ngacf_amd", "lib", "libngacf_hip.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = load_golden("g1_layer_k8_concat")
    U, I = int(g["U"]), int(g["I"])
    plan = build_plan(_t(g["edge"], dev), U, I)
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.randn(U + I, 64, device=dev, generator=gen)
    W = torch.randn(2, 64, 64, device=dev, generator=gen) * 0.2
    avec = torch.randn(2, 64, device=dev, generator=gen) * 0.2
    for heads in (8, 1):
        got = ns["stage"](plan, X, W, avec, heads)
        want = NF.stage(X, W, avec, plan, heads)
        assert torch.equal(got, want)


def test_g6_spga_layer_and_spgat_forward_backward(dev):
    """SPGA family (SPGA.py:330-421) on the bipartite double cover: the layer against the reference's own outputs
    (G6, k=8/concat and k=64/no-concat), SpGAT forward + every gradient against a float64 torch statement."""
    from ngacf_amd.SPGA import SpGAT, SpGraphAttentionLayer
    g = load_golden("g6_spga_layer")
    N = int(g["N"])
    X = _t(g["X"], dev)
    edge = _t(g["edge"], dev)
    adj = torch.zeros(N, N, device=dev)
    adj[edge[0], edge[1]] = 1.0
    for tag, k, concat in (("k8", 8, True), ("k64", 64, False)):
        l = SpGraphAttentionLayer(64, k, dropout=0.0, alpha=0.2, concat=concat).to(dev).eval()
        with torch.no_grad():
            l.W.copy_(_t(g["W_" + tag], dev)); l.a.copy_(_t(g["a_" + tag], dev))
        for a in (adj, edge):                                    # dense adjacency (reference call) or the edge list
            assert _maxabs(l(X, a), g["out_" + tag]) < 1e-5, tag
    # SpGAT, train mode without dropout, all gradients
    torch.manual_seed(2)
    m = SpGAT(64, 8, 64, 0.0, 0.2, 8).to(dev).train()
    Xr = X.clone().requires_grad_(True)
    out = m(Xr, adj)
    wgt = torch.randn(N, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    (out * wgt).sum().backward()

    def ref_layer(x, W, a, concat):
        k = W.shape[1]
        h = x @ W
        s = h[edge[0]] @ a[0, :k] + h[edge[1]] @ a[0, k:]
        e = torch.exp(-torch.nn.functional.leaky_relu(s, 0.2))
        R = torch.zeros(N, dtype=x.dtype, device=dev).index_add_(0, edge[0], e)
        hp = torch.zeros(N, k, dtype=x.dtype, device=dev).index_add_(0, edge[0], e[:, None] * h[edge[1]]) / R[:, None]
        return torch.nn.functional.elu(hp) if concat else hp

    P = {n: p.detach().double().requires_grad_(True) for n, p in m.named_parameters()}
    X64 = X.double().requires_grad_(True)
    x1 = torch.cat([ref_layer(X64, P["attention_%d.W" % h], P["attention_%d.a" % h], True) for h in range(8)], 1)
    ref = torch.nn.functional.elu(ref_layer(x1, P["out_att.W"], P["out_att.a"], False))
    (ref * wgt.double()).sum().backward()
    assert _maxabs(out, ref.detach().cpu().numpy()) < 1e-4
    assert _maxabs(Xr.grad, X64.grad.cpu().numpy()) < 2e-4 * float(X64.grad.abs().max())
    for n, p in m.named_parameters():
        assert _maxabs(p.grad, P[n].grad.cpu().numpy()) < 5e-4 * max(float(P[n].grad.abs().max()), 1e-9), n


def test_g15_spga_model_wrappers(dev):
    """SPGACF / MultiLayerSPGA / SPGAMGP (SPGA.py:85-316) on the HIP path against the reference's own scores; the
    reference's state_dicts load strictly; a backward pass reaches the embeddings."""
    from ngacf_amd import SPGA
    g = load_golden("g15_spga_models")
    U, I = int(g["U"]), int(g["I"])
    L = (torch.from_numpy(g["L_row"]), torch.from_numpy(g["L_col"]), torch.from_numpy(g["L_val"]))
    mask = _t(g["dense_adj"], dev).float()
    u, it = _t(g["u"], dev), _t(g["item"], dev)
    for kind, cls, layers in (("spgacf", SPGA.SPGACF, [64, 64]), ("multi", SPGA.MultiLayerSPGA, [64]), ("mgp", SPGA.SPGAMGP, [64, 64])):
        m = cls(U, I, L, 64, layers, 0.0)
        sd = {k[len(kind) + 3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(kind + ".p.")}
        res = m.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        m = m.to(dev).eval()
        sc = m(u, it, mask)
        want = g[kind + ".scores"]
        assert _maxabs(sc, want) < 1e-5 * max(float(np.abs(want).max()), 1.0), kind
        m.train()
        m(u, it, mask).sum().backward()
        assert float(m.uEmbd.weight.grad.abs().sum()) > 0 and bool(torch.isfinite(m.uEmbd.weight.grad).all())


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_next_stage_transform_in_the_traversal_epilogue(dev, dt):
    """ngacf_prop_fwd_next against the two-launch form it replaces (ngacf_prop_fwd, then ngacf_transform_fwd on the stored
    table): same `out` / agg / R AND the same H' / S', bit for bit -- the epilogue runs the stand-alone launch's tile step
    (bf16: csrc/mfma_rows.hpp, 32x32x16 bf16 matrix cores; fp32: the exact-fp32 16x16x4 matrix cores = transform.hip's fmaf
    chains) on rows that are still on chip.
    Every row class (short rows, long rows, hub rows finished in-kernel, an isolated item), both next-stage widths, one head and
    8-wide heads, with and without the next stage's feature dropout, with and without writing `out`."""
    from ngacf_amd import ops
    from ngacf_amd.plan import build_plan
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    rs = np.random.RandomState(5)
    U, I = 700, 500
    edge = _random_graph(U, I, 30000, 5, hub=450)
    plan = build_plan(_t(edge, dev), U, I, seg_len=128)
    assert plan.nsplit > 0 and plan.nseg > plan.nparts and plan.nshort > 0       # hub rows in slices, one-wave long rows, short rows
    N = U + I
    X = _t(rs.randn(N, 64).astype(np.float32), dev)
    W = _t((rs.randn(2, 64, 64) * 0.2).astype(np.float32), dev)
    avec = _t((rs.randn(2, 64) * 0.3).astype(np.float32), dev)
    for dout, heads2, p2 in ((64, 8, 0.0), (64, 1, 0.0), (128, 16, 0.0), (128, 1, 0.3), (64, 8, 0.25)):
        W2 = _t((rs.randn(2, 64, dout) * 0.2).astype(np.float32), dev)
        a2 = _t((rs.randn(2, dout) * 0.3).astype(np.float32), dev)
        if dt == "bf16":
            H, S, panels = ops.transform_fwd(X, W, avec, U, I, 8, tdt, pack=[W2])
            H0, S0 = ops.transform_fwd(X, W, avec, U, I, 8, tdt)
            assert torch.equal(H, H0) and torch.equal(S, S0)                   # the side jobs leave the transform alone
            panel = panels[0]
        else:
            H, S = ops.transform_fwd(X, W, avec, U, I, 8, tdt)
            panel = W2                                                         # fp32: the weights themselves
        for drop, save in ((0.0, False), (0.2, True)):
            ref = ops.prop_fwd(plan, H, S, S, avec, 8, drop_p=drop, seed=11, save=save)
            out_ref = ref[0] if save else ref
            H2r, S2r = ops.transform_fwd(out_ref, W2, a2, U, I, heads2, tdt, feat_drop_p=p2, seed=77)
            for want_out in (True, False):
                nxt = ops.NextStage(panel, a2, heads2, p2, 77)
                got = ops.prop_fwd(plan, H, S, S, avec, 8, drop_p=drop, seed=11, save=save, next=nxt, want_out=want_out)
                if want_out:
                    for a_, b_ in zip(got if save else (got,), ref if save else (ref,)):
                        assert torch.equal(a_, b_)
                elif save:
                    assert got[0] is None and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
                else:
                    assert got is None
                assert nxt.H.dtype == tdt
                assert torch.equal(nxt.H, H2r), (dt, dout, heads2, p2, drop, save, want_out, float((nxt.H.float() - H2r.float()).abs().max()))
                assert torch.equal(nxt.S, S2r), (dt, dout, heads2, p2, drop, save, want_out, float((nxt.S - S2r).abs().max()))


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
@pytest.mark.parametrize("kind", ["spuigacf", "multi", "d128L4"])
def test_fused_stack_equals_unfused_stack(dev, kind, dt):
    """The whole stack with NGACF_FUSE_NEXT on (default) and off: evaluation table, train-mode scores with dropout (same seeds =
    same masks) and every gradient, bf16 and fp32 storage.  (Both forms are checked against the reference's float64 numbers by the
    golden tests; this is the A/B of the two launch plans.)"""
    import ngacf_amd
    from ngacf_amd import functional as NF
    U, I = 900, 700
    edge = _t(_random_graph(U, I, 40000, 3, hub=500), dev)
    mk = {"spuigacf": lambda: ngacf_amd.SPUIGACF(U, I, 64, [64, 64], 0.1),
          "multi": lambda: ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.1),
          "d128L4": lambda: ngacf_amd.SPUIStackedGACF(U, I, 128, [64, 64], 0.1, stages=4)}[kind]
    torch.manual_seed(4)
    m = mk().to(dev)
    with torch.no_grad():
        m.uEmbd.weight.mul_(30.0); m.iEmbd.weight.mul_(30.0)
    if dt == "bf16":
        ngacf_amd.set_compute_dtype(m, torch.bfloat16)
    g = torch.Generator(device=dev); g.manual_seed(1)
    u, pi, ni = (torch.randint(0, k, (256,), device=dev, generator=g) for k in (U, I, I))
    res = {}
    was = NF.FUSE_NEXT
    try:
        for fuse in (True, False):
            NF.FUSE_NEXT = fuse
            m.eval()
            with torch.no_grad():
                F = m.propagate(edge).float()
            m.train()
            m.zero_grad(set_to_none=True)
            torch.manual_seed(9)
            loss = ngacf_amd.BPRLoss()(m(u, pi, edge), m(u, ni, edge))
            loss.backward()
            res[fuse] = (F, float(loss.detach()), {n: p.grad.clone() for n, p in m.named_parameters()})
    finally:
        NF.FUSE_NEXT = was
    assert torch.equal(res[True][0], res[False][0]) and res[True][1] == res[False][1]        # the same bits on both launch plans
    for n in res[True][2]:
        # the backward is the same code on the same saved tensors, and no kernel of the step orders its sums by timing (the score
        # gradient's float atomics are gone: ngacf_score_bwd): every gradient, bit for bit
        assert torch.equal(res[True][2][n], res[False][2][n]), n


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_ten_graphed_train_steps_are_bit_reproducible_under_a_seed(dev, dt):
    """Two runs from the same initial state and seeds, ten replayed reference-semantics train steps each (two propagations with
    dropout, scoring, BPR loss, backward, Adam): identical parameters, bit for bit.  Batches with repeated users and items (the
    score gradient's rows collect several pairs), hub rows cut into slices (finished by their last-arriving slice)."""
    import copy
    import ngacf_amd
    from ngacf_amd.graphs import GraphedBPRStep
    from ngacf_amd.optim import Adam
    U, I, B = 900, 700, 512
    edge = _t(_random_graph(U, I, 40000, 3, hub=500), dev)
    torch.manual_seed(4)
    base = ngacf_amd.SPUIMultiGACF(U, I, 64, [64, 64], 0.1).to(dev)
    with torch.no_grad():
        base.uEmbd.weight.mul_(30.0); base.iEmbd.weight.mul_(30.0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    batches = [torch.stack([torch.randint(0, U // 4, (B,), device=dev, generator=g), torch.randint(0, I // 4, (B,), device=dev, generator=g),
                            torch.randint(0, I // 4, (B,), device=dev, generator=g)]) for _ in range(10)]
    finals = []
    for run in range(2):
        m = copy.deepcopy(base)
        if dt == "bf16":
            ngacf_amd.set_compute_dtype(m, torch.bfloat16)
        m.train()
        opt = Adam(m.parameters(), lr=1e-2, weight_decay=1e-5)
        torch.manual_seed(77)
        step = GraphedBPRStep(m, ngacf_amd.BPRLoss(), opt, edge, B)
        losses = [float(step(b).detach()) for b in batches]
        torch.cuda.synchronize()
        finals.append(({n: p.detach().clone() for n, p in m.named_parameters()}, losses))
    assert finals[0][1] == finals[1][1]
    for n in finals[0][0]:
        assert torch.equal(finals[0][0][n], finals[1][0][n]), n
    assert finals[0][1][0] != finals[0][1][-1]                       # (the steps did move the parameters)


_STAGE_OUT_SCRIPT = r'''
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, %r)
from ngacf_amd import ops
from ngacf_amd.plan import build_plan
dev = torch.device("cuda:0")
rs = np.random.RandomState(9)
U, I = 700, 501                                   # (a last tile that is not full)
N = U + I
src = rs.randint(0, U, 30000); dst = rs.randint(0, I, 30000)
edge = torch.from_numpy(np.unique(np.stack([src, dst]), axis=1)).to(dev)
plan = build_plan(edge, U, I, seg_len=128)
h = hashlib.sha1()
def eat(*ts):
    for t in ts:
        h.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
X = torch.from_numpy(rs.randn(N, 64).astype(np.float32)).to(dev)
for xdt in (torch.float32, torch.bfloat16):
    for dout, heads in ((64, 8), (64, 1), (128, 16), (128, 1)):
        W = torch.from_numpy((rs.randn(2, 64, dout) * 0.2).astype(np.float32)).to(dev)
        a = torch.from_numpy((rs.randn(2, dout) * 0.3).astype(np.float32)).to(dev)
        H, S = ops.transform_fwd(X.to(xdt), W, a, U, I, heads, torch.bfloat16, feat_drop_p=0.2, seed=5)
        eat(H, S)
W = torch.from_numpy((rs.randn(2, 64, 64) * 0.2).astype(np.float32)).to(dev)
a = torch.from_numpy((rs.randn(2, 64) * 0.3).astype(np.float32)).to(dev)
for dout, heads in ((64, 8), (64, 1), (128, 16), (128, 1)):
    W2 = torch.from_numpy((rs.randn(2, 64, dout) * 0.2).astype(np.float32)).to(dev)
    a2 = torch.from_numpy((rs.randn(2, dout) * 0.3).astype(np.float32)).to(dev)
    H, S, panels = ops.transform_fwd(X, W, a, U, I, 8, torch.bfloat16, pack=[W2])
    nxt = ops.NextStage(panels[0], a2, heads, 0.1, 77)
    out = ops.prop_fwd(plan, H, S, S, a, 8, drop_p=0.1, seed=11, next=nxt)
    eat(out, nxt.H, nxt.S)
torch.cuda.synchronize()
print("DIGEST", h.hexdigest())
'''


def test_h_rows_leave_the_same_through_the_lds_tile_as_directly(dev):
    """The bf16 transform's H' rows are written either piece by piece from the accumulator layout or -- for tables that stream to
    HBM (mfma_rows.hpp: stage_out_pays, >= 64 MB) -- through the wave's LDS tile as whole 128-byte lines; NGACF_STAGE_OUT=0 / 1
    forces one form everywhere.  Same bits from the stand-alone launch (fp32 and bf16 input rows, 64 / 128 wide, 8-wide heads and
    one head, feature dropout, a ragged last tile) and from the traversal epilogue (ngacf_prop_fwd_next)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for force in ("0", "1"):
        env = dict(os.environ, NGACF_STAGE_OUT=force)
        r = subprocess.run([sys.executable, "-c", _STAGE_OUT_SCRIPT % root], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        digests.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert digests[0] == digests[1]
