"""GPU parity of the dense multi-head attention path (csrc/mha.hip, ngacf_amd/modules.py) through the
C ABI: against the reference's own outputs and float64 gradients (tests/golden/g14_dense_mha.npz), against the numpy
oracle with the dropout masks restated on the host, and at ml100k size against a float64 torch statement."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ngacf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _t(x, dev, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev).to(dtype)


def _rel(a, b):
    b = torch.as_tensor(b, device=a.device).double()
    return float((a.detach().double() - b).abs().max() / b.abs().max().clamp_min(1e-300))


def _load(module, g, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}
    missing = module.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return module


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_g14_multi_head_attention_forward_and_gradients(dev, tag):
    """MultiHeadAttention (modules.py:62-107): output vs the reference's float64 run and every gradient (inputs, the four
    Linear layers) vs the reference's float64 autograd; Nq = 45, Nk = 70 (not multiples of the 32-wide tiles), one fully
    masked query row (uniform attention, no gradient through its scores).  Exact fp32 matrix cores: 1e-5."""
    from ngacf_amd.modules import MultiHeadAttention
    g = load_golden("g14_dense_mha")
    m = _load(MultiHeadAttention(int(g["heads"]), 64, 0.0), g, "mha.p.").to(dev)
    q = _t(g["q_in"], dev).requires_grad_(True)
    kv = _t(g["k_in"], dev).requires_grad_(True)
    mask = None if tag == "nomask" else _t(g["mask"], dev)
    out = m(q, kv, kv, mask)
    tol = 1e-5
    assert _rel(out, g["mha.%s.out_f64" % tag]) < tol
    (out * _t(g["G"], dev)).sum().backward()
    errs = {"q_in": _rel(q.grad, g["mha.%s.g.q_in" % tag]), "kv_in": _rel(kv.grad, g["mha.%s.g.kv_in" % tag])}
    for k, p in m.named_parameters():
        if k == "k_linear.bias":
            # softmax is invariant to a constant added to every key: this gradient is analytically 0 (the reference's
            # float64 run leaves ~1e-17); compare it on the scale of its neighbour, the k weight gradient
            scale = float(np.abs(g["mha.%s.g.k_linear.weight" % tag]).max())
            errs[k] = float(p.grad.abs().max()) / scale
            assert np.abs(g["mha.%s.g.%s" % (tag, k)]).max() < 1e-12 * scale
        else:
            errs[k] = _rel(p.grad, g["mha.%s.g.%s" % (tag, k)])
    assert max(errs.values()) < 2e-5, errs
    if tag == "mask":                                     # the fully masked row: uniform attention over all keys
        V = torch.nn.functional.linear(kv.detach(), m.v_linear.weight, m.v_linear.bias)
        want = torch.nn.functional.linear(V.mean(0, keepdim=True), m.out.weight, m.out.bias)
        assert float((out[3] - want[0]).abs().max()) < 1e-5


def test_g14_attlayer_and_models(dev):
    """ATTLayer (modules.py:7-18), GACFV1 (GACFmodel1.py) and GACFMask (GACFMask.py) forward against the reference's outputs;
    state_dicts load strictly (same parameter names)."""
    from ngacf_amd.modules import ATTLayer, GACFMask, GACFV1
    g = load_golden("g14_dense_mha")
    att = _load(ATTLayer(64, int(g["heads"]), 0.0), g, "att.p.").to(dev)
    out = att(_t(g["q_in"], dev), _t(g["k_in"], dev))
    assert float((out - _t(g["att.out"], dev)).abs().max()) < 1e-5
    U, I = int(g["U"]), int(g["I"])
    L = (torch.from_numpy(g["L_row"]), torch.from_numpy(g["L_col"]), torch.from_numpy(g["L_val"]))
    u, it = torch.from_numpy(g["u"]).to(dev), torch.from_numpy(g["item"]).to(dev)
    m1 = _load(GACFV1(U, I, L, 64, [64, 64], 0.0), g, "v1.p.").to(dev).eval()
    s1 = m1(u, it)
    assert _rel(s1, g["v1_scores"]) < 1e-5
    m2 = _load(GACFMask(U, I, L, 64, [64, 64], 0.0), g, "mk.p.").to(dev).eval()
    mask = _t(g["dense_adj"], dev)
    s2 = m2(u, it, mask)
    assert _rel(s2, g["mk_scores"]) < 1e-5
    # and they train: a backward pass through attention, Laplacian products and embeddings gives finite gradients
    m2.train()
    m2(u, it, mask).sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m2.parameters())
    assert float(m2.gacf.uEmbd.weight.grad.abs().sum()) > 0


def test_attention_dropout_matches_host_restated_masks(dev):
    """Train mode: dropout multiplies the softmax PROBABILITIES (modules.py:41-42; the normaliser is undropped).  The
    device masks are the counter hash of (seed, head, query, key); restated on the host they go into the numpy oracle."""
    from ngacf_amd import ops, rng
    rs = np.random.RandomState(2)
    Nq, Nk, dim, heads, p, seed = 70, 50, 64, 8, 0.3, 4242
    Q, K, V = (rs.randn(n, dim).astype(np.float32) for n in (Nq, Nk, Nk))
    mask = (rs.rand(Nq, Nk) < 0.5).astype(np.uint8)
    qq, kk = np.meshgrid(np.arange(Nq), np.arange(Nk), indexing="ij")
    pm = np.stack([rng.keep_scale(rng.edge_bits(seed, h, qq, kk), p) for h in range(heads)]).astype(np.float64)
    ident = {"%s.weight" % n: np.eye(dim) for n in ("q_linear", "k_linear", "v_linear", "out")}
    ident.update({"%s.bias" % n: np.zeros(dim) for n in ("q_linear", "k_linear", "v_linear", "out")})
    # the oracle applies the linears (identity here) and the core; K and V differ, so call with separate k, v
    want = O.multi_head_attention(ident, Q.astype(np.float64), K.astype(np.float64), V.astype(np.float64), heads, mask, pm)
    Od, LSE = ops.mha_fwd(_t(Q, dev), _t(K, dev), _t(V, dev), heads, torch.from_numpy(mask).to(dev), p, seed)
    assert _rel(Od, want) < 1e-5
    assert 0.25 < float((pm == 0).mean()) < 0.35
    # gradients with the same masks: float64 torch autograd of the restated formula
    Qt, Kt, Vt = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (Q, K, V))
    split = lambda x: x.view(-1, heads, dim // heads).transpose(0, 1)
    s = split(Qt) @ split(Kt).transpose(1, 2) / np.sqrt(dim // heads)
    s = s.masked_fill(torch.from_numpy(mask)[None] == 0, -1e9)
    Pm = torch.softmax(s, -1) * torch.from_numpy(pm)
    o = (Pm @ split(Vt)).transpose(0, 1).reshape(Nq, dim)
    G = rs.randn(Nq, dim).astype(np.float32)
    (o * torch.from_numpy(G)).sum().backward()
    dQ, dK, dV = ops.mha_bwd(_t(Q, dev), _t(K, dev), _t(V, dev), Od, LSE, _t(G, dev), heads, torch.from_numpy(mask).to(dev), p, seed)
    assert _rel(dQ, Qt.grad) < 2e-5 and _rel(dK, Kt.grad) < 2e-5 and _rel(dV, Vt.grad) < 2e-5


@pytest.mark.parametrize("dim,heads", [(64, 8), (128, 8), (256, 8)])
def test_ml100k_size_attention_vs_float64(dev, dim, heads):
    """The reference's largest case: all N = 2625 ml100k nodes attend to each other under the adjacency mask
    (ATTLayer_mask); d_k = 8, 16, 32.  Forward and the three input gradients vs float64 torch on the GPU."""
    from ngacf_amd import data as D, ops
    u, i, _ = D.load_ml100k()
    U, I = 943, 1682
    N = U + I
    mask = torch.zeros(N, N, dtype=torch.uint8, device=dev)
    ut, it = torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev) + U
    mask[ut, it] = 1; mask[it, ut] = 1
    mask[torch.arange(N), torch.arange(N)] = 1
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    Q, K, V, G = (torch.randn(N, dim, device=dev, generator=gen) * s for s in (1.0, 1.0, 1.0, 1.0))
    Od, LSE = ops.mha_fwd(Q, K, V, heads, mask)
    dQ, dK, dV = ops.mha_bwd(Q, K, V, Od, LSE, G, heads, mask)
    Qd, Kd, Vd = (x.double().requires_grad_(True) for x in (Q, K, V))
    dk = dim // heads
    split = lambda x: x.view(-1, heads, dk).transpose(0, 1)
    s = (split(Qd) @ split(Kd).transpose(1, 2) / np.sqrt(dk)).masked_fill(mask[None] == 0, -1e9)
    o = (torch.softmax(s, -1) @ split(Vd)).transpose(0, 1).reshape(N, dim)
    (o * G.double()).sum().backward()
    assert _rel(Od, o.detach()) < 1e-5
    assert _rel(dQ, Qd.grad) < 5e-5 and _rel(dK, Kd.grad) < 5e-5 and _rel(dV, Vd.grad) < 5e-5
    # the functional form with the reference's [h, N, d_k] layout (modules.py:32-46)
    from ngacf_amd.modules import attention
    Of = attention(split(Q).contiguous(), split(K).contiguous(), split(V).contiguous(), dk, mask)
    assert torch.equal(Of.transpose(0, 1).reshape(N, dim), Od)
    # refused shapes
    with pytest.raises(ValueError):
        ops.mha_fwd(Q[:, :48].contiguous(), K[:, :48].contiguous(), V[:, :48].contiguous(), 8)


def test_g16_gacf_variants(dev):
    """GACFV2 .. GACFV5 (GACFmodel2-5.py) on the HIP path against the reference's scores; V3-V5 load GACFV1's state_dict
    (same parameter names), V2 its own -- strictly."""
    from ngacf_amd import modules as M
    g14, g = load_golden("g14_dense_mha"), load_golden("g16_gacf_variants")
    U, I = int(g14["U"]), int(g14["I"])
    L = (torch.from_numpy(g14["L_row"]), torch.from_numpy(g14["L_col"]), torch.from_numpy(g14["L_val"]))
    u, it = torch.from_numpy(g14["u"]).to(dev), torch.from_numpy(g14["item"]).to(dev)
    for v, cls in ((3, M.GACFV3), (4, M.GACFV4), (5, M.GACFV5)):
        m = _load(cls(U, I, L, 64, [64, 64], 0.0), g14, "v1.p.").to(dev).eval()
        assert _rel(m(u, it), g["v%d_scores" % v]) < 1e-5, v
    m2 = _load(M.GACFV2(U, I, L, 64, [64, 64], 0.0), g, "v2.p.").to(dev).eval()
    assert _rel(m2(u, it), g["v2_scores"]) < 1e-5


def test_g17_gacfv6_and_gacfv3x(dev):
    """GACFV6 (its Laplacian built from the rating table, as the reference's constructor does) and GACFV3x against the
    reference's scores; both load the reference's state_dict strictly."""
    import pandas as pd
    from ngacf_amd import modules as M
    g14, g = load_golden("g14_dense_mha"), load_golden("g17_gacf_mlp_heads")
    U, I = int(g14["U"]), int(g14["I"])
    e = g14["edge"]
    u, it = torch.from_numpy(g14["u"]).to(dev), torch.from_numpy(g14["item"]).to(dev)
    rt = pd.DataFrame({"userId": e[0], "itemId": e[1], "rating": g["rating"].astype(np.float64)})
    m6 = _load(M.GACFV6(U, I, rt, 64, [64, 64, 64], 0.0), g, "v6.p.").to(dev).eval()
    r, c, v = M.rating_laplacian_coo(e[0], e[1], g["rating"], U, I)
    ours = torch.sparse_coo_tensor(torch.stack([r, c]), v, (U + I, U + I)).coalesce().to_dense()
    ref = torch.sparse_coo_tensor(torch.from_numpy(np.stack([g["v6_L_row"], g["v6_L_col"]])), torch.from_numpy(g["v6_L_val"]), (U + I, U + I)).to_dense()
    assert float((ours - ref).abs().max()) < 1e-6
    assert _rel(m6(u, it), g["v6_scores"]) < 2e-5
    L = (torch.from_numpy(g14["L_row"]), torch.from_numpy(g14["L_col"]), torch.from_numpy(g14["L_val"]))
    m3 = _load(M.GACFV3x(U, I, L, 64, [64, 64, 64], 0.0), g, "v3x.p.").to(dev).eval()
    assert _rel(m3(u, it), g["v3x_scores"]) < 2e-5
    # gradients flow through the MLP head, the attention kernels and the Laplacian products
    m3.train()
    m3(u, it).sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m3.parameters())


def test_default_width_256_models_vs_oracle(dev):
    """The reference's DEFAULT configuration of the dense-attention models (embedSize 256, layers 256-128-64: GACFmodel1.py:43,
    GACFmodel6.py:47) now constructs and runs (the first rounds refused width 256): GACFV1 and GACFV6 with default
    arguments against the numpy oracle (itself pinned by g14 / g16 / g17 at width 64)."""
    import scipy.sparse as sp
    from oracle import ngacf_oracle as O
    from ngacf_amd import modules as M
    g14 = load_golden("g14_dense_mha")
    U, I = int(g14["U"]), int(g14["I"])
    N = U + I
    L = (torch.from_numpy(g14["L_row"]), torch.from_numpy(g14["L_col"]), torch.from_numpy(g14["L_val"]))
    Ls = sp.coo_matrix((g14["L_val"].astype(np.float64), (g14["L_row"], g14["L_col"])), shape=(N, N)).tocsr()
    u, it = g14["u"], g14["item"]
    torch.manual_seed(8)
    m1 = M.GACFV1(U, I, L, 256, [256, 128, 64], 0.0)
    assert m1.gacf.uEmbd.weight.shape[1] == 256 and len(m1.gacf.GALayers) == 2
    with torch.no_grad():
        m1.gacf.uEmbd.weight.mul_(30.0); m1.gacf.iEmbd.weight.mul_(30.0)
    p1 = {k: v.detach().numpy().astype(np.float64) for k, v in m1.state_dict().items()}
    got = m1.to(dev).eval()(torch.from_numpy(u).to(dev), torch.from_numpy(it).to(dev))
    assert _rel(got, O.gacfv1_scores(p1, Ls, u, it, U, I)) < 5e-5
    layer_default = M.GACFV1_layer(U, I, L)                               # default arguments: embedSize 256, layers (256, 128, 64)
    assert layer_default.uEmbd.weight.shape[1] == 256
    m6 = M.GACFV6(U, I, L)                                                # default widths, a ready Laplacian
    with torch.no_grad():
        m6.uEmbd.weight.mul_(30.0); m6.iEmbd.weight.mul_(30.0)
    p6 = {k: v.detach().numpy().astype(np.float64) for k, v in m6.state_dict().items()}
    got6 = m6.to(dev).eval()(torch.from_numpy(u).to(dev), torch.from_numpy(it).to(dev))
    ref6 = O.gacf_mlp_scores(p6, Ls, u, it, U, I, "v6")
    assert float((got6.detach().cpu().double() - torch.from_numpy(ref6)).abs().max()) < 5e-5 * max(float(np.abs(ref6).max()), 1.0)
