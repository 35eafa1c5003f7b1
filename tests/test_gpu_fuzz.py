"""Randomised parity: one propagation stage (transform + traversal, forward and backward) on random small graphs with
random shapes of the plan (short-row cut-off, segment length, hubs, isolated items, ragged sizes), against the float64
oracle.  NGACF_FUZZ_ITERS sets the number of cases (default 12: a few seconds; a soak run uses hundreds)."""
import os

import numpy as np
import pytest
import torch

from oracle import ngacf_oracle as O

pytestmark = pytest.mark.gpu


def _case(seed):
    rs = np.random.RandomState(seed)
    U, I = int(rs.randint(1, 400)), int(rs.randint(1, 300))
    E = int(rs.randint(U, U + 1 + rs.randint(0, 6000)))
    u = np.concatenate([np.arange(U), rs.randint(0, U, E - U)])
    hi = max(1, I - int(rs.randint(0, 4)))                                  # some trailing items stay isolated
    i = rs.randint(0, hi, E)
    if rs.rand() < 0.5 and I > 1:                                            # a hub item and a hub user
        nh = int(rs.randint(1, U + 1))
        u = np.concatenate([u, rs.permutation(U)[:nh]]); i = np.concatenate([i, np.zeros(nh, dtype=np.int64)])
        ni = int(rs.randint(1, I + 1))
        u = np.concatenate([u, np.zeros(ni, dtype=np.int64)]); i = np.concatenate([i, rs.permutation(I)[:ni]])
    key = np.unique(u.astype(np.int64) * I + i)
    edge = np.stack([key // I, key % I])
    heads = int(rs.choice([1, 8]))
    short_max = int(rs.choice([-1, 0, 1, 4, 64, 1 << 20]))
    seg_len = int(rs.choice([1, 3, 8, 64, 256]))
    return U, I, edge, heads, short_max, seg_len, rs


@pytest.mark.parametrize("seed", range(int(os.environ.get("NGACF_FUZZ_ITERS", "12"))))
def test_random_stage_forward_backward_vs_oracle(seed):
    from ngacf_amd import functional as NF
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    U, I, edge, heads, short_max, seg_len, rs = _case(1000 + seed)
    N, d = U + I, 64
    k = d // heads
    X = rs.randn(N, d) * 0.7
    Wl = [(rs.randn(d, k) * 0.3, rs.randn(d, k) * 0.3, rs.randn(1, 2 * k) * 0.4) for _ in range(heads)]
    wgt = rs.randn(N, d)
    # oracle: heads side by side (SPUIGACF.py:212), ELU inside; gradient of sum(out * wgt)
    # half of the cases in train mode: edge dropout on the numerators (SPUIGACF.py:375) and feature dropout in front of
    # the stage (:208), both from the host statement of the counter hash (bit-exact with the device, ngacf_amd/rng.py)
    p_drop = float(rs.choice([0.0, 0.0, 0.1, 0.5]))
    dseed = int(rs.randint(1, 2 ** 31))
    em = [None] * heads
    if p_drop > 0:
        from ngacf_amd import rng
        nn_, cc_ = np.meshgrid(np.arange(N), np.arange(d), indexing="ij")
        X_in = X
        X = X * rng.keep_scale(rng.feat_bits(dseed, nn_, cc_), p_drop).astype(np.float64)      # what the stage sees
        em = [rng.keep_scale(rng.edge_bits(dseed, h, edge[0], edge[1] + U), p_drop).astype(np.float64) for h in range(heads)]
    cache = [O.layer_forward(X, Wu, Wi, a, edge, U, I, concat=True, edge_mask=em[h]) for h, (Wu, Wi, a) in enumerate(Wl)]
    out_o = np.concatenate([c["out"] for c in cache], 1)
    if min(float(np.abs(c["s"]).min()) for c in cache if c["s"].size) < 1e-5:
        # an edge score within fp32 rounding of LeakyReLU's kink: float64 and fp32 may take different slopes there
        # (seen once in 8.6 M edge-heads: seed 74); the gradient is not defined to that precision
        pytest.skip("edge score at the LeakyReLU kink")
    gX_o = np.zeros_like(X)
    gW_o = []
    for h, ((Wu, Wi, a), c) in enumerate(zip(Wl, cache)):
        gx, gwu, gwi, ga = O.layer_backward(wgt[:, h * k:(h + 1) * k], c, X, Wu, Wi, a, edge, U, I, concat=True, edge_mask=em[h])
        gX_o += gx
        gW_o.append((gwu, gwi, ga))
    if p_drop > 0:                                                          # chain rule through the feature mask
        gX_o = gX_o * rng.keep_scale(rng.feat_bits(dseed, nn_, cc_), p_drop).astype(np.float64)
        X = X_in
    # hub layout: the default (workgroup per row beyond 4 * seg_len), no hubs (slices + partial sums), or both
    hub_cap = [None, 0, int(rs.randint(1, 200))][seed % 3]
    long_max = None if seed % 2 else seg_len
    plan = build_plan(torch.from_numpy(edge).to(dev), U, I, seg_len=seg_len, short_max=short_max, hub_cap=hub_cap, long_max=long_max)
    Xt = torch.from_numpy(X).float().to(dev).requires_grad_(True)
    W = torch.from_numpy(np.stack([np.concatenate([w[0] for w in Wl], 1), np.concatenate([w[1] for w in Wl], 1)])).float().to(dev).requires_grad_(True)
    av = torch.from_numpy(np.stack([np.concatenate([w[2][0, :k] for w in Wl]), np.concatenate([w[2][0, k:] for w in Wl])])).float().to(dev).requires_grad_(True)
    out = NF.stage(Xt, W, av, plan, heads, 0.2, True, p_drop, p_drop, dseed)
    (out * torch.from_numpy(wgt).float().to(dev)).sum().backward()
    tag = (seed, U, I, edge.shape[1], heads, short_max, seg_len, p_drop, hub_cap, long_max, plan.nhub, plan.nsplit)
    assert float(np.abs(out.detach().cpu().numpy() - out_o).max()) < 1e-4 * max(1.0, float(np.abs(out_o).max())), tag
    assert float(np.abs(Xt.grad.cpu().numpy() - gX_o).max()) < 2e-4 * max(1.0, float(np.abs(gX_o).max())), tag
    gWu = np.concatenate([g[0] for g in gW_o], 1); gWi = np.concatenate([g[1] for g in gW_o], 1)
    assert float(np.abs(W.grad[0].cpu().numpy() - gWu).max()) < 2e-4 * max(1.0, float(np.abs(gWu).max())), tag
    assert float(np.abs(W.grad[1].cpu().numpy() - gWi).max()) < 2e-4 * max(1.0, float(np.abs(gWi).max())), tag
    ga_u = np.concatenate([g[2][0, :k] for g in gW_o]); ga_i = np.concatenate([g[2][0, k:] for g in gW_o])
    assert float(np.abs(av.grad[0].cpu().numpy() - ga_u).max()) < 2e-4 * max(1.0, float(np.abs(ga_u).max())), tag
    assert float(np.abs(av.grad[1].cpu().numpy() - ga_i).max()) < 2e-4 * max(1.0, float(np.abs(ga_i).max())), tag


@pytest.mark.parametrize("seed", range(max(8, int(os.environ.get("NGACF_FUZZ_ITERS", "12")) // 3)))
def test_random_eval_topk_and_sampler_vs_host(seed, monkeypatch):
    """ngacf_eval_topk against masked scores + stable sort (integer tables: exact, with ties), ngacf_sample_pairs against
    its numpy statement, on random sizes: ragged user blocks, tiny pools, K from 1 to 32, users with few candidates;
    the kernel's form (wave-private / LDS-shared with 8 or 12 waves) drawn per case."""
    from ngacf_amd import ops, rng
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(5000 + seed)
    form = ["wave", "lds4", "lds8", "lds12"][seed % 4]
    monkeypatch.setenv("NGACF_TOPK_FORM", "wave" if form == "wave" else "lds")
    if form != "wave":
        monkeypatch.setenv("NGACF_TOPK_LDS_WAVES", form[3:])
    U, I = int(rs.randint(1, 300)), int(rs.randint(1, 900))
    d = int(rs.choice([64, 128]))
    K = int(rs.randint(1, 33))
    dt = torch.bfloat16 if rs.rand() < 0.5 else torch.float32
    g = torch.Generator(device=dev).manual_seed(seed)
    F = torch.randint(-2, 3, (U + I, d), device=dev, generator=g).float()
    pool = torch.rand(I, device=dev, generator=g) < float(rs.choice([0.05, 0.5, 1.0]))
    pool[int(rs.randint(0, I))] = True
    deg = rs.randint(0, min(I, 90) + 1, U)
    deg[rs.randint(0, U)] = min(I, int(rs.randint(0, 400)))                     # one row beyond the 64-entry LDS cache
    rows = np.repeat(np.arange(U), deg)
    cols = np.concatenate([rs.permutation(I)[:n] for n in deg]) if rows.size else np.zeros(0, dtype=np.int64)
    key = np.unique(rows.astype(np.int64) * I + cols)
    ku, ki = torch.from_numpy(key // I).to(dev), torch.from_numpy(key % I).to(dev)
    ptr = torch.zeros(U + 1, dtype=torch.long, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(ku, minlength=U), 0)
    users = torch.from_numpy(rs.permutation(U)[:int(rs.randint(1, U + 1))]).to(dev)
    top, top_s = ops.eval_topk(F.to(dt), users, U, I, pool.to(torch.uint8), ptr, ki.to(torch.int32), K)
    sc = F[users] @ F[U:].t()
    neg = torch.finfo(torch.float32).min
    sc[:, ~pool] = neg
    mask = torch.zeros(U, I, dtype=torch.bool, device=dev); mask[ku, ki] = True
    sc[mask[users]] = neg
    order = torch.sort(sc, dim=1, descending=True, stable=True)
    kk = min(K, I)
    want_i, want_s = order.indices[:, :kk], order.values[:, :kk]
    valid = want_s > neg
    tag = (seed, U, I, d, K, str(dt), form)
    assert torch.equal(top[:, :kk].long()[valid], want_i[valid]) and bool((top[:, :kk][~valid] == -1).all()), tag
    assert bool((top[:, kk:] == -1).all()) and torch.equal(top_s[:, :kk][valid], want_s[valid]), tag
    # sampler: only users with a train positive and a legal negative can be sampled
    ok_users = [u for u in range(U) if deg[u] > 0 and int((pool.cpu().numpy() & ~mask[u].cpu().numpy()).sum()) > 0]
    if ok_users:
        B = int(rs.randint(1, 700))
        bu = torch.from_numpy(rs.choice(ok_users, B)).to(dev)
        pool_items = torch.nonzero(pool).squeeze(1).to(torch.int32)
        sd = int(rs.randint(0, 2 ** 40))
        upn, unres = ops.sample_pairs(bu, 0, B, ptr, ki.to(torch.int32), pool_items, sd)
        pos, ng, un = rng.sample_pairs_host(bu.cpu().numpy(), ptr.cpu().numpy(), ki.to(torch.int32).cpu().numpy(), pool_items.cpu().numpy(), sd)
        assert int(unres) == un, tag
        assert np.array_equal(upn[1].cpu().numpy(), pos) and np.array_equal(upn[2].cpu().numpy(), ng), tag


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("NGACF_FUZZ_ITERS", "12"))))
def test_flagged_rows_stage_equals_full_stage_on_those_rows(seed, dtype):
    """`needed` rows (the last stage of a scored batch, SPUIGACF.py:49-52 reads 2B rows of it): the batch's rows of the output
    match the full stage's, and with a gradient that is zero outside them every input gradient matches the full stage's --
    same terms, summed in another order (a wave per batch row instead of a lane group): rounding differences only."""
    from ngacf_amd import functional as NF, ops
    from ngacf_amd.plan import build_plan
    dev = torch.device("cuda:0")
    U, I, edge, heads, short_max, seg_len, rs = _case(5000 + seed)
    N, d = U + I, 64
    dt = torch.float32 if dtype == "fp32" else torch.bfloat16
    plan = build_plan(torch.as_tensor(edge, device=dev), U, I, seg_len=seg_len, short_max=short_max,
                      hub_cap=int(rs.choice([0, 0, 2 * seg_len])))
    X0 = torch.as_tensor(rs.randn(N, d) * 0.7, dtype=torch.float32, device=dev)
    W0 = torch.as_tensor(rs.randn(2, d, d) * 0.3, dtype=torch.float32, device=dev)
    a0 = torch.as_tensor(rs.randn(2, d) * 0.4, dtype=torch.float32, device=dev)
    B = int(rs.randint(1, 40))
    ub = torch.as_tensor(rs.randint(0, U, B), device=dev)
    ib = torch.as_tensor(rs.randint(0, I, B), device=dev)
    needed = ops.mark_rows(ub, ib, U, I)
    want_flags = np.zeros(N, np.uint8); want_flags[ub.cpu().numpy()] = 1; want_flags[U + ib.cpu().numpy()] = 1
    assert np.array_equal(needed.mask().cpu().numpy(), want_flags.astype(bool)) and needed.n == 2 * B
    assert np.array_equal(needed.rows[:2 * B].cpu().numpy(), np.concatenate([ub.cpu().numpy(), U + ib.cpu().numpy()]))
    nb = needed.mask()
    wgt = torch.as_tensor(rs.randn(N, d), dtype=torch.float32, device=dev) * nb[:, None]
    p_drop = float(rs.choice([0.0, 0.3]))
    res = []
    for flags in (None, needed):
        X, W, a = (t.clone().requires_grad_() for t in (X0, W0, a0))
        out = NF.stage(X, W, a, plan, heads, 0.2, True, p_drop, p_drop, 77 + seed, dt, needed=flags)
        (out.float() * wgt).sum().backward()
        res.append((out.detach(), X.grad, W.grad, a.grad))
    (o_f, *g_f), (o_n, *g_n) = res
    otol = (1e-5 if dtype == "fp32" else 1e-2) * float(o_f[nb].float().abs().max()) + 1e-12
    assert float((o_f[nb].float() - o_n[nb].float()).abs().max()) <= otol
    for name, x, y in zip(("dX", "dW", "da"), g_f, g_n):
        assert torch.isfinite(y).all(), name
        tol = (2e-5 if dtype == "fp32" else 2e-2) * float(x.abs().max()) + 1e-12
        assert float((x - y).abs().max()) <= tol, (name, float((x - y).abs().max()), float(x.abs().max()))
