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
    cache = [O.layer_forward(X, Wu, Wi, a, edge, U, I, concat=True) for Wu, Wi, a in Wl]
    out_o = np.concatenate([c["out"] for c in cache], 1)
    if min(float(np.abs(c["s"]).min()) for c in cache if c["s"].size) < 1e-5:
        # an edge score within fp32 rounding of LeakyReLU's kink: float64 and fp32 may take different slopes there
        # (seen once in 8.6 M edge-heads: seed 74); the gradient is not defined to that precision
        pytest.skip("edge score at the LeakyReLU kink")
    gX_o = np.zeros_like(X)
    gW_o = []
    for h, ((Wu, Wi, a), c) in enumerate(zip(Wl, cache)):
        gx, gwu, gwi, ga = O.layer_backward(wgt[:, h * k:(h + 1) * k], c, X, Wu, Wi, a, edge, U, I, concat=True)
        gX_o += gx
        gW_o.append((gwu, gwi, ga))
    plan = build_plan(torch.from_numpy(edge).to(dev), U, I, seg_len=seg_len, short_max=short_max)
    Xt = torch.from_numpy(X).float().to(dev).requires_grad_(True)
    W = torch.from_numpy(np.stack([np.concatenate([w[0] for w in Wl], 1), np.concatenate([w[1] for w in Wl], 1)])).float().to(dev).requires_grad_(True)
    av = torch.from_numpy(np.stack([np.concatenate([w[2][0, :k] for w in Wl]), np.concatenate([w[2][0, k:] for w in Wl])])).float().to(dev).requires_grad_(True)
    out = NF.stage(Xt, W, av, plan, heads, 0.2, True)
    (out * torch.from_numpy(wgt).float().to(dev)).sum().backward()
    tag = (seed, U, I, edge.shape[1], heads, short_max, seg_len)
    assert float(np.abs(out.detach().cpu().numpy() - out_o).max()) < 1e-4 * max(1.0, float(np.abs(out_o).max())), tag
    assert float(np.abs(Xt.grad.cpu().numpy() - gX_o).max()) < 2e-4 * max(1.0, float(np.abs(gX_o).max())), tag
    gWu = np.concatenate([g[0] for g in gW_o], 1); gWi = np.concatenate([g[1] for g in gW_o], 1)
    assert float(np.abs(W.grad[0].cpu().numpy() - gWu).max()) < 2e-4 * max(1.0, float(np.abs(gWu).max())), tag
    assert float(np.abs(W.grad[1].cpu().numpy() - gWi).max()) < 2e-4 * max(1.0, float(np.abs(gWi).max())), tag
    ga_u = np.concatenate([g[2][0, :k] for g in gW_o]); ga_i = np.concatenate([g[2][0, k:] for g in gW_o])
    assert float(np.abs(av.grad[0].cpu().numpy() - ga_u).max()) < 2e-4 * max(1.0, float(np.abs(ga_u).max())), tag
    assert float(np.abs(av.grad[1].cpu().numpy() - ga_i).max()) < 2e-4 * max(1.0, float(np.abs(ga_i).max())), tag
