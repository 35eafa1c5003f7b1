"""CPU baseline for bench.py's `cpu_baseline` leg: the propagation forward written with
the SAME torch CPU operator sequence the reference executes per layer
(graphattention/SPUIGACF.py:356-397: mm, index gather, cat, a.mm, leaky_relu, exp,
sparse_coo_tensor, 4x torch.sparse.mm, div, NaN->0, residual, cat, elu), one layer per
head, heads concatenated (:212), so its timing stands in for "the reference's CPU forward
on this box's host cores" -- the reference's own Python cannot travel to the GPU box.

TEST INFRASTRUCTURE ONLY (see oracle/ngacf_oracle.py).  Pinned to the golden fixtures by
tests/test_oracle_golden.py::test_cpu_baseline_matches_goldens.
"""
import time

import torch
import torch.nn.functional as F


def layer(x, W_u, W_i, a, edge, U, I, alpha, concat):
    hu = torch.mm(x[:U], W_u)
    hi = torch.mm(x[U:U + I], W_i)
    pair = torch.cat((hu[edge[0]], hi[edge[1]]), dim=1).t()
    e = torch.exp(-F.leaky_relu(a.mm(pair).squeeze(0), alpha))
    att = torch.sparse_coo_tensor(edge, e, (U, I))
    rs = torch.sparse.mm(att, torch.ones(I, 1, dtype=x.dtype))
    cs = torch.sparse.mm(att.t(), torch.ones(U, 1, dtype=x.dtype))
    nu = torch.sparse.mm(att, hi).div(rs)
    ni = torch.sparse.mm(att.t(), hu).div(cs)
    ni[torch.isnan(ni)] = 0.0
    out = torch.cat([hu + nu, hi + ni], dim=0)
    return F.elu(out) if concat else out


def forward(params, x, edge, U, I, stages, alpha=0.2):
    """params: reference-keyed dict of torch CPU tensors; stages: list of lists of prefixes."""
    for si, names in enumerate(stages):
        last = si == len(stages) - 1
        outs = [layer(x, params[n + ".W_u"], params[n + ".W_i"], params[n + ".a"], edge, U, I, alpha, not last)
                for n in names]
        x = torch.cat(outs, dim=1)
        if last:
            x = F.elu(x)
    return x


def time_forward(params, x, edge, U, I, stages, budget_s=15.0, max_iters=5):
    """(seconds per forward, iterations timed).  One untimed warm-up, then forwards until the
    budget is spent."""
    with torch.no_grad():
        forward(params, x, edge, U, I, stages)
        n, t0 = 0, time.perf_counter()
        while n < max_iters and (n == 0 or time.perf_counter() - t0 < budget_s):
            forward(params, x, edge, U, I, stages)
            n += 1
        return (time.perf_counter() - t0) / n, n
