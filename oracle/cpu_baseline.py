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


# ---------------------------------------------------------------------------------------------------------------------------
# The train leg ("... and 2 timed train steps"): one reference-semantics BPR step = two propagations with
# independent dropout (train_eval_Gowalla.py:131-133 calls the model twice), scoring (SPUIGACF.py:49-52), BPRLoss (BPRLoss.py:8-9),
# loss.backward() through torch's own autograd of the same operator sequence (the sparse.mm backward is what makes the
# reference's step 70-95 x its forward at Gowalla scale) and torch.optim.Adam(weight_decay) (run_Gowalla.py:114).
# ---------------------------------------------------------------------------------------------------------------------------
def layer_train(x, W_u, W_i, a, edge, U, I, alpha, concat, p):
    hu = torch.mm(x[:U], W_u)
    hi = torch.mm(x[U:U + I], W_i)
    pair = torch.cat((hu[edge[0]], hi[edge[1]]), dim=1).t()
    e = torch.exp(-F.leaky_relu(a.mm(pair).squeeze(0), alpha))
    att = torch.sparse_coo_tensor(edge, e, (U, I))
    rs = torch.sparse.mm(att, torch.ones(I, 1, dtype=x.dtype))          # denominators: undropped e (SPUIGACF.py:367,370)
    cs = torch.sparse.mm(att.t(), torch.ones(U, 1, dtype=x.dtype))
    attd = torch.sparse_coo_tensor(edge, F.dropout(e, p, True), (U, I))   # numerators: dropout on e (:375)
    nu = torch.sparse.mm(attd, hi).div(rs)
    ni = torch.sparse.mm(attd.t(), hu).div(cs)
    ni = torch.where(torch.isnan(ni), torch.zeros_like(ni), ni)
    out = torch.cat([hu + nu, hi + ni], dim=0)
    return F.elu(out) if concat else out


def forward_train(params, x, edge, U, I, stages, p, alpha=0.2):
    for si, names in enumerate(stages):
        last = si == len(stages) - 1
        x = F.dropout(x, p, True)                                       # SPUIGACF.py:208,213
        outs = [layer_train(x, params[n + ".W_u"], params[n + ".W_i"], params[n + ".a"], edge, U, I, alpha, not last, p) for n in names]
        x = torch.cat(outs, dim=1)
        if last:
            x = F.elu(x)
    return x


def time_train_steps(params, edge, U, I, stages, B=2048, droprate=0.1, steps=2, seed=7, warm=False):
    """(seconds per train step, steps timed).  warm: one untimed step first (a step is ~10 s of CPU work on 10^5 edges: the
    one-time allocations of the first step are noise, so bench.py times from the first step on)."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=1e-3, weight_decay=1e-5)
    g = torch.Generator().manual_seed(seed)
    ub = torch.randint(0, U, (B,), generator=g)
    pb = torch.randint(0, I, (B,), generator=g)
    nb = torch.randint(0, I, (B,), generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        x = torch.cat([leaves["uEmbd.weight"], leaves["iEmbd.weight"]])
        f1 = forward_train(leaves, x, edge, U, I, stages, droprate)
        f2 = forward_train(leaves, x, edge, U, I, stages, droprate)
        pos = (f1[ub] * f1[U + pb]).sum(1)
        neg = (f2[ub] * f2[U + nb]).sum(1)
        loss = -F.logsigmoid(pos - neg).mean()
        loss.backward()
        opt.step()
        return float(loss.detach())
    if warm:
        step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    return (time.perf_counter() - t0) / steps, steps
