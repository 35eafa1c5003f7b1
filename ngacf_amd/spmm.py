"""Static-weight propagation over an N x N sparse matrix (GPLayer, SPUIGACF.py:178-185;
the Laplacians of data/loadGowalla.py:197-227)."""
import torch

from .plan import GraphPlan, _segments, validate_plan


def build_square_plan(rows, cols, vals, N, seg_len=256):
    """COO (rows, cols, vals) of an N x N matrix -> (plan, val) for ops.spmm_fwd.  Duplicates
    are summed like torch's coalesce()."""
    dev = rows.device
    key = rows.long() * N + cols.long()
    uniq, inv = torch.unique(key, return_inverse=True)
    val = torch.zeros(uniq.numel(), dtype=torch.float32, device=dev).index_add_(0, inv, vals.float())
    r, c = uniq // N, uniq % N
    rowptr = torch.zeros(N + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=N), 0)
    segs, split, nparts = _segments(rowptr, torch.arange(N, device=dev), seg_len)
    plan = GraphPlan(N, 0, int(uniq.numel()), rowptr, c.to(torch.int32).contiguous(), segs, split, nparts, seg_len)
    plan._all = (segs, split, nparts)
    validate_square(plan)
    return plan, val.contiguous()


def validate_square(p):
    if p.col.numel():
        assert int(p.col.min()) >= 0 and int(p.col.max()) < p.N
    if p.nseg:
        s = p.segs.long()
        assert int(s[:, 0].max()) < p.N and int(s[:, 2].max()) <= p.col.numel() and bool((s[:, 1] <= s[:, 2]).all())
        assert int(s[:, 3].max()) < max(p.nparts, 1)


class _SpMM(torch.autograd.Function):
    """out = (L + I) @ X through ngacf_spmm_fwd; the backward is the same product with the transposed matrix."""

    @staticmethod
    def forward(ctx, X, plan, val, plan_t, val_t, self_loop):
        from . import ops
        ctx.cfg = (plan_t, val_t, self_loop)
        return ops.spmm_fwd(plan, val, X.contiguous(), self_loop)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        plan_t, val_t, self_loop = ctx.cfg
        return ops.spmm_fwd(plan_t, val_t, g.contiguous(), self_loop), None, None, None, None, None


class StaticPropagation:
    """GPLayer.forward (SPUIGACF.py:178-185): features -> (L + I) @ features for a fixed sparse L given as COO."""

    def __init__(self, rows, cols, vals, N, seg_len=256):
        self.plan, self.val = build_square_plan(rows, cols, vals, N, seg_len)
        self.plan_t, self.val_t = build_square_plan(cols, rows, vals, N, seg_len)

    def __call__(self, X, self_loop=True):
        return _SpMM.apply(X, self.plan, self.val, self.plan_t, self.val_t, self_loop)


def plain_adj_coo(indices, U, I, rating=None):
    """`plain_adj` of data/loadGowalla.py:184-205: the (U+I) x (U+I) block matrix [[0, R], [R^T, 0]] with the ratings as
    values (ones when `rating` is None), as COO (rows, cols, vals); duplicates are summed by the plan builder like
    scipy's / torch's coalesce."""
    u, i = indices[0].long(), indices[1].long() + U
    v = torch.ones(u.numel(), dtype=torch.float32, device=indices.device) if rating is None else rating.float()
    return torch.cat([u, i]), torch.cat([i, u]), torch.cat([v, v])


def mean_adj_coo(indices, U, I):
    """`mean_adj` of data/loadGowalla.py:186-227: D^-1/2 A D^-1/2 over the (U+I)x(U+I) bipartite block matrix,
    as COO (rows, cols, vals) on the device of `indices` ([2,E], items not offset)."""
    u, i = indices[0].long(), indices[1].long() + U
    r, c = torch.cat([u, i]), torch.cat([i, u])
    deg = torch.bincount(r, minlength=U + I).float()
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))
    return r, c, dis[r] * dis[c]


def norm_adj_coo(indices, U, I):
    """`norm_adj` of data/loadGowalla.py:196-222: (D+I)^-1/2 (A + I) (D+I)^-1/2 -- the self loop is inside the
    normalisation.  COO (rows, cols, vals) over the (U+I) x (U+I) block matrix, diagonal entries last."""
    u, i = indices[0].long(), indices[1].long() + U
    n = torch.arange(U + I, device=indices.device)
    r, c = torch.cat([u, i, n]), torch.cat([i, u, n])
    deg = torch.bincount(r, minlength=U + I).float()           # >= 1 everywhere: the loop counts
    dis = deg.pow(-0.5)
    return r, c, dis[r] * dis[c]
