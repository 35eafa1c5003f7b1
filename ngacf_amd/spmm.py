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
