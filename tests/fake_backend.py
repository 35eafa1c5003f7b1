"""TEST-ONLY stand-in for ngacf_amd.ops on CPU tensors (plain torch), used to run the node-partitioned
choreography of ngacf_amd/dist.py under gloo where no GPU exists.  It mirrors the semantics of the C-ABI
entry points (row windows, global-row tables, two-step backward), not their speed.  The product never
imports this file."""
import torch
import torch.nn.functional as F


def _types(U, N, dev):
    return (torch.arange(N, device=dev) >= U).long()


def _win_rows(U, win, dev):
    u_lo, u_n, i_lo, i_n = win
    return torch.cat([torch.arange(u_lo, u_lo + u_n, device=dev), torch.arange(U + i_lo, U + i_lo + i_n, device=dev)])


def transform_fwd_rows(X, W, avec, H, S, U, I, heads, win, x_local, feat_drop_p=0.0, seed=0):
    assert feat_drop_p == 0.0
    rows = _win_rows(U, win, H.device)
    Xw = X if x_local else X[rows]
    t = (rows >= U).long()
    Hw = torch.einsum("nk,nkd->nd", Xw.float(), W[t])
    H[rows] = Hw.to(H.dtype)
    hw = H.shape[1] // heads
    S[rows] = (H[rows].float() * avec[t]).view(-1, heads, hw).sum(2)


def _half_ranges(plan, half):
    if half is None:
        return plan.row_ranges
    return [(lo, hi) for lo, hi in plan.row_ranges if (lo < plan.U) == (half == "user") and hi > lo]


def _edges(plan, half=None):
    dst, src = [], []
    for lo, hi in _half_ranges(plan, half):
        e0, e1 = int(plan.rowptr[lo]), int(plan.rowptr[hi])
        deg = plan.rowptr[lo + 1:hi + 1] - plan.rowptr[lo:hi]
        dst.append(torch.repeat_interleave(torch.arange(lo, hi), deg))
        src.append(plan.col[e0:e1].long())
    return torch.cat(dst), torch.cat(src)


def prop_fwd(plan, H, S_dst, S_src, avec, heads, alpha=0.2, elu=True, residual=True, score_from_row=None,
             drop_p=0.0, seed=0, save=False, out=None, half=None, agg=None, R=None):
    assert drop_p == 0.0
    N, d = H.shape
    hw = d // heads
    dst, src = _edges(plan, half)
    agg_out, R_out = agg, R
    assert score_from_row, "dist.py never gathers S: source scores come from the gathered rows"
    t = _types(plan.U, N, H.device)
    s_src = (H[src].float() * avec[t[src]]).view(-1, heads, hw).sum(2)
    s = S_dst[dst] + s_src
    e = torch.exp(-F.leaky_relu(s, alpha))                                             # [E', heads]
    en = torch.where(s > 0, torch.zeros_like(e), e)                                    # the edges with a non-positive score
    R = torch.zeros(N, heads).index_add_(0, dst, e)
    Rn = torch.zeros(N, heads).index_add_(0, dst, en)
    num = torch.zeros(N, d).index_add_(0, dst, e.repeat_interleave(hw, 1) * H[src].float())
    numn = torch.zeros(N, d).index_add_(0, dst, en.repeat_interleave(hw, 1) * H[src].float())
    Rw = R.repeat_interleave(hw, 1)
    agg = torch.where(Rw > 0, num / Rw.clamp_min(1e-38), torch.zeros_like(num))
    aggn = torch.where(Rw > 0, numn / Rw.clamp_min(1e-38), torch.zeros_like(numn))
    pre = (H.float() + agg) if residual else agg
    o = F.elu(pre) if elu else pre
    res = torch.zeros(N, d, dtype=H.dtype) if out is None else out
    rows = torch.cat([torch.arange(lo, hi) for lo, hi in _half_ranges(plan, half)])
    res[rows] = o[rows].to(H.dtype)
    if save:                                              # two planes each (ops.prop_fwd); half launches fill the caller's tables
        if agg_out is None:
            agg_out, R_out = torch.zeros(2, N, d, dtype=H.dtype), torch.zeros(2, N, heads)
        agg_out[0, rows], agg_out[1, rows] = agg[rows].to(H.dtype), aggn[rows].to(H.dtype)
        R_out[0, rows], R_out[1, rows] = R[rows], Rn[rows]
        return res, agg_out, R_out
    return res


def prop_bwd_node(out, agg, R, gout, S_src, G, Gn, ST, dS, U, heads, win, gout_local, elu=True, alpha=0.2):
    rows = _win_rows(U, win, out.device)
    d = out.shape[1]
    hw = d // heads
    g = (gout if gout_local else gout[rows]).float().clone()
    if elu:
        o = out[rows].float()
        g = g * torch.where(o > 0, torch.ones_like(o), o + 1.0)
    r, rn = R[0, rows], R[1, rows]
    inv = torch.where(r > 0, 1.0 / r.clamp_min(1e-38), torch.zeros_like(r)).repeat_interleave(hw, 1)
    gn = (g * inv).to(Gn.dtype)
    G[rows] = g
    Gn[rows] = gn
    t = (gn.float() * agg[0, rows].float()).view(-1, heads, hw).sum(2)
    tn = (gn.float() * agg[1, rows].float()).view(-1, heads, hw).sum(2)
    ST.view(-1, heads, 2)[rows] = torch.stack([S_src[rows], t], dim=2)           # pairs (S, t)
    dS[rows] = (1.0 - alpha) * (r * tn - rn * t)                                 # P1: the rows' own half of dS (include/ngacf.h)


def prop_bwd_edges(plan, H, S_dst, avec, G, Gn, ST, heads, alpha=0.2, residual=True, score_from_row=True,
                   drop_p=0.0, seed=0, half=None, dH=None, dS=None):
    assert drop_p == 0.0 and dS is not None
    N, d = H.shape
    hw = d // heads
    dst, src = _edges(plan, half)
    Hf, Gf = H.float(), Gn.float()
    st = ST.view(-1, heads, 2)
    s = S_dst[dst] + st[src, :, 0]
    e = torch.exp(-F.leaky_relu(s, alpha))
    dot = (Gf[src] * Hf[dst]).view(-1, heads, hw).sum(2)
    de = dot - st[src, :, 1]
    ds = -e * de * torch.where(s > 0, torch.ones_like(s), torch.full_like(s, alpha))
    dS_full = dS + torch.zeros(N, heads).index_add_(0, dst, ds)
    acc = torch.zeros(N, d).index_add_(0, dst, e.repeat_interleave(hw, 1) * Gf[src])
    t = _types(plan.U, N, H.device)
    dHf = (G if residual else torch.zeros_like(G)) + acc + dS_full.repeat_interleave(hw, 1) * avec[t]
    keep = torch.zeros(N, dtype=torch.bool)
    for lo, hi in _half_ranges(plan, half):
        keep[lo:hi] = True
    if dH is None:
        dH = torch.zeros(N, d, dtype=H.dtype)
    dH[keep] = dHf[keep].to(dH.dtype)
    dS[keep] = dS_full[keep]
    return dH, dS


def transform_bwd_rows(X, W, H, dH, dS, U, I, heads, win, x_local, feat_drop_p=0.0, seed=0):
    assert feat_drop_p == 0.0
    rows = _win_rows(U, win, H.device)
    Xw = (X if x_local else X[rows]).float()
    t = (rows >= U).long()
    g = dH[rows]
    dX = torch.einsum("nd,nkd->nk", g, W[t])
    hw = H.shape[1] // heads
    dW = torch.zeros_like(W)
    da = torch.zeros(2, H.shape[1])
    for ty in (0, 1):
        m = t == ty
        dW[ty] = Xw[m].t() @ g[m]
        da[ty] = (dS[rows][m].repeat_interleave(hw, 1) * H[rows][m].float()).sum(0)
    if not x_local:
        full = torch.zeros(U + I, X.shape[1])
        full[rows] = dX
        dX = full
    return dX, dW, da


def score(Fm, u, i, U):
    return (Fm[u] * Fm[i + U]).sum(1)


def eval_topk(Fm, users, U, I, pool_mask, train_ptr, train_items, K=20):
    """Stand-in for ops.eval_topk (same signature, CPU tensors): candidates = pool minus the user's train positives, best first,
    ties to the lower item id (a stable sort of the negated scores)."""
    sc = Fm[users].double() @ Fm[U:U + I].double().t()
    sc[:, pool_mask == 0] = -float("inf")
    for b, u in enumerate(users.tolist()):
        sc[b, train_items[int(train_ptr[u]):int(train_ptr[u + 1])].long()] = -float("inf")
    order = torch.sort(-sc, dim=1, stable=True).indices[:, :K]
    top_s = torch.gather(sc, 1, order)
    top_i = torch.where(torch.isinf(top_s), torch.full_like(order, -1), order)
    return top_i.to(torch.int32), top_s.float()
