"""CPU oracle for the NGACF graph-attention propagation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ngacf_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker.

It is a numpy/scipy restatement (dtype follows the inputs, so float64 runs are
possible) of these reference functions -- citations are into /root/reference:

  layer_forward      graphattention/SPUIGACF.py:340-400  SpUIGraphAttentionLayer.forward
  layer_backward     autograd of the above
  spuigat_forward    graphattention/SPUIGACF.py:207-215  SpUIGAT.forward
                     graphattention/SPUIGACF.py:246-256  SPUIMultiGAT.forward
  spuigat_backward   autograd of the above
  model_scores       graphattention/SPUIGACF.py:41-52    SPUIGACF.forward
  bpr_loss           graphattention/BPRLoss.py:8-9
  spga_layer_forward graphattention/SPGA.py:381-418      SpGraphAttentionLayer.forward
  static_spmm        graphattention/SPUIGACF.py:178-185  GPLayer.forward
  precision_at_k ... graphattention/metrics.py:10-98
  ranklist / eval    train_eval_Gowalla.py:356-429

Pinning: tests/test_oracle_golden.py checks every function here against the
fixtures in tests/golden/, which oracle/gen_golden.py produced by importing and
running the reference itself in the build container.

Parameters travel as a flat ``dict`` keyed exactly like the reference
``state_dict`` (``uEmbd.weight``, ``gat.attention_0.W_u`` ... ) so fixtures,
oracle and product share one naming scheme.
"""
import heapq

import numpy as np
import scipy.sparse as sp


# --------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------
def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def elu_grad_from_pre(pre):
    return np.where(pre > 0, 1.0, np.exp(np.minimum(pre, 0))).astype(pre.dtype)


def leaky_relu(x, alpha):
    return np.where(x > 0, x, alpha * x)


# --------------------------------------------------------------------------
# one SpUIGraphAttentionLayer  (SPUIGACF.py:340-400)
# --------------------------------------------------------------------------
def layer_forward(X, W_u, W_i, a, edge, U, I, alpha=0.2, concat=True, edge_mask=None):
    """edge: int [2,E], row 0 = user id, row 1 = item id (NOT offset by U).

    edge_mask (optional, [E]) is the dropout multiplier applied to the numerators
    only (SPUIGACF.py:375); denominators keep the undropped e (:367,:370).
    Returns a dict with the output and every intermediate the backward needs.
    """
    dt = X.dtype
    eu, ei = edge[0], edge[1]
    k = W_u.shape[1]
    u_h = X[:U] @ W_u                                     # :356
    i_h = X[U:U + I] @ W_i                                # :357
    a_u, a_i = a[0, :k], a[0, k:]
    s = u_h[eu] @ a_u + i_h[ei] @ a_i                     # :359-361 (a·[hu‖hi])
    e = np.exp(-leaky_relu(s, alpha)).astype(dt)          # :361
    A = sp.csr_matrix((e, (eu, ei)), shape=(U, I))
    R = np.asarray(A.sum(axis=1)).reshape(U).astype(dt)   # :367 e_rowsum
    C = np.asarray(A.sum(axis=0)).reshape(I).astype(dt)   # :370 e_colsum
    if (R == 0).any():                                    # :368
        raise AssertionError("user with zero attention row-sum")
    me = e if edge_mask is None else (e * edge_mask).astype(dt)   # :375
    Am = sp.csr_matrix((me, (eu, ei)), shape=(U, I))
    agg_u = np.asarray(Am @ i_h) / R[:, None]             # :380-382
    with np.errstate(divide="ignore", invalid="ignore"):
        agg_i = np.asarray(Am.T @ u_h) / C[:, None]       # :387-388
    agg_i[np.isnan(agg_i)] = 0.0                          # :389
    pre = np.concatenate([u_h + agg_u, i_h + agg_i], 0).astype(dt)   # :383,:390,:393
    out = elu(pre).astype(dt) if concat else pre          # :395-400
    return dict(out=out, pre=pre, u_h=u_h, i_h=i_h, s=s, e=e, me=me, R=R, C=C,
                agg_u=agg_u.astype(dt), agg_i=agg_i.astype(dt))


def layer_backward(g_out, fwd, X, W_u, W_i, a, edge, U, I, alpha=0.2, concat=True, edge_mask=None):
    """Analytic gradient of layer_forward.  Returns
    dX, dW_u, dW_i, da."""
    dt = X.dtype
    eu, ei = edge[0], edge[1]
    k = W_u.shape[1]
    a_u, a_i = a[0, :k], a[0, k:]
    u_h, i_h, s, e, R, C = fwd["u_h"], fwd["i_h"], fwd["s"], fwd["e"], fwd["R"], fwd["C"]
    m = np.ones_like(e) if edge_mask is None else edge_mask.astype(dt)
    g = g_out * elu_grad_from_pre(fwd["pre"]) if concat else g_out
    gU, gI = g[:U], g[U:]
    live = C > 0                                          # isolated items: NaN->0 is a constant
    invR = 1.0 / R
    invC = np.where(live, 1.0 / np.where(live, C, 1.0), 0.0).astype(dt)
    gI_eff = np.where(live[:, None], gI, 0.0)
    dA = gU * invR[:, None]                               # dL/d(sum_i m e Hi)
    dB = gI_eff * invC[:, None]
    dR = -np.sum(gU * fwd["agg_u"], 1) * invR             # -(gU·A)/R² with agg=A/R
    dC = -np.sum(gI_eff * fwd["agg_i"], 1) * invC
    de = (m * np.sum(dA[eu] * i_h[ei], 1) + dR[eu]
          + m * np.sum(dB[ei] * u_h[eu], 1) + dC[ei])
    ds = -e * de * np.where(s > 0, 1.0, alpha)
    Mm = sp.csr_matrix((m * e, (eu, ei)), shape=(U, I))
    Ds = sp.csr_matrix((ds, (eu, ei)), shape=(U, I))
    dsu = np.asarray(Ds.sum(axis=1)).reshape(U)
    dsi = np.asarray(Ds.sum(axis=0)).reshape(I)
    du_h = gU + np.asarray(Mm @ dB) + dsu[:, None] * a_u[None, :]
    di_h = gI + np.asarray(Mm.T @ dA) + dsi[:, None] * a_i[None, :]
    da = np.concatenate([dsu @ u_h, dsi @ i_h])[None, :]
    dW_u = X[:U].T @ du_h
    dW_i = X[U:U + I].T @ di_h
    dX = np.concatenate([du_h @ W_u.T, di_h @ W_i.T], 0)
    return dX.astype(dt), dW_u.astype(dt), dW_i.astype(dt), da.astype(dt)


# --------------------------------------------------------------------------
# SpUIGAT / SPUIMultiGAT stacks  (SPUIGACF.py:207-215, :246-256)
# --------------------------------------------------------------------------
def stage_names(multi=False, nheads=8, hidden=None):
    """Per propagation stage: list of state_dict prefixes of its layers.  `hidden` = number of 8-head stages of
    the depth-generalised stack (ngacf_amd.StackedSpUIGAT, same naming as SPUIMultiGAT: attention{s}_{h})."""
    if hidden is not None:
        return [["gat.attention%d_%d" % (s + 1, h) for h in range(nheads)] for s in range(hidden)] + [["gat.out_att"]]
    if multi:
        return [["gat.attention1_%d" % h for h in range(nheads)],
                ["gat.attention2_%d" % h for h in range(nheads)],
                ["gat.out_att"]]
    return [["gat.attention_%d" % h for h in range(nheads)], ["gat.out_att"]]


def spuigat_forward(params, X, edge, U, I, multi=False, alpha=0.2, nheads=8,
                    feat_masks=None, edge_masks=None, keep=False, hidden=None):
    """Eval-mode stack when the mask arguments are None.  feat_masks[stage] is the
    multiplier of the F.dropout in front of that stage (SPUIGACF.py:208,213);
    edge_masks[stage][head] the per-layer edge dropout multiplier (:375).
    The last stage is the concat=False layer followed by F.elu (:214)."""
    stages = stage_names(multi, nheads, hidden)
    x = X
    tape = []
    for si, names in enumerate(stages):
        if feat_masks is not None and feat_masks[si] is not None:
            x = (x * feat_masks[si]).astype(X.dtype)
        last = si == len(stages) - 1
        outs, fw = [], []
        for hi, nm in enumerate(names):
            em = None if edge_masks is None else edge_masks[si][hi]
            f = layer_forward(x, params[nm + ".W_u"], params[nm + ".W_i"], params[nm + ".a"],
                              edge, U, I, alpha, concat=not last, edge_mask=em)
            outs.append(f["out"])
            fw.append(f)
        y = np.concatenate(outs, 1)
        if last:
            y = elu(y).astype(X.dtype)                    # :214 outer F.elu
        tape.append(dict(x=x, fw=fw, y=y))
        x = y
    return (x, tape) if keep else x


def spuigat_backward(g, tape, params, edge, U, I, multi=False, alpha=0.2, nheads=8,
                     feat_masks=None, edge_masks=None, hidden=None):
    """Gradient of spuigat_forward w.r.t. X and all layer parameters."""
    stages = stage_names(multi, nheads, hidden)
    grads = {}
    for si in reversed(range(len(stages))):
        names = stages[si]
        last = si == len(stages) - 1
        t = tape[si]
        if last:
            g = g * elu_grad_from_pre(t["fw"][0]["pre"])
        dx = np.zeros_like(t["x"])
        col = 0
        for hi, nm in enumerate(names):
            k = params[nm + ".W_u"].shape[1]
            em = None if edge_masks is None else edge_masks[si][hi]
            dX, dWu, dWi, da = layer_backward(g[:, col:col + k], t["fw"][hi], t["x"],
                                              params[nm + ".W_u"], params[nm + ".W_i"], params[nm + ".a"],
                                              edge, U, I, alpha, concat=not last, edge_mask=em)
            grads[nm + ".W_u"], grads[nm + ".W_i"], grads[nm + ".a"] = dWu, dWi, da
            dx += dX
            col += k
        if feat_masks is not None and feat_masks[si] is not None:
            dx = dx * feat_masks[si]
        g = dx
    return g, grads


# --------------------------------------------------------------------------
# SPUIGACF.forward + BPRLoss  (SPUIGACF.py:41-52, BPRLoss.py:8-9)
# --------------------------------------------------------------------------
def features_of(params):
    return np.concatenate([params["uEmbd.weight"], params["iEmbd.weight"]], 0)   # :30-39


def model_scores(params, userIdx, itemIdx, edge, multi=False, **kw):
    U, I = params["uEmbd.weight"].shape[0], params["iEmbd.weight"].shape[0]
    F = spuigat_forward(params, features_of(params), edge, U, I, multi=multi, **kw)
    return np.sum(F[userIdx] * F[itemIdx + U], 1)                                  # :49-52


def bpr_loss(pos, neg):
    x = pos - neg
    return -np.mean(np.log(1.0 / (1.0 + np.exp(-x))))                              # BPRLoss.py:9


def bpr_step_grads(params, userIdx, posIdx, negIdx, edge, multi=False, alpha=0.2, nheads=8, hidden=None):
    """loss and d(loss)/d(every parameter) for one reference train step without
    dropout: two propagations (train_eval_Gowalla.py:131-134) that are identical
    when droprate == 0, so their gradients add on one tape."""
    U, I = params["uEmbd.weight"].shape[0], params["iEmbd.weight"].shape[0]
    X = features_of(params)
    F, tape = spuigat_forward(params, X, edge, U, I, multi=multi, alpha=alpha, nheads=nheads, keep=True, hidden=hidden)
    fu, fp, fn = F[userIdx], F[posIdx + U], F[negIdx + U]
    pos, neg = np.sum(fu * fp, 1), np.sum(fu * fn, 1)
    loss = bpr_loss(pos, neg)
    B = len(userIdx)
    gx = -(1.0 / (1.0 + np.exp(pos - neg))) / B          # dloss/dpos ; dloss/dneg = -gx
    gF = np.zeros_like(F)
    np.add.at(gF, userIdx, gx[:, None] * (fp - fn))
    np.add.at(gF, posIdx + U, gx[:, None] * fu)
    np.add.at(gF, negIdx + U, -gx[:, None] * fu)
    dX, grads = spuigat_backward(gF, tape, params, edge, U, I, multi=multi, alpha=alpha, nheads=nheads, hidden=hidden)
    grads["uEmbd.weight"], grads["iEmbd.weight"] = dX[:U], dX[U:]
    return loss, pos, neg, grads


# --------------------------------------------------------------------------
# homogeneous sparse GAT layer (SPGA.py:381-418) and static-weight SpMM
# --------------------------------------------------------------------------
def spga_layer_forward(X, W, a, edge, alpha=0.2, concat=True):
    """edge [2,E] = adj.nonzero().t() over an N x N adjacency; no residual."""
    N, k = X.shape[0], W.shape[1]
    h = X @ W
    s = h[edge[0]] @ a[0, :k] + h[edge[1]] @ a[0, k:]
    e = np.exp(-leaky_relu(s, alpha)).astype(X.dtype)
    A = sp.csr_matrix((e, (edge[0], edge[1])), shape=(N, N))
    R = np.asarray(A.sum(axis=1)).reshape(N)
    hp = np.asarray(A @ h) / R[:, None]
    return (elu(hp) if concat else hp).astype(X.dtype)


def spgat_forward(p, X, edge, prefix="gat.", nheads=8, alpha=0.2):
    """SpGAT.forward in eval mode (SPGA.py:351-356): concat of the heads (ELU inside), then ELU(out_att)."""
    heads = [spga_layer_forward(X, p["%sattention_%d.W" % (prefix, h)], p["%sattention_%d.a" % (prefix, h)], edge, alpha, True)
             for h in range(nheads)]
    x = np.concatenate(heads, 1)
    return elu(spga_layer_forward(x, p[prefix + "out_att.W"], p[prefix + "out_att.a"], edge, alpha, False))


def spga_model_scores(kind, p, L, edge, u, it, U, I):
    """SPGACF / MultiLayerSPGA / SPGAMGP forward (SPGA.py:85-316), eval mode.  L: scipy N x N Laplacian; edge: [2,E'] the
    nonzeros of the mask SpGAT attends over."""
    F = np.concatenate([p["uEmbd.weight"], p["iEmbd.weight"]])
    L1 = L + sp.eye(U + I, format="csr")
    if kind == "spgacf":
        E = spgat_forward(p, F, edge)
    elif kind == "multi":
        final, s = [F], 0
        while "Affinelayers.%d.weight" % s in p:
            F = spgat_forward(p, F, edge, "GATlayers.%d." % s)
            F = np.maximum(L1 @ F @ p["Affinelayers.%d.weight" % s].T + p["Affinelayers.%d.bias" % s], 0)
            final.append(F)
            s += 1
        E = np.concatenate(final, 1)
    else:
        F = spgat_forward(p, F, edge)
        final, s = [F], 0
        while "Affinelayers.%d.weight" % s in p:
            F = np.maximum(L1 @ F @ p["Affinelayers.%d.weight" % s].T + p["Affinelayers.%d.bias" % s], 0)
            final.append(F)
            s += 1
        E = np.concatenate(final, 1)
    return np.sum(E[u] * E[it + U], 1)


def static_spmm(rows, cols, vals, X, add_self_loop=True):
    """GPLayer.forward (SPUIGACF.py:178-185): (L + I) @ X."""
    N = X.shape[0]
    L = sp.csr_matrix((vals, (rows, cols)), shape=(N, N))
    if add_self_loop:
        L = L + sp.eye(N, dtype=vals.dtype, format="csr")
    return np.asarray(L @ X).astype(X.dtype)


def mean_adj(edge, U, I, rating=None):
    """data/loadGowalla.py:186-227 adj_type == 'mean_adj': D^-1/2 A D^-1/2 over the
    N x N bipartite block matrix.  Returns COO (rows, cols, vals)."""
    N = U + I
    r = np.ones(edge.shape[1]) if rating is None else rating.astype(np.float64)
    A = sp.coo_matrix((np.concatenate([r, r]),
                       (np.concatenate([edge[0], edge[1] + U]), np.concatenate([edge[1] + U, edge[0]]))),
                      shape=(N, N)).tocsr()
    deg = np.asarray(A.sum(axis=1)).reshape(N)
    with np.errstate(divide="ignore"):
        dis = np.power(deg, -0.5)
    dis[np.isinf(dis)] = 0.0
    L = (sp.diags(dis) @ A @ sp.diags(dis)).tocoo()
    return L.row, L.col, L.data


def plain_adj(edge, U, I, rating=None):
    """data/loadGowalla.py:184-205 adj_type == 'plain_adj': vstack([R placed at column offset U, R^T])."""
    N = U + I
    r = np.ones(edge.shape[1]) if rating is None else np.asarray(rating, dtype=np.float64)
    A = sp.coo_matrix((np.concatenate([r, r]),
                       (np.concatenate([edge[0], edge[1] + U]), np.concatenate([edge[1] + U, edge[0]]))), shape=(N, N)).tocsr()
    A.sum_duplicates()
    A = A.tocoo()
    return A.row, A.col, A.data


def norm_adj(edge, U, I):
    """data/loadGowalla.py:196-222 adj_type == 'norm_adj': normalize_adj(adj + selfLoop)."""
    N = U + I
    r = np.ones(edge.shape[1])
    A = sp.coo_matrix((np.concatenate([r, r]),
                       (np.concatenate([edge[0], edge[1] + U]), np.concatenate([edge[1] + U, edge[0]]))),
                      shape=(N, N)).tocsr() + sp.eye(N, format="csr")
    deg = np.asarray(A.sum(axis=1)).reshape(N)
    dis = np.power(deg, -0.5)
    L = (sp.diags(dis) @ A @ sp.diags(dis)).tocoo()
    return L.row, L.col, L.data


# --------------------------------------------------------------------------
# dense multi-head attention (graphattention/modules.py:32-107)
# --------------------------------------------------------------------------
def multi_head_attention(p, q, k, v, heads, mask=None, prob_mask=None, prefix=""):
    """MultiHeadAttention.forward (modules.py:92-107) with `attention` (:32-46).  p: reference-keyed parameters
    ({prefix}q_linear.weight [dim,dim] ... out.bias); mask [Nq,Nk] (0 = masked -> score -1e9); prob_mask [heads,Nq,Nk]:
    dropout multipliers applied to the softmax probabilities (:41-42)."""
    lin = lambda x, n: x @ p[prefix + n + ".weight"].T + p[prefix + n + ".bias"]
    dim = q.shape[1]
    dk = dim // heads
    split = lambda x: x.reshape(-1, heads, dk).transpose(1, 0, 2)                    # :96-98  [h, N, d_k]
    Q, K, V = split(lin(q, "q_linear")), split(lin(k, "k_linear")), split(lin(v, "v_linear"))
    scores = Q @ K.transpose(0, 2, 1) / np.sqrt(dk)                                  # :33
    if mask is not None:
        scores = np.where(np.asarray(mask)[None] == 0, -1e9, scores)                 # :36
    scores = scores - scores.max(-1, keepdims=True)
    P = np.exp(scores)
    P = P / P.sum(-1, keepdims=True)                                                 # :41
    if prob_mask is not None:
        P = P * prob_mask
    concat = (P @ V).transpose(1, 0, 2).reshape(-1, dim)                             # :104
    return lin(concat, "out")


def att_layer(p, users, items, heads, prefix=""):
    """ATTLayer.forward (modules.py:13-18)."""
    u = multi_head_attention(p, items, users, users, heads, prefix=prefix + "att_item.")
    i = multi_head_attention(p, users, items, items, heads, prefix=prefix + "att_user.")
    return np.concatenate([u, i], 0)


def gacfv1_scores(p, L, u, it, U, I, heads=8, variant=1):
    """GACFV1.forward (GACFmodel1.py:87-115, GALayer :29-40) and, by `variant`, GACFV3 / V4 / V5 (GACFmodel3.py:29-40,
    GACFmodel4.py:30-42, GACFmodel5.py:30-42: same parameters, other inputs to the two Laplacian products);
    L: scipy sparse N x N Laplacian."""
    F = np.concatenate([p["gacf.uEmbd.weight"], p["gacf.iEmbd.weight"]])
    N = U + I
    L1 = L + sp.eye(N, format="csr")
    final = [F]
    s = 0
    while "gacf.GALayers.%d.affine1.weight" % s in p:
        pre = "gacf.GALayers.%d." % s
        A = att_layer(p, F[:U], F[U:], heads, pre + "att.")
        first = A if variant in (1, 3) else F
        second = A * A if variant in (1, 4) else A
        f1 = np.maximum(L1 @ first @ p[pre + "affine1.weight"].T + p[pre + "affine1.bias"], 0)
        f2 = np.maximum(L @ second @ p[pre + "affine2.weight"].T + p[pre + "affine2.bias"], 0)
        F = f1 + f2
        final.append(F)
        s += 1
    E = np.concatenate(final, 1)
    return np.sum(E[u] * E[it + U], 1)


def rating_laplacian(u, it, rating, U, I):
    """GACFV6.buildLaplacianMat (GACFmodel6.py:139-161): D^-1/2 A D^-1/2 over the (U+I) x (U+I) block matrix whose entries are the
    RATINGS (duplicates summed by coo_matrix) and whose degrees count the non-zero entries of a row ((A > 0).sum)."""
    N = U + I
    R = sp.coo_matrix((np.asarray(rating, dtype=np.float64), (u, it)), shape=(U, I)).tocsr()
    A = sp.bmat([[None, R], [R.T, None]], format="csr")
    deg = np.asarray((A > 0).sum(1)).ravel().astype(np.float64)
    with np.errstate(divide="ignore"):
        d = np.power(deg, -0.5)
    D = sp.diags(d)
    return (D @ A @ D).tocsr()


def gacf_mlp_scores(p, L, u, it, U, I, kind, heads=8):
    """GACFV6.forward (GACFmodel6.py:96-122, GALayer :29-43) / GACFV3x.forward (GACFmodel3.py:170-194, GALayer :29-40):
    stage outputs concatenated, the user row and the item row side by side through transForm1..3 (ReLU, ReLU, none)."""
    F = np.concatenate([p["uEmbd.weight"], p["iEmbd.weight"]])
    N = U + I
    L1 = L + sp.eye(N, format="csr")
    final, s = [F], 0
    while "GALayers.%d.affine1.weight" % s in p:
        pre = "GALayers.%d." % s
        A = att_layer(p, F[:U], F[U:], heads, pre + "att.")
        if kind == "v6":
            F = (L1 @ F @ p[pre + "affine1.weight"].T + p[pre + "affine1.bias"]) + (A @ p[pre + "affine2.weight"].T + p[pre + "affine2.bias"])
        else:
            f1 = np.maximum(L1 @ A @ p[pre + "affine1.weight"].T + p[pre + "affine1.bias"], 0)
            f2 = np.maximum(L @ A @ p[pre + "affine2.weight"].T + p[pre + "affine2.bias"], 0)
            F = np.maximum(f1 + f2, 0)
        final.append(F)
        s += 1
    E = np.concatenate(final, 1)
    x = np.concatenate([E[u], E[it + U]], 1)
    x = np.maximum(x @ p["transForm1.weight"].T + p["transForm1.bias"], 0)
    x = np.maximum(x @ p["transForm2.weight"].T + p["transForm2.bias"], 0)
    return (x @ p["transForm3.weight"].T + p["transForm3.bias"]).ravel()


def gacfv2_scores(p, L, u, it, U, I, heads=8):
    """GACFV2.forward (GACFmodel2.py:88-135): ATTLayer -> (L + I) -> Linear -> ReLU per layer."""
    F = np.concatenate([p["gacf.uEmbd.weight"], p["gacf.iEmbd.weight"]])
    L1 = L + sp.eye(U + I, format="csr")
    final, s = [F], 0
    while "gacf.Affinelayers.%d.weight" % s in p:
        A = att_layer(p, F[:U], F[U:], heads, "gacf.ATTlayers.%d." % s)
        F = np.maximum(L1 @ A @ p["gacf.Affinelayers.%d.weight" % s].T + p["gacf.Affinelayers.%d.bias" % s], 0)
        final.append(F)
        s += 1
    E = np.concatenate(final, 1)
    return np.sum(E[u] * E[it + U], 1)


def gacfmask_scores(p, L, mask, u, it, U, I, heads=8):
    """GACFMask.forward (GACFMask.py:89-136)."""
    F = np.concatenate([p["gacf.uEmbd.weight"], p["gacf.iEmbd.weight"]])
    N = U + I
    L1 = L + sp.eye(N, format="csr")
    final = [F]
    s = 0
    while "gacf.Affinelayers.%d.weight" % s in p:
        A = multi_head_attention(p, F, F, F, heads, mask, prefix="gacf.ATTlayers.%d.att." % s)
        F = np.maximum(L1 @ A @ p["gacf.Affinelayers.%d.weight" % s].T + p["gacf.Affinelayers.%d.bias" % s], 0)
        final.append(F)
        s += 1
    E = np.concatenate(final, 1)
    return np.sum(E[u] * E[it + U], 1)


# --------------------------------------------------------------------------
# metrics (graphattention/metrics.py) and eval ranking (train_eval_Gowalla.py)
# --------------------------------------------------------------------------
def precision_at_k(r, k):                                 # metrics.py:10-21
    return float(np.mean(np.asarray(r)[:k]))


def dcg_at_k(r, k):                                       # metrics.py:45-60, method=1
    r = np.asarray(r, dtype=np.float64)[:k]
    if r.size:
        return float(np.sum(r / np.log2(np.arange(2, r.size + 2))))
    return 0.0


def ndcg_at_k(r, k):                                      # metrics.py:63-73
    # the ideal DCG is built from the *retrieved* list itself -- the reference's quirk
    best = dcg_at_k(sorted(r, reverse=True), k)
    if not best:
        return 0.0
    return dcg_at_k(r, k) / best


def recall_at_k(r, k, all_pos_num):                       # metrics.py:76-78
    return float(np.sum(np.asarray(r, dtype=np.float64)[:k]) / all_pos_num)


def hit_at_k(r, k):                                       # metrics.py:81-86
    return 1.0 if np.sum(np.asarray(r)[:k]) > 0 else 0.0


def ranklist(user_pos_test, test_items, rating, Ks=(1, 5, 10, 20)):
    """train_eval_Gowalla.py:370-385 ranklist_by_heapq: top-K over the candidate
    items in iteration order, ties resolved as heapq.nlargest does."""
    score = {i: rating[i] for i in test_items}
    top = heapq.nlargest(max(Ks), score, key=score.get)
    pos = set(user_pos_test)
    return [1 if i in pos else 0 for i in top]


def performance(user_pos_test, r, Ks=(1, 5, 10, 20)):     # train_eval_Gowalla.py:419-429
    return dict(precision=np.array([precision_at_k(r, K) for K in Ks]),
                recall=np.array([recall_at_k(r, K, len(user_pos_test)) for K in Ks]),
                ndcg=np.array([ndcg_at_k(r, K) for K in Ks]),
                hit_ratio=np.array([hit_at_k(r, K) for K in Ks]))


def eval_all_items(score_rows, users, train_pos, test_pos, item_pool, n_norm, Ks=(1, 5, 10, 20)):
    """eval_neg_all on a ready score matrix (train_eval_Gowalla.py:274-354).

    score_rows[j] are the all-item scores of users[j]; candidates are the items of
    item_pool not in the user's train positives (loadGowalla.py:63-67), hits are the
    test positives; sums are divided by n_norm = number of *train* users (:283)."""
    res = dict(precision=np.zeros(len(Ks)), recall=np.zeros(len(Ks)),
               ndcg=np.zeros(len(Ks)), hit_ratio=np.zeros(len(Ks)))
    for row, u in zip(score_rows, users):
        cand = item_pool - train_pos[u]
        r = ranklist(test_pos[u], cand, row, Ks)
        p = performance(test_pos[u], r, Ks)
        for key in res:
            res[key] += p[key] / n_norm
    return res


# ---------------------------------------------------------------------------------------------------------
# --train_mode NegSampling / --eval_mode SampledNeg on a score table (pinned by fixture g18: the reference's own loops)
# ---------------------------------------------------------------------------------------------------------
def neg_sample_epoch_loss(S, users, pos, negs, batch_size):
    """train_eval_Gowalla.py:36-88 with predictions = S[u, i]: per batch BCEWithLogitsLoss (mean over the rows' 5 logits: the row's
    item with label 1, its 4 negatives with label 0), summed over the batches and divided by the number of rows (:88)."""
    S = np.asarray(S, dtype=np.float64)
    n, total = len(users), 0.0
    for b in range(n // batch_size + 1):
        sl = slice(b * batch_size, (b + 1) * batch_size)
        if len(users[sl]) == 0:
            continue
        items = np.concatenate([np.asarray(pos[sl])[:, None], np.asarray(negs[sl])], 1)
        x = S[np.asarray(users[sl])[:, None], items]
        y = np.zeros_like(x)
        y[:, 0] = 1.0
        total += float(np.mean(np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))))       # BCEWithLogitsLoss, stable form
    return total / n


def sampled_hr_ndcg(S, users, pos, negs, top_k=10):
    """train_eval_Gowalla.py:193-259 + graphattention/evaluation.py:4-13 with predictions = S[u, i]: the row's item among the
    top_k of its 100 scores -> hit, NDCG = 1 / log2(index + 2); means over the rows."""
    S = np.asarray(S, dtype=np.float64)
    hr, nd = [], []
    for u, p, ng in zip(users, pos, negs):
        items = np.concatenate([[p], ng])
        sc = S[u, items]
        top = items[np.argsort(-sc, kind="stable")[:top_k]]          # torch.topk's order, exact ties aside
        if p in top:
            idx = int(np.nonzero(top == p)[0][0])
            hr.append(1.0)
            nd.append(1.0 / np.log2(idx + 2))
        else:
            hr.append(0.0)
            nd.append(0.0)
    return float(np.mean(hr)), float(np.mean(nd))
