"""Drop-in surface of the reference's dense multi-head attention (graphattention/modules.py) and of the two model
families built on it, backed by the HIP flash-style kernel ngacf_mha_fwd / ngacf_mha_bwd (csrc/mha.hip):

  attention(q, k, v, d_k, mask=None, dropout=None)                      modules.py:32-46
  MultiHeadAttention(heads, dim, dropout=0.1).forward(q, k, v, mask)    modules.py:62-107   (same parameters / keys)
  ATTLayer(dim, heads, dropout).forward(userFeatures, itemFeatures)     modules.py:7-18
  ATTLayer_mask(dim, heads, dropout).forward(features, mask)            modules.py:20-30
  GALayer / GACFV1_layer / GACFV1 (and GACFV3, GACFV4, GACFV5)          GACFmodel{1,3,4,5}.py
  GACFV2                                                                GACFmodel2.py:34-135
  GACFV3x, GACFV6 (three-layer MLP head on cat(user row, item row))     GACFmodel3.py:114-200, GACFmodel6.py:12-161
  GPLayer / GACFV2_layer / GACFMask                                     GACFMask.py:22-136

What differs from the reference is how the attention core runs: the [heads, Nq, Nk] score and probability tensors are
never materialised (the reference holds both, plus autograd's copies), the q/k/v projections are read in place (no
view/transpose/contiguous), the Laplacian products go through the static-weight traversal (ngacf_spmm_fwd), and the
per-call Python index lists / sparse `L + I` additions of the reference (GACFmodel1.py:33-34,71-76) are gone.
O(Nq*Nk) work remains: this path is for ml100k / ml1m sized graphs, as in the reference.
"""
import torch
import torch.nn as nn

from . import ops
from .SPUIGACF import _fresh_seed


class _MHACore(torch.autograd.Function):
    """softmax(q k^T / sqrt(d_k) [masked]) -> dropout -> @ v for all heads, on [N, dim] projections."""

    @staticmethod
    def forward(ctx, Q, K, V, mask_u8, heads, drop_p, seed):
        Q, K, V = Q.contiguous().float(), K.contiguous().float(), V.contiguous().float()
        O, LSE = ops.mha_fwd(Q, K, V, heads, mask_u8, drop_p, seed, None)
        ctx.save_for_backward(Q, K, V, O, LSE)
        ctx.cfg = (mask_u8, heads, drop_p, seed)
        return O

    @staticmethod
    def backward(ctx, dO):
        Q, K, V, O, LSE = ctx.saved_tensors
        mask_u8, heads, drop_p, seed = ctx.cfg
        dQ, dK, dV = ops.mha_bwd(Q, K, V, O, LSE, dO.contiguous().float(), heads, mask_u8, drop_p, seed, None)
        return dQ, dK, dV, None, None, None, None


class _MaskCache:
    """`mask == 0` test of modules.py:36 precomputed once per mask tensor as packed bits (ops.pack_mask_bits: one 32-bit word
    per query and key tile; the entry holds the keyed tensor, see SPUIGACF._PlanCache)."""

    def __init__(self):
        self._hit = None

    def get(self, mask):
        key = (mask.data_ptr(), tuple(mask.shape), str(mask.device), mask._version)
        if self._hit is None or self._hit[0] != key:
            self._hit = (key, mask, ops.pack_mask_bits(mask))
        return self._hit[2]


def attention(q, k, v, d_k, mask=None, dropout=None):
    """modules.py:32-46 with the reference's layout: q [h, Nq, d_k], k, v [h, Nk, d_k] -> [h, Nq, d_k].
    `dropout`: an nn.Dropout (its p and training flag are honoured; masks come from the counter hash)."""
    h, Nq, dk = q.shape
    assert dk == d_k
    flat = lambda t: t.transpose(0, 1).reshape(t.shape[1], h * dk)
    p = dropout.p if (dropout is not None and dropout.training) else 0.0
    m8 = None if mask is None else ops.pack_mask_bits(mask)
    O = _MHACore.apply(flat(q), flat(k), flat(v), m8, h, p, _fresh_seed() if p > 0 else 0)
    return O.view(Nq, h, dk).transpose(0, 1)


class MultiHeadAttention(nn.Module):
    def __init__(self, heads, dim, dropout=0.1):
        super().__init__()
        self.att = None
        self.dim, self.d_k, self.h = dim, dim // heads, heads
        self.q_linear = nn.Linear(dim, dim)
        self.v_linear = nn.Linear(dim, dim)
        self.k_linear = nn.Linear(dim, dim)
        self.dropout = nn.Dropout(dropout)
        self.out = nn.Linear(dim, dim)
        for lin in (self.q_linear, self.v_linear, self.k_linear, self.out):       # modules.py:80-91
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.zero_()
        self._masks = _MaskCache()

    def forward(self, q, k, v, mask=None):
        # the projections are plain library GEMMs; their [N, dim] outputs ARE the per-head layout the kernel reads
        Q, K, V = self.q_linear(q), self.k_linear(k), self.v_linear(v)
        p = self.dropout.p if self.training else 0.0
        m8 = None if mask is None else self._masks.get(mask)
        concat = _MHACore.apply(Q, K, V, m8, self.h, p, _fresh_seed() if p > 0 else 0)
        return self.out(concat)                                                   # :104-106


class ATTLayer(nn.Module):
    def __init__(self, dim, heads, dropout):
        super().__init__()
        self.att_user = MultiHeadAttention(heads, dim, dropout)
        self.att_item = MultiHeadAttention(heads, dim, dropout)

    def forward(self, userFeatures, itemFeatures):
        u = self.att_item(itemFeatures, userFeatures, userFeatures)               # :14 (items query the users ...)
        i = self.att_user(userFeatures, itemFeatures, itemFeatures)               # (... and the result is named user side)
        return torch.cat([u, i], dim=0)


class ATTLayer_mask(nn.Module):
    def __init__(self, dim, heads, dropout):
        super().__init__()
        self.att = MultiHeadAttention(heads, dim, dropout)

    def forward(self, features, mask):
        return self.att(features, features, features, mask)


class _Laplacian:
    """The fixed N x N Laplacian of a model as a static-weight propagation (built on first use, per device)."""

    def __init__(self, adj, N):
        self._adj, self._N, self._prop = adj, N, None

    def get(self, dev):
        if self._prop is None or self._prop.plan.device != dev:
            from .spmm import StaticPropagation
            if isinstance(self._adj, (tuple, list)):
                r, c, v = self._adj
            else:
                a = self._adj.coalesce()
                r, c, v = a.indices()[0], a.indices()[1], a.values()
            self._prop = StaticPropagation(r.to(dev), c.to(dev), v.to(dev).float(), self._N)
        return self._prop


def _check_width(d):
    if d not in (64, 128, 256):
        raise ValueError("layer widths must be 64, 128 or 256 on the accelerated path (got %d)" % d)


class GALayer(nn.Module):
    """The attention + graph-convolution layer of GACFmodel{1,3,4,5}.py (`GALayer`, same parameters in all four; the
    variants differ in what feeds the two Laplacian products, graphattention/README.md:3-20):

        variant 1 (GACFmodel1.py:29-40)   f1 = ReLU(aff1((L+I) A)),  f2 = ReLU(aff2(L (A * A)))      A = ATTLayer(users, items)
        variant 3 (GACFmodel3.py:29-40)   f1 = ReLU(aff1((L+I) A)),  f2 = ReLU(aff2(L A))
        variant 4 (GACFmodel4.py:30-42)   f1 = ReLU(aff1((L+I) F)),  f2 = ReLU(aff2(L (A * A)))      F = cat(users, items)
        variant 5 (GACFmodel5.py:30-42)   f1 = ReLU(aff1((L+I) F)),  f2 = ReLU(aff2(L A))
    """
    variant = 1

    def __init__(self, input_dim, out_dim, heads, dropout, variant=None):
        super().__init__()
        _check_width(input_dim)
        if variant is not None:
            self.variant = variant
        if self.variant not in (1, 3, 4, 5):
            raise ValueError("GALayer variant must be 1, 3, 4 or 5")
        self.att = ATTLayer(input_dim, heads, dropout)
        self.affine1 = nn.Linear(input_dim, out_dim)
        self.affine2 = nn.Linear(input_dim, out_dim)
        for lin in (self.affine1, self.affine2):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.zero_()

    def forward(self, userFeatures, itemFeatures, propagation):
        A = self.att(userFeatures, itemFeatures)
        first = A if self.variant in (1, 3) else torch.cat([userFeatures, itemFeatures], dim=0)
        second = A * A if self.variant in (1, 4) else A
        f1 = torch.relu(self.affine1(propagation(first, self_loop=True)))                       # (L + I) @ .
        f2 = torch.relu(self.affine2(propagation(second, self_loop=False)))                     # L @ .
        return f1 + f2


class GACFV1_layer(nn.Module):
    variant = 1

    def __init__(self, userNum, itemNum, adj, embedSize=256, layers=(256, 128, 64), droprate=0.2, useCuda=True):
        super().__init__()
        self.useCuda, self.userNum, self.itemNum, self.droprate = useCuda, userNum, itemNum, droprate
        self.uEmbd = nn.Embedding(userNum, embedSize)
        self.iEmbd = nn.Embedding(itemNum, embedSize)
        self.GALayers = nn.ModuleList(GALayer(a, b, 8, droprate, self.variant) for a, b in zip(layers[:-1], layers[1:]))
        self._lap = _Laplacian(adj, userNum + itemNum)
        nn.init.normal_(self.uEmbd.weight, std=0.01)
        nn.init.normal_(self.iEmbd.weight, std=0.01)

    def forward(self, userIdx, itemIdx):
        U = self.userNum
        features = torch.cat([self.uEmbd.weight, self.iEmbd.weight], dim=0)
        prop = self._lap.get(features.device)
        final = [features]
        for ga in self.GALayers:
            features = ga(features[:U], features[U:], prop)
            final.append(features)
        return torch.cat(final, dim=1)


class GACFV3_layer(GACFV1_layer):
    variant = 3


class GACFV4_layer(GACFV1_layer):
    variant = 4


class GACFV5_layer(GACFV1_layer):
    variant = 5


class GACFV1(nn.Module):
    """GACFmodel1.py:87-115 (and, through `_layer`, GACFV3 GACFmodel3.py:99-112, GACFV4 GACFmodel4.py:101-115, GACFV5
    GACFmodel5.py:102-117): concatenated stage outputs, dot-product score."""
    _layer = GACFV1_layer

    def __init__(self, userNum, itemNum, adj, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self.userNum = userNum
        self.gacf = self._layer(userNum, itemNum, adj, embedSize, layers, droprate)

    def forward(self, userIdx, itemIdx):
        e = self.gacf(userIdx, itemIdx)
        return torch.sum(e[userIdx] * e[itemIdx + self.userNum], dim=1)


class GACFV3(GACFV1):
    _layer = GACFV3_layer


class GACFV4(GACFV1):
    _layer = GACFV4_layer


class GACFV5(GACFV1):
    _layer = GACFV5_layer


class GALayerV6(nn.Module):
    """GACFmodel6.py:12-43: affine1((L + I) @ cat(users, items)) + affine2(ATTLayer(users, items)) -- the attention output goes
    straight into the second Linear (no Laplacian product, no ReLU inside the layer)."""

    def __init__(self, input_dim, out_dim, heads, dropout):
        super().__init__()
        _check_width(input_dim)
        self.att = ATTLayer(input_dim, heads, dropout)
        self.affine1 = nn.Linear(input_dim, out_dim)
        self.affine2 = nn.Linear(input_dim, out_dim)
        for lin in (self.affine1, self.affine2):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.zero_()

    def forward(self, userFeatures, itemFeatures, propagation):
        f1 = self.affine1(propagation(torch.cat([userFeatures, itemFeatures], dim=0), self_loop=True))
        return f1 + self.affine2(self.att(userFeatures, itemFeatures))


def rating_laplacian_coo(user_ids, item_ids, rating, U, I):
    """GACFV6.buildLaplacianMat (GACFmodel6.py:139-161) as COO (rows, cols, vals): D^-1/2 A D^-1/2 over the (U+I) x (U+I) block
    matrix whose entries are the RATINGS (duplicate pairs summed, as coo_matrix does) and whose degrees count a row's non-zero
    entries ((A > 0).sum)."""
    u, i = torch.as_tensor(user_ids).long(), torch.as_tensor(item_ids).long()
    key, inv = torch.unique(u * I + i, return_inverse=True)
    val = torch.zeros(key.numel(), dtype=torch.float64).index_add_(0, inv, torch.as_tensor(rating, dtype=torch.float64))
    uu, ii = key // I, key % I + U
    r, c, v = torch.cat([uu, ii]), torch.cat([ii, uu]), torch.cat([val, val])
    deg = torch.bincount(r[v > 0], minlength=U + I).double()
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.full_like(deg, float("inf")))
    return r, c, (dis[r] * v * dis[c]).float()


class _GACFMlpHead(nn.Module):
    """Shared body of GACFV6 and GACFV3x: embeddings -> GALayers (stage outputs concatenated) -> cat(user row, item row) ->
    transForm1 (ReLU) -> transForm2 (ReLU) -> transForm3 -> flatten (GACFmodel6.py:96-122, GACFmodel3.py:170-194)."""

    def __init__(self, userNum, itemNum, adj, embedSize, layers, droprate, useCuda, make_layer, relu_outside):
        super().__init__()
        layers = list(layers)
        self.useCuda, self.userNum, self.itemNum, self.droprate = useCuda, userNum, itemNum, droprate
        self.uEmbd = nn.Embedding(userNum, embedSize)
        self.iEmbd = nn.Embedding(itemNum, embedSize)
        self.GALayers = nn.ModuleList()
        self.leakyRelu = nn.LeakyReLU()                                   # (held, never applied: GACFmodel6.py:62)
        self.transForm1 = nn.Linear(sum(layers) * 2, 64)
        self.transForm2 = nn.Linear(64, 32)
        self.transForm3 = nn.Linear(32, 1)
        for a, b in zip(layers[:-1], layers[1:]):
            self.GALayers.append(make_layer(a, b, 8, droprate))
        self._lap = _Laplacian(adj, userNum + itemNum)
        self._relu_outside = relu_outside
        nn.init.normal_(self.uEmbd.weight, std=0.01)
        nn.init.normal_(self.iEmbd.weight, std=0.01)
        for lin in (self.transForm1, self.transForm2, self.transForm3):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.zero_()

    def forward(self, userIdx, itemIdx):
        U = self.userNum
        features = torch.cat([self.uEmbd.weight, self.iEmbd.weight], dim=0)
        prop = self._lap.get(features.device)
        final = [features]
        for ga in self.GALayers:
            features = ga(features[:U], features[U:], prop)
            if self._relu_outside:
                features = torch.relu(features)
            final.append(features)
        final = torch.cat(final, dim=1)
        embd = torch.cat([final[userIdx], final[itemIdx + U]], dim=1)
        embd = torch.relu(self.transForm1(embd))
        embd = torch.relu(self.transForm2(embd))
        return self.transForm3(embd).flatten()


class GACFV6(_GACFMlpHead):
    """GACFmodel6.py:45-161.  `rt`: the rating table -- a DataFrame / mapping with userId, itemId, rating columns (the reference
    builds its Laplacian from it, :139-161), or a ready (rows, cols, vals) triple / torch sparse tensor."""

    def __init__(self, userNum, itemNum, rt, embedSize=256, layers=(256, 128, 64), droprate=0.2, useCuda=True):
        if hasattr(rt, "__getitem__") and not isinstance(rt, (tuple, list)) and not torch.is_tensor(rt):
            import numpy as np
            adj = rating_laplacian_coo(np.asarray(rt["userId"]), np.asarray(rt["itemId"]), np.asarray(rt["rating"], dtype=np.float64),
                                       userNum, itemNum)
        else:
            adj = rt
        super().__init__(userNum, itemNum, adj, embedSize, layers, droprate, useCuda, GALayerV6, relu_outside=False)


class GACFV3x(_GACFMlpHead):
    """GACFmodel3.py:114-200: GACFV3's layers (attention through both Laplacian products), a ReLU around each, the MLP head."""

    def __init__(self, userNum, itemNum, adj, embedSize=256, layers=(256, 128, 64), droprate=0.2, useCuda=True):
        super().__init__(userNum, itemNum, adj, embedSize, layers, droprate, useCuda,
                         lambda a, b, h, p: GALayer(a, b, h, p, variant=3), relu_outside=True)


class GPLayer(nn.Module):
    """GACFMask.py:22-32: features -> (L + I) @ features."""

    def forward(self, features, propagation, selfLoop=True):
        return propagation(features, self_loop=bool(selfLoop))


class GACFV2_layer(nn.Module):
    """Per layer attention, (L + I) propagation, Linear, ReLU; all stage outputs concatenated.
    masked=True : GACFMask.py:35-106 -- masked self-attention over all nodes (ATTLayer_mask), forward(u, i, mask);
    masked=False: GACFmodel2.py:34-120 -- user/item cross attention (ATTLayer), forward(u, i)."""
    masked = True

    def __init__(self, userNum, itemNum, adj, embedSize=256, layers=(256, 128, 64), droprate=0.2, useCuda=True):
        super().__init__()
        self.useCuda, self.userNum, self.itemNum, self.droprate = useCuda, userNum, itemNum, droprate
        self.uEmbd = nn.Embedding(userNum, embedSize)
        self.iEmbd = nn.Embedding(itemNum, embedSize)
        self.ATTlayers, self.GPlayers, self.Affinelayers = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for a, b in zip(layers[:-1], layers[1:]):
            _check_width(a)
            self.ATTlayers.append(ATTLayer_mask(a, 8, droprate) if self.masked else ATTLayer(a, 8, droprate))
            self.GPlayers.append(GPLayer())
            self.Affinelayers.append(nn.Linear(a, b))
        self._lap = _Laplacian(adj, userNum + itemNum)
        nn.init.normal_(self.uEmbd.weight, std=0.01)
        nn.init.normal_(self.iEmbd.weight, std=0.01)
        for m in self.Affinelayers:
            nn.init.xavier_uniform_(m.weight)
            m.bias.data.zero_()

    def forward(self, userIdx, itemIdx, mask=None):
        U = self.userNum
        features = torch.cat([self.uEmbd.weight, self.iEmbd.weight], dim=0)
        prop = self._lap.get(features.device)
        final = [features]
        for att, gp, aff in zip(self.ATTlayers, self.GPlayers, self.Affinelayers):
            attended = att(features, mask) if self.masked else att(features[:U], features[U:])
            features = torch.relu(aff(gp(attended, prop, True)))
            final.append(features)
        return torch.cat(final, dim=1)


class GACFV2_plain_layer(GACFV2_layer):
    masked = False


class GACFV2(nn.Module):
    """GACFmodel2.py:122-135."""

    def __init__(self, userNum, itemNum, adj, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self.userNum = userNum
        self.gacf = GACFV2_plain_layer(userNum, itemNum, adj, embedSize, layers, droprate)

    def forward(self, userIdx, itemIdx):
        e = self.gacf(userIdx, itemIdx)
        return torch.sum(e[userIdx] * e[itemIdx + self.userNum], dim=1)


class GACFMask(nn.Module):
    def __init__(self, userNum, itemNum, adj, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self.userNum = userNum
        self.gacf = GACFV2_layer(userNum, itemNum, adj, embedSize, layers, droprate)

    def forward(self, userIdx, itemIdx, mask):
        e = self.gacf(userIdx, itemIdx, mask)
        return torch.sum(e[userIdx] * e[itemIdx + self.userNum], dim=1)


def attention_flops(Nq, Nk, dim):
    """2 * (q k^T) + 2 * (p v) over all heads: 4 * Nq * Nk * dim flop (forward)."""
    return 4.0 * Nq * Nk * dim


def attention_exps(Nq, Nk, heads):
    return float(Nq) * Nk * heads

