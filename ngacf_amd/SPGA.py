"""The SPGA family's sparse layer on the HIP path (graphattention/SPGA.py:330-421, duplicated in
graphattention/modules.py:180-242): homogeneous N x N graph, one shared W, edges = adj.nonzero(),
e_ij = exp(-LeakyReLU(a . [h_i || h_j])), h'_i = sum_j drop(e_ij) h_j / sum_j e_ij, NO residual, ELU when concat.

Same traversal kernels as SPUIGACF, run on the graph's bipartite double cover: node i appears once as a
destination ("user" i) and once as a source ("item" i), edge (i -> j) becomes (user i, item j), W_u = W_i = W,
a_u = a[:k] (destination half), a_i = a[k:] (source half), residual flag cleared.  The cover is exactly the
structure the kernels (and their pull-only backward) are written for, so forward AND backward come for free; the
price is a duplicated table, acceptable for this secondary path.  Feature dropout (SpGAT, :352,:354) is applied
once, before the duplication, so both roles of a node see the same mask as in the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as NF
from .SPUIGACF import _PlanCache
from .plan import build_plan
from .SPUIGACF import _fresh_seed


def _edges_of(adj):
    """adj: dense [N,N] (reference: adj.nonzero().t(), SPGA.py:384), torch sparse, or an integer [2,E] edge list."""
    if adj.is_sparse:
        return adj.coalesce().indices()
    if adj.dim() == 2 and adj.shape[0] == 2 and not adj.dtype.is_floating_point and adj.shape[1] != 2:
        return adj
    return adj.nonzero().t()


class _CoverPlans(_PlanCache):
    """Plans of the double cover, cached like the bipartite ones (the entry holds the keyed `adj`, see _PlanCache)."""

    def get(self, adj, N, seg_len=256):
        def build():
            edge = _edges_of(adj)
            # a node without any outgoing edge has a zero attention row sum: the reference divides by it and trips its
            # NaN assert (SPGA.py:406,411); build_plan raises instead
            return build_plan(edge, N, N, seg_len=seg_len, device=edge.device if edge.is_cuda else None)
        key_t = adj if not adj.is_sparse else adj._values()
        return super().get(key_t, N, N, seg_len, builder=build)


def _cover_stage(x, W, a_dst, a_src, plan, heads, alpha, elu, drop_p, h_dtype=None):
    N = x.shape[0]
    x2 = torch.cat([x, x], 0)
    W2 = torch.stack([W, W])
    avec = torch.stack([a_dst, a_src])
    out = NF.stage(x2, W2, avec, plan, heads, alpha, elu, drop_p, 0.0, _fresh_seed() if drop_p > 0 else 0, h_dtype,
                   residual=False)
    return out[:N]


class SpGraphAttentionLayer(nn.Module):
    """SPGA.py:358-421.  forward(input [N,in], adj) -> [N,out]."""

    def __init__(self, in_features, out_features, dropout, alpha, concat=True):
        super().__init__()
        self.in_features, self.out_features, self.alpha, self.concat = in_features, out_features, alpha, concat
        self.W = nn.Parameter(torch.zeros(in_features, out_features))
        nn.init.xavier_normal_(self.W.data, gain=1.414)                  # :370-371
        self.a = nn.Parameter(torch.zeros(1, 2 * out_features))
        nn.init.xavier_normal_(self.a.data, gain=1.414)                  # :373-374
        self.droprate = dropout
        self._plans = _CoverPlans()

    def forward(self, input, adj):
        N, k = input.shape[0], self.out_features
        plan = self._plans.get(adj, N)
        p = self.droprate if self.training else 0.0
        if k in (64, 128):
            return _cover_stage(input, self.W, self.a[0, :k], self.a[0, k:], plan, 1, self.alpha, self.concat, p)
        if k != 8:
            raise ValueError("SpGraphAttentionLayer supports out_features 8, 64 or 128 (got %d)" % k)
        # a lone 8-wide head = head 0 of an 8-head stage whose other heads have zero weights
        W = torch.cat([self.W, input.new_zeros(self.in_features, 56)], 1)
        z = input.new_zeros(56)
        out = _cover_stage(input, W, torch.cat([self.a[0, :k], z]), torch.cat([self.a[0, k:], z]), plan, 8,
                           self.alpha, self.concat, p)
        return out[:, :k]

    def __repr__(self):
        return self.__class__.__name__ + " (" + str(self.in_features) + " -> " + str(self.out_features) + ")"


class SpGAT(nn.Module):
    """SPGA.py:330-356: dropout -> concat of nheads 8-wide heads (ELU) -> dropout -> ELU(out_att).  The heads of the
    first stage run as ONE fused 8-head traversal (nhid = 8, nheads = 8)."""

    def __init__(self, nfeat, nhid, nclass, dropout, alpha, nheads):
        super().__init__()
        if nhid != 8 or nhid * nheads not in (64, 128) or nclass not in (64, 128):
            raise ValueError("supported: nhid=8, nhid*nheads and nclass in {64,128}")
        self.dropout = dropout
        self.attentions = [SpGraphAttentionLayer(nfeat, nhid, dropout=dropout, alpha=alpha, concat=True) for _ in range(nheads)]
        for i, att in enumerate(self.attentions):
            self.add_module("attention_{}".format(i), att)
        self.out_att = SpGraphAttentionLayer(nhid * nheads, nclass, dropout=dropout, alpha=alpha, concat=False)
        self._plans = _CoverPlans()

    def forward(self, x, adj):
        N = x.shape[0]
        plan = self._plans.get(adj, N)
        p = self.dropout if self.training else 0.0
        k = self.attentions[0].out_features
        x = F.dropout(x, self.dropout, training=self.training)                                   # :352
        W = torch.cat([a.W for a in self.attentions], 1)
        a_dst = torch.cat([a.a[0, :k] for a in self.attentions])
        a_src = torch.cat([a.a[0, k:] for a in self.attentions])
        x = _cover_stage(x, W, a_dst, a_src, plan, len(self.attentions), self.attentions[0].alpha, True, p)   # :353
        x = F.dropout(x, self.dropout, training=self.training)                                   # :354
        o, ko = self.out_att, self.out_att.out_features
        return _cover_stage(x, o.W, o.a[0, :ko], o.a[0, ko:], plan, 1, o.alpha, True, p)         # :355 (ELU outside)


# ---------------------------------------------------------------------------------------------------------
# the model wrappers of graphattention/SPGA.py:85-316 (the reference's run_Mask.py that drove them is absent; kept for
# completeness of the SPGA family).  `adj` = the N x N Laplacian (torch sparse COO or a (rows, cols, vals) triple) for the
# GPLayer products, `mask` = the N x N adjacency SpGAT attends over (dense, sparse, or a [2,E] edge list).
# ---------------------------------------------------------------------------------------------------------
class _SPGABase(nn.Module):
    def _embeddings(self, userNum, itemNum, embedSize, droprate, useCuda):
        self.useCuda, self.userNum, self.itemNum, self.droprate = useCuda, userNum, itemNum, droprate
        self.uEmbd = nn.Embedding(userNum, embedSize)
        self.iEmbd = nn.Embedding(itemNum, embedSize)

    def _init_weight_(self):
        nn.init.normal_(self.uEmbd.weight, std=0.01)
        nn.init.normal_(self.iEmbd.weight, std=0.01)

    def _features(self):
        return torch.cat([self.uEmbd.weight, self.iEmbd.weight], dim=0)

    def _score(self, features, userIdx, itemIdx):
        return torch.sum(features[userIdx] * features[itemIdx + self.userNum], dim=1)

    def _propagation(self, dev):
        from .modules import _Laplacian
        if getattr(self, "_lap", None) is None:
            object.__setattr__(self, "_lap", _Laplacian(self.LaplacianMat, self.userNum + self.itemNum))
        return self._lap.get(dev)


class GPLayer(nn.Module):
    """SPGA.py:318-328: (L + I) @ features."""

    def forward(self, features, propagation, selfLoop=True):
        return propagation(features, self_loop=bool(selfLoop))


class SPGACF(_SPGABase):
    """SPGA.py:85-149: embeddings -> SpGAT over the mask -> dot-product score."""

    def __init__(self, userNum, itemNum, adj, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self._embeddings(userNum, itemNum, embedSize, droprate, useCuda)
        self.gat = SpGAT(nfeat=embedSize, nhid=8, nclass=embedSize, dropout=droprate, nheads=8, alpha=0.2)
        self._init_weight_()

    def forward(self, userIdx, itemIdx, mask):
        return self._score(self.gat(self._features(), mask), userIdx, itemIdx)


class MultiLayerSPGA(_SPGABase):
    """SPGA.py:151-234: per layer SpGAT -> (L + I) propagation -> Linear -> ReLU; all stage outputs concatenated."""

    def __init__(self, userNum, itemNum, adj, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self._embeddings(userNum, itemNum, embedSize, droprate, useCuda)
        self.GATlayers, self.GPlayers, self.Affinelayers = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.LaplacianMat = adj
        dims = [embedSize] + list(layers)                                # :172 (without mutating the caller's list)
        for From, To in zip(dims[:-1], dims[1:]):
            self.GATlayers.append(SpGAT(nfeat=From, nhid=8, nclass=From, dropout=droprate, nheads=8, alpha=0.2))
            self.GPlayers.append(GPLayer())
            self.Affinelayers.append(nn.Linear(From, To))
        self._init_weight_()

    def forward(self, userIdx, itemIdx, mask):
        features = self._features()
        prop = self._propagation(features.device)
        final = [features]
        for gat, gp, aff in zip(self.GATlayers, self.GPlayers, self.Affinelayers):
            features = torch.relu(aff(gp(gat(features, mask), prop, True)))
            final.append(features)
        return self._score(torch.cat(final, dim=1), userIdx, itemIdx)


class SPGAMGP(_SPGABase):
    """SPGA.py:236-316: one SpGAT, then len(layers) x [(L + I) propagation -> Linear -> ReLU], concatenated."""

    def __init__(self, userNum, itemNum, adj, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self._embeddings(userNum, itemNum, embedSize, droprate, useCuda)
        self.GPlayers, self.Affinelayers = nn.ModuleList(), nn.ModuleList()
        self.LaplacianMat = adj
        self.gat = SpGAT(nfeat=embedSize, nhid=8, nclass=embedSize, dropout=droprate, nheads=8, alpha=0.2)
        dims = [embedSize] + list(layers)
        for From, To in zip(dims[:-1], dims[1:]):
            self.GPlayers.append(GPLayer())
            self.Affinelayers.append(nn.Linear(From, To))
        self._init_weight_()

    def forward(self, userIdx, itemIdx, mask):
        features = self.gat(self._features(), mask)
        prop = self._propagation(features.device)
        final = [features]
        for gp, aff in zip(self.GPlayers, self.Affinelayers):
            features = torch.relu(aff(gp(features, prop, True)))
            final.append(features)
        return self._score(torch.cat(final, dim=1), userIdx, itemIdx)
