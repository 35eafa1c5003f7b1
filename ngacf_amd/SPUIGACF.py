"""Drop-in module surface of the reference's graphattention/SPUIGACF.py, backed by the
HIP library.  Constructor signatures, forward signatures, parameter names/shapes and
therefore state_dict keys are the reference's:

  SPUIGACF(userNum, itemNum, embedSize, layers, droprate, useCuda=True)        SPUIGACF.py:7
      .forward(userIdx[B], itemIdx[B], mask[2,E] | [1,2,E]) -> scores[B]       SPUIGACF.py:41
  SPUIMultiGACF (three propagation stages)                                     SPUIGACF.py:54
  SPUIGAGPCF  (+ GPLayer static-weight propagation and Linear/ReLU stack)      SPUIGACF.py:103
  SpUIGAT(nfeat, nhid, nclass, dropout, alpha, nheads).forward(uidx, iidx, features, adj)   :187
  SPUIMultiGAT                                                                 :217
  SpUIGraphAttentionLayer(in_dim, out_dim, dropout, alpha, concat)
      .forward(uidx, iidx, features, indices)                                  :258,:340

What differs from the reference is only *how* a stage is computed: the eight heads of a
stage run as one fused 64-wide pass (weights concatenated on the fly, so checkpoints stay
interchangeable), and dropout masks come from a counter hash instead of torch's RNG.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as NF
from .plan import build_plan, squeeze_mask

_PLAN_CACHE_MAX = 4
_SEED_STRIDE = 0x1E3779B97F4A7C15
_CTR_STRIDE = 0x2545F4914F6CDD1D          # device-counter step (odd, unrelated to the per-call stride)


def _content_digest(t):
    """(sum, weighted sum) of an integer tensor: changes when any element changes or two elements swap.  One device
    reduction + a host read (a synchronisation): used only by the opt-in check below."""
    f = t.reshape(-1).long()
    w = (torch.arange(f.numel(), device=f.device) % 65521) + 1
    return int(f.sum()), int((f * w).sum())


_VERIFY_PLAN_DIGEST = __import__("os").environ.get("NGACF_PLAN_DIGEST") == "1"


class _PlanCache:
    """Per-module cache of graph plans keyed by the `mask` tensor the caller passes on every forward
    (train_eval_Gowalla.py:131-132 passes the same `adj`; :106 re-uploads it every epoch).

    The key is (address, shape, device, in-place version counter) AND the entry holds a reference to the keyed tensor:
    while an entry lives its storage cannot be freed, so a DIFFERENT adjacency can never be allocated at the cached
    address (address + shape alone would silently reuse the old plan).  A new tensor -- even with equal
    content -- has its own storage and gets its own plan (one build per epoch under the reference's loop); an in-place
    write bumps `_version`.  Writes that bypass the version counter (`.data`, raw pointers) are the caller's to announce
    (invalidate()); NGACF_PLAN_DIGEST=1 verifies a content digest on every hit (one sync each: debugging).
    Cost of holding the reference: the mask's bytes stay allocated per entry (at most _PLAN_CACHE_MAX)."""

    def __init__(self):
        self._d = {}

    def invalidate(self):
        self._d.clear()

    def get(self, indices, U, I, seg_len, builder=None):
        key = (indices.data_ptr(), tuple(indices.shape), str(indices.device), indices._version, U, I, seg_len)
        hit = self._d.get(key)
        if hit is not None and _VERIFY_PLAN_DIGEST and hit[2] != _content_digest(indices):
            raise RuntimeError("the adjacency tensor changed under a cached plan without a version bump "
                               "(a write through .data or a raw pointer): call invalidate()")
        if hit is None:
            if len(self._d) >= _PLAN_CACHE_MAX:
                self._d.pop(next(iter(self._d)))
            build = builder or (lambda: build_plan(indices, U, I, seg_len=seg_len, device=indices.device if indices.is_cuda else None))
            hit = (build(), indices, _content_digest(indices) if _VERIFY_PLAN_DIGEST else None)
            self._d[key] = hit
        return hit[0]


class _PanelRef:
    """The packed weight panel of stage s, available once the stack's first transform has run (functional.StageLink.panels)."""
    __slots__ = ("link", "s")

    def __init__(self, link, s):
        self.link, self.s = link, s

    def get(self):
        return self.link.panels[self.s]


def _refuse_replica(module):
    """nn.DataParallel (the reference's --parallel path, parallel.py:94-130) shallow-copies the module per device and
    runs the replicas on threads: every replica would redo the whole propagation (that is what the reference does) and
    the stage / table caches of this module are not built for it.  Refuse loudly instead of reading another device's
    weights; the multi-GPU path is ngacf_amd.dist.ShardedSPUIGACF (one process per GPU, node-partitioned)."""
    if getattr(module, "_is_replica", False):
        raise RuntimeError("ngacf_amd modules cannot run as nn.DataParallel replicas (reference --parallel True, "
                           "parallel.py:94-130): use ngacf_amd.dist.ShardedSPUIGACF under torchrun (one process per GPU)")


def _fresh_seed():
    # torch's CPU generator: reproducible under torch.manual_seed (run_Gowalla.py:191), no device sync
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class _FuseHeads(torch.autograd.Function):
    """Per-head parameters (the reference's state_dict layout) -> the fused stage operands
    W [2,din,dout] and avec [2,dout].  The backward hands every head its slice of dW/davec as a
    view, so no zero-fill/copy kernels run per head.  Nothing is saved: it may run twice."""

    @staticmethod
    def forward(ctx, k, *params):
        h = len(params) // 3
        Wu = torch.cat(params[:h], 1)
        Wi = torch.cat(params[h:2 * h], 1)
        A = torch.cat(params[2 * h:], 0)                                  # [h, 2k]
        ctx.k, ctx.h = k, h
        return torch.stack([Wu, Wi]), torch.stack([A[:, :k].reshape(-1), A[:, k:].reshape(-1)])

    @staticmethod
    def backward(ctx, dW, da):
        k, h = ctx.k, ctx.h
        gu = [dW[0][:, j * k:(j + 1) * k] for j in range(h)]
        gi = [dW[1][:, j * k:(j + 1) * k] for j in range(h)]
        dA = torch.cat([da[0].view(h, k), da[1].view(h, k)], 1)           # [h, 2k]
        ga = [dA[j:j + 1] for j in range(h)]
        return (None, *gu, *gi, *ga)


def _fuse(layers):
    """Concatenate the heads of a stage: W [2,din,dout], avec [2,dout]."""
    ps = [l.W_u for l in layers] + [l.W_i for l in layers] + [l.a for l in layers]
    return _FuseHeads.apply(layers[0].out_dim, *ps)


class _FusedCache:
    """(W, avec) of a stage, reused while no parameter of the stage changed (both forwards of a
    BPR step, every forward of an evaluation).  Keyed by the parameters' version counters."""

    def __init__(self):
        self._d = {}

    def get(self, s, layers):
        ps = [l.W_u for l in layers] + [l.W_i for l in layers] + [l.a for l in layers]
        key = (tuple(p._version for p in ps), tuple(p.data_ptr() for p in ps), torch.is_grad_enabled(),
               any(p.requires_grad for p in ps))
        hit = self._d.get(s)
        if hit is None or hit[0] != key:
            hit = (key, _FuseHeads.apply(layers[0].out_dim, *ps))
            self._d[s] = hit
        return hit[1]


class SpUIGraphAttentionLayer(nn.Module):
    """Parameter holder + standalone layer call (SPUIGACF.py:258-403)."""

    def __init__(self, in_dim, out_dim, dropout, alpha, concat=True):
        super().__init__()
        self.in_dim, self.out_dim, self.alpha, self.concat = in_dim, out_dim, alpha, concat
        self.W_u = nn.Parameter(torch.zeros(in_dim, out_dim))
        nn.init.xavier_normal_(self.W_u.data, gain=1.414)            # :270-273
        self.W_i = nn.Parameter(torch.zeros(in_dim, out_dim))
        nn.init.xavier_normal_(self.W_i.data, gain=1.414)
        self.a = nn.Parameter(torch.zeros(1, 2 * out_dim))
        nn.init.xavier_normal_(self.a.data, gain=1.414)              # :277-278
        self.droprate = dropout
        self._plans = _PlanCache()
        self.seg_len = 256

    # ---- attached mode: inside a GAT stack the weights of all heads of a stage live in ONE fused parameter
    # (no per-head concat / gradient-slice kernels per step); the layer then only presents views of it.
    def _attach(self, owner, stage, head):
        for n in ("W_u", "W_i", "a"):
            del self._parameters[n]
        object.__setattr__(self, "_owner", (owner, stage, head))     # plain attribute: no module cycle

    def __getattr__(self, name):
        if name in ("W_u", "W_i", "a"):
            own = self.__dict__.get("_owner")
            if own is not None:
                return own[0]._head_view(own[1], own[2], name)
        return super().__getattr__(name)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if self.__dict__.get("_owner") is not None:
            return                                                    # the owning stack consumes this layer's keys
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def forward(self, uidx, iidx, features, indices):
        _refuse_replica(self)
        U, I = int(uidx.numel()), int(iidx.numel())
        plan = self._plans.get(squeeze_mask(indices), U, I, self.seg_len)
        p = self.droprate if self.training else 0.0
        k = self.out_dim
        if k in (64, 128):
            W, avec = _fuse([self])
            return NF.stage(features, W, avec, plan, 1, self.alpha, self.concat, p, 0.0, _fresh_seed())
        if k != 8:
            raise ValueError("standalone layer supports out_dim 8, 64 or 128 (got %d)" % k)
        # a lone 8-wide head = head 0 of a fused 8-head stage whose other heads have zero weights
        pad = features.new_zeros(self.in_dim, 56)
        W = torch.stack([torch.cat([self.W_u, pad], 1), torch.cat([self.W_i, pad], 1)])
        z = features.new_zeros(56)
        avec = torch.stack([torch.cat([self.a[0, :k], z]), torch.cat([self.a[0, k:], z])])
        out = NF.stage(features, W, avec, plan, 8, self.alpha, self.concat, p, 0.0, _fresh_seed())
        return out[:, :k]

    def __repr__(self):
        return self.__class__.__name__ + " (" + str(self.in_dim) + " -> " + str(self.out_dim) + ")"


class _GATBase(nn.Module):
    """Shared stage runner.  `self._stages` lists the layer groups in execution order."""

    compute_dtype = torch.float32    # storage type of H and of the stage outputs (fp32 | bf16)
    seg_len = 256

    def _finalize_stages(self):
        """Move the per-head weights the layer constructors drew (reference init, SPUIGACF.py:270-278) into one
        fused parameter pair per stage: stage{s}_W [2,din,dout] (W_u heads | W_i heads), stage{s}_a [2,dout]."""
        self._layer_names = {id(m): n for n, m in self.named_modules() if isinstance(m, SpUIGraphAttentionLayer)}
        for s, layers in enumerate(self._stages()):
            k = layers[0].out_dim
            W = torch.stack([torch.cat([l.W_u.data for l in layers], 1), torch.cat([l.W_i.data for l in layers], 1)])
            a = torch.stack([torch.cat([l.a.data[0, :k] for l in layers]), torch.cat([l.a.data[0, k:] for l in layers])])
            self.register_parameter("stage%d_W" % s, nn.Parameter(W.contiguous()))
            self.register_parameter("stage%d_a" % s, nn.Parameter(a.contiguous()))
            for h, l in enumerate(layers):
                l._attach(self, s, h)

    def stage_params(self, s):
        return getattr(self, "stage%d_W" % s), getattr(self, "stage%d_a" % s)

    def _stage_operands(self, s):
        """stage_params for the forward: behind the gradient sink of the caller, if there is one (functional.grad_sink)"""
        W, a = self.stage_params(s)
        return NF.through_sink(W), NF.through_sink(a)

    def stage_shapes(self):
        """[(din, dout, heads)] per stage"""
        return [(l[0].in_dim, l[0].out_dim * len(l), len(l)) for l in self._stages()]

    def stage_layer_names(self):
        """reference state_dict prefixes (relative to this module) of every stage's layers, in order"""
        return [[self._layer_names[id(l)] for l in layers] for layers in self._stages()]

    def _head_view(self, s, h, name):
        W, a = self.stage_params(s)
        k = self._stages()[s][0].out_dim
        sl = slice(h * k, (h + 1) * k)
        if name == "W_u":
            return W[0][:, sl]
        if name == "W_i":
            return W[1][:, sl]
        return torch.cat([a[0][sl], a[1][sl]])[None]

    def reference_named(self, what="data"):
        """{reference state_dict key (relative): tensor} of the weights (what='data') or gradients ('grad')."""
        out = {}
        for s, (layers, names) in enumerate(zip(self._stages(), self.stage_layer_names())):
            W, a = self.stage_params(s)
            if what == "grad":
                W, a = W.grad, a.grad
            k = layers[0].out_dim
            for h, n in enumerate(names):
                sl = slice(h * k, (h + 1) * k)
                out[n + ".W_u"], out[n + ".W_i"] = W[0][:, sl], W[1][:, sl]
                out[n + ".a"] = torch.cat([a[0][sl], a[1][sl]])[None]
        return out

    # state_dict in the reference's per-head layout (checkpoint interchange, run_Gowalla.py:127-130,143)
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for n, t in self.reference_named().items():
            destination[prefix + n] = t if keep_vars else t.detach().clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        with torch.no_grad():
            for s, (layers, names) in enumerate(zip(self._stages(), self.stage_layer_names())):
                W, a = self.stage_params(s)
                k = layers[0].out_dim
                for h, n in enumerate(names):
                    sl = slice(h * k, (h + 1) * k)
                    for key, dst in ((n + ".W_u", W[0][:, sl]), (n + ".W_i", W[1][:, sl])):
                        if prefix + key in state_dict:
                            src = state_dict[prefix + key]
                            if tuple(src.shape) != tuple(dst.shape):
                                error_msgs.append("size mismatch for %s: %s vs %s" % (prefix + key, tuple(src.shape), tuple(dst.shape)))
                            else:
                                dst.copy_(src)
                        elif strict:
                            missing_keys.append(prefix + key)
                    if prefix + n + ".a" in state_dict:
                        src = state_dict[prefix + n + ".a"]
                        if tuple(src.shape) != (1, 2 * k):
                            error_msgs.append("size mismatch for %s" % (prefix + n + ".a"))
                        else:
                            a[0][sl].copy_(src[0, :k])
                            a[1][sl].copy_(src[0, k:])
                    elif strict:
                        missing_keys.append(prefix + n + ".a")
            for p_ in self.parameters(recurse=False):
                p_._version                                            # (copy_ above already bumped the versions)

    def _run(self, uidx, iidx, features, adj, needed=None, score=None):
        """needed: ops.BatchRows (or a closure returning one) for the LAST stage (the rows the caller will read); earlier stages run in full.
        score = (userIdx, [itemIdx, ...]) with needed: return the tuple of score vectors (SPUIGACF.py:49-52) instead of the table
        -- the last stage and the scoring are one autograd node."""
        _refuse_replica(self)
        U, I = int(uidx.numel()), int(iidx.numel())
        plan = self._plans.get(squeeze_mask(adj), U, I, self.seg_len)
        p = self.dropout if self.training else 0.0
        seed_dev = None
        if p > 0 and getattr(self, "_seed_ctr", None) is not None:
            # graph-capturable seeding: seed = host part + device counter.  The host part differs per forward call
            # (a constant inside a captured graph); the counter is advanced once per step by its owner
            # (advance_device_seed(), a captured add in GraphedBPRStep) BEFORE the forwards, so a forward and its
            # backward read the same value and no per-call snapshot kernel is needed.
            self._seed_call += 1
            base = (self._seed_base + self._seed_call * _SEED_STRIDE) & 0x3FFFFFFFFFFFFFFF
            seed_dev = self._seed_ctr
        else:
            base = _fresh_seed() if p > 0 else 0
        x = features
        stages = self._stages()
        L = len(stages)
        ops_ = [self._stage_operands(s) for s in range(L)]
        # Which stage transforms ride in the previous stage's traversal (ngacf_prop_fwd_next)?  Stage s+1's can when stage s is a
        # bf16 8-head stage of 64 columns whose hub rows finish in-kernel, and stage s+1 maps 64 -> 64 | 128 columns.  Their
        # weight panels are packed by the FIRST transform's launch (functional.StageLink).
        shapes = self.stage_shapes()
        fuse = [False] * L                                   # fuse[s]: stage s's transform is done by stage s-1's traversal
        if NF.FUSE_NEXT and features.is_cuda:
            for s in range(1, L):
                fuse[s] = (NF.ops.can_fuse_next(plan, self.compute_dtype, shapes[s - 1][1], shapes[s - 1][2])
                           and shapes[s][0] == 64 and shapes[s][1] in (64, 128))
        # The backward's mirror image (ngacf_prop_bwd_dx): stage s's edge backward also produces dX of its transform and runs stage
        # s-1's step 1 on it -- bf16 64 -> 64 stages whose input is the fp32 table (s = 0) or an 8-head stage's bf16 output.
        fuse_b = [False] * L
        if NF.ops.FUSE_BWD and features.is_cuda and torch.is_grad_enabled() and self.compute_dtype == torch.bfloat16:
            for s in range(L):
                x_dtype = features.dtype if s == 0 else self.compute_dtype
                fuse_b[s] = NF.ops.can_fuse_bwd(plan, self.compute_dtype, shapes[s][1], shapes[s][0], x_dtype)
        link = NF.StageLink() if any(fuse) or any(fuse_b) else None
        if link is not None:
            link.defer_dw = NF.sink_defers()
        if link is not None and self.compute_dtype == torch.bfloat16:
            link.pack = {s: ops_[s][0].detach() for s in range(1, L) if fuse[s]}       # bf16 fragments, packed by the first transform's launch
            link.pack_t = {s: ops_[s][0].detach() for s in range(L) if fuse_b[s]}
        elif link is not None:
            link.panels = {s: ops_[s][0].detach().contiguous().float() for s in range(1, L) if fuse[s]}   # fp32: the weights themselves
        for s, layers in enumerate(stages):
            W, avec = ops_[s]
            # feature dropout in front of every stage (:208,:213) and the ELU behind it (:397 inside
            # the concat layers, :214 around out_att) are fused into the stage
            if callable(needed) and s == L - 1:
                needed = needed()                           # the batch's rows are marked HERE, in front of the stage that wants them
            if score is not None and needed is not None and s == L - 1:
                return NF.stage_score(x, W, avec, plan, len(layers), needed, score[0], score[1], layers[0].alpha, True, p, p,
                                      NF.stage_seed(base, s), self.compute_dtype, seed_dev, link=link, sidx=s)
            nxt = None
            if s + 1 < L and fuse[s + 1]:
                nxt = NF.ops.NextStage(None, ops_[s + 1][1].detach().contiguous().float(), len(stages[s + 1]), p, NF.stage_seed(base, s + 1))
                nxt.panel = _PanelRef(link, s + 1)           # resolved at launch: the stack's first transform packs the panels
            x = NF.stage(x, W, avec, plan, len(layers), layers[0].alpha, True, p, p,
                         NF.stage_seed(base, s), self.compute_dtype, seed_dev,
                         needed=needed if s == L - 1 else None, link=link, nxt=nxt, sidx=s)
            if getattr(self, "debug_asserts", False) and needed is not None and s == L - 1:
                assert not bool(torch.isnan(x[needed.mask()]).any()), "NaN after propagation stage %d" % s
            elif getattr(self, "debug_asserts", False) and x is not None:
                # the reference's per-layer checks (SPUIGACF.py:384,391); a device sync each, so off by default.
                # (:368, "every user has an edge", is enforced once at plan build.)
                assert not bool(torch.isnan(x).any()), "NaN after propagation stage %d" % s
        return x

    def use_device_seed(self, device, on=True):
        """Dropout seeds from a device-side counter (needed under hipGraph capture: a host-drawn seed
        would be frozen into the graph)."""
        if on:
            self._seed_base = _fresh_seed()
            self._seed_call = 0
            self._seed_ctr = torch.zeros(1, dtype=torch.int64, device=device)
        else:
            self._seed_ctr = None

    def advance_device_seed(self):
        """One (capturable) add on the device counter: call once per training step, before the forwards."""
        if getattr(self, "_seed_ctr", None) is not None:
            self._seed_ctr.add_(_CTR_STRIDE)


class SpUIGAT(_GATBase):
    def __init__(self, nfeat, nhid, nclass, dropout, alpha, nheads):
        super().__init__()
        self.dropout = dropout
        self.attentions = [SpUIGraphAttentionLayer(nfeat, nhid, dropout=dropout, alpha=alpha, concat=True)
                           for _ in range(nheads)]
        for i, attention in enumerate(self.attentions):
            self.add_module("attention_{}".format(i), attention)
        self.out_att = SpUIGraphAttentionLayer(nhid * nheads, nclass, dropout=dropout, alpha=alpha, concat=False)
        self._plans = _PlanCache()
        self._finalize_stages()

    def _stages(self):
        return [self.attentions, [self.out_att]]

    def forward(self, uidx, iidx, features, adj, needed=None, score=None):
        return self._run(uidx, iidx, features, adj, needed, score)


class SPUIMultiGAT(_GATBase):
    def __init__(self, nfeat, nhid, nclass, dropout, alpha, nheads):
        super().__init__()
        self.dropout = dropout
        self.attentions1 = [SpUIGraphAttentionLayer(nfeat, nhid, dropout=dropout, alpha=alpha, concat=True)
                            for _ in range(nheads)]
        self.attentions2 = [SpUIGraphAttentionLayer(nfeat, nhid, dropout=dropout, alpha=alpha, concat=True)
                            for _ in range(nheads)]
        for i, attention in enumerate(self.attentions1):
            self.add_module("attention1_{}".format(i), attention)
        for i, attention in enumerate(self.attentions2):
            self.add_module("attention2_{}".format(i), attention)
        self.out_att = SpUIGraphAttentionLayer(nhid * nheads, nclass, dropout=dropout, alpha=alpha, concat=False)
        self._plans = _PlanCache()
        self._finalize_stages()

    def _stages(self):
        return [self.attentions1, self.attentions2, [self.out_att]]

    def forward(self, uidx, iidx, features, adj, needed=None, score=None):
        return self._run(uidx, iidx, features, adj, needed, score)


class StackedSpUIGAT(_GATBase):
    """Generalised depth (BASELINE config "d=128 L=4"): n_hidden 8-head stages + the out layer.
    Not in the reference; state_dict keys follow SPUIMultiGAT's numbering (attention{s}_{h})."""

    def __init__(self, nfeat, nhid, nclass, dropout, alpha, nheads, n_hidden):
        super().__init__()
        self.dropout = dropout
        self.hidden = []
        for s in range(n_hidden):
            grp = [SpUIGraphAttentionLayer(nfeat if s == 0 else nhid * nheads, nhid, dropout=dropout, alpha=alpha,
                                           concat=True) for _ in range(nheads)]
            for h, l in enumerate(grp):
                self.add_module("attention{}_{}".format(s + 1, h), l)
            self.hidden.append(grp)
        self.out_att = SpUIGraphAttentionLayer(nhid * nheads, nclass, dropout=dropout, alpha=alpha, concat=False)
        self._plans = _PlanCache()
        self._finalize_stages()

    def _stages(self):
        return self.hidden + [[self.out_att]]

    def forward(self, uidx, iidx, features, adj, needed=None, score=None):
        return self._run(uidx, iidx, features, adj, needed, score)


class _Table(torch.autograd.Function):
    """features = cat([uEmbd.weight, iEmbd.weight]) (SPUIGACF.py:36-38) without the copy: both embedding
    parameters alias one [N,d] table (see _CFBase._link_tables), which is returned as is; the backward hands
    each parameter its slice of the gradient."""

    @staticmethod
    def forward(ctx, wu, wi, flat):
        ctx.U = wu.shape[0]
        return flat.view(flat.shape)

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.U], g[ctx.U:], None


class _CFBase(nn.Module):
    def _init_weight_(self):
        nn.init.normal_(self.uEmbd.weight, std=0.01)                 # SPUIGACF.py:26-28
        nn.init.normal_(self.iEmbd.weight, std=0.01)
        self._link_tables()

    def _link_tables(self):
        """Make uEmbd.weight and iEmbd.weight views of one contiguous [U+I, d] table (same Parameter objects,
        same values; state_dict, optimizers and in-place updates are unaffected)."""
        wu, wi = self.uEmbd.weight, self.iEmbd.weight
        U = wu.shape[0]
        flat = torch.empty(U + wi.shape[0], wu.shape[1], device=wu.device, dtype=wu.dtype)
        flat[:U] = wu.data
        flat[U:] = wi.data
        wu.data, wi.data = flat[:U], flat[U:]
        object.__setattr__(self, "_flat", flat)

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)                               # .to()/.cuda()/.float() move the halves apart
        if "uEmbd" in self._modules and "iEmbd" in self._modules:
            self._link_tables()
        return r

    def _features(self):
        wu, wi, flat = self.uEmbd.weight, self.iEmbd.weight, self.__dict__.get("_flat")
        if flat is None or flat.device != wu.device or wu.data_ptr() != flat.data_ptr() \
                or wi.data_ptr() != flat.data_ptr() + wu.numel() * wu.element_size():
            self._link_tables()                                       # somebody re-assigned .data: alias again
            flat = self._flat
        return _Table.apply(NF.through_sink(wu), NF.through_sink(wi), flat)

    def getFeatureMat(self):
        """SPUIGACF.py:30-39 without the per-call Python index lists."""
        dev = self.uEmbd.weight.device
        idx = getattr(self, "_idx", None)
        if idx is None or idx[0].device != dev:                       # built once, not per call
            idx = (torch.arange(self.userNum, device=dev), torch.arange(self.itemNum, device=dev) + self.userNum)
            self._idx = idx
        return idx[0], idx[1], self._features()

    # A scored batch reads B user rows and B item rows of the LAST stage's output (:49-52): that stage is computed for those
    # rows only (ngacf_prop_fwd / _bwd `needed`), forward and backward -- same numbers, ~1/17 of the last stage's edges at
    # B = 2048 on a Gowalla-size graph.  False: every stage in full (A/B timing, tests).
    sparse_last_stage = os.environ.get("NGACF_SPARSE_LAST", "1") != "0"
    sparse_max_fraction = float(os.environ.get("NGACF_SPARSE_MAX_FRACTION", "0.2"))

    def propagate(self, mask, needed=None, score=None):
        _refuse_replica(self)
        uidx, iidx, features = self.getFeatureMat()
        return self.gat(uidx, iidx, features, squeeze_mask(mask), needed, score)        # :42-46

    def score_table(self, mask):
        """The [U + I, w] table whose rows' dot products are the model's scores (SPUIGACF.py:49-52): the propagated features
        (evaluation ranks every item against it once)."""
        return self.propagate(mask)

    def _sparse_batch(self, userIdx, n_item_lists):
        """Does the batch list few enough rows for the rows-only last stage to pay?  (a host-side decision)"""
        _refuse_replica(self)
        # measured crossover with the full stage at ~23 % of the rows (B = 8192 of a 70 839-row graph,
        # profiles/r02_sparse_probe_gowalla.log); beyond 1/5 the full stage runs
        return (self.sparse_last_stage
                and (1 + n_item_lists) * userIdx.numel() <= self.sparse_max_fraction * (self.userNum + self.itemNum))

    def _batch_rows(self, userIdx, *itemIdxs):
        """ops.BatchRows of the batch (None when the full stage runs).  The modules call it through a closure, right in front
        of the last stage: its two small launches then sit next to the other chain's big kernels of a two-chain train step,
        not at the head of both chains at once."""
        if not self._sparse_batch(userIdx, len(itemIdxs)):
            return None
        rows = None
        d = self.gat.stage_shapes()[-1][1] if torch.is_grad_enabled() else 0          # width of the last stage's output
        for it in itemIdxs:
            rows = NF.ops.mark_rows(userIdx.contiguous().long(), it.contiguous().long(), self.userNum, self.itemNum, rows, grad_d=d)
        return rows

    def forward(self, userIdx, itemIdx, mask):
        if self._sparse_batch(userIdx, 1):
            return self.propagate(mask, lambda: self._batch_rows(userIdx, itemIdx), (userIdx, [itemIdx]))[0]
        return NF.score(self.propagate(mask), userIdx, itemIdx, self.userNum)    # :49-52

    def forward_pairs(self, userIdx, posIdx, negIdx, mask):
        """OPT-IN, not reference semantics: one propagation shared by the positive and the negative scoring of a
        BPR step (the reference calls forward() twice, train_eval_Gowalla.py:128-129, i.e. two propagations with
        independent dropout masks).  Identical to two forward() calls when droprate == 0 or in eval mode."""
        if self._sparse_batch(userIdx, 2):
            return self.propagate(mask, lambda: self._batch_rows(userIdx, posIdx, negIdx), (userIdx, [posIdx, negIdx]))
        features = self.propagate(mask)
        return (NF.score(features, userIdx, posIdx, self.userNum), NF.score(features, userIdx, negIdx, self.userNum))


class SPUIGACF(_CFBase):
    def __init__(self, userNum, itemNum, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self.useCuda, self.userNum, self.itemNum, self.droprate = useCuda, userNum, itemNum, droprate
        self.uEmbd = nn.Embedding(userNum, embedSize)
        self.iEmbd = nn.Embedding(itemNum, embedSize)
        self.gat = SpUIGAT(nfeat=embedSize, nhid=8, nclass=embedSize, dropout=droprate, nheads=8, alpha=0.2)
        self._init_weight_()


class SPUIMultiGACF(_CFBase):
    def __init__(self, userNum, itemNum, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self.useCuda, self.userNum, self.itemNum, self.droprate = useCuda, userNum, itemNum, droprate
        self.uEmbd = nn.Embedding(userNum, embedSize)
        self.iEmbd = nn.Embedding(itemNum, embedSize)
        self.gat = SPUIMultiGAT(nfeat=embedSize, nhid=8, nclass=embedSize, dropout=droprate, nheads=8, alpha=0.2)
        self._init_weight_()


class GPLayer(nn.Module):
    """SPUIGACF.py:174-185: features -> (laplacianMat + selfLoop) @ features (static-weight propagation)."""

    def forward(self, features, propagation, selfLoop=True):
        return propagation(features, self_loop=bool(selfLoop))


class SPUIGAGPCF(_CFBase):
    """SPUIGACF.py:103-172: SpUIGAT, then len(layers) x [(L + I) @ X -> Linear -> ReLU], all stage outputs
    concatenated and scored by a dot product.  `adj` is the Laplacian as a torch sparse COO tensor (what
    data/loadGowalla.py:get_adj_mat returns for adj_type 'mean_adj' / 'norm_adj') or a (rows, cols, vals) triple."""

    def __init__(self, userNum, itemNum, adj, embedSize, layers, droprate, useCuda=True):
        super().__init__()
        self.useCuda, self.userNum, self.itemNum, self.droprate = useCuda, userNum, itemNum, droprate
        self.uEmbd = nn.Embedding(userNum, embedSize)
        self.iEmbd = nn.Embedding(itemNum, embedSize)
        self.gat = SpUIGAT(nfeat=embedSize, nhid=8, nclass=embedSize, dropout=droprate, nheads=8, alpha=0.2)
        self.GPlayers = nn.ModuleList()
        self.Affinelayers = nn.ModuleList()
        dims = [embedSize] + list(layers)                                # :127 (without mutating the caller's list)
        for From, To in zip(dims[:-1], dims[1:]):
            self.GPlayers.append(GPLayer())
            self.Affinelayers.append(nn.Linear(From, To))
        self._adj, self._prop = adj, None
        self._init_weight_()

    def _propagation(self, dev):
        if self._prop is None:
            from .spmm import StaticPropagation
            if isinstance(self._adj, (tuple, list)):
                r, c, v = self._adj
            else:
                a = self._adj.coalesce()
                r, c, v = a.indices()[0], a.indices()[1], a.values()
            self._prop = StaticPropagation(r.to(dev), c.to(dev), v.to(dev).float(), self.userNum + self.itemNum)
        return self._prop

    def score_table(self, mask):
        """SPUIGACF.py:160-167: the GAT's features and every GP layer's output side by side ([U + I, 64 + sum(layers)])."""
        features = self.propagate(mask).float()
        prop = self._propagation(features.device)
        final = [features]
        for gp, aff in zip(self.GPlayers, self.Affinelayers):
            if features.shape[1] not in (64, 128):
                raise ValueError("GPLayer widths must be 64 or 128 on the accelerated path")
            features = torch.relu(aff(gp(features, prop, True)))          # :164-165
            final.append(features)
        return torch.cat(final, dim=1)                                    # :167

    def forward(self, userIdx, itemIdx, mask):
        final = self.score_table(mask)
        return torch.sum(final[userIdx] * final[itemIdx + self.userNum], dim=1)


class SPUIStackedGACF(_CFBase):
    """Depth-generalised model for the BASELINE "L=4" config (stages = n_hidden + 1)."""

    def __init__(self, userNum, itemNum, embedSize, layers, droprate, useCuda=True, stages=4):
        super().__init__()
        self.useCuda, self.userNum, self.itemNum, self.droprate = useCuda, userNum, itemNum, droprate
        self.uEmbd = nn.Embedding(userNum, embedSize)
        self.iEmbd = nn.Embedding(itemNum, embedSize)
        self.gat = StackedSpUIGAT(embedSize, 8, embedSize, droprate, 0.2, 8, stages - 1)
        self._init_weight_()


def reference_named_grads(model):
    """{reference state_dict key: gradient} for a *CF model (embeddings + the per-head slices of the fused stage
    gradients), e.g. to compare with the reference's named_parameters() gradients."""
    out = {"uEmbd.weight": model.uEmbd.weight.grad, "iEmbd.weight": model.iEmbd.weight.grad}
    out.update({"gat." + k: v for k, v in model.gat.reference_named("grad").items()})
    return out


def set_compute_dtype(model, dtype):
    """bf16 storage of H / stage outputs (fp32 accumulate) for every GAT stack inside `model`."""
    for m in model.modules():
        if isinstance(m, _GATBase):
            m.compute_dtype = dtype
    return model
