"""autograd glue: one torch.autograd.Function per C-ABI forward/backward pair.  The math
lives in the HIP library; these classes only route tensors and saved state."""
import threading

import torch

from . import ops

_STAGE_SEED_STRIDE = 0x9E3779B97F4A7C15


FUSE_NEXT = __import__("os").environ.get("NGACF_FUSE_NEXT", "1") != "0"     # A/B switch: "0" = every transform as its own launch


class StageLink:
    """What travels from one stage of a stack to the next when the next stage's transform rides in this stage's traversal
    (ngacf_prop_fwd_next): `pack` (stage 0 only) = the later stages' weights to pack in the first transform's launch, `panels` =
    the packed panels by stage index, `pre` = (H, S) of the coming stage, produced by the previous traversal."""
    __slots__ = ("pack", "panels", "pre", "pack_t", "panels_t", "saved", "node_pre", "defer_dw", "dw_jobs", "bwd_stages", "dw_ctx")

    def __init__(self):
        self.pack, self.panels, self.pre = None, {}, None
        # the backward's mirror image (ngacf_prop_bwd_dx): pack_t = {stage: W} whose dX fragments the first transform's launch
        # packs too, panels_t = those panels.  fp32 storage: saved[s] = (out, agg, R, S, elu, alpha) of stage s's forward, read by
        # stage s+1's dX kernel (ngacf_transform_dx_node), which runs stage s's step 1 on its dX and parks (dX, Gn, ST, dS) in node_pre[s]
        self.pack_t, self.panels_t, self.saved, self.node_pre = None, {}, {}, {}
        # defer_dw (a gradient sink is active: nobody reads a weight gradient before the stack's backward has finished): the stages
        # whose dX came out of the edge backward leave their dW / davec to ONE batched launch behind the stack's last backward
        # (ngacf_transform_dw_batch); dw_jobs = the pending jobs, bwd_stages = the stages that have a backward
        self.defer_dw, self.dw_jobs, self.bwd_stages = False, [], set()
        self.dw_ctx = None           # (plan, seed_dev, stream) of the pending jobs, for the end-of-backward flush (_defer_dw)


def _transform_or_pre(X, W, avec, plan, heads, h_dtype, feat_drop_p, seed, seed_dev, link):
    """(H, S) of the stage: handed over by the previous traversal (link.pre), or computed here -- the first transform of a stack
    also packs the later stages' weight panels (link.pack = {stage: W})."""
    if link is not None and link.pre is not None:
        pre, link.pre = link.pre, None
        return pre
    if link is not None and (link.pack or link.pack_t):
        idx, idx_t = sorted(link.pack or {}), sorted(link.pack_t or {})
        H, S, panels = ops.transform_fwd(X, W, avec, plan.U, plan.I, heads, h_dtype, feat_drop_p, seed, seed_dev,
                                         pack=[link.pack[k].contiguous().float() for k in idx] + [link.pack_t[k].contiguous().float() for k in idx_t],
                                         pack_kind=[0] * len(idx) + [1] * len(idx_t))
        link.panels.update(zip(idx, panels[:len(idx)]))
        link.panels_t.update(zip(idx_t, panels[len(idx):]))
        link.pack = link.pack_t = None
        return H, S
    return ops.transform_fwd(X, W, avec, plan.U, plan.I, heads, h_dtype, feat_drop_p, seed, seed_dev)


def _remember(link, sidx, out, agg, R, S, elu, alpha, needed):
    """forward: leave this stage's saved tables where the NEXT stage's backward finds them (StageLink.saved)"""
    if link is None or sidx is None:
        return None
    link.bwd_stages.add(sidx)
    if needed is None:                                   # (a stage computed for a batch's rows only is nobody's previous stage)
        link.saved[sidx] = (out, agg, R, S, elu, alpha)
    return (link, sidx)


# The stage of a scored batch (row flags) keeps its separate launches by default: most of its rows finish without a single load,
# and a tail with a barrier, a tile step and the previous stage's rows costs it more (38 -> 55 us stand-alone) than the launches
# it replaces (dX 4 us, the previous stage's step 1 10 us); "1" fuses it too (tested, bitwise the same numbers).
FUSE_SPARSE = __import__("os").environ.get("NGACF_FUSE_SPARSE", "0") != "0"


def _bwd_links(bw, gout, X, H, plan, feat_drop_p, seed, want_dx, sparse=False):
    """backward of stage s: node = (Gn, ST, dS) when the next stage's backward already ran this stage's step 1 on exactly this gout;
    dx = ops.BwdDx when this stage's launch can produce dX itself."""
    if bw is None or X is None:
        return None, None
    link, s = bw
    node = dx = None
    pre = link.node_pre.pop(s, None)
    if pre is not None and pre[0] is not None and pre[0].data_ptr() == gout.data_ptr() and pre[0].dtype == gout.dtype and gout.is_contiguous():
        node = (pre[1], pre[2], pre[3])
    if want_dx and s in link.panels_t and (FUSE_SPARSE or not sparse):
        if ops.can_fuse_bwd(plan, H.dtype, H.shape[1], X.shape[1], X.dtype):
            dx = ops.BwdDx(link.panels_t[s], X.dtype, feat_drop_p, seed)
    return node, dx


DEFER_DW = __import__("os").environ.get("NGACF_DEFER_DW", "1") != "0"      # A/B switch of the batched, deferred weight gradients
FUSE_DX_NODE = __import__("os").environ.get("NGACF_FUSE_DX_NODE", "1") != "0"   # A/B switch (fp32 storage): the previous stage's step 1 in the dX kernel


def _launch_dw(link):
    jobs, link.dw_jobs = link.dw_jobs, []
    plan, seed_dev, _ = link.dw_ctx
    for k in range(0, len(jobs), ops._lib.DW_BATCH_MAX):
        ops.transform_dw_batch(jobs[k:k + ops._lib.DW_BATCH_MAX], plan.U, plan.I, seed_dev)


def _flush_dw(bw, plan, seed_dev):
    """the stack's last backward stage: every deferred weight gradient in one launch"""
    if bw is None:
        return
    link, s = bw
    if link.dw_jobs and s == min(link.bwd_stages):
        _launch_dw(link)


def _defer_dw(link, job, plan, seed_dev):
    """Park a stage's weight-gradient job for the chain's batched launch.  The launch normally rides behind the backward of the
    stack's LOWEST stage (_flush_dw) -- but that backward need not run at all (torch.autograd.grad on later stages' parameters, a
    frozen first stage whose inputs need no gradient): the first job of a backward pass therefore also queues an end-of-pass
    callback that launches whatever is still pending, on the stream the jobs were recorded on, so that no handed-out dW / davec
    tensor ever stays unwritten."""
    first = not link.dw_jobs
    link.dw_jobs.append(job)
    link.dw_ctx = (plan, seed_dev, torch.cuda.current_stream() if job[0].is_cuda else None)

    def _at_end():
        if link.dw_jobs:
            st = link.dw_ctx[2]
            if st is not None:
                with torch.cuda.stream(st):
                    _launch_dw(link)
            else:
                _launch_dw(link)
    if first:
        torch.autograd.Variable._execution_engine.queue_callback(_at_end)


def _transform_bwd_after(bw, dx, X, W, H, dH, dS, plan, heads, feat_drop_p, seed, seed_dev):
    if dx is None:
        f32_stage = H.dtype == torch.float32 and X is not None and X.dtype == torch.float32
        if (bw is not None and bw[0].defer_dw and DEFER_DW and X is not None and (H.dtype == torch.bfloat16 or f32_stage)
                and tuple(W.shape) == (2, 64, 64) and plan.n_rows == plan.N):
            # (bf16: the stage of a scored batch, which keeps its edge backward unfused; fp32 storage: every stage:) dX now -- the next
            # backward stage waits for it -- and the weights' gradient with the chain's other stages in the batched launch: 9 us in
            # front of the next stage instead of 15 + 9 (dW + dX, then the reduction of its 1024 partial sums); fp32: 24 instead of 39 + 18
            prev = bw[0].saved.get(bw[1] - 1) if f32_stage and FUSE_DX_NODE else None
            if prev is not None and (prev[0].data_ptr() != X.data_ptr() or prev[0].dtype != torch.float32 or tuple(prev[3].shape) != (plan.N, 8)):
                prev = None                                  # X is not the previous stage's fp32 output table in 8-column heads
            if prev is not None:                             # fp32 storage: the previous stage's step 1 in the dX kernel's epilogue
                dX, Gn, ST, P1 = ops.transform_dx_node(dH, W, plan.U, plan.I, feat_drop_p, seed, seed_dev, prev)
                bw[0].node_pre[bw[1] - 1] = (dX, Gn, ST, P1)
            else:
                dX, _, _ = ops.transform_bwd(X, W, H, dH, dS, plan.U, plan.I, heads, feat_drop_p, seed, seed_dev, part="dx")
            dW, da = torch.empty(2, 64, 64, dtype=torch.float32, device=H.device), torch.empty(2, 64, dtype=torch.float32, device=H.device)
            _defer_dw(bw[0], (X, H, dH, dS, heads, feat_drop_p, seed, dW, da), plan, seed_dev)
            _flush_dw(bw, plan, seed_dev)                    # (fp32 storage: every stage comes through here, the chain's first one too)
            return dX, dW, da
        r = ops.transform_bwd(X, W, H, dH, dS, plan.U, plan.I, heads, feat_drop_p, seed, seed_dev)
        _flush_dw(bw, plan, seed_dev)
        return r
    link, s = bw
    if link.defer_dw and DEFER_DW and tuple(W.shape) == (2, 64, 64) and H.dtype == torch.bfloat16:
        dW, da = torch.empty(2, 64, 64, dtype=torch.float32, device=H.device), torch.empty(2, 64, dtype=torch.float32, device=H.device)
        _defer_dw(link, (X, H, dH, dS, heads, feat_drop_p, seed, dW, da), plan, seed_dev)
    else:
        _, dW, da = ops.transform_bwd(X, W, H, dH, dS, plan.U, plan.I, heads, feat_drop_p, seed, seed_dev, part="dw")
    _flush_dw(bw, plan, seed_dev)
    return dx.dX, dW, da


class _Stage(torch.autograd.Function):
    """One propagation stage: transform (X @ W per node type, scores) + attention
    propagation over both directions.  Mirrors what one list-comprehension over
    SpUIGraphAttentionLayer heads computes in SPUIGACF.py:212 (or the out_att call :214).
    link / nxt (StageLink, ops.NextStage): the stage's own transform may have been done by the previous stage's traversal, and
    this stage's traversal may do the next one's (nxt.H / nxt.S are filled and handed on through link.pre)."""

    @staticmethod
    def forward(ctx, X, W, avec, plan, heads, alpha, elu, drop_p, feat_drop_p, seed, h_dtype, seed_dev=None, residual=True, grad_mode=True,
                needed=None, link=None, nxt=None, sidx=None):
        X = X.contiguous()
        W = W.contiguous().float()
        avec = avec.contiguous().float()
        H, S = _transform_or_pre(X, W, avec, plan, heads, h_dtype, feat_drop_p, seed, seed_dev, link)
        # needs_input_grad ignores torch.no_grad() and grad mode is always off inside forward(): the caller's mode comes in as
        # an argument, so an eval / no_grad forward does not allocate and write the backward's agg [N,d] fp32 + R tables
        need = grad_mode and any(ctx.needs_input_grad[:3])
        if need:
            out, agg, R = ops.prop_fwd(plan, H, S, S, avec, heads, alpha, elu, residual, None, drop_p, seed, save=True,
                                       seed_dev=seed_dev, needed=needed, next=nxt)
            ctx.save_for_backward(X, W, avec, H, S, out, agg, R)
            ctx.cfg = (plan, heads, alpha, elu, drop_p, feat_drop_p, seed, seed_dev, residual, needed)
            ctx.bw = _remember(link, sidx, out, agg, R, S, elu, alpha, needed)
        else:
            out = ops.prop_fwd(plan, H, S, S, avec, heads, alpha, elu, residual, None, drop_p, seed, seed_dev=seed_dev, needed=needed, next=nxt)
        if nxt is not None:
            link.pre = (nxt.H, nxt.S)
        return out

    @staticmethod
    def backward(ctx, gout):
        X, W, avec, H, S, out, agg, R = ctx.saved_tensors
        plan, heads, alpha, elu, drop_p, feat_drop_p, seed, seed_dev, residual, needed = ctx.cfg
        node, dx = _bwd_links(ctx.bw, gout, X, H, plan, feat_drop_p, seed, ctx.needs_input_grad[0])
        gout = gout.contiguous().to(H.dtype)              # already H's type under autograd: no cast kernel
        # (needed: the rows of this stage's output the caller read -- gout is zero elsewhere and `out` is not valid there)
        dH, dS = ops.prop_bwd(plan, H, S, S, avec, out, agg, R, gout, heads, alpha, elu, residual, None, drop_p, seed, seed_dev, needed,
                              dx=dx, node=node)
        dX, dW, da = _transform_bwd_after(ctx.bw, dx, X, W, H, dH, dS, plan, heads, feat_drop_p, seed, seed_dev)
        return (dX, dW, da) + (None,) * 15


class _StageScore(torch.autograd.Function):
    """The LAST stage of a scored batch and its scores as one node: SPUIGACF.py:212-214 + :49-52.  Only the batch's rows of
    the stage exist (ops.BatchRows), forward and backward; the gradient of the scores goes into the rows of `rows.g32`
    (zeroed when the batch was marked) and the traversal's backward reads it there in fp32 -- no [N, d] zero fill, no cast."""

    @staticmethod
    def forward(ctx, X, W, avec, plan, heads, alpha, elu, drop_p, feat_drop_p, seed, h_dtype, seed_dev, residual, grad_mode, rows, link, sidx, u, *items):
        U, I = plan.U, plan.I
        X, W, avec = (X.contiguous() if X is not None else None), W.contiguous().float(), avec.contiguous().float()
        H, S = _transform_or_pre(X, W, avec, plan, heads, h_dtype, feat_drop_p, seed, seed_dev, link)
        u = u.contiguous().long()
        items = [it.contiguous().long() for it in items]
        if grad_mode and ((X is not None and X.requires_grad) or W.requires_grad or avec.requires_grad):
            if rows.g32 is None:
                raise ValueError("stage_score under autograd needs ops.mark_rows(..., grad_d=d)")
            out, agg, R = ops.prop_fwd(plan, H, S, S, avec, heads, alpha, elu, residual, None, drop_p, seed, save=True,
                                       seed_dev=seed_dev, needed=rows)
            ctx.save_for_backward(X, W, avec, H, S, out, agg, R, u, *items)
            ctx.cfg = (plan, heads, alpha, elu, drop_p, feat_drop_p, seed, seed_dev, residual, rows)
            ctx.bw = _remember(link, sidx, out, agg, R, S, elu, alpha, rows)
        else:
            out = ops.prop_fwd(plan, H, S, S, avec, heads, alpha, elu, residual, None, drop_p, seed, seed_dev=seed_dev, needed=rows)
        return tuple(ops.score_fwd(out, u, it, U) for it in items)

    @staticmethod
    def backward(ctx, *gs):
        X, W, avec, H, S, out, agg, R, u, *items = ctx.saved_tensors
        plan, heads, alpha, elu, drop_p, feat_drop_p, seed, seed_dev, residual, rows = ctx.cfg
        if rows.consumed:
            raise RuntimeError("second backward through the same scored batch: its gradient rows were zeroed once, by the forward "
                               "(run the forward again, or set sparse_last_stage = False for retain_graph use)")
        rows.consumed = True
        for it, g in zip(items, gs):
            if g is not None:
                ops.score_bwd(out, u, it, g.contiguous().float(), plan.U, into=rows.g32)
        _, dx = _bwd_links(ctx.bw, rows.g32, X, H, plan, feat_drop_p, seed, ctx.needs_input_grad[0], sparse=True)
        dH, dS = ops.prop_bwd(plan, H, S, S, avec, out, agg, R, rows.g32, heads, alpha, elu, residual, None, drop_p, seed, seed_dev,
                              rows, gout_f32=True, dx=dx)
        dX, dW, da = _transform_bwd_after(ctx.bw, dx, X, W, H, dH, dS, plan, heads, feat_drop_p, seed, seed_dev)
        return (dX, dW, da) + (None,) * (15 + len(items))


def stage_score(X, W, avec, plan, heads, rows, u, items, alpha=0.2, elu=True, drop_p=0.0, feat_drop_p=0.0, seed=0, h_dtype=None,
                seed_dev=None, residual=True, link=None, sidx=None):
    """scores[k][b] = out[u[b]] . out[U + items[k][b]] of the stage's output, computed for the rows of `rows` only."""
    return _StageScore.apply(X, W, avec, plan, heads, alpha, elu, drop_p, feat_drop_p, seed, X.dtype if h_dtype is None else h_dtype,
                             seed_dev, residual, torch.is_grad_enabled(), rows, link, sidx, u, *items)


def stage(X, W, avec, plan, heads, alpha=0.2, elu=True, drop_p=0.0, feat_drop_p=0.0, seed=0, h_dtype=None, seed_dev=None,
          residual=True, needed=None, link=None, nxt=None, sidx=None):
    """needed: ops.BatchRows (ops.mark_rows) -- ONLY those rows of the result are computed and may be read; the
    gradient flowing back must be zero elsewhere (true for a result that is only read through those rows).
    link / nxt: see _Stage.  Without autograd (evaluation) and with nxt, the stage's own output table is never written: the
    result is None and the next stage runs from link.pre alone."""
    h_dtype = X.dtype if h_dtype is None else h_dtype
    grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (X, W, avec))
    if not grad and (nxt is not None or X is None):
        # no autograd: X is None when the previous stage handed over (H, S) without ever writing its output table
        Wc, ac = W.contiguous().float(), avec.contiguous().float()
        H, S = _transform_or_pre(X.contiguous() if X is not None else None, Wc, ac, plan, heads, h_dtype, feat_drop_p, seed, seed_dev, link)
        out = ops.prop_fwd(plan, H, S, S, ac, heads, alpha, elu, residual, None, drop_p, seed, seed_dev=seed_dev, needed=needed, next=nxt,
                           want_out=nxt is None)
        if nxt is not None:
            link.pre = (nxt.H, nxt.S)
        return out
    return _Stage.apply(X, W, avec, plan, heads, alpha, elu, drop_p, feat_drop_p, seed, h_dtype, seed_dev, residual, torch.is_grad_enabled(),
                        needed, link, nxt, sidx)


def stage_seed(base, s):
    return (base + (s + 1) * _STAGE_SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF


class _Sink(torch.autograd.Function):
    """Identity on a parameter whose backward parks the gradient in sink[key] instead of sending it on: autograd then never
    accumulates into .grad, and the owner of the sink decides how the pieces are summed (ngacf_adam_step takes two)."""

    @staticmethod
    def forward(ctx, p, sink, key, deferred=False):
        ctx.sink, ctx.key, ctx.deferred = sink, key, deferred
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        prev = ctx.sink.get(ctx.key)
        if prev is not None and ctx.deferred:
            # a deferred sink's pieces may still be unwritten (the chain's batched dW launch fills them at the end of the backward):
            # adding two of them here would read uninitialised memory, silently
            raise RuntimeError("grad_sink(deferred_ok=True) takes ONE gradient piece per parameter: a parameter was read twice inside "
                               "the block (or two chains share the sink); give each chain its own sink, or drop deferred_ok")
        ctx.sink[ctx.key] = g if prev is None else prev + g
        return None, None, None, None


_SINKS = threading.local()


class grad_sink:
    """with grad_sink(d): parameters read by the modules of this package inside the block (stage weights, embedding tables)
    deliver their gradient to d[parameter] in the backward, not to parameter.grad.
    deferred_ok=True: the OWNER of the sink promises to read it only after the whole backward pass has finished (and its streams
    have been joined).  Only then may a stage hand the sink a weight gradient that is FILLED LATER -- by the chain's one batched
    ngacf_transform_dw_batch launch (StageLink.defer_dw); such a sink takes one piece per parameter (a second piece would be
    added to memory that may not have been written yet: it raises).  Without the flag every piece is complete when it arrives."""

    def __init__(self, sink, deferred_ok=False):
        self.sink, self.deferred_ok = sink, bool(deferred_ok)

    def __enter__(self):                     # (sink None: an explicit "no sink here" -- parameters get .grad as usual)
        stack = _SINKS.__dict__.setdefault("stack", [])
        stack.append((self.sink, self.deferred_ok))
        return self.sink

    def __exit__(self, *exc):
        _SINKS.stack.pop()


def sink_active():
    stack = _SINKS.__dict__.get("stack")
    return bool(stack) and stack[-1][0] is not None


def sink_defers():
    """Is the innermost sink one whose owner accepts gradients that are filled after they were handed over (grad_sink(deferred_ok=True))?"""
    stack = _SINKS.__dict__.get("stack")
    return bool(stack) and stack[-1][0] is not None and stack[-1][1]


def through_sink(p):
    """p itself, or (inside a grad_sink block, for a parameter that wants a gradient) p behind the sink"""
    stack = _SINKS.__dict__.get("stack")
    if not stack or stack[-1][0] is None or not p.requires_grad or not torch.is_grad_enabled():
        return p
    return _Sink.apply(p, stack[-1][0], p, stack[-1][1])


class _Score(torch.autograd.Function):
    """SPUIGACF.py:49-52: sum(features[userIdx] * features[itemIdx + U], dim=1)."""

    @staticmethod
    def forward(ctx, F, u, i, U):
        F = F.contiguous()
        u, i = u.contiguous().long(), i.contiguous().long()
        ctx.save_for_backward(F, u, i)
        ctx.U = U
        return ops.score_fwd(F, u, i, U)

    @staticmethod
    def backward(ctx, g):
        F, u, i = ctx.saved_tensors
        gF = ops.score_bwd(F, u, i, g.contiguous().float(), ctx.U)
        return gF.to(F.dtype), None, None, None


def score(F, u, i, U):
    return _Score.apply(F, u, i, U)


class _BPR(torch.autograd.Function):
    """graphattention/BPRLoss.py:8-9."""

    @staticmethod
    def forward(ctx, pos, neg):
        pos, neg = pos.contiguous().float(), neg.contiguous().float()
        ctx.save_for_backward(pos, neg)
        return ops.bpr_fwd(pos, neg)

    @staticmethod
    def backward(ctx, g):
        pos, neg = ctx.saved_tensors
        return ops.bpr_bwd(pos, neg, g)


def bpr_loss(pos, neg):
    return _BPR.apply(pos, neg)


class _ScorePairBPR(torch.autograd.Function):
    """loss = BPR(score(Fa, u, pos), score(Fb, u, neg)) as ONE autograd node (SPUIGACF.py:49-52 twice + BPRLoss.py:8-9).
    Used by the graphed step whose two propagations run as parallel branches: one node means both table gradients
    leave the loss at the same point of the stream order, so both backward branches start together."""

    @staticmethod
    def forward(ctx, Fa, Fb, u, pos, neg, U):
        Fa, Fb = Fa.contiguous(), Fb.contiguous()
        u, pos, neg = u.contiguous().long(), pos.contiguous().long(), neg.contiguous().long()
        ps, ns = ops.score_fwd(Fa, u, pos, U), ops.score_fwd(Fb, u, neg, U)
        ctx.save_for_backward(Fa, Fb, u, pos, neg, ps, ns)
        ctx.U = U
        return ops.bpr_fwd(ps, ns)

    @staticmethod
    def backward(ctx, g):
        Fa, Fb, u, pos, neg, ps, ns = ctx.saved_tensors
        gp, gn = ops.bpr_bwd(ps, ns, g)
        gFa = ops.score_bwd(Fa, u, pos, gp, ctx.U).to(Fa.dtype)
        gFb = ops.score_bwd(Fb, u, neg, gn, ctx.U).to(Fb.dtype)
        return gFa, gFb, None, None, None, None


def score_pair_bpr(Fa, Fb, u, pos, neg, U):
    return _ScorePairBPR.apply(Fa, Fb, u, pos, neg, U)
