"""hipGraph capture of the launch-bound loops (MI355X-first: HIP graphs instead of a tracing compiler).

A reference-semantics BPR step is ~180 small launches (2 propagations x 3 stages x {transform, traversal,
hub-row pass} forward and backward, scoring, loss, per-head gradient plumbing, Adam); eager mode is bound by the
host's launch rate, not by the GPU.  `GraphedBPRStep` captures one whole step -- forward x2, loss, backward,
optimizer -- into one graph and replays it; dropout seeds come from a device-side counter that a captured add
advances, so every replay draws fresh masks.  `GraphedPropagate` does the same for the eval-mode propagation.
"""
import os

import torch


# the warm-up runs on a side stream by design; autograd's stream-mismatch warning is expected there
try:
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
except AttributeError:      # older torch
    pass


def _capture_mode():
    """With a process group alive, c10d's watchdog thread polls collective events with hipEventQuery; under the default
    "global" capture mode such a call from ANY thread while this thread captures is an error that aborts the process
    (seen as an intermittent SIGABRT in ProcessGroupNCCL::Watchdog::run).  Thread-local mode restricts only this thread."""
    import torch.distributed as dist
    return "thread_local" if dist.is_available() and dist.is_initialized() else "global"


def quiesce_collectives():
    """Call right before a capture when a process group is alive.  c10d's watchdog keeps every EAGER collective in a list until
    one of its 100 ms polls sees the work's end event complete; if such a poll comes while c10d's own stream has joined a capture
    (it does as soon as the captured region issues a collective), HIP answers the event query with hipErrorCapturedEvent -- the
    event's STREAM is capturing, although this event was recorded long before -- and the watchdog aborts the process
    (tools/capture_race.py: async all-gathers in front of a capture die within two iterations; with this pause 60 of 60 pass).
    So: finish the eager work, then give the watchdog time to retire it, and issue no collective in between."""
    import time
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        torch.cuda.synchronize()
        time.sleep(float(os.environ.get("NGACF_CAPTURE_QUIESCE_S", "0.5")))


def _gats(model):
    """Everything in the model that draws dropout seeds (the GAT stacks; the sharded model of ngacf_amd/dist.py)."""
    return [m for m in model.modules() if hasattr(m, "advance_device_seed") and hasattr(m, "use_device_seed")]


def _chain(model, k):
    """The node-partitioned model runs its two propagation chains on two communicators (ShardedSPUIGACF.chain); others: nothing."""
    import contextlib
    return model.chain(k, inline=True) if hasattr(model, "chain") else contextlib.nullcontext()


class GraphedBPRStep:
    """step(users, pos, neg) -> loss (a device scalar, valid until the next call).

    optim must be capturable (torch.optim.Adam(..., capturable=True)); batch size is fixed at capture."""

    def __init__(self, model, lossfn, optim, adj, batch_size, warmup=3, shared_propagation=False, data=None,
                 preserve_state=False, concurrent_forwards=True, fuse_loss_node=False, after_backward=None, split_grads=True):
        """data: a train_eval.Interactions -> the graph also SAMPLES its batch (ngacf_sample_pairs over the rows
        train_u[offset : offset+B], offset and sampler seed advanced by captured adds): use begin_epoch()/replay().
        preserve_state: put parameters and optimizer state back after the warm-up steps (a training loop must not
        see them)."""
        dev = next(model.parameters()).device
        self.model, self.optim = model, optim
        for g in _gats(model):
            g.use_device_seed(dev, True)
        self.data = data
        if data is not None:
            from . import ops
            from .train_eval import SAMPLE_STRIDE
            if data.train_u.numel() < batch_size:
                raise ValueError("GraphedBPRStep samples full batches: %d interaction rows < batch size %d"
                                 % (data.train_u.numel(), batch_size))
            self.offset = torch.zeros(1, dtype=torch.int64, device=dev)       # first row of the batch in train_u
            self.sample_seed = torch.zeros(1, dtype=torch.int64, device=dev)
            self.unresolved = torch.zeros(1, dtype=torch.int32, device=dev)
            self.total = torch.zeros((), device=dev)                          # sum of the batch losses of the epoch
        saved = None
        if preserve_state:
            saved = ([p.detach().clone() for p in model.parameters()],
                     {id(t): t.detach().clone() for st in optim.state.values() for t in st.values() if torch.is_tensor(t)})
        self.upn = torch.zeros(3, batch_size, dtype=torch.long, device=dev)     # users / positives / negatives
        self.u, self.p, self.n = self.upn[0], self.upn[1], self.upn[2]
        model.train()
        gats = _gats(model)
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        from . import functional as NF
        from .BPRLoss import BPRLoss
        from .SPUIGACF import _CFBase
        # Optional: scoring and loss of both branches as one autograd node (same numbers), so that both backward
        # branches start at the same point.  Measured SLOWER (0.97 vs 0.94 ms/step): in lock-step the two branches run
        # their big traversal kernels at the same time and their launch-bound kernels at the same time, while the
        # natural skew of two separate score nodes lets one branch's small kernels hide under the other's traversal.
        fused_loss = (fuse_loss_node and isinstance(lossfn, BPRLoss) and isinstance(model, _CFBase)
                      and type(model).forward is _CFBase.forward)
        # after a step of this kind .grad of the sunk parameters is None (the pieces went straight to the optimizer)
        # (with after_backward -- the node-partitioned model's weight-gradient all-reduce -- only when that hook takes the sinks)
        hook_takes_sinks = after_backward is not None and getattr(getattr(after_backward, "__self__", None), "reduce_grads_takes_sinks", False)
        # (the node-partitioned model's step runs its two chains one after the other -- concurrent branches do not capture there --
        # but parks their gradients the same way: eight accumulation kernels and three loss kernels less on its one queue)
        one_queue = not concurrent_forwards and hasattr(model, "chain")
        split_grads = (split_grads and (concurrent_forwards or one_queue) and not shared_propagation and not fused_loss
                       and getattr(optim, "supports_grad_pairs", False) and (after_backward is None or hook_takes_sinks))
        sinks = ({}, {})
        direct_bpr = type(lossfn) is BPRLoss and split_grads           # this step owns its backward: see the join below
        own_steps = hasattr(optim, "step_counters")        # ngacf_amd.optim.Adam: its step counts join the step's one tick launch
        from .SPUIGACF import _CTR_STRIDE as CTR_STRIDE

        def step():
            if data is not None:                            # the batch itself: rows offset .. offset+B of train_u
                ops.sample_pairs(data.train_u, 0, batch_size, data.train_ptr, data.train_items32, data.pool_items32,
                                 0, out=self.upn, unresolved=self.unresolved, offset_dev=self.offset,
                                 seed_dev=self.sample_seed)
            # the scalar bookkeeping of the step -- fresh dropout masks every replay, the sampler's next rows, Adam's step
            # counts -- as ONE launch (each was a one-element add kernel on the serial part of the step)
            ints = [(g._seed_ctr, CTR_STRIDE) for g in gats if getattr(g, "_seed_ctr", None) is not None]
            if data is not None:
                ints += [(self.offset, batch_size), (self.sample_seed, SAMPLE_STRIDE)]
            NF.ops.tick([t for t, _ in ints], [k for _, k in ints], optim.step_counters() if own_steps else [])
            optim.zero_grad(set_to_none=True)
            if shared_propagation:                          # opt-in: one propagation per step (model.forward_pairs)
                loss = lossfn(*model.forward_pairs(self.u, self.p, self.n, adj))
            elif concurrent_forwards or (one_queue and split_grads):
                # reference semantics (two forwards, two propagations) as two parallel branches of the graph: the
                # chains are independent until the loss, and autograd replays each branch's backward on the stream
                # of its forward, so the many small launch-bound kernels of one chain fill the gaps of the other
                # (one_queue: the same step with both chains on the current stream)
                import contextlib
                cur = torch.cuda.current_stream()
                if concurrent_forwards:
                    sa.wait_stream(cur)
                    sb.wait_stream(cur)
                on_a = (lambda: torch.cuda.stream(sa)) if concurrent_forwards else contextlib.nullcontext
                on_b = (lambda: torch.cuda.stream(sb)) if concurrent_forwards else contextlib.nullcontext
                chain = _chain if concurrent_forwards else (lambda m, k: contextlib.nullcontext())
                if fused_loss:
                    with torch.cuda.stream(sa):
                        fa = model.propagate(adj)
                    with torch.cuda.stream(sb):
                        fb = model.propagate(adj)
                    cur.wait_stream(sa)
                    cur.wait_stream(sb)
                    loss = NF.score_pair_bpr(fa, fb, self.u, self.p, self.n, model.userNum)   # = lossfn(model(u,p), model(u,n))
                else:
                    # split_grads: every branch parks its parameter gradients in its own sink and the optimizer's kernel adds
                    # the two (ngacf_adam_step's grads2) -- autograd would run one accumulation kernel per parameter, eight
                    # launch-bound kernels at the very end of the slower branch
                    # (deferred_ok: this step reads its sinks after the whole backward and after both branch streams were joined)
                    if one_queue and hasattr(model, "forward_two"):
                        # (the node-partitioned model: both propagations in lock-step, one grouped all-gather per stage for the two)
                        ps, ns = model.forward_two(self.u, self.p, self.n, adj, sinks=sinks if split_grads else None)
                    else:
                        with on_a(), NF.grad_sink(sinks[0] if split_grads else None, deferred_ok=True), chain(model, 0):
                            ps = model(self.u, self.p, adj)
                        with on_b(), NF.grad_sink(sinks[1] if split_grads else None, deferred_ok=True), chain(model, 1):
                            ns = model(self.u, self.n, adj)
                    if concurrent_forwards:
                        cur.wait_stream(sa)
                        cur.wait_stream(sb)
                    if direct_bpr:
                        # BPRLoss and the start of its backward as ONE launch: the scores' gradients go straight into the two
                        # branches (the loss node, autograd's ones-like fill and the second BPR kernel sat on the serial join)
                        loss, gp, gn = NF.ops.bpr_fwd_bwd(ps.detach().float().contiguous(), ns.detach().float().contiguous())
                        torch.autograd.backward([ps, ns], [gp.to(ps.dtype), gn.to(ns.dtype)])
                    else:
                        loss = lossfn(ps, ns)
            else:                                           # reference semantics: two forwards, two propagations
                loss = lossfn(model(self.u, self.p, adj), model(self.u, self.n, adj))
            if not (direct_bpr and (concurrent_forwards or one_queue) and not shared_propagation and not fused_loss):
                loss.backward()
            if after_backward is not None and not (sinks[0] or sinks[1]):
                after_backward()                            # e.g. the sharded model's grouped weight-gradient all-reduce
            if sinks[0] or sinks[1]:
                cur = torch.cuda.current_stream()           # the sinks were filled on the branch streams; no AccumulateGrad
                if concurrent_forwards:                     # node synchronised them with this one
                    cur.wait_stream(sa)
                    cur.wait_stream(sb)
                if after_backward is not None:              # (the pieces of both chains in the hook's one grouped launch)
                    after_backward(sinks)
                keys = list(dict.fromkeys(list(sinks[0]) + list(sinks[1])))
                optim.step(grad_pairs={k: (sinks[0].get(k), sinks[1].get(k)) for k in keys}, steps_incremented=own_steps)
                sinks[0].clear()
                sinks[1].clear()
            elif own_steps:
                optim.step(steps_incremented=True)
            else:
                optim.step()
            if data is not None:
                self.total.add_(loss.detach())
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up off the capture: plans, caches, allocator
            for _ in range(warmup):
                if data is not None:
                    self.offset.zero_()                     # every warm-up step samples rows [0, B): n >= B is all it needs
                step()
            if data is not None:
                self.offset.zero_()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optim.zero_grad(set_to_none=True)
        quiesce_collectives()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            self.loss = step()
        if data is not None:                                # (the capture itself does not execute; begin_epoch() resets anyway)
            self.offset.zero_()
            self.unresolved.zero_()
        if saved is not None:                               # undo the warm-up updates, in place (the graph holds these tensors)
            with torch.no_grad():
                for p, v in zip(model.parameters(), saved[0]):
                    p.copy_(v)
                for st in optim.state.values():
                    for t in st.values():
                        if torch.is_tensor(t):
                            t.copy_(saved[1][id(t)]) if id(t) in saved[1] else t.zero_()

    def begin_epoch(self, seed):
        """Sampling mode: start at row 0 of train_u with sampler seed `seed`."""
        self.offset.zero_()
        self.sample_seed.fill_(int(seed))
        self.total.zero_()
        self.unresolved.zero_()

    def replay(self):
        self.graph.replay()
        return self.loss

    def __call__(self, users, pos=None, neg=None):
        """step(users, pos, neg), or step(upn) with the three index rows packed as one [3, B] tensor (one copy)."""
        if pos is None:
            self.upn.copy_(users)
        else:
            self.u.copy_(users)
            self.p.copy_(pos)
            self.n.copy_(neg)
        self.graph.replay()
        return self.loss


class GraphedPropagate:
    """propagate() -> the [N, d] propagated embeddings of the model in eval mode (static output buffer)."""

    def __init__(self, model, adj, warmup=2, stream=None):
        """stream: the capture stream (default: torch's shared one).  Per-stream state of the launches -- the hub rows' arrival
        counters, plan.tickets_for -- is keyed by it: graphs that are to run CONCURRENTLY must be captured on different streams."""
        self.model = model
        model.eval()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                model.propagate(adj)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        quiesce_collectives()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=stream, capture_error_mode=_capture_mode()):
            self.out = model.propagate(adj)

    def __call__(self):
        self.graph.replay()
        return self.out


def graphed_epoch_step(model, lossfn, optim, adj, batch_size, data, after_backward=None):
    """The sampling GraphedBPRStep of (model, optimizer, adjacency, batch size, data), captured once and kept on the
    model (train_bpr(graph=True) asks for it every epoch).  after_backward (the node-partitioned model's gradient
    all-reduce): captured between backward and optimizer; such a step runs its two forwards one after the other."""
    key = (id(optim), adj.data_ptr(), tuple(adj.shape), batch_size, id(data), after_backward is not None)
    cache = model.__dict__.setdefault("_graphed_steps", {})
    if key not in cache:
        cache.clear()
        cache[key] = GraphedBPRStep(model, lossfn, optim, adj, batch_size, data=data, preserve_state=True,
                                    concurrent_forwards=after_backward is None or getattr(model, "concurrent_chains", False),
                                    after_backward=after_backward)
    return cache[key]
