"""Node-partitioned propagation over P GPUs: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI), one all-gather of the transformed table per
propagation stage.  The reference is "replicas only" (parallel.py:94-196: every GPU recomputes all E edges).

Partition: users and items are each cut into P equal blocks (tables padded to P*ceil(U/P) and P*ceil(I/P)
rows with isolated zero nodes), so rank p's rows are contiguous in the global order and the all-gather
lands IN PLACE in the global table -- no re-layout copy, kernels unchanged.  Rank p owns: its embedding rows
(+ their Adam state), the CSR rows of its user block and the CSC rows of its item block.

Per stage, forward:   local transform (window) -> all-gather H -> local pull traversal (source scores are
                      recomputed from the gathered rows, so the score table never travels)
           backward:  local node step -> all-gather Gn, ST -> local edge step -> local transform backward
                      -> all-reduce of the 16 K small-weight gradients (per stage, or once per step with
                      comm.defer_weight_reduce + reduce_grads()).  No reduce-scatter, no atomics.

`backend` is the module that runs the local compute (default: ngacf_amd.ops = the HIP library).  The CPU
tests inject a torch stand-in to exercise partition + collectives under gloo; the product never does.
"""
import math

import torch
import torch.distributed as dist
import torch.nn as nn

from . import functional as NF
from . import ops as _hip_ops
from .plan import build_plan, squeeze_mask
from .SPUIGACF import SpUIGAT, SPUIMultiGAT, StackedSpUIGAT, _fresh_seed, _SEED_STRIDE, _CTR_STRIDE


class Partition:
    def __init__(self, U, I, world, rank):
        self.U, self.I, self.P, self.rank = U, I, world, rank
        self.cu, self.ci = math.ceil(U / world), math.ceil(I / world)
        self.Up, self.Ip = self.cu * world, self.ci * world          # padded table sizes
        self.N = self.Up + self.Ip
        self.u_lo, self.i_lo = rank * self.cu, rank * self.ci
        self.win = (self.u_lo, self.cu, self.i_lo, self.ci)
        self.n_local = self.cu + self.ci

    @property
    def row_ranges(self):
        return [(self.u_lo, self.u_lo + self.cu), (self.Up + self.i_lo, self.Up + self.i_lo + self.ci)]

    def local_rows(self, full):
        """[n_local, ...] copy of this rank's rows of a global-row tensor (users first)."""
        return torch.cat([full[self.u_lo:self.u_lo + self.cu], full[self.Up + self.i_lo:self.Up + self.i_lo + self.ci]])

    def n_real_local(self):
        nu = max(0, min(self.U, self.u_lo + self.cu) - self.u_lo)
        ni = max(0, min(self.I, self.i_lo + self.ci) - self.i_lo)
        return nu, ni


class Comm:
    """The collectives of the path.  In-place all-gather: rank p's block of `full` is already where the
    gathered table wants it.  Under RCCL every call is ONE grouped launch (ncclGroupStart/End through c10d's
    coalescing manager) however many tables and halves it carries: at Gowalla scale a stage is bound by the
    number of collective launches, not by their bytes."""

    def __init__(self, part, group=None):
        self.part, self.group = part, group
        self.world = part.P
        self._base = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self._coalesce = self._base and hasattr(dist.distributed_c10d, "_coalescing_manager")
        self.force = False                    # rehearsal on one rank: issue the collectives even when world == 1
        self.defer_weight_reduce = False      # True: the caller sums weight gradients once per step (reduce_grads)
        self.launches = 0                     # grouped collective launches issued (for the tests / the bench line)

    def _grouped(self, n_ops):
        import contextlib
        self.launches += 1 if (self._coalesce or n_ops == 0) else n_ops
        if self._coalesce and n_ops > 1:
            return dist.distributed_c10d._coalescing_manager(group=self.group)
        return contextlib.nullcontext()

    def all_gather_rows(self, *fulls):
        """Every `full` is a global-row table [Up+Ip, ...] whose local user block and item block are filled."""
        p = self.part
        if self.world == 1 and not self.force:
            return
        pairs = []
        for full in fulls:
            for lo, n, tot in ((0, p.cu, p.Up), (p.Up, p.ci, p.Ip)):
                pairs.append((full[lo:lo + tot], full[lo + p.rank * n: lo + (p.rank + 1) * n], n))
        with self._grouped(len(pairs)):
            for out, mine, n in pairs:
                if self._base:
                    dist.all_gather_into_tensor(out, mine, group=self.group)
                else:                                             # gloo: list form
                    chunks = [out[r * n:(r + 1) * n] for r in range(self.world)]
                    dist.all_gather(chunks, mine.clone(), group=self.group)

    def all_reduce_sum(self, *ts):
        if self.world > 1 or self.force:
            with self._grouped(len(ts)):
                for t in ts:
                    dist.all_reduce(t, group=self.group)
        return ts[0] if len(ts) == 1 else ts


class _ShardedStage(torch.autograd.Function):
    """One propagation stage on this rank's rows.  Input/output are LOCAL ([n_local, d], users first)."""

    @staticmethod
    def forward(ctx, X, W, avec, part, plan, comm, be, heads, alpha, drop_p, feat_drop_p, seed, h_dtype, seed_dev=None, grad_mode=True):
        X = X.contiguous()
        W, avec = W.contiguous().float(), avec.contiguous().float()
        dev, N, dout = X.device, part.N, W.shape[2]
        H = torch.empty(N, dout, dtype=h_dtype, device=dev)
        S = torch.empty(N, heads, dtype=torch.float32, device=dev)
        sd = {} if seed_dev is None else {"seed_dev": seed_dev}        # (the CPU test stand-in has no device seeds)
        be.transform_fwd_rows(X, W, avec, H, S, part.Up, part.Ip, heads, part.win, True, feat_drop_p, seed, **sd)
        comm.all_gather_rows(H)            # S stays local: source scores are recomputed from the gathered rows
        need = grad_mode and any(ctx.needs_input_grad[:3])    # (grad mode is off inside forward(): the caller's comes in)
        res = be.prop_fwd(plan, H, S, S, avec, heads, alpha, True, True, True, drop_p, seed, save=need, **sd)
        out = res[0] if need else res
        if need:
            ctx.save_for_backward(X, W, avec, H, S, out, res[1], res[2])
            ctx.cfg = (part, plan, comm, be, heads, alpha, drop_p, feat_drop_p, seed, sd)
        return part.local_rows(out)

    @staticmethod
    def backward(ctx, g_local):
        X, W, avec, H, S, out, agg, R = ctx.saved_tensors
        part, plan, comm, be, heads, alpha, drop_p, feat_drop_p, seed, sd = ctx.cfg
        dev, N, d = H.device, part.N, H.shape[1]
        g_local = g_local.contiguous().to(H.dtype)
        G = torch.empty(N, d, dtype=torch.float32, device=dev)
        Gn = torch.empty(N, d, dtype=H.dtype, device=dev)
        ST = torch.empty(N, heads, dtype=torch.float32, device=dev)       # t[n,h]
        be.prop_bwd_node(out, agg, R, g_local, S, G, Gn, ST, part.Up, heads, part.win, True, True)
        comm.all_gather_rows(Gn, ST)                                       # one grouped launch
        dH, dS = be.prop_bwd_edges(plan, H, S, avec, G, Gn, ST, heads, alpha, True, True, drop_p, seed, **sd)
        dX, dW, da = be.transform_bwd_rows(X, W, H, dH, dS, part.Up, part.Ip, heads, part.win, True, feat_drop_p, seed, **sd)
        if not comm.defer_weight_reduce:
            comm.all_reduce_sum(dW, da)
        return (dX.to(X.dtype), dW, da) + (None,) * 12


class _GatherRows(torch.autograd.Function):
    """Local rows -> the full table on every rank.  Every rank computes the same loss on the same batch, so
    the gradient of the full table is identical everywhere and its local slice needs no reduction."""

    @staticmethod
    def forward(ctx, local, part, comm):
        full = torch.zeros(part.N, local.shape[1], dtype=local.dtype, device=local.device)
        full[part.u_lo:part.u_lo + part.cu] = local[:part.cu]
        full[part.Up + part.i_lo:part.Up + part.i_lo + part.ci] = local[part.cu:]
        comm.all_gather_rows(full)
        ctx.part = part
        return full

    @staticmethod
    def backward(ctx, g):
        return ctx.part.local_rows(g), None, None


class ShardedSPUIGACF(nn.Module):
    """SPUIGACF / SPUIMultiGACF / SPUIStackedGACF with the embedding tables sharded by node blocks.
    forward(userIdx, itemIdx, mask) has the reference signature (SPUIGACF.py:41) and returns the scores of the
    whole batch on every rank."""

    def __init__(self, userNum, itemNum, embedSize, layers, droprate, world=None, rank=None, stages=2,
                 group=None, backend=None, device=None):
        super().__init__()
        world = dist.get_world_size(group) if world is None else world
        rank = dist.get_rank(group) if rank is None else rank
        self.userNum, self.itemNum, self.droprate = userNum, itemNum, droprate
        self.part = Partition(userNum, itemNum, world, rank)
        self.comm = Comm(self.part, group)
        self.be = backend if backend is not None else _hip_ops
        p = self.part
        self.uEmbd = nn.Embedding(p.cu, embedSize)
        self.iEmbd = nn.Embedding(p.ci, embedSize)
        if stages == 2:
            self.gat = SpUIGAT(embedSize, 8, embedSize, droprate, 0.2, 8)
        elif stages == 3 and embedSize == 64:
            self.gat = SPUIMultiGAT(embedSize, 8, embedSize, droprate, 0.2, 8)
        else:
            self.gat = StackedSpUIGAT(embedSize, 8, embedSize, droprate, 0.2, 8, stages - 1)
        # same values on every rank as the unsharded model with the same seed: draw the full tables, keep a slice
        full_u = torch.empty(userNum, embedSize).normal_(std=0.01)       # SPUIGACF.py:26-28
        full_i = torch.empty(itemNum, embedSize).normal_(std=0.01)
        self.load_full_embeddings(full_u, full_i)
        self._plans = {}
        self.compute_dtype = torch.float32
        self.seg_len = 256

    # ---- parameter plumbing -------------------------------------------------------------------------------
    def load_full_embeddings(self, full_u, full_i):
        p = self.part
        nu, ni = p.n_real_local()
        with torch.no_grad():
            self.uEmbd.weight.zero_()
            self.iEmbd.weight.zero_()
            self.uEmbd.weight[:nu] = full_u[p.u_lo:p.u_lo + nu]
            self.iEmbd.weight[:ni] = full_i[p.i_lo:p.i_lo + ni]

    def load_full_state_dict(self, sd):
        """Take an unsharded (reference-keyed) state_dict: slice the embedding rows, copy the rest."""
        self.load_full_embeddings(sd["uEmbd.weight"], sd["iEmbd.weight"])
        self.gat.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("gat.")})

    def full_state_dict(self):
        """The reference-keyed state_dict on every rank (all-gather of the embedding shards)."""
        p = self.part
        dev = self.uEmbd.weight.device
        tab = torch.zeros(p.N, self.uEmbd.weight.shape[1], device=dev)
        tab[p.u_lo:p.u_lo + p.cu] = self.uEmbd.weight.detach()
        tab[p.Up + p.i_lo:p.Up + p.i_lo + p.ci] = self.iEmbd.weight.detach()
        self.comm.all_gather_rows(tab)
        sd = {"uEmbd.weight": tab[:p.U].clone(), "iEmbd.weight": tab[p.Up:p.Up + p.I].clone()}
        sd.update({"gat." + k: v.detach().clone() for k, v in self.gat.state_dict().items()})
        return sd

    def reduce_grads(self):
        """With comm.defer_weight_reduce: sum the stage-weight gradients over the ranks in ONE grouped all-reduce,
        after backward() and before the optimizer step (embedding rows are owned, their gradients are local)."""
        gs = [p.grad for p in self.gat.parameters() if p.grad is not None]
        if gs:
            self.comm.all_reduce_sum(*gs)

    # ---- forward ------------------------------------------------------------------------------------------
    def _plan(self, mask):
        mask = squeeze_mask(mask)
        key = (mask.data_ptr(), tuple(mask.shape), mask._version, self.seg_len)
        if key not in self._plans:
            p = self.part
            self._plans.clear()
            self._plans[key] = build_plan(mask, p.Up, p.Ip, seg_len=self.seg_len, require_user_edges=False,
                                          row_range=p.row_ranges)
        return self._plans[key]

    def use_device_seed(self, device, on=True):
        """Dropout seeds = per-call host constant + a device counter (advance_device_seed): what a captured hipGraph needs
        to draw fresh masks on every replay (same scheme as SPUIGACF._GATBase)."""
        if on:
            self._seed_base, self._seed_call = _fresh_seed(), 0
            self._seed_ctr = torch.zeros(1, dtype=torch.int64, device=device)
        else:
            self._seed_ctr = None

    def advance_device_seed(self):
        if getattr(self, "_seed_ctr", None) is not None:
            self._seed_ctr.add_(_CTR_STRIDE)

    def propagate_local(self, mask):
        plan = self._plan(mask)
        drop = self.droprate if self.training else 0.0
        seed_dev = None
        if drop > 0 and getattr(self, "_seed_ctr", None) is not None:
            self._seed_call += 1
            base = (self._seed_base + self._seed_call * _SEED_STRIDE) & 0x3FFFFFFFFFFFFFFF
            seed_dev = self._seed_ctr
        else:
            base = _fresh_seed() if drop > 0 else 0                      # same CPU generator state on every rank
        x = torch.cat([self.uEmbd.weight, self.iEmbd.weight], 0)
        for s, layers in enumerate(self.gat._stages()):
            W, avec = self.gat.stage_params(s)
            x = _ShardedStage.apply(x, W, avec, self.part, plan, self.comm, self.be, len(layers), layers[0].alpha,
                                    drop, drop, NF.stage_seed(base, s), self.compute_dtype, seed_dev, torch.is_grad_enabled())
        return x

    def propagate(self, mask):
        """Full [Up+Ip, d] table of propagated embeddings (items start at row Up)."""
        return _GatherRows.apply(self.propagate_local(mask), self.part, self.comm)

    def forward(self, userIdx, itemIdx, mask):
        F = self.propagate(mask)
        return self.be.score(F, userIdx, itemIdx, self.part.Up) if hasattr(self.be, "score") \
            else NF.score(F, userIdx, itemIdx, self.part.Up)


# ---------------------------------------------------------------------------------------------------------
# bench.py --gpus N (N > 1): strong scaling of the named workload over a node partition
# ---------------------------------------------------------------------------------------------------------
def bench_main(args, ng, rank, world, dev, U, I, E_target, d, elem, hbm_peak, emit=print):
    """Strong scaling of the named workload over a node partition.  Order of events: (1) forward timed with eager
    launches, (2) train step timed with eager launches, (3) the forward again as ONE replayed hipGraph (kernels +
    grouped RCCL launches).  (3) replaces (1)'s number only if it completes; a watchdog prints the eager line and
    exits if the captured path stalls, so a capture problem can never cost the measurement."""
    import json
    import os
    import threading
    import time
    from . import data as D
    torch.manual_seed(2019)
    edge = D.synthetic_bipartite(U, I, E_target, seed=2019, device=dev)   # same graph on every rank (same device type)
    E = int(edge.shape[1])
    model = ShardedSPUIGACF(U, I, d, [64, 64], args.droprate, world, rank, stages=args.stages).to(dev)
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.seg_len = args.seg_len
    L = len(model.gat._stages())
    model.comm.force = world == 1                 # one-rank rehearsal (NGACF_BENCH_FORCE_DIST=1): still issue the collectives
    model.comm.defer_weight_reduce = True
    trav = L * 2 * E                              # all ranks together traverse every edge once per forward

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, warmup, steps):
        for _ in range(warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    def line(t_fwd, launch, launches_fwd, extra):
        out = {"metric": "propagation edges/sec (directed edge traversals, forward, all stages)",
               "value": round(trav * args.steps / t_fwd, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * t_fwd / args.steps, 4), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "%s-scale bipartite graph U=%d I=%d E=%d, d=%d, L=%d stages, node-partitioned over %d GPUs"
                                      % (args.workload, U, I, E, d, L, world),
                          "edges_directed_per_step": trav, "launch": launch,
                          "parallelism": "node partition x%d, 1 grouped all-gather launch per stage (%d per forward)"
                                         % (world, launches_fwd)},
               "roofline": None, "cpu_baseline": None}
        out.update(extra)
        return json.dumps(out)

    # ---- (1) forward, eager ------------------------------------------------------------------------------------
    model.eval()
    fwd = lambda: model.propagate_local(edge)
    with torch.no_grad():
        model.comm.launches = 0
        fwd()
        launches_fwd = model.comm.launches
        t_eager = timed(fwd, args.warmup, args.steps)
    extra = {"ms_per_step_eager": round(1e3 * t_eager / args.steps, 4)}

    # ---- (2) train step, eager ---------------------------------------------------------------------------------
    if not args.no_train:
        model.train()
        lossfn = ng.BPRLoss()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        g = torch.Generator(device=dev)
        g.manual_seed(7)                                                  # same batch on every rank
        B = args.batch
        ub = torch.randint(0, U, (B,), device=dev, generator=g)
        pb = torch.randint(0, I, (B,), device=dev, generator=g)
        nb = torch.randint(0, I, (B,), device=dev, generator=g)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = lossfn(model(ub, pb, edge), model(ub, nb, edge))
            loss.backward()
            model.reduce_grads()                                          # one grouped all-reduce of the stage weights
            opt.step()
        ksteps = max(2, min(args.steps, 20))
        model.comm.launches = 0
        step()
        per_step = model.comm.launches
        t_tr = timed(step, min(args.warmup, 3), ksteps)
        extra.update({"bpr_pairs_per_s": round(B * ksteps / t_tr, 1), "train_ms_per_step": round(1e3 * t_tr / ksteps, 3),
                      "train_steps": ksteps, "batch": B, "collective_launches_per_train_step": per_step})
        model.eval()

    eager_line = line(t_eager, "eager", launches_fwd, extra)
    if getattr(args, "eager", False) or os.environ.get("NGACF_DIST_GRAPH") == "0":
        if rank == 0:
            emit(eager_line)
        dist.destroy_process_group()
        return

    # ---- (3) forward as one hipGraph, under a watchdog -----------------------------------------------------------
    lock, done = threading.Lock(), []

    def finish(text, hard):
        with lock:
            if not done:
                done.append(1)
                if rank == 0:
                    emit(text)
        if hard:                                   # a rank may be stuck inside a collective: leave without teardown
            os._exit(3)                            # (non-zero: a stalled capture must not look like a clean run)

    dog = threading.Timer(float(os.environ.get("NGACF_DIST_GRAPH_TIMEOUT", "90")), finish, (eager_line, True))
    dog.daemon = True
    dog.start()
    try:
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd()
            torch.cuda.current_stream().wait_stream(side)
            sync()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fwd()
            t_graph = timed(graph.replay, args.warmup, args.steps)
    except Exception as e:                         # capture refused here (maybe not on the peers): the eager numbers stand
        import sys
        print("[bench] sharded hipGraph capture not used: %r" % (e,), file=sys.stderr, flush=True)
        finish(eager_line, True)
    better = t_graph < t_eager                     # t_* are already max-over-ranks: every rank decides alike
    best = line(t_graph, "hipGraph replay (kernels + grouped RCCL launches)", launches_fwd, extra) if better else eager_line
    # ---- (4) the train step as one hipGraph (sampler-free: fixed batch, as in (2)), same watchdog ------------------
    if not args.no_train:
        dog.cancel()
        dog = threading.Timer(float(os.environ.get("NGACF_DIST_GRAPH_TIMEOUT", "90")), finish, (best, True))
        dog.daemon = True
        dog.start()
        try:
            from .graphs import GraphedBPRStep
            model.train()
            opt_g = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True, fused=True)
            gs = GraphedBPRStep(model, lossfn, opt_g, edge, B, concurrent_forwards=False, after_backward=model.reduce_grads)
            upn = torch.stack([ub, pb, nb])
            t_trg = timed(lambda: gs(upn), min(args.warmup, 3), ksteps)
            if t_trg < t_tr:
                extra = dict(extra, bpr_pairs_per_s=round(B * ksteps / t_trg, 1), train_ms_per_step=round(1e3 * t_trg / ksteps, 3),
                             train_launch="hipGraph replay (kernels + grouped RCCL launches + fused Adam)",
                             train_ms_per_step_eager=round(1e3 * t_tr / ksteps, 3))
                best = line(t_graph, "hipGraph replay (kernels + grouped RCCL launches)", launches_fwd, extra) if better \
                    else line(t_eager, "eager", launches_fwd, extra)
        except Exception as e:
            import sys
            print("[bench] sharded hipGraph train step not used: %r" % (e,), file=sys.stderr, flush=True)
            finish(best, True)
    dog.cancel()
    finish(best, False)
    dist.destroy_process_group()
