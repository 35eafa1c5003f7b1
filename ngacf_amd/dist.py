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
                      -> all-reduce of the 16 K small-weight gradients.  No reduce-scatter, no atomics.

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
from .SPUIGACF import SpUIGAT, SPUIMultiGAT, StackedSpUIGAT, _fresh_seed


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
    gathered table wants it."""

    def __init__(self, part, group=None):
        self.part, self.group = part, group
        self.world = part.P
        self._base = dist.is_initialized() and dist.get_backend(group) == "nccl"

    def all_gather_rows(self, full):
        p = self.part
        if self.world == 1:
            return
        for lo, n, tot in ((0, p.cu, p.Up), (p.Up, p.ci, p.Ip)):
            out = full[lo:lo + tot]
            mine = full[lo + p.rank * n: lo + (p.rank + 1) * n]
            if self._base:
                dist.all_gather_into_tensor(out, mine, group=self.group)
            else:                                                 # gloo: list form
                chunks = [out[r * n:(r + 1) * n] for r in range(self.world)]
                dist.all_gather(chunks, mine.clone(), group=self.group)

    def all_reduce_sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, group=self.group)
        return t


class _ShardedStage(torch.autograd.Function):
    """One propagation stage on this rank's rows.  Input/output are LOCAL ([n_local, d], users first)."""

    @staticmethod
    def forward(ctx, X, W, avec, part, plan, comm, be, heads, alpha, drop_p, feat_drop_p, seed, h_dtype):
        X = X.contiguous()
        W, avec = W.contiguous().float(), avec.contiguous().float()
        dev, N, dout = X.device, part.N, W.shape[2]
        H = torch.empty(N, dout, dtype=h_dtype, device=dev)
        S = torch.empty(N, heads, dtype=torch.float32, device=dev)
        be.transform_fwd_rows(X, W, avec, H, S, part.Up, part.Ip, heads, part.win, True, feat_drop_p, seed)
        comm.all_gather_rows(H)            # S stays local: source scores are recomputed from the gathered rows
        need = any(ctx.needs_input_grad[:3])
        res = be.prop_fwd(plan, H, S, S, avec, heads, alpha, True, True, True, drop_p, seed, save=need)
        out = res[0] if need else res
        if need:
            ctx.save_for_backward(X, W, avec, H, S, out, res[1], res[2])
            ctx.cfg = (part, plan, comm, be, heads, alpha, drop_p, feat_drop_p, seed)
        return part.local_rows(out)

    @staticmethod
    def backward(ctx, g_local):
        X, W, avec, H, S, out, agg, R = ctx.saved_tensors
        part, plan, comm, be, heads, alpha, drop_p, feat_drop_p, seed = ctx.cfg
        dev, N, d = H.device, part.N, H.shape[1]
        g_local = g_local.contiguous().to(H.dtype)
        G = torch.empty(N, d, dtype=torch.float32, device=dev)
        Gn = torch.empty(N, d, dtype=H.dtype, device=dev)
        ST = torch.empty(N, heads, dtype=torch.float32, device=dev)       # t[n,h]
        be.prop_bwd_node(out, agg, R, g_local, S, G, Gn, ST, part.Up, heads, part.win, True, True)
        comm.all_gather_rows(Gn)
        comm.all_gather_rows(ST)
        dH, dS = be.prop_bwd_edges(plan, H, S, avec, G, Gn, ST, heads, alpha, True, True, drop_p, seed)
        dX, dW, da = be.transform_bwd_rows(X, W, H, dH, dS, part.Up, part.Ip, heads, part.win, True, feat_drop_p, seed)
        comm.all_reduce_sum(dW)
        comm.all_reduce_sum(da)
        return (dX.to(X.dtype), dW, da) + (None,) * 10


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

    def propagate_local(self, mask):
        plan = self._plan(mask)
        drop = self.droprate if self.training else 0.0
        base = _fresh_seed() if drop > 0 else 0                          # same CPU generator state on every rank
        x = torch.cat([self.uEmbd.weight, self.iEmbd.weight], 0)
        for s, layers in enumerate(self.gat._stages()):
            W, avec = self.gat.stage_params(s)
            x = _ShardedStage.apply(x, W, avec, self.part, plan, self.comm, self.be, len(layers), layers[0].alpha,
                                    drop, drop, NF.stage_seed(base, s), self.compute_dtype)
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
def bench_main(args, ng, rank, world, dev, U, I, E_target, d, elem, hbm_peak):
    import json
    import time
    from . import data as D
    torch.manual_seed(2019)
    edge = D.synthetic_bipartite(U, I, E_target, seed=2019, device=dev)   # same graph on every rank (same device type)
    E = int(edge.shape[1])
    model = ShardedSPUIGACF(U, I, d, [64, 64], args.droprate, world, rank, stages=args.stages).to(dev)
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model.seg_len = args.seg_len
    L = len(model.gat._stages())

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    model.eval()
    with torch.no_grad():
        for _ in range(args.warmup):
            model.propagate_local(edge)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.propagate_local(edge)
        sync()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_fwd = float(t)
    trav = L * 2 * E                                                     # all ranks together traverse every edge once
    out = {"metric": "propagation edges/sec (directed edge traversals, forward, all stages)",
           "value": round(trav * args.steps / t_fwd, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(1e3 * t_fwd / args.steps, 4), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "%s-scale bipartite graph U=%d I=%d E=%d, d=%d, L=%d stages, node-partitioned over %d GPUs"
                                  % (args.workload, U, I, E, d, L, world),
                      "edges_directed_per_step": trav, "parallelism": "node partition x%d, 1 all-gather per stage" % world},
           "roofline": None, "cpu_baseline": None}
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
            opt.step()
        ksteps = max(2, min(args.steps, 20))
        for _ in range(min(args.warmup, 3)):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(ksteps):
            step()
        sync()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out.update({"bpr_pairs_per_s": round(B * ksteps / float(t), 1), "train_ms_per_step": round(1e3 * float(t) / ksteps, 3),
                    "train_steps": ksteps, "batch": B})
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()
