"""Node-partitioned propagation over P GPUs: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI), the transformed table all-gathered once per propagation stage -- hidden behind the traversal.
The reference is "replicas only" (parallel.py:94-196: every GPU recomputes all E edges).

Partition.  Nodes are renumbered once, at ingest: users sorted by degree are dealt to the P ranks in snake order
(likewise items), internal id = rank * block + slot.  Every rank then owns a CONTIGUOUS block of cu = ceil(U/P) user
rows and ci = ceil(I/P) item rows of the internal numbering, the blocks are equal in rows (the all-gather lands IN
PLACE in the global-row table, no padding beyond U mod P) AND balanced in nnz (hub rows are spread round-robin) whatever
the id order of the data set.  Rank p owns its embedding rows (+ Adam state), the CSR rows of its user block and the
CSC rows of its item block -- and only those edges: the plan is built from the ~2E/P directed edges that end in the
rank's rows (ngacf_amd.plan.build_plan_from_keys); no rank sorts or stores the global edge list.

Schedule (one autograd node for the whole stack, `_ShardedStack`).  A traversal of the user rows reads only ITEM rows
of H (and vice versa), so a stage is two half-launches with different needs:

    T(s,t)  transform the local rows of type t for stage s, start the all-gather of that half of H_s   (async)
    V(s,A)  wait for the all-gather of the OTHER type, traverse the local dst rows of type A

  forward   T(0,B) T(0,A) | V(s,A) T(s+1,A) V(s,B) T(s+1,B) | ...   with A alternating user/item from stage to stage:
            the all-gather that V(s,B) waits for runs under V(s,A), the one V(s+1,B) needs (started by T(s+1,A)) runs
            under V(s,B) -- only the very first all-gather of a forward is exposed.
  backward  the mirror image with the node step N(s,t) (-> Gn and (S, t) tables, all-gathered per half), the edge step
            E(s,A) and the transform backward per half; dX of a half is the next stage's incoming gradient, so
            N(s-1,A) and its all-gather start while E(s,B) still runs.

The collectives are issued with async_op=True (c10d runs them on the communicator's own stream; wait() is a stream
wait, not a host wait), two tables of one half as ONE grouped launch.  `overlap=False` waits right after every start:
the serial schedule of round 1, kept for A/B timing.  Scoring exchanges 2B rows (one small all-reduce per forward),
not the [N,d] table.  No reduce-scatter, no atomics; results are bitwise independent of the schedule.

Small tables (Comm.whole_below: a stage's table brings a rank < 32 MiB) take the WHOLE-STAGE schedule instead: T(s, both halves),
one grouped all-gather of the stage's table, one traversal launch over all local rows (a gather is latency there, not bytes: hiding
it behind a half-launch buys nothing; two collective launches and two launches' ramps per stage cost).  Same kernels, same bits.

`backend` is the module that runs the local compute (default: ngacf_amd.ops = the HIP library).  The CPU tests inject a
torch stand-in to exercise partition + collectives + schedule under gloo; the product never does.
"""
import contextlib
import math
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import functional as NF
from . import ops as _hip_ops
from .plan import build_plan, build_plan_from_keys, squeeze_mask
from .SPUIGACF import SpUIGAT, SPUIMultiGAT, StackedSpUIGAT, _PlanCache, _fresh_seed, _SEED_STRIDE, _CTR_STRIDE

OTHER = {"user": "item", "item": "user"}


def _deal(deg, P, block):
    """original id -> internal id: ids sorted by degree (descending, stable) are dealt to the P ranks in snake order."""
    n = int(deg.numel())
    order = torch.sort(deg, descending=True, stable=True).indices
    j = torch.arange(n, device=deg.device)
    t, k = j // P, j % P
    r = torch.where(t % 2 == 0, k, P - 1 - k)
    perm = torch.empty(n, dtype=torch.long, device=deg.device)
    perm[order] = r * block + t
    return perm


class Partition:
    """Equal blocks of the INTERNAL numbering.  deg_u / deg_i given: degree-balanced renumbering (see the module
    docstring); without them the internal numbering is the data set's own (blocks of consecutive ids)."""

    def __init__(self, U, I, world, rank, deg_u=None, deg_i=None):
        self.U, self.I, self.P, self.rank = U, I, world, rank
        self.cu, self.ci = math.ceil(U / world), math.ceil(I / world)
        self.Up, self.Ip = self.cu * world, self.ci * world          # padded table sizes
        self.N = self.Up + self.Ip
        self.u_lo, self.i_lo = rank * self.cu, rank * self.ci
        self.win = (self.u_lo, self.cu, self.i_lo, self.ci)
        self.n_local = self.cu + self.ci
        self.perm_u = _deal(deg_u.long(), world, self.cu) if deg_u is not None else None
        self.perm_i = _deal(deg_i.long(), world, self.ci) if deg_i is not None else None
        self._cache = {}

    # ---- numbering ---------------------------------------------------------------------------------------------
    def _on(self, name, dev):
        key = (name, str(dev))
        if key not in self._cache:
            perm = getattr(self, "perm_" + name[-1])
            n, npad = (self.U, self.Up) if name[-1] == "u" else (self.I, self.Ip)
            if name.startswith("perm"):
                t = perm.to(dev) if perm is not None else torch.arange(n, device=dev)
            else:                                           # inverse: internal -> original, -1 on the padding slots
                t = torch.full((npad,), -1, dtype=torch.long, device=dev)
                p = perm.to(dev) if perm is not None else torch.arange(n, device=dev)
                t[p] = torch.arange(n, device=dev)
            self._cache[key] = t
        return self._cache[key]

    def internal_users(self, idx):
        return idx if self.perm_u is None else self._on("perm_u", idx.device)[idx]

    def internal_items(self, idx):
        return idx if self.perm_i is None else self._on("perm_i", idx.device)[idx]

    def map_edges(self, edge):
        """[2,E] (user id, item id) of the data set -> internal numbering."""
        if self.perm_u is None and self.perm_i is None:
            return edge
        return torch.stack([self.internal_users(edge[0]), self.internal_items(edge[1])])

    def local_original_ids(self, dev="cpu"):
        """(user ids, item ids) of the data set behind this rank's local rows, -1 on padding slots."""
        return (self._on("inv_u", dev)[self.u_lo:self.u_lo + self.cu], self._on("inv_i", dev)[self.i_lo:self.i_lo + self.ci])

    def to_original(self, tab):
        """[Up+Ip, ...] internal-row table -> [U+I, ...] in the data set's numbering."""
        return torch.cat([tab[:self.Up][self._on("perm_u", tab.device)], tab[self.Up:][self._on("perm_i", tab.device)]])

    # ---- rows --------------------------------------------------------------------------------------------------
    @property
    def row_ranges(self):
        return [(self.u_lo, self.u_lo + self.cu), (self.Up + self.i_lo, self.Up + self.i_lo + self.ci)]

    def half_win(self, t):
        return (self.u_lo, self.cu, 0, 0) if t == "user" else (0, 0, self.i_lo, self.ci)

    def block(self, full, t):
        """this rank's rows of type t of a global-row table (a contiguous view)"""
        return full[self.u_lo:self.u_lo + self.cu] if t == "user" else full[self.Up + self.i_lo:self.Up + self.i_lo + self.ci]

    def local_rows(self, full):
        """[n_local, ...] copy of this rank's rows of a global-row tensor (users first)."""
        return torch.cat([self.block(full, "user"), self.block(full, "item")])


class _Handle:
    def __init__(self, works=(), cm=None, done=None):
        self.works, self.cm, self.done = list(works), cm, done

    def wait(self):
        if self.cm is not None:
            self.cm.wait()
        for w in self.works:
            w.wait()
        self.works, self.cm = [], None
        if self.done is not None:
            self.done()
            self.done = None


class Comm:
    """The collectives of the path.  In-place all-gather: rank p's block of `full` is already where the gathered table
    wants it.  Under RCCL the tables of one call travel as ONE grouped launch (ncclGroupStart/End through c10d's
    coalescing manager): at Gowalla scale a stage is bound by the number of collective launches, not by their bytes."""

    def __init__(self, part, group=None):
        self.part, self.group = part, group
        self.world = part.P
        self._base = dist.is_initialized() and dist.get_backend(group) == "nccl"
        self._coalesce = self._base and hasattr(dist.distributed_c10d, "_coalescing_manager")
        self.force = False                    # rehearsal on one rank: issue the collectives even when world == 1
        self.defer_weight_reduce = False      # True: the caller sums weight gradients once per step (reduce_grads)
        self.overlap = True                   # False: wait right after every start (serial schedule)
        self.launches = 0                     # collective launches issued (for the tests / the bench line)
        self.whole_gathers = 0                # ... of them whole-table gathers (start_gather2: the whole-stage schedule)
        self.timer = None                     # list -> (start event, end event, bytes received) per gather (bench, serial mode)
        self.dry_run = False                  # tools/shard_compute.py: one rank's kernels of a P-way partition, no collectives
        self.inline = False                   # True: every gather synchronously in the caller's stream order (start_gather)
        # A gather that brings this rank fewer bytes than this goes in the caller's stream order too: an asynchronous launch hides
        # the transfer behind the other half's traversal but pays a fork and a join (~10 us each inside a replayed graph) -- about
        # what 4 MiB take at the ~300 GB/s a rank receives over xGMI.  Gowalla-scale half tables stay below it; the 10^9-edge graph's
        # 240 MB halves do not.  (NGACF_DIST_INLINE_BYTES=0: always asynchronous, the schedule of rounds 1-4.)
        self.inline_below = int(os.environ.get("NGACF_DIST_INLINE_BYTES", str(4 << 20)))
        # A stage whose whole table brings this rank fewer bytes than this runs as ONE collective + ONE traversal launch (the
        # whole-stage schedule of _ShardedStack) instead of two half-table gathers hidden behind two half-launches: where a gather is
        # latency, not bytes (Gowalla / Yelp-scale tables: 9-36 MB in all), hiding it behind a 5-25 us half-launch buys nothing and the
        # second collective launch and the second launch's ramp and tail are pure cost (one RCCL rank, Gowalla scale: forward 0.120
        # -> see DESIGN.md section 6).  The 10^9-edge graph's 1.9 GB tables stay on the pipelined halves.  0: never.
        self.whole_below = int(os.environ.get("NGACF_DIST_WHOLE_BYTES", str(32 << 20)))

    peer_store = False                        # (ngacf_amd.peer.PeerComm: gathers by stores into the peers' tables)

    def twin(self):
        """A SECOND communicator over the same ranks (its own process group = its own RCCL communicator and its own c10d stream),
        with this one's switches: the two propagation chains of a train step run as concurrent branches, each on its own
        communicator -- collectives of different communicators share nothing, so the order in which the GPU picks the two
        branches' kernels need not agree between the ranks (one communicator driven from two streams would need exactly that).
        dist.new_group is itself a collective call: every rank creates the twin at the same point (ShardedSPUIGACF.__init__)."""
        if not dist.is_initialized():
            return self
        ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
        t = Comm(self.part, dist.new_group(ranks=ranks, backend=dist.get_backend(self.group)))
        t._primary = self
        return t

    _primary = None
    _MIRRORED = ("force", "defer_weight_reduce", "overlap", "dry_run", "inline_below", "whole_below")

    def sync_switches(self):
        """(a twin takes the switches of its primary at every use: bench.py and the tests set them on model.comm after construction)"""
        if self._primary is not None:
            for k in self._MIRRORED:
                setattr(self, k, getattr(self._primary, k))
        return self

    @property
    def active(self):
        return (self.world > 1 or self.force) and not self.dry_run

    # Gathered tables are asked for here so that a transport that needs symmetric memory can hand out its own
    # (PeerComm); this one has no such need: plain tensors, nothing to lease, nothing to join, nothing to check.
    def new_lease(self):
        return None

    def table(self, key, shape, dtype, device, lease=None):
        return torch.empty(*shape, dtype=dtype, device=device)

    def join(self):
        pass

    def check(self):
        pass

    def _region(self, full, t):
        p = self.part
        lo, n, tot = (0, p.cu, p.Up) if t == "user" else (p.Up, p.ci, p.Ip)
        return full[lo:lo + tot], full[lo + p.rank * n: lo + (p.rank + 1) * n], n

    def whole_stage(self, table_bytes):
        """Should a stage whose global-row table(s) hold `table_bytes` run as one collective + one launch?  (Comm.whole_below)"""
        return (not self.peer_store) and table_bytes * (self.world - 1) // max(self.world, 1) < self.whole_below

    def start_gather(self, t, *fulls):
        """All-gather the half `t` ('user' / 'item') of every global-row table in `fulls`; returns a handle to wait on."""
        if not self.active:
            return _Handle()
        return self._gather([self._region(f, t) for f in fulls], fulls[0].is_cuda)

    def start_gather2(self, *fulls):
        """BOTH halves of every table in `fulls` as one grouped launch (the whole-stage schedule)."""
        if not self.active:
            return _Handle()
        self.whole_gathers += 1
        return self._gather([self._region(f, t) for f in fulls for t in ("user", "item")], fulls[0].is_cuda, inline=True)

    def _gather(self, regions, on_gpu, inline=False):
        """inline: in the caller's stream order whatever the size (the whole-stage schedule has nothing to run beside its gather)"""
        ev = None
        if self.timer is not None and on_gpu:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        self.launches += 1 if (self._coalesce or len(regions) == 1) else len(regions)
        recv = sum(o.numel() * o.element_size() for o, _, _ in regions) * (self.world - 1) // max(self.world, 1)
        if inline or self.inline or (self.overlap and recv < self.inline_below):
            # IN the caller's stream order (c10d runs a synchronous collective on the current stream): no communicator-side stream,
            # no fork / join around the launch.  The graphed train step's two concurrent chains use this form -- the OTHER chain's
            # work is what hides a gather then -- because a fork / join nested inside a forked branch crashes hipGraph
            # instantiation on this runtime (segfault in capture_end: tools/graph_topology_probe.py).
            if self._base:
                ctx = dist.distributed_c10d._coalescing_manager(group=self.group) if (self._coalesce and len(regions) > 1) \
                    else contextlib.nullcontext()
                with ctx:
                    for out, mine, _ in regions:
                        dist.all_gather_into_tensor(out, mine, group=self.group)
            else:
                for out, mine, n in regions:
                    dist.all_gather([out[r * n:(r + 1) * n] for r in range(self.world)], mine.clone(), group=self.group)
            h = _Handle()
        elif self._base:
            if self._coalesce and len(regions) > 1:
                with dist.distributed_c10d._coalescing_manager(group=self.group, async_ops=True) as cm:
                    for out, mine, _ in regions:
                        dist.all_gather_into_tensor(out, mine, group=self.group)
                h = _Handle(cm=cm)
            else:
                h = _Handle([dist.all_gather_into_tensor(out, mine, group=self.group, async_op=True) for out, mine, _ in regions])
        else:                                             # gloo: list form
            works = []
            for out, mine, n in regions:
                chunks = [out[r * n:(r + 1) * n] for r in range(self.world)]
                works.append(dist.all_gather(chunks, mine.clone(), group=self.group, async_op=True))
            h = _Handle(works)
        if ev is not None:
            nbytes = sum(o.numel() * o.element_size() for o, _, _ in regions) * (self.world - 1) // max(self.world, 1)

            def done(ev=ev, nbytes=nbytes):
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.timer.append((ev, e1, nbytes))
            h.done = done
        if not self.overlap:
            h.wait()
        return h

    def all_gather_rows(self, *fulls):
        """Both halves of every table, synchronous (state_dict / evaluation plumbing, not the hot path)."""
        for t in ("user", "item"):
            self.start_gather(t, *fulls).wait()

    def all_reduce_sum(self, *ts):
        if self.active:
            self.launches += 1 if (self._coalesce or len(ts) == 1) else len(ts)
            ctx = dist.distributed_c10d._coalescing_manager(group=self.group) if (self._coalesce and len(ts) > 1) \
                else contextlib.nullcontext()
            with ctx:
                for t in ts:
                    dist.all_reduce(t, group=self.group)
        return ts[0] if len(ts) == 1 else ts


class _ShardedStack(torch.autograd.Function):
    """All propagation stages on this rank's rows, pipelined against the all-gathers (module docstring).
    apply(Xu, Xi, W_0, a_0, ..., W_{L-1}, a_{L-1}, cfg) -> (out_u [cu,d], out_i [ci,d]) of the last stage."""

    @staticmethod
    def forward(ctx, Xu, Xi, *rest):
        cfg = rest[-1]
        params = rest[:-1]
        part, plan, comm, be = cfg["part"], cfg["plan"], cfg["comm"], cfg["be"]
        L = len(params) // 2
        dev = Xu.device
        N = part.N
        need = cfg["grad_mode"] and any(ctx.needs_input_grad[:2 + 2 * L])
        X = {"user": Xu.contiguous(), "item": Xi.contiguous()}
        sd = {} if cfg["seed_dev"] is None else {"seed_dev": cfg["seed_dev"]}     # (the CPU stand-in has no device seeds)
        Ws = [params[2 * s].contiguous().float() for s in range(L)]
        As = [params[2 * s + 1].contiguous().float() for s in range(L)]
        saved, pending = [], {}
        Hs, Ss = [None] * L, [None] * L
        if comm.peer_store and comm.active and L < 2:
            raise ValueError("peer-store all-gather needs a stack of at least two stages (ngacf_amd/peer.py)")
        lease = comm.new_lease()

        def new_H(s):
            dout = Ws[s].shape[2]
            Hs[s] = comm.table(("H", s), (N, dout), cfg["h_dtype"], dev, lease)
            Ss[s] = torch.empty(N, cfg["heads"][s], dtype=torch.float32, device=dev)

        def T(s, t):
            if Hs[s] is None:
                new_H(s)
            be.transform_fwd_rows(X[t], Ws[s], As[s], Hs[s], Ss[s], part.Up, part.Ip, cfg["heads"][s], part.half_win(t), True,
                                  cfg["feat_drop"], cfg["seeds"][s], **sd)
            pending[(s, t)] = comm.start_gather(t, Hs[s])

        # Stage s+1's transform of a half rides in stage s's half-launch traversal where the library can do that
        # (ngacf_prop_fwd_next: rows of 64 columns in 8 heads, hub rows finished in-kernel): V(s,A) and T(s+1,A) become ONE launch,
        # four launches less per three-stage forward.  bf16 panels are packed by one small launch in front; fp32 takes the weights.
        fuse = [False] * L
        if cfg.get("fuse_next") and hasattr(be, "can_fuse_next"):
            for s in range(1, L):
                fuse[s] = (be.can_fuse_next(plan, cfg["h_dtype"], Ws[s - 1].shape[2], cfg["heads"][s - 1])
                           and Ws[s].shape[1] == 64 and Ws[s].shape[2] in (64, 128))
        panels = {}
        if any(fuse):
            idx = [s for s in range(1, L) if fuse[s]]
            if cfg["h_dtype"] == torch.bfloat16:
                panels = dict(zip(idx, be.pack_panels([Ws[s] for s in idx])))
            else:
                panels = {s: Ws[s] for s in idx}

        # Whole-stage schedule (Comm.whole_below): T(s, both halves), ONE grouped all-gather of the stage's table, ONE traversal launch
        # over all local rows.  Same kernels on the same rows as the half schedule: the same bits.
        whole = bool(cfg.get("whole"))
        out = None
        xflat = cfg.get("x_flat")
        if xflat is not None and not (xflat.data_ptr() == Xu.data_ptr() and Xu.is_contiguous() and Xi.is_contiguous()
                                      and Xi.data_ptr() == xflat.data_ptr() + Xu.numel() * Xu.element_size()):
            xflat = None                                             # (the inputs are not the model's linked tables)
        if whole:
            new_H(0)
            if xflat is not None:
                be.transform_fwd_rows(xflat, Ws[0], As[0], Hs[0], Ss[0], part.Up, part.Ip, cfg["heads"][0], part.win, True,
                                      cfg["feat_drop"], cfg["seeds"][0], **sd)
            else:
                for t in ("user", "item"):
                    be.transform_fwd_rows(X[t], Ws[0], As[0], Hs[0], Ss[0], part.Up, part.Ip, cfg["heads"][0], part.half_win(t), True,
                                          cfg["feat_drop"], cfg["seeds"][0], **sd)
            h = comm.start_gather2(Hs[0])
            for s in range(L):
                H, S, heads = Hs[s], Ss[s], cfg["heads"][s]
                dout = H.shape[1]
                fused_next = s + 1 < L and fuse[s + 1]
                keep_out = need or not fused_next
                out = torch.empty(N, dout, dtype=H.dtype, device=dev) if keep_out else None
                agg = torch.empty(2, N, dout, dtype=H.dtype, device=dev) if need else None
                R = torch.empty(2, N, heads, dtype=torch.float32, device=dev) if need else None
                if s + 1 < L and Hs[s + 1] is None:
                    new_H(s + 1)
                Xs = dict(X)
                h.wait()
                if fused_next:
                    nxt = be.NextStage(panels[s + 1], As[s + 1], cfg["heads"][s + 1], cfg["feat_drop"], cfg["seeds"][s + 1],
                                       H=Hs[s + 1], S=Ss[s + 1])
                    be.prop_fwd(plan, H, S, S, As[s], heads, cfg["alpha"], True, True, True, cfg["drop"], cfg["seeds"][s],
                                save=need, out=out, agg=agg, R=R, next=nxt, want_out=keep_out, **sd)
                else:
                    be.prop_fwd(plan, H, S, S, As[s], heads, cfg["alpha"], True, True, True, cfg["drop"], cfg["seeds"][s],
                                save=need, out=out, agg=agg, R=R, **sd)
                if keep_out:
                    X = {t: part.block(out, t) for t in ("user", "item")}
                if s + 1 < L:
                    if not fused_next:
                        for t in ("user", "item"):
                            be.transform_fwd_rows(X[t], Ws[s + 1], As[s + 1], Hs[s + 1], Ss[s + 1], part.Up, part.Ip, cfg["heads"][s + 1],
                                                  part.half_win(t), True, cfg["feat_drop"], cfg["seeds"][s + 1], **sd)
                    h = comm.start_gather2(Hs[s + 1])
                if need:
                    saved.append((Xs["user"], Xs["item"], Ws[s], As[s], H, S, out, agg, R))
        first = "user"
        if not whole:
            T(0, OTHER[first])
            T(0, first)
        for s in (range(L) if not whole else ()):                    # the pipelined half-table schedule
            A, B = first, OTHER[first]
            H, S, heads = Hs[s], Ss[s], cfg["heads"][s]
            dout = H.shape[1]
            fused_next = s + 1 < L and fuse[s + 1]
            keep_out = need or not fused_next                        # an evaluation forward never reads a fused stage's own table
            out = torch.empty(N, dout, dtype=H.dtype, device=dev) if keep_out else None
            agg = torch.empty(2, N, dout, dtype=H.dtype, device=dev) if need else None       # (two planes: ops.prop_fwd)
            R = torch.empty(2, N, heads, dtype=torch.float32, device=dev) if need else None
            if fused_next and Hs[s + 1] is None:
                new_H(s + 1)
            Xs = dict(X)
            for t in (A, B):
                pending.pop((s, OTHER[t])).wait()                    # sources of the dst half t
                if fused_next:
                    nxt = be.NextStage(panels[s + 1], As[s + 1], cfg["heads"][s + 1], cfg["feat_drop"], cfg["seeds"][s + 1],
                                       H=Hs[s + 1], S=Ss[s + 1])
                    be.prop_fwd(plan, H, S, S, As[s], heads, cfg["alpha"], True, True, True, cfg["drop"], cfg["seeds"][s],
                                save=need, out=out, half=t, agg=agg, R=R, next=nxt, want_out=keep_out, **sd)
                    if keep_out:
                        X[t] = part.block(out, t)
                    pending[(s + 1, t)] = comm.start_gather(t, Hs[s + 1])
                else:
                    be.prop_fwd(plan, H, S, S, As[s], heads, cfg["alpha"], True, True, True, cfg["drop"], cfg["seeds"][s],
                                save=need, out=out, half=t, agg=agg, R=R, **sd)
                    if s + 1 < L:
                        X[t] = part.block(out, t)
                        T(s + 1, t)
            if need:
                saved.append((Xs["user"], Xs["item"], Ws[s], As[s], H, S, out, agg, R))
            first = B                                                # the next stage starts with the other half
        comm.join()
        if need:
            ctx.xflat = xflat if whole else None
            ctx.cfg, ctx.L = cfg, L
            ctx.saved_stages = saved                                 # (held on ctx: views of the stage tables)
            ctx.lease = lease                                        # (the H tables go back to the pool with the node)
        elif lease is not None:
            lease.release()
        return part.block(out, "user"), part.block(out, "item")

    @staticmethod
    def backward(ctx, gU, gI):
        cfg, L = ctx.cfg, ctx.L
        part, plan, comm, be = cfg["part"], cfg["plan"], cfg["comm"], cfg["be"]
        sd = {} if cfg["seed_dev"] is None else {"seed_dev": cfg["seed_dev"]}
        N = part.N
        saved = ctx.saved_stages
        h_dtype = saved[0][4].dtype
        g = {"user": gU.contiguous().to(h_dtype), "item": gI.contiguous().to(h_dtype)}
        dev = gU.device
        tabs, pending = [None] * L, {}
        lease = comm.new_lease()

        whole = bool(cfg.get("whole"))
        zeroed = None
        if whole:
            # every stage's dH and dS out of ONE zeroed buffer (one fill per backward instead of two per stage: a launch is ~5 us on
            # the step's one queue, where the whole-stage schedule runs)
            sizes = []
            for s in range(L):
                d, heads = saved[s][4].shape[1], cfg["heads"][s]
                sizes += [(N * d * saved[s][4].element_size() + 255) // 256 * 256, (N * heads * 4 + 255) // 256 * 256]
            zeroed, zoff = torch.zeros(sum(sizes), dtype=torch.uint8, device=dev), [0]
            for z in sizes:
                zoff.append(zoff[-1] + z)

        def alloc(s):
            H = saved[s][4]
            if tabs[s] is None:
                d, heads = H.shape[1], cfg["heads"][s]
                if zeroed is not None:
                    dH0 = zeroed[zoff[2 * s]:zoff[2 * s] + N * d * H.element_size()].view(H.dtype).view(N, d)
                    dS0 = zeroed[zoff[2 * s + 1]:zoff[2 * s + 1] + N * heads * 4].view(torch.float32).view(N, heads)
                else:
                    dH0, dS0 = torch.zeros(N, d, dtype=H.dtype, device=dev), torch.zeros(N, heads, dtype=torch.float32, device=dev)
                tabs[s] = (torch.empty(N, d, dtype=torch.float32, device=dev), comm.table(("Gn", s), (N, d), H.dtype, dev, lease),
                           comm.table(("t", s), (N, 2 * heads), torch.float32, dev, lease),     # pairs (S, t) per head
                           dH0, dS0)
            return tabs[s]

        def Nstep(s, t, gather=True):
            _, _, _, _, H, S, out, agg, R = saved[s]
            G, Gn, ST, _, dS = alloc(s)
            # (step 1 leaves the rows' own half of dS -- P1 -- in dS; the edge step of the same half finishes it in place)
            be.prop_bwd_node(out, agg, R, g[t], S, G, Gn, ST, dS, part.Up, cfg["heads"][s], part.half_win(t), True, True, cfg["alpha"])
            if gather:
                pending[(s, t)] = comm.start_gather(t, Gn, ST)      # one grouped launch

        def Nboth(s, gfull=None, glocal=None):
            """whole-stage schedule: the node step of all local rows, ONE gather.  gfull: the incoming gradient as a global-row
            table (what the next stage's transform backward leaves), glocal: as one [cu + ci, d] table of the local rows (what the
            scores' row exchange leaves) -> one launch; neither: g's two local halves -> two."""
            if gfull is None and glocal is None:
                Nstep(s, "user", False)
                Nstep(s, "item", False)
            else:
                _, _, _, _, H, S, out, agg, R = saved[s]
                G, Gn, ST, _, dS = alloc(s)
                be.prop_bwd_node(out, agg, R, gfull if gfull is not None else glocal, S, G, Gn, ST, dS, part.Up, cfg["heads"][s],
                                 part.win, gfull is None, True, cfg["alpha"])
            pending[(s, "both")] = comm.start_gather2(tabs[s][1], tabs[s][2])

        first = "user"
        grads = [None] * (2 * L)
        if whole:
            base = g["user"]._base                                      # the two incoming halves as views of ONE table?
            if not (base is not None and base is g["item"]._base and base.dim() == 2 and base.is_contiguous() and base.dtype == h_dtype
                    and base.shape[0] == part.cu + part.ci and g["user"].data_ptr() == base.data_ptr()
                    and g["item"].data_ptr() == base.data_ptr() + part.cu * base.shape[1] * base.element_size()):
                base = None
            Nboth(L - 1, glocal=base)
            for s in range(L - 1, -1, -1):
                Xu, Xi, W, a, H, S, out, agg, R = saved[s]
                G, Gn, ST, dH, dS = tabs[s]
                heads = cfg["heads"][s]
                pending.pop((s, "both")).wait()
                be.prop_bwd_edges(plan, H, S, a, G, Gn, ST, heads, cfg["alpha"], True, True, cfg["drop"], cfg["seeds"][s],
                                  dH=dH, dS=dS, **sd)
                if s > 0:
                    # this stage's input rows ARE the previous stage's output table (global rows): the transform backward of all local
                    # rows as one launch, its dX a global-row table that the previous stage's node step reads in place
                    dX, dW, da = be.transform_bwd_rows(saved[s - 1][6], W, H, dH, dS, part.Up, part.Ip, heads, part.win, False,
                                                       cfg["feat_drop"], cfg["seeds"][s], **sd)
                    Nboth(s - 1, dX if dX.dtype == h_dtype else dX.to(h_dtype))
                elif getattr(ctx, "xflat", None) is not None:
                    dX, dW, da = be.transform_bwd_rows(ctx.xflat, W, H, dH, dS, part.Up, part.Ip, heads, part.win, True,
                                                       cfg["feat_drop"], cfg["seeds"][s], **sd)
                    g["user"], g["item"] = dX[:part.cu], dX[part.cu:]
                else:
                    parts = {}
                    for t, Xt in (("user", Xu), ("item", Xi)):
                        dX, dW_, da_ = be.transform_bwd_rows(Xt, W, H, dH, dS, part.Up, part.Ip, heads, part.half_win(t), True,
                                                             cfg["feat_drop"], cfg["seeds"][s], **sd)
                        parts[t] = (dW_, da_)
                        g[t] = dX
                    dW = torch.stack([parts["user"][0][0], parts["item"][0][1]])
                    da = torch.stack([parts["user"][1][0], parts["item"][1][1]])
                if not comm.defer_weight_reduce:
                    comm.all_reduce_sum(dW, da)
                grads[2 * s], grads[2 * s + 1] = dW, da
        else:
            Nstep(L - 1, OTHER[first])
            Nstep(L - 1, first)
        for s in (range(L - 1, -1, -1) if not whole else ()):        # the pipelined half-table schedule
            Xu, Xi, W, a, H, S, out, agg, R = saved[s]
            Xs = {"user": Xu, "item": Xi}
            G, Gn, ST, dH, dS = tabs[s]
            heads = cfg["heads"][s]
            parts = {}
            for t in (first, OTHER[first]):
                pending.pop((s, OTHER[t])).wait()
                be.prop_bwd_edges(plan, H, S, a, G, Gn, ST, heads, cfg["alpha"], True, True, cfg["drop"], cfg["seeds"][s],
                                  half=t, dH=dH, dS=dS, **sd)
                dX, dW, da = be.transform_bwd_rows(Xs[t], W, H, dH, dS, part.Up, part.Ip, heads, part.half_win(t), True,
                                                   cfg["feat_drop"], cfg["seeds"][s], **sd)
                parts[t] = (dW, da)
                g[t] = dX if s == 0 else dX.to(h_dtype)
                if s > 0:
                    Nstep(s - 1, t)
            dW = torch.stack([parts["user"][0][0], parts["item"][0][1]])         # each half owns its weight panel
            da = torch.stack([parts["user"][1][0], parts["item"][1][1]])
            if not comm.defer_weight_reduce:
                comm.all_reduce_sum(dW, da)
            grads[2 * s], grads[2 * s + 1] = dW, da
            first = OTHER[first]
        comm.join()
        ctx.saved_stages = None
        ctx.xflat = None
        for ls in (lease, getattr(ctx, "lease", None)):
            if ls is not None:
                ls.release()
        ctx.lease = None
        return (g["user"], g["item"]) + tuple(grads) + (None,)


class _ShardedStackTwo(torch.autograd.Function):
    """The TWO propagations of a BPR step (train_eval_Gowalla.py:128-129: independent dropout masks, the same parameters) on the
    whole-stage schedule, in lock-step: stage s of chain A, stage s of chain B, then ONE grouped all-gather that carries both
    chains' tables -- half the collective launches of two _ShardedStack calls (where a gather is latency, the regime of this
    schedule, that is half the exposed latency), and the weight panels packed once.  Same kernels, same rows, same seeds per
    chain: the same bits as two calls.
    apply(Xu, Xi, W_0, a_0, ..., (chain A's inputs)  Xu, Xi, W_0, a_0, ..., (chain B's: the same parameters, their own autograd slots)  cfg)
    -> (out_u, out_i) of chain A, (out_u, out_i) of chain B."""

    @staticmethod
    def forward(ctx, *args):
        cfg = args[-1]
        L = cfg["L"]
        n_in = 2 + 2 * L
        ins = [args[k * n_in:(k + 1) * n_in] for k in range(2)]
        part, plan, comm, be = cfg["part"], cfg["plan"], cfg["comm"], cfg["be"]
        N, dev = part.N, ins[0][0].device
        need = cfg["grad_mode"] and any(ctx.needs_input_grad[:2 * n_in])
        sd = {} if cfg["seed_dev"] is None else {"seed_dev": cfg["seed_dev"]}
        Ws = [ins[0][2 + 2 * s].contiguous().float() for s in range(L)]          # (both chains read the same parameters)
        As = [ins[0][3 + 2 * s].contiguous().float() for s in range(L)]
        heads, seeds = cfg["heads"], cfg["seeds2"]                                # seeds2[k][s]: chain k, stage s
        fuse = [False] * L
        if cfg.get("fuse_next") and hasattr(be, "can_fuse_next"):
            for s in range(1, L):
                fuse[s] = (be.can_fuse_next(plan, cfg["h_dtype"], Ws[s - 1].shape[2], heads[s - 1])
                           and Ws[s].shape[1] == 64 and Ws[s].shape[2] in (64, 128))
        panels = {}
        if any(fuse):
            idx = [s for s in range(1, L) if fuse[s]]
            panels = dict(zip(idx, be.pack_panels([Ws[s] for s in idx]))) if cfg["h_dtype"] == torch.bfloat16 else {s: Ws[s] for s in idx}
        xflat = cfg.get("x_flat")
        Xu0, Xi0 = ins[0][0], ins[0][1]
        if xflat is not None and not (xflat.data_ptr() == Xu0.data_ptr() and Xu0.is_contiguous() and Xi0.is_contiguous()
                                      and Xi0.data_ptr() == xflat.data_ptr() + Xu0.numel() * Xu0.element_size()):
            xflat = None
        st = [{"X": {"user": i_[0].contiguous(), "item": i_[1].contiguous()}, "H": [None] * L, "S": [None] * L, "saved": [], "out": None}
              for i_ in ins]

        def new_H(c, s):
            c["H"][s] = torch.empty(N, Ws[s].shape[2], dtype=cfg["h_dtype"], device=dev)
            c["S"][s] = torch.empty(N, heads[s], dtype=torch.float32, device=dev)

        def transform(c, k, s):                                                    # all local rows of stage s, chain k
            if s == 0 and xflat is not None:
                be.transform_fwd_rows(xflat, Ws[0], As[0], c["H"][0], c["S"][0], part.Up, part.Ip, heads[0], part.win, True,
                                      cfg["feat_drop"], seeds[k][0], **sd)
            else:
                for t in ("user", "item"):
                    be.transform_fwd_rows(c["X"][t], Ws[s], As[s], c["H"][s], c["S"][s], part.Up, part.Ip, heads[s], part.half_win(t), True,
                                          cfg["feat_drop"], seeds[k][s], **sd)

        for k, c in enumerate(st):
            new_H(c, 0)
            transform(c, k, 0)
        h = comm.start_gather2(st[0]["H"][0], st[1]["H"][0])
        for s in range(L):
            fused_next = s + 1 < L and fuse[s + 1]
            keep_out = need or not fused_next
            h.wait()
            for k, c in enumerate(st):
                H, S = c["H"][s], c["S"][s]
                dout = H.shape[1]
                out = torch.empty(N, dout, dtype=H.dtype, device=dev) if keep_out else None
                agg = torch.empty(2, N, dout, dtype=H.dtype, device=dev) if need else None
                R = torch.empty(2, N, heads[s], dtype=torch.float32, device=dev) if need else None
                if s + 1 < L:
                    new_H(c, s + 1)
                Xs = dict(c["X"])
                if fused_next:
                    nxt = be.NextStage(panels[s + 1], As[s + 1], heads[s + 1], cfg["feat_drop"], seeds[k][s + 1], H=c["H"][s + 1], S=c["S"][s + 1])
                    be.prop_fwd(plan, H, S, S, As[s], heads[s], cfg["alpha"], True, True, True, cfg["drop"], seeds[k][s],
                                save=need, out=out, agg=agg, R=R, next=nxt, want_out=keep_out, **sd)
                else:
                    be.prop_fwd(plan, H, S, S, As[s], heads[s], cfg["alpha"], True, True, True, cfg["drop"], seeds[k][s],
                                save=need, out=out, agg=agg, R=R, **sd)
                if keep_out:
                    c["X"] = {t: part.block(out, t) for t in ("user", "item")}
                    c["out"] = out
                if s + 1 < L and not fused_next:
                    transform(c, k, s + 1)
                if need:
                    c["saved"].append((Xs["user"], Xs["item"], H, S, out, agg, R))
            if s + 1 < L:
                h = comm.start_gather2(st[0]["H"][s + 1], st[1]["H"][s + 1])
        if need:
            ctx.cfg, ctx.L, ctx.xflat = cfg, L, xflat
            ctx.Ws, ctx.As = Ws, As
            ctx.saved2 = [c["saved"] for c in st]
        return tuple(part.block(c["out"], t) for c in st for t in ("user", "item"))

    @staticmethod
    def backward(ctx, gUA, gIA, gUB, gIB):
        cfg, L = ctx.cfg, ctx.L
        part, plan, comm, be = cfg["part"], cfg["plan"], cfg["comm"], cfg["be"]
        sd = {} if cfg["seed_dev"] is None else {"seed_dev": cfg["seed_dev"]}
        N, heads, seeds, Ws, As = part.N, cfg["heads"], cfg["seeds2"], ctx.Ws, ctx.As
        saved2 = ctx.saved2
        h_dtype = saved2[0][0][2].dtype
        dev = gUA.device
        gin = [{"user": gUA.contiguous().to(h_dtype), "item": gIA.contiguous().to(h_dtype)},
               {"user": gUB.contiguous().to(h_dtype), "item": gIB.contiguous().to(h_dtype)}]
        # every stage's dH and dS of both chains out of ONE zeroed buffer
        sizes = []
        for k in range(2):
            for s in range(L):
                H = saved2[k][s][2]
                sizes += [(N * H.shape[1] * H.element_size() + 255) // 256 * 256, (N * heads[s] * 4 + 255) // 256 * 256]
        zeroed, zoff = torch.zeros(sum(sizes), dtype=torch.uint8, device=dev), [0]
        for z in sizes:
            zoff.append(zoff[-1] + z)
        tabs = [[None] * L, [None] * L]

        def alloc(k, s):
            H = saved2[k][s][2]
            d, j = H.shape[1], 2 * (k * L + s)
            if tabs[k][s] is None:
                tabs[k][s] = (torch.empty(N, d, dtype=torch.float32, device=dev), torch.empty(N, d, dtype=H.dtype, device=dev),
                              torch.empty(N, 2 * heads[s], dtype=torch.float32, device=dev),
                              zeroed[zoff[j]:zoff[j] + N * d * H.element_size()].view(H.dtype).view(N, d),
                              zeroed[zoff[j + 1]:zoff[j + 1] + N * heads[s] * 4].view(torch.float32).view(N, heads[s]))
            return tabs[k][s]

        def node(k, s, gfull=None, glocal=None):
            _, _, H, S, out, agg, R = saved2[k][s]
            G, Gn, ST, _, dS = alloc(k, s)
            if gfull is None and glocal is None:
                for t in ("user", "item"):
                    be.prop_bwd_node(out, agg, R, gin[k][t], S, G, Gn, ST, dS, part.Up, heads[s], part.half_win(t), True, True, cfg["alpha"])
            else:
                be.prop_bwd_node(out, agg, R, gfull if gfull is not None else glocal, S, G, Gn, ST, dS, part.Up, heads[s], part.win,
                                 gfull is None, True, cfg["alpha"])

        def one_table(gd):                                                        # the two incoming halves as views of ONE [cu + ci, d] table?
            base = gd["user"]._base
            ok = (base is not None and base is gd["item"]._base and base.dim() == 2 and base.is_contiguous() and base.dtype == h_dtype
                  and base.shape[0] == part.cu + part.ci and gd["user"].data_ptr() == base.data_ptr()
                  and gd["item"].data_ptr() == base.data_ptr() + part.cu * base.shape[1] * base.element_size())
            return base if ok else None

        for k in range(2):
            node(k, L - 1, glocal=one_table(gin[k]))
        h = comm.start_gather2(tabs[0][L - 1][1], tabs[0][L - 1][2], tabs[1][L - 1][1], tabs[1][L - 1][2])
        grads = [[None] * (2 * L), [None] * (2 * L)]
        gx = [None, None]
        for s in range(L - 1, -1, -1):
            h.wait()
            for k in range(2):
                Xu, Xi, H, S, out, agg, R = saved2[k][s]
                G, Gn, ST, dH, dS = tabs[k][s]
                W, a = Ws[s], As[s]
                be.prop_bwd_edges(plan, H, S, a, G, Gn, ST, heads[s], cfg["alpha"], True, True, cfg["drop"], seeds[k][s], dH=dH, dS=dS, **sd)
                if s > 0:
                    dX, dW, da = be.transform_bwd_rows(saved2[k][s - 1][4], W, H, dH, dS, part.Up, part.Ip, heads[s], part.win, False,
                                                       cfg["feat_drop"], seeds[k][s], **sd)
                    node(k, s - 1, gfull=dX if dX.dtype == h_dtype else dX.to(h_dtype))
                elif ctx.xflat is not None:
                    dX, dW, da = be.transform_bwd_rows(ctx.xflat, W, H, dH, dS, part.Up, part.Ip, heads[s], part.win, True,
                                                       cfg["feat_drop"], seeds[k][s], **sd)
                    gx[k] = (dX[:part.cu], dX[part.cu:])
                else:
                    parts = {}
                    for t, Xt in (("user", Xu), ("item", Xi)):
                        dXt, dW_, da_ = be.transform_bwd_rows(Xt, W, H, dH, dS, part.Up, part.Ip, heads[s], part.half_win(t), True,
                                                              cfg["feat_drop"], seeds[k][s], **sd)
                        parts[t] = (dXt, dW_, da_)
                    gx[k] = (parts["user"][0], parts["item"][0])
                    dW = torch.stack([parts["user"][1][0], parts["item"][1][1]])
                    da = torch.stack([parts["user"][2][0], parts["item"][2][1]])
                grads[k][2 * s], grads[k][2 * s + 1] = dW, da
            if s > 0:
                h = comm.start_gather2(tabs[0][s - 1][1], tabs[0][s - 1][2], tabs[1][s - 1][1], tabs[1][s - 1][2])
            if not comm.defer_weight_reduce:
                comm.all_reduce_sum(grads[0][2 * s], grads[0][2 * s + 1], grads[1][2 * s], grads[1][2 * s + 1])
        ctx.saved2 = None
        ctx.xflat = None
        return tuple(gx[0]) + tuple(grads[0]) + tuple(gx[1]) + tuple(grads[1]) + (None,)


class _RowExchange(torch.autograd.Function):
    """rows[b] = table row `idx[b]` (internal global row) on EVERY rank: each rank contributes the rows it owns, zeros
    elsewhere, one all-reduce of [B,d] sums them (one non-zero term per row: exact).  Replaces the all-gather of the
    whole [N,d] table for 3 x B rows.  Every rank computes the same loss, so the gradient of a row is identical on
    all ranks and its owner adds it locally: no communication in the backward.  (The torch stand-in of the CPU tests;
    the product runs _RowExchangeHIP.)"""

    @staticmethod
    def forward(ctx, out_u, out_i, idx, part, comm):
        local = torch.cat([out_u, out_i])
        is_u = idx < part.Up
        pos = torch.where(is_u, idx - part.u_lo, idx - part.Up - part.i_lo + part.cu)
        mine = torch.where(is_u, (idx >= part.u_lo) & (idx < part.u_lo + part.cu),
                           (idx >= part.Up + part.i_lo) & (idx < part.Up + part.i_lo + part.ci))
        pos = torch.where(mine, pos, torch.zeros_like(pos))
        rows = local[pos].float() * mine[:, None].float()
        comm.all_reduce_sum(rows)
        ctx.save_for_backward(pos, mine)
        ctx.shape = (out_u.shape[0], out_i.shape[0], out_u.dtype)
        return rows

    @staticmethod
    def backward(ctx, g):
        pos, mine = ctx.saved_tensors
        cu, ci, dt = ctx.shape
        d = torch.zeros(cu + ci, g.shape[1], dtype=torch.float32, device=g.device)
        d.index_add_(0, pos, g.float() * mine[:, None].float())
        return d[:cu].to(dt), d[cu:].to(dt), None, None, None


class _RowExchangeHIP(torch.autograd.Function):
    """The same on the HIP library: ONE launch writes the rows this rank owns (and remembers where they sit: ngacf_shard_rows,
    data-set ids -> internal ids -> local rows inside the kernel), the all-reduce, and in the backward ONE launch adds the
    gradient rows into their owners' rows (ngacf_rows_scatter: no atomics).  The torch form above costs ~25 index / compare /
    gather launches of 5 us per forward -- a quarter of the node-partitioned train step at Gowalla scale."""

    @staticmethod
    def forward(ctx, out_u, out_i, u, it, part, comm):
        dev = out_u.device
        pu = part._on("perm_u", dev) if part.perm_u is not None else None
        pi = part._on("perm_i", dev) if part.perm_i is not None else None
        rows, pos = _hip_ops.shard_rows(out_u.contiguous(), out_i.contiguous(), u, it, pu, pi, part.U, part.I, part.u_lo, part.i_lo)
        comm.all_reduce_sum(rows)
        ctx.save_for_backward(pos)
        ctx.shape = (out_u.shape[0], out_i.shape[0], out_u.dtype)
        return rows

    @staticmethod
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        cu, ci, dt = ctx.shape
        du, di = _hip_ops.rows_scatter(pos, g.contiguous().float(), cu, ci, dt)
        return du, di, None, None, None, None


class _ExchangeScoreHIP(torch.autograd.Function):
    """scores [B] of the batch = _RowExchangeHIP + the pair scoring as ONE autograd node: the backward forms the gradient rows of
    the 2B exchanged rows from the scores' gradient inside the scatter launch (ngacf_rows_scatter_scored) -- the score kernel's own
    backward over the [2B, d] row table (a duplicate scan over row numbers that are all distinct: 13 us) and its launch are gone."""

    @staticmethod
    def forward(ctx, out_u, out_i, u, it, part, comm, ar):
        dev = out_u.device
        pu = part._on("perm_u", dev) if part.perm_u is not None else None
        pi = part._on("perm_i", dev) if part.perm_i is not None else None
        rows, pos = _hip_ops.shard_rows(out_u.contiguous(), out_i.contiguous(), u, it, pu, pi, part.U, part.I, part.u_lo, part.i_lo)
        comm.all_reduce_sum(rows)
        ctx.save_for_backward(pos, rows)
        ctx.shape = (out_u.shape[0], out_i.shape[0], out_u.dtype)
        return _hip_ops.score_fwd(rows, ar, ar, ar.numel())

    @staticmethod
    def backward(ctx, gs):
        pos, rows = ctx.saved_tensors
        cu, ci, dt = ctx.shape
        du, di = _hip_ops.rows_scatter_scored(pos, rows, gs.contiguous().float(), cu, ci, dt)
        return du, di, None, None, None, None, None


class _ExchangeScoreTwoHIP(torch.autograd.Function):
    """_ExchangeScoreHIP for the two propagations of a step at once: (scores of (u, pos) on chain A's table, scores of (u, neg) on chain
    B's), the two row exchanges as ONE grouped all-reduce."""

    @staticmethod
    def forward(ctx, ua, ia, ub, ib, u, pos, neg, part, comm, ar):
        dev = ua.device
        pu = part._on("perm_u", dev) if part.perm_u is not None else None
        pi = part._on("perm_i", dev) if part.perm_i is not None else None
        ra, pa = _hip_ops.shard_rows(ua.contiguous(), ia.contiguous(), u, pos, pu, pi, part.U, part.I, part.u_lo, part.i_lo)
        rb, pb = _hip_ops.shard_rows(ub.contiguous(), ib.contiguous(), u, neg, pu, pi, part.U, part.I, part.u_lo, part.i_lo)
        comm.all_reduce_sum(ra, rb)
        ctx.save_for_backward(pa, ra, pb, rb)
        ctx.shape = (ua.shape[0], ia.shape[0], ua.dtype)
        return _hip_ops.score_fwd(ra, ar, ar, ar.numel()), _hip_ops.score_fwd(rb, ar, ar, ar.numel())

    @staticmethod
    def backward(ctx, ga, gb):
        pa, ra, pb, rb = ctx.saved_tensors
        cu, ci, dt = ctx.shape
        dua, dia = _hip_ops.rows_scatter_scored(pa, ra, ga.contiguous().float(), cu, ci, dt)
        dub, dib = _hip_ops.rows_scatter_scored(pb, rb, gb.contiguous().float(), cu, ci, dt)
        return dua, dia, dub, dib, None, None, None, None, None, None


class ShardedSPUIGACF(nn.Module):
    """SPUIGACF / SPUIMultiGACF / SPUIStackedGACF with the embedding tables sharded by node blocks.
    forward(userIdx, itemIdx, mask) has the reference signature (SPUIGACF.py:41) and returns the scores of the
    whole batch on every rank.  `adj` (the [2,E] mask the forward will see) or `degrees=(deg_u, deg_i)` at construction
    selects the degree-balanced partition; without either the blocks are runs of consecutive ids."""

    def __init__(self, userNum, itemNum, embedSize, layers, droprate, world=None, rank=None, stages=2,
                 group=None, backend=None, device=None, adj=None, degrees=None, partition=None, peer_store=None):
        super().__init__()
        world = dist.get_world_size(group) if world is None else world
        rank = dist.get_rank(group) if rank is None else rank
        self.userNum, self.itemNum, self.droprate = userNum, itemNum, droprate
        if adj is not None and degrees is None:
            a = squeeze_mask(adj)
            degrees = (torch.bincount(a[0], minlength=userNum), torch.bincount(a[1], minlength=itemNum))
        self.part = partition if partition is not None else \
            Partition(userNum, itemNum, world, rank, *(degrees if degrees is not None else (None, None)))
        # peer_store (or NGACF_PEER_GATHER=1): the half-table gathers of the propagation go by stores into the peers' tables
        # (ngacf_amd/peer.py) instead of one RCCL all-gather launch each; everything else stays on the process group
        if peer_store is None:
            peer_store = os.environ.get("NGACF_PEER_GATHER", "0") == "1"
        if peer_store:
            from .peer import PeerComm
            self.comm = PeerComm(self.part, group, device=device)
        else:
            self.comm = Comm(self.part, group)
        # the second propagation chain of a train step on its own communicator (Comm.twin; NGACF_DIST_TWO_COMMS=0: one chain after
        # the other on the one communicator, the schedule of rounds 1-4).  Peer stores keep one chain at a time (shared tables).
        self._comms = [self.comm, self.comm]
        if not peer_store and dist.is_initialized() and os.environ.get("NGACF_DIST_TWO_COMMS", "1") != "0":
            self._comms = [self.comm, self.comm.twin()]
        self._chain = 0
        self.be = backend if backend is not None else _hip_ops
        p = self.part
        # Same values on every rank -- and at every world size -- as the unsharded model under the same torch seed: the random
        # draws happen in the unsharded constructor's order (SPUIGACF.py:7-28: nn.Embedding's own N(0,1) init of both tables, the
        # layers' xavier draws, then N(0, 0.01^2) over both tables), on full-size tables of which this rank keeps its rows.
        # Tables beyond 1 GiB are drawn block-wise instead: every rank seeds its own block.
        small = (userNum + itemNum) * embedSize * 4 <= (1 << 30)
        if small:
            torch.empty(userNum, embedSize).normal_()
            torch.empty(itemNum, embedSize).normal_()
        self.uEmbd = nn.Embedding(p.cu, embedSize, _weight=torch.zeros(p.cu, embedSize))
        self.iEmbd = nn.Embedding(p.ci, embedSize, _weight=torch.zeros(p.ci, embedSize))
        if stages == 2:
            self.gat = SpUIGAT(embedSize, 8, embedSize, droprate, 0.2, 8)
        elif stages == 3 and embedSize == 64:
            self.gat = SPUIMultiGAT(embedSize, 8, embedSize, droprate, 0.2, 8)
        else:
            self.gat = StackedSpUIGAT(embedSize, 8, embedSize, droprate, 0.2, 8, stages - 1)
        if small:
            full_u = torch.empty(userNum, embedSize).normal_(std=0.01)       # SPUIGACF.py:26-28
            full_i = torch.empty(itemNum, embedSize).normal_(std=0.01)
            self.load_full_embeddings(full_u, full_i)
        else:
            g = torch.Generator().manual_seed(2019 + rank)
            ou, oi = p.local_original_ids()
            with torch.no_grad():
                self.uEmbd.weight.normal_(std=0.01, generator=g)
                self.iEmbd.weight.normal_(std=0.01, generator=g)
                self.uEmbd.weight[ou < 0] = 0                                 # padding slots: isolated zero nodes
                self.iEmbd.weight[oi < 0] = 0
        self._plans = _PlanCache()
        self._attached = None
        self.compute_dtype = torch.float32
        self.seg_len = 256

    # ---- parameter plumbing -------------------------------------------------------------------------------
    def _link_tables(self):
        """uEmbd.weight and iEmbd.weight as views of ONE contiguous [cu + ci, d] table (same Parameter objects, same values:
        SPUIGACF._CFBase._link_tables): the whole-stage schedule transforms all local rows of the first stage, forward and
        backward, in one launch each."""
        wu, wi = self.uEmbd.weight, self.iEmbd.weight
        flat = torch.empty(wu.shape[0] + wi.shape[0], wu.shape[1], device=wu.device, dtype=wu.dtype)
        flat[:wu.shape[0]] = wu.data
        flat[wu.shape[0]:] = wi.data
        wu.data, wi.data = flat[:wu.shape[0]], flat[wu.shape[0]:]
        object.__setattr__(self, "_flat", flat)

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)                               # .to() / .cuda() move the halves apart
        if "uEmbd" in self._modules and "iEmbd" in self._modules:
            self._link_tables()
        return r

    def _flat_local(self):
        wu, wi, flat = self.uEmbd.weight, self.iEmbd.weight, self.__dict__.get("_flat")
        if flat is None or flat.device != wu.device or flat.dtype != wu.dtype or wu.data_ptr() != flat.data_ptr() \
                or wi.data_ptr() != flat.data_ptr() + wu.numel() * wu.element_size():
            self._link_tables()                                       # somebody re-assigned .data: alias again
            flat = self._flat
        return flat

    def load_full_embeddings(self, full_u, full_i):
        ou, oi = self.part.local_original_ids(full_u.device)
        with torch.no_grad():
            for w, full, ids in ((self.uEmbd.weight, full_u, ou), (self.iEmbd.weight, full_i, oi)):
                w.zero_()                                           # padding slots: isolated zero nodes
                ok = ids >= 0
                w[ok.to(w.device)] = full[ids[ok]].to(w.device, w.dtype)

    def load_full_state_dict(self, sd):
        """Take an unsharded (reference-keyed) state_dict: pick this rank's embedding rows, copy the rest."""
        self.load_full_embeddings(sd["uEmbd.weight"], sd["iEmbd.weight"])
        self.gat.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("gat.")})

    def _gathered(self, u_local, i_local):
        p = self.part
        tab = torch.zeros(p.N, u_local.shape[1], device=u_local.device, dtype=u_local.dtype)
        p.block(tab, "user").copy_(u_local)
        p.block(tab, "item").copy_(i_local)
        self.comm.all_gather_rows(tab)
        return p.to_original(tab)

    def full_state_dict(self):
        """The reference-keyed state_dict on every rank (all-gather of the embedding shards, data-set numbering)."""
        full = self._gathered(self.uEmbd.weight.detach(), self.iEmbd.weight.detach())
        sd = {"uEmbd.weight": full[:self.userNum].clone(), "iEmbd.weight": full[self.userNum:].clone()}
        sd.update({"gat." + k: v.detach().clone() for k, v in self.gat.state_dict().items()})
        return sd

    def local_grads_original(self):
        """((user ids, grad rows), (item ids, grad rows), padding-slot grads) of this rank's embedding gradients, ids in
        the data set's numbering."""
        ou, oi = self.part.local_original_ids(self.uEmbd.weight.device)
        gu, gi = self.uEmbd.weight.grad, self.iEmbd.weight.grad
        return (ou[ou >= 0], gu[ou >= 0]), (oi[oi >= 0], gi[oi >= 0]), torch.cat([gu[ou < 0], gi[oi < 0]])

    def collective_launches(self, reset=False):
        """collective launches issued so far on both communicators"""
        cs = list({id(c): c for c in self._comms + [getattr(self, "_comm0_inline", None)] if c is not None}.values())
        n = sum(c.launches for c in cs)
        if reset:
            for c in cs:
                c.launches = 0
        return n

    def reduce_grads(self, sinks=None):
        """With comm.defer_weight_reduce: sum the stage-weight gradients over the ranks in ONE grouped all-reduce,
        after backward() and before the optimizer step (embedding rows are owned, their gradients are local).
        sinks: the gradient sinks of a step whose chains parked their pieces there (functional.grad_sink: .grad stays None) --
        the stage weights' pieces of every sink travel in the one grouped launch."""
        if sinks is None:
            gs = [p.grad for p in self.gat.parameters() if p.grad is not None]
        else:
            mine = {id(p) for p in self.gat.parameters()}
            gs = [g for sk in sinks for p, g in sk.items() if id(p) in mine and g is not None]
        if gs:
            self.comm.all_reduce_sum(*gs)

    reduce_grads_takes_sinks = True

    # ---- graph ---------------------------------------------------------------------------------------------
    def attach_plan(self, plan):
        """Use a plan built elsewhere (sharded_graph: the global edge list never exists); forward's `mask` is then ignored."""
        self._attached = plan

    def _plan(self, mask):
        """The sharded plan of `mask` through SPUIGACF._PlanCache: keyed by (address, shape, device, version) of the tensor
        AND holding a reference to it, so a different adjacency can never be allocated at a cached address (the reference's
        loop re-uploads `adj` every epoch, train_eval_Gowalla.py:106)."""
        if self._attached is not None:
            return self._attached
        mask = squeeze_mask(mask)
        p = self.part
        return self._plans.get(mask, p.Up, p.Ip, self.seg_len,
                               builder=lambda: build_plan(p.map_edges(mask.long()), p.Up, p.Ip, seg_len=self.seg_len,
                                                          require_user_edges=False, row_range=p.row_ranges))

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

    # ---- forward ------------------------------------------------------------------------------------------
    @property
    def concurrent_chains(self):
        """Should a CAPTURED train step run its two forwards as concurrent branches (each on its own communicator)?  The plumbing is
        in place and tested (second communicator, in-order gathers: tests/test_dist_cpu.py; chain(k) in graphs.GraphedBPRStep), but
        hipGraph instantiation of that step SEGFAULTS on this runtime (capture_end, one RCCL rank, with the collectives on the
        communicators' own streams and with them in the branches' own order alike: profiles/README.md, r05 rehearsal) -- the
        same failure tools/graph_topology_probe.py shows for cross-branch waits.  So: opt-in, NGACF_DIST_CONCURRENT_GRAPH=1."""
        return self._comms[1] is not self._comms[0] and os.environ.get("NGACF_DIST_CONCURRENT_GRAPH", "0") == "1"

    @contextlib.contextmanager
    def chain(self, k, inline=False):
        """with model.chain(k): forwards issued inside run on communicator k (0 / 1) -- graphs.GraphedBPRStep puts the two
        propagations of a step on two streams and two communicators; their backwards follow (the autograd node holds its own).
        inline: that chain's gathers in its stream's own order (Comm.inline): what a captured concurrent chain needs."""
        was, self._chain = (self._chain, self._inline), int(k) & 1
        self._inline = bool(inline)
        try:
            yield self
        finally:
            self._chain, self._inline = was

    _inline = False

    def _comm_now(self):
        c = self._comms[self._chain].sync_switches()
        if self._inline and c._primary is None and self._comms[1] is not self._comms[0]:
            # (chain 0 of a concurrent pair: the primary communicator itself must not stay inline for its other users)
            c = self._inline_view(c)
        elif self._inline:
            c.inline = True
        else:
            c.inline = False
        return c

    def _inline_view(self, c):
        v = getattr(self, "_comm0_inline", None)
        if v is None:
            v = Comm(c.part, c.group)
            v._primary = c
            self._comm0_inline = v
        v.sync_switches()
        v.inline = True
        return v

    def propagate_local(self, mask):
        """(out_u [cu,d], out_i [ci,d]): this rank's rows of the propagated embeddings."""
        plan = self._plan(mask)
        drop = self.droprate if self.training else 0.0
        seed_dev = None
        if drop > 0 and getattr(self, "_seed_ctr", None) is not None:
            self._seed_call += 1
            base = (self._seed_base + self._seed_call * _SEED_STRIDE) & 0x3FFFFFFFFFFFFFFF
            seed_dev = self._seed_ctr
        else:
            base = _fresh_seed() if drop > 0 else 0                      # same CPU generator state on every rank
        stages = self.gat._stages()
        params = []
        for s in range(len(stages)):
            params.extend(NF.through_sink(p_) for p_ in self.gat.stage_params(s))      # (behind the caller's gradient sink, if any)
        comm = self._comm_now()
        elem = 2 if self.compute_dtype == torch.bfloat16 else 4
        widest = max(int(self.gat.stage_params(s)[0].shape[-1]) for s in range(len(stages)))
        cfg = {"part": self.part, "plan": plan, "comm": comm, "be": self.be, "heads": [len(l) for l in stages],
               "whole": comm.whole_stage(self.part.N * widest * elem),
               "x_flat": self._flat_local(),            # (the two local tables as one: the whole-stage schedule's first stage)
               "alpha": stages[0][0].alpha, "drop": drop, "feat_drop": drop,
               "seeds": [NF.stage_seed(base, s) for s in range(len(stages))], "h_dtype": self.compute_dtype,
               "seed_dev": seed_dev, "grad_mode": torch.is_grad_enabled(), "fuse_next": NF.FUSE_NEXT}
        return _ShardedStack.apply(NF.through_sink(self.uEmbd.weight), NF.through_sink(self.iEmbd.weight), *params, cfg)

    def forward_two(self, userIdx, posIdx, negIdx, mask, sinks=None):
        """(scores of (userIdx, posIdx), scores of (userIdx, negIdx)) = forward(userIdx, posIdx, mask), forward(userIdx, negIdx, mask)
        of the reference's BPR step (train_eval_Gowalla.py:128-129: two propagations, independent dropout) -- as ONE lock-step pass
        where the whole-stage schedule applies (_ShardedStackTwo: one grouped all-gather per stage for both chains), else the two
        calls.  sinks: (sink of the first forward, sink of the second) -- functional.grad_sink of each chain's parameter reads."""
        import contextlib
        comm = self._comm_now()
        elem = 2 if self.compute_dtype == torch.bfloat16 else 4
        stages = self.gat._stages()
        widest = max(int(self.gat.stage_params(s)[0].shape[-1]) for s in range(len(stages)))
        sk = sinks if sinks is not None else (None, None)
        ctxs = [(NF.grad_sink(sk[k], deferred_ok=True) if sinks is not None else contextlib.nullcontext()) for k in range(2)]
        if not (self.be is _hip_ops or hasattr(self.be, "prop_fwd")) or not comm.whole_stage(self.part.N * widest * elem) \
                or os.environ.get("NGACF_DIST_LOCKSTEP", "1") == "0":
            with ctxs[0]:
                ps = self.forward(userIdx, posIdx, mask)
            with ctxs[1]:
                ns = self.forward(userIdx, negIdx, mask)
            return ps, ns
        plan = self._plan(mask)
        drop = self.droprate if self.training else 0.0
        bases, seed_dev = [], None
        for k in range(2):                                           # the seeds two forward() calls would draw, in their order
            if drop > 0 and getattr(self, "_seed_ctr", None) is not None:
                self._seed_call += 1
                bases.append((self._seed_base + self._seed_call * _SEED_STRIDE) & 0x3FFFFFFFFFFFFFFF)
                seed_dev = self._seed_ctr
            else:
                bases.append(_fresh_seed() if drop > 0 else 0)
        L = len(stages)
        ins = []
        for k in range(2):
            with ctxs[k]:
                ins += [NF.through_sink(self.uEmbd.weight), NF.through_sink(self.iEmbd.weight)]
                for s in range(L):
                    ins.extend(NF.through_sink(p_) for p_ in self.gat.stage_params(s))
        cfg = {"part": self.part, "plan": plan, "comm": comm, "be": self.be, "heads": [len(l) for l in stages], "L": L,
               "x_flat": self._flat_local(), "alpha": stages[0][0].alpha, "drop": drop, "feat_drop": drop,
               "seeds2": [[NF.stage_seed(b, s) for s in range(L)] for b in bases], "h_dtype": self.compute_dtype,
               "seed_dev": seed_dev, "grad_mode": torch.is_grad_enabled(), "fuse_next": NF.FUSE_NEXT}
        outs = _ShardedStackTwo.apply(*ins, cfg)
        B = userIdx.numel()
        if self.be is _hip_ops and 2 * B <= _hip_ops.ROWS_SCATTER_MAX_N and outs[0].shape[1] in _hip_ops.SUPPORTED_D \
                and os.environ.get("NGACF_DIST_FUSED_SCORE", "1") != "0":
            return _ExchangeScoreTwoHIP.apply(*outs, userIdx.long().contiguous(), posIdx.long().contiguous(), negIdx.long().contiguous(),
                                              self.part, comm, self._arange_of(B, outs[0].device))
        return self._scores(outs[0], outs[1], userIdx, posIdx), self._scores(outs[2], outs[3], userIdx, negIdx)

    def _arange_of(self, B, dev):
        ar = self.__dict__.get("_arange")                                            # (built once per batch size, not per forward)
        if ar is None or ar.numel() != B or ar.device != dev:
            ar = torch.arange(B, device=dev)
            self.__dict__["_arange"] = ar
        return ar

    def propagate(self, mask):
        """The full [U+I, d] table of propagated embeddings in the data set's numbering, on every rank (evaluation)."""
        out_u, out_i = self.propagate_local(mask)
        return self._gathered(out_u.detach(), out_i.detach())

    def forward(self, userIdx, itemIdx, mask):
        out_u, out_i = self.propagate_local(mask)
        return self._scores(out_u, out_i, userIdx, itemIdx)

    def _scores(self, out_u, out_i, userIdx, itemIdx):
        """the batch's scores from this rank's rows of the propagated table: the 2B scored rows exchanged, then SPUIGACF.py:49-52"""
        p = self.part
        B = userIdx.numel()
        if self.be is _hip_ops and 2 * B <= _hip_ops.ROWS_SCATTER_MAX_N and out_u.shape[1] in _hip_ops.SUPPORTED_D \
                and os.environ.get("NGACF_DIST_FUSED_SCORE", "1") != "0":
            ar = self.__dict__.get("_arange")
            if ar is None or ar.numel() != B or ar.device != out_u.device:
                ar = torch.arange(B, device=out_u.device)
                self.__dict__["_arange"] = ar
            return _ExchangeScoreHIP.apply(out_u, out_i, userIdx.long().contiguous(), itemIdx.long().contiguous(), p, self._comm_now(), ar)
        if self.be is _hip_ops and 2 * B <= _hip_ops.ROWS_SCATTER_MAX_N:
            rows = _RowExchangeHIP.apply(out_u, out_i, userIdx.long().contiguous(), itemIdx.long().contiguous(), p, self._comm_now())
        else:
            idx = torch.cat([p.internal_users(userIdx.long()), p.internal_items(itemIdx.long()) + p.Up])
            rows = _RowExchange.apply(out_u, out_i, idx, p, self._comm_now())       # [2B, d]: user rows then item rows
        ar = self.__dict__.get("_arange")                                            # (built once per batch size, not per forward)
        if ar is None or ar.numel() != B or ar.device != rows.device:
            ar = torch.arange(B, device=rows.device)
            self.__dict__["_arange"] = ar
        score = self.be.score if hasattr(self.be, "score") else NF.score
        return score(rows, ar, ar, B)                                                # SPUIGACF.py:49-52


# ---------------------------------------------------------------------------------------------------------
# graphs that never exist whole: the chunked generator, kept shard by shard
# ---------------------------------------------------------------------------------------------------------
def sharded_graph(shape, world, rank, dev, seed=2019, seg_len=256, chunk_users=1 << 18):
    """(Partition, plan, E, first_chunk_edges) of a chunk-generated graph for rank `rank` of `world`: every rank runs the
    chunked generator (ngacf_amd.data.synthetic_bipartite_chunks; any rank can produce any chunk) twice -- degrees first,
    then the edges, of which it keeps only those ending in its rows.  Peak memory: one chunk + the rank's shard.
    shape = (U, I, E) or a workload name.  first_chunk_edges: the [2,e] edges of the first user chunk in the data set's
    numbering (the CPU baseline's sample)."""
    from . import data as D
    U, I, E_target = D.SHAPES[shape] if isinstance(shape, str) else shape
    deg_u = torch.zeros(U, dtype=torch.long, device=dev)
    deg_i = torch.zeros(I, dtype=torch.long, device=dev)
    E = 0
    for lo, hi, key in D.synthetic_bipartite_chunks(U, I, E_target, seed, dev, chunk_users=chunk_users):
        deg_u[lo:hi] = torch.bincount(key // I - lo, minlength=hi - lo)
        deg_i += torch.bincount(key % I, minlength=I)
        E += int(key.numel())
    part = Partition(U, I, world, rank, deg_u, deg_i)
    del deg_u, deg_i
    pu, pi = part._on("perm_u", dev), part._on("perm_i", dev)
    csr, csc, first = [], [], None
    for lo, hi, key in D.synthetic_bipartite_chunks(U, I, E_target, seed, dev, chunk_users=chunk_users):
        if first is None:
            first = torch.stack([key // I, key % I])
        iu, ii = pu[key // I], pi[key % I]
        m = (iu >= part.u_lo) & (iu < part.u_lo + part.cu)
        csr.append(iu[m] * part.Ip + ii[m])
        m = (ii >= part.i_lo) & (ii < part.i_lo + part.ci)
        csc.append(ii[m] * part.Up + iu[m])
    csr_key = torch.sort(torch.cat(csr)).values
    del csr
    csc_key = torch.sort(torch.cat(csc)).values
    del csc
    plan = build_plan_from_keys(csr_key, csc_key, part.Up, part.Ip, part.row_ranges, seg_len)
    return part, plan, E, first


# ---------------------------------------------------------------------------------------------------------
# the scaling probe of the default bench line: the forward on the 10^9-edge graph (BASELINE config 5)
# ---------------------------------------------------------------------------------------------------------
PROBE_WORKLOAD = "synthetic1b"


def scaling_probe(args, world, rank, dev, d=64):
    """The eval forward of the 10^9-edge synthetic graph (10 M x 5 M, mean degree 133: the configuration where a rank's
    traversal lasts about as long as its all-gather) at THIS GPU count -- reported inside the default
    Gowalla line so that one `bench.py --gpus N` sweep also yields a strong-scaling curve in the regime where the node
    partition can pay.  world == 1: the plain single-GPU model (the best one-GPU implementation is the baseline);
    world > 1: the sharded model (chunk-generated shards, pipelined all-gathers), replayed as a hipGraph when the capture
    completes.  Returns a dict (rank-independent)."""
    import time
    from . import benchkit as BK, data as D
    wl = PROBE_WORKLOAD
    U, I, _ = D.SHAPES[wl]
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    elem = 2 if args.dtype == "bf16" else 4
    t_setup = time.perf_counter()
    if world == 1:
        import ngacf_amd as ng
        edge = D.workload_graph(wl, dev)
        E = int(edge.shape[1])
        model = ng.SPUIMultiGACF(U, I, d, [64, 64], 0.0).to(dev).eval()
        ng.set_compute_dtype(model, dt)
        fwd = lambda: model.propagate(edge)
        sync = torch.cuda.synchronize
        allmax = lambda x: x
        local_edges, n_local = 2 * E, U + I
    else:
        part, plan, E, _ = sharded_graph(wl, world, rank, dev, seg_len=args.seg_len)
        model = ShardedSPUIGACF(U, I, d, [64, 64], 0.0, world, rank, stages=3, partition=part).to(dev).eval()
        model.attach_plan(plan)
        model.compute_dtype = dt
        model.comm.overlap = not args.no_overlap
        edge = torch.zeros(2, 1, dtype=torch.long, device=dev)
        fwd = lambda: model.propagate_local(edge)

        def sync():
            dist.barrier()
            torch.cuda.synchronize()

        def allmax(x):
            t = torch.tensor([x], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)
        local_edges, n_local = plan.n_edges_directed, part.n_local
    L = 3
    setup_s = time.perf_counter() - t_setup
    K = max(3, min(args.steps, 10))
    with torch.no_grad():
        fwd()
        sync()
        with _hip_ops.KernelTimer("ngacf_prop_fwd") as kt:
            for _ in range(3):
                fwd()
            torch.cuda.synchronize()
        durs = kt.durations_ms()
        t_eager, _ = BK.timed_steps(fwd, K, 1, sync, torch)
        t_eager = allmax(t_eager)
        best, launch = t_eager, "eager"
        try:
            from .graphs import quiesce_collectives
            graph = torch.cuda.CUDAGraph()
            mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
            quiesce_collectives()                      # (the watchdog must have retired every eager collective: graphs.py)
            with torch.cuda.graph(graph, capture_error_mode=mode):
                fwd()
            graph.replay()
            t_graph, _ = BK.timed_steps(graph.replay, K, 1, sync, torch)
            t_graph = allmax(t_graph)
            if t_graph < best:
                best, launch = t_graph, "hipGraph replay"
        except Exception as e:                                     # the eager number stands
            import sys
            print("[bench] probe capture not used: %r" % (e,), file=sys.stderr, flush=True)
    ag = None
    if world > 1:                                       # the measured all-gather receive rate beside the 340 GB/s that >= 6 x needs
        try:
            with torch.no_grad():
                ag = allgather_summary(allgather_timing(model, fwd, 2, not args.no_overlap), "RCCL all-gather (in place)")
        except Exception as e:                          # (a secondary figure must not cost the probe)
            import sys
            print("[bench] probe all-gather timing failed: %r" % (e,), file=sys.stderr, flush=True)
    stage_ms = sum(durs) / 3 / L
    bytes_stage = local_edges * (4 + d * elem) + n_local * (2 * d * elem + 8)
    achieved = -allmax(-bytes_stage / (stage_ms * 1e-3) / 1e9)
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traffic, traffic_src = BK.traffic_from_profile(root, wl, args.dtype, 3, d) if world == 1 else (None, None)
    roofline = {"bound": "hbm", "kernel": "prop_fwd_kernel + the hub rows' second kernel (11 M slices: beyond the in-kernel finish)",
                "achieved": round(achieved, 1), "peak": BK.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / BK.HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "counter_GBps": round(traffic / (stage_ms * 1e-3) / 1e9, 1) if traffic else None,
                "bytes_per_launch": int(bytes_stage), "avg_launch_us": round(stage_ms * 1e3, 1), "regime": "HBM"}
    return {"workload": "%s bipartite graph U=%d I=%d E=%d, d=%d, L=3, eval forward" % (wl, U, I, E, d), "n_gpus": world,
            "roofline": roofline,
            "value": round(L * 2 * E * K / best, 1), "unit": "edges/s", "ms_per_pass": round(1e3 * best / K, 3), "passes": K,
            "launch": launch, "ms_per_pass_eager": round(1e3 * t_eager / K, 3), "dtype": args.dtype,
            "prop_fwd_GBps_slowest_rank": round(achieved, 1), "prop_fwd_frac_of_hbm_peak": round(achieved / BK.HBM_PEAK_GBS, 4),
            "stage_kernel_ms": round(stage_ms, 3), "local_directed_edges": int(local_edges), "setup_s": round(setup_s, 1),
            "allgather": ag,
            "schedule": "1 GPU" if world == 1 else ("pipelined half-table all-gathers" if not args.no_overlap else "serial all-gathers")}


NEEDED_RECEIVE_GBPS_FOR_6X = 340.0        # >= 6 x at 8 GPUs on the 10^9-edge graph needs this all-gather receive rate per GPU


def allgather_timing(model, fwd, n_passes, restore_overlap):
    """[(ms from start to wait, bytes this rank received)] of every half-table all-gather of `n_passes` SERIAL forward passes
    (each gather waited for right behind its start: Comm.timer) -- the measured receive rate the multi-GPU projection rests on."""
    comm = model.comm
    comm.overlap, comm.timer = False, []
    try:
        for _ in range(n_passes):
            fwd()
        torch.cuda.synchronize()
        return [(a.elapsed_time(b), nb) for a, b, nb in comm.timer]
    finally:
        comm.overlap, comm.timer = restore_overlap, None


def allgather_summary(ag, transport):
    """the `allgather` sub-dict of a bench line from allgather_timing()'s records"""
    if not ag:
        return None
    ms = sum(x for x, _ in ag) / len(ag)
    nb = sum(b for _, b in ag) / len(ag)
    rate = nb / max(ms * 1e-3, 1e-12) / 1e9
    return {"avg_us_serial": round(ms * 1e3, 2), "avg_bytes_received_per_rank": int(nb), "GBps_received_per_rank": round(rate, 1),
            "needed_GBps_for_6x_at_8_gpus_on_1e9_edges": NEEDED_RECEIVE_GBPS_FOR_6X, "transport": transport}


# ---------------------------------------------------------------------------------------------------------
# bench.py --gpus N (N > 1): strong scaling of the named workload over a node partition
# ---------------------------------------------------------------------------------------------------------
def bench_main(args, ng, rank, world, dev, wl, d, elem, emit, cpu_baseline_leg):
    """Strong scaling of the named workload over a node partition.  Order of events: (1) forward timed with eager
    launches (+ per-launch kernel events, + all-gather events in a serial pass), (2) train step timed with eager
    launches, (3) the forward again as ONE replayed hipGraph (kernels + RCCL launches), (4) the train step as one
    hipGraph.  (3)/(4) replace the eager numbers only if they complete; a watchdog prints the best line so far and
    exits if the captured path stalls, so a capture problem can never cost the measurement."""
    import json
    import os
    import sys
    import threading
    import time
    from . import benchkit as BK, data as D
    torch.manual_seed(2019)
    U, I, E_target = D.SHAPES[wl]
    big = wl in D.BIG_WORKLOADS

    def no_line():
        # nothing complete to report: a NON-ZERO status (the rc-0 exit further down is only for a stalled optional re-run
        # behind a complete eager line)
        print("[bench] rank %d: stalled before any complete measurement (a collective or a kernel never finished)" % rank,
              file=sys.stderr, flush=True)
        os._exit(3)
    dog0 = threading.Timer(float(os.environ.get("NGACF_DIST_EAGER_TIMEOUT", "900")), no_line)
    dog0.daemon = True
    dog0.start()
    if big:
        part, plan, E, sample_edge = sharded_graph(wl, world, rank, dev, seg_len=args.seg_len)
        model = ShardedSPUIGACF(U, I, d, [64, 64], args.droprate, world, rank, stages=args.stages, partition=part).to(dev)
        model.attach_plan(plan)
        edge = torch.zeros(2, 1, dtype=torch.long, device=dev)          # placeholder mask (the plan is attached)
    else:
        edge = D.workload_graph(wl, dev)                                # same graph on every rank (same device type)
        E = int(edge.shape[1])
        sample_edge = edge
        model = ShardedSPUIGACF(U, I, d, [64, 64], args.droprate, world, rank, stages=args.stages, adj=edge).to(dev)
        model.seg_len = args.seg_len
        plan = model._plan(edge)
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    L = len(model.gat._stages())
    model.comm.force = world == 1                 # one-rank rehearsal (NGACF_BENCH_FORCE_DIST=1): still issue the collectives
    model.comm.defer_weight_reduce = True
    model.comm.overlap = not args.no_overlap
    trav = L * 2 * E                              # all ranks together traverse every edge once per pass
    part = model.part

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    model.eval()
    fwd = lambda: model.propagate_local(edge)

    # ---- (1) forward, eager --------------------------------------------------------------------------------
    with torch.no_grad():
        model.comm.launches = 0
        fwd()
        launches_fwd = model.comm.launches
        t_pass = allmax(BK.pass_time(fwd, sync))
        R = R_eager = args.passes_per_step or BK.passes_per_step(t_pass, args.steps, args.target_seconds)
        for _ in range(args.warmup):
            fwd()
        t_eager, win_eager = BK.timed_steps(fwd, args.steps, R, sync, torch)
        t_eager = allmax(t_eager)
        # per-launch kernel durations (this rank) and all-gather durations (serial pass: start -> wait)
        n_probe = max(args.steps, 10 if big else 50)
        with _hip_ops.KernelTimer("ngacf_prop_fwd") as kt:
            for _ in range(n_probe):
                fwd()
            torch.cuda.synchronize()
        durs = kt.durations_ms()
        ag = allgather_timing(model, fwd, n_probe, not args.no_overlap)
    n_launch = len(durs) // n_probe                                          # 2 half-launches per stage, or 1 launch (whole-stage schedule)
    whole = model.comm.whole_stage(part.N * d * elem)
    stage_ms = sum(durs) / n_probe / L                                       # the launch(es) of a stage
    local_edges = plan.n_edges_directed
    bytes_stage = local_edges * (4 + d * elem) + part.n_local * (2 * d * elem + 8)
    achieved = -allmax(-bytes_stage / (stage_ms * 1e-3) / 1e9)               # the slowest rank's rate (min over ranks)
    regime = BK.gather_regime(part.N * d * elem, max(part.Up, part.Ip) * d * elem)
    ag_ms = sum(x for x, _ in ag) / max(len(ag), 1)
    ag_bytes = sum(nb for _, nb in ag) / max(len(ag), 1)
    roofline = {"bound": "hbm", "kernel": "prop_fwd_kernel (ngacf_prop_fwd: %s), slowest rank"
                                          % ("one launch per stage over the rank's rows" if whole else "2 half-launches per stage"),
                "achieved": round(achieved, 1), "peak": BK.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / BK.HBM_PEAK_GBS, 4),
                "traffic": None, "regime": regime["regime"],
                "applicable_roof_GBps": regime["roof_GBps"], "frac_of_applicable_roof": round(achieved / regime["roof_GBps"], 4),
                "bytes_per_stage_per_rank": int(bytes_stage), "avg_stage_kernel_us": round(stage_ms * 1e3, 2),
                "launches_per_forward": n_launch, "local_directed_edges": int(local_edges),
                "allgather": dict(allgather_summary(ag, "peer stores through hipIpc-mapped tables + flag words (NGACF_PEER_GATHER=1)"
                                                         if model.comm.peer_store else "RCCL all-gather (in place)") or
                                  {"avg_us_serial": round(ag_ms * 1e3, 2), "avg_bytes_received_per_rank": int(ag_bytes), "GBps_received_per_rank": None},
                                  per_forward=launches_fwd)}
    extra = {"ms_per_pass_eager": round(1e3 * t_eager / args.steps / R, 5), "roofline": roofline}
    model.comm.check()                            # (peer stores: no wait has timed out so far)

    def line(t_fwd, windows, launch, extra, R=None):
        R = R_eager if R is None else R
        med = BK.median(windows)
        out = {"metric": "propagation edges/sec (directed edge traversals, forward, all stages)",
               "value": round(trav * R * args.steps / t_fwd, 1), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(1e3 * t_fwd / args.steps, 4), "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "%s-scale bipartite graph U=%d I=%d E=%d, d=%d, L=%d stages, node-partitioned over %d GPUs"
                                      % (wl, U, I, E, d, L, world),
                          "passes_per_step": R, "edges_directed_per_pass": trav, "edges_directed_per_step": trav * R,
                          "launch": launch,
                          "parallelism": "degree-balanced node partition x%d, sharded plan (%d of %d directed edges on rank 0), "
                                         "%d %s all-gathers per forward, %s"
                                         % (world, local_edges, 2 * E, launches_fwd, "whole-table" if whole else "async half-table",
                                            "one gather + one launch per stage (small table: Comm.whole_below)" if whole else
                                            ("overlapped with the traversal of the other half" if model.comm.overlap else "serial"))},
               "ms_per_pass": round(1e3 * t_fwd / args.steps / R, 5), "timed_region_s": round(t_fwd, 4),
               "value_median_window": round(trav * R / (med * 1e-3), 1) if med > 0 else None,
               "roofline": None, "cpu_baseline": None}
        out.update(extra)
        return json.dumps(out)

    # ---- (2) train step, eager ---------------------------------------------------------------------------------
    if not args.no_train:
        model.train()
        lossfn = ng.BPRLoss()
        from .optim import Adam as _Adam
        opt = _Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        g = torch.Generator(device=dev)
        g.manual_seed(7)                                                  # same batch on every rank
        B = args.batch
        ub = torch.randint(0, U, (B,), device=dev, generator=g)
        pb = torch.randint(0, I, (B,), device=dev, generator=g)
        nb = torch.randint(0, I, (B,), device=dev, generator=g)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = lossfn(*model.forward_two(ub, pb, nb, edge))            # the step's two propagations (lock-step where the whole-stage schedule applies)
            loss.backward()
            model.reduce_grads()                                          # one grouped all-reduce of the stage weights
            opt.step()
        model.collective_launches(reset=True)
        step()
        per_step = model.collective_launches()
        sync()
        t0 = time.perf_counter()
        step()
        sync()
        ksteps = int(max(min(args.steps, 20), min(500, 0.3 / max(allmax(time.perf_counter() - t0), 1e-6))))
        sync()
        t0 = time.perf_counter()
        for _ in range(ksteps):
            step()
        sync()
        t_tr = allmax(time.perf_counter() - t0)
        extra.update({"bpr_pairs_per_s": round(B * ksteps / t_tr, 1), "train_ms_per_step": round(1e3 * t_tr / ksteps, 3),
                      "train_steps": ksteps, "batch": B, "collective_launches_per_train_step": per_step})
        model.eval()

    # ---- CPU baseline (rank 0; the others wait at the next barrier) ---------------------------------------------
    if not args.no_cpu_baseline:
        names = [["gat." + n for n in nm] for nm in model.gat.stage_layer_names()]
        sd_full = None if big else model.full_state_dict()       # collective: every rank calls it
        if rank == 0:
            try:
                if big:                            # the sample's rows only: never the whole table on one rank
                    su = int(sample_edge[0].max()) + 1
                    items, inv = torch.unique(sample_edge[1], return_inverse=True)
                    e_c = torch.stack([sample_edge[0], inv]).cpu()
                    sdc = {"gat." + k: v.detach().float().cpu() for k, v in model.gat.state_dict().items()}
                    gen = torch.Generator().manual_seed(1)
                    sdc["uEmbd.weight"] = torch.empty(su, d).normal_(std=0.01, generator=gen)
                    sdc["iEmbd.weight"] = torch.empty(int(items.numel()), d).normal_(std=0.01, generator=gen)
                    extra["cpu_baseline"] = cpu_baseline_leg(torch, sdc, names, e_c, su, int(items.numel()), L, args.cpu_budget,
                                                             "the subgraph of the first %d users of the %s graph (%d edges, %d items touched)"
                                                             % (su, wl, int(e_c.shape[1]), int(items.numel())))
                else:
                    extra["cpu_baseline"] = cpu_baseline_leg(torch, {k: v.float().cpu() for k, v in sd_full.items()}, names,
                                                             edge.cpu(), U, I, L, args.cpu_budget, "the whole %s graph" % wl)
            except Exception as e:                 # the baseline must not cost the measurement
                print("[bench] cpu_baseline failed: %r" % (e,), file=sys.stderr, flush=True)
        dist.barrier()

    eager_line = line(t_eager, win_eager, "eager", extra)
    dog0.cancel()
    if getattr(args, "eager", False) or os.environ.get("NGACF_DIST_GRAPH") == "0":
        if rank == 0:
            emit(eager_line)
        dist.destroy_process_group()
        return

    # ---- (3) forward as one hipGraph, under a watchdog -----------------------------------------------------------
    lock, done = threading.Lock(), []

    def finish(text, hard, tag=None):
        with lock:
            if not done:
                done.append(1)
                if rank == 0:
                    if tag:
                        text = json.dumps(dict(json.loads(text), graph_fallback=tag))
                    emit(text)
        if hard:
            # A rank may be stuck inside a collective: leave without teardown.  Exit status 0: the eager measurement in
            # the line is complete and valid; what failed is the optional captured re-run, and the line says so
            # (graph_fallback) instead of the process status.
            os._exit(0)

    timeout = float(os.environ.get("NGACF_DIST_GRAPH_TIMEOUT", "120"))
    dog = threading.Timer(timeout, finish, (eager_line, True, "timeout: hipGraph capture/replay of the sharded forward stalled"))
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
            from .graphs import quiesce_collectives
            graph = torch.cuda.CUDAGraph()
            # thread-local capture mode: the c10d watchdog thread may query events while this thread captures -- and it must have
            # retired every eager collective before c10d's stream joins the capture (graphs.quiesce_collectives)
            quiesce_collectives()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                fwd()
            for _ in range(args.warmup):
                graph.replay()
            # the replayed pass is shorter than the eager one: its own pass count
            R_graph = args.passes_per_step or BK.passes_per_step(allmax(BK.pass_time(graph.replay, sync)), args.steps, args.target_seconds)
            t_graph, win_graph = BK.timed_steps(graph.replay, args.steps, R_graph, sync, torch)
            t_graph = allmax(t_graph)
            model.comm.check()
    except Exception as e:                         # capture refused here (maybe not on the peers): the eager numbers stand
        print("[bench] sharded hipGraph capture not used: %r" % (e,), file=sys.stderr, flush=True)
        finish(eager_line, True, "capture failed: %r" % (e,))
    better = t_graph / R_graph < t_eager / R_eager    # t_* are already max-over-ranks: every rank decides alike
    glaunch = "hipGraph replay (kernels + %s)" % ("peer-store push / wait kernels" if model.comm.peer_store else "RCCL launches")
    best = line(t_graph, win_graph, glaunch, extra, R_graph) if better else eager_line
    # ---- (4) the train step as one hipGraph (sampler-free: fixed batch, as in (2)), same watchdog ------------------
    if not args.no_train:
        dog.cancel()
        dog = threading.Timer(timeout, finish, (best, True, "timeout: hipGraph capture/replay of the sharded train step stalled"))
        dog.daemon = True
        dog.start()
        try:
            from .graphs import GraphedBPRStep
            model.train()
            opt_g = _Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
            gs = GraphedBPRStep(model, lossfn, opt_g, edge, B, concurrent_forwards=model.concurrent_chains, after_backward=model.reduce_grads)
            upn = torch.stack([ub, pb, nb])
            for _ in range(3):
                gs(upn)
            sync()
            t0 = time.perf_counter()
            for _ in range(ksteps):
                gs(upn)
            sync()
            t_trg = allmax(time.perf_counter() - t0)
            if t_trg < t_tr:
                extra = dict(extra, bpr_pairs_per_s=round(B * ksteps / t_trg, 1), train_ms_per_step=round(1e3 * t_trg / ksteps, 3),
                             train_launch="hipGraph replay (kernels + RCCL launches + fused Adam; %s)"
                                          % ("the two propagation chains as concurrent branches on two communicators" if model.concurrent_chains
                                             else ("both chains in lock-step on one queue, one grouped all-gather per stage for the two"
                                                   if model.comm.whole_stage(part.N * d * elem) else "one chain after the other")),
                             train_ms_per_step_eager=round(1e3 * t_tr / ksteps, 3))
                best = line(t_graph, win_graph, glaunch, extra, R_graph) if better else line(t_eager, win_eager, "eager", extra)
        except Exception as e:
            print("[bench] sharded hipGraph train step not used: %r" % (e,), file=sys.stderr, flush=True)
            finish(best, True, "train-step capture failed: %r" % (e,))
    dog.cancel()
    # ---- (5) the scaling probe on the 10^9-edge graph (default workload only), same watchdog ---------------------
    if wl == "gowalla" and not args.no_probe:
        dog = threading.Timer(float(os.environ.get("NGACF_PROBE_TIMEOUT", "240")), finish,
                              (best, True, "timeout: the scaling probe stalled (headline numbers complete)"))
        dog.daemon = True
        dog.start()
        try:
            probe = scaling_probe(args, world, rank, dev, d)
            best = json.dumps(dict(json.loads(best), scaling_probe=probe))
        except Exception as e:
            print("[bench] scaling probe failed: %r" % (e,), file=sys.stderr, flush=True)
            best = json.dumps(dict(json.loads(best), scaling_probe={"error": repr(e)}))
        dog.cancel()
    finish(best, False)
    dist.destroy_process_group()
