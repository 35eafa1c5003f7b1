"""Checked wrappers around the C ABI (include/ngacf.h).  Each wrapper verifies device,
dtype, contiguity and element counts of every tensor before a raw pointer crosses the
boundary -- a wrong size would be an out-of-bounds access on the GPU, not a Python error --
then launches on the caller's current stream.  No torch math happens here.
"""
import ctypes

import torch

from . import _lib
from ._lib import BF16, F32, FLAG_ELU, FLAG_RESIDUAL, FLAG_SCORE_ROW, lib

_DT = {torch.float32: F32, torch.bfloat16: BF16}
SUPPORTED_D = (64, 128)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device (the C-level getters cost ~0.3 us; the
    torch.cuda.current_stream() object path ~5 us, which matters when a forward is ~20 launches)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Opt-in per-KERNEL timing (bench.py's roofline leg): inside the block the traversal and transform launches of this
    process (any host thread: autograd's backward runs on its own) carry start / stop events of their dispatch packets (ngacf_timing_begin / _end, hipExtLaunchKernel) -- the
    kernel's own duration, without the host's launch gaps; it is the number a rocprofv3 kernel trace reports (an event pair
    AROUND a launch call read 5-8 us longer in round 2).  Eager launches only (not inside a stream capture).
      with KernelTimer() as t: ...;   t.durations_ms(TAG_PROP_FWD) -> [ms per launch, in launch order]"""
    TAG_PROP_FWD, TAG_PROP_FWD_SPLIT, TAG_TRANSFORM_FWD, TAG_PROP_BWD = 1, 2, 3, 4

    def __init__(self, name=None, cap=1 << 20):
        self.cap, self._res = cap, None

    def __enter__(self):
        _lib.check(lib.ngacf_timing_begin(), "ngacf_timing_begin")
        return self

    def __exit__(self, *exc):
        ms = (ctypes.c_float * self.cap)()
        tags = (ctypes.c_int32 * self.cap)()
        n = lib.ngacf_timing_end(ms, tags, self.cap)
        if n < 0:
            raise _lib.NgacfError("ngacf_timing_end failed: %d" % n)
        self._res = [(tags[k], ms[k]) for k in range(n)]

    def durations_ms(self, tag=None):
        """tag None: one entry per ngacf_prop_fwd CALL = its main kernel + the hub rows' second kernel where one ran (plans
        beyond INKERNEL_MAX_PARTS); a tag: that kernel family alone."""
        if tag is not None:
            return [d for t, d in self._res if t == tag]
        out = []
        for t, d in self._res:
            if t == self.TAG_PROP_FWD:
                out.append(d)
            elif t == self.TAG_PROP_FWD_SPLIT and out:
                out[-1] += d
        return out


def _chk(t, name, dtype, numel, dev, align=16):
    if t is None:
        raise ValueError("%s is required" % name)
    if not t.is_cuda or t.device != dev:
        raise ValueError("%s must live on %s, is on %s" % (name, dev, t.device))
    if dtype is not None and t.dtype != dtype:
        raise ValueError("%s must be %s, is %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if t.numel() != numel:
        raise ValueError("%s must have %d elements, has %d" % (name, numel, t.numel()))
    if t.data_ptr() % align:
        raise ValueError("%s must be %d-byte aligned" % (name, align))
    return t.data_ptr()


def _heads_ok(d, heads):
    if d not in SUPPORTED_D or heads not in (1, d // 8):
        raise ValueError("unsupported row width / heads: d=%d heads=%d (d in %s, heads in {1, d/8})" % (d, heads, SUPPORTED_D))


def _sd(seed_dev, dev):
    """optional device-side seed addend: int64[1] on `dev`"""
    if seed_dev is None:
        return 0
    return _chk(seed_dev, "seed_dev", torch.int64, 1, dev)


def transform_fwd(X, W, avec, U, I, heads, h_dtype=torch.float32, feat_drop_p=0.0, seed=0, seed_dev=None, pack=None, pack_kind=None):
    """H[N,dout], S[N,heads] = ngacf_transform_fwd(X[N,din], W[2,din,dout], avec[2,dout]).
    pack = [W_k float[2,64,dout_k], ...] (bf16 storage only): the same launch also writes those stages' weight panels as
    matrix-core fragments (ngacf_transform_fwd_pack); returns (H, S, [panel_k uint8 tensors]) then.  pack_kind[k] = 0: the
    forward fragments (prop_fwd(next=...)); 1: the fragments of dX = dH @ W^T (prop_bwd(dx=...))."""
    dev, N = X.device, U + I
    din, dout = W.shape[1], W.shape[2]
    _heads_ok(dout, heads)
    if din not in SUPPORTED_D:
        raise ValueError("unsupported din=%d" % din)
    xp = _chk(X, "X", None, N * din, dev)
    if X.dtype not in _DT:
        raise ValueError("X dtype %s" % X.dtype)
    wp = _chk(W, "W", torch.float32, 2 * din * dout, dev)
    ap = _chk(avec, "avec", torch.float32, 2 * dout, dev)
    H = torch.empty(N, dout, dtype=h_dtype, device=dev)
    S = torch.empty(N, heads, dtype=torch.float32, device=dev)
    if pack:
        if h_dtype != torch.bfloat16 or len(pack) > _lib.MAX_PACK:
            raise ValueError("transform_fwd(pack=...): bf16 storage, at most %d panels" % _lib.MAX_PACK)
        n = len(pack)
        for Wk in pack:
            if Wk.dim() != 3 or Wk.shape[0] != 2 or Wk.shape[1] != 64 or Wk.shape[2] not in (64, 128):
                raise ValueError("pack: W [2, 64, 64 | 128], got %s" % (tuple(Wk.shape),))
        panels = [torch.empty(2 * Wk.shape[1] * Wk.shape[2] * 2, dtype=torch.uint8, device=dev) for Wk in pack]
        P, Iarr = ctypes.c_void_p * n, ctypes.c_int * n
        pw = P(*[_chk(Wk, "pack W", torch.float32, Wk.numel(), dev) for Wk in pack])
        pf = P(*[t.data_ptr() for t in panels])
        kinds = [0] * n if pack_kind is None else [int(k) for k in pack_kind]
        if len(kinds) != n or any(k not in (0, 1) for k in kinds) or any(k == 1 and Wk.shape[2] != 64 for k, Wk in zip(kinds, pack)):
            raise ValueError("pack_kind: one of 0 / 1 per panel; kind 1 for [2, 64, 64] weights")
        _lib.check(lib.ngacf_transform_fwd_pack(xp, _DT[X.dtype], wp, ap, H.data_ptr(), _DT[h_dtype], S.data_ptr(), U, I, din, dout, heads,
                                                float(feat_drop_p), int(seed), _sd(seed_dev, dev), n, pw, pf,
                                                Iarr(*[int(Wk.shape[1]) for Wk in pack]), Iarr(*[int(Wk.shape[2]) for Wk in pack]),
                                                Iarr(*kinds), _stream()),
                   "ngacf_transform_fwd_pack")
        return H, S, panels
    _lib.check(lib.ngacf_transform_fwd(xp, _DT[X.dtype], wp, ap, H.data_ptr(), _DT[h_dtype], S.data_ptr(),
                                       U, I, din, dout, heads, float(feat_drop_p), int(seed), _sd(seed_dev, dev),
                                       _stream()), "ngacf_transform_fwd")
    return H, S


class NextStage:
    """The next stage's transform riding in this stage's traversal epilogue (ngacf_prop_fwd_next): its weight panel (bf16 storage:
    from transform_fwd(pack=...); fp32 storage: the fp32 W [2, 64, dout] itself), avec [2, dout], heads, feature dropout and seed.
    prop_fwd(next=...) fills .H / .S."""
    __slots__ = ("panel", "avec", "dout", "heads", "feat_drop_p", "seed", "H", "S")

    def __init__(self, panel, avec, heads, feat_drop_p=0.0, seed=0, H=None, S=None):
        self.panel, self.avec, self.dout, self.heads = panel, avec, int(avec.shape[1]), int(heads)
        self.feat_drop_p, self.seed, self.H, self.S = float(feat_drop_p), int(seed), H, S


def can_fuse_next(plan, H_dtype, d, heads, needed=None):
    """Can ngacf_prop_fwd_next serve this launch?  (rows of 64 columns in 8 heads, hub rows finished in-kernel or absent)"""
    return (H_dtype in _DT and d == 64 and heads == 8 and needed is None and plan.nhub == 0
            and (plan.nsplit == 0 or (INKERNEL_HUB_FINISH and plan.nparts <= INKERNEL_MAX_PARTS)))


def _plan_ptrs(plan, dev, half=None, inkernel=False):
    """The plan arguments of ngacf_prop_fwd / ngacf_prop_bwd.  half = "user" / "item": only that half of the launch's
    work order ([user long segments][user short rows] / [item long segments][item short rows]) -- the lists are
    contiguous per half, so a half is a pointer offset and a count; partial-sum slots keep their global numbers.
    The node-partitioned path launches the halves separately to run one under the other's all-gather."""
    if plan.device != dev:
        raise ValueError("plan is on %s, tensors on %s" % (plan.device, dev))
    segs = _chk(plan.segs, "plan.segs", torch.int32, plan.nseg * 4, dev) if plan.nseg else 0
    split = _chk(plan.split, "plan.split", torch.int32, plan.nsplit * 3, dev) if plan.nsplit else 0
    col = _chk(plan.col, "plan.col", torch.int32, plan.n_edges_directed, dev) if plan.n_edges_directed else 0
    srow = _chk(plan.srec, "plan.srec", torch.int32, plan.nshort * 4, dev) if plan.nshort else 0
    rowptr = _chk(plan.rowptr32, "plan.rowptr32", torch.int32, plan.N + 1, dev)
    hrow = _chk(plan.hrow, "plan.hrow", torch.int32, plan.nhub, dev, align=4) if plan.nhub else 0
    if half is None:
        return (segs, plan.nseg, split, plan.nsplit, srow, plan.nshort, plan.nseg_user, plan.nshort_user,
                hrow, plan.nhub, plan.nhub_user, rowptr, col)
    if inkernel:        # hub rows finished by their last slice: the kernel looks rows up in the WHOLE split table (slot -> row
        su = si = (split, plan.nsplit)              # indices are global) and no second kernel walks a half's share of it
    else:
        su = (split, plan.nsplit_user)
        nsp = plan.nsplit - plan.nsplit_user
        si = (split + 12 * plan.nsplit_user if nsp else 0, nsp)
    if half == "user":
        return (segs, plan.nseg_user, su[0], su[1], srow, plan.nshort_user, plan.nseg_user, plan.nshort_user,
                hrow, plan.nhub_user, plan.nhub_user, rowptr, col)
    if half == "item":
        ns, nsh = plan.nseg - plan.nseg_user, plan.nshort - plan.nshort_user
        nh = plan.nhub - plan.nhub_user
        return (segs + 16 * plan.nseg_user if ns else 0, ns, si[0], si[1],
                srow + 16 * plan.nshort_user if nsh else 0, nsh, 0, 0, hrow + 4 * plan.nhub_user if nh else 0, nh, 0, rowptr, col)
    raise ValueError("half must be None, 'user' or 'item'")


# Hub rows are finished by their last-arriving slice INSIDE the launch (write-through stores of the partial sums, an
# agent-scope ticket per row, agent-scope atomic loads by the last arriver -- no fence, no waiting; bitwise equal to the
# two-kernel path, tests/test_gpu_parity.py): the forward's serial chain loses one 5 us kernel per stage (0.126 -> 0.115 ms
# per 3-stage pass).  An earlier form with an acquire FENCE in the last arriver invalidated the XCD's L2 under a kernel
# that lives on its cache hits and cost what it saved (31.5 us against 27 + 4.9).  NGACF_INKERNEL_HUB=0: the plain second
# kernel (also used for plans with very many slices, INKERNEL_MAX_PARTS).
INKERNEL_HUB_FINISH = __import__("os").environ.get("NGACF_INKERNEL_HUB", "1") != "0"


INKERNEL_MAX_PARTS = 1 << 18      # beyond this many slices the second kernel wins: the 10^9-edge graph (1.27 M hub rows, 11 M
                                  # segments) runs a stage in 35.6 ms with it, 36.8 ms with write-through stores + tickets


def _hub(plan, dev, half=None):
    if not INKERNEL_HUB_FINISH or plan.nsplit == 0 or plan.nparts > INKERNEL_MAX_PARTS:
        return 0, 0
    return (_chk(plan.part_split, "plan.part_split", torch.int32, plan.nparts, dev),
            _chk(plan.tickets_for(_stream()), "plan.tickets", torch.int32, plan.nsplit, dev, align=4))


class BatchRows:
    """The rows of the propagated table a scored batch reads (SPUIGACF.py:49-52): one flag bit per row (int32 words) and the
    same rows as an int32 list (first `n` entries; duplicates allowed), both written by ngacf_mark_rows."""
    __slots__ = ("needed", "rows", "n", "U", "I", "g32", "consumed")

    def __init__(self, needed, rows, n, U, I, g32=None):
        self.needed, self.rows, self.n, self.U, self.I = needed, rows, n, U, I
        self.g32 = g32          # float [U+I, d], ZERO at the batch's rows (the rest is never read): where the scores' gradient lands
        self.consumed = False   # one backward per marking: a second one would add to the first one's gradient rows

    def mask(self):
        """bool [U+I] (tests, debug asserts)"""
        r = torch.arange(self.U + self.I, device=self.needed.device)
        return ((self.needed[r >> 5] >> (r & 31)) & 1).bool()


def mark_rows(u, i, U, I, into=None, grad_d=0):
    """BatchRows of the pairs (u[b], i[b]).  `into` given: ADD the items i to it (u is ignored: forward_pairs marks
    (user, positive) first, then the negatives of the same users).  grad_d = d: also a float [U+I, d] table whose batch
    rows are zeroed (BatchRows.g32), for the backward of the scores."""
    dev, B = i.device, i.numel()
    if into is None:
        into = BatchRows(torch.empty(((U + I + 31) // 32 + 15) // 16 * 16, dtype=torch.int32, device=dev)[:(U + I + 31) // 32], torch.empty(max(3 * B, 1), dtype=torch.int32, device=dev), 0, U, I,
                         torch.empty(U + I, grad_d, dtype=torch.float32, device=dev) if grad_d else None)
        up, clear, add = (_idx(u, "userIdx", B, dev) if B else 0), 1, 2 * B
    else:
        if into.n + B > into.rows.numel() or (into.U, into.I) != (U, I):
            raise ValueError("mark_rows: the row list has no room for %d more rows" % B)
        up, clear, add = 0, 0, B
    _lib.check(lib.ngacf_mark_rows(up, _idx(i, "itemIdx", B, dev) if B else 0, _chk(into.needed, "needed", torch.int32, (U + I + 31) // 32, dev, align=4),
                                   into.rows.data_ptr() + 4 * into.n, into.g32.data_ptr() if into.g32 is not None else 0, B, U, I,
                                   into.g32.shape[1] if into.g32 is not None else 4, clear, _stream()), "ngacf_mark_rows")
    into.n += add
    return into


def _needed_args(needed, plan, N, dev):
    if needed is None:
        return 0, 0, 0, 0
    if needed.U + needed.I != N:
        raise ValueError("needed: row flags of a %d-row table for a table of %d rows" % (needed.U + needed.I, N))
    return _chk(needed.needed, "needed", torch.int32, (N + 31) // 32, dev, align=4), needed.rows.data_ptr(), needed.n, int(plan.short_max)


def _scratch(scratch, plan, d, heads, dev, planes=1):
    """the hub rows' partial-sum slots; planes = 2 for a forward that saves (a slot then also holds the negative-side sums)"""
    if scratch is None:
        return (torch.empty(max(plan.nparts, 1), planes * d, dtype=torch.float32, device=dev),
                torch.empty(max(plan.nparts, 1), planes * heads, dtype=torch.float32, device=dev))
    pa, pr = scratch
    _chk(pa, "scratch[0]", torch.float32, max(plan.nparts, 1) * planes * d, dev)
    _chk(pr, "scratch[1]", torch.float32, max(plan.nparts, 1) * planes * heads, dev, align=4)
    return pa, pr


def prop_fwd(plan, H, S_dst, S_src, avec, heads, alpha=0.2, elu=True, residual=True, score_from_row=None,
             drop_p=0.0, seed=0, save=False, out=None, seed_dev=None, half=None, agg=None, R=None, needed=None, scratch=None,
             next=None, want_out=True, roof=0):
    """One propagation stage over plan's dst rows.  Returns out (and agg [2, N, d], R [2, N, heads] when save=True: plane 0 = the
    normalised aggregate and the denominators, plane 1 = the same sums over the edges with a non-positive score, which the
    backward's row-local half of dS needs -- include/ngacf.h).
    half = "user" / "item": only the user / item dst rows (two calls with the same out / agg / R cover the stage).
    scratch = (float [nparts, d], float [nparts, heads]) ([nparts, 2 d], [nparts, 2 heads] with save): the hub rows' partial-sum
    slots (default: fresh buffers; the soak test hands in buffers it has poisoned).
    next = NextStage: the launch also produces the next stage's H' [N, dout'] (bf16) and S' [N, heads'] in next.H / next.S; with
    want_out=False (an evaluation forward) `out` is not written at all and None is returned in its place.
    roof = 1 / 2: MEASUREMENT ONLY (tools/gather_roof.py) -- the stripped instantiation (NGACF_ROOF_ADD / NGACF_ROOF_XOR): the same
    gathers and stores, no attention arithmetic; `out` is meaningless."""
    dev, N, d = H.device, plan.N, H.shape[1]
    if roof and (save or next is not None or needed is not None or drop_p or d != 64 or roof not in (1, 2)):
        raise ValueError("prop_fwd(roof=...): the plain eval launch of a 64-column stage only")
    _heads_ok(d, heads)
    if score_from_row is None:
        score_from_row = True      # recomputing a.h from the gathered row beats a second gather (profiles/)
    hp = _chk(H, "H", None, N * d, dev)
    if H.dtype not in _DT:
        raise ValueError("H dtype %s" % H.dtype)
    sd = _chk(S_dst, "S_dst", torch.float32, N * heads, dev)
    ss = _chk(S_src, "S_src", torch.float32, N * heads, dev)
    ap = _chk(avec, "avec", torch.float32, 2 * d, dev)
    hub = _hub(plan, dev, half)
    pp = _plan_ptrs(plan, dev, half, inkernel=hub[1] != 0)
    if next is not None and (not can_fuse_next(plan, H.dtype, d, heads, needed) or not score_from_row):
        raise ValueError("prop_fwd(next=...): not available for this launch (ops.can_fuse_next)")
    if next is None or want_out:
        if out is None:
            out = torch.empty(N, d, dtype=H.dtype, device=dev)
        op = _chk(out, "out", H.dtype, N * d, dev)
    else:
        out, op = None, 0
    if save:
        agg = torch.empty(2, N, d, dtype=H.dtype, device=dev) if agg is None else agg    # (storage type: bf16 tables halve it)
        R = torch.empty(2, N, heads, dtype=torch.float32, device=dev) if R is None else R
        _chk(agg, "agg", H.dtype, 2 * N * d, dev)
        _chk(R, "R", torch.float32, 2 * N * heads, dev)
    else:
        agg = R = None
    pa, pr = _scratch(scratch, plan, d, heads, dev, 2 if save else 1)
    flags = (FLAG_ELU if elu else 0) | (FLAG_RESIDUAL if residual else 0) | (FLAG_SCORE_ROW if score_from_row else 0) | _lib.FLAG_ROOF[roof]
    if next is not None:
        # (a half launch of the node-partitioned path hands in the global-row tables of the next stage: both halves fill them)
        if next.H is None:
            next.H = torch.empty(N, next.dout, dtype=H.dtype, device=dev)
            next.S = torch.empty(N, next.heads, dtype=torch.float32, device=dev)
        _chk(next.H, "next.H", H.dtype, N * next.dout, dev)
        _chk(next.S, "next.S", torch.float32, N * next.heads, dev)
        panel = next.panel.get() if hasattr(next.panel, "get") else next.panel
        # bf16 storage: the packed fragments of transform_fwd(pack=...); fp32 storage: the next stage's fp32 W [2, 64, dout] itself
        pdt, pn = (torch.uint8, 2 * 64 * next.dout * 2) if H.dtype == torch.bfloat16 else (torch.float32, 2 * 64 * next.dout)
        ns = _lib.NextStage(_chk(panel, "next.panel", pdt, pn, dev), _chk(next.avec, "next.avec", torch.float32, 2 * next.dout, dev),
                            next.H.data_ptr(), next.S.data_ptr(), next.dout, next.heads, next.feat_drop_p, next.seed & 0xFFFFFFFFFFFFFFFF)
        _lib.check(lib.ngacf_prop_fwd_next(*pp, hp,
                          _DT[H.dtype], sd, ss, ap, op, agg.data_ptr() if save else 0, R.data_ptr() if save else 0,
                          pa.data_ptr(), pr.data_ptr(), *hub, 0, 0, 0, 0, plan.U, N, d, heads, float(alpha),
                          float(drop_p), int(seed), _sd(seed_dev, dev), flags, ctypes.byref(ns), _stream()), "ngacf_prop_fwd_next")
        return (out, agg, R) if save else out
    _lib.check(lib.ngacf_prop_fwd(*pp, hp,
                      _DT[H.dtype], sd, ss, ap, op, agg.data_ptr() if save else 0, R.data_ptr() if save else 0,
                      pa.data_ptr(), pr.data_ptr(), *hub, *_needed_args(needed, plan, N, dev),
                      plan.U, N, d, heads, float(alpha),
                      float(drop_p), int(seed), _sd(seed_dev, dev), flags, _stream()), "ngacf_prop_fwd")
    return (out, agg, R) if save else out


class BwdDx:
    """The stage's transform backward riding in its edge backward (ngacf_prop_bwd_dx): `panel` = the kind-1 fragments of the
    stage's W [2, 64, 64] (transform_fwd(pack=..., pack_kind=1) / pack_panels), x_dtype = the type of the stage's input X (dX's),
    the stage's feature dropout and seed.  prop_bwd(dx=...) fills .dX."""
    __slots__ = ("panel", "x_dtype", "feat_drop_p", "seed", "dX")

    def __init__(self, panel, x_dtype, feat_drop_p=0.0, seed=0):
        self.panel, self.x_dtype, self.feat_drop_p, self.seed = panel, x_dtype, float(feat_drop_p), int(seed)
        self.dX = None


# The backward's launch plan: 0 = dX and the node steps as their own launches; 1 (default) = the stage's dX in the edge backward's tail.
# Round 4 had a level 2 (+ the previous stage's step 1 in the same tail); three runs each of the replayed two-chain step (round 5,
# Gowalla bf16): 0.6220 / 0.6180 / 0.6194 ms for 0 / 1 / 2, Yelp 0.731 (1) against 0.743 (2) -- the step is bound by the sum of its
# traversal launches, the extra fusion bought nothing, and level 2 was deleted (kernel tail, C struct, tests).
FUSE_BWD = min(int(__import__("os").environ.get("NGACF_FUSE_BWD", "1")), 1)


def can_fuse_bwd(plan, H_dtype, d, din, x_dtype):
    """Can ngacf_prop_bwd_dx serve this stage's backward?  (bf16 rows, a 64 -> 64 stage, the whole graph in the plan, hub rows
    finished in-kernel or absent)"""
    return (bool(FUSE_BWD) and H_dtype == torch.bfloat16 and d == 64 and din == 64 and plan.nhub == 0 and plan.n_rows == plan.N
            and (plan.nsplit == 0 or (INKERNEL_HUB_FINISH and plan.nparts <= INKERNEL_MAX_PARTS))
            and x_dtype in (torch.bfloat16, torch.float32))


def prop_bwd(plan, H, S_dst, S_src, avec, out, agg, R, gout, heads, alpha=0.2, elu=True, residual=True,
             score_from_row=None, drop_p=0.0, seed=0, seed_dev=None, needed=None, gout_f32=False, scratch=None, dx=None, node=None, roof=0):
    """Returns dH [N,d] in H's dtype (with the dS*avec term) and dS [N,heads] fp32.  agg [2, N, d], R [2, N, heads]: as
    prop_fwd(save=True) returned them.  scratch: as in prop_fwd (one plane).
    node = (Gn [N,d] in H's dtype, ST float [N,heads,2], dS float [N,heads] holding P1): step 1 has been run elsewhere (a later
    stage's prop_bwd(dx=...)) on this gout; that dS is finished in place and returned.
    score_from_row: accepted for symmetry with prop_fwd and ignored -- the backward reads a neighbour's score from its (S, t) pair.
    dx = BwdDx: the launch also writes the gradient of the stage's input (dx.dX).
    roof = 1 / 2: MEASUREMENT ONLY (tools/gather_roof.py): the edge step stripped to its gathers and stores (results meaningless)."""
    dev, N, d = H.device, plan.N, H.shape[1]
    if roof and (dx is not None or needed is not None or drop_p or d != 64 or roof not in (1, 2)):
        raise ValueError("prop_bwd(roof=...): the dense edge backward of a 64-column stage, alone")
    _heads_ok(d, heads)
    hp = _chk(H, "H", None, N * d, dev)
    sd = _chk(S_dst, "S_dst", torch.float32, N * heads, dev)
    ss = _chk(S_src, "S_src", torch.float32, N * heads, dev)
    ap = _chk(avec, "avec", torch.float32, 2 * d, dev)
    op = _chk(out, "out", H.dtype, N * d, dev)
    gp = _chk(agg, "agg", H.dtype, 2 * N * d, dev)
    rp = _chk(R, "R", torch.float32, 2 * N * heads, dev)
    go = _chk(gout, "gout", torch.float32 if gout_f32 else H.dtype, N * d, dev)
    pp = _plan_ptrs(plan, dev)
    f32 = dict(dtype=torch.float32, device=dev)
    dH = torch.empty(N, d, dtype=H.dtype, device=dev)      # (no G table: step 2 recomputes g = gout * ELU'(out))
    if node is None:
        Gn, st, dS = torch.empty(N, d, dtype=H.dtype, device=dev), torch.empty(N, heads, 2, **f32), torch.empty(N, heads, **f32)
    else:
        Gn, st, dS = node
        _chk(Gn, "node Gn", H.dtype, N * d, dev)
        _chk(st, "node ST", torch.float32, N * heads * 2, dev)
        _chk(dS, "node dS", torch.float32, N * heads, dev)
    pa, pr = _scratch(scratch, plan, d, heads, dev)
    if plan.n_rows != N:
        dH.zero_()
        if node is None:
            dS.zero_()
    flags = ((FLAG_ELU if elu else 0) | (FLAG_RESIDUAL if residual else 0) | FLAG_SCORE_ROW |
             (_lib.FLAG_GOUT_F32 if gout_f32 else 0) | (_lib.FLAG_SKIP_NODE if node is not None else 0) | _lib.FLAG_ROOF[roof])
    if dx is not None:
        din = 64
        if not can_fuse_bwd(plan, H.dtype, d, din, dx.x_dtype):
            raise ValueError("prop_bwd(dx=...): not available for this launch (ops.can_fuse_bwd)")
        dx.dX = torch.empty(N, din, dtype=dx.x_dtype, device=dev)
        panel = dx.panel.get() if hasattr(dx.panel, "get") else dx.panel
        st_ = _lib.BwdDx(_chk(panel, "dx.panel", torch.uint8, 2 * din * d * 2, dev), dx.dX.data_ptr(), din, _DT[dx.x_dtype],
                         dx.feat_drop_p, dx.seed & 0xFFFFFFFFFFFFFFFF)
        _lib.check(lib.ngacf_prop_bwd_dx(*pp, hp, _DT[H.dtype], sd, ss, ap,
                                         op, gp, rp, go, 0, Gn.data_ptr(), st.data_ptr(), dH.data_ptr(),
                                         dS.data_ptr(), pa.data_ptr(), pr.data_ptr(), *_hub(plan, dev), *_needed_args(needed, plan, N, dev)[:3], plan.U, N, d, heads, float(alpha),
                                         float(drop_p), int(seed), _sd(seed_dev, dev), flags, ctypes.byref(st_), _stream()), "ngacf_prop_bwd_dx")
        return dH, dS
    _lib.check(lib.ngacf_prop_bwd(*pp, hp, _DT[H.dtype], sd, ss, ap,
                                  op, gp, rp, go, 0, Gn.data_ptr(), st.data_ptr(), dH.data_ptr(),
                                  dS.data_ptr(), pa.data_ptr(), pr.data_ptr(), *_hub(plan, dev), *_needed_args(needed, plan, N, dev)[:3], plan.U, N, d, heads, float(alpha),
                                  float(drop_p), int(seed), _sd(seed_dev, dev), flags, _stream()), "ngacf_prop_bwd")
    return dH, dS


def transform_bwd(X, W, H, dH, dS, U, I, heads, feat_drop_p=0.0, seed=0, seed_dev=None, part="both"):
    """Returns dX [N,din] (X's dtype), dW [2,din,dout], davec [2,dout] (fp32).  part = "dx" / "dw": only that half
    (the other results are None) -- the halves are independent and may run on different streams."""
    dev, N = X.device, U + I
    din, dout = W.shape[1], W.shape[2]
    _heads_ok(dout, heads)
    xp = _chk(X, "X", None, N * din, dev)
    wp = _chk(W, "W", torch.float32, 2 * din * dout, dev)
    hp = _chk(H, "H", None, N * dout, dev)
    gp = _chk(dH, "dH", H.dtype, N * dout, dev)
    sp = _chk(dS, "dS", torch.float32, N * heads, dev)
    f32 = dict(dtype=torch.float32, device=dev)
    dX = dW = da = None
    ptr = lambda t: t.data_ptr() if t is not None else 0
    if part != "dw":
        dX = torch.empty(N, din, dtype=X.dtype, device=dev)
    work = None
    if part != "dx":
        dW = torch.empty(2, din, dout, **f32)
        da = torch.empty(2, dout, **f32)
        work = torch.empty(_lib.TBWD_BLOCKS, din * dout + dout, **f32)
    _lib.check(lib.ngacf_transform_bwd(xp, _DT[X.dtype], wp, hp, _DT[H.dtype], gp, sp, ptr(dX), ptr(dW),
                                       ptr(da), ptr(work), U, I, din, dout, heads,
                                       float(feat_drop_p), int(seed), _sd(seed_dev, dev), _stream()), "ngacf_transform_bwd")
    return dX, dW, da


def transform_dx_node(dH, W, U, I, feat_drop_p, seed, seed_dev, prev):
    """fp32 storage, a 64 -> 64 stage: dX = mask * (dH @ W^T) and, in the same launch, step 1 of the previous stage's backward on it
    (ngacf_transform_dx_node).  prev = (out, agg, R, S, elu, alpha) of that stage's forward (8-column heads).  Returns
    dX, Gn, ST [N, 8, 2], dS [N, 8] (P1): the previous stage's prop_bwd takes node=(Gn, ST, dS)."""
    dev, N = dH.device, U + I
    p_out, p_agg, p_R, p_S, p_elu, p_alpha = prev
    f32 = dict(dtype=torch.float32, device=dev)
    dX, Gn = torch.empty(N, 64, **f32), torch.empty(N, 64, **f32)
    ST, dS = torch.empty(N, 8, 2, **f32), torch.empty(N, 8, **f32)
    pv = _lib.DxPrev(_chk(p_out, "prev out", torch.float32, N * 64, dev), _chk(p_agg, "prev agg", torch.float32, 2 * N * 64, dev),
                     _chk(p_R, "prev R", torch.float32, 2 * N * 8, dev), _chk(p_S, "prev S", torch.float32, N * 8, dev),
                     Gn.data_ptr(), ST.data_ptr(), dS.data_ptr(), float(p_alpha), FLAG_ELU if p_elu else 0)
    _lib.check(lib.ngacf_transform_dx_node(_chk(dH, "dH", torch.float32, N * 64, dev), _chk(W, "W", torch.float32, 2 * 64 * 64, dev),
                                           dX.data_ptr(), U, I, float(feat_drop_p), int(seed), _sd(seed_dev, dev), ctypes.byref(pv), _stream()),
               "ngacf_transform_dx_node")
    return dX, Gn, ST, dS


DW_BATCH_BLOCKS = int(__import__("os").environ.get("NGACF_DW_BATCH_BLOCKS", "192"))     # workgroups per job of the batched dW launch


def transform_dw_batch(jobs, U, I, seed_dev=None):
    """The weight halves of transform_bwd for several 64 -> 64 stages (bf16 or fp32 storage) in one launch + one reduction
    (ngacf_transform_dw_batch).  jobs: [(X, H, dH, dS, heads, feat_drop_p, seed, dW, da)] with dW float [2, 64, 64] and
    da float [2, 64] given (they may have been handed out before this call fills them)."""
    n = len(jobs)
    if n == 0:
        return
    if n > _lib.DW_BATCH_MAX:
        raise ValueError("transform_dw_batch: at most %d jobs" % _lib.DW_BATCH_MAX)
    dev, N = jobs[0][0].device, U + I
    arr = (_lib.DwJob * n)()
    for k, (X, H, dH, dS, heads, p, seed, dW, da) in enumerate(jobs):
        if X.dtype not in _DT or H.dtype not in _DT or H.dtype != jobs[0][1].dtype or (H.dtype == torch.float32 and X.dtype != torch.float32):
            raise ValueError("transform_dw_batch: one storage type per call; fp32 storage takes fp32 X")
        arr[k] = _lib.DwJob(_chk(X, "X", None, N * 64, dev), _chk(H, "H", None, N * 64, dev), _chk(dH, "dH", H.dtype, N * 64, dev),
                            _chk(dS, "dS", torch.float32, N * heads, dev), _chk(dW, "dW", torch.float32, 2 * 64 * 64, dev),
                            _chk(da, "davec", torch.float32, 2 * 64, dev), _DT[X.dtype], _DT[H.dtype], 64, 64, int(heads), float(p),
                            int(seed) & 0xFFFFFFFFFFFFFFFF)
    work = torch.empty(n, DW_BATCH_BLOCKS, 64 * 64 + 64, dtype=torch.float32, device=dev)
    _lib.check(lib.ngacf_transform_dw_batch(arr, n, work.data_ptr(), DW_BATCH_BLOCKS, U, I, _sd(seed_dev, dev), _stream()),
               "ngacf_transform_dw_batch")


# ---------------------------------------------------------------------------------------------------------
# index safety of the data-dependent gathers.  The VALUES of userIdx / itemIdx / users live on the device, so they
# cannot be range-checked here without a synchronisation per call; the kernels check them instead: an out-of-range
# id is never turned into an address (score = NaN, no gradient written, top-K row = -1) and is counted in a
# per-device int32 counter.  raise_if_bad_indices() turns the count into the IndexError the reference's advanced
# indexing raises (SPUIGACF.py:49-50) -- like a CUDA device-side assert it surfaces at the next host check, which the
# train / eval loops do where they synchronise anyway (loss.item(), metric transfer).  NGACF_SYNC_INDEX_CHECK=1 checks
# after every scoring call (one sync each; debugging).
# ---------------------------------------------------------------------------------------------------------
_BAD = {}
SYNC_INDEX_CHECK = __import__("os").environ.get("NGACF_SYNC_INDEX_CHECK") == "1"


def _bad_counter(dev):
    c = _BAD.get(dev)
    if c is None:
        c = _BAD[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return c


def _same_device(key, dev):
    """Does the counter's device `key` (always indexed: cuda:0) match `dev`?  An index-less device ("cuda", what
    Interactions(device="cuda") stores) means the current device."""
    if dev is None:
        return True
    dev = torch.device(dev)
    if key.type != dev.type:
        return False
    want = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else None)
    return key.index == want


def bad_index_count(dev=None):
    """Out-of-range indices seen by the scoring / ranking kernels since the last reset (synchronises)."""
    return sum(int(c.item()) for d, c in _BAD.items() if _same_device(d, dev))


def raise_if_bad_indices(dev=None):
    n = bad_index_count(dev)
    if n:
        for d, c in _BAD.items():
            if _same_device(d, dev):
                c.zero_()
        raise IndexError("%d user/item index value(s) out of range reached the scoring kernels (their scores are NaN, "
                         "their gradients were dropped)" % n)


def _idx(t, name, B, dev):
    return _chk(t, name, torch.int64, B, dev, align=8)       # read element-wise: a row of a packed [3, B] batch is fine


def score_fwd(F, u, i, U):
    dev, d, B = F.device, F.shape[1], u.numel()
    if d not in SUPPORTED_D or F.dtype not in _DT:
        raise ValueError("unsupported F: d=%d dtype=%s" % (d, F.dtype))
    I = F.shape[0] - U
    if not 0 < U < F.shape[0]:
        raise ValueError("F has %d rows, U=%d" % (F.shape[0], U))
    fp = _chk(F, "F", None, F.shape[0] * d, dev)
    score = torch.empty(B, dtype=torch.float32, device=dev)
    if B:
        _lib.check(lib.ngacf_score_fwd(fp, _DT[F.dtype], _idx(u, "userIdx", B, dev), _idx(i, "itemIdx", B, dev),
                                       score.data_ptr(), _bad_counter(dev).data_ptr(), B, U, I, d, _stream()), "ngacf_score_fwd")
        if SYNC_INDEX_CHECK:
            raise_if_bad_indices(dev)
    return score


def score_bwd(F, u, i, gscore, U, into=None):
    """gF float [N, d] of sum(F[u] * F[U + i], 1).  into: ADD to that table instead of a fresh zero one (BatchRows.g32)."""
    dev, d, B = F.device, F.shape[1], u.numel()
    I = F.shape[0] - U
    if not 0 < U < F.shape[0]:
        raise ValueError("F has %d rows, U=%d" % (F.shape[0], U))
    fp = _chk(F, "F", None, F.shape[0] * d, dev)
    gF = torch.zeros(F.shape[0], d, dtype=torch.float32, device=dev) if into is None else into
    if into is not None:
        _chk(into, "into", torch.float32, F.shape[0] * d, dev)
    if B:
        gp = _chk(gscore, "gscore", torch.float32, B, dev)
        _lib.check(lib.ngacf_score_bwd(fp, _DT[F.dtype], _idx(u, "userIdx", B, dev), _idx(i, "itemIdx", B, dev),
                                       gp, gF.data_ptr(), _bad_counter(dev).data_ptr(), B, U, I, d, _stream()), "ngacf_score_bwd")
    return gF


ROWS_SCATTER_MAX_N = 16384


def shard_rows(Tu, Ti, u, i, perm_u, perm_i, U, I, u_lo, i_lo):
    """(rows float [2B, d], pos int32 [2B]) of ngacf_shard_rows: the batch's rows this rank owns (zeros elsewhere) and their local
    row numbers (-1 elsewhere).  Tu [cu, d], Ti [ci, d]: the rank's local tables."""
    dev, d, B = Tu.device, Tu.shape[1], u.numel()
    cu, ci = Tu.shape[0], Ti.shape[0]
    if d not in SUPPORTED_D or Tu.dtype not in _DT or Ti.dtype != Tu.dtype or Ti.shape[1] != d:
        raise ValueError("shard_rows: tables [cu, d] / [ci, d] of one supported type, d in %s" % (SUPPORTED_D,))
    rows = torch.empty(2 * B, d, dtype=torch.float32, device=dev)
    pos = torch.empty(2 * B, dtype=torch.int32, device=dev)
    if B:
        _lib.check(lib.ngacf_shard_rows(_chk(Tu, "Tu", None, cu * d, dev) if cu else 0, _chk(Ti, "Ti", None, ci * d, dev) if ci else 0, _DT[Tu.dtype],
                                        _idx(u, "userIdx", B, dev), _idx(i, "itemIdx", B, dev),
                                        _chk(perm_u, "perm_u", torch.int64, U, dev, align=8) if perm_u is not None else 0,
                                        _chk(perm_i, "perm_i", torch.int64, I, dev, align=8) if perm_i is not None else 0,
                                        rows.data_ptr(), pos.data_ptr(), _bad_counter(dev).data_ptr(), B, U, I, int(u_lo), cu, int(i_lo), ci, d,
                                        _stream()), "ngacf_shard_rows")
    return rows, pos


def rows_scatter(pos, g, cu, ci, dtype):
    """(du [cu, d], di [ci, d]) in `dtype`: row pos[b] = the sum of the g[b'] naming it (ngacf_rows_scatter), zeros elsewhere."""
    dev, (n, d) = g.device, g.shape
    if n > ROWS_SCATTER_MAX_N or d not in SUPPORTED_D or dtype not in _DT:
        raise ValueError("rows_scatter: at most %d rows of d in %s" % (ROWS_SCATTER_MAX_N, SUPPORTED_D))
    flat = torch.zeros(cu + ci, d, dtype=dtype, device=dev)           # (one table, one fill; the halves are views of it)
    du, di = flat[:cu], flat[cu:]
    if n:
        _lib.check(lib.ngacf_rows_scatter(_chk(pos, "pos", torch.int32, n, dev, align=4), _chk(g, "g", torch.float32, n * d, dev),
                                          du.data_ptr(), di.data_ptr(), _DT[dtype], n, cu, d, _stream()), "ngacf_rows_scatter")
    return du, di


def rows_scatter_scored(pos, rows, gscore, cu, ci, dtype):
    """rows_scatter of the gradient rows of a scored batch, formed on the fly from the exchanged rows [2B, d] (fp32) and the scores'
    gradient [B] (ngacf_rows_scatter_scored): what score_bwd over the row table + rows_scatter leave, in one launch."""
    dev, (n, d) = rows.device, rows.shape
    if n > ROWS_SCATTER_MAX_N or n % 2 or d not in SUPPORTED_D or dtype not in _DT:
        raise ValueError("rows_scatter_scored: 2B <= %d rows of d in %s" % (ROWS_SCATTER_MAX_N, SUPPORTED_D))
    flat = torch.zeros(cu + ci, d, dtype=dtype, device=dev)
    du, di = flat[:cu], flat[cu:]
    if n:
        _lib.check(lib.ngacf_rows_scatter_scored(_chk(pos, "pos", torch.int32, n, dev, align=4), _chk(rows, "rows", torch.float32, n * d, dev),
                                                 _chk(gscore, "gscore", torch.float32, n // 2, dev, align=4), du.data_ptr(), di.data_ptr(),
                                                 _DT[dtype], n, cu, d, _stream()), "ngacf_rows_scatter_scored")
    return du, di


def bpr_fwd(pos, neg):
    dev, B = pos.device, pos.numel()
    loss = torch.empty((), dtype=torch.float32, device=dev)
    _lib.check(lib.ngacf_bpr_fwd(_chk(pos, "pos", torch.float32, B, dev), _chk(neg, "neg", torch.float32, B, dev),
                                 loss.data_ptr(), B, _stream()), "ngacf_bpr_fwd")
    return loss


def bpr_fwd_bwd(pos, neg):
    """loss, d loss / d pos, d loss / d neg (for a unit gradient of the loss) in one launch"""
    dev, B = pos.device, pos.numel()
    loss = torch.empty((), dtype=torch.float32, device=dev)
    gpos, gneg = torch.empty_like(pos), torch.empty_like(neg)
    _lib.check(lib.ngacf_bpr_fwd_bwd(_chk(pos, "pos", torch.float32, B, dev), _chk(neg, "neg", torch.float32, B, dev), loss.data_ptr(),
                                     gpos.data_ptr(), gneg.data_ptr(), B, _stream()), "ngacf_bpr_fwd_bwd")
    return loss, gpos, gneg


def bpr_bwd(pos, neg, gloss):
    dev, B = pos.device, pos.numel()
    gpos, gneg = torch.empty_like(pos), torch.empty_like(neg)
    gl = gloss.reshape(1).contiguous().float()
    _lib.check(lib.ngacf_bpr_bwd(_chk(pos, "pos", torch.float32, B, dev), _chk(neg, "neg", torch.float32, B, dev),
                                 gl.data_ptr(), gpos.data_ptr(), gneg.data_ptr(), B, _stream()), "ngacf_bpr_bwd")
    return gpos, gneg


def spmm_fwd(plan, val, X, self_loop=True):
    dev, N, d = X.device, plan.N, X.shape[1]
    if d not in SUPPORTED_D + (256,) or X.dtype not in _DT:          # (256: the dense-attention models' default width)
        raise ValueError("unsupported X: d=%d dtype=%s" % (d, X.dtype))
    xp = _chk(X, "X", None, N * d, dev)
    vp = _chk(val, "val", torch.float32, plan.n_edges_directed, dev)
    segs_t, split_t, nparts = plan.all_rows_segments()
    nseg, nsplit = int(segs_t.shape[0]), int(split_t.shape[0])
    segs = _chk(segs_t, "segs", torch.int32, nseg * 4, dev) if nseg else 0
    split = _chk(split_t, "split", torch.int32, nsplit * 3, dev) if nsplit else 0
    col = _chk(plan.col, "plan.col", torch.int32, plan.n_edges_directed, dev)
    out = torch.empty_like(X)
    pa = torch.empty(max(nparts, 1), d, dtype=torch.float32, device=dev)
    _lib.check(lib.ngacf_spmm_fwd(segs, nseg, split, nsplit, col, vp, xp, _DT[X.dtype], out.data_ptr(),
                                  pa.data_ptr(), N, d, int(bool(self_loop)), _stream()), "ngacf_spmm_fwd")
    return out


def pack_mask_bits(mask):
    """[Nq, Nk] tensor (anything comparable with 0; modules.py:36 tests `mask == 0`) -> int32 [Nq, ceil(Nk/32)], bit (k & 31)
    of word k // 32 set where mask[q, k] != 0: the layout ngacf_mha_fwd / _bwd read (one word per query and key tile)."""
    Nq, Nk = mask.shape
    W = (Nk + 31) // 32
    m = torch.zeros(Nq, W * 32, dtype=torch.int64, device=mask.device)
    m[:, :Nk] = (mask != 0)
    w = (m.view(Nq, W, 32) << torch.arange(32, device=mask.device)).sum(2)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


def _mask_bits(mask, Nq, Nk, dev):
    """(pointer, tensor to keep alive until the launch is queued): a mask packed here is a temporary -- dropping it before
    the launch would hand its block to the next allocation (the output table) while the pointer is still in use."""
    if mask is None:
        return 0, None
    if mask.dtype != torch.int32 or tuple(mask.shape) != (Nq, (Nk + 31) // 32):
        mask = pack_mask_bits(mask)
    return _chk(mask, "mask bits", torch.int32, Nq * ((Nk + 31) // 32), dev, align=4), mask


def mha_fwd(Q, K, V, heads, mask=None, drop_p=0.0, seed=0, seed_dev=None):
    """O [Nq,dim], LSE [Nq,heads,2] = ngacf_mha_fwd(Q [Nq,dim], K [Nk,dim], V [Nk,dim], mask).  mask: None, the packed
    int32 words of pack_mask_bits, or any [Nq, Nk] tensor (packed here, per call: cache the packed form for repeated use)."""
    dev, (Nq, dim), Nk = Q.device, Q.shape, K.shape[0]
    if dim % heads or dim // heads not in (8, 16, 32):
        raise ValueError("unsupported attention shape: dim=%d heads=%d (d_k must be 8, 16 or 32)" % (dim, heads))
    qp = _chk(Q, "Q", torch.float32, Nq * dim, dev)
    kp = _chk(K, "K", torch.float32, Nk * dim, dev)
    vp = _chk(V, "V", torch.float32, Nk * dim, dev)
    mp, _keep = _mask_bits(mask, Nq, Nk, dev)
    O = torch.empty(Nq, dim, dtype=torch.float32, device=dev)
    LSE = torch.empty(Nq, heads, 2, dtype=torch.float32, device=dev)
    if Nq and Nk:
        _lib.check(lib.ngacf_mha_fwd(qp, kp, vp, mp, O.data_ptr(), LSE.data_ptr(), Nq, Nk, dim, heads,
                          float(drop_p), int(seed), _sd(seed_dev, dev), _stream()), "ngacf_mha_fwd")
    return O, LSE


def mha_bwd(Q, K, V, O, LSE, dO, heads, mask=None, drop_p=0.0, seed=0, seed_dev=None):
    """dQ, dK, dV of ngacf_mha_fwd."""
    dev, (Nq, dim), Nk = Q.device, Q.shape, K.shape[0]
    qp = _chk(Q, "Q", torch.float32, Nq * dim, dev)
    kp = _chk(K, "K", torch.float32, Nk * dim, dev)
    vp = _chk(V, "V", torch.float32, Nk * dim, dev)
    op = _chk(O, "O", torch.float32, Nq * dim, dev)
    lp = _chk(LSE, "LSE", torch.float32, Nq * heads * 2, dev)
    gp = _chk(dO, "dO", torch.float32, Nq * dim, dev)
    mp, _keep = _mask_bits(mask, Nq, Nk, dev)
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    delta = torch.empty(Nq, heads, dtype=torch.float32, device=dev)
    if Nq and Nk:
        _lib.check(lib.ngacf_mha_bwd(qp, kp, vp, mp, op, lp, gp, dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(), delta.data_ptr(),
                                     Nq, Nk, dim, heads, float(drop_p), int(seed), _sd(seed_dev, dev),
                                     _stream()), "ngacf_mha_bwd")
    return dQ, dK, dV


def dropout_bits(kind, seed, a, b, c=None):
    dev, n = a.device, a.numel()
    bits = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(lib.ngacf_dropout_bits(kind, int(seed), _chk(a, "a", torch.int32, n, dev), _chk(b, "b", torch.int32, n, dev),
                                      _chk(c, "c", torch.int32, n, dev) if c is not None else 0,
                                      bits.data_ptr(), n, _stream()), "ngacf_dropout_bits")
    return bits


# ---------------------------------------------------------------------------
# row-window forms used by the node-partitioned (multi-GPU) path, ngacf_amd/dist.py
# ---------------------------------------------------------------------------
def pack_panels(Ws, kinds=None):
    """bf16 matrix-core fragment panels of the stage weights Ws = [W [2, 64, dout], ...] for prop_fwd(next=...) (kind 0) or
    prop_bwd(dx=...) (kind 1), by a launch of their own (ngacf_transform_fwd_pack with no rows: the node-partitioned path, whose
    transforms run over row windows; the single-GPU stack packs inside its first transform's launch instead)."""
    dev, n = Ws[0].device, len(Ws)
    kinds = [0] * n if kinds is None else [int(k) for k in kinds]
    if n > _lib.MAX_PACK:
        raise ValueError("pack_panels: at most %d panels" % _lib.MAX_PACK)
    for Wk in Ws:
        if Wk.dim() != 3 or Wk.shape[0] != 2 or Wk.shape[1] != 64 or Wk.shape[2] not in (64, 128):
            raise ValueError("pack: W [2, 64, 64 | 128], got %s" % (tuple(Wk.shape),))
    panels = [torch.empty(2 * Wk.shape[1] * Wk.shape[2] * 2, dtype=torch.uint8, device=dev) for Wk in Ws]
    P, Iarr = ctypes.c_void_p * n, ctypes.c_int * n
    _lib.check(lib.ngacf_transform_fwd_pack(0, F32, 0, 0, 0, BF16, 0, 0, 0, 64, 64, 8, 0.0, 0, 0, n,
                                            P(*[_chk(Wk, "pack W", torch.float32, Wk.numel(), dev) for Wk in Ws]), P(*[t.data_ptr() for t in panels]),
                                            Iarr(*[int(Wk.shape[1]) for Wk in Ws]), Iarr(*[int(Wk.shape[2]) for Wk in Ws]), Iarr(*kinds), _stream()),
               "ngacf_transform_fwd_pack")
    return panels


def transform_fwd_rows(X, W, avec, H, S, U, I, heads, win, x_local, feat_drop_p=0.0, seed=0, seed_dev=None):
    """Transform the rows of `win` = (u_lo, u_n, i_lo, i_n) into the global-row tables H [N,dout], S [N,heads].
    X is [u_n+i_n, din] when x_local else [N, din]."""
    dev, N = H.device, U + I
    din, dout = W.shape[1], W.shape[2]
    _heads_ok(dout, heads)
    u_lo, u_n, i_lo, i_n = win
    xp = _chk(X, "X", None, ((u_n + i_n) if x_local else N) * din, dev)
    wp = _chk(W, "W", torch.float32, 2 * din * dout, dev)
    ap = _chk(avec, "avec", torch.float32, 2 * dout, dev)
    hp = _chk(H, "H", None, N * dout, dev)
    sp = _chk(S, "S", torch.float32, N * heads, dev)
    _lib.check(lib.ngacf_transform_fwd_rows(xp, _DT[X.dtype], wp, ap, hp, _DT[H.dtype], sp, U, I, din, dout, heads,
                                            u_lo, u_n, i_lo, i_n, int(bool(x_local)), float(feat_drop_p), int(seed), _sd(seed_dev, dev),
                                            _stream()), "ngacf_transform_fwd_rows")


def transform_bwd_rows(X, W, H, dH, dS, U, I, heads, win, x_local, feat_drop_p=0.0, seed=0, seed_dev=None):
    """dX ([u_n+i_n, din] when x_local), and the window's partial dW [2,din,dout], davec [2,dout]."""
    dev, N = H.device, U + I
    din, dout = W.shape[1], W.shape[2]
    _heads_ok(dout, heads)
    u_lo, u_n, i_lo, i_n = win
    nx = (u_n + i_n) if x_local else N
    xp = _chk(X, "X", None, nx * din, dev)
    wp = _chk(W, "W", torch.float32, 2 * din * dout, dev)
    hp = _chk(H, "H", None, N * dout, dev)
    gp = _chk(dH, "dH", H.dtype, N * dout, dev)
    sp = _chk(dS, "dS", torch.float32, N * heads, dev)
    f32 = dict(dtype=torch.float32, device=dev)
    dX = torch.empty(nx, din, dtype=X.dtype, device=dev)
    dW = torch.empty(2, din, dout, **f32)
    da = torch.empty(2, dout, **f32)
    work = torch.empty(_lib.TBWD_BLOCKS, din * dout + dout, **f32)
    _lib.check(lib.ngacf_transform_bwd_rows(xp, _DT[X.dtype], wp, hp, _DT[H.dtype], gp, sp, dX.data_ptr(), dW.data_ptr(),
                                            da.data_ptr(), work.data_ptr(), U, I, din, dout, heads,
                                            u_lo, u_n, i_lo, i_n, int(bool(x_local)), float(feat_drop_p), int(seed), _sd(seed_dev, dev),
                                            _stream()), "ngacf_transform_bwd_rows")
    return dX, dW, da


def prop_bwd_node(out, agg, R, gout, S_src, G, Gn, ST, dS, U, heads, win, gout_local, elu=True, alpha=0.2):
    """Step 1 of the propagation backward over the rows of `win`; writes G, Gn, ST [N, heads, 2] (pairs (S, t)) and P1 into dS,
    at global rows.  agg [2, N, d], R [2, N, heads]: the forward's saved planes."""
    dev, N, d = out.device, out.shape[0], out.shape[1]
    _heads_ok(d, heads)
    u_lo, u_n, i_lo, i_n = win
    op = _chk(out, "out", None, N * d, dev)
    ap = _chk(agg, "agg", out.dtype, 2 * N * d, dev)
    rp = _chk(R, "R", torch.float32, 2 * N * heads, dev)
    go = _chk(gout, "gout", out.dtype, ((u_n + i_n) if gout_local else N) * d, dev)
    ss = _chk(S_src, "S_src", torch.float32, N * heads, dev)
    gp = _chk(G, "G", torch.float32, N * d, dev)
    gn = _chk(Gn, "Gn", out.dtype, N * d, dev)
    st = _chk(ST, "ST", torch.float32, N * heads * 2, dev)
    ds = _chk(dS, "dS", torch.float32, N * heads, dev)
    _lib.check(lib.ngacf_prop_bwd_node(op, _DT[out.dtype], ap, rp, go, ss, gp, gn, st, ds, U, N, d, heads, float(alpha),
                                       FLAG_ELU if elu else 0, u_lo, u_n, i_lo, i_n, int(bool(gout_local)), _stream()),
               "ngacf_prop_bwd_node")


def prop_bwd_edges(plan, H, S_dst, avec, G, Gn, ST, heads, alpha=0.2, residual=True, score_from_row=True,
                   drop_p=0.0, seed=0, seed_dev=None, half=None, dH=None, dS=None):
    """Step 2 alone (G for the plan's rows, Gn and ST for every source row, and dS holding step 1's P1 for the plan's rows).
    Returns dH, dS (rows of the plan valid; dS finished in place).
    half = "user" / "item": only those dst rows, written into the dH given (or a fresh zeroed table)."""
    dev, N, d = H.device, plan.N, H.shape[1]
    _heads_ok(d, heads)
    if dS is None:
        raise ValueError("prop_bwd_edges: dS must hold step 1's P1 (prop_bwd_node) for the rows of the plan")
    hp = _chk(H, "H", None, N * d, dev)
    sd = _chk(S_dst, "S_dst", torch.float32, N * heads, dev)
    ap = _chk(avec, "avec", torch.float32, 2 * d, dev)
    gp = _chk(G, "G", torch.float32, N * d, dev)
    gn = _chk(Gn, "Gn", H.dtype, N * d, dev)
    st = _chk(ST, "ST", torch.float32, N * heads * 2, dev)
    hub = _hub(plan, dev, half)
    pp = _plan_ptrs(plan, dev, half, inkernel=hub[1] != 0)
    f32 = dict(dtype=torch.float32, device=dev)
    dH = torch.zeros(N, d, dtype=H.dtype, device=dev) if dH is None else dH
    _chk(dH, "dH", H.dtype, N * d, dev)
    _chk(dS, "dS", torch.float32, N * heads, dev)
    pa = torch.empty(max(plan.nparts, 1), d, **f32)
    pr = torch.empty(max(plan.nparts, 1), heads, **f32)
    flags = (FLAG_RESIDUAL if residual else 0) | FLAG_SCORE_ROW | _lib.FLAG_SKIP_NODE
    _lib.check(lib.ngacf_prop_bwd(*pp, hp, _DT[H.dtype], sd, sd, ap, hp, gp, sd, gp, gp, gn, st, dH.data_ptr(),
                                  dS.data_ptr(), pa.data_ptr(), pr.data_ptr(), *hub, 0, 0, 0, plan.U, N, d, heads, float(alpha),
                                  float(drop_p), int(seed), _sd(seed_dev, dev), flags, _stream()), "ngacf_prop_bwd")
    return dH, dS


def eval_topk(F, users, U, I, pool_mask, train_ptr, train_items, K=20):
    """top_items [B,K] int32 (-1 = no candidate left), top_scores [B,K] fp32: the K best candidate items of every
    user, candidates = pool minus the user's train positives, ties to the lower item id (ngacf_eval_topk)."""
    dev, d, B = F.device, F.shape[1], users.numel()
    if d not in SUPPORTED_D or F.dtype not in _DT:
        raise ValueError("unsupported F: d=%d dtype=%s" % (d, F.dtype))
    if not 0 < K <= 32:
        raise ValueError("K must be in 1..32")
    fp = _chk(F, "F", None, (U + I) * d, dev)
    top_i = torch.empty(B, K, dtype=torch.int32, device=dev)
    top_s = torch.empty(B, K, dtype=torch.float32, device=dev)
    if B:
        _lib.check(lib.ngacf_eval_topk(fp, _DT[F.dtype], _chk(users, "users", torch.int64, B, dev), B, U, I, d,
                                       _chk(pool_mask, "pool_mask", torch.uint8, I, dev),
                                       _chk(train_ptr, "train_ptr", torch.int64, U + 1, dev),
                                       _chk(train_items, "train_items", torch.int32, train_items.numel(), dev),
                                       K, top_i.data_ptr(), top_s.data_ptr(), _bad_counter(dev).data_ptr(), _stream()),
                   "ngacf_eval_topk")
    return top_i, top_s


def sample_pairs(train_u, offset, B, train_ptr, train_items32, pool_items32, seed, out=None, unresolved=None,
                 offset_dev=None, seed_dev=None):
    """(users, pos, neg) of the batch rows train_u[offset : offset+B] (ngacf_sample_pairs).  `out` = an int64 [3,B]
    buffer to fill (GraphedBPRStep's static input); `offset_dev`/`seed_dev` = optional int64[1] device addends."""
    dev = train_u.device
    if out is None:
        out = torch.empty(3, B, dtype=torch.int64, device=dev)
    if unresolved is None:
        unresolved = torch.zeros(1, dtype=torch.int32, device=dev)
    if B:
        U = train_ptr.numel() - 1
        n_rows = train_u.numel()
        if offset < 0 or (offset_dev is None and offset + B > n_rows):
            raise ValueError("batch rows [%d, %d) run past the %d interaction rows" % (offset, offset + B, n_rows))
        _chk(out, "out", torch.int64, 3 * B, dev)
        _lib.check(lib.ngacf_sample_pairs(_chk(train_u, "train_u", torch.int64, n_rows, dev), n_rows, int(offset), _sd(offset_dev, dev),
                                          _chk(train_ptr, "train_ptr", torch.int64, U + 1, dev), U,
                                          _chk(train_items32, "train_items", torch.int32, train_items32.numel(), dev),
                                          _chk(pool_items32, "pool_items", torch.int32, pool_items32.numel(), dev),
                                          pool_items32.numel(), int(seed) & 0xFFFFFFFFFFFFFFFF, _sd(seed_dev, dev),
                                          out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                          _chk(unresolved, "unresolved", torch.int32, 1, dev), B, _stream()), "ngacf_sample_pairs")
    return out, unresolved


def adam_step(params, grads, exp_avg, exp_avg_sq, steps, lr, beta1, beta2, eps, weight_decay, grads2=None):
    """ngacf_adam_step over up to _lib.ADAM_MAX_TENSORS fp32 tensors (one launch); steps[t]: 0-dim device float, already
    incremented.  grads2: optional list (entries may be None) of second gradients, g = grads[t] + grads2[t]."""
    n = len(params)
    if not (n == len(grads) == len(exp_avg) == len(exp_avg_sq) == len(steps)) or n > _lib.ADAM_MAX_TENSORS:
        raise ValueError("adam_step: %d tensors (lists of equal length, at most %d)" % (n, _lib.ADAM_MAX_TENSORS))
    if n == 0:
        return
    dev = params[0].device
    P = ctypes.c_void_p * n
    numel = [p.numel() for p in params]

    def table(ts, name, optional=False):
        return P(*[(None if optional and t is None else _chk(t, name, torch.float32, numel[k], dev)) for k, t in enumerate(ts)])
    tp, tg, tm, tv = table(params, "param"), table(grads, "grad"), table(exp_avg, "exp_avg"), table(exp_avg_sq, "exp_avg_sq")
    if grads2 is not None and len(grads2) != n:
        raise ValueError("adam_step: %d second gradients for %d tensors" % (len(grads2), n))
    t2 = table(grads2, "grad2", True) if grads2 is not None and any(g is not None for g in grads2) else None
    ts = P(*[_chk(s_, "step", torch.float32, 1, dev, align=4) for s_ in steps])
    nn = (ctypes.c_int64 * n)(*numel)
    _lib.check(lib.ngacf_adam_step(tp, tg, t2, tm, tv, ts, nn, n, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                   _stream()), "ngacf_adam_step")


def tick(int_counters=(), int_strides=(), float_counters=()):
    """One launch for the scalar bookkeeping of a train step: int64 device counters += strides, float device counters += 1."""
    ni, nf = len(int_counters), len(float_counters)
    if ni != len(int_strides) or ni > _lib.TICK_MAX or nf > _lib.TICK_MAX:
        raise ValueError("tick: %d int counters / %d strides / %d float counters (at most %d each)" % (ni, len(int_strides), nf, _lib.TICK_MAX))
    if ni + nf == 0:
        return
    dev = (int_counters[0] if ni else float_counters[0]).device
    P = ctypes.c_void_p
    ic = (P * max(ni, 1))(*[_chk(t, "int counter", torch.int64, 1, dev, align=8) for t in int_counters])
    st = (ctypes.c_int64 * max(ni, 1))(*[int(x) for x in int_strides])
    fc = (P * max(nf, 1))(*[_chk(t, "float counter", torch.float32, 1, dev, align=4) for t in float_counters])
    _lib.check(lib.ngacf_tick(ic, st, ni, fc, nf, _stream()), "ngacf_tick")
