"""Graph plan: the reference's COO `indices` ([2,E] int64, row 0 user id, row 1 item id,
as returned by data/loadGowalla.py:253 `.coalesce().indices()` and kept by
run_Gowalla.py:94) turned once into the traversal layout of include/ngacf.h:

  col    int32[2E]   CSR of the user rows followed by the CSC of the item rows, sources
                     as global node ids (items offset by U)
  hrow   int32[H]    hub rows (long_max < degree <= hub_cap): one workgroup per row, folded through LDS
  segs   int32[S,4]  {dst row, beg, end, part}: rows of short_max < degree <= long_max are one segment (one wave);
                     rows beyond hub_cap are cut into seg_len slices with their own partial-sum slot; longest first
  split  int32[P,3]  {dst row, first slot, slot count} of the cut rows

Everything is a torch tensor on the graph's device (PyTorch owns the memory).  Built
with torch ops so a 1e9-edge graph can be planned on the GPU; the builder checks the
invariants the kernels rely on, because a bad index is a GPU fault, not an exception.
"""
import torch


_TICKET_SETS = 80        # arrival-counter sets per plan: one per stream that launches it (torch hands out streams from pools of
                         # 32 per priority: every handle a process can meet gets a row of the pool, none is ever allocated later)


class GraphPlan:
    def __init__(self, U, I, E, rowptr, col, segs, split, nparts, seg_len, row_lo=0, row_hi=None,
                 srow=None, short_max=0, row_ranges=None, hrow=None, nhub_user=0):
        self.U, self.I, self.N, self.E = U, I, U + I, E
        self.rowptr, self.col, self.segs, self.split = rowptr, col, segs, split
        self.nseg, self.nsplit, self.nparts = int(segs.shape[0]), int(split.shape[0]), int(nparts)
        self.seg_len = seg_len
        self.row_lo, self.row_hi = row_lo, (U + I if row_hi is None else row_hi)
        # dst rows this plan covers: a list of [lo, hi) ranges (one range = everything; two = a rank's
        # user block and item block in node-partitioned runs)
        self.row_ranges = list(row_ranges) if row_ranges is not None else [(self.row_lo, self.row_hi)]
        self.n_rows = sum(hi - lo for lo, hi in self.row_ranges)
        self.device = col.device
        self.n_edges_directed = int(col.shape[0])
        # short rows (degree <= short_max), sorted by degree descending: one lane group per row
        self.srow = srow if srow is not None else torch.empty(0, dtype=torch.int32, device=col.device)
        self.nshort, self.short_max = int(self.srow.shape[0]), short_max
        # what the kernels read: one 16-byte record {row, first edge, degree, 0} per short row (padding: row -1, no edges) -- a
        # wave learns its rows' edge ranges with ONE load instead of the chain srow -> rowptr[row], rowptr[row + 1]
        rr = self.srow.long().clamp(min=0)
        live = (self.srow >= 0)
        beg = torch.where(live, rowptr[rr], torch.zeros_like(rr))
        deg = torch.where(live, rowptr[rr + 1] - rowptr[rr], torch.zeros_like(rr))
        self.srec = torch.stack([self.srow.long(), beg, deg, torch.zeros_like(rr)], 1).to(torch.int32).contiguous()
        self.nseg_user, self.nshort_user, self.nsplit_user = self.nseg, self.nshort, self.nsplit   # build_plan sets the user/item split
        # hub rows (degree > seg_len, up to hub_cap): one WORKGROUP per row, its four waves stride over the row's edges and
        # fold through LDS -- no partial sums in memory, no second kernel.  Users first, each half by degree descending.
        self.hrow = hrow if hrow is not None else torch.empty(0, dtype=torch.int32, device=col.device)
        self.nhub, self.nhub_user = int(self.hrow.shape[0]), int(nhub_user)
        self.rowptr32 = rowptr.to(torch.int32).contiguous()
        # hub rows: slot -> hub row, and one arrival counter per hub row (zero between launches)
        if self.nsplit:
            self.part_split = torch.repeat_interleave(torch.arange(self.nsplit, device=col.device), split[:, 2].long()).to(torch.int32)
        else:
            self.part_split = torch.empty(0, dtype=torch.int32, device=col.device)
        # (a pool allocated HERE: a buffer created during a hipGraph capture would put its zero fill into the graph)
        self._ticket_pool = torch.zeros(_TICKET_SETS, max(self.nsplit, 1), dtype=torch.int32, device=col.device)
        self.tickets = self._ticket_pool[0, :self.nsplit]
        self._tickets = {}

    def tickets_for(self, stream_handle):
        """The arrival counters for launches on ONE stream (zero between launches: the last slice resets its row's counter).
        Launches on different streams may overlap (the two propagations of a train step), so each stream has its own."""
        t = self._tickets.get(stream_handle)
        if t is None:
            k = len(self._tickets)
            if k < _TICKET_SETS:
                t = self._ticket_pool[k, :self.nsplit]
            else:                                   # more streams than sets: a fresh one (its zero fill is a launch)
                with torch.cuda.device(self.device):
                    t = torch.zeros(max(self.nsplit, 1), dtype=torch.int32, device=self.device)[:self.nsplit]
            self._tickets[stream_handle] = t
        return t

    @property
    def degrees(self):
        return self.rowptr[1:] - self.rowptr[:-1]

    def all_rows_segments(self):
        """(segs, split, nparts) covering EVERY dst row of the range as wave-per-segment work
        items (used by the static-weight SpMM path)."""
        if getattr(self, "_all", None) is None:
            rows = torch.cat([torch.arange(lo, hi, device=self.device) for lo, hi in self.row_ranges])
            self._all = _segments(self.rowptr, rows, self.seg_len)
        return self._all

    def to(self, device):
        p = GraphPlan(self.U, self.I, self.E, self.rowptr.to(device), self.col.to(device),
                      self.segs.to(device), self.split.to(device), self.nparts, self.seg_len,
                      self.row_lo, self.row_hi, self.srow.to(device), self.short_max, self.row_ranges,
                      self.hrow.to(device), self.nhub_user)
        p.nseg_user, p.nshort_user, p.nsplit_user = self.nseg_user, self.nshort_user, self.nsplit_user
        return p


def _segments(rowptr, rows, seg_len):
    """rows: int64 tensor of the dst rows to cover (global ids); rowptr is indexed by global id."""
    dev = rowptr.device
    beg_r, end_r = rowptr[rows], rowptr[rows + 1]
    deg = end_r - beg_r
    npr = torch.clamp((deg + seg_len - 1) // seg_len, min=1)          # an empty row still needs its epilogue
    nseg = int(npr.sum())
    ridx = torch.repeat_interleave(torch.arange(rows.numel(), device=dev), npr)
    first = torch.cumsum(npr, 0) - npr
    k = torch.arange(nseg, device=dev) - first[ridx]
    beg = beg_r[ridx] + k * seg_len
    end = torch.minimum(beg + seg_len, end_r[ridx])
    cut = npr[ridx] > 1
    part = torch.where(cut, torch.cumsum(cut.long(), 0) - 1, torch.full_like(beg, -1))
    segs = torch.stack([rows[ridx], beg, end, part], 1)
    order = torch.sort(end - beg, descending=True, stable=True).indices   # longest first: short tail
    segs = segs[order].to(torch.int32).contiguous()
    cut_rows = npr > 1
    first_part = torch.cumsum(torch.where(cut_rows, npr, torch.zeros_like(npr)), 0) - npr
    split = torch.stack([rows[cut_rows], first_part[cut_rows], npr[cut_rows]], 1).to(torch.int32).contiguous()
    return segs, split, int(cut.sum())


def squeeze_mask(mask):
    """The reference's `mask.squeeze()` (SPUIGACF.py:42) for the [1,2,E] DataParallel form -- without collapsing a
    one-edge graph's [2,1] into [2]."""
    return mask[0] if mask.dim() == 3 and mask.shape[0] == 1 else mask


def build_plan(indices, U, I, seg_len=256, device=None, require_user_edges=True, row_range=None, short_max=64, hub_cap=0,
               long_max=None):
    """indices: [2,E] (or [1,2,E], run under DataParallel: SPUIGACF.py:42 squeezes it) integer
    tensor.  Duplicate pairs are merged like `.coalesce()`.  `row_range=(lo,hi)` -- or a list of such
    ranges -- keeps only those dst rows (node partition of ngacf_amd/dist.py); sources stay global.
    With a row range only the edges that END in the range are kept, sorted and stored (a rank of a P-way partition
    holds ~2E/P column indices, not 2E): the filter is one pass over the E pairs, no global sort."""
    indices = squeeze_mask(indices)
    if indices.dim() != 2 or indices.shape[0] != 2:
        raise ValueError("indices must be [2,E], got %s" % (tuple(indices.shape),))
    dev = torch.device(device) if device is not None else indices.device
    idx = indices.to(dev).long()
    N = U + I
    if 2 * idx.shape[1] >= 2 ** 31:
        raise ValueError("2E must stay below 2^31 (int32 edge offsets)")
    if idx.shape[1]:
        if int(idx[0].min()) < 0 or int(idx[0].max()) >= U or int(idx[1].min()) < 0 or int(idx[1].max()) >= I:
            raise ValueError("edge endpoint out of range for U=%d, I=%d" % (U, I))
    ranges = _ranges(row_range, N)
    eu, ei = idx[0], idx[1]
    if row_range is None:
        csr_key = torch.unique(eu * I + ei)                  # row-major sorted + unique == coalesce()
        csc_key = torch.sort((csr_key % I) * U + csr_key // I).values      # CSC order: item-major, users ascending
    else:
        in_u = torch.zeros(eu.numel(), dtype=torch.bool, device=dev)
        in_i = torch.zeros(eu.numel(), dtype=torch.bool, device=dev)
        for lo, hi in ranges:
            in_u |= (eu >= lo) & (eu < hi)
            in_i |= (ei + U >= lo) & (ei + U < hi)
        csr_key = torch.unique(eu[in_u] * I + ei[in_u])
        csc_key = torch.unique(ei[in_i] * U + eu[in_i])
    if require_user_edges and row_range is None and U:
        deg_u = torch.bincount(csr_key // I, minlength=U)
        if int(deg_u.min()) == 0:
            # the reference asserts e_rowsum != 0 for every user (SPUIGACF.py:368)
            raise ValueError("user %d has no interaction: the reference asserts on a zero attention row-sum"
                             % int(torch.argmin(deg_u)))
    return build_plan_from_keys(csr_key, csc_key, U, I, ranges, seg_len, short_max, hub_cap, long_max)


def _ranges(row_range, N):
    if row_range is None:
        ranges = [(0, N)]
    elif isinstance(row_range[0], (tuple, list)):
        ranges = [tuple(r) for r in row_range]
    else:
        ranges = [tuple(row_range)]
    for lo, hi in ranges:
        if not 0 <= lo <= hi <= N:
            raise ValueError("row range %s outside [0, %d]" % ((lo, hi), N))
    return ranges


def default_hub_cap(n_edges_directed):
    """Largest degree a single workgroup takes when the workgroup-per-row layout is asked for (hub_cap=None).  NOT the
    default layout: measured, it loses.  A wave's step over 32 edges is latency-bound (~1 us, not the ~0.15 us of the
    launch-wide average), so a 14 k-edge row in one workgroup is a 110 us tail on a 27 us launch (Gowalla scale: 34 -> 99 us
    per launch), and on the 10^9-edge graph 1.27 M rows of 300-1000 edges as one workgroup each doubled the wave count for
    the same edges (111 -> 207 ms per pass).  Per-row parallelism has to grow with the degree -- which is what the
    256-edge slices + slot-order partial-sum pass do.  Kept as a tested option (build_plan(hub_cap=None))."""
    return max(4096, int(0.015 * n_edges_directed))


def build_plan_from_keys(csr_key, csc_key, U, I, ranges, seg_len=256, short_max=64, hub_cap=0, long_max=None):
    """The plan of the dst rows in `ranges` from the two sorted, unique key arrays a rank owns:
         csr_key = user * I + item   for every edge whose USER row is in the ranges,
         csc_key = item * U + user   for every edge whose ITEM row is in the ranges
    (a full plan passes every edge in both).  rowptr keeps global-row indexing -- N+1 int32, empty outside the ranges --
    so the kernels are the same for whole and sharded plans; `col` holds only these edges."""
    dev = csr_key.device
    N = U + I
    if csr_key.numel() + csc_key.numel() >= 2 ** 31:
        raise ValueError("a plan's directed edges must stay below 2^31 (int32 edge offsets)")
    deg_u = torch.bincount(csr_key // I, minlength=U) if csr_key.numel() else torch.zeros(U, dtype=torch.long, device=dev)
    deg_i = torch.bincount(csc_key // U, minlength=I) if csc_key.numel() else torch.zeros(I, dtype=torch.long, device=dev)
    E = int(csr_key.numel())
    rowptr = torch.zeros(N + 1, dtype=torch.long, device=dev)
    rowptr[1:] = torch.cumsum(torch.cat([deg_u, deg_i]), 0)
    col = torch.cat([csr_key % I + U, csc_key % U]).to(torch.int32).contiguous()
    lo, hi = ranges[0][0], ranges[-1][1]
    rows = torch.cat([torch.arange(a, b, device=dev) for a, b in ranges])
    deg = rowptr[rows + 1] - rowptr[rows]
    is_short = deg <= short_max
    is_user = rows < U
    # work order of a launch: user rows first, then item rows (each half gathers only the other type's table)
    sr = []
    for m in (is_short & is_user, is_short & ~is_user):
        order = torch.sort(deg[m], descending=True, stable=True).indices
        sr.append(rows[m][order])
    pad = (-sr[0].numel()) % 16
    nshort_user = sr[0].numel() + pad
    srow = torch.cat([sr[0], torch.full((pad,), -1, dtype=torch.long, device=dev), sr[1]]).to(torch.int32).contiguous()
    # Classes by degree:  short (<= short_max): a lane group per row;  long (<= long_max): one wave per row;  beyond: cut
    # into seg_len-edge slices (one wave each) + the slot-order partial-sum pass.  hub_cap = 0 (default): long_max =
    # seg_len and nothing else.  OPTIONAL workgroup-per-row class (hub_cap = None or > 0): rows of long_max < degree <=
    # hub_cap take one workgroup each, folded through LDS, long_max = 4 * seg_len -- measured slower (default_hub_cap).
    cap = default_hub_cap(int(col.numel())) if hub_cap is None else int(hub_cap)
    lmax = (4 * seg_len if cap > 0 else seg_len) if long_max is None else int(long_max)
    lmax = max(lmax, short_max) if cap > 0 else seg_len
    is_hub = ~is_short & (deg > lmax) & (deg <= cap)
    is_big = ~is_short & ~is_hub & (deg > lmax)
    is_long = ~is_short & ~is_hub & ~is_big
    hr = []
    for m in (is_hub & is_user, is_hub & ~is_user):
        order = torch.sort(deg[m], descending=True, stable=True).indices
        hr.append(rows[m][order])
    hrow = torch.cat(hr).to(torch.int32).contiguous()
    nhub_user = int(hr[0].numel())

    def both(sel_user):
        """segments of the long rows (one each) and of the rows beyond the cap (slices), slices' slots numbered first"""
        big = _segments(rowptr, rows[is_big & sel_user], seg_len)
        one = _segments(rowptr, rows[is_long & sel_user], max(lmax, 1))
        segs_ = torch.cat([big[0], one[0]])
        order = torch.sort(segs_[:, 2] - segs_[:, 1], descending=True, stable=True).indices    # longest first
        return segs_[order].contiguous(), big[1], big[2]
    seg_u, split_u, np_u = both(is_user)
    seg_i, split_i, np_i = both(~is_user)
    seg_i[:, 3] = torch.where(seg_i[:, 3] >= 0, seg_i[:, 3] + np_u, seg_i[:, 3])
    split_i[:, 1] += np_u
    segs = torch.cat([seg_u, seg_i]).contiguous()
    split = torch.cat([split_u, split_i]).contiguous()
    nparts = np_u + np_i
    plan = GraphPlan(U, I, E, rowptr, col, segs, split, nparts, seg_len, lo, hi, srow, short_max, ranges, hrow, nhub_user)
    plan.hub_cap, plan.long_max = cap, lmax
    plan.nseg_user, plan.nshort_user, plan.nsplit_user = int(seg_u.shape[0]), int(nshort_user), int(split_u.shape[0])
    validate_plan(plan)
    return plan


def validate_plan(p):
    """Host-side check of every bound the kernels assume."""
    n2e = p.col.numel()
    if n2e:
        assert int(p.col.min()) >= 0 and int(p.col.max()) < p.N, "source id out of range"
    if p.nseg:
        s = p.segs.long()
        assert int(s[:, 0].min()) >= 0 and int(s[:, 0].max()) < p.N, "segment row out of range"
        assert bool((s[:, 1] <= s[:, 2]).all()) and int(s[:, 1].min()) >= 0 and int(s[:, 2].max()) <= n2e, \
            "segment edge range out of bounds"
        assert int(s[:, 3].max()) < max(p.nparts, 1) and int(s[:, 3].min()) >= -1, "partial-sum slot out of range"
        # user rows only see item sources and vice versa
        urow = s[:, 0] < p.U
        nonempty = s[:, 1] < s[:, 2]
        if bool((urow & nonempty).any()):
            assert int(p.col[s[urow & nonempty, 1]].min()) >= p.U
        if bool((~urow & nonempty).any()):
            assert int(p.col[s[~urow & nonempty, 1]].max()) < p.U
    if p.nsplit:
        sp = p.split.long()
        assert int((sp[:, 1] + sp[:, 2]).max()) <= p.nparts and int(sp[:, 1].min()) >= 0
        assert 0 <= p.nsplit_user <= p.nsplit and bool((sp[:p.nsplit_user, 0] < p.U).all()) and bool((sp[p.nsplit_user:, 0] >= p.U).all()), \
            "hub rows: users first, then items"
        ps = p.part_split.long()
        assert ps.numel() == p.nparts and bool((sp[ps, 1] <= torch.arange(p.nparts, device=ps.device)).all()) \
            and bool((torch.arange(p.nparts, device=ps.device) < sp[ps, 1] + sp[ps, 2]).all()), "slot -> hub row map is inconsistent"
    n_real_short = 0
    if p.nshort:
        assert p.nshort_user % 16 == 0 or p.nshort_user == p.nshort
        for part, want_user in ((p.srow[:p.nshort_user].long(), True), (p.srow[p.nshort_user:].long(), False)):
            real = part[part >= 0]
            n_real_short += int(real.numel())
            if want_user:
                assert bool((part[:real.numel()] >= 0).all()), "padding must trail the user list"
            else:
                assert real.numel() == part.numel(), "no padding inside the item list"
            if real.numel():
                assert int(real.max()) < p.N and bool(((real < p.U) == want_user).all()), "short row in the wrong half"
                d = p.rowptr[real + 1] - p.rowptr[real]
                assert int(d.max()) <= p.short_max and bool((d[1:] <= d[:-1]).all()), \
                    "short rows must be sorted by degree, descending, within each half"
    if p.nseg:
        su = p.segs[:p.nseg_user, 0].long()
        si = p.segs[p.nseg_user:, 0].long()
        assert (su.numel() == 0 or int(su.max()) < p.U) and (si.numel() == 0 or int(si.min()) >= p.U)
    assert int(p.rowptr[-1]) == n2e and int(p.rowptr[-1]) < 2 ** 31
    if p.nhub:
        h = p.hrow.long()
        assert 0 <= p.nhub_user <= p.nhub and int(h.min()) >= 0 and int(h.max()) < p.N
        assert bool((h[:p.nhub_user] < p.U).all()) and bool((h[p.nhub_user:] >= p.U).all()), "hub rows: users first, then items"
        hd = p.rowptr[h + 1] - p.rowptr[h]
        assert int(hd.min()) > getattr(p, "long_max", p.seg_len) and int(torch.unique(h).numel()) == p.nhub, \
            "hub rows are beyond long_max, each listed once"
        if p.nseg:
            assert not bool(torch.isin(h, p.segs[:, 0].long()).any()), "a hub row must not also be a segment row"
    covered = n_real_short + (int(torch.unique(p.segs[:, 0]).numel()) if p.nseg else 0) + p.nhub
    assert covered == p.n_rows, "every dst row of the plan's ranges must be covered exactly once"
