"""Adjacency sources for the propagation path.

  * synthetic_bipartite: the power-law bipartite generator of SURVEY.md §8(d) at Gowalla /
    Yelp / 1e9-edge shapes (the train CSVs of the reference's datasets are not available);
  * load_ml100k / ui_mat_indices: the reference's `ui_mat` adjacency (data/loadGowalla.py:
    184-253): coo(userId, itemId) over train+test interactions, coalesced -> [2,E] indices.
"""
import math
import os

import numpy as np
import torch

SHAPES = {          # (U, I, E) -- SURVEY.md §8
    "ml100k": (943, 1682, 100000),
    "gowalla": (29858, 40981, 1027370),
    "yelp": (45919, 45538, 1185065),
    # the REAL interaction lists the reference ships (its test halves; the train halves are missing blobs): real_graph() below
    "gowalla_real": (29858, 40981, 217242),
    "yelp_real": (45918, 45538, 253578),                       # (one of the 45 919 users has no pair in the test half: dropped)
    "synthetic1b": (10_000_000, 5_000_000, 1_000_000_000),
    "synthetic200m": (10_000_000, 5_000_000, 200_000_000),     # same node tables (1.9 GB bf16: far beyond the 256 MB
}                                                               # Infinity Cache), a fifth of the edges: true-HBM regime


def synthetic_bipartite(U, I, E, seed=2019, device="cpu", alpha_pop=0.8, sigma=1.0, chunk=1 << 26):
    """[2,E'] int64 indices, row-major sorted and unique (E' within ~1 % of E).  Item popularity
    p_i ~ (rank+1)^-alpha_pop, user degree ~ lognormal(sigma) rescaled to the target, every user
    has at least one edge (reference precondition, SPUIGACF.py:368), items may stay isolated."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    deg = torch.exp(sigma * torch.randn(U, device=dev, generator=g, dtype=torch.float64))
    over = 1.06                                            # duplicates are merged below
    deg = torch.clamp((deg * (E * over / float(deg.sum()))).round().long(), min=1, max=I)
    pop = torch.pow(torch.arange(1, I + 1, device=dev, dtype=torch.float64), -alpha_pop)
    perm = torch.randperm(I, device=dev, generator=g)      # popularity rank -> item id
    cdf = torch.cumsum(pop / pop.sum(), 0)
    total = int(deg.sum())
    users = torch.repeat_interleave(torch.arange(U, device=dev), deg)
    keys = []
    for s in range(0, total, chunk):
        n = min(chunk, total - s)
        r = torch.rand(n, device=dev, generator=g, dtype=torch.float64)
        items = perm[torch.searchsorted(cdf, r).clamp_(max=I - 1)]
        keys.append(torch.unique(users[s:s + n] * I + items))
    key = torch.unique(torch.cat(keys)) if len(keys) > 1 else keys[0]
    extra = key.numel() - E
    if extra > 0:                                          # thin out, never a user's first edge
        u = key // I
        first = torch.ones_like(u, dtype=torch.bool)
        first[1:] = u[1:] != u[:-1]
        cand = torch.nonzero(~first).squeeze(1)
        drop = cand[torch.randperm(cand.numel(), device=dev, generator=g)[:extra]]
        keep = torch.ones_like(first)
        keep[drop] = False
        key = key[keep]
    return torch.stack([key // I, key % I])


def synthetic_bipartite_chunks(U, I, E, seed=2019, device="cpu", alpha_pop=0.8, sigma=1.0, chunk_users=1 << 18):
    """The same generator family, produced USER CHUNK BY USER CHUNK: yields (u_lo, u_hi, keys) with `keys` the sorted
    unique int64 edge keys (user * I + item) of the users [u_lo, u_hi).  Every chunk has its own counter-seeded
    generator, so any rank can produce any chunk and all ranks agree on the graph without ever holding it whole (the
    10^9-edge graph of BASELINE config 5 is built shard-wise on the device this way: ngacf_amd.dist.sharded_graph).
    A user's edges live in exactly one chunk, so the concatenation is globally sorted and unique; E' is within ~2 % of E."""
    dev = torch.device(device)
    g0 = torch.Generator(device=dev)
    g0.manual_seed(seed)
    raw = torch.exp(sigma * torch.randn(U, device=dev, generator=g0, dtype=torch.float64))
    over = 1.06
    deg = torch.clamp((raw * (E * over / float(raw.sum()))).round().long(), min=1, max=I)
    target = torch.clamp((raw * (E / float(raw.sum()))).round().long(), min=1, max=I)     # per-user share of E
    pop = torch.pow(torch.arange(1, I + 1, device=dev, dtype=torch.float64), -alpha_pop)
    perm = torch.randperm(I, device=dev, generator=g0)
    cdf = torch.cumsum(pop / pop.sum(), 0)
    del raw, pop
    for c, lo in enumerate(range(0, U, chunk_users)):
        hi = min(U, lo + chunk_users)
        g = torch.Generator(device=dev)
        g.manual_seed((seed * 1000003 + c + 1) & 0x7FFFFFFFFFFFFFFF)
        d = deg[lo:hi]
        users = torch.repeat_interleave(torch.arange(lo, hi, device=dev), d)
        r = torch.rand(users.numel(), device=dev, generator=g, dtype=torch.float64)
        items = perm[torch.searchsorted(cdf, r).clamp_(max=I - 1)]
        key = torch.unique(users * I + items)
        extra = int(key.numel() - int(target[lo:hi].sum()))
        if extra > 0:                                      # thin out to the chunk's share, never a user's first edge
            u = key // I
            first = torch.ones_like(u, dtype=torch.bool)
            first[1:] = u[1:] != u[:-1]
            cand = torch.nonzero(~first).squeeze(1)
            drop = cand[torch.randperm(cand.numel(), device=dev, generator=g)[:extra]]
            keep = torch.ones_like(first)
            keep[drop] = False
            key = key[keep]
        yield lo, hi, key


def synthetic_bipartite_big(U, I, E, seed=2019, device="cpu", chunk_users=1 << 18):
    """[2,E'] indices of the chunked generator, whole (one GPU holds it: the N=1 leg of the big workloads)."""
    keys = [k for _, _, k in synthetic_bipartite_chunks(U, I, E, seed, device, chunk_users=chunk_users)]
    key = torch.cat(keys)
    return torch.stack([key // I, key % I])


def degree_order(edge, U, I):
    """(perm_u, perm_i), dataset id -> internal id: users and items renumbered by degree, descending (stable: ties keep the
    data set's order).  The rows a traversal gathers most often then sit next to each other -- measured on the bench graphs, whose
    generator (like real logs) scatters popular items over the id range: the attention-propagation pass runs 2.4 % faster at
    Gowalla scale, 3.4 % at Yelp scale and 11.6 % on the 2 x 10^8-edge graph (7.5 instead of 8.4 ms per stage = 0.94 of the HBM
    peak: DRAM pages and translations are shared by the hot rows; tools/renumber_probe.py).  A bijection: results are the same
    graph's, under other labels.  run_Gowalla.py renumbers at ingest (--renumber degree) and maps back in its checkpoints."""
    perms = []
    for ids, n in ((edge[0], U), (edge[1], I)):
        deg = torch.bincount(ids, minlength=n)
        order = torch.argsort(deg, descending=True, stable=True)          # internal id -> dataset id
        perm = torch.empty_like(order)
        perm[order] = torch.arange(n, device=order.device)
        perms.append(perm)
    return perms[0], perms[1]


def renumber(edge, perm_u, perm_i):
    """[2,E] indices under the internal numbering, row-major sorted again (the order the plan builder and the reference's
    coalesce() produce)."""
    u, i = perm_u[edge[0]], perm_i[edge[1]]
    key = torch.sort(u * int(perm_i.numel()) + i).values
    return torch.stack([key // int(perm_i.numel()), key % int(perm_i.numel())])


BIG_WORKLOADS = ("synthetic200m", "synthetic1b")
REAL_WORKLOADS = ("gowalla_real", "yelp_real")


def real_pairs(name):
    """(user int64[n], item int64[n], U, I): the interaction pairs of the reference's data/Gowalla/g_test.csv or data/Yelp/y_test.csv
    (loader layout data/loadGowalla.py:35-45), from the data fixture oracle/gen_real_pairs.py wrote
    (data/real/<name>_test_pairs.npz: two id columns, sorted row-major, unique = the coalesced ui_mat of :244-253).  A DATA file
    beside the package, not a test fixture (the product does not reach into tests/)."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "real", "%s_test_pairs.npz" % name)
    z = np.load(path)
    return z["user"].astype(np.int64), z["item"].astype(np.int64), int(z["U"]), int(z["I"])


def real_graph(name, device, isolated_users="drop"):
    """([2,E] indices, U, I) of the real test-half graph `gowalla_real` / `yelp_real`.  A user without a pair is what the
    reference's layer asserts against (SPUIGACF.py:368; here: ValueError at plan build) -- the Yelp test half has one.
    isolated_users = "drop": such users leave the id space (the others keep their order); "keep": the graph as the file has it
    (building a plan on it raises).  Isolated ITEMS stay: the reference handles them (:389)."""
    u, i, U, I = real_pairs(name[:-5] if name.endswith("_real") else name)
    if isolated_users == "drop":
        has = np.zeros(U, dtype=bool)
        has[u] = True
        if not has.all():
            new = np.cumsum(has) - 1
            u, U = new[u], int(has.sum())
    elif isolated_users != "keep":
        raise ValueError("isolated_users: 'drop' or 'keep'")
    return torch.from_numpy(np.stack([u, i])).to(device), U, I


def cluster_order(edge, U, I):
    """(perm_u, perm_i), dataset id -> internal id: a locality order BEYOND the degree order -- items by degree (descending);
    users by their most popular item (then by degree), so that users who share a hot item sit together; items then re-sorted by
    their heaviest user's position (ties: by degree).  Real check-in data cluster (a city's users visit that city's places);
    the generator's independent draws do not (tools/renumber_probe.py measures both)."""
    dev = edge.device
    du, di = torch.bincount(edge[0], minlength=U), torch.bincount(edge[1], minlength=I)
    rank_i = torch.empty(I, dtype=torch.long, device=dev)
    rank_i[torch.argsort(di, descending=True, stable=True)] = torch.arange(I, device=dev)
    best = torch.full((U,), I, dtype=torch.long, device=dev).scatter_reduce(0, edge[0], rank_i[edge[1]], "amin")
    order_u = torch.argsort(best * (int(du.max()) + 1) + (int(du.max()) - du), stable=True)       # by hottest item, then by degree
    perm_u = torch.empty(U, dtype=torch.long, device=dev)
    perm_u[order_u] = torch.arange(U, device=dev)
    first = torch.full((I,), U, dtype=torch.long, device=dev).scatter_reduce(0, edge[1], perm_u[edge[0]], "amin")
    order_i = torch.argsort(first * (int(di.max()) + 1) + (int(di.max()) - di), stable=True)
    perm_i = torch.empty(I, dtype=torch.long, device=dev)
    perm_i[order_i] = torch.arange(I, device=dev)
    return perm_u, perm_i


def rcm_order(edge, U, I):
    """(perm_u, perm_i) from a reverse Cuthill-McKee order of the bipartite graph (scipy, host): neighbours get nearby numbers."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    u, i = edge[0].cpu().numpy(), edge[1].cpu().numpy() + U
    N = U + I
    a = sp.coo_matrix((np.ones(2 * len(u), dtype=np.int8), (np.concatenate([u, i]), np.concatenate([i, u]))), shape=(N, N)).tocsr()
    order = reverse_cuthill_mckee(a, symmetric_mode=True).astype(np.int64)        # position -> node
    pos = np.empty(N, dtype=np.int64)
    pos[order] = np.arange(N)
    perm_u = np.argsort(np.argsort(pos[:U], kind="stable"), kind="stable")
    perm_i = np.argsort(np.argsort(pos[U:], kind="stable"), kind="stable")
    return torch.from_numpy(perm_u).to(edge.device), torch.from_numpy(perm_i).to(edge.device)


def workload_graph(name, device, seed=2019):
    """[2,E] indices of a named bench workload.  Gowalla / Yelp scale: synthetic_bipartite (the graphs of round 1);
    ml100k: the real file; synthetic200m / synthetic1b: the chunked generator (same graph at every GPU count)."""
    U, I, E = SHAPES[name]
    if name in REAL_WORKLOADS:
        return real_graph(name, device)[0]
    if name == "ml100k":
        u, i, _ = load_ml100k()
        return torch.from_numpy(ui_mat_indices(u, i, I)).to(device)
    if name in BIG_WORKLOADS:
        return synthetic_bipartite_big(U, I, E, seed, device)
    return synthetic_bipartite(U, I, E, seed=seed, device=device)


def uniform_bipartite(U, I, E, seed=2019, device="cpu"):
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    u = torch.cat([torch.arange(U, device=dev), torch.randint(0, U, (max(E - U, 0),), device=dev, generator=g)])
    i = torch.randint(0, I, (u.numel(),), device=dev, generator=g)
    key = torch.unique(u * I + i)
    return torch.stack([key // I, key % I])


def ui_mat_indices(user_ids, item_ids, I):
    """`.coalesce().indices()` of coo(userId, itemId): row-major sorted, duplicates merged."""
    key = np.unique(np.asarray(user_ids, dtype=np.int64) * I + np.asarray(item_ids, dtype=np.int64))
    return np.stack([key // I, key % I])


def load_ml100k(path=None, with_time=False):
    """(userId, itemId, rating[, timestamp]) zero-based, in file order.  `path` may be the reference's
    data/1K/u.data (tab separated, data/loadGowalla.py:19-21) or the committed fixture
    tests/golden/ml100k_interactions.npz (same rows, same order)."""
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                            "tests", "golden", "ml100k_interactions.npz")
    if path.endswith(".npz"):
        d = np.load(path)
        r = (d["userId"].astype(np.int64), d["itemId"].astype(np.int64), d["rating"].astype(np.int64))
        return r + (d["timestamp"].astype(np.int64),) if with_time else r
    raw = np.loadtxt(path, dtype=np.int64, delimiter="\t")
    r = (raw[:, 0] - 1, raw[:, 1] - 1, raw[:, 2])
    return r + (raw[:, 3],) if with_time else r


def algorithmic_bytes_stage(U, I, E, din, dout, elem_bytes):
    """forward bytes of one propagation stage."""
    N = U + I
    trav = 2 * E * (4 + dout * elem_bytes) + N * (2 * dout * elem_bytes + 8)
    trans = N * (din + dout) * elem_bytes
    return trav, trans


# ---------------------------------------------------------------------------------------------------------
# on-disk adjacency caches of the reference (data/loadGowalla.py:229-253: scipy `save_npz` files next to the CSVs)
# ---------------------------------------------------------------------------------------------------------
ADJ_FILES = {"ui_mat": "s_ui_adj.npz", "plain_adj": "s_plain_adj.npz", "norm_adj": "s_norm_adj.npz", "mean_adj": "s_mean_adj.npz"}


def load_adj_npz(path):
    """A scipy sparse .npz (any format save_npz wrote) -> (indices [2,nnz] int64, values [nnz] float32, shape),
    in `.coalesce()` order (row-major sorted, duplicates summed) like get_adj_mat's return (:253)."""
    import scipy.sparse as sp
    m = sp.load_npz(path).tocsr()
    m.sum_duplicates()
    m.sort_indices()
    coo = m.tocoo()
    return (np.stack([coo.row.astype(np.int64), coo.col.astype(np.int64)]), coo.data.astype(np.float32), tuple(m.shape))


def save_adj_npz(path, indices, values, shape):
    import scipy.sparse as sp
    indices = np.asarray(indices)
    sp.save_npz(path, sp.coo_matrix((np.asarray(values), (indices[0], indices[1])), shape=shape))


def load_ui_mat(root, U=None, I=None):
    """`s_ui_adj.npz` (the U x I interaction matrix, adj_type 'ui_mat') -> the [2,E] mask SPUIGACF.forward takes.
    The stored ratings are ignored by the attention path, as in the reference (only `.indices()` is used,
    train_eval_Gowalla.py:97-104)."""
    idx, _, shape = load_adj_npz(os.path.join(root, ADJ_FILES["ui_mat"]))
    if U is not None and (shape[0] > U or shape[1] > I):
        raise ValueError("s_ui_adj.npz is %s, model is %d x %d" % (shape, U, I))
    return idx
