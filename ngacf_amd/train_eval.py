"""Counterparts of the reference's callers of the hot path:

  train_bpr     train_eval_Gowalla.py:90-144   sequential un-shuffled batches; per row the positive
                                               AND the negative are re-sampled from the user's sets
                                               (data/loadGowalla.py:69-77); 2 forwards per step;
                                               returns total_loss / len(train rows)
  eval_neg_all  train_eval_Gowalla.py:274-354  all-item ranking; candidates = item pool minus the
                                               user's *train* positives (loadGowalla.py:63-67), hits =
                                               test positives, top-20 (heapq.nlargest :370-385), metrics
                                               of graphattention/metrics.py incl. the NDCG quirk
                                               (:63-73), sums divided by the number of *train* users (:283)

  train_neg_sample  train_eval_Gowalla.py:36-88  --train_mode NegSampling: per train row its item (label 1) and 4 sampled
                                               negatives (label 0) from item pool - ALL the user's positives
                                               (loadGowalla.py:63-67,79-83); BCEWithLogitsLoss; total_loss / rows
  eval_neg_sample   train_eval_Gowalla.py:193-259  --eval_mode SampledNeg: per test row its item and 99 sampled negatives,
                                               top-10 of the 100 scores, HR / NDCG (graphattention/evaluation.py:4-13)
  split_loo         data/loadGowalla.py:307-313   the leave-one-out split those two modes use on ml100k / ml1m

What changes is where the work happens: the per-user negative `set`s (the 60-100 GB of README.md:19,25)
become a sorted key array + rejection sampling on the device, and the evaluation computes the propagated
embeddings once (the reference recomputes a bit-identical propagation for every 64 x 2048 score block).
"""
import numpy as np
import torch

from . import ops

KS = (1, 5, 10, 20)


class Interactions:
    """Train/test interactions as device tensors.

    train_u/train_i: rows of train_df in file order (the batch order of train_bpr);
    item pool: the distinct items of train ∪ test (`rt`, run_Gowalla.py:46,85)."""

    def __init__(self, train_u, train_i, test_u, test_i, U, I, device, renumber=None):
        """renumber="degree": users and items get internal ids by their degree in the TRAIN interactions (ngacf_amd.data.degree_order;
        the test half must not shape anything the ranking sees: ngacf_eval_topk breaks exact score ties toward the lower internal
        item id, i.e. toward the item with more train interactions, then the lower data-set id -- the reference breaks them in its
        sets' iteration order; exact ties need equal fp32 / bf16 dot products) -- every tensor of this object, the adjacency it
        hands out and hence the model speak the internal numbering; perm_u / perm_i (data set id -> internal id) are kept for
        whoever has to speak the data set's (checkpoints)."""
        dev = torch.device(device)
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.long, device=dev)
        self.U, self.I, self.device = U, I, dev
        self.train_u, self.train_i = t(train_u), t(train_i)
        self.test_u, self.test_i = t(test_u), t(test_i)
        self.perm_u = self.perm_i = None
        if renumber == "degree":
            from .data import degree_order
            key = torch.unique(self.train_u * I + self.train_i)
            self.perm_u, self.perm_i = degree_order(torch.stack([key // I, key % I]), U, I)
            self.train_u, self.train_i = self.perm_u[self.train_u], self.perm_i[self.train_i]
            self.test_u, self.test_i = self.perm_u[self.test_u], self.perm_i[self.test_i]
        elif renumber not in (None, "none"):
            raise ValueError("renumber: 'degree' or None")
        self.train_keys = torch.unique(self.train_u * I + self.train_i)            # sorted (user, item) keys
        self.test_keys = torch.unique(self.test_u * I + self.test_i)
        ku = self.train_keys // I
        self.train_ptr = torch.zeros(U + 1, dtype=torch.long, device=dev)
        self.train_ptr[1:] = torch.cumsum(torch.bincount(ku, minlength=U), 0)     # CSR of the train positives
        self.train_items = self.train_keys % I
        self.train_items32 = self.train_items.to(torch.int32)
        pool = torch.zeros(I, dtype=torch.bool, device=dev)
        pool[self.train_i] = True
        pool[self.test_i] = True
        self.pool_mask = pool
        self.pool_mask_u8 = pool.to(torch.uint8)
        self.pool_items = torch.nonzero(pool).squeeze(1)
        self.pool_items32 = self.pool_items.to(torch.int32)
        self.train_users = torch.unique(self.train_u)                              # len(train_pos_neg), :283
        tu = self.test_keys // I
        self.test_ptr = torch.zeros(U + 1, dtype=torch.long, device=dev)
        self.test_ptr[1:] = torch.cumsum(torch.bincount(tu, minlength=U), 0)
        self.test_items = self.test_keys % I

    def all_positives(self):
        """CSR of every user's positives over train and test (positives_negtives(rt), data/loadGowalla.py:63-67): what the
        NegSampling / SampledNeg modes draw their negatives away from.  (ptr int64 [U+1], items int32)"""
        if getattr(self, "_allpos", None) is None:
            key = torch.unique(torch.cat([self.train_keys, self.test_keys]))
            ptr = torch.zeros(self.U + 1, dtype=torch.long, device=self.device)
            ptr[1:] = torch.cumsum(torch.bincount(key // self.I, minlength=self.U), 0)
            self._allpos = (ptr, (key % self.I).to(torch.int32))
        return self._allpos

    def adjacency(self):
        """`ui_mat` indices over train ∪ test, coalesced (data/loadGowalla.py:184-253, run_Gowalla.py:82,94)."""
        key = torch.unique(torch.cat([self.train_keys, self.test_keys]))
        return torch.stack([key // self.I, key % self.I])


def sample_pairs(data, users, generator=None, max_rounds=64):
    """For every user of the batch: one positive uniformly from the user's train positives and one negative
    uniformly from (item pool − train positives)  (data/loadGowalla.py:69-77).  Device-side rejection
    sampling against the sorted key array replaces the per-user Python sets."""
    dev, I = data.device, data.I
    g = generator
    beg, end = data.train_ptr[users], data.train_ptr[users + 1]
    r = torch.rand(users.numel(), device=dev, generator=g)
    pos = data.train_items[beg + torch.clamp((r * (end - beg)).long(), max=(end - beg - 1).clamp_min(0))]
    neg = torch.empty_like(pos)
    todo = torch.arange(users.numel(), device=dev)
    for _ in range(max_rounds):
        cand = data.pool_items[torch.randint(0, data.pool_items.numel(), (todo.numel(),), device=dev, generator=g)]
        key = users[todo] * I + cand
        j = torch.searchsorted(data.train_keys, key).clamp_(max=data.train_keys.numel() - 1)
        ok = data.train_keys[j] != key
        neg[todo[ok]] = cand[ok]
        todo = todo[~ok]
        if todo.numel() == 0:
            break
    else:
        raise RuntimeError("negative sampling did not converge (a user is positive on the whole item pool)")
    return pos, neg


SAMPLE_STRIDE = 0x2545F4914F6CDD1D                    # sampler seed step from one batch to the next


def _epoch_seed(generator, dev):
    if generator is None:
        return int(torch.randint(0, 2 ** 62, (1,)).item())                # torch's CPU generator (torch.manual_seed)
    return int(torch.randint(0, 2 ** 62, (1,), device=generator.device, generator=generator).item())


def train_bpr(model, batch_size, data, adj, optim, lossfn, generator=None, log_every=0, graph=False, after_backward=None):
    """One epoch (train_eval_Gowalla.py:90-144).  On the GPU the pairs come from ngacf_sample_pairs (one launch per
    batch, no host synchronisation); with graph=True every full batch is ONE hipGraph replay (sampler, 2 forwards,
    loss, backward, optimizer: ngacf_amd.graphs.GraphedBPRStep; the optimizer must be capturable) and only a ragged
    last batch runs eagerly.  Both modes draw the same pairs for the same seed.
    after_backward: called between loss.backward() and optim.step() (the node-partitioned model's grouped all-reduce of
    the stage-weight gradients, ShardedSPUIGACF.reduce_grads)."""
    model.train()
    n = data.train_u.numel()
    B = batch_size
    if data.device.type != "cuda":
        return _train_bpr_host_sampler(model, B, data, adj, optim, lossfn, generator, log_every, after_backward)
    seed0 = _epoch_seed(generator, data.device)
    total = torch.zeros((), device=data.device)
    unres = torch.zeros(1, dtype=torch.int32, device=data.device)
    first_eager = 0
    if graph and n >= B:
        from .graphs import graphed_epoch_step
        gs = graphed_epoch_step(model, lossfn, optim, adj, B, data, after_backward=after_backward)
        gs.begin_epoch(seed0)
        for _ in range(n // B):
            gs.replay()
        total += gs.total
        unres += gs.unresolved
        first_eager = n // B
    n_batches = n // B + 1                                            # train_eval_Gowalla.py:108
    for b in range(first_eager, n_batches):
        rows = min(B, n - b * B)                                      # sequential slices, no shuffling (:111-115)
        if rows <= 0:
            continue
        upn, _ = ops.sample_pairs(data.train_u, b * B, rows, data.train_ptr, data.train_items32, data.pool_items32,
                                  seed0 + b * SAMPLE_STRIDE, unresolved=unres)
        optim.zero_grad(set_to_none=True)
        pos_scores = model(upn[0], upn[1], adj)                       # two full propagations with independent
        neg_scores = model(upn[0], upn[2], adj)                       # dropout masks (:131-132)
        loss = lossfn(pos_scores, neg_scores)
        loss.backward()
        if after_backward is not None:
            after_backward()
        optim.step()
        total += loss.detach()
        if log_every and b % log_every == 0:
            print("  batch %d/%d loss %.5f" % (b, n_batches, float(loss.detach())), flush=True)
    if int(unres) != 0:
        raise RuntimeError("negative sampling did not converge (a user is positive on the whole item pool), or a "
                           "batch row / user id outside the interaction table reached the sampler")
    ops.raise_if_bad_indices(data.device)                             # (the epoch's one host check of the index counters)
    return float(total) / n                                           # :144 (sum of batch means / rows)


def _train_bpr_host_sampler(model, batch_size, data, adj, optim, lossfn, generator=None, log_every=0, after_backward=None):
    """The same loop with the torch statement of the sampler (CPU tensors: host-logic tests)."""
    n = data.train_u.numel()
    total = torch.zeros((), device=data.device)
    n_batches = n // batch_size + 1
    for b in range(n_batches):
        users = data.train_u[b * batch_size:(b + 1) * batch_size]
        if users.numel() == 0:
            continue
        pos, neg = sample_pairs(data, users, generator)
        optim.zero_grad(set_to_none=True)
        loss = lossfn(model(users, pos, adj), model(users, neg, adj))
        loss.backward()
        if after_backward is not None:
            after_backward()
        optim.step()
        total += loss.detach()
    return float(total) / n


def metrics_from_hits(r, npos, n_norm):
    """r: [B,20] 0/1 hit lists (rank order), npos: [B] number of test positives.  Returns the four
    metric vectors @KS summed over the batch and divided by n_norm (graphattention/metrics.py)."""
    r = r.double()
    disc = 1.0 / torch.log2(torch.arange(2, 22, device=r.device, dtype=torch.double))
    ideal = torch.sort(r, dim=1, descending=True).values              # the reference's quirk: ideal DCG from r itself
    out = {k: [] for k in ("precision", "recall", "ndcg", "hit_ratio")}
    for K in KS:
        rk = r[:, :K]
        hits = rk.sum(1)
        dcg = (rk * disc[:K]).sum(1)
        idcg = (ideal[:, :K] * disc[:K]).sum(1)
        out["precision"].append((hits / K).sum() / n_norm)
        out["recall"].append((hits / npos.double()).sum() / n_norm)
        out["ndcg"].append(torch.where(idcg > 0, dcg / idcg.clamp_min(1e-300), torch.zeros_like(dcg)).sum() / n_norm)
        out["hit_ratio"].append((hits > 0).double().sum() / n_norm)
    return {k: torch.stack(v) for k, v in out.items()}


@torch.no_grad()
def eval_neg_all(model, batch_size, data, adj, user_batch=1 << 20, return_scores=False, shard=None, all_reduce=None, topk=None):
    """All-item evaluation.  Users = those with test positives that also appear in train (the merge at
    train_eval_Gowalla.py:307); normaliser = number of train users (:283).
    shard = (rank, world) + all_reduce(tensor): every rank ranks its share of the users (users[rank::world]) on the
    propagated table and the metric sums are added over the ranks (ngacf_amd/sharded_loop.py); topk: stand-in for
    ops.eval_topk with the same signature (the CPU tests of the loop logic; the product never passes it)."""
    model.eval()
    F = getattr(model, "score_table", model.propagate)(adj).contiguous()     # once: eval mode is deterministic
    U, I = data.U, data.I
    has_test = (data.test_ptr[1:] - data.test_ptr[:-1]) > 0
    has_train = (data.train_ptr[1:] - data.train_ptr[:-1]) > 0
    users = torch.nonzero(has_test & has_train).squeeze(1)
    if shard is not None:
        users = users[shard[0]::shard[1]].contiguous()
    n_norm = float(data.train_users.numel())
    res = {k: torch.zeros(len(KS), dtype=torch.double, device=data.device) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    kept = []
    # (a table wider than the ranking kernel takes -- SPUIGAGPCF's concatenated stage outputs -- is ranked by the torch statement)
    rank_items = topk if topk is not None else (ops.eval_topk if F.shape[1] in ops.SUPPORTED_D else topk_torch)
    for s in range(0, users.numel(), user_batch):
        ub = users[s:s + user_batch].contiguous()
        # scoring F_u . F_i^T on the matrix cores fused with the top-20 selection (ngacf_eval_topk): candidates =
        # pool - train positives, best first, ties to the lower item id (= the stable sort of heapq.nlargest's input)
        top, top_s = rank_items(F, ub, U, I, data.pool_mask_u8, data.train_ptr, data.train_items32, 20)
        if return_scores:
            kept.append(F[ub].float() @ F[U:].float().t())
        valid = top >= 0                                              # fewer than 20 candidates
        key = ub[:, None] * I + top.long().clamp_(min=0)
        j = torch.searchsorted(data.test_keys, key.reshape(-1)).clamp_(max=data.test_keys.numel() - 1)
        r = (data.test_keys[j] == key.reshape(-1)).reshape(top.shape) & valid
        npos = data.test_ptr[ub + 1] - data.test_ptr[ub]
        m = metrics_from_hits(r, npos, n_norm)
        for k in res:
            res[k] += m[k]
    if all_reduce is not None:
        for k in res:
            all_reduce(res[k])
    out = {k: v.cpu().numpy() for k, v in res.items()}
    if data.device.type == "cuda":
        ops.raise_if_bad_indices(data.device)
    out["auc"] = 0.0                                                  # the reference reports 0 (:385)
    if return_scores:
        return out, users, torch.cat(kept)
    return out


def topk_torch(F, users, U, I, pool_mask, train_ptr, train_items, K=20, chunk=2048):
    """ops.eval_topk's contract in torch ops (a GEMM, a mask, a stable sort): the K best candidates of every user, candidates =
    pool - the user's train positives, ties to the lower item id, -1 where fewer than K exist.  Serves score tables of widths the
    ranking kernel does not take (SPUIGAGPCF's 64 + sum(layers) columns); the headline models never come here."""
    tops, vals = [], []
    items = F[U:].float()
    for s in range(0, users.numel(), chunk):
        ub = users[s:s + chunk]
        sc = F[ub].float() @ items.t()
        sc = sc.masked_fill(~pool_mask.bool()[None, :], float("-inf"))
        cnt = train_ptr[ub + 1] - train_ptr[ub]
        rows = torch.repeat_interleave(torch.arange(ub.numel(), device=F.device), cnt)
        sc[rows, _gather_rows(train_ptr, train_items.long(), ub)] = float("-inf")
        v, ix = torch.sort(sc, dim=1, descending=True, stable=True)
        v, ix = v[:, :K], ix[:, :K].to(torch.int32)
        ix[torch.isinf(v) & (v < 0)] = -1
        tops.append(ix)
        vals.append(v)
    return torch.cat(tops), torch.cat(vals)


def _pair_scores(F, u, i, U):
    """score[b] = F[u[b]] . F[U + i[b]]: ngacf_score_fwd for the widths it takes, the torch statement for wider tables"""
    if F.shape[1] in ops.SUPPORTED_D and F.is_cuda:
        return ops.score_fwd(F, u, i, U)
    return (F[u].float() * F[U + i].float()).sum(1)


def _gather_rows(ptr, items, rows):
    """Concatenate CSR rows `rows` without a Python loop."""
    beg, cnt = ptr[rows], ptr[rows + 1] - ptr[rows]
    total = int(cnt.sum())
    start = torch.cumsum(cnt, 0) - cnt
    idx = torch.arange(total, device=ptr.device) - torch.repeat_interleave(start, cnt) + torch.repeat_interleave(beg, cnt)
    return items[idx]


# ---------------------------------------------------------------------------------------------------------
# --train_mode NegSampling / --eval_mode SampledNeg (run_Gowalla.py:163-183; train_eval_Gowalla.py:36-88, :193-259)
# ---------------------------------------------------------------------------------------------------------
def split_loo(u, i, t):
    """data/loadGowalla.py:307-313: per user the row with the latest timestamp is the test row, the others train -- ties go to
    the row that comes first in the file (pandas rank(method='first', ascending=False)).  Returns (train rows, test rows), both
    in file order, as index arrays."""
    u, t = np.asarray(u), np.asarray(t)
    order = np.lexsort((np.arange(len(u)), -t.astype(np.int64), u))          # by user, then time descending, then file order
    first = np.ones(len(u), dtype=bool)
    first[1:] = u[order][1:] != u[order][:-1]
    test = np.zeros(len(u), dtype=bool)
    test[order[first]] = True
    return np.nonzero(~test)[0], np.nonzero(test)[0]


def sample_negatives(data, users, k, seed, margin=None):
    """[n, k] DISTINCT negatives per row, uniform over item pool - all positives of the row's user (random.sample(negative_items,
    k), loadGowalla.py:81,102): ngacf_sample_pairs draws `margin` candidates per row in one launch (rejection against the user's
    positives), the first k distinct ones are kept.  Rows that end with fewer than k (a user with fewer than k negatives) are
    counted in the returned int tensor."""
    dev = users.device
    M = margin or (2 * k if k <= 8 else k + max(32, k // 2))
    ptr, items32 = data.all_positives()
    rep = users.repeat_interleave(M).contiguous()
    upn, unres = ops.sample_pairs(rep, 0, rep.numel(), ptr, items32, data.pool_items32, seed)
    c = upn[2].view(-1, M)
    earlier = torch.ones(M, M, dtype=torch.bool, device=dev).tril(-1)         # [j, i]: i < j
    dup = ((c[:, :, None] == c[:, None, :]) & earlier).any(2)                 # candidate j repeats an earlier one of its row
    keep = ~dup
    rank = torch.cumsum(keep, 1)
    sel = keep & (rank <= k)
    short = (rank[:, -1] < k).sum() + unres.sum()
    pos = torch.argsort((~sel).to(torch.int8), dim=1, stable=True)[:, :k]      # the first k kept candidates, in draw order
    return torch.gather(c, 1, pos), short


def train_neg_sample(model, batch_size, data, adj, optim, lossfn=None, generator=None, negatives=None, log_every=0):
    """One epoch of --train_mode NegSampling (train_eval_Gowalla.py:36-88): sequential batches of train rows; per row the
    row's item with label 1 and 4 negatives with label 0; predictions = model(u x 5, items) in ONE propagation; BCEWithLogitsLoss
    (torch's, as the reference's run_Gowalla.py:109; the mean over the 5 B logits); returns total_loss / len(train rows) (:88).
    negatives: int [rows, 4] to use instead of sampling (tests: the draws a reference run recorded)."""
    import torch.nn.functional as Fn
    model.train()
    n, B = data.train_u.numel(), batch_size
    seed0 = _epoch_seed(generator, data.device) if negatives is None else 0
    total = torch.zeros((), device=data.device, dtype=torch.float64)
    short = torch.zeros((), device=data.device, dtype=torch.long)
    n_batches = n // B + 1                                            # :48
    for b in range(n_batches):
        users, pos = data.train_u[b * B:(b + 1) * B], data.train_i[b * B:(b + 1) * B]
        if users.numel() == 0:
            continue
        if negatives is None:
            neg, sh = sample_negatives(data, users, 4, seed0 + b * SAMPLE_STRIDE)
            short += sh
        else:
            neg = negatives[b * B:(b + 1) * B].to(data.device).long()
        items = torch.cat([pos[:, None], neg], 1)                    # (rows, 5): the positive first (:66-68)
        labels = torch.zeros_like(items, dtype=torch.float32)
        labels[:, 0] = 1.0
        u5 = users[:, None].expand_as(items).reshape(-1)             # every user 5 times (:75)
        optim.zero_grad(set_to_none=True)
        pred = model(u5.contiguous(), items.reshape(-1).contiguous(), adj)
        loss = (lossfn or Fn.binary_cross_entropy_with_logits)(pred.float(), labels.reshape(-1))
        loss.backward()
        optim.step()
        total += loss.detach().double()
        if log_every and b % log_every == 0:
            print("  batch %d/%d loss %.5f" % (b, n_batches, float(loss.detach())), flush=True)
    if negatives is None and int(short) != 0:
        raise RuntimeError("negative sampling: a user has fewer than 4 negatives, or the sampler did not converge")
    if data.device.type == "cuda":
        ops.raise_if_bad_indices(data.device)
    return float(total) / n


@torch.no_grad()
def eval_neg_sample(model, batch_size, data, adj, top_k=10, generator=None, negatives=None):
    """--eval_mode SampledNeg (train_eval_Gowalla.py:193-259): per test row the row's item and 99 sampled negatives; the row is a
    hit when its item is among the top_k of the 100 scores, NDCG = 1 / log2(position + 2) (graphattention/evaluation.py:4-13);
    returns (mean HR, mean NDCG) over the test rows.  The propagation runs ONCE (the reference: once per batch of
    batch_size // 8 rows, bit-identical each time); the position of the row's item = the number of sampled negatives that score
    higher (what torch.topk over the 100 scores yields, exact ties aside).
    negatives: int [rows, 99] to use instead of sampling."""
    model.eval()
    F = getattr(model, "score_table", model.propagate)(adj).contiguous()
    U = data.U
    users, pos = data.test_u, data.test_i
    n = users.numel()
    seed0 = _epoch_seed(generator, data.device) if negatives is None else 0
    hits = torch.zeros((), dtype=torch.float64, device=data.device)
    ndcg = torch.zeros((), dtype=torch.float64, device=data.device)
    short = torch.zeros((), device=data.device, dtype=torch.long)
    B = max(batch_size // 8, 1)                                       # :197
    for b in range(0, n, B):
        ub, pb = users[b:b + B], pos[b:b + B]
        if negatives is None:
            neg, sh = sample_negatives(data, ub, 99, seed0 + (b // B) * SAMPLE_STRIDE)
            short += sh
        else:
            neg = negatives[b:b + B].to(data.device).long()
        items = torch.cat([pb[:, None], neg], 1)                     # (rows, 100)
        u100 = ub[:, None].expand_as(items).reshape(-1).contiguous()
        sc = _pair_scores(F, u100, items.reshape(-1).contiguous(), U).view(items.shape)      # ngacf_score_fwd: F_u . F_i
        place = (sc[:, 1:] > sc[:, :1]).sum(1)                        # negatives ahead of the row's item
        hit = place < top_k
        hits += hit.double().sum()
        ndcg += torch.where(hit, 1.0 / torch.log2(place.double() + 2.0), torch.zeros_like(place, dtype=torch.float64)).sum()
    if negatives is None and int(short) != 0:
        raise RuntimeError("negative sampling: a user has fewer than 99 negatives, or the sampler did not converge")
    if data.device.type == "cuda":
        ops.raise_if_bad_indices(data.device)
    return float(hits) / n, float(ndcg) / n
