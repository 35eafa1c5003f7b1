"""Host (numpy) statement of the counter-hash dropout RNG of csrc/ngacf_common.hpp.
The kernels keep an element iff bits >= floor(p * 2^32) and scale it by 1/(1-p).
tests/test_gpu_parity.py pins the device bits to these functions bit for bit."""
import numpy as np

K1, K2, K3 = np.uint32(0x9E3779B1), np.uint32(0x85EBCA77), np.uint32(0xC2B2AE3D)


def lowbias32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def salt_of(seed, stream):
    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        return lowbias32(lo ^ lowbias32(np.uint32(hi + np.uint32(stream) * K3)))


def feat_bits(seed, n, c):
    n, c = np.asarray(n, dtype=np.uint32), np.asarray(c, dtype=np.uint32)
    with np.errstate(over="ignore"):
        return lowbias32((n * K1) ^ (c * K2 + salt_of(seed, 0)))


def edge_bits(seed, head, u, i_global):
    u, i = np.asarray(u, dtype=np.uint32), np.asarray(i_global, dtype=np.uint32)
    with np.errstate(over="ignore"):
        return lowbias32((u * K1) ^ (i * K2 + salt_of(seed, 1 + head)))


def threshold(p):
    t = float(np.float32(p)) * 4294967296.0
    return np.uint32(0xFFFFFFFF) if t >= 4294967295.0 else np.uint32(int(t))


def keep_scale(bits, p):
    """multiplier in {0, 1/(1-p)} (float32) for each element"""
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(bits >= threshold(p), scale, np.float32(0.0)).astype(np.float32)


def sample_pairs_host(users, train_ptr, train_items, pool_items, seed, max_rounds=64):
    """Host statement of csrc/sampler.hip (ngacf_sample_pairs): (pos, neg, unresolved) for the batch rows `users`."""
    users = np.asarray(users, dtype=np.int64)
    salt = salt_of(int(seed) & 0xFFFFFFFFFFFFFFFF, 0x5A17)
    B, P = users.size, len(pool_items)
    r = np.arange(B, dtype=np.uint32)
    p0 = np.asarray(train_ptr)[users]
    deg = np.asarray(train_ptr)[users + 1] - p0
    with np.errstate(over="ignore"):
        bp = lowbias32((r * K1) ^ salt)
    pi = (bp.astype(np.uint64) * np.maximum(deg, 1).astype(np.uint64)) >> np.uint64(32)
    pos = np.where(deg > 0, np.asarray(train_items)[np.minimum(p0 + pi.astype(np.int64), len(train_items) - 1)], 0).astype(np.int64)
    neg = np.full(B, -1, dtype=np.int64)
    for rnd in range(max_rounds):
        todo = np.nonzero(neg < 0)[0]
        if todo.size == 0:
            break
        with np.errstate(over="ignore"):
            bn = lowbias32((r[todo] * K1) ^ (np.uint32(rnd + 1) * K2 + salt))
        cand = np.asarray(pool_items)[((bn.astype(np.uint64) * np.uint64(P)) >> np.uint64(32)).astype(np.int64)]
        for k, c in zip(todo, cand):
            row = train_items[p0[k]:p0[k] + deg[k]]
            j = np.searchsorted(row, c)
            if j >= deg[k] or row[j] != c:
                neg[k] = c
    unresolved = int((neg < 0).sum())
    neg[neg < 0] = pool_items[0]
    return pos, neg, unresolved
