"""Test infrastructure, build container only: the REAL check-in / review topology the reference ships.

Reads the two interaction files mounted with the reference -- data/Gowalla/g_test.csv (217 242 pairs over 29 858 users x 40 981
items) and data/Yelp/y_test.csv (253 578 pairs over 45 919 x 45 538); the train halves are missing blobs -- with the loader's own
column layout (data/loadGowalla.py:35-45: headerless `userId,itemId,rating`) and writes the id columns as data fixtures:
    data/real/gowalla_test_pairs.npz, data/real/yelp_test_pairs.npz   {user int32[n], item int32[n], U, I}
Data, like tests/golden/ml100k_interactions.npz: numbers only, no source text.  The GPU box never reads /root/reference.
    python oracle/gen_real_pairs.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/data"
FILES = {"gowalla": ("Gowalla/g_test.csv", 29858, 40981), "yelp": ("Yelp/y_test.csv", 45919, 45538)}      # shapes: SURVEY.md section 8


def main():
    for name, (rel, U, I) in FILES.items():
        a = np.loadtxt(os.path.join(REF, rel), delimiter=",", dtype=np.int64)
        u, i = a[:, 0], a[:, 1]
        assert u.min() >= 0 and i.min() >= 0 and u.max() < U and i.max() < I, (name, u.max(), i.max())
        key = np.unique(u * I + i)                                    # the coalesced ui_mat of these pairs (loadGowalla.py:244-253)
        out = os.path.join(ROOT, "data", "real", "%s_test_pairs.npz" % name)
        np.savez_compressed(out, user=(key // I).astype(np.int32), item=(key % I).astype(np.int32), U=np.int32(U), I=np.int32(I))
        du, di = np.bincount(key // I, minlength=U), np.bincount(key % I, minlength=I)
        print("%s: %d pairs (%d unique), users with a pair %d / %d, items with a pair %d / %d, max degree user %d item %d -> %s (%d bytes)"
              % (name, len(u), len(key), int((du > 0).sum()), U, int((di > 0).sum()), I, du.max(), di.max(), out, os.path.getsize(out)))


if __name__ == "__main__":
    sys.exit(main())
