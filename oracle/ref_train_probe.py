#!/usr/bin/env python3
"""A8 end to end: the REFERENCE's README smoke run (README.md:13-15: ml100k, SPUIGACF, PairSampling / AllNeg, lr 0.002,
wd 1e-6, droprate 0.2, 2 epochs, seed 2019) executed here on the CPU -- the reference's own train_bpr
(train_eval_Gowalla.py:90-144) and eval_neg_all (:274-354) on the split its runner makes (run_Gowalla.py:80 under the
seeds of :191-193) -- for several seeds of Python's `random` (which the reference leaves unseeded: its run-to-run
spread).  Writes tests/golden/g12_ref_ml100k_train.npz: per run the two epoch losses and the metric vectors.
TEST INFRASTRUCTURE; runs only in the build container (about 3 minutes per run on 8 cores).

  python oracle/ref_train_probe.py [random seeds ...]      (default: 0 1 2)

Shims (no reference source is modified or copied): device='cuda' -> 'cpu' (TorchFunctionMode); Tensor.cuda / Module.cuda ->
identity; torch.cuda.device_count -> 1; empty_cache -> no-op; the 0-dim loss's backward(torch.ones(1)) -> backward();
np.asfarray alias.  get_adj_mat (which writes an .npz next to the reference's data) is not called: the adjacency is
built in memory with the reference's buildLaplacianMat, as oracle/gen_golden.py does.
"""
import os
import random
import sys
import time

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

if not hasattr(np, "asfarray"):
    np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class CpuDevice(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get("device", None) == "cuda":
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def one_run(rseed, epochs=2):
    from sklearn.model_selection import train_test_split
    import graphattention.SPUIGACF as R
    from graphattention.BPRLoss import BPRLoss
    import train_eval_Gowalla as TE
    from data.loadGowalla import (load100KRatings, train_pos_neg_exclude_test, test_positives, buildLaplacianMat,
                                  scipySP_torchSP)
    torch.manual_seed(2019)                               # run_Gowalla.py:191-193
    np.random.seed(2019)
    random.seed(rseed)                                    # the reference never seeds `random`: this is its free variable
    rt = load100KRatings(REF + "/data/1K")
    U, I = int(rt["userId"].max()), int(rt["itemId"].max())
    rt["userId"] = rt["userId"] - 1
    rt["itemId"] = rt["itemId"] - 1
    train_df, test_rows = train_test_split(rt, test_size=0.2)                          # :80
    adj = scipySP_torchSP(buildLaplacianMat(rt, U, I, "ui_mat")).coalesce().indices()  # :82,:94 (no cache file)
    train_pos_neg = train_pos_neg_exclude_test(rt, train_df)                           # :85
    test_df = test_positives(test_rows)                                                # :87
    model = R.SPUIGACF(U, I, embedSize=64, layers=[64, 64], droprate=0.2, useCuda=False)    # createModels :96-99
    lossfn = BPRLoss()
    optim = torch.optim.Adam(model.parameters(), lr=0.002, weight_decay=0.000001)      # README.md:15
    losses = []
    for ep in range(epochs):
        t0 = time.time()
        losses.append(TE.train_bpr(model, 2048, train_df, train_pos_neg, adj, optim, lossfn, False))
        print("random seed %d epoch %d: loss %.6g (%.0f s)" % (rseed, ep, losses[-1], time.time() - t0), flush=True)
    res = TE.eval_neg_all(model, 2048, test_df, train_pos_neg, adj, I, False)
    print("random seed %d metrics:" % rseed, {k: np.round(v, 4).tolist() if k != "auc" else v for k, v in res.items()}, flush=True)
    return losses, res


def main():
    seeds = [int(x) for x in sys.argv[1:]] or [0, 1, 2]
    real = (torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.device_count, torch.cuda.empty_cache, torch.Tensor.backward)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.device_count = lambda: 1
    torch.cuda.empty_cache = lambda: None
    _bw = torch.Tensor.backward
    torch.Tensor.backward = lambda self, gradient=None, **k: _bw(self, None if (gradient is not None and self.dim() == 0) else gradient, **k)
    out = {"random_seeds": np.array(seeds)}
    try:
        with CpuDevice():
            for j, s in enumerate(seeds):
                losses, res = one_run(s)
                out["loss.%d" % j] = np.array(losses)
                for k in ("precision", "recall", "ndcg", "hit_ratio"):
                    out["%s.%d" % (k, j)] = np.asarray(res[k], dtype=np.float64)
    finally:
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.cuda.device_count, torch.cuda.empty_cache, torch.Tensor.backward = real
    path = os.path.join(OUT, "g12_ref_ml100k_train.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
