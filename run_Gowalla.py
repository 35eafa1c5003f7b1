#!/usr/bin/env python3
"""Counterpart of the reference's run_Gowalla.py (:41-194) on the HIP path: same flags, same epoch /
eval / checkpoint cadence, same scalar names.  Additions: --dtype, --stages, --data_root, --synthetic.

  python run_Gowalla.py --dataset ml100k --model SPUIGACF --adj_type ui_mat --train_mode PairSampling \
      --eval_mode AllNeg --epochs 2 --eval_every 1 --lr 0.002 --weight_decay 0.000001 --droprate 0.2
"""
import argparse
import ast
import os
import time

import numpy as np
import torch

import ngacf_amd
from ngacf_amd import checkpoint as CK, data as D
from ngacf_amd.train_eval import Interactions, eval_neg_all, train_bpr


def prepare_data(args, dev):
    """run_Gowalla.py:41-94: interactions, 80/20 split for ml100k/ml1m PairSampling (:80), given train/test CSVs
    for Gowalla/Yelp (data/loadGowalla.py:35-45)."""
    if args.synthetic:
        U, I, E = D.SHAPES[args.synthetic]
        idx = D.synthetic_bipartite(U, I, E, seed=args.seed, device="cpu").numpy()
        u, i = idx[0], idx[1]
        split = True
    elif args.dataset == "ml100k":
        path = os.path.join(args.data_root, "1K", "u.data") if args.data_root else None
        u, i, _ = D.load_ml100k(path if path and os.path.exists(path) else None)
        U, I = int(u.max()) + 1, int(i.max()) + 1
        split = True
    elif args.dataset in ("Gowalla", "Yelp"):
        pre = {"Gowalla": "g", "Yelp": "y"}[args.dataset]
        root = os.path.join(args.data_root or ".", args.dataset)
        tr = np.loadtxt(os.path.join(root, pre + "_train.csv"), delimiter=",", dtype=np.int64, usecols=(0, 1))
        te = np.loadtxt(os.path.join(root, pre + "_test.csv"), delimiter=",", dtype=np.int64, usecols=(0, 1))
        U = int(max(tr[:, 0].max(), te[:, 0].max())) + 1
        I = int(max(tr[:, 1].max(), te[:, 1].max())) + 1
        return Interactions(tr[:, 0], tr[:, 1], te[:, 0], te[:, 1], U, I, dev)
    else:
        raise SystemExit("dataset %s: provide --data_root or --synthetic" % args.dataset)
    if split:
        try:                                      # the reference's split (:80), global numpy RNG seeded at :193
            from sklearn.model_selection import train_test_split
            tr, te = train_test_split(np.arange(len(u)), test_size=0.2)
        except ImportError:
            perm = np.random.permutation(len(u))
            te, tr = perm[:len(u) // 5], perm[len(u) // 5:]
        # keep users with at least one train interaction evaluable: test rows of train-less users are dropped
        return Interactions(u[tr], i[tr], u[te], i[te], U, I, dev)


def create_model(args, U, I, dev):
    kw = dict(embedSize=args.embedSize, layers=args.layers, droprate=args.droprate)
    if args.stages:
        m = ngacf_amd.SPUIStackedGACF(U, I, stages=args.stages, **kw)
    elif args.model == "SPUIGACF":
        m = ngacf_amd.SPUIGACF(U, I, **kw)
    elif args.model == "SPUIMultiGACF":
        m = ngacf_amd.SPUIMultiGACF(U, I, **kw)
    else:
        raise SystemExit("model %s is not on the accelerated path" % args.model)
    m = m.to(dev)
    ngacf_amd.set_compute_dtype(m, torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    return m


def main(args):
    dev = torch.device("cuda", int(args.gpu_id.split(",")[0]))
    torch.cuda.set_device(dev)
    data = prepare_data(args, dev)
    adj = data.adjacency()
    print("userNum:{}, itemNum:{}".format(data.U, data.I))
    print("lenth of traindf", data.train_u.numel(), "lenth of test_df", data.test_u.numel())
    print("adj.shape", tuple(adj.shape))
    model = create_model(args, data.U, data.I, dev)
    lossfn = ngacf_amd.BPRLoss()
    # torch.optim.Adam's arithmetic and state_dict; its step() is one ngacf_adam_step launch with the step counters on the
    # device, so the optimizer step is part of the replayed hipGraph of a batch (train_bpr(graph=True))
    optim = ngacf_amd.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    os.makedirs("ckpts", exist_ok=True)
    if args.resume_from:
        # a checkpoint of the reference (run_Gowalla.py:143) or of this runner: both halves in the reference's layout;
        # the optimizer state is converted to the fused stage parameters (ngacf_amd/checkpoint.py)
        CK.load_checkpoint("ckpts/{}_{}_{:03d}.pkl".format(args.model, args.dataset, args.resume_from), model, optim, map_location=dev)
        print("=> loaded checkpoint 'ckpts/{}_{:03d}.pkl'".format(args.model, args.resume_from))
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed)
    history = []
    for epoch in range(args.resume_from, args.epochs):
        t0 = time.time()
        loss = train_bpr(model, args.batch_size, data, adj, optim, lossfn, gen, graph=not args.eager)
        torch.cuda.synchronize()
        print("------epoch:{}, train_loss:{:5f}, time consuming:{:.2f}s".format(epoch, loss, time.time() - t0), flush=True)
        if (epoch + 1) % args.save_every == 0:
            CK.save_checkpoint("ckpts/{}_{}_{:03d}.pkl".format(args.model, args.dataset, epoch + 1), model, optim)
        if (epoch + 1) % args.eval_every == 0:
            t0 = time.time()
            m = eval_neg_all(model, args.batch_size, data, adj)
            print("epoch:{} metrics:{}".format(epoch, {k: np.round(v, 5).tolist() if k != "auc" else v for k, v in m.items()}))
            print("The time of evaluate epoch {:03d} is: {:.2f}s".format(epoch, time.time() - t0), flush=True)
            history.append((epoch, loss, m))
    return history


def parse(argv=None):
    p = argparse.ArgumentParser(description="Neural Graph Attention Collaborative Filtering (MI355X path)")
    p.add_argument("--dataset", type=str, default="ml100k")
    p.add_argument("--model", type=str, default="SPUIGACF")
    p.add_argument("--adj_type", type=str, default="ui_mat")
    p.add_argument("--epochs", type=int, default=200)
    p.add_argument("--eval_every", type=int, default=10)
    p.add_argument("--save_every", type=int, default=10)
    p.add_argument("--resume_from", type=int, default=0)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--weight_decay", type=float, default=0.00001)
    p.add_argument("--batch_size", type=int, default=2048)
    p.add_argument("--droprate", type=float, default=0.1)
    p.add_argument("--train_rate", type=float, default=0.7)
    p.add_argument("--seed", type=int, default=2019)
    p.add_argument("--embedSize", type=int, default=64)
    p.add_argument("--layers", type=ast.literal_eval, default=[64, 64])
    p.add_argument("--train_mode", type=str, default="PairSampling")
    p.add_argument("--eval_mode", type=str, default="AllNeg")
    p.add_argument("--parallel", type=ast.literal_eval, default=False)
    p.add_argument("--gpu_id", type=str, default="0")
    p.add_argument("--dtype", type=str, default="fp32", choices=["fp32", "bf16"])
    p.add_argument("--stages", type=int, default=0, help="0 = the model's own depth; L>0 = SPUIStackedGACF with L stages")
    p.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of one hipGraph replay per batch")
    p.add_argument("--data_root", type=str, default="")
    p.add_argument("--synthetic", type=str, default="", choices=["", "ml100k", "gowalla", "yelp"])
    a = p.parse_args(argv)
    if a.parallel:
        # the reference's --parallel True wraps the model in nn.DataParallel (parallel.py:94-130): every GPU redoes the whole
        # propagation.  This package's modules refuse to run as replicas; the multi-GPU path is one process per GPU
        raise SystemExit("--parallel True (nn.DataParallel replicas) is not supported: the multi-GPU path is "
                         "ngacf_amd.dist.ShardedSPUIGACF under torchrun, e.g. `python bench.py --gpus 8`")
    if a.adj_type != "ui_mat" or a.train_mode != "PairSampling" or a.eval_mode != "AllNeg":
        raise SystemExit("the accelerated path is --adj_type ui_mat --train_mode PairSampling --eval_mode AllNeg (README.md:13-29)")
    return a


if __name__ == "__main__":
    args = parse()
    torch.manual_seed(args.seed)                 # run_Gowalla.py:191-193
    torch.cuda.manual_seed_all(args.seed)
    np.random.seed(args.seed)
    main(args)
