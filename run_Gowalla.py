#!/usr/bin/env python3
"""Counterpart of the reference's run_Gowalla.py (:41-194) on the HIP path: same flags, same epoch /
eval / checkpoint cadence, same tensorboard scalar names (written when tensorboardX or torch's tensorboard writer is installed:
loss/train_loss, metrics@K/{precision,recall,ndcg,hit_ratio}, metrics/HR, metrics/NDCG -- :139,149-159).
Additions: --dtype, --stages, --data_root, --synthetic, --renumber.

Modes (:163-183): --train_mode PairSampling --eval_mode AllNeg (BPR, all-item ranking: the README's commands) and
--train_mode NegSampling --eval_mode SampledNeg (BCE on 1 + 4 sampled items, HR / NDCG@10 on 1 + 99: ml100k is split leave-one-out,
:77).  --model SPUIGACF | SPUIMultiGACF | SPUIGAGPCF.  --adj_type: the attention layers read (user, item) pairs, i.e. ui_mat
(SPUIGACF.py:340-350); SPUIGAGPCF additionally propagates through a Laplacian -- the matrix --adj_type names (mean_adj when it says
ui_mat).  (The reference's own createModels cannot build SPUIGAGPCF: :101-102 reads an `adj` that is not in scope, and its
prepareData hands the model ONE index list for both uses.)

  python run_Gowalla.py --dataset ml100k --model SPUIGACF --adj_type ui_mat --train_mode PairSampling \
      --eval_mode AllNeg --epochs 2 --eval_every 1 --lr 0.002 --weight_decay 0.000001 --droprate 0.2

Several GPUs (where the reference has --parallel True = nn.DataParallel replicas, :104-113): one process per GPU,

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run_Gowalla.py <the same flags>

runs the SAME loop on a node partition: embedding table, Adam state and edges sharded over the ranks, one RCCL all-gather per
half-table per propagation stage, every rank ranking its share of the users in the evaluation (ngacf_amd/sharded_loop.py).
"""
import argparse
import ast
import os
import time

import numpy as np
import torch

import ngacf_amd
from ngacf_amd import checkpoint as CK, data as D, sharded_loop as SL
from ngacf_amd.train_eval import Interactions, eval_neg_all, eval_neg_sample, split_loo, train_bpr, train_neg_sample


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
        u, i, _, t = D.load_ml100k(path if path and os.path.exists(path) else None, with_time=True)
        U, I = int(u.max()) + 1, int(i.max()) + 1
        if args.train_mode == "NegSampling":           # leave-one-out: a user's latest row is the test row (:77, loadGowalla.py:307-313)
            tr, te = split_loo(u, i, t)
            return Interactions(u[tr], i[tr], u[te], i[te], U, I, dev, renumber=args.renumber)
        split = True
    elif args.dataset in ("Gowalla", "Yelp"):
        pre = {"Gowalla": "g", "Yelp": "y"}[args.dataset]
        root = os.path.join(args.data_root or ".", args.dataset)
        tr = np.loadtxt(os.path.join(root, pre + "_train.csv"), delimiter=",", dtype=np.int64, usecols=(0, 1))
        te = np.loadtxt(os.path.join(root, pre + "_test.csv"), delimiter=",", dtype=np.int64, usecols=(0, 1))
        U = int(max(tr[:, 0].max(), te[:, 0].max())) + 1
        I = int(max(tr[:, 1].max(), te[:, 1].max())) + 1
        return Interactions(tr[:, 0], tr[:, 1], te[:, 0], te[:, 1], U, I, dev, renumber=args.renumber)
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
        return Interactions(u[tr], i[tr], u[te], i[te], U, I, dev, renumber=args.renumber)


def laplacian(args, adj, U, I):
    """the matrix SPUIGAGPCF's GPLayers propagate through (data/loadGowalla.py:184-227), as COO over the U + I nodes"""
    from ngacf_amd import spmm
    kind = "mean_adj" if args.adj_type == "ui_mat" else args.adj_type
    return {"mean_adj": spmm.mean_adj_coo, "norm_adj": spmm.norm_adj_coo, "plain_adj": spmm.plain_adj_coo}[kind](adj, U, I)


def create_model(args, U, I, dev, adj=None):
    kw = dict(embedSize=args.embedSize, layers=args.layers, droprate=args.droprate)
    if args.stages:
        m = ngacf_amd.SPUIStackedGACF(U, I, stages=args.stages, **kw)
    elif args.model == "SPUIGACF":
        m = ngacf_amd.SPUIGACF(U, I, **kw)
    elif args.model == "SPUIMultiGACF":
        m = ngacf_amd.SPUIMultiGACF(U, I, **kw)
    elif args.model == "SPUIGAGPCF":                   # (:101-102)
        m = ngacf_amd.SPUIGAGPCF(U, I, laplacian(args, adj, U, I), **kw)
    else:
        raise SystemExit("model %s is not on the accelerated path" % args.model)
    m = m.to(dev)
    ngacf_amd.set_compute_dtype(m, torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    return m


def main(args):
    sharded = SL.in_torchrun()
    if sharded:                                       # one process per GPU under torchrun: LOCAL_RANK picks the device
        rank, world, dev = SL.init()
    else:
        rank, world = 0, 1
        dev = torch.device("cuda", int(args.gpu_id.split(",")[0]))
        torch.cuda.set_device(dev)
    say = print if rank == 0 else (lambda *a, **k: None)
    data = prepare_data(args, dev)
    adj = data.adjacency()
    say("userNum:{}, itemNum:{}".format(data.U, data.I))
    say("lenth of traindf", data.train_u.numel(), "lenth of test_df", data.test_u.numel())
    say("adj.shape", tuple(adj.shape))
    # nodes renumbered by degree at ingest (ngacf_amd.data.degree_order): the model, the sampler and the evaluation speak the
    # internal numbering; checkpoints are written and read in the data set's (the reference's) numbering
    perm = (data.perm_u, data.perm_i) if data.perm_u is not None else None
    if sharded:
        model = SL.create_model(args.model, args.stages, data.U, data.I, args.embedSize, args.layers, args.droprate, adj, dev,
                                torch.bfloat16 if args.dtype == "bf16" else torch.float32)
        say("node partition over %d ranks: %d user rows + %d item rows per rank, %d of %d directed edges on rank 0; half-table gathers: %s"
            % (world, model.part.cu, model.part.ci, model._plan(adj).n_edges_directed, 2 * adj.shape[1],
               "peer stores (NGACF_PEER_GATHER=1)" if model.comm.peer_store else "all-gather"))
        train_fn, eval_fn, save_fn, load_fn = SL.train_bpr, SL.eval_neg_all, SL.save_checkpoint, SL.load_checkpoint
    else:
        model = create_model(args, data.U, data.I, dev, adj)
        train_fn, eval_fn, save_fn, load_fn = train_bpr, eval_neg_all, CK.save_checkpoint, CK.load_checkpoint
    pair_mode = args.train_mode == "PairSampling"
    lossfn = ngacf_amd.BPRLoss() if pair_mode else torch.nn.BCEWithLogitsLoss()      # :104-113
    writer = summary_writer(args) if rank == 0 else None
    # torch.optim.Adam's arithmetic and state_dict; its step() is one ngacf_adam_step launch with the step counters on the
    # device, so the optimizer step is part of the replayed hipGraph of a batch (train_bpr(graph=True))
    optim = ngacf_amd.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
    if rank == 0:
        os.makedirs("ckpts", exist_ok=True)
    if args.resume_from:
        # a checkpoint of the reference (run_Gowalla.py:143) or of this runner (any world size): both halves in the reference's
        # layout; the optimizer state is converted to the fused stage parameters (ngacf_amd/checkpoint.py)
        load_fn("ckpts/{}_{}_{:03d}.pkl".format(args.model, args.dataset, args.resume_from), model, optim, map_location=dev, perm=perm)
        say("=> loaded checkpoint 'ckpts/{}_{:03d}.pkl'".format(args.model, args.resume_from))
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed)                        # (the same on every rank: every rank draws the same batches)
    history = []
    # a sharded step replays as ONE hipGraph (kernels + RCCL launches) only on request: captured collectives are rehearsed on
    # one rank only, the eager loop is the safe default there
    graph = (not args.eager) if not sharded else args.sharded_graph
    for epoch in range(args.resume_from, args.epochs):
        t0 = time.time()
        if pair_mode:
            loss = train_fn(model, args.batch_size, data, adj, optim, lossfn, gen, graph=graph and args.model != "SPUIGAGPCF")
        else:
            loss = train_neg_sample(model, args.batch_size, data, adj, optim, lossfn, gen)
        torch.cuda.synchronize()
        if sharded:
            model.comm.check()                        # (peer-store transport: a timed-out wait means stale rows -- stop before saving / evaluating)
        if writer is not None:
            writer.add_scalar("loss/train_loss", loss, epoch)                    # :139
        say("------epoch:{}, train_loss:{:5f}, time consuming:{:.2f}s".format(epoch, loss, time.time() - t0), flush=True)
        if (epoch + 1) % args.save_every == 0:
            save_fn("ckpts/{}_{}_{:03d}.pkl".format(args.model, args.dataset, epoch + 1), model, optim, perm=perm)
        if (epoch + 1) % args.eval_every == 0:
            t0 = time.time()
            if args.eval_mode == "AllNeg":
                m = eval_fn(model, args.batch_size, data, adj)
                if sharded:
                    model.comm.check()
                say("epoch:{} metrics:{}".format(epoch, {k: np.round(v, 5).tolist() if k != "auc" else v for k, v in m.items()}))
                if writer is not None:
                    for k_i, K in enumerate([1, 5, 10, 20]):                     # :149-154
                        for name in ("precision", "recall", "ndcg", "hit_ratio"):
                            writer.add_scalar("metrics@{}/{}".format(K, name), float(m[name][k_i]), epoch)
            else:
                HR, NDCG = eval_neg_sample(model, args.batch_size, data, adj, 10, gen)    # :156
                say("epoch:{}, HR:{:5f}, NDCG:{:5f}".format(epoch, HR, NDCG))
                if writer is not None:
                    writer.add_scalar("metrics/HR", HR, epoch)                   # :158-159
                    writer.add_scalar("metrics/NDCG", NDCG, epoch)
                m = {"HR": HR, "NDCG": NDCG}
            say("The time of evaluate epoch {:03d} is: {:.2f}s".format(epoch, time.time() - t0), flush=True)
            history.append((epoch, loss, m))
    if writer is not None:
        writer.close()
    if sharded:
        import torch.distributed as dist
        model.comm.check()                            # (peer-store transport: no wait ever timed out)
        dist.barrier()
    return history


def summary_writer(args):
    """The reference's SummaryWriter (run_Gowalla.py:120-122: results under ./runs) when a tensorboard writer is installed; None
    otherwise (tensorboardX is not a dependency of the accelerated path)."""
    comment = "_DS:{}_M:{}_E:{}_L:{}_lr:{}_wd:{}_dp:{}_rs:{}_parallel:{}".format(
        args.dataset, args.model, args.embedSize, args.layers, args.lr, args.weight_decay, args.droprate, args.seed, args.parallel)
    for mod in ("tensorboardX", "torch.utils.tensorboard"):
        try:
            return __import__(mod, fromlist=["SummaryWriter"]).SummaryWriter(comment=comment)
        except Exception:
            continue
    return None


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
    p.add_argument("--sharded_graph", action="store_true", help="under torchrun: replay each batch as one hipGraph (kernels + RCCL launches)")
    p.add_argument("--renumber", type=str, default="degree", choices=["degree", "none"],
                   help="internal node numbering: by degree, descending (hot rows contiguous: faster gathers) or the data set's; "
                        "checkpoints are in the data set's numbering either way")
    p.add_argument("--data_root", type=str, default="")
    p.add_argument("--synthetic", type=str, default="", choices=["", "ml100k", "gowalla", "yelp"])
    a = p.parse_args(argv)
    if a.parallel:
        # the reference's --parallel True wraps the model in nn.DataParallel (parallel.py:94-130): every GPU redoes the whole
        # propagation.  Here several GPUs mean one process per GPU on a node partition: the same script under torchrun
        import sys
        raise SystemExit("--parallel True (nn.DataParallel replicas, every GPU redoing the whole propagation) is replaced by a node "
                         "partition with one process per GPU.  Run:\n  " + SL.torchrun_command(sys.argv[1:] if argv is None else argv))
    modes = (a.train_mode, a.eval_mode)
    if modes not in (("PairSampling", "AllNeg"), ("NegSampling", "SampledNeg")):
        raise SystemExit("--train_mode / --eval_mode: PairSampling + AllNeg or NegSampling + SampledNeg (the two combinations the "
                         "reference's prepareData serves, run_Gowalla.py:84-92)")
    if a.adj_type not in ("ui_mat", "plain_adj", "norm_adj", "mean_adj"):
        raise SystemExit("--adj_type: ui_mat, plain_adj, norm_adj or mean_adj")
    if a.adj_type != "ui_mat" and a.model != "SPUIGAGPCF":
        raise SystemExit("--adj_type %s: the attention layers of %s read (user, item) pairs, i.e. ui_mat (SPUIGACF.py:340-350); the "
                         "Laplacian variants feed SPUIGAGPCF's static-weight layers" % (a.adj_type, a.model))
    if SL.in_torchrun() and (modes[0] != "PairSampling" or a.model == "SPUIGAGPCF"):
        raise SystemExit("under torchrun (node partition): --model SPUIGACF | SPUIMultiGACF with PairSampling + AllNeg")
    return a


if __name__ == "__main__":
    args = parse()
    torch.manual_seed(args.seed)                 # run_Gowalla.py:191-193
    torch.cuda.manual_seed_all(args.seed)
    np.random.seed(args.seed)
    main(args)
