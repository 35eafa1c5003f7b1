"""world_size-2 (and 3) gloo runs of the node-partitioned path on CPU.  The local compute is the TEST-ONLY torch
stand-in tests/fake_backend.py (there is no GPU here); what is under test is ngacf_amd/dist.py: the partition,
the padded numbering, the in-place all-gathers in forward AND backward, the weight-gradient all-reduce and the
autograd wiring.  Expected values: the reference's own float64 run (tests/golden/g234_*.npz)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tag, multi, defer, q, d=64, stages=None):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import fake_backend
        from ngacf_amd.dist import ShardedSPUIGACF
        g = dict(np.load(tag if os.path.isabs(tag) else os.path.join(HERE, "golden", "g234_%s.npz" % tag)))
        U, I = int(g["U"]), int(g["I"])
        sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
        m = ShardedSPUIGACF(U, I, d, [64, 64], 0.0, world, rank, stages=stages or (3 if multi else 2), backend=fake_backend)
        m.load_full_state_dict(sd)
        m.comm.defer_weight_reduce = defer
        m.train()
        edge = torch.from_numpy(g["edge"])
        u, pos, neg = (torch.from_numpy(g[k]) for k in ("u", "pos", "neg"))
        ps, ns = m(u, pos, edge), m(u, neg, edge)
        loss = -torch.log(torch.sigmoid(ps - ns)).mean()
        loss.backward()
        if defer:
            m.reduce_grads()                  # one all-reduce per step instead of one per stage
        p = m.part
        nu, ni = p.n_real_local()
        err = {}
        smax = float(np.abs(g["pos_f64"]).max())
        err["scores"] = max(float((ps.detach().double() - torch.from_numpy(g["pos_f64"])).abs().max()),
                            float((ns.detach().double() - torch.from_numpy(g["neg_f64"])).abs().max())) / smax
        err["loss"] = abs(float(loss.detach()) - float(g["loss_f64"]))
        gu, gi = g["g_f64.uEmbd.weight"], g["g_f64.iEmbd.weight"]
        scale = max(np.abs(gu).max(), np.abs(gi).max())
        err["emb"] = max(float(np.abs(m.uEmbd.weight.grad[:nu].numpy() - gu[p.u_lo:p.u_lo + nu]).max()),
                         float(np.abs(m.iEmbd.weight.grad[:ni].numpy() - gi[p.i_lo:p.i_lo + ni]).max())) / scale
        err["pad"] = max(float(m.uEmbd.weight.grad[nu:].abs().max()) if nu < p.cu else 0.0,
                         float(m.iEmbd.weight.grad[ni:].abs().max()) if ni < p.ci else 0.0)
        worst = 0.0
        for k, gr in m.gat.reference_named("grad").items():
            ref = g["g_f64.gat." + k]
            worst = max(worst, float(np.abs(gr.numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-12))
        err["gat"] = worst
        full = m.full_state_dict()
        err["sd"] = max(float((full[k] - sd[k]).abs().max()) for k in sd)
        err["keys"] = set(full) == set(sd)
        q.put((rank, err))
        dist.destroy_process_group()
    except Exception as e:                                  # surface the failure in the parent
        import traceback
        q.put((rank, {"exception": "%s\n%s" % (e, traceback.format_exc())}))


@pytest.mark.parametrize("world,tag,multi,defer", [(2, "spuigacf", False, False), (2, "spuimultigacf", True, True),
                                                   (3, "spuigacf", False, True)])
def test_sharded_model_matches_reference_under_gloo(world, tag, multi, defer):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tag, multi, defer, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert "exception" not in err, err.get("exception")
        assert err["scores"] < 1e-5 and err["loss"] < 1e-5, (rank, err)
        assert err["emb"] < 2e-4 and err["gat"] < 2e-4, (rank, err)          # fp32 stand-in vs float64 reference
        assert err["pad"] == 0.0 and err["sd"] == 0.0 and err["keys"], (rank, err)


def test_partition_covers_every_node_once():
    from ngacf_amd.dist import Partition
    for U, I, P in ((50, 40, 2), (50, 40, 3), (7, 5, 8), (29858, 40981, 8)):
        seen = np.zeros(0, dtype=np.int64)
        parts = [Partition(U, I, P, r) for r in range(P)]
        assert all(p.Up >= U and p.Ip >= I and p.Up % P == 0 and p.Ip % P == 0 for p in parts)
        rows = np.concatenate([np.concatenate([np.arange(lo, hi) for lo, hi in p.row_ranges]) for p in parts])
        assert np.array_equal(np.sort(rows), np.arange(parts[0].N))
        assert sum(sum(p.n_real_local()) for p in parts) == U + I


def test_sharded_d128_four_stages_matches_oracle_under_gloo(tmp_path):
    """BASELINE config 3 (embedSize 128, L = 4, table sharded with one all-gather per stage) at world size 2: the
    expected values come from the float64 oracle on the unsharded model."""
    import torch
    from oracle import ngacf_oracle as O
    import ngacf_amd
    rs = np.random.RandomState(4)
    U, I, B = 70, 55, 48
    uu = np.concatenate([np.arange(U), rs.randint(0, U, 900)]); ii = np.concatenate([rs.randint(0, I - 3, U), rs.randint(0, I - 3, 900)])
    key = np.unique(uu.astype(np.int64) * I + ii)
    edge = np.stack([key // I, key % I])
    torch.manual_seed(9)
    ref = ngacf_amd.SPUIStackedGACF(U, I, 128, [64, 64], 0.0, useCuda=False, stages=4)
    sd = {k: v.detach().numpy().astype(np.float64) for k, v in ref.state_dict().items()}
    sd["uEmbd.weight"] *= 8; sd["iEmbd.weight"] *= 8      # scores O(1): the worker's plain fp32 -log(sigmoid) stays finite
    u, pos, neg = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
    loss, ps, ns, grads = O.bpr_step_grads(sd, u, pos, neg, edge, hidden=3)
    path = str(tmp_path / "d128.npz")
    np.savez(path, U=U, I=I, edge=edge, u=u, pos=pos, neg=neg, pos_f64=ps, neg_f64=ns, loss_f64=loss,
             **{"p." + k: v.astype(np.float32) for k, v in sd.items()}, **{"g_f64." + k: v for k, v in grads.items()})
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, path, False, True, q, 128, 4)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, err in res:
        assert "exception" not in err, err.get("exception")
        assert err["scores"] < 1e-5 and err["loss"] < 1e-5, (rank, err)
        assert err["emb"] < 5e-4 and err["gat"] < 5e-4, (rank, err)
        assert err["pad"] == 0.0 and err["sd"] == 0.0 and err["keys"], (rank, err)
