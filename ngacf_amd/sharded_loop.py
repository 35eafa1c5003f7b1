"""The train / eval loop of run_Gowalla.py over a node partition: `torchrun --nproc-per-node P run_Gowalla.py ...`, one process per
GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI).  It stands where the reference's `--parallel True` stands
(run_Gowalla.py:104-113, train_eval_Gowalla.py:97-104,131-138, parallel.py:94-196: nn.DataParallel replicas that each redo the whole
propagation); here the embedding table, its Adam state and the edges are SHARDED (ngacf_amd/dist.py) and every rank runs the same
loop:

  train   every rank draws the SAME batch (the sampler is a counter hash of (seed, row): same seed -> same pairs everywhere),
          scores it through ShardedSPUIGACF.forward (propagation over its own rows, one all-gather per half-table per stage, a
          2B-row exchange for the scores), takes the same loss, and the backward leaves each rank with the gradients of ITS
          embedding rows; the stage weights' gradients are summed in one grouped all-reduce (reduce_grads) in front of the
          optimizer step, so the replicated weights stay bit-identical on all ranks.
  eval    the propagated table is all-gathered once (data-set numbering), rank r ranks the users r, r+P, r+2P, ... with
          ngacf_eval_topk, the metric sums are all-reduced.
  ckpt    rank 0 writes the REFERENCE's layout (run_Gowalla.py:142-143): the embedding shards and their Adam moments are
          gathered; any world size (or the single-GPU runner, or the reference) can resume from it.

Nothing here computes: the kernels are the library's, the collectives torch.distributed's.
"""
import os

import torch
import torch.distributed as dist

from . import checkpoint as CK
from .dist import ShardedSPUIGACF
from .train_eval import eval_neg_all as _eval_neg_all, train_bpr as _train_bpr


def torchrun_command(argv, nproc="<GPUs>"):
    """The command line that replaces the reference's `--parallel True`."""
    args = [a for a in argv if a not in ("--parallel", "True", "--parallel=True")]
    return "python -m torch.distributed.run --nnodes=1 --nproc-per-node %s --master-addr 127.0.0.1 run_Gowalla.py %s" % (nproc, " ".join(args))


def in_torchrun():
    return int(os.environ.get("WORLD_SIZE", "1")) > 1


def init(device=None, backend=None):
    """(rank, world, device).  backend: "nccl" on GPUs (RCCL), "gloo" on CPU tensors (tests) or when NGACF_DIST_BACKEND says so."""
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    if device is None:
        device = torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")
    backend = backend or os.environ.get("NGACF_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, world, device


def create_model(model_name, stages, userNum, itemNum, embedSize, layers, droprate, adj, device, dtype=torch.float32, backend=None,
                 world=None, rank=None):
    """ShardedSPUIGACF for --model SPUIGACF (2 stages) / SPUIMultiGACF (3) / --stages L; same initial values on every rank as the
    unsharded model under the same torch seed (the full tables are drawn, each rank keeps its rows)."""
    L = stages or {"SPUIGACF": 2, "SPUIMultiGACF": 3}.get(model_name)
    if L is None:
        raise SystemExit("model %s is not on the accelerated path" % model_name)
    m = ShardedSPUIGACF(userNum, itemNum, embedSize, layers, droprate, world=world, rank=rank, stages=L, adj=adj, backend=backend).to(device)
    m.compute_dtype = dtype
    m.comm.defer_weight_reduce = True                     # one grouped all-reduce per step (reduce_grads), not one per stage
    return m


def train_bpr(model, batch_size, data, adj, optim, lossfn, generator=None, log_every=0, graph=False):
    """train_eval_Gowalla.py:90-144 on the sharded model: the single-GPU loop with reduce_grads() between backward and step.
    Returns the same epoch loss on every rank."""
    return _train_bpr(model, batch_size, data, adj, optim, lossfn, generator, log_every, graph, after_backward=model.reduce_grads)


def eval_neg_all(model, batch_size, data, adj, topk=None):
    """train_eval_Gowalla.py:274-354: every rank ranks its share of the users; identical metric dict on every rank."""
    p = model.part
    return _eval_neg_all(model, batch_size, data, adj, shard=(p.rank, p.P), topk=topk,
                         all_reduce=(lambda t: dist.all_reduce(t, group=model.comm.group)) if model.comm.active else None)


# ---------------------------------------------------------------------------------------------------------------------------
# checkpoints in the reference's layout
# ---------------------------------------------------------------------------------------------------------------------------
def _full_table(model, u_local, i_local):
    """[U + I, d] in the data set's numbering from this rank's user / item rows (an all-gather; every rank gets it)."""
    full = model._gathered(u_local, i_local)
    return full[:model.userNum].clone(), full[model.userNum:].clone()


def full_optim_state(model, optim):
    """The optimizer's state_dict in the reference's parameter layout with the embedding moments gathered to full tables
    (collective: every rank calls it, every rank gets the same dict)."""
    sd = optim.state_dict()
    ref = CK.optim_state_to_reference(model, sd)          # per-head stage entries; entries 0 / 1 still hold the LOCAL rows
    st_u, st_i = ref["state"].get(0), ref["state"].get(1)
    if st_u is not None and st_i is not None:
        dev = model.uEmbd.weight.device
        for k in list(st_u):
            if torch.is_tensor(st_u[k]) and st_u[k].dim() == 2:
                fu, fi = _full_table(model, st_u[k].to(dev), st_i[k].to(dev))
                st_u[k], st_i[k] = fu.cpu(), fi.cpu()
    return ref


def save_checkpoint(path, model, optim, perm=None):
    """run_Gowalla.py:142-143: {'model': ..., 'optim': ...} in the reference's layout, written by rank 0 (collective call).
    perm: (perm_u, perm_i) of a run that renumbered its nodes at ingest -- the file speaks the data set's numbering."""
    msd = model.full_state_dict()
    osd = full_optim_state(model, optim)
    if model.part.rank == 0:
        msd = {k: v.cpu() for k, v in msd.items()}
        if perm is not None:
            msd, osd = CK.renumber_reference_checkpoint(msd, osd, perm[0].cpu(), perm[1].cpu())
        torch.save({"model": msd, "optim": osd}, path)
    if model.comm.active:
        dist.barrier(group=model.comm.group)


def load_checkpoint(path, model, optim, map_location=None, perm=None):
    """run_Gowalla.py:127-130 for the sharded model: a checkpoint written by the reference, by the single-GPU runner or by
    save_checkpoint above at ANY world size.  Every rank reads the file and keeps its rows.  perm: as in save_checkpoint."""
    ck = torch.load(path, map_location="cpu")
    osd = ck["optim"]
    n = len(osd["param_groups"][0]["params"]) if len(osd["param_groups"]) == 1 else -1
    if n != CK.n_reference_params(model):
        raise ValueError("the sharded runner resumes from checkpoints in the reference's optimizer layout (%d parameters), got %d"
                         % (CK.n_reference_params(model), n))
    if perm is not None:
        ck = dict(ck)
        ck["model"], osd = CK.renumber_reference_checkpoint(ck["model"], osd, CK._inverse(perm[0].cpu()), CK._inverse(perm[1].cpu()))
    model.load_full_state_dict(ck["model"])
    ou, oi = model.part.local_original_ids("cpu")
    ids = osd["param_groups"][0]["params"]
    state = dict(osd["state"])
    for j, orig in ((0, ou), (1, oi)):                    # entries 0 / 1: uEmbd.weight / iEmbd.weight (reference order)
        st = state.get(ids[j])
        if st is None:
            continue
        st = dict(st)
        for k, v in list(st.items()):
            if torch.is_tensor(v) and v.dim() == 2:
                loc = torch.zeros(orig.numel(), v.shape[1], dtype=v.dtype)
                ok = orig >= 0
                loc[ok] = v[orig[ok]]                     # padding slots keep zero moments
                st[k] = loc
        state[ids[j]] = st
    osd = CK.optim_state_from_reference(model, {"state": state, "param_groups": osd["param_groups"]})
    keep = {k: v for k, v in optim.state_dict()["param_groups"][0].items() if k in ("fused", "capturable", "foreach", "differentiable")}
    osd["param_groups"][0].update(keep)
    optim.load_state_dict(osd)
    return ck
