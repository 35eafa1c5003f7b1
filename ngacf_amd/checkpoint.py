"""Checkpoint interchange with the reference (run_Gowalla.py:127-130 resume, :142-143 save):

    torch.save({'model': model.state_dict(), 'optim': optim.state_dict()}, 'ckpts/{model}_{dataset}_{epoch:03d}.pkl')

The MODEL half interchanges as is: state_dict() / load_state_dict() of the *CF modules present the reference's per-head keys
(SPUIGACF.py:270-278: gat.attention_0.W_u [64,8], .a [1,16], ...).  The OPTIMIZER half does not by itself: this package
keeps ONE fused parameter pair per propagation stage (stage{s}_W [2,din,dout], stage{s}_a [2,dout]), so Adam holds 6
state entries for SPUIGACF where the reference's Adam holds 29 (53 for SPUIMultiGACF), and torch refuses to load a state
with a different parameter count.  The two functions below convert Adam-family state between the layouts -- element for
element, exp_avg / exp_avg_sq / max_exp_avg_sq sliced or assembled exactly like the weights, `step` copied -- so that

  * a checkpoint the reference wrote resumes here  (load_checkpoint),
  * a checkpoint written here resumes in the reference (save_checkpoint writes the reference's layout).
"""
import torch


def _layout(model):
    """[(reference key, index of our parameter, take, put)] in the reference's model.parameters() order.
    take(T) -> the reference-shaped piece of a tensor T shaped like our parameter; put(T, piece) writes it back."""
    names = [n for n, _ in model.named_parameters()]
    idx = {n: j for j, n in enumerate(names)}
    out = []
    for key in ("uEmbd.weight", "iEmbd.weight"):
        out.append((key, idx[key], lambda T: T, lambda T, v: T.copy_(v)))
    extra = [n for n in names if not (n in ("uEmbd.weight", "iEmbd.weight") or n.startswith("gat.stage"))]
    gat = model.gat
    for s, (layer_names, (din, dout, heads)) in enumerate(zip(gat.stage_layer_names(), gat.stage_shapes())):
        k = dout // heads
        iW, ia = idx["gat.stage%d_W" % s], idx["gat.stage%d_a" % s]
        for h, ln in enumerate(layer_names):
            sl = slice(h * k, (h + 1) * k)
            out.append(("gat.%s.W_u" % ln, iW, lambda T, sl=sl: T[0][:, sl], lambda T, v, sl=sl: T[0][:, sl].copy_(v)))
            out.append(("gat.%s.W_i" % ln, iW, lambda T, sl=sl: T[1][:, sl], lambda T, v, sl=sl: T[1][:, sl].copy_(v)))
            out.append(("gat.%s.a" % ln, ia, lambda T, sl=sl: torch.cat([T[0][sl], T[1][sl]])[None],
                        lambda T, v, sl=sl, k=k: (T[0][sl].copy_(v[0, :k]), T[1][sl].copy_(v[0, k:]))))
    for n in extra:                                     # e.g. SPUIGAGPCF's Affinelayers: same tensors on both sides
        out.append((n, idx[n], lambda T: T, lambda T, v: T.copy_(v)))
    return out, names


def _is_param_shaped(v, p):
    return torch.is_tensor(v) and v.dim() > 0 and tuple(v.shape) == tuple(p.shape)


def optim_state_to_reference(model, optim_sd):
    """An optimizer state_dict of THIS package's model (one param group) -> the reference's 29/53-parameter layout."""
    layout, names = _layout(model)
    params = [p for _, p in model.named_parameters()]
    if len(optim_sd["param_groups"]) != 1 or len(optim_sd["param_groups"][0]["params"]) != len(names):
        raise ValueError("expected one parameter group over the model's %d parameters" % len(names))
    ours = optim_sd["param_groups"][0]["params"]
    state = {}
    for j, (_, pi, take, _) in enumerate(layout):
        st = optim_sd["state"].get(ours[pi])
        if st is None:
            continue
        state[j] = {k: (take(v).detach().clone().contiguous() if _is_param_shaped(v, params[pi])
                        else (v.detach().clone() if torch.is_tensor(v) else v)) for k, v in st.items()}
    group = dict(optim_sd["param_groups"][0], params=list(range(len(layout))))
    # The reference's own layout (torch.optim.Adam defaults, run_Gowalla.py:114): default implementation flags and `step` as a
    # CPU float32 0-dim tensor.  ngacf_amd.optim.Adam runs with capturable=True and device-side steps; a reference Adam that
    # inherited those would take the slow capturable path and assert on CPU parameters.  (load_checkpoint re-applies the
    # RUNNING optimizer's flags and torch moves `step` where that optimizer wants it.)
    for k, v in (("capturable", False), ("foreach", None), ("fused", None)):
        if k in group:
            group[k] = v
    for st in state.values():
        if torch.is_tensor(st.get("step")):
            st["step"] = st["step"].detach().to("cpu", torch.float32).reshape(())
    return {"state": state, "param_groups": [group]}


def optim_state_from_reference(model, ref_sd):
    """The reference's optimizer state (29/53 parameters, model.parameters() order of the reference) -> this package's
    layout.  The heads of a stage must carry the same `step` (Adam steps every parameter together)."""
    layout, names = _layout(model)
    params = [p for _, p in model.named_parameters()]
    if len(ref_sd["param_groups"]) != 1 or len(ref_sd["param_groups"][0]["params"]) != len(layout):
        raise ValueError("expected one parameter group over the reference's %d parameters, got %s"
                         % (len(layout), [len(g["params"]) for g in ref_sd["param_groups"]]))
    ref_ids = ref_sd["param_groups"][0]["params"]
    state = {}
    for j, (key, pi, _, put) in enumerate(layout):
        st = ref_sd["state"].get(ref_ids[j])
        if st is None:
            continue
        dst = state.setdefault(pi, {})
        for k, v in st.items():
            if torch.is_tensor(v) and v.dim() > 0:
                if k not in dst:
                    dst[k] = torch.zeros_like(params[pi], dtype=v.dtype, device=v.device)
                put(dst[k], v)
            else:
                if k in dst and float(dst[k]) != float(v):
                    raise ValueError("%s: the heads of one stage carry different `%s` values" % (key, k))
                dst[k] = v.detach().clone() if torch.is_tensor(v) else v
    group = dict(ref_sd["param_groups"][0], params=list(range(len(names))))
    return {"state": state, "param_groups": [group]}


def n_reference_params(model):
    return len(_layout(model)[0])


def renumber_reference_checkpoint(model_sd, optim_ref_sd, idx_u, idx_i):
    """A checkpoint in the reference's layout with its embedding rows (and their Adam moments: optimizer entries 0 / 1) taken
    in another order: out_row[j] = row[idx[j]].  Saving a run that renumbered its nodes at ingest: idx = perm (data set id ->
    internal id) puts the rows back in the data set's order; loading: idx = the inverse."""
    msd = dict(model_sd)
    for k, idx in (("uEmbd.weight", idx_u), ("iEmbd.weight", idx_i)):
        msd[k] = msd[k][idx.to(msd[k].device)]
    osd = {"state": dict(optim_ref_sd["state"]), "param_groups": optim_ref_sd["param_groups"]}
    ids = optim_ref_sd["param_groups"][0]["params"]
    for j, idx in ((0, idx_u), (1, idx_i)):
        st = osd["state"].get(ids[j])
        if st is not None:
            osd["state"][ids[j]] = {k: (v[idx.to(v.device)] if torch.is_tensor(v) and v.dim() == 2 and v.shape[0] == idx.numel() else v)
                                    for k, v in st.items()}
    return msd, osd


def _inverse(perm):
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=perm.device)
    return inv


def save_checkpoint(path, model, optim, perm=None):
    """run_Gowalla.py:142-143 -- both halves in the reference's layout.  perm = (perm_u, perm_i) of a run that renumbered its
    nodes at ingest (Interactions(renumber="degree")): the file is written in the DATA SET's numbering all the same."""
    msd, osd = model.state_dict(), optim_state_to_reference(model, optim.state_dict())
    if perm is not None:
        msd, osd = renumber_reference_checkpoint(msd, osd, perm[0], perm[1])
    torch.save({"model": msd, "optim": osd}, path)


def load_checkpoint(path, model, optim, map_location=None, perm=None):
    """run_Gowalla.py:127-130: a checkpoint written by the reference or by save_checkpoint (or, for completeness, one in
    this package's own optimizer layout).  perm: as in save_checkpoint (the file speaks the data set's numbering)."""
    ck = torch.load(path, map_location=map_location)
    osd = ck["optim"]
    n = len(osd["param_groups"][0]["params"]) if len(osd["param_groups"]) == 1 else -1
    if perm is not None:
        if n != n_reference_params(model):
            raise ValueError("a run with renumbered nodes resumes from checkpoints in the reference's optimizer layout")
        ck = dict(ck)
        ck["model"], osd = renumber_reference_checkpoint(ck["model"], osd, _inverse(perm[0]), _inverse(perm[1]))
    model.load_state_dict(ck["model"])
    if n == n_reference_params(model) and n != len(list(model.parameters())):
        osd = optim_state_from_reference(model, osd)
    # the hyper-parameters of the running optimizer win where the implementations differ (fused / capturable flags)
    keep = {k: v for k, v in optim.state_dict()["param_groups"][0].items() if k in ("fused", "capturable", "foreach", "differentiable")}
    osd["param_groups"][0].update(keep)
    optim.load_state_dict(osd)
    return ck
