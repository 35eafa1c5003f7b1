"""Pins oracle/ngacf_oracle.py to the fixtures the reference itself produced
(oracle/gen_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, params_of
from oracle import ngacf_oracle as O


def _maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


@pytest.mark.parametrize("tag,concat", [("k8_concat", True), ("k64_noconcat", False)])
def test_g1_layer(tag, concat):
    g = load_golden("g1_layer_" + tag)
    U, I = int(g["U"]), int(g["I"])
    f = O.layer_forward(g["X"], g["W_u"], g["W_i"], g["a"], g["edge"], U, I, 0.2, concat)
    assert _maxabs(f["out"], g["out"]) < 2e-6
    assert _maxabs(f["R"], g["e_rowsum"]) < 1e-5
    assert _maxabs(f["C"], g["e_colsum"]) < 1e-5
    assert (g["e_colsum"][37:] == 0).all()          # the three isolated items
    assert np.array_equal(f["out"][U + 37:], f["i_h"][37:] if not concat else O.elu(f["i_h"][37:]))


@pytest.mark.parametrize("tag,multi", [("spuigacf", False), ("spuimultigacf", True)])
def test_g234_stack_scores_loss_grads(tag, multi):
    g = load_golden("g234_" + tag)
    U, I = int(g["U"]), int(g["I"])
    p32 = params_of(g)
    # fp32 forward vs reference fp32
    F = O.spuigat_forward(p32, O.features_of(p32), g["edge"], U, I, multi=multi)
    assert _maxabs(F, g["F_f32"]) < 5e-6
    # float64 run vs reference float64: the restatement is exact to rounding
    p64 = {k: v.astype(np.float64) for k, v in p32.items()}
    loss, pos, neg, grads = O.bpr_step_grads(p64, g["u"], g["pos"], g["neg"], g["edge"], multi=multi)
    assert _maxabs(pos, g["pos_f64"]) < 1e-12
    assert _maxabs(neg, g["neg_f64"]) < 1e-12
    assert abs(loss - float(g["loss_f64"])) < 1e-12
    for k, v in grads.items():
        ref = g["g_f64." + k]
        assert _maxabs(v, ref) <= 1e-11 * max(1.0, float(np.abs(ref).max())), k
    assert set(grads) == {k[len("g_f64."):] for k in g if k.startswith("g_f64.")}
    # fp32 grads of the reference sit within fp32 noise of the float64 truth
    for k in grads:
        scale = max(float(np.abs(g["g_f64." + k]).max()), 1e-12)
        assert _maxabs(g["g_f32." + k], g["g_f64." + k]) / scale < 1e-3


def test_g4b_fixed_dropout_forward_and_grads():
    g = load_golden("g4b_fixed_dropout")
    U, I = int(g["U"]), int(g["I"])
    p = {k: v.astype(np.float64) for k, v in params_of(g).items()}
    em = [[g["emasks"][h].astype(np.float64) for h in range(8)], [g["emasks"][8].astype(np.float64)]]
    fm = [g["fmasks"][0].astype(np.float64), g["fmasks"][1].astype(np.float64)]
    X = O.features_of(p)
    F, tape = O.spuigat_forward(p, X, g["edge"], U, I, feat_masks=fm, edge_masks=em, keep=True)
    sc = np.sum(F[g["u"]] * F[g["item"] + U], 1)
    assert _maxabs(sc, g["scores"]) < 1e-5
    gF = np.zeros_like(F)
    np.add.at(gF, g["u"], F[g["item"] + U])
    np.add.at(gF, g["item"] + U, F[g["u"]])
    dX, grads = O.spuigat_backward(gF, tape, p, g["edge"], U, I, feat_masks=fm, edge_masks=em)
    grads["uEmbd.weight"], grads["iEmbd.weight"] = dX[:U], dX[U:]
    for k, v in grads.items():
        ref = g["g." + k]
        assert _maxabs(v, ref) <= 2e-4 * max(float(np.abs(ref).max()), 1e-6), k     # reference ran in fp32


def test_g6_spga_layer():
    g = load_golden("g6_spga_layer")
    for tag, concat in (("k8", True), ("k64", False)):
        out = O.spga_layer_forward(g["X"], g["W_" + tag], g["a_" + tag], g["edge"], 0.2, concat)
        assert _maxabs(out, g["out_" + tag]) < 5e-6


def test_g7_static_spmm_and_mean_adj():
    g = load_golden("g7_static_spmm")
    U, I = int(g["U"]), int(g["I"])
    out = O.static_spmm(g["L_row"], g["L_col"], g["L_val"], g["X"])
    assert _maxabs(out, g["out"]) < 5e-6
    r, c, v = O.mean_adj(g["edge"], U, I)
    import scipy.sparse as sp
    mine = sp.coo_matrix((v, (r, c)), shape=(U + I, U + I)).toarray()
    ref = sp.coo_matrix((g["L_val"], (g["L_row"], g["L_col"])), shape=(U + I, U + I)).toarray()
    assert _maxabs(mine, ref) < 1e-6


def test_g8_metrics_and_ranking():
    g = load_golden("g8_metrics")
    Ks = (1, 5, 10, 20)
    for r, n, perf in zip(g["rvecs"], g["npos"], g["perf"]):
        p = O.performance(list(range(int(n))), list(r), Ks)
        for c, key in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
            assert np.allclose(p[key], perf[c], atol=1e-12), key
    # the inflated-NDCG quirk: a single hit at rank 20 still scores 1/log2(21)/1
    assert abs(O.ndcg_at_k([0] * 19 + [1], 20) - (1 / np.log2(21)) / 1.0) < 1e-12
    r = O.ranklist(list(g["user_pos"]), [int(x) for x in g["test_items"]], list(g["ratings"]), Ks)
    assert r == list(g["r"])


def test_g9_ml100k_adjacency_digest():
    g = load_golden("g9_g10_ml100k")
    d = load_golden("ml100k_interactions")
    u, i = d["userId"].astype(np.int64), d["itemId"].astype(np.int64)
    U, I = int(g["shape"][0]), int(g["shape"][1])
    assert (U, I) == (943, 1682) and int(g["nnz"]) == 100000
    key = np.unique(u * I + i)                                   # coalesce(): row-major sorted, unique
    idx = np.stack([key // I, key % I])
    assert idx.shape[1] == 100000
    assert hashlib.sha256(np.ascontiguousarray(idx).tobytes()).hexdigest() == str(g["sha256"])
    assert np.array_equal(idx[:, :16], g["first16"]) and np.array_equal(idx[:, -16:], g["last16"])


def test_g10_ml100k_medium_graph_eval_output():
    g = load_golden("g9_g10_ml100k")
    d = load_golden("ml100k_interactions")
    U, I = int(g["shape"][0]), int(g["shape"][1])
    key = np.unique(d["userId"].astype(np.int64) * I + d["itemId"].astype(np.int64))
    edge = np.stack([key // I, key % I])
    X = (np.random.RandomState(int(g["emb_seed"])).randn(U + I, 64) * 0.01).astype(np.float32)
    p = params_of(g)
    F = O.spuigat_forward(p, X, edge, U, I)
    assert _maxabs(F[g["rows"]], g["F_rows"]) < 1e-6
    assert abs(float(F.astype(np.float64).sum()) - float(g["F_sum"])) < 1e-3
    assert abs(float(np.abs(F.astype(np.float64)).sum()) - float(g["F_abs_sum"])) < 1e-3


@pytest.mark.parametrize("tag,multi", [("spuigacf", False), ("spuimultigacf", True)])
def test_cpu_baseline_matches_goldens(tag, multi):
    """bench.py's CPU leg (reference operator order, torch CPU) reproduces the reference output."""
    import torch
    from oracle import cpu_baseline as CB
    g = load_golden("g234_" + tag)
    U, I = int(g["U"]), int(g["I"])
    p = {k: torch.from_numpy(v) for k, v in params_of(g).items()}
    x = torch.cat([p["uEmbd.weight"], p["iEmbd.weight"]])
    out = CB.forward(p, x, torch.from_numpy(g["edge"]), U, I, O.stage_names(multi))
    assert _maxabs(out.numpy(), g["F_f32"]) < 5e-6


def test_g7b_norm_adj_matches_reference_builder():
    """adj_type 'norm_adj' (data/loadGowalla.py:196-222) as the reference's own buildLaplacianMat produced it:
    pins the oracle restatement and the host-side builder."""
    import scipy.sparse as sp
    import torch
    from ngacf_amd.spmm import norm_adj_coo
    g = load_golden("g7b_norm_adj")
    U, I, edge = int(g["U"]), int(g["I"]), g["edge"]
    N = U + I
    ref = sp.coo_matrix((g["val"], (g["row"], g["col"])), shape=(N, N)).toarray()
    r, c, v = O.norm_adj(edge, U, I)
    assert np.abs(sp.coo_matrix((v, (r, c)), shape=(N, N)).toarray() - ref).max() < 1e-12
    r, c, v = norm_adj_coo(torch.from_numpy(edge), U, I)
    assert np.abs(sp.coo_matrix((v.numpy().astype(np.float64), (r.numpy(), c.numpy())), shape=(N, N)).toarray() - ref).max() < 1e-6


def test_g7c_plain_adj_matches_reference_builder():
    """adj_type 'plain_adj' (data/loadGowalla.py:184-205) as the reference's own buildLaplacianMat produced it, with
    ratings as values: pins the oracle restatement and ngacf_amd.spmm.plain_adj_coo (through the plan builder's coalesce)."""
    import scipy.sparse as sp
    import torch
    from ngacf_amd.spmm import plain_adj_coo
    g = load_golden("g7c_plain_adj")
    U, I, edge, rating = int(g["U"]), int(g["I"]), g["edge"], g["rating"]
    N = U + I
    assert tuple(g["shape"]) == (N, N)
    ref = sp.coo_matrix((g["val"], (g["row"], g["col"])), shape=(N, N)).toarray()
    r, c, v = O.plain_adj(edge, U, I, rating)
    assert np.abs(sp.coo_matrix((v, (r, c)), shape=(N, N)).toarray() - ref).max() == 0
    r, c, v = plain_adj_coo(torch.from_numpy(edge), U, I, torch.from_numpy(rating))
    assert np.abs(sp.coo_matrix((v.numpy().astype(np.float64), (r.numpy(), c.numpy())), shape=(N, N)).toarray() - ref).max() == 0


def test_g14_dense_multi_head_attention_and_the_models_on_it():
    """N4: the numpy restatement of graphattention/modules.py (MultiHeadAttention with / without mask incl. a fully
    masked row, ATTLayer) and of GACFV1 / GACFMask against the reference's own outputs."""
    import scipy.sparse as sp
    g = load_golden("g14_dense_mha")
    q, kv, heads = g["q_in"].astype(np.float64), g["k_in"].astype(np.float64), int(g["heads"])
    p = {k[len("mha.p."):]: v.astype(np.float64) for k, v in g.items() if k.startswith("mha.p.")}
    for tag, mask in (("nomask", None), ("mask", g["mask"])):
        out = O.multi_head_attention(p, q, kv, kv, heads, mask)
        assert np.abs(out - g["mha.%s.out_f64" % tag]).max() < 1e-10
        assert np.abs(out - g["mha.%s.out_f32" % tag]).max() < 1e-4
    pa = {k[len("att.p."):]: v.astype(np.float64) for k, v in g.items() if k.startswith("att.p.")}
    assert np.abs(O.att_layer(pa, q, kv, heads) - g["att.out"]).max() < 1e-5
    U, I = int(g["U"]), int(g["I"])
    N = U + I
    L = sp.coo_matrix((g["L_val"].astype(np.float64), (g["L_row"], g["L_col"])), shape=(N, N)).tocsr()
    p1 = {k[len("v1.p."):]: v.astype(np.float64) for k, v in g.items() if k.startswith("v1.p.")}
    s1 = O.gacfv1_scores(p1, L, g["u"], g["item"], U, I)
    assert np.abs(s1 - g["v1_scores"]).max() < 1e-4 * max(np.abs(g["v1_scores"]).max(), 1.0)
    p2 = {k[len("mk.p."):]: v.astype(np.float64) for k, v in g.items() if k.startswith("mk.p.")}
    s2 = O.gacfmask_scores(p2, L, g["dense_adj"], g["u"], g["item"], U, I)
    assert np.abs(s2 - g["mk_scores"]).max() < 1e-4 * max(np.abs(g["mk_scores"]).max(), 1.0)


def test_g16_gacf_variants():
    """GACFV2 .. GACFV5 (GACFmodel2-5.py): the numpy restatement against the reference's scores (V3-V5 on g14's V1 parameters)."""
    import scipy.sparse as sp
    g14, g = load_golden("g14_dense_mha"), load_golden("g16_gacf_variants")
    U, I = int(g14["U"]), int(g14["I"])
    N = U + I
    L = sp.coo_matrix((g14["L_val"].astype(np.float64), (g14["L_row"], g14["L_col"])), shape=(N, N)).tocsr()
    p1 = {k[len("v1.p."):]: v.astype(np.float64) for k, v in g14.items() if k.startswith("v1.p.")}
    for v in (3, 4, 5):
        got = O.gacfv1_scores(p1, L, g14["u"], g14["item"], U, I, variant=v)
        want = g["v%d_scores" % v]
        assert np.abs(got - want).max() < 1e-4 * max(np.abs(want).max(), 1.0), v
    p2 = {k[len("v2.p."):]: v.astype(np.float64) for k, v in g.items() if k.startswith("v2.p.")}
    got = O.gacfv2_scores(p2, L, g14["u"], g14["item"], U, I)
    assert np.abs(got - g["v2_scores"]).max() < 1e-4 * max(np.abs(g["v2_scores"]).max(), 1.0)


def test_g15_spga_model_wrappers():
    """SPGACF / MultiLayerSPGA / SPGAMGP (SPGA.py:85-316): the numpy restatement against the reference's scores."""
    import scipy.sparse as sp
    g = load_golden("g15_spga_models")
    U, I = int(g["U"]), int(g["I"])
    N = U + I
    L = sp.coo_matrix((g["L_val"].astype(np.float64), (g["L_row"], g["L_col"])), shape=(N, N)).tocsr()
    edge = np.stack(np.nonzero(g["dense_adj"]))
    for kind in ("spgacf", "multi", "mgp"):
        p = {k[len(kind) + 3:]: v.astype(np.float64) for k, v in g.items() if k.startswith(kind + ".p.")}
        got = O.spga_model_scores(kind, p, L, edge, g["u"], g["item"], U, I)
        want = g[kind + ".scores"]
        assert np.abs(got - want).max() < 1e-4 * max(np.abs(want).max(), 1.0), kind


def _g11_factors(U, I, seed=7):
    """The rank-64 score tables of oracle/gen_golden.py:g11_factors (data: seeded draws), S = A @ B^T."""
    rs = np.random.RandomState(seed)
    cont = (rs.randn(U, 64).astype(np.float32).astype(np.float64), rs.randn(I, 64).astype(np.float32).astype(np.float64))
    ties = (rs.randint(-2, 3, (U, 64)).astype(np.float64), rs.randint(-2, 3, (I, 64)).astype(np.float64))
    return {"cont": cont, "ties": ties}


def g11_split():
    """(u, i of ml100k in file order, train rows, test rows, fixture) of the reference's seeded train_test_split."""
    from ngacf_amd import data as D
    g = load_golden("g11_eval_neg_all")
    u, i, _ = D.load_ml100k()
    return u, i, g["train_rows"].astype(np.int64), g["test_rows"].astype(np.int64), g


@pytest.mark.parametrize("tag", ["cont", "ties"])
def test_g11_eval_all_items_matches_the_references_eval_neg_all(tag):
    """A8 pin (ii): oracle.eval_all_items against the reference's own eval_neg_all (train_eval_Gowalla.py:274-354) on a
    fixed score table and the reference's seeded ml100k split -- per-K precision / recall / ndcg / hit_ratio, with the
    normaliser len(train_pos_neg) and (table 'ties') the tie order of heapq.nlargest over the reference's sets."""
    u, i, tr, te, g = g11_split()
    U, I = int(g["U"]), int(g["I"])
    A, B = _g11_factors(U, I)[tag]
    S = (A @ B.T).astype(np.float32)
    train_pos, test_pos = {}, {}
    for a, b in zip(u[tr], i[tr]):
        train_pos.setdefault(int(a), set()).add(int(b))
    for a, b in zip(u[te], i[te]):
        test_pos.setdefault(int(a), set()).add(int(b))
    assert len(train_pos) == int(g["n_train_users"]) and len(test_pos) == int(g["n_test_users"])
    users = [int(x) for x in g["test_users_order"] if int(x) in train_pos]       # the merge at :307
    pool = set(int(x) for x in np.unique(i))
    got = O.eval_all_items(S[users], users, train_pos, test_pos, pool, float(len(train_pos)))
    for k in ("precision", "recall", "ndcg", "hit_ratio"):
        assert np.allclose(got[k], g["%s.%s" % (tag, k)], rtol=0, atol=1e-12), (k, got[k], g["%s.%s" % (tag, k)])
    assert float(g[tag + ".auc"]) == 0.0


def test_g17_gacfv6_and_gacfv3x():
    """GACFV6 / GACFV3x (GACFmodel6.py, GACFmodel3.py:114-200): the numpy restatement -- incl. the Laplacian GACFV6 builds from
    the rating table itself -- against the reference's scores and its Laplacian."""
    import scipy.sparse as sp
    g14, g = load_golden("g14_dense_mha"), load_golden("g17_gacf_mlp_heads")
    U, I = int(g14["U"]), int(g14["I"])
    N = U + I
    e = g14["edge"]
    L6 = O.rating_laplacian(e[0], e[1], g["rating"], U, I)
    ref6 = sp.coo_matrix((g["v6_L_val"].astype(np.float64), (g["v6_L_row"], g["v6_L_col"])), shape=(N, N)).tocsr()
    assert abs(L6 - ref6).max() < 1e-6
    L = sp.coo_matrix((g14["L_val"].astype(np.float64), (g14["L_row"], g14["L_col"])), shape=(N, N)).tocsr()
    for kind, Lk in (("v6", L6), ("v3x", L)):
        p = {k[len(kind) + 3:]: v.astype(np.float64) for k, v in g.items() if k.startswith(kind + ".p.")}
        got = O.gacf_mlp_scores(p, Lk, g14["u"], g14["item"], U, I, kind)
        ref = g[kind + "_scores"]
        assert np.abs(got - ref).max() < 2e-5 * max(np.abs(ref).max(), 1.0), (kind, np.abs(got - ref).max())


def g18_split():
    """(u, i, t of ml100k in file order, train rows, test rows, fixture) of the reference's own split_loo run."""
    from ngacf_amd import data as D
    g = load_golden("g18_neg_sampling_modes")
    u, i, _, t = D.load_ml100k(with_time=True)
    return u, i, t, g["train_rows"].astype(np.int64), g["test_rows"].astype(np.int64), g


def g18_table(g):
    A, B = _g11_factors(int(g["U"]), int(g["I"]))["cont"]
    return A, B, (float(g["table_scale"]) * (A @ B.T)).astype(np.float32)


def test_g18_neg_sampling_and_sampled_eval_match_the_references_own_loops():
    """--train_mode NegSampling / --eval_mode SampledNeg (run_Gowalla.py:163-183): the leave-one-out split (data/loadGowalla.py:
    307-313), the epoch loss of train_neg_sample (train_eval_Gowalla.py:36-88) and HR / NDCG@10 of eval_neg_sample (:193-259) on a
    fixed score table and the negatives the reference's run drew -- against the numbers that run printed (fixture g18)."""
    from ngacf_amd.train_eval import split_loo
    u, i, t, tr, te, g = g18_split()
    got_tr, got_te = split_loo(u, i, t)
    assert np.array_equal(got_tr, tr) and np.array_equal(got_te, te)           # the same rows, in file order
    _, _, S = g18_table(g)
    assert g["train_negs"].shape == (len(tr), 4) and g["test_negs"].shape == (len(te), 99)
    loss = O.neg_sample_epoch_loss(S, u[tr], i[tr], g["train_negs"], int(g["batch_size"]))
    assert abs(loss - float(g["train_loss"])) < 1e-9 * abs(float(g["train_loss"])) + 1e-12, (loss, float(g["train_loss"]))
    hr, nd = O.sampled_hr_ndcg(S, u[te], i[te], g["test_negs"], 10)
    assert abs(hr - float(g["HR"])) < 1e-12 and abs(nd - float(g["NDCG"])) < 1e-12, (hr, nd)
