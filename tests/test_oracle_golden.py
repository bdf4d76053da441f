"""CPU tests: the oracle (oracle/ngcf_oracle.py + oracle/ngcf_oracle.c) against the golden vectors
captured from the imported reference (oracle/make_golden.py).  No GPU, no /root/reference."""
import ctypes as C

import numpy as np
import pytest
import torch

import ngcf_oracle as orc
from conftest import FWD_CASES, load_golden
from golden_util import batch_of, lap_list_of, layer_params, sd_of


def _feats(sd):
    return {"age": sd["age_emb.weight"], "sex": sd["sex_emb.weight"], "month": sd["month_emb.weight"],
            "day": sd["day_emb.weight"], "dow": sd["dow_emb.weight"]}


@pytest.mark.parametrize("name", FWD_CASES)
def test_torch_oracle_is_bit_exact_on_golden_forward(name):
    g = load_golden(name)
    if str(g["torch_version"]) != torch.__version__:
        pytest.skip("fixture captured with another torch build: bit-exactness is only promised for the same one")
    sd, b = sd_of(g), batch_of(g)
    n_user = int(g["meta"][0])
    user_w = sd["user_embedding.weight"].clone()
    orc.feature_inject_torch(user_w, _feats(sd), b["u_id"], b["age"], b["sex"], b["month"], b["day"], b["dow"],
                             float(g["meta"][3]))
    assert np.array_equal(user_w.numpy(), g["out_user_weight_after"])
    yi = orc.select_year_index(b["year"])
    assert yi == int(g["year_idx"])
    L = lap_list_of(g)[yi]
    w1, b1, w2, b2 = layer_params(sd, len(g["layers"]))
    all_E = orc.propagate_torch(L, user_w, sd["item_embedding.weight"], w1, b1, w2, b2)
    assert np.array_equal(all_E.numpy(), g["out_all_E"])
    u, p, n = orc.gather_torch(all_E, n_user, b["u_id"], b["pos_item"], b["neg_item"])
    assert np.array_equal(u.numpy(), g["out_u"]) and np.array_equal(p.numpy(), g["out_p"])
    assert n.numel() == g["out_n"].size and np.array_equal(n.numpy().reshape(g["out_n"].shape), g["out_n"])


def test_torch_oracle_train_mode_dropout_matches_reference_generator():
    g = load_golden("fwd_train_dropout")
    if str(g["torch_version"]) != torch.__version__:
        pytest.skip("fixture captured with another torch build")
    sd, b = sd_of(g), batch_of(g)
    user_w = sd["user_embedding.weight"].clone()
    orc.feature_inject_torch(user_w, _feats(sd), b["u_id"], b["age"], b["sex"], b["month"], b["day"], b["dow"], 1.0)
    L = lap_list_of(g)[int(g["year_idx"])]
    w1, b1, w2, b2 = layer_params(sd, 3)
    torch.set_rng_state(torch.from_numpy(g["rng_state"]))
    all_E = orc.propagate_torch(L, user_w, sd["item_embedding.weight"], w1, b1, w2, b2,
                                mess_dropout=[float(x) for x in g["mess"]], training=True,
                                node_dropout=float(g["meta"][5]), node_flag=True)
    assert np.array_equal(all_E.numpy(), g["out_all_E"])
    # kept-edge sets of the cumulative node dropout
    torch.set_rng_state(torch.from_numpy(g["rng_state"]))
    Lk = L
    for k in range(3):
        Lk = orc.sparse_dropout_torch(Lk, float(g["meta"][5]))
        assert np.array_equal(Lk._indices()[0].numpy(), g[f"kept_rows_{k}"])
        assert np.array_equal(Lk._indices()[1].numpy(), g[f"kept_cols_{k}"])
        assert np.array_equal(Lk._values().numpy(), g[f"kept_vals_{k}"])
        if k > 0:
            assert len(g[f"kept_rows_{k}"]) < len(g[f"kept_rows_{k-1}"])   # cumulative thinning


@pytest.mark.parametrize("name", FWD_CASES)
def test_f64_oracle_agrees_with_reference_to_stated_tolerance(name):
    g = load_golden(name)
    sd = sd_of(g)
    yi = int(g["year_idx"])
    E0 = np.concatenate([g["out_user_weight_after"], sd["item_embedding.weight"].numpy()])
    w1, b1, w2, b2 = layer_params(sd, len(g["layers"]))
    f64 = orc.propagate_f64(g[f"lap{yi}_rows"], g[f"lap{yi}_cols"], g[f"lap{yi}_vals"], E0,
                            [w.numpy() for w in w1], [b.numpy() for b in b1], [w.numpy() for w in w2],
                            [b.numpy() for b in b2])
    np.testing.assert_allclose(g["out_all_E"], f64, atol=2e-5, rtol=2e-3)   # SURVEY.md 8c stated tolerance
    assert np.abs(g["out_all_E"] - f64).max() < 5e-6


def test_c_oracle_spmm_is_bit_exact_with_torch_mm(oracle_clib):
    g = load_golden("fwd_sigC_demo")
    yi = int(g["year_idx"])
    rows, cols, vals = g[f"lap{yi}_rows"], g[f"lap{yi}_cols"], g[f"lap{yi}_vals"]
    N = int(g["meta"][0] + g["meta"][1])
    E = np.ascontiguousarray(g["out_all_E"][:, :65])
    L = lap_list_of(g)[yi]
    ref = torch.mm(L, torch.from_numpy(E)).numpy()
    out = np.empty_like(E)
    fn = oracle_clib.ngcf_oracle_spmm_coo_f32
    fn.restype = None
    fn(rows.ctypes.data_as(C.c_void_p), cols.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p),
       C.c_int64(len(vals)), C.c_int64(N), E.ctypes.data_as(C.c_void_p), C.c_int64(65), C.c_int(65),
       out.ctypes.data_as(C.c_void_p), C.c_int64(65))
    assert np.array_equal(out, ref)
    # CSR form, same order inside a row -> same bits
    rowptr = np.zeros(N + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    out2 = np.empty_like(E)
    col32 = cols.astype(np.int32)
    fn2 = oracle_clib.ngcf_oracle_spmm_csr_f32
    fn2.restype = None
    fn2(rowptr.ctypes.data_as(C.c_void_p), col32.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p),
        C.c_int64(N), E.ctypes.data_as(C.c_void_p), C.c_int64(65), C.c_int(65), out2.ctypes.data_as(C.c_void_p),
        C.c_int64(65))
    assert np.array_equal(out2, ref)


def test_c_oracle_dense_layer_matches_golden_block(oracle_clib):
    g = load_golden("fwd_sigA_small")
    sd = sd_of(g)
    yi = int(g["year_idx"])
    E = np.ascontiguousarray(np.concatenate([g["out_user_weight_after"], sd["item_embedding.weight"].numpy()]))
    L = lap_list_of(g)[yi]
    LE = np.ascontiguousarray(torch.mm(L, torch.from_numpy(E)).numpy())
    N, d = E.shape
    norm = np.empty((N, 65), np.float32)
    carry = np.empty((N, 65), np.float32)
    W1, B1 = sd["w1_list.0.weight"].numpy(), sd["w1_list.0.bias"].numpy()
    W2, B2 = sd["w2_list.0.weight"].numpy(), sd["w2_list.0.bias"].numpy()
    fn = oracle_clib.ngcf_oracle_layer_dense_f32
    fn.restype = None
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    fn(p(LE), C.c_int64(d), p(E), C.c_int64(d), C.c_int64(N), C.c_int(d), C.c_int(65), p(W1), p(B1), p(W2), p(B2),
       p(carry), C.c_int64(65), p(norm), C.c_int64(65))
    np.testing.assert_allclose(norm, g["out_all_E"][:, 65:130], atol=2e-6, rtol=2e-5)


def test_bpr_oracles_against_reference_outputs(oracle_clib):
    g = load_golden("bpr")
    fn = oracle_clib.ngcf_oracle_bpr_f32
    fn.restype = C.c_float
    for tag in ("full", "bcast", "one"):
        u, p, n = (np.ascontiguousarray(g[f"{tag}_{k}"]) for k in "upn")
        wd, bs = g[f"{tag}_wd_bs"]
        ref = float(g[f"{tag}_loss"])
        t = orc.bpr_torch(torch.from_numpy(u), torch.from_numpy(p), torch.from_numpy(n), float(wd), int(bs))
        if str(g["torch_version"]) == torch.__version__:
            assert float(t) == ref
        assert abs(orc.bpr_f64(u, p, n, float(wd), int(bs)) - ref) <= 1e-5 * abs(ref)
        c = fn(u.ctypes.data_as(C.c_void_p), C.c_int64(u.shape[0]), p.ctypes.data_as(C.c_void_p),
               C.c_int64(p.shape[0]), n.ctypes.data_as(C.c_void_p), C.c_int64(n.shape[0]), C.c_int(u.shape[1]),
               C.c_float(wd), C.c_float(bs))
        assert abs(c - ref) <= 1e-5 * abs(ref)


def test_laplacian_builder_oracle_matches_reference_matrix():
    g = load_golden("matrix")
    for tag in ("toy", "mid"):
        U, I = (int(x) for x in g[f"{tag}_dims"])
        laps = orc.build_laplacian_list(g[f"{tag}_in_year"], g[f"{tag}_in_userid"], g[f"{tag}_in_itemid"],
                                        g[f"{tag}_in_visitor"], U, I)
        for yi in (0, 1):
            r, c, v = laps[yi]
            assert np.array_equal(r, g[f"{tag}_lap{yi}_rows"]) and np.array_equal(c, g[f"{tag}_lap{yi}_cols"])
            assert np.array_equal(v, g[f"{tag}_lap{yi}_vals"])
        # the carry-over quirk (matrix.py:33,45): a year-18 edge that year 19 does not touch is still in slice 1
        yr, uu, ii, vv = (g[f"{tag}_in_{k}"] for k in ("year", "userid", "itemid", "visitor"))
        p19 = {(int(a), int(b)) for a, b in zip(uu[yr == 19], ii[yr == 19])}
        only18 = [(int(a), int(b)) for a, b, w in zip(uu[yr == 18], ii[yr == 18], vv[yr == 18])
                  if w != 0 and (int(a), int(b)) not in p19]
        assert only18
        edges1 = set(zip(laps[1][0].tolist(), laps[1][1].tolist()))
        assert all((a, U + b) in edges1 for a, b in only18)
