"""The sparse Laplacian builder (mirror of model/matrix.py) against the reference's own output (golden fixture)."""
import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden


def _frame(g, tag):
    return pd.DataFrame({k: g[f"{tag}_in_{k}"] for k in ("year", "userid", "itemid", "visitor")})


@pytest.mark.parametrize("tag", ["toy", "mid"])
def test_matrix_builder_is_bit_exact_with_reference(tag):
    from seoul_tourism_recommendation_ngcf_amd.matrix import Matrix
    g = load_golden("matrix")
    U, I = (int(x) for x in g[f"{tag}_dims"])
    m = Matrix(total_df=_frame(g, tag), cols=["year", "userid", "itemid", "visitor"], rating_col="visitor",
               num_dict={"user": U, "item": I}, folder_path="", save_data=False, device=torch.device("cpu"))
    laps = m.create_matrix()
    assert len(laps) == 2
    for yi, L in enumerate(laps):
        assert L.is_sparse and tuple(L.shape) == (U + I, U + I) and L.dtype == torch.float32
        assert not L.is_coalesced()                                   # like torch.sparse.FloatTensor(...)
        idx, val = L._indices().numpy(), L._values().numpy()
        assert np.array_equal(idx[0], g[f"{tag}_lap{yi}_rows"]) and np.array_equal(idx[1], g[f"{tag}_lap{yi}_cols"])
        assert np.array_equal(val, g[f"{tag}_lap{yi}_vals"])


@pytest.mark.gpu
def test_matrix_builder_on_device_feeds_the_engine():
    import seoul_tourism_recommendation_ngcf_amd as pkg
    from seoul_tourism_recommendation_ngcf_amd.matrix import Matrix
    g = load_golden("matrix")
    U, I = (int(x) for x in g["mid_dims"])
    dev = torch.device("cuda:0")
    laps = Matrix(_frame(g, "mid"), ["year", "userid", "itemid", "visitor"], "visitor", {"user": U, "item": I},
                  device=dev).create_matrix()
    assert np.array_equal(laps[1]._values().cpu().numpy(), g["mid_lap1_vals"])
    num_dict = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    model = pkg.NGCF(65, [64, 64], 0.3, [0.1, 0.1], 1.0, laps, num_dict, 8, dev).to(dev).eval()
    with torch.no_grad():
        all_E = model.propagate(1)
    assert all_E.shape == (U + I, 65 + 128) and torch.isfinite(all_E).all()


def test_lap_list_writer_and_allow_list_reader_round_trip(tmp_path):
    """`Matrix(save_data=True)` writes what matrix.py:70-75 writes (a pickle of the list of sparse COO tensors); the reader of
    demo.py:22-27,63-67 is an allow-list unpickler.  Only a file this test wrote itself is ever opened."""
    import pickle
    from seoul_tourism_recommendation_ngcf_amd.matrix import LapListUnpickler, Matrix, load_lap_list
    g = load_golden("matrix")
    U, I = (int(x) for x in g["mid_dims"])
    m = Matrix(total_df=_frame(g, "mid"), cols=["year", "userid", "itemid", "visitor"], rating_col="visitor",
               num_dict={"user": U, "item": I}, folder_path=str(tmp_path), save_data=True, device=torch.device("cpu"))
    laps = m.create_matrix()
    assert m.saved_path and m.saved_path.startswith(str(tmp_path)) and m.saved_path.endswith(".pkl")
    with open(m.saved_path, "rb") as f:                                # the file is the plain pickle of the list
        head = f.read(2)
    assert head[:1] == b"\x80"
    back = load_lap_list(m.saved_path)
    assert len(back) == len(laps) == 2
    for a, b in zip(back, laps):
        assert a.is_sparse and a.device.type == "cpu" and tuple(a.shape) == tuple(b.shape)
        assert torch.equal(a._indices(), b._indices()) and torch.equal(a._values(), b._values())
    # anything but the sparse-tensor rebuild helpers is refused, whatever it is
    evil = tmp_path / "evil.pkl"
    evil.write_bytes(pickle.dumps([print]))
    with pytest.raises(pickle.UnpicklingError, match="refusing"):
        with open(evil, "rb") as f:
            LapListUnpickler(f).load()
    notlist = tmp_path / "dict.pkl"
    notlist.write_bytes(pickle.dumps({"a": 1}))
    with pytest.raises(pickle.UnpicklingError, match="not a lap_list"):
        load_lap_list(str(notlist))


@pytest.mark.gpu
def test_matrix_builder_at_c3_scale_on_device():
    """The builder on 50 M interactions (BASELINE configs[2] size) on the device: same pattern, order and values as the
    bench's graph generator (which normalises the same triplets independently), and a second year that carries the first
    year's edges over (matrix.py:33,45)."""
    import seoul_tourism_recommendation_ngcf_amd as pkg
    from seoul_tourism_recommendation_ngcf_amd.matrix import laplacian_slices
    dev = torch.device("cuda:0")
    U, I = 1_000_000, 100_000
    u, i, w = pkg.graphs.synthetic_interactions(U, I, 50_000_000, seed=2603, device=dev)
    coo = pkg.graphs._normalise(u, i, w, U, I)
    n = int(u.numel())
    half = n // 2
    year = torch.cat([torch.full((half,), 18, device=dev), torch.full((n - half,), 19, device=dev)])
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))     # unsorted input, years mixed
    k = int((perm < half).nonzero()[0])                          # a year-18 record first: years are taken in order of appearance
    perm[[0, k]] = perm[[k, 0]]
    sl = laplacian_slices(year[perm], u[perm], i[perm], w[perm], U, I, device=dev)
    assert sorted(sl) == [0, 1]
    rows, cols, vals = sl[1]                                     # year 19 holds every edge (carry-over) = the full graph
    assert torch.equal(rows, coo["rows"]) and torch.equal(cols, coo["cols"])
    np.testing.assert_allclose(vals.cpu().numpy(), coo["vals"].cpu().numpy(), rtol=1e-6, atol=0)   # d^-1/2: numpy float32 power (matrix.py:56) vs torch pow, 1 ulp apart on 0.006 % of the entries
    r0 = sl[0][0]
    assert int(r0.numel()) < int(rows.numel()) and bool((r0[1:] >= r0[:-1]).all())
