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
