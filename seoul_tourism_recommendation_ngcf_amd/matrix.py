"""`Matrix` - sparse, device-side builder of `lap_list` (mirror of model/matrix.py:12-83).

The reference builds every year slice through dense `N x N` float64 matrices (matrix.py:55-62), which is
impossible beyond a few thousand nodes.  This builder produces the same `lap_list` - a list of torch sparse COO
fp32 `[N, N]` tensors indexed by `year % 18` - from the triplets alone, and keeps the reference's quirks:

* `R` is never cleared between years (matrix.py:33,45): a later slice also holds the earlier years' edges, a
  re-assigned (user, item) takes the newer rating, an explicit 0 rating removes the edge;
* the degree is the COUNT of stored non-zeros per row (matrix.py:55), the values keep their weights;
* `d^-1/2` is float32 (matrix.py:56), the product `(d_i * a_ij) * d_j` float64, the result cast to float32;
* both triangles are stored, entries row-major sorted, tensor flagged un-coalesced (matrix.py:79-83).

Bit-exact against the reference on the golden fixtures (tests/test_matrix.py).  Same constructor arguments as the
reference class; `total_df` is a pandas frame with the columns named in `cols`.
"""
from __future__ import annotations

import numpy as np
import torch


def laplacian_slices(year, userid, itemid, rating, n_user: int, n_item: int, device="cpu"):
    """{year_idx: (rows, cols, vals)} from int64/float32 triplet tensors; see the module docstring."""
    year = torch.as_tensor(year, dtype=torch.int64, device=device)
    userid = torch.as_tensor(userid, dtype=torch.int64, device=device)
    itemid = torch.as_tensor(itemid, dtype=torch.int64, device=device)
    rating = torch.as_tensor(rating, dtype=torch.float32, device=device)
    N = n_user + n_item
    years = []
    for y in year.tolist():                     # pandas .unique(): order of first appearance
        if y not in years:
            years.append(y)
    # state of R: sorted unique keys (u * n_item + i) with their current value
    keys = torch.empty(0, dtype=torch.int64, device=device)
    vals = torch.empty(0, dtype=torch.float32, device=device)
    out = {}
    for y in years:
        sel = year == y
        k_new = userid[sel] * n_item + itemid[sel]
        v_new = rating[sel]
        # dok assignment: within one statement the last occurrence of a key wins; across years newer wins
        k_all = torch.cat([keys, k_new])
        v_all = torch.cat([vals, v_new])
        order = torch.sort(k_all, stable=True).indices
        k_s, v_s = k_all[order], v_all[order]
        last = torch.ones_like(k_s, dtype=torch.bool)
        last[:-1] = k_s[1:] != k_s[:-1]
        keys, vals = k_s[last], v_s[last]
        keep = vals != 0                                     # assigning 0 to a dok entry deletes it
        keys, vals = keys[keep], vals[keep]
        u, i = keys // n_item, keys % n_item
        deg = torch.bincount(torch.cat([u, i + n_user]), minlength=N)
        # the reference's exact call (matrix.py:56): numpy's float32 power is not correctly rounded, so the same
        # routine is used on this N-sized vector (host) to stay bit-identical
        with np.errstate(divide="ignore"):
            ds_np = np.power(deg.cpu().numpy().astype(np.float64)[:, None], -0.5, dtype=np.float32).squeeze(1)
        ds_np[np.isinf(ds_np)] = 0.0
        ds = torch.from_numpy(ds_np).to(device)
        w = vals.to(torch.float64)
        v_ui = (ds[u].double() * w) * ds[i + n_user].double()         # (D^-1/2 . A) . D^-1/2, float64 (matrix.py:62)
        v_iu = (ds[i + n_user].double() * w) * ds[u].double()
        it_order = torch.sort(i, stable=True).indices                 # item rows: sorted by (i, u)
        rows = torch.cat([u, (i + n_user)[it_order]])
        cols = torch.cat([i + n_user, u[it_order]])
        v = torch.cat([v_ui, v_iu[it_order]]).to(torch.float32)
        nz = v != 0
        out[int(y) % 18] = (rows[nz], cols[nz], v[nz])
    return out


class Matrix(torch.nn.Module):
    """Same surface as the reference's `Matrix` (matrix.py:12-76): `create_matrix()` -> `lap_list`."""

    def __init__(self, total_df, cols: list, rating_col: str, num_dict: dict, folder_path: str = "",
                 save_data: bool = False, device="cpu"):
        super().__init__()
        self.df = total_df[cols]
        self.rating_col = rating_col
        self.folder_path = folder_path
        self.save_data = save_data
        self.device = device
        self.n_user = num_dict['user']
        self.n_item = num_dict['item']
        self.lap_list = [[] for _ in self.df['year'].unique()]

    def create_matrix(self):
        N = self.n_user + self.n_item
        slices = laplacian_slices(self.df['year'].values, self.df['userid'].values, self.df['itemid'].values,
                                  self.df[self.rating_col].values, self.n_user, self.n_item, self.device)
        for yi, (r, c, v) in slices.items():
            self.lap_list[yi] = torch.sparse_coo_tensor(torch.stack([r, c]), v, (N, N))
        if self.save_data:
            raise NotImplementedError("pickling lap_list (matrix.py:70-75) is left to the caller: torch.save(lap_list, path)")
        return self.lap_list
