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

import io
import os
import pickle
from datetime import datetime

import numpy as np
import torch


def laplacian_slices(year, userid, itemid, rating, n_user: int, n_item: int, device="cpu"):
    """{year_idx: (rows, cols, vals)} from int64/float32 triplet tensors; see the module docstring."""
    year = torch.as_tensor(year, dtype=torch.int64, device=device)
    userid = torch.as_tensor(userid, dtype=torch.int64, device=device)
    itemid = torch.as_tensor(itemid, dtype=torch.int64, device=device)
    rating = torch.as_tensor(rating, dtype=torch.float32, device=device)
    N = n_user + n_item
    # pandas .unique(): order of first appearance
    uniq, inv = torch.unique(year, return_inverse=True)
    first = torch.full((uniq.numel(),), year.numel(), dtype=torch.int64, device=year.device)
    first.scatter_reduce_(0, inv, torch.arange(year.numel(), device=year.device), reduce="amin")
    years = [int(uniq[k]) for k in torch.argsort(first).tolist()]
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
        self.file_tag = "implicit"        # file name tag of save_data (the reference formats its CLI arguments in here)
        self.saved_path = None

    def create_matrix(self):
        N = self.n_user + self.n_item
        slices = laplacian_slices(self.df['year'].values, self.df['userid'].values, self.df['itemid'].values,
                                  self.df[self.rating_col].values, self.n_user, self.n_item, self.device)
        for yi, (r, c, v) in slices.items():
            self.lap_list[yi] = torch.sparse_coo_tensor(torch.stack([r, c]), v, (N, N))
        if self.save_data:                                   # matrix.py:70-75
            self.saved_path = save_lap_list(self.lap_list, self.folder_path, self.file_tag)
        return self.lap_list


# ------------------------------------------------------------------------------------------------
# lap_list.pkl: the writer of matrix.py:70-75 and the reader of demo.py:22-27,63-67
# ------------------------------------------------------------------------------------------------
def save_lap_list(lap_list, folder_path: str, tag: str = "implicit") -> str:
    """What `Matrix.create_matrix(save_data=True)` writes (matrix.py:70-75): `pickle.dump(lap_list, f)` - a plain pickle of
    the list of torch sparse COO tensors, wherever they live - into `lap_list_<tag>_<month>_<day>_<hour>_<minute>.pkl` under
    `folder_path`.  (The reference puts its argparse hyper-parameters into `<tag>`; the CLI is out of scope here, so the
    caller chooses the tag.)  Returns the path."""
    d1 = datetime.now()
    path = os.path.join(folder_path, f"lap_list_{tag}_{d1.month}_{d1.day}_{d1.hour}_{d1.minute}.pkl")
    with open(path, "wb") as f:
        pickle.dump(list(lap_list), f)
    return path


class LapListUnpickler(pickle.Unpickler):
    """The reader of demo.py:22-27 (`CPU_Unpickler`) with an allow-list: a `lap_list` pickle only ever needs the handful of
    torch rebuild helpers a sparse COO tensor reduces to.  Tensor storages arrive as `torch.storage._load_from_bytes(b)`;
    like the reference's hook they are routed through `torch.load(..., map_location='cpu')` - here with `weights_only=True`,
    so nothing in the inner blob is executed either - which is what lets a file written on a GPU machine open anywhere.
    Any other global in the stream raises `pickle.UnpicklingError`."""

    _ALLOWED = {
        ("torch._utils", "_rebuild_sparse_tensor"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
        ("torch.serialization", "_get_layout"), ("torch", "Size"), ("collections", "OrderedDict"),
    }

    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return lambda b: torch.load(io.BytesIO(b), map_location="cpu", weights_only=True)
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"lap_list pickle refers to {module}.{name}, which is not needed to rebuild sparse "
                                     "tensors: refusing to load it")


def load_lap_list(path: str, device="cpu"):
    """`lap_list` from a file written by `save_lap_list` / matrix.py:70-75, on the CPU (demo.py:63-67) or moved to `device`.
    Only files this process's user trusts enough to parse: the allow-list bounds what the stream can name, it does not
    make a hostile pickle harmless in general."""
    with open(path, "rb") as f:
        laps = LapListUnpickler(f).load()
    if not isinstance(laps, list) or not all(isinstance(t, torch.Tensor) and t.is_sparse for t in laps):
        raise pickle.UnpicklingError("not a lap_list: expected a list of torch sparse COO tensors")
    return [t.to(device) for t in laps]
