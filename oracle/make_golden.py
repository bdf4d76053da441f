"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

Run:  python oracle/make_golden.py            (needs /root/reference; CPU only)

The reference's Python never enters this repository: this script imports
`/root/reference/model/{NGCF,bprloss,matrix}.py`, drives them on small seeded inputs and
stores inputs + outputs as data.  While doing so it asserts that `oracle/ngcf_oracle.py`
(the torch restatement) is BIT-EXACT against the reference on every case - that assertion is
the oracle's pin; `tests/test_oracle_golden.py` re-checks the oracle against the stored
vectors wherever the reference is absent (the GPU box).

Fixtures are data only: tensors in, tensors out, plus the torch/numpy versions used.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference/model"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

from NGCF import NGCF as RefNGCF            # noqa: E402  (reference, imported not copied)
from bprloss import BPR as RefBPR          # noqa: E402
import ngcf_oracle as orc                   # noqa: E402


def make_laplacian(n_user, n_item, density, heavy_items, rng, carry=None):
    """Random weighted bipartite slice, count-degree normalised like matrix.py:55-62."""
    R = (rng.random((n_user, n_item)) < density)
    for h in heavy_items:
        R[:, h] |= rng.random(n_user) < 0.9
    W = np.where(R, rng.uniform(0.5, 5.0, R.shape), 0.0).astype(np.float32)
    if carry is not None:
        W = np.where(W != 0, W, carry)
    N = n_user + n_item
    A = np.zeros((N, N), dtype=np.float64)
    A[:n_user, n_user:] = W
    A[n_user:, :n_user] = W.T
    deg = np.count_nonzero(A, axis=1)
    with np.errstate(divide="ignore"):
        ds = np.power(deg.astype(np.float64), -0.5).astype(np.float32)
    ds[np.isinf(ds)] = 0
    Lap = (ds[:, None].astype(np.float64) * A) * ds[None, :].astype(np.float64)
    r, c = np.nonzero(Lap)
    v = Lap[r, c].astype(np.float32)
    return r.astype(np.int64), c.astype(np.int64), v, W


def to_sparse(r, c, v, N):
    return torch.sparse_coo_tensor(torch.from_numpy(np.stack([r, c])), torch.from_numpy(v), (N, N))


def state_to_np(sd):
    return {"sd__" + k: v.detach().numpy().copy() for k, v in sd.items()}


def run_forward_case(name, n_user, n_item, embed, layers, emb_ratio, years, B, seed, neg_empty,
                     dup_users, heavy, train=False, node_dropout=0.3, mess=(0.1, 0.1, 0.1)):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    N = n_user + n_item
    r0, c0, v0, W0 = make_laplacian(n_user, n_item, 0.3, heavy, rng)
    r1, c1, v1, _ = make_laplacian(n_user, n_item, 0.2, heavy, rng, carry=W0)
    lap = [to_sparse(r0, c0, v0, N), to_sparse(r1, c1, v1, N)]
    num_dict = {"user": n_user, "item": n_item, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
    model = RefNGCF(embed_size=embed, layer_size=list(layers), node_dropout=node_dropout,
                    mess_dropout=list(mess)[:len(layers)], emb_ratio=emb_ratio, lap_list=lap,
                    num_dict=num_dict, batch_size=B, device=torch.device("cpu"))
    model.train(train)
    sd_before = {k: v.clone() for k, v in model.state_dict().items()}

    u_id = torch.from_numpy(rng.integers(0, n_user, B))
    if dup_users and B >= 4:
        u_id[B // 2] = u_id[0]
        u_id[B - 1] = u_id[1]
    batch = dict(
        year=torch.from_numpy(np.asarray(years, dtype=np.int64)),
        u_id=u_id,
        age=torch.from_numpy(rng.integers(0, 76, B)), sex=torch.from_numpy(rng.integers(0, 2, B)),
        month=torch.from_numpy(rng.integers(0, 13, B)), day=torch.from_numpy(rng.integers(0, 32, B)),
        dow=torch.from_numpy(rng.integers(0, 7, B)),
        pos_item=torch.from_numpy(rng.integers(0, n_item, B)),
        neg_item=torch.empty(0) if neg_empty else torch.from_numpy(rng.integers(0, n_item, B)),
    )
    rng_state = torch.get_rng_state()
    with torch.no_grad():
        u, p, n = model(node_flag=train, **batch)
    all_E = torch.cat((model.all_users_emb, model.all_items_emb), 0)
    user_after = model.user_embedding.weight.detach().clone()

    # ---- pin the oracle: bit-exact against the reference ---------------------------------
    user_w = sd_before["user_embedding.weight"].clone()
    feats = {"age": sd_before["age_emb.weight"], "sex": sd_before["sex_emb.weight"],
             "month": sd_before["month_emb.weight"], "day": sd_before["day_emb.weight"],
             "dow": sd_before["dow_emb.weight"]}
    orc.feature_inject_torch(user_w, feats, batch["u_id"], batch["age"], batch["sex"], batch["month"],
                             batch["day"], batch["dow"], emb_ratio)
    assert torch.equal(user_w, user_after), name + ": feature injection not bit-exact"
    yi = orc.select_year_index(batch["year"])
    L = len(layers)
    w1 = [sd_before[f"w1_list.{k}.weight"] for k in range(L)]
    b1 = [sd_before[f"w1_list.{k}.bias"] for k in range(L)]
    w2 = [sd_before[f"w2_list.{k}.weight"] for k in range(L)]
    b2 = [sd_before[f"w2_list.{k}.bias"] for k in range(L)]
    torch.set_rng_state(rng_state)
    with torch.no_grad():
        o_all, carries = orc.propagate_torch(lap[yi], user_w, sd_before["item_embedding.weight"], w1, b1, w2, b2,
                                             mess_dropout=list(mess)[:L], training=train,
                                             node_dropout=node_dropout, node_flag=train, return_carry=True)
    assert torch.equal(o_all, all_E), name + ": propagation not bit-exact"
    ou, op, on = orc.gather_torch(o_all, n_user, batch["u_id"], batch["pos_item"], batch["neg_item"])
    assert torch.equal(ou, u) and torch.equal(op, p) and torch.equal(on, n), name + ": gathers"

    # kept-edge sets of the cumulative node dropout (train case only), for the "next" row
    extra = {}
    if train:
        torch.set_rng_state(rng_state)
        Lk = lap[yi]
        for k in range(L):
            Lk = orc.sparse_dropout_torch(Lk, node_dropout)
            extra[f"kept_rows_{k}"] = Lk._indices()[0].numpy().copy()
            extra[f"kept_cols_{k}"] = Lk._indices()[1].numpy().copy()
            extra[f"kept_vals_{k}"] = Lk._values().numpy().copy()
        for k, c in enumerate(carries):     # message-dropout masks are the zeros of the carries
            extra[f"carry_{k}"] = c.numpy().copy()

    f64 = None
    if not train:
        rr, cc, vv = (r0, c0, v0) if yi == 0 else (r1, c1, v1)
        E0 = torch.cat((user_w, sd_before["item_embedding.weight"]), 0).numpy()
        f64 = orc.propagate_f64(rr, cc, vv, E0, [w.numpy() for w in w1], [b.numpy() for b in b1],
                                [w.numpy() for w in w2], [b.numpy() for b in b2])
        err = np.abs(f64 - all_E.numpy().astype(np.float64)).max()
        print(f"  {name}: reference fp32 vs fp64 max-abs {err:.3e}")

    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        meta=np.asarray([n_user, n_item, embed, emb_ratio, int(train), node_dropout, seed], dtype=np.float64),
        layers=np.asarray(layers, dtype=np.int64), mess=np.asarray(mess[:L], dtype=np.float64),
        torch_version=np.asarray(torch.__version__), rng_state=rng_state.numpy(),
        lap0_rows=r0, lap0_cols=c0, lap0_vals=v0, lap1_rows=r1, lap1_cols=c1, lap1_vals=v1,
        **{"in__" + k: (v.numpy() if v.numel() else np.zeros((0,), np.float32)) for k, v in batch.items()},
        **state_to_np(sd_before),
        out_u=u.numpy(), out_p=p.numpy(), out_n=(n.numpy() if n.numel() else np.zeros((0,), np.float32)),
        out_all_E=all_E.numpy(), out_user_weight_after=user_after.numpy(),
        year_idx=np.asarray(yi), **extra)
    print(f"  wrote {name}: N={N} nnz={len(v0)}/{len(v1)} D={all_E.shape[1]} year_idx={yi}")


def run_bpr_cases():
    rng = np.random.default_rng(77)
    out = {}
    for tag, (B, Bp, Bn, D) in {"full": (37, 37, 37, 193), "bcast": (25, 1, 25, 260), "one": (1, 1, 1, 65)}.items():
        u = torch.from_numpy(rng.normal(0, 0.4, (B, D)).astype(np.float32)).requires_grad_()
        p = torch.from_numpy(rng.normal(0, 0.4, (Bp, D)).astype(np.float32)).requires_grad_()
        n = torch.from_numpy(rng.normal(0, 0.4, (Bn, D)).astype(np.float32)).requires_grad_()
        crit = RefBPR(weight_decay=0.025, batch_size=1024 if tag == "full" else 25)
        loss = crit(u, p, n)
        loss.backward()
        o = orc.bpr_torch(u.detach(), p.detach(), n.detach(), 0.025, crit.batch_size)
        assert torch.equal(o, loss.detach()), "bpr oracle not bit-exact: " + tag
        f64 = orc.bpr_f64(u.detach().numpy(), p.detach().numpy(), n.detach().numpy(), 0.025, crit.batch_size)
        assert abs(f64 - float(loss)) <= 1e-5 * abs(f64)
        out.update({f"{tag}_u": u.detach().numpy(), f"{tag}_p": p.detach().numpy(), f"{tag}_n": n.detach().numpy(),
                    f"{tag}_loss": loss.detach().numpy(), f"{tag}_gu": u.grad.numpy(), f"{tag}_gp": p.grad.numpy(),
                    f"{tag}_gn": n.grad.numpy(), f"{tag}_wd_bs": np.asarray([0.025, crit.batch_size])})
    np.savez_compressed(os.path.join(OUT, "bpr.npz"), torch_version=np.asarray(torch.__version__), **out)
    print("  wrote bpr")


def run_matrix_case():
    """Drive the reference `Matrix.create_matrix` (matrix.py:41-83) on toy frames."""
    import pandas as pd
    np.mat = np.asmatrix                       # harness-side alias; NumPy 2 removed np.mat (matrix.py:81)
    argv = sys.argv
    sys.argv = ["x"]                           # parsers.py:16 parses argv at import
    try:
        from matrix import Matrix as RefMatrix
    finally:
        sys.argv = argv
    rng = np.random.default_rng(5)
    out = {}
    for tag, (U, I, rows_per_year) in {"toy": (5, 2, 6), "mid": (200, 20, 900)}.items():
        frames = []
        for y in (18, 19):
            # unique (user, item) pairs per year, as in the real data (one row per user-day x destination)
            pair = rng.choice(U * I, size=min(rows_per_year, U * I), replace=False)
            uu, ii = pair // I, pair % I
            rows_per_year = len(pair)
            vis = rng.uniform(0.5, 5.0, rows_per_year).astype(np.float32)
            vis[rng.random(rows_per_year) < 0.25] = 0.0          # bottom-quartile -> 0, utils.py:117-121
            frames.append(pd.DataFrame({"year": y, "userid": uu, "itemid": ii, "visitor": vis}))
        df = pd.concat(frames, ignore_index=True)
        m = RefMatrix(total_df=df, cols=["year", "userid", "itemid", "visitor"], rating_col="visitor",
                      num_dict={"user": U, "item": I}, folder_path="/tmp", save_data=False,
                      device=torch.device("cpu"))
        laps = m.create_matrix()
        mine = orc.build_laplacian_list(df["year"].values, df["userid"].values, df["itemid"].values,
                                        df["visitor"].values, U, I)
        out[f"{tag}_dims"] = np.asarray([U, I])
        for k in ("year", "userid", "itemid", "visitor"):
            out[f"{tag}_in_{k}"] = df[k].values
        for yi, lap in enumerate(laps):
            idx, val = lap._indices().numpy(), lap._values().numpy()
            out[f"{tag}_lap{yi}_rows"], out[f"{tag}_lap{yi}_cols"], out[f"{tag}_lap{yi}_vals"] = idx[0], idx[1], val
            r, c, v = mine[yi]
            assert np.array_equal(r, idx[0]) and np.array_equal(c, idx[1]), f"matrix {tag}/{yi}: pattern"
            assert np.array_equal(v, val), f"matrix {tag}/{yi}: values not bit-exact"
    np.savez_compressed(os.path.join(OUT, "matrix.npz"), **out)
    print("  wrote matrix")


def main():
    os.makedirs(OUT, exist_ok=True)
    print("generating golden vectors with torch", torch.__version__, "numpy", np.__version__)
    # Sig-A shape (2 layers, 65->[65,65]), duplicates in u_id, both years in the batch -> year_idx 0
    run_forward_case("fwd_sigA_small", 50, 7, 65, (65, 65), 1.0, [19, 18, 18, 19, 18, 18, 19, 18], 8, 11,
                     neg_empty=False, dup_users=True, heavy=[2])
    # Sig-C shape (demo.py: 65->[64,64,64]), year=[0] length-1, empty neg, heavy item rows
    run_forward_case("fwd_sigC_demo", 200, 24, 65, (64, 64, 64), 1.0, [0], 12, 12,
                     neg_empty=True, dup_users=False, heavy=[0, 5])
    # year 19 -> slice 1, emb_ratio < 1, Sig-B shape
    run_forward_case("fwd_sigB_y19", 120, 16, 65, (65, 65, 65), 0.7, [19] * 10, 10, 13,
                     neg_empty=False, dup_users=True, heavy=[3])
    # wide rectangular first layer 130->[128]*3 (BASELINE d=128 recipe (i))
    run_forward_case("fwd_130_128", 96, 16, 130, (128, 128, 128), 1.0, [18] * 16, 16, 14,
                     neg_empty=False, dup_users=False, heavy=[1])
    # training-mode forward: cumulative node dropout + message dropout from the CPU generator
    run_forward_case("fwd_train_dropout", 90, 12, 65, (65, 65, 65), 1.0, [18] * 8, 8, 15,
                     neg_empty=False, dup_users=False, heavy=[4], train=True)
    run_bpr_cases()
    run_matrix_case()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"total fixture bytes: {tot}")


if __name__ == "__main__":
    main()
