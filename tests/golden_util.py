"""Helpers shared by the CPU and GPU tests: rebuild reference-shaped inputs from a golden fixture."""
import numpy as np
import torch

NUM_DICT_KEYS = {"sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}


def sd_of(g):
    return {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}


def batch_of(g):
    out = {}
    for k, v in g.items():
        if k.startswith("in__"):
            t = torch.from_numpy(v)
            out[k[4:]] = t if v.size or k != "in__neg_item" else torch.empty(0)
    return out


def lap_list_of(g, device="cpu"):
    n_user, n_item = int(g["meta"][0]), int(g["meta"][1])
    N = n_user + n_item
    laps = []
    for s in (0, 1):
        idx = torch.from_numpy(np.stack([g[f"lap{s}_rows"], g[f"lap{s}_cols"]]))
        laps.append(torch.sparse_coo_tensor(idx, torch.from_numpy(g[f"lap{s}_vals"]), (N, N)).to(device))
    return laps


def ctor_args(g, lap_list, device):
    n_user, n_item, embed, ratio = int(g["meta"][0]), int(g["meta"][1]), int(g["meta"][2]), float(g["meta"][3])
    num_dict = dict(NUM_DICT_KEYS, user=n_user, item=n_item)
    return dict(embed_size=embed, layer_size=[int(x) for x in g["layers"]], node_dropout=float(g["meta"][5]),
                mess_dropout=[float(x) for x in g["mess"]], emb_ratio=ratio, lap_list=lap_list, num_dict=num_dict,
                batch_size=int(g["in__u_id"].shape[0]), device=device)


def layer_params(sd, n_layer):
    w1 = [sd[f"w1_list.{k}.weight"] for k in range(n_layer)]
    b1 = [sd[f"w1_list.{k}.bias"] for k in range(n_layer)]
    w2 = [sd[f"w2_list.{k}.weight"] for k in range(n_layer)]
    b2 = [sd[f"w2_list.{k}.bias"] for k in range(n_layer)]
    return w1, b1, w2, b2
