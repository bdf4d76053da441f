"""The 23 checkpoints shipped with the reference load into the mirror module with strict=True.
Only runs where /root/reference exists (the build container); weights_only=True, nothing is unpickled."""
import glob
import os

import pytest
import torch

CKPTS = sorted(glob.glob("/root/reference/model/saved_model_data/*.pth") + glob.glob("/root/reference/model/saved_data_layer2/*.pth"))


@pytest.mark.skipif(not CKPTS, reason="reference checkout not present (GPU box)")
def test_all_reference_checkpoints_load_strict():
    from seoul_tourism_recommendation_ngcf_amd import NGCF
    assert len(CKPTS) == 23
    num_dict = {"user": 5840, "item": 100, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}   # num_dict.pkl
    sigs = set()
    for path in CKPTS:
        sd = torch.load(path, weights_only=True, map_location="cpu")
        n_layer = sum(1 for k in sd if k.startswith("w1_list.") and k.endswith(".weight"))
        layers = [int(sd[f"w1_list.{k}.weight"].shape[0]) for k in range(n_layer)]
        embed = int(sd["user_embedding.weight"].shape[1])
        model = NGCF(embed, layers, 0.3, [0.1] * n_layer, 1.0, [], num_dict, 512, torch.device("cpu"))
        assert list(model.state_dict().keys()) == list(sd.keys()), os.path.basename(path)
        model.load_state_dict(sd, strict=True)
        assert torch.equal(model.w2_list[n_layer - 1].bias, sd[f"w2_list.{n_layer - 1}.bias"])
        sigs.add((n_layer, tuple(layers)))
    assert sigs == {(2, (65, 65)), (3, (65, 65, 65)), (3, (64, 64, 64))}          # Sig-A, Sig-B, Sig-C
