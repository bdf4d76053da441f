"""CPU tests: the C-ABI library loads and exports every symbol include/ngcf_hip.h declares; the
nn.Module mirror keeps the reference's constructor/state_dict surface; and there is NO CPU fallback."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden
from golden_util import batch_of, ctor_args, lap_list_of, sd_of


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ngcf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ngcf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from seoul_tourism_recommendation_ngcf_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ngcf_hip.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.PROTOTYPES) == declared
    assert lib.ngcf_target_arch() == b"gfx950"
    assert os.path.dirname(_lib.lib_path()).endswith("seoul_tourism_recommendation_ngcf_amd")   # in-tree


def test_library_has_a_gfx950_code_object():
    from seoul_tourism_recommendation_ngcf_amd import _lib
    _lib.load()
    blob = open(_lib.lib_path(), "rb").read()
    assert b"gfx950" in blob and b"spmm_kernel" in blob and b"layer_dense_kernel" in blob


def test_host_only_entry_points():
    from seoul_tourism_recommendation_ngcf_amd import _lib, engine
    lib = _lib.load()
    assert lib.ngcf_dense_workspace_bytes(128, 128) > 2 * 128 * 128 * 4
    assert lib.ngcf_dense_workspace_bytes(128, 1024) == -1          # d_out > 512 unsupported
    assert lib.ngcf_bpr_workspace_bytes(1024) >= 1024 // 4 * 8
    # nnz-balanced cut: 4 light rows + 1 heavy row
    rowptr = torch.tensor([0, 1, 2, 3, 4, 104], dtype=torch.int64)
    assert engine.shard_plan(rowptr, 0, 5, 2) == [0, 4, 5]
    b = engine.shard_plan(torch.arange(0, 101, dtype=torch.int64), 0, 100, 4)
    assert b == [0, 25, 50, 75, 100]
    b = engine.shard_plan(torch.zeros(11, dtype=torch.int64), 0, 10, 3)     # empty graph: split rows evenly
    assert b[0] == 0 and b[-1] == 10 and all(x <= y for x, y in zip(b, b[1:]))
    # error path + message
    rc = lib.ngcf_shard_plan(None, 0, 1, 1, None)
    assert rc == _lib.ERR_ARG and b"shard_plan" in lib.ngcf_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)


@pytest.mark.parametrize("name", ["fwd_sigA_small", "fwd_sigC_demo", "fwd_sigB_y19", "fwd_130_128"])
def test_state_dict_surface_matches_reference_checkpoints(name):
    from seoul_tourism_recommendation_ngcf_amd import NGCF
    g = load_golden(name)
    sd = sd_of(g)
    model = NGCF(**ctor_args(g, lap_list_of(g), torch.device("cpu")))
    mine = model.state_dict()
    assert list(mine.keys()) == list(sd.keys())                       # same keys, same order
    assert [tuple(v.shape) for v in mine.values()] == [tuple(v.shape) for v in sd.values()]
    assert all(v.dtype == torch.float32 for v in mine.values())
    model.load_state_dict(sd, strict=True)
    assert torch.equal(model.user_embedding.weight, sd["user_embedding.weight"])
    assert len(list(model.parameters())) == len(sd)                    # Adam sees every tensor (main.py:74)
    assert model.n_layer == len(g["layers"]) and model.emb_size == int(g["meta"][2])
    # dropout modules exist but add no keys; eval()/train() toggles them (experiment.py:72)
    assert len(model.mess_dropout_list) == model.n_layer
    model.eval()
    assert not model.mess_dropout_list[0].training


def test_init_follows_reference_initialisers():
    from seoul_tourism_recommendation_ngcf_amd import NGCF
    g = load_golden("fwd_sigA_small")
    torch.manual_seed(0)
    model = NGCF(**ctor_args(g, lap_list_of(g), torch.device("cpu")))
    # kaiming_uniform_(a=0) on [n, d]: bound = sqrt(6 / d)   (NGCF.py:58-68)
    w = model.user_embedding.weight
    bound = (6.0 / w.shape[1]) ** 0.5
    assert float(w.abs().max()) <= bound and float(w.abs().max()) > 0.8 * bound
    fw = model.age_emb.weight
    assert fw.shape == (76, 13) and float(fw.abs().max()) <= (6.0 / 13) ** 0.5


@pytest.mark.parametrize("name", ["fwd_sigA_small", "fwd_sigC_demo", "fwd_130_128"])
def test_same_seed_gives_the_reference_initial_state_dict(name):
    """The fixtures hold the reference module's state_dict right after construction under torch.manual_seed(seed)
    (oracle/make_golden.py: run_forward_case); the mirror consumes the generator in the same order (NGCF.py:56-78:
    embeddings, then W1_k, W2_k per layer), so every tensor comes out bit-identical."""
    from seoul_tourism_recommendation_ngcf_amd import NGCF
    g = load_golden(name)
    torch.manual_seed(int(g["meta"][6]))
    model = NGCF(**ctor_args(g, lap_list_of(g), torch.device("cpu")))
    for k, v in sd_of(g).items():
        assert torch.equal(model.state_dict()[k], v), k


def test_no_cpu_fallback_forward_raises():
    from seoul_tourism_recommendation_ngcf_amd import BPR, NGCF
    g = load_golden("fwd_sigA_small")
    model = NGCF(**ctor_args(g, lap_list_of(g), torch.device("cpu")))
    model.load_state_dict(sd_of(g))
    with pytest.raises(RuntimeError, match="no CPU"):
        model(node_flag=False, **batch_of(g))
    with pytest.raises(RuntimeError, match="ROCm device"):
        BPR(0.025, 8)(torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(2, 4))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "seoul_tourism_recommendation_ngcf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "ngcf_oracle" not in text and "oracle/" not in text, f
