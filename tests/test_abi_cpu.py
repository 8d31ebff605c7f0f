"""CPU checks: the C-ABI library loads and exports every symbol include/py4cast_hip.h declares;
host-side mirrors of the reference API (registry, NamedTensor, loss construction errors)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "py4cast_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(p4c_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from py4cast_amd import _lib

    handle = _lib.lib()  # raises if not built
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/py4cast_hip.h but not exported"
    # every declared symbol also has a ctypes signature (no untyped calls)
    assert set(names) == set(_lib.all_symbols())
    assert handle.p4c_version() == 100


def test_tall_skinny_dispatch_rules_of_the_library():
    """Host-side rules of the EPA kernels (no GPU call): which (dtype, width) pairs the matrix-core kernels take -- what
    py4cast_amd/ops_ts.py asks before it chunks a product into 64-column launches -- and the token splits of ``gram``."""
    from py4cast_amd import _lib

    h, F32, BF16 = _lib.lib(), _lib.F32, _lib.BF16
    assert h.p4c_ts_apply_wide_ok(BF16, BF16, 32, 64) == 1 and h.p4c_ts_apply_wide_ok(BF16, F32, 256, 256) == 1
    assert h.p4c_ts_apply_wide_ok(F32, F32, 32, 64) == 0          # the fp32 flavour keeps the exact VALU kernels
    assert h.p4c_ts_apply_wide_ok(BF16, BF16, 12, 36) == 0        # widths off the 8-column granularity
    assert h.p4c_ts_apply_wide_ok(BF16, BF16, 264, 8) == 0        # reduction wider than the 16 chunks of 16 a lane holds
    assert h.p4c_ts_gram_wide_ok(BF16, BF16, 256, 256) == 1 and h.p4c_ts_gram_wide_ok(BF16, F32, 32, 32) == 0
    assert h.p4c_ts_gram_wide_ok(BF16, BF16, 24, 40) == 1 and h.p4c_ts_gram_wide_ok(BF16, BF16, 20, 40) == 0
    # 256 tokens per split from 4 096 tokens, 64 below; never more than 256 splits
    assert [h.p4c_ts_gram_splits(n) for n in (31, 256, 1024, 4096, 16384, 1 << 20)] == [1, 4, 16, 16, 64, 256]


def test_no_cpu_fallback():
    from py4cast_amd import _lib, ops

    x = torch.zeros(1, 1, 4, 4, 2)
    with pytest.raises(_lib.P4CError):
        ops.build_x(x, torch.zeros(1, 4, 4, 1), torch.zeros(1, 4, 4, 1))
    # the widened model ops: no CPU path either (a CPU tensor raises before any arithmetic)
    from py4cast_amd import ops_graph, ops_mlp, ops_rows
    from py4cast_amd.ops_attention import window_attention

    es = ops_graph.EdgeSet(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long), 2, 2)
    rows = torch.zeros(4, 64)
    for call in (lambda: ops_graph.edge_gather_add(rows, None, None, es),
                 lambda: ops_graph.aggregate_sum(rows, es),
                 lambda: ops_rows.row_layer_norm(rows, torch.ones(64), torch.zeros(64)),
                 lambda: ops_rows.row_linear(rows, torch.zeros(64, 64)),
                 lambda: ops_mlp.row_mlp(rows.bfloat16(), torch.zeros(64, 64), None, torch.zeros(64, 64), None),
                 lambda: window_attention(torch.zeros(1, 7, 7, 24), None, 1, 7)):
        with pytest.raises(_lib.P4CError):
            call()


def test_namedtensor_api():
    from py4cast_amd.namedtensor import NamedTensor

    t = NamedTensor(torch.arange(2 * 3 * 4 * 5 * 6.0).reshape(2, 3, 4, 5, 6), ["batch", "timestep", "lat", "lon", "features"], list("abcdef"))
    assert t.spatial_dim_idx == [2, 3] and t.num_spatial_dims == 2
    assert t.dim_size("timestep") == 3 and t.dim_index("lon") == 3
    s = t.select_dim("timestep", 1)
    assert s.names == ["batch", "lat", "lon", "features"] and s.tensor.shape == (2, 4, 5, 6)
    assert t.index_select_tensor_dim("timestep", range(1, 3)).shape == (2, 2, 4, 5, 6)
    t2 = t.clone()
    t2.flatten_("ngrid", 2, 3)
    assert t2.names == ["batch", "timestep", "ngrid", "features"] and t2.spatial_dim_idx == [2]
    with pytest.raises(ValueError):
        NamedTensor(torch.zeros(2, 3), ["a", "features"], ["x"])


def test_registry_contract():
    from py4cast_amd import models
    from py4cast_amd.base import ModelABC

    assert "HalfUNet" in models.registry  # the MI355X plugin was discovered through its module-name prefix
    # the reference's registry keys this build provides natively (tests/test_models.py:145-165 of the reference)
    assert {"HalfUNet", "SwinUNetR", "GraphLAM", "HiLAM", "HiLAMParallel", "Identity"} <= set(models.registry)
    for name, kls in models.registry.items():
        assert issubclass(kls, ModelABC) and kls.register
    with pytest.raises(KeyError):
        models.get_model_kls_and_settings("NoSuchModel", {})


def test_loss_construction_errors():
    from py4cast_amd.losses import CombinedLoss, WeightedLoss

    with pytest.raises(NameError):  # losses.py:25-31
        WeightedLoss("NoSuchLoss")
    with pytest.raises(KeyError):  # losses.py:271 globals()[...]
        CombinedLoss([{"class": "Nope", "params": {}}])
    c = CombinedLoss([{"class": "WeightedLoss", "weight": 2.0, "params": {"loss": "L1Loss", "reduction": "none"}}])
    assert c.losses[0][1] == 2.0


def test_strategy_validation():
    from helpers import make_dataset_info, register_test_models, synthetic_case
    from py4cast_amd.lightning import AutoRegressiveLightning

    register_test_models()
    case = synthetic_case(H=8, W=8, F=3)
    info = make_dataset_info(case, Ff=5)
    with pytest.raises(AttributeError):  # lightning.py:218-222
        AutoRegressiveLightning({}, info, None, model_name="TinyConvModel", training_strategy="nope")
    with pytest.raises(AttributeError):  # lightning.py:213-217
        AutoRegressiveLightning({}, info, None, model_name="TinyConvModel", num_input_steps=2, num_inter_steps=2)
    lm = AutoRegressiveLightning({}, info, None, model_name="TinyConvModel", training_strategy="diff_ar", num_inter_steps=2)
    with pytest.raises(ValueError):  # lightning.py:688-692
        lm._strategy_params()
    assert lm.model.in_channels == 3 + 4 + 5  # lightning.py:256-261
    assert lm.grid_static_features.shape == (2, 8, 8, 4) and lm.interior_mask.shape == (8, 8, 1)
    assert hasattr(lm, "interior_mask_s")  # registered by WeightedLoss.prepare (losses.py:65-71)


def test_product_package_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure and the reference never travels: no file of the product package (nor the plugin,
    nor bin/) may import / open either."""
    import glob
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "py4cast_amd", "*.py")) + glob.glob(os.path.join(root, "py4cast_amd", "csrc", "*")) + \
        [os.path.join(root, "py4cast_plugin_mi355x.py")] + glob.glob(os.path.join(root, "bin", "*.py"))
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|/root/reference", re.M)
    bad = [f for f in files if os.path.isfile(f) and not f.endswith((".o", ".so")) and pat.search(open(f, errors="ignore").read())]
    assert not bad, bad


def test_mask_tensor_matches_reference_loop():
    """lightning.py:769-785: the vectorised block masking draws the same permutation from the CPU generator and clears exactly the
    pixels the reference's Python loop clears (bit-exact index op)."""
    from oracle import rollout as orollout
    from py4cast_amd.lightning import AutoRegressiveLightning

    class Holder:
        mask_ratio = 0.6

    for (H, W) in ((16, 16), (20, 37), (9, 64)):
        x = torch.randn(2, H, W, 3)
        torch.manual_seed(5)
        got = AutoRegressiveLightning.mask_tensor(Holder(), x)
        torch.manual_seed(5)
        idx = torch.randperm(H * W)[: int((1 - Holder.mask_ratio) * H * W)]
        ref = orollout.mask_tensor(x, Holder.mask_ratio, idx)
        assert torch.equal(got, ref), (H, W)
