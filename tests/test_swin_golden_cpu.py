"""The oracle's Swin block (oracle/swinunetr.py::SwinBlock on oracle/window_attention.py) against golden vectors of an INDEPENDENT
implementation -- transformers' SwinLayer (tests/golden/make_golden_swin.py; the reference's own SwinUNETR comes from mfai / MONAI, absent
here): window partition order, cyclic-shift mask, relative-position index bit for bit; block output and input gradient <= 1e-6 in
float64.  This pins the part of the SwinUNETR oracle that the north star's K5 kernel (windowed attention) is checked against."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "swin_layer_*.npz")))


def load(path):
    z = np.load(path, allow_pickle=False)
    meta = eval(str(z["meta"]))
    return meta, z


def oracle_block(meta, z, dtype=torch.float64):
    from oracle.swinunetr import SwinBlock

    blk = SwinBlock(meta["dim"], meta["heads"], meta["window"], meta["shift"]).to(dtype)
    blk.load_state_dict({k[2:]: torch.from_numpy(z[k]).to(dtype) for k in z.files if k.startswith("w_")})
    return blk


def test_fixtures_present():
    assert len(FILES) == 5


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[11:-4] for p in FILES])
def test_window_bookkeeping_matches_transformers_bit_for_bit(path):
    from oracle import window_attention as owa

    meta, z = load(path)
    ws, H, W = meta["window"], meta["H"], meta["W"]
    Hp, Wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
    assert np.array_equal(owa.relative_position_index(ws).view(-1).numpy(), z["relative_position_index"].reshape(-1))
    idx = torch.arange(Hp * Wp, dtype=torch.float64).view(1, Hp, Wp, 1)
    assert np.array_equal(owa.window_partition(idx, ws).view(-1, ws * ws).long().numpy(), z["window_partition_index"])
    if meta["effective_shift"] > 0:
        assert np.array_equal(owa.shift_mask(Hp, Wp, ws, meta["effective_shift"], torch.float64).numpy(), z["attn_mask"])
    else:
        assert z["attn_mask"].shape[0] == 0


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[11:-4] for p in FILES])
def test_oracle_swin_block_reproduces_transformers_layer(path):
    meta, z = load(path)
    blk = oracle_block(meta, z)
    x = torch.from_numpy(z["x"]).double().requires_grad_(True)
    out = blk(x)
    out.backward(torch.from_numpy(z["gy"]).double())
    ref_out, ref_dx = torch.from_numpy(z["out"]), torch.from_numpy(z["dx"])
    assert float((out - ref_out).abs().max() / ref_out.abs().max()) <= 1e-6
    assert float((x.grad - ref_dx).abs().max() / ref_dx.abs().max()) <= 1e-6


MERGE_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "swin_merge_*.npz")))


def test_merge_fixtures_present():
    assert len(MERGE_FILES) == 3


@pytest.mark.parametrize("path", MERGE_FILES, ids=[os.path.basename(p)[11:-4] for p in MERGE_FILES])
def test_oracle_patch_merging_reproduces_transformers(path):
    """oracle/swinunetr.py::PatchMerging against transformers' SwinPatchMerging (tests/golden/make_golden_swin_merge.py): the 2 x 2
    concatenation order and the padding of odd grids bit for bit (an index map through the same slicing), output and input gradient
    <= 1e-6 in float64"""
    from oracle.swinunetr import PatchMerging

    meta, z = load(path)
    m = PatchMerging(meta["dim"]).double()
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith("w_")})
    H, W, C = meta["H"], meta["W"], meta["dim"]
    probe = PatchMerging(C).double()       # identity norm / reduction: the merged layout itself
    probe.norm, probe.reduction = torch.nn.Identity(), torch.nn.Identity()
    idx = torch.arange(H * W * C, dtype=torch.float64).view(1, H, W, C) + 1.0
    assert np.array_equal(probe(idx).long().numpy(), z["merge_index"])
    x = torch.from_numpy(z["x"]).double().requires_grad_(True)
    out = m(x)
    out.backward(torch.from_numpy(z["gy"]).double())
    ref_out, ref_dx = torch.from_numpy(z["out"]), torch.from_numpy(z["dx"])
    assert float((out - ref_out).abs().max() / ref_out.abs().max()) <= 1e-6
    assert float((x.grad - ref_dx).abs().max() / ref_dx.abs().max()) <= 1e-6
