"""The HIP Swin block (py4cast_amd/swinunetr.py::SwinBlock: row LayerNorm, row GEMMs, windowed attention on MFMA with LDS-staged Q/K/V
tiles) against the golden vectors of transformers' SwinLayer (tests/golden/make_golden_swin.py): fp32 flavour <= 1e-4 forward (the
north star's bar) and <= 1e-3 for the input gradient; bf16 flavour at the bf16 bar."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "swin_layer_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[11:-4] for p in FILES])
@pytest.mark.parametrize("dtype,fwd_tol,bwd_tol", [(torch.float32, 1e-4, 1e-3), (torch.bfloat16, 3e-2, 6e-2)])
def test_swin_block_matches_transformers_golden(gpu_device, path, dtype, fwd_tol, bwd_tol):
    from py4cast_amd.swinunetr import SwinBlock

    z = np.load(path, allow_pickle=False)
    meta = eval(str(z["meta"]))
    blk = SwinBlock(meta["dim"], meta["heads"], meta["window"], meta["shift"]).to(gpu_device)
    missing, unexpected = blk.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith("w_")}, strict=False)
    assert not unexpected and all("relative_position_index" in m or "_rpi" in m for m in missing), (missing, unexpected)
    x = torch.from_numpy(z["x"]).to(gpu_device).to(dtype).requires_grad_(True)
    out = blk(x)
    out.backward(torch.from_numpy(z["gy"]).to(gpu_device).to(dtype))
    ref_out, ref_dx = torch.from_numpy(z["out"]), torch.from_numpy(z["dx"])
    err = float((out.detach().double().cpu() - ref_out).norm() / ref_out.norm())
    gerr = float((x.grad.double().cpu() - ref_dx).norm() / ref_dx.norm())
    assert err <= fwd_tol and gerr <= bwd_tol, (err, gerr)


MERGE_FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "swin_merge_*.npz")))


@pytest.mark.parametrize("path", MERGE_FILES, ids=[os.path.basename(p)[11:-4] for p in MERGE_FILES])
@pytest.mark.parametrize("dtype,fwd_tol,bwd_tol", [(torch.float32, 1e-4, 1e-3), (torch.bfloat16, 3e-2, 6e-2)])
def test_patch_merging_matches_transformers_golden(gpu_device, path, dtype, fwd_tol, bwd_tol):
    """py4cast_amd/swinunetr.py::PatchMerging (one gather of the 2 x 2 patches, row LayerNorm, the reduction on the row-GEMM kernels)
    against transformers' SwinPatchMerging, even and odd grids (tests/golden/make_golden_swin_merge.py)"""
    from py4cast_amd.swinunetr import PatchMerging

    z = np.load(path, allow_pickle=False)
    meta = eval(str(z["meta"]))
    m = PatchMerging(meta["dim"]).to(gpu_device)
    m.load_state_dict({k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith("w_")})
    x = torch.from_numpy(z["x"]).to(gpu_device).to(dtype).requires_grad_(True)
    out = m(x)
    out.backward(torch.from_numpy(z["gy"]).to(gpu_device).to(dtype))
    ref_out, ref_dx = torch.from_numpy(z["out"]), torch.from_numpy(z["dx"])
    assert tuple(out.shape) == tuple(ref_out.shape)
    err = float((out.detach().double().cpu() - ref_out).norm() / ref_out.norm())
    gerr = float((x.grad.double().cpu() - ref_dx).norm() / ref_dx.norm())
    assert err <= fwd_tol and gerr <= bwd_tol, (err, gerr)
