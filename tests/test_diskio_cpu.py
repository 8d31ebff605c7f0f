"""Host side of SURVEY.md 8f-4: the reference's on-disk layouts (.npy plane per date and parameter, statistics .pt files)."""
import datetime as dt

import numpy as np
import pytest
import torch

from py4cast_amd import diskio


def test_stats_file_round_trip(tmp_path):
    stats = {"t2m_2m": {"mean": torch.tensor(280.5), "std": torch.tensor(7.25), "min": torch.tensor(230.0), "max": torch.tensor(320.0)},
             "u_500hpa": {"mean": 1.5, "std": np.float32(9.0)}}
    f = tmp_path / "parameters_stats.pt"
    diskio.save_stats(stats, f)
    raw = torch.load(f, "cpu", weights_only=True)          # what the reference's Stats.__post_init__ does (access.py:359-360)
    assert set(raw) == set(stats) and raw["t2m_2m"]["std"].dim() == 0 and raw["u_500hpa"]["mean"].dtype == torch.float32
    st = diskio.load_stats(f)
    assert torch.equal(st.to_list("mean", ["u_500hpa", "t2m_2m"]), torch.tensor([1.5, 280.5]))


def test_titan_plane_path_matches_reference_naming(tmp_path):
    d = dt.datetime(2023, 3, 19, 12, 0)
    p = diskio.titan_plane_path(tmp_path, "aro_t2m", 2, "heightAboveGround", d)
    assert str(p).endswith("data/2023-03-19_12h00/aro_t2m_2m.npy")
    p = diskio.titan_plane_path(tmp_path, "aro_z", 500, "isobaricInhPa", d)
    assert str(p).endswith("data/2023-03-19_12h00/aro_z_500hpa.npy")


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16])
def test_plane_reader_equals_np_load(tmp_path, dtype):
    rng = np.random.default_rng(0)
    F, B, T, H, W = 3, 2, 2, 5, 7
    paths, ref = [], np.zeros((F, B, T, H, W), np.float32)
    for f in range(F):
        pf = []
        for b in range(B):
            pb = []
            for t in range(T):
                arr = (rng.standard_normal((H, W)) * 100).astype(dtype)
                p = tmp_path / f"p{f}_{b}_{t}.npy"
                np.save(p, arr)
                pb.append(p)
                ref[f, b, t] = np.load(p).astype(np.float32)
            pf.append(pb)
        paths.append(pf)
    reader = diskio.NpyPlaneReader((H, W), F, B, T, device=None, pin=False)
    got = reader.read(paths)
    assert got.shape == (F, B, T, H, W) and np.array_equal(got.numpy(), ref)
    with pytest.raises(ValueError):
        diskio.NpyPlaneReader((H, W + 1), F, B, T, pin=False).read(paths)
    with pytest.raises(ValueError):
        reader.read(paths[:2])
