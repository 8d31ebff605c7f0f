"""Golden vectors for Swin's patch merging (the 2 x 2 strided concatenation order, the zero padding of odd grids, LayerNorm(4C), the
bias-free reduction 4C -> 2C) from an INDEPENDENT implementation importable in the build container:
``transformers.models.swin.modeling_swin.SwinPatchMerging`` (same generator pattern as make_golden_swin.py; the reference's SwinUNETR
comes from mfai v5.0.1 / MONAI, absent here -- SURVEY.md section 8c).  Checked by tests/test_swin_golden_cpu.py (oracle/swinunetr.py::
PatchMerging, <= 1e-6) and tests/test_swin_golden_gpu.py (py4cast_amd/swinunetr.py::PatchMerging, fp32 <= 1e-4).

Run in the build container only (``python tests/golden/make_golden_swin_merge.py``): writes tests/golden/swin_merge_*.npz (numbers only)."""
import os

import numpy as np
import torch
from transformers.models.swin.modeling_swin import SwinPatchMerging

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("even_24_28x28", 24, 28, 28), ("odd_h_24_21x28", 24, 21, 28), ("odd_both_48_13x7", 48, 13, 7)]


def main():
    for name, dim, H, W in CASES:
        torch.manual_seed(sum(map(ord, name)))
        m = SwinPatchMerging(dim).double().eval()
        with torch.no_grad():
            for p in m.parameters():
                p.normal_(0.0, 0.2)
            m.norm.weight.add_(1.0)
        B = 2
        x = torch.randn(B, H, W, dim).float().double().requires_grad_(True)
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        gy = torch.randn(B, Ho, Wo, 2 * dim).float().double()
        out = m(x.view(B, H * W, dim), (H, W)).view(B, Ho, Wo, 2 * dim)
        out.backward(gy)
        # which input pixel / channel lands in which merged channel: the merge of an index map
        idx = torch.arange(H * W * dim, dtype=torch.float64).view(1, H, W, dim) + 1.0       # (0 = padding)
        xp = m.maybe_pad(idx, H, W)
        merged = torch.cat([xp[:, row::2, col::2, :] for col in range(2) for row in range(2)], dim=-1).long()
        np.savez_compressed(os.path.join(HERE, f"swin_merge_{name}.npz"),
                            **{"w_norm.weight": m.norm.weight.detach().numpy(), "w_norm.bias": m.norm.bias.detach().numpy(),
                               "w_reduction.weight": m.reduction.weight.detach().numpy()},
                            x=x.detach().numpy().astype(np.float32), gy=gy.numpy().astype(np.float32), out=out.detach().numpy(), dx=x.grad.numpy(),
                            merge_index=merged.numpy().astype(np.int64),
                            meta=np.array(repr({"dim": dim, "H": H, "W": W, "transformers": __import__("transformers").__version__})))
        print(name, out.shape)


if __name__ == "__main__":
    main()
