"""Golden vectors for the Swin block (window partition, cyclic shift + mask, relative-position bias, attention, MLP) from an INDEPENDENT
implementation that is importable in the build container: ``transformers.models.swin.modeling_swin.SwinLayer`` (transformers 5.15.0;
the reference's own SwinUNETR comes from mfai v5.0.1 / MONAI, absent here -- SURVEY.md section 8c).  The oracle's restatement
(oracle/swinunetr.py::SwinBlock + oracle/window_attention.py) and the HIP block (py4cast_amd/swinunetr.py::SwinBlock) are checked against
these fixtures: tests/test_swin_golden_cpu.py (oracle, <= 1e-6) and tests/test_widen_gpu.py::test_swin_block_matches_transformers_golden
(HIP, fp32 flavour <= 1e-4).

Run in the build container only (``python tests/golden/make_golden_swin.py``): writes tests/golden/swin_layer_*.npz.  The fixtures hold
numbers only: inputs, weights under the ORACLE's parameter names, the layer's output, the input gradient for a fixed upstream gradient,
the attention mask and the relative-position index transformers builds."""
import os

import numpy as np
import torch
from transformers.models.swin.configuration_swin import SwinConfig
from transformers.models.swin.modeling_swin import SwinLayer, window_partition

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [
    # name, dim, heads, H, W, shift
    ("plain_24x3_28x28", 24, 3, 28, 28, 0),
    ("shift_24x3_28x28", 24, 3, 28, 28, 3),
    ("shift_pad_24x3_20x26", 24, 3, 20, 26, 3),      # H, W not multiples of the window: padded to 21 x 28
    ("shift_48x6_14x21", 48, 6, 14, 21, 3),
    ("small_grid_24x3_7x7", 24, 3, 7, 7, 3),          # grid == window: the shift is dropped
]


def main():
    for name, dim, heads, H, W, shift in CASES:
        torch.manual_seed(sum(map(ord, name)))
        cfg = SwinConfig(window_size=7, mlp_ratio=4.0, qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                         hidden_act="gelu", layer_norm_eps=1e-5)
        layer = SwinLayer(cfg, dim, (H, W), heads, drop_path_rate=0.0, shift_size=shift).double().eval()
        with torch.no_grad():
            for p in layer.parameters():
                p.normal_(0.0, 0.2)
            layer.layernorm_before.weight.add_(1.0)
            layer.layernorm_after.weight.add_(1.0)
        B = 1
        x = torch.randn(B, H, W, dim).double().requires_grad_(True)      # float32-representable inputs: stored as float32
        gy = torch.randn(B, H, W, dim).double()
        out = layer(x.view(B, H * W, dim), (H, W))[0].view(B, H, W, dim)
        out.backward(gy)
        att = layer.attention            # transformers 5.x names: q_proj / k_proj / v_proj / o_proj, relative_position_bias.*, mlp.fc1 / fc2
        sd = {
            "norm1.weight": layer.layernorm_before.weight, "norm1.bias": layer.layernorm_before.bias,
            "qkv.weight": torch.cat([att.q_proj.weight, att.k_proj.weight, att.v_proj.weight], 0),
            "qkv.bias": torch.cat([att.q_proj.bias, att.k_proj.bias, att.v_proj.bias], 0),
            "proj.weight": att.o_proj.weight, "proj.bias": att.o_proj.bias,
            "relative_position_bias_table": att.relative_position_bias.relative_position_bias_table,
            "norm2.weight": layer.layernorm_after.weight, "norm2.bias": layer.layernorm_after.bias,
            "fc1.weight": layer.mlp.fc1.weight, "fc1.bias": layer.mlp.fc1.bias, "fc2.weight": layer.mlp.fc2.weight, "fc2.bias": layer.mlp.fc2.bias,
        }
        Hp, Wp = (H + 6) // 7 * 7, (W + 6) // 7 * 7
        eff_shift = int(layer.shift_size)
        mask = layer.get_attn_mask(Hp, Wp, dtype=torch.float64, device=x.device)
        # window partition of a pixel-index map: which (padded) pixel lands at which (window, position)
        idx_map = torch.arange(Hp * Wp, dtype=torch.float64).view(1, Hp, Wp, 1)
        part = window_partition(idx_map, 7).view(-1, 49).long()
        out_npz = {f"w_{k}": v.detach().numpy().astype(np.float64) for k, v in sd.items()}
        out_npz.update(x=x.detach().numpy().astype(np.float32), gy=gy.numpy().astype(np.float32), out=out.detach().numpy(), dx=x.grad.numpy(),
                       relative_position_index=att.relative_position_bias.relative_position_index.numpy().astype(np.int64),
                       window_partition_index=part.numpy().astype(np.int64),
                       attn_mask=(mask.numpy() if mask is not None else np.zeros((0, 49, 49))),
                       meta=np.array(repr({"dim": dim, "heads": heads, "H": H, "W": W, "shift": shift, "effective_shift": eff_shift,
                                           "window": 7, "transformers": __import__("transformers").__version__})))
        path = os.path.join(HERE, f"swin_layer_{name}.npz")
        np.savez_compressed(path, **out_npz)
        print(path, os.path.getsize(path) // 1024, "KiB", "effective shift", eff_shift)


if __name__ == "__main__":
    main()
