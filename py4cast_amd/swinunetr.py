"""
SwinUNETR with the fused MI355X window attention -- the model behind ``model_name: SwinUNetR``
(config/CLI/model/swinunetr.yaml:19-30: depths [2,2,2,2], num_heads [3,6,12,24], feature_size 24, norm_name instance,
downsample merging, use_v2 False).  The reference takes the class from mfai v5.0.1 (py4cast/models.py:10-20), which wraps MONAI's
2-D SwinUNETR (Hatamizadeh et al. 2022): patch-2 embedding, four Swin stages (window 7, alternating shift 0 / 3, relative position
bias, patch merging at the end of each stage) and a UNETR convolutional decoder (residual conv blocks with instance norm, transposed
conv up-sampling, skip connections).  PARITY UNPINNED against mfai (absent here); the arithmetic is checked against
oracle/swinunetr.py (the same parameters through roll / window_partition / softmax / window_reverse).

What runs where
* window attention of every Swin block: ONE HIP kernel each way (csrc/attention.hip through py4cast_amd.ops_attention) on the
  (B, Hp, Wp, 3C) output of the qkv Linear -- shift, window partition, head split, bias, mask, softmax, PV and all inverses;
* LayerNorms of the Swin blocks: csrc/rows.hip (row LayerNorm) when the row fits its limits;
* UNETR decoder (round 2): features-last throughout, no layout change anywhere; its 3x3 / 1x1 convolutions with <= 96 input and
  <= 64 output channels -- everything from the 1/4 resolution upwards -- on the MFMA conv kernels (ops_model.conv_nhwc: forward,
  data and weight gradient); the 2x2 transposed convolutions, the patch embedding and the output convolution are GEMMs over
  pixel blocks (library); instance norm + LeakyReLU + residual add = one native node each (csrc/inorm.hip, ops_inorm).  The four wider
  convolutions at <= 1/8 resolution still go through MIOpen (_conv_hw).
* Linear layers: library GEMMs.
Input / output are features-last (B, H, W, C); H and W must be multiples of 32.
"""

from dataclasses import dataclass, field
from typing import Tuple

import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from . import ops_inorm as ON
from . import ops_gemm as G
from . import ops_model as OM
from . import ops_rows as R
from .base import ModelABC, ModelType
from .ops_attention import window_attention

try:
    from dataclasses_json import dataclass_json
except Exception:  # pragma: no cover
    def dataclass_json(cls):
        return cls


@dataclass_json
@dataclass(slots=True)
class SwinUNetRSettings:
    depths: Tuple[int, ...] = (2, 2, 2, 2)
    num_heads: Tuple[int, ...] = (3, 6, 12, 24)
    feature_size: int = 24
    norm_name: str = "instance"
    drop_rate: float = 0.0
    attn_drop_rate: float = 0.0
    dropout_path_rate: float = 0.0
    normalize: bool = True
    use_checkpoint: bool = False
    downsample: str = "merging"
    use_v2: bool = False
    window_size: int = 7
    activation_dtype: str = "f32"   # "bf16": token tensors of the Swin stages stored as bf16


def relative_position_index(ws: int) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _row_ln_ok(x: torch.Tensor) -> bool:
    C = x.shape[-1]
    return (C * x.element_size()) % 16 == 0 and C * x.element_size() <= 1024


def _layer_norm(m: nn.LayerNorm, x: torch.Tensor, real=None) -> torch.Tensor:
    """real = (H, W): x is a PADDED map (B,Hp,Wp,C) whose tokens beyond (H, W) are padding -- their rows come out zero"""
    C = x.shape[-1]
    L.require_cuda(x)   # the model has no CPU path (rows beyond the kernel's limits use the GPU library's LayerNorm)
    if _row_ln_ok(x):
        mask = None if real is None else (x.shape[1], x.shape[2], real[0], real[1])
        return R.row_layer_norm(x.reshape(-1, C), m.weight, m.bias, m.eps, mask=mask).view(x.shape)
    if real is not None:
        raise L.P4CError("SwinUNetRMI355X: a padded stage needs the native row LayerNorm")
    return F.layer_norm(x.float(), m.normalized_shape, m.weight, m.bias, m.eps).to(x.dtype)


class _TableRows(torch.autograd.Function):
    """table[index] for a STATIC index (Swin's relative-position bias: 2 401 = 49 x 49 rows gathered from a (169, heads) table).  The
    library's backward is a sort-based index_put (60 us per call, 24 calls per step); with the index fixed its inverse is a table
    too: `rows_of[r]` lists the gradient rows that belong to table row r (padded with the index of an appended zero row), and the
    gradient is one gather + one sum over that list -- a fixed order, no atomics, no host synchronisation (capturable)."""

    @staticmethod
    def forward(ctx, table, index, rows_of):
        ctx.save_for_backward(rows_of)
        return table[index]

    @staticmethod
    def backward(ctx, dy):
        rows_of, = ctx.saved_tensors
        return F.pad(dy, (0, 0, 0, 1))[rows_of].sum(dim=1), None, None


def inverse_index_table(index: torch.Tensor, rows: int) -> torch.Tensor:
    """(rows, max multiplicity) positions of `index` holding each value, padded with len(index)."""
    index = index.view(-1)
    counts = torch.bincount(index, minlength=rows)
    order = torch.argsort(index, stable=True)
    out = torch.full((rows, int(counts.max())), index.numel(), dtype=torch.long)
    start = 0
    for r, c in enumerate(counts.tolist()):
        out[r, :c] = order[start:start + c]
        start += c
    return out


def _native(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype == torch.bfloat16 and L.diag_switch("P4C_SWIN_LIBRARY") != "1"


def _lin(x: torch.Tensor, w: torch.Tensor, b=None, res=None) -> torch.Tensor:
    """x W^T + b (+ res): the streaming row-GEMM kernels (csrc/rowgemm.hip) for the two large stages' narrow layers they were measured
    on, the tiled MFMA GEMM with its fused epilogue (csrc/gemm.hip, round 5) for the deep stages' wide ones -- library GEMMs only for
    the fp32 flavour / widths off the 8-feature granularity"""
    if _native(x) and G.supported(x, w) and not R._row_gemm_ok(x, w, b):
        return G.linear(x, w, b, res)
    if res is not None:
        return R.linear_res(x, w, b, res)
    return R.linear_nd(x, w, b)


def _linear(m: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    return _lin(x, m.weight, m.bias)


def _mlp(fc1: nn.Linear, fc2: nn.Linear, x: torch.Tensor, res: torch.Tensor) -> torch.Tensor:
    """res + fc2(gelu(fc1(x))): one autograd node with GELU / GELU' in the GEMM epilogues where the tiled GEMM carries both layers"""
    if (_native(x) and G.supported(x, fc1.weight) and G.supported(x, fc2.weight) and not R._row_gemm_ok(x, fc1.weight, fc1.bias)):
        return G.mlp(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, res)
    if _native(x) and R.row_mlp_gelu_ok(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias):
        return R.row_mlp_gelu(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, res)       # the two large stages: streaming row GEMMs
    return res + _linear(fc2, F.gelu(_linear(fc1, x)))


class SwinBlock(nn.Module):
    def __init__(self, dim: int, heads: int, ws: int, shift: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.dim, self.heads, self.ws, self.shift = dim, heads, ws, shift
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        idx = relative_position_index(ws)
        self.register_buffer("relative_position_index", idx, persistent=False)
        self.register_buffer("_rpi_rows_of", inverse_index_table(idx, (2 * ws - 1) ** 2), persistent=False)   # for the gather's backward
        self.norm2 = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)

    def forward(self, x: torch.Tensor, real=None) -> torch.Tensor:
        """real = (H, W): x arrives PADDED to multiples of the window (B,Hp,Wp,C) and stays so (`padded_stage`): MONAI's per-block
        `F.pad(norm1(x))` ... `x[:, :h, :w]` becomes a LayerNorm that writes zero rows for the padding tokens -- everything after it is
        row-wise or the attention itself, so the real tokens see exactly the same numbers and the padding rows of the residual stream
        (never read by a real token: norm1 masks them again) carry no gradient."""
        B, H, W, C = x.shape
        ws = self.ws
        if real is not None:
            H, W = real
        shift = self.shift if min(H, W) > ws else 0
        h = _layer_norm(self.norm1, x, real)
        pb, pr = (0, 0) if real is not None else ((-H) % ws, (-W) % ws)
        if pb or pr:
            h = F.pad(h, (0, 0, 0, pr, 0, pb))
        N = ws * ws
        bias = _TableRows.apply(self.relative_position_bias_table, self.relative_position_index.view(-1),
                                self._rpi_rows_of).view(N, N, self.heads).permute(2, 0, 1)
        a = window_attention(_linear(self.qkv, h), bias, self.heads, ws, shift)
        if pb or pr:
            x = x + _linear(self.proj, a)[:, :H, :W, :]
        else:
            x = _lin(a, self.proj.weight, self.proj.bias, res=x)       # the residual in the projection's epilogue
        return _mlp(self.fc1, self.fc2, _layer_norm(self.norm2, x), x)


def padded_stage(blocks, t: torch.Tensor) -> torch.Tensor:
    """The blocks of one Swin stage.  A map that is not a multiple of the window (512 x 512 input, window 7: 256, 128, 64, 32 -> 259, 133,
    70, 35) is padded ONCE for the stage instead of once per block (and cropped once: a view, the patch merging gathers from it):
    per block that was a fill + a copy forward, a slice + an add, and their zero fill / copy / add backward -- 1.3 ms of the 25 ms step."""
    B, H, W, C = t.shape
    ws = blocks[0].ws
    pb, pr = (-H) % ws, (-W) % ws
    if not (pb or pr) or not (t.is_cuda and _row_ln_ok(t)) or L.diag_switch("P4C_SWIN_PAD_PER_BLOCK") == "1":
        for blk in blocks:
            t = blk(t)
        return t
    t = F.pad(t, (0, 0, 0, pr, 0, pb))
    for blk in blocks:
        t = blk(t, real=(H, W))
    return t[:, :H, :W, :]


class PatchMerging(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # [x(0::2,0::2) | x(1::2,0::2) | x(0::2,1::2) | x(1::2,1::2)] along the channels (MONAI's order) as ONE gather of the 2 x 2
        # patches: the concatenation of four strided slices cost, backward, four zero fills, four strided copies and three additions
        # an odd grid is zero-padded at the bottom / right first (MONAI's PatchMergingV2, transformers' SwinPatchMerging: the golden
        # vectors of tests/golden/make_golden_swin_merge.py pin order, padding, LayerNorm and reduction)
        B, H, W, C = x.shape
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
            H, W = H + H % 2, W + W % 2
        x = x.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, H // 2, W // 2, 4 * C)
        return _linear(self.reduction, _layer_norm(self.norm, x))


def _conv_hw(m: nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """A bias-free "same" convolution on a features-last tensor (B,H,W,C).  Up to 96 input and 64 output channels -- every
    convolution of the decoder from the 1/4 resolution upwards, where its time goes -- run on the MFMA conv kernels (forward, data
    and weight gradient: ops_model.conv_nhwc); the few wider ones at <= 1/8 resolution go through the library in its own layout."""
    if OM.conv_nhwc_supported(x, m.weight):
        return OM.conv_nhwc(x, m.weight)
    if _native(x) and m.bias is None and G.conv_supported(x, m.weight) and m.kernel_size[0] in (1, 3) and m.padding == (m.kernel_size[0] // 2,) * 2:
        return G.conv2d_nhwc(x.contiguous(), m.weight)       # the wide ones (96 ... 384 channels): implicit GEMM, csrc/gemm.hip
    y = OM.library_conv2d(x.permute(0, 3, 1, 2).contiguous(), R.param_as(m.weight, x.dtype), None, padding=m.padding)
    return y.permute(0, 2, 3, 1).contiguous()


def _inorm(m: nn.InstanceNorm2d, x: torch.Tensor, slope: float = 1.0, res=None) -> torch.Tensor:
    """leaky_relu(InstanceNorm2d(affine)(x) (+ res), slope) on a features-last tensor as one native node (csrc/inorm.hip): statistics
    over (H, W) per sample and channel in fp32; slope = 1: no activation."""
    return ON.instance_norm_act(x, m.weight, m.bias, m.eps, slope, res)


class ResBlock(nn.Module):
    """MONAI's UnetResBlock: conv3x3 - IN - LeakyReLU - conv3x3 - IN, (+ 1x1 conv - IN on the skip when channels differ), LeakyReLU.
    Features-last (B,H,W,C) in and out."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.norm1 = nn.InstanceNorm2d(cout, affine=True)
        self.norm2 = nn.InstanceNorm2d(cout, affine=True)
        self.down = cin != cout
        if self.down:
            self.conv3 = nn.Conv2d(cin, cout, 1, bias=False)
            self.norm3 = nn.InstanceNorm2d(cout, affine=True)

    def forward(self, x):
        out = _inorm(self.norm1, _conv_hw(self.conv1, x), 0.01)
        res = _inorm(self.norm3, _conv_hw(self.conv3, x)) if self.down else x
        return _inorm(self.norm2, _conv_hw(self.conv2, out), 0.01, res)    # lrelu(norm2(conv2(out)) + res), one pass


class UpBlock(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.transp_conv = nn.ConvTranspose2d(cin, cout, 2, stride=2, bias=False)
        self.conv_block = ResBlock(2 * cout, cout)

    def forward(self, x, skip):
        # 2x2 / stride-2 transposed convolution = one GEMM per pixel block: (B*H*W, cin) @ (cin, 2*2*cout), then the 2x2 outputs of
        # every input pixel are interleaved into the up-sampled grid
        B, H, W, cin = x.shape
        wt = self.transp_conv.weight                                         # (cin, cout, 2, 2)
        cout = wt.shape[1]
        if _native(x) and cin % 8 == 0 and cout % 2 == 0:
            up = _lin(x.reshape(-1, cin), wt.permute(2, 3, 1, 0).reshape(4 * cout, cin))
        else:
            up = x.reshape(-1, cin) @ wt.permute(0, 2, 3, 1).reshape(cin, 4 * cout).to(x.dtype)
        up = up.view(B, H, W, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, cout)
        return self.conv_block(torch.cat([up, skip], dim=-1))


class SwinUNetRMI355X(ModelABC, nn.Module):
    settings_kls = SwinUNetRSettings
    onnx_supported: bool = False
    supported_num_spatial_dims = (2,)
    num_spatial_dims: int = 2
    features_last: bool = True
    model_type = ModelType.VISION_TRANSFORMER
    register: bool = True

    def __init__(self, in_channels: int, out_channels: int, input_shape: Tuple[int, ...] = None,
                 settings: SwinUNetRSettings = SwinUNetRSettings(), *args, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.input_shape = in_channels, out_channels, input_shape
        self._settings = settings
        if settings.use_v2 or settings.downsample != "merging" or settings.norm_name != "instance":
            raise NotImplementedError("SwinUNetR: only use_v2=False, downsample='merging', norm_name='instance' (the reference yaml)")
        if input_shape is not None and (input_shape[0] % 32 or input_shape[1] % 32):
            raise ValueError(f"SwinUNetR: grid {tuple(input_shape)} must be a multiple of 32 (patch 2 x four 2x merges)")
        fs, ws = settings.feature_size, settings.window_size
        self.patch_embed = nn.Conv2d(in_channels, fs, 2, stride=2)
        self.stages = nn.ModuleList()
        self.merges = nn.ModuleList()
        for i, (depth, heads) in enumerate(zip(settings.depths, settings.num_heads)):
            dim = fs * 2 ** i
            self.stages.append(nn.ModuleList([SwinBlock(dim, heads, ws, 0 if j % 2 == 0 else ws // 2) for j in range(depth)]))
            self.merges.append(PatchMerging(dim))
        self.encoder1 = ResBlock(in_channels, fs)
        self.encoder2 = ResBlock(fs, fs)
        self.encoder3 = ResBlock(2 * fs, 2 * fs)
        self.encoder4 = ResBlock(4 * fs, 4 * fs)
        self.encoder10 = ResBlock(16 * fs, 16 * fs)
        self.decoder5 = UpBlock(16 * fs, 8 * fs)
        self.decoder4 = UpBlock(8 * fs, 4 * fs)
        self.decoder3 = UpBlock(4 * fs, 2 * fs)
        self.decoder2 = UpBlock(2 * fs, fs)
        self.decoder1 = UpBlock(fs, fs)
        self.out = nn.Conv2d(fs, out_channels, 1)
        self.timed_entry_points = ("p4c_window_attn_fwd", "p4c_window_attn_bwd", "p4c_row_layernorm_fwd", "p4c_row_layernorm_bwd",
                                   "p4c_row_gemm", "p4c_row_gemm_wgrad", "p4c_gemm_nt", "p4c_gemm_tn")
        self.roofline_from_entry_points = True   # bench.py: time every call of the entry points above
        self.prefers_hip_graph = True            # ~10^3-10^4 launches per training step: replay them from a HIP graph (trainer.GraphedTrainingStep)
        self._unit_affine = {}                   # (C, device) -> constant (ones, zeros) rows of the hidden states' non-affine LayerNorm
        self.check_required_attributes()

    @property
    def settings(self) -> SwinUNetRSettings:
        return self._settings

    def _hidden(self, t: torch.Tensor, dt) -> torch.Tensor:
        """A Swin hidden state handed to the decoder: layer-normalised over channels without affine (MONAI's proj_out); stays
        features-last."""
        if self._settings.normalize:
            C = t.shape[-1]
            if (C * t.element_size()) % 16 == 0 and C * t.element_size() <= 1024:
                # the native row LayerNorm with constant unit weight / zero bias (statistics in fp32 from the rows as they are): the
                # library route was a cast to fp32, an fp32 LayerNorm and a cast back -- 130 us for the first hidden state alone
                key = (C, t.device)
                if key not in self._unit_affine:
                    self._unit_affine[key] = (torch.ones(C, device=t.device), torch.zeros(C, device=t.device))
                g, b = self._unit_affine[key]
                return R.row_layer_norm(t.reshape(-1, C), g, b, 1e-5).view(t.shape).to(dt)
            t = F.layer_norm(t.float(), (C,))
        return t.to(dt)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,H,W,C) -> (B,H,W,out).  Everything is features-last: no layout change anywhere in the network."""
        L.require_cuda(x)
        dt = torch.bfloat16 if self._settings.activation_dtype == "bf16" else torch.float32
        B, H, W, Cx = x.shape
        C = self.in_channels
        if Cx != C and (Cx < C or self.rollout_input_format is None or Cx != self.rollout_input_format[1]):
            raise L.P4CError(f"SwinUNetRMI355X: {Cx} input channels, expected {C}")
        xin = x.to(dt)
        # patch embedding: 2x2 / stride-2 convolution with bias = a GEMM over the 2x2 patches
        pw = self.patch_embed.weight.permute(0, 2, 3, 1)                      # (fs, 2, 2, C)
        xp, Cp = xin, C
        if C % 2:   # rows of 4 C features: an even C makes them multiples of 8 (16-byte rows for the row-GEMM kernels); zero weights there
            pw, Cp = F.pad(pw, (0, 1)), C + 1
        if Cx > C:      # input already zero-padded by build_x (rollout_input_format): a channel slice of it, no padded copy
            xp = xin[..., :Cp]
        elif C % 2:
            xp = F.pad(xin, (0, 1))
        patches = xp.view(B, H // 2, 2, W // 2, 2, Cp).permute(0, 1, 3, 2, 4, 5).reshape(B, H // 2, W // 2, 4 * Cp)
        t = R.linear_nd(patches, pw.reshape(pw.shape[0], 4 * Cp), self.patch_embed.bias)
        hidden = [self._hidden(t, dt)]
        for blocks, merge in zip(self.stages, self.merges):
            t = padded_stage(blocks, t)
            t = merge(t)
            hidden.append(self._hidden(t, dt))
        enc0 = self.encoder1(xin)
        enc1 = self.encoder2(hidden[0])
        enc2 = self.encoder3(hidden[1])
        enc3 = self.encoder4(hidden[2])
        dec4 = self.encoder10(hidden[4])
        dec3 = self.decoder5(dec4, hidden[3])
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        dec0 = self.decoder2(dec1, enc1)
        out = self.decoder1(dec0, enc0)
        ow = self.out.weight
        w2, b2, O = ow.view(ow.shape[0], ow.shape[1]), self.out.bias, ow.shape[0]
        if self.rollout_padded_output and O % 8 and out.dtype == torch.bfloat16:
            # inside the rollout rows wider than out_channels are welcome (the state update reads the first out_channels features):
            # zero weight rows up to a multiple of 8 put the output head (524 288 rows x 24 -> 60) on the row-GEMM kernels -- 60 outputs
            # keep them away, and the library GEMMs of this tall-skinny shape take 90 us forward alone
            pad = (-O) % 8
            w2, b2 = F.pad(w2, (0, 0, 0, pad)), None if b2 is None else F.pad(b2, (0, pad))
        y = R.linear_nd(out, w2, b2)
        return y if y.dtype == x.dtype or not x.dtype.is_floating_point else y.to(x.dtype)

    rollout_padded_output = False   # set by the rollout around its calls: rows wider than out_channels are welcome
    rollout_param_proxies = True    # the rollout may run each AR step on stand-ins of the parameters (trainer.RolloutParamProxies)

    @property
    def rollout_input_format(self):
        """(dtype, channel count) the rollout's build_x should emit for this model: bf16 rows zero-padded to the 32-channel multiple
        the first convolutions run on -- otherwise every AR step casts the fp32 input, pads it to an even channel count for the patch
        GEMM and to 96 channels for each of encoder1's two convolutions (and runs the adjoints of all that).  fp32 flavour: None."""
        if self._settings.activation_dtype != "bf16" or self.in_channels > 96 or L.diag_switch("P4C_NO_ROLLOUT_FORMAT") == "1":
            return None
        return torch.bfloat16, (self.in_channels + 31) // 32 * 32

    # ------------------------------------------------------------------ bench.py hook
    def roofline(self, ktimes, B, H, W):
        """Achieved HBM rate of the native entry point that takes the most time (algorithmic bytes stated by the wrappers:
        window attention forward = read qkv + write out, backward = read qkv and dout + write dqkv)."""
        from . import _lib as L

        nbytes = L.kernel_bytes()
        names = [k for k in ktimes if k in nbytes]
        if not names:
            return None
        name = max(names, key=lambda k: ktimes[k][0] * ktimes[k][1])
        calls, avg_ms = ktimes[name]
        gbs = nbytes[name] / (calls * avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": f"{name} (all launches)", "achieved": gbs, "peak": 8000.0, "unit": "GB/s",
                "frac": gbs / 8000.0, "traffic": None, "algorithmic_bytes_per_launch": nbytes[name] / calls,
                "avg_launch_ms": avg_ms, "launches": calls,
                "all": {k: {"calls": ktimes[k][0], "avg_ms": round(ktimes[k][1], 4),
                            "GBps": round(nbytes[k] / (ktimes[k][0] * ktimes[k][1] * 1e-3) / 1e9, 1)} for k in names}}
