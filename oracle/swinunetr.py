"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (mfai v5.0.1 / MONAI are absent).

Torch-native restatement of the 2-D SwinUNETR selected by ``model_name: SwinUNetR`` (config/CLI/model/swinunetr.yaml:19-30),
written the way the published implementation runs it: every Swin block goes through torch.roll, window_partition, per-head
softmax attention with relative position bias and shift mask, window_reverse (oracle/window_attention.py); LayerNorm, Linear,
convolutions and instance norms are torch's.  Parameter names match py4cast_amd.swinunetr.SwinUNetRMI355X (one state_dict).
"""

import torch
import torch.nn.functional as F
from torch import nn

from . import window_attention as owa


class SwinBlock(nn.Module):
    def __init__(self, dim, heads, ws, shift, mlp_ratio=4.0):
        super().__init__()
        self.heads, self.ws, self.shift = heads, ws, shift
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.norm2 = nn.LayerNorm(dim)
        self.fc1 = nn.Linear(dim, int(dim * mlp_ratio))
        self.fc2 = nn.Linear(int(dim * mlp_ratio), dim)

    def forward(self, x):
        B, H, W, C = x.shape
        ws = self.ws
        shift = self.shift if min(H, W) > ws else 0
        h = self.norm1(x)
        pb, pr = (-H) % ws, (-W) % ws
        h = F.pad(h, (0, 0, 0, pr, 0, pb))
        N = ws * ws
        idx = owa.relative_position_index(ws).view(-1)
        bias = self.relative_position_bias_table[idx].view(N, N, self.heads).permute(2, 0, 1)
        a = self.proj(owa.window_attention(self.qkv(h), bias, self.heads, ws, shift))[:, :H, :W, :]
        x = x + a
        return x + self.fc2(F.gelu(self.fc1(self.norm2(x))))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def forward(self, x):
        # (B, H, W, C): an odd grid is zero-padded at the bottom / right first, as MONAI's PatchMergingV2 and transformers'
        # SwinPatchMerging do (pinned to the latter: tests/golden/make_golden_swin_merge.py)
        H, W = x.shape[1], x.shape[2]
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], dim=-1)
        return self.reduction(self.norm(x))


class ResBlock(nn.Module):
    def __init__(self, cin, cout):
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
        out = F.leaky_relu(self.norm1(self.conv1(x)), 0.01)
        out = self.norm2(self.conv2(out))
        res = self.norm3(self.conv3(x)) if self.down else x
        return F.leaky_relu(out + res, 0.01)


class UpBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.transp_conv = nn.ConvTranspose2d(cin, cout, 2, stride=2, bias=False)
        self.conv_block = ResBlock(2 * cout, cout)

    def forward(self, x, skip):
        return self.conv_block(torch.cat([self.transp_conv(x), skip], dim=1))


class SwinUNetR(nn.Module):
    def __init__(self, in_channels, out_channels, depths=(2, 2, 2, 2), num_heads=(3, 6, 12, 24), feature_size=24, window_size=7,
                 normalize=True):
        super().__init__()
        fs, ws = feature_size, window_size
        self.normalize = normalize
        self.patch_embed = nn.Conv2d(in_channels, fs, 2, stride=2)
        self.stages, self.merges = nn.ModuleList(), nn.ModuleList()
        for i, (depth, heads) in enumerate(zip(depths, num_heads)):
            dim = fs * 2 ** i
            self.stages.append(nn.ModuleList([SwinBlock(dim, heads, ws, 0 if j % 2 == 0 else ws // 2) for j in range(depth)]))
            self.merges.append(PatchMerging(dim))
        self.encoder1, self.encoder2 = ResBlock(in_channels, fs), ResBlock(fs, fs)
        self.encoder3, self.encoder4 = ResBlock(2 * fs, 2 * fs), ResBlock(4 * fs, 4 * fs)
        self.encoder10 = ResBlock(16 * fs, 16 * fs)
        self.decoder5, self.decoder4 = UpBlock(16 * fs, 8 * fs), UpBlock(8 * fs, 4 * fs)
        self.decoder3, self.decoder2, self.decoder1 = UpBlock(4 * fs, 2 * fs), UpBlock(2 * fs, fs), UpBlock(fs, fs)
        self.out = nn.Conv2d(fs, out_channels, 1)

    def _hidden(self, t):
        if self.normalize:
            t = F.layer_norm(t, (t.shape[-1],))
        return t.permute(0, 3, 1, 2)

    def forward(self, x):
        xin = x.permute(0, 3, 1, 2)
        t = self.patch_embed(xin).permute(0, 2, 3, 1)
        hidden = [self._hidden(t)]
        for blocks, merge in zip(self.stages, self.merges):
            for blk in blocks:
                t = blk(t)
            t = merge(t)
            hidden.append(self._hidden(t))
        enc0, enc1 = self.encoder1(xin), self.encoder2(hidden[0])
        enc2, enc3 = self.encoder3(hidden[1]), self.encoder4(hidden[2])
        dec3 = self.decoder5(self.encoder10(hidden[4]), hidden[3])
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        dec0 = self.decoder2(dec1, enc1)
        return self.out(self.decoder1(dec0, enc0)).permute(0, 2, 3, 1)
