"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.

Torch-native restatement of the Half-UNet used by py4cast (``model_name: HalfUNet``,
config/CLI/model/halfunet.yaml:19-26: num_filters 64, dilation 1, bias False, use_ghost False,
last_activation Identity, absolute_pos_embed False, autopad_enabled True).

The reference takes the network from the third-party package ``mfai`` v5.0.1
(requirements.txt:26; imported at py4cast/models.py:10) which is NOT in /root/reference and not
installable here, and the reference's tests pin no numeric result for it
(tests/test_models.py:64-142 only check "trains without raising").  This file restates the
published architecture (Lu et al. 2022, "Half-UNet") the way mfai builds it:

  5 encoder blocks [conv3x3 -> BatchNorm2d -> ReLU] x2 at ``num_filters`` channels, 2x2 max-pool
  between blocks; every level bilinearly up-sampled (align_corners=False) to full resolution and
  SUMMED; one decoder block of the same shape; 1x1 output conv; last activation.
  Ghost variant: conv to half the channels + depthwise 3x3 "cheap op", concatenated.

Parameter names follow mfai's (``encoder1.enc1conv1.weight`` ...), so a state_dict of this module
loads into the HIP model and vice-versa.  ``norm="group"`` swaps BatchNorm2d for GroupNorm (the
variant named by BASELINE.json's north star).
"""

from collections import OrderedDict
from functools import reduce

import torch
from torch import nn


class GhostModule(nn.Module):
    def __init__(self, in_channels, out_channels=64, bias=False, kernel_size=3, dilation=1, norm="batch", groups=8):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels // 2, kernel_size, padding="same", bias=bias, dilation=dilation)
        self.sepconv = nn.Conv2d(out_channels // 2, out_channels // 2, 3, padding="same", groups=out_channels // 2, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels) if norm == "batch" else nn.GroupNorm(groups, out_channels)
        self.relu = nn.ReLU()

    def forward(self, x):
        x = self.conv(x)
        x2 = self.sepconv(x)
        return self.relu(self.bn(torch.cat([x, x2], dim=1)))


class HalfUNetRef(nn.Module):
    """NCHW in, NCHW out (mfai's HalfUNet has features_last=False)."""

    def __init__(self, in_channels, out_channels, num_filters=64, dilation=1, bias=False, use_ghost=False,
                 last_activation="Identity", norm="batch", groups=8):
        super().__init__()
        nf = num_filters
        blk = lambda cin, name: self._block(cin, nf, name, bias, use_ghost, dilation, norm, groups)
        self.encoder1 = blk(in_channels, "enc1")
        self.pool1 = nn.MaxPool2d(2, 2)
        self.encoder2 = blk(nf, "enc2")
        self.up2 = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        self.pool2 = nn.MaxPool2d(2, 2)
        self.encoder3 = blk(nf, "enc3")
        self.up3 = nn.Upsample(scale_factor=4, mode="bilinear", align_corners=False)
        self.pool3 = nn.MaxPool2d(2, 2)
        self.encoder4 = blk(nf, "enc4")
        self.up4 = nn.Upsample(scale_factor=8, mode="bilinear", align_corners=False)
        self.pool4 = nn.MaxPool2d(2, 2)
        self.encoder5 = blk(nf, "enc5")
        self.up5 = nn.Upsample(scale_factor=16, mode="bilinear", align_corners=False)
        self.decoder = blk(nf, "decoder")
        self.outconv = nn.Conv2d(nf, out_channels, kernel_size=1, bias=bias)
        self.activation = getattr(nn, last_activation)()

    @staticmethod
    def _block(in_channels, features, name, bias, use_ghost, dilation, norm, groups):
        mk_norm = (lambda: nn.BatchNorm2d(features)) if norm == "batch" else (lambda: nn.GroupNorm(groups, features))
        if use_ghost:
            layers = [
                (name + "ghost1", GhostModule(in_channels, features, bias, dilation=dilation, norm=norm, groups=groups)),
                (name + "ghost2", GhostModule(features, features, bias, dilation=dilation, norm=norm, groups=groups)),
            ]
        else:
            layers = [
                (name + "conv1", nn.Conv2d(in_channels, features, 3, padding="same", bias=bias, dilation=dilation)),
                (name + "norm1", mk_norm()),
                (name + "relu1", nn.ReLU(inplace=True)),
                (name + "conv2", nn.Conv2d(features, features, 3, padding="same", bias=bias, dilation=dilation)),
                (name + "norm2", mk_norm()),
                (name + "relu2", nn.ReLU(inplace=True)),
            ]
        return nn.Sequential(OrderedDict(layers))

    def forward(self, x):
        enc1 = self.encoder1(x)
        enc2 = self.encoder2(self.pool1(enc1))
        enc3 = self.encoder3(self.pool2(enc2))
        enc4 = self.encoder4(self.pool3(enc3))
        enc5 = self.encoder5(self.pool4(enc4))
        summed = reduce(
            torch.Tensor.add_,
            [enc1, self.up2(enc2), self.up3(enc3), self.up4(enc4), self.up5(enc5)],
            torch.zeros_like(enc1),
        )
        return self.activation(self.outconv(self.decoder(summed)))
