"""
UNETR++ on MI355X -- the model behind ``model_name: UNetRPP`` (config/CLI/model/unetrpp.yaml:19-35; BASELINE.json configuration 5).
The reference takes the class from mfai v5.0.1 (py4cast/models.py:10-20), which is absent here: PARITY UNPINNED; the architecture is
restated from the published one (oracle/unetrpp.py spells out every block) and the arithmetic is checked against that oracle, whose
attention forms the matrices literally.

What runs where (state after round 6; DESIGN.md section 7 has the numbers)
* efficient paired attention (EPA), forward and backward, as ONE autograd node (``ops_ts.epa_core``): the tall-skinny HIP kernels
  of csrc/tallskinny.hip on the matrix cores -- ``gram`` (q^T k with the squared column norms of q and k from one pass over the
  16 384 ... 256 tokens of a stage) and ``apply`` (v A^T, q M, S VP^T: per-token small products, the spatial branch's softmax and
  its adjoint in their epilogues), addressed in place inside the qkvv projection's output (no head split / transpose / contiguous
  copies; the literal formulation makes eight of them per block).  The d x d and d x p matrices in between (column norms,
  temperature, channel softmax, scaled token projection) are ONE native launch each way (``p4c_epa_small_fwd/bwd``).
* (x + pos_embed) and its LayerNorm: one pass of csrc/rows.hip; qkvv / out_proj / out_proj2 / conv8 / the 2 x 2 down-samplings /
  the 1 x 1 up-convolutions and every 128 ... 1024-channel 3 x 3 convolution with its batch norm: csrc/gemm.hip (tiled GEMM and
  implicit-GEMM convolution with bias / residual / statistics epilogues, round 5); bilinear up-sampling + skip: csrc/resize.hip.
* the two full-resolution 64-channel residual blocks (encoder1 / decoder2): the MFMA convolution kernels of csrc/conv_rows.hip /
  conv_bf16.hip + the native instance-norm nodes (csrc/inorm.hip); the 1 x 1 output head: a row GEMM (csrc/rowgemm.hip).
* the token-axis projection E = F and its adjoints: the same tall-skinny kernels, k / v_sa read in place over all channels of a sample
  (round 6; a gather + hipBLASLt before); the sums of the token splits, the bias / temperature gradient tails, the published x_SA merge
  (a tiled transpose each way) and the channel dropout in front of conv8 (applied by conv51's last batch-norm pass) are native too;
  a tensor with several consumers (the block's skip, a residual block's input) is handed from consumer to consumer so that its
  gradients are added inside the backward launches (``passthrough``).  What is still tensor-library code: DESIGN.md section 3.12.

``UNetRPPSettings.published_block`` (round 6, default True) selects the block AS PUBLISHED (Shaker et al., the code mfai wraps), so
that state dicts have the published keys and a checkpoint trained there means the same function here:
* the spatial-attention branch is merged as the published code writes it, ``(attn_SA @ v_SA^T).permute(0, 3, 1, 2).reshape(B, N, C)``
  -- a fixed permutation of the (N x C) entries that mixes tokens and channels (``p4c_ts_merge_published``: one tiled transpose);
* ``conv8 = Sequential(Dropout2d(0.1), Conv2d)`` (keys ``conv8.1.*``; the channel dropout is drawn in training mode, p =
  ``conv8_dropout``), ``E`` and ``F`` are one Linear registered under both names (keys ``E.*`` and ``F.*``), and the two attention
  dropouts exist as modules (p = ``dropout_rate``, which must be 0 here);
``published_block=False`` is the restated block of rounds 2-5: x_SA merged head-major per token (``permute(0, 2, 1, 3)``: a free view of
the kernels' output), a bare ``conv8`` convolution, no dropout (a bit-reproducible step).  mfai 5.0.1 is absent from this container, so
both forms are checked against oracle/unetrpp.py with the same switch (PARITY UNPINNED); bench.py names the form it ran.
Input / output are features-last (B, H, W, C); H and W must be multiples of 8 * downsampling_rate.  ``attention_code``
("torch" | "flash" | "manual" in mfai) is accepted and ignored: all of them are this one fused formulation.
"""

import os
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from . import ops_gemm as G
from . import ops_inorm as ON
from . import ops_model as OM
from . import ops_rows as R
from . import ops_ts as TS
from .base import ModelABC, ModelType

try:
    from dataclasses_json import dataclass_json
except Exception:  # pragma: no cover
    def dataclass_json(cls):
        return cls


@dataclass_json
@dataclass(slots=True)
class UNetRPPSettings:
    hidden_size: int = 256
    num_heads_encoder: int = 4
    num_heads_decoder: int = 4
    pos_embed: str = "perceptron"
    norm_name: str = "instance"
    dropout_rate: float = 0.0
    depths: Tuple[int, ...] = (3, 3, 3, 3)
    conv_op: str = "Conv2d"
    do_ds: bool = False
    spatial_dims: int = 2
    linear_upsampling: bool = False
    downsampling_rate: int = 4
    decoder_proj_size: int = 64
    encoder_proj_sizes: Tuple[int, ...] = (64, 64, 64, 32)
    add_skip_connections: bool = True
    attention_code: str = "torch"
    activation_dtype: str = "f32"   # "bf16": token tensors and convolutions in bf16 (trainer.precision bf16)
    published_block: bool = True    # the transformer block as published / as mfai wraps it (module docstring); False: the restated block
    conv8_dropout: float = 0.1      # published_block: p of the channel dropout in front of conv8 (hard-coded 0.1 in the published code)


def _norm(name, ch):
    if name == "instance":
        return nn.InstanceNorm2d(ch, affine=True)
    if name == "batch":
        return nn.BatchNorm2d(ch)
    raise NotImplementedError(f"UNetRPP: norm_name={name}")


def _native(x: torch.Tensor) -> bool:
    """bf16 activations on the GPU: the round-5 kernels (csrc/gemm.hip) carry the wide Linears / convolutions / batch norms"""
    return x.is_cuda and x.dtype == torch.bfloat16 and L.diag_switch("P4C_UNETRPP_LIBRARY") != "1"


def _lin(x: torch.Tensor, w: torch.Tensor, b=None, res=None) -> torch.Tensor:
    """x W^T + b (+ res): the streaming row-GEMM kernels for the narrow layers they were measured on (csrc/rowgemm.hip), the tiled
    MFMA GEMM for the rest (csrc/gemm.hip), library GEMMs only for the fp32 flavour / odd widths"""
    if _native(x) and G.supported(x, w) and not R._row_gemm_ok(x, w, b):
        return G.linear(x, w, b, res)
    y = R.linear_nd(x, w, b)
    return y if res is None else y + res


def _linear(m: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    return _lin(x, m.weight, m.bias)


def _conv(m, x):
    """A convolution on an NCHW-shaped view of features-last memory (channels_last), by shape:
      * 3x3 / 1x1 "same", bias-free, to 64 channels from <= 96 (the full-resolution residual blocks at the yaml's hidden_size
        1024: the HalfUNet kernels' shapes): the native MFMA kernels, forward, data and weight gradient;
      * kernel == stride > 1 without padding (the patch stem and the 2x2 down-samplings): one GEMM over the features-last pixel
        blocks (ops_rows.linear_nd: gradients as GEMMs too) -- the library's strided weight-gradient kernel for the stem took
        5 ms per call (405 -> 368 ms per 6-step training step at the 512 x 512 bench sizes);
      * the rest (the 3x3 blocks wider than 64 channels at <= 1/4 resolution and the 1x1 convolutions with bias): the library.
        (The 1x1 ones as GEMMs measured slower, and their HIP-graph replays lost the loss to NaN: not used.)"""
    if type(m) is nn.Conv2d and m.dilation == (1, 1) and m.groups == 1 and m.kernel_size[0] == m.kernel_size[1]:
        k = m.kernel_size[0]
        same = m.stride == (1, 1) and m.padding == (k // 2,) * 2
        if m.bias is None and same and OM.conv_nhwc_supported(x, m.weight):
            y = OM.conv_nhwc(x.permute(0, 2, 3, 1), m.weight)              # (B,H,W,64)
            return y.permute(0, 3, 1, 2)                                  # NCHW-shaped view of features-last memory (channels_last)
        if k > 1 and m.stride == (k, k) and m.padding == (0, 0) and x.shape[2] % k == 0 and x.shape[3] % k == 0:
            xl = x.permute(0, 2, 3, 1)
            if xl.shape[-1] > m.in_channels:      # zero-padded rows from build_x (rollout_input_format): the real channels only
                xl = xl[..., : m.in_channels]
            B, H, W, C = xl.shape
            if k > 1:
                xl = xl.reshape(B, H // k, k, W // k, k, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // k, W // k, k * k * C)
            w = m.weight.permute(0, 2, 3, 1).reshape(m.weight.shape[0], k * k * C)
            return _lin(xl, w, m.bias).permute(0, 3, 1, 2)
    if (type(m) is nn.Conv2d and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros" and m.stride == (1, 1)
            and m.kernel_size[0] == m.kernel_size[1] and m.kernel_size[0] in (1, 3) and m.padding == (m.kernel_size[0] // 2,) * 2
            and _native(x) and x.permute(0, 2, 3, 1).is_contiguous() and G.conv_supported(x.permute(0, 2, 3, 1), m.weight)):
        # 3x3 "same" / 1x1 at any width: the implicit-GEMM kernel on the features-last map (no im2col, no layout change)
        return G.conv2d_nhwc(x.permute(0, 2, 3, 1), m.weight, m.bias).permute(0, 3, 1, 2)
    if isinstance(m, nn.Conv2d) and x.shape[1] > m.in_channels:      # zero-padded rows from build_x: the library takes the real channels
        x = x[:, : m.in_channels]
    if type(m) is nn.Conv2d and m.padding_mode == "zeros":
        return OM.library_conv2d(x, R.param_as(m.weight, x.dtype), None if m.bias is None else R.param_as(m.bias, x.dtype), m.stride,
                                 m.padding, m.dilation, m.groups)
    return m._conv_forward(x, m.weight.to(x.dtype), None if m.bias is None else m.bias.to(x.dtype))


def _conv_transpose(m: nn.ConvTranspose2d, x: torch.Tensor) -> torch.Tensor:
    """kernel == stride transposed convolution (bias-free): (rows, cin) @ (cin, s*s*cout), then the s x s outputs of every input
    pixel are interleaved into the up-sampled grid (one copy)."""
    s = m.stride[0]
    xl = x.permute(0, 2, 3, 1)
    B, H, W, cin = xl.shape
    wt = m.weight                                                        # (cin, cout, s, s)
    cout = wt.shape[1]
    wr = wt.permute(2, 3, 1, 0).reshape(s * s * cout, cin)               # Linear weight: (outputs, cin)
    up = _lin(xl, wr).view(B, H, W, s, s, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, H * s, W * s, cout)
    return up.permute(0, 3, 1, 2)


def _nrm(m: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """group / batch / instance norm with fp32 parameters and statistics on a tensor of the activation dtype"""
    if isinstance(m, nn.GroupNorm) and x.is_cuda and ON.supported(x.permute(0, 2, 3, 1)):
        # the stem's GroupNorm (one group over 128 x 128 x 128 values per sample at the yaml's sizes: the library reduces each
        # sample with ONE workgroup, 340 us): per-channel sums + one streaming pass on the instance-norm kernels
        # (handed on NCHW-contiguous, as the library's GroupNorm does: with a channels_last result here MIOpen's immediate-mode
        # heuristics picked other -- CK -- solvers for the 3x3 convolutions downstream and the step took 1 130 ms instead of 570)
        y = ON.group_norm(x.permute(0, 2, 3, 1), m.num_groups, m.weight, m.bias, m.eps).permute(0, 3, 1, 2)
        return y if _native(x) else y.contiguous()      # (no library convolution downstream on the native path: the layout stays)
    if x.dtype in (torch.float32, torch.float64) or isinstance(m, nn.BatchNorm2d):   # the library's batch norm takes bf16 activations with fp32 parameters / statistics
        return m(x)
    return m(x.float()).to(x.dtype)


def _layer_norm(m: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    C = x.shape[-1]
    L.require_cuda(x)
    if (C * x.element_size()) % 16 == 0 and C * x.element_size() <= 1024:
        return R.row_layer_norm(x.reshape(-1, C), m.weight, m.bias, m.eps).view(x.shape)
    return F.layer_norm(x.float(), m.normalized_shape, m.weight, m.bias, m.eps).to(x.dtype)


class ResBlock(nn.Module):
    def __init__(self, cin, cout, norm):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.norm1, self.norm2 = _norm(norm, cout), _norm(norm, cout)
        self.down = cin != cout
        if self.down:
            self.conv3 = nn.Conv2d(cin, cout, 1, bias=False)
            self.norm3 = _norm(norm, cout)

    def forward(self, x, pass_input=False, post=None):
        """``pass_input``: returns (out, x') with x' the input as handed on by this block's last consumer of it (features-last) -- a
        caller that uses x once more (the transformer block's skip) takes x' and its gradient is added inside this block's backward
        launches; x' is None when the block ran on a route without that (the caller then uses its own tensor).
        ``post`` = (m (B, C) fp32 >= 0, factor): the output is multiplied by m[b, c] * factor -- a channel dropout's draw -- inside the
        last normalisation pass where the route has one (ops_gemm.batch_norm_act), by a tensor-library product otherwise."""
        out, xp, applied = self._forward(x, pass_input, post)
        if post is not None and not applied:
            out = out * (post[0] * post[1]).to(out.dtype)[:, :, None, None]
        return (out, xp) if pass_input else out

    def _forward(self, x, pass_input=False, post=None):
        """(out NCHW-shaped, the input handed on or None, whether ``post`` was applied)"""
        # (the norms act on the convolutions' OUTPUT channels: encoder1's 69-channel input is no obstacle)
        cout = self.conv1.weight.shape[0]
        if (isinstance(self.norm1, nn.InstanceNorm2d) and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last)
                and x.dtype in (torch.float32, torch.bfloat16) and cout % 4 == 0 and cout <= 1024):
            # instance-norm blocks on features-last memory (the full-resolution blocks): norm + LeakyReLU (+ residual) as native nodes
            inorm = lambda m, t, slope=1.0, res=None: ON.instance_norm_act(  # noqa: E731
                t.permute(0, 2, 3, 1), m.weight, m.bias, m.eps, slope, None if res is None else res.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            y = inorm(self.norm1, _conv(self.conv1, x).contiguous(memory_format=torch.channels_last), 0.01)
            r = inorm(self.norm3, _conv(self.conv3, x).contiguous(memory_format=torch.channels_last)) if self.down else x
            return inorm(self.norm2, _conv(self.conv2, y).contiguous(memory_format=torch.channels_last), 0.01, r), None, False
        if (isinstance(self.norm1, nn.BatchNorm2d) and _native(x) and x.permute(0, 2, 3, 1).is_contiguous()
                and G.conv_supported(x.permute(0, 2, 3, 1), self.conv1.weight) and cout % 8 == 0 and cout <= 1024):
            # the 128 ... 1024-channel blocks of the transformer stages: implicit-GEMM convolutions whose drain leaves the batch-norm
            # sums, BatchNorm2d + LeakyReLU (+ residual) as one streaming node each way (ops_gemm: csrc/gemm.hip + csrc/inorm.hip)
            # x has two consumers here (conv1 and the residual path) and, with pass_input, one more in the caller: each consumer hands
            # the tensor on to the next (`passthrough`), so the gradients meet inside the backward launches of this chain -- the residual's
            # in the batch-norm backward, the sum in conv1's data gradient -- instead of in element-wise additions of autograd (round 6)
            xl = x.permute(0, 2, 3, 1)
            y1, st1, xp = G.conv2d_nhwc(xl, self.conv1.weight, want_stats=True, passthrough=True)
            a1 = G.batch_norm_act(y1, st1, self.norm1, 0.01)
            y2, st2 = G.conv2d_nhwc(a1, self.conv2.weight, want_stats=True)
            if self.down:
                if pass_input:
                    y3, st3, xp = G.conv2d_nhwc(xp, self.conv3.weight, want_stats=True, passthrough=True)
                else:
                    y3, st3 = G.conv2d_nhwc(xp, self.conv3.weight, want_stats=True)
                rl = G.batch_norm_act(y3, st3, self.norm3, 1.0)
                res_pass = False
            else:
                rl, res_pass = xp, pass_input
            mul, factor = (None, 1.0) if post is None else post
            out = G.batch_norm_act(y2, st2, self.norm2, 0.01, rl, res_passthrough=res_pass, mul=mul, mul_factor=factor)
            if res_pass:
                out, xp = out
            return out.permute(0, 3, 1, 2), (xp if pass_input else None), post is not None
        r = x
        y = F.leaky_relu(_nrm(self.norm1, _conv(self.conv1, x)), 0.01)
        y = _nrm(self.norm2, _conv(self.conv2, y))
        if self.down:
            r = _nrm(self.norm3, _conv(self.conv3, r))
        return F.leaky_relu(y + r, 0.01), None, False


class _SplitQKVV(torch.autograd.Function):
    """q, k, v_ca, v_sa as (B, h, N, d) views of the (B, N, 4, h, d) projection -- as one autograd node.  Through autograd's own view
    nodes each of the four gradients came back as a zero-filled full-size tensor plus a copy, and the four were then summed (four
    fills, four copies and three additions of the whole (B, N, 4C) tensor per block); here the four gradients are copied into one
    buffer."""

    @staticmethod
    def forward(ctx, qkvv):
        ctx.shape = qkvv.shape
        return tuple(qkvv[:, :, i].permute(0, 2, 1, 3) for i in range(4))

    @staticmethod
    def backward(ctx, *grads):
        B, N, _, h, d = ctx.shape
        ref = next(g for g in grads if g is not None)
        out = torch.empty(B, N, 4, h, d, dtype=ref.dtype, device=ref.device)
        for i, g in enumerate(grads):
            if g is None:
                out[:, :, i].zero_()
            else:
                out[:, :, i].copy_(g.permute(0, 2, 1, 3))
        return out


class EPA(nn.Module):
    """Efficient paired attention on the tall-skinny kernels.  With q, k, v_ca, v_sa the (N x d) token matrices of a head:
        G = q^T k,  nq = ||q columns||,  nk likewise          (gram: three reductions over the tokens)
        A = softmax(t1 * G / (nq nk^T))                        (d x d, torch)
        x_ca = v_ca A^T                                        (apply)
        KP = E(k^T), VP = E(v_sa^T)                            (d x p, the token-axis Linear: library GEMM)
        S = softmax(t2 * q (KP / nq))                          (apply -> N x p, row softmax)
        x_sa = S VP^T                                          (apply)
    Normalising q along the tokens scales COLUMN i of q by 1 / nq_i, which is folded into the small matrices (rows of KP, the
    outer product under G): the N x d normalised copies of q and k are never formed."""

    def __init__(self, tokens, hidden, proj, heads, published=False, attn_drop=0.0):
        super().__init__()
        self.heads, self.published = heads, published
        self.temperature = nn.Parameter(torch.ones(heads, 1, 1))
        self.temperature2 = nn.Parameter(torch.ones(heads, 1, 1))
        self.qkvv = nn.Linear(hidden, hidden * 4, bias=False)
        self.E = nn.Linear(tokens, proj)
        if published:
            # the published module list: ``self.E = self.F = nn.Linear(..)`` -- ONE Linear under two names -- and two attention dropouts
            # (identity at p = 0).  The alias is NOT registered as a second sub-module here: torch.func.functional_call (the rollout's
            # per-step parameter stand-ins, trainer.RolloutParamProxies) swaps tensors per NAME and leaves a twice-registered module
            # holding the stand-in afterwards.  The state dict carries the published keys all the same: ``F.*`` is added on save and
            # accepted on load by the two hooks below; ``self.F`` is a property.
            self.attn_drop, self.attn_drop_2 = nn.Dropout(attn_drop), nn.Dropout(attn_drop)
            self._register_state_dict_hook(EPA._add_f_keys)
            self._register_load_state_dict_pre_hook(EPA._take_f_keys)
        self.out_proj = nn.Linear(hidden, hidden // 2)
        self.out_proj2 = nn.Linear(hidden, hidden // 2)

    @property
    def F(self):
        """the published code's second name of the token-axis Linear (``self.E = self.F = nn.Linear(input_size, proj_size)``)"""
        return self.E

    @staticmethod
    def _add_f_keys(module, state_dict, prefix, local_metadata):
        for k in ("weight", "bias"):
            if prefix + "E." + k in state_dict:
                state_dict[prefix + "F." + k] = state_dict[prefix + "E." + k]

    @staticmethod
    def _take_f_keys(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for k in ("weight", "bias"):
            f = state_dict.pop(prefix + "F." + k, None)
            if f is not None and prefix + "E." + k not in state_dict:
                state_dict[prefix + "E." + k] = f       # a checkpoint that kept only the second name

    def _merge_sa(self, x_sa, B, N, C):
        """(B, h, N, d) -> (B, N, C): head-major per token (restated block: a view of the kernels' token-major output), or the published
        code's ``permute(0, 3, 1, 2).reshape(B, N, C)`` -- the (B, d, h, N) order read as (N, C): one gather copy"""
        if self.published:
            if TS.merge_published_ok(x_sa) and L.diag_switch("P4C_EPA_LIB_MERGE") != "1":
                return TS.merge_published(x_sa)      # (a tiled transpose each way instead of the tensor library's strided gather)
            return x_sa.permute(0, 3, 1, 2).reshape(B, N, C)
        return x_sa.permute(0, 2, 1, 3).reshape(B, N, C)

    def _project_out(self, x_sa, x_ca, res, gamma):
        """cat(out_proj(x_sa), out_proj2(x_ca)) -- or, given the block's residual t and layer scale gamma, t + gamma * that cat with
        gamma folded into the two weights and both halves written by the GEMMs' epilogues (no cat, no scale, no add launch)."""
        if res is None or not (_native(x_sa) and G.supported(x_sa, self.out_proj.weight) and self.out_proj.weight.shape[0] % 8 == 0):
            y = torch.cat([_linear(self.out_proj, x_sa), _linear(self.out_proj2, x_ca)], dim=-1)
            return y if res is None else res + R.param_as(gamma, y.dtype) * y
        h = self.out_proj.weight.shape[0]
        if L.diag_switch("P4C_UNETRPP_GAMMA_MUL") == "1":      # the round's first form: gamma * W, gamma * b as torch products
            g1, g2 = gamma[:h], gamma[h:]
            return G.cat_linear_res(x_sa, g1.unsqueeze(1) * self.out_proj.weight, g1 * self.out_proj.bias,
                                    x_ca, g2.unsqueeze(1) * self.out_proj2.weight, g2 * self.out_proj2.bias, res)
        return G.cat_linear_res(x_sa, self.out_proj.weight, self.out_proj.bias, x_ca, self.out_proj2.weight, self.out_proj2.bias, res,
                                gamma=gamma)

    def forward(self, x, res=None, gamma=None):
        """res / gamma given: returns res + gamma * EPA(x) (the transformer block's residual update)"""
        B, N, C = x.shape
        h, d = self.heads, C // self.heads
        qkvv = _linear(self.qkvv, x).view(B, N, 4, h, d)
        if x.is_cuda and L.diag_switch("P4C_EPA_CORE") != "0" and TS.epa_core_ok(qkvv, self.E.out_features):
            # the attention between the projections as one node: its backward writes dq / dk / dv straight into the gradient of qkvv
            x_sa, x_ca = TS.epa_core(qkvv, self.E.weight, self.E.bias, self.temperature, self.temperature2)
            return self._project_out(self._merge_sa(x_sa, B, N, C), x_ca.permute(0, 2, 1, 3).reshape(B, N, C), res, gamma)
        q, k, v_ca, v_sa = _SplitQKVV.apply(qkvv)                                          # (B,h,N,d) views, nothing copied
        if d % 4:
            raise L.P4CError(f"UNetRPP: head width {d} must be a multiple of 4")
        eps = 1e-12                                                                        # F.normalize's clamp
        fused_small = d <= 64 and self.E.out_features <= 64 and x.is_cuda and L.diag_switch("P4C_NO_EPA_SMALL") != "1"
        if fused_small and x.dtype == torch.bfloat16 and L.diag_switch("P4C_NO_GRAM_NORMS") != "1":
            G, Gq, Gk = TS.gram_norms(q, k)          # q^T k and the squared column norms of q and k from one pass over q and k
        else:
            G, Gq, Gk = TS.gram(q, k), TS.gram(q, q), TS.gram(k, k)
        if not fused_small:
            nq = torch.diagonal(Gq, dim1=-2, dim2=-1).clamp_min(0).sqrt().clamp_min(eps)    # (B,h,d)
            nk = torch.diagonal(Gk, dim1=-2, dim2=-1).clamp_min(0).sqrt().clamp_min(eps)
            A = (G / (nq.unsqueeze(-1) * nk.unsqueeze(-2)) * self.temperature).softmax(dim=-1)
            x_ca = TS.apply(v_ca, A.transpose(-1, -2)).permute(0, 2, 1, 3).reshape(B, N, C)
        # token-axis projection (shared weights): (B, C, N) @ (N, p) for k and v_sa at once -- a library GEMM
        W, bias = R.param_as(self.E.weight, x.dtype), self.E.bias.float()
        kv = torch.stack([k.permute(0, 2, 1, 3).reshape(B, N, C), v_sa.permute(0, 2, 1, 3).reshape(B, N, C)], dim=1)   # (B,2,N,C)
        proj = R.add_bias((kv.transpose(-1, -2) @ W.t()).float(), bias)                              # (B,2,C,p); bias gradient as a GEMM
        KP, VP = proj[:, 0].view(B, h, d, -1), proj[:, 1].view(B, h, d, -1)
        if fused_small:
            # the small matrices (norms, channel-attention softmax, scaled projection) in one native launch each way: ops_ts.epa_small
            At, Mq = TS.epa_small(G, Gq, Gk, KP, self.temperature, self.temperature2)
            x_ca = TS.apply(v_ca, At).permute(0, 2, 1, 3).reshape(B, N, C)
        else:
            Mq = KP / nq.unsqueeze(-1) * self.temperature2
        if x.is_cuda and L.diag_switch("P4C_NO_EPA_SPATIAL") != "1" and TS.spatial_fused_ok(q, Mq.shape[-1]):
            # softmax (and its adjoint) in the epilogue of the apply that produces its argument: ops_ts.epa_spatial
            x_sa = self._merge_sa(TS.epa_spatial(q, Mq, VP.transpose(-1, -2)), B, N, C)
        else:
            S = TS.apply(q, Mq).softmax(dim=-1)                                                      # (B,h,N,p), token-major memory
            x_sa = self._merge_sa(TS.apply(S, VP.transpose(-1, -2)), B, N, C)
        return self._project_out(x_sa, x_ca, res, gamma)


class TransformerBlock(nn.Module):
    def __init__(self, tokens, hidden, proj, heads, published=False, conv8_dropout=0.0, attn_drop=0.0):
        super().__init__()
        self.norm = nn.LayerNorm(hidden)
        self.gamma = nn.Parameter(1e-6 * torch.ones(hidden))
        self.epa_block = EPA(tokens, hidden, proj, heads, published, attn_drop)
        self.conv51 = ResBlock(hidden, hidden, "batch")
        # published: Sequential(Dropout(0.1, False), Conv) -- state-dict keys conv8.1.*
        self.conv8 = nn.Sequential(nn.Dropout2d(conv8_dropout, False), nn.Conv2d(hidden, hidden, 1)) if published else nn.Conv2d(hidden, hidden, 1)
        self.published = published
        self.pos_embed = nn.Parameter(torch.zeros(1, tokens, hidden))

    def _conv8_in(self, r_nchw):
        """the channel dropout in front of conv8 (published block, training mode, p > 0) on an NCHW-shaped tensor"""
        if self.published and self.training and self.conv8[0].p > 0:
            return self.conv8[0](r_nchw)
        return r_nchw

    def forward(self, x):
        B, C, H, W = x.shape
        conv8 = self.conv8[1] if self.published else self.conv8
        xl = x.permute(0, 2, 3, 1)
        if _native(x) and xl.is_contiguous() and R.add_layer_norm_supported(xl) and C % 16 == 0:
            # features-last all the way: (x + pos) and its LayerNorm from one read, the EPA's two output projections write
            # t + gamma * (...) from their epilogues, the 1x1 conv8 adds its bias and the skip in its epilogue
            t, ln = R.add_layer_norm(xl.reshape(B, H * W, C), self.pos_embed, self.norm.weight, self.norm.bias, self.norm.eps)
            t = self.epa_block(ln, res=t, gamma=self.gamma)
            skip = t.reshape(B, H, W, C)
            if isinstance(self.conv51, ResBlock):
                # the channel dropout in front of conv8 (published block, training mode): the draw per (sample, channel) as
                # F.dropout2d makes it, applied by conv51's last normalisation pass (and its backward passes) instead of by a
                # full-size product each way (round 6)
                post = None
                if self.published and self.training and 0 < self.conv8[0].p < 1 and L.diag_switch("P4C_UNETRPP_LIB_DROPOUT") != "1":
                    keep = 1.0 - self.conv8[0].p
                    post = (torch.empty(B, C, dtype=torch.float32, device=x.device).bernoulli_(keep), 1.0 / keep)
                r, handed = self.conv51(skip.permute(0, 3, 1, 2), pass_input=True, post=post)   # (skip comes back from its consumers)
                skip = skip if handed is None else handed
                r = (r if post is not None else self._conv8_in(r)).permute(0, 2, 3, 1)
            else:
                r = self._conv8_in(self.conv51(skip.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
            if G.conv_supported(r, conv8.weight) and r.is_contiguous():
                return G.conv2d_nhwc(r, conv8.weight, conv8.bias, res=skip).permute(0, 3, 1, 2)
            return (skip + _conv(conv8, r.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        t = x.reshape(B, C, H * W).permute(0, 2, 1) + R.param_as(self.pos_embed, x.dtype)
        t = t + R.param_as(self.gamma, x.dtype) * self.epa_block(_layer_norm(self.norm, t.contiguous()))
        skip = t.reshape(B, H, W, C).permute(0, 3, 1, 2)
        return skip + _conv(conv8, self._conv8_in(self.conv51(skip)))


class _FeaturesLast(torch.autograd.Function):
    """An NCHW-contiguous tensor as an NCHW-shaped view of features-last memory (one copy), its gradient handed back NCHW-contiguous
    (one copy) whatever layout it arrives in."""

    @staticmethod
    def forward(ctx, x):
        return x.contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


class UpBlock(nn.Module):
    def __init__(self, cin, cout, scale, tokens, proj, heads, depth, conv_decoder, linear, norm, block_kw=None):
        super().__init__()
        self.scale, self.linear = scale, linear
        self.up_conv = nn.Conv2d(cin, cout, 1) if linear else nn.ConvTranspose2d(cin, cout, scale, stride=scale, bias=False)
        if conv_decoder:
            self.decoder_block = nn.ModuleList([ResBlock(cout, cout, norm)])
        else:
            self.decoder_block = nn.ModuleList([nn.Sequential(*[TransformerBlock(tokens, cout, proj, heads, **(block_kw or {}))
                                                                for _ in range(depth)])])

    def forward(self, x, skip):
        if (self.linear and _native(x) and x.shape[1] % 8 == 0 and self.up_conv.out_channels % 8 == 0 and float(self.scale).is_integer()
                and x.permute(0, 2, 3, 1).is_contiguous()):
            # 1x1 convolution on the small grid (it commutes with the interpolation, see below), then up-sample + skip in one native pass
            # (csrc/resize.hip: features-last both ways, gather-form backward -- no atomics, no layout copies)
            y = _conv(self.up_conv, x).permute(0, 2, 3, 1)
            return self.decoder_block[0](G.upsample_add(y, skip.permute(0, 2, 3, 1), int(self.scale)).permute(0, 3, 1, 2))
        if self.linear:
            # mfai: up_conv(interpolate(x)).  A 1x1 convolution (per pixel, across channels) and the bilinear interpolation (per
            # channel, across pixels, weights summing to 1 -- the bias passes through) commute: the convolution runs on the small
            # grid (1 / scale^2 of the rows) and the interpolation on the convolution's fewer channels; same function, one rounding
            # moved (decoder2 of the yaml: 128 -> 64 channels at 128 x 128 instead of 512 x 512)
            x = F.interpolate(_conv(self.up_conv, x), scale_factor=self.scale, mode="bilinear", align_corners=False)
        else:
            x = _conv_transpose(self.up_conv, x)
        if self.linear and isinstance(self.decoder_block[0], ResBlock) and x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
            # decoder2: the interpolated x is NCHW-contiguous and so was the sum -- which sent the full-resolution residual block's three
            # instance norms through the library in fp32 (two casts each) and its convolutions through a layout copy.  _FeaturesLast
            # hands the block an NCHW-shaped view of features-last memory and the interpolation an NCHW-contiguous gradient (its
            # backward on a channels_last gradient is an atomics kernel: 7.9 s per step, not reproducible)
            return self.decoder_block[0](skip + _FeaturesLast.apply(x))
        return self.decoder_block[0](x + skip)


class UNetRPPMI355X(ModelABC, nn.Module):
    settings_kls = UNetRPPSettings
    onnx_supported = False
    supported_num_spatial_dims = (2,)
    num_spatial_dims = 2
    features_last = True
    model_type = ModelType.VISION_TRANSFORMER
    register = True
    is_native_hip = True   # common_step: the precision is the model's (activation_dtype), no torch.autocast around it

    def __init__(self, in_channels: int, out_channels: int, input_shape: Tuple[int, int] = None,
                 settings: UNetRPPSettings = UNetRPPSettings(), *args, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.input_shape = in_channels, out_channels, input_shape
        self._settings = s = settings
        if s.spatial_dims != 2 or s.conv_op != "Conv2d" or s.do_ds or s.dropout_rate != 0.0 or not s.add_skip_connections:
            raise NotImplementedError("UNetRPPMI355X: 2-D, no deep supervision, no dropout, with skip connections")
        if s.hidden_size % 16 or len(s.depths) != 4 or len(s.encoder_proj_sizes) != 4:
            raise NotImplementedError("UNetRPPMI355X: hidden_size must be a multiple of 16, four stages")
        H, W = input_shape
        r = s.downsampling_rate
        if H % (8 * r) or W % (8 * r):
            raise L.P4CError(f"UNetRPPMI355X: grid {H}x{W} must be a multiple of {8 * r}")
        fs = s.hidden_size // 16
        dims = [fs * 2, fs * 4, fs * 8, fs * 16]
        tokens = [(H // (r * 2**i)) * (W // (r * 2**i)) for i in range(4)]
        self.act_dtype = torch.bfloat16 if s.activation_dtype == "bf16" else torch.float32
        self.downsample_layers = nn.ModuleList()
        g0 = in_channels if dims[0] % in_channels == 0 else 1
        self.downsample_layers.append(nn.Sequential(nn.Conv2d(in_channels, dims[0], r, stride=r, bias=False), nn.GroupNorm(g0, dims[0])))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(nn.Conv2d(dims[i], dims[i + 1], 2, stride=2, bias=False), nn.GroupNorm(dims[i], dims[i + 1])))
        if not 0.0 <= s.conv8_dropout < 1.0:
            raise ValueError("UNetRPPMI355X: conv8_dropout must be in [0, 1)")
        bkw = dict(published=bool(s.published_block), conv8_dropout=float(s.conv8_dropout), attn_drop=float(s.dropout_rate))
        self.stages = nn.ModuleList([nn.Sequential(*[TransformerBlock(tokens[i], dims[i], s.encoder_proj_sizes[i], s.num_heads_encoder, **bkw)
                                                     for _ in range(s.depths[i])]) for i in range(4)])
        up = (s.decoder_proj_size, s.num_heads_decoder, 3)
        self.encoder1 = ResBlock(in_channels, fs, s.norm_name)
        self.decoder5 = UpBlock(dims[3], dims[2], 2, tokens[2], *up, False, s.linear_upsampling, s.norm_name, bkw)
        self.decoder4 = UpBlock(dims[2], dims[1], 2, tokens[1], *up, False, s.linear_upsampling, s.norm_name, bkw)
        self.decoder3 = UpBlock(dims[1], dims[0], 2, tokens[0], *up, False, s.linear_upsampling, s.norm_name, bkw)
        self.decoder2 = UpBlock(dims[0], fs, r, H * W, *up, True, s.linear_upsampling, s.norm_name, bkw)
        self.out1 = nn.Conv2d(fs, out_channels, 1)
        self.timed_entry_points = ("p4c_ts_gram", "p4c_ts_apply", "p4c_row_add_layernorm_fwd", "p4c_row_add_layernorm_bwd", "p4c_gemm_nt", "p4c_gemm_tn")
        self.roofline_from_entry_points = True   # bench.py: time every call of the native entry points above
        self.prefers_hip_graph = True            # ~10^3-10^4 launches per training step: replay them from a HIP graph (trainer.GraphedTrainingStep)
        self.check_required_attributes()

    @property
    def settings(self):
        return self._settings

    def roofline(self, ktimes, B, H, W):
        """bench.py: achieved HBM rate of the native entry point that takes the most time (algorithmic bytes stated by the wrappers
        in ops_ts / ops_rows next to each call, over HIP-event durations of every call)."""
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

    rollout_padded_output = False   # set by the rollout around its calls: rows wider than out_channels are welcome
    rollout_param_proxies = True    # the rollout may run each AR step on stand-ins of the parameters (trainer.RolloutParamProxies)

    @property
    def rollout_input_format(self):
        """(dtype, channel count) the rollout's build_x should emit for this model (see SwinUNetRMI355X.rollout_input_format): bf16 rows
        zero-padded to the 32-channel multiple the full-resolution convolutions of encoder1 run on; the stem reads the real channels."""
        if self.act_dtype != torch.bfloat16 or self.in_channels > 96 or L.diag_switch("P4C_NO_ROLLOUT_FORMAT") == "1":
            return None
        return torch.bfloat16, (self.in_channels + 31) // 32 * 32

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B,H,W,in_channels) -> (B,H,W,out_channels)."""
        L.require_cuda(x)
        out_dtype = x.dtype
        fmt = self.rollout_input_format
        if x.shape[-1] != self.in_channels and (fmt is None or x.shape[-1] != fmt[1] or x.dtype != fmt[0]):
            raise L.P4CError(f"UNetRPPMI355X: {x.shape[-1]} input channels, expected {self.in_channels}")
        x = x.to(self.act_dtype).contiguous().permute(0, 3, 1, 2)   # NCHW-shaped view of features-last memory (channels_last)
        hidden, h = [], x
        for i in range(4):
            ds = self.downsample_layers[i]
            h = self.stages[i](_nrm(ds[1], _conv(ds[0], h)))
            hidden.append(h)
        conv_block = self.encoder1(x)
        dec3 = self.decoder5(hidden[3], hidden[2])
        dec2 = self.decoder4(dec3, hidden[1])
        dec1 = self.decoder3(dec2, hidden[0])
        out = self.decoder2(dec1, conv_block)
        # (the 1x1 output convolution has a bias and goes to the library: NCHW-contiguous input -- on an NCHW-shaped view of features-last
        # memory the library's deterministic solver for this shape took 1.1 s per step)
        o1 = self.out1
        if (out.dtype == torch.bfloat16 and o1.kernel_size == (1, 1) and out.permute(0, 2, 3, 1).is_contiguous() and o1.in_channels % 8 == 0
                and L.diag_switch("P4C_NO_OUT1_ROWS") != "1"):
            # the 1x1 output convolution (with bias) as a GEMM over the features-last pixel rows: no NCHW copy of the 64-channel map in
            # front of the library, no permuted result; inside the rollout the 60 outputs are padded to 64 (zero weight rows) and go
            # out as they are -- the row-GEMM kernel takes multiples of 8, the state update reads the first out_channels features
            O = o1.out_channels
            w2, b2 = o1.weight.view(O, o1.in_channels), o1.bias
            if self.rollout_padded_output and O % 8:
                pad = (-O) % 8
                w2, b2 = F.pad(w2, (0, 0, 0, pad)), None if b2 is None else F.pad(b2, (0, pad))
            y = R.linear_nd(out.permute(0, 2, 3, 1), w2, b2)
        else:
            y = _conv(o1, out.contiguous()).permute(0, 2, 3, 1)
        return y if y.dtype == out_dtype or not out_dtype.is_floating_point else y.to(out_dtype)
