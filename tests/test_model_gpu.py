"""
GPU parity of the model kernels (conv on the fp32 matrix cores, norms, pool, up-sample-and-sum, the
HalfUNet plan) against plain PyTorch fp32 references on the CPU (oracle/halfunet.py for the network).
Tolerance: 1e-4 relative (north-star bar for fp32 forward outputs); gradients 1e-3.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.parametrize("CI,CIreal,ks,H,W", [(64, 64, 3, 16, 32), (96, 69, 3, 20, 40), (32, 10, 3, 8, 8), (64, 64, 1, 12, 36),
                                                (64, 60, 3, 4, 4), (96, 96, 1, 8, 64)])
def test_conv_fwd_matches_torch(gpu_device, CI, CIreal, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(CI + ks + H)
    B, CO = 2, 64
    x = torch.randn(B, H, W, CI, generator=g)
    x[..., CIreal:] = 0
    w = torch.randn(CO, CIreal, ks, ks, generator=g) * 0.1
    scale = torch.rand(B, CI, generator=g) + 0.5
    shift = torch.randn(B, CI, generator=g) * 0.3
    for transform in (False, True):
        xin = x
        if transform:
            xin = torch.relu(x * scale[:, None, None, :] + shift[:, None, None, :])
            xin = xin.clone()
            xin[..., CIreal:] = xin[..., CIreal:]  # padded channels: weights are zero there
        ref = Fn.conv2d(xin[..., :CIreal].permute(0, 3, 1, 2), w, padding=ks // 2).permute(0, 2, 3, 1)
        wp = om.prep_weights(w.to(gpu_device), False, 64, CI)
        out, stats = om.conv_fwd(x.to(gpu_device), wp, ks, in_scale=scale.to(gpu_device) if transform else None,
                                 in_shift=shift.to(gpu_device) if transform else None, in_relu=transform, want_stats=True)
        assert rel_err(out, ref) < 1e-5
        s = stats.sum(0).cpu()
        np.testing.assert_allclose(s[0].numpy(), ref.sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(s[1].numpy(), (ref**2).sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("CIreal,ks,H,W", [(64, 3, 16, 32), (60, 3, 12, 20), (64, 1, 8, 40)])
def test_conv_data_grad_matches_autograd(gpu_device, CIreal, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(7)
    B, CO = 2, 64
    x = torch.randn(B, CIreal, H, W, generator=g, requires_grad=True)
    w = torch.randn(CO, CIreal, ks, ks, generator=g) * 0.1
    dout = torch.randn(B, H, W, CO, generator=g)
    Fn.conv2d(x, w, padding=ks // 2).backward(dout.permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)
    wp = om.prep_weights(w.to(gpu_device), True, 64, 64)
    got = om.conv_fwd(dout.to(gpu_device), wp, ks)
    assert rel_err(got[..., :CIreal], ref) < 1e-5
    assert float(got[..., CIreal:].abs().sum()) == 0.0


@pytest.mark.parametrize("CI,CIreal,CO,ks,H,W", [(64, 64, 64, 3, 16, 32), (96, 69, 64, 3, 24, 40), (32, 10, 64, 3, 8, 8),
                                                   (64, 64, 60, 1, 12, 36)])
def test_conv_weight_grad_matches_autograd(gpu_device, CI, CIreal, CO, ks, H, W):
    from py4cast_amd import ops_model as om

    g = torch.Generator().manual_seed(11)
    B = 2
    x = torch.randn(B, H, W, CI, generator=g)
    x[..., CIreal:] = 0
    scale = torch.rand(B, CI, generator=g) + 0.5
    shift = torch.randn(B, CI, generator=g) * 0.3
    dout = torch.randn(B, H, W, 64, generator=g)
    dout[..., CO:] = 0
    w = torch.zeros(CO, CIreal, ks, ks, requires_grad=True)
    xin = torch.relu(x * scale[:, None, None, :] + shift[:, None, None, :])
    Fn.conv2d(xin[..., :CIreal].permute(0, 3, 1, 2), w, padding=ks // 2).backward(dout[..., :CO].permute(0, 3, 1, 2))
    grad = torch.ones(CO, CIreal, ks, ks, device=gpu_device)  # accumulation semantics: += on top of ones
    om.conv_wgrad(x.to(gpu_device), dout.to(gpu_device), ks, CO, CIreal, grad, scale.to(gpu_device), shift.to(gpu_device), True)
    assert rel_err(grad - 1.0, w.grad) < 2e-5


def _make_pair(cin, cout, norm, device, seed=0):
    from oracle.halfunet import HalfUNetRef
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    torch.manual_seed(seed)
    ref = HalfUNetRef(cin, cout, norm=norm)
    with torch.no_grad():  # non-trivial affine parameters
        for m in ref.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.GroupNorm)):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    model = HalfUNetMI355X(cin, cout, (32, 32), HalfUNetSettings(norm=norm))
    missing = model.load_state_dict(ref.state_dict(), strict=True)
    return ref, model.to(device)


@pytest.mark.parametrize("norm,cin,cout,H,W", [("batch", 69, 60, 32, 32), ("group", 46, 21, 48, 32), ("batch", 10, 1, 64, 64)])
def test_halfunet_forward_backward_match_oracle(gpu_device, norm, cin, cout, H, W):
    ref, model = _make_pair(cin, cout, norm, gpu_device)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, H, W, cin, generator=g)
    gy = torch.randn(2, H, W, cout, generator=g)
    # float64 oracle: deep BatchNorm levels (4x4 maps) amplify fp32 rounding in the gradients, so fp32-vs-fp32
    # comparisons measure the reference's own noise as much as ours
    import copy
    ref32 = ref
    ref = copy.deepcopy(ref32).double()
    xr = x.double().requires_grad_(True)
    ref.train()
    yr = ref(xr.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    (yr * gy.double()).sum().backward()
    xg = x.to(gpu_device).requires_grad_(True)
    model.train()
    yg = model(xg)
    (yg * gy.to(gpu_device)).sum().backward()
    assert yg.shape == yr.shape
    assert rel_err(yg, yr) < 1e-4
    nchk = min(cin, 64)
    assert rel_err(xg.grad[..., :nchk], xr.grad[..., :nchk]) < 1e-3
    sd = dict(ref.named_parameters())
    for name, p in model.named_parameters():
        assert rel_err(p.grad, sd[name].grad) < 1e-3, name
    if norm == "batch":  # running statistics follow torch's update rule
        rb = dict(ref.named_buffers())
        for name, buf in model.named_buffers():
            if buf.dtype.is_floating_point:
                np.testing.assert_allclose(buf.cpu().numpy(), rb[name].float().numpy(), rtol=1e-4, atol=1e-5, err_msg=name)
            else:
                assert int(buf) == int(rb[name])
        # eval mode uses the running statistics
        ref.eval(); model.eval()
        with torch.no_grad():
            ye = model(x.to(gpu_device))
            yre = ref(x.double().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        assert rel_err(ye, yre) < 1e-4


def test_halfunet_rejects_unsupported_settings():
    from py4cast_amd.halfunet import HalfUNetMI355X, HalfUNetSettings

    with pytest.raises(NotImplementedError):
        HalfUNetMI355X(10, 1, (64, 64), HalfUNetSettings(use_ghost=True))
    with pytest.raises(NotImplementedError):
        HalfUNetMI355X(10, 1, (64, 64), HalfUNetSettings(num_filters=32))


def test_training_step_with_halfunet_matches_oracle(gpu_device):
    """AutoRegressiveLightning + HalfUNet on HIP kernels: loss and BPTT gradients vs the CPU oracle."""
    from helpers import make_batch, make_dataset_info, synthetic_case
    from oracle import losses as olosses
    from oracle import rollout as orollout
    from oracle.halfunet import HalfUNetRef
    from py4cast_amd.lightning import AutoRegressiveLightning

    case = synthetic_case(seed=5, B=2, T=3, H=32, W=32, F=12, Ff=5, Fs=4, border=2)
    info = make_dataset_info(case, 5)
    torch.manual_seed(0)
    lm = AutoRegressiveLightning(
        {}, info, None, num_pred_steps_train=3, batch_size=2, model_name="HalfUNet",
        losses=[{"class": "WeightedLoss", "weight": 1.0, "params": {"loss": "MSELoss", "reduction": "none"}}],
        training_strategy="scaled_ar",
    )
    ref = HalfUNetRef(12 + 4 + 5, 12)
    ref.load_state_dict(lm.model.state_dict())
    lm = lm.to(gpu_device)
    lm.train()
    loss = lm.training_step(make_batch(case, gpu_device), 0)
    loss.backward()
    B = 2
    statics = case["statics"].unsqueeze(0).expand(B, *case["statics"].shape)
    interior = 1.0 - case["border_mask"]
    ref.train()
    pred = orollout.rollout(ref, case["inputs"], case["forcing"], case["outputs"], statics, case["border_mask"], interior,
                            case["diff_std"], case["diff_mean"], "scaled_ar", 1, False, "train", features_second=True)
    wts = olosses.weighted_loss_weights(case["state_weight"], case["diff_std"], "mse")
    lref = olosses.training_loss(pred, case["outputs"], False, [("WeightedLoss", 1.0, dict(weights=wts, interior_mask=interior, kind="mse"))])
    lref.backward()
    assert abs(loss.item() - lref.item()) / abs(lref.item()) < 1e-4
    sd = dict(ref.named_parameters())
    for name, p in lm.model.named_parameters():
        assert rel_err(p.grad, sd[name].grad) < 5e-3, name
