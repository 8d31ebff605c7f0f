"""Round 6: the small kernels that took the tensor library's place around the tall-skinny products of UNETR++'s EPA block
(csrc/tallskinny.hip: p4c_ts_reduce_splits, p4c_ts_reduce_transpose, p4c_ts_colsums, p4c_ts_merge_published) against torch on the same
operands.  Sums: fp32 in another order (1e-5); the merge is data movement: bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("A,S,R,segs,bias_len", [(2, 64, 16, (64, 8, 8), 0), (2, 16, 2, (128 * 64,), 64), (1, 1, 3, (32, 4), 0), (3, 7, 5, (48,), 16),
                                                 (2, 37, 4, (1024, 32, 32), 0)])
def test_reduce_splits_matches_torch(gpu_device, A, S, R, segs, bias_len):
    from py4cast_amd import ops_ts as TS

    torch.manual_seed(A * 100 + S)
    E = sum(segs)
    part = torch.randn(A, S, R, E, device=gpu_device)
    bias = torch.randn(bias_len, device=gpu_device) if bias_len else None
    outs = [torch.full((A, R, n), float("nan"), device=gpu_device) for n in segs]
    TS.reduce_splits(part, outs, bias=bias)
    ref = part.double().sum(dim=1)
    if bias is not None:
        ref = ref + bias.double().repeat(E // bias_len)
    o = 0
    for out, n in zip(outs, segs):
        torch.testing.assert_close(out.double(), ref[..., o:o + n], rtol=1e-5, atol=1e-5)
        o += n
    again = [torch.empty_like(t) for t in outs]
    TS.reduce_splits(part, again, bias=bias)
    assert all(torch.equal(a, b) for a, b in zip(outs, again))          # fixed order: reruns reproduce every bit
    acc = [t.clone() for t in outs]
    TS.reduce_splits(part, acc, bias=bias, accumulate=True)
    for a, b in zip(acc, outs):
        torch.testing.assert_close(a, 2 * b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("S,R,E", [(4, 16384, 64), (4, 4096, 32), (2, 100, 8), (1, 33, 64)])
def test_reduce_transpose_matches_torch(gpu_device, S, R, E):
    from py4cast_amd import _lib as L

    torch.manual_seed(S + R)
    part = torch.randn(S, R, E, device=gpu_device)
    out = torch.full((E, R), float("nan"), device=gpu_device)
    L.call("p4c_ts_reduce_transpose", L.ptr(part), S, R, E, L.ptr(out), 0, L.stream(gpu_device))
    ref = part.double().sum(dim=0).t()
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    L.call("p4c_ts_reduce_transpose", L.ptr(part), S, R, E, L.ptr(out), 1, L.stream(gpu_device))
    torch.testing.assert_close(out.double(), 2 * ref, rtol=1e-5, atol=1e-5)


def test_colsums_matches_torch(gpu_device):
    from py4cast_amd import ops_ts as TS

    torch.manual_seed(5)
    xs = [torch.randn(r, c, device=gpu_device) for r, c in ((2, 16), (2, 16), (1024, 64), (77, 5))]
    outs = [torch.full((x.shape[1],), float("nan"), device=gpu_device) for x in xs]
    TS.colsums(list(zip(xs, outs)))
    for x, o in zip(xs, outs):
        torch.testing.assert_close(o.double(), x.double().sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,H,N,d", [(2, 16, 16384, 8), (2, 4, 1024, 32), (1, 3, 72, 8), (2, 4, 256, 64), (1, 1, 64, 8)])
def test_published_merge_is_the_permute_reshape_of_the_published_code(gpu_device, B, H, N, d):
    from py4cast_amd import ops_ts as TS

    torch.manual_seed(N + d)
    C = H * d
    tok = torch.randn(B, N, H, d, device=gpu_device).to(torch.bfloat16)
    x_sa = tok.permute(0, 2, 1, 3).detach().requires_grad_(True)            # the (B, heads, N, d) view the apply kernels hand on
    assert TS.merge_published_ok(x_sa)
    y = TS.merge_published(x_sa)
    ref_in = tok.permute(0, 2, 1, 3).detach().requires_grad_(True)
    ref = ref_in.permute(0, 3, 1, 2).reshape(B, N, C)                        # mfai v5.0.1 UNetRPP, EPA.forward
    assert torch.equal(y, ref)
    w = torch.randn(B, N, C, device=gpu_device).to(torch.bfloat16)
    y.backward(w)
    ref.backward(w)
    assert torch.equal(x_sa.grad, ref_in.grad)
    assert x_sa.grad.stride(3) == 1 or x_sa.grad.shape[3] == 1              # handed back token-major: the block's backward reads it in place


def test_add_layer_norm_table_gradient_goes_into_its_grad_buffer(gpu_device):
    """ops_rows.add_layer_norm: with a gradient buffer on the table (position embedding) its gradient -- the sum of dt over the samples --
    is ADDED into that buffer by one native pass (p4c_sum_leading); without one it comes back through autograd.  Same numbers."""
    from py4cast_amd import ops_rows as R

    torch.manual_seed(3)
    B, N, C = 2, 1024, 128
    x = torch.randn(B, N, C, device=gpu_device).to(torch.bfloat16)
    pos = torch.nn.Parameter(torch.randn(1, N, C, device=gpu_device) * 0.1)
    gamma = torch.nn.Parameter(torch.rand(C, device=gpu_device) + 0.5)
    beta = torch.nn.Parameter(torch.randn(C, device=gpu_device) * 0.1)
    w1 = torch.randn(B, N, C, device=gpu_device).to(torch.bfloat16)
    w2 = torch.randn(B, N, C, device=gpu_device).to(torch.bfloat16)

    def run():
        t, ln = R.add_layer_norm(x, pos, gamma, beta, 1e-5)
        ((t * w1).float().sum() + (ln * w2).float().sum()).backward()

    run()                                   # no buffers yet: through autograd
    ref = pos.grad.clone()
    assert float(ref.abs().max()) > 0
    run()                                   # buffers exist: added in place
    torch.testing.assert_close(pos.grad, 2 * ref, rtol=1e-6, atol=1e-6)
    # and the reference itself against the tensor library on the same bf16 dt
    xr = x.detach().clone().requires_grad_(True)
    t = xr + pos.detach().to(torch.bfloat16)
    ln = torch.nn.functional.layer_norm(t.float(), (C,), gamma.detach(), beta.detach(), 1e-5).to(torch.bfloat16)
    ((t * w1).float().sum() + (ln * w2).float().sum()).backward()
    cos = torch.nn.functional.cosine_similarity(ref.flatten(), xr.grad.float().sum(0).flatten(), dim=0)
    assert float(cos) > 0.999
