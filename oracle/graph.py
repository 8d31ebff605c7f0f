"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (the GNN lives in mfai v5.0.1, absent here).

Torch-native restatement of the edge passes of an InteractionNet layer (GraphLAM / HiLAM, selected by
config/CLI/model/graphlam.yaml:19-26: hidden_dims 64, hidden_layers 1, processor_layers 4, mesh_aggr sum), written the way
neural-lam / mfai run them: index_select + concat feeding the edge MLP's first Linear, index_add_ for the aggregation.
"""

import torch
import torch.nn.functional as F

ACTS = {None: lambda v: v, "none": lambda v: v, "relu": F.relu, "silu": F.silu}


def edge_gather_add(base, a, src, b, dst, act=None):
    """act(base[e] + a[src[e]] + b[dst[e]])"""
    pre = 0
    if base is not None:
        pre = pre + base
    if a is not None:
        pre = pre + a.index_select(0, src.long())
    if b is not None:
        pre = pre + b.index_select(0, dst.long())
    return ACTS[act](pre)


def aggregate_sum(msg, dst, n):
    out = torch.zeros(n, msg.shape[-1], dtype=msg.dtype)
    return out.index_add_(0, dst.long(), msg)


def edge_mlp_first_layer_concat(e, xs, xr, src, dst, w, bias, act="silu"):
    """The un-distributed form: act(Linear(cat[e, xs[src], xr[dst]])), w: (C_out, 3C).  Used to check that distributing the
    Linear over the concat (what the kernels rely on) is the same function."""
    cat = torch.cat([e, xs.index_select(0, src.long()), xr.index_select(0, dst.long())], dim=-1)
    return ACTS[act](F.linear(cat, w, bias))
