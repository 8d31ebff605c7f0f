"""
HiLAMParallel on the MI355X edge kernels -- the model behind ``model_name: HiLAMParallel``
(config/CLI/model/hilamparallel.yaml: hidden_dims 64, hidden_layers 1, processor_layers 4, mesh_aggr sum).  Same encoder,
mesh initialisation, read-out and decoder as HiLAM (py4cast_amd.hilam); the processor differs: every layer is ONE InteractionNet
over the union of all mesh edges (same-level, up and down of every level) in which each edge set has its own edge MLP and each
level its own node-update MLP (neural-lam's ``edge_chunk_sizes`` / ``aggr_chunk_sizes``, which mfai follows), so all levels
exchange information in every layer instead of sweeping down and up.  PARITY UNPINNED against mfai (absent here); checked
against oracle/hilam.py::HiLamParallel.

Per layer and edge set: node projections of the distributed first Linear (one launch per node tensor and direction,
ops_nodeproj.node_proj), ONE fused row-MLP kernel over the set's edges (ops_mlp.row_mlp with gathered addends), one CSR segment sum into its receiver level; per level the messages of its
(up to three) incoming edge sets are added and one fused node-update kernel runs.
"""

from dataclasses import dataclass
from typing import List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import ops_graph as G
from . import ops_mlp as M
from . import ops_nodeproj as NP
from . import graphlam as _gl
from .graphlam import _run, cached_static_embeddings, grid_rows, make_mlp, output_rows
from .hilam import HiLamMI355X, HiLamSettings

try:
    from dataclasses_json import dataclass_json
except Exception:  # pragma: no cover
    def dataclass_json(cls):
        return cls


@dataclass_json
@dataclass(slots=True)
class HiLamParallelSettings(HiLamSettings):
    pass


class ParallelLayer(nn.Module):
    """One processor layer: an edge MLP per edge set, a node-update MLP per level (neural-lam's SplitMLPs)."""

    def __init__(self, hidden: int, hidden_layers: int, n_sets: int, n_levels: int):
        super().__init__()
        bp = [hidden] * (hidden_layers + 1)
        self.edge_mlps = nn.ModuleList([make_mlp([3 * hidden] + bp) for _ in range(n_sets)])
        self.aggr_mlps = nn.ModuleList([make_mlp([2 * hidden] + bp) for _ in range(n_levels)])


def _edge_messages(mlp: nn.Sequential, send, rec, edge_rep, edges: G.EdgeSet):
    """msg, edge_rep + msg of one edge set (the first Linear distributed over cat[e, x_s[src], x_r[dst]])."""
    C = edge_rep.shape[1]
    lin0, lin1, ln = mlp[0], mlp[2], mlp[3]
    if edge_rep.dtype == torch.bfloat16 and C == 64:
        if send is rec:
            a, b = NP.node_proj(rec, [lin0.weight[:, C:2 * C], lin0.weight[:, 2 * C:]], _gl.GRADS_IN_PLACE)
        else:
            a, = NP.node_proj(send, [lin0.weight[:, C:2 * C]], _gl.GRADS_IN_PLACE)
            b, = NP.node_proj(rec, [lin0.weight[:, 2 * C:]], _gl.GRADS_IN_PLACE)
        return M.row_mlp(edge_rep, lin0.weight[:, :C], lin0.bias, lin1.weight, lin1.bias, ln.weight, ln.bias, ln.eps,
                         ga=a, gb=b, edges=edges, res=edge_rep, grads_in_place=_gl.GRADS_IN_PLACE)
    base = F.linear(edge_rep, lin0.weight[:, :C], lin0.bias)
    h = G.edge_gather_add(base, F.linear(send, lin0.weight[:, C:2 * C]), F.linear(rec, lin0.weight[:, 2 * C:]), edges, "silu")
    msg = _run(mlp[2:], h)
    return msg, edge_rep + msg


def _node_update(mlp: nn.Sequential, rec, agg):
    C = rec.shape[1]
    al0, al1, aln = mlp[0], mlp[2], mlp[3]
    if rec.dtype == torch.bfloat16 and C == 64:
        part, rec = NP.node_proj(rec, [al0.weight[:, :C]], _gl.GRADS_IN_PLACE, passthrough=True)
        return M.row_mlp(agg, al0.weight[:, C:], al0.bias, al1.weight, al1.bias, aln.weight, aln.bias, aln.eps, ga=part, res=rec,
                         want_out=False, grads_in_place=_gl.GRADS_IN_PLACE)[1]
    return _run(mlp, torch.cat([rec, agg], dim=-1), res=rec)


class HiLamParallelMI355X(HiLamMI355X):
    settings_kls = HiLamParallelSettings
    register: bool = True

    def __init__(self, in_channels: int, out_channels: int, input_shape: Tuple[int, ...] = None,
                 settings: HiLamParallelSettings = HiLamParallelSettings(), *args, **kwargs):
        super().__init__(in_channels, out_channels, input_shape, settings, *args, **kwargs)
        for name in ("mesh_down_gnns", "mesh_down_same_gnns", "mesh_up_gnns", "mesh_up_same_gnns"):
            delattr(self, name)    # HiLAM's sweeping processor is replaced
        Lv = self.num_levels
        # edge sets in neural-lam's order: same-level of every level, up of every level pair, down of every level pair
        self._sets = [(f"same{l}", l, l) for l in range(Lv)] + [(f"up{l}", l, l + 1) for l in range(Lv - 1)] \
            + [(f"down{l}", l + 1, l) for l in range(Lv - 1)]
        self.processor = nn.ModuleList([ParallelLayer(settings.hidden_dims, settings.hidden_layers, len(self._sets), Lv)
                                        for _ in range(settings.processor_layers)])

    def _process(self, levels: List[torch.Tensor], same_e, up_e, down_e, es):
        Lv = self.num_levels
        edge_reps = list(same_e) + list(up_e) + list(down_e)
        for layer in self.processor:
            agg = [None] * Lv
            new_edges = []
            for k, (name, ls, lr) in enumerate(self._sets):
                msg, new_e = _edge_messages(layer.edge_mlps[k], levels[ls], levels[lr], edge_reps[k], es[name])
                new_edges.append(new_e)
                part = G.aggregate_sum(msg, es[name])
                agg[lr] = part if agg[lr] is None else agg[lr] + part
            levels = [_node_update(layer.aggr_mlps[l], levels[l], agg[l]) for l in range(Lv)]
            edge_reps = new_edges
        return levels, edge_reps[:Lv], edge_reps[Lv:2 * Lv - 1], edge_reps[2 * Lv - 1:]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, N, _ = x.shape
        dt = torch.bfloat16 if self._settings.activation_dtype == "bf16" else torch.float32
        es, Lv = self._edges(B, x.device), self.num_levels
        grid = _run(self.grid_embedder, grid_rows(self, x, dt))
        embedders = [self.g2m_embedder, self.m2g_embedder] + list(self.mesh_embedders) + list(self.mesh_same_embedders) \
            + list(self.mesh_up_embedders) + list(self.mesh_down_embedders)
        names = ["g2m_features", "m2g_features"] + [f"mesh_pos_{l}" for l in range(Lv)] + [f"same_features_{l}" for l in range(Lv)] \
            + [f"up_features_{l}" for l in range(Lv - 1)] + [f"down_features_{l}" for l in range(Lv - 1)]
        embs = list(cached_static_embeddings(self, embedders, [getattr(self, n) for n in names], B, dt))
        g2m_e, m2g_e = embs[0], embs[1]
        levels = embs[2:2 + Lv]
        same_e = embs[2 + Lv:2 + 2 * Lv]
        up_e = embs[2 + 2 * Lv:2 + 2 * Lv + (Lv - 1)]
        down_e = embs[2 + 2 * Lv + (Lv - 1):]
        levels[0] = self.g2m_gnn(grid, levels[0], g2m_e, es["g2m"])
        grid = _run(self.encoding_grid_mlp, grid, res=grid)
        for l in range(1, Lv):                                              # mesh init, bottom-up
            levels[l], up_e[l - 1] = self.mesh_init_gnns[l - 1](levels[l - 1], levels[l], up_e[l - 1], es[f"up{l - 1}"])
        levels, same_e, up_e, down_e = self._process(levels, same_e, up_e, down_e, es)
        for l in range(Lv - 2, -1, -1):                                     # read-out, top-down
            levels[l] = self.mesh_read_gnns[l](levels[l + 1], levels[l], down_e[l], es[f"down{l}"])
        grid = self.m2g_gnn(levels[0], grid, m2g_e, es["m2g"])
        return output_rows(self, grid, x, B, N)
