"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (mfai v5.0.1, requirements.txt:26, is absent).

Torch-native restatement of GraphLAM (``model_name: GraphLam``, config/CLI/model/graphlam.yaml:19-26) as neural-lam / mfai run
it: encode-process-decode with InteractionNets whose edge MLP consumes ``cat[edge, sender[src], receiver[dst]]`` and whose
aggregation is ``index_add_``.  Takes the graph (edge lists, features) as DATA; parameter names match
py4cast_amd.graphlam.GraphLamMI355X so one state_dict drives both.
"""

import torch
from torch import nn


def make_mlp(blueprint, layer_norm=True):
    layers = []
    for i, (a, b) in enumerate(zip(blueprint[:-1], blueprint[1:])):
        layers.append(nn.Linear(a, b))
        if i != len(blueprint) - 2:
            layers.append(nn.SiLU())
    if layer_norm:
        layers.append(nn.LayerNorm(blueprint[-1]))
    return nn.Sequential(*layers)


class InteractionNet(nn.Module):
    def __init__(self, hidden, hidden_layers=1, update_edges=True, aggr="sum"):
        super().__init__()
        self.update_edges, self.aggr = update_edges, aggr     # neural-lam: PyG MessagePassing aggr = "sum" | "mean"
        self.edge_mlp = make_mlp([3 * hidden] + [hidden] * (hidden_layers + 1))
        self.aggr_mlp = make_mlp([2 * hidden] + [hidden] * (hidden_layers + 1))

    def forward(self, send_rep, rec_rep, edge_rep, index):
        # (B, N, C) node tensors, (B, E, C) edge tensors; index (2, E) [sender, receiver]
        msg = self.edge_mlp(torch.cat([edge_rep, send_rep[:, index[0]], rec_rep[:, index[1]]], dim=-1))
        agg = torch.zeros_like(rec_rep).index_add_(1, index[1], msg)
        if self.aggr == "mean":
            deg = torch.zeros(rec_rep.shape[1], dtype=rec_rep.dtype, device=rec_rep.device).index_add_(0, index[1], torch.ones_like(index[1], dtype=rec_rep.dtype))
            agg = agg / deg.clamp_min(1).view(1, -1, 1)
        rec_rep = rec_rep + self.aggr_mlp(torch.cat([rec_rep, agg], dim=-1))
        return (rec_rep, edge_rep + msg) if self.update_edges else rec_rep


class GraphLam(nn.Module):
    def __init__(self, in_channels, out_channels, graph, hidden=64, hidden_layers=1, processor_layers=4, mesh_aggr="sum"):
        super().__init__()
        self.graph = graph  # dict: g2m, m2m, m2g (2,E) long; *_feat (E,3); mesh_pos (M,2)
        bp = [hidden] * (hidden_layers + 1)
        self.grid_embedder = make_mlp([in_channels] + bp)
        self.g2m_embedder = make_mlp([3] + bp)
        self.m2g_embedder = make_mlp([3] + bp)
        self.mesh_embedder = make_mlp([2] + bp)
        self.m2m_embedder = make_mlp([3] + bp)
        self.g2m_gnn = InteractionNet(hidden, hidden_layers, update_edges=False)
        self.encoding_grid_mlp = make_mlp([hidden] + bp)
        self.processor = nn.ModuleList([InteractionNet(hidden, hidden_layers, aggr=mesh_aggr) for _ in range(processor_layers)])
        self.m2g_gnn = InteractionNet(hidden, hidden_layers, update_edges=False)
        self.output_map = make_mlp(bp + [out_channels], layer_norm=False)

    def forward(self, x):
        g = self.graph
        B = x.shape[0]
        ex = lambda t: t.unsqueeze(0).expand(B, *t.shape)  # noqa: E731
        grid = self.grid_embedder(x)
        g2m_e, m2g_e, m2m_e = (ex(emb(g[k].to(x.dtype))) for emb, k in
                               ((self.g2m_embedder, "g2m_feat"), (self.m2g_embedder, "m2g_feat"), (self.m2m_embedder, "m2m_feat")))
        mesh = ex(self.mesh_embedder(g["mesh_pos"].to(x.dtype)))
        mesh = self.g2m_gnn(grid, mesh, g2m_e, g["g2m"])
        grid = grid + self.encoding_grid_mlp(grid)
        for layer in self.processor:
            mesh, m2m_e = layer(mesh, mesh, m2m_e, g["m2m"])
        grid = self.m2g_gnn(mesh, grid, m2g_e, g["m2g"])
        return self.output_map(grid)
