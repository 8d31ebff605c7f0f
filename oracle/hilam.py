"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED (mfai v5.0.1 is absent).

Torch-native restatement of HiLAM (``model_name: HiLAM``, config/CLI/model/hilam.yaml) as neural-lam / mfai run it:
hierarchical mesh, InteractionNets on cat[edge, sender[src], receiver[dst]] with index_add_ aggregation; mesh init (up), per
processor layer a down sweep and an up sweep (vertical edges then same-level edges per level), read-out (down), decode.  The
graph is DATA (dict of edge lists / features per level); parameter names match py4cast_amd.hilam.HiLamMI355X.
"""

import torch
from torch import nn

from .graphlam import InteractionNet, make_mlp


class HiLam(nn.Module):
    def __init__(self, in_channels, out_channels, graph, hidden=64, hidden_layers=1, processor_layers=4):
        super().__init__()
        self.graph = graph
        self.num_levels = Lv = len(graph["mesh_pos"])
        bp = [hidden] * (hidden_layers + 1)
        mlps = lambda n, cin: nn.ModuleList([make_mlp([cin] + bp) for _ in range(n)])  # noqa: E731
        gnns = lambda n, upd=True: nn.ModuleList([InteractionNet(hidden, hidden_layers, update_edges=upd) for _ in range(n)])  # noqa: E731
        self.grid_embedder = make_mlp([in_channels] + bp)
        self.g2m_embedder, self.m2g_embedder = make_mlp([3] + bp), make_mlp([3] + bp)
        self.mesh_embedders, self.mesh_same_embedders = mlps(Lv, 2), mlps(Lv, 3)
        self.mesh_up_embedders, self.mesh_down_embedders = mlps(Lv - 1, 3), mlps(Lv - 1, 3)
        self.g2m_gnn = InteractionNet(hidden, hidden_layers, update_edges=False)
        self.encoding_grid_mlp = make_mlp([hidden] + bp)
        self.mesh_init_gnns, self.mesh_read_gnns = gnns(Lv - 1), gnns(Lv - 1, False)
        P = processor_layers
        self.mesh_down_gnns = nn.ModuleList([gnns(Lv - 1) for _ in range(P)])
        self.mesh_down_same_gnns = nn.ModuleList([gnns(Lv) for _ in range(P)])
        self.mesh_up_gnns = nn.ModuleList([gnns(Lv - 1) for _ in range(P)])
        self.mesh_up_same_gnns = nn.ModuleList([gnns(Lv) for _ in range(P)])
        self.m2g_gnn = InteractionNet(hidden, hidden_layers, update_edges=False)
        self.output_map = make_mlp(bp + [out_channels], layer_norm=False)

    def forward(self, x):
        g, Lv, B = self.graph, self.num_levels, x.shape[0]
        ex = lambda t: t.unsqueeze(0).expand(B, *t.shape)  # noqa: E731
        f = lambda t: t.to(x.dtype)  # noqa: E731
        grid = self.grid_embedder(x)
        g2m_e, m2g_e = ex(self.g2m_embedder(f(g["g2m_feat"]))), ex(self.m2g_embedder(f(g["m2g_feat"])))
        levels = [ex(self.mesh_embedders[l](f(g["mesh_pos"][l]))) for l in range(Lv)]
        same_e = [ex(self.mesh_same_embedders[l](f(g["same_feat"][l]))) for l in range(Lv)]
        up_e = [ex(self.mesh_up_embedders[l](f(g["up_feat"][l]))) for l in range(Lv - 1)]
        down_e = [ex(self.mesh_down_embedders[l](f(g["down_feat"][l]))) for l in range(Lv - 1)]
        levels[0] = self.g2m_gnn(grid, levels[0], g2m_e, g["g2m"])
        grid = grid + self.encoding_grid_mlp(grid)
        for l in range(1, Lv):
            levels[l], up_e[l - 1] = self.mesh_init_gnns[l - 1](levels[l - 1], levels[l], up_e[l - 1], g["up"][l - 1])
        for down_g, down_s, up_g, up_s in zip(self.mesh_down_gnns, self.mesh_down_same_gnns, self.mesh_up_gnns, self.mesh_up_same_gnns):
            levels[-1], same_e[-1] = down_s[-1](levels[-1], levels[-1], same_e[-1], g["same"][-1])
            for l in range(Lv - 2, -1, -1):
                new, down_e[l] = down_g[l](levels[l + 1], levels[l], down_e[l], g["down"][l])
                levels[l], same_e[l] = down_s[l](new, new, same_e[l], g["same"][l])
            levels[0], same_e[0] = up_s[0](levels[0], levels[0], same_e[0], g["same"][0])
            for l in range(1, Lv):
                new, up_e[l - 1] = up_g[l - 1](levels[l - 1], levels[l], up_e[l - 1], g["up"][l - 1])
                levels[l], same_e[l] = up_s[l](new, new, same_e[l], g["same"][l])
        for l in range(Lv - 2, -1, -1):
            levels[l] = self.mesh_read_gnns[l](levels[l + 1], levels[l], down_e[l], g["down"][l])
        grid = self.m2g_gnn(levels[0], grid, m2g_e, g["m2g"])
        return self.output_map(grid)


class ParallelLayer(nn.Module):
    def __init__(self, hidden, hidden_layers, n_sets, n_levels):
        super().__init__()
        bp = [hidden] * (hidden_layers + 1)
        self.edge_mlps = nn.ModuleList([make_mlp([3 * hidden] + bp) for _ in range(n_sets)])
        self.aggr_mlps = nn.ModuleList([make_mlp([2 * hidden] + bp) for _ in range(n_levels)])


class HiLamParallel(HiLam):
    """``model_name: HiLAMParallel`` (config/CLI/model/hilamparallel.yaml): HiLAM's encoder / mesh init / read-out / decoder with a
    processor whose every layer is one InteractionNet over ALL mesh edges (same-level, up, down), one edge MLP per edge set and one
    node-update MLP per level (neural-lam's SplitMLPs), written on the joined node / edge tensors as neural-lam runs it."""

    def __init__(self, in_channels, out_channels, graph, hidden=64, hidden_layers=1, processor_layers=4):
        super().__init__(in_channels, out_channels, graph, hidden, hidden_layers, processor_layers)
        for name in ("mesh_down_gnns", "mesh_down_same_gnns", "mesh_up_gnns", "mesh_up_same_gnns"):
            delattr(self, name)
        Lv = self.num_levels
        self.processor = nn.ModuleList([ParallelLayer(hidden, hidden_layers, 3 * Lv - 2, Lv) for _ in range(processor_layers)])

    def forward(self, x):
        g, Lv, B = self.graph, self.num_levels, x.shape[0]
        ex = lambda t: t.unsqueeze(0).expand(B, *t.shape)  # noqa: E731
        f = lambda t: t.to(x.dtype)  # noqa: E731
        grid = self.grid_embedder(x)
        g2m_e, m2g_e = ex(self.g2m_embedder(f(g["g2m_feat"]))), ex(self.m2g_embedder(f(g["m2g_feat"])))
        levels = [ex(self.mesh_embedders[l](f(g["mesh_pos"][l]))) for l in range(Lv)]
        same_e = [ex(self.mesh_same_embedders[l](f(g["same_feat"][l]))) for l in range(Lv)]
        up_e = [ex(self.mesh_up_embedders[l](f(g["up_feat"][l]))) for l in range(Lv - 1)]
        down_e = [ex(self.mesh_down_embedders[l](f(g["down_feat"][l]))) for l in range(Lv - 1)]
        levels[0] = self.g2m_gnn(grid, levels[0], g2m_e, g["g2m"])
        grid = grid + self.encoding_grid_mlp(grid)
        for l in range(1, Lv):
            levels[l], up_e[l - 1] = self.mesh_init_gnns[l - 1](levels[l - 1], levels[l], up_e[l - 1], g["up"][l - 1])
        # joined tensors with global node numbering (level offsets), as neural-lam's hi_processor_step
        sizes = [t.shape[1] for t in levels]
        offs = [sum(sizes[:l]) for l in range(Lv)]
        index = [g["same"][l] + offs[l] for l in range(Lv)]
        index += [torch.stack([g["up"][l][0] + offs[l], g["up"][l][1] + offs[l + 1]]) for l in range(Lv - 1)]
        index += [torch.stack([g["down"][l][0] + offs[l + 1], g["down"][l][1] + offs[l]]) for l in range(Lv - 1)]
        sections = [e.shape[1] for e in index]
        total = torch.cat(index, dim=1)
        mesh = torch.cat(levels, dim=1)
        edge = torch.cat(same_e + up_e + down_e, dim=1)
        for layer in self.processor:
            cat_in = torch.cat([edge, mesh[:, total[0]], mesh[:, total[1]]], dim=-1)
            msg = torch.cat([mlp(c) for mlp, c in zip(layer.edge_mlps, torch.split(cat_in, sections, dim=1))], dim=1)
            agg = torch.zeros_like(mesh).index_add_(1, total[1], msg)
            upd = torch.cat([mlp(c) for mlp, c in zip(layer.aggr_mlps, torch.split(torch.cat([mesh, agg], dim=-1), sizes, dim=1))], dim=1)
            mesh = mesh + upd
            edge = edge + msg
        levels = list(torch.split(mesh, sizes, dim=1))
        secs = torch.split(edge, sections, dim=1)
        down_e = list(secs[2 * Lv - 1:])
        for l in range(Lv - 2, -1, -1):
            levels[l] = self.mesh_read_gnns[l](levels[l + 1], levels[l], down_e[l], g["down"][l])
        grid = self.m2g_gnn(levels[0], grid, m2g_e, g["m2g"])
        return self.output_map(grid)
