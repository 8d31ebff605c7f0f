"""
HiLAM on the MI355X edge kernels -- the model behind ``model_name: HiLAM`` (config/CLI/model/hilam.yaml: hidden_dims 64,
hidden_layers 1, processor_layers 4, mesh_aggr sum).  Same building blocks as py4cast_amd.graphlam (the reference takes both
from mfai v5.0.1, which follows neural-lam): a hierarchy of mesh levels, each 3x coarser; after grid -> mesh the upper levels are
initialised bottom-up, every processor layer sweeps down (down edges then same-level edges per level) and up again, a read-out
sweep brings the result to the bottom level, then mesh -> grid and the output MLP.  PARITY UNPINNED against mfai; checked
against oracle/hilam.py.  Every MLP, node projection, edge pass and parameter-gradient reduction runs on the HIP kernels (see
graphlam.py; round 6: 2 775 launches per training step at 512 x 512 instead of 5 109, 30.6 ms instead of 44.0).
"""

import os
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
from torch import nn

from . import ops_graph as G
from .base import ModelABC, ModelType
from .graph_build import HiMeshGraph, build_hierarchical_graph, hi_graph_path
from .graphlam import GraphLamMI355X, InteractionNet, _run, cached_static_embeddings, grid_rows, make_mlp, output_rows, rollout_format

try:
    from dataclasses_json import dataclass_json
except Exception:  # pragma: no cover
    def dataclass_json(cls):
        return cls


@dataclass_json
@dataclass(slots=True)
class HiLamSettings:
    tmp_dir: str = "/tmp"  # nosec B108 -- same default as the reference yaml
    hidden_dims: int = 64
    hidden_layers: int = 1
    processor_layers: int = 4
    mesh_aggr: str = "sum"
    use_checkpointing: bool = False
    offload_to_cpu: bool = False
    mesh_levels: int = 0
    activation_dtype: str = "f32"


class HiLamMI355X(ModelABC, nn.Module):
    settings_kls = HiLamSettings
    onnx_supported: bool = False
    supported_num_spatial_dims = (1,)
    num_spatial_dims: int = 1
    features_last: bool = True
    model_type = ModelType.GRAPH
    register: bool = True

    def __init__(self, in_channels: int, out_channels: int, input_shape: Tuple[int, ...] = None,
                 settings: HiLamSettings = HiLamSettings(), *args, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.input_shape = in_channels, out_channels, input_shape
        self._settings = settings
        if settings.mesh_aggr != "sum":
            raise NotImplementedError("mesh_aggr: only 'sum' is implemented (the reference yaml's value)")
        path = hi_graph_path(settings.tmp_dir, input_shape, settings.mesh_levels)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run {type(self).__name__}.rank_zero_setup(settings, meshgrid) first")
        g = HiMeshGraph.load(path)
        self.n_grid, self.n_mesh = g.n_grid, list(g.n_mesh)
        self.num_levels = Lv = len(g.n_mesh)
        reg = lambda name, t: self.register_buffer(name, t, persistent=False)  # noqa: E731
        for k in ("g2m", "m2g"):
            reg(f"{k}_index", getattr(g, k))
            reg(f"{k}_features", getattr(g, f"{k}_feat"))
        for l in range(Lv):
            reg(f"mesh_pos_{l}", g.mesh_pos[l])
            reg(f"same_index_{l}", g.same[l])
            reg(f"same_features_{l}", g.same_feat[l])
        for l in range(Lv - 1):
            for k in ("up", "down"):
                reg(f"{k}_index_{l}", getattr(g, k)[l])
                reg(f"{k}_features_{l}", getattr(g, f"{k}_feat")[l])
        self._edge_cache: Dict[tuple, Dict[str, G.EdgeSet]] = {}
        self._static_cache = None

        h, hl, P = settings.hidden_dims, settings.hidden_layers, settings.processor_layers
        bp = [h] * (hl + 1)
        mlps = lambda n, cin: nn.ModuleList([make_mlp([cin] + bp) for _ in range(n)])  # noqa: E731
        gnns = lambda n, upd=True: nn.ModuleList([InteractionNet(h, hl, update_edges=upd) for _ in range(n)])  # noqa: E731
        self.grid_embedder = make_mlp([in_channels] + bp)
        self.g2m_embedder, self.m2g_embedder = make_mlp([3] + bp), make_mlp([3] + bp)
        self.mesh_embedders = mlps(Lv, 2)
        self.mesh_same_embedders = mlps(Lv, 3)
        self.mesh_up_embedders, self.mesh_down_embedders = mlps(Lv - 1, 3), mlps(Lv - 1, 3)
        self.g2m_gnn = InteractionNet(h, hl, update_edges=False)
        self.encoding_grid_mlp = make_mlp([h] + bp)
        self.mesh_init_gnns = gnns(Lv - 1)
        self.mesh_read_gnns = gnns(Lv - 1, False)
        self.mesh_down_gnns = nn.ModuleList([gnns(Lv - 1) for _ in range(P)])
        self.mesh_down_same_gnns = nn.ModuleList([gnns(Lv) for _ in range(P)])
        self.mesh_up_gnns = nn.ModuleList([gnns(Lv - 1) for _ in range(P)])
        self.mesh_up_same_gnns = nn.ModuleList([gnns(Lv) for _ in range(P)])
        self.m2g_gnn = InteractionNet(h, hl, update_edges=False)
        self.output_map = make_mlp(bp + [out_channels], layer_norm=False)
        self.timed_entry_points = ("p4c_edge_gather_add_fwd", "p4c_edge_gather_add_bwd", "p4c_segment_sum", "p4c_row_layernorm_fwd",
                                   "p4c_row_layernorm_bwd", "p4c_row_linear_wgrad", "p4c_row_mlp_fwd", "p4c_row_mlp_bwd", "p4c_row_mlp_bwd_accumulate")
        self.roofline_from_entry_points = True
        self.prefers_hip_graph = True   # ~10^4 launches per training step: host-bound when launched eagerly (trainer.GraphedTrainingStep)
        self.check_required_attributes()

    @property
    def settings(self) -> HiLamSettings:
        return self._settings

    roofline = GraphLamMI355X.roofline   # bench.py hook: same entry points, same bookkeeping

    @classmethod
    def rank_zero_setup(cls, settings: HiLamSettings, meshgrid: torch.Tensor):
        shape = tuple(meshgrid.shape[1:])
        path = hi_graph_path(settings.tmp_dir, shape, settings.mesh_levels)
        if not os.path.exists(path):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            build_hierarchical_graph(meshgrid, settings.mesh_levels).save(path)

    def _edges(self, B: int, device) -> Dict[str, G.EdgeSet]:
        key = (B, str(device))
        if key not in self._edge_cache:
            nm, Lv = self.n_mesh, self.num_levels
            spec = {"g2m": (self.g2m_index, self.n_grid, nm[0]), "m2g": (self.m2g_index, nm[0], self.n_grid)}
            for l in range(Lv):
                spec[f"same{l}"] = (getattr(self, f"same_index_{l}"), nm[l], nm[l])
            for l in range(Lv - 1):
                spec[f"up{l}"] = (getattr(self, f"up_index_{l}"), nm[l], nm[l + 1])
                spec[f"down{l}"] = (getattr(self, f"down_index_{l}"), nm[l + 1], nm[l])
            sets = {}
            for k, (idx, ns, nr) in spec.items():
                src = torch.cat([idx[0] + b * ns for b in range(B)])
                dst = torch.cat([idx[1] + b * nr for b in range(B)])
                sets[k] = G.EdgeSet(src, dst, B * ns, B * nr).to(device)
            self._edge_cache[key] = sets
        return self._edge_cache[key]

    rollout_padded_output = False   # (see graphlam.output_rows)

    @property
    def rollout_input_format(self):
        return rollout_format(self)

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
        levels: List[torch.Tensor] = embs[2:2 + Lv]
        same_e = embs[2 + Lv:2 + 2 * Lv]
        up_e = embs[2 + 2 * Lv:2 + 2 * Lv + (Lv - 1)]
        down_e = embs[2 + 2 * Lv + (Lv - 1):]

        levels[0] = self.g2m_gnn(grid, levels[0], g2m_e, es["g2m"])
        grid = _run(self.encoding_grid_mlp, grid, res=grid)
        for l in range(1, Lv):                                              # mesh init, bottom-up
            levels[l], up_e[l - 1] = self.mesh_init_gnns[l - 1](levels[l - 1], levels[l], up_e[l - 1], es[f"up{l - 1}"])
        for down_g, down_s, up_g, up_s in zip(self.mesh_down_gnns, self.mesh_down_same_gnns, self.mesh_up_gnns, self.mesh_up_same_gnns):
            levels[-1], same_e[-1] = down_s[-1](levels[-1], levels[-1], same_e[-1], es[f"same{Lv - 1}"])
            for l in range(Lv - 2, -1, -1):                                 # down sweep
                new, down_e[l] = down_g[l](levels[l + 1], levels[l], down_e[l], es[f"down{l}"])
                levels[l], same_e[l] = down_s[l](new, new, same_e[l], es[f"same{l}"])
            levels[0], same_e[0] = up_s[0](levels[0], levels[0], same_e[0], es["same0"])
            for l in range(1, Lv):                                          # up sweep
                new, up_e[l - 1] = up_g[l - 1](levels[l - 1], levels[l], up_e[l - 1], es[f"up{l - 1}"])
                levels[l], same_e[l] = up_s[l](new, new, same_e[l], es[f"same{l}"])
        for l in range(Lv - 2, -1, -1):                                     # read-out, top-down
            levels[l] = self.mesh_read_gnns[l](levels[l + 1], levels[l], down_e[l], es[f"down{l}"])
        grid = self.m2g_gnn(levels[0], grid, m2g_e, es["m2g"])
        return output_rows(self, grid, x, B, N)
