"""
GraphLAM on the MI355X edge kernels -- the model behind ``model_name: GraphLam``
(config/CLI/model/graphlam.yaml:19-26: hidden_dims 64, hidden_layers 1, processor_layers 4, mesh_aggr sum, tmp_dir).
The reference takes the class from mfai v5.0.1 (py4cast/models.py:10-20), which follows neural-lam's encode-process-decode
GNN: grid/mesh/edge embedders (MLP = Linear-SiLU-Linear-LayerNorm), InteractionNet grid->mesh, ``processor_layers``
InteractionNets on the mesh, InteractionNet mesh->grid, output MLP.  PARITY UNPINNED against mfai (absent here); the
arithmetic is checked against oracle/graphlam.py, which runs the same parameters through index_select / cat / index_add_.

What runs where (bf16 activations, ``activation_dtype: bf16``; state after round 6)
* every MLP (Linear - SiLU - Linear - LayerNorm [+ residual], hidden 64): ONE fused kernel each way (csrc/mlp.hip through
  py4cast_amd.ops_mlp.row_mlp); on edges the first Linear is distributed over ``cat[e, x_s[src], x_r[dst]]`` and the sender / receiver
  parts enter the kernel as gathered addends, so nothing of size E x 3C is ever formed;
* the node projections of that distributed first Linear (and the receiver half of the node-update MLP's): ONE launch per direction for
  all blocks that multiply the same node tensor (csrc/nodeproj.hip through py4cast_amd.ops_nodeproj.node_proj), the residual's gradient
  summed inside the data-gradient launch;
* aggregation: a CSR segment sum (csrc/graph.hip; no atomics, reproducible), its adjoint an edge gather;
* parameter gradients: added straight into the parameters' ``.grad`` buffers by one batched reduction per backward pass
  (ops_nodeproj.GradQueue) -- no per-parameter AccumulateGrad launch, no per-call reduction launch;
* the fp32 flavour (``activation_dtype: f32``, parity) keeps library GEMMs for the Linears and the row LayerNorm / gather kernels.
The batch dimension is folded into the node dimension (edge lists replicated with node offsets, cached per batch size).
Graph models receive (B, ngrid, C_in) and return (B, ngrid, F) (py4cast/lightning.py:526-535).
"""

import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib as L
from . import ops_graph as G
from . import ops_mlp as M
from . import ops_nodeproj as NP
from . import ops_rows as R
from .base import ModelABC, ModelType
from .graph_build import MeshGraph, build_mesh_graph, graph_path

try:  # the reference's settings classes are dataclass_json dataclasses (py4cast_plugin_example.py:12-17)
    from dataclasses_json import dataclass_json
except Exception:  # pragma: no cover
    def dataclass_json(cls):
        return cls


@dataclass_json
@dataclass(slots=True)
class GraphLamSettings:
    tmp_dir: str = "/tmp"  # nosec B108 -- same default as the reference yaml
    hidden_dims: int = 64
    hidden_layers: int = 1
    processor_layers: int = 4
    mesh_aggr: str = "sum"
    use_checkpointing: bool = False
    offload_to_cpu: bool = False
    mesh_levels: int = 0            # 0 = every level that fits (neural-lam's multiscale mesh)
    activation_dtype: str = "f32"   # "bf16": node / edge representations stored as bf16


# The fused MLP kernels add their parameter gradients straight into the parameters' existing .grad buffers (one reduction launch)
# instead of returning them for autograd's AccumulateGrad (six small `+=` launches per MLP application, ~10^3-10^4 per step).
# Only when every parameter already HAS a gradient buffer (FlatDDP / Trainer allocate them; otherwise the ordinary path runs);
# set to False for code that needs torch.autograd.grad() with respect to these parameters.
GRADS_IN_PLACE = True


def make_mlp(blueprint, layer_norm=True) -> nn.Sequential:
    layers = []
    for i, (a, b) in enumerate(zip(blueprint[:-1], blueprint[1:])):
        layers.append(nn.Linear(a, b))
        if i != len(blueprint) - 2:
            layers.append(nn.SiLU())
    if layer_norm:
        layers.append(nn.LayerNorm(blueprint[-1]))
    return nn.Sequential(*layers)


def _linear(m: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """A Linear on rows of x's dtype (fp32 master parameters).  bf16 rows go through ops_rows.row_linear (native weight
    gradient), with the input / output features zero-padded to the kernel's shapes (K multiple of 16, 64 outputs)."""
    w, b = m.weight, m.bias
    if x.dtype != torch.bfloat16:
        R.L.require_cuda(x)   # fp32 parity flavour: the GPU library's GEMM (there is no CPU path)
        return F.linear(x, w, b)
    O, K = w.shape
    kp, op = (-K) % 16, (64 - O) if O < 64 else 0
    if kp:
        x, w = F.pad(x, (0, kp)), F.pad(w, (0, kp))
    if op:
        w, b = F.pad(w, (0, 0, 0, op)), F.pad(b, (0, op))
    y = R.row_linear(x, w, b)
    return y[:, :O] if op else y


def _fusable(mlp: nn.Sequential, x: torch.Tensor) -> bool:
    """Linear - SiLU - Linear [- LayerNorm] with 64 hidden features on bf16 rows: the fused row-MLP kernel's shape."""
    n = len(mlp)
    if n not in (3, 4) or not (isinstance(mlp[0], nn.Linear) and isinstance(mlp[1], nn.SiLU) and isinstance(mlp[2], nn.Linear)):
        return False
    if n == 4 and not (isinstance(mlp[3], nn.LayerNorm) and mlp[3].normalized_shape == (64,)):
        return False
    return M.supported(x, mlp[0].weight, mlp[2].weight) and x.shape[0] >= 1


def output_rows(model: nn.Module, grid: torch.Tensor, x: torch.Tensor, B: int, N: int) -> torch.Tensor:
    """The output MLP on the grid rows -> (B, N, out_channels) in x's dtype; inside the py4cast_amd rollout (``rollout_padded_output``
    set by the caller) the fused kernel's 64-wide rows go out as they are: the state update reads the first out_channels features,
    and neither the sliced copy nor -- backward -- a zero-filled 64-wide gradient and a copy into it exist."""
    if model.rollout_padded_output and _fusable(model.output_map, grid):
        return _run(model.output_map, grid, keep_pad=True).to(x.dtype).reshape(B, N, -1)
    return _run(model.output_map, grid).to(x.dtype).reshape(B, N, model.out_channels)


def _run(mlp: nn.Sequential, x: torch.Tensor, res: Optional[torch.Tensor] = None, keep_pad: bool = False) -> torch.Tensor:
    """An MLP on rows of x's dtype; ``res`` is added to the result.  bf16 rows of the standard shape take ONE fused kernel each way
    (ops_mlp.row_mlp); other shapes run layer by layer (library GEMMs + the row LayerNorm / weight-gradient kernels)."""
    if _fusable(mlp, x):
        ln = mlp[3] if len(mlp) == 4 else None
        out, out_res = M.row_mlp(x, mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias,
                                 None if ln is None else ln.weight, None if ln is None else ln.bias,
                                 1e-5 if ln is None else ln.eps, res=res, want_out=res is None, grads_in_place=GRADS_IN_PLACE)
        y = out if res is None else out_res
        return y if mlp[2].out_features == 64 or keep_pad else y[:, : mlp[2].out_features]
    for m in mlp:
        if isinstance(m, nn.Linear):
            x = _linear(m, x)
        elif isinstance(m, nn.LayerNorm):
            x = R.row_layer_norm(x.contiguous(), m.weight, m.bias, m.eps, res)
            res = None
        else:
            x = m(x)
    return x if res is None else x + res


def grid_rows(model: nn.Module, x: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    """(B, N, C) grid input as (B*N, C') rows of the activation dtype.  Rows that arrive zero-padded from build_x
    (``rollout_input_format``) go to the fused MLP as they are (its K is a multiple of 16; the weight image knows the real width)."""
    B, N, C = x.shape
    rows = x.reshape(B * N, C).to(dt)
    if C != model.in_channels:
        fmt = rollout_format(model)
        if fmt is None or C != fmt[1] or x.dtype != fmt[0]:
            raise L.P4CError(f"{type(model).__name__}: {C} input features, expected {model.in_channels}")
        if not _fusable(model.grid_embedder, rows):
            rows = rows[:, : model.in_channels]
    return rows


def rollout_format(model: nn.Module):
    """(dtype, feature count) the rollout's build_x should emit for a mesh-GNN: bf16 rows zero-padded to the fused MLP's multiple of 16
    -- otherwise every AR step casts the fp32 rows and pads them (two passes over the grid input, and their adjoints).  fp32: None."""
    if model._settings.activation_dtype != "bf16" or model.in_channels > M.MAX_K or L.diag_switch("P4C_NO_ROLLOUT_FORMAT") == "1":
        return None
    return torch.bfloat16, (model.in_channels + 15) // 16 * 16


def cached_static_embeddings(model: nn.Module, embedders, feats, B: int, dt: torch.dtype):
    """Embeddings of the graph's static edge / mesh-node features, batch-expanded.  They depend on the parameters only, so the AR
    steps of one rollout (the model is called once per step, py4cast/lightning.py:591-596) share them: computed once per parameter
    version (and autograd mode); their autograd graph is walked once by the rollout's backward, which sums the steps' gradients,
    and is rebuilt by the next forward (gradient accumulation over micro-batches with unchanged parameters)."""
    params = [p for e in embedders for p in e.parameters()]
    key = (B, dt, torch.is_grad_enabled(), L.PARAM_EPOCH[0], tuple(p._version for p in params), tuple(p.data_ptr() for p in params))

    def compute():
        rep = lambda t: t.unsqueeze(0).expand(B, *t.shape).reshape(B * t.shape[0], t.shape[1])  # noqa: E731
        return tuple(rep(_run(e, f.to(dt))) for e, f in zip(embedders, feats))

    if torch.cuda.is_current_stream_capturing():
        # a HIP graph must derive the embeddings with ITS OWN kernels (a replay has to see the current parameters, and a tensor of
        # an eager warm-up step would be read after the allocator has recycled it): once per capture when the capturing code
        # opened a scope, else once per call; the module-level cache is neither read nor written here
        scope = L.capture_cache()
        if scope is None:
            return compute()
        skey = ("static_emb", id(model)) + key
        if skey not in scope:
            scope[skey] = compute()
        return scope[skey]
    cache = getattr(model, "_static_cache", None)
    if cache is None or cache[0] != key or not L.owners_alive(cache[2], params):
        embs = compute()
        model._static_cache = (key, embs, L.owner_refs(params))

        def drop(grad):
            model._static_cache = None

        for t in embs:
            if t.requires_grad:
                t.register_hook(drop)
    return model._static_cache[1]


class InteractionNet(nn.Module):
    """neural-lam's InteractionNet; parameter layout identical to the concat formulation (edge_mlp.0.weight is (C, 3C))."""

    def __init__(self, hidden: int, hidden_layers: int = 1, update_edges: bool = True, aggr: str = "sum"):
        super().__init__()
        if aggr not in ("sum", "mean"):
            raise NotImplementedError(f"InteractionNet: aggr={aggr!r} (neural-lam: 'sum' or 'mean')")
        self.hidden, self.update_edges, self.aggr = hidden, update_edges, aggr
        self.edge_mlp = make_mlp([3 * hidden] + [hidden] * (hidden_layers + 1))
        self.aggr_mlp = make_mlp([2 * hidden] + [hidden] * (hidden_layers + 1))

    def forward(self, send_rep, rec_rep, edge_rep, edges: G.EdgeSet):
        C = self.hidden
        lin0, lin1, ln = self.edge_mlp[0], self.edge_mlp[2], self.edge_mlp[3]
        part, rec_res = None, rec_rep
        if edge_rep.dtype == torch.bfloat16 and C == 64 and edge_rep.shape[0] >= 1:
            # sender / receiver parts of the first Linear once per NODE, everything per EDGE in one kernel:
            # e W_e + a[src] + b[dst] + bias -> SiLU -> Linear -> LayerNorm -> msg (and edge_rep + msg)
            # the projections of one node tensor are ONE launch each way (ops_nodeproj.node_proj: their input gradients are one K = 64 n
            # product, their weight gradients one launch into the batched reduction); `part` is the receiver part of the node-update
            # MLP's first Linear, used after the aggregation below
            # (rec_res IS rec_rep, handed back by the projection node: the residual's gradient is then summed inside that node's
            # data-gradient launch instead of by an element-wise launch of autograd's)
            al0 = self.aggr_mlp[0]
            if send_rep is rec_rep:
                a, b, part, rec_res = NP.node_proj(rec_rep, [lin0.weight[:, C:2 * C], lin0.weight[:, 2 * C:], al0.weight[:, :C]],
                                                   GRADS_IN_PLACE, passthrough=True)
            else:
                a, = NP.node_proj(send_rep, [lin0.weight[:, C:2 * C]], GRADS_IN_PLACE)
                b, part, rec_res = NP.node_proj(rec_rep, [lin0.weight[:, 2 * C:], al0.weight[:, :C]], GRADS_IN_PLACE, passthrough=True)
            msg, new_edge = M.row_mlp(edge_rep, lin0.weight[:, :C], lin0.bias, lin1.weight, lin1.bias, ln.weight, ln.bias, ln.eps,
                                      ga=a, gb=b, edges=edges, res=edge_rep if self.update_edges else None,
                                      grads_in_place=GRADS_IN_PLACE)
        else:
            if edge_rep.dtype == torch.bfloat16:
                base = R.row_linear(edge_rep, lin0.weight[:, :C], lin0.bias)          # E x C
                a = R.row_linear(send_rep, lin0.weight[:, C:2 * C], grads_in_place=GRADS_IN_PLACE)   # N_s x C
                b = R.row_linear(rec_rep, lin0.weight[:, 2 * C:], grads_in_place=GRADS_IN_PLACE)     # N_r x C
            else:
                base = F.linear(edge_rep, lin0.weight[:, :C], lin0.bias)
                a = F.linear(send_rep, lin0.weight[:, C:2 * C])
                b = F.linear(rec_rep, lin0.weight[:, 2 * C:])
            h = G.edge_gather_add(base, a, b, edges, "silu")                           # first Linear + SiLU of the edge MLP
            msg = _run(self.edge_mlp[2:], h)
            new_edge = edge_rep + msg if self.update_edges else None
        agg = G.aggregate_sum(msg, edges)
        if self.aggr == "mean":      # neural-lam's mesh_aggr: mean -- the sum over a receiver's edges divided by their number
            B = agg.shape[0] // edges.n_dst if agg.shape[0] != edges.n_dst else 1
            inv = edges.inv_degree(agg.dtype)
            agg = agg * (inv if B == 1 else inv.repeat(B, 1))
        al0, al1, aln = self.aggr_mlp[0], self.aggr_mlp[2], self.aggr_mlp[3]
        if rec_rep.dtype == torch.bfloat16 and C == 64 and rec_rep.shape[0] >= 1:
            # Linear over cat[x_r, agg] = x_r W[:, :C]^T (small library GEMM, row-aligned addend) + agg W[:, C:]^T (fused kernel's x)
            if part is None:
                part, rec_res = NP.node_proj(rec_rep, [al0.weight[:, :C]], GRADS_IN_PLACE, passthrough=True)
            _, rec_rep = M.row_mlp(agg, al0.weight[:, C:], al0.bias, al1.weight, al1.bias, aln.weight, aln.bias, aln.eps,
                                   ga=part, res=rec_res, want_out=False, grads_in_place=GRADS_IN_PLACE)
        else:
            rec_rep = _run(self.aggr_mlp, torch.cat([rec_rep, agg], dim=-1), res=rec_rep)
        if self.update_edges:
            return rec_rep, new_edge
        return rec_rep

class GraphLamMI355X(ModelABC, nn.Module):
    settings_kls = GraphLamSettings
    onnx_supported: bool = False
    supported_num_spatial_dims = (1,)
    num_spatial_dims: int = 1
    features_last: bool = True
    model_type = ModelType.GRAPH
    register: bool = True

    def __init__(self, in_channels: int, out_channels: int, input_shape: Tuple[int, ...] = None,
                 settings: GraphLamSettings = GraphLamSettings(), *args, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.input_shape = in_channels, out_channels, input_shape
        self._settings = settings
        if settings.mesh_aggr not in ("sum", "mean"):
            raise NotImplementedError(f"mesh_aggr={settings.mesh_aggr!r}: 'sum' (the reference yaml's value) or 'mean'")
        path = graph_path(settings.tmp_dir, input_shape, settings.mesh_levels)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: run {type(self).__name__}.rank_zero_setup(settings, meshgrid) first "
                                    "(py4cast does, lightning.py:141-144)")
        graph = MeshGraph.load(path)
        self.n_grid, self.n_mesh = graph.n_grid, graph.n_mesh
        for k in ("g2m", "m2m", "m2g"):
            self.register_buffer(f"{k}_index", getattr(graph, k), persistent=False)
            self.register_buffer(f"{k}_features", getattr(graph, f"{k}_feat"), persistent=False)
        self.register_buffer("mesh_static_features", graph.mesh_pos, persistent=False)
        self._edge_cache: Dict[tuple, Dict[str, G.EdgeSet]] = {}
        self._static_cache = None

        h, L_ = settings.hidden_dims, settings.hidden_layers
        bp = [h] * (L_ + 1)
        self.grid_embedder = make_mlp([in_channels] + bp)
        self.g2m_embedder = make_mlp([3] + bp)
        self.m2g_embedder = make_mlp([3] + bp)
        self.mesh_embedder = make_mlp([2] + bp)
        self.m2m_embedder = make_mlp([3] + bp)
        self.g2m_gnn = InteractionNet(h, L_, update_edges=False)
        self.encoding_grid_mlp = make_mlp([h] + bp)
        self.processor = nn.ModuleList([InteractionNet(h, L_, update_edges=True, aggr=settings.mesh_aggr)    # (the mesh processor only,
                                        for _ in range(settings.processor_layers)])                           # as in neural-lam)
        self.m2g_gnn = InteractionNet(h, L_, update_edges=False)
        self.output_map = make_mlp(bp + [out_channels], layer_norm=False)
        self.timed_entry_points = ("p4c_edge_gather_add_fwd", "p4c_edge_gather_add_bwd", "p4c_segment_sum",
                                   "p4c_row_layernorm_fwd", "p4c_row_layernorm_bwd", "p4c_row_linear_wgrad",
                                   "p4c_row_mlp_fwd", "p4c_row_mlp_bwd", "p4c_row_mlp_bwd_accumulate")
        self.roofline_from_entry_points = True   # bench.py: time every call of the entry points above
        self.prefers_hip_graph = True            # ~10^3 launches per training step: replay them from a HIP graph (trainer.GraphedTrainingStep)
        self.check_required_attributes()

    @property
    def settings(self) -> GraphLamSettings:
        return self._settings

    @classmethod
    def rank_zero_setup(cls, settings: GraphLamSettings, meshgrid: torch.Tensor):
        """Builds the mesh graph once and stores it under settings.tmp_dir (lightning.py:141-144)."""
        shape = tuple(meshgrid.shape[1:])
        path = graph_path(settings.tmp_dir, shape, settings.mesh_levels)
        if not os.path.exists(path):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            build_mesh_graph(meshgrid, settings.mesh_levels).save(path)

    def _edges(self, B: int, device) -> Dict[str, G.EdgeSet]:
        key = (B, str(device))
        if key not in self._edge_cache:
            sets = {}
            sizes = {"g2m": (self.n_grid, self.n_mesh), "m2m": (self.n_mesh, self.n_mesh), "m2g": (self.n_mesh, self.n_grid)}
            for k, (ns, nr) in sizes.items():
                idx = getattr(self, f"{k}_index")
                src = torch.cat([idx[0] + b * ns for b in range(B)])
                dst = torch.cat([idx[1] + b * nr for b in range(B)])
                sets[k] = G.EdgeSet(src, dst, B * ns, B * nr).to(device)
            self._edge_cache[key] = sets
        return self._edge_cache[key]

    def _static_embeddings(self, B: int, dt: torch.dtype):
        embedders = (self.g2m_embedder, self.m2g_embedder, self.m2m_embedder, self.mesh_embedder)
        feats = (self.g2m_features, self.m2g_features, self.m2m_features, self.mesh_static_features)
        return cached_static_embeddings(self, embedders, feats, B, dt)

    rollout_padded_output = False   # set by the rollout around its calls: rows wider than out_channels are welcome (see output_rows)

    @property
    def rollout_input_format(self):
        return rollout_format(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, N, _ = x.shape
        dt = torch.bfloat16 if self._settings.activation_dtype == "bf16" else torch.float32
        es = self._edges(B, x.device)
        grid = _run(self.grid_embedder, grid_rows(self, x, dt))
        g2m_e, m2g_e, m2m_e, mesh = self._static_embeddings(B, dt)
        mesh = self.g2m_gnn(grid, mesh, g2m_e, es["g2m"])
        grid = _run(self.encoding_grid_mlp, grid, res=grid)
        for layer in self.processor:
            mesh, m2m_e = layer(mesh, mesh, m2m_e, es["m2m"])
        grid = self.m2g_gnn(mesh, grid, m2g_e, es["m2g"])
        return output_rows(self, grid, x, B, N)

    # ------------------------------------------------------------------ bench.py hook
    def roofline(self, ktimes, B, H, W):
        """Achieved HBM rate of the native entry point that takes the most time: algorithmic bytes of its calls (stated by the
        wrappers in ops_graph / ops_rows next to each call) over their HIP-event durations."""
        from . import _lib as L

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
