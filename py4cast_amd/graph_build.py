"""Mesh-graph construction for the GNN models (host side, runs once: the reference calls the model's
``rank_zero_setup(settings, meshgrid)`` from rank 0 before training, py4cast/lightning.py:141-144, and mfai's GraphLAM writes
the graph to ``settings.tmp_dir``, config/CLI/model/graphlam.yaml:20).

Recipe = neural-lam's ``create_mesh`` (Oskarsson et al. 2023), which mfai follows: a regular mesh lattice refined by 3 per
level; non-hierarchical ("multiscale") models merge the m2m edges of every level onto the finest lattice; grid -> mesh edges
inside 0.67 x the mesh spacing; mesh -> grid edges from the 4 nearest mesh nodes; edge features = (length, dx, dy) divided
by the longest edge, mesh node features = positions divided by the largest coordinate.
"""

import math
import os
from dataclasses import dataclass
from typing import List

import numpy as np
import torch


@dataclass
class MeshGraph:
    n_grid: int
    n_mesh: int
    mesh_pos: torch.Tensor         # (n_mesh, 2) float32, normalised
    g2m: torch.Tensor              # (2, E) int64: [grid sender, mesh receiver]
    g2m_feat: torch.Tensor         # (E, 3)
    m2m: torch.Tensor              # (2, E): [mesh sender, mesh receiver]
    m2m_feat: torch.Tensor
    m2g: torch.Tensor              # (2, E): [mesh sender, grid receiver]
    m2g_feat: torch.Tensor

    def save(self, path: str):
        torch.save(self.__dict__, path)

    @staticmethod
    def load(path: str) -> "MeshGraph":
        return MeshGraph(**torch.load(path, weights_only=True))


def _edge_features(pos_s: np.ndarray, pos_r: np.ndarray, edges: np.ndarray) -> np.ndarray:
    d = pos_s[edges[0]] - pos_r[edges[1]]
    ln = np.sqrt((d ** 2).sum(1, keepdims=True))
    f = np.concatenate([ln, d], axis=1)
    return (f / max(float(ln.max()), 1e-12)).astype(np.float32)


def _lattice_edges(n_y: int, n_x: int, stride: int, off: int) -> np.ndarray:
    """8-neighbour, both directions, between the lattice nodes (off + i*stride, off + j*stride)."""
    ys, xs = np.arange(off, n_y, stride), np.arange(off, n_x, stride)
    if len(ys) < 2 and len(xs) < 2:
        return np.zeros((2, 0), dtype=np.int64)
    yy, xx = np.meshgrid(ys, xs, indexing="ij")
    out = []
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
        ny, nx = yy + dy * stride, xx + dx * stride
        ok = (ny >= 0) & (ny < n_y) & (nx >= 0) & (nx < n_x)
        a, b = (yy * n_x + xx)[ok], (ny * n_x + nx)[ok]
        out += [np.stack([a, b]), np.stack([b, a])]
    return np.concatenate(out, axis=1).astype(np.int64)


def build_mesh_graph(meshgrid: torch.Tensor, levels: int = 0, refine: int = 3) -> MeshGraph:
    """meshgrid: (2, H, W) grid-node coordinates (Statics.meshgrid, base.py:216-230).  levels = 0: as many as fit."""
    from scipy.spatial import cKDTree

    xy = meshgrid.detach().cpu().double().numpy()
    _, H, W = xy.shape
    grid_pos = xy.reshape(2, -1).T                                   # (H*W, 2), row-major == the "ngrid" flattening
    nlev = int(math.log(max(H, W)) / math.log(refine))
    nleaf = refine ** nlev
    n = max(nleaf // refine, 2)                                      # finest mesh lattice: n x n
    lo, hi = grid_pos.min(0), grid_pos.max(0)
    step = (hi - lo) / n
    my, mx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    mesh_pos = np.stack([lo[0] + (mx.ravel() + 0.5) * step[0], lo[1] + (my.ravel() + 0.5) * step[1]], axis=1)

    max_levels = max(nlev - 1, 1)
    levels = max_levels if levels <= 0 else min(levels, max_levels)
    m2m: List[np.ndarray] = []
    for lev in range(levels):
        stride = refine ** lev
        e = _lattice_edges(n, n, stride, (stride - 1) // 2)
        if e.shape[1]:
            m2m.append(e)
    m2m_e = np.concatenate(m2m, axis=1)

    dm = float(np.sqrt((step ** 2).sum()))                           # diagonal mesh spacing
    tree_grid = cKDTree(grid_pos)
    lists = tree_grid.query_ball_point(mesh_pos, 0.67 * dm)
    g2m_e = np.array([[g, m] for m, gl in enumerate(lists) for g in gl], dtype=np.int64).T.reshape(2, -1)
    tree_mesh = cKDTree(mesh_pos)
    _, nn = tree_mesh.query(grid_pos, k=min(4, len(mesh_pos)))
    nn = nn.reshape(len(grid_pos), -1)
    m2g_e = np.stack([nn.ravel(), np.repeat(np.arange(len(grid_pos)), nn.shape[1])]).astype(np.int64)

    t = torch.from_numpy
    return MeshGraph(
        n_grid=H * W, n_mesh=n * n,
        mesh_pos=t((mesh_pos / max(float(np.abs(mesh_pos).max()), 1e-12)).astype(np.float32)),
        g2m=t(g2m_e), g2m_feat=t(_edge_features(grid_pos, mesh_pos, g2m_e)),
        m2m=t(m2m_e), m2m_feat=t(_edge_features(mesh_pos, mesh_pos, m2m_e)),
        m2g=t(m2g_e), m2g_feat=t(_edge_features(mesh_pos, grid_pos, m2g_e)),
    )


def graph_path(tmp_dir: str, shape, levels: int) -> str:
    return os.path.join(str(tmp_dir), f"p4c_mesh_graph_{shape[0]}x{shape[1]}_l{levels}.pt")


# ---------------------------------------------------------------------------------------------- hierarchical mesh (HiLAM)
@dataclass
class HiMeshGraph:
    """neural-lam's hierarchical mesh: level 0 is the finest lattice (n x n), each further level is 3x coarser; `same` edges
    inside a level, `up` edges from every node to its nearest node one level up, `down` = the reversed `up` edges."""

    n_grid: int
    n_mesh: List[int]               # nodes per level
    mesh_pos: List[torch.Tensor]    # per level (n_l, 2)
    g2m: torch.Tensor               # grid -> level 0
    g2m_feat: torch.Tensor
    m2g: torch.Tensor               # level 0 -> grid
    m2g_feat: torch.Tensor
    same: List[torch.Tensor]        # per level (2, E)
    same_feat: List[torch.Tensor]
    up: List[torch.Tensor]          # level l -> l+1, l = 0 .. L-2
    up_feat: List[torch.Tensor]
    down: List[torch.Tensor]        # level l+1 -> l
    down_feat: List[torch.Tensor]

    def save(self, path: str):
        torch.save(self.__dict__, path)

    @staticmethod
    def load(path: str) -> "HiMeshGraph":
        return HiMeshGraph(**torch.load(path, weights_only=True))


def build_hierarchical_graph(meshgrid: torch.Tensor, levels: int = 0, refine: int = 3) -> HiMeshGraph:
    from scipy.spatial import cKDTree

    xy = meshgrid.detach().cpu().double().numpy()
    _, H, W = xy.shape
    grid_pos = xy.reshape(2, -1).T
    nlev = int(math.log(max(H, W)) / math.log(refine))
    nleaf = refine ** nlev
    lo, hi = grid_pos.min(0), grid_pos.max(0)
    sizes = []
    n = nleaf // refine
    while n >= 2 and (levels <= 0 or len(sizes) < levels):
        sizes.append(n)
        n //= refine
    if not sizes:
        sizes = [2]
    pos, same = [], []
    for n in sizes:
        step = (hi - lo) / n
        my, mx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        pos.append(np.stack([lo[0] + (mx.ravel() + 0.5) * step[0], lo[1] + (my.ravel() + 0.5) * step[1]], axis=1))
        same.append(_lattice_edges(n, n, 1, 0))
    up, down = [], []
    for lev in range(len(sizes) - 1):
        _, nn = cKDTree(pos[lev + 1]).query(pos[lev], k=1)
        e = np.stack([np.arange(len(pos[lev])), nn.ravel()]).astype(np.int64)
        up.append(e)
        down.append(e[::-1].copy())
    step0 = (hi - lo) / sizes[0]
    dm = float(np.sqrt((step0 ** 2).sum()))
    lists = cKDTree(grid_pos).query_ball_point(pos[0], 0.67 * dm)
    g2m_e = np.array([[g, m] for m, gl in enumerate(lists) for g in gl], dtype=np.int64).T.reshape(2, -1)
    _, nn = cKDTree(pos[0]).query(grid_pos, k=min(4, len(pos[0])))
    nn = nn.reshape(len(grid_pos), -1)
    m2g_e = np.stack([nn.ravel(), np.repeat(np.arange(len(grid_pos)), nn.shape[1])]).astype(np.int64)

    t = torch.from_numpy
    scale = max(max(float(np.abs(p).max()) for p in pos), 1e-12)
    return HiMeshGraph(
        n_grid=H * W, n_mesh=[len(p) for p in pos], mesh_pos=[t((p / scale).astype(np.float32)) for p in pos],
        g2m=t(g2m_e), g2m_feat=t(_edge_features(grid_pos, pos[0], g2m_e)),
        m2g=t(m2g_e), m2g_feat=t(_edge_features(pos[0], grid_pos, m2g_e)),
        same=[t(e) for e in same], same_feat=[t(_edge_features(p, p, e)) for p, e in zip(pos, same)],
        up=[t(e) for e in up], up_feat=[t(_edge_features(pos[l], pos[l + 1], e)) for l, e in enumerate(up)],
        down=[t(e) for e in down], down_feat=[t(_edge_features(pos[l + 1], pos[l], e)) for l, e in enumerate(down)],
    )


def hi_graph_path(tmp_dir: str, shape, levels: int) -> str:
    return os.path.join(str(tmp_dir), f"p4c_hi_mesh_graph_{shape[0]}x{shape[1]}_l{levels}.pt")
