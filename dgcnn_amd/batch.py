"""Batched-graph container and collate (host logic, no GPU needed).

Restates the input-producer contract of the reference hot path:
the reference feeds ``Model.forward`` a PyG ``Batch`` built by
``torch_geometric.loader.DataLoader`` (/root/reference/train.py:108-109); the model
reads exactly three attributes of it -- ``data.x``, ``data.edge_index``,
``data.batch`` (/root/reference/model.py:27) -- and the training loop reads
``sample.y`` and calls ``sample.to(device)`` (/root/reference/train.py:36).

PyG is not available at run time on the GPU box, so this module provides the
minimal duck-typed equivalent:

* :class:`Graph`   -- one graph (x [n,F] f32, edge_index [2,e] i64, y int).
* :class:`Batch`   -- disjoint union of graphs: ``x`` row-concatenated,
  ``edge_index`` concatenated with per-graph node offsets (block-diagonal
  adjacency, graphs contiguous), ``batch`` sorted graph ids, ``y`` [B].
* :func:`collate`  -- list[Graph] -> Batch (what PyG's collate does for these fields).
* :func:`indegree_feature` -- the ``Indegree`` pre-transform of
  /root/reference/utils.py:18-33 (in-degree / per-graph max in-degree appended
  as the LAST feature column, or the only column when the graph has no features).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

import torch


@dataclass
class Graph:
    x: torch.Tensor            # [n, F] float32
    edge_index: torch.Tensor   # [2, e] int64, row0 = source, row1 = target
    y: int = 0
    coalesced_undirected: bool = False   # sorted by (src,dst), no dups/self loops, both directions

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])


class Batch:
    """Duck-typed stand-in for the PyG ``Batch`` the reference model consumes.

    Only the attributes the hot path touches exist: ``x``, ``edge_index``,
    ``batch``, ``y`` (+ ``num_graphs``).  ``to(device)`` mirrors
    /root/reference/train.py:36 (``sample.to(device)``) and returns a new Batch.
    """

    __slots__ = ("x", "edge_index", "batch", "y", "num_graphs", "coalesced_undirected", "max_nodes", "max_edges")

    def __init__(self, x, edge_index, batch, y=None, num_graphs: Optional[int] = None,
                 coalesced_undirected: bool = False, max_nodes: int = 0, max_edges: int = 0):
        if x.dim() != 2:
            raise ValueError(f"x must be [N,F], got {tuple(x.shape)}")
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError(f"edge_index must be [2,E], got {tuple(edge_index.shape)}")
        if batch.dim() != 1 or batch.shape[0] != x.shape[0]:
            raise ValueError("batch must be [N] and match x")
        self.x = x
        self.edge_index = edge_index
        self.batch = batch
        self.y = y
        if num_graphs is None:
            if y is not None:
                num_graphs = int(y.shape[0])
            else:
                num_graphs = int(batch[-1].item()) + 1 if batch.numel() else 0
        self.num_graphs = int(num_graphs)
        # host-side promise about the edge list layout (see DGCNN_FLAG_COALESCED_UNDIRECTED in
        # include/dgcnn_hip.h); verified on the device, never trusted blindly
        self.coalesced_undirected = bool(coalesced_undirected)
        # host-known upper bound of the node count of any single graph (0 = unknown); lets the
        # forward pick the graph-per-workgroup kernel without a device sync.  Verified on the device.
        self.max_nodes = int(max_nodes)
        self.max_edges = int(max_edges)     # same, for the directed-edge count of any single graph

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def to(self, device, non_blocking: bool = False) -> "Batch":
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)
        return Batch(mv(self.x), mv(self.edge_index), mv(self.batch), mv(self.y), self.num_graphs,
                     self.coalesced_undirected, self.max_nodes, self.max_edges)

    def pin_memory(self) -> "Batch":
        mv = lambda t: None if t is None else t.pin_memory()
        return Batch(mv(self.x), mv(self.edge_index), mv(self.batch), mv(self.y), self.num_graphs,
                     self.coalesced_undirected, self.max_nodes, self.max_edges)

    def __repr__(self) -> str:
        return (f"Batch(graphs={self.num_graphs}, nodes={self.num_nodes}, "
                f"edges={self.num_edges}, F={self.x.shape[1]}, device={self.x.device})")


def collate(graphs: Sequence[Graph]) -> Batch:
    """Disjoint union of ``graphs`` (PyG collate semantics for x/edge_index/batch/y)."""
    if len(graphs) == 0:
        raise ValueError("cannot collate an empty list of graphs")
    xs, eis, bs, ys = [], [], [], []
    off = 0
    for g, gr in enumerate(graphs):
        n = gr.num_nodes
        xs.append(gr.x)
        eis.append(gr.edge_index + off)
        bs.append(torch.full((n,), g, dtype=torch.int64))
        ys.append(int(gr.y))
        off += n
    # concatenating coalesced undirected graphs with growing node offsets keeps the union coalesced
    cu = all(getattr(gr, "coalesced_undirected", False) for gr in graphs)
    return Batch(torch.cat(xs, 0).contiguous(),
                 torch.cat(eis, 1).contiguous(),
                 torch.cat(bs, 0),
                 torch.tensor(ys, dtype=torch.int64),
                 num_graphs=len(graphs), coalesced_undirected=cu,
                 max_nodes=max(gr.num_nodes for gr in graphs), max_edges=max(gr.num_edges for gr in graphs))


def indegree_feature(edge_index: torch.Tensor, num_nodes: int,
                     x: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``Indegree(norm=True, max_value=None, cat=True)`` of /root/reference/utils.py:18-33.

    deg = in-degree counted on ``edge_index[1]``; divided by the graph's max
    in-degree; appended as the last column of ``x`` (or returned alone).
    A graph with no edges gives 0/0 = NaN in the reference; we keep that
    behaviour out of the generator by never emitting edgeless graphs.
    """
    deg = torch.zeros(num_nodes, dtype=torch.float32)
    if edge_index.numel():
        deg.scatter_add_(0, edge_index[1], torch.ones(edge_index.shape[1], dtype=torch.float32))
    deg = deg / deg.max()
    deg = deg.view(-1, 1)
    if x is not None:
        x = x.view(-1, 1) if x.dim() == 1 else x
        return torch.cat([x, deg.to(x.dtype)], dim=-1)
    return deg


def split_batch(b: Batch, parts: int) -> List[Batch]:
    """Split a Batch into ``parts`` contiguous graph ranges, balanced by
    sum(nodes + edges) per graph rather than by graph count (SURVEY.md §8 E1:
    degree skew).  Every part is non-empty when ``parts <= num_graphs``.
    Host-side; used by the data-parallel sharding in :mod:`dgcnn_amd.dist`.
    """
    B = b.num_graphs
    if parts <= 0:
        raise ValueError("parts must be positive")
    if parts > B:
        raise ValueError(f"cannot split {B} graphs into {parts} non-empty parts")
    batch = b.batch.cpu()
    ei = b.edge_index.cpu()
    n_per = torch.bincount(batch, minlength=B)
    e_graph = batch[ei[1]] if ei.numel() else torch.zeros(0, dtype=torch.int64)
    e_per = torch.bincount(e_graph, minlength=B)
    cost = (n_per + e_per).to(torch.float64)
    csum = torch.cumsum(cost, 0)
    total = float(csum[-1])
    cuts = [0]
    for p in range(1, parts):
        target = total * p / parts
        g = int(torch.searchsorted(csum, torch.tensor(target, dtype=torch.float64)).item()) + 1
        g = max(g, cuts[-1] + 1)            # non-empty
        g = min(g, B - (parts - p))         # leave >=1 graph for each later part
        cuts.append(g)
    cuts.append(B)
    node_ptr = torch.zeros(B + 1, dtype=torch.int64)
    node_ptr[1:] = torch.cumsum(n_per, 0)
    out = []
    for p in range(parts):
        g0, g1 = cuts[p], cuts[p + 1]
        n0, n1 = int(node_ptr[g0]), int(node_ptr[g1])
        emask = (e_graph >= g0) & (e_graph < g1)
        sub_ei = ei[:, emask] - n0
        dev = b.x.device
        out.append(Batch(b.x[n0:n1], sub_ei.to(dev), (b.batch[n0:n1] - g0),
                         None if b.y is None else b.y[g0:g1], num_graphs=g1 - g0,
                         coalesced_undirected=b.coalesced_undirected,
                         max_nodes=int(n_per[g0:g1].max()), max_edges=int(e_per[g0:g1].max())))
    return out
