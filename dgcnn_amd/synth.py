"""Seeded synthetic graph generator for the BASELINE.json configs (host logic).

The real TU datasets are not shipped with the reference (only the 10-fold index
files are; ``TUDataset`` would download them, /root/reference/train.py:81-86) and
there is no network, so every workload here is synthetic, shaped after the dataset
statistics in SURVEY.md §8(d) D2.  Feature layout follows what
``TUDataset(use_node_attr=True)`` + ``Indegree`` produce
(/root/reference/utils.py:18-33): [continuous attrs, one-hot labels, in-degree /
per-graph max in-degree] with the degree column LAST; label-less sets (COLLAB,
IMDB) have that single degree column only (/root/reference/utils.py:30-31).

Graph ``g`` of a workload is drawn from ``numpy.random.default_rng(seed + g)``
(base seed 324 = the reference's default, /root/reference/train.py:24), so any
sub-range of graphs can be regenerated independently (used by data-parallel ranks).

Edges: simple undirected G(n, p) with p = mean_deg / (n - 1), both directions
emitted, sorted by (src, dst) like a coalesced TU file, no self loops, never empty.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from .batch import Batch, Graph, collate, indegree_feature

BASE_SEED = 324


@dataclass(frozen=True)
class Shape:
    name: str
    num_features: int
    num_classes: int
    n_attr: int          # continuous N(0,1) attribute columns
    n_onehot: int        # one-hot label columns
    mean_deg: float      # target mean degree; <=0 means "dense ego-net": min(66, n-1)
    draw_n: Callable[[np.random.Generator], int]


def _clip_round(v: float, lo: int, hi: int) -> int:
    return int(min(max(int(round(v)), lo), hi))


SHAPES: Dict[str, Shape] = {
    # MUTAG-shape: n~U{10..28}, deg 2.2, F=8 (one-hot 7 + deg), C=2
    "MUTAG": Shape("MUTAG", 8, 2, 0, 7, 2.2, lambda r: int(r.integers(10, 29))),
    # PROTEINS-shape: n = clip(round(LogN(ln 26, 0.9)), 4, 620), deg 3.7, F=5 (1 attr + 3 one-hot + deg)
    "PROTEINS": Shape("PROTEINS", 5, 2, 1, 3, 3.7,
                      lambda r: _clip_round(r.lognormal(np.log(26.0), 0.9), 4, 620)),
    # COLLAB-shape (BASELINE cfg): n = clip(round(N(75,30)), 32, 492), deg ~37, F=1, C=3
    "COLLAB": Shape("COLLAB", 1, 3, 0, 0, 37.0,
                    lambda r: _clip_round(r.normal(75.0, 30.0), 32, 492)),
    # COLLAB real-like: dense ego nets, deg ~ min(66, n-1)
    "COLLAB_REAL": Shape("COLLAB_REAL", 1, 3, 0, 0, -66.0,
                         lambda r: _clip_round(r.normal(75.0, 30.0), 32, 492)),
    # DD-shape: n = clip(round(LogN(ln 240, 0.6)), 30, 5748), deg 5, F=90 (one-hot 89 + deg), C=2
    "DD": Shape("DD", 90, 2, 0, 89, 5.0,
                lambda r: _clip_round(r.lognormal(np.log(240.0), 0.6), 30, 5748)),
    # NCI1-shape (not a BASELINE config; the one TU shape ABOVE the aggregate-first width with small graphs): F=38, C=2
    "NCI1": Shape("NCI1", 38, 2, 0, 37, 2.16,
                  lambda r: _clip_round(r.normal(30.0, 10.0), 8, 111)),
    # IMDB-B-shape (not a BASELINE config; used as an extra small case): F=1, C=2
    "IMDB": Shape("IMDB", 1, 2, 0, 0, 9.0,
                  lambda r: _clip_round(r.normal(20.0, 6.0), 12, 136)),
}


def _gnp_edges(rng: np.random.Generator, n: int, p: float) -> np.ndarray:
    """Undirected simple G(n,p) as an upper-triangular pair list [m,2] (i<j)."""
    p = float(min(max(p, 0.0), 1.0))
    if n < 2:
        return np.zeros((0, 2), dtype=np.int64)
    if n <= 2048:
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(iu.shape[0]) < p
        return np.stack([iu[keep], ju[keep]], 1).astype(np.int64)
    # large sparse graphs: draw the edge count, then distinct random pairs
    total = n * (n - 1) // 2
    m = int(rng.binomial(total, p))
    pairs = set()
    while len(pairs) < m:
        need = m - len(pairs)
        a = rng.integers(0, n, size=2 * need + 16)
        b = rng.integers(0, n, size=2 * need + 16)
        for i, j in zip(a.tolist(), b.tolist()):
            if i == j:
                continue
            if i > j:
                i, j = j, i
            pairs.add((i, j))
            if len(pairs) >= m:
                break
    arr = np.array(sorted(pairs), dtype=np.int64).reshape(-1, 2)
    return arr


def make_graph(shape: Shape, g: int, seed: int = BASE_SEED, force_n: Optional[int] = None,
               labels: str = "random") -> Graph:
    """``labels="random"``: class drawn independently of the graph (throughput workloads; nothing to learn).
    ``labels="structure"``: the class sets the edge density (class c -> mean degree x (0.6 + 0.8 c)), a task DGCNN
    learns quickly from the degree feature -- used by the end-to-end training tests and the driver's synthetic mode."""
    rng = np.random.default_rng(seed + g)
    n = int(force_n) if force_n is not None else shape.draw_n(rng)
    n = max(n, 2)
    deg = shape.mean_deg if shape.mean_deg > 0 else min(-shape.mean_deg, n - 1)
    y_struct = None
    if labels == "structure":
        y_struct = int(np.random.default_rng(seed * 7919 + g).integers(0, shape.num_classes))
        deg = deg * (0.6 + 0.8 * y_struct)
    elif labels != "random":
        raise ValueError("labels must be 'random' or 'structure'")
    p = deg / (n - 1)
    und = _gnp_edges(rng, n, p)
    tries = 0
    while und.shape[0] == 0:                      # never emit an edgeless graph
        tries += 1
        und = _gnp_edges(rng, n, min(1.0, p * (1 + tries)))
    src = np.concatenate([und[:, 0], und[:, 1]])
    dst = np.concatenate([und[:, 1], und[:, 0]])
    order = np.lexsort((dst, src))                # sorted by (src, dst)
    ei = torch.from_numpy(np.stack([src[order], dst[order]], 0).astype(np.int64))
    cols: List[torch.Tensor] = []
    if shape.n_attr:
        cols.append(torch.from_numpy(rng.standard_normal((n, shape.n_attr)).astype(np.float32)))
    if shape.n_onehot:
        lab = rng.integers(0, shape.n_onehot, size=n)
        oh = np.zeros((n, shape.n_onehot), dtype=np.float32)
        oh[np.arange(n), lab] = 1.0
        cols.append(torch.from_numpy(oh))
    feat = torch.cat(cols, 1) if cols else None
    x = indegree_feature(ei, n, feat).contiguous()
    assert x.shape[1] == shape.num_features, (x.shape, shape)
    y = int(rng.integers(0, shape.num_classes))
    if y_struct is not None:
        y = y_struct
    return Graph(x=x, edge_index=ei, y=y, coalesced_undirected=True)


def make_graphs(name: str, count: int, start: int = 0, seed: int = BASE_SEED,
                force_first_n: Optional[int] = None, labels: str = "random") -> List[Graph]:
    shape = SHAPES[name]
    out = []
    for g in range(start, start + count):
        fn = force_first_n if (force_first_n is not None and g == start) else None
        out.append(make_graph(shape, g, seed, fn, labels))
    return out


def make_batch(name: str, batch_size: int = 50, start: int = 0, seed: int = BASE_SEED,
               force_first_n: Optional[int] = None) -> Batch:
    """One collated batch of ``batch_size`` synthetic graphs of workload ``name``."""
    return collate(make_graphs(name, batch_size, start, seed, force_first_n))


def make_batches(name: str, num_graphs: int, batch_size: int, start: int = 0,
                 seed: int = BASE_SEED) -> List[Batch]:
    """``num_graphs`` graphs chopped into consecutive batches (last may be short,
    like the reference loader without ``drop_last``, /root/reference/train.py:108-109)."""
    graphs = make_graphs(name, num_graphs, start, seed)
    return [collate(graphs[i:i + batch_size]) for i in range(0, num_graphs, batch_size)]


def tile_batch(b: Batch, times: int) -> Batch:
    """Repeat a batch ``times`` times as one larger disjoint union (batch-size sweeps)."""
    N = b.num_nodes
    xs = b.x.repeat(times, 1)
    eis = torch.cat([b.edge_index + t * N for t in range(times)], 1)
    bs = torch.cat([b.batch + t * b.num_graphs for t in range(times)], 0)
    ys = None if b.y is None else b.y.repeat(times)
    return Batch(xs, eis, bs, ys, num_graphs=b.num_graphs * times,
                 coalesced_undirected=b.coalesced_undirected, max_nodes=b.max_nodes,
                 max_edges=b.max_edges)
