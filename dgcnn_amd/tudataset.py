"""TU-format graph-classification datasets, the ``Indegree`` pre-transform, 10-fold index files and a
mini-batch loader (host logic).

What the reference does with PyG (/root/reference/train.py:81-86,103-109, /root/reference/utils.py:18-33):

    data_set = TUDataset(f'data/{name}', name, pre_transform=Indegree(), use_node_attr=True)
    train_idx = np.loadtxt('data/%s/10fold_idx/train_idx-%d.txt' % (name, fold), dtype=np.int32)
    DataLoader(data_set[train_idx], batch_size, shuffle=True)

PyG and the raw datasets are not available at run time (no network), so this module restates the parts
of that pipeline that touch the hot path's INPUT LAYOUT, from the published TU file format
(https://chrsmrrs.github.io/datasets/docs/format/) and PyG's documented ``read_tu_data`` semantics:

* ``<DS>_A.txt``                 one "row, col" pair per line, 1-based GLOBAL node ids (directed entries)
* ``<DS>_graph_indicator.txt``   graph id (1-based) of every node, nodes of a graph contiguous
* ``<DS>_graph_labels.txt``      one class label per graph (mapped to 0..C-1 in sorted order of the distinct values)
* ``<DS>_node_labels.txt``       optional; per column shifted to start at 0 and one-hot encoded
* ``<DS>_node_attributes.txt``   optional continuous attributes (``use_node_attr=True`` keeps them), comma separated

Feature layout = [attributes, one-hot labels, in-degree / max in-degree of the graph] (degree LAST), or
the degree column alone for label-less sets such as COLLAB / IMDB (utils.py:30-31).  Edge lists are made
0-based per graph, self loops dropped and coalesced (sorted by (src,dst), duplicates removed) as PyG does,
so a symmetric TU file yields ``coalesced_undirected`` graphs (the fast graph-prep path).
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .batch import Batch, Graph, collate, indegree_feature


def _find(root: str, name: str, suffix: str) -> Optional[str]:
    for d in (os.path.join(root, name, "raw"), os.path.join(root, "raw"), os.path.join(root, name), root):
        p = os.path.join(d, f"{name}_{suffix}.txt")
        if os.path.exists(p):
            return p
    return None


def _loadtxt(path: str, dtype, ncols_hint: Optional[int] = None) -> np.ndarray:
    arr = np.loadtxt(path, delimiter=",", dtype=dtype, ndmin=2)
    return arr


class TUData:
    """A TU dataset held as a list of :class:`dgcnn_amd.batch.Graph` (features already include the
    ``Indegree`` column).  ``num_features`` / ``num_classes`` are what the reference reads from
    ``data_set`` at /root/reference/train.py:87,97."""

    def __init__(self, graphs: List[Graph], num_classes: int, name: str = ""):
        self.graphs = graphs
        self.num_classes = int(num_classes)
        self.num_features = int(graphs[0].x.shape[1]) if graphs else 0
        self.name = name

    def __len__(self) -> int:
        return len(self.graphs)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return self.graphs[int(idx)]
        idx = torch.as_tensor(idx).tolist() if not isinstance(idx, (list, tuple)) else list(idx)
        return TUData([self.graphs[int(i)] for i in idx], self.num_classes, self.name)


def read_tu_dataset(root: str, name: str, use_node_attr: bool = True, add_indegree: bool = True) -> TUData:
    """Parse ``<root>/<name>/raw/<name>_*.txt`` (also ``<root>/<name>/`` or ``<root>/``)."""
    pa = _find(root, name, "A")
    pi = _find(root, name, "graph_indicator")
    py = _find(root, name, "graph_labels")
    if not (pa and pi and py):
        raise FileNotFoundError(
            f"TU files for {name!r} not found under {root!r} (need {name}_A.txt, {name}_graph_indicator.txt, "
            f"{name}_graph_labels.txt); the reference would download them (train.py:81-86), there is no network here")
    A = _loadtxt(pa, np.int64) - 1                               # [E,2] 0-based global ids
    gid = _loadtxt(pi, np.int64).reshape(-1) - 1                 # [N]
    ylab = _loadtxt(py, np.int64).reshape(-1)
    if ylab.ndim != 1:
        ylab = ylab[:, 0]
    N = gid.shape[0]
    if np.any(np.diff(gid) < 0):
        raise ValueError("graph_indicator must be sorted (nodes of a graph contiguous)")
    G = int(gid.max()) + 1
    cols = []
    pattr = _find(root, name, "node_attributes")
    if use_node_attr and pattr:
        cols.append(_loadtxt(pattr, np.float32).reshape(N, -1))
    plab = _find(root, name, "node_labels")
    if plab:
        lab = _loadtxt(plab, np.int64).reshape(N, -1)
        lab = lab - lab.min(axis=0, keepdims=True)
        for c in range(lab.shape[1]):
            k = int(lab[:, c].max()) + 1
            oh = np.zeros((N, k), dtype=np.float32)
            oh[np.arange(N), lab[:, c]] = 1.0
            cols.append(oh)
    xall = np.concatenate(cols, axis=1) if cols else None
    # labels -> 0..C-1 in sorted order of distinct values (torch.unique(return_inverse) in PyG)
    uniq, yinv = np.unique(ylab, return_inverse=True)
    node_ptr = np.searchsorted(gid, np.arange(G + 1))
    # edges: drop self loops, group by graph of the source, coalesce
    src, dst = A[:, 0], A[:, 1]
    keep = src != dst
    src, dst = src[keep], dst[keep]
    if np.any(gid[src] != gid[dst]):
        raise ValueError("an edge connects two different graphs")
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    if src.size:
        dup = np.concatenate([[False], (src[1:] == src[:-1]) & (dst[1:] == dst[:-1])])
        src, dst = src[~dup], dst[~dup]
    eg = gid[src] if src.size else np.zeros(0, dtype=np.int64)
    edge_ptr = np.searchsorted(eg, np.arange(G + 1))
    graphs: List[Graph] = []
    for g in range(G):
        n0, n1 = int(node_ptr[g]), int(node_ptr[g + 1])
        e0, e1 = int(edge_ptr[g]), int(edge_ptr[g + 1])
        ei = torch.from_numpy(np.stack([src[e0:e1] - n0, dst[e0:e1] - n0], 0).astype(np.int64))
        x = None if xall is None else torch.from_numpy(xall[n0:n1].copy())
        if add_indegree:
            x = indegree_feature(ei, n1 - n0, x).contiguous()
        elif x is None:
            x = torch.ones(n1 - n0, 1)
        # symmetric edge set?  (coalesced already)  -> fast graph-prep path may be promised
        fw = set(zip(ei[0].tolist(), ei[1].tolist()))
        sym = all((d, s) in fw for s, d in fw)
        graphs.append(Graph(x=x, edge_index=ei, y=int(yinv[g]), coalesced_undirected=bool(sym)))
    return TUData(graphs, num_classes=len(uniq), name=name)


def write_tu_dataset(root: str, name: str, graphs: Sequence[Graph], node_labels: Optional[Sequence[np.ndarray]] = None,
                     node_attrs: Optional[Sequence[np.ndarray]] = None, class_values: Optional[Sequence[int]] = None) -> str:
    """Write graphs in TU text format under ``<root>/<name>/raw`` (tests, and exporting synthetic sets)."""
    d = os.path.join(root, name, "raw")
    os.makedirs(d, exist_ok=True)
    off = 0
    with open(os.path.join(d, f"{name}_A.txt"), "w") as fa, \
            open(os.path.join(d, f"{name}_graph_indicator.txt"), "w") as fi, \
            open(os.path.join(d, f"{name}_graph_labels.txt"), "w") as fy:
        for g, gr in enumerate(graphs):
            for s, t in zip(gr.edge_index[0].tolist(), gr.edge_index[1].tolist()):
                fa.write(f"{s + off + 1}, {t + off + 1}\n")
            for _ in range(gr.num_nodes):
                fi.write(f"{g + 1}\n")
            fy.write(f"{class_values[gr.y] if class_values is not None else gr.y}\n")
            off += gr.num_nodes
    if node_labels is not None:
        with open(os.path.join(d, f"{name}_node_labels.txt"), "w") as f:
            for lab in node_labels:
                for v in np.asarray(lab).reshape(-1).tolist():
                    f.write(f"{int(v)}\n")
    if node_attrs is not None:
        with open(os.path.join(d, f"{name}_node_attributes.txt"), "w") as f:
            for at in node_attrs:
                for row in np.asarray(at, dtype=np.float64).reshape(len(at), -1):
                    f.write(", ".join(repr(float(v)) for v in row) + "\n")
    return d


def read_fold_indices(data_dir: str, fold: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``data/<DS>/10fold_idx/{train,test}_idx-<fold>.txt`` -> (train_idx, test_idx), as
    /root/reference/train.py:103-106 (``np.loadtxt(..., dtype=np.int32)``)."""
    tr = np.loadtxt(os.path.join(data_dir, "10fold_idx", f"train_idx-{fold}.txt"), dtype=np.int32)
    te = np.loadtxt(os.path.join(data_dir, "10fold_idx", f"test_idx-{fold}.txt"), dtype=np.int32)
    return torch.as_tensor(tr, dtype=torch.long).reshape(-1), torch.as_tensor(te, dtype=torch.long).reshape(-1)


def make_fold_indices(num_graphs: int, fold: int, folds: int = 10, seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Deterministic k-fold partition for datasets that ship no index files (synthetic runs)."""
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(num_graphs, generator=g)
    test = perm[(fold - 1)::folds]
    mask = torch.ones(num_graphs, dtype=torch.bool)
    mask[test] = False
    return torch.arange(num_graphs)[mask], test.sort().values


class GraphLoader:
    """``DataLoader(dataset, batch_size, shuffle)`` of /root/reference/train.py:108-109 for this build's
    ``Graph`` lists: yields collated :class:`Batch` objects (last batch may be short, no ``drop_last``),
    optionally moved to ``device``.  ``len(loader)`` = number of batches, ``loader.num_samples`` = graphs."""

    def __init__(self, graphs: Sequence[Graph], batch_size: int, shuffle: bool = False,
                 generator: Optional[torch.Generator] = None, device=None):
        self.graphs = list(graphs.graphs) if isinstance(graphs, TUData) else list(graphs)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.generator = generator
        self.device = device
        self.num_samples = len(self.graphs)

    def __len__(self) -> int:
        return (self.num_samples + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Batch]:
        n = self.num_samples
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for i in range(0, n, self.batch_size):
            b = collate([self.graphs[j] for j in order[i:i + self.batch_size]])
            yield b.to(self.device) if self.device is not None else b
