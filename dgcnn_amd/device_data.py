"""Device-resident graph dataset and on-device mini-batch assembly.

The reference re-collates every batch on the host (PyG ``DataLoader``, /root/reference/train.py:108-109) and ships it
over PCIe; with a ~60 us training step that host loop would be >90 % of an epoch.  A TU dataset is small (COLLAB: 5 000
graphs, 372 k nodes, 24.6 M directed edges ~ 0.4 GB), so here it is uploaded ONCE (:class:`DeviceDataset`) and every
batch is assembled by one kernel launch (``dgcnn_collate``) from a list of graph ids; the only per-batch host work is
a numpy cumsum over B node / edge counts.  The resulting :class:`dgcnn_amd.batch.Batch` is bit-identical to
``collate([graphs[i] for i in ids]).to(device)``.

``DeviceLoader`` mirrors ``GraphLoader`` / the reference's ``DataLoader(data_set[idx], batch_size, shuffle)``; it keeps a
small ring of output buffers so the training loop can hold batch i+1 (look-ahead for the pipelined step) while batch
i computes.
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .batch import Batch, Graph


class DeviceDataset:
    def __init__(self, graphs: Sequence[Graph], device="cuda"):
        graphs = list(graphs.graphs) if hasattr(graphs, "graphs") else list(graphs)
        if not graphs:
            raise ValueError("empty dataset")
        self.device = torch.device(device)
        self.num_graphs = len(graphs)
        self.num_features = int(graphs[0].x.shape[1])
        nn_ = np.array([g.num_nodes for g in graphs], dtype=np.int64)
        ne_ = np.array([g.num_edges for g in graphs], dtype=np.int64)
        self.nodes_per_graph, self.edges_per_graph = nn_, ne_          # host copies: prefix sums and max hints per batch
        node_ptr = np.concatenate([[0], np.cumsum(nn_)])
        edge_ptr = np.concatenate([[0], np.cumsum(ne_)])
        self.total_nodes, self.total_edges = int(node_ptr[-1]), int(edge_ptr[-1])
        self.coalesced_undirected = all(getattr(g, "coalesced_undirected", False) for g in graphs)
        dev = self.device
        self.x_all = torch.cat([g.x for g in graphs], 0).contiguous().to(dev)
        self.ei_all = torch.cat([g.edge_index for g in graphs], 1).contiguous().to(dev)       # graph-local node ids
        self.node_ptr = torch.from_numpy(node_ptr).to(dev)
        self.edge_ptr = torch.from_numpy(edge_ptr).to(dev)
        self.y_cpu = torch.tensor([int(g.y) for g in graphs], dtype=torch.int64)
        self.y_all = self.y_cpu.to(dev)

    def __len__(self) -> int:
        return self.num_graphs

    def assemble(self, ids: np.ndarray, out: Optional[dict] = None) -> Batch:
        """Batch of graphs ``ids`` (order kept), assembled on the device by ONE launch.  ``out``: reusable buffer dict."""
        ids = np.asarray(ids, dtype=np.int64)
        B = int(ids.shape[0])
        nn_, ne_ = self.nodes_per_graph[ids], self.edges_per_graph[ids]
        meta = np.empty(3 * B + 2, dtype=np.int64)          # [node prefix (B+1) | edge prefix (B+1) | ids (B)] : one H2D
        meta[0] = 0; np.cumsum(nn_, out=meta[1:B + 1])
        meta[B + 1] = 0; np.cumsum(ne_, out=meta[B + 2:2 * B + 2])
        meta[2 * B + 2:] = ids
        N, E = int(meta[B]), int(meta[2 * B + 1])
        dev, F = self.device, self.num_features
        if out is None:
            out = {}
        def buf(name, numel, dtype):
            t = out.get(name)
            if t is None or t.numel() < numel:
                t = out[name] = torch.empty(max(int(numel * 1.25), 16), dtype=dtype, device=dev)
            return t
        meta_h = out.get("meta_h")
        if meta_h is None or meta_h.numel() < meta.shape[0]:
            meta_h = out["meta_h"] = torch.empty(max(2 * meta.shape[0], 64), dtype=torch.int64).pin_memory()
        evt = out.get("evt")
        if evt is not None:
            evt.synchronize()            # the previous async upload from this pinned buffer has been consumed (normally long ago)
        else:
            evt = out["evt"] = torch.cuda.Event()
        meta_h[:meta.shape[0]].copy_(torch.from_numpy(meta))
        meta_d = buf("meta_d", meta.shape[0], torch.int64)
        meta_d[:meta.shape[0]].copy_(meta_h[:meta.shape[0]], non_blocking=True)
        evt.record()
        x = buf("x", N * F, torch.float32)[:N * F].view(N, F)
        ei = buf("ei", 2 * max(E, 1), torch.int64)[:2 * E].view(2, E)
        bt = buf("batch", N, torch.int64)[:N]
        y = buf("y", B, torch.int64)[:B]
        stream = torch.cuda.current_stream(dev).cuda_stream
        mp = meta_d.data_ptr()
        _lib.check(_lib.lib().dgcnn_collate(B, F, N, E, self.total_edges, mp + 8 * (2 * B + 2), mp,
                                            mp + 8 * (B + 1), self.x_all.data_ptr(),
                                            self.ei_all.data_ptr() if self.total_edges else None, self.node_ptr.data_ptr(),
                                            self.edge_ptr.data_ptr(), self.y_all.data_ptr(), x.data_ptr(),
                                            ei.data_ptr() if E else None, bt.data_ptr(), y.data_ptr(), stream),
                   "dgcnn_collate")
        return Batch(x, ei, bt, y, B, self.coalesced_undirected, int(nn_.max()), int(ne_.max()))


class DeviceLoader:
    """``DataLoader(data_set[idx], batch_size, shuffle)`` (train.py:108-109) over a :class:`DeviceDataset`.

    ``indices``: the subset (a fold's train or test ids); ``shuffle`` draws a fresh permutation per epoch from
    ``generator``.  ``ring`` output-buffer sets are cycled, so a yielded batch stays valid while the next ``ring - 1``
    batches are produced (the training loop's one-batch look-ahead needs 2; default 3)."""

    def __init__(self, dataset: DeviceDataset, batch_size: int, indices=None, shuffle: bool = False,
                 generator: Optional[torch.Generator] = None, ring: int = 3):
        self.ds = dataset
        self.batch_size = int(batch_size)
        self.indices = np.arange(len(dataset), dtype=np.int64) if indices is None else \
            np.asarray(torch.as_tensor(indices).cpu().numpy() if not isinstance(indices, np.ndarray) else indices, dtype=np.int64)
        self.shuffle = shuffle
        self.generator = generator
        self.num_samples = int(self.indices.shape[0])
        self._bufs: List[dict] = [{} for _ in range(max(2, ring))]
        self._k = 0

    def __len__(self) -> int:
        return (self.num_samples + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Batch]:
        n = self.num_samples
        order = self.indices[torch.randperm(n, generator=self.generator).numpy()] if self.shuffle else self.indices
        for i in range(0, n, self.batch_size):
            out = self._bufs[self._k % len(self._bufs)]
            self._k += 1
            yield self.ds.assemble(order[i:i + self.batch_size], out)
