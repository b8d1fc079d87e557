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

    def __del__(self):
        try:
            L = _lib.lib()
            for ev in getattr(self, "_events", []):
                L.dgcnn_event_destroy(ev)
        except Exception:
            pass

    def assemble(self, ids: np.ndarray, out: Optional[dict] = None, ids_dev_ptr: Optional[int] = None) -> Batch:
        """Batch of graphs ``ids`` (order kept), assembled on the device: ONE C call (``dgcnn_collate_ids``: prefix sums
        in C, a few-hundred-byte asynchronous upload, one kernel launch).  ``out``: reusable buffer dict (a ring slot)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        B = int(ids.shape[0])
        dev, F = self.device, self.num_features
        if out is None:
            out = {}
        st = out.get("state")
        if st is None or st["B"] < B:
            ev = _lib.c_void_p()
            _lib.check(_lib.lib().dgcnn_event_create(_lib.ctypes.byref(ev)), "dgcnn_event_create")
            self.__dict__.setdefault("_events", []).append(ev)
            st = out["state"] = {"B": max(B, 64), "ev": ev, "capN": 0, "capE": 0,
                                 "meta_h": torch.empty(3 * max(B, 64) + 2, dtype=torch.int64).pin_memory(),
                                 "meta_d": torch.empty(3 * max(B, 64) + 2, dtype=torch.int64, device=dev),
                                 "y": torch.empty(max(B, 64), dtype=torch.int64, device=dev),
                                 "sizes": np.zeros(4, dtype=np.int64)}
        sizes = st["sizes"]
        L = _lib.lib()
        stream = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
        for attempt in (0, 1):
            rc = L.dgcnn_collate_ids(B, F, ids.ctypes.data, ids_dev_ptr, self.nodes_per_graph.ctypes.data,
                                     self.edges_per_graph.ctypes.data,
                                     self.num_graphs, st["meta_h"].data_ptr(), st["meta_d"].data_ptr(), st["ev"],
                                     self.total_edges, self.x_all.data_ptr(),
                                     self.ei_all.data_ptr() if self.total_edges else None, self.node_ptr.data_ptr(),
                                     self.edge_ptr.data_ptr(), self.y_all.data_ptr(), st["capN"], st["capE"],
                                     st["x"].data_ptr() if st["capN"] else None, st["ei"].data_ptr() if st["capE"] else None,
                                     st["bt"].data_ptr() if st["capN"] else None, st["y"].data_ptr(), sizes.ctypes.data, stream)
            if rc == 0:
                break
            if rc != -3 or attempt == 1:
                _lib.check(rc, "dgcnn_collate_ids")
            # buffers too small for this batch: grow (with slack) and call again
            N, E = int(sizes[0]), int(sizes[1])
            if N > st["capN"]:
                st["capN"] = int(N * 1.25) + 16
                st["x"] = torch.empty(st["capN"] * F, dtype=torch.float32, device=dev)
                st["bt"] = torch.empty(st["capN"], dtype=torch.int64, device=dev)
            if E > st["capE"]:
                st["capE"] = int(E * 1.25) + 16
                st["ei"] = torch.empty(2 * st["capE"], dtype=torch.int64, device=dev)
        N, E = int(sizes[0]), int(sizes[1])
        x = st["x"][:N * F].view(N, F)
        ei = st["ei"][:2 * E].view(2, E) if E else torch.zeros(2, 0, dtype=torch.int64, device=dev)
        return Batch(x, ei, st["bt"][:N], st["y"][:B], B, self.coalesced_undirected, int(sizes[2]), int(sizes[3]))


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
        order = np.ascontiguousarray(order, dtype=np.int64)
        # the epoch's permutation goes to the device ONCE; batches then need no upload of their own (B <= 256).
        # Two generations are kept alive: the previous epoch's last batches may still be in flight.
        order_dev = torch.from_numpy(order).to(self.ds.device)
        self._order_keep = (getattr(self, "_order_keep", (None, None))[1], order_dev)
        base = order_dev.data_ptr()
        for i in range(0, n, self.batch_size):
            out = self._bufs[self._k % len(self._bufs)]
            self._k += 1
            yield self.ds.assemble(order[i:i + self.batch_size], out, base + 8 * i)
