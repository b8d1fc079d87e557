"""ORACLE (test infrastructure, NOT product code) -- fp32 torch-op restatement of the
reference hot path, op for op.

PARITY UNPINNED: the reference (/root/reference/model.py) delegates all graph
arithmetic to PyTorch Geometric (``GCNConv``, ``SortAggregation``,
``remove_self_loops``; /root/reference/model.py:5-6).  PyG is an un-vendored,
un-pinned dependency (README only says ``pip install torch-geometric``,
/root/reference/README.md:15-22; API use bounds it to >= 2.1), it is not installed
in the build container, there is no network, and the reference ships no tests or
golden vectors.  This file therefore restates the *published* PyG 2.x algorithm for
those three symbols from knowledge of its public source, anchored on the reference's
own call sites; it could not be diffed against PyG itself.  What pins it instead:
README parameter-count KATs, shape KATs, closed-form hand KATs and the independent
fp64 dense formulation in ``ref_dense.py`` (see tests/test_oracle_*.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``dgcnn_amd``) never does.

Op sequence followed (file:line are into /root/reference):

* ``remove_self_loops``            model.py:28   -> :func:`remove_self_loops`
* ``GCNConv.forward`` x4           model.py:30-33 -> :func:`gcn_norm` + :func:`gcn_conv`
    gcn_norm : append one self loop per node at the END of the edge list (weight 1),
               deg = scatter_add(w at target), dis = deg^-1/2 (inf -> 0),
               w = dis[src] * w * dis[dst]          (recomputed per layer, cached=False)
    linear   : h = x @ W^T (no bias) BEFORE aggregation
    propagate: msg = w[:,None] * h.index_select(0, src); out = zeros.scatter_add_(0, dst, msg)
    bias     : out + b
* ``torch.tanh`` x4, ``torch.cat`` model.py:30-34
* ``SortAggregation(k=30)``        model.py:17,35 -> :func:`sort_pool`
* tail (Conv1d/MaxPool1d/Linear/Dropout/log_softmax) model.py:36-43
* training step (NLL mean, backward, Adam defaults) train.py:37-42 -> :func:`train_step`
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

K_SORT = 30          # model.py:17
HID = 32             # model.py:13-15
CAT = 97             # 32+32+32+1, model.py:34


def remove_self_loops(edge_index: torch.Tensor) -> torch.Tensor:
    """PyG ``remove_self_loops``: boolean mask ``row != col``, order preserved (model.py:28)."""
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask]


def gcn_norm(edge_index: torch.Tensor, num_nodes: int, dtype=torch.float32):
    """PyG ``gcn_norm(add_self_loops=True, improved=False, flow='source_to_target')``.

    Input has no self loops (they were removed at model.py:28), so
    ``add_remaining_self_loops`` == append ``arange(N)`` loops with weight 1 at the end.
    """
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    ei = torch.cat([edge_index, torch.stack([loop, loop], 0)], dim=1)
    w = torch.ones(ei.shape[1], dtype=dtype, device=edge_index.device)
    row, col = ei[0], ei[1]
    deg = torch.zeros(num_nodes, dtype=dtype, device=edge_index.device).scatter_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0)
    w = dis[row] * w * dis[col]
    return ei, w


def gcn_conv(x: torch.Tensor, edge_index: torch.Tensor, weight: torch.Tensor,
             bias: torch.Tensor) -> torch.Tensor:
    """One ``GCNConv.forward`` (normalize=True, bias=True, cached=False)."""
    n = x.shape[0]
    ei, w = gcn_norm(edge_index, n, x.dtype)
    row, col = ei[0], ei[1]
    h = x @ weight.t()
    msg = w.view(-1, 1) * h.index_select(0, row)
    out = torch.zeros(n, h.shape[1], dtype=h.dtype, device=h.device)
    out.scatter_add_(0, col.view(-1, 1).expand_as(msg), msg)
    return out + bias


def to_dense_batch(x: torch.Tensor, batch: torch.Tensor, fill_value, num_graphs: int):
    """PyG ``to_dense_batch``: [N,D] -> [B,Nmax,D] padded with ``fill_value``."""
    n_per = torch.bincount(batch, minlength=num_graphs)
    nmax = int(n_per.max()) if n_per.numel() else 0
    ptr = torch.zeros(num_graphs + 1, dtype=torch.int64, device=x.device)
    ptr[1:] = torch.cumsum(n_per, 0)
    pos = torch.arange(x.shape[0], device=x.device) - ptr[batch]
    dense = x.new_full((num_graphs * nmax, x.shape[1]), fill_value)
    dense[batch * nmax + pos] = x
    return dense.view(num_graphs, nmax, x.shape[1])


def sort_pool(x: torch.Tensor, batch: torch.Tensor, k: int = K_SORT,
              num_graphs: Optional[int] = None, stable: bool = False) -> torch.Tensor:
    """PyG ``SortAggregation(k).forward`` (model.py:35).

    ``stable=False`` is what the reference runs (torch default => tie order undefined).
    ``stable=True`` gives the documented tie-break of this build (lower node index
    first among equal keys), used by the tie tests.
    """
    if num_graphs is None:
        num_graphs = int(batch.max()) + 1
    fill_value = x.detach().min() - 1
    dense = to_dense_batch(x, batch, fill_value, num_graphs)
    B, N, D = dense.shape
    _, perm = dense[:, :, -1].sort(dim=-1, descending=True, stable=stable)
    arange = torch.arange(B, dtype=torch.long, device=perm.device) * N
    perm = perm + arange.view(-1, 1)
    dense = dense.view(B * N, D)[perm].view(B, N, D)
    if N >= k:
        dense = dense[:, :k].contiguous()
    else:
        pad = dense.new_full((B, k - N, D), fill_value)
        dense = torch.cat([dense, pad], dim=1)
    dense[dense == fill_value] = 0
    return dense.view(B, k * D)


class RefGCNConv(nn.Module):
    """Parameter container with PyG's key names (``lin.weight``, ``bias``) and init
    (glorot-uniform weight, zero bias)."""

    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.lin = nn.Linear(fin, fout, bias=False)
        self.bias = nn.Parameter(torch.zeros(fout))
        a = math.sqrt(6.0 / (fin + fout))
        with torch.no_grad():
            self.lin.weight.uniform_(-a, a)

    def forward(self, x, edge_index):
        return gcn_conv(x, edge_index, self.lin.weight, self.bias)


class RefModel(nn.Module):
    """Same attribute names / state_dict keys / forward order as model.py:9-45."""

    def __init__(self, num_features: int, num_classes: int):
        super().__init__()
        self.conv1 = RefGCNConv(num_features, HID)
        self.conv2 = RefGCNConv(HID, HID)
        self.conv3 = RefGCNConv(HID, HID)
        self.conv4 = RefGCNConv(HID, 1)
        self.conv5 = nn.Conv1d(1, 16, CAT, CAT)
        self.conv6 = nn.Conv1d(16, 32, 5, 1)
        self.pool = nn.MaxPool1d(2, 2)
        self.classifier_1 = nn.Linear(352, 128)
        self.drop_out = nn.Dropout(0.5)
        self.classifier_2 = nn.Linear(128, num_classes)
        self.stable_sort = False

    def graph_features(self, data):
        """[N,97] concat of the four tanh(GCN) outputs (model.py:28-34)."""
        x, edge_index = data.x, data.edge_index
        edge_index = remove_self_loops(edge_index)
        x_1 = torch.tanh(self.conv1(x, edge_index))
        x_2 = torch.tanh(self.conv2(x_1, edge_index))
        x_3 = torch.tanh(self.conv3(x_2, edge_index))
        x_4 = torch.tanh(self.conv4(x_3, edge_index))
        return torch.cat([x_1, x_2, x_3, x_4], dim=-1)

    def tail(self, pooled, dropout_mask=None):
        """model.py:36-43.  ``dropout_mask`` ([B,128] of 0/1) replaces torch's RNG so a
        run can be compared with a kernel that drew its own mask."""
        x = pooled.view(pooled.size(0), 1, pooled.size(-1))
        x = F.relu(self.conv5(x))
        x = self.pool(x)
        x = F.relu(self.conv6(x))
        x = x.view(x.size(0), -1)
        out = F.relu(self.classifier_1(x))
        if dropout_mask is not None:
            out = out * dropout_mask.to(out.dtype) * 2.0
        else:
            out = self.drop_out(out)
        return F.log_softmax(self.classifier_2(out), dim=-1)

    def forward(self, data, dropout_mask=None):
        num_graphs = getattr(data, "num_graphs", None)
        xcat = self.graph_features(data)
        pooled = sort_pool(xcat, data.batch, K_SORT, num_graphs, stable=self.stable_sort)
        return self.tail(pooled, dropout_mask)


def nll_mean(logp: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``nn.NLLLoss()`` default reduction='mean' (train.py:98)."""
    return F.nll_loss(logp, y)


def train_step(model: nn.Module, optimizer: torch.optim.Optimizer, data, y,
               dropout_mask=None):
    """One iteration of the reference loop body, train.py:37-45.
    Returns (loss float, number correct)."""
    pred = model(data, dropout_mask) if dropout_mask is not None else model(data)
    loss = nll_mean(pred, y)
    loss.backward()
    optimizer.step()
    optimizer.zero_grad()
    return float(loss.item()), int((pred.argmax(dim=1) == y).sum().item())


def eval_step(model: nn.Module, data, y):
    """Body of the reference ``test()`` loop, train.py:57-64."""
    with torch.no_grad():
        pred = model(data)
        loss = nll_mean(pred, y)
    return float(loss.item()), int((pred.argmax(dim=1) == y).sum().item())


def count_parameters(num_features: int, num_classes: int) -> int:
    m = RefModel(num_features, num_classes)
    return sum(p.numel() for p in m.parameters())
