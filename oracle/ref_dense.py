"""ORACLE (test infrastructure, NOT product code) -- independent fp64 dense-matrix
formulation of the reference hot path.

PARITY UNPINNED (see ``ref_ops.py`` header: PyG absent, no reference tests/goldens).
This second formulation shares NO code with the edge-list restatement in
``ref_ops.py``: every graph is turned into a dense adjacency matrix and the layer is
evaluated as the textbook symmetric GCN

    A~ = A + I            (A[t, s] = number of non-self-loop edges s -> t)
    D~ = diag(rowsum(A~)) (in-degree + 1)
    H' = tanh( D~^-1/2 A~ D~^-1/2 (H W^T) + b )

which is what PyG ``GCNConv`` (as called at /root/reference/model.py:13-16,30-33)
computes.  SortPooling (model.py:17,35) is done per graph with a STABLE descending
sort on the last channel (ties -> lower node index first; the reference's own tie
order is undefined), top-k rows, zero padding, node-major flattening.  The tail
(model.py:36-43) is written with explicit matmuls/windows instead of Conv1d/MaxPool1d.

Everything is float64 and differentiable through torch autograd, so it also yields
reference gradients.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

K_SORT = 30


def _p(sd: Dict[str, torch.Tensor], key: str) -> torch.Tensor:
    return sd[key].detach().to(torch.float64).clone().requires_grad_(True)


def load_params(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """fp64 leaf copies of a reference-keyed state_dict (``convN.lin.weight`` ...)."""
    keys = []
    for i in (1, 2, 3, 4):
        keys += [f"conv{i}.lin.weight", f"conv{i}.bias"]
    keys += ["conv5.weight", "conv5.bias", "conv6.weight", "conv6.bias",
             "classifier_1.weight", "classifier_1.bias",
             "classifier_2.weight", "classifier_2.bias"]
    return {k: _p(state_dict, k) for k in keys}


def dense_norm_adj(edge_index: torch.Tensor, n0: int, n1: int) -> torch.Tensor:
    """D~^-1/2 (A+I) D~^-1/2 for the graph owning nodes [n0, n1), fp64 dense."""
    n = n1 - n0
    A = torch.zeros(n, n, dtype=torch.float64)
    src, dst = edge_index[0], edge_index[1]
    m = (dst >= n0) & (dst < n1) & (src != dst)
    s = (src[m] - n0).to(torch.int64)
    t = (dst[m] - n0).to(torch.int64)
    if ((s < 0) | (s >= n)).any():
        raise ValueError("edge crosses a graph boundary: batch is not block-diagonal")
    A.index_put_((t, s), torch.ones(s.shape[0], dtype=torch.float64), accumulate=True)
    At = A + torch.eye(n, dtype=torch.float64)
    deg = At.sum(dim=1)
    dis = deg.pow(-0.5)
    return dis.view(-1, 1) * At * dis.view(1, -1)


def graph_features_dense(params, x, edge_index, batch, num_graphs):
    """[N,97] fp64: concat of the four tanh(GCN) layers, graph by graph."""
    x = x.to(torch.float64)
    n_per = torch.bincount(batch, minlength=num_graphs)
    ptr = torch.zeros(num_graphs + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(n_per, 0)
    outs = []
    for g in range(num_graphs):
        n0, n1 = int(ptr[g]), int(ptr[g + 1])
        if n1 == n0:
            continue
        Ah = dense_norm_adj(edge_index, n0, n1)
        h = x[n0:n1]
        layers = []
        for i in (1, 2, 3, 4):
            W = params[f"conv{i}.lin.weight"]
            b = params[f"conv{i}.bias"]
            h = torch.tanh(Ah @ (h @ W.t()) + b)
            layers.append(h)
        outs.append(torch.cat(layers, dim=1))
    return torch.cat(outs, dim=0), ptr


def sort_pool_dense(xcat, ptr, num_graphs, k=K_SORT, perm_override=None):
    """[B, k*97] fp64; stable descending sort on the last channel; zero padded.
    Also returns perm [B,k] (global node index or -1).

    ``perm_override`` ([B,k] global node ids, -1 = padding) replaces the oracle's own
    ordering: used by the tie-aware parity tests, which first check that the kernel's
    permutation is *a* valid descending top-k of the oracle's keys (up to a tolerance)
    and then push that same permutation through the oracle's tail."""
    D = xcat.shape[1]
    rows, perms = [], []
    for g in range(num_graphs):
        n0, n1 = int(ptr[g]), int(ptr[g + 1])
        n = n1 - n0
        key = xcat[n0:n1, -1].detach()
        if perm_override is not None:
            po = perm_override[g].to(torch.int64)
            order = po[po >= 0] - n0
        else:
            order = torch.sort(key, descending=True, stable=True).indices if n else key.new_zeros(0, dtype=torch.int64)
        m = min(n, k)
        sel = xcat[n0:n1][order[:m]]
        if m < k:
            sel = torch.cat([sel, xcat.new_zeros(k - m, D)], dim=0)
        rows.append(sel.reshape(1, k * D))
        p = torch.full((k,), -1, dtype=torch.int64)
        p[:m] = order[:m] + n0
        perms.append(p)
    return torch.cat(rows, 0), torch.stack(perms, 0)


def tail_dense(params, pooled, dropout_mask=None, k=K_SORT):
    """model.py:36-43 with explicit algebra.  pooled [B, k*97] -> log-probs [B,C]."""
    B = pooled.shape[0]
    W5 = params["conv5.weight"].reshape(16, 97)          # Conv1d(1,16,97,97): per-slot linear
    z5 = pooled.reshape(B, k, 97) @ W5.t() + params["conv5.bias"]      # [B,k,16]
    a5 = torch.relu(z5).transpose(1, 2)                                # [B,16,k]
    pooled2 = torch.maximum(a5[:, :, 0:k - (k % 2):2], a5[:, :, 1:k:2])  # MaxPool1d(2,2) -> [B,16,k//2]
    T = pooled2.shape[2]
    W6 = params["conv6.weight"]                                         # [32,16,5]
    Tout = T - 5 + 1
    wins = torch.stack([pooled2[:, :, d:d + Tout] for d in range(5)], dim=3)   # [B,16,Tout,5]
    z6 = torch.einsum("bctd,ocd->bot", wins, W6) + params["conv6.bias"].view(1, -1, 1)
    a6 = torch.relu(z6)                                                 # [B,32,11]
    flat = a6.reshape(B, -1)                                            # channel-major (x.view(B,-1))
    a1 = torch.relu(flat @ params["classifier_1.weight"].t() + params["classifier_1.bias"])
    if dropout_mask is not None:
        a1 = a1 * dropout_mask.to(torch.float64) * 2.0
    logits = a1 @ params["classifier_2.weight"].t() + params["classifier_2.bias"]
    return logits - torch.logsumexp(logits, dim=1, keepdim=True)


def forward_dense(state_dict, x, edge_index, batch, num_graphs: Optional[int] = None,
                  dropout_mask=None, return_all: bool = False, perm_override=None):
    """Full forward in fp64.  ``dropout_mask=None`` means eval mode (no dropout)."""
    x, edge_index, batch = x.cpu(), edge_index.cpu(), batch.cpu()
    if num_graphs is None:
        num_graphs = int(batch.max()) + 1
    params = load_params(state_dict)
    xcat, ptr = graph_features_dense(params, x, edge_index, batch, num_graphs)
    pooled, perm = sort_pool_dense(xcat, ptr, num_graphs, perm_override=perm_override)
    logp = tail_dense(params, pooled, dropout_mask)
    if return_all:
        return logp, dict(params=params, xcat=xcat, pooled=pooled, perm=perm, ptr=ptr)
    return logp


def loss_and_grads_dense(state_dict, x, edge_index, batch, y, num_graphs=None, dropout_mask=None,
                         perm_override=None):
    """Mean NLL (train.py:98) and d(loss)/d(param) in fp64."""
    logp, aux = forward_dense(state_dict, x, edge_index, batch, num_graphs, dropout_mask, True,
                              perm_override)
    y = y.cpu()
    loss = -logp[torch.arange(logp.shape[0]), y].mean()
    names = list(aux["params"].keys())
    grads = torch.autograd.grad(loss, [aux["params"][n] for n in names], allow_unused=True)
    g = {n: (torch.zeros_like(aux["params"][n]) if gi is None else gi) for n, gi in zip(names, grads)}
    return logp.detach(), loss.detach(), g, aux


def sort_margin(xcat: torch.Tensor, ptr: torch.Tensor, k: int = K_SORT,
                equiv_tol: float = 1e-9) -> float:
    """Smallest gap between sort keys that decides membership/order of the top-k of any
    graph, ignoring pairs of nodes whose whole 97-channel rows coincide (automorphic
    nodes: swapping them does not change the output).  Tests use it to assert a batch
    is 'tie-free' at the tolerance they compare with (SURVEY.md semantics trap #2:
    near ties between non-equivalent nodes legitimately flip whole rows)."""
    best = float("inf")
    xc = xcat.detach()
    for g in range(ptr.numel() - 1):
        n0, n1 = int(ptr[g]), int(ptr[g + 1])
        rows = xc[n0:n1]
        order = torch.sort(rows[:, -1], descending=True, stable=True).indices
        m = min(order.numel(), k + 1)
        for a in range(m - 1):
            ra, rb = rows[order[a]], rows[order[a + 1]]
            if float((ra - rb).abs().max()) <= equiv_tol:
                continue
            best = min(best, float((ra[-1] - rb[-1]).abs()))
    return best


def check_perm_valid(xcat: torch.Tensor, ptr: torch.Tensor, perm: torch.Tensor,
                     k: int = K_SORT, tol: float = 1e-5):
    """Is ``perm`` ([B,k], -1 padded) a valid SortPooling selection of ``xcat`` up to
    ``tol`` on the keys?  Returns (ok, message).  Valid means, per graph: exactly
    min(n,k) distinct in-graph nodes, keys non-increasing within ``tol``, and no
    excluded node has a key more than ``tol`` above the smallest selected key."""
    xc = xcat.detach()
    for g in range(ptr.numel() - 1):
        n0, n1 = int(ptr[g]), int(ptr[g + 1])
        n = n1 - n0
        m = min(n, k)
        p = perm[g].to(torch.int64)
        sel = p[:m]
        if (p[m:] != -1).any():
            return False, f"graph {g}: padding slots not -1"
        if m == 0:
            continue
        if (sel < n0).any() or (sel >= n1).any():
            return False, f"graph {g}: node outside graph"
        if torch.unique(sel).numel() != m:
            return False, f"graph {g}: duplicate node"
        keys = xc[sel, -1]
        if m > 1 and float((keys[1:] - keys[:-1]).max()) > tol:
            return False, f"graph {g}: keys not descending within tol"
        mask = torch.ones(n, dtype=torch.bool)
        mask[sel - n0] = False
        if mask.any():
            rest = xc[n0:n1, -1][mask]
            if float(rest.max() - keys.min()) > tol:
                return False, f"graph {g}: an excluded key beats a selected one by > tol"
    return True, "ok"
