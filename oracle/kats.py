"""ORACLE (test infrastructure) -- hand-computed known-answer tests.

PARITY UNPINNED (see ``ref_ops.py``).  The reference ships no tests or golden vectors
(SURVEY.md §4), so these closed-form cases -- worked out by hand from the symmetric
GCN definition, not by running any implementation -- are what pins the oracle and the
HIP kernels on tiny graphs.  Each expected value is an
explicit expression of rationals and square roots.

Notation: in-degree+1 = d~ ;  out[t] = sum_{s->t} h[s] / sqrt(d~_s d~_t) + h[t] / d~_t + b,
with h = x W^T  (PyG GCNConv as called at /root/reference/model.py:13-16,30-33).
"""
from __future__ import annotations

from math import sqrt
from typing import List, NamedTuple

import numpy as np


class GcnKat(NamedTuple):
    name: str
    x: np.ndarray            # [n, F]
    edge_index: np.ndarray   # [2, E] (src, dst)
    weight: np.ndarray       # [Fout, F]
    bias: np.ndarray         # [Fout]
    expected: np.ndarray     # [n, Fout] pre-activation (before tanh)


def _ei(pairs):
    if not pairs:
        return np.zeros((2, 0), dtype=np.int64)
    return np.array(pairs, dtype=np.int64).T.copy()


def gcn_kats() -> List[GcnKat]:
    k: List[GcnKat] = []
    one = np.array([[1.0]])
    zero = np.array([0.0])
    r2, r3, r6 = sqrt(2.0), sqrt(3.0), sqrt(6.0)

    # 1. two-node path 0<->1: d~ = (2,2); A^ = [[1/2,1/2],[1/2,1/2]]
    k.append(GcnKat("path2", np.array([[1.0], [3.0]]), _ei([(0, 1), (1, 0)]), one, zero,
                    np.array([[2.0], [2.0]])))

    # 2. star, centre 0, leaves 1,2: d~ = (3,2,2)
    #    out0 = x0/3 + (x1+x2)/sqrt6 ; out1 = x1/2 + x0/sqrt6 ; out2 = x2/2 + x0/sqrt6
    k.append(GcnKat("star3", np.array([[1.0], [2.0], [3.0]]),
                    _ei([(0, 1), (0, 2), (1, 0), (2, 0)]), one, zero,
                    np.array([[1.0 / 3 + 5.0 / r6], [1.0 + 1.0 / r6], [1.5 + 1.0 / r6]])))

    # 3. triangle: d~ = (3,3,3); every entry of A^ is 1/3 -> out = sum/3
    k.append(GcnKat("triangle", np.array([[1.0], [2.0], [6.0]]),
                    _ei([(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)]), one, zero,
                    np.array([[3.0], [3.0], [3.0]])))

    # 4. isolated node next to a path: node 2 has d~ = 1 -> out2 = x2 W + b
    k.append(GcnKat("isolated", np.array([[1.0], [3.0], [7.0]]), _ei([(0, 1), (1, 0)]),
                    np.array([[2.0]]), np.array([0.5]),
                    np.array([[4.5], [4.5], [14.5]])))

    # 5. an input self loop (0,0) must be ignored (removed, then exactly one re-added)
    k.append(GcnKat("selfloop_ignored", np.array([[1.0], [3.0]]),
                    _ei([(0, 0), (0, 1), (1, 0)]), one, zero,
                    np.array([[2.0], [2.0]])))

    # 6. one directed edge 0->1: d~ = (1,2).  out0 = x0 ; out1 = x1/2 + x0/sqrt2
    k.append(GcnKat("directed", np.array([[4.0], [6.0]]), _ei([(0, 1)]), one, zero,
                    np.array([[4.0], [3.0 + 4.0 / r2]])))

    # 7. duplicated directed edge 0->1 twice: d~ = (1,3). out1 = x1/3 + 2 x0/sqrt3
    k.append(GcnKat("multiedge", np.array([[1.0], [3.0]]), _ei([(0, 1), (0, 1)]), one, zero,
                    np.array([[1.0], [1.0 + 2.0 / r3]])))

    # 8. two features -> two outputs, bias: path2 with W = [[1,0],[1,-1]], b = (0.25,-1)
    #    h = [[1,-1],[3,-1]] (x = [[1,2],[3,4]]);  out = (h0+h1)/2 + b for both nodes
    k.append(GcnKat("path2_F2", np.array([[1.0, 2.0], [3.0, 4.0]]), _ei([(0, 1), (1, 0)]),
                    np.array([[1.0, 0.0], [1.0, -1.0]]), np.array([0.25, -1.0]),
                    np.array([[2.25, -2.0], [2.25, -2.0]])))

    # 9. two disjoint graphs in one batch (block diagonal): path2 (nodes 0,1) + directed (2->3)
    k.append(GcnKat("two_graphs", np.array([[1.0], [3.0], [4.0], [6.0]]),
                    _ei([(0, 1), (1, 0), (2, 3)]), one, zero,
                    np.array([[2.0], [2.0], [4.0], [3.0 + 4.0 / r2]])))
    return k


class SortKat(NamedTuple):
    name: str
    x: np.ndarray        # [N, D]
    batch: np.ndarray    # [N]
    k: int
    expected: np.ndarray  # [B, k*D]
    perm: np.ndarray      # [B, k] global node ids, -1 = padding


def sortpool_kats() -> List[SortKat]:
    out: List[SortKat] = []
    # D = 2, k = 3.  Graph 0: 2 nodes (n<k); graph 1: 3 nodes (n=k); graph 2: 5 nodes (n>k)
    x = np.array([
        [10.0, 0.2], [11.0, 0.7],                                  # g0
        [20.0, -0.5], [21.0, 0.9], [22.0, 0.1],                    # g1
        [30.0, 0.3], [31.0, -0.9], [32.0, 0.8], [33.0, 0.0], [34.0, 0.5],   # g2
    ])
    batch = np.array([0, 0, 1, 1, 1, 2, 2, 2, 2, 2])
    exp = np.array([
        [11.0, 0.7, 10.0, 0.2, 0.0, 0.0],
        [21.0, 0.9, 22.0, 0.1, 20.0, -0.5],
        [32.0, 0.8, 34.0, 0.5, 30.0, 0.3],
    ])
    perm = np.array([[1, 0, -1], [3, 4, 2], [7, 9, 5]])
    out.append(SortKat("mixed_sizes", x, batch, 3, exp, perm))

    # every graph smaller than k (batch max < k): k = 4, sizes 1 and 2
    x2 = np.array([[1.0, 0.5], [2.0, -0.1], [3.0, 0.4]])
    b2 = np.array([0, 1, 1])
    exp2 = np.array([
        [1.0, 0.5, 0, 0, 0, 0, 0, 0],
        [3.0, 0.4, 2.0, -0.1, 0, 0, 0, 0],
    ])
    out.append(SortKat("all_small", x2, b2, 4, exp2, np.array([[0, -1, -1, -1], [2, 1, -1, -1]])))

    # exact ties: documented tie-break of this build = lower node index first
    x3 = np.array([[1.0, 0.5], [2.0, 0.5], [3.0, 0.9], [4.0, 0.5]])
    b3 = np.array([0, 0, 0, 0])
    exp3 = np.array([[3.0, 0.9, 1.0, 0.5, 2.0, 0.5]])
    out.append(SortKat("ties_stable", x3, b3, 3, exp3, np.array([[2, 0, 1]])))
    return out


# README parameter-count KAT: /root/reference/README.md:62-105 -> (F, C) -> #parameters
README_PARAM_COUNTS = {
    "MUTAG": (8, 2, 52035),
    "PTC": (19, 2, 52387),
    "NCI1": (38, 2, 52995),
    "PROTEINS": (5, 2, 51939),
    "DD": (90, 2, 54659),
    "COLLAB": (1, 3, 51940),
    "IMDB-B": (1, 2, 51811),
    "IMDB-M": (1, 3, 51940),
}
