"""Dense per-graph block form of the aggregation (gcn_dense.hip): the bit-packed adjacency and work-item map built by
graph preparation are compared BIT-EXACT with a numpy restatement; the kernels go through the same oracle parity
protocol as the CSR-gather kernels (tests/parity_util.py), must agree with them within fp32 summation-order noise, be
run-to-run reproducible and independent of batch composition."""
import numpy as np
import pytest
import torch

from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import Batch, collate
from parity_util import check_backward_parity, check_forward_parity, cpu_state_dict, gpu_xcat, make_model

pytestmark = pytest.mark.gpu


def _bitmap_reference(b):
    """numpy restatement of dg_prep.h's dense structures: (words [31*N] u32, item records [(n0, n, r0)...], shares [3073])"""
    N, B = b.num_nodes, b.num_graphs
    ptr = np.searchsorted(b.batch.numpy(), np.arange(B + 1))
    words = np.zeros(31 * N, dtype=np.uint32)
    src, dst = b.edge_index[0].numpy(), b.edge_index[1].numpy()
    gid = b.batch.numpy()
    def cls(n):
        k32 = (n + 31) // 32
        return 0 if k32 <= 1 else 1 if k32 <= 2 else 2 if k32 <= 4 else 3 if k32 <= 8 else 4
    recs, costs = [], []
    for g in range(B):
        n0, n1 = int(ptr[g]), int(ptr[g + 1])
        n = n1 - n0
        for r in range((n + 127) // 128):            # items of 128 rows; cost = 3 * (pipeline stages of 64 k-rows) + 1
            recs.append((n0, n, 128 * r)); costs.append(3 * ((n + 63) // 64) + 1)
    tot = max(sum(costs), 1)
    split = np.zeros(3073, dtype=np.int64)
    c0 = 0
    for w, ic in enumerate(costs):
        c1 = c0 + ic
        for k in range(c0 * 3072 // tot + 1, min(c1 * 3072 // tot, 3072) + 1):
            split[k] = w + 1
        c0 = c1
    def setbit(i, j):
        g = gid[i]; n0, n1 = int(ptr[g]), int(ptr[g + 1])
        S = 1 << cls(n1 - n0)
        words[N * (S - 1) + i * S + ((j - n0) >> 5)] |= np.uint32(1 << ((j - n0) & 31))
    for i in range(N):
        setbit(i, i)
    for s, d in zip(src, dst):
        setbit(int(s), int(d))
    return words, np.array(recs, dtype=np.int32).reshape(-1, 3), split


@pytest.mark.parametrize("name,bs", [("MUTAG", 9), ("PROTEINS", 7), ("COLLAB", 6), ("COLLAB_REAL", 5), ("IMDB", 11)])
def test_dense_structures_bit_exact(name, bs):
    sh = synth.SHAPES[name]
    b = synth.make_batch(name, bs, start=40)
    if b.max_nodes > 512:
        pytest.skip("graph above the dense bound")
    m = make_model(sh.num_features, sh.num_classes)
    m.agg_mode, m.use_chain = "dense", False
    m.eval()
    with torch.no_grad():
        m(b.to("cuda"))
    m.check_errors()
    words, recs, split = _bitmap_reference(b)
    got_w = m.last_workspace_view("adjbits").cpu().numpy().view(np.uint32)[:31 * b.num_nodes]
    tab = m.last_workspace_view("dmap").cpu().numpy()
    np.testing.assert_array_equal(tab[:3073], split)                       # equal-cost shares
    assert tab[3073] == len(recs)
    np.testing.assert_array_equal(tab[3096:3096 + 3 * len(recs)].reshape(-1, 3), recs)
    # only the words of each row's OWN stride class are defined content (the other classes are never written)
    ptr = np.searchsorted(b.batch.numpy(), np.arange(b.num_graphs + 1))
    N = b.num_nodes
    for g in range(b.num_graphs):
        n0, n1 = int(ptr[g]), int(ptr[g + 1])
        k32 = (n1 - n0 + 31) // 32
        S = 1 if k32 <= 1 else 2 if k32 <= 2 else 4 if k32 <= 4 else 8 if k32 <= 8 else 16
        lo, hi = N * (S - 1) + n0 * S, N * (S - 1) + n1 * S
        np.testing.assert_array_equal(got_w[lo:hi], words[lo:hi], err_msg=f"graph {g}")


WORKLOADS = [("MUTAG", 50), ("PROTEINS", 24), ("COLLAB", 50), ("COLLAB_REAL", 50), ("IMDB", 50), ("COLLAB", 256)]


@pytest.mark.parametrize("name,bs", WORKLOADS, ids=[f"{w[0]}-{w[1]}" for w in WORKLOADS])
def test_dense_forward_backward_vs_oracle_and_vs_gather(name, bs):
    sh = synth.SHAPES[name]
    start = 1000
    b = synth.make_batch(name, bs, start=start)
    while b.max_nodes > 512:
        start += bs
        b = synth.make_batch(name, bs, start=start)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    m.agg_mode, m.use_chain = "dense", False
    check_forward_parity(m, b, sd)
    xd = gpu_xcat(m)
    m.agg_mode = "sparse"
    check_forward_parity(m, b, sd)
    xs = gpu_xcat(m)
    assert float((xd - xs).abs().max()) <= 4e-6          # same sums, different order
    m.agg_mode, m.use_chain = "dense", False
    check_backward_parity(m, b, sd)


@pytest.mark.parametrize("F", [1, 2, 3, 7, 13, 16, 17, 32, 33, 40])
def test_dense_raw_feature_widths(F):
    base = synth.make_batch("COLLAB" if F % 2 else "PROTEINS", 12, start=77)
    g = torch.Generator().manual_seed(F)
    b = Batch(torch.randn(base.x.shape[0], F, generator=g), base.edge_index, base.batch, base.y, base.num_graphs,
              base.coalesced_undirected, base.max_nodes, base.max_edges)
    m = make_model(F, 3)
    m.agg_mode, m.use_chain = "dense", False
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


def test_dense_results_do_not_depend_on_batch_composition_and_are_reproducible():
    sh = synth.SHAPES["COLLAB"]
    graphs = synth.make_graphs("COLLAB", 96, start=500)
    m = make_model(sh.num_features, sh.num_classes)
    m.agg_mode, m.use_chain = "dense", False
    m.eval()
    with torch.no_grad():
        big = collate(graphs).to("cuda")
        lp1 = m(big).clone()
        x1 = gpu_xcat(m)
        lp2 = m(big).clone()
        assert torch.equal(lp1, lp2) and torch.equal(x1, gpu_xcat(m))
        parts = [m(collate(graphs[k:k + 32]).to("cuda")).clone() for k in range(0, 96, 32)]
        assert torch.equal(torch.cat(parts), lp1)
        rev = m(collate(graphs[::-1]).to("cuda")).clone()
        assert torch.equal(rev.flip(0), lp1)


# ---- fused dense forward: one workgroup per graph, conv1..conv4 + readout in one launch (small batches) ----------
FUSED_WORKLOADS = [("MUTAG", 50), ("PROTEINS", 20), ("COLLAB", 50), ("COLLAB_REAL", 30), ("IMDB", 50), ("COLLAB", 1), ("COLLAB", 128)]


@pytest.mark.parametrize("name,bs", FUSED_WORKLOADS, ids=[f"{w[0]}-{w[1]}" for w in FUSED_WORKLOADS])
def test_fused_dense_forward_vs_oracle_and_backward_through_it(name, bs):
    sh = synth.SHAPES[name]
    start = 1000
    b = synth.make_batch(name, bs, start=start)
    while b.max_nodes > 192:
        start += bs
        b = synth.make_batch(name, bs, start=start)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    m.use_fused, m.agg_mode, m.use_chain = True, "dense", False
    check_forward_parity(m, b, sd)
    xf = gpu_xcat(m)
    m.use_fused, m.agg_mode = False, "sparse"
    check_forward_parity(m, b, sd)
    assert float((xf - gpu_xcat(m)).abs().max()) <= 4e-6
    m.use_fused, m.agg_mode, m.use_chain = True, "dense", False
    check_backward_parity(m, b, sd)          # backward (tiled kernels) from the activations the fused forward saved


@pytest.mark.parametrize("F", [1, 2, 3, 7, 13, 16, 17, 32])
def test_fused_dense_raw_feature_widths(F):
    base = synth.make_batch("COLLAB" if F % 2 else "PROTEINS", 12, start=77)
    g = torch.Generator().manual_seed(F)
    b = Batch(torch.randn(base.x.shape[0], F, generator=g), base.edge_index, base.batch, base.y, base.num_graphs,
              base.coalesced_undirected, base.max_nodes, base.max_edges)
    m = make_model(F, 3)
    m.use_fused, m.agg_mode, m.use_chain = True, "dense", False
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


def test_fused_dense_reproducible_composition_independent_and_pipelined():
    """the one-launch forward (opt-in: FORCE_FUSED + AGG_DENSE): results are bit-identical run to run, for any batch
    composition and graph order, and the trainer's pipelined step (graph prep of the next batch riding on the fused
    launch) reproduces the unpipelined one bit for bit"""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    graphs = synth.make_graphs("COLLAB", 96, start=700)
    m = make_model(sh.num_features, sh.num_classes)
    m.eval()
    with torch.no_grad():
        full = collate(graphs).to("cuda")
        m.use_fused, m.agg_mode, m.use_chain = True, "dense", False
        lp = m(full).clone()
        xa = gpu_xcat(m)
        assert torch.equal(m(full), lp) and torch.equal(gpu_xcat(m), xa)
        parts = [m(collate(graphs[k:k + 24]).to("cuda")).clone() for k in range(0, 96, 24)]
        assert torch.equal(torch.cat(parts), lp)
        assert torch.equal(m(collate(graphs[::-1]).to("cuda")).flip(0), lp)
    m.check_errors()
    batches = [collate(graphs[k:k + 32]).to("cuda") for k in range(0, 96, 32)]
    res = []
    for look in (False, True):
        mm = make_model(sh.num_features, sh.num_classes)
        mm.train(); mm._seed_base, mm._fwd_count = 5, 0
        mm.use_fused, mm.agg_mode, mm.use_chain = True, "dense", False
        tr = Trainer(mm)
        for k in range(6):
            tr.train_step(batches[k % 3], batches[k % 3].y, next_data=batches[(k + 1) % 3] if look else None)
        torch.cuda.synchronize()
        tr.read_metrics()
        res.append(mm.flat_params.clone())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("route", ["forward", "pipelined"])
def test_dense_form_detects_a_missing_reverse_edge(route):
    """the dense form relies on A == A^T (one bitmap for both directions): a coalesced_undirected promise that does not
    hold must be flagged -- checked on the bitmap itself (one word per edge) after it has been built"""
    from dgcnn_amd import _lib
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    good = synth.make_batch("COLLAB", 12, start=900)
    ei = good.edge_index
    keep = torch.ones(ei.shape[1], dtype=torch.bool)
    keep[ei.shape[1] // 2] = False                         # drop ONE directed edge: still sorted, no longer symmetric
    bad = Batch(good.x, ei[:, keep].contiguous(), good.batch, good.y, good.num_graphs, True, good.max_nodes, good.max_edges)
    m = make_model(sh.num_features, sh.num_classes)
    m.agg_mode, m.use_chain = "dense", False
    if route == "forward":
        m.eval()
        with torch.no_grad():
            m(good.to("cuda")); m.check_errors()
            m(bad.to("cuda"))
        with pytest.raises(_lib.DgcnnError):
            m.check_errors()
    else:
        m.train()
        tr = Trainer(m)
        g, b = good.to("cuda"), bad.to("cuda")
        tr.train_step(g, g.y, next_data=b)                 # bad's structures (and the check) are built during this step
        tr.train_step(b, b.y, next_data=g)
        with pytest.raises(_lib.DgcnnError):
            tr.read_metrics()


# ---- edge cases of the dense forms: word / tile / stage / class boundaries, isolated nodes, single-node graphs ----------
def _sized_batch(sizes, F=3, seed=0, isolated=()):
    """graphs of exactly the given node counts (G(n,p), both directions, sorted by (src,dst)); graph indices in
    `isolated` get NO edges at all for their first node (and single-node graphs have none by construction)"""
    from dgcnn_amd.batch import Graph
    rng = np.random.default_rng(seed)
    gs = []
    for gi, n in enumerate(sizes):
        if n == 1:
            ei = torch.zeros(2, 0, dtype=torch.int64)
        else:
            p = min(1.0, 6.0 / max(n - 1, 1)) if n > 40 else 0.5
            m = np.triu(rng.random((n, n)) < p, 1)
            if gi in isolated:
                m[0, :] = False
            if not m.any():
                m[n - 2, n - 1] = True
            a, b = np.nonzero(m)
            src, dst = np.concatenate([a, b]), np.concatenate([b, a])
            o = np.lexsort((dst, src))
            ei = torch.from_numpy(np.stack([src[o], dst[o]]).astype(np.int64))
        x = torch.from_numpy(rng.standard_normal((n, F)).astype(np.float32))
        gs.append(Graph(x=x, edge_index=ei, y=int(rng.integers(0, 2)), coalesced_undirected=True))
    return collate(gs)


@pytest.mark.parametrize("sizes,isolated", [
    ([31, 32, 33, 63, 64, 65], ()), ([127, 128, 129, 16, 15, 17], ()), ([191, 192, 193, 1, 2, 3], (0, 2)),
    ([255, 256, 257], ()), ([511, 512, 5], (1,)), ([1, 1, 1, 40], ()), ([2], ())],
    ids=["words", "stage", "fused_limit", "class8", "max512", "single_nodes", "one_tiny_graph"])
def test_dense_boundary_sizes_isolated_nodes_and_single_node_graphs(sizes, isolated):
    b = _sized_batch(sizes, seed=sum(sizes), isolated=isolated)
    m = make_model(3, 2)
    sd = cpu_state_dict(m)
    m.agg_mode, m.use_chain = "dense", False
    check_forward_parity(m, b, sd)
    xd = gpu_xcat(m)
    check_backward_parity(m, b, sd)
    m.agg_mode = "sparse"
    check_forward_parity(m, b, sd)
    assert float((xd - gpu_xcat(m)).abs().max()) <= 4e-6
    if max(sizes) <= 192:
        m.agg_mode, m.use_fused, m.use_chain = "dense", True, False
        check_forward_parity(m, b, sd)
        assert float((xd - gpu_xcat(m)).abs().max()) <= 4e-6


def test_dense_is_not_taken_above_512_nodes_and_the_bf16_leg_says_so():
    from dgcnn_amd import _lib
    b = _sized_batch([513, 20], seed=7)
    m = make_model(3, 2)
    sd = cpu_state_dict(m)
    m.agg_mode, m.use_chain = "dense", False                    # asked for, not admissible: the gather form runs, results stay right
    check_forward_parity(m, b, sd)
    m.compute_dtype = "bf16"
    m.eval()
    with pytest.raises(_lib.DgcnnError):
        with torch.no_grad():
            m(b.to("cuda"))
