"""Graph-chain kernels (gcn_chain.hip): conv1..conv4 of a graph inside one workgroup (forward) and the matching backward
chain.  Same oracle parity protocol as every other kernel family (tests/parity_util.py); agreement with the CSR-gather
family within fp32 summation-order noise; run-to-run reproducibility; independence of batch composition; both size
classes (<= 128 nodes, 129..512 nodes) and their boundaries."""
import numpy as np
import pytest
import torch

from dgcnn_amd import synth
from dgcnn_amd.batch import Batch, collate
from parity_util import check_backward_parity, check_forward_parity, cpu_state_dict, gpu_xcat, make_model
from test_gpu_dense import _sized_batch

pytestmark = pytest.mark.gpu

WORKLOADS = [("MUTAG", 50), ("PROTEINS", 24), ("COLLAB", 50), ("COLLAB_REAL", 50), ("IMDB", 50), ("COLLAB", 256)]


def _chain(m):
    m.agg_mode, m.use_chain, m.use_fused = "dense", True, None


@pytest.mark.parametrize("name,bs", WORKLOADS, ids=[f"{w[0]}-{w[1]}" for w in WORKLOADS])
def test_chain_forward_backward_vs_oracle_and_vs_gather(name, bs):
    sh = synth.SHAPES[name]
    start = 1000
    b = synth.make_batch(name, bs, start=start)
    while b.max_nodes > 512:
        start += bs
        b = synth.make_batch(name, bs, start=start)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    _chain(m)
    check_forward_parity(m, b, sd)
    xc = gpu_xcat(m)
    m.agg_mode, m.use_chain = "sparse", False
    check_forward_parity(m, b, sd)
    assert float((xc - gpu_xcat(m)).abs().max()) <= 4e-6          # same sums, different order
    _chain(m)
    check_backward_parity(m, b, sd)


@pytest.mark.parametrize("F", [1, 2, 3, 4, 5, 7, 13, 16, 17, 20, 31, 32])
def test_chain_raw_feature_widths(F):
    base = synth.make_batch("COLLAB" if F % 2 else "PROTEINS", 12, start=77)
    g = torch.Generator().manual_seed(F)
    b = Batch(torch.randn(base.x.shape[0], F, generator=g), base.edge_index, base.batch, base.y, base.num_graphs,
              base.coalesced_undirected, base.max_nodes, base.max_edges)
    m = make_model(F, 3)
    _chain(m)
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


@pytest.mark.parametrize("sizes,isolated", [
    ([31, 32, 33, 63, 64, 65], ()), ([127, 128, 129, 16, 15, 17], ()), ([191, 192, 193, 1, 2, 3], (0, 2)),
    ([255, 256, 257], ()), ([511, 512, 5], (1,)), ([1, 1, 1, 40], ()), ([2], ()), ([96, 97, 111, 112, 113], ())],
    ids=["words", "class_boundary", "mixed", "class8", "max512", "single_nodes", "one_tiny_graph", "tile_gap"])
def test_chain_boundary_sizes_isolated_nodes_and_single_node_graphs(sizes, isolated):
    b = _sized_batch(sizes, seed=sum(sizes), isolated=isolated)
    m = make_model(3, 2)
    sd = cpu_state_dict(m)
    _chain(m)
    check_forward_parity(m, b, sd)
    xc = gpu_xcat(m)
    check_backward_parity(m, b, sd)
    m.agg_mode, m.use_chain = "sparse", False
    check_forward_parity(m, b, sd)
    assert float((xc - gpu_xcat(m)).abs().max()) <= 4e-6


@pytest.mark.parametrize("name,bs", [("MUTAG", 50), ("PROTEINS", 24), ("COLLAB", 50), ("IMDB", 20)])
def test_chain_forward_with_gather_backward(name, bs):
    """small batches: the chain forward (bitmap + schedule built, reverse edges checked per edge in graph prep) feeding the
    CSR-gather backward kernels -- the library's own choice when neither aggregation form is forced"""
    sh = synth.SHAPES[name]
    start = 2000
    b = synth.make_batch(name, bs, start=start)
    while b.max_nodes > 512:
        start += bs
        b = synth.make_batch(name, bs, start=start)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    m.use_chain = True
    check_forward_parity(m, b, sd)
    xc = gpu_xcat(m)
    check_backward_parity(m, b, sd)
    m.use_chain = False
    check_forward_parity(m, b, sd)
    assert float((xc - gpu_xcat(m)).abs().max()) <= 4e-6


def test_chain_forward_flags_a_missing_reverse_edge_without_the_bitmap_check():
    from dgcnn_amd import _lib
    good = synth.make_batch("COLLAB", 12, start=900)
    ei = good.edge_index
    keep = torch.ones(ei.shape[1], dtype=torch.bool)
    keep[ei.shape[1] // 2] = False
    bad = Batch(good.x, ei[:, keep].contiguous(), good.batch, good.y, good.num_graphs, True, good.max_nodes, good.max_edges)
    m = make_model(1, 3)
    m.use_chain = True
    m.eval()
    with torch.no_grad():
        m(good.to("cuda")); m.check_errors()
        m(bad.to("cuda"))
    with pytest.raises(_lib.DgcnnError):
        m.check_errors()


def test_chain_schedule_is_the_stable_sort_by_tile_count():
    """graph preparation's schedule (dg_prep.h): {first node, node count} of every graph, ordered by 16-row tile count
    descending, ties by graph index; graphs above 128 nodes first, their number in the table header -- bit-exact"""
    b = _sized_batch([5, 130, 64, 17, 16, 300, 128, 129, 1, 33, 48, 47, 200, 96], seed=3)
    m = make_model(3, 2)
    _chain(m)
    m.eval()
    with torch.no_grad():
        m(b.to("cuda"))
    m.check_errors()
    N, B = b.num_nodes, b.num_graphs
    ptr = np.searchsorted(b.batch.numpy(), np.arange(B + 1))
    sizes = np.diff(ptr)
    order = sorted(range(B), key=lambda g: (-((sizes[g] + 15) // 16), g))
    tab = m.last_workspace_view("dmap").cpu().numpy()
    s0 = (3096 + 3 * (N // 128 + B + 1) + 1) & ~1
    got = tab[s0:s0 + 2 * B].reshape(B, 2)
    want = np.array([[ptr[g], sizes[g]] for g in order])
    np.testing.assert_array_equal(got, want)
    assert tab[3074] == int((sizes > 128).sum())


def test_chain_without_max_nodes_hint_runs_both_size_classes():
    b = _sized_batch([130, 20, 300, 64], seed=11)
    b = Batch(b.x, b.edge_index, b.batch, b.y, b.num_graphs, True, 512, b.max_edges)      # loose (but valid) bound
    m = make_model(3, 2)
    sd = cpu_state_dict(m)
    _chain(m)
    check_forward_parity(m, b, sd)


def test_chain_results_do_not_depend_on_batch_composition_and_are_reproducible():
    sh = synth.SHAPES["COLLAB"]
    graphs = synth.make_graphs("COLLAB", 96, start=500)
    m = make_model(sh.num_features, sh.num_classes)
    _chain(m)
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


def test_chain_training_steps_match_the_per_layer_family():
    """a few fused training steps (Trainer, pipelined graph prep) through the chain kernels stay within rounding of the
    same steps through the CSR-gather kernels"""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    graphs = synth.make_graphs("COLLAB", 96, start=300)
    batches = [collate(graphs[k:k + 32]).to("cuda") for k in range(0, 96, 32)]
    res = []
    for chain in (True, False):
        mm = make_model(sh.num_features, sh.num_classes)
        mm.train(); mm._seed_base, mm._fwd_count = 5, 0
        if chain:
            _chain(mm)
        else:
            mm.agg_mode, mm.use_chain = "sparse", False
        tr = Trainer(mm)
        for k in range(6):
            tr.train_step(batches[k % 3], batches[k % 3].y, next_data=batches[(k + 1) % 3])
        torch.cuda.synchronize()
        loss, _ = tr.read_metrics()
        res.append((mm.flat_params.clone(), loss))
    assert abs(res[0][1] - res[1][1]) <= 1e-4
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-4      # six Adam steps of lr 1e-3 from within-rounding gradients


@pytest.mark.parametrize("bs,hint", [(300, 256), (300, 200), (700, 256), (50, 100), (1100, 128)])
def test_chain_path_flags_a_max_nodes_hint_that_is_too_small(bs, hint):
    """on the chain path the kernels are CHOSEN from the host's max_nodes hint (size-class launch skipped
    at <= 256, the 256-node chain backward taken), so a graph above the hint would be left out of them and its rows stay stale.
    Graph preparation now checks the hint on the device: such a batch is flagged (check_errors raises), through the eval
    forward, the drop-in training forward and the pipelined Trainer step (rider / side-stream preparation)."""
    from dgcnn_amd import _lib
    from dgcnn_amd.batch import Batch
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    b = synth.make_batch("COLLAB", bs, start=4000, force_first_n=hint + 7)      # one graph just above the hint
    assert b.max_nodes >= hint + 7
    lie = Batch(b.x, b.edge_index, b.batch, b.y, b.num_graphs, True, hint, b.max_edges)
    m = make_model(sh.num_features, sh.num_classes)
    m.use_chain = True
    m.eval()
    with torch.no_grad():
        m(lie.to("cuda"))
    with pytest.raises(_lib.DgcnnError):
        m.check_errors()
    # the honest hint on the same graphs is clean
    with torch.no_grad():
        m(b.to("cuda"))
    m.check_errors()
    # pipelined training: the lie is the NEXT batch of a clean step (its preparation rides / runs on the side stream)
    m.train()
    tr = Trainer(m)
    good, bad = b.to("cuda"), lie.to("cuda")
    tr.train_step(good, good.y, next_data=bad)
    tr.train_step(bad, bad.y, next_data=good)
    torch.cuda.synchronize()
    with pytest.raises(_lib.DgcnnError):
        tr.read_metrics()
