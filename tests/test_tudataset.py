"""Host-side data pipeline: TU text format reader, Indegree layout, fold files, loader."""
import os

import numpy as np
import pytest
import torch

from dgcnn_amd import synth
from dgcnn_amd.batch import Graph
from dgcnn_amd.tudataset import (GraphLoader, make_fold_indices, read_fold_indices, read_tu_dataset,
                                 write_tu_dataset)


def _toy(tmp_path):
    # graph 0: path 0-1-2 (both directions) + a self loop + a duplicate entry ; graph 1: single directed edge
    g0 = Graph(torch.zeros(3, 1), torch.tensor([[0, 1, 1, 2, 0, 0], [1, 0, 2, 1, 0, 1]]), y=1)
    g1 = Graph(torch.zeros(2, 1), torch.tensor([[0], [1]]), y=0)
    labels = [np.array([5, 7, 5]), np.array([9, 7])]             # raw label values 5,7,9 -> 0,2,4 after shift: one-hot width 5
    attrs = [np.array([[0.5], [1.5], [2.5]]), np.array([[3.5], [4.5]])]
    write_tu_dataset(str(tmp_path), "TOY", [g0, g1], node_labels=labels, node_attrs=attrs, class_values=[-1, 1])
    return str(tmp_path)


def test_read_tu_format_layout_and_cleaning(tmp_path):
    root = _toy(tmp_path)
    ds = read_tu_dataset(os.path.join(root, "TOY"), "TOY")
    assert len(ds) == 2 and ds.num_classes == 2
    a, b = ds[0], ds[1]
    # self loop dropped, duplicate removed, sorted by (src,dst)
    assert a.edge_index.tolist() == [[0, 1, 1, 2], [1, 0, 2, 1]] and a.coalesced_undirected
    assert b.edge_index.tolist() == [[0], [1]] and not b.coalesced_undirected
    # features = [attr | one-hot(label - min) | indegree/max], degree LAST (utils.py:18-33)
    assert ds.num_features == 1 + 5 + 1
    assert a.x[:, 0].tolist() == [0.5, 1.5, 2.5]
    assert a.x[:, 1:6].argmax(1).tolist() == [0, 2, 0] and a.x[:, 1:6].sum(1).tolist() == [1, 1, 1]
    assert torch.allclose(a.x[:, -1], torch.tensor([1.0, 2.0, 1.0]) / 2.0)
    assert torch.allclose(b.x[:, -1], torch.tensor([0.0, 1.0]))
    # graph labels -1/1 -> 0/1 in sorted order; graph 0 had class_values[1] = 1 -> 1
    assert (a.y, b.y) == (1, 0)
    sub = ds[torch.tensor([1])]
    assert len(sub) == 1 and sub.graphs[0] is b


def test_label_less_dataset_gets_degree_column_only(tmp_path):
    graphs = synth.make_graphs("IMDB", 5, start=3)
    write_tu_dataset(str(tmp_path), "IM", graphs)
    ds = read_tu_dataset(os.path.join(str(tmp_path), "IM"), "IM")
    assert ds.num_features == 1
    for g, h in zip(graphs, ds.graphs):
        assert torch.equal(g.edge_index, h.edge_index) and torch.allclose(g.x, h.x) and h.coalesced_undirected


def test_round_trip_of_synthetic_proteins_shape(tmp_path):
    graphs = synth.make_graphs("PROTEINS", 4, start=11)
    labels = [g.x[:, 1:4].argmax(1).numpy() for g in graphs]
    attrs = [g.x[:, :1].numpy() for g in graphs]
    write_tu_dataset(str(tmp_path), "PR", graphs, node_labels=labels, node_attrs=attrs)
    ds = read_tu_dataset(os.path.join(str(tmp_path), "PR"), "PR")
    for g, h in zip(graphs, ds.graphs):
        assert torch.equal(g.edge_index, h.edge_index)
        # one-hot width is the number of DISTINCT-range labels present in the file (PyG semantics)
        assert torch.allclose(g.x[:, 0], h.x[:, 0], atol=1e-6) and torch.allclose(g.x[:, -1], h.x[:, -1])


def test_missing_files_raise_helpfully(tmp_path):
    with pytest.raises(FileNotFoundError):
        read_tu_dataset(str(tmp_path), "NOPE")


def test_fold_index_files_and_synthetic_folds(tmp_path):
    d = tmp_path / "DS" / "10fold_idx"
    d.mkdir(parents=True)
    (d / "train_idx-3.txt").write_text("0\n2\n4\n")
    (d / "test_idx-3.txt").write_text("1\n3\n")
    tr, te = read_fold_indices(str(tmp_path / "DS"), 3)
    assert tr.tolist() == [0, 2, 4] and te.tolist() == [1, 3] and tr.dtype == torch.long
    seen = []
    for f in range(1, 6):
        tr, te = make_fold_indices(23, f, folds=5, seed=7)
        assert len(set(tr.tolist()) & set(te.tolist())) == 0 and len(tr) + len(te) == 23
        seen += te.tolist()
    assert sorted(seen) == list(range(23))          # the test folds partition the dataset


def test_reference_fold_files_are_partitions_if_present():
    ref = "/root/reference/data/MUTAG"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not mounted (GPU box)")
    tr, te = read_fold_indices(ref, 1)
    assert len(set(tr.tolist()) & set(te.tolist())) == 0 and len(tr) + len(te) == 188     # README.md:62-72


def test_loader_batches_shuffle_and_short_last_batch():
    graphs = synth.make_graphs("MUTAG", 23)
    ld = GraphLoader(graphs, 10)
    sizes = [b.num_graphs for b in ld]
    assert sizes == [10, 10, 3] and len(ld) == 3 and ld.num_samples == 23
    g1 = torch.Generator().manual_seed(5); g2 = torch.Generator().manual_seed(5)
    a = [b.y.tolist() for b in GraphLoader(graphs, 10, shuffle=True, generator=g1)]
    c = [b.y.tolist() for b in GraphLoader(graphs, 10, shuffle=True, generator=g2)]
    assert a == c
    first = next(iter(GraphLoader(graphs, 10)))
    assert first.coalesced_undirected and first.max_nodes > 0 and first.max_edges > 0


@pytest.mark.gpu
def test_cli_driver_two_folds_on_synthetic(tmp_path):
    from dgcnn_amd import cli
    opt = cli.get_args(["--data_type", "MUTAG", "--synthetic", "60", "--num_epochs", "3", "--folds", "2",
                        "--batch_size", "16", "--out_dir", str(tmp_path)])
    over = cli.run(opt)
    assert len(over["test_accuracy"]) == 2 and all(0.0 <= a <= 100.0 for a in over["test_accuracy"])
    assert os.path.exists(tmp_path / "epochs" / "MUTAG_1.pth")
    sd = torch.load(tmp_path / "epochs" / "MUTAG_2.pth")
    assert "conv1.lin.weight" in sd and sd["conv1.lin.weight"].shape == (32, 8)
    rows = (tmp_path / "statistics" / "MUTAG_results_1.csv").read_text().strip().splitlines()
    assert rows[0] == "epoch,train_loss,test_loss,train_accuracy,test_accuracy" and len(rows) == 4
    assert (tmp_path / "statistics" / "MUTAG_results_overall.csv").exists()


@pytest.mark.gpu
def test_device_collate_is_bit_identical_to_host_collate_and_trains():
    """dgcnn_collate (one launch from a device-resident dataset) == collate(...).to(device), for shuffled ids, a short
    last batch, a feature width > 1, and through the loader's buffer ring with look-ahead."""
    from dgcnn_amd.batch import collate
    from dgcnn_amd.device_data import DeviceDataset, DeviceLoader
    for name in ("PROTEINS", "COLLAB"):
        graphs = synth.make_graphs(name, 37, start=5)
        ds = DeviceDataset(graphs)
        rng = np.random.default_rng(3)
        for ids in (rng.permutation(37)[:10], np.array([36]), np.arange(37)):
            ref = collate([graphs[i] for i in ids])
            got = ds.assemble(ids)
            assert got.num_graphs == ref.num_graphs and got.coalesced_undirected == ref.coalesced_undirected
            assert (got.max_nodes, got.max_edges) == (ref.max_nodes, ref.max_edges)
            assert torch.equal(got.x.cpu(), ref.x) and torch.equal(got.edge_index.cpu(), ref.edge_index)
            assert torch.equal(got.batch.cpu(), ref.batch) and torch.equal(got.y.cpu(), ref.y)
    # loader: same batches as the host loader under the same shuffle; earlier batch still intact after the next one
    graphs = synth.make_graphs("MUTAG", 23, labels="structure")
    ds = DeviceDataset(graphs)
    g1 = torch.Generator().manual_seed(9); g2 = torch.Generator().manual_seed(9)
    host = list(GraphLoader(graphs, 10, shuffle=True, generator=g1))
    it = iter(DeviceLoader(ds, 10, shuffle=True, generator=g2))
    prev = next(it)
    for k, h in enumerate(host):
        nxt = next(it, None)
        torch.cuda.synchronize()
        assert torch.equal(prev.x.cpu(), h.x) and torch.equal(prev.edge_index.cpu(), h.edge_index), k
        assert torch.equal(prev.y.cpu(), h.y)
        prev = nxt
    assert prev is None


@pytest.mark.gpu
def test_device_loader_large_batches_with_short_scan_mode_remainder():
    """batch_size > 256 takes the staging-buffer upload route, a short last batch (<= 256) the in-kernel scan route;
    over more than `ring` batches the short batch lands on a ring slot whose earlier upload may still be queued.  The
    scan route must not touch that slot's pinned staging buffer: every batch equals the host collate."""
    from dgcnn_amd.batch import collate
    from dgcnn_amd.device_data import DeviceDataset, DeviceLoader
    graphs = synth.make_graphs("MUTAG", 4 * 512 + 100, start=0)
    ds = DeviceDataset(graphs)
    for rep in range(3):                      # several epochs back to back without a sync between batches
        got = []
        for b in DeviceLoader(ds, 512, shuffle=False, ring=4):
            got.append((b.x.clone(), b.edge_index.clone(), b.batch.clone(), b.y.clone(), b.num_graphs))
        torch.cuda.synchronize()
        assert [g[4] for g in got] == [512, 512, 512, 512, 100]
        for k, g in enumerate(got):
            ref = collate(graphs[512 * k: 512 * k + g[4]])
            assert torch.equal(g[0].cpu(), ref.x) and torch.equal(g[1].cpu(), ref.edge_index), (rep, k)
            assert torch.equal(g[2].cpu(), ref.batch) and torch.equal(g[3].cpu(), ref.y), (rep, k)
