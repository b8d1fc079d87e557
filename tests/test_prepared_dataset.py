"""Prepared dataset (SURVEY.md §8(f) N3 as written): per-graph CSR / dinv / pre-scaled features / bitmap rows built ONCE per
dataset (dgcnn_dataset_prepare), a batch = a copy with offset adds (dgcnn_assemble, csrc/dg_assemble.h) -- replacing the
reference's per-batch host collate (/root/reference/train.py:108-109) and the per-batch graph preparation (remove_self_loops +
the 4 gcn_norm calls of /root/reference/model.py:28-33).  Pure data movement + the same preparation kernels: everything here
is compared BIT FOR BIT with the per-batch path (which the oracle tests cover)."""
import numpy as np
import pytest
import torch

from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import collate
from parity_util import cpu_state_dict, make_model

pytestmark = pytest.mark.gpu


def _ws_regions(m, names):
    return {n: m.last_workspace_view(n).clone() for n in names}


CASES = [("COLLAB", 120, 50, None), ("MUTAG", 90, 50, None), ("PROTEINS", 80, 50, None), ("IMDB", 70, 33, None),
         ("DD", 40, 12, None), ("DD", 30, 8, 700), ("COLLAB", 700, 600, None), ("COLLAB_REAL", 60, 20, None)]


@pytest.mark.parametrize("name,G,B,force", CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}-{c[3]}" for c in CASES])
def test_assembled_batch_equals_per_batch_preparation_bit_for_bit(name, G, B, force):
    """eval forward through Model.forward: structures (graph_ptr, dinv, and the CSR where the batch's form reads one), every
    activation, the SortPooling permutation and the log-probabilities; then the drop-in backward's 16 gradients"""
    from dgcnn_amd.device_data import PreparedDataset
    sh = synth.SHAPES[name]
    graphs = synth.make_graphs(name, G, start=300)
    if force:        # a graph above the 512-node bound of the bitmap forms: no bitmap rows for it, its batch takes the CSR kernels
        graphs[3] = synth.make_graphs(name, 1, start=9000, force_first_n=force)[0]
    ds = PreparedDataset(graphs, bitmap=True)
    rng = np.random.default_rng(5)
    ids = rng.permutation(G)[:B]
    if force:
        ids[0] = 3
    ref_b = collate([graphs[i] for i in ids])
    pb = ds.batch_of(ids)
    assert (pb.num_nodes, pb.num_edges, pb.max_nodes, pb.max_edges) == (ref_b.num_nodes, ref_b.num_edges, ref_b.max_nodes, ref_b.max_edges)
    m = make_model(sh.num_features, sh.num_classes)
    names = ["graph_ptr", "dinv", "x1", "x2", "x3", "x4", "perm", "pooled", "ax"] if sh.num_features <= 32 else \
            ["graph_ptr", "dinv", "x1", "x2", "x3", "x4", "perm", "pooled"]
    m.eval()
    with torch.no_grad():
        lp_ref = m(ref_b.to("cuda")).clone()
    m.check_errors()
    r_ref = _ws_regions(m, names)
    fl = m._mode_flags() | _lib.FLAG_COALESCED_UNDIRECTED
    form = _lib.lib().dgcnn_forward_form(ref_b.num_nodes, ref_b.num_edges, B, sh.num_features, fl, ref_b.max_nodes)
    csr_ref = _ws_regions(m, ["rowptr", "colidx"]) if not form & 1 else None
    with torch.no_grad():
        lp = m(pb).clone()
    m.check_errors()
    r = _ws_regions(m, names)
    assert torch.equal(pb.x, ref_b.x.cuda()) and torch.equal(pb.batch, ref_b.batch.cuda()) and torch.equal(pb.y, ref_b.y.cuda())
    for n in names:
        assert torch.equal(r[n], r_ref[n]), n
    if csr_ref is not None:
        c = _ws_regions(m, ["rowptr", "colidx"])
        assert torch.equal(c["rowptr"], csr_ref["rowptr"]) and torch.equal(c["colidx"], csr_ref["colidx"])
    assert torch.equal(lp, lp_ref)
    # drop-in training forward + autograd backward (same dropout stream)
    grads = []
    for data in (ref_b.to("cuda"), pb):
        m.train(); m._seed_base, m._fwd_count = 5, 0
        m.zero_grad(set_to_none=True)
        out = m(data)
        torch.nn.functional.nll_loss(out, data.y).backward()
        grads.append([p.grad.detach().clone() for p in m._param_list()])
    for a, b, p in zip(grads[0], grads[1], m.state_dict().keys()):
        assert torch.equal(a, b), p


MODES = [("NCI1", 90, 50, dict(agg_mode="dense")),                       # dense form, F > 32: conv1's backward is the gather kernel
         ("NCI1", 90, 50, dict()),                                        # (the library's own choice for that shape)
         ("COLLAB", 90, 50, dict(agg_mode="dense", use_fused=True)),      # forced graph-per-workgroup forward in the dense form
         ("COLLAB", 90, 50, dict(use_fused=True)),                        # ... and in the CSR form
         ("PROTEINS", 80, 40, dict(agg_mode="dense", use_chain=False))]   # dense per-layer kernels, F <= 32: the bitmap alone


@pytest.mark.parametrize("name,G,B,mode", MODES, ids=[f"{c[0]}-{'-'.join(f'{k}={v}' for k, v in c[3].items()) or 'auto'}" for c in MODES])
def test_assembled_batch_carries_every_structure_its_forced_form_reads(name, G, B, mode):
    """forms in which a kernel reads the CSR although dg_form says dense (conv1's own backward above the
    aggregate-first width; a forced graph-per-workgroup forward): the assembled batch must give the per-batch path's
    log-probabilities and gradients bit for bit.  The workspace slot's CSR is wiped between the two runs so that a structure the
    assembly skipped cannot be inherited from the per-batch run."""
    from dgcnn_amd.device_data import PreparedDataset
    sh = synth.SHAPES[name]
    graphs = [g for g in synth.make_graphs(name, G, start=700) if g.num_nodes <= 256]
    ds = PreparedDataset(graphs)
    ids = np.random.default_rng(11).permutation(len(graphs))[:B]
    ref_b = collate([graphs[i] for i in ids])
    pb = ds.batch_of(ids)
    m = make_model(sh.num_features, sh.num_classes)
    for k, v in mode.items():
        setattr(m, k, v)
    res = []
    for data in (ref_b.to("cuda"), pb):
        m.eval()
        with torch.no_grad():
            lp = m(data).clone()
        m.check_errors()
        m.train(); m._seed_base, m._fwd_count = 5, 0
        m.zero_grad(set_to_none=True)
        out = m(data)
        torch.nn.functional.nll_loss(out, data.y).backward()
        m.check_errors()
        res.append((lp, [p.grad.detach().clone() for p in m._param_list()]))
        for name_ in ("rowptr", "colidx"):      # wipe the CSR of this workspace slot (zeros: wrong numbers, never a wild read)
            m.last_workspace_view(name_).zero_()
    assert torch.equal(res[0][0], res[1][0])
    for a, b, p in zip(res[0][1], res[1][1], m.state_dict().keys()):
        assert torch.equal(a, b), p


def test_dataset_of_large_graphs_skips_the_bitmap_and_keeps_its_batches_on_the_csr_kernels():
    """DD-like sets (more than 5 % of the graphs above 512 nodes): no bitmap rows are built (124 B per dataset node of dead
    weight), every batch carries the CSR-only mode flags -- also a batch whose graphs would all admit the bitmap forms -- and
    equals the per-batch path restricted to the same kernels bit for bit"""
    from dgcnn_amd.device_data import PreparedDataset
    sh = synth.SHAPES["DD"]
    graphs = synth.make_graphs("DD", 40, start=300)
    graphs += [synth.make_graphs("DD", 1, start=9100 + k, force_first_n=560 + 40 * k)[0] for k in range(4)]
    assert np.mean([g.num_nodes > 512 for g in graphs]) > 0.05
    ds = PreparedDataset(graphs)
    assert ds.adj_bits is None
    small = np.array([i for i, g in enumerate(graphs) if g.num_nodes <= 256][:8])
    mixed = np.arange(10)
    m = make_model(sh.num_features, sh.num_classes)
    for ids in (small, mixed):
        pb = ds.batch_of(ids)
        assert pb.mode_flags == (_lib.FLAG_AGG_SPARSE | _lib.FLAG_NO_CHAIN)
        ref_b = collate([graphs[i] for i in ids])
        m.agg_mode, m.use_chain = "sparse", False
        m.eval()
        with torch.no_grad():
            lp_ref = m(ref_b.to("cuda")).clone()
        m.check_errors()
        del m.__dict__["agg_mode"], m.__dict__["use_chain"]
        with torch.no_grad():
            lp = m(pb).clone()
        m.check_errors()
        assert torch.equal(lp, lp_ref)
    with pytest.raises(_lib.DgcnnError):
        PreparedDataset(graphs, bitmap=False).batch_of(small).to("cpu")


TRAJ = [("COLLAB", 260, 50), ("MUTAG", 200, 50), ("PROTEINS", 150, 32), ("COLLAB", 1400, 300), ("COLLAB", 2600, 600),
        ("DD", 60, 10)]


@pytest.mark.parametrize("name,G,bs", TRAJ, ids=[f"{c[0]}-{c[1]}-{c[2]}" for c in TRAJ])
def test_prepared_loader_trains_the_same_trajectory_as_the_per_batch_loader(name, G, bs):
    """two epochs of Trainer.train_epoch (pipelined: the next batch's assembly rides on the step / runs on the side stream /
    in-stream, depending on the batch size) + a test epoch, prepared vs per-batch DeviceLoader under the same shuffle: identical
    losses, accuracies and final parameters, bit for bit; short last batches included"""
    from dgcnn_amd.device_data import DeviceDataset, DeviceLoader, PreparedDataset
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES[name]
    graphs = synth.make_graphs(name, G, start=50, labels="structure")
    if name == "COLLAB":
        graphs = [g for g in graphs if g.num_nodes <= 256]
    G = len(graphs)
    out = []
    for prepared in (False, True):
        ds = PreparedDataset(graphs, bitmap=True) if prepared else DeviceDataset(graphs)
        m = make_model(sh.num_features, sh.num_classes)
        m._seed_base, m._fwd_count = 9, 0
        tr = Trainer(m)
        gen = torch.Generator().manual_seed(4)
        ld = DeviceLoader(ds, bs, shuffle=True, generator=gen, prepared=prepared)
        stats = [tr.train_epoch(ld, G) for _ in range(2)]
        m.eval()
        stats.append(tr.test_epoch(DeviceLoader(ds, bs, prepared=prepared), G))
        torch.cuda.synchronize()
        out.append((stats, m.flat_params.clone()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])


def test_prepared_batches_of_the_benched_shape_take_the_one_launch_kernel_with_the_assembly_riding():
    """COLLAB-50 from a prepared dataset: form = CHAIN_TAIL, and a pipelined step whose NEXT batch is prepared leaves that
    batch's structures in the other workspace slot (compared with a stand-alone dgcnn_assemble of the same ids)"""
    from dgcnn_amd.device_data import PreparedDataset
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    graphs = [g for g in synth.make_graphs("COLLAB", 160, start=0) if g.num_nodes <= 256]
    ds = PreparedDataset(graphs, keep_edge_lists=False)
    a, b = ds.batch_of(np.arange(50)), ds.batch_of(np.arange(50, 100))
    m = make_model(sh.num_features, sh.num_classes)
    fl = m._mode_flags() | _lib.FLAG_COALESCED_UNDIRECTED
    assert _lib.lib().dgcnn_forward_form(a.num_nodes, a.num_edges, 50, 1, fl, a.max_nodes) & 4
    m.train()
    tr = Trainer(m)
    tr.train_step(a, a.y, next_data=b)
    tr.train_step(b, b.y, next_data=a)
    torch.cuda.synchronize()
    tr.read_metrics()
    ref = collate([graphs[i] for i in range(50, 100)])
    assert torch.equal(b.x.cpu(), ref.x) and torch.equal(b.y.cpu(), ref.y) and torch.equal(b.batch.cpu(), ref.batch)
    assert torch.equal(m.last_workspace_view("graph_ptr").cpu(), torch.searchsorted(ref.batch, torch.arange(51)).int())


def test_bad_graph_ids_and_foreign_prefix_sums_are_flagged():
    from dgcnn_amd.device_data import PreparedBatch, PreparedDataset
    sh = synth.SHAPES["MUTAG"]
    graphs = synth.make_graphs("MUTAG", 30, start=0)
    ds = PreparedDataset(graphs)
    m = make_model(sh.num_features, sh.num_classes).eval()
    good = ds.batch_of(np.arange(10))
    with torch.no_grad():
        m(good)
    m.check_errors()
    ids_d, meta = good._keep
    bad_ids = ids_d.clone(); bad_ids[4] = 999
    bad = PreparedBatch(ds, good.x, good.batch, good.y, 10, good.num_nodes, good.num_edges, good.max_nodes, good.max_edges,
                        bad_ids.data_ptr(), good.onode_ptr, good.oedge_ptr, keep=(bad_ids, meta))
    with torch.no_grad():
        m(bad)
    with pytest.raises(_lib.DgcnnError):
        m.check_errors()
    other = ds.batch_of(np.arange(10, 20))          # prefix sums of ANOTHER graph list under these ids
    if other.num_nodes >= good.num_nodes:
        mix = PreparedBatch(ds, good.x, good.batch, good.y, 10, good.num_nodes, good.num_edges, good.max_nodes, good.max_edges,
                            good.ids_ptr, other.onode_ptr, other.oedge_ptr, keep=(good._keep, other._keep))
        with torch.no_grad():
            m(mix)
        with pytest.raises(_lib.DgcnnError):
            m.check_errors()


def test_prepared_dataset_refuses_edge_lists_that_break_the_promise():
    from dgcnn_amd.batch import Graph
    from dgcnn_amd.device_data import PreparedDataset
    graphs = synth.make_graphs("MUTAG", 12, start=0)
    g = graphs[5]
    ei = g.edge_index.clone()
    ei = ei[:, 1:]                                   # drop one direction of one edge: no longer undirected
    broken = Graph(g.x, ei, g.y, True)               # the host still promises coalesced + undirected
    with pytest.raises(_lib.DgcnnError):
        PreparedDataset(graphs[:5] + [broken] + graphs[6:])
