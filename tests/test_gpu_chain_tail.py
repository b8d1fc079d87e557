"""The benched kernel on the benched workload, against the oracle.

``k_chain_readout_tail`` (gcn_chain.hip; DGCNN_FORM_CHAIN_TAIL) is what ``Trainer.train_step`` runs with DEFAULT flags for
coalesced-undirected batches of <= 256 graphs of <= 256 nodes each: chain forward (conv1..conv4, reference model.py:30-33),
SortPooling + dense tail forward (model.py:35-45) and the readout backward + conv4's backward (train.py:40) of one graph in
one workgroup.  It is 63 % of the headline step.  Every case below first asserts that this form IS the one the library
takes for the batch (``dgcnn_forward_form(...) & DGCNN_FORM_CHAIN_TAIL``), then compares loss, #correct and ALL 16
gradients with the independent fp64 dense oracle evaluated on the kernel's own dropout mask and SortPooling permutation
(the permutation itself validated as a legal top-k of the oracle's keys), and the per-node activations [N,97]."""
import numpy as np
import pytest
import torch

from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import Batch
from oracle import ref_dense
from parity_util import KEY_TOL, XCAT_TOL, cpu_state_dict, gpu_xcat, grads_close, load_fixture, make_model

pytestmark = pytest.mark.gpu

FORM_CHAIN_TAIL = 4
FORM_STEP = 8          # ... with the whole GCN backward inside the same launch (round 4)
KEYS = ["conv1.lin.weight", "conv1.bias", "conv2.lin.weight", "conv2.bias", "conv3.lin.weight", "conv3.bias",
        "conv4.lin.weight", "conv4.bias", "conv5.weight", "conv5.bias", "conv6.weight", "conv6.bias",
        "classifier_1.weight", "classifier_1.bias", "classifier_2.weight", "classifier_2.bias"]


def form_of(m, b):
    fl = m._mode_flags() | (_lib.FLAG_COALESCED_UNDIRECTED if b.coalesced_undirected else 0)
    return _lib.lib().dgcnn_forward_form(b.num_nodes, b.num_edges, b.num_graphs, int(b.x.shape[1]), fl, int(b.max_nodes or 0))


def batch_with_small_graphs(name, bs, start, limit=256):
    """first seeded batch of the shape whose largest graph has <= `limit` nodes (the one-launch kernel's admission bound)"""
    for k in range(64):
        b = synth.make_batch(name, bs, start=start + k * bs)
        if b.max_nodes <= limit:
            return b
    raise AssertionError(f"no {name} batch of {bs} graphs with max_nodes <= {limit}")


def fused_step_vs_oracle(m, b_cpu, training=True):
    from dgcnn_amd.train import Trainer
    sd = cpu_state_dict(m)
    assert form_of(m, b_cpu) > 0 and form_of(m, b_cpu) & FORM_CHAIN_TAIL, \
        f"library does not take the one-launch training kernel for this batch (form {form_of(m, b_cpu)})"
    b = b_cpu.to("cuda")
    if training:
        m.train(); m._seed_base, m._fwd_count = 11, 0
    else:
        m.eval()
    tr = Trainer(m)
    tr.reset_metrics()
    tr.train_step(b, b.y)
    torch.cuda.synchronize()
    m.check_errors()
    lsum, correct = tr.read_metrics()
    perm = m.last_workspace_view("perm").cpu()
    mask = m.last_workspace_view("drop_mask").cpu() if training else None
    if training:
        frac = float(mask.float().mean())
        assert 0.35 < frac < 0.65, frac
    logp_ref, loss_ref, g_ref, aux = ref_dense.loss_and_grads_dense(sd, b_cpu.x, b_cpu.edge_index, b_cpu.batch, b_cpu.y,
                                                                    b_cpu.num_graphs, dropout_mask=mask, perm_override=perm)
    # forward half: activations and the legality of the kernel's selection on the ORACLE's keys
    err_x = float((gpu_xcat(m).double() - aux["xcat"].detach()).abs().max())
    assert err_x <= XCAT_TOL, f"per-node activations differ by {err_x:.3e}"
    ok, msg = ref_dense.check_perm_valid(aux["xcat"], aux["ptr"], perm, tol=KEY_TOL)
    assert ok, msg
    assert abs(lsum - float(loss_ref)) < 1e-5, (lsum, float(loss_ref))
    C = logp_ref.shape[1]
    top2 = logp_ref.detach().topk(min(2, C), dim=1).values
    sure = (top2[:, 0] - top2[:, -1]) > 1e-4
    want = (logp_ref.detach().argmax(1) == b_cpu.y)
    assert abs(correct - float(want.sum())) <= float((~sure).sum())
    g = tr.grads.cpu()
    worst = {}
    for p, off, key in zip(m._param_list(), m._offsets, KEYS):
        good, md, sc = grads_close(g[off:off + p.numel()], g_ref[key].reshape(-1))
        worst[key] = (md, sc)
        assert good, f"grad {key}: max diff {md:.3e} at scale {sc:.3e}"
    return tr, sd, worst


CASES = [("COLLAB", 50), ("MUTAG", 50), ("PROTEINS", 50), ("IMDB", 50), ("COLLAB", 256), ("COLLAB_REAL", 50), ("COLLAB", 1), ("COLLAB", 7)]


@pytest.mark.parametrize("name,bs", CASES, ids=[f"{c[0]}-{c[1]}" for c in CASES])
def test_one_launch_training_kernel_vs_fp64_oracle(name, bs):
    """Trainer.train_step, default flags, BASELINE workloads at their batch size (COLLAB-50 is the benched headline;
    COLLAB-256 config 5's global batch = the kernel's upper admission bound): loss, #correct, all 16 gradients"""
    sh = synth.SHAPES[name]
    b_cpu = batch_with_small_graphs(name, bs, start=1000)
    assert b_cpu.coalesced_undirected
    m = make_model(sh.num_features, sh.num_classes)
    # round 4: the kernel also runs conv4 / conv3 / conv2's backward and conv1's weight gradient of its graph (FORM_STEP) --
    # the eight GCN gradients checked below come out of k_chain_readout_tail's partial rows, one per graph
    assert form_of(m, b_cpu) & FORM_STEP, form_of(m, b_cpu)
    fused_step_vs_oracle(m, b_cpu)


@pytest.mark.parametrize("F", [2, 8, 9, 13, 16, 17, 24, 31, 32])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_one_launch_training_kernel_raw_feature_widths(F, dtype):
    """every instantiation of the step kernel by feature width (F <= 8, <= 16, <= 32: staging items per thread, conv1's weight
    table depth, the in-kernel conv1 weight gradient with one or two column blocks, its one- and two-round cross-wave sums):
    Trainer.train_step vs the fp64 oracle, all 16 gradients; a 150-node graph (10 live waves) and a 230-node graph (15) in
    the batch.  bf16: equal to the nn.Module route of the same leg (the one held to the leg's stated tolerances)"""
    from dgcnn_amd.train import Trainer
    base = None
    for k in range(64):
        base = synth.make_batch("COLLAB" if F % 2 else "PROTEINS", 14, start=300 + 14 * k, force_first_n=150 if F % 2 else 230)
        if base.max_nodes <= 256:
            break
    assert base.max_nodes <= 256
    g = torch.Generator().manual_seed(F)
    b_cpu = Batch(torch.randn(base.x.shape[0], F, generator=g), base.edge_index, base.batch, base.y, base.num_graphs,
                  base.coalesced_undirected, base.max_nodes, base.max_edges)
    m = make_model(F, 3)
    m.compute_dtype = dtype
    assert form_of(m, b_cpu) & FORM_STEP, form_of(m, b_cpu)
    if dtype == "fp32":
        fused_step_vs_oracle(m, b_cpu)
        return
    m.train(); m._seed_base, m._fwd_count = 5, 0
    tr = Trainer(m)
    b = b_cpu.to("cuda")
    tr.train_step(b, b.y)
    torch.cuda.synchronize()
    m.check_errors()
    ga, pa = tr.grads.cpu().clone(), m.last_workspace_view("perm").cpu().clone()
    m2 = make_model(F, 3)
    m2.compute_dtype = dtype
    m2.train(); m2._seed_base, m2._fwd_count = 5, 0
    out = m2(b)
    torch.nn.functional.nll_loss(out, b.y).backward()
    m2.check_errors()
    assert torch.equal(m2.last_workspace_view("perm").cpu(), pa)
    sc = float(ga.abs().max())
    for p, off in zip(m2._param_list(), m2._offsets):
        d = float((ga[off:off + p.numel()] - p.grad.detach().reshape(-1).cpu()).abs().max())
        assert d <= 2e-5 * sc + 1e-9, (off, d, sc)


@pytest.mark.parametrize("C", [2, 7, 8, 9, 16, 17, 33, 64])
def test_one_launch_training_kernel_class_counts(C):
    """classifier_2 / log_softmax / the loss gradient of the step kernel by class count (a wave per class in the forward; the
    backward keeps the first 8 rows of classifier_2 in registers, rows 8..15 in LDS, the rest in place): vs the fp64 oracle"""
    base = batch_with_small_graphs("COLLAB", 20, start=640)
    g = torch.Generator().manual_seed(C)
    y = torch.randint(0, C, (base.num_graphs,), generator=g)
    b_cpu = Batch(base.x, base.edge_index, base.batch, y, base.num_graphs, base.coalesced_undirected, base.max_nodes, base.max_edges)
    m = make_model(1, C)
    assert form_of(m, b_cpu) & FORM_STEP
    fused_step_vs_oracle(m, b_cpu)


@pytest.mark.parametrize("sizes,isolated", [
    ([31, 32, 33, 63, 64, 65], ()), ([127, 128, 129, 16, 15, 17], ()), ([191, 192, 193, 1, 2, 3], (0, 2)),
    ([255, 256, 29, 30, 31], ()), ([1, 1, 1, 40], ()), ([2], ()), ([96, 97, 111, 112, 113, 209, 240], (1, 5))],
    ids=["words", "tile_boundary", "mixed", "max256_and_k", "single_nodes", "one_tiny_graph", "tile_gap"])
def test_one_launch_training_kernel_boundary_sizes_isolated_and_single_node_graphs(sizes, isolated):
    """graph sizes at the 32-node bitmap-word and 16-node tile boundaries, at SortPooling's k = 30 and at the kernel's bound of
    256 nodes, single-node graphs (no edge at all) and nodes without neighbours: Trainer.train_step vs the fp64 oracle"""
    from test_gpu_dense import _sized_batch
    b_cpu = _sized_batch(sizes, seed=sum(sizes), isolated=isolated)
    m = make_model(3, 2)
    # (more graphs than 16-node tiles in the batch -- three single-node graphs and one of 40 nodes -- leaves the GCN backward to
    #  the launch-per-layer kernels: one partial row per graph would exceed the rows k_wgrad's workspace holds)
    want = FORM_CHAIN_TAIL if sizes == [1, 1, 1, 40] else FORM_STEP
    assert form_of(m, b_cpu) & want, form_of(m, b_cpu)
    fused_step_vs_oracle(m, b_cpu)


STEP_CASES = [("COLLAB", 50, "fp32"), ("MUTAG", 50, "fp32"), ("PROTEINS", 50, "fp32"), ("COLLAB", 256, "fp32"), ("COLLAB", 3, "fp32"),
              ("COLLAB", 50, "bf16"), ("PROTEINS", 50, "bf16"), ("MUTAG", 50, "bf16")]


@pytest.mark.parametrize("name,bs,dtype", STEP_CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}" for c in STEP_CASES])
def test_in_kernel_gcn_backward_equals_the_launch_per_layer_form(name, bs, dtype):
    """the same training step with the GCN backward inside k_chain_readout_tail (default) and as the round-3 launches
    (conv4's in the kernel, conv3 / conv2 as gather kernels; dgcnn_step_kernel_enable(0)): same dropout mask, same selection,
    identical loss, gradients equal to fp32 summation-order noise; eval mode and a 190-node graph (12 live waves) included.
    bf16 cases (BASELINE config 3): the kernel's forward half keeps hs_2 / hs_3 as one bf16 part, its backward is the same fp32
    code reading fp32 weight tables rebuilt in LDS; a third run takes the nn.Module route (k_chain_fwd_q's bf16 form, the
    readout and the backward as separate launches: the route test_gpu_configs.py checks against the fp64 oracle) on the same
    dropout seed"""
    from dgcnn_amd.train import Trainer
    L = _lib.lib()
    sh = synth.SHAPES[name]
    b_cpu = batch_with_small_graphs(name, bs, start=2000)
    if name == "COLLAB" and bs == 50:
        for k in range(64):                                                        # T = 12 live waves, K32 = 6
            b_cpu = synth.make_batch("COLLAB", 50, start=2000 + 50 * k, force_first_n=190)
            if b_cpu.max_nodes <= 256:
                break
        assert 190 <= b_cpu.max_nodes <= 256
    res = []
    prev = L.dgcnn_step_kernel_enable(1)
    try:
        for on in (1, 0):
            L.dgcnn_step_kernel_enable(on)
            m = make_model(sh.num_features, sh.num_classes)
            m.compute_dtype = dtype
            assert bool(form_of(m, b_cpu) & FORM_STEP) == bool(on)
            m.train(); m._seed_base, m._fwd_count = 5, 0
            tr = Trainer(m)
            tr.reset_metrics()
            b = b_cpu.to("cuda")
            tr.train_step(b, b.y)
            torch.cuda.synchronize()
            m.check_errors()
            res.append((tr.read_metrics(), tr.grads.cpu().clone(), m.last_workspace_view("perm").cpu().clone(),
                        m.flat_params.detach().cpu().clone()))
    finally:
        L.dgcnn_step_kernel_enable(prev)
    (ma, ga, pa, wa), (mb, gb, pb, wb) = res
    assert torch.equal(pa, pb)
    assert ma == mb, (ma, mb)                                   # the forward half is the same code: bitwise
    sc = float(gb.abs().max())
    assert float((ga - gb).abs().max()) <= 2e-5 * sc + 1e-9, (float((ga - gb).abs().max()), sc)
    assert float((wa - wb).abs().max()) <= 2.1e-3               # one Adam step: |dw| <= lr on either route
    if dtype == "bf16":
        m = make_model(sh.num_features, sh.num_classes)
        m.compute_dtype = dtype
        m.train(); m._seed_base, m._fwd_count = 5, 0
        b = b_cpu.to("cuda")
        out = m(b)
        torch.nn.functional.nll_loss(out, b.y).backward()
        m.check_errors()
        assert torch.equal(m.last_workspace_view("perm").cpu(), pa)
        for p, off in zip(m._param_list(), m._offsets):
            d = float((ga[off:off + p.numel()] - p.grad.detach().reshape(-1).cpu()).abs().max())
            assert d <= 2e-5 * sc + 1e-9, (off, d, sc)
        # ... and the leg really ran in bf16: the fp32 step's gradients differ by more than order noise
        m32 = make_model(sh.num_features, sh.num_classes)
        m32.train(); m32._seed_base, m32._fwd_count = 5, 0
        t32 = Trainer(m32)
        t32.train_step(b, b.y)
        torch.cuda.synchronize()
        assert float((t32.grads.cpu() - ga).abs().max()) > 1e-4 * sc


def test_bench_pool_batches_take_the_one_launch_kernel():
    """every batch of bench.py's default pool (40 COLLAB batches of 50, graph ids from 0) runs in the form this file tests"""
    sh = synth.SHAPES["COLLAB"]
    m = make_model(sh.num_features, sh.num_classes)
    graphs = synth.make_graphs("COLLAB", 40 * 50, start=0)
    forms = [form_of(m, synth.collate(graphs[i:i + 50])) for i in range(0, len(graphs), 50)]
    assert all(f > 0 and f & FORM_CHAIN_TAIL for f in forms), forms


def test_admission_bounds_of_the_one_launch_kernel():
    """257 graphs, a graph above 256 nodes, or no coalesced-undirected promise: another form (each oracle-tested elsewhere)"""
    sh = synth.SHAPES["COLLAB"]
    m = make_model(sh.num_features, sh.num_classes)
    assert not form_of(m, synth.make_batch("COLLAB", 257, start=0)) & FORM_CHAIN_TAIL
    big = synth.make_batch("COLLAB", 50, start=0, force_first_n=300)
    assert big.max_nodes == 300 and not form_of(m, big) & FORM_CHAIN_TAIL
    b = batch_with_small_graphs("COLLAB", 50, 1000)
    raw = Batch(b.x, b.edge_index, b.batch, b.y)          # no promise
    assert not form_of(m, raw) & FORM_CHAIN_TAIL
    m.use_chain = False
    assert not form_of(m, b) & FORM_CHAIN_TAIL


@pytest.mark.parametrize("name", ["mutag_b6", "proteins_b5", "collab_b4"])
def test_golden_step_fixture_through_the_one_launch_kernel(golden_dir, name):
    """the committed step fixtures WITH the coalesced-undirected promise their edge lists satisfy,
    i.e. through k_chain_readout_tail: stored fp64 loss, gradients, parameters after one Adam step, stored permutation"""
    from dgcnn_amd.train import Trainer
    z, sd, grads, b = load_fixture(golden_dir, name, coalesced_undirected=True)
    m = make_model(int(z["num_features"]), int(z["num_classes"]), sd)
    assert form_of(m, b) & FORM_CHAIN_TAIL
    m.eval()
    tr = Trainer(m)
    tr.reset_metrics()
    before = m.flat_params.clone()
    tr.train_step(b.to("cuda"), b.y.to("cuda"))
    loss, _ = tr.read_metrics()
    m.check_errors()
    assert abs(loss - float(z["loss_eval_f64"])) < 1e-5
    np.testing.assert_array_equal(m.last_workspace_view("perm").cpu().numpy(), z["perm"])
    flat, g = m.flat_params.cpu(), tr.grads.cpu()
    for p, off, key in zip(m._param_list(), m._offsets, KEYS):
        n = p.numel()
        g_ref = torch.from_numpy(z["grad_eval:" + key]).reshape(-1)
        a_ref = torch.from_numpy(z["adam1:" + key]).reshape(-1)
        good, md, sc = grads_close(g[off:off + n], g_ref)
        assert good, f"grad {key}: max diff {md:.3e} at scale {sc:.3e}"
        sure = g_ref.abs() > 1e-3 * g_ref.abs().max().clamp_min(1e-30)
        assert (flat[off:off + n][sure] - a_ref[sure]).abs().max() < 5e-6, key
        assert torch.equal(flat[off:off + n][g_ref == 0], before.cpu()[off:off + n][g_ref == 0]), key


@pytest.mark.parametrize("name", ["mutag_b6", "proteins_b5", "collab_b4"])
def test_golden_fixture_training_mode_through_the_one_launch_kernel(golden_dir, name):
    z, sd, grads, b = load_fixture(golden_dir, name, coalesced_undirected=True)
    m = make_model(int(z["num_features"]), int(z["num_classes"]), sd)
    fused_step_vs_oracle(m, b)
    np.testing.assert_array_equal(m.last_workspace_view("perm").cpu().numpy(), z["perm"])


def test_pipelined_steps_of_the_one_launch_kernel_track_the_oracle_trajectory():
    """three consecutive pipelined steps (next batch's graph preparation riding on the previous launch) on COLLAB-50: after
    every step the gradient matches the oracle evaluated at the parameters the PREVIOUS steps left (read back from the device)"""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    bs_cpu = [batch_with_small_graphs("COLLAB", 50, 2000 + 500 * i) for i in range(3)]
    bs = [b.to("cuda") for b in bs_cpu]
    m = make_model(sh.num_features, sh.num_classes)
    m.train(); m._seed_base, m._fwd_count = 3, 0
    tr = Trainer(m)
    for i in range(3):
        assert form_of(m, bs_cpu[i]) & FORM_CHAIN_TAIL
        sd = cpu_state_dict(m)
        tr.reset_metrics()
        tr.train_step(bs[i], bs[i].y, next_data=bs[(i + 1) % 3])
        torch.cuda.synchronize()
        lsum, _ = tr.read_metrics()
        perm = m.last_workspace_view("perm").cpu(); mask = m.last_workspace_view("drop_mask").cpu()
        b = bs_cpu[i]
        _, loss_ref, g_ref, _ = ref_dense.loss_and_grads_dense(sd, b.x, b.edge_index, b.batch, b.y, b.num_graphs,
                                                               dropout_mask=mask, perm_override=perm)
        assert abs(lsum - float(loss_ref)) < 1e-5
        g = tr.grads.cpu()
        for p, off, key in zip(m._param_list(), m._offsets, KEYS):
            good, md, sc = grads_close(g[off:off + p.numel()], g_ref[key].reshape(-1))
            assert good, f"step {i} grad {key}: max diff {md:.3e} at scale {sc:.3e}"


@pytest.mark.parametrize("violation", ["missing_reverse", "unsorted", "edge_leaves_graph"])
def test_fused_preparation_flags_a_broken_promise_in_the_next_batch(violation):
    """the NEXT batch's graph preparation runs as rider workgroups of the training kernel's launch (both phases, phase B behind a
    device counter; the reverse-edge check reads the batch's own edge list there): a next batch whose coalesced + undirected
    promise is violated must be flagged exactly as by the stand-alone preparation -- read_metrics raises -- and a clean next
    batch must not be"""
    from dgcnn_amd.batch import Batch
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    good = [synth.make_batch("COLLAB", 50, start=4000 + 50 * k) for k in range(3)]
    assert all(g.coalesced_undirected and g.max_nodes <= 256 for g in good)
    b = good[1]
    ei = b.edge_index.clone()
    if violation == "missing_reverse":
        s, d = int(ei[0, 777]), int(ei[1, 777])
        ei = ei[:, ~((ei[0] == d) & (ei[1] == s))]
    elif violation == "unsorted":
        ei[:, [100, 101]] = ei[:, [101, 100]]
    else:      # an edge between two different graphs (both directions, list kept sorted by source)
        ptr = torch.searchsorted(b.batch, torch.arange(b.num_graphs + 1))
        u, v = int(ptr[0]), int(ptr[1])                                     # first node of graph 0, first node of graph 1
        extra = torch.tensor([[u, v], [v, u]])
        ei = torch.cat([ei, extra], 1)
        order = torch.argsort(ei[0] * b.num_nodes + ei[1])
        ei = ei[:, order]
    bad = Batch(b.x, ei.contiguous(), b.batch, b.y, b.num_graphs, True, b.max_nodes, max(b.max_edges, ei.shape[1]))
    for nxt, expect_error in ((good[1], False), (bad, True)):
        m = make_model(sh.num_features, sh.num_classes)
        m.train()
        tr = Trainer(m)
        tr.reset_metrics()
        b0, b1 = good[0].to("cuda"), nxt.to("cuda")
        tr.train_step(b0, b0.y, next_data=b1)          # (first step: no counter yet -- classic riders)
        tr.train_step(b1, b1.y, next_data=b0)
        tr.train_step(b0, b0.y, next_data=b1)          # the fused form prepares b1 here ...
        tr.train_step(b1, b1.y)                        # ... and this step consumes it
        torch.cuda.synchronize()
        if expect_error:
            with pytest.raises(_lib.DgcnnError):
                tr.read_metrics()
        else:
            tr.read_metrics()


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("violation", ["missing_reverse", "none"])
def test_reverse_edge_check_inside_the_one_launch_kernels_from_96_graphs_on(mode, violation):
    """from 96 graphs per batch on the reverse-edge half of the coalesced + undirected promise is verified by the one-launch
    training / evaluation kernel itself, on the LDS image of each graph's bitmap (round 5: phase B of the preparation, a rider of
    k_wgrad at these sizes, no longer searches every edge's reverse): one missing reverse edge in ONE graph of 128 must be
    flagged, stand-alone and as the look-ahead batch of a pipelined step, and a clean batch must not be"""
    from dgcnn_amd.batch import Batch
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    good = [batch_with_small_graphs("COLLAB", 128, start=6000 + 1000 * k) for k in range(2)]
    b = good[1]
    ei = b.edge_index.clone()
    if violation == "missing_reverse":
        e = int(ei.shape[1] * 0.61)
        sn, dn = int(ei[0, e]), int(ei[1, e])
        ei = ei[:, ~((ei[0] == dn) & (ei[1] == sn))]
        assert ei.shape[1] == b.edge_index.shape[1] - 1
    nxt = Batch(b.x, ei.contiguous(), b.batch, b.y, b.num_graphs, True, b.max_nodes, b.max_edges)
    for pipelined in (False, True):
        m = make_model(sh.num_features, sh.num_classes)
        m.train(mode == "train")
        tr = Trainer(m)
        tr.reset_metrics()
        fn = tr.train_step if mode == "train" else tr.eval_step
        b0, b1 = good[0].to("cuda"), nxt.to("cuda")
        if pipelined:
            fn(b0, b0.y, next_data=b1)
        fn(b1, b1.y)
        torch.cuda.synchronize()
        if violation == "none":
            tr.read_metrics()
        else:
            with pytest.raises(_lib.DgcnnError):
                tr.read_metrics()


def _run_steps(batches, nsteps, exclusive, stream=None, hog=None):
    """`nsteps` pipelined training steps over `batches` (round robin, look-ahead) on `stream`; returns the final parameters"""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    m = make_model(sh.num_features, sh.num_classes)
    m.train(); m._seed_base, m._fwd_count = 21, 0
    nb = len(batches)
    if stream is not None:
        stream.wait_stream(torch.cuda.current_stream())      # (the model was initialised on the current stream)
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        tr = Trainer(m, exclusive_device=exclusive)
        tr.reset_metrics()
        for i in range(nsteps):
            if hog is not None:
                hog()
            tr.train_step(batches[i % nb], batches[i % nb].y, next_data=batches[(i + 1) % nb])
    return m, tr


def test_fused_preparation_beside_a_busy_second_stream_is_bit_identical_and_unflagged():
    """the step kernel's in-launch wait (phase-B workgroups polling a counter that the phase-A workgroups of the
    SAME launch advance) beside foreign work -- a second stream keeps every compute unit busy with long matrix products while 60
    pipelined steps run.  Workgroups of one launch are dispatched in index order, so a waiter can never precede its producers;
    the foreign kernels never wait for anything of ours.  Must equal the undisturbed run bit for bit, no batch flagged."""
    batches = [batch_with_small_graphs("COLLAB", 50, start=7000 + 300 * k).to("cuda") for k in range(4)]
    ref_m, ref_tr = _run_steps(batches, 60, exclusive=True)
    torch.cuda.synchronize()
    ref_tr.read_metrics()
    ref = ref_m.flat_params.clone()
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    junk = []

    def hog():
        with torch.cuda.stream(side):
            junk.append(torch.mm(a, a))           # ~1 ms of every CU's time per call
            if len(junk) > 4:
                junk.pop(0)
    main = torch.cuda.Stream()
    main.wait_stream(torch.cuda.current_stream())
    m, tr = _run_steps(batches, 60, exclusive=True, stream=main, hog=hog)
    torch.cuda.synchronize()
    tr.read_metrics()                              # raises if any batch was flagged (layout promise or wait time-out)
    assert torch.equal(m.flat_params, ref)


def test_two_trainers_on_two_streams_without_the_exclusive_promise_keep_the_classic_riders():
    """two training loops in one process on two streams, interleaved launch by launch: the hazardous constellation for in-launch
    waits (each launch's waiting workgroups could hold the compute units the other's producers need).  Without the promise
    (exclusive_device=False) phase B rides on k_wgrad -- nobody waits on the device -- and both loops reproduce the single run"""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["COLLAB"]
    batches = [batch_with_small_graphs("COLLAB", 50, start=7000 + 300 * k).to("cuda") for k in range(4)]
    ref_m, ref_tr = _run_steps(batches, 24, exclusive=False)
    torch.cuda.synchronize()
    ref = ref_m.flat_params.clone()
    ex_m, ex_tr = _run_steps(batches, 24, exclusive=True)      # (and the promise changes nothing but the launch that carries phase B)
    torch.cuda.synchronize()
    assert torch.equal(ex_m.flat_params, ref)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ms, trs = [], []
    for st in streams:
        m = make_model(sh.num_features, sh.num_classes)
        m.train(); m._seed_base, m._fwd_count = 21, 0
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            ms.append(m); trs.append(Trainer(m, exclusive_device=False))
    for i in range(24):
        for st, tr in zip(streams, trs):
            with torch.cuda.stream(st):
                tr.train_step(batches[i % 4], batches[i % 4].y, next_data=batches[(i + 1) % 4])
    torch.cuda.synchronize()
    for m, tr in zip(ms, trs):
        tr.read_metrics()
        assert torch.equal(m.flat_params, ref)
