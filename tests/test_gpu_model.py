"""Whole-model GPU parity: the HIP path through the drop-in ``Model`` vs the oracle, on the
committed golden fixtures and on the BASELINE.json workload shapes (tie-aware protocol, see
tests/parity_util.py), plus the fused training step and its reproducibility."""
import numpy as np
import pytest
import torch

from dgcnn_amd import synth
from oracle import ref_dense, ref_ops
from parity_util import (LOGIT_TOL, check_backward_parity, check_forward_parity, cpu_state_dict, grads_close,
                         load_fixture, make_model)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["mutag_b6", "proteins_b5", "collab_b4"])
def test_golden_fixture_forward(golden_dir, name):
    z, sd, grads, b = load_fixture(golden_dir, name)
    m = make_model(int(z["num_features"]), int(z["num_classes"]), sd)
    logp, perm, err, err_x = check_forward_parity(m, b, sd)
    if float(z["sort_margin"]) >= 1e-5:
        # tie-free fixture (smallest deciding key gap >= 10x the fp32 key noise; the COLLAB one was searched for it):
        # must match the stored vectors directly, permutation included
        np.testing.assert_array_equal(perm.numpy(), z["perm"])
        np.testing.assert_allclose(logp.numpy(), z["logp_eval_f64"], rtol=0, atol=LOGIT_TOL)
        np.testing.assert_allclose(logp.numpy(), z["logp_eval_f32"], rtol=0, atol=LOGIT_TOL)


@pytest.mark.parametrize("name", ["mutag_b6", "proteins_b5", "collab_b4"])
def test_golden_fixture_gradients(golden_dir, name):
    """with the fixture's dropout mask unavailable to the kernel (it draws its own), compare through
    the oracle run on the kernel's mask; additionally the tie-free fixtures' stored perm must match."""
    z, sd, grads, b = load_fixture(golden_dir, name)
    m = make_model(int(z["num_features"]), int(z["num_classes"]), sd)
    check_backward_parity(m, b, sd)
    np.testing.assert_array_equal(m.last_workspace_view("perm").cpu().numpy(), z["perm"])


@pytest.mark.parametrize("name", ["mutag_b6", "proteins_b5", "collab_b4"])
def test_golden_step_fixture_loss_grads_and_post_adam_parameters(golden_dir, name):
    """SURVEY 8(c) C5 item 6, the step fixture: one training step of the fused Trainer in eval mode (no dropout mask to
    share) against the stored fp64 loss, gradients and parameters after ONE Adam step (torch defaults)."""
    from dgcnn_amd.train import Trainer
    z, sd, grads, b = load_fixture(golden_dir, name)
    m = make_model(int(z["num_features"]), int(z["num_classes"]), sd)
    m.eval()
    tr = Trainer(m)
    tr.reset_metrics()
    before = m.flat_params.clone()
    tr.train_step(b.to("cuda"), b.y.to("cuda"))
    loss, _ = tr.read_metrics()
    assert abs(loss - float(z["loss_eval_f64"])) < 1e-5
    offs = m._offsets
    flat, g = m.flat_params.cpu(), tr.grads.cpu()
    for p, off, key in zip(m._param_list(), offs, ["conv1.lin.weight", "conv1.bias", "conv2.lin.weight", "conv2.bias",
                                                  "conv3.lin.weight", "conv3.bias", "conv4.lin.weight", "conv4.bias",
                                                  "conv5.weight", "conv5.bias", "conv6.weight", "conv6.bias",
                                                  "classifier_1.weight", "classifier_1.bias", "classifier_2.weight",
                                                  "classifier_2.bias"]):
        n = p.numel()
        g_ref = torch.from_numpy(z["grad_eval:" + key]).reshape(-1)
        a_ref = torch.from_numpy(z["adam1:" + key]).reshape(-1)
        assert grads_close(g[off:off + n], g_ref), key
        # the first Adam step moves every element by ~lr * sign(g): compare where the sign is well determined
        sure = g_ref.abs() > 1e-3 * g_ref.abs().max().clamp_min(1e-30)
        assert (flat[off:off + n][sure] - a_ref[sure]).abs().max() < 5e-6, key
        assert torch.equal(flat[off:off + n][g_ref == 0], before.cpu()[off:off + n][g_ref == 0]), key


WORKLOADS = [("MUTAG", 50, None), ("PROTEINS", 50, None), ("COLLAB", 50, None), ("COLLAB_REAL", 50, None),
             ("IMDB", 50, None), ("DD", 50, None), ("DD", 8, 5748), ("DD", 50, 5748)]


@pytest.mark.parametrize("name,bs,force", WORKLOADS, ids=[f"{w[0]}-{w[1]}-{w[2]}" for w in WORKLOADS])
def test_workload_forward_and_backward(name, bs, force):
    """BASELINE.json configs at their batch size (DD also with the forced 5748-node graph: SURVEY D2 item 4's stress batch,
    at batch 8 and at the full batch of 50 -- the fp64 oracle's dense block of that graph is 264 MB)."""
    sh = synth.SHAPES[name]
    b = synth.make_batch(name, bs, start=1000, force_first_n=force)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


@pytest.mark.parametrize("F", [1, 2, 3, 7, 13, 16, 19, 32, 33, 40])
@pytest.mark.parametrize("fused", [False, True])
def test_raw_feature_widths_aggregate_first_and_linear_first(F, fused):
    """conv1 runs aggregate-first ((A x) W) for F <= 32 and linear-first (A (x W)) above: every lanes-per-row
    shape of the F-wide gather, both sides of the switch, forward + gradients (dW1 comes from the saved A x)."""
    base = synth.make_batch("COLLAB" if F % 2 else "PROTEINS", 12, start=77)
    g = torch.Generator().manual_seed(F)
    from dgcnn_amd.batch import Batch
    b = Batch(torch.randn(base.x.shape[0], F, generator=g), base.edge_index, base.batch, base.y, base.num_graphs,
              base.coalesced_undirected, base.max_nodes, base.max_edges)
    m = make_model(F, 3)
    m.use_fused = fused
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    if F <= 32:     # the slab saved for backward is exactly A_hat x
        ax = m.last_workspace_view("ax").cpu().double()
        ptr = torch.searchsorted(b.batch, torch.arange(b.num_graphs + 1))
        for gi in range(b.num_graphs):
            n0, n1 = int(ptr[gi]), int(ptr[gi + 1])
            A = ref_dense.dense_norm_adj(b.edge_index, n0, n1)
            np.testing.assert_allclose(ax[n0:n1].numpy(), (A @ b.x[n0:n1].double()).numpy(), rtol=0, atol=2e-5)
    check_backward_parity(m, b, sd)


@pytest.mark.parametrize("name,bs", [("MUTAG", 129), ("MUTAG", 540), ("PROTEINS", 260)])
def test_large_batch_two_stage_weight_gradients(name, bs):
    """large grids use the shallower gather depth / 64-register variants; from 257 graphs the readout pair switches to
    its two-workgroups-per-CU form: same parity bar as the reference-sized batches."""
    sh = synth.SHAPES[name]
    b = synth.make_batch(name, bs, start=300)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


def test_persistent_aggregation_kernel_is_bit_identical_to_the_tiled_one():
    """>= 2048 node tiles select the persistent, software-pipelined 32-wide aggregation kernel.  A graph's result must
    not depend on the batch it travels in, so 2400 graphs in one batch (persistent kernel) must reproduce, bit for bit,
    the log-probabilities of the same graphs in 4 batches of 600 (one tile per workgroup)."""
    from dgcnn_amd.batch import collate
    sh = synth.SHAPES["MUTAG"]
    graphs = synth.make_graphs("MUTAG", 2400, start=9000)
    m = make_model(sh.num_features, sh.num_classes)
    m.eval()
    with torch.no_grad():
        big = collate(graphs).to("cuda")
        assert (big.x.shape[0] + 15) // 16 >= 2048
        lp_big = m(big).clone()
        m.check_errors()
        parts = []
        for k in range(4):
            parts.append(m(collate(graphs[600 * k:600 * (k + 1)]).to("cuda")).clone())
            m.check_errors()
    assert torch.equal(lp_big, torch.cat(parts))


def test_edge_cases_isolated_selfloops_single_graph_empty_edges():
    # one graph, n < k, isolated nodes, input self loops, duplicate edge
    x = torch.randn(7, 5)
    ei = torch.tensor([[0, 1, 2, 2, 3, 3, 0], [1, 0, 2, 3, 2, 3, 1]])
    from dgcnn_amd.batch import Batch
    b = Batch(x, ei, torch.zeros(7, dtype=torch.int64), torch.tensor([1]))
    m = make_model(5, 2)
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)
    # no edges at all: every node is isolated -> out = tanh(x W + b)
    b2 = Batch(x, torch.zeros(2, 0, dtype=torch.int64), torch.tensor([0, 0, 0, 1, 1, 1, 1]), torch.tensor([0, 1]))
    check_forward_parity(m, b2, sd)
    check_backward_parity(m, b2, sd)


def test_broken_layout_promise_is_reported_not_miscomputed():
    from dgcnn_amd import _lib
    from dgcnn_amd.batch import Batch
    sh = synth.SHAPES["MUTAG"]
    b = synth.make_batch("MUTAG", 4, start=5)
    ei = b.edge_index.clone()
    ei[:, [0, 1]] = ei[:, [1, 0]]                      # no longer sorted by (src,dst)
    m = make_model(sh.num_features, sh.num_classes).eval()
    sd = cpu_state_dict(m)
    with torch.no_grad():
        m(Batch(b.x, ei, b.batch, b.y, coalesced_undirected=True).to("cuda"))
    with pytest.raises(_lib.DgcnnError):
        m.check_errors()
    # without the promise the general path handles the same edge list correctly
    check_forward_parity(m, Batch(b.x, ei, b.batch, b.y), sd)
    # out-of-range node id
    ei2 = b.edge_index.clone(); ei2[0, 0] = b.num_nodes + 3
    with torch.no_grad():
        m(Batch(b.x, ei2, b.batch, b.y).to("cuda"))
    with pytest.raises(_lib.DgcnnError):
        m.check_errors()


def test_fast_and_general_prep_paths_give_identical_model_output():
    from dgcnn_amd.batch import Batch
    sh = synth.SHAPES["COLLAB"]
    b = synth.make_batch("COLLAB", 12, start=3000)
    m = make_model(sh.num_features, sh.num_classes).eval()
    m.use_chain = False          # same kernel family on both sides (the chain forward needs the promise): the CSRs are what is compared
    with torch.no_grad():
        fast = m(b.to("cuda")).clone()
        gen = m(Batch(b.x, b.edge_index, b.batch, b.y, coalesced_undirected=False).to("cuda")).clone()
    assert torch.equal(fast, gen)


@pytest.mark.parametrize("name,bs", [("MUTAG", 50), ("PROTEINS", 50), ("COLLAB", 50), ("COLLAB_REAL", 20), ("DD", 6)])
def test_fused_graph_per_workgroup_path_is_bit_identical_to_tiled_path(name, bs):
    from dgcnn_amd import _lib
    sh = synth.SHAPES[name]
    b = synth.make_batch(name, bs, start=700)
    lim = _lib.lib().dgcnn_fused_max_nodes(sh.num_features)
    if b.max_nodes > lim:
        pytest.skip(f"largest graph {b.max_nodes} > fused limit {lim}")
    assert b.max_edges > 0
    m = make_model(sh.num_features, sh.num_classes)
    bg = b.to("cuda")
    outs = {}
    for fused in (True, False):
        m.use_fused = fused
        m.train(); m._seed_base, m._fwd_count = 5, 0
        lp = m(bg)
        torch.nn.functional.nll_loss(lp, bg.y).backward()
        outs[fused] = (lp.detach().clone(), m._last_flat_grad.clone(),
                       m.last_workspace_view("x3").clone(), m.last_workspace_view("perm").clone())
        m.check_errors()
    for a, c in zip(outs[True], outs[False]):
        assert torch.equal(a, c)


def test_fused_path_flags_bad_hints():
    from dgcnn_amd import _lib
    from dgcnn_amd.batch import Batch
    sh = synth.SHAPES["PROTEINS"]
    b = synth.make_batch("PROTEINS", 6, start=20)
    m = make_model(sh.num_features, sh.num_classes).eval()
    m.use_fused = True
    # max_nodes hint smaller than the largest graph
    assert b.max_nodes > 16
    small = Batch(b.x, b.edge_index, b.batch, b.y, coalesced_undirected=True, max_nodes=16, max_edges=b.max_edges)
    with torch.no_grad():
        m(small.to("cuda"))
    with pytest.raises(_lib.DgcnnError):
        m.check_errors()
    # an edge that leaves its graph (not block diagonal) under the fused path
    ei = b.edge_index.clone()
    n_first = int((b.batch == 0).sum())
    extra = torch.tensor([[0, n_first], [n_first, 0]])
    bad = Batch(b.x, torch.cat([ei, extra], 1), b.batch, b.y, coalesced_undirected=False, max_nodes=b.max_nodes,
                max_edges=b.max_edges + 2)
    with torch.no_grad():
        m(bad.to("cuda"))
    with pytest.raises(_lib.DgcnnError):
        m.check_errors()
    # the tiled path accepts cross-graph edges (it is a plain sparse aggregation)
    m.use_fused = False
    sd = cpu_state_dict(m)
    with torch.no_grad():
        m(bad.to("cuda"))
    m.check_errors()


def test_result_independent_of_batch_composition():
    """a graph's log-probs do not depend on which other graphs share the batch."""
    sh = synth.SHAPES["PROTEINS"]
    graphs = synth.make_graphs("PROTEINS", 6, start=77)
    from dgcnn_amd.batch import collate
    m = make_model(sh.num_features, sh.num_classes).eval()
    with torch.no_grad():
        full = m(collate(graphs).to("cuda")).cpu()
        solo = torch.cat([m(collate([g]).to("cuda")).cpu() for g in graphs])
    assert torch.equal(full, solo)        # same kernels, same per-graph order -> bitwise


def test_bitwise_reproducible_forward_backward():
    sh = synth.SHAPES["COLLAB"]
    b = synth.make_batch("COLLAB", 20, start=300).to("cuda")
    outs = []
    for _ in range(3):
        m = make_model(sh.num_features, sh.num_classes)
        m.train()
        m._seed_base, m._fwd_count = 7, 0
        lp = m(b)
        torch.nn.functional.nll_loss(lp, b.y).backward()
        outs.append((lp.detach().clone(), m._last_flat_grad.clone()))
    for lp, g in outs[1:]:
        assert torch.equal(lp, outs[0][0]) and torch.equal(g, outs[0][1])


def test_fused_train_step_equals_dropin_route_and_oracle_loss():
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["PROTEINS"]
    b_cpu = synth.make_batch("PROTEINS", 50, start=500)
    b = b_cpu.to("cuda")
    m1 = make_model(sh.num_features, sh.num_classes)
    m2 = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m1)
    for m in (m1, m2):
        m.train(); m._seed_base, m._fwd_count = 11, 0
    # route A: reference loop body with torch loss + torch Adam on the drop-in Model (train.py:37-42)
    opt = torch.optim.Adam(m1.parameters())
    pred = m1(b)
    loss = torch.nn.NLLLoss()(pred, b.y)
    loss.backward()
    gradA = m1._last_flat_grad.clone()
    opt.step(); opt.zero_grad()
    # route B: fused trainer
    tr = Trainer(m2)
    tr.forward_backward(b, b.y)
    gradB = tr.grads.clone()
    tr.optimizer_step()
    assert torch.equal(gradA, gradB)                    # same kernels either way (NLL grad in-kernel)
    np.testing.assert_allclose(m2.flat_params.cpu().numpy(), m1.flat_params.cpu().numpy(), rtol=1e-5, atol=1e-7)
    lsum, correct = tr.read_metrics()
    assert abs(lsum - float(loss.detach().cpu())) < 1e-5
    assert correct == float((pred.argmax(1) == b.y).sum().item())
    assert float(tr.grads.abs().max()) == 0.0           # zero_grad fused into the Adam kernel
    # route C: optimizer fused into the weight-gradient kernel (dgcnn_model_backward_step)
    m3 = make_model(sh.num_features, sh.num_classes)
    m3.train(); m3._seed_base, m3._fwd_count = 11, 0
    tr3 = Trainer(m3)
    tr3.train_step(b, b.y)
    # (the one-launch training kernel evaluates conv4's backward as a block product on the matrix cores, the per-op route
    #  as a CSR gather: the same sums in a different order -- equal within fp32 rounding, not bit for bit)
    ga, gb = tr3.grads.double(), gradB.double()
    assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()) + 1e-9
    np.testing.assert_allclose(m3.flat_params.cpu().numpy(), m2.flat_params.cpu().numpy(), rtol=1e-4, atol=2e-6)
    # and the loss agrees with the oracle on the kernel's own mask/perm
    mask = m2.last_workspace_view("drop_mask").cpu(); perm = m2.last_workspace_view("perm").cpu()
    _, loss_ref, _, _ = ref_dense.loss_and_grads_dense(sd, b_cpu.x, b_cpu.edge_index, b_cpu.batch, b_cpu.y,
                                                      b_cpu.num_graphs, dropout_mask=mask, perm_override=perm)
    assert abs(lsum - float(loss_ref)) < 1e-5


@pytest.mark.parametrize("name,total,bs", [("COLLAB", 60, 12), ("MUTAG", 2400, 600)], ids=["riders", "side_stream"])
def test_pipelined_step_gives_identical_training(name, total, bs):
    """dgcnn_pipeline_train_step (one call per step; graph prep of batch i+1 during step i -- as rider workgroups of the
    step's launches at the reference's batch sizes, as launches on the library's side stream from 512 graphs per step)
    must not change a single bit, whether or not the promised next batch actually comes next."""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES[name]
    batches = [b.to("cuda") for b in synth.make_batches(name, total, bs, start=4000)]
    outs = []
    for mode in ("plain", "pipelined", "pipelined_no_lookahead", "broken_promise", "interleaved_eval"):
        m = make_model(sh.num_features, sh.num_classes)
        m.train(); m._seed_base, m._fwd_count = 3, 0
        tr = Trainer(m)
        for it in range(12):
            b = batches[it % len(batches)]
            nxt = batches[(it + 1) % len(batches)]
            if mode == "plain":
                tr.train_step(b, b.y)
            elif mode == "pipelined":
                tr.train_step(b, b.y, next_data=nxt)
            elif mode == "pipelined_no_lookahead":
                tr.pipelined_step(b, b.y)
            elif mode == "broken_promise":        # promise a batch that does not come next
                tr.train_step(b, b.y, next_data=batches[(it + 2) % len(batches)])
            else:
                tr.train_step(b, b.y, next_data=nxt)
                if it % 3 == 0:                   # an unpipelined call in between must not clobber the prepared slot
                    keep = tr.metrics.clone()
                    fc = m._fwd_count
                    tr.eval_step(batches[-1], batches[-1].y)
                    tr.metrics.copy_(keep); m._fwd_count = fc; m.train()
        torch.cuda.synchronize()
        m.check_errors()
        outs.append((m.flat_params.clone(), tr.metrics.clone()))
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])


def test_pipelined_epochs_with_mixed_batch_sizes_cross_the_side_stream_threshold():
    """epochs whose batches alternate between the side-stream route (>= 512 graphs) and the rider route (short last batch,
    small batches): the look-ahead preparation of a batch may come from either, whatever the size of the step it overlaps
    with -- parameters and metrics identical to the un-pipelined steps, bit for bit, over two epochs"""
    from dgcnn_amd.train import Trainer
    from dgcnn_amd.batch import collate
    sh = synth.SHAPES["MUTAG"]
    graphs = synth.make_graphs("MUTAG", 1500, start=7000)
    cuts = [0, 600, 1200, 1250, 1500]                       # 600 (side), 600 (side), 50 (riders), 250 (riders)
    batches = [collate(graphs[a:b]).to("cuda") for a, b in zip(cuts[:-1], cuts[1:])]
    outs = []
    for pipelined in (False, True):
        m = make_model(sh.num_features, sh.num_classes)
        m.train(); m._seed_base, m._fwd_count = 9, 0
        tr = Trainer(m)
        for _ in range(2):
            if pipelined:
                tr.train_epoch(batches, 1500)
            else:
                tr.reset_metrics()
                for b in batches:
                    tr.train_step(b, b.y)
        torch.cuda.synchronize()
        m.check_errors()
        outs.append((m.flat_params.clone(), tr.metrics.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_pipelined_step_without_rider_fast_path_prepares_in_stream():
    """general edge lists (no coalesced_undirected promise) and edge-less batches cannot ride; the next batch is then
    prepared in-stream after the step -- same results as the unpipelined calls."""
    from dgcnn_amd.batch import Batch
    from dgcnn_amd.train import Trainer
    base = [b for b in synth.make_batches("MUTAG", 40, 10, start=50)]
    general = []
    for b in base:
        perm = torch.randperm(b.edge_index.shape[1], generator=torch.Generator().manual_seed(1))
        general.append(Batch(b.x, b.edge_index[:, perm], b.batch, b.y, b.num_graphs, False, b.max_nodes, b.max_edges))
    noedge = Batch(base[0].x, torch.zeros(2, 0, dtype=torch.int64), base[0].batch, base[0].y, base[0].num_graphs)
    seq = [g.to("cuda") for g in general] + [noedge.to("cuda")]
    outs = []
    for look in (False, True):
        m = make_model(8, 2)
        m.train(); m._seed_base, m._fwd_count = 9, 0
        tr = Trainer(m)
        for it in range(10):
            b = seq[it % len(seq)]
            tr.train_step(b, b.y, next_data=seq[(it + 1) % len(seq)] if look else None)
        torch.cuda.synchronize()
        m.check_errors()
        outs.append(m.flat_params.clone())
    assert torch.equal(outs[0], outs[1])


def test_pipelined_step_data_parallel_route_matches_fused_adam():
    """exp_avg == NULL selects forward+backward only (DP: all-reduce, then dgcnn_adam_step): same update."""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["MUTAG"]
    batches = [b.to("cuda") for b in synth.make_batches("MUTAG", 40, 10, start=10)]
    res = []
    for fuse in (True, False):
        m = make_model(sh.num_features, sh.num_classes)
        m.train(); m._seed_base, m._fwd_count = 5, 0
        tr = Trainer(m)
        for it in range(6):
            b = batches[it % 4]
            tr.pipelined_step(b, b.y, batches[(it + 1) % 4], fuse_adam=fuse)
            if not fuse:
                tr.optimizer_step()
        torch.cuda.synchronize()
        res.append(m.flat_params.clone())
    torch.testing.assert_close(res[0], res[1], rtol=1e-6, atol=1e-7)


def test_pipeline_refuses_unprepared_batch_claimed_as_prepared():
    from dgcnn_amd import _lib
    from dgcnn_amd.train import Trainer
    b = synth.make_batch("MUTAG", 5).to("cuda")
    m = make_model(8, 2); m.train()
    tr = Trainer(m)
    tr.pipelined_step(b, b.y)                       # creates the pipeline, nothing prepared
    ent = tr._step_args(b, b.y)
    ent[2].flags |= _lib.FLAG_PREPARED
    rc = _lib.lib().dgcnn_pipeline_train_step(tr._pipe, _lib.ctypes.byref(ent[2]), None,
                                              torch.cuda.current_stream().cuda_stream)
    assert rc == -1


def test_training_reduces_loss_like_reference_loop():
    """Overfit 3 fixed batches with the fused Trainer; mean loss must drop (the qualitative
    behaviour of /root/reference/results/*.png), and eval mode must be deterministic."""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["MUTAG"]
    batches = [b.to("cuda") for b in synth.make_batches("MUTAG", 150, 50, start=2000)]
    m = make_model(sh.num_features, sh.num_classes)
    tr = Trainer(m)
    first, _ = tr.train_epoch(batches, 150)
    for _ in range(60):
        last, acc = tr.train_epoch(batches, 150)
    assert last < first - 0.05, (first, last)
    l1, a1 = tr.test_epoch(batches, 150)
    l2, a2 = tr.test_epoch(batches, 150)
    assert l1 == l2 and a1 == a2 and 0.0 <= a1 <= 100.0


def test_cpu_oracle_training_step_matches_gpu_grads_fp32():
    """fp32 op-sequence oracle (the CPU baseline that bench.py times) vs GPU grads, eval-mode
    dropout-free comparison so no mask plumbing is involved."""
    sh = synth.SHAPES["MUTAG"]
    b_cpu = synth.make_batch("MUTAG", 30, start=900)
    m = make_model(sh.num_features, sh.num_classes).eval()
    sd = cpu_state_dict(m)
    ref = ref_ops.RefModel(sh.num_features, sh.num_classes); ref.load_state_dict(sd); ref.eval(); ref.stable_sort = True
    b = b_cpu.to("cuda")
    lp = m(b); torch.nn.functional.nll_loss(lp, b.y).backward()
    lr = ref(b_cpu); torch.nn.functional.nll_loss(lr, b_cpu.y).backward()
    assert float((lp.detach().cpu() - lr.detach()).abs().max()) <= LOGIT_TOL
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), rtol=2e-3,
                                   atol=2e-5 * float(q.grad.abs().max()) + 1e-9, err_msg=k)


def _dropin_loop(model, opt, batches, steps):
    """the reference's loop body, train.py:36-45 (without the .item() bookkeeping)"""
    crit = torch.nn.NLLLoss()
    for it in range(steps):
        data = batches[it % len(batches)]
        pred = model(data)
        loss = crit(pred, data.y)
        loss.backward()
        opt.step(); opt.zero_grad()


def test_dropin_optimizer_takes_the_flat_route_and_matches_torch_adam():
    """``from dgcnn_amd.optim import Adam`` in place of ``from torch.optim import Adam`` (train.py:11): loss.backward()
    leaves the 16 gradients as views of one flat buffer, the optimizer recognises it and updates with one kernel; the
    trajectory equals torch.optim.Adam's, and the state_dict has torch's per-parameter layout."""
    from dgcnn_amd.optim import Adam as FlatAdam
    sh = synth.SHAPES["PROTEINS"]
    batches = [b.to("cuda") for b in synth.make_batches("PROTEINS", 30, 10, start=7)]
    res = []
    for kind in ("torch", "flat"):
        m = make_model(sh.num_features, sh.num_classes)
        m.train(); m._seed_base, m._fwd_count = 11, 0
        opt = torch.optim.Adam(m.parameters()) if kind == "torch" else FlatAdam(m.parameters())
        _dropin_loop(m, opt, batches, 6)
        if kind == "flat":
            # gradients of the last backward: views of ONE buffer, not copies
            pred = m(batches[0]); torch.nn.NLLLoss()(pred, batches[0].y).backward()
            base = m._last_flat_grad.untyped_storage().data_ptr()
            assert all(p.grad.untyped_storage().data_ptr() == base for p in m.parameters())
            assert len(opt._flat) == 1                       # fused route engaged
            sd = opt.state_dict()
            assert len(sd["state"]) == 16 and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
            assert float(sd["state"][3]["step"]) == 6.0
            # state round trip into torch's own Adam
            t = torch.optim.Adam(m.parameters()); t.load_state_dict(sd)
            opt.zero_grad()
        torch.cuda.synchronize()
        res.append(m.flat_params.clone())
    # same gradients bit for bit; the one-kernel update rounds differently from torch's multi-tensor formulation
    # (<= ~1e-6 absolute after 6 steps of size 1e-3)
    torch.testing.assert_close(res[0], res[1], rtol=1e-4, atol=5e-6)


def test_dropin_optimizer_resume_from_loaded_state_continues_the_same_trajectory():
    """load_state_dict on an optimizer that has already stepped (flat layout cached) must continue from the LOADED
    moments, not from the ones it had before: 3 steps, checkpoint, 3 more == checkpoint re-loaded into
    the same optimizer object after 2 stray steps, then the same 3."""
    import copy
    from dgcnn_amd.optim import Adam as FlatAdam
    sh = synth.SHAPES["PROTEINS"]
    batches = [b.to("cuda") for b in synth.make_batches("PROTEINS", 30, 10, start=7)]
    m = make_model(sh.num_features, sh.num_classes)
    m.train(); m._seed_base, m._fwd_count = 11, 0
    opt = FlatAdam(m.parameters())
    _dropin_loop(m, opt, batches, 3)
    torch.cuda.synchronize()
    ck_model = copy.deepcopy(m.state_dict()); ck_opt = copy.deepcopy(opt.state_dict()); ck_cnt = m._fwd_count
    _dropin_loop(m, opt, batches, 3)
    torch.cuda.synchronize()
    want = m.flat_params.clone()
    _dropin_loop(m, opt, batches, 2)               # stray steps: moments and counter move on
    m.load_state_dict(ck_model); opt.load_state_dict(ck_opt); m._fwd_count = ck_cnt
    _dropin_loop(m, opt, batches, 3)
    torch.cuda.synchronize()
    assert torch.equal(m.flat_params, want)
    assert float(opt.state_dict()["state"][0]["step"]) == 6.0


def test_dropin_optimizer_fallback_after_the_flat_route_keeps_adams_step_count():
    """a step whose gradients do NOT mirror the flat layout (here: one ``p.grad`` replaced by a copy, as a user-side gradient
    transform would) after steps that did: the fallback to torch's Adam must count ONE step (the flat route keeps one shared
    counter for the 16 parameters; handed to torch's multi-tensor Adam as it is, it was incremented 16 times), and later steps
    return to the flat route from the same moments"""
    from dgcnn_amd.optim import Adam as FlatAdam
    sh = synth.SHAPES["PROTEINS"]
    batches = [b.to("cuda") for b in synth.make_batches("PROTEINS", 30, 10, start=7)]
    res = []
    for kind in ("torch", "flat"):
        m = make_model(sh.num_features, sh.num_classes)
        m.train(); m._seed_base, m._fwd_count = 11, 0
        opt = torch.optim.Adam(m.parameters()) if kind == "torch" else FlatAdam(m.parameters())
        crit = torch.nn.NLLLoss()
        for it in range(6):
            data = batches[it % len(batches)]
            crit(m(data), data.y).backward()
            if it == 2:
                m.classifier_2.bias.grad = m.classifier_2.bias.grad.clone()       # no longer a view of the flat gradient buffer
            opt.step(); opt.zero_grad()
            if kind == "flat":
                assert len(opt._flat) == (0 if it == 2 else 1), it
        torch.cuda.synchronize()
        if kind == "flat":
            assert [float(v["step"]) for v in opt.state_dict()["state"].values()] == [6.0] * 16
        res.append(m.flat_params.clone())
    torch.testing.assert_close(res[0], res[1], rtol=1e-4, atol=5e-6)


def test_dropin_optimizer_falls_back_for_foreign_parameters():
    from dgcnn_amd.optim import Adam as FlatAdam
    lin = torch.nn.Linear(5, 3).cuda()
    ref = torch.nn.Linear(5, 3).cuda(); ref.load_state_dict(lin.state_dict())
    o1, o2 = FlatAdam(lin.parameters()), torch.optim.Adam(ref.parameters())
    x = torch.randn(7, 5, device="cuda")
    for _ in range(3):
        for mod, o in ((lin, o1), (ref, o2)):
            mod(x).square().sum().backward(); o.step(); o.zero_grad()
    torch.testing.assert_close(lin.weight, ref.weight); torch.testing.assert_close(lin.bias, ref.bias)
    assert len(o1._flat) == 0


def test_end_to_end_training_learns_a_structural_task_and_generalises():
    """Forward, SortPooling, backward, Adam and the epoch bookkeeping must ALL be right for this to work: graphs whose
    class sets their edge density (dgcnn_amd.synth labels="structure"), trained with the reference's recipe (batch 50,
    Adam 1e-3, dropout 0.5) -- held-out accuracy must end far above chance (the qualitative content of
    /root/reference/results/*.png)."""
    from dgcnn_amd.train import Trainer
    from dgcnn_amd.tudataset import GraphLoader
    sh = synth.SHAPES["PROTEINS"]
    graphs = synth.make_graphs("PROTEINS", 500, start=123, labels="structure")
    train, test = graphs[:400], graphs[400:]
    m = make_model(sh.num_features, sh.num_classes)
    tr = Trainer(m)
    gen = torch.Generator().manual_seed(1)
    tl = GraphLoader(train, 50, shuffle=True, generator=gen, device="cuda")
    vl = GraphLoader(test, 50, device="cuda")
    first = None
    for epoch in range(25):
        loss, acc = tr.train_epoch(tl, len(train))
        first = loss if first is None else first
    vloss, vacc = tr.test_epoch(vl, len(test))
    assert loss < 0.6 * first, (first, loss)
    assert acc > 85.0 and vacc > 85.0, (acc, vacc)


def test_eval_step_metrics_match_the_reference_bookkeeping():
    """dgcnn_model_eval_step: forward in eval mode + metrics[0] += NLLLoss-mean, metrics[1] += #correct (train.py:59-64)."""
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["PROTEINS"]
    batches = [b.to("cuda") for b in synth.make_batches("PROTEINS", 130, 50, start=3)]      # 50, 50, 30 graphs
    m = make_model(sh.num_features, sh.num_classes)
    m.train()
    tr = Trainer(m)
    tr.reset_metrics()
    want_loss, want_correct = 0.0, 0
    for b in batches:
        lp = tr.eval_step(b, b.y).clone()
        assert m.training                        # eval_step leaves the module's mode alone
        lp2 = tr.eval_step(b, b.y).clone()       # eval mode: no dropout, deterministic
        assert torch.equal(lp, lp2)
        want_loss += 2 * float(torch.nn.functional.nll_loss(lp, b.y))
        want_correct += 2 * int((lp.argmax(dim=1) == b.y).sum())
    loss, correct = tr.read_metrics()
    assert abs(loss - want_loss) < 1e-4 * max(1.0, abs(want_loss)) and correct == want_correct


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_general_edge_lists_match_the_oracle(seed):
    """Arbitrary edge lists as a user might hand them over -- directed, unsorted, with duplicates, self loops, isolated
    nodes, graphs of 1..40 nodes (some below k = 30), several feature widths -- through the general graph-prep path:
    forward and gradients within the parity bar.  (The oracle implements PyG's semantics: self loops removed, duplicate
    edges counted twice.)"""
    from dgcnn_amd.batch import Batch
    g = torch.Generator().manual_seed(1000 + seed)
    B = int(torch.randint(1, 9, (1,), generator=g))
    F = [1, 3, 8, 17, 33, 6][seed]
    sizes = torch.randint(1, 41, (B,), generator=g).tolist()
    xs, eis, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes):
        m_edges = int(torch.randint(0, 4 * n + 1, (1,), generator=g))
        src = torch.randint(0, n, (m_edges,), generator=g)
        dst = torch.randint(0, n, (m_edges,), generator=g)       # self loops and duplicates happen naturally
        eis.append(torch.stack([src, dst]) + off)
        xs.append(torch.randn(n, F, generator=g))
        bs.append(torch.full((n,), gi, dtype=torch.int64))
        off += n
    ei = torch.cat(eis, 1)
    ei = ei[:, torch.randperm(ei.shape[1], generator=g)]          # globally shuffled: nothing is sorted
    b = Batch(torch.cat(xs), ei, torch.cat(bs), torch.randint(0, 3, (B,), generator=g), B)
    m = make_model(F, 3)
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


def test_two_stage_weight_gradient_reduction_in_a_subprocess():
    """Batches above DG_WG_TWO_STAGE_B (1024) graphs reduce the per-graph weight-gradient partials in two stages (chunk
    partials + split-K classifier_1 GEMM, then the final sums).  A batch that large is too slow for the CPU oracle, so a
    child process lowers the threshold through the environment and runs the usual parity check on 300 graphs."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from dgcnn_amd import synth\n"
        "from parity_util import check_backward_parity, check_forward_parity, cpu_state_dict, make_model\n"
        "sh = synth.SHAPES['MUTAG']; b = synth.make_batch('MUTAG', 300, start=41)\n"
        "m = make_model(sh.num_features, sh.num_classes); sd = cpu_state_dict(m)\n"
        "check_forward_parity(m, b, sd); check_backward_parity(m, b, sd); print('TWO_STAGE_OK')\n"
    ) % (root, os.path.join(root, "tests"))
    env = dict(os.environ, DG_WG_TWO_STAGE_B="128")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "TWO_STAGE_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def test_trainer_uses_tensors_rebound_on_a_cached_batch_object():
    """the Trainer caches the argument block of a batch object (pointers included); a caller that assigns NEW tensors to
    ``batch.x`` / ``batch.edge_index`` on the same object (PyG-style transforms do) must get the new values, on the
    current batch and on the look-ahead batch alike"""
    from dgcnn_amd.batch import Batch
    from dgcnn_amd.train import Trainer
    sh = synth.SHAPES["PROTEINS"]
    b1 = synth.make_batch("PROTEINS", 8, start=11).to("cuda")
    b2 = synth.make_batch("PROTEINS", 8, start=40).to("cuda")
    x_new = (b1.x * 0.5 + 0.25).contiguous()
    res = []
    for rebind in (True, False):
        m = make_model(sh.num_features, sh.num_classes)
        m.train(); m._seed_base, m._fwd_count = 9, 0
        tr = Trainer(m)
        a = Batch(b1.x, b1.edge_index, b1.batch, b1.y, b1.num_graphs, b1.coalesced_undirected, b1.max_nodes, b1.max_edges)
        tr.train_step(a, a.y, next_data=b2)
        tr.train_step(b2, b2.y, next_data=a)            # `a` is cached twice over: as a current and as a look-ahead batch
        if rebind:
            a.x = x_new                                  # same object, new tensor
            nxt = a
        else:
            nxt = Batch(x_new, b1.edge_index, b1.batch, b1.y, b1.num_graphs, b1.coalesced_undirected, b1.max_nodes, b1.max_edges)
        tr.train_step(nxt, nxt.y)
        torch.cuda.synchronize()
        tr.read_metrics()
        res.append(m.flat_params.clone())
    assert torch.equal(res[0], res[1])
