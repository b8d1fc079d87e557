"""Batched classifier (classifier.hip): from 257 graphs per launch classifier_1 / classifier_2 (reference model.py:21-23,
41-45) and their backward leave the per-graph readout kernels and run as GEMMs over 16 graphs per workgroup.  Same oracle
parity bar as the per-graph form (tests/parity_util.py); same dropout mask function; same loss / accuracy bookkeeping;
results of a graph do not depend on which form ran beyond fp32 summation order."""
import numpy as np
import pytest
import torch

from dgcnn_amd import synth
from dgcnn_amd.batch import Batch
from oracle import ref_dense
from parity_util import check_backward_parity, check_forward_parity, cpu_state_dict, grads_close, make_model

pytestmark = pytest.mark.gpu

KEYS = ["conv1.lin.weight", "conv1.bias", "conv2.lin.weight", "conv2.bias", "conv3.lin.weight", "conv3.bias",
        "conv4.lin.weight", "conv4.bias", "conv5.weight", "conv5.bias", "conv6.weight", "conv6.bias",
        "classifier_1.weight", "classifier_1.bias", "classifier_2.weight", "classifier_2.bias"]


def _batch(name, bs, C):
    b = synth.make_batch(name, bs, start=300)
    if C != synth.SHAPES[name].num_classes:
        y = (torch.arange(bs) * 7 + 3) % C
        b = Batch(b.x, b.edge_index, b.batch, y, b.num_graphs, b.coalesced_undirected, b.max_nodes, b.max_edges)
    return b


CASES = [("MUTAG", 257, 2), ("MUTAG", 272, 2), ("MUTAG", 540, 2), ("PROTEINS", 263, 2), ("MUTAG", 300, 11), ("MUTAG", 259, 64)]


@pytest.mark.parametrize("name,bs,C", CASES, ids=[f"{c[0]}-{c[1]}-C{c[2]}" for c in CASES])
def test_batched_classifier_forward_and_dropin_backward_vs_oracle(name, bs, C):
    """the drop-in route: forward through the batched classifier (no labels: forward only), autograd backward through the
    per-graph readout backward on the activations the batched form stored"""
    b = _batch(name, bs, C)
    m = make_model(synth.SHAPES[name].num_features, C)
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


STEP_CASES = [c + (False,) for c in CASES] + [("MUTAG", 300, 2, True), ("PROTEINS", 263, 2, True), ("COLLAB", 270, 3, True)]


@pytest.mark.parametrize("name,bs,C,chain", STEP_CASES, ids=[f"{c[0]}-{c[1]}-C{c[2]}{'-chain' if c[3] else ''}" for c in STEP_CASES])
def test_batched_classifier_training_step_vs_oracle(name, bs, C, chain):
    """the fused training step (labels in the kernel): batched classifier forward + backward, readout backward from its
    gz6 -- loss, accuracy and every gradient against the fp64 oracle on the kernel's own dropout mask and permutation.
    `chain`: the dense graph-chain kernels forced (the library takes them from ~28 k nodes on), whose large-batch backward
    reads SPARSE SortPooling-gradient slabs (rows of the selected nodes + a flag per node)"""
    from dgcnn_amd.train import Trainer
    b_cpu = _batch(name, bs, C)
    if chain:
        start = 300
        while b_cpu.max_nodes > 256:      # (the chain backward admits graphs of up to 256 nodes)
            start += bs
            b_cpu = synth.make_batch(name, bs, start=start)
    b = b_cpu.to("cuda")
    m = make_model(synth.SHAPES[name].num_features, C)
    sd = cpu_state_dict(m)
    if chain:
        m.agg_mode, m.use_chain = "dense", True
    m.train(); m._seed_base, m._fwd_count = 7, 0
    tr = Trainer(m)
    tr.reset_metrics()
    tr.train_step(b, b.y)
    torch.cuda.synchronize()
    m.check_errors()
    lsum, correct = tr.read_metrics()
    mask = m.last_workspace_view("drop_mask").cpu(); perm = m.last_workspace_view("perm").cpu()
    frac = float(mask.float().mean())
    assert 0.4 < frac < 0.6, frac
    logp_ref, loss_ref, g_ref, _ = ref_dense.loss_and_grads_dense(sd, b_cpu.x, b_cpu.edge_index, b_cpu.batch, b_cpu.y,
                                                                  b_cpu.num_graphs, dropout_mask=mask, perm_override=perm)
    assert abs(lsum - float(loss_ref)) < 1e-5
    # accuracy: count of first-max predictions equal to the label, where the oracle's margin leaves no doubt
    top2 = logp_ref.detach().topk(min(2, C), dim=1).values
    sure = (top2[:, 0] - top2[:, -1]) > 1e-4 if C > 1 else torch.ones(bs, dtype=torch.bool)
    want = (logp_ref.detach().argmax(1) == b_cpu.y)
    assert abs(correct - float(want.sum())) <= float((~sure).sum())
    g = tr.grads.cpu()
    for p, off, key in zip(m._param_list(), m._offsets, KEYS):
        good, md, sc = grads_close(g[off:off + p.numel()], g_ref[key].reshape(-1))
        assert good, f"grad {key}: max diff {md:.3e} at scale {sc:.3e}"


def test_batched_and_per_graph_forms_agree_on_the_same_graphs():
    """graph g of a 300-graph batch (batched form) and of its first 256 graphs (per-graph form): identical dropout mask,
    log-probabilities within fp32 summation order, identical SortPooling permutation"""
    from dgcnn_amd.batch import collate
    sh = synth.SHAPES["MUTAG"]
    graphs = synth.make_graphs("MUTAG", 300, start=40)
    m = make_model(sh.num_features, sh.num_classes)
    m.train(); m._seed_base = 5
    out = []
    for k in (300, 256):
        m._fwd_count = 0
        bb = collate(graphs[:k]).to("cuda")
        lp = m(bb).detach().clone()
        out.append((lp, m.last_workspace_view("drop_mask").clone(), m.last_workspace_view("perm").clone(),
                    m.last_workspace_view("a1d").clone()))
    (lpa, ma, pa, aa), (lpb, mb, pb, ab) = out
    assert torch.equal(ma[:256], mb) and torch.equal(pa[:256], pb)
    assert float((lpa[:256] - lpb).abs().max()) <= 2e-5
    assert float((aa[:256] - ab).abs().max()) <= 2e-5


def test_batched_classifier_bad_label_poisons_the_loss_and_is_reported():
    from dgcnn_amd.train import Trainer
    b_cpu = _batch("MUTAG", 260, 2)
    y = b_cpu.y.clone(); y[137] = 5
    b = b_cpu.to("cuda")
    m = make_model(synth.SHAPES["MUTAG"].num_features, 2)
    m.train()
    tr = Trainer(m)
    tr.reset_metrics()
    tr.train_step(b, y.to("cuda"))
    torch.cuda.synchronize()
    with pytest.raises(Exception):
        tr.read_metrics()


def test_walking_readout_backward_and_column_window_sums_in_a_subprocess():
    """Above DG_WG_TWO_STAGE_B (1024) graphs the readout backward of a fused training step WALKS four graphs per workgroup and
    leaves one conv5 / conv6 partial row per workgroup (k_tail_bwd_walk), and k_wgrad sums the partial rows in two column
    windows.  A child process lowers the threshold through the environment and runs this file's fused-step parity cases
    (257..540 graphs, batch sizes that are and are not multiples of four, the forced chain cases with sparse slabs)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DG_WG_TWO_STAGE_B="128")
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.abspath(__file__), "-k",
                          "training_step_vs_oracle or bad_label"], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0 and " passed" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
