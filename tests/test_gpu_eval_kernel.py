"""One-launch evaluation / inference kernel (round 5).

``k_chain_readout_eval`` (gcn_chain.hip; DGCNN_FORM_EVAL) is what ``Model.forward``, ``dgcnn_model_eval_step`` and
``Trainer.eval_step`` run with DEFAULT flags for coalesced-undirected batches of <= 256 graphs of <= 256 nodes each: the
reference's whole ``Model.forward`` (/root/reference/model.py:26-45) as called from ``test()`` (train.py:57-62) -- graph
convolutions, SortPooling, dense tail -- and, with labels, the batch's loss / #correct (train.py:63-64), in ONE launch.
Every case first asserts that this form IS the one the library takes, then compares with the fp64 dense oracle through
the tie-aware protocol of parity_util (activations, legality of the selection, log-probabilities <= 1e-4)."""
import os

import numpy as np
import pytest
import torch

from dgcnn_amd import _lib, synth
from dgcnn_amd.train import Trainer
from parity_util import check_backward_parity, check_forward_parity, cpu_state_dict, load_fixture, make_model

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def form_of(m, b):
    fl = m._mode_flags() | (_lib.FLAG_COALESCED_UNDIRECTED if b.coalesced_undirected else 0)
    return _lib.lib().dgcnn_forward_form(b.num_nodes, b.num_edges, b.num_graphs, int(b.x.shape[1]), fl, int(b.max_nodes or 0))


def small_batch(name, bs, start, limit=256):
    for k in range(64):
        b = synth.make_batch(name, bs, start=start + k * bs)
        if b.max_nodes <= limit:
            return b
    raise AssertionError(f"no {name} batch of {bs} graphs with max_nodes <= {limit}")


def expected_metrics(logp, y, scale):
    """k_eval_metrics' arithmetic (tail.hip), which the kernel's last workgroup reproduces: 256 threads, thread t takes graphs
    t, t + 256, ..., then a binary tree -- all in fp32"""
    B = logp.shape[0]
    sl, sc = np.zeros(256, np.float32), np.zeros(256, np.float32)
    lp = logp.numpy().astype(np.float32)
    am = lp.argmax(1)                 # (first index of the maximum, as the kernel's scan)
    for t in range(256):
        l, c = np.float32(0), np.float32(0)
        for g in range(t, B, 256):
            l = np.float32(l - lp[g, int(y[g])])
            c = np.float32(c + (1.0 if am[g] == int(y[g]) else 0.0))
        sl[t], sc[t] = l, c
    st = 128
    while st >= 1:
        sl[:st] = sl[:st] + sl[st:2 * st]
        sc[:st] = sc[:st] + sc[st:2 * st]
        st //= 2
    return float(np.float32(sl[0] * np.float32(scale))), float(sc[0])


CASES = [("COLLAB", 50), ("MUTAG", 50), ("PROTEINS", 50), ("COLLAB", 256), ("IMDB", 50), ("COLLAB_REAL", 50), ("COLLAB", 1)]


@pytest.mark.parametrize("name,bs", CASES, ids=[f"{c[0]}-{c[1]}" for c in CASES])
def test_one_launch_eval_kernel_vs_fp64_oracle(name, bs):
    sh = synth.SHAPES[name]
    b_cpu = small_batch(name, bs, start=2000)
    m = make_model(sh.num_features, sh.num_classes)
    assert form_of(m, b_cpu) & _lib.FORM_EVAL, form_of(m, b_cpu)
    sd = cpu_state_dict(m)
    logp, perm, err, err_x = check_forward_parity(m, b_cpu, sd)
    # the same batch through Trainer.eval_step: same log-probabilities bit for bit (same kernel, metrics added), and the metrics
    # equal k_eval_metrics' fixed-order sums of those log-probabilities
    tr = Trainer(m)
    tr.reset_metrics()
    b = b_cpu.to("cuda")
    lp2 = tr.eval_step(b, b.y).cpu()
    loss, correct = tr.read_metrics()
    assert torch.equal(lp2, logp)
    el, ec = expected_metrics(logp, b_cpu.y, 1.0 / bs)
    assert loss == el and correct == ec, ((loss, correct), (el, ec))


@pytest.mark.parametrize("fixture", ["mutag_b6", "proteins_b5", "collab_b4"])
def test_golden_fixtures_through_the_eval_kernel(fixture):
    z, sd, grads, b_cpu = load_fixture(GOLDEN, fixture, coalesced_undirected=True)
    m = make_model(int(z["num_features"]), int(z["num_classes"]), sd)
    assert form_of(m, b_cpu) & _lib.FORM_EVAL
    m.eval()
    with torch.no_grad():
        lp = m(b_cpu.to("cuda")).cpu()
    m.check_errors()
    np.testing.assert_array_equal(m.last_workspace_view("perm").cpu().numpy(), z["perm"])
    np.testing.assert_allclose(lp.numpy(), z["logp_eval_f64"], rtol=0, atol=1e-4)


def test_eval_kernel_equals_the_two_launch_route_and_accumulates_over_batches():
    """dgcnn_eval_kernel_enable(0) keeps chain forward + readout forward + k_eval_metrics as launches of their own: same
    selection, log-probabilities to fp32 order noise, and metrics accumulated over several batches (with look-ahead
    preparation riding) equal to the per-batch sums"""
    L = _lib.lib()
    sh = synth.SHAPES["COLLAB"]
    bs = [small_batch("COLLAB", 50, start=3000 + 400 * k).to("cuda") for k in range(5)]
    m = make_model(sh.num_features, sh.num_classes)
    out = {}
    prev = L.dgcnn_eval_kernel_enable(1)
    try:
        for on in (1, 0):
            L.dgcnn_eval_kernel_enable(on)
            assert bool(form_of(m, bs[0]) & _lib.FORM_EVAL) == bool(on)
            tr = Trainer(m)
            tr.reset_metrics()
            lps, perms = [], []
            for k, b in enumerate(bs):
                lps.append(tr.eval_step(b, b.y, next_data=bs[k + 1] if k + 1 < len(bs) else None).clone())
                perms.append(m.last_workspace_view("perm").clone())
            out[on] = (lps, perms, tr.read_metrics())
    finally:
        L.dgcnn_eval_kernel_enable(prev)
    for a, b in zip(out[1][1], out[0][1]):
        assert torch.equal(a, b)
    for a, b in zip(out[1][0], out[0][0]):
        assert float((a - b).abs().max()) <= 2e-6
    # metrics of the one-launch route: the fixed-order sums, batch after batch, into one fp32 accumulator
    acc_l, acc_c = np.float32(0), np.float32(0)
    for lp, b in zip(out[1][0], bs):
        el, ec = expected_metrics(lp.cpu(), b.y.cpu(), 1.0 / 50)
        acc_l, acc_c = np.float32(acc_l + np.float32(el)), np.float32(acc_c + np.float32(ec))
    assert out[1][2] == (float(acc_l), float(acc_c))
    assert abs(out[1][2][0] - out[0][2][0]) <= 1e-5 and out[1][2][1] == out[0][2][1]


def test_training_mode_forward_through_the_eval_kernel_feeds_the_drop_in_backward():
    """Model.forward in TRAINING mode (the reference's loop body, train.py:37-40) takes the same launch -- dropout mask and
    every activation the autograd backward reads are saved as by the two launches it replaces: gradients vs the fp64 oracle"""
    sh = synth.SHAPES["PROTEINS"]
    b_cpu = small_batch("PROTEINS", 50, start=5000)
    m = make_model(sh.num_features, sh.num_classes)
    assert form_of(m, b_cpu) & _lib.FORM_EVAL
    check_backward_parity(m, b_cpu, cpu_state_dict(m))


def test_eval_metrics_with_a_label_out_of_range_poison_the_loss_not_the_memory():
    sh = synth.SHAPES["MUTAG"]
    b_cpu = small_batch("MUTAG", 20, start=100)
    m = make_model(sh.num_features, sh.num_classes)
    tr = Trainer(m)
    b = b_cpu.to("cuda")
    y = b.y.clone(); y[3] = 7
    tr.reset_metrics()
    tr.eval_step(b, y)
    with pytest.raises(_lib.DgcnnError):
        tr.read_metrics()
