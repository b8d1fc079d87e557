"""One-launch evaluation kernel for batches with a graph of 257..512 nodes (round 6).

``k_chain_readout_eval<.., MAXN = 512>`` (gcn_chain.hip): the chain body with two row tiles per wave (the body of
``k_chain_fwd_q<16, .., LOOP = false>``) followed by the readout in the same launch, so that ``test()`` of the reference
(/root/reference/train.py:49-66) stays one launch on PROTEINS-like sets.  Reached with ``DGCNN_FLAG_INFERENCE`` (the model attribute
``inference_one_launch`` under ``torch.no_grad()`` / ``Trainer.eval_step``) or, for chain-form batches, with
``dgcnn_eval_kernel_enable(2)``.  Same protocol as tests/test_gpu_eval_kernel.py: the form is asserted first, then the fp64 oracle
(activations, legality of the selection, log-probabilities <= 1e-4), then agreement with the launch-per-layer route."""
import pytest
import torch

from dgcnn_amd import _lib, synth
from dgcnn_amd.train import Trainer
from parity_util import check_forward_parity, cpu_state_dict, gpu_xcat, make_model
from test_gpu_dense import _sized_batch
from test_gpu_eval_kernel import expected_metrics

pytestmark = pytest.mark.gpu


def form_of(m, b, extra=0):
    fl = m._mode_flags() | (_lib.FLAG_COALESCED_UNDIRECTED if b.coalesced_undirected else 0) | extra
    return _lib.lib().dgcnn_forward_form(b.num_nodes, b.num_edges, b.num_graphs, int(b.x.shape[1]), fl, int(b.max_nodes or 0))


def wide_batch(name, bs, start):
    """first seeded batch of the shape whose largest graph has 257..512 nodes"""
    for k in range(400):
        b = synth.make_batch(name, bs, start=start + k * bs)
        if 256 < b.max_nodes <= 512:
            return b
    raise AssertionError(f"no {name} batch of {bs} graphs with a largest graph of 257..512 nodes")


SIZED = [([257, 300, 511, 512, 5, 130], ()), ([512], ()), ([1, 258, 2, 400, 33], (1,)), ([384] * 7 + [16, 17], ())]


@pytest.mark.parametrize("sizes,isolated", SIZED, ids=["mixed", "one_512", "tiny_and_wide", "many_384"])
@pytest.mark.parametrize("F", [3, 12, 20])
def test_wide_eval_kernel_vs_fp64_oracle_on_sized_graphs(sizes, isolated, F):
    b = _sized_batch(sizes, F=F, seed=sum(sizes) + F, isolated=isolated)
    m = make_model(F, 2)
    sd = cpu_state_dict(m)
    assert not form_of(m, b) & _lib.FORM_EVAL                      # default: launch per layer for such a batch
    m.inference_one_launch = True
    assert form_of(m, b, _lib.FLAG_INFERENCE) & _lib.FORM_EVAL
    logp, perm, err, err_x = check_forward_parity(m, b, sd)        # (eval mode under no_grad: the flag is on)
    xw = gpu_xcat(m)
    m.inference_one_launch = False
    logp2, perm2, _, _ = check_forward_parity(m, b, sd)
    assert float((xw - gpu_xcat(m)).abs().max()) <= 4e-6           # same sums, different order
    assert float((logp - logp2).abs().max()) <= 1e-4


@pytest.mark.parametrize("name,bs", [("PROTEINS", 50), ("PROTEINS", 13), ("PROTEINS", 128)])
def test_wide_eval_kernel_on_workload_batches_with_metrics(name, bs):
    sh = synth.SHAPES[name]
    b_cpu = wide_batch(name, bs, start=7000)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    m.inference_one_launch = True
    assert form_of(m, b_cpu, _lib.FLAG_INFERENCE) & _lib.FORM_EVAL
    logp, perm, err, err_x = check_forward_parity(m, b_cpu, sd)
    # Trainer.eval_step: same kernel + metrics folded in by the launch's last workgroup; with look-ahead preparation of a second
    # wide batch and of a narrow one (the rider forms), and consumed in that order
    b2_cpu = wide_batch(name, bs, start=9000)
    b3_cpu = synth.make_batch(name, bs, start=100)
    tr = Trainer(m)
    tr.reset_metrics()
    b, b2, b3 = b_cpu.to("cuda"), b2_cpu.to("cuda"), b3_cpu.to("cuda")
    lp = tr.eval_step(b, b.y, next_data=b2).cpu()
    assert torch.equal(lp, logp)
    loss, correct = tr.read_metrics()
    el, ec = expected_metrics(logp, b_cpu.y, 1.0 / bs)
    assert abs(loss - el) <= 1e-5 and correct == ec, ((loss, correct), (el, ec))
    lp2 = tr.eval_step(b2, b2.y, next_data=b3).clone()
    lp3 = tr.eval_step(b3, b3.y).clone()
    tr.read_metrics()
    m2 = make_model(sh.num_features, sh.num_classes)
    m2.inference_one_launch = True
    m2.eval()
    with torch.no_grad():
        assert torch.equal(m2(b2), lp2)
        assert torch.equal(m2(b3), lp3)
    m2.check_errors()


def test_wide_eval_kernel_through_the_switch_for_chain_form_batches_and_a_training_step_in_between():
    """dgcnn_eval_kernel_enable(2): a chain-form batch (DGCNN_FLAG_CHAIN) with a graph of 257..512 nodes takes the one-launch form
    without the inference flag; a training step of the same batch object keeps the launch-per-layer backward route and the
    look-ahead machinery prepares again where the kind of step changes"""
    L = _lib.lib()
    sh = synth.SHAPES["PROTEINS"]
    b_cpu = wide_batch("PROTEINS", 24, start=2000)
    m = make_model(sh.num_features, sh.num_classes)
    sd = cpu_state_dict(m)
    m.use_chain = True
    prev = L.dgcnn_eval_kernel_enable(1)
    try:
        assert not form_of(m, b_cpu) & _lib.FORM_EVAL
        logp1, _, _, _ = check_forward_parity(m, b_cpu, sd)
        x1 = gpu_xcat(m)
        L.dgcnn_eval_kernel_enable(2)
        assert form_of(m, b_cpu) & _lib.FORM_EVAL
        logp2, _, _, _ = check_forward_parity(m, b_cpu, sd)
        assert float((x1 - gpu_xcat(m)).abs().max()) <= 4e-6       # (the same chain body either way)
        assert float((logp1 - logp2).abs().max()) <= 2e-6
    finally:
        L.dgcnn_eval_kernel_enable(prev)
    # eval -> train -> eval over the same two batches with look-ahead, inference flag on: trajectories equal the ones without it
    res = []
    for flag in (True, False):
        mm = make_model(sh.num_features, sh.num_classes)
        mm.inference_one_launch = flag
        mm.train(); mm._seed_base, mm._fwd_count = 3, 0
        tr = Trainer(mm)
        tr.reset_metrics()
        a, b = b_cpu.to("cuda"), wide_batch("PROTEINS", 24, start=4000).to("cuda")
        o1 = tr.eval_step(a, a.y, next_data=b).clone()
        tr.train_step(b, b.y, next_data=a)
        o2 = tr.eval_step(a, a.y).clone()
        torch.cuda.synchronize()
        tr.read_metrics()
        res.append((o1, o2, mm.flat_params.clone()))
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-4
    assert float((res[0][2] - res[1][2]).abs().max()) <= 1e-6      # the training step itself is the same route either way
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-4
