"""Device code on the CPU SIMT emulation (emu/, tests/emu_util.py) -- part of the `-m "not gpu"` suite, small sizes.

The kernel sources are compiled as plain C++ against a stand-in for the HIP headers and run lane by lane; wave-level instructions
and workgroup barriers are resolved by a scheduler.  What this checks is the ARITHMETIC AND INDEXING of the kernels against the
fp64 oracle; what it cannot check is timing, memory ordering and races.  Two kinds of cases:

* calibration -- kernels with green GPU records from rounds 1-5 (the one-launch evaluation kernel, the one-launch training kernel,
  the launch-per-layer gather route, graph preparation): they must reproduce the oracle here too, which pins the emulation's
  reading of the matrix instructions, the DPP controls and the LDS transpose read;
* round 6 -- device code that had never run on a GPU when it was committed (the pool was closed): the 512-node evaluation kernel,
  the persistent chain kernels with their loop-end barrier, the bitmap check of the two-launch chain route.

The full GPU suite runs on the emulation with `DGCNN_EMU=1 python -m pytest tests -m gpu` (hours of CPU; profiles/r06_emu_gputest.txt)."""
import numpy as np
import pytest
import torch

from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import Batch
from dgcnn_amd.train import Trainer
from emu_util import emulated, read_metrics
from oracle import ref_dense
from parity_util import (check_backward_parity, check_forward_parity, cpu_state_dict, gpu_xcat, grads_close, make_model)
from test_gpu_dense import _sized_batch

KEYS = ["conv1.lin.weight", "conv1.bias", "conv2.lin.weight", "conv2.bias", "conv3.lin.weight", "conv3.bias", "conv4.lin.weight",
        "conv4.bias", "conv5.weight", "conv5.bias", "conv6.weight", "conv6.bias", "classifier_1.weight", "classifier_1.bias",
        "classifier_2.weight", "classifier_2.bias"]


def form_of(L, m, b, extra=0):
    fl = m._mode_flags() | (_lib.FLAG_COALESCED_UNDIRECTED if b.coalesced_undirected else 0) | extra
    return L.dgcnn_forward_form(b.num_nodes, b.num_edges, b.num_graphs, int(b.x.shape[1]), fl, int(b.max_nodes or 0))


@pytest.fixture(scope="module")
def emu():
    with emulated() as L:
        yield L


# ---- calibration: kernels with green GPU records -------------------------------------------------------------------------------
@pytest.mark.parametrize("name,bs", [("MUTAG", 6), ("COLLAB", 3), ("PROTEINS", 5)])
def test_emu_calibration_one_launch_eval_kernel(emu, name, bs):
    sh = synth.SHAPES[name]
    b = synth.make_batch(name, bs, start=10)
    while b.max_nodes > 256:
        b = synth.make_batch(name, bs, start=b.num_nodes)
    m = make_model(sh.num_features, sh.num_classes, device="cpu")
    assert form_of(emu, m, b) & _lib.FORM_EVAL
    check_forward_parity(m, b, cpu_state_dict(m))


def _step_grads_vs_oracle(m, b):
    sd = cpu_state_dict(m)
    m.train(); m._seed_base, m._fwd_count = 11, 0
    tr = Trainer(m)
    tr.reset_metrics()
    tr.train_step(b, b.y)
    lsum, _ = read_metrics(tr)
    perm, mask = m.last_workspace_view("perm"), m.last_workspace_view("drop_mask")
    logp_ref, loss_ref, g_ref, aux = ref_dense.loss_and_grads_dense(sd, b.x, b.edge_index, b.batch, b.y, b.num_graphs,
                                                                    dropout_mask=mask, perm_override=perm)
    assert float((gpu_xcat(m).double() - aux["xcat"].detach()).abs().max()) <= 2e-5
    assert abs(lsum - float(loss_ref)) < 1e-5
    for p, off, key in zip(m._param_list(), m._offsets, KEYS):
        good, md, sc = grads_close(tr.grads[off:off + p.numel()], g_ref[key].reshape(-1))
        assert good, f"grad {key}: max diff {md:.3e} at scale {sc:.3e}"


@pytest.mark.parametrize("sizes,F", [([20, 17, 33], 3), ([130, 9], 12)])
def test_emu_calibration_one_launch_training_kernel_and_wgrad(emu, sizes, F):
    """k_chain_readout_tail (chain forward + readout + readout backward + the whole GCN backward of a graph) + k_wgrad (+ Adam)"""
    b = _sized_batch(sizes, F=F, seed=3)
    m = make_model(F, 2, device="cpu")
    assert form_of(emu, m, b) & _lib.FORM_STEP
    _step_grads_vs_oracle(m, b)


def test_emu_calibration_launch_per_layer_gather_route(emu):
    """general edge list (no layout promise: duplicates, a self loop, a directed edge): general graph preparation, the wave-per-node
    gather kernels forward and backward, k_readout_tail / k_tail_bwd, k_wgrad"""
    g = torch.Generator().manual_seed(5)
    n = 40                                                   # graph 0: nodes 0..24, graph 1: nodes 25..39
    s0, d0 = torch.randint(0, 25, (100,), generator=g), torch.randint(0, 25, (100,), generator=g)
    s1, d1 = torch.randint(25, 40, (60,), generator=g), torch.randint(25, 40, (60,), generator=g)
    src, dst = torch.cat([s0, s1]), torch.cat([d0, d1])
    ei = torch.stack([torch.cat([src, dst, torch.tensor([3, 3, 30])]), torch.cat([dst, src, torch.tensor([3, 9, 31])])])
    b = Batch(torch.randn(n, 4, generator=g), ei, torch.cat([torch.zeros(25), torch.ones(15)]).long(), torch.tensor([0, 1]), 2)
    m = make_model(4, 2, device="cpu")
    sd = cpu_state_dict(m)
    assert form_of(emu, m, b) == 0
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


# ---- round 6: device code that had not run on a GPU when it was committed -------------------------------------------------------
@pytest.mark.parametrize("sizes,F", [([257, 40], 3), ([300, 512, 5], 20)])
def test_emu_round6_wide_eval_kernel(emu, sizes, F):
    """k_chain_readout_eval<.., MAXN = 512> (DGCNN_FLAG_INFERENCE) vs the fp64 oracle and vs the launch-per-layer route"""
    b = _sized_batch(sizes, F=F, seed=sum(sizes))
    m = make_model(F, 2, device="cpu")
    sd = cpu_state_dict(m)
    assert not form_of(emu, m, b) & _lib.FORM_EVAL
    m.inference_one_launch = True
    assert form_of(emu, m, b, _lib.FLAG_INFERENCE) & _lib.FORM_EVAL
    logp, _, _, _ = check_forward_parity(m, b, sd)
    xw = gpu_xcat(m)
    m.inference_one_launch = False
    logp2, _, _, _ = check_forward_parity(m, b, sd)
    assert float((xw - gpu_xcat(m)).abs().max()) <= 4e-6 and float((logp - logp2).abs().max()) <= 1e-4
    # ... and through Trainer.eval_step with the metrics folded in by the launch's last workgroup
    m.inference_one_launch = True
    tr = Trainer(m)
    tr.reset_metrics()
    lp = tr.eval_step(b, b.y).clone()
    loss, correct = read_metrics(tr)
    assert torch.equal(lp, logp)
    want_loss = float(-(logp[torch.arange(b.num_graphs), b.y]).sum() / b.num_graphs)
    assert abs(loss - want_loss) <= 1e-5 and correct == float((logp.argmax(1) == b.y).sum())


def test_emu_round6_persistent_chain_kernels_walk_several_graphs(emu):
    """k_chain_fwd_q<8, .., LOOP> and k_chain_bwd_a / _b<8, LOOP> with more graphs than persistent workgroups (every workgroup
    walks >= 2 graphs, loop-end barriers executed), graphs of 129..150 nodes among tiny ones: forward and every gradient vs fp64"""
    rng = np.random.default_rng(2)
    sizes = [int(v) for v in rng.integers(3, 9, size=524)]
    for k in (0, 7, 300, 523):
        sizes[k] = int(rng.integers(129, 151))
    b = _sized_batch(sizes, seed=8)
    m = make_model(3, 2, device="cpu")
    sd = cpu_state_dict(m)
    m.agg_mode, m.use_chain = "dense", True
    f = form_of(emu, m, b)
    assert f & _lib.FORM_CHAIN and f & _lib.FORM_DENSE and b.num_graphs > 512
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


def test_emu_round6_two_launch_chain_route_checks_the_bitmap_itself(emu):
    """100 graphs prepared with the reverse-edge check left to a one-launch kernel, consumed by the two-launch chain route
    (dgcnn_eval_kernel_enable(0)): one missing reverse edge is flagged, a clean batch is not"""
    sizes = [6] * 99 + [9]
    good = _sized_batch(sizes, seed=4)
    ei = good.edge_index
    e = int(ei.shape[1] * 0.6)
    sn, dn = int(ei[0, e]), int(ei[1, e])
    bad = Batch(good.x, ei[:, ~((ei[0] == dn) & (ei[1] == sn))].contiguous(), good.batch, good.y, good.num_graphs, True,
                good.max_nodes, good.max_edges)
    prev = emu.dgcnn_eval_kernel_enable(0)
    try:
        for b, flagged in ((good, False), (bad, True)):
            m = make_model(3, 2, device="cpu")
            m.eval()
            with torch.no_grad():
                m(b)
            if flagged:
                with pytest.raises(_lib.DgcnnError):
                    m.check_errors()
            else:
                m.check_errors()
    finally:
        emu.dgcnn_eval_kernel_enable(prev)


# ---- the GPU tests themselves, a quick subset, on the emulation -------------------------------------------------------------------
QUICK_GPU_TESTS = [
    "tests/test_gpu_kernels.py",                                                   # every C entry point vs the oracle (85 cases)
    "tests/test_gpu_model.py::test_golden_fixture_forward",                        # tests/golden/*.npz through Model.forward
    "tests/test_gpu_model.py::test_golden_fixture_gradients",
    "tests/test_gpu_model.py::test_golden_step_fixture_loss_grads_and_post_adam_parameters",
    "tests/test_gpu_eval_kernel.py::test_golden_fixtures_through_the_eval_kernel",
    "tests/test_gpu_chain_tail.py::test_golden_step_fixture_through_the_one_launch_kernel",
    "tests/test_gpu_chain_tail.py::test_golden_fixture_training_mode_through_the_one_launch_kernel",
    "tests/test_gpu_chain.py::test_chain_boundary_sizes_isolated_nodes_and_single_node_graphs",
    "tests/test_gpu_chain.py::test_chain_raw_feature_widths",
    "tests/test_gpu_dense.py::test_dense_boundary_sizes_isolated_nodes_and_single_node_graphs",
    "tests/test_tudataset.py",
]


def test_emu_quick_subset_of_the_gpu_tests_in_a_subprocess():
    """`DGCNN_EMU=1 python -m pytest -m gpu <quick subset>`: the GPU tests run VERBATIM on the emulation library (tests/conftest.py
    maps "cuda" to the CPU for that process) -- the golden fixtures through the forward, the evaluation kernel and the one-launch
    training kernel, every C entry point, the chain / dense kernels at their size-class boundaries, the TU reader's device path.
    A subprocess because the device mapping is process-wide and must not leak into the tests that check the product REFUSES CPU tensors."""
    import os
    import re
    import subprocess
    import sys
    from emu_util import ROOT, build_emu
    build_emu()
    env = dict(os.environ, DGCNN_EMU="1")
    env.pop("DGCNN_HIP_LIB", None)
    nw = max(1, min(4, os.cpu_count() or 1))
    res = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-n", str(nw), "-p", "no:cacheprovider"] + QUICK_GPU_TESTS,
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = res.stdout[-3000:]
    m = re.search(r"(\d+) passed", tail)
    assert res.returncode == 0 and m and int(m.group(1)) >= 120 and "failed" not in tail.split("\n")[-2], tail


# ---- LDS race detector (emu_rt.cpp; `make -C emu race`) ---------------------------------------------------------------------------
def test_emu_race_detector_finds_no_cross_wave_lds_race_in_the_default_build():
    """The emulation library built with load / store instrumentation keeps, per LDS word, the wave that wrote it and the waves that
    read it in the current barrier epoch; two different waves on one word in one epoch, one of them writing, is a race on the
    hardware whatever the timing.  Run in a subprocess (its own library): the persistent chain kernels walking several graphs per
    workgroup (the loop-end barrier's case), the one-launch training kernel + k_wgrad, the evaluation kernel, the
    gather route -- parity AND zero race events.  (The same batch shapes on a build WITHOUT the loop-end barriers: 67 k / 177 k
    events and wrong results: profiles/r06_race_test.txt.)"""
    import os
    import subprocess
    import sys
    from emu_util import EMU_DIR, ROOT
    res = subprocess.run(["make", "-C", EMU_DIR, "race", "-j8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lib = os.path.join(EMU_DIR, "libdgcnn_emu_race.so")
    assert res.returncode == 0 and os.path.exists(lib), res.stdout[-3000:]
    code = r'''
import sys, ctypes
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from emu_util import emulated
from dgcnn_amd.train import Trainer
from parity_util import make_model, cpu_state_dict, check_forward_parity, check_backward_parity
from test_gpu_dense import _sized_batch
with emulated() as L:
    L.dg_emu_race_count.restype = ctypes.c_ulonglong
    b = _sized_batch([150, 140, 133] + [6] * 300 + [129, 9] * 3, seed=8)        # more graphs than persistent workgroups: walks of 2..3
    m = make_model(3, 2, device="cpu"); sd = cpu_state_dict(m)
    m.agg_mode, m.use_chain = "dense", True
    check_forward_parity(m, b, sd); check_backward_parity(m, b, sd)
    b2 = _sized_batch([130, 40, 9], F=12, seed=3)
    m2 = make_model(12, 2, device="cpu"); sd2 = cpu_state_dict(m2)
    check_forward_parity(m2, b2, sd2)                                         # evaluation kernel
    m2.train(); tr = Trainer(m2); tr.train_step(b2, b2.y); tr.read_metrics()   # step kernel + k_wgrad
    m3 = make_model(12, 2, device="cpu"); sd3 = cpu_state_dict(m3)
    m3.use_chain, m3.agg_mode = False, "sparse"
    check_backward_parity(m3, b2, sd3)                                        # gather route
    print("RACE_EVENTS", L.dg_emu_race_count())
'''
    env = dict(os.environ, DG_EMU_RACE="1", DGCNN_EMU_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=1500)
    assert r.returncode == 0 and "RACE_EVENTS 0" in r.stdout, r.stdout[-3000:]
