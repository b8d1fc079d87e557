"""GPU tests of the kernels the launch-per-layer route takes for sparse batches of many nodes (DD at the reference's batch of
50, BASELINE.json config 4): the eight-lanes-per-node ("narrow") gather forward / backward and the LDS-staged first linear.
Both promise the SAME fp32 operations in the same order as the forms they replace, so they are compared bit for bit with
those (dgcnn_narrow_gather_enable(0); a 4-byte-misaligned input keeps the direct first linear); the whole-model parity of
these batches against the fp64 oracle is tests/test_gpu_model.py::test_workload_forward_and_backward (DD-50, DD-50 with the
5748-node graph), which runs the narrow forms by default.  Replaces /root/reference/model.py:30-33 and their backward."""
import numpy as np
import pytest
import torch

from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import Graph, collate
from parity_util import make_model, cpu_state_dict, check_forward_parity, check_backward_parity

pytestmark = pytest.mark.gpu
DEV = "cuda"
FORM_CHAIN, FORM_DENSE = 2, 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def hub_batch(F=7, seed=5):
    """sparse graphs of every shape the eight-lane gather has to get right: isolated nodes, rows of exactly 8 / 9 / 16 / 17
    neighbours, a 699-neighbour hub (88 trips of one lane group while its seven neighbours idle), directed edges (in-degree
    != out-degree), duplicates, self loops; 6 x 1200 + 700 + 33 nodes, mean degree ~3"""
    g = torch.Generator().manual_seed(seed)
    graphs = []
    for k in range(6):
        n = 1200
        m = 3 * n
        s = torch.randint(0, n, (m,), generator=g); d = torch.randint(0, n, (m,), generator=g)
        e = torch.stack([s, d])
        extra = []
        for node, deg in ((0, 8), (1, 9), (2, 16), (3, 17), (4, 64), (5, 65)):      # exact in-degrees at the trip boundaries
            keep = e[1] != node
            e = e[:, keep]
            src = torch.arange(100, 100 + deg)
            extra.append(torch.stack([src, torch.full((deg,), node)]))
        e = torch.cat([e] + extra + [e[:, :5], torch.tensor([[7, 8], [7, 8]])], 1)  # duplicates + two self loops
        graphs.append(Graph(torch.randn(n, F, generator=g), e.contiguous(), int(k % 2)))
    n = 700
    star = torch.stack([torch.arange(1, n), torch.zeros(n - 1, dtype=torch.int64)])
    star = torch.cat([star, star.flip(0)], 1)
    graphs.append(Graph(torch.randn(n, F, generator=g), star.contiguous(), 1))
    graphs.append(Graph(torch.randn(33, F, generator=g), torch.zeros(2, 0, dtype=torch.int64), 0))   # 33 isolated nodes
    return collate(graphs)


def step_results(b_cpu, F, C, narrow):
    """one training step of a fresh model through Trainer.train_step with the narrow forms on / off"""
    from dgcnn_amd.train import Trainer
    L = _lib.lib()
    prev = L.dgcnn_narrow_gather_enable(narrow)
    try:
        m = make_model(F, C)
        b = b_cpu.to(DEV)
        form = L.dgcnn_forward_form(b.num_nodes, b.num_edges, b.num_graphs, F, 0, b.max_nodes)
        assert form & (FORM_CHAIN | FORM_DENSE) == 0, form            # the launch-per-layer gather route
        m.train(); m._seed_base, m._fwd_count = 5, 0
        tr = Trainer(m)
        tr.reset_metrics()
        tr.train_step(b, b.y)
        torch.cuda.synchronize()
        m.check_errors()
        views = {k: m.last_workspace_view(k).cpu().clone() for k in ("x1", "x2", "x3", "x4", "perm")}
        return tr.read_metrics(), tr.grads.cpu().clone(), m.flat_params.detach().cpu().clone(), views
    finally:
        L.dgcnn_narrow_gather_enable(prev)


CASES = [("DD", 50, None), ("DD", 50, 5748), ("hub7", 0, None), ("hub40", 0, None)]


@pytest.mark.parametrize("name,bs,force", CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}" for c in CASES])
def test_narrow_gather_forms_reproduce_the_wave_per_node_forms_bit_for_bit(name, bs, force):
    """same batch, same weights, same dropout seed through the two kernel families: node activations, selection, loss,
    every gradient element and the parameters after the fused Adam step are IDENTICAL (the narrow forms perform the wide
    forms' fp32 additions in the same order; the partial weight-gradient rows have the same owners)"""
    if name.startswith("hub"):
        F, C = int(name[3:]), 2
        b_cpu = hub_batch(F)
    else:
        sh = synth.SHAPES[name]
        F, C = sh.num_features, sh.num_classes
        b_cpu = synth.make_batch(name, bs, start=3000, force_first_n=force)
    assert b_cpu.num_nodes > 4096 and b_cpu.num_edges <= 8 * b_cpu.num_nodes          # the admission rule of the narrow forms
    (ma, ga, wa, va), (mb, gb, wb, vb) = step_results(b_cpu, F, C, 1), step_results(b_cpu, F, C, 0)
    for k in va:
        assert torch.equal(va[k], vb[k]), k
    assert ma == mb, (ma, mb)
    assert torch.equal(ga, gb), float((ga - gb).abs().max())
    assert torch.equal(wa, wb)
    assert float(ga.abs().max()) > 0


def test_hub_batch_whole_model_parity_vs_the_fp64_oracle():
    """the narrow forms against the independent fp64 oracle on the batch of boundary in-degrees (forward and all gradients)"""
    b = hub_batch(7)
    m = make_model(7, 2)
    sd = cpu_state_dict(m)
    check_forward_parity(m, b, sd)
    check_backward_parity(m, b, sd)


def run_gcn_fwd(x_dev, rowptr, colidx, dinv, W, bias, N, Fin):
    L = _lib.lib()
    out = torch.full((N, 32), float("nan"), device=DEV)
    hs = torch.empty(N, 32, device=DEV)
    _lib.check(L.dgcnn_gcn_fwd(N, rowptr.data_ptr(), colidx.data_ptr(), dinv.data_ptr(), x_dev.data_ptr(), Fin, W.data_ptr(),
                               bias.data_ptr(), 32, out.data_ptr(), hs.data_ptr(), 0, None, _stream()), "gcn_fwd")
    torch.cuda.synchronize()
    return out.cpu(), hs.cpu()


@pytest.mark.parametrize("Fin,N", [(90, 14563), (33, 1000), (64, 517), (128, 300), (37, 16), (100, 15), (2, 4099), (129, 200)])
def test_staged_first_linear_equals_the_direct_form(Fin, N):
    """dgcnn_gcn_fwd's stand-alone first linear: 16-byte aligned input -> tile staged in LDS by 16-byte loads; the same values
    at a 4-byte offset -> the direct form.  Same matrix-instruction sequence on the same operands: identical bits.  (129 > the
    staged form's widest input: both runs take the direct form.)  Sizes: DD's, partial last tiles, a single partial tile."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(Fin * 7 + N)
    deg = 3
    ei = torch.randint(0, N, (2, deg * N), generator=g)
    batch = torch.zeros(N, dtype=torch.int64)
    E = ei.shape[1]
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=DEV); rowptr_t = torch.empty(N + 1, dtype=torch.int32, device=DEV)
    colidx = torch.zeros(E, dtype=torch.int32, device=DEV); colidx_t = torch.zeros(E, dtype=torch.int32, device=DEV)
    dinv = torch.empty(N, dtype=torch.float32, device=DEV)
    gptr = torch.empty(2, dtype=torch.int32, device=DEV)
    scratch = torch.empty(2 * N + 4 + 64, dtype=torch.int32, device=DEV)
    err = torch.zeros(4, dtype=torch.int32, device=DEV)
    _lib.check(L.dgcnn_graph_prep(ei.to(DEV).data_ptr(), E, batch.to(DEV).data_ptr(), N, 1, rowptr.data_ptr(), colidx.data_ptr(),
                                  rowptr_t.data_ptr(), colidx_t.data_ptr(), dinv.data_ptr(), gptr.data_ptr(), scratch.data_ptr(),
                                  err.data_ptr(), 0, None, None, _stream()), "prep")
    x = torch.randn(N, Fin, generator=g)
    W = (torch.randn(32, Fin, generator=g) / np.sqrt(Fin)).to(DEV).contiguous()
    bias = (torch.randn(32, generator=g) * 0.1).to(DEV)
    xa = x.to(DEV).contiguous()
    buf = torch.empty(N * Fin + 1, device=DEV)
    xb = buf[1:].view(N, Fin)
    xb.copy_(xa)
    assert xa.data_ptr() % 16 == 0 and xb.data_ptr() % 16 == 4
    oa, ha = run_gcn_fwd(xa, rowptr, colidx, dinv, W, bias, N, Fin)
    ob, hb = run_gcn_fwd(xb, rowptr, colidx, dinv, W, bias, N, Fin)
    assert torch.equal(ha, hb), float((ha - hb).abs().max())
    assert torch.equal(oa, ob)
    ref = (x.double() @ W.cpu().double().t()) * dinv.cpu().double()[:, None]
    assert float((ha.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
