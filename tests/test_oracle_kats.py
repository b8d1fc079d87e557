"""Pins the oracle: README parameter counts, shape chain, closed-form
hand KATs, and agreement of its two independent formulations.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import kats, ref_dense, ref_ops


@pytest.mark.parametrize("name", list(kats.README_PARAM_COUNTS))
def test_readme_parameter_counts(name):
    # /root/reference/README.md:95-105 -- number of parameters per dataset
    F, C, expected = kats.README_PARAM_COUNTS[name]
    assert ref_ops.count_parameters(F, C) == expected


def test_shape_chain():
    # /root/reference/model.py:18-21,36-40: 2910 -> [16,30] -> [16,15] -> [32,11] -> 352
    m = ref_ops.RefModel(1, 3)
    x = torch.randn(3, 1, 30 * 97)
    a = torch.relu(m.conv5(x)); assert a.shape == (3, 16, 30)
    a = m.pool(a); assert a.shape == (3, 16, 15)
    a = torch.relu(m.conv6(a)); assert a.shape == (3, 32, 11)
    assert a.view(3, -1).shape[1] == 352 == m.classifier_1.in_features


def test_state_dict_keys_match_reference_names():
    # PyG GCNConv keys are convN.bias / convN.lin.weight
    keys = set(ref_ops.RefModel(5, 2).state_dict().keys())
    want = {f"conv{i}.{s}" for i in (1, 2, 3, 4) for s in ("bias", "lin.weight")}
    want |= {f"{m}.{s}" for m in ("conv5", "conv6", "classifier_1", "classifier_2") for s in ("weight", "bias")}
    assert keys == want


@pytest.mark.parametrize("kat", kats.gcn_kats(), ids=lambda k: k.name)
def test_gcn_hand_kats_ref_ops(kat):
    x = torch.tensor(kat.x, dtype=torch.float64)
    ei = ref_ops.remove_self_loops(torch.tensor(kat.edge_index, dtype=torch.int64))
    out = ref_ops.gcn_conv(x, ei, torch.tensor(kat.weight, dtype=torch.float64),
                           torch.tensor(kat.bias, dtype=torch.float64))
    np.testing.assert_allclose(out.numpy(), kat.expected, rtol=0, atol=1e-13)


@pytest.mark.parametrize("kat", kats.gcn_kats(), ids=lambda k: k.name)
def test_gcn_hand_kats_ref_dense(kat):
    n = kat.x.shape[0]
    ei = torch.tensor(kat.edge_index, dtype=torch.int64)
    # dense oracle works per graph: find connected blocks via the KAT's own structure
    if kat.name == "two_graphs":
        blocks = [(0, 2), (2, 4)]
    else:
        blocks = [(0, n)]
    x = torch.tensor(kat.x, dtype=torch.float64)
    W = torch.tensor(kat.weight, dtype=torch.float64)
    b = torch.tensor(kat.bias, dtype=torch.float64)
    outs = []
    for n0, n1 in blocks:
        Ah = ref_dense.dense_norm_adj(ei, n0, n1)
        outs.append(Ah @ (x[n0:n1] @ W.t()) + b)
    np.testing.assert_allclose(torch.cat(outs).numpy(), kat.expected, rtol=0, atol=1e-13)


@pytest.mark.parametrize("kat", kats.sortpool_kats(), ids=lambda k: k.name)
def test_sortpool_hand_kats(kat):
    x = torch.tensor(kat.x, dtype=torch.float64)
    batch = torch.tensor(kat.batch, dtype=torch.int64)
    B = int(batch.max()) + 1
    out = ref_ops.sort_pool(x, batch, kat.k, B, stable=True)
    np.testing.assert_array_equal(out.numpy(), kat.expected)
    ptr = torch.zeros(B + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(torch.bincount(batch, minlength=B), 0)
    out2, perm = ref_dense.sort_pool_dense(x, ptr, B, kat.k)
    np.testing.assert_array_equal(out2.numpy(), kat.expected)
    np.testing.assert_array_equal(perm.numpy(), kat.perm)
    ok, msg = ref_dense.check_perm_valid(x, ptr, perm, kat.k, tol=0.0)
    assert ok, msg


def test_sortpool_result_independent_of_batch_composition():
    # pooling a graph alone == pooling it inside a batch with a larger graph
    torch.manual_seed(0)
    xa, xb = torch.randn(7, 97), torch.randn(50, 97)
    alone = ref_ops.sort_pool(xa, torch.zeros(7, dtype=torch.int64), 30, 1, stable=True)
    both = ref_ops.sort_pool(torch.cat([xa, xb]), torch.cat([torch.zeros(7), torch.ones(50)]).long(), 30, 2, stable=True)
    assert torch.equal(alone[0], both[0])


def _load(golden_dir, name):
    z = np.load(f"{golden_dir}/{name}.npz")
    sd = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")}
    grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad:")}
    return z, sd, grads


@pytest.mark.parametrize("name", ["mutag_b6", "proteins_b5", "collab_b4"])
def test_golden_consistent_with_both_formulations(golden_dir, name):
    """The committed vectors are reproduced by the fp32 edge-list oracle AND by the fp64
    dense oracle (independent code paths) -- guards against fixture/oracle drift."""
    z, sd, grads = _load(golden_dir, name)
    from dgcnn_amd.batch import Batch
    b = Batch(torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"]), torch.from_numpy(z["batch"]),
              torch.from_numpy(z["y"]))
    m = ref_ops.RefModel(int(z["num_features"]), int(z["num_classes"]))
    m.load_state_dict(sd)
    m.eval(); m.stable_sort = True
    with torch.no_grad():
        lp = m(b)
    np.testing.assert_allclose(lp.numpy(), z["logp_eval_f32"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(lp.numpy(), z["logp_eval_f64"], rtol=0, atol=2e-6)
    lp64 = ref_dense.forward_dense(sd, b.x, b.edge_index, b.batch, b.num_graphs)
    np.testing.assert_allclose(lp64.detach().numpy(), z["logp_eval_f64"], rtol=0, atol=1e-12)
    # training step with the stored dropout mask: fp32 autograd vs stored fp64 grads
    m.train()
    mask = torch.from_numpy(z["dropout_mask"])
    # use the fixture's permutation so a near-tie cannot flip rows between formulations
    xcat = m.graph_features(b)
    pooled, _ = ref_dense.sort_pool_dense(xcat, torch.tensor(np.concatenate([[0], np.cumsum(np.bincount(z["batch"]))])),
                                          b.num_graphs, perm_override=torch.from_numpy(z["perm"]))
    logp = m.tail(pooled, mask)
    loss = ref_ops.nll_mean(logp, b.y)
    loss.backward()
    assert abs(float(loss.detach()) - float(z["loss_train_f64"])) < 1e-5
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), grads[k].numpy(), rtol=2e-3, atol=2e-6, err_msg=k)


def test_unstable_sort_differs_only_on_ties():
    # the reference's default sort is unstable (SURVEY trap #2); without ties both agree
    torch.manual_seed(1)
    x = torch.randn(40, 97)
    b = torch.zeros(40, dtype=torch.int64)
    assert torch.equal(ref_ops.sort_pool(x, b, 30, 1, stable=False), ref_ops.sort_pool(x, b, 30, 1, stable=True))


@pytest.mark.parametrize("name", ["mutag_b6", "proteins_b5", "collab_b4"])
def test_step_fixture_is_self_consistent(golden_dir, name):
    """the eval-mode step stored in the fixtures: fp64 dense loss / gradients are reproduced by the independent fp32
    edge-list formulation, and the stored post-Adam parameters are torch.optim.Adam's first step."""
    import numpy as np
    from dgcnn_amd.batch import Batch
    from oracle import ref_ops
    z = np.load(f"{golden_dir}/{name}.npz")
    b = Batch(torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"]), torch.from_numpy(z["batch"]),
              torch.from_numpy(z["y"]))
    model = ref_ops.RefModel(int(z["num_features"]), int(z["num_classes"]))
    model.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")})
    model.eval(); model.stable_sort = True
    opt = torch.optim.Adam(model.parameters())
    loss = torch.nn.NLLLoss()(model(b), b.y)
    loss.backward()
    assert abs(float(loss) - float(z["loss_eval_f64"])) < 2e-5
    named = dict(model.named_parameters())
    if float(z["sort_margin"]) >= 1e-4:          # tie-free fixtures: both formulations select the same nodes
        for k, p in named.items():
            g_ref = torch.from_numpy(z["grad_eval:" + k])
            assert torch.allclose(p.grad, g_ref, rtol=2e-3, atol=2e-5 * float(g_ref.abs().max()) + 1e-9), k
        opt.step()
        for k, p in named.items():
            g_ref = torch.from_numpy(z["grad_eval:" + k])
            sure = g_ref.abs() > 1e-3 * g_ref.abs().max().clamp_min(1e-30)
            assert (p.detach()[sure] - torch.from_numpy(z["adam1:" + k])[sure]).abs().max() < 5e-6, k


def test_one_command_pin_reports_unpinned_without_pyg():
    """`python -m oracle.make_golden --from-reference` is the one-command pin: with torch_geometric
    present it diffs the committed fixtures against the real /root/reference/model.py; today PyG is absent, so it must
    say so -- the parity status of this build is 'unpinned' and nothing may pretend otherwise."""
    from oracle import make_golden
    lines = []
    rc = make_golden.pin_against_reference(out=lines.append)
    try:
        import torch_geometric  # noqa: F401
        have = True
    except Exception:
        have = False
    if have:
        assert rc in (0, 2), "\n".join(lines)       # 2: PyG present but /root/reference absent (GPU box)
    else:
        assert rc == 2 and make_golden.UNPINNED_MSG in lines[0]
