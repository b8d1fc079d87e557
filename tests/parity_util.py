"""Shared helpers for the GPU parity tests: tie-aware comparison of the HIP path against the
oracle (tests may import ``oracle``; the product never does)."""
from __future__ import annotations

import numpy as np
import torch

from dgcnn_amd.batch import Batch
from oracle import ref_dense

LOGIT_TOL = 1e-4          # BASELINE.json north_star: "logits within 1e-4 fp32"
XCAT_TOL = 2e-5           # per-node activations after 4 fp32 layers vs the fp64 oracle
KEY_TOL = 2e-5            # sort keys closer than this may legitimately order either way


def load_fixture(golden_dir, name, coalesced_undirected=False):
    """`coalesced_undirected`: build the Batch WITH the promise the fixtures' edge lists satisfy (sorted by (src,dst), both
    directions, no self loops -- asserted here), which routes a fused training step through k_chain_readout_tail"""
    z = np.load(f"{golden_dir}/{name}.npz")
    sd = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")}
    grads = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad:")}
    b = Batch(torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"]), torch.from_numpy(z["batch"]),
              torch.from_numpy(z["y"]))
    if coalesced_undirected:
        ei = b.edge_index
        key = ei[0] * (int(ei.max()) + 1) + ei[1]
        assert bool((key[1:] > key[:-1]).all()) and bool((ei[0] != ei[1]).all())
        rkey = ei[1] * (int(ei.max()) + 1) + ei[0]
        assert torch.equal(torch.sort(rkey).values, key)
        n_per = torch.bincount(b.batch, minlength=b.num_graphs)
        e_per = torch.bincount(b.batch[ei[0]], minlength=b.num_graphs)
        b = Batch(b.x, b.edge_index, b.batch, b.y, b.num_graphs, True, int(n_per.max()), int(e_per.max()))
    return z, sd, grads, b


def make_model(F, C, sd=None, seed=324, device="cuda"):
    from dgcnn_amd.model import Model
    torch.manual_seed(seed)
    m = Model(F, C)
    if sd is not None:
        m.load_state_dict(sd)
    else:
        with torch.no_grad():     # non-zero GCN biases so the bias path is exercised
            for i in (1, 2, 3, 4):
                getattr(m, f"conv{i}").bias.uniform_(-0.1, 0.1)
    return m.to(device)


def cpu_state_dict(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def gpu_xcat(m):
    return torch.cat([m.last_workspace_view("x1"), m.last_workspace_view("x2"), m.last_workspace_view("x3"),
                      m.last_workspace_view("x4").view(-1, 1)], dim=1).cpu()


def check_forward_parity(m, b_cpu, sd, logit_tol=LOGIT_TOL):
    """eval-mode forward on the GPU vs the fp64 dense oracle, tie-aware:
       1. per-node activations [N,97] agree within XCAT_TOL,
       2. the kernel's SortPooling permutation is a valid top-k of the ORACLE's keys within KEY_TOL,
       3. log-probs agree within 1e-4 with the oracle evaluated on that same permutation."""
    m.eval()
    with torch.no_grad():
        logp = m(b_cpu.to("cuda")).cpu()
    m.check_errors()
    _, aux = ref_dense.forward_dense(sd, b_cpu.x, b_cpu.edge_index, b_cpu.batch, b_cpu.num_graphs, return_all=True)
    xc = gpu_xcat(m)
    err_x = float((xc.double() - aux["xcat"].detach()).abs().max())
    assert err_x <= XCAT_TOL, f"per-node activations differ by {err_x:.3e}"
    perm = m.last_workspace_view("perm").cpu()
    ok, msg = ref_dense.check_perm_valid(aux["xcat"], aux["ptr"], perm, tol=KEY_TOL)
    assert ok, msg
    ref = ref_dense.forward_dense(sd, b_cpu.x, b_cpu.edge_index, b_cpu.batch, b_cpu.num_graphs,
                                  perm_override=perm).detach()
    err = float((logp.double() - ref).abs().max())
    assert err <= logit_tol, f"log-probs differ by {err:.3e} (> {logit_tol})"
    return logp, perm, err, err_x


def grads_close(g_gpu, g_ref, rtol=1e-3, atol_rel=2e-5):
    """allclose with an absolute floor relative to the largest reference entry of that tensor."""
    g_ref = g_ref.double()
    scale = float(g_ref.abs().max())
    tol = rtol * g_ref.abs() + atol_rel * max(scale, 1e-12) + 1e-9
    diff = (g_gpu.double() - g_ref).abs()
    bad = diff > tol
    return (not bool(bad.any())), float(diff.max()), scale


def check_backward_parity(m, b_cpu, sd):
    """training-mode forward+backward on the GPU (drop-in autograd route) vs fp64 oracle gradients,
    using the kernel's own dropout mask and SortPooling permutation."""
    m.train()
    bg = b_cpu.to("cuda")
    logp = m(bg)
    loss = torch.nn.functional.nll_loss(logp, bg.y)
    m.zero_grad(set_to_none=True)
    loss.backward()
    mask = m.last_workspace_view("drop_mask").cpu()
    perm = m.last_workspace_view("perm").cpu()
    logp_ref, loss_ref, g_ref, aux = ref_dense.loss_and_grads_dense(
        sd, b_cpu.x, b_cpu.edge_index, b_cpu.batch, b_cpu.y, b_cpu.num_graphs, dropout_mask=mask,
        perm_override=perm)
    ok, msg = ref_dense.check_perm_valid(aux["xcat"], aux["ptr"], perm, tol=KEY_TOL)
    assert ok, msg
    assert abs(float(loss.detach().cpu()) - float(loss_ref)) <= 1e-5, (float(loss), float(loss_ref))
    worst = {}
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        good, md, sc = grads_close(p.grad.detach().cpu(), g_ref[name])
        worst[name] = (md, sc)
        assert good, f"grad {name}: max diff {md:.3e} at scale {sc:.3e}"
    frac = float(mask.float().mean())
    assert 0.3 < frac < 0.7, f"dropout keep fraction {frac}"
    return worst
