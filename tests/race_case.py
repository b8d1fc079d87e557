"""Large-batch parity case aimed at the persistent graph-chain kernels' loop-end hazard (VERDICT r5 weak 2 / ADVICE r5 medium):
`k_chain_fwd_q<8, .., LOOP>` and `k_chain_bwd_a<8, LOOP>` walk several graphs per workgroup and re-stage the LDS bitmap rows `bl`
for the next graph at the loop top, while the previous iteration's LAST phase reads `bl` word by word.  Every graph here has
129..256 nodes (two row tiles on some waves, one on others: maximally unequal per-wave work, row stride 8 words, so the rows a wave
reads are staged by OTHER waves), the batch has more graphs than persistent workgroups (every workgroup walks >= 2 graphs), and big
and small graphs alternate in each workgroup's walk.

Run as a script it executes the case against whatever library DGCNN_HIP_LIB names (tests/test_gpu_chain.py runs it once on the
product build and once on variants/lib_racedelay.so, where waves 0/3/6 enter the last phase ~30 k cycles late)."""
import sys

import numpy as np
import torch

import os

B_RACE = int(os.environ.get("DGCNN_RACE_B", "1100"))      # (the CPU emulation run of this case uses 600: still > 512 workgroups' worth)


def race_sizes(B=B_RACE, seed=6):
    rng = np.random.default_rng(seed)
    lo = rng.integers(129, 145, size=B)          # 9 tiles: wave 0 carries two tiles, waves 1..7 one
    hi = rng.integers(241, 257, size=B)          # 16 tiles: every wave two
    mid = rng.integers(145, 241, size=B)
    pick = rng.integers(0, 3, size=B)
    return [int(v) for v in np.where(pick == 0, lo, np.where(pick == 1, hi, mid))]


def run_case(B=B_RACE, verbose=False):
    from parity_util import check_backward_parity, check_forward_parity, cpu_state_dict, gpu_xcat, make_model
    from test_gpu_dense import _sized_batch
    from dgcnn_amd import _lib
    b = _sized_batch(race_sizes(B), seed=66)
    assert 129 <= int(torch.bincount(b.batch).min()) and b.max_nodes <= 256
    m = make_model(3, 2)
    sd = cpu_state_dict(m)
    m.agg_mode, m.use_chain, m.use_fused = "dense", True, None          # graph-chain kernels: persistent form above 256 graphs
    check_forward_parity(m, b, sd)                                      # [N,97] <= 2e-5, legal top-k, log-probs <= 1e-4 vs fp64
    xc = gpu_xcat(m)
    check_backward_parity(m, b, sd)                                     # 16 gradients rtol 1e-3 vs fp64 (k_chain_bwd_a / _b)
    # twice the same: run-to-run bit-reproducible (GPU only, no oracle)
    bg = b.to("cuda")
    m.eval()
    with torch.no_grad():
        m(bg)
    assert torch.equal(xc, gpu_xcat(m))
    # the per-layer dense route on the same batch (launch per layer, nothing walks graphs inside a workgroup's LDS image)
    m.agg_mode, m.use_chain = "dense", False
    m.eval()
    with torch.no_grad():
        m(bg)
    m.check_errors()
    xd = gpu_xcat(m)
    d_dense = float((xc - xd).abs().max())
    # the same graphs through the one-graph-per-workgroup chain form (<= 256 graphs per call: LOOP = false, nothing is re-staged)
    from dgcnn_amd.batch import collate
    from dgcnn_amd.batch import Graph
    m.agg_mode, m.use_chain = "dense", True
    ptr = np.searchsorted(b.batch.numpy(), np.arange(b.num_graphs + 1))
    d_one = 0.0
    eq_one = True
    m.eval()
    eptr = np.searchsorted(b.batch[b.edge_index[0]].numpy(), np.arange(b.num_graphs + 1))
    for g0 in range(0, b.num_graphs, 200):
        g1 = min(g0 + 200, b.num_graphs)
        gs = []
        for g in range(g0, g1):
            ei = b.edge_index[:, eptr[g]:eptr[g + 1]] - int(ptr[g])
            gs.append(Graph(x=b.x[ptr[g]:ptr[g + 1]], edge_index=ei, y=int(b.y[g]), coalesced_undirected=True))
        sub = collate(gs)
        with torch.no_grad():
            m(sub.to("cuda"))
        m.check_errors()
        xs = gpu_xcat(m)
        ref = xc[ptr[g0]:ptr[g1]]
        d_one = max(d_one, float((xs - ref).abs().max()))
        eq_one = eq_one and torch.equal(xs, ref)
    if verbose:
        print(f"race case: lib={_lib.LIB_PATH} B={b.num_graphs} N={b.num_nodes} chain-vs-per-layer-dense max|d|={d_dense:.3e} "
              f"bit_equal={torch.equal(xc, xd)} chain(persistent)-vs-chain(one graph per workgroup) max|d|={d_one:.3e} bit_equal={eq_one}")
    assert d_dense <= 4e-6, d_dense                                     # same sums, different order
    assert d_one <= 4e-6, d_one                                         # (bit_equal is reported; a staged-over bitmap row shows as ~1e-1)
    return d_dense, d_one


if __name__ == "__main__":
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    run_case(int(sys.argv[1]) if len(sys.argv) > 1 else B_RACE, verbose=True)
    print("RACE_CASE_OK")
