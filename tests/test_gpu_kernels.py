"""GPU parity tests of the individual C-ABI entry points against the oracle (bit-exact for
integer/index work and pure data movement; stated tolerances for floating point)."""
import ctypes

import numpy as np
import pytest
import torch

from dgcnn_amd import _lib, synth
from oracle import kats, ref_ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def run_prep(ei, batch, N, B, flags=0):
    L = _lib.lib()
    E = ei.shape[1]
    ei_d, b_d = ei.to(DEV).contiguous(), batch.to(DEV).contiguous()
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=DEV)
    rowptr_t = torch.empty(N + 1, dtype=torch.int32, device=DEV)
    colidx = torch.full((max(E, 1),), -7, dtype=torch.int32, device=DEV)
    colidx_t = torch.full((max(E, 1),), -7, dtype=torch.int32, device=DEV)
    dinv = torch.empty(N, dtype=torch.float32, device=DEV)
    gptr = torch.empty(B + 1, dtype=torch.int32, device=DEV)
    scratch = torch.empty(2 * N + 4 + 64, dtype=torch.int32, device=DEV)
    err = torch.ones(4, dtype=torch.int32, device=DEV)
    _lib.check(L.dgcnn_graph_prep(ei_d.data_ptr() if E else None, E, b_d.data_ptr(), N, B, rowptr.data_ptr(),
                                  colidx.data_ptr(), rowptr_t.data_ptr(), colidx_t.data_ptr(), dinv.data_ptr(),
                                  gptr.data_ptr(), scratch.data_ptr(), err.data_ptr(), flags, None, None, _stream()), "prep")
    torch.cuda.synchronize()
    e = err.cpu().tolist()
    return rowptr.cpu(), colidx.cpu(), rowptr_t.cpu(), colidx_t.cpu(), dinv.cpu(), gptr.cpu(), (e[0], e[1])


def csr_reference(ei, N):
    """numpy CSR by target / by source, self loops dropped, neighbours ascending (duplicates kept)."""
    src, dst = ei[0].numpy(), ei[1].numpy()
    keep = src != dst
    src, dst = src[keep], dst[keep]
    def build(rows, cols):
        order = np.lexsort((cols, rows))
        r, c = rows[order], cols[order]
        ptr = np.zeros(N + 1, dtype=np.int64)
        np.add.at(ptr, r + 1, 1)
        return np.cumsum(ptr).astype(np.int32), c.astype(np.int32)
    rp, ci = build(dst, src)
    rpt, cit = build(src, dst)
    indeg = np.diff(rp)
    return rp, ci, rpt, cit, indeg


def random_multigraph(seed, sizes, deg, self_loops=True, dups=True, directed=True):
    g = torch.Generator().manual_seed(seed)
    eis, bs, off = [], [], 0
    for gi, n in enumerate(sizes):
        m = int(n * deg)
        s = torch.randint(0, n, (m,), generator=g)
        d = torch.randint(0, n, (m,), generator=g)
        if not self_loops:
            keep = s != d
            s, d = s[keep], d[keep]
        e = torch.stack([s, d])
        if not directed:
            e = torch.cat([e, e.flip(0)], 1)
        if dups and m > 2:
            e = torch.cat([e, e[:, :3]], 1)
        eis.append(e + off)
        bs.append(torch.full((n,), gi, dtype=torch.int64))
        off += n
    return torch.cat(eis, 1).contiguous(), torch.cat(bs), off


@pytest.mark.parametrize("case", ["mixed", "undirected", "no_edges", "hub_lds", "hub_global", "single_node_graphs"])
def test_graph_prep_bit_exact(case):
    if case == "mixed":
        ei, batch, N = random_multigraph(1, [5, 1, 40, 17, 300], 3.0)
    elif case == "undirected":
        ei, batch, N = random_multigraph(2, [64, 65, 2], 6.0, self_loops=False, dups=False, directed=False)
    elif case == "no_edges":
        ei, batch, N = torch.zeros(2, 0, dtype=torch.int64), torch.tensor([0, 0, 1]), 3
    elif case == "hub_lds":       # one row of degree 3000 -> workgroup bitonic in LDS
        n = 3001
        leaves = torch.randperm(n - 1) + 1
        ei = torch.stack([leaves, torch.zeros(n - 1, dtype=torch.int64)])
        ei = torch.cat([ei, ei.flip(0)], 1)
        batch, N = torch.zeros(n, dtype=torch.int64), n
    elif case == "hub_global":    # degree 9000 > 8192 -> in-place global bitonic path
        n = 9001
        leaves = torch.randperm(n - 1) + 1
        ei = torch.stack([leaves, torch.zeros(n - 1, dtype=torch.int64)])
        batch, N = torch.zeros(n, dtype=torch.int64), n
    else:
        ei = torch.tensor([[0, 1], [1, 0]])
        batch, N = torch.tensor([0, 0, 1, 2, 4]), 5     # graph 3 is EMPTY, graphs 1,2,4 single nodes
    B = int(batch.max()) + 1
    rp, ci, rpt, cit, dinv, gptr, err = run_prep(ei, batch, N, B)
    assert err == (0, 0)
    erp, eci, erpt, ecit, indeg = csr_reference(ei, N)
    np.testing.assert_array_equal(rp.numpy(), erp)
    np.testing.assert_array_equal(rpt.numpy(), erpt)
    np.testing.assert_array_equal(ci.numpy()[:erp[-1]], eci)
    np.testing.assert_array_equal(cit.numpy()[:erpt[-1]], ecit)
    want = torch.from_numpy(indeg.astype(np.float32) + 1).pow(-0.5)
    np.testing.assert_allclose(dinv.numpy(), want.numpy(), rtol=1.2e-7, atol=0)
    egp = np.searchsorted(batch.numpy(), np.arange(B + 1), side="left").astype(np.int32)
    np.testing.assert_array_equal(gptr.numpy(), egp)


def test_graph_prep_flags_out_of_range_edges():
    ei = torch.tensor([[0, 1, 7], [1, 0, 0]])
    *_, err = run_prep(ei, torch.zeros(3, dtype=torch.int64), 3, 1)
    assert err[0] != 0


@pytest.mark.parametrize("name", ["MUTAG", "COLLAB", "DD"])
def test_graph_prep_fast_path_equals_general_path(name):
    """DGCNN_FLAG_COALESCED_UNDIRECTED (no atomics, no sort, one launch) gives bit-identical structure."""
    b = synth.make_batch(name, 20, start=400)
    assert b.coalesced_undirected
    gen = run_prep(b.edge_index, b.batch, b.num_nodes, b.num_graphs, 0)
    fast = run_prep(b.edge_index, b.batch, b.num_nodes, b.num_graphs, _lib.FLAG_COALESCED_UNDIRECTED)
    assert fast[6] == (0, 0) and gen[6] == (0, 0)
    nnz = int(gen[0][-1])
    for a, c in zip(gen[:6], fast[:6]):
        if a.numel() == b.num_edges:
            assert torch.equal(a[:nnz], c[:nnz])
        else:
            assert torch.equal(a, c)


@pytest.mark.parametrize("violation", ["unsorted", "missing_reverse", "self_loop", "duplicate"])
def test_graph_prep_fast_path_detects_broken_promise(violation):
    b = synth.make_batch("PROTEINS", 4, start=10)
    ei = b.edge_index.clone()
    if violation == "unsorted":
        ei[:, [0, 1]] = ei[:, [1, 0]]
    elif violation == "missing_reverse":
        s, d = int(ei[0, 5]), int(ei[1, 5])
        keep = ~((ei[0] == d) & (ei[1] == s))
        ei = ei[:, keep]
    elif violation == "self_loop":
        ei[1, 3] = ei[0, 3]
    else:
        ei = torch.cat([ei[:, :4], ei[:, 3:]], 1)
    *_, err = run_prep(ei, b.batch, b.num_nodes, b.num_graphs, _lib.FLAG_COALESCED_UNDIRECTED)
    assert err[1] != 0


def run_gcn(x, ei, W, b, Fout):
    L = _lib.lib()
    N, Fin = x.shape
    batch = torch.zeros(N, dtype=torch.int64)
    E = ei.shape[1]
    ei_d = ei.to(DEV).contiguous()
    rowptr = torch.empty(N + 1, dtype=torch.int32, device=DEV)
    rowptr_t = torch.empty(N + 1, dtype=torch.int32, device=DEV)
    colidx = torch.zeros(max(E, 1), dtype=torch.int32, device=DEV)
    colidx_t = torch.zeros(max(E, 1), dtype=torch.int32, device=DEV)
    dinv = torch.empty(N, dtype=torch.float32, device=DEV)
    gptr = torch.empty(2, dtype=torch.int32, device=DEV)
    scratch = torch.empty(2 * N + 4 + 64, dtype=torch.int32, device=DEV)
    err = torch.zeros(4, dtype=torch.int32, device=DEV)
    bd = batch.to(DEV)
    _lib.check(L.dgcnn_graph_prep(ei_d.data_ptr() if E else None, E, bd.data_ptr(), N, 1, rowptr.data_ptr(),
                                  colidx.data_ptr(), rowptr_t.data_ptr(), colidx_t.data_ptr(), dinv.data_ptr(),
                                  gptr.data_ptr(), scratch.data_ptr(), err.data_ptr(), 0, None, None, _stream()), "prep")
    xd, Wd, bd2 = x.to(DEV).contiguous(), W.to(DEV).contiguous(), b.to(DEV).contiguous()
    out = torch.full((N, Fout), float("nan"), device=DEV)
    hs = torch.empty(N, Fout, device=DEV)
    _lib.check(L.dgcnn_gcn_fwd(N, rowptr.data_ptr(), colidx.data_ptr(), dinv.data_ptr(), xd.data_ptr(), Fin,
                               Wd.data_ptr(), bd2.data_ptr(), Fout, out.data_ptr(), hs.data_ptr(), 0, None, _stream()),
               "gcn_fwd")
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("kat", kats.gcn_kats(), ids=lambda k: k.name)
@pytest.mark.parametrize("Fout", [32, 1])
def test_gcn_layer_hand_kats(kat, Fout):
    """closed-form KATs through graph_prep + gcn_fwd; the KAT's weight rows are embedded in a
    [Fout,F] matrix (remaining rows zero -> tanh(bias=0) = 0)."""
    fo = kat.weight.shape[0]
    if Fout == 1 and fo != 1:
        pytest.skip("KAT has 2 outputs")
    x = torch.tensor(kat.x, dtype=torch.float32)
    F = x.shape[1]
    W = torch.zeros(Fout, F); W[:fo] = torch.tensor(kat.weight, dtype=torch.float32)
    b = torch.zeros(Fout); b[:fo] = torch.tensor(kat.bias, dtype=torch.float32)
    out = run_gcn(x, torch.tensor(kat.edge_index), W, b, Fout)
    np.testing.assert_allclose(out[:, :fo].numpy(), np.tanh(kat.expected), rtol=0, atol=2e-6)
    if Fout > fo:
        assert float(out[:, fo:].abs().max()) == 0.0


@pytest.mark.parametrize("Fin,Fout,n,deg", [(1, 32, 300, 20.0), (5, 32, 77, 3.0), (38, 32, 500, 4.0), (90, 32, 260, 5.0),
                                            (32, 32, 1000, 37.0), (32, 1, 1000, 37.0), (200, 32, 50, 2.0),
                                            (32, 32, 130, 100.0)])
def test_gcn_layer_vs_oracle(Fin, Fout, n, deg):
    g = torch.Generator().manual_seed(Fin * 1000 + n)
    ei, _, N = random_multigraph(Fin + n, [n], deg)
    x = torch.randn(N, Fin, generator=g)
    W = torch.randn(Fout, Fin, generator=g) * (1.0 / np.sqrt(Fin))
    b = torch.randn(Fout, generator=g) * 0.1
    out = run_gcn(x, ei, W, b, Fout)
    ei_ns = ref_ops.remove_self_loops(ei)
    ref64 = torch.tanh(ref_ops.gcn_conv(x.double(), ei_ns, W.double(), b.double()))
    ref32 = torch.tanh(ref_ops.gcn_conv(x, ei_ns, W, b))
    e64 = float((out.double() - ref64).abs().max())
    e32 = float((ref32.double() - ref64).abs().max())
    assert e64 <= 3e-6, (e64, e32)          # fp32 kernel vs fp64 truth; the fp32 CPU oracle itself is at e32


def run_sortpool(x, batch, B):
    L = _lib.lib()
    N = x.shape[0]
    gptr = torch.from_numpy(np.searchsorted(batch.numpy(), np.arange(B + 1)).astype(np.int32)).to(DEV)
    xs = [x[:, :32].contiguous().to(DEV), x[:, 32:64].contiguous().to(DEV), x[:, 64:96].contiguous().to(DEV),
          x[:, 96].contiguous().to(DEV)]
    pooled = torch.full((B, 30 * 97), float("nan"), device=DEV)
    perm = torch.full((B, 30), -9, dtype=torch.int32, device=DEV)
    _lib.check(L.dgcnn_sortpool_fwd(N, B, gptr.data_ptr(), *[t.data_ptr() for t in xs], pooled.data_ptr(),
                                    perm.data_ptr(), _stream()), "sortpool_fwd")
    torch.cuda.synchronize()
    return pooled.cpu(), perm.cpu(), gptr


@pytest.mark.parametrize("sizes", [[5, 30, 31, 64, 1], [200, 256, 257], [300, 1000, 4096], [4097, 6000, 29]],
                         ids=["small", "rank256", "bitonic", "select"])
def test_sortpool_bit_exact_vs_oracle(sizes):
    g = torch.Generator().manual_seed(sum(sizes))
    N = sum(sizes)
    x = torch.randn(N, 97, generator=g)
    x[:, 96] = torch.tanh(x[:, 96])
    batch = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(sizes)])
    pooled, perm, _ = run_sortpool(x, batch, len(sizes))
    ref = ref_ops.sort_pool(x, batch, 30, len(sizes), stable=True)
    assert torch.equal(pooled, ref)            # pure data movement: bit exact


def test_sortpool_ties_lower_index_first_and_signed_zero():
    x = torch.zeros(70, 97)
    x[:, 0] = torch.arange(70)
    x[:, 96] = 0.25                 # all keys tie -> nodes 0..29 in index order
    x[3, 96] = -0.0; x[4, 96] = 0.0; x[5, 96] = 0.5
    x[6:, 96] = -1.0
    batch = torch.zeros(70, dtype=torch.int64)
    pooled, perm, _ = run_sortpool(x, batch, 1)
    assert perm[0, :6].tolist() == [5, 0, 1, 2, 3, 4]      # -0.0 and +0.0 tie, index order kept
    assert perm[0, 6:].tolist() == list(range(6, 30))
    ref = ref_ops.sort_pool(x, batch, 30, 1, stable=True)
    assert torch.equal(pooled, ref)


@pytest.mark.parametrize("sizes", [[300], [1500, 257, 2048], [2049, 4096], [700, 40, 4500]])
@pytest.mark.parametrize("levels", [1, 3, 40])
def test_sortpool_heavy_ties_in_mid_size_graphs(sizes, levels):
    """graphs of 257..2048 nodes go through the radix select: keys drawn from very few distinct values make hundreds of
    nodes tie with the K-th key (candidate overflow -> the full-sort fallback) or exactly at a digit boundary; the
    selection must still be the stable descending order (lower index first) of the oracle, bit for bit."""
    g = torch.Generator().manual_seed(100 + levels)
    N = sum(sizes)
    x = torch.randn(N, 97, generator=g)
    vals = torch.tensor([0.5, -0.25, 0.75, 1e-30, -1e-30] + [float(v) for v in torch.randn(40, generator=g)])[:max(levels, 1)]
    x[:, 96] = vals[torch.randint(0, len(vals), (N,), generator=g)]
    batch = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(sizes)])
    pooled, perm, _ = run_sortpool(x, batch, len(sizes))
    ref = ref_ops.sort_pool(x, batch, 30, len(sizes), stable=True)
    assert torch.equal(pooled, ref)


@pytest.mark.parametrize("kat", kats.sortpool_kats(), ids=lambda k: k.name)
def test_sortpool_hand_kats_via_k30(kat):
    """The ABI fixes k=30; embed the D=2 KAT in channels (0, 96) and compare the first rows."""
    N = kat.x.shape[0]
    x = torch.zeros(N, 97)
    x[:, 0] = torch.tensor(kat.x[:, 0], dtype=torch.float32)
    x[:, 96] = torch.tensor(kat.x[:, 1], dtype=torch.float32)
    batch = torch.tensor(kat.batch)
    B = int(batch.max()) + 1
    pooled, perm, _ = run_sortpool(x, batch, B)
    ref = ref_ops.sort_pool(x, batch, 30, B, stable=True)
    assert torch.equal(pooled, ref)
    for g in range(B):
        n = int((batch == g).sum())
        want = [p for p in kat.perm[g].tolist() if p >= 0]
        if n <= kat.k:          # then the KAT's order is the full order
            assert perm[g, :len(want)].tolist() == want


def test_sortpool_bwd_exact():
    L = _lib.lib()
    sizes = [10, 45, 30]
    N, B = sum(sizes), 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 97, generator=g, requires_grad=True)
    batch = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(sizes)])
    pooled, perm, gptr = run_sortpool(x.detach(), batch, B)
    gp = torch.randn(B, 2910, generator=g)
    ref_ops.sort_pool(x, batch, 30, B, stable=True).backward(gp)
    outs = [torch.full((N, 32), float("nan"), device=DEV) for _ in range(3)] + [torch.full((N,), float("nan"), device=DEV)]
    permd, gpd = perm.to(DEV), gp.to(DEV)
    _lib.check(L.dgcnn_sortpool_bwd(N, B, gptr.data_ptr(), permd.data_ptr(), gpd.data_ptr(),
                                    *[t.data_ptr() for t in outs], _stream()), "sortpool_bwd")
    torch.cuda.synchronize()
    got = torch.cat([outs[0], outs[1], outs[2], outs[3].view(-1, 1)], 1).cpu()
    # padded slots carry gradient only into real rows; oracle zeroes the masked ones the same way
    assert torch.equal(got, x.grad)


def test_adam_matches_torch_adam():
    L = _lib.lib()
    g = torch.Generator().manual_seed(9)
    n = 5000
    p0 = torch.randn(n, generator=g)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt])            # defaults, as /root/reference/train.py:99
    p = p0.clone().to(DEV); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    for step in range(1, 6):
        gr = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g)))
        pt.grad = gr.clone()
        opt.step()
        gd = gr.to(DEV)
        _lib.check(L.dgcnn_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, step, 1e-3, 0.9,
                                     0.999, 1e-8, 1, _stream()), "adam")
        torch.cuda.synchronize()
        assert float(gd.abs().max()) == 0.0          # fused zero_grad
        np.testing.assert_allclose(p.cpu().numpy(), pt.detach().numpy(), rtol=2e-6, atol=2e-7)


# ---- dgcnn_gcn_bwd: one link of the backward chain, production kernels, vs torch autograd of the oracle's gcn_conv ----
def _prep_device(ei, batch, N, B, flags=0, dense=False):
    L = _lib.lib()
    E = ei.shape[1]
    d = {"ei": ei.to(DEV).contiguous(), "batch": batch.to(DEV).contiguous()}
    d["rowptr"] = torch.empty(N + 1, dtype=torch.int32, device=DEV)
    d["rowptr_t"] = torch.empty(N + 1, dtype=torch.int32, device=DEV)
    d["colidx"] = torch.zeros(max(E, 1), dtype=torch.int32, device=DEV)
    d["colidx_t"] = torch.zeros(max(E, 1), dtype=torch.int32, device=DEV)
    d["dinv"] = torch.empty(N, dtype=torch.float32, device=DEV)
    d["gptr"] = torch.empty(B + 1, dtype=torch.int32, device=DEV)
    scratch = torch.empty(2 * N + B + 4 + 64, dtype=torch.int32, device=DEV)
    err = torch.zeros(4, dtype=torch.int32, device=DEV)
    bits = tab = None
    if dense:
        bits = torch.empty(int(L.dgcnn_dense_bitmap_words(N)) + 4, dtype=torch.int32, device=DEV)
        tab = torch.empty(int(L.dgcnn_dense_table_ints(N, B)), dtype=torch.int32, device=DEV)
    _lib.check(L.dgcnn_graph_prep(d["ei"].data_ptr() if E else None, E, d["batch"].data_ptr(), N, B, d["rowptr"].data_ptr(),
                                  d["colidx"].data_ptr(), d["rowptr_t"].data_ptr(), d["colidx_t"].data_ptr(),
                                  d["dinv"].data_ptr(), d["gptr"].data_ptr(), scratch.data_ptr(), err.data_ptr(), flags,
                                  bits.data_ptr() if dense else None, tab.data_ptr() if dense else None, _stream()), "prep")
    torch.cuda.synchronize()
    assert err.cpu().tolist()[:2] == [0, 0]
    d["view"] = None
    if dense:
        v = _lib.DenseView()
        v.B, v.graph_ptr, v.item_table, v.adj_bits = B, d["gptr"].data_ptr(), tab.data_ptr(), bits.data_ptr()
        d["view"], d["_keep"] = v, (bits, tab)
    return d


def _run_gcn_bwd(d, N, gas, Fout, W, x_prev, first, gp_prev=None, ax=None):
    L = _lib.lib()
    Fin = x_prev.shape[1]
    t = lambda a: None if a is None else a.to(DEV).float().contiguous()
    gas_d, W_d, xp_d, gp_d, ax_d = t(gas), t(W), t(x_prev), t(gp_prev), t(ax)
    nb = int(L.dgcnn_gcn_bwd_scratch_bytes(N, Fin, Fout))
    scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)
    gas_prev = torch.full((N, 32), float("nan"), device=DEV)
    gW = torch.full((32, Fin) if Fout == 32 else (32,), float("nan"), device=DEV)
    gb = torch.full((32,), float("nan"), device=DEV)
    Fa = 0 if ax is None else ax.shape[1]
    gWaf = torch.full((32, max(Fa, 1)), float("nan"), device=DEV)
    p = lambda a: None if a is None else a.data_ptr()
    view = ctypes.byref(d["view"]) if d["view"] is not None else None
    _lib.check(L.dgcnn_gcn_bwd(N, d["rowptr_t"].data_ptr(), d["colidx_t"].data_ptr(), d["dinv"].data_ptr(), p(gas_d), Fout,
                               p(W_d), p(xp_d), Fin, int(first), p(gp_d), gas_prev.data_ptr(), gW.data_ptr(), gb.data_ptr(),
                               p(ax_d), Fa, gWaf.data_ptr() if ax is not None else None, view, scratch.data_ptr(), nb,
                               _stream()), "gcn_bwd")
    torch.cuda.synchronize()
    return gas_prev.cpu(), gW.cpu(), gb.cpu(), gWaf.cpu()


def _close(a, b, rtol=2e-4):
    scale = float(b.abs().max())
    err = float((a.double() - b.double()).abs().max())
    assert err <= rtol * max(scale, 1e-12) + 1e-7, (err, scale)


@pytest.mark.parametrize("form", ["conv32", "conv32_af", "first", "conv4"])
@pytest.mark.parametrize("dense", [False, True], ids=["gather", "dense"])
@pytest.mark.parametrize("workload,bs", [("PROTEINS", 9), ("COLLAB", 7)])
def test_gcn_bwd_link_vs_autograd(form, dense, workload, bs):
    """x_l = tanh(gcn_conv(x_prev, W, b)) with x_prev = tanh(z): for an upstream gradient G on x_l and an extra gradient gp
    on x_prev, autograd of the oracle gives dL/dW_l, dL/dz (-> gas_prev = dinv * dL/dz, gb_prev = sum dL/dz)."""
    start = 300
    b = synth.make_batch(workload, bs, start=start)
    while b.max_nodes > 512:
        start += bs
        b = synth.make_batch(workload, bs, start=start)
    N = b.num_nodes
    d = _prep_device(b.edge_index, b.batch, N, b.num_graphs, _lib.FLAG_COALESCED_UNDIRECTED, dense=dense)
    dinv = d["dinv"].cpu().double()
    g = torch.Generator().manual_seed(N)
    Fout = 1 if form == "conv4" else 32
    Fin = 19 if form == "first" else 32
    W = (torch.randn(Fout, Fin, generator=g) / np.sqrt(Fin)).double().requires_grad_(True)
    bias = (0.1 * torch.randn(Fout, generator=g)).double()
    G = torch.randn(N, Fout, generator=g).double()
    ei = ref_ops.remove_self_loops(b.edge_index)
    if form == "first":
        x_prev = torch.randn(N, Fin, generator=g).double()
        out = torch.tanh(ref_ops.gcn_conv(x_prev, ei, W, bias))
        (out * G).sum().backward()
        gas = dinv.view(-1, 1) * G * (1 - out.detach() ** 2)
        _, gW, _, _ = _run_gcn_bwd(d, N, gas, 32, None, x_prev, True)
        _close(gW, W.grad)
        return
    gp = torch.randn(N, 32, generator=g).double()
    ax = W1 = None
    if form == "conv32_af":       # x_prev = tanh(ax W1^T + b1) with ax = A_hat x the saved slab of an aggregate-first conv1
        Fa = 5
        ax = torch.randn(N, Fa, generator=g).double()
        W1 = (torch.randn(32, Fa, generator=g) / np.sqrt(Fa)).double().requires_grad_(True)
        z = ax @ W1.t()
        z.retain_grad()
    else:
        z = torch.randn(N, 32, generator=g).double().requires_grad_(True)
    x_prev = torch.tanh(z)
    out = torch.tanh(ref_ops.gcn_conv(x_prev, ei, W, bias))
    ((out * G).sum() + (x_prev * gp).sum()).backward()
    gas = dinv.view(-1, 1) * G * (1 - out.detach() ** 2)
    gas_prev, gW, gb, gWaf = _run_gcn_bwd(d, N, gas, Fout, W.detach(), x_prev.detach(), False, gp, ax)
    _close(gW.reshape(-1), W.grad.reshape(-1))
    _close(gb, z.grad.sum(0))
    if form == "conv32_af":
        _close(gWaf, W1.grad)
    else:
        _close(gas_prev, dinv.view(-1, 1) * z.grad)


def test_gcn_fwd_dense_and_bf16_storage_flag():
    """dgcnn_gcn_fwd with the dense view equals the gather form within fp32 order noise; with DGCNN_FLAG_BF16 the
    pre-scaled linear output is stored in bf16 (stated tolerance 1e-2 on |out| <= 1)"""
    L = _lib.lib()
    b = synth.make_batch("COLLAB", 6, start=11)
    N = b.num_nodes
    d = _prep_device(b.edge_index, b.batch, N, b.num_graphs, _lib.FLAG_COALESCED_UNDIRECTED, dense=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 32, generator=g); W = torch.randn(32, 32, generator=g) / np.sqrt(32); bias = 0.1 * torch.randn(32, generator=g)
    xd, Wd, bd = x.to(DEV), W.to(DEV), bias.to(DEV)
    outs = {}
    for name, flags, view in (("gather", 0, None), ("dense", _lib.FLAG_AGG_DENSE, d["view"]), ("bf16", _lib.FLAG_BF16, d["view"])):
        out = torch.full((N, 32), float("nan"), device=DEV)
        hs = torch.empty(N, 32, device=DEV)
        _lib.check(L.dgcnn_gcn_fwd(N, d["rowptr"].data_ptr(), d["colidx"].data_ptr(), d["dinv"].data_ptr(), xd.data_ptr(), 32,
                                   Wd.data_ptr(), bd.data_ptr(), 32, out.data_ptr(), hs.data_ptr(), flags,
                                   ctypes.byref(view) if view is not None else None, _stream()), "gcn_fwd")
        torch.cuda.synchronize()
        outs[name] = out.cpu()
    ref = torch.tanh(ref_ops.gcn_conv(x.double(), ref_ops.remove_self_loops(b.edge_index), W.double(), bias.double()))
    assert float((outs["gather"].double() - ref).abs().max()) <= 3e-6
    assert float((outs["dense"].double() - ref).abs().max()) <= 3e-6
    e = float((outs["bf16"].double() - ref).abs().max())
    assert 1e-5 < e <= 1e-2, e


# ---- one-shot gradient exchange (csrc/peer.hip): the bounded wait and its all-rank verdict, driven through the C ABI ----
def _peer_call(L, world, rank, grads, flags, tag, params, m, v, err, stream=0, step=1):
    import ctypes
    arr = ctypes.c_void_p * world
    return L.dgcnn_allreduce_adam_step(world, rank, arr(*[g.data_ptr() for g in grads]), arr(*[f.data_ptr() for f in flags]), tag,
                                       params.data_ptr(), m.data_ptr(), v.data_ptr(), None, params.numel(), step, 1e-3, 0.9,
                                       0.999, 1e-8, err.data_ptr(), stream)


def test_one_shot_exchange_times_out_consistently_and_applies_the_step_on_no_rank():
    """a rank that never publishes: the waiting rank gives up after the bounded wait (no hang), flags err[0], and leaves its
    replica untouched; the LATE rank then finds the gradient in place but also the other's abort verdict and skips the step
    too -- replicas stay identical, both ranks report the error"""
    L = _lib.lib()
    n = 5000
    dev = "cuda"
    torch.manual_seed(0)
    grads = [torch.randn(n, device=dev), torch.randn(n, device=dev)]
    flags = [torch.zeros(16, dtype=torch.int32, device=dev) for _ in range(2)]
    p0 = torch.randn(n, device=dev); p1 = p0.clone()
    st = [(torch.zeros(n, device=dev), torch.zeros(n, device=dev)) for _ in range(2)]
    errs = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(2)]
    keep = p0.clone()
    assert L.dgcnn_peer_set_timeout_ms(30) == 0
    try:
        assert _peer_call(L, 2, 0, grads, flags, 1, p0, st[0][0], st[0][1], errs[0]) == 0
        torch.cuda.synchronize()                                   # returns: the wait is bounded
        assert int(errs[0][0]) == 1 and torch.equal(p0, keep) and float(st[0][0].abs().sum()) == 0.0
        assert _peer_call(L, 2, 1, grads, flags, 1, p1, st[1][0], st[1][1], errs[1]) == 0      # the late rank
        torch.cuda.synchronize()
        assert int(errs[1][0]) == 1 and torch.equal(p1, keep)
    finally:
        L.dgcnn_peer_set_timeout_ms(20000)


def test_one_shot_exchange_two_ranks_on_two_streams_apply_the_identical_step():
    L = _lib.lib()
    n = 52000
    dev = "cuda"
    torch.manual_seed(1)
    grads = [torch.randn(n, device=dev), torch.randn(n, device=dev)]
    flags = [torch.zeros(16, dtype=torch.int32, device=dev) for _ in range(2)]
    p = [torch.randn(n, device=dev)]; p.append(p[0].clone())
    st = [(torch.zeros(n, device=dev), torch.zeros(n, device=dev)) for _ in range(2)]
    errs = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(2)]
    ref = p[0].clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    ref.grad = grads[0] + grads[1]
    opt.step()
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    assert _peer_call(L, 2, 0, grads, flags, 1, p[0], st[0][0], st[0][1], errs[0], stream=s0.cuda_stream) == 0
    assert _peer_call(L, 2, 1, grads, flags, 1, p[1], st[1][0], st[1][1], errs[1], stream=s1.cuda_stream) == 0
    torch.cuda.synchronize()
    assert int(errs[0][0]) == 0 and int(errs[1][0]) == 0
    assert torch.equal(p[0], p[1])
    assert torch.allclose(p[0], ref.detach(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n", [104128, 4, 5000, 52004])
def test_one_shot_exchange_adam_equals_k_adam_bit_for_bit(n):
    """the exchange kernel's Adam (four elements per thread, operands requested before the flag protocol) against
    dgcnn_adam_step on the same gradient, state and step number: identical bits in the parameters and both moments (the
    Adam element has a pinned operation sequence, dg_adam_elem: left to the contraction heuristic the two kernels compiled to
    different fused / unfused forms).  Also: a length that is not a multiple of 4 floats is refused."""
    L = _lib.lib()
    dev = "cuda"
    torch.manual_seed(n)
    g = torch.randn(n, device=dev) * 1e-2
    p0 = torch.randn(n, device=dev); m0 = torch.randn(n, device=dev) * 1e-3; v0 = torch.rand(n, device=dev) * 1e-4
    pa, ma, va, ga = p0.clone(), m0.clone(), v0.clone(), g.clone()
    _lib.check(L.dgcnn_adam_step(pa.data_ptr(), ga.data_ptr(), ma.data_ptr(), va.data_ptr(), n, 3, 1e-3, 0.9, 0.999, 1e-8, 0,
                                 _stream()), "adam")
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    flags = [torch.zeros(16, dtype=torch.int32, device=dev)]
    err = torch.zeros(4, dtype=torch.int32, device=dev)
    gp = (ctypes.c_void_p * 1)(g.data_ptr()); fp = (ctypes.c_void_p * 1)(flags[0].data_ptr())
    rc = L.dgcnn_allreduce_adam_step(1, 0, gp, fp, 1, pb.data_ptr(), mb.data_ptr(), vb.data_ptr(), None, n, 3, 1e-3, 0.9, 0.999,
                                     1e-8, err.data_ptr(), _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert int(err[0]) == 0
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    if n > 8:
        assert L.dgcnn_allreduce_adam_step(1, 0, gp, fp, 2, pb.data_ptr(), mb.data_ptr(), vb.data_ptr(), None, n - 1, 3, 1e-3, 0.9,
                                           0.999, 1e-8, err.data_ptr(), _stream()) == -1
