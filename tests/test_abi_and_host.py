"""CPU-only checks: the C-ABI library builds/loads and exports every symbol include/*.h declares,
host-side logic (collate, split, generator, layouts), and that the product refuses to run
without the GPU path (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from dgcnn_amd import _lib, synth
from dgcnn_amd.batch import Batch, Graph, collate, indegree_feature, split_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dgcnn_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dgcnn_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound(built_lib):
    names = _declared_symbols()
    assert len(names) >= 12
    assert set(names) == set(_lib.SIGNATURES), (names, sorted(_lib.SIGNATURES))
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} not exported by libdgcnn_hip.so"
    assert built_lib.dgcnn_version() == _lib.ABI_VERSION


def test_library_has_gfx950_code_object():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={_lib.LIB_PATH}"], capture_output=True, text=True)
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, out.stdout + out.stderr


@pytest.mark.parametrize("name", ["MUTAG", "PTC", "NCI1", "PROTEINS", "DD", "COLLAB", "IMDB-B", "IMDB-M"])
def test_param_layout_matches_readme_counts(built_lib, name):
    # /root/reference/README.md:95-105 -- flat layout holds exactly the reference's parameters
    from oracle.kats import README_PARAM_COUNTS
    F, C, expected = README_PARAM_COUNTS[name]
    offs, total = _lib.param_layout(F, C)
    sizes = [32 * F, 32, 1024, 32, 1024, 32, 32, 1, 16 * 97, 16, 32 * 16 * 5, 32, 128 * 352, 128, C * 128, C]
    assert sum(sizes) == expected
    for o, s, nxt in zip(offs, sizes, offs[1:] + [total]):
        assert o % 4 == 0 and o + s <= nxt
    from dgcnn_amd.model import Model
    m = Model(F, C)
    assert sum(p.numel() for p in m.parameters()) == expected
    assert [tuple(p.shape) for p in m._param_list()][:2] == [(32, F), (32,)]


def test_bad_shapes_rejected(built_lib):
    with pytest.raises(_lib.DgcnnError):
        _lib.param_layout(0, 2)
    with pytest.raises(_lib.DgcnnError):
        _lib.param_layout(4, 1000)
    assert built_lib.dgcnn_workspace_bytes(-1, 0, 1, 1, 2) < 0
    # null pointers are refused before any launch
    assert built_lib.dgcnn_model_forward(10, 0, 1, 1, 2, None, None, None, None, None, None, 0, 0, 0, 0, 0, 1, None) == -1
    assert built_lib.dgcnn_adam_step(None, None, None, None, 10, 1, 1e-3, .9, .999, 1e-8, 1, None) == -1


def test_workspace_regions_disjoint_and_aligned(built_lib):
    N, E, B, F, C = 1000, 7000, 13, 5, 2
    total = _lib.workspace_bytes(N, E, B, F, C)
    names = ["err", "cnt_in", "cnt_out", "rowptr", "rowptr_t", "colidx", "colidx_t", "dinv", "graph_ptr", "hsA", "hsB",
             "h4s", "x1", "x2", "x3", "x4", "perm", "pooled", "a5", "a6", "a1d", "drop_mask", "dlogit", "gz1", "gz6",
             "gz5", "gp1", "gp2", "gp3", "gas4", "gasA", "gasB", "lossv", "gb4p", "pa4", "pb3", "pb2", "pb1"]
    offs = [_lib.workspace_offset(n, N, E, B, F, C) for n in names]
    assert offs == sorted(offs) and len(set(offs)) == len(offs)
    assert all(o % 256 == 0 for o in offs) and offs[-1] < total


def test_model_refuses_cpu_tensors():
    """No silent fallback: a CPU batch must raise, not run some eager path."""
    from dgcnn_amd.model import Model
    m = Model(8, 2)
    b = synth.make_batch("MUTAG", 3)
    with pytest.raises(_lib.DgcnnError):
        m(b)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dgcnn_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
    src = open(os.path.join(ROOT, "model.py")).read()
    assert "oracle" not in src


def test_state_dict_roundtrip_with_reference_keys():
    from dgcnn_amd.model import Model
    from oracle.ref_ops import RefModel
    ref = RefModel(5, 2)
    m = Model(5, 2)
    m.load_state_dict(ref.state_dict())       # a reference-keyed checkpoint loads (train.py:129)
    assert set(m.state_dict().keys()) == set(ref.state_dict().keys())
    for k, v in ref.state_dict().items():
        assert torch.equal(m.state_dict()[k], v)


def test_flatten_parameters_views_cpu():
    from dgcnn_amd.model import Model
    m = Model(5, 2)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = m.flatten_parameters()
    assert m._is_flat()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    with torch.no_grad():
        flat.zero_()
    assert all(float(p.abs().sum()) == 0 for p in m.parameters())


# ---- host data logic ---------------------------------------------------------------------------
def test_collate_block_diagonal():
    g0 = Graph(torch.ones(3, 2), torch.tensor([[0, 1], [1, 2]]), 1)
    g1 = Graph(torch.zeros(2, 2), torch.tensor([[0], [1]]), 0)
    b = collate([g0, g1])
    assert b.num_graphs == 2 and b.num_nodes == 5
    assert b.batch.tolist() == [0, 0, 0, 1, 1]
    assert b.edge_index.tolist() == [[0, 1, 3], [1, 2, 4]]
    assert b.y.tolist() == [1, 0]


def test_indegree_feature_matches_reference_definition():
    # /root/reference/utils.py:18-33: in-degree on edge_index[1], / max, appended LAST
    ei = torch.tensor([[0, 1, 2, 2], [1, 0, 0, 1]])
    x = torch.tensor([[5.0], [6.0], [7.0]])
    out = indegree_feature(ei, 3, x)
    assert out.shape == (3, 2)
    assert torch.allclose(out[:, 1], torch.tensor([2.0, 2.0, 0.0]) / 2.0)
    assert torch.equal(out[:, 0], x[:, 0])
    assert indegree_feature(ei, 3, None).shape == (3, 1)


@pytest.mark.parametrize("name", ["MUTAG", "PROTEINS", "COLLAB", "COLLAB_REAL", "DD", "IMDB"])
def test_generator_shapes_and_determinism(name):
    sh = synth.SHAPES[name]
    b1 = synth.make_batch(name, 6, start=3)
    b2 = synth.make_batch(name, 6, start=3)
    assert torch.equal(b1.x, b2.x) and torch.equal(b1.edge_index, b2.edge_index)
    assert b1.x.shape[1] == sh.num_features and b1.x.dtype == torch.float32
    assert b1.edge_index.dtype == torch.int64 and int(b1.y.max()) < sh.num_classes
    src, dst = b1.edge_index
    assert not bool((src == dst).any())
    assert torch.equal(b1.batch[src], b1.batch[dst])          # block diagonal
    # undirected: the reversed edge set equals the edge set
    fw = set(zip(src.tolist(), dst.tolist()))
    assert fw == set(zip(dst.tolist(), src.tolist()))
    # degree column is last and normalised by the per-graph max
    assert float(b1.x[:, -1].max()) == 1.0 and float(b1.x[:, -1].min()) >= 0.0
    # any sub-range can be regenerated independently (data-parallel ranks rely on this)
    b3 = synth.make_batch(name, 2, start=5)
    n_first4 = int((b1.batch < 2).sum())
    assert torch.equal(b3.x, b1.x[n_first4:n_first4 + b3.num_nodes])


def test_dd_forced_large_graph():
    b = synth.make_batch("DD", 2, start=0, force_first_n=5748)
    assert int((b.batch == 0).sum()) == 5748


def test_split_batch_partitions_graphs():
    b = synth.make_batch("PROTEINS", 11, start=40)
    parts = split_batch(b, 4)
    assert sum(p.num_graphs for p in parts) == 11 and all(p.num_graphs >= 1 for p in parts)
    assert sum(p.num_nodes for p in parts) == b.num_nodes
    assert sum(p.num_edges for p in parts) == b.num_edges
    assert torch.equal(torch.cat([p.x for p in parts]), b.x)
    assert torch.equal(torch.cat([p.y for p in parts]), b.y)
    for p in parts:
        assert int(p.batch.min()) == 0 and int(p.batch.max()) == p.num_graphs - 1
        assert int(p.edge_index.min()) >= 0 and int(p.edge_index.max()) < p.num_nodes
    with pytest.raises(ValueError):
        split_batch(b, 12)


def test_host_side_of_the_library_under_address_sanitizer():
    """SURVEY §5 / VERDICT r1 hygiene: one -fsanitize=address pass over the host side of the C ABI (layout queries, argument
    validation, step-args plumbing, the host prefix sums of dgcnn_collate_ids), no GPU needed.  Builds the sanitizer
    variant of the library once (tools/build_variant.sh) and runs tools/asan_host_check.py under the ASan runtime."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang) or shutil.which("make") is None:
        pytest.skip("ROCm clang not present")
    rt = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("ASan runtime not present")
    lib = os.path.join(root, "dgcnn_amd", "variants", "lib_asan.so")
    srcs = [os.path.join(root, "dgcnn_amd", "csrc", f) for f in os.listdir(os.path.join(root, "dgcnn_amd", "csrc"))
            if f.endswith((".hip", ".h"))] + [os.path.join(root, "include", "dgcnn_hip.h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in srcs):
        r = subprocess.run(["bash", os.path.join(root, "tools", "build_variant.sh"), "asan", "-fsanitize=address -shared-libasan -g"],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and os.path.exists(lib), r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "asan_host_check.py"), lib], capture_output=True, text=True,
                       env=env, timeout=300)
    assert "ASAN_HOST_OK" in r.stdout and "AddressSanitizer" not in r.stderr, r.stdout[-1000:] + r.stderr[-3000:]


def test_forward_form_is_a_pure_function_of_host_known_numbers(built_lib):
    """dgcnn_forward_form (which kernel families a batch takes): no GPU involved.  The reference's batch of 50 COLLAB-shaped
    graphs takes the one-launch chain + readout training kernel over a gather backward; 2048 graphs the persistent chain
    forward over the dense/chain backward; every opt-out flag and every missing precondition falls back."""
    L = built_lib
    CU, DENSE, CHAIN, SPARSE, NOCHAIN, BF16, TILED = (_lib.FLAG_COALESCED_UNDIRECTED, _lib.FLAG_AGG_DENSE, _lib.FLAG_CHAIN,
                                                      _lib.FLAG_AGG_SPARSE, _lib.FLAG_NO_CHAIN, _lib.FLAG_BF16, _lib.FLAG_FORCE_TILED)
    f = L.dgcnn_forward_form
    N50, E50 = 3800, 140000
    EV = _lib.FORM_EVAL                                            # (round 5) forwards without an in-launch backward: one launch too
    assert f(N50, E50, 50, 1, CU, 180) == 2 | 4 | 8 | EV           # chain forward + one-launch training kernel + its in-kernel GCN backward
    assert f(40, 80, 50, 1, CU, 4) == 2 | 4 | EV                   # fewer 16-node tiles than graphs: no partial row per graph, round-3 form
    assert f(N50, E50, 50, 1, CU, 300) == 0                        # a graph above 256 nodes in a small batch: gather kernels
    assert f(N50, E50, 50, 1, CU | CHAIN, 300) == 2                # ... unless asked for (no one-launch kernel above 256 nodes)
    assert f(N50, E50, 50, 1, 0, 180) == 0                         # no coalesced + undirected promise: no bitmap
    assert f(N50, E50, 50, 1, CU, 0) == 0                          # no node bound
    assert f(N50, E50, 50, 90, CU, 180) == 0                       # F > 32: linear-first conv1, per-layer kernels
    assert f(N50, E50, 50, 1, CU | NOCHAIN, 180) == 0
    assert f(N50, E50, 50, 1, CU | SPARSE, 180) == 0
    assert f(N50, E50, 50, 1, CU | TILED, 180) == 0
    assert f(N50, E50, 50, 1, CU | DENSE, 180) == 1 | 2            # dense backward forced: chain forward, separate readout launches
    assert f(N50, E50, 50, 1, CU | BF16, 180) == 2 | 4 | 8 | EV    # bf16 leg: the same one-launch kernels (bf16 image in the forward half)
    prev = L.dgcnn_eval_kernel_enable(0)                           # the switch of the one-launch evaluation kernel (tests, A/B)
    try:
        assert prev == 1 and f(N50, E50, 50, 1, CU, 180) == 2 | 4 | 8
    finally:
        L.dgcnn_eval_kernel_enable(prev)
    # round 6 (ABI v20): forward-only use of a small batch with a graph of 257..512 nodes -> chain form + the one-launch evaluation
    # kernel (never the training forms); beyond 512 nodes or 256 graphs the flag changes nothing; switch at 2: chain-form batches too
    INF = _lib.FLAG_INFERENCE
    assert f(N50, E50, 50, 1, CU | INF, 300) == 2 | EV and f(N50, E50, 50, 1, CU | INF, 512) == 2 | EV
    assert f(N50, E50, 50, 1, CU | INF, 513) == 0 and f(N50, E50, 50, 1, CU | INF, 180) == f(N50, E50, 50, 1, CU, 180)
    assert f(N50, E50, 50, 90, CU | INF, 300) == 0 and f(N50, E50, 50, 1, INF, 300) == 0
    assert f(N50 * 6, E50 * 6, 300, 1, CU | INF, 300) == f(N50 * 6, E50 * 6, 300, 1, CU, 300)
    prev = L.dgcnn_eval_kernel_enable(2)
    try:
        assert f(N50, E50, 50, 1, CU | CHAIN, 300) == 2 | EV and f(N50, E50, 50, 1, CU, 300) == 0
        assert f(N50, E50, 50, 1, CU, 180) == 2 | 4 | 8 | EV
    finally:
        L.dgcnn_eval_kernel_enable(prev)
    N2k, E2k = 153000, 5700000
    assert f(N2k, E2k, 2048, 1, CU, 250) == 1 | 2
    assert f(N2k, E2k, 2048, 1, CU, 600) == 0                      # above the dense bound of 512 nodes
    assert f(N2k, E2k, 2048, 1, CU | NOCHAIN, 250) == 1
    assert f(0, 0, 1, 1, 0, 0) < 0 and f(10, 10, 1, 0, 0, 0) < 0   # bad sizes: error code


def test_bench_result_line_stays_parseable_whatever_the_repeat_count():
    """BENCH_r03 lost its line: 2 961 repeats were listed in front of everything else (27 KB) and the driver, which reads
    the tail of stdout, could not parse it.  The line is now a fixed-size summary: < 6 KB with 3 000 repeats and every
    optional object present, and it carries the contract fields, `roofline` and `cpu_baseline`."""
    import json
    import random
    import bench
    args = bench.parse(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    rnd = random.Random(0)
    reps = [20 * 50.5e-6 * (1 + 0.01 * rnd.random()) for _ in range(3000)]
    long_note = "x" * 400
    roof = {"bound": "hbm", "kernel": "k_chain_readout_tail", "achieved": 209.123, "peak": 8000.0, "unit": "GB/s", "frac": 0.02614,
            "model": "SURVEY 8(d) D4: 4 aggregation calls + sort-pool", "algorithmic_bytes_per_launch": 6590000, "avg_launch_us": 31.505,
            "avg_launch_us_kernel_trace": 31.4, "launches_measured": 400, "traffic": 12430000, "traffic_source": "live rocprofv3 --pmc",
            "frac_of_peak_on_measured_traffic": 0.0493, "frac_with_tail_model": 0.0401}
    cpu = {"value": 4142.7, "unit": "graphs/s", "cores": 32, "kind": "port", "value_1_thread": 1395.0, "ms_per_step": 12.07,
           "ms_per_step_by_threads": {str(t): 12.0 for t in (1, 4, 8, 16, 32, 64, 128, 256)}, "host_logical_cpus": 256, "sample": long_note}
    out = bench.build_result(args, reps, gb=50, world=1, strong=False, share=False, F=1, C=3, nb=40, Bavg=50.0, avgN=3831.2,
                             avgE=140837.5, exchange={"mode": "none", "note": ""}, loss_mean=1.0986, correct_frac=0.3333,
                             extra={"fwd_bwd_only_graphs_per_s_rank0": 1.0e6, "pmc_note": long_note[:200]},
                             roofline=roof, roofline_large=dict(roof, batch=2048, step_ms=0.268, graphs_per_s=7.6e6),
                             cpu=cpu, dropin={"unchanged_loop_us": 470.0, "with_dgcnn_amd_optim_adam_us": 250.0, "steps": 200})
    line = json.dumps(out)
    assert len(line) < bench.RESULT_LINE_MAX == 6000, len(line)
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["repeats"] == 3000 and back["repeats_ms_per_step"]["count"] == 3000
    assert isinstance(back["repeats_ms_per_step"], dict)          # a summary, never the list
    assert abs(back["ms_per_step"] - 0.0507) < 1e-3
    assert "workload" in back["config"] and "model" not in back["config"]
