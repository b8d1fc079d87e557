"""ctypes binding of libdgcnn_hip.so (the C ABI declared in include/dgcnn_hip.h).

The shared library is built IN-TREE next to this file (``dgcnn_amd/libdgcnn_hip.so``) by
``dgcnn_amd/csrc/Makefile`` (``hipcc --offload-arch=gfx950``); ``build()`` runs that.
There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGCNN_HIP_LIB") or os.path.join(_HERE, "libdgcnn_hip.so")   # env override: A/B experiments
CSRC = os.path.join(_HERE, "csrc")
ABI_VERSION = 20
FLAG_COALESCED_UNDIRECTED = 1
FLAG_FORCE_FUSED = 2
FLAG_FORCE_TILED = 4
FLAG_PREPARED = 8
FLAG_AGG_SPARSE = 16      # never use the dense per-graph block aggregation
FLAG_AGG_DENSE = 32       # use it whenever the batch admits it (coalesced_undirected, max_nodes <= 512)
FLAG_CHAIN = 128          # graph-chain kernels (conv1..conv4 of a graph inside one workgroup) whenever admissible
FLAG_NO_CHAIN = 256       # never
FLAG_EXCLUSIVE_DEVICE = 512   # pipelined steps: nothing else runs on this device (admits the in-launch wait of the fused preparation)
FLAG_INFERENCE = 1024     # forward-only use of the batch: one-launch evaluation kernel also for graphs of 257..512 nodes (ABI v20)
FLAG_BF16 = 64            # bf16 leg: pre-scaled linear outputs stored bf16, X.W on the bf16 matrix cores

FORM_DENSE, FORM_CHAIN, FORM_CHAIN_TAIL, FORM_STEP, FORM_EVAL = 1, 2, 4, 8, 16      # dgcnn_forward_form bits

K = 30
CAT = 97
HID1 = 128
FLAT = 352
NUM_SEG = 16

class DenseView(ctypes.Structure):
    """``dgcnn_dense_view`` of include/dgcnn_hip.h"""
    _fields_ = [("B", ctypes.c_int32), ("reserved_", ctypes.c_int32), ("graph_ptr", c_void_p), ("item_table", c_void_p),
                ("adj_bits", c_void_p)]


class StepArgs(ctypes.Structure):
    """``dgcnn_step_args`` of include/dgcnn_hip.h (field order and types must match)."""
    _fields_ = [("N", ctypes.c_int32), ("E", ctypes.c_int32), ("B", ctypes.c_int32), ("F", ctypes.c_int32),
                ("C", ctypes.c_int32), ("training", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("max_nodes", ctypes.c_int32), ("max_edges", ctypes.c_int32), ("epoch", ctypes.c_uint32),
                ("seed", c_uint64), ("step", c_int64), ("lr", c_float), ("beta1", c_float), ("beta2", c_float),
                ("eps", c_float), ("loss_scale", c_float), ("reserved_", ctypes.c_int32),
                ("params", c_void_p), ("x", c_void_p), ("edge_index", c_void_p), ("batch", c_void_p),
                ("y", c_void_p), ("ws", c_void_p), ("logp", c_void_p), ("grads", c_void_p), ("metrics", c_void_p),
                ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("ds", c_void_p), ("ds_ids", c_void_p), ("ds_onode", c_void_p), ("ds_oedge", c_void_p)]


class Dataset(ctypes.Structure):
    """``dgcnn_dataset`` of include/dgcnn_hip.h: a dataset's graph structures prepared once"""
    _fields_ = [("G", c_int64), ("Ntot", c_int64), ("Etot", c_int64), ("F", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("node_ptr", c_void_p), ("y", c_void_p), ("x", c_void_p), ("rowptr", c_void_p), ("colidx", c_void_p),
                ("dinv", c_void_p), ("xs", c_void_p), ("adj_bits", c_void_p)]


# name -> (restype, argtypes); must list every symbol include/dgcnn_hip.h declares
SIGNATURES = {
    "dgcnn_model_eval_step": (c_int, [ctypes.POINTER(StepArgs), c_void_p]),
    "dgcnn_pipeline_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "dgcnn_pipeline_destroy": (c_int, [c_void_p]),
    "dgcnn_pipeline_train_step": (c_int, [c_void_p, ctypes.POINTER(StepArgs), ctypes.POINTER(StepArgs), c_void_p]),
    "dgcnn_pipeline_eval_step": (c_int, [c_void_p, ctypes.POINTER(StepArgs), ctypes.POINTER(StepArgs), c_void_p]),
    "dgcnn_version": (c_int, []),
    "dgcnn_eval_kernel_enable": (c_int, [c_int]),
    "dgcnn_param_layout": (c_int64, [c_int, c_int, ctypes.POINTER(c_int64)]),
    "dgcnn_workspace_bytes": (c_int64, [c_int] * 5),
    "dgcnn_workspace_offset": (c_int64, [c_char_p] + [c_int] * 5),
    "dgcnn_graph_prep": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int] + [c_void_p] * 8 + [c_int, c_void_p, c_void_p,
                                 c_void_p]),
    "dgcnn_dense_table_ints": (c_int64, [c_int, c_int]),
    "dgcnn_dense_bitmap_words": (c_int64, [c_int]),
    "dgcnn_gcn_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                              c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "dgcnn_gcn_bwd_scratch_bytes": (c_int64, [c_int, c_int, c_int]),
    "dgcnn_gcn_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                              c_int64, c_void_p]),
    "dgcnn_sortpool_fwd": (c_int, [c_int, c_int] + [c_void_p] * 7 + [c_void_p]),
    "dgcnn_sortpool_bwd": (c_int, [c_int, c_int] + [c_void_p] * 7 + [c_void_p]),
    "dgcnn_model_forward": (c_int, [c_int] * 5 + [c_void_p] * 6 + [c_int, c_uint64, c_int, c_int, c_int,
                                    ctypes.c_uint32, c_void_p]),
    "dgcnn_debug_phase_clocks": (c_int, [c_void_p]),
    "dgcnn_model_prepare": (c_int, [c_int] * 5 + [c_void_p] * 4 + [c_int, c_int, ctypes.c_uint32, c_void_p]),
    "dgcnn_forward_form": (c_int, [c_int] * 6),
    "dgcnn_step_kernel_enable": (c_int, [c_int]),
    "dgcnn_narrow_gather_enable": (c_int, [c_int]),
    "dgcnn_fused_max_nodes": (c_int, [c_int]),
    "dgcnn_fused_fits": (c_int, [c_int, c_int, c_int]),
    "dgcnn_model_backward": (c_int, [c_int] * 5 + [c_void_p] * 6 + [c_float, c_int, c_void_p, c_void_p, c_int, c_int,
                                     c_void_p]),
    "dgcnn_model_backward_step": (c_int, [c_int] * 5 + [c_void_p] * 5 + [c_float, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int, c_int, c_void_p]),
    "dgcnn_adam_step": (c_int, [c_void_p] * 4 + [c_int64, c_int64, c_float, c_float, c_float, c_float, c_int,
                                c_void_p]),
    "dgcnn_collate": (c_int, [c_int, c_int, c_int64, c_int64, c_int64] + [c_void_p] * 13),
    "dgcnn_collate_ids": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64]
                          + [c_void_p] * 5 + [c_int64, c_int64] + [c_void_p] * 5 + [c_void_p]),
    "dgcnn_dataset_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dgcnn_assemble": (c_int, [c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 7 + [c_int, c_int, ctypes.c_uint32,
                               c_void_p]),
    "dgcnn_accumulate_metrics": (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dgcnn_peer_alloc": (c_int, [c_int64, ctypes.POINTER(c_void_p), c_void_p]),
    "dgcnn_peer_open": (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "dgcnn_peer_set_timeout_ms": (c_int, [c_int]),
    "dgcnn_peer_last_alloc_finegrained": (c_int, []),
    "dgcnn_peer_close": (c_int, [c_void_p]),
    "dgcnn_peer_free": (c_int, [c_void_p]),
    "dgcnn_allreduce_adam_step": (c_int, [c_int, c_int, c_void_p, c_void_p, ctypes.c_uint32, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_int64, c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "dgcnn_profile_next_forward": (c_int, [c_int, c_void_p, c_void_p]),
    "dgcnn_event_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "dgcnn_event_record": (c_int, [c_void_p, c_void_p]),
    "dgcnn_event_elapsed_ms": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_float)]),
    "dgcnn_event_destroy": (c_int, [c_void_p]),
}

_ERR = {-1: "DGCNN_EINVAL (bad size / null pointer)", -2: "DGCNN_ELAUNCH (HIP launch error)",
        -3: "DGCNN_EUNSUPPORTED (shape outside this build)"}


class DgcnnError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libdgcnn_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    # `racedelay`: the test build of the persistent chain kernels (tests/test_gpu_chain.py), next to the product library
    cmd = ["make", "-C", CSRC, "-j4", "all", "racedelay"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise DgcnnError(f"building {LIB_PATH} failed (exit {res.returncode}):\n{res.stdout[-4000:]}")
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the library; raises loudly when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DgcnnError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            f"Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C {CSRC}`.")
    import torch  # noqa: F401  -- load PyTorch-ROCm's libamdhip64.so.7 FIRST so both share one HIP runtime
    L = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise DgcnnError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    v = L.dgcnn_version()
    if v != ABI_VERSION:
        raise DgcnnError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}; rebuild")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DgcnnError(f"{what} failed: {_ERR.get(rc, rc)}")


def param_layout(F: int, C: int):
    """(offsets[16], padded_total) of the flat parameter buffer, straight from the C ABI."""
    arr = (c_int64 * NUM_SEG)()
    total = lib().dgcnn_param_layout(F, C, arr)
    if total < 0:
        raise DgcnnError(f"dgcnn_param_layout(F={F}, C={C}) failed: {_ERR.get(total, total)}")
    return list(arr), int(total)


def workspace_bytes(N: int, E: int, B: int, F: int, C: int) -> int:
    n = lib().dgcnn_workspace_bytes(N, E, B, F, C)
    if n < 0:
        raise DgcnnError(f"dgcnn_workspace_bytes failed: {_ERR.get(n, n)}")
    return int(n)


def workspace_offset(name: str, N: int, E: int, B: int, F: int, C: int) -> int:
    o = lib().dgcnn_workspace_offset(name.encode(), N, E, B, F, C)
    if o < 0:
        raise DgcnnError(f"unknown workspace region {name!r}")
    return int(o)


# (region, dtype-name, shape-lambda) for tests/tools that look inside the workspace
def ws_view(ws, name: str, N: int, E: int, B: int, F: int, C: int):
    """Typed torch view of a named workspace region (tests and tools)."""
    import torch
    shapes = {
        "err": (torch.int32, (8,)), "rowptr": (torch.int32, (N + 1,)), "rowptr_t": (torch.int32, (N + 1,)),
        "colidx": (torch.int32, (E,)), "colidx_t": (torch.int32, (E,)), "dinv": (torch.float32, (N,)),
        "graph_ptr": (torch.int32, (B + 1,)),
        "x1": (torch.float32, (N, 32)), "x2": (torch.float32, (N, 32)), "x3": (torch.float32, (N, 32)),
        "x4": (torch.float32, (N,)), "perm": (torch.int32, (B, K)), "pooled": (torch.float32, (B, K * CAT)),
        "a5": (torch.float32, (B, 16, K)), "a6": (torch.float32, (B, FLAT)), "a1d": (torch.float32, (B, HID1)),
        "drop_mask": (torch.uint8, (B, HID1)), "dlogit": (torch.float32, (B, C)),
        "gp1": (torch.float32, (N, 32)), "gp2": (torch.float32, (N, 32)), "gp3": (torch.float32, (N, 32)),
        "gas4": (torch.float32, (N,)), "lossv": (torch.float32, (B, 2)), "ax": (torch.float32, (N, F)),
        "adjbits": (torch.int32, (31 * N,)), "dmap": (torch.int32, (3096 + 3 * (N // 128 + B + 1) + 2 * B + 4,)),
    }
    dt, shape = shapes[name]
    off = workspace_offset(name, N, E, B, F, C)
    numel = 1
    for s in shape:
        numel *= s
    nbytes = numel * torch.empty(0, dtype=dt).element_size()
    return ws[off:off + nbytes].view(dt).view(shape)
