"""Run the Python host layer (dgcnn_amd.Model / Trainer) on top of emu/libdgcnn_emu.so -- the kernel sources compiled as plain C++
against the CPU SIMT emulation of emu/include/hip/hip_runtime.h -- so that device ARITHMETIC AND INDEXING can be checked against
the oracle without a GPU (round 6: the GPU pool was closed to this repository).  TEST INFRASTRUCTURE: the patches below live for
the duration of the ``emulated()`` context only; the product keeps refusing CPU tensors (tests/test_abi_and_host.py checks that).

What the emulation does not show: timing, memory ordering, data races (one lane runs at a time).  Its semantics of the wave-level
instructions (MFMA operand layouts, DPP controls, the LDS transpose read) are pinned by the kernels with green GPU records from
rounds 1-5, which reproduce the fp64 oracle under it (tests/test_emu_kernels.py, "calibration" cases)."""
from __future__ import annotations

import contextlib
import ctypes
import os
import subprocess
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "emu")
EMU_LIB = os.environ.get("DGCNN_EMU_LIB") or os.path.join(EMU_DIR, "libdgcnn_emu.so")      # (env: a variant build, see emu/Makefile)


def build_emu(verbose: bool = False) -> str:
    if os.environ.get("DGCNN_EMU_LIB"):
        return EMU_LIB
    import fcntl
    with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lk:      # (xdist workers all arrive here: one make at a time)
        fcntl.flock(lk, fcntl.LOCK_EX)
        res = subprocess.run(["make", "-C", EMU_DIR, "-j8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
    if res.returncode != 0 or not os.path.exists(EMU_LIB):
        raise RuntimeError(f"building {EMU_LIB} failed:\n{res.stdout[-4000:]}")
    return EMU_LIB


_EMU = None


def emu_lib() -> ctypes.CDLL:
    global _EMU
    if _EMU is None:
        from dgcnn_amd import _lib
        build_emu()
        L = ctypes.CDLL(EMU_LIB)
        for name, (res, args) in _lib.SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        assert L.dgcnn_version() == _lib.ABI_VERSION
        _EMU = L
    return _EMU


@contextlib.contextmanager
def emulated():
    """inside: dgcnn_amd's ctypes calls go to the emulation library and CPU tensors are accepted as 'device' buffers"""
    from dgcnn_amd import _lib, batch as batch_mod, model as model_mod
    L = emu_lib()
    saved = (_lib._lib, torch.cuda.current_stream, torch._C._cuda_getCurrentRawStream, torch.cuda.synchronize,
             model_mod.Model._check_inputs, batch_mod.Batch.to)
    _lib._lib = L
    torch.cuda.current_stream = lambda device=None: types.SimpleNamespace(cuda_stream=0)
    torch._C._cuda_getCurrentRawStream = lambda idx=None: 0
    torch.cuda.synchronize = lambda device=None: None

    def check_inputs(x, edge_index, batch):
        if x.dtype != torch.float32 or x.dim() != 2:
            raise _lib.DgcnnError(f"data.x must be [N,F] float32, got {x.dtype} {tuple(x.shape)}")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise _lib.DgcnnError("data.edge_index must be [2,E] int64")
        if batch.dtype != torch.int64 or batch.shape[0] != x.shape[0]:
            raise _lib.DgcnnError("data.batch must be [N] int64")

    model_mod.Model._check_inputs = staticmethod(check_inputs)
    batch_mod.Batch.to = lambda self, device: self
    try:
        yield L
    finally:
        (_lib._lib, torch.cuda.current_stream, torch._C._cuda_getCurrentRawStream, torch.cuda.synchronize,
         chk, bto) = saved
        model_mod.Model._check_inputs = staticmethod(chk)
        batch_mod.Batch.to = bto


def read_metrics(tr):
    """Trainer.read_metrics + the error words of every workspace slot in use (on a CPU 'device' read_metrics skips them)"""
    slots = [(sl["ws"], sl["dims"]) for sl in tr._slots if sl.get("dims") and sl["ws"] is not None]
    if slots:
        tr.model.check_errors(slots, since=tr._err_checked)
        tr._err_checked = tr.model._epoch
    return tr.read_metrics()


# ---- whole-session mode (DGCNN_EMU=1 python -m pytest tests -m gpu ...): the GPU tests themselves on the emulation ---------------
_CUDA_NAMES = ("cuda", "cuda:0")
_SESSION = None


def _is_cuda_dev(d) -> bool:
    if isinstance(d, str):
        return d in _CUDA_NAMES
    return isinstance(d, torch.device) and d.type == "cuda"


def install_global() -> None:
    """map every request for the device "cuda" to the CPU and route dgcnn_amd to the emulation library, for the rest of the process"""
    global _SESSION
    if _SESSION is not None:
        return
    _SESSION = emulated()                            # (kept alive: a collected generator would run its `finally` and undo the patches)
    _SESSION.__enter__()
    t_to = torch.Tensor.to

    def to(self, *args, **kw):
        args = tuple("cpu" if _is_cuda_dev(a) else a for a in args)
        if _is_cuda_dev(kw.get("device")):
            kw["device"] = "cpu"
        return t_to(self, *args, **kw)

    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.cuda.manual_seed_all = lambda seed: None
    torch.cuda.current_device = lambda: 0
    for name in ("empty", "zeros", "ones", "full", "randn", "rand", "randint", "tensor", "arange", "empty_like", "zeros_like",
                 "ones_like", "full_like", "as_tensor", "from_numpy", "linspace", "eye", "randperm"):
        fn = getattr(torch, name)

        def wrap(fn=fn):
            def f(*a, **k):
                if _is_cuda_dev(k.get("device")):
                    k["device"] = "cpu"
                return fn(*a, **k)
            return f
        setattr(torch, name, wrap())
    m_to = torch.nn.Module.to

    def mod_to(self, *args, **kw):
        args = tuple("cpu" if _is_cuda_dev(a) else a for a in args)
        if _is_cuda_dev(kw.get("device")):
            kw["device"] = "cpu"
        return m_to(self, *args, **kw)

    torch.nn.Module.to = mod_to
    torch.nn.Module.cuda = lambda self, *a, **k: self
